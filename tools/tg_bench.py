"""Kernel experiment: time of rigidity.transpose_graph at the fine stage's size (n = 100k points, k = 100 random targets)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rig = importlib.import_module("4dgaussians_amd.rigidity")
n, k = 100_000, 100
idx = torch.randint(0, n, (1, n, k), dtype=torch.int32, device="cuda")
for _ in range(3):
    rig.transpose_graph(idx, trusted=True)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    rig.transpose_graph(idx, trusted=True)
b.record(); torch.cuda.synchronize()
print(f"transpose_graph n={n} k={k}: {a.elapsed_time(b) / 20 * 1e3:.1f} us per call")
