"""Host time of FusedAdamW.step() for a model-sized parameter set (60 tensors): the optimizer's share of the time the
Python side needs to issue an iteration."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
om = importlib.import_module("4dgaussians_amd.optim")
ps = [torch.nn.Parameter(torch.randn(64, 64, device="cuda")) for _ in range(60)]
for p in ps:
    p.grad = torch.randn_like(p)
opt = om.FusedAdamW([{"params": ps[:30], "lr": 1e-3}, {"params": ps[30:], "lr": 1e-4}])
for _ in range(5):
    opt.step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200):
    opt.step()
host = (time.perf_counter() - t) / 200
torch.cuda.synchronize()
print(f"FusedAdamW.step(): {host * 1e6:.0f} us of host time per call (60 tensors, 2 groups)")
