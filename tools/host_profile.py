"""Host-side cost of one C2 training step: cProfile over the step function of bench.py's workload
(diagnostic; prints the heaviest Python/C entry points by cumulative time)."""
import cProfile, importlib, io, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
syn = importlib.import_module("4dgaussians_amd.synthetic")
R = importlib.import_module("4dgaussians_amd.renderer")
Dm = importlib.import_module("4dgaussians_amd.scene.deformation")
F = importlib.import_module("4dgaussians_amd.fused")
dev = torch.device("cuda", 0)
cfg = syn.CONFIGS["C2"]
_, cam0, raw = syn.make_workload("C2")
W, H = cfg["W"], cfg["H"]
torch.manual_seed(0)
pc = R.GaussianParams(raw, dev)
deform = Dm.deform_network(bench.deform_args("C2"))
deform.deformation_net.set_aabb([1.3] * 3, [-1.3] * 3)
deform = deform.to(dev)
cam = R.Camera(syn.look_at_camera(W, H), dev, time=0.3)
bg = torch.ones(3, device=dev)
target = torch.rand(3, H, W, device=dev)
params = list(pc.parameters()) + list(deform.parameters())

def step():
    image, *_ = R.render(cam, pc, deform, bg)
    loss = F.l1_loss(image, target)
    loss.backward()
    for p in params:
        p.grad = None

for _ in range(10): step()
torch.cuda.synchronize()
N = 100
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/step, wall %.3f ms/step" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(N): step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
