"""Host cost of one training step: the C2 step's Python / launch path on a scene so small that the GPU is never the
limit (P = 2000, 64x64), timed as wall clock per step and profiled (cProfile sees the main thread: forward + the
engine call; the autograd thread's share is the rest)."""
import cProfile, importlib, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
syn = importlib.import_module("4dgaussians_amd.synthetic"); R = importlib.import_module("4dgaussians_amd.renderer")
Dm = importlib.import_module("4dgaussians_amd.scene.deformation"); F = importlib.import_module("4dgaussians_amd.fused")
dev = torch.device("cuda")
_, _, raw = syn.make_workload("C2", P=2000)
pc = R.GaussianParams(raw, dev)
Wl = importlib.import_module("4dgaussians_amd.workload")
deform = Dm.deform_network(Wl.deform_args("C2")); deform.deformation_net.set_aabb([1.3] * 3, [-1.3] * 3); deform = deform.to(dev)
cam = R.Camera(syn.look_at_camera(64, 64, azimuth=0.7), dev, time=0.37)
bg = torch.ones(3, device=dev); target = torch.rand(3, 64, 64, device=dev)
params = list(pc.parameters()) + list(deform.parameters())

def step():
    image, *_ = R.render(cam, pc, deform, bg)
    F.l1_loss(image, target).backward()
    for p in params:
        p.grad = None

for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print(f"host-bound step: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms")
tf = tb = 0.0
for _ in range(200):
    a = time.perf_counter()
    image, *_ = R.render(cam, pc, deform, bg)
    loss = F.l1_loss(image, target)
    b = time.perf_counter()
    loss.backward()
    c = time.perf_counter()
    for p in params:
        p.grad = None
    tf += b - a; tb += c - b
print(f"forward {tf / 200 * 1e3:.3f} ms, backward {tb / 200 * 1e3:.3f} ms")
prof = cProfile.Profile(); prof.enable()
for _ in range(200):
    step()
prof.disable()
pstats.Stats(prof).sort_stats("cumtime").print_stats(45)
