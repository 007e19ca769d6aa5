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

WITH_OPT = bool(os.environ.get("HOST_OPT"))        # the optimizer-inclusive iteration (bench field with_optimizer)
AUTOGRAD = bool(os.environ.get("HOST_AUTOGRAD"))   # render_with_l1 + loss.backward() instead of render_l1_step
if WITH_OPT:
    reg = importlib.import_module("4dgaussians_amd.scene.regulation"); om = importlib.import_module("4dgaussians_amd.optim")
    opt = om.FusedAdamW([{"params": [pc._xyz], "lr": 0.00016}, {"params": list(deform.get_mlp_parameters()), "lr": 0.00016},
                         {"params": list(deform.get_grid_parameters()), "lr": 0.0016}, {"params": [pc._features], "lr": 0.0025},
                         {"params": [pc._opacity], "lr": 0.05}, {"params": [pc._scaling], "lr": 0.005},
                         {"params": [pc._rotation], "lr": 0.001}], lr=0.0001, eps=1e-8)


def step():
    # the bench's step: render + L1 in one library call per direction, issued back to back (render_l1_step) or through autograd
    if AUTOGRAD:
        R.render_with_l1(cam, pc, deform, bg, target)[0].backward()
    else:
        R.render_l1_step(cam, pc, deform, bg, target)
    if WITH_OPT:
        reg.compute_regulation(deform, 0.01, 0.0001, 0.0, 0.0001, detail=False)["total_reg"].backward()
        opt.step()
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
if AUTOGRAD:
    tf = tb = 0.0
    for _ in range(200):
        a = time.perf_counter()
        loss = R.render_with_l1(cam, pc, deform, bg, target)[0]
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
