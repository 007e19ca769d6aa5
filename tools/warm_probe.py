"""Diagnostic: wall time of each of the first training steps (each step synchronised), to see how long the
process takes to reach steady state (allocator, clocks, code-object loading)."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
syn = importlib.import_module("4dgaussians_amd.synthetic"); R = importlib.import_module("4dgaussians_amd.renderer")
Dm = importlib.import_module("4dgaussians_amd.scene.deformation"); F = importlib.import_module("4dgaussians_amd.fused")
dev = torch.device("cuda", 0); cfg = syn.CONFIGS["C2"]; _, cam0, raw = syn.make_workload("C2"); W, H = cfg["W"], cfg["H"]
pc = R.GaussianParams(raw, dev); deform = Dm.deform_network(bench.deform_args("C2")); deform.deformation_net.set_aabb([1.3]*3, [-1.3]*3); deform = deform.to(dev)
cam = R.Camera(syn.look_at_camera(W, H), dev, time=0.3); bg = torch.ones(3, device=dev); target = torch.rand(3, H, W, device=dev)
params = list(pc.parameters()) + list(deform.parameters())
def step():
    image, *_ = R.render(cam, pc, deform, bg); F.l1_loss(image, target).backward()
    for p in params: p.grad = None
import gc
if os.environ.get("NOGC"): gc.disable()
ts = []
for i in range(120):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join("%.2f" % t for t in ts))
t0 = time.perf_counter()
for i in range(50): step()
torch.cuda.synchronize(); print("async 50: %.3f ms/step" % ((time.perf_counter() - t0) / 50 * 1e3))
time.sleep(0.5)
t0 = time.perf_counter()
for i in range(10): step()
torch.cuda.synchronize(); print("after 0.5 s idle, async 10: %.3f ms/step" % ((time.perf_counter() - t0) / 10 * 1e3))
