"""Host-side timeline of one C2 training step (diagnostic): where does the wall time go between
the stage kernels?  Prints host milliseconds per segment, with and without device syncs."""
import importlib, sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
syn = importlib.import_module("4dgaussians_amd.synthetic")
R = importlib.import_module("4dgaussians_amd.renderer")
Dm = importlib.import_module("4dgaussians_amd.scene.deformation")
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
S = torch.cuda.synchronize

def step(sync_each):
    t = [time.perf_counter()]
    image, *_ = R.render(cam, pc, deform, bg)
    if sync_each: S()
    t.append(time.perf_counter())
    loss = (image - target).abs().mean()
    if sync_each: S()
    t.append(time.perf_counter())
    loss.backward()
    if sync_each: S()
    t.append(time.perf_counter())
    for p in params: p.grad = None
    S()
    t.append(time.perf_counter())
    return np.diff(t) * 1e3

for mode in (False, True):
    for _ in range(5): step(mode)
    acc = np.mean([step(mode) for _ in range(30)], axis=0)
    print("sync_each" if mode else "async    ", "render %.3f  loss %.3f  backward %.3f  tail %.3f  total %.3f ms" % (*acc, acc.sum()))

# raw host cost of the torch elementwise part: same ops without any custom kernels
x = torch.randn(100000, 3, device=dev, requires_grad=True)
for _ in range(3):
    S(); t0 = time.perf_counter()
    for _ in range(100):
        y = torch.exp(x)
    S(); print("100 torch.exp launches: %.3f ms" % ((time.perf_counter() - t0) * 1e3))
