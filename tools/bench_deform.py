"""Times the deformation field alone (forward, forward+backward) at a BASELINE config's shape on cuda:0.
usage: python tools/bench_deform.py [C2|C5] [P]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
Wl = importlib.import_module("4dgaussians_amd.workload")
D = importlib.import_module("4dgaussians_amd.scene.deformation")
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
P = int(sys.argv[2]) if len(sys.argv) > 2 else {"C2": 100_000, "C5": 1_000_000}[name]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = D.deform_network(Wl.deform_args(name))
net.deformation_net.set_aabb([1.3] * 3, [-1.3] * 3)
net = net.to(dev)
x = [torch.rand(P, 3, device=dev) * 2.6 - 1.3, torch.randn(P, 3, device=dev) - 3, torch.randn(P, 4, device=dev), torch.randn(P, 1, device=dev),
     torch.randn(P, 16, 3, device=dev) * 0.3]
if os.environ.get("SORTED"):
    q = ((x[0] + 1.3) / 2.6 * 1023).clamp(0, 1023).long()
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    order = torch.argsort(code)
    x = [t[order].contiguous() for t in x]
for t in x:
    t.requires_grad_(True)
tm, fc = torch.full((P, 1), 0.4, device=dev), torch.zeros(P, 1, device=dev)
gs = [torch.randn_like(x[0]), torch.randn_like(x[1]), torch.randn_like(x[2]), torch.randn_like(x[4])]

def fwd():
    with torch.no_grad():
        net.forward_dynamic(*x, tm, fc, None)

def fwdbwd():
    o = net.forward_dynamic(*x, tm, fc, None)
    torch.autograd.backward([o[0], o[1], o[2], o[4]], gs)
    for p in list(net.parameters()) + x:
        p.grad = None

for fn, label in ((fwd, "forward (no grad)"), (fwdbwd, "forward + backward")):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{name} P={P} {label}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms")
