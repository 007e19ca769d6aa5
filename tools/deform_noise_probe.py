"""Diagnostic: float32 noise floor of the deformation gradients.  Runs the CPU oracle in float32 and in
float64 and the HIP path on the same inputs, and prints max errors relative to each tensor's max."""
import os, sys, types, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import pkg
from test_gpu_deformation import make_args
from oracle import deformation_ref as R

cfg = R.DeformConfig(defor_depth=0, plane_channels=32, resolution=[64, 64, 64, 75], multires=[1, 2])
torch.manual_seed(3)
net = pkg("scene.deformation").deform_network(make_args(cfg))
with torch.no_grad():
    for n, p in net.named_parameters():
        if "deform" in n and "bias" in n:
            p.add_(0.02 * torch.randn_like(p))
sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
P = 20000
g = torch.Generator().manual_seed(5)
ins = dict(points=torch.rand(P, 3, generator=g) * 3.4 - 1.7, scales=torch.randn(P, 3, generator=g) * 0.5 - 3,
           rotations=torch.randn(P, 4, generator=g), opacity=torch.randn(P, 1, generator=g),
           shs=torch.randn(P, 16, 3, generator=g) * 0.3)
time, force = torch.full((P, 1), 0.61), torch.zeros(P, 1)
ws = [torch.randn(s, generator=g) for s in ((P, 3), (P, 3), (P, 4), (P, 1), (P, 16, 3))]

def run_oracle(dt):
    sdo = {k: (v.to(dt) if v.is_floating_point() else v).clone().requires_grad_(v.dtype == torch.float32 and "poc" not in k and "aabb" not in k) for k, v in sd.items()}
    xo = {k: v.to(dt).clone().requires_grad_(True) for k, v in ins.items()}
    want = R.forward_dynamic(sdo, cfg, xo["points"], xo["scales"], xo["rotations"], xo["opacity"], xo["shs"], time.to(dt), force.to(dt))
    sum((o * w.to(dt)).sum() for o, w in zip(want, ws)).backward()
    out = {"d" + k: v.grad for k, v in xo.items() if v.grad is not None}
    out.update({k: v.grad for k, v in sdo.items() if v.grad is not None})
    return out

o32, o64 = run_oracle(torch.float32), run_oracle(torch.float64)
net = net.cuda()
xg = {k: v.cuda().requires_grad_(True) for k, v in ins.items()}
got = net.forward_dynamic(xg["points"], xg["scales"], xg["rotations"], xg["opacity"], xg["shs"], time.cuda(), force.cuda(), None)
sum((o * w.cuda()).sum() for o, w in zip(got, ws)).backward()
hip = {"d" + k: v.grad for k, v in xg.items() if v.grad is not None}
hip.update({k: p.grad for k, p in net.named_parameters() if p.grad is not None})
rel = lambda a, b: float((a.double().cpu() - b.double()).abs().max()) / max(float(b.abs().max()), 1e-20)
print("%-50s %12s %12s %12s" % ("tensor", "hip-vs-f64", "f32-vs-f64", "hip-vs-f32"))
for k in o64:
    if k in hip:
        print("%-50s %12.3e %12.3e %12.3e" % (k[-50:], rel(hip[k], o64[k]), rel(o32[k], o64[k]), rel(hip[k], o32[k])))
for k in ("drotations", "dpoints"):
    e = (hip[k].double().cpu() - o32[k].double()).abs() / float(o32[k].abs().max())
    rows = (e.max(dim=1).values > 2e-5).nonzero().flatten()
    print(k, "rows with err > 2e-5:", rows.numel(), rows[:20].tolist(), "max at", int(e.max(dim=1).values.argmax()))
k = "deformation_net.rotations_deform.1.weight"
e = (hip[k].double().cpu() - o32[k].double()).abs() / float(o32[k].abs().max())
print(k, "cols with err > 2e-5:", (e.max(dim=0).values > 2e-5).nonzero().flatten().tolist())
print("rows:", (e.max(dim=1).values > 2e-5).nonzero().flatten().tolist())
