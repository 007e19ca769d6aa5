"""Helper of test_gpu_mfma_modes.py (not collected by pytest): runs the deformation field of this process's GS4D_MFMA mode
(read once per process by the library) on the reference-generated golden cases and prints, per case, the largest output
error (abs, relative to max(1, |x|max)) and the largest gradient error (relative to the tensor's max) together with the
tensors they occur on."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa: F401  (puts the repository root on sys.path)


def run():
    import torch
    import test_gpu_deformation as T
    from golden_utils import CASES, load_case
    out = {"mode": os.environ.get("GS4D_MFMA", "f32"), "cases": {}}
    for case in CASES:
        cfg, sd, ins, outs, ws, gins, grads = load_case(case)
        net = T.build_net(cfg, sd)
        x = {k: ins[k].cuda().requires_grad_(True) for k in ("points", "scales", "rotations", "opacity", "shs")}
        got = net.forward_dynamic(x["points"], x["scales"], x["rotations"], x["opacity"], x["shs"], ins["time"].cuda(),
                                  ins["force"].cuda(), None)
        fwd = {}
        for n, g in zip(T.ORDER, got):
            want = outs[n].double()
            fwd[n] = float((g.detach().cpu().double() - want).abs().max()) / max(1.0, float(want.abs().max()))
        sum((g * ws[n].cuda()).sum() for n, g in zip(T.ORDER, got)).backward()
        bwd = {}
        for k, want in gins.items():
            if want.numel():
                bwd["d" + k] = float((x[k].grad.cpu().double() - want.double()).abs().max()) / max(float(want.abs().max()), 1e-20)
        params = dict(net.named_parameters())
        for k, want in grads.items():
            bwd[k] = float((params[k].grad.cpu().double() - want.double()).abs().max()) / max(float(want.abs().max()), 1e-20)
        wf, wb = max(fwd, key=fwd.get), max(bwd, key=bwd.get)
        out["cases"][case] = {"forward": fwd[wf], "forward_on": wf, "backward": bwd[wb], "backward_on": wb}
    return out


if __name__ == "__main__":
    print("RESULT " + json.dumps(run()))
