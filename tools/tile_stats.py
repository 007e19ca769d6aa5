"""Diagnostic: per-tile list lengths of the C2 bench scene (after calibration) -- how long is the serial
chain of the heaviest tiles compared with the average?"""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
syn = importlib.import_module("4dgaussians_amd.synthetic"); R = importlib.import_module("4dgaussians_amd.renderer")
Dm = importlib.import_module("4dgaussians_amd.scene.deformation")
Craw = importlib.import_module("4dgaussians_amd.diff_gaussian_rasterization")._C
dev = torch.device("cuda", 0); cfg = syn.CONFIGS["C2"]; _, cam0, raw = syn.make_workload("C2"); W, H = cfg["W"], cfg["H"]; P = raw["xyz"].shape[0]
torch.manual_seed(syn.seed_for("C2"))
pc = R.GaussianParams(raw, dev); deform = Dm.deform_network(bench.deform_args("C2")); deform.deformation_net.set_aabb([1.3]*3, [-1.3]*3); deform = deform.to(dev)
cam_np = syn.look_at_camera(W, H, azimuth=0.7); cam = R.Camera(cam_np, dev, time=float(np.random.default_rng(1234).uniform(0, 1)))
bg = torch.ones(3, device=dev)
with torch.no_grad():
    pc._scaling.add_(float(os.environ.get("OFFSET", "0.4982")))
    m3, sc, rot, op, sh = deform.forward_dynamic(pc._xyz, pc._scaling, pc._rotation, pc._opacity, pc._features, torch.full((P, 1), cam.time, device=dev), torch.zeros(P, 1, device=dev), None)
    e = torch.Tensor([])
    Rn, col, dep, radii, gb, bb, ib = Craw.rasterize_gaussians(bg, m3, e, torch.sigmoid(op), torch.exp(sc), torch.nn.functional.normalize(rot), 1.0, e, cam.world_view_transform, cam.full_proj_transform, cam_np.tanfovx, cam_np.tanfovy, H, W, sh, 3, cam.camera_center, False, False)
    v = Craw.state_view(P, Rn, W, H, gb, bb, ib)
    rg = v["ranges"].cpu().numpy().astype(np.int64); ln = rg[:, 1] - rg[:, 0]
    nc = v["n_contrib"].cpu().numpy().astype(np.int64)
    gy, gx = (H + 15) // 16, (W + 15) // 16
    pad = np.zeros((gy * 16, gx * 16), np.int64); pad[:H, :W] = nc
    tm = pad.reshape(gy, 16, gx, 16).max(axis=(1, 3)).reshape(-1)
    print("R", Rn, "tiles", ln.size, "list len mean %.0f p50 %d p90 %d p99 %d max %d" % (ln.mean(), *np.percentile(ln, [50, 90, 99]).astype(int), ln.max()))
    print("processed (tile max n_contrib) mean %.0f p50 %d p90 %d p99 %d max %d  sum %d" % (tm.mean(), *np.percentile(tm, [50, 90, 99]).astype(int), tm.max(), tm.sum()))
    print("radii>0:", int((radii > 0).sum()), " max radius", int(radii.max()), " mean pixel n_contrib %.1f" % nc.mean())
