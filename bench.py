"""bench.py -- headline benchmark of the MI355X 4DGS render path.

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): fine-stage train iterations/s, one iteration = forward + backward of ONE
view: HexPlane deformation query -> activations -> rasterizer forward -> L1 loss -> rasterizer
backward -> deformation backward (+ the gradient all-reduce when N > 1).  Optimizer / densify are
outside the metric.  Workload: BASELINE.json configs[1] ("C2"): 800x800,
100k Gaussians, SH degree 3, mutant-shaped deformation network, synthetic seeded data.
Scaling is weak: every rank renders its own view of the same scene each step (view-parallel DP).

`--mode fps` times render_for_image (deformation forward + rasterizer forward, no grad) instead.

One JSON line is printed by rank 0.  Besides the contract fields it carries
  roofline      -- for the dominant kernel (blend_backward): algorithmic bytes per launch / its
                   average duration measured with HIP events on the launch stream in the timed region
  cpu_baseline  -- the CPU oracle (scalar C rasterizer + PyTorch-CPU deformation) on one core for the
                   same workload, rank 0 / N == 1 only
  stages_ms     -- average device milliseconds of every kernel stage (same HIP-event mechanism)
  with_optimizer -- (train mode) the same iteration plus the HexPlane grid regularisers in the loss and one fused
                   AdamW step over all parameters ("report optimizer-inclusive separately"),
                   timed in its own loop AFTER everything else (it changes the parameters)
"""
from __future__ import annotations

import argparse
import gc
import ctypes as C
import importlib
import json
import os
import sys
import time
import types

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_PEAK_TFLOPS = 157.3   # vector / f32-MFMA peak

STAGES = ("hexplane_forward", "mlp_forward", "activate_forward", "preprocess", "scan", "emit_keys", "radix_sort", "tile_ranges",
          "blend_forward", "l1_loss_forward", "l1_loss_backward", "inverse_permutation", "blend_backward", "geom_backward",
          "activate_backward", "mlp_backward", "hexplane_backward", "weight_grads", "plane_reg_forward", "plane_reg_backward", "adamw")


def deform_args(name: str):
    """Shape-matched deformation configs: mutant-shaped for C1-C4, force_blend-shaped for C5."""
    a = types.SimpleNamespace(
        net_width=64, defor_depth=0, posebase_pe=10, scale_rotation_pe=2, opacity_pe=2, grid_pe=0, force_pe=2,
        time_pe=2, bounds=1.6, multires=[1, 2], no_dx=False, no_ds=False, no_dr=False, no_do=True, no_dshs=False,
        empty_voxel=False, static_mlp=False, apply_rotation=True, use_time=True, use_force=False,
        blend_time_force=True, prev_frames=0, no_grid=False,
        kplanes_config={"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32,
                        "resolution": [64, 64, 64, 75]})
    if name in ("C3", "C4"):
        a.kplanes_config["resolution"] = [64, 64, 64, 150]
    if name == "C5":
        a.kplanes_config.update(output_coordinate_dim=128, resolution=[64, 64, 64, 256])
        a.defor_depth, a.force_pe, a.time_pe, a.use_force = 5, 4, 4, True
    return a


_T0 = time.perf_counter()


def _dbg(msg):
    if os.environ.get("GS4D_BENCH_DEBUG"):
        print(f"[bench rank {os.environ.get('RANK', '0')} +{time.perf_counter() - _T0:7.2f}s] {msg}", file=sys.stderr, flush=True)


def main():
    # stdout carries the ONE JSON line and nothing else: whatever a library prints there (RCCL's version banner at
    # communicator set-up) is sent to stderr
    result_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--P", type=int, default=None, help="override the Gaussian count (then the line is not the headline config)")
    ap.add_argument("--mode", default="train", choices=["train", "fps"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimizer-pass", action="store_true", help="skip the separate optimizer-inclusive timing loop")
    ap.add_argument("--no-stage-events", action="store_true",
                    help="do not record per-stage HIP events in the timed region (then no roofline / stages_ms)")
    args = ap.parse_args()
    if os.environ.get("GS4D_BENCH_DEBUG", "").isdigit() and int(os.environ["GS4D_BENCH_DEBUG"]) > 1:
        import faulthandler                       # GS4D_BENCH_DEBUG=<seconds>: dump every thread's stack if still running then
        faulthandler.dump_traceback_later(int(os.environ["GS4D_BENCH_DEBUG"]), exit=False)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GS4D_BENCH_REHEARSE_DP=1 with one rank: run the multi-rank code path (hooks, bucket, RCCL all-reduces) in a
    # process group of ONE rank -- the rehearsal a one-GPU box allows; the line then says "rehearsal" and is no result
    rehearse = world == 1 and bool(os.environ.get("GS4D_BENCH_REHEARSE_DP"))
    dp = world > 1 or rehearse
    if dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        # RCCL ("nccl") on a multi-GPU node; GS4D_DIST_BACKEND=gloo routes the collectives through host memory
        dist.init_process_group(os.environ.get("GS4D_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    _dbg("process group ready")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU product path)")
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    syn = importlib.import_module("4dgaussians_amd.synthetic")
    R = importlib.import_module("4dgaussians_amd.renderer")
    Dm = importlib.import_module("4dgaussians_amd.scene.deformation")
    par = importlib.import_module("4dgaussians_amd.parallel")
    F = importlib.import_module("4dgaussians_amd.fused")
    cabi = importlib.import_module("4dgaussians_amd._cabi")
    L = cabi.lib()

    cfg = syn.CONFIGS[args.config]
    _, cam0, raw = syn.make_workload(args.config, P=args.P)
    P = raw["xyz"].shape[0]
    W, H = cfg["W"], cfg["H"]
    torch.manual_seed(syn.seed_for(args.config))
    pc = R.GaussianParams(raw, dev)
    deform = Dm.deform_network(deform_args(args.config))
    deform.deformation_net.set_aabb([1.3, 1.3, 1.3], [-1.3, -1.3, -1.3])
    deform = deform.to(dev)
    if dp:  # replicas start identical
        par.GradBucket.exchange_in_world_of_one = rehearse
        for p in list(pc.parameters()) + list(deform.parameters()):
            dist.broadcast(par._memory_order_view(p.data), 0)      # a contiguous 1-D view, also of channels-last planes
    _dbg("parameters replicated")
    # each rank looks at the scene from its own azimuth / timestamp (view-parallel)
    rng = np.random.default_rng(1234 + rank)
    cam_np = syn.look_at_camera(W, H, azimuth=0.7 + 0.6 * rank)
    cam = R.Camera(cam_np, dev, time=float(rng.uniform(0, 1)), force=0.0)
    bg = torch.ones(3, device=dev)
    target = torch.rand(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(99 + rank))
    Craw = importlib.import_module("4dgaussians_amd.diff_gaussian_rasterization")._C

    def rasterize_untimed():
        """One untimed forward through the C-level API -> (num_rendered, state buffers): workload statistics."""
        with torch.no_grad():
            m3, sc, rot, op, sh = deform.forward_dynamic(pc._xyz, pc._scaling, pc._rotation, pc._opacity, pc._features,
                                                         torch.full((P, 1), cam.time, device=dev),
                                                         torch.zeros(P, 1, device=dev), None)
            e = torch.Tensor([])
            return Craw.rasterize_gaussians(
                bg, m3, e, torch.sigmoid(op), torch.exp(sc), torch.nn.functional.normalize(rot), 1.0, e,
                cam.world_view_transform, cam.full_proj_transform, cam_np.tanfovx, cam_np.tanfovy, H, W, sh, 3,
                cam.camera_center, False, False)

    # s0 is chosen per config so that the RENDERED scene has ~8 tiles per Gaussian.  The
    # random-init deformation heads shift the log-scales, so s0 is calibrated on the deformed scene: a uniform
    # log-scale offset found by a few fixed-point steps on rank 0's view (tiles ~ scale^2), applied on every rank.
    log_s0_offset = 0.0
    for _ in range(8):
        ratio = rasterize_untimed()[0] / P
        if abs(ratio - 8.0) < 0.2 or ratio == 0:
            break
        d = 0.5 * float(np.log(8.0 / ratio))
        log_s0_offset += d
        with torch.no_grad():
            pc._scaling.add_(d)
    if dp:
        off = torch.tensor([log_s0_offset], dtype=torch.float64, device=dev)
        dist.broadcast(off, 0)
        with torch.no_grad():
            pc._scaling.add_(float(off.item()) - log_s0_offset)
        log_s0_offset = float(off.item())
    _dbg(f"scene calibrated, offset {log_s0_offset:.4f}")
    raw = dict(raw, log_scales=(raw["log_scales"] + np.float32(log_s0_offset)).astype(np.float32))
    all_params = list(pc.parameters()) + list(deform.parameters())
    # ONE flat bucket for the Gaussian and the deformation gradients, exchanged in three pieces that overlap with the
    # backward pass: (1) the SH gradients are final when the rasterizer's backward has run (tensor hook), (2) position /
    # scale / rotation / plane / force_time_embed gradients are final after the deformation backward's data phase and
    # travel while its weight-gradient products run, (3) the Linear weights and the opacities go at the end.
    planes = [p for level in deform.deformation_net.grid.grids for p in level]
    ft = deform.deformation_net.force_time_embed
    data_params = [pc._xyz, pc._scaling, pc._rotation, *planes, ft.weight, ft.bias]
    comm, collectives = None, "torch.distributed"
    if dp and dist.get_backend() == "nccl" and not os.environ.get("GS4D_BENCH_TORCH_COLLECTIVES"):
        try:                                    # collective: all ranks succeed or all fall back together
            comm, collectives = par.RcclComm(dev), "ncclAllReduce on a side stream (parallel.RcclComm)"
        except RuntimeError as e:
            print(f"[bench rank {rank}] direct RCCL communicator unavailable, using torch.distributed collectives: {e}", file=sys.stderr)
    _dbg(f"gradient exchange through {collectives}")
    bucket = par.GradBucket(all_params, early=[pc._features, *data_params], comm=comm)
    dp_hooks = os.environ.get("GS4D_BENCH_DP_HOOKS", "all")       # experiments: "all" | "shs" | "none"

    def shs_hook(g):
        bucket.early_reduce(pc._features, g)

    def data_hook(d_xyz, d_scaling, d_rotation, plane_grads, g_ftw, g_ftb):
        bucket.early_reduce_many(data_params, [d_xyz, d_scaling, d_rotation, *plane_grads, g_ftw, g_ftb])

    hook = shs_hook if dp and dp_hooks in ("all", "shs") else None
    if dp and dp_hooks == "all":
        deform.deformation_net.data_grad_hook = data_hook

    def train_step():
        image, *_ = R.render(cam, pc, deform, bg, shs_grad_hook=hook)
        loss = F.l1_loss(image, target)
        if dp:
            loss = loss / world                           # SUM over ranks == mean over the global batch
        loss.backward()
        if dp:
            bucket.all_reduce()
        for p in all_params:
            p.grad = None

    def fps_step():
        R.render_for_image(cam, pc, deform, bg)

    step = train_step if args.mode == "train" else fps_step

    def sync():
        torch.cuda.synchronize()
        if dp:
            dist.barrier()
            torch.cuda.synchronize()

    # Every recorded HIP event is a barrier packet that costs device idle time (the full 18-stage set cost
    # ~1.3 ms per step), so the timed region brackets ONLY the dominant kernel's stage; the other stages are
    # timed in a separate, untimed pass below and reported as diagnostics.
    dominant = "blend_backward" if args.mode == "train" else "blend_forward"
    # the set-up above leaves ~1e5 long-lived Python objects; freezing them keeps the cyclic collector's
    # passes over the young generations cheap (a full collection in the timed region cost ~36 ms once)
    gc.collect()
    gc.freeze()
    for i in range(args.warmup):
        step()
        _dbg(f"warmup step {i} issued")
    sync()
    _dbg("warmup done")
    L.gs4d_profile_reset()
    L.gs4d_profile_select(None if args.no_stage_events else dominant.encode())
    L.gs4d_profile_enable(0 if args.no_stage_events else 1)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
    t_issued = time.perf_counter() - t0
    sync()
    dt = time.perf_counter() - t0
    _dbg(f"host had issued the last step after {t_issued / args.steps * 1e3:.3f} ms per step")
    _dbg(f"timed region done: {dt / args.steps * 1e3:.3f} ms per step")
    L.gs4d_profile_enable(0)
    if dp:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    def read_stages():
        out = {}
        for s in STAGES:
            ms, calls = C.c_double(0), C.c_long(0)
            L.gs4d_profile_read(s.encode(), C.byref(ms), C.byref(calls))
            if calls.value:
                out[s] = ms.value / calls.value
        return out

    timed = read_stages()                   # the dominant stage, HIP events inside the timed region
    stages = dict(timed)
    if not args.no_stage_events:            # every stage, untimed diagnostic pass
        L.gs4d_profile_reset()
        L.gs4d_profile_select(None)
        L.gs4d_profile_enable(1)
        for _ in range(min(args.steps, 10)):
            step()
        sync()
        L.gs4d_profile_enable(0)
        stages = read_stages()
        stages.update(timed)

    if rank == 0:
        # workload statistics for the roofline numerators (one extra untimed forward through the C-level API)
        with torch.no_grad():
            Rn, _, _, radii, gb, bb, ib = rasterize_untimed()
            view = Craw.state_view(P, Rn, W, H, gb, bb, ib)
            nc = view["n_contrib"].to(torch.int64)
            gy, gx = (H + 15) // 16, (W + 15) // 16
            pad = torch.zeros(gy * 16, gx * 16, dtype=torch.int64, device=dev)
            pad[:H, :W] = nc
            tile_max = pad.view(gy, 16, gx, 16).amax(dim=(1, 3))
            pairs_bwd = int(tile_max.sum().item()) * 256        # (pixel, Gaussian) pairs the backward evaluates
        def pmc_traffic(kernel_prefix):
            """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/), or None."""
            try:
                d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic_v11.json")))
                for k, v in d["kernels"].items():
                    if k.startswith(kernel_prefix) and args.config == "C2" and args.P is None:
                        return (2.0 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024.0
            except (OSError, KeyError, ValueError):
                pass
            return None

        roofline = None
        if args.mode == "train" and "blend_backward" in stages:
            # 44 B/instance staged + 40 B/instance gradient rows + 24 B/pixel (dL/dcolor 12, dL/ddepth 4, final_T 4, n_contrib 4)
            alg_bytes = 84.0 * Rn + 24.0 * W * H
            sec = stages["blend_backward"] * 1e-3
            achieved = alg_bytes / sec / 1e9
            roofline = {"kernel": "blend_backward_kernel", "bound": "hbm", "achieved": round(achieved, 2),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic("blend_backward_kernel"),
                        "traffic_note": "bytes per launch = 2*FETCH_SIZE + WRITE_SIZE from profiles/r01_pmc_traffic_v11.json (separate rocprofv3 --pmc passes)",
                        "avg_launch_ms": round(stages["blend_backward"], 4), "algorithmic_bytes": alg_bytes,
                        "note": "kernel is fp32-VALU/LDS bound, not HBM bound: see valu_* fields",
                        "valu_pairs": pairs_bwd, "valu_flop_per_pair": 60,
                        "valu_achieved_tflops": round(pairs_bwd * 60 / sec / 1e12, 3), "valu_peak_tflops": FP32_PEAK_TFLOPS,
                        "valu_frac": round(pairs_bwd * 60 / sec / 1e12 / FP32_PEAK_TFLOPS, 4)}
        elif "blend_forward" in stages:
            alg_bytes = 44.0 * Rn + 24.0 * W * H
            sec = stages["blend_forward"] * 1e-3
            achieved = alg_bytes / sec / 1e9
            roofline = {"kernel": "blend_forward_kernel", "bound": "hbm", "achieved": round(achieved, 2),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic("blend_forward_kernel"),
                        "avg_launch_ms": round(stages["blend_forward"], 4), "algorithmic_bytes": alg_bytes}

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, syn, raw, cam_np, cam.time, deform, target.cpu().numpy())

        unit = "view-iterations/s (fwd+bwd)" if args.mode == "train" else "frames/s"
        metric = "train iters/sec (fwd+bwd)" if args.mode == "train" else "render FPS"
        out = {
            "metric": metric, "value": round(world * args.steps / dt, 3), "unit": unit, "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {W}x{H}, P={P} Gaussians, SH degree 3, "
                                   f"{'mutant' if args.config != 'C5' else 'force_blend'}-shaped HexPlane deformation, "
                                   f"1 view per GPU per step", "global_batch_views": world,
                       "parallelism": f"view-dp{world}", "num_rendered": Rn, "tiles_per_gaussian": round(Rn / P, 3),
                       "log_s0_offset": round(log_s0_offset, 4),
                       "grad_allreduce_bytes": bucket.nbytes, "collectives": collectives if dp else None},
            "roofline": roofline, "cpu_baseline": cpu,
            "stages_ms": {k: round(v, 4) for k, v in stages.items()},
            "stages_note": f"{dominant} timed with HIP events inside the timed region; the other stages in a separate untimed pass",
        }
    # optimizer-inclusive iteration (all ranks; after every other measurement because it moves the parameters)
    with_opt = None
    if args.mode == "train" and not args.no_optimizer_pass:
        opt_mod = importlib.import_module("4dgaussians_amd.optim")
        reg = importlib.import_module("4dgaussians_amd.scene.regulation")
        # learning rates / weights: arguments/__init__.py OptimizationParams defaults of the reference
        opt = opt_mod.FusedAdamW([
            {"params": [pc._xyz], "lr": 0.00016, "name": "xyz"},
            {"params": list(deform.get_mlp_parameters()), "lr": 0.00016, "name": "deformation"},
            {"params": list(deform.get_grid_parameters()), "lr": 0.0016, "name": "grid"},
            {"params": [pc._features], "lr": 0.0025, "name": "f_dc"},
            {"params": [pc._opacity], "lr": 0.05, "name": "opacity"},
            {"params": [pc._scaling], "lr": 0.005, "name": "scaling"},
            {"params": [pc._rotation], "lr": 0.001, "name": "rotation"}], lr=0.0001, eps=1e-8)

        def opt_step():
            image, *_ = R.render(cam, pc, deform, bg, shs_grad_hook=hook)
            regul = reg.compute_regulation(deform, 0.01, 0.0001, 0.0, 0.0001, detail=False)["total_reg"]
            if not dp:
                (F.l1_loss(image, target) + regul).backward()
            else:
                # the regularisers see only the (replicated) planes: every rank adds their gradient locally AFTER the
                # exchange of the render-path gradients ("view-independent"), no communication
                (F.l1_loss(image, target) / world).backward()
                bucket.all_reduce()
                regul.backward()
            opt.step()
            for p in all_params:
                p.grad = None

        for _ in range(args.warmup):
            opt_step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            opt_step()
        sync()
        dto = time.perf_counter() - t0
        if dp:
            t = torch.tensor([dto], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dto = float(t.item())
        with_opt = {"value": round(world * args.steps / dto, 3), "unit": "view-iterations/s (fwd+bwd+regularisers+AdamW)",
                    "ms_per_step": round(dto / args.steps * 1e3, 4),
                    "note": "same iteration + HexPlane grid regularisers (scene/gaussian_model.py:679-706) + one fused AdamW "
                            "step over all parameters (scene/gaussian_model.py:195-206); not the headline metric"}
    # the fork's complete fine-stage iteration (all ranks; last, it moves the parameters too): render() with the two
    # extra deformation queries and the k = 100 kNN regularisers (gaussian_renderer/renderer.py:173-203), weighted as
    # train.py:191-196, + grid regularisers (train.py:207-214) + AdamW
    full_stage = None
    if args.mode == "train" and not args.no_optimizer_pass:
        def full_step():
            image, _, _, _, _, rd, _ = R.render(cam, pc, deform, bg, shs_grad_hook=hook, regularise=True)
            regul = reg.compute_regulation(deform, 0.01, 0.0001, 0.0, 0.0001, detail=False)["total_reg"]
            knn_terms = (0.000001 * rd["momentum"] + rd["knn_xyz"] + 0.1 * rd["knn_long"] + rd["knn_rot"]
                         + 0.001 * rd["shs"] + 0.001 * rd["scales"])
            if not dp:
                (F.l1_loss(image, target) + regul + knn_terms).backward()
            else:
                # the data-phase hook must not run here: positions / planes receive more gradient from the two extra
                # queries after the centre query's backward, so only the SH hook stays early
                ((F.l1_loss(image, target) + knn_terms) / world).backward()
                bucket.all_reduce()
                regul.backward()
            opt.step()
            for p in all_params:
                p.grad = None

        saved_hook = deform.deformation_net.data_grad_hook
        deform.deformation_net.data_grad_hook = None
        hook_saved, hook = hook, None          # the deformed SHs of the centre query are not the only SH gradient path here
        for _ in range(max(2, args.warmup // 2)):
            full_step()
        sync()
        t0 = time.perf_counter()
        nfull = max(3, args.steps // 3)
        for _ in range(nfull):
            full_step()
        sync()
        dtf = time.perf_counter() - t0
        hook = hook_saved
        deform.deformation_net.data_grad_hook = saved_hook
        if dp:
            t = torch.tensor([dtf], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtf = float(t.item())
        full_stage = {"value": round(world * nfull / dtf, 3), "unit": "view-iterations/s (3 deformation queries + kNN regularisers + fwd+bwd + grid regularisers + AdamW)",
                      "ms_per_step": round(dtf / nfull * 1e3, 4), "steps": nfull,
                      "note": "the fork's whole fine-stage iteration per view (gaussian_renderer/renderer.py:173-203, train.py:191-220,328); not the headline metric"}
    if rank == 0:
        out["with_optimizer"] = with_opt
        out["full_fine_stage"] = full_stage
        if rehearse:
            out["rehearsal"] = "multi-rank code path in a process group of one rank: not a result"
        print(json.dumps(out), file=result_out, flush=True)
    if dp:
        dist.barrier()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


def cpu_baseline(args, syn, raw, cam_np, t_view, deform, target):
    """The CPU oracle on ONE core: PyTorch-CPU deformation restatement + scalar C rasterizer, one full
    iteration of the same workload (forward only for --mode fps).  Reported, never used as the product."""
    from oracle import deformation_ref as DR
    from oracle import rasterizer_oracle as RO
    torch.set_num_threads(1)
    a = deform.deformation_net.args
    kc = a.kplanes_config
    cfg = DR.DeformConfig(net_width=a.net_width, defor_depth=a.defor_depth, posebase_pe=a.posebase_pe,
                          scale_rotation_pe=a.scale_rotation_pe, force_pe=a.force_pe, time_pe=a.time_pe,
                          plane_channels=kc["output_coordinate_dim"], resolution=list(kc["resolution"]),
                          multires=list(a.multires), use_time=a.use_time, use_force=a.use_force)
    train = args.mode == "train"
    sd = {k: v.detach().cpu().contiguous().requires_grad_(train and v.dtype == torch.float32 and "poc" not in k and "aabb" not in k)
          for k, v in deform.state_dict().items()}
    P = raw["xyz"].shape[0]
    x = {k: torch.from_numpy(raw[k]).requires_grad_(train) for k in ("xyz", "log_scales", "rot", "opacity_logit", "sh")}
    t0 = time.perf_counter()
    with torch.set_grad_enabled(train):
        m3, sc, rot, op, sh = DR.forward_dynamic(sd, cfg, x["xyz"], x["log_scales"], x["rot"], x["opacity_logit"], x["sh"],
                                                 torch.full((P, 1), float(t_view)), torch.zeros(P, 1))
        sc_a, rot_a, op_a = torch.exp(sc), torch.nn.functional.normalize(rot), torch.sigmoid(op)
    f = RO.forward(np.ones(3, np.float32), m3.detach().numpy(), None, op_a.detach().numpy(), sc_a.detach().numpy(),
                   rot_a.detach().numpy(), 1.0, None, cam_np.viewmatrix, cam_np.projmatrix, cam_np.tanfovx, cam_np.tanfovy,
                   cam_np.image_height, cam_np.image_width, sh.detach().numpy(), 3, cam_np.campos)
    if train:
        gc = (np.sign(f["color"] - target) / target.size).astype(np.float32)
        g = RO.backward(f, gc, np.zeros((1, cam_np.image_height, cam_np.image_width), np.float32))
        tg = lambda k, shape: torch.from_numpy(g[k].reshape(shape))
        torch.autograd.backward([m3, sc_a, rot_a, op_a, sh],
                                [tg("dL_dmeans3D", m3.shape), tg("dL_dscales", sc_a.shape), tg("dL_drotations", rot_a.shape),
                                 tg("dL_dopacity", op_a.shape), tg("dL_dsh", sh.shape)])
    sec = time.perf_counter() - t0
    return {"value": round(1.0 / sec, 5), "unit": "view-iterations/s (fwd+bwd)" if train else "frames/s", "cores": 1,
            "kind": "port", "seconds": round(sec, 2),
            "sample": f"one full {'fwd+bwd iteration' if train else 'frame'} of the same workload (P={P}, "
                      f"{cam_np.image_width}x{cam_np.image_height}) on 1 thread: oracle/deformation_ref.py + oracle/rasterizer_ref.c"}


if __name__ == "__main__":
    main()
