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
  roofline      -- for the dominant kernel (blend_backward, fp32-VALU bound): SURVEY 8(d)'s flops per (pixel, Gaussian)
                   pair x pairs per launch / its average duration measured with HIP events on the launch stream in the
                   timed region, against the fp32 vector peak; the HBM view of the same launch under "hbm"
  render_fps    -- (train mode) BASELINE.json's second metric from the same run: render_for_image frames/s with the
                   blend_forward roofline
  cpu_baseline  -- the CPU oracle (scalar C rasterizer + PyTorch-CPU deformation) on one core for the
                   same workload, rank 0 / N == 1 only; host core count / CPU model and the all-core deformation time beside it
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
          "blend_forward", "l1_loss_forward", "l1_loss_backward", "blend_backward", "geom_backward",
          "activate_backward", "mlp_backward", "hexplane_backward", "weight_grads", "plane_reg_forward", "plane_reg_backward", "adamw")


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
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--P", type=int, default=None, help="override the Gaussian count (then the line is not the headline config)")
    ap.add_argument("--views", type=int, default=8,
                    help="distinct (camera, timestamp, target) views each rank cycles through, one per step (train.py:168-189 draws a random view per iteration)")
    ap.add_argument("--views-per-gpu", type=int, default=1,
                    help="views every rank renders per step (BASELINE configs C3-C5 put 2-8 on a rank): one deformation query for all of them "
                         "(renderer.render_batch), one rasterizer pass each; value counts views")
    ap.add_argument("--no-spatial-order", action="store_true",
                    help="keep the generator's i.i.d. row order instead of storing the Gaussians in the Morton order of their positions")
    ap.add_argument("--mode", default="train", choices=["train", "fps"])
    ap.add_argument("--autograd-step", action="store_true",
                    help="train step as render_with_l1(...) + loss.backward() instead of renderer.render_l1_step "
                         "(same library calls and gradients; the host then has autograd work between them)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimizer-pass", action="store_true", help="skip the separate optimizer-inclusive timing loop")
    ap.add_argument("--no-fps-pass", action="store_true", help="(train mode) skip the render-FPS loop that fills `render_fps`")
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
    Wl = importlib.import_module("4dgaussians_amd.workload")
    par = importlib.import_module("4dgaussians_amd.parallel")
    F = importlib.import_module("4dgaussians_amd.fused")
    cabi = importlib.import_module("4dgaussians_amd._cabi")
    L = cabi.lib()
    if dp and not rehearse and dist.get_world_size() != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={dist.get_world_size()}: "
                         "start it as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    if not dp and args.gpus != 1:
        raise SystemExit(f"bench.py --gpus {args.gpus} needs one process per GPU (torch.distributed.run); got a single process")

    # the scene every rank holds (identical seeds -> identical replicas) and this rank's view of it; the parity tests
    # build the same thing through the same function (tests/test_gpu_c2.py)
    scene = Wl.build_scene(args.config, dev, P=args.P, rank=rank, calibrate=False, spatial_order=not args.no_spatial_order)
    pc, deform, cam, cam_np, bg, target = (scene[k] for k in ("pc", "deform", "cam", "cam_np", "bg", "target"))
    P, W, H = scene["P"], scene["W"], scene["H"]
    # the views this rank cycles through (view 0 = the scene's own view: calibration, parity tests, CPU baseline)
    views = Wl.build_views(args.config, dev, rank=rank, n=max(1, args.views))
    V = len(views)
    step_no = [0]

    def next_view():
        v = views[step_no[0] % V]
        step_no[0] += 1
        return v

    if dp:  # replicas start identical
        par.GradBucket.exchange_in_world_of_one = rehearse
        for p in list(pc.parameters()) + list(deform.parameters()):
            dist.broadcast(par._memory_order_view(p.data), 0)      # a contiguous 1-D view, also of channels-last planes
    _dbg("parameters replicated")
    Craw = importlib.import_module("4dgaussians_amd.diff_gaussian_rasterization")._C

    def rasterize_untimed():
        return Wl.rasterize_untimed(pc, deform, cam, cam_np, bg)

    # ~8 tiles per Gaussian on the DEFORMED scene (workload.calibrate_log_scale_offset); with several
    # ranks the offset found on rank 0's view is applied everywhere
    log_s0_offset = Wl.calibrate_log_scale_offset(pc, deform, cam, cam_np, bg)
    if dp:
        off = torch.tensor([log_s0_offset], dtype=torch.float64, device=dev)
        dist.broadcast(off, 0)
        with torch.no_grad():
            pc._scaling.add_(float(off.item()) - log_s0_offset)
        log_s0_offset = float(off.item())
    _dbg(f"scene calibrated, offset {log_s0_offset:.4f}")
    raw = dict(scene["raw"], log_scales=pc._scaling.detach().cpu().numpy().copy())
    all_params = list(pc.parameters()) + list(deform.parameters())
    # ONE flat bucket for the Gaussian and the deformation gradients, exchanged in three pieces that overlap with the
    # backward pass: (1) the SH gradients are final when the rasterizer's backward has run (tensor hook), (2) position /
    # scale / rotation / plane / force_time_embed gradients are final after the deformation backward's data phase and
    # travel while its weight-gradient products run, (3) the Linear weights and the opacities go at the end.
    planes = [p for level in deform.deformation_net.grid.grids for p in level]
    ft = deform.deformation_net.force_time_embed
    data_params = [pc._xyz, pc._scaling, pc._rotation, *planes, ft.weight, ft.bias]
    comm, collectives, collectives_fallback = None, "torch.distributed", None
    if dp and dist.get_backend() == "nccl" and not os.environ.get("GS4D_BENCH_TORCH_COLLECTIVES"):
        try:                                    # collective: all ranks succeed or all fall back together
            comm, collectives = par.RcclComm(dev), "ncclAllReduce on a side stream (parallel.RcclComm)"
        except RuntimeError as e:
            # not silent: the line carries "collectives_fallback" with the reason, and GS4D_BENCH_STRICT_COLLECTIVES=1
            # turns it into a non-zero exit (the collectives are RCCL either way: torch's "nccl" backend)
            collectives_fallback = str(e)
            print(f"[bench rank {rank}] direct RCCL communicator unavailable, using torch.distributed collectives: {e}", file=sys.stderr)
            if os.environ.get("GS4D_BENCH_STRICT_COLLECTIVES"):
                dist.destroy_process_group()
                raise SystemExit(3)
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

    def binning_bytes_per_instance(n_rendered):
        import ctypes as C
        Rn = max(int(n_rendered), 1)
        L.gs4d_binning_bytes_forward_only.restype = C.c_size_t
        return {"train": round(L.gs4d_binning_bytes(C.c_int(Rn)) / Rn, 1),
                "forward_only": round(L.gs4d_binning_bytes_forward_only(C.c_int(Rn)) / Rn, 1)}

    VPG = max(1, args.views_per_gpu)
    FUSED_STEP = not args.autograd_step and not R.STAGED_ONLY

    def batch_step():
        vs = [next_view() for _ in range(VPG)]
        outs = R.render_batch([v.cam for v in vs], pc, deform, bg, targets=[v.target for v in vs])
        loss = outs[0][0]
        for o in outs[1:]:
            loss = loss + o[0]
        (loss / (VPG * world)).backward()                 # mean over the global batch of views
        if dp:
            bucket.all_reduce()
        for p in all_params:
            p.grad = None

    def train_step():
        if VPG > 1:
            return batch_step()
        v = next_view()
        # render + L1 in one library call per direction (renderer.render_with_l1 -> gs4d_render_forward / _backward)
        if FUSED_STEP:
            # forward + L1 + backward issued back to back, gradients straight into .grad (renderer.render_l1_step);
            # the SUM over ranks of (1 / world) * loss is the mean over the global batch
            R.render_l1_step(v.cam, pc, deform, bg, v.target, loss_scale=1.0 / world if dp else 1.0, shs_grad_hook=hook)
        else:
            loss, *_ = R.render_with_l1(v.cam, pc, deform, bg, v.target, shs_grad_hook=hook)
            if dp:
                loss = loss / world                           # SUM over ranks == mean over the global batch
            loss.backward()
        if dp:
            bucket.all_reduce()
        for p in all_params:
            p.grad = None

    def fps_step():
        R.render_for_image(next_view().cam, pc, deform, bg)

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
    stats0 = cabi.forward_stats()
    spec0 = cabi.speculation_stats()
    bin0 = cabi.binning_stats_full()
    step_no[0] = 0                          # the timed region renders views 0, 1, ..., steps - 1 (mod V)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
    t_issued = time.perf_counter() - t0
    sync()
    dt = time.perf_counter() - t0
    stats1 = cabi.forward_stats()
    spec1 = cabi.speculation_stats()
    bin1 = cabi.binning_stats_full()
    nviews_timed = args.steps * (VPG if args.mode == "train" else 1)
    view_weight = [sum(1 for i in range(nviews_timed) if i % V == j) / nviews_timed for j in range(V)]
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

    def hex_order_counts():
        oc = deform.deformation_net.__dict__.get("_hex_order_cache")
        return None if oc is None else {"ranked": oc.records, "reused": oc.reuses, "note": "whole run, not only the timed region"}

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

    # render FPS (BASELINE.json's second metric; definition render.py:58-77 without the to8b D2H copy) in the SAME
    # default run, so that the driver's line carries it: render_for_image = deformation forward + rasterizer forward
    # under no_grad, timed between synchronisations like the train loop, blend_forward bracketed by HIP events
    fps_dt, fps_frames, fps_blend_ms = None, 0, None
    if args.mode == "train" and not args.no_fps_pass:
        for _ in range(max(3, args.warmup)):
            fps_step()
        sync()
        L.gs4d_profile_reset()
        L.gs4d_profile_select(None if args.no_stage_events else b"blend_forward")
        L.gs4d_profile_enable(0 if args.no_stage_events else 1)
        fps_frames = max(100, 5 * args.steps)
        step_no[0] = 0
        t0 = time.perf_counter()
        for _ in range(fps_frames):
            fps_step()
        sync()
        fps_dt = time.perf_counter() - t0
        L.gs4d_profile_enable(0)
        fps_blend_ms = read_stages().get("blend_forward")
        if dp:
            t = torch.tensor([fps_dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            fps_dt = float(t.item())

    if rank == 0:
        # workload statistics for the roofline numerators: one extra untimed forward per view through the C-level API,
        # averaged with the weight each view had in the timed region
        def view_stats(v):
            with torch.no_grad():
                Rv, _, _, radii, gb, bb, ib = Wl.rasterize_untimed(pc, deform, v.cam, v.cam_np, bg)
                view = Craw.state_view(P, Rv, W, H, gb, bb, ib)
                nc = view["n_contrib"].to(torch.int64)
                gy, gx = (H + 15) // 16, (W + 15) // 16
                pad = torch.zeros(gy * 16, gx * 16, dtype=torch.int64, device=dev)
                pad[:H, :W] = nc
                tile_max = pad.view(gy, 16, gx, 16).amax(dim=(1, 3))
                return Rv, int(tile_max.sum().item()) * 256     # instances; (pixel, Gaussian) pairs the blends evaluate
        per_view = [view_stats(v) for v in views]
        Rn = sum(w * r for w, (r, _) in zip(view_weight, per_view))
        pairs_bwd = sum(w * p for w, (_, p) in zip(view_weight, per_view))
        fps_weight = [sum(1 for i in range(fps_frames) if i % V == j) / max(fps_frames, 1) for j in range(V)]
        Rn_fps = sum(w * r for w, (r, _) in zip(fps_weight, per_view))
        pairs_fps = sum(w * p for w, (_, p) in zip(fps_weight, per_view))

        def pmc_file(kind):
            """This round's committed rocprofv3 PMC summary (profiles/r03_pmc_<kind>.json, written by
            tools/profile_round.sh from separate counter passes over this same command), else last round's."""
            for r in ("r03", "r02"):
                path = os.path.join(ROOT, "profiles", f"{r}_pmc_{kind}.json")
                if os.path.exists(path):
                    return path
            return None

        def pmc_traffic(kernel_prefix):
            """(HBM bytes per launch, source file) of a kernel from the FETCH_SIZE / WRITE_SIZE passes, or (None, None)
            for another workload or when the file is absent."""
            path = pmc_file("traffic")
            try:
                d = json.load(open(path))
                for k, v in d["kernels"].items():
                    if k.startswith(kernel_prefix) and args.config == "C2" and args.P is None:
                        return (2.0 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024.0, os.path.relpath(path, ROOT)
            except (OSError, KeyError, ValueError, TypeError):
                pass
            return None, None

        def pmc_lds(kernel_prefix, stage_ms):
            """LDS traffic of a kernel from the SQ_INSTS_LDS pass: wave-level LDS instructions per launch x 64 lanes x the
            bytes per lane the kernel's source uses (tools/pmc_lds.py records both), over this run's launch duration,
            against the ~150 TB/s aggregate LDS peak (MI355X_MICROARCH.md)."""
            path = pmc_file("lds")
            try:
                d = json.load(open(path))
                for k, v in d["kernels"].items():
                    if k.startswith(kernel_prefix) and args.config == "C2" and args.P is None:
                        gb = v["lds_bytes_per_launch"] / (stage_ms * 1e-3) / 1e9
                        return {"lds_GBps": round(gb, 1), "lds_peak_GBps": 150000.0, "lds_frac": round(gb / 150000.0, 4),
                                "lds_insts_per_launch": v["SQ_INSTS_LDS"], "bytes_per_lane_assumed": v["bytes_per_lane"],
                                "bank_conflict_cycles_per_active_lds_cycle": v.get("conflict_ratio"),
                                "source": os.path.relpath(path, ROOT)}
            except (OSError, KeyError, ValueError, TypeError):
                pass
            return None

        def blend_roofline(kernel, stage_ms, alg_bytes, flop_per_pair, pairs):
            """Both blend kernels are bound by fp32 VALU issue (PMC: VALU busy ~76 %, MFMA pipe idle,
            HBM at a few per cent), so `bound` / `frac` are quoted against the fp32 vector peak with SURVEY 8(d)'s
            per-pair flop figure; the HBM view of the same launch is kept beside it."""
            sec = stage_ms * 1e-3
            tf = pairs * flop_per_pair / sec / 1e12
            traffic, src = pmc_traffic(kernel)
            return {"kernel": kernel.split("<")[0], "bound": "valu", "achieved": round(tf, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tf / FP32_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": src,
                    "avg_launch_ms": round(stage_ms, 4), "pairs": int(pairs), "flop_per_pair": flop_per_pair,
                    "hbm": {"algorithmic_bytes": alg_bytes, "achieved_GBps": round(alg_bytes / sec / 1e9, 2),
                            "peak_GBps": HBM_PEAK_GBS, "frac": round(alg_bytes / sec / 1e9 / HBM_PEAK_GBS, 5)},
                    "lds": pmc_lds(kernel, stage_ms),
                    "note": "pairs = sum over tiles of 256 x (list entries up to the tile's last contributor), mean over the "
                            "views of the timed region; flop_per_pair from SURVEY.md 8(d); peak = fp32 vector / f32-MFMA 157.3 TFLOP/s"}

        roofline = None
        if args.mode == "train" and "blend_backward" in stages:
            # 44 B/instance staged + 40 B/instance gradient rows + 24 B/pixel (dL/dcolor 12, dL/ddepth 4, final_T 4, n_contrib 4)
            roofline = blend_roofline("blend_backward_kernel", stages["blend_backward"], 84.0 * Rn + 24.0 * W * H, 60, pairs_bwd)
        elif "blend_forward" in stages:
            roofline = blend_roofline("blend_forward_kernel<false, false>", stages["blend_forward"], 44.0 * Rn + 24.0 * W * H, 25, pairs_bwd)

        # one roofline entry per stage that takes >= 5 % of the step (SURVEY.md 8(d)'s algorithmic figures x the units of
        # one launch / the stage's device time / the peak that bounds it)
        net = deform.deformation_net
        lin = [m for seq in (net.feature_out2, net.pos_deform, net.scales_deform, net.rotations_deform, net.shs_deform)
               for m in seq if isinstance(m, torch.nn.Linear)]
        mlp_flops = 2.0 * sum(m.in_features * m.out_features for m in lin)          # per Gaussian, forward
        kc = net.args.kplanes_config
        Lv, Cp = len(net.args.multires), int(kc["output_coordinate_dim"])
        gather_bytes = Lv * 6 * 4 * Cp * 4.0 + Lv * Cp * 4.0                         # corner rows gathered + features written
        L2_PEAK_GBS = 34500.0                                                        # MI355X_MICROARCH.md: aggregate L2
        PD = P * (VPG if args.mode == "train" else 1)                                # rows of one deformation launch
        spec = {  # stage -> (bound, units of work per launch, unit, peak, kernel name prefix in the PMC files, what)
            "hexplane_forward": ("l2-gather", PD * gather_bytes, "GB/s", L2_PEAK_GBS, "hexplane_forward_shared_kernel",
                                 "levels*6*4*C*4 B gathered + L*C*4 B written per Gaussian (SURVEY upper bound: served by L2 / Infinity Cache, the planes themselves are read from HBM once)"),
            "mlp_forward": ("mfma", PD * mlp_flops, "TFLOP/s", FP32_PEAK_TFLOPS, "heads_forward_kernel", "2*sum(in*out) FLOP per Gaussian, exact-f32 MFMA"),
            "preprocess": ("hbm", 311.0 * P, "GB/s", HBM_PEAK_GBS, "preprocess_kernel", "311 B per Gaussian"),
            "radix_sort": ("hbm", 144.0 * Rn, "GB/s", HBM_PEAK_GBS, "segment_depth_sort_kernel", "24 B per instance and pass x 6 LSD passes (the tile-first sort moves 72 B per instance)"),
            "blend_forward": ("valu", 25.0 * pairs_bwd, "TFLOP/s", FP32_PEAK_TFLOPS, "blend_forward_kernel<false, true>", "25 FLOP per (pixel, Gaussian) pair"),
            "blend_backward": ("valu", 60.0 * pairs_bwd, "TFLOP/s", FP32_PEAK_TFLOPS, "blend_backward_kernel", "60 FLOP per (pixel, Gaussian) pair"),
            "geom_backward": ("hbm", 567.0 * P + 48.0 * Rn, "GB/s", HBM_PEAK_GBS, "geom_backward_kernel", "567 B per Gaussian + 48 B per instance row"),
            "mlp_backward": ("mfma", PD * mlp_flops, "TFLOP/s", FP32_PEAK_TFLOPS, "heads_backward_kernel", "input gradients of every Linear: the forward's FLOP count"),
            "weight_grads": ("mfma", PD * mlp_flops, "TFLOP/s", FP32_PEAK_TFLOPS, "dw_partial_kernel", "weight gradients of every Linear: the forward's FLOP count"),
            "hexplane_backward": ("l2-gather", 2.0 * PD * gather_bytes, "GB/s", L2_PEAK_GBS, "hexplane_backward_point_shared_kernel", "twice the forward's gather figure (read the corners, add the gradients)"),
        }
        DEFORM_STAGES = ("hexplane_forward", "mlp_forward", "activate_forward", "activate_backward", "mlp_backward", "hexplane_backward", "weight_grads")
        launches = lambda st: 1 if (st in DEFORM_STAGES or args.mode != "train") else VPG
        step_ms = sum(v * launches(k) for k, v in stages.items()) or 1.0
        rooflines = []
        for st, (bound, work, unit, peak, kern, what) in spec.items():
            ms = stages.get(st)
            if not ms or ms * launches(st) < 0.05 * step_ms:
                continue
            ach = work / (ms * 1e-3) / (1e12 if unit == "TFLOP/s" else 1e9)
            traffic, src = pmc_traffic(kern)
            e = {"stage": st, "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
                 "stage_ms": round(ms, 4), "launches_per_step": launches(st), "share_of_step": round(ms * launches(st) / step_ms, 3),
                 "work_per_launch": work, "what": what,
                 "dominant_kernel": kern.split("<")[0], "dominant_kernel_traffic": traffic, "traffic_source": src}
            if st in ("radix_sort", "blend_forward", "blend_backward"):
                e["lds"] = pmc_lds(kern, ms)
            rooflines.append(e)

        render_fps = None
        if fps_dt is not None:
            render_fps = {"metric": "render FPS", "value": round(world * fps_frames / fps_dt, 2), "unit": "frames/s",
                          "ms_per_frame": round(fps_dt / fps_frames * 1e3, 4), "frames": fps_frames, "n_gpus": world,
                          "what": f"render_for_image (deformation forward + rasterizer forward, no grad) of the same scene at {W}x{H}, {V} views cycled",
                          # the forward-only instantiation (no segment checkpoints for a backward)
                          "roofline": blend_roofline("blend_forward_kernel<false, false>", fps_blend_ms, 44.0 * Rn_fps + 24.0 * W * H, 25, pairs_fps) if fps_blend_ms else None}

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, raw, cam_np, cam.time, deform, target.cpu().numpy())      # view 0

        unit = "view-iterations/s (fwd+bwd)" if args.mode == "train" else "frames/s"
        metric = "train iters/sec (fwd+bwd)" if args.mode == "train" else "render FPS"
        out = {
            "metric": metric, "value": round(world * nviews_timed / dt, 3), "unit": unit, "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {W}x{H}, P={P} Gaussians, SH degree 3, "
                                   f"{'mutant' if args.config != 'C5' else 'force_blend'}-shaped HexPlane deformation, "
                                   f"{VPG if args.mode == 'train' else 1} view(s) per GPU per step, {V} distinct (camera, timestamp, target) views cycled",
                       "global_batch_views": world * (VPG if args.mode == "train" else 1), "views_cycled": V,
                       "row_order": "generator (i.i.d.)" if args.no_spatial_order else "Morton order of the positions (GaussianModel(spatial_order=True) / reorder_spatially)",
                       "parallelism": f"view-dp{world}", "num_rendered": round(Rn, 1), "tiles_per_gaussian": round(Rn / P, 3),
                       "num_rendered_per_view": [r for r, _ in per_view],
                       "log_s0_offset": round(log_s0_offset, 4),
                       "grad_allreduce_bytes": bucket.nbytes, "collectives": collectives if dp else None,
                       "collectives_fallback": collectives_fallback,
                       "num_rendered_wait": dict(zip(("mailbox", "copy_sync_fallback"), cabi.mailbox_stats())),
                       # the timed region only: waits that fell back to copy + sync, forwards whose speculative binning
                       # chunk (largest recent R + 1/8) was too small, forwards whose binning + blend were issued before
                       # num_rendered had been read / of those, issued again; plane-gradient point orders ranked / reused
                       "timed_region": {"copy_sync_fallback": stats1[1] - stats0[1], "binning_second_allocations": stats1[2] - stats0[2],
                                        "speculative_forwards": {"issued": spec1[0] - spec0[0], "reissued": spec1[1] - spec0[1]},
                                        # how the forwards were binned (include/gs4d.h: gs4d_binning_stats) and the last
                                        # forward's two numbers that the choice rests on
                                        "binning": {"direct_placement": bin1[0] - bin0[0], "radix_passes": bin1[1] - bin0[1],
                                                    "last_adds": bin1[2], "last_rendered": bin1[3]},
                                        "hex_order": hex_order_counts()},
                       # memory of the per-view binning state (keys, values, sort buffers, histograms + the blend's segment
                       # checkpoints that only a training forward reserves), per rendered instance
                       "binning_bytes_per_instance": binning_bytes_per_instance(Rn),
                       "render_path": "staged" if R.STAGED_ONLY else
                                      "gs4d_render_forward / gs4d_render_backward (one call per direction)" +
                                      (", issued back to back by renderer.render_l1_step" if FUSED_STEP and VPG == 1 else
                                       ", through autograd")},
            "roofline": roofline, "rooflines": rooflines, "cpu_baseline": cpu, "render_fps": render_fps,
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
            v = next_view()
            cam, target = v.cam, v.target
            if FUSED_STEP:
                # render + L1 + backward in one go; the regularisers see only the (replicated) planes, so every rank adds
                # their gradient locally AFTER the exchange of the render-path gradients
                R.render_l1_step(cam, pc, deform, bg, target, loss_scale=1.0 / world if dp else 1.0, shs_grad_hook=hook)
                if dp:
                    bucket.all_reduce()
                reg.compute_regulation(deform, 0.01, 0.0001, 0.0, 0.0001, detail=False)["total_reg"].backward()
                opt.step()
                for p in all_params:
                    p.grad = None
                return
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

        # these iterations MOVE the parameters (random targets: the splats grow and num_rendered with them), so the loop is
        # kept short: it times the iteration at the bench scene, not at whatever 200 optimizer steps turn it into
        opt_steps, opt_warm = min(args.steps, 30), min(args.warmup, 5)
        for _ in range(opt_warm):
            opt_step()
        sync()
        t0 = time.perf_counter()
        for _ in range(opt_steps):
            opt_step()
        sync()
        dto = time.perf_counter() - t0
        if dp:
            t = torch.tensor([dto], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dto = float(t.item())
        with_opt = {"value": round(world * opt_steps / dto, 3), "unit": "view-iterations/s (fwd+bwd+regularisers+AdamW)",
                    "ms_per_step": round(dto / opt_steps * 1e3, 4), "steps": opt_steps,
                    "note": "same iteration + HexPlane grid regularisers (scene/gaussian_model.py:679-706) + one fused AdamW "
                            "step over all parameters (scene/gaussian_model.py:195-206); not the headline metric"}
    # the fork's complete fine-stage iteration (all ranks; last, it moves the parameters too): render() with the two
    # extra deformation queries and the k = 100 kNN regularisers (gaussian_renderer/renderer.py:173-203), weighted as
    # train.py:191-196, + grid regularisers (train.py:207-214) + AdamW
    full_stage = None
    if args.mode == "train" and not args.no_optimizer_pass:
        def full_step():
            v = next_view()
            cam, target = v.cam, v.target
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
        for _ in range(max(2, min(args.warmup, 5) // 2)):
            full_step()
        sync()
        t0 = time.perf_counter()
        nfull = min(max(3, args.steps), 30)
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, raw, cam_np, t_view, deform, target):
    """The CPU oracle (oracle/compose.py: PyTorch-CPU deformation restatement + scalar C rasterizer) on ONE core: one
    full iteration of the same workload (forward only for --mode fps), plus -- SURVEY.md 8(d) -- the deformation
    part alone on all the cores this process may use.  Reported, never used as the product."""
    from oracle import compose
    train = args.mode == "train"
    cfg = compose.config_from_args(deform.deformation_net.args)
    sd = deform.state_dict()
    P = raw["xyz"].shape[0]
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads_all = max(1, min(affinity, int(os.environ.get("OMP_NUM_THREADS", affinity))))
    torch.set_num_threads(1)
    one = compose.step(cfg, sd, raw, cam_np, t_view, target=target, train=train)
    sec = one["seconds"]
    # the deformation field alone (forward, and backward from random upstream gradients) on one and on all cores
    from oracle import deformation_ref as DR

    def deformation_only(threads):
        torch.set_num_threads(threads)
        sdg = {k: v.detach().cpu().contiguous().clone().requires_grad_(train and v.dtype == torch.float32 and "poc" not in k and "aabb" not in k)
               for k, v in sd.items()}
        x = [torch.from_numpy(raw[k]).clone().requires_grad_(train) for k in ("xyz", "log_scales", "rot", "opacity_logit", "sh")]
        t0 = time.perf_counter()
        with torch.set_grad_enabled(train):
            outs = DR.forward_dynamic(sdg, cfg, *x, torch.full((P, 1), float(t_view)), torch.zeros(P, 1))
            if train:
                torch.autograd.backward([outs[0], outs[1], outs[2], outs[4]],
                                        [torch.ones_like(outs[0]), torch.ones_like(outs[1]), torch.ones_like(outs[2]), torch.ones_like(outs[4])])
        return time.perf_counter() - t0

    d1 = deformation_only(1)
    dall = deformation_only(threads_all)
    torch.set_num_threads(threads_all)
    what = "fwd+bwd iteration" if train else "frame"
    return {"value": round(1.0 / sec, 5), "unit": "view-iterations/s (fwd+bwd)" if train else "frames/s", "cores": 1,
            "kind": "port", "seconds": round(sec, 2),
            "sample": f"one full {what} of the same workload (P={P}, {cam_np.image_width}x{cam_np.image_height}) on 1 thread: "
                      "oracle/compose.py = oracle/deformation_ref.py + oracle/rasterizer_ref.c",
            "host": {"nproc": os.cpu_count(), "usable_cores": affinity, "cpu_model": cpu_model()},
            "deformation_only": {"what": f"deformation field {'fwd+bwd' if train else 'fwd'} of the same P points (PyTorch-CPU restatement)",
                                 "seconds_1_thread": round(d1, 3), "seconds_all_threads": round(dall, 3), "threads": threads_all}}


if __name__ == "__main__":
    main()
