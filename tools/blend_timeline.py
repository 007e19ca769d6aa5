"""Kernel experiment: per-workgroup timeline of blend_backward_kernel on the C2 bench scene.

Needs the instrumented library (tools/build_timeline_lib.sh -> 4dgaussians_amd/libgs4d_tl.so) and
GS4D_LIB_NAME=libgs4d_tl.so.  Prints how the kernel's span compares with the sum of the workgroups' own durations, the
number of workgroups in flight over time and the busiest / idlest compute units."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
Wl = importlib.import_module("4dgaussians_amd.workload")
R = importlib.import_module("4dgaussians_amd.renderer")
F = importlib.import_module("4dgaussians_amd.fused")
cabi = importlib.import_module("4dgaussians_amd._cabi")

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
which = sys.argv[2] if len(sys.argv) > 2 else "blend_bwd"          # or "binning": segment_depth_sort_kernel
dev = torch.device("cuda", 0)
scene = Wl.build_scene(cfg, dev, calibrate=True)
pc, deform, cam, bg, target = (scene[k] for k in ("pc", "deform", "cam", "bg", "target"))
W, H = scene["W"], scene["H"]
for _ in range(4):
    image, *_ = R.render(cam, pc, deform, bg)
    F.l1_loss(image, target).backward()
    for p in list(pc.parameters()) + list(deform.parameters()):
        p.grad = None
torch.cuda.synchronize()
L = cabi.lib()
gx, gy = (W + 15) // 16, (H + 15) // 16
T = gx * gy
if which == "binning":
    T = T + (T + 3) // 4
buf = np.zeros((65536, 4), np.uint64)
rd = getattr(L, "gs4d_debug_timeline_read_" + which)
rd.argtypes = [C.c_void_p, C.c_size_t]
rc = rd(buf.ctypes.data, buf.nbytes)
assert rc == 0, rc
b = buf[:min(T, 65536)]
if which == "blend_bwd":           # every block of the last launch (segment slots + tiles) has a recent start time
    b = buf[buf[:, 0] > buf[:, 0].max() - np.uint64(100000)]
    T = b.shape[0]
t0 = b[:, 0].astype(np.int64); t1 = b[:, 1].astype(np.int64)
hw = (b[:, 2] & np.uint64(0xffffffff)).astype(np.int64); xcc = (b[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
n = (b[:, 3] >> np.uint64(32)).astype(np.int64); visits = (b[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
start = t0.min(); span = (t1.max() - start) * 10e-3   # 100 MHz counter -> us
dur = (t1 - t0) * 10e-3
if which == "binning":
    ntl = gx * gy
    wg = np.arange(T) < ntl
    for name, m in (("workgroup form", wg & (dur > 0.3)), ("wave form", ~wg)):
        if m.any():
            print(f"{name}: {int(m.sum())} workgroups, duration mean {dur[m].mean():.2f} p50 {np.median(dur[m]):.2f} max {dur[m].max():.2f} us, start {s_us[m].min() if False else (t0[m].min() - start) * 10e-3:.1f}..{(t0[m].max() - start) * 10e-3:.1f}, last end {(t1[m].max() - start) * 10e-3:.1f} us")
    o = np.argsort(-dur)[:6]
    print("longest: " + "  ".join(f"blk {i} n {n[i]} {(t0[i]-start)*10e-3:.1f}->{(t1[i]-start)*10e-3:.1f}" for i in o))
print(f"tiles {T}  kernel span {span:.1f} us  sum of workgroup durations {dur.sum():.0f} us  mean {dur.mean():.2f} max {dur.max():.1f} us")
print("list length mean %.0f p50 %d p90 %d p99 %d max %d;  wave-0 visits mean %.0f p90 %d p99 %d max %d" % (
    n.mean(), *np.percentile(n, [50, 90, 99]).astype(int), n.max(), visits.mean(), *np.percentile(visits, [90, 99]).astype(int), visits.max()))
# workgroups in flight over time
edges = np.linspace(0, span, 21)
mid = 0.5 * (edges[1:] + edges[:-1])
s_us = (t0 - start) * 10e-3; e_us = (t1 - start) * 10e-3
print("in flight at t (us): " + "  ".join(f"{m:.0f}:{int(((s_us <= m) & (e_us > m)).sum())}" for m in mid))
print("started by t: " + "  ".join(f"{m:.0f}:{int((s_us <= m).sum())}" for m in mid))
# per compute unit: (xcc, se, cu) from HW_ID: cu_id bits 8..11, sh_id bit 12, se_id bits 13..15 (gfx9 layout)
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
uk, inv = np.unique(key, return_inverse=True)
busy = np.zeros(uk.size); last = np.zeros(uk.size); cnt = np.zeros(uk.size, np.int64); vis = np.zeros(uk.size, np.int64)
np.add.at(busy, inv, dur); np.maximum.at(last, inv, e_us); np.add.at(cnt, inv, 1); np.add.at(vis, inv, visits)
print(f"compute units seen {uk.size}; workgroups per CU min {cnt.min()} max {cnt.max()}; last end per CU: min {last.min():.1f} p50 {np.median(last):.1f} max {last.max():.1f} us")
print(f"visits per CU (wave 0): min {vis.min()} p50 {int(np.median(vis))} max {vis.max()}")
o = np.argsort(-dur)[:8]
print("longest workgroups: " + "  ".join(f"tile {i} n {n[i]} v {visits[i]} {s_us[i]:.0f}->{e_us[i]:.0f}" for i in o))
if os.environ.get("TL_CLOCK"):
    cyc = b[:, 2].astype(np.float64); long = dur > 20
    print(f"s_memtime ticks per us over workgroups longer than 20 us: median {np.median(cyc[long] / dur[long]):.1f} (min {(cyc[long] / dur[long]).min():.1f}, max {(cyc[long] / dur[long]).max():.1f})")
# time per visit as a function of when it ran
late = e_us > 0.6 * span
print(f"us per visit: all {dur.sum() / max(visits.sum(), 1):.3f}; workgroups ending after 60% of the span {dur[late].sum() / max(visits[late].sum(), 1):.3f} ({late.sum()} workgroups)")
