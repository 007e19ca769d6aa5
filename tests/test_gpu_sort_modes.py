"""The binning stage has three sort strategies (csrc/binning.hip): tile-first (default: two stable passes on the tile
bits, then one in-LDS depth sort per tile -- one wave per list of at most 1024 entries, one workgroup per longer list,
and a chunked global path for lists longer than the LDS capacity), the
six-pass LSD sort (GS4D_SORT=lsd) and direct placement (GS4D_SORT=direct: instances counted per tile, written straight
into their tile's list in arbitrary order, the per-tile sort orders equal depths by Gaussian index).  All code paths must produce the SAME sorted keys, point list, ranges,
n_contrib and image, bit for bit (the default path is checked against the oracle in test_gpu_rasterizer.py).  The
strategy is read once per process, so the other two run in child processes."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _child(env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sort_mode_child.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return [l for l in r.stdout.strip().splitlines() if l and l[0].isdigit()]


def test_sort_strategies_agree_bit_for_bit():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sort_mode_child
    here = sort_mode_child.digests()                       # this process: default strategy, lists sorted in LDS
    assert len(here) == 6
    longest = max(int(l.split()[1]) for l in here)
    assert longest > 512                                    # all four instantiations of the wave form are used
    assert _child({"GS4D_SORT": "lsd"}) == here
    assert _child({"GS4D_SORT_SEGMENT_CAP": "64"}) == here
    assert _child({"GS4D_SORT_SEGMENT_CAP": "0"}) == here   # every list through the global path
    assert _child({"GS4D_SORT_WAVE_CAP": "0"}) == here      # every list by a whole workgroup
    assert _child({"GS4D_SORT_WAVE_CAP": "100"}) == here    # another split between the two forms
    assert _child({"GS4D_SORT": "direct"}) == here
    assert _child({"GS4D_SORT": "direct", "GS4D_SORT_SEGMENT_CAP": "0"}) == here
    assert _child({"GS4D_SORT": "direct", "GS4D_SORT_WAVE_CAP": "0"}) == here
    assert _child({"GS4D_SORT": "direct", "GS4D_SORT_WAVE_CAP": "100", "GS4D_SORT_SEGMENT_CAP": "700"}) == here


def test_binning_follows_the_row_order_of_the_model():
    """Unforced (GS4D_SORT unset), the forward bins by direct placement when the counts of the PREVIOUS forward of the thread
    say it is cheaper (scan_block_sums_kernel: num_rendered + 7 x the device-scope adds placement needs <= 2.4 M; the adds
    are few when every 256 consecutive rows touch few tiles together), by the radix passes otherwise: i.i.d. rows stay on
    the passes, the same Gaussians in Morton order switch after one forward, and both give the same lists and image bit
    for bit."""
    if os.environ.get("GS4D_SORT"):
        pytest.skip("binning strategy forced by GS4D_SORT")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import numpy as np
    import sort_mode_child
    syn = importlib.import_module("4dgaussians_amd.synthetic")
    wl = importlib.import_module("4dgaussians_amd.workload")
    cabi = importlib.import_module("4dgaussians_amd._cabi")
    P, W, H = 30000, 1280, 960
    cam = syn.look_at_camera(W, H)
    raw = syn.make_gaussians(P, 11, 0.012)
    a = syn.activate(raw)
    d0, p0 = cabi.binning_stats()
    first = sort_mode_child.digest_scene(a, cam, P, W, H)
    d1, p1 = cabi.binning_stats()
    again = sort_mode_child.digest_scene(a, cam, P, W, H)
    d2, p2 = cabi.binning_stats()
    assert again == first
    assert (d2 - d1, p2 - p1) == (0, 1)                      # rows in no order: the passes, whatever ran before this test
    perm = wl.morton_permutation(raw["xyz"])
    m = {k: np.ascontiguousarray(v[perm]) for k, v in a.items()}
    by_passes = sort_mode_child.digest_scene(m, cam, P, W, H)      # decided by the i.i.d. forward before it
    d3, p3 = cabi.binning_stats()
    assert (d3 - d2, p3 - p2) == (0, 1)
    by_placement = sort_mode_child.digest_scene(m, cam, P, W, H)
    d4, p4 = cabi.binning_stats()
    assert (d4 - d3, p4 - p3) == (1, 0)
    assert by_placement == by_passes
    assert sort_mode_child.digest_scene(a, cam, P, W, H) == first          # still direct (decided by the Morton forward) ...
    d5, p5 = cabi.binning_stats()
    assert (d5 - d4, p5 - p4) == (1, 0)
    assert sort_mode_child.digest_scene(a, cam, P, W, H) == first          # ... and back on the passes
    assert cabi.binning_stats() == (d5, p5 + 1)
