"""The binning stage has two sort strategies (csrc/binning.hip): tile-first (default: two stable passes on the tile
bits, then one in-LDS depth sort per tile -- one wave per list of at most 1024 entries, one workgroup per longer list,
and a chunked global path for lists longer than the LDS capacity) and the
six-pass LSD sort (GS4D_SORT=lsd).  All code paths must produce the SAME sorted keys, point list, ranges,
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
    assert len(here) == 3
    longest = max(int(l.split()[1]) for l in here)
    assert longest > 512                                    # all four instantiations of the wave form are used
    assert _child({"GS4D_SORT": "lsd"}) == here
    assert _child({"GS4D_SORT_SEGMENT_CAP": "64"}) == here
    assert _child({"GS4D_SORT_SEGMENT_CAP": "0"}) == here   # every list through the global path
    assert _child({"GS4D_SORT_WAVE_CAP": "0"}) == here      # every list by a whole workgroup
    assert _child({"GS4D_SORT_WAVE_CAP": "100"}) == here    # another split between the two forms
