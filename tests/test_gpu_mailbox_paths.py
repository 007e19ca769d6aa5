"""the num_rendered hand-over of gs4d_rasterize_forward has three paths -- the pinned-mailbox poll, the
poll timing out into copy + synchronise (GS4D_MAILBOX_SPIN_US=0 makes every call time out), and no mailbox at all
(GS4D_NO_MAILBOX=1) -- and only the first runs in an ordinary test.  Each mode renders the same scenes in a process of
its own; outputs (images, radii, sorted lists, ranges, n_contrib, final_T) must be identical byte for byte, num_rendered
equal, a prefilter violation reported on every path, and the statistics must show which path ran."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def child(**env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mailbox_child.py")], env=dict(os.environ, **env),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


def test_every_num_rendered_path_gives_the_same_render():
    assert "GS4D_NO_MAILBOX" not in os.environ and "GS4D_MAILBOX_SPIN_US" not in os.environ
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mailbox_child
    normal = mailbox_child.run()
    timeout = child(GS4D_MAILBOX_SPIN_US="0")
    nobox = child(GS4D_NO_MAILBOX="1")
    print("normal", normal["fast"], normal["slow"], "| spin 0 us", timeout["fast"], timeout["slow"], "| no mailbox", nobox["fast"], nobox["slow"])
    assert normal["scenes"] == timeout["scenes"] == nobox["scenes"]
    assert len({s["R"] for s in normal["scenes"]}) == 2 and all(s["R"] > 0 for s in normal["scenes"])
    assert normal["prefilter"] == timeout["prefilter"] == nobox["prefilter"] == "filtered although prefiltered"
    assert normal["fast"] > 0                                   # the ordinary run receives the count through the mailbox
    # a zero spin budget still polls once before it gives up, so a post that is already there counts as fast; the
    # copy + synchronise path must have run for the rest, and without a mailbox for every call
    assert timeout["slow"] > 0 and timeout["fast"] + timeout["slow"] == 7
    assert nobox["fast"] == 0 and nobox["slow"] == 7
    assert normal["regrow"] >= 1        # the third pair starts from the small scene's guess: the chunk is requested twice
