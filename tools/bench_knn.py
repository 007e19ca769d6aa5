"""k = 100 nearest-neighbour query: device time per call at the fork's densification cap
(P = 14 000, train.py:307) and at C2's P = 100 000, next to the CPU oracle (a restatement of the reference's brute
force) timed on a bounded sample of the queries.  Prints one JSON line per size."""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P_ = importlib.import_module("4dgaussians_amd.pointops")
from oracle import knn_oracle as K  # noqa: E402

for n in (14_000, 100_000, 1_000_000):
    rng = np.random.default_rng(n)
    pts = rng.uniform(-1.3, 1.3, (n, 3)).astype(np.float32)
    x = torch.from_numpy(pts).cuda()
    off = torch.tensor([n], dtype=torch.int32)
    for _ in range(3):
        P_.knnquery(100, x, None, off, off)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        idx, dist = P_.knnquery(100, x, None, off, off)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    sample = max(1, min(n, int(2e8 // n)))              # ~2e8 distance evaluations on one CPU core
    t0 = time.perf_counter()
    K.knn_query(pts, pts[:sample], 100)
    cpu_ms_full = (time.perf_counter() - t0) * 1e3 * n / sample
    print(json.dumps({"op": "knn_query k=100 (self)", "points": n, "gpu_ms": round(ms, 3),
                      "queries_per_s": round(n / ms * 1e3), "out_bytes": n * 800,
                      "cpu_oracle_ms_extrapolated": round(cpu_ms_full, 1), "cpu_sample_queries": sample, "cores": 1}), flush=True)
