"""Helpers to load tests/golden/deformation_*.npz (data produced by the reference itself)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["mutant_small", "force_small", "plain_small"]


def load_case(name):
    from oracle.deformation_ref import DeformConfig
    d = np.load(os.path.join(GOLDEN, f"deformation_{name}.npz"))
    raw = json.loads(str(d["cfg_json"]))
    cfg = DeformConfig(**{k: v for k, v in raw.items()})
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("sd/")}
    ins = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("in/")}
    outs = {k[4:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("out/")}
    ws = {k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w/")}
    gins = {k[4:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("gin/")}
    grads = {k[5:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("grad/")}
    return cfg, sd, ins, outs, ws, gins, grads
