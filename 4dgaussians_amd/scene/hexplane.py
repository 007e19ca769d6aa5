"""HexPlaneField parameter container, mirroring scene/hexplane.py:134-228 of the reference.

Same constructor arguments, same state_dict keys and shapes (`aabb`, `grids.{level}.{plane}`
of shape [1, C, H, W], the never-called `feature_aggr.*`), same U(0.5, 1.5) initialisation
(scene/hexplane.py:94) and the same flipped aabb convention (`set_aabb(xyz_max, xyz_min)`,
scene/hexplane.py:188-190).  The planes are kept in torch.channels_last memory format: the
logical shape stays [1, C, H, W] (so reference checkpoints load unchanged) while the bytes are
[H][W][C], which makes one bilinear corner one contiguous row for the HIP gather kernel.
The sampling itself lives in libgs4d.so (csrc/deform.hip) and is driven from
4dgaussians_amd/scene/deformation.py; this module has no forward of its own.
"""
import itertools

import torch
import torch.nn as nn


def init_grid_param(grid_nd, in_dim, out_dim, reso, a=0.5, b=1.5):
    assert in_dim == len(reso), "Resolution must have same number of elements as input-dimension"
    assert grid_nd <= in_dim
    coo_combs = list(itertools.combinations(range(in_dim), grid_nd))
    if len(coo_combs) == 10:  # drop the (t, f) plane, scene/hexplane.py:77-78
        coo_combs = coo_combs[:-1]
    grid_coefs = nn.ParameterList()
    for coo_comb in coo_combs:
        p = torch.empty([1, out_dim] + [reso[cc] for cc in coo_comb[::-1]])
        nn.init.uniform_(p, a=a, b=b)
        grid_coefs.append(nn.Parameter(p.contiguous(memory_format=torch.channels_last)))
    return grid_coefs


class HexPlaneField(nn.Module):
    def __init__(self, bounds, planeconfig, multires, hidden_width):
        super().__init__()
        aabb = torch.tensor([[bounds, bounds, bounds], [-bounds, -bounds, -bounds]], dtype=torch.float32)
        self.aabb = nn.Parameter(aabb, requires_grad=False)
        self.grid_config = [planeconfig]
        self.multiscale_res_multipliers = multires
        self.concat_features = True
        self.grids = nn.ModuleList()
        self.feat_dim = 0
        for res in self.multiscale_res_multipliers:
            config = self.grid_config[0].copy()
            config["resolution"] = [r * res for r in config["resolution"][:3]] + config["resolution"][3:]
            gp = init_grid_param(config["grid_dimensions"], config["input_coordinate_dim"],
                                 config["output_coordinate_dim"], config["resolution"])
            self.feat_dim += gp[-1].shape[1]
            self.grids.append(gp)
        self.hidden_width = hidden_width
        self.recur_feat_dim = int(self.feat_dim + self.hidden_width)
        self.feature_aggr = nn.Sequential(  # present in the reference's state_dict, never called
            nn.Linear(self.recur_feat_dim, self.recur_feat_dim), nn.ReLU(), nn.Linear(self.recur_feat_dim, self.feat_dim))

    @property
    def get_aabb(self):
        return self.aabb[0], self.aabb[1]

    def set_aabb(self, xyz_max, xyz_min):
        aabb = torch.tensor([xyz_max, xyz_min], dtype=torch.float32, device=self.aabb.device)
        self.aabb = nn.Parameter(aabb, requires_grad=False)
