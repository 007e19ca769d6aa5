"""TEST INFRASTRUCTURE ONLY (imported by tests/): CPU restatement of the fine-stage regulariser dictionaries of the
reference, gaussian_renderer/renderer.py:151-203, with the reference's own torch expressions (the gathers
`vel[idx] - vel[None, :, None]` are materialised as the reference does) and an exact brute-force neighbour search
in place of pointops2's knnquery (pointops2/src/knnquery/knnquery_cuda_kernel.cu:65-108; utils/general_utils.py:144-158
applies sqrt and returns int64 indices).  Parity status: the expressions are the reference's PyTorch ones run on the
CPU -- the reference's own tests hold no vectors for them ("parity unpinned" against recorded outputs)."""
import torch


def knn_bruteforce(p, k):
    """p (n, 3) -> idx int64 (1, n, k), dist (1, n, k): ascending distance, ties by lower index."""
    d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1)
    order = torch.argsort(d2, dim=1, stable=True)[:, :k]
    return order[None], torch.sqrt(torch.gather(d2, 1, order))[None]


def neighbor_diff_norm(x, idx):
    return torch.norm(x[idx] - x[None, :, None], p=2, dim=-1)          # renderer.py:163-165,194-195,201-202


def temporal_regulation(means3D, means3D_final, means3D_prev, means3D_nxt, scales_final, scales_prev, scales_nxt,
                        shs_final, shs_prev, shs_nxt, rotations_final, rotations_prev, rotations_next, idx, dist):
    """renderer.py:183-203 given the neighbour lists."""
    n, k = means3D.shape[0], idx.shape[-1]
    reg = {}
    momentum_all = torch.abs(means3D_nxt + means3D_prev - 2 * means3D_final)                      # :183
    reg["momentum"] = momentum_all.mean()                                                          # :184
    reg["scales"] = torch.abs(scales_nxt + scales_prev - 2 * scales_final).sum(dim=1).mean()       # :185
    reg["shs"] = torch.abs(shs_nxt + shs_prev - 2 * shs_final).mean()                              # :186
    vel_xyz = (means3D_nxt - means3D_final) / 2                                                    # :189
    vel_long = (means3D_nxt - means3D)                                                             # :190
    weight = torch.exp(-100 * dist)                                                                # :193
    mom_xyz_dist = neighbor_diff_norm(vel_xyz, idx)                                                # :194
    mom_long_dist = neighbor_diff_norm(vel_long, idx)                                              # :195
    reg["knn_xyz"] = (weight * mom_xyz_dist).sum() / k / n                                         # :196
    reg["knn_long"] = (weight * torch.abs(mom_long_dist - mom_xyz_dist)).sum() / k / n             # :197
    rot_v1 = rotations_final - rotations_prev                                                      # :199
    rot_v2 = rotations_next - rotations_final                                                      # :200
    mom_rot_dist1 = neighbor_diff_norm(rot_v1, idx)                                                # :201
    mom_rot_dist2 = neighbor_diff_norm(rot_v2, idx)                                                # :202
    reg["knn_rot"] = (weight * (mom_rot_dist1 + mom_rot_dist2).mean()).sum() / k / n               # :203
    return reg


def recurrent_regulation(means3D_in, means3D_canonical, means3D_final, rotations_in, rotations_final, scales_in,
                         scales_final, shs_in, shs_final, idx, dist):
    """renderer.py:155-171 given the neighbour lists."""
    n, k = means3D_in.shape[0], idx.shape[-1]
    reg = {}
    vel_xyz = means3D_final - means3D_in
    vel_long = means3D_final - means3D_canonical
    vel_rot = rotations_final - rotations_in
    vel_scale = scales_final - scales_in
    vel_shs = shs_final - shs_in
    weight = torch.exp(-100 * dist)
    mom_xyz_dist = neighbor_diff_norm(vel_xyz, idx)
    mom_long_dist = neighbor_diff_norm(vel_long, idx)
    mom_rot_dist = neighbor_diff_norm(vel_rot, idx)
    reg["knn_xyz"] = (weight * mom_xyz_dist).sum() / k / n
    reg["knn_long"] = (weight * torch.abs(mom_long_dist - mom_xyz_dist)).sum() / k / n
    reg["knn_rot"] = (weight * mom_rot_dist).sum() / k / n
    reg["momentum"] = vel_xyz.abs().sum(dim=1).mean()
    reg["scales"] = vel_scale.abs().sum(dim=1).mean()
    reg["shs"] = vel_shs.mean()
    return reg
