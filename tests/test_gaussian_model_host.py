"""Host-side logic of 4dgaussians_amd/scene/gaussian_model.py on the CPU: the PLY
layout and round trip, the checkpoint tuple, the learning-rate schedule, and the optimizer-state surgery of
densify / clone / split / prune, with torch.optim.AdamW injected in place of the fused optimizer (the surgery
only touches `param_groups` and `state`, which FusedAdamW shares with it)."""
import types

import numpy as np
import pytest
import torch

from conftest import pkg
from test_gpu_deformation import make_args
from oracle import deformation_ref as R

GM = pkg("scene.gaussian_model")


def opt_args():
    return types.SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016,
                                 position_lr_delay_mult=0.01, position_lr_max_steps=20000, deformation_lr_init=0.00016,
                                 deformation_lr_final=0.000016, deformation_lr_delay_mult=0.01, grid_lr_init=0.0016,
                                 grid_lr_final=0.00016, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001)


def model(n=50, seed=0):
    cfg = R.DeformConfig(plane_channels=8, resolution=[4, 4, 4, 3], multires=[1])
    torch.manual_seed(seed)
    m = GM.GaussianModel(3, make_args(cfg), device="cpu", optimizer_cls=torch.optim.AdamW)
    g = torch.Generator().manual_seed(seed)
    P = lambda *s: torch.nn.Parameter(torch.randn(*s, generator=g))
    m._xyz, m._features_dc, m._features_rest = P(n, 3), P(n, 1, 3), P(n, 15, 3)
    m._opacity, m._scaling, m._rotation = P(n, 1), torch.nn.Parameter(torch.randn(n, 3, generator=g) * 0.3 - 2), P(n, 4)
    m._deformation_table = torch.ones(n, dtype=torch.bool)
    m.max_radii2D = torch.zeros(n)
    m.spatial_lr_scale = 5.0
    m.training_setup(opt_args())
    return m


def step(m):
    for grp in m.optimizer.param_groups:
        for p in grp["params"]:
            if p.requires_grad and p.dim() and "grid" not in grp["name"] and grp["name"] != "deformation":
                p.grad = torch.ones_like(p)
    m.optimizer.step()
    m.optimizer.zero_grad(set_to_none=True)


def test_ply_layout_and_round_trip(tmp_path):
    m = model(37)
    path = str(tmp_path / "point_cloud" / "iteration_10" / "point_cloud.ply")
    m.save_ply(path)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    names = [l.split()[2] for l in lines[3:]]
    assert all(l.startswith("property float ") for l in lines[3:])
    # scene/gaussian_model.py:244-256: x,y,z,nx,ny,nz, 3 f_dc, 45 f_rest, opacity, 3 scales, 4 rotations = 62 floats
    assert names == ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(45)] + \
        ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
    table = np.frombuffer(body, "<f4").reshape(37, 62)
    np.testing.assert_array_equal(table[:, :3], m._xyz.detach().numpy())
    assert (table[:, 3:6] == 0).all()
    # f_rest is stored channel-major: (P, 15, 3) -> transpose(1, 2) -> (P, 3, 15) -> flatten  (gaussian_model.py:285)
    np.testing.assert_array_equal(table[:, 9:54], m._features_rest.detach().transpose(1, 2).flatten(1).numpy())
    m2 = model(5, seed=1)
    m2.load_ply(path)
    for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        assert torch.equal(getattr(m2, k).detach(), getattr(m, k).detach()), k
        assert getattr(m2, k).requires_grad
    assert m2.active_sh_degree == 3
    # ascii PLY files (other writers) load too
    apath = str(tmp_path / "a.ply")
    with open(apath, "w") as f:
        f.write("\n".join(lines[:1] + ["format ascii 1.0"] + lines[2:]) + "\nend_header\n")
        np.savetxt(f, table, fmt="%.9g")
    m3 = model(5, seed=2)
    m3.load_ply(apath)
    assert torch.allclose(m3._scaling.detach(), m._scaling.detach(), rtol=1e-6)


def test_lr_schedule_and_groups():
    m = model()
    assert [g["name"] for g in m.optimizer.param_groups] == ["xyz", "deformation", "grid", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    assert abs(m.optimizer.param_groups[0]["lr"] - 0.00016 * 5.0) < 1e-12 and m.optimizer.param_groups[4]["lr"] == 0.0025 / 20.0
    m.update_learning_rate(10000)
    want = np.exp(0.5 * np.log(0.00016 * 5) + 0.5 * np.log(0.0000016 * 5))       # delay_steps = 0 -> pure log-lerp
    assert abs(m.optimizer.param_groups[0]["lr"] - want) < 1e-12
    assert abs(m.optimizer.param_groups[2]["lr"] - np.exp(0.5 * np.log(0.0016 * 5) + 0.5 * np.log(0.00016 * 5))) < 1e-12
    f = GM.get_expon_lr_func(1e-2, 1e-4, lr_delay_steps=100, lr_delay_mult=0.1, max_steps=1000)
    assert abs(f(0) - 1e-3) < 1e-12 and abs(f(1000) - 1e-4) < 1e-12 and f(-1) == 0.0


def test_prune_clone_split_keep_optimizer_state_aligned():
    m = model(60)
    step(m)
    old = {g["name"]: (g["params"][0].detach().clone(), m.optimizer.state[g["params"][0]]["exp_avg"].clone())
           for g in m.optimizer.param_groups if len(g["params"]) == 1}
    mask = torch.zeros(60, dtype=torch.bool)
    mask[::3] = True
    m.prune_points(mask)
    assert m.get_xyz.shape[0] == 40 and m.denom.shape == (40, 1) and m._deformation_table.shape == (40,)
    for g in m.optimizer.param_groups:
        if len(g["params"]) == 1:
            p = g["params"][0]
            assert torch.equal(p.detach(), old[g["name"]][0][~mask])
            assert torch.equal(m.optimizer.state[p]["exp_avg"], old[g["name"]][1][~mask])
            assert p is getattr(m, {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
                                    "scaling": "_scaling", "rotation": "_rotation"}[g["name"]])
    step(m)                                                        # state and parameters still line up

    # clone: small Gaussians with a large accumulated gradient are duplicated, moments of the copies start at zero
    n = m.get_xyz.shape[0]
    grads = torch.zeros(n, 1)
    grads[:5] = 1.0
    m.percent_dense = 1e9                                          # everything counts as "small"
    before = m._xyz.detach().clone()
    m.densify_and_clone(grads, 0.5, 1.0)
    assert m.get_xyz.shape[0] == n + 5
    assert torch.equal(m._xyz.detach()[n:], before[:5])
    assert float(m.optimizer.state[m._xyz]["exp_avg"][n:].abs().max()) == 0.0
    assert m.xyz_gradient_accum.shape == (n + 5, 1) and float(m.xyz_gradient_accum.abs().max()) == 0.0

    # split: large Gaussians are replaced by N = 2 samples with scales / (0.8 N)
    n = m.get_xyz.shape[0]
    m.percent_dense = 0.0                                          # everything counts as "large"
    grads = torch.zeros(n, 1)
    grads[3] = grads[7] = 1.0
    s_before = m.get_scaling.detach().clone()
    torch.manual_seed(0)
    m.densify_and_split(grads, 0.5, 1.0)
    assert m.get_xyz.shape[0] == n + 2 * 2 - 2
    assert torch.allclose(m.get_scaling.detach()[-4:], torch.cat([s_before[[3, 7]], s_before[[3, 7]]]) / 1.6, rtol=1e-5)
    step(m)

    # prune by opacity / screen size / world size (gaussian_model.py:583-591)
    n = m.get_xyz.shape[0]
    with torch.no_grad():
        m._opacity[:] = 3.0
        m._opacity[0] = -10.0                                      # sigmoid -> ~0: below min_opacity
        m._scaling[:] = -3.0
        m._scaling[1, 0] = 2.0                                     # exp(2) > 0.1 * extent
    m.max_radii2D = torch.zeros(n)
    m.max_radii2D[2] = 100.0
    m.prune(0.0, 0.005, 1.0, 40)
    assert m.get_xyz.shape[0] == n - 3
    m.reset_opacity()
    assert float(m.get_opacity.max()) <= 0.01 + 1e-6 and m._opacity is m.optimizer.param_groups[5]["params"][0]
    try:
        m.grow()
        assert False
    except NotImplementedError:
        pass


def test_capture_restore_and_deformation_files(tmp_path):
    m = model(20)
    step(m)
    ckpt = m.capture()
    assert len(ckpt) == 14
    torch.save((ckpt, 7), str(tmp_path / "chkpnt_fine_7.pth"))
    (args_back, it) = torch.load(str(tmp_path / "chkpnt_fine_7.pth"), weights_only=False)   # our own file
    m2 = model(3, seed=4)
    m2.restore(args_back, opt_args())
    assert it == 7 and torch.equal(m2._xyz.detach(), m._xyz.detach()) and m2.spatial_lr_scale == 5.0
    assert torch.equal(m2.optimizer.state[m2._xyz]["exp_avg"], m.optimizer.state[m._xyz]["exp_avg"])
    for (k, a), (_, b) in zip(m._deformation.state_dict().items(), m2._deformation.state_dict().items()):
        assert torch.equal(a, b), k
    m._deformation_accum = torch.rand(20, 3)
    m.save_deformation(str(tmp_path / "pc"))
    m3 = model(20, seed=9)
    m3.load_model(str(tmp_path / "pc"))
    assert torch.equal(m3._deformation_accum, m._deformation_accum)
    for (k, a), (_, b) in zip(m._deformation.state_dict().items(), m3._deformation.state_dict().items()):
        assert torch.equal(a, b), k


def test_ply_known_answer_bytes(tmp_path):
    """Byte-level known-answer test of the PLY writer / reader.  PARITY UNPINNED against a
    reference-written file: the reference ships no .ply and `plyfile` (its writer, scene/gaussian_model.py:290-295) is
    not importable here.  The expected bytes below are written out by hand from the PLY 1.0 format as plyfile emits it
    for PlyData([PlyElement.describe(structured_array_of_f4, 'vertex')]): text header with one `property float <name>`
    line per field ('f4' -> "float"), `end_header\\n`, then the records as packed little-endian float32."""
    import struct
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "opacity"]
    rows = [[1.0, -2.5, 3.25, 0.0, 0.0, 0.0, 0.5, -4.0], [1e-3, 2e10, -0.0, 0.0, 0.0, 0.0, float("inf"), 7.0]]
    want = (b"ply\nformat binary_little_endian 1.0\nelement vertex 2\n"
            b"property float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n"
            b"property float f_dc_0\nproperty float opacity\nend_header\n")
    want += b"".join(struct.pack("<8f", *r) for r in rows)
    # spot values spelled out as hex: 1.0 = 3f800000, -2.5 = c0200000, 3.25 = 40500000 (little endian on disk)
    assert want[want.index(b"end_header\n") + 11:][:12] == bytes.fromhex("0000803f" "000020c0" "00005040")
    path = str(tmp_path / "kat.ply")
    GM.write_ply(path, names, np.asarray(rows, np.float64))          # float64 input is narrowed to float32 on write
    assert open(path, "rb").read() == want
    back = GM.read_ply(path)
    assert list(back) == names
    np.testing.assert_array_equal(np.stack([back[n] for n in names], 1), np.asarray(rows, np.float32))

    # a point cloud as the reference's storePly writes it (scene/dataset_readers.py: x y z nx ny nz float, red green blue
    # uchar; 27-byte records), with a comment line and a second element after the vertices, as other writers add
    head = (b"ply\nformat binary_little_endian 1.0\ncomment made by hand\nelement vertex 3\n"
            b"property float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n"
            b"property uchar red\nproperty uchar green\nproperty uchar blue\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n")
    recs = [(0.5, 1.5, -1.0, 0.0, 0.0, 1.0, 255, 0, 17), (2.0, 0.0, 0.0, 1.0, 0.0, 0.0, 1, 2, 3), (-7.0, 8.0, 9.0, 0.0, 1.0, 0.0, 128, 64, 32)]
    body = b"".join(struct.pack("<6f3B", *r) for r in recs)
    assert len(body) == 3 * 27
    p2 = str(tmp_path / "points3D.ply")
    open(p2, "wb").write(head + body)
    v = GM.read_ply(p2)
    np.testing.assert_array_equal(v["x"], np.float32([0.5, 2.0, -7.0]))
    np.testing.assert_array_equal(v["nz"], np.float32([1.0, 0.0, 0.0]))
    assert v["red"].dtype == np.uint8 and v["blue"].tolist() == [17, 3, 32] and v["green"].tolist() == [0, 2, 64]
    # big-endian files of the same content read the same
    p3 = str(tmp_path / "be.ply")
    open(p3, "wb").write(head.replace(b"binary_little_endian", b"binary_big_endian") + b"".join(struct.pack(">6f3B", *r) for r in recs))
    v3 = GM.read_ply(p3)
    for k in v:
        np.testing.assert_array_equal(v3[k], v[k])
    # truncated header / wrong magic fail loudly
    open(p3, "wb").write(b"ply\nformat binary_little_endian 1.0\nelement vertex 1\n")
    with pytest.raises(ValueError):
        GM.read_ply(p3)
    open(p3, "wb").write(b"plx\n")
    with pytest.raises(ValueError):
        GM.read_ply(p3)


def test_formats_match_the_reference_text(tmp_path):
    """The PLY vertex layout and the checkpoint tuple against tests/golden/format_golden.json, which
    tests/golden/make_format_golden.py reads out of the REFERENCE's own source (construct_list_of_attributes executed on
    the shapes of an SH-degree-3 model; dtype and column order from save_ply; tuple order from capture / restore,
    scene/gaussian_model.py:88-120,244-256,281-295) -- the format is pinned to the reference text, not to this
    repository's own known-answer bytes."""
    import json
    import os
    import struct
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "format_golden.json")))
    m = model(5, seed=3)
    # property names / order / dtype as the header states them, and the record as the bytes hold it
    assert m.construct_list_of_attributes() == gold["ply"]["properties"]
    path = str(tmp_path / "pc.ply")
    m.save_ply(path)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0"
    assert lines[2] == f"element {gold['ply']['element']} 5"
    want_type = {"f4": "float"}[gold["ply"]["dtype"]]
    assert [l for l in lines[3:] if l.startswith("property")] == [f"property {want_type} {n}" for n in gold["ply"]["properties"]]
    n = gold["ply"]["n_properties"]
    assert len(body) == 5 * n * 4
    # column order of a record = the reference's np.concatenate tuple (xyz, normals, f_dc, f_rest, opacities, scale, rotation)
    cols = {"xyz": m._xyz.detach().numpy(), "normals": np.zeros((5, 3), np.float32),
            "f_dc": m._features_dc.detach().transpose(1, 2).flatten(start_dim=1).numpy(),
            "f_rest": m._features_rest.detach().transpose(1, 2).flatten(start_dim=1).numpy(),
            "opacities": m._opacity.detach().numpy(), "scale": m._scaling.detach().numpy(), "rotation": m._rotation.detach().numpy()}
    want = np.concatenate([cols[c] for c in gold["ply"]["record_columns"]], axis=1).astype("<f4")
    got = np.frombuffer(body, dtype="<f4").reshape(5, n)
    np.testing.assert_array_equal(got, want)
    assert struct.unpack("<f", body[:4])[0] == float(m._xyz[0, 0])
    # checkpoint tuple: same length, same element at every position
    ckpt = m.capture()
    assert len(ckpt) == gold["checkpoint"]["length"]
    for i, name in enumerate(gold["checkpoint"]["capture"]):
        if name.endswith("()"):
            obj = m
            for part in name[:-2].split("."):
                obj = getattr(obj, part)
            ref = obj()
            assert type(ckpt[i]) is type(ref) and list(ckpt[i].keys()) == list(ref.keys()), name
        else:
            ref = getattr(m, name)
            assert ckpt[i] is ref or ckpt[i] == ref, name


def test_spatial_order_is_a_pure_permutation_of_the_same_edit():
    """spatial_order=True (no reference counterpart): densify leaves the rows in the Morton order of their positions; the
    model is the SAME set of rows -- parameters, Adam moments and the deformation table move together -- as the one
    the reference order produces, and reorder_spatially() alone changes nothing but the order."""
    def run(spatial):
        m = model(80, seed=2)
        m.spatial_order = spatial
        step(m)
        torch.manual_seed(11)
        with torch.no_grad():
            m._scaling[:] = torch.where(torch.arange(80)[:, None] % 2 == 0, torch.tensor(-1.0), torch.tensor(-6.0))
        m.percent_dense = 0.05
        m.xyz_gradient_accum = torch.rand(80, 1, generator=torch.Generator().manual_seed(3))
        m.denom = torch.ones(80, 1)
        m._deformation_table = torch.arange(80) % 3 == 0
        m.densify(0.5, 0.005, 1.0, None)
        return m

    a, b = run(False), run(True)
    assert a._xyz.shape == b._xyz.shape and a._xyz.shape[0] > 80
    key = lambda m: torch.cat([m._xyz.detach(), m._scaling.detach(), m._opacity.detach(), m._features_dc.detach().flatten(1),
                               m.optimizer.state[m._xyz]["exp_avg"], m.optimizer.state[m._rotation]["exp_avg_sq"],
                               m._deformation_table[:, None].float()], dim=1)
    ka, kb = key(a), key(b)
    sort_rows = lambda k: k[np.lexsort(k.numpy().T[::-1])]
    assert torch.equal(sort_rows(ka), sort_rows(kb))                       # the same rows ...
    order = GM.GaussianModel._morton_order(b._xyz.detach())
    assert torch.equal(order, torch.arange(order.numel()))                 # ... stored in Morton order
    assert not torch.equal(GM.GaussianModel._morton_order(a._xyz.detach()), torch.arange(order.numel()))
    before = key(a)
    perm = a.reorder_spatially()
    assert torch.equal(key(a), before[perm]) and a.denom.shape == (a._xyz.shape[0], 1)
    step(a)                                                                # optimizer state still lines up
