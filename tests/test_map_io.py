"""point_cloud.ply -> device layout.  CPU: header parsing, column lookup by name,
the writer/oracle round trip.  GPU (-m gpu): the HIP conversion kernel against the oracle, through the C ABI."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from gs_localization_amd import map_io
from oracle import map_oracle as MO


def _raw(P, deg, seed=0):
    rng = np.random.default_rng(seed)
    M = (deg + 1) ** 2
    f = np.float32
    return dict(xyz=rng.normal(size=(P, 3)).astype(f), features_dc=rng.normal(size=(P, 1, 3)).astype(f),
                features_rest=rng.normal(size=(P, M - 1, 3)).astype(f), opacity=rng.normal(size=(P, 1)).astype(f) * 3,
                scaling=(rng.normal(size=(P, 3)) - 3).astype(f), rotation=rng.normal(size=(P, 4)).astype(f))


def _write(tmp_path, raw, name="point_cloud.ply"):
    path = str(tmp_path / name)
    map_io.write_ply(path, raw["xyz"], raw["features_dc"], raw["features_rest"], raw["opacity"], raw["scaling"], raw["rotation"])
    return path


@pytest.mark.parametrize("deg", [0, 1, 3])
def test_writer_reader_oracle_round_trip(tmp_path, deg):
    raw = _raw(257, deg)
    path = _write(tmp_path, raw)
    rows, names = map_io.read_ply(path)
    assert rows.shape == (257, 14 + 3 + 3 * ((deg + 1) ** 2 - 1))          # + nx ny nz
    assert names[:6] == ["x", "y", "z", "nx", "ny", "nz"] and names[-4:] == ["rot_0", "rot_1", "rot_2", "rot_3"]
    back = MO.load_ply_ref(np.asarray(rows), names, max_sh_degree=deg)
    for k in raw:
        np.testing.assert_array_equal(back[k], raw[k], err_msg=k)


def test_columns_follow_names_not_positions(tmp_path):
    raw = _raw(5, 1)
    path = _write(tmp_path, raw)
    rows, names = map_io.read_ply(path)
    cols, n_rest = map_io.columns(names, max_sh_degree=1)
    assert n_rest == 9 and len(cols) == 14 + 9
    # canonical order: x y z f_dc_0..2 f_rest_0..8 opacity scale rot ; the file also holds nx ny nz
    assert [names[c] for c in cols[:6]] == ["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2"]
    assert [names[c] for c in cols[6:15]] == [f"f_rest_{i}" for i in range(9)]
    # a file with shuffled property order gives the same values per name (suffixes sorted numerically, f_rest_10 after f_rest_9)
    perm = np.random.default_rng(1).permutation(len(names))
    names2 = [names[i] for i in perm]
    cols2, _ = map_io.columns(names2)
    assert [names2[c] for c in cols2] == [names[c] for c in cols]
    with pytest.raises(ValueError):
        map_io.columns(names, max_sh_degree=3)          # the reference's assert on the number of f_rest_ properties
    with pytest.raises(ValueError):
        map_io.columns([n for n in names if n != "opacity"])


def test_reader_rejects_garbage_and_reads_ascii(tmp_path):
    bad = tmp_path / "bad.ply"
    bad.write_bytes(b"not a ply\n")
    with pytest.raises(ValueError):
        map_io.read_ply(str(bad))
    asc = tmp_path / "a.ply"
    asc.write_text("ply\nformat ascii 1.0\ncomment x\nelement vertex 2\nproperty float x\nproperty float y\nend_header\n1 2\n3.5 4\n")
    rows, names = map_io.read_ply(str(asc))
    assert names == ["x", "y"] and rows.tolist() == [[1.0, 2.0], [3.5, 4.0]]
    mixed = tmp_path / "m.ply"
    hdr = b"ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty double x\nproperty uchar y\nend_header\n"
    rec = np.zeros(2, dtype=[("x", "<f8"), ("y", "u1")]); rec["x"] = [1.5, -2.0]; rec["y"] = [7, 9]
    mixed.write_bytes(hdr + rec.tobytes())
    rows, names = map_io.read_ply(str(mixed))
    assert rows.dtype == np.float32 and rows.tolist() == [[1.5, 7.0], [-2.0, 9.0]]


def test_oracle_activations_match_torch():
    import torch
    raw = _raw(64, 2, seed=3)
    raw["rotation"][0] = 0                                     # F.normalize eps path
    act = MO.activate(raw)
    t = lambda a: torch.tensor(a)
    np.testing.assert_allclose(act["opacity"], torch.sigmoid(t(raw["opacity"])).numpy(), rtol=2e-7, atol=1e-9)
    np.testing.assert_allclose(act["scaling"], torch.exp(t(raw["scaling"])).numpy(), rtol=2e-7)
    np.testing.assert_allclose(act["rotation"], torch.nn.functional.normalize(t(raw["rotation"])).numpy(), rtol=3e-7, atol=1e-9)
    assert act["features"].shape == (64, 9, 3)
    np.testing.assert_array_equal(act["features"], torch.cat((t(raw["features_dc"]), t(raw["features_rest"])), dim=1).numpy())


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("P,deg,shuffle", [(1, 3, False), (63, 0, False), (1000, 3, False), (4097, 3, True), (300, 1, True)])
def test_kernel_matches_oracle(tmp_path, P, deg, shuffle):
    import torch
    from gs_localization_amd import _lib
    lib = _lib.load()
    raw = _raw(P, deg, seed=P)
    raw["rotation"][0] = 0
    path = _write(tmp_path, raw)
    rows, names = map_io.read_ply(path)
    rows = np.array(rows)
    if shuffle:                                                # another writer's property order
        perm = np.random.default_rng(2).permutation(len(names))
        rows, names = np.ascontiguousarray(rows[:, perm]), [names[i] for i in perm]
    cols, n_rest = map_io.columns(names, max_sh_degree=deg)
    M = (deg + 1) ** 2
    dev = torch.device("cuda:0")
    rows_d = torch.from_numpy(rows).to(dev)
    e = lambda *s: torch.full(s, float("nan"), dtype=torch.float32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    ref_raw = MO.load_ply_ref(rows, names, max_sh_degree=deg)
    for activate in (0, 1):
        means, shs, opac, scales, rots = e(P, 3), e(P, M, 3), e(P, 1), e(P, 3), e(P, 4)
        _lib.check(lib.gsr_map_from_ply_rows(P, p(rows_d), rows.shape[1], (C.c_int * len(cols))(*cols), n_rest, activate,
                                             p(means), p(shs), p(opac), p(scales), p(rots), None))
        torch.cuda.synchronize()
        ref = MO.activate(ref_raw) if activate else dict(xyz=ref_raw["xyz"], opacity=ref_raw["opacity"], scaling=ref_raw["scaling"],
                                                         rotation=ref_raw["rotation"],
                                                         features=np.concatenate((ref_raw["features_dc"], ref_raw["features_rest"]), axis=1))
        np.testing.assert_array_equal(means.cpu().numpy(), ref["xyz"])                 # byte moves: exact
        np.testing.assert_array_equal(shs.cpu().numpy(), ref["features"])
        tol = dict(rtol=3e-7, atol=1e-9) if activate else dict(rtol=0, atol=0)          # exp / sigmoid / sqrt: <= 2 ulp
        np.testing.assert_allclose(opac.cpu().numpy(), ref["opacity"], **tol)
        np.testing.assert_allclose(scales.cpu().numpy(), ref["scaling"], **tol)
        np.testing.assert_allclose(rots.cpu().numpy(), ref["rotation"], **tol)


@pytest.mark.gpu
def test_bad_arguments_are_refused():
    import torch
    from gs_localization_amd import _lib
    lib = _lib.load()
    d = torch.zeros(62, device="cuda:0")
    cols = (C.c_int * 59)(*range(59))
    p = C.c_void_p(d.data_ptr())
    assert lib.gsr_map_from_ply_rows(1, p, 62, cols, 46, 1, p, p, p, p, p, None) < 0          # n_rest not an SH basis size
    cols[3] = 62
    assert lib.gsr_map_from_ply_rows(1, p, 62, cols, 45, 1, p, p, p, p, p, None) < 0          # column outside the row
    assert lib.gsr_map_from_ply_rows(0, None, 62, None, 45, 1, None, None, None, None, None, None) == 0


@pytest.mark.gpu
def test_render_from_ply_equals_render_from_tensors(tmp_path):
    """The whole route: raw parameters -> point_cloud.ply -> GaussianMap.from_ply -> render, against the same map
    built from torch-activated tensors."""
    import torch
    from gs_localization_amd import scenes as S
    from tests import replay as PL
    sc = S.small(P=3000, W=96, H=64, sh_degree=3, seed=5)
    raw = dict(xyz=sc.means3D, features_dc=sc.shs[:, :1], features_rest=sc.shs[:, 1:],
               opacity=np.log(sc.opacities / (1 - sc.opacities)).reshape(-1, 1), scaling=np.log(sc.scales),
               rotation=sc.rotations * 1.7)                    # un-normalised on disk
    path = _write(tmp_path, raw)
    dev = torch.device("cuda:0")
    m_ply = PL.GaussianMap.from_ply(path, device=dev, max_sh_degree=3)
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=dev)
    m_ref = PL.GaussianMap(t(raw["xyz"]), torch.cat((t(raw["features_dc"]), t(raw["features_rest"])), dim=1), torch.sigmoid(t(raw["opacity"])),
                           torch.exp(t(raw["scaling"])), torch.nn.functional.normalize(t(raw["rotation"])), 3)
    assert m_ply.active_sh_degree == 3
    H, W = sc.H, sc.W
    vp = PL.QueryFrame(0, PL.intrinsics_projection(sc, dev), sc, dev)
    bg = torch.zeros(3, device=dev)
    with torch.no_grad():
        a = PL.render(vp, m_ply, bg)
        b = PL.render(vp, m_ref, bg)
    for k in ("render", "depth", "opacity"):
        d = (a[k] - b[k]).abs().sum() / b[k].abs().sum().clamp_min(1e-12)
        assert float(d) <= 1e-5, k
    assert int((a["radii"] != b["radii"]).sum()) <= 2
