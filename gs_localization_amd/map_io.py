"""point_cloud.ply <-> the packed device layout of the localisation map.

Host side of the reference's `GaussianModel.load_ply` / `save_ply`
(gs_localization/pipelines/tools/gaussian_model.py:377-467, gaussian_splatting/scene/gaussian_model.py:177-256)
without `plyfile`: the header is parsed here, the vertex rows are uploaded exactly as stored and one HIP kernel
(`gsr_map_from_ply_rows`, include/gsr.h) gathers the columns, lays the SH coefficients out as [P, M, 3] and
applies the activations once.  There is no CPU path for the conversion itself."""
import ctypes as C
import os

import numpy as np

_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
              "char": "i1", "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
              "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def read_ply(path):
    """Returns (rows, names): rows = float32 array [P, len(names)] of the `vertex` element (a read-only memmap
    when the file stores plain little-endian floats, as 3DGS writes it), names = property names in file order."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex, seen_vertex = None, None, [], False, False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if seen_vertex and tok[1] != "vertex":
                    in_vertex = False
                if tok[1] == "vertex":
                    if seen_vertex:
                        raise ValueError(f"{path}: more than one vertex element")
                    in_vertex, seen_vertex, count = True, True, int(tok[2])
                elif not seen_vertex:
                    raise ValueError(f"{path}: elements before `vertex` are not supported")
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties in the vertex element are not supported")
                if tok[1] not in _PLY_TYPES:
                    raise ValueError(f"{path}: unknown property type {tok[1]}")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if count is None or not props:
            raise ValueError(f"{path}: no vertex element")
        offset = f.tell()
    names = [n for n, _ in props]
    if fmt == "binary_little_endian":
        if all(t == "<f4" for _, t in props):
            rows = np.memmap(path, dtype="<f4", mode="r", offset=offset, shape=(count, len(props)))
        else:
            rec = np.memmap(path, dtype=np.dtype(props), mode="r", offset=offset, shape=(count,))
            rows = np.stack([np.asarray(rec[n], np.float32) for n in names], axis=1)
    elif fmt == "ascii":
        rows = _read_ascii(path, offset, count, len(props))
    else:
        raise ValueError(f"{path}: PLY format {fmt} is not supported (binary_little_endian or ascii)")
    return rows, names


def _read_ascii(path, offset, count, ncol):
    with open(path, "rb") as f:
        f.seek(offset)
        data = np.array(f.read().split()[:count * ncol], dtype=np.float32)
    if data.size != count * ncol:
        raise ValueError(f"{path}: truncated ascii vertex data")
    return data.reshape(count, ncol)


def columns(names, max_sh_degree=None):
    """Column indices in the order gsr_map_from_ply_rows expects (x, y, z, f_dc_0..2, f_rest_*, opacity, scale_*,
    rot_*), found by NAME and numeric suffix like the reference (load_ply sorts f_rest_/scale_/rot names by suffix).
    Returns (cols, n_rest)."""
    pos = {n: i for i, n in enumerate(names)}

    def numbered(prefix):
        got = sorted((int(n.split("_")[-1]), i) for n, i in pos.items() if n.startswith(prefix))
        return [i for _, i in got]
    try:
        cols = [pos["x"], pos["y"], pos["z"], pos["f_dc_0"], pos["f_dc_1"], pos["f_dc_2"]]
        rest = numbered("f_rest_")
        cols += rest + [pos["opacity"]]
    except KeyError as e:
        raise ValueError(f"PLY vertex element lacks property {e}") from None
    scale, rot = numbered("scale_"), numbered("rot")
    if len(scale) != 3 or len(rot) != 4:
        raise ValueError("PLY vertex element needs scale_0..2 and rot_0..3")
    n_rest = len(rest)
    if n_rest % 3 != 0 or round((n_rest // 3 + 1) ** 0.5) ** 2 != n_rest // 3 + 1:
        raise ValueError(f"{n_rest} f_rest_ properties do not form an SH basis")
    if max_sh_degree is not None and n_rest != 3 * (max_sh_degree + 1) ** 2 - 3:     # the reference's assert
        raise ValueError(f"expected {3 * (max_sh_degree + 1) ** 2 - 3} f_rest_ properties, found {n_rest}")
    return cols + scale + rot, n_rest


def write_ply(path, xyz, features_dc, features_rest, opacity, scaling, rotation):
    """save_ply (gaussian_model.py:197-213): RAW parameters (logit opacity, log scale, un-normalised quaternion),
    features_dc [P,1,3], features_rest [P,M-1,3]; property order of construct_list_of_attributes."""
    xyz = np.asarray(xyz, np.float32)
    P = xyz.shape[0]
    f_dc = np.asarray(features_dc, np.float32).transpose(0, 2, 1).reshape(P, -1)
    f_rest = np.asarray(features_rest, np.float32).transpose(0, 2, 1).reshape(P, -1)
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(f_dc.shape[1])] + \
            [f"f_rest_{i}" for i in range(f_rest.shape[1])] + ["opacity"] + [f"scale_{i}" for i in range(3)] + \
            [f"rot_{i}" for i in range(4)]
    rows = np.concatenate([xyz, np.zeros_like(xyz), f_dc, f_rest, np.asarray(opacity, np.float32).reshape(P, 1),
                           np.asarray(scaling, np.float32).reshape(P, 3), np.asarray(rotation, np.float32).reshape(P, 4)],
                          axis=1).astype("<f4")
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\n")
        f.write(f"element vertex {P}\n".encode())
        for n in names:
            f.write(f"property float {n}\n".encode())
        f.write(b"end_header\n")
        f.write(rows.tobytes())
    return names


def load_map_tensors(path, device, activate=True, max_sh_degree=None):
    """point_cloud.ply -> (means3D, shs [P,M,3], opacities [P,1], scales, rotations, sh_degree) on `device`
    through the HIP kernel.  Raises if the extension is missing (no CPU conversion path)."""
    import torch
    from . import _lib
    lib = _lib.load()
    rows, names = read_ply(path)
    cols, n_rest = columns(names, max_sh_degree)
    P, row_floats = rows.shape
    M = 1 + n_rest // 3
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("load_map_tensors needs a HIP device")
    with torch.cuda.device(dev):
        rows_d = torch.from_numpy(np.array(rows, dtype=np.float32, order="C")).to(dev, non_blocking=False)
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        means, shs, opac, scales, rots = e(P, 3), e(P, M, 3), e(P, 1), e(P, 3), e(P, 4)
        carr = (C.c_int * len(cols))(*cols)
        p = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(lib.gsr_map_from_ply_rows(P, p(rows_d), row_floats, carr, n_rest, 1 if activate else 0, p(means), p(shs),
                                             p(opac), p(scales), p(rots), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.current_stream().synchronize()      # rows_d is freed on return
    return means, shs, opac, scales, rots, int(round(M ** 0.5)) - 1
