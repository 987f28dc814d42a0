"""CPU restatement (numpy, float32) of the reference's map loading -- TEST INFRASTRUCTURE ONLY.

  load_ply_ref   gs_localization/pipelines/tools/gaussian_model.py:377-467 (= gaussian_splatting/scene/
                 gaussian_model.py:215-256): per-property gathering, (P, 3, M-1) reshape + transpose(1, 2)
  activate       the getters render() calls every iteration, tools/gaussian_model.py:77-96:
                 exp / sigmoid / torch.nn.functional.normalize (eps 1e-12) / cat(features_dc, features_rest)

Parity unpinned against an execution of the reference: its loader needs `plyfile`, which is not in this image;
the restatement is anchored on the reference's own writer/reader pair (property names and order of
construct_list_of_attributes, gaussian_model.py:177-189) and round-trips through a writer restated from save_ply."""
import numpy as np


def load_ply_ref(rows, names, max_sh_degree=3):
    el = {n: np.asarray(rows[:, i], np.float32) for i, n in enumerate(names)}      # plydata.elements[0][name]
    xyz = np.stack((el["x"], el["y"], el["z"]), axis=1)
    opacities = el["opacity"][..., np.newaxis]
    P = xyz.shape[0]
    features_dc = np.zeros((P, 3, 1), np.float32)
    for c in range(3):
        features_dc[:, c, 0] = el[f"f_dc_{c}"]
    extra = sorted([n for n in names if n.startswith("f_rest_")], key=lambda x: int(x.split("_")[-1]))
    assert len(extra) == 3 * (max_sh_degree + 1) ** 2 - 3
    features_extra = np.zeros((P, len(extra)), np.float32)
    for i, n in enumerate(extra):
        features_extra[:, i] = el[n]
    features_extra = features_extra.reshape((P, 3, (max_sh_degree + 1) ** 2 - 1))
    scale_names = sorted([n for n in names if n.startswith("scale_")], key=lambda x: int(x.split("_")[-1]))
    rot_names = sorted([n for n in names if n.startswith("rot")], key=lambda x: int(x.split("_")[-1]))
    scales = np.stack([el[n] for n in scale_names], axis=1)
    rots = np.stack([el[n] for n in rot_names], axis=1)
    return dict(xyz=xyz, features_dc=features_dc.transpose(0, 2, 1).copy(), features_rest=features_extra.transpose(0, 2, 1).copy(),
                opacity=opacities, scaling=scales, rotation=rots)


def activate(raw):
    f32 = np.float32
    q = raw["rotation"].astype(f32)
    nrm = np.maximum(np.sqrt((q * q).sum(axis=1, keepdims=True, dtype=f32)), f32(1e-12))
    return dict(xyz=raw["xyz"], features=np.concatenate((raw["features_dc"], raw["features_rest"]), axis=1),
                opacity=(f32(1) / (f32(1) + np.exp(-raw["opacity"].astype(f32)))).astype(f32),
                scaling=np.exp(raw["scaling"].astype(f32)).astype(f32), rotation=(q / nrm).astype(f32))
