"""CPU restatement of distCUDA2 -- TEST INFRASTRUCTURE ONLY.

gaussian_splatting/submodules/simple-knn/simple_knn.cu:153-220: for every point the three smallest squared distances
to OTHER points (by index; coincident points count with distance 0), summed and divided by 3 in float32.  The
reference's Morton order and box pruning only decide which candidates are visited -- the pruning is conservative, so
its result is the exact 3-NN mean; here the neighbours come from a k-d tree and the distances are re-evaluated in
float32 as dx*dx + dy*dy + dz*dz.  With fewer than four points the untouched FLT_MAX entries make the mean overflow
to inf, as in the reference.

Parity unpinned against an execution of the reference (CUDA, not buildable here); cross-checked against brute force."""
import numpy as np


def dist2_knn3(points):
    pts = np.ascontiguousarray(points, np.float32)
    P = pts.shape[0]
    out = np.empty(P, np.float32)
    FMAX = np.float32(3.402823466e38)
    if P == 0:
        return out
    from scipy.spatial import cKDTree
    k = min(P, 8)                     # a few spare candidates: float32 re-evaluation may reorder near ties
    _, nbr = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=k)
    nbr = nbr.reshape(P, k)
    for i in range(P):
        cand = nbr[i][nbr[i] != i][: max(k - 1, 0)]
        d = pts[cand] - pts[i]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        best = np.sort(d2.astype(np.float32))[:3]
        best = np.concatenate([best, np.full(3 - best.size, FMAX, np.float32)])
        with np.errstate(over="ignore"):
            out[i] = (best[0] + best[1] + best[2]) / np.float32(3.0)
    return out


def dist2_knn3_bruteforce(points):
    pts = np.ascontiguousarray(points, np.float32)
    P = pts.shape[0]
    d = pts[:, None, :] - pts[None, :, :]
    d2 = ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).astype(np.float32)
    d2[np.arange(P), np.arange(P)] = np.float32(3.402823466e38)
    best = np.sort(d2, axis=1)[:, :3]
    if best.shape[1] < 3:
        best = np.concatenate([best, np.full((P, 3 - best.shape[1]), 3.402823466e38, np.float32)], axis=1)
    with np.errstate(over="ignore"):
        return ((best[:, 0] + best[:, 1] + best[:, 2]) / np.float32(3.0)).astype(np.float32)
