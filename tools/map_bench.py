"""Times gsr_map_from_ply_rows (point_cloud.ply rows -> device layout) on a 1 M-Gaussian SH3 map:
bytes moved / kernel time against the HBM peak."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gs_localization_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
P, RF, NREST = 1_000_000, 62, 45
names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(45)] + ["opacity"] + \
        [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
from gs_localization_amd import map_io
cols, n_rest = map_io.columns(names, 3)
rows = torch.randn(P, RF, device=dev)
e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
means, shs, opac, scales, rots = e(P, 3), e(P, 16, 3), e(P, 1), e(P, 3), e(P, 4)
p = lambda t: C.c_void_p(t.data_ptr())
carr = (C.c_int * len(cols))(*cols)
def run():
    _lib.check(lib.gsr_map_from_ply_rows(P, p(rows), RF, carr, n_rest, 1, p(means), p(shs), p(opac), p(scales), p(rots),
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)))
for _ in range(3): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): run()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
bytes_moved = P * (RF * 4 + (3 + 48 + 1 + 3 + 4) * 4)
print(f"gsr_map_from_ply_rows: P={P}, {bytes_moved/1e6:.0f} MB in+out, {ms*1e3:.1f} us, {bytes_moved/ms/1e6:.0f} GB/s = {bytes_moved/ms/1e6/8000:.2f} of the 8 TB/s HBM peak")
