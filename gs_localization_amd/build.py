"""Builds libgsr_hip.so (the C-ABI library of include/gsr.h) for gfx950 with hipcc, in-tree."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
SRC = os.path.join(_HERE, "csrc", "gsr_api.hip")
DEPS = [SRC] + [os.path.join(_HERE, "csrc", f) for f in ("gsr_kernels.h", "gsr_device.h", "gsr_gradmask.h")] + [os.path.join(_ROOT, "include", "gsr.h")]
# GSR_LIB_PATH: diagnostic builds (GSR_TIMING, GSR_DEFS=...) go to a path of their own and are loaded from there (_lib.py reads the
# same variable), so that an experiment never leaves a non-default library where the tests and bench.py look for the product build.
OUT = os.environ.get("GSR_LIB_PATH") or os.path.join(_HERE, "libgsr_hip.so")


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def build(force=False, verbose=False):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           # one lane already holds the wave total when we issue an atomic: keep hipcc from wrapping it in
           # its own (iterative) cross-lane reduction loop
           "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
           # the SLP vectorizer pairs neighbouring list entries of the unrolled compositing bodies on the packed-fp32
           # instructions; building the register pairs costs more v_mov than the packed maths saves (K6 55 -> 50 us)
           "-fno-slp-vectorize",
           "-I" + os.path.join(_ROOT, "include"), "-o", OUT, SRC]
    if (os.environ.get("GSR_TIMING") or os.environ.get("GSR_DEFS")) and not os.environ.get("GSR_LIB_PATH") and not os.environ.get("GSR_ALLOW_INPLACE_VARIANT"):
        raise RuntimeError("a diagnostic build (GSR_TIMING / GSR_DEFS) must not overwrite the product library: set GSR_LIB_PATH=<other file>")
    if os.environ.get("GSR_TIMING"):      # diagnostic build: per-phase clocks inside the compositing kernels
        cmd.insert(1, "-DGSR_TIMING=1")
    if os.environ.get("GSR_DEFS"):        # experiments: extra -D switches, e.g. GSR_DEFS="-DGSR_K8_SPAN=384"
        for d in reversed(os.environ["GSR_DEFS"].split()):
            cmd.insert(1, d)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


EXT_SRC = os.path.join(_HERE, "csrc", "gsrcall.c")
EXT_OUT = os.path.join(_HERE, "_gsrcall.so")


def build_ext(force=False, verbose=False):
    """_gsrcall: the CPython hop of the drop-in packages into the C ABI (csrc/gsrcall.c): plain gcc, no torch headers, links the
    library next to it.  Only for the product library (a diagnostic build under GSR_LIB_PATH keeps the ctypes route)."""
    import sysconfig
    lib = os.path.join(_HERE, "libgsr_hip.so")
    deps = [EXT_SRC, os.path.join(_ROOT, "include", "gsr.h")]
    if not force and os.path.exists(EXT_OUT) and all(os.path.getmtime(EXT_OUT) >= os.path.getmtime(d) for d in deps):
        return EXT_OUT
    if not os.path.exists(lib):
        raise RuntimeError("build libgsr_hip.so first")
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        raise RuntimeError("gcc not found")
    cmd = [cc, "-O2", "-std=c99", "-Wall", "-shared", "-fPIC", "-I" + os.path.join(_ROOT, "include"), "-I" + sysconfig.get_paths()["include"],
           EXT_SRC, "-o", EXT_OUT, "-L" + _HERE, "-l:libgsr_hip.so", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return EXT_OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    if not os.environ.get("GSR_LIB_PATH"):
        print(build_ext(force=True, verbose=True))
