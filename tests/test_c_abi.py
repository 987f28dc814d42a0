"""The C ABI from plain C (no torch, no Python in the loop): include/gsr.h must be a valid C header, and a C program linked
against libgsr_hip.so renders and differentiates a three-splat scene (tests/c_abi/smoke.c)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "smoke.c")


def test_header_is_plain_c():
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "gsr.h")])


def _build(tmp_path):
    from gs_localization_amd import build as B
    B.build()
    exe = str(tmp_path / "c_abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), SRC,
                           "-L" + os.path.join(ROOT, "gs_localization_amd"), "-lgsr_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "gs_localization_amd") + ":/opt/rocm/lib", "-o", exe])
    return exe


def test_c_consumer_compiles_and_links(tmp_path):
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_c_consumer_renders_and_differentiates(tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "c abi smoke ok" in r.stdout, (r.stdout, r.stderr)
