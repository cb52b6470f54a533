"""The drop-in boundary is a C ABI: a C99 translation unit includes include/kirag_amd.h (gcc -pedantic, no C++), binds every entry point with
dlsym and runs the host-only ones (tests/capi/capi_smoke.c)."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_header_is_plain_c_and_library_binds_from_c(tmp_path):
    exe = str(tmp_path / "capi_smoke")
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(REPO, "include"),
                           os.path.join(REPO, "tests", "capi", "capi_smoke.c"), "-o", exe, "-ldl", "-lm"])
    # strict ISO C check of the header alone
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(REPO, "include", "kirag_amd.h")])
    out = subprocess.run([exe, os.path.join(REPO, "kirag_amd", "libkirag_amd.so")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "capi_smoke ok" in out.stdout


def test_every_declared_symbol_is_bound_by_the_c_consumer_and_by_ctypes():
    """include/kirag_amd.h is the boundary: every function it declares must be bound by the plain-C consumer (dlsym in capi_smoke.c) and by the
    ctypes table the Python shims use (the C program covered 24 of the 40 exports of ABI 4)."""
    import re
    import sys
    sys.path.insert(0, REPO)
    from kirag_amd import _lib
    header = open(os.path.join(REPO, "include", "kirag_amd.h")).read()
    declared = re.findall(r"^(?:const\s+)?[A-Za-z_0-9]+\*?\s+\*?(kr_[a-z0-9_]+)\s*\(", header, re.M)
    assert len(declared) >= 42 and len(set(declared)) == len(declared)
    csrc = open(os.path.join(REPO, "tests", "capi", "capi_smoke.c")).read()
    bound = set(re.findall(r"BIND\((kr_[a-z0-9_]+)\)", csrc))
    assert set(declared) <= bound, sorted(set(declared) - bound)
    assert set(declared) == set(_lib.SIGNATURES), sorted(set(declared) ^ set(_lib.SIGNATURES))
