"""The C ABI driven from plain C: tests/host/abi_c_smoke.c is compiled with gcc -std=c11 against
include/c25519_hip.h, linked to the in-tree libc25519hip.so and run on the GPU (RFC 7748 6.1, 1*B, a small MSM, the
NONE status).  The compile step alone also runs on CPU (test_abi_header_is_valid_c) -- the header must stay C."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "abi_c_smoke.c")
LIBDIR = os.path.join(ROOT, "curve25519-dalek_amd", "lib")


def test_abi_header_is_valid_c():
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", SRC])


@pytest.mark.gpu
def test_abi_from_plain_c(tmp_path):
    exe = str(tmp_path / "abi_c_smoke")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-o", exe, SRC, "-L" + LIBDIR, "-lc25519hip", "-Wl,-rpath," + LIBDIR,
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "abi_c_smoke ok" in out.stdout
