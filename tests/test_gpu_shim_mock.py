"""A compiled stand-in for the Rust shim (no Rust toolchain in this image): tests/host/shim_mock.cpp holds the
reference's in-memory types (FieldElement51([u64; 5]), EdwardsPoint{X, Y, Z, T}), copies them limb by limb into the raw160 layout,
maps statuses to Option / panic, dispatches on a size threshold as edwards.rs:1025 does, and runs the reference's own consistency
shapes through it (edwards.rs:2276-2335 multiscalar_consistency_n_{100,250,500,1000}, :2364-2411 precomputed vs plain,
:2084-2129 mul_base_clamped vs mul_clamped with a basepoint table of a point with torsion).  The compile step alone also runs on CPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "shim_mock.cpp")
LIBDIR = os.path.join(ROOT, "curve25519-dalek_amd", "lib")
ORCDIR = os.path.join(ROOT, "oracle")


def test_shim_mock_compiles():
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", SRC])


@pytest.mark.gpu
def test_reference_consistency_shapes_through_the_shim(tmp_path):
    from oracle import orc
    orc.lib()                                              # builds oracle/liboracle.so if it is missing (the mock's "serial backend")
    exe = str(tmp_path / "shim_mock")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, SRC, "-L" + LIBDIR, "-lc25519hip", "-Wl,-rpath," + LIBDIR,
                           "-L" + ORCDIR, "-l:liboracle.so", "-Wl,-rpath," + ORCDIR, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64"])
    out = subprocess.run([exe, "3"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "shim_mock ok" in out.stdout
