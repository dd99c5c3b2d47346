"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (oracle/Makefile: liboracle_san.so): the known-answer
tests and the host transcript / restatement checks run once more against the instrumented build, in a process that preloads
libasan.  The oracle is what every parity claim rests on; a stray out-of-bounds table or digit access in it must not pass
as "bit-exact"."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")


def _libasan():
    gcc = shutil.which("gcc")
    if not gcc:
        return None
    p = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_libasan() is None, reason="gcc / libasan not available")
def test_oracle_known_answers_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", ORACLE, "liboracle_san.so"])
    env = dict(os.environ)
    env["ORACLE_LIB"] = os.path.join(ORACLE, "liboracle_san.so")
    env["LD_PRELOAD"] = _libasan()
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_kat.py")],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.skipif(_libasan() is None or shutil.which("g++") is None, reason="g++ / libasan not available")
def test_host_transcript_under_asan_ubsan(tmp_path):
    """csrc/transcript_host.h (product host code: the strict z-mode's sponge) instrumented: every Keccak-f form the host has on an exactly 200-byte heap
    state, transcripts of 0 ... 1000 signatures on exactly-sized heap buffers (tests/host/transcript_san.cpp)."""
    exe = str(tmp_path / "transcript_san")
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=c++17",
                           "-I", os.path.join(ROOT, "curve25519-dalek_amd", "csrc"), os.path.join(ROOT, "tests", "host", "transcript_san.cpp"), "-o", exe])
    env = dict(os.environ); env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1"; env["UBSAN_OPTIONS"] = "halt_on_error=1"
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "sanitized ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
