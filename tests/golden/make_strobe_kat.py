#!/usr/bin/env python3
"""Makes tests/golden/strobe_kat.json from the independent STROBE / Merlin reference in tests/pyref.py (general-form
implementation of the public STROBE v1.0.2 and FIPS 202 specifications; validated against hashlib's SHA3-256 / SHAKE128 and
Merlin's published conformance vector before anything is written).

The reference (ed25519-dalek/src/batch/transcript.rs, over the un-vendored strobe-rs 0.13.0 / keccak 0.2.0) holds no
byte-level vector for its transcript, its RNG finalisation or any z_i, so these vectors are what pins
 (a) the STROBE operations meta_ad / ad / prf / KEY, continuations (`more`) and rate-boundary crossings (op scripts), and
 (b) the z_i of verify_batch (batch.rs:168-222: hram / sig.s framing, "rng" + KEY(32 zero bytes), one PRF of 16 bytes per z)
for the oracle (orc_strobe_script, orc_batch_transcript_zs), the library's host transcript (ed25519_batch_transcript_zs)
and the engine (c25519_debug_batch_zs).  Run from the repository root: python tests/golden/make_strobe_kat.py
"""
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import pyref  # noqa: E402


def run_script(proto, ops):
    s = pyref.Strobe128(proto)
    S = pyref.Strobe128
    flags = {"meta_ad": S.M | S.A, "ad": S.A, "prf": S.I | S.A | S.C, "key": S.A | S.C}
    out = b""
    for name, more, payload in ops:
        r = s.operate(flags[name], payload, more=more)
        if name == "prf":
            out += r
    return out


def main():
    for n in (0, 1, 135, 136, 137, 500):
        m = (bytes(range(256)) * 2)[:n]
        assert pyref.sponge(136, 0x06, m, 32) == hashlib.sha3_256(m).digest()
        assert pyref.sponge(168, 0x1F, m, 400) == hashlib.shake_128(m).digest(400)
    t = pyref.MerlinTranscript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"

    rng = random.Random(0xC25519)
    rb = lambda n: bytes(rng.getrandbits(8) for _ in range(n))
    scripts = []
    # hand-made shapes: KEY with non-zero data, KEY / PRF across the rate boundary (R = 166), continuations, empty payloads
    shapes = [
        [("key", False, rb(32)), ("prf", False, 16)],
        [("meta_ad", False, b"rng"), ("key", False, bytes(32)), ("meta_ad", False, (16).to_bytes(4, "little")), ("prf", False, 16)],
        [("ad", False, rb(150)), ("key", False, rb(40)), ("prf", False, 200), ("key", False, rb(170)), ("prf", False, 1)],
        [("meta_ad", False, b"w"), ("meta_ad", True, (5).to_bytes(4, "little")), ("key", False, rb(5)), ("key", True, rb(161)), ("key", True, rb(166)), ("prf", False, 64), ("prf", True, 400)],
        [("ad", False, rb(164)), ("prf", False, 8)],                      # the op header itself lands on the boundary
        [("ad", False, rb(163)), ("key", False, rb(1)), ("prf", False, 3)],
        [("ad", False, b""), ("key", False, b""), ("prf", False, 0), ("prf", False, 32)],
        [("key", False, rb(166 * 3)), ("prf", False, 166 * 2 + 1)],
    ]
    for ops in shapes:
        scripts.append(ops)
    for _ in range(24):                                                    # random scripts
        ops, last = [], None
        for _ in range(rng.randrange(3, 14)):
            name = rng.choice(["meta_ad", "ad", "prf", "key", "key"])
            more = (name == last) and rng.random() < 0.3
            n = rng.choice([0, 1, 2, 16, 31, 32, 64, 165, 166, 167, 333])
            ops.append((name, more, n if name == "prf" else rb(n)))
            last = name
        ops.append(("prf", False, 32))
        scripts.append(ops)
    out_scripts = []
    for ops in scripts:
        proto = rng.choice([b"Merlin v1.0", b"c25519-hip test", b""])
        out_scripts.append({"proto": proto.hex(),
                            "ops": [[name, bool(more), (payload if name == "prf" else payload.hex())] for name, more, payload in ops],
                            "out": run_script(proto, ops).hex()})
    zs = []
    for n in (0, 1, 2, 3, 4, 7, 33, 150):
        hr = [rb(64) for _ in range(n)]
        ss = [rb(32) for _ in range(n)]
        zs.append({"hram": [h.hex() for h in hr], "s": [s.hex() for s in ss], "z": [z.hex() for z in pyref.batch_transcript_zs(hr, ss)]})
    with open(os.path.join(HERE, "strobe_kat.json"), "w") as fh:
        json.dump({"scripts": out_scripts, "batch_zs": zs}, fh, indent=0)
    print("wrote %d scripts, %d z batches" % (len(out_scripts), len(zs)))


if __name__ == "__main__":
    main()
