#!/usr/bin/env python3
"""Extract the reference's own golden vectors / known-answer constants into JSON fixtures.

Run in the build container (reads /root/reference, which does NOT exist on the GPU box):
    python tests/golden/extract_vectors.py
Writes tests/golden/{rust_arrays.json, ed25519_testvectors.txt, ed25519_validation.json}.

rust_arrays.json: every numeric array literal found in the listed Rust files, keyed
  "<file>" -> "<enclosing fn or ->:<nearest preceding static/const/let name>:<ordinal>" -> [ints]
so that tests can address e.g. field.rs "-:A_BYTES:0" or x25519_tests.rs
"rfc7748_ladder_test1_vectorset1:expected:0".  Only data (byte/limb arrays) is extracted, no code.
"""
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

FILES = [
    "curve25519-dalek/src/field.rs",
    "curve25519-dalek/src/edwards.rs",
    "curve25519-dalek/src/scalar.rs",
    "curve25519-dalek/src/ristretto.rs",
    "curve25519-dalek/src/montgomery.rs",
    "curve25519-dalek/src/constants.rs",
    "curve25519-dalek/src/backend/serial/u64/constants.rs",
    "curve25519-dalek/src/backend/serial/u64/scalar.rs",
    "x25519-dalek/tests/x25519_tests.rs",
    "ed25519-dalek/tests/ed25519.rs",
]

NUM = r"-?(?:0x[0-9a-fA-F_]+|\d[\d_]*)(?:u8|u64|i8|u32|usize)?"
ARR = re.compile(r"\[\s*(?:" + NUM + r"\s*,\s*)+(?:" + NUM + r")?\s*\]")
DECL = re.compile(r"\b(?:static|const|let(?:\s+mut)?)\s+(\w+)")
FN = re.compile(r"\bfn\s+(\w+)")
MAX_PER_NAME = 48


def parse_num(tok):
    tok = re.sub(r"(u8|u64|i8|u32|usize)$", "", tok).replace("_", "")
    neg = tok.startswith("-")
    tok = tok.lstrip("-")
    v = int(tok, 16) if tok.lower().startswith("0x") else int(tok, 10)
    return -v if neg else v


def extract(path):
    src = open(path).read()
    src = re.sub(r"//[^\n]*", lambda m: " " * len(m.group(0)), src)  # strip comments, keep offsets
    decls = [(m.start(), m.group(1)) for m in DECL.finditer(src)]
    fns = [(m.start(), m.group(1)) for m in FN.finditer(src)]
    out, counts = {}, {}
    for m in ARR.finditer(src):
        nums = [parse_num(t) for t in re.findall(NUM, m.group(0))]
        if len(nums) < 4:
            continue
        pos = m.start()
        fn = "-"
        for p, n in fns:
            if p < pos:
                fn = n
            else:
                break
        name = "?"
        for p, n in decls:
            if p < pos:
                name = n
            else:
                break
        # module-level statics are not "inside" the preceding fn: detect by indentation-free decl
        key0 = "%s:%s" % (fn, name)
        k = counts.get(key0, 0)
        counts[key0] = k + 1
        if k >= MAX_PER_NAME:
            continue
        out["%s:%d" % (key0, k)] = nums
    return out


def main():
    res = {}
    for f in FILES:
        res[f] = extract(os.path.join(REF, f))
    with open(os.path.join(HERE, "rust_arrays.json"), "w") as fh:
        json.dump(res, fh, separators=(",", ":"), sort_keys=True)
    # Ed25519 sign/verify triples (ed25519-dalek/TESTVECTORS, from ed25519.cr.yp.to sign.input)
    with open(os.path.join(REF, "ed25519-dalek/TESTVECTORS")) as fh, \
            open(os.path.join(HERE, "ed25519_testvectors.txt"), "w") as out:
        out.write(fh.read())
    # C2SP/CCTV edge-case vectors (ed25519-dalek/VALIDATIONVECTORS), compacted
    with open(os.path.join(REF, "ed25519-dalek/VALIDATIONVECTORS")) as fh:
        vv = json.load(fh)
    compact = [{"number": v["number"], "key": v["key"], "sig": v["sig"], "msg": v["msg"],
                "flags": v.get("flags") or []} for v in vv]
    with open(os.path.join(HERE, "ed25519_validation.json"), "w") as fh:
        json.dump(compact, fh, separators=(",", ":"))
    print({k: len(v) for k, v in res.items()}, len(compact))


if __name__ == "__main__":
    main()
