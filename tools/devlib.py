"""Development hook of the tools (NOT part of the product): run a tool against another build of the library -- the tuning build with its C25519_* A/B knobs, the
bound-checking debug build, a variant made by tools/build_variant.sh -- named by the variable C25519_HIP_LIB of the TOOL's environment.  The package itself reads
no environment (tests/test_abi_cpu.py); a tool asks for the other build with an explicit select_library() call, which is what apply() does."""
import os


def apply(pkg):
    p = os.environ.get("C25519_HIP_LIB")
    if p:
        pkg.select_library(p)
    return pkg
