// Tuning knobs of the library (A/B arms, profiling switches, pass-size overrides).
#pragma once
#include <stdlib.h>
// ---- tuning knobs ------------------------------------------------------------------------------------------------------------
// The RELEASE library (lib/libc25519hip.so) reads NOTHING from the environment: C25519_KNOB(name, default) is the default, a compile-time
// constant, and no "C25519_..." string is left in the binary (tests/test_abi_cpu.py asserts it on the built file).  The A/B arms, the profiling
// switches and the pass-size overrides the tests use to run many small passes exist only in the TUNING build (make tune ->
// lib/libc25519hip_tune.so, -DC25519_TUNING: the same sources, every knob read once per process from C25519_<name>); the tests and tools that
// need a knob point C25519_HIP_LIB (a Python-side variable of engine.py) at that file.
#ifdef C25519_TUNING
static inline long long c25519_knob_env(const char *name, long long dflt) { const char *e = getenv(name); return e ? atoll(e) : dflt; }
#define C25519_KNOB(name, dflt) ((int)c25519_knob_env("C25519_" name, (dflt)))
#define C25519_KNOB_LL(name, dflt) c25519_knob_env("C25519_" name, (dflt))
#else
#define C25519_KNOB(name, dflt) (dflt)
#define C25519_KNOB_LL(name, dflt) ((long long)(dflt))
#endif
