// Device-side field self-test kernel ("K1"): raw limbs in, canonical bytes out, so that the
// field layer AS COMPILED FOR THE GPU (asm pins, chained carries, bound classes at their extremes) is compared directly
// with big-integer arithmetic -- not only through composite outputs.  Included by kernels.hip (C25519_CHAIN 1) and
// finish.hip (C25519_CHAIN 0): the same source is checked in both carry forms.
//   op 0: fe_mul(a: wide, b: loose)   1: fe_sq(a: loose)   2: fe_invert(fe_carry(a))   3: fe_to_words(a: wide)
//   op 4: fe_pow_p58(fe_carry(a))     5: fe_sub_w(a: loose, b: loose)   6: fe_carry(a: any u32 limbs)
//   op 7: fe_mul(fe_sub(a: tight, b: tight), fe_add(a, b))   (the bound classes as the point formulas chain them)
#pragma once
#include "devio.h"

namespace c25519 {

template <int CHAIN_TAG>
__global__ void __launch_bounds__(256) k_selftest_field(int op, const u32 *__restrict__ a, const u32 *__restrict__ b, u64 n, uint8_t *__restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    feW x; feL y;
    for (int q = 0; q < 10; q++) { x.v[q] = a[10 * i + q]; y.v[q] = b ? b[10 * i + q] : 0u; }
    feL xl; feT xt, yt;
    for (int q = 0; q < 10; q++) { xl.v[q] = x.v[q]; xt.v[q] = x.v[q]; yt.v[q] = y.v[q]; }
    feW r;
    switch (op) {
    case 0: r = fe_mul(x, y); break;
    case 1: r = fe_sq(xl); break;
    case 2: r = fe_invert(fe_carry(x)); break;
    case 3: r = x; break;
    case 4: r = fe_pow_p58(fe_carry(x)); break;
    case 5: r = fe_sub_w(xl, y); break;
    case 6: r = fe_carry(x); break;
    default: r = fe_mul(fe_sub(xt, yt), fe_add(xt, yt)); break;
    }
    u32 w[8];
    fe_to_words(r, w);
    store8(out, i, w);
}

}  // namespace c25519
