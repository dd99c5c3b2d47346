// Device-side field self-test kernel ("K1"): raw limbs in, canonical bytes out, so that the
// field layer AS COMPILED FOR THE GPU (asm pins, chained carries, bound classes at their extremes) is compared directly
// with big-integer arithmetic -- not only through composite outputs.  Included by kernels.hip (C25519_CHAIN 1) and
// finish.hip (C25519_CHAIN 0): the same source is checked in both carry forms.
//   op 0: fe_mul(a: wide, b: loose)   1: fe_sq(a: loose)   2: fe_invert(fe_carry(a))   3: fe_to_words(a: wide)
//   op 4: fe_pow_p58(fe_carry(a))     5: fe_sub_w(a: loose, b: loose)   6: fe_carry(a: any u32 limbs)
//   op 7: fe_mul(fe_sub(a: tight, b: tight), fe_add(a, b))   (the bound classes as the point formulas chain them)
//   op 8-11: the lockstep multiplier of k_accumulate (fe26x.h), a: wide, b: loose --
//            8: fe_mul_chain_n<3>{(a,b),(b,b),(a,b)}[0] = a b   9: the same, [1] = b^2
//           10: fe_mul_chain_n<4>{(b,b),(b,b),(a,b),(b,b)}[2] = a b   11: the same, [3] = b^2
#pragma once
#include "devio.h"
#include "fe26x.h"

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
#if defined(__HIP_DEVICE_COMPILE__)
    case 8: case 9: {
        feW f[3]; feL g[3]; feT o[3];
        f[0] = x; g[0] = y; f[1] = y; g[1] = y; f[2] = x; g[2] = y;
        fe_mul_chain_n<3>(o, f, g);
        r = op == 8 ? o[0] : o[1];
        break;
    }
    case 10: case 11: {
        feW f[4]; feL g[4]; feT o[4];
        f[0] = y; g[0] = y; f[1] = y; g[1] = y; f[2] = x; g[2] = y; f[3] = y; g[3] = y;
        fe_mul_chain_n<4>(o, f, g);
        r = op == 10 ? o[2] : o[3];
        break;
    }
#endif
    default: r = fe_mul(fe_sub(xt, yt), fe_add(xt, yt)); break;
    }
    u32 w[8];
    fe_to_words(r, w);
    store8(out, i, w);
}

}  // namespace c25519
