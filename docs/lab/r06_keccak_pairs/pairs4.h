
#include <immintrin.h>
namespace kp {
__attribute__((target("avx512f,avx512vl"))) static void keccak_f_pairs4(uint64_t a[25]) {
    static const uint64_t RC[24] = C25519_KECCAK_RC;
    alignas(16) static const uint64_t RH[5][3][2] = {{{0, 36}, {3, 41}, {18, 0}}, {{1, 44}, {10, 45}, {2, 0}}, {{62, 6}, {43, 15}, {61, 0}}, {{28, 55}, {25, 21}, {56, 0}}, {{27, 20}, {39, 8}, {14, 0}}};
    __m128i r00 = _mm_set_epi64x((long long)a[5], (long long)a[0]), r01 = _mm_set_epi64x((long long)a[15], (long long)a[10]), r02 = _mm_loadl_epi64((const __m128i *)(a + 20));
    __m128i r10 = _mm_set_epi64x((long long)a[6], (long long)a[1]), r11 = _mm_set_epi64x((long long)a[16], (long long)a[11]), r12 = _mm_loadl_epi64((const __m128i *)(a + 21));
    __m128i r20 = _mm_set_epi64x((long long)a[7], (long long)a[2]), r21 = _mm_set_epi64x((long long)a[17], (long long)a[12]), r22 = _mm_loadl_epi64((const __m128i *)(a + 22));
    __m128i r30 = _mm_set_epi64x((long long)a[8], (long long)a[3]), r31 = _mm_set_epi64x((long long)a[18], (long long)a[13]), r32 = _mm_loadl_epi64((const __m128i *)(a + 23));
    __m128i r40 = _mm_set_epi64x((long long)a[9], (long long)a[4]), r41 = _mm_set_epi64x((long long)a[19], (long long)a[14]), r42 = _mm_loadl_epi64((const __m128i *)(a + 24));
    for (int rnd = 0; rnd < 24; rnd++) {
        __m128i c0 = _mm_xor_si128(_mm_xor_si128(r00, r01), r02); c0 = _mm_xor_si128(c0, _mm_shuffle_epi32(c0, 0x4E));
        __m128i c1 = _mm_xor_si128(_mm_xor_si128(r10, r11), r12); c1 = _mm_xor_si128(c1, _mm_shuffle_epi32(c1, 0x4E));
        __m128i c2 = _mm_xor_si128(_mm_xor_si128(r20, r21), r22); c2 = _mm_xor_si128(c2, _mm_shuffle_epi32(c2, 0x4E));
        __m128i c3 = _mm_xor_si128(_mm_xor_si128(r30, r31), r32); c3 = _mm_xor_si128(c3, _mm_shuffle_epi32(c3, 0x4E));
        __m128i c4 = _mm_xor_si128(_mm_xor_si128(r40, r41), r42); c4 = _mm_xor_si128(c4, _mm_shuffle_epi32(c4, 0x4E));
        const __m128i d0 = _mm_xor_si128(c4, _mm_rol_epi64(c1, 1));
        const __m128i d1 = _mm_xor_si128(c0, _mm_rol_epi64(c2, 1));
        const __m128i d2 = _mm_xor_si128(c1, _mm_rol_epi64(c3, 1));
        const __m128i d3 = _mm_xor_si128(c2, _mm_rol_epi64(c4, 1));
        const __m128i d4 = _mm_xor_si128(c3, _mm_rol_epi64(c0, 1));
        const __m128i t00 = _mm_rolv_epi64(_mm_xor_si128(r00, d0), _mm_load_si128((const __m128i *)RH[0][0]));
        const __m128i t01 = _mm_rolv_epi64(_mm_xor_si128(r01, d0), _mm_load_si128((const __m128i *)RH[0][1]));
        const __m128i t02 = _mm_rolv_epi64(_mm_xor_si128(r02, d0), _mm_load_si128((const __m128i *)RH[0][2]));
        const __m128i t10 = _mm_rolv_epi64(_mm_xor_si128(r10, d1), _mm_load_si128((const __m128i *)RH[1][0]));
        const __m128i t11 = _mm_rolv_epi64(_mm_xor_si128(r11, d1), _mm_load_si128((const __m128i *)RH[1][1]));
        const __m128i t12 = _mm_rolv_epi64(_mm_xor_si128(r12, d1), _mm_load_si128((const __m128i *)RH[1][2]));
        const __m128i t20 = _mm_rolv_epi64(_mm_xor_si128(r20, d2), _mm_load_si128((const __m128i *)RH[2][0]));
        const __m128i t21 = _mm_rolv_epi64(_mm_xor_si128(r21, d2), _mm_load_si128((const __m128i *)RH[2][1]));
        const __m128i t22 = _mm_rolv_epi64(_mm_xor_si128(r22, d2), _mm_load_si128((const __m128i *)RH[2][2]));
        const __m128i t30 = _mm_rolv_epi64(_mm_xor_si128(r30, d3), _mm_load_si128((const __m128i *)RH[3][0]));
        const __m128i t31 = _mm_rolv_epi64(_mm_xor_si128(r31, d3), _mm_load_si128((const __m128i *)RH[3][1]));
        const __m128i t32 = _mm_rolv_epi64(_mm_xor_si128(r32, d3), _mm_load_si128((const __m128i *)RH[3][2]));
        const __m128i t40 = _mm_rolv_epi64(_mm_xor_si128(r40, d4), _mm_load_si128((const __m128i *)RH[4][0]));
        const __m128i t41 = _mm_rolv_epi64(_mm_xor_si128(r41, d4), _mm_load_si128((const __m128i *)RH[4][1]));
        const __m128i t42 = _mm_rolv_epi64(_mm_xor_si128(r42, d4), _mm_load_si128((const __m128i *)RH[4][2]));
        const __m128i n00 = _mm_unpacklo_epi64(t00, t30), n01 = _mm_unpacklo_epi64(t10, t40), n02 = _mm_move_epi64(t20);
        const __m128i n10 = _mm_unpackhi_epi64(t10, t40), n11 = _mm_unpackhi_epi64(t20, t00), n12 = _mm_srli_si128(t30, 8);
        const __m128i n20 = _mm_unpacklo_epi64(t21, t01), n21 = _mm_unpacklo_epi64(t31, t11), n22 = _mm_move_epi64(t41);
        const __m128i n30 = _mm_unpackhi_epi64(t31, t11), n31 = _mm_unpackhi_epi64(t41, t21), n32 = _mm_srli_si128(t01, 8);
        const __m128i n40 = _mm_unpacklo_epi64(t42, t22), n41 = _mm_unpacklo_epi64(t02, t32), n42 = _mm_move_epi64(t12);
        r00 = _mm_xor_si128(n00, _mm_andnot_si128(n10, n20));
        r01 = _mm_xor_si128(n01, _mm_andnot_si128(n11, n21));
        r02 = _mm_xor_si128(n02, _mm_andnot_si128(n12, n22));
        r10 = _mm_xor_si128(n10, _mm_andnot_si128(n20, n30));
        r11 = _mm_xor_si128(n11, _mm_andnot_si128(n21, n31));
        r12 = _mm_xor_si128(n12, _mm_andnot_si128(n22, n32));
        r20 = _mm_xor_si128(n20, _mm_andnot_si128(n30, n40));
        r21 = _mm_xor_si128(n21, _mm_andnot_si128(n31, n41));
        r22 = _mm_xor_si128(n22, _mm_andnot_si128(n32, n42));
        r30 = _mm_xor_si128(n30, _mm_andnot_si128(n40, n00));
        r31 = _mm_xor_si128(n31, _mm_andnot_si128(n41, n01));
        r32 = _mm_xor_si128(n32, _mm_andnot_si128(n42, n02));
        r40 = _mm_xor_si128(n40, _mm_andnot_si128(n00, n10));
        r41 = _mm_xor_si128(n41, _mm_andnot_si128(n01, n11));
        r42 = _mm_xor_si128(n42, _mm_andnot_si128(n02, n12));
        r00 = _mm_xor_si128(r00, _mm_loadl_epi64((const __m128i *)(RC + rnd)));
    }
    a[0] = (uint64_t)_mm_cvtsi128_si64(r00); a[5] = (uint64_t)_mm_extract_epi64(r00, 1); a[10] = (uint64_t)_mm_cvtsi128_si64(r01); a[15] = (uint64_t)_mm_extract_epi64(r01, 1); a[20] = (uint64_t)_mm_cvtsi128_si64(r02);
    a[1] = (uint64_t)_mm_cvtsi128_si64(r10); a[6] = (uint64_t)_mm_extract_epi64(r10, 1); a[11] = (uint64_t)_mm_cvtsi128_si64(r11); a[16] = (uint64_t)_mm_extract_epi64(r11, 1); a[21] = (uint64_t)_mm_cvtsi128_si64(r12);
    a[2] = (uint64_t)_mm_cvtsi128_si64(r20); a[7] = (uint64_t)_mm_extract_epi64(r20, 1); a[12] = (uint64_t)_mm_cvtsi128_si64(r21); a[17] = (uint64_t)_mm_extract_epi64(r21, 1); a[22] = (uint64_t)_mm_cvtsi128_si64(r22);
    a[3] = (uint64_t)_mm_cvtsi128_si64(r30); a[8] = (uint64_t)_mm_extract_epi64(r30, 1); a[13] = (uint64_t)_mm_cvtsi128_si64(r31); a[18] = (uint64_t)_mm_extract_epi64(r31, 1); a[23] = (uint64_t)_mm_cvtsi128_si64(r32);
    a[4] = (uint64_t)_mm_cvtsi128_si64(r40); a[9] = (uint64_t)_mm_extract_epi64(r40, 1); a[14] = (uint64_t)_mm_cvtsi128_si64(r41); a[19] = (uint64_t)_mm_extract_epi64(r41, 1); a[24] = (uint64_t)_mm_cvtsi128_si64(r42);
}
}
