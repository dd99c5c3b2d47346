// Shared pieces of the MSM sort (internal): wave priorities, the window digit, the partition entry, the long-bucket work list and the
// constants the host side needs to size LDS and workspaces.  Included by msm.hip (driver), msm_sort.hip (the chunk-local sort of every
// plain MSM) and msm_sort_matrix.hip (round 2's digit-matrix sort, kept for the merged layout of the precomputed tables).
#pragma once
#include <hip/hip_runtime.h>
#include "devio.h"
#include "msm_internal.h"

// Wave priorities (s_setprio): the issue arbiter of a SIMD takes the highest priority first and the oldest wave within it.
// k_accumulate and the decompression run for a millisecond with every wave always ready to issue, and the short
// latency-bound kernels of the NEXT pass (digits, sort, hash chain) that share the SIMDs with them are younger: at equal
// priority they only get the issue slots the old waves leave.  They are the critical path, so they go first.
#ifndef C25519_PRIO
#define C25519_PRIO 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && C25519_PRIO
#define C25519_PRIO_CHAIN() __builtin_amdgcn_s_setprio(3)
#define C25519_PRIO_SIDE() __builtin_amdgcn_s_setprio(2)
#ifndef C25519_PRIO_LONG_LEVEL
#define C25519_PRIO_LONG_LEVEL 2
#endif
#define C25519_PRIO_LONG() __builtin_amdgcn_s_setprio(C25519_PRIO_LONG_LEVEL)
#else
#define C25519_PRIO_LONG() do { } while (0)
#define C25519_PRIO_CHAIN() do { } while (0)
#define C25519_PRIO_SIDE() do { } while (0)
#endif

namespace c25519 {

// signed digit of window k from the stored value
// (a top digit above `half` can only come from a scalar with bit 255 set: k_digits has flagged it and the call fails;
// it is dropped here so that no kernel indexes past its tables)
__device__ __forceinline__ int digit_of(u32 v, int k, const msm_geom &g) {
    return (k >= g.first_unsigned) ? (v <= (u32)g.half ? (int)v : 0) : (int)v - (1 << (g.wid[k] - 1));     // msm_layout: the top two windows are unsigned
}

constexpr int PART_BPS_MAX = 512;        // buckets per slice: 2^g.bps_log2 <= this, chosen so that a bin holds ~16 K entries (512: windows of 17 bits)
#ifndef C25519_PART_CAP
#define C25519_PART_CAP 17408
#endif
constexpr int PART_CAP = C25519_PART_CAP;          // bin capacity of the LDS path of pass 2 (mean <= 16384, sigma 128; larger bins take the global path)
// terms per pass-1 block: the staging buffer (4 bytes per term) plus 18 counters per slice must leave room for two blocks
// per CU (2 x 80 KB of the 160 KB LDS)
static inline int part_chunk(int SL) { return SL <= 128 ? 16384 : 15360; }

// intermediate entry of the two-level sorts: bucket within the slice << sh | sign << (sh - 1) | term index, sh = 24 for slices of up to 256 buckets
// (23-bit term index) and 23 for 512 (22-bit index: passes of at most 2^22 terms, msm_sort_params)
__host__ __device__ __forceinline__ int part_entry_shift(const msm_geom &g) { return g.bps_log2 > 8 ? 32 - g.bps_log2 : 24; }
// bps: log2 of the buckets per slice of THIS window (g.bps[k] in the chunk-local sort, g.bps_log2 in the digit-matrix sort)
__device__ __forceinline__ bool part_entry(u32 v, int k, const msm_geom &g, int bps, u32 t, u32 &slice, u32 &entry) {
    int d = digit_of(v, k, g);
    if (d == 0) return false;
    u32 b = (u32)((d > 0 ? d : -d) - 1);
    const int sh = part_entry_shift(g);
    slice = b >> bps;
    entry = ((b & ((1u << bps) - 1u)) << sh) | (d < 0 ? (1u << (sh - 1)) : 0u) | t;
    return true;
}
// sorted-list entry (what k_accumulate reads): term index | sign << 31
__device__ __forceinline__ u32 part_entry_final(u32 ev, int sh) { return (ev & ((1u << (sh - 1)) - 1u)) | ((ev >> (sh - 1)) << 31); }

constexpr int SWEEP_TPT = 8, SWEEP_THREADS = 1024, SWEEP_WAVES = SWEEP_THREADS / 64, SWEEP_CHUNK = SWEEP_THREADS * SWEEP_TPT;
constexpr int P2G_ITER = 48;

// a long bucket's list is cut into segments of LONG_SEG entries: one work item each (k_long_segments)
struct long_item { u32 gid, lo, hi, first; };
// what the bucket order needs from one bucket with c entries: its length class (a 256-bin block-local histogram) and, for a
// list beyond the cap, its long-bucket work items
__device__ __forceinline__ void order_note_bucket(u32 c, u64 G, const msm_geom &g, const u32 *__restrict__ base, u32 *h, u32 max_items, long_item *__restrict__ items,
                                                  u32 *__restrict__ counters, u32 *__restrict__ long_gids, u32 *__restrict__ long_first) {
    atomicAdd(&h[255u - (c > 255u ? 255u : c)], 1u);
    if (c > g.long_cap) {
        const int k = (int)(G / g.half), b = (int)(G % g.half);
        const u32 lo = base[(u64)k * (g.half + 1) + b], hi = lo + c;
        const u32 nseg = (c + LONG_SEG - 1) / LONG_SEG;
        const u32 first = atomicAdd(&counters[0], nseg);
        const u32 lb = atomicAdd(&counters[1], 1u);
        long_gids[lb] = (u32)G;
        long_first[lb] = first;
        // number of segments of this bucket is recomputed by the combiner from base[]
        for (u32 sg = 0; sg < nseg && first + sg < max_items; sg++) {
            long_item it; it.gid = (u32)G; it.lo = lo + sg * LONG_SEG; it.hi = (it.lo + LONG_SEG < hi) ? it.lo + LONG_SEG : hi; it.first = first;
            items[first + sg] = it;
        }
    }
}

}  // namespace c25519
