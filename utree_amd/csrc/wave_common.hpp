// wave_common.hpp -- device helpers shared by the kernel files: XT_getIX32 on the device image,
// wave64 intrinsics wrappers, base coding, k-mer words from the packed LDS stream.  gfx950 / wave64 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_common.hpp"

#ifndef UTREE_RUN_HEAD
#define UTREE_RUN_HEAD 7          /* records of a run fetched together, 8-byte records (<= 8: the image pads 8 records) */
#endif
#ifndef UTREE_RUN_HEAD2
#define UTREE_RUN_HEAD2 3         /* ... 16- and 32-byte records */
#endif

namespace utk {

// ------------------------------------------------------------------------------------------------
// XT_getIX32 (itree.c:720-730) on the device image
// ------------------------------------------------------------------------------------------------
template <typename OFF> __device__ __forceinline__ void coarse_bin(const utk_image &im, uint32_t p, uint64_t &s, uint64_t &e) {
    const OFF *c = (const OFF *)im.coarse;
    s = c[p]; e = c[p + 1];                                      // itree.c:724
}

// exact-match search in a run of MIN records that ascends by key
template <int W, int I> static __device__ uint32_t min_find(const uint64_t *mrecs, uint64_t lo, uint64_t hi, const MinKey<W> &q) {
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        const Entry<W, I> e = load_entry<W, I>(mrecs, mid);
        const MinKey<W> k = mrec_key<W, I>(e);
        if (mkey_lt<W>(k, q)) lo = mid + 1;
        else if (mkey_eq<W>(k, q)) return mrec_rank<W, I>(e);
        else hi = mid;
    }
    return INVALID;
}

// Second half of a lookup, given the table slot of the word's minimizer.  Words whose 24-bit bin is not
// strictly ascending (COMPRESS' first-bin quirk) or any word of a non-monotone table take the reference's own
// probe sequence over the FILE records instead: only that reproduces its answers there.
template <int W, int I, bool EXC, typename OFF>
__device__ __forceinline__ uint32_t resolve_entry(const utk_image &im, const Entry<W, I> &t, const MinKey<W> &mk, uint64_t khi,
                                                  uint64_t klo) {
    if constexpr (EXC) {
        const uint32_t p = word_prefix<W>(khi, klo);
        if ((im.irreg[p >> 5] >> (p & 31)) & 1u) {
            uint64_t s, e;
            coarse_bin<OFF>(im, p, s, e);
            if (s >= e || e > im.n_nodes) return INVALID;        // itree.c:726
            return exact_probe<W, I>(im.recs, s, e, word_suffix<W>(khi, klo));
        }
    }
    const uint32_t flag = mrec_flag<W, I>(t);
    if (flag == 0) return mkey_eq<W>(mrec_key<W, I>(t), mk) ? mrec_rank<W, I>(t) : INVALID;   // the slot's only node
    if (flag == 1) return INVALID;                                                            // empty slot
    const uint64_t d = t.w[RecTraits<W, I>::KW];
    const uint64_t start = d & M40, n = (d >> 40) & 0x3FFFFFull, end = start + n;
    // A run = the k-mers that share this minimizer.  Minimizers are minima, so most nodes sit in runs of 2-8 records (mean
    // 3.5 at 0.28 nodes per slot; 98 % within 8).  The first RUN_HEAD records are fetched together -- contiguous, so the
    // compiler emits 16-byte loads -- and only longer runs go on with a binary search, one round trip per step.  Measured
    // on one box, 4 M x 150 bp reads: 2 records 5.78 ms, 4: 5.25, 6: 4.96, 7: 4.87, 8: 5.08 (r01).
    constexpr int RUN_HEAD = RecTraits<W, I>::EW == 1 ? UTREE_RUN_HEAD : UTREE_RUN_HEAD2;
    Entry<W, I> r[RUN_HEAD];
#pragma unroll
    for (int i = 0; i < RUN_HEAD; ++i) r[i] = load_entry<W, I>(im.mrecs, start + i);
#pragma unroll
    for (int i = 0; i < RUN_HEAD; ++i) {
        if ((uint64_t)i >= n) return INVALID;
        const MinKey<W> k = mrec_key<W, I>(r[i]);
        if (mkey_eq<W>(k, mk)) return mrec_rank<W, I>(r[i]);
        if (mkey_lt<W>(mk, k)) return INVALID;                          // the run ascends by key
    }
    if (n <= (uint64_t)RUN_HEAD) return INVALID;
    return min_find<W, I>(im.mrecs, start + RUN_HEAD, end, mk);
}

template <int W, int I, bool EXC, typename OFF>
__device__ __forceinline__ uint32_t lookup_word(const utk_image &im, uint64_t khi, uint64_t klo) {
    uint64_t slot; MinKey<W> mk;
    min_split<W>(khi, klo, 24 + im.fine_bits, slot, mk);
    const Entry<W, I> t = load_slot<W, I>(im.table, slot);
    return resolve_entry<W, I, EXC, OFF>(im, t, mk, khi, klo);
}

// ------------------------------------------------------------------------------------------------
// wave64 helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t lanes_below(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// wave-uniform values the compiler cannot prove uniform: pin them to scalar registers so that the loops and
// address arithmetic they drive run on the scalar unit instead of as exec-masked vector code
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
    return ((uint64_t)uni32((uint32_t)(v >> 32)) << 32) | uni32((uint32_t)v);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wave execute in issue order; this only stops the compiler from moving a
    // lane's LDS reads above another lane's LDS writes.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// base byte -> 2-bit code and "bad" flag (itree.c:110-121).  A=0 C=1 G=2 T=3, either case.
__device__ __forceinline__ void base_code(uint32_t b, uint32_t &code, bool &bad) {
    uint32_t u = b & 0xDFu;
    bad = !(u == 'A' || u == 'C' || u == 'G' || u == 'T');
    uint32_t g = (b >> 1) & 3u;                        // A:0 C:1 G:3 T:2
    code = g ^ (g >> 1);
}

// k-mer word of the window that starts at base i, from the big-endian packed 2-bit stream in LDS
// (word j holds bases 16j..16j+15, base 16j in the top two bits).  itree.c:924: first base most significant.
template <int W> __device__ __forceinline__ void window_word(const uint32_t *sw, uint32_t i, uint64_t &khi, uint64_t &klo) {
    uint32_t j = i >> 4, sh = 32u - ((i & 15u) << 1);          // sh in [2,32]
    uint64_t a = ((uint64_t)sw[j] << 32) | sw[j + 1];
    uint64_t b = ((uint64_t)sw[j + 1] << 32) | sw[j + 2];
    uint32_t x0 = (uint32_t)(a >> sh), x1 = (uint32_t)(b >> sh);
    if constexpr (W == 16) {
        uint64_t c = ((uint64_t)sw[j + 2] << 32) | sw[j + 3];
        uint64_t d = ((uint64_t)sw[j + 3] << 32) | sw[j + 4];
        uint32_t x2 = (uint32_t)(c >> sh), x3 = (uint32_t)(d >> sh);
        khi = ((uint64_t)x0 << 32) | x1; klo = ((uint64_t)x2 << 32) | x3;
    } else { khi = 0; klo = ((uint64_t)x0 << 32) | x1; }
}

}  // namespace utk
