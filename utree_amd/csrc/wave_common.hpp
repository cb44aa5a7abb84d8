// wave_common.hpp -- device helpers shared by the kernel files: XT_getIX32 on the device image,
// wave64 intrinsics wrappers, base coding, k-mer words from the packed LDS stream.  gfx950 / wave64 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_common.hpp"

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

// A bucket = 64 or 128 bytes (image.bucket_words) of entries, ascending by key and filled from entry 0, unused ones flagged empty; when
// more nodes fall into a bucket its LAST entry is an overflow descriptor instead.  The wave-per-read kernels take 64 bytes at a time (16
// registers, what they have room for): of a 128-byte bucket the lower half first, the upper half -- the same line: an L1 / L2 hit -- only
// when the lower half is full and did not hold the key.  (The lane-per-read pass, lanes_kernel.hip, fetches a whole bucket with a quad
// of lanes.)
template <int W, int I> struct BucketOf { static constexpr int CAP = RecTraits<W, I>::HCAP; };   // entries of a half
template <int W, int I> struct Bucket { Entry<W, I> e[BucketOf<W, I>::CAP]; };

// first 64 bytes of bucket `bucket` of the table
template <int W, int I> __device__ __forceinline__ Bucket<W, I> load_bucket(const uint64_t *__restrict__ table, uint64_t bucket, uint32_t bucket_words) {
    Bucket<W, I> b;
    constexpr int HCAP = BucketOf<W, I>::CAP, EW = RecTraits<W, I>::EW;
#pragma unroll
    for (int i = 0; i < HCAP; ++i) b.e[i] = load_slot<W, I>(table, bucket * (bucket_words / EW) + i);   // contiguous: 16-byte non-temporal loads
    return b;
}

// The 64 bytes at a byte address computed from the LDS region table: the address space is stated (global), or the loads would be
// flat ones.  Default cache policy on purpose: the four 16-byte loads of a lane -- and those of the ~9 lanes whose windows share
// the bucket -- then merge in the L1's miss queue into one request to the L2; with the non-temporal policy (which serves single
// 8-byte random reads best, load_slot) every one of the four instructions went to the L2 on its own (TCC_HIT 3x TCC_MISS) and the
// kernel was 12 % slower (same-box, round 2).
template <int W, int I> __device__ __forceinline__ Bucket<W, I> load_bucket_at(uint64_t addr) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(1))) u64x2 *gptr;
    const gptr p = (gptr)addr;
    Bucket<W, I> b;
    constexpr int EW = RecTraits<W, I>::EW, HCAP = BucketOf<W, I>::CAP;
    u64x2 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = p[q];
#pragma unroll
    for (int i = 0; i < HCAP; ++i)
#pragma unroll
        for (int x = 0; x < EW; ++x) { const int w = i * EW + x; b.e[i].w[x] = (w & 1) ? v[w >> 1].y : v[w >> 1].x; }
    return b;
}

// The 8 entries of half a k = 32 / u16-label bucket against (tag = hash low bits << 5 | minimizer position, rest):
// an entry's low word is its rest, its high word {flag2 | tag13 | 0 | rank16}.  One compare and one select per entry pick the
// high word of the entry whose rest matches; the tag (and with it the flag: 0 = a record) is checked once.  Two entries with
// the same rest and different tags are possible (the same 16 outer bases around a minimizer at two positions): then, and when
// an empty entry or the overflow descriptor happens to carry the rest's bit pattern, every entry is looked at in full.
__device__ __forceinline__ uint32_t scan_bucket82(const Bucket<8, 2> &b, uint32_t tag, uint32_t rest) {
    uint32_t sel = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < 8; ++i) sel = (uint32_t)b.e[i].w[0] == rest ? (uint32_t)(b.e[i].w[0] >> 32) : sel;
    uint32_t rank = INVALID;
    if ((sel >> 17) == tag) rank = sel & 0xFFFFu;
    else if (sel != 0xFFFFFFFFu) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if ((uint32_t)b.e[i].w[0] == rest && (uint32_t)(b.e[i].w[0] >> 49) == tag) rank = (uint32_t)(b.e[i].w[0] >> 32) & 0xFFFFu;
    }
    return rank == 0xFFFFu ? INVALID : rank;
}

// the rank a half holds for the key, or INVALID
template <int W, int I> __device__ __forceinline__ uint32_t scan_half(const Bucket<W, I> &b, const MinKey<W> &mk) {
    constexpr int HCAP = BucketOf<W, I>::CAP;
    if constexpr (W == 8 && I == 2) return scan_bucket82(b, (uint32_t)(mk.lo >> 32), (uint32_t)mk.lo);
    else {
        // straight-line scan.  An entry is the record of `mk` exactly when its key word without the 16 label bits equals
        // {flag 0 | mk} (an empty entry or the overflow descriptor has a non-zero flag, so neither can): one AND and one
        // 64-bit compare per entry; the label bits of the matching entry are picked up raw and decoded once at the end.
        constexpr int KW = RecTraits<W, I>::KW;
        const uint64_t want = W == 16 ? mk.hi << 16 : (((mk.lo >> 32) << 49) | (mk.lo & 0xFFFFFFFFull));
        uint32_t raw = INVALID;
#pragma unroll
        for (int i = 0; i < HCAP; ++i) {
            bool hit;
            if constexpr (W == 16) hit = (b.e[i].w[KW] & ~0xFFFFull) == want && b.e[i].w[0] == mk.lo;
            else hit = b.e[i].w[0] == want;                                              // (k = 32, u32 labels: the rank16 field is 0)
            raw = hit ? (uint32_t)b.e[i].w[I == 4 ? KW + 1 : KW] : raw;
        }
        uint32_t rank = raw;
        if constexpr (I == 2) { rank = raw & 0xFFFFu; rank = rank == 0xFFFFu ? INVALID : rank; }
        return rank;
    }
}

// Second half of a lookup, given the LOWER HALF of the bucket of the word's minimizer and the bucket's address.  Words whose 24-bit
// bin is not strictly ascending (COMPRESS' first-bin quirk) or any word of a non-monotone table take the reference's own probe
// sequence over the FILE records instead: only that reproduces its answers there.
template <int W, int I, bool EXC, typename OFF>
__device__ __forceinline__ uint32_t resolve_bucket(const utk_image &im, const Bucket<W, I> &b, uint64_t baddr, const MinKey<W> &mk, uint64_t khi,
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
    constexpr int HCAP = BucketOf<W, I>::CAP;
    uint32_t rank = scan_half<W, I>(b, mk);
    uint64_t last = b.e[HCAP - 1].w[RecTraits<W, I>::KW];                                 // the key word of the last entry looked at
    if (im.bucket_words == 16u && rank == INVALID && (last >> 62) != 1) {                 // a 128-byte bucket whose lower half is full: the upper one
        const Bucket<W, I> u = load_bucket_at<W, I>(baddr + 64);
        rank = scan_half<W, I>(u, mk);
        last = u.e[HCAP - 1].w[RecTraits<W, I>::KW];
    } else if (im.bucket_words == 16u) last = 0;                                           // (the lower half's last entry is never a descriptor)
    if ((last >> 62) == 2 && rank == INVALID) {                                            // the rest of the bucket's nodes
        uint64_t start = ovf_first<W, I>(last), n = ovf_count(last);
        if constexpr (W == 8) {
            if ((last & OVF_HAS_DIR) && (im.flags & UTREE_F_OVF_CHAINS))                  // a heavy run stored as chains
                return chain_find<I>(im.mrecs + (last & M39) * RecTraits<W, I>::EW, (uint32_t)n, (uint32_t)(mk.lo >> 32) & 31u, (uint32_t)mk.lo);
        }
        if (last & OVF_HAS_DIR) {                                                         // a heavy run: only the records of the key's own position
            const uint16_t *dir = (const uint16_t *)(im.mrecs + (last & M39) * RecTraits<W, I>::EW);
            const uint32_t pos = W == 16 ? (uint32_t)(mk.hi >> 32) & 63u : (uint32_t)(mk.lo >> 32) & 31u;
            const uint32_t a = dir[pos], b = dir[pos + 1];
            n = b - a; start += a;
        }
        rank = min_find<W, I>(im.mrecs, start, start + n, mk);
    }
    return rank;
}

// The same for k = 32 from the window loop's pieces.
template <int I, bool EXC, typename OFF>
__device__ __forceinline__ uint32_t resolve_bucket8(const utk_image &im, const Bucket<8, I> &b, uint64_t baddr, uint32_t hlow, uint32_t pos, uint32_t rest,
                                                    uint32_t x0, uint32_t x1) {
    MinKey<8> mk; mk.hi = 0; mk.lo = ((uint64_t)((hlow << 5) | pos) << 32) | rest;
    return resolve_bucket<8, I, EXC, OFF>(im, b, baddr, mk, 0ull, ((uint64_t)x0 << 32) | x1);
}

// PACKSIZE=16 (W = 4): a k-mer is one 32-bit word, and the image holds the answer of XT_getIX32 for every one of the 2^32 words -- the
// label's rank, or all ones -- in a direct-address table (built at load time with the reference's own probe order where a bin is not
// strictly ascending): one 2- or 4-byte load per window, nothing to search
template <int I> __device__ __forceinline__ uint32_t direct_rank(const utk_image &im, uint32_t word) {
    if constexpr (I == 2) { const uint32_t r = ((const uint16_t *)im.table)[word]; return r == 0xFFFFu ? INVALID : r; }
    else return ((const uint32_t *)im.table)[word];
}

template <int W, int I, bool EXC, typename OFF>
__device__ __forceinline__ uint32_t lookup_word(const utk_image &im, uint64_t khi, uint64_t klo) {
    if constexpr (W == 4) return direct_rank<I>(im, (uint32_t)klo);
    uint64_t bucket; MinKey<W> mk;
    min_split<W>(khi, klo, im.regions, bucket, mk);
    const Bucket<W, I> b = load_bucket<W, I>(im.table, bucket, im.bucket_words);
    return resolve_bucket<W, I, EXC, OFF>(im, b, (uint64_t)(uintptr_t)im.table + bucket * (8u * im.bucket_words), mk, khi, klo);
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
