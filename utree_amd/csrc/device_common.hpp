// device_common.hpp -- device-side formats shared by kernels.hip (search) and image_build.hip (load time).
//
// Two record formats live in the device image:
//
//  FILE records  -- the node dump as the .ctr orders it (ascending inside each 24-bit-prefix bin), one per
//                   node, EW 8-byte words: {flag8=0 | rank16 | suffix40} (k=32, u16 labels) ... used by the
//                   load-time kernels and by the reference-exact probe path (bins that are not ascending).
//  MIN records   -- the same nodes re-ordered by (minimizer hash, position, rest): the bucket of a k-mer is
//                   the hash of its MINIMIZER (the 16-mer of the k-mer with the smallest hash), so the ~8
//                   consecutive windows of a read that share a minimizer probe the same 128-B lines.
//                   {flag2 | key | rank16}: key = {low hash bits | minimizer position | the other k-16 bases}.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "utree_internal.h"

namespace utk {

constexpr uint64_t M40 = (1ull << 40) - 1;
constexpr uint32_t INVALID = 0xFFFFFFFFu;

// A bucket of the table is 64 bytes (image.bucket_words = 8: half an HBM line -- the kernels scan half as many entries per lookup and are
// 5-10 % faster) or one whole 128-byte line (bucket_words = 16: the same lines fetched, fewer overflowing buckets at twice the load: an
// image a third smaller).  Chosen when the image is built.
template <int W, int I> struct RecTraits {
    static constexpr int EW = (W == 16 ? 2 : 1) * (I == 4 ? 2 : 1);   // 8-byte words per record / table slot
    static constexpr int KW = (W == 16 ? 1 : 0);                       // word holding the top key bits, flag, rank16
    static constexpr int HCAP = 8 / EW;                                // entries of 64 bytes (a small bucket, or half a large one): 8, 4, 4, 2
};
template <int W, int I> struct Entry { uint64_t w[RecTraits<W, I>::EW]; };

template <int W, int I> __device__ __forceinline__ Entry<W, I> load_entry(const uint64_t *base, uint64_t i) {
    constexpr int EW = RecTraits<W, I>::EW;
    Entry<W, I> e;
    if constexpr (EW == 1) e.w[0] = base[i];
    else if constexpr (EW == 2) {
        const ulonglong2 v = *(const ulonglong2 *)(base + i * 2);
        e.w[0] = v.x; e.w[1] = v.y;
    } else {
        const ulonglong2 v0 = *(const ulonglong2 *)(base + i * 4), v1 = *(const ulonglong2 *)(base + i * 4 + 2);
        e.w[0] = v0.x; e.w[1] = v0.y; e.w[2] = v1.x; e.w[3] = v1.y;
    }
    return e;
}
// Table slots are read once per lookup from a table far larger than any cache: the non-temporal policy
// (`nt`) serves such random lines ~12 % faster than the default one (profiles/r01/membench_cache_policy.txt).
template <int W, int I> __device__ __forceinline__ Entry<W, I> load_slot(const uint64_t *base, uint64_t i) {
    constexpr int EW = RecTraits<W, I>::EW;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    Entry<W, I> e;
    if constexpr (EW == 1) e.w[0] = __builtin_nontemporal_load(base + i);
    else if constexpr (EW == 2) {
        const u64x2 v = __builtin_nontemporal_load((const u64x2 *)(base + i * 2));
        e.w[0] = v.x; e.w[1] = v.y;
    } else {
        const u64x2 v0 = __builtin_nontemporal_load((const u64x2 *)(base + i * 4));
        const u64x2 v1 = __builtin_nontemporal_load((const u64x2 *)(base + i * 4 + 2));
        e.w[0] = v0.x; e.w[1] = v0.y; e.w[2] = v1.x; e.w[3] = v1.y;
    }
    return e;
}

// ------------------------------------------------------------------------------------------------
// FILE records: key = the stored suffix (low 8*(W-3) bits of the word), as the reference compares it
//   W=8, I=2: {0 | rank16 | suffix40}      W=8, I=4: {suffix40}{rank32}
//   W=16,I=2: {lo64}{0 | rank16 | hi40}    W=16,I=4: {lo64}{hi40}{rank32}{0}
// ------------------------------------------------------------------------------------------------
template <int W> struct Key { uint64_t hi, lo; };   // hi = top 40 suffix bits for W=16, else 0

template <int W> __device__ __forceinline__ bool key_eq(const Key<W> &a, const Key<W> &b) {
    if constexpr (W == 16) return a.lo == b.lo && a.hi == b.hi; else return a.lo == b.lo;
}
template <int W> __device__ __forceinline__ bool key_lt(const Key<W> &a, const Key<W> &b) {
    if constexpr (W == 16) return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); else return a.lo < b.lo;
}
template <int W> __device__ __forceinline__ bool key_le(const Key<W> &a, const Key<W> &b) { return !key_lt<W>(b, a); }

template <int W, int I> __device__ __forceinline__ Key<W> file_key(const uint64_t *recs, uint64_t i) {
    constexpr int EW = RecTraits<W, I>::EW;
    Key<W> k;
    if constexpr (W == 16) { const ulonglong2 v = *(const ulonglong2 *)(recs + i * EW); k.lo = v.x; k.hi = v.y & M40; }
    else { k.lo = recs[i * EW] & M40; k.hi = 0; }
    return k;
}
template <int W, int I> __device__ __forceinline__ uint32_t file_rank(const uint64_t *recs, uint64_t i) {
    constexpr int EW = RecTraits<W, I>::EW, KW = RecTraits<W, I>::KW;
    if constexpr (I == 4) return (uint32_t)recs[i * EW + KW + 1];
    else {
        uint32_t r = (uint32_t)(recs[i * EW + KW] >> 40) & 0xFFFFu;
        return r == 0xFFFFu ? INVALID : r;
    }
}
// split of the 2k-bit word khi:klo into 24-bit prefix (itree.c:684) and suffix key (itree.c:685)
template <int W> __device__ __forceinline__ uint32_t word_prefix(uint64_t khi, uint64_t klo) {
    if constexpr (W == 4) return (uint32_t)(klo >> 8);                  // PACKSIZE=16: a 32-bit word, 24 prefix + 8 suffix bits
    else return (uint32_t)(((W == 16) ? khi : klo) >> 40);
}
template <int W> __device__ __forceinline__ Key<W> word_suffix(uint64_t khi, uint64_t klo) {
    Key<W> q;
    if constexpr (W == 16) { q.hi = khi & M40; q.lo = klo; } else if constexpr (W == 4) { q.hi = 0; q.lo = klo & 0xFFull; } else { q.hi = 0; q.lo = klo & M40; }
    return q;
}

// The reference's probe sequence, verbatim in behaviour (itree.c:699-707, 728): p = first record of the
// bin; over the remaining e-s-1 records probe record w+1 past p; "<= query" moves p there.
template <int W, int I> __device__ uint32_t exact_probe(const uint64_t *recs, uint64_t s, uint64_t e, const Key<W> &q) {
    uint64_t p = s, size = e - s - 1;
    while (size) {
        uint64_t w = size >> 1;
        Key<W> k = file_key<W, I>(recs, p + w + 1);
        if (key_le<W>(k, q)) { p += w + 1; size -= w + 1; }
        else size = w;
    }
    Key<W> k = file_key<W, I>(recs, p);
    return key_eq<W>(k, q) ? file_rank<W, I>(recs, p) : INVALID;
}

// ------------------------------------------------------------------------------------------------
// Minimizers.  m = 16 bases (32 bits).  Since image version 11 the order among a k-mer's 16-mers is CANONICAL: a 16-mer is ranked by
// mix32(the smaller of itself and its reverse complement) >> MIN_LOW_BITS, so a k-mer and its reverse complement choose the same
// double-stranded 16-mer, and the minimizer of a k-mer is the LEFTMOST 16-mer of the smallest rank ("view f").  mix32 is a bijection on
// 32 bits, so the hash h of the canonical 16-mer plus one ORIENTATION bit o (1: the k-mer holds the reverse complement of the canonical
// 16-mer) identify the minimizer exactly, and {position, the other k-16 bases} the k-mer.  The table is made of PAIRS of buckets: pair =
// what the hash addresses, bucket = 2 * pair + o -- the two orientations of one canonical 16-mer are the two halves of one (or two
// adjacent) HBM line(s), so that ONE fetch serves a read's window (in bucket o) and the reverse complement of that window (itree.c:891-898
// walks it as a second sequence) in bucket 1 - o: see lanes_core.hpp, BS.
// The reverse complement of a window w with view f(w) = (h, o, pos) is looked for under the MIRRORED view (h, 1 - o, K-16-pos): that is the
// view "g" of the k-mer x = rc(w) -- its RIGHTMOST 16-mer of the smallest rank, with o = 1 also when the 16-mer is its own reverse
// complement.  g(x) differs from f(x) only when two 16-mers of x tie in the 23 bits of the rank or the minimizer is a palindrome; such
// k-mers are stored under both views (image_build.hip: assign_k), and a lookup finds exactly one of them.
// ------------------------------------------------------------------------------------------------
// Multiply, fold the high half down, multiply: four instructions (the search kernels hash every 16-mer of every read) against the
// eight of the murmur3 finaliser the images up to version 7 used; the bits that matter -- the top ones: order and bucket -- mix as
// well on random and on low-complexity reads (runs per 150 bp read 14.12 vs 14.13, bucket load chi-square equal).
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA6Bu;
    return x;
}
// reverse complement of a 16-mer (first base in the top two bits; A=0 C=1 G=2 T=3: the complement is code ^ 3): v_bfrev, the two bits of
// every base swapped back, all bits flipped
__host__ __device__ __forceinline__ uint32_t rc16(uint32_t m) {
    const uint32_t y = __builtin_bitreverse32(m);
    return ~(((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1));
}
// The strand-independent hash of a 16-mer m with reverse complement r (UTREE_CANON_MODE, utree_internal.h) and its orientation bit: 1 when the
// hash is the reverse complement's.  of: as view f has it (0 for a 16-mer that is its own reverse complement), og: as view g has it (1).
__host__ __device__ __forceinline__ uint32_t canon_of(uint32_t m, uint32_t r, uint32_t &of, uint32_t &og) {
#if UTREE_CANON_MODE == 2
    const uint32_t hf = mix32(m), hr = mix32(r);
    of = hr < hf ? 1u : 0u; og = hr <= hf ? 1u : 0u;
    return hr < hf ? hr : hf;
#else
    of = m > r ? 1u : 0u; og = m >= r ? 1u : 0u;
    return mix32(m > r ? r : m);
#endif
}
__host__ __device__ __forceinline__ uint32_t canon_hash(uint32_t m, uint32_t &o) {
    uint32_t og;
    return canon_of(m, rc16(m), o, og);
}
// the hash alone (the search kernels' walk has m and r rolling)
__host__ __device__ __forceinline__ uint32_t canon_key(uint32_t m, uint32_t r) {
#if UTREE_CANON_MODE == 2
    const uint32_t hf = mix32(m), hr = mix32(r);
    return hr < hf ? hr : hf;
#else
    return mix32(m > r ? r : m);
#endif
}
// 16-mers are ordered by their hash WITHOUT its MIN_LOW_BITS low bits (leftmost on ties): the search kernels then fit
// { hash bits | position in the tile } into 32 bits and slide the minimum with one v_min_u32 per step.  The bucket is still
// addressed by the chosen 16-mer's full hash.
constexpr uint32_t MIN_LOW_BITS = 9;

// MIN key: W=8: lo = {hlow8 | pos5 | rest32} (45 bits), hi unused.  W=16: lo = low 64 bits of the 96-bit rest,
// hi = {hlow8 | pos6 | rest_hi32} (46 bits).
template <int W> struct MinKey { uint64_t hi, lo; };
template <int W> __device__ __forceinline__ bool mkey_eq(const MinKey<W> &a, const MinKey<W> &b) {
    if constexpr (W == 16) return a.lo == b.lo && a.hi == b.hi; else return a.lo == b.lo;
}
template <int W> __device__ __forceinline__ bool mkey_lt(const MinKey<W> &a, const MinKey<W> &b) {
    if constexpr (W == 16) return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); else return a.lo < b.lo;
}

// The bases of khi:klo outside its 16-mer at base position pos.
// W=8 : rest = 16 bases (32 bits) in rest_lo.   W=16: rest = 48 bases (96 bits) in rest_hi32:rest_lo.
template <int W> __device__ __forceinline__ void min_rest(uint64_t khi, uint64_t klo, uint32_t pos, uint32_t &rest_hi32, uint64_t &rest_lo) {
    // word = [pos bases | 16-mer | the other bases]; rest = [pos bases | the other bases].  With T = the word without its
    // last 16 bases, L = the word without its first 16 bases and M = ones over the bases after the 16-mer:
    // rest = (T & ~M) | (L & M)  -- one bit-select per 32 bits.
    if constexpr (W == 8) {
        const uint32_t T = (uint32_t)(klo >> 32), L = (uint32_t)klo;
        const uint32_t M = (uint32_t)(0xFFFFFFFFull >> (2 * pos));
        rest_lo = (T & ~M) | (L & M);
        rest_hi32 = 0;
    } else {
        const uint64_t T_hi = khi >> 32, T_lo = (khi << 32) | (klo >> 32);     // word >> 32 (96 bits: T_hi is 32 bits)
        const uint64_t L_hi = khi & 0xFFFFFFFFull, L_lo = klo;                 // word & (2^96 - 1)
        const uint32_t s = 2 * pos;                                            // 0..96
        const uint64_t M_hi = s < 32 ? (0xFFFFFFFFull >> s) : 0ull;            // (2^96 - 1) >> s, as 32 + 64 bits
        const uint64_t M_lo = s <= 32 ? ~0ull : (s >= 96 ? 0ull : (~0ull >> (s - 32)));
        rest_lo = (T_lo & ~M_lo) | (L_lo & M_lo);
        rest_hi32 = (uint32_t)((T_hi & ~M_hi) | (L_hi & M_hi));
    }
}

// 16-mer j of the word khi:klo
template <int W> __device__ __forceinline__ uint32_t mer_at(uint64_t khi, uint64_t klo, uint32_t j) {
    if constexpr (W == 8) return (uint32_t)(klo >> (32 - 2 * j));
    else { const unsigned __int128 w = ((unsigned __int128)khi << 64) | klo; return (uint32_t)(w >> (96 - 2 * j)); }
}
// Both views of the word khi:klo by direct evaluation (load time; the search kernels get view f from a sliding minimum over
// per-position hashes shared by the lanes of a wave): f = leftmost 16-mer of the smallest rank, g = rightmost, o_g = 1 for a palindrome.
template <int W> __device__ __forceinline__ void minimizer_views(uint64_t khi, uint64_t klo, uint32_t &hf, uint32_t &of, uint32_t &pf,
                                                                uint32_t &hg, uint32_t &og, uint32_t &pg) {
    // (k = 64: positions UTREE_MIN_MARGIN .. K-16 - UTREE_MIN_MARGIN only -- utree_internal.h)
    constexpr uint32_t J0 = UTREE_MIN_MARGIN(W);
    uint32_t best = 0xFFFFFFFFu;
    hf = hg = of = og = pf = pg = 0;
    for (uint32_t j = J0; j <= 4u * W - 16u - J0; ++j) {
        const uint32_t m = mer_at<W>(khi, klo, j);
        uint32_t o1, o2;
        const uint32_t hh = canon_of(m, rc16(m), o1, o2), key = hh >> MIN_LOW_BITS;
        if (j == J0 || key < best) { best = key; hf = hh; of = o1; pf = j; }
        if (j == J0 || key <= best) { hg = hh; og = o2; pg = j; }
    }
}
// The four bases around a minimizer -- two in front of it, two behind: in `rest` they are neighbours, bases pos-2 .. pos+1 -- in the minimizer's
// canonical orientation (o = 1: the reverse complement of the four): the same eight bits for a window and for its reverse complement, whose
// minimizer is the same 16-mer read the other way.  k = 64 only (UTREE_MIN_MARGIN: the four bases exist in every k-mer of the run).
__host__ __device__ __forceinline__ uint32_t ext_canon(uint32_t four, uint32_t o) { return o ? (rc16(four << 24) & 0xFFu) : four; }
template <int W> __host__ __device__ __forceinline__ uint32_t min_ext(uint32_t rest_hi32, uint64_t rest_lo, uint32_t pos, uint32_t o) {
    if constexpr (W != 16) return 0u;
    else {
        const unsigned __int128 r = ((unsigned __int128)rest_hi32 << 64) | rest_lo;    // 48 bases, base 0 on top
        return ext_canon((uint32_t)(r >> (92u - 2u * pos)) & 0xFFu, o);               // pos in 2 .. 46
    }
}
// view f alone
template <int W> __device__ __forceinline__ void minimizer(uint64_t khi, uint64_t klo, uint32_t &h, uint32_t &o, uint32_t &pos,
                                                          uint32_t &rest_hi32, uint64_t &rest_lo) {
    uint32_t hg, og, pg;
    minimizer_views<W>(khi, klo, h, o, pos, hg, og, pg);
    min_rest<W>(khi, klo, pos, rest_hi32, rest_lo);
}

// Bucket of a minimizer (hash of the canonical 16-mer, orientation) and the hash bits the bucket does not imply (utree_image_header.regions):
// region r = the hash's top 8 bits has nb_r SLOTS (any number from 2^16 to 2^24) over its 2^24 hash values, slot = (h24 * nb_r) >> 24 -- one
// multiply-high of (h << 8) --, so that a slot spans at most 256 consecutive hash values and the hash's low 8 bits tell them apart; a slot is
// one PAIR of buckets (k = 64 where a slot is one hash value: sub_r pairs, picked by the four bases around the minimizer); bucket = 2 * pair + o.
__host__ __device__ __forceinline__ uint32_t bucket_in_region(uint32_t h, uint32_t nb) {
    return (uint32_t)(((uint64_t)(h << 8) * nb) >> 32);
}
// (ext: min_ext -- k = 64: a slot of the region has sub_r pairs, the minimizer's four neighbours pick one; k = 32: 0, and sub_r is 1)
__host__ __device__ __forceinline__ uint64_t pair_of(uint64_t e, uint32_t h, uint32_t ext) {
    const uint32_t nb = (uint32_t)e & ((1u << UTREE_REGION_NB_BITS) - 1u), sub = (uint32_t)(e >> UTREE_REGION_NB_BITS) & ((1u << UTREE_REGION_SUB_BITS) - 1u);
    return (e >> UTREE_REGION_BASE_SHIFT) + (uint64_t)bucket_in_region(h, nb) * sub + ((ext * sub) >> 8);
}
__host__ __device__ __forceinline__ void bucket_of(const uint64_t *__restrict__ regions, uint32_t h, uint32_t o, uint32_t ext, uint64_t &bucket, uint32_t &hlow) {
    bucket = 2u * pair_of(regions[h >> 24], h, ext) + o;
    hlow = h & 0xFFu;
}

// bucket and MIN key from (h, o, pos)
template <int W> __device__ __forceinline__ void min_finish(uint64_t khi, uint64_t klo, uint32_t h, uint32_t o, uint32_t pos,
                                                           const uint64_t *__restrict__ regions, uint64_t &bucket, MinKey<W> &mk) {
    uint32_t rh, hl; uint64_t rl;
    min_rest<W>(khi, klo, pos, rh, rl);
    bucket_of(regions, h, o, min_ext<W>(rh, rl, pos, o), bucket, hl);
    const uint64_t hlow = hl;
    if constexpr (W == 8) { mk.hi = 0; mk.lo = (hlow << 37) | ((uint64_t)pos << 32) | rl; }
    else { mk.lo = rl; mk.hi = (hlow << 38) | ((uint64_t)pos << 32) | rh; }
}
template <int W> __device__ __forceinline__ void min_split(uint64_t khi, uint64_t klo, const uint64_t *__restrict__ regions,
                                                          uint64_t &bucket, MinKey<W> &mk) {
    uint32_t h, o, pos, rh; uint64_t rl;
    minimizer<W>(khi, klo, h, o, pos, rh, rl);
    min_finish<W>(khi, klo, h, o, pos, regions, bucket, mk);
}

// MIN records / bucket entries.  flag (top 2 bits of word KW): 0 record, 1 empty entry, 2 (a bucket's LAST entry only)
// overflow: run {count22 | start40} of the bucket's remaining records in the sorted array
//   W=8, I=2: {flag2 | hlow8 pos5 | 0 | rank16 | rest32}    W=8, I=4: {flag2 | hlow8 pos5 | 0 | 0 | rest32}{rank32}
//   W=16,I=2: {rest lo64}{flag2 | key_hi46 | rank16}        W=16,I=4: {rest lo64}{flag2 | key_hi46 | 0}{rank32}{0}
// (W=8 keeps the 32 "rest" bits in its low word and everything else in its high word: the bucket scan of the wave-per-read kernels
// is then one 32-bit compare and one select per entry, wave_common.hpp::scan_bucket82.  The zero bit between position and rank makes
// `high word >> 16` both the tag {flag, hash bits, position} << 1 -- one subtract and one compare test it against a run's range of
// tags -- and, in its low six bits, the shift 2 * position that brings a window's outer bases under the entry's rest: the lane-per-read
// pass, lanes_kernel.hip, checks an entry with five instructions.)
constexpr uint64_t MFLAG_EMPTY = 1ull << 62, MFLAG_RUN = 2ull << 62;
// An overflow descriptor is {flag 2 | count22 | heavy1 | start39}.  A HEAVY run -- more than OVF_DIR_MIN records, all of one hash value: the k-mers
// of many related genomes around one minimizer -- is stored in one of two ways:
// * k = 64 (and k = 32 with UTREE_OVF_CHAINS=0): a DIRECTORY in front of the records: K-16+2 16-bit offsets, [p] = the run's first record whose minimizer
//   position is >= p ([K-15] = count), in OVF_DIR_SLOTS record slots.  A window's record can only be among those of the window's own position,
//   so a search starts in a range of count / (K-15) records instead of the whole run.
// * k = 32 (image flag UTREE_F_OVF_CHAINS): as CHAINS.  The k-mers around one occurrence of the minimizer in a genome -- positions 16 down to 0 as the
//   k-mer slides to the right -- are windows of ONE 48-base stretch, and related genomes share most of it: a chain is a maximal sequence of records at
//   consecutive positions p0 .. p1 each of which overlaps the next in 31 bases (image_build.hip: ChainRun links them), so it is described by the 16 bases
//   in front of the minimizer (A, those the records know, the rest 0) and the 16 behind it (B): record p of the chain has rest (A:B) >> 2p.  The run is
//   {A:B}{p0 8 | p1 8 | 0 16 | index of the chain's first rank 32} per chain (count22 = chains), then one rank per record, chain after chain.  A window
//   (position p, rest r) is in the database iff some chain has p0 <= p <= p1 and (uint32)((A:B) >> 2p) == r -- every record is in exactly one chain --, and
//   then its rank is ranks[first + p - p0]: 16 bytes per chain and 2 (4) per k-mer instead of 8 (16) per k-mer, and a search is one pass over a dozen
//   chains plus one load instead of a directory trip and a bisection.
// (image_build.hip: ovf_count_k, ovf_move_k; lanes_core.hpp, wave_common.hpp: the searches).
constexpr uint64_t M39 = (1ull << 39) - 1, OVF_HAS_DIR = 1ull << 39;
constexpr uint32_t OVF_DIR_MIN = 32;
template <int W, int I> struct OvfDir { static constexpr uint32_t SLOTS = (2u * (4u * W - 16u + 2u) + 8u * RecTraits<W, I>::EW - 1u) / (8u * RecTraits<W, I>::EW); };
// start of the records and count of the run a descriptor names
__host__ __device__ __forceinline__ uint64_t ovf_count(uint64_t d) { return (d >> 40) & 0x3FFFFFull; }
// (directory format)
template <int W, int I> __host__ __device__ __forceinline__ uint64_t ovf_first(uint64_t d) { return (d & M39) + ((d & OVF_HAS_DIR) ? OvfDir<W, I>::SLOTS : 0u); }
// minimizer position of a MIN record
template <int W, int I> __device__ __forceinline__ uint32_t mrec_pos(const uint64_t *rec) {
    if constexpr (W == 16) return (uint32_t)(rec[1] >> 48) & 63u; else return (uint32_t)(rec[0] >> 49) & 31u;
}
template <int W, int I> __device__ __forceinline__ uint32_t mrec_hlow(const uint64_t *rec) {
    if constexpr (W == 16) return (uint32_t)(rec[1] >> 54) & 0xFFu; else return (uint32_t)(rec[0] >> 54) & 0xFFu;
}
// A window (minimizer position pos, outer bases rest) against chain {w0, w1}: the index of its rank, or ~0
__device__ __forceinline__ uint32_t chain_hit(uint64_t w0, uint64_t w1, uint32_t pos, uint32_t rest) {
    const uint32_t p0 = (uint32_t)(w1 >> 56), p1 = (uint32_t)(w1 >> 48) & 0xFFu;
    return (pos - p0 <= p1 - p0 && (uint32_t)(w0 >> (2u * pos)) == rest) ? (uint32_t)w1 + (pos - p0) : ~0u;
}
// ... against a run of n chains at `run` (one lane, one chain after the other: the wave-per-read kernels)
template <int I> __device__ __forceinline__ uint32_t chain_find(const uint64_t *run, uint32_t n, uint32_t pos, uint32_t rest) {
    uint32_t at = ~0u;
    for (uint32_t c = 0; c < n && at == ~0u; ++c) at = chain_hit(run[2u * c], run[2u * c + 1u], pos, rest);
    if (at == ~0u) return INVALID;
    if constexpr (I == 2) { const uint32_t r = ((const uint16_t *)(run + 2u * n))[at]; return r == 0xFFFFu ? INVALID : r; }
    else return ((const uint32_t *)(run + 2u * n))[at];
}
constexpr uint64_t M46 = (1ull << 46) - 1;

template <int W, int I> __device__ __forceinline__ uint32_t mrec_flag(const Entry<W, I> &e) { return (uint32_t)(e.w[RecTraits<W, I>::KW] >> 62); }
template <int W, int I> __device__ __forceinline__ MinKey<W> mrec_key(const Entry<W, I> &e) {
    MinKey<W> k;
    if constexpr (W == 16) { k.lo = e.w[0]; k.hi = (e.w[1] >> 16) & M46; }
    else { k.hi = 0; k.lo = (((e.w[0] >> 49) & 0x1FFFull) << 32) | (e.w[0] & 0xFFFFFFFFull); }
    return k;
}
template <int W, int I> __device__ __forceinline__ uint32_t mrec_rank(const Entry<W, I> &e) {
    if constexpr (I == 4) return (uint32_t)e.w[RecTraits<W, I>::KW + 1];
    else {
        uint32_t r = (W == 8 ? (uint32_t)(e.w[0] >> 32) : (uint32_t)e.w[RecTraits<W, I>::KW]) & 0xFFFFu;
        return r == 0xFFFFu ? INVALID : r;
    }
}
template <int W, int I> __device__ __forceinline__ Entry<W, I> make_mrec(const MinKey<W> &k, uint32_t rank) {
    Entry<W, I> e;
#pragma unroll
    for (int j = 0; j < RecTraits<W, I>::EW; ++j) e.w[j] = 0;
    const uint64_t r16 = (I == 2) ? (rank == INVALID ? 0xFFFFull : (uint64_t)(rank & 0xFFFFu)) : 0ull;
    if constexpr (W == 16) { e.w[0] = k.lo; e.w[1] = (k.hi << 16) | r16; }
    else { e.w[0] = ((k.lo >> 32) << 49) | (r16 << 32) | (k.lo & 0xFFFFFFFFull); }
    if constexpr (I == 4) e.w[RecTraits<W, I>::KW + 1] = rank;
    return e;
}

// ------------------------------------------------------------------------------------------------
// dispatch on (W, I) and on the image flavour
// ------------------------------------------------------------------------------------------------
template <int V> using IC = std::integral_constant<int, V>;

template <typename Fn> int dispatch_wi(uint32_t W, uint32_t I, Fn &&fn) {
    if (W == 8 && I == 2) fn(IC<8>{}, IC<2>{});
    else if (W == 8 && I == 4) fn(IC<8>{}, IC<4>{});
    else if (W == 16 && I == 2) fn(IC<16>{}, IC<2>{});
    else if (W == 16 && I == 4) fn(IC<16>{}, IC<4>{});
    else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

// ... and W = 4 (PACKSIZE=16) for the code that exists for it: repacking, the bin-table check, COMPRESS
template <typename Fn> int dispatch_wi_all(uint32_t W, uint32_t I, Fn &&fn) {
    if (W == 4 && I == 2) fn(IC<4>{}, IC<2>{});
    else if (W == 4 && I == 4) fn(IC<4>{}, IC<4>{});
    else return dispatch_wi(W, I, fn);
    return (int)hipGetLastError();
}

template <typename Fn> int dispatch_img(const utk_image *im, Fn &&fn) {
    const bool exc = (im->flags & (UTREE_F_IRREGULAR | UTREE_F_GENERIC)) != 0;
    const bool o64 = (im->flags & UTREE_F_OFF64) != 0;
    return dispatch_wi(im->W, im->I, [&](auto w, auto i) {
        if (exc && o64) fn(w, i, std::true_type{}, uint64_t{});
        else if (exc) fn(w, i, std::true_type{}, uint32_t{});
        else if (o64) fn(w, i, std::false_type{}, uint64_t{});
        else fn(w, i, std::false_type{}, uint32_t{});
    });
}

// the GG search kernels also exist for W = 4 (PACKSIZE=16): a direct-address table holds every word's answer, the irregular bins'
// included (image_build.hip: direct_*_k), so neither the exact-probe path nor 64-bit offsets are instantiated for it
template <typename Fn> int dispatch_img_all(const utk_image *im, Fn &&fn) {
    if (im->W == 4) {
        if (im->I == 2) fn(IC<4>{}, IC<2>{}, std::false_type{}, uint32_t{});
        else if (im->I == 4) fn(IC<4>{}, IC<4>{}, std::false_type{}, uint32_t{});
        else return (int)hipErrorInvalidValue;
        return (int)hipGetLastError();
    }
    return dispatch_img(im, fn);
}

}  // namespace utk
