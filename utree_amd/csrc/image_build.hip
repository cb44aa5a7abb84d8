// image_build.hip -- load-time kernels: turn the on-disk pieces of a .ctr (bin table + packed node dump, already in
// HBM) into the device image of DESIGN.md §3.  Runs once per database; everything here streams or sorts.
//
//   repack_k     on-disk SZ-byte records -> FILE records (8-byte words, label index -> strcmp rank)
//   widen/validate   bin table at the image's offset width; per-bin "strictly ascending" check (irregular bitmap)
//   assign_k     per node: minimizer hash, position, rest, 24-bit prefix
//   (rocprim radix sorts: nodes ordered by (hash, position, rest))
//   emit_k       MIN records in that order
//   bucket_k     128-byte buckets addressed by the minimizer hash through the region table: records inline, overflow descriptor
#include <hip/hip_runtime.h>
#include <cstring>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include "device_common.hpp"

using namespace utk;

namespace {

template <int W, int I>
__global__ void repack_k(const uint8_t *__restrict__ raw, uint64_t count, const uint32_t *__restrict__ ix2rank,
                         uint32_t n_labels, uint64_t *__restrict__ recs, unsigned long long *__restrict__ invalid) {
    constexpr int SZ = W + I - 3, SB = W - 3, EW = RecTraits<W, I>::EW;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t *p = raw + i * SZ;
        uint64_t lo = 0, hi = 0;
#pragma unroll
        for (int b = 0; b < SB; ++b) {
            if (b < 8) lo |= (uint64_t)p[b] << (8 * b); else hi |= (uint64_t)p[b] << (8 * (b - 8));
        }
        uint32_t ix = 0;
#pragma unroll
        for (int b = 0; b < I; ++b) ix |= (uint32_t)p[SB + b] << (8 * b);
        uint32_t rank = ix < n_labels ? ix2rank[ix] : INVALID;      // itree.c:929 `ix < maxIX`
        if (rank == INVALID) atomicAdd(invalid, 1ull);               // (no well-formed database has such a node)
        uint64_t *o = recs + i * EW;
        const uint64_t r16 = rank == INVALID ? 0xFFFFull : (uint64_t)rank;
        if constexpr (W <= 8 && I == 2) { o[0] = lo | (r16 << 40); }             // (W = 4, PACKSIZE=16: a suffix of 8 bits in the same place)
        else if constexpr (W <= 8 && I == 4) { o[0] = lo; o[1] = rank; }
        else if constexpr (W == 16 && I == 2) { o[0] = lo; o[1] = hi | (r16 << 40); }
        else { o[0] = lo; o[1] = hi; o[2] = rank; o[3] = 0; }
    }
}

// bin table: on-disk width -> the image's OFF width (zero-extended, itree.c:756-759)
template <typename OFF> __global__ void widen_binix_k(const void *raw, uint32_t width, OFF *coarse) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= UTREE_NUMBINS) return;
    coarse[i] = (OFF)(width == 4 ? (uint64_t)((const uint32_t *)raw)[i] : ((const uint64_t *)raw)[i]);
}

// the 24-bit bin that holds node j of a monotone table: largest p with coarse[p] <= j (< coarse[p+1])
template <typename OFF> __device__ __forceinline__ uint32_t bin_of(const OFF *__restrict__ coarse, uint64_t j) {
    uint32_t a = 0, b = UTREE_NUMBINS - 1;
    while (b - a > 1) { const uint32_t m = a + ((b - a) >> 1); if ((uint64_t)coarse[m] <= j) a = m; else b = m; }
    return a;
}

// Work is spread over NODES, not bins: a `.ctr` may put billions of nodes into one bin (the reference only ever
// binary-searches inside a bin, itree.c:699-707), and one thread per bin would then walk it alone.
template <typename OFF>
__global__ void monotone_k(const OFF *__restrict__ coarse, uint64_t n_nodes, unsigned long long *counters) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= UTREE_NUMBINS - 1) return;
    const uint64_t s = coarse[p], e = coarse[p + 1];
    if (s > e || e > n_nodes) counters[1] = 1;
}
constexpr uint32_t NODE_RUN = 16;     // consecutive nodes per thread: one binary search, then the bin only moves forward
template <int W, int I, typename OFF>
__global__ void validate_k(const OFF *__restrict__ coarse, const uint64_t *__restrict__ recs, uint64_t n_nodes,
                           uint32_t *irreg, unsigned long long *counters) {
    if (counters[1]) return;                                           // not monotone: every bin takes the exact-probe path anyway
    const uint64_t c0 = coarse[0], cN = coarse[UTREE_NUMBINS - 1];
    for (uint64_t j0 = c0 + 1 + ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * NODE_RUN; j0 < cN;
         j0 += (uint64_t)gridDim.x * blockDim.x * NODE_RUN) {
        uint32_t p = bin_of<OFF>(coarse, j0);
        const uint64_t j1 = j0 + NODE_RUN < cN ? j0 + NODE_RUN : cN;
        for (uint64_t j = j0; j < j1; ++j) {
            while ((uint64_t)coarse[p + 1] <= j) ++p;
            if (j == (uint64_t)coarse[p]) continue;                    // first node of its bin
            if (!key_lt<W>(file_key<W, I>(recs, j - 1), file_key<W, I>(recs, j))) {
                const uint32_t bit = 1u << (p & 31);                   // look first: a bin that is all out of order would
                if (!(__atomic_load_n(&irreg[p >> 5], __ATOMIC_RELAXED) & bit) &&   // otherwise queue one atomic per node
                    !(atomicOr(&irreg[p >> 5], bit) & bit)) atomicAdd(&counters[0], 1ull);
            }
        }
    }
}

__global__ void fill_pad_k(uint64_t *p, uint32_t words) {
    if (threadIdx.x < words) p[threadIdx.x] = ~0ull;
}
__global__ void fill_u64_k(uint64_t *p, uint64_t n, uint64_t v, uint32_t stride, uint32_t at) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        p[i * stride + at] = v;
}
template <typename IDX> __global__ void iota_k(IDX *p, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = (IDX)i;
}
template <typename IDX, typename T> __global__ void gather_k(const T *__restrict__ src, const IDX *__restrict__ idx, T *__restrict__ dst, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[idx[i]];
}

// minimizer pieces of every node the bin table reaches (nodes [c0, c0+m)); one thread per 16 consecutive nodes, the first one's bin by
// binary search.  Entry t < m is node t under its view f; a node whose view g differs (device_common.hpp) appends a second entry behind the
// m first ones -- SRC[slot] = t, one atomic per such node -- as long as there is room (dup_cap); *ndup counts them all.
// H carries the orientation with the hash: H[t] = h, O[t >> 5] bit t & 31 = o  (entries >= m: their own words, written whole by one thread
// each would race -- so orientation bits are kept per entry in a byte array)
template <int W, int I, typename OFF>
__global__ void assign_k(const OFF *__restrict__ coarse, const uint64_t *__restrict__ recs, uint64_t c0, uint64_t m, uint64_t dup_cap,
                         uint32_t *__restrict__ H, uint8_t *__restrict__ O, uint64_t *__restrict__ K1, uint64_t *__restrict__ K2,
                         uint64_t *__restrict__ SRC, unsigned long long *__restrict__ ndup) {
    for (uint64_t t0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * NODE_RUN; t0 < m; t0 += (uint64_t)gridDim.x * blockDim.x * NODE_RUN) {
        uint64_t p = bin_of<OFF>(coarse, c0 + t0);
        const uint64_t t1 = t0 + NODE_RUN < m ? t0 + NODE_RUN : m;
        for (uint64_t t = t0; t < t1; ++t) {
            const uint64_t j = c0 + t;
            while ((uint64_t)coarse[p + 1] <= j) ++p;
            const Key<W> k = file_key<W, I>(recs, j);
            uint64_t khi, klo;
            if constexpr (W == 16) { khi = (p << 40) | k.hi; klo = k.lo; } else { khi = 0; klo = (p << 40) | k.lo; }
            uint32_t hf, of, pf, hg, og, pg, rh; uint64_t rl;
            minimizer_views<W>(khi, klo, hf, of, pf, hg, og, pg);
            min_rest<W>(khi, klo, pf, rh, rl);
            H[t] = hf; O[t] = (uint8_t)of;
            if constexpr (W == 16) { K1[t] = rl; K2[t] = ((uint64_t)pf << 32) | rh; }
            else K1[t] = ((uint64_t)pf << 32) | rl;
            if (pg != pf || og != of) {
                const unsigned long long slot = atomicAdd(ndup, 1ull);
                if (slot < dup_cap) {
                    const uint64_t u = m + slot;
                    min_rest<W>(khi, klo, pg, rh, rl);
                    H[u] = hg; O[u] = (uint8_t)og; SRC[slot] = t;
                    if constexpr (W == 16) { K1[u] = rl; K2[u] = ((uint64_t)pg << 32) | rh; }
                    else K1[u] = ((uint64_t)pg << 32) | rl;
                }
            }
        }
    }
}

// the four bases around entry i's minimizer (device_common.hpp: min_ext) from its sort keys
template <int W> __device__ __forceinline__ uint32_t entry_ext(const uint64_t *__restrict__ K1, const uint64_t *__restrict__ K2, const uint8_t *__restrict__ O, uint64_t i) {
    if constexpr (W == 16) return min_ext<W>((uint32_t)K2[i], K1[i], (uint32_t)(K2[i] >> 32), O[i]); else return 0u;
}

// MIN record j = entry idx[j] in the final (bucket, hash bits, pos, rest) order; entry i is node i, or node SRC[i - m] under its second view
template <int W, int I, typename IDX>
__global__ void emit_k(const uint64_t *__restrict__ recs, uint64_t c0, const IDX *__restrict__ idx, const uint32_t *__restrict__ H, const uint8_t *__restrict__ O,
                       const uint64_t *__restrict__ K1, const uint64_t *__restrict__ K2, const uint64_t *__restrict__ SRC, uint64_t m_nodes,
                       const uint64_t *__restrict__ regions, uint64_t m, uint64_t *__restrict__ out, uint64_t *__restrict__ Bs) {
    constexpr int EW = RecTraits<W, I>::EW;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = idx[j];
        const uint32_t h = H[i];
        uint64_t bucket; uint32_t hl;
        bucket_of(regions, h, O[i], entry_ext<W>(K1, K2, O, i), bucket, hl);
        const uint64_t hlow = hl;
        MinKey<W> mk;
        if constexpr (W == 16) { mk.lo = K1[i]; mk.hi = (hlow << 38) | K2[i]; } else { mk.hi = 0; mk.lo = (hlow << 37) | K1[i]; }
        const uint64_t node = i < m_nodes ? i : SRC[i - m_nodes];
        const Entry<W, I> e = make_mrec<W, I>(mk, file_rank<W, I>(recs, c0 + node));
#pragma unroll
        for (int x = 0; x < EW; ++x) out[j * EW + x] = e.w[x];
        Bs[j] = bucket;
    }
}

// Buckets from the sorted records: the first thread of every bucket's run copies up to CAP records into the bucket (the
// rest of the bucket stays flagged empty); a longer run leaves CAP-1 records inline and an overflow descriptor last.
template <int W, int I>
__global__ void bucket_k(const uint64_t *__restrict__ Bs, const uint64_t *__restrict__ mrecs, uint64_t base, uint64_t m,
                         uint64_t *__restrict__ table, uint32_t bw, uint64_t run_max, unsigned long long *overflow) {
    // Bs / mrecs: this part's sorted bucket numbers and records (a part = a range of whole buckets); base = MIN records in earlier parts
    constexpr int EW = RecTraits<W, I>::EW, KW = RecTraits<W, I>::KW;
    const int CAP = (int)bw / EW;                                                     // entries of a bucket (bw = its 8-byte words: 8 or 16)
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = Bs[j];
        if (j && Bs[j - 1] == b) continue;                                            // not the first record of its bucket
        uint64_t n = 1;
        while (j + n < m && Bs[j + n] == b) ++n;
        uint64_t *o = table + b * bw;
        const uint64_t inl = n <= (uint64_t)CAP ? n : (uint64_t)CAP - 1;
        for (uint64_t q = 0; q < inl; ++q)
#pragma unroll
            for (int x = 0; x < EW; ++x) o[q * EW + x] = mrecs[(j + q) * EW + x];     // flag bits of a record are 0
        if (n > (uint64_t)CAP) {
            uint64_t rest = n - inl;
            // a run the 22-bit count cannot describe: the descriptor saturates (never followed: flag_saturated_k sends the words
            // of these nodes' bins down the exact-probe path instead) and the bucket is counted
            if (rest > run_max) { atomicAdd(overflow, 1ull); rest = (1ull << 22) - 1; }
#pragma unroll
            for (int x = 0; x < EW; ++x) o[inl * EW + x] = 0;
            o[inl * EW + KW] = MFLAG_RUN | (rest << 40) | ((base + j + inl) & M39);
        }
    }
}

// Nodes of a bucket whose run saturated its descriptor: their 24-bit bins are flagged irregular, so that every word of those
// bins takes the reference's own probe sequence over the FILE records (exact_probe) and never reads the saturated descriptor's
// count.  Only launched for a part in which bucket_k counted such a bucket (e.g. millions of k-mers that contain A^16, whose
// hash is 0: all of them share the first bucket).
template <int W, int I, typename OFF, typename IDX>
__global__ void flag_saturated_k(const uint64_t *__restrict__ Bs, const IDX *__restrict__ idx, uint64_t m, uint64_t c0, const uint64_t *__restrict__ SRC, uint64_t m_nodes,
                                 const OFF *__restrict__ coarse, const uint64_t *__restrict__ table, uint32_t bw,
                                 uint32_t *irreg, unsigned long long *counters) {
    constexpr int EW = RecTraits<W, I>::EW, KW = RecTraits<W, I>::KW;
    const int CAP = (int)bw / EW;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = Bs[j];
        const uint64_t d = table[b * bw + (uint64_t)(CAP - 1) * EW + KW];
        if ((d >> 62) != 2 || ((d >> 40) & 0x3FFFFFull) != 0x3FFFFFull) continue;
        const uint64_t i = idx[j], node = i < m_nodes ? i : SRC[i - m_nodes];
        const uint32_t p = bin_of<OFF>(coarse, c0 + node);
        const uint32_t bit = 1u << (p & 31);
        if (!(__atomic_load_n(&irreg[p >> 5], __ATOMIC_RELAXED) & bit) && !(atomicOr(&irreg[p >> 5], bit) & bit)) atomicAdd(&counters[1], 1ull);
    }
}

// ---- chains (k = 32; device_common.hpp: OvfChain) ----
// A heavy run's records ascend by (position, rest).  The k-mer of position p holds the p bases in front of the minimizer and the 16 - p behind it;
// the k-mer one base to the LEFT on the same sequence has position p + 1: one more base in front, one fewer behind.  Record j at p - 1 and record
// i at p can follow each other iff rest_j >> 2 == rest_i & 0x3FFFFFFF (the 15 bases they share).  All the records at p - 1 with one such value (up
// to four, consecutive) fit all the records at p with it (up to four, one per leading base): the t-th of the former is linked to the t-th of the
// latter.  Every record has at most one link each way, so the links cut the run into chains, each the k-mers of one stretch of sequence.
struct ChainRun {
    const uint64_t *r;           // the run's records (one word each: W = 8; the rank of u32 labels in a second word)
    uint32_t ew, n;
    uint32_t g[18];              // [p] = first record of position >= p; [17] = n
    __device__ __forceinline__ uint32_t rest(uint32_t i) const { return (uint32_t)r[(uint64_t)i * ew]; }
    __device__ __forceinline__ uint32_t pos(uint32_t i) const { return (uint32_t)(r[(uint64_t)i * ew] >> 49) & 31u; }
    __device__ uint32_t lb(uint32_t p, uint32_t v) const {              // first record of position p with rest >= v (g[p + 1]: none)
        uint32_t lo = g[p], hi = g[p + 1];
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rest(mid) < v) lo = mid + 1; else hi = mid; }
        return lo;
    }
    __device__ void init(const uint64_t *recs, uint32_t ew_, uint32_t n_) {
        r = recs; ew = ew_; n = n_;
        uint32_t p = 0;
        for (uint32_t q = 0; q < n; ++q) { const uint32_t pq = pos(q); while (p <= pq && p < 18u) g[p++] = q; }
        while (p < 18u) g[p++] = n;
    }
    // the record i (position p >= 1) follows, or ~0
    __device__ uint32_t pred(uint32_t i, uint32_t p) const {
        if (p == 0u) return ~0u;
        const uint32_t ri = rest(i), low30 = ri & 0x3FFFFFFFu, x = ri >> 30;
        uint32_t t = 0;                                                  // records at p with the same 15 bases and a smaller leading base
        for (uint32_t y = 0; y < x; ++y) { const uint32_t v = (y << 30) | low30, a = lb(p, v); t += (a < g[p + 1] && rest(a) == v) ? 1u : 0u; }
        const uint32_t a = lb(p - 1u, low30 << 2);
        return (a + t < g[p] && (rest(a + t) >> 2) == low30) ? a + t : ~0u;
    }
    // the record that follows j (position p <= 15), or ~0
    __device__ uint32_t succ(uint32_t j, uint32_t p) const {
        if (p >= 16u) return ~0u;
        const uint32_t low30 = rest(j) >> 2;
        uint32_t t = j - lb(p, low30 << 2);
        for (uint32_t y = 0; y < 4u; ++y) {
            const uint32_t v = (y << 30) | low30, a = lb(p + 1u, v);
            if (a < g[p + 2u] && rest(a) == v) { if (t == 0u) return a; --t; }
        }
        return ~0u;
    }
};
// 8-byte words of a run written as chains: 2 per chain + the ranks
__host__ __device__ __forceinline__ uint32_t chain_words(uint32_t n_chains, uint32_t n_rec, uint32_t label_bytes) { return 2u * n_chains + (n_rec * label_bytes + 7u) / 8u; }

// After the table is built only the overflow runs of the MIN array are ever read again (records that sit inline in a bucket
// are found there).  ovf_count_k / ovf_move_k keep those runs alone, packed in bucket order, and point the descriptors at
// their new places: the sorted array of ALL nodes (8-32 bytes per node) leaves the image.
// cnt[b] = record slots of bucket b's run in the packed area; bit 31: the run is heavy (directory or chains in front / instead)
template <int W, int I>
__global__ void ovf_count_k(const uint64_t *__restrict__ table, uint64_t n_buckets, uint32_t bw, const uint64_t *__restrict__ mrecs, int dir_on, int chains_on, uint32_t *__restrict__ cnt) {
    constexpr int EW = RecTraits<W, I>::EW, KW = RecTraits<W, I>::KW;
    const int CAP = (int)bw / EW;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_buckets; b += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t d = table[b * bw + (uint64_t)(CAP - 1) * EW + KW];
        uint32_t c = (d >> 62) == 2 ? (uint32_t)((d >> 40) & 0x3FFFFFull) : 0u;
        if (c == 0x3FFFFFu) c = 0;           // a saturated run is never followed (its nodes' bins take the exact-probe path): nothing to keep
        // a heavy run of ONE hash value (sorted by key: the first and the last record tell) gets a position directory in front (device_common.hpp),
        // or becomes a list of chains
        if (c > OVF_DIR_MIN && c <= 0xFFFFu && (dir_on || chains_on)) {
            const uint64_t src = d & M39;
            if (mrec_hlow<W, I>(mrecs + src * EW) == mrec_hlow<W, I>(mrecs + (src + c - 1) * EW)) {
                if constexpr (W == 8) {
                    if (chains_on) {
                        ChainRun cr;
                        cr.init(mrecs + src * EW, EW, c);
                        uint32_t heads = 0;
                        for (uint32_t i = 0; i < c; ++i) heads += cr.pred(i, cr.pos(i)) == ~0u ? 1u : 0u;
                        c = ((chain_words(heads, c, I) + EW - 1u) / EW) | 0x80000000u;
                    } else c += OvfDir<W, I>::SLOTS | 0x80000000u;
                } else c += OvfDir<W, I>::SLOTS | 0x80000000u;
            }
        }
        cnt[b] = c;
    }
}
template <int W, int I>
__global__ void ovf_move_k(uint64_t *__restrict__ table, uint64_t n_buckets, uint32_t bw, const uint32_t *__restrict__ cnt,
                           const uint64_t *__restrict__ prefix, const uint64_t *__restrict__ mrecs, int chains_on, uint64_t *__restrict__ packed) {
    constexpr int EW = RecTraits<W, I>::EW, KW = RecTraits<W, I>::KW;
    const int CAP = (int)bw / EW;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_buckets; b += (uint64_t)gridDim.x * blockDim.x) {
        const bool dir = (cnt[b] & 0x80000000u) != 0u;
        const uint32_t c = cnt[b] & 0x7FFFFFFFu;                                  // slots in the packed area: the directory's, if any, and the records'
        uint64_t *dp = table + b * bw + (uint64_t)(CAP - 1) * EW + KW;
        if (!c) {
            // a saturated descriptor becomes an empty run: a word that is no node finds nothing there (words of nodes never come here)
            if ((*dp >> 62) == 2) *dp = MFLAG_RUN;
            continue;
        }
        const uint64_t d = *dp, src = d & M39, dst = prefix[b];
        if constexpr (W == 8) {
            if (dir && chains_on) {
                // [2 words per chain: {16 bases in front of the minimizer : 16 behind it}, {first position 8 | last position 8 | 0 | index of the first rank 32}]
                // [one rank per record, chain after chain, each in position order]
                const uint32_t n = (uint32_t)ovf_count(d);
                ChainRun cr;
                cr.init(mrecs + src * EW, EW, n);
                uint64_t *out = packed + dst * EW;
                uint32_t heads = 0;
                for (uint32_t i = 0; i < n; ++i) heads += cr.pred(i, cr.pos(i)) == ~0u ? 1u : 0u;
                for (uint32_t x = 0; x < c * EW; ++x) out[x] = 0;
                uint16_t *r16 = (uint16_t *)(out + 2u * heads);
                uint32_t *r32 = (uint32_t *)(out + 2u * heads);
                uint32_t ch = 0, at = 0;
                for (uint32_t i = 0; i < n; ++i) {
                    const uint32_t p0 = cr.pos(i);
                    if (cr.pred(i, p0) != ~0u) continue;
                    uint32_t cur = i, p = p0;
                    const uint32_t first = at;
                    for (;;) {
                        const uint64_t *rec = mrecs + (src + cur) * EW;
                        if constexpr (I == 2) r16[at] = (uint16_t)(rec[0] >> 32); else r32[at] = (uint32_t)rec[1];
                        ++at;
                        const uint32_t nx = cr.succ(cur, p);
                        if (nx == ~0u) break;
                        cur = nx; ++p;
                    }
                    const uint32_t rl = cr.rest(cur), rf = cr.rest(i);
                    const uint64_t A = p ? (uint64_t)(rl >> (32u - 2u * p)) : 0ull, B = p0 < 16u ? (uint64_t)(uint32_t)(rf << (2u * p0)) : 0ull;
                    out[2u * ch] = (A << 32) | B;
                    out[2u * ch + 1u] = ((uint64_t)p0 << 56) | ((uint64_t)p << 48) | first;
                    ++ch;
                }
                *dp = (d & ~(M39 | OVF_HAS_DIR | (0x3FFFFFull << 40))) | ((uint64_t)heads << 40) | (dst & M39) | OVF_HAS_DIR;
                continue;
            }
        }
        const uint32_t hs = dir ? OvfDir<W, I>::SLOTS : 0u, nrec = c - hs;
        if (dir) {
            // [p] = records of the run with a minimizer position below p (they ascend by position: the hash bits in front of it are one value's)
            uint16_t *dv = (uint16_t *)(packed + dst * EW);
            for (uint32_t x = 0; x < hs * EW * 4u; ++x) dv[x] = (uint16_t)nrec;
            uint32_t p = 0;
            for (uint32_t q = 0; q < nrec; ++q) {
                const uint32_t pq = mrec_pos<W, I>(mrecs + (src + q) * EW);
                while (p <= pq) dv[p++] = (uint16_t)q;
            }
        }
        for (uint64_t q = 0; q < (uint64_t)nrec * EW; ++q) packed[(dst + hs) * EW + q] = mrecs[src * EW + q];
        *dp = (d & ~(M39 | OVF_HAS_DIR)) | (dst & M39) | (dir ? OVF_HAS_DIR : 0ull);
    }
}

// COMPRESS (itree.c:1282-1286, 1306-1309): per record, drop the 3 prefix bytes; per prefix, the smallest NON-ZERO index
template <int W, int I>
__global__ void compress_chunk_k(const uint8_t *__restrict__ in, uint64_t first, uint64_t count, unsigned long long *first_ix,
                                 uint8_t *__restrict__ out) {
    constexpr int DR = W + I, SZ = W + I - 3, SB = W - 3;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t *p = in + t * DR;
        uint8_t *o = out + t * SZ;
#pragma unroll
        for (int b = 0; b < SB; ++b) o[b] = p[b];
#pragma unroll
        for (int b = 0; b < I; ++b) o[SB + b] = p[W + b];
        const uint32_t v = ((uint32_t)p[W - 1] << 16) | ((uint32_t)p[W - 2] << 8) | p[W - 3];      // top 24 bits of the LE word
        const uint64_t gi = first + t;
        if (gi) atomicMin(&first_ix[v], (unsigned long long)gi);              // `if (!BinIx[v]) BinIx[v] = i` never records i == 0
    }
}

unsigned grid_for(uint64_t n) {
    uint64_t b = (n + 255) / 256;
    return (unsigned)(b > (1u << 20) ? (1u << 20) : (b ? b : 1));
}

// stable LSD pass: order idx by key[idx] (bits [0, nbits))
template <typename IDX>
int sort_pass(const uint64_t *key_by_node, uint32_t nbits, IDX *&idx, IDX *&idx_alt, uint64_t *kg, uint64_t *kg_alt, uint64_t m,
              void *tmp, size_t tmp_bytes, hipStream_t st) {
    gather_k<IDX, uint64_t><<<grid_for(m), 256, 0, st>>>(key_by_node, idx, kg, m);
    hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, kg, kg_alt, idx, idx_alt, m, 0, nbits, st);
    if (e != hipSuccess) return (int)e;
    IDX *t = idx; idx = idx_alt; idx_alt = t;
    return (int)hipGetLastError();
}
// the last pass: by { bucket | the hash's low 8 bits } -- the bucket is monotone in the hash, but a bucket's up to 256 consecutive
// hash values need not ascend in their low 8 bits, and inside a bucket (and its overflow run) records ascend by their KEY, whose top
// field those 8 bits are
template <int W, typename IDX> __global__ void gather_bkey_k(const uint32_t *__restrict__ H, const uint8_t *__restrict__ O, const uint64_t *__restrict__ K1, const uint64_t *__restrict__ K2,
                                                             const IDX *__restrict__ idx, const uint64_t *__restrict__ regions, uint64_t *__restrict__ dst, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t b; uint32_t hl;
        const uint64_t t = idx[i];
        bucket_of(regions, H[t], O[t], entry_ext<W>(K1, K2, O, t), b, hl);
        dst[i] = (b << 8) | hl;
    }
}
template <int W, typename IDX>
int sort_pass_bucket(const uint32_t *H, const uint8_t *O, const uint64_t *K1, const uint64_t *K2, const uint64_t *regions, uint32_t nbits, IDX *&idx, IDX *&idx_alt, uint64_t *kg, uint64_t *kg_alt, uint64_t m,
                     void *tmp, size_t tmp_bytes, hipStream_t st) {
    gather_bkey_k<W, IDX><<<grid_for(m), 256, 0, st>>>(H, O, K1, K2, idx, regions, kg, m);
    hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, kg, kg_alt, idx, idx_alt, m, 0, nbits, st);
    if (e != hipSuccess) return (int)e;
    IDX *t = idx; idx = idx_alt; idx_alt = t;
    return (int)hipGetLastError();
}

// Parts of the hash range [lo, hi) (hi as uint64: the last part ends at 2^32).  The minimizer hash is a MINIMUM of 17 or
// 49 hashes, so it crowds towards 0: parts are cut at quantiles of a sample, rounded to table-slot boundaries.
struct part_bounds { uint64_t cut[65]; uint32_t n; };                 // part q = [cut[q], cut[q+1])
__global__ void part_hist_k(const uint32_t *__restrict__ H, uint64_t m, part_bounds pbnd, unsigned long long *__restrict__ counts) {
    __shared__ unsigned int s[64];
    if (threadIdx.x < 64) s[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t h = H[i];
        uint32_t q = 0;
        while (q + 1 < pbnd.n && h >= pbnd.cut[q + 1]) ++q;
        atomicAdd(&s[q], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64 && s[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)s[threadIdx.x]);
}
__global__ void sample_k(const uint32_t *__restrict__ H, uint64_t m, uint64_t stride, uint32_t n, uint32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = H[(uint64_t)i * stride < m ? (uint64_t)i * stride : m - 1];
}
constexpr uint64_t SEL_CHUNK = 1ull << 30;
struct in_part {
    uint64_t lo, hi;
    __device__ bool operator()(uint32_t h) const { return h >= lo && h < hi; }
};

// The whole MIN structure for nodes [c0, c0+m): table (2^B slots, pre-filled here) and MIN records.
// The (hash, position, rest) order is produced by stable LSD radix passes over node indices, which cost 32 bytes per
// node on top of the 12-20 bytes of keys; when that does not fit beside the image (trees of billions of nodes) the
// nodes are handled in 2^pb parts by the top bits of the hash -- parts are contiguous in the final order.
template <int W, int I, typename OFF, typename IDX>
int build_min(const OFF *coarse, const uint64_t *recs, uint64_t c0, const uint64_t m_nodes, uint64_t dup_cap, const uint64_t *regions, const uint64_t *h_regions, uint64_t n_buckets,
              uint32_t bw, uint64_t *table, uint64_t *mrecs, uint32_t *irreg, unsigned long long *d_overflow, uint64_t *n_min, int *views, hipStream_t st) {
    constexpr int EW = RecTraits<W, I>::EW, KW = RecTraits<W, I>::KW;
    const uint64_t nslots = n_buckets * (bw / EW);                           // entries, all flagged empty to begin with
    for (int x = 0; x < EW; ++x) fill_u64_k<<<grid_for(nslots), 256, 0, st>>>(table, nslots, x == KW ? MFLAG_EMPTY : 0ull, EW, x);
    *n_min = 0; *views = 1;
    if (!m_nodes) return (int)hipGetLastError();
    if (dup_cap > m_nodes) dup_cap = m_nodes;
    uint64_t m = m_nodes;                                                    // entries: the nodes, then the second views (known after assign_k)
    uint32_t *H = nullptr;
    uint8_t *O = nullptr;
    uint64_t *K1 = nullptr, *K2 = nullptr, *kg = nullptr, *kg2 = nullptr, *SRC = nullptr, *Bs = nullptr;
    unsigned long long *d_ndup = nullptr;
    IDX *idx = nullptr, *idx2 = nullptr;
    unsigned long long *d_counts = nullptr;
    void *tmp = nullptr;
    size_t t64 = 0, t32 = 0, tsel = 0;
    int rc = 0;
    uint32_t nparts = 1;
    part_bounds pbnd;
    uint64_t cap = 0, base = 0;
    unsigned long long h_counts[65] = {0}, sat_seen = 0;
    // longest run an overflow descriptor may describe (22-bit count, all ones = saturated); UTREE_BUCKET_RUN_MAX lowers it (tests)
    uint64_t run_max = (1ull << 22) - 2;
    { const char *e = getenv("UTREE_BUCKET_RUN_MAX"); if (e && atoll(e) > 0 && (uint64_t)atoll(e) < run_max) run_max = (uint64_t)atoll(e); }
    const bool chat = getenv("UTREE_TIMING") != nullptr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = (int)e_; goto done; } } while (0)
    CK(hipMalloc((void **)&H, (m_nodes + dup_cap) * 4)); CK(hipMalloc((void **)&O, m_nodes + dup_cap)); CK(hipMalloc((void **)&K1, (m_nodes + dup_cap) * 8));
    if (W == 16) CK(hipMalloc((void **)&K2, (m_nodes + dup_cap) * 8));
    CK(hipMalloc((void **)&SRC, (dup_cap ? dup_cap : 1) * 8)); CK(hipMalloc((void **)&d_ndup, 8));
    CK(hipMemsetAsync(d_ndup, 0, 8, st));
    if (chat) { CK(hipStreamSynchronize(st)); fprintf(stderr, "[utree_amd] image: table cleared, key buffers allocated\n"); }
    assign_k<W, I, OFF><<<grid_for((m_nodes + NODE_RUN - 1) / NODE_RUN), 256, 0, st>>>(coarse, recs, c0, m_nodes, dup_cap, H, O, K1, K2, SRC, d_ndup);
    {
        unsigned long long nd = 0;
        CK(hipMemcpyAsync(&nd, d_ndup, 8, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        if (nd > dup_cap) { *views = 0; nd = 0; }                            // too many: none is used, and the image says so
        m = m_nodes + nd;
        if (chat) fprintf(stderr, "[utree_amd] image: minimizers of %llu nodes assigned, %llu second views%s\n", (unsigned long long)m_nodes, nd, *views ? "" : " (dropped: beyond the build area)");
    }
    cap = m;
    {
        // smallest number of parts whose sort buffers fit what is left of the HBM, each sort below 2^31 items
        size_t free_b = 0, total_b = 0;
        CK(hipMemGetInfo(&free_b, &total_b));
        const char *env = getenv("UTREE_BUILD_PARTS");
        const uint64_t per = 2 * sizeof(IDX) + 16;
        for (nparts = 1; nparts < 64; ++nparts) {
            const uint64_t c = m / nparts + m / (nparts * 8) + 4096;                   // 12 % slack for uneven parts
            if (c < (1ull << 31) && c * per + (c >> 2) + ((uint64_t)1 << 30) <= free_b) break;
        }
        if (env && *env && atoi(env) >= 1 && atoi(env) <= 64) nparts = (uint32_t)atoi(env);
        pbnd.n = nparts; pbnd.cut[0] = 0; pbnd.cut[nparts] = 1ull << 32;
        if (nparts > 1) {
            constexpr uint32_t NS = 1u << 18;
            uint32_t *d_s = nullptr;
            uint32_t *h_s = (uint32_t *)malloc(sizeof(uint32_t) * NS);
            if (!h_s) { rc = (int)hipErrorOutOfMemory; goto done; }
            const uint32_t ns = m < NS ? (uint32_t)m : NS;
            if (hipMalloc((void **)&d_s, NS * 4) != hipSuccess) { free(h_s); rc = (int)hipErrorOutOfMemory; goto done; }
            sample_k<<<(ns + 255) / 256, 256, 0, st>>>(H, m, m / ns, ns, d_s);
            hipError_t e1 = hipMemcpyAsync(h_s, d_s, 4ull * ns, hipMemcpyDeviceToHost, st), e2 = hipStreamSynchronize(st);
            (void)hipFree(d_s);
            if (e1 != hipSuccess || e2 != hipSuccess) { free(h_s); rc = (int)(e1 != hipSuccess ? e1 : e2); goto done; }
            std::sort(h_s, h_s + ns);
            for (uint32_t q = 1; q < nparts; ++q) {
                // cuts fall on bucket boundaries: the smallest hash of the bucket that holds the sampled quantile
                const uint32_t hq = h_s[(uint64_t)ns * q / nparts];
                const uint64_t nb = h_regions[hq >> 24] & ((1ull << UTREE_REGION_NB_BITS) - 1);
                const uint64_t bl = bucket_in_region(hq, (uint32_t)nb);
                uint64_t c = ((uint64_t)(hq >> 24) << 24) | (((bl << 24) + nb - 1) / nb);
                pbnd.cut[q] = c < pbnd.cut[q - 1] ? pbnd.cut[q - 1] : c;
            }
            free(h_s);
            CK(hipMalloc((void **)&d_counts, 64 * 8));
            CK(hipMemsetAsync(d_counts, 0, 64 * 8, st));
            part_hist_k<<<(grid_for(m) > 4096 ? 4096 : grid_for(m)), 256, 0, st>>>(H, m, pbnd, d_counts);
            CK(hipMemcpyAsync(h_counts, d_counts, 64 * 8, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            cap = 0;
            for (uint32_t q = 0; q < nparts; ++q) if (h_counts[q] > cap) cap = h_counts[q];
            if (!cap) cap = 1;
            if (chat) {
                fprintf(stderr, "[utree_amd] image: %u parts, largest %llu nodes (free HBM %.1f GiB); cuts", nparts, (unsigned long long)cap, (double)free_b / 1073741824.0);
                for (uint32_t q = 1; q < nparts && q < 8; ++q) fprintf(stderr, " %llx", (unsigned long long)pbnd.cut[q]);
                fprintf(stderr, "; counts");
                for (uint32_t q = 0; q < nparts && q < 8; ++q) fprintf(stderr, " %llu", h_counts[q]);
                fprintf(stderr, "\n");
            }
        }
    }
    CK(hipMalloc((void **)&idx, cap * sizeof(IDX))); CK(hipMalloc((void **)&idx2, cap * sizeof(IDX)));
    CK(hipMalloc((void **)&kg, cap * 8)); CK(hipMalloc((void **)&kg2, cap * 8));
    CK(rocprim::radix_sort_pairs(nullptr, t64, kg, kg2, idx, idx2, cap, 0, 64, st));
    CK(rocprim::radix_sort_pairs(nullptr, t32, (uint32_t *)kg, (uint32_t *)kg2, idx, idx2, cap, 0, 32, st));
    if (t32 > t64) t64 = t32;
    if (nparts > 1) {
        auto flags = rocprim::make_transform_iterator(H, in_part{0, 1});
        CK(rocprim::select(nullptr, tsel, rocprim::counting_iterator<IDX>(0), flags, idx, d_counts, SEL_CHUNK < m ? SEL_CHUNK : m, st));
        if (tsel > t64) t64 = tsel;
    }
    CK(hipMalloc(&tmp, t64 ? t64 : 8));
    for (uint32_t q = 0; q < nparts; ++q) {
        uint64_t mq = m;
        if (nparts > 1) {
            mq = h_counts[q];
            if (!mq) continue;
            // node indices of this part, ascending; the selection runs over 2^30 nodes at a time
            uint64_t got = 0;
            for (uint64_t a = 0; a < m; a += SEL_CHUNK) {
                const uint64_t cnt = m - a < SEL_CHUNK ? m - a : SEL_CHUNK;
                auto flags = rocprim::make_transform_iterator(H + a, in_part{pbnd.cut[q], pbnd.cut[q + 1]});
                unsigned long long sel = 0;
                CK(rocprim::select(tmp, t64, rocprim::counting_iterator<IDX>((IDX)a), flags, idx + got, d_counts, cnt, st));
                CK(hipMemcpyAsync(&sel, d_counts, 8, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                got += sel;
                if (got > mq) { rc = (int)hipErrorUnknown; goto done; }
            }
            if (got != mq) { rc = (int)hipErrorUnknown; goto done; }
            if (chat) fprintf(stderr, "[utree_amd] image: part %u of %u, %llu nodes selected\n", q + 1, nparts, (unsigned long long)mq);
        } else iota_k<IDX><<<grid_for(m), 256, 0, st>>>(idx, m);
        // least significant key first; every pass is stable
        if (W == 16) {
            if ((rc = sort_pass<IDX>(K1, 64, idx, idx2, kg, kg2, mq, tmp, t64, st))) goto done;      // rest, low 64 bits
            if ((rc = sort_pass<IDX>(K2, 38, idx, idx2, kg, kg2, mq, tmp, t64, st))) goto done;      // position | rest high 32
        } else {
            if ((rc = sort_pass<IDX>(K1, 37, idx, idx2, kg, kg2, mq, tmp, t64, st))) goto done;      // position | rest
        }
        {
            uint32_t bbits = 1;
            while (bbits < 56 && (n_buckets >> bbits)) ++bbits;
            if ((rc = sort_pass_bucket<W, IDX>(H, O, K1, K2, regions, 8 + bbits, idx, idx2, kg, kg2, mq, tmp, t64, st))) goto done;   // bucket (pair, orientation), low hash bits
        }
        Bs = kg;
        // emit_k recomputes the buckets from H[idx], O[idx]: Bs becomes the array of bucket numbers in the final order
        emit_k<W, I, IDX><<<grid_for(mq), 256, 0, st>>>(recs, c0, idx, H, O, K1, K2, SRC, m_nodes, regions, mq, mrecs + base * EW, Bs);
        bucket_k<W, I><<<grid_for(mq), 256, 0, st>>>(Bs, mrecs + base * EW, base, mq, table, bw, run_max, d_overflow);
        CK(hipGetLastError());
        {
            unsigned long long sat = 0;
            CK(hipMemcpyAsync(&sat, d_overflow, 8, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            if (sat != sat_seen) {                  // this part has saturated buckets: their nodes' bins take the exact-probe path
                flag_saturated_k<W, I, OFF, IDX><<<grid_for(mq), 256, 0, st>>>(Bs, idx, mq, c0, SRC, m_nodes, coarse, table, bw, irreg, d_overflow);
                CK(hipGetLastError());
                sat_seen = sat;
            }
        }
        if (chat) { CK(hipStreamSynchronize(st)); fprintf(stderr, "[utree_amd] image: part %u sorted and emitted\n", q + 1); }
        base += mq;
    }
    CK(hipStreamSynchronize(st));
    if (base != m) rc = (int)hipErrorUnknown;
    *n_min = m;
done:
#undef CK
    (void)hipFree(H); (void)hipFree(K1); (void)hipFree(K2); (void)hipFree(idx); (void)hipFree(idx2); (void)hipFree(kg);
    (void)hipFree(kg2); (void)hipFree(tmp); (void)hipFree(d_counts); (void)hipFree(O); (void)hipFree(SRC); (void)hipFree(d_ndup);
    return rc;
}

// ---- PACKSIZE=16 (W = 4): the direct-address table.  A k-mer is a 32-bit word = 24-bit prefix (its bin) + 8-bit suffix, so every word's
// answer fits a table of 2^32 ranks.  direct_fill_k: one thread per 16 consecutive nodes -- the node of bin p with suffix s answers word
// p << 8 | s.  That IS the reference's answer wherever a bin is strictly ascending (any exact-match search finds the one record); a bin that
// is not (COMPRESS' first-bin quirk, duplicates, unsorted input) is flagged by validate_k, and direct_exact_k asks the reference's own probe
// sequence (exact_probe: itree.c:699-707) for each of the bin's 256 possible suffixes -- every bin of a table that is not monotone. ----
template <int I> __device__ __forceinline__ void direct_store(void *table, uint32_t word, uint32_t rank) {
    if constexpr (I == 2) ((uint16_t *)table)[word] = rank == INVALID ? (uint16_t)0xFFFFu : (uint16_t)rank;
    else ((uint32_t *)table)[word] = rank;
}
template <int I, typename OFF>
__global__ void direct_fill_k(const OFF *__restrict__ coarse, const uint64_t *__restrict__ recs, uint64_t c0, uint64_t m, void *__restrict__ table) {
    for (uint64_t t0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * NODE_RUN; t0 < m; t0 += (uint64_t)gridDim.x * blockDim.x * NODE_RUN) {
        uint64_t p = bin_of<OFF>(coarse, c0 + t0);
        const uint64_t t1 = t0 + NODE_RUN < m ? t0 + NODE_RUN : m;
        for (uint64_t t = t0; t < t1; ++t) {
            const uint64_t j = c0 + t;
            while ((uint64_t)coarse[p + 1] <= j) ++p;
            direct_store<I>(table, (uint32_t)(p << 8) | (uint32_t)(file_key<4, I>(recs, j).lo & 0xFFu), file_rank<4, I>(recs, j));
        }
    }
}
template <int I, typename OFF>
__global__ void direct_exact_k(const OFF *__restrict__ coarse, const uint64_t *__restrict__ recs, uint64_t n_nodes, const uint32_t *__restrict__ irreg,
                               void *__restrict__ table) {
    // one workgroup of 256 threads per bin and pass over the bins: thread s answers suffix s
    for (uint32_t p = blockIdx.x; p < (1u << 24); p += gridDim.x) {
        if (!((irreg[p >> 5] >> (p & 31)) & 1u)) continue;
        const uint64_t s = coarse[p], e = coarse[p + 1];
        uint32_t rank = INVALID;
        if (s < e && e <= n_nodes) {                                     // itree.c:726; a bin that leaves the node array counts as empty
            Key<4> q; q.hi = 0; q.lo = threadIdx.x;
            rank = exact_probe<4, I>(recs, s, e, q);
        }
        direct_store<I>(table, (p << 8) | threadIdx.x, rank);
    }
}

}  // namespace

extern "C" {

/* PACKSIZE=16: d_table (2^32 entries of I bytes, all ones = no node) from the FILE records; irregular bins as validate_k flagged them (all of
 * them when the bin table is not monotone: `generic`) */
int utk_build_direct(uint32_t I_, int off64, int generic, const void *d_coarse, const uint64_t *d_recs, uint64_t n_nodes, uint64_t c0, uint64_t m,
                     const uint32_t *d_irreg, uint64_t n_irregular, void *d_table, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(d_table, 0xFF, ((size_t)1 << 32) * I_, st);
    if (e != hipSuccess) return (int)e;
    if (I_ != 2 && I_ != 4) return (int)hipErrorInvalidValue;
#define GO(I__, OFF__) do { \
        if (!generic && m) direct_fill_k<I__, OFF__><<<grid_for((m + NODE_RUN - 1) / NODE_RUN), 256, 0, st>>>((const OFF__ *)d_coarse, d_recs, c0, m, d_table); \
        if (generic || n_irregular) direct_exact_k<I__, OFF__><<<dim3(generic ? 65536u : 4096u), 256, 0, st>>>((const OFF__ *)d_coarse, d_recs, n_nodes, d_irreg, d_table); \
    } while (0)
    if (I_ == 2) { if (off64) GO(2, uint64_t); else GO(2, uint32_t); }
    else { if (off64) GO(4, uint64_t); else GO(4, uint32_t); }
#undef GO
    return (int)hipGetLastError();
}

int utk_repack(uint32_t W_, uint32_t I_, const void *d_raw, uint64_t count, const uint32_t *d_ix2rank,
               uint32_t n_labels, uint64_t *d_recs, unsigned long long *d_invalid, void *stream) {
    if (!count) return 0;
    return dispatch_wi_all(W_, I_, [&](auto w, auto i) {
        repack_k<decltype(w)::value, decltype(i)::value><<<dim3(grid_for(count) > 65536 ? 65536 : grid_for(count)), dim3(256), 0, (hipStream_t)stream>>>(
            (const uint8_t *)d_raw, count, d_ix2rank, n_labels, d_recs, d_invalid);
    });
}

int utk_widen_binix(const void *d_raw_binix, uint32_t width, int off64, void *d_coarse, void *stream) {
    if (off64) widen_binix_k<uint64_t><<<dim3((UTREE_NUMBINS + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(d_raw_binix, width, (uint64_t *)d_coarse);
    else widen_binix_k<uint32_t><<<dim3((UTREE_NUMBINS + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(d_raw_binix, width, (uint32_t *)d_coarse);
    return (int)hipGetLastError();
}

int utk_validate(uint32_t W_, uint32_t I_, int off64, const void *d_coarse, const uint64_t *d_recs, uint64_t n_nodes,
                 uint32_t *d_irreg, unsigned long long *d_counters, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (off64) monotone_k<uint64_t><<<dim3((UTREE_NUMBINS + 255) / 256), dim3(256), 0, st>>>((const uint64_t *)d_coarse, n_nodes, d_counters);
    else monotone_k<uint32_t><<<dim3((UTREE_NUMBINS + 255) / 256), dim3(256), 0, st>>>((const uint32_t *)d_coarse, n_nodes, d_counters);
    const unsigned blocks = grid_for(n_nodes) > 65536 ? 65536 : grid_for(n_nodes);
    return dispatch_wi_all(W_, I_, [&](auto w, auto i) {
        constexpr int W = decltype(w)::value, I = decltype(i)::value;
        if (off64) validate_k<W, I, uint64_t><<<dim3(blocks), dim3(256), 0, st>>>((const uint64_t *)d_coarse, d_recs, n_nodes, d_irreg, d_counters);
        else validate_k<W, I, uint32_t><<<dim3(blocks), dim3(256), 0, st>>>((const uint32_t *)d_coarse, d_recs, n_nodes, d_irreg, d_counters);
    });
}

int utk_compress_chunk(uint32_t W_, uint32_t I_, const void *d_in, uint64_t first, uint64_t count, unsigned long long *d_first,
                       void *d_out, void *stream) {
    if (!count) return 0;
    return dispatch_wi_all(W_, I_, [&](auto w, auto i) {
        compress_chunk_k<decltype(w)::value, decltype(i)::value><<<dim3(grid_for(count) > 16384 ? 16384 : grid_for(count)), dim3(256), 0, (hipStream_t)stream>>>(
            (const uint8_t *)d_in, first, count, d_first, (uint8_t *)d_out);
    });
}

int utk_fill_recs_pad(uint64_t *d_recs_end, uint32_t words, void *stream) {
    fill_pad_k<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>(d_recs_end, words);
    return (int)hipGetLastError();
}

struct widen32 { __device__ uint64_t operator()(uint32_t v) const { return v & 0x7FFFFFFFu; } };   // (bit 31 of a count: the run gets a directory)
/* Pack the overflow runs to the front of d_mrecs (bucket order) and repoint the buckets' descriptors; *n_kept = records kept. */
int utk_compact_overflow(uint32_t W_, uint32_t I_, uint64_t *d_table, uint64_t n_buckets, uint32_t bw, uint64_t *d_mrecs, uint64_t *n_kept, int *chains, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    uint32_t *cnt = nullptr;
    uint64_t *prefix = nullptr, *packed = nullptr;
    void *tmp = nullptr;
    size_t tb = 0;
    int rc = 0;
    const uint32_t EW = utree_rec_words(W_, I_);
    const char *de = getenv("UTREE_OVF_DIR");                            /* =0: no position directories (A/B) */
    const int dir_on = !(de && atoi(de) == 0);
    const char *ce = getenv("UTREE_OVF_CHAINS");                         /* =1: k = 32: heavy runs as chains (device_common.hpp) instead of records behind a directory.
                                                                            Opt-in: a third of the bytes, parity-tested, and 25 % SLOWER on related genomes with a lane per window
                                                                            walking the chains */
    const int chains_on = W_ == 8 && dir_on && ce && atoi(ce) == 1;
    *n_kept = 0;
    *chains = chains_on;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = (int)e_; goto done; } } while (0)
    CK(hipMalloc((void **)&cnt, (n_buckets + 1) * 4));
    CK(hipMalloc((void **)&prefix, (n_buckets + 1) * 8));
    CK(hipMemsetAsync(cnt + n_buckets, 0, 4, st));
    {
        int drc = dispatch_wi(W_, I_, [&](auto w, auto i) {
            ovf_count_k<decltype(w)::value, decltype(i)::value><<<dim3(grid_for(n_buckets) > 65536 ? 65536 : grid_for(n_buckets)), dim3(256), 0, st>>>(d_table, n_buckets, bw, d_mrecs, dir_on, chains_on, cnt);
        });
        if (drc) { rc = drc; goto done; }
    }
    {
        auto in = rocprim::make_transform_iterator((const uint32_t *)cnt, widen32());
        CK(rocprim::exclusive_scan(nullptr, tb, in, prefix, (uint64_t)0, (size_t)n_buckets + 1, rocprim::plus<uint64_t>(), st));
        CK(hipMalloc(&tmp, tb ? tb : 8));
        CK(rocprim::exclusive_scan(tmp, tb, in, prefix, (uint64_t)0, (size_t)n_buckets + 1, rocprim::plus<uint64_t>(), st));
        unsigned long long total = 0;
        CK(hipMemcpyAsync(&total, prefix + n_buckets, 8, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        *n_kept = total;
        if (total) {
            CK(hipMalloc((void **)&packed, (size_t)total * EW * 8));
            int drc = dispatch_wi(W_, I_, [&](auto w, auto i) {
                ovf_move_k<decltype(w)::value, decltype(i)::value><<<dim3(grid_for(n_buckets) > 65536 ? 65536 : grid_for(n_buckets)), dim3(256), 0, st>>>(
                    d_table, n_buckets, bw, cnt, prefix, d_mrecs, chains_on, packed);
            });
            if (drc) { rc = drc; goto done; }
            CK(hipMemcpyAsync(d_mrecs, packed, (size_t)total * EW * 8, hipMemcpyDeviceToDevice, st));
        }
        CK(hipStreamSynchronize(st));
    }
done:
#undef CK
    (void)hipFree(cnt); (void)hipFree(prefix); (void)hipFree(packed); (void)hipFree(tmp);
    return rc;
}

/* nodes [c0, c0+m) = what the (monotone) bin table reaches.  d_overflow[0] += buckets whose run saturates the descriptor,
 * d_overflow[1] += bins newly flagged in d_irreg because of them. */
int utk_build_min(uint32_t W_, uint32_t I_, int off64, const void *d_coarse, const uint64_t *d_recs, uint64_t c0, uint64_t m, uint64_t dup_cap,
                  const uint64_t *d_regions, const uint64_t *h_regions, uint64_t n_buckets, uint32_t bw, uint64_t *d_table, uint64_t *d_mrecs, uint32_t *d_irreg,
                  unsigned long long *d_overflow, uint64_t *n_min, int *views, void *stream) {
    int rc = 0;
    int drc = dispatch_wi(W_, I_, [&](auto w, auto i) {
        constexpr int W = decltype(w)::value, I = decltype(i)::value;
        hipStream_t st = (hipStream_t)stream;
        const bool idx64 = m + dup_cap >= 0xFFFFFFFFull;
        if (off64 && idx64) rc = build_min<W, I, uint64_t, uint64_t>((const uint64_t *)d_coarse, d_recs, c0, m, dup_cap, d_regions, h_regions, n_buckets, bw, d_table, d_mrecs, d_irreg, d_overflow, n_min, views, st);
        else if (off64) rc = build_min<W, I, uint64_t, uint32_t>((const uint64_t *)d_coarse, d_recs, c0, m, dup_cap, d_regions, h_regions, n_buckets, bw, d_table, d_mrecs, d_irreg, d_overflow, n_min, views, st);
        else rc = build_min<W, I, uint32_t, uint32_t>((const uint32_t *)d_coarse, d_recs, c0, m, dup_cap, d_regions, h_regions, n_buckets, bw, d_table, d_mrecs, d_irreg, d_overflow, n_min, views, st);
    });
    return rc ? rc : drc;
}

}  // extern "C"
