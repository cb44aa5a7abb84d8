// kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the SEARCH_GG path and their C launchers.
//
// Written for wave64 / gfx950 only.  Integer, pointer-chasing work: no MFMA.  What bounds each kernel,
// its algorithmic bytes and the HBM layout are in DESIGN.md; the reference behaviour each kernel
// reproduces is cited as itree.c:line.
//
//   repack_k        on-disk SZ-byte records -> 8-byte-aligned {suffix,rank} records   (load time)
//   validate_k      per-bin ascending check, irregular-bin bitmap                      (load time)
//   build_fine_k    24+F-bit prefix index by lower_bound inside each 24-bit bin        (load time)
//   classify_short  one wavefront per read: stage through LDS, roll k-mers, look up, tally
//   classify_long   one workgroup per read for reads that do not fit a wavefront's LDS slice
//   vote_k          one lane per read: rank-wise LCA descent on the sorted unique label list
//   lookup_k        XT_getIX32 alone (tests, micro-benchmarks)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "utree_internal.h"

namespace {

constexpr uint64_t M40 = (1ull << 40) - 1;
constexpr uint32_t INVALID = 0xFFFFFFFFu;

// ------------------------------------------------------------------------------------------------
// keys, records, table entries
// ------------------------------------------------------------------------------------------------
// One format serves both the sorted record array and the direct-mapped prefix table.
// EW 8-byte words per record (a power of two, so an entry never straddles a 128-B line):
//   W=8, I=2 (EW 1): {flag8 | rank16 | suffix40}
//   W=8, I=4 (EW 2): {flag8 | 0 | suffix40} {rank32}
//   W=16,I=2 (EW 2): {suffix lo64} {flag8 | rank16 | suffix hi40}
//   W=16,I=4 (EW 4): {suffix lo64} {flag8 | 0 | suffix hi40} {rank32} {0}
// flag (top byte of the word holding the top suffix bits): 0 = a record, 1 = empty table slot,
// 2 = table slot that points at a run of >= 2 records: {2 | count16 | start40}.
template <int W> struct Key { uint64_t hi, lo; };   // hi = top 40 suffix bits for W=16, else 0

template <int W> __device__ __forceinline__ bool key_eq(const Key<W> &a, const Key<W> &b) {
    if constexpr (W == 16) return a.lo == b.lo && a.hi == b.hi; else return a.lo == b.lo;
}
template <int W> __device__ __forceinline__ bool key_lt(const Key<W> &a, const Key<W> &b) {
    if constexpr (W == 16) return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); else return a.lo < b.lo;
}
template <int W> __device__ __forceinline__ bool key_le(const Key<W> &a, const Key<W> &b) { return !key_lt<W>(b, a); }

template <int W, int I> struct RecTraits {
    static constexpr int EW = (W == 16 ? 2 : 1) * (I == 4 ? 2 : 1);
    static constexpr int KW = (W == 16 ? 1 : 0);            // word with the top suffix bits and the flag
};
constexpr uint64_t FLAG_EMPTY = 1ull << 56, FLAG_RUN = 2ull << 56;

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
template <int W, int I> __device__ __forceinline__ Key<W> entry_key(const Entry<W, I> &e) {
    Key<W> k;
    if constexpr (W == 16) { k.lo = e.w[0]; k.hi = e.w[1] & M40; } else { k.lo = e.w[0] & M40; k.hi = 0; }
    return k;
}
template <int W, int I> __device__ __forceinline__ uint32_t entry_rank(const Entry<W, I> &e) {
    if constexpr (I == 4) return (uint32_t)e.w[RecTraits<W, I>::KW + 1];
    else {
        uint32_t r = (uint32_t)(e.w[RecTraits<W, I>::KW] >> 40) & 0xFFFFu;
        return r == 0xFFFFu ? INVALID : r;
    }
}
template <int W, int I> __device__ __forceinline__ uint32_t entry_flag(const Entry<W, I> &e) {
    return (uint32_t)(e.w[RecTraits<W, I>::KW] >> 56);
}
template <int W, int I> __device__ __forceinline__ Key<W> load_key(const uint64_t *recs, uint64_t i) {
    constexpr int EW = RecTraits<W, I>::EW;
    Key<W> k;
    if constexpr (W == 16) { const ulonglong2 v = *(const ulonglong2 *)(recs + i * EW); k.lo = v.x; k.hi = v.y & M40; }
    else { k.lo = recs[i * EW] & M40; k.hi = 0; }
    return k;
}
template <int W, int I> __device__ __forceinline__ uint32_t load_rank(const uint64_t *recs, uint64_t i) {
    return entry_rank<W, I>(load_entry<W, I>(recs, i));
}

// The reference's probe sequence, verbatim in behaviour (itree.c:699-707, 728): p = first record of the
// bin; over the remaining e-s-1 records probe record w+1 past p; "<= query" moves p there.
template <int W, int I> __device__ uint32_t exact_probe(const uint64_t *recs, uint64_t s, uint64_t e, const Key<W> &q) {
    uint64_t p = s, size = e - s - 1;
    while (size) {
        uint64_t w = size >> 1;
        Key<W> k = load_key<W, I>(recs, p + w + 1);
        if (key_le<W>(k, q)) { p += w + 1; size -= w + 1; }
        else size = w;
    }
    Key<W> k = load_key<W, I>(recs, p);
    return key_eq<W>(k, q) ? load_rank<W, I>(recs, p) : INVALID;
}

// exact-match search in a strictly ascending run [lo, hi): equals the reference's result there
template <int W, int I> __device__ uint32_t sorted_find(const uint64_t *recs, uint64_t lo, uint64_t hi, const Key<W> &q) {
    while (lo < hi) {
        uint64_t mid = lo + ((hi - lo) >> 1);
        Key<W> k = load_key<W, I>(recs, mid);
        if (key_lt<W>(k, q)) lo = mid + 1;
        else if (key_eq<W>(k, q)) return load_rank<W, I>(recs, mid);
        else hi = mid;
    }
    return INVALID;
}

template <typename OFF> __device__ __forceinline__ void coarse_bin(const utk_image &im, uint32_t p, uint64_t &s, uint64_t &e) {
    const OFF *c = (const OFF *)im.coarse;
    s = c[p]; e = c[p + 1];                                      // itree.c:724
}

// Second half of a lookup, given the table entry of the word's (24+F)-bit prefix.
template <int W, int I, bool EXC, typename OFF>
__device__ __forceinline__ uint32_t resolve_entry(const utk_image &im, const Entry<W, I> &t, uint32_t p, const Key<W> &q) {
    if constexpr (EXC) {
        if ((im.irreg[p >> 5] >> (p & 31)) & 1u) {               // bin not strictly ascending (or generic mode)
            uint64_t s, e;
            coarse_bin<OFF>(im, p, s, e);
            if (s >= e || e > im.n_nodes) return INVALID;        // itree.c:726
            return exact_probe<W, I>(im.recs, s, e, q);
        }
    }
    const uint32_t flag = entry_flag<W, I>(t);
    if (flag == 0) return key_eq<W>(entry_key<W, I>(t), q) ? entry_rank<W, I>(t) : INVALID;   // the bin's only record
    if (flag == 1) return INVALID;                                                            // empty fine bin
    // a run of >= 2 records in the sorted array
    const uint64_t d = t.w[RecTraits<W, I>::KW];
    uint64_t start = d & M40, cnt = (d >> 40) & 0xFFFFu, end = start + cnt;
    if (cnt == 0xFFFFu) { uint64_t s; coarse_bin<OFF>(im, p, s, end); if (start < s) start = s; }   // saturated: search to the bin end
    const Entry<W, I> r0 = load_entry<W, I>(im.recs, start), r1 = load_entry<W, I>(im.recs, start + 1);
    const Key<W> k0 = entry_key<W, I>(r0), k1 = entry_key<W, I>(r1);
    if (key_eq<W>(k0, q)) return entry_rank<W, I>(r0);
    if (key_lt<W>(q, k0)) return INVALID;
    if (key_eq<W>(k1, q)) return entry_rank<W, I>(r1);
    if (end - start == 2 || key_lt<W>(q, k1)) return INVALID;
    return sorted_find<W, I>(im.recs, start + 2, end, q);
}

// split of the 2k-bit word khi:klo into 24-bit prefix (itree.c:684), suffix key and table slot
template <int W> __device__ __forceinline__ void split_word(const utk_image &im, uint64_t khi, uint64_t klo, uint32_t &p,
                                                            Key<W> &q, uint64_t &slot) {
    const uint64_t top = (W == 16) ? khi : klo;           // the 64 bits holding prefix (24) + first 40 suffix bits
    p = (uint32_t)(top >> 40);
    if constexpr (W == 16) { q.hi = khi & M40; q.lo = klo; } else { q.hi = 0; q.lo = klo & M40; }
    slot = top >> (40 - im.fine_bits);
}

// XT_getIX32 (itree.c:720-730) on the device image: ONE 128-B line for all but the few percent of words
// whose fine bin holds two or more records.
template <int W, int I, bool EXC, typename OFF>
__device__ __forceinline__ uint32_t lookup_word(const utk_image &im, uint64_t khi, uint64_t klo) {
    uint32_t p; Key<W> q; uint64_t slot;
    split_word<W>(im, khi, klo, p, q, slot);
    const Entry<W, I> t = load_slot<W, I>(im.table, slot);
    return resolve_entry<W, I, EXC, OFF>(im, t, p, q);
}

// ------------------------------------------------------------------------------------------------
// load-time kernels
// ------------------------------------------------------------------------------------------------
template <int W, int I>
__global__ void repack_k(const uint8_t *__restrict__ raw, uint64_t count, const uint32_t *__restrict__ ix2rank,
                         uint32_t n_labels, uint64_t *__restrict__ recs) {
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
        uint64_t *o = recs + i * EW;
        const uint64_t r16 = rank == INVALID ? 0xFFFFull : (uint64_t)rank;
        if constexpr (W == 8 && I == 2) { o[0] = lo | (r16 << 40); }
        else if constexpr (W == 8 && I == 4) { o[0] = lo; o[1] = rank; }
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

template <int W, int I, typename OFF>
__global__ void validate_k(const OFF *__restrict__ coarse, const uint64_t *__restrict__ recs, uint64_t n_nodes,
                           uint32_t *irreg, unsigned long long *counters) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= UTREE_NUMBINS - 1) return;
    uint64_t s = coarse[p], e = coarse[p + 1];
    if (s > e || e > n_nodes) { counters[1] = 1; return; }
    if (e - s < 2) return;
    Key<W> prev = load_key<W, I>(recs, s);
    for (uint64_t j = s + 1; j < e; ++j) {
        Key<W> cur = load_key<W, I>(recs, j);
        if (!key_lt<W>(prev, cur)) {
            atomicOr(&irreg[p >> 5], 1u << (p & 31));
            atomicAdd(&counters[0], 1ull);
            return;
        }
        prev = cur;
    }
}

// Direct-mapped table over (24+F)-bit prefixes: slot q describes the records of 24-bit bin q>>F whose next
// F suffix bits equal q & (2^F-1): none, exactly one (stored inline) or a run in the sorted array.
template <int W, int I, typename OFF>
__global__ void build_table_k(const OFF *__restrict__ coarse, const uint64_t *__restrict__ recs, uint32_t F,
                              uint64_t *__restrict__ table) {
    constexpr int EW = RecTraits<W, I>::EW, KW = RecTraits<W, I>::KW;
    const uint64_t nslots = 1ull << (24 + F);
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nslots; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p = q >> F, f = q & ((1ull << F) - 1);
        const uint64_t s = coarse[p], e = coarse[p + 1];
        uint64_t fs = s, fe = e;
        if (s < e) {
            // first record with (suffix >> (SUF-F)) >= f, and >= f+1
            auto lower = [&](uint64_t ff) {
                if (ff >> F) return e;
                Key<W> t;
                if constexpr (W == 16) { t.hi = ff << (40 - F); t.lo = 0; } else { t.hi = 0; t.lo = ff << (40 - F); }
                uint64_t lo = s, hi = e;
                while (lo < hi) {
                    uint64_t mid = lo + ((hi - lo) >> 1);
                    if (key_lt<W>(load_key<W, I>(recs, mid), t)) lo = mid + 1; else hi = mid;
                }
                return lo;
            };
            fs = f ? lower(f) : s;
            fe = lower(f + 1);
        } else fe = fs;
        uint64_t *o = table + q * EW;
        const uint64_t n = fe > fs ? fe - fs : 0;
#pragma unroll
        for (int j = 0; j < EW; ++j) o[j] = 0;
        if (n == 0) o[KW] = FLAG_EMPTY;
        else if (n == 1) {
#pragma unroll
            for (int j = 0; j < EW; ++j) o[j] = recs[fs * EW + j];       // flag byte of a record is 0
        } else o[KW] = FLAG_RUN | ((n < 0xFFFFu ? n : 0xFFFFull) << 40) | (fs & M40);
    }
}

__global__ void fill_pad_k(uint64_t *p, uint32_t words) {
    if (threadIdx.x < words) p[threadIdx.x] = ~0ull;
}

// ------------------------------------------------------------------------------------------------
// wave64 helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t lanes_below(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
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

__device__ __forceinline__ void store_result(utree_result *out, uint32_t label, int32_t cut, uint32_t found,
                                             uint32_t uix, uint32_t sl, uint32_t ol) {
    uint32_t *o = (uint32_t *)out;
    o[0] = label; o[1] = (uint32_t)cut; o[2] = found; o[3] = uix; o[4] = sl; o[5] = ol;
}

// ------------------------------------------------------------------------------------------------
// route_k: only launched when a batch may hold reads beyond SHORT_CAP.  Lists them for the mid-length pass
// (<= MID_CAP staged bases) or for classify_long_k, with one atomic per wavefront of 64 reads.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void route_k(const uint32_t *__restrict__ len, uint32_t n_reads, int do_rc, utk_workspace ws) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t total = 0;
    if (r < n_reads) total = do_rc ? 2 * (uint64_t)len[r] + 1 : len[r];
    const bool mid = total > UTREE_SHORT_CAP && total <= UTREE_MID_CAP, lng = total > UTREE_MID_CAP;
    const uint64_t mm = __ballot(mid), ml = __ballot(lng);
    const uint32_t lane = lane_id();
    if (mm) {
        unsigned long long b = 0;
        if (lane == 0) b = atomicAdd(&ws.cursors[UTREE_CUR_MID], (unsigned long long)__popcll(mm));
        b = __shfl(b, 0);
        if (mid) ws.mid_list[b + lanes_below(mm)] = r;
    }
    if (ml) {
        unsigned long long b = 0;
        if (lane == 0) b = atomicAdd(&ws.cursors[UTREE_CUR_LONG], (unsigned long long)__popcll(ml));
        b = __shfl(b, 0);
        if (lng) ws.long_list[b + lanes_below(ml)] = r;
    }
}

// ------------------------------------------------------------------------------------------------
// classify_short: one wavefront per read (reads whose staged length fits UTREE_SHORT_CAP bases)
// ------------------------------------------------------------------------------------------------
constexpr int SHORT_CAP = UTREE_SHORT_CAP;          // 150 bp + reverse strand fits
constexpr int MID_CAP = UTREE_MID_CAP;              // 1 kb + reverse strand fits; longer reads take classify_long_k
constexpr int WAVES_PER_BLOCK = 4;
constexpr uint32_t TALLY_CHUNK = UTREE_TALLY_CHUNK;
constexpr int32_t CUT_PENDING = -3;                 // result.cut while a read waits for vote_k

// 8 waves/SIMD for the default record format measured 4 % faster than 5 (r01: 325 vs 313 M reads/s) even with a
// few spilled dwords; the wider formats keep their registers.  (Voting inside this kernel, 64 parked reads per
// wave, was tried and measured slower at every occupancy: 295-331 vs 343 M reads/s with the separate vote_k.)
// CAP = staged bases a wavefront's LDS slice holds.  CAP = SHORT_CAP walks all reads of the batch and routes the
// longer ones to the mid / long lists; CAP = MID_CAP (LISTED) walks the mid list.  Its 37 KB of LDS per
// workgroup allow 4 workgroups per CU, so it may use 128 VGPRs.
template <int W, int I, bool EXC, typename OFF, int CAP, bool LISTED>
__global__ __launch_bounds__(256, CAP > SHORT_CAP ? 4 : ((W == 8 && I == 2) ? 8 : 5))
void classify_short_k(utk_image im, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ off,
                      const uint32_t *__restrict__ len, uint32_t n_reads, int do_rc, utree_result *__restrict__ out,
                      utk_workspace ws) {
    constexpr uint32_t K = 4 * W;
    constexpr int NCH = CAP / 64, NWORDS = CAP / 16 + 6;
    __shared__ uint32_t s_words[WAVES_PER_BLOCK][NWORDS];
    __shared__ uint64_t s_bad[WAVES_PER_BLOCK][NCH + 2];
    __shared__ uint32_t s_hits[WAVES_PER_BLOCK][CAP];
    const uint32_t lane = lane_id();
    const uint32_t wv = threadIdx.x >> 6;
    uint32_t *sw = s_words[wv];
    uint8_t *sb = (uint8_t *)sw;
    uint64_t *sbad = s_bad[wv];
    uint32_t *hits = s_hits[wv];
    const uint32_t wave_gid = blockIdx.x * WAVES_PER_BLOCK + wv, n_waves = gridDim.x * WAVES_PER_BLOCK;
    unsigned long long chunk_base = 0;
    uint32_t chunk_left = 0;

    const uint32_t n_items = LISTED ? (uint32_t)ws.cursors[UTREE_CUR_MID] : n_reads;
    for (uint32_t item = wave_gid; item < n_items; item += n_waves) {
        const uint32_t r = LISTED ? ws.mid_list[item] : item;
        const uint32_t L = len[r];
        const uint64_t o = off[r];
        const uint64_t total64 = do_rc ? 2 * (uint64_t)L + 1 : L;
        if (total64 > (uint64_t)CAP) {                     // route_k listed it for the mid-length pass or classify_long_k
            continue;
        }
        const uint32_t total = (uint32_t)total64;
        if (total < K) {                                   // no window: no hit, no output line
            if (lane == 0) store_result(&out[r], 0, -2, 0, 0, 0, 0);
            continue;
        }
        const uint32_t nwin = total - K + 1;
        const uint32_t nch = (total + 63) >> 6;
        // ---- stage: bytes -> 2-bit codes packed big-endian in LDS, bad-base ballots ----
        // byte loads are issued in groups (the whole read for CAP = SHORT_CAP) before the first one is consumed
        constexpr int G = NCH <= 5 ? NCH : 4;
        for (uint32_t c0 = 0; c0 < nch; c0 += G) {
            uint32_t raw[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint32_t j = (c0 + g) * 64 + lane;
                raw[g] = 0;
                if (j < L) raw[g] = bases[o + j];
                else if (j > L && j < total) raw[g] = 0x100u | bases[o + (2 * L - j)];      // reverse strand: complement
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint32_t c = c0 + g;
                if (c < nch) {
                    uint32_t code; bool bad;
                    base_code(raw[g] & 0xFFu, code, bad);
                    code ^= (raw[g] >> 8) * 3u;
                    uint64_t bm = __ballot(bad);
                    uint32_t t = (code << 2) | (uint32_t)__shfl_down((int)code, 1);
                    uint32_t u = (t << 4) | (uint32_t)__shfl_down((int)t, 2);
                    if ((lane & 3u) == 0) sb[(c * 16 + (lane >> 2)) ^ 3u] = (uint8_t)u;
                    if (lane == 0) sbad[c] = bm;
                }
            }
        }
        if (lane == 0) sbad[nch] = ~0ull;
        wave_lds_fence();
        // ---- windows: lane l takes windows l, l+64, ... (itree.c:906-933); two rounds of table loads in flight ----
        uint32_t F = 0;
        for (uint32_t it = 0; it * 64 < nwin; it += 2) {
            bool ok[2]; uint32_t p[2]; Key<W> q[2]; uint64_t slot[2]; Entry<W, I> t[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t i = (it + h) * 64 + lane;
                ok[h] = false;
                if ((it + h) * 64 < nwin) {
                    uint64_t b0 = sbad[it + h], b1 = sbad[it + h + 1];
                    uint64_t x = (b0 >> lane) | (lane ? (b1 << (64 - lane)) : 0ull);     // bad flags of bases i..i+63
                    ok[h] = (K == 64) ? (x == 0) : ((uint32_t)x == 0);
                }
                if (ok[h]) {
                    uint64_t khi, klo;
                    window_word<W>(sw, i, khi, klo);
                    split_word<W>(im, khi, klo, p[h], q[h], slot[h]);
                    t[h] = load_slot<W, I>(im.table, slot[h]);
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t rank = INVALID;
                if (ok[h]) rank = resolve_entry<W, I, EXC, OFF>(im, t[h], p[h], q[h]);
                bool hit = rank != INVALID;                     // itree.c:929-931
                uint64_t hm = __ballot(hit);
                if (hit) hits[F + lanes_below(hm)] = rank;
                F += (uint32_t)__popcll(hm);
            }
        }
        wave_lds_fence();
        // ---- tally (itree.c:1028-1040): unique labels with counts, ascending rank = strcmp order ----
        if (F == 0) { if (lane == 0) store_result(&out[r], 0, -2, 0, 0, 0, 0); continue; }
        const uint32_t h0 = hits[0];
        if (F == 1) { if (lane == 0) store_result(&out[r], im.rank2ix[h0], -2, 1, 1, 0, 0); continue; }
        // all hits equal?  (the common case for reads from one taxon)
        uint32_t mn = INVALID, mx = 0;
        for (uint32_t j = lane; j < F; j += 64) { uint32_t h = hits[j]; mn = h < mn ? h : mn; mx = h > mx ? h : mx; }
        mn = wave_min_u32(mn);
        mx = ~wave_min_u32(~mx);
        if (mn == mx) { if (lane == 0) store_result(&out[r], im.rank2ix[h0], -2, F, 1, 0, 0); continue; }
        // (rank,count) list space: every wave sub-allocates from chunks it reserves with ONE atomic per
        // TALLY_CHUNK entries (a per-read atomic on one address serialises the whole chip)
        if (F > chunk_left) {
            unsigned long long nb = 0;
            const uint32_t need = F > TALLY_CHUNK ? F : TALLY_CHUNK;
            if (lane == 0) nb = atomicAdd(&ws.cursors[0], (unsigned long long)need);
            chunk_base = __shfl(nb, 0);
            chunk_left = need;
        }
        const unsigned long long base = chunk_base;
        uint32_t uix = 0, cur = mn;
        for (;;) {
            uint32_t c = 0, nxt = INVALID;
            for (uint32_t j = lane; j < F; j += 64) {
                uint32_t h = hits[j];
                c += h == cur;
                if (h > cur && h < nxt) nxt = h;
            }
            c = wave_sum_u32(c);
            nxt = wave_min_u32(nxt);
            if (lane == 0) ws.tally[base + uix] = (uint64_t)cur | ((uint64_t)c << 32);
            ++uix;
            if (nxt == INVALID) break;
            cur = nxt;
        }
        chunk_base += uix; chunk_left -= uix;
        // vote_k finishes this read: cut = CUT_PENDING marks it, sl/ol carry the tally offset
        if (lane == 0) store_result(&out[r], im.rank2ix[h0], CUT_PENDING, F, uix, (uint32_t)base, (uint32_t)(base >> 32));
    }
}

// ------------------------------------------------------------------------------------------------
// classify_long: one workgroup per read, any length (itree.c:836: lines up to 16 MiB).  The read is walked in
// tiles staged through LDS (two table slots in flight per lane); hits go to a per-workgroup label histogram in
// HBM and set a bit in a touched-label bitmap; the bitmap is then swept in rank order, which yields the same
// sorted (rank,count) list the wave kernel emits, and only touched histogram entries are read and cleared.
// ------------------------------------------------------------------------------------------------
constexpr int LONG_TILE = 4096;                       // windows per tile
constexpr int LONG_THREADS = 256;
constexpr uint32_t LONG_LDS_BITWORDS = 2048;          // labels whose bitmap fits LDS (65 536); else a bitmap in HBM

template <int W, int I, bool EXC, typename OFF>
__global__ __launch_bounds__(LONG_THREADS) void classify_long_k(utk_image im, const uint8_t *__restrict__ bases,
                                                                const uint64_t *__restrict__ off,
                                                                const uint32_t *__restrict__ len, int do_rc,
                                                                utree_result *__restrict__ out, utk_workspace ws) {
    constexpr uint32_t K = 4 * W;
    constexpr uint32_t STAGE = LONG_TILE + 64;          // bases staged per tile (tile + K-1, rounded up)
    __shared__ uint32_t s_words[STAGE / 16 + 8];
    __shared__ uint64_t s_bad[STAGE / 64 + 2];
    __shared__ uint32_t s_scan[LONG_THREADS / 64 + 1];
    __shared__ uint32_t s_touch[LONG_LDS_BITWORDS];
    __shared__ unsigned long long s_base;
    __shared__ uint32_t s_first;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    uint8_t *sb = (uint8_t *)s_words;
    const uint32_t nl = im.n_labels, nbw = (nl + 31) >> 5;
    uint32_t *hist = ws.hist + (size_t)blockIdx.x * nl;                       // all zero between reads
    uint32_t *touch = nbw <= LONG_LDS_BITWORDS ? s_touch : ws.touch + (size_t)blockIdx.x * nbw;   // all zero between reads
    const uint32_t n_long = (uint32_t)ws.cursors[UTREE_CUR_LONG];
    if (nbw <= LONG_LDS_BITWORDS) for (uint32_t x = tid; x < nbw; x += LONG_THREADS) s_touch[x] = 0;
    __syncthreads();

    for (uint32_t li = blockIdx.x; li < n_long; li += gridDim.x) {
        const uint32_t r = ws.long_list[li];
        const uint64_t L64 = len[r];
        const uint64_t o = off[r];
        const uint64_t total = do_rc ? 2 * L64 + 1 : L64;
        const uint64_t nwin = total >= K ? total - K + 1 : 0;
        uint32_t my_hits = 0;
        if (tid == 0) s_first = INVALID;
        for (uint64_t w0 = 0; w0 < nwin; w0 += LONG_TILE) {
            // stage bases [w0, w0+STAGE)
            for (uint32_t c = wv; c < STAGE / 64; c += LONG_THREADS / 64) {
                uint64_t j = w0 + (uint64_t)c * 64 + lane;
                uint32_t code = 0; bool bad = true;
                if (j < L64) base_code(bases[o + j], code, bad);
                else if (j > L64 && j < total) { base_code(bases[o + (2 * L64 - j)], code, bad); code ^= 3u; }
                uint64_t bm = __ballot(bad);
                uint32_t t = (code << 2) | (uint32_t)__shfl_down((int)code, 1);
                uint32_t u = (t << 4) | (uint32_t)__shfl_down((int)t, 2);
                if ((lane & 3u) == 0) sb[(c * 16 + (lane >> 2)) ^ 3u] = (uint8_t)u;
                if (lane == 0) s_bad[c] = bm;
            }
            if (tid == 0) s_bad[STAGE / 64] = ~0ull;
            __syncthreads();
            const uint32_t tile_n = (uint32_t)(nwin - w0 < LONG_TILE ? nwin - w0 : LONG_TILE);
            for (uint32_t i0 = tid; i0 < tile_n; i0 += 2 * LONG_THREADS) {
                bool ok[2]; uint32_t p[2]; Key<W> q[2]; uint64_t slot[2]; Entry<W, I> t[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t i = i0 + h * LONG_THREADS;
                    ok[h] = false;
                    if (i < tile_n) {
                        const uint32_t ch = i >> 6, bit = i & 63u;
                        uint64_t x = (s_bad[ch] >> bit) | (bit ? (s_bad[ch + 1] << (64 - bit)) : 0ull);
                        ok[h] = (K == 64) ? (x == 0) : ((uint32_t)x == 0);
                    }
                    if (ok[h]) {
                        uint64_t khi, klo;
                        window_word<W>(s_words, i, khi, klo);
                        split_word<W>(im, khi, klo, p[h], q[h], slot[h]);
                        t[h] = load_slot<W, I>(im.table, slot[h]);
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (!ok[h]) continue;
                    const uint32_t rank = resolve_entry<W, I, EXC, OFF>(im, t[h], p[h], q[h]);
                    if (rank != INVALID) {                              // itree.c:929-931
                        atomicAdd(&hist[rank], 1u);
                        atomicOr(&touch[rank >> 5], 1u << (rank & 31u));
                        ++my_hits;
                    }
                }
            }
            __syncthreads();
        }
        // F = total hits
        uint32_t f = wave_sum_u32(my_hits);
        if (lane == 0) s_scan[wv] = f;
        __syncthreads();
        uint32_t F = 0;
        for (uint32_t w = 0; w < LONG_THREADS / 64; ++w) F += s_scan[w];
        __syncthreads();
        if (F == 0) { if (tid == 0) store_result(&out[r], 0, -2, 0, 0, 0, 0); continue; }
        // the other waves' global atomics are complete (barrier above waits vmcnt) and live in L2: read them there
        const uint32_t per = (nbw + LONG_THREADS - 1) / LONG_THREADS;
        const uint32_t lo = tid * per < nbw ? tid * per : nbw, hi = lo + per < nbw ? lo + per : nbw;
        uint32_t cnt = 0;
        for (uint32_t x = lo; x < hi; ++x) cnt += (uint32_t)__popc(__hip_atomic_load(&touch[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        uint32_t inc = cnt;                               // inclusive prefix over threads: wave scan + wave totals
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(inc, d); if (lane >= (uint32_t)d) inc += t; }
        if (lane == 63) s_scan[wv] = inc;
        __syncthreads();
        uint32_t wbase = 0, uix = 0;
        for (uint32_t w = 0; w < LONG_THREADS / 64; ++w) { if (w < wv) wbase += s_scan[w]; uix += s_scan[w]; }
        uint32_t pos = wbase + inc - cnt;
        if (tid == 0) s_base = uix > 1 ? atomicAdd(&ws.cursors[0], (unsigned long long)uix) : 0ull;
        __syncthreads();
        const unsigned long long base = s_base;
        for (uint32_t x = lo; x < hi; ++x) {
            uint32_t bits = __hip_atomic_load(&touch[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!bits) continue;
            __hip_atomic_store(&touch[x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (bits) {
                const uint32_t rk = x * 32 + (uint32_t)__builtin_ctz(bits);
                bits &= bits - 1;
                const uint32_t c = __hip_atomic_load(&hist[rk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&hist[rk], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (uix > 1) ws.tally[base + pos] = (uint64_t)rk | ((uint64_t)c << 32);
                else s_first = rk;
                ++pos;
            }
        }
        __syncthreads();
        if (tid == 0) {
            if (uix == 1) store_result(&out[r], im.rank2ix[s_first], -2, F, 1, 0, 0);
            else store_result(&out[r], 0, CUT_PENDING, F, uix, (uint32_t)base, (uint32_t)(base >> 32));
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// vote_k: greedy rank-wise descent with the 75 % cutoff (itree.c:1044-1088), one lane per read.
// T is the read's distinct labels in strcmp order (= ascending rank) with their hit counts.
// All state is 32-bit unsigned with wrap-around, like the reference's.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t cut_of(uint32_t x) {          // itree.c:1044,1046 (TAXACUT = 4)
    uint32_t c = x - x / 4;
    c += (x >> 1) >= c;
    return c;
}

__global__ __launch_bounds__(256) void vote_k(utk_image im, utree_result *__restrict__ out, utk_workspace ws, uint32_t n_reads) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t *res = (const uint32_t *)&out[r];
    if ((int32_t)res[1] != CUT_PENDING) return;            // finished by the classify kernel (0 or 1 distinct label)
    const uint32_t F = res[2], uix = res[3];
    const uint64_t *T = ws.tally + ((uint64_t)res[4] | ((uint64_t)res[5] << 32));
    const char *blob = im.label_blob;
    const uint32_t *loff = im.label_off;
#define T_RANK(z) ((uint32_t)T[z])
#define T_CNT(z) ((uint32_t)(T[z] >> 32))
    uint32_t cutoff = cut_of(F);
    uint32_t st = 0, ed = uix, dv = 0xFFFFFFFFu, orun = F, sl, ol;
    for (;;) {
        uint32_t run = T_CNT(st), td = dv;
        for (uint32_t z = st + 1; z < ed; ++z) {
            const char *s1 = blob + loff[T_RANK(z - 1)], *s2 = blob + loff[T_RANK(z)];
            const uint32_t nz = T_CNT(z);
            bool aside = false;
            if (!s1[dv + (dv == 0xFFFFFFFFu)]) aside = true;                 // itree.c:1052
            else {
                // itree.c:1060-1061: td = first index > dv where s1 ends, differs from s2, or is ';'.
                // Eight bytes per step (labels are NUL-terminated inside a zero-padded blob).
                td = dv + 1;
                for (;;) {
                    uint64_t x1, x2;
                    __builtin_memcpy(&x1, s1 + td, 8);
                    __builtin_memcpy(&x2, s2 + td, 8);
                    const uint64_t semi = x1 ^ 0x3B3B3B3B3B3B3B3Bull;
                    // bytes of interest -> 0x00 in one of the three words; classic exact zero-byte detector
                    const uint64_t d = x1 ^ x2;
                    uint64_t m = (((x1 - 0x0101010101010101ull) & ~x1) | ((semi - 0x0101010101010101ull) & ~semi)) & 0x8080808080808080ull;
                    // a difference: mark every non-zero byte of d (exact: no borrow tricks)
                    m |= (((d & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | d) & 0x8080808080808080ull;
                    if (m) { td += (uint32_t)(__builtin_ctzll(m) >> 3); break; }
                    td += 8;
                }
                const char a = s1[td], b = s2[td];
                if (a == b) { run += nz; continue; }                           // itree.c:1062
                const char before = td ? s1[td - 1] : 0;
                if ((!a && b == ';') || ((a == ';' || !a) && before == '_')) aside = true;   // 1063
                else if (run >= cutoff) { ed = z; break; }                     // itree.c:1068
                else { run = nz; st = z; continue; }                           // itree.c:1069
            }
            if (aside) {                                                       // 1053-1056 / 1064-1067
                run = nz; st = z;
                orun -= T_CNT(z - 1);
                cutoff = cut_of(orun);
            }
        }
        sl = run; ol = orun;                                                   // itree.c:1071
        if (run < cutoff) break;                                               // itree.c:1072
        if (st + 1 >= ed) {                                                    // itree.c:1073-1079
            if (T_CNT(ed - 1) >= cutoff) dv = 0xFFFFFFFEu;
            break;
        }
        orun = run; dv = td; cutoff = cut_of(run);                             // itree.c:1082-1085
    }
    const uint32_t rk = T_RANK(ed - 1);
    int32_t cut;
    if (dv == 0xFFFFFFFFu) cut = -1;                                           // itree.c:1087
    else if (dv == 0xFFFFFFFEu) cut = -2;
    else { uint32_t Ls = loff[rk + 1] - loff[rk] - 1; cut = (int32_t)(dv < Ls ? dv : Ls); }   // 1088
    store_result(&out[r], im.rank2ix[rk], cut, F, uix, sl, ol);
#undef T_RANK
#undef T_CNT
}

template <int W, int I, bool EXC, typename OFF>
__global__ void lookup_k(utk_image im, const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo, uint64_t n,
                         uint32_t *__restrict__ ix) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t rank = lookup_word<W, I, EXC, OFF>(im, W == 16 ? hi[i] : 0ull, lo[i]);
        ix[i] = rank == INVALID ? INVALID : im.rank2ix[rank];
    }
}

// ------------------------------------------------------------------------------------------------
// dispatch on (W, I, EXC, OFF64)
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

}  // namespace

extern "C" {

int utk_repack(uint32_t W_, uint32_t I_, const void *d_raw, uint64_t count, const uint32_t *d_ix2rank,
               uint32_t n_labels, uint64_t *d_recs, void *stream) {
    if (!count) return 0;
    uint64_t blocks = (count + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    return dispatch_wi(W_, I_, [&](auto w, auto i) {
        repack_k<decltype(w)::value, decltype(i)::value><<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(
            (const uint8_t *)d_raw, count, d_ix2rank, n_labels, d_recs);
    });
}

int utk_widen_binix(const void *d_raw_binix, uint32_t width, int off64, void *d_coarse, void *stream) {
    if (off64) widen_binix_k<uint64_t><<<dim3((UTREE_NUMBINS + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(d_raw_binix, width, (uint64_t *)d_coarse);
    else widen_binix_k<uint32_t><<<dim3((UTREE_NUMBINS + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(d_raw_binix, width, (uint32_t *)d_coarse);
    return (int)hipGetLastError();
}

int utk_validate(uint32_t W_, uint32_t I_, int off64, const void *d_coarse, const uint64_t *d_recs, uint64_t n_nodes,
                 uint32_t *d_irreg, unsigned long long *d_counters, void *stream) {
    return dispatch_wi(W_, I_, [&](auto w, auto i) {
        constexpr int W = decltype(w)::value, I = decltype(i)::value;
        if (off64) validate_k<W, I, uint64_t><<<dim3((UTREE_NUMBINS + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(
            (const uint64_t *)d_coarse, d_recs, n_nodes, d_irreg, d_counters);
        else validate_k<W, I, uint32_t><<<dim3((UTREE_NUMBINS + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(
            (const uint32_t *)d_coarse, d_recs, n_nodes, d_irreg, d_counters);
    });
}

int utk_build_table(uint32_t W_, uint32_t I_, int off64, const void *d_coarse, const uint64_t *d_recs,
                    uint32_t fine_bits, uint64_t *d_table, void *stream) {
    uint64_t nslots = 1ull << (24 + fine_bits);
    uint64_t blocks = (nslots + 255) / 256;
    if (blocks > (1u << 20)) blocks = 1u << 20;
    return dispatch_wi(W_, I_, [&](auto w, auto i) {
        constexpr int W = decltype(w)::value, I = decltype(i)::value;
        if (off64) build_table_k<W, I, uint64_t><<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(
            (const uint64_t *)d_coarse, d_recs, fine_bits, d_table);
        else build_table_k<W, I, uint32_t><<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(
            (const uint32_t *)d_coarse, d_recs, fine_bits, d_table);
    });
}

int utk_fill_recs_pad(uint64_t *d_recs_end, uint32_t words, void *stream) {
    fill_pad_k<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>(d_recs_end, words);
    return (int)hipGetLastError();
}

int utk_classify_short(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len,
                       uint32_t n_reads, int do_rc, utree_result *d_out, const utk_workspace *ws, int n_cu, void *stream) {
    if (!n_reads) return 0;
    uint32_t blocks = (n_reads + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    uint32_t cap = (uint32_t)n_cu * 8u;
    if (blocks > cap) blocks = cap;
    return dispatch_img(im, [&](auto w, auto i, auto exc, auto offt) {
        classify_short_k<decltype(w)::value, decltype(i)::value, decltype(exc)::value, decltype(offt), SHORT_CAP, false>
            <<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(*im, d_bases, d_off, d_len, n_reads, do_rc, d_out, *ws);
    });
}

int utk_route(const uint32_t *d_len, uint32_t n_reads, int do_rc, const utk_workspace *ws, void *stream) {
    if (!n_reads) return 0;
    route_k<<<dim3((n_reads + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(d_len, n_reads, do_rc, *ws);
    return (int)hipGetLastError();
}

// reads of 321..2112 staged bases, listed by route_k
int utk_classify_mid(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len,
                     uint32_t n_reads, int do_rc, utree_result *d_out, const utk_workspace *ws, int n_cu, void *stream) {
    if (!n_reads) return 0;
    uint32_t blocks = (n_reads + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    uint32_t cap = (uint32_t)n_cu * 4u;
    if (blocks > cap) blocks = cap;
    return dispatch_img(im, [&](auto w, auto i, auto exc, auto offt) {
        classify_short_k<decltype(w)::value, decltype(i)::value, decltype(exc)::value, decltype(offt), MID_CAP, true>
            <<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(*im, d_bases, d_off, d_len, n_reads, do_rc, d_out, *ws);
    });
}

int utk_classify_long(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len,
                      int do_rc, utree_result *d_out, const utk_workspace *ws, int n_cu, void *stream) {
    (void)n_cu;
    if (!ws->long_blocks) return 0;
    return dispatch_img(im, [&](auto w, auto i, auto exc, auto offt) {
        classify_long_k<decltype(w)::value, decltype(i)::value, decltype(exc)::value, decltype(offt)>
            <<<dim3(ws->long_blocks), dim3(LONG_THREADS), 0, (hipStream_t)stream>>>(*im, d_bases, d_off, d_len, do_rc, d_out, *ws);
    });
}

int utk_vote(const utk_image *im, utree_result *d_out, const utk_workspace *ws, uint32_t n_reads, void *stream) {
    if (!n_reads) return 0;
    vote_k<<<dim3((n_reads + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(*im, d_out, *ws, n_reads);
    return (int)hipGetLastError();
}

int utk_lookup(const utk_image *im, const uint64_t *d_hi, const uint64_t *d_lo, uint64_t n, uint32_t *d_ix, void *stream) {
    if (!n) return 0;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    return dispatch_img(im, [&](auto w, auto i, auto exc, auto offt) {
        lookup_k<decltype(w)::value, decltype(i)::value, decltype(exc)::value, decltype(offt)>
            <<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(*im, d_hi, d_lo, n, d_ix);
    });
}

const char *utk_classify_short_name(uint32_t W, uint32_t I) {
    (void)W; (void)I;
    return "classify_short_k";
}
const char *utk_classify_long_name(void) { return "classify_long_k"; }

}  // extern "C"
