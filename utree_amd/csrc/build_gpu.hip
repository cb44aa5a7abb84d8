// build_gpu.hip -- database BUILD on the device (`utree-build` / `utree-buildGG`: itree.c -D BUILD / BUILD_GG).
// gfx950 only.  SURVEY.md §8(f) rank 3.
//
// The reference inserts one k-mer at a time into a forest of pointer BSTs (itree.c:242-307, 437-473).  What reaches the
// `.ubt` is, per distinct k-mer, a left fold over the labels it was seen with IN INPUT ORDER:
//     BUILD:     all the same label -> that label, else BAD                                         (xeTreeU, 262-266)
//     BUILD_GG:  state <- state cut before the last ';' it shares with the new label; fewer than 2 shared ';' -> BAD;
//                every occurrence whose label differs from the current state cuts again              (xeTreeU_RF, 280-303)
// and the labels such cuts create are numbered in the order the cuts happen (addSampleUd, 297), interleaved with the
// references' own labels (addSampleU, 583).  So here:
//
//   count_k / emit_k   every position of every reference in parallel: complevel filter (itree.c:595-606), k-mer from 4-byte
//                      loads with byte-parallel base coding, ordered compaction -> (k-mer, input position | label);
//                      run once per range of k-mers (hist_k sizes the ranges)
//   rocprim            stable radix sort by k-mer (input is in input order, so equal k-mers stay in input order)
//   fold_k             one thread per distinct k-mer replays its occurrences; labels are ids into the "universe" of all
//                      ';'-prefixes of the references' labels (host-built), the cut is a table lookup; the first time
//                      (2*position+1) each universe label is produced is kept with atomicMin
//   rocprim::select    k-mers that are not BAD, still ascending = the reference's in-order dump (399-417)
//   pack_k             (word, ix) records as the file holds them + nodes per label
// The host (build.c) turns first-use times into label indices between fold and pack.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include "utree_internal.h"
#include "build_gpu.h"

namespace {

constexpr uint32_t ST_BAD = 0xFFFFFFFFu, ST_SKIP = 0xFFFFFFFEu;
constexpr int BLOCK = 256;

struct dev_in {
    const uint8_t *fa;
    const uint64_t *seq_off;
    const uint32_t *seq_len;
    const uint64_t *pos_prefix;      // [n_refs + 1]: positions (k-mer ends) before reference r
    const uint32_t *ref_u;           // universe id of reference r's label
    uint32_t n_refs, K, lv;
    uint64_t total_pos;
};

// four bases at once: 2-bit codes in the low bits of each byte, and one "bad" bit per byte (cf. itree.c:110-121)
__device__ __forceinline__ void code4(uint32_t w, uint32_t &x, uint32_t &badnib) {
    x = (w >> 1) & 0x03030303u;
    x ^= (x >> 1) & 0x01010101u;
    const uint32_t want = __builtin_amdgcn_perm(0u, 0x54474341u, x);       // the letter each code stands for: A C G T
    const uint32_t diff = want ^ (w & 0xDFDFDFDFu);
    const uint32_t t = (((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) & 0x80808080u;
    badnib = t;
}

// position g of the batch: which reference, and is there a k-mer to add (itree.c:593-617)?
template <int W>
__device__ __forceinline__ bool eval_pos(const dev_in &in, uint64_t g, uint64_t &hi, uint64_t &lo, uint32_t &r, bool &first) {
    uint32_t a = 0, b = in.n_refs;
    while (b - a > 1) { const uint32_t m = a + ((b - a) >> 1); if (in.pos_prefix[m] <= g) a = m; else b = m; }
    r = a;
    const uint64_t rel = g - in.pos_prefix[a];
    first = rel == 0;
    const uint8_t *src = in.fa + in.seq_off[a] + rel;                       // first base of (lv filter bases + k-mer)
    const uint32_t lv = in.lv;
    // the lv bases before the k-mer must read A, G, C, T (595-606)
    if (lv >= 1) { if ((src[0] & 0xDFu) != 'A') return false; }
    if (lv >= 2) { if ((src[1] & 0xDFu) != 'G') return false; }
    if (lv >= 3) { if ((src[2] & 0xDFu) != 'C') return false; }
    if (lv >= 4) { if ((src[3] & 0xDFu) != 'T') return false; }
    src += lv;
    uint32_t bad = 0;
    uint64_t acc[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < W; ++q) {                                           // W dwords = 4W bases
        uint32_t w, x, nb;
        __builtin_memcpy(&w, src + 4 * q, 4);
        code4(w, x, nb);
        bad |= nb;
        const uint64_t byte = (x * 0x40100401u) >> 24;                      // first base in the top two bits
        acc[q >> 3] = (acc[q >> 3] << 8) | byte;
    }
    if (bad) return false;                                                  // 609-612
    if (W == 8) { hi = 0; lo = acc[0]; } else { hi = acc[0]; lo = acc[1]; }
    return true;
}

// k-mers are handled in passes over ranges of their top 12 bits ("buckets"): each pass sorts fewer than 2^31 items and
// its buffers fit what is left of the HBM; the ranges are contiguous in the final order.
constexpr uint32_t BUCKET_BITS = 12, N_BUCKETS = 1u << BUCKET_BITS;
template <int W> __device__ __forceinline__ uint32_t bucket_of(uint64_t hi, uint64_t lo) {
    return (uint32_t)((W == 16 ? hi : lo) >> (64 - BUCKET_BITS));
}

template <int W>
__global__ __launch_bounds__(BLOCK) void hist_k(dev_in in, unsigned long long *__restrict__ hist) {
    __shared__ unsigned int s[N_BUCKETS];
    for (uint32_t i = threadIdx.x; i < N_BUCKETS; i += BLOCK) s[i] = 0;
    __syncthreads();
    for (uint64_t g = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; g < in.total_pos; g += (uint64_t)gridDim.x * BLOCK) {
        uint64_t hi, lo; uint32_t r; bool f;
        if (eval_pos<W>(in, g, hi, lo, r, f)) atomicAdd(&s[bucket_of<W>(hi, lo)], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < N_BUCKETS; i += BLOCK) if (s[i]) atomicAdd(&hist[i], (unsigned long long)s[i]);
}

template <int W>
__global__ __launch_bounds__(BLOCK) void count_k(dev_in in, uint32_t b_lo, uint32_t b_hi, uint32_t *__restrict__ block_counts) {
    const uint64_t g = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    bool ok = false;
    if (g < in.total_pos) {
        uint64_t hi, lo; uint32_t r; bool f;
        ok = eval_pos<W>(in, g, hi, lo, r, f);
        if (ok) { const uint32_t b = bucket_of<W>(hi, lo); ok = b >= b_lo && b < b_hi; }
    }
    __shared__ uint32_t s_cnt[BLOCK / 64];
    const uint64_t m = __ballot(ok);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

template <int W>
__global__ __launch_bounds__(BLOCK) void emit_k(dev_in in, uint32_t b_lo, uint32_t b_hi, const uint64_t *__restrict__ block_off,
                                                uint64_t *__restrict__ key_lo, uint64_t *__restrict__ key_hi,
                                                uint64_t *__restrict__ val) {
    const uint64_t g = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    bool ok = false, first = false;
    uint64_t hi = 0, lo = 0;
    uint32_t r = 0;
    if (g < in.total_pos) {
        ok = eval_pos<W>(in, g, hi, lo, r, first);
        if (ok) { const uint32_t b = bucket_of<W>(hi, lo); ok = b >= b_lo && b < b_hi; }
    }
    __shared__ uint32_t s_cnt[BLOCK / 64];
    const uint64_t m = __ballot(ok);
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    for (uint32_t w = 0; w < wv; ++w) before += s_cnt[w];
    const uint64_t at = block_off[blockIdx.x] + before;                     // this pass's occurrences before this position, in input order
    if (ok) {
        key_lo[at] = lo;
        if (W == 16) key_hi[at] = hi;
        val[at] = (g << 24) | in.ref_u[r];                                  // position in the input (the event clock) | label
    }
}

__global__ void iota_k(uint64_t *idx, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = i;
}
__global__ void gather64_k(const uint64_t *__restrict__ src, const uint64_t *__restrict__ idx, uint64_t *__restrict__ dst, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

struct universe {
    const char *blob; const uint64_t *off;         // NUL-terminated strings
    const uint32_t *trunc_off, *trunc_ids;         // trunc_ids[trunc_off[u] + m - 1] = u cut before its m-th ';'
};

// itree.c:286-295: ';' the two labels have in common before their first difference
__device__ __forceinline__ uint32_t shared_semicolons(const universe &U, uint32_t a, uint32_t b) {
    const char *x = U.blob + U.off[a], *y = U.blob + U.off[b];
    uint32_t n = 0;
    for (;; ++x, ++y) {
        const char c = *x;
        if (c != *y || !c) return n;
        n += c == ';';
    }
}

template <int W, bool GG>
__global__ __launch_bounds__(BLOCK) void fold_k(const uint64_t *__restrict__ key_lo, const uint64_t *__restrict__ key_hi,
                                                const uint64_t *__restrict__ val, uint64_t n, universe U,
                                                unsigned long long *__restrict__ first_time, uint32_t *__restrict__ state_out,
                                                unsigned long long *__restrict__ n_distinct) {
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint64_t lo = j < n ? key_lo[j] : 0, hi = (W == 16 && j < n) ? key_hi[j] : 0;
    const bool head = j < n && !(j && key_lo[j - 1] == lo && (W == 8 || key_hi[j - 1] == hi));
    {   // distinct k-mers of the pass (the reference's NumsInserted, itree.c:448,466): one atomic per wavefront
        const uint64_t hm = __ballot(head);
        if (hm && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(hm)) atomicAdd(n_distinct, (unsigned long long)__popcll(hm));
    }
    if (j >= n) return;
    if (!head) { state_out[j] = ST_SKIP; return; }
    uint32_t st = (uint32_t)(val[j] & 0xFFFFFFu);                           // first occurrence: its reference's label
    for (uint64_t t = j + 1; t < n && key_lo[t] == lo && (W == 8 || key_hi[t] == hi); ++t) {
        const uint64_t v = val[t];
        const uint32_t nu = (uint32_t)(v & 0xFFFFFFu);
        if (nu == st) continue;                                             // same label: nothing happens (262, 280)
        if (!GG) { st = ST_BAD; break; }                                    // 264; BAD is absorbing (263)
        const uint32_t m = shared_semicolons(U, st, nu);
        if (m < 2) { st = ST_BAD; break; }                                  // critical_cutoff (74, 295); absorbing (281)
        st = U.trunc_ids[U.trunc_off[st] + m - 1];                          // 296-301
        atomicMin(&first_time[st], (unsigned long long)(((v >> 24) << 1) | 1ull));   // event clock: 2*position+1
    }
    state_out[j] = st;
}

struct is_node { __device__ bool operator()(uint32_t s) const { return s < ST_SKIP; } };

template <int W, int I>
__global__ __launch_bounds__(BLOCK) void pack_k(const uint64_t *__restrict__ lo, const uint64_t *__restrict__ hi,
                                                const uint32_t *__restrict__ st, const uint32_t *__restrict__ ix_of_u, uint64_t first,
                                                uint64_t count, uint8_t *__restrict__ out, unsigned long long *__restrict__ per_label) {
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= count) return;
    const uint32_t ix = ix_of_u[st[first + i]];
    uint8_t *o = out + i * (uint64_t)(W + I);
    const uint64_t l = lo[first + i];
#pragma unroll
    for (int b = 0; b < 8; ++b) o[b] = (uint8_t)(l >> (8 * b));            // fwrite(&word) of a little-endian WTYPE (402-404)
    if (W == 16) {
        const uint64_t h = hi[first + i];
#pragma unroll
        for (int b = 0; b < 8; ++b) o[8 + b] = (uint8_t)(h >> (8 * b));
    }
#pragma unroll
    for (int b = 0; b < I; ++b) o[W + b] = (uint8_t)(ix >> (8 * b));
    atomicAdd(&per_label[ix], 1ull);                                        // ++cnts[tree->ix] (412)
}

#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[utree_amd] build: %s: %s\n", #x, hipGetErrorString(e_)); rc = UTREE_E_HIP; goto fail; } } while (0)

template <typename T> int dmalloc(T **p, uint64_t n) { return hipMalloc((void **)p, (n ? n : 1) * sizeof(T)) == hipSuccess ? 0 : 1; }

}  // namespace

struct build_seg { uint64_t n; uint64_t *lo, *hi; uint32_t *st; };
struct utk_build_state {
    int device, W, I;
    uint64_t n_occ, n_nodes;
    build_seg *seg; uint32_t n_seg;         // nodes (k-mer, universe label), ascending, one segment per pass
};

extern "C" {

void utk_build_free(utk_build_state *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (uint32_t i = 0; i < s->n_seg; ++i) {
        if (s->seg[i].lo) (void)hipFree(s->seg[i].lo);
        if (s->seg[i].hi) (void)hipFree(s->seg[i].hi);
        if (s->seg[i].st) (void)hipFree(s->seg[i].st);
    }
    free(s->seg);
    free(s);
}

/* one pass: the k-mers whose bucket is in [b_lo, b_hi): emit, stable sort, replay, keep what is not BAD */
static int build_pass(const utk_build_job *job, const dev_in &in, const universe &U, uint32_t b_lo, uint32_t b_hi, uint64_t n_expect,
                      uint32_t *d_counts, uint64_t *d_block_off, uint64_t n_blocks, unsigned long long *d_first,
                      unsigned long long *d_distinct, build_seg *seg) {
    int rc = UTREE_OK;
    const int W = (int)job->W;
    uint64_t *k_lo = nullptr, *k_hi = nullptr, *k_val = nullptr, *a_lo = nullptr, *a_val = nullptr, *idx = nullptr, *idx2 = nullptr, *g_hi = nullptr;
    uint32_t *d_state = nullptr;
    unsigned long long *d_nsel = nullptr;
    void *tmp = nullptr;
    size_t tb = 0;
    uint64_t n = 0;
    memset(seg, 0, sizeof *seg);
    if (W == 8) count_k<8><<<dim3((unsigned)n_blocks), dim3(BLOCK)>>>(in, b_lo, b_hi, d_counts);
    else count_k<16><<<dim3((unsigned)n_blocks), dim3(BLOCK)>>>(in, b_lo, b_hi, d_counts);
    HK(hipGetLastError());
    HK(rocprim::exclusive_scan(nullptr, tb, d_counts, d_block_off, (uint64_t)0, n_blocks + 1, rocprim::plus<uint64_t>()));
    HK(hipMalloc(&tmp, tb ? tb : 8));
    HK(rocprim::exclusive_scan(tmp, tb, d_counts, d_block_off, (uint64_t)0, n_blocks + 1, rocprim::plus<uint64_t>()));
    HK(hipMemcpy(&n, d_block_off + n_blocks, 8, hipMemcpyDeviceToHost));
    HK(hipFree(tmp)); tmp = nullptr;
    if (n != n_expect || n >= (1ull << 31)) { rc = UTREE_E_HIP; goto fail; }
    if (!n) return UTREE_OK;
    {
        const unsigned gb = (unsigned)((n + BLOCK - 1) / BLOCK);
        if (dmalloc(&k_lo, n) || dmalloc(&k_val, n) || (W == 16 && dmalloc(&k_hi, n))) { rc = UTREE_E_NOMEM; goto fail; }
        if (W == 8) emit_k<8><<<dim3((unsigned)n_blocks), dim3(BLOCK)>>>(in, b_lo, b_hi, d_block_off, k_lo, k_hi, k_val);
        else emit_k<16><<<dim3((unsigned)n_blocks), dim3(BLOCK)>>>(in, b_lo, b_hi, d_block_off, k_lo, k_hi, k_val);
        HK(hipGetLastError());
        // ---- stable sort by k-mer ----
        if (W == 8) {
            if (dmalloc(&a_lo, n) || dmalloc(&a_val, n)) { rc = UTREE_E_NOMEM; goto fail; }
            HK(rocprim::radix_sort_pairs(nullptr, tb, k_lo, a_lo, k_val, a_val, n, 0, 64));
            HK(hipMalloc(&tmp, tb ? tb : 8));
            HK(rocprim::radix_sort_pairs(tmp, tb, k_lo, a_lo, k_val, a_val, n, 0, 64));
            HK(hipDeviceSynchronize());
            HK(hipFree(tmp)); tmp = nullptr;
            HK(hipFree(k_lo)); HK(hipFree(k_val));
            k_lo = a_lo; k_val = a_val; a_lo = a_val = nullptr;
        } else {
            // least significant half first, then the most significant half (both stable); permutations carried as indices
            if (dmalloc(&idx, n) || dmalloc(&idx2, n) || dmalloc(&a_lo, n)) { rc = UTREE_E_NOMEM; goto fail; }
            iota_k<<<dim3(gb), dim3(BLOCK)>>>(idx, n);
            HK(rocprim::radix_sort_pairs(nullptr, tb, k_lo, a_lo, idx, idx2, n, 0, 64));
            HK(hipMalloc(&tmp, tb ? tb : 8));
            HK(rocprim::radix_sort_pairs(tmp, tb, k_lo, a_lo, idx, idx2, n, 0, 64));      /* a_lo sorted, idx2 = order by lo */
            if (dmalloc(&g_hi, n)) { rc = UTREE_E_NOMEM; goto fail; }
            gather64_k<<<dim3(gb), dim3(BLOCK)>>>(k_hi, idx2, g_hi, n);
            HK(rocprim::radix_sort_pairs(tmp, tb, g_hi, k_hi, idx2, idx, n, 0, 64));      /* k_hi sorted, idx = final order */
            gather64_k<<<dim3(gb), dim3(BLOCK)>>>(k_lo, idx, a_lo, n);
            gather64_k<<<dim3(gb), dim3(BLOCK)>>>(k_val, idx, g_hi, n);
            HK(hipDeviceSynchronize());
            HK(hipFree(tmp)); tmp = nullptr;
            HK(hipFree(k_lo)); HK(hipFree(k_val)); HK(hipFree(idx)); HK(hipFree(idx2));
            k_lo = a_lo; k_val = g_hi; a_lo = g_hi = idx = idx2 = nullptr;
        }
        // ---- replay each k-mer's occurrences ----
        if (dmalloc(&d_state, n)) { rc = UTREE_E_NOMEM; goto fail; }
        if (W == 8 && job->gg) fold_k<8, true><<<dim3(gb), dim3(BLOCK)>>>(k_lo, k_hi, k_val, n, U, d_first, d_state, d_distinct);
        else if (W == 8) fold_k<8, false><<<dim3(gb), dim3(BLOCK)>>>(k_lo, k_hi, k_val, n, U, d_first, d_state, d_distinct);
        else if (job->gg) fold_k<16, true><<<dim3(gb), dim3(BLOCK)>>>(k_lo, k_hi, k_val, n, U, d_first, d_state, d_distinct);
        else fold_k<16, false><<<dim3(gb), dim3(BLOCK)>>>(k_lo, k_hi, k_val, n, U, d_first, d_state, d_distinct);
        HK(hipGetLastError());
        HK(hipFree(k_val)); k_val = nullptr;
        // ---- keep what is not BAD, ascending ----
        {
            auto flags = rocprim::make_transform_iterator(d_state, is_node());
            size_t tb2 = 0;
            unsigned long long nn = 0;
            if (dmalloc(&d_nsel, 1)) { rc = UTREE_E_NOMEM; goto fail; }
            HK(rocprim::select(nullptr, tb, d_state, flags, d_state, d_nsel, n));
            HK(rocprim::select(nullptr, tb2, k_lo, flags, k_lo, d_nsel, n));
            if (tb2 > tb) tb = tb2;
            HK(hipMalloc(&tmp, tb ? tb : 8));
            if (dmalloc(&a_lo, n)) { rc = UTREE_E_NOMEM; goto fail; }
            HK(rocprim::select(tmp, tb, k_lo, flags, a_lo, d_nsel, n));
            HK(hipMemcpy(&nn, d_nsel, 8, hipMemcpyDeviceToHost));
            if (dmalloc(&seg->lo, nn) || dmalloc(&seg->st, nn) || (W == 16 && dmalloc(&seg->hi, nn))) { rc = UTREE_E_NOMEM; goto fail; }
            HK(hipMemcpy(seg->lo, a_lo, 8 * nn, hipMemcpyDeviceToDevice));
            if (W == 16) {
                HK(rocprim::select(tmp, tb, k_hi, flags, a_lo, d_nsel, n));
                HK(hipMemcpy(seg->hi, a_lo, 8 * nn, hipMemcpyDeviceToDevice));
            }
            HK(rocprim::select(tmp, tb, d_state, flags, seg->st, d_nsel, n));
            HK(hipDeviceSynchronize());
            seg->n = nn;
        }
    }
fail:
    (void)hipDeviceSynchronize();
    if (tmp) (void)hipFree(tmp);
    if (k_lo) (void)hipFree(k_lo);
    if (k_hi) (void)hipFree(k_hi);
    if (k_val) (void)hipFree(k_val);
    if (a_lo) (void)hipFree(a_lo);
    if (a_val) (void)hipFree(a_val);
    if (idx) (void)hipFree(idx);
    if (idx2) (void)hipFree(idx2);
    if (g_hi) (void)hipFree(g_hi);
    if (d_state) (void)hipFree(d_state);
    if (d_nsel) (void)hipFree(d_nsel);
    return rc;
}

int utk_build_phase1(const utk_build_job *job, utk_build_result *res, utk_build_state **out_state) {
    int rc = UTREE_OK;
    const int W = (int)job->W;
    uint8_t *d_fa = nullptr; uint64_t *d_seq_off = nullptr, *d_prefix = nullptr, *d_uoff = nullptr, *d_block_off = nullptr;
    uint32_t *d_seq_len = nullptr, *d_ref_u = nullptr, *d_toff = nullptr, *d_tids = nullptr, *d_counts = nullptr;
    char *d_ublob = nullptr;
    unsigned long long *d_first = nullptr, *d_hist = nullptr;
    uint64_t *h_prefix = nullptr;
    unsigned long long *h_hist = nullptr;
    utk_build_state *S = (utk_build_state *)calloc(1, sizeof *S);
    if (!S) return UTREE_E_NOMEM;
    S->device = job->device; S->W = W; S->I = (int)job->I;
    memset(res, 0, sizeof *res);
    HK(hipSetDevice(job->device));
    {
        const uint32_t K = 4u * job->W, kv = K - 1 + job->lv;
        h_prefix = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)job->n_refs + 1));
        h_hist = (unsigned long long *)calloc(N_BUCKETS, sizeof(unsigned long long));
        res->h_first_time = (uint64_t *)malloc(8 * ((size_t)job->n_u + 1));
        res->h_ref_time = (uint64_t *)malloc(8 * ((size_t)job->n_refs + 1));
        if (!h_prefix || !h_hist || !res->h_first_time || !res->h_ref_time) { rc = UTREE_E_NOMEM; goto fail; }
        uint64_t tot = 0;
        for (uint32_t r = 0; r < job->n_refs; ++r) {
            h_prefix[r] = tot; res->h_ref_time[r] = 2 * tot;            /* the reference's label event: before its first position */
            tot += job->h_seq_len[r] > kv ? job->h_seq_len[r] - kv : 0;
        }
        h_prefix[job->n_refs] = tot;
        res->total_pos = tot;
        memset(res->h_first_time, 0xFF, 8 * (size_t)job->n_u);
        if (!tot) goto done;                                             /* "Error: no k-mers." is the caller's call */
        if (tot >= (1ull << 40) || job->n_u >= (1u << 24)) { rc = UTREE_E_UNSUPPORTED; goto fail; }
        size_t free_b = 0, total_b = 0;
        HK(hipMemGetInfo(&free_b, &total_b));
        if (job->fa_bytes + (uint64_t)(1u << 30) > free_b) { rc = UTREE_E_NOMEM; goto fail; }
        if (dmalloc(&d_fa, job->fa_bytes + 8) || dmalloc(&d_seq_off, job->n_refs) || dmalloc(&d_seq_len, job->n_refs) ||
            dmalloc(&d_prefix, (uint64_t)job->n_refs + 1) || dmalloc(&d_ref_u, job->n_refs) ||
            dmalloc(&d_ublob, job->ublob_bytes + 8) || dmalloc(&d_uoff, job->n_u) || dmalloc(&d_toff, (uint64_t)job->n_u + 1) ||
            dmalloc(&d_tids, job->n_trunc) || dmalloc(&d_first, job->n_u) || dmalloc(&d_hist, N_BUCKETS)) { rc = UTREE_E_NOMEM; goto fail; }
        HK(hipMemcpy(d_fa, job->h_fa, job->fa_bytes, hipMemcpyHostToDevice));
        HK(hipMemcpy(d_seq_off, job->h_seq_off, 8ull * job->n_refs, hipMemcpyHostToDevice));
        HK(hipMemcpy(d_seq_len, job->h_seq_len, 4ull * job->n_refs, hipMemcpyHostToDevice));
        HK(hipMemcpy(d_prefix, h_prefix, 8ull * (job->n_refs + 1ull), hipMemcpyHostToDevice));
        HK(hipMemcpy(d_ref_u, job->h_ref_u, 4ull * job->n_refs, hipMemcpyHostToDevice));
        HK(hipMemcpy(d_ublob, job->h_ublob, job->ublob_bytes, hipMemcpyHostToDevice));
        HK(hipMemcpy(d_uoff, job->h_uoff, 8ull * job->n_u, hipMemcpyHostToDevice));
        HK(hipMemcpy(d_toff, job->h_trunc_off, 4ull * (job->n_u + 1ull), hipMemcpyHostToDevice));
        if (job->n_trunc) HK(hipMemcpy(d_tids, job->h_trunc_ids, 4ull * job->n_trunc, hipMemcpyHostToDevice));
        HK(hipMemset(d_first, 0xFF, 8ull * job->n_u));
        HK(hipMemset(d_hist, 0, 8ull * N_BUCKETS));
        dev_in in = {d_fa, d_seq_off, d_seq_len, d_prefix, d_ref_u, job->n_refs, K, job->lv, tot};
        universe U = {d_ublob, d_uoff, d_toff, d_tids};
        const uint64_t n_blocks = (tot + BLOCK - 1) / BLOCK;
        if (n_blocks > 0x7FFFFFFFull) { rc = UTREE_E_UNSUPPORTED; goto fail; }
        if (dmalloc(&d_counts, n_blocks + 1) || dmalloc(&d_block_off, n_blocks + 1)) { rc = UTREE_E_NOMEM; goto fail; }
        HK(hipMemset(d_counts, 0, 4 * (n_blocks + 1)));
        {
            const unsigned hb = n_blocks > 8192 ? 8192u : (unsigned)n_blocks;
            if (W == 8) hist_k<8><<<dim3(hb), dim3(BLOCK)>>>(in, d_hist); else hist_k<16><<<dim3(hb), dim3(BLOCK)>>>(in, d_hist);
            HK(hipGetLastError());
            HK(hipMemcpy(h_hist, d_hist, 8ull * N_BUCKETS, hipMemcpyDeviceToHost));
        }
        uint64_t n = 0;
        for (uint32_t b = 0; b < N_BUCKETS; ++b) n += h_hist[b];
        res->n_occ = S->n_occ = n;
        if (!n) goto done;
        HK(hipMemset(d_hist, 0, 8));                                     /* the histogram is on the host now: word 0 counts distinct k-mers */
        // passes: contiguous bucket ranges of at most `limit` occurrences
        HK(hipMemGetInfo(&free_b, &total_b));
        uint64_t limit = (free_b > ((uint64_t)2 << 30) ? free_b - ((uint64_t)2 << 30) : 0) / (W == 16 ? 72ull : 44ull);
        if (limit > (1ull << 31) - 1) limit = (1ull << 31) - 1;
        { const char *e = getenv("UTREE_BUILD_PASS_KMERS"); if (e && atoll(e) > 0 && (uint64_t)atoll(e) < limit) limit = (uint64_t)atoll(e); }
        S->seg = (build_seg *)calloc(N_BUCKETS, sizeof(build_seg));
        if (!S->seg) { rc = UTREE_E_NOMEM; goto fail; }
        for (uint32_t b = 0; b < N_BUCKETS;) {
            uint64_t cnt = h_hist[b];
            uint32_t e = b + 1;
            if (cnt > limit) { rc = n > limit && limit < (1ull << 31) - 1 ? UTREE_E_NOMEM : UTREE_E_UNSUPPORTED; goto fail; }   /* one bucket too large */
            while (e < N_BUCKETS && cnt + h_hist[e] <= limit) cnt += h_hist[e++];
            if (cnt) {
                rc = build_pass(job, in, U, b, e, cnt, d_counts, d_block_off, n_blocks, d_first, d_hist, &S->seg[S->n_seg]);
                if (rc) { S->n_seg++; goto fail; }
                S->n_nodes += S->seg[S->n_seg].n;
                S->n_seg++;
            }
            b = e;
        }
        res->n_nodes = S->n_nodes;
        res->n_passes = S->n_seg;
        { unsigned long long nd = 0; HK(hipMemcpy(&nd, d_hist, 8, hipMemcpyDeviceToHost)); res->n_distinct = nd; }
        HK(hipMemcpy(res->h_first_time, d_first, 8ull * job->n_u, hipMemcpyDeviceToHost));
    }
done:
    *out_state = S;
    S = nullptr;
fail:
    (void)hipDeviceSynchronize();
    if (d_fa) (void)hipFree(d_fa);
    if (d_seq_off) (void)hipFree(d_seq_off);
    if (d_seq_len) (void)hipFree(d_seq_len);
    if (d_prefix) (void)hipFree(d_prefix);
    if (d_ref_u) (void)hipFree(d_ref_u);
    if (d_ublob) (void)hipFree(d_ublob);
    if (d_uoff) (void)hipFree(d_uoff);
    if (d_toff) (void)hipFree(d_toff);
    if (d_tids) (void)hipFree(d_tids);
    if (d_first) (void)hipFree(d_first);
    if (d_hist) (void)hipFree(d_hist);
    if (d_counts) (void)hipFree(d_counts);
    if (d_block_off) (void)hipFree(d_block_off);
    free(h_prefix); free(h_hist);
    if (S) { utk_build_free(S); if (rc == UTREE_OK) rc = UTREE_E_HIP; }
    if (rc) { free(res->h_ref_time); free(res->h_first_time); res->h_ref_time = res->h_first_time = NULL; }
    return rc;
}

/* Records (word, ix) to fd in file order; h_per_label[ix] += nodes with that label. */
int utk_build_phase2(utk_build_state *S, const uint32_t *h_ix_of_u, uint32_t n_u, uint32_t n_labels, int fd, uint64_t *h_per_label) {
    int rc = UTREE_OK;
    uint32_t *d_ix = nullptr; unsigned long long *d_cnt = nullptr; uint8_t *d_out = nullptr; uint8_t *h_out = nullptr;
    const uint64_t CH = 32ull << 20, rec = (uint64_t)(S->W + S->I);
    HK(hipSetDevice(S->device));
    if (dmalloc(&d_ix, n_u) || dmalloc(&d_cnt, n_labels) || dmalloc(&d_out, CH * rec)) { rc = UTREE_E_NOMEM; goto fail; }
    HK(hipHostMalloc((void **)&h_out, CH * rec, hipHostMallocDefault));
    HK(hipMemcpy(d_ix, h_ix_of_u, 4ull * n_u, hipMemcpyHostToDevice));
    HK(hipMemset(d_cnt, 0, 8ull * (n_labels ? n_labels : 1)));
    for (uint32_t sg = 0; sg < S->n_seg; ++sg) {
        const build_seg &G = S->seg[sg];
        for (uint64_t first = 0; first < G.n; first += CH) {
            const uint64_t cnt = G.n - first < CH ? G.n - first : CH;
            const unsigned gb = (unsigned)((cnt + BLOCK - 1) / BLOCK);
            if (S->W == 8 && S->I == 2) pack_k<8, 2><<<dim3(gb), dim3(BLOCK)>>>(G.lo, G.hi, G.st, d_ix, first, cnt, d_out, d_cnt);
            else if (S->W == 8) pack_k<8, 4><<<dim3(gb), dim3(BLOCK)>>>(G.lo, G.hi, G.st, d_ix, first, cnt, d_out, d_cnt);
            else if (S->I == 2) pack_k<16, 2><<<dim3(gb), dim3(BLOCK)>>>(G.lo, G.hi, G.st, d_ix, first, cnt, d_out, d_cnt);
            else pack_k<16, 4><<<dim3(gb), dim3(BLOCK)>>>(G.lo, G.hi, G.st, d_ix, first, cnt, d_out, d_cnt);
            HK(hipGetLastError());
            HK(hipMemcpy(h_out, d_out, cnt * rec, hipMemcpyDeviceToHost));
            uint64_t done = 0;
            while (done < cnt * rec) {
                ssize_t w = write(fd, h_out + done, cnt * rec - done);
                if (w <= 0) { rc = UTREE_E_IO; goto fail; }
                done += (uint64_t)w;
            }
        }
    }
    HK(hipMemcpy(h_per_label, d_cnt, 8ull * n_labels, hipMemcpyDeviceToHost));
fail:
    if (d_ix) (void)hipFree(d_ix);
    if (d_cnt) (void)hipFree(d_cnt);
    if (d_out) (void)hipFree(d_out);
    if (h_out) (void)hipHostFree(h_out);
    return rc;
}

}  // extern "C"
