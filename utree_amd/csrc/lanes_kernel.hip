// lanes_kernel.hip -- the lane-per-read pass's small kernels (routing a mixed batch, pieces of long reads) and its launch entry points.
// The pass itself, classify_lanes_k, is in lanes_core.hpp; its instantiations are built by lanes_part.hip, once per (k, label width,
// bucket size).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "device_common.hpp"
#include "wave_common.hpp"

using namespace utk;

#ifndef UTREE_LANES_WAVES
#define UTREE_LANES_WAVES 4
#endif

extern "C" {
#define PART_DECL(W_, I_, NL_, BS_) int utk_lanes_part_##W_##_##I_##_##NL_##_##BS_(int segs, int irr, int mode, const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, \
                                                                              const uint32_t *d_len, uint32_t n_reads, int do_rc, utree_result *d_out, const utk_workspace *ws, \
                                                                              int n_cu, void *stream, uint32_t cls);
PART_DECL(8, 2, 1, 0) PART_DECL(8, 2, 1, 1) PART_DECL(8, 2, 2, 0) PART_DECL(8, 4, 1, 0) PART_DECL(8, 4, 1, 1) PART_DECL(8, 4, 2, 0)
PART_DECL(16, 2, 1, 0) PART_DECL(16, 2, 1, 1) PART_DECL(16, 2, 2, 0)
#undef PART_DECL
}

namespace {

constexpr uint32_t LCAP = UTREE_LANES_CAP;            // bases a lane's slot holds
constexpr int32_t CUT_PENDING = -3, RANK_PENDING = -4;   // as in kernels.hip (vote_k finishes those results)
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ void store_result(utree_result *out, uint32_t label, int32_t cut, uint32_t found, uint32_t uix, uint32_t sl, uint32_t ol) {
    uint32_t *o = (uint32_t *)out;
    o[0] = label; o[1] = (uint32_t)cut; o[2] = found; o[3] = uix; o[4] = sl; o[5] = ol;
}

}  // namespace

// both strands in one pass: 64-byte buckets whose image stores every k-mer under its mirrored view too (UTREE_LANES_BS=0: the reverse strand
// as a second sequence, as for the other images)
extern "C" int utk_lanes_both_strands(const utk_image *im, int do_rc) {
    const char *e = getenv("UTREE_LANES_BS");
    return do_rc && im->bucket_words == 8u && (im->flags & UTREE_F_STRAND_VIEWS) && !(e && atoi(e) == 0);
}

namespace {

// one instantiation of classify_lanes_k (lanes_part.hip), chosen by the image: k, label width, bucket size
int lanes_launch(const utk_image *im, int segs, int mode, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len, uint32_t n_reads, int do_rc,
                 utree_result *d_out, const utk_workspace *ws, int n_cu, void *stream, uint32_t cls) {
    const int irr = im->irr_n != 0, nl = im->bucket_words == 16u ? 2 : 1;
    const int bs = utk_lanes_both_strands(im, do_rc);
#define PART(W_, I_, NL_, BS_) return utk_lanes_part_##W_##_##I_##_##NL_##_##BS_(segs, irr, mode, im, d_bases, d_off, d_len, n_reads, do_rc, d_out, ws, n_cu, stream, cls)
    if (im->W == 16) { if (nl == 2) PART(16, 2, 2, 0); if (bs) PART(16, 2, 1, 1); PART(16, 2, 1, 0); }
    if (im->I == 4) { if (nl == 2) PART(8, 4, 2, 0); if (bs) PART(8, 4, 1, 1); PART(8, 4, 1, 0); }
    if (nl == 2) PART(8, 2, 2, 0);
    if (bs) PART(8, 2, 1, 1);
    PART(8, 2, 1, 0);
#undef PART
}

// ---- long reads in pieces (PIECE instantiations above) -------------------------------------------------------------------------
// pieces_k: every entry of ws.long_list is cut into pieces of PW windows; one atomic per read reserves its places on ws.pieces
__global__ __launch_bounds__(256) void pieces_k(const uint32_t *__restrict__ len, uint32_t K, uint32_t PW, utk_workspace ws) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n_long = (uint32_t)ws.cursors[UTREE_CUR_LONG];
    // (the capacities are bounds on what a batch of the stated total_bases / max_len can hold; a batch beyond them is reported, and every
    // consumer of the lists stops at the capacity: no kernel reads an entry that was not written or writes past a table)
    if (n_long > ws.n_long_cap) { if (i == 0) ws.cursors[UTREE_CUR_ERROR] = UTREE_DEVERR_LONG_CAP; n_long = ws.n_long_cap; }
    if (i >= n_long) return;
    const uint64_t L = len[ws.long_list[i]];
    const uint64_t nwin = L >= K ? L - K + 1 : 0;
    const uint32_t np = (uint32_t)((nwin + PW - 1) / PW);
    if (!np) return;
    const unsigned long long at = atomicAdd(&ws.cursors[UTREE_CUR_PIECES], (unsigned long long)np);
    if (at + np > ws.n_pieces_cap) ws.cursors[UTREE_CUR_ERROR] = UTREE_DEVERR_PIECES_CAP;
    for (uint32_t p = 0; p < np && at + p < ws.n_pieces_cap; ++p) ws.pieces[at + p] = ((uint64_t)i << 32) | p;   // (every entry below the capacity is some read's)
}

// finish_long_k: one wavefront per long read, a lane per slot of its table: the distinct labels in strcmp order (= ascending rank) with
// their counts go on the tally list like a wave-per-read kernel's, vote_k does the rest; a read the pieces pass could not finish goes
// on ws.long_left for classify_long_k
__global__ __launch_bounds__(256) void finish_long_k(utree_result *__restrict__ out, utk_workspace ws) {
    const uint32_t lane = lane_id();
    uint32_t n_long = (uint32_t)ws.cursors[UTREE_CUR_LONG];
    if (n_long > ws.n_long_cap) n_long = ws.n_long_cap;                  // (pieces_k has reported it)
    // (a resident grid walking the entries: a workgroup per four reads is bound by the rate at which workgroups are dispatched)
    for (uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6); i < n_long; i += gridDim.x * 4u) {
    const uint32_t r = ws.long_list[i];
    if (ws.lflag[i]) {
        if (lane == 0) ws.long_left[atomicAdd(&ws.cursors[UTREE_CUR_LEFT], 1ull)] = r;
        continue;
    }
    const uint32_t rank = ws.ltab_rank[(size_t)i * UTREE_LONG_SLOTS + lane], cnt = ws.ltab_cnt[(size_t)i * UTREE_LONG_SLOTS + lane];
    const bool valid = rank != 0xFFFFFFFFu;
    const uint64_t vm = ballot64(valid);
    const uint32_t nu = (uint32_t)__popcll(vm);
    const uint32_t F = wave_sum_u32(valid ? cnt : 0u);
    if (nu == 0) { if (lane == 0) store_result(&out[r], 0, -2, 0, 0, 0, 0); continue; }
    if (nu == 1) { if (valid) store_result(&out[r], rank, RANK_PENDING, F, 1, 0, 0); continue; }
    uint32_t place = 0;
    for (uint64_t m = vm; m; m &= m - 1) {
        const uint32_t other = (uint32_t)__builtin_amdgcn_readlane((int)rank, (int)__builtin_ctzll(m));
        place += other < rank ? 1u : 0u;
    }
    const unsigned long long base = ws.ltally_base + (unsigned long long)i * UTREE_LONG_SLOTS;   // the entry's own place: no reservation
    if (valid) ws.tally[base + place] = (uint64_t)rank | ((uint64_t)cnt << 32);
    if (lane == 0) store_result(&out[r], 0, CUT_PENDING, F, nu, (uint32_t)base, (uint32_t)(base >> 32));
    }
}

// what classify_long_k is left with: its list is ws.long_left from here on, its count takes the place of the long-read count
__global__ void left_count_k(utk_workspace ws) { ws.cursors[UTREE_CUR_LONG] = ws.cursors[UTREE_CUR_LEFT]; }

// lanes_route_k: a batch whose reads differ in length is split by the number of lanes a read needs -- class c: 2^c lanes, reads of
// up to cap[c] bases -- and every class goes through the instantiation for its size (MODE 1); reads beyond sixteen lanes, or whose
// two strands would not fit the wave-per-read pass that finishes what this pass leaves (mid_cap staged bases), are listed as long and
// go through in pieces.  One atomic per class and 64 reads.
struct lane_caps { uint32_t cap[5]; uint32_t mid_cap; };
constexpr uint32_t ROUTE_CHUNK = 4096;                                      // reads per workgroup of lanes_route_k
__global__ __launch_bounds__(256) void lanes_route_k(const uint32_t *__restrict__ len, uint32_t n_reads, int do_rc, utk_workspace ws, lane_caps lc) {
    // (a workgroup counts the classes of its 4096 reads in LDS, reserves its places on the lists with one atomic per class, then
    // writes: one atomic per wave and class on a single counter took 0.75 ms per 4 M reads of one class)
    __shared__ uint32_t s_cnt[6];
    __shared__ unsigned long long s_base[6];
    const uint32_t lane = lane_id(), lo = blockIdx.x * ROUTE_CHUNK;
    if (threadIdx.x < 6) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t cls[ROUTE_CHUNK / 256];
#pragma unroll
    for (uint32_t j = 0; j < ROUTE_CHUNK / 256; ++j) {
        const uint32_t r = lo + j * 256u + threadIdx.x;
        uint32_t c = 6;                                                     // no read
        if (r < n_reads) {
            const uint32_t L = len[r];
            const uint64_t staged = do_rc ? 2ull * L + 1 : L;
            c = L <= lc.cap[0] ? 0u : L <= lc.cap[1] ? 1u : L <= lc.cap[2] ? 2u : L <= lc.cap[3] ? 3u : L <= lc.cap[4] ? 4u : 5u;
            if (c && staged > lc.mid_cap) c = 5;
        }
        cls[j] = c;
#pragma unroll
        for (uint32_t v = 0; v < 6; ++v) {
            const uint64_t m = ballot64(c == v);
            if (m && lane == (uint32_t)__builtin_ctzll(m)) atomicAdd(&s_cnt[v], (uint32_t)__popcll(m));
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const uint32_t n = s_cnt[threadIdx.x];
        s_base[threadIdx.x] = n ? atomicAdd(&ws.cursors[threadIdx.x < 5 ? UTREE_CUR_CLASS + threadIdx.x : UTREE_CUR_LONG], (unsigned long long)n) : 0ull;
        s_cnt[threadIdx.x] = 0;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < ROUTE_CHUNK / 256; ++j) {
        const uint32_t r = lo + j * 256u + threadIdx.x, c = cls[j];
        // (the reads of one lane, class 0, are counted, not listed: their launch walks the batch itself and passes over the others)
#pragma unroll
        for (uint32_t v = 1; v < 6; ++v) {
            const uint64_t m = ballot64(c == v);
            if (!m) continue;
            const uint32_t lead = (uint32_t)__builtin_ctzll(m);
            uint32_t at = 0;
            if (lane == lead) at = atomicAdd(&s_cnt[v], (uint32_t)__popcll(m));
            at = (uint32_t)__builtin_amdgcn_readlane((int)at, (int)lead);
            if (c == v) {
                const unsigned long long p = s_base[v] + at + lanes_below(m);
                if (v < 5) ws.cls_list[(size_t)v * ws.cls_stride + p] = r; else ws.long_list[p] = r;
            }
        }
    }
}

}  // namespace

extern "C" {

// images this pass takes: k = 32 with u16 or u32 labels (u32: fewer than 2^19 - 1 of them, a tally slot keeps 19 bits of rank), k = 64 with
// u16 labels; a table with at most four irregular bins; buckets on 128-byte boundaries
int utk_lanes_image_ok(const utk_image *im) {
    const bool fmt = (im->W == 8 && (im->I == 2 || (im->I == 4 && im->n_labels < (1u << 19) - 1u))) || (im->W == 16 && im->I == 2);
    // (a node whose label index is beyond the label list can never be a hit, itree.c:929: the scan here does not test for that)
    return fmt && im->irr_n <= 4u && ((uintptr_t)im->table & (8u * im->bucket_words - 1u)) == 0 && !(im->flags & UTREE_F_INVALID_RANKS);
}

// reads of up to this many bases take 2^c lanes (c = 0 .. 4)
static uint32_t lanes_cap(const utk_image *im, int c) { return ((1u << c) - 1u) * (LCAP - 4u * im->W + 1u) + LCAP; }
uint32_t utk_lanes_max_len(const utk_image *im) { return lanes_cap(im, 4); }

static int lanes_pieces_launch(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len, int do_rc,
                               utree_result *d_out, const utk_workspace *ws, int n_cu, void *stream) {
    // (the grid of the pieces pass is the resident one: the number of pieces is on the device)
    return lanes_launch(im, 16, 2, d_bases, d_off, d_len, 0x40000000u, do_rc, d_out, ws, n_cu, stream, 0u);
}

int utk_classify_long_pieces(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len, int do_rc,
                             utree_result *d_out, const utk_workspace *ws, int n_cu, void *stream) {
    if (!ws->pieces || !ws->n_long_cap) return (int)hipErrorInvalidValue;
    const uint32_t K = 4u * im->W, PW = 16u * (LCAP - K + 1u);
    hipStream_t st = (hipStream_t)stream;
    pieces_k<<<dim3((ws->n_long_cap + 255) / 256), dim3(256), 0, st>>>(d_len, K, PW, *ws);
    int rc = lanes_pieces_launch(im, d_bases, d_off, d_len, do_rc, d_out, ws, n_cu, stream);
    if (rc) return rc;
    { uint32_t fb = (ws->n_long_cap + 3) / 4, fcap = (uint32_t)n_cu * 8u; finish_long_k<<<dim3(fb < fcap ? fb : fcap), dim3(256), 0, st>>>(d_out, *ws); }
    left_count_k<<<dim3(1), dim3(1), 0, st>>>(*ws);
    return (int)hipGetLastError();
}

// A batch this pass takes whole with ONE instantiation, no routing: every read within one lane (LCAP bases).
int utk_lanes_ok(const utk_image *im, uint32_t max_len, int do_rc) {
    (void)do_rc;
    return utk_lanes_image_ok(im) && max_len <= LCAP;
}

// lanes per read for a read of max_len bases (1, 2, 4, 8, 16)
int utk_lanes_segs(const utk_image *im, uint32_t max_len) {
    for (int c = 0; c < 4; ++c) if (max_len <= lanes_cap(im, c)) return 1 << c;
    return 16;
}

int utk_classify_lanes(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len, uint32_t n_reads,
                       uint32_t max_len, int do_rc, utree_result *d_out, const utk_workspace *ws, int n_cu, void *stream) {
    if (!n_reads) return 0;
    if (max_len > LCAP) return (int)hipErrorInvalidValue;
    return lanes_launch(im, 1, 0, d_bases, d_off, d_len, n_reads, do_rc, d_out, ws, n_cu, stream, 0u);
}

// A batch of mixed lengths: reads are listed by the lanes they need (lanes_route_k; longer ones on ws->long_list for the pieces pass) and
// every class that can hold a read of the batch runs as its own launch.
int utk_classify_lanes_mixed(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len, uint32_t n_reads,
                             uint32_t max_len, int do_rc, utree_result *d_out, const utk_workspace *ws, int n_cu, void *stream) {
    if (!n_reads) return 0;
    if (!ws->cls_list) return (int)hipErrorInvalidValue;
    lane_caps lc;
    for (int c = 0; c < 5; ++c) lc.cap[c] = lanes_cap(im, c);
    lc.mid_cap = UTREE_MID_CAP;
    lanes_route_k<<<dim3((n_reads + ROUTE_CHUNK - 1) / ROUTE_CHUNK), dim3(256), 0, (hipStream_t)stream>>>(d_len, n_reads, do_rc, *ws, lc);
    int rc = (int)hipGetLastError();
    int max_cls = 0;
    while (max_cls < 4 && max_len > lc.cap[max_cls]) ++max_cls;             // the largest class a read of the batch can need
    // ONE launch: its wavefronts work through the classes one after the other (classify_lanes_mixed_k; a launch per class, the shape until round 3,
    // cost +5.4 % instead of +4.9 % for 1 % of 300 bp reads among 16 M of 150 bp: profiles/r04/mixed_*_16M.json)
    if (!rc) rc = lanes_launch(im, 0, 3, d_bases, d_off, d_len, n_reads, do_rc, d_out, ws, n_cu, stream, (uint32_t)max_cls);
    return rc;
}


}  // extern "C"
