#pragma once
// lanes_core.hpp -- classify_lanes_k<W>: the 150-bp-class pass of the SEARCH_GG path with ONE LANE PER READ (gfx950 / wave64).
//
// classify_short_k (kernels.hip) gives a read to a wavefront: every step of the read's chain -- bytes, 2-bit stream, hashes,
// sliding minimum, bucket, scan, tally -- is a round trip of that one wave through LDS or memory, and 8 waves per SIMD is all the
// latency hiding there is.  Here a wavefront takes 64 reads at once and the chain is paid once per 64 reads:
//
//   phase 0  lane = read   the read's bytes -> 2-bit codes, packed big-endian, in the lane's LDS slot (itree.c:110-121)
//   phase A  lane = read   all lanes walk their read base by base IN STEP (position is wave-uniform): rolling 16-mer, hash,
//                          sliding minimum over the K-15 16-mers of a window in REGISTERS (van Herk / Gil-Werman: one suffix
//                          minimum per block of K-15 keys, one prefix minimum, one combine per window), and every maximal run
//                          of windows that share their minimizer is appended to ONE list for the wave (ballot + mbcnt)
//   phase B  lane = run    64 runs at a time: minimizer -> bucket (one 64-byte fetch per RUN, a quad of lanes per bucket, two
//                          more batches in flight); every ENTRY of the bucket names the one window it could be the record of
//                          (minimizer position minus the entry's position field): in the run? same outer bases? -> a hit for
//                          the run's read.  Buckets that continue in an overflow run: short runs are read whole, a lane per
//                          record, longer ones searched per window
//   phase C  lane = read   tally of the read's hits (itree.c:1028-1040): distinct labels ascending with counts, result record
//
// With RC (BS instantiations: an image with 64-byte buckets that carries UTREE_F_STRAND_VIEWS) both strands are served by ONE pass: minimizers
// are canonical (device_common.hpp), so the reverse complement of a window has the window's own minimizer run, mirrored -- the other bucket
// of the same pair (the other half of the line the quad fetches anyway), positions K-16-p, the outer bases reversed and complemented -- and
// every run's scan tests the entries of bucket o against its windows and those of bucket 1-o against their reverse complements.  Without BS
// the read's reverse complement takes phases A and B a second time in the same slot.  The hits a read gets are the same
// (window, record) pairs classify_short_k finds: it asks, per window, which entry of the minimizer's bucket carries the window's
// key {hash bits, position, outer bases}; this kernel asks, per entry, which window of the run has that key.  Reads this kernel
// does not finish -- two or more bases other than ACGTacgt, more distinct labels than LANES_TSLOTS, a wave whose run list is full -- go on the
// batch's list for the wave-per-read kernel (utk_classify_listed), which also remains the kernel for u32 labels, irregular tables,
// longer reads and databases whose reads hit in most windows (DESIGN.md sections 5c, 11).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "device_common.hpp"
#include "wave_common.hpp"

using namespace utk;

#ifndef UTREE_LANES_WAVES
#define UTREE_LANES_WAVES 4
#endif
#ifndef UTREE_LANES_WPS
#define UTREE_LANES_WPS 3                             /* wavefronts per SIMD the kernel is compiled for (LDS allows 12 per CU) */
#endif
#ifndef UTREE_LANES_WPS64
#define UTREE_LANES_WPS64 3                           /* ... its k = 64 instantiation (same-box: 2 per SIMD 1.66 ms, 3 per SIMD 1.41 ms per 4 M reads) */
#endif
#ifndef UTREE_LANES_TSLOTS
#define UTREE_LANES_TSLOTS 12                         /* distinct labels a read's tally table holds */
#endif

namespace {

#ifdef UTREE_LANES_TIMERS
__device__ unsigned long long g_lphase[8];
#define LT_DECL unsigned long long lt_t = __builtin_readcyclecounter(), lt_acc[6] = {0, 0, 0, 0, 0, 0};
#define LT(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); lt_acc[i] += n_ - lt_t; lt_t = n_; } while (0)
#else
#define LT_DECL
#define LT(i)
#endif

constexpr int LANES_WAVES = UTREE_LANES_WAVES;        // waves per workgroup
constexpr uint32_t LCAP = UTREE_LANES_CAP;            // bases a lane's slot holds
constexpr uint32_t NWORD = LCAP / 16;                 // stream words with data
static_assert(LCAP % 32 == 0, "slot geometry");
static_assert(LCAP + 16 + 48 <= 256, "positions are 8-bit fields of a run record");
// per k-mer length (W = 8: k = 32, W = 16: k = 64)
template <int W> struct Geo {
    static constexpr uint32_t K = 4 * W, OFS = UTREE_MIN_MARGIN(W), NB = K - 15 - 2 * OFS;   // bases of a window; its 16-mers that may be its minimizer: those at OFS .. OFS + NB - 1 (k = 64 leaves two at either end out: utree_internal.h)
    static constexpr uint32_t NA = W == 16 ? 3 : 1;                    // words of bases in front of / behind the minimizer a k-mer may reach
    static constexpr uint32_t FRONT = NA;                              // pad words in front of a lane's slot
    static constexpr uint32_t STRIDE = NWORD + FRONT + 2;              // ... and two behind; odd: lane slots fall on different banks
    static constexpr uint32_t RUNS = W == 16 ? 512 : 1280;             // runs per 64 lanes (150 bp reads: mean 14.1 per lane for k = 32, 4.4 for k = 64; pieces of long reads: 15.3 per lane of 129 windows -- 980 +- 20 per wavefront, and a wavefront whose list is full leaves its reads to the wave-per-read kernels: 1024 did that to 120 of 400 000 10 kb reads, 1.4 ms of classify_long_k per 17 ms step)
    static constexpr uint32_t DNONE = 63;                              // minimizer offset no run has (<= 48)
    static_assert((STRIDE & 1) == 1, "slot geometry");
};
constexpr uint32_t TSLOTS = UTREE_LANES_TSLOTS;       // distinct labels per read this kernel keeps count of
constexpr uint32_t T_EMPTY = 0xFFFFFFFFu;             // an unused slot of a tally table (rank 0xFFFF is no label's)
constexpr int32_t CUT_PENDING = -3, RANK_PENDING = -4;   // as in kernels.hip (vote_k finishes those results)

__device__ __forceinline__ uint32_t low_bytes(uint32_t n) { return n >= 4u ? 0xFFFFFFFFu : ((1u << (8u * n)) - 1u); }
// (the lane mask of a condition as the compare leaves it: __ballot() takes an int and costs a select and a second compare)
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { return ~wave_min_u32(~v); }

__device__ __forceinline__ void store_result(utree_result *out, uint32_t label, int32_t cut, uint32_t found, uint32_t uix, uint32_t sl, uint32_t ol) {
    uint32_t *o = (uint32_t *)out;
    o[0] = label; o[1] = (uint32_t)cut; o[2] = found; o[3] = uix; o[4] = sl; o[5] = ol;
}

typedef const __attribute__((address_space(1))) uint32_t *gptr32;


// k = 64: the 48 outer bases (96 bits, r0 the top word) of the window that starts `pos` bases (0..48) before its minimizer, from the
// 48 bases in front of the minimizer (A[0] first) and the 48 behind it: (A:B) >> 2 pos, low 96 bits -- a word shift by two
// selects, then one funnel shift per word
__device__ __forceinline__ void rest96(const uint32_t (&A)[3], const uint32_t (&B)[3], uint32_t pos, uint32_t &r0, uint32_t &r1, uint32_t &r2) {
    const uint32_t sh = 2u * pos, bs = sh & 31u;
    const bool w1 = (sh & 32u) != 0u, w2 = (sh & 64u) != 0u;
    const uint32_t v[6] = {A[0], A[1], A[2], B[0], B[1], B[2]};
    uint32_t y[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) y[t] = w1 ? (t ? v[t - 1] : 0u) : v[t];
    uint32_t x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = w2 ? y[i] : y[2 + i];
    r0 = __builtin_amdgcn_alignbit(x[0], x[1], bs); r1 = __builtin_amdgcn_alignbit(x[1], x[2], bs); r2 = __builtin_amdgcn_alignbit(x[2], x[3], bs);
}

// SEGS lanes per read (1, 2, 4, 8 or 16): lane s of a read takes its windows s SW .. s SW + SW - 1 (SW = LCAP - K + 1: what a slot's bases
// hold) and the LCAP bases from s SW on that they lie in -- reads of up to (SEGS - 1) SW + LCAP bases, 64 / SEGS of them per wavefront
// IRR: the table has a few irregular bins (COMPRESS' first-bin quirk): their words need the reference's own probe sequence
// (wave_common.hpp: resolve_bucket), so a read with a window in one of them is left to the wave-per-read kernel
// MODE 0: the items are the batch's reads 0 .. n_reads - 1 (with `cls` set: of a mixed batch, whose longer reads are on other launches'
// lists -- they are passed over here; the one-lane class needs no list and no trip for it).  MODE 1 (LISTED): the reads of one length class of a mixed batch, listed by
// lanes_route_k (ws.cls_list, class `cls`; their number is on the device).  MODE 2 (PIECE): the items are not reads but pieces of
// long reads (ws.pieces: SEGS SW windows of a read on ws.long_list each); a piece's tally goes into its read's table in HBM
// (ws.ltab_*), which finish_long_k turns into the read's result
// I: bytes of a label index (2, or 4 with k = 32: tally slots then keep 19 bits of rank and 13 of count, which bounds the labels of such
// an image -- utk_lanes_image_ok)
// NL: 16-byte loads a lane makes per bucket -- 1: the image has 64-byte buckets (a quad of lanes fetches one with one request), 2: 128-byte
// buckets (two requests, the two halves of one line)
// BS: both strands from one pass (NL = 1 only; do_rc is set): a quad fetches both buckets of a run's pair -- two requests, the two halves of one line
template <int W, int I, int SEGS, bool IRR, int MODE, int NL, bool BS>
__device__ __forceinline__ void lanes_body(const utk_image &im, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ off, const uint32_t *__restrict__ len,
                                           uint32_t n_reads, const int do_rc, utree_result *__restrict__ out, const utk_workspace &ws, const uint32_t cls,
                                           uint32_t *stream, uint32_t *runs, uint32_t *tab, uint32_t *full, uint32_t *pref, uint64_t *ost, const uint64_t *s_reg,
                                           const uint32_t wv) {
    // (the wavefront's LDS -- slots, run list, tally tables, the region table -- is the kernel's: LANES_PROLOGUE; one kernel may run this body for
    // several lanes-per-read classes one after the other, classify_lanes_mixed_k)
    constexpr bool PIECE = MODE == 2, LISTED = MODE == 1;
    static_assert(!BS || NL == 1, "both strands in one pass: 64-byte buckets");
    constexpr int NLX = BS ? 2 : NL;                                 // 16-byte loads per lane and run: the pipeline is that of the 128-byte buckets
    using G = Geo<W>;
    constexpr uint32_t K = G::K, NB = G::NB, OFS = G::OFS, NA = G::NA, STRIDE = G::STRIDE, FRONT = G::FRONT, RUNS_CAP = G::RUNS;
    constexpr uint32_t SW = LCAP - K + 1, RPW = 64 / SEGS;            // windows per lane; reads per wavefront
    constexpr uint32_t SEGSH = SEGS == 16 ? 4 : SEGS == 8 ? 3 : SEGS == 4 ? 2 : SEGS == 2 ? 1 : 0;
    constexpr uint32_t TSR = TSLOTS * SEGS < 48u ? TSLOTS * SEGS : 48u;   // tally slots per read: the table space of its lanes, up to 48 labels
    static_assert(SEGS == 1 || SEGS == 2 || SEGS == 4 || SEGS == 8 || SEGS == 16, "lanes per read");
    constexpr int EW = RecTraits<W, I>::EW, KW = RecTraits<W, I>::KW;
    static_assert(EW <= 2, "entries of 8 or 16 bytes");
    // a tally slot is {rank << CB | count}: a read of this kernel has at most 2 * 2064 hits (sixteen lanes, both strands)
    constexpr uint32_t CB = I == 2 ? 16u : 13u, CMASK = (1u << CB) - 1u;
    // a hit waiting for its push is {lane of its read << QS | rank}
    constexpr uint32_t QS = I == 2 ? 16u : 20u, QMASK = (1u << QS) - 1u;
    const uint64_t tbl = (uint64_t)(uintptr_t)im.table;
    // the bucket of a minimizer: the hash of its canonical form picks the pair, the orientation the bucket (device_common.hpp: bucket_of)
    // (k = 64: `ext`, the four bases around the minimizer in its canonical orientation, picks one of the pairs of the hash value's slot)
    // (the bucket's NUMBER: 32 bits -- 2^32 buckets are 256 GiB --, which is what a run keeps of its minimizer's hash between prepare and issue)
    auto bucket_index = [&](uint32_t h, uint32_t o, uint32_t ext) -> uint32_t {
        const uint64_t re = s_reg[h >> 24];
        uint32_t slot = __umulhi(h << 8, (uint32_t)re & ((1u << UTREE_REGION_NB_BITS) - 1u));
        if constexpr (W == 16) {
            const uint32_t sub = (uint32_t)(re >> UTREE_REGION_NB_BITS) & ((1u << UTREE_REGION_SUB_BITS) - 1u);
            slot = slot * sub + ((ext * sub) >> 8);
        }
        return 2u * ((uint32_t)(re >> UTREE_REGION_BASE_SHIFT) + slot) + o;
    };
    auto bucket_addr = [&](uint32_t h, uint32_t o, uint32_t ext) -> uint64_t { return tbl + ((uint64_t)bucket_index(h, o, ext) << (NL == 2 ? 7 : 6)); };
    // the four bases around a minimizer from the words in front of it and behind it (device_common.hpp: min_ext)
    auto ext_of = [&](const uint32_t (&A)[G::NA], const uint32_t (&B)[G::NA], uint32_t o) -> uint32_t {
        if constexpr (W == 16) return ext_canon(((A[G::NA - 1u] & 0xFu) << 4) | (B[0] >> 28), o); else return 0u;
    };
    const uint32_t lane = lane_id();
    uint32_t *sl = stream + lane * STRIDE + FRONT;                        // the lane's slot, word 0

    unsigned long long chunk_base = 0;
    uint32_t chunk_left = 0;
    const uint32_t wave_gid = blockIdx.x * LANES_WAVES + wv;
    if constexpr (PIECE) {                                              // (the items: pieces -- as many as were listed: pieces_k stops, and reports, at the capacity)
        const unsigned long long np = ws.cursors[UTREE_CUR_PIECES];
        n_reads = (uint32_t)(np < ws.n_pieces_cap ? np : ws.n_pieces_cap);
    }
    if constexpr (LISTED) n_reads = (uint32_t)ws.cursors[UTREE_CUR_CLASS + cls];
    if constexpr (MODE != 0) { if (!n_reads) return; }                  // (a length class no read of the batch fell into, no long read: nothing to set up)
    if constexpr (MODE == 0) { if (cls && !ws.cursors[UTREE_CUR_CLASS]) return; }   // (a mixed batch without a read of one lane)
    const uint32_t *const lst = LISTED ? ws.cls_list + (size_t)cls * ws.cls_stride : nullptr;
    unsigned long long *parts = ws.cursors + 64 + (PIECE ? 2 : LISTED ? 3 + cls : 0) * (UTREE_WORK_PARTS * UTREE_WORK_STRIDE);
    const uint32_t part_len = ((n_reads + UTREE_WORK_PARTS - 1) / UTREE_WORK_PARTS + 63u) / 64u * 64u;
    uint32_t part = wave_gid % UTREE_WORK_PARTS, parts_left = UTREE_WORK_PARTS;

    LT_DECL
    for (;;) {
        // ---- the next 64 / SEGS reads (one atomic per grab; a used-up part is left for good) ----
        uint32_t item = 0, item_end = 0;
        bool got = false;
        while (parts_left) {
            unsigned long long *ctr = parts + part * UTREE_WORK_STRIDE;
            const uint64_t lo = (uint64_t)part * part_len;
            const uint32_t avail = lo >= n_reads ? 0u : (uint32_t)(n_reads - lo < part_len ? n_reads - lo : part_len);
            unsigned long long g = ~0ull;
            if (lane == 0 && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < avail) g = atomicAdd(ctr, (unsigned long long)RPW);
            const uint32_t taken = uni32((uint32_t)(g > 0xFFFFFFFFull ? 0xFFFFFFFFull : g));
            if (taken < avail) {
                item = (uint32_t)lo + taken;
                item_end = taken + RPW < avail ? item + RPW : (uint32_t)lo + avail;
                got = true;
                break;
            }
            part = part + 1 == UTREE_WORK_PARTS ? 0 : part + 1;
            --parts_left;
        }
        if (!got) break;
        LT(0);

        // ---- phase 0: bytes -> packed 2-bit codes in the lane's slot ----
        // (L, o: the lane's piece of its read -- the whole read with one lane per read)
        uint32_t L = 0;
        uint64_t o = 0;
        bool exc = false, other = false;                                  // other: the read is another launch's (MODE 0 of a mixed batch)
        {
            const uint32_t r0 = item + (lane >> SEGSH), piece = lane & (SEGS - 1u);
            if (r0 < item_end) {
                if constexpr (PIECE) {
                    const uint64_t pr = ws.pieces[r0];                                // {entry of long_list, piece of that read}
                    const uint32_t rd = ws.long_list[(uint32_t)(pr >> 32)];
                    const uint64_t at = (uint64_t)(uint32_t)pr * (SEGS * SW) + piece * SW;   // the lane's first base in the read
                    const uint64_t Lr = len[rd];
                    if (Lr > at) { L = (uint32_t)(Lr - at < LCAP ? Lr - at : LCAP); o = off[rd] + at; }
                } else {
                    const uint32_t rd = LISTED ? lst[r0] : r0;
                    const uint32_t Lr = len[rd];
                    if (Lr > (SEGS - 1u) * SW + LCAP) { if (MODE == 0 && SEGS == 1 && cls) other = true; else exc = true; }   // longer than this instantiation holds
                    else if (Lr > piece * SW) { L = umin(LCAP, Lr - piece * SW); o = off[rd] + piece * SW; }
                }
            }
        }
        uint32_t badpos;                                                  // the read's one base that is not ACGTacgt, or far away
        {
            const uint64_t a = (uint64_t)(uintptr_t)bases + o;
            const uint32_t mf = (uint32_t)a & 3u;                         // the caller's buffer itself need not be aligned
            const gptr32 p = (gptr32)(a - mf);
            const uint32_t nd = L ? (L + mf + 3u) >> 2 : 0u;              // a dword is only touched when it holds a byte of the read
            uint32_t bad = 0, nbad = 0, badg = 0;
            // every dword the read touches, requested before the first is used: one memory round trip for the 64 reads (indices
            // past the read's last dword repeat it; a lane without a read loads nothing)
            constexpr uint32_t NRAW = LCAP / 4 + 1;
            uint32_t raw[NRAW];
#pragma unroll
            for (uint32_t d = 0; d < NRAW; ++d) raw[d] = 0u;
            if (nd) {
#pragma unroll
                for (uint32_t d = 0; d < NRAW; ++d) raw[d] = p[umin(d, nd - 1u)];
            }
            // (Lm = the read's length for the byte masks: when the grab's reads all have the same length -- the usual case -- it is
            // a scalar and the masks cost no vector instruction)
            auto convert = [&](const uint32_t Lm) {
#pragma unroll
                for (uint32_t c = 0; c < NWORD / 2; ++c) {                // 32 bases = 8 dwords = 2 stream words per step
                    uint32_t w[2] = {0u, 0u};
#pragma unroll
                    for (uint32_t g = 0; g < 8; ++g) {
                        const uint32_t gi = c * 8 + g;
                        const uint32_t fm = low_bytes(Lm > 4u * gi ? Lm - 4u * gi : 0u);                   // bytes of the read
                        const uint32_t word = __builtin_amdgcn_alignbyte(raw[gi + 1], raw[gi], mf) & fm;    // source bytes 4gi .. 4gi+3
                        const uint32_t g2 = (word >> 1) & 0x03030303u;
                        const uint32_t letter = __builtin_amdgcn_perm(0u, 0x47544341u, g2);                // 0 1 2 3 -> A C T G
                        const uint32_t z = (word & 0xDFDFDFDFu) ^ letter;                                  // non-zero byte = not ACGTacgt
                        const uint32_t nz = (((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;
                        bad |= nz & fm;                                                                    // one bad base per read is followed up: where it is
                        { const uint32_t pc = (uint32_t)__builtin_popcount(nz & fm); nbad += pc; badg += pc * gi; }
                        const uint32_t code = g2 ^ ((g2 >> 1) & 0x01010101u);                              // A=0 C=1 G=2 T=3
                        const uint32_t packed = (code * 0x40100401u) >> 24;                                // c0<<6 | c1<<4 | c2<<2 | c3
                        w[g >> 2] = (w[g >> 2] << 8) | packed;
                    }
                    sl[2 * c] = w[0]; sl[2 * c + 1] = w[1];
                }
            };
            const uint32_t L0 = uni32(L);
            if (ballot64(L != L0) == 0ull) convert(L0); else convert(L);
            if (nbad > 1u) { exc = true; L = 0; }                          // two or more: left to the wave-per-read kernel
            badpos = nbad == 1u ? 4u * badg + ((uint32_t)__builtin_ctz(bad) >> 3) : 0xFFFF0000u;
        }
        const uint32_t nwin = L >= K ? L - (K - 1u) : 0u;
#pragma unroll
        for (uint32_t i = 0; i < TSLOTS; ++i) tab[i * 64u + lane] = T_EMPTY;      // (the whole table space, whatever its division)
        if (lane < 2) full[lane] = 0;
        const uint32_t maxnwin = uni32(wave_max_u32(nwin));
        wave_lds_fence();
        LT(1);

        bool wave_full = false;
        // With RC the read's reverse complement is a second pass over the same slot (itree.c:891-898 appends it behind a separator
        // that no window spans: two independent sequences, one list of hits).
        for (int strand = 0; strand < (do_rc && !BS ? 2 : 1); ++strand) {
        if (strand) {
            // the slot's bases 0 .. 16 NWORD - 1 reversed and complemented word by word, then moved up by the 16 NWORD - L bases that
            // now lead; the bad base moves with them
            uint32_t rw[NWORD];
#pragma unroll
            for (uint32_t i = 0; i < NWORD; ++i) {
                const uint32_t y = __builtin_bitreverse32(sl[NWORD - 1 - i]);                 // groups reversed, the two bits of a group swapped
                rw[i] = ~(((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1));
            }
            wave_lds_fence();
#pragma unroll
            for (uint32_t i = 0; i < NWORD; ++i) sl[i] = rw[i];
            wave_lds_fence();
            const uint32_t lead = 16u * NWORD - L, lw = lead >> 4, lb = 2u * (lead & 15u);
#pragma unroll
            for (uint32_t i = 0; i < NWORD; ++i) {
                const uint32_t a = sl[i + lw], b = sl[i + lw + 1];                             // (past the slot for the read's last words: bases no window uses)
                rw[i] = lb ? __builtin_amdgcn_alignbit(a, b, 32u - lb) : a;
            }
            wave_lds_fence();
#pragma unroll
            for (uint32_t i = 0; i < NWORD; ++i) sl[i] = rw[i];
            if (badpos < L) badpos = L - 1u - badpos;
            wave_lds_fence();
        }
        // ---- phase A: minimizer runs of all 64 reads, in step ----
        uint32_t nruns = 0;
        if (maxnwin) {
            uint32_t A[NB];
            uint32_t m16 = sl[0], r16 = rc16(sl[0]);                        // the 16-mer at the walk's position and its reverse complement
            // (a 16-mer's rank is the hash of its canonical form, the smaller of the two: one xor, one funnel shift and one minimum per
            // position on top of the forward walk)
#define ROLL16(p_) { const uint32_t b_ = (sl[(p_) >> 4] >> (30u - 2u * ((p_) & 15u))) & 3u; m16 = (m16 << 2) | b_; r16 = __builtin_amdgcn_alignbit(b_ ^ 3u, r16, 2u); }
#define CKEY() (canon_key(m16, r16) & ~0x1FFu)
            // (a window's bin is its first 12 bases: the top 24 bits of the 16-mer it starts with -- every 16-mer passes here)
            auto irregular = [&](uint32_t m, uint32_t u) {                  // u: the 16-mer's first base = the window it starts
                if constexpr (IRR) {
                    const uint32_t pf = m >> 8;
                    if (u < nwin && (pf == im.irr_p[0] || pf == im.irr_p[1] || pf == im.irr_p[2] || pf == im.irr_p[3])) exc = true;
                }
            };
            irregular(m16, 0u);
#pragma unroll
            for (uint32_t p = 16; p < 16 + OFS; ++p) { ROLL16(p) irregular(m16, p - 15u); }   // (16-mers in front of the first candidate)
            A[0] = CKEY() | OFS;
#pragma unroll
            for (uint32_t p = 16 + OFS; p < K - OFS; ++p) {
                ROLL16(p)
                irregular(m16, p - 15u);
                A[p - 15 - OFS] = CKEY() | (p - 15u);
            }
            // A window that does not exist (beyond the read's last) or holds the bad base (itree.c:919-927) carries the key ~0 -- no
            // 16-mer's -- instead of its minimizer's: the run before it ends there like at any change of minimizer, and a "run" of
            // such windows is never listed.  `prev` starts as one, and the wave's last step is one for every lane: no flush case.
            uint32_t run_first = 0, prev = 0xFFFFFFFFu;
            const uint32_t lanec = lane << 24;
            // (CLEAN: no read of the grab has a bad base and all have one length -- the usual grab --: whether a window exists is then
            // a scalar question, three vector instructions per position less)
            const uint32_t nwin_u = uni32(nwin);
#define PHASE_A_BLOCKS(CLEAN_) \
            for (uint32_t b = 0;; ++b) { \
_Pragma("unroll") \
                for (int rr = (int)NB - 2; rr >= 0; --rr) A[rr] = umin(A[rr], A[rr + 1]); \
                uint32_t P = 0; \
                bool done = false; \
_Pragma("unroll") \
                for (uint32_t rr = 0; rr < NB; ++rr) { \
                    const uint32_t s = NB * b + rr; \
                    uint32_t wmin; \
                    if (rr == 0) wmin = A[0]; \
                    else { \
                        const uint32_t p = s + (K - 1u - OFS); \
                        ROLL16(p) \
                        irregular(m16, s + (OFS + NB - 1u)); \
                        const uint32_t k = CKEY() | (s + (OFS + NB - 1u)); \
                        const uint32_t Sr = A[rr]; \
                        A[rr - 1] = k; \
                        P = rr == 1 ? k : umin(P, k); \
                        wmin = umin(Sr, P); \
                    } \
                    uint32_t wv; \
                    if (CLEAN_) wv = s < nwin_u ? wmin : 0xFFFFFFFFu; \
                    else wv = ((badpos - s) > (K - 1u) && s < nwin) ? wmin : 0xFFFFFFFFu; \
                    const bool changed = wv != prev; \
                    const bool emit = changed && prev != 0xFFFFFFFFu; \
                    const uint64_t em = ballot64(emit); \
                    if (em) { \
                        const uint32_t idx = __builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, nruns)); \
 \
                        if (emit && idx < RUNS_CAP) runs[idx] = ((prev & 0xFFu) | (run_first << 8)) + ((s << 16) + lanec); \
                        nruns += (uint32_t)__popcll(em); \
                    } \
                    if (changed) run_first = s; \
                    prev = wv; \
                    if ((rr % 6u) == 0u && s >= maxnwin) { done = true; break; } \
                } \
                if (done) break; \
                const uint32_t p = NB * (b + 1u) + (K - 1u - OFS); \
                ROLL16(p) \
                irregular(m16, NB * (b + 1u) + (OFS + NB - 1u)); \
                A[NB - 1] = CKEY() | (NB * (b + 1u) + (OFS + NB - 1u)); \
            }
#ifdef UTREE_LANES_NOCLEAN
            const bool clean = false;
#else
            const bool clean = ballot64(badpos < 0xFFFF0000u || nwin != nwin_u) == 0ull;
#endif
            if (clean) { PHASE_A_BLOCKS(true) } else { PHASE_A_BLOCKS(false) }
#undef PHASE_A_BLOCKS
#undef ROLL16
#undef CKEY
            nruns = uni32(nruns);
        }
        if (nruns > RUNS_CAP) { wave_full = true; nruns = 0; }                         // every read of the grab goes on the list
        wave_lds_fence();
        LT(2);

        // ---- phase B: 64 runs at a time.  A lane works out its own run's bucket address and what a scan needs of the run; the 64
        // buckets then come in with FOUR loads of 16 buckets each, the four lanes of a quad fetching the four 16-byte quarters of
        // the bucket of the run ONE of them prepared (one request per bucket to the memory pipeline instead of four: a lane fetching
        // its whole bucket halves the chip's random-line rate, profiles/r01/membench_random_lines.txt), and every lane scans its
        // quarter -- two entries -- of its quad's four runs; what it needs of a run it gets from the lane that prepared it by a
        // quad broadcast (DPP), nothing goes through LDS.  Two more batches of 64 buckets are in flight meanwhile.
        // (a lane beyond the list repeats the list's last run -- its load stays inside the table -- as a run no entry can belong to)
        uint32_t n_ovf = 0, ovf_room = 0;                                  // ovf_room: run records phase B has read (prepared) so far
        // what a scan needs of a run: {its bucket, first window | minimizer position - first << 8 | windows - 1
        // << 14 | read << 20, the range of entry tags its windows have, the 16 (k = 64: 48) bases before the minimizer, the 16 (48) behind it}.
        // An entry's tag is {flag2 | hash low bits | minimizer position in the k-mer} (its high half-word; k = 32: times two, the zero bit
        // below the position included): the run's windows have the tags tlo .. tlo + span -- hash bits of the run's minimizer, positions
        // d - (windows - 1) .. d -- so that "a record of this run's minimizer, for one of its windows" is one subtract and one compare.
        // (pk bit 31: the orientation o of the run's minimizer in the read -- the bucket of the pair its windows' records are in.  BS: tr, RA, RB
        // are the same for the windows' reverse complements: they have the mirrored positions K-16-p in bucket 1-o, and their outer bases are
        // the reverse complement of the 2 (K-16) bases around the minimizer, read the same way)
        struct RunRegs { uint32_t b, pk, t, A[NA], B[NA], tr, RA[NA], RB[NA]; };   // b: the run's bucket (its number)
        constexpr uint32_t TSH = W == 8 ? 1u : 0u, PB = W == 8 ? 5u : 6u;             // tag scale; bits of the position field
        constexpr uint32_t KM = K - 16u;                                              // the largest minimizer position
        auto rev_context = [&](const uint32_t (&A)[NA], const uint32_t (&B)[NA], uint32_t (&RA)[NA], uint32_t (&RB)[NA]) {
#pragma unroll
            for (uint32_t i = 0; i < NA; ++i) { RA[i] = rc16(B[NA - 1u - i]); RB[i] = rc16(A[NA - 1u - i]); }
        };
        // a run's context from its record: the words around the minimizer come from the slot of the run's read
        auto context = [&](uint32_t q, uint32_t ustar, uint32_t &m, uint32_t (&A)[NA], uint32_t (&B)[NA]) {
            const uint32_t *sq = stream + q * STRIDE + FRONT + (ustar >> 4);
            const uint32_t rr = ustar & 15u, sh = (32u - 2u * rr) & 31u;
            uint32_t w[2 * NA + 2];
#pragma unroll
            for (uint32_t t = 0; t < 2 * NA + 2; ++t) w[t] = sq[(int)t - (int)NA];
            // 16 bases from base ustar + 16 t: words w[NA + t], w[NA + t + 1]
#pragma unroll
            for (uint32_t i = 0; i < NA; ++i) {
                A[i] = rr ? __builtin_amdgcn_alignbit(w[i], w[i + 1], sh) : w[i];                         // t = i - NA: in front of the minimizer
                B[i] = rr ? __builtin_amdgcn_alignbit(w[NA + 1 + i], w[NA + 2 + i], sh) : w[NA + 1 + i];   // t = 1 + i: behind it
            }
            m = rr ? __builtin_amdgcn_alignbit(w[NA], w[NA + 1], sh) : w[NA];
        };
        auto prepare = [&](uint32_t it, RunRegs &c) {
            const uint32_t idx = it * 64u + lane;
            const bool act = idx < nruns;
            const uint32_t rec = runs[act ? idx : nruns - 1u];
            const uint32_t q = rec >> 24, ustar = rec & 0xFFu, first = (rec >> 8) & 0xFFu;
            const uint32_t end = (rec >> 16) & 0xFFu;
            uint32_t m, o;
            context(q, ustar, m, c.A, c.B);
            const uint32_t h = canon_hash(m, o);
            c.b = bucket_index(h, o, ext_of(c.A, c.B, o));
            // (beyond the list: an offset no run has -- no entry's position field names a window of that run)
            const uint32_t dl = act ? ((ustar - first) << 8) | ((end - 1u - first) << 14) : (G::DNONE << 8);
            c.pk = first | dl | (q << 20) | (o << 31);
            // (beyond the list: a range no entry's tag lies in)
            c.t = act ? (((((h & 0xFFu) << PB) | (ustar - (end - 1u))) << TSH) | (((end - 1u - first) << TSH) << 16)) : 0xFFFFu;
            if constexpr (BS) {
                rev_context(c.A, c.B, c.RA, c.RB);
                c.tr = act ? (((((h & 0xFFu) << PB) | (KM - (ustar - first))) << TSH) | (((end - 1u - first) << TSH) << 16)) : 0xFFFFu;
            }
        };
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) u32x4 *gptr128;
#define QUAD_BCAST(v, k) ((uint32_t)__builtin_amdgcn_mov_dpp((int)(v), (k) * 0x55, 0xF, 0xF, true))
        // 64-byte buckets (NL = 1): a batch of 64 runs is fetched with FOUR loads of 16 buckets each -- lane j of a quad fetches bytes 16 j ..
        // 16 j + 15 of the bucket of each of the quad's four runs: P[k] of run k --, three batches deep.
        // 128-byte buckets (NL = 2): a batch is fetched and scanned in two HALVES of 32 runs -- the quad takes two of its four runs at a
        // time (runs 2 half, 2 half + 1); lane j fetches bytes 16 j .. 16 j + 15 of both 64-byte halves of each of the two buckets: P[0],
        // P[1] of the first run, P[2], P[3] of the second; the two requests of a quad for one bucket are the two halves of one line --,
        // three half-batches deep.  Either way 16 registers per unit in flight, 8 KB of buckets in flight while one unit is scanned.
        // BS: the two loads are the run's own bucket (o) and the other one of its pair -- the other half of the same line
#define ISSUE_K(k, at) { const uint64_t b_ = ((uint64_t)QUAD_BCAST(ahi, k) << 32) | (QUAD_BCAST(alo, k) | mine); \
                         P[at] = *(gptr128)b_; if constexpr (NL == 2) P[at + 1] = *(gptr128)(b_ | 64u); if constexpr (BS) P[at + 1] = *(gptr128)(b_ ^ 64u); }
        auto issue = [&](const RunRegs &c, const uint32_t half, u32x4 (&P)[4]) {
            const uint64_t a = tbl + ((uint64_t)c.b << (NL == 2 ? 7 : 6));                              // aligned to the bucket's size
            const uint32_t alo = (uint32_t)a, ahi = (uint32_t)(a >> 32), mine = 16u * (lane & 3u);
            if constexpr (NLX == 1) { ISSUE_K(0, 0) ISSUE_K(1, 1) ISSUE_K(2, 2) ISSUE_K(3, 3) }
            else if (half == 0u) { ISSUE_K(0, 0) ISSUE_K(1, 2) } else { ISSUE_K(2, 0) ISSUE_K(3, 2) }
        };
#undef ISSUE_K
        auto push = [&](uint32_t q, uint32_t rank) {                  // q: the lane whose slot the hit was found in
            const uint32_t rd = q >> SEGSH;
            uint32_t *t = tab + rd;
            uint32_t i = 0;
            for (; i < TSR; ++i) {                                    // (a slot's rank never changes once it is set)
                uint32_t cur = t[i * RPW];
                if (cur == T_EMPTY) {
                    cur = atomicCAS(&t[i * RPW], T_EMPTY, (rank << CB) | 1u);
                    if (cur == T_EMPTY) break;
                }
                if ((cur >> CB) == rank) { atomicAdd(&t[i * RPW], 1u); break; }
            }
            if (i == TSR) atomicOr(&full[rd >> 5], 1u << (rd & 31u));
        };
        // hits of a batch wait in two registers per lane (lane of the read << QS | rank, the later one in `p0`) and go to the reads' lists
        // once per batch.  `last`: these 16 bytes are the bucket's last in this lane's share (the second load); the quad's fourth lane
        // then holds the bucket's last entry, which says whether the bucket continues in an overflow run
        // `rev`: the entries are tested against the reverse complements of the run's windows (ct, cA, cB: the mirrored tags and bases)
        auto scan1 = [&](uint32_t ct, uint32_t cpk, const uint32_t (&cA)[NA], const uint32_t (&cB)[NA], const u32x4 &Pk, const bool last, const bool rev,
                         uint32_t &p0, uint32_t &p1, uint32_t &np) {
            const uint32_t tlo = ct & 0xFFFFu, span = ct >> 16;
            const uint32_t qs = I == 2 ? ((cpk >> 4) & 0x3F0000u) : (cpk & 0x3F00000u);  // read << QS
            bool hit0, hit1, more;
            uint32_t rank0, rank1;
            if constexpr (W == 8 && I == 2) {
                // two entries, each rest | {flag2 hlow8 pos5 0 rank16}; the window that starts pos bases before the minimizer has the outer
                // bases AB >> 2 pos, and 2 pos is what the low six bits of the high word >> 16 hold
                const uint64_t AB = ((uint64_t)cA[0] << 32) | cB[0];
                const uint32_t lo0 = Pk.x, hi0 = Pk.y, lo1 = Pk.z, hi1 = Pk.w;
                const uint32_t s0 = hi0 >> 16, s1 = hi1 >> 16;
                hit0 = (s0 - tlo) <= span && (uint32_t)(AB >> (s0 & 63u)) == lo0;
                hit1 = (s1 - tlo) <= span && (uint32_t)(AB >> (s1 & 63u)) == lo1;
                rank0 = hi0 & 0xFFFFu; rank1 = hi1 & 0xFFFFu;
                more = (hi1 >> 30) == 2u;
            } else if constexpr (W == 8) {
                // one entry: rest | {flag2 hlow8 pos5 0 | 0} | rank32 | 0
                const uint64_t AB = ((uint64_t)cA[0] << 32) | cB[0];
                const uint32_t lo0 = Pk.x, hi0 = Pk.y;
                const uint32_t s0 = hi0 >> 16;
                hit0 = (s0 - tlo) <= span && (uint32_t)(AB >> (s0 & 63u)) == lo0;
                hit1 = false;
                rank0 = Pk.z; rank1 = 0;
                more = (hi0 >> 30) == 2u;
            } else {
                // one entry: {rest low 64}{flag2 | hlow8 pos6 rest-high 32 | rank16}
                const uint32_t s0 = Pk.w >> 16, pos = s0 & 63u;
                uint32_t r0, r1, r2;
                rest96(cA, cB, pos, r0, r1, r2);
                hit0 = (s0 - tlo) <= span && r0 == __builtin_amdgcn_alignbit(Pk.w, Pk.z, 16u) && r1 == Pk.y && r2 == Pk.x;
                hit1 = false;
                rank0 = Pk.z & 0xFFFFu; rank1 = 0;
                more = (Pk.w >> 30) == 2u;
            }
            // (a third hit of a lane in one batch sends the oldest waiting one to its list first: a fraction of a percent of the lanes)
            if (ballot64((hit0 || hit1) && np >= 1u)) {
                if (hit0 && np >= 2u) push(p1 >> QS, p1 & QMASK);
                if (hit1 && np + (hit0 ? 1u : 0u) >= 2u) { const uint32_t w = hit0 ? p0 : p1; push(w >> QS, w & QMASK); }
            }
            p1 = hit0 ? p0 : p1; p0 = hit0 ? (qs | rank0) : p0; np += hit0 ? 1u : 0u;
            if constexpr (W == 8 && I == 2) { p1 = hit1 ? p0 : p1; p0 = hit1 ? (qs | rank1) : p0; np += hit1 ? 1u : 0u; }
            // The bucket continues in an overflow run (its last entry says so: the quad's fourth lane sees it): its windows are
            // looked up there after the loop, one lane per WINDOW.  The run's record goes to the front of the run list --
            // phase B has read further than that: the slots of at least 64 more runs than it has scanned.
            if (last) {
                const uint32_t d = (cpk >> 8) & 63u, lenm1 = (cpk >> 14) & 63u;
                const uint64_t om = ballot64((lane & 3u) == 3u && d != G::DNONE && more);
                if (om) {
                    const uint32_t first = cpk & 0xFFu, q = (cpk >> 20) & 63u;
                    // (the list of overflowing runs grows over the front of the run list, behind what phase B has read: ovf_room records.  One
                    // overflowing bucket per run always fits; with both strands a run can have two -- a database most of whose buckets overflow in
                    // both orientations then runs out of room, and the read is left to the wave-per-read kernel like one with too many labels)
                    const uint32_t at = n_ovf + lanes_below(om);
                    if (((om >> lane) & 1ull) && at >= ovf_room) { const uint32_t rd = q >> SEGSH; atomicOr(&full[rd >> 5], 1u << (rd & 31u)); }
                    if (((om >> lane) & 1ull) && at < ovf_room) {
                        runs[at] = (first + d) | (first << 8) | ((first + lenm1 + 1u) << 16) | (q << 24) | (rev ? 0x80000000u : 0u);   // (bit 31: the strand)
#ifndef UTREE_LANES_REFETCH_DESC
                        // (the descriptor is in this lane's registers: the first 64 of a grab's overflowing runs keep it, and the
                        // overflow stage's first round starts without the trip that fetches it again)
                        if (at < 64u) ost[at] = (W == 8 && I == 4) ? (((uint64_t)Pk.y << 32) | Pk.x) : (((uint64_t)Pk.w << 32) | Pk.z);
#endif
                    }
                    n_ovf = umin(n_ovf + (uint32_t)__popcll(om), ovf_room);
                }
            }
        };
        auto scan = [&](const RunRegs &c, const uint32_t half, const u32x4 (&P)[4]) {
            uint32_t p0 = 0, p1 = 0, np = 0;
#define SCAN_K(k, at) { uint32_t bA[NA], bB[NA]; \
                        _Pragma("unroll") for (uint32_t i = 0; i < NA; ++i) { bA[i] = QUAD_BCAST(c.A[i], k); bB[i] = QUAD_BCAST(c.B[i], k); } \
                        const uint32_t bt = QUAD_BCAST(c.t, k), bpk = QUAD_BCAST(c.pk, k); \
                        scan1(bt, bpk, bA, bB, P[at], NL == 1, false, p0, p1, np); if constexpr (NL == 2) scan1(bt, bpk, bA, bB, P[at + 1], true, false, p0, p1, np); \
                        if constexpr (BS) { \
                            _Pragma("unroll") for (uint32_t i = 0; i < NA; ++i) { bA[i] = QUAD_BCAST(c.RA[i], k); bB[i] = QUAD_BCAST(c.RB[i], k); } \
                            scan1(QUAD_BCAST(c.tr, k), bpk, bA, bB, P[at + 1], true, true, p0, p1, np); } }
            if constexpr (NLX == 1) { SCAN_K(0, 0) SCAN_K(1, 1) SCAN_K(2, 2) SCAN_K(3, 3) }
            else if (half == 0u) { SCAN_K(0, 0) SCAN_K(1, 2) } else { SCAN_K(2, 0) SCAN_K(3, 2) }
#undef SCAN_K
            if (ballot64(np != 0u)) {
                if (np >= 1u) push(p0 >> QS, p0 & QMASK);
                if (ballot64(np >= 2u)) { if (np >= 2u) push(p1 >> QS, p1 & QMASK); }
            }
        };
        if (nruns) {
            // No branch around a load: the waits then count them (a batch beyond the list repeats the list's last run as one no entry
            // belongs to).
            const uint32_t nit = (nruns + 63u) >> 6;
            RunRegs R0, R1, R2;
            u32x4 P0[4], P1[4], P2[4];
            if constexpr (NLX == 1) {
                // three batches deep: two (8 KB of buckets) in flight while one is scanned
                prepare(0u, R0); issue(R0, 0u, P0);
                prepare(1u, R1); issue(R1, 0u, P1);
                for (uint32_t it = 0; it < nit; it += 3) {
                    prepare(it + 2, R2); issue(R2, 0u, P2);
                    ovf_room = umin(nruns, 64u * (it + 3u));
                    scan(R0, 0u, P0);
                    prepare(it + 3, R0); issue(R0, 0u, P0);
                    ovf_room = umin(nruns, 64u * (it + 4u));
                    if (it + 1 < nit) scan(R1, 0u, P1);
                    prepare(it + 4, R1); issue(R1, 0u, P1);
                    ovf_room = umin(nruns, 64u * (it + 5u));
                    if (it + 2 < nit) scan(R2, 0u, P2);
                }
            } else {
                // three half-batches deep: batch `it` lives in R[it mod 3], its halves in P[(2 it) mod 3], P[(2 it + 1) mod 3] -- the loop
                // body is six half-batches, after which the names repeat
                prepare(0u, R0); issue(R0, 0u, P0); issue(R0, 1u, P1);
                for (uint32_t it = 0; it < nit; it += 3) {
                    prepare(it + 1, R1); issue(R1, 0u, P2);
                    ovf_room = umin(nruns, 64u * (it + 2u));
                    scan(R0, 0u, P0);
                    issue(R1, 1u, P0);
                    scan(R0, 1u, P1);
                    prepare(it + 2, R2); issue(R2, 0u, P1);
                    ovf_room = umin(nruns, 64u * (it + 3u));
                    if (it + 1 < nit) scan(R1, 0u, P2);
                    issue(R2, 1u, P2);
                    if (it + 1 < nit) scan(R1, 1u, P0);
                    prepare(it + 3, R0); issue(R0, 0u, P0);
                    ovf_room = umin(nruns, 64u * (it + 4u));
                    if (it + 2 < nit) scan(R2, 0u, P1);
                    issue(R0, 1u, P1);
                    if (it + 2 < nit) scan(R2, 1u, P2);
                }
            }
        }
#undef QUAD_BCAST
        LT(5);
        // ---- runs whose bucket overflows.  The bucket's last entry names the run of MIN records that holds the rest of its nodes.  A
        // short run (up to OVF_SCAN records: a minimizer with a few more nodes than a bucket holds) is read whole, one lane per RECORD,
        // and every record treated like a bucket entry -- which window of the run is it the record of? --: two round trips
        // (descriptor, records) however many windows the run has.  A long one (a minimizer shared by the k-mers of many related
        // genomes) is searched per window by bisection like the wave-per-read kernel does (wave_common.hpp: min_find), one lane per
        // WINDOW and OVF_WAYS windows per lane at a time: the searches' dependent loads overlap. ----
#ifndef UTREE_LANES_OVF_WAYS
#define UTREE_LANES_OVF_WAYS 4
#endif
        // (the threshold is the image's: 32 records, 16 for a database most of whose nodes sit in overflow runs -- related genomes --, where
        // same-box 16 is 3.5 % faster and 64 or more 12-20 % slower; UTREE_OVF_SCAN overrides)
        constexpr uint32_t OVF_WAYS = UTREE_LANES_OVF_WAYS;
        const uint32_t OVF_SCAN = im.ovf_scan;
        const bool OVF_CHAINS = W == 8 && (im.flags & UTREE_F_OVF_CHAINS) != 0u;      // heavy runs are lists of chains (device_common.hpp)
        // the descriptor of overflowing run i: the key word of its bucket's last entry (a line phase B has fetched: an L2 hit mostly)
        auto fetch_desc = [&](uint32_t i) -> uint64_t {
            uint64_t dsc = 0;
            if (i < n_ovf) {
                const uint32_t rec = runs[i];
                uint32_t m, o, A[NA], B[NA];
                context((rec >> 24) & 63u, rec & 0xFFu, m, A, B);
                const uint32_t h = canon_hash(m, o);
                const uint64_t baddr = bucket_addr(h, o ^ (rec >> 31), ext_of(A, B, o));   // (the reverse strand's records: the other bucket of the pair)
                dsc = *(const __attribute__((address_space(1))) uint64_t *)(baddr + (64u * NL - 8u * EW + 8u * KW));
            }
            return dsc;
        };
        uint64_t dsc_next = 0;                                                     // requested one round ahead
        for (uint32_t ib = 0; ib < n_ovf; ib += 64) {
            wave_lds_fence();
            const uint32_t i = ib + lane;
            uint32_t nrec = 0, wn = 0;                                             // records to scan / windows to search of the lane's item
            // round 0's descriptors were kept by the scans that found the runs; every later round's were requested during the round
            // before it: no round waits for them
#ifdef UTREE_LANES_REFETCH_DESC
            uint64_t dsc = fetch_desc(i);
#else
            uint64_t dsc = ib == 0u ? ost[lane] : dsc_next;
            if (ib + 64u < n_ovf) dsc_next = fetch_desc(i + 64u);
#endif
            if (i < n_ovf) {
                const uint32_t rec = runs[i];
                const uint64_t n = ovf_count(dsc);
                ost[lane] = dsc;
                // (a heavy run stored as chains counts its chains: searched per window whatever their number)
                if (n <= OVF_SCAN && !(OVF_CHAINS && (dsc & OVF_HAS_DIR))) nrec = (uint32_t)n; else wn = ((rec >> 16) & 0xFFu) - ((rec >> 8) & 0xFFu);
            }
            // (t-th unit of work -> item: inclusive prefix sums in LDS, first item whose sum exceeds t)
            auto spread = [&](uint32_t mine) -> uint32_t {
                uint32_t incl = mine;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t t = __shfl_up(incl, dd); if (lane >= (uint32_t)dd) incl += t; }
                wave_lds_fence();
                pref[lane] = incl;
                wave_lds_fence();
                return uni32((uint32_t)__shfl(incl, 63));
            };
            auto item_of = [&](uint32_t t, uint32_t &within) -> uint32_t {
                uint32_t lo = 0, hi = 63;
#pragma unroll
                for (int st = 0; st < 6; ++st) { const uint32_t mid = (lo + hi) >> 1; if (pref[mid] <= t) lo = mid + 1; else hi = mid; }
                within = t - (lo ? pref[lo - 1] : 0u);
                return lo;
            };
            const uint32_t total_rec = spread(nrec);
            // (k = 64: two blocks of 64 records per step, both blocks' records requested before the first is looked at -- the loop is one memory
            // round trip per step, and a grab of a k = 64 database has a few hundred such records: same box 1.33 -> 1.31 ms; k = 32, config 2, with
            // twenty overflowing runs per grab: +2-3 % with two blocks, one it is)
            constexpr uint32_t RB = W == 16 ? 2 : 1;
            for (uint32_t t0 = 0; t0 < total_rec; t0 += 64 * RB) {
                Entry<W, I> e[RB];
                uint32_t recs[RB];
#pragma unroll
                for (uint32_t u = 0; u < RB; ++u) {
                    const uint32_t t = t0 + 64u * u + lane;
                    recs[u] = 0;
#pragma unroll
                    for (int x = 0; x < EW; ++x) e[u].w[x] = 0;
                    if (t < total_rec) {
                        uint32_t j;
                        const uint32_t it_ = item_of(t, j);
                        recs[u] = runs[ib + it_];
                        e[u] = load_entry<W, I>(im.mrecs, ovf_first<W, I>(ost[it_]) + j);
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < RB; ++u) {
                    const uint32_t t = t0 + 64u * u + lane;
                    if (t < total_rec) {
                        const uint32_t rec = recs[u];
                        const uint32_t q = (rec >> 24) & 63u, ustar = rec & 0xFFu, first = (rec >> 8) & 0xFFu, lenm1 = ((rec >> 16) & 0xFFu) - 1u - first;
                        const bool rev = BS && (rec >> 31) != 0u;
                        uint32_t m, o, A[NA], B[NA];
                        context(q, ustar, m, A, B);
                        if constexpr (BS) { if (rev) { uint32_t RA[NA], RB[NA]; rev_context(A, B, RA, RB);
_Pragma("unroll") for (uint32_t x = 0; x < NA; ++x) { A[x] = RA[x]; B[x] = RB[x]; } } }
                        const uint32_t h = canon_hash(m, o);
                        const uint32_t hlow = h & 0xFFu;
                        // the window an entry of position `pos` is the record of starts at ustar - pos; on the reverse strand at ustar - (K-16 - pos)
                        const uint32_t wbase = rev ? ustar - first - KM : ustar - first;
                        bool hit;
                        uint32_t rank;
                        if constexpr (W == 8) {
                            const uint32_t lo = (uint32_t)e[u].w[0], hi = (uint32_t)(e[u].w[0] >> 32), pos = (hi >> 17) & 31u;
                            hit = (hi >> 22) == hlow && (rev ? wbase + pos : wbase - pos) <= lenm1 && (uint32_t)((((uint64_t)A[0] << 32) | B[0]) >> (2u * pos)) == lo;
                            if constexpr (I == 2) rank = hi & 0xFFFFu; else { rank = (uint32_t)e[u].w[1]; hit = hit && rank != INVALID; }
                        } else {
                            const uint32_t z = (uint32_t)e[u].w[1], wq = (uint32_t)(e[u].w[1] >> 32), pos = (wq >> 16) & 63u;
                            uint32_t r0, r1, r2;
                            rest96(A, B, pos, r0, r1, r2);
                            hit = (wq >> 22) == hlow && (rev ? wbase + pos : wbase - pos) <= lenm1 && r0 == __builtin_amdgcn_alignbit(wq, z, 16u) &&
                                  r1 == (uint32_t)(e[u].w[0] >> 32) && r2 == (uint32_t)e[u].w[0];
                            rank = z & 0xFFFFu;
                        }
                        if (hit && (I == 4 || rank != 0xFFFFu)) push(q, rank);
                    }
                }
            }
            const uint32_t total_win = spread(wn);
            for (uint32_t t0 = 0; t0 < total_win; t0 += 64 * OVF_WAYS) {
                uint64_t lo[OVF_WAYS], hi[OVF_WAYS];                               // the searches' ranges (empty: done or no window)
                MinKey<W> mk[OVF_WAYS];
                uint32_t qs[OVF_WAYS];
                const uint16_t *dirp[OVF_WAYS];                                    // where a heavy run's directory says which of its records have the window's position
                const uint64_t *chp[OVF_WAYS];                                     // a heavy run stored as chains: the run, its chains, where the window's rank is
                uint32_t chn[OVF_WAYS], chat[OVF_WAYS];
#pragma unroll
                for (uint32_t u = 0; u < OVF_WAYS; ++u) {
                    const uint32_t t = t0 + 64u * u + lane;
                    lo[u] = hi[u] = 0; qs[u] = 0; mk[u].lo = mk[u].hi = 0;
                    dirp[u] = nullptr;
                    chp[u] = im.mrecs; chn[u] = 0; chat[u] = ~0u;
                    if (t < total_win) {
                        uint32_t j;
                        const uint32_t it_ = item_of(t, j);
                        const uint32_t rec = runs[ib + it_];
                        const uint32_t q = (rec >> 24) & 63u, ustar = rec & 0xFFu, first = (rec >> 8) & 0xFFu;
                        const bool rev = BS && (rec >> 31) != 0u;
                        // the window's minimizer position, 0..K-16 (its reverse complement's: mirrored)
                        const uint32_t pos = rev ? KM - (ustar - (first + j)) : ustar - (first + j);
                        uint32_t m, o, A[NA], B[NA];
                        context(q, ustar, m, A, B);
                        if constexpr (BS) { if (rev) { uint32_t RA[NA], RB[NA]; rev_context(A, B, RA, RB);
_Pragma("unroll") for (uint32_t x = 0; x < NA; ++x) { A[x] = RA[x]; B[x] = RB[x]; } } }
                        const uint32_t h = canon_hash(m, o);
                        const uint32_t hlow = h & 0xFFu;
                        const uint64_t dsc = ost[it_];
                        lo[u] = ovf_first<W, I>(dsc); hi[u] = lo[u] + ovf_count(dsc); qs[u] = q;
                        if (OVF_CHAINS && (dsc & OVF_HAS_DIR)) { chp[u] = im.mrecs + (dsc & M39) * EW; chn[u] = (uint32_t)ovf_count(dsc); hi[u] = lo[u]; }
                        else if (dsc & OVF_HAS_DIR) dirp[u] = (const uint16_t *)(im.mrecs + (dsc & M39) * EW) + pos;
                        if constexpr (W == 8) {
                            const uint32_t rest = (uint32_t)((((uint64_t)A[0] << 32) | B[0]) >> (2u * pos));
                            mk[u].hi = 0; mk[u].lo = ((uint64_t)((hlow << 5) | pos) << 32) | rest;
                        } else {
                            uint32_t r0, r1, r2;
                            rest96(A, B, pos, r0, r1, r2);
                            mk[u].lo = ((uint64_t)r1 << 32) | r2; mk[u].hi = ((uint64_t)hlow << 38) | ((uint64_t)pos << 32) | r0;
                        }
                    }
                }
                // a heavy run stored as chains: every search of the round walks its run's chains in step, two chains a turn (16 bytes each; the windows of one
                // run read the same addresses), then one load fetches the rank
                if constexpr (W == 8) {
                    bool anych = false;
#pragma unroll
                    for (uint32_t u = 0; u < OVF_WAYS; ++u) anych = anych || chn[u] != 0u;
                    if (OVF_CHAINS && ballot64(anych)) {
                        for (uint32_t c = 0;; c += 2) {
                            bool more = false;
#pragma unroll
                            for (uint32_t u = 0; u < OVF_WAYS; ++u) more = more || (c < chn[u] && chat[u] == ~0u);
                            if (!ballot64(more)) break;
                            uint64_t h0[OVF_WAYS][2], h1[OVF_WAYS][2];
#pragma unroll
                            for (uint32_t u = 0; u < OVF_WAYS; ++u) {
                                const uint32_t c0 = c < chn[u] ? c : 0u, c1 = c + 1u < chn[u] ? c + 1u : 0u;     // (beyond the run's chains: its first again, or the image's first bytes)
                                h0[u][0] = chp[u][2u * c0]; h0[u][1] = chp[u][2u * c0 + 1u];
                                h1[u][0] = chp[u][2u * c1]; h1[u][1] = chp[u][2u * c1 + 1u];
                            }
#pragma unroll
                            for (uint32_t u = 0; u < OVF_WAYS; ++u) {
                                const uint32_t pos = (uint32_t)(mk[u].lo >> 32) & 31u, rest = (uint32_t)mk[u].lo;
                                const uint32_t a0 = chain_hit(h0[u][0], h0[u][1], pos, rest), a1 = chain_hit(h1[u][0], h1[u][1], pos, rest);
                                if (c < chn[u] && chat[u] == ~0u) chat[u] = a0;
                                if (c + 1u < chn[u] && chat[u] == ~0u) chat[u] = a1;
                            }
                        }
                        uint32_t rk[OVF_WAYS];
#pragma unroll
                        for (uint32_t u = 0; u < OVF_WAYS; ++u) {
                            const uint32_t at = chat[u] == ~0u ? 0u : chat[u];
                            if constexpr (I == 2) rk[u] = ((const uint16_t *)(chp[u] + 2u * chn[u]))[at]; else rk[u] = ((const uint32_t *)(chp[u] + 2u * chn[u]))[at];
                        }
#pragma unroll
                        for (uint32_t u = 0; u < OVF_WAYS; ++u)
                            if (chat[u] != ~0u && (I == 4 ? rk[u] != INVALID : rk[u] != 0xFFFFu)) push(qs[u], rk[u]);
                    }
                }
                // a heavy run: its directory narrows the range to the records of the window's own minimizer position (one trip for all the searches of the
                // round; a lane without a directory reads the image's first bytes)
                bool anydir = false;
#pragma unroll
                for (uint32_t u = 0; u < OVF_WAYS; ++u) anydir = anydir || dirp[u] != nullptr;
                if (ballot64(anydir)) {                                            // (no run of the round has one -- the usual case outside related genomes --: no trip)
                    uint32_t da[OVF_WAYS], db[OVF_WAYS];
#pragma unroll
                    for (uint32_t u = 0; u < OVF_WAYS; ++u) {
                        const uint16_t *dp_ = dirp[u] ? dirp[u] : (const uint16_t *)im.mrecs;
                        da[u] = dp_[0]; db[u] = dp_[1];
                    }
#pragma unroll
                    for (uint32_t u = 0; u < OVF_WAYS; ++u) if (dirp[u]) { hi[u] = lo[u] + db[u]; lo[u] += da[u]; }
                }
                // exact-match bisection in runs that ascend by key, OVF_WAYS searches per lane in step (a finished one reads record 0 of its range again: no branch around a load)
                for (;;) {
                    bool any = false;
#pragma unroll
                    for (uint32_t u = 0; u < OVF_WAYS; ++u) any = any || lo[u] < hi[u];
                    if (!ballot64(any)) break;
                    Entry<W, I> e[OVF_WAYS];
                    uint64_t mid[OVF_WAYS];
#pragma unroll
                    for (uint32_t u = 0; u < OVF_WAYS; ++u) { mid[u] = lo[u] + ((hi[u] - lo[u]) >> 1); e[u] = load_entry<W, I>(im.mrecs, mid[u]); }
#pragma unroll
                    for (uint32_t u = 0; u < OVF_WAYS; ++u) {
                        if (lo[u] < hi[u]) {
                            const MinKey<W> k = mrec_key<W, I>(e[u]);
                            if (mkey_lt<W>(k, mk[u])) lo[u] = mid[u] + 1;
                            else if (mkey_eq<W>(k, mk[u])) { const uint32_t rank = mrec_rank<W, I>(e[u]); if (rank != INVALID) push(qs[u], rank); hi[u] = lo[u]; }
                            else hi[u] = mid[u];
                        }
                    }
                }
            }
        }
        wave_lds_fence();
        }   // strand
        LT(3);

        // ---- phase C: tally (itree.c:1028-1040), result records, the list of reads left to the wave-per-read kernel ----
        // (lane i < 64 / SEGS finishes read i of the grab; what its lanes could not take is in their `exc`)
        const uint64_t excm = ballot64(exc);
        const bool have = lane < RPW && item + lane < item_end && !other;
        const uint32_t r = LISTED ? (have ? lst[item + lane] : 0u) : item + lane;    // the read (PIECE: the piece)
        exc = ((excm >> ((lane * SEGS) & 63u)) & ((1ull << SEGS) - 1ull)) != 0ull || ((full[lane >> 5] >> (lane & 31u)) & 1u) != 0u || wave_full;
        if constexpr (PIECE) {
            // a piece's table is added to its read's: one lane per (piece, slot), claim-or-find the rank, add the count
            if (have) pref[lane] = (uint32_t)(ws.pieces[r] >> 32);               // the read's entry of long_list
            if (have && exc) ws.lflag[pref[lane]] = 1u;                            // the whole read is classify_long_k's
            const uint64_t okm = ballot64(have && !exc);
            wave_lds_fence();
            for (uint32_t x0 = 0; x0 < RPW * TSR; x0 += 64) {
                const uint32_t x = x0 + lane, pc = x / TSR, slot = x % TSR;
                if (x < RPW * TSR && ((okm >> pc) & 1ull)) {
                    const uint32_t e = tab[slot * RPW + pc];
                    if (e != T_EMPTY) {
                        const uint32_t li = pref[pc], rank = e >> CB;
                        uint32_t *tr = ws.ltab_rank + (size_t)li * UTREE_LONG_SLOTS, *tc = ws.ltab_cnt + (size_t)li * UTREE_LONG_SLOTS;
                        uint32_t sidx = (rank * 0x9E3779B1u) >> 26, tries = 0;     // 64 slots, linear probing
                        for (; tries < UTREE_LONG_SLOTS; ++tries, sidx = (sidx + 1u) & (UTREE_LONG_SLOTS - 1u)) {
                            uint32_t cur = __hip_atomic_load(&tr[sidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (cur == 0xFFFFFFFFu) cur = atomicCAS(&tr[sidx], 0xFFFFFFFFu, rank);
                            if (cur == 0xFFFFFFFFu || cur == rank) { atomicAdd(&tc[sidx], e & CMASK); break; }
                        }
                        if (tries == UTREE_LONG_SLOTS) ws.lflag[li] = 1u;          // more labels than the read's table holds
                    }
                }
            }
            wave_lds_fence();
            LT(4);
            continue;
        }
        const uint64_t xm = ballot64(have && exc);
        if (xm) {
            unsigned long long xb = 0;
            if (lane == 0) xb = atomicAdd(&ws.cursors[UTREE_CUR_MID], (unsigned long long)__popcll(xm));
            xb = uni64(xb);
            if (have && exc) ws.mid_list[xb + lanes_below(xm)] = r;
        }
        const bool live = have && !exc;
        // the read's table: nu distinct labels (slots 0 .. nu-1), F hits in all
        const uint32_t *tq = tab + (lane < RPW ? lane : 0u);
        uint32_t nu = 0, F = 0;
        if (live) {
#pragma unroll
            for (uint32_t i = 0; i < TSR; ++i) { const uint32_t e = tq[i * RPW]; if (e != T_EMPTY) { ++nu; F += e & CMASK; } }
        }
        const uint32_t first_rank = tq[0] >> CB;
        const uint32_t maxnu = uni32(wave_max_u32(nu));
        // space for the (rank, count) lists of the reads with two or more labels: one reservation per wave and TALLY_CHUNK
        const uint32_t need = nu >= 2u ? nu : 0u;
        uint32_t incl = need;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t t = __shfl_up(incl, dd); if (lane >= (uint32_t)dd) incl += t; }
        const uint32_t total = uni32((uint32_t)__shfl(incl, 63));
        if (total > chunk_left) {
            unsigned long long nb = 0;
            if (lane == 0) {
                nb = atomicAdd(&ws.cursors[0], (unsigned long long)UTREE_TALLY_CHUNK);
                // (the workspace is sized so that this cannot happen, dev_image.c: carve; if it does, the batch is reported as failed
                // and the wave writes into the first chunk instead of past the end)
                if (nb + UTREE_TALLY_CHUNK > ws.tally_cap) { ws.cursors[UTREE_CUR_ERROR] = UTREE_DEVERR_TALLY_CAP; nb = 0; }
            }
            chunk_base = uni64(nb);
            chunk_left = UTREE_TALLY_CHUNK;
        }
        const unsigned long long my_base = chunk_base + (incl - need);
        chunk_base += total; chunk_left -= total;
        // ascending rank = strcmp order (itree.c:1041): an entry's place is the number of the read's labels below its own
        if (maxnu >= 2u) {
            for (uint32_t i = 0; i < maxnu; ++i) {
                const uint32_t e = tq[i * RPW];
                uint32_t place = 0;
                for (uint32_t j = 0; j < maxnu; ++j) { const uint32_t x = tq[j * RPW]; place += (j < nu && (x >> CB) < (e >> CB)) ? 1u : 0u; }
                if (need && i < nu) ws.tally[my_base + place] = (uint64_t)(e >> CB) | ((uint64_t)(e & CMASK) << 32);
            }
        }
        if (live) {
            if (F == 0) store_result(&out[r], 0, -2, 0, 0, 0, 0);
            else if (nu == 1) store_result(&out[r], first_rank, RANK_PENDING, F, 1, 0, 0);      // (vote_k looks the file-order index up: doing it here is +-0, same box)
            else store_result(&out[r], 0, CUT_PENDING, F, nu, (uint32_t)my_base, (uint32_t)(my_base >> 32));
        }
        wave_lds_fence();
        LT(4);
    }
#ifdef UTREE_LANES_TIMERS
    if (lane == 0) { for (int q = 0; q < 6; ++q) atomicAdd(&g_lphase[q], lt_acc[q]); atomicAdd(&g_lphase[7], 1ull); }
#endif
}

// the wavefront's LDS and what is set up once per launch: slots (13 or 15 words per lane), run list, tally tables {slot}{read} (a read's hits are
// tallied as they are found: TSR slots {rank << 16 | count} per read, filled from slot 0 -- itree.c:1031-1040 needs the distinct labels with
// their counts, in any order; a lane's walk over its own slots is conflict-free), the reads with more labels than slots, a prefix-sum scratch,
// the overflow descriptors of up to 64 runs, the region table {first pair << 25 | pairs}
#define LANES_PROLOGUE(W_) \
    using G_ = Geo<W_>; \
    __shared__ uint32_t s_stream[LANES_WAVES][64 * G_::STRIDE]; \
    __shared__ uint32_t s_runs[LANES_WAVES][G_::RUNS]; \
    __shared__ uint32_t s_tab[LANES_WAVES][64 * TSLOTS]; \
    __shared__ uint32_t s_full[LANES_WAVES][2]; \
    __shared__ uint32_t s_pref[LANES_WAVES][64]; \
    __shared__ uint64_t s_ost[LANES_WAVES][64]; \
    __shared__ uint64_t s_reg[256]; \
    for (uint32_t x = threadIdx.x; x < 256; x += blockDim.x) s_reg[x] = im.regions[x]; \
    const uint32_t wv = uni32(threadIdx.x >> 6); \
    { uint32_t *sl_ = s_stream[wv] + lane_id() * G_::STRIDE + G_::FRONT; \
      _Pragma("unroll") for (uint32_t i = 1; i <= G_::FRONT; ++i) sl_[-(int)i] = 0;      /* pads: zero for good */ \
      sl_[NWORD] = 0; sl_[NWORD + 1] = 0; } \
    __syncthreads();
#define LANES_LDS s_stream[wv], s_runs[wv], s_tab[wv], s_full[wv], s_pref[wv], s_ost[wv], s_reg, wv

template <int W, int I, int SEGS, bool IRR, int MODE, int NL, bool BS>
__global__ __launch_bounds__(LANES_WAVES * 64, W == 16 ? UTREE_LANES_WPS64 : UTREE_LANES_WPS)
void classify_lanes_k(utk_image im, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ off, const uint32_t *__restrict__ len,
                      uint32_t n_reads, int do_rc, utree_result *__restrict__ out, utk_workspace ws, uint32_t cls) {
    LANES_PROLOGUE(W)
    lanes_body<W, I, SEGS, IRR, MODE, NL, BS>(im, bases, off, len, n_reads, do_rc, out, ws, cls, LANES_LDS);
}

// A batch of mixed read lengths in ONE launch: lanes_route_k has listed the reads by the lanes they need; a wavefront works through the classes
// one after the other -- the reads of one lane (the plain walk over the batch, passing over the longer ones), then the listed reads of two, four,
// eight and sixteen lanes -- so that a class with few reads costs its grabs, not a launch with its start and its drain (a launch per class:
// +10.6 % for 1 % longer reads in a 16 M-read batch, profiles/r03/mixed_batches_16M_launches.json).  max_cls: the largest class a read of the
// batch can need.
// (the bodies are inlined: called as functions -- one register allocation each -- their LDS pointers become generic ones and every LDS access a flat
// one: 1.87 x the time, profiles/r04/mixed_one_launch_called_16M.json; inlined five times over, the hot body spills 70-90 registers and is still the faster)
template <int W, int I, bool IRR, int NL, bool BS>
__global__ __launch_bounds__(LANES_WAVES * 64, W == 16 ? UTREE_LANES_WPS64 : UTREE_LANES_WPS)
void classify_lanes_mixed_k(utk_image im, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ off, const uint32_t *__restrict__ len,
                            uint32_t n_reads, int do_rc, utree_result *__restrict__ out, utk_workspace ws, uint32_t max_cls) {
    LANES_PROLOGUE(W)
    lanes_body<W, I, 1, IRR, 0, NL, BS>(im, bases, off, len, n_reads, do_rc, out, ws, 1u, LANES_LDS);
    if (max_cls >= 1u) lanes_body<W, I, 2, IRR, 1, NL, BS>(im, bases, off, len, n_reads, do_rc, out, ws, 1u, LANES_LDS);
    if (max_cls >= 2u) lanes_body<W, I, 4, IRR, 1, NL, BS>(im, bases, off, len, n_reads, do_rc, out, ws, 2u, LANES_LDS);
    if (max_cls >= 3u) lanes_body<W, I, 8, IRR, 1, NL, BS>(im, bases, off, len, n_reads, do_rc, out, ws, 3u, LANES_LDS);
    if (max_cls >= 4u) lanes_body<W, I, 16, IRR, 1, NL, BS>(im, bases, off, len, n_reads, do_rc, out, ws, 4u, LANES_LDS);
}

static inline uint32_t lanes_resident_blocks(int W, int n_cu) {
    const uint32_t wps = W == 16 ? UTREE_LANES_WPS64 : UTREE_LANES_WPS;
    return (uint32_t)n_cu * (4u * wps / LANES_WAVES > 0 ? 4u * wps / LANES_WAVES : 1u);
}

template <int W, int I, int SEGS, bool IRR, int MODE, int NL, bool BS = false>
static int launch_lanes(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len, uint32_t n_reads,
                        int do_rc, utree_result *d_out, const utk_workspace *ws, int n_cu, void *stream, uint32_t cls = 0) {
    static_assert(NL == 1 || NL == 2, "bucket size");
    uint32_t blocks = (n_reads + (64u / SEGS) * LANES_WAVES - 1) / ((64u / SEGS) * LANES_WAVES);
    const uint32_t cap = lanes_resident_blocks(W, n_cu);
    if (blocks > cap) blocks = cap;
    classify_lanes_k<W, I, SEGS, IRR, MODE, NL, BS><<<dim3(blocks), dim3(LANES_WAVES * 64), 0, (hipStream_t)stream>>>(*im, d_bases, d_off, d_len, n_reads, do_rc, d_out, *ws, cls);
    return (int)hipGetLastError();
}

template <int W, int I, bool IRR, int NL, bool BS = false>
static int launch_lanes_mixed(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len, uint32_t n_reads,
                              int do_rc, utree_result *d_out, const utk_workspace *ws, int n_cu, void *stream, uint32_t max_cls) {
    // (the classes' read counts are on the device: the grid is what the batch's reads of one lane could fill, at most the resident one)
    uint32_t blocks = (n_reads + 64u * LANES_WAVES - 1) / (64u * LANES_WAVES);
    const uint32_t cap = lanes_resident_blocks(W, n_cu);
    if (blocks > cap) blocks = cap;
    classify_lanes_mixed_k<W, I, IRR, NL, BS><<<dim3(blocks), dim3(LANES_WAVES * 64), 0, (hipStream_t)stream>>>(*im, d_bases, d_off, d_len, n_reads, do_rc, d_out, *ws, max_cls);
    return (int)hipGetLastError();
}

}  // namespace
