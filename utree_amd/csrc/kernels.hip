// kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the SEARCH_GG path and their C launchers.
//
// Written for wave64 / gfx950 only.  Integer, pointer-chasing work: no MFMA.  What bounds each kernel,
// its algorithmic bytes and the HBM layout are in DESIGN.md; the reference behaviour each kernel
// reproduces is cited as itree.c:line.
//
//   route_k         lists the reads the main pass cannot hold (mid-length pass / classify_long_k)
//   classify_short  one wavefront per read, software-pipelined over a grab of reads: raw bytes -> LDS (LDS-DMA, one
//                   read ahead), 2-bit staging, sliding minimizers, bucket lookups, tally of distinct labels
//   classify_long   one workgroup per read for reads that do not fit a wavefront's LDS slice
//   vote_k          one lane per read: rank-wise LCA descent on the sorted unique label list
//   lookup_k        XT_getIX32 alone (tests, micro-benchmarks)
// (the load-time kernels that build the device image live in image_build.hip)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "device_common.hpp"
#include "wave_common.hpp"

using namespace utk;

namespace {

// Phase timers (measurement builds only, -DUTREE_PHASE_TIMERS): per-wave cycle counts between TICK()s, summed over all waves
// into g_phase[] and printed by utk_phase_dump().  s_memtime is itself a scalar memory read (it waits for the wave's LDS
// traffic), so the figures show where a wave's time goes, not exact costs.
#ifdef UTREE_PHASE_TIMERS
__device__ unsigned long long g_phase[16];
struct PhaseT { unsigned long long t, acc[12]; };
#define PH_DECL PhaseT ph_; ph_.t = __builtin_readcyclecounter(); for (int q_ = 0; q_ < 12; ++q_) ph_.acc[q_] = 0;
#define TICKP(ph, i) do { const unsigned long long n_ = __builtin_readcyclecounter(); (ph).acc[i] += n_ - (ph).t; (ph).t = n_; } while (0)
#define TICK(i) TICKP(ph_, i)
#define PH_ARG , PhaseT &ph_
#define PH_PASS , ph_
#define PH_WAITVM asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define PH_DECL
#define TICK(i)
#define PH_ARG
#define PH_PASS
#define PH_WAITVM
#endif


constexpr uint32_t KEY_TILE = 128;                    // windows per sliding-minimizer tile (256 where the LDS budget allows)

// Every 16-mer or window word a lane extracts from the packed 2-bit stream starts at a base index congruent to its lane id
// mod 64 (tiles and rounds start at multiples of 64 bases), so the two words it straddles sit at a fixed per-lane offset from
// a wave-uniform word index and its shift never changes: one v_alignbit_b32 per 32 bits, no per-use address arithmetic.
//   lane & 15 = r != 0: bits = ({sw[q], sw[q+1]} >> (32 - 2r)) low 32     r == 0: bits = sw[q] = ({sw[q-1], sw[q]} >> 0) low 32
// (sw[-1] must be readable: the callers' arrays carry a pad word in front).
struct LaneBits {
    const uint32_t *swl;      // sw + (lane >> 4) - (r == 0)
    uint32_t sh;              // (32 - 2r) & 31
};
__device__ __forceinline__ LaneBits lane_bits(const uint32_t *sw, uint32_t lane) {
    LaneBits b;
    const uint32_t r = lane & 15u;
    b.swl = sw + (lane >> 4) - (r == 0 ? 1u : 0u);
    b.sh = (32u - 2u * r) & 31u;
    return b;
}
// 16 bases (32 bits) starting at base 16 * word + lane of the stream `s` points into (s = lb.swl + a uniform word index)
__device__ __forceinline__ uint32_t mmer_l(const uint32_t *s, uint32_t word, uint32_t sh) {
    return __builtin_amdgcn_alignbit(s[word], s[word + 1], sh);
}

// Sliding minimizers for a tile of windows, shared by the lanes of one wave (each 16-mer is hashed ONCE instead of
// once per window that contains it).  The order among 16-mers is that of minimizer<W>() (device_common.hpp): the hash
// without its MIN_LOW_BITS low bits, leftmost on ties -- which makes a 32-bit key { hash bits | position in the tile }
// enough, and the sliding minimum one v_min_u32 per step.  After the call, for the window that starts at tile position t:
//     p = min(Kk[t + FIRST], Kk[t + FIRST + NEXT]) & MIN_POS_MASK  =  tile position of its minimizer,  Hh[p] = that 16-mer's full hash,
// the same (h, pos) minimizer<W>() computes from the word.  Kk and Hh need (NCH + 1) * 64 entries.
// In place: step s replaces K[t] by min(K[t], K[t+s]); chunks ascend, so chunk c still sees chunk c+1's old values.
// Straight-line per chunk count NCH (the dispatcher below picks it): all LDS offsets are immediates off the per-lane
// pointers Kl = Kk + lane, Hl = Hh + lane.  Positions past the read's last full 16-mer hash stale bytes; no valid window
// looks at them (a window's 16-mers lie inside the read).
constexpr uint32_t MIN_POS_MASK = (1u << MIN_LOW_BITS) - 1u;       // a tile has < 2^MIN_LOW_BITS positions
// (k = 64: the candidates of window t are the 16-mers t+2 .. t+46 -- UTREE_MIN_MARGIN)
template <int W> struct MinWin { static constexpr uint32_t FIRST = UTREE_MIN_MARGIN(W), NEXT = (W == 16) ? 13u : 1u; };   // [t,t+16)+[t+1,t+17) / [t+2,t+34)+[t+15,t+47)
template <int W, uint32_t NCH>
__device__ __forceinline__ void build_minkeys_n(const uint32_t *s, uint32_t sh, uint32_t *Kl, uint32_t *Hl, uint64_t *Ob, uint32_t lane) {
    uint32_t x[NCH];
#pragma unroll
    for (uint32_t c = 0; c < NCH; ++c) x[c] = mmer_l(s, 4 * c, sh);          // all chunks' words are requested before the first is hashed
    // (the lane id through an opaque move: otherwise the compiler keeps lane | 64, lane | 128, ... in registers of their own for
    // the whole kernel, and the 64 registers that 8 waves per SIMD allow are all taken)
    uint32_t lo = lane;
    asm volatile("" : "+v"(lo));
#pragma unroll
    for (uint32_t c = 0; c < NCH; ++c) {
        // (image version 11: a 16-mer is ranked and addressed by the hash of its canonical form; which orientation the read holds picks the
        // bucket of the pair -- one bit per position, kept as the chunk's ballot)
        uint32_t o;
        const uint32_t h = canon_hash(x[c], o);
        Kl[c * 64] = ((h & ~MIN_POS_MASK) | lo) + c * 64;
        Hl[c * 64] = h;
        Ob[c] = __builtin_amdgcn_ballot_w64(o != 0u);
    }
    Kl[NCH * 64] = ~0u;
    wave_lds_fence();
    // every step reads ALL chunks before it writes any (one LDS round trip per step, not one per chunk and step)
#pragma unroll
    for (uint32_t st = 1; st <= ((W == 16) ? 16u : 8u); st <<= 1) {
        uint32_t a[NCH], b[NCH];
#pragma unroll
        for (uint32_t c = 0; c < NCH; ++c) { a[c] = Kl[c * 64]; b[c] = Kl[c * 64 + st]; }
        wave_lds_fence();
#pragma unroll
        for (uint32_t c = 0; c < NCH; ++c) Kl[c * 64] = b[c] < a[c] ? b[c] : a[c];
        wave_lds_fence();
    }
}
template <int W, uint32_t TILE>
__device__ __forceinline__ void build_minkeys(const uint32_t *s, uint32_t sh, uint32_t *Kl, uint32_t *Hl, uint64_t *Ob, uint32_t npos, uint32_t lane) {
    constexpr uint32_t MAXCH = (TILE + 4 * W - 16 + 63) / 64;
    static_assert(MAXCH <= 5, "tile too large");
    const uint32_t nch = (npos + 63) >> 6;                 // wave-uniform
    if (nch <= 1) build_minkeys_n<W, 1>(s, sh, Kl, Hl, Ob, lane);
    else if (nch == 2) build_minkeys_n<W, 2>(s, sh, Kl, Hl, Ob, lane);
    else if (nch == 3 || MAXCH == 3) build_minkeys_n<W, 3>(s, sh, Kl, Hl, Ob, lane);
    else if (nch == 4) build_minkeys_n<W, MAXCH >= 4 ? 4 : 3>(s, sh, Kl, Hl, Ob, lane);
    else build_minkeys_n<W, MAXCH >= 5 ? 5 : 3>(s, sh, Kl, Hl, Ob, lane);
}

// The block's copy of the image's region table in the form the window loop wants it (stage_regions):
//   s_reg[r] = the header's entry, first pair << 34 | pairs per slot << 25 | slots of the region (device_common.hpp: pair_of)
struct RegionLds { const uint64_t *reg; uint64_t table; uint32_t bshift; };   // bshift: log2 of a bucket's bytes

// One wave looks up the windows [w0, w0+n) of a staged buffer (lb / sw = packed bases, sbad = bad-base bit words, both
// indexed from the buffer's first base; w0 a multiple of 64) and hands every lane's result to on_rank(rank) -- called by all
// lanes, INVALID for lanes without a hit -- once per 64 windows.  One bucket (64 bytes) is in flight per lane: its records
// arrive in one round trip, and two buckets in flight would not fit the registers.  CHECKBAD = false: the caller knows the
// windows hold no bad base (most reads), and the per-window test disappears.
template <int W, int I, bool EXC, typename OFF, uint32_t TILE, bool CHECKBAD, bool PAIRS, typename HitFn>
__device__ __forceinline__ void wave_scan_windows(const utk_image &im, const LaneBits &lb, const uint64_t *sbad, uint64_t *Kk,
                                                  uint32_t w0, uint32_t n, const RegionLds &rg, uint32_t lane, HitFn &&on_rank PH_ARG) {
    constexpr uint32_t K = 4 * W;
    if constexpr (W == 4) {
        // PACKSIZE=16: no minimizers, no buckets -- lane l of round `it` takes window w0 + 64 it + l, whose 16 bases ARE the word
        for (uint32_t it = 0; it * 64 < n; ++it) {
            bool ok = lane < n - it * 64;
            if constexpr (CHECKBAD) {
                const uint32_t ch = (w0 + it * 64) >> 6;
                const uint64_t b0 = sbad[ch], b1 = sbad[ch + 1];
                const uint64_t x = (b0 >> lane) | (lane ? (b1 << (64u - lane)) : 0ull);   // bad flags of bases i..i+63
                ok = ok && ((uint32_t)x & 0xFFFFu) == 0u;
            }
            uint32_t rank = INVALID;
            if (ok) rank = direct_rank<I>(im, mmer_l(lb.swl + ((w0 + it * 64) >> 4), 0, lb.sh));
            on_rank(rank);                                                   // itree.c:929-931
        }
        return;
    }
    static_assert(TILE + 64 <= (1u << MIN_LOW_BITS), "tile positions must fit the key's position field");
    uint32_t *Kp = (uint32_t *)Kk;                                          // the wave's 8 * (TILE + 128) bytes: keys, then hashes
    uint32_t *Kl = Kp + lane, *Hl = Kp + (TILE + 128) + lane;
    const uint32_t *Hh = Kp + (TILE + 128);
    uint64_t *Ob = Kk + (TILE + 128) - 8;                                   // orientation ballots of the tile's (at most five) chunks: the hash area's unused tail
    static_assert(((TILE + 4 * W - 16 + 63) / 64) * 64 + 16 <= TILE + 128, "the ballots share the hash area");
    const uint32_t nlane = 0u - lane;
    for (uint32_t wb = 0; wb < n; wb += TILE) {
        const uint32_t tn = n - wb < TILE ? n - wb : TILE;
        const uint32_t *st = lb.swl + ((w0 + wb) >> 4);                      // the tile's first word, per lane
        build_minkeys<W, TILE>(st, lb.sh, Kl, Hl, Ob, tn + K - 16, lane);   // 16-mers of windows w0+wb .. w0+wb+tn-1
        TICK(2);
#if defined(UTREE_ABLATE) && UTREE_ABLATE == 2
        continue;
#endif
        // A round = 64 windows, lane l taking window it*64 + l of the tile.  locate(): the bucket (and the key bits that go with
        // it) of the lane's window in round `it`; resolve(): the rank its bucket holds for the window.
        auto locate = [&](uint32_t it, uint64_t &baddr, uint32_t &tag) {
            const uint32_t *Kr = Kl + it * 64;
            const uint32_t ka = Kr[MinWin<W>::FIRST], kb = Kr[MinWin<W>::FIRST + MinWin<W>::NEXT];
            const uint32_t p = (kb < ka ? kb : ka) & MIN_POS_MASK;
            const uint32_t h = Hh[p];
            const uint32_t o = (uint32_t)(Ob[p >> 6] >> (p & 63u)) & 1u;    // the read holds the canonical 16-mer (0) or its reverse complement
            const uint32_t pos = p + nlane - it * 64;                       // minimizer position inside the window
            const uint64_t re = rg.reg[h >> 24];
            uint32_t bl = __umulhi(h << 8, (uint32_t)re & ((1u << UTREE_REGION_NB_BITS) - 1u));
            if constexpr (W == 16) {
                // the slot's pair: by the four bases around the minimizer, bases pos-2 .. pos+17 of the window being a 20-mer whose ends they are
                const uint32_t *sr = st + it * 4;
                const uint32_t x0 = mmer_l(sr, 0, lb.sh), x1 = mmer_l(sr, 1, lb.sh), x2 = mmer_l(sr, 2, lb.sh), x3 = mmer_l(sr, 3, lb.sh);
                const unsigned __int128 w = ((unsigned __int128)(((uint64_t)x0 << 32) | x1) << 64) | (((uint64_t)x2 << 32) | x3);
                const uint64_t v = (uint64_t)(w >> (92u - 2u * pos));                  // its low 40 bits: the 20-mer
                const uint32_t sub = (uint32_t)(re >> UTREE_REGION_NB_BITS) & ((1u << UTREE_REGION_SUB_BITS) - 1u);
                bl = bl * sub + ((ext_canon((uint32_t)((v >> 32) & 0xF0u) | (uint32_t)(v & 0xFu), o) * sub) >> 8);
            }
#ifdef UTREE_ABLATE_L2
            baddr = rg.table + ((uint64_t)(bl & 0x1FFFu) << rg.bshift);     // timing experiment: every bucket inside 1 MB (L2 hits), answers wrong
#else
            baddr = rg.table + ((2u * ((re >> UTREE_REGION_BASE_SHIFT) + bl) + o) << rg.bshift);
#endif
            tag = ((h & 0xFFu) << 6) | pos;                                 // hash bits the bucket does not imply | position (< 64)
        };
        auto resolve = [&](uint32_t it, const Bucket<W, I> &bk, uint64_t baddr, uint32_t tag) -> uint32_t {
            const uint32_t *sr = st + it * 4;
            const uint32_t pos = tag & 63u, hlow = tag >> 6;
            if constexpr (W == 8) {
                const uint32_t x0 = mmer_l(sr, 0, lb.sh), x1 = mmer_l(sr, 1, lb.sh);
                const uint32_t M = (uint32_t)(0xFFFFFFFFull >> (2 * pos));           // ones over the bases after the minimizer
                const uint32_t rest = x0 ^ ((x0 ^ x1) & M);                          // bit-select: x1 where M, x0 elsewhere
                return resolve_bucket8<I, EXC, OFF>(im, bk, baddr, hlow, pos, rest, x0, x1);
            } else {
                const uint32_t x0 = mmer_l(sr, 0, lb.sh), x1 = mmer_l(sr, 1, lb.sh), x2 = mmer_l(sr, 2, lb.sh), x3 = mmer_l(sr, 3, lb.sh);
                const uint64_t wh = ((uint64_t)x0 << 32) | x1, wl = ((uint64_t)x2 << 32) | x3;
                uint32_t rh; uint64_t rl;
                min_rest<W>(wh, wl, pos, rh, rl);
                MinKey<W> mk;
                mk.lo = rl; mk.hi = ((uint64_t)hlow << 38) | ((uint64_t)pos << 32) | rh;
                return resolve_bucket<W, I, EXC, OFF>(im, bk, baddr, mk, wh, wl);
            }
        };
        auto window_ok = [&](uint32_t it, uint32_t left) -> bool {
            bool ok = lane < left;
            if constexpr (CHECKBAD) {
                const uint32_t ch = (w0 + wb + it * 64) >> 6;               // the round's first base is 64-aligned: bit = lane
                const uint64_t b0 = sbad[ch], b1 = sbad[ch + 1];
                const uint64_t x = (b0 >> lane) | (lane ? (b1 << (64u - lane)) : 0ull);   // bad flags of bases i..i+63
                ok = ok && ((K == 64) ? (x == 0) : ((uint32_t)x == 0));
            }
            return ok;
        };
        // PAIRS: two rounds' buckets are requested together -- one memory wait per 128 windows instead of two -- at the price of
        // 16 more registers (7 instead of 8 waves per SIMD).  Otherwise one bucket is in flight per lane.
        constexpr uint32_t STEP = PAIRS ? 2 : 1;
        for (uint32_t it = 0; it * 64 < tn; it += STEP) {
            const uint32_t left0 = tn - it * 64, left1 = left0 > 64 ? left0 - 64 : 0;
            const bool ok0 = window_ok(it, left0);
            uint64_t a0 = 0; uint32_t t0 = 0;
            Bucket<W, I> bk0;
            if (ok0) { locate(it, a0, t0); bk0 = load_bucket_at<W, I>(a0); }
            TICK(3);
            if constexpr (PAIRS) {
                const bool ok1 = left1 && window_ok(it + 1, left1);
                uint64_t a1 = 0; uint32_t t1 = 0;
                Bucket<W, I> bk1;
                if (ok1) { locate(it + 1, a1, t1); bk1 = load_bucket_at<W, I>(a1); }
                PH_WAITVM;
                TICK(4);
                uint32_t rank0 = INVALID, rank1 = INVALID;
                if (ok0) rank0 = resolve(it, bk0, a0, t0);
                if (ok1) rank1 = resolve(it + 1, bk1, a1, t1);
                TICK(5);
                on_rank(rank0);                                              // itree.c:929-931
                if (left1) on_rank(rank1);
                TICK(6);
            } else {
                PH_WAITVM;
                TICK(4);
                uint32_t rank0 = INVALID;
                if (ok0) rank0 = resolve(it, bk0, a0, t0);
                TICK(5);
                on_rank(rank0);                                              // itree.c:929-931
                TICK(6);
            }
        }
        wave_lds_fence();
    }
}

// the image's region table -> LDS in the window loop's form (all threads of the workgroup; followed by a barrier)
__device__ __forceinline__ void stage_regions(const utk_image &im, uint64_t *s_reg) {
    for (uint32_t x = threadIdx.x; x < 256; x += blockDim.x) s_reg[x] = im.regions[x];
    __syncthreads();
}

// `n` entries of (rank, count) list space from the batch's cursor (one lane calls this).  The workspace is sized so that the space
// cannot run out (dev_image.c: carve); should it -- a caller's total_bases that does not describe the batch, or the test hook --, the
// batch's error word is raised (utree_classify_poll reports it) and the caller writes at the start of the list area instead of past its
// end.
__device__ __forceinline__ unsigned long long reserve_tally(const utk_workspace &ws, unsigned long long n) {
    unsigned long long nb = atomicAdd(&ws.cursors[0], n);
    if (nb + n > ws.tally_cap) { ws.cursors[UTREE_CUR_ERROR] = UTREE_DEVERR_TALLY_CAP; nb = 0; }
    return nb;
}

__device__ __forceinline__ void store_result(utree_result *out, uint32_t label, int32_t cut, uint32_t found,
                                             uint32_t uix, uint32_t sl, uint32_t ol) {
    uint32_t *o = (uint32_t *)out;
    // the six words are materialised here: hoisted out of the read loop as constant vectors they cost six registers per call
    // site for the whole kernel (and went to scratch at 64 registers)
    uint32_t v0 = label, v1 = (uint32_t)cut, v2 = found, v3 = uix, v4 = sl, v5 = ol;
    asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5));
    o[0] = v0; o[1] = v1; o[2] = v2; o[3] = v3; o[4] = v4; o[5] = v5;
}

// ------------------------------------------------------------------------------------------------
// route_k: only launched when a batch may hold reads beyond SHORT_CAP.  Lists them for the mid-length pass
// (<= MID_CAP staged bases) or for classify_long_k, with one atomic per wavefront of 64 reads.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void route_k(const uint32_t *__restrict__ len, uint32_t n_reads, int do_rc, utk_workspace ws) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t total = 0;
    if (r < n_reads) total = do_rc ? 2 * (uint64_t)len[r] + 1 : len[r];
    const bool mid = total > ws.short_cap && total <= ws.mid_limit, lng = total > ws.mid_limit;
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
// fetch_raw / stage_read: a read's bytes -> LDS as they are (global_load_lds: no registers are held while they fly, so the
// NEXT read's bytes travel during the current read's window lookups), then -> 2-bit codes packed big-endian in LDS (sb:
// one byte per 4 bases) and one "bad base" bit per base (sbadb: bit j%8 of byte j/8), FOUR bases per lane.  With do_rc
// the staged sequence is the read, one separator (a bad base: itree.c:1005-1012 never lets a window span both strands)
// and the reverse complement: staged base j > L is the complement of source base 2L - j -- both strands come from the
// same raw bytes.
//
// fetch_raw copies the ALIGNED dwords that hold the read (a dword is only touched when it holds at least one byte of the
// read, so no load leaves the caller's buffer): raw[1 + d] = dword d, source byte s is raw byte 4 + mf + s with
// mf = (address of the read's first byte) & 3; raw[0] is a dummy so that the reverse strand's last group may reach "before" the read.
// stage_read: a lane owns the staged bases 4g..4g+3; its four unaligned source bytes come from two raw dwords through
// v_alignbyte.  Coding is byte-parallel: (b >> 1) & 3 maps A C T G (either case) to 0 1 2 3 and everything else
// somewhere; v_perm turns that back into the letter it stands for, and a base is bad when that is not the (upper-cased)
// input byte (itree.c:110-121).  code = g ^ (g >> 1) gives A=0 C=1 G=2 T=3; the four codes of a group are gathered into
// one byte by a multiplication whose partial products do not overlap.
// ------------------------------------------------------------------------------------------------
template <int CAP> struct RawBuf { static constexpr uint32_t DW = ((CAP / 4 + 1 + 63) / 64) * 64; };   // dwords a read may span, per 64 lanes

__device__ __forceinline__ uint32_t low_bytes(uint32_t n) {            // 0xFF in the n lowest bytes, n clamped to 0..4
    return n >= 4u ? 0xFFFFFFFFu : ((1u << (8u * n)) - 1u);
}

template <int CAP>
__device__ __forceinline__ void fetch_raw(const uint8_t *__restrict__ bases, uint64_t o, uint32_t L, uint32_t *raw, uint32_t lane) {
    const uint32_t mf = (uint32_t)(uintptr_t)(bases + o) & 3u;         // the caller's buffer itself need not be aligned
    const uint32_t nd = (L + mf + 3u) >> 2;                            // <= CAP/4 + 1
    const uint8_t *p = bases + o - mf;
    for (uint32_t d0 = 0; d0 < nd; d0 += 64) {
        const uint32_t d = d0 + lane;
        if (d < nd) {
            // Written as assembly on purpose: the compiler makes every later LDS access wait for an LDS-DMA load it knows
            // about (it cannot tell which LDS bytes the load writes), which would stall the window phase at its first LDS
            // read.  This load is invisible to its counters; the consumer waits explicitly (s_waitcnt vmcnt(0) before
            // stage_read).  Memory operations retire in order, so the compiler's own vmcnt waits stay sufficient.
            const uint8_t *src = p + 4ull * d;
            const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(raw + 1 + d0);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(src), "s"(dst) : "memory");   // m0: reserved, nothing else in this file uses it
        }
    }
}

// Returns whether any staged base (index < total) is bad -- wave-uniform; when none is, the window loop skips its per-window test.
template <int CAP>
__device__ __forceinline__ bool stage_read(const uint32_t *raw, uint32_t mf, uint32_t L, uint32_t total, int do_rc,
                                           uint8_t *sb, uint8_t *sbadb, uint32_t lane) {
    constexpr uint32_t DW = RawBuf<CAP>::DW;
    const uint32_t ngroups = (total + 3u) >> 2;
    bool any = false;
    const uint32_t mr = (mf + 2u * L + 1u) & 3u;                       // reverse: misalignment of a group's lowest source byte (2L - 3 - 4g)
    for (uint32_t g0 = 0; g0 < ngroups; g0 += 64) {
        const uint32_t g = g0 + lane;
        const uint32_t gi = g < DW ? g : DW - 1u;
        const uint32_t fm = low_bytes(4u * g < L ? L - 4u * g : 0u);                       // staged bytes with j < L
        uint32_t word = __builtin_amdgcn_alignbyte(raw[2u + gi], raw[1u + gi], mf) & fm;    // source bytes 4g .. 4g+3
        uint32_t rm = 0;
        if (do_rc) {
            // source bytes q .. q+3 with q = 2L - 3 - 4g >= -4, staged in reverse order
            int32_t i4 = ((int32_t)(mf + 2u * L) - 3 - (int32_t)(4u * g)) >> 2;
            i4 = i4 < -1 ? -1 : (i4 > (int32_t)DW - 1 ? (int32_t)DW - 1 : i4);
            const uint32_t w = __builtin_amdgcn_alignbyte(raw[2 + i4], raw[1 + i4], mr);
            const uint32_t rev = __builtin_amdgcn_perm(0u, w, 0x00010203u);
            // staged bytes with L < j <= 2L
            const uint32_t from = L + 1u > 4u * g ? L + 1u - 4u * g : 0u, to = 2u * L + 1u > 4u * g ? 2u * L + 1u - 4u * g : 0u;
            rm = low_bytes(to) & ~low_bytes(from);
            word |= rev & rm;
        }
        const uint32_t g2 = (word >> 1) & 0x03030303u;
        const uint32_t letter = __builtin_amdgcn_perm(0u, 0x47544341u, g2);            // 0 1 2 3 -> A C T G
        const uint32_t z = (word & 0xDFDFDFDFu) ^ letter;                              // non-zero byte = bad base
        const uint32_t nz = (((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;
        any = any || (nz & low_bytes(total > 4u * g ? total - 4u * g : 0u)) != 0u;     // bytes past the read's end do not count
        const uint32_t nib = (nz * 0x00204081u) >> 28;                                 // bits 7,15,23,31 -> 0..3
        const uint32_t nib_next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nib, 0x101, 0xF, 0xF, true);   // row_shl:1 = lane + 1
        uint32_t code = g2 ^ ((g2 >> 1) & 0x01010101u);
        code ^= rm & 0x03030303u;                                                      // complement on the reverse strand
        const uint32_t packed = (code * 0x40100401u) >> 24;                            // c0<<6 | c1<<4 | c2<<2 | c3
        if (g < ngroups) {
            sb[g ^ 3u] = (uint8_t)packed;
            if (!(lane & 1u)) sbadb[g >> 1] = (uint8_t)(nib | (nib_next << 4));
        }
    }
    return __ballot(any) != 0ull;
}

// ------------------------------------------------------------------------------------------------
// classify_short: one wavefront per read (reads whose staged length fits UTREE_SHORT_CAP bases)
// ------------------------------------------------------------------------------------------------
constexpr int SHORT_CAP = UTREE_SHORT_CAP;          // 150 bp + reverse strand fits
constexpr int SHORT2_CAP = UTREE_SHORT2_CAP;        // 300 bp + reverse strand fits (used when a batch's longest read needs it)
constexpr int MID_CAP = UTREE_MID_CAP;              // 1 kb + reverse strand fits; longer reads take classify_long_k
constexpr int WAVES_PER_BLOCK = 4;
#ifndef UTREE_WORK_GRAB
#define UTREE_WORK_GRAB 32
#endif
constexpr uint32_t WORK_GRAB = UTREE_WORK_GRAB;                   // reads a wave takes per visit to a work counter (16: +0.3 %, 64: +2 %, 128: +7 % time)
constexpr uint32_t TALLY_CHUNK = UTREE_TALLY_CHUNK;
constexpr uint32_t TALLY_DIRECT = UTREE_TALLY_CHUNK / 16;   // hit lists this long get their own reservation
constexpr int32_t CUT_PENDING = -3;                 // result.cut while a read waits for vote_k
constexpr int32_t RANK_PENDING = -4;                // one distinct label: result.label holds its RANK until vote_k looks up the
                                                    // label index (a dependent load classify_short_k would otherwise wait for)

// 8 waves/SIMD for u16-label databases measured faster than 5 even with a few spilled dwords (k = 32: +4 % in r01;
// k = 64: 932 -> 1020 M reads/s, same-box); with u32 labels the hit list's LDS caps the occupancy at 5-6 anyway.  (Voting inside this kernel, 64 parked reads per
// wave, was tried and measured slower at every occupancy: 295-331 vs 343 M reads/s with the separate vote_k.)
// CAP = staged bases a wavefront's LDS slice holds.  CAP = SHORT_CAP walks all reads of the batch and routes the
// longer ones to the mid / long lists; CAP = MID_CAP (LISTED) walks the mid list.  Its 37 KB of LDS per
// workgroup allow 4 workgroups per CU, so it may use 128 VGPRs.
template <int W, int I, bool EXC, typename OFF, int CAP, bool LISTED, int RCMODE = 2>   // RCMODE 0 / 1: strand handling known at compile time
#ifndef UTREE_SHORT_MIN_WAVES
#define UTREE_SHORT_MIN_WAVES 8
#endif
// (80 scalar registers at 8 waves per SIMD: the surplus sits in the lanes of a vector register, one v_readlane per use; lifting the
// cap with amdgpu_waves_per_eu(1, 8) removes those and costs the eighth wave -- measured 9 % slower)
__global__ __launch_bounds__(256, CAP > SHORT2_CAP ? (I == 2 ? 4 : 3) : (I == 2 ? UTREE_SHORT_MIN_WAVES : 5))
void classify_short_k(utk_image im, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ off,
                      const uint32_t *__restrict__ len, uint32_t n_reads, int do_rc_arg, utree_result *__restrict__ out,
                      utk_workspace ws) {
    // (the pass behind the lane-per-read pass, with nothing left over -- the usual batch --: gone before it sets anything up; 25 -> 6 us)
    if constexpr (LISTED) { if (!ws.cursors[UTREE_CUR_MID]) return; }
    // the 150-bp-class kernel is compiled per strand mode: the other mode's code and registers disappear
    const int do_rc = RCMODE == 2 ? do_rc_arg : RCMODE;
    constexpr uint32_t K = 4 * W;
    constexpr int NCH = CAP / 64, NWORDS = CAP / 16 + 6;
    __shared__ uint32_t s_words[WAVES_PER_BLOCK][NWORDS + 2];          // two pad words in front: lane_bits() may look one word back
    __shared__ uint64_t s_bad[WAVES_PER_BLOCK][NCH + 2];
    using HIT = typename std::conditional<I == 2, uint16_t, uint32_t>::type;     // ranks of u16-label databases fit 16 bits
    __shared__ HIT s_hits[WAVES_PER_BLOCK][CAP];
    // a 150 bp read with its reverse strand (270 windows) is two tiles of 256 instead of three of 128
    constexpr uint32_t TILE = CAP <= SHORT_CAP ? 256u : KEY_TILE;
    // two rounds' buckets in flight at once (one memory wait per 128 windows): 16 more registers, 7 waves per SIMD -- measured
    // 5 % slower than one bucket at 8 waves (same-box, round 2); -DUTREE_PAIRS builds it for comparison
#ifdef UTREE_PAIRS
    constexpr bool PAIRS = CAP <= SHORT2_CAP && I == 2;
#else
    constexpr bool PAIRS = false;
#endif
    __shared__ uint64_t s_keys[WAVES_PER_BLOCK][TILE + 128];
    __shared__ uint64_t s_reg[256];
    // the mid-length pass has no room for a raw buffer of its own and less to gain: its raw bytes pass through the hit
    // list's space (consumed by stage_read before the first hit is written) and are not requested ahead
    constexpr bool PREFETCH = CAP <= SHORT2_CAP;
    static_assert(PREFETCH || sizeof(HIT) * CAP >= 4 * (RawBuf<CAP>::DW + 4), "raw bytes must fit the hit list");
    __shared__ uint32_t s_raw[PREFETCH ? WAVES_PER_BLOCK : 1][PREFETCH ? RawBuf<CAP>::DW + 4 : 1];
    stage_regions(im, s_reg);
    const RegionLds rg = {s_reg, (uint64_t)(uintptr_t)im.table, im.bucket_words == 16u ? 7u : 6u};
    const uint32_t lane = lane_id();
    const uint32_t wv = uni32(threadIdx.x >> 6);
    uint32_t *raw = PREFETCH ? s_raw[wv] : (uint32_t *)s_hits[wv];
    uint32_t *sw = s_words[wv] + 2;
    const LaneBits lb = lane_bits(sw, lane);
    uint8_t *sb = (uint8_t *)sw;
    uint64_t *sbad = s_bad[wv];
    HIT *hits = s_hits[wv];
    uint64_t *Kk = s_keys[wv];
    const uint32_t wave_gid = blockIdx.x * WAVES_PER_BLOCK + wv, n_waves = gridDim.x * WAVES_PER_BLOCK;
    unsigned long long chunk_base = 0;
    uint32_t chunk_left = 0;

    // Reads are handed out dynamically, WORK_GRAB at a time per wave (one atomic per grab): the grid need not match
    // the kernel's residency and long and short reads balance out.
    const uint32_t n_items = LISTED ? (uint32_t)ws.cursors[UTREE_CUR_MID] : n_reads;
    // work counters: one per contiguous part of the items (utree_internal.h); this wave starts at its own part
    unsigned long long *parts = ws.cursors + 64 + (LISTED ? UTREE_WORK_PARTS * UTREE_WORK_STRIDE : 0);
    const uint32_t part_len = ((n_items + UTREE_WORK_PARTS - 1) / UTREE_WORK_PARTS + WORK_GRAB - 1) / WORK_GRAB * WORK_GRAB;
    uint32_t part = wave_gid % UTREE_WORK_PARTS, parts_left = UTREE_WORK_PARTS;
    (void)n_waves;
    // Software pipeline over the reads of a grab.  A read's bytes are requested one read ahead (fetch_raw, into LDS) and
    // become its packed form (stage_read) right after the PREVIOUS read's window lookups, before that read's tally and
    // stores: at that point every load of the wave has been waited for anyway, so the bytes are there without a further
    // wait -- in particular without waiting for the previous read's stores to be acknowledged.  The pipeline restarts at
    // every grab.
    auto stageable = [&](uint32_t len_) { const uint64_t t = do_rc ? 2 * (uint64_t)len_ + 1 : len_; return t <= (uint64_t)CAP && t >= K; };
    auto stage = [&](uint64_t o_, uint32_t L_) -> bool {   // raw bytes -> 2-bit codes packed big-endian in LDS, bad-base bits; any bad base?
        const uint32_t total_ = do_rc ? 2 * L_ + 1 : L_;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_fence();
        const bool any_ = stage_read<CAP>(raw, (uint32_t)(uintptr_t)(bases + o_) & 3u, L_, total_, do_rc, sb, (uint8_t *)sbad, lane);
        if (lane == 0) sbad[(total_ + 63) >> 6] = ~0ull;
        wave_lds_fence();
        return any_;
    };
    // ---- tally (itree.c:1028-1040) of a read with F hits in hits[]: unique labels with counts, ascending rank = strcmp order ----
    auto finish = [&](uint32_t r, uint32_t F) {
            if (F == 0) { if (lane == 0) store_result(&out[r], 0, -2, 0, 0, 0, 0); return; }
            const uint32_t h0 = hits[0];
            if (F == 1) { if (lane == 0) store_result(&out[r], h0, RANK_PENDING, 1, 1, 0, 0); return; }
            if (F <= 64) {
                // Up to one hit per lane: peel off distinct labels with readlane + ballot (no LDS shuffles).
                const bool mine = lane < F;
                const uint32_t hv = mine ? hits[lane] : 0u;
                uint64_t left = __ballot(mine);
                uint32_t nu = 0, myv = 0, myc = 0;                   // lane u keeps distinct label u and its count
                while (left) {
                    const uint32_t lead = (uint32_t)__builtin_ctzll(left);
                    const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)hv, (int)lead);
                    const uint64_t m = __ballot(mine && hv == v) & left;
                    if (lane == nu) { myv = v; myc = (uint32_t)__popcll(m); }
                    ++nu; left &= ~m;
                }
                if (nu == 1) { if (lane == 0) store_result(&out[r], h0, RANK_PENDING, F, 1, 0, 0); return; }
                if (nu > chunk_left) {
                    unsigned long long nb = 0;
                    if (lane == 0) nb = reserve_tally(ws, (unsigned long long)TALLY_CHUNK);
                    chunk_base = uni64(nb);
                    chunk_left = TALLY_CHUNK;
                }
                // ascending rank = strcmp order (itree.c:1041): lane u's place = distinct labels smaller than its own
                uint32_t place = 0;
                for (uint32_t u = 0; u < nu; ++u) place += (uint32_t)__builtin_amdgcn_readlane((int)myv, (int)u) < myv;
                if (lane < nu) ws.tally[chunk_base + place] = (uint64_t)myv | ((uint64_t)myc << 32);
                if (lane == 0) store_result(&out[r], 0, CUT_PENDING, F, nu, (uint32_t)chunk_base, (uint32_t)(chunk_base >> 32));
                chunk_base += nu; chunk_left -= nu;
                return;
            }
            // many hits (long or densely covered reads): distinct labels by repeated wave-min over the LDS list
            uint32_t mn = INVALID, mx = 0;
            for (uint32_t j = lane; j < F; j += 64) { uint32_t h = hits[j]; mn = h < mn ? h : mn; mx = h > mx ? h : mx; }
            mn = wave_min_u32(mn);
            mx = ~wave_min_u32(~mx);
            if (mn == mx) { if (lane == 0) store_result(&out[r], h0, RANK_PENDING, F, 1, 0, 0); return; }
            // (rank,count) list space: every wave sub-allocates from TALLY_CHUNK-entry chunks it reserves with ONE atomic
            // (a per-read atomic on one address serialises the whole chip).  A refill abandons < TALLY_DIRECT entries of
            // the old chunk; reads with more hits than that take their space directly, so the workspace bound
            // windows * 9/8 + waves * TALLY_CHUNK (dev_image.c) always holds.
            unsigned long long base;
            const bool direct = (uint32_t)CAP >= TALLY_DIRECT && F >= TALLY_DIRECT;   // (a 320-base slice never holds that many hits)
            if (direct) {
                unsigned long long nb = 0;
                if (lane == 0) nb = reserve_tally(ws, (unsigned long long)F);
                base = uni64(nb);
            } else {
                if (F > chunk_left) {
                    unsigned long long nb = 0;
                    if (lane == 0) nb = reserve_tally(ws, (unsigned long long)TALLY_CHUNK);
                    chunk_base = uni64(nb);
                    chunk_left = TALLY_CHUNK;
                }
                base = chunk_base;
            }
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
            if (!direct) { chunk_base += uix; chunk_left -= uix; }
            // vote_k finishes this read: cut = CUT_PENDING marks it, sl/ol carry the tally offset
            if (lane == 0) store_result(&out[r], 0, CUT_PENDING, F, uix, (uint32_t)base, (uint32_t)(base >> 32));
    };
    uint32_t item = 0, item_end = 0;
    PH_DECL
    bool ahead = false;                                    // the current read's parameters are known and, if stageable, it is staged
    bool bad_cur = true, bad_next = true;                  // a staged read holds a bad base (the strands' separator counts)
    uint32_t r = 0, L = 0;
    uint64_t o = 0;
    // the grab's read parameters: lane i holds read (grab start + i)'s index, length and offset -- one round trip per grab
    // instead of one per read (each of those stalled the wave: a load's result needs every earlier store acknowledged too)
    uint32_t grab0 = 0, g_r = 0, g_len = 0;
    uint64_t g_off = 0;
    auto params = [&](uint32_t it_, uint32_t &r_, uint32_t &L_, uint64_t &o_) {
        const int k_ = (int)(it_ - grab0);
        r_ = LISTED ? (uint32_t)__builtin_amdgcn_readlane((int)g_r, k_) : it_;
        L_ = (uint32_t)__builtin_amdgcn_readlane((int)g_len, k_);
        o_ = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(g_off >> 32), k_) << 32) |
             (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)g_off, k_);
    };
    static_assert(WORK_GRAB <= 64, "a grab's parameters live in the lanes of one register");
    for (;;) {
        if (item == item_end) {
            // next grab: from this wave's current part; a part that is used up (a plain load tells, no atomic) is left for good
            bool got = false;
            while (parts_left) {
                unsigned long long *ctr = parts + part * UTREE_WORK_STRIDE;
                const uint64_t lo = (uint64_t)part * part_len;
                const uint32_t avail = lo >= n_items ? 0u : (uint32_t)(n_items - lo < part_len ? n_items - lo : part_len);
                unsigned long long g = ~0ull;
                if (lane == 0 && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < avail)
                    g = atomicAdd(ctr, (unsigned long long)WORK_GRAB);
                const uint32_t taken = uni32((uint32_t)(g > 0xFFFFFFFFull ? 0xFFFFFFFFull : g));
                if (taken < avail) {
                    item = (uint32_t)lo + taken;
                    item_end = taken + WORK_GRAB < avail ? item + WORK_GRAB : (uint32_t)lo + avail;
                    got = true;
                    break;
                }
                part = part + 1 == UTREE_WORK_PARTS ? 0 : part + 1;
                --parts_left;
            }
            if (!got) break;
            ahead = false;
            grab0 = item;
            if (item + lane < item_end) {
                g_r = LISTED ? ws.mid_list[item + lane] : item + lane;
                g_len = len[g_r]; g_off = off[g_r];
            }
            TICK(0);
        }
        if (!ahead) {
            params(item, r, L, o);
            if (stageable(L)) { fetch_raw<CAP>(bases, o, L, raw, lane); bad_cur = stage(o, L); }
        }
        ++item;
        ahead = PREFETCH && item < item_end;
        uint32_t r_next = 0, L_next = 0;
        uint64_t o_next = 0;
        bool stage_next = false;
        if (ahead) {
            params(item, r_next, L_next, o_next);
            stage_next = stageable(L_next);
            if (stage_next) fetch_raw<CAP>(bases, o_next, L_next, raw, lane);     // the raw buffer is free: this read is staged
        }
        TICK(1);
        const uint64_t total64 = do_rc ? 2 * (uint64_t)L + 1 : L;
        // total64 > CAP: route_k listed it for the mid-length pass or classify_long_k, nothing to do here
        if (total64 <= (uint64_t)CAP) {
            const uint32_t total = (uint32_t)total64;
            uint32_t F = 0;
#if !(defined(UTREE_ABLATE) && UTREE_ABLATE == 1)            /* ablation builds (profiles/run_pmc_variants.sh): 1 = staging only, */
            if (total >= K) {                                      /* 2 = + sliding minimizers, 3 = + window lookups, no tally        */
                // ---- windows: lane l takes windows l, l+64, ... (itree.c:906-933) ----
                auto on_rank = [&](uint32_t rank) {
                    const bool hit = rank != INVALID;
                    const uint64_t hm = __ballot(hit);
                    if (hit) hits[F + lanes_below(hm)] = (HIT)rank;
                    F += (uint32_t)__popcll(hm);
                };
                // a read without any bad base (most) takes the loop without the per-window test; with both strands staged the
                // separator is one, so that instantiation only has the tested loop
                if constexpr (RCMODE == 1) wave_scan_windows<W, I, EXC, OFF, TILE, true, PAIRS>(im, lb, sbad, Kk, 0u, total - K + 1, rg, lane, on_rank PH_PASS);
                else if (!bad_cur) wave_scan_windows<W, I, EXC, OFF, TILE, false, PAIRS>(im, lb, sbad, Kk, 0u, total - K + 1, rg, lane, on_rank PH_PASS);
                else wave_scan_windows<W, I, EXC, OFF, TILE, true, PAIRS>(im, lb, sbad, Kk, 0u, total - K + 1, rg, lane, on_rank PH_PASS);
                wave_lds_fence();
            }
#endif
            TICK(9);
            if (stage_next) bad_next = stage(o_next, L_next);   // sw / sbad now belong to the next read; hits[] to this one
            TICK(7);
#if defined(UTREE_ABLATE)
            if (lane == 0) store_result(&out[r], sw[0] + hits[0], -2, F, 0, 0, 0);
#else
            finish(r, F);                                  // F == 0 (no window: no hit, no output line) included
#endif
            TICK(8);
        } else if (stage_next) bad_next = stage(o_next, L_next);
        r = r_next; L = L_next; o = o_next; bad_cur = bad_next;
    }
#ifdef UTREE_PHASE_TIMERS
    if (lane == 0) { for (int q = 0; q < 12; ++q) atomicAdd(&g_phase[q], ph_.acc[q]); atomicAdd(&g_phase[12], 1ull); }
#endif
}

// ------------------------------------------------------------------------------------------------
// classify_long: one workgroup per read, any length (itree.c:836: lines up to 16 MiB).  The read is walked in
// tiles staged through LDS (one bucket in flight per lane); hits go to a per-workgroup label histogram in
// HBM and set a bit in a touched-label bitmap; the bitmap is then swept in rank order, which yields the same
// sorted (rank,count) list the wave kernel emits, and only touched histogram entries are read and cleared.
// ------------------------------------------------------------------------------------------------
constexpr int LONG_TILE = 2048;                       // windows per tile: 512 per wave
constexpr int LONG_THREADS = 256;
constexpr uint32_t LONG_LDS_BITWORDS = 2048;          // labels whose bitmap fits LDS (65 536); else a bitmap in HBM

// 8 waves per SIMD (64 registers, no scratch) against the compiler's own choice of 5 (81 registers): 18-22 % faster, same-box;
// the instantiations with the reference-exact probe path (EXC: irregular bins) would spill and keep 5.
template <int W, int I, bool EXC, typename OFF>
__global__ __launch_bounds__(LONG_THREADS, EXC ? 5 : 8) void classify_long_k(utk_image im, const uint8_t *__restrict__ bases,
                                                                const uint64_t *__restrict__ off,
                                                                const uint32_t *__restrict__ len, int do_rc,
                                                                utree_result *__restrict__ out, utk_workspace ws) {
    constexpr uint32_t K = 4 * W;
    constexpr uint32_t STAGE = LONG_TILE + 64;          // bases staged per tile (tile + K-1, rounded up)
    __shared__ uint32_t s_words_pad[STAGE / 16 + 8 + 2];  // two pad words in front: lane_bits() may look one word back
    __shared__ uint64_t s_bad[STAGE / 64 + 2];
    __shared__ uint32_t s_scan[LONG_THREADS / 64 + 1];
    __shared__ uint32_t s_touch[LONG_LDS_BITWORDS];
    __shared__ uint64_t s_keys[LONG_THREADS / 64][KEY_TILE + 128];
    __shared__ unsigned long long s_base;
    __shared__ uint64_t s_reg[256];
    __shared__ uint32_t s_work, s_single;               // the read this workgroup took; the label of a one-label read
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = uni32(tid >> 6);
    uint32_t *s_words = s_words_pad + 2;
    uint8_t *sb = (uint8_t *)s_words;
    const LaneBits lb = lane_bits(s_words, lane);
    const uint32_t nl = im.n_labels, nbw = (nl + 31) >> 5;
    uint32_t *hist = ws.hist + (size_t)blockIdx.x * nl;                       // all zero between reads
    uint32_t *touch = nbw <= LONG_LDS_BITWORDS ? s_touch : ws.touch + (size_t)blockIdx.x * nbw;   // all zero between reads
    const uint32_t n_long = (uint32_t)ws.cursors[UTREE_CUR_LONG];
    if (nbw <= LONG_LDS_BITWORDS) for (uint32_t x = tid; x < nbw; x += LONG_THREADS) s_touch[x] = 0;
    stage_regions(im, s_reg);
    const RegionLds rg = {s_reg, (uint64_t)(uintptr_t)im.table, im.bucket_words == 16u ? 7u : 6u};

    for (;;) {
        // long reads differ in length by orders of magnitude: hand them out one at a time
        __syncthreads();                                  // everyone is done with the previous read (and with s_work)
        if (tid == 0) s_work = (uint32_t)atomicAdd(&ws.cursors[UTREE_CUR_WORK_LONG], 1ull);
        __syncthreads();
        const uint32_t li = s_work;
        if (li >= n_long) break;
        const uint32_t r = ws.long_list[li];
        const uint64_t L64 = len[r];
        const uint64_t o = off[r];
        const uint64_t total = do_rc ? 2 * L64 + 1 : L64;
        const uint64_t nwin = total >= K ? total - K + 1 : 0;
        uint32_t my_hits = 0;
        if (tid == 0) s_single = INVALID;
        for (uint64_t w0 = 0; w0 < nwin; w0 += LONG_TILE) {
            // stage bases [w0, w0+STAGE): a wave's byte loads are all issued before the first is used (one memory wait per tile,
            // not one per 64 bases).  Fetching them a tile ahead, during the previous tile's lookups, costs the registers that
            // keep 5 waves per SIMD: measured 4-13 % slower.
            constexpr uint32_t NW = LONG_THREADS / 64, PER = (STAGE / 64 + NW - 1) / NW;
            uint32_t rawb[PER];
#pragma unroll
            for (uint32_t q = 0; q < PER; ++q) {
                const uint32_t c = wv + q * NW;
                const uint64_t j = w0 + (uint64_t)c * 64 + lane;
                rawb[q] = 0x200u;                                              // no base here: bad
                if (c < STAGE / 64) {
                    if (j < L64) rawb[q] = bases[o + j];
                    else if (j > L64 && j < total) rawb[q] = 0x100u | bases[o + (2 * L64 - j)];   // reverse strand: complement
                }
            }
#pragma unroll
            for (uint32_t q = 0; q < PER; ++q) {
                const uint32_t c = wv + q * NW;
                if (c >= STAGE / 64) break;
                uint32_t code = 0; bool bad = true;
                if (!(rawb[q] & 0x200u)) { base_code(rawb[q] & 0xFFu, code, bad); code ^= (rawb[q] >> 8) * 3u; }
                uint64_t bm = __ballot(bad);
                uint32_t t = (code << 2) | (uint32_t)__shfl_down((int)code, 1);
                uint32_t u = (t << 4) | (uint32_t)__shfl_down((int)t, 2);
                if ((lane & 3u) == 0) sb[(c * 16 + (lane >> 2)) ^ 3u] = (uint8_t)u;
                if (lane == 0) s_bad[c] = bm;
            }
            if (tid == 0) s_bad[STAGE / 64] = ~0ull;
            __syncthreads();
            const uint32_t tile_n = (uint32_t)(nwin - w0 < LONG_TILE ? nwin - w0 : LONG_TILE);
            // every wave takes a quarter of the tile and runs the same window scan as the wave-per-read kernel
            constexpr uint32_t PER_WAVE = LONG_TILE / (LONG_THREADS / 64);
            const uint32_t a = wv * PER_WAVE < tile_n ? wv * PER_WAVE : tile_n;
            const uint32_t b = a + PER_WAVE < tile_n ? a + PER_WAVE : tile_n;
            PH_DECL
            wave_scan_windows<W, I, EXC, OFF, KEY_TILE, true, false>(im, lb, s_bad, s_keys[wv], a, b - a, rg, lane, [&](uint32_t rank) {
                // one atomic per DISTINCT label of the 64 windows, not per hit: a read's hits mostly share a few labels,
                // and 64 atomics on one address serialise in L2
                const bool hit = rank != INVALID;
                uint64_t left = __ballot(hit);
                my_hits += hit;
                while (left) {
                    const uint32_t lead = (uint32_t)__builtin_ctzll(left);
                    const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)rank, (int)lead);
                    const uint64_t m = __ballot(hit && rank == v) & left;
                    if (lane == lead) {
                        atomicAdd(&hist[v], (uint32_t)__popcll(m));
                        atomicOr(&touch[v >> 5], 1u << (v & 31u));
                    }
                    left &= ~m;
                }
            } PH_PASS);
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
        if (tid == 0) s_base = uix > 1 ? reserve_tally(ws, (unsigned long long)uix) : 0ull;
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
                else s_single = rk;
                ++pos;
            }
        }
        __syncthreads();
        if (tid == 0) {
            if (uix == 1) store_result(&out[r], im.rank2ix[s_single], -2, F, 1, 0, 0);
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

// bytes p .. p+15 of a label with ONE 16-byte request (labels are byte strings: the load is unaligned)
__device__ __forceinline__ void label16(const char *s, uint32_t p, uint64_t &x, uint64_t &y) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2), aligned(1)));
    const u64x2 v = *(const u64x2 *)(s + p);
    x = v.x; y = v.y;
}

// exact per-byte flags (0x80 in the byte): x1 byte == 0, x1 byte == ';', or x1 and x2 bytes differ; the lowest set flag is
// the first place the reference's scan stops (itree.c:1060-1061)
__device__ __forceinline__ uint64_t stop_flags(uint64_t x1, uint64_t x2) {
    const uint64_t semi = x1 ^ 0x3B3B3B3B3B3B3B3Bull, d = x1 ^ x2;
    uint64_t m = (((x1 - 0x0101010101010101ull) & ~x1) | ((semi - 0x0101010101010101ull) & ~semi)) & 0x8080808080808080ull;
    m |= (((d & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | d) & 0x8080808080808080ull;
    return m;
}

// One pair of the reference's scan (itree.c:1060-1061) from byte `start` of s1 on: td = the first index where s1 ends, differs from
// s2 or is ';'; a, b = the bytes of s1, s2 there; before = s1[td - 1] (0 for td == 0: "not '_'").
__device__ __forceinline__ void pair_scan(const char *s1, const char *s2, uint32_t start, uint32_t &td, uint32_t &a, uint32_t &b, uint32_t &before) {
    uint32_t base = start, prevlast = start ? (uint32_t)(uint8_t)s1[start - 1] : 0u;
    for (;;) {
        uint64_t x1, y1, x2, y2;
        label16(s1, base, x1, y1); label16(s2, base, x2, y2);
        uint64_t m = stop_flags(x1, x2);
        if (!m) { prevlast = (uint32_t)(x1 >> 56); base += 8; x1 = y1; x2 = y2; m = stop_flags(x1, x2); }
        if (m) {
            const uint32_t idx = (uint32_t)(__builtin_ctzll(m) >> 3);
            td = base + idx;
            a = (uint32_t)(x1 >> (8 * idx)) & 0xFFu; b = (uint32_t)(x2 >> (8 * idx)) & 0xFFu;
            before = idx ? ((uint32_t)(x1 >> (8 * idx - 8)) & 0xFFu) : prevlast;
            return;
        }
        prevlast = (uint32_t)(x1 >> 56); base += 8;
    }
}

// A label's utk_vote_rec (utree_internal.h) and its fields at token level t.
struct VRec { uint64_t w[4]; };
__device__ __forceinline__ VRec vrec(const uint64_t *vt, uint32_t rank) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const u64x2 *p = (const u64x2 *)(vt + (size_t)rank * 4);
    const u64x2 a = p[0], b = p[1];
    VRec v; v.w[0] = a.x; v.w[1] = a.y; v.w[2] = b.x; v.w[3] = b.y;
    return v;
}
__device__ __forceinline__ uint32_t v_pid(const VRec &v, uint32_t t) { return (uint32_t)((t < 4 ? v.w[0] : v.w[1]) >> (16u * (t & 3u))) & 0xFFFFu; }
__device__ __forceinline__ uint32_t v_end(const VRec &v, uint32_t t) { return (uint32_t)(v.w[2] >> (8u * t)) & 0xFFu; }
__device__ __forceinline__ bool v_exists(const VRec &v, uint32_t t) { return ((uint32_t)v.w[3] >> t) & 1u; }
__device__ __forceinline__ bool v_more(const VRec &v, uint32_t t) { return ((uint32_t)v.w[3] >> (8u + t)) & 1u; }
__device__ __forceinline__ bool v_us(const VRec &v, uint32_t t) { return ((uint32_t)v.w[3] >> (16u + t)) & 1u; }
__device__ __forceinline__ uint32_t v_len(const VRec &v) { return (uint32_t)(v.w[3] >> 32) & 0xFFu; }   // the label's length
__device__ __forceinline__ uint32_t v_ix(const VRec &v) { return (uint32_t)(v.w[3] >> 48); }             // its file-order index

// The vote with the label table (UTREE_F_VOTE_TABLE): the same state machine as below, every byte scan replaced by what the table says
// about it.  The labels of the active group [st, ed) agree up to byte dv; t is the token that holds byte dv + 1 (0 at the start), the
// same for all of them.  For a pair (prev, cur) of the group the reference's scan from dv + 1 stops at prev's next terminator e1 =
// tok_end[t] unless the two differ before:
//   same id at t (and cur has a token t)   -> both have the bytes up to e1 and a terminator there: td = e1, a / b = ';' or end
//   ids differ                             -> they differ at or before e1.  The decision needs a = prev[td] and before = prev[td - 1]
//        only to tell "less specific" (td = e1 and the byte before is '_': an empty rank like s__) from "differs": unless prev's token
//        ends in '_' the answer is "differs" wherever td lies.  td itself -- the next dv -- matters only when this pair ends the walk
//        with the group winning and the descent going on: then, and for tokens ending in '_', the bytes are read (pair_scan).
__device__ __forceinline__ void vote_table(const utk_image &im, utree_result *out_r, const uint64_t *T, uint32_t F, uint32_t uix) {
    const uint64_t *vt = im.vote_tab;
    const char *blob = im.label_blob;
    const uint32_t *loff = im.label_off;
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    uint32_t cutoff = cut_of(F);
    uint32_t st = 0, ed = uix, dv = NONE, t = 0, orun = F, sl, ol;
    // (every list entry and label record is loaded once per level at most: the group's last entry -- needed for the shared levels -- is
    // also the walk's last `cur`, and whatever ends the descent has its record in registers already)
    uint32_t ix_last, n_last, len_last;                                      // the label the result is cut from: file-order index, count, length
    for (;;) {
        const uint64_t t_st = T[st];
        VRec pv = vrec(vt, (uint32_t)t_st);
        const uint64_t t_ed = ed - 1u == st ? t_st : T[ed - 1u];
        const VRec lv = ed - 1u == st ? pv : vrec(vt, (uint32_t)t_ed);
        {
            // levels at which the whole group carries one token: the group's first and last label agree through token L - 1, each followed
            // by ';' (ids are equal through the last shared token, `more` bits are set through the last ';')
            const uint64_t d0 = pv.w[0] ^ lv.w[0], d1 = pv.w[1] ^ lv.w[1];
            const uint32_t same = d0 ? (uint32_t)__builtin_ctzll(d0) >> 4 : 4u + (d1 ? (uint32_t)__builtin_ctzll(d1) >> 4 : 4u);
            const uint32_t both = ((uint32_t)(pv.w[3] & lv.w[3]) >> 8) & 0xFFu;             // ';' after token in both
            const uint32_t semis = (uint32_t)__builtin_ctz(~both);                          // tokens 0 .. semis - 1 end in ';' in both
            const uint32_t L = same < semis ? same : semis;
            if (L > t) { dv = v_end(pv, L - 1u); t = L; }
        }
        uint32_t run = (uint32_t)(t_st >> 32), td = dv, c1 = run, rp = (uint32_t)t_st;
        bool td_is_end = false;                                                             // td is prev's terminator at level t
        bool stopped = false;
        for (uint32_t z = st + 1; z < ed; ++z) {
            const bool is_last = z == ed - 1u;
            const uint64_t tz = is_last ? t_ed : T[z];
            const uint32_t nz = (uint32_t)(tz >> 32), rc = (uint32_t)tz;
            const VRec cv = is_last ? lv : vrec(vt, rc);
            bool aside = false, stop = false;
            // previous label exhausted at dv (itree.c:1052): it has no token t, or nothing is agreed yet and it is the empty string
            if (t >= 8u || !v_exists(pv, t) || (dv == NONE && v_end(pv, 0) == 0u && !v_more(pv, 0))) aside = true;
            else {
                uint32_t a, b, before;
                bool known = true;
                if (v_exists(cv, t) && v_pid(cv, t) == v_pid(pv, t)) {
                    td = v_end(pv, t); td_is_end = true;
                    a = v_more(pv, t) ? ';' : 0u; b = v_more(cv, t) ? ';' : 0u; before = v_us(pv, t) ? '_' : 'x';
                } else if (v_us(pv, t) || !v_exists(cv, t)) {
                    pair_scan(blob + loff[rp], blob + loff[rc], dv + 1u, td, a, b, before);   // (dv + 1 = 0 while nothing is agreed)
                    td_is_end = td == v_end(pv, t);
                } else { a = 'a'; b = 'b'; before = 'x'; known = false; }                     // they differ inside the token or at its end: "differs" either way
                if (a == b) run += nz;                                                         // same token: itree.c:1062
                else if ((!a && b == ';') || ((a == ';' || !a) && before == '_')) aside = true;   // less specific: 1063
                else if (run >= cutoff) {                                                      // group wins: itree.c:1068
                    ed = z; stop = true;
                    if (!known) {                                                              // the next dv, should the descent go on
                        pair_scan(blob + loff[rp], blob + loff[rc], dv + 1u, td, a, b, before);
                        td_is_end = td == v_end(pv, t);
                    }
                } else { run = nz; st = z; }                                                   // restart: itree.c:1069
            }
            if (aside) {                                                       // 1053-1056 / 1064-1067
                run = nz; st = z;
                orun -= c1;
                cutoff = cut_of(orun);
            }
            if (stop) { stopped = true; break; }
            pv = cv; c1 = nz; rp = rc;
        }
        // the group's last label [ed - 1]: the one before the pair that stopped the walk, else the list's last
        ix_last = stopped ? v_ix(pv) : v_ix(lv);                                // (pv is the record of rank rp)
        n_last = stopped ? c1 : (uint32_t)(t_ed >> 32);
        len_last = stopped ? v_len(pv) : v_len(lv);
        sl = run; ol = orun;                                                   // itree.c:1071
        if (run < cutoff) break;                                               // itree.c:1072
        if (st + 1 >= ed) {                                                    // itree.c:1073-1079
            if (n_last >= cutoff) dv = 0xFFFFFFFEu;
            break;
        }
        orun = run; cutoff = cut_of(run);                                      // itree.c:1082-1085
        if (td != dv) { dv = td; if (td_is_end) ++t; }
    }
    int32_t cut;
    if (dv == NONE) cut = -1;                                                  // itree.c:1087
    else if (dv == 0xFFFFFFFEu) cut = -2;
    else cut = (int32_t)(dv < len_last ? dv : len_last);                       // 1088
    store_result(out_r, ix_last, cut, F, uix, sl, ol);
}

// vote_table_k: vote_k for images with the label table (a kernel of its own: with both in one, the byte scans' registers cost the table path
// three of its eight wavefronts per SIMD)
#ifdef UTREE_VOTE_WPE8
__attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
__global__ __launch_bounds__(256) void vote_table_k(utk_image im, utree_result *__restrict__ out, utk_workspace ws, uint32_t n_reads) {
#ifndef UTREE_VOTE_UNSORTED
    // The vote is a per-lane state machine: a wavefront executes the union of its lanes' paths, and a read of two labels takes a fraction of
    // the steps a read of four takes.  The workgroup's 256 reads are therefore dealt out again by what they need: reads of two labels fill the
    // workgroup's lanes from the bottom, reads of more from the top (the two kinds meet in one wavefront at most); a read of one label or none
    // is finished where it stands.
    __shared__ uint32_t s_n[2][4];
    __shared__ uint16_t s_who[256];
    const uint32_t tid = threadIdx.x, wv = tid >> 6, lane = tid & 63u;
    const uint32_t r0 = blockIdx.x * blockDim.x + tid;
    uint32_t cls = 2;                                      // 0: two labels, 1: more, 2: nothing left to do
    if (r0 < n_reads) {
        const uint32_t *res = (const uint32_t *)&out[r0];
        const int32_t cut = (int32_t)res[1];
        if (cut == RANK_PENDING) {                         // one distinct label: only its file-order index is missing
            uint32_t *o = (uint32_t *)&out[r0];
            o[0] = im.rank2ix[res[0]]; o[1] = (uint32_t)-2;
        } else if (cut == CUT_PENDING) cls = res[3] == 2u ? 0u : 1u;
    }
    const uint64_t m0 = __builtin_amdgcn_ballot_w64(cls == 0u), m1 = __builtin_amdgcn_ballot_w64(cls == 1u);
    s_who[tid] = 0xFFFFu;
    if (lane == 0) { s_n[0][wv] = (uint32_t)__popcll(m0); s_n[1][wv] = (uint32_t)__popcll(m1); }
    __syncthreads();
    if (cls < 2u) {
        uint32_t before = 0;
        for (uint32_t w = 0; w < wv; ++w) before += s_n[cls][w];
        before += (uint32_t)__popcll((cls ? m1 : m0) & ((1ull << lane) - 1ull));
        s_who[cls ? 255u - before : before] = (uint16_t)tid;
    }
    __syncthreads();
    const uint32_t who = s_who[tid];
    if (who == 0xFFFFu) return;
    const uint32_t r = blockIdx.x * blockDim.x + who;
#else
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    {
        const uint32_t *res = (const uint32_t *)&out[r];
        if ((int32_t)res[1] == RANK_PENDING) {             // one distinct label: only its file-order index is missing
            uint32_t *o = (uint32_t *)&out[r];
            o[0] = im.rank2ix[res[0]]; o[1] = (uint32_t)-2;
            return;
        }
        if ((int32_t)res[1] != CUT_PENDING) return;        // finished by the classify kernel (no hit, or classify_long_k)
    }
#endif
    const uint32_t *res = (const uint32_t *)&out[r];
    vote_table(im, &out[r], ws.tally + ((uint64_t)res[4] | ((uint64_t)res[5] << 32)), res[2], res[3]);
}

__global__ __launch_bounds__(256) void vote_k(utk_image im, utree_result *__restrict__ out, utk_workspace ws, uint32_t n_reads) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t *res = (const uint32_t *)&out[r];
    if ((int32_t)res[1] == RANK_PENDING) {                 // one distinct label: only its file-order index is missing
        uint32_t *o = (uint32_t *)&out[r];
        o[0] = im.rank2ix[res[0]]; o[1] = (uint32_t)-2;
        return;
    }
    if ((int32_t)res[1] != CUT_PENDING) return;            // finished by the classify kernel (no hit, or classify_long_k)
    const uint32_t F = res[2], uix = res[3];
    const uint64_t *T = ws.tally + ((uint64_t)res[4] | ((uint64_t)res[5] << 32));
    const char *blob = im.label_blob;
    const uint32_t *loff = im.label_off;
    uint32_t cutoff = cut_of(F);
    uint32_t st = 0, ed = uix, dv = 0xFFFFFFFFu, orun = F, sl, ol;
    for (;;) {
        const uint64_t t_st = T[st];
        {
            // Levels at which every label of the group [st, ed) carries the same token change nothing but dv (each pair
            // stops at the token's ';' with equal bytes: run ends up as orun, st / ed / cutoff stay): dv moves to that
            // ';'.  The list is in strcmp order, so a prefix the group's first and last label share is shared by all of
            // them: ONE scan of those two labels finds the last ';' they share beyond dv, instead of a pass over all
            // pairs per level.  The level that follows is the first one with something to decide.
            const char *sa = blob + loff[(uint32_t)t_st], *sz = blob + loff[(uint32_t)T[ed - 1]];
            uint32_t base = dv + (dv == 0xFFFFFFFFu), q = 0xFFFFFFFFu;
            bool first = dv != 0xFFFFFFFFu;                                   // then byte `base` (= dv) itself does not count
            for (;;) {
                uint64_t a[2], b[2];
                label16(sa, base, a[0], a[1]); label16(sz, base, b[0], b[1]);
                bool stop = false;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint64_t x = a[h], d = a[h] ^ b[h], sx = a[h] ^ 0x3B3B3B3B3B3B3B3Bull;
                    uint64_t end = ((x - 0x0101010101010101ull) & ~x) & 0x8080808080808080ull;               // label ends
                    end |= (((d & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | d) & 0x8080808080808080ull;   // or differs
                    uint64_t semi = ~(((sx & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | sx) & 0x8080808080808080ull;   // ';'
                    if (h == 0 && first) { end &= ~0xFFull; semi &= ~0xFFull; }
                    if (end) semi &= (end & (0ull - end)) - 1ull;             // only those before the first end / difference
                    if (semi) q = base + 8u * (uint32_t)h + ((63u - (uint32_t)__builtin_clzll(semi)) >> 3);
                    if (end) { stop = true; break; }
                }
                if (stop) break;
                base += 16; first = false;
            }
            if (q != 0xFFFFFFFFu) dv = q;
        }
        uint32_t run = (uint32_t)(t_st >> 32), td = dv;
        // (s1, c1) = label text and count of the list entry just before z; it moves along with z
        const char *s1 = blob + loff[(uint32_t)t_st];
        uint32_t c1 = run;
        const uint32_t probe = dv + (dv == 0xFFFFFFFFu);                      // 0 while nothing is agreed, else dv
        const bool skip0 = dv != 0xFFFFFFFFu;                                // then byte `probe` itself is not compared
        // bytes probe..probe+15 of the entry before z: they were this entry's (x2, y2) one step earlier, so they are kept
        uint64_t x1_first, y1_first;
        label16(s1, probe, x1_first, y1_first);
        for (uint32_t z = st + 1; z < ed; ++z) {
            const uint64_t tz = T[z];
            const uint32_t nz = (uint32_t)(tz >> 32);
            const char *s2 = blob + loff[(uint32_t)tz];
            uint64_t x1 = x1_first, y1 = y1_first;
            uint64_t x2_first, y2_first;
            label16(s2, probe, x2_first, y2_first);
            bool aside = false, stop = false;
            if (!(x1 & 0xFFull)) aside = true;                                // previous label exhausted: itree.c:1052
            else {
                // itree.c:1060-1061: td = first index > dv where s1 ends, differs from s2, or is ';' -- sixteen bytes per
                // step (two loads in flight; the blob is zero padded by 64 bytes, dev_image.c)
                uint64_t x2 = x2_first, y2 = y2_first;
                uint32_t base = probe, idx, prevlast = 0;
                bool first = true;
                for (;;) {
                    uint64_t m = stop_flags(x1, x2);
                    if (first && skip0) m &= ~0xFFull;
                    if (m) { idx = (uint32_t)(__builtin_ctzll(m) >> 3); break; }
                    prevlast = (uint32_t)(x1 >> 56);
                    base += 8; first = false;
                    m = stop_flags(y1, y2);
                    x1 = y1; x2 = y2;
                    if (m) { idx = (uint32_t)(__builtin_ctzll(m) >> 3); break; }
                    prevlast = (uint32_t)(x1 >> 56);
                    base += 8;
                    label16(s1, base, x1, y1); label16(s2, base, x2, y2);
                }
                td = base + idx;
                const uint32_t a = (uint32_t)(x1 >> (8 * idx)) & 0xFFu, b = (uint32_t)(x2 >> (8 * idx)) & 0xFFu;
                // s1[td-1]; reading before the string (td == 0) counts as "not '_'"
                const uint32_t before = idx ? ((uint32_t)(x1 >> (8 * idx - 8)) & 0xFFu) : (first ? 0u : prevlast);
                if (a == b) run += nz;                                                         // same token: itree.c:1062
                else if ((!a && b == ';') || ((a == ';' || !a) && before == '_')) aside = true;   // less specific: 1063
                else if (run >= cutoff) { ed = z; stop = true; }                               // group wins: itree.c:1068
                else { run = nz; st = z; }                                                     // restart: itree.c:1069
            }
            if (aside) {                                                       // 1053-1056 / 1064-1067
                run = nz; st = z;
                orun -= c1;
                cutoff = cut_of(orun);
            }
            if (stop) break;
            s1 = s2; c1 = nz; x1_first = x2_first; y1_first = y2_first;
        }
        sl = run; ol = orun;                                                   // itree.c:1071
        if (run < cutoff) break;                                               // itree.c:1072
        if (st + 1 >= ed) {                                                    // itree.c:1073-1079
            if ((uint32_t)(T[ed - 1] >> 32) >= cutoff) dv = 0xFFFFFFFEu;
            break;
        }
        orun = run; dv = td; cutoff = cut_of(run);                             // itree.c:1082-1085
    }
    const uint32_t rk = (uint32_t)T[ed - 1];
    int32_t cut;
    if (dv == 0xFFFFFFFFu) cut = -1;                                           // itree.c:1087
    else if (dv == 0xFFFFFFFEu) cut = -2;
    else { uint32_t Ls = loff[rk + 1] - loff[rk] - 1; cut = (int32_t)(dv < Ls ? dv : Ls); }   // 1088
    store_result(&out[r], im.rank2ix[rk], cut, F, uix, sl, ol);
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
// model_k: measurement aid for bench.py's byte model (not on the search path).  One wavefront per read (longer reads in pieces
// of MODEL_CAP windows); per read it counts the valid windows, the DISTINCT 64-byte buckets they address (what the
// classify kernels must fetch at least once per read), the distinct 128-byte HBM lines those buckets lie in, and the
// distinct buckets that end in an overflow descriptor.  Windows and buckets are evaluated the slow, direct way
// (minimizer<W>() per window) -- deliberately independent of the sliding-minimum code of the search kernels.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t MODEL_CAP = 640;
template <int W, int I>
__global__ __launch_bounds__(256) void model_k(utk_image im, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ off,
                                               const uint32_t *__restrict__ len, uint32_t n_reads, int do_rc,
                                               unsigned long long *__restrict__ counts) {
    constexpr uint32_t K = 4 * W;
    __shared__ uint64_t s_b[4][MODEL_CAP];
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    unsigned long long c_reads = 0, c_win = 0, c_buck = 0, c_line = 0, c_over = 0;
    for (uint32_t r = blockIdx.x * 4 + wv; r < n_reads; r += gridDim.x * 4) {
        const uint32_t L = len[r];
        const uint64_t o = off[r];
        const uint64_t total = do_rc ? 2ull * L + 1 : L;
        ++c_reads;
        if (total < K) continue;
        const uint64_t nwin_all = total - K + 1;
        // longer reads in pieces of MODEL_CAP windows: buckets are counted as distinct within a piece (a bucket that two pieces
        // share counts twice: at most one per piece, < 1 %)
        for (uint64_t w0 = 0; w0 < nwin_all; w0 += MODEL_CAP) {
            const uint32_t nwin = (uint32_t)(nwin_all - w0 < MODEL_CAP ? nwin_all - w0 : MODEL_CAP);
            for (uint32_t i = lane; i < nwin; i += 64) {
                uint64_t khi = 0, klo = 0;
                bool ok = true;
                for (uint64_t j = w0 + i; j < w0 + i + K; ++j) {
                    uint32_t code = 0; bool bad = true;
                    if (j < L) base_code(bases[o + j], code, bad);
                    else if (j > L) { base_code(bases[o + (2ull * L - j)], code, bad); code ^= 3u; }
                    ok = ok && !bad;
                    khi = (khi << 2) | (klo >> 62); klo = (klo << 2) | code;
                }
                uint64_t bucket = ~0ull;
                if (ok) { MinKey<W> mk; min_split<W>(W == 16 ? khi : 0ull, klo, im.regions, bucket, mk); }
                s_b[wv][i] = bucket;
            }
            wave_lds_fence();
            for (uint32_t i = lane; i < nwin; i += 64) {
                const uint64_t b = s_b[wv][i];
                if (b == ~0ull) continue;
                ++c_win;
                bool first_b = true, first_l = true;
                for (uint32_t j = 0; j < i; ++j) { const uint64_t x = s_b[wv][j]; first_b = first_b && x != b; }
                if (im.bucket_words == 16u) first_l = first_b;                           // (the bucket is the 128-byte line)
                else for (uint32_t j = 0; j < i; ++j) { const uint64_t x = s_b[wv][j]; first_l = first_l && (x == ~0ull || (x >> 1) != (b >> 1)); }
                c_buck += first_b; c_line += first_l;
                if (first_b) {
                    constexpr int EW = RecTraits<W, I>::EW, KW = RecTraits<W, I>::KW;
                    c_over += (im.table[b * im.bucket_words + (uint64_t)(im.bucket_words / EW - 1) * EW + KW] >> 62) == 2;
                }
            }
            wave_lds_fence();
        }
    }
    // lane 0 alone counted the reads; the other figures are summed over the lanes
    unsigned long long v[4] = {c_win, c_buck, c_line, c_over};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned long long x = v[q];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d);
        if (lane == 0 && x) atomicAdd(&counts[1 + q], x);
    }
    if (lane == 0 && c_reads) atomicAdd(&counts[0], c_reads);
}

}  // namespace

extern "C" {

int utk_classify_short(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len,
                       uint32_t n_reads, int do_rc, utree_result *d_out, const utk_workspace *ws, int n_cu, void *stream) {
    if (!n_reads) return 0;
    uint32_t blocks = (n_reads + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    uint32_t cap = (uint32_t)n_cu * 8u;
    if (blocks > cap) blocks = cap;
    return dispatch_img_all(im, [&](auto w, auto i, auto exc, auto offt) {
        if (ws->short_cap == UTREE_SHORT2_CAP)
            classify_short_k<decltype(w)::value, decltype(i)::value, decltype(exc)::value, decltype(offt), SHORT2_CAP, false>
                <<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(*im, d_bases, d_off, d_len, n_reads, do_rc, d_out, *ws);
        else if (do_rc)
            classify_short_k<decltype(w)::value, decltype(i)::value, decltype(exc)::value, decltype(offt), SHORT_CAP, false, 1>
                <<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(*im, d_bases, d_off, d_len, n_reads, do_rc, d_out, *ws);
        else
            classify_short_k<decltype(w)::value, decltype(i)::value, decltype(exc)::value, decltype(offt), SHORT_CAP, false, 0>
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
    uint32_t cap = (uint32_t)n_cu * 5u;
    if (blocks > cap) blocks = cap;
    return dispatch_img_all(im, [&](auto w, auto i, auto exc, auto offt) {
        classify_short_k<decltype(w)::value, decltype(i)::value, decltype(exc)::value, decltype(offt), MID_CAP, true>
            <<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(*im, d_bases, d_off, d_len, n_reads, do_rc, d_out, *ws);
    });
}

// what the lane-per-read pass (lanes_kernel.hip) listed, with the wave-per-read instantiation the batch's longest read needs
int utk_classify_listed(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len,
                        uint32_t n_reads, uint32_t max_len, int do_rc, utree_result *d_out, const utk_workspace *ws, int n_cu, void *stream) {
    if (!n_reads) return 0;
    // the list is short (a database that makes it long turns the lane-per-read pass off, dev_image.c): a quarter of the resident
    // grid finds that out sooner than a full one
    uint32_t blocks = (n_reads + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    uint32_t cap = (uint32_t)n_cu * 2u;
    if (blocks > cap) blocks = cap;
    const uint64_t total = do_rc ? 2ull * max_len + 1 : max_len;                    // staged bases of the longest read
    return dispatch_img(im, [&](auto w, auto i, auto exc, auto offt) {
        if (total > SHORT2_CAP)
            classify_short_k<decltype(w)::value, decltype(i)::value, decltype(exc)::value, decltype(offt), MID_CAP, true>
                <<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(*im, d_bases, d_off, d_len, n_reads, do_rc, d_out, *ws);
        else if (do_rc || total > SHORT_CAP)
            classify_short_k<decltype(w)::value, decltype(i)::value, decltype(exc)::value, decltype(offt), SHORT2_CAP, true>
                <<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(*im, d_bases, d_off, d_len, n_reads, do_rc, d_out, *ws);
        else
            classify_short_k<decltype(w)::value, decltype(i)::value, decltype(exc)::value, decltype(offt), SHORT_CAP, true, 0>
                <<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(*im, d_bases, d_off, d_len, n_reads, do_rc, d_out, *ws);
    });
}

int utk_classify_long(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len,
                      int do_rc, utree_result *d_out, const utk_workspace *ws, int n_cu, void *stream) {
    (void)n_cu;
    if (!ws->long_blocks) return 0;
    return dispatch_img_all(im, [&](auto w, auto i, auto exc, auto offt) {
        classify_long_k<decltype(w)::value, decltype(i)::value, decltype(exc)::value, decltype(offt)>
            <<<dim3(ws->long_blocks), dim3(LONG_THREADS), 0, (hipStream_t)stream>>>(*im, d_bases, d_off, d_len, do_rc, d_out, *ws);
    });
}

int utk_vote(const utk_image *im, utree_result *d_out, const utk_workspace *ws, uint32_t n_reads, void *stream) {
    if (!n_reads) return 0;
    if (im->vote_tab) vote_table_k<<<dim3((n_reads + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(*im, d_out, *ws, n_reads);
    else vote_k<<<dim3((n_reads + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(*im, d_out, *ws, n_reads);
    return (int)hipGetLastError();
}

int utk_lookup(const utk_image *im, const uint64_t *d_hi, const uint64_t *d_lo, uint64_t n, uint32_t *d_ix, void *stream) {
    if (!n) return 0;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    return dispatch_img_all(im, [&](auto w, auto i, auto exc, auto offt) {
        lookup_k<decltype(w)::value, decltype(i)::value, decltype(exc)::value, decltype(offt)>
            <<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(*im, d_hi, d_lo, n, d_ix);
    });
}

#ifdef UTREE_PHASE_TIMERS
void utk_phase_dump(void) {
    unsigned long long h[16];
    static const char *nm[12] = {"grab", "next read's len/off + fetch issue", "minimizers", "round: address (LDS chain) + load issue", "round: wait for the bucket",
                                 "round: scan", "round: hit compaction", "stage next read (waits for its bytes)", "tally + result", "after windows", "", ""};
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof h) != hipSuccess) return;
    unsigned long long tot = 0;
    for (int q = 0; q < 12; ++q) tot += h[q];
    fprintf(stderr, "[phase timers] %llu waves, %.3g cycles per wave\n", h[12], h[12] ? (double)tot / h[12] : 0.0);
    for (int q = 0; q < 10; ++q) fprintf(stderr, "  %-45s %5.1f %%\n", nm[q], tot ? 100.0 * h[q] / tot : 0.0);
    memset(h, 0, sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase), h, sizeof h);
}
#endif

int utk_model_counts(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len, uint32_t n_reads,
                     int do_rc, unsigned long long *d_counts, void *stream) {
    if (!n_reads) return 0;
    uint32_t blocks = (n_reads + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    return dispatch_wi(im->W, im->I, [&](auto w, auto i) {
        model_k<decltype(w)::value, decltype(i)::value><<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(*im, d_bases, d_off, d_len, n_reads,
                                                                                                      do_rc, d_counts);
    });
}

// The dominant kernel's name with its template arguments, as rocprofv3 prints them: bench.py matches it against the kernel
// the kept profile (profiles/traffic.json) was taken from.
static const char *kernel_signature(char *buf, size_t cap, const char *name, const utk_image *im, int slice, int listed, int rcmode) {
    const bool exc = (im->flags & (UTREE_F_IRREGULAR | UTREE_F_GENERIC)) != 0, o64 = (im->flags & UTREE_F_OFF64) != 0;
    if (slice) snprintf(buf, cap, "%s<%u, %u, %s, %s, %d, %s, %d>", name, im->W, im->I, exc ? "true" : "false", o64 ? "unsigned long" : "unsigned int",
                        slice, listed ? "true" : "false", rcmode);
    else snprintf(buf, cap, "%s<%u, %u, %s, %s>", name, im->W, im->I, exc ? "true" : "false", o64 ? "unsigned long" : "unsigned int");
    return buf;
}
const char *utk_classify_short_name(const utk_image *im, uint32_t short_cap, int mid, int do_rc, char *buf, size_t cap) {
    if (mid) return kernel_signature(buf, cap, "classify_short_k", im, MID_CAP, 1, 2);
    if (short_cap == UTREE_SHORT2_CAP) return kernel_signature(buf, cap, "classify_short_k", im, SHORT2_CAP, 0, 2);
    return kernel_signature(buf, cap, "classify_short_k", im, SHORT_CAP, 0, do_rc ? 1 : 0);
}
const char *utk_classify_long_name(const utk_image *im, char *buf, size_t cap) { return kernel_signature(buf, cap, "classify_long_k", im, 0, 0, 0); }

}  // extern "C"
