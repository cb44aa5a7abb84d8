// rank_kernels.hip -- the rank-specific search (`xtree-search`: itree.c -D SEARCH, doCollapse = 0 branch,
// itree.c:969-1007) on the device image.  gfx950 / wave64 only.  SURVEY.md §8(f) rank 1.
//
// Same framing, windows and node lookups as the SEARCH_GG path; what differs (all of it reproduced bit for bit):
//
//  * hit selection (XT_SHALLOWVOTE, itree.c:948-951): after a hit at a window the next S-1 windows are not examined
//    (S = PACKSIZE / SPARSITY), and the reference's word register is left shifted 2S-1 bases with only S new ones,
//    so until PACKSIZE bases have gone by the words it looks up are
//         (word of the hit << 2(d+S-1)) | (the d bases after the hit)          S <= d < PACKSIZE
//    rather than the query's k-mers; a hit on such a word restarts this from that word.
//  * the vote (itree.c:980-1003) counts the read's hits PLUS one entry that an earlier read left in the never-cleared
//    hit array (`if (!kingsMen++)`, 982): entry n of the latest earlier read with more than n hits (n = this read's
//    hit count), or 0.  Reads are therefore not independent.  The kernels keep every read's hit list, answer the
//    "latest earlier read with more hits" query with a 64-ary max tree over the hit counts, and carry the array from
//    batch to batch in `state`.
//  * output: label with the most votes, 1 - secondMost/most, most; only if most >= TOLERANCE_THRESHOLD and
//    most >= SLACK * secondMost (1000-1002).
//
//   rank_hits_k    one wavefront per read, any length (staged through LDS in 960-window segments)
//   block_max_k    max tree over the hit counts (three launches)
//   rank_vote_k    one wavefront per read: carried entry, tally, most / secondMost
//   rank_state_k   folds the batch into the carried array
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "device_common.hpp"
#include "wave_common.hpp"

using namespace utk;

namespace {

constexpr int RK_CAP = 1024;                       // staged bases per segment
constexpr uint32_t RK_SEGW = RK_CAP - 64;          // windows per segment (their last window needs K-1 <= 63 more bases)
constexpr int RK_WAVES = 4;
constexpr uint32_t RK_GRAB_HITS = 16, RK_GRAB_VOTE = 64;
constexpr uint32_t RK_CHUNK = UTREE_TALLY_CHUNK, RK_DIRECT = UTREE_TALLY_CHUNK / 16;
constexpr uint32_t NONE = 0xFFFFFFFFu;

__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t l) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint64_t t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}

// The word the reference looks up d bases after a hit on w0 (S <= d < K), given the query's real k-mer there.
template <int W>
__device__ __forceinline__ void register_word(uint64_t w0hi, uint64_t w0lo, uint64_t th, uint64_t tl, uint32_t d, uint32_t S,
                                              uint64_t &eh, uint64_t &el) {
    const uint32_t sh = 2u * (d + S - 1u);
    if constexpr (W == 8) {
        eh = 0;
        el = (sh >= 64u ? 0ull : (w0lo << sh)) | (tl & ((1ull << (2u * d)) - 1ull));
    } else {
        typedef unsigned __int128 u128;
        const u128 w0 = ((u128)w0hi << 64) | w0lo, t = ((u128)th << 64) | tl;
        const u128 e = (sh >= 128u ? (u128)0 : (w0 << sh)) | (t & ((((u128)1) << (2u * d)) - 1));
        eh = (uint64_t)(e >> 64); el = (uint64_t)e;
    }
}

// ------------------------------------------------------------------------------------------------
// rank_hits_k: the hits the reference keeps for each read, in order, as file-order label indices
// ------------------------------------------------------------------------------------------------
template <int W, int I, bool EXC, typename OFF>
__global__ __launch_bounds__(256) void rank_hits_k(utk_image im, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ off,
                                                   const uint32_t *__restrict__ len, uint32_t n_reads, int do_rc, utk_rank_ws ws) {
    constexpr uint32_t K = 4 * W;
    constexpr int NCH = RK_CAP / 64, NWORDS = RK_CAP / 16 + 6;
    __shared__ uint32_t s_words[RK_WAVES][NWORDS];
    __shared__ uint64_t s_bad[RK_WAVES][NCH + 2];
    const uint32_t lane = lane_id();
    const uint32_t wv = uni32(threadIdx.x >> 6);
    uint32_t *sw = s_words[wv];
    uint8_t *sb = (uint8_t *)sw;
    uint64_t *sbad = s_bad[wv];
    const uint32_t S = ws.step;
    unsigned long long chunk_base = 0;
    uint32_t chunk_left = 0;
    uint32_t item = 0, item_end = 0;
    for (;;) {
        if (item == item_end) {                                         // dynamic work distribution, as in classify_short_k
            unsigned long long g = 0;
            if (lane == 0) g = atomicAdd(&ws.cursors[1], (unsigned long long)RK_GRAB_HITS);
            item = uni32((uint32_t)g);
            if (item >= n_reads) break;
            item_end = item + RK_GRAB_HITS < n_reads ? item + RK_GRAB_HITS : n_reads;
        }
        const uint32_t r = item++;
        const uint32_t L = uni32(len[r]);
        const uint64_t o = uni64(off[r]);
        const uint32_t total = do_rc ? 2u * L + 1u : L;                  // host side bounds L (itree.c:836: 16 MiB lines)
        if (total < K) { if (lane == 0) { ws.nh[r] = 0; ws.hoff[r] = 0; } continue; }
        const uint32_t nwin = total - K + 1;
        const uint32_t cap_r = (nwin + S - 1) / S;                       // kept hits are >= S windows apart
        unsigned long long base;
        const bool direct = cap_r >= RK_DIRECT;
        if (direct) {
            unsigned long long nb = 0;
            if (lane == 0) nb = atomicAdd(&ws.cursors[0], (unsigned long long)cap_r);
            base = uni64(nb);
        } else {
            if (cap_r > chunk_left) {
                unsigned long long nb = 0;
                if (lane == 0) nb = atomicAdd(&ws.cursors[0], (unsigned long long)RK_CHUNK);
                chunk_base = uni64(nb);
                chunk_left = RK_CHUNK;
            }
            base = chunk_base;
            chunk_base += cap_r; chunk_left -= cap_r;
        }
        uint32_t n = 0, h0 = 0;
        bool have = false;                                               // a hit within the last K windows: (h0, w0)
        uint64_t w0hi = 0, w0lo = 0;
        for (uint32_t seg0 = 0; seg0 < nwin; seg0 += RK_SEGW) {
            const uint32_t nw = nwin - seg0 < RK_SEGW ? nwin - seg0 : RK_SEGW;
            const uint32_t nb = nw + K - 1, nch = (nb + 63) >> 6;
            // ---- stage bases [seg0, seg0+nb) of fwd + 'N' + revcomp (itree.c:891-898) as 2-bit codes + bad-base ballots
            for (uint32_t c = 0; c < nch; ++c) {
                const uint32_t jl = c * 64 + lane, j = seg0 + jl;
                uint32_t raw = 0;                                        // 0 is a bad base: padding and the 'N' separator
                if (jl < nb) {
                    if (j < L) raw = bases[o + j];
                    else if (j > L) raw = 0x100u | bases[o + (2u * L - j)];
                }
                uint32_t code; bool bad;
                base_code(raw & 0xFFu, code, bad);
                code ^= (raw >> 8) * 3u;                                 // complement
                const uint64_t bm = __ballot(bad);
                const uint32_t t = (code << 2) | (uint32_t)__shfl_down((int)code, 1);
                const uint32_t u = (t << 4) | (uint32_t)__shfl_down((int)t, 2);
                if ((lane & 3u) == 0) sb[(c * 16 + (lane >> 2)) ^ 3u] = (uint8_t)u;
                if (lane == 0) sbad[c] = bm;
            }
            if (lane == 0) { sbad[nch] = ~0ull; sbad[nch + 1] = ~0ull; }
            wave_lds_fence();
            // ---- 64 windows at a time, in order
            for (uint32_t rb = 0; rb < nw; rb += 64) {
                const uint32_t il = rb + lane, i = seg0 + il;            // window start: in the segment / in the read
                bool valid = false;
                if (il < nw) {
                    const uint32_t ch = il >> 6, bit = il & 63u;
                    const uint64_t b0 = sbad[ch], b1 = sbad[ch + 1];
                    const uint64_t x = (b0 >> bit) | (bit ? (b1 << (64 - bit)) : 0ull);
                    valid = (K == 64) ? (x == 0) : ((uint32_t)x == 0);
                }
                uint64_t th = 0, tl = 0;
                if (valid) window_word<W>(sw, il, th, tl);
                uint64_t ch_ = 0, cl_ = 0;                               // the word this lane last looked up, and its answer
                uint32_t rank = INVALID;
                bool cached = false;
                uint32_t lo_lane = 0;
                for (;;) {
                    bool elig = valid && lane >= lo_lane;
                    uint64_t eh = th, el = tl;
                    if (elig && have) {
                        const uint32_t d = i - h0;
                        if (d < S) elig = false;                         // never examined (itree.c:950)
                        else if (d < K) register_word<W>(w0hi, w0lo, th, tl, d, S, eh, el);
                    }
                    if (elig && !(cached && eh == ch_ && el == cl_)) {
                        rank = lookup_word<W, I, EXC, OFF>(im, eh, el);
                        ch_ = eh; cl_ = el; cached = true;
                    }
                    const uint64_t hm = __ballot(elig && rank != INVALID);   // itree.c:929
                    if (!hm) break;
                    const uint32_t p = (uint32_t)__builtin_ctzll(hm);   // lanes before p saw the right register: first hit is final
                    if (lane == p) ws.hits[base + n] = im.rank2ix[rank]; // itree.c:951
                    ++n;
                    have = true; h0 = seg0 + rb + p;
                    w0hi = readlane64(eh, p); w0lo = readlane64(el, p);
                    lo_lane = p + 1;
                    if (lo_lane >= 64) break;
                }
            }
            wave_lds_fence();
        }
        if (lane == 0) { ws.nh[r] = n; ws.hoff[r] = base; }
    }
}

// out[w] = max(in[64w .. 64w+63])
__global__ __launch_bounds__(256) void block_max_k(const uint32_t *__restrict__ in, uint32_t n, uint32_t *__restrict__ out) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v = idx < n ? in[idx] : 0u;
    v = wave_max_u32(v);
    if ((threadIdx.x & 63u) == 0 && idx < n) out[idx >> 6] = v;
}

// highest index in [base, base+64) below `limit` whose value exceeds n
__device__ __forceinline__ uint32_t pick_last_greater(const uint32_t *a, uint32_t base, uint32_t limit, uint32_t n, uint32_t lane) {
    const uint32_t idx = base + lane;
    const uint32_t v = idx < limit ? a[idx] : 0u;
    const uint64_t m = __ballot(v > n);
    return m ? base + 63u - (uint32_t)__builtin_clzll(m) : NONE;
}

// the latest read q < r with more than n hits (NONE if there is none in this batch)
__device__ __forceinline__ uint32_t prev_greater(const utk_rank_ws &ws, uint32_t r, uint32_t n, uint32_t lane) {
    const uint32_t b0 = r >> 6;
    uint32_t q = pick_last_greater(ws.nh, b0 << 6, r, n, lane);
    if (q != NONE) return q;
    const uint32_t g1 = b0 >> 6;
    uint32_t blk = pick_last_greater(ws.lvl[0], g1 << 6, b0, n, lane);
    if (blk == NONE) {
        const uint32_t s2 = g1 >> 6;
        uint32_t grp = pick_last_greater(ws.lvl[1], s2 << 6, g1, n, lane);
        if (grp == NONE) {
            uint32_t sup = NONE;
            for (int64_t b = (int64_t)((s2 >> 6) << 6); b >= 0 && sup == NONE; b -= 64)
                sup = pick_last_greater(ws.lvl[2], (uint32_t)b, s2, n, lane);
            if (sup == NONE) return NONE;
            grp = pick_last_greater(ws.lvl[1], sup << 6, (sup << 6) + 64u, n, lane);   // earlier groups are complete
        }
        blk = pick_last_greater(ws.lvl[0], grp << 6, (grp << 6) + 64u, n, lane);
    }
    return pick_last_greater(ws.nh, blk << 6, (blk << 6) + 64u, n, lane);
}

// ------------------------------------------------------------------------------------------------
// rank_vote_k: itree.c:980-1003.  LONG = false: reads with up to 63 hits, one entry per lane; LONG = true: the
// others, counted in a per-wavefront label histogram in HBM (one wavefront per workgroup).
// ------------------------------------------------------------------------------------------------
template <bool LONG>
__global__ __launch_bounds__(LONG ? 64 : 256) void rank_vote_k(utree_result *__restrict__ out, uint32_t n_reads, uint32_t n_labels,
                                                              utk_rank_ws ws) {
    const uint32_t lane = lane_id();
    const uint32_t wave_gid = uni32(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    constexpr uint32_t GRAB = LONG ? 1u : RK_GRAB_VOTE;
    uint32_t item = 0, item_end = 0;
    for (;;) {
        if (item == item_end) {
            unsigned long long g = 0;
            if (lane == 0) g = atomicAdd(&ws.cursors[LONG ? 3 : 2], (unsigned long long)GRAB);
            item = uni32((uint32_t)g);
            if (item >= n_reads) break;
            item_end = item + GRAB < n_reads ? item + GRAB : n_reads;
        }
        const uint32_t r = item++;
        const uint32_t n = uni32(ws.nh[r]);
        if (n == 0) {                                                   // no hit: no line (itree.c:980)
            if (!LONG && lane == 0) { uint32_t *o = (uint32_t *)&out[r]; o[0] = 0; o[1] = (uint32_t)-2; o[2] = 0; o[3] = 0; o[4] = 0; o[5] = 0; }
            continue;
        }
        if ((n + 1 > 64) != LONG) continue;
        // the entry past the read's own hits (itree.c:982-988)
        const uint32_t q = prev_greater(ws, r, n, lane);
        const uint32_t extra = uni32(q == NONE ? ws.state[n] : ws.hits[ws.hoff[q] + n]);
        const uint64_t base = uni64(ws.hoff[r]);
        uint32_t most = 0, second = 0, most_ix = 0;
        if constexpr (!LONG) {
            const bool mine = lane <= n;
            const uint32_t hv = lane < n ? ws.hits[base + lane] : extra;
            uint64_t left = __ballot(mine);
            while (left) {                                               // labels in order of first appearance (988-997)
                const uint32_t lead = (uint32_t)__builtin_ctzll(left);
                const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)hv, (int)lead);
                const uint64_t m = __ballot(mine && hv == v) & left;
                const uint32_t c = (uint32_t)__popcll(m);
                if (c > most) { second = most; most = c; most_ix = v; }
                else if (c > second) second = c;
                left &= ~m;
            }
        } else {
            uint32_t *hist = ws.hist + (size_t)wave_gid * n_labels;
            for (uint32_t j = lane; j <= n; j += 64) atomicAdd(&hist[j < n ? ws.hits[base + j] : extra], 1u);   // 984-985
            __threadfence();
            uint64_t best = 0;                                           // most votes, earliest first appearance
            for (uint32_t j = lane; j <= n; j += 64) {
                const uint32_t h = j < n ? ws.hits[base + j] : extra;
                const uint32_t c = atomicAdd(&hist[h], 0u);
                const uint64_t key = ((uint64_t)c << 32) | (0xFFFFFFFFu - j);
                best = key > best ? key : best;
            }
            best = wave_max_u64(best);
            most = (uint32_t)(best >> 32);
            const uint32_t jm = 0xFFFFFFFFu - (uint32_t)best;
            most_ix = uni32(jm < n ? ws.hits[base + jm] : extra);
            for (uint32_t j = lane; j <= n; j += 64) {
                const uint32_t h = j < n ? ws.hits[base + j] : extra;
                if (h != most_ix) { const uint32_t c = atomicAdd(&hist[h], 0u); second = c > second ? c : second; }
            }
            second = wave_max_u32(second);
            __threadfence();
            for (uint32_t j = lane; j <= n; j += 64) atomicExch(&hist[j < n ? ws.hits[base + j] : extra], 0u);   // 996
            __threadfence();
        }
        // itree.c:1000 (int arithmetic)
        const bool drop = (int)most < (int)ws.tolerance || (int)most < (int)(ws.slack * second);
        if (lane == 0) {
            uint32_t *o = (uint32_t *)&out[r];
            o[0] = most_ix; o[1] = (uint32_t)(drop ? -4 : -2); o[2] = n; o[3] = 0; o[4] = most; o[5] = second;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// rank_state_k: after the batch, state[j] = entry j of the LAST read with more than j hits, where there is one.
// One wavefront per 64 reads; read q owns the indices from the largest hit count after it up to its own.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rank_state_k(uint32_t n_reads, utk_rank_ws ws) {
    const uint32_t lane = lane_id();
    const uint32_t b = uni32(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if ((uint64_t)b * 64 >= n_reads) return;
    const uint32_t q = b * 64 + lane;
    const uint32_t v = q < n_reads ? ws.nh[q] : 0u;
    // largest count in later blocks
    uint32_t after = 0;
    {
        const uint32_t g1 = b >> 6, s2 = g1 >> 6;
        uint32_t idx = (g1 << 6) + lane;
        if (idx > b && idx < ws.nlvl[0]) after = ws.lvl[0][idx];
        idx = (s2 << 6) + lane;
        if (idx > g1 && idx < ws.nlvl[1]) { const uint32_t t = ws.lvl[1][idx]; after = t > after ? t : after; }
        for (uint32_t base = (s2 >> 6) << 6; base < ws.nlvl[2]; base += 64) {
            idx = base + lane;
            if (idx > s2 && idx < ws.nlvl[2]) { const uint32_t t = ws.lvl[2][idx]; after = t > after ? t : after; }
        }
        after = wave_max_u32(after);
    }
    // largest count after this read: later lanes of the block, then `after`
    uint32_t sfx = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_down((int)sfx, o);
        if (lane + o < 64 && t > sfx) sfx = t;
    }
    uint32_t later = (uint32_t)__shfl_down((int)sfx, 1);
    if (lane == 63) later = 0;
    later = later > after ? later : after;
    uint64_t todo = __ballot(v > later);
    const uint64_t myoff = q < n_reads ? ws.hoff[q] : 0ull;
    while (todo) {
        const uint32_t l = (uint32_t)__builtin_ctzll(todo);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)later, (int)l);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
        const uint64_t src = readlane64(myoff, l);
        for (uint32_t j = lo + lane; j < hi; j += 64) ws.state[j] = ws.hits[src + j];
        todo &= todo - 1;
    }
}

}  // namespace

extern "C" {

int utk_rank_hits(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len, uint32_t n_reads,
                  int do_rc, const utk_rank_ws *ws, int n_cu, void *stream) {
    if (!n_reads) return 0;
    uint32_t blocks = (n_reads + RK_WAVES - 1) / RK_WAVES;
    const uint32_t cap = (uint32_t)n_cu * 8u;                            // 32 wavefronts per CU: dev_image's hit-list bound assumes it
    if (blocks > cap) blocks = cap;
    return dispatch_img(im, [&](auto w, auto i, auto exc, auto offt) {
        rank_hits_k<decltype(w)::value, decltype(i)::value, decltype(exc)::value, decltype(offt)>
            <<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(*im, d_bases, d_off, d_len, n_reads, do_rc, *ws);
    });
}

int utk_rank_levels(const utk_rank_ws *ws, uint32_t n_reads, void *stream) {
    const uint32_t *in = ws->nh;
    uint32_t n = n_reads;
    for (int l = 0; l < 3; ++l) {
        if (!n) break;
        block_max_k<<<dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(in, n, ws->lvl[l]);
        in = ws->lvl[l]; n = ws->nlvl[l];
    }
    return (int)hipGetLastError();
}

int utk_rank_vote(const utk_image *im, utree_result *d_out, uint32_t n_reads, const utk_rank_ws *ws, int n_cu, void *stream) {
    if (!n_reads) return 0;
    uint32_t blocks = (n_reads + 4 * RK_GRAB_VOTE - 1) / (4 * RK_GRAB_VOTE);
    const uint32_t cap = (uint32_t)n_cu * 8u;
    if (blocks > cap) blocks = cap;
    rank_vote_k<false><<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(d_out, n_reads, im->n_labels, *ws);
    if (ws->hist_waves)
        rank_vote_k<true><<<dim3(ws->hist_waves), dim3(64), 0, (hipStream_t)stream>>>(d_out, n_reads, im->n_labels, *ws);
    return (int)hipGetLastError();
}

int utk_rank_state(uint32_t n_reads, const utk_rank_ws *ws, void *stream) {
    if (!n_reads) return 0;
    const uint32_t waves = (n_reads + 63) / 64;
    rank_state_k<<<dim3((waves + 3) / 4), dim3(256), 0, (hipStream_t)stream>>>(n_reads, *ws);
    return (int)hipGetLastError();
}

}  // extern "C"
