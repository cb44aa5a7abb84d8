// text_kernels.hip -- the byte work on either side of the classify kernels, on the device: read framing (a2) and output
// formatting (a10) for the whole-file search (search.c, device text pipeline).  gfx950 / wave64 only.
//
// The reference frames reads with two fgets per read under `omp critical` (itree.c:866-890) and prints with fprintf under
// the stdio lock (itree.c:1032, 1040, 1096); here a chunk of the FASTA goes to HBM as it stands and comes back as the
// chunk's output text, so the host only moves bytes.
//
//   nl_count_k / nl_scan_k / nl_emit_k   positions of the chunk's newlines (16 bytes per lane, exact byte masks)
//   frame_k       one lane per read: name and sequence spans as XT_INITIATE_WS yields them (itree.c:879-890)
//   fmt_len_k     one lane per read: length of its output line (0 when the read has no hit: itree.c:1028)
//   fmt_write_k   one wavefront per 64 reads: each lane builds its read's numeric tail in LDS, then the 64 lines are
//                 written one after the other by all lanes together (coalesced stores)
//
// Only WELL-FORMED input is handled here.  Anything else -- a NUL byte, a line of LINELEN-1 bytes or more (fgets would
// split it), a header that does not start with '>', a sequence line that does, an odd number of lines -- raises a flag,
// and search.c re-runs the file through the host framing of fasta.c, which reproduces the reference's behaviour on
// malformed input case by case (error text, exit code, the reads classified before the error).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include "utree_internal.h"
#include "text_kernels.h"

namespace {

constexpr uint32_t TB = 256;                       // threads per block
constexpr uint32_t BYTES_PER_BLOCK = TB * 16;
constexpr uint32_t LINELEN = 16777216u;            // itree.c:836

// 0x80 in every byte of x that is zero (exact: no borrow between bytes)
__device__ __forceinline__ uint64_t zero_bytes(uint64_t x) {
    return ~(((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x | 0x7F7F7F7F7F7F7F7Full);
}
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// the lane's 16 bytes at offset 16*t (bytes at or beyond n read as 0xFF: neither newline nor NUL)
__device__ __forceinline__ void load16(const uint8_t *__restrict__ buf, uint64_t n, uint64_t t, uint64_t &a, uint64_t &b) {
    const uint64_t o = t * 16;
    a = b = ~0ull;
    if (o + 16 <= n) { const ulonglong2 v = *(const ulonglong2 *)(buf + o); a = v.x; b = v.y; }
    else if (o < n) {
        for (uint32_t i = 0; i < 16 && o + i < n; ++i) {
            const uint64_t c = buf[o + i];
            if (i < 8) a = (a & ~(0xFFull << (8 * i))) | (c << (8 * i));
            else b = (b & ~(0xFFull << (8 * (i - 8)))) | (c << (8 * (i - 8)));
        }
    }
}

__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t *s_part) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const uint32_t w = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0) s_part[w] = v;
    __syncthreads();
    uint32_t t = 0;
    for (uint32_t i = 0; i < TB / 64; ++i) t += s_part[i];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(TB) void nl_count_k(const uint8_t *__restrict__ buf, uint64_t n, uint32_t *__restrict__ counts,
                                                 utk_text_meta *__restrict__ meta) {
    __shared__ uint32_t s_part[TB / 64];
    const uint64_t t = (uint64_t)blockIdx.x * TB + threadIdx.x;
    uint64_t a, b;
    load16(buf, n, t, a, b);
    const uint64_t na = zero_bytes(a ^ 0x0A0A0A0A0A0A0A0Aull), nb = zero_bytes(b ^ 0x0A0A0A0A0A0A0A0Aull);
    if (zero_bytes(a) | zero_bytes(b)) atomicOr(&meta->flags, UTK_TEXT_NUL);              // strlen (itree.c:887) would stop there
    const uint32_t c = block_sum((uint32_t)__popcll(na) + (uint32_t)__popcll(nb), s_part);
    if (threadIdx.x == 0) counts[blockIdx.x] = c;
}

// exclusive prefix of the per-block counts, one workgroup (a chunk has at most a few ten thousand blocks)
__global__ __launch_bounds__(1024) void nl_scan_k(uint32_t *__restrict__ counts, uint32_t nb, utk_text_meta *__restrict__ meta) {
    __shared__ uint32_t s_tot[1024];
    const uint32_t per = (nb + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += counts[i];
    s_tot[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {                                            // Hillis-Steele, inclusive
        const uint32_t v = threadIdx.x >= d ? s_tot[threadIdx.x - d] : 0;
        __syncthreads();
        s_tot[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = s_tot[threadIdx.x] - sum;
    for (uint32_t i = lo; i < hi; ++i) { const uint32_t c = counts[i]; counts[i] = run; run += c; }
    if (threadIdx.x == 1023) meta->n_lines = s_tot[1023];
}

__global__ __launch_bounds__(TB) void nl_emit_k(const uint8_t *__restrict__ buf, uint64_t n, const uint32_t *__restrict__ prefix,
                                                uint32_t max_lines, uint32_t *__restrict__ nl) {
    __shared__ uint32_t s_part[TB / 64];
    const uint64_t t = (uint64_t)blockIdx.x * TB + threadIdx.x;
    uint64_t a, b;
    load16(buf, n, t, a, b);
    uint64_t na = zero_bytes(a ^ 0x0A0A0A0A0A0A0A0Aull), nb = zero_bytes(b ^ 0x0A0A0A0A0A0A0A0Aull);
    const uint32_t c = (uint32_t)__popcll(na) + (uint32_t)__popcll(nb);
    uint32_t inc = c;                                                                    // inclusive scan over the wave
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(inc, d); if (lane >= (uint32_t)d) inc += v; }
    if (lane == 63) s_part[w] = inc;
    __syncthreads();
    uint32_t at = prefix[blockIdx.x] + inc - c;
    for (uint32_t i = 0; i < w; ++i) at += s_part[i];
    const uint32_t o = (uint32_t)(t * 16);
    while (na) { const uint32_t i = (uint32_t)__builtin_ctzll(na) >> 3; if (at < max_lines) nl[at] = o + i; ++at; na &= na - 1; }
    while (nb) { const uint32_t i = (uint32_t)__builtin_ctzll(nb) >> 3; if (at < max_lines) nl[at] = o + 8 + i; ++at; nb &= nb - 1; }
}

// one lane per read: lines 2r (header) and 2r+1 (sequence).  itree.c:879-890
__global__ __launch_bounds__(TB) void frame_k(const uint8_t *__restrict__ buf, const uint32_t *__restrict__ nl, uint32_t max_reads,
                                              uint64_t *__restrict__ seq_off, uint32_t *__restrict__ seq_len,
                                              uint32_t *__restrict__ name_off, uint32_t *__restrict__ name_len,
                                              utk_text_meta *__restrict__ meta) {
    // the read count comes from the newline pass on the same stream (no host round trip in between); the grid covers the
    // most reads the chunk's bytes could hold
    uint32_t n_reads = meta->n_lines >> 1;
    n_reads = n_reads < max_reads ? n_reads : max_reads;
    if (blockIdx.x * TB >= n_reads) return;
    const uint32_t r = blockIdx.x * TB + threadIdx.x;
    uint32_t length = 0, bad = 0;
    if (r < n_reads) {
        const uint32_t s0 = r ? nl[2 * r - 1] + 1 : 0, e0 = nl[2 * r], s1 = e0 + 1, e1 = nl[2 * r + 1];
        if (buf[s0] != '>') bad |= UTK_TEXT_NO_HEADER;                                    // itree.c:880
        if (buf[s1] == '>') bad |= UTK_TEXT_SEQ_HEADER;                                   // itree.c:886
        if (e0 - s0 + 1 > LINELEN - 1 || e1 - s1 + 1 > LINELEN - 1) bad |= UTK_TEXT_LONG_LINE;   // fgets(…, LINELEN) splits such a line
        uint32_t e = s0 + 1;                                                              // name: up to the first space / newline (881)
        while (e < e0 && buf[e] != ' ') ++e;
        length = e1 - s1;                                                                 // the line minus its '\n' (889)
        if (length && buf[s1 + length - 1] == '\r') --length;                             // itree.c:890
        seq_off[r] = s1; seq_len[r] = length;
        name_off[r] = s0 + 1; name_len[r] = e - (s0 + 1);
    }
    // per wave: one atomic each for the longest read, the bases and the flags
    uint32_t mx = length;
    uint64_t tot = length;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t m2 = __shfl_xor(mx, o); mx = m2 > mx ? m2 : mx;
        tot += __shfl_xor(tot, o);
        bad |= __shfl_xor(bad, o);
    }
    if ((threadIdx.x & 63u) == 0) {
        if (mx) atomicMax(&meta->max_len, mx);
        if (tot) atomicAdd(&meta->total_bases, (unsigned long long)tot);
        if (bad) atomicOr(&meta->flags, bad);
    }
}

__device__ __forceinline__ uint32_t dec_digits(uint32_t v) {
    return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u :
           v < 10000000u ? 7u : v < 100000000u ? 8u : v < 1000000000u ? 9u : 10u;
}
// label bytes the line prints (itree.c:1087-1088, 1096): -1 none, -2 the whole label, else its first `cut` bytes
__device__ __forceinline__ uint32_t label_bytes(const utree_result &q, const uint32_t *__restrict__ label_off, uint32_t rank) {
    const uint32_t full = label_off[rank + 1] - label_off[rank] - 1;
    if (q.cut == -1) return 0;
    if (q.cut == -2) return full;
    return (uint32_t)q.cut < full ? (uint32_t)q.cut : full;
}
__device__ __forceinline__ uint32_t tail_bytes(const utree_result &q) {                   // "\t<found>\t" + "1\t*" | "<uix>\t<sl>;<ol>" + "\n"
    return 1 + dec_digits(q.found) + 1 + (q.uix == 1 ? 3u : dec_digits(q.uix) + 1 + dec_digits(q.sl) + 1 + dec_digits(q.ol)) + 1;
}

__global__ __launch_bounds__(TB) void fmt_len_k(const utree_result *__restrict__ res, const uint32_t *__restrict__ name_len,
                                                const uint32_t *__restrict__ ix2rank, const uint32_t *__restrict__ label_off,
                                                uint32_t n_labels, uint32_t n_reads, uint32_t *__restrict__ line_len,
                                                utk_text_meta *__restrict__ meta) {
    const uint32_t r = blockIdx.x * TB + threadIdx.x;
    uint32_t len = 0, bad = 0;
    if (r < n_reads) {
        const utree_result q = res[r];
        if (q.found) {                                                                    // itree.c:1028: no hit, no line
            if (q.label >= n_labels) bad = 1;
            else len = name_len[r] + 1 + label_bytes(q, label_off, ix2rank[q.label]) + tail_bytes(q);
        }
        line_len[r] = len;
    }
    const uint64_t gm = __ballot(len != 0), bm = __ballot(bad != 0);
    if ((threadIdx.x & 63u) == 0) {
        if (gm) atomicAdd(&meta->good_finds, (unsigned long long)__popcll(gm));          // itree.c:1029
        if (bm) atomicOr(&meta->flags, UTK_TEXT_BAD_LABEL);
    }
}

__device__ __forceinline__ uint8_t *put_dec(uint8_t *o, uint32_t v) {
    const uint32_t d = dec_digits(v);
    for (uint32_t i = d; i-- > 0;) { o[i] = (uint8_t)('0' + v % 10u); v /= 10u; }
    return o + d;
}

constexpr uint32_t TAIL_MAX = 48;                   // 1+10+1+10+1+10+1+10+1 = 45
// One wavefront per 64 consecutive reads.  Lane l prepares read l's numeric tail in LDS and keeps its spans; then the 64
// lines are written in turn, every lane storing one byte per pass (name, label, tail: contiguous, so coalesced).
__global__ __launch_bounds__(TB) void fmt_write_k(const uint8_t *__restrict__ buf, const utree_result *__restrict__ res,
                                                  const uint32_t *__restrict__ name_off, const uint32_t *__restrict__ name_len,
                                                  const uint32_t *__restrict__ line_len, const uint64_t *__restrict__ line_off,
                                                  const uint32_t *__restrict__ ix2rank, const uint32_t *__restrict__ label_off,
                                                  const char *__restrict__ label_blob, uint32_t n_reads, uint8_t *__restrict__ out) {
    __shared__ uint8_t s_tail[TB / 64][64][TAIL_MAX];
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t r = (blockIdx.x * (TB / 64) + wv) * 64 + lane;
    uint32_t my_len = 0, my_name_off = 0, my_name_len = 0, my_lab_off = 0, my_lab_len = 0, my_tail = 0;
    uint64_t my_out = 0;
    if (r < n_reads) {
        my_len = line_len[r];
        if (my_len) {
            const utree_result q = res[r];
            const uint32_t rank = ix2rank[q.label];
            my_out = line_off[r];
            my_name_off = name_off[r]; my_name_len = name_len[r];
            my_lab_off = label_off[rank]; my_lab_len = label_bytes(q, label_off, rank);
            uint8_t *t = s_tail[wv][lane], *o = t;
            *o++ = '\t';
            o = put_dec(o, q.found);
            *o++ = '\t';
            if (q.uix == 1) { *o++ = '1'; *o++ = '\t'; *o++ = '*'; }                      // itree.c:1032, 1040
            else { o = put_dec(o, q.uix); *o++ = '\t'; o = put_dec(o, q.sl); *o++ = ';'; o = put_dec(o, q.ol); }   // 1096
            *o++ = '\n';
            my_tail = (uint32_t)(o - t);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint64_t todo = __ballot(my_len != 0);
    while (todo) {
        const int i = __builtin_ctzll(todo);
        todo &= todo - 1;
        const uint32_t nlen = (uint32_t)__builtin_amdgcn_readlane((int)my_name_len, i), noff = (uint32_t)__builtin_amdgcn_readlane((int)my_name_off, i);
        const uint32_t llen = (uint32_t)__builtin_amdgcn_readlane((int)my_lab_len, i), loff = (uint32_t)__builtin_amdgcn_readlane((int)my_lab_off, i);
        const uint32_t tlen = (uint32_t)__builtin_amdgcn_readlane((int)my_tail, i);
        const uint64_t o = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_out >> 32), i) << 32) |
                           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_out, i);
        uint8_t *dst = out + o;
        for (uint32_t j = lane; j < nlen; j += 64) dst[j] = buf[noff + j];
        if (lane == 0) dst[nlen] = '\t';
        dst += nlen + 1;
        for (uint32_t j = lane; j < llen; j += 64) dst[j] = (uint8_t)label_blob[loff + j];
        dst += llen;
        if (lane < tlen) dst[lane] = s_tail[wv][i][lane];
    }
}

struct widen { __device__ uint64_t operator()(uint32_t v) const { return v; } };

__global__ void fmt_total_k(const uint32_t *__restrict__ line_len, const uint64_t *__restrict__ line_off, uint32_t n_reads,
                            utk_text_meta *__restrict__ meta) {
    if (threadIdx.x == 0 && blockIdx.x == 0) meta->out_bytes = n_reads ? line_off[n_reads - 1] + line_len[n_reads - 1] : 0;
}

}  // namespace

extern "C" {

size_t utk_text_scan_temp_bytes(uint32_t max_reads) {
    size_t tb = 0;
    auto in = rocprim::make_transform_iterator((const uint32_t *)nullptr, widen());
    if (rocprim::exclusive_scan(nullptr, tb, in, (uint64_t *)nullptr, (uint64_t)0, (size_t)max_reads, rocprim::plus<uint64_t>()) != hipSuccess) return 0;
    return tb + 256;
}

int utk_text_newlines(const uint8_t *d_buf, uint64_t n, uint32_t *d_counts, uint32_t *d_nl, uint32_t max_lines, utk_text_meta *d_meta,
                      void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(d_meta, 0, sizeof(utk_text_meta), st) != hipSuccess) return (int)hipGetLastError();
    if (!n) return 0;
    const uint32_t nb = (uint32_t)((n + BYTES_PER_BLOCK - 1) / BYTES_PER_BLOCK);
    nl_count_k<<<dim3(nb), dim3(TB), 0, st>>>(d_buf, n, d_counts, d_meta);
    nl_scan_k<<<dim3(1), dim3(1024), 0, st>>>(d_counts, nb, d_meta);
    nl_emit_k<<<dim3(nb), dim3(TB), 0, st>>>(d_buf, n, d_counts, max_lines, d_nl);
    return (int)hipGetLastError();
}

int utk_text_frame(const uint8_t *d_buf, const uint32_t *d_nl, uint32_t max_reads, uint32_t grid_reads, uint64_t *d_seq_off,
                   uint32_t *d_seq_len, uint32_t *d_name_off, uint32_t *d_name_len, utk_text_meta *d_meta, void *stream) {
    if (grid_reads > max_reads) grid_reads = max_reads;
    if (!grid_reads) return 0;
    frame_k<<<dim3((grid_reads + TB - 1) / TB), dim3(TB), 0, (hipStream_t)stream>>>(d_buf, d_nl, max_reads, d_seq_off, d_seq_len, d_name_off,
                                                                                     d_name_len, d_meta);
    return (int)hipGetLastError();
}

int utk_text_format(const utk_image *im, const uint32_t *d_ix2rank, const uint8_t *d_buf, const utree_result *d_res,
                    const uint32_t *d_name_off, const uint32_t *d_name_len, uint32_t n_reads, uint32_t *d_line_len,
                    uint64_t *d_line_off, void *d_scan_tmp, size_t scan_tmp_bytes, uint8_t *d_out, uint64_t out_cap,
                    utk_text_meta *d_meta, int phase, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!n_reads) return 0;
    if (phase == 0) {                                  // lengths, offsets, total (the host checks the total against out_cap)
        fmt_len_k<<<dim3((n_reads + TB - 1) / TB), dim3(TB), 0, st>>>(d_res, d_name_len, d_ix2rank, im->label_off, im->n_labels, n_reads,
                                                                       d_line_len, d_meta);
        auto in = rocprim::make_transform_iterator((const uint32_t *)d_line_len, widen());
        size_t tb = scan_tmp_bytes;
        hipError_t e = rocprim::exclusive_scan(d_scan_tmp, tb, in, d_line_off, (uint64_t)0, (size_t)n_reads, rocprim::plus<uint64_t>(), st);
        if (e != hipSuccess) return (int)e;
        fmt_total_k<<<dim3(1), dim3(64), 0, st>>>(d_line_len, d_line_off, n_reads, d_meta);
    } else {
        (void)out_cap;
        fmt_write_k<<<dim3((n_reads + TB - 1) / TB), dim3(TB), 0, st>>>(d_buf, d_res, d_name_off, d_name_len, d_line_len, d_line_off,
                                                                         d_ix2rank, im->label_off, im->label_blob, n_reads, d_out);
    }
    return (int)hipGetLastError();
}

}  // extern "C"
