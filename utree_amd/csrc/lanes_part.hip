// lanes_part.hip -- the instantiations of the lane-per-read kernels (lanes_core.hpp) for ONE (W, I, NL, BS) = (k / 4, label-index bytes, 16-byte loads per
// bucket, both strands from one pass), compiled once per combination (Makefile: -DLANES_W= -DLANES_I= -DLANES_NL= -DLANES_BS=) so that the sets build
// side by side: a batch of reads of one lane (MODE 0), pieces of long reads (MODE 2), a batch of mixed lengths (classify_lanes_mixed_k: MODE 0 and the
// listed classes, MODE 1, in one launch).
#include "lanes_core.hpp"

#define CAT_(a, b, c, d, e) a##b##_##c##_##d##_##e
#define CAT(a, b, c, d, e) CAT_(a, b, c, d, e)
#if LANES_BS
#define BS_ true
#else
#define BS_ false
#endif

extern "C" int CAT(utk_lanes_part_, LANES_W, LANES_I, LANES_NL, LANES_BS)(int segs, int irr, int mode, const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off,
                                                                          const uint32_t *d_len, uint32_t n_reads, int do_rc, utree_result *d_out, const utk_workspace *ws,
                                                                          int n_cu, void *stream, uint32_t cls) {
#define GO(S_, M_) return irr ? launch_lanes<LANES_W, LANES_I, S_, true, M_, LANES_NL, BS_>(im, d_bases, d_off, d_len, n_reads, do_rc, d_out, ws, n_cu, stream, cls) \
                              : launch_lanes<LANES_W, LANES_I, S_, false, M_, LANES_NL, BS_>(im, d_bases, d_off, d_len, n_reads, do_rc, d_out, ws, n_cu, stream, cls)
    if (BS_ && !do_rc) return (int)hipErrorInvalidValue;
    if (mode == 3)                                             // a batch of mixed lengths, every lanes-per-read class in one launch (cls: the largest class)
        return irr ? launch_lanes_mixed<LANES_W, LANES_I, true, LANES_NL, BS_>(im, d_bases, d_off, d_len, n_reads, do_rc, d_out, ws, n_cu, stream, cls)
                   : launch_lanes_mixed<LANES_W, LANES_I, false, LANES_NL, BS_>(im, d_bases, d_off, d_len, n_reads, do_rc, d_out, ws, n_cu, stream, cls);
    if (mode == 0 && segs == 1) GO(1, 0);                      // a batch of reads of up to 160 bases, whole
    if (mode == 2 && segs == 16) GO(16, 2);                    // pieces of long reads
#undef GO
    return (int)hipErrorInvalidValue;
}

// (phase timers, a debugging build: the counters of the k = 32 (-DUTREE_LANES_TIMERS_W=16: k = 64) / u16 labels / 64-byte buckets set)
#ifndef UTREE_LANES_TIMERS_W
#define UTREE_LANES_TIMERS_W 8
#endif
#ifndef UTREE_LANES_TIMERS_BS
#define UTREE_LANES_TIMERS_BS 0                                 // 1: the both-strands set's counters
#endif
#if defined(UTREE_LANES_TIMERS) && LANES_W == UTREE_LANES_TIMERS_W && LANES_I == 2 && LANES_NL == 1 && LANES_BS == UTREE_LANES_TIMERS_BS
extern "C" {
void utk_lanes_phase_dump(void) {
    unsigned long long h[8];
    static const char *nm[6] = {"grab", "phase 0: bytes -> codes", "phase A: minimizer runs", "phase B: overflow runs", "phase C: tally, results", "phase B: buckets"};
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lphase), sizeof h) != hipSuccess) return;
    unsigned long long tot = 0;
    for (int q = 0; q < 6; ++q) tot += h[q];
    fprintf(stderr, "[lanes phase timers] %llu waves, %.4g cycles per wave\n", h[7], h[7] ? (double)tot / h[7] : 0.0);
    for (int q = 0; q < 6; ++q) fprintf(stderr, "  %-28s %5.1f %%\n", nm[q], tot ? 100.0 * h[q] / tot : 0.0);
    memset(h, 0, sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lphase), h, sizeof h);
}
}
#endif
