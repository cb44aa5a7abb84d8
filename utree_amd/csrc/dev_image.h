/* dev_image.h -- private definition of utree_dev. */
#ifndef UTREE_DEV_IMAGE_H
#define UTREE_DEV_IMAGE_H
#include "utree_internal.h"

#define UTREE_MAX_PENDING 256
#define UTREE_LANES_WINDOW 16

struct utree_dev {
    int device, n_cu, owns;
    void *image;
    size_t image_bytes;
    utree_image_header hdr;
    utk_image kimg;
    /* HIP-event timing of the dominant kernel (enabled by the first utree_classify_kernel_time call) */
    int timing_on, n_claimed, n_events, last_long;   /* last_long: the last batch's dominant kernel was classify_long_k */
    int last_mid, last_rc, last_lanes, last_pieces, last_mixed; uint32_t last_short_cap;  /* ... and which wave-per-read instantiation it was otherwise        */
    char kernel_sig[160];
    void *events[2 * UTREE_MAX_PENDING];
    unsigned char recorded[UTREE_MAX_PENDING];  /* pair i holds a complete bracket */
    double ms_total;
    uint64_t launches;
    /* rank-specific search: the reference's never-cleared hit array as later reads see it (rank.c) */
    void *rank_state;
    uint64_t rank_state_cap;
    /* lane-per-read pass: reads it left to the wave-per-read kernel, read back without a wait (a ring of pinned words; ~0 =
     * not arrived or consumed), summed here; a database whose reads mostly exceed what that pass keeps (hit-dense) turns it off */
    volatile unsigned long long *lanes_ring;     /* [64][2]: {reads left over, error word} of a batch */
    uint32_t lanes_ring_reads[64];               /* reads of the batch whose words slot i awaits (0: not a lane-pass batch) */
    unsigned char ring_inflight[64];             /* 0 free, 1 a copy was posted into the slot */
    unsigned lanes_ring_next;
    unsigned long long win_left[UTREE_LANES_WINDOW], win_reads[UTREE_LANES_WINDOW];   /* the last lane-pass batches */
    unsigned win_next, lanes_skipped;
    unsigned long long dev_error;                /* first error word that came back and was not polled yet */
    char ring_busy;                              /* spin lock of everything above */
    /* whole-file search: per-lane pinned / device buffers, kept between searches (search_dev.c) */
    void *search_ctx;
};

void utree_dev_set_hip_error(int err, const char *what);
const char *utree_last_hip_error(void);
int utree_pick_fine_bits(const utree_ctr *ctr, int fine_bits);
void utree_search_ctx_free(void *ctx);

#endif
