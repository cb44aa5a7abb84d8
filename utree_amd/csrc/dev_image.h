/* dev_image.h -- private definition of utree_dev. */
#ifndef UTREE_DEV_IMAGE_H
#define UTREE_DEV_IMAGE_H
#include "utree_internal.h"

#define UTREE_MAX_PENDING 256

struct utree_dev {
    int device, n_cu, owns;
    void *image;
    size_t image_bytes;
    utree_image_header hdr;
    utk_image kimg;
    /* HIP-event timing of the dominant kernel (enabled by the first utree_classify_kernel_time call) */
    int timing_on, n_claimed, n_events, last_long;   /* last_long: the last batch's dominant kernel was classify_long_k */
    int last_mid, last_rc, last_lanes, last_pieces; uint32_t last_short_cap;  /* ... and which wave-per-read instantiation it was otherwise        */
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
    volatile unsigned long long *lanes_ring;
    uint32_t lanes_ring_reads[64];
    unsigned lanes_ring_next;
    unsigned long long lanes_reads, lanes_left;
    int lanes_off;
    /* whole-file search: per-lane pinned / device buffers, kept between searches (search_dev.c) */
    void *search_ctx;
};

void utree_dev_set_hip_error(int err, const char *what);
const char *utree_last_hip_error(void);
int utree_pick_fine_bits(const utree_ctr *ctr, int fine_bits);
void utree_search_ctx_free(void *ctx);

#endif
