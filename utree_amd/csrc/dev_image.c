/* dev_image.c -- device image of a database and the batch entry points (host orchestration, C).
 *
 * The image replaces UTree.Dump / UTree.BinIx (itree.c:140-141) as XT_getIX32 (itree.c:720) sees them.
 * Layout in HBM (one flat allocation, offsets only -- DESIGN.md §3):
 *
 *   [header 4 KiB][table: 128-byte buckets over the minimizer hash][MIN records: nodes by (bucket, key): overflow runs]
 *   [bin table 2^24+1][irregular-bin bitmap 2 MiB][label offsets][labels in strcmp order][rank -> file index]
 *   [FILE records: nodes as the file orders them -- kept in the image only when some bin needs the exact probe path]
 */
#define _FILE_OFFSET_BITS 64
#define _GNU_SOURCE
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <fcntl.h>
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "ctr_host.h"
#include "dev_image.h"

/* (HBM exhausted is UTREE_E_NOMEM, not "no device": a database too large for what is free says so) */
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { utree_dev_set_hip_error((int)e_, #x); rc = e_ == hipErrorOutOfMemory ? UTREE_E_NOMEM : UTREE_E_HIP; goto fail; } } while (0)
#define KCHK(x) do { int e_ = (x); if (e_ != 0) { utree_dev_set_hip_error(e_, #x); rc = e_ == (int)hipErrorOutOfMemory ? UTREE_E_NOMEM : UTREE_E_HIP; goto fail; } } while (0)

static __thread char g_hip_msg[256];
void utree_dev_set_hip_error(int err, const char *what) {
    /* (the runtime's message first: `what` is the text of the failing call and may be longer than the buffer) */
    snprintf(g_hip_msg, sizeof g_hip_msg, "%s (%d) in %s", hipGetErrorString((hipError_t)err), err, what);
    if (getenv("UTREE_DEBUG")) fprintf(stderr, "[utree_amd] HIP error: %s\n", g_hip_msg);
}
const char *utree_last_hip_error(void) { return g_hip_msg; }

#include <time.h>
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static int timing_on(void) { return getenv("UTREE_TIMING") != NULL || getenv("UTREE_DEBUG") != NULL; }

static uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

/* A bucket's size, in 8-byte words: 8 (64 bytes) unless UTREE_BUCKET_BYTES=128 asks for line-sized buckets (16).  Same-box on config 2
 * (profiles/r03/ab_v8_vs_r03_128B_buckets.txt, DESIGN_APPENDIX.md section 0.2): 64-byte buckets make the classify kernels 5-10 % faster -- half the entries to scan
 * per lookup, one request per bucket instead of two --, 128-byte buckets make the image a third smaller (twice the nodes per bucket at the
 * same overflow rate) and every fetched byte one that is looked at. */
/* The default depends on the database: a 16-mer's hash is one of 2^32 values and the minimizer is a MINIMUM, so at the dense end of the hash
 * range a single value is shared by N (K - 15) / 2^32 nodes -- 4.8 for config 2 (1.217 G 32-mers), 6.5 for config 5 (568 M 64-mers) -- and there is
 * at most one bucket per value.  Once that number passes what a 64-byte bucket holds, most dense-end lookups continue in an overflow run and the
 * line-sized bucket is the faster one (same box, kernel ms per 4 M reads, 64 / 128 bytes: k = 32: 2.0 G nodes 1.76 / 1.78, 2.7 G 1.96 / 1.87, 3.5 G
 * 2.23 / 1.86; k = 64: 568 M 1.29 / 1.53, 1.2 G 1.68 / 1.67, 2.0 G 2.46 / 2.26; profiles/r03/scale_bucket_sizes.txt).  UTREE_BUCKET_BYTES=64|128
 * overrides. */
static uint32_t bucket_words_default(uint64_t n_nodes, uint32_t W) {
    const char *e = getenv("UTREE_BUCKET_BYTES");
    if (e && atoi(e) == 128) return 16u;
    if (e && atoi(e) == 64) return 8u;
    const char *t = getenv("UTREE_BUCKET128_NODES");                    /* test hook: the size from which the automatic choice is 128 bytes */
    const uint64_t from = t && atoll(t) > 0 ? (uint64_t)atoll(t) : (W == 16 ? 1250000000ull : 2200000000ull);
    return n_nodes > from ? 16u : 8u;
}

/* Buckets per hash region (utree_image_header.regions) for a tree of N nodes: a node's minimizer hash is the smallest of m = K-15
 * hashes, so region r -- the hashes whose top 8 bits are r -- expects N ((1 - r/256)^m - (1 - (r+1)/256)^m) of the nodes.  The region
 * gets as many buckets as bring a bucket to TARGET nodes: any number (bucket = base + ((h24 * nb) >> 24)), between 2^16 -- a bucket
 * spans at most 256 hash values: the record key has 8 bits for them -- and 2^(16+F), at most one bucket per hash value (F = fine_bits:
 * 8 = as many as the density asks, 0 = 256 values per bucket everywhere).  Default TARGET: 128-byte buckets are filled to 56 % (config
 * 2: 9 of 16 entries, image 17.6 GiB, 0.3 overflowing buckets per 150 bp read), 64-byte buckets to 37.5 % (3 of 8, 22.9 GiB, 0.28).  The
 * dense end of the hash range has several nodes per hash VALUE, so a bucket there holds the nodes of one, two or three values: a
 * mixture, not one Poisson mean -- which is why the small buckets want the lower load (DESIGN_APPENDIX.md sections 0.2 and 3 have the sweeps). */
/* share of a bucket's NODES that sit in a bucket of more than `cap` of them, the bucket's load being Poisson(lam): sum_{k > cap} k P(k) / lam */
static double pois_tail_nodes(double lam, uint32_t cap) {
    if (lam <= 0) return 0;
    double p = exp(-lam), s = 0;
    for (uint32_t k = 1; k <= cap; ++k) { p *= lam / k; s += k * p; }
    s = 1.0 - s / lam;
    return s < 0 ? 0 : s;
}
/* UTREE_CANON_MODE 1: only every other hash value is some canonical 16-mer's, so a pair of `v` = 2^24 / nb hash values holds Binomial(v, 1/2)
 * occupied ones, each with Poisson(lam1) nodes per bucket -- lumps: the share of the region's nodes in overflowing buckets */
static double lumpy_overflow(double nb, double expect, uint32_t cap) {
    const double v = 16777216.0 / nb, lam1 = expect / 16777216.0;
    if (v > 48.0) return pois_tail_nodes(expect / (2.0 * nb), cap);    /* dozens of values per pair: no lumps to speak of */
    const int lo = (int)floor(v);
    const double fr = v - lo;
    double tot = 0, totw = 0;
    for (int which = 0; which < 2; ++which) {
        const int V = lo + which;
        const double w = which ? fr : 1.0 - fr;
        if (!V || w <= 0) continue;
        double pn = pow(0.5, V);                                        /* Binomial(V, 1/2): n = 0 */
        for (int n = 0; n <= V && n <= 64; ++n) {
            const double mean = n * lam1;
            tot += w * pn * mean * pois_tail_nodes(mean, cap); totw += w * pn * mean;
            pn = pn * (double)(V - n) / (double)(n + 1);
        }
    }
    return totw > 0 ? tot / totw : 0;
}

/* k = 64 (UTREE_MIN_MARGIN): where a region has a slot per hash value and a value still holds more nodes than the design load, the slot is
 * `sub` pairs of buckets, picked by the four bases around the minimizer (256 combinations: a pair gets floor or ceil(256 / sub) of them).
 * UTREE_SUB_SLICES=0: none (A/B). */
static int sub_slices_on(uint32_t W, uint32_t F) {
    const char *e = getenv("UTREE_SUB_SLICES");
    return W == 16 && F >= 8 && !(e && e[0] == '0');
}
static uint64_t compute_regions_sub(uint64_t n_nodes, uint32_t W, uint32_t I, uint32_t bucket_words, uint32_t F, int sub_on, uint64_t regions[256]) {
    /* (image version 11: the region table counts PAIRS of buckets -- the two orientations of a canonical 16-mer --, a pair is sized for
     * 2 TARGET nodes) */
    const double m = (UTREE_CANON_MODE == 2 ? 2.0 : 1.0) * (4.0 * W - 15.0 - 2.0 * UTREE_MIN_MARGIN(W));   /* mode 2: the smallest of 2 (K - 15) hashes */
    const uint32_t cap_entries = bucket_words / utree_rec_words(W, I);
    const char *te = getenv("UTREE_BUCKET_TARGET");                     /* nodes per bucket; experiments only */
    const double target = te && atof(te) > 0 ? atof(te) : (bucket_words == 16 ? 0.5625 : 0.375) * cap_entries;
    const char *se = getenv("UTREE_LUMP_SLACK");                        /* experiments only */
    const double slack = se ? atof(se) : 1.5;
    const double p0 = pois_tail_nodes(target, cap_entries);
    const uint64_t nb_max = 1ull << (16 + (F > 8 ? 8 : F)), nb_min = 1ull << 16;
    uint64_t base = 0;
    for (int r = 0; r < 256; ++r) {
        const double expect = (double)n_nodes * (pow(1.0 - r / 256.0, m) - pow(1.0 - (r + 1) / 256.0, m));
        double want = ceil(expect / (2.0 * target));
        uint64_t nb = want >= (double)nb_max ? nb_max : (uint64_t)want, sub = 1;
        if (nb < nb_min) nb = nb_min;
        if (nb > nb_max) nb = nb_max;
        /* mode 1: more pairs where a pair holds so few hash values that the occupied ones make lumps: up to the overflow share the design
         * load has without them (times `slack`), at most one pair per value */
        if (UTREE_CANON_MODE != 2 && slack > 0)
            while (nb < nb_max && lumpy_overflow((double)nb, expect, cap_entries) > slack * p0 + 1e-4) { nb += nb / 20 + 1; if (nb > nb_max) nb = nb_max; }
        /* ... and beyond one slot per value (k = 64): an occupied value -- every other one in mode 1 -- has expect / 2^23 nodes, half of them per
         * orientation, spread over the slot's pairs by the 256 combinations of four bases; the fullest pair gets ceil(256 / sub) of them */
        if (sub_on && nb == (1ull << 24)) {
            const double per_value = expect / 16777216.0 * (UTREE_CANON_MODE == 2 ? 0.5 : 1.0);   /* nodes per occupied value and orientation */
            while (sub < 256 && pois_tail_nodes(per_value * ceil(256.0 / (double)sub) / 256.0, cap_entries) > slack * p0 + 1e-4) ++sub;
        }
        /* test hook: that many pairs per slot in every region (the addressing does not need a slot to be one hash value) */
        { const char *ts = getenv("UTREE_TEST_SUB"); if (sub_on && ts && atoi(ts) >= 1 && atoi(ts) <= 256) sub = (uint64_t)atoi(ts); }
        if (regions) regions[r] = (base << UTREE_REGION_BASE_SHIFT) | (sub << UTREE_REGION_NB_BITS) | nb;
        base += nb * sub;
    }
    return 2 * base;
}
static uint64_t compute_regions(uint64_t n_nodes, uint32_t W, uint32_t I, uint32_t bucket_words, uint32_t F, uint64_t regions[256]) {
    /* sub-slices as long as the table stays within the cap (utree_pick_fine_bits lowers F only after they are gone) */
    if (sub_slices_on(W, F)) {
        const char *cap_env = getenv("UTREE_TABLE_MAX_GB");
        const double cap = (cap_env && atof(cap_env) > 0 ? atof(cap_env) : 96.0) * 1073741824.0;
        if ((double)compute_regions_sub(n_nodes, W, I, bucket_words, F, 1, NULL) * 8.0 * bucket_words <= cap)
            return compute_regions_sub(n_nodes, W, I, bucket_words, F, 1, regions);
    }
    return compute_regions_sub(n_nodes, W, I, bucket_words, F, 0, regions);
}

/* MIN records beyond one per node the build area holds: the second views of k-mers whose two views differ (ties in the 23-bit rank,
 * palindromic minimizers: a few in a million on random sequence, a few per cent in repeats).  A database with more gets none and its
 * image does not carry UTREE_F_STRAND_VIEWS (the lane pass then walks the reverse strand as a second sequence). */
static uint64_t dup_capacity(uint64_t n_nodes) {
    const char *e = getenv("UTREE_DUP_CAP");                            /* test hook */
    if (e && atoll(e) >= 0) return (uint64_t)atoll(e);
    return n_nodes / 8 + 65536;
}

int utree_pick_fine_bits(const utree_ctr *ctr, int fine_bits) {
    const char *env = getenv("UTREE_FINE_BITS");
    if (fine_bits == UTREE_FINE_AUTO && env && *env) fine_bits = atoi(env);
    if (fine_bits == UTREE_FINE_AUTO) {
        /* as fine as the density asks, within a memory cap for the table */
        const char *cap_env = getenv("UTREE_TABLE_MAX_GB");
        double cap = (cap_env && atof(cap_env) > 0 ? atof(cap_env) : 96.0) * 1073741824.0;
        int F = 8;
        const uint32_t bw = bucket_words_default(ctr->info.n_nodes, ctr->info.W);
        while (F > 0 && (double)compute_regions(ctr->info.n_nodes, ctr->info.W, ctr->info.I, bw, (uint32_t)F, NULL) * 8.0 * bw > cap) --F;
        return F;
    }
    if (fine_bits < 0) fine_bits = 0;
    if (fine_bits > 8) fine_bits = 8;
    return fine_bits;
}

static void layout(const utree_ctr *ctr, uint32_t F, utree_image_header *h) {
    memset(h, 0, sizeof *h);
    h->magic = UTREE_IMG_MAGIC; h->version = UTREE_IMG_VERSION;   /* 6: k = 32 / u16-label records keep the rest in their low word (scan_bucket82); 7: the MIN area keeps the overflow runs only; 8: four-instruction minimizer hash; 9: 128-byte buckets, any number per region */
    h->W = ctr->info.W; h->I = ctr->info.I; h->k = ctr->info.k;
    h->fine_bits = F; h->rec_words = utree_rec_words(h->W, h->I);
    h->n_labels = ctr->info.n_labels; h->n_nodes = ctr->info.n_nodes;
    /* UTREE_FORCE_OFF64: test hook that runs the 64-bit-offset instantiations (N >= 2^32-1 databases) on small files */
    h->flags = (ctr->info.binix_width == 8 || getenv("UTREE_FORCE_OFF64")) ? UTREE_F_OFF64 : 0;
    h->bucket_words = bucket_words_default(h->n_nodes, h->W);
    uint64_t off = UTREE_IMG_HEADER_BYTES;
    if (h->W == 4) {
        /* PACKSIZE=16: a k-mer is a 32-bit word, so the image holds XT_getIX32's answer for every word: 2^32 ranks of I bytes (8 or 16 GiB whatever
         * the database's size), one load per window.  No buckets, no MIN records; the FILE records only while the table is built. */
        h->flags |= UTREE_F_DIRECT;
        h->bucket_words = 8; h->n_slots = 0; h->fine_bits = 0;
        h->off_table = off; off = align_up(off + ((uint64_t)1 << 32) * h->I, 4096);
        h->off_mrecs = off; off = align_up(off + 64, 4096);
    } else {
    h->n_slots = compute_regions(h->n_nodes, h->W, h->I, h->bucket_words, F, h->regions);
    h->off_table = off; off = align_up(off + h->n_slots * 8 * h->bucket_words, 4096);
    h->off_mrecs = off; off = align_up(off + (h->n_nodes + dup_capacity(h->n_nodes) + 8) * h->rec_words * 8, 4096);
    }
    h->off_coarse = off; off = align_up(off + (uint64_t)UTREE_NUMBINS * ((h->flags & UTREE_F_OFF64) ? 8 : 4), 256);
    h->off_irreg = off; off = align_up(off + (1u << 24) / 8, 256);
    h->off_label_off = off; off = align_up(off + ((uint64_t)h->n_labels + 1) * 4, 256);
    uint64_t blob = 0;
    for (uint32_t i = 0; i < h->n_labels; ++i) blob += (uint64_t)ctr->label_len[i] + 1;
    h->label_blob_bytes = blob;
    h->off_label_blob = off; off = align_up(off + blob + 64, 256);
    h->off_rank2ix = off; off = align_up(off + (uint64_t)h->n_labels * 4, 256);
    h->off_vote = off; off = align_up(off + (uint64_t)h->n_labels * sizeof(utk_vote_rec), 4096);
    /* last, so that an image without irregular bins can leave it out of what is broadcast */
    h->off_recs = off; off = align_up(off + (h->n_nodes + 8) * h->rec_words * 8, 4096);
    h->total_bytes = off;
}

size_t utree_dev_image_bytes(const utree_ctr *ctr, int fine_bits) {
    if (!ctr) return 0;
    utree_image_header h;
    layout(ctr, (uint32_t)utree_pick_fine_bits(ctr, fine_bits), &h);
    return (size_t)h.total_bytes;
}

static void bind_image(utree_dev *d) {
    char *b = (char *)d->image;
    d->kimg.table = (const uint64_t *)(b + d->hdr.off_table);
    d->kimg.regions = (const uint64_t *)(b + offsetof(utree_image_header, regions));
    d->kimg.mrecs = (const uint64_t *)(b + d->hdr.off_mrecs);
    d->kimg.recs = (const uint64_t *)(b + d->hdr.off_recs);
    d->kimg.coarse = b + d->hdr.off_coarse;
    d->kimg.irreg = (const uint32_t *)(b + d->hdr.off_irreg);
    d->kimg.label_off = (const uint32_t *)(b + d->hdr.off_label_off);
    d->kimg.label_blob = b + d->hdr.off_label_blob;
    d->kimg.rank2ix = (const uint32_t *)(b + d->hdr.off_rank2ix);
    d->kimg.vote_tab = (d->hdr.flags & UTREE_F_VOTE_TABLE) && !getenv("UTREE_VOTE_BYTES") ? (const uint64_t *)(b + d->hdr.off_vote) : NULL;
    d->kimg.n_nodes = d->hdr.n_nodes;
    d->kimg.n_labels = d->hdr.n_labels;
    d->kimg.fine_bits = d->hdr.fine_bits;
    d->kimg.flags = d->hdr.flags;
    d->kimg.W = d->hdr.W; d->kimg.I = d->hdr.I;
    d->kimg.bucket_words = d->hdr.bucket_words;
    {   /* n_min = records kept in overflow runs */
        const char *e = getenv("UTREE_OVF_SCAN");
        d->kimg.ovf_scan = e && atoi(e) > 0 && atoi(e) <= 1024 ? (uint32_t)atoi(e) : (d->hdr.n_min * 4 > d->hdr.n_nodes ? 16u : 32u);
    }
    d->kimg.irr_n = 0;
    for (int i = 0; i < 4; ++i) d->kimg.irr_p[i] = 0xFFFFFFFFu;
    if (d->hdr.flags & UTREE_F_GENERIC) d->kimg.irr_n = 0xFFFFFFFFu;
    else if (d->hdr.flags & UTREE_F_IRREGULAR) {
        /* list the irregular bins if they are few (one pass over the 2 MB bitmap, once per handle) */
        d->kimg.irr_n = 0xFFFFFFFFu;
        uint32_t *bm = d->hdr.n_irregular <= 4 ? (uint32_t *)malloc((1u << 24) / 8) : NULL;
        if (bm && hipMemcpy(bm, d->kimg.irreg, (1u << 24) / 8, hipMemcpyDeviceToHost) == hipSuccess) {
            uint32_t n = 0, p[4] = {0, 0, 0, 0};
            for (uint32_t w = 0; w < (1u << 24) / 32 && n <= 4; ++w)
                for (uint32_t x = bm[w]; x && n <= 4; x &= x - 1) { if (n < 4) p[n] = w * 32 + (uint32_t)__builtin_ctz(x); ++n; }
            if (n <= 4) { d->kimg.irr_n = n; for (uint32_t i = 0; i < n; ++i) d->kimg.irr_p[i] = p[i]; }
        } else (void)hipGetLastError();
        free(bm);
    }
}

static void lanes_ring_init(utree_dev *d);
static int device_ok(int device, int *n_cu) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return UTREE_E_HIP;
    if (hipSetDevice(device) != hipSuccess) return UTREE_E_HIP;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return UTREE_E_HIP;
    if (n_cu) *n_cu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    return UTREE_OK;
}

/* ---- build ---------------------------------------------------------------------------------- */
/* bytes [src, src+size) of the image -> [dst, dst+size), dst < src, on one stream (pieces no longer than the shift never overlap) */
static int move_down(char *img, uint64_t dst, uint64_t src, uint64_t size, hipStream_t st) {
    if (dst == src || !size) return 0;
    const uint64_t step = src - dst;
    for (uint64_t done = 0; done < size; done += step) {
        const uint64_t n = size - done < step ? size - done : step;
        if (hipMemcpyAsync(img + dst + done, img + src + done, n, hipMemcpyDeviceToDevice, st) != hipSuccess) return 1;
    }
    return 0;
}

/* utk_vote_rec of every label, rank order; 0 when some label is not of the shape the table describes (then vote_k reads label
 * bytes, as it does for databases with 32-bit label indices) */
static int build_vote_table(const utree_ctr *ctr, utk_vote_rec *out) {
    const uint32_t n = ctr->info.n_labels;
    if (!n || n > 65535u || ctr->info.I != 2) return 0;
    /* prefixes interned in an open-addressing table: {hash, first rank with these bytes} */
    size_t cap = 1;
    while (cap < (size_t)n * 16) cap <<= 1;
    uint64_t *hh = (uint64_t *)calloc(cap, sizeof(uint64_t));
    uint32_t *hv = (uint32_t *)malloc(cap * sizeof(uint32_t)), *hl = (uint32_t *)malloc(cap * sizeof(uint32_t));
    int ok = hh && hv && hl;
    for (uint32_t r = 0; ok && r < n; ++r) {
        const uint32_t ix = ctr->rank2ix[r];
        const char *s = ctr->labels[ix];
        const uint32_t len = ctr->label_len[ix];
        utk_vote_rec *v = &out[r];
        if (len > 255u) { ok = 0; break; }
        uint32_t t = 0;
        uint64_t h = 1469598103934665603ull;                                    /* FNV-1a over the prefix so far */
        for (uint32_t i = 0; i <= len; ++i) {
            if (i < len && s[i] != ';') { h = (h ^ (uint8_t)s[i]) * 1099511628211ull; continue; }
            if (i < len && s[i] == '\0') { ok = 0; break; }
            if (t >= 8) { ok = 0; break; }                                       /* more tokens than the table has levels */
            /* token t ends at i (';' or the label's end): intern bytes [0, i) */
            const uint64_t key = h | 1ull;                                      /* (0 marks a free slot) */
            size_t at = (size_t)(key * 0x9E3779B97F4A7C15ull >> 17) & (cap - 1);
            uint32_t id = r;
            for (;; at = (at + 1) & (cap - 1)) {
                if (!hh[at]) { hh[at] = key; hv[at] = r; hl[at] = i; break; }
                if (hh[at] == key && hl[at] == i && !memcmp(ctr->labels[ctr->rank2ix[hv[at]]], s, i)) { id = hv[at]; break; }
            }
            v->pid[t] = (uint16_t)id; v->tok_end[t] = (uint8_t)i;
            v->exists |= (uint8_t)(1u << t);
            if (i < len) v->more |= (uint8_t)(1u << t);
            if (i > 0 && s[i - 1] == '_') v->us |= (uint8_t)(1u << t);
            ++t;
            if (i < len) h = (h ^ (uint8_t)';') * 1099511628211ull;
        }
        if (!ok) break;
        v->n_tok = (uint8_t)t; v->len = (uint8_t)len; v->ix = (uint16_t)ix;
        for (; t < 8; ++t) { v->pid[t] = 0xFFFFu; v->tok_end[t] = (uint8_t)len; }
    }
    free(hh); free(hv); free(hl);
    return ok;
}

typedef struct {
    utree_dev *d;
    const utree_ctr *ctr;
    uint32_t *d_ix2rank;
    unsigned long long *d_counters;
    unsigned long long *d_invalid;          /* nodes whose label index is >= the number of labels */
    hipStream_t stream;
} builder;

static int build_begin(builder *b, const utree_ctr *ctr, int device, int fine_bits, void *d_image, size_t image_bytes,
                       hipStream_t stream) {
    int rc = UTREE_OK, n_cu = 0;
    double tb0 = now_s();
    memset(b, 0, sizeof *b);
    if ((rc = device_ok(device, &n_cu))) return rc;
    double tb1 = now_s();
    utree_dev *d = (utree_dev *)calloc(1, sizeof *d);
    if (!d) return UTREE_E_NOMEM;
    b->d = d; b->ctr = ctr; b->stream = stream;
    d->device = device; d->n_cu = n_cu;
    lanes_ring_init(d);
    layout(ctr, (uint32_t)utree_pick_fine_bits(ctr, fine_bits), &d->hdr);
    if (d_image) {
        if (image_bytes < d->hdr.total_bytes || ((uintptr_t)d_image & 127u)) { rc = UTREE_E_ARG; goto fail; }   /* buckets are 128-byte lines */
        d->image = d_image; d->owns = 0;
    } else {
        HIPCHK(hipMalloc(&d->image, d->hdr.total_bytes));
        d->owns = 1;
    }
    if (timing_on()) fprintf(stderr, "[utree_amd] image: device init %.3f s, hipMalloc(%.1f GiB) %.3f s\n", tb1 - tb0, (double)d->hdr.total_bytes / 1073741824.0, now_s() - tb1);
    d->image_bytes = d->hdr.total_bytes;
    bind_image(d);
    char *img = (char *)d->image;
    HIPCHK(hipMemsetAsync(img, 0, UTREE_IMG_HEADER_BYTES, stream));
    HIPCHK(hipMemsetAsync(img + d->hdr.off_irreg, 0, (1u << 24) / 8, stream));
    /* labels in strcmp order */
    {
        uint32_t n = d->hdr.n_labels;
        uint32_t *loff = (uint32_t *)malloc(((size_t)n + 1) * 4);
        char *blob = (char *)calloc(d->hdr.label_blob_bytes + 64, 1);
        if (!loff || !blob) { free(loff); free(blob); rc = UTREE_E_NOMEM; goto fail; }
        uint64_t o = 0;
        for (uint32_t r = 0; r < n; ++r) {
            uint32_t ix = ctr->rank2ix[r];
            loff[r] = (uint32_t)o;
            memcpy(blob + o, ctr->labels[ix], ctr->label_len[ix]);
            o += (uint64_t)ctr->label_len[ix] + 1;
        }
        loff[n] = (uint32_t)o;
        hipError_t e1 = hipMemcpyAsync(img + d->hdr.off_label_off, loff, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, stream);
        hipError_t e2 = hipMemcpyAsync(img + d->hdr.off_label_blob, blob, d->hdr.label_blob_bytes + 64, hipMemcpyHostToDevice, stream);
        hipError_t e3 = hipMemcpyAsync(img + d->hdr.off_rank2ix, ctr->rank2ix, (size_t)n * 4, hipMemcpyHostToDevice, stream);
        utk_vote_rec *vt = (utk_vote_rec *)calloc((size_t)n + 1, sizeof(utk_vote_rec));
        hipError_t e5 = hipSuccess;
        if (vt && build_vote_table(ctr, vt)) {
            d->hdr.flags |= UTREE_F_VOTE_TABLE;
            e5 = hipMemcpyAsync(img + d->hdr.off_vote, vt, (size_t)n * sizeof(utk_vote_rec), hipMemcpyHostToDevice, stream);
        }
        hipError_t e4 = hipStreamSynchronize(stream);
        free(loff); free(blob); free(vt);
        HIPCHK(e1); HIPCHK(e2); HIPCHK(e3); HIPCHK(e4); HIPCHK(e5);
        bind_image(d);
        HIPCHK(hipMalloc((void **)&b->d_ix2rank, (size_t)n * 4));
        HIPCHK(hipMemcpyAsync(b->d_ix2rank, ctr->ix2rank, (size_t)n * 4, hipMemcpyHostToDevice, stream));
        HIPCHK(hipMalloc((void **)&b->d_invalid, 8));
        HIPCHK(hipMemsetAsync(b->d_invalid, 0, 8, stream));
    }
    return UTREE_OK;
fail:
    if (b->d_ix2rank) hipFree(b->d_ix2rank);
    if (b->d_invalid) hipFree(b->d_invalid);
    if (d->owns && d->image) hipFree(d->image);
    if (d->rank_state) hipFree(d->rank_state);
    if (d->lanes_ring) hipHostFree((void *)d->lanes_ring);
    free(d);
    b->d = NULL;
    return rc;
}

/* records [first, first+count) given as packed on-disk bytes in HBM */
static int build_chunk(builder *b, const void *d_raw, uint64_t first, uint64_t count) {
    utree_dev *d = b->d;
    uint64_t *recs = (uint64_t *)((char *)d->image + d->hdr.off_recs) + first * d->hdr.rec_words;
    int e = utk_repack(d->hdr.W, d->hdr.I, d_raw, count, b->d_ix2rank, d->hdr.n_labels, recs, b->d_invalid, b->stream);
    if (e) { utree_dev_set_hip_error(e, "utk_repack"); return UTREE_E_HIP; }
    return UTREE_OK;
}

static int build_finish(builder *b, const void *d_binix_raw) {
    int rc = UTREE_OK;
    utree_dev *d = b->d;
    const utree_ctr *ctr = b->ctr;
    hipStream_t st = b->stream;
    char *img = (char *)d->image;
    uint64_t *recs = (uint64_t *)(img + d->hdr.off_recs);
    void *coarse = img + d->hdr.off_coarse;
    const int off64 = (d->hdr.flags & UTREE_F_OFF64) != 0;
    unsigned long long counters[2] = {0, 0}, invalid = 0;
    HIPCHK(hipMemcpyAsync(&invalid, b->d_invalid, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMalloc((void **)&b->d_counters, 16));
    HIPCHK(hipMemsetAsync(b->d_counters, 0, 16, st));
    KCHK(utk_fill_recs_pad(recs + d->hdr.n_nodes * d->hdr.rec_words, 8 * d->hdr.rec_words, st));
    KCHK(utk_widen_binix(d_binix_raw, ctr->info.binix_width, off64, coarse, st));
    KCHK(utk_validate(d->hdr.W, d->hdr.I, off64, coarse, recs, d->hdr.n_nodes, (uint32_t *)(img + d->hdr.off_irreg),
                      b->d_counters, st));
    HIPCHK(hipMemcpyAsync(counters, b->d_counters, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (timing_on()) fprintf(stderr, "[utree_amd] image: bin table checked (%llu irregular bins%s)\n", counters[0], counters[1] ? ", not monotone" : "");
    d->hdr.n_irregular = counters[0];
    if (invalid) d->hdr.flags |= UTREE_F_INVALID_RANKS;
    uint64_t full_bytes = d->hdr.total_bytes;
    if (d->hdr.W == 4) {
        /* PACKSIZE=16: the direct-address table from the FILE records (image_build.hip: direct_*_k); the records then leave the image */
        uint64_t c0 = 0, cN = 0;
        if (!counters[1]) {
            if (ctr->info.binix_width == 4) { uint32_t a, z; memcpy(&a, ctr->binix_raw, 4); memcpy(&z, (const char *)ctr->binix_raw + 4 * (size_t)(UTREE_NUMBINS - 1), 4); c0 = a; cN = z; }
            else { memcpy(&c0, ctr->binix_raw, 8); memcpy(&cN, (const char *)ctr->binix_raw + 8 * (size_t)(UTREE_NUMBINS - 1), 8); }
        } else HIPCHK(hipMemsetAsync(img + d->hdr.off_irreg, 0xFF, (1u << 24) / 8, st));
        KCHK(utk_build_direct(d->hdr.I, off64, counters[1] != 0, coarse, recs, d->hdr.n_nodes, c0, cN - c0, (const uint32_t *)(img + d->hdr.off_irreg), counters[0],
                              img + d->hdr.off_table, st));
        HIPCHK(hipStreamSynchronize(st));
        if (counters[1]) d->hdr.flags |= UTREE_F_GENERIC; else if (counters[0]) d->hdr.flags |= UTREE_F_IRREGULAR;   /* (what the table was built from: informational) */
        d->hdr.n_min = 0;
        d->hdr.total_bytes = d->hdr.off_recs;                    /* (off_recs is the last area: the image ends in front of it) */
        if (timing_on()) fprintf(stderr, "[utree_amd] image: direct-address table of 2^32 x %u bytes built (%llu irregular bins%s answered by the reference's probe order)\n",
                                 d->hdr.I, counters[0], counters[1] ? ", bin table not monotone: every bin" : "");
    } else
    if (counters[1]) {
        /* bin table not monotone (never written by the reference's COMPRESS): trust it verbatim like the
         * reference does -- every bin takes the exact probe path over [BinIx[p], BinIx[p+1]) */
        d->hdr.flags |= UTREE_F_GENERIC;
        d->hdr.fine_bits = 0;
        d->hdr.n_min = 0;
        HIPCHK(hipMemsetAsync(img + d->hdr.off_irreg, 0xFF, (1u << 24) / 8, st));   /* the buckets are never read: every bin is "irregular" */
    } else {
        if (counters[0]) d->hdr.flags |= UTREE_F_IRREGULAR;
        /* a monotone table reaches the contiguous node range [BinIx[0], BinIx[2^24]) */
        uint64_t c0, cN;
        if (ctr->info.binix_width == 4) { uint32_t a, z; memcpy(&a, ctr->binix_raw, 4); memcpy(&z, (const char *)ctr->binix_raw + 4 * (size_t)(UTREE_NUMBINS - 1), 4); c0 = a; cN = z; }
        else { memcpy(&c0, ctr->binix_raw, 8); memcpy(&cN, (const char *)ctr->binix_raw + 8 * (size_t)(UTREE_NUMBINS - 1), 8); }
        d->hdr.n_min = cN - c0;
        HIPCHK(hipMemsetAsync(b->d_counters, 0, 16, st));
        HIPCHK(hipMemcpyAsync(img, &d->hdr, sizeof d->hdr, hipMemcpyHostToDevice, st));   /* the kernels read the region table there */
        {
            uint64_t n_min = 0;
            int views = 0;
            KCHK(utk_build_min(d->hdr.W, d->hdr.I, off64, coarse, recs, c0, d->hdr.n_min, dup_capacity(d->hdr.n_nodes), (const uint64_t *)(img + offsetof(utree_image_header, regions)),
                               d->hdr.regions, d->hdr.n_slots, d->hdr.bucket_words, (uint64_t *)(img + d->hdr.off_table), (uint64_t *)(img + d->hdr.off_mrecs),
                               (uint32_t *)(img + d->hdr.off_irreg), b->d_counters, &n_min, &views, st));
            if (timing_on()) fprintf(stderr, "[utree_amd] image: %llu records for %llu nodes (second views%s)\n", (unsigned long long)n_min, (unsigned long long)d->hdr.n_min, views ? "" : ": too many, none stored");
            d->hdr.n_min = n_min;
            if (views) d->hdr.flags |= UTREE_F_STRAND_VIEWS;
        }
        KCHK(utk_fill_recs_pad((uint64_t *)(img + d->hdr.off_mrecs) + d->hdr.n_min * d->hdr.rec_words, 8 * d->hdr.rec_words, st));
        HIPCHK(hipMemcpyAsync(counters, b->d_counters, 16, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (counters[0]) {
            /* buckets with more nodes than an overflow descriptor can count (a minimizer shared by millions of k-mers, e.g.
             * A^16, whose hash is 0): the bins of their nodes were flagged irregular, so their words take the reference's own
             * probe sequence over the FILE records.  Any `.ctr` the reference loads is searched. */
            d->hdr.flags |= UTREE_F_IRREGULAR;
            d->hdr.n_irregular += counters[1];
            if (timing_on()) fprintf(stderr, "[utree_amd] image: %llu bucket(s) beyond the overflow descriptor's range: %llu bins take the exact-probe path\n", counters[0], counters[1]);
        }
        /* Of the sorted MIN array only the buckets' overflow runs are read again: pack them and let everything behind them move
         * down (the array of ALL nodes was a third of the image).  The FILE records go last and stay only when some bin needs
         * the exact-probe path. */
        {
            uint64_t kept = 0;
            const utree_image_header o = d->hdr;
            int chains = 0;
            KCHK(utk_compact_overflow(o.W, o.I, (uint64_t *)(img + o.off_table), o.n_slots, o.bucket_words, (uint64_t *)(img + o.off_mrecs), &kept, &chains, st));
            if (chains) d->hdr.flags |= UTREE_F_OVF_CHAINS;
            KCHK(utk_fill_recs_pad((uint64_t *)(img + o.off_mrecs) + kept * o.rec_words, 8 * o.rec_words, st));
            utree_image_header *h = &d->hdr;
            const uint64_t coarse_b = (uint64_t)UTREE_NUMBINS * ((o.flags & UTREE_F_OFF64) ? 8 : 4), irreg_b = (1u << 24) / 8;
            const uint64_t loff_b = ((uint64_t)o.n_labels + 1) * 4, blob_b = o.label_blob_bytes + 64, r2i_b = (uint64_t)o.n_labels * 4;
            const uint64_t vote_b = (uint64_t)o.n_labels * sizeof(utk_vote_rec);
            const uint64_t recs_b = (o.n_nodes + 8) * o.rec_words * 8;
            uint64_t off = align_up(o.off_mrecs + (kept + 8) * o.rec_words * 8, 4096);
            h->n_min = kept;
            h->off_coarse = off; off = align_up(off + coarse_b, 256);
            h->off_irreg = off; off = align_up(off + irreg_b, 256);
            h->off_label_off = off; off = align_up(off + loff_b, 256);
            h->off_label_blob = off; off = align_up(off + blob_b, 256);
            h->off_rank2ix = off; off = align_up(off + r2i_b, 256);
            h->off_vote = off; off = align_up(off + vote_b, 4096);
            h->off_recs = off;
            if (h->flags & UTREE_F_IRREGULAR) off = align_up(off + recs_b, 4096);
            h->total_bytes = off;
            if (move_down(img, h->off_coarse, o.off_coarse, coarse_b, st) || move_down(img, h->off_irreg, o.off_irreg, irreg_b, st) ||
                move_down(img, h->off_label_off, o.off_label_off, loff_b, st) || move_down(img, h->off_label_blob, o.off_label_blob, blob_b, st) ||
                move_down(img, h->off_rank2ix, o.off_rank2ix, r2i_b, st) || move_down(img, h->off_vote, o.off_vote, vote_b, st) ||
                ((h->flags & UTREE_F_IRREGULAR) && move_down(img, h->off_recs, o.off_recs, recs_b, st))) { rc = UTREE_E_HIP; goto fail; }
            HIPCHK(hipStreamSynchronize(st));
            if (timing_on()) fprintf(stderr, "[utree_amd] image: %llu of %llu nodes in overflow runs; packed image %.2f GiB (built in %.2f GiB)\n",
                                     (unsigned long long)kept, (unsigned long long)o.n_nodes, (double)h->total_bytes / 1073741824.0, (double)full_bytes / 1073741824.0);
        }
    }
    d->image_bytes = d->hdr.total_bytes;
    HIPCHK(hipMemcpyAsync(img, &d->hdr, sizeof d->hdr, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    /* an image this handle allocated itself is moved into an allocation of its final size: the build area goes back to the device */
    if (d->owns && d->hdr.total_bytes + ((uint64_t)256 << 20) < full_bytes && !getenv("UTREE_KEEP_BUILD_AREA")) {
        void *small = NULL;
        if (hipMalloc(&small, d->hdr.total_bytes) == hipSuccess) {
            hipError_t e1 = hipMemcpyAsync(small, d->image, d->hdr.total_bytes, hipMemcpyDeviceToDevice, st), e2 = hipStreamSynchronize(st);
            if (e1 == hipSuccess && e2 == hipSuccess) { hipFree(d->image); d->image = small; }
            else { hipFree(small); (void)hipGetLastError(); }
        } else (void)hipGetLastError();
    }
    bind_image(d);
fail:
    if (b->d_counters) hipFree(b->d_counters);
    if (b->d_ix2rank) hipFree(b->d_ix2rank);
    if (b->d_invalid) hipFree(b->d_invalid);
    b->d_counters = NULL; b->d_ix2rank = NULL; b->d_invalid = NULL;
    if (rc) { utree_dev_free(d); b->d = NULL; }
    return rc;
}

int utree_dev_build(const utree_ctr *ctr, int device, int fine_bits, const void *d_binix, const void *d_records,
                    void *d_image, size_t image_bytes, void *stream, utree_dev **out) {
    if (!ctr || !d_binix || !d_records || !out) return UTREE_E_ARG;
    *out = NULL;
    builder b;
    int rc = build_begin(&b, ctr, device, fine_bits, d_image, image_bytes, (hipStream_t)stream);
    if (rc) return rc;
    rc = build_chunk(&b, d_records, 0, ctr->info.n_nodes);
    if (rc) { hipFree(b.d_ix2rank); hipFree(b.d_invalid); utree_dev_free(b.d); return rc; }
    rc = build_finish(&b, d_binix);
    if (rc) return rc;
    *out = b.d;
    return UTREE_OK;
}

/* where the calling process's last utree_dev_upload spent its time (bench.py's database-load figure) */
static double g_upload_s[4];
int utree_dev_upload_seconds(double *h_out4) {
    if (!h_out4) return UTREE_E_ARG;
    for (int i = 0; i < 4; ++i) h_out4[i] = g_upload_s[i];
    return UTREE_OK;
}

int utree_dev_upload(const utree_ctr *ctr, int device, int fine_bits, utree_dev **out) {
    if (!ctr || !out) return UTREE_E_ARG;
    *out = NULL;
    if (!ctr->path && !ctr->h_records) return UTREE_E_ARG;
    builder b;
    double t0 = now_s();
    int rc = build_begin(&b, ctr, device, fine_bits, NULL, 0, NULL);
    if (rc) return rc;
    double t1 = now_s(), t2 = t1, t3 = t1;
    const size_t SZ = ctr->info.SZ;
    /* (the node dump goes file -> pinned memory -> HBM in 128 MiB pieces, two in flight; the page-cache copy is the cost -- one thread moves
     * ~5 GB/s -- so a small team reads each piece: 8.6 GB in 0.6 s instead of 1.7) */
    const size_t chunk_recs = ((size_t)128 << 20) / SZ;
    const size_t chunk_bytes = chunk_recs * SZ;
    void *h_pin[2] = {NULL, NULL}, *d_raw[2] = {NULL, NULL}, *d_binix = NULL;
    hipEvent_t ev[2] = {NULL, NULL};
    int fd = -1;
    for (int i = 0; i < 2; ++i) {
        HIPCHK(hipHostMalloc(&h_pin[i], chunk_bytes, hipHostMallocDefault));
        HIPCHK(hipMalloc(&d_raw[i], chunk_bytes));
        HIPCHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    }
    if (ctr->path) {
        fd = open(ctr->path, O_RDONLY);
        if (fd < 0) { rc = UTREE_E_IO; goto fail; }
    }
    uint64_t done = 0, N = ctr->info.n_nodes;
    for (int slot = 0; done < N; slot ^= 1) {
        uint64_t cnt = N - done < chunk_recs ? N - done : chunk_recs;
        size_t bytes = (size_t)cnt * SZ;
        HIPCHK(hipEventSynchronize(ev[slot]));                 /* the pinned buffer is free again */
        if (fd >= 0) {
            int T = 8, bad = 0;
            if ((size_t)T > bytes / ((size_t)4 << 20) + 1) T = (int)(bytes / ((size_t)4 << 20) + 1);
#pragma omp parallel for num_threads(T) schedule(static, 1) reduction(| : bad)
            for (int t = 0; t < T; ++t) {
                size_t a = bytes * (size_t)t / (size_t)T, e = bytes * (size_t)(t + 1) / (size_t)T;
                while (a < e) {
                    ssize_t r = pread(fd, (char *)h_pin[slot] + a, e - a, (off_t)(ctr->records_file_off + done * SZ + a));
                    if (r <= 0) { bad |= 1; break; }
                    a += (size_t)r;
                }
            }
            if (bad) { rc = UTREE_E_FORMAT; goto fail; }                   /* "Error in reading tree." itree.c:768 */
        } else memcpy(h_pin[slot], ctr->h_records + done * SZ, bytes);
        HIPCHK(hipMemcpyAsync(d_raw[slot], h_pin[slot], bytes, hipMemcpyHostToDevice, NULL));
        rc = build_chunk(&b, d_raw[slot], done, cnt);
        if (rc) goto fail;
        HIPCHK(hipEventRecord(ev[slot], NULL));
        done += cnt;
    }
    HIPCHK(hipDeviceSynchronize());
    t2 = now_s();
    if (timing_on()) fprintf(stderr, "[utree_amd] image: %llu nodes streamed and repacked in %.3f s\n", (unsigned long long)N, t2 - t1);
    HIPCHK(hipMalloc(&d_binix, (size_t)UTREE_NUMBINS * ctr->info.binix_width));
    HIPCHK(hipMemcpyAsync(d_binix, ctr->binix_raw, (size_t)UTREE_NUMBINS * ctr->info.binix_width, hipMemcpyHostToDevice, NULL));
    rc = build_finish(&b, d_binix);
    t3 = now_s();
    if (timing_on()) fprintf(stderr, "[utree_amd] image: alloc+labels %.3f s, stream+repack nodes %.3f s, validate+sort+table %.3f s\n", t1 - t0, t2 - t1, t3 - t2);
    g_upload_s[0] = t1 - t0; g_upload_s[1] = t2 - t1; g_upload_s[2] = t3 - t2; g_upload_s[3] = t3 - t0;
    if (!rc) *out = b.d;
    b.d = NULL;
fail:
    if (fd >= 0) close(fd);
    hipDeviceSynchronize();
    for (int i = 0; i < 2; ++i) {
        if (h_pin[i]) hipHostFree(h_pin[i]);
        if (d_raw[i]) hipFree(d_raw[i]);
        if (ev[i]) hipEventDestroy(ev[i]);
    }
    if (d_binix) hipFree(d_binix);
    if (rc && b.d) { if (b.d_ix2rank) hipFree(b.d_ix2rank); if (b.d_invalid) hipFree(b.d_invalid); utree_dev_free(b.d); }
    return rc;
}

int utree_dev_image(const utree_dev *dev, void **d_image, size_t *bytes) {
    if (!dev) return UTREE_E_ARG;
    if (d_image) *d_image = dev->image;
    if (bytes) *bytes = dev->image_bytes;
    return UTREE_OK;
}

int utree_dev_attach(const utree_ctr *ctr, int device, void *d_image, size_t bytes, utree_dev **out) {
    int rc = UTREE_OK, n_cu = 0;
    if (!d_image || !out || bytes < UTREE_IMG_HEADER_BYTES || ((uintptr_t)d_image & 127u)) return UTREE_E_ARG;   /* buckets are 128-byte lines */
    *out = NULL;
    if ((rc = device_ok(device, &n_cu))) return rc;
    utree_dev *d = (utree_dev *)calloc(1, sizeof *d);
    if (!d) return UTREE_E_NOMEM;
    d->device = device; d->n_cu = n_cu; d->image = d_image; d->owns = 0;
    lanes_ring_init(d);
    HIPCHK(hipMemcpy(&d->hdr, d_image, sizeof d->hdr, hipMemcpyDeviceToHost));
    if (d->hdr.magic != UTREE_IMG_MAGIC || d->hdr.version != UTREE_IMG_VERSION || d->hdr.total_bytes > bytes ||
        (d->hdr.bucket_words != 8 && d->hdr.bucket_words != 16)) { rc = UTREE_E_FORMAT; goto fail; }
    if (ctr && (ctr->info.W != d->hdr.W || ctr->info.I != d->hdr.I || ctr->info.n_nodes != d->hdr.n_nodes ||
                ctr->info.n_labels != d->hdr.n_labels)) { rc = UTREE_E_ARG; goto fail; }
    d->image_bytes = d->hdr.total_bytes;
    bind_image(d);
    *out = d;
    return UTREE_OK;
fail:
    if (d->lanes_ring) hipHostFree((void *)d->lanes_ring);
    free(d);
    return rc;
}

static void lanes_ring_init(utree_dev *d) {
    void *p = NULL;
    if (hipSetDevice(d->device) == hipSuccess && hipHostMalloc(&p, 128 * sizeof(unsigned long long), hipHostMallocDefault) == hipSuccess) {
        memset(p, 0xFF, 128 * sizeof(unsigned long long));
        d->lanes_ring = (volatile unsigned long long *)p;
    }
}

void utree_dev_free(utree_dev *d) {
    if (!d) return;
    hipSetDevice(d->device);
    if (d->lanes_ring) hipHostFree((void *)d->lanes_ring);
    for (int i = 0; i < d->n_events; ++i) hipEventDestroy(d->events[i]);
    if (d->search_ctx) utree_search_ctx_free(d->search_ctx);
    if (d->owns && d->image) hipFree(d->image);
    if (d->rank_state) hipFree(d->rank_state);
    free(d);
}

int utree_dev_get_info(const utree_dev *d, utree_dev_info *info) {
    if (!d || !info) return UTREE_E_ARG;
    info->fine_bits = d->hdr.fine_bits;
    info->record_bytes = d->hdr.rec_words * 8;
    info->image_bytes = d->image_bytes;
    info->irregular_bins = d->hdr.n_irregular;
    info->generic_mode = (d->hdr.flags & UTREE_F_GENERIC) != 0;
    info->device = d->device;
    info->vote_table = d->kimg.vote_tab != NULL;
    info->lane_pass = utk_lanes_image_ok(&d->kimg) != 0;
    info->bucket_bytes = 8u * d->hdr.bucket_words;
    info->strand_views = (d->hdr.flags & UTREE_F_STRAND_VIEWS) ? 1u : 0u;
    if (d->hdr.flags & UTREE_F_DIRECT) info->bucket_bytes = 0;
    info->overflow_chains = (d->hdr.flags & UTREE_F_OVF_CHAINS) ? 1u : 0u;
    info->pad0 = 0;
    info->overflow_bytes = (d->hdr.flags & UTREE_F_DIRECT) ? 0 : d->hdr.n_min * d->hdr.rec_words * 8;
    return UTREE_OK;
}

/* ---- batches --------------------------------------------------------------------------------- */
#define LONG_BLOCKS_PER_CU 8

/* reads of up to this many staged bases take the wave-per-read mid pass, longer ones classify_long_k */
static uint32_t mid_limit(void) {
    static uint32_t v = 0;
    if (!v) {
        const char *e = getenv("UTREE_MID_LIMIT");
        v = UTREE_MID_DEFAULT;
        if (e && atoi(e) >= (int)UTREE_SHORT2_CAP && atoi(e) <= (int)UTREE_MID_CAP) v = (uint32_t)atoi(e);
    }
    return v;
}

/* UTREE_LANE_PASS=0 keeps every batch on the wave-per-read kernels (comparison runs) */
static int lanes_enabled(void) { const char *e = getenv("UTREE_LANE_PASS"); return !(e && e[0] == '0'); }

static void ring_lock(utree_dev *d) { while (__atomic_test_and_set(&d->ring_busy, __ATOMIC_ACQUIRE)) ; }
static void ring_unlock(utree_dev *d) { __atomic_clear(&d->ring_busy, __ATOMIC_RELEASE); }

/* Two words come back from every batch without a wait (a ring of pinned slots; ~0 = not arrived or consumed): the reads the lane-per-read
 * pass left to the wave-per-read kernel, and the batch's error word.
 *  - The left-over counts of the last UTREE_LANES_WINDOW lane-pass batches decide whether the pass is worth running: with more than a
 *    quarter of at least 256 Ki reads left over (reads with more distinct labels than a lane's tally table holds, say) the next batches
 *    go to the wave-per-read kernels alone -- except every eighth, which keeps the window current, so the pass comes back when the
 *    input changes.  (Round 2 latched this for the handle's lifetime.)
 *  - A non-zero error word is kept until utree_classify_poll hands it out.
 * Callers on several threads share the handle (the lanes of search_dev.c): everything here happens under the ring's lock. */
static void ring_collect(utree_dev *d) {
    if (!d->lanes_ring) return;
    ring_lock(d);
    for (unsigned i = 0; i < 64; ++i) {
        const unsigned long long left = d->lanes_ring[2 * i], err = d->lanes_ring[2 * i + 1];
        if (left == ~0ull || err == ~0ull) continue;
        d->lanes_ring[2 * i] = ~0ull; d->lanes_ring[2 * i + 1] = ~0ull;
        d->ring_inflight[i] = 0;
        if (err && !d->dev_error) d->dev_error = err;
        if (timing_on()) fprintf(stderr, "[utree_amd] batch report: %llu of %u reads left by the lane pass, error word %llu\n", left, d->lanes_ring_reads[i], err);
        if (d->lanes_ring_reads[i]) {
            /* (a batch the pass did well on ends a bad spell at once: the window starts again with it) */
            if (left * 4 <= d->lanes_ring_reads[i]) { unsigned long long n = 0, l = 0; for (unsigned w = 0; w < UTREE_LANES_WINDOW; ++w) { n += d->win_reads[w]; l += d->win_left[w]; }
                if (n >= (1u << 18) && l * 4 > n) { memset(d->win_left, 0, sizeof d->win_left); memset(d->win_reads, 0, sizeof d->win_reads); } }
            const unsigned w = d->win_next++ % UTREE_LANES_WINDOW;
            d->win_left[w] = left; d->win_reads[w] = d->lanes_ring_reads[i];
        }
        d->lanes_ring_reads[i] = 0;
    }
    ring_unlock(d);
}
/* run the lane-per-read pass on this batch? */
static int lanes_worth(utree_dev *d) {
    unsigned long long n = 0, l = 0;
    ring_lock(d);
    for (unsigned w = 0; w < UTREE_LANES_WINDOW; ++w) { n += d->win_reads[w]; l += d->win_left[w]; }
    int go = !(n >= (1u << 18) && l * 4 > n);
    if (!go && (++d->lanes_skipped & 7u) == 0) go = 1;                      /* a probe */
    ring_unlock(d);
    return go;
}
/* the batch's two words into the next free ring slot (on the batch's stream, behind its kernels) */
static int ring_post(utree_dev *d, const utk_workspace *w, uint32_t lane_reads, hipStream_t st) {
    if (!d->lanes_ring) return 0;
    ring_lock(d);
    unsigned slot = 64;
    for (unsigned k = 0; k < 64; ++k) {
        const unsigned c = (d->lanes_ring_next + k) & 63u;
        if (!d->ring_inflight[c]) { slot = c; break; }
    }
    if (slot < 64) { d->lanes_ring_next = slot + 1; d->ring_inflight[slot] = 1; d->lanes_ring_reads[slot] = lane_reads; }
    ring_unlock(d);
    if (slot == 64) {
        /* all 64 slots wait for their copies -- more than 64 batches in flight on one handle: this batch's words are fetched with a wait (no
         * batch's error word is dropped) */
        unsigned long long two[2] = {0, 0};
        if (hipMemcpyAsync(two, w->cursors + UTREE_CUR_MID, 16, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 1;
        ring_lock(d);
        if (two[1] && !d->dev_error) d->dev_error = two[1];
        ring_unlock(d);
        ring_collect(d);
        return 0;
    }
    if (hipMemcpyAsync((void *)&d->lanes_ring[2 * slot], w->cursors + UTREE_CUR_MID, 16, hipMemcpyDeviceToHost, st) != hipSuccess) {
        ring_lock(d);                                                   /* the copy was not posted: the slot is free again */
        d->ring_inflight[slot] = 0; d->lanes_ring_reads[slot] = 0;
        ring_unlock(d);
        return 1;
    }
    return 0;
}

int utree_classify_poll(utree_dev *d) {
    if (!d) return UTREE_E_ARG;
    ring_collect(d);
    ring_lock(d);
    const unsigned long long e = d->dev_error;
    d->dev_error = 0;
    ring_unlock(d);
    if (e) {
        snprintf(g_hip_msg, sizeof g_hip_msg, "a batch's kernels reported error %llu (%s)", e,
                 e == UTREE_DEVERR_TALLY_CAP ? "tally lists beyond the workspace: total_bases / max_len did not describe the batch" :
                 e == UTREE_DEVERR_LONG_CAP ? "more long reads than the workspace holds" :
                 e == UTREE_DEVERR_PIECES_CAP ? "more pieces of long reads than the workspace holds" : "unknown");
        return UTREE_E_DEVICE;
    }
    return UTREE_OK;
}

/* reads of at least this many bases are "long" for the lane-per-read pass: beyond sixteen lanes, or -- with both strands -- beyond
 * what the wave-per-read pass that finishes its left-overs stages */
static uint32_t lanes_long_min(const utree_dev *d, int do_rc) {
    const uint32_t a = utk_lanes_max_len(&d->kimg), b = do_rc ? (UTREE_MID_CAP - 1u) / 2u : UTREE_MID_CAP;
    return (a < b ? a : b) + 1u;
}

static void carve(const utree_dev *d, void *ws, uint32_t n_reads, uint64_t total_bases, uint32_t max_len, int do_rc,
                  utk_workspace *w, size_t *bytes) {
    uint64_t off = 0;
    char *b = (char *)ws;
    w->cursors = (unsigned long long *)(b + off); off = align_up(off + UTREE_CURSOR_BYTES, 256);
    /* (rank, count) lists: a read needs at most one entry per window.  Waves sub-allocate from UTREE_TALLY_CHUNK-entry chunks: a
     * refill abandons fewer than UTREE_TALLY_CHUNK / 16 entries of the old chunk (longer lists reserve exactly their length), i.e.
     * less than 1/15 of what it used, and every wave may leave ONE chunk part-used -- the resident waves of the 150-bp-class pass
     * (8 per SIMD: 32 per CU) and, in a batch that has mid-length reads, those of the mid pass (5 workgroups of 4 per CU) on top:
     * both passes draw from the same cursor.  A kernel that would pass the bound raises the batch's error word (UTREE_DEVERR_TALLY_CAP)
     * instead of writing there.  UTREE_TEST_TALLY_CAP: test hook, a capacity too small on purpose. */
    /* (a batch of the lane-per-read pass runs fewer waves of its own, but leaves reads to a listed pass: both terms, always; a mixed
     * batch runs up to five class launches one after the other, each leaving its waves' chunks part-used) */
    const uint64_t waves_per_cu = 32 + 20 + 5 * 12;
    w->tally_cap = ((do_rc ? 2 : 1) * total_bases + (uint64_t)n_reads) * 9 / 8 + (uint64_t)d->n_cu * waves_per_cu * UTREE_TALLY_CHUNK + 4096;
    { const char *e = getenv("UTREE_TEST_TALLY_CAP"); if (e && atoll(e) > 0 && (uint64_t)atoll(e) < w->tally_cap) w->tally_cap = (uint64_t)atoll(e); }
    /* (a kernel that finds the capacity exceeded writes its list at the start of the area instead: room for the longest single list) */
    { uint64_t room = w->tally_cap; if (room < UTREE_TALLY_CHUNK) room = UTREE_TALLY_CHUNK; if (room < d->hdr.n_labels) room = d->hdr.n_labels;
      w->tally = (uint64_t *)(b + off); off = align_up(off + room * 8, 256); }
    w->long_list = (uint32_t *)(b + off); off = align_up(off + (uint64_t)n_reads * 4, 256);
    w->mid_list = (uint32_t *)(b + off); off = align_up(off + (uint64_t)n_reads * 4, 256);
    uint64_t max_total = do_rc ? 2 * (uint64_t)max_len + 1 : max_len;
    w->long_blocks = 0; w->hist = NULL; w->touch = NULL;
    /* the main wave-per-read pass comes in two sizes; the larger one when the batch's longest read needs it */
    w->short_cap = max_total > UTREE_SHORT_CAP ? UTREE_SHORT2_CAP : UTREE_SHORT_CAP;
    w->mid_reads = max_total > w->short_cap;
    w->mid_limit = mid_limit();
    /* (what the image allows, not what UTREE_LANE_PASS says at this moment: a workspace sized once -- the whole-file search keeps its
     * lanes' workspaces -- must hold for every batch whichever kernels take it) */
    const int lanes_img = utk_lanes_image_ok(&d->kimg);
    if (max_total > w->mid_limit || (lanes_img && max_len > utk_lanes_max_len(&d->kimg))) {
        w->long_blocks = (uint32_t)d->n_cu * LONG_BLOCKS_PER_CU;
        if (w->long_blocks > n_reads) w->long_blocks = n_reads;
        w->hist = (uint32_t *)(b + off); off = align_up(off + (uint64_t)w->long_blocks * d->hdr.n_labels * 4, 256);
        w->touch = (uint32_t *)(b + off); off = align_up(off + (uint64_t)w->long_blocks * ((d->hdr.n_labels + 31) / 32) * 4, 256);
    }
    /* the lane-per-read pass on a batch of mixed lengths: the reads listed by the lanes they need; long reads in pieces: a tally table
     * per read that can be long, the list of pieces.  (Carved whenever the image takes that pass: whether a batch uses it is decided
     * per batch, the workspace's size must not depend on that.) */
    w->pieces = NULL; w->ltab_rank = w->ltab_cnt = w->lflag = w->long_left = NULL; w->n_long_cap = 0; w->ltally_base = 0;
    w->cls_list = NULL; w->cls_stride = 0; w->n_pieces_cap = 0;
    if (lanes_img && max_len > UTREE_LANES_CAP) {
        w->cls_stride = (n_reads + 63u) & ~63u;
        w->cls_list = (uint32_t *)(b + off); off = align_up(off + (uint64_t)5 * w->cls_stride * 4, 256);
    }
    if (lanes_img && w->long_blocks) {
        uint64_t cap = total_bases / lanes_long_min(d, do_rc) + 1;
        if (cap > n_reads) cap = n_reads;
        const uint64_t piece_windows = 16ull * (UTREE_LANES_CAP - 4 * d->hdr.W + 1);
        const uint64_t n_pieces = total_bases / piece_windows + cap + 1;
        w->n_long_cap = (uint32_t)cap;
        w->n_pieces_cap = n_pieces;
        w->pieces = (uint64_t *)(b + off); off = align_up(off + n_pieces * 8, 256);
        w->ltab_rank = (uint32_t *)(b + off); off = align_up(off + cap * UTREE_LONG_SLOTS * 4, 256);
        w->ltab_cnt = (uint32_t *)(b + off); off = align_up(off + cap * UTREE_LONG_SLOTS * 4, 256);
        w->lflag = (uint32_t *)(b + off); off = align_up(off + cap * 4, 256);
        w->long_left = (uint32_t *)(b + off); off = align_up(off + cap * 4, 256);
        /* the long reads' tally lists at fixed places behind everything the other passes reserve (no reservation traffic) */
        w->ltally_base = (uint64_t)((uint64_t *)(b + off) - w->tally); off = align_up(off + cap * UTREE_LONG_SLOTS * 8, 256);
        /* test hooks: capacities too small on purpose (the areas keep their size): the kernels must report, not overrun */
        { const char *e = getenv("UTREE_TEST_LONG_CAP"); if (e && atoll(e) > 0 && (uint64_t)atoll(e) < w->n_long_cap) w->n_long_cap = (uint32_t)atoll(e); }
        { const char *e = getenv("UTREE_TEST_PIECES_CAP"); if (e && atoll(e) > 0 && (uint64_t)atoll(e) < w->n_pieces_cap) w->n_pieces_cap = (uint64_t)atoll(e); }
    }
    *bytes = (size_t)off;
}

size_t utree_classify_workspace_bytes(const utree_dev *dev, uint32_t n_reads, uint64_t total_bases, uint32_t max_len, int do_rc) {
    if (!dev) return 0;
    utk_workspace w; size_t bytes;
    carve(dev, NULL, n_reads, total_bases, max_len, do_rc, &w, &bytes);
    return bytes;
}

int utree_classify_batch(utree_dev *d, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len,
                         uint32_t n_reads, uint64_t total_bases, uint32_t max_len, int do_rc, utree_result *d_out,
                         void *d_workspace, size_t workspace_bytes, void *stream) {
    int rc = UTREE_OK;
    if (!d || !d_out || (!d_workspace && n_reads)) return UTREE_E_ARG;
    if (!n_reads) return UTREE_OK;
    if (!d_bases || !d_off || !d_len) return UTREE_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    utk_workspace w; size_t need;
    carve(d, d_workspace, n_reads, total_bases, max_len, do_rc, &w, &need);
    if (workspace_bytes < need) return UTREE_E_ARG;
    HIPCHK(hipSetDevice(d->device));
    HIPCHK(hipMemsetAsync(w.cursors, 0, UTREE_CURSOR_BYTES, st));
    if (w.long_blocks) {
        HIPCHK(hipMemsetAsync(w.hist, 0, (size_t)w.long_blocks * d->hdr.n_labels * 4, st));
        HIPCHK(hipMemsetAsync(w.touch, 0, (size_t)w.long_blocks * ((d->hdr.n_labels + 31) / 32) * 4, st));
    }
    /* HIP events on the launch stream bracket the dominant kernel (bench.py's roofline leg) */
    /* (callers on several threads share the handle -- the lanes of search_dev.c --: a call claims its pair of events with one
     * atomic, the events exist since utree_classify_kernel_time switched the bracket on, and a pair counts only once both of its
     * events are recorded) */
    hipEvent_t e0 = NULL, e1 = NULL;
    int tslot = -1;
    if (__atomic_load_n(&d->timing_on, __ATOMIC_ACQUIRE)) {
        tslot = __atomic_fetch_add(&d->n_claimed, 1, __ATOMIC_RELAXED);
        if (tslot >= 0 && tslot < UTREE_MAX_PENDING && 2 * tslot + 1 < d->n_events) { e0 = d->events[2 * tslot]; e1 = d->events[2 * tslot + 1]; }
        else tslot = -1;
    }
    ring_collect(d);
    const int lanes = lanes_enabled() && utk_lanes_image_ok(&d->kimg) && lanes_worth(d);
    if (lanes) {
        /* ---- the lane-per-read pass: a batch of reads of up to 160 bases goes through whole with one lane per read; any other batch is
         * split by the lanes a read needs (one launch per size), its long reads -- beyond sixteen lanes -- go through in pieces, and
         * classify_long_k finishes the few the pieces pass gives up on.  What the pass leaves (several bad bases, more labels than a
         * read's table holds) is on mid_list for the wave-per-read kernel. ---- */
        const int mixed = max_len > UTREE_LANES_CAP;
        d->last_long = w.long_blocks != 0; d->last_mid = 0; d->last_rc = do_rc; d->last_short_cap = w.short_cap;
        d->last_lanes = utk_lanes_segs(&d->kimg, max_len); d->last_mixed = mixed; d->last_pieces = w.long_blocks != 0;
        if (e0 && !w.long_blocks) HIPCHK(hipEventRecord(e0, st));
        if (!mixed) KCHK(utk_classify_lanes(&d->kimg, d_bases, d_off, d_len, n_reads, max_len, do_rc, d_out, &w, d->n_cu, st));
        /* (the classes' launches go one after the other on `stream`: forked onto side streams with events they were 3-5 % slower,
         * profiles/r03/mixed_batches*.json) */
        else KCHK(utk_classify_lanes_mixed(&d->kimg, d_bases, d_off, d_len, n_reads, max_len, do_rc, d_out, &w, d->n_cu, st));
        if (w.long_blocks) {
            if (e0) HIPCHK(hipEventRecord(e0, st));
            HIPCHK(hipMemsetAsync(w.ltab_rank, 0xFF, (size_t)w.n_long_cap * UTREE_LONG_SLOTS * 4, st));
            HIPCHK(hipMemsetAsync(w.ltab_cnt, 0, (size_t)w.n_long_cap * UTREE_LONG_SLOTS * 4, st));
            HIPCHK(hipMemsetAsync(w.lflag, 0, (size_t)w.n_long_cap * 4, st));
            KCHK(utk_classify_long_pieces(&d->kimg, d_bases, d_off, d_len, do_rc, d_out, &w, d->n_cu, st));
            utk_workspace wl = w;
            wl.long_list = w.long_left;
            wl.long_blocks = w.long_blocks >= 8 ? w.long_blocks / 8 : 1;     /* few reads are left: a grid that finds that out quickly */
            KCHK(utk_classify_long(&d->kimg, d_bases, d_off, d_len, do_rc, d_out, &wl, d->n_cu, st));
            if (e0) { HIPCHK(hipEventRecord(e1, st)); d->recorded[tslot] = 1; }
        }
        KCHK(utk_classify_listed(&d->kimg, d_bases, d_off, d_len, n_reads, max_len, do_rc, d_out, &w, d->n_cu, st));
        /* (the bracket of a batch without long reads ends behind the pass over what the lane pass left: on a workload that leaves many reads over
         * that pass is part of the dominant work) */
        if (e0 && !w.long_blocks) { HIPCHK(hipEventRecord(e1, st)); d->recorded[tslot] = 1; }
    } else {
        /* ---- the wave-per-read kernels: images the lane-per-read pass does not take (k = 64 with u32 labels, many irregular bins, a
         * non-monotone bin table), and batches it is not worth running on ---- */
        /* the bracket goes around the batch's dominant kernel: the long-read kernel when the batch has long reads,
         * else the mid-length pass when it has mid-length reads, else the 150-bp-class kernel */
        const int dominant = w.long_blocks ? 2 : (w.mid_reads ? 1 : 0);
        d->last_long = dominant == 2; d->last_mid = dominant == 1; d->last_rc = do_rc; d->last_short_cap = w.short_cap;
        d->last_lanes = 0; d->last_mixed = 0; d->last_pieces = 0;
        if (w.mid_reads) KCHK(utk_route(d_len, n_reads, do_rc, &w, st));
        if (e0 && dominant == 0) HIPCHK(hipEventRecord(e0, st));
        KCHK(utk_classify_short(&d->kimg, d_bases, d_off, d_len, n_reads, do_rc, d_out, &w, d->n_cu, st));
        if (e0 && dominant == 0) { HIPCHK(hipEventRecord(e1, st)); d->recorded[tslot] = 1; }
        if (w.mid_reads) {
            if (e0 && dominant == 1) HIPCHK(hipEventRecord(e0, st));
            KCHK(utk_classify_mid(&d->kimg, d_bases, d_off, d_len, n_reads, do_rc, d_out, &w, d->n_cu, st));
            if (e0 && dominant == 1) { HIPCHK(hipEventRecord(e1, st)); d->recorded[tslot] = 1; }
        }
        if (w.long_blocks) {
            if (e0) HIPCHK(hipEventRecord(e0, st));
            KCHK(utk_classify_long(&d->kimg, d_bases, d_off, d_len, do_rc, d_out, &w, d->n_cu, st));
            if (e0) { HIPCHK(hipEventRecord(e1, st)); d->recorded[tslot] = 1; }
        }
    }
    KCHK(utk_vote(&d->kimg, d_out, &w, n_reads, st));
    /* the reads the lane pass left, and the batch's error word, come back behind the kernels without a wait */
    if (ring_post(d, &w, lanes ? n_reads : 0, st)) { (void)hipGetLastError(); }
    /* an earlier batch's error that has arrived meanwhile is this call's to report too (utree_classify_poll after the stream has
     * drained reports this batch's own) */
    { ring_lock(d); const unsigned long long pe = d->dev_error; ring_unlock(d);
      if (pe) { rc = UTREE_E_DEVICE; snprintf(g_hip_msg, sizeof g_hip_msg, "an earlier batch's kernels reported error %llu: utree_classify_poll returns (and clears) it", pe); } }
#ifdef UTREE_LANES_TIMERS
    { extern void utk_lanes_phase_dump(void); static int lcalls; if (++lcalls == 6) utk_lanes_phase_dump(); }
#endif
#ifdef UTREE_PHASE_TIMERS
    { extern void utk_phase_dump(void); static int calls; if (++calls == 12) utk_phase_dump(); }
#endif
fail:
    return rc;
}

int utree_lookup_words(utree_dev *d, const uint64_t *d_hi, const uint64_t *d_lo, uint64_t n, uint32_t *d_ix, void *stream) {
    int rc = UTREE_OK;
    if (!d || !d_lo || !d_ix) return UTREE_E_ARG;
    if (d->hdr.W == 16 && !d_hi) return UTREE_E_ARG;
    HIPCHK(hipSetDevice(d->device));
    KCHK(utk_lookup(&d->kimg, d_hi, d_lo, n, d_ix, (hipStream_t)stream));
fail:
    return rc;
}

const char *utree_classify_kernel_name(const utree_dev *dc) {
    if (!dc) return "";
    utree_dev *d = (utree_dev *)dc;                       /* the signature string lives in the handle */
    if (d->last_lanes && d->last_pieces) {
        snprintf(d->kernel_sig, sizeof d->kernel_sig, "classify_lanes_k<%u, %u, 16, %s, 2, %u, %s>", d->hdr.W, d->hdr.I, d->kimg.irr_n ? "true" : "false", d->hdr.bucket_words / 8,
                 utk_lanes_both_strands(&d->kimg, d->last_rc) ? "true" : "false");
        return d->kernel_sig;
    }
    if (d->last_lanes && d->last_mixed) {
        /* a batch of mixed lengths: one launch whose wavefronts work through the lanes-per-read classes */
        snprintf(d->kernel_sig, sizeof d->kernel_sig, "classify_lanes_mixed_k<%u, %u, %s, %u, %s>", d->hdr.W, d->hdr.I, d->kimg.irr_n ? "true" : "false",
                 d->hdr.bucket_words / 8, utk_lanes_both_strands(&d->kimg, d->last_rc) ? "true" : "false");
        return d->kernel_sig;
    }
    if (d->last_lanes) {
        snprintf(d->kernel_sig, sizeof d->kernel_sig, "classify_lanes_k<%u, %u, 1, %s, 0, %u, %s>", d->hdr.W, d->hdr.I, d->kimg.irr_n ? "true" : "false",
                 d->hdr.bucket_words / 8, utk_lanes_both_strands(&d->kimg, d->last_rc) ? "true" : "false");
        return d->kernel_sig;
    }
    return d->last_long ? utk_classify_long_name(&d->kimg, d->kernel_sig, sizeof d->kernel_sig)
                        : utk_classify_short_name(&d->kimg, d->last_short_cap ? d->last_short_cap : UTREE_SHORT_CAP, d->last_mid, d->last_rc,
                                                  d->kernel_sig, sizeof d->kernel_sig);
}

/* measurement aid (bench.py's byte model): counts[0..4] = reads looked at (up to 640 staged bases), valid windows, distinct
 * 64-byte buckets per read summed, distinct 128-byte lines per read summed, distinct buckets with an overflow descriptor */
int utree_model_counts(utree_dev *d, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len, uint32_t n_reads,
                       int do_rc, uint64_t *h_counts5, void *stream) {
    int rc = UTREE_OK;
    unsigned long long *dc = NULL;
    if (!d || !d_bases || !d_off || !d_len || !h_counts5) return UTREE_E_ARG;
    HIPCHK(hipSetDevice(d->device));
    HIPCHK(hipMalloc((void **)&dc, 64));
    HIPCHK(hipMemsetAsync(dc, 0, 64, (hipStream_t)stream));
    KCHK(utk_model_counts(&d->kimg, d_bases, d_off, d_len, n_reads, do_rc, dc, stream));
    HIPCHK(hipMemcpyAsync(h_counts5, dc, 40, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
fail:
    if (dc) hipFree(dc);
    return rc;
}

int utree_classify_kernel_time(utree_dev *d, int reset, double *ms_total, uint64_t *launches) {
    int rc = UTREE_OK;
    if (!d) return UTREE_E_ARG;
    HIPCHK(hipSetDevice(d->device));
    /* (called between batches: no utree_classify_batch is running on this handle) */
    int claimed = __atomic_load_n(&d->n_claimed, __ATOMIC_RELAXED);
    if (claimed > UTREE_MAX_PENDING) claimed = UTREE_MAX_PENDING;
    for (int i = 0; i < claimed; ++i) {
        float ms = 0.f;
        if (!d->recorded[i]) continue;                          /* a call that failed between its two events */
        d->recorded[i] = 0;
        HIPCHK(hipEventSynchronize(d->events[2 * i + 1]));
        HIPCHK(hipEventElapsedTime(&ms, d->events[2 * i], d->events[2 * i + 1]));
        d->ms_total += ms; d->launches++;
    }
    while (d->n_events < 2 * UTREE_MAX_PENDING) { HIPCHK(hipEventCreate((hipEvent_t *)&d->events[d->n_events])); d->n_events++; }
    __atomic_store_n(&d->n_claimed, 0, __ATOMIC_RELAXED);
    if (ms_total) *ms_total = d->ms_total;
    if (launches) *launches = d->launches;
    if (reset) { d->ms_total = 0; d->launches = 0; }
    __atomic_store_n(&d->timing_on, 1, __ATOMIC_RELEASE);
fail:
    return rc;
}
