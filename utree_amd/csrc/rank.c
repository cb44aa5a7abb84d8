/* rank.c -- host side of the rank-specific search (`xtree-search`: itree.c -D SEARCH, 969-1007) behind the
 * C-ABI: workspace layout, the per-batch kernel sequence (rank_kernels.hip), the carried vote state, and the
 * output lines (itree.c:1002).  SURVEY.md §8(f) rank 1.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ctr_host.h"
#include "dev_image.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { utree_dev_set_hip_error((int)e_, #x); rc = UTREE_E_HIP; goto fail; } } while (0)
#define KCHK(x) do { int e_ = (x); if (e_ != 0) { utree_dev_set_hip_error(e_, #x); rc = UTREE_E_HIP; goto fail; } } while (0)

#define RANK_MAX_LEN (1u << 30)                 /* the reference's lines end at 16 MiB (itree.c:836) */
#define RANK_HIST_BYTES ((uint64_t)512 << 20)   /* HBM for the long vote's per-wavefront label counters */

static uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

void utree_rank_params_default(utree_rank_params *p) {
    if (!p) return;
    p->slack = 2; p->sparsity = 4; p->tolerance = 2;                  /* itree.c:952-960 */
}

static int params_ok(const utree_dev *d, const utree_rank_params *p) {
    /* PACKSIZE / SPARSITY must be at least one window: with SPARSITY > PACKSIZE the reference never advances */
    return p && p->sparsity >= 1 && p->sparsity <= 4 * d->hdr.W && p->slack <= (1u << 20) && p->tolerance <= (1u << 30);
}

/* most hits one read of this batch can keep */
static uint64_t max_hits(const utree_dev *d, uint32_t max_len, int do_rc, uint32_t step) {
    uint64_t K = 4ull * d->hdr.W, total = do_rc ? 2 * (uint64_t)max_len + 1 : max_len;
    if (total < K) return 0;
    return (total - K + 1 + step - 1) / step;
}

static void carve(const utree_dev *d, void *ws, uint32_t n_reads, uint64_t total_bases, uint32_t max_len, int do_rc,
                  const utree_rank_params *p, utk_rank_ws *w, size_t *bytes) {
    uint64_t off = 0;
    char *b = (char *)ws;
    memset(w, 0, sizeof *w);
    w->step = 4 * d->hdr.W / p->sparsity;
    w->slack = p->slack; w->tolerance = p->tolerance;
    w->cursors = (unsigned long long *)(b + off); off = align_up(off + 512, 256);
    w->nh = (uint32_t *)(b + off); off = align_up(off + ((uint64_t)n_reads + 64) * 4, 256);
    w->hoff = (uint64_t *)(b + off); off = align_up(off + ((uint64_t)n_reads + 64) * 8, 256);
    uint32_t n = n_reads;
    for (int l = 0; l < 3; ++l) {
        n = (n + 63) / 64;
        w->nlvl[l] = n;
        w->lvl[l] = (uint32_t *)(b + off); off = align_up(off + ((uint64_t)n + 64) * 4, 256);
    }
    /* a read reserves ceil(windows / step) entries; a chunk refill abandons < 1/16 of a chunk (rank_kernels.hip) */
    uint64_t windows = (do_rc ? 2 : 1) * total_bases + (uint64_t)n_reads;
    w->hits_cap = (windows / w->step + (uint64_t)n_reads) * 9 / 8 + (uint64_t)d->n_cu * 32 * UTREE_TALLY_CHUNK + 4096;
    w->hits = (uint32_t *)(b + off); off = align_up(off + w->hits_cap * 4, 256);
    if (max_hits(d, max_len, do_rc, w->step) + 1 > 64) {                  /* some read may not fit one entry per lane */
        uint64_t per = (uint64_t)d->hdr.n_labels * 4, waves = RANK_HIST_BYTES / (per ? per : 1);
        if (waves > (uint64_t)d->n_cu * 8) waves = (uint64_t)d->n_cu * 8;
        if (waves > n_reads) waves = n_reads;
        if (waves < 1) waves = 1;
        w->hist_waves = (uint32_t)waves;
        w->hist = (uint32_t *)(b + off); off = align_up(off + waves * per, 256);
    }
    *bytes = (size_t)off;
}

size_t utree_rank_workspace_bytes(const utree_dev *dev, uint32_t n_reads, uint64_t total_bases, uint32_t max_len, int do_rc,
                                  const utree_rank_params *params) {
    if (!dev || !params_ok(dev, params)) return 0;
    utk_rank_ws w; size_t bytes;
    carve(dev, NULL, n_reads, total_bases, max_len, do_rc, params, &w, &bytes);
    return bytes;
}

/* the carried array must reach index (hits of the longest read); it only ever grows */
static int state_reserve(utree_dev *d, uint64_t need, hipStream_t st) {
    int rc = UTREE_OK;
    if (need <= d->rank_state_cap) return UTREE_OK;
    uint64_t cap = d->rank_state_cap ? d->rank_state_cap : 4096;
    while (cap < need) cap *= 2;
    void *nu = NULL;
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMalloc(&nu, cap * 4));
    /* stream-ordered on the caller's stream: its kernels read the array next, and a non-blocking stream has no
     * implicit ordering with the null stream */
    HIPCHK(hipMemsetAsync(nu, 0, cap * 4, st));                         /* untouched entries read as label 0 */
    if (d->rank_state) {
        HIPCHK(hipMemcpyAsync(nu, d->rank_state, d->rank_state_cap * 4, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipStreamSynchronize(st));                               /* the old array is freed next */
        HIPCHK(hipFree(d->rank_state));
    }
    d->rank_state = nu; d->rank_state_cap = cap;
    return UTREE_OK;
fail:
    if (nu) hipFree(nu);
    return rc;
}

int utree_rank_reset(utree_dev *d) {
    int rc = UTREE_OK;
    if (!d) return UTREE_E_ARG;
    HIPCHK(hipSetDevice(d->device));
    if (d->rank_state) {
        HIPCHK(hipMemset(d->rank_state, 0, d->rank_state_cap * 4));
        HIPCHK(hipDeviceSynchronize());                                 /* the batches' streams are non-blocking: order by completion */
    }
fail:
    return rc;
}

int utree_rank_batch(utree_dev *d, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len, uint32_t n_reads,
                     uint64_t total_bases, uint32_t max_len, int do_rc, const utree_rank_params *params,
                     utree_result *d_out, void *d_workspace, size_t workspace_bytes, void *stream) {
    int rc = UTREE_OK;
    if (!d || !d_out || (!d_workspace && n_reads) || !params_ok(d, params) || max_len > RANK_MAX_LEN) return UTREE_E_ARG;
    if (d->hdr.W == 4) return UTREE_E_UNSUPPORTED;      /* PACKSIZE=16 trees: the GG search only (the rank-specific kernels exist for k = 32 and 64) */
    if (!n_reads) return UTREE_OK;
    if (!d_bases || !d_off || !d_len) return UTREE_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    utk_rank_ws w; size_t need;
    carve(d, d_workspace, n_reads, total_bases, max_len, do_rc, params, &w, &need);
    if (workspace_bytes < need) return UTREE_E_ARG;
    HIPCHK(hipSetDevice(d->device));
    if ((rc = state_reserve(d, max_hits(d, max_len, do_rc, w.step) + 2, st))) return rc;
    w.state = (uint32_t *)d->rank_state; w.state_cap = (uint32_t)d->rank_state_cap;
    HIPCHK(hipMemsetAsync(w.cursors, 0, 512, st));
    if (w.hist_waves) HIPCHK(hipMemsetAsync(w.hist, 0, (size_t)w.hist_waves * d->hdr.n_labels * 4, st));
    KCHK(utk_rank_hits(&d->kimg, d_bases, d_off, d_len, n_reads, do_rc, &w, d->n_cu, st));
    KCHK(utk_rank_levels(&w, n_reads, st));
    KCHK(utk_rank_vote(&d->kimg, d_out, n_reads, &w, d->n_cu, st));
    KCHK(utk_rank_state(n_reads, &w, st));
fail:
    return rc;
}

/* "%s\t%s\t%f\t%d\n" (itree.c:1002) for the reads the reference prints: found > 0 and cut == -2 */
size_t utree_format_rank_records(const utree_ctr *ctr, const uint8_t *h_buf, const uint64_t *name_off, const uint32_t *name_len,
                                 const utree_result *res, size_t n, char *out, size_t cap, uint64_t *good_finds) {
    char *o = out;
    uint64_t good = 0;
    for (size_t i = 0; i < n; ++i) {
        const utree_result *r = &res[i];
        if (!r->found || r->cut != -2) continue;                        /* itree.c:980, 1000 */
        if (r->label >= ctr->info.n_labels || !r->sl) return (size_t)-1;
        size_t ll = ctr->label_len[r->label];
        if ((size_t)(o - out) + name_len[i] + ll + 64 > cap) return (size_t)-1;
        memcpy(o, h_buf + name_off[i], name_len[i]); o += name_len[i];
        *o++ = '\t';
        memcpy(o, ctr->labels[r->label], ll); o += ll;
        o += sprintf(o, "\t%f\t%d\n", (double)1 - (double)(int)r->ol / (int)r->sl, (int)r->sl);
        ++good;
    }
    if (good_finds) *good_finds += good;
    return (size_t)(o - out);
}
