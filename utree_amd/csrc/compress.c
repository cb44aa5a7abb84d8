/* compress.c -- `.ubt` -> `.ctr`: replaces XT_cmp32 (itree.c:1234-1315).
 *
 * The node dump is streamed through the GPU in chunks: one kernel pass drops the 3 prefix bytes of every
 * record (itree.c:1306-1309) and finds, per 24-bit prefix, the first node index that is not 0 -- which is
 * exactly what the reference's `if (!BinIx[v]) BinIx[v] = i` computes (itree.c:1282-1286), first-bin quirk
 * included.  The 2^24+1-entry table fix-up (itree.c:1287-1289) and the label tail are done on the host.
 * The output file is byte-identical to the reference's xtree-compress.
 */
#define _FILE_OFFSET_BITS 64
#define _GNU_SOURCE
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "utree_internal.h"

#define HIPC(x) do { if ((x) != hipSuccess) { rc = UTREE_E_HIP; goto done; } } while (0)

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

/* label tail as readSamplesFPdelim(…, delim = 0) leaves it (itree.c:1154-1211): labels in first-seen order;
 * a line's count goes to SampCnts[sampIX] -- the NEWEST label so far, also when the line repeats an older label */
static int rewrite_labels(const char *text, size_t len, char **out, size_t *out_len, uint64_t *n_labels, uint64_t *total) {
    size_t lines = 1;
    for (size_t i = 0; i < len; ++i) lines += text[i] == '\n';
    char *buf = (char *)malloc(len + 1);
    char **lab = (char **)malloc(sizeof(char *) * (lines + 1));
    uint64_t *cnt = (uint64_t *)calloc(lines + 1, sizeof(uint64_t));
    size_t cap = 64;
    while (cap < 2 * lines) cap <<= 1;
    uint32_t *table = (uint32_t *)malloc(cap * sizeof(uint32_t));
    if (!buf || !lab || !cnt || !table) { free(buf); free(lab); free(cnt); free(table); return UTREE_E_NOMEM; }
    memcpy(buf, text, len); buf[len] = 0;
    memset(table, 0xFF, cap * sizeof(uint32_t));
    uint64_t n = 0;
    char *p = buf, *end = buf + len;
    while (p < end) {
        char *nl = (char *)memchr(p, '\n', (size_t)(end - p));
        char *stop = nl ? nl : end;
        char *tab = (char *)memchr(p, '\t', (size_t)(stop - p));
        char *lend = tab ? tab : stop;
        const char *num = tab ? tab + 1 : stop;
        char saved = *stop; *stop = 0;                           /* atol reads up to the line end */
        uint64_t c = (uint64_t)atol(num);
        *stop = saved;
        *lend = 0;
        size_t L = strlen(p);
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < L; ++i) h = (h ^ (unsigned char)p[i]) * 1099511628211ull;
        h &= cap - 1;
        for (;;) {
            uint32_t v = table[h];
            if (v == 0xFFFFFFFFu) { table[h] = (uint32_t)n; lab[n++] = p; break; }
            if (!strcmp(lab[v], p)) break;
            h = (h + 1) & (cap - 1);
        }
        if (n) cnt[n - 1] = c;                                   /* SampCnts[sampIX] = atol(src+1), itree.c:1206 */
        p = stop + 1;
    }
    size_t need = 0;
    for (uint64_t i = 0; i < n; ++i) need += strlen(lab[i]) + 24;
    char *o = (char *)malloc(need + 1);
    if (!o) { free(buf); free(lab); free(cnt); free(table); return UTREE_E_NOMEM; }
    size_t w = 0;
    uint64_t tot = 0;
    for (uint64_t i = 0; i < n; ++i) { tot += cnt[i]; w += (size_t)sprintf(o + w, "%s\t%llu\n", lab[i], (unsigned long long)cnt[i]); }   /* itree.c:1313 */
    *out = o; *out_len = w; *n_labels = n; *total = tot;
    free(buf); free(lab); free(cnt); free(table);
    return UTREE_OK;
}

int utree_compress_file(const char *ubt_path, const char *ctr_path, int device, utree_compress_stats *stats) {
    if (!ubt_path || !ctr_path) return UTREE_E_ARG;
    int rc = UTREE_OK, fd = -1, fo = -1;
    double t0 = now_s();
    void *h_in[2] = {NULL, NULL}, *h_out[2] = {NULL, NULL}, *d_in[2] = {NULL, NULL}, *d_out[2] = {NULL, NULL};
    unsigned long long *d_first = NULL, *h_first = NULL;
    hipStream_t st[2] = {NULL, NULL};
    char *labels = NULL, *tail = NULL;
    uint8_t *binix = NULL;
    fd = open(ubt_path, O_RDONLY);
    if (fd < 0) return UTREE_E_IO;                                            /* "Invalid input filename", itree.c:1236 */
    uint64_t meta[4] = {0, 0, 0, 0};
    if (pread(fd, meta, 32, 0) != 32 || !meta[3]) { close(fd); return UTREE_E_FORMAT; }      /* itree.c:1239 */
    const uint64_t W = meta[0], I = meta[2], N = meta[3];
    if (meta[1] != 0 || !(W == 4 || W == 8 || W == 16) || !(I == 2 || I == 4)) { close(fd); return UTREE_E_UNSUPPORTED; }   /* PACKSIZE 16, 32, 64 */
    const size_t DR = (size_t)(W + I), SZ = (size_t)(W + I - 3);
    off_t fsize = lseek(fd, 0, SEEK_END);
    if ((uint64_t)fsize < 32 + N * DR) { close(fd); return UTREE_E_FORMAT; }
    const int ixsz = N < 0xFFFFFFFFull ? 4 : 8;                               /* itree.c:1303 */
    const uint64_t rec_off = 32 + (uint64_t)UTREE_NUMBINS * ixsz;
    fo = open(ctr_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fo < 0) { close(fd); return UTREE_E_IO; }                             /* "Invalid output filename", itree.c:1299 */
    if (hipSetDevice(device) != hipSuccess) { rc = UTREE_E_HIP; goto done; }
    const size_t chunk = ((size_t)32 << 20) / DR;                             /* records per chunk */
    for (int i = 0; i < 2; ++i) {
        HIPC(hipHostMalloc(&h_in[i], chunk * DR, hipHostMallocDefault));
        HIPC(hipHostMalloc(&h_out[i], chunk * SZ, hipHostMallocDefault));
        HIPC(hipMalloc(&d_in[i], chunk * DR));
        HIPC(hipMalloc(&d_out[i], chunk * SZ));
        HIPC(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    }
    HIPC(hipMalloc((void **)&d_first, (size_t)UTREE_NUMBINS * 8));
    HIPC(hipMemset(d_first, 0xFF, (size_t)UTREE_NUMBINS * 8));
    HIPC(hipDeviceSynchronize());                                             /* the chunk streams are non-blocking: no implicit order with the null stream */
    {
        uint64_t done = 0, pend_first[2] = {0, 0}, pend_cnt[2] = {0, 0};
        for (int slot = 0; done < N || pend_cnt[0] || pend_cnt[1]; slot ^= 1) {
            if (pend_cnt[slot]) {                                             /* drain what this slot produced last time */
                HIPC(hipStreamSynchronize(st[slot]));
                size_t bytes = (size_t)pend_cnt[slot] * SZ, w = 0;
                while (w < bytes) {
                    ssize_t k = pwrite(fo, (char *)h_out[slot] + w, bytes - w, (off_t)(rec_off + pend_first[slot] * SZ + w));
                    if (k <= 0) { rc = UTREE_E_IO; goto done; }
                    w += (size_t)k;
                }
                pend_cnt[slot] = 0;
            }
            if (done >= N) continue;
            uint64_t cnt = N - done < chunk ? N - done : chunk;
            size_t bytes = (size_t)cnt * DR, got = 0;
            while (got < bytes) {
                ssize_t k = pread(fd, (char *)h_in[slot] + got, bytes - got, (off_t)(32 + done * DR + got));
                if (k <= 0) { rc = UTREE_E_FORMAT; goto done; }
                got += (size_t)k;
            }
            HIPC(hipMemcpyAsync(d_in[slot], h_in[slot], bytes, hipMemcpyHostToDevice, st[slot]));
            if (utk_compress_chunk((uint32_t)W, (uint32_t)I, d_in[slot], done, cnt, d_first, d_out[slot], st[slot])) { rc = UTREE_E_HIP; goto done; }
            HIPC(hipMemcpyAsync(h_out[slot], d_out[slot], (size_t)cnt * SZ, hipMemcpyDeviceToHost, st[slot]));
            pend_first[slot] = done; pend_cnt[slot] = cnt;
            done += cnt;
        }
    }
    /* bin table (itree.c:1281-1289) */
    h_first = (unsigned long long *)malloc((size_t)UTREE_NUMBINS * 8);
    binix = (uint8_t *)malloc((size_t)UTREE_NUMBINS * ixsz);
    if (!h_first || !binix) { rc = UTREE_E_NOMEM; goto done; }
    HIPC(hipMemcpy(h_first, d_first, (size_t)UTREE_NUMBINS * 8, hipMemcpyDeviceToHost));
    {
        uint64_t *B = (uint64_t *)h_first;
        for (size_t i = 0; i < UTREE_NUMBINS; ++i) if (B[i] == ~0ull) B[i] = 0;      /* never set: calloc'd 0 */
        B[UTREE_NUMBINS - 1] = N;                                                  /* itree.c:1287 */
        size_t u = 0; for (; !B[u]; ++u); B[u] = 0;                                /* itree.c:1288 */
        for (size_t i = UTREE_NUMBINS - 2; i > u; --i) if (!B[i]) B[i] = B[i + 1]; /* itree.c:1289 */
        for (size_t i = 0; i < UTREE_NUMBINS; ++i) {
            if (ixsz == 4) { uint32_t v = (uint32_t)B[i]; memcpy(binix + 4 * i, &v, 4); } else memcpy(binix + 8 * i, &B[i], 8);
        }
    }
    /* labels (itree.c:1270, 1310-1313) */
    {
        size_t tlen = (size_t)((uint64_t)fsize - (32 + N * DR));
        tail = (char *)malloc(tlen + 1);
        if (!tail) { rc = UTREE_E_NOMEM; goto done; }
        if (tlen && pread(fd, tail, tlen, (off_t)(32 + N * DR)) != (ssize_t)tlen) { rc = UTREE_E_IO; goto done; }
        size_t llen = 0; uint64_t nl = 0, tot = 0;
        rc = rewrite_labels(tail, tlen, &labels, &llen, &nl, &tot);
        if (rc) goto done;
        if (pwrite(fo, meta, 32, 0) != 32) { rc = UTREE_E_IO; goto done; }             /* itree.c:1301 */
        if (pwrite(fo, binix, (size_t)UTREE_NUMBINS * ixsz, 32) != (ssize_t)((size_t)UTREE_NUMBINS * ixsz)) { rc = UTREE_E_IO; goto done; }
        if (llen && pwrite(fo, labels, llen, (off_t)(rec_off + N * SZ)) != (ssize_t)llen) { rc = UTREE_E_IO; goto done; }
        if (stats) { stats->n_nodes = N; stats->n_labels = nl; stats->label_count_total = tot; stats->W = (uint32_t)W; stats->I = (uint32_t)I; }
    }
done:
    if (stats) stats->seconds = now_s() - t0;
    if (fd >= 0) close(fd);
    if (fo >= 0) close(fo);
    for (int i = 0; i < 2; ++i) {
        if (h_in[i]) hipHostFree(h_in[i]);
        if (h_out[i]) hipHostFree(h_out[i]);
        if (d_in[i]) hipFree(d_in[i]);
        if (d_out[i]) hipFree(d_out[i]);
        if (st[i]) hipStreamDestroy(st[i]);
    }
    if (d_first) hipFree(d_first);
    free(h_first); free(binix); free(tail); free(labels);
    return rc;
}
