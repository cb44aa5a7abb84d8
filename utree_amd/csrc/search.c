/* search.c -- whole-file search: XT_doSearch32's GG branch (itree.c:833-1108) over one or more device
 * images.  Host orchestration in C: read the FASTA in large chunks into pinned memory, frame the reads
 * (fasta.c), shard them contiguously over the GPUs, copy each shard's byte span to HBM as it stands,
 * run the batch kernels, copy the 24-byte results back, format and write the lines in input order
 * (= what the reference writes with one thread; with more threads it writes a permutation).
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
#ifdef _OPENMP
#include <omp.h>
#endif
#include "ctr_host.h"
#include "dev_image.h"

#define CHUNK_BYTES ((size_t)96 << 20)        /* must hold two maximal (16 MiB) lines                    */
#define MAX_READS_PER_BATCH ((size_t)4 << 20)
#define LINELEN_MAX 16777216u                 /* itree.c:836 */

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
    utree_dev *dev;
    hipStream_t stream;
    uint8_t *d_buf; uint64_t *d_off; uint32_t *d_len; utree_result *d_out; void *d_ws; size_t ws_bytes;
    hipEvent_t k0, k1;
    size_t first, count;                       /* shard of the current batch                              */
} gpu_ctx;

static void free_ctx(gpu_ctx *g) {
    if (!g->dev) return;
    hipSetDevice(g->dev->device);
    if (g->d_buf) hipFree(g->d_buf);
    if (g->d_off) hipFree(g->d_off);
    if (g->d_len) hipFree(g->d_len);
    if (g->d_out) hipFree(g->d_out);
    if (g->d_ws) hipFree(g->d_ws);
    if (g->k0) hipEventDestroy(g->k0);
    if (g->k1) hipEventDestroy(g->k1);
    if (g->stream) hipStreamDestroy(g->stream);
}

#define HIPOK(x) do { if ((x) != hipSuccess) { rc = UTREE_E_HIP; goto done; } } while (0)

int utree_search_file(const utree_ctr *ctr, utree_dev **devs, int n_dev, const char *fasta_path, const char *out_path,
                      int do_rc, int host_threads, utree_search_stats *stats) {
    if (!ctr || !devs || n_dev < 1 || !fasta_path || !out_path) return UTREE_E_ARG;
    int rc = UTREE_OK;
    double t_start = now_s(), t_kernels = 0;
    utree_search_stats st;
    memset(&st, 0, sizeof st);
    int fd = open(fasta_path, O_RDONLY);
    FILE *fo = fopen(out_path, "wb");
    if (fd < 0 || !fo) { if (fd >= 0) close(fd); if (fo) fclose(fo); return UTREE_E_IO; }   /* itree.c:835 */
#ifdef _OPENMP
    if (host_threads <= 0) host_threads = omp_get_max_threads();
#else
    host_threads = 1;
#endif
    uint8_t *h_buf = NULL;
    uint64_t *seq_off = NULL, *name_off = NULL, *rel_off = NULL;
    uint32_t *seq_len = NULL, *name_len = NULL;
    utree_result *h_res = NULL;
    char **fmt_buf = NULL; size_t *fmt_cap = NULL, *fmt_len = NULL;
    gpu_ctx *G = (gpu_ctx *)calloc((size_t)n_dev, sizeof(gpu_ctx));
    if (!G) { rc = UTREE_E_NOMEM; goto done; }
    HIPOK(hipHostMalloc((void **)&h_buf, CHUNK_BYTES + 64, hipHostMallocDefault));
    seq_off = (uint64_t *)malloc(MAX_READS_PER_BATCH * 8); name_off = (uint64_t *)malloc(MAX_READS_PER_BATCH * 8);
    rel_off = (uint64_t *)malloc(MAX_READS_PER_BATCH * 8);
    seq_len = (uint32_t *)malloc(MAX_READS_PER_BATCH * 4); name_len = (uint32_t *)malloc(MAX_READS_PER_BATCH * 4);
    HIPOK(hipHostMalloc((void **)&h_res, MAX_READS_PER_BATCH * sizeof(utree_result), hipHostMallocDefault));
    fmt_buf = (char **)calloc((size_t)host_threads, sizeof(char *));
    fmt_cap = (size_t *)calloc((size_t)host_threads, sizeof(size_t));
    fmt_len = (size_t *)calloc((size_t)host_threads, sizeof(size_t));
    if (!seq_off || !name_off || !rel_off || !seq_len || !name_len || !fmt_buf || !fmt_cap || !fmt_len) { rc = UTREE_E_NOMEM; goto done; }
    for (int g = 0; g < n_dev; ++g) {
        G[g].dev = devs[g];
        HIPOK(hipSetDevice(devs[g]->device));
        HIPOK(hipStreamCreateWithFlags(&G[g].stream, hipStreamNonBlocking));
        HIPOK(hipEventCreate(&G[g].k0)); HIPOK(hipEventCreate(&G[g].k1));
        HIPOK(hipMalloc((void **)&G[g].d_buf, CHUNK_BYTES + 64));
        HIPOK(hipMalloc((void **)&G[g].d_off, MAX_READS_PER_BATCH * 8));
        HIPOK(hipMalloc((void **)&G[g].d_len, MAX_READS_PER_BATCH * 4));
        HIPOK(hipMalloc((void **)&G[g].d_out, MAX_READS_PER_BATCH * sizeof(utree_result)));
        G[g].ws_bytes = utree_classify_workspace_bytes(devs[g], (uint32_t)MAX_READS_PER_BATCH, CHUNK_BYTES, LINELEN_MAX, do_rc);
        HIPOK(hipMalloc(&G[g].d_ws, G[g].ws_bytes));
    }
    uint32_t max_label = 0;
    for (uint32_t i = 0; i < ctr->info.n_labels; ++i) if (ctr->label_len[i] > max_label) max_label = ctr->label_len[i];

    size_t have = 0;               /* bytes in h_buf */
    int eof = 0;
    uint64_t next_progress = 1048576;
    while (!eof || have) {
        while (!eof && have < CHUNK_BYTES) {
            ssize_t r = read(fd, h_buf + have, CHUNK_BYTES - have);
            if (r < 0) { rc = UTREE_E_IO; goto done; }
            if (r == 0) { eof = 1; break; }
            have += (size_t)r;
        }
        if (!have) break;
        size_t nr = 0, used = 0;
        utree_fasta_error ferr;
        int frc = utree_fasta_frame(h_buf, have, eof, MAX_READS_PER_BATCH, seq_off, seq_len, name_off, name_len, &nr, &used, &ferr);
        if (frc != UTREE_OK && frc != UTREE_E_FASTA) { rc = frc; goto done; }
        if (!nr && frc == UTREE_OK && used == 0) {
            if (eof) break;
            rc = UTREE_E_FASTA; ferr.code = 5; st.fasta_error = ferr; goto done;   /* a line pair larger than the chunk */
        }
        /* shard contiguously over the devices and launch */
        double tk0 = now_s();
        size_t per = (nr + (size_t)n_dev - 1) / (size_t)n_dev;
        for (int g = 0; g < n_dev && nr; ++g) {
            gpu_ctx *c = &G[g];
            c->first = (size_t)g * per; if (c->first > nr) c->first = nr;
            c->count = c->first + per <= nr ? per : nr - c->first;
            if (!c->count) continue;
            size_t lo = (size_t)seq_off[c->first], last = c->first + c->count - 1;
            size_t hi = (size_t)seq_off[last] + seq_len[last];
            uint64_t total = 0; uint32_t mx = 0;
            for (size_t i = c->first; i <= last; ++i) { rel_off[i] = seq_off[i] - lo; total += seq_len[i]; if (seq_len[i] > mx) mx = seq_len[i]; }
            HIPOK(hipSetDevice(c->dev->device));
            HIPOK(hipMemcpyAsync(c->d_buf, h_buf + lo, hi - lo, hipMemcpyHostToDevice, c->stream));
            HIPOK(hipMemcpyAsync(c->d_off, rel_off + c->first, c->count * 8, hipMemcpyHostToDevice, c->stream));
            HIPOK(hipMemcpyAsync(c->d_len, seq_len + c->first, c->count * 4, hipMemcpyHostToDevice, c->stream));
            int e = utree_classify_batch(c->dev, c->d_buf, c->d_off, c->d_len, (uint32_t)c->count, total, mx, do_rc, c->d_out,
                                         c->d_ws, c->ws_bytes, c->stream);
            if (e) { rc = e; goto done; }
            HIPOK(hipMemcpyAsync(h_res + c->first, c->d_out, c->count * sizeof(utree_result), hipMemcpyDeviceToHost, c->stream));
        }
        for (int g = 0; g < n_dev && nr; ++g) {
            if (!G[g].count) continue;
            HIPOK(hipSetDevice(G[g].dev->device));
            HIPOK(hipStreamSynchronize(G[g].stream));
        }
        t_kernels += now_s() - tk0;
        /* format in parallel, write in input order (itree.c:1032, 1040, 1096) */
        int T = host_threads;
        if ((size_t)T > nr) T = nr ? (int)nr : 1;
        int fmt_fail = 0;
#pragma omp parallel for num_threads(T) schedule(static, 1)
        for (int t = 0; t < T; ++t) {
            size_t a = nr * (size_t)t / (size_t)T, b = nr * (size_t)(t + 1) / (size_t)T;
            size_t need = 0;
            for (size_t i = a; i < b; ++i) need += (size_t)name_len[i] + max_label + 48;
            if (need > fmt_cap[t]) { free(fmt_buf[t]); fmt_buf[t] = (char *)malloc(need + 64); fmt_cap[t] = fmt_buf[t] ? need + 64 : 0; }
            uint64_t good = 0;
            size_t L = fmt_buf[t] ? utree_format_records(ctr, h_buf, name_off + a, name_len + a, h_res + a, b - a, fmt_buf[t], fmt_cap[t], &good) : (size_t)-1;
            fmt_len[t] = L == (size_t)-1 ? 0 : L;
#pragma omp critical(utree_fmt)
            {
                if (L == (size_t)-1) fmt_fail = 1;
                else st.good_finds += good;
            }
        }
        if (fmt_fail) { rc = UTREE_E_NOMEM; goto done; }
        for (int t = 0; t < T; ++t) if (fmt_len[t] && fwrite(fmt_buf[t], 1, fmt_len[t], fo) != fmt_len[t]) { rc = UTREE_E_IO; goto done; }
        st.n_reads += nr;
        while (st.n_reads >= next_progress) {                          /* itree.c:878 */
            printf("Searched %llu queries...\n", (unsigned long long)next_progress);
            next_progress += 1048576;
        }
        if (frc == UTREE_E_FASTA) { st.fasta_error = ferr; st.fasta_error.read_index += st.n_reads - nr; rc = UTREE_E_FASTA; goto done; }
        memmove(h_buf, h_buf + used, have - used);
        have -= used;
        if (eof && !have) break;
    }
done:
    if (fd >= 0) close(fd);
    if (fo) fclose(fo);
    if (G) { for (int g = 0; g < n_dev; ++g) free_ctx(&G[g]); free(G); }
    if (h_buf) hipHostFree(h_buf);
    if (h_res) hipHostFree(h_res);
    free(seq_off); free(name_off); free(rel_off); free(seq_len); free(name_len);
    if (fmt_buf) for (int t = 0; t < host_threads; ++t) free(fmt_buf[t]);
    free(fmt_buf); free(fmt_cap); free(fmt_len);
    st.seconds_total = now_s() - t_start;
    st.seconds_kernels = t_kernels;
    if (stats) *stats = st;
    return rc;
}
