/* search.c -- whole-file search: XT_doSearch32's GG branch (itree.c:833-1108) over one or more device
 * images.  Host orchestration in C, four overlapped stages connected by a ring of chunk slots:
 *
 *   reader   : parallel pread of the next ~96 MiB of FASTA into pinned memory, frame the reads (fasta.c;
 *              the reference does this under `omp critical`, itree.c:867-874 -- its scaling limit)
 *   gpu      : shard the framed reads contiguously over the GPUs; per GPU copy the shard's byte span to HBM
 *              as it stands, run the batch kernels, copy the 24-byte results back
 *   formatter: format the lines with a thread team
 *   writer   : write them in input order (= what the reference writes with one thread; with more threads it
 *              writes a permutation)
 */
#define _FILE_OFFSET_BITS 64
#define _GNU_SOURCE
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "ctr_host.h"
#include "dev_image.h"
#include "search_dev.h"

#define CHUNK_BYTES ((size_t)96 << 20)        /* must hold two maximal (16 MiB) lines                    */
#define MAX_READS_PER_BATCH ((size_t)2 << 20)  /* more reads in a chunk (tiny reads) simply take another batch */
#define LINELEN_MAX 16777216u                 /* itree.c:836 */
#define NSLOTS 4
#define READ_THREADS 4

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

enum { S_EMPTY = 0, S_FRAMED, S_DONE, S_FORMATTED };
#define FMT_MAX_THREADS 16

typedef struct {
    uint8_t *h_buf;                            /* pinned: the chunk as read from the file                   */
    size_t have;                               /* valid bytes                                               */
    size_t nr, used;                           /* framed reads, bytes they cover                            */
    uint64_t *seq_off, *name_off, *rel_off;
    uint32_t *seq_len, *name_len;
    utree_result *h_res;                       /* pinned                                                    */
    int frame_rc, last;
    utree_fasta_error ferr;
    int state;
    char *fmt_buf[FMT_MAX_THREADS];            /* formatted lines, one piece per formatting thread, in input order */
    size_t fmt_cap[FMT_MAX_THREADS], fmt_len[FMT_MAX_THREADS];
    int fmt_T;
    uint64_t good;
} slot_t;

static int slot_alloc(slot_t *s) {
    if (s->h_buf) return UTREE_OK;
    if (hipHostMalloc((void **)&s->h_buf, CHUNK_BYTES + 64, hipHostMallocDefault) != hipSuccess) return UTREE_E_NOMEM;
    if (hipHostMalloc((void **)&s->h_res, MAX_READS_PER_BATCH * sizeof(utree_result), hipHostMallocDefault) != hipSuccess) return UTREE_E_NOMEM;
    if (hipHostMalloc((void **)&s->rel_off, MAX_READS_PER_BATCH * 8, hipHostMallocDefault) != hipSuccess) return UTREE_E_NOMEM;
    if (hipHostMalloc((void **)&s->seq_len, MAX_READS_PER_BATCH * 4, hipHostMallocDefault) != hipSuccess) return UTREE_E_NOMEM;
    s->seq_off = (uint64_t *)malloc(MAX_READS_PER_BATCH * 8); s->name_off = (uint64_t *)malloc(MAX_READS_PER_BATCH * 8);
    s->name_len = (uint32_t *)malloc(MAX_READS_PER_BATCH * 4);
    return (s->seq_off && s->name_off && s->name_len) ? UTREE_OK : UTREE_E_NOMEM;
}

typedef struct {
    utree_dev *dev;
    hipStream_t stream;
    uint8_t *d_buf; uint64_t *d_off; uint32_t *d_len; utree_result *d_out; void *d_ws; size_t ws_bytes;
} gpu_ctx;

typedef struct {
    const utree_ctr *ctr;
    gpu_ctx *G; int n_dev;
    int fd, fo;                                 /* input, output                                            */
    off_t out_pos, start_off;                   /* start_off: where the input begins for this pipeline (behind what the device pipeline wrote to a pipe) */
    int do_rc, host_threads;
    const utree_rank_params *rank;              /* non-NULL: the rank-specific `xtree-search` (rank.c), one device  */
    int input_format;                           /* UTREE_INPUT_*: opt-in FASTQ / multi-line FASTA (+ gzip via zlib)   */
    gzFile gz;
    slot_t slot[NSLOTS];
    pthread_mutex_t mu; pthread_cond_t cv;
    int rc;                                     /* first error of any stage                                  */
    int stop;
    utree_search_stats st;
    double t_read, t_frame, t_gpu, t_format, t_write;
    uint32_t max_label;
    uint64_t progress_printed;                  /* progress lines the device pipeline printed before it handed the file back */
} pipe_t;

static void set_error(pipe_t *P, int rc) {
    pthread_mutex_lock(&P->mu);
    if (!P->rc) P->rc = rc;
    P->stop = 1;
    pthread_cond_broadcast(&P->cv);
    pthread_mutex_unlock(&P->mu);
}
/* wait until slot reaches `state` (or the pipeline stops); returns 0 if stopped */
static int wait_state(pipe_t *P, slot_t *s, int state) {
    pthread_mutex_lock(&P->mu);
    while (s->state != state && !P->stop) pthread_cond_wait(&P->cv, &P->mu);
    int ok = s->state == state;
    pthread_mutex_unlock(&P->mu);
    return ok;
}
static void set_state(pipe_t *P, slot_t *s, int state) {
    pthread_mutex_lock(&P->mu);
    s->state = state;
    pthread_cond_broadcast(&P->cv);
    pthread_mutex_unlock(&P->mu);
}

/* ---- stage 1: read + frame ------------------------------------------------------------------- */
static void *reader_main(void *arg) {
    pipe_t *P = (pipe_t *)arg;
    off_t file_pos = P->start_off;
    size_t carry = 0;                           /* bytes of an incomplete read carried into the next chunk */
    const uint8_t *carry_src = NULL;
    int eof = 0;
    for (int i = 0; !eof || carry; ++i) {
        slot_t *s = &P->slot[i % NSLOTS];
        if (!wait_state(P, s, S_EMPTY)) return NULL;
        if (i < NSLOTS) {                       /* pinned memory is slow to allocate: do it while earlier chunks are in flight */
            int arc = slot_alloc(s);
            if (arc) { set_error(P, arc); return NULL; }
        }
        if (carry) memmove(s->h_buf, carry_src, carry);
        size_t have = carry;
        double t0 = now_s();
        if (!eof && P->gz) {                      /* opt-in formats: zlib reads plain and gzip input alike */
            size_t want = CHUNK_BYTES - have, done = 0;
            while (done < want) {
                int r = gzread(P->gz, s->h_buf + have + done, (unsigned)(want - done > (1u << 30) ? (1u << 30) : want - done));
                if (r < 0) { set_error(P, UTREE_E_IO); return NULL; }
                if (r == 0) { eof = 1; break; }
                done += (size_t)r;
            }
            have += done;
        } else if (!eof) {
            /* parallel pread: the page-cache copy is the cost; several threads stream it */
            size_t want = CHUNK_BYTES - have;
            ssize_t got[READ_THREADS];
            int T = READ_THREADS;
#pragma omp parallel for num_threads(T) schedule(static, 1)
            for (int t = 0; t < T; ++t) {
                size_t a = want * (size_t)t / (size_t)T, b = want * (size_t)(t + 1) / (size_t)T, done = 0;
                got[t] = 0;
                while (done < b - a) {
                    ssize_t r = pread(P->fd, s->h_buf + have + a + done, b - a - done, file_pos + (off_t)(a + done));
                    if (r < 0) { got[t] = -1; break; }
                    if (r == 0) break;
                    done += (size_t)r; got[t] = (ssize_t)done;
                }
            }
            size_t total = 0;
            for (int t = 0; t < T; ++t) {
                if (got[t] < 0) { set_error(P, UTREE_E_IO); return NULL; }
                size_t seg = want * (size_t)(t + 1) / (size_t)T - want * (size_t)t / (size_t)T;
                total += (size_t)got[t];
                if ((size_t)got[t] < seg) { eof = 1; break; }       /* short segment: end of file inside it */
            }
            have += total; file_pos += (off_t)total;
        }
        double t1 = now_s();
        s->have = have;
        s->nr = 0; s->used = 0; s->frame_rc = UTREE_OK; s->last = 0;
        if (have && P->input_format == UTREE_INPUT_AUTO)
            P->input_format = s->h_buf[0] == '@' ? UTREE_INPUT_FASTQ : UTREE_INPUT_FASTA_MULTILINE;
        if (have) {
            s->frame_rc = P->input_format != UTREE_INPUT_REFERENCE
                ? utree_reads_frame(s->h_buf, have, eof, P->input_format, MAX_READS_PER_BATCH, s->seq_off, s->seq_len, s->name_off,
                                    s->name_len, &s->nr, &s->used, &s->ferr)
                : utree_fasta_frame(s->h_buf, have, eof, MAX_READS_PER_BATCH, s->seq_off, s->seq_len, s->name_off,
                                            s->name_len, &s->nr, &s->used, &s->ferr);
            if (s->frame_rc != UTREE_OK && s->frame_rc != UTREE_E_FASTA) { set_error(P, s->frame_rc); return NULL; }
            if (s->frame_rc == UTREE_OK && !s->nr && !s->used && !eof) {
                s->frame_rc = UTREE_E_FASTA; s->ferr.code = 5; s->ferr.read_index = 0;   /* a line pair larger than a chunk */
            }
        }
        P->t_read += t1 - t0; P->t_frame += now_s() - t1;
        carry = have - s->used; carry_src = s->h_buf + s->used;
        if (s->frame_rc == UTREE_E_FASTA) { carry = 0; eof = 1; }
        if (eof && !carry) s->last = 1;
        if (eof && carry && s->frame_rc == UTREE_OK && s->used == 0 && s->nr == 0) { s->last = 1; carry = 0; }
        set_state(P, s, S_FRAMED);
        if (s->last) break;
    }
    return NULL;
}

/* ---- stage 2: GPU ---------------------------------------------------------------------------- */
#define HIPOK(x) do { if ((x) != hipSuccess) { set_error(P, UTREE_E_HIP); return NULL; } } while (0)

static void *gpu_main(void *arg) {
    pipe_t *P = (pipe_t *)arg;
    for (int i = 0;; ++i) {
        slot_t *s = &P->slot[i % NSLOTS];
        if (!wait_state(P, s, S_FRAMED)) return NULL;
        double t0 = now_s();
        size_t nr = s->nr, n_dev = (size_t)P->n_dev;
        size_t per = (nr + n_dev - 1) / n_dev;
        for (size_t g = 0; g < n_dev && nr; ++g) {
            gpu_ctx *c = &P->G[g];
            size_t first = g * per; if (first > nr) first = nr;
            size_t count = first + per <= nr ? per : nr - first;
            if (!count) continue;
            size_t last = first + count - 1;
            size_t lo = (size_t)s->seq_off[first], hi = (size_t)s->seq_off[last] + s->seq_len[last];
            uint64_t total = 0; uint32_t mx = 0;
            for (size_t r = first; r <= last; ++r) {
                s->rel_off[r] = s->seq_off[r] - lo; total += s->seq_len[r];
                if (s->seq_len[r] > mx) mx = s->seq_len[r];
            }
            HIPOK(hipSetDevice(c->dev->device));
            HIPOK(hipMemcpyAsync(c->d_buf, s->h_buf + lo, hi - lo, hipMemcpyHostToDevice, c->stream));
            HIPOK(hipMemcpyAsync(c->d_off, s->rel_off + first, count * 8, hipMemcpyHostToDevice, c->stream));
            HIPOK(hipMemcpyAsync(c->d_len, s->seq_len + first, count * 4, hipMemcpyHostToDevice, c->stream));
            int e = P->rank ? utree_rank_batch(c->dev, c->d_buf, c->d_off, c->d_len, (uint32_t)count, total, mx, P->do_rc,
                                               P->rank, c->d_out, c->d_ws, c->ws_bytes, c->stream)
                            : utree_classify_batch(c->dev, c->d_buf, c->d_off, c->d_len, (uint32_t)count, total, mx, P->do_rc,
                                                   c->d_out, c->d_ws, c->ws_bytes, c->stream);
            if (e) { set_error(P, e); return NULL; }
            HIPOK(hipMemcpyAsync(s->h_res + first, c->d_out, count * sizeof(utree_result), hipMemcpyDeviceToHost, c->stream));
        }
        for (size_t g = 0; g < n_dev && nr; ++g) {
            HIPOK(hipSetDevice(P->G[g].dev->device));
            HIPOK(hipStreamSynchronize(P->G[g].stream));
            if (!P->rank) { int pe = utree_classify_poll(P->G[g].dev); if (pe) { set_error(P, pe); return NULL; } }   /* the batches' error words */
        }
        P->t_gpu += now_s() - t0;
        int last = s->last;
        set_state(P, s, S_DONE);
        if (last) return NULL;
    }
}

/* ---- stage 3: format ------------------------------------------------------------------------ */
static void *format_main(void *arg) {
    pipe_t *P = (pipe_t *)arg;
    int T0 = P->host_threads;
    for (int i = 0;; ++i) {
        slot_t *s = &P->slot[i % NSLOTS];
        if (!wait_state(P, s, S_DONE)) break;
        double t0 = now_s();
        size_t nr = s->nr;
        int T = T0;
        if ((size_t)T > nr / 4096 + 1) T = (int)(nr / 4096 + 1);
        int fail = 0;
        uint64_t good_total = 0;
#pragma omp parallel for num_threads(T) schedule(static, 1) reduction(+ : good_total) reduction(| : fail)
        for (int t = 0; t < T; ++t) {
            size_t a = nr * (size_t)t / (size_t)T, b = nr * (size_t)(t + 1) / (size_t)T, need = 64;
            for (size_t r = a; r < b; ++r) {
                const utree_result *q = &s->h_res[r];
                if (!q->found) continue;
                uint32_t ll = q->label < P->ctr->info.n_labels ? P->ctr->label_len[q->label] : 0;
                need += (size_t)s->name_len[r] + ll + 64;
            }
            if (need > s->fmt_cap[t]) { free(s->fmt_buf[t]); s->fmt_buf[t] = (char *)malloc(need + need / 4); s->fmt_cap[t] = s->fmt_buf[t] ? need + need / 4 : 0; }
            uint64_t good = 0;
            size_t L = !s->fmt_buf[t] ? (size_t)-1
                       : P->rank ? utree_format_rank_records(P->ctr, s->h_buf, s->name_off + a, s->name_len + a, s->h_res + a, b - a,
                                                             s->fmt_buf[t], s->fmt_cap[t], &good)
                                 : utree_format_records(P->ctr, s->h_buf, s->name_off + a, s->name_len + a, s->h_res + a, b - a,
                                                        s->fmt_buf[t], s->fmt_cap[t], &good);
            if (L == (size_t)-1) { fail |= 1; s->fmt_len[t] = 0; } else { s->fmt_len[t] = L; good_total += good; }
        }
        P->t_format += now_s() - t0;
        if (fail) { set_error(P, UTREE_E_NOMEM); break; }
        s->fmt_T = T; s->good = good_total;
        int last = s->last;
        set_state(P, s, S_FORMATTED);
        if (last || s->frame_rc == UTREE_E_FASTA) break;
    }
    return NULL;
}

/* ---- stage 4: write -------------------------------------------------------------------------- */
static void *writer_main(void *arg) {
    pipe_t *P = (pipe_t *)arg;
    uint64_t next_progress = 1048576 * (P->progress_printed + 1);
    for (int i = 0;; ++i) {
        slot_t *s = &P->slot[i % NSLOTS];
        if (!wait_state(P, s, S_FORMATTED)) break;
        double t1 = now_s();
        size_t nr = s->nr;
        /* pieces in input order; writes to one file serialise in the kernel anyway, so one thread issues them */
        for (int t = 0; t < s->fmt_T; ++t) {
            size_t done = 0;
            while (done < s->fmt_len[t]) {
                ssize_t w = write(P->fo, s->fmt_buf[t] + done, s->fmt_len[t] - done);
                if (w <= 0) { set_error(P, UTREE_E_IO); return NULL; }
                done += (size_t)w;
            }
            P->out_pos += (off_t)s->fmt_len[t];
        }
        P->t_write += now_s() - t1;
        P->st.good_finds += s->good;
        P->st.n_reads += nr;
        while (P->st.n_reads >= next_progress) {                                  /* itree.c:878 */
            printf("Searched %llu queries...\n", (unsigned long long)next_progress);
            next_progress += 1048576;
        }
        int last = s->last;
        if (s->frame_rc == UTREE_E_FASTA) {                                       /* reads before the bad one are written */
            P->st.fasta_error = s->ferr;
            P->st.fasta_error.read_index += P->st.n_reads - nr;
            set_error(P, UTREE_E_FASTA);
            break;
        }
        set_state(P, s, S_EMPTY);
        if (last) break;
    }
    return NULL;
}

static void free_ctx(gpu_ctx *g) {
    if (!g->dev) return;
    hipSetDevice(g->dev->device);
    if (g->d_buf) hipFree(g->d_buf);
    if (g->d_off) hipFree(g->d_off);
    if (g->d_len) hipFree(g->d_len);
    if (g->d_out) hipFree(g->d_out);
    if (g->d_ws) hipFree(g->d_ws);
    if (g->stream) hipStreamDestroy(g->stream);
}

#define HIPM(x) do { if ((x) != hipSuccess) { rc = UTREE_E_HIP; goto done; } } while (0)

static int search_file(const utree_ctr *ctr, utree_dev **devs, int n_dev, const char *fasta_path, const char *out_path,
                       int do_rc, const utree_rank_params *rank, int host_threads, int input_format, utree_search_stats *stats) {
    if (!ctr || !devs || n_dev < 1 || !fasta_path || !out_path || input_format < 0 || input_format > UTREE_INPUT_AUTO) return UTREE_E_ARG;
    int rc = UTREE_OK;
    uint64_t dev_printed = 0;
    utree_search_resume resume;
    memset(&resume, 0, sizeof resume); resume.fo = -1;
    /* The GG search on the reference's input format takes the device text pipeline (search_dev.c); it hands back input it
     * does not take -- malformed records, NUL bytes, lines fgets would split -- and the host framing below then reproduces
     * the reference on it case by case.  UTREE_HOST_TEXT=1 forces the host pipeline (tests, A/B). */
    if (!rank && input_format == UTREE_INPUT_REFERENCE && !getenv("UTREE_HOST_TEXT")) {
        rc = utree_search_file_device(ctr, devs, n_dev, fasta_path, out_path, do_rc, host_threads, stats, &dev_printed, &resume);
        if (rc != UTREE_RETRY_HOST) return rc;
        rc = UTREE_OK;
    }
    double t_start = now_s();
    pipe_t *P = (pipe_t *)calloc(1, sizeof *P);
    if (!P) { if (resume.fo >= 0) close(resume.fo); return UTREE_E_NOMEM; }
    P->ctr = ctr; P->n_dev = n_dev; P->do_rc = do_rc; P->rank = rank; P->input_format = input_format;
    P->progress_printed = dev_printed;
    P->fd = open(fasta_path, O_RDONLY);
    if (resume.fo >= 0) {
        /* the output is a pipe and the device pipeline has written the chunks in front of `in_off`: go on from there, on the same descriptor */
        P->fo = resume.fo; P->start_off = (off_t)resume.in_off;
        P->st.n_reads = resume.n_reads; P->st.good_finds = resume.good_finds; P->st.bytes_in = resume.bytes_in; P->st.bytes_out = resume.bytes_out;
    } else if (utree_output_parts() > 1) {
        /* UTREE_OUTPUT_PARTS: this pipeline writes in input order with one writer, so all of the output is part 000 and the other parts are
         * empty files -- the parts' concatenation is the output, as with the device pipeline's side-by-side parts */
        char name[4096];
        for (int p = utree_output_parts() - 1; p >= 0; --p) {
            snprintf(name, sizeof name, "%s.part%03d", out_path, p);
            const int f = open(name, O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (p == 0) P->fo = f; else if (f >= 0) close(f); else { P->fo = -1; break; }
        }
    } else
    P->fo = open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);                   /* fopen(outfile, "wb"), itree.c:834 */
    if (P->fd < 0 || P->fo < 0) {                                                 /* itree.c:835 */
        if (P->fd >= 0) close(P->fd);
        if (P->fo >= 0) close(P->fo);
        free(P);
        return UTREE_E_IO;
    }
    if (input_format != UTREE_INPUT_REFERENCE) {
        P->gz = gzdopen(dup(P->fd), "rb");
        if (!P->gz) { close(P->fd); close(P->fo); free(P); return UTREE_E_IO; }
        gzbuffer(P->gz, 1u << 20);
    }
#ifdef _OPENMP
    if (host_threads <= 0) host_threads = omp_get_max_threads();
#else
    host_threads = 1;
#endif
    if (host_threads > 16) host_threads = 16;                                     /* formatting saturates well before that */
    P->host_threads = host_threads;
    pthread_mutex_init(&P->mu, NULL);
    pthread_cond_init(&P->cv, NULL);
    for (uint32_t i = 0; i < ctr->info.n_labels; ++i) if (ctr->label_len[i] > P->max_label) P->max_label = ctr->label_len[i];
    P->G = (gpu_ctx *)calloc((size_t)n_dev, sizeof(gpu_ctx));
    if (!P->G) { rc = UTREE_E_NOMEM; goto done; }
    for (int g = 0; g < n_dev; ++g) {
        gpu_ctx *c = &P->G[g];
        c->dev = devs[g];
        HIPM(hipSetDevice(devs[g]->device));
        HIPM(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HIPM(hipMalloc((void **)&c->d_buf, CHUNK_BYTES + 64));
        HIPM(hipMalloc((void **)&c->d_off, MAX_READS_PER_BATCH * 8));
        HIPM(hipMalloc((void **)&c->d_len, MAX_READS_PER_BATCH * 4));
        HIPM(hipMalloc((void **)&c->d_out, MAX_READS_PER_BATCH * sizeof(utree_result)));
        c->ws_bytes = rank ? utree_rank_workspace_bytes(devs[g], (uint32_t)MAX_READS_PER_BATCH, CHUNK_BYTES, LINELEN_MAX, do_rc, rank)
                           : utree_classify_workspace_bytes(devs[g], (uint32_t)MAX_READS_PER_BATCH, CHUNK_BYTES, LINELEN_MAX, do_rc);
        if (!c->ws_bytes) { rc = UTREE_E_ARG; goto done; }
        HIPM(hipMalloc(&c->d_ws, c->ws_bytes));
    }
    {
        pthread_t tr, tg, tf, tw;
        pthread_create(&tr, NULL, reader_main, P);
        pthread_create(&tg, NULL, gpu_main, P);
        pthread_create(&tf, NULL, format_main, P);
        pthread_create(&tw, NULL, writer_main, P);
        pthread_join(tw, NULL);
        /* the writer ends last on success; on error make sure the others leave their waits */
        pthread_mutex_lock(&P->mu); P->stop = 1; pthread_cond_broadcast(&P->cv); pthread_mutex_unlock(&P->mu);
        pthread_join(tr, NULL);
        pthread_join(tg, NULL);
        pthread_join(tf, NULL);
        rc = P->rc;
    }
done:
    if (P->gz) gzclose(P->gz);
    if (P->fd >= 0) close(P->fd);
    if (P->fo >= 0) close(P->fo);
    if (P->G) { for (int g = 0; g < n_dev; ++g) free_ctx(&P->G[g]); free(P->G); }
    for (int i = 0; i < NSLOTS; ++i) {
        slot_t *s = &P->slot[i];
        if (s->h_buf) hipHostFree(s->h_buf);
        if (s->h_res) hipHostFree(s->h_res);
        if (s->rel_off) hipHostFree(s->rel_off);
        if (s->seq_len) hipHostFree(s->seq_len);
        free(s->seq_off); free(s->name_off); free(s->name_len);
        for (int t = 0; t < FMT_MAX_THREADS; ++t) free(s->fmt_buf[t]);
    }
    P->st.seconds_total = now_s() - t_start;
    P->st.seconds_kernels = P->t_gpu;
    P->st.pipeline = 0; P->st.n_lanes = 1;
    P->st.seconds_read = P->t_read; P->st.seconds_frame = P->t_frame; P->st.seconds_classify_format = P->t_gpu;
    P->st.seconds_d2h = P->t_format; P->st.seconds_write = P->t_write;
    if (getenv("UTREE_DEBUG") || getenv("UTREE_TIMING"))
        fprintf(stderr, "[utree_amd] stages: read %.3f s, frame %.3f s | gpu %.3f s | format %.3f s, write %.3f s (overlapped)\n",
                P->t_read, P->t_frame, P->t_gpu, P->t_format, P->t_write);
    if (stats) *stats = P->st;
    pthread_mutex_destroy(&P->mu);
    pthread_cond_destroy(&P->cv);
    free(P);
    return rc;
}

int utree_search_file(const utree_ctr *ctr, utree_dev **devs, int n_dev, const char *fasta_path, const char *out_path,
                      int do_rc, int host_threads, utree_search_stats *stats) {
    return search_file(ctr, devs, n_dev, fasta_path, out_path, do_rc, NULL, host_threads, UTREE_INPUT_REFERENCE, stats);
}
int utree_search_file_opts(const utree_ctr *ctr, utree_dev **devs, int n_dev, const char *reads_path, const char *out_path,
                           int do_rc, int host_threads, int input_format, utree_search_stats *stats) {
    return search_file(ctr, devs, n_dev, reads_path, out_path, do_rc, NULL, host_threads, input_format, stats);
}

/* XT_doSearch32(utree, in, out, 0, speed, doRC) (itree.c:1376 without DO_GG): the same three-stage pipeline; the
 * batches go to ONE device in file order because each read's vote depends on the reads before it (rank.c). */
int utree_rank_search_file(const utree_ctr *ctr, utree_dev *dev, const char *fasta_path, const char *out_path, int do_rc,
                           const utree_rank_params *params, int host_threads, utree_search_stats *stats) {
    return utree_rank_search_file_opts(ctr, dev, fasta_path, out_path, do_rc, params, host_threads, UTREE_INPUT_REFERENCE, stats);
}
int utree_rank_search_file_opts(const utree_ctr *ctr, utree_dev *dev, const char *reads_path, const char *out_path, int do_rc,
                                const utree_rank_params *params, int host_threads, int input_format, utree_search_stats *stats) {
    if (!dev || !params) return UTREE_E_ARG;
    int rc = utree_rank_reset(dev);
    if (rc) return rc;
    return search_file(ctr, &dev, 1, reads_path, out_path, do_rc, params, host_threads, input_format, stats);
}
