/* search_dev.c -- whole-file search with the byte work on the GPU (the "device text pipeline"): XT_doSearch32's GG branch
 * (itree.c:833-1108) where the host only moves bytes.
 *
 *   file --pread--> pinned chunk --H2D--> [newline scan, framing | classify, vote | format] --D2H--> pinned text --pwrite--> file
 *
 * The reference frames reads under `omp critical` (itree.c:867-874) and prints under the stdio lock (1032, 1040, 1096); the
 * round-1 pipeline (search.c) framed and formatted with host thread teams, which on a GPU box's CPU share (16 cores per
 * GPU) cost more than everything else together.  Here a chunk of the FASTA is cut at a record boundary ("\n>": in
 * well-formed input only header lines begin with '>'), goes to HBM as it stands, and comes back as its output text.
 *
 * Chunks are independent, so they are handed out to LANES -- host threads that each take the next chunk and carry it
 * through all stages on their own stream and buffers; several lanes per GPU overlap one chunk's file read with another's
 * kernels and a third's write, and with more GPUs the lanes simply belong to different devices (reads shard by chunk, no
 * exchange).  The only ordering between chunks is the output offset: chunk c is written at the sum of the text lengths of
 * the chunks before it, published in chunk order.
 *
 * Input the kernels do not take (NUL bytes, lines fgets would split, malformed records, records larger than a chunk,
 * more reads or text per chunk than the buffers hold) makes the whole search return UTREE_RETRY_HOST: search.c then runs
 * the file through the host framing, which reproduces the reference case by case on malformed input.
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
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "ctr_host.h"
#include "dev_image.h"
#include "search_dev.h"
#include "text_kernels.h"

#define DCHUNK_BYTES ((size_t)64 << 20)
#define DOUT_BYTES   ((size_t)128 << 20)       /* output text a chunk may produce                                  */
#define DMAX_READS   ((uint32_t)4 << 20)       /* reads per chunk (a chunk of shorter records takes the host path)  */
#define LINELEN_MAX 16777216u                  /* itree.c:836                                                      */
#define MAX_LANES 64

/* bytes per chunk: DCHUNK_BYTES; UTREE_CHUNK_BYTES lowers it (tests: many chunk boundaries in a small file) */
static size_t chunk_bytes(void) {
    const char *e = getenv("UTREE_CHUNK_BYTES");
    if (e && atoll(e) >= 64 && (size_t)atoll(e) < DCHUNK_BYTES) return (size_t)atoll(e);
    return DCHUNK_BYTES;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---- per-device buffers, kept in the utree_dev between searches (pinned memory is slow to allocate) ---- */
typedef struct {
    hipStream_t stream;
    uint8_t *h_in, *h_out;                      /* pinned                                                           */
    utk_text_meta *h_meta;                      /* pinned                                                           */
    uint8_t *d_in, *d_out;
    uint32_t *d_counts, *d_nl, *d_seq_len, *d_name_off, *d_name_len, *d_line_len;
    uint64_t *d_seq_off, *d_line_off;
    utree_result *d_res;
    utk_text_meta *d_meta;
    void *d_scan, *d_ws;
    size_t scan_bytes, ws_bytes;
    int ws_rc;                                  /* do_rc the workspace was sized for                                */
    int pageable;                               /* bit 0 / 1 / 2: h_in / h_out / h_meta are ordinary memory (no pinned memory was to be had) */
} lane_buf;

struct utree_search_ctx {
    int device, n_lanes;
    uint32_t n_labels;
    uint32_t *d_ix2rank;
    lane_buf lane[MAX_LANES];
};

static void host_free(void *p, int pageable) { if (!p) return; if (pageable) free(p); else hipHostFree(p); }
/* Pinned host memory, or -- when the host has none left to pin (it is a resource all the GPUs' users of a machine share) -- ordinary memory:
 * the copies to and from it are then staged by the runtime, slower, and the search still runs. */
static void *host_alloc(size_t bytes, int *pageable, const char *what) {
    void *p = NULL;
    *pageable = 0;
    if (!getenv("UTREE_TEST_NO_PINNED") && hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess) return p;
    const hipError_t e = hipGetLastError();
    p = NULL;
    if (posix_memalign(&p, 4096, bytes) != 0) p = NULL;
    if (getenv("UTREE_DEBUG") || getenv("UTREE_TIMING") || !p)
        fprintf(stderr, "[utree_amd] warning: no pinned host memory for %s (%zu bytes: %s); %s\n", what, bytes, hipGetErrorString(e), p ? "using pageable memory" : "and no ordinary memory either");
    *pageable = 1;
    return p;
}

static void lane_free(lane_buf *b) {
    host_free(b->h_in, b->pageable & 1);
    host_free(b->h_out, b->pageable & 2);
    host_free(b->h_meta, b->pageable & 4);
    void *dp[] = {b->d_in, b->d_out, b->d_counts, b->d_nl, b->d_seq_len, b->d_name_off, b->d_name_len, b->d_line_len, b->d_seq_off,
                  b->d_line_off, b->d_res, b->d_meta, b->d_scan, b->d_ws};
    for (size_t i = 0; i < sizeof dp / sizeof dp[0]; ++i) if (dp[i]) hipFree(dp[i]);
    if (b->stream) hipStreamDestroy(b->stream);
    memset(b, 0, sizeof *b);
}

void utree_search_ctx_free(void *p) {
    struct utree_search_ctx *c = (struct utree_search_ctx *)p;
    if (!c) return;
    hipSetDevice(c->device);
    for (int i = 0; i < c->n_lanes; ++i) lane_free(&c->lane[i]);
    if (c->d_ix2rank) hipFree(c->d_ix2rank);
    free(c);
}

/* (a lane whose allocation fails half way is freed whole: the next search starts it from nothing) */
#define HA(x) do { const hipError_t e_ = (x); if (e_ != hipSuccess) { (void)hipGetLastError(); size_t fr_ = 0, to_ = 0; (void)hipMemGetInfo(&fr_, &to_); \
    fprintf(stderr, "[utree_amd] warning: device %d: %s failed (%s; %.1f of %.1f GiB of HBM free)\n", dev->device, #x, hipGetErrorString(e_), (double)fr_ / 1073741824.0, (double)to_ / 1073741824.0); \
    lane_free(b); return UTREE_E_NOMEM; } } while (0)
#define HH(field, bytes, bit) do { int pg_ = 0; b->field = host_alloc((bytes), &pg_, #field); if (pg_) b->pageable |= (bit); if (!b->field) { lane_free(b); return UTREE_E_NOMEM; } } while (0)
static int lane_alloc(utree_dev *dev, lane_buf *b, int do_rc) {
    if (!b->stream) {
        b->ws_rc = -1;
        if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); b->stream = NULL; return UTREE_E_HIP; }
        HH(h_in, DCHUNK_BYTES + 64, 1);
        HH(h_out, DOUT_BYTES, 2);
        HH(h_meta, sizeof(utk_text_meta), 4);
        HA(hipMalloc((void **)&b->d_in, DCHUNK_BYTES + 256));
        HA(hipMalloc((void **)&b->d_out, DOUT_BYTES));
        HA(hipMalloc((void **)&b->d_counts, (DCHUNK_BYTES / 4096 + 2) * 4));
        HA(hipMalloc((void **)&b->d_nl, (size_t)2 * DMAX_READS * 4));
        HA(hipMalloc((void **)&b->d_seq_off, (size_t)DMAX_READS * 8));
        HA(hipMalloc((void **)&b->d_seq_len, (size_t)DMAX_READS * 4));
        HA(hipMalloc((void **)&b->d_name_off, (size_t)DMAX_READS * 4));
        HA(hipMalloc((void **)&b->d_name_len, (size_t)DMAX_READS * 4));
        HA(hipMalloc((void **)&b->d_line_len, (size_t)DMAX_READS * 4));
        HA(hipMalloc((void **)&b->d_line_off, (size_t)DMAX_READS * 8));
        HA(hipMalloc((void **)&b->d_res, (size_t)DMAX_READS * sizeof(utree_result)));
        HA(hipMalloc((void **)&b->d_meta, sizeof(utk_text_meta)));
        b->scan_bytes = utk_text_scan_temp_bytes(DMAX_READS);
        if (!b->scan_bytes) { lane_free(b); return UTREE_E_HIP; }
        HA(hipMalloc(&b->d_scan, b->scan_bytes));
    }
    if (b->ws_rc < do_rc) {                    /* the both-strands workspace also serves forward-only searches */
        if (b->d_ws) { hipFree(b->d_ws); b->d_ws = NULL; }
        b->ws_bytes = utree_classify_workspace_bytes(dev, DMAX_READS, DCHUNK_BYTES, LINELEN_MAX, do_rc);
        if (!b->ws_bytes) { lane_free(b); return UTREE_E_ARG; }
        HA(hipMalloc(&b->d_ws, b->ws_bytes));
        b->ws_rc = do_rc;
    }
    return UTREE_OK;
}

/* the device's context; lanes get their buffers on first use (lane_alloc), or all at once through utree_search_prepare */
static int ctx_get(const utree_ctr *ctr, utree_dev *dev, int n_lanes, struct utree_search_ctx **out) {
    if (hipSetDevice(dev->device) != hipSuccess) return UTREE_E_HIP;
    struct utree_search_ctx *c = (struct utree_search_ctx *)dev->search_ctx;
    if (c && c->n_labels != ctr->info.n_labels) { utree_search_ctx_free(c); dev->search_ctx = c = NULL; }
    if (!c) {
        c = (struct utree_search_ctx *)calloc(1, sizeof *c);
        if (!c) return UTREE_E_NOMEM;
        c->device = dev->device; c->n_labels = ctr->info.n_labels;
        /* (published only when complete: a later search must not find a context without its table) */
        if (hipMalloc((void **)&c->d_ix2rank, ((size_t)c->n_labels + 1) * 4) != hipSuccess) { (void)hipGetLastError(); free(c); return UTREE_E_NOMEM; }
        if (hipMemcpy(c->d_ix2rank, ctr->ix2rank, (size_t)c->n_labels * 4, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError(); hipFree(c->d_ix2rank); free(c); return UTREE_E_HIP;
        }
        dev->search_ctx = c;
    }
    if (n_lanes > MAX_LANES) n_lanes = MAX_LANES;
    if (n_lanes > c->n_lanes) c->n_lanes = n_lanes;
    *out = c;
    return UTREE_OK;
}

/* ---- the search --------------------------------------------------------------------------------------------------- */
typedef struct dpipe dpipe;
typedef struct {
    dpipe *P;
    utree_dev *dev;
    lane_buf *b;
    int read_threads;
    pthread_t th;
    double t_read, t_frame, t_classify, t_d2h, t_wait, t_wlock, t_write;
    uint64_t chunks;
} lane_t;

struct dpipe {
    const utree_ctr *ctr;
    int fd, fo, do_rc;
    off_t file_size;
    size_t chunk_bytes;
    pthread_mutex_t mu; pthread_cond_t cv;
    /* one pwrite at a time PER FILE: writers of one file serialise on its inode anyway, and concurrent ones only fight over it (tmpfs,
     * measured: 4 writers 3.0 GB/s, one 5.7 GB/s).  Different files do not: UTREE_OUTPUT_PARTS=P writes <out>.part000 ... <out>.part(P-1) --
     * part p takes the chunks that start in the p-th P-th of the input, so the parts' concatenation is the one-file output -- and fills them side
     * by side (tools/hostio_probe3.c: 1 file 7.5 GB/s, 4 files 28.8, 16 files 93) */
    int n_parts;
    int fo_part[UTREE_MAX_OUT_PARTS];
    pthread_mutex_t wmu[UTREE_MAX_OUT_PARTS];
    uint64_t taken_part[UTREE_MAX_OUT_PARTS], published_part[UTREE_MAX_OUT_PARTS];   /* chunks handed out / whose output offset is fixed, per part */
    off_t cum_part[UTREE_MAX_OUT_PARTS];        /* text bytes in front of a part's next chunk                         */
    off_t next_off; uint64_t n_taken;           /* chunk dispenser: where the last chunk handed out ends, chunks handed out */
    off_t next_off_part[UTREE_MAX_OUT_PARTS], part_end[UTREE_MAX_OUT_PARTS];   /* ... per part of the input (find_parts); parts take turns */
    int next_part;
    off_t cum_out;                              /* text bytes of all published chunks                               */
    int seq_out; uint64_t written;              /* the output cannot seek (pipe, FIFO, tty): chunks are written in order with write() */
    uint64_t printed;                           /* progress lines already on stdout (a host re-run must not repeat them) */
    uint64_t n_reads, good, next_progress, bytes_in;
    int rc, stop;                               /* first error; UTREE_RETRY_HOST = input for the host path          */
    int lanes_alive; lane_t *lane0;             /* lanes that have not given up for want of buffers (lane_main); the test hook's surviving lane */
    /* seq_out: a chunk that needs the host framing does not stop the chunks in front of it -- they are written, in order, and the host
     * pipeline continues from that chunk's first byte (search_dev.h: utree_search_resume); the chunks behind it are dropped */
    int retry_set; uint64_t retry_at; off_t retry_off;
    uint8_t tail[(64 << 10) + 8];
};

static void dfail(dpipe *P, int rc) {
    pthread_mutex_lock(&P->mu);
    if (!P->rc) P->rc = rc;
    P->stop = 1;
    pthread_cond_broadcast(&P->cv);
    pthread_mutex_unlock(&P->mu);
}

/* chunk `c`, which starts at byte `off` of the input, is not for this pipeline (called with or without the lock: `locked`) */
static void dretry(dpipe *P, uint64_t c, off_t off, int locked) {
    if (!locked) pthread_mutex_lock(&P->mu);
    if (!P->seq_out) { if (!P->rc) P->rc = UTREE_RETRY_HOST; P->stop = 1; }
    else if (!P->retry_set || c < P->retry_at) { P->retry_set = 1; P->retry_at = c; P->retry_off = off; }
    pthread_cond_broadcast(&P->cv);
    if (!locked) pthread_mutex_unlock(&P->mu);
}
#define DROPPED(P, c) ((P)->retry_set && (c) > (P)->retry_at)      /* a chunk behind the one the host pipeline continues from */

/* Next chunk [off, off+len): ends after a '\n' that a '>' follows, or at the end of its part of the file.  Called with the lock held.
 * Returns 0 = none left, 1 = chunk, -1 = no record boundary inside a chunk's worth of bytes.
 * The input is cut into n_parts consecutive ranges at record boundaries (find_parts; one range = the whole file without UTREE_OUTPUT_PARTS)
 * and the chunks are dealt out from the parts in turn, so that the lanes at work write to DIFFERENT output files. */
static int take_chunk(dpipe *P, off_t *off, size_t *len, int *final, uint64_t *index, int *part, uint64_t *in_part) {
    int p = -1;
    for (int k = 0; k < P->n_parts; ++k) {
        const int q = (P->next_part + k) % P->n_parts;
        if (P->next_off_part[q] < P->part_end[q]) { p = q; break; }
    }
    if (p < 0) return 0;
    P->next_part = (p + 1) % P->n_parts;
    const off_t a = P->next_off_part[p], end = P->part_end[p];
    *part = p;
    *in_part = P->taken_part[p];
    const size_t cb = P->chunk_bytes;
    size_t want = (size_t)(end - a < (off_t)cb ? end - a : (off_t)cb);
    if (a + (off_t)want == end) {                                /* the rest of the part: it ends at a record boundary, or with the file */
        *off = a; *len = want; *final = end == P->file_size; *index = P->n_taken++; P->taken_part[p]++; P->next_off_part[p] = end;
        P->next_off = a + (off_t)want;
        return 1;
    }
    /* look backwards from the end of the range for "\n>", a window at a time (one more byte: the '>' may be the byte after) */
    size_t hi = want;                                            /* candidates: newline positions < hi (relative to a) */
    while (hi > 0) {
        const size_t win = hi < (64u << 10) ? hi : (64u << 10), lo = hi - win;
        size_t got = 0;
        while (got < win + 1) {
            ssize_t r = pread(P->fd, P->tail + got, win + 1 - got, a + (off_t)(lo + got));
            if (r <= 0) return -1;
            got += (size_t)r;
        }
        for (size_t i = win; i-- > 0;) {
            if (P->tail[i] == '\n' && P->tail[i + 1] == '>') {
                *off = a; *len = lo + i + 1; *final = 0; *index = P->n_taken++; P->taken_part[p]++;
                P->next_off_part[p] = a + (off_t)(lo + i + 1);
                P->next_off = P->next_off_part[p];
                return 1;
            }
        }
        hi = lo;
    }
    P->next_off = a;                                             /* (where the host pipeline would have to go on from) */
    return -1;
}

/* the parts of the input: part p = [begin_p, begin_{p+1}), begin_p = the first record start at or behind p / n_parts of the file (a '>' that
 * follows a newline; none: the part is empty and its neighbour in front takes the bytes) */
static int find_parts(dpipe *P) {
    off_t begin[UTREE_MAX_OUT_PARTS + 1];
    begin[0] = 0; begin[P->n_parts] = P->file_size;
    for (int p = 1; p < P->n_parts; ++p) {
        off_t at = (off_t)((unsigned __int128)P->file_size * (unsigned)p / (unsigned)P->n_parts), found = P->file_size;
        if (at < begin[p - 1]) at = begin[p - 1];
        while (at < P->file_size && found == P->file_size) {
            const size_t win = (size_t)(P->file_size - at < (off_t)(64 << 10) ? P->file_size - at : (off_t)(64 << 10));
            size_t got = 0;
            while (got < win) {
                ssize_t r = pread(P->fd, P->tail + got, win - got, at + (off_t)got);
                if (r <= 0) return UTREE_E_IO;
                got += (size_t)r;
            }
            for (size_t i = 0; i + 1 < win; ++i) if (P->tail[i] == '\n' && P->tail[i + 1] == '>') { found = at + (off_t)i + 1; break; }
            at += (off_t)(win > 1 ? win - 1 : 1);                /* (the window's last byte is looked at again as a first byte) */
        }
        begin[p] = found;
    }
    for (int p = 0; p < P->n_parts; ++p) { P->next_off_part[p] = begin[p]; P->part_end[p] = begin[p + 1]; }
    return UTREE_OK;
}

#define LH(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); dfail(P, UTREE_E_HIP); return NULL; } } while (0)
#define LK(x) do { if ((x) != 0) { dfail(P, UTREE_E_HIP); return NULL; } } while (0)

static void *lane_main(void *arg) {
    lane_t *L = (lane_t *)arg;
    dpipe *P = L->P;
    lane_buf *b = L->b;
    if (hipSetDevice(L->dev->device) != hipSuccess) { dfail(P, UTREE_E_HIP); return NULL; }
    for (;;) {
        off_t off = 0; size_t len = 0; int final = 0, part = 0; uint64_t c = 0, cp = 0;      /* c: the chunk's number; cp: its number within its output part */
        if (!b->stream || b->ws_rc < P->do_rc) {                  /* before this lane's first chunk: its buffers (only if there is a chunk to take) */
            pthread_mutex_lock(&P->mu);
            int work = !(P->stop || P->retry_set);
            if (work) { work = 0; for (int k = 0; k < P->n_parts; ++k) if (P->next_off_part[k] < P->part_end[k]) work = 1; }
            pthread_mutex_unlock(&P->mu);
            if (!work) return NULL;
            int arc = getenv("UTREE_TEST_LANE_NOMEM") && L != P->lane0 ? UTREE_E_NOMEM : lane_alloc(L->dev, b, P->do_rc);
            if (arc) {
                /* a lane that cannot get its buffers (~200 MB of pinned host memory, ~1 GB of HBM) steps aside: the other lanes take the chunks.
                 * Only when no lane is left does the search fail. */
                pthread_mutex_lock(&P->mu);
                const int left = --P->lanes_alive;
                pthread_mutex_unlock(&P->mu);
                if (getenv("UTREE_DEBUG") || getenv("UTREE_TIMING")) fprintf(stderr, "[utree_amd] device %d: a lane could not allocate its buffers (%s); %d lane(s) left\n", L->dev->device, utree_strerror(arc), left);
                if (left <= 0) dfail(P, arc);
                return NULL;
            }
        }
        pthread_mutex_lock(&P->mu);
        int got = (P->stop || P->retry_set) ? 0 : take_chunk(P, &off, &len, &final, &c, &part, &cp);
        if (got < 0) dretry(P, P->n_taken, P->next_off, 1);                    /* no record boundary within a chunk's bytes */
        pthread_mutex_unlock(&P->mu);
        if (got <= 0) return NULL;
        /* ---- file -> pinned memory (the page-cache copy), a small team ---- */
        double t0 = now_s();
        {
            int T = L->read_threads, bad = 0;
            if ((size_t)T > len / ((size_t)4 << 20) + 1) T = (int)(len / ((size_t)4 << 20) + 1);
#pragma omp parallel for num_threads(T) schedule(static, 1) reduction(| : bad)
            for (int t = 0; t < T; ++t) {
                size_t a = len * (size_t)t / (size_t)T, e = len * (size_t)(t + 1) / (size_t)T;
                while (a < e) {
                    ssize_t r = pread(P->fd, b->h_in + a, e - a, off + (off_t)a);
                    if (r <= 0) { bad |= 1; break; }
                    a += (size_t)r;
                }
            }
            if (bad) { dfail(P, UTREE_E_IO); return NULL; }
        }
        size_t n = len;
        if (final && b->h_in[n - 1] != '\n') b->h_in[n++] = '\n';      /* the last line needs no newline (fgets ends at EOF) */
        double t1 = now_s();
        L->t_read += t1 - t0;
        /* ---- newlines, framing ---- */
        LH(hipMemcpyAsync(b->d_in, b->h_in, n, hipMemcpyHostToDevice, b->stream));
        LK(utk_text_newlines(b->d_in, n, b->d_counts, b->d_nl, 2 * DMAX_READS, b->d_meta, b->stream));
        /* (the smallest well-formed record is ">\n\n": three bytes -- the reference takes it as a read of length 0) */
        const uint32_t grid_reads = n / 3 + 1 > DMAX_READS ? DMAX_READS : (uint32_t)(n / 3 + 1);
        LK(utk_text_frame(b->d_in, b->d_nl, DMAX_READS, grid_reads, b->d_seq_off, b->d_seq_len, b->d_name_off, b->d_name_len,
                          b->d_meta, b->stream));
        LH(hipMemcpyAsync(b->h_meta, b->d_meta, sizeof(utk_text_meta), hipMemcpyDeviceToHost, b->stream));
        LH(hipStreamSynchronize(b->stream));
        double t2 = now_s();
        L->t_frame += t2 - t1;
        const utk_text_meta m1 = *b->h_meta;
        if (m1.flags || (m1.n_lines & 1u) || m1.n_lines > 2 * DMAX_READS || m1.n_lines / 2 > grid_reads) { dretry(P, c, off, 0); return NULL; }
        const uint32_t nr = m1.n_lines / 2;
        /* ---- classify (kernels.hip), then the output text ---- */
        if (nr) {
            int e = utree_classify_batch(L->dev, b->d_in, b->d_seq_off, b->d_seq_len, nr, m1.total_bases, m1.max_len, P->do_rc, b->d_res,
                                         b->d_ws, b->ws_bytes, b->stream);
            if (e) { dfail(P, e); return NULL; }
            LK(utk_text_format(&L->dev->kimg, ((struct utree_search_ctx *)L->dev->search_ctx)->d_ix2rank, b->d_in, b->d_res, b->d_name_off,
                               b->d_name_len, nr, b->d_line_len, b->d_line_off, b->d_scan, b->scan_bytes, b->d_out, DOUT_BYTES, b->d_meta, 0,
                               b->stream));
        }
        LH(hipMemcpyAsync(b->h_meta, b->d_meta, sizeof(utk_text_meta), hipMemcpyDeviceToHost, b->stream));
        LH(hipStreamSynchronize(b->stream));
        double t3 = now_s();
        L->t_classify += t3 - t2;
        const utk_text_meta m2 = *b->h_meta;
        if (nr) { int pe = utree_classify_poll(L->dev); if (pe) { dfail(P, pe); return NULL; } }   /* the batch's error word came back with it */
        if (m2.flags || m2.out_bytes > DOUT_BYTES) { dretry(P, c, off, 0); return NULL; }
        if (nr && m2.out_bytes) {
            LK(utk_text_format(&L->dev->kimg, ((struct utree_search_ctx *)L->dev->search_ctx)->d_ix2rank, b->d_in, b->d_res, b->d_name_off,
                               b->d_name_len, nr, b->d_line_len, b->d_line_off, b->d_scan, b->scan_bytes, b->d_out, DOUT_BYTES, b->d_meta, 1,
                               b->stream));
            LH(hipMemcpyAsync(b->h_out, b->d_out, (size_t)m2.out_bytes, hipMemcpyDeviceToHost, b->stream));
        }
        /* ---- the chunk's place in the output: after the text of every earlier chunk (input order, like one thread) ---- */
        off_t base;
        pthread_mutex_lock(&P->mu);
        while (P->published_part[part] != cp && !P->stop && !DROPPED(P, c)) pthread_cond_wait(&P->cv, &P->mu);
        if (P->stop || DROPPED(P, c)) { pthread_mutex_unlock(&P->mu); hipStreamSynchronize(b->stream); return NULL; }
        base = P->cum_part[part];
        P->cum_part[part] += (off_t)m2.out_bytes;
        P->cum_out += (off_t)m2.out_bytes;
        P->n_reads += nr; P->good += nr ? m2.good_finds : 0; P->bytes_in += len;
        while (P->n_reads >= P->next_progress) {                                   /* itree.c:878 */
            printf("Searched %llu queries...\n", (unsigned long long)P->next_progress);
            P->next_progress += 1048576;
            P->printed++;
        }
        P->published_part[part]++;
        pthread_cond_broadcast(&P->cv);
        pthread_mutex_unlock(&P->mu);
        double t4 = now_s();
        L->t_wait += t4 - t3;
        LH(hipStreamSynchronize(b->stream));
        double t5 = now_s();
        L->t_d2h += t5 - t4;
        size_t done = 0;
        if (P->seq_out) {                                           /* wait for every earlier chunk's text to be out */
            pthread_mutex_lock(&P->mu);
            while (P->written != c && !P->stop && !DROPPED(P, c)) pthread_cond_wait(&P->cv, &P->mu);
            const int stop = P->stop || DROPPED(P, c);
            pthread_mutex_unlock(&P->mu);
            if (stop) return NULL;
        }
        pthread_mutex_lock(&P->wmu[part]);
        double t6 = now_s();
        while (done < (size_t)m2.out_bytes) {
            ssize_t w = P->seq_out ? write(P->fo, b->h_out + done, (size_t)m2.out_bytes - done)
                                   : pwrite(P->fo_part[part], b->h_out + done, (size_t)m2.out_bytes - done, base + (off_t)done);
            if (w <= 0) { pthread_mutex_unlock(&P->wmu[part]); dfail(P, UTREE_E_IO); return NULL; }
            done += (size_t)w;
        }
        pthread_mutex_unlock(&P->wmu[part]);
        if (P->seq_out) {
            pthread_mutex_lock(&P->mu);
            P->written++;
            pthread_cond_broadcast(&P->cv);
            pthread_mutex_unlock(&P->mu);
        }
        L->t_wlock += t6 - t5;
        L->t_write += now_s() - t6;
        L->chunks++;
    }
}

static int lanes_per_device(int n_dev) {
    const char *e = getenv("UTREE_LANES");
    if (e && atoi(e) >= 1 && atoi(e) <= 16) return atoi(e);
    return n_dev <= 2 ? 4 : 3;
}

int utree_output_parts(void) {
    const char *e = getenv("UTREE_OUTPUT_PARTS");
    const int p = e ? atoi(e) : 1;
    return p < 1 ? 1 : p > UTREE_MAX_OUT_PARTS ? UTREE_MAX_OUT_PARTS : p;
}

int utree_search_prepare(const utree_ctr *ctr, utree_dev **devs, int n_dev, int do_rc) {
    if (!ctr || !devs || n_dev < 1) return UTREE_E_ARG;
    const int K = lanes_per_device(n_dev);
    for (int g = 0; g < n_dev; ++g) {
        struct utree_search_ctx *c = NULL;
        int rc = ctx_get(ctr, devs[g], K, &c);
        for (int i = 0; !rc && i < K; ++i) rc = lane_alloc(devs[g], &c->lane[i], do_rc);
        if (rc) return rc;
    }
    return UTREE_OK;
}

int utree_search_file_device(const utree_ctr *ctr, utree_dev **devs, int n_dev, const char *fasta_path, const char *out_path,
                             int do_rc, int host_threads, utree_search_stats *stats, uint64_t *progress_printed, utree_search_resume *resume) {
    const double t_start = now_s();
    if (progress_printed) *progress_printed = 0;
    if (resume) { memset(resume, 0, sizeof *resume); resume->fo = -1; }
    dpipe *P = (dpipe *)calloc(1, sizeof *P);
    if (!P) return UTREE_E_NOMEM;
    P->ctr = ctr; P->do_rc = do_rc; P->next_progress = 1048576;
    P->fd = open(fasta_path, O_RDONLY);
    P->n_parts = utree_output_parts();
    for (int p = 0; p < P->n_parts; ++p) {
        char name[4096];
        if (P->n_parts > 1) snprintf(name, sizeof name, "%s.part%03d", out_path, p);
        P->fo_part[p] = open(P->n_parts > 1 ? name : out_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);   /* fopen(outfile, "wb"), itree.c:834 */
        if (P->fo_part[p] < 0) { for (int q = 0; q < p; ++q) close(P->fo_part[q]); P->fo_part[0] = -1; break; }
    }
    P->fo = P->fo_part[0];
    if (resume) resume->parts = P->n_parts;
    struct stat sb;
    if (P->fd < 0 || P->fo < 0 || fstat(P->fd, &sb) != 0) {                       /* itree.c:835 */
        if (P->fd >= 0) close(P->fd);
        if (P->fo >= 0) for (int p = 0; p < P->n_parts; ++p) close(P->fo_part[p]);
        free(P);
        return UTREE_E_IO;
    }
    P->seq_out = P->n_parts == 1 && lseek(P->fo, 0, SEEK_CUR) == (off_t)-1;       /* `out` is a pipe, a FIFO, a tty: no pwrite there */
    if (!S_ISREG(sb.st_mode)) {                                                   /* the input is a pipe: no pread */
        if (P->seq_out && resume) resume->fo = P->fo; else for (int p = 0; p < P->n_parts; ++p) close(P->fo_part[p]);   /* (an output that is a pipe too stays open for the host pipeline) */
        close(P->fd); free(P);
        return UTREE_RETRY_HOST;
    }
    P->file_size = sb.st_size;
    P->chunk_bytes = chunk_bytes();
    { int prc = find_parts(P); if (prc) { for (int p = 0; p < P->n_parts; ++p) close(P->fo_part[p]); close(P->fd); free(P); return prc; } }
    const int K = lanes_per_device(n_dev), n_lanes = K * n_dev;
    int rc = UTREE_OK;
    for (int g = 0; g < n_dev && !rc; ++g) { struct utree_search_ctx *c = NULL; rc = ctx_get(ctr, devs[g], K, &c); }
#ifdef _OPENMP
    if (host_threads <= 0) host_threads = omp_get_max_threads();
#else
    host_threads = 1;
#endif
    if (host_threads > 16 * n_dev) host_threads = 16 * n_dev;                     /* a GPU's share of the host cores */
    int per_lane = host_threads / n_lanes;
    if (per_lane < 1) per_lane = 1;
    if (per_lane > 8) per_lane = 8;
    lane_t *lanes = (lane_t *)calloc((size_t)n_lanes, sizeof(lane_t));
    if (!lanes && !rc) rc = UTREE_E_NOMEM;
    pthread_mutex_init(&P->mu, NULL);
    for (int p = 0; p < P->n_parts; ++p) pthread_mutex_init(&P->wmu[p], NULL);
    pthread_cond_init(&P->cv, NULL);
    if (!rc) {
        int started = 0;
        for (int i = 0; i < n_lanes; ++i) {                                       /* lane i: device i % n_dev, so consecutive chunks go to different GPUs */
            lanes[i].P = P; lanes[i].dev = devs[i % n_dev];
            lanes[i].b = &((struct utree_search_ctx *)devs[i % n_dev]->search_ctx)->lane[i / n_dev];
            lanes[i].read_threads = per_lane;
            if (i == 0) { P->lane0 = &lanes[0]; P->lanes_alive = n_lanes; }
            if (pthread_create(&lanes[i].th, NULL, lane_main, &lanes[i])) { dfail(P, UTREE_E_NOMEM); break; }
            ++started;
        }
        for (int i = 0; i < started; ++i) pthread_join(lanes[i].th, NULL);
        rc = P->rc;
    }
    if (!rc && P->retry_set) {
        /* every chunk in front of `retry_at` is written; the host pipeline takes the rest of the input and the open descriptor */
        rc = UTREE_RETRY_HOST;
        if (resume) {
            resume->fo = P->fo; resume->in_off = (long long)P->retry_off;
            resume->n_reads = P->n_reads; resume->good_finds = P->good; resume->bytes_in = P->bytes_in; resume->bytes_out = (uint64_t)P->cum_out;
        }
    }
    close(P->fd);
    if (!(resume && resume->fo == P->fo)) for (int p = 0; p < P->n_parts; ++p) close(P->fo_part[p]);
    if (progress_printed) *progress_printed = P->printed;
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->n_reads = P->n_reads; stats->good_finds = P->good;
        stats->bytes_in = P->bytes_in; stats->bytes_out = (uint64_t)P->cum_out;
        stats->pipeline = 1; stats->n_lanes = n_lanes;
        for (int i = 0; lanes && i < n_lanes; ++i) {
            stats->seconds_read += lanes[i].t_read; stats->seconds_frame += lanes[i].t_frame;
            stats->seconds_classify_format += lanes[i].t_classify; stats->seconds_d2h += lanes[i].t_d2h;
            stats->seconds_order_wait += lanes[i].t_wait + lanes[i].t_wlock; stats->seconds_write += lanes[i].t_write;
        }
        stats->seconds_kernels = stats->seconds_frame + stats->seconds_classify_format;
        stats->seconds_total = now_s() - t_start;
    }
    if ((getenv("UTREE_DEBUG") || getenv("UTREE_TIMING")) && lanes && rc == UTREE_OK) {
        double r = 0, f = 0, c = 0, d = 0, w = 0, o = 0;
        for (int i = 0; i < n_lanes; ++i) { r += lanes[i].t_read; f += lanes[i].t_frame; c += lanes[i].t_classify; d += lanes[i].t_d2h; o += lanes[i].t_wait + lanes[i].t_wlock; w += lanes[i].t_write; }
        fprintf(stderr, "[utree_amd] device text pipeline, %d lanes x %d read threads, lane-seconds: read %.3f | H2D+frame %.3f | classify+format %.3f | "
                        "order + write-turn wait %.3f | D2H %.3f | write %.3f; wall %.3f s\n", n_lanes, per_lane, r, f, c, o, d, w, now_s() - t_start);
    }
    pthread_mutex_destroy(&P->mu);
    for (int p = 0; p < P->n_parts; ++p) pthread_mutex_destroy(&P->wmu[p]);
    pthread_cond_destroy(&P->cv);
    free(lanes);
    free(P);
    return rc;
}
