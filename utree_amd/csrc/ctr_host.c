/* ctr_host.c -- host side of a `.ctr` database: header, bin table, labels.
 *
 * Replaces XT_read32 (itree.c:733-828) and readSamplesFPdelim (itree.c:1154-1223) for the search path.
 * Differences by design: errors are returned, not exit()ed; the node dump stays in the file (it is
 * streamed to HBM by dev_image.c); the bin table is kept at its on-disk width and zero-extended on the
 * device (the reference relies on malloc returning zeroed pages for the upper halves, itree.c:756-759).
 */
#define _FILE_OFFSET_BITS 64
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ctr_host.h"

const char *utree_strerror(int code) {
    switch (code) {
        case UTREE_OK: return "ok";
        case UTREE_E_IO: return "cannot open or read file";
        case UTREE_E_FORMAT: return "Tree malformatted.";
        case UTREE_E_UNSUPPORTED: return "unsupported PACKSIZE / CNTTYPE / IXTYPE in tree header";
        case UTREE_E_NOMEM: return "out of memory";
        case UTREE_E_HIP: return "HIP error (is a gfx950 device visible?)";
        case UTREE_E_ARG: return "bad argument";
        case UTREE_E_NOLABELS: return "No annotation found in tree file.";
        case UTREE_E_FASTA: return "malformed query file";
        case UTREE_E_RCCL: return "RCCL error";
        case UTREE_E_BUILD: return "BUILD input rejected";
        case UTREE_E_DEVICE: return "a kernel found the batch's workspace too small";
        default: return "unknown error";
    }
}
int utree_abi_version(void) { return UTREE_ABI_VERSION; }

/* ---- label table: index = order of first appearance; repeats map to the first index (itree.c:191-220) ---- */
static uint64_t hash_bytes(const char *s, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < n; ++i) h = (h ^ (unsigned char)s[i]) * 0x100000001B3ull;
    return h ^ (h >> 29);
}

static int cmp_label_ix(const void *a, const void *b, void *arg) {
    const utree_ctr *c = (const utree_ctr *)arg;
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return strcmp(c->labels[x], c->labels[y]);                        /* byStr, itree.c:830-832 */
}

static int parse_labels(utree_ctr *c, const char *text, size_t len) {
    c->label_text = (char *)malloc(len + 1);
    if (!c->label_text) return UTREE_E_NOMEM;
    memcpy(c->label_text, text, len);
    c->label_text[len] = 0;
    c->label_text_len = len;
    size_t lines = 1;
    for (size_t i = 0; i < len; ++i) lines += text[i] == '\n';
    size_t cap = 64;
    while (cap < 2 * lines) cap <<= 1;
    uint32_t *table = (uint32_t *)malloc(cap * sizeof(uint32_t));
    c->labels = (char **)malloc((lines + 1) * sizeof(char *));
    c->label_len = (uint32_t *)malloc((lines + 1) * sizeof(uint32_t));
    if (!table || !c->labels || !c->label_len) { free(table); return UTREE_E_NOMEM; }
    memset(table, 0xFF, cap * sizeof(uint32_t));
    uint32_t n = 0;
    char *p = c->label_text, *end = c->label_text + len;
    while (p < end) {
        char *nl = (char *)memchr(p, '\n', (size_t)(end - p));
        char *stop = nl ? nl : end;
        char *tab = (char *)memchr(p, '\t', (size_t)(stop - p));      /* label = up to the first TAB (1161) */
        char *lend = tab ? tab : stop;                                /* no TAB: reference runs off; whole line */
        *lend = 0;
        size_t L = strlen(p);                                         /* an embedded NUL ends it, as strcmp sees it */
        uint64_t h = hash_bytes(p, L) & (cap - 1);
        for (;;) {
            uint32_t v = table[h];
            if (v == 0xFFFFFFFFu) { table[h] = n; c->labels[n] = p; c->label_len[n] = (uint32_t)L; ++n; break; }
            if (c->label_len[v] == L && !memcmp(c->labels[v], p, L)) break;
            h = (h + 1) & (cap - 1);
        }
        p = stop + 1;
    }
    free(table);
    c->info.n_labels = n;
    if (!n) return UTREE_E_NOLABELS;
    /* strcmp order: the device stores ranks so that "sort by label text" (itree.c:1041) is an integer sort */
    c->rank2ix = (uint32_t *)malloc(n * sizeof(uint32_t));
    c->ix2rank = (uint32_t *)malloc(n * sizeof(uint32_t));
    if (!c->rank2ix || !c->ix2rank) return UTREE_E_NOMEM;
    for (uint32_t i = 0; i < n; ++i) c->rank2ix[i] = i;
    qsort_r(c->rank2ix, n, sizeof(uint32_t), cmp_label_ix, c);
    for (uint32_t r = 0; r < n; ++r) c->ix2rank[c->rank2ix[r]] = r;
    return UTREE_OK;
}

static int check_header(uint64_t W, uint64_t cnt, uint64_t I, uint64_t N) {
    if (!N) return UTREE_E_FORMAT;                                     /* itree.c:738 */
    if (cnt != 0) return UTREE_E_UNSUPPORTED;                          /* NO_COUNT builds only (itree.c:34) */
    if (!(W == 4 || W == 8 || W == 16) || !(I == 2 || I == 4)) return UTREE_E_UNSUPPORTED;   /* PACKSIZE 16, 32, 64 (README.md:87-88; 4 and 8 do not compile in the reference) */
    return UTREE_OK;
}

static void fill_info(utree_ctr *c, uint64_t W, uint64_t I, uint64_t N) {
    c->info.W = (uint32_t)W; c->info.I = (uint32_t)I; c->info.k = (uint32_t)(4 * W);
    c->info.SZ = (uint32_t)(W + I - 3);
    c->info.n_nodes = N;
    c->info.binix_width = N < 0xFFFFFFFFull ? 4 : 8;                   /* itree.c:757 */
}

static uint64_t binix_at(const utree_ctr *c, size_t i) {
    if (c->info.binix_width == 4) { uint32_t v; memcpy(&v, (const char *)c->binix_raw + 4 * i, 4); return v; }
    uint64_t v; memcpy(&v, (const char *)c->binix_raw + 8 * i, 8); return v;
}

int utree_ctr_open(const char *path, utree_ctr **out) {
    if (!path || !out) return UTREE_E_ARG;
    *out = NULL;
    FILE *fp = fopen(path, "rb");
    if (!fp) return UTREE_E_IO;                                        /* "Invalid DB file", itree.c:735 */
    uint64_t meta[4] = {0, 0, 0, 0};
    if (fread(meta, 8, 4, fp) < 4) { fclose(fp); return UTREE_E_FORMAT; }
    int rc = check_header(meta[0], meta[1], meta[2], meta[3]);
    if (rc) { fclose(fp); return rc; }
    utree_ctr *c = (utree_ctr *)calloc(1, sizeof *c);
    if (!c) { fclose(fp); return UTREE_E_NOMEM; }
    fill_info(c, meta[0], meta[2], meta[3]);
    c->hdr_W_raw = meta[0]; c->hdr_cnt_raw = meta[1]; c->hdr_I_raw = meta[2];
    c->path = strdup(path);
    size_t bbytes = (size_t)UTREE_NUMBINS * c->info.binix_width;
    c->binix_raw = malloc(bbytes);
    if (!c->binix_raw || !c->path) { fclose(fp); utree_ctr_close(c); return UTREE_E_NOMEM; }
    if (fread(c->binix_raw, 1, bbytes, fp) != bbytes) { fclose(fp); utree_ctr_close(c); return UTREE_E_FORMAT; }
    c->bins_read = UTREE_NUMBINS;
    c->records_file_off = 32 + (uint64_t)bbytes;
    uint64_t rec_bytes = c->info.n_nodes * c->info.SZ;
    fseeko(fp, 0, SEEK_END);
    uint64_t fsize = (uint64_t)ftello(fp);
    c->info.file_bytes = fsize;
    if (fsize < c->records_file_off + rec_bytes) { fclose(fp); utree_ctr_close(c); return UTREE_E_FORMAT; }  /* 768 */
    uint64_t text_off = c->records_file_off + rec_bytes;
    size_t tlen = (size_t)(fsize - text_off);
    char *text = (char *)malloc(tlen + 1);
    if (!text) { fclose(fp); utree_ctr_close(c); return UTREE_E_NOMEM; }
    fseeko(fp, (off_t)text_off, SEEK_SET);
    if (fread(text, 1, tlen, fp) != tlen) { free(text); fclose(fp); utree_ctr_close(c); return UTREE_E_IO; }
    fclose(fp);
    rc = parse_labels(c, text, tlen);
    free(text);
    if (rc) { utree_ctr_close(c); return rc; }
    c->info.bin_total = binix_at(c, UTREE_NUMBINS - 1);
    *out = c;
    return UTREE_OK;
}

int utree_ctr_from_memory(uint32_t W, uint32_t I, uint64_t n_nodes, const void *binix, uint32_t binix_width,
                          const void *h_records, const char *label_text, size_t label_len, utree_ctr **out) {
    if (!binix || !out || !label_text) return UTREE_E_ARG;
    *out = NULL;
    int rc = check_header(W, 0, I, n_nodes);
    if (rc) return rc;
    utree_ctr *c = (utree_ctr *)calloc(1, sizeof *c);
    if (!c) return UTREE_E_NOMEM;
    fill_info(c, W, I, n_nodes);
    c->hdr_W_raw = W; c->hdr_I_raw = I;
    if (binix_width != c->info.binix_width) { utree_ctr_close(c); return UTREE_E_ARG; }
    size_t bbytes = (size_t)UTREE_NUMBINS * binix_width;
    c->binix_raw = malloc(bbytes);
    if (!c->binix_raw) { utree_ctr_close(c); return UTREE_E_NOMEM; }
    memcpy(c->binix_raw, binix, bbytes);
    c->bins_read = UTREE_NUMBINS;
    if (h_records) {
        size_t rb = (size_t)(n_nodes * c->info.SZ);
        c->h_records = (uint8_t *)malloc(rb ? rb : 1);
        if (!c->h_records) { utree_ctr_close(c); return UTREE_E_NOMEM; }
        memcpy(c->h_records, h_records, rb);
    }
    rc = parse_labels(c, label_text, label_len);
    if (rc) { utree_ctr_close(c); return rc; }
    c->info.bin_total = binix_at(c, UTREE_NUMBINS - 1);
    c->info.file_bytes = 32 + bbytes + n_nodes * c->info.SZ + label_len;
    *out = c;
    return UTREE_OK;
}

void utree_ctr_close(utree_ctr *c) {
    if (!c) return;
    free(c->path); free(c->binix_raw); free(c->h_records); free(c->label_text); free(c->labels);
    free(c->label_len); free(c->rank2ix); free(c->ix2rank); free(c);
}

int utree_ctr_get_info(const utree_ctr *c, utree_ctr_info *info) {
    if (!c || !info) return UTREE_E_ARG;
    *info = c->info;
    return UTREE_OK;
}

const char *utree_ctr_label(const utree_ctr *c, uint32_t ix, uint32_t *len) {
    if (!c || ix >= c->info.n_labels) return NULL;
    if (len) *len = c->label_len[ix];
    return c->labels[ix];
}
