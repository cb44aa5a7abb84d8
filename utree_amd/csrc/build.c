/* build.c -- host side of the database BUILD (`utree-build`, `utree-buildGG`; itree.c main 1379-1407) behind the C-ABI:
 * the `name \t label` map (itree.c:505-571), FASTA framing (573-590), the label "universe" the device folds over, the
 * numbering of labels in the reference's order of creation, and the `.ubt` / `.log` files (1317-1343, 1225-1232).
 * The k-mer work is in build_gpu.hip.  SURVEY.md §8(f) rank 3.
 */
#define _FILE_OFFSET_BITS 64
#define _GNU_SOURCE
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "../../include/utree_amd.h"
#include "build_gpu.h"

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

static uint8_t *read_all(const char *path, uint64_t *n) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return NULL;
    off_t sz = lseek(fd, 0, SEEK_END);
    uint8_t *b = sz >= 0 ? (uint8_t *)malloc((size_t)sz + 16) : NULL;
    uint64_t got = 0;
    while (b && got < (uint64_t)sz) {
        ssize_t r = pread(fd, b + got, (size_t)((uint64_t)sz - got), (off_t)got);
        if (r <= 0) { free(b); b = NULL; break; }
        got += (uint64_t)r;
    }
    close(fd);
    if (b) { memset(b + sz, 0, 16); *n = (uint64_t)sz; }
    return b;
}

/* ---- interning of byte strings (ptr, len) -> dense ids, strings kept NUL-terminated in one blob ---- */
typedef struct { char *blob; uint64_t blob_n, blob_cap; uint64_t *off; uint32_t *len; uint32_t n, cap; uint32_t *slot; uint32_t nslot; } strtab;
static uint64_t hbytes(const char *s, size_t n) { uint64_t h = 1469598103934665603ull; for (size_t i = 0; i < n; ++i) { h ^= (uint8_t)s[i]; h *= 1099511628211ull; } return h; }
static int st_rehash(strtab *t) {
    uint32_t ns = t->nslot ? t->nslot * 2 : 4096;
    uint32_t *sl = (uint32_t *)malloc(sizeof(uint32_t) * ns);
    if (!sl) return -1;
    memset(sl, 0xFF, sizeof(uint32_t) * ns);
    for (uint32_t i = 0; i < t->n; ++i) {
        uint64_t h = hbytes(t->blob + t->off[i], t->len[i]) & (ns - 1);
        while (sl[h] != 0xFFFFFFFFu) h = (h + 1) & (ns - 1);
        sl[h] = i;
    }
    free(t->slot); t->slot = sl; t->nslot = ns;
    return 0;
}
/* id of s[0..n); *fresh = 1 when it was added now.  0xFFFFFFFF on allocation failure. */
static uint32_t st_intern(strtab *t, const char *s, size_t n, int *fresh) {
    if ((uint64_t)t->n * 2 >= t->nslot && st_rehash(t)) return 0xFFFFFFFFu;
    uint64_t h = hbytes(s, n) & (t->nslot - 1);
    while (t->slot[h] != 0xFFFFFFFFu) {
        uint32_t i = t->slot[h];
        if (t->len[i] == n && !memcmp(t->blob + t->off[i], s, n)) { *fresh = 0; return i; }
        h = (h + 1) & (t->nslot - 1);
    }
    if (t->n == t->cap) {
        uint32_t nc = t->cap ? t->cap * 2 : 1024;
        uint64_t *o = (uint64_t *)realloc(t->off, sizeof(uint64_t) * nc);
        uint32_t *l = (uint32_t *)realloc(t->len, sizeof(uint32_t) * nc);
        if (o) t->off = o;
        if (l) t->len = l;
        if (!o || !l) return 0xFFFFFFFFu;
        t->cap = nc;
    }
    if (t->blob_n + n + 1 > t->blob_cap) {
        uint64_t nc = t->blob_cap ? t->blob_cap * 2 : (1u << 16);
        while (nc < t->blob_n + n + 1) nc *= 2;
        char *b = (char *)realloc(t->blob, nc);
        if (!b) return 0xFFFFFFFFu;
        t->blob = b; t->blob_cap = nc;
    }
    memcpy(t->blob + t->blob_n, s, n); t->blob[t->blob_n + n] = 0;
    t->off[t->n] = t->blob_n; t->len[t->n] = (uint32_t)n;
    t->blob_n += n + 1;
    t->slot[h] = t->n;
    *fresh = 1;
    return t->n++;
}
static void st_free(strtab *t) { free(t->blob); free(t->off); free(t->len); free(t->slot); }

typedef struct { const char *name, *label; } mapent;
static int by_name(const void *a, const void *b) {                                    /* xcmp, itree.c:486-489 */
    const unsigned char *x = (const unsigned char *)((const mapent *)a)->name, *y = (const unsigned char *)((const mapent *)b)->name;
    while (*x == *y) { if (!*x) return 0; ++x; ++y; }
    return (int)*x - (int)*y;
}
/* crBST (itree.c:475-484) with the reference's probe order; key = NUL-terminated */
static long find_name(const mapent *e, size_t last, const char *key) {
    const mapent *p = e;
    size_t sz = last;
    while (sz) {
        size_t w = sz >> 1;
        const char *r = p[w + 1].name, *k = key;
        while (*r == *k) { if (!*r) return (long)(p + w + 1 - e); ++r; ++k; }
        if (*r < *k) { p += w + 1; sz -= w + 1; } else sz = w;
    }
    return strcmp(p->name, key) ? -1 : (long)(p - e);
}

typedef struct { uint64_t time; uint32_t seq, u; } event;
static int by_time(const void *a, const void *b) {
    const event *x = (const event *)a, *y = (const event *)b;
    if (x->time != y->time) return x->time < y->time ? -1 : 1;
    return x->seq < y->seq ? -1 : x->seq > y->seq;
}

static int write_all(int fd, const void *p, size_t n) {
    const char *c = (const char *)p;
    while (n) { ssize_t w = write(fd, c, n); if (w <= 0) return -1; c += w; n -= (size_t)w; }
    return 0;
}

int utree_build_file(const char *fasta_path, const char *map_path, const char *ubt_path, uint32_t W, uint32_t I, int complevel,
                     int gg, int device, utree_build_stats *stats) {
    if (!fasta_path || !map_path || !ubt_path || (W != 8 && W != 16) || (I != 2 && I != 4) || complevel < 0 || complevel > 4)
        return UTREE_E_ARG;
    utree_build_stats st; memset(&st, 0, sizeof st);
    st.W = W; st.I = I;
    const double t0 = now_s();
    int rc = UTREE_OK;
    const uint32_t EMPTY = (I == 2 ? 0xFFFFu : 0xFFFFFFFFu) - 1;                       /* itree.c:105-106 */
    uint64_t fn = 0, mn = 0;
    uint8_t *fa = read_all(fasta_path, &fn), *mp = read_all(map_path, &mn);
    mapent *ent = NULL;
    uint64_t *seq_off = NULL; uint32_t *seq_len = NULL, *ref_u = NULL, *trunc_off = NULL, *trunc_ids = NULL, *ix_of_u = NULL;
    uint64_t *per_label = NULL;
    event *ev = NULL;
    strtab U; memset(&U, 0, sizeof U);
    utk_build_state *S = NULL;
    utk_build_result res; memset(&res, 0, sizeof res);
    uint32_t n_trunc = 0, trunc_cap = 0, n_refs = 0, ref_cap = 0, n_labels = 0;
    int fd = -1;
    if (!fa || !mp) { rc = UTREE_E_IO; goto done; }                                     /* "Invalid input file(s)" (504): exit 1 */
    if (!mn) { rc = UTREE_E_IO; st.error_kind = UTREE_BUILD_E_MAP_EMPTY; goto done; }   /* "Input map empty." (512): exit 1 */
    /* ---- the map (itree.c:513-571; name column 0, label column 1) ---- */
    size_t lines = 0;
    for (uint64_t i = 0; i < mn; ++i) lines += mp[i] == '\n';
    if (mp[mn - 1] != '\n') ++lines;
    st.map_bytes = mn; st.map_lines = lines;                                            /* "Parsed map. %llu bytes, %llu lines." (510, 515) */
    ent = (mapent *)malloc(sizeof(mapent) * (lines ? lines : 1));
    if (!ent) { rc = UTREE_E_NOMEM; goto done; }
    {
        char *ptr = (char *)mp;
        for (size_t i = 0; i < lines; ++i) {
            st.error_line = i;
#define MAP_FAIL(why) do { rc = UTREE_E_BUILD; st.error_kind = UTREE_BUILD_E_MAP; st.map_error = (why); goto done; } while (0)
            if (*ptr == '\n' || *ptr == '\r') MAP_FAIL(UTREE_MAP_E_BLANK_NAME);                /* 530-533 */
            if (*ptr == '\t') MAP_FAIL(UTREE_MAP_E_EXTRA_TAB);                                  /* 537 */
            ent[i].name = ptr;
            while (*++ptr != '\t') if (!*ptr) MAP_FAIL(UTREE_MAP_E_NO_TAB);                     /* 538 */
            *ptr++ = 0;
            if (*ptr == '\n' || *ptr == '\r') { st.error_line = i + 1; MAP_FAIL(UTREE_MAP_E_BLANK_LABEL); }   /* 541-544 */
            ent[i].label = ptr;
            while (*ptr != '\n') {                                                     /* 547-551: a last line without '\n' is an error */
                if (!*ptr) MAP_FAIL(UTREE_MAP_E_NO_NEWLINE);
                if (*ptr == '\r' || *ptr == '\t') *ptr = 0;
                ptr++;
            }
#undef MAP_FAIL
            *ptr++ = 0;
        }
    }
    qsort(ent, lines, sizeof(mapent), by_name);
    /* ---- references: (header line, sequence line) pairs (573-590); labels -> universe ids ---- */
    {
        uint64_t pos = 0, ns = 0;
        while (pos < fn) {
            ++ns;
            st.error_line = ns;
            uint8_t *nl = (uint8_t *)memchr(fa + pos, '\n', fn - pos);
            uint64_t hl = nl ? (uint64_t)(nl - (fa + pos)) + 1 : fn - pos;
            uint8_t saved = 0;
            if (nl) { saved = *nl; *nl = 0; }                                          /* 577-578: the name is the rest of the line */
            long pre = find_name(ent, lines - 1, (const char *)fa + pos + 1);
            if (nl) *nl = saved;
            if (pre < 0) { rc = UTREE_E_BUILD; st.error_kind = UTREE_BUILD_E_NAME; st.n_seqs = ns; goto done; }   /* 582: exit 4 */
            /* the label and every cut of it before a ';' (what xeTreeU_RF can turn it into, itree.c:286-301) */
            const char *lab = ent[pre].label;
            size_t ll = strlen(lab);
            int fresh;
            uint32_t u = st_intern(&U, lab, ll, &fresh);
            if (u == 0xFFFFFFFFu) { rc = UTREE_E_NOMEM; goto done; }
            if (fresh) {
                uint32_t semis = 0;
                for (size_t q = 0; q < ll; ++q) semis += lab[q] == ';';
                if (U.n + semis + 8 > trunc_cap || n_trunc + semis + 8 > trunc_cap) {
                    uint32_t nc = trunc_cap ? trunc_cap * 2 : 4096;
                    while (nc < U.n + semis + 8 || nc < n_trunc + semis + 8) nc *= 2;
                    uint32_t *a = (uint32_t *)realloc(trunc_off, sizeof(uint32_t) * nc), *b = (uint32_t *)realloc(trunc_ids, sizeof(uint32_t) * nc);
                    if (a) trunc_off = a;
                    if (b) trunc_ids = b;
                    if (!a || !b) { rc = UTREE_E_NOMEM; goto done; }
                    trunc_cap = nc;
                }
                const uint32_t list = n_trunc;
                trunc_off[u] = list;
                uint32_t m = 0;
                for (size_t q = 0; q < ll; ++q) if (lab[q] == ';') {
                    int f2;
                    uint32_t c = st_intern(&U, lab, q, &f2);
                    if (c == 0xFFFFFFFFu) { rc = UTREE_E_NOMEM; goto done; }
                    if (f2) trunc_off[c] = list;                                        /* its own cuts are the first m entries of this list */
                    trunc_ids[n_trunc++] = c;
                    ++m;
                }
            }
            pos += hl;
            if (pos >= fn) { rc = UTREE_E_BUILD; st.error_kind = UTREE_BUILD_E_FASTA; st.n_seqs = ns; goto done; }   /* 585-586: exit 2 */
            nl = (uint8_t *)memchr(fa + pos, '\n', fn - pos);
            uint64_t sl = nl ? (uint64_t)(nl - (fa + pos)) + 1 : fn - pos;
            const uint8_t *z = (const uint8_t *)memchr(fa + pos, 0, sl);
            uint64_t length = z ? (uint64_t)(z - (fa + pos)) : sl;                      /* 588: strlen */
            if (length && fa[pos + length - 1] == '\n') --length;                       /* 589 */
            if (length && fa[pos + length - 1] == '\r') --length;                       /* 590 */
            if (length > 0xFFFFFFFFull) { rc = UTREE_E_UNSUPPORTED; goto done; }
            if (n_refs == ref_cap) {
                uint32_t nc = ref_cap ? ref_cap * 2 : 1024;
                uint64_t *a = (uint64_t *)realloc(seq_off, 8ull * nc);
                uint32_t *b = (uint32_t *)realloc(seq_len, 4ull * nc), *c = (uint32_t *)realloc(ref_u, 4ull * nc);
                if (a) seq_off = a;
                if (b) seq_len = b;
                if (c) ref_u = c;
                if (!a || !b || !c) { rc = UTREE_E_NOMEM; goto done; }
                ref_cap = nc;
            }
            seq_off[n_refs] = pos; seq_len[n_refs] = (uint32_t)length; ref_u[n_refs] = u;
            ++n_refs;
            pos += sl;
        }
        st.n_seqs = ns;
    }
    if (!n_refs) { rc = UTREE_E_BUILD; st.error_kind = UTREE_BUILD_E_NO_KMERS; goto done; }
    if (U.n + 1 > trunc_cap) { uint32_t *a = (uint32_t *)realloc(trunc_off, sizeof(uint32_t) * (U.n + 8)); if (!a) { rc = UTREE_E_NOMEM; goto done; } trunc_off = a; }
    trunc_off[U.n] = n_trunc;
    /* ---- device: k-mers, stable sort, per-k-mer replay ---- */
    {
        utk_build_job job = {fa, fn, seq_off, seq_len, ref_u, n_refs, U.blob, U.blob_n, U.off, U.n, trunc_off, trunc_ids, n_trunc,
                             W, I, (uint32_t)complevel, gg, device};
        rc = utk_build_phase1(&job, &res, &S);
        if (rc) goto done;
    }
    st.n_kmers = res.n_occ; st.n_nodes = res.n_nodes; st.n_distinct = res.n_distinct;
    if (!res.n_occ) { rc = UTREE_E_BUILD; st.error_kind = UTREE_BUILD_E_NO_KMERS; goto done; }          /* 631: exit 2 */
    /* ---- label indices in the reference's order of creation: a reference's label when the reference is parsed (583),
     *      a cut label at the collision that first produced it (297); the clock is the position in the input ---- */
    {
        uint64_t nev = 0;
        ev = (event *)malloc(sizeof(event) * ((size_t)n_refs + U.n + 1));
        ix_of_u = (uint32_t *)malloc(sizeof(uint32_t) * (U.n + 1));
        if (!ev || !ix_of_u) { rc = UTREE_E_NOMEM; goto done; }
        for (uint32_t r = 0; r < n_refs; ++r) { ev[nev].time = res.h_ref_time[r]; ev[nev].seq = r; ev[nev].u = ref_u[r]; ++nev; }
        for (uint32_t u = 0; u < U.n; ++u) if (res.h_first_time[u] != ~0ull) { ev[nev].time = res.h_first_time[u]; ev[nev].seq = 0; ev[nev].u = u; ++nev; }
        qsort(ev, nev, sizeof(event), by_time);
        memset(ix_of_u, 0xFF, sizeof(uint32_t) * (U.n + 1));
        for (uint64_t i = 0; i < nev; ++i) if (ix_of_u[ev[i].u] == 0xFFFFFFFFu) ix_of_u[ev[i].u] = n_labels++;
        if (n_labels >= EMPTY) { rc = UTREE_E_UNSUPPORTED; goto done; }                 /* would collide with EMPTY_IX / BAD_IX */
    }
    st.n_labels = n_labels;
    /* ---- the files (UT_writeTreeBinary 1317-1343, UT_writeSamples 1225-1232) ---- */
    fd = open(ubt_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { rc = UTREE_E_IO; goto done; }
    {
        uint64_t md[4] = {W, 0, I, res.n_nodes};
        per_label = (uint64_t *)calloc(n_labels ? n_labels : 1, sizeof(uint64_t));
        if (!per_label) { rc = UTREE_E_NOMEM; goto done; }
        if (write_all(fd, md, 32)) { rc = UTREE_E_IO; goto done; }
        rc = utk_build_phase2(S, ix_of_u, U.n, n_labels, fd, per_label);
        if (rc) goto done;
        /* label lines in index order */
        uint32_t *u_of_ix = (uint32_t *)malloc(sizeof(uint32_t) * (n_labels ? n_labels : 1));
        if (!u_of_ix) { rc = UTREE_E_NOMEM; goto done; }
        for (uint32_t u = 0; u < U.n; ++u) if (ix_of_u[u] != 0xFFFFFFFFu) u_of_ix[ix_of_u[u]] = u;
        size_t cap = (size_t)U.blob_n + 32ull * n_labels + 64, o = 0;
        char *txt = (char *)malloc(cap);
        if (!txt) { free(u_of_ix); rc = UTREE_E_NOMEM; goto done; }
        for (uint32_t i = 0; i < n_labels; ++i)
            o += (size_t)sprintf(txt + o, "%s\t%llu\n", U.blob + U.off[u_of_ix[i]], (unsigned long long)per_label[i]);
        int werr = write_all(fd, txt, o);
        char *logp = (char *)malloc(strlen(ubt_path) + 16);
        if (logp) {
            sprintf(logp, "%s%s.log", ubt_path, gg ? ".gg" : "");                       /* itree.c:1405 */
            int lf = open(logp, O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (lf >= 0) { werr |= write_all(lf, txt, o); close(lf); }
            free(logp);
        }
        free(txt); free(u_of_ix);
        if (werr) { rc = UTREE_E_IO; goto done; }
    }
done:
    if (fd >= 0) close(fd);
    if (S) utk_build_free(S);
    free(res.h_first_time); free(res.h_ref_time);
    free(ev); free(ix_of_u); free(per_label);
    free(seq_off); free(seq_len); free(ref_u); free(trunc_off); free(trunc_ids);
    st_free(&U);
    free(ent); free(fa); free(mp);
    st.seconds = now_s() - t0;
    if (stats) *stats = st;
    return rc;
}
