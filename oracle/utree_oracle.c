/* utree_oracle.c -- CPU restatement of the UTree SEARCH_GG hot path (plain C, gcc).
 *
 * TEST INFRASTRUCTURE ONLY (see utree_oracle.h).  Parity status: PINNED against the genuine reference
 * via tests/golden/ (tests/test_oracle_golden.py).
 *
 * This is a restatement in our own words of what /root/reference/itree.c does when compiled with
 * -D SEARCH_GG.  It is written from the behaviour, not copied; citations give
 * the reference lines each piece follows.
 */
#define _FILE_OFFSET_BITS 64
#include "utree_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

struct orc_db {
    uint32_t W, I;           /* bytes of a packed k-mer word / of a label index (file header [0],[2]) */
    uint32_t k;              /* bases per k-mer = 4*W                                                  */
    uint32_t SZ;             /* bytes per stored record = W + I - 3            (itree.c:691)           */
    uint32_t sufbytes;       /* W - 3                                          (itree.c:692 CMPWDSZ)   */
    uint64_t n_nodes;
    uint64_t *binix;         /* 2^24+1 bin starts, zero-extended               (itree.c:756-759)       */
    uint8_t *recs;           /* n_nodes * SZ bytes (+ slack)                   (itree.c:766-767)       */
    uint32_t n_labels;       /* = maxIX                                         (itree.c:855)           */
    char **labels;           /* first-seen order, duplicates collapsed         (itree.c:191-220)       */
    char *label_blob;
};

static void set_err(char *err, size_t n, const char *msg) {
    if (err && n) { snprintf(err, n, "%s", msg); }
}

/* ---------------------------------------------------------------------------------------------
 * a1  base -> 2-bit code (itree.c:110-121).  Bytes >= 0x80 index the table out of range in the
 * reference (signed char); we define them as "bad" like every other non-ACGT byte.
 * ------------------------------------------------------------------------------------------- */
static inline int base_code(uint8_t c) {
    switch (c) {
        case 'a': case 'A': return 0;
        case 'c': case 'C': return 1;
        case 'g': case 'G': return 2;
        case 't': case 'T': return 3;
        default: return -1;
    }
}

/* ---------------------------------------------------------------------------------------------
 * Label table (itree.c:1154-1223, 191-220): each line up to its first TAB is a label; the index of
 * a label is the order of its first appearance; a repeated label maps to its first index.
 * ------------------------------------------------------------------------------------------- */
typedef struct { uint32_t *slot; uint32_t cap; } strset;
static uint64_t str_hash(const char *s, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= (uint8_t)s[i]; h *= 1099511628211ull; }
    return h;
}
static int parse_labels(orc_db *db, const char *text, size_t len) {
    db->label_blob = (char *)malloc(len + 1);
    if (!db->label_blob) return -1;
    memcpy(db->label_blob, text, len);
    db->label_blob[len] = 0;
    size_t nlines = 0;
    for (size_t i = 0; i < len; ++i) nlines += text[i] == '\n';
    nlines += 1;
    db->labels = (char **)malloc(sizeof(char *) * (nlines + 1));
    strset set; set.cap = 16; while (set.cap < 2 * nlines + 2) set.cap <<= 1;
    set.slot = (uint32_t *)malloc(sizeof(uint32_t) * set.cap);
    if (!db->labels || !set.slot) return -1;
    memset(set.slot, 0xFF, sizeof(uint32_t) * set.cap);
    uint32_t n = 0;
    size_t p = 0;
    while (p < len) {
        size_t e = p;
        while (e < len && db->label_blob[e] != '\n') ++e;       /* one fgets line (itree.c:1157)    */
        size_t t = p;
        while (t < e && db->label_blob[t] != '\t') ++t;         /* cut at first TAB (itree.c:1161)  */
        /* a line without a TAB runs off the buffer in the reference; we take the whole line */
        db->label_blob[t] = 0;
        const char *s = db->label_blob + p;
        size_t sl = strlen(s);
        uint64_t h = str_hash(s, sl) & (set.cap - 1);
        for (;;) {
            uint32_t v = set.slot[h];
            if (v == 0xFFFFFFFFu) { set.slot[h] = n; db->labels[n++] = db->label_blob + p; break; }
            if (!strcmp(db->labels[v], s)) break;               /* duplicate -> first index         */
            h = (h + 1) & (set.cap - 1);
        }
        p = e + 1;
    }
    free(set.slot);
    db->n_labels = n;
    return 0;
}

static orc_db *db_alloc(uint32_t W, uint32_t I, uint64_t n_nodes, char *err, size_t errlen) {
    if (!(W == 4 || W == 8 || W == 16) || !(I == 2 || I == 4)) {
        set_err(err, errlen, "unsupported W/I"); return NULL;
    }
    orc_db *db = (orc_db *)calloc(1, sizeof *db);
    if (!db) return NULL;
    db->W = W; db->I = I; db->k = 4 * W; db->SZ = W + I - 3; db->sufbytes = W - 3; db->n_nodes = n_nodes;
    db->binix = (uint64_t *)calloc(ORC_NUMBINS, sizeof(uint64_t));
    db->recs = (uint8_t *)calloc(n_nodes * db->SZ + 32, 1);      /* slack as itree.c:766             */
    if (!db->binix || !db->recs) { orc_db_free(db); set_err(err, errlen, "out of memory"); return NULL; }
    return db;
}

orc_db *orc_db_load(const char *path, char *err, size_t errlen) {
    FILE *fp = fopen(path, "rb");
    if (!fp) { set_err(err, errlen, "Invalid DB file"); return NULL; }          /* itree.c:735 */
    uint64_t meta[4] = {0, 0, 0, 0};
    if (fread(meta, 8, 4, fp) < 4 || !meta[3]) {                                 /* itree.c:738 */
        fclose(fp); set_err(err, errlen, "Tree malformatted."); return NULL;
    }
    if (meta[1] != 0) { fclose(fp); set_err(err, errlen, "count field not supported"); return NULL; }
    orc_db *db = db_alloc((uint32_t)meta[0], (uint32_t)meta[2], meta[3], err, errlen);
    if (!db) { fclose(fp); return NULL; }
    /* bin starts: 4-byte entries iff N < UINT32_MAX, else 8 (itree.c:757-759) */
    int ixsz = db->n_nodes < 0xFFFFFFFFull ? 4 : 8;
    uint8_t *raw = (uint8_t *)malloc((size_t)ORC_NUMBINS * ixsz);
    if (!raw || fread(raw, ixsz, ORC_NUMBINS, fp) != ORC_NUMBINS) {
        free(raw); fclose(fp); orc_db_free(db); set_err(err, errlen, "short bin table"); return NULL;
    }
    for (size_t i = 0; i < ORC_NUMBINS; ++i) {
        if (ixsz == 4) { uint32_t v; memcpy(&v, raw + 4 * i, 4); db->binix[i] = v; }
        else memcpy(&db->binix[i], raw + 8 * i, 8);
    }
    free(raw);
    if (fread(db->recs, db->SZ, db->n_nodes, fp) != db->n_nodes) {               /* itree.c:767-768 */
        fclose(fp); orc_db_free(db); set_err(err, errlen, "Error in reading tree."); return NULL;
    }
    /* the rest of the file is label text (itree.c:775) */
    size_t cap = 1 << 16, len = 0;
    char *text = (char *)malloc(cap);
    for (;;) {
        if (len == cap) { cap *= 2; text = (char *)realloc(text, cap); }
        size_t r = fread(text + len, 1, cap - len, fp);
        if (!r) break;
        len += r;
    }
    fclose(fp);
    int rc = parse_labels(db, text, len);
    free(text);
    if (rc || !db->n_labels) { orc_db_free(db); set_err(err, errlen, "No annotation found in tree file."); return NULL; }
    return db;
}

orc_db *orc_db_from_memory(uint32_t W, uint32_t I, uint64_t n_nodes, const uint64_t *binix,
                           const uint8_t *records, const char *label_text, size_t label_len,
                           char *err, size_t errlen) {
    orc_db *db = db_alloc(W, I, n_nodes, err, errlen);
    if (!db) return NULL;
    memcpy(db->binix, binix, sizeof(uint64_t) * ORC_NUMBINS);
    memcpy(db->recs, records, n_nodes * db->SZ);
    if (parse_labels(db, label_text, label_len) || !db->n_labels) {
        orc_db_free(db); set_err(err, errlen, "No annotation found in tree file."); return NULL;
    }
    return db;
}

void orc_db_free(orc_db *db) {
    if (!db) return;
    free(db->binix); free(db->recs); free(db->labels); free(db->label_blob); free(db);
}
uint32_t orc_db_W(const orc_db *db) { return db->W; }
uint32_t orc_db_I(const orc_db *db) { return db->I; }
uint64_t orc_db_nodes(const orc_db *db) { return db->n_nodes; }
uint32_t orc_db_labels(const orc_db *db) { return db->n_labels; }
const char *orc_db_label(const orc_db *db, uint32_t ix) { return ix < db->n_labels ? db->labels[ix] : NULL; }

/* ---------------------------------------------------------------------------------------------
 * a4  k-mer windows (itree.c:903-927).  Windows are visited in order of their last base i = k-1 ..
 * len-1.  A window is looked up iff none of its k bases is "bad"; the word is the base-4 number of
 * its bases, first base most significant, in exactly 2k bits.  (The reference reaches the same set
 * by rolling `w` and, on a bad base at b, jumping to the window that starts at b+1: itree.c:923.)
 * ------------------------------------------------------------------------------------------- */
typedef void (*window_fn)(void *ctx, size_t end_pos, u128 word);

static void for_each_window(const uint8_t *seq, size_t len, uint32_t k, window_fn fn, void *ctx) {
    u128 w = 0;
    const u128 keep = (k == 64) ? ~(u128)0 : (((u128)1 << (2 * k)) - 1);
    size_t good = 0;                      /* length of the current run of good bases ending at i */
    for (size_t i = 0; i < len; ++i) {
        int c = base_code(seq[i]);
        if (c < 0) { good = 0; w = 0; continue; }
        w = ((w << 2) | (u128)c) & keep;  /* shifting discards the oldest base (itree.c:924)     */
        if (++good >= k) fn(ctx, i, w);
    }
}

typedef struct { uint32_t *end_pos; uint64_t *hi, *lo; size_t n, cap; } win_sink;
static void win_collect(void *ctx, size_t end_pos, u128 word) {
    win_sink *s = (win_sink *)ctx;
    if (s->n < s->cap) {
        if (s->end_pos) s->end_pos[s->n] = (uint32_t)end_pos;
        if (s->hi) s->hi[s->n] = (uint64_t)(word >> 64);
        if (s->lo) s->lo[s->n] = (uint64_t)word;
    }
    s->n++;
}
size_t orc_windows(const uint8_t *seq, size_t len, int k, uint32_t *end_pos, uint64_t *hi, uint64_t *lo,
                   size_t cap) {
    win_sink s = {end_pos, hi, lo, 0, cap};
    for_each_window(seq, len, (uint32_t)k, win_collect, &s);
    return s.n;
}

/* ---------------------------------------------------------------------------------------------
 * a5  node lookup (itree.c:720-730, 699-707, 674-686).
 * prefix = top 24 bits of the 2k-bit word; suffix = low 8*(W-3) bits.  Bin [s,e) from the bin table.
 * Search: p = record s; over the remaining e-s-1 records, repeatedly probe the record w+1 past p
 * (w = size/2): if its suffix <= query, move p there and drop w+1 records, else keep the first w.
 * Finally p matches iff its suffix equals the query.  (With an ascending bin this is "last record
 * <= query".  We keep the exact probe order so that malformed / first-bin-quirk bins agree too.)
 * ------------------------------------------------------------------------------------------- */
static inline u128 rec_suffix(const orc_db *db, const uint8_t *rec) {
    u128 v = 0;
    memcpy(&v, rec, db->sufbytes);        /* little-endian low bytes (itree.c:676 & MASK)        */
    return v;
}
static inline uint32_t rec_ix(const orc_db *db, const uint8_t *rec) {
    uint32_t v = 0;
    memcpy(&v, rec + db->sufbytes, db->I); /* itree.c:681 */
    return v;
}
static inline uint32_t lookup_word(const orc_db *db, u128 word) {
    const unsigned sxbits = 2 * db->k - 24;                       /* itree.c:694 */
    uint32_t prefix = (uint32_t)(word >> sxbits);
    u128 sx = word & ((((u128)1) << sxbits) - 1);
    uint64_t s = db->binix[prefix], e = db->binix[prefix + 1];    /* itree.c:724 */
    if (s >= e) return ORC_BAD_IX;                                 /* itree.c:726 */
    const uint8_t *p = db->recs + (size_t)db->SZ * s;
    uint64_t size = e - s - 1;
    while (size) {                                                 /* itree.c:701-705 */
        uint64_t w = size >> 1;
        const uint8_t *probe = p + (size_t)db->SZ * (w + 1);
        if (rec_suffix(db, probe) <= sx) { p = probe; size -= w + 1; }
        else size = w;
    }
    if (rec_suffix(db, p) != sx) return ORC_BAD_IX;                /* itree.c:706 */
    uint32_t ix = rec_ix(db, p);
    if (db->I == 2 && ix == 0xFFFFu) return ORC_BAD_IX;            /* stored BAD_IX (itree.c:105) */
    return ix;
}
uint32_t orc_lookup(const orc_db *db, uint64_t hi, uint64_t lo) {
    return lookup_word(db, ((u128)hi << 64) | lo);
}

/* ---------------------------------------------------------------------------------------------
 * a7-a9  tally, sort, vote (itree.c:1028-1088).
 * ------------------------------------------------------------------------------------------- */
typedef struct { const char *s; uint32_t n; uint32_t ix; } tax_cnt;
static int by_label(const void *a, const void *b) {               /* itree.c:830-832 */
    return strcmp(((const tax_cnt *)a)->s, ((const tax_cnt *)b)->s);
}
static inline uint32_t cut_of(uint32_t x) {                        /* itree.c:1044,1046 */
    uint32_t c = x - x / 4;
    c += (x >> 1) >= c;
    return c;
}

typedef struct { uint32_t *hist; tax_cnt *tc; } vote_scratch;

static void vote_with(const orc_db *db, const uint32_t *hits, uint32_t F, vote_scratch *sc, orc_result *r) {
    r->found = F; r->uix = 0; r->sl = 0; r->ol = 0; r->cut = -2; r->label = 0;
    if (!F) return;                                                /* itree.c:1028 */
    r->label = hits[0];
    if (F == 1) { r->uix = 1; return; }                            /* itree.c:1031-1032 */
    for (uint32_t i = 0; i < F; ++i) ++sc->hist[hits[i]];           /* itree.c:1033-1034 */
    uint32_t uix = 0;
    for (uint32_t i = F; i; --i) {                                  /* itree.c:1036-1038 */
        uint32_t t = hits[i - 1];
        if (sc->hist[t]) { sc->tc[uix].s = db->labels[t]; sc->tc[uix].n = sc->hist[t]; sc->tc[uix].ix = t; ++uix; sc->hist[t] = 0; }
    }
    r->uix = uix;
    if (uix == 1) return;                                           /* itree.c:1039-1040 */
    tax_cnt *T = sc->tc;
    qsort(T, uix, sizeof *T, by_label);                             /* itree.c:1041 */

    /* Greedy descent.  All state is 32-bit unsigned with wrap-around, as in the reference. */
    uint32_t cutoff = cut_of(F);
    uint32_t st = 0, ed = uix, dv = 0xFFFFFFFFu /* "-1": nothing agreed yet */, orun = F, sl, ol;
    for (;;) {
        uint32_t run = T[st].n, td = dv;
        for (uint32_t z = st + 1; z < ed; ++z) {
            const char *s1 = T[z - 1].s, *s2 = T[z].s;
            uint32_t probe = dv + (dv == 0xFFFFFFFFu);               /* 0 when nothing agreed     */
            int set_aside = 0;
            if (!s1[probe]) set_aside = 1;                           /* prev exhausted: 1052      */
            else {
                for (td = dv + 1; s1[td] && s1[td] == s2[td]; ++td)  /* itree.c:1060-1061         */
                    if (s1[td] == ';') break;
                if (s1[td] == s2[td]) { run += T[z].n; continue; }    /* same token: 1062          */
                /* s1[td-1] with td==0 reads before the string in the reference; treat as not '_' */
                char before = td ? s1[td - 1] : 0;
                if ((!s1[td] && s2[td] == ';') || ((s1[td] == ';' || !s1[td]) && before == '_'))
                    set_aside = 1;                                   /* less specific: 1063       */
                else if (run >= cutoff) { ed = z; break; }           /* group wins: 1068          */
                else { run = T[z].n; st = z; continue; }             /* restart: 1069             */
            }
            if (set_aside) {                                         /* itree.c:1053-1056,1064-1067 */
                run = T[z].n; st = z;
                orun -= T[z - 1].n;
                cutoff = cut_of(orun);
            }
        }
        sl = run; ol = orun;                                         /* itree.c:1071 */
        if (run < cutoff) break;                                     /* itree.c:1072 */
        if (st + 1 >= ed) {                                          /* itree.c:1073-1079 */
            if (T[ed - 1].n >= cutoff) dv = 0xFFFFFFFEu;             /* "-2": whole label         */
            break;
        }
        orun = run; dv = td; cutoff = cut_of(run);                   /* itree.c:1082-1085 */
    }
    r->sl = sl; r->ol = ol;
    r->label = T[ed - 1].ix;
    if (dv == 0xFFFFFFFFu) r->cut = -1;                              /* itree.c:1087 */
    else if (dv == 0xFFFFFFFEu) r->cut = -2;
    else {
        /* first dv bytes, but printing stops at the label's NUL (memcpy + %s, itree.c:1088,1096) */
        size_t L = strlen(T[ed - 1].s);
        r->cut = (int32_t)(dv < L ? dv : L);
    }
}

void orc_vote(const orc_db *db, const uint32_t *hits, uint32_t nhits, orc_result *res) {
    vote_scratch sc;
    sc.hist = (uint32_t *)calloc(db->n_labels, sizeof(uint32_t));
    sc.tc = (tax_cnt *)malloc(sizeof(tax_cnt) * (db->n_labels + 1));
    vote_with(db, hits, nhits, &sc, res);
    free(sc.hist); free(sc.tc);
}

/* a10  one output line (itree.c:1032, 1040, 1096). */
size_t orc_format(const orc_db *db, const char *name, size_t name_len, const orc_result *r, char *out,
                  size_t cap) {
    if (!r->found) return 0;
    const char *lab = db->labels[r->label];
    size_t lablen = r->cut == -1 ? 0 : r->cut == -2 ? strlen(lab) : (size_t)r->cut;
    size_t need = name_len + 1 + lablen + 64;
    if (need > cap) return 0;
    char *o = out;
    memcpy(o, name, name_len); o += name_len; *o++ = '\t';
    memcpy(o, lab, lablen); o += lablen;
    if (r->uix == 1) o += sprintf(o, "\t%u\t1\t*\n", r->found);
    else o += sprintf(o, "\t%u\t%u\t%u;%u\n", r->found, r->uix, r->sl, r->ol);
    return (size_t)(o - out);
}

/* a3  reverse complement (itree.c:838-841, 891-898): fwd, 'N', then complement of fwd reversed;
 * anything that is not ACGTacgt complements to 'N'. */
void orc_revcomp_append(const uint8_t *src, size_t len, uint8_t *dst) {
    memmove(dst, src, len);
    dst[len] = 'N';
    for (size_t j = 0; j < len; ++j) {
        uint8_t c = src[len - 1 - j], o;
        switch (c) {
            case 'A': case 'a': o = 'T'; break;
            case 'C': case 'c': o = 'G'; break;
            case 'G': case 'g': o = 'C'; break;
            case 'T': case 't': o = 'A'; break;
            default: o = 'N';
        }
        dst[len + 1 + j] = o;
    }
}

/* per-thread working set (itree.c:1012-1016) */
typedef struct {
    vote_scratch sc;
    uint32_t *hits; size_t hits_cap;
    uint8_t *rcbuf; size_t rc_cap;
    const orc_db *db;
    size_t nhits;
} worker;

static int worker_init(worker *w, const orc_db *db) {
    memset(w, 0, sizeof *w);
    w->db = db;
    w->sc.hist = (uint32_t *)calloc(db->n_labels, sizeof(uint32_t));
    w->sc.tc = (tax_cnt *)malloc(sizeof(tax_cnt) * (db->n_labels + 1));
    return (w->sc.hist && w->sc.tc) ? 0 : -1;
}
static void worker_free(worker *w) { free(w->sc.hist); free(w->sc.tc); free(w->hits); free(w->rcbuf); }

static void hit_sink(void *ctx, size_t end_pos, u128 word) {
    (void)end_pos;
    worker *w = (worker *)ctx;
    uint32_t ix = lookup_word(w->db, word);
    if (ix < w->db->n_labels) w->hits[w->nhits++] = ix;              /* itree.c:929-931, 935 */
}

static void classify_one(worker *w, const uint8_t *seq, size_t len, int do_rc, orc_result *res) {
    const orc_db *db = w->db;
    if (do_rc) {                                                     /* itree.c:891-898 */
        size_t need = 2 * len + 1;
        if (need > w->rc_cap) { free(w->rcbuf); w->rcbuf = (uint8_t *)malloc(need + 64); w->rc_cap = need + 64; }
        orc_revcomp_append(seq, len, w->rcbuf);
        seq = w->rcbuf; len = need;
    }
    if (len + 1 > w->hits_cap) { free(w->hits); w->hits = (uint32_t *)malloc(sizeof(uint32_t) * (len + 64)); w->hits_cap = len + 64; }
    w->nhits = 0;
    for_each_window(seq, len, db->k, hit_sink, w);
    vote_with(db, w->hits, (uint32_t)w->nhits, &w->sc, res);
}

void orc_classify_read(const orc_db *db, const uint8_t *seq, size_t len, int do_rc, orc_result *res) {
    worker w;
    if (worker_init(&w, db)) { memset(res, 0, sizeof *res); return; }
    classify_one(&w, seq, len, do_rc, res);
    worker_free(&w);
}

void orc_classify_batch(const orc_db *db, const uint8_t *buf, const uint64_t *off, const uint32_t *len,
                        size_t n, int do_rc, int threads, orc_result *out) {
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
#pragma omp parallel num_threads(threads)
    {
        worker w;
        int ok = !worker_init(&w, db);
#pragma omp for schedule(dynamic, 256)
        for (size_t i = 0; i < n; ++i) {
            if (ok) classify_one(&w, buf + off[i], len[i], do_rc, &out[i]);
            else memset(&out[i], 0, sizeof out[i]);
        }
        worker_free(&w);
    }
}

/* ---------------------------------------------------------------------------------------------
 * a2 + driver (itree.c:860-901, 1008-1107).  Reads are framed exactly as the reference frames them
 * with fgets(…, 16 MiB): header line, sequence line.  Output lines are written in input order, which
 * is what the reference produces with one thread.
 * ------------------------------------------------------------------------------------------- */
#define ORC_LINELEN 16777216u   /* itree.c:836 */

/* next "fgets line": at most LINELEN-1 bytes, ends after '\n' */
static size_t next_line(const uint8_t *buf, size_t n, size_t pos) {
    size_t lim = n - pos < ORC_LINELEN - 1 ? n - pos : ORC_LINELEN - 1;
    const uint8_t *nl = (const uint8_t *)memchr(buf + pos, '\n', lim);
    return nl ? (size_t)(nl - (buf + pos)) + 1 : lim;
}

typedef struct { size_t name, name_len, seq, seq_len; } frame;

/* Reads the whole file and frames it (itree.c:866-890).  Returns the reference's exit code for the first
 * malformed record (reads before it are still processed, as the reference has already printed them). */
static int load_and_frame(const char *fasta, uint8_t **pbuf, frame **pfr, size_t *pnr, char *err, size_t errlen) {
    FILE *fp = fopen(fasta, "rb");
    if (!fp) { set_err(err, errlen, "Invalid input files"); return 1; }                       /* itree.c:835 */
    fseeko(fp, 0, SEEK_END);
    size_t n = (size_t)ftello(fp);
    fseeko(fp, 0, SEEK_SET);
    uint8_t *buf = (uint8_t *)malloc(n + 1);
    if (!buf || fread(buf, 1, n, fp) != n) { fclose(fp); free(buf); set_err(err, errlen, "read error"); return 3; }
    fclose(fp);
    buf[n] = 0;
    size_t cap = 1024, nr = 0;
    frame *fr = (frame *)malloc(sizeof(frame) * cap);
    int rc = 0;
    size_t pos = 0;
    while (pos < n) {
        size_t hl = next_line(buf, n, pos);
        size_t spos = pos + hl;
        if (spos >= n) { snprintf(err, errlen, "ERROR: can't read sequence L %zu", nr); rc = 2; break; }   /* 872 */
        size_t sl_ = next_line(buf, n, spos);
        if (buf[pos] != '>') { snprintf(err, errlen, "ERROR: no header '>' [L %zu]", nr + 1); rc = 2; break; } /* 880 */
        /* name: after '>' up to first NUL, space or newline (itree.c:881) */
        size_t e = pos + 1;
        while (e < pos + hl && buf[e] && buf[e] != ' ' && buf[e] != '\n') ++e;
        if (buf[spos] == '>') { snprintf(err, errlen, "ERROR: sequence begins '>' [L %zu]", nr + 1); rc = 2; break; } /* 886 */
        /* strlen stops at an embedded NUL (itree.c:887) */
        const uint8_t *z = (const uint8_t *)memchr(buf + spos, 0, sl_);
        size_t length = z ? (size_t)(z - (buf + spos)) : sl_;
        if (!length) { snprintf(err, errlen, "ERROR: empty query line %zu", nr + 1); rc = 2; break; }      /* 888 */
        if (buf[spos + length - 1] == '\n') --length;                                                     /* 889 */
        if (length && buf[spos + length - 1] == '\r') --length;                                           /* 890 */
        if (nr == cap) { cap *= 2; fr = (frame *)realloc(fr, sizeof(frame) * cap); }
        fr[nr].name = pos + 1; fr[nr].name_len = e - (pos + 1); fr[nr].seq = spos; fr[nr].seq_len = length;
        ++nr;
        pos = spos + sl_;
    }
    *pbuf = buf; *pfr = fr; *pnr = nr;
    return rc;
}

int orc_search_file(const orc_db *db, const char *fasta, const char *outp, int threads, int do_rc,
                    uint64_t *n_reads, uint64_t *good_finds, char *err, size_t errlen) {
    uint8_t *buf = NULL; frame *fr = NULL; size_t nr = 0;
    FILE *fo = fopen(outp, "wb");
    int rc = load_and_frame(fasta, &buf, &fr, &nr, err, errlen);
    if (!buf) { if (fo) fclose(fo); return rc; }
    if (!fo) { free(buf); free(fr); set_err(err, errlen, "Invalid output file"); return 1; }
    /* pass 2: classify (parallel), pass 3: write in order */
    orc_result *res = (orc_result *)calloc(nr ? nr : 1, sizeof(orc_result));
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
#pragma omp parallel num_threads(threads)
    {
        worker w;
        int ok = !worker_init(&w, db);
#pragma omp for schedule(dynamic, 64)
        for (size_t i = 0; i < nr; ++i)
            if (ok) classify_one(&w, buf + fr[i].seq, fr[i].seq_len, do_rc, &res[i]);
        worker_free(&w);
    }
    uint64_t good = 0;
    size_t lcap = 1 << 20;
    char *line = (char *)malloc(lcap);
    for (size_t i = 0; i < nr; ++i) {
        if (!res[i].found) continue;
        ++good;
        size_t need = fr[i].name_len + 70000 + 64;
        if (need > lcap) { lcap = need * 2; line = (char *)realloc(line, lcap); }
        size_t L = orc_format(db, (const char *)buf + fr[i].name, fr[i].name_len, &res[i], line, lcap);
        fwrite(line, 1, L, fo);
    }
    fclose(fo);
    free(line); free(res); free(fr); free(buf);
    if (n_reads) *n_reads = nr;
    if (good_finds) *good_finds = good;
    return rc;
}

/* =============================================================================================
 * Rank-specific search: `xtree-search` (itree.c -D SEARCH; main passes doCollapse = 0, itree.c:1376).
 * Same framing, windows and lookups; the differences are restated below.  The reference runs this
 * branch on ONE thread (no omp pragma in it, itree.c:969-1007), and reads are NOT independent:
 *
 *  - hit selection (XT_SHALLOWVOTE, itree.c:948-951): after a hit at the window ending at i the next
 *    window examined ends at i + PACKSIZE/SPARSITY (the macro adds PACKSIZE/SPARSITY-1, the loop 1);
 *    windows in between are never looked up, and the roller's word register is left in a state that
 *    is NOT the query's k-mer for the next PACKSIZE - PACKSIZE/SPARSITY windows (see rank_hits).
 *  - the vote (980-1003) runs over kingsMen+1 entries of AllTheKingsHorses: `if (!kingsMen++)` (982)
 *    is never true when foundUniq > 0, and its post-increment makes the loops at 984 and 988 read ONE
 *    entry past the read's own hits.  That entry is whatever an earlier read with more hits left at
 *    that index -- the array is allocated once (970) and never cleared -- or 0 while untouched
 *    (a 64 MiB malloc is served by fresh zero pages; [probed] with the genuine binary).
 *    So the output of a read depends on the reads before it; rank_state carries that array.
 *  - most / secondMost (986-997): first-come maximum of the per-label counts and the runner-up;
 *    a line is printed iff most >= TOLERANCE_THRESHOLD and most >= SLACK*secondMost (1000), as
 *    name \t label \t %f \t %d with 1 - secondMost/most and most (1002).
 * ============================================================================================= */
struct orc_rank_state {
    uint32_t *horses;  size_t cap;      /* AllTheKingsHorses (itree.c:970), persistent                */
    uint32_t *hashes;                   /* Hashes (971)                                              */
    uint8_t *rcbuf; size_t rc_cap;
};

orc_rank_state *orc_rank_state_new(const orc_db *db) {
    orc_rank_state *st = (orc_rank_state *)calloc(1, sizeof *st);
    if (!st) return NULL;
    st->cap = 1024;
    st->horses = (uint32_t *)calloc(st->cap, sizeof(uint32_t));
    st->hashes = (uint32_t *)calloc(db->n_labels ? db->n_labels : 1, sizeof(uint32_t));
    if (!st->horses || !st->hashes) { orc_rank_state_free(st); return NULL; }
    return st;
}
void orc_rank_state_free(orc_rank_state *st) {
    if (!st) return;
    free(st->horses); free(st->hashes); free(st->rcbuf); free(st);
}

/* Hit selection with the reference's word register (itree.c:906-933 with XT_SHALLOWVOTE, 948-951).
 * Let S = PACKSIZE/SPARSITY.  After a hit at the window ending at i0 whose looked-up word was w0, the next
 * window examined ends at i0+S.  For S < PACKSIZE-1 the roller takes its "continue from z" path (920): it
 * first shifts the register by (i-z-1) = S-1 bases and then shifts in ALL S bases z+1..i, so the register
 * ends up shifted by 2S-1 bases with only S new ones: S-1 zero ("A") positions are left in the middle and
 * the oldest bases are not yet gone.  That word -- not the query's real k-mer -- is what gets looked up,
 * and the register rolls on from it one base at a time, so every window ending at i0+d, S <= d < PACKSIZE,
 * is looked up as   (w0 << 2(d+S-1)) | (the d bases i0+1..i0+d)   in 2*PACKSIZE-bit arithmetic; from
 * d = PACKSIZE on the real k-mer is back.  A hit on such a word starts the same thing over from it.
 * A bad base clears everything (923), exactly as in for_each_window. */
static size_t rank_hits(const orc_db *db, const uint8_t *seq, size_t len, uint32_t sparsity, uint32_t *hits) {
    const uint32_t k = db->k;
    const u128 keep = (k == 64) ? ~(u128)0 : (((u128)1 << (2 * k)) - 1);
    const size_t S = k / sparsity;
    u128 real = 0, w0 = 0;                /* the query's real k-mer register; the word of the last hit  */
    size_t good = 0, i0 = 0, n = 0;
    int after_hit = 0;
    for (size_t i = 0; i < len; ++i) {
        int c = base_code(seq[i]);
        if (c < 0) { good = 0; real = 0; after_hit = 0; continue; }
        real = ((real << 2) | (u128)c) & keep;
        if (++good < k) continue;
        u128 word = real;
        if (after_hit) {
            size_t d = i - i0;
            if (d < S) continue;                                       /* not examined (950) */
            if (d < k) {
                size_t sh = 2 * (d + S - 1);
                u128 newbits = real & ((((u128)1) << (2 * d)) - 1);
                word = ((sh >= 2 * (size_t)k ? (u128)0 : (w0 << sh)) | newbits) & keep;
            }
        }
        uint32_t ix = lookup_word(db, word);
        if (ix >= db->n_labels) continue;                              /* itree.c:929 */
        hits[n++] = ix;                                                /* 951 */
        after_hit = 1; i0 = i; w0 = word;
    }
    return n;
}

void orc_rank_read(const orc_db *db, orc_rank_state *st, const uint8_t *seq, size_t len, int do_rc,
                   const orc_rank_params *prm, orc_rank_result *res) {
    memset(res, 0, sizeof *res);
    if (do_rc) {
        size_t need = 2 * len + 1;
        if (need > st->rc_cap) { free(st->rcbuf); st->rcbuf = (uint8_t *)malloc(need + 64); st->rc_cap = need + 64; }
        orc_revcomp_append(seq, len, st->rcbuf);
        seq = st->rcbuf; len = need;
    }
    if (len + 2 > st->cap) {                                           /* the reference's array is simply huge */
        size_t nc = len + 1024;
        uint32_t *h = (uint32_t *)realloc(st->horses, nc * sizeof(uint32_t));
        if (!h) return;
        memset(h + st->cap, 0, (nc - st->cap) * sizeof(uint32_t));
        st->horses = h; st->cap = nc;
    }
    const size_t nh = rank_hits(db, seq, len, prm->sparsity, st->horses);
    res->found = (uint32_t)nh;
    if (!nh) return;                                                   /* itree.c:980 */
    uint32_t *A = st->horses, *H = st->hashes;
    size_t km = nh + 1;                                                /* 982: kingsMen++ */
    for (size_t i = 0; i < km; ++i) ++H[A[i]];                         /* 984-985 */
    uint32_t most = 0, second = 0, most_ix = 0;
    for (size_t i = 0; i < km; ++i) {                                  /* 988-997 */
        uint32_t c = H[A[i]];
        if (c > most) { second = most; most_ix = A[i]; most = c; }
        else if (c > second) second = c;
        H[A[i]] = 0;
    }
    res->label = most_ix; res->most = most; res->second = second;
    res->printed = !((int)most < (int)prm->tolerance || (int)most < (int)(prm->slack * second));   /* 1000 */
}

size_t orc_rank_format(const orc_db *db, const char *name, size_t name_len, const orc_rank_result *r, char *out,
                       size_t cap) {
    if (!r->found || !r->printed) return 0;
    const char *lab = db->labels[r->label];
    size_t lablen = strlen(lab);
    if (name_len + lablen + 96 > cap) return 0;
    char *o = out;
    memcpy(o, name, name_len); o += name_len; *o++ = '\t';
    memcpy(o, lab, lablen); o += lablen;
    o += sprintf(o, "\t%f\t%d\n", (double)1 - (double)r->second / r->most, (int)r->most);       /* 1002 */
    return (size_t)(o - out);
}

int orc_rank_search_file(const orc_db *db, const char *fasta, const char *outp, int do_rc,
                         const orc_rank_params *prm, uint64_t *n_reads, uint64_t *good_finds, char *err,
                         size_t errlen) {
    uint8_t *buf = NULL; frame *fr = NULL; size_t nr = 0;
    FILE *fo = fopen(outp, "wb");
    int rc = load_and_frame(fasta, &buf, &fr, &nr, err, errlen);
    if (!buf) { if (fo) fclose(fo); return rc; }
    if (!fo) { free(buf); free(fr); set_err(err, errlen, "Invalid output file"); return 1; }
    orc_rank_state *st = orc_rank_state_new(db);
    uint64_t good = 0;
    size_t lcap = 1 << 20;
    char *line = (char *)malloc(lcap);
    for (size_t i = 0; st && i < nr; ++i) {
        orc_rank_result r;
        orc_rank_read(db, st, buf + fr[i].seq, fr[i].seq_len, do_rc, prm, &r);
        if (!r.found || !r.printed) continue;                          /* 981, 1000: ++goodFinds / --goodFinds */
        ++good;
        size_t need = fr[i].name_len + 70000 + 96;
        if (need > lcap) { lcap = need * 2; line = (char *)realloc(line, lcap); }
        size_t L = orc_rank_format(db, (const char *)buf + fr[i].name, fr[i].name_len, &r, line, lcap);
        fwrite(line, 1, L, fo);
    }
    fclose(fo);
    orc_rank_state_free(st);
    free(line); free(fr); free(buf);
    if (n_reads) *n_reads = nr;
    if (good_finds) *good_finds = good;
    return rc;
}
