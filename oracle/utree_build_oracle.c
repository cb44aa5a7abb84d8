/* utree_build_oracle.c -- CPU restatement of the reference's database BUILD (`utree-build`, `utree-buildGG`):
 * itree.c -D BUILD / -D BUILD_GG, main 1379-1407 -> UT_parseSampFastaExternOSFA (501-635) -> UT_addWordIx /
 * UT_addWordIxRF (437-473) -> xeTreeU / xeTreeU_RF (242-307) -> UT_writeTreeBinary (1317-1343) + UT_writeSamples
 * (1225-1232).  SURVEY.md §8(f) rank 3.
 *
 * TEST INFRASTRUCTURE ONLY (see utree_oracle.h).  Pinned by tests/test_oracle_golden.py against `.ubt` files the
 * genuine binaries wrote (tests/golden/make_golden.py build).
 *
 * The reference keeps one pointer BST per 24-bit prefix and walks the references one k-mer at a time; what ends up in
 * the `.ubt` depends only on, per distinct k-mer, the ORDER of the labels it was seen with -- the tree shape and the
 * rebalancing (349-386, 419-431) never show.  This restatement keeps a hash map k-mer -> label index and replays the
 * same sequence of events:
 *   - a reference's label gets its index the first time a reference with that label is parsed (addSampleU, 597);
 *   - BUILD: a k-mer seen with two different labels becomes BAD for good (xeTreeU, 262-266);
 *   - BUILD_GG: such a k-mer is relabelled to what its current label and the new one share, cut before the LAST ';'
 *     they have in common; fewer than critical_cutoff = 2 (74) common ';' -> BAD.  The cut label is looked up / created
 *     in the same label table (addSampleUd, 297), so labels created by collisions are numbered in the order the
 *     collisions happen, interleaved with the references' own labels.  Every further collision cuts again: a k-mer
 *     shared by many references loses one rank per extra occurrence whose label differs from its current one.
 *   - the dump is in-order per prefix = ascending k-mers, BAD ones left out (399-417); label lines carry the number of
 *     nodes per label (1324, 1336).
 */
#define _FILE_OFFSET_BITS 64
#define _GNU_SOURCE
#include "utree_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

static void set_err(char *err, size_t n, const char *msg) { if (err && n) snprintf(err, n, "%s", msg); }

static uint8_t *slurp(const char *path, size_t *n) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseeko(f, 0, SEEK_END);
    *n = (size_t)ftello(f);
    fseeko(f, 0, SEEK_SET);
    uint8_t *b = (uint8_t *)malloc(*n + 2);
    if (b && *n && fread(b, 1, *n, f) != *n) { free(b); b = NULL; }
    fclose(f);
    if (b) { b[*n] = 0; b[*n + 1] = 0; }
    return b;
}

/* ---- label table: strings -> indices in order of creation (ADDSAMP, itree.c:186-219) ---- */
typedef struct { char **str; uint32_t n, cap; uint32_t *slot; uint32_t nslot; } labtab;
static uint64_t hstr(const char *s) { uint64_t h = 1469598103934665603ull; while (*s) { h ^= (uint8_t)*s++; h *= 1099511628211ull; } return h; }
static void lt_grow(labtab *t) {
    uint32_t ns = t->nslot ? t->nslot * 2 : 1024;
    uint32_t *sl = (uint32_t *)malloc(sizeof(uint32_t) * ns);
    memset(sl, 0xFF, sizeof(uint32_t) * ns);
    for (uint32_t i = 0; i < t->n; ++i) { uint64_t h = hstr(t->str[i]) & (ns - 1); while (sl[h] != 0xFFFFFFFFu) h = (h + 1) & (ns - 1); sl[h] = i; }
    free(t->slot); t->slot = sl; t->nslot = ns;
}
static uint32_t lt_intern(labtab *t, const char *s) {
    if ((uint64_t)t->n * 2 >= t->nslot) lt_grow(t);
    uint64_t h = hstr(s) & (t->nslot - 1);
    while (t->slot[h] != 0xFFFFFFFFu) { if (!strcmp(t->str[t->slot[h]], s)) return t->slot[h]; h = (h + 1) & (t->nslot - 1); }
    if (t->n == t->cap) { t->cap = t->cap ? t->cap * 2 : 256; t->str = (char **)realloc(t->str, sizeof(char *) * t->cap); }
    t->str[t->n] = strdup(s);
    t->slot[h] = t->n;
    return t->n++;
}

/* ---- k-mer map: word -> label index / BAD ---- */
typedef struct { u128 *key; uint32_t *val; uint8_t *used; uint64_t n, cap; } kmap;
static uint64_t hword(u128 w) { uint64_t x = (uint64_t)w ^ ((uint64_t)(w >> 64) * 0x9E3779B97F4A7C15ull); x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; return x; }
static void km_init(kmap *m, uint64_t cap) { m->cap = cap; m->n = 0; m->key = (u128 *)malloc(sizeof(u128) * cap); m->val = (uint32_t *)malloc(4 * cap); m->used = (uint8_t *)calloc(cap, 1); }
static void km_grow(kmap *m) {
    kmap o = *m;
    km_init(m, o.cap * 2);
    for (uint64_t i = 0; i < o.cap; ++i) if (o.used[i]) {
        uint64_t h = hword(o.key[i]) & (m->cap - 1);
        while (m->used[h]) h = (h + 1) & (m->cap - 1);
        m->used[h] = 1; m->key[h] = o.key[i]; m->val[h] = o.val[i]; m->n++;
    }
    free(o.key); free(o.val); free(o.used);
}
/* slot of w (inserted with `init` when absent; *fresh says which) */
static uint64_t km_slot(kmap *m, u128 w, uint32_t init, int *fresh) {
    if (m->n * 10 >= m->cap * 6) km_grow(m);
    uint64_t h = hword(w) & (m->cap - 1);
    while (m->used[h]) { if (m->key[h] == w) { *fresh = 0; return h; } h = (h + 1) & (m->cap - 1); }
    m->used[h] = 1; m->key[h] = w; m->val[h] = init; m->n++;
    *fresh = 1;
    return h;
}

static inline int code_of(uint8_t c) {                              /* itree.c:110-121 */
    switch (c) { case 'a': case 'A': return 0; case 'c': case 'C': return 1; case 'g': case 'G': return 2; case 't': case 'T': return 3; default: return 255; }
}

typedef struct { const char *name, *label; } mapent;
static int by_name(const void *a, const void *b) {                  /* xcmp (486-489): unsigned byte order */
    const unsigned char *x = (const unsigned char *)((const mapent *)a)->name, *y = (const unsigned char *)((const mapent *)b)->name;
    while (*x == *y) { if (!*x) return 0; ++x; ++y; }
    return (int)*x - (int)*y;
}
/* crBST (itree.c:475-484): same probe sequence as the reference's search (matters only for duplicate names) */
static long find_name(const mapent *e, size_t n_minus_1, const char *key) {
    const mapent *p = e;
    size_t sz = n_minus_1;
    while (sz) {
        size_t w = sz >> 1;
        const char *r = p[w + 1].name, *k = key;
        while (*r == *k) { if (!*r) return (long)(p + w + 1 - e); ++r; ++k; }
        if (*r < *k) { p += w + 1; sz -= w + 1; } else sz = w;       /* plain char compare, as the reference */
    }
    return strcmp(p->name, key) ? -1 : (long)(p - e);
}

typedef struct { u128 w; uint32_t ix; } node;
static int by_word(const void *a, const void *b) { u128 x = ((const node *)a)->w, y = ((const node *)b)->w; return x < y ? -1 : x > y; }

int orc_build_file(const char *fasta, const char *map, const char *out_ubt, int W, int I, int complevel, int gg,
                   uint64_t *n_seqs, uint64_t *n_nodes, uint64_t *n_labels, char *err, size_t errlen) {
    if ((W != 8 && W != 16) || (I != 2 && I != 4) || complevel < 0) { set_err(err, errlen, "bad W/I/complevel"); return 3; }
    const uint32_t K = 4u * (uint32_t)W, k1 = K - 1, lv = (uint32_t)complevel, kv = k1 + lv;
    const uint32_t BAD = I == 2 ? 0xFFFFu : 0xFFFFFFFFu, EMPTY = BAD - 1;             /* itree.c:105-106 */
    size_t fn = 0, mn = 0;
    uint8_t *fa = slurp(fasta, &fn), *mp = slurp(map, &mn);
    if (!fa || !mp) { free(fa); free(mp); set_err(err, errlen, "Invalid input file(s)"); return 1; }          /* 504 */
    if (!mn) { free(fa); free(mp); set_err(err, errlen, "Input map empty."); return 1; }                       /* 512 */
    /* ---- map (itree.c:513-561, ixCol = 0, lblCol = 1) ---- */
    size_t lines = 0;
    for (size_t i = 0; i < mn; ++i) lines += mp[i] == '\n';
    if (mp[mn - 1] != '\n') ++lines;
    mapent *ent = (mapent *)malloc(sizeof(mapent) * (lines ? lines : 1));
    char *ptr = (char *)mp;
    int rc = 0;
    for (size_t i = 0; i < lines && !rc; ++i) {
        if (*ptr == '\n' || *ptr == '\r') { snprintf(err, errlen, "ERROR: map line %zu Blank indices are NOT ALLOWED.", i); rc = 2; break; }
        ent[i].name = ptr;
        if (*ptr == '\t') { snprintf(err, errlen, "map: extra tab, line %zu", i); rc = 2; break; }
        while (*++ptr != '\t') if (!*ptr) { snprintf(err, errlen, "Err tab1: %zu", i); rc = 2; break; }
        if (rc) break;
        *ptr++ = 0;
        if (*ptr == '\n' || *ptr == '\r') { snprintf(err, errlen, "ERROR: map line %zu Blank labels are NOT ALLOWED.", i + 1); rc = 2; break; }
        ent[i].label = ptr;
        while (*ptr != '\n') {
            if (!*ptr) { snprintf(err, errlen, "Err line counter: %zu", i); rc = 2; break; }
            if (*ptr == '\r' || *ptr == '\t') *ptr = 0;
            ptr++;
        }
        if (rc) break;
        *ptr++ = 0;
    }
    if (rc) { free(ent); free(fa); free(mp); return rc; }
    qsort(ent, lines, sizeof(mapent), by_name);                                          /* 563-571 */
    /* ---- references, one (header, sequence) line pair at a time (573-625) ---- */
    labtab lt; memset(&lt, 0, sizeof lt);
    kmap km; km_init(&km, 1u << 16);
    uint64_t ns = 0;
    size_t pos = 0;
    char cut[65536 + 8];
    while (pos < fn && !rc) {
        ++ns;
        uint8_t *nl = (uint8_t *)memchr(fa + pos, '\n', fn - pos);
        size_t hl = nl ? (size_t)(nl - (fa + pos)) + 1 : fn - pos;
        if (nl) *nl = 0;                                                                 /* 577-578 */
        long pre = find_name(ent, lines - 1, (const char *)fa + pos + 1);               /* 579: everything after the first byte */
        if (pre < 0) { snprintf(err, errlen, "Error: taxon map incomplete (line %llu)", (unsigned long long)ns); rc = 4; break; }
        uint32_t ix = lt_intern(&lt, ent[pre].label);                                    /* 583 addSampleU */
        if (ix >= EMPTY) { set_err(err, errlen, "too many labels for IXTYPE"); rc = 3; break; }
        pos += hl;
        if (pos >= fn) { snprintf(err, errlen, "Error parsing FASTA (1pass): %llu", (unsigned long long)ns); rc = 2; break; }   /* 585-586 */
        nl = (uint8_t *)memchr(fa + pos, '\n', fn - pos);
        size_t sl = nl ? (size_t)(nl - (fa + pos)) + 1 : fn - pos;
        const uint8_t *src = fa + pos;
        const uint8_t *z = (const uint8_t *)memchr(src, 0, sl);
        uint32_t length = (uint32_t)(z ? (size_t)(z - src) : sl);                        /* 588 strlen */
        if (length && src[length - 1] == '\n') --length;                                 /* 589 */
        if (length && src[length - 1] == '\r') --length;                                 /* 590 */
        pos += sl;
        for (uint32_t i = kv; i < length; ++i) {                                         /* 593-617 */
            const uint32_t s = i - kv;
            if (lv >= 1 && code_of(src[s]) != 0) continue;                               /* the lv bases before the k-mer: A, G, C, T */
            if (lv >= 2 && code_of(src[s + 1]) != 2) continue;
            if (lv >= 3 && code_of(src[s + 2]) != 1) continue;
            if (lv >= 4 && code_of(src[s + 3]) != 3) continue;
            u128 w = 0;
            uint32_t j = i - k1;
            int bad = 0;
            for (; j <= i; ++j) {
                int c = code_of(src[j]);
                if (c == 255) { bad = 1; break; }
                w = (w << 2) | (u128)c;
            }
            if (bad) { i += j - (i - k1) + lv; continue; }                               /* 612: resume right after the bad base */
            if (K == 32) w &= (u128)0xFFFFFFFFFFFFFFFFull;
            int fresh;
            uint64_t h = km_slot(&km, w, ix, &fresh);
            if (fresh || km.val[h] == ix) continue;                                      /* 262 / 280 */
            if (km.val[h] >= EMPTY) continue;                                            /* already bad (263, 281) */
            if (!gg) { km.val[h] = BAD; continue; }                                      /* 264 */
            const char *old = lt.str[km.val[h]], *nw = lt.str[ix];                       /* 282-296 */
            uint32_t numP = 0, ixP = 0, q = 0;
            while (old[q] == nw[q] && old[q]) { if (old[q] == ';') { ++numP; ixP = q; } ++q; }   /* distinct interned strings differ somewhere */
            if (numP < 2) { km.val[h] = BAD; continue; }                                 /* critical_cutoff (74, 295) */
            if (ixP > 65535) ixP = 65535;
            memcpy(cut, old, ixP); cut[ixP] = 0;
            uint32_t nix = lt_intern(&lt, cut);                                          /* 299 addSampleUd */
            if (nix >= EMPTY) { set_err(err, errlen, "too many labels for IXTYPE"); rc = 3; break; }
            km.val[h] = nix;
        }
    }
    if (!rc && !km.n) { set_err(err, errlen, "Error: no k-mers. Bad input/params!"); rc = 2; }                 /* 631 */
    if (!rc) {
        /* ---- dump (UT_writeTreeBinary 1317-1343, UT_writeSamples 1225-1232) ---- */
        node *nd = (node *)malloc(sizeof(node) * (km.n ? km.n : 1));
        uint64_t nn = 0;
        for (uint64_t i = 0; i < km.cap; ++i) if (km.used[i] && km.val[i] < EMPTY) { nd[nn].w = km.key[i]; nd[nn].ix = km.val[i]; ++nn; }
        qsort(nd, nn, sizeof(node), by_word);
        uint64_t *cnt = (uint64_t *)calloc(lt.n ? lt.n : 1, sizeof(uint64_t));
        FILE *of = fopen(out_ubt, "wb");
        if (!of) { set_err(err, errlen, "Invalid output filename"); rc = 1; }
        else {
            uint64_t md[4] = {(uint64_t)W, 0, (uint64_t)I, nn};
            fwrite(md, 8, 4, of);
            for (uint64_t i = 0; i < nn; ++i) { fwrite(&nd[i].w, (size_t)W, 1, of); fwrite(&nd[i].ix, (size_t)I, 1, of); ++cnt[nd[i].ix]; }
            for (uint32_t i = 0; i < lt.n; ++i) fprintf(of, "%s\t%llu\n", lt.str[i], (unsigned long long)cnt[i]);
            fclose(of);
            char *logp = (char *)malloc(strlen(out_ubt) + 16);
            sprintf(logp, "%s%s.log", out_ubt, gg ? ".gg" : "");                         /* itree.c:1405 */
            FILE *lf = fopen(logp, "wb");
            if (lf) { for (uint32_t i = 0; i < lt.n; ++i) fprintf(lf, "%s\t%llu\n", lt.str[i], (unsigned long long)cnt[i]); fclose(lf); }
            free(logp);
        }
        if (n_nodes) *n_nodes = nn;
        free(nd); free(cnt);
    }
    if (n_seqs) *n_seqs = rc ? ns : ns;
    if (n_labels) *n_labels = lt.n;
    for (uint32_t i = 0; i < lt.n; ++i) free(lt.str[i]);
    free(lt.str); free(lt.slot); free(km.key); free(km.val); free(km.used); free(ent); free(fa); free(mp);
    return rc;
}
