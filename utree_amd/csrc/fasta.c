/* fasta.c -- read framing and output formatting (host, C).
 *
 * utree_fasta_frame reproduces how XT_INITIATE_WS frames reads with two fgets(…, 16 MiB) per read
 * (itree.c:860-890) -- without its `omp critical` section: framing is a scan for newlines over a chunk
 * that is already in memory, and the sequence bytes are not copied (they go to HBM as they stand).
 * utree_format_records writes the lines of itree.c:1032 / 1040 / 1096.
 */
#include <string.h>
#include "ctr_host.h"

#define LINELEN 16777216u                    /* itree.c:836 */

/* One fgets "line" starting at pos: up to LINELEN-1 bytes, ending after '\n'.  *complete = 0 when the
 * buffer ended before the line did (more bytes may follow in the next chunk). */
static size_t line_at(const uint8_t *buf, size_t n, size_t pos, int *complete) {
    size_t lim = n - pos;
    if (lim > LINELEN - 1) lim = LINELEN - 1;
    const uint8_t *nl = (const uint8_t *)memchr(buf + pos, '\n', lim);
    if (nl) { *complete = 1; return (size_t)(nl - (buf + pos)) + 1; }
    *complete = (lim == LINELEN - 1);
    return lim;
}

int utree_fasta_frame(const uint8_t *buf, size_t n, int final, size_t max_reads, uint64_t *seq_off, uint32_t *seq_len,
                      uint64_t *name_off, uint32_t *name_len, size_t *n_reads, size_t *consumed, utree_fasta_error *err) {
    if (!n_reads || !consumed || (n && !buf)) return UTREE_E_ARG;
    size_t pos = 0, nr = 0;
    int rc = UTREE_OK;
    if (err) { err->code = 0; err->read_index = 0; }
    while (pos < n && nr < max_reads) {
        int c1, c2;
        size_t hl = line_at(buf, n, pos, &c1);
        if (!c1 && !final) break;                                   /* header continues in the next chunk */
        size_t spos = pos + hl;
        if (spos >= n) {
            if (!final) break;
            if (err) { err->code = 1; err->read_index = nr; }        /* "can't read sequence", itree.c:872 */
            rc = UTREE_E_FASTA; break;
        }
        size_t sl = line_at(buf, n, spos, &c2);
        if (!c2 && !final) break;
        if (buf[pos] != '>') {                                       /* itree.c:880 */
            if (err) { err->code = 2; err->read_index = nr + 1; }
            rc = UTREE_E_FASTA; break;
        }
        size_t e = pos + 1;                                          /* name ends at NUL, space or newline (881) */
        while (e < pos + hl && buf[e] && buf[e] != ' ' && buf[e] != '\n') ++e;
        if (buf[spos] == '>') {                                      /* itree.c:886 */
            if (err) { err->code = 3; err->read_index = nr + 1; }
            rc = UTREE_E_FASTA; break;
        }
        const uint8_t *z = (const uint8_t *)memchr(buf + spos, 0, sl);   /* strlen (887) stops at a NUL */
        size_t length = z ? (size_t)(z - (buf + spos)) : sl;
        if (!length) {                                               /* itree.c:888 */
            if (err) { err->code = 4; err->read_index = nr + 1; }
            rc = UTREE_E_FASTA; break;
        }
        if (buf[spos + length - 1] == '\n') --length;                /* itree.c:889 */
        if (length && buf[spos + length - 1] == '\r') --length;      /* itree.c:890 */
        seq_off[nr] = spos; seq_len[nr] = (uint32_t)length;
        name_off[nr] = pos + 1; name_len[nr] = (uint32_t)(e - (pos + 1));
        ++nr;
        pos = spos + sl;
    }
    *n_reads = nr;
    *consumed = pos;
    return rc;
}

static inline char *put_u32(char *o, uint32_t v) {
    char tmp[10];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *o++ = tmp[--n];
    return o;
}

size_t utree_format_records(const utree_ctr *ctr, const uint8_t *h_buf, const uint64_t *name_off,
                            const uint32_t *name_len, const utree_result *res, size_t n, char *out, size_t cap,
                            uint64_t *good_finds) {
    char *o = out;
    uint64_t good = 0;
    for (size_t i = 0; i < n; ++i) {
        const utree_result *r = &res[i];
        if (!r->found) continue;                                     /* itree.c:1028: no hit, no line */
        if (r->label >= ctr->info.n_labels) return (size_t)-1;
        const char *lab = ctr->labels[r->label];
        size_t ll = r->cut == -1 ? 0 : r->cut == -2 ? ctr->label_len[r->label] : (size_t)r->cut;
        if (ll > ctr->label_len[r->label]) ll = ctr->label_len[r->label];
        size_t need = (size_t)name_len[i] + ll + 48;
        if ((size_t)(o - out) + need > cap) return (size_t)-1;
        memcpy(o, h_buf + name_off[i], name_len[i]); o += name_len[i];
        *o++ = '\t';
        memcpy(o, lab, ll); o += ll;
        *o++ = '\t';
        o = put_u32(o, r->found);
        *o++ = '\t';
        if (r->uix == 1) { *o++ = '1'; *o++ = '\t'; *o++ = '*'; }    /* itree.c:1032, 1040 */
        else {                                                       /* itree.c:1096 */
            o = put_u32(o, r->uix); *o++ = '\t';
            o = put_u32(o, r->sl); *o++ = ';';
            o = put_u32(o, r->ol);
        }
        *o++ = '\n';
        ++good;
    }
    if (good_finds) *good_finds += good;
    return (size_t)(o - out);
}
