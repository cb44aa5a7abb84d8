/* fasta.c -- read framing and output formatting (host, C).
 *
 * utree_fasta_frame reproduces how XT_INITIATE_WS frames reads with two fgets(…, 16 MiB) per read
 * (itree.c:860-890) -- without its `omp critical` section: framing is a scan for newlines over a chunk
 * that is already in memory, and the sequence bytes are not copied (they go to HBM as they stand).
 * utree_format_records writes the lines of itree.c:1032 / 1040 / 1096.
 */
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
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

/* One read starting at `pos` (which must be the start of a header line).  Returns 0 = framed, 1 = needs more
 * bytes (only when !final), 2 = malformed (err filled). */
static int frame_one(const uint8_t *buf, size_t n, int final, size_t pos, size_t idx, uint64_t *seq_off, uint32_t *seq_len,
                     uint64_t *name_off, uint32_t *name_len, size_t *next, utree_fasta_error *err) {
    int c1, c2;
    size_t hl = line_at(buf, n, pos, &c1);
    if (!c1 && !final) return 1;                                  /* header continues in the next chunk */
    size_t spos = pos + hl;
    if (spos >= n) {
        if (!final) return 1;
        err->code = 1; err->read_index = idx;                      /* "can't read sequence", itree.c:872 */
        return 2;
    }
    size_t sl = line_at(buf, n, spos, &c2);
    if (!c2 && !final) return 1;
    if (buf[pos] != '>') { err->code = 2; err->read_index = idx + 1; return 2; }        /* itree.c:880 */
    size_t e = pos + 1;                                            /* name ends at NUL, space or newline (881) */
    while (e < pos + hl && buf[e] && buf[e] != ' ' && buf[e] != '\n') ++e;
    if (buf[spos] == '>') { err->code = 3; err->read_index = idx + 1; return 2; }       /* itree.c:886 */
    const uint8_t *z = (const uint8_t *)memchr(buf + spos, 0, sl); /* strlen (887) stops at a NUL */
    size_t length = z ? (size_t)(z - (buf + spos)) : sl;
    if (!length) { err->code = 4; err->read_index = idx + 1; return 2; }                /* itree.c:888 */
    if (buf[spos + length - 1] == '\n') --length;                  /* itree.c:889 */
    if (length && buf[spos + length - 1] == '\r') --length;        /* itree.c:890 */
    seq_off[idx] = spos; seq_len[idx] = (uint32_t)length;
    name_off[idx] = pos + 1; name_len[idx] = (uint32_t)(e - (pos + 1));
    *next = spos + sl;
    return 0;
}

static int frame_serial(const uint8_t *buf, size_t n, int final, size_t max_reads, uint64_t *seq_off, uint32_t *seq_len,
                        uint64_t *name_off, uint32_t *name_len, size_t *n_reads, size_t *consumed, utree_fasta_error *err) {
    size_t pos = 0, nr = 0;
    int rc = UTREE_OK;
    while (pos < n && nr < max_reads) {
        size_t next = pos;
        int r = frame_one(buf, n, final, pos, nr, seq_off, seq_len, name_off, name_len, &next, err);
        if (r == 1) break;
        if (r == 2) { rc = UTREE_E_FASTA; break; }
        ++nr; pos = next;
    }
    *n_reads = nr;
    *consumed = pos;
    return rc;
}

#define FRAME_MAX_SEG 64

int utree_fasta_frame(const uint8_t *buf, size_t n, int final, size_t max_reads, uint64_t *seq_off, uint32_t *seq_len,
                      uint64_t *name_off, uint32_t *name_len, size_t *n_reads, size_t *consumed, utree_fasta_error *err) {
    utree_fasta_error dummy;
    if (!n_reads || !consumed || (n && !buf)) return UTREE_E_ARG;
    if (!err) err = &dummy;
    err->code = 0; err->read_index = 0;
    int T = 1;
#ifdef _OPENMP
    T = omp_get_max_threads();
    if (T > 8) T = 8;
#endif
    if ((size_t)T > n / ((size_t)4 << 20)) T = (int)(n / ((size_t)4 << 20));      /* >= 4 MiB per segment */
    if (T < 2 || getenv("UTREE_FRAME_SERIAL")) return frame_serial(buf, n, final, max_reads, seq_off, seq_len, name_off, name_len, n_reads, consumed, err);
    if (T > FRAME_MAX_SEG) T = FRAME_MAX_SEG;
    /* pass 1: newlines per byte segment; a line of >= LINELEN-1 bytes needs the fgets-splitting serial path */
    size_t cnt[FRAME_MAX_SEG + 1], first_nl[FRAME_MAX_SEG], last_nl[FRAME_MAX_SEG], maxgap[FRAME_MAX_SEG];
#pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int t = 0; t < T; ++t) {
        size_t a = n * (size_t)t / (size_t)T, b = n * (size_t)(t + 1) / (size_t)T, c = 0, prev = (size_t)-1, gap = 0, f = (size_t)-1;
        const uint8_t *p = buf + a, *endp = buf + b;
        while (p < endp) {
            const uint8_t *q = (const uint8_t *)memchr(p, '\n', (size_t)(endp - p));
            if (!q) break;
            size_t at = (size_t)(q - buf);
            if (f == (size_t)-1) f = at;
            else if (at - prev > gap) gap = at - prev;
            prev = at; ++c; p = q + 1;
        }
        cnt[t] = c; first_nl[t] = f; last_nl[t] = prev; maxgap[t] = gap;
    }
    size_t total_nl = 0, prev_nl = (size_t)-1;
    int giant = 0;
    for (int t = 0; t < T; ++t) {
        if (maxgap[t] >= LINELEN - 1) giant = 1;
        if (cnt[t]) {
            size_t startgap = prev_nl == (size_t)-1 ? first_nl[t] + 1 : first_nl[t] - prev_nl;
            if (startgap >= LINELEN - 1) giant = 1;
            prev_nl = last_nl[t];
        }
        size_t c = cnt[t]; cnt[t] = total_nl; total_nl += c;
    }
    cnt[T] = total_nl;
    size_t tail = prev_nl == (size_t)-1 ? n : n - (prev_nl + 1);       /* bytes after the last newline */
    if (tail >= LINELEN - 1) giant = 1;
    size_t lines = total_nl + ((final && tail) ? 1 : 0);
    size_t reads = lines / 2;
    if (giant || reads > max_reads || reads < (size_t)T)
        return frame_serial(buf, n, final, max_reads, seq_off, seq_len, name_off, name_len, n_reads, consumed, err);
    /* pass 2: the thread in whose segment a HEADER line (even line index) starts frames that read */
    size_t err_idx[FRAME_MAX_SEG]; utree_fasta_error errs[FRAME_MAX_SEG]; size_t endpos[FRAME_MAX_SEG];
#pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int t = 0; t < T; ++t) {
        size_t a = n * (size_t)t / (size_t)T, b = n * (size_t)(t + 1) / (size_t)T;
        err_idx[t] = (size_t)-1; endpos[t] = 0;
        /* first line that STARTS in [a,b): position a itself if a == 0 or buf[a-1] == '\n', else after the first newline */
        size_t pos, line_ix;
        if (a == 0) { pos = 0; line_ix = 0; }
        else if (buf[a - 1] == '\n') { pos = a; line_ix = cnt[t]; }
        else { if (first_nl[t] == (size_t)-1 || cnt[t + 1] == cnt[t]) continue; pos = first_nl[t] + 1; line_ix = cnt[t] + 1; }
        if (line_ix & 1) {                                           /* a sequence line: belongs to the previous thread's read */
            const uint8_t *q = (const uint8_t *)memchr(buf + pos, '\n', n - pos);
            if (!q) continue;
            pos = (size_t)(q - buf) + 1; ++line_ix;
        }
        size_t idx = line_ix / 2;
        while (pos < b && pos < n && idx < reads) {
            size_t next = pos;
            utree_fasta_error e1 = {0, 0};
            int r = frame_one(buf, n, final, pos, idx, seq_off, seq_len, name_off, name_len, &next, &e1);
            if (r == 1) break;
            if (r == 2) { err_idx[t] = idx; errs[t] = e1; break; }
            ++idx; pos = next;
        }
        endpos[t] = pos;
    }
    size_t nr = reads, used = 0;
    int rc = UTREE_OK;
    for (int t = 0; t < T; ++t) if (err_idx[t] != (size_t)-1 && err_idx[t] < nr) { nr = err_idx[t]; *err = errs[t]; rc = UTREE_E_FASTA; }
    if (rc == UTREE_OK) {
        for (int t = 0; t < T; ++t) if (endpos[t] > used) used = endpos[t];
        /* a trailing error-free remainder in final mode (odd line count) is the reference's "can't read sequence" */
        if (final && (lines & 1)) {
            size_t next = used;
            utree_fasta_error e1 = {0, 0};
            int r = used < n ? frame_one(buf, n, final, used, nr, seq_off, seq_len, name_off, name_len, &next, &e1) : 0;
            if (r == 2) { *err = e1; rc = UTREE_E_FASTA; }
        }
    } else used = nr ? (size_t)(seq_off[nr - 1] + seq_len[nr - 1]) : 0;
    *n_reads = nr;
    *consumed = used;
    return rc;
}

/* ---- opt-in input formats the reference does not read: FASTQ and multi-line FASTA ----------
 * Serial framing of complete records in h_buf[0..n).  Same outputs as utree_fasta_frame; multi-line FASTA sequences are
 * compacted IN PLACE (line ends removed) so that the device sees contiguous bases.  Names follow the reference's rule
 * (bytes after the first byte of the header up to the first space / newline / NUL, itree.c:881). */
static size_t line_end(const uint8_t *buf, size_t n, size_t pos, int *complete) {
    const uint8_t *nl = (const uint8_t *)memchr(buf + pos, '\n', n - pos);
    if (nl) { *complete = 1; return (size_t)(nl - buf) + 1; }
    *complete = 0;
    return n;
}
static size_t strip_eol(const uint8_t *buf, size_t a, size_t b) {          /* [a,b) minus one trailing "\n" and one "\r" */
    if (b > a && buf[b - 1] == '\n') --b;
    if (b > a && buf[b - 1] == '\r') --b;
    return b;
}
int utree_reads_frame(uint8_t *buf, size_t n, int final, int format, size_t max_reads, uint64_t *seq_off, uint32_t *seq_len,
                      uint64_t *name_off, uint32_t *name_len, size_t *n_reads, size_t *consumed, utree_fasta_error *err) {
    if (!buf || !seq_off || !seq_len || !name_off || !name_len || !n_reads || !consumed || !err) return UTREE_E_ARG;
    if (format != UTREE_INPUT_FASTQ && format != UTREE_INPUT_FASTA_MULTILINE) return UTREE_E_ARG;
    size_t pos = 0, nr = 0;
    err->code = 0; err->read_index = 0;
    const uint8_t lead = format == UTREE_INPUT_FASTQ ? '@' : '>';
    while (pos < n && nr < max_reads) {
        int c1;
        size_t h_end = line_end(buf, n, pos, &c1);
        if (!c1 && !final) break;                                       /* header continues in the next chunk */
        if (buf[pos] != lead) { err->code = 2; err->read_index = nr + 1; *n_reads = nr; *consumed = pos; return UTREE_E_FASTA; }
        size_t e = pos + 1;
        while (e < h_end && buf[e] && buf[e] != ' ' && buf[e] != '\n') ++e;
        size_t s0, s1, next;
        if (format == UTREE_INPUT_FASTQ) {
            int c2, c3, c4;
            if (h_end >= n) { if (!final) break; err->code = 1; err->read_index = nr; *n_reads = nr; *consumed = pos; return UTREE_E_FASTA; }
            size_t q_end = line_end(buf, n, h_end, &c2);               /* sequence line */
            if (!c2 && !final) break;
            size_t p_end = q_end < n ? line_end(buf, n, q_end, &c3) : n;   /* '+' line */
            if ((q_end >= n || !c3) && !final) break;
            size_t l_end = p_end < n ? line_end(buf, n, p_end, &c4) : n;   /* quality line */
            if ((p_end >= n || !c4) && !final) break;
            if (q_end >= n || buf[q_end] != '+') { err->code = 3; err->read_index = nr + 1; *n_reads = nr; *consumed = pos; return UTREE_E_FASTA; }
            s0 = h_end; s1 = strip_eol(buf, h_end, q_end);
            next = l_end;
        } else {
            /* sequence lines up to the next header line (or the end of the input) */
            size_t p = h_end, w = h_end;
            int closed = 0;
            while (p < n) {
                if (buf[p] == '>') { closed = 1; break; }
                int cl;
                size_t le = line_end(buf, n, p, &cl);
                if (!cl && !final) { p = n + 1; break; }               /* line not complete yet */
                p = le;
            }
            if (p == n + 1 || (!closed && !final)) break;               /* the record may go on in the next chunk */
            /* compact in place */
            size_t q = h_end;
            while (q < p) {
                int cl;
                size_t le = line_end(buf, n, q, &cl);
                if (le > p) le = p;
                size_t b = strip_eol(buf, q, le);
                if (w != q) memmove(buf + w, buf + q, b - q);
                w += b - q;
                q = le;
            }
            s0 = h_end; s1 = w;
            next = p;
        }
        if (s1 - s0 > 0x3FFFFFFFu) { err->code = 5; err->read_index = nr + 1; *n_reads = nr; *consumed = pos; return UTREE_E_FASTA; }
        seq_off[nr] = s0; seq_len[nr] = (uint32_t)(s1 - s0);
        name_off[nr] = pos + 1; name_len[nr] = (uint32_t)(e - (pos + 1));
        ++nr;
        pos = next;
    }
    *n_reads = nr;
    *consumed = pos;
    return UTREE_OK;
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
