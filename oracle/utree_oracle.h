/* utree_oracle.h -- CPU restatement of the UTree SEARCH_GG hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity checker for the HIP product in utree_amd/.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (libutree_amd.so, xtree-searchGG) never links, imports or executes anything under oracle/.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement byte-for-byte against
 * outputs of the genuine reference (itree.c -D SEARCH_GG, compiled by oracle/Makefile `make ref` into
 * oracle/_ref/) captured in tests/golden/ by tests/golden/make_golden.py.
 *
 * Every function cites the reference lines (/root/reference/itree.c) it restates.
 */
#ifndef UTREE_ORACLE_H
#define UTREE_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_BAD_IX 0xFFFFFFFFu      /* "not found" (itree.c:105 BAD_IX widened to 32 bits) */
#define ORC_NUMBINS ((1u << 24) + 1) /* itree.c:693 */

typedef struct orc_db orc_db;

/* One read's classification, the fields itree.c:1032/1040/1096 print. */
typedef struct {
    uint32_t label;     /* label index whose text is (partly) printed; undefined when found==0      */
    int32_t  cut;       /* -2 = whole label text, -1 = empty taxon, >=0 = first `cut` bytes of label */
    uint32_t found;     /* foundUniq: number of k-mer hits (0 => the reference prints no line)       */
    uint32_t uix;       /* number of distinct labels among the hits                                   */
    uint32_t sl, ol;    /* support pair; printed as "*" instead when uix==1                           */
} orc_result;

/* itree.c:733-828 XT_read32 + 1154-1223 readSamplesFPdelim.  NULL on failure (err filled). */
orc_db *orc_db_load(const char *path, char *err, size_t errlen);
/* Same DB from memory in the on-disk pieces (records are SZ-byte packed, binix already widened to u64;
 * label text as in the file tail).  Copies everything. */
orc_db *orc_db_from_memory(uint32_t W, uint32_t I, uint64_t n_nodes, const uint64_t *binix,
                           const uint8_t *records, const char *label_text, size_t label_len,
                           char *err, size_t errlen);
void orc_db_free(orc_db *db);
uint32_t orc_db_W(const orc_db *db);
uint32_t orc_db_I(const orc_db *db);
uint64_t orc_db_nodes(const orc_db *db);
uint32_t orc_db_labels(const orc_db *db);
const char *orc_db_label(const orc_db *db, uint32_t ix);

/* itree.c:903-927: enumerate k-mer windows. Writes for each looked-up window its end position and
 * the packed word (hi:lo; hi is 0 for k<=32). Returns number of windows (<= cap written). */
size_t orc_windows(const uint8_t *seq, size_t len, int k, uint32_t *end_pos, uint64_t *hi, uint64_t *lo,
                   size_t cap);
/* itree.c:720-730 XT_getIX32 + 699-707 xtSuffixBS.  Returns raw stored ix or ORC_BAD_IX. */
uint32_t orc_lookup(const orc_db *db, uint64_t hi, uint64_t lo);
/* itree.c:1028-1096: tally + sort + vote on a hit list (label indices < nlabels). */
void orc_vote(const orc_db *db, const uint32_t *hits, uint32_t nhits, orc_result *res);
/* itree.c:1032,1040,1096: format one output line (returns bytes written, 0 when res->found==0). */
size_t orc_format(const orc_db *db, const char *name, size_t name_len, const orc_result *res, char *out,
                  size_t cap);
/* itree.c:891-898: forward + 'N' + reverse complement into dst (2*len+1 bytes). */
void orc_revcomp_append(const uint8_t *src, size_t len, uint8_t *dst);

/* Whole path, one read from memory (a3..a9). scratch-free convenience used by tests. */
void orc_classify_read(const orc_db *db, const uint8_t *seq, size_t len, int do_rc, orc_result *res);
/* Many reads from memory, OpenMP over reads (the bench's cpu_baseline "port"). */
void orc_classify_batch(const orc_db *db, const uint8_t *buf, const uint64_t *off, const uint32_t *len,
                        size_t n, int do_rc, int threads, orc_result *out);
/* itree.c:833-1108 XT_doSearch32, GG branch: FASTA file -> classification file, lines in INPUT order
 * (= the reference run with 1 thread).  Returns 0 ok, or the reference's exit code (1,2,3). */
int orc_search_file(const orc_db *db, const char *fasta, const char *out, int threads, int do_rc,
                    uint64_t *n_reads, uint64_t *good_finds, char *err, size_t errlen);

/* ---- rank-specific search: `xtree-search`, itree.c -D SEARCH (doCollapse = 0 branch, 969-1007) --------
 * Sequential by construction: each read's vote also counts one entry left in the hit array by an
 * earlier read (itree.c:982), so a state object is threaded through the reads in file order. */
typedef struct { uint32_t slack, sparsity, tolerance; } orc_rank_params;   /* itree.c:952-960: 2, 4, 2 */
typedef struct {
    uint32_t label;     /* mostIX                                                                      */
    uint32_t printed;   /* 1 iff the reference prints a line (itree.c:1000-1002)                        */
    uint32_t found;     /* hits kept after the sparsity skip (foundUniq == kingsMen)                    */
    uint32_t most, second;
} orc_rank_result;
typedef struct orc_rank_state orc_rank_state;
orc_rank_state *orc_rank_state_new(const orc_db *db);
void orc_rank_state_free(orc_rank_state *st);
void orc_rank_read(const orc_db *db, orc_rank_state *st, const uint8_t *seq, size_t len, int do_rc,
                   const orc_rank_params *prm, orc_rank_result *res);
size_t orc_rank_format(const orc_db *db, const char *name, size_t name_len, const orc_rank_result *r, char *out,
                       size_t cap);
int orc_rank_search_file(const orc_db *db, const char *fasta, const char *out, int do_rc,
                         const orc_rank_params *prm, uint64_t *n_reads, uint64_t *good_finds, char *err,
                         size_t errlen);

/* ---- database BUILD: `utree-build` / `utree-buildGG` (itree.c -D BUILD / BUILD_GG), utree_build_oracle.c ---------
 * FASTA (header line + sequence line per reference) + `name \t label` map -> `.ubt` and `<out>[.gg].log`.
 * Returns 0 or the reference's exit code (1 files, 2 malformed map / FASTA / no k-mers, 4 name not in the map). */
int orc_build_file(const char *fasta, const char *map, const char *out_ubt, int W, int I, int complevel, int gg,
                   uint64_t *n_seqs, uint64_t *n_nodes, uint64_t *n_labels, char *err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif
