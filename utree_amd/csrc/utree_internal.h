/* utree_internal.h -- shared between the C host files and kernels.hip (not part of the public ABI). */
#ifndef UTREE_INTERNAL_H
#define UTREE_INTERNAL_H
#include <stddef.h>
#include <stdint.h>
#include "../../include/utree_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define UTREE_NUMBINS ((1u << 24) + 1u)          /* itree.c:693 */
#define UTREE_INVALID 0xFFFFFFFFu
#define UTREE_IMG_MAGIC 0x31474d4945525455ull    /* "UTREIMG1" */
#define UTREE_IMG_HEADER_BYTES 4096u
#define UTREE_IMG_VERSION 14u                    /* 9: buckets of 64 or 128 bytes; any number of buckets per hash region; 10: the vote records carry the file-order index; 11: canonical minimizers, strand-paired buckets; 12: heavy overflow runs start with a position directory; 13: k = 64: minimizers keep two bases' distance from the k-mer's ends and those four bases split a hash value's pair of buckets (UTREE_MIN_MARGIN, sub-slices); 14: k = 32: heavy overflow runs are stored as CHAINS of consecutive k-mers (UTREE_F_OVF_CHAINS) */
/* How a 16-mer gets its strand-independent rank and address (image version 11, device_common.hpp):
 *   1: the hash of its canonical form, the smaller of the 16-mer and its reverse complement -- three vector instructions per base of a read
 *      on top of the forward walk, but only every other hash value is some canonical 16-mer's: where the table has a pair of buckets per
 *      hash value (the dense end of the range) every other pair stays empty, and regions of a few values per pair must be sized for the lumps
 *   2: the smaller of the two hashes -- a second hash per base (seven instructions), every value in use, the table as small as before */
#ifndef UTREE_CANON_MODE
#define UTREE_CANON_MODE 1
#endif
#define UTREE_REGION_NB_BITS 25                  /* regions[r] = base_r << 34 | sub_r << 25 | nb_r (nb_r <= 2^24 slots of hash values, each of sub_r <= 256 pairs) */
#define UTREE_REGION_SUB_BITS 9
#define UTREE_REGION_BASE_SHIFT (UTREE_REGION_NB_BITS + UTREE_REGION_SUB_BITS)
/* k = 64 (W = 16): a window's minimizer is chosen among the 16-mers that keep UTREE_MIN_MARGIN bases' distance from both of the window's
 * ends (positions 2 .. 46: 45 of the 49), so that the two bases on either side of it belong to every k-mer it is the minimizer of -- the same
 * four bases for all the windows of a run, read in the minimizer's canonical orientation: eight more address bits.  Where the hash range is
 * so crowded that ONE hash value holds more nodes than a bucket (568 M nodes: 6.5 per value and orientation at the dense end, four entries
 * per bucket) they split the value's pair of buckets into up to 256 (device_common.hpp: bucket_of; dev_image.c: compute_regions).  k = 32
 * keeps all its 17 positions: a margin would cost it a quarter more minimizer runs per read. */
#define UTREE_MIN_MARGIN(W_) ((W_) == 16 ? 2u : 0u)
#define UTREE_TALLY_CHUNK 8192u                  /* tally entries a wave reserves with one atomic              */
#define UTREE_CUR_LONG 32                        /* cursors[] index of the long-read counter (own 256-B line)   */
#define UTREE_CUR_MID 16                         /* ... of the mid-length-read counter                           */
#define UTREE_CUR_WORK 48                        /* ... of the next unclaimed read (150-bp-class pass)           */
#define UTREE_CUR_WORK_LONG 40                   /* ... of the next unclaimed long-list entry                    */
#define UTREE_CUR_WORK_MID 56                    /* ... of the next unclaimed mid-list entry                     */
/* The reads of the wave-per-read passes are handed out from UTREE_WORK_PARTS counters, each on its own 128-byte line, one
 * per contiguous part of the batch: 8192 resident waves on ONE counter queue up at its L2 channel (a grab of 32 reads then
 * costs a wave ~14 us, measured); a wave starts at part (its index mod PARTS) and moves on when a part is used up. */
#define UTREE_WORK_PARTS 64
#define UTREE_WORK_STRIDE 16                     /* 8-byte words between two counters                            */
#define UTREE_CUR_PIECES 8                       /* ... of the pieces of long reads listed for the lane-per-read pass */
#define UTREE_CUR_LEFT 24                        /* ... of the long reads that pass left to classify_long_k            */
#define UTREE_CUR_CLASS 9                        /* ... [9 .. 13]: reads of a mixed batch that take 1, 2, 4, 8, 16 lanes (lanes_route_k) */
#define UTREE_CUR_ERROR 17                       /* ... the batch's error word (UTREE_DEVERR_*; next to UTREE_CUR_MID: both come back in one copy) */
#define UTREE_CURSOR_BYTES (512 + 8 * UTREE_WORK_PARTS * UTREE_WORK_STRIDE * 8)   /* cursors[] + part counters (main, mid, pieces, five lane classes) */
/* what a kernel found that the workspace's sizing rules out (utree_classify_batch's caller gave a wrong total_bases / max_len, or a
 * test hook shrank the workspace): the batch's results are not to be used -- utree_classify_poll reports UTREE_E_DEVICE */
#define UTREE_DEVERR_TALLY_CAP 1ull              /* (rank, count) lists beyond tally_cap                               */
#define UTREE_DEVERR_LONG_CAP 2ull               /* more long reads than n_long_cap                                    */
#define UTREE_DEVERR_PIECES_CAP 3ull             /* more pieces of long reads than n_pieces_cap                        */
#define UTREE_LONG_SLOTS 64u                     /* {rank, count} slots of a long read's tally table in HBM (pieces mode) */
#define UTREE_SHORT_CAP 320u                     /* staged bases (incl. RC) the wave-per-read kernel holds      */
#define UTREE_SHORT2_CAP 640u                    /* ... its second size: 250-300 bp reads with the reverse strand */
#define UTREE_MID_DEFAULT 2112u                  /* = UTREE_MID_CAP: since r01l the wave-per-read pass beats classify_long_k up to its capacity (2100 bp: 65 vs 46 M reads/s); UTREE_MID_LIMIT overrides */
#define UTREE_LANES_CAP 160u                     /* bases the lane-per-read kernel (lanes_kernel.hip) holds per read */
#define UTREE_MID_CAP 2112u                      /* ... and its mid-length instantiation; longer: classify_long */

/* image flags */
#define UTREE_F_IRREGULAR 1u   /* some bins are not strictly ascending: bitmap present, exact probe path   */
#define UTREE_F_GENERIC   2u   /* bin table not monotone: fine_bits = 0, every lookup takes the exact path  */
#define UTREE_F_OFF64     4u   /* bin-table offsets are 64-bit (n_nodes >= UINT32_MAX)                       */
#define UTREE_F_VOTE_TABLE 16u /* labels are short and token-structured: vote_k takes its decisions from the table at off_vote */
#define UTREE_F_STRAND_VIEWS 32u /* every k-mer is also stored under its mirrored view where that differs (device_common.hpp): a window's reverse
                                    complement is found from the window's own minimizer run, in the other bucket of the pair */
#define UTREE_F_DIRECT 64u       /* PACKSIZE=16 (W = 4): off_table is a direct-address table, 2^32 ranks of I bytes (all ones: no node), every word's
                                    answer as the reference gives it; no buckets, no records                               */
#define UTREE_F_OVF_CHAINS 128u  /* k = 32: a heavy overflow run (descriptor bit 39) is a list of CHAINS -- the k-mers that walk one stretch of sequence across the
                                    minimizer, 16 bytes per chain + one rank per k-mer -- instead of a position directory + records (device_common.hpp) */
#define UTREE_F_INVALID_RANKS 8u   /* some node's label index is >= the number of labels (itree.c:929: never a hit): wave-per-read kernels only */

/* At offset 0 of the flat device image (position independent: offsets, never pointers). */
typedef struct {
    uint64_t magic;
    uint32_t version, W, I, k;
    uint32_t fine_bits, rec_words, n_labels, flags;
    uint64_t n_nodes;
    uint64_t n_slots;                /* buckets in the table: two (the orientations) per pair, pairs summed over the 256 hash regions */
    uint64_t n_min;                  /* MIN records = nodes the bin table reaches                          */
    uint64_t off_table, off_mrecs, off_recs, off_coarse, off_irreg, off_label_off, off_label_blob, off_rank2ix;
    uint64_t label_blob_bytes;
    uint64_t n_irregular;
    uint64_t total_bytes;
    uint64_t off_vote;               /* 32 bytes per label, rank order (UTREE_F_VOTE_TABLE): see utk_vote_rec                    */
    uint32_t bucket_words, pad0;     /* 8-byte words of a bucket: 8 (64 bytes: the faster kernels) or 16 (a whole 128-byte line: the smaller image) */
    /* Bucket addressing: region r = top 8 bits of the minimizer hash h (of the canonical 16-mer); regions[r] = base_r << 34 | sub_r << 25 | nb_r and
     * pair = base_r + (((h & 0xFFFFFF) * nb_r) >> 24) * sub_r + ((e * sub_r) >> 8) with 2^16 <= nb_r <= 2^24 slots in the region, sub_r pairs of
     * buckets per slot (1 unless nb_r = 2^24 and k = 64; e = the four bases around the minimizer, canonical), bucket = 2 pair + orientation,
     * so a pair spans at most 256 consecutive hash values and the low 8 bits of h go into the record key.  The hash is a MINIMUM of K-15
     * hashes, so nodes crowd towards h = 0: every region gets the number of pairs its expected share of the nodes asks for. */
    uint64_t regions[256];
} utree_image_header;

/* A label as the vote (itree.c:1044-1088) needs it, when every label of the database has at most 8 tokens (';'-separated), at most
 * 255 bytes, and ranks fit 16 bits: per token t the id of the label's prefix through that token (= the smallest rank of a label with
 * the same bytes up to the token's end and a terminator there), the byte offset of the token's terminator, and three bits -- the
 * token exists, a ';' follows it (else the label ends there), the byte before its terminator is '_'.  Two labels that agree up to
 * token t-1 have the same token t exactly when their ids at t are equal: the vote's byte scans become integer compares. */
typedef struct {
    uint16_t pid[8];
    uint8_t tok_end[8];
    uint8_t exists, more, us, n_tok;
    uint8_t len, pad;                /* the label's length */
    uint16_t ix;                     /* its index in the file's order (rank2ix): the vote's result needs no further lookup */
} utk_vote_rec;

/* What kernels take by value. */
typedef struct {
    const uint64_t *table;           /* buckets of bucket_words x 8 bytes: bucket_words / rec_words entries each, ascending by key; see regions[] */
    const uint64_t *regions;         /* [256] in the image header                                              */
    const uint64_t *mrecs;           /* MIN records: nodes ordered by (minimizer hash, position, rest)        */
    const uint64_t *recs;            /* FILE records: nodes as the file orders them (exact-probe path only)   */
    const void *coarse;              /* the file's bin table: uint32_t* or uint64_t* (UTREE_F_OFF64)         */
    const uint32_t *irreg;           /* 2^24-bit bitmap                                                      */
    const uint32_t *label_off;       /* [n_labels+1], rank order                                             */
    const char *label_blob;          /* NUL-terminated labels in strcmp order                                */
    const uint32_t *rank2ix;
    const uint64_t *vote_tab;        /* utk_vote_rec per rank, or NULL: vote_k reads the label bytes                       */
    uint64_t n_nodes;
    uint32_t n_labels, fine_bits, flags, W, I;
    uint32_t bucket_words;           /* 8 or 16                                                              */
    uint32_t ovf_scan;               /* overflow runs of up to this many records are read whole by the lane pass, longer ones searched */
    /* the 24-bit prefixes of the irregular bins when there are at most four of them (COMPRESS' first-bin quirk makes one or two;
     * unused entries are ~0), else irr_n = ~0: what lets the lane-per-read pass tell the reads it must leave alone */
    uint32_t irr_n, irr_p[4];
} utk_image;

static inline uint32_t utree_rec_words(uint32_t W, uint32_t I) { return (W == 16 ? 2u : 1u) * (I == 4 ? 2u : 1u); }   /* (W = 4 as W = 8) */

/* ---- launchers implemented in kernels.hip (all asynchronous on `stream`, return hipError_t as int) ---- */
/* *d_invalid += records whose label index is >= n_labels */
int utk_repack(uint32_t W, uint32_t I, const void *d_raw, uint64_t count, const uint32_t *d_ix2rank,
               uint32_t n_labels, uint64_t *d_recs_at_first, unsigned long long *d_invalid, void *stream);
int utk_widen_binix(const void *d_raw_binix, uint32_t width, int off64, void *d_coarse, void *stream);
/* counters[0] += irregular bins, counters[1] = 1 if the table is not monotone / exceeds n_nodes */
int utk_validate(uint32_t W, uint32_t I, int off64, const void *d_coarse, const uint64_t *d_recs, uint64_t n_nodes,
                 uint32_t *d_irreg, unsigned long long *d_counters, void *stream);
/* dup_cap: MIN records the area at d_mrecs holds beyond the m nodes' own (second views, UTREE_F_STRAND_VIEWS); *n_min = records written,
 * *views = 1 when every node that has a second view got it (else none did and the image must not claim the flag) */
int utk_build_min(uint32_t W, uint32_t I, int off64, const void *d_coarse, const uint64_t *d_recs, uint64_t c0, uint64_t m, uint64_t dup_cap,
                  const uint64_t *d_regions, const uint64_t *h_regions, uint64_t n_buckets, uint32_t bucket_words, uint64_t *d_table, uint64_t *d_mrecs,
                  uint32_t *d_irreg, unsigned long long *d_overflow, uint64_t *n_min, int *views, void *stream);
int utk_build_direct(uint32_t I, int off64, int generic, const void *d_coarse, const uint64_t *d_recs, uint64_t n_nodes, uint64_t c0, uint64_t m,
                     const uint32_t *d_irreg, uint64_t n_irregular, void *d_table, void *stream);
/* *chains = 1: heavy runs were written as chains (k = 32 unless UTREE_OVF_CHAINS=0): the image carries UTREE_F_OVF_CHAINS */
int utk_compact_overflow(uint32_t W, uint32_t I, uint64_t *d_table, uint64_t n_buckets, uint32_t bucket_words, uint64_t *d_mrecs, uint64_t *n_kept, int *chains, void *stream);
int utk_compress_chunk(uint32_t W, uint32_t I, const void *d_in, uint64_t first, uint64_t count, unsigned long long *d_first,
                       void *d_out, void *stream);
int utk_fill_recs_pad(uint64_t *d_recs_end, uint32_t words, void *stream);

/* workspace layout for one batch */
typedef struct {
    unsigned long long *cursors;     /* [0] tally bump, [UTREE_CUR_LONG] long-read count; then the work-part counters:
                                      * UTREE_CURSOR_BYTES, zeroed per batch */
    uint64_t *tally;                 /* (rank, count) pairs packed as rank | count<<32                        */
    uint64_t tally_cap;
    uint32_t *long_list;             /* [n_reads]                                                             */
    uint32_t *mid_list;              /* [n_reads]                                                             */
    uint32_t *hist;                  /* long path: [long_blocks][n_labels]                                    */
    uint32_t *touch;                 /* long path: touched-label bitmap when it does not fit LDS              */
    uint32_t long_blocks, mid_reads; /* mid_reads != 0: the batch may hold mid-length reads                   */
    uint32_t short_cap;              /* UTREE_SHORT_CAP or UTREE_SHORT2_CAP: what the batch's main wave-per-read pass holds */
    uint32_t mid_limit;              /* staged bases up to which the wave-per-read mid pass is used (<= UTREE_MID_CAP) */
    /* long reads through the lane-per-read pass, cut into pieces of sixteen lanes (lanes_kernel.hip); NULL: not for this batch */
    uint64_t *pieces;                /* {index in long_list << 32 | piece}                                      */
    uint32_t *ltab_rank, *ltab_cnt;  /* per long_list entry: UTREE_LONG_SLOTS ranks (~0: free) and hit counts    */
    uint32_t *lflag;                 /* per long_list entry: 1 = left to classify_long_k                        */
    uint32_t *long_left;             /* the reads so left                                                       */
    uint32_t n_long_cap;             /* long_list entries the tables above hold                                  */
    uint64_t n_pieces_cap;           /* entries of `pieces`                                                      */
    /* a batch of mixed read lengths through the lane-per-read pass: reads listed by the lanes they need (class c: 2^c lanes) */
    uint32_t *cls_list;              /* [5][cls_stride]; NULL: the batch is not routed                           */
    uint32_t cls_stride;
    uint64_t ltally_base;            /* ws.tally index of the long reads' (rank, count) lists: UTREE_LONG_SLOTS each */
} utk_workspace;

int utk_classify_short(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len,
                       uint32_t n_reads, int do_rc, utree_result *d_out, const utk_workspace *ws, int n_cu,
                       void *stream);
/* lane-per-read pass (lanes_kernel.hip): whether it takes this image / batch; reads it leaves go on ws->mid_list */
int utk_lanes_ok(const utk_image *im, uint32_t max_len, int do_rc);
int utk_lanes_segs(const utk_image *im, uint32_t max_len);      /* lanes per read (1 .. 16) for a batch's longest read */
/* long reads (ws->long_list) in pieces through the lane-per-read pass; what it cannot finish ends up on ws->long_left with its
 * count in cursors[UTREE_CUR_LONG], for utk_classify_long with long_list = long_left */
int utk_lanes_image_ok(const utk_image *im);
int utk_lanes_both_strands(const utk_image *im, int do_rc);      /* the BS instantiations take this batch: both strands from one pass */
int utk_classify_long_pieces(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len, int do_rc,
                             utree_result *d_out, const utk_workspace *ws, int n_cu, void *stream);
int utk_classify_lanes(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len, uint32_t n_reads,
                       uint32_t max_len, int do_rc, utree_result *d_out, const utk_workspace *ws, int n_cu, void *stream);
/* a batch of mixed lengths: routed by lanes per read, one launch per class; reads beyond sixteen lanes end up on ws->long_list */
int utk_classify_lanes_mixed(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len, uint32_t n_reads,
                             uint32_t max_len, int do_rc, utree_result *d_out, const utk_workspace *ws, int n_cu, void *stream);
uint32_t utk_lanes_max_len(const utk_image *im);                 /* longest read sixteen lanes hold */
int utk_route(const uint32_t *d_len, uint32_t n_reads, int do_rc, const utk_workspace *ws, void *stream);
int utk_classify_mid(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len,
                     uint32_t n_reads, int do_rc, utree_result *d_out, const utk_workspace *ws, int n_cu, void *stream);
/* the reads on ws->mid_list with the 150-bp-class wave-per-read kernel (what the lane-per-read pass left) */
int utk_classify_listed(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len,
                        uint32_t n_reads, uint32_t max_len, int do_rc, utree_result *d_out, const utk_workspace *ws, int n_cu, void *stream);
int utk_classify_long(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len,
                      int do_rc, utree_result *d_out, const utk_workspace *ws, int n_cu, void *stream);
int utk_vote(const utk_image *im, utree_result *d_out, const utk_workspace *ws, uint32_t n_reads, void *stream);
int utk_lookup(const utk_image *im, const uint64_t *d_hi, const uint64_t *d_lo, uint64_t n, uint32_t *d_ix,
               void *stream);
/* ---- rank-specific search (`xtree-search`, itree.c -D SEARCH): rank_kernels.hip --------------------------- */
typedef struct {
    unsigned long long *cursors;     /* [0] hit-list bump, [1..3] work counters (hits, short vote, long vote); 512 B */
    uint32_t *nh;                    /* [n_reads] hits kept per read (foundUniq == kingsMen, itree.c:930,951)  */
    uint64_t *hoff;                  /* [n_reads] where the read's hits start in `hits`                        */
    uint32_t *hits;                  /* file-order label indices, in hit order                                 */
    uint64_t hits_cap;
    uint32_t *lvl[3];                /* maxima of nh over 64, 4096, 262144 consecutive reads                   */
    uint32_t nlvl[3];
    uint32_t *hist;                  /* long vote: [hist_waves][n_labels] counters, all zero between reads     */
    uint32_t hist_waves;
    uint32_t *state;                 /* the reference's never-cleared hit array, as far as later reads can see */
    uint32_t state_cap;              /* it: state[j] = entry j of the latest read that had more than j hits    */
    uint32_t step;                   /* PACKSIZE / SPARSITY (itree.c:950)                                      */
    uint32_t slack, tolerance;       /* SLACK, TOLERANCE_THRESHOLD (itree.c:1000)                              */
} utk_rank_ws;

int utk_rank_hits(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len,
                  uint32_t n_reads, int do_rc, const utk_rank_ws *ws, int n_cu, void *stream);
int utk_rank_levels(const utk_rank_ws *ws, uint32_t n_reads, void *stream);
int utk_rank_vote(const utk_image *im, utree_result *d_out, uint32_t n_reads, const utk_rank_ws *ws, int n_cu,
                  void *stream);
int utk_rank_state(uint32_t n_reads, const utk_rank_ws *ws, void *stream);

const char *utk_classify_short_name(const utk_image *im, uint32_t short_cap, int mid, int do_rc, char *buf, size_t cap);
const char *utk_classify_long_name(const utk_image *im, char *buf, size_t cap);
int utk_model_counts(const utk_image *im, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len, uint32_t n_reads,
                     int do_rc, unsigned long long *d_counts, void *stream);

#ifdef __cplusplus
}
#endif
#endif
