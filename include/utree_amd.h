/* utree_amd.h -- C-ABI of the MI355X-native SEARCH_GG path (libutree_amd.so).
 *
 * Drop-in boundary for ONE path of knights-lab/UTree: `xtree-searchGG` (itree.c compiled -D SEARCH_GG).
 * The reference has no FFI; its path sits behind two internal C seams and one operator:
 *
 *     UTree *XT_read32(char *db, char delim)                                   itree.c:733
 *     size_t XT_doSearch32(UTree*, char *in, char *out, int doCollapse(=8),
 *                          int lv(ignored), int doRC)                          itree.c:833
 *     IXTYPE XT_getIX32(UTree*, WTYPE word)                                    itree.c:720
 *
 * Every entry point below names the seam / lines it replaces.  Conventions:
 *   - plain pointers and sizes only; `d_` = device (HBM) pointer, `h_` = host pointer;
 *   - nothing calls exit(): functions return UTREE_OK or an error code; the CLI (xtree-searchGG) maps the
 *     codes to the reference's exit codes and messages;
 *   - `.ctr` files are consumed unchanged (layout: itree.c:1301-1313 writer, 736-775 reader);
 *   - PACKSIZE / IXTYPE are compile-time in the reference (itree.c:35-70) and run-time here: W in {4,8,16}
 *     (k = 16, 32, 64: every PACKSIZE the reference compiles with, README.md:87-88) and I in {2,4} are dispatched from the file
 *     header; PACKSIZE=16 trees are searched (GG) and compressed, not built or searched rank-specifically;
 *   - there is no CPU fallback: every compute entry point needs a gfx950 device and fails with
 *     UTREE_E_HIP otherwise.
 */
#ifndef UTREE_AMD_H
#define UTREE_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define UTREE_ABI_VERSION 4

enum {
    UTREE_OK = 0,
    UTREE_E_IO = 1,          /* cannot open / read ("Invalid DB file", itree.c:735; "Invalid input files", 835) */
    UTREE_E_FORMAT = 2,      /* "Tree malformatted." (738), short bin table / node dump (768)                   */
    UTREE_E_UNSUPPORTED = 3, /* header names a W / count / I this build has no kernel for (746-751)             */
    UTREE_E_NOMEM = 4,       /* host or device allocation failed (862, 1018)                                    */
    UTREE_E_HIP = 5,         /* HIP runtime error, or no gfx950 device                                          */
    UTREE_E_ARG = 6,         /* bad argument                                                                    */
    UTREE_E_NOLABELS = 7,    /* no label text after the node dump ("No annotation found in tree file.", 776)    */
    UTREE_E_FASTA = 8,       /* malformed read: details in utree_fasta_error (872, 880, 886, 888 -> exit 2)     */
    UTREE_E_RCCL = 9,
    UTREE_E_BUILD = 10,      /* BUILD input rejected: details in utree_build_stats.error_kind                   */
    UTREE_E_DEVICE = 11      /* a batch's kernels found the workspace too small for it (utree_classify_poll)    */
};

const char *utree_strerror(int code);
/* what the calling thread's last UTREE_E_HIP / UTREE_E_DEVICE was: the failing HIP call and the runtime's message for it (no
 * counterpart in the reference, which has no device; "" when there was none) */
const char *utree_last_hip_error(void);
int utree_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Host side of the database: header, bin table, labels.   Replaces XT_read32 (itree.c:733-828) and
 * readSamplesFPdelim (itree.c:1154-1223).  The node dump is NOT copied to host memory: it is streamed
 * from the file to HBM by utree_dev_upload.
 * ---------------------------------------------------------------------------------------------- */
typedef struct utree_ctr utree_ctr;

typedef struct {
    uint32_t W;            /* bytes per packed k-mer word: header[0] (4 => k=16, 8 => k=32, 16 => k=64)  */
    uint32_t I;            /* bytes per label index:       header[2]                                     */
    uint32_t k;            /* 4*W                                                                        */
    uint32_t SZ;           /* bytes per stored record = W + I - 3 (itree.c:691)                           */
    uint64_t n_nodes;      /* header[3]                                                                  */
    uint32_t n_labels;     /* distinct labels, = maxIX (itree.c:855)                                      */
    uint32_t binix_width;  /* 4 iff n_nodes < UINT32_MAX else 8 (itree.c:757)                             */
    uint64_t bin_total;    /* last bin-table entry; the reference only warns if != n_nodes (792-793)      */
    uint64_t file_bytes;
} utree_ctr_info;

int utree_ctr_open(const char *path, utree_ctr **out);
/* Same object from pieces already in memory (synthetic DBs, tests). `binix` has 2^24+1 entries of
 * `binix_width` bytes; `h_records` (n_nodes*SZ bytes) may be NULL when the records are handed over on the
 * device (utree_dev_build).  `label_text` is the file tail verbatim.  Everything given is copied. */
int utree_ctr_from_memory(uint32_t W, uint32_t I, uint64_t n_nodes, const void *binix, uint32_t binix_width,
                          const void *h_records, const char *label_text, size_t label_len, utree_ctr **out);
void utree_ctr_close(utree_ctr *ctr);
int utree_ctr_get_info(const utree_ctr *ctr, utree_ctr_info *info);
/* Label text by file-order index (UTree.SampStrings[ix], itree.c:134,856). NULL when ix >= n_labels. */
const char *utree_ctr_label(const utree_ctr *ctr, uint32_t ix, uint32_t *len);

/* ------------------------------------------------------------------------------------------------
 * Device image: one flat HBM allocation holding the table of 64-byte buckets addressed by (canonical) minimizer hash and orientation, the
 * 8-byte-aligned records, the labels in strcmp order and the rank tables (layout: DESIGN.md §3).  It replaces UTree.Dump / UTree.BinIx
 * (itree.c:140-141) as seen by XT_getIX32.  Because it is flat and position independent, ONE RCCL
 * broadcast replicates a database to the other GPUs of a node.
 * ---------------------------------------------------------------------------------------------- */
typedef struct utree_dev utree_dev;

#define UTREE_FINE_AUTO (-1)

/* Bytes of HBM the image needs for `ctr`.  `fine_bits` (0..8) bounds the width of a bucket from below, 2^(8-fine_bits)
 * hash values: UTREE_FINE_AUTO = 8 unless the table would exceed UTREE_TABLE_MAX_GB (default 96; then the largest value that
 * fits).  k = 64 at fine_bits 8: where one hash value still holds more nodes than a bucket its slot is several pairs of buckets
 *  as long as the table stays within the same cap -- 47 GiB for 568 M 64-mers (8.5 GB on disk). */
size_t utree_dev_image_bytes(const utree_ctr *ctr, int fine_bits);
/* Stream the node dump (from the .ctr file or the host copy given to utree_ctr_from_memory) to `device`
 * and build the image there. */
int utree_dev_upload(const utree_ctr *ctr, int device, int fine_bits, utree_dev **out);
/* seconds of the calling process's last utree_dev_upload: {device + image allocation and labels, node dump file -> pinned memory -> HBM
 * (repacked as it arrives), bin-table check + minimizer sort + buckets, all of it} -- XT_read32's time (itree.c:733-828) has no such split */
int utree_dev_upload_seconds(double *h_out4);
/* Build from raw on-disk pieces that already sit in HBM on `device`: `d_binix` = (2^24+1) entries of
 * ctr's binix_width, `d_records` = n_nodes*SZ packed bytes.  If `d_image` is non-NULL it must have
 * utree_dev_image_bytes() bytes and the image is built in place (caller-owned, e.g. a torch tensor). */
int utree_dev_build(const utree_ctr *ctr, int device, int fine_bits, const void *d_binix, const void *d_records,
                    void *d_image, size_t image_bytes, void *stream, utree_dev **out);
/* The flat image (for ncclBroadcast / torch.distributed.broadcast) ... */
int utree_dev_image(const utree_dev *dev, void **d_image, size_t *bytes);
/* ... and adopting a received copy on another device (not owned by the handle). */
int utree_dev_attach(const utree_ctr *ctr, int device, void *d_image, size_t bytes, utree_dev **out);
void utree_dev_free(utree_dev *dev);

typedef struct {
    uint32_t fine_bits;
    uint32_t record_bytes;      /* bytes per in-HBM record (8, 16 or 24)                                 */
    uint64_t image_bytes;
    uint64_t irregular_bins;    /* bins not strictly ascending (e.g. COMPRESS' first-bin quirk): searched
                                   with the reference's exact probe sequence                             */
    uint32_t generic_mode;      /* 1: bin table not monotone -> every lookup uses the exact probe path   */
    int32_t  device;
    uint32_t vote_table;        /* 1: the image carries the label table the vote decides from (every label
                                   has at most 8 ';'-separated tokens and 255 bytes, 16-bit label indices)   */
    uint32_t lane_pass;         /* 1: the lane-per-read classify kernels take this image (else the
                                   wave-per-read kernels: k = 64 with 32-bit labels, many irregular bins)   */
    uint32_t bucket_bytes;      /* 64 (default) or 128 (UTREE_BUCKET_BYTES=128 when the image is built: a third
                                   less HBM, classify kernels 5-10 % slower); 0: a PACKSIZE=16 tree, whose image is a
                                   direct-address table of all 2^32 words' answers                         */
    uint32_t strand_views;      /* 1: the image stores every k-mer under its mirrored minimizer view too (where that differs), so a
                                   search with RC finds a window and its reverse complement in ONE pass over the read: both are
                                   in the two buckets of one pair                     */
    uint32_t overflow_chains;   /* 1: heavy overflow runs (one minimizer's k-mers in many related genomes) are stored as chains of
                                   consecutive k-mers (k = 32); 0 with UTREE_OVF_CHAINS=0 at build time  (ABI 4) */
    uint32_t pad0;
    uint64_t overflow_bytes;    /* bytes of the image's overflow area                                         (ABI 4) */
} utree_dev_info;
int utree_dev_get_info(const utree_dev *dev, utree_dev_info *info);

/* Replicate dev0's image to the other devices by ncclBroadcast (RCCL over xGMI; pieces of at most 1 GiB) and attach it
 * there: what the reference's worker team gets by sharing one UTree in host memory (itree.c:1009-1018).  devices[0] must be
 * dev0's device, no device twice.  out[0] = dev0; out[1..] own their replicas; on failure nothing is handed back.
 * Rehearsal on one GPU: with UTREE_RCCL_FORCE=1 in the environment n_devices == 1 still builds the communicator and sends the
 * image through ncclBroadcast into a second allocation on the same card, and out[0] is a NEW handle owning that replica
 * (the caller still owns dev0). */
int utree_dev_replicate(const utree_ctr *ctr, utree_dev *dev0, const int *devices, int n_devices, utree_dev **out);
/* seconds the calling process's last utree_dev_replicate / utree_dev_replicate_rank spent between the first ncclBroadcast of
 * the image and the drained streams (0 when it had nothing to send) */
double utree_dev_replicate_seconds(void);
/* What the command line does with n_devices > 1: utree_dev_replicate; if that fails, a warning on stderr and
 * utree_dev_upload(ctr, devices[i], fine_bits) for every other device -- each GPU then reads the database from the host over
 * PCIe.  *how (may be NULL) says which it was. */
enum { UTREE_FANOUT_NONE = 0, UTREE_FANOUT_BROADCAST = 1, UTREE_FANOUT_UPLOAD = 2 };
int utree_dev_fanout(const utree_ctr *ctr, utree_dev *dev0, const int *devices, int n_devices, int fine_bits, utree_dev **out, int *how);
/* the same with the two steps passed in (utree_dev_fanout passes utree_dev_replicate and utree_dev_upload): the seam the
 * fallback branch is tested through on a machine without a GPU */
typedef int (*utree_replicate_fn)(const utree_ctr *, utree_dev *, const int *, int, utree_dev **);
typedef int (*utree_upload_fn)(const utree_ctr *, int, int, utree_dev **);
int utree_dev_fanout_with(const utree_ctr *ctr, utree_dev *dev0, const int *devices, int n_devices, int fine_bits, utree_dev **out, int *how,
                          utree_replicate_fn replicate, utree_upload_fn upload);
/* The same broadcast with one PROCESS per GPU: the root makes an id (utree_rccl_unique_id, UTREE_RCCL_ID_BYTES bytes) and hands
 * it to the other ranks over the launcher's control channel; every rank then calls utree_dev_replicate_rank with its own
 * device.  Root: dev0 = its image, *out = dev0.  Others: dev0 = NULL, `ctr` may describe the database (checked against the
 * image header) or be NULL; *out = a handle that owns the received copy.  UTREE_RCCL_FORCE=1: world == 1 runs the whole
 * sequence on a communicator of one rank and *out owns a replica on the same card, as for utree_dev_replicate. */
#define UTREE_RCCL_ID_BYTES 128
int utree_rccl_unique_id(void *id_out, size_t cap);
int utree_dev_replicate_rank(const utree_ctr *ctr, utree_dev *dev0, int device, int rank, int world, int root, const void *id_bytes,
                             size_t id_len, utree_dev **out);

/* ------------------------------------------------------------------------------------------------
 * The hot path.  Replaces, for a batch of reads, the body of XT_doSearch32's GG branch:
 *   reverse-complement append (itree.c:891-898), k-mer roller XT_WORD_SEARCH (903-933), node lookup
 *   XT_getIX32 (720-730, 699-707), hit filter (929-931), tally (1031-1040), sort (1041), vote (1044-1088).
 * One utree_result per read, in input order.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t label;   /* file-order label index whose text is printed (Tax_Cnt[ed-1] / first hit)          */
    int32_t  cut;     /* -2: whole label; -1: empty taxon; >=0: first `cut` bytes of the label (1087-1088)  */
    uint32_t found;   /* foundUniq; 0 => the reference prints no line for this read (1028)                  */
    uint32_t uix;     /* distinct labels among the hits; 1 => "*" instead of "sl;ol" (1032, 1040)           */
    uint32_t sl, ol;  /* support pair of the last level examined (1071)                                     */
} utree_result;

/* Device workspace a batch needs (tally lists, vote worklist). */
size_t utree_classify_workspace_bytes(const utree_dev *dev, uint32_t n_reads, uint64_t total_bases, uint32_t max_len,
                                      int do_rc);
/* d_bases: raw sequence bytes as they stand in the FASTA (any case, any byte); read r is
 * d_bases[d_off[r] .. d_off[r]+d_len[r]).  total_bases = sum of d_len, max_len = max of d_len (both are
 * by-products of framing; they size the workspace and select the long-read kernel).  Asynchronous on
 * `stream` (a hipStream_t, NULL = default stream); d_out is valid once the stream has drained. */
int utree_classify_batch(utree_dev *dev, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len,
                         uint32_t n_reads, uint64_t total_bases, uint32_t max_len, int do_rc, utree_result *d_out,
                         void *d_workspace, size_t workspace_bytes, void *stream);
/* A batch's kernels check what the workspace's sizing rules out -- (rank, count) lists past their capacity, more long reads or
 * pieces of long reads than the tables hold: total_bases / max_len did not describe the batch -- and report it in an error word
 * that comes back with the batch (no kernel writes past a buffer, none drops a read silently).  Once `stream` has drained,
 * utree_classify_poll returns UTREE_E_DEVICE if a batch finished since the last call reported something (its results are not to
 * be used; utree_last_hip_error says what), else UTREE_OK.  utree_classify_batch also returns UTREE_E_DEVICE when an EARLIER
 * batch's report has arrived by the time it is called (the new batch has been launched all the same; the condition stays until
 * utree_classify_poll has returned it once).  utree_search_file polls after every chunk. */
int utree_classify_poll(utree_dev *dev);
/* The innermost operator alone (XT_getIX32, itree.c:720): words (hi:lo, hi = 0 for k = 32) -> stored label
 * index, 0xFFFFFFFF when absent or when the stored index is >= n_labels.  For tests and micro-benchmarks. */
int utree_lookup_words(utree_dev *dev, const uint64_t *d_hi, const uint64_t *d_lo, uint64_t n, uint32_t *d_ix,
                       void *stream);
/* Name of the dominant kernel as rocprofv3 reports it and the wall time (ms) HIP events measured around
 * its launches since the last call with reset != 0 (bench.py's roofline leg). */
const char *utree_classify_kernel_name(const utree_dev *dev);
/* Measurement aid for the byte model of the bucketed image (bench.py, DESIGN.md sections 4 and 6): over the reads of up to 640 staged
 * bases, h_counts5 = { reads, valid k-mer windows, distinct 64-byte buckets per read (summed), distinct 128-byte HBM lines per
 * read (summed), distinct buckets that carry an overflow descriptor }.  Synchronous.  Evaluated window by window with the
 * load-time minimizer code, independently of the search kernels' sliding minimum. */
int utree_model_counts(utree_dev *dev, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len, uint32_t n_reads,
                       int do_rc, uint64_t *h_counts5, void *stream);
int utree_classify_kernel_time(utree_dev *dev, int reset, double *ms_total, uint64_t *launches);

/* ------------------------------------------------------------------------------------------------
 * Host framing and formatting.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int      code;        /* 0, or the reference's condition: 1 can't read sequence (872), 2 no header '>'
                             (880), 3 sequence begins '>' (886), 4 empty query line (888)                 */
    uint64_t read_index;  /* 1-based read number as the reference prints it                               */
} utree_fasta_error;

/* Frame reads in h_buf[0..n) exactly as XT_INITIATE_WS does with two fgets per read (itree.c:866-890):
 * name = bytes after '>' up to the first space / newline / NUL; sequence = next line minus one '\n' then
 * one '\r'.  When `final` is 0 an incomplete trailing read is left for the next call (*consumed < n).
 * Arrays must hold max_reads entries.  Returns UTREE_OK or UTREE_E_FASTA (reads framed before the bad one
 * are still returned, as the reference classifies them before it exits). */
int utree_fasta_frame(const uint8_t *h_buf, size_t n, int final, size_t max_reads, uint64_t *seq_off,
                      uint32_t *seq_len, uint64_t *name_off, uint32_t *name_len, size_t *n_reads,
                      size_t *consumed, utree_fasta_error *err);
/* Output lines (itree.c:1032, 1040, 1096) for n reads into h_out; reads with found == 0 emit nothing.
 * Returns bytes written, or (size_t)-1 if cap is too small.  *good_finds += lines written (1029). */
size_t utree_format_records(const utree_ctr *ctr, const uint8_t *h_buf, const uint64_t *name_off,
                            const uint32_t *name_len, const utree_result *h_res, size_t n, char *h_out, size_t cap,
                            uint64_t *good_finds);

/* Opt-in input formats the reference does not read.  UTREE_INPUT_REFERENCE is the reference's
 * framing (utree_fasta_frame); the others frame complete records serially and, for multi-line FASTA, compact the sequence
 * lines in place.  Error codes: 1 truncated record, 2 record does not start with '@' / '>', 3 FASTQ separator line is not
 * '+', 5 sequence too long.  With a non-reference format the *_opts file functions also read gzip-compressed input. */
enum { UTREE_INPUT_REFERENCE = 0, UTREE_INPUT_FASTQ = 1, UTREE_INPUT_FASTA_MULTILINE = 2, UTREE_INPUT_AUTO = 3 };
int utree_reads_frame(uint8_t *h_buf, size_t n, int final, int format, size_t max_reads, uint64_t *seq_off,
                      uint32_t *seq_len, uint64_t *name_off, uint32_t *name_len, size_t *n_reads, size_t *consumed,
                      utree_fasta_error *err);

/* ------------------------------------------------------------------------------------------------
 * Whole search = XT_doSearch32(utree, in, out, 8, speed, doRC) (itree.c:833-1108, GG branch), reads
 * sharded over `n_dev` device images, output lines in input order (= the reference with 1 thread).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t n_reads;       /* return value of XT_doSearch32: sequences parsed (itree.c:1107)             */
    uint64_t good_finds;    /* "Good finds: %llu" (1106)                                                  */
    double   seconds_total, seconds_kernels;
    utree_fasta_error fasta_error;
    /* Where the time went.  pipeline 1 = device text pipeline (framing and formatting on the GPU; the host only moves
     * bytes): the figures are LANE-seconds, summed over the n_lanes host threads that each carry one chunk at a time
     * through all stages.  pipeline 0 = host framing / formatting (rank-specific search, opt-in input formats, malformed
     * input): seconds of the four overlapped stage threads; seconds_frame = host framing, seconds_classify_format =
     * H2D + kernels + D2H, seconds_d2h = host formatting. */
    int      pipeline, n_lanes;
    double   seconds_read;              /* file -> pinned memory                                                      */
    double   seconds_frame;             /* H2D of the chunk + newline scan + framing kernels (until the host knows the read count) */
    double   seconds_classify_format;   /* classify + vote + line lengths (until the host knows the text length)      */
    double   seconds_order_wait;        /* waiting for the earlier chunks' text lengths (output offsets are in input order) and for the file (one writer at a time) */
    double   seconds_d2h;               /* format kernel + D2H of the text                                            */
    double   seconds_write;             /* pwrite                                                                     */
    uint64_t bytes_in, bytes_out;
} utree_search_stats;

int utree_search_file(const utree_ctr *ctr, utree_dev **devs, int n_dev, const char *fasta_path,
                      const char *out_path, int do_rc, int host_threads, utree_search_stats *stats);
/* Optional: allocate the search's pinned and device buffers now (they are kept in the device handles and reused by later
 * searches), so that the first search does not pay for them -- "database resident" then includes them. */
int utree_search_prepare(const utree_ctr *ctr, utree_dev **devs, int n_dev, int do_rc);
/* Same with an opt-in input format (UTREE_INPUT_*; AUTO looks at the first byte); gzip input is read through zlib. */
int utree_search_file_opts(const utree_ctr *ctr, utree_dev **devs, int n_dev, const char *reads_path,
                           const char *out_path, int do_rc, int host_threads, int input_format, utree_search_stats *stats);

/* ------------------------------------------------------------------------------------------------
 * Rank-specific search = the `xtree-search` binary (itree.c -D SEARCH: XT_doSearch32 with doCollapse = 0,
 * itree.c:969-1007, 1376).  SURVEY.md §8(f) rank 1.  Same `.ctr`, framing, windows and node lookups as the GG
 * path; different hit selection and vote, with the reference's compile-time knobs as run-time parameters.
 *
 * Reads are NOT independent here: the reference's vote also counts one entry an earlier read left in its hit
 * array (itree.c:982), so batches must be submitted in file order, on one device image per input file; the image
 * carries that array from batch to batch.  utree_rank_reset() starts a new file.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t slack;       /* SLACK (itree.c:955, default 2): most >= slack * secondMost, or no line               */
    uint32_t sparsity;    /* SPARSITY (itree.c:958, default 4): a hit skips PACKSIZE/sparsity - 1 windows (950)   */
    uint32_t tolerance;   /* TOLERANCE_THRESHOLD (itree.c:952, default 2): most >= tolerance, or no line          */
} utree_rank_params;
void utree_rank_params_default(utree_rank_params *p);

size_t utree_rank_workspace_bytes(const utree_dev *dev, uint32_t n_reads, uint64_t total_bases, uint32_t max_len,
                                  int do_rc, const utree_rank_params *params);
/* Arguments as utree_classify_batch.  d_out[r]: found = hits kept (foundUniq, itree.c:930), label = mostIX,
 * sl = most, ol = secondMost (itree.c:986-997), cut = -2 if the reference prints the read (1000-1002) else -4;
 * uix is 0. */
int utree_rank_batch(utree_dev *dev, const uint8_t *d_bases, const uint64_t *d_off, const uint32_t *d_len,
                     uint32_t n_reads, uint64_t total_bases, uint32_t max_len, int do_rc,
                     const utree_rank_params *params, utree_result *d_out, void *d_workspace, size_t workspace_bytes,
                     void *stream);
/* Forget the carried array (a fresh process of the reference). */
int utree_rank_reset(utree_dev *dev);
/* "name \t label \t %f \t %d \n" (itree.c:1002); same conventions as utree_format_records. */
size_t utree_format_rank_records(const utree_ctr *ctr, const uint8_t *h_buf, const uint64_t *name_off,
                                 const uint32_t *name_len, const utree_result *h_res, size_t n, char *h_out, size_t cap,
                                 uint64_t *good_finds);
/* Whole file on ONE device image (the reference runs this branch on one thread); resets the carried array first. */
int utree_rank_search_file(const utree_ctr *ctr, utree_dev *dev, const char *fasta_path, const char *out_path,
                           int do_rc, const utree_rank_params *params, int host_threads, utree_search_stats *stats);
int utree_rank_search_file_opts(const utree_ctr *ctr, utree_dev *dev, const char *reads_path, const char *out_path,
                                int do_rc, const utree_rank_params *params, int host_threads, int input_format,
                                utree_search_stats *stats);

/* ------------------------------------------------------------------------------------------------
 * `.ubt` -> `.ctr` = XT_cmp32(filename, outfile) (itree.c:1234-1315; `xtree-compress`), SURVEY.md §8(f) rank 2.
 * Node dump streamed through `device`; output byte-identical to the reference's, first-bin quirk included.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t n_nodes, n_labels;
    uint64_t label_count_total;   /* "Total nodes in tree: %llu" (itree.c:1314): sum of the label counts */
    uint32_t W, I;
    double   seconds;
} utree_compress_stats;

int utree_compress_file(const char *ubt_path, const char *ctr_path, int device, utree_compress_stats *stats);

/* ------------------------------------------------------------------------------------------------
 * Database BUILD = the `utree-build` / `utree-buildGG` binaries (itree.c -D BUILD / BUILD_GG, main 1379-1407:
 * UT_parseSampFastaExternOSFA + UT_writeTreeBinary + UT_writeSamples).  SURVEY.md §8(f) rank 3.
 * FASTA (one header line + one sequence line per reference) + `name \t label` map -> `<ubt_path>` and
 * `<ubt_path>.gg.log` (gg) / `<ubt_path>.log`, byte-identical to the reference's.  PACKSIZE (4*W), IXTYPE (I bytes) and
 * the compression level (itree.c:595-606) are run-time arguments; `gg` selects BUILD_GG's relabelling of colliding
 * k-mers (itree.c:268-307) instead of BUILD's "collision = BAD" (242-266).  k-mers are extracted, sorted and folded
 * on `device`; everything has to fit its HBM at once (about 36 B per k-mer occurrence plus the FASTA itself).
 * ---------------------------------------------------------------------------------------------- */
enum {
    UTREE_BUILD_E_MAP_EMPTY = 1,  /* "Input map empty." (itree.c:512)                         -> reference exit 1 */
    UTREE_BUILD_E_MAP = 2,        /* malformed map line `error_line` (533-553)                -> exit 2           */
    UTREE_BUILD_E_FASTA = 3,      /* header without sequence, reference `error_line` (585)    -> exit 2           */
    UTREE_BUILD_E_NO_KMERS = 4,   /* "Error: no k-mers. Bad input/params!" (631)              -> exit 2           */
    UTREE_BUILD_E_NAME = 5        /* "Error: taxon map incomplete (line %u)" (582)            -> exit 4           */
};
/* which check of the map parser failed (utree_build_stats.map_error when error_kind == UTREE_BUILD_E_MAP); the command
 * line prints the reference's text for each */
enum {
    UTREE_MAP_E_BLANK_NAME = 1,   /* "ERROR: map line %llu\nBlank indices are NOT ALLOWED." (itree.c:530-533)               */
    UTREE_MAP_E_EXTRA_TAB = 2,    /* "map: extra tab, line %llu" (537)                                                      */
    UTREE_MAP_E_NO_TAB = 3,       /* "Err tab1: %llu" (538)                                                                 */
    UTREE_MAP_E_BLANK_LABEL = 4,  /* "\nERROR: map line %llu\nBlank labels are NOT ALLOWED." (541-544)                      */
    UTREE_MAP_E_NO_NEWLINE = 5    /* "Err line counter: %llu" (548): the text ends inside a label                           */
};
typedef struct {
    uint64_t n_seqs;        /* references parsed (return value of UT_parseSampFastaExternOSFA)                    */
    uint64_t n_kmers;       /* k-mers added, repeats included                                                     */
    uint64_t n_nodes;       /* "Total nodes in tree: %llu" (itree.c:1337)                                         */
    uint64_t n_labels;      /* "[%llu labels]"                                                                    */
    uint64_t error_line;
    int      error_kind;    /* UTREE_BUILD_E_* when the call returns UTREE_E_BUILD (or UTREE_E_IO for MAP_EMPTY)  */
    uint32_t W, I;
    double   seconds;
    uint64_t n_distinct;    /* distinct k-mers, BAD ones included: "Done with sequence parse: %llu k-mers made" (626-630) */
    uint64_t map_bytes, map_lines;   /* "Parsed map. %llu bytes, %llu lines." (510, 515)                                  */
    int      map_error;     /* UTREE_MAP_E_* when error_kind == UTREE_BUILD_E_MAP                                         */
} utree_build_stats;

int utree_build_file(const char *fasta_path, const char *map_path, const char *ubt_path, uint32_t W, uint32_t I,
                     int complevel, int gg, int device, utree_build_stats *stats);

#ifdef __cplusplus
}
#endif
#endif
