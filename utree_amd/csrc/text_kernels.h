/* text_kernels.h -- launchers of text_kernels.hip (device-side read framing and output formatting), private. */
#ifndef UTREE_TEXT_KERNELS_H
#define UTREE_TEXT_KERNELS_H
#include <stddef.h>
#include <stdint.h>
#include "utree_internal.h"
#ifdef __cplusplus
extern "C" {
#endif

/* flags: input the device pipeline does not take (search.c then re-runs the file through the host framing) */
#define UTK_TEXT_NUL        1u     /* a NUL byte: strlen / the name scan would stop there (itree.c:881, 887)        */
#define UTK_TEXT_NO_HEADER  2u     /* "ERROR: no header '>'" (itree.c:880)                                          */
#define UTK_TEXT_SEQ_HEADER 4u     /* "ERROR: sequence begins '>'" (itree.c:886)                                    */
#define UTK_TEXT_LONG_LINE  8u     /* a line fgets(…, LINELEN) would split (itree.c:836, 869-871)                   */
#define UTK_TEXT_BAD_LABEL  16u    /* a result names a label the database does not have (cannot happen)              */

/* small per-chunk record the kernels fill and the host reads back (pinned copy) */
typedef struct {
    uint32_t n_lines;               /* newlines in the chunk                                                        */
    uint32_t flags;
    uint32_t max_len;               /* longest sequence                                                             */
    uint32_t pad;
    unsigned long long total_bases;
    unsigned long long good_finds;  /* reads that print a line (itree.c:1029)                                       */
    unsigned long long out_bytes;   /* bytes of the chunk's output text                                             */
} utk_text_meta;

size_t utk_text_scan_temp_bytes(uint32_t max_reads);
/* zeroes *d_meta; d_counts: one uint32 per 4096 input bytes; d_nl[i] = position of newline i (i < max_lines) */
int utk_text_newlines(const uint8_t *d_buf, uint64_t n, uint32_t *d_counts, uint32_t *d_nl, uint32_t max_lines, utk_text_meta *d_meta,
                      void *stream);
/* reads = d_meta->n_lines / 2 (read on the device), at most max_reads; grid_reads = the most reads the chunk's bytes can hold */
int utk_text_frame(const uint8_t *d_buf, const uint32_t *d_nl, uint32_t max_reads, uint32_t grid_reads, uint64_t *d_seq_off,
                   uint32_t *d_seq_len, uint32_t *d_name_off, uint32_t *d_name_len, utk_text_meta *d_meta, void *stream);
/* phase 0: line lengths, their exclusive prefix, meta->out_bytes and meta->good_finds; phase 1: the text into d_out */
int utk_text_format(const utk_image *im, const uint32_t *d_ix2rank, const uint8_t *d_buf, const utree_result *d_res,
                    const uint32_t *d_name_off, const uint32_t *d_name_len, uint32_t n_reads, uint32_t *d_line_len,
                    uint64_t *d_line_off, void *d_scan_tmp, size_t scan_tmp_bytes, uint8_t *d_out, uint64_t out_cap,
                    utk_text_meta *d_meta, int phase, void *stream);

#ifdef __cplusplus
}
#endif
#endif
