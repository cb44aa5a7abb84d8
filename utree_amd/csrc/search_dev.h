/* search_dev.h -- the device text pipeline of the whole-file search (search_dev.c), private. */
#ifndef UTREE_SEARCH_DEV_H
#define UTREE_SEARCH_DEV_H
#include "utree_internal.h"

/* not an error of the ABI: the input needs the host framing (search.c runs it next) */
#define UTREE_RETRY_HOST 1000

/* When the output cannot seek (a pipe, a FIFO, a tty) the chunks in front of the one that needs the host framing are already out: the
 * host pipeline then CONTINUES -- same descriptor, the input from `in_off` on, the counts so far -- instead of starting over (O_TRUNC does
 * nothing to a pipe: its reader would get the first chunks twice, and closing a FIFO can end its reader).  fo < 0: nothing to continue from,
 * the host pipeline opens the output itself. */
typedef struct { int fo; long long in_off; uint64_t n_reads, good_finds, bytes_in, bytes_out; int parts; } utree_search_resume;
/* parts > 1 (UTREE_OUTPUT_PARTS): the output goes to <out>.part000 ... in input order; a search the host pipeline takes over writes all of it
 * into part 000 and leaves the others empty (the parts' concatenation is the output either way) */
#define UTREE_MAX_OUT_PARTS 64
int utree_output_parts(void);

int utree_search_file_device(const utree_ctr *ctr, utree_dev **devs, int n_dev, const char *fasta_path, const char *out_path,
                             int do_rc, int host_threads, utree_search_stats *stats, uint64_t *progress_printed, utree_search_resume *resume);
/* *progress_printed: "Searched N queries..." lines already on stdout when the call gives up with UTREE_RETRY_HOST: the host
 * pipeline that runs the file again does not print those a second time (the reference prints each once, itree.c:878) */
void utree_search_ctx_free(void *ctx);

#endif
