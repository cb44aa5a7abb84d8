/* search_dev.h -- the device text pipeline of the whole-file search (search_dev.c), private. */
#ifndef UTREE_SEARCH_DEV_H
#define UTREE_SEARCH_DEV_H
#include "utree_internal.h"

/* not an error of the ABI: the input needs the host framing (search.c runs it next) */
#define UTREE_RETRY_HOST 1000

int utree_search_file_device(const utree_ctr *ctr, utree_dev **devs, int n_dev, const char *fasta_path, const char *out_path,
                             int do_rc, int host_threads, utree_search_stats *stats, uint64_t *progress_printed);
/* *progress_printed: "Searched N queries..." lines already on stdout when the call gives up with UTREE_RETRY_HOST: the host
 * pipeline that runs the file again does not print those a second time (the reference prints each once, itree.c:878) */
void utree_search_ctx_free(void *ctx);

#endif
