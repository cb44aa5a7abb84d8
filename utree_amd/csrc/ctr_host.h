/* ctr_host.h -- private definition of utree_ctr (host side of a database). */
#ifndef UTREE_CTR_HOST_H
#define UTREE_CTR_HOST_H
#include "utree_internal.h"

struct utree_ctr {
    utree_ctr_info info;
    uint64_t hdr_W_raw, hdr_cnt_raw, hdr_I_raw;
    char *path;                 /* file the node dump is streamed from (NULL for from_memory)             */
    uint64_t records_file_off;
    void *binix_raw;            /* 2^24+1 entries at the on-disk width                                     */
    uint64_t bins_read;
    uint8_t *h_records;         /* optional host copy (from_memory)                                        */
    char *label_text;           /* file tail; labels[] point into it (TABs/newlines replaced by NUL)       */
    size_t label_text_len;
    char **labels;              /* file order = label index                                                */
    uint32_t *label_len;
    uint32_t *rank2ix, *ix2rank;/* strcmp order <-> file order                                             */
};

#endif
