/* build_gpu.h -- interface between build.c (host: map, FASTA framing, label universe, label numbering, file assembly)
 * and build_gpu.hip (device pipeline).  Private. */
#ifndef UTREE_BUILD_GPU_H
#define UTREE_BUILD_GPU_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    const uint8_t *h_fa; uint64_t fa_bytes;                 /* the FASTA file as it stands                              */
    const uint64_t *h_seq_off; const uint32_t *h_seq_len;   /* [n_refs] sequence line of every reference (itree.c:588-590) */
    const uint32_t *h_ref_u; uint32_t n_refs;               /* universe id of every reference's label                   */
    const char *h_ublob; uint64_t ublob_bytes;              /* universe: every label and every ';'-prefix of one, NUL-terminated */
    const uint64_t *h_uoff; uint32_t n_u;
    const uint32_t *h_trunc_off;                            /* [n_u + 1]                                                */
    const uint32_t *h_trunc_ids; uint32_t n_trunc;          /* trunc_ids[trunc_off[u] + m - 1] = u cut before its m-th ';' */
    uint32_t W, I, lv;
    int gg, device;
} utk_build_job;

typedef struct {
    uint64_t total_pos, n_occ, n_nodes;  /* positions examined, k-mers added (with repeats), nodes kept                 */
    uint64_t n_distinct;                 /* distinct k-mers, BAD ones included: the reference's "k-mers made" (itree.c:626-630) */
    uint32_t n_passes;
    uint64_t *h_first_time;              /* [n_u] 2*position+1 of the first collision that produced the label, or ~0    */
    uint64_t *h_ref_time;                /* [n_refs] 2*(positions before reference r): when its own label is created    */
} utk_build_result;

typedef struct utk_build_state utk_build_state;
int utk_build_phase1(const utk_build_job *job, utk_build_result *res, utk_build_state **state);
int utk_build_phase2(utk_build_state *state, const uint32_t *h_ix_of_u, uint32_t n_u, uint32_t n_labels, int fd,
                     uint64_t *h_per_label);
void utk_build_free(utk_build_state *state);

#ifdef __cplusplus
}
#endif
#endif
