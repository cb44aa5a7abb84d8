/* main_build.c -- the `utree-buildGG` (-DUTREE_BUILD_GG) and `utree-build` command lines (itree.c:1379-1407, README.md:71-83):
 *
 *     utree-build[GG] input_fasta.fa labels.map output.ubt threads{0=auto} [complevel]
 *
 * Same positional arguments, same files written (`output.ubt`, `output.ubt[.gg].log`), same exit codes (1 files, 2 malformed
 * map / FASTA / no k-mers, 3 out of memory, 4 name not in the map).  `threads` is accepted and ignored (the reference's
 * parse loop is sequential too, itree.c:575).  The reference's compile-time -D PACKSIZE / -D IXTYPE come from the
 * environment: UTREE_PACKSIZE=32|64 (default 32), UTREE_IXTYPE=16|32 (default 16); UTREE_DEVICE picks the GPU.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../../include/utree_amd.h"

#define VER "[v2.0RF SigNature Edition]"          /* itree.c:1350 */
#ifdef UTREE_BUILD_GG
#define DO_GG 1
#else
#define DO_GG 0
#endif

int main(int argc, char *argv[]) {
    if (argc < 5) {                                                                       /* itree.c:1380-1382 */
        printf(VER " usage: utree-build%s input_fasta.fa labels.map output.ubt threads{0=auto} [complevel]\n", DO_GG ? "GG" : "");
        exit(1);
    }
    printf("This is UTree " VER "\n");
    {                                                                                     /* itree.c:1385-1389, 1395 */
        int threads = atoi(argv[4]);
#ifdef _OPENMP
        if (!threads) threads = omp_get_max_threads();
#else
        if (!threads) threads = 1;
#endif
        printf("Using up to %d threads.\n", threads);
        puts("Tree initialized.");
    }
    unsigned cl = 1;                                                                      /* itree.c:1396-1398 */
    if (argc > 5) cl = (unsigned)atoi(argv[5]);
    printf("Setting compression level to %u\n", cl);
    uint32_t W = 8, I = 2;
    if (getenv("UTREE_PACKSIZE") && atoi(getenv("UTREE_PACKSIZE")) == 64) W = 16;
    if (getenv("UTREE_IXTYPE") && atoi(getenv("UTREE_IXTYPE")) == 32) I = 4;
    int device = getenv("UTREE_DEVICE") ? atoi(getenv("UTREE_DEVICE")) : 0;
    utree_build_stats st;
    int rc = utree_build_file(argv[1], argv[2], argv[3], W, I, (int)cl, DO_GG, device, &st);
    if (rc == UTREE_E_IO && st.error_kind == UTREE_BUILD_E_MAP_EMPTY) { printf("Parsed map. 0 bytes"); puts("\nInput map empty."); exit(1); }   /* itree.c:510-511 */
    if (rc == UTREE_E_IO) { puts("Invalid input file(s)"); exit(1); }                     /* itree.c:504 */
    if (st.map_lines) printf("Parsed map. %llu bytes, %llu lines.\n", (unsigned long long)st.map_bytes, (unsigned long long)st.map_lines);   /* 510, 515 */
    if (rc == UTREE_E_BUILD) {
        const unsigned long long el = (unsigned long long)st.error_line;
        switch (st.error_kind) {
            case UTREE_BUILD_E_MAP:
                switch (st.map_error) {                                                   /* the reference's text per check */
                    case UTREE_MAP_E_BLANK_NAME: printf("ERROR: map line %llu\nBlank indices are NOT ALLOWED.\n", el); break;       /* 531 */
                    case UTREE_MAP_E_EXTRA_TAB: printf("map: extra tab, line %llu\n", el); break;                                  /* 537 */
                    case UTREE_MAP_E_NO_TAB: printf("Err tab1: %llu\n", el); break;                                                /* 538 */
                    case UTREE_MAP_E_BLANK_LABEL: printf("\nERROR: map line %llu\nBlank labels are NOT ALLOWED.\n", el); break;    /* 542 */
                    default: printf("Err line counter: %llu\n", el); break;                                                       /* 548 */
                }
                exit(2);
            case UTREE_BUILD_E_FASTA: printf("Error parsing FASTA (1pass): %llu", (unsigned long long)st.error_line); exit(2);   /* 586 */
            case UTREE_BUILD_E_NO_KMERS: puts("Done with sequence parse: 0 k-mers made"); puts("Error: no k-mers. Bad input/params!"); exit(2);                                    /* 631 */
            case UTREE_BUILD_E_NAME: printf("Error: taxon map incomplete (line %u)\n", (unsigned)st.error_line); exit(4);         /* 582 */
            default: exit(2);
        }
    }
    if (rc) { fprintf(stderr, "ERROR: %s\n", utree_strerror(rc)); exit(3); }
    printf("Done with sequence parse: %llu k-mers made\n", (unsigned long long)st.n_distinct);   /* itree.c:630: distinct k-mers */
    puts("File parsed.");
    printf("Total nodes in tree: %llu [%llu labels]\n", (unsigned long long)st.n_nodes, (unsigned long long)st.n_labels);   /* 1337 */
    puts("Tree written.");
    fprintf(stderr, "[utree_amd] build %.3f s: %llu references, %llu k-mers, %llu nodes\n", st.seconds, (unsigned long long)st.n_seqs,
            (unsigned long long)st.n_kmers, (unsigned long long)st.n_nodes);
    exit(0);
}
