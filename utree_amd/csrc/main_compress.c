/* main_compress.c -- `xtree-compress preTree.ubt compTree.ctr`, unchanged command line (itree.c:1352-1355). */
#include <stdio.h>
#include <stdlib.h>
#include "../../include/utree_amd.h"
#define VER "[v2.0RF SigNature Edition]"
static const char *TYPEARR[17] = {"NA", "uint8_t", "uint16_t", "NA", "uint32_t", "NA", "NA", "NA", "uint64_t", "NA", "NA",
                                  "NA", "NA", "NA", "NA", "NA", "__uint128_t"};
int main(int argc, char *argv[]) {
    if (argc != 3) { puts(VER " usage: xtree-compress preTree.ubt compTree.ctr"); exit(1); }
    utree_compress_stats st = {0};
    int rc = utree_compress_file(argv[1], argv[2], 0, &st);
    if (rc == UTREE_E_IO) { puts("Invalid input filename"); exit(0); }               /* itree.c:1236 / 1299 */
    if (rc == UTREE_E_FORMAT) { puts("Tree malformatted."); exit(0); }               /* itree.c:1239 */
    if (rc == UTREE_E_UNSUPPORTED) {                                                 /* the reference's words for a tree its build does not read (itree.c:1247-1251) */
        unsigned long long md[4] = {0, 0, 0, 0};
        FILE *dp = fopen(argv[1], "rb");
        if (dp) { if (fread(md, sizeof *md, 4, dp) != 4) md[0] = 0; fclose(dp); }
        printf("ERROR. Input tree requires PACKSIZE=%u, CNTTYPE=%s, IXTYPE=%s\n", (unsigned)(md[0] << 2), md[1] <= 16 ? TYPEARR[md[1]] : "NA", md[2] <= 16 ? TYPEARR[md[2]] : "NA");
        exit(0);
    }
    if (rc) { fprintf(stderr, "ERROR: %s\n", utree_strerror(rc)); exit(3); }
    printf("Nodes in input tree: %llu (PACKSIZE=%u, CNTTYPE=%s, IXTYPE=%s, el=%u)\n", (unsigned long long)st.n_nodes, st.W << 2,
           TYPEARR[0], TYPEARR[st.I], st.W + st.I);                                  /* itree.c:1254 */
    if (st.n_nodes < 0xFFFFFFFFull) puts("Using 32-bit counters");                   /* itree.c:1278 */
    printf("Total nodes in tree: %llu [%llu labels]\n", (unsigned long long)st.label_count_total, (unsigned long long)st.n_labels);
    fprintf(stderr, "[utree_amd] compress %.3f s (%.1f M nodes/s)\n", st.seconds, st.seconds > 0 ? st.n_nodes / st.seconds / 1e6 : 0.0);
    exit(0);
}
