/* main_searchgg.c -- the `xtree-searchGG` command line, unchanged (itree.c:1357-1377, README.md:5-7):
 *
 *     xtree-searchGG compTree.ctr fastaToSearch.fa output.txt [threads] [SPEED <X>] [RC]
 *
 * and, compiled with -DUTREE_RANK_SPECIFIC, the rank-specific `xtree-search` (itree.c -D SEARCH; README.md:71-76),
 * whose compile-time knobs SLACK / SPARSITY / TOLERANCE_THRESHOLD (itree.c:952-960) are read from the
 * environment here: UTREE_SLACK, UTREE_SPARSITY, UTREE_TOLERANCE (defaults 2, 4, 2).  It uses one GPU.
 *
 * Same positional arguments, same stdout banners, same exit codes (0 bad/malformed DB, 1 usage or input
 * file, 2 malformed read, 3 out of memory / short tree).  New behaviour is reachable only through
 * environment variables so the command line stays bit-compatible:
 *     UTREE_GPUS=<n>        number of GPUs to use (default: all visible)
 *     UTREE_FINE_BITS=<F>   extra prefix bits of the device index (default: auto)
 *     UTREE_INPUT=auto|fastq|fasta   opt-in: FASTQ / multi-line FASTA records, plain or gzip (default: the reference's
 *                           two-lines-per-read framing, bit-compatible)
 *     UTREE_OUTPUT_PARTS=<P> opt-in: the output as P files output.txt.part000 ... (their concatenation is output.txt as the reference writes
 *                           it with one thread); ONE new file fills at ~6 GB/s on a Linux host whatever writes it, P files P times that
 * `threads` sizes the host formatting team (the GPU does the search).  `SPEED` is parsed and ignored, as
 * in the reference (itree.c:858, 907-918).
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../../include/utree_amd.h"

#define VER "[v2.0RF SigNature Edition]"          /* itree.c:1350 */
#ifdef UTREE_RANK_SPECIFIC
#define DO_GG 0
#else
#define DO_GG 1
#endif
static const char *TYPEARR[17] = {"NA", "uint8_t", "uint16_t", "NA", "uint32_t", "NA", "NA", "NA", "uint64_t", "NA", "NA",
                                  "NA", "NA", "NA", "NA", "NA", "__uint128_t"};

int main(int argc, char *argv[]) {
    if (argc < 4) {                                                                       /* itree.c:1358-1360 */
        printf(VER " usage: xtree-search%s compTree.ctr fastaToSearch.fa output.txt [threads] [SPEED <X>] [RC]\n", DO_GG ? "GG" : "");
        exit(1);
    }
    printf("This is UTree " VER "\n");
    int doRC = !strcmp(argv[argc - 1], "RC"), threads = 1;                                /* itree.c:1362-1364 */
    argc -= doRC;
    int speed = 0;
    if (!strcmp(argv[argc - 2], "SPEED")) speed = atoi(argv[argc - 1]), argc -= 2;
    printf("Reverse complement consideration is %sabled.\n", doRC ? "en" : "dis");
    printf("Searching at speed %d.\n", speed);
#ifdef _OPENMP
    threads = argc >= 5 ? atoi(argv[4]) : omp_get_max_threads();                          /* itree.c:1368-1370 */
#else
    threads = argc >= 5 ? atoi(argv[4]) : 1;
#endif
    printf("Using up to %d threads.\n", threads);

    utree_ctr *ctr = NULL;
    int rc = utree_ctr_open(argv[1], &ctr);
    if (rc == UTREE_E_IO) { puts("Invalid DB file"); exit(0); }                          /* itree.c:735 */
    if (rc == UTREE_E_FORMAT) { puts("Tree malformatted."); exit(0); }                   /* itree.c:738 */
    if (rc == UTREE_E_UNSUPPORTED) {
        /* the reference's own words for a tree its build does not read (itree.c:746-751): what the header asks for */
        uint64_t md[4] = {0, 0, 0, 0};
        FILE *dp = fopen(argv[1], "rb");
        if (dp) { if (fread(md, sizeof *md, 4, dp) != 4) md[0] = 0; fclose(dp); }
        printf("ERROR. Input tree requires PACKSIZE=%u, CNTTYPE=%s, IXTYPE=%s\n", (unsigned)(md[0] << 2), md[1] <= 16 ? TYPEARR[md[1]] : "NA", md[2] <= 16 ? TYPEARR[md[2]] : "NA");
        exit(0);
    }
    if (rc == UTREE_E_NOLABELS) { puts("No annotation found in tree file."); exit(0); }   /* itree.c:776 */
    if (rc) { fprintf(stderr, "%s\n", utree_strerror(rc)); exit(3); }
    utree_ctr_info ci;
    utree_ctr_get_info(ctr, &ci);
    if (ci.binix_width == 4) puts("Using 32-bit counters");                              /* itree.c:754-755 */
    else puts("Holey smokes, a tree of over 4 billion k-mers. Here goes...");
    printf("%llu elements read.\n", (unsigned long long)((1u << 24) + 1));                /* itree.c:761 */
    printf("Nodes in input tree: %llu (PACKSIZE=%u, CNTTYPE=%s, IXTYPE=%s, SZ=%d)\n", (unsigned long long)ci.n_nodes,
           ci.W << 2, TYPEARR[0], TYPEARR[ci.I], (int)ci.SZ);                             /* itree.c:764-765 */

    int n_vis = 0;
    if (hipGetDeviceCount(&n_vis) != hipSuccess || n_vis < 1) { fputs("ERROR: no gfx950 device visible\n", stderr); exit(3); }
    int n_dev = n_vis;
    const char *eg = getenv("UTREE_GPUS");
    if (eg && atoi(eg) > 0 && atoi(eg) < n_dev) n_dev = atoi(eg);
    if (!DO_GG) n_dev = 1;                                                                /* reads depend on their predecessors */
    utree_dev **devs = (utree_dev **)calloc((size_t)n_dev, sizeof(utree_dev *));
    int *ids = (int *)calloc((size_t)n_dev, sizeof(int));
    for (int i = 0; i < n_dev; ++i) ids[i] = i;
    rc = utree_dev_upload(ctr, 0, UTREE_FINE_AUTO, &devs[0]);
    if (rc == UTREE_E_FORMAT) { puts("Error in reading tree."); exit(3); }                /* itree.c:768 */
    if (rc) { fprintf(stderr, "ERROR: device image: %s\n", utree_strerror(rc)); exit(3); }
    printf("Read %llu nodes.\n", (unsigned long long)ci.n_nodes);                         /* itree.c:769 */
    if (ci.bin_total != ci.n_nodes)                                                       /* itree.c:792-793 */
        printf("Warning: detected nodes %u != %u\n", (unsigned)ci.bin_total, (unsigned)ci.n_nodes);
    utree_dev *built = devs[0];
    int how = UTREE_FANOUT_NONE;
    rc = utree_dev_fanout(ctr, built, ids, n_dev, UTREE_FINE_AUTO, devs, &how);           /* RCCL broadcast; on failure every GPU loads over PCIe */
    if (rc) { fprintf(stderr, "ERROR: tree on %d GPUs: %s\n", n_dev, utree_strerror(rc)); exit(3); }
    if (how == UTREE_FANOUT_BROADCAST)
        fprintf(stderr, "[utree_amd] tree replicated to %d GPU(s) by RCCL broadcast in %.3f s\n", n_dev, utree_dev_replicate_seconds());
#ifndef UTREE_RANK_SPECIFIC
    { int prc = utree_search_prepare(ctr, devs, n_dev, doRC);
      if (prc) fprintf(stderr, "[utree_amd] warning: the search buffers could not be allocated ahead (%s); the search allocates them itself\n", utree_strerror(prc)); }                                   /* the search's pinned / device buffers: part of "database resident" */
#endif
    puts("Tree read.");                                                                   /* itree.c:826 */
    utree_dev_info di;
    utree_dev_get_info(devs[0], &di);
    fprintf(stderr, "[utree_amd] %d GPU(s), image %.2f GiB, fine_bits=%u, irregular bins=%llu%s\n", n_dev,
            (double)di.image_bytes / 1073741824.0, di.fine_bits, (unsigned long long)di.irregular_bins,
            di.generic_mode ? " (generic mode)" : "");
    fflush(stdout);

    utree_search_stats st;
    int fmt = UTREE_INPUT_REFERENCE;
    const char *ei = getenv("UTREE_INPUT");
    if (ei && !strcmp(ei, "auto")) fmt = UTREE_INPUT_AUTO;
    else if (ei && !strcmp(ei, "fastq")) fmt = UTREE_INPUT_FASTQ;
    else if (ei && !strcmp(ei, "fasta")) fmt = UTREE_INPUT_FASTA_MULTILINE;
#ifdef UTREE_RANK_SPECIFIC
    utree_rank_params prm;
    utree_rank_params_default(&prm);
    if (getenv("UTREE_SLACK")) prm.slack = (uint32_t)atoi(getenv("UTREE_SLACK"));
    if (getenv("UTREE_SPARSITY")) prm.sparsity = (uint32_t)atoi(getenv("UTREE_SPARSITY"));
    if (getenv("UTREE_TOLERANCE")) prm.tolerance = (uint32_t)atoi(getenv("UTREE_TOLERANCE"));
    rc = utree_rank_search_file_opts(ctr, devs[0], argv[2], argv[3], doRC, &prm, threads, fmt, &st);
#else
    rc = utree_search_file_opts(ctr, devs, n_dev, argv[2], argv[3], doRC, threads, fmt, &st);
#endif
    if (rc == UTREE_E_IO) { puts("Invalid input files"); exit(1); }                      /* itree.c:835 */
    if (rc == UTREE_E_FASTA) {
        switch (st.fasta_error.code) {                                                    /* itree.c:872, 880, 886, 888 */
            case 1: fprintf(stderr, "ERROR: can't read sequence L %llu\n", (unsigned long long)st.fasta_error.read_index); break;
            case 2: fprintf(stderr, "ERROR: no header '>' [L %llu]\n", (unsigned long long)st.fasta_error.read_index); break;
            case 3: fprintf(stderr, "ERROR: sequence begins '>' [L %llu]\n", (unsigned long long)st.fasta_error.read_index); break;
            case 4: fprintf(stderr, "ERROR: empty query line %llu\n", (unsigned long long)st.fasta_error.read_index); break;
            default: fprintf(stderr, "ERROR: query line too long\n");
        }
        exit(2);
    }
    if (rc) { fprintf(stderr, "ERROR: %s\n", utree_strerror(rc)); exit(3); }
    printf("Good finds: %llu\n", (unsigned long long)st.good_finds);                      /* itree.c:1106 */
    printf("Searched %llu queries\n", (unsigned long long)st.n_reads);                    /* itree.c:1375 */
    fprintf(stderr, "[utree_amd] search %.3f s (%.0f reads/s), GPU batches %.3f s%s\n", st.seconds_total,
            st.seconds_total > 0 ? (double)st.n_reads / st.seconds_total : 0.0, st.seconds_kernels,
            st.pipeline ? " (lane-seconds; framing and formatting on the GPU)" : "");
    for (int i = n_dev - 1; i >= 0; --i) utree_dev_free(devs[i]);
    if (devs[0] != built) utree_dev_free(built);                                          /* UTREE_RCCL_FORCE: devs[0] was a replica */
    utree_ctr_close(ctr);
    exit(0);
}
