/* tools/hostio_probe3.c -- does filling SEVERAL new files at once go faster than filling one?  (one file: ~6 GB/s whatever the thread count,
 * tools/hostio_probe.c -- writers of one file serialise on its inode.)  P threads, each pwrite()s 1 GiB of fresh pages into a file of its own
 * in the given directory (default /dev/shm).   gcc -O2 -fopenmp tools/hostio_probe3.c -o /tmp/hostio_probe3 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv) {
    const char *dir = argc > 1 ? argv[1] : "/dev/shm";
    const size_t per = (size_t)1 << 30, blk = (size_t)8 << 20;
    char *src = malloc(blk);
    memset(src, 'x', blk);
    int Ps[] = {1, 2, 4, 8, 16};
    for (int k = 0; k < 5; ++k) {
        int P = Ps[k];
        double t0 = now();
#pragma omp parallel num_threads(P)
        {
            char path[512];
            snprintf(path, sizeof path, "%s/hostio3_%d.tmp", dir, omp_get_thread_num());
            int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
            for (size_t o = 0; fd >= 0 && o < per; o += blk) if (pwrite(fd, src, blk, (off_t)o) != (ssize_t)blk) break;
            if (fd >= 0) close(fd);
        }
        double dt = now() - t0;
        printf("%2d files at once, 1 GiB each: %.2f GB/s in all\n", P, P * (double)per / dt / 1e9);
        for (int i = 0; i < P; ++i) { char path[512]; snprintf(path, sizeof path, "%s/hostio3_%d.tmp", dir, i); unlink(path); }
    }
    return 0;
}
