/* hostio_probe2.c -- how fast can ONE output file in tmpfs be filled?  (measurement tool, not part of the product)
 * pwrite serialises on the inode lock (hostio_probe: ~9 GB/s whatever the thread count); alternatives measured here:
 * memcpy into a fresh MAP_SHARED mapping by T threads (page faults), the same after MADV_POPULATE_WRITE / fallocate,
 * hipHostRegister of the fresh mapping + D2H straight into the page cache, hipHostMalloc cost.
 *
 *   gcc -O2 -fopenmp -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/hostio_probe2.c -o tools/hostio_probe2 -L/opt/rocm/lib -lamdhip64
 */
#define _GNU_SOURCE
#define _FILE_OFFSET_BITS 64
#include <hip/hip_runtime_api.h>
#include <errno.h>
#include <fcntl.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); } } while (0)

static int fresh(const char *path, size_t bytes, int prealloc) {
    unlink(path);
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { perror("open"); exit(1); }
    if (prealloc) { if (posix_fallocate(fd, 0, (off_t)bytes)) perror("fallocate"); }
    else if (ftruncate(fd, (off_t)bytes)) perror("ftruncate");
    return fd;
}

int main(int argc, char **argv) {
    const char *dir = argc > 1 ? argv[1] : "/dev/shm";
    const size_t total = (size_t)3 << 30, CH = (size_t)96 << 20, nch = total / CH;
    char path[512];
    snprintf(path, sizeof path, "%s/hostio_probe2.out", dir);
    uint8_t *pin[2];
    double t0 = now_s();
    CK(hipHostMalloc((void **)&pin[0], CH, hipHostMallocDefault));
    double t1 = now_s();
    CK(hipHostMalloc((void **)&pin[1], 4 * CH, hipHostMallocDefault));
    double t2 = now_s();
    printf("hipHostMalloc: first %.0f MiB %.3f s (incl. runtime init), next %.0f MiB %.3f s (%.2f GB/s)\n", CH / 1048576.0, t1 - t0, 4 * CH / 1048576.0,
           t2 - t1, 4.0 * CH / (t2 - t1) / 1e9);
    memset(pin[0], 'x', CH);
    void *d = NULL;
    CK(hipMalloc(&d, CH));
    CK(hipMemset(d, 'y', CH));
    hipStream_t s1;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    for (int T = 1; T <= 16; T *= 2) {
        int fd = fresh(path, total, 0);
        uint8_t *m = (uint8_t *)mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        t0 = now_s();
#pragma omp parallel for num_threads(T) schedule(dynamic, 1)
        for (size_t c = 0; c < nch * 8; ++c) memcpy(m + c * (CH / 8), pin[0] + (c & 7) * (CH / 8), CH / 8);
        printf("memcpy into a fresh mapping, %2d threads: %6.2f GB/s\n", T, (double)total / (now_s() - t0) / 1e9);
        munmap(m, total); close(fd);
    }
    for (int T = 4; T <= 16; T *= 2) {
        int fd = fresh(path, total, 0);
        uint8_t *m = (uint8_t *)mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        t0 = now_s();
        int bad = 0;
#pragma omp parallel for num_threads(T) schedule(dynamic, 1) reduction(| : bad)
        for (size_t c = 0; c < nch * 8; ++c) if (madvise(m + c * (CH / 8), CH / 8, MADV_POPULATE_WRITE)) bad |= 1;
        double tp = now_s() - t0;
        t0 = now_s();
#pragma omp parallel for num_threads(T) schedule(dynamic, 1)
        for (size_t c = 0; c < nch * 8; ++c) memcpy(m + c * (CH / 8), pin[0] + (c & 7) * (CH / 8), CH / 8);
        double tc = now_s() - t0;
        printf("MADV_POPULATE_WRITE %2d threads: %s %6.2f GB/s, then memcpy %6.2f GB/s, together %6.2f GB/s\n", T, bad ? "(failed)" : "", total / tp / 1e9,
               total / tc / 1e9, total / (tp + tc) / 1e9);
        munmap(m, total); close(fd);
    }
    {
        t0 = now_s();
        int fd = fresh(path, total, 1);
        printf("posix_fallocate %.1f GiB: %.3f s (%.2f GB/s)\n", total / 1073741824.0, now_s() - t0, total / (now_s() - t0) / 1e9);
        uint8_t *m = (uint8_t *)mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        t0 = now_s();
#pragma omp parallel for num_threads(8) schedule(dynamic, 1)
        for (size_t c = 0; c < nch * 8; ++c) memcpy(m + c * (CH / 8), pin[0] + (c & 7) * (CH / 8), CH / 8);
        printf("memcpy into the preallocated mapping, 8 threads: %6.2f GB/s\n", (double)total / (now_s() - t0) / 1e9);
        munmap(m, total); close(fd);
    }
    {   /* register the fresh mapping and let the copy engine write the page cache */
        int fd = fresh(path, total, 0);
        uint8_t *m = (uint8_t *)mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        t0 = now_s();
        hipError_t e = hipHostRegister(m, total, hipHostRegisterDefault);
        double tr = now_s() - t0;
        printf("hipHostRegister(fresh MAP_SHARED mapping, %.1f GiB): %s, %.3f s (%.2f GB/s)\n", total / 1073741824.0, hipGetErrorString(e), tr, total / tr / 1e9);
        if (e == hipSuccess) {
            t0 = now_s();
            for (size_t i = 0; i < nch; ++i) CK(hipMemcpyAsync(m + i * CH, d, CH, hipMemcpyDeviceToHost, s1));
            CK(hipStreamSynchronize(s1));
            printf("D2H into the registered mapping: %6.2f GB/s (first byte %c)\n", (double)total / (now_s() - t0) / 1e9, m[0]);
            t0 = now_s();
            CK(hipHostUnregister(m));
            printf("hipHostUnregister: %.3f s\n", now_s() - t0);
        } else (void)hipGetLastError();
        munmap(m, total); close(fd);
    }
    {   /* the same in pieces registered by several threads */
        for (int T = 2; T <= 8; T *= 2) {
            int fd = fresh(path, total, 0);
            uint8_t *m = (uint8_t *)mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            t0 = now_s();
            int bad = 0;
#pragma omp parallel for num_threads(T) schedule(dynamic, 1) reduction(| : bad)
            for (size_t c = 0; c < nch; ++c) if (hipHostRegister(m + c * CH, CH, hipHostRegisterDefault) != hipSuccess) bad |= 1;
            double tr = now_s() - t0;
            printf("hipHostRegister in %zu pieces by %d threads: %s %.3f s (%.2f GB/s)\n", nch, T, bad ? "(failed)" : "", tr, total / tr / 1e9);
            for (size_t c = 0; c < nch; ++c) (void)hipHostUnregister(m + c * CH);
            (void)hipGetLastError();
            munmap(m, total); close(fd);
        }
    }
    {   /* input side: register an existing file's mapping in pieces by several threads */
        int fd = fresh(path, total, 0);
        uint8_t *m = (uint8_t *)mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
#pragma omp parallel for num_threads(8) schedule(dynamic, 1)
        for (size_t c = 0; c < nch * 8; ++c) memset(m + c * (CH / 8), 'z', CH / 8);
        munmap(m, total);
        for (int T = 1; T <= 8; T *= 2) {
            m = (uint8_t *)mmap(NULL, total, PROT_READ, MAP_SHARED, fd, 0);
            t0 = now_s();
            int bad = 0;
#pragma omp parallel for num_threads(T) schedule(dynamic, 1) reduction(| : bad)
            for (size_t c = 0; c < nch; ++c) if (hipHostRegister(m + c * CH, CH, hipHostRegisterReadOnly) != hipSuccess) bad |= 1;
            double tr = now_s() - t0;
            printf("existing file: hipHostRegister(read-only) in %zu pieces by %d threads: %s %.3f s (%.2f GB/s)\n", nch, T, bad ? "(failed)" : "", tr, total / tr / 1e9);
            if (!bad && T == 1) {
                t0 = now_s();
                for (size_t i = 0; i < nch; ++i) CK(hipMemcpyAsync(d, m + i * CH, CH, hipMemcpyHostToDevice, s1));
                CK(hipStreamSynchronize(s1));
                printf("H2D from it: %6.2f GB/s\n", (double)total / (now_s() - t0) / 1e9);
            }
            for (size_t c = 0; c < nch; ++c) (void)hipHostUnregister(m + c * CH);
            (void)hipGetLastError();
            munmap(m, total);
        }
        close(fd);
    }
    unlink(path);
    return 0;
}
