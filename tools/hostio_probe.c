/* hostio_probe.c -- what the GPU box's host side can move (measurement tool, not part of the product):
 * page-cache reads into pinned memory by thread count, pinned H2D / D2H, hipHostRegister of a mapped tmpfs file,
 * parallel pwrite.  Sizes the stages of csrc/search.c.
 *
 *   gcc -O2 -fopenmp -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/hostio_probe.c -o tools/hostio_probe -L/opt/rocm/lib -lamdhip64
 */
#define _GNU_SOURCE
#define _FILE_OFFSET_BITS 64
#include <hip/hip_runtime_api.h>
#include <fcntl.h>
#include <omp.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); } } while (0)

int main(int argc, char **argv) {
    const char *dir = argc > 1 ? argv[1] : "/dev/shm";
    size_t total = (size_t)(argc > 2 ? atof(argv[2]) : 4.0) * ((size_t)1 << 30);
    const size_t CH = (size_t)96 << 20;
    char path[512], path2[512];
    snprintf(path, sizeof path, "%s/hostio_probe.bin", dir);
    snprintf(path2, sizeof path2, "%s/hostio_probe.out", dir);
    cpu_set_t cs; CPU_ZERO(&cs); sched_getaffinity(0, sizeof cs, &cs);
    printf("online cpus %ld, affinity %d, omp max threads %d\n", sysconf(_SC_NPROCESSORS_ONLN), CPU_COUNT(&cs), omp_get_max_threads());
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) { char b[128]; if (fgets(b, sizeof b, f)) printf("cgroup cpu.max: %s", b); fclose(f); }
    f = fopen("/sys/fs/cgroup/memory.max", "r");
    if (f) { char b[128]; if (fgets(b, sizeof b, f)) printf("cgroup memory.max: %s", b); fclose(f); }
    uint8_t *pin[4];
    for (int i = 0; i < 4; ++i) CK(hipHostMalloc((void **)&pin[i], CH, hipHostMallocDefault));
    /* write the file (parallel pwrite, also the write measurement) */
    for (int i = 0; i < 4; ++i) memset(pin[i], 'A' + i, CH);
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { perror("open"); return 1; }
    size_t nch = total / CH;
    for (int T = 1; T <= 32; T *= 2) {
        double t0 = now_s();
#pragma omp parallel for num_threads(T) schedule(dynamic, 1)
        for (size_t c = 0; c < nch * 4; ++c) {
            size_t off = c * (CH / 4), done = 0;
            while (done < CH / 4) { ssize_t w = pwrite(fd, pin[c & 3] + done, CH / 4 - done, (off_t)(off + done)); if (w <= 0) break; done += (size_t)w; }
        }
        double dt = now_s() - t0;
        printf("pwrite  %2d threads: %6.2f GB/s%s\n", T, (double)(nch * CH) / dt / 1e9, T == 1 ? " (first pass allocates the pages)" : "");
    }
    for (int T = 1; T <= 32; T *= 2) {
        double t0 = now_s();
#pragma omp parallel for num_threads(T) schedule(dynamic, 1)
        for (size_t c = 0; c < nch * 8; ++c) {
            size_t off = c * (CH / 8), done = 0;
            uint8_t *dst = pin[(c / 8) & 3] + (c & 7) * (CH / 8);
            while (done < CH / 8) { ssize_t r = pread(fd, dst + done, CH / 8 - done, (off_t)(off + done)); if (r <= 0) break; done += (size_t)r; }
        }
        double dt = now_s() - t0;
        printf("pread -> pinned %2d threads: %6.2f GB/s\n", T, (double)(nch * CH) / dt / 1e9);
    }
    /* plain memcpy pinned -> pinned, for comparison */
    for (int T = 1; T <= 32; T *= 4) {
        double t0 = now_s();
        for (int rep = 0; rep < 8; ++rep) {
#pragma omp parallel for num_threads(T) schedule(static, 1)
            for (int c = 0; c < 64; ++c) memcpy(pin[1] + (size_t)c * (CH / 64), pin[0] + (size_t)c * (CH / 64), CH / 64);
        }
        printf("memcpy pinned->pinned %2d threads: %6.2f GB/s\n", T, 8.0 * CH / (now_s() - t0) / 1e9);
    }
    void *d = NULL, *d2 = NULL;
    CK(hipMalloc(&d, CH)); CK(hipMalloc(&d2, CH));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipMemcpy(d, pin[0], CH, hipMemcpyHostToDevice));
    double t0 = now_s();
    for (int i = 0; i < 16; ++i) CK(hipMemcpyAsync(d, pin[i & 3], CH, hipMemcpyHostToDevice, s1));
    CK(hipStreamSynchronize(s1));
    printf("H2D pinned, one stream: %6.2f GB/s\n", 16.0 * CH / (now_s() - t0) / 1e9);
    t0 = now_s();
    for (int i = 0; i < 16; ++i) CK(hipMemcpyAsync(pin[i & 3], d, CH, hipMemcpyDeviceToHost, s1));
    CK(hipStreamSynchronize(s1));
    printf("D2H pinned, one stream: %6.2f GB/s\n", 16.0 * CH / (now_s() - t0) / 1e9);
    t0 = now_s();
    for (int i = 0; i < 16; ++i) { CK(hipMemcpyAsync(d, pin[i & 1], CH, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(pin[2 + (i & 1)], d2, CH, hipMemcpyDeviceToHost, s2)); }
    CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
    printf("H2D + D2H concurrently: %6.2f GB/s each way\n", 16.0 * CH / (now_s() - t0) / 1e9);
    /* mapped file registered with the runtime: DMA straight from the page cache */
    size_t map_bytes = total < ((size_t)2 << 30) ? total : ((size_t)2 << 30);
    void *m = mmap(NULL, map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) perror("mmap");
    else {
        t0 = now_s();
        hipError_t e = hipHostRegister(m, map_bytes, hipHostRegisterDefault);
        double tr = now_s() - t0;
        printf("hipHostRegister(mmap MAP_SHARED tmpfs, %.1f GiB): %s, %.3f s (%.2f GB/s)\n", map_bytes / 1073741824.0, hipGetErrorString(e), tr, map_bytes / tr / 1e9);
        if (e == hipSuccess) {
            t0 = now_s();
            size_t nn = map_bytes / CH;
            for (size_t i = 0; i < nn; ++i) CK(hipMemcpyAsync(d, (char *)m + i * CH, CH, hipMemcpyHostToDevice, s1));
            CK(hipStreamSynchronize(s1));
            printf("H2D from the registered mapping: %6.2f GB/s\n", (double)nn * CH / (now_s() - t0) / 1e9);
            t0 = now_s();
            CK(hipHostUnregister(m));
            printf("hipHostUnregister: %.3f s\n", now_s() - t0);
        } else (void)hipGetLastError();
        munmap(m, map_bytes);
    }
    m = mmap(NULL, map_bytes, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m != MAP_FAILED) {
        t0 = now_s();
        hipError_t e = hipHostRegister(m, map_bytes, hipHostRegisterDefault);
        double tr = now_s() - t0;
        printf("hipHostRegister(mmap MAP_PRIVATE read-only, %.1f GiB): %s, %.3f s\n", map_bytes / 1073741824.0, hipGetErrorString(e), tr);
        if (e == hipSuccess) {
            t0 = now_s();
            size_t nn = map_bytes / CH;
            for (size_t i = 0; i < nn; ++i) CK(hipMemcpyAsync(d, (char *)m + i * CH, CH, hipMemcpyHostToDevice, s1));
            CK(hipStreamSynchronize(s1));
            printf("H2D from the registered private mapping: %6.2f GB/s\n", (double)nn * CH / (now_s() - t0) / 1e9);
            CK(hipHostUnregister(m));
        } else (void)hipGetLastError();
        munmap(m, map_bytes);
    }
    /* pageable H2D (what hipMemcpy does with an unregistered mapping) */
    {
        void *p = malloc(CH); memset(p, 1, CH);
        t0 = now_s();
        for (int i = 0; i < 4; ++i) CK(hipMemcpy(d, p, CH, hipMemcpyHostToDevice));
        printf("H2D pageable: %6.2f GB/s\n", 4.0 * CH / (now_s() - t0) / 1e9);
        free(p);
    }
    close(fd);
    unlink(path); unlink(path2);
    return 0;
}
