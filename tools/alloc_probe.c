/* tools/alloc_probe.c -- what a fresh device allocation costs on this box: hipMalloc, first touch (hipMemset), second touch, hipFree, for a few sizes,
 * twice in one process.  gcc -O2 tools/alloc_probe.c -I/opt/rocm/include -L/opt/rocm/lib -lamdhip64 -o /tmp/alloc_probe */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <time.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(void) {
    double t0 = now();
    hipSetDevice(0); hipFree(0);
    printf("runtime init %.3f s\n", now() - t0);
    size_t gib[] = {8, 32, 64, 64, 32, 100};
    for (int i = 0; i < 6; ++i) {
        void *p = NULL;
        size_t n = gib[i] << 30;
        double a = now();
        if (hipMalloc(&p, n) != hipSuccess) { printf("%zu GiB: hipMalloc failed\n", gib[i]); continue; }
        double b = now();
        hipMemset(p, 0, n); hipDeviceSynchronize();
        double c = now();
        hipMemset(p, 1, n); hipDeviceSynchronize();
        double d = now();
        hipFree(p);
        double e = now();
        printf("%3zu GiB: hipMalloc %.3f s, first memset %.3f s, second memset %.3f s, hipFree %.3f s\n", gib[i], b - a, c - b, d - c, e - d);
    }
    return 0;
}
