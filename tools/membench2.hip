// membench2.hip -- calibration: does a cache-policy modifier make a random 8-byte read cost less than a
// whole 128-B line of HBM traffic?  (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
template <int MODE> __device__ __forceinline__ uint64_t ld(const uint64_t *p) {
    uint64_t v;
    if (MODE == 0) v = *p;
    if (MODE == 1) v = __builtin_nontemporal_load(p);
    if (MODE == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 5) asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 6) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MODE>
__global__ __launch_bounds__(256) void gather_k(const uint64_t *__restrict__ tab, uint64_t mask8, uint64_t per_thread, uint64_t *__restrict__ out) {
    uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t acc = 0, s = tid * 0x9E3779B97F4A7C15ull + 1;
    for (uint64_t i = 0; i < per_thread; ++i) {
        s = mix(s + i);
        acc += ld<MODE>(tab + (s & mask8));              // random 8-byte word anywhere in the table
    }
    out[tid] = acc;
}
int main(int argc, char **argv) {
    size_t gib = argc > 1 ? atoi(argv[1]) : 32;
    size_t bytes = gib << 30;
    uint64_t *tab, *out;
    if (hipMalloc(&tab, bytes) != hipSuccess) return 1;
    hipMemset(tab, 1, bytes);
    const int blocks = 256 * 8, threads = 256;
    hipMalloc(&out, (size_t)blocks * threads * 8);
    uint64_t mask8 = bytes / 8 - 1, per_thread = 512;
    const char *names[7] = {"plain", "nontemporal builtin", "sc1", "sc0 sc1", "sc0 sc1 nt", "nt", "sc0"};
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 7; ++m) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            switch (m) {
                case 0: gather_k<0><<<blocks, threads>>>(tab, mask8, per_thread, out); break;
                case 1: gather_k<1><<<blocks, threads>>>(tab, mask8, per_thread, out); break;
                case 2: gather_k<2><<<blocks, threads>>>(tab, mask8, per_thread, out); break;
                case 3: gather_k<3><<<blocks, threads>>>(tab, mask8, per_thread, out); break;
                case 4: gather_k<4><<<blocks, threads>>>(tab, mask8, per_thread, out); break;
                case 5: gather_k<5><<<blocks, threads>>>(tab, mask8, per_thread, out); break;
                case 6: gather_k<6><<<blocks, threads>>>(tab, mask8, per_thread, out); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-22s %8.3f ms  %7.2f G loads/s\n", names[m], ms, (double)blocks * threads * per_thread / ms / 1e6);
        }
    return 0;
}
