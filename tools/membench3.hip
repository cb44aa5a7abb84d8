// membench3.hip -- calibration for a minimizer-bucketed layout: lanes of a wave share table lines in groups.
//   per iteration every lane does: index load (8 B) -> dependent bucket load (16 B)
//   GROUP = number of consecutive lanes that share the same index slot / bucket (1 = today's random pattern)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
template <int GROUP, bool DEP>
__global__ __launch_bounds__(256) void gather_k(const uint64_t *__restrict__ idx, const uint64_t *__restrict__ rec, uint64_t mask_idx,
                                                uint64_t mask_rec, uint64_t per_thread, uint64_t *__restrict__ out) {
    uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t gid = tid / GROUP;
    uint64_t acc = 0, s = gid * 0x9E3779B97F4A7C15ull + 1;
    for (uint64_t i = 0; i < per_thread; ++i) {
        s = mix(s + i);
        uint64_t e = __builtin_nontemporal_load(idx + (s & mask_idx));
        acc += e;
        if (DEP) {
            uint64_t b = ((e ^ s) & mask_rec) & ~1ull;
            const ulonglong2 v = *(const ulonglong2 *)(rec + b);
            acc += v.x ^ v.y;
        }
    }
    out[tid] = acc;
}
int main() {
    size_t ib = (size_t)32 << 30, rb = (size_t)16 << 30;
    uint64_t *idx, *rec, *out;
    if (hipMalloc(&idx, ib) != hipSuccess || hipMalloc(&rec, rb) != hipSuccess) return 1;
    hipMemset(idx, 1, ib); hipMemset(rec, 2, rb);
    const int blocks = 256 * 8, threads = 256;
    hipMalloc(&out, (size_t)blocks * threads * 8);
    uint64_t per_thread = 256;
#define RUN(G, D) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); \
    gather_k<G, D><<<blocks, threads>>>(idx, rec, ib / 8 - 1, rb / 8 - 1, per_thread, out); hipDeviceSynchronize(); \
    hipEventRecord(e0); gather_k<G, D><<<blocks, threads>>>(idx, rec, ib / 8 - 1, rb / 8 - 1, per_thread, out); hipEventRecord(e1); hipEventSynchronize(e1); \
    float ms; hipEventElapsedTime(&ms, e0, e1); \
    printf("group %2d  %s  %8.3f ms  %8.2f G lane-lookups/s\n", G, D ? "index+bucket" : "index only  ", ms, (double)blocks * threads * per_thread / ms / 1e6); }
    RUN(1, false) RUN(2, false) RUN(4, false) RUN(8, false) RUN(16, false) RUN(64, false)
    RUN(1, true) RUN(2, true) RUN(4, true) RUN(8, true) RUN(16, true)
    return 0;
}
