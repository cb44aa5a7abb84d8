// membench4.hip -- what rate of random 64-byte sectors does the lane pass's fetch structure reach when NOTHING else runs?  Persistent wavefronts
// (W per SIMD, 168-register budget emulated by launch bounds), each loop step a "batch": every lane requests four 16-byte quarters (the quad
// pattern of classify_lanes_k: lane j of a quad takes quarter j of four buckets), D batches in flight while one is consumed.
// usage: membench4 [GiB]   build: hipcc -O3 --offload-arch=gfx950 tools/membench4.hip -o /tmp/membench4
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
template <int D, int WPS>
__global__ __launch_bounds__(256, WPS) void batches_k(const uint8_t *__restrict__ tab, uint64_t sectors_mask, uint32_t batches, uint64_t *__restrict__ out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    uint64_t s = (uint64_t)(tid >> 2) * 0x9E3779B97F4A7C15ull + 1;          // one stream of bucket indices per quad
    u32x4 P[D][4];
    uint32_t acc = 0;
    auto issue = [&](u32x4 (&p)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { s = mix(s + k); p[k] = *(const u32x4 *)(tab + ((s & sectors_mask) << 6) + 16u * (lane & 3u)); }
    };
    auto consume = [&](const u32x4 (&p)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += p[k].x ^ p[k].y ^ p[k].z ^ p[k].w;
    };
#pragma unroll
    for (int d = 0; d < D - 1; ++d) issue(P[d]);
    for (uint32_t b = 0; b < batches; b += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) { issue(P[(d + D - 1) % D]); consume(P[d]); }
    }
    out[tid] = acc;
}
template <int D, int WPS> static void run(const uint8_t *tab, uint64_t mask, uint64_t *out, int n_cu) {
    const uint32_t batches = 1200 / D * D;
    const int blocks = n_cu * WPS;                                           // 4 waves per block: WPS waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        batches_k<D, WPS><<<blocks, 256>>>(tab, mask, batches, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double sectors = (double)blocks * 64 /* quads */ * 4 * (batches + D - 1);
        if (rep) printf("%d waves per SIMD, %d batches of 64 sectors in flight per wave: %7.3f ms  %6.2f G sectors/s\n", WPS, D - 1, ms, sectors / ms / 1e6);
    }
}
int main(int argc, char **argv) {
    size_t gib = argc > 1 ? atoi(argv[1]) : 16;
    size_t bytes = gib << 30;
    uint8_t *tab; uint64_t *out;
    hipMalloc(&tab, bytes); hipMemset(tab, 1, bytes);
    hipMalloc(&out, (size_t)256 * 8 * 256 * 8 * 4);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const uint64_t mask = bytes / 64 - 1;
    run<2, 3>(tab, mask, out, pr.multiProcessorCount);
    run<3, 3>(tab, mask, out, pr.multiProcessorCount);
    run<4, 3>(tab, mask, out, pr.multiProcessorCount);
    run<6, 3>(tab, mask, out, pr.multiProcessorCount);
    run<3, 2>(tab, mask, out, pr.multiProcessorCount);
    run<3, 1>(tab, mask, out, pr.multiProcessorCount);
    run<6, 1>(tab, mask, out, pr.multiProcessorCount);
    run<3, 4>(tab, mask, out, pr.multiProcessorCount);
    run<3, 8>(tab, mask, out, pr.multiProcessorCount);
    return 0;
}
