// membench.hip -- calibration micro-benchmark (not part of the product): how fast can the MI355X serve
// random small reads from a table much larger than the Infinity Cache, and at what granularity?
//   A  one 8-byte load per random 128-B line
//   B  two 8-byte loads in the SAME random 128-B line (offsets 0 and 64)
//   C  two 8-byte loads in two DIFFERENT random lines (independent)
//   D  two DEPENDENT loads (second address derived from the first value)
//   E  one 16-byte load per random line; F one 64-byte (4 x 16 B) per random 64-B-aligned sector
// build: hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
template <int MODE>
__global__ __launch_bounds__(256) void gather_k(const uint64_t *__restrict__ tab, uint64_t lines_mask, uint64_t per_thread,
                                                uint64_t *__restrict__ out) {
    uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t acc = 0, s = tid * 0x9E3779B97F4A7C15ull + 1;
    for (uint64_t i = 0; i < per_thread; ++i) {
        s = mix(s + i);
        uint64_t line = s & lines_mask;                  // 128-B line index
        const uint64_t *p = tab + line * 16;
        if (MODE == 0) acc += p[0];
        if (MODE == 1) acc += p[0] + p[8];
        if (MODE == 2) { uint64_t l2 = mix(s) & lines_mask; acc += p[0] + tab[l2 * 16]; }
        if (MODE == 3) { uint64_t v = p[0]; uint64_t l2 = (v ^ s) & lines_mask; acc += tab[l2 * 16 + 1]; }
        if (MODE == 4) { const ulonglong2 *q = (const ulonglong2 *)p; ulonglong2 v = q[0]; acc += v.x + v.y; }
        if (MODE == 6) { const ulonglong2 *q = (const ulonglong2 *)(tab + (mix((tid >> 2) * 0x9E3779B97F4A7C15ull + i + 77) & lines_mask) * 16 + ((mix(tid >> 2) >> 40) & 1) * 8) + (tid & 3); ulonglong2 a = q[0]; acc += a.x + a.y; }
        if (MODE == 7) { const uint64_t b0 = (mix((tid >> 2) * 0x9E3779B97F4A7C15ull + i * 4 + 77)); 
            ulonglong2 a[4];
            for (int k = 0; k < 4; ++k) { const ulonglong2 *q = (const ulonglong2 *)(tab + (mix(b0 + k) & lines_mask) * 16) + (tid & 3); a[k] = q[0]; }
            acc += a[0].x + a[1].y + a[2].x + a[3].y; }
        if (MODE == 8) { const ulonglong2 *q = (const ulonglong2 *)(tab + (mix((tid >> 3) * 0x9E3779B97F4A7C15ull + i + 77) & lines_mask) * 16) + (tid & 7); ulonglong2 a = q[0]; acc += a.x + a.y; }
        if (MODE == 9) { const ulonglong2 *q = (const ulonglong2 *)(tab + (mix((tid >> 2) * 0x9E3779B97F4A7C15ull + i + 77) & lines_mask) * 16) + (tid & 3); ulonglong2 a = q[0], b = q[4]; acc += a.x + b.y; }
        if (MODE == 5) { const ulonglong2 *q = (const ulonglong2 *)(p + ((s >> 40) & 1) * 8); ulonglong2 a = q[0], b = q[1], c = q[2], d = q[3]; acc += a.x + b.y + c.x + d.y; }
    }
    out[tid] = acc;
}

int main(int argc, char **argv) {
    size_t gib = argc > 1 ? atoi(argv[1]) : 16;
    size_t bytes = gib << 30;
    uint64_t *tab, *out;
    hipMalloc(&tab, bytes);
    hipMemset(tab, 1, bytes);
    const int blocks = 256 * 8, threads = 256;
    hipMalloc(&out, (size_t)blocks * threads * 8);
    uint64_t lines_mask = bytes / 128 - 1;
    uint64_t per_thread = 512;
    const char *names[10] = {"A 1x8B/line", "B 2x8B same 128B line", "C 2x8B two lines", "D 2 dependent loads", "E 1x16B/line", "F 64B sector", "G 64B sector per QUAD (x4 iters)", "H 4 sectors per quad, 4 loads in flight",
                             "I 128B line per OCT (x8 iters), one load", "J 128B line per QUAD (x4 iters), two loads"};
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 10; ++m) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            switch (m) {
                case 0: gather_k<0><<<blocks, threads>>>(tab, lines_mask, per_thread, out); break;
                case 1: gather_k<1><<<blocks, threads>>>(tab, lines_mask, per_thread, out); break;
                case 2: gather_k<2><<<blocks, threads>>>(tab, lines_mask, per_thread, out); break;
                case 3: gather_k<3><<<blocks, threads>>>(tab, lines_mask, per_thread, out); break;
                case 4: gather_k<4><<<blocks, threads>>>(tab, lines_mask, per_thread, out); break;
                case 5: gather_k<5><<<blocks, threads>>>(tab, lines_mask, per_thread, out); break;
                case 6: gather_k<6><<<blocks, threads>>>(tab, lines_mask, per_thread, out); break;
                case 7: gather_k<7><<<blocks, threads>>>(tab, lines_mask, per_thread, out); break;
                case 8: gather_k<8><<<blocks, threads>>>(tab, lines_mask, per_thread, out); break;
                case 9: gather_k<9><<<blocks, threads>>>(tab, lines_mask, per_thread, out); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double iters = (double)blocks * threads * per_thread;
            if (rep) printf("%-44s %8.3f ms  %7.2f G iters/s\n", names[m], ms, iters / ms / 1e6);
        }
    return 0;
}
