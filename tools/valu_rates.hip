// Issue rate of a few VALU instructions on gfx950 (how many cycles a SIMD spends per wave64 instruction).
// build: hipcc -O3 --offload-arch=gfx950 tools/valu_rates.hip -o gpurun_out/valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define REP 64
#define ITER 2000
template <int OP> __global__ __launch_bounds__(256) void rate_k(uint32_t *out, uint32_t seed) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3u + 1u, a2 = a0 ^ 0x1234567u, a3 = a0 + 77u;
    uint32_t b0 = a1 ^ 5u, b1 = a2 + 9u, b2 = a3 * 7u, b3 = a0 ^ 0xABCDEu;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (OP == 0) {   // v_add_u32
                asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                             "v_add_u32 %0, %0, %5\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
            } else if (OP == 1) {   // v_mul_lo_u32
                asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n"
                             "v_mul_lo_u32 %0, %0, %5\n v_mul_lo_u32 %1, %1, %5\n v_mul_lo_u32 %2, %2, %5\n v_mul_lo_u32 %3, %3, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
            } else if (OP == 2) {   // v_mul_u32_u24
                asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4\n"
                             "v_mul_u32_u24 %0, %0, %5\n v_mul_u32_u24 %1, %1, %5\n v_mul_u32_u24 %2, %2, %5\n v_mul_u32_u24 %3, %3, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
            } else if (OP == 3) {   // v_lshrrev_b64
                asm volatile("v_lshrrev_b64 %0, 3, %0\n v_lshrrev_b64 %1, 5, %1\n v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 2, %1\n"
                             "v_lshrrev_b64 %0, 3, %0\n v_lshrrev_b64 %1, 5, %1\n v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 2, %1"
                             : "+v"(*(uint64_t *)&a0), "+v"(*(uint64_t *)&a2));
            } else if (OP == 4) {   // v_cmp_lt_u64 (+ nothing else)
                asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %1, %0\n v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %1, %0\n"
                             "v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %1, %0\n v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %1, %0"
                             : : "v"(*(uint64_t *)&a0), "v"(*(uint64_t *)&a2) : "vcc");
            } else if (OP == 5) {   // v_perm_b32
                asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5\n"
                             "v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
            } else if (OP == 6) {   // v_lshl_add_u64
                asm volatile("v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %1, %1, 1, %0\n v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %1, %1, 1, %0\n"
                             "v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %1, %1, 1, %0\n v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %1, %1, 1, %0"
                             : "+v"(*(uint64_t *)&a0), "+v"(*(uint64_t *)&a2));
            } else if (OP == 7) {   // v_cndmask_b32 with vcc
                asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                             "v_cndmask_b32 %0, %0, %5, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %5, vcc\n v_cndmask_b32 %3, %3, %5, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc");
            } else if (OP == 8) {   // v_mov_b32 with DPP row_shr:1
                asm volatile("v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %0, %4 row_shr:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %5 row_shr:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %5 row_shr:2 row_mask:0xf bank_mask:0xf"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
            } else if (OP == 9) {
                asm volatile("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4\n v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc", "s10", "s11");
            } else if (OP == 10) {
                asm volatile("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5\n v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc", "s10", "s11");
            } else if (OP == 11) {
                asm volatile("v_bfi_b32 %0, %4, %0, %5\n v_bfi_b32 %1, %4, %1, %5\n v_bfi_b32 %2, %4, %2, %5\n v_bfi_b32 %3, %4, %3, %5\n v_bfi_b32 %0, %4, %0, %5\n v_bfi_b32 %1, %4, %1, %5\n v_bfi_b32 %2, %4, %2, %5\n v_bfi_b32 %3, %4, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc", "s10", "s11");
            } else if (OP == 12) {
                asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc", "s10", "s11");
            } else if (OP == 13) {
                asm volatile("v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc", "s10", "s11");
            } else if (OP == 14) {
                asm volatile("v_cmp_eq_u32 vcc, %0, %4\n v_cmp_eq_u32 vcc, %1, %4\n v_cmp_eq_u32 vcc, %2, %4\n v_cmp_eq_u32 vcc, %3, %4\n v_cmp_eq_u32 vcc, %0, %4\n v_cmp_eq_u32 vcc, %1, %4\n v_cmp_eq_u32 vcc, %2, %4\n v_cmp_eq_u32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc", "s10", "s11");
            } else if (OP == 15) {
                asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc", "s10", "s11");
            } else if (OP == 16) {
                asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %4, s[10:11]\n v_cndmask_b32_e64 %2, %2, %4, s[10:11]\n v_cndmask_b32_e64 %3, %3, %4, s[10:11]\n v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %4, s[10:11]\n v_cndmask_b32_e64 %2, %2, %4, s[10:11]\n v_cndmask_b32_e64 %3, %3, %4, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc", "s10", "s11");
            } else if (OP == 17) {
                asm volatile("v_alignbit_b32 %0, %0, %4, 7\n v_alignbit_b32 %1, %1, %4, 7\n v_alignbit_b32 %2, %2, %4, 7\n v_alignbit_b32 %3, %3, %4, 7\n v_alignbit_b32 %0, %0, %4, 7\n v_alignbit_b32 %1, %1, %4, 7\n v_alignbit_b32 %2, %2, %4, 7\n v_alignbit_b32 %3, %3, %4, 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc", "s10", "s11");
            } else if (OP == 18) {
                asm volatile("v_or3_b32 %0, %0, %4, %5\n v_or3_b32 %1, %1, %4, %5\n v_or3_b32 %2, %2, %4, %5\n v_or3_b32 %3, %3, %4, %5\n v_or3_b32 %0, %0, %4, %5\n v_or3_b32 %1, %1, %4, %5\n v_or3_b32 %2, %2, %4, %5\n v_or3_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc", "s10", "s11");
            } else if (OP == 19) {
                asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_u32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_lt_u32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %5, vcc\n v_cmp_lt_u32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %5, vcc\n v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_u32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_lt_u32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %5, vcc\n v_cmp_lt_u32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %5, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc", "s10", "s11");
            } else if (OP == 20) {
                asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5\n v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc", "s10", "s11");
            } else if (OP == 21) {
                asm volatile("v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %4\n v_max_u32 %2, %2, %4\n v_max_u32 %3, %3, %4\n v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %4\n v_max_u32 %2, %2, %4\n v_max_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc", "s10", "s11");
            } else if (OP == 22) {
                asm volatile("v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4\n v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc", "s10", "s11");
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ b2 ^ b3;
}
template <int OP> static void run(const char *name, uint32_t *d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;                       // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    rate_k<OP><<<blocks, 256>>>(d, 1);
    hipEventRecord(e0);
    rate_k<OP><<<blocks, 256>>>(d, 2);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 8 waves x ITER x REP instructions
    const double inst = 8.0 * ITER * REP, cyc = ms * 1e-3 * 2.4e9;
    printf("%-16s %8.3f ms   %.2f cycles per wave-instruction at 2.4 GHz\n", name, ms, cyc / inst);
}
int main() {
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_u32", d); run<1>("v_mul_lo_u32", d); run<2>("v_mul_u32_u24", d); run<3>("v_lshrrev_b64", d);
    run<4>("v_cmp_lt_u64", d); run<5>("v_perm_b32", d); run<6>("v_lshl_add_u64", d); run<7>("v_cndmask_b32", d); run<8>("v_mov_b32_dpp", d); run<9>("v_min_u32", d); run<10>("v_and_or_b32", d); run<11>("v_bfi_b32", d); run<12>("v_xor_b32", d); run<13>("v_lshrrev_b32", d); run<14>("v_cmp_eq_u32", d); run<15>("v_cndmask dst!=src", d); run<16>("v_cndmask e64 sgpr", d); run<17>("v_alignbit_b32", d); run<18>("v_or3_b32", d); run<19>("cmp+cndmask pair", d); run<20>("v_mad_u32_u24", d); run<21>("v_max_u32", d); run<22>("v_sub_u32", d);
    return 0;
}
