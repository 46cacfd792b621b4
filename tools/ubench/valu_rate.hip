// Micro-benchmark: issue rate of the VALU opcodes the scans are made of (gfx950).  One workgroup of W waves per CU-slot,
// each wave runs N iterations of 8 independent instructions of one kind; cycles measured with s_memtime.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
template <int KIND>
__global__ void k(unsigned *out, unsigned long long *cyc, int iters) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned b = out[0], c = out[1];
    unsigned long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {
            asm volatile("v_bitop3_b32 %0, %0, %8, %9 bitop3:0xde\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0xde\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0xde\n"
                         "v_bitop3_b32 %3, %3, %8, %9 bitop3:0xde\n v_bitop3_b32 %4, %4, %8, %9 bitop3:0xde\n v_bitop3_b32 %5, %5, %8, %9 bitop3:0xde\n"
                         "v_bitop3_b32 %6, %6, %8, %9 bitop3:0xde\n v_bitop3_b32 %7, %7, %8, %9 bitop3:0xde\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 1) {
            asm volatile("v_or3_b32 %0, %0, %8, %9\n v_or3_b32 %1, %1, %8, %9\n v_or3_b32 %2, %2, %8, %9\n v_or3_b32 %3, %3, %8, %9\n"
                         "v_or3_b32 %4, %4, %8, %9\n v_or3_b32 %5, %5, %8, %9\n v_or3_b32 %6, %6, %8, %9\n v_or3_b32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 2) {
            asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                         "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 3) {
            asm volatile("v_alignbit_b32 %0, %0, %8, 31\n v_alignbit_b32 %1, %1, %8, 31\n v_alignbit_b32 %2, %2, %8, 31\n v_alignbit_b32 %3, %3, %8, 31\n"
                         "v_alignbit_b32 %4, %4, %8, 31\n v_alignbit_b32 %5, %5, %8, 31\n v_alignbit_b32 %6, %6, %8, 31\n v_alignbit_b32 %7, %7, %8, 31\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 4) {
            asm volatile("v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n"
                         "v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 5) {
            asm volatile("v_lshlrev_b32_sdwa %0, %8, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                         "v_lshlrev_b32_sdwa %1, %8, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                         "v_lshlrev_b32_sdwa %2, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                         "v_lshlrev_b32_sdwa %3, %8, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                         "v_lshlrev_b32_sdwa %4, %8, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                         "v_lshlrev_b32_sdwa %5, %8, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                         "v_lshlrev_b32_sdwa %6, %8, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                         "v_lshlrev_b32_sdwa %7, %8, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 6) {   // dependent chain of bitop3 (latency)
            asm volatile("v_bitop3_b32 %0, %0, %8, %9 bitop3:0xde\n v_bitop3_b32 %0, %0, %8, %9 bitop3:0xde\n v_bitop3_b32 %0, %0, %8, %9 bitop3:0xde\n"
                         "v_bitop3_b32 %0, %0, %8, %9 bitop3:0xde\n v_bitop3_b32 %0, %0, %8, %9 bitop3:0xde\n v_bitop3_b32 %0, %0, %8, %9 bitop3:0xde\n"
                         "v_bitop3_b32 %0, %0, %8, %9 bitop3:0xde\n v_bitop3_b32 %0, %0, %8, %9 bitop3:0xde\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 7) {   // dependent chain of v_and (latency)
            asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %0, %0, %8\n v_and_b32 %0, %0, %8\n v_and_b32 %0, %0, %8\n"
                         "v_and_b32 %0, %0, %8\n v_and_b32 %0, %0, %8\n v_and_b32 %0, %0, %8\n v_and_b32 %0, %0, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 8) {
            asm volatile("v_cmp_lt_i32 vcc, %0, %8\n v_addc_co_u32 %0, vcc, %0, %0, vcc\nv_cmp_lt_i32 vcc, %1, %8\n v_addc_co_u32 %1, vcc, %1, %1, vcc\nv_cmp_lt_i32 vcc, %2, %8\n v_addc_co_u32 %2, vcc, %2, %2, vcc\nv_cmp_lt_i32 vcc, %3, %8\n v_addc_co_u32 %3, vcc, %3, %3, vcc\nv_cmp_lt_i32 vcc, %4, %8\n v_addc_co_u32 %4, vcc, %4, %4, vcc\nv_cmp_lt_i32 vcc, %5, %8\n v_addc_co_u32 %5, vcc, %5, %5, vcc\nv_cmp_lt_i32 vcc, %6, %8\n v_addc_co_u32 %6, vcc, %6, %6, vcc\nv_cmp_lt_i32 vcc, %7, %8\n v_addc_co_u32 %7, vcc, %7, %7, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "memory", "vcc");
        } else if (KIND == 9) {
            asm volatile("v_add_co_u32 %0, vcc, %0, %0\n v_addc_co_u32 %0, vcc, 0, %0, vcc\nv_add_co_u32 %1, vcc, %1, %1\n v_addc_co_u32 %1, vcc, 0, %1, vcc\nv_add_co_u32 %2, vcc, %2, %2\n v_addc_co_u32 %2, vcc, 0, %2, vcc\nv_add_co_u32 %3, vcc, %3, %3\n v_addc_co_u32 %3, vcc, 0, %3, vcc\nv_add_co_u32 %4, vcc, %4, %4\n v_addc_co_u32 %4, vcc, 0, %4, vcc\nv_add_co_u32 %5, vcc, %5, %5\n v_addc_co_u32 %5, vcc, 0, %5, vcc\nv_add_co_u32 %6, vcc, %6, %6\n v_addc_co_u32 %6, vcc, 0, %6, vcc\nv_add_co_u32 %7, vcc, %7, %7\n v_addc_co_u32 %7, vcc, 0, %7, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "memory", "vcc");
        } else if (KIND == 10) {
            asm volatile("v_add_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\nv_add_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\nv_add_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\nv_add_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\nv_add_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\nv_add_u32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\nv_add_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\nv_add_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "memory");
        } else if (KIND == 11) {
            asm volatile("v_min_i32 %0, %0, %8\nv_min_i32 %1, %1, %8\nv_min_i32 %2, %2, %8\nv_min_i32 %3, %3, %8\nv_min_i32 %4, %4, %8\nv_min_i32 %5, %5, %8\nv_min_i32 %6, %6, %8\nv_min_i32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "memory");
        } else if (KIND == 12) {
            asm volatile("v_lshlrev_b32 %0, 1, %0\nv_lshlrev_b32 %1, 1, %1\nv_lshlrev_b32 %2, 1, %2\nv_lshlrev_b32 %3, 1, %3\nv_lshlrev_b32 %4, 1, %4\nv_lshlrev_b32 %5, 1, %5\nv_lshlrev_b32 %6, 1, %6\nv_lshlrev_b32 %7, 1, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "memory");
        } else if (KIND == 13) {
            asm volatile("v_lshl_add_u32 %0, %0, 2, %8\nv_lshl_add_u32 %1, %1, 2, %8\nv_lshl_add_u32 %2, %2, 2, %8\nv_lshl_add_u32 %3, %3, 2, %8\nv_lshl_add_u32 %4, %4, 2, %8\nv_lshl_add_u32 %5, %5, 2, %8\nv_lshl_add_u32 %6, %6, 2, %8\nv_lshl_add_u32 %7, %7, 2, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "memory");
        } else if (KIND == 14) {
            asm volatile("v_sub_u32 %0, %0, %8\nv_sub_u32 %1, %1, %8\nv_sub_u32 %2, %2, %8\nv_sub_u32 %3, %3, %8\nv_sub_u32 %4, %4, %8\nv_sub_u32 %5, %5, %8\nv_sub_u32 %6, %6, %8\nv_sub_u32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "memory");
        } else if (KIND == 15) {
            asm volatile("v_bfe_u32 %0, %0, 8, 8\nv_bfe_u32 %1, %1, 8, 8\nv_bfe_u32 %2, %2, 8, 8\nv_bfe_u32 %3, %3, 8, 8\nv_bfe_u32 %4, %4, 8, 8\nv_bfe_u32 %5, %5, 8, 8\nv_bfe_u32 %6, %6, 8, 8\nv_bfe_u32 %7, %7, 8, 8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "memory");
        } else if (KIND == 16) {
            asm volatile("v_and_or_b32 %0, %0, %8, %9\nv_and_or_b32 %1, %1, %8, %9\nv_and_or_b32 %2, %2, %8, %9\nv_and_or_b32 %3, %3, %8, %9\nv_and_or_b32 %4, %4, %8, %9\nv_and_or_b32 %5, %5, %8, %9\nv_and_or_b32 %6, %6, %8, %9\nv_and_or_b32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "memory");
        } else if (KIND == 17) {
            asm volatile("v_xor_b32 %0, %0, %8\n v_bitop3_b32 %0, %0, %8, %9 bitop3:0xde\nv_xor_b32 %1, %1, %8\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0xde\nv_xor_b32 %2, %2, %8\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0xde\nv_xor_b32 %3, %3, %8\n v_bitop3_b32 %3, %3, %8, %9 bitop3:0xde\nv_xor_b32 %4, %4, %8\n v_bitop3_b32 %4, %4, %8, %9 bitop3:0xde\nv_xor_b32 %5, %5, %8\n v_bitop3_b32 %5, %5, %8, %9 bitop3:0xde\nv_xor_b32 %6, %6, %8\n v_bitop3_b32 %6, %6, %8, %9 bitop3:0xde\nv_xor_b32 %7, %7, %8\n v_bitop3_b32 %7, %7, %8, %9 bitop3:0xde\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "memory");
        } else if (KIND == 18) {
            asm volatile("v_and_b32 %0, %0, %8\n v_or3_b32 %0, %0, %8, %9\nv_and_b32 %1, %1, %8\n v_or3_b32 %1, %1, %8, %9\nv_and_b32 %2, %2, %8\n v_or3_b32 %2, %2, %8, %9\nv_and_b32 %3, %3, %8\n v_or3_b32 %3, %3, %8, %9\nv_and_b32 %4, %4, %8\n v_or3_b32 %4, %4, %8, %9\nv_and_b32 %5, %5, %8\n v_or3_b32 %5, %5, %8, %9\nv_and_b32 %6, %6, %8\n v_or3_b32 %6, %6, %8, %9\nv_and_b32 %7, %7, %8\n v_or3_b32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "memory");
        } else if (KIND == 19) {
            asm volatile("v_cmp_lt_i32 vcc, %0, %8\nv_cmp_lt_i32 vcc, %1, %8\nv_cmp_lt_i32 vcc, %2, %8\nv_cmp_lt_i32 vcc, %3, %8\nv_cmp_lt_i32 vcc, %4, %8\nv_cmp_lt_i32 vcc, %5, %8\nv_cmp_lt_i32 vcc, %6, %8\nv_cmp_lt_i32 vcc, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "memory", "vcc");
        } else if (KIND == 20) {
            asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "memory", "vcc");
        }
    }
    unsigned long long t1 = clock64();
    out[2 + blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *name, int waves_per_block, int blocks) {
    unsigned *out; unsigned long long *cyc;
    hipMalloc(&out, (2 + blocks * waves_per_block * 64) * 4);
    hipMemset(out, 0x55, 8);
    hipMalloc(&cyc, blocks * 8);
    const int iters = 200000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(waves_per_block * 64), 0, 0, out, cyc, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(waves_per_block * 64), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    double ns_per_instr_wave = ms * 1e6 / (8.0 * iters);   // per asm group of 8 lines (pairs count as one)
    printf("%-30s waves/CU %2d: %.2f ns per instruction per wave (%.2f s_memtime ticks) -> SIMD issue every %.2f ns\n", name,
           waves_per_block, ns_per_instr_wave, avg / (8.0 * iters), ns_per_instr_wave / (waves_per_block > 4 ? waves_per_block / 4.0 : 1.0));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {4, 16, 32}) {
        run<2>("v_and_b32 (VOP2) [x1]", w, 256);
        run<0>("v_bitop3_b32 [x1]", w, 256);
        run<1>("v_or3_b32 [x1]", w, 256);
        run<3>("v_alignbit_b32 [x1]", w, 256);
        run<8>("v_cmp_lt_i32 + v_addc_co (pair) [x2]", w, 256);
        run<9>("v_add_co_u32 + v_addc_co (pair) [x2]", w, 256);
        run<10>("v_add_u32_sdwa byte [x1]", w, 256);
        run<11>("v_min_i32 [x1]", w, 256);
        run<12>("v_lshlrev_b32 (VOP2) [x1]", w, 256);
        run<13>("v_lshl_add_u32 [x1]", w, 256);
        run<14>("v_sub_u32 [x1]", w, 256);
        run<15>("v_bfe_u32 [x1]", w, 256);
        run<16>("v_and_or_b32 [x1]", w, 256);
        run<17>("v_xor_b32 + v_bitop3 mix [x2]", w, 256);
        run<18>("v_and_b32 + v_or3 mix [x2]", w, 256);
        run<19>("v_cmp_lt_i32 only [x1]", w, 256);
        run<20>("v_cndmask_b32 (vcc) [x1]", w, 256);
    }
    return 0;
}
