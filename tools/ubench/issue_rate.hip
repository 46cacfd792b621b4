// Micro-benchmark (gfx950): how fast ONE wave issues VALU instructions, and how that scales with waves per SIMD.
// Body = 96 v_bitop3 / v_and instructions, fully unrolled (no loop overhead inside the body), arranged as DEP chains:
// DEP = 1: every instruction depends on the previous one; DEP = 2, 4, 8: that many independent chains interleaved.
// Build: hipcc -O3 --offload-arch=gfx950 issue_rate.hip -o issue_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int DEP>
__global__ void k(unsigned *out, unsigned long long *cyc, int iters) {
    unsigned a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 7 + i;
    unsigned b = out[threadIdx.x & 63], c = out[64 + (threadIdx.x & 63)];
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 96; u++) {
            const int j = u % DEP;
            if (u & 1) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf1" : "+v"(a[j]) : "v"(b), "v"(c));
            else asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
        }
    }
    unsigned long long t1 = clock64();
    unsigned x = 0;
    for (int i = 0; i < 8; i++) x ^= a[i];
    out[128 + blockIdx.x * blockDim.x + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int DEP>
void run(int waves_per_cu) {
    const int blocks = 256, nt = waves_per_cu * 64, iters = 2000;
    unsigned *out; unsigned long long *cyc;
    hipMalloc(&out, (128 + (size_t)blocks * nt) * 4);
    hipMemset(out, 0x55, 512);
    hipMalloc(&cyc, (size_t)blocks * waves_per_cu * 8);
    hipLaunchKernelGGL(k<DEP>, dim3(blocks), dim3(nt), 0, 0, out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<DEP>, dim3(blocks), dim3(nt), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * waves_per_cu);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    const double n = 96.0 * iters;
    printf("chains %d  waves/SIMD %d: %.2f shader cycles per instruction per wave (s_memtime), %.2f cycles of SIMD time per instruction (wall, 2.4 GHz)\n",
           DEP, waves_per_cu / 4, avg / n, ms * 1e-3 * 2.4e9 / n / (waves_per_cu / 4));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {4, 8, 12, 16, 24, 32}) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); }
    return 0;
}
