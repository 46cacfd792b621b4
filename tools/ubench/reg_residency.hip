// Companion of lds_residency.hip: does a 256-thread workgroup with NV live VGPRs per lane and L bytes of LDS get the
// residency the register table promises (96 registers -> 5 waves per SIMD)?  Same spin method.
//   hipcc --offload-arch=gfx950 -O2 -o reg_residency reg_residency.hip && ./reg_residency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

extern __shared__ unsigned lds[];

template <int NV, int WG, int HS = 0>   // HS: also claim a high scalar register (sgpr_count ~ 100) and a little scratch
__global__ __launch_bounds__(256, WG) void spin(unsigned long long cycles, unsigned *sink, const unsigned *src) {
    if (HS >= 1) asm volatile("s_mov_b32 s95, 0" ::: "s95");
    volatile unsigned spill[HS >= 2 ? 15 : 1];
    if (HS >= 2) { for (int i = 0; i < 15; i++) spill[i] = src[i]; }
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned v[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = src[threadIdx.x + 256 * i];
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) {
#pragma unroll
        for (int i = 0; i < NV; i++) v[i] = v[i] * 1664525u + v[(i + 1) % NV];
    }
    unsigned acc = lds[(threadIdx.x + 1) & 255];
    if (HS >= 2) { for (int i = 0; i < 15; i++) acc += spill[i]; }
#pragma unroll
    for (int i = 0; i < NV; i++) acc ^= v[i];
    if (acc == 0x12345u) sink[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NV, int WG, int HS = 0>
void run(int cus, unsigned *sink, unsigned *src) {
    const void *fn = (const void *)spin<NV, WG, HS>;
    CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipFuncAttributes at;
    CK(hipFuncGetAttributes(&at, fn));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int sizes[] = {8192, 31520};
    for (int L : sizes) {
        int occ = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 256, (size_t)L));
        printf("live values %3d, launch bounds (256, %d), HS %d: numRegs %d, scratch %zu, L = %6d B: occupancy API %d;", NV, WG, HS, at.numRegs, (size_t)at.localSizeBytes, L, occ);
        for (int n = 3; n <= 8; n++) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(a));
                hipLaunchKernelGGL((spin<NV, WG, HS>), dim3(cus * n), dim3(256), (size_t)L, 0, 400000ull, sink, src);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms;
                CK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            printf(" n=%d %.3f", n, best);
        }
        printf(" ms\n");
    }
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned *sink, *src;
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&src, 256 * 128 * 4));
    CK(hipMemset(src, 1, 256 * 128 * 4));
    run<40, 4>(cus, sink, src);
    run<84, 4>(cus, sink, src);
    run<84, 5>(cus, sink, src);
    run<84, 5, 1>(cus, sink, src);
    run<84, 5, 2>(cus, sink, src);
    run<40, 4, 1>(cus, sink, src);
    run<40, 4, 2>(cus, sink, src);
    return 0;
}
