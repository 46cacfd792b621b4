// How many 256-thread workgroups with L bytes of dynamic LDS does a CU of gfx950 really hold?  (The occupancy API divides
// 160 KB by L; the allocator may round L up.)  Every workgroup spins for a fixed number of shader cycles; a grid of
// n * 256 workgroups takes one spin if n fit per CU and two if they do not.
//   hipcc --offload-arch=gfx950 -O2 -o lds_residency lds_residency.hip && ./lds_residency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

extern __shared__ unsigned lds[];

__global__ __launch_bounds__(256) void spin(unsigned long long cycles, unsigned *sink) {
    lds[threadIdx.x] = threadIdx.x;   // the allocation is used
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned acc = lds[(threadIdx.x + 1) & 255];
    while (__builtin_readcyclecounter() - t0 < cycles) acc = acc * 1664525u + 1013904223u;
    if (acc == 0x12345u) sink[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned *sink;
    CK(hipMalloc(&sink, 64));
    CK(hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const unsigned long long cycles = 400000;   // of the 100 MHz s_memtime counter or of the shader clock, whichever the builtin reads: the ratio is what counts
    printf("CUs %d, LDS per CU %zu\n", cus, (size_t)prop.maxSharedMemoryPerMultiProcessor);
    const int sizes[] = {16384, 20480, 24576, 26624, 27648, 28672, 30720, 31744, 32256, 32512, 32768, 33024, 36864, 40960, 49152, 53248, 54272, 54613, 65536, 81920};
    for (int L : sizes) {
        int occ = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin, 256, (size_t)L));
        printf("L = %6d B: occupancy API %d;", L, occ);
        for (int n = 1; n <= 8; n++) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(spin, dim3(cus * n), dim3(256), (size_t)L, 0, cycles, sink);
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
    return 0;
}
