// Micro-benchmark (gfx950): streaming read rate of a 122 MB buffer with 16-byte loads, the way the prescan transpose
// kernel and the demux kernel's encode phase read the window buffer.  Variants: loads in flight per lane, grid shape.
// Build: hipcc -O3 --offload-arch=gfx950 stream_rate.hip -o stream_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int INFLIGHT>
__global__ __launch_bounds__(256) void k(const uint4 *__restrict__ src, size_t n16, unsigned *out, int chunk16) {
    // workgroup b streams [b * chunk16, (b + 1) * chunk16): 256 lanes x 16 B contiguous per load instruction
    unsigned acc = 0;
    for (size_t base = (size_t)blockIdx.x * chunk16; base < n16; base += (size_t)gridDim.x * chunk16) {
        for (int q0 = threadIdx.x; q0 < chunk16; q0 += INFLIGHT * 256) {
            uint4 v[INFLIGHT];
#pragma unroll
            for (int u = 0; u < INFLIGHT; u++) v[u] = (q0 + u * 256 < chunk16 && base + q0 + u * 256 < n16) ? src[base + q0 + u * 256] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < INFLIGHT; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int INFLIGHT>
void run(const uint4 *src, size_t n16, unsigned *out, int grid, int chunk16) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<INFLIGHT>, dim3(grid), dim3(256), 0, 0, src, n16, out, chunk16);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k<INFLIGHT>, dim3(grid), dim3(256), 0, 0, src, n16, out, chunk16);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("in flight %2d, grid %5d, chunk %6d KB: %.1f us per pass, %.2f TB/s\n", INFLIGHT, grid, chunk16 / 64, ms * 100.0,
           (double)n16 * 16 / (ms * 1e-4) / 1e12);
}

int main() {
    const size_t bytes = 765000ull * 160, n16 = bytes / 16;
    uint4 *src; unsigned *out;
    hipMalloc(&src, bytes); hipMemset(src, 1, bytes); hipMalloc(&out, 64);
    for (int grid : {748, 1024, 2048, 4096}) {
        const int chunk16 = (int)((n16 + grid - 1) / grid);
        run<4>(src, n16, out, grid, chunk16);
        run<8>(src, n16, out, grid, chunk16);
        run<16>(src, n16, out, grid, chunk16);
    }
    run<8>(src, n16, out, 1024, 10240);   // 160 KB tiles (the transpose kernel's), persistent over 1024 workgroups
    run<8>(src, n16, out, 2048, 2560);
    return 0;
}
