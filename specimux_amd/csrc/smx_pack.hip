// smx_pack.hip -- 4-bit window transport across the host boundary (gfx950).
//
// The hot path reads two `search_len` end windows per read and nothing else (demultiplex.py:757-766, :612-624).  The
// kernels consume 2-4 bits per base, the ASCII windows ship 8: over PCIe the link, not the kernels, bounds the
// host-buffer path.  The packer (smx_io.cpp: smx_pack_windows4*) therefore writes the kernels' 4-bit text codes
// (smx_internal.h kCodeChars: A C G T N R Y K M S W B D H V, 15 = anything else) -- 2 * ceil(S / 2) bytes per read instead
// of 2 * S -- and this kernel turns them back into the ASCII window layout in HBM, where the prescan and demux kernels
// pick them up unchanged.  Memory bound: reads S bytes, writes 2 S bytes per read.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

namespace smx {

// code -> ASCII letter; code 15 -> 0 (the byte smx_pack_windows pads with; every LUT maps it back to 15)
__device__ __forceinline__ unsigned unpack4(unsigned nib4) {   // four codes, one per byte (values 0..15) -> four letters
    // "ACGTNRYK" / "MSWBDHV\0": two v_perm lookups of eight entries each, merged by bit 3 of every code
    const unsigned lo = __builtin_amdgcn_perm(0x4B59524Eu, 0x54474341u, nib4 & 0x07070707u);   // bytes 4-7: N R Y K, 0-3: A C G T
    const unsigned hi = __builtin_amdgcn_perm(0x00564844u, 0x4257534Du, nib4 & 0x07070707u);   // bytes 4-7: D H V \0, 0-3: M S W B
    const unsigned m = ((nib4 >> 3) & 0x01010101u) * 0xFFu;
    return (lo & ~m) | (hi & m);
}

// Fast path (S a multiple of 8): one lane turns 4 packed bytes (8 bases) into 8 window bytes.
__global__ __launch_bounds__(256) void unpack_windows_kernel8(const uint32_t *__restrict__ packed, uint8_t *__restrict__ windows,
                                                              uint32_t n_reads, int S, int pstride, int wstride) {
    const int upr = S >> 2;   // 8-base units per read: S / 8 for the head + S / 8 for the tail
    const uint64_t total = (uint64_t)n_reads * (uint64_t)upr;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / (uint64_t)upr), u = (uint32_t)(i - (uint64_t)r * upr);
        const uint32_t w = packed[((size_t)r * pstride >> 2) + u];   // head units then tail units: S / 2 bytes each, contiguous
        // byte b of w holds bases 2b (low nibble) and 2b + 1 (high nibble)
        const unsigned even = w & 0x0F0F0F0Fu, odd = (w >> 4) & 0x0F0F0F0Fu;
        // interleave: letters of bases 0..3 = even.b0, odd.b0, even.b1, odd.b1
        const unsigned c03 = __builtin_amdgcn_perm(odd, even, 0x05010400u);
        const unsigned c47 = __builtin_amdgcn_perm(odd, even, 0x07030602u);
        uint2 o;
        o.x = unpack4(c03);
        o.y = unpack4(c47);
        *(uint2 *)(windows + (size_t)r * wstride + (size_t)u * 8) = o;   // head occupies [0, S), tail [S, 2 S): unit u starts at 8 u
    }
}

// General path (any S): one lane per window byte.
__global__ __launch_bounds__(256) void unpack_windows_kernel1(const uint8_t *__restrict__ packed, uint8_t *__restrict__ windows,
                                                              uint32_t n_reads, int S, int pstride, int wstride) {
    const int hb = (S + 1) >> 1;
    const uint64_t total = (uint64_t)n_reads * (uint64_t)wstride;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / (uint64_t)wstride);
        const int pos = (int)(i - (uint64_t)r * wstride);
        unsigned ch = 0;
        if (pos < 2 * S) {
            const int end = pos >= S, j = end ? pos - S : pos;
            const unsigned b = packed[(size_t)r * pstride + (size_t)(end ? hb : 0) + (size_t)(j >> 1)];
            const unsigned code = (j & 1) ? (b >> 4) : (b & 15u);
            ch = unpack4(code) & 0xFFu;
        }
        windows[i] = (uint8_t)ch;
    }
}

}  // namespace smx

extern "C" int smx_launch_unpack_windows(void *stream, const uint8_t *d_packed, uint8_t *d_windows, uint32_t n_reads, int S,
                                         int pstride, int wstride, int n_cu) {
    if (n_reads == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if ((S & 7) == 0 && (pstride & 3) == 0 && (wstride & 7) == 0) {
        const uint64_t total = (uint64_t)n_reads * (uint64_t)(S >> 2);
        const unsigned grid = (unsigned)std::min<uint64_t>((total + 255) / 256, (uint64_t)n_cu * 16);
        hipLaunchKernelGGL(smx::unpack_windows_kernel8, dim3(grid), dim3(256), 0, s, (const uint32_t *)d_packed, d_windows, n_reads, S,
                           pstride, wstride);
    } else {
        const uint64_t total = (uint64_t)n_reads * (uint64_t)wstride;
        const unsigned grid = (unsigned)std::min<uint64_t>((total + 255) / 256, (uint64_t)n_cu * 16);
        hipLaunchKernelGGL(smx::unpack_windows_kernel1, dim3(grid), dim3(256), 0, s, d_packed, d_windows, n_reads, S, pstride, wstride);
    }
    return (int)hipGetLastError();
}
