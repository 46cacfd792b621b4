// smx_prescan.hip -- gfx950 kernel of the primer prescan (algorithm and phases: smx_prescan_core.h).
//
// One workgroup of NW waves per tile of 1024 reads (32 groups of 32).  LDS: the tile's 2-bit text planes
// (2 * S / 16 blocks of 33 dwords per group: 42 KB at search_len 80) + a per-wave scratch of 2 x (nsym + 1) x 64 dwords
// for the current / next column's base-occurrence words.  Three workgroups of two waves share a CU at S = 80.
// Wave w aligns primers w, w + NW, ... (the pattern letters are wave-uniform: kernel-argument loads); its 64 lanes are
// the tile's 32 groups x 2 ends.  A lane keeps the DP column (2 x rows), the rows' scratch addresses, the 5-plane gap
// counter and the 32 flag words of the current 16-column chunk in registers:
// ~200 VGPRs, two waves per SIMD.  Output: one flag word per (primer, end, 16-column chunk, read), layout
// [primer * 2 + end][chunk][read] (prescan_decode turns the chunk words of one alignment into distance / ends).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smx_prescan_core.h"

namespace smx {

template <int NW, int MR, int NX>   // NX = extra (degenerate-letter) symbol rows; MR = DP rows compiled in: 24 when every primer has <= 24 nt, else 31 (one variant per kernel:
                           // two DP bodies in one kernel made the register allocator spill hundreds of registers)
__global__ __launch_bounds__(NW * 64, 2) void prescan_kernel(PreDesc D, const uint8_t *__restrict__ windows,
                                                             const int32_t *__restrict__ lens, uint32_t n_reads, int stride, unsigned *__restrict__ out,
                                                             uint32_t npad, uint32_t ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned plds[];
    constexpr int NT = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: the primer loop and the row switch are scalar branches
    const int CH = D.S >> 4, ppr = 2 * CH;
    unsigned *planes = plds;
    unsigned *scratch = plds + PRE_G * ppr * PRE_BLK + 64 + wave * PRE_SCRATCH;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t r0 = tile * PRE_TILE;
        // ---- phase 1: the tile's windows are one contiguous run of 16-byte pieces (stride = ppr * 16)
        {
            const int npieces = PRE_TILE * ppr;
            int read = tid / ppr, c = tid - read * ppr;
            const int dr = NT / ppr, dc = NT - dr * ppr;
            const uint4 *src = (const uint4 *)(windows + (size_t)r0 * stride);
#pragma unroll 8
            for (int q = tid; q < npieces; q += NT) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (r0 + (uint32_t)read < n_reads) {
                    v = src[q];
                    if (c < CH) {   // head window of a short read (rare): right-aligned, see prescan_short_head_piece
                        const int L = lens[r0 + (uint32_t)read];
                        if (L < D.S) {
                            unsigned w4[4];
                            prescan_short_head_piece(windows + (size_t)(r0 + (uint32_t)read) * stride, c, D.S, L, w4);
                            v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                        }
                    }
                }
                prescan_store_piece(planes, read, c, ppr, v.x, v.y, v.z, v.w);
                read += dr; c += dc;
                if (c >= ppr) { c -= ppr; read++; }
            }
        }
        __syncthreads();
        // ---- phase 2: in-place bit transposes, one block per lane and round
        for (int b = tid; b < PRE_G * ppr; b += NT) {
            const int g = b / ppr;
            prescan_transpose_block(planes, b, b - g * ppr, CH);
        }
        __syncthreads();
        // ---- phase 3 + 4: this wave's primers
        const int g = lane >> 1, X = lane & 1;
        for (int p = wave; p < D.NP; p += NW) {
            prescan_dp<MR, NX>(planes, scratch, lane, g, X, CH, ppr, D, p,
                               out + (size_t)(2 * p + X) * CH * npad + r0 + (uint32_t)g * 32u, npad);
        }
        __syncthreads();   // the next tile's phase 1 rewrites the planes
    }
}

}  // namespace smx

extern "C" size_t smx_prescan_lds_bytes(int S, int nsym, int nw) {
    return ((size_t)smx::PRE_G * (2 * (S >> 4)) * smx::PRE_BLK + 64 + (size_t)nw * smx::PRE_SCRATCH) * 4;
}

#define SMX_PRE_VARIANTS(X)                                                                        \
    X(2, 24, 0) X(2, 24, 4) X(2, 31, 0) X(2, 31, 4) X(4, 24, 0) X(4, 24, 4) X(4, 31, 0) X(4, 31, 4)

static const void *prescan_fn(int nw, int mr, int nx) {
    const int mrv = mr <= 24 ? 24 : 31, nxv = nx > 0 ? 4 : 0, nwv = nw == 4 ? 4 : 2;
#define X(NWV, MRV, NXV) if (nwv == NWV && mrv == MRV && nxv == NXV) return (const void *)smx::prescan_kernel<NWV, MRV, NXV>;
    SMX_PRE_VARIANTS(X)
#undef X
    return nullptr;
}

// nw = 2 or 4 waves per workgroup, mr = longest primer of the panel, nx = number of degenerate-letter symbols;
// grid = resident workgroups (the caller sizes it)
extern "C" int smx_launch_prescan(const smx::PreDesc *D, int nw, int mr, int nx, int grid, size_t lds_bytes, void *stream,
                                  const uint8_t *d_windows, const int32_t *d_lens, uint32_t n_reads, int stride,
                                  unsigned *d_out, uint32_t npad) {
    static_assert(smx::PRE_MAXROWS == 31 && smx::PRE_MAXSYM == 8, "variant table");
    const uint32_t ntiles = (n_reads + smx::PRE_TILE - 1) / smx::PRE_TILE;
    hipStream_t s = (hipStream_t)stream;
    const int mrv = mr <= 24 ? 24 : 31, nxv = nx > 0 ? 4 : 0, nwv = nw == 4 ? 4 : 2;
#define X(NWV, MRV, NXV)                                                                                                  \
    if (nwv == NWV && mrv == MRV && nxv == NXV)                                                                           \
        hipLaunchKernelGGL((smx::prescan_kernel<NWV, MRV, NXV>), dim3(grid), dim3(NWV * 64), lds_bytes, s, *D, d_windows, \
                           d_lens, n_reads, stride, d_out, npad, ntiles);
    SMX_PRE_VARIANTS(X)
#undef X
    return (int)hipGetLastError();
}

extern "C" int smx_prescan_set_lds_limit(int nw, int mr, int nx, size_t bytes) {
    return (int)hipFuncSetAttribute(prescan_fn(nw, mr, nx), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

extern "C" int smx_prescan_occupancy(int nw, int mr, int nx, size_t lds_bytes, int *blocks_per_cu) {
    return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, prescan_fn(nw, mr, nx), nw * 64, lds_bytes);
}
