// smx_prescan.hip -- gfx950 kernels of the primer prescan (algorithm: smx_prescan_core.h).
//
// prescan_transpose_kernel: one 256-thread workgroup per tile of 1024 reads (32 groups of 32); LDS = the tile's packed
//   2-bit codes (2 * S / 16 blocks of 33 dwords per group: 42 KB at search_len 80); memory bound (reads the windows once).
// prescan_dp_kernel: one wave per (tile, primer) -- the pattern letters are wave-uniform kernel-argument loads; its 64
//   lanes are the tile's 32 groups x 2 ends.  No tile in LDS (a 4.6 KB scratch for the current / next column's
//   base-occurrence words only): residency is set by registers.  A lane keeps the DP column (2 x rows), the rows' scratch
//   addresses, the 5-plane gap counter, the 32 flag words of the current 16-column chunk and two four-column groups of
//   plane words in registers.  Output: one flag word per (primer, end, 16-column chunk, read), layout
//   [tile][primer * 2 + end][chunk][read in tile] (prescan_decode turns the chunk words of one alignment into
//   distance / ends), plus one match word per (tile, primer * 2 + end, 32-read group): bit r = that read reaches the
//   primer's threshold somewhere in the window ([tile][primer * 2 + end][group]; the demux kernel of a many-primer panel
//   keeps per-alignment state for the flagged alignments only).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smx_prescan_core.h"

namespace smx {

// ---- transpose kernel: windows -> 2-bit planes, bit-sliced over the 32 reads of a group (layout: prescan_plane_word)
// NT = 320 threads when the tile's 32 x (2 S / 16) blocks are a multiple of 320 (search_len 80: 320 blocks = exactly one
// transpose pass of five waves instead of one full pass of four plus one with a single wave busy), else 256.
template <int NT>
__global__ __launch_bounds__(NT) void prescan_transpose_kernel(int S, const uint8_t *__restrict__ windows,
                                                               const int32_t *__restrict__ lens, uint32_t n_reads, int stride,
                                                               unsigned *__restrict__ gplanes, uint32_t ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned plds[];
    const int tid = threadIdx.x;
    const int CH = S >> 4, ppr = 2 * CH;
    unsigned *planes = plds;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t r0 = tile * PRE_TILE;
        // ---- phase 1: the tile's windows are one contiguous run of 16-byte pieces (stride = ppr * 16): plain streaming
        // loads, eight in flight per lane
        {
            const int npieces = PRE_TILE * ppr;
            int read = tid / ppr, c = tid - read * ppr;
            const int dr = NT / ppr, dc = NT - dr * ppr;
            const uint4 *src = (const uint4 *)(windows + (size_t)r0 * stride);
            if (r0 + PRE_TILE <= n_reads) {   // whole tile in range (all but the last): no per-load guard, so that the
                                              // compiler keeps a batch of loads in flight instead of one per branch
                // npieces = 1024 * ppr is a multiple of 8 * NT (ppr is even): eight unconditional loads per batch.  (With a
                // bounds test per load the compiler waits for each load before it branches to the next: 40 serial round trips.)
                for (int q0 = tid; q0 < npieces; q0 += 8 * NT) {
                    uint4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) v[u] = src[q0 + u * NT];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        prescan_store_piece(planes, read, c, ppr, v[u].x, v[u].y, v[u].z, v[u].w);
                        read += dr; c += dc;
                        if (c >= ppr) { c -= ppr; read++; }
                    }
                }
            } else {
                for (int q = tid; q < npieces; q += NT) {
                    uint4 v = make_uint4(0u, 0u, 0u, 0u);
                    if (r0 + (uint32_t)read < n_reads) v = src[q];
                    prescan_store_piece(planes, read, c, ppr, v.x, v.y, v.z, v.w);
                    read += dr; c += dc;
                    if (c >= ppr) { c -= ppr; read++; }
                }
            }
        }
        __syncthreads();
        // ---- phase 1b: reads shorter than the window (rare): their head pieces again, right-aligned
        // (prescan_short_head_piece); kept out of the streaming loop, where the length load would sit in front of a branch
        {
            constexpr int NL = (PRE_TILE + NT - 1) / NT;
            int Ls[NL];
#pragma unroll
            for (int u = 0; u < NL; u++) {   // all length loads first
                const uint32_t rd = r0 + (uint32_t)(tid + u * NT);
                Ls[u] = lens[rd < n_reads ? rd : n_reads - 1];
            }
#pragma unroll
            for (int u = 0; u < NL; u++) {
                const int read = tid + u * NT;
                if (read < PRE_TILE && r0 + (uint32_t)read < n_reads && Ls[u] < S) {
                    const uint8_t *row = windows + (size_t)(r0 + (uint32_t)read) * stride;
                    for (int c = 0; c < CH; c++) {
                        unsigned w4[4];
                        prescan_short_head_piece(row, c, S, Ls[u], w4);
                        prescan_store_piece(planes, read, c, ppr, w4[0], w4[1], w4[2], w4[3]);
                    }
                }
            }
        }
        __syncthreads();
        // ---- phase 2: bit transposes, one block per lane and round; 8 x 16-byte stores per block
        uint4 *gp = (uint4 *)(gplanes + (size_t)tile * CH * 8 * 64 * 4);
        for (int b2 = tid; b2 < PRE_G * ppr; b2 += NT) {   // b2 = c * 32 + g: 32 consecutive lanes store 32 consecutive groups
            const int c = b2 >> 5, g = b2 & 31;
            unsigned o[32];
            prescan_transpose_block(planes, g * ppr + c, c, CH, o);
            const int chunk = prescan_block_chunk(c, CH), lane = prescan_block_lane(g, c, CH);
#pragma unroll
            for (int q = 0; q < 8; q++)
                gp[((size_t)chunk * 64 + lane) * 8 + q] = make_uint4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
        }
        __syncthreads();   // the next tile's phase 1 rewrites the staging blocks
    }
}

// ---- DP kernel: one wave per (tile, primer); NX = extra (degenerate-letter) symbol rows; MR = DP rows compiled in: 22 or
// 24 when every primer has at most that many nt, else 31 (one variant per kernel: two DP bodies in one kernel made the register
// allocator spill hundreds of registers)
#ifndef SMX_PRE_WAVES
#define SMX_PRE_WAVES 2   // waves per SIMD the DP kernel's register allocation aims at
#endif
template <int MR, int NX, int MT>
__global__ __launch_bounds__(64, SMX_PRE_WAVES) void prescan_dp_kernel(PreDesc D, const unsigned *__restrict__ gplanes,
                                                           unsigned *__restrict__ out, unsigned *__restrict__ match,
                                                           uint32_t ntiles) {
    __shared__ unsigned scratch[PRE_SCRATCH];
    const int lane = threadIdx.x;
    const int CH = D.S >> 4;
    const uint32_t nwork = ntiles * (uint32_t)D.NP;
    for (uint32_t wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
        const uint32_t tile = wi / (uint32_t)D.NP;
        const int p = (int)(wi - tile * (uint32_t)D.NP);
        const int g = lane >> 1, X = lane & 1;
        // tile-major output: the CH x 2 NP words of a read sit within its tile's 4 * 2 NP * CH KB
        prescan_dp<MR, NX, MT>(gplanes + (size_t)tile * CH * 8 * 64 * 4, scratch, lane, CH, D, p,
                           out + ((size_t)tile * (2 * D.NP) + (size_t)(2 * p + X)) * CH * PRE_TILE + (uint32_t)g * 32u, PRE_TILE,
                           match + ((size_t)tile * (2 * D.NP) + (size_t)(2 * p + X)) * PRE_G + (uint32_t)g);
    }
}

}  // namespace smx

// the streaming phase wants 1024 x pieces-per-read to be a multiple of 8 x threads, the transpose phase 32 x pieces-per-read
// blocks a multiple of the threads: 320 threads when pieces-per-read (2 S / 16) is a multiple of 10
extern "C" int smx_prescan_transpose_threads(int S) { return ((2 * (S >> 4)) % 10) == 0 ? 320 : 256; }

extern "C" size_t smx_prescan_lds_bytes(int S) {   // the transpose kernel's staging blocks
    return ((size_t)smx::PRE_G * (2 * (S >> 4)) * smx::PRE_BLK + 64) * 4;
}

#define SMX_PRE_VARIANTS(X) X(22, 0) X(22, 4) X(24, 0) X(24, 4) X(31, 0) X(31, 4)
static int prescan_rows(int mr) { return mr <= 22 ? 22 : (mr <= 24 ? 24 : 31); }   // DP rows compiled in: inert rows cost as much as live ones

static const void *prescan_fn(int mr, int nx) {
    const int mrv = prescan_rows(mr), nxv = nx > 0 ? 4 : 0;
#define X(MRV, NXV) if (mrv == MRV && nxv == NXV) return (const void *)smx::prescan_dp_kernel<MRV, NXV, 1>;
    SMX_PRE_VARIANTS(X)
#undef X
    return nullptr;
}

// mr = longest primer of the panel, nx = largest number of degenerate-letter symbols of one primer;
// grid_t / grid_d = resident workgroups of the two kernels (the caller sizes them); d_match = nullptr: no match words
extern "C" int smx_launch_prescan(const smx::PreDesc *D, int mr, int nx, int grid_t, size_t lds_t, int grid_d, void *stream,
                                  const uint8_t *d_windows, const int32_t *d_lens, uint32_t n_reads, int stride,
                                  unsigned *d_planes, unsigned *d_out, unsigned *d_match, void *ev_mid) {
    static_assert(smx::PRE_MAXROWS == 31 && smx::PRE_MAXSYM == 8, "variant table");
    const uint32_t ntiles = (n_reads + smx::PRE_TILE - 1) / smx::PRE_TILE;
    hipStream_t s = (hipStream_t)stream;
    if (smx_prescan_transpose_threads(D->S) == 320)
        hipLaunchKernelGGL(smx::prescan_transpose_kernel<320>, dim3(grid_t), dim3(320), lds_t, s, D->S, d_windows, d_lens, n_reads,
                           stride, d_planes, ntiles);
    else
        hipLaunchKernelGGL(smx::prescan_transpose_kernel<256>, dim3(grid_t), dim3(256), lds_t, s, D->S, d_windows, d_lens, n_reads,
                           stride, d_planes, ntiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (ev_mid) (void)hipEventRecord((hipEvent_t)ev_mid, s);   // diagnostic: boundary between the two kernels
    const int mrv = prescan_rows(mr), nxv = nx > 0 ? 4 : 0;
#define X(MRV, NXV)                                                                                            \
    if (mrv == MRV && nxv == NXV) {                                                                            \
        if (d_match)                                                                                           \
            hipLaunchKernelGGL((smx::prescan_dp_kernel<MRV, NXV, 1>), dim3(grid_d), dim3(64), 0, s, *D, d_planes, d_out, d_match, ntiles); \
        else                                                                                                   \
            hipLaunchKernelGGL((smx::prescan_dp_kernel<MRV, NXV, 0>), dim3(grid_d), dim3(64), 0, s, *D, d_planes, d_out, d_match, ntiles); \
    }
    SMX_PRE_VARIANTS(X)
#undef X
    return (int)hipGetLastError();
}

extern "C" int smx_prescan_set_lds_limit(size_t bytes) {
    hipError_t e = hipFuncSetAttribute((const void *)smx::prescan_transpose_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipError_t e2 = hipFuncSetAttribute((const void *)smx::prescan_transpose_kernel<320>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return (int)(e != hipSuccess ? e : e2);
}

extern "C" int smx_prescan_occupancy(int S, int mr, int nx, size_t lds_t, int *blocks_t, int *blocks_d) {
    const int nt = smx_prescan_transpose_threads(S);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(
        blocks_t, nt == 320 ? (const void *)smx::prescan_transpose_kernel<320> : (const void *)smx::prescan_transpose_kernel<256>, nt, lds_t);
    if (e != hipSuccess) return (int)e;
    return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_d, prescan_fn(mr, nx), 64, 0);
}
