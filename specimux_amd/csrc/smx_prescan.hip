// smx_prescan.hip -- gfx950 kernels of the primer prescan (algorithm: smx_prescan_core.h).
//
// prescan_transpose_kernel: one 256-thread workgroup per sub-tile of 256 reads (8 groups of 32); LDS = the sub-tile's packed
//   2-bit codes (2 * S / 16 x 8 blocks of 33 dwords: 11.6 KB at search_len 80); memory bound (reads the windows once).
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
// One 256-thread workgroup per SUB-TILE of 256 reads (8 groups; four sub-tiles fill one DP tile of 1024 reads): every lane
// issues all of its 16-byte window loads at once (2 S / 16 per lane: the whole sub-tile is in flight after one instruction
// burst), packs, stages in 10 KB of LDS, and 8 x (2 S / 16) lanes transpose one 32 x 32 bit block each.  Small workgroups on
// purpose: a batch is ~3 000 sub-tiles against ~2 000 resident workgroups, so loads of one workgroup overlap the transposes
// and stores of its neighbours -- with one workgroup per 1024-read tile the whole launch was a single load -> pack -> barrier ->
// transpose -> store sequence per CU slot (no steady state: 748 tiles on 768 slots), at 2.9 TB/s.
// Also written here, for the demux kernel (codes2 != nullptr): the same 2-bit codes row-major per read in DP order
// (codes2_word) -- it no longer reads the ASCII windows of reads whose windows are pure upper-case ACGT -- and one flag byte
// per read: 1 = a window holds something else, or the read is shorter than the window (naflag; such reads take the demux
// kernel's ASCII path; the ones with other characters also its scalar scan).
constexpr int PRE_TNT = 256;
__global__ __launch_bounds__(PRE_TNT) void prescan_transpose_kernel(int S, const uint8_t *__restrict__ windows,
                                                                    const int32_t *__restrict__ lens, uint32_t n_reads, int stride,
                                                                    unsigned *__restrict__ gplanes, unsigned *__restrict__ codes2,
                                                                    uint8_t *__restrict__ naflag, uint32_t nsub) {
    extern __shared__ __attribute__((aligned(16))) unsigned plds[];
    constexpr int NT = PRE_TNT, SUBR = PRE_SUBG * 32;
    const int tid = threadIdx.x;
    const int CH = S >> 4, ppr = 2 * CH;
    unsigned *planes = plds;
    unsigned *flagL = plds + ppr * PRE_CS;   // [SUBR] per read: non-zero = not pure ACGT
    for (uint32_t sub = blockIdx.x; sub < nsub; sub += gridDim.x) {
        const uint32_t r0 = sub * SUBR;
        flagL[tid] = 0u;
        // this lane's read length (phase 1b), requested together with the window loads: one memory round trip, not two
        const uint32_t rd_mine = r0 + (uint32_t)tid;
        const int L_mine = lens[rd_mine < n_reads ? rd_mine : n_reads - 1];
        __syncthreads();
        // ---- phase 1: the sub-tile's windows are one contiguous run of 16-byte pieces (stride = ppr * 16); piece q = (read
        // q / ppr, piece c = q % ppr); lane tid takes pieces tid + u * NT, u < ppr
        {
            int read = tid / ppr, c = tid - read * ppr;
            const int dr = NT / ppr, dc = NT - dr * ppr;
            const uint4 *src = (const uint4 *)(windows + (size_t)r0 * stride);
            unsigned *c2 = codes2 ? codes2 + (size_t)r0 * ppr : nullptr;
            auto piece = [&](const uint4 &v) {
                const unsigned bad = acgt_mismatch(v.x) | acgt_mismatch(v.y) | acgt_mismatch(v.z) | acgt_mismatch(v.w);
                if (bad) flagL[read] = 1u;
                const unsigned z = prescan_store_piece(planes, read, c, v.x, v.y, v.z, v.w);
                if (c2) {
                    int end, chunk;
                    const unsigned zz = codes2_from_piece(z, c, CH, &end, &chunk);
                    c2[codes2_word((size_t)read, CH, end, chunk)] = zz;
                }
                read += dr; c += dc;
                if (c >= ppr) { c -= ppr; read++; }
            };
            if (r0 + SUBR <= n_reads) {   // whole sub-tile in range (all but the last): unconditional loads, issued in bursts
                int u0 = 0;
                for (; u0 + 10 <= ppr; u0 += 10) {
                    uint4 v[10];
#pragma unroll
                    for (int u = 0; u < 10; u++) v[u] = src[tid + (u0 + u) * NT];
#pragma unroll
                    for (int u = 0; u < 10; u++) piece(v[u]);
                }
                for (; u0 + 2 <= ppr; u0 += 2) {   // (ppr is even)
                    uint4 v[2];
#pragma unroll
                    for (int u = 0; u < 2; u++) v[u] = src[tid + (u0 + u) * NT];
#pragma unroll
                    for (int u = 0; u < 2; u++) piece(v[u]);
                }
            } else {
                for (int u = 0; u < ppr; u++) {
                    uint4 v = make_uint4(0u, 0u, 0u, 0u);
                    const bool in = r0 + (uint32_t)read < n_reads;
                    if (in) v = src[tid + u * NT];
                    unsigned *keep = c2;
                    if (!in) c2 = nullptr;   // (no codes2 rows beyond the batch)
                    piece(v);
                    c2 = keep;
                }
            }
        }
        __syncthreads();
        // ---- phase 1b: reads shorter than the window (rare): their head pieces again, right-aligned
        // (prescan_short_head_piece); kept out of the streaming loop, where the length load would sit in front of a branch
        {
            const uint32_t rd = rd_mine;
            const int L = L_mine;
            if (rd < n_reads) {
                if (naflag) naflag[rd] = (uint8_t)(flagL[tid] != 0u || L < S);   // 1: the demux kernel encodes this read from ASCII
                if (L < S) {
                    const uint8_t *row = windows + (size_t)rd * stride;
                    for (int c = 0; c < CH; c++) {
                        unsigned w4[4];
                        prescan_short_head_piece(row, c, S, L, w4);
                        const unsigned z = prescan_store_piece(planes, tid, c, w4[0], w4[1], w4[2], w4[3]);
                        if (codes2) {
                            int end, chunk;
                            const unsigned zz = codes2_from_piece(z, c, CH, &end, &chunk);
                            codes2[codes2_word((size_t)rd, CH, end, chunk)] = zz;
                        }
                    }
                }
            }
        }
        __syncthreads();
        // ---- phase 2: bit transposes, one block per lane; 8 x 16-byte stores per block.  Lane -> (piece c fastest, group g):
        // the 32 lanes of a half-wave read 32 different LDS banks (smx_prescan_core.h PRE_CS)
        {
            const uint32_t tile = sub >> 2;
            const int g0 = (int)(sub & 3u) * PRE_SUBG;
            uint4 *gp = (uint4 *)(gplanes + (size_t)tile * CH * 8 * 64 * 4);
            for (int b2 = tid; b2 < PRE_SUBG * ppr; b2 += NT) {
                const int g = b2 / ppr, c = b2 - g * ppr;
                unsigned o[32];
                prescan_transpose_block(planes, g, c, CH, o);
                const int chunk = prescan_block_chunk(c, CH), lane = prescan_block_lane(g0 + g, c, CH);
#pragma unroll
                for (int q = 0; q < 8; q++)
                    gp[((size_t)chunk * 64 + lane) * 8 + q] = make_uint4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
            }
        }
        __syncthreads();   // the next sub-tile's phase 1 rewrites the staging blocks
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
    // work item -> (tile, primer): the NP waves of one tile read the same planes, so they get workgroup ids that are equal
    // mod 8 -- blocks b and b + 8 share an XCD (and its L2) -- and close together: item = (group of 8 tiles, primer, tile in
    // the group).  Speed only: with consecutive ids the planes of every tile were fetched from HBM once per primer.
    const uint32_t per8 = 8u * (uint32_t)D.NP;
    const uint32_t nwork = ((ntiles + 7u) >> 3) * per8;
    for (uint32_t wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
        const uint32_t grp = wi / per8, rem = wi - grp * per8;
        const uint32_t tile = grp * 8u + (rem & 7u);
        const int p = (int)(rem >> 3);
        if (tile >= ntiles) continue;
        const int g = lane >> 1, X = lane & 1;
        // tile-major output: the CH x 2 NP words of a read sit within its tile's 4 * 2 NP * CH KB
        prescan_dp<MR, NX, MT>(gplanes + (size_t)tile * CH * 8 * 64 * 4, scratch, lane, CH, D, p,
                           out + ((size_t)tile * (2 * D.NP) + (size_t)(2 * p + X)) * CH * PRE_TILE + (uint32_t)g * 32u, PRE_TILE,
                           match + ((size_t)tile * (2 * D.NP) + (size_t)(2 * p + X)) * PRE_G + (uint32_t)g);
    }
}

}  // namespace smx

extern "C" int smx_prescan_transpose_threads(int S) { (void)S; return smx::PRE_TNT; }

extern "C" size_t smx_prescan_lds_bytes(int S) {   // the transpose kernel's staging blocks + one flag word per read of the sub-tile
    return ((size_t)(2 * (S >> 4)) * smx::PRE_CS + smx::PRE_SUBG * 32) * 4;
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
                                  unsigned *d_planes, unsigned *d_out, unsigned *d_match, void *ev_mid, unsigned *d_codes2,
                                  uint8_t *d_naflag) {
    static_assert(smx::PRE_MAXROWS == 31 && smx::PRE_MAXSYM == 8, "variant table");
    const uint32_t ntiles = (n_reads + smx::PRE_TILE - 1) / smx::PRE_TILE;
    const uint32_t nsub = ntiles * (smx::PRE_G / smx::PRE_SUBG);   // every sub-tile of the last tile: the DP kernel reads whole tiles
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(smx::prescan_transpose_kernel, dim3(grid_t), dim3(smx::PRE_TNT), lds_t, s, D->S, d_windows, d_lens, n_reads,
                       stride, d_planes, d_codes2, d_naflag, nsub);
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
    return (int)hipFuncSetAttribute((const void *)smx::prescan_transpose_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

extern "C" int smx_prescan_occupancy(int S, int mr, int nx, size_t lds_t, int *blocks_t, int *blocks_d) {
    (void)S;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_t, (const void *)smx::prescan_transpose_kernel, smx::PRE_TNT, lds_t);
    if (e != hipSuccess) return (int)e;
    return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_d, prescan_fn(mr, nx), 64, 0);
}
