// smx_internal.h -- device-side view of the panel, shared by smx_kernels.hip and smx_api.cpp.
#ifndef SMX_INTERNAL_H
#define SMX_INTERNAL_H
#include <stddef.h>
#include <stdint.h>
#include "smx.h"
#include "smx_prescan_core.h"

// LDS a CU of gfx950 gives to workgroups (in 512-byte granules): measured, tools/ubench/lds_residency.hip
#define SMX_LDS_POOL ((size_t)159744)

namespace smx {

// Text alphabet of the kernels: 16 codes.  Code 15 ("other") never matches any pattern character.
// Order matters only for the LUTs built in smx_api.cpp.
static const char kCodeChars[16] = {'A', 'C', 'G', 'T', 'N', 'R', 'Y', 'K', 'M', 'S', 'W', 'B', 'D', 'H', 'V', 0};

// One specimen as the scorer needs it, 32 bytes: two of these tables -- per (b1, b2) barcode pair the FIRST specimen in
// file order (spec = -1: none), per specimen its own record -- so that specimen_exact is one load in the usual case.
struct SpecRec {
    unsigned long long p1m, p2m;   // forward / reverse primers the specimen is registered with (bit = primer index)
    int spec, next, pool, pad;     // specimen index, next specimen with the same barcode pair (-1: none), its pool
};

// All pointers are DEVICE pointers (one allocation, see smx_api.cpp).  Passed to kernels by value.
struct DevPanel {
    int NP, NB, NS, NPAIR;
    int n_pbc;        // total length of the per-primer barcode lists
    int S;            // search_len
    int wstride;      // bytes per read in the window buffer
    int kidx;         // max_dist_index
    int bmax;         // Specimens.b_length()
    int pfmin;        // prefilter min length, 0 = off
    int preorient, trim, derep, minlen, maxlen;
    int maxB;         // max barcodes per primer
    int need_starts;  // HW start locations required (trim primers / tails)
    const unsigned long long *ppeq;   // NP*16 : bit i of ppeq[p*16+c] = eq(primer_rc[i], code c)
    const unsigned long long *prpeq;  // NP*16 : same for the reversed pattern
    const unsigned *bpeq;             // NB*16
    const unsigned char *lut;         // [0,256): ASCII -> code; [256,512): ASCII -> code of the complement
    const int *pm, *pk, *pdir, *pfidx;    // per primer: pattern length, k_p, direction, file index
    const int *pbc_off, *pbc;             // per primer barcode list (global barcode indices)
    const int *bm;                        // per barcode length
    const int *pair_f, *pair_r, *pair_pool;
    const SpecRec *pairrec;               // NB*NB: the first specimen (file order) with (b1, b2), as a whole record; spec = -1: none
    const SpecRec *specrec;               // per specimen: its record (next = the following specimen with the same barcode pair)
    // bit-sliced barcode scan (lean mode): all barcodes one length bs_m <= 16, k <= 7.
    // bs_re[((p * 16 + row) * 16 + code) * MBW + w] = bitmask over primer p's barcode list: barcode_rc[row] eq code
    int bs_ok, bs_m;
    int n_bstab;              // distinct per-primer tables in bs_re (primers with the same barcode list share one)
    const int *bs_tab;        // per primer: its table
    int cap_hits, cap_ents;   // test hook (SMX_TEST_CAPS=h,e): force small barcode rounds; 0 = default sizing
    int no_sp;                // SMX_NO_SPECIALISE (bit 0) / SMX_NO_SPECIALISE_NP (bit 1), read once at smx_panel_create
    const unsigned *bs_re;
    unsigned long long *dbg_phase;        // SMX_PHASE_TIMING=1: [grid][16] cycle sums per phase (diagnostic build-in)
};

// Compact mode and the redo launch behind it (smx_kernels.hip, demux_kernel).
struct DemuxAux {
    const unsigned *match;   // prescan match words [tile of 1024 reads][alignment][32-read group]; compact mode only
    unsigned *ovf_list;      // compact launch: the tiles it could not hold (appended); redo launch: the tiles to process
    int nitems;              // > 0: compact mode, this many per-alignment records per tile (<= 256)
    int redo;                // 1: this launch processes the reads of ovf_list's tiles (Rc reads each) and nothing else
    int Rc;                  // reads per tile of the compact launch
    int chain;               // 1: another launch of this batch follows: the extra-record and overflow counters stay
    const unsigned *codes2;  // prescan: row-major 2-bit codes per read in DP order (smx_prescan_core.h codes2_word); nullptr: encode from ASCII
    const uint8_t *naflag;   // prescan: per read, 1 = a window holds something other than upper-case ACGT (ASCII path for that read)
};

}  // namespace smx

extern "C" {
int smx_launch_demux(const smx::DevPanel *P, int use64, int R, int grid, size_t lds_bytes, void *stream,
                     const uint8_t *d_windows, const int32_t *d_lens, uint32_t n_reads, smx_op *d_ops, smx_op *d_extra,
                     uint32_t extra_cap, uint32_t *d_n_extra, uint64_t *d_counts, smx_hit *d_hits, int8_t *d_bdist,
                     unsigned *d_tile_counter, int use_slots, const unsigned *d_pre, uint32_t npad, const smx::DemuxAux *aux);
// primer prescan (smx_prescan.hip)
size_t smx_prescan_lds_bytes(int S);
int smx_launch_prescan(const smx::PreDesc *D, int mr, int nx, int grid_t, size_t lds_t, int grid_d, void *stream,
                       const uint8_t *d_windows, const int32_t *d_lens, uint32_t n_reads, int stride, unsigned *d_planes,
                       unsigned *d_out, unsigned *d_match, void *ev_mid, unsigned *d_codes2, uint8_t *d_naflag);
int smx_prescan_set_lds_limit(size_t bytes);
int smx_prescan_occupancy(int S, int mr, int nx, size_t lds_t, int *blocks_t, int *blocks_d);
int smx_prescan_transpose_threads(int S);
size_t smx_demux_lds_bytes(int use64, int NP, int NB, int S, int R, int maxB, int need_starts, int npmeta, int kidx,
                           int slots, int bs, int nitems, int ncand, int tails, int nbstab);
int smx_set_demux_lds_limit(int use64, size_t bytes);
int smx_demux_sp_query(const smx::DevPanel *P, int use64, int use_slots, int cm, int R, int nitems, int have_prescan);
int smx_query_occupancy(const smx::DevPanel *P, int use64, int use_slots, int cm, int R, int nitems, size_t lds_bytes,
                        int *blocks_per_cu, int have_prescan);
int smx_launch_align(void *stream, const unsigned long long *d_peq, const unsigned long long *d_rpeq, int m,
                     const unsigned char *d_tcodes, int n, int k, int mode, int *d_dist, unsigned char *d_endflag,
                     int *d_starts);
int smx_launch_align_batch(void *stream, const unsigned long long *d_qpeq, const int *d_qlen, const unsigned *d_qidx,
                           const unsigned char *d_tcodes, const unsigned long long *d_toff, const int *d_k,
                           const unsigned char *d_modes, unsigned n, unsigned char *d_ws, int *d_dist, int *d_nloc,
                           int *d_starts, int *d_ends, unsigned cap);
}
#endif
