// smx_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the specimux hot path.
//
// One fused kernel per read batch ("demux kernel"), no intermediate HBM traffic: workgroups of 256 threads pull
// tiles of R reads from a global tile queue; everything between the windows and the 32-byte result records lives
// in LDS (layout: make_layout).
// The tile loop is software-pipelined: phase 4 of tile t (one wave) runs beside phase 1 of tile t+1 (three waves).
//     phase 0  panel tables staged once per workgroup: primer Peq [code][primer] (patterns left-aligned), bit-sliced
//              barcode tables [primer][32-barcode word][row][code], ASCII->code LUTs, pair / barcode lists
//     phase 1  coalesced 16-byte loads of the two `search_len` end windows, ASCII -> 4-bit IUPAC code,
//              window A reverse-complemented on the fly           (demultiplex.py:142, :757-766)
//     phase 2  primer scan: one lane per (read, primer, end): Myers/Hyyro bit-vector HW (infix) DP,
//              all optimal end columns kept as an LDS bitmask     (match_one_end :755-770, align_seq)
//              + orientation votes from the same alignments       (determine_orientation :602-638, A.6)
//     phase 3  barcode scan per optimal primer location ("entry"), exact-set prefilter rule per entry
//              lean mode : bit-sliced banded SHW DP, one lane aligns the primer's whole barcode list (32 per word)
//                          in a 7-row register window, barcodes padded to a fixed height by wildcard rows so the
//                          scan is straight-line code; results are OR-ed into per-hit "barcodes seen at distance
//                          d" bitmasks
//              slots mode: one lane per (entry, barcode), bit-vector SHW, LDS atomicMin keeps the best location
//                          per barcode (needed for --trim tails and the parity dumps)
//                                                                  (match_one_end :778-815, bloom_filter.py:176)
//     phase 4  scorer, one lane per read: candidate enumeration, select_best_matches, dereplicate_*,
//              resolve_specimen, trim extents -> smx_op records + counters
//                                                                  (demultiplex.py:108-598, models.py:278-328)
//
// Pure integer work: no MFMA.  The kernel is bound by VALU issue; instruction selection follows the measured opcode
// rates of tools/ubench (bit-ops / add / sub / v_bitop3 are fast; shifts, min, compares, SDWA and three-source VOP3
// forms are about half rate; generic integer division is avoided in per-item code).  See DESIGN.md for the data
// layout, the exactness arguments and the rooflines.
#include <hip/hip_runtime.h>
#ifndef SMX_PART
#error "compile with -DSMX_PART=1, 2 or 3 (see the Makefile)"
#endif
#include <stdint.h>
#include "smx.h"
#include "smx_internal.h"

namespace smx {

// ------------------------------------------------------------------------------------------------
// Myers / Hyyro bit-vector column step.  Bits above the pattern length carry garbage
// that never flows downwards (adds carry upwards, shifts go left), so no masking is needed.
template <typename W, bool PREFIX>
__device__ __forceinline__ void myers_step(W Eq, W &Pv, W &Mv, int &score, int top) {
    W Xv = Eq | Mv;
    W Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
    W Ph = Mv | ~(Xh | Pv);
    W Mh = Pv & Xh;
    score += (int)((Ph >> top) & 1) - (int)((Mh >> top) & 1);
    Ph = (Ph << 1) | (W)(PREFIX ? 1 : 0);
    Mh <<= 1;
    Pv = Mh | ~(Xv | Ph);
    Mv = Ph & Xv;
}

// HW column step for a 32-bit pattern LEFT-ALIGNED in its word (row m-1 = bit 31; the padding rows below the
// pattern have Eq = 0 and stay inert: Pv = 1, Mv = Ph = Mh = 0), so the score delta is two plain shifts of
// the top bit (one of them arithmetic: -1) and a three-operand add.
__device__ __forceinline__ void myers_step_hw_top(unsigned Eq, unsigned &Pv, unsigned &Mv, int &score) {
    unsigned Xv = Eq | Mv;
    unsigned Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
    unsigned Ph = Mv | ~(Xh | Pv);
    unsigned Mh = Pv & Xh;
    score = score + (int)(Ph >> 31) + ((int)Mh >> 31);
    // x + x instead of x << 1: measured on gfx950 (tools/ubench/valu_rate.hip, primer_col.hip) v_add_u32 issues
    // faster than v_lshlrev_b32; the asm keeps LLVM from canonicalising the add back into a shift
    asm("v_add_u32 %0, %1, %1" : "=v"(Ph) : "v"(Ph));
    asm("v_add_u32 %0, %1, %1" : "=v"(Mh) : "v"(Mh));
    Pv = Mh | ~(Xv | Ph);
    Mv = Ph & Xv;
}

// Geometry of one end string q of length L.  The stored window holds
// q[L-Sp : L], Sp = min(S, L); window coordinate j <-> q index j + base.
struct EndGeom {
    int Sp;     // stored window length
    int base;   // q index of window position 0
    int j_lo;   // window position where match_one_end's primer target starts
    int shift;  // reference coordinate = (j - j_lo) + shift
};
__device__ __forceinline__ EndGeom end_geom(int L, int S) {
    EndGeom g;
    g.Sp = L < S ? L : S;
    g.base = L - g.Sp;
    if (L >= S) { g.j_lo = 0; g.shift = L - S; }
    else if (L == S - 1) { g.j_lo = 0; g.shift = 0; }           // start == -1 means 0 (alignment.py:37)
    else { int lo = 2 * L - S; g.j_lo = lo > 0 ? lo : 0; g.shift = L - S; }  // negative slice start wraps
    return g;
}

// Barcode target of align_seq(b_rc, q, k, bstart, L, SHW) (demultiplex.py:787-800) in window coordinates.
struct BcGeom {
    int tj0;     // window position of the first target base (may be >= Sp: empty target)
    int delta;   // reference coordinate of window position j = j + base - delta ... see bc_abs()
    bool pf_same; // prefilter slice sequence[bstart:] is the same string as the alignment target
};
__device__ __forceinline__ BcGeom bc_geom(int L, int base, int bstart) {
    BcGeom b;
    int s_ = (bstart == -1) ? 0 : bstart;
    int lo = s_ < 0 ? (L + s_ > 0 ? L + s_ : 0) : (s_ < L ? s_ : L);
    int plo = bstart < 0 ? (L + bstart > 0 ? L + bstart : 0) : (bstart < L ? bstart : L);
    b.tj0 = lo - base;
    b.delta = lo - s_;      // reference coordinate = q index - delta
    b.pf_same = (plo == lo);
    return b;
}

// ------------------------------------------------------------------------------------------------
// LDS-resident per (read, primer, end) record.
// Tile geometry of the two-primer default-flags kernel (SP = 2): reads per tile, and the resident workgroups per CU it is
// built for.  Five: its tile is 31.5 KB of LDS (five of them fit a CU: tools/ubench/lds_residency.hip) and the register
// allocator is held to 96 VGPRs (8 values spilled).  Measured: 56-read tiles at four per CU 0.255 ms, at five 0.228 ms.
#ifndef SMX_SP2_R
#define SMX_SP2_R 64
#define SMX_SP2_WG 5
#endif

struct HitL {
    int tail_end;       // reference coord: max optimal end over all within-k barcodes (slots mode), valid if bbest >= 0
    short nloc;
    short ntied;
    short first_tied;   // local index in the primer's barcode list
    signed char pdist;  // -1 no primer match (primers <= 64 nt)
    signed char bbest;  // -1 none, -2 not searched (barcodes <= 32 nt)
    unsigned char jstar;   // window position of the first optimal primer end
    unsigned char fs_j;    // window position of its start (need_starts only)
    unsigned char flags;   // bit0: orientation vote, bit1: barcode search needed
    unsigned char pad;
};
static_assert(sizeof(HitL) == 16, "HitL layout");

struct TileLayout {   // byte offsets into dynamic LDS
    int tacc, ppeq, prpeq, bpeq, bsre, lut, pmeta, codes, namask, lens, ocnt, rflag, hits, masks, tiem, bres, dmask, ents, offsA, offsB, queue, emit, opsL, aggr, etail,
        hmap, clist, pmask, hcand, total;
    int NI;      // per-alignment records a tile keeps: R * H (dense) or the compact capacity
    int CS;      // bytes per code row (odd number of dwords: conflict-free column reads across rows)
    int lNPs, lNBs;  // their log2
    int NPs, NBs;  // power-of-two strides of the transposed Peq tables: entry [code][pattern]
    int G, logG;   // barcode slots per (hit) group: power of two >= maxB
    int MBW;     // tie-mask words per hit
    int BSP;     // bit-sliced table: words per primer
    int CAPH, CAPE;  // hits / (hit, location) entries processed per barcode round
};

struct EntL {   // one optimal primer location of one searched hit = one barcode target
    unsigned short hit;     // r * H + h
    unsigned short slot;    // first bres slot of the hit in this round
    unsigned char tj0;      // window position of the first target base
    unsigned char loc_ord;  // ordinal of the location (ascending end)
    unsigned char ncol;     // target columns available (capped)
    unsigned char ok;       // 0: target empty or rejected by the prefilter rule
    short delta;            // reference coordinate of window position j at this location = j + base - delta (bc_geom)
    short pad;
};
static_assert(sizeof(EntL) == 12, "EntL layout");

// the scorer looks at the candidates of matched alignments only (panels with more than two primer pairs)
__host__ __device__ inline bool layout_cfilt(int H, int ncand) { return H <= 64 && ncand <= 64 && ncand > 4; }
__host__ __device__ inline bool layout_tie_in_rec(int MBW, int slots, int tails) { return MBW == 1 && !slots && !tails; }
#define OCNT_NA 0x8000

template <typename PW>
__host__ __device__ inline TileLayout make_layout(int NP, int NB, int S, int R, int maxB, int need_starts,
                                                    int npmeta, int kidx, int slots, int bs, int ncand, int cap_hits = 0,
                                                    int cap_ents = 0, int nitems = 0, int tails = 1, int nbstab = -1) {
    // nitems > 0: compact mode.  The tile keeps per-alignment state (hit record, end mask, scan slots) for at most nitems
    // of its R * H alignments -- the ones the prescan's match words flag -- plus one shared "no match" record; hmap maps
    // (read, alignment) to its record.  Panels with many primers spend most of their LDS on alignments that never match.
    TileLayout t;
    int H = 2 * NP, MW = (S + 31) / 32;
    t.CS = 4 * (((S + 3) / 4) | 1);
    t.NPs = 1; t.lNPs = 0; while (t.NPs < NP) { t.NPs <<= 1; t.lNPs++; }   // table index = (code << log2) | pattern:
    t.NBs = 1; t.lNBs = 0; while (t.NBs < NB) { t.NBs <<= 1; t.lNBs++; }   // no integer multiply in the inner loops
    t.G = 1; t.logG = 0;
    while (t.G < maxB) { t.G <<= 1; t.logG++; }
    t.MBW = (maxB + 31) / 32;
    // barcode rounds.  slots mode (--trim tails, parity dumps): one 4-byte atomicMin slot per (hit, barcode),
    // at most 16 KiB worth of hits per round.  lean mode: per hit only (k+1) x MBW words of "barcodes seen at
    // distance d" bitmasks, at most 4 KiB per round.
    const int sentinel = nitems > 0 ? 1 : 0;
    t.NI = nitems > 0 ? nitems : R * H;
    int dense = t.NI;
    // (compact mode: up to 8 KiB, so that the searched hits of a 64-read tile usually go in ONE round -- the several-round
    // path has serial bookkeeping)
    int fit = slots ? (16 * 1024) / (t.G * 4) : ((nitems > 0 ? 8 : 3) * 1024) / ((kidx + 1) * t.MBW * 4);
    t.CAPH = dense < fit ? dense : (fit < 1 ? 1 : fit);
    t.CAPE = t.CAPH + t.CAPH / 4 < 256 ? 256 : t.CAPH + t.CAPH / 4;   // >= 256 = max locations of one hit (progress)
    if (nitems > 0) t.CAPE = t.CAPH + t.CAPH / 2 < 256 ? 256 : t.CAPH + t.CAPH / 2;   // (noisy reads of a wide window: several optimal ends per hit)
    if (cap_hits > 0 && cap_hits < t.CAPH) t.CAPH = cap_hits;          // test hook: many small rounds
    if (cap_ents >= S && cap_ents < t.CAPE) t.CAPE = cap_ents;         // (a hit has at most S locations)
    // Order: regions whose offsets depend on the tile geometry only (R, S) come first -- compile-time constants in the
    // default-flags kernel --, then the per-alignment regions (R * 2 NP or the compact capacity), the panel tables last.
    int o = 0;
    t.tacc = o;  o += 11 * 8 + 8;
    t.lut = o;   o += 512;
    t.aggr = o;  o += 12 * 4;
    o = (o + 15) & ~15;
    t.codes = o; o += R * 2 * t.CS;
    t.namask = o; o += R * 2 * MW * 4;         // per code row: bit j set = code[j] is not A/C/G/T
    t.lens = o;  o += 2 * R * 4;               // double-buffered: the next tile is encoded while this one is scored
    t.ocnt = o;  o += 2 * R * 4;               // per read: forward votes | OCNT_NA (a window holds something other than upper-case
                                               // ACGT: scalar primer scan) | reverse votes << 16; double-buffered
    t.rflag = o;                               // (round 3: the flag is bit 15 of ocnt, no region of its own)
    o = (o + 7) & ~7;
    const bool cf = layout_cfilt(H, ncand);
    t.pmask = o; o += cf ? R * 8 : 0;          // per read: bit h = alignment h found its primer (scorer: which candidates to look at)
    o = (o + 15) & ~15;
    t.hits = o;  o += (t.NI + sentinel) * (int)sizeof(HitL);
    // one tie-mask word per record and nothing else wanting HitL::tail_end (it belongs to slots mode and --trim tails): the word
    // lives there, no array of its own
    t.tiem = o;  o += layout_tie_in_rec(t.MBW, slots, tails) ? 0 : (t.NI + sentinel) * t.MBW * 4;
    t.clist = o; o += nitems > 0 ? ((t.NI + 1) & ~1) * 2 : 0;       // record -> read * H + alignment
    // time-shared regions: {location entries} are dead after the barcode scan -> staged result records;
    // {primer end masks, scans, queue} are dead once the scorer starts -> its per-(read, candidate) trim shifts (Q8)
    t.bres = t.dmask = o; o += slots ? t.CAPH * t.G * 4 : t.CAPH * (kidx + 1) * t.MBW * 4;
    {
        int c = t.CAPE * (int)sizeof(EntL), d = R * 32;
        o = (o + 15) & ~15;
        t.ents = t.opsL = o; o += c > d ? c : d;
    }
    t.etail = o; o += (bs && !slots && tails) ? t.CAPE * 4 : 0;   // --trim tails on the lean path (BSV == 3): end of the kept alignment per entry
    t.masks = t.emit = o; o += t.NI * MW * 4;
    o = (o + 3) & ~3;
    t.offsA = o; o += ((t.NI + 2) & ~1) * 2;   // 16-bit scans: a tile has at most NI * S < 65536 locations
    t.offsB = o; o += ((t.NI + 2) & ~1) * 2 < 16 ? 16 : ((t.NI + 2) & ~1) * 2;   // (>= 16 bytes: the one-round path parks four wave sums here)
    t.queue = o; o += ((t.NI + 1) & ~1) * 2;
    if (o < t.emit + R * ncand * 4) o = t.emit + R * ncand * 4;
    o = (o + 7) & ~7;
    t.hmap = o;  o += nitems > 0 ? ((R * H + 1) & ~1) * 2 : 0;     // (read, alignment) -> record; t.NI = the shared "no match" record
    o = (o + 7) & ~7;
    t.hcand = o; o += cf ? H * 8 : 0;          // per alignment: the candidates (pair * 2 + orientation) it belongs to
    o = (o + 15) & ~15;
    t.ppeq = o;  o += t.NPs * 16 * (int)sizeof(PW);
    t.prpeq = o; o += (need_starts ? t.NPs * 16 * (int)sizeof(PW) : 0);
    t.bpeq = o;  o += (bs && !slots) ? 0 : t.NBs * 16 * 4;
    t.BSP = 16 * 16 + 4;                        // words per (primer, 32-barcode word) block of the bit-sliced table (+4: bank skew)
    t.bsre = o;  o += (bs && !slots) ? (nbstab > 0 ? nbstab : NP) * t.MBW * t.BSP * 4 : 0;   // one table per distinct barcode list
    t.pmeta = o; o += npmeta * 4;
    t.total = (o + 15) & ~15;
    return t;
}

// exclusive scan of a[0..n) in LDS by ONE wave (all 64 lanes of it call this); a[n] = total.
__device__ inline void wave_exclusive_scan(unsigned short *a, int n) {
    int lane = threadIdx.x & 63;
    int chunk = (n + 63) / 64;
    int lo = lane * chunk, hi = lo + chunk < n ? lo + chunk : n;
    int sum = 0;
    for (int i = lo; i < hi; i++) sum += a[i];
    int incl = sum;
    for (int d = 1; d < 64; d <<= 1) {
        int v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    int run = incl - sum;
    for (int i = lo; i < hi; i++) { int v = a[i]; a[i] = (unsigned short)run; run += v; }
    if (lane == 63) a[n] = (unsigned short)incl;
}

// loc_ord-th (0-based) optimal end at or after jstar in a location bitmask
__device__ __forceinline__ int nth_location(const unsigned *mrow, int MW, int jstar, int loc_ord) {
    int seen = 0;
    for (int w = jstar >> 5; w < MW; w++) {
        unsigned word = mrow[w];
        if (w == (jstar >> 5)) word &= ~0u << (jstar & 31);
        int pc = __popc(word);
        if (seen + pc > loc_ord) {
            for (int t = loc_ord - seen; t > 0; t--) word &= word - 1;
            return w * 32 + __ffs(word) - 1;
        }
        seen += pc;
    }
    return -1;
}

// ------------------------------------------------------------------------------------------------
// Bit-sliced SHW scan: ONE lane aligns up to 32 barcodes (one word of the primer's barcode list) against one
// target at once.  Bit b of every word belongs to barcode b.  Per DP cell (row i = barcode position, column c =
// target position) the unit-cost recurrence on the vertical / horizontal deltas in {-1,0,+1} is
//     Z = Eq | Mh_in | Mv_in                       (the cell's minimum is the diagonal value)
//     Ph_out = Mv_in | ~(Z | Pv_in)   Mh_out = Pv_in & Z      (bottom edge, handed to the next column)
//     Pv_out = Mh_in | ~(Z | Ph_in)   Mv_out = Ph_in & Z      (right edge, handed to the next row)
// with SHW boundaries D[0][j] = j, D[i][0] = i.  D[m][c] is kept as a bit-sliced 5-bit counter; seen[d] collects
// the barcodes whose last-row score equalled d (<= k) at some column: exactly what the per-hit distance-level
// bitmasks of the lean summary need (the lowest non-empty level is the best distance, its bits are the tie set).
// Ukkonen band: an alignment of cost <= k never leaves the cells with |column - row| <= k, so only those are
// computed (banded values D' >= D, and D' == D wherever D <= k -- all that seen[] needs).  No boundary special
// cases are required: a row below the band has never been touched and still holds its initial vertical delta
// (+1), which is what "left neighbour = infinity" means for the cell that enters the band; the top in-band cell
// takes (+1) as horizontal delta from above, like row 0 does.  The tracked score B_c = D'(bottom in-band row, c):
// while the band's bottom edge is still descending (c + k <= m - 1) the bottom cell has no left neighbour and
// B_c = B_(c-1) + (1 - Z); once it sits on the last row, B_c = B_(c-1) + (Ph - Mh) as in the full DP.  B_0 = k.
template <int KL>   // KL = number of distance levels kept (k + 1 <= KL): 4 or 8
__device__ __forceinline__ void bitsliced_shw(const unsigned *re, const unsigned char *cw, int ncol, int m,
                                              int kidx, unsigned (&seen)[KL]) {
    unsigned Pv[16], Mv[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { Pv[i] = ~0u; Mv[i] = 0u; }
    const int b0 = kidx < m ? kidx : m;   // D(bottom row of column 0)
    unsigned s0 = (b0 & 1) ? ~0u : 0u, s1 = (b0 & 2) ? ~0u : 0u, s2 = (b0 & 4) ? ~0u : 0u, s3 = (b0 & 8) ? ~0u : 0u,
             s4 = (b0 & 16) ? ~0u : 0u;
#pragma unroll
    for (int d = 0; d < KL; d++) seen[d] = 0u;
    const int ncols = m + kidx;
    constexpr int rs = 16;   // table block layout [row][code]
    for (int c = 0; c < ncols; c++) {
        const unsigned code = c < ncol ? (unsigned)cw[c] : 15u;   // past the window: code 15 matches nothing
        const unsigned *rc = re + code;
        const int rlo = c - kidx, rhi = (c + kidx < m - 1) ? c + kidx : m - 1;   // in-band rows of this column
        const unsigned rows = (rhi >= 0 ? (2u << rhi) - 1u : 0u) & (rlo > 0 ? ~0u << rlo : ~0u);   // one scalar test per row
        unsigned Ph = ~0u, Mh = 0u, Zb = 0u;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if ((rows >> i) & 1u) {
                const unsigned Eq = rc[i * rs];
                const unsigned Z = Eq | Mh | Mv[i];
                const unsigned nPh = Mv[i] | ~(Z | Pv[i]);
                const unsigned nMh = Pv[i] & Z;
                const unsigned nPv = Mh | ~(Z | Ph);
                const unsigned nMv = Ph & Z;
                Pv[i] = nPv; Mv[i] = nMv; Ph = nPh; Mh = nMh;
                Zb = Z;   // the last executed row's Z (rhi)
            }
        }
        unsigned inc, dec;
        if (c + kidx <= m - 1) { inc = ~Zb; dec = 0u; }   // bottom edge still descending: +1 unless the diagonal is free
        else { inc = Ph; dec = Mh; }                      // on the last row: horizontal delta of row m
        {   // score += inc - dec (disjoint masks), ripple through the five planes
            unsigned cy = inc, t;
            t = s0 & cy; s0 ^= cy; cy = t;
            t = s1 & cy; s1 ^= cy; cy = t;
            t = s2 & cy; s2 ^= cy; cy = t;
            t = s3 & cy; s3 ^= cy; cy = t;
            s4 ^= cy;
            unsigned bw = dec;
            t = ~s0 & bw; s0 ^= bw; bw = t;
            t = ~s1 & bw; s1 ^= bw; bw = t;
            t = ~s2 & bw; s2 ^= bw; bw = t;
            t = ~s3 & bw; s3 ^= bw; bw = t;
            s4 ^= bw;
        }
        if (c >= m - kidx - 1) {   // the tracked cell is on the last row from here on
            const unsigned live = c < ncol ? ~0u : 0u;
            const unsigned hi = ~(s4 | s3) & live;
#pragma unroll
            for (int d = 0; d < KL; d++)
                if (d <= kidx)
                    seen[d] |= hi & ((d & 1) ? s0 : ~s0) & ((d & 2) ? s1 : ~s1) & ((d & 4) ? s2 : ~s2);
        }
    }
}

// Banded DP with a fixed band half-width KB >= k (a wider band than k is still exact: it contains the k band); the
// 2*KB+1 live rows are kept in a circular register window (row r lives in slot r mod WIN, all slot indices static).
// Every barcode is PADDED to M rows by wildcard rows (Eq = all ones for every text code, also
// past the end of the window): the loop bounds no longer depend on the barcode length, so the whole scan unrolls into
// straight-line code -- static row tests, LDS reads at immediate offsets issued ahead of their use, no scalar branches.
// Exactness: a path that reaches (m, j) continues for free along its diagonal to (M, j + M - m), and every other way
// into row M costs at least as much, so min over the live columns of row M equals min over the live columns of row m;
// the lean summary only ever uses that minimum (the lowest non-empty distance level and its bits).  Column c' of row M
// is live iff c' - (M - m) lies inside the window.  M + KB columns instead of m + k: 19 vs 16 for 13-nt barcodes, at
// less than half the instructions per column.
template <int KB, int M>
__device__ __forceinline__ void bitsliced_shw_pad(const unsigned *re, const unsigned char *cw, int ncol, int m,
                                                  int kidx, unsigned (&seen)[KB + 1]) {
    constexpr int WIN = 2 * KB + 1, NC = M + KB;
    static_assert(KB < M && M <= 16, "band / padding out of range");
    unsigned Pw[WIN], Mw[WIN];
#pragma unroll
    for (int i = 0; i < WIN; i++) { Pw[i] = ~0u; Mw[i] = 0u; }
    constexpr int b0 = KB;   // D(bottom in-band row of column 0)
    unsigned s0 = (b0 & 1) ? ~0u : 0u, s1 = (b0 & 2) ? ~0u : 0u, s2 = (b0 & 4) ? ~0u : 0u, s3 = (b0 & 8) ? ~0u : 0u,
             s4 = 0u;
#pragma unroll
    for (int d = 0; d <= KB; d++) seen[d] = 0u;
    constexpr int rs = 16;   // table block layout [row][code]: every Eq read is base + code at an immediate offset
    const int nlive = ncol + (M - m);
#pragma unroll
    for (int c = 0; c < NC; c++) {
        // unconditional read (past the window it hits other LDS bytes, never out of the allocation's reach: rows are
        // followed by >= 32 bytes of other regions) + select: no branch, so the column stays one basic block
        const unsigned raw = (unsigned)cw[c];
        const unsigned code = c < ncol ? raw : 15u;
        const unsigned *rc = re + code;
        Pw[(c + KB) % WIN] = ~0u; Mw[(c + KB) % WIN] = 0u;   // the row entering the band: initial vertical delta
        unsigned Ph = ~0u, Mh = 0u, Zb = 0u;
#pragma unroll
        for (int w = 0; w < WIN; w++) {
            const int row = c - KB + w;
            const int sl = ((row % WIN) + WIN) % WIN;
            if (row >= 0 && row < M) {
                const unsigned Eq = rc[row * rs];
                const unsigned Z = Eq | Mh | Mw[sl];
                const unsigned nPh = Mw[sl] | ~(Z | Pw[sl]);
                const unsigned nMh = Pw[sl] & Z;
                const unsigned nPv = Mh | ~(Z | Ph);
                const unsigned nMv = Ph & Z;
                Pw[sl] = nPv; Mw[sl] = nMv; Ph = nPh; Mh = nMh;
                Zb = Z;
            }
        }
        const bool descending = c + KB <= M - 1;
        {
            unsigned cy = descending ? ~Zb : Ph, t;
            t = s0 & cy; s0 ^= cy; cy = t;
            t = s1 & cy; s1 ^= cy; cy = t;
            t = s2 & cy; s2 ^= cy; cy = t;
            t = s3 & cy; s3 ^= cy; cy = t;
            s4 ^= cy;
        }
        if (!descending) {
            unsigned bw = Mh, t;
            t = ~s0 & bw; s0 ^= bw; bw = t;
            t = ~s1 & bw; s1 ^= bw; bw = t;
            t = ~s2 & bw; s2 ^= bw; bw = t;
            t = ~s3 & bw; s3 ^= bw; bw = t;
            s4 ^= bw;
        }
        if (c >= M - KB - 1) {   // the tracked cell sits on the last row from here on
            const unsigned live = c < nlive ? ~0u : 0u;
            const unsigned hi = ~(s4 | s3) & live;
#pragma unroll
            for (int d = 0; d <= KB; d++)   // levels above k are computed too (no per-level select); the caller ignores them
                seen[d] |= hi & ((d & 1) ? s0 : ~s0) & ((d & 2) ? s1 : ~s1) & ((d & 4) ? s2 : ~s2);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the live ranges column-sized (otherwise ~130 LDS reads get hoisted and spill)
    }
}

// ------------------------------------------------------------------------------------------------
// The same scan for `--trim tails` (models.py:300-319): besides the distance levels it reports, for this location, the
// minimum level of every barcode (ML[d]) and the last live column at which a barcode selected by `want` sits at its own
// minimum -- the end of the alignment the reference keeps for that barcode (all optimal ends of its best alignment).
// Row M, column c' corresponds to row m, column c' - (M - m): a minimum of row m travels down its diagonal for free.
template <int KB, int M>
__device__ __forceinline__ void bitsliced_shw_pad_tails(const unsigned *re, const unsigned char *cw, int ncol, int m,
                                                        int kidx, const unsigned (&want)[KB + 1], unsigned (&seen)[KB + 1],
                                                        unsigned (&ML)[KB + 1], int &tailcol) {
    unsigned Hw[2 * KB + 1][KB + 1];
    constexpr int WIN = 2 * KB + 1, NC = M + KB;
    static_assert(KB < M && M <= 16, "band / padding out of range");
    unsigned Pw[WIN], Mw[WIN];
#pragma unroll
    for (int i = 0; i < WIN; i++) { Pw[i] = ~0u; Mw[i] = 0u; }
    constexpr int b0 = KB;   // D(bottom in-band row of column 0)
    unsigned s0 = (b0 & 1) ? ~0u : 0u, s1 = (b0 & 2) ? ~0u : 0u, s2 = (b0 & 4) ? ~0u : 0u, s3 = (b0 & 8) ? ~0u : 0u,
             s4 = 0u;
#pragma unroll
    for (int d = 0; d <= KB; d++) seen[d] = 0u;
    constexpr int rs = 16;   // table block layout [row][code]: every Eq read is base + code at an immediate offset
    const int nlive = ncol + (M - m);
#pragma unroll
    for (int c = 0; c < NC; c++) {
        // unconditional read (past the window it hits other LDS bytes, never out of the allocation's reach: rows are
        // followed by >= 32 bytes of other regions) + select: no branch, so the column stays one basic block
        const unsigned raw = (unsigned)cw[c];
        const unsigned code = c < ncol ? raw : 15u;
        const unsigned *rc = re + code;
        Pw[(c + KB) % WIN] = ~0u; Mw[(c + KB) % WIN] = 0u;   // the row entering the band: initial vertical delta
        unsigned Ph = ~0u, Mh = 0u, Zb = 0u;
#pragma unroll
        for (int w = 0; w < WIN; w++) {
            const int row = c - KB + w;
            const int sl = ((row % WIN) + WIN) % WIN;
            if (row >= 0 && row < M) {
                const unsigned Eq = rc[row * rs];
                const unsigned Z = Eq | Mh | Mw[sl];
                const unsigned nPh = Mw[sl] | ~(Z | Pw[sl]);
                const unsigned nMh = Pw[sl] & Z;
                const unsigned nPv = Mh | ~(Z | Ph);
                const unsigned nMv = Ph & Z;
                Pw[sl] = nPv; Mw[sl] = nMv; Ph = nPh; Mh = nMh;
                Zb = Z;
            }
        }
        const bool descending = c + KB <= M - 1;
        {
            unsigned cy = descending ? ~Zb : Ph, t;
            t = s0 & cy; s0 ^= cy; cy = t;
            t = s1 & cy; s1 ^= cy; cy = t;
            t = s2 & cy; s2 ^= cy; cy = t;
            t = s3 & cy; s3 ^= cy; cy = t;
            s4 ^= cy;
        }
        if (!descending) {
            unsigned bw = Mh, t;
            t = ~s0 & bw; s0 ^= bw; bw = t;
            t = ~s1 & bw; s1 ^= bw; bw = t;
            t = ~s2 & bw; s2 ^= bw; bw = t;
            t = ~s3 & bw; s3 ^= bw; bw = t;
            s4 ^= bw;
        }
        if (c >= M - KB - 1) {   // the tracked cell sits on the last row from here on
            const unsigned live = c < nlive ? ~0u : 0u;
            const unsigned hi = ~(s4 | s3) & live;
#pragma unroll
            for (int d = 0; d <= KB; d++) {
                const unsigned hd = hi & ((d & 1) ? s0 : ~s0) & ((d & 2) ? s1 : ~s1) & ((d & 4) ? s2 : ~s2);
                Hw[c - (M - KB - 1)][d] = hd;
                seen[d] |= hd;
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the live ranges column-sized (otherwise ~130 LDS reads get hoisted and spill)
    }
    unsigned lower = 0;
#pragma unroll
    for (int d = 0; d <= KB; d++) {
        ML[d] = d <= kidx ? (seen[d] & ~lower) : 0u;
        lower |= seen[d];
    }
    tailcol = -1;
#pragma unroll
    for (int ci = 0; ci < 2 * KB + 1; ci++) {
        unsigned any = 0;
#pragma unroll
        for (int d = 0; d <= KB; d++) any |= Hw[ci][d] & ML[d] & want[d];
        tailcol = any ? (M - KB - 1) + ci - (M - m) : tailcol;   // in columns of the unpadded problem
    }
}

// ------------------------------------------------------------------------------------------------

// Scorer helpers (phase 4).  Everything is indexed, nothing is string keyed.
// Small per-panel tables staged in LDS (the scorer and the barcode scan read them constantly).
struct LPanel {
    const int *pm, *pk, *pdir, *pfidx, *pbc_off, *pbc, *bm, *pair_f, *pair_r, *pair_pool, *bstab;
};

struct ReadCtx {
    const DevPanel *P;
    LPanel LP;
    const HitL *hits;          // this read's H records (compact mode: the tile's records, indexed through hmap)
    const unsigned *tiem;      // the records' tie bitmasks (barcodes at the best distance): word w of record i = tiem[i * tstr + w]
    int tstr;
    const unsigned short *hmap;   // compact mode: this read's alignment -> record; nullptr: record = alignment
    const unsigned long long *hcand;   // per alignment: bitmask of the candidates it belongs to; nullptr: look at every candidate
    int trim, derep;              // the panel's --trim / --dereplicate (compile-time constants in the default-flags kernel)
    int npair;                    // primer pairs (a constant 1 in the two-primer kernel)
    unsigned long long pm;        // this read's alignments with a primer match (bit h)
    unsigned long long live;      // candidates (pair * 2 + orientation) worth looking at: one of their alignments matched, and
                                  // the orientation is allowed; all ones when there is no candidate table (hcand == nullptr)
    // Panels with many primer pairs: a candidate scores > 0 only if one of its two alignments found its primer, so only the
    // candidates of this read's matched alignments are looked at (a handful of the 2 * NPAIR).  Nothing else changes: every
    // quantity the scorers compute is a function of the candidates with a positive score.
    __device__ __forceinline__ void set_live(int ori) {
        live = ~0ull;
        if (hcand) {
            live = 0ull;
            for (unsigned long long m = pm; m; m &= m - 1ull) live |= hcand[__ffsll((long long)m) - 1];
            if (ori == 2) live &= 0xAAAAAAAAAAAAAAAAull;        // reverse-complement candidates only (odd ids)
            else if (ori == 1) live &= 0x5555555555555555ull;
        }
    }
    __device__ __forceinline__ bool cand_live(int ci) const { return !hcand || ((live >> ci) & 1ull); }
    int MBW;
    __device__ __forceinline__ int rec(int h) const { return hmap ? (int)hmap[h] : h; }
    __device__ __forceinline__ const HitL &hit(int h) const { return hits[rec(h)]; }
    int L, S;
    EndGeom g;
};

struct CandView {
    int f, r, o;               // forward primer, reverse primer, orientation 0 = as read, 1 = reverse complement
    int h1, h2;                // hit indices (primer*2 + end)
    bool p1, p2, b1, b2;
    int p1d, p2d, b1d, b2d;
};

__device__ __forceinline__ CandView cand_view(const ReadCtx &c, int pair, int o) {
    CandView v;
    v.f = c.LP.pair_f[pair];
    v.r = c.LP.pair_r[pair];
    v.o = o;
    v.h1 = v.f * 2 + (o == 0 ? 0 : 1);   // forward primer: end A (of rs) when the read is kept as is
    v.h2 = v.r * 2 + (o == 0 ? 1 : 0);
    const HitL &a = c.hit(v.h1);
    const HitL &b = c.hit(v.h2);
    v.p1 = a.pdist >= 0;
    v.p2 = b.pdist >= 0;
    v.b1 = v.p1 && a.bbest >= 0;
    v.b2 = v.p2 && b.bbest >= 0;
    v.p1d = v.p1 ? a.pdist : -1;
    v.p2d = v.p2 ? b.pdist : -1;
    v.b1d = v.b1 ? a.bbest : -1;
    v.b2d = v.b2 ? b.bbest : -1;
    return v;
}

__device__ __forceinline__ int cand_score(const CandView &v) {   // demultiplex.py:226-236
    if (v.p1 && v.p2 && v.b1 && v.b2) return 5;
    if (v.p1 && v.p2 && (v.b1 || v.b2)) return 4;
    if ((v.p1 || v.p2) && (v.b1 || v.b2)) return 3;
    if (v.p1 && v.p2) return 2;
    if (v.p1 || v.p2) return 1;
    return 0;
}

// next tied barcode (local index >= from) of hit h in canonical order; -1 when exhausted.  (Find-first-set on the tie
// mask words: the scorers enumerate tie sets in nested loops, and a bit-by-bit scan of up to maxB positions per call was
// most of the general scorer's instruction stream.)
__device__ inline int next_tied(const ReadCtx &c, int h, int from) {
    int p = h >> 1;
    int nb = c.LP.pbc_off[p + 1] - c.LP.pbc_off[p];
    if (from >= nb) return -1;
    const unsigned *tm = c.tiem + c.rec(h) * c.tstr;
    int w = from >> 5;
    unsigned word = tm[w] & (~0u << (from & 31));
    for (;;) {
        if (word) {
            const int i = w * 32 + __ffs((int)word) - 1;
            return i < nb ? i : -1;
        }
        if (++w >= c.MBW) return -1;
        word = tm[w];
    }
}
__device__ __forceinline__ int global_bc(const ReadCtx &c, int h, int local) {
    return c.LP.pbc[c.LP.pbc_off[h >> 1] + local];
}
__device__ inline bool tied_has_global(const ReadCtx &c, int h, int gb) {
    for (int i = next_tied(c, h, 0); i >= 0; i = next_tied(c, h, i + 1))
        if (global_bc(c, h, i) == gb) return true;
    return false;
}

// Specimens.specimen_for_exact_match (databases.py:232-245): first specimen in file order.
// One 32-byte load in the usual case -- the (b1, b2) cell holds the first specimen's whole record (masks, pool, next) --
// instead of three dependent ones (head of the chain, its primer masks, its pool): the scorer wave sits on a tile's
// critical path and every dependent global load is most of a microsecond.  The flat tables are gone from the device:
// five fewer pointers held in (spilled) SGPRs.
__device__ inline int specimen_exact(const DevPanel *P, int gb1, int gb2, int f, int r, int *pool_out = nullptr) {
    SpecRec rec = P->pairrec[(size_t)gb1 * P->NB + gb2];
    while (rec.spec >= 0) {
        if (((rec.p1m >> f) & 1) && ((rec.p2m >> r) & 1)) {
            if (pool_out) *pool_out = rec.pool;
            return rec.spec;
        }
        if (rec.next < 0) break;
        rec = P->specrec[rec.next];
    }
    return -1;
}
// Specimens.specimens_for_barcodes_and_primers (databases.py): how many specimens carry (b1, b2) with these primers, and
// the first of them in file order
__device__ inline void specimens_for(const DevPanel *P, int gb1, int gb2, int f, int r, int &first, int &cnt) {
    SpecRec rec = P->pairrec[(size_t)gb1 * P->NB + gb2];
    while (rec.spec >= 0) {
        if (((rec.p1m >> f) & 1) && ((rec.p2m >> r) & 1)) {
            cnt++;
            if (first < 0 || rec.spec < first) first = rec.spec;
        }
        if (rec.next < 0) break;
        rec = P->specrec[rec.next];
    }
}

// Trim extents of a candidate whose stored locations were already shifted by `cum` (Q8).
// models.py:278-319 with p1 locations mirrored by AlignmentResult.reversed (models.py:58-63).
__device__ inline void cand_extent(const ReadCtx &c, const CandView &v, int cum, int &s, int &e) {
    const DevPanel *P = c.P;
    int L = c.L;
    const HitL &a = c.hit(v.h1);
    const HitL &b = c.hit(v.h2);
    // first primer location in reference coordinates of the searched string
    int a_end = 0, a_start = 0, b_end = 0, b_start = 0;
    if (v.p1) { a_end = (int)a.jstar - c.g.j_lo + c.g.shift; a_start = (int)a.fs_j - c.g.j_lo + c.g.shift; }
    if (v.p2) { b_end = (int)b.jstar - c.g.j_lo + c.g.shift; b_start = (int)b.fs_j - c.g.j_lo + c.g.shift; }
    s = 0; e = L;
    if (c.trim == SMX_TRIM_BARCODES) {
        if (v.p1) s = (L - a_end - 1) - cum;
        if (v.p2) e = (b_end + 1) - cum;
    } else if (c.trim == SMX_TRIM_PRIMERS || c.trim == SMX_TRIM_TAILS) {
        int ps = 0, pe = L;
        if (v.p1) ps = (L - a_start - 1) + 1 - cum;
        if (v.p2) pe = b_start - cum;
        if (c.trim == SMX_TRIM_PRIMERS) { s = ps; e = pe; }
        else {
            if (v.b1) s = (L - a.tail_end - 1) - cum;
            else { s = ps - P->bmax; if (s < 0) s = 0; }
            if (v.b2) e = (b.tail_end + 1) - cum;
            else { e = pe + P->bmax; if (e > L) e = L; }
        }
    }
}

struct Emitter {
    const ReadCtx *c;
    smx_op *primary;           // LDS staging slot of this read's first record
    smx_op *extra;
    unsigned extra_cap;
    unsigned *n_extra;
    unsigned long long *counts;
    int *cum;                  // LDS, one int per candidate of this read: the trim shift its earlier emissions have added up (Q8)
    int *aggr;                 // LDS block aggregates
    unsigned read;
    int n;                     // ops emitted so far
    bool matched;
    bool overflow;
};

// create_write_operation (demultiplex.py:30-103) in index form.
__device__ inline void emit_op(Emitter &E, const CandView *v, int cand_id, int sample, int rtype, int pool,
                               int barcode, unsigned xflags) {
    const ReadCtx &c = *E.c;
    smx_op op;
    op.read = E.read;
    op.n_ops = 0;
    op.flags = (unsigned char)xflags;
    op.barcode = (short)barcode;
    int s = 0, e = c.L;
    if (v) {
        op.dist[0] = (int8_t)v->p1d; op.dist[1] = (int8_t)v->b1d; op.dist[2] = (int8_t)v->b2d; op.dist[3] = (int8_t)v->p2d;
        op.p1 = (short)(v->p1 ? v->f : -1);
        op.p2 = (short)(v->p2 ? v->r : -1);
        if (v->o) op.flags |= SMX_OPF_REVERSE;
    } else {
        op.dist[0] = op.dist[1] = op.dist[2] = op.dist[3] = -1;
        op.p1 = op.p2 = -1;
    }
    bool fallback = false;
    if (c.trim != SMX_TRIM_NONE) {
        if (v) {
            // Q8: the same CandidateMatch emitted earlier already had its locations shifted by the trim start of each of
            // those emissions.  One running sum per candidate: any number of emissions per read (a tie storm at a large
            // index distance emits one record per tied specimen, hundreds for one read).
            const int cum = E.cum[cand_id];
            cand_extent(c, *v, cum, s, e);
            if (s < e) E.cum[cand_id] = cum + s;
        }
        if (s >= e) { fallback = true; s = 0; e = c.L; }   // Q12
    }
    op.trim_start = s;
    op.trim_end = e;
    if (fallback) {
        op.flags |= SMX_OPF_TRIM_EMPTY;
        op.sample = -1; op.pool = -1; op.p1 = -1; op.p2 = -1; op.barcode = -1;
        op.rtype = SMX_R_UNKNOWN;
    } else {
        op.sample = sample;
        op.pool = (short)pool;
        op.rtype = (unsigned char)rtype;
    }
    if (rtype == SMX_R_FULL || rtype == SMX_R_DEREP_FULL) E.matched = true;
    // counters
    int cls = (op.rtype == SMX_R_UNKNOWN) ? 2 : ((op.rtype == SMX_R_PARTIAL_FWD || op.rtype == SMX_R_PARTIAL_REV) ? 1 : 0);
    atomicAdd(&E.aggr[3 + cls], 1);
    if (cls == 0 && op.sample >= 0) atomicAdd(&E.counts[SMX_CNT_SPECIMEN0 + op.sample], 1ull);
    if (E.n == 0) {
        *E.primary = op;
    } else {
        unsigned slot = atomicAdd(E.n_extra, 1u);
        if (slot < E.extra_cap) E.extra[slot] = op;
    }
    E.n++;
    if (E.n > 0xFFFF) E.overflow = true;   // n_ops is 16 bits wide
}

// resolve_specimen for a non-full candidate (demultiplex.py:576-589) -> emits.
__device__ inline void emit_partial_or_unknown(Emitter &E, const CandView &v, int cand_id, int pool) {
    const ReadCtx &c = *E.c;
    if (v.b1 && !v.b2 && c.hit(v.h1).ntied == 1)
        emit_op(E, &v, cand_id, -1, SMX_R_PARTIAL_FWD, pool, global_bc(c, v.h1, c.hit(v.h1).first_tied), 0);
    else if (v.b2 && !v.b1 && c.hit(v.h2).ntied == 1)
        emit_op(E, &v, cand_id, -1, SMX_R_PARTIAL_REV, pool, global_bc(c, v.h2, c.hit(v.h2).first_tied), 0);
    else
        emit_op(E, &v, cand_id, -1, SMX_R_UNKNOWN, pool, -1, 0);
}

// Iterate the best-scoring candidates in find_candidate_matches order.
#define FOR_BEST_CANDS(c, ori, best, ...)                                               \
    for (int _pair = 0; _pair < (c).npair; _pair++)                                      \
        for (int _o = 0; _o < 2; _o++) {                                                 \
            if ((_o == 0 && (ori) == 2) || (_o == 1 && (ori) == 1)) continue;            \
            if (!(c).cand_live(_pair * 2 + _o)) continue;                                \
            CandView v = cand_view((c), _pair, _o);                                      \
            if (!(v.p1 || v.p2)) continue;                                               \
            if (cand_score(v) != (best)) continue;                                       \
            int cand_id = _pair * 2 + _o;                                                \
            int cand_pool = (c).LP.pair_pool[_pair];                                     \
            (void)cand_id; (void)cand_pool;                                              \
            __VA_ARGS__                                                                  \
        }

// key for dereplicate_matches' specimen groups (demultiplex.py:371-378); all lexicographic, small ints
__device__ __forceinline__ int key_full(const LPanel &P, const CandView &v) {
    return ((v.b1d + v.b2d) << 20) | ((v.p1d + v.p2d) << 12) | (P.pfidx[v.f] + P.pfidx[v.r]);
}
__device__ __forceinline__ int key_partial(const LPanel &P, const CandView &v, bool fwd) {   // :442-463
    int cnt = (v.p1 ? 1 : 0) + (v.p2 ? 1 : 0);
    int pd = (v.p1 ? v.p1d : 0) + (v.p2 ? v.p2d : 0);
    int fi = (v.p1 ? P.pfidx[v.f] : 0) + (v.p2 ? P.pfidx[v.r] : 0);
    return ((fwd ? v.b1d : v.b2d) << 24) | ((2 - cnt) << 22) | (pd << 12) | fi;
}
__device__ __forceinline__ int key_unknown(const LPanel &P, const CandView &v) {             // :502-528
    int cnt = (v.p1 ? 1 : 0) + (v.p2 ? 1 : 0);
    int pd = (v.p1 ? v.p1d : 0) + (v.p2 ? v.p2d : 0);
    int fi = (v.p1 ? P.pfidx[v.f] : 999) + (v.p2 ? P.pfidx[v.r] : 999);
    return ((2 - cnt) << 22) | (pd << 12) | fi;
}

// does candidate v map some tied (b1,b2) combination to specimen `spec`?  (spec = -1: to ANY specimen)
__device__ inline bool cand_has_specimen(const ReadCtx &c, const CandView &v, int spec) {
    for (int i = next_tied(c, v.h1, 0); i >= 0; i = next_tied(c, v.h1, i + 1))
        for (int j = next_tied(c, v.h2, 0); j >= 0; j = next_tied(c, v.h2, j + 1)) {
            int s = specimen_exact(c.P, global_bc(c, v.h1, i), global_bc(c, v.h2, j), v.f, v.r);
            if (s >= 0 && (spec < 0 || s == spec)) return true;
        }
    return false;
}

// Fast scorer of the hot kernel: handles "no candidate" and "exactly one candidate at the best score, no
// barcode tie".  Every dereplication group then has one member and the reference's machinery reduces to a
// single emission.  Returns false for everything else (several best candidates, ties): the caller then runs
// score_general, the reference's selection / dereplication in full, for that read.
// `G` lanes (a power of two, consecutive, `sub` = this lane's index among them) share one read: the candidate loops are
// split over them and reduced by xor-shuffles (tiles of panels with many primers hold fewer than 64 reads, so the scorer
// wave has lanes to spare, and with 10 candidates the loops are most of its instruction stream); everything after the
// reductions -- one record in the usual case -- is done by sub-lane 0, the others return true at once.
__device__ inline bool score_fast(Emitter &E, int ori, int sub, int G) {
    const ReadCtx &c = *E.c;
    const DevPanel *P = c.P;
    // ---- select_best_matches (demultiplex.py:216-259): best score, how many carry it, the first of them
    int best = 0, nbest = 0, first = 0x7FFF;
    const int ncand = c.npair * 2;
    const unsigned long long live = c.live;   // (ReadCtx::set_live)
    if (c.hcand) {
        for (unsigned long long m = live; m; m &= m - 1ull) {
            const int ci = __ffsll((long long)m) - 1;
            if ((ci & (G - 1)) != sub) continue;
            CandView v = cand_view(c, ci >> 1, ci & 1);
            int sc = cand_score(v);
            if (sc > best) { best = sc; nbest = 1; first = ci; }
            else if (sc == best) { nbest++; first = ci < first ? ci : first; }
        }
    } else
    for (int ci = sub; ci < ncand; ci += G) {
        const int pair = ci >> 1, o = ci & 1;
        if ((o == 0 && ori == 2) || (o == 1 && ori == 1)) continue;
        CandView v = cand_view(c, pair, o);
        int sc = cand_score(v);
        if (sc > best) { best = sc; nbest = 1; first = ci; }
        else if (sc == best) { nbest++; first = ci < first ? ci : first; }
    }
    for (int d = 1; d < G; d <<= 1) {
        const int ob = __shfl_xor(best, d, 64), on = __shfl_xor(nbest, d, 64), of = __shfl_xor(first, d, 64);
        if (ob > best) { best = ob; nbest = on; first = of; }
        else if (ob == best) { nbest += on; first = of < first ? of : first; }
    }
    const int only_pair = first >> 1, only_o = first & 1;
    // The divergent case analysis below only picks the parameters of the ONE record this read emits; the record
    // itself is built once, after the lanes have converged again (emit_op is by far the longest piece of code here).
    bool has_v = true;
    int pair = only_pair, o = only_o, sample = -1, rtype = SMX_R_UNKNOWN, barcode = -1;
    const bool lead = sub == 0;   // the lane that analyses the winner and emits; its partners only take part in the shuffles
    unsigned xflags = 0;
    int more1 = -1, more2 = -1, more3 = -1;   // further specimens of a tied full match
    int repeat = 1;                           // identical UNKNOWN records (one per tied barcode of a partial match)
    int spool = -2;                           // the specimen's pool when the lookup already brought it along
    if (best == 0) {   // no candidate at all (demultiplex.py:202-210)
        has_v = false; pair = 0; o = 0;
    } else if (best <= 2 && nbest > 1) {
        // several primer-only candidates (typically both orientations of one pair): no barcode logic involved.
        // dereplicate=best -> dereplicate_unknown_matches: stable minimum of (-primer_count, primer_dist, file index);
        // dereplicate=none -> every best candidate is written as UNKNOWN (demultiplex.py:181-197, :480-538): general path
        if (c.derep == SMX_DEREP_NONE) return sub != 0;
        int wkey = 0x7FFFFFFF, wci = 0x7FFF;
        for (int ci = sub; ci < ncand; ci += G) {
            const int pr = ci >> 1, oo = ci & 1;
            if ((oo == 0 && ori == 2) || (oo == 1 && ori == 1)) continue;
            if (!((live >> ci) & 1ull)) continue;
            CandView v = cand_view(c, pr, oo);
            if (cand_score(v) != best) continue;
            int k = key_unknown(c.LP, v);
            if (k < wkey) { wkey = k; wci = ci; }   // first minimum in candidate order within this lane's stride
        }
        for (int d = 1; d < G; d <<= 1) {
            const int ok = __shfl_xor(wkey, d, 64), oc = __shfl_xor(wci, d, 64);
            if (ok < wkey || (ok == wkey && oc < wci)) { wkey = ok; wci = oc; }
        }
        pair = wci >> 1; o = wci & 1;
    } else if (best <= 4 && nbest > 1) {
        // several partial candidates (typically the pairs that share one primer, when only that primer's end of the read is
        // good).  dereplicate_partial_matches (demultiplex.py:396-477) groups them by (direction, barcode); when every one
        // of them has a single untied barcode and it is the same (direction, barcode) for all, there is one group and its
        // winner -- stable minimum of key_partial -- is the one record.  Anything else: the general scorer.
        if (!lead) return true;
        if (c.derep != SMX_DEREP_BEST) return false;
        int gdir = -1, ggb = -1, wkey = 0x7FFFFFFF, wci = -1;
        for (int ci = 0; ci < ncand; ci++) {
            if (!((live >> ci) & 1ull)) continue;
            if (((ci & 1) == 0 && ori == 2) || ((ci & 1) == 1 && ori == 1)) continue;
            const CandView v = cand_view(c, ci >> 1, ci & 1);
            if (cand_score(v) != best) continue;
            const bool fwd = v.b1;
            const int h = fwd ? v.h1 : v.h2;
            const HitL &x = c.hit(h);
            if (x.ntied != 1) return false;
            const int gb = global_bc(c, h, x.first_tied);
            if (gdir < 0) { gdir = fwd ? 1 : 0; ggb = gb; }
            else if (gdir != (fwd ? 1 : 0) || ggb != gb) return false;
            const int k = key_partial(c.LP, v, fwd);
            if (k < wkey) { wkey = k; wci = ci; }
        }
        pair = wci >> 1; o = wci & 1;
        const CandView w = cand_view(c, pair, o);   // resolve_specimen of the winner (demultiplex.py:576-589)
        if (w.b1 && !w.b2) { rtype = SMX_R_PARTIAL_FWD; barcode = ggb; }
        else if (w.b2 && !w.b1) { rtype = SMX_R_PARTIAL_REV; barcode = ggb; }
    } else {
        if (!lead) return true;
        if (nbest != 1) return false;
        const CandView v = cand_view(c, pair, o);
        const HitL &a = c.hit(v.h1), &b = c.hit(v.h2);
        const bool t1 = !v.b1 || a.ntied == 1, t2 = !v.b2 || b.ntied == 1;
        if (best <= 2) {
            // dereplicate_unknown_matches / resolve_specimen: UNKNOWN either way
        } else if (best <= 4) {   // one (direction, barcode) group: resolve_specimen (demultiplex.py:576-589)
            if (!(t1 && t2)) {
                // tied barcodes on the one end that has any: dereplicate_partial_matches makes one group per tied barcode,
                // each with this candidate as its only member, and resolve_specimen turns each into UNKNOWN (the barcode is
                // ambiguous): ntied identical records
                if (c.derep != SMX_DEREP_BEST) return false;
                repeat = v.b1 ? a.ntied : b.ntied;
            }
            else if (v.b1 && !v.b2) { rtype = SMX_R_PARTIAL_FWD; barcode = global_bc(c, v.h1, a.first_tied); }
            else if (v.b2 && !v.b1) { rtype = SMX_R_PARTIAL_REV; barcode = global_bc(c, v.h2, b.first_tied); }
        } else if (c.derep != SMX_DEREP_BEST) {
            // --dereplicate none: resolve_specimen's full branch (demultiplex.py:555-575) for untied barcodes --
            // specimens_for_barcodes_and_primers in file order: one -> FULL_MATCH, several -> the first + MULTIPLE
            if (!(t1 && t2)) return false;
            const int g1 = global_bc(c, v.h1, a.first_tied), g2 = global_bc(c, v.h2, b.first_tied);
            int first = -1, cnt = 0;
            specimens_for(P, g1, g2, v.f, v.r, first, cnt);
            if (cnt > 0) { sample = first; rtype = cnt > 1 ? SMX_R_MULTIPLE : SMX_R_FULL; }
            else xflags = SMX_OPF_NO_SPECIMEN;
        } else {
            if (t1 && t2) {
                int spec = specimen_exact(P, global_bc(c, v.h1, a.first_tied), global_bc(c, v.h2, b.first_tied), v.f, v.r, &spool);
                if (spec >= 0) { sample = spec; rtype = SMX_R_DEREP_FULL; }
                else xflags = SMX_OPF_NO_SPECIMEN;
            } else {
                // one candidate, tied barcodes (by far the most frequent reason to leave the single-record path):
                // dereplicate_matches (demultiplex.py:262-393) with one member per group = one DEREP_FULL record per
                // distinct specimen among the tied (b1, b2) combinations, in first-appearance order
                int s0 = -1, s1 = -1, s2 = -1, s3 = -1, ns = 0;
                for (int i = next_tied(c, v.h1, 0); i >= 0; i = next_tied(c, v.h1, i + 1))
                    for (int j = next_tied(c, v.h2, 0); j >= 0; j = next_tied(c, v.h2, j + 1)) {
                        int sp = specimen_exact(P, global_bc(c, v.h1, i), global_bc(c, v.h2, j), v.f, v.r);
                        if (sp < 0 || sp == s0 || sp == s1 || sp == s2 || sp == s3) continue;
                        if (ns == 4) return false;   // more groups than this path keeps: the general scorer takes over
                        if (ns == 0) s0 = sp; else if (ns == 1) s1 = sp; else if (ns == 2) s2 = sp; else s3 = sp;
                        ns++;
                    }
                if (ns == 0) xflags = SMX_OPF_NO_SPECIMEN;
                else {
                    sample = s0; rtype = SMX_R_DEREP_FULL;
                    more1 = s1; more2 = s2; more3 = s3;
                }
            }
        }
    }
    if (!lead) return true;
    const CandView v = cand_view(c, pair, o);
    const int pool = !has_v ? -1 : (sample >= 0 ? (spool != -2 ? spool : P->specrec[sample].pool) : c.LP.pair_pool[pair]);
    emit_op(E, has_v ? &v : nullptr, pair * 2 + o, sample, rtype, pool, barcode, xflags);
    for (int e = 1; e < repeat; e++) emit_op(E, &v, pair * 2 + o, -1, SMX_R_UNKNOWN, pool, -1, 0);   // rare
    if (more1 >= 0) {   // rare
        for (int e = 0; e < 3; e++) {
            int sp = e == 0 ? more1 : (e == 1 ? more2 : more3);
            if (sp >= 0) emit_op(E, &v, pair * 2 + o, sp, SMX_R_DEREP_FULL, P->specrec[sp].pool, -1, 0);
        }
    }
    return true;
}

// General scorer: the reference's selection / dereplication in full (reads score_fast declines, ~0.1 %).
__device__ inline void score_general(Emitter &E, int ori) {
    const ReadCtx &c = *E.c;
    const DevPanel *P = c.P;
    int best = 0;
    for (int pair = 0; pair < c.npair; pair++)
        for (int o = 0; o < 2; o++) {
            if ((o == 0 && ori == 2) || (o == 1 && ori == 1)) continue;
            if (!c.cand_live(pair * 2 + o)) continue;
            CandView v = cand_view(c, pair, o);
            int sc = cand_score(v);
            best = sc > best ? sc : best;
        }
    if (best == 0) {
        emit_op(E, nullptr, 0, -1, SMX_R_UNKNOWN, -1, -1, 0);
        return;
    }
    if (c.derep == SMX_DEREP_NONE) {   // demultiplex.py:181-197
        FOR_BEST_CANDS(c, ori, best, {
            if (best == 5) {   // resolve_specimen full branch (:555-575): specimens_for_barcodes_and_primers
                int first = -1, cnt = 0;
                for (int i = next_tied(c, v.h1, 0); i >= 0; i = next_tied(c, v.h1, i + 1))
                    for (int j = next_tied(c, v.h2, 0); j >= 0; j = next_tied(c, v.h2, j + 1)) {
                        int g1 = global_bc(c, v.h1, i), g2 = global_bc(c, v.h2, j);
                        specimens_for(P, g1, g2, v.f, v.r, first, cnt);
                    }
                if (cnt > 1) emit_op(E, &v, cand_id, first, SMX_R_MULTIPLE, P->specrec[first].pool, -1, 0);
                else if (cnt == 1) emit_op(E, &v, cand_id, first, SMX_R_FULL, P->specrec[first].pool, -1, 0);
                else emit_op(E, &v, cand_id, -1, SMX_R_UNKNOWN, cand_pool, -1, SMX_OPF_NO_SPECIMEN);
            } else {
                emit_partial_or_unknown(E, v, cand_id, cand_pool);
            }
        })
        return;
    }
    // ---- dereplicate_matches (demultiplex.py:262-393); the best list is homogeneous in score
    if (best == 5) {
        // expanded entries in order: (candidate, tied b1, tied b2) -> specimen, or one None entry per
        // candidate without any specimen.  Groups keep first-appearance order (dict insertion).
        bool none_done = false;
        FOR_BEST_CANDS(c, ori, best, {
            bool any = false;
            for (int i = next_tied(c, v.h1, 0); i >= 0; i = next_tied(c, v.h1, i + 1))
                for (int j = next_tied(c, v.h2, 0); j >= 0; j = next_tied(c, v.h2, j + 1)) {
                    int spec = specimen_exact(P, global_bc(c, v.h1, i), global_bc(c, v.h2, j), v.f, v.r);
                    if (spec < 0) continue;
                    any = true;
                    // first appearance of this specimen? (earlier candidate, or earlier combo of this one)
                    bool seen = false;
                    {
                        const int cur_id = cand_id; const int ci = i; const int cj = j;
                        FOR_BEST_CANDS(c, ori, best, {
                            if (seen || cand_id > cur_id) continue;
                            if (cand_id < cur_id) { if (cand_has_specimen(c, v, spec)) seen = true; continue; }
                            for (int i2 = next_tied(c, v.h1, 0); i2 >= 0 && !seen; i2 = next_tied(c, v.h1, i2 + 1))
                                for (int j2 = next_tied(c, v.h2, 0); j2 >= 0; j2 = next_tied(c, v.h2, j2 + 1)) {
                                    if (i2 > ci || (i2 == ci && j2 >= cj)) break;
                                    if (specimen_exact(P, global_bc(c, v.h1, i2), global_bc(c, v.h2, j2), v.f, v.r) == spec) { seen = true; break; }
                                }
                        })
                    }
                    if (seen) continue;
                    // the group's winner: stable min of (b1d+b2d, p1d+p2d, file index sum) over its entries
                    int wkey = 0x7FFFFFFF, wid = -1;
                    FOR_BEST_CANDS(c, ori, best, {
                        int k = key_full(c.LP, v);
                        if (k < wkey && cand_has_specimen(c, v, spec)) { wkey = k; wid = cand_id; }
                    })
                    CandView w = cand_view(c, wid >> 1, wid & 1);
                    emit_op(E, &w, wid, spec, SMX_R_DEREP_FULL, P->specrec[spec].pool, -1, 0);
                }
            if (!any && !none_done) {
                // the None group sits where its first entry appeared; it holds every best candidate
                // without a specimen ("other_matches", :361-363) -> resolve_specimen -> UNKNOWN (Q10)
                none_done = true;
                const int first_id = cand_id;
                FOR_BEST_CANDS(c, ori, best, {
                    if (cand_id < first_id) continue;
                    if (!cand_has_specimen(c, v, -1))
                        emit_op(E, &v, cand_id, -1, SMX_R_UNKNOWN, cand_pool, -1, SMX_OPF_NO_SPECIMEN);
                })
            }
        })
        return;
    }
    if (best >= 3) {
        // dereplicate_partial_matches (:396-477): groups keyed (direction, barcode) for every tied barcode
        FOR_BEST_CANDS(c, ori, best, {
            bool fwd = v.b1;
            int h = fwd ? v.h1 : v.h2;
            for (int i = next_tied(c, h, 0); i >= 0; i = next_tied(c, h, i + 1)) {
                int gb = global_bc(c, h, i);
                const int cur_id = cand_id;
                bool seen = false;
                int wkey = 0x7FFFFFFF, wid = -1;
                FOR_BEST_CANDS(c, ori, best, {
                    if (v.b1 != fwd) continue;
                    int h2 = fwd ? v.h1 : v.h2;
                    if (!tied_has_global(c, h2, gb)) continue;
                    if (cand_id < cur_id) seen = true;
                    int k = key_partial(c.LP, v, fwd);
                    if (k < wkey) { wkey = k; wid = cand_id; }
                })
                if (seen) continue;
                CandView w = cand_view(c, wid >> 1, wid & 1);
                emit_partial_or_unknown(E, w, wid, c.LP.pair_pool[wid >> 1]);
            }
        })
        return;
    }
    // dereplicate_unknown_matches (:480-538): one stable minimum
    {
        int wkey = 0x7FFFFFFF, wid = -1;
        FOR_BEST_CANDS(c, ori, best, {
            int k = key_unknown(c.LP, v);
            if (k < wkey) { wkey = k; wid = cand_id; }
        })
        CandView w = cand_view(c, wid >> 1, wid & 1);
        emit_op(E, &w, wid, -1, SMX_R_UNKNOWN, c.LP.pair_pool[wid >> 1], -1, 0);
    }
}

// ------------------------------------------------------------------------------------------------
// BSV selects the barcode scan compiled into the kernel (one variant per kernel keeps their register allocations
// apart): 0 = per-barcode bit-vector scan only, 1 = bit-sliced, k <= 3 (padded 7-row window), 2 = bit-sliced, k 4..7,
// 3 = variant 1 + the --trim tails extent on the lean path (<= 32 barcodes per primer).
// CM = 1: compact mode compiled in (its own instantiation: the dense kernel's register allocation stays as it was).
// CM = 2: the dense redo launch behind a compact one (tiles come from the overflow list); the plain dense kernel (CM = 0)
// carries none of that state.
// SP = 1: the reference's default flags as compile-time facts -- search_len 80, index edit distance 3, --trim barcodes (no
// primer start scans), --dereplicate best, pre-orientation on, no length filter -- and at most 32 barcodes per primer (one
// tie-mask word).  The kernel holds > 100 uniform
// values and spills hundreds of SGPRs; every dimension that is a constant is one fewer of them, and the loops over mask
// words / distance levels / window chunks get constant trip counts.
// The kernel's state and its phases.  One object per thread, every member function inlined: after inlining the members are
// plain registers / uniform values, exactly as the locals of the former single 970-line function were -- but each phase can be
// read (and a variant reasoned about) on its own: setup -> stage_panel (phase 0) -> run: per tile phase2_primers ->
// phase3a_entries -> phase3_rounds (phase3b_barcodes, phase3c_summary) -> zero_for_next -> phase4_score || phase1_encode ->
// phase5_store.
template <typename PW, int NT, int BSV, int CM, int SP>
struct DemuxTile {
    static constexpr bool sp = SP != 0;
    // compact mode (aux.nitems > 0, lean launches of many-primer panels behind the prescan): records only for the
    // alignments the prescan's match words flag; a tile with more flagged alignments than records is put on the
    // overflow list and left to the dense redo launch that follows (aux.redo)
    static constexpr bool cmode = CM == 1;
    static constexpr bool redo = CM == 2;
    static constexpr int PWBITS = (int)sizeof(PW) * 8;
    // ---- kernel arguments
    unsigned char *lds;
    const DevPanel *P;
    const uint8_t *windows;
    const int32_t *lens;
    uint32_t n_reads;
    smx_op *ops, *extra;
    uint32_t extra_cap;
    uint32_t *n_extra;
    unsigned long long *counts;
    smx_hit *dbg_hits;
    int8_t *dbg_bdist;
    unsigned *tile_counter;
    const unsigned *pre;
    DemuxAux aux;
    // ---- launch-wide facts (compile-time constants in the default-flags kernels)
    int R, NP, NB, S, H, MW, maxB, need_starts, n_pbc, NPAIR, npmeta, use_slots, use_bs, ncand;
    TileLayout T;
    int CS, NPs, NBs, lNPs, lNBs, G, logG, MBW;
    int stride, kidx, pfmin, preorient, minlen, maxlen, lH, SW;
    unsigned Hmagic, hcmagic;
    uint32_t redo_ratio, n_tiles;
    bool cfilt, timing;
    unsigned long long *tacc, *dbg_phase;
    // ---- LDS regions (layout: make_layout)
    PW *ppeq, *prpeq;
    unsigned *bpeq, *bsre;
    unsigned char *lut, *codes;
    unsigned *namask;
    int *lensL, *ocnt;
    HitL *hits;
    unsigned *masks, *tieb, *bres, *dmask;   // tieb[i * tstr + w]: tie-mask word w of record i
    int tstr;
    EntL *ents;
    int *etail;
    unsigned short *offsA, *offsB;
    unsigned short *queue;
    int *cumL;
    smx_op *opsL;
    int *aggr;
    unsigned short *hmap, *clist;
    unsigned long long *pmask, *hcand;
    LPanel LP;
    // ---- per thread / per tile
    int tid, wave;
    uint32_t cur, r0, nxt;      // tile being processed, its first read, the tile being encoded beside its scorer
    int par, nr, nh, nI;        // cur's lens/ocnt buffer; reads, alignments, alignments with a record
    bool live;                  // false: nothing to do for `cur` (none yet, or a compact tile left to the redo launch)
    unsigned popped;
    int *lensC, *ocntC;
    bool prelisted;
    int nq, nE_pre;

    // The launch counters re-arm themselves: tile_counter = {tile queue head, overflow tiles, finished workgroups, extra
    // records}.  The last workgroup to get here zeroes the queue head; the last launch of a batch (aux.chain == 0) also
    // publishes the extra-record count and zeroes the rest, so a launch needs no memset in front of it (two small fills
    // per batch were ~2 % of a 0.4 ms launch).
    __device__ __forceinline__ void workgroup_done() {
        __threadfence();
        const unsigned prev = atomicAdd(tile_counter + 2, 1u);
        if (prev == gridDim.x - 1) {
            __threadfence();
            tile_counter[0] = 0; tile_counter[2] = 0;
            if (!cmode) {   // (a compact launch is always followed by its redo launch: the extra-record and overflow counters stay)
                *n_extra = atomicAdd(tile_counter + 3, 0u);
                tile_counter[1] = 0; tile_counter[3] = 0;
            }
        }
    }
    // Integer division by the (uniform, runtime) item strides: a generic `x / H` is a ~30-instruction sequence on the
    // VALU and sits in every per-item loop.  H is a power of two for 1, 2, 4, 8 ... primers (shift); otherwise one
    // v_mul_hi with ceil(2^32 / H), exact for x < 2^32 / H (items are < 2^16).
    __device__ __forceinline__ int divH(int x) const { return lH >= 0 ? (x >> lH) : (int)__umulhi((unsigned)x, Hmagic); }
    // tile -> reads.  Normal launch: tile t = reads [t R, t R + R).  Redo launch: the overflow list holds tiles of Rc reads
    // each; every one of them is cut into ceil(Rc / R) tiles of this launch.
    __device__ __forceinline__ void tile_span(uint32_t t, uint32_t &first, int &count) const {
        if (!redo) {
            first = t * (uint32_t)R;
            count = (int)((n_reads - first) < (uint32_t)R ? (n_reads - first) : (uint32_t)R);
        } else {
            const uint32_t seg = t / redo_ratio, sub = t - seg * redo_ratio;
            const uint32_t base = aux.ovf_list[seg] * (uint32_t)aux.Rc;
            const uint32_t end = base + (uint32_t)aux.Rc < n_reads ? base + (uint32_t)aux.Rc : n_reads;
            first = base + sub * (uint32_t)R;
            count = first < end ? (int)((end - first) < (uint32_t)R ? (end - first) : (uint32_t)R) : 0;
            if (count == 0) first = base;   // keep addresses formed from it in range
        }
    }
    // A phase boundary: optional timing (SMX_PHASE_TIMING=1: thread 0 accumulates s_memtime deltas per phase, in LDS so that
    // the accumulators cost no registers in the production path), and tid is laundered so that per-thread values are
    // recomputed by the phase that needs them instead of being computed early and carried (spilled) across the
    // register-hungry scans
    __device__ __forceinline__ void stamp(int i) {
        asm volatile("" : "+v"(tid));
        if (timing) { unsigned long long _t = clock64(); tacc[i] += _t - tacc[10]; tacc[10] = _t; }
    }

    // Everything that does not change during the launch.  Returns false when this workgroup has nothing to do at all (the
    // usual redo launch: an empty overflow list).
    __device__ __forceinline__ bool setup(unsigned char *lds_, const DevPanel *P_, const uint8_t *windows_, const int32_t *lens_,
                                          uint32_t n_reads_, int R_arg, smx_op *ops_, smx_op *extra_, uint32_t extra_cap_,
                                          uint32_t *n_extra_, unsigned long long *counts_, smx_hit *dbg_hits_, int8_t *dbg_bdist_arg,
                                          unsigned *tile_counter_, int use_slots_arg, const unsigned *pre_, const DemuxAux &aux_) {
        lds = lds_; P = P_; windows = windows_; lens = lens_; n_reads = n_reads_; ops = ops_; extra = extra_; extra_cap = extra_cap_;
        n_extra = n_extra_; counts = counts_; dbg_hits = dbg_hits_; tile_counter = tile_counter_; pre = pre_; aux = aux_;
        // SP = 3: SP = 1 with search_len 160 and 32-read tiles (the wide-window stress shape of a many-primer panel)
        R = SP == 3 ? 32 : (SP == 2 ? SMX_SP2_R : (sp ? 64 : R_arg));   // (the specialised kernels are only launched with these tile sizes)
        // SP = 2: SP = 1 for a panel with two primers (one forward, one reverse: a single amplicon) -- the tile's 256
        // alignments are one per lane, every LDS offset in front of the panel tables is a constant
        NP = SP == 2 ? 2 : P->NP; NB = P->NB; S = SP == 3 ? 160 : (sp ? 80 : P->S); H = 2 * NP; MW = (S + 31) / 32; maxB = P->maxB;
        need_starts = sp ? 0 : P->need_starts;
        n_pbc = P->n_pbc; NPAIR = SP == 2 ? 1 : P->NPAIR;
        npmeta = 6 * NP + 1 + n_pbc + NB + 3 * NPAIR;
        // (the default-flags kernels are lean bit-sliced launches by construction: no slots-mode state in them)
        use_slots = SP != 0 ? 0 : use_slots_arg;
        dbg_bdist = SP != 0 ? nullptr : dbg_bdist_arg;
        use_bs = SP != 0 ? 1 : ((BSV != 0 && P->bs_ok && !use_slots) ? 1 : 0);
        ncand = 2 * NPAIR;
        const int tails = sp ? 0 : (P->trim == SMX_TRIM_TAILS ? 1 : 0);
        T = make_layout<PW>(NP, NB, S, R, maxB, need_starts, npmeta, sp ? 3 : P->kidx, use_slots, P->bs_ok, ncand,
                            sp ? 0 : P->cap_hits, sp ? 0 : P->cap_ents, (sp && CM == 1) ? 256 : aux.nitems, tails, P->n_bstab);
        ppeq = (PW *)(lds + T.ppeq);        // [code][primer], stride NPs
        prpeq = (PW *)(lds + T.prpeq);
        bpeq = (unsigned *)(lds + T.bpeq);   // [code][barcode], stride NBs
        bsre = (unsigned *)(lds + T.bsre);   // bit-sliced table [primer][row][code][word] (lean mode)
        lut = lds + T.lut;
        codes = lds + T.codes;
        namask = (unsigned *)(lds + T.namask);
        lensL = (int *)(lds + T.lens);
        ocnt = (int *)(lds + T.ocnt);
        hits = (HitL *)(lds + T.hits);
        masks = (unsigned *)(lds + T.masks);
        if (layout_tie_in_rec(T.MBW, use_slots, tails)) { tieb = (unsigned *)(lds + T.hits); tstr = (int)(sizeof(HitL) / 4); }   // HitL::tail_end is word 0
        else { tieb = (unsigned *)(lds + T.tiem); tstr = sp ? 1 : T.MBW; }
        bres = (unsigned *)(lds + T.bres);
        dmask = (unsigned *)(lds + T.dmask);   // lean mode: [hit in round][distance][MBW] barcode bitmasks
        ents = (EntL *)(lds + T.ents);
        etail = (int *)(lds + T.etail);     // BSV == 3 only
        offsA = (unsigned short *)(lds + T.offsA);    // exclusive scan of searched locations per hit
        offsB = (unsigned short *)(lds + T.offsB);    // exclusive scan of searched hits (rank)
        queue = (unsigned short *)(lds + T.queue);   // rank -> hit
        cumL = (int *)(lds + T.emit);      // scorer: [read][candidate] accumulated trim shift
        opsL = (smx_op *)(lds + T.opsL);
        aggr = (int *)(lds + T.aggr);      // [0..7] counters, [8] round end rank, [9] next tile, [10] fallback items, [11] flagged alignments
        hmap = (unsigned short *)(lds + T.hmap);     // compact mode: read * H + alignment -> record (T.NI = "no match")
        clist = (unsigned short *)(lds + T.clist);   // compact mode: record -> read * H + alignment
        pmask = (unsigned long long *)(lds + T.pmask);   // per read: alignments with a primer match
        hcand = (unsigned long long *)(lds + T.hcand);   // per alignment: its candidates
        cfilt = layout_cfilt(H, NPAIR * 2);   // the scorer looks at matched alignments' candidates only
        tid = threadIdx.x;
        wave = tid >> 6;
        if (redo && tile_counter[1] == 0) {   // the usual redo launch: nothing on the list (every workgroup sees the same count:
            if (tid == 0) workgroup_done();       // it is only zeroed once all of them have passed this point)
            return false;
        }
        CS = T.CS; NPs = T.NPs; NBs = T.NBs; lNPs = T.lNPs; lNBs = T.lNBs; G = T.G; logG = T.logG; MBW = sp ? 1 : T.MBW;
        stride = SP == 3 ? 320 : (sp ? 160 : P->wstride);
        kidx = sp ? 3 : P->kidx; pfmin = P->pfmin;
        preorient = sp ? 1 : P->preorient; minlen = sp ? -1 : P->minlen; maxlen = sp ? -1 : P->maxlen;
        lH = (H & (H - 1)) == 0 ? 31 - __clz(H) : -1;
        Hmagic = (unsigned)((0x100000000ull + (unsigned)H - 1) / (unsigned)H);
        hcmagic = (unsigned)((0x100000000ull + (unsigned)(S >> 4) - 1) / (unsigned)((S >> 4) > 0 ? (S >> 4) : 1));   // chunk / (S/16)
        redo_ratio = redo ? (uint32_t)((aux.Rc + R - 1) / R) : 1u;
        n_tiles = redo ? tile_counter[1] * redo_ratio : (n_reads + R - 1) / R;
        tacc = (unsigned long long *)(lds + T.tacc);   // [0..9] sums, [10] previous stamp
        // (the default-flags kernels carry no phase timing: SMX_PHASE_TIMING runs use the generic instantiation)
        dbg_phase = sp ? nullptr : P->dbg_phase;
        timing = dbg_phase != nullptr && tid == 0;
        SW = (R + 63) >> 6;              // scorer waves
        return true;
    }

    __device__ __forceinline__ void stage_panel() {
        // ---- phase 0: stage the panel (transposed: consecutive lanes = consecutive patterns hit distinct banks)
        for (int i = tid; i < NP * 16; i += NT) {
            int p = i >> 4, c = i & 15;
            ppeq[c * NPs + p] = (PW)P->ppeq[i] << (PWBITS - P->pm[p]);   // left-aligned: row m-1 is the top bit
            if (need_starts) prpeq[c * NPs + p] = (PW)P->prpeq[i];
        }
        if (!use_bs)
            for (int i = tid; i < NB * 16; i += NT) bpeq[(i & 15) * NBs + (i >> 4)] = P->bpeq[i];
        if (use_bs)
            for (int i = tid; i < P->n_bstab * T.MBW * 256; i += NT) {
                int blk = i >> 8;   // (primer, word) block: [row][code], the row stride is a compile-time 16 words
                bsre[blk * T.BSP + (i & 255)] = P->bs_re[i];
            }
        for (int i = tid; i < 512; i += NT) lut[i] = P->lut[i];
        int *pmeta = (int *)(lds + T.pmeta);
        {
            int *q = pmeta;
            LP.pm = q; q += NP; LP.pk = q; q += NP; LP.pdir = q; q += NP; LP.pfidx = q; q += NP;
            LP.pbc_off = q; q += NP + 1; LP.pbc = q; q += n_pbc; LP.bm = q; q += NB;
            LP.pair_f = q; q += NPAIR; LP.pair_r = q; q += NPAIR; LP.pair_pool = q; q += NPAIR;
            LP.bstab = q;
            for (int i = tid; i < NP; i += NT) q[i] = P->bs_tab[i];
            for (int i = tid; i < NP; i += NT) {
                pmeta[i] = P->pm[i]; pmeta[NP + i] = P->pk[i]; pmeta[2 * NP + i] = P->pdir[i]; pmeta[3 * NP + i] = P->pfidx[i];
            }
            for (int i = tid; i <= NP; i += NT) pmeta[4 * NP + i] = P->pbc_off[i];
            for (int i = tid; i < n_pbc; i += NT) pmeta[5 * NP + 1 + i] = P->pbc[i];
            for (int i = tid; i < NB; i += NT) pmeta[5 * NP + 1 + n_pbc + i] = P->bm[i];
            for (int i = tid; i < NPAIR; i += NT) {
                int *b = pmeta + 5 * NP + 1 + n_pbc + NB;
                b[i] = P->pair_f[i]; b[NPAIR + i] = P->pair_r[i]; b[2 * NPAIR + i] = P->pair_pool[i];
            }
        }
        if (tid < 12) aggr[tid] = 0;
        if (cfilt) for (int i = tid; i < R; i += NT) pmask[i] = 0ull;
        if (cfilt) for (int i = tid; i < H; i += NT) hcand[i] = 0ull;
        if (cmode && tid == 0) {   // the shared record of every alignment that is not flagged: what the primer scan writes for "no match"
            HitL hn;
            hn.tail_end = -1; hn.nloc = 0; hn.ntied = 0; hn.first_tied = -1; hn.pdist = -1; hn.bbest = -2; hn.jstar = 0; hn.fs_j = 0;
            hn.flags = 0; hn.pad = 0;
            hits[T.NI] = hn;
            for (int w = 0; w < T.MBW; w++) tieb[T.NI * tstr + w] = 0;
        }
        __syncthreads();
        if (cfilt)
            for (int ci = tid; ci < NPAIR * 2; ci += NT) {   // candidate ci = (pair, orientation): its two alignments (cand_view)
                const int pair = ci >> 1, o = ci & 1;
                const int h1 = LP.pair_f[pair] * 2 + (o == 0 ? 0 : 1), h2 = LP.pair_r[pair] * 2 + (o == 0 ? 1 : 0);
                atomicOr(&hcand[h1], 1ull << ci);
                atomicOr(&hcand[h2], 1ull << ci);
            }
        __syncthreads();

        if (timing) for (int i = 0; i < 11; i++) tacc[i] = 0;
    }

    // ---- phase 2, one alignment: primer scan, one lane per (read, primer, end).  With the prescan (pre != nullptr) every
    // alignment it covers is decoded from its flag words; the others (windows with anything but upper-case ACGT) are queued
    // and scanned afterwards, packed into the lowest lanes, so that one such read does not make its whole wave run
    // the 80-column scalar scan.
    __device__ __forceinline__ void primer_item(int item, const bool use_pre) {   // item = record index
        const int dn = cmode ? (int)clist[item] : item;
        int r = divH(dn), h = dn - __mul24(r, H), p = h >> 1, X = h & 1;
        int L = lensC[r];
        EndGeom g = end_geom(L, S);
        const unsigned char *cw = codes + __mul24(r * 2 + X, CS);
        const PW *peq = ppeq + p;
        const int m = LP.pm[p], k = LP.pk[p], top = PWBITS - 1;
        PW Pvv = ~(PW)0, Mv = 0;
        int score = m, best = m + 1, jstar = 0, cnt = 0;
        unsigned *mrow = masks + __mul24(item, MW);
        // With the prescan: its flag words cover the window's first Sp columns = the whole stored window, which is the
        // primer target unless the read is shorter than search_len AND its target starts inside the window
        // (g.j_lo > 0): then only the orientation vote comes from the prescan (determine_orientation looks
        // at the whole string) and the target [j_lo, Sp) -- at most search_len / 2 columns -- is scanned below.
        bool pre_done = false, pre_omatch = false;
        if (use_pre) {
            const uint32_t gr = r0 + (uint32_t)r;   // flag words: [tile of 1024 reads][alignment h][chunk][read in tile]
            const unsigned *pw = pre + ((size_t)(gr >> 10) * H + h) * (S >> 4) * PRE_TILE + (gr & (PRE_TILE - 1));
            const int bfull = S == 160 ? prescan_decode<10>(pw, PRE_TILE, 10, MW, m, k, g.Sp, mrow, &jstar, &cnt)
                            : S == 80 ? prescan_decode<5>(pw, PRE_TILE, 5, MW, m, k, g.Sp, mrow, &jstar, &cnt)   // the default -l
                                      : prescan_decode<0>(pw, PRE_TILE, S >> 4, MW, m, k, g.Sp, mrow, &jstar, &cnt);
            pre_omatch = bfull <= k;
            if (g.j_lo == 0) { pre_done = true; best = pre_omatch ? bfull : m + 1; }
            else { jstar = 0; cnt = 0; }
        }
        if (pre_done) {
        } else if (sizeof(PW) == 4 && g.j_lo == 0 && g.Sp == S && (S & 3) == 0) {
            // common case (full window, 32-bit patterns): eight columns per unrolled block, one byte read per
            // column (cheaper than unpacking a dword of codes: tools/ubench/primer_col.hip), uniform trip counts;
            // the "new minimum" / "above minimum" flags are funnel-shifted into bit-reversed words
            const unsigned *pq = (const unsigned *)ppeq + p;
            unsigned Pu = ~0u, Mu = 0;
            int sc = m, bst = m + 1;
            const int esh = lNPs + 2;   // byte offset of Eq[code] = code << esh: one v_lshl_add per column
#define SMX_PCOL(J) do { const unsigned e_ = *(const unsigned *)((const char *)pq + ((unsigned)cw[J] << esh)); \
                     myers_step_hw_top(e_, Pu, Mu, sc);                                              \
                     ltw = __builtin_amdgcn_alignbit(ltw, (unsigned)(sc - bst), 31);                  \
                     gtw = __builtin_amdgcn_alignbit(gtw, (unsigned)(bst - sc), 31);                  \
                     bst = sc < bst ? sc : bst; } while (0)
            for (int w = 0; w < MW; w++) {
                const int ncols = S - w * 32 < 32 ? S - w * 32 : 32;   // uniform, a multiple of 4
                // ltw: "new minimum" flags, gtw: "above the minimum" flags; the newest column is bit 0
                unsigned gtw = 0, ltw = 0;
                int j = w * 32;
                const int jend = j + ncols;
                for (; j + 8 <= jend; j += 8) {
                    SMX_PCOL(j); SMX_PCOL(j + 1); SMX_PCOL(j + 2); SMX_PCOL(j + 3);
                    SMX_PCOL(j + 4); SMX_PCOL(j + 5); SMX_PCOL(j + 6); SMX_PCOL(j + 7);
                }
                if (j < jend) { SMX_PCOL(j); SMX_PCOL(j + 1); SMX_PCOL(j + 2); SMX_PCOL(j + 3); }
                // column c of this word sits at bit ncols-1-c
                if (ncols > 0) {
                    mrow[w] = __brev(~gtw) >> (32 - ncols);
                    if (ltw) jstar = w * 32 + ncols - __ffs(ltw);
                } else mrow[w] = 0;
            }
#undef SMX_PCOL
            best = bst; score = sc;
            Pvv = (PW)Pu; Mv = (PW)Mu;
            for (int w = jstar >> 5; w < MW; w++) {
                unsigned word = mrow[w];
                if (w == (jstar >> 5)) word &= ~0u << (jstar & 31);
                cnt += __popc(word);
            }
        } else if (g.j_lo == 0) {
            // common case: the target is the whole stored window; branch-free bookkeeping per column
            const int Sp = g.Sp;
            for (int w = 0; w < MW; w++) {
                unsigned word = 0;
                int jend = (w + 1) * 32 < Sp ? (w + 1) * 32 : Sp;
#pragma unroll 4
                for (int j = w * 32; j < jend; j++) {
                    myers_step<PW, false>(peq[(int)cw[j] << lNPs], Pvv, Mv, score, top);
                    bool lt = score < best;
                    best = lt ? score : best;
                    jstar = lt ? j : jstar;
                    word |= (score == best ? 1u : 0u) << (j & 31);
                }
                mrow[w] = word;
            }
            // number of optimal ends = bits at or after the first occurrence of the final minimum
            for (int w = jstar >> 5; w < MW; w++) {
                unsigned word = mrow[w];
                if (w == (jstar >> 5)) word &= ~0u << (jstar & 31);
                cnt += __popc(word);
            }
        } else {
            for (int w = 0; w < MW; w++) mrow[w] = 0;
            for (int j = g.j_lo; j < g.Sp; j++) {
                myers_step<PW, false>(peq[(int)cw[j] << lNPs], Pvv, Mv, score, top);
                if (score < best) { best = score; jstar = j; cnt = 0; }
                if (score == best) { cnt++; mrow[j >> 5] |= 1u << (j & 31); }   // bits before jstar are stale: ignored
            }
        }
        bool matched = best <= k;
        bool omatch = matched;
        if (g.j_lo > 0 && use_pre) omatch = pre_omatch;
        else if (g.j_lo > 0) {   // short read: determine_orientation looks at the whole string (Q1)
            PW P2 = ~(PW)0, M2 = 0;
            int sc = m, b2 = m + 1;
            for (int j = 0; j < g.Sp; j++) {
                myers_step<PW, false>(peq[(int)cw[j] << lNPs], P2, M2, sc, top);
                b2 = sc < b2 ? sc : b2;
            }
            omatch = b2 <= k;
        }
        int fs_j = jstar;
        if (matched && need_starts) {
            // edlib's start rule: SHW of the reversed pattern over the reversed target prefix,
            // LAST optimal position = smallest start
            const PW *rpeq = prpeq + p;
            PW P2 = ~(PW)0, M2 = 0;
            int sc = m, lastc = 1;
            int maxc = m + best;
            for (int c = 1; c <= maxc; c++) {
                int j = jstar - (c - 1);
                if (j < g.j_lo) break;
                myers_step<PW, true>(rpeq[(int)cw[j] << lNPs], P2, M2, sc, m - 1);   // rpeq is right-aligned
                if (sc == best) lastc = c;
            }
            fs_j = jstar - (lastc - 1);
        }
        HitL hl;
        hl.tail_end = -1;
        hl.pdist = (signed char)(matched ? best : -1);
        hl.nloc = (short)(matched ? cnt : 0);
        hl.bbest = -2; hl.ntied = 0; hl.first_tied = -1;
        hl.jstar = (unsigned char)jstar; hl.fs_j = (unsigned char)fs_j;
        hl.flags = (unsigned char)(omatch ? 1 : 0); hl.pad = 0;
        hits[item] = hl;
        if (cfilt && matched) atomicOr(&pmask[r], 1ull << h);
        if (omatch) {
            // determine_orientation via A.6: fwd primer in A / rev primer in B vote "forward"
            int dir = LP.pdir[p];
            int vote_fwd = (dir == 0) ? (X == 0) : (X == 1);
            atomicAdd(&ocntC[r], vote_fwd ? 1 : 0x10000);
        }
    }

    __device__ __forceinline__ void phase2_primers() {
        unsigned short *fbq = queue;   // fallback items (the rank -> hit queue is not live before phase 3a)
        if (cmode) {
            // which alignments get a record: the ones the prescan flags, and every alignment of a read the prescan
            // does not cover (rflag).  All match-word loads of a batch are issued before the first is used.
            for (int i0 = tid; i0 < nh; i0 += 4 * NT) {
                unsigned mw[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = i0 + u * NT < nh ? i0 + u * NT : nh - 1;
                    const int r = divH(i), h = i - __mul24(r, H);
                    const uint32_t gr = r0 + (uint32_t)r;
                    mw[u] = aux.match[((size_t)(gr >> 10) * H + h) * PRE_G + ((gr & (PRE_TILE - 1)) >> 5)] >> (gr & 31);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = i0 + u * NT;
                    if (i < nh) {
                        const bool flagged = (mw[u] & 1u) || (ocntC[divH(i)] & OCNT_NA) != 0;
                        int rec = T.NI;
                        if (flagged) {
                            const int sl = atomicAdd(&aggr[11], 1);
                            if (sl < T.NI) { clist[sl] = (unsigned short)i; rec = sl; }
                        }
                        hmap[i] = (unsigned short)rec;
                    }
                }
            }
            __syncthreads();
            nI = aggr[11];
            if (nI > T.NI) {   // does not fit: the redo launch takes this tile's reads
                live = false;
                if (tid == 0) aux.ovf_list[atomicAdd(tile_counter + 1, 1u)] = cur;
            }
        }
        if (!live) {
        } else if (pre != nullptr) {
            for (int item = tid; item < nI; item += NT) {
                if ((ocntC[divH(cmode ? (int)clist[item] : item)] & OCNT_NA) == 0) primer_item(item, true);
                else fbq[atomicAdd(&aggr[10], 1)] = (unsigned short)item;
            }
            __syncthreads();
            const int nfb = aggr[10];
            if (nfb > 0) {   // uniform; usually nothing was queued and the tile goes straight on
                for (int i = tid; i < nfb; i += NT) primer_item((int)fbq[i], false);
                __syncthreads();
                if (tid == 0) aggr[10] = 0;   // read by everyone before the barrier above; next written in the next tile's phase 2
            }
        } else {
            for (int item = tid; item < nI; item += NT) primer_item(item, false);
            __syncthreads();
        }

    }

    // ---- phase 3a helpers: which ends need barcodes (find_candidate_matches:677-741) -> number of locations to search
    __device__ __forceinline__ int locations_needed(int item) {
        const int dn = cmode ? (int)clist[item] : item;
        int r = divH(dn), h = dn - __mul24(r, H), p = h >> 1, X = h & 1;
        int L = lensC[r];
        int f = ocntC[r] & 0x7FFF, rv = ocntC[r] >> 16;
        int ori = 3;   // bit0: as-read candidates allowed, bit1: reverse-complement candidates allowed
        if (preorient) { if (f > 0 && rv == 0) ori = 1; else if (rv > 0 && f == 0) ori = 2; }
        bool filtered = (minlen != -1 && L < minlen) || (maxlen != -1 && L > maxlen);
        int dir = LP.pdir[p];
        bool in_fwd = (dir == 0) ? (X == 0) : (X == 1);   // used by as-read candidates
        bool needed = !filtered && ((in_fwd && (ori & 1)) || (!in_fwd && (ori & 2)));
        HitL &hl = hits[item];
        int n = 0;
        if (needed && hl.pdist >= 0) {
            hl.flags |= 2;
            hl.bbest = -1;
            n = hl.nloc;
        }
        for (int w = 0; w < MBW; w++) tieb[__mul24(item, tstr) + w] = 0;
        return n;
    }
    // the optimal locations of one searched hit -> entries e, e+1, ... (target start, prefilter verdict)
    __device__ __forceinline__ void list_entries(int item, int rank_in_round, int e) {
        const int dn = cmode ? (int)clist[item] : item;
        int r = divH(dn), h = dn - __mul24(r, H), X = h & 1;
        const HitL &hl = hits[item];
        int L = lensC[r];
        EndGeom g = end_geom(L, S);
        const unsigned *mrow = masks + __mul24(item, MW);
        const unsigned *na = namask + __mul24(r * 2 + X, MW);
        int ord = 0;
        for (int w = hl.jstar >> 5; w < MW; w++) {
            unsigned word = mrow[w];
            if (w == (hl.jstar >> 5)) word &= ~0u << (hl.jstar & 31);
            while (word) {
                int je = w * 32 + __ffs(word) - 1;
                word &= word - 1;
                int bstart = (je - g.j_lo) + g.shift + 1;
                BcGeom bg = bc_geom(L, g.base, bstart);
                int ncol = g.Sp - bg.tj0;
                bool ok = ncol > 0;
                if (ok && pfmin > 0) {   // exact-set restatement of BloomPrefilter.match (Q7)
                    if (!bg.pf_same || ncol < pfmin) ok = false;
                    else {   // any non-ACGT code among target[0 : pfmin) ?
                        int wi = bg.tj0 >> 5, sh = bg.tj0 & 31;
                        unsigned long long two = ((unsigned long long)(wi + 1 < MW ? na[wi + 1] : 0u) << 32) | na[wi];
                        if ((two >> sh) & ((1ull << pfmin) - 1ull)) ok = false;
                    }
                }
                EntL en;
                en.hit = (unsigned short)item;
                en.slot = (unsigned short)(rank_in_round << logG);
                en.tj0 = (unsigned char)(ok ? bg.tj0 : 0);
                en.loc_ord = (unsigned char)ord;
                en.ncol = (unsigned char)(ncol > 255 ? 255 : (ncol < 0 ? 0 : ncol));
                en.ok = ok ? 1 : 0;
                en.delta = (short)bg.delta; en.pad = 0;
                ents[e++] = en;
                ord++;
            }
        }
    }

    // ---- phase 3a: orientation, which ends need barcodes; block-wide scans of locations and searched hits;
    //      one *entry* per optimal primer location of every searched hit
    __device__ __forceinline__ void phase3a_entries() {
        if (dbg_bdist)   // (slots mode only: never compact)
            for (int i = tid; i < nh * maxB; i += NT) dbg_bdist[(size_t)r0 * H * maxB + i] = -1;
        prelisted = false;   // the usual case: one item per lane and everything fits one round ->
        nq = 0; nE_pre = 0;  // scans by shuffles, entries listed by the hit's own lane, three barriers fewer
        if (nI <= NT) {
            const int item = tid;
            const int n = item < nI ? locations_needed(item) : 0;
            const unsigned pk = (unsigned)n | ((n > 0 ? 1u : 0u) << 16);   // locations | searched hits << 16
            unsigned incl = pk;
            for (int d = 1; d < 64; d <<= 1) {
                unsigned v = (unsigned)__shfl_up((int)incl, d, 64);
                if ((tid & 63) >= d) incl += v;
            }
            unsigned *wsum = (unsigned *)offsB;
            if ((tid & 63) == 63) wsum[wave] = incl;
            __syncthreads();
            unsigned base = 0, total = 0;
            for (int w = 0; w < NT / 64; w++) { unsigned v = wsum[w]; total += v; base += w < wave ? v : 0u; }
            const unsigned excl = base + incl - pk;
            nq = (int)(total >> 16);
            const int totE = (int)(total & 0xFFFFu);
            if (nq <= T.CAPH && totE <= T.CAPE) {
                prelisted = true;
                nE_pre = totE;
                if (n > 0) {
                    queue[excl >> 16] = (unsigned short)item;
                    if (BSV == 3) offsA[item] = (unsigned short)(excl & 0xFFFFu);   // first entry of the hit (tails pass)
                    list_entries(item, (int)(excl >> 16), (int)(excl & 0xFFFFu));
                }
                if (use_slots) { for (int i = tid; i < (nq << logG); i += NT) bres[i] = 0xFFFFFFFFu; }
                else { for (int i = tid; i < nq * (kidx + 1) * MBW; i += NT) dmask[i] = 0; }
                __syncthreads();
            } else {   // several rounds: hand the scans over to the general code below
                __syncthreads();   // wsum (in offsB) has been read by everyone
                if (item < nI) {
                    offsA[item] = (unsigned short)(excl & 0xFFFFu);
                    offsB[item] = (unsigned short)(excl >> 16);
                    if (n > 0) queue[excl >> 16] = (unsigned short)item;
                }
                if (tid == 0) { offsA[nI] = (unsigned short)totE; offsB[nI] = (unsigned short)nq; }
                __syncthreads();
            }
        } else {
            for (int item = tid; item < nI; item += NT) {
                int n = locations_needed(item);
                offsA[item] = (unsigned short)n;
                offsB[item] = n > 0 ? 1 : 0;
            }
            __syncthreads();
            if (wave == 0) wave_exclusive_scan(offsA, nI);
            else if (wave == 1) wave_exclusive_scan(offsB, nI);
            __syncthreads();
            for (int item = tid; item < nI; item += NT)
                if (offsB[item + 1] != offsB[item]) queue[offsB[item]] = (unsigned short)item;
            nq = offsB[nI];
            __syncthreads();
        }

    }

    // 3b: barcode scan of the round's nE entries
    __device__ __forceinline__ void phase3b_barcodes(int nE) {
        // 3b (lean, uniform barcode length): bit-sliced scan, one lane per (entry, 32-barcode word).  Two separate
        // loops (k is uniform for the launch) so that the register allocation of one variant never meets the other's.
        if (BSV != 0 && use_bs) {
            const int bsm = P->bs_m;
            for (int item = tid; item < nE * MBW; item += NT) {
                int ei = item, w = 0;
                if (MBW != 1) { ei = item / MBW; w = item - ei * MBW; }   // MBW == 1 (<= 32 barcodes per primer) is the usual case
                const EntL en = ents[ei];
                if (BSV == 3) etail[ei] = -0x7FFFFFFF;
                if (!en.ok) continue;
                const int hh = cmode ? (int)clist[en.hit] : (int)en.hit, r = divH(hh), h = hh - __mul24(r, H), p = h >> 1, X = h & 1;
                const unsigned char *cwt = codes + __mul24(r * 2 + X, CS) + en.tj0;
                unsigned *dm = dmask + __mul24(__mul24(en.slot >> logG, kidx + 1), MBW) + w;
                if constexpr (BSV == 3) {
                    unsigned seen[4], ML[4];
                    const unsigned want[4] = {~0u, ~0u, ~0u, ~0u};
                    int tc = -1;
                    const unsigned *reb = bsre + __mul24(__mul24(LP.bstab[p], MBW) + w, T.BSP);
                    if (bsm == 13) bitsliced_shw_pad_tails<3, 13>(reb, cwt, en.ncol, bsm, kidx, want, seen, ML, tc);
                    else if (bsm <= 8 && bsm > 3) bitsliced_shw_pad_tails<3, 8>(reb, cwt, en.ncol, bsm, kidx, want, seen, ML, tc);
                    else if (bsm <= 12 && bsm > 3) bitsliced_shw_pad_tails<3, 12>(reb, cwt, en.ncol, bsm, kidx, want, seen, ML, tc);
                    else bitsliced_shw_pad_tails<3, 16>(reb, cwt, en.ncol, bsm, kidx, want, seen, ML, tc);
                    // end of the kept alignment in reference coordinates, valid as the hit's tail if this is its only
                    // location (the summary redoes hits with several locations)
                    etail[ei] = tc >= 0 ? (int)en.tj0 + tc + end_geom(lensC[r], S).base - (int)en.delta : -0x7FFFFFFF;
#pragma unroll
                    for (int d = 0; d < 4; d++)
                        if (d <= kidx && seen[d]) atomicOr(&dm[d * MBW], seen[d]);
                } else if constexpr (BSV == 1) {
                    unsigned seen[4];
                    // padded height: the smallest instantiated M >= barcode length (uniform branch)
                    const unsigned *reb = bsre + __mul24(__mul24(LP.bstab[p], MBW) + w, T.BSP);
                    if (bsm == 13) bitsliced_shw_pad<3, 13>(reb, cwt, en.ncol, bsm, kidx, seen);
                    else if (bsm <= 8 && bsm > 3) bitsliced_shw_pad<3, 8>(reb, cwt, en.ncol, bsm, kidx, seen);
                    else if (bsm <= 12 && bsm > 3) bitsliced_shw_pad<3, 12>(reb, cwt, en.ncol, bsm, kidx, seen);
                    else bitsliced_shw_pad<3, 16>(reb, cwt, en.ncol, bsm, kidx, seen);
#pragma unroll
                    for (int d = 0; d < 4; d++)
                        if (d <= kidx && seen[d]) atomicOr(&dm[d * MBW], seen[d]);
                } else if constexpr (BSV == 2) {
                    unsigned seen[8];
                    const unsigned *reb8 = bsre + __mul24(__mul24(LP.bstab[p], MBW) + w, T.BSP);
                    if (kidx == 4 && bsm > 4 && bsm <= 16) {   // k = 4: the padded straight-line scan with a 9-row window
                        unsigned s5[5];
                        if (bsm == 13) bitsliced_shw_pad<4, 13>(reb8, cwt, en.ncol, bsm, kidx, s5);
                        else bitsliced_shw_pad<4, 16>(reb8, cwt, en.ncol, bsm, kidx, s5);
#pragma unroll
                        for (int d = 0; d < 8; d++) seen[d] = d < 5 ? s5[d] : 0u;
                    } else
                        bitsliced_shw<8>(reb8, cwt, en.ncol, bsm, kidx, seen);
#pragma unroll
                    for (int d = 0; d < 8; d++)
                        if (d <= kidx && seen[d]) atomicOr(&dm[d * MBW], seen[d]);
                }
            }
        } else
        // 3b: barcode scan, one lane per (entry, barcode slot)
        for (int item = tid; item < (nE << logG); item += NT) {
            const EntL en = ents[item >> logG];
            int bi = item & (G - 1);
            int hh = cmode ? (int)clist[en.hit] : (int)en.hit, r = divH(hh), h = hh - __mul24(r, H), p = h >> 1, X = h & 1;
            int nb = LP.pbc_off[p + 1] - LP.pbc_off[p];
            if (bi >= nb || !en.ok) continue;
            int gb = LP.pbc[LP.pbc_off[p] + bi];
            int m = LP.bm[gb], top = m - 1;
            const unsigned char *cw = codes + (r * 2 + X) * CS + en.tj0;
            int ncol = en.ncol < m + kidx ? en.ncol : m + kidx;
            const unsigned *peq = bpeq + gb;
            unsigned Pvv = ~0u, Mv = 0;
            int score = m, best = m + 1, firstc = 0, lastc = 0;
#pragma unroll 4
            for (int c = 0; c < ncol; c++) {
                myers_step<unsigned, true>(peq[(int)cw[c] << lNBs], Pvv, Mv, score, top);
                bool lt = score < best;
                best = lt ? score : best;
                firstc = lt ? c : firstc;
                lastc = (score == best) ? c : lastc;
            }
            if (best <= kidx) {
                if (use_slots)
                    atomicMin(&bres[en.slot + bi], ((unsigned)best << 24) | ((unsigned)en.loc_ord << 16) |
                                                       ((unsigned)(en.tj0 + firstc) << 8) | (unsigned)(en.tj0 + lastc));
                else   // barcode bi seen at distance `best` (any location): the lowest non-empty level wins
                    atomicOr(&dmask[(((en.slot >> logG) * (kidx + 1)) + best) * MBW + (bi >> 5)], 1u << (bi & 31));
            }
        }

    }

    __device__ __forceinline__ void phase3c_summary(int q0, int q1, int e_base) {
        // 3c: per searched hit: best distance, tie set (ballot), tails extent.  Lanes = (hit, barcode slot):
        // group-of-G reductions by xor-shuffles, the tie bitmask straight from the ballot.
        if (!use_slots) {
            // lean mode: best = lowest distance level with any barcode, tie set = that level's bitmask
            for (int q = q0 + tid; q < q1; q += NT) {
                int item = queue[q];
                const unsigned *dm = dmask + (q - q0) * (kidx + 1) * MBW;
                for (int d = 0; d <= kidx; d++) {
                    int nt = 0, first = -1;
                    for (int w = 0; w < MBW; w++) {
                        unsigned m = dm[d * MBW + w];
                        if (m && first < 0) first = w * 32 + __ffs(m) - 1;
                        nt += __popc(m);
                    }
                    if (nt) {
                        HitL &hl = hits[item];
                        hl.bbest = (signed char)d; hl.ntied = (short)nt; hl.first_tied = (short)first;
                        for (int w = 0; w < MBW; w++) tieb[item * tstr + w] = dm[d * MBW + w];
                        break;
                    }
                }
                if constexpr (BSV == 3) {
                    // --trim tails: max over the within-k barcodes of the last optimal end of the alignment the
                    // reference keeps for each (its FIRST location at its best distance, match_one_end :806-811)
                    HitL &hl = hits[item];
                    if (hl.bbest >= 0) {
                        const int ne = hl.nloc, e0 = offsA[item] - e_base;
                        int t = -0x7FFFFFFF;
                        if (ne == 1) t = etail[e0];
                        else {
                            const int dn = cmode ? (int)clist[item] : item;
                            const int r = divH(dn), h = dn - __mul24(r, H), p = h >> 1, X = h & 1;
                            const int base = end_geom(lensC[r], S).base, bsm = P->bs_m;
                            const unsigned *reb = bsre + __mul24(LP.bstab[p], T.BSP);   // MBW == 1 on this path
                            unsigned GM[4], prev[4] = {0u, 0u, 0u, 0u}, lower = 0;
#pragma unroll
                            for (int d = 0; d < 4; d++) { GM[d] = d <= kidx ? (dm[d] & ~lower) : 0u; lower |= d <= kidx ? dm[d] : 0u; }
                            for (int e = e0; e < e0 + ne; e++) {
                                const EntL en = ents[e];
                                if (!en.ok) continue;
                                unsigned want[4], seen[4], ML[4];
#pragma unroll
                                for (int d = 0; d < 4; d++) want[d] = GM[d] & ~prev[d];
                                if (!(want[0] | want[1] | want[2] | want[3])) break;   // every barcode has its location
                                const unsigned char *cwt = codes + __mul24(r * 2 + X, CS) + en.tj0;
                                int tc = -1;
                                if (bsm == 13) bitsliced_shw_pad_tails<3, 13>(reb, cwt, en.ncol, bsm, kidx, want, seen, ML, tc);
                                else if (bsm <= 8 && bsm > 3) bitsliced_shw_pad_tails<3, 8>(reb, cwt, en.ncol, bsm, kidx, want, seen, ML, tc);
                                else if (bsm <= 12 && bsm > 3) bitsliced_shw_pad_tails<3, 12>(reb, cwt, en.ncol, bsm, kidx, want, seen, ML, tc);
                                else bitsliced_shw_pad_tails<3, 16>(reb, cwt, en.ncol, bsm, kidx, want, seen, ML, tc);
                                if (tc >= 0) { int v = (int)en.tj0 + tc + base - (int)en.delta; t = v > t ? v : t; }
#pragma unroll
                                for (int d = 0; d < 4; d++) prev[d] |= ML[d] & GM[d];
                            }
                        }
                        hl.tail_end = t;
                    }
                }
            }
        } else if (G <= 64) {
            const int nslots = (q1 - q0) << logG;
            for (int base_i = wave * 64; base_i < nslots; base_i += NT) {
                const int i = base_i + (tid & 63);
                const bool in = i < nslots;
                const int q = q0 + ((in ? i : 0) >> logG), sl = i & (G - 1);
                const int item = queue[q];
                const int r = divH(item), h = item - __mul24(r, H), p = h >> 1;
                const int nb = LP.pbc_off[p + 1] - LP.pbc_off[p];
                const bool live = in && sl < nb;
                unsigned v = live ? bres[i] : 0xFFFFFFFFu;
                if (live && dbg_bdist)
                    dbg_bdist[(size_t)r0 * H * maxB + (size_t)item * maxB + sl] = (v == 0xFFFFFFFFu) ? (int8_t)-1 : (int8_t)(v >> 24);
                const int L = lensC[r];
                const EndGeom g = end_geom(L, S);
                int last_abs = -0x7FFFFFFF;
                if (v != 0xFFFFFFFFu) {
                    int delta = 0;
                    if (L < S) {   // short read: the location's slice start may have wrapped (Q1)
                        int je = nth_location(masks + (size_t)item * MW, MW, hits[item].jstar, (v >> 16) & 0xFF);
                        delta = bc_geom(L, g.base, (je - g.j_lo) + g.shift + 1).delta;
                    }
                    last_abs = (int)(v & 0xFF) + g.base - delta;
                }
                unsigned dmin = v >> 24;
                int tail = last_abs;
                for (int d = 1; d < G; d <<= 1) {
                    unsigned o = (unsigned)__shfl_xor((int)dmin, d, 64);
                    int t = __shfl_xor(tail, d, 64);
                    dmin = o < dmin ? o : dmin;
                    tail = t > tail ? t : tail;
                }
                const bool tied = (v >> 24) == dmin && dmin != 255;
                unsigned long long bal = __ballot(tied);
                const int gshift = (tid & 63) & ~(G - 1);
                unsigned long long gm = (G == 64) ? bal : ((bal >> gshift) & ((1ull << G) - 1ull));
                int first = gm ? __ffsll((long long)gm) - 1 : 0;
                if (in && sl == 0 && dmin != 255) {
                    HitL &hl = hits[item];
                    hl.bbest = (signed char)dmin; hl.ntied = (short)__popcll(gm); hl.first_tied = (short)first;
                    hl.tail_end = tail;
                    tieb[item * tstr] = (unsigned)gm;
                    if (MBW > 1) tieb[item * tstr + 1] = (unsigned)(gm >> 32);
                }
            }
        } else
        for (int q = q0 + tid; q < q1; q += NT) {
            int item = queue[q];
            HitL &hl = hits[item];
            int r = divH(item), h = item - __mul24(r, H), p = h >> 1;
            int nb = LP.pbc_off[p + 1] - LP.pbc_off[p];
            const unsigned *br = bres + ((q - q0) << logG);
            unsigned best = 255;
            for (int i = 0; i < nb; i++) { unsigned d = br[i] >> 24; best = d < best ? d : best; }
            if (dbg_bdist)
                for (int i = 0; i < nb; i++)
                    dbg_bdist[(size_t)r0 * H * maxB + (size_t)item * maxB + i] =
                        (br[i] == 0xFFFFFFFFu) ? (int8_t)-1 : (int8_t)(br[i] >> 24);
            if (best == 255) continue;
            int L = lensC[r];
            EndGeom g = end_geom(L, S);
            int ntied = 0, first = -1, tail = -0x7FFFFFFF;
            for (int i = 0; i < nb; i++) {
                unsigned v = br[i];
                if (v == 0xFFFFFFFFu) continue;
                int delta = 0;
                if (L < S) {   // short read: the location's slice start may have wrapped (Q1)
                    int je = nth_location(masks + (size_t)item * MW, MW, hl.jstar, (v >> 16) & 0xFF);
                    delta = bc_geom(L, g.base, (je - g.j_lo) + g.shift + 1).delta;
                }
                int last_abs = (int)(v & 0xFF) + g.base - delta;
                tail = last_abs > tail ? last_abs : tail;
                if ((v >> 24) == best) {
                    if (ntied == 0) first = i;
                    ntied++;
                    tieb[item * tstr + (i >> 5)] |= 1u << (i & 31);
                }
            }
            hl.bbest = (signed char)best; hl.ntied = (short)ntied; hl.first_tied = (short)first;
            hl.tail_end = tail;
        }

    }

    // ---- phase 3b/3c in rounds of at most CAPH searched hits and CAPE (hit, location) entries
    __device__ __forceinline__ void phase3_rounds() {
        for (int q0 = 0; q0 < nq;) {
            int q1, nE, e_base = 0;
            if (prelisted) { q1 = nq; nE = nE_pre; }
            else {
                e_base = offsA[queue[q0]];
                if (nq - q0 <= T.CAPH && offsA[nI] - e_base <= T.CAPE) q1 = nq;   // everything left fits
                else {
                    // the round takes hits q0 .. q1 - 1: as many as fit both capacities.  "Hits q0 .. q - 1 fit" is monotone in
                    // q, so every lane tests its own q and the largest one that passes wins.
                    if (tid == 0) aggr[8] = q0 + 1;   // one hit always fits: CAPE >= 256 >= its locations
                    __syncthreads();
                    for (int q = q0 + 2 + tid; q <= nq && q - q0 <= T.CAPH; q += NT) {
                        const int hq = queue[q - 1];
                        if (offsA[hq] + hits[hq].nloc - e_base <= T.CAPE) atomicMax(&aggr[8], q);
                    }
                    __syncthreads();
                    q1 = aggr[8];
                }
                const int e_end = (q1 < nq) ? offsA[queue[q1]] : offsA[nI];
                nE = e_end - e_base;
                // entries: one thread per searched hit of the round lists its optimal locations
                for (int q = q0 + tid; q < q1; q += NT) {
                    int item = queue[q];
                    list_entries(item, q - q0, offsA[item] - e_base);
                }
                if (use_slots) { for (int i = tid; i < ((q1 - q0) << logG); i += NT) bres[i] = 0xFFFFFFFFu; }
                else { for (int i = tid; i < (q1 - q0) * (kidx + 1) * MBW; i += NT) dmask[i] = 0; }
                __syncthreads();
            }
            stamp(3);
            phase3b_barcodes(nE);
            __syncthreads();
            stamp(4);
            phase3c_summary(q0, q1, e_base);
            __syncthreads();
            stamp(5);
            q0 = q1;
        }
    }

    __device__ __forceinline__ void zero_for_next() {
        // the encode target buffers: namask is OR-ed into, the next tile's orientation votes are counted up
        for (int i = tid; i < R * 2 * MW; i += NT) namask[i] = 0;
        for (int i = tid; i < R; i += NT) ocnt[(par ^ 1) * R + i] = 0;
        if (live) for (int i = tid; i < nr * ncand; i += NT) cumL[i] = 0;   // (the masks / scan arrays it overlays are dead by now)
        if (tid == 0) { aggr[9] = (int)popped; aggr[11] = 0; }
        __syncthreads();
        nxt = (uint32_t)aggr[9];

    }

    // ---- phase 4: the scorer of this tile (lowest wave(s), one lane per read)
    __device__ __forceinline__ void phase4_score() {
        unsigned long long t_sc0 = 0;
        if (timing) t_sc0 = clock64();   // diagnostic: the scorer wave's own time inside the shared region
        // G lanes per read while the tile is smaller than the wave (R <= 32: panels with many primers)
        const int lG = R <= 16 ? 2 : (R <= 32 ? 1 : 0), G = 1 << lG;
        const int r = tid >> lG, sub = tid & (G - 1);
        if (r < nr) {
            int L = lensC[r];
            bool filtered = (minlen != -1 && L < minlen) || (maxlen != -1 && L > maxlen);
            const unsigned long long pm_r = cfilt ? pmask[r] : 0ull;   // (the lanes sharing a read sit in one wave: all have read it
            if (cfilt && sub == 0) pmask[r] = 0ull;     //  when the lead lane clears it for the next tile)
            if (sub == 0) atomicAdd(&aggr[0], 1);
            if (filtered) {
                if (sub == 0) {
                    atomicAdd(&aggr[2], 1);
                    smx_op op;
                    op.sample = -1; op.trim_start = 0; op.trim_end = 0; op.pool = -1; op.p1 = op.p2 = -1; op.barcode = -1;
                    op.dist[0] = op.dist[1] = op.dist[2] = op.dist[3] = -1;
                    op.rtype = SMX_R_FILTERED; op.flags = 0; op.n_ops = 0; op.read = r0 + r;
                    opsL[r] = op;
                }
            } else {
                ReadCtx c;
                c.P = P; c.LP = LP; c.MBW = MBW; c.L = L; c.S = S;
                c.trim = sp ? (int)SMX_TRIM_BARCODES : P->trim; c.derep = sp ? (int)SMX_DEREP_BEST : P->derep;
                c.npair = NPAIR;
                c.tstr = tstr;
                if (cmode) { c.hits = hits; c.tiem = tieb; c.hmap = hmap + r * H; }
                else { c.hits = hits + r * H; c.tiem = tieb + r * H * tstr; c.hmap = nullptr; }
                c.hcand = cfilt ? hcand : nullptr;
                c.pm = pm_r;
                c.g = end_geom(L, S);
                int f = ocntC[r] & 0x7FFF, rv = ocntC[r] >> 16;
                int ori = 3;
                if (preorient) { if (f > 0 && rv == 0) ori = 1; else if (rv > 0 && f == 0) ori = 2; }
                c.set_live(ori);
                Emitter E;
                E.c = &c; E.primary = opsL + r; E.extra = extra; E.extra_cap = extra_cap;
                E.n_extra = tile_counter + 3;   // see the kernel's epilogue
                E.counts = counts; E.cum = cumL + r * ncand; E.aggr = aggr; E.read = r0 + r;
                E.n = 0; E.matched = false; E.overflow = false;
                const bool done = score_fast(E, ori, sub, G);
                if (sub == 0) {
                    if (!done) score_general(E, ori);
                    opsL[r].n_ops = (uint16_t)E.n;
                    if (E.matched) atomicAdd(&aggr[1], 1);
                    if (E.n > 1) atomicAdd(&aggr[6], 1);
                    if (E.overflow) atomicAdd(&aggr[7], 1);
                }
            }
        }
        if (timing) tacc[9] += clock64() - t_sc0;

    }

    // ---- phase 1 (of the NEXT tile, all other waves; every wave while the pipeline fills)
    __device__ __forceinline__ void phase1_encode() {
        // ---- phase 1 (of the NEXT tile): windows -> codes.  16-byte coalesced loads; A = revcomp of the head window.
        const int wid = live ? tid - 64 * SW : tid, nw = live ? NT - 64 * SW : NT;
        const bool timing_enc = dbg_phase != nullptr && tid == 64 * SW;   // diagnostic: this wave's own encode time
        unsigned long long t_enc0 = 0;
        if (timing_enc) t_enc0 = clock64();
        uint32_t r0n;
        int nrn;
        tile_span(nxt, r0n, nrn);
        int *lensN = lensL + (par ^ 1) * R;
        if (wid < nrn) lensN[wid] = lens[r0n + wid];
        // one 16-byte chunk: generic per-byte encode (short reads, search_len not a multiple of 16)
        auto encode_bytes = [&](int r, int L, int cpos, const uint4 &v) {
            int Sp = L < S ? L : S;
            unsigned w[4] = {v.x, v.y, v.z, v.w};
            unsigned char *rowA = codes + (r * 2 + 0) * CS, *rowB = codes + (r * 2 + 1) * CS;
            for (int b = 0; b < 16; b++) {
                int pos = cpos + b;
                unsigned ch = (w[b >> 2] >> ((b & 3) * 8)) & 0xFF;
                if (pos < S) {            // head byte i -> A[Sp-1-i] = code(complement)
                    if (pos < Sp) {
                        unsigned cd = lut[256 + ch];
                        int j = Sp - 1 - pos;
                        rowA[j] = (unsigned char)cd;
                        if (cd > 3) { atomicOr(&namask[(r * 2 + 0) * MW + (j >> 5)], 1u << (j & 31)); atomicOr(&ocnt[(par ^ 1) * R + r], OCNT_NA); }
                    }
                } else if (pos < 2 * S) { // tail byte j -> B[j]
                    int j = pos - S;
                    if (j < Sp) {
                        unsigned cd = lut[ch];
                        rowB[j] = (unsigned char)cd;
                        if (cd > 3) { atomicOr(&namask[(r * 2 + 1) * MW + (j >> 5)], 1u << (j & 31)); atomicOr(&ocnt[(par ^ 1) * R + r], OCNT_NA); }
                    }
                }
            }
        };
        // (the default-flags kernels are only launched behind the prescan: their ASCII fast path is compiled out)
        if ((S & 15) == 0 && (sp || aux.codes2 != nullptr)) {
            // behind the prescan: the transpose kernel has already turned every pure-ACGT window into 2-bit codes, row-major
            // per read and in DP order (end A reverse-complemented): one dword per (read, end, 16-column chunk), the tile's
            // dwords contiguous.  Four shift/mask/swap steps spell the dword out as 16 code bytes.  Reads the transpose kernel
            // flagged (a window with anything but upper-case ACGT) and reads shorter than the window take the ASCII path.
            const int hc = S >> 4, per = 2 * hc;
            const unsigned *c2 = aux.codes2 + (size_t)r0n * per;
            const unsigned permagic = (unsigned)((0x100000000ull + (unsigned)per - 1) / (unsigned)per);
            for (int ci = wid; ci < nrn * per; ci += nw) {
                const unsigned z = c2[ci];
                const int r = (int)__umulhi((unsigned)ci, permagic), rem = ci - __mul24(r, per);
                const int end = rem >= hc ? 1 : 0, c = rem - (end ? hc : 0);
                const unsigned flagged = aux.naflag[r0n + r];   // 1: shorter than the window, or not pure upper-case ACGT
                if (!flagged) {
                    unsigned *dst = (unsigned *)(codes + __mul24(r * 2 + end, CS)) + 4 * c;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const unsigned x = (z >> (2 * q)) & 0x03030303u;       // A 0, C 1, T 2, G 3 ...
                        dst[q] = x ^ ((x >> 1) & 0x01010101u);               // ... -> the kernels' A 0, C 1, G 2, T 3
                    }
                } else {   // the same 16 window bytes from the ASCII buffer (any 16-byte piece of that end: all get visited)
                    const uint4 v = *(const uint4 *)(windows + (size_t)(r0n + r) * stride + (end ? S : 0) + 16 * c);
                    encode_bytes(r, lens[r0n + r], (end ? S : 0) + 16 * c, v);
                }
            }
        } else if (!sp && (S & 15) == 0) {
            // fast path: chunks never straddle the head/tail boundary; items are ordered [all head chunks]
            // [all tail chunks] so that a wave is (almost always) uniform in role.  Full windows (len >= S):
            // four LUT lookups -> one packed dword store; the head is written reversed (reverse complement).
            const int hc = S >> 4, nhead = nrn * hc, S4 = S >> 2;
            for (int ci = wid; ci < 2 * nhead; ci += nw) {
                const bool tail = ci >= nhead;
                const int k = tail ? ci - nhead : ci;
                const int r = (int)__umulhi((unsigned)k, hcmagic), c = k - __mul24(r, hc);
                const uint4 v = *(const uint4 *)(windows + (size_t)(r0n + r) * stride + (tail ? S : 0) + 16 * c);
                const int L = lens[r0n + r];
                // ACGT fast path, four bases per dword without the LUT: (ch >> 1) & 3 maps A,C,T,G -> 0,1,2,3; swapping 2 and 3
                // gives the code, xor 3 the complement's code; one v_perm rebuilds the four letters from the codes and a
                // compare proves that the dword held nothing but upper-case ACGT (anything else: per-byte LUT path below)
                const unsigned w[4] = {v.x, v.y, v.z, v.w};
                unsigned pk[4];
                bool acgt = L >= S;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const unsigned x = (w[q] >> 1) & 0x03030303u;
                    const unsigned y = x ^ ((x >> 1) & 0x01010101u);
                    acgt = acgt && (__builtin_amdgcn_perm(0u, 0x54474341u, y) == w[q]);
                    pk[q] = tail ? y : __builtin_amdgcn_perm(0u, y ^ 0x03030303u, 0x00010203u);
                }
                if (acgt) {
                    unsigned *dst = (unsigned *)(codes + __mul24(r * 2 + (tail ? 1 : 0), CS));
#pragma unroll
                    for (int q = 0; q < 4; q++) dst[tail ? 4 * c + q : S4 - 1 - (4 * c + q)] = pk[q];
                } else {
                    encode_bytes(r, L, (tail ? S : 0) + 16 * c, v);
                }
            }
        } else if (!sp) {
            const int chunks = stride / 16;
            const uint4 *src = (const uint4 *)(windows + (size_t)r0n * stride);
            for (int ci = wid; ci < nrn * chunks; ci += nw) {
                int r = ci / chunks;
                encode_bytes(r, lens[r0n + r], (ci - r * chunks) * 16, src[ci]);
            }
        }
        if (timing_enc) tacc[8] += clock64() - t_enc0;
    }

    __device__ __forceinline__ void phase5_store() {
        // result records: LDS -> HBM, 16 bytes per lane, fully coalesced
        if (live) {
            const uint4 *srcv = (const uint4 *)opsL;
            uint4 *dstv = (uint4 *)(ops + r0);
            for (int i = tid; i < nr * 2; i += NT) dstv[i] = srcv[i];
        }
        // ---- optional parity dump
        if (live && dbg_hits) {
            for (int item = tid; item < nh; item += NT) {
                int r = divH(item);
                const HitL &hl = hits[cmode ? (int)hmap[item] : item];
                EndGeom g = end_geom(lensC[r], S);
                smx_hit o;
                o.pdist = hl.pdist; o.nloc = hl.nloc;
                o.first_start = hl.pdist >= 0 ? (int)hl.fs_j - g.j_lo + g.shift : -1;
                o.first_end = hl.pdist >= 0 ? (int)hl.jstar - g.j_lo + g.shift : -1;
                o.bbest = hl.bbest; o.ntied = hl.ntied;
                o.first_tied = (short)((hl.bbest >= 0) ? LP.pbc[LP.pbc_off[(item - r * H) >> 1] + hl.first_tied] : -1);
                o.tail_end = (hl.bbest >= 0 && tstr != (int)(sizeof(HitL) / 4)) ? hl.tail_end : -1;   // (tie word in the slot: no extent kept)
                o.flags = (int16_t)(hl.flags & 1);
                dbg_hits[(size_t)(r0 + r) * H + (item - r * H)] = o;
            }
        }
        if (dbg_hits) __syncthreads();   // the dump reads hits[]; the next tile's primer scan rewrites them

    }

    // dynamic tile queue: workgroups pull tiles from a global counter (zeroed on the stream before the
    // launch), so the tail is one tile long whatever the residency turns out to be
    // Software pipeline over tiles: while the lowest wave(s) run the scorer of tile t (one lane per read), the other
    // waves load and encode tile t+1.  codes / namask are dead by then; lens and the orientation votes are double-buffered.
    __device__ __forceinline__ void run() {
        cur = 0xFFFFFFFFu;                // tile being processed (none yet)
        par = 1;                          // cur's lens/ocnt buffer; the tile being encoded uses par ^ 1
        for (;;) {
            // keep per-thread address arithmetic inside the tile body: hoisted out of this loop it lives in VGPRs across
            // every phase and ends up spilled to scratch (HBM traffic, reload latency); recomputing it is a few ALU ops
            asm volatile("" : "+v"(tid));
            const bool have = cur != 0xFFFFFFFFu;
            // the tile to encode during this iteration: the queue pop is issued here, its result is only parked in LDS after
            // the barcode phases (a returning global atomic takes microseconds; storing it at once stalled wave 0, and with
            // it the first barrier of every tile)
            popped = 0;
            if (tid == 0) popped = atomicAdd(tile_counter, 1u);
            r0 = 0u;
            nr = 0;
            if (have) tile_span(cur, r0, nr);
            nh = nr * H;
            nI = nh;          // alignments with a record: all of them, or (compact mode) the flagged ones
            live = have;
            lensC = lensL + par * R; ocntC = ocnt + par * R;
            if (timing) tacc[10] = clock64();
            if (have) {
                phase2_primers();
                stamp(1);
                if (live) {
                    phase3a_entries();
                    stamp(2);
                    phase3_rounds();
                }
            }
            zero_for_next();
            stamp(0);
            // ---- phase 4 || phase 1: the scorer of this tile (lowest wave(s), one lane per read) runs beside the
            // load + encode of the next tile (all other waves; every wave while the pipeline fills)
            if (live && wave < SW) phase4_score();
            else if (nxt < n_tiles) phase1_encode();
            __syncthreads();
            stamp(6);
            phase5_store();
            stamp(7);
            if (nxt >= n_tiles) break;   // nothing was encoded: the queue is drained
            cur = nxt;
            par ^= 1;
            // (no barrier needed here: the next writers of hits / masks / opsL come after the barriers of phase 2)
        }
    }

    __device__ __forceinline__ void finish() {
        if (timing)
            for (int i = 0; i < 10; i++) dbg_phase[(size_t)blockIdx.x * 16 + i] += tacc[i];
        if (timing) dbg_phase[(size_t)blockIdx.x * 16 + 14] += tacc[0] != 0 || tacc[1] != 0 ? 1 : 0;   // did this workgroup get any tile?
        if (dbg_phase != nullptr && (tid & 63) == 0)   // placement of this wave: HW_ID (simd, wave slot, cu, se)
            dbg_phase[(size_t)blockIdx.x * 16 + 10 + wave] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    #undef STAMP
        // block aggregates -> global counters
        if (tid < 8 && aggr[tid]) atomicAdd(&counts[tid], (unsigned long long)aggr[tid]);
        __syncthreads();
        if (tid == 0) workgroup_done();

    }
};

// ------------------------------------------------------------------------------------------------
// BSV selects the barcode scan compiled into the kernel: see the comment in front of DemuxTile's template parameters above.
template <typename PW, int NT, int BSV, int CM = 0, int SP = 0>   // NT = 256 threads per workgroup (tiles of up to 64 reads)
__global__ __launch_bounds__(NT, SP == 2 ? SMX_SP2_WG : 4) void demux_kernel(DevPanel Pv, const uint8_t *__restrict__ windows,
                                                    const int32_t *__restrict__ lens, uint32_t n_reads, int R_arg,
                                                    smx_op *__restrict__ ops, smx_op *__restrict__ extra,
                                                    uint32_t extra_cap, uint32_t *n_extra,
                                                    unsigned long long *counts, smx_hit *dbg_hits, int8_t *dbg_bdist_arg,
                                                    unsigned *tile_counter, int use_slots_arg,
                                                    const unsigned *__restrict__ pre, uint32_t npad, DemuxAux aux) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    (void)npad;
    DemuxTile<PW, NT, BSV, CM, SP> D;
    if (!D.setup(lds, &Pv, windows, lens, n_reads, R_arg, ops, extra, extra_cap, n_extra, counts, dbg_hits, dbg_bdist_arg,
                 tile_counter, use_slots_arg, pre, aux))
        return;
    D.stage_panel();
    D.run();
    D.finish();
}

#if SMX_PART == 1
// ------------------------------------------------------------------------------------------------
// Single alignment (smx_align): one lane, same column step, arbitrary target length.
__global__ void align_kernel(const unsigned long long *peq, const unsigned long long *rpeq, int m,
                             const unsigned char *tcodes, int n, int k, int mode, int *out_dist,
                             unsigned char *endflag, int *starts) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long Pv = ~0ull, Mv = 0;
    int score = m, best = m + 1, top = m - 1;
    for (int j = 0; j < n; j++) {
        if (mode == 0) myers_step<unsigned long long, false>(peq[tcodes[j]], Pv, Mv, score, top);
        else myers_step<unsigned long long, true>(peq[tcodes[j]], Pv, Mv, score, top);
        if (score < best) best = score;
        endflag[j] = (unsigned char)score;   // raw score, thresholded below
    }
    if (best > k) { *out_dist = -1; return; }
    *out_dist = best;
    for (int j = 0; j < n; j++) {
        bool hit = endflag[j] == (unsigned char)best;
        endflag[j] = hit ? 1 : 0;
        starts[j] = 0;
        if (hit && mode == 0) {
            unsigned long long P2 = ~0ull, M2 = 0;
            int sc = m, lastc = 1;
            for (int c = 1; c <= m + best && j - (c - 1) >= 0; c++) {
                myers_step<unsigned long long, true>(rpeq[tcodes[j - (c - 1)]], P2, M2, sc, top);
                if (sc == best) lastc = c;
            }
            starts[j] = j - (lastc - 1);
        }
    }
}

// Many alignments per launch (smx_align_batch): one lane per alignment, the same column step.  q_peq: per distinct query
// 32 words (Peq of the query, then Peq of the reversed query); the raw scores of alignment i go to ws[toff[i] ..).
__global__ __launch_bounds__(64) void align_batch_kernel(const unsigned long long *__restrict__ q_peq, const int *__restrict__ q_len,
                                                        const unsigned *__restrict__ qidx, const unsigned char *__restrict__ tcodes,
                                                        const unsigned long long *__restrict__ toff, const int *__restrict__ kk,
                                                        const unsigned char *__restrict__ modes, unsigned n, unsigned char *ws,
                                                        int *__restrict__ out_dist, int *__restrict__ out_nloc,
                                                        int *__restrict__ out_starts, int *__restrict__ out_ends, unsigned cap) {
    const unsigned i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    const unsigned long long *peq = q_peq + (size_t)qidx[i] * 32, *rpeq = peq + 16;
    const int m = q_len[qidx[i]], k = kk[i], mode = modes[i], top = m - 1;
    const unsigned char *t = tcodes + toff[i];
    int nt = (int)(toff[i + 1] - toff[i]);
    if (mode == 1 && nt > m + k) nt = m + k;   // SHW: an end beyond m + k costs more than k
    unsigned char *sc = ws + toff[i];
    unsigned long long Pv = ~0ull, Mv = 0;
    int score = m, best = m + 1;
    for (int j = 0; j < nt; j++) {
        if (mode == 0) myers_step<unsigned long long, false>(peq[t[j]], Pv, Mv, score, top);
        else myers_step<unsigned long long, true>(peq[t[j]], Pv, Mv, score, top);
        best = score < best ? score : best;
        sc[j] = (unsigned char)score;
    }
    if (best > k) { out_dist[i] = -1; out_nloc[i] = 0; return; }
    out_dist[i] = best;
    int cnt = 0;
    for (int j = 0; j < nt; j++) {
        if (sc[j] != (unsigned char)best) continue;
        if ((unsigned)cnt < cap) {
            int st = 0;
            if (mode == 0) {   // edlib's start rule: last optimal position of the reversed problem
                unsigned long long P2 = ~0ull, M2 = 0;
                int s2 = m, lastc = 1;
                for (int c = 1; c <= m + best && j - (c - 1) >= 0; c++) {
                    myers_step<unsigned long long, true>(rpeq[t[j - (c - 1)]], P2, M2, s2, top);
                    if (s2 == best) lastc = c;
                }
                st = j - (lastc - 1);
            }
            out_starts[(size_t)i * cap + cnt] = st;
            out_ends[(size_t)i * cap + cnt] = j;
        }
        cnt++;
    }
    out_nloc[i] = cnt;
}

#endif   // SMX_PART == 1 (alignment kernels)

}  // namespace smx

// ------------------------------------------------------------------------------------------------
// launch glue used by smx_api.cpp
// The kernel's instantiations -- one per (primer word width, barcode scan variant, compact / redo mode, default-flags
// specialisation): 19 of them, half a minute of compile time each -- are spread over three translation units: this file is
// compiled three times (-DSMX_PART=1 / 2 / 3, in parallel), each part instantiates the kernels its table names, part 1 also
// holds the alignment kernels and the glue.
#define SMX_FN(B, C, S_) (const void *)smx::demux_kernel<unsigned, 256, B, C, S_>
#if SMX_PART == 1
extern "C" const void *smx_demux_fn_p1(int use64, int bsv, int cm, int sp) {   // 64-bit primer words; per-barcode scan (BSV 0)
    (void)sp;
    if (use64) {
        switch (bsv) {
            case 0: return (const void *)smx::demux_kernel<unsigned long long, 256, 0>;
            case 1: return (const void *)smx::demux_kernel<unsigned long long, 256, 1>;
            case 2: return (const void *)smx::demux_kernel<unsigned long long, 256, 2>;
            default: return (const void *)smx::demux_kernel<unsigned long long, 256, 3>;
        }
    }
    return cm == 0 ? SMX_FN(0, 0, 0) : (cm == 1 ? SMX_FN(0, 1, 0) : SMX_FN(0, 2, 0));
}
#elif SMX_PART == 2
extern "C" const void *smx_demux_fn_p2(int use64, int bsv, int cm, int sp) {   // bit-sliced scans k <= 3 and k 4..7, generic
    (void)use64; (void)sp;
    if (bsv == 1) return cm == 0 ? SMX_FN(1, 0, 0) : (cm == 1 ? SMX_FN(1, 1, 0) : SMX_FN(1, 2, 0));
    return cm == 0 ? SMX_FN(2, 0, 0) : (cm == 1 ? SMX_FN(2, 1, 0) : SMX_FN(2, 2, 0));
}
#else
extern "C" const void *smx_demux_fn_p3(int use64, int bsv, int cm, int sp) {   // the tails variant; the default-flags kernels
    (void)use64;
    if (bsv == 3) return cm == 0 ? SMX_FN(3, 0, 0) : (cm == 1 ? SMX_FN(3, 1, 0) : SMX_FN(3, 2, 0));
    if (sp == 2) return SMX_FN(1, 0, 2);
    if (sp == 3) return SMX_FN(1, 1, 3);
    return cm == 1 ? SMX_FN(1, 1, 1) : SMX_FN(1, 0, 1);
}
#endif
#undef SMX_FN

#if SMX_PART == 1
extern "C" const void *smx_demux_fn_p2(int use64, int bsv, int cm, int sp);
extern "C" const void *smx_demux_fn_p3(int use64, int bsv, int cm, int sp);
namespace {
const void *demux_fn(int use64, int bsv, int cm, int sp) {
    if (use64 || bsv == 0) return smx_demux_fn_p1(use64, bsv, cm, sp);
    if (bsv == 3 || (sp && bsv == 1 && cm != 2)) return smx_demux_fn_p3(use64, bsv, cm, sp);
    return smx_demux_fn_p2(use64, bsv, cm, sp);
}
int demux_bsv(const smx::DevPanel *P, int use_slots) {
    return (use_slots || !P->bs_ok) ? 0 : (P->kidx < 4 ? (P->trim == SMX_TRIM_TAILS ? 3 : 1) : 2);
}
// the default-flags specialisation applies to the k <= 3 lean kernel (not its tails variant, not the redo launch) with
// 64-read tiles
int demux_sp(const smx::DevPanel *P, int use64, int bsv, int cm, int R, int nitems, bool have_codes2) {
    const bool flags = have_codes2 && !use64 && bsv == 1 && cm != 2 && (cm == 0 || nitems == 256) && !P->cap_hits && !P->cap_ents &&
                       P->kidx == 3 && P->maxB <= 32 && !P->need_starts && P->trim == SMX_TRIM_BARCODES && P->derep == SMX_DEREP_BEST &&
                       P->preorient && P->minlen == -1 && P->maxlen == -1 && !P->dbg_phase && !(P->no_sp & 1);
    if (!flags) return 0;
    if (P->S == 160 && R == 32 && cm == 1) return 3;                       // wide windows, compact 32-read tiles
    if (P->S != 80) return 0;
    if (R == SMX_SP2_R && cm == 0 && P->NP == 2 && P->NPAIR == 1 && !(P->no_sp & 2)) return 2;
    return R == 64 ? 1 : 0;
}
}  // namespace

// which default-flags instantiation a launch with these parameters would get (0: the generic kernel)
extern "C" int smx_demux_sp_query(const smx::DevPanel *P, int use64, int use_slots, int cm, int R, int nitems, int have_prescan) {
    return demux_sp(P, use64, demux_bsv(P, use_slots), cm, R, nitems, have_prescan != 0);
}

extern "C" int smx_launch_demux(const smx::DevPanel *P, int use64, int R, int grid, size_t lds_bytes, void *stream,
                                const uint8_t *d_windows, const int32_t *d_lens, uint32_t n_reads, smx_op *d_ops,
                                smx_op *d_extra, uint32_t extra_cap, uint32_t *d_n_extra, uint64_t *d_counts,
                                smx_hit *d_hits, int8_t *d_bdist, unsigned *d_tile_counter, int use_slots,
                                const unsigned *d_pre, uint32_t npad, const smx::DemuxAux *aux_in) {
    smx::DemuxAux aux = {nullptr, nullptr, 0, 0, 0, 0, nullptr, nullptr};
    if (aux_in) aux = *aux_in;
    if (aux.nitems > 0 && (use_slots || !d_pre || !aux.match || !aux.ovf_list || aux.nitems > 256)) return (int)hipErrorInvalidValue;
    if (aux.redo && (!aux.ovf_list || aux.Rc < 1)) return (int)hipErrorInvalidValue;
    if ((aux.nitems > 0 || aux.redo) && use64) return (int)hipErrorInvalidValue;   // (the prescan serves primers of <= 31 nt only)
    if (aux.nitems > 0 && !aux.chain) return (int)hipErrorInvalidValue;              // a compact launch needs its redo launch
    if (R > 64) return (int)hipErrorInvalidValue;
    // d_tile_counter = {tile queue head, overflow tiles, finished workgroups, extra records}: zero at allocation, re-armed
    // by the last workgroup of every launch (of the last launch of a chain).  Slots mode never uses the bit-sliced scan.
    const int bsv = demux_bsv(P, use_slots), cm = aux.nitems > 0 ? 1 : (aux.redo ? 2 : 0);
    const void *fn = demux_fn(use64, bsv, cm, demux_sp(P, use64, bsv, cm, R, aux.nitems, aux.codes2 != nullptr && d_pre != nullptr));
    smx::DevPanel pv = *P;
    unsigned long long *counts = (unsigned long long *)d_counts;
    void *args[] = {&pv, &d_windows, &d_lens, &n_reads, &R, &d_ops, &d_extra, &extra_cap, &d_n_extra, &counts, &d_hits, &d_bdist,
                    &d_tile_counter, &use_slots, &d_pre, &npad, &aux};
    hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(256), args, lds_bytes, (hipStream_t)stream);
    return (int)(e != hipSuccess ? e : hipGetLastError());
}

extern "C" size_t smx_demux_lds_bytes(int use64, int NP, int NB, int S, int R, int maxB, int need_starts, int npmeta,
                                      int kidx, int slots, int bs, int nitems, int ncand, int tails, int nbstab) {
    return use64 ? (size_t)smx::make_layout<unsigned long long>(NP, NB, S, R, maxB, need_starts, npmeta, kidx, slots, bs, ncand, 0, 0, nitems, tails, nbstab).total
                 : (size_t)smx::make_layout<unsigned>(NP, NB, S, R, maxB, need_starts, npmeta, kidx, slots, bs, ncand, 0, 0, nitems, tails, nbstab).total;
}

extern "C" int smx_set_demux_lds_limit(int use64, size_t bytes) {
    hipError_t e = hipSuccess;
    for (int bsv = 0; bsv < 4; bsv++)
        for (int cm = 0; cm < (use64 ? 1 : 3); cm++)
            for (int sp = 0; sp < (use64 ? 1 : 4); sp++) {
                if (sp && (bsv != 1 || cm == 2)) continue;
                if (sp == 2 && cm != 0) continue;
                if (sp == 3 && cm != 1) continue;
                hipError_t r = hipFuncSetAttribute(demux_fn(use64, bsv, cm, sp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
                if (r != hipSuccess) e = r;
            }
    return (int)e;
}

// resident workgroups per CU of the kernel a launch with these parameters would use (cm: 0 dense, 1 compact, 2 redo)
extern "C" int smx_query_occupancy(const smx::DevPanel *P, int use64, int use_slots, int cm, int R, int nitems, size_t lds_bytes,
                                   int *blocks_per_cu, int have_prescan) {
    const int bsv = demux_bsv(P, use_slots);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, demux_fn(use64, bsv, cm, demux_sp(P, use64, bsv, cm, R, nitems, have_prescan != 0)),
                                                             256, lds_bytes);
    // The API divides the CU's 163 840 bytes by the request.  The hardware hands LDS out in 512-byte granules from 159 744
    // bytes (tools/ubench/lds_residency.hip: a 32 256-byte workgroup is resident four times, not five; 54 272 bytes twice,
    // not three times): workgroups the grid counts on but the CU cannot hold would start after the tile queue has drained.
    if (e == hipSuccess && lds_bytes > 0) {
        const int fit = (int)(SMX_LDS_POOL / ((lds_bytes + 511) & ~(size_t)511));
        if (*blocks_per_cu > fit) *blocks_per_cu = fit < 1 ? 1 : fit;
    }
    return (int)e;
}

extern "C" int smx_launch_align(void *stream, const unsigned long long *d_peq, const unsigned long long *d_rpeq, int m,
                                const unsigned char *d_tcodes, int n, int k, int mode, int *d_dist,
                                unsigned char *d_endflag, int *d_starts) {
    hipLaunchKernelGGL(smx::align_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_peq, d_rpeq, m, d_tcodes, n, k,
                       mode, d_dist, d_endflag, d_starts);
    return (int)hipGetLastError();
}

extern "C" int smx_launch_align_batch(void *stream, const unsigned long long *d_qpeq, const int *d_qlen, const unsigned *d_qidx,
                                      const unsigned char *d_tcodes, const unsigned long long *d_toff, const int *d_k,
                                      const unsigned char *d_modes, unsigned n, unsigned char *d_ws, int *d_dist, int *d_nloc,
                                      int *d_starts, int *d_ends, unsigned cap) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(smx::align_batch_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_qpeq, d_qlen, d_qidx,
                       d_tcodes, d_toff, d_k, d_modes, n, d_ws, d_dist, d_nloc, d_starts, d_ends, cap);
    return (int)hipGetLastError();
}
#endif   // SMX_PART == 1 (glue)
