// smx_prescan_core.h -- the primer prescan ("K_A"): Myers-equivalent HW (infix) alignment of every primer against both
// end windows of every read, BIT-SLICED ACROSS READS: one lane aligns one primer against one end of 32 reads at once
// (bit r of every word belongs to read r of the lane's 32-read group).  Replaces the per-(read, primer, end) bit-vector
// scan of the demux kernel for the common case -- full-length windows of upper-case A/C/G/T -- at ~1/4 of its VALU
// work: 5 three-input bit-ops per DP cell serve 32 reads, against ~22 instructions per text column per read.
// Reads with any other character in a window are redone by the demux kernel's scalar scan; the results are
// bit-identical (reference: match_one_end demultiplex.py:755-770, align_seq alignment.py:21-50, edlib HW mode).
// Reads shorter than search_len are covered too: the DP is causal (column j depends on columns <= j only), so the flags
// of the first n columns are those of the n-column text; the head window of a short read is right-aligned while it is
// packed, which makes revcomp(head) start at column 0 like the scalar scan's window A, and the consumer decodes the
// first min(len, search_len) columns only.
//
// This header is host/device code: the kernel (smx_prescan.hip) and the CPU unit test (tests/cpu/prescan_sim.cpp) run
// the same functions; on the host a "lane" is a loop index and LDS is a plain array.
//
// Two kernels, both over tiles of PRE_G x 32 reads:
//   transpose kernel (one workgroup per tile, memory bound)
//     phase 1  coalesced 16-byte loads of the windows; 16 ASCII bases -> one dword of 2-bit codes ((ch >> 1) & 3:
//              A 0, C 1, T 2, G 3; 8 VALU per 16 bases), staged in LDS as 32 x 32 bit blocks [group][16-column chunk][read]
//     phase 2  32 x 32 bit transposes in registers: word (column, plane) over the 32 reads of the group; end A blocks (the
//              head window) reversed and complemented: the DP then sees revcomp(head) like the scalar scan.  The planes
//              go to HBM as [tile][chunk][lane = group * 2 + end][32 words].
//   DP kernel (one wave per (tile, primer); no tile in LDS, so residency is set by registers alone)
//     lane = (group, end).  Per column: the four base-occurrence words (and the unions a degenerate primer letter
//              needs) go to a per-lane LDS scratch; row i reads its Eq word from a precomputed address (the pattern
//              letter selects the scratch row): no per-cell select instruction.
//              Unit-cost cell on the vertical / horizontal deltas (5 bit-ops), HW boundaries (top row free).
//              Last-row bookkeeping: gap = score - running minimum as a 5-plane bit-sliced counter;
//              lt = "new minimum here", e = "at the minimum here" -- two flag words per column.
//              Every 16 columns the 16 lt + 16 e words are transposed back (one 32 x 32 transpose): one word per read =
//              lt flags (bits 0-15) | e flags (bits 16-31) of the chunk's columns, written to HBM as
//              [tile][primer * 2 + end][chunk][read].  The consumer (prescan_decode, run by the demux kernel per
//              alignment) needs nothing else: best = m - popcount(lt), jstar = last lt column (= first column at the
//              final minimum), optimal ends = e flags from jstar on.
#ifndef SMX_PRESCAN_CORE_H
#define SMX_PRESCAN_CORE_H
#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define SMX_HD __host__ __device__ __forceinline__
#else
#define SMX_HD inline
#endif

namespace smx {

constexpr int PRE_G = 32;          // 32-read groups per tile: one wave = 32 groups x 2 ends
constexpr int PRE_TILE = PRE_G * 32;
constexpr int PRE_MAXROWS = 31;    // primers up to 31 nt (5-plane gap counter)
constexpr int PRE_MAXSYM = 8;      // distinct pattern letters (as A/C/G/T sets) per panel
constexpr int PRE_BLK = 33;        // dwords per 32 x 32 bit block in LDS (odd: conflict-free column access)
constexpr int PRE_SUBG = 8;        // 32-read groups per transpose workgroup: a sub-tile of 256 reads (a tile = 4 sub-tiles)
// Staging layout of a sub-tile: word (piece c, group g, read rr) at c * PRE_CS + g * PRE_BLK + rr.  ds_write_b32 / ds_read_b32
// bank = word mod 32 within each 32-lane half: phase 1 writes with (c, rr) varying -> bank 3 c + rr (+ g), phase 2 reads
// with (c, g) varying -> bank 3 c + g (+ r): both (nearly) conflict-free because PRE_CS = 3 (mod 32), PRE_BLK = 1 (mod 32).
constexpr int PRE_CS = 291;        // >= PRE_SUBG * PRE_BLK
static_assert(PRE_CS >= PRE_SUBG * PRE_BLK && PRE_CS % 32 == 3, "staging stride");
constexpr int PRE_SCRATCH = 2 * (PRE_MAXSYM + 1) * 64;   // dwords of per-wave scratch: [2 buffers][symbol][lane]

// Host-built description of the patterns (device copy passed by value to the kernel).
struct PreDesc {
    int NP, S, nsym, pure4;        // nsym = 4 + the largest number of degenerate-letter sets any one primer uses
    uint8_t m[64], k[64];
    uint8_t symmask[64][PRE_MAXSYM];   // per primer: symbol -> bit0 A, bit1 C, bit2 T, bit3 G (2-bit text code order);
                                       // symbols 0..3 are {A}, {C}, {T}, {G}, the others that primer's degenerate letters
    uint8_t sym[64][32];           // pattern letter of row i of primer p, as an index into its symbol table
};

// out[r] bit q = in[q] bit r
SMX_HD void transpose32(unsigned (&a)[32]) {
#define SMX_TSTAGE(J, MASK)                                                                    \
    _Pragma("unroll") for (int k_ = 0; k_ < 32; k_++) if ((k_ & J) == 0) {                     \
        const unsigned t_ = ((a[k_] >> J) ^ a[k_ + J]) & MASK;                                 \
        a[k_ + J] ^= t_;                                                                       \
        a[k_] ^= t_ << J;                                                                      \
    }
    SMX_TSTAGE(16, 0x0000FFFFu)
    SMX_TSTAGE(8, 0x00FF00FFu)
    SMX_TSTAGE(4, 0x0F0F0F0Fu)
    SMX_TSTAGE(2, 0x33333333u)
    SMX_TSTAGE(1, 0x55555555u)
#undef SMX_TSTAGE
}

// 16 ASCII bases (four little-endian dwords) -> 32 bits: byte i, bit pair kq <-> base 4 * kq + i; code = (ch >> 1) & 3
SMX_HD unsigned pack16(unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
    unsigned z = (w0 >> 1) & 0x03030303u;
    z |= (w1 << 1) & 0x0C0C0C0Cu;
    z |= (w2 << 3) & 0x30303030u;
    z |= (w3 << 5) & 0xC0C0C0C0u;
    return z;
}
// bit q of a packed dword belongs to base t of the chunk, plane (code bit) pl
SMX_HD int pack_t(int q) { return 4 * ((q & 7) >> 1) + (q >> 3); }
SMX_HD int pack_pl(int q) { return q & 1; }

// The same packed dword for the reverse complement of the 16 bases: base t <-> 15 - t (byte i <-> 3 - i, bit pair kq <-> 3 - kq),
// complement = code ^ 2 in the (ch >> 1) & 3 coding (A 0 <-> T 2, C 1 <-> G 3)
SMX_HD unsigned pack16_revcomp(unsigned z) {
#if defined(__HIP_DEVICE_COMPILE__)
    z = __builtin_amdgcn_perm(0u, z, 0x00010203u);
#else
    z = (z >> 24) | ((z >> 8) & 0xFF00u) | ((z << 8) & 0xFF0000u) | (z << 24);
#endif
    z = ((z & 0x0F0F0F0Fu) << 4) | ((z >> 4) & 0x0F0F0F0Fu);
    z = ((z & 0x33333333u) << 2) | ((z >> 2) & 0x33333333u);
    return z ^ 0xAAAAAAAAu;
}
// Does the dword hold nothing but upper-case A / C / G / T?  0 iff it does (OR the results of a window's dwords).
SMX_HD unsigned acgt_mismatch(unsigned w) {
    const unsigned x = (w >> 1) & 0x03030303u;   // A 0, C 1, T 2, G 3
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(0u, 0x47544341u, x) ^ w;
#else
    static const unsigned char L[4] = {'A', 'C', 'T', 'G'};
    unsigned r = 0;
    for (int i = 0; i < 4; i++) r |= (unsigned)L[(x >> (8 * i)) & 3] << (8 * i);
    return r ^ w;
#endif
}

// Row-major 2-bit codes for the demux kernel ("codes2"): per read 2 * CH dwords, [end][16-column chunk] in DP order (end A =
// reverse complement of the head window, end B = the tail window); dword = pack16 of the chunk's 16 columns.
SMX_HD size_t codes2_word(size_t read, int CH, int end, int chunk) { return (read * 2 + (size_t)end) * (size_t)CH + (size_t)chunk; }
// piece c of a window row (c < CH: head, else tail), its packed dword z -> (dword of codes2, where it goes)
SMX_HD unsigned codes2_from_piece(unsigned z, int c, int CH, int *end, int *chunk) {
    if (c < CH) { *end = 0; *chunk = CH - 1 - c; return pack16_revcomp(z); }
    *end = 1; *chunk = c - CH;
    return z;
}

// ---- consumer side: one alignment from its CH chunk words (w[c * cstride], c = 0 .. CH-1).  Returns the best distance
// (> k: no match) and, for a match, jstar, the number of optimal ends and the S-bit mask of optimal end columns
// (mrow: MW = ceil(S / 32) words, bits below jstar cleared).
// Only the first n_valid columns count (reads shorter than the window).  CHT > 0: chunk count known at compile time
// (straight-line code: all loads first, no per-chunk branches); CHT = 0: run-time CH.
template <int CHT>
SMX_HD int prescan_decode(const unsigned *w, size_t cstride, int CH, int MW, int m, int k, int n_valid, unsigned *mrow,
                          int *jstar_out, int *nloc_out) {
    unsigned short *mrow16 = (unsigned short *)mrow;
    int nlt = 0, jstar = 0;
    auto chunk = [&](int c, unsigned xraw) {
        const int nv = n_valid - 16 * c;   // valid columns of this chunk
        const unsigned keep = nv >= 16 ? 0xFFFFu : (nv > 0 ? (1u << nv) - 1u : 0u);
        const unsigned x = xraw & (keep | (keep << 16));
        const unsigned lt16 = x & 0xFFFFu;
#if defined(__HIP_DEVICE_COMPILE__)
        nlt += __popc(lt16);
        if (lt16) jstar = 16 * c + 31 - __clz((int)lt16);
#else
        nlt += __builtin_popcount(lt16);
        if (lt16) jstar = 16 * c + 31 - __builtin_clz(lt16);
#endif
        mrow16[c] = (unsigned short)(x >> 16);
    };
    if (CHT > 0) {
        unsigned xs[CHT > 0 ? CHT : 1];   // all loads are issued before the first is used: one memory round trip
#pragma unroll
        for (int c = 0; c < CHT; c++) xs[c] = w[(size_t)c * cstride];
#pragma unroll
        for (int c = 0; c < CHT; c++) chunk(c, xs[c]);
        CH = CHT;
    } else {
        for (int c0 = 0; c0 < CH; c0 += 4) {
            unsigned xs[4];
#pragma unroll
            for (int u = 0; u < 4; u++) xs[u] = c0 + u < CH ? w[(size_t)(c0 + u) * cstride] : 0u;
#pragma unroll
            for (int u = 0; u < 4; u++) if (c0 + u < CH) chunk(c0 + u, xs[u]);
        }
    }
    if (CH & 1) mrow16[CH] = 0;
    const int best = m - nlt;
    int nloc = 0;
    if (best <= k) {
        for (int i = 0; i < MW; i++) {
            unsigned v = mrow[i];
            if (i < (jstar >> 5)) v = 0;
            else if (i == (jstar >> 5)) v &= ~0u << (jstar & 31);
            mrow[i] = v;
#if defined(__HIP_DEVICE_COMPILE__)
            nloc += __popc(v);
#else
            nloc += __builtin_popcount(v);
#endif
        }
    }
    *jstar_out = jstar;
    *nloc_out = nloc;
    return best;
}

// ------------------------------------------------------------------------------------------------
// Phase 1, one 16-byte piece: piece q of the tile = (read q / ppr, piece c = q % ppr of its window row); ppr = pieces per
// read = 2 * S / 16.  `w` = the four dwords of the piece.  Block (g, c) holds the packed dwords of the group's 32 reads.
// (read_in_sub = read index inside the 256-read sub-tile)
SMX_HD unsigned prescan_store_piece(unsigned *planes, int read_in_sub, int c, unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
    const int g = read_in_sub >> 5, rr = read_in_sub & 31;
    const unsigned z = pack16(w0, w1, w2, w3);
    planes[c * PRE_CS + g * PRE_BLK + rr] = z;
    return z;
}

// Head window piece c of a read shorter than S: the stored window holds head[0 : L) left-aligned; the prescan wants it
// right-aligned (byte x of the S-byte head window <- head[x - (S - L)]) so that its reverse complement starts at column 0.
SMX_HD void prescan_short_head_piece(const uint8_t *row, int c, int S, int L, unsigned (&w)[4]) {
    const int sh = S - (L < 0 ? 0 : L);
    const int x0 = 16 * c - sh;            // source byte of the piece's first byte (negative: before the window, reads as 0)
    const int base = x0 & ~3, sh8 = (x0 & 3) * 8;
    unsigned d[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {          // five aligned dwords cover the 16 bytes; all loads unconditional (clamped)
        const int idx = base + 4 * k;
        const int cl = idx < 0 ? 0 : (idx > S - 4 ? S - 4 : idx);
        const unsigned v = *(const unsigned *)(row + cl);
        d[k] = (idx >= 0 && idx <= S - 4) ? v : 0u;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = sh8 ? (d[k] >> sh8) | (d[k + 1] << (32 - sh8)) : d[k];
}

// Phase 2, one block: 32 x 32 bit transpose, result in DP order.  Pieces c < CH are the head window (end A: reversed and
// complemented), pieces c >= CH the tail window (end B).  DP order: out[2 * t' + plane] for DP column t' of the chunk.
SMX_HD void prescan_transpose_block(const unsigned *planes, int g, int c, int CH, unsigned (&out)[32]) {
    unsigned a[32];
    const unsigned *pb = planes + c * PRE_CS + g * PRE_BLK;
#pragma unroll
    for (int r = 0; r < 32; r++) a[r] = pb[r];
    transpose32(a);
    const bool endA = c < CH;
#pragma unroll
    for (int d = 0; d < 32; d++) {   // word d = (DP column t' = d >> 1, plane d & 1); packed bit q = 8 * (t & 3) + 2 * (t >> 2) + plane
        const int tB = d >> 1, tA = 15 - (d >> 1), pl = d & 1;
        const unsigned wB = a[8 * (tB & 3) + 2 * (tB >> 2) + pl];
        const unsigned wA = pl ? ~a[8 * (tA & 3) + 2 * (tA >> 2) + pl] : a[8 * (tA & 3) + 2 * (tA >> 2) + pl];
        out[d] = endA ? wA : wB;
    }
}
// where block (g, c) of a tile lives in the HBM plane buffer: chunk, lane = g * 2 + end; its 32 words are contiguous
// (128 bytes: whole cache lines for the transposing lane's eight 16-byte stores; the DP lane reads them back in four
// pairs of 16-byte loads per chunk -- 40 load instructions per wave and tile, whatever their coalescing)
SMX_HD int prescan_block_chunk(int c, int CH) { return c < CH ? CH - 1 - c : c - CH; }
SMX_HD int prescan_block_lane(int g, int c, int CH) { return g * 2 + (c < CH ? 0 : 1); }
SMX_HD size_t prescan_plane_word(int chunk, int lane, int d) { return ((size_t)chunk * 64 + lane) * 32 + d; }

// Occurrence words of one text column -> this lane's scratch column (sc points at [buffer][symbol 0][lane]).
// NX = extra symbol rows compiled in (0, or PRE_MAXSYM - 4 for panels with degenerate primer letters: unions of the four
// base words under wave-uniform masks; a run-time loop here would put control flow into every unrolled DP column).
template <int NX>
SMX_HD void prescan_write_occ(unsigned *sc, unsigned b0, unsigned b1, const unsigned (&xm)[NX > 0 ? NX : 1][4]) {
    const unsigned E0 = ~b0 & ~b1, E1 = b0 & ~b1, E2 = ~b0 & b1, E3 = b0 & b1;   // A, C, T, G
    sc[0] = E0; sc[64] = E1; sc[128] = E2; sc[192] = E3;
#pragma unroll
    for (int x = 0; x < NX; x++)
        sc[(4 + x) * 64] = (E0 & xm[x][0]) | (E1 & xm[x][1]) | (E2 & xm[x][2]) | (E3 & xm[x][3]);
}

// The DP: one lane = primer p against one (group, end) of the tile's reads: lane = group * 2 + end.  `gpl` = the tile's
// planes in the HBM layout (prescan_plane_word), `scratch` = this wave's PRE_SCRATCH dwords of LDS.
// MR = rows compiled in; a pattern of m <= MR rows occupies rows MR - m .. MR - 1.  The rows above it are inert: they
// read the all-ones word of scratch row PRE_MAXSYM and start with a zero vertical delta, so every delta on them stays 0 and
// the first pattern row sees the free top row of the HW alignment (straight-line code, no per-row branch; jumping into
// the unrolled rows instead cost hundreds of register copies per column).
// MT = 1: also count the new minima per read and write the match word (the compact demux tiles read it; other panels
// skip the nine bit-ops per column).
template <int MR, int NX, int MT = 1>
SMX_HD void prescan_dp(const unsigned *gpl, unsigned *scratch, int lane, int CH, const PreDesc &D, int p, unsigned *wout,
                       size_t cstride, unsigned *mout) {
    const int m = D.m[p], skip = MR - m;
    unsigned Pv[MR], Mv[MR];
    int aoff[MR];
#pragma unroll
    for (int i = 0; i < MR; i++) {
        Pv[i] = i >= skip ? ~0u : 0u; Mv[i] = 0u;
        aoff[i] = (i >= skip ? (int)D.sym[p][i - skip] : PRE_MAXSYM) * 64 + lane;
    }
    // (the compiler merges these four byte reads of the kernel-argument struct into one s_load_dword; its SGPR base must stay
    // dword aligned -- scalar loads ignore the two low bits of the base -- which holds as long as the base is the kernarg
    // pointer itself and 8 p + 148 the offset.  A row-split variant of this kernel, round 3, got (kernarg + p) as base -- reused
    // from the m[p] byte read -- and every primer with p % 4 != 0 silently read another primer's masks: the c3 panel's GPU
    // tests, whose degenerate primers sit at p = 2, 3, 6, 7, are the guard.)
    unsigned xm[NX > 0 ? NX : 1][4];   // uniform: all-ones where extra symbol 4 + x contains A / C / T / G
#pragma unroll
    for (int x = 0; x < (NX > 0 ? NX : 1); x++)
#pragma unroll
        for (int b = 0; b < 4; b++) xm[x][b] = ((D.symmask[p][4 + x] >> b) & 1) ? ~0u : 0u;   // unused symbols have an empty mask
    unsigned g0 = 0, g1 = 0, g2 = 0, g3 = 0, g4 = 0, zero = ~0u;   // gap = score - running minimum, starts at 0
    unsigned n0 = 0, n1 = 0, n2 = 0, n3 = 0, n4 = 0;               // number of new minima so far (best = m - that), bit-sliced
    constexpr int bufw = PRE_SCRATCH / 2;   // compile-time buffer stride: the second buffer is an immediate offset
    unsigned *sc0 = scratch + lane, *sc1 = scratch + bufw + lane;
    sc0[PRE_MAXSYM * 64] = ~0u; sc1[PRE_MAXSYM * 64] = ~0u;   // the inert rows' Eq word (a row no symbol uses)
    // plane words of the current and the next four-column group (8 words each: column t of a group = words 2t, 2t + 1),
    // fetched one group ahead
    unsigned pw[2][8];
    auto fetch = [&](int grp, unsigned (&dst)[8]) {   // grp = chunk * 4 + group in chunk; two 16-byte loads per lane
        const int chunk = grp >> 2, q = (grp & 3) * 2;
#if defined(__HIP_DEVICE_COMPILE__)
        const uint4 *g4 = (const uint4 *)gpl;
        const uint4 u0 = g4[((size_t)chunk * 64 + lane) * 8 + q], u1 = g4[((size_t)chunk * 64 + lane) * 8 + q + 1];
        dst[0] = u0.x; dst[1] = u0.y; dst[2] = u0.z; dst[3] = u0.w; dst[4] = u1.x; dst[5] = u1.y; dst[6] = u1.z; dst[7] = u1.w;
#else
        for (int j = 0; j < 8; j++) dst[j] = gpl[prescan_plane_word(chunk, lane, 4 * q + j)];
#endif
    };
    fetch(0, pw[0]);
    prescan_write_occ<NX>(sc0, pw[0][0], pw[0][1], xm);   // prologue: column 0
    unsigned eqb[2][8];   // Eq words of the current / next row group
#pragma unroll
    for (int u = 0; u < 8; u++) eqb[0][u] = scratch[aoff[u]];   // column 0, group 0 (16 * NG columns per chunk: even parity)
    for (int ch = 0; ch < CH; ch++) {
        unsigned fl[32];   // [0,16): lt flags, [16,32): e flags of this chunk's columns
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int gq = t >> 2, tq = t & 3, cur = gq & 1;   // four groups per chunk: the parity restarts with every chunk
            if (tq == 0 && (ch * 4 + gq + 1 < CH * 4)) fetch(ch * 4 + gq + 1, pw[cur ^ 1]);
            // next column's occurrence words go to the other scratch buffer while this column's are read
            if (tq < 3) prescan_write_occ<NX>((t & 1) ? sc0 : sc1, pw[cur][2 * tq + 2], pw[cur][2 * tq + 3], xm);
            else if (ch * 4 + gq + 1 < CH * 4) prescan_write_occ<NX>((t & 1) ? sc0 : sc1, pw[cur ^ 1][0], pw[cur ^ 1][1], xm);
            const unsigned *sc = ((t & 1) ? scratch + bufw : scratch);
            unsigned Ph = 0u, Mh = 0u;   // HW: the top row is free
            // rows in groups of eight.  The Eq words of the NEXT group (of the next column after the last group: its
            // occurrence words were written one column ahead) are requested before this group's cells run, so one
            // counted wait per group covers its eight reads; the scheduling fences pin that order.
            constexpr int NG = (MR + 7) / 8;
#pragma unroll
            for (int gi = 0; gi < NG; gi++) {
                const int cur = (t * NG + gi) & 1;
                const int gn = gi + 1 < NG ? gi + 1 : 0;                        // next group ...
                const unsigned *scn = gi + 1 < NG ? sc : ((t & 1) ? scratch : scratch + bufw);   // ... and its buffer
#pragma unroll
                for (int u = 0; u < 8; u++) if (gn * 8 + u < MR) eqb[cur ^ 1][u] = scn[aoff[gn * 8 + u]];
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                for (int u = 0; u < 8; u++) if (gi * 8 + u < MR) {
                    const int i = gi * 8 + u;
#if defined(__HIP_DEVICE_COMPILE__)
                    // the cell as exactly five instructions, in this order: left to itself the compiler interleaves the
                    // cells of several columns (more instructions per cell and hundreds of spilled registers).
                    // v_bitop3 truth table index = s0 << 2 | s1 << 1 | s2: 0xfe = a | b | c, 0xf1 = a | ~(b | c)
                    unsigned Z, nPh, nMh;
                    asm volatile("v_bitop3_b32 %[z], %[eq], %[mh], %[mv] bitop3:0xfe\n\t"
                                 "v_bitop3_b32 %[pho], %[mv], %[z], %[pv] bitop3:0xf1\n\t"
                                 "v_and_b32 %[mho], %[pv], %[z]\n\t"
                                 "v_bitop3_b32 %[pv], %[mh], %[z], %[ph] bitop3:0xf1\n\t"
                                 "v_and_b32 %[mv], %[ph], %[z]"
                                 : [z] "=&v"(Z), [pho] "=&v"(nPh), [mho] "=&v"(nMh), [pv] "+v"(Pv[i]), [mv] "+v"(Mv[i])
                                 : [eq] "v"(eqb[cur][u]), [mh] "v"(Mh), [ph] "v"(Ph));
                    Ph = nPh; Mh = nMh;
#else
                    const unsigned Z = eqb[cur][u] | Mh | Mv[i];
                    const unsigned nPh = Mv[i] | ~(Z | Pv[i]);
                    const unsigned nMh = Pv[i] & Z;
                    const unsigned nPv = Mh | ~(Z | Ph);
                    const unsigned nMv = Ph & Z;
                    Pv[i] = nPv; Mv[i] = nMv; Ph = nPh; Mh = nMh;
#endif
                }
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            // last row: score += Ph - Mh; gap = score - running minimum
            const unsigned lt = zero & Mh;        // at the minimum and going down: a new minimum, the gap stays 0
            unsigned cy = Ph, bw = Mh ^ lt, tt;
            tt = g0 & cy; g0 ^= cy; cy = tt;
            tt = g1 & cy; g1 ^= cy; cy = tt;
            tt = g2 & cy; g2 ^= cy; cy = tt;
            tt = g3 & cy; g3 ^= cy; cy = tt;
            g4 ^= cy;
            tt = ~g0 & bw; g0 ^= bw; bw = tt;
            tt = ~g1 & bw; g1 ^= bw; bw = tt;
            tt = ~g2 & bw; g2 ^= bw; bw = tt;
            tt = ~g3 & bw; g3 ^= bw; bw = tt;
            g4 ^= bw;
            zero = ~(g0 | g1 | g2 | g3 | g4);
            if (MT) {
                cy = lt;
                tt = n0 & cy; n0 ^= cy; cy = tt;
                tt = n1 & cy; n1 ^= cy; cy = tt;
                tt = n2 & cy; n2 ^= cy; cy = tt;
                tt = n3 & cy; n3 ^= cy; cy = tt;
                n4 ^= cy;
            }
            fl[t] = lt;
            fl[16 + t] = zero;
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);   // keep the live ranges column-sized: no hoisting of later columns' LDS reads
#endif
        }
        transpose32(fl);   // -> one word per read: lt flags | e flags << 16 of this chunk
#if defined(__HIP_DEVICE_COMPILE__)
        uint4 *dst = (uint4 *)(wout + (size_t)ch * cstride);
#pragma unroll
        for (int r = 0; r < 8; r++) dst[r] = make_uint4(fl[4 * r], fl[4 * r + 1], fl[4 * r + 2], fl[4 * r + 3]);
#else
        for (int r = 0; r < 32; r++) wout[(size_t)ch * cstride + r] = fl[r];
#endif
    }
    // match word: bit r = read r reaches distance <= k somewhere in the S columns  <=>  new minima >= m - k (bit-sliced
    // compare against the uniform threshold).  For a read shorter than the window this is a superset of "matches within
    // its own columns" (a minimum over fewer columns is not smaller): the consumer uses it to skip alignments only.
    if (MT) {
        const unsigned thr = (unsigned)(m - (int)D.k[p]);
        const unsigned nb[5] = {n0, n1, n2, n3, n4};
        unsigned gt = 0u, eq = ~0u;
#pragma unroll
        for (int b = 4; b >= 0; b--) {
            const unsigned tb = ((thr >> b) & 1u) ? ~0u : 0u;
            gt |= eq & nb[b] & ~tb;
            eq &= ~(nb[b] ^ tb);
        }
        *mout = gt | eq;
    }
}


// ------------------------------------------------------------------------------------------------
// Host: PreDesc from the searched patterns.  eq(pattern letter, text base) is the panel's equality relation.  Returns
// false when the prescan cannot serve this panel (the demux kernel then scans every alignment itself).
inline bool prescan_build_desc(PreDesc *D, int NP, int S, const char *const *patterns, const int *lens, const int *ks,
                               bool (*eq)(unsigned char, unsigned char)) {
    if (NP < 1 || NP > 64 || S < 16 || S > 256 || (S & 15) != 0) return false;
    D->NP = NP; D->S = S;
    static const char bases[4] = {'A', 'C', 'T', 'G'};   // 2-bit text code order
    D->nsym = 4;
    for (int p = 0; p < NP; p++) {
        if (lens[p] < 1 || lens[p] > PRE_MAXROWS || ks[p] < 0 || ks[p] >= lens[p]) return false;
        D->m[p] = (uint8_t)lens[p];
        D->k[p] = (uint8_t)ks[p];
        int ns = 4;
        for (int s = 0; s < PRE_MAXSYM; s++) D->symmask[p][s] = s < 4 ? (uint8_t)(1u << s) : 0;
        for (int i = 0; i < lens[p]; i++) {
            unsigned mk = 0;
            for (int b = 0; b < 4; b++) if (eq((unsigned char)patterns[p][i], (unsigned char)bases[b])) mk |= 1u << b;
            int s = 0;
            while (s < ns && D->symmask[p][s] != mk) s++;
            if (s == ns) {
                if (ns == PRE_MAXSYM) return false;   // more distinct degenerate letters in one primer than scratch rows
                D->symmask[p][ns++] = (uint8_t)mk;
            }
            D->sym[p][i] = (uint8_t)s;
        }
        if (ns > D->nsym) D->nsym = ns;
    }
    D->pure4 = D->nsym == 4;
    return true;
}

}  // namespace smx
#endif
