// smx_api.cpp -- host side of libsmx.so: panel compilation, launch glue, the C ABI of include/smx.h.
// Compiled with hipcc together with smx_kernels.hip.  No CPU implementation of the hot path lives here:
// every compute entry point needs a HIP device and fails with SMX_ERR_DEVICE otherwise.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "smx.h"
#include "smx_internal.h"
#include "smx_prescan_core.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) return fail(SMX_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// ---- IUPAC equality (reference constants.py:13-20): symmetric, NOT transitive
struct EqTable {
    bool eq[128][128];
    EqTable() {
        memset(eq, 0, sizeof(eq));
        for (int c = 0; c < 128; c++) eq[c][c] = true;
        const char *pairs[] = {"YC", "YT", "RA", "RG", "NA", "NC", "NG", "NT", "WA", "WT", "MA", "MC", "SC", "SG",
                               "KG", "KT", "BC", "BG", "BT", "DA", "DG", "DT", "HA", "HC", "HT", "VA", "VC", "VG"};
        for (const char *p : pairs) {
            eq[(int)p[0]][(int)p[1]] = true;
            eq[(int)p[1]][(int)p[0]] = true;
        }
    }
};
const EqTable &eqt() {
    static EqTable t;
    return t;
}

int code_of(unsigned char ch) {
    for (int c = 0; c < 15; c++)
        if (smx::kCodeChars[c] == (char)ch) return c;
    return 15;
}

unsigned char complement_of(unsigned char ch) {   // Bio.Seq complement, ambiguous DNA table, case kept, U->A
    static const char *from = "ACGTMRWSYKVHDBXNUacgtmrwsykvhdbxnu";
    static const char *to = "TGCAKYWSRMBDHVXNAtgcakywsrmbdhvxna";
    for (int i = 0; from[i]; i++)
        if ((unsigned char)from[i] == ch) return (unsigned char)to[i];
    return ch;
}

// bit i of peq[c] = eq(pattern[i], char of code c); code 15 never matches
bool build_peq(const char *pat, int m, unsigned long long *peq16, std::string *bad) {
    for (int c = 0; c < 16; c++) peq16[c] = 0;
    for (int i = 0; i < m; i++) {
        unsigned char pc = (unsigned char)pat[i];
        if (pc >= 128 || code_of(pc) == 15) {
            if (bad) *bad = std::string("pattern character '") + (char)pc + "' is outside the IUPAC DNA alphabet";
            return false;
        }
        for (int c = 0; c < 15; c++)
            if (eqt().eq[pc][(int)smx::kCodeChars[c]]) peq16[c] |= 1ull << i;
    }
    return true;
}

template <typename T>
size_t blob_add(std::vector<unsigned char> &blob, const std::vector<T> &v) {
    size_t off = (blob.size() + 15) & ~(size_t)15;
    blob.resize(off + std::max<size_t>(v.size() * sizeof(T), 16));
    if (!v.empty()) memcpy(blob.data() + off, v.data(), v.size() * sizeof(T));
    return off;
}

}  // namespace

#define SMX_MAX_STREAMS 16   // distinct streams one panel may be launched on

extern "C" size_t smx_packed_stride_for(int32_t S);   // smx_io.cpp
extern "C" int smx_launch_unpack_windows(void *stream, const uint8_t *d_packed, uint8_t *d_windows, uint32_t n_reads, int S,
                                         int pstride, int wstride, int n_cu);   // smx_pack.hip

struct DevBuf {   // grow-only device buffer of the host-buffer convenience path
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct smx_panel {
    smx::DevPanel hp;                 // scalar fields valid; pointers filled at upload
    std::vector<unsigned char> blob;  // host image of the device allocation
    size_t o_ppeq, o_prpeq, o_bpeq, o_lut, o_pm, o_pk, o_pdir, o_pfidx, o_pbc_off, o_pbc, o_bm, o_pair_f, o_pair_r,
        o_pair_pool, o_bsre, o_bstab = 0, o_pairrec = 0, o_specrec = 0;
    int use64 = 0;
    bool env_no_lean_tails = false, env_force_slots = false, env_debug_overflow = false, env_debug = false;   // read once at create
    int R = 0;          // lean mode tile (no per-barcode slots)
    size_t lds = 0;
    int R_slots = 0;    // slots mode tile (--trim tails, parity dumps)
    size_t lds_slots = 0;
    // device state (lazy, one device per process)
    void *d_blob = nullptr;
    int device = -1;
    int n_cu = 0;
    int blocks_per_cu = 1, blocks_per_cu_slots = 1;   // resident workgroups per CU: lean / slots kernel
    std::mutex ws_mutex;                     // smx_batch_run is serialised per panel (one workspace)
    DevBuf ws[8];                            // windows, lens, ops, extra, n_extra, counts, hits, bdist
    // Launch counters {tile queue head, -, finished workgroups, extra records}, 64 bytes per slot, self re-arming.
    // One slot per stream the panel has been launched on: launches on one stream are ordered, launches on different
    // streams (double-buffered pipelines) each pull tiles from their own queue.
    unsigned *d_tile_counter = nullptr;
    std::mutex tc_mutex;
    std::vector<void *> tc_streams;          // slot -> stream (valid where tc_used)
    std::vector<char> tc_used;               // a lane gives its slot back when it is destroyed: the next new stream reuses it
    // primer prescan (smx_prescan.hip): bit-sliced HW alignment of every primer over both end windows, run in front of
    // the demux kernel, which then only redoes the alignments the prescan cannot take (smx_prescan_core.h)
    bool pre_ok = false;
    smx::PreDesc pre;
    int pre_mr = 24, pre_nx = 0, pre_blocks_t = 1, pre_blocks_d = 8;   // longest primer, degenerate symbols, residency
    size_t pre_lds = 0;                      // transpose kernel staging
    DevBuf pre_planes[SMX_MAX_STREAMS];      // per stream slot: the 2-bit text planes of the batch (read-tile major)
    DevBuf pre_recs[SMX_MAX_STREAMS];
    DevBuf pre_match[SMX_MAX_STREAMS];       // per stream slot: match words [tile][2 * NP][32 groups] (bit = read reaches the threshold)
    DevBuf pre_codes[SMX_MAX_STREAMS];       // per stream slot: row-major 2-bit codes [read][end][chunk] + one flag byte per read behind them
    // compact mode of the lean kernel (panels with many primers): tiles of Rc reads that keep per-alignment records only
    // for the nitems alignments the match words flag; a tile that needs more goes on the overflow list and is redone by a
    // dense launch (R, lds) right behind the compact one.  nitems == 0: off.
    int Rc = 0, nitems = 0, blocks_per_cu_c = 1;
    int share = 1;      // smx_panel_set_streams: batches the caller keeps in flight on as many streams
    size_t lds_c = 0;
    DevBuf ovf[SMX_MAX_STREAMS];             // per stream slot: overflow list, one entry per compact tile
    hipEvent_t kev[4] = {nullptr, nullptr, nullptr, nullptr};   // smx_debug_kernel_times: start, after transpose, after DP, end
    bool kev_on = false, kev_pre = false;        // per stream slot: [2 * NP][search_len / 16][n_reads rounded up to a tile] flag words
    unsigned long long *d_phase = nullptr;   // SMX_PHASE_TIMING diagnostic
    int phase_grid = 0;
};

extern "C" {

// shared with smx_io.cpp: set the thread-local message and return `code`
int smx_set_error(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

int smx_abi_version(void) { return SMX_ABI_VERSION; }
const char *smx_last_error(void) { return g_err.c_str(); }

int smx_device_init(int device, int *n_devices) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(SMX_ERR_DEVICE, "no HIP device available (%s)", hipGetErrorString(e));
    if (n_devices) *n_devices = n;
    if (device < 0 || device >= n) return fail(SMX_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    return SMX_OK;
}

int smx_panel_create(const smx_panel_desc *d, smx_panel **out) {
    if (!d || !out) return fail(SMX_ERR_ARG, "null argument");
    if (d->abi_version != SMX_ABI_VERSION) return fail(SMX_ERR_ARG, "ABI version mismatch: %u", d->abi_version);
    const int NP = (int)d->n_primers, NB = (int)d->n_barcodes, NS = (int)d->n_specimens, NPAIR = (int)d->n_pairs;
    if (NP <= 0 || NB <= 0 || NS <= 0 || NPAIR <= 0) return fail(SMX_ERR_ARG, "empty panel");
    if (NP > 64) return fail(SMX_ERR_UNSUPPORTED, "more than 64 distinct primers (%d)", NP);
    if (NPAIR > 127) return fail(SMX_ERR_UNSUPPORTED, "more than 127 primer pairs (%d)", NPAIR);
    if (NB > 4096) return fail(SMX_ERR_UNSUPPORTED, "more than 4096 distinct barcodes (%d)", NB);   // (b1, b2) table: 32 B per pair
    if (d->search_len < 1 || d->search_len > 256)
        return fail(SMX_ERR_UNSUPPORTED, "search_len %d outside 1..256", d->search_len);
    if (d->k_index < 0 || d->k_index > 32) return fail(SMX_ERR_UNSUPPORTED, "index edit distance %d outside 0..32", d->k_index);
    if (d->trim < 0 || d->trim > 3 || d->dereplicate < 0 || d->dereplicate > 1) return fail(SMX_ERR_ARG, "bad trim/dereplicate");

    smx_panel *P = new smx_panel();
    smx::DevPanel &h = P->hp;
    memset(&h, 0, sizeof(h));
    h.NP = NP; h.NB = NB; h.NS = NS; h.NPAIR = NPAIR;
    h.S = d->search_len;
    h.wstride = ((2 * h.S) + 15) & ~15;
    h.kidx = d->k_index;
    h.bmax = d->barcode_len_max;
    h.pfmin = d->prefilter_min_len;
    h.preorient = d->preorient ? 1 : 0;
    h.trim = d->trim;
    h.derep = d->dereplicate;
    h.minlen = d->min_length;
    h.maxlen = d->max_length;
    h.need_starts = (d->trim == SMX_TRIM_PRIMERS || d->trim == SMX_TRIM_TAILS || d->want_starts) ? 1 : 0;

    std::string bad;
    std::vector<unsigned long long> ppeq(NP * 16), prpeq(NP * 16);
    std::vector<int> pm(NP), pk(NP), pdir(NP), pfidx(NP), pbc_off(NP + 1);
    int maxm = 0, maxB = 1;
    for (int p = 0; p < NP; p++) {
        int a = (int)d->primer_rc_off[p], m = (int)d->primer_rc_off[p + 1] - a;
        if (m < 1 || m > 64) { delete P; return fail(SMX_ERR_UNSUPPORTED, "primer %d has length %d (supported 1..64)", p, m); }
        std::string pat(d->primer_rc + a, m), rpat(pat.rbegin(), pat.rend());
        if (!build_peq(pat.data(), m, &ppeq[p * 16], &bad) || !build_peq(rpat.data(), m, &prpeq[p * 16], &bad)) {
            delete P;
            return fail(SMX_ERR_UNSUPPORTED, "primer %d: %s", p, bad.c_str());
        }
        pm[p] = m;
        pk[p] = d->primer_k[p];
        if (pk[p] < 0 || pk[p] >= m) { delete P; return fail(SMX_ERR_UNSUPPORTED, "primer %d: edit distance %d must be in 0..len-1", p, pk[p]); }
        pdir[p] = d->primer_dir[p] ? 1 : 0;
        pfidx[p] = d->primer_file_index[p];
        if (pfidx[p] < 0 || pfidx[p] > 2000) { delete P; return fail(SMX_ERR_UNSUPPORTED, "primer file index %d outside 0..2000", pfidx[p]); }
        pbc_off[p] = (int)d->primer_bc_off[p];
        maxm = std::max(maxm, m);
        maxB = std::max(maxB, (int)(d->primer_bc_off[p + 1] - d->primer_bc_off[p]));
    }
    pbc_off[NP] = (int)d->primer_bc_off[NP];
    {   // prescan description: the searched patterns as A/C/G/T sets per row
        std::vector<std::string> pats(NP);
        std::vector<const char *> pp(NP);
        std::vector<int> pl(NP);
        for (int p = 0; p < NP; p++) {
            pats[p].assign(d->primer_rc + d->primer_rc_off[p], d->primer_rc_off[p + 1] - d->primer_rc_off[p]);
            pp[p] = pats[p].c_str();
            pl[p] = (int)pats[p].size();
        }
        memset(&P->pre, 0, sizeof(P->pre));
        P->pre_ok = maxm <= smx::PRE_MAXROWS && !getenv("SMX_NO_PRESCAN") &&
                    smx::prescan_build_desc(&P->pre, NP, h.S, pp.data(), pl.data(), pk.data(),
                                            [](unsigned char a, unsigned char b) { return a < 128 && b < 128 && eqt().eq[a][b]; });
        if (P->pre_ok) {
            P->pre_mr = maxm;
            P->pre_nx = P->pre.nsym - 4;
            P->pre_lds = smx_prescan_lds_bytes(h.S);
            if (P->pre_lds > 160 * 1024) P->pre_ok = false;
        }
    }
    if (maxB > 1024) { delete P; return fail(SMX_ERR_UNSUPPORTED, "more than 1024 barcodes on one primer (%d)", maxB); }
    std::vector<int> pbc(std::max(pbc_off[NP], 1));
    for (int i = 0; i < pbc_off[NP]; i++) {
        pbc[i] = (int)d->primer_bc[i];
        if (pbc[i] < 0 || pbc[i] >= NB) { delete P; return fail(SMX_ERR_ARG, "primer_bc[%d] out of range", i); }
    }
    std::vector<unsigned> bpeq(NB * 16);
    std::vector<int> bm(NB);
    for (int b = 0; b < NB; b++) {
        int a = (int)d->barcode_rc_off[b], m = (int)d->barcode_rc_off[b + 1] - a;
        if (m < 1 || m > 32) { delete P; return fail(SMX_ERR_UNSUPPORTED, "barcode %d has length %d (supported 1..32)", b, m); }
        if (d->k_index >= m) { delete P; return fail(SMX_ERR_UNSUPPORTED, "index edit distance %d >= barcode length %d", d->k_index, m); }
        unsigned long long t[16];
        if (!build_peq(d->barcode_rc + a, m, t, &bad)) { delete P; return fail(SMX_ERR_UNSUPPORTED, "barcode %d: %s", b, bad.c_str()); }
        for (int c = 0; c < 16; c++) bpeq[b * 16 + c] = (unsigned)t[c];
        bm[b] = m;
    }
    // bit-sliced barcode tables: usable when every barcode has the same length <= 16 and k <= 7
    const int MBWh = (maxB + 31) / 32;
    bool bs_ok = d->k_index <= 7 && !getenv("SMX_NO_BITSLICE");
    for (int b = 0; b < NB; b++) if (bm[b] != bm[0] || bm[b] > 16) bs_ok = false;
    std::vector<unsigned> bsre((size_t)NP * 16 * 16 * MBWh, 0u);
    if (bs_ok)
        for (int p = 0; p < NP; p++)
            for (int li = pbc_off[p]; li < pbc_off[p + 1]; li++) {
                int gb = pbc[li], bi = li - pbc_off[p];
                for (int row = 0; row < 16; row++)   // rows past the barcode are wildcards (bitsliced_shw_pad)
                    for (int c = 0; c < 16; c++)
                        if (row >= bm[gb] || ((bpeq[gb * 16 + c] >> row) & 1u))
                            bsre[((((size_t)p * MBWh + (bi >> 5)) * 16 + row) * 16 + c)] |= 1u << (bi & 31);   // [primer][word][row][code]
            }
    // primers with the same barcode list (the same kit in every pool) share one table: the 8-primer panel keeps 2 of 8 in LDS
    std::vector<int> bs_tab(NP, 0);
    {
        const size_t tw = (size_t)16 * 16 * MBWh;
        std::vector<unsigned> packed;
        int nt = 0;
        for (int p = 0; p < NP; p++) {
            int found = -1;
            for (int q = 0; q < nt && found < 0; q++)
                if (std::equal(bsre.begin() + (size_t)p * tw, bsre.begin() + (size_t)(p + 1) * tw, packed.begin() + (size_t)q * tw)) found = q;
            if (found < 0) { packed.insert(packed.end(), bsre.begin() + (size_t)p * tw, bsre.begin() + (size_t)(p + 1) * tw); found = nt++; }
            bs_tab[p] = found;
        }
        if (getenv("SMX_NO_TABLE_SHARING")) { nt = NP; for (int p = 0; p < NP; p++) bs_tab[p] = p; }   // A/B and test hook
        else bsre.swap(packed);
        h.n_bstab = nt;
    }
    h.bs_ok = bs_ok ? 1 : 0;
    if (const char *e = getenv("SMX_TEST_CAPS")) sscanf(e, "%d,%d", &h.cap_hits, &h.cap_ents);
    // the remaining test / A-B switches are read here, once: launches never look at the environment
    h.no_sp = (getenv("SMX_NO_SPECIALISE") ? 1 : 0) | (getenv("SMX_NO_SPECIALISE_NP") ? 2 : 0);
    P->env_no_lean_tails = getenv("SMX_NO_LEAN_TAILS") != nullptr;
    P->env_force_slots = getenv("SMX_FORCE_SLOTS") != nullptr;
    P->env_debug_overflow = getenv("SMX_DEBUG_OVERFLOW") != nullptr;
    P->env_debug = getenv("SMX_DEBUG") != nullptr;
    h.bs_m = bm[0];
    if (h.pfmin != 0 && h.pfmin < 2) { delete P; return fail(SMX_ERR_UNSUPPORTED, "prefilter min length %d < 2: disable the prefilter", h.pfmin); }
    if (h.bmax + h.kidx > 200) { delete P; return fail(SMX_ERR_UNSUPPORTED, "barcode length + k too large"); }
    h.maxB = maxB;
    h.n_pbc = pbc_off[NP];
    std::vector<unsigned char> lut(512);
    for (int c = 0; c < 256; c++) {
        lut[c] = (unsigned char)code_of((unsigned char)c);
        lut[256 + c] = (unsigned char)code_of(complement_of((unsigned char)c));
    }
    std::vector<int> pair_f(NPAIR), pair_r(NPAIR), pair_pool(NPAIR);
    for (int i = 0; i < NPAIR; i++) {
        pair_f[i] = (int)d->pair_fwd[i];
        pair_r[i] = (int)d->pair_rev[i];
        pair_pool[i] = d->pair_pool[i];
        if (pair_f[i] >= NP || pair_r[i] >= NP || pdir[pair_f[i]] != 0 || pdir[pair_r[i]] != 1) {
            delete P;
            return fail(SMX_ERR_ARG, "pair %d is not (forward primer, reverse primer)", i);
        }
    }
    // (b1,b2) -> chain of specimens in file order (Specimens.specimen_for_exact_match walks file order)
    std::vector<int> pairhead((size_t)NB * NB, -1), spec_next(NS, -1), spec_pool(NS), tail((size_t)NB * NB, -1);
    std::vector<unsigned long long> p1m(NS), p2m(NS);
    for (int s = 0; s < NS; s++) {
        unsigned b1 = d->spec_b1[s], b2 = d->spec_b2[s];
        if (b1 >= (unsigned)NB || b2 >= (unsigned)NB) { delete P; return fail(SMX_ERR_ARG, "specimen %d barcode index out of range", s); }
        size_t key = (size_t)b1 * NB + b2;
        if (pairhead[key] < 0) pairhead[key] = s; else spec_next[tail[key]] = s;
        tail[key] = s;
        p1m[s] = d->spec_p1mask[s];
        p2m[s] = d->spec_p2mask[s];
        spec_pool[s] = d->spec_pool[s];
    }
    auto &B = P->blob;
    P->o_ppeq = blob_add(B, ppeq); P->o_prpeq = blob_add(B, prpeq); P->o_bpeq = blob_add(B, bpeq); P->o_lut = blob_add(B, lut);
    P->o_pm = blob_add(B, pm); P->o_pk = blob_add(B, pk); P->o_pdir = blob_add(B, pdir); P->o_pfidx = blob_add(B, pfidx);
    P->o_pbc_off = blob_add(B, pbc_off); P->o_pbc = blob_add(B, pbc); P->o_bm = blob_add(B, bm);
    P->o_pair_f = blob_add(B, pair_f); P->o_pair_r = blob_add(B, pair_r); P->o_pair_pool = blob_add(B, pair_pool);
    {   // packed lookup tables (32 bytes per barcode pair)
        std::vector<smx::SpecRec> specrec(NS), pairrec((size_t)NB * NB);
        for (int s2 = 0; s2 < NS; s2++) specrec[s2] = {p1m[s2], p2m[s2], s2, spec_next[s2], spec_pool[s2], 0};
        for (size_t k2 = 0; k2 < pairrec.size(); k2++)
            pairrec[k2] = pairhead[k2] >= 0 ? specrec[pairhead[k2]] : smx::SpecRec{0ull, 0ull, -1, -1, -1, 0};
        P->o_pairrec = blob_add(B, pairrec); P->o_specrec = blob_add(B, specrec);
    }
    P->o_bsre = blob_add(B, bsre);
    P->o_bstab = blob_add(B, bs_tab);

    P->use64 = maxm > 32 ? 1 : 0;
    // tile size: largest R in {64, 32, ...} whose LDS image lets 4 workgroups share a CU's 160 KiB
    // (SMX_TILE_R / SMX_LDS_BUDGET override for tuning experiments)
    // Tile size: the largest R (<= 64 reads, one scorer lane per read) whose tile fits a quarter of the CU's LDS, so that
    // four workgroups stay resident; but a tile twice as large at three workgroups per CU keeps more reads in flight
    // (6R vs 4R) and wins for panels with many primers (measured on the 8-primer panel: R = 32 x 3 beats R = 16 x 4 by 7 %,
    // R = 64 x 2 loses 45 %).  SMX_TILE_R / SMX_LDS_BUDGET override for tuning experiments.
    const bool budget_forced = getenv("SMX_LDS_BUDGET") != nullptr;
    auto lds_blocks = [](size_t need) { return (int)(SMX_LDS_POOL / ((need + 511) & ~(size_t)511)); };   // workgroups of `need` bytes a CU holds
    size_t budget = (SMX_LDS_POOL / 4) & ~(size_t)511;
    if (budget_forced) budget = (size_t)atol(getenv("SMX_LDS_BUDGET"));
    int rmax = 64;
    if (const char *e = getenv("SMX_TILE_R")) rmax = std::max(1, std::min(64, atoi(e)));
    const int npmeta = 6 * NP + 1 + h.n_pbc + NB + 3 * NPAIR;
    const int tails = h.trim == SMX_TRIM_TAILS ? 1 : 0;   // the per-entry extent array of the lean tails kernel (BSV 3)
    for (int slots = 0; slots < 2; slots++) {
        auto pick = [&](size_t bud, int *Rout, size_t *need_out) {
            for (int R = rmax; R >= 1; R >>= 1) {
                size_t need = smx_demux_lds_bytes(P->use64, NP, NB, h.S, R, maxB, h.need_starts, npmeta, h.kidx, slots,
                                                  h.bs_ok, 0, 2 * NPAIR, tails, h.n_bstab);
                if (need <= bud || R == 1) { *Rout = R; *need_out = need; return; }
            }
        };
        int R4 = 1, R3 = 1;
        size_t n4 = 0, n3 = 0;
        pick(budget, &R4, &n4);
        if (!budget_forced) pick((SMX_LDS_POOL / 3) & ~(size_t)511, &R3, &n3);
        const bool three = !budget_forced && !slots && R3 > R4;   // (the slots kernel measured 2.5 % slower that way)
        if (slots) { P->R_slots = three ? R3 : R4; P->lds_slots = three ? n3 : n4; }
        else { P->R = three ? R3 : R4; P->lds = three ? n3 : n4; }
    }
    if (const char *e = getenv("SMX_LDS_PAD")) P->lds += (size_t)atol(e);   // tuning experiment: residency vs LDS size
    // compact mode: worth it when the dense tile had to shrink (R * 2 NP records do not fit) and the prescan is there to say
    // which alignments matter.  Largest tile (multiples of 8 reads) at four workgroups per CU, or a larger one at three if
    // that keeps more reads in flight (8-primer panel: the kernel's time falls as a + b / reads in flight from R = 24 x 4 to
    // R = 64 x 3).  SMX_COMPACT=0 turns it off, SMX_COMPACT_ITEMS / SMX_COMPACT_R are test / tuning hooks.
    {
        const char *ce = getenv("SMX_COMPACT");
        int items = 256;
        if (const char *e = getenv("SMX_COMPACT_ITEMS")) items = std::max(2 * NP, std::min(256, atoi(e)));
        if (P->pre_ok && !(ce && atoi(ce) == 0) && (P->R < 64 || getenv("SMX_COMPACT_ITEMS"))) {
            auto need_c = [&](int R) { return smx_demux_lds_bytes(P->use64, NP, NB, h.S, R, maxB, h.need_starts, npmeta, h.kidx, 0, h.bs_ok, items, 2 * NPAIR, tails, h.n_bstab); };
            int best_R = 0, best_blocks = 0;
            size_t best_need = 0;
            for (int R = 64; R >= 8; R -= 8) {
                const size_t need = need_c(R);
                if (need > SMX_LDS_POOL) continue;
                const int blocks = std::min(4, lds_blocks(need));
                if (blocks < 3) continue;   // two workgroups per CU lose more to exposed latency than their larger tiles win back
                                            // (measured: 8-primer panel, -l 160: R = 64 x 2 is 40 % slower than R = 40 x 3)
                if (R * blocks > best_R * best_blocks) { best_R = R; best_blocks = blocks; best_need = need; }
            }
            // a tile size that has a default-flags instantiation wins over a larger generic one (wide-window stress shape:
            // 32-read tiles on `SP = 3` 1.11 ms per 10^6 reads, 48-read tiles on the generic compact kernel 1.15)
            for (int R = 64; R >= 8; R -= 8) {
                const size_t need = need_c(R);
                if (need > SMX_LDS_POOL || lds_blocks(need) < 3) continue;
                if (smx_demux_sp_query(&h, P->use64, 0, 1, R, items, 1) != 0) {
                    if (R != best_R) { best_R = R; best_blocks = std::min(4, lds_blocks(need)); best_need = need; }
                    break;
                }
            }
            if (const char *e = getenv("SMX_COMPACT_R")) { best_R = std::max(1, std::min(64, atoi(e))); best_need = need_c(best_R); best_blocks = 1; }
            const int dense_blocks = std::min(4, lds_blocks(P->lds));
            if (best_R > 0 && (best_R * best_blocks >= P->R * dense_blocks || getenv("SMX_COMPACT_ITEMS") || getenv("SMX_COMPACT_R"))) {
                P->Rc = best_R; P->nitems = items; P->lds_c = best_need;
            }
        }
    }
    if (P->lds > 160 * 1024 || P->lds_slots > 160 * 1024) {
        const size_t need = std::max(P->lds, P->lds_slots);
        delete P;
        return fail(SMX_ERR_UNSUPPORTED, "panel needs %zu bytes of LDS per read tile", need);
    }
    *out = P;
    return SMX_OK;
}

void smx_panel_destroy(smx_panel *P) {
    if (!P) return;
    if (P->d_phase) {   // diagnostic: print the per-phase share of block cycles
        std::vector<unsigned long long> h((size_t)P->phase_grid * 16);
        if (hipMemcpy(h.data(), P->d_phase, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            unsigned long long sum[10] = {0}, tot = 0;
            for (int b = 0; b < P->phase_grid; b++)
                for (int i = 0; i < 10; i++) { sum[i] += h[(size_t)b * 16 + i]; tot += h[(size_t)b * 16 + i]; }
            const char *names[8] = {"zero+barrier", "primer_scan", "orient+scan", "entries", "barcode_scan", "summary",
                                    "scorer||encode", "store"};
            fprintf(stderr, "[smx phase timing] R=%d lds=%zu blocks/CU=%d:", P->R, P->lds, P->blocks_per_cu);
            tot -= sum[8] + sum[9];   // [8] = the first encode wave's own encode time (inside the scorer||encode region), [9] = the scorer wave's own time
            for (int i = 0; i < 8; i++) fprintf(stderr, " %s=%.1f%%", names[i], tot ? 100.0 * sum[i] / tot : 0.0);
            fprintf(stderr, " (inside the region: scorer wave %.1f%%, first encode wave %.1f%%)", tot ? 100.0 * sum[9] / tot : 0.0, tot ? 100.0 * sum[8] / tot : 0.0);
            fprintf(stderr, "\n");
            if (P->env_debug) {   // where did wave w of each workgroup land?  hist[w][simd]
                {
                    unsigned long long worked = 0, launches = 0;
                    for (int b = 0; b < P->phase_grid; b++) { worked += h[(size_t)b * 16 + 14]; }
                    fprintf(stderr, "[smx placement] workgroup-launches that processed at least one tile: %llu (grid %d)\n", worked, P->phase_grid);
                    (void)launches;
                }
                int hist[4][4] = {{0}};
                for (int b = 0; b < P->phase_grid; b++)
                    for (int w = 0; w < 4; w++) hist[w][(h[(size_t)b * 16 + 10 + w] >> 4) & 3]++;
                for (int w = 0; w < 4; w++)
                    fprintf(stderr, "[smx placement] wave %d on simd 0..3: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
                for (int b = 0; b < 12 && b < P->phase_grid; b++) {
                    unsigned v = (unsigned)h[(size_t)b * 16 + 10];
                    fprintf(stderr, "[smx placement] block %d wave0: slot %u simd %u cu %u sh %u se %u\n", b, v & 15, (v >> 4) & 3, (v >> 8) & 15, (v >> 12) & 1, (v >> 13) & 7);
                }
            }
        }
        (void)hipFree(P->d_phase);
    }
    if (P->d_blob) (void)hipFree(P->d_blob);
    if (P->d_tile_counter) (void)hipFree(P->d_tile_counter);
    for (auto &b : P->ws) b.release();
    for (auto &b : P->pre_recs) b.release();
    for (auto &b : P->pre_match) b.release();
    for (auto &b : P->pre_codes) b.release();
    for (auto &b : P->ovf) b.release();
    for (auto &b : P->pre_planes) b.release();
    for (auto &e : P->kev) if (e) (void)hipEventDestroy(e);
    delete P;
}

size_t smx_counts_len(const smx_panel *P) { return P ? (size_t)SMX_CNT_SPECIMEN0 + P->hp.NS : 0; }
size_t smx_window_stride(const smx_panel *P) { return P ? (size_t)P->hp.wstride : 0; }
size_t smx_packed_stride(const smx_panel *P) { return P ? smx_packed_stride_for(P->hp.S) : 0; }
size_t smx_hits_per_read(const smx_panel *P) { return P ? (size_t)2 * P->hp.NP : 0; }
size_t smx_bdist_per_read(const smx_panel *P) { return P ? (size_t)2 * P->hp.NP * P->hp.maxB : 0; }

int smx_pack_windows(const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, int32_t S, uint8_t *windows,
                     int32_t *lens) {
    if (!bases || !offsets || !windows || !lens || S < 1) return fail(SMX_ERR_ARG, "null argument");
    const size_t stride = ((size_t)(2 * S) + 15) & ~(size_t)15;
    for (uint32_t i = 0; i < n_reads; i++) {
        uint64_t a = offsets[i], b = offsets[i + 1];
        if (b < a || b - a > 0x7FFFFFFFull) return fail(SMX_ERR_ARG, "read %u: bad offsets", i);
        int L = (int)(b - a), Sp = L < S ? L : S;
        uint8_t *w = windows + (size_t)i * stride;
        memset(w, 0, stride);
        memcpy(w, bases + a, (size_t)Sp);
        memcpy(w + S, bases + b - Sp, (size_t)Sp);
        lens[i] = L;
    }
    return SMX_OK;
}

// The launch-counter slot of `stream` on this panel (-1: none / no free slot).  claim: take a free slot for a new stream.
static int stream_slot(smx_panel *P, void *stream, bool claim) {
    std::lock_guard<std::mutex> g(P->tc_mutex);
    int free_slot = -1;
    for (size_t i = 0; i < P->tc_streams.size(); i++) {
        if (P->tc_used[i] && P->tc_streams[i] == stream) return (int)i;
        if (!P->tc_used[i] && free_slot < 0) free_slot = (int)i;
    }
    if (!claim) return -1;
    if (free_slot < 0) {
        if (P->tc_streams.size() == SMX_MAX_STREAMS) return -1;
        P->tc_streams.push_back(stream);
        P->tc_used.push_back(1);
        return (int)P->tc_streams.size() - 1;
    }
    P->tc_streams[free_slot] = stream;
    P->tc_used[free_slot] = 1;
    return free_slot;
}

// A stream is going away (synchronised by the caller): its slot -- counters and prescan buffers -- is free for the next one.
static void stream_release(smx_panel *P, void *stream) {
    std::lock_guard<std::mutex> g(P->tc_mutex);
    for (size_t i = 0; i < P->tc_streams.size(); i++)
        if (P->tc_used[i] && P->tc_streams[i] == stream) P->tc_used[i] = 0;
}

// After a failed or out-of-step launch on `stream` (already synchronised): re-arm THAT stream's launch counters only.
// Other lanes of the panel may have kernels in flight on their own slots.
static void stream_reset_counters(smx_panel *P, void *stream) {
    const int sl = stream_slot(P, stream, false);
    if (sl >= 0 && P->d_tile_counter) (void)hipMemset(P->d_tile_counter + 16 * sl, 0, 64);
}

static int ensure_device(smx_panel *P) {
    if (P->d_blob) return SMX_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(SMX_ERR_DEVICE, "libsmx has no CPU path: no HIP device available (%s)", hipGetErrorString(e));
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    P->device = dev;
    P->n_cu = prop.multiProcessorCount;
    HIP_TRY(hipMalloc(&P->d_blob, P->blob.size()));
    HIP_TRY(hipMemcpy(P->d_blob, P->blob.data(), P->blob.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void **)&P->d_tile_counter, 64 * SMX_MAX_STREAMS));
    HIP_TRY(hipMemset(P->d_tile_counter, 0, 64 * SMX_MAX_STREAMS));   // the kernel re-arms these counters itself after every launch
    unsigned char *b = (unsigned char *)P->d_blob;
    smx::DevPanel &h = P->hp;
    h.ppeq = (const unsigned long long *)(b + P->o_ppeq);
    h.prpeq = (const unsigned long long *)(b + P->o_prpeq);
    h.bpeq = (const unsigned *)(b + P->o_bpeq);
    h.lut = b + P->o_lut;
    h.pm = (const int *)(b + P->o_pm); h.pk = (const int *)(b + P->o_pk);
    h.pdir = (const int *)(b + P->o_pdir); h.pfidx = (const int *)(b + P->o_pfidx);
    h.pbc_off = (const int *)(b + P->o_pbc_off); h.pbc = (const int *)(b + P->o_pbc); h.bm = (const int *)(b + P->o_bm);
    h.pair_f = (const int *)(b + P->o_pair_f); h.pair_r = (const int *)(b + P->o_pair_r);
    h.pair_pool = (const int *)(b + P->o_pair_pool);
    h.pairrec = (const smx::SpecRec *)(b + P->o_pairrec);
    h.specrec = (const smx::SpecRec *)(b + P->o_specrec);
    h.bs_re = (const unsigned *)(b + P->o_bsre);
    h.bs_tab = (const int *)(b + P->o_bstab);
    if (std::max(std::max(P->lds, P->lds_slots), P->lds_c) > 64 * 1024) {
        const size_t lim = std::max(std::max(P->lds, P->lds_slots), P->lds_c);
        int rc = smx_set_demux_lds_limit(P->use64, lim);
        if (rc != 0) return fail(SMX_ERR_DEVICE, "cannot raise the dynamic LDS limit to %zu bytes", lim);
    }
    // persistent grid = exactly the resident workgroups (tiles are pulled from a queue): a workgroup that starts
    // after the queue has drained would only pay the panel staging and its one-time register spills
    {
        int occ = 0;   // (behind a compact launch the dense lean kernel runs as the redo instantiation)
        if (smx_query_occupancy(&P->hp, P->use64, 0, P->nitems > 0 ? 2 : 0, P->R, 0, P->lds, &occ, P->pre_ok ? 1 : 0) != 0 || occ < 1) occ = 4;
        P->blocks_per_cu = occ;
        occ = 0;
        if (smx_query_occupancy(&P->hp, P->use64, 1, 0, P->R_slots, 0, P->lds_slots, &occ, P->pre_ok ? 1 : 0) != 0 || occ < 1) occ = 4;
        P->blocks_per_cu_slots = occ;
        if (P->nitems > 0) {
            occ = 0;
            if (smx_query_occupancy(&P->hp, P->use64, 0, 1, P->Rc, P->nitems, P->lds_c, &occ, P->pre_ok ? 1 : 0) != 0 || occ < 1) occ = 4;
            P->blocks_per_cu_c = occ;
        }
    }
    if (P->pre_ok) {
        if (P->pre_lds > 64 * 1024 && smx_prescan_set_lds_limit(P->pre_lds) != 0)
            return fail(SMX_ERR_DEVICE, "cannot raise the prescan kernel's dynamic LDS limit to %zu bytes", P->pre_lds);
        int occ_t = 0, occ_d = 0;
        if (smx_prescan_occupancy(P->hp.S, P->pre_mr, P->pre_nx, P->pre_lds, &occ_t, &occ_d) != 0 || occ_t < 1 || occ_d < 1) { occ_t = 1; occ_d = 8; }
        P->pre_blocks_t = occ_t;
        P->pre_blocks_d = occ_d;
        if (P->env_debug) fprintf(stderr, "[smx] prescan: transpose %d workgroups/CU (lds %zu), DP %d waves/CU\n", occ_t, P->pre_lds, occ_d);
    }
    if (const char *e = getenv("SMX_BLOCKS_PER_CU")) P->blocks_per_cu = P->blocks_per_cu_slots = P->blocks_per_cu_c = std::max(1, atoi(e));
    if (P->env_debug) {
        int occ = -1;
        (void)smx_query_occupancy(&P->hp, P->use64, 0, P->nitems > 0 ? 2 : 0, P->R, 0, P->lds, &occ, P->pre_ok ? 1 : 0);
        fprintf(stderr, "[smx] lean R=%d lds=%zu | slots R=%d lds=%zu | occupancy API (lean): %d blocks/CU, grid multiplier %d, CUs %d\n",
                P->R, P->lds, P->R_slots, P->lds_slots, occ, P->blocks_per_cu, P->n_cu);
        if (P->nitems > 0)
            fprintf(stderr, "[smx] compact lean tiles: R=%d, %d records, lds=%zu, %d blocks/CU (overflow tiles redone dense)\n", P->Rc,
                    P->nitems, P->lds_c, P->blocks_per_cu_c);
    }
    if (getenv("SMX_PHASE_TIMING")) {
        P->phase_grid = P->n_cu * std::max(std::max(P->blocks_per_cu, P->blocks_per_cu_slots), P->blocks_per_cu_c);
        HIP_TRY(hipMalloc((void **)&P->d_phase, (size_t)P->phase_grid * 16 * 8));
        HIP_TRY(hipMemset(P->d_phase, 0, (size_t)P->phase_grid * 16 * 8));
        h.dbg_phase = P->d_phase;
    }
    return SMX_OK;
}

int smx_batch_run_device(const smx_panel *Pc, void *stream, const uint8_t *d_windows, const int32_t *d_lens,
                         uint32_t n_reads, smx_op *d_ops, smx_op *d_extra, uint32_t extra_cap, uint32_t *d_n_extra,
                         uint64_t *d_counts, smx_hit *d_hits, int8_t *d_bdist) {
    smx_panel *P = const_cast<smx_panel *>(Pc);
    if (!P || !d_windows || !d_lens || !d_ops || !d_n_extra || !d_counts) return fail(SMX_ERR_ARG, "null argument");
    if (extra_cap && !d_extra) return fail(SMX_ERR_ARG, "extra_cap without extra buffer");
    if (((uintptr_t)d_windows & 15) != 0) return fail(SMX_ERR_ARG, "window buffer must be 16-byte aligned");
    int rc = ensure_device(P);
    if (rc) return rc;
    if (n_reads == 0) {   // nothing to launch: the count the kernel would have written
        if (hipMemsetAsync(d_n_extra, 0, sizeof(uint32_t), (hipStream_t)stream) != hipSuccess)
            return fail(SMX_ERR_DEVICE, "cannot clear the extra-record counter");
        return SMX_OK;
    }
    // slots mode keeps one result slot per (hit, barcode): needed for the per-barcode distance dump (d_bdist) and for
    // --trim tails when the bit-sliced scan cannot report the tail extent (it can for k <= 3 and at most 32 barcodes
    // per primer).  The hit dump alone (d_hits) comes from whichever kernel the flags select, so that the parity tests
    // see the hit table of the kernel that is benchmarked; its tail_end is defined only where that kernel computes it.
    const bool lean_tails = P->hp.bs_ok && P->hp.kidx < 4 && P->hp.maxB <= 32 && !P->env_no_lean_tails;
    const int use_slots = ((P->hp.trim == SMX_TRIM_TAILS && !lean_tails) || d_bdist || P->env_force_slots) ? 1 : 0;
    const bool compact = !use_slots && P->nitems > 0 && P->pre_ok;
    const int R = use_slots ? P->R_slots : P->R;
    const size_t lds = use_slots ? P->lds_slots : P->lds;
    unsigned *tc = nullptr;
    size_t slot = 0;
    {
        int sl = stream_slot(P, stream, true);
        if (sl < 0) return fail(SMX_ERR_UNSUPPORTED, "one panel launched on more than %d streams at a time", SMX_MAX_STREAMS);
        slot = (size_t)sl;
        tc = P->d_tile_counter + 16 * slot;
    }
    // primer prescan in front of the demux kernel (same stream: ordered)
    const unsigned *d_pre = nullptr, *d_codes2 = nullptr;
    const uint8_t *d_naflag = nullptr;
    uint32_t npad = 0;
    if (P->kev_on) { (void)hipEventRecord(P->kev[0], (hipStream_t)stream); P->kev_pre = P->pre_ok; }
    if (P->pre_ok) {
        npad = (n_reads + smx::PRE_TILE - 1) / smx::PRE_TILE * smx::PRE_TILE;
        const size_t need = (size_t)2 * P->hp.NP * (P->hp.S >> 4) * npad * sizeof(unsigned);
        DevBuf &pb = P->pre_recs[slot], &pp = P->pre_planes[slot], &pm = P->pre_match[slot], &ov = P->ovf[slot];
        const size_t need_planes = (size_t)(npad / smx::PRE_TILE) * (P->hp.S >> 4) * 8 * 64 * 4 * sizeof(unsigned);
        const size_t need_match = P->nitems > 0 ? (size_t)(npad / smx::PRE_TILE) * 2 * P->hp.NP * smx::PRE_G * sizeof(unsigned) : 0;
        const size_t need_ovf = compact ? ((size_t)n_reads / P->Rc + 2) * sizeof(unsigned) : 0;
        DevBuf &pc = P->pre_codes[slot];
        const size_t codes_bytes = (size_t)npad * 2 * (P->hp.S >> 4) * sizeof(unsigned);   // 2-bit codes, then the flag bytes
        const size_t need_codes = codes_bytes + npad;
        if (need > pb.cap || need_planes > pp.cap || need_match > pm.cap || need_ovf > ov.cap || need_codes > pc.cap) {
            if (pb.p || pp.p) (void)hipStreamSynchronize((hipStream_t)stream);   // earlier launches on this stream still use them
            hipError_t pe = pb.ensure(need);
            if (pe == hipSuccess) pe = pp.ensure(need_planes);
            if (pe == hipSuccess) pe = pc.ensure(need_codes);
            if (pe == hipSuccess && need_match) pe = pm.ensure(need_match);
            if (pe == hipSuccess && need_ovf) pe = ov.ensure(need_ovf);
            if (pe != hipSuccess) return fail(SMX_ERR_DEVICE, "prescan buffers: %s", hipGetErrorString(pe));
        }
        const uint32_t ptiles = npad / smx::PRE_TILE;
        // one workgroup per 256-read sub-tile (a few microseconds of work each): the hardware hands them out as slots free up,
        // which balances better than ~2.3 loop iterations per resident workgroup
        const int grid_t = (int)(ptiles * (smx::PRE_G / smx::PRE_SUBG));
        // (DP work items come in groups of 8 tiles x NP primers; grid a multiple of 8: blocks b and b + 8 share an XCD)
        const int grid_d = (int)std::min<uint32_t>(((ptiles + 7) / 8) * 8 * (uint32_t)P->hp.NP, (uint32_t)(P->n_cu * P->pre_blocks_d));
        d_codes2 = (const unsigned *)pc.p;
        d_naflag = (const uint8_t *)pc.p + codes_bytes;
        int pe = smx_launch_prescan(&P->pre, P->pre_mr, P->pre_nx, grid_t, P->pre_lds, grid_d, stream, d_windows, d_lens, n_reads,
                                    P->hp.wstride, (unsigned *)pp.p, (unsigned *)pb.p, P->nitems > 0 ? (unsigned *)pm.p : nullptr,
                                    P->kev_on ? (void *)P->kev[1] : nullptr, (unsigned *)pc.p, (uint8_t *)pc.p + codes_bytes);
        if (pe != 0) return fail(SMX_ERR_DEVICE, "prescan kernel launch failed: %s", hipGetErrorString((hipError_t)pe));
        d_pre = (const unsigned *)pb.p;
        if (P->kev_on) (void)hipEventRecord(P->kev[2], (hipStream_t)stream);
    }
    uint32_t tiles = (n_reads + R - 1) / R;
    // (batches in flight on several streams share the CUs' workgroup slots: each demux launch takes its part of them, so
    // that the next batch's memory-bound and VALU-bound prescan kernels run beside this batch's latency-bound demux kernel)
    auto slots_of = [&](int blocks) {
        int per = std::max(1, (blocks + P->share - 1) / P->share);
        // two launches side by side must not claim more slots than the CU has: nothing of the next batch's prescan kernels
        // would fit beside them (five slots, two streams: 3 + 3 measured 0.335 ms per step, 2 + 2 0.306-0.319)
        if (P->share == 2 && per * 2 > blocks) per = std::max(1, blocks / 2);
        return (uint32_t)(P->n_cu * per);
    };
    int grid = (int)std::min<uint32_t>(tiles, slots_of(use_slots ? P->blocks_per_cu_slots : P->blocks_per_cu));
    int e;
    if (compact) {
        // compact launch over all reads, then the dense launch over the reads of the tiles it put on the overflow list
        // (usually none: its workgroups find an empty list and leave)
        smx::DemuxAux ax = {(const unsigned *)P->pre_match[slot].p, (unsigned *)P->ovf[slot].p, P->nitems, 0, P->Rc, 1, d_codes2, d_naflag};
        const uint32_t ctiles = (n_reads + P->Rc - 1) / P->Rc;
        const int cgrid = (int)std::min<uint32_t>(ctiles, slots_of(P->blocks_per_cu_c));
        e = smx_launch_demux(&P->hp, P->use64, P->Rc, cgrid, P->lds_c, stream, d_windows, d_lens, n_reads, d_ops, d_extra,
                             extra_cap, d_n_extra, d_counts, d_hits, d_bdist, tc, 0, d_pre, npad, &ax);
        if (e == 0 && P->env_debug_overflow) {   // diagnostic: how many compact tiles went on the overflow list
            unsigned n_ovf = 0;
            (void)hipMemcpyAsync(&n_ovf, tc + 1, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream);
            (void)hipStreamSynchronize((hipStream_t)stream);
            fprintf(stderr, "[smx] compact launch: %u of %u tiles left to the redo launch\n", n_ovf, ctiles);
        }
        if (e == 0) {
            // the redo launch usually finds an empty list: one workgroup per CU is enough to start with (its workgroups
            // loop over the list), and an empty 256-workgroup launch costs less than an empty full-residency one
            smx::DemuxAux rx = {nullptr, (unsigned *)P->ovf[slot].p, 0, 1, P->Rc, 0, d_codes2, d_naflag};
            grid = std::min(grid, P->n_cu);
            e = smx_launch_demux(&P->hp, P->use64, R, grid, lds, stream, d_windows, d_lens, n_reads, d_ops, d_extra,
                                 extra_cap, d_n_extra, d_counts, d_hits, d_bdist, tc, 0, d_pre, npad, &rx);
        }
    } else {
        smx::DemuxAux dx = {nullptr, nullptr, 0, 0, 0, 0, d_codes2, d_naflag};
        e = smx_launch_demux(&P->hp, P->use64, R, grid, lds, stream, d_windows, d_lens, n_reads, d_ops, d_extra,
                             extra_cap, d_n_extra, d_counts, d_hits, d_bdist, tc, use_slots, d_pre, npad, &dx);
    }
    if (e != 0) return fail(SMX_ERR_DEVICE, "demux kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    if (P->kev_on) (void)hipEventRecord(P->kev[3], (hipStream_t)stream);
    return SMX_OK;
}

int smx_unpack_windows_device(const smx_panel *Pc, void *stream, const uint8_t *d_packed, uint32_t n_reads, uint8_t *d_windows) {
    smx_panel *P = const_cast<smx_panel *>(Pc);
    if (!P || !d_packed || !d_windows) return fail(SMX_ERR_ARG, "null argument");
    if (((uintptr_t)d_packed & 3) != 0 || ((uintptr_t)d_windows & 15) != 0) return fail(SMX_ERR_ARG, "window buffers must be 16-byte aligned");
    int rc = ensure_device(P);
    if (rc) return rc;
    int e = smx_launch_unpack_windows(stream, d_packed, d_windows, n_reads, P->hp.S, (int)smx_packed_stride_for(P->hp.S), P->hp.wstride, P->n_cu);
    if (e != 0) return fail(SMX_ERR_DEVICE, "unpack kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    return SMX_OK;
}

int smx_panel_set_streams(smx_panel *P, int n_streams) {
    if (!P || n_streams < 1 || n_streams > SMX_MAX_STREAMS) return fail(SMX_ERR_ARG, "n_streams outside 1..%d", SMX_MAX_STREAMS);
    P->share = n_streams;
    return SMX_OK;
}

int smx_debug_kernel_times(smx_panel *P, int enable, float ms[3]) {
    if (!P) return fail(SMX_ERR_ARG, "null argument");
    if (ms) {
        ms[0] = ms[1] = ms[2] = 0.f;
        if (P->kev_on && P->kev[3]) {
            HIP_TRY(hipEventSynchronize(P->kev[3]));
            if (P->kev_pre) {
                HIP_TRY(hipEventElapsedTime(&ms[0], P->kev[0], P->kev[1]));
                HIP_TRY(hipEventElapsedTime(&ms[1], P->kev[1], P->kev[2]));
                HIP_TRY(hipEventElapsedTime(&ms[2], P->kev[2], P->kev[3]));
            } else {
                HIP_TRY(hipEventElapsedTime(&ms[2], P->kev[0], P->kev[3]));
            }
        }
    }
    if (enable && !P->kev[0]) {
        int rc = ensure_device(P);
        if (rc) return rc;
        for (auto &e : P->kev) HIP_TRY(hipEventCreate(&e));
    }
    P->kev_on = enable != 0;
    return SMX_OK;
}

int smx_batch_run(const smx_panel *Pc, const uint8_t *windows, const int32_t *lens, uint32_t n_reads, smx_op *ops,
                  smx_op *extra, uint32_t extra_cap, uint32_t *n_extra, uint64_t *counts, smx_hit *hits, int8_t *bdist) {
    smx_panel *P = const_cast<smx_panel *>(Pc);
    if (!P || !windows || !lens || !ops || !n_extra || !counts) return fail(SMX_ERR_ARG, "null argument");
    int rc = ensure_device(P);
    if (rc) return rc;
    *n_extra = 0;
    if (n_reads == 0) return SMX_OK;
    const size_t wbytes = (size_t)n_reads * P->hp.wstride, ncnt = smx_counts_len(P);
    const size_t hbytes = hits ? (size_t)n_reads * smx_hits_per_read(P) * sizeof(smx_hit) : 0;
    const size_t bbytes = bdist ? (size_t)n_reads * smx_bdist_per_read(P) : 0;
    std::lock_guard<std::mutex> guard(P->ws_mutex);
#define TRY_C(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(SMX_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); } while (0)
    TRY_C(P->ws[0].ensure(wbytes));
    TRY_C(P->ws[1].ensure((size_t)n_reads * 4));
    TRY_C(P->ws[2].ensure((size_t)n_reads * sizeof(smx_op)));
    TRY_C(P->ws[3].ensure(std::max<size_t>((size_t)extra_cap * sizeof(smx_op), 32)));
    TRY_C(P->ws[4].ensure(16));
    TRY_C(P->ws[5].ensure(ncnt * 8));
    if (hits) TRY_C(P->ws[6].ensure(hbytes));
    if (bdist) TRY_C(P->ws[7].ensure(bbytes));
    unsigned char *dw = (unsigned char *)P->ws[0].p, *dl = (unsigned char *)P->ws[1].p, *dop = (unsigned char *)P->ws[2].p,
                  *dex = (unsigned char *)P->ws[3].p, *dn = (unsigned char *)P->ws[4].p, *dc = (unsigned char *)P->ws[5].p,
                  *dh = hits ? (unsigned char *)P->ws[6].p : nullptr, *db = bdist ? (unsigned char *)P->ws[7].p : nullptr;
    TRY_C(hipMemcpy(dw, windows, wbytes, hipMemcpyHostToDevice));
    TRY_C(hipMemcpy(dl, lens, (size_t)n_reads * 4, hipMemcpyHostToDevice));
    TRY_C(hipMemset(dn, 0, 16));
    TRY_C(hipMemset(dc, 0, ncnt * 8));
    rc = smx_batch_run_device(P, nullptr, dw, (const int32_t *)dl, n_reads, (smx_op *)dop, (smx_op *)dex, extra_cap,
                              (uint32_t *)dn, (uint64_t *)dc, (smx_hit *)dh, (int8_t *)db);
    if (rc) return rc;
    {
        hipError_t se = hipDeviceSynchronize();
        if (se != hipSuccess) {   // an aborted launch leaves the self re-arming counters in an unknown state
            stream_reset_counters(P, nullptr);
            return fail(SMX_ERR_DEVICE, "demux kernel failed: %s", hipGetErrorString(se));
        }
    }
    std::vector<uint64_t> c(ncnt);
    TRY_C(hipMemcpy(ops, dop, (size_t)n_reads * sizeof(smx_op), hipMemcpyDeviceToHost));
    TRY_C(hipMemcpy(n_extra, dn, 4, hipMemcpyDeviceToHost));
    TRY_C(hipMemcpy(c.data(), dc, ncnt * 8, hipMemcpyDeviceToHost));
    if (extra && extra_cap)
        TRY_C(hipMemcpy(extra, dex, (size_t)std::min<uint32_t>(*n_extra, extra_cap) * sizeof(smx_op), hipMemcpyDeviceToHost));
    if (hits) TRY_C(hipMemcpy(hits, dh, hbytes, hipMemcpyDeviceToHost));
    if (bdist) TRY_C(hipMemcpy(bdist, db, bbytes, hipMemcpyDeviceToHost));
#undef TRY_C
    if (c[SMX_CNT_TOTAL] != n_reads) {
        // every read is counted exactly once by the tile that scored it: anything else means tiles were skipped or
        // repeated (a tile queue that did not start at zero) and the records above cannot be trusted
        stream_reset_counters(P, nullptr);
        return fail(SMX_ERR_DEVICE, "demux kernel processed %llu of %u reads (tile queue out of step); counters reset",
                    (unsigned long long)c[SMX_CNT_TOTAL], n_reads);
    }
    for (size_t i = 0; i < ncnt; i++) counts[i] += c[i];
    if (c[SMX_CNT_OVERFLOW]) return fail(SMX_ERR_OVERFLOW, "%llu read(s) produced more than 65535 write operations",
                                         (unsigned long long)c[SMX_CNT_OVERFLOW]);
    if (*n_extra > extra_cap) return fail(SMX_ERR_OVERFLOW, "extra buffer too small: need %u records, have %u", *n_extra, extra_cap);
    return SMX_OK;
}

int smx_align(const char *query, int qlen, const char *target, int tlen, int k, int mode, int *dist, int *starts,
              int *ends, int cap, int *nloc) {
    if (!query || !target || !dist || !nloc) return fail(SMX_ERR_ARG, "null argument");
    if (qlen < 1 || qlen > 64) return fail(SMX_ERR_UNSUPPORTED, "query length %d outside 1..64", qlen);
    if (mode != 0 && mode != 1) return fail(SMX_ERR_UNSUPPORTED, "mode %d (0 = HW, 1 = SHW)", mode);
    if (tlen < 1) return fail(SMX_ERR_UNSUPPORTED, "empty target");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(SMX_ERR_DEVICE, "libsmx has no CPU path: no HIP device");
    unsigned long long peq[32];
    std::string bad, q(query, qlen), rq(q.rbegin(), q.rend());
    if (!build_peq(q.data(), qlen, peq, &bad) || !build_peq(rq.data(), qlen, peq + 16, &bad))
        return fail(SMX_ERR_UNSUPPORTED, "%s", bad.c_str());
    std::vector<unsigned char> codes(tlen);
    for (int i = 0; i < tlen; i++) codes[i] = (unsigned char)code_of((unsigned char)target[i]);
    unsigned char *d = nullptr;
    size_t o_codes = 256, o_flag = o_codes + ((tlen + 15) & ~15), o_starts = o_flag + ((tlen + 15) & ~15),
           o_dist = o_starts + (size_t)tlen * 4, total = o_dist + 16;
    HIP_TRY(hipMalloc((void **)&d, total));
    hipError_t e = hipMemcpy(d, peq, 256, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o_codes, codes.data(), tlen, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = (hipError_t)smx_launch_align(nullptr, (unsigned long long *)d, (unsigned long long *)d + 16, qlen, d + o_codes,
                                         tlen, k, mode, (int *)(d + o_dist), d + o_flag, (int *)(d + o_starts));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    std::vector<unsigned char> flag(tlen);
    std::vector<int> st(tlen);
    if (e == hipSuccess) e = hipMemcpy(dist, d + o_dist, 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(flag.data(), d + o_flag, tlen, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(st.data(), d + o_starts, (size_t)tlen * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(SMX_ERR_DEVICE, "smx_align: %s", hipGetErrorString(e));
    int cnt = 0;
    if (*dist >= 0)
        for (int j = 0; j < tlen; j++)
            if (flag[j]) {
                if (cnt < cap && starts && ends) { starts[cnt] = st[j]; ends[cnt] = j; }
                cnt++;
            }
    *nloc = cnt;
    return SMX_OK;
}

// ---- batched alignments: grow-only device workspace shared by all calls (serialised)
namespace {
std::mutex g_align_mutex;
DevBuf g_align_ws[12];
}

int smx_align_batch(const char *queries, const uint32_t *qoff, uint32_t n_queries, const char *targets, const uint64_t *toff,
                    const uint32_t *qidx, const int32_t *k, const uint8_t *mode, uint32_t n, int32_t *dist, int32_t *nloc,
                    int32_t *starts, int32_t *ends, uint32_t cap) {
    if (!queries || !qoff || !targets || !toff || !qidx || !k || !mode || !dist || !nloc || (cap && (!starts || !ends)))
        return fail(SMX_ERR_ARG, "null argument");
    if (n == 0) return SMX_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(SMX_ERR_DEVICE, "libsmx has no CPU path: no HIP device");
    std::vector<unsigned long long> qpeq((size_t)n_queries * 32);
    std::vector<int> qlen(n_queries);
    std::string bad;
    for (uint32_t q = 0; q < n_queries; q++) {
        const int m = (int)(qoff[q + 1] - qoff[q]);
        if (m < 1 || m > 64) return fail(SMX_ERR_UNSUPPORTED, "query %u: length %d outside 1..64", q, m);
        std::string s(queries + qoff[q], m), r(s.rbegin(), s.rend());
        if (!build_peq(s.data(), m, &qpeq[(size_t)q * 32], &bad) || !build_peq(r.data(), m, &qpeq[(size_t)q * 32 + 16], &bad))
            return fail(SMX_ERR_UNSUPPORTED, "query %u: %s", q, bad.c_str());
        qlen[q] = m;
    }
    const uint64_t tbytes = toff[n];
    std::vector<unsigned char> codes(tbytes);
    for (uint32_t i = 0; i < n; i++) {
        if (toff[i + 1] <= toff[i]) return fail(SMX_ERR_UNSUPPORTED, "alignment %u: empty target", i);
        if (qidx[i] >= n_queries || mode[i] > 1) return fail(SMX_ERR_ARG, "alignment %u: bad query index or mode", i);
        if (k[i] < 0 || k[i] > 250) return fail(SMX_ERR_ARG, "alignment %u: bad max distance", i);
    }
    for (uint64_t j = 0; j < tbytes; j++) codes[j] = (unsigned char)code_of((unsigned char)targets[j]);
    std::lock_guard<std::mutex> guard(g_align_mutex);
    DevBuf *B = g_align_ws;
    const size_t sz[11] = {qpeq.size() * 8, qlen.size() * 4, (size_t)n * 4, (size_t)tbytes, ((size_t)n + 1) * 8, (size_t)n * 4, (size_t)n,
                           (size_t)tbytes, (size_t)n * 4, (size_t)n * 4, (size_t)n * cap * 4};
    for (int b = 0; b < 11; b++) HIP_TRY(B[b].ensure(std::max<size_t>(sz[b], 16)));
    HIP_TRY(B[11].ensure(std::max<size_t>(sz[10], 16)));
    HIP_TRY(hipMemcpy(B[0].p, qpeq.data(), sz[0], hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(B[1].p, qlen.data(), sz[1], hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(B[2].p, qidx, sz[2], hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(B[3].p, codes.data(), sz[3], hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(B[4].p, toff, sz[4], hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(B[5].p, k, sz[5], hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(B[6].p, mode, sz[6], hipMemcpyHostToDevice));
    int e = smx_launch_align_batch(nullptr, (const unsigned long long *)B[0].p, (const int *)B[1].p, (const unsigned *)B[2].p,
                                   (const unsigned char *)B[3].p, (const unsigned long long *)B[4].p, (const int *)B[5].p,
                                   (const unsigned char *)B[6].p, n, (unsigned char *)B[7].p, (int *)B[8].p, (int *)B[9].p,
                                   (int *)B[10].p, (int *)B[11].p, cap);
    if (e != 0) return fail(SMX_ERR_DEVICE, "alignment kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(dist, B[8].p, sz[8], hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(nloc, B[9].p, sz[9], hipMemcpyDeviceToHost));
    if (cap) {
        HIP_TRY(hipMemcpy(starts, B[10].p, sz[10], hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(ends, B[11].p, sz[10], hipMemcpyDeviceToHost));
    }
    return SMX_OK;
}

// ---- lanes: asynchronous host-buffer path (pinned staging, one stream per lane)
struct smx_lane {
    smx_panel *P = nullptr;
    uint32_t cap = 0, n = 0;
    hipStream_t stream = nullptr;
    // pinned host staging
    uint8_t *h_windows = nullptr;
    int32_t *h_lens = nullptr;
    smx_op *h_ops = nullptr, *h_extra = nullptr;
    uint64_t *h_counts = nullptr;   // counts vector followed by one word holding n_extra
    // device
    uint8_t *d_windows = nullptr;
    uint8_t *d_packed = nullptr;    // 4-bit windows as they arrive over PCIe (smx_lane_submit_packed)
    int32_t *d_lens = nullptr;
    smx_op *d_ops = nullptr, *d_extra = nullptr;
    uint64_t *d_counts = nullptr;   // same layout as h_counts
    bool busy = false;
};

void smx_lane_destroy(smx_lane *L) {
    if (!L) return;
    if (L->stream) {
        (void)hipStreamSynchronize(L->stream);
        if (L->P) stream_release(L->P, L->stream);   // the panel's per-stream slot (launch counters, prescan buffers) is free again
    }
    if (L->h_windows) (void)hipHostFree(L->h_windows);
    if (L->h_lens) (void)hipHostFree(L->h_lens);
    if (L->h_ops) (void)hipHostFree(L->h_ops);
    if (L->h_extra) (void)hipHostFree(L->h_extra);
    if (L->h_counts) (void)hipHostFree(L->h_counts);
    if (L->d_windows) (void)hipFree(L->d_windows);
    if (L->d_packed) (void)hipFree(L->d_packed);
    if (L->d_lens) (void)hipFree(L->d_lens);
    if (L->d_ops) (void)hipFree(L->d_ops);
    if (L->d_extra) (void)hipFree(L->d_extra);
    if (L->d_counts) (void)hipFree(L->d_counts);
    if (L->stream) (void)hipStreamDestroy(L->stream);
    delete L;
}

int smx_lane_create(const smx_panel *Pc, uint32_t max_reads, smx_lane **out) {
    smx_panel *P = const_cast<smx_panel *>(Pc);
    if (!P || !out || max_reads == 0) return fail(SMX_ERR_ARG, "null argument");
    int rc = ensure_device(P);
    if (rc) return rc;
    smx_lane *L = new smx_lane();
    L->P = P;
    L->cap = max_reads;
    const size_t wb = (size_t)max_reads * P->hp.wstride, ob = (size_t)max_reads * sizeof(smx_op), cb = (smx_counts_len(P) + 1) * 8;
#define LANE_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { smx_lane_destroy(L); return fail(SMX_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); } } while (0)
    LANE_TRY(hipStreamCreateWithFlags(&L->stream, hipStreamNonBlocking));
    LANE_TRY(hipHostMalloc((void **)&L->h_windows, wb, hipHostMallocDefault));
    LANE_TRY(hipHostMalloc((void **)&L->h_lens, (size_t)max_reads * 4, hipHostMallocDefault));
    LANE_TRY(hipHostMalloc((void **)&L->h_ops, ob, hipHostMallocDefault));
    LANE_TRY(hipHostMalloc((void **)&L->h_extra, ob, hipHostMallocDefault));
    LANE_TRY(hipHostMalloc((void **)&L->h_counts, cb, hipHostMallocDefault));
    LANE_TRY(hipMalloc((void **)&L->d_windows, wb));
    LANE_TRY(hipMalloc((void **)&L->d_packed, (size_t)max_reads * smx_packed_stride_for(P->hp.S)));
    LANE_TRY(hipMalloc((void **)&L->d_lens, (size_t)max_reads * 4));
    LANE_TRY(hipMalloc((void **)&L->d_ops, ob));
    LANE_TRY(hipMalloc((void **)&L->d_extra, ob));
    LANE_TRY(hipMalloc((void **)&L->d_counts, cb));
#undef LANE_TRY
    *out = L;
    return SMX_OK;
}

uint8_t *smx_lane_windows(smx_lane *L) { return L ? L->h_windows : nullptr; }
int32_t *smx_lane_lens(smx_lane *L) { return L ? L->h_lens : nullptr; }

static int lane_submit(smx_lane *L, uint32_t n_reads, bool packed);
int smx_lane_submit(smx_lane *L, uint32_t n_reads) { return lane_submit(L, n_reads, false); }
int smx_lane_submit_packed(smx_lane *L, uint32_t n_reads) { return lane_submit(L, n_reads, true); }

static int lane_submit(smx_lane *L, uint32_t n_reads, bool packed) {
    if (!L) return fail(SMX_ERR_ARG, "null argument");
    if (L->busy) return fail(SMX_ERR_ARG, "lane already has a batch in flight: smx_lane_wait first");
    HIP_TRY(hipSetDevice(L->P->device));   // lanes are driven from reader / writer threads: device selection is per thread
    if (n_reads > L->cap) return fail(SMX_ERR_ARG, "batch of %u reads exceeds the lane capacity %u", n_reads, L->cap);
    smx_panel *P = L->P;
    const size_t ncnt = smx_counts_len(P);
    L->n = n_reads;
    HIP_TRY(hipMemsetAsync(L->d_counts, 0, (ncnt + 1) * 8, L->stream));
    if (n_reads) {
        if (packed) {   // the staging holds 4-bit windows: half the bytes over the link, unpacked into the ASCII layout on the device
            const size_t ps = smx_packed_stride_for(P->hp.S);
            HIP_TRY(hipMemcpyAsync(L->d_packed, L->h_windows, (size_t)n_reads * ps, hipMemcpyHostToDevice, L->stream));
            int ue = smx_launch_unpack_windows(L->stream, L->d_packed, L->d_windows, n_reads, P->hp.S, (int)ps, P->hp.wstride, P->n_cu);
            if (ue != 0) return fail(SMX_ERR_DEVICE, "unpack kernel launch failed: %s", hipGetErrorString((hipError_t)ue));
        } else
            HIP_TRY(hipMemcpyAsync(L->d_windows, L->h_windows, (size_t)n_reads * P->hp.wstride, hipMemcpyHostToDevice, L->stream));
        HIP_TRY(hipMemcpyAsync(L->d_lens, L->h_lens, (size_t)n_reads * 4, hipMemcpyHostToDevice, L->stream));
        int rc = smx_batch_run_device(P, L->stream, L->d_windows, L->d_lens, n_reads, L->d_ops, L->d_extra, L->cap,
                                      (uint32_t *)(L->d_counts + ncnt), L->d_counts, nullptr, nullptr);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(L->h_ops, L->d_ops, (size_t)n_reads * sizeof(smx_op), hipMemcpyDeviceToHost, L->stream));
        // extra records are rare: the count is not known on the host yet, so a fixed small prefix travels with the batch and
        // smx_lane_wait fetches the rest if there is more
        HIP_TRY(hipMemcpyAsync(L->h_extra, L->d_extra, (size_t)std::min<uint32_t>(L->cap, std::max<uint32_t>(4096u, n_reads / 64)) * sizeof(smx_op),
                               hipMemcpyDeviceToHost, L->stream));
    }
    HIP_TRY(hipMemcpyAsync(L->h_counts, L->d_counts, (ncnt + 1) * 8, hipMemcpyDeviceToHost, L->stream));
    L->busy = true;
    return SMX_OK;
}

int smx_lane_wait(smx_lane *L, const smx_op **ops, const smx_op **extra, uint32_t *n_extra, uint64_t *counts) {
    if (!L || !n_extra || !counts) return fail(SMX_ERR_ARG, "null argument");
    if (!L->busy) return fail(SMX_ERR_ARG, "lane has no batch in flight");
    smx_panel *P = L->P;
    HIP_TRY(hipSetDevice(P->device));
    const size_t ncnt = smx_counts_len(P);
    L->busy = false;
    {
        hipError_t se = hipStreamSynchronize(L->stream);
        if (se != hipSuccess) {
            stream_reset_counters(P, L->stream);
            return fail(SMX_ERR_DEVICE, "lane batch failed: %s", hipGetErrorString(se));
        }
    }
    const uint32_t ne = (uint32_t)L->h_counts[ncnt];
    *n_extra = ne;
    if (L->h_counts[SMX_CNT_TOTAL] != L->n) {
        stream_reset_counters(P, L->stream);
        return fail(SMX_ERR_DEVICE, "demux kernel processed %llu of %u reads (tile queue out of step); counters reset",
                    (unsigned long long)L->h_counts[SMX_CNT_TOTAL], L->n);
    }
    if (ne > L->cap) return fail(SMX_ERR_OVERFLOW, "extra buffer too small: need %u records, have %u", ne, L->cap);
    const uint32_t sent = std::min<uint32_t>(L->cap, std::max<uint32_t>(4096u, L->n / 64));
    if (ne > sent)   // the rest of the extra records
        HIP_TRY(hipMemcpy(L->h_extra + sent, L->d_extra + sent, (size_t)(ne - sent) * sizeof(smx_op), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < ncnt; i++) counts[i] += L->h_counts[i];
    if (ops) *ops = L->h_ops;
    if (extra) *extra = L->h_extra;
    if (L->h_counts[SMX_CNT_OVERFLOW])
        return fail(SMX_ERR_OVERFLOW, "%llu read(s) produced more than 65535 write operations", (unsigned long long)L->h_counts[SMX_CNT_OVERFLOW]);
    return SMX_OK;
}

// ---- RCCL over xGMI: one communicator per process (one process per GPU)
int smx_comm_unique_id(uint8_t id[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId u;
    ncclResult_t r = ncclGetUniqueId(&u);
    if (r != ncclSuccess) return fail(SMX_ERR_DEVICE, "ncclGetUniqueId: %s", ncclGetErrorString(r));
    memcpy(id, &u, 128);
    return SMX_OK;
}

int smx_comm_init(const uint8_t id[128], int n_ranks, int rank, void **comm_out) {
    if (!id || !comm_out) return fail(SMX_ERR_ARG, "null argument");
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclComm_t comm;
    ncclResult_t r = ncclCommInitRank(&comm, n_ranks, u, rank);
    if (r != ncclSuccess) return fail(SMX_ERR_DEVICE, "ncclCommInitRank: %s", ncclGetErrorString(r));
    *comm_out = (void *)comm;
    return SMX_OK;
}

int smx_counts_allreduce(uint64_t *d_counts, size_t n, void *comm, void *stream) {
    if (!d_counts || !comm) return fail(SMX_ERR_ARG, "null argument");
    ncclResult_t r = ncclAllReduce(d_counts, d_counts, n, ncclUint64, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
    if (r != ncclSuccess) return fail(SMX_ERR_DEVICE, "ncclAllReduce: %s", ncclGetErrorString(r));
    return SMX_OK;
}

void smx_comm_destroy(void *comm) {
    if (comm) (void)ncclCommDestroy((ncclComm_t)comm);
}

}  // extern "C"
