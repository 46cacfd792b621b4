// CPU simulation of the primer prescan (specimux_amd/csrc/smx_prescan_core.h): the same host/device functions the
// gfx950 kernel runs, with lanes as loop indices and LDS as an array, checked against a plain O(mn) DP per
// (read, primer, end).  Built and run by tests/test_prescan_cpu.py (g++, no GPU).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "smx_prescan_core.h"

using namespace smx;

static bool eq_iupac(unsigned char p, unsigned char t) {
    static const char *pairs[] = {"YC", "YT", "RA", "RG", "NA", "NC", "NG", "NT", "WA", "WT", "MA", "MC", "SC", "SG",
                                  "KG", "KT", "BC", "BG", "BT", "DA", "DG", "DT", "HA", "HC", "HT", "VA", "VC", "VG"};
    if (p == t) return true;
    for (const char *q : pairs)
        if ((q[0] == p && q[1] == t) || (q[1] == p && q[0] == t)) return true;
    return false;
}

static char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; }

// scalar reference: HW alignment, the score of every column
static std::vector<int> reference_scores(const std::string &pat, const std::string &text) {
    const int m = (int)pat.size(), n = (int)text.size();
    std::vector<int> col(m + 1), score(n);
    for (int i = 0; i <= m; i++) col[i] = i;
    for (int j = 0; j < n; j++) {
        int diag = col[0];
        col[0] = 0;
        for (int i = 1; i <= m; i++) {
            int v = std::min(std::min(col[i] + 1, col[i - 1] + 1), diag + (eq_iupac((unsigned char)pat[i - 1], (unsigned char)text[j]) ? 0 : 1));
            diag = col[i];
            col[i] = v;
        }
        score[j] = col[m];
    }
    return score;
}

int main(int argc, char **argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 80;
    const unsigned seed = argc > 2 ? (unsigned)atoi(argv[2]) : 1;
    const int CH = S / 16, ppr = 2 * CH;
    std::mt19937 rng(seed);
    // patterns = primer reverse complements as the kernel searches them; one degenerate, one short, one 31-mer
    std::vector<std::string> pats = {"TTACTTCCTCTAAATGACCAAG", "GCATATCAATAAGCGGAGGA", "GTGARTCATCGAATCTTTG", "ACGTNACGTYACGKA",
                                     "ACGTTGCATGCCATGACTGACTAGCTAGCAT"};
    std::vector<int> lens, ks = {7, 6, 5, 3, 9};
    if (argc > 3) { pats.erase(pats.begin() + 3); ks.erase(ks.begin() + 3); }   // one degenerate letter only: nsym = 5
    std::vector<const char *> pp;
    for (auto &s : pats) { lens.push_back((int)s.size()); pp.push_back(s.c_str()); }
    const int NP = (int)pats.size();
    PreDesc D;
    memset(&D, 0, sizeof(D));
    if (!prescan_build_desc(&D, NP, S, pp.data(), lens.data(), ks.data(), eq_iupac)) { printf("desc failed\n"); return 2; }
    // a tile of reads: windows [read][2 * S] ASCII (head | tail)
    const int n = PRE_TILE;
    std::vector<unsigned char> win((size_t)n * 2 * S);
    auto rnd_base = [&] { return "ACGT"[rng() & 3]; };
    std::vector<std::string> heads(n), tails(n);
    std::vector<int> rlen(n);
    for (int r = 0; r < n; r++) {
        std::string h(S, 'A'), t(S, 'A');
        for (auto &c : h) c = rnd_base();
        for (auto &c : t) c = rnd_base();
        // plant mutated copies (in the orientation the scan sees them): end B sees the tail as is, end A revcomp(head)
        for (int e = 0; e < 2; e++) {
            if (rng() % 4 == 0) continue;
            const std::string &pat = pats[rng() % NP];
            std::string cp;
            for (char c : pat) {   // instantiate degenerate letters, then mutate
                char b = c;
                if (!strchr("ACGT", c)) { do { b = rnd_base(); } while (!eq_iupac((unsigned char)c, (unsigned char)b)); }
                unsigned u = rng() % 100;
                if (u < 6) b = rnd_base();
                else if (u < 9) continue;            // deletion
                else if (u < 12) cp.push_back(rnd_base());   // insertion
                cp.push_back(b);
            }
            std::string tgt = e ? t : h;
            if ((int)cp.size() >= S) continue;
            int pos = (int)(rng() % (S - cp.size() + 1));
            if (rng() % 8 == 0) pos = S - (int)cp.size();   // flush with the window end
            if (rng() % 8 == 0) pos = 0;
            if (e) { tgt.replace(pos, cp.size(), cp); t = tgt; }
            else {   // the scan sees revcomp(head): write revcomp(cp) into the head
                std::string rc(cp.rbegin(), cp.rend());
                for (auto &c : rc) c = comp(c);
                tgt.replace(pos, cp.size(), rc);
                h = tgt;
            }
        }
        if (r % 97 == 0) { h.assign(S, 'A'); t.assign(S, 'T'); }   // low-complexity: many optimal ends (overflow path)
        // every 9th read is shorter than the window: both stored windows hold its min(len, S) bases, zero padded
        int L = S + 100;
        if (r % 9 == 0) { L = (int)(rng() % (S + 1)); if (r % 27 == 0) L = S - 1 - (int)(rng() % 3); if (L < 0) L = 0; }
        rlen[r] = L;
        if (L < S) { h.resize(L); t.resize(L); }
        heads[r] = h; tails[r] = t;
        memset(&win[(size_t)r * 2 * S], 0, 2 * S);
        memcpy(&win[(size_t)r * 2 * S], h.data(), h.size());
        memcpy(&win[(size_t)r * 2 * S + S], t.data(), t.size());
    }
    // ---- the kernel's phases on the host: four sub-tiles of 256 reads, each staged, transposed and written out on its own
    // some reads get a character outside upper-case ACGT (the flag byte must say so; their codes are then unused)
    std::vector<int> dirty(n, 0);
    for (int r = 5; r < n; r += 41) {
        if (rlen[r] < 1) continue;
        const int pos = (int)(rng() % (unsigned)std::min(rlen[r], S));
        const bool tail = (r & 1) != 0;
        win[(size_t)r * 2 * S + (tail ? S : 0) + pos] = (unsigned char)"NacgtRY-"[rng() % 8];
        dirty[r] = 1;
    }
    std::vector<unsigned> gpl((size_t)CH * 8 * 64 * 4, 0u);   // the tile's planes in the HBM layout
    std::vector<unsigned> codes2((size_t)n * ppr, 0u);
    std::vector<unsigned char> naflag(n, 0);
    long bad = 0;
    for (int sub = 0; sub < PRE_G / PRE_SUBG; sub++) {
        std::vector<unsigned> planes((size_t)ppr * PRE_CS + 64, 0u);
        const int r0 = sub * PRE_SUBG * 32;
        for (int q = 0; q < PRE_SUBG * 32 * ppr; q++) {   // phase 1
            const int rs = q / ppr, c = q % ppr, read = r0 + rs;
            unsigned w[4];
            memcpy(w, &win[(size_t)read * 2 * S + 16 * c], 16);
            if (acgt_mismatch(w[0]) | acgt_mismatch(w[1]) | acgt_mismatch(w[2]) | acgt_mismatch(w[3])) naflag[read] = 1;
            if (c < CH && rlen[read] < S) prescan_short_head_piece(&win[(size_t)read * 2 * S], c, S, rlen[read], w);
            const unsigned z = prescan_store_piece(planes.data(), rs, c, w[0], w[1], w[2], w[3]);
            int end, chunk;
            const unsigned zz = codes2_from_piece(z, c, CH, &end, &chunk);
            codes2[codes2_word((size_t)read, CH, end, chunk)] = zz;
        }
        for (int b = 0; b < PRE_SUBG * ppr; b++) {   // phase 2
            const int g = b / ppr, c = b % ppr;
            unsigned o[32];
            prescan_transpose_block(planes.data(), g, c, CH, o);
            for (int d = 0; d < 32; d++)
                gpl[prescan_plane_word(prescan_block_chunk(c, CH), prescan_block_lane(sub * PRE_SUBG + g, c, CH), d)] = o[d];
        }
    }
    // the row-major codes and the flag byte, read by read: column t of end X in DP order = the text the DP sees
    for (int read = 0; read < n; read++) {
        const bool expect_flag = dirty[read] || rlen[read] < S;   // (zero padding of a short read is not ACGT either)
        if ((naflag[read] != 0) != expect_flag) { if (bad < 10) printf("FLAG read %d: %d, expected %d\n", read, naflag[read], (int)expect_flag); bad++; }
        if (dirty[read]) {   // back to plain ACGT for the alignment checks below (the planes of such reads are not consumed)
            continue;
        }
        for (int X = 0; X < 2; X++) {
            std::string text;
            if (X) text = tails[read];
            else { text.assign(heads[read].rbegin(), heads[read].rend()); for (auto &c : text) c = comp(c); }
            for (int t = 0; t < (int)text.size(); t++) {
                const unsigned z = codes2[codes2_word((size_t)read, CH, X, t >> 4)];
                const int tt = t & 15, kq = tt >> 2, i = tt & 3;
                const char got = "ACTG"[(z >> (8 * i + 2 * kq)) & 3u];
                if (got != text[t]) { if (bad < 10) printf("CODES2 read %d end %d column %d: %c, expected %c\n", read, X, t, got, text[t]); bad++; break; }
            }
        }
    }
    std::vector<unsigned> scratch(PRE_SCRATCH);
    long checked = 0, matched = 0, multi = 0;
    const int MW = (S + 31) / 32;
    std::vector<unsigned> words((size_t)CH * 32);
    for (int p = 0; p < NP; p++)
        for (int lane = 0; lane < 64; lane++) {   // phase 3: lane = (group, end)
            const int g = lane >> 1, X = lane & 1;
            unsigned mword = 0;
            if (D.m[p] <= 22 && (p & 1)) prescan_dp<22, PRE_MAXSYM - 4>(gpl.data(), scratch.data(), lane, CH, D, p, words.data(), 32, &mword);   // (both row counts that fit)
            else if (D.m[p] <= 24) prescan_dp<24, PRE_MAXSYM - 4>(gpl.data(), scratch.data(), lane, CH, D, p, words.data(), 32, &mword);
            else prescan_dp<31, PRE_MAXSYM - 4>(gpl.data(), scratch.data(), lane, CH, D, p, words.data(), 32, &mword);
            for (int r = 0; r < 32; r++) {
                const int read = g * 32 + r;
                if (dirty[read]) continue;   // not pure ACGT: the demux kernel scans such reads itself
                std::string text;
                if (X) text = tails[read];
                else { text.assign(heads[read].rbegin(), heads[read].rend()); for (auto &c : text) c = comp(c); }
                const std::vector<int> score = reference_scores(pats[p], text);
                const int NV = (int)text.size();   // min(len, S) columns count
                // (1) the raw flag words: lt = new running minimum (starting from m), e = at the running minimum
                const int m = (int)pats[p].size();
                int run = m;
                bool ok = true;
                for (int j = 0; j < NV; j++) {
                    const bool lt = score[j] < run;
                    if (lt) run = score[j];
                    const bool e = score[j] == run;
                    const unsigned w = words[(size_t)(j >> 4) * 32 + r];
                    if (((w >> (j & 15)) & 1u) != (unsigned)lt || ((w >> (16 + (j & 15))) & 1u) != (unsigned)e) ok = false;
                }
                // (2) the consumer's decode: distance, first optimal end, all optimal ends
                unsigned mrow[9];
                int jstar = -1, nloc = -1;
                const int best = CH == 5 ? prescan_decode<5>(words.data() + r, 32, CH, MW, m, ks[p], NV, mrow, &jstar, &nloc)
                                         : prescan_decode<0>(words.data() + r, 32, CH, MW, m, ks[p], NV, mrow, &jstar, &nloc);
                if (best != run) ok = false;
                // (3) the match word: exact for full windows, a superset for short reads
                {
                    const bool mb = (mword >> r) & 1u;
                    if (NV == S ? mb != (run <= ks[p]) : (run <= ks[p] && !mb)) ok = false;
                }
                if (run <= ks[p]) {
                    matched++;
                    int ejs = -1, en = 0;
                    for (int j = 0; j < S; j++)
                        if (j < NV && score[j] == run) {
                            if (ejs < 0) ejs = j;
                            en++;
                            if (!((mrow[j >> 5] >> (j & 31)) & 1u)) ok = false;
                        } else if ((mrow[j >> 5] >> (j & 31)) & 1u) ok = false;
                    if (ejs != jstar || en != nloc) ok = false;
                    if (en > 1) multi++;
                }
                checked++;
                if (!ok) {
                    if (bad < 10) printf("MISMATCH read %d primer %d end %d (best %d, expected %d)\n", read, p, X, best, run);
                    bad++;
                }
            }
        }
    const long ovf = multi;
    printf("S=%d seed=%u: %ld alignments checked, %ld matched, %ld with several optimal ends, %ld mismatches\n", S, seed, checked, matched, ovf, bad);
    return bad ? 1 : 0;
}
