/* tools/cpu_native/cpu_scan.c -- CONTEXT ONLY: a best-effort native CPU comparator for bench.py's gpu_over_cpu (BASELINE.md
 * section 2, "B-native").  Never imported, linked or called by the package, the tests or the oracle.
 *
 * It does the ALIGNMENT WORK of the hot path for configs[1] with the textbook CPU method -- Myers' bit-vector algorithm,
 * one 64-bit word per pattern, OpenMP over reads:
 *   per read: 2 primers x 2 end windows, HW scan over search_len columns (all columns, minimum + first optimal end);
 *   per matched (primer, end) on the orientation the votes select: every barcode of that primer, SHW scan over the first
 *   len + k bases after the primer end, minimum distance and tie count.
 * No scorer, no record formatting, no I/O: the reported reads/s is an UPPER bound on what a native CPU demultiplexer
 * built this way would reach on these cores (the selection / dereplication logic on top is cheap next to the scans).
 *
 * Input: a binary file of n x (2 * S) window bytes (head | tail, as smx_pack_windows cuts them) and the patterns on the
 * command line.  gcc -O3 -march=native -fopenmp cpu_scan.c -o cpu_scan */
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t peq_of(const char *pat, int m, char c) {
    uint64_t v = 0;
    for (int i = 0; i < m; i++) if (pat[i] == c) v |= 1ull << i;
    return v;
}
static const char *ALPHA = "ACGT";
static inline int code(unsigned char ch) { return (ch >> 1) & 3; }   /* A 0, C 1, T 2, G 3 */
static const char CODE2CHAR[4] = {'A', 'C', 'T', 'G'};
static inline char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; }

typedef struct { uint64_t peq[4]; int m; } Pat;
static void pat_init(Pat *p, const char *s) {
    p->m = (int)strlen(s);
    for (int c = 0; c < 4; c++) p->peq[c] = peq_of(s, p->m, CODE2CHAR[c]);
}

/* HW (infix): minimum over all columns and the first column that reaches it */
static inline int hw_scan(const Pat *p, const unsigned char *t, int n, int *jstar) {
    uint64_t Pv = ~0ull, Mv = 0;
    int score = p->m, best = p->m + 1, js = 0;
    const uint64_t top = 1ull << (p->m - 1);
    for (int j = 0; j < n; j++) {
        const uint64_t Eq = p->peq[code(t[j])];
        const uint64_t Xv = Eq | Mv, Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
        uint64_t Ph = Mv | ~(Xh | Pv), Mh = Pv & Xh;
        score += (Ph & top) ? 1 : 0;
        score -= (Mh & top) ? 1 : 0;
        Ph <<= 1; Mh <<= 1;
        Pv = Mh | ~(Xv | Ph);
        Mv = Ph & Xv;
        if (score < best) { best = score; js = j; }
    }
    *jstar = js;
    return best;
}
/* SHW (prefix) */
static inline int shw_scan(const Pat *p, const unsigned char *t, int n) {
    uint64_t Pv = ~0ull, Mv = 0;
    int score = p->m, best = p->m + 1;
    const uint64_t top = 1ull << (p->m - 1);
    for (int j = 0; j < n; j++) {
        const uint64_t Eq = p->peq[code(t[j])];
        const uint64_t Xv = Eq | Mv, Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
        uint64_t Ph = Mv | ~(Xh | Pv), Mh = Pv & Xh;
        score += (Ph & top) ? 1 : 0;
        score -= (Mh & top) ? 1 : 0;
        Ph = (Ph << 1) | 1ull; Mh <<= 1;
        Pv = Mh | ~(Xv | Ph);
        Mv = Ph & Xv;
        if (score < best) best = score;
    }
    return best;
}

int main(int argc, char **argv) {
    if (argc < 8) { fprintf(stderr, "usage: cpu_scan windows.bin n S kidx primer_rc0 k0 primer_rc1 k1 nb0 bc... nb1 bc...\n"); return 2; }
    const char *path = argv[1];
    const long n = atol(argv[2]);
    const int S = atoi(argv[3]), kidx = atoi(argv[4]);
    Pat prim[2]; int pk[2];
    pat_init(&prim[0], argv[5]); pk[0] = atoi(argv[6]);
    pat_init(&prim[1], argv[7]); pk[1] = atoi(argv[8]);
    int a = 9, nb[2];
    Pat *bc[2];
    for (int p = 0; p < 2; p++) {
        nb[p] = atoi(argv[a++]);
        bc[p] = (Pat *)malloc(sizeof(Pat) * nb[p]);
        for (int i = 0; i < nb[p]; i++) pat_init(&bc[p][i], argv[a++]);
    }
    unsigned char *win = (unsigned char *)malloc((size_t)n * 2 * S);
    FILE *f = fopen(path, "rb");
    if (!f || fread(win, 1, (size_t)n * 2 * S, f) != (size_t)n * 2 * S) { fprintf(stderr, "cannot read %s\n", path); return 2; }
    fclose(f);
    long matched_ends = 0, best_sum = 0;
    double t0 = omp_get_wtime();
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : matched_ends, best_sum)
    for (long r = 0; r < n; r++) {
        unsigned char A[256], *B = win + (size_t)r * 2 * S + S;
        const unsigned char *head = win + (size_t)r * 2 * S;
        for (int j = 0; j < S; j++) A[j] = (unsigned char)comp((char)head[S - 1 - j]);   /* window A = revcomp(head) */
        for (int p = 0; p < 2; p++)
            for (int X = 0; X < 2; X++) {
                const unsigned char *t = X ? B : A;
                int js;
                const int d = hw_scan(&prim[p], t, S, &js);
                if (d > pk[p]) continue;
                matched_ends++;
                const int n_t = S - (js + 1);
                int best = 99;
                for (int i = 0; i < nb[p]; i++) {
                    const int cols = n_t < bc[p][i].m + kidx ? n_t : bc[p][i].m + kidx;
                    const int bd = shw_scan(&bc[p][i], t + js + 1, cols);
                    if (bd < best) best = bd;
                }
                best_sum += best;
            }
    }
    double dt = omp_get_wtime() - t0;
    printf("{\"reads\": %ld, \"seconds\": %.4f, \"reads_per_s\": %.1f, \"threads\": %d, \"matched_ends\": %ld, \"checksum\": %ld}\n", n, dt,
           n / dt, omp_get_max_threads(), matched_ends, best_sum);
    (void)ALPHA;
    return 0;
}
