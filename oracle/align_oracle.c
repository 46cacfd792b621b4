/*
 * oracle/align_oracle.c -- TEST INFRASTRUCTURE ONLY (parity oracle, never shipped).
 *
 * Plain O(m*n) dynamic-programming restatement of the alignment primitive the
 * reference calls on its hot path:
 *     edlib.align(query, target, mode, task='locations', k,
 *                 additionalEqualities=IUPAC_EQUIV)
 * at /root/reference/src/specimux/alignment.py:42 (modes HW/SHW) and the plain
 * NW distance used for thresholds at src/specimux/orchestration.py:552.
 *
 * edlib itself (Martinsos/edlib >= 1.1.2, pyproject.toml:28) is NOT vendored in
 * the reference and is not installed here, so this file restates its published
 * semantics:
 *   - unit-cost edit distance, match iff eq(a,b) (28 symmetric, non transitive
 *     IUPAC pairs of src/specimux/constants.py:13-20, or exact equality);
 *   - HW: free gaps before/after query in target (D[0][j]=0);
 *     SHW: target prefix, free end gap (D[0][j]=j); NW: global;
 *   - result: best = min over end columns; if k>=0 and best>k -> -1, no locations;
 *     else EVERY end column with D[m][j]==best, ascending, inclusive 0-based;
 *   - HW start for an end e: the SMALLEST start s with NW(query,target[s..e])==best
 *     (edlib runs SHW on the reversed strings and takes the LAST optimal position);
 *     SHW/NW: start 0;
 *   - empty query or target: HW/SHW give editDistance=len(query) IGNORING k and one
 *     location (None,-1); NW gives max(len) and end tlen-1.
 *
 * This is deliberately the dumbest possible implementation: no bit vectors, no
 * banding.  The HIP kernels (Myers bit-vector) are checked against it.
 */
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#define ORA_NONE INT_MIN /* the Python binding's `None` start */

static unsigned char g_eq_iupac[256][256];
static int g_eq_ready = 0;

static void build_eq(void) {
    /* src/specimux/constants.py:13-20 */
    static const char *pairs[] = {
        "YC", "YT", "RA", "RG", "NA", "NC", "NG", "NT", "WA", "WT", "MA", "MC",
        "SC", "SG", "KG", "KT", "BC", "BG", "BT", "DA", "DG", "DT", "HA", "HC",
        "HT", "VA", "VC", "VG"};
    memset(g_eq_iupac, 0, sizeof(g_eq_iupac));
    for (int c = 0; c < 256; c++) g_eq_iupac[c][c] = 1;
    for (size_t i = 0; i < sizeof(pairs) / sizeof(pairs[0]); i++) {
        unsigned char a = (unsigned char)pairs[i][0], b = (unsigned char)pairs[i][1];
        g_eq_iupac[a][b] = 1;
        g_eq_iupac[b][a] = 1;
    }
    g_eq_ready = 1;
}

static inline int eqc(unsigned char a, unsigned char b, int iupac) {
    return iupac ? g_eq_iupac[a][b] : (a == b);
}

static inline int min3(int a, int b, int c) {
    int m = a < b ? a : b;
    return m < c ? m : c;
}

/* Fill last[j] = D[m][j], j = 0..n for the given mode. */
static void dp_last_row(const unsigned char *q, int m, const unsigned char *t, int n,
                        int mode, int iupac, int *last) {
    int *prev = (int *)malloc(sizeof(int) * (size_t)(m + 1));
    int *cur = (int *)malloc(sizeof(int) * (size_t)(m + 1));
    for (int i = 0; i <= m; i++) prev[i] = i; /* column 0: D[i][0] = i */
    last[0] = prev[m];
    for (int j = 1; j <= n; j++) {
        cur[0] = (mode == 0) ? 0 : j;
        for (int i = 1; i <= m; i++) {
            int sub = prev[i - 1] + (eqc(q[i - 1], t[j - 1], iupac) ? 0 : 1);
            cur[i] = min3(sub, prev[i] + 1, cur[i - 1] + 1);
        }
        last[j] = cur[m];
        int *tmp = prev; prev = cur; cur = tmp;
    }
    free(prev);
    free(cur);
}

/*
 * mode: 0 HW, 1 SHW, 2 NW.  k<0: no threshold.  iupac: 1 = additionalEqualities.
 * Returns 0; *dist = -1 when best > k.  starts[i]==ORA_NONE mirrors (None,-1).
 * *nloc is the TRUE number of locations; at most cap are written.
 */
int oracle_align(const char *q_, int m, const char *t_, int n, int k, int mode, int iupac,
                 int *dist, int *starts, int *ends, int cap, int *nloc) {
    const unsigned char *q = (const unsigned char *)q_;
    const unsigned char *t = (const unsigned char *)t_;
    if (!g_eq_ready) build_eq();
    *nloc = 0;
    if (m == 0 || n == 0) {
        if (mode == 2) {
            *dist = m > n ? m : n;
            if (cap > 0) { starts[0] = 0; ends[0] = n - 1; }
        } else {
            *dist = m; /* k ignored: alignment.py:44-46 clamps afterwards */
            if (cap > 0) { starts[0] = ORA_NONE; ends[0] = -1; }
        }
        *nloc = 1;
        return 0;
    }
    int *last = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    dp_last_row(q, m, t, n, mode, iupac, last);
    int best;
    if (mode == 2) {
        best = last[n];
    } else {
        best = INT_MAX;
        for (int j = 1; j <= n; j++) if (last[j] < best) best = last[j];
    }
    if (k >= 0 && best > k) {
        *dist = -1;
        free(last);
        return 0;
    }
    *dist = best;
    int cnt = 0;
    unsigned char *rq = NULL, *rt = NULL;
    int *rlast = NULL;
    if (mode == 0) {
        rq = (unsigned char *)malloc((size_t)m);
        for (int i = 0; i < m; i++) rq[i] = q[m - 1 - i];
        rt = (unsigned char *)malloc((size_t)n);
        rlast = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    }
    for (int j = (mode == 2 ? n : 1); j <= n; j++) {
        if (last[j] != best) continue;
        int e = j - 1, s = 0;
        if (mode == 0) {
            /* reversed target prefix t[0..e], SHW of reversed query, take the LAST
             * optimal end position p  ->  start = e - p  (smallest start). */
            int len = e + 1;
            for (int i = 0; i < len; i++) rt[i] = t[e - i];
            dp_last_row(rq, m, rt, len, 1, iupac, rlast);
            int p = -1;
            for (int c = 1; c <= len; c++) if (rlast[c] == best) p = c - 1;
            s = e - p; /* p>=0 is guaranteed: min over starts equals the HW score */
        }
        if (cnt < cap) { starts[cnt] = s; ends[cnt] = e; }
        cnt++;
    }
    *nloc = cnt;
    free(last);
    free(rq);
    free(rt);
    free(rlast);
    return 0;
}
