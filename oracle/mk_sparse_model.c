/* mk_sparse_model.c -- TEST INFRASTRUCTURE ONLY (never linked, loaded or called by the product path).
 *
 * CPU model of the device's SPARSE order-exact Munkres emulation (multiple-object-tracking_amd/csrc/mk_sparse.hip).
 *
 * The reference (trackers/hungarian/hungarian.cpp:29-368) works on the dense matrix.  The emulation keeps, per row,
 * only its K smallest entries (the candidate lists of the fast path) and runs the reference's state machine on them:
 * same scan orders (step 1 :93-101 rows ascending / first zero by column; step 3 :249-275 columns ascending, rows
 * ascending, one hit per column and sweep), same float64 element updates in the same order (:355-364), same zero test
 * fabs(x) < DBL_EPSILON.  An entry that is NOT in a candidate list has the value
 *        c[i][j] - rowmin_i + A_i(t) - S_j(t)  >=  c[i][j] - rowmin_i - S_j(final)
 * (A_i: what step 5 added to row i so far, S_j: what it subtracted from column j; both only grow).  If that lower bound
 * is > margin for EVERY non-candidate entry, none of them was ever zero or the minimum h of a step 5 (its value after
 * the step is still positive, so it was strictly above h), hence the dense run and the sparse run are the same run.
 * The check is done after the fact with the final S_j; if it fails the caller must run the dense emulation.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MKS_KMAX 16

typedef struct { int status; /* 0 ok, 1 a-posteriori check failed, 2 not applicable */ long primes, s5, aug; double maxS; } mks_info;

static int cmp_cand(double a, int ja, double b, int jb) { return a < b || (a == b && ja < jb); }

int mks_solve(const double* c, int nR, int nC, int K, double margin, int* assignment, mks_info* info)
{
    memset(info, 0, sizeof *info);
    for (int i = 0; i < nR; i++) assignment[i] = -1;
    if (nR <= 0 || nC <= 0 || nR > nC || K < 1 || K > MKS_KMAX) { info->status = 2; return 2; }
    if (K > nC) K = nC;
    int* cj = malloc(sizeof(int) * (size_t)nR * K); double* d = malloc(sizeof(double) * (size_t)nR * K);
    double* rowmin = malloc(sizeof(double) * (size_t)nR);
    for (int i = 0; i < nR; i++) {                                     /* candidate lists: K smallest, (cost, column) order */
        int nk = 0;
        for (int j = 0; j < nC; j++) {
            const double x = c[i + (size_t)nR * j];
            int p = nk;
            if (nk == K) { if (!cmp_cand(x, j, d[i * K + K - 1], cj[i * K + K - 1])) continue; p = K - 1; } else nk++;
            while (p > 0 && cmp_cand(x, j, d[i * K + p - 1], cj[i * K + p - 1])) { d[i * K + p] = d[i * K + p - 1]; cj[i * K + p] = cj[i * K + p - 1]; p--; }
            d[i * K + p] = x; cj[i * K + p] = j;
        }
        rowmin[i] = d[i * K];
        for (int k = 0; k < K; k++) d[i * K + k] -= rowmin[i];         /* hungarian.cpp:83-89 */
    }
    unsigned char* covC = calloc((size_t)nC, 1), *covR = calloc((size_t)nR, 1);
    int* starC = malloc(sizeof(int) * (size_t)nR), *starR = malloc(sizeof(int) * (size_t)nC), *primeC = malloc(sizeof(int) * (size_t)nR);
    double* S = calloc((size_t)nC, sizeof(double));
    for (int r = 0; r < nR; r++) { starC[r] = -1; primeC[r] = -1; }
    for (int j = 0; j < nC; j++) starR[j] = -1;
#define ISZ(r, k) (fabs(d[(r) * K + (k)]) < DBL_EPSILON)
    for (int r = 0; r < nR; r++) {                                     /* step 1 (:93-101): first zero by COLUMN index whose column is free */
        int best = -1;
        for (int k = 0; k < K; k++) if (ISZ(r, k) && !covC[cj[r * K + k]] && (best < 0 || cj[r * K + k] < best)) best = cj[r * K + k];
        if (best >= 0) { starC[r] = best; starR[best] = r; covC[best] = 1; }
    }
    /* transposed lists: rows (ascending) that hold column j as a candidate */
    int* tptr = calloc((size_t)nC + 1, sizeof(int)); int* trow = malloc(sizeof(int) * (size_t)nR * K); int* tk = malloc(sizeof(int) * (size_t)nR * K);
    for (int r = 0; r < nR; r++) for (int k = 0; k < K; k++) tptr[cj[r * K + k] + 1]++;
    for (int j = 0; j < nC; j++) tptr[j + 1] += tptr[j];
    { int* cur = malloc(sizeof(int) * (size_t)nC); memcpy(cur, tptr, sizeof(int) * (size_t)nC);
      for (int r = 0; r < nR; r++) for (int k = 0; k < K; k++) { const int j = cj[r * K + k]; trow[cur[j]] = r; tk[cur[j]] = k; cur[j]++; }
      free(cur); }
    for (;;) {
        int n = 0; for (int j = 0; j < nC; j++) n += covC[j];
        if (n == nR) break;                                            /* step 2b (:216-237) */
        int jumped = 0;
        for (;;) {
            int zerosFound = 1;
            while (zerosFound && !jumped) {                            /* step 3 (:240-280) */
                zerosFound = 0;
                for (int col = 0; col < nC && !jumped; col++) {
                    if (covC[col]) continue;
                    for (int t = tptr[col]; t < tptr[col + 1]; t++) {
                        const int r = trow[t];
                        if (covR[r] || !ISZ(r, tk[t])) continue;
                        primeC[r] = col; info->primes++;
                        if (starC[r] < 0) {                            /* step 4 (:283-334) */
                            info->aug++;
                            int cr = r, cc = col;
                            for (;;) { const int old_r = starR[cc]; starC[cr] = cc; starR[cc] = cr; if (old_r < 0) break; cc = primeC[old_r]; cr = old_r; }
                            for (int q = 0; q < nR; q++) { primeC[q] = -1; covR[q] = 0; }
                            for (int q = 0; q < nC; q++) covC[q] = starR[q] >= 0;      /* step 2a (:192-213) */
                            jumped = 1;
                        } else { covR[r] = 1; covC[starC[r]] = 0; zerosFound = 1; }
                        break;
                    }
                }
            }
            if (jumped) break;
            double h = DBL_MAX;                                        /* step 5 (:337-368) */
            for (int r = 0; r < nR; r++) if (!covR[r]) for (int k = 0; k < K; k++) if (!covC[cj[r * K + k]] && d[r * K + k] < h) h = d[r * K + k];
            if (h == DBL_MAX) { info->status = 1; goto out; }          /* no candidate entry among uncovered x uncovered: the minimum lies outside the lists */
            for (int r = 0; r < nR; r++) for (int k = 0; k < K; k++) {
                double x = d[r * K + k];
                if (covR[r]) x += h;
                if (!covC[cj[r * K + k]]) x -= h;
                d[r * K + k] = x;
            }
            for (int j = 0; j < nC; j++) if (!covC[j]) S[j] += h;
            info->s5++;
        }
    }
    /* a-posteriori check of every entry outside the candidate lists */
    for (int i = 0; i < nR && !info->status; i++) {
        const double lc = c[i + (size_t)nR * cj[i * K + K - 1]]; const int lj = cj[i * K + K - 1];
        for (int j = 0; j < nC; j++) {
            const double x = c[i + (size_t)nR * j];
            if (!cmp_cand(lc, lj, x, j)) continue;                     /* rank <= K: a candidate */
            if (!(x - rowmin[i] - S[j] > margin)) { info->status = 1; break; }
        }
    }
    for (int j = 0; j < nC; j++) if (S[j] > info->maxS) info->maxS = S[j];
out:
    if (!info->status) for (int r = 0; r < nR; r++) assignment[r] = starC[r];
    free(cj); free(d); free(rowmin); free(covC); free(covR); free(starC); free(starR); free(primeC); free(S); free(tptr); free(trow); free(tk);
    return info->status;
}
