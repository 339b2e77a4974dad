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
#include <stdio.h>
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MKS_KMAX 16

typedef struct { int status; /* 0 ok, 1 a-posteriori check failed, 2 not applicable */ long primes, s5, aug; double maxS; long iters; /* batched variant: event-loop iterations */ } mks_info;

static int cmp_cand(double a, int ja, double b, int jb) { return a < b || (a == b && ja < jb); }

/* union-find over the COLUMNS of the zero graph (lazy variant): a row ties all columns it holds a zero in together */
static int uf_find(int* p, int x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }
static void uf_union(int* p, unsigned char* dirty, int a, int b) { a = uf_find(p, a); b = uf_find(p, b); if (a == b) return; if (b < a) { int t = a; a = b; b = t; } p[b] = a; dirty[a] |= dirty[b]; }
static unsigned long long mix64(unsigned long long h, unsigned long long v) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); return h; }

/* lazy: at an augmentation only the connected components of the zero graph that are DIRTY are reset (see the comment at the reset);
 * trace (optional, n_trace entries): a hash of (row covers, column covers, primes, stars) at every step-5 entry */
static int mks_run(const double* c, int nR, int nC, int K, double margin, int* assignment, mks_info* info, int batch, int lazy, unsigned long long* trace, int n_trace)
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
    if (batch) {
        /* The device's BATCHED event loop (mk_sparse.hip): up to `batch` consecutive events of one sweep are taken together when
         * they cannot influence each other -- distinct first rows, every row but possibly the last one starred, and no column
         * uncovered by an earlier event of the batch that could produce an event of its own before a later column of the batch.
         * Same primes, covers and augmentations in the same order as the loop below; only the iteration count differs. */
        int* bc = malloc(sizeof(int) * (size_t)batch), *br = malloc(sizeof(int) * (size_t)batch), *bs = malloc(sizeof(int) * (size_t)batch);
        int from = 0, found = 0, nstar = 0;
        for (int j = 0; j < nC; j++) nstar += covC[j];
        while (nstar < nR) {
            /* candidates: uncovered columns >= from with a live zero, ascending */
            int nc = 0;
            for (int col = from; col < nC && nc < batch; col++) {
                if (covC[col]) continue;
                for (int t = tptr[col]; t < tptr[col + 1]; t++) {
                    const int r = trow[t];
                    if (covR[r] || !ISZ(r, tk[t])) continue;
                    bc[nc] = col; br[nc] = r; bs[nc] = starC[r]; nc++;
                    break;
                }
            }
            info->iters++;
            if (!nc) {
                if (found) { found = 0; from = 0; continue; }
                double h = DBL_MAX;                                        /* step 5 (:337-368) */
                for (int r = 0; r < nR; r++) if (!covR[r]) for (int k = 0; k < K; k++) if (!covC[cj[r * K + k]] && d[r * K + k] < h) h = d[r * K + k];
                if (h == DBL_MAX) { info->status = 1; free(bc); free(br); free(bs); goto out; }
                for (int r = 0; r < nR; r++) for (int k = 0; k < K; k++) {
                    double x = d[r * K + k];
                    if (covR[r]) x += h;
                    if (!covC[cj[r * K + k]]) x -= h;
                    d[r * K + k] = x;
                }
                for (int j = 0; j < nC; j++) if (!covC[j]) S[j] += h;
                info->s5++;
                from = 0; found = 0;
                continue;
            }
            /* the longest prefix that is safe to take at once */
            int f = nc, smin = 0x7FFFFFFF;
            for (int q = 0; q < nc; q++) {
                if (bs[q] < 0 || bs[q] <= bc[q]) continue;
                int live = 0;                                              /* a live zero in the star's column besides the row being covered */
                for (int t = tptr[bs[q]]; t < tptr[bs[q] + 1]; t++) { const int r = trow[t]; if (r != br[q] && !covR[r] && ISZ(r, tk[t])) { live = 1; break; } }
                if (live && bs[q] < smin) smin = bs[q];
            }
            for (int q = 0; q < nc; q++) {
                int stop = bc[q] > smin;
                for (int i = 0; i < q && !stop; i++) if (br[i] == br[q]) stop = 1;
                if (stop) { f = q; break; }
                if (bs[q] < 0) { f = q + 1; break; }
            }
            static int dbg = -1; if (dbg < 0) dbg = getenv("MKS_DEBUG") != NULL;
            if (dbg) { int why = 0; if (f < nc) { if (bc[f] > smin) why = 1; else why = 2; } if (f > 0 && bs[f - 1] < 0) why = 3; fprintf(stderr, "nc %d f %d why %d\n", nc, f, why); }
            for (int q = 0; q < f; q++) {
                const int r = br[q], col = bc[q];
                primeC[r] = col; info->primes++;
                if (starC[r] < 0) {                                        /* step 4 (:283-334) */
                    info->aug++;
                    int cr = r, cc = col;
                    for (;;) { const int old_r = starR[cc]; starC[cr] = cc; starR[cc] = cr; if (old_r < 0) break; cc = primeC[old_r]; cr = old_r; }
                    for (int z = 0; z < nR; z++) { primeC[z] = -1; covR[z] = 0; }
                    for (int z = 0; z < nC; z++) covC[z] = starR[z] >= 0;
                    nstar++; from = 0; found = 0;
                } else { covR[r] = 1; covC[starC[r]] = 0; found = 1; from = col + 1; }
            }
        }
        free(bc); free(br); free(bs);
    } else {
        /* LAZY RESET (lazy != 0; what the device runs since round 4).  Between two step 5s the connected components of the zero
         * graph (columns tied together by every row that holds zeros in them) evolve independently of each other: the sweep visits
         * columns in ascending order and passes repeat while anything happened, so what a component does in pass p depends on its
         * own state only.  After an augmentation the reference uncovers every row and re-grows its whole alternating forest from
         * scratch (:324-334, :192-213).  For a component that (1) was grown from scratch in a phase that ran until nothing was left
         * to do, and (2) has had no zero appear or vanish and no star change since, that re-growth reproduces covers and primes
         * exactly -- so it is skipped.  Everything else is DIRTY and is reset: the component of the augmenting path, components
         * whose zero masks a step 5 changed (they continued from an older state; the reference re-grows them in sweep order over
         * the NEW zeros, which may prime other columns), and components reset in a phase that was itself cut short by an
         * augmentation.  Components only ever merge here (a vanished zero does not split them): resetting a clean component
         * with a dirty one is exact, just not minimal. */
        int* par = malloc(sizeof(int) * (size_t)nC); unsigned char* dirty = malloc((size_t)nC);
        for (int j = 0; j < nC; j++) { par[j] = j; dirty[j] = 1; }         /* phase 0 grows everything from scratch */
        for (int r = 0; r < nR; r++) { int first = -1; for (int k = 0; k < K; k++) if (ISZ(r, k)) { if (first < 0) first = cj[r * K + k]; else uf_union(par, dirty, first, cj[r * K + k]); } }
        int nstar = 0; for (int j = 0; j < nC; j++) nstar += covC[j];
        int scratch_phase = 1;                                             /* the running phase started from a reset (or is phase 0) */
        int tn = 0;
        while (nstar < nR) {                                               /* step 2b (:216-237) */
            int jumped = 0, zerosFound = 1;
            while (zerosFound && !jumped) {                                /* step 3 (:240-280) */
                zerosFound = 0;
                for (int col = 0; col < nC && !jumped; col++) {
                    if (covC[col]) continue;
                    for (int t = tptr[col]; t < tptr[col + 1]; t++) {
                        const int r = trow[t];
                        if (covR[r] || !ISZ(r, tk[t])) continue;
                        primeC[r] = col; info->primes++;
                        if (starC[r] < 0) {                                /* step 4 (:283-334) */
                            info->aug++;
                            int cr = r, cc = col;
                            for (;;) { const int old_r = starR[cc]; starC[cr] = cc; starR[cc] = cr; if (old_r < 0) break; cc = primeC[old_r]; cr = old_r; }
                            if (!lazy) {
                                for (int q = 0; q < nR; q++) { primeC[q] = -1; covR[q] = 0; }
                                for (int q = 0; q < nC; q++) covC[q] = starR[q] >= 0;      /* step 2a (:192-213) */
                            } else {
                                dirty[uf_find(par, col)] = 1;              /* the path lies in one component (its edges are zeros) */
                                primeC[r] = -1;
                                for (int q = 0; q < nR; q++) if (covR[q] && dirty[uf_find(par, primeC[q])]) { primeC[q] = -1; covR[q] = 0; }
                                for (int q = 0; q < nC; q++) if (dirty[uf_find(par, q)]) covC[q] = starR[q] >= 0;
                            }
                            nstar++; jumped = 1; scratch_phase = 1;
                        } else { covR[r] = 1; covC[starC[r]] = 0; zerosFound = 1; }
                        break;
                    }
                }
            }
            if (jumped) continue;
            if (trace && tn < n_trace) {                                   /* state at step-5 entry */
                unsigned long long h = 1469598103934665603ull;
                for (int q = 0; q < nR; q++) h = mix64(h, ((unsigned long long)covR[q] << 40) ^ ((unsigned long long)(unsigned)(primeC[q] + 1) << 20) ^ (unsigned)(starC[q] + 1));
                for (int q = 0; q < nC; q++) h = mix64(h, covC[q]);
                trace[tn] = h;
            }
            tn++;
            if (lazy && scratch_phase) memset(dirty, 0, (size_t)nC);      /* a from-scratch phase ran to its end: everything is clean */
            scratch_phase = 0;
            double h = DBL_MAX;                                            /* step 5 (:337-368) */
            for (int r = 0; r < nR; r++) if (!covR[r]) for (int k = 0; k < K; k++) if (!covC[cj[r * K + k]] && d[r * K + k] < h) h = d[r * K + k];
            if (h == DBL_MAX) { info->status = 1; free(par); free(dirty); goto out; }   /* no candidate entry among uncovered x uncovered: the minimum lies outside the lists */
            for (int r = 0; r < nR; r++) {
                int changed = 0;
                for (int k = 0; k < K; k++) {
                    const int was = ISZ(r, k);
                    double x = d[r * K + k];
                    if (covR[r]) x += h;
                    if (!covC[cj[r * K + k]]) x -= h;
                    d[r * K + k] = x;
                    if (was != ISZ(r, k)) { changed = 1; if (lazy) dirty[uf_find(par, cj[r * K + k])] = 1; }
                }
                if (lazy && changed) { int first = -1; for (int k = 0; k < K; k++) if (ISZ(r, k)) { if (first < 0) first = cj[r * K + k]; else uf_union(par, dirty, first, cj[r * K + k]); } }
            }
            for (int j = 0; j < nC; j++) if (!covC[j]) S[j] += h;
            info->s5++;
        }
        info->iters = tn;
        free(par); free(dirty);
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
    (void)0;
    free(cj); free(d); free(rowmin); free(covC); free(covR); free(starC); free(starR); free(primeC); free(S); free(tptr); free(trow); free(tk);
    return info->status;
}

int mks_solve(const double* c, int nR, int nC, int K, double margin, int* assignment, mks_info* info) { return mks_run(c, nR, nC, K, margin, assignment, info, 0, 0, NULL, 0); }
/* the same run with the lazy reset (mode 1) or the reference's full reset (mode 0), recording a hash of the machine's state at every step-5 entry */
int mks_solve_traced(const double* c, int nR, int nC, int K, double margin, int lazy, int* assignment, mks_info* info, unsigned long long* trace, int n_trace) { return mks_run(c, nR, nC, K, margin, assignment, info, 0, lazy, trace, n_trace); }
/* same run with the event loop taken `batch` (<= 64 on the device) events at a time where that is provably order-neutral */
int mks_solve_batched(const double* c, int nR, int nC, int K, double margin, int batch, int* assignment, mks_info* info) { return mks_run(c, nR, nC, K, margin, assignment, info, batch < 1 ? 1 : batch, 0, NULL, 0); }
