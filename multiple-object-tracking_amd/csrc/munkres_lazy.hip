// munkres_lazy.hip -- order-exact Munkres (trackers/hungarian/hungarian.cpp:29-368) with LAZY column updates.
//
// Same state machine, scan orders, zero test and per-element float64 operation sequence as the reference
// (see assoc_kernels.hip for the eager emulation).  The difference is WHEN the element updates of step 5
// (:337-368) are applied.  The reference touches every element of every uncovered column and every covered row
// in each step 5; on one workgroup that is bound by a single CU's memory bandwidth (~55 columns x 8 KB x 3 per
// step, ~60 steps per frame at 1024 tracks).  Here:
//   * every step 5 appends (h, covered rows) to a log in LDS;
//   * a column's stored values are brought up to date ("materialised") by replaying the log entries it has not
//     seen -- element (r,c) at step k gets +h_k if row r was covered, then -h_k if column c was uncovered, which
//     for a column that is not kept eager means "c had never been starred yet" (k < startime[c]);
//   * never-starred columns (always uncovered) carry only cminU[c] = min over the uncovered rows of their true
//     values (exact: subtracting h is monotone) and the row attaining it; h of a step 5 is the minimum of these
//     and of the few eager columns; a lazy column is touched only when cminU - h becomes a zero ("hot"), when
//     the row attaining its minimum gets covered, or when it is starred;
//   * columns that are uncovered AND hold a star (uncovered by step 3) or hold a zero in a covered row are kept
//     eager (at most 8, elementwise update exactly like the reference);
//   * when rows are uncovered again (step 4) only the (old covered rows) x (lazy columns) elements are evaluated.
// The scheme was validated against an eagerly updated shadow matrix (values, bitmaps, h) before being written
// for the GPU; overflow of the small tables (log 120 entries, 8 eager columns, 8 covered rows) materialises
// every column and continues with the eager step 5.
#include "assoc_common.h"

using namespace assoc;

namespace {

#define LZ_LOG 120
#define LZ_MAXE 8
#define LZ_MAXCOV 8

struct LzShared {
    u64 bm[MK_MAXN * MK_MAXW];
    short starColOfRow[MK_MAXN], starRowOfCol[MK_MAXN], primeColOfRow[MK_MAXN];
    unsigned short list[MK_MAXN];
    union {
        struct { unsigned short clist[MK_MAXN]; unsigned short crosscnt[MK_MAXN]; unsigned int taken32[2 * MK_MAXW], cont32[2 * MK_MAXW]; } init;
        u64 cminkey[MK_MAXN];
    } u;
    int argrow[MK_MAXN];
    unsigned short applied[MK_MAXN], startime[MK_MAXN];
    double logH[128];
    u64 rowmask[64][2];                    // for the <= 64 rows ever covered while the log runs: bit k = covered at log step k
    unsigned char covid[MK_MAXN];          // row -> index into rowmask (valid where everCov is set)
    u64 covR[MK_MAXW], covC[MK_MAXW], hz[MK_MAXW], lazyM[MK_MAXW], everCov[MK_MAXW];   // everCov: rows covered in any logged step
    u64 redk[16];
    unsigned short elist[LZ_MAXE + 16];
    unsigned short crows[LZ_MAXCOV + 8];
    int flag[16];   // 0 action | 1 aux | 2 nE | 3 - | 4 ncrows | 5 K | 6 eager | 7 ev_row | 8 ev_col | 9 ev_sc | 10 last | 11 counter
};

enum { F_ACTION = 0, F_AUX = 1, F_NE = 2, F_NID = 3, F_NCR = 4, F_K = 5, F_EAGER = 6, F_ROW = 7, F_COL = 8, F_SC = 9, F_LAST = 10, F_CNT = 11 };

struct Ctx {
    LzShared& S; double* __restrict__ d; int nR, nC, wordsR, wordsC, tid, lane, wave;
};

__device__ __forceinline__ bool bit1024(const u64* m, int i) { return (m[i >> 6] >> (i & 63)) & 1; }

__device__ __forceinline__ void hz_set(LzShared& S, int c, bool on)
{
    unsigned int* wp = reinterpret_cast<unsigned int*>(&S.hz[c >> 6]) + ((c & 63) >> 5);
    if (on) atomicOr(wp, 1u << (c & 31)); else atomicAnd(wp, ~(1u << (c & 31)));
}

// true value of element (r, c): stored value with the pending log entries applied in order
__device__ __forceinline__ double replay(const LzShared& S, double x, int c, int r, int K)
{
    const int st = S.startime[c];
    const int a0 = S.applied[c];
    if (!bit1024(S.everCov, r)) {
        // this row was never covered while the log was recorded: only the column part applies (-h while the
        // column had not been starred yet); nothing at all for a column that was starred before the log began
        const int kend = min(K, st);
        for (int k = a0; k < kend; k++) x -= S.logH[k];
        return x;
    }
    const int id = S.covid[r];
    const u64 m0 = S.rowmask[id][0], m1 = S.rowmask[id][1];
    for (int k = a0; k < K; k++) {
        const double h = S.logH[k];
        const bool add = ((k < 64 ? m0 >> k : m1 >> (k - 64)) & 1) != 0;
        if (add) x += h;            // row was covered: += h      (hungarian.cpp:355-358)
        if (k < st) x -= h;         // column was uncovered: -= h (hungarian.cpp:361-364)
    }
    return x;
}

// Materialise `n` columns (workgroup-wide, 8 per round): replay, store, rebuild the zero bits of every row, refresh
// hz; with want_min also cminkey / argrow = min over the UNCOVERED rows.  Uniform arguments, contains barriers.
__device__ void materialize_cols(const Ctx& X, const unsigned short* cols, int n, bool want_min)
{
    LzShared& S = X.S;
    const int r = X.tid, K = S.flag[F_K];
    const size_t rclamp = (size_t)min(r, X.nR - 1);
    const bool rowcov = (S.covR[X.wave] >> X.lane) & 1;
    for (int base = 0; base < n; base += 4) {
        double v[4]; u64 key[4];
        if (want_min) {
            if (X.tid < 4 && base + X.tid < n) { const int c = cols[base + X.tid]; S.u.cminkey[c] = ~0ull; S.argrow[c] = 0x7fffffff; }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = X.d[rclamp + (size_t)X.nR * cols[min(base + j, n - 1)]];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            key[j] = ~0ull;
            if (base + j < n) {
                const int c = cols[base + j];
                const bool pending = S.applied[c] < K;
                const double x = replay(S, v[j], c, r, K);
                if (pending && r < X.nR) X.d[(size_t)r + (size_t)X.nR * c] = x;
                const bool z = r < X.nR && fabs(x) < DBL_EPSILON;
                const u64 bal = __ballot(z);
                if (X.lane == 0) S.bm[c * MK_MAXW + X.wave] = bal;
                if (want_min) {
                    if (r < X.nR && !rowcov) key[j] = dkey(x);
                    u64 wk = key[j];
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) { const u64 o = __shfl_xor(wk, off); if (o < wk) wk = o; }
                    if (X.lane == 0 && wk != ~0ull) atomicMin(&S.u.cminkey[c], wk);
                }
            }
        }
        __syncthreads();
        if (want_min) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (base + j < n) {
                    const int c = cols[base + j];
                    const u64 bal = __ballot(key[j] != ~0ull && key[j] == S.u.cminkey[c]);
                    if (X.lane == 0 && bal) atomicMin(&S.argrow[c], X.wave * 64 + __ffsll((long long)bal) - 1);
                }
            }
        }
        if (X.tid < 4 && base + X.tid < n) {
            const int c = cols[base + X.tid];
            S.applied[c] = (unsigned short)K;
            bool any = false;
            for (int w = 0; w < X.wordsR; w++) any |= S.bm[c * MK_MAXW + w] != 0;
            hz_set(S, c, any);
        }
        __syncthreads();
        if (want_min && X.tid < 4 && base + X.tid < n) { const int c = cols[base + X.tid]; if (S.argrow[c] == 0x7fffffff) S.argrow[c] = -1; }
    }
}

// every column up to date; optionally restart the log
__device__ void flush_all(const Ctx& X, bool reset_log)
{
    LzShared& S = X.S;
    for (int c = X.tid; c < X.nC; c += MK_THREADS) S.list[c] = (unsigned short)c;
    __syncthreads();
    materialize_cols(X, S.list, X.nC, false);
    if (reset_log) {
        for (int c = X.tid; c < MK_MAXN; c += MK_THREADS) { S.applied[c] = 0; S.startime[c] = (c < X.nC && S.starRowOfCol[c] >= 0) ? 0 : 0xFFFF; }
        if (X.tid == 0) { S.flag[F_K] = 0; S.flag[F_NID] = 0; }
        if (X.tid < MK_MAXW) S.everCov[X.tid] = 0;
    }
    __syncthreads();
}

// reference step 5 on fully materialised data (overflow fallback): identical to assoc_kernels.hip
__device__ void step5_eager(const Ctx& X)
{
    LzShared& S = X.S; double* __restrict__ d = X.d;
    const int nR = X.nR, nC = X.nC, tid = X.tid, lane = X.lane, wave = X.wave;
    const int ncu = S.flag[F_AUX];
    const int r = tid;
    const u64 cw = S.covR[wave];
    const bool mine = r < nR && !((cw >> lane) & 1);
    double h = DBL_MAX; double v[16];
    const size_t rclamp = (size_t)min(r, nR - 1);
    for (int k0 = 0; k0 < ncu; k0 += 16) {
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = d[rclamp + (size_t)nR * S.list[min(k0 + k, ncu - 1)]];
#pragma unroll
        for (int k = 0; k < 16; k++) if (mine && k0 + k < ncu && v[k] < h) h = v[k];
    }
    u64 hk = dkey(h);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const u64 o = __shfl_xor(hk, off); if (o < hk) hk = o; }
    if (lane == 0) S.redk[wave] = hk;
    __syncthreads();
    hk = S.redk[0];
#pragma unroll
    for (int w = 1; w < 16; w++) { const u64 o = S.redk[w]; if (o < hk) hk = o; }
    h = dunkey(hk);
    for (int k0 = 0; k0 < ncu; k0 += 16) {
        if (ncu > 16) {
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = d[rclamp + (size_t)nR * S.list[min(k0 + k, ncu - 1)]];
        }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (k0 + k < ncu) {
                const int c = S.list[k0 + k];
                bool z = false;
                if (mine) { const double nv = v[k] - h; d[(size_t)r + (size_t)nR * c] = nv; z = fabs(nv) < DBL_EPSILON; }
                const u64 bal = __ballot(z);
                if (lane == 0 && wave < X.wordsR) S.bm[c * MK_MAXW + wave] = (S.bm[c * MK_MAXW + wave] & cw) | (bal & ~cw);
            }
        }
    }
    bool anyCov = false;
    for (int w = 0; w < X.wordsR; w++) anyCov |= S.covR[w] != 0;
    if (anyCov) {
        __syncthreads();
        for (int w = 0; w < X.wordsR; w++) {
            u64 rows = S.covR[w];
            while (rows) {
                const int rr = w * 64 + (__ffsll((long long)rows) - 1); rows &= rows - 1;
                for (int c = tid; c < nC; c += MK_THREADS) {
                    double x = d[(size_t)rr + (size_t)nR * c] + h;
                    if (!bit1024(S.covC, c)) x -= h;
                    d[(size_t)rr + (size_t)nR * c] = x;
                    const bool z = fabs(x) < DBL_EPSILON;
                    unsigned int* wp = reinterpret_cast<unsigned int*>(&S.bm[c * MK_MAXW + (rr >> 6)]) + ((rr & 63) >> 5);
                    const unsigned int bit = 1u << (rr & 31);
                    const bool cur = (*wp & bit) != 0;
                    if (z != cur) { if (z) atomicOr(wp, bit); else atomicAnd(wp, ~bit); }
                }
            }
        }
    }
    __syncthreads();
    {
        const bool unc = tid < nC && !((S.covC[wave] >> lane) & 1);
        bool has = false;
        if (unc) for (int w = 0; w < X.wordsR; w++) has |= S.bm[tid * MK_MAXW + w] != 0;
        const u64 bal = __ballot(has);
        if (lane == 0) S.hz[wave] = (S.hz[wave] & S.covC[wave]) | bal;
    }
    __syncthreads();
}

// switch to the eager fallback: everything materialised, no lazy bookkeeping from here on
__device__ void go_eager(const Ctx& X)
{
    flush_all(X, false);
    if (X.tid == 0) { X.S.flag[F_EAGER] = 1; X.S.flag[F_NE] = 0; }
    __syncthreads();
}

__global__ void __launch_bounds__(MK_THREADS) munkres_lazy_kernel(AssocArgs a, int want_cost)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lz_raw[];
    LzShared& S = *reinterpret_cast<LzShared*>(lz_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int nR, nC; bool rowsTrk; resolve_dims(a, nR, nC, rowsTrk);
    double* __restrict__ d = a.ws.dist;
    int* stat = a.ws.status;
    const long long t_begin = wall_clock64();
    if (nR <= 0 || nC <= 0) { if (tid == 0) *a.ws.cost = 0.0; for (int r = tid; r < max(nR, 0); r += MK_THREADS) a.ws.assignment[r] = -1; return; }
    const int wordsR = (nR + 63) >> 6, wordsC = (nC + 63) >> 6;
    const bool perRow = nR <= nC;
    const int minDim = perRow ? nR : nC;
    const Ctx X{ S, d, nR, nC, wordsR, wordsC, tid, lane, wave };
    int n_s4 = 0, n_s5 = 0, n_sw = 0, n_find = 0, n_hot = 0; long long t_s3 = 0, t_s5 = 0, t_h4 = 0, t_h5 = 0;

    for (int i = tid; i < MK_MAXN; i += MK_THREADS) { S.starColOfRow[i] = -1; S.starRowOfCol[i] = -1; S.primeColOfRow[i] = -1; }
    if (tid < MK_MAXW) { S.covR[tid] = 0; S.covC[tid] = 0; S.lazyM[tid] = 0; S.everCov[tid] = 0; }
    if (tid < 2 * MK_MAXW) { S.u.init.taken32[tid] = 0; S.u.init.cont32[tid] = 0; }
    if (tid < 16) S.flag[tid] = 0;
    // ---- steps 1 + 2a: initial stars (hungarian.cpp:93-101 / :128-139), see assoc_kernels.hip ----
    {
        const int nL = perRow ? nR : nC, nX = perRow ? nC : nR;
        const int W = perRow ? wordsC : wordsR, WX = perRow ? wordsR : wordsC;
        const u64* lineBm = perRow ? a.ws.zr : a.ws.zc;
        const u64* crossBm = perRow ? a.ws.zc : a.ws.zr;
        for (int i = tid; i < nL * W; i += MK_THREADS) S.bm[(i / W) * MK_MAXW + (i % W)] = lineBm[i];
        for (int x = tid; x < nX; x += MK_THREADS) {
            int cnt = 0;
            for (int w = 0; w < WX; w++) cnt += __popcll(crossBm[(size_t)x * WX + w]);
            S.u.init.crosscnt[x] = (unsigned short)min(cnt, 65535);
        }
        __syncthreads();
        for (int l = tid; l < nL; l += MK_THREADS) {
            int fz = -1;
            for (int w = 0; w < W; w++) { const u64 m = S.bm[l * MK_MAXW + w]; if (m) { fz = w * 64 + __ffsll((long long)m) - 1; break; } }
            if (fz >= 0) {
                if (S.u.init.crosscnt[fz] == 1) {
                    const int row = perRow ? l : fz, col = perRow ? fz : l;
                    S.starColOfRow[row] = (short)col; S.starRowOfCol[col] = (short)row;
                    atomicOr(&S.u.init.taken32[fz >> 5], 1u << (fz & 31));
                } else atomicOr(&S.u.init.cont32[l >> 5], 1u << (l & 31));
            }
        }
        __syncthreads();
        if (wave == 0) {
            const u64 cont = (lane < MK_MAXW) ? ((u64)S.u.init.cont32[2 * lane] | ((u64)S.u.init.cont32[2 * lane + 1] << 32)) : 0;
            const int ncont = wave_list_bits(cont, 0, S.u.init.clist, lane);
            u64 taken = (lane < W) ? ((u64)S.u.init.taken32[2 * lane] | ((u64)S.u.init.taken32[2 * lane + 1] << 32)) : 0;
            for (int q0 = 0; q0 < ncont; q0 += 8) {
                u64 m[8]; int ln[8];
#pragma unroll
                for (int k = 0; k < 8; k++) { ln[k] = (q0 + k < ncont) ? S.u.init.clist[q0 + k] : -1; m[k] = (lane < W && ln[k] >= 0) ? S.bm[ln[k] * MK_MAXW + lane] : 0; }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (ln[k] < 0) break;
                    const int x = wave_first_bit(m[k] & ~taken, lane, W);
                    if (x >= 0) {
                        if (lane == (x >> 6)) taken |= 1ull << (x & 63);
                        if (lane == 0) {
                            const int row = perRow ? ln[k] : x, col = perRow ? x : ln[k];
                            S.starColOfRow[row] = (short)col; S.starRowOfCol[col] = (short)row;
                        }
                    }
                }
            }
        }
        __syncthreads();
        {
            const bool has = tid < nC && S.starRowOfCol[tid] >= 0;
            const u64 bal = __ballot(has);
            if (lane == 0) S.covC[wave] = bal;
        }
        __syncthreads();
    }
    int ncov = 0;
    for (int w = 0; w < wordsC; w++) ncov += __popcll(S.covC[w]);
    bool done = (ncov == minDim);                                      // step 2b (:216-237)
    const u64 vC = (lane < wordsC) ? ((lane == wordsC - 1 && (nC & 63)) ? ((1ull << (nC & 63)) - 1) : ~0ull) : 0;
    if (!done) {
        __syncthreads();
        for (int i = tid; i < nC * wordsR; i += MK_THREADS) S.bm[(i / wordsR) * MK_MAXW + (i % wordsR)] = a.ws.zc[i];
        for (int c = tid; c < MK_MAXN; c += MK_THREADS) {
            S.applied[c] = 0; S.startime[c] = (c < nC && S.starRowOfCol[c] >= 0) ? 0 : 0xFFFF; S.argrow[c] = -1; S.u.cminkey[c] = ~0ull;
        }
        __syncthreads();
        {
            bool has = false;
            if (tid < nC) for (int w = 0; w < wordsR; w++) has |= S.bm[tid * MK_MAXW + w] != 0;
            const u64 bal = __ballot(has);
            const bool lz = tid < nC && S.starRowOfCol[tid] < 0;           // never starred = uncovered = lazy
            const u64 bl = __ballot(lz);
            if (lane == 0) { S.hz[wave] = bal; S.lazyM[wave] = bl; }
        }
        __syncthreads();
        // cminU / argrow of every lazy column (stored values are the true values: the log is empty)
        if (wave == 0) { const int nl = wave_list_bits((lane < MK_MAXW) ? S.lazyM[lane] : 0, 0, S.list, lane); if (lane == 0) S.flag[F_AUX] = nl; }
        __syncthreads();
        materialize_cols(X, S.list, S.flag[F_AUX], true);
    }
    const long long t_init = wall_clock64();
    u64 cC = (lane < MK_MAXW) ? S.covC[lane] : 0, cR16 = 0, phaseUnc = 0, hz = (lane < MK_MAXW) ? S.hz[lane] : 0;
    bool covRany = false;
    int from = 0; bool found_in_sweep = false;                         // state of the current column sweep (:249), survives the handlers
    int guard = 0;
    while (!done) {
        const long long t_a = wall_clock64();
        // ========== steps 3 / 4 / 2a / 2b: wave 0, until the other waves are needed ==========
        if (wave == 0) {
            const bool eager = S.flag[F_EAGER] != 0;
            int action = 0;      // 2 step 5 | 3 finished | 4 covered a row (lazy bookkeeping) | 5 augmented (lazy bookkeeping)
            while (action == 0) {
                if (++n_sw > 8 * MK_MAXN * MK_MAXN) { action = 3; break; }
                int col = -1, row = -1;
                if (!covRany && from == 0) {
                    const u64 hm = (lane < MK_MAXW) ? (hz & ~cC & vC) : 0;
                    col = wave_first_bit(hm, lane, wordsC);
                    if (col < 0) { action = 2; break; }
                    const u64 mw = (lane < wordsR) ? S.bm[col * MK_MAXW + lane] : 0;
                    row = wave_first_bit(mw, lane, wordsR);
                    if (row < 0) { if (lane == (col >> 6)) hz &= ~(1ull << (col & 63)); continue; }
                } else {
                    const int q = lane >> 4, wd = lane & 15;
                    const int fw = from >> 6;
                    u64 cand = (lane < MK_MAXW) ? (~cC & vC & (hz | phaseUnc)) : 0;
                    if (lane < fw) cand = 0; else if (lane == fw) cand &= ~0ull << (from & 63);
                    for (;;) {
                        int cs[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            cs[j] = wave_first_bit(cand, lane, MK_MAXW);
                            if (cs[j] >= 0 && lane == (cs[j] >> 6)) cand &= ~(1ull << (cs[j] & 63));
                        }
                        if (cs[0] < 0) break;
                        const int myc = q == 0 ? cs[0] : q == 1 ? cs[1] : q == 2 ? cs[2] : cs[3];
                        u64 m = 0;
                        if (myc >= 0 && wd < wordsR) m = S.bm[myc * MK_MAXW + wd] & ~cR16;
                        const u64 bal = __ballot(m != 0);
                        if (bal) {
                            const int fl = __ffsll((long long)bal) - 1;
                            const int qq = fl >> 4;
                            col = qq == 0 ? cs[0] : qq == 1 ? cs[1] : qq == 2 ? cs[2] : cs[3];
                            const u64 word = readlane64(m, fl);
                            row = (fl & 15) * 64 + __ffsll((long long)word) - 1;
                            break;
                        }
                        if (cs[3] < 0) break;
                    }
                    if (col < 0) {
                        if (found_in_sweep) { found_in_sweep = false; from = 0; continue; }
                        action = 2; break;
                    }
                }
                const int sc = S.starColOfRow[row];
                if (lane == 0) S.primeColOfRow[row] = (short)col;      // prime zero (:255)
                if (sc < 0) {
                    // ---------- step 4 (:283-334) ----------
                    n_s4++;
                    int last = col;
                    if (lane == 0) {
                        int cr = row, cc = col;
                        for (int it = 0; it <= nR + nC; it++) {
                            const int old_r = S.starRowOfCol[cc];
                            S.starColOfRow[cr] = (short)cc; S.starRowOfCol[cc] = (short)cr;
                            if (old_r < 0) break;
                            cc = S.primeColOfRow[old_r]; cr = old_r;
                            if (cc < 0) break;
                        }
                        last = cc;
                    }
                    last = __builtin_amdgcn_readfirstlane(last);
                    if (lane < MK_MAXW) { u64 t = cR16; while (t) { const int r2 = lane * 64 + __ffsll((long long)t) - 1; S.primeColOfRow[r2] = -1; t &= t - 1; } }
                    if (lane == 0) S.primeColOfRow[row] = -1;
                    cR16 = 0; covRany = false;
                    cC |= phaseUnc; if (last >= 0 && lane == (last >> 6)) cC |= 1ull << (last & 63);
                    phaseUnc = 0;
                    if (lane < MK_MAXW) { S.covR[lane] = 0; S.covC[lane] = cC; }
                    int total = 0;
                    for (int w = 0; w < wordsC; w++) total += __popcll(readlane64(cC, w));
                    from = 0; found_in_sweep = false;
                    if (total == minDim) { action = 3; break; }
                    if (!eager) { if (lane == 0) S.flag[F_LAST] = last; action = 5; break; }   // S.crows still lists the rows that were covered
                    continue;
                }
                if ((lane & 15) == (row >> 6)) cR16 |= 1ull << (row & 63);                       // :270
                covRany = true; n_find++;
                if (lane < MK_MAXW) {
                    if (lane == (sc >> 6)) { cC &= ~(1ull << (sc & 63)); phaseUnc |= 1ull << (sc & 63); }   // :271
                    S.covR[lane] = cR16; S.covC[lane] = cC;
                }
                found_in_sweep = true;
                from = col + 1;
                if (!eager) {
                    if (lane == 0) {
                        const int n = S.flag[F_NCR];
                        if (n < LZ_MAXCOV + 8) S.crows[n] = (unsigned short)row;
                        S.flag[F_NCR] = n + 1; S.flag[F_ROW] = row; S.flag[F_COL] = col; S.flag[F_SC] = sc;
                    }
                    action = 4; break;
                }
            }
            if (action == 2 && eager) { const int ncu = wave_list_bits(~cC & vC, 0, S.list, lane); if (lane == 0) S.flag[F_AUX] = ncu; }
            if (lane == 0) S.flag[F_ACTION] = action;
        }
        __syncthreads();
        const int action = S.flag[F_ACTION];
        const long long t_b = wall_clock64();
        t_s3 += t_b - t_a;
        if (action == 3) { done = true; break; }
        const bool eager = S.flag[F_EAGER] != 0;
        if (action == 2) {
            n_s5++;
            if (eager) step5_eager(X);
            else {
                // ================= step 5, lazy =================
                if (S.flag[F_K] >= LZ_LOG || S.flag[F_NID] + LZ_MAXCOV > 64) flush_all(X, true);   // log (or row-id table) full: apply it everywhere and restart
                const int K = S.flag[F_K], nE = S.flag[F_NE], ncr = S.flag[F_NCR];
                const int r = tid;
                const bool rowcov = (S.covR[wave] >> lane) & 1;
                const size_t rclamp = (size_t)min(r, nR - 1);
                u64 best = ~0ull;
                if (tid < nC && bit1024(S.lazyM, tid)) best = S.u.cminkey[tid];
                double v[LZ_MAXE];
                if (nE > 0) {
#pragma unroll
                    for (int k = 0; k < LZ_MAXE; k++) v[k] = (k < 2 || nE > 2) ? d[rclamp + (size_t)nR * S.elist[min(k, nE - 1)]] : 0.0;   // uniform guard: 2 or 8 loads in flight
#pragma unroll
                    for (int k = 0; k < LZ_MAXE; k++) if (k < nE && r < nR && !rowcov) { const u64 kk = dkey(v[k]); if (kk < best) best = kk; }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { const u64 o = __shfl_xor(best, off); if (o < best) best = o; }
                if (lane == 0) S.redk[wave] = best;
                if (tid == 0) S.flag[F_CNT] = 0;
                __syncthreads();
                best = S.redk[0];
#pragma unroll
                for (int w = 1; w < 16; w++) { const u64 o = S.redk[w]; if (o < best) best = o; }
                const double h = dunkey(best);
                if (tid == 0) {
                    S.logH[K] = h;
                    for (int q = 0; q < ncr; q++) {
                        const int rr = S.crows[q];
                        int id;
                        if (bit1024(S.everCov, rr)) id = S.covid[rr];
                        else { id = S.flag[F_NID]++; S.covid[rr] = (unsigned char)id; S.rowmask[id][0] = 0; S.rowmask[id][1] = 0; S.everCov[rr >> 6] |= 1ull << (rr & 63); }
                        S.rowmask[id][K >> 6] |= 1ull << (K & 63);
                    }
                }
                // eager columns: the reference's elementwise update (covered rows +h, then -h: the column is uncovered)
                if (nE > 0) {
#pragma unroll
                    for (int k = 0; k < LZ_MAXE; k++) {
                        if (k < nE) {
                            const int c = S.elist[k];
                            bool z = false;
                            if (r < nR) { const double nv = rowcov ? (v[k] + h) - h : v[k] - h; d[(size_t)r + (size_t)nR * c] = nv; z = fabs(nv) < DBL_EPSILON; }
                            const u64 bal = __ballot(z);
                            if (lane == 0) S.bm[c * MK_MAXW + wave] = bal;
                        }
                    }
                }
                // lazy columns: cminU -= h; the ones that reach zero get materialised
                if (tid < nC && bit1024(S.lazyM, tid)) {
                    const double t = dunkey(S.u.cminkey[tid]) - h;
                    S.u.cminkey[tid] = dkey(t);
                    if (fabs(t) < DBL_EPSILON) { const int q = atomicAdd(&S.flag[F_CNT], 1); S.list[q] = (unsigned short)tid; }
                }
                __syncthreads();
                if (tid == 0) S.flag[F_K] = K + 1;
                if (tid < nE) {
                    const int c = S.elist[tid];
                    S.applied[c] = (unsigned short)(K + 1);
                    bool any = false;
                    for (int w = 0; w < wordsR; w++) any |= S.bm[c * MK_MAXW + w] != 0;
                    hz_set(S, c, any);
                }
                __syncthreads();
                const int nhot = S.flag[F_CNT];
                n_hot += nhot;
                if (nhot) materialize_cols(X, S.list, nhot, false);
            }
            t_s5 += wall_clock64() - t_b;
        } else if (action == 4) {
            // ======= a row was covered and its star column uncovered (:270-271): lazy bookkeeping =======
            const int row = S.flag[F_ROW], sc = S.flag[F_SC];
            if (tid == 0) S.flag[F_CNT] = 0;
            __syncthreads();
            // lazy columns holding a zero in the newly covered row become eager (their zero now sits in a covered row)
            if (tid < nC && bit1024(S.lazyM, tid) && ((S.bm[tid * MK_MAXW + (row >> 6)] >> (row & 63)) & 1)) {
                const int q = atomicAdd(&S.flag[F_CNT], 1); S.list[1 + q] = (unsigned short)tid;
            }
            __syncthreads();
            const int nmov = S.flag[F_CNT];
            const bool overflow = (S.flag[F_NE] + 1 + nmov > LZ_MAXE) || (S.flag[F_NCR] > LZ_MAXCOV);
            if (overflow) go_eager(X);
            else {
                if (tid == 0) {
                    S.list[0] = (unsigned short)sc;
                    int ne = S.flag[F_NE];
                    S.elist[ne++] = (unsigned short)sc;
                    for (int q = 0; q < nmov; q++) { const int c = S.list[1 + q]; S.elist[ne++] = (unsigned short)c; S.lazyM[c >> 6] &= ~(1ull << (c & 63)); }
                    S.flag[F_NE] = ne;
                }
                __syncthreads();
                materialize_cols(X, S.list, 1 + nmov, false);          // sc always has pending covered-row updates to check; movers only if stale
                // lazy columns whose minimum sat in the newly covered row: rescan over the uncovered rows
                if (tid == 0) S.flag[F_CNT] = 0;
                __syncthreads();
                if (tid < nC && bit1024(S.lazyM, tid) && S.argrow[tid] == row) { const int q = atomicAdd(&S.flag[F_CNT], 1); S.list[q] = (unsigned short)tid; }
                __syncthreads();
                const int nres = S.flag[F_CNT];
                if (nres) materialize_cols(X, S.list, nres, true);
            }
            t_h4 += wall_clock64() - t_b;
        } else if (action == 5) {
            // ======= augmented (step 4): star time of the new column, rows uncovered again =======
            const int last = S.flag[F_LAST], K = S.flag[F_K], ncr = min(S.flag[F_NCR], LZ_MAXCOV);
            if (tid == 0) {
                if (last >= 0) { if (S.startime[last] == 0xFFFF) S.startime[last] = (unsigned short)K; S.lazyM[last >> 6] &= ~(1ull << (last & 63)); }
                int n = 0;
                const int ne = S.flag[F_NE];
                for (int q = 0; q < ne; q++) { const int c = S.elist[q]; if (S.starRowOfCol[c] < 0) { S.lazyM[c >> 6] |= 1ull << (c & 63); S.list[n++] = (unsigned short)c; } }
                S.flag[F_NE] = 0; S.flag[F_CNT] = n;
            }
            __syncthreads();
            const int nback = S.flag[F_CNT];
            if (nback) materialize_cols(X, S.list, nback, true);       // never-starred eager columns return to the lazy set
            // (old covered rows) x (lazy columns): true value -> zero bit, cminU, argrow
            if (ncr > 0 && tid < nC && bit1024(S.lazyM, tid)) {
                double xs[LZ_MAXCOV];
#pragma unroll
                for (int i = 0; i < LZ_MAXCOV; i++) xs[i] = d[(size_t)S.crows[min(i, ncr - 1)] + (size_t)nR * tid];   // all loads in flight
#pragma unroll
                for (int i = 0; i < LZ_MAXCOV; i++) {
                    if (i < ncr) {
                        const int rr = S.crows[i];
                        const double x = replay(S, xs[i], tid, rr, K);
                        const u64 ki = dkey(x);
                        const bool z = fabs(x) < DBL_EPSILON;
                        u64& w = S.bm[tid * MK_MAXW + (rr >> 6)];
                        if (z) { w |= 1ull << (rr & 63); hz_set(S, tid, true); } else w &= ~(1ull << (rr & 63));
                        if (ki < S.u.cminkey[tid]) { S.u.cminkey[tid] = ki; S.argrow[tid] = rr; }
                    }
                }
            }
            if (tid == 0) S.flag[F_NCR] = 0;
            __syncthreads();
            t_h5 += wall_clock64() - t_b;
        }
        if (wave == 0) hz = (lane < MK_MAXW) ? S.hz[lane] : 0;
        if (++guard > 16 * MK_MAXN * MK_MAXN) break;
    }
    __syncthreads();
    if (tid == 0) {
        stat[0] = n_s4; stat[1] = n_s5; stat[2] = n_sw; stat[3] = n_find; stat[14] = n_hot; stat[15] = S.flag[F_EAGER];
        stat[8] = (int)(t_init - t_begin); stat[9] = (int)t_s3; stat[10] = (int)(t_h4 + t_h5); stat[11] = (int)t_s5; stat[12] = (int)(wall_clock64() - t_begin);
    }
    for (int r = tid; r < nR; r += MK_THREADS) a.ws.assignment[r] = S.starColOfRow[r];
    if (want_cost) {
        double* vals = reinterpret_cast<double*>(S.bm);
        __syncthreads();
        for (int r = tid; r < nR; r += MK_THREADS) { const int c = S.starColOfRow[r]; vals[r] = (c >= 0) ? elem_cost(a, r, c, nR, rowsTrk) : 0.0; }
        __syncthreads();
        if (tid == 0) { double cst = 0.0; for (int r = 0; r < nR; r++) if (S.starColOfRow[r] >= 0) cst += vals[r]; *a.ws.cost = cst; }
    }
}

} // namespace

hipError_t launch_munkres_lazy(const assoc::AssocArgs& a, int want_cost, hipStream_t s)
{
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(munkres_lazy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LzShared));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(munkres_lazy_kernel, dim3(1), dim3(MK_THREADS), sizeof(LzShared), s, a, want_cost);
    return hipGetLastError();
}
