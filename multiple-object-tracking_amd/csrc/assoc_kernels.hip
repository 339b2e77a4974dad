// assoc_kernels.hip -- association cost matrix (top/td.cpp:386-457) and an
// order-exact device emulation of the reference's Munkres
// (trackers/hungarian/hungarian.cpp:29-368).
//
// Why emulate instead of solving: the reference's result depends on its scan
// orders whenever optimal assignments tie, and on its own float64 update
// sequence through the absolute zero test fabs(x) < DBL_EPSILON.  The device
// path therefore keeps the reference's state machine and element arithmetic
// and only parallelises *inside* each step:
//   * O(n^2) preparation (costs, row/column minimum, first zero bitmaps) runs
//     grid-wide in tiles of 64x64;
//   * the sequential part runs in ONE 1024-thread workgroup whose LDS holds the
//     zero structure as bitmaps (n <= 1024 -> 128 KB), so "first uncovered
//     zero in column-major order" is a ballot + ctz instead of a scan;
//   * step 5 touches only covered rows and uncovered columns.
#include "mot_dev.h"
#include <float.h>

namespace {

#define MK_MAXN 1024
#define MK_MAXW 16                  // 64-bit words per bitmap line
#define MK_THREADS 1024

typedef unsigned long long u64;

__device__ __forceinline__ u64 dkey(double v) { u64 b = (u64)__double_as_longlong(v); return (b >> 63) ? ~b : (b | 0x8000000000000000ull); }
__device__ __forceinline__ double dunkey(u64 k) { u64 b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k; return __longlong_as_double((long long)b); }

// td.cpp:407-419
__device__ __forceinline__ double pair_cost(const bbox_t a, const bbox_t d)
{
    const int cxi = (a.l + a.r) >> 1, cyi = (a.t + a.b) >> 1;
    const int cxj = (d.l + d.r) >> 1, cyj = (d.t + d.b) >> 1;
    double dist = 0.0;
    dist += sqrt((double)((cxi - cxj) * (cxi - cxj) + (cyi - cyj) * (cyi - cyj))) * (1.0 / ((double)MOT_FRAME_W));
    if (a.type != d.type) dist += 1.0;
    return dist;
}

struct AssocArgs {
    const bbox_t* trk; const bbox_t* det; const int* nT_dev; int nT; int nD;
    const double* user; int userR, userC;
    AssocWs ws;
    u64* linemin;        // [MK_MAXN] keys
    int* dims;           // [4]: nR, nC, rowsAreTrackers, minIsPerRow
    double* cost_only;   // assoc_cost_kernel output
};

__device__ __forceinline__ void resolve_dims(const AssocArgs& a, int& nR, int& nC, bool& rowsTrk)
{
    if (a.user) { nR = a.userR; nC = a.userC; rowsTrk = true; return; }
    const int nT = a.nT_dev ? *a.nT_dev : a.nT;
    if (nT < a.nD) { nR = nT; nC = a.nD; rowsTrk = true; }            // td.cpp:388,462-465
    else { nR = a.nD; nC = nT; rowsTrk = false; }
}

__device__ __forceinline__ double elem_cost(const AssocArgs& a, int r, int c, int nR, bool rowsTrk)
{
    if (a.user) return a.user[(size_t)r + (size_t)nR * c];
    return rowsTrk ? pair_cost(a.trk[r], a.det[c]) : pair_cost(a.trk[c], a.det[r]);
}

// pass 1: per-line minimum (rows if nR <= nC, hungarian.cpp:69-81; else columns, :107-119)
__global__ void __launch_bounds__(256) assoc_min_kernel(AssocArgs a)
{
    int nR, nC; bool rowsTrk; resolve_dims(a, nR, nC, rowsTrk);
    const int r = blockIdx.x * 64 + (threadIdx.x & 63);
    const int c0 = blockIdx.y * 64, wave = threadIdx.x >> 6;
    if (blockIdx.x * 64 >= nR || c0 >= nC) return;
    const bool perRow = nR <= nC;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { a.dims[0] = nR; a.dims[1] = nC; a.dims[2] = rowsTrk; a.dims[3] = perRow; }
    u64 best = ~0ull;
    for (int cc = wave; cc < 64; cc += 4) {
        const int c = c0 + cc;
        if (c >= nC) break;
        u64 k = ~0ull;
        if (r < nR) k = dkey(elem_cost(a, r, c, nR, rowsTrk));
        if (perRow) { if (k < best) best = k; }
        else {
            // column minimum over this tile's 64 rows
            u64 m = k;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { u64 o = __shfl_down(m, off); if (o < m) m = o; }
            if ((threadIdx.x & 63) == 0) atomicMin(&a.linemin[c], m);
        }
    }
    if (perRow && r < nR && best != ~0ull) atomicMin(&a.linemin[r], best);
}

// pass 2: working matrix d = cost - linemin, zero bitmaps in both orientations
__global__ void __launch_bounds__(256) assoc_sub_kernel(AssocArgs a)
{
    __shared__ unsigned int zr_lo[64], zr_hi[64];
    int nR, nC; bool rowsTrk; resolve_dims(a, nR, nC, rowsTrk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 64 + lane, c0 = blockIdx.y * 64;
    if (blockIdx.x * 64 >= nR || c0 >= nC) return;
    const bool perRow = nR <= nC;
    const int wordsR = (nR + 63) >> 6, wordsC = (nC + 63) >> 6;
    if (threadIdx.x < 64) { zr_lo[threadIdx.x] = 0; zr_hi[threadIdx.x] = 0; }
    __syncthreads();
    const double rmin = (perRow && r < nR) ? dunkey(a.linemin[r]) : 0.0;
    for (int cc = wave; cc < 64; cc += 4) {
        const int c = c0 + cc;
        if (c >= nC) break;
        bool z = false;
        if (r < nR) {
            const double v = elem_cost(a, r, c, nR, rowsTrk);
            const double d = v - (perRow ? rmin : dunkey(a.linemin[c]));
            a.ws.dist[(size_t)r + (size_t)nR * c] = d;
            z = fabs(d) < DBL_EPSILON;
        }
        const u64 bal = __ballot(z);
        if (lane == 0) a.ws.zc[(size_t)c * wordsR + blockIdx.x] = bal;
        if (z) { if (cc < 32) atomicOr(&zr_lo[lane], 1u << cc); else atomicOr(&zr_hi[lane], 1u << (cc - 32)); }
    }
    __syncthreads();
    if (threadIdx.x < 64 && r < nR) a.ws.zr[(size_t)r * wordsC + blockIdx.y] = ((u64)zr_hi[threadIdx.x] << 32) | zr_lo[threadIdx.x];
}

// ---------------------------------------------------------------------------
// the sequential state machine, one workgroup
// ---------------------------------------------------------------------------
struct MkShared {
    u64 bm[MK_MAXN * MK_MAXW];      // zero bitmap, [line][W]
    short starColOfRow[MK_MAXN];
    short starRowOfCol[MK_MAXN];
    short primeColOfRow[MK_MAXN];
    unsigned short list[MK_MAXN];
    u64 covR[MK_MAXW], covC[MK_MAXW], validR[MK_MAXW], validC[MK_MAXW];
    double red[MK_THREADS / 64];
    int flag[8];
};

__device__ __forceinline__ int wave_first_bit(u64 m, int lane, int nwords)
{   // lanes < nwords hold words of a line; returns index of the first set bit, or -1
    const u64 bal = __ballot(lane < nwords && m != 0);
    if (!bal) return -1;
    const int w = __ffsll((long long)bal) - 1;
    const u64 word = __shfl(m, w);
    return w * 64 + (__ffsll((long long)word) - 1);
}

// list of uncovered columns >= from, ascending; executed by wave 0.  returns count
__device__ int build_uncovered_cols(MkShared& S, int from, int wordsC, int lane)
{
    u64 w = 0;
    if (lane < wordsC) {
        w = ~S.covC[lane] & S.validC[lane];
        const int fw = from >> 6;
        if (lane < fw) w = 0;
        else if (lane == fw) w &= ~0ull << (from & 63);
    }
    int cnt = __popcll(w), pre = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(pre, off); if (lane >= off) pre += t; }
    const int total = __shfl(pre, 63);
    int pos = pre - cnt;
    while (w) { const int b = __ffsll((long long)w) - 1; S.list[pos++] = (unsigned short)(lane * 64 + b); w &= w - 1; }
    __threadfence_block();                                            // list is read by other lanes of this wave
    return total;
}

__global__ void __launch_bounds__(MK_THREADS) munkres_kernel(AssocArgs a, int want_cost)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char mk_raw[];
    MkShared& S = *reinterpret_cast<MkShared*>(mk_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int nR, nC; bool rowsTrk; resolve_dims(a, nR, nC, rowsTrk);
    double* __restrict__ d = a.ws.dist;
    int* stat = a.ws.status;
    if (tid < 4) stat[tid] = 0;
    if (nR <= 0 || nC <= 0) { if (tid == 0) *a.ws.cost = 0.0; for (int r = tid; r < max(nR, 0); r += MK_THREADS) a.ws.assignment[r] = -1; return; }
    const int wordsR = (nR + 63) >> 6, wordsC = (nC + 63) >> 6;
    const bool perRow = nR <= nC;
    const int minDim = perRow ? nR : nC;

    for (int i = tid; i < MK_MAXN; i += MK_THREADS) { S.starColOfRow[i] = -1; S.starRowOfCol[i] = -1; S.primeColOfRow[i] = -1; }
    if (tid < MK_MAXW) {
        S.covR[tid] = 0; S.covC[tid] = 0;
        S.validR[tid] = (tid < wordsR) ? ((tid == wordsR - 1 && (nR & 63)) ? ((1ull << (nR & 63)) - 1) : ~0ull) : 0;
        S.validC[tid] = (tid < wordsC) ? ((tid == wordsC - 1 && (nC & 63)) ? ((1ull << (nC & 63)) - 1) : ~0ull) : 0;
    }
    // ---- steps 1 + 2a: initial stars (hungarian.cpp:93-101 / :128-139) ----
    // lines = rows (perRow) scanned in order, each takes its first zero whose cross line is still free
    {
        const int nL = perRow ? nR : nC, W = perRow ? wordsC : wordsR;
        const u64* src = perRow ? a.ws.zr : a.ws.zc;
        for (int i = tid; i < nL * W; i += MK_THREADS) S.bm[(i / W) * MK_MAXW + (i % W)] = src[i];
        __syncthreads();
        if (wave == 0) {
            u64 taken = 0;                                            // lane w holds word w of the taken-cross-line mask
            for (int l0 = 0; l0 < nL; l0 += 8) {
                u64 m[8];
#pragma unroll
                for (int k = 0; k < 8; k++) m[k] = (lane < W && l0 + k < nL) ? S.bm[(l0 + k) * MK_MAXW + lane] : 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (l0 + k >= nL) break;
                    const u64 mm = m[k] & ~taken;
                    const int x = wave_first_bit(mm, lane, W);
                    if (x >= 0) {
                        if (lane == (x >> 6)) taken |= 1ull << (x & 63);
                        if (lane == 0) {
                            const int row = perRow ? (l0 + k) : x, col = perRow ? x : (l0 + k);
                            S.starColOfRow[row] = (short)col; S.starRowOfCol[col] = (short)row;
                        }
                    }
                }
            }
        }
        __syncthreads();
        // covered columns = starred columns (both branches end with exactly that; rows uncovered :138-139)
        if (tid < wordsC) {
            u64 w = 0;
            for (int b = 0; b < 64; b++) { const int c = tid * 64 + b; if (c < nC && S.starRowOfCol[c] >= 0) w |= 1ull << b; }
            S.covC[tid] = w;
        }
        __syncthreads();
    }
    // load the column-major bitmap for the main loop
    auto count_cov = [&]() { int n = 0; for (int w = 0; w < wordsC; w++) n += __popcll(S.covC[w]); return n; };
    bool done = (count_cov() == minDim);                               // step 2b (:216-237)
    if (!done) {
        __syncthreads();
        for (int i = tid; i < nC * wordsR; i += MK_THREADS) S.bm[(i / wordsR) * MK_MAXW + (i % wordsR)] = a.ws.zc[i];
        __syncthreads();
    }
    int guard = 0;
    while (!done) {
        // ================= step 3 (:240-280), wave 0 =================
        if (wave == 0) {
            int action = 0;                                           // 1: augmented (go to 2a), 2: no zeros (go to 5)
            bool zerosFound = true;
            while (action == 0) {
                if (!zerosFound) { action = 2; break; }
                zerosFound = false;
                if (lane == 0) atomicAdd(&stat[2], 1);
                int from = 0;
                int cnt = build_uncovered_cols(S, from, wordsC, lane);
                int pos = 0;
                while (pos < cnt) {
                    const int c = (pos + lane < cnt) ? S.list[pos + lane] : -1;
                    bool hit = false;
                    if (c >= 0) for (int w = 0; w < wordsR; w++) hit |= (S.bm[c * MK_MAXW + w] & ~S.covR[w]) != 0;
                    const u64 bal = __ballot(hit);
                    if (!bal) { pos += 64; continue; }
                    const int fl = __ffsll((long long)bal) - 1;
                    const int col = __shfl(c, fl);
                    const u64 mw = (lane < wordsR) ? (S.bm[col * MK_MAXW + lane] & ~S.covR[lane]) : 0;
                    const int row = wave_first_bit(mw, lane, wordsR);
                    const int sc = S.starColOfRow[row];
                    if (lane == 0) S.primeColOfRow[row] = (short)col;  // prime zero (:255)
                    if (sc < 0) {
                        // ---------- step 4 (:283-334): augment along the star/prime path ----------
                        if (lane == 0) {
                            atomicAdd(&stat[0], 1);
                            int cr = row, cc = col;
                            for (;;) {
                                const int old_r = S.starRowOfCol[cc];
                                S.starColOfRow[cr] = (short)cc; S.starRowOfCol[cc] = (short)cr;
                                if (old_r < 0) break;
                                const int pc = S.primeColOfRow[old_r];
                                cr = old_r; cc = pc;
                            }
                        }
                        action = 1;
                        break;
                    }
                    if (lane == 0) { S.covR[row >> 6] |= 1ull << (row & 63); S.covC[sc >> 6] &= ~(1ull << (sc & 63)); } // :270-271
                    __threadfence_block();
                    zerosFound = true;
                    cnt = build_uncovered_cols(S, col + 1, wordsC, lane);  // the sweep continues with the next column (:273)
                    pos = 0;
                }
            }
            if (lane == 0) S.flag[0] = action;
        }
        __syncthreads();
        const int action = S.flag[0];
        if (action == 1) {
            // delete primes, uncover rows (:324-330); step 2a (:198-209): cover starred columns
            for (int i = tid; i < nR; i += MK_THREADS) S.primeColOfRow[i] = -1;
            if (tid < MK_MAXW) S.covR[tid] = 0;
            __syncthreads();
            if (tid < wordsC) {
                u64 w = S.covC[tid];
                for (int b = 0; b < 64; b++) { const int c = tid * 64 + b; if (c < nC && S.starRowOfCol[c] >= 0) w |= 1ull << b; }
                S.covC[tid] = w;
            }
            __syncthreads();
            done = (count_cov() == minDim);
        } else {
            // ================= step 5 (:337-368) =================
            if (tid == 0) atomicAdd(&stat[1], 1);
            // h = min over uncovered rows x uncovered columns
            int ncu = 0;
            if (wave == 0) { ncu = build_uncovered_cols(S, 0, wordsC, lane); if (lane == 0) S.flag[1] = ncu; }
            __syncthreads();
            ncu = S.flag[1];
            double h = DBL_MAX;
            for (int r = tid; r < nR; r += MK_THREADS) {
                if ((S.covR[r >> 6] >> (r & 63)) & 1) continue;
                for (int k = 0; k < ncu; k++) { const double v = d[(size_t)r + (size_t)nR * S.list[k]]; if (v < h) h = v; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_down(h, off); if (o < h) h = o; }
            if (lane == 0) S.red[wave] = h;
            __syncthreads();
            h = S.red[0];
            for (int w = 1; w < MK_THREADS / 64; w++) if (S.red[w] < h) h = S.red[w];
            // (a) uncovered rows in uncovered columns: d -= h ; bitmap bits of uncovered rows rebuilt per word.
            //     wave-task = (64-row word, 4 columns) so four independent loads are in flight per lane
            {
                const int ntask = wordsR * ((ncu + 3) >> 2);
                for (int t = wave; t < ntask; t += MK_THREADS / 64) {
                    const int w = t % wordsR, k0 = (t / wordsR) << 2;
                    const int r = w * 64 + lane;
                    const u64 cw = S.covR[w];
                    const bool mine = r < nR && !((cw >> lane) & 1);
                    double v[4]; int cs[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { cs[u] = (k0 + u < ncu) ? S.list[k0 + u] : -1; v[u] = (mine && cs[u] >= 0) ? d[(size_t)r + (size_t)nR * cs[u]] : 1.0; }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (cs[u] < 0) continue;
                        bool z = false;
                        if (mine) { const double nv = v[u] - h; d[(size_t)r + (size_t)nR * cs[u]] = nv; z = fabs(nv) < DBL_EPSILON; }
                        const u64 bal = __ballot(z);
                        if (lane == 0) S.bm[cs[u] * MK_MAXW + w] = (S.bm[cs[u] * MK_MAXW + w] & cw) | (bal & ~cw);
                    }
                }
            }
            __syncthreads();
            // (b) covered rows, every column: d += h, and -h again where the column is uncovered (same order as :355-364)
            for (int w = 0; w < wordsR; w++) {
                u64 rows = S.covR[w];
                while (rows) {
                    const int r = w * 64 + (__ffsll((long long)rows) - 1); rows &= rows - 1;
                    for (int c = tid; c < nC; c += MK_THREADS) {
                        double v = d[(size_t)r + (size_t)nR * c] + h;
                        if (!((S.covC[c >> 6] >> (c & 63)) & 1)) v -= h;
                        d[(size_t)r + (size_t)nR * c] = v;
                        const bool z = fabs(v) < DBL_EPSILON;
                        unsigned int* wp = reinterpret_cast<unsigned int*>(&S.bm[c * MK_MAXW + (r >> 6)]) + ((r & 63) >> 5);
                        const unsigned int bit = 1u << (r & 31);
                        const bool cur = (*wp & bit) != 0;
                        if (z != cur) { if (z) atomicOr(wp, bit); else atomicAnd(wp, ~bit); }
                    }
                }
            }
            __syncthreads();
        }
        if (++guard > 4 * MK_MAXN * MK_MAXN) break;                    // cannot happen for finite costs
    }
    __syncthreads();
    // buildassignmentvector (:161-176) + computeassignmentcost (:179-189)
    for (int r = tid; r < nR; r += MK_THREADS) a.ws.assignment[r] = S.starColOfRow[r];
    if (want_cost) {
        double* vals = reinterpret_cast<double*>(S.bm);                // bitmap no longer needed
        __syncthreads();
        for (int r = tid; r < nR; r += MK_THREADS) { const int c = S.starColOfRow[r]; vals[r] = (c >= 0) ? elem_cost(a, r, c, nR, rowsTrk) : 0.0; }
        __syncthreads();
        if (tid == 0) { double cst = 0.0; for (int r = 0; r < nR; r++) if (S.starColOfRow[r] >= 0) cst += vals[r]; *a.ws.cost = cst; }
    }
}

__global__ void assoc_cost_kernel(AssocArgs a)
{   // plain td.cpp:386-457 matrix, for mot_cost_matrix()
    int nR, nC; bool rowsTrk; resolve_dims(a, nR, nC, rowsTrk);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nR * nC) return;
    const int r = (int)(i % nR), c = (int)(i / nR);
    a.cost_only[i] = elem_cost(a, r, c, nR, rowsTrk);
}
} // namespace

hipError_t launch_cost_matrix(const bbox_t* trk, int nT, const bbox_t* det, int nD, double* dist_out, hipStream_t s)
{
    if (nT <= 0 || nD <= 0) return hipSuccess;
    AssocArgs a{}; a.trk = trk; a.det = det; a.nT_dev = nullptr; a.nT = nT; a.nD = nD; a.user = nullptr; a.cost_only = dist_out;
    const size_t n = (size_t)nT * nD;
    hipLaunchKernelGGL(assoc_cost_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_assoc(const AssocWs& ws, const bbox_t* trk, const int* nT_dev, int nT, const bbox_t* det, int nD,
                        const double* user_dist, int nR, int nC, int want_cost, hipStream_t s)
{
    AssocArgs a;
    a.trk = trk; a.det = det; a.nT_dev = nT_dev; a.nT = nT; a.nD = nD;
    a.user = user_dist; a.userR = nR; a.userC = nC;
    a.ws = ws;
    a.linemin = ws.linemin;
    a.dims = ws.status + 4;
    a.cost_only = nullptr;
    int maxR, maxC;
    if (user_dist) { maxR = nR; maxC = nC; }
    else { maxR = nT < nD ? nT : nD; maxC = nT < nD ? nD : nT; if (nT_dev) { maxR = nD < nT ? nD : nT; maxC = nD > nT ? nD : nT; } }
    if (maxR > MK_MAXN || maxC > MK_MAXN) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(a.linemin, 0xFF, sizeof(u64) * MK_MAXN, s);
    if (e != hipSuccess) return e;
    if (maxR > 0 && maxC > 0) {
        // nT_dev: the true nT is <= nT (the host-side upper bound); tiles outside exit early
        const int gR = ((nT_dev ? (nD > nT ? nD : nT) : maxR) + 63) / 64, gC = ((nT_dev ? (nD > nT ? nD : nT) : maxC) + 63) / 64;
        hipLaunchKernelGGL(assoc_min_kernel, dim3(gR, gC), dim3(256), 0, s, a);
        hipLaunchKernelGGL(assoc_sub_kernel, dim3(gR, gC), dim3(256), 0, s, a);
    }
    static bool attr_set = false;
    if (!attr_set) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(munkres_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MkShared));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(munkres_kernel, dim3(1), dim3(MK_THREADS), sizeof(MkShared), s, a, want_cost);
    return hipGetLastError();
}
