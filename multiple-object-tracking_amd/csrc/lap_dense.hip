// lap_dense.hip -- the assignment fast path's DENSE solver: what certifies a frame with far matches.
//
// A frame with a missed detection AND a false positive forces far matches (every line of the smaller side must be assigned,
// hungarian.cpp:29-368).  The optimal prices then form "cones" around the far-matched rows that the sparse solver
// (lap_kernels.hip: K = 8 nearest candidates per row) cannot express: it gives up or its prices fail the dense check.
// This file solves such a frame exactly: a Jonker-Volgenant shortest-augmenting-path solver over ALL n x n entries --
//   lap_cost_rm_kernel   the cost matrix, row-major, into the association workspace (chip-wide, ~20 us);
//   lap_dense_kernel     one 1024-thread workgroup, thread = column: every row takes its nearest column if it is the lowest
//                        claimant (tight under v = 0), then one Dijkstra search per free row over the whole matrix (one
//                        coalesced 8 KB row read, one relaxation and one workgroup arg-min per settled column), dual update,
//                        augmentation; u, v and the matching go where the sparse solver would have put them;
// -- and hands the result to the SAME dense check and uniqueness certificate as the sparse solver's (lap_verify_kernel,
// lap_certify.h).  Nothing is trusted: a wrong matching or infeasible price fails the check and the frame goes to the order-exact
// emulation as before.  Both kernels return at once unless the sparse solver gave up or its prices failed the check, and the
// host launches them only for streams that recently needed them (AssocWs::dense_hint).
//
// Cost of a 1000 x 1000 frame with 4 % misses + 3 % false positives: 90-160 free rows after the greedy start, 3,700-4,700
// settled columns in all (a far-matched row settles its whole cone: up to 380).
#include "assoc_common.h"
#include "mot_env.h"

using namespace assoc;

namespace {

struct DenseShared {
    double v[MK_MAXN], u[MK_MAXN], dist[MK_MAXN];
    u64 wkey[2][MK_THREADS / 64]; int widx[2][MK_THREADS / 64];     // per-wave arg-min of a step, double-buffered
    double red[MK_THREADS / 64];
    int claim[MK_MAXN];
    short pred[MK_MAXN], rowOfCol[MK_MAXN], colOfRow[MK_MAXN];
    unsigned short flist[MK_MAXN];
    int wave_tot[MK_THREADS / 64];
    int nfree;
};

// the dense kernels have work iff the sparse solver gave up (SOLVE == 1) or its prices failed the dense check
__device__ __forceinline__ bool dense_wanted(const LapWs& L)
{
    const int solve = L.hdr[LAP_H_SOLVE];
    return !L.hdr[LAP_H_BAD] && (solve == 1 || (solve == 0 && L.hdr[LAP_H_VIOL] != 0));
}

__global__ void __launch_bounds__(256) lap_cost_rm_kernel(AssocArgs a)
{
    __shared__ bbox_t colb[64];
    int nR, nC; bool rowsTrk; resolve_dims(a, nR, nC, rowsTrk);
    const LapWs& L = a.ws.lap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    if (nR <= 0 || nC <= 0 || nR > nC || r0 >= nR || c0 >= nC) return;
    if (!dense_wanted(L)) return;
    if (!a.user && threadIdx.x < 64 && c0 + (int)threadIdx.x < nC) colb[threadIdx.x] = rowsTrk ? a.det[c0 + threadIdx.x] : a.trk[c0 + threadIdx.x];
    __syncthreads();
    const int c = c0 + lane;                                           // lanes along the row: coalesced row-major stores
    for (int rr = wave; rr < 64; rr += 4) {
        const int r = r0 + rr;
        if (r >= nR || c >= nC) continue;
        double cst;
        if (a.user) cst = a.user[(size_t)r + (size_t)nR * c];
        else { const bbox_t rb = rowsTrk ? a.trk[r] : a.det[r]; cst = rowsTrk ? pair_cost(rb, colb[lane]) : pair_cost(colb[lane], rb); }
        a.ws.dist[(size_t)r * nC + c] = cst;
    }
}

__global__ void __launch_bounds__(MK_THREADS) lap_dense_kernel(AssocArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dn_raw[];
    DenseShared& S = *reinterpret_cast<DenseShared*>(dn_raw);
    int nR, nC; bool rowsTrk; resolve_dims(a, nR, nC, rowsTrk);
    const LapWs& L = a.ws.lap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (nR <= 0 || nC <= 0 || nR > nC) return;
    if (!dense_wanted(L)) return;
    const long long t_begin = wall_clock64();
    const double* __restrict__ C = a.ws.dist;                          // row-major [nR][nC] (lap_cost_rm_kernel)
    // ---- start: v = 0, u_i = row minimum, every row claims its nearest column, the lowest claimant gets it ----
    S.v[tid] = 0.0; S.claim[tid] = 0x7FFFFFFF; S.rowOfCol[tid] = -1; S.colOfRow[tid] = -1;
    __syncthreads();
    int j0 = 0; double rmin = 0.0;
    if (tid < nR) { j0 = L.ccol[(size_t)tid * LAP_K]; rmin = L.ccost[(size_t)tid * LAP_K]; S.u[tid] = rmin; atomicMin(&S.claim[j0], tid); }
    __syncthreads();
    bool isfree = false;
    if (tid < nR) { if (S.claim[j0] == tid) { S.colOfRow[tid] = (short)j0; S.rowOfCol[j0] = (short)tid; } else isfree = true; }
    {   // free rows, ascending
        const u64 bal = __ballot(isfree);
        if (lane == 0) S.wave_tot[wave] = __popcll(bal);
        __syncthreads();
        int off = 0, tot = 0;
        for (int w = 0; w < MK_THREADS / 64; w++) { const int t = S.wave_tot[w]; if (w < wave) off += t; tot += t; }
        if (isfree) S.flist[off + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)tid;
        if (tid == 0) S.nfree = tot;
        __syncthreads();
    }
    const int nfree = S.nfree;
    int steps = 0; bool failed = false;
    // ---- one shortest augmenting path per free row ----
    for (int q = 0; q < nfree; q++) {
        const int s = S.flist[q];
        const double us = S.u[s];
        double myd = DBL_MAX; bool scanned = false; int mypred = s;
        if (tid < nC) myd = (C[(size_t)s * nC + tid] - us) - S.v[tid];
        int jend = -1; double dend = 0.0;
        for (int it = 0; it <= nC; it++) {
            // workgroup arg-min of the unsettled distances (lowest column on equal keys): one barrier, double-buffered partials
            const int pb = it & 1;
            u64 key = (tid < nC && !scanned) ? dkey(myd) : ~0ull;
            const u64 wk = wave_min_u64_dpp(key);
            const u64 hit = __ballot(key == wk);
            if (lane == 0) { S.wkey[pb][wave] = wk; S.widx[pb][wave] = wave * 64 + (__ffsll((long long)hit) - 1); }
            __syncthreads();
            // every wave reduces the 16 partials in its first 16 lanes (two LDS loads per wave); the lowest wave = lowest column wins
            const u64 k2 = lane < MK_THREADS / 64 ? S.wkey[pb][lane] : ~0ull;
            const int i2 = lane < MK_THREADS / 64 ? S.widx[pb][lane] : 0;
            const u64 bk = wave_min_u64_dpp(k2);
            const u64 hitw = __ballot(k2 == bk);
            const int bj = __builtin_amdgcn_readlane(i2, __ffsll((long long)hitw) - 1);
            if (bk == ~0ull) { jend = -2; break; }                     // cannot happen while a free column exists
            const double dj = dunkey(bk);
            const int i = S.rowOfCol[bj];
            if (i < 0) { jend = bj; dend = dj; break; }                // a free column: the path ends here
            steps++;
            if (tid == bj) { scanned = true; S.dist[bj] = dj; }        // settled: its distance is final (the dual update needs it)
            if (tid < nC && !scanned) {
                const double nd = ((dj + C[(size_t)i * nC + tid]) - S.u[i]) - S.v[tid];
                if (nd < myd) { myd = nd; mypred = i; }
            }
        }
        if (jend < 0) { failed = true; break; }                           // (uniform)
        // dual update (settled columns and their rows, the start row), then the augmentation along the predecessors
        S.pred[tid] = (short)mypred;
        if (scanned) { const double delta = dend - S.dist[tid]; const int i = S.rowOfCol[tid]; S.u[i] += delta; S.v[tid] -= delta; }
        if (tid == 0) S.u[s] += dend;
        __syncthreads();
        if (tid == 0) {
            int j = jend;
            for (int guard = 0; guard <= nR; guard++) {
                const int i = S.pred[j];
                S.rowOfCol[j] = (short)i;
                const int jn = S.colOfRow[i];
                S.colOfRow[i] = (short)j;
                if (i == s) break;
                j = jn;
            }
        }
        __syncthreads();
    }
    // ---- results where the sparse solver puts them; Gamma = sum_i (c[i][M(i)] - rowmin_i) and the margins (lap_kernels.hip) ----
    double g = 0.0;
    if (tid < nR) {
        const int j = S.colOfRow[tid];
        // u from the cost of the matched entry: tight by construction, like the sparse solver's (the search's own u has accumulated
        // roundings); the dense check then decides about feasibility everywhere else
        const double cm = j >= 0 ? C[(size_t)tid * nC + j] : 0.0;
        L.u[tid] = j >= 0 ? cm - S.v[j] : S.u[tid];
        if (j >= 0) g = cm - rmin;
    }
    L.v[tid] = S.v[tid]; L.colOfRow[tid] = S.colOfRow[tid]; L.rowOfCol[tid] = S.rowOfCol[tid];
#pragma unroll
    for (int off2 = 32; off2 > 0; off2 >>= 1) g += __shfl_xor(g, off2);
    if (lane == 0) S.red[wave] = g;
    __syncthreads();
    if (tid == 0) {
        double gamma = 0.0;
        for (int w = 0; w < MK_THREADS / 64; w++) gamma += S.red[w];
        const double cmax = L.dhdr[3];                                 // from the sparse solver's pass over the row maxima
        const double mag = cmax + gamma;
        const double n3 = (double)nC * (double)nC * (double)nC;
        L.dhdr[0] = fmax(1e-9, 1e-15 * n3) * mag;                      // eps, tol: as in lap_solve_kernel
        L.dhdr[1] = 1e-12 * mag;
        L.dhdr[2] = gamma;
        L.hdr[LAP_H_NEDGES] = 0; L.hdr[LAP_H_VIOL] = 0;
        L.hdr[LAP_H_SOLVE] = failed ? 1 : 0;
        L.hdr[LAP_H_DENSE] = 1;                                        // the second dual check runs; the verdict reports "dense solver"
        L.hdr[LAP_H_DSTAT] = steps; L.hdr[LAP_H_DSTAT + 1] = nfree; L.hdr[LAP_H_DSTAT + 2] = (int)(wall_clock64() - t_begin);
    }
}

} // namespace

hipError_t launch_lap_verify_again(const AssocArgs& a, int gR, int gC, hipStream_t s);   // lap_kernels.hip

// between the sparse solver's dual check and the certificate: cost matrix, dense solver, dual check of its result
hipError_t launch_lap_dense(const AssocArgs& a, int gR, int gC, hipStream_t s)
{
    hipError_t e = mot_impl::func_lds_once(reinterpret_cast<const void*>(lap_dense_kernel), (int)sizeof(DenseShared)); if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lap_cost_rm_kernel, dim3(gR, gC), dim3(256), 0, s, a);
    mot_impl::lds_poison(s);                                           // (debug) MOT_LDS_POISON
    hipLaunchKernelGGL(lap_dense_kernel, dim3(1), dim3(MK_THREADS), sizeof(DenseShared), s, a);
    return launch_lap_verify_again(a, gR, gC, s);
}
