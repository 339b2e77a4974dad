// lap_kernels.hip -- assignment fast path in front of the order-exact Munkres emulation (assoc_kernels.hip).
//
// The reference's assignmentoptimal (trackers/hungarian/hungarian.cpp:29-368) returns SOME minimum-cost assignment;
// which one depends on its scan orders only when optima tie.  So whenever the optimum is provably unique -- with a
// margin above everything the reference's own float64 rounding can move -- any exact solver returns the reference's
// assignment.  Three kernels (rows <= columns, the only shape top/td.cpp:462-469 produces):
//   lap_rowscan_kernel  one wavefront per row: row minimum (hungarian.cpp:69-81) and the LAP_K smallest entries of the
//                       row, lowest column first on equal cost (candidate lists), largest cost of the matrix
//   lap_solve_kernel    ONE workgroup, everything in LDS: shortest-augmenting-path searches on the candidate graph.
//                       All free rows search concurrently (one lane each, 8 per wavefront) on the same snapshot of
//                       (prices, matching); a search commits iff it holds the lock -- lowest searcher id -- of every
//                       column it scanned and of its end column, which makes concurrent dual updates commute (proof
//                       sketch in DESIGN.md 4.3).  Row duals are implicit (u_i = c[i][M(i)] - v[M(i)]), so matched
//                       edges are tight by construction.  The solver is UNTRUSTED: whatever it returns is checked by
//   lap_verify_kernel   the dense pass (it also produces what the Munkres kernel needs if it has to run: working
//                       matrix d = cost - row minimum and the zero bitmaps, i.e. assoc_sub_kernel's work): reduced
//                       cost r = c - u_i - v_j of EVERY entry must be >= -tol, prices <= 0 and exactly 0 on free
//                       columns (dual feasibility + complementary slackness => optimal), and every entry with r < eps
//                       that is not matched is recorded as an edge "row i could take the column of row i'".
// The Munkres kernel then runs lap_certify (lap_certify.h): the near-tight digraph is acyclic <=> every other
// assignment costs at least eps more => the reference returns exactly this one.  Otherwise (a tie, or any doubt) the
// order-exact emulation runs as before.  lap_model.c (CPU model, test infrastructure) is the CPU model of this file; tests/test_lap_model.py
// fuzzes "certified => equal to the reference" on CPU, tests/test_gpu_parity.py on the device.
#include "assoc_common.h"

using namespace assoc;

namespace {

__device__ __forceinline__ u64 wave_min_u64(u64 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const u64 o = __shfl_xor(v, off); if (o < v) v = o; }
    return v;
}
__device__ __forceinline__ u64 wave_max_u64(u64 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const u64 o = __shfl_xor(v, off); if (o > v) v = o; }
    return v;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const unsigned o = __shfl_xor(v, off); if (o < v) v = o; }
    return v;
}

// ---- stage 1 ----------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) lap_rowscan_kernel(AssocArgs a)
{
    int nR, nC; bool rowsTrk; resolve_dims(a, nR, nC, rowsTrk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0) {                        // the housekeeping assoc_min_kernel does
        if (a.ws.ctl) { for (int i = 0; i < 2 * MK_HELPERS; i++) a.ws.ctl[CTL_PARTIAL + i * MK_PARTIAL_STRIDE] = MK_HSENT; for (int i = 0; i < 64; i++) a.ws.ctl[CTL_COV + i] = 0; }
        a.dims[0] = nR; a.dims[1] = nC; a.dims[2] = rowsTrk; a.dims[3] = nR <= nC;
    }
    const int r = blockIdx.x * 4 + wave;
    if (r >= nR || nC <= 0) return;                                   // wave-uniform
    const LapWs& L = a.ws.lap;
    bbox_t rb = {};
    if (!a.user) rb = rowsTrk ? a.trk[r] : a.det[r];
    u64 key[MK_MAXN / 64];
    u64 mx = 0; bool bad = false;
#pragma unroll
    for (int t = 0; t < MK_MAXN / 64; t++) {
        const int j = t * 64 + lane;
        key[t] = ~0ull;
        if (j < nC) {
            double cst;
            if (a.user) cst = a.user[(size_t)r + (size_t)nR * j];
            else cst = rowsTrk ? pair_cost(rb, a.det[j]) : pair_cost(a.trk[j], rb);
            bad |= !(cst >= 0.0 && cst <= DBL_MAX);
            key[t] = dkey(cst);
            if (key[t] > mx) mx = key[t];
        }
    }
    for (int k = 0; k < LAP_K; k++) {
        u64 lk = ~0ull; int lt = 0;
#pragma unroll
        for (int t = 0; t < MK_MAXN / 64; t++) if (key[t] < lk) { lk = key[t]; lt = t; }
        const u64 wm = wave_min_u64(lk);
        const unsigned mycol = (lk == wm && lk != ~0ull) ? (unsigned)(lt * 64 + lane) : 0xFFFFFFFFu;
        const unsigned wc = wave_min_u32(mycol);                      // equal cost: the lowest column
        if (lane == 0) {
            L.ccol[(size_t)r * LAP_K + k] = wc == 0xFFFFFFFFu ? (unsigned short)0xFFFF : (unsigned short)wc;
            L.ccost[(size_t)r * LAP_K + k] = wc == 0xFFFFFFFFu ? DBL_MAX : dunkey(wm);
            if (k == 0) a.linemin[r] = wm;                            // hungarian.cpp:69-81
        }
        if (mycol == wc && wc != 0xFFFFFFFFu) {
#pragma unroll
            for (int t = 0; t < MK_MAXN / 64; t++) if (t == lt) key[t] = ~0ull;
        }
    }
    mx = wave_max_u64(mx);
    if (lane == 0) atomicMax(L.cmaxkey, mx);
    if (__ballot(bad) && lane == 0) atomicOr(&L.hdr[LAP_H_BAD], 1);
}

// ---- stage 2 ----------------------------------------------------------------------------------------------------
struct LapShared {
    double cc[MK_MAXN * LAP_K];          // candidate costs, [row][k]
    double v[MK_MAXN];                   // column prices
    double sd[LAP_S * LAP_TS];           // per search: tentative distance of every touched column
    double sdelta[LAP_S];
    double red[MK_THREADS / 64];
    unsigned lock[MK_MAXN];
    unsigned short cj[MK_MAXN * LAP_K];  // candidate columns
    unsigned short scol[LAP_S * LAP_TS];
    short spred[LAP_S * LAP_TS];         // row that reached the column ...
    unsigned char spk[LAP_S * LAP_TS];   // ... through its k-th candidate
    unsigned char sscan[LAP_S * LAP_TS];
    short rowOfCol[MK_MAXN], colOfRow[MK_MAXN];
    unsigned short flist[MK_MAXN];
    unsigned char matchK[MK_MAXN];
    short snt[LAP_S], send[LAP_S];
    int wave_tot[MK_THREADS / 64];
    int flag[8];
};
static_assert(sizeof(LapShared) <= MOT_LDS_LIMIT, "lap_solve_kernel LDS");

__global__ void __launch_bounds__(MK_THREADS) lap_solve_kernel(AssocArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lap_raw[];
    LapShared& S = *reinterpret_cast<LapShared*>(lap_raw);
    int nR, nC; bool rowsTrk; resolve_dims(a, nR, nC, rowsTrk);
    const LapWs& L = a.ws.lap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long t_begin = wall_clock64();
    if (nR <= 0 || nC <= 0 || nR > nC) { if (tid == 0) L.hdr[LAP_H_SOLVE] = 5; return; }
    if (L.hdr[LAP_H_BAD]) { if (tid == 0) L.hdr[LAP_H_SOLVE] = 5; return; }
    for (int i = tid; i < nR * LAP_K; i += MK_THREADS) { S.cc[i] = L.ccost[i]; S.cj[i] = L.ccol[i]; }
    S.v[tid] = 0.0; S.rowOfCol[tid] = -1; S.colOfRow[tid] = -1; S.lock[tid] = 0xFFFFFFFFu;
    if (tid < 8) S.flag[tid] = 0;
    __syncthreads();
    // greedy start: every row asks for its cheapest column, the lowest row wins
    if (tid < nR) atomicMin(&S.lock[S.cj[tid * LAP_K]], (unsigned)tid);
    __syncthreads();
    if (tid < nR) { const int j = S.cj[tid * LAP_K]; if (S.lock[j] == (unsigned)tid) { S.rowOfCol[j] = (short)tid; S.colOfRow[tid] = (short)j; S.matchK[tid] = 0; } }
    __syncthreads();
    int rounds = 0, free0 = -1, searches = 0, commits = 0;
    for (;;) {
        // free rows, ascending
        const bool isfree = tid < nR && S.colOfRow[tid] < 0;
        const u64 bal = __ballot(isfree);
        if (lane == 0) S.wave_tot[wave] = __popcll(bal);
        S.lock[tid] = 0xFFFFFFFFu;
        __syncthreads();
        int off = 0, nf = 0;
        for (int w = 0; w < MK_THREADS / 64; w++) { const int t = S.wave_tot[w]; if (w < wave) off += t; nf += t; }
        if (isfree) S.flist[off + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)tid;
        if (free0 < 0) free0 = nf;
        if (nf == 0) break;
        if (++rounds > 4 * MK_MAXN) { if (tid == 0) S.flag[0] = 1; __syncthreads(); break; }
        __syncthreads();
        const int ns = min(nf, LAP_S);
        searches += ns;
        // searcher q runs on lane q / 16 of wave q % 16: eight lanes per wavefront, sixteen wavefronts in flight
        const int q = lane * (MK_THREADS / 64) + wave;
        const bool searcher = lane < LAP_S / (MK_THREADS / 64) && q < ns;
        double* sd = S.sd + q * LAP_TS; unsigned short* scol = S.scol + q * LAP_TS; short* spred = S.spred + q * LAP_TS;
        unsigned char* spk = S.spk + q * LAP_TS; unsigned char* sscan = S.sscan + q * LAP_TS;
        int nt = 0, jend = -1; double Delta = 0.0; bool ok = false;
        if (searcher) {
            const int s0 = S.flist[q];
            double us = DBL_MAX;
            for (int k = 0; k < LAP_K; k++) { const int j = S.cj[s0 * LAP_K + k]; if (j == 0xFFFF) break; const double x = S.cc[s0 * LAP_K + k] - S.v[j]; if (x < us) us = x; }
            for (int k = 0; k < LAP_K; k++) {
                const int j = S.cj[s0 * LAP_K + k]; if (j == 0xFFFF) break;
                scol[nt] = (unsigned short)j; sd[nt] = (S.cc[s0 * LAP_K + k] - S.v[j]) - us; spred[nt] = (short)s0; spk[nt] = (unsigned char)k; sscan[nt] = 0; nt++;
            }
            bool fail = false;
            for (;;) {
                int b = -1; double best = DBL_MAX;
                for (int t = 0; t < nt; t++) if (!sscan[t] && sd[t] < best) { best = sd[t]; b = t; }
                if (b < 0) { fail = true; break; }                    // no augmenting path inside the candidate graph
                Delta = best;
                const int j = scol[b];
                const int i = S.rowOfCol[j];
                if (i < 0) { jend = b; break; }
                sscan[b] = 1;
                const double ui = S.cc[i * LAP_K + S.matchK[i]] - S.v[j];
                for (int k = 0; k < LAP_K; k++) {
                    const int j2 = S.cj[i * LAP_K + k];
                    if (j2 == 0xFFFF) break;
                    if (j2 == j) continue;
                    const double nd = best + ((S.cc[i * LAP_K + k] - S.v[j2]) - ui);
                    int t = 0; while (t < nt && scol[t] != j2) t++;
                    if (t < nt) { if (!sscan[t] && nd < sd[t]) { sd[t] = nd; spred[t] = (short)i; spk[t] = (unsigned char)k; } }
                    else if (nt == LAP_TS) { fail = true; break; }
                    else { scol[t] = (unsigned short)j2; sd[t] = nd; spred[t] = (short)i; spk[t] = (unsigned char)k; sscan[t] = 0; nt++; }
                }
                if (fail) break;
            }
            ok = !fail;
            if (fail) S.flag[0] = 1;
            else for (int t = 0; t < nt; t++) if (sscan[t] || t == jend) atomicMin(&S.lock[scol[t]], (unsigned)q);
        }
        __syncthreads();
        if (S.flag[0]) break;
        if (searcher && ok) {
            bool mine = true;
            for (int t = 0; t < nt; t++) if ((sscan[t] || t == jend) && S.lock[scol[t]] != (unsigned)q) mine = false;
            if (mine) {
                const int s0 = S.flist[q];
                commits++;
                for (int t = 0; t < nt; t++) if (sscan[t]) S.v[scol[t]] -= (Delta - sd[t]);
                int t = jend;
                for (int guard = 0; guard <= LAP_TS; guard++) {
                    const int j = scol[t], i = spred[t];
                    const int pj = S.colOfRow[i];
                    S.colOfRow[i] = (short)j; S.matchK[i] = spk[t]; S.rowOfCol[j] = (short)i;
                    if (i == s0) break;
                    t = 0; while (t < nt && scol[t] != pj) t++;
                    if (t >= nt) { S.flag[0] = 2; break; }           // cannot happen: a tree row's column is a scanned column
                }
            }
        }
        __syncthreads();
        if (S.flag[0]) break;
    }
    __syncthreads();
    const int status = S.flag[0] ? 1 : 0;
    // duals, Gamma = sum_i (c[i][M(i)] - rowmin_i), margins
    double g = 0.0;
    if (!status && tid < nR) {
        const int j = S.colOfRow[tid];
        const double cm = S.cc[tid * LAP_K + S.matchK[tid]];
        L.u[tid] = cm - S.v[j];
        g = cm - S.cc[tid * LAP_K];
    }
#pragma unroll
    for (int off2 = 32; off2 > 0; off2 >>= 1) g += __shfl_xor(g, off2);
    if (lane == 0) S.red[wave] = g;
    L.v[tid] = S.v[tid]; L.colOfRow[tid] = S.colOfRow[tid]; L.rowOfCol[tid] = S.rowOfCol[tid];
    // commits per thread are only statistics: sum them
    int cs = commits;
#pragma unroll
    for (int off2 = 32; off2 > 0; off2 >>= 1) cs += __shfl_xor(cs, off2);
    __syncthreads();
    if (lane == 0) S.wave_tot[wave] = cs;
    __syncthreads();
    if (tid == 0) {
        double gamma = 0.0; int ctot = 0;
        for (int w = 0; w < MK_THREADS / 64; w++) { gamma += S.red[w]; ctot += S.wave_tot[w]; }
        const double cmax = dunkey(*L.cmaxkey);
        const double mag = cmax + gamma;
        const double n3 = (double)nC * (double)nC * (double)nC;
        L.dhdr[0] = fmax(1e-9, 1e-15 * n3) * mag;                      // eps: see the header of lap_model.c (CPU model, test infrastructure)
        L.dhdr[1] = 1e-12 * mag;                                       // tol
        L.dhdr[2] = gamma; L.dhdr[3] = cmax;
        L.hdr[LAP_H_SOLVE] = status;
        L.hdr[LAP_H_LAST + 1] = rounds; L.hdr[LAP_H_LAST + 2] = free0; L.hdr[LAP_H_LAST + 3] = searches; L.hdr[LAP_H_LAST + 4] = ctot;
        L.hdr[LAP_H_LAST + 7] = (int)(wall_clock64() - t_begin);
    }
}

// ---- stage 3 ----------------------------------------------------------------------------------------------------
// assoc_sub_kernel (working matrix + zero bitmaps for the Munkres kernel) + the dense dual check + near-tight edges
__global__ void __launch_bounds__(256) lap_verify_kernel(AssocArgs a)
{
    __shared__ unsigned int zr_lo[64], zr_hi[64];
    int nR, nC; bool rowsTrk; resolve_dims(a, nR, nC, rowsTrk);
    const LapWs& L = a.ws.lap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 64 + lane, c0 = blockIdx.y * 64;
    if (blockIdx.x * 64 >= nR || c0 >= nC) return;
    const int wordsR = (nR + 63) >> 6, wordsC = (nC + 63) >> 6;
    if (threadIdx.x < 64) { zr_lo[threadIdx.x] = 0; zr_hi[threadIdx.x] = 0; }
    __syncthreads();
    const bool solved = L.hdr[LAP_H_SOLVE] == 0;
    const double eps = L.dhdr[0], tol = L.dhdr[1];
    const double rmin = r < nR ? dunkey(a.linemin[r]) : 0.0;
    const double ur = (solved && r < nR) ? L.u[r] : 0.0;
    const int mr = (solved && r < nR) ? (int)L.colOfRow[r] : -1;
    bbox_t rb = {};
    if (!a.user && r < nR) rb = rowsTrk ? a.trk[r] : a.det[r];
    bool viol = false;
    for (int cc = wave; cc < 64; cc += 4) {
        const int c = c0 + cc;
        if (c >= nC) break;
        bool z = false;
        if (r < nR) {
            double cst;
            if (a.user) cst = a.user[(size_t)r + (size_t)nR * c];
            else cst = rowsTrk ? pair_cost(rb, a.det[c]) : pair_cost(a.trk[c], rb);
            const double d = cst - rmin;
            a.ws.dist[(size_t)r + (size_t)nR * c] = d;
            z = fabs(d) < DBL_EPSILON;
            if (solved && c != mr) {
                const double red = (cst - L.v[c]) - ur;
                if (!(red >= -tol)) viol = true;
                else if (red < eps) {
                    const int e = atomicAdd(&L.hdr[LAP_H_NEDGES], 1);
                    const int owner = L.rowOfCol[c];
                    if (e < LAP_EDGES) L.edges[e] = ((unsigned)r << 16) | (unsigned)(owner >= 0 ? owner : nR);
                }
            }
        }
        const u64 bal = __ballot(z);
        if (lane == 0) a.ws.zc[(size_t)c * wordsR + blockIdx.x] = bal;
        if (z) { if (cc < 32) atomicOr(&zr_lo[lane], 1u << cc); else atomicOr(&zr_hi[lane], 1u << (cc - 32)); }
    }
    // per column (first row tile only): prices <= 0, exactly 0 on free columns; "a free column could take this column's row"
    if (solved && blockIdx.x == 0 && threadIdx.x < 64 && c0 + (int)threadIdx.x < nC) {
        const int c = c0 + threadIdx.x;
        const double vc = L.v[c];
        const int owner = L.rowOfCol[c];
        if (!(vc <= 0.0) || (owner < 0 && vc != 0.0)) viol = true;
        else if (owner >= 0 && nC > nR && -vc < eps) {
            const int e = atomicAdd(&L.hdr[LAP_H_NEDGES], 1);
            if (e < LAP_EDGES) L.edges[e] = ((unsigned)nR << 16) | (unsigned)owner;
        }
    }
    if (__syncthreads_or(viol) && threadIdx.x == 0) atomicOr(&L.hdr[LAP_H_VIOL], 1);
    if (threadIdx.x < 64 && r < nR) a.ws.zr[(size_t)r * wordsC + blockIdx.y] = ((u64)zr_hi[threadIdx.x] << 32) | zr_lo[threadIdx.x];
}

} // namespace

hipError_t launch_lap_front(const AssocArgs& a, int gR, int gC, hipStream_t s)
{
    static int attr_dev = -1;                                          // per-device function attribute
    int dev = 0; hipError_t e = hipGetDevice(&dev); if (e != hipSuccess) return e;
    if (attr_dev != dev) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(lap_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LapShared));
        if (e != hipSuccess) return e;
        attr_dev = dev;
    }
    hipLaunchKernelGGL(lap_rowscan_kernel, dim3((gR * 64 + 3) / 4), dim3(256), 0, s, a);
    hipLaunchKernelGGL(lap_solve_kernel, dim3(1), dim3(MK_THREADS), sizeof(LapShared), s, a);
    hipLaunchKernelGGL(lap_verify_kernel, dim3(gR, gC), dim3(256), 0, s, a);
    return hipGetLastError();
}
