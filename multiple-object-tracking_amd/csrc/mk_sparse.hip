// mk_sparse.hip -- SPARSE order-exact emulation of the reference's Munkres (trackers/hungarian/hungarian.cpp:29-368)
// and its after-the-fact validity check.
//
// The dense emulation (assoc_kernels.hip) moves an n x n float64 matrix through every step 5.  Here every row keeps only
// its LAP_K smallest entries (the candidate lists lap_rowscan_kernel produced) and the reference's state machine runs on
// them, in LDS, with the reference's scan orders (step 1 :93-101, step 3 :249-275: columns ascending, rows ascending,
// one hit per column and sweep), element arithmetic (:355-364: +h on covered rows first, then -h on uncovered columns)
// and zero test fabs(x) < DBL_EPSILON.  An entry OUTSIDE the lists has, at any time,
//        value = c[i][j] - rowmin_i + A_i(t) - S_j(t)  >=  c[i][j] - rowmin_i - S_j(final)
// (A_i / S_j: what step 5 has added to row i / subtracted from column j so far; both only grow).  mk_postcheck_kernel
// tests that bound against a margin for every such entry, with the S_j this run ended with: if it holds, no outside
// entry was ever zero or the minimum of a step 5 (its value after that step is still positive, so it was above h),
// so the dense run and this run are step by step the same run and the assignment is the reference's, ties included.
// If it fails (dense / adversarial matrices, false-positive detections whose partner is far away) the final kernel
// runs the dense emulation as before.  CPU model: mk_sparse_model.c (test infrastructure).
//
// One 1024-thread workgroup.  Thread = row for step 5 and the set-up; steps 3 / 4 / 2a / 2b run in wavefront 0:
//   * zeros of a row: bit mask over its candidates; zeros of a column: the transposed candidate lists (rows ascending);
//   * per column two slot masks over its transposed list: tzero (slots holding a zero) and tlive (... in an UNCOVERED row),
//     and hz = {c : tlive[c] != 0} as a bit mask: "the next uncovered column with an uncovered zero" is one masked
//     find-first-bit, "its first uncovered zero row" one LDS round trip, and covering a row one LDS atomic per zero of
//     that row; after an augmentation (all rows uncovered, :324-330) tlive = tzero.
#include "mk_sparse_body.h"
#include "mot_env.h"

using namespace assoc;

namespace {

// post_fused (box costs): the after-the-fact check runs in this workgroup as a disc query per row over a grid of the column boxes
// (same bound per examined entry as mk_postcheck_kernel; an entry that is not examined satisfies it a fortiori), and -- device loop -- an
// accepted run is committed here (lifecycle step), so a tie frame needs no further kernel of the chain.  The body lives in
// mk_sparse_body.h: the solver's launch runs it as its second workgroup (lap_kernels.hip); this kernel is the stand-alone form (caller
// matrices, streams with the dense solver armed).
template <bool TIMING>
__global__ void __launch_bounds__(MK_THREADS) mk_sparse_kernel(AssocArgs a, int mk_batch, int post_fused, LifeArgs life)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_raw[];
    mk_sparse_run<false, TIMING>(a, mk_batch, post_fused, life, sp_raw);
}

// every entry outside the candidate lists against the final S_j (see the header); grid of 64 x 64 tiles
__global__ void __launch_bounds__(256) mk_postcheck_kernel(AssocArgs a)
{
    __shared__ bbox_t colb[64]; __shared__ double colS[64]; __shared__ int big;
    int nR, nC; bool rowsTrk; resolve_dims(a, nR, nC, rowsTrk);
    const LapWs& L = a.ws.lap;
    if (L.hdr[LAP_H_MODE] != 1) return;                                // certified, or dense emulation anyway
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 64 + lane, c0 = blockIdx.y * 64;
    if (blockIdx.x * 64 >= nR || c0 >= nC) return;
    if (threadIdx.x == 0) big = 0;
    __syncthreads();
    if (threadIdx.x < 64 && c0 + (int)threadIdx.x < nC) {
        const int c = c0 + threadIdx.x;
        if (!a.user) { const bbox_t cb = rowsTrk ? a.det[c] : a.trk[c]; colb[threadIdx.x] = cb; if (!box_small(cb)) big = 1; }
        colS[threadIdx.x] = L.spS[c];
    }
    bool viol = false;
    unsigned short lj = 0xFFFF; double rmin = 0.0, lc = 0.0; bbox_t rb = {};
    if (r < nR) {
        lj = L.ccol[(size_t)r * SPK + SPK - 1];
        rmin = L.ccost[(size_t)r * SPK]; lc = L.ccost[(size_t)r * SPK + SPK - 1];
        if (!a.user) rb = rowsTrk ? a.trk[r] : a.det[r];
    }
    const double margin = 1e-9 * (1.0 + L.dhdr[3]);                    // 1e-9 * (1 + largest cost): far above the reference's accumulated rounding
    __syncthreads();
    const bool est_ok = !a.user && !big && box_small(rb);              // float estimate of the cost usable (assoc_common.h)
    if (r < nR && lj != 0xFFFF) {                                      // (a row with fewer than LAP_K columns has every entry in its list)
#pragma unroll 4
        for (int cc = wave; cc < 64; cc += 4) {
            const int c = c0 + cc;
            if (c >= nC) break;
            double cst;
            if (a.user) cst = a.user[(size_t)r + (size_t)nR * c];
            else {
                int d2; bool pen; if (rowsTrk) pair_d2(rb, colb[cc], d2, pen); else pair_d2(colb[cc], rb, d2, pen);
                if (est_ok) {                                          // clearly outside the list AND clearly above the bound, or clearly inside the list: done
                    const double e = (double)cost_of_d2_f32(d2, pen);
                    if (e < lc - 2.0 * PAIR_COST_F32_ERR) continue;
                    if (e > lc + 2.0 * PAIR_COST_F32_ERR && e - rmin - colS[cc] > margin + 2.0 * PAIR_COST_F32_ERR) continue;
                }
                cst = rowsTrk ? pair_cost(rb, colb[cc]) : pair_cost(colb[cc], rb);
            }
            const bool outside = cst > lc || (cst == lc && c > (int)lj);
            if (outside && !(cst - rmin - colS[cc] > margin)) viol = true;
        }
    }
    if (__syncthreads_or(viol) && threadIdx.x == 0) atomicOr(&L.hdr[LAP_H_SPVIOL], 1);
}

} // namespace

hipError_t launch_mk_sparse(const AssocArgs& a, int gR, int gC, hipStream_t s, const LifeArgs& life)
{
    hipError_t e = mot_impl::func_lds_once(reinterpret_cast<const void*>(mk_sparse_kernel<false>), (int)sizeof(SpShared)); if (e != hipSuccess) return e;
    e = mot_impl::func_lds_once(reinterpret_cast<const void*>(mk_sparse_kernel<true>), (int)sizeof(SpShared)); if (e != hipSuccess) return e;
    const int batch = mot_impl::env().mk_batch;                        // MOT_MK_BATCH / MOT_MK_LAZY=0 (the reference's full reset after every augmentation) / MOT_MK_TIMING
    // box costs: the after-the-fact check (and the lifecycle step) run inside the emulation's workgroup; caller matrices keep the dense pass
    const int post_fused = !a.user ? 1 : 0;
    mot_impl::lds_poison(s);                                           // (debug) MOT_LDS_POISON
    if (batch & SP_TIMING) hipLaunchKernelGGL(mk_sparse_kernel<true>, dim3(1), dim3(MK_THREADS), sizeof(SpShared), s, a, batch, post_fused, life);
    else hipLaunchKernelGGL(mk_sparse_kernel<false>, dim3(1), dim3(MK_THREADS), sizeof(SpShared), s, a, batch, post_fused, life);
    if (!post_fused) hipLaunchKernelGGL(mk_postcheck_kernel, dim3(gR, gC), dim3(256), 0, s, a);
    return hipGetLastError();
}
