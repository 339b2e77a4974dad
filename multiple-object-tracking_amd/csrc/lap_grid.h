// lap_grid.h -- uniform grid over the column boxes of an association problem (box costs, td.cpp:386-457), built in LDS by one
// 1024-thread workgroup.  The cost of (row, column) is the centroid distance / 1280 (+ 1.0 across classes), so "every column whose
// cost to this row can be below x" is a disc query: the passes that used to visit all n x n entries (dual check of the LAP solver,
// after-the-fact check of the sparse Munkres emulation) visit a handful of cells per row instead.
#pragma once
#include "assoc_common.h"

namespace assoc {

#define LAPG_CELL_SHIFT 5                  /* 32 x 32 px cells */
#define LAPG_GW ((MOT_FRAME_W >> LAPG_CELL_SHIFT) + 1)
#define LAPG_GH ((MOT_FRAME_H >> LAPG_CELL_SHIFT) + 1)
#define LAPG_NCELL (LAPG_GW * LAPG_GH)
static_assert(LAPG_NCELL <= MK_MAXN, "one thread per grid cell");

struct ColGrid {
    int cellCnt[MK_MAXN];                // per cell: count, then fill cursor
    unsigned short cellStart[MK_MAXN + 1];
    unsigned short sorted[MK_MAXN];      // column indices, cell by cell (the cells of a grid row are contiguous)
    short ccx[MK_MAXN], ccy[MK_MAXN];    // column centroids ((l + r) >> 1, (t + b) >> 1, td.cpp:407-410)
    int ctype[MK_MAXN];
};

// all 1024 threads; wave_tot: 16 ints of LDS.  Returns true (workgroup-uniform) if some box of either side lies outside the
// +-1400 px range the short / int arithmetic assumes (box_small): the caller then scans every column instead of querying.
__device__ inline bool grid_build(ColGrid& G, const AssocArgs& a, int nR, int nC, bool rowsTrk, int* wave_tot)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    G.cellCnt[tid] = 0;
    __syncthreads();
    bool small = true; int mycell = 0;
    if (tid < nC) {
        const bbox_t cb = rowsTrk ? a.det[tid] : a.trk[tid];
        small = box_small(cb);
        const int cx = (cb.l + cb.r) >> 1, cy = (cb.t + cb.b) >> 1;
        G.ccx[tid] = (short)cx; G.ccy[tid] = (short)cy; G.ctype[tid] = cb.type;
        const int gx = min(max(cx >> LAPG_CELL_SHIFT, 0), LAPG_GW - 1), gy = min(max(cy >> LAPG_CELL_SHIFT, 0), LAPG_GH - 1);
        mycell = gy * LAPG_GW + gx;
        atomicAdd(&G.cellCnt[mycell], 1);
    }
    if (tid < nR) small &= box_small(rowsTrk ? a.trk[tid] : a.det[tid]);
    const bool big = __syncthreads_or(!small) != 0;
    {   // exclusive prefix sum of the cell counts (one thread per cell)
        const int cnt = G.cellCnt[tid];
        int pre = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(pre, off); if (lane >= off) pre += t; }
        if (lane == 63) wave_tot[wave] = pre;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; w++) base += wave_tot[w];
        G.cellStart[tid] = (unsigned short)(base + pre - cnt);
        if (tid == MK_THREADS - 1) G.cellStart[MK_MAXN] = (unsigned short)(base + pre);
        G.cellCnt[tid] = 0;                                            // now the fill cursor
    }
    __syncthreads();
    if (tid < nC) G.sorted[G.cellStart[mycell] + atomicAdd(&G.cellCnt[mycell], 1)] = (unsigned short)tid;
    __syncthreads();
    return big;
}

// visits every column j of class `type` whose centroid lies within Ri px (Euclidean) of (cx, cy): f(j, d2)
template <typename F>
__device__ __forceinline__ void grid_query(const ColGrid& G, int cx, int cy, int type, int Ri, F&& f)
{
    const int R2 = Ri * Ri;
    const int gx0 = min(max((cx - Ri) >> LAPG_CELL_SHIFT, 0), LAPG_GW - 1), gx1 = min(max((cx + Ri) >> LAPG_CELL_SHIFT, 0), LAPG_GW - 1);
    const int gy0 = min(max((cy - Ri) >> LAPG_CELL_SHIFT, 0), LAPG_GH - 1), gy1 = min(max((cy + Ri) >> LAPG_CELL_SHIFT, 0), LAPG_GH - 1);
    for (int gy = gy0; gy <= gy1; gy++) {
        const int q0 = G.cellStart[gy * LAPG_GW + gx0], q1 = G.cellStart[gy * LAPG_GW + gx1 + 1];
        for (int q = q0; q < q1; q++) {
            const int j = G.sorted[q];
            if (G.ctype[j] != type) continue;
            const int dx = cx - (int)G.ccx[j], dy = cy - (int)G.ccy[j];
            const int d2 = dx * dx + dy * dy;
            if (d2 <= R2) f(j, d2);
        }
    }
}

} // namespace assoc
