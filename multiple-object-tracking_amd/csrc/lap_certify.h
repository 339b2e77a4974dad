// lap_certify.h -- last stage of the assignment fast path (lap_kernels.hip), executed by the workgroup of
// mk_sparse_kernel (mk_sparse.hip) before it would start the order-exact emulation.
//
// Input: the solver's matching M and duals, already checked entry by entry by lap_verify_kernel (LAP_H_VIOL), and the
// list of near-tight edges  i -> i'  ("row i could take the column of row i' at a reduced cost below eps"; node nR
// stands for all free columns).  Any other assignment M' differs from M along alternating cycles of that digraph's
// complete version, and cost(M') - cost(M) = sum of the reduced costs on them >= (eps - n * tol) unless every edge of
// a cycle is near-tight.  So: no directed cycle among the near-tight edges  =>  M is the unique optimum by more than
// anything float64 rounding moves inside the reference (margin: lap_model.c (CPU model, test infrastructure))  =>  hungarian.cpp:29-368 returns M.
#pragma once
#include "assoc_common.h"

namespace assoc {

// scratch: >= LAP_EDGES unsigned + 2 * (MK_MAXN + 64) bytes of LDS.  Returns 0 (workgroup-uniform) iff certified, else the
// reason: 1 solver gave up / not applicable, 2 infeasible dual, 3 too many near-tight edges, 4 tied optima.
// in_lds: the caller (the fused dual check of lap_solve_kernel) has already put the edge list into scratch[0..ne) and passes ne / viol / solve
// itself; otherwise they come from the header and the list from L.edges (written by lap_verify_kernel in an earlier launch).
__device__ inline int lap_certify(const LapWs& L, int nR, int nC, unsigned* scratch, int* flag2, bool in_lds = false, int ne_in = 0, int viol_in = 0, int solve_in = 0)
{
    const int tid = threadIdx.x;
    unsigned* ed = scratch;
    unsigned char* alive = reinterpret_cast<unsigned char*>(scratch + LAP_EDGES);
    unsigned char* hasout = alive + MK_MAXN + 64;
    const int solve = in_lds ? solve_in : L.hdr[LAP_H_SOLVE], viol = in_lds ? viol_in : L.hdr[LAP_H_VIOL], ne = in_lds ? ne_in : L.hdr[LAP_H_NEDGES], bad = L.hdr[LAP_H_BAD];
    int reason = 0;                                                     // 0 certified, 1 solver gave up / n.a., 2 infeasible dual, 3 too many near-tight edges, 4 tie
    if (solve != 0 || bad) reason = 1; else if (viol) reason = 2; else if (ne > LAP_EDGES) reason = 3;
    int ncyc = 0;
    if (!reason && ne <= 16 * 64) {
        // the usual case, a few hundred edges: wavefront 0 alone, no workgroup barriers (LDS operations of one wavefront execute in order).
        // Only edge sources can survive a peel round, so the rounds walk the edge list, never the node range: alive[] starts as "is the
        // source of some edge", hasout[] holds the number of the last round in which the node still had an edge into the alive set.
        if (tid < 64) {
            const int lane = tid;
            unsigned* aw = reinterpret_cast<unsigned*>(alive); unsigned* hw = reinterpret_cast<unsigned*>(hasout);
            for (int i = lane; i < (MK_MAXN + 64) / 4; i += 64) { aw[i] = 0; hw[i] = 0; }
            for (int e = lane; e < ne; e += 64) { const unsigned x = in_lds ? ed[e] : L.edges[e]; ed[e] = x; alive[x >> 16] = 1; }
            for (int it = 1;; it++) {
                for (int e = lane; e < ne; e += 64) { const unsigned x = ed[e]; if (alive[x >> 16] && alive[x & 0xFFFF]) hasout[x >> 16] = (unsigned char)it; }
                bool ch = false;
                for (int e = lane; e < ne; e += 64) { const unsigned x = ed[e]; if (alive[x >> 16] && hasout[x >> 16] != (unsigned char)it) { alive[x >> 16] = 0; ch = true; } }
                if (!__ballot(ch)) break;
                if (it == 250) { for (int i = lane; i < (MK_MAXN + 64) / 4; i += 64) hw[i] = 0; it = 0; }   // (round numbers are bytes: start over)
            }
            int mine = 0;
            for (int e = lane; e < ne; e += 64) if (alive[ed[e] >> 16]) mine = 1;
            if (lane == 0) flag2[1] = 0;
            if (__ballot(mine != 0) && lane == 0) flag2[1] = 1;
        }
        __syncthreads();
        ncyc = flag2[1] ? 2 : 0;                                        // (a count is only kept by the block path below)
        if (ncyc) reason = 4;
    } else if (!reason) {
        if (!in_lds) for (int e = tid; e < ne; e += MK_THREADS) ed[e] = L.edges[e];
        for (int i = tid; i <= nR; i += MK_THREADS) alive[i] = 1;
        __syncthreads();
        // peel nodes without an edge into the still-alive set; what survives lies on or leads into a cycle
        for (int it = 0; it <= nR + 1; it++) {
            for (int i = tid; i <= nR; i += MK_THREADS) hasout[i] = 0;
            if (tid == 0) flag2[0] = 0;
            __syncthreads();
            for (int e = tid; e < ne; e += MK_THREADS) { const unsigned x = ed[e]; const int s = x >> 16, d = x & 0xFFFF; if (alive[s] && alive[d]) hasout[s] = 1; }
            __syncthreads();
            for (int i = tid; i <= nR; i += MK_THREADS) if (alive[i] && !hasout[i]) { alive[i] = 0; flag2[0] = 1; }
            __syncthreads();
            const int changed = flag2[0];
            __syncthreads();
            if (!changed) break;
        }
        int mine = 0;
        for (int i = tid; i <= nR; i += MK_THREADS) mine += alive[i];
        ncyc = __syncthreads_count(mine != 0);                          // nodes = threads here (nR + 1 <= 1025: thread 0 may hold two)
        if (ncyc) reason = 4;
    }
    if (tid == 0) {
        L.hdr[LAP_H_LAST + 0] = reason; L.hdr[LAP_H_LAST + 5] = ne; L.hdr[LAP_H_LAST + 6] = ncyc;
        L.hdr[LAP_H_CUM + reason] += 1;                                  // [32] certified, [33..36] not certified, by reason
    }
    return reason;
}

} // namespace assoc
