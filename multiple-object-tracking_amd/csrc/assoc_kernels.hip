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
//   * step 5 touches only covered rows and uncovered columns; above 512 lines it runs on 16 helper workgroups that
//     own 64 columns each and exchange cover masks, partial minima and new zero bits with the controller workgroup
//     as self-tagged 8-byte granules (mk_helper_loop, assoc_common.h);
//   * below 65 lines the Munkres workgroup computes costs, minima and bitmaps itself (mk_fused_cost).
#include "assoc_common.h"
#include "mot_env.h"
#include "dl_lifecycle.h"
#include <stdlib.h>

using namespace assoc;

namespace {

// pass 1: per-line minimum (rows if nR <= nC, hungarian.cpp:69-81; else columns, :107-119)
__global__ void __launch_bounds__(256) assoc_min_kernel(AssocArgs a)
{
    int nR, nC; bool rowsTrk; resolve_dims(a, nR, nC, rowsTrk);
    const int r = blockIdx.x * 64 + (threadIdx.x & 63);
    const int c0 = blockIdx.y * 64, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && a.ws.ctl) { for (int i = 0; i < 2 * MK_HELPERS; i++) a.ws.ctl[CTL_PARTIAL + i * MK_PARTIAL_STRIDE] = MK_HSENT; for (int i = 0; i < 64; i++) a.ws.ctl[CTL_COV + i] = 0; }
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

// pass 2: working matrix d = cost - linemin, zero bitmaps in both orientations.  lazy: behind the fast path (lap_kernels.hip,
// mk_sparse.hip) this is only needed when the dense emulation has to run -- every workgroup checks the verdict and leaves
__global__ void __launch_bounds__(256) assoc_sub_kernel(AssocArgs a, int lazy)
{
    __shared__ unsigned int zr_lo[64], zr_hi[64];
    if (lazy) {
        // stream emulation (round 6): the emulation's kernel may be publishing MODE / SPVIOL while this launch runs (MODE = 1 is stored before the
        // post-check's SPVIOL): only a COMMITTED frame (DONE, which never goes back) is skipped; a launch that straddles the commit prepares a
        // matrix nobody reads
        if (a.stream_emu) { if (a.ws.lap.hdr[LAP_H_DONE]) return; }
        else { const int mode = a.ws.lap.hdr[LAP_H_MODE]; if (mode == 0 || (mode == 1 && !a.ws.lap.hdr[LAP_H_SPVIOL])) return; }
    }
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
#define MK_HELP_MIN 512
#define MK_LAP_MIN_LINES 96          /* default smallest problem for the fast path (lap_kernels.hip) */
#define MK_XCDS 8                /* MI355X: 8 XCDs, workgroup i of a launch goes to XCD i % 8 */
#define MK_SPIN_LIMIT 4000000   /* bounded spins: a lost partner ends the wait after seconds instead of hanging the GPU */
struct MkShared {
    u64 bm[MK_MAXN * MK_MAXW];      // zero bitmap, [line][16 words]; row-major during init, column-major afterwards
    short starColOfRow[MK_MAXN];
    short starRowOfCol[MK_MAXN];
    short primeColOfRow[MK_MAXN];
    unsigned short list[MK_MAXN];   // uncovered columns, ascending
    unsigned short clist[MK_MAXN];  // init: contested lines, ascending
    unsigned short crosscnt[MK_MAXN];
    u64 covR[MK_MAXW], covC[MK_MAXW], hz[MK_MAXW];
    unsigned char hcols[64];        // helper workgroups: my covered columns (bit positions of my column word), ascending
    u64 hzl[MK_MAXW];               // after a step 5: uncovered columns that hold a zero in an UNCOVERED row (exact)
    unsigned int taken32[2 * MK_MAXW], cont32[2 * MK_MAXW];
    double red[MK_THREADS / 64];
    u64 hbits;
    int flag[8];
};


// ---- helper workgroups: own 64 consecutive columns each; execute the bulk of step 5 on them -----------------
// Column ownership is fixed, so a column's elements are only ever touched by one CU (no cross-CU visibility issue for
// the matrix itself); only the small control block crosses CUs.  One iteration = one step 5:
//   poll the cover-mask granules -> minimum over (uncovered rows) x (my uncovered columns) -> PARTIAL[g]
//   poll h -> apply (:355-364) to my uncovered columns and +h to (covered rows) x (my covered columns); every value
//             needed is already in registers (loaded before the wait) -> ballots of the zero test as tagged granules
// All spins are executed by wave 0 as a whole on a wave-uniform (scalar) condition and are bounded: a spin loop
// confined to one LANE is a divergent loop, and the structuriser may run the other lanes of that wave (and with them
// the workgroup barrier behind the spin) ahead of it -- seen on gfx950: the barrier released before lane 0 had polled.
// force_cov (test hook, MOT_MUNKRES_HELPERS=2): every summary reports a zero, so the controller takes the per-row granule path
__device__ void mk_helper_loop(const AssocArgs& a, MkShared& S, int nR, int nC, int force_cov)
{
    u64* ctl = a.ws.ctl;
    double* __restrict__ d = a.ws.dist;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = (int)(blockIdx.x / MK_XCDS) - 1;   // helpers = blocks 8, 16, ..., 128
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int wordsR = (nR + 63) >> 6;
    const int r = tid;
    const size_t rclamp = (size_t)min(r, nR - 1);
    const int cbase = g * 64;
    const int nvalid = min(max(nC - cbase, 0), 64);
    const u64 validC = nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1);
    const size_t coff = (size_t)nR * min(cbase + lane, nC - 1);        // covered-rows part: lane = column
    unsigned step = 0;
    const unsigned epoch = (unsigned)ctl_ld(ctl + CTL_EPOCH);          // written by the previous launch's controller
    if (tid == 0) ctl_st(ctl + CTL_XCCTAB + g, (u64)xcc_id() | ((u64)(epoch + 1) << 32));   // where do I run?
    bool fast = false;                                                  // the controller's verdict, read after its first publish
    for (;;) {
        ++step;
        const unsigned tag = epoch + step;
        if (uwave == 0) {
            int spins = 0; u64 gv; int state;                           // 0 published, 1 exit, 2 timed out
            for (;;) {
                gv = ctl_ld(ctl + CTL_COV + lane);
                const unsigned tg = (unsigned)(gv >> 32);
                if (!__ballot(tg != tag)) { state = 0; break; }
                if (__ballot(tg == MK_TAG_EXIT && tag != MK_TAG_EXIT) == ~0ull) { state = 1; break; }
                if (++spins > MK_SPIN_LIMIT) { state = 2; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            const u64 lo = (u64)__shfl((unsigned)gv, (lane & 31) * 2) , hi = (u64)__shfl((unsigned)gv, (lane & 31) * 2 + 1);
            const u64 cv = lo | (hi << 32);                              // lanes 0..15: COVR words, 16..31: COVC words
            if (lane < MK_MAXW) S.covR[lane] = cv; else if (lane < 2 * MK_MAXW) S.covC[lane - MK_MAXW] = cv;
            const u64 um = ~readlane64(cv, MK_MAXW + g) & validC;        // my uncovered columns
            if ((um >> lane) & 1) S.list[__popcll(um & ((1ull << lane) - 1))] = (unsigned short)(cbase + lane);
            if (lane == 0) {
                S.flag[3] = state; S.flag[2] = __popcll(um); S.flag[6] = 0;
                if (step == 1 && state == 0) { const u64 md = ctl_ld(ctl + CTL_MODE); S.flag[5] = ((unsigned)(md >> 32) == epoch + 1) ? (int)(md & 1) : 0; }
            }
        }
        __syncthreads();
        if (S.flag[3]) return;                                          // the controller is done / went away
        if (step == 1) fast = S.flag[5] != 0;
        const int nmine = S.flag[2];
        const bool rowcov = (S.covR[wave] >> lane) & 1;
        const bool mine = r < nR && !rowcov;
        // work in the update phase: my uncovered columns, or covered rows crossing my covered columns
        int ncr = 0;
        for (int w = 0; w < wordsR; w++) ncr += __popcll(S.covR[w]);
        const bool part2 = ncr > 0 && (S.covC[g] & validC) != 0;
        // ---- phase A: minimum key over (uncovered rows) x (my uncovered columns); 8 loads in flight per thread ----
        double v[8];
        u64 best = ~0ull;
        if (nmine > 0) {
#pragma unroll
            for (int k = 0; k < 8; k++) if (k < nmine) v[k] = d[rclamp + (size_t)nR * (unsigned)__builtin_amdgcn_readfirstlane((int)S.list[k])];   // uniform guard
        }
        // wave 1 (wave 0 is the poller): the covered rows, ascending, for the update of my covered columns (read behind the barrier below)
        if (uwave == 1 && part2) wave_list_bits((lane < MK_MAXW) ? S.covR[lane] : 0, 0, S.clist, lane);   // (== ncr entries)
        if (nmine > 0) {
#pragma unroll
            for (int k = 0; k < 8; k++) if (mine && k < nmine) { const u64 kk = dkey(v[k]); if (kk < best) best = kk; }
            for (int q0 = 8; q0 < nmine; q0 += 8) {
                double w[8];
#pragma unroll
                for (int k = 0; k < 8; k++) w[k] = d[rclamp + (size_t)nR * (unsigned)__builtin_amdgcn_readfirstlane((int)S.list[min(q0 + k, nmine - 1)])];
#pragma unroll
                for (int k = 0; k < 8; k++) if (mine && q0 + k < nmine) { const u64 kk = dkey(w[k]); if (kk < best) best = kk; }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const u64 o = __shfl_xor(best, off); if (o < best) best = o; }
        if (lane == 0) S.red[wave] = __longlong_as_double((long long)best);
        __syncthreads();
        if (uwave == 0) {
            u64 b = (lane < MK_THREADS / 64) ? (u64)__double_as_longlong(S.red[lane]) : ~0ull;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) { const u64 o = __shfl_xor(b, off); if (o < b) b = o; }
            if (lane == 0) ctl_stx(ctl + CTL_PARTIAL + ((step & 1) * MK_HELPERS + g) * MK_PARTIAL_STRIDE, b < MK_KEY_NONE ? b : MK_KEY_NONE, fast);   // the word is its own flag
            if (nmine > 0 || part2) {
                // ---- h = minimum of the 16 partial minima: every helper with work reads them itself (no detour
                // through the controller) ----
                int spins = 0; u64 pk; bool lost = false;
                for (;;) {
                    pk = ctl_ld(ctl + CTL_PARTIAL + ((step & 1) * MK_HELPERS + (lane & (MK_HELPERS - 1))) * MK_PARTIAL_STRIDE);
                    if (!__ballot(pk == MK_HSENT)) break;
                    if (++spins > MK_SPIN_LIMIT) { lost = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) { const u64 o = __shfl_xor(pk, off); if (o < pk) pk = o; }
                if (lane == 0) { S.flag[3] = lost ? 2 : 0; S.hbits = (u64)__double_as_longlong(dunkey(pk)); }
            }
        }
        __syncthreads();
        if (nmine == 0 && !part2) continue;                            // nothing of mine changes in this step 5 (the controller expects no bits from me)
        if (S.flag[3]) return;
        const double h = __longlong_as_double((long long)S.hbits);
        const u64 tagw = (u64)tag << 32;                                 // every granule carries the tag of its step 5
        // ---- phase B (:355-364) on my uncovered columns; the ballots are staged in LDS (the helpers do not use the zero
        // bitmap) and each column's 32 granules leave as ONE 256-byte store ----
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (k < nmine) {                                            // uniform
                const int c = __builtin_amdgcn_readfirstlane((int)S.list[k]);
                const double x = rowcov ? (v[k] + h) - h : v[k] - h;
                if (r < nR) d[(size_t)r + (size_t)nR * c] = x;
                const u64 bal = __ballot(r < nR && fabs(x) < DBL_EPSILON);
                if (lane == 0) S.bm[k * MK_MAXW + wave] = bal;
            }
        }
        __syncthreads();
        if (uwave < min(nmine, 8) && lane < 2 * MK_MAXW) {
            const int c = S.list[uwave];
            const u64 bal = S.bm[uwave * MK_MAXW + (lane >> 1)];
            ctl_stx(ctl + CTL_BMOUT + (size_t)c * MK_MAXW * 2 + lane, ((lane & 1) ? bal >> 32 : bal & 0xFFFFFFFFull) | tagw, fast);
        }
        for (int q0 = 8; q0 < nmine; q0 += 8) {
            double w[8];
#pragma unroll
            for (int k = 0; k < 8; k++) w[k] = d[rclamp + (size_t)nR * (unsigned)__builtin_amdgcn_readfirstlane((int)S.list[min(q0 + k, nmine - 1)])];
            __syncthreads();                                            // the staging words of the previous batch have been read
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (q0 + k < nmine) {
                    const int c = __builtin_amdgcn_readfirstlane((int)S.list[q0 + k]);
                    const double x = rowcov ? (w[k] + h) - h : w[k] - h;
                    if (r < nR) d[(size_t)r + (size_t)nR * c] = x;
                    const u64 bal = __ballot(r < nR && fabs(x) < DBL_EPSILON);
                    if (lane == 0) S.bm[k * MK_MAXW + wave] = bal;
                }
            }
            __syncthreads();
            if (uwave < min(nmine - q0, 8) && lane < 2 * MK_MAXW) {
                const int c = S.list[q0 + uwave];
                const u64 bal = S.bm[uwave * MK_MAXW + (lane >> 1)];
                ctl_stx(ctl + CTL_BMOUT + (size_t)c * MK_MAXW * 2 + lane, ((lane & 1) ? bal >> 32 : bal & 0xFFFFFFFFull) | tagw, fast);
            }
        }
        // (behind phase B: the controller merges the uncovered columns' words while this runs)
        // ---- covered rows of my COVERED columns: += h (:355-358).  Such an entry was >= -rounding noise and h > 0, so it is NOT a zero
        // afterwards -- except in a pathological rounding case, which is counted: the controller clears the bits of these entries itself when
        // every helper reports a count of 0 (CTL_COVSUM) and reads the per-row granules (CTL_COVBITS) only otherwise.  Work item = (one of my
        // covered columns, 64 consecutive entries of the covered-row list), lane = row: the matrix is column-major, so a wave instruction
        // touches a few neighbouring lines instead of 64 scattered ones; eight items in flight per wave ----
        if (part2) {
            const u64 mycov = S.covC[g] & validC;
            const int ncc = __popcll(mycov);
            if (uwave == 2 && ((mycov >> lane) & 1)) S.hcols[__popcll(mycov & ((1ull << lane) - 1ull))] = (unsigned char)lane;   // my covered columns, ascending
            __syncthreads();
            const int nchunk = (ncr + 63) >> 6, nitem = ncc * nchunk;
            bool exc = false;
            for (int t0 = uwave * 8; t0 < nitem; t0 += 8 * (MK_THREADS / 64)) {
                double x[8]; unsigned off[8]; bool ok[8];                // (element index < 2^20)
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int t = min(t0 + q, nitem - 1);
                    const int ci = t / nchunk, ch = t - ci * nchunk;
                    const int ri = ch * 64 + lane;
                    ok[q] = t0 + q < nitem && ri < ncr;
                    off[q] = (unsigned)S.clist[min(ri, ncr - 1)] + (unsigned)nR * (unsigned)(cbase + S.hcols[ci]);
                }
#pragma unroll
                for (int q = 0; q < 8; q++) x[q] = d[off[q]];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    x[q] += h;
                    if (ok[q]) { d[off[q]] = x[q]; exc |= fabs(x[q]) < DBL_EPSILON; }
                }
            }
            if (__ballot(exc) && lane == 0) atomicAdd(&S.flag[6], 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the updated entries are out before anybody re-reads them below
            __syncthreads();
            if (S.flag[6] + force_cov > 0) {
                // a zero among them (or the test hook): the zero bits of every covered row over my covered columns as tagged granules,
                // read back from the updated entries (lane = column, as the controller's merge expects them)
                const bool act = cbase + lane < nC && ((mycov >> lane) & 1);
                for (int i0 = uwave * 8; i0 < ncr; i0 += 8 * (MK_THREADS / 64)) {
                    int rr[8]; double y[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) rr[q] = __builtin_amdgcn_readfirstlane((int)S.clist[min(i0 + q, ncr - 1)]);
#pragma unroll
                    for (int q = 0; q < 8; q++) y[q] = d[(size_t)rr[q] + coff];
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        if (i0 + q < ncr) {
                            const u64 bal = __ballot(act && fabs(y[q]) < DBL_EPSILON);
                            if (lane < 2) ctl_stx(ctl + CTL_COVBITS + ((size_t)g * MK_MAXN + i0 + q) * 2 + lane, (lane ? bal >> 32 : bal & 0xFFFFFFFFull) | tagw, fast);
                        }
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's granules have landed before the summary (below, behind a barrier) can
            }
        }
        __syncthreads();                                               // S.list / S.covR are rewritten by wave 0 for the next step
        if (part2 && tid == 0) ctl_stx(ctl + CTL_COVSUM + g * MK_PARTIAL_STRIDE, tagw | (u64)(unsigned)(S.flag[6] + force_cov), fast);   // every wave's covered-row granules are out
    }
}

// ---- small problems: cost, line minima, working matrix and zero bitmaps computed by the Munkres workgroup itself --
// same arithmetic as assoc_min_kernel / assoc_sub_kernel, two launches and a memset fewer per frame.  The layout
// handles up to MK_FUSE_MAX lines, but one CU's fp64 sqrt rate makes it slower than the tiled kernels beyond
// MK_FUSE_LINES (64 lines: 22 vs 28 us; 256 lines: 85 vs 48 us).
// Thread = (row r = tid & 255, column class tid >> 8); scratch lives in the not yet used bitmap region.
#define MK_FUSE_MAX 256
#define MK_FUSE_LINES 64
__device__ void mk_fused_cost(const AssocArgs& a, MkShared& S, int nR, int nC, bool rowsTrk)
{
    const int tid = threadIdx.x, lane = tid & 63;
    // thread = (row, column class); as many column classes as the row count leaves room for (64 rows: 16 classes of 4 columns each,
    // instead of 4 classes with three quarters of the workgroup idle -- the float64 sqrt chains are what this phase costs)
    const int RP = nR <= 64 ? 64 : (nR <= 128 ? 128 : MK_FUSE_MAX), NPART = MK_THREADS / RP;
    const int r = tid & (RP - 1), part = tid / RP;
    const bool perRow = nR <= nC;
    const int wordsR = (nR + 63) >> 6, wordsC = (nC + 63) >> 6;
    u64* lm = S.bm;                                                    // [256] keys of the line minima
    u64* zrs = S.bm + MK_FUSE_MAX;                                     // [256 rows][4 words] row-major zero bitmap
    bbox_t* colb = reinterpret_cast<bbox_t*>(S.bm + 2 * MK_FUSE_MAX + MK_THREADS);   // [256] boxes of the column side
    if (tid < MK_FUSE_MAX) lm[tid] = ~0ull;
    zrs[tid] = 0;
    // the boxes of both sides once: the row's box in registers, the column boxes in LDS (td.cpp:407-419 per element)
    bbox_t rowb = {};
    if (!a.user) {
        if (r < nR) rowb = rowsTrk ? a.trk[r] : a.det[r];
        if (tid < nC) colb[tid] = rowsTrk ? a.det[tid] : a.trk[tid];
    }
    __syncthreads();
    auto cost = [&](int c) -> double {
        if (a.user) return a.user[(size_t)r + (size_t)nR * c];
        return rowsTrk ? pair_cost(rowb, colb[c]) : pair_cost(colb[c], rowb);
    };
    // a thread's costs are kept for the second pass when they fit eight registers (64 x 64: four per thread): the float64 divide / sqrt chains
    // of pair_cost are what this phase costs, and they ran twice
    const bool keep = (nC + NPART - 1) / NPART <= 8;                   // workgroup-uniform
    double cv[8];
    u64 best = ~0ull;                                                  // pass 1 (hungarian.cpp:69-81 / :107-119)
    if (keep) {
#pragma unroll
        for (int q = 0; q < 8; q++) { const int c = part + q * NPART; cv[q] = (r < nR && c < nC) ? cost(c) : 0.0; }
    }
    for (int c = part, q = 0; c < nC; c += NPART, q++) {
        double cq = 0.0;
        if (keep) {
#pragma unroll
            for (int t = 0; t < 8; t++) cq = (t == q) ? cv[t] : cq;
        }
        const u64 kk = r < nR ? dkey(keep ? cq : cost(c)) : ~0ull;
        if (perRow) { if (kk < best) best = kk; }
        else {
            u64 m = kk;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const u64 o = __shfl_down(m, off); if (o < m) m = o; }
            if (lane == 0) atomicMin(&lm[c], m);
        }
    }
    if (perRow && r < nR && best != ~0ull) atomicMin(&lm[r], best);
    __syncthreads();
    const double rmin = (perRow && r < nR) ? dunkey(lm[r]) : 0.0;      // pass 2
    for (int c = part, q = 0; c < nC; c += NPART, q++) {
        bool z = false;
        if (r < nR) {
            double cq = 0.0;
            if (keep) {
#pragma unroll
                for (int t = 0; t < 8; t++) cq = (t == q) ? cv[t] : cq;
            }
            const double v = keep ? cq : cost(c);
            const double dv = v - (perRow ? rmin : dunkey(lm[c]));
            a.ws.dist[(size_t)r + (size_t)nR * c] = dv;
            z = fabs(dv) < DBL_EPSILON;
        }
        const u64 bal = __ballot(z);
        if (lane == 0 && (r >> 6) < wordsR) a.ws.zc[(size_t)c * wordsR + (r >> 6)] = bal;
        if (z) atomicOr(&zrs[r * 4 + (c >> 6)], 1ull << (c & 63));
    }
    __syncthreads();
    for (int i = tid; i < nR * wordsC; i += MK_THREADS) { const int rr = i / wordsC, w = i - rr * wordsC; a.ws.zr[(size_t)rr * wordsC + w] = zrs[rr * 4 + w]; }
    __syncthreads();
}

// Working matrix d = cost - row minimum (hungarian.cpp:83-89) and the zero bitmaps in both orientations, computed by the Munkres
// workgroup itself (thread = row, rows <= columns): what assoc_sub_kernel does chip-wide.  Behind the fast path the dense
// emulation is the rare last resort, so its preparation is not launched per frame any more (one early-exit launch less on every
// frame); a stream that keeps needing it gets the chip-wide kernel back through the host-side hint.  ~0.4 ms at 1024 x 1024.
// cc_rmin (patch step of a provisionally committed frame): the line minima have been re-armed since; the row's minimum is its first candidate
__device__ void mk_prepare_dense(const AssocArgs& a, int nR, int nC, bool rowsTrk, const double* cc_rmin = nullptr)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = tid, wordsR = (nR + 63) >> 6, wordsC = (nC + 63) >> 6;
    const double rmin = r < nR ? (cc_rmin ? cc_rmin[(size_t)r * LAP_K] : dunkey(a.linemin[r])) : 0.0;            // the row scan left the row minima here
    u64 zw = 0;
    for (int c = 0; c < nC; c++) {
        bool z = false;
        if (r < nR) {
            const double d = elem_cost(a, r, c, nR, rowsTrk) - rmin;
            a.ws.dist[(size_t)r + (size_t)nR * c] = d;
            z = fabs(d) < DBL_EPSILON;
        }
        const u64 bal = __ballot(z);
        if (lane == 0 && wave < wordsR) a.ws.zc[(size_t)c * wordsR + wave] = bal;
        if (z) zw |= 1ull << (c & 63);
        if ((c & 63) == 63 || c == nC - 1) { if (r < nR) a.ws.zr[(size_t)r * wordsC + (c >> 6)] = zw; zw = 0; }
    }
    __threadfence_block();
    __syncthreads();
}

// what the last kernel of a fast-path chain leaves behind: statistics, the scheduling hints for the host, re-armed header
__device__ inline void lap_final_bookkeeping(const AssocArgs& a, int mode, bool certified)
{
    const LapWs& L = a.ws.lap;
    L.hdr[LAP_H_LAST + 15] = mode == 0 ? 0 : (certified ? 1 : 2);   // what this frame used: 0 certificate, 1 sparse emulation, 2 dense emulation
    const int outcome = L.hdr[LAP_H_LAST], dense_ran = L.hdr[LAP_H_DENSE];
    if (dense_ran) { L.hdr[LAP_H_DSTAT + 3] += 1; if (mode == 0) L.hdr[LAP_H_DSTAT + 4] += 1; }
    if (a.ws.dense_hint) *reinterpret_cast<volatile int*>(a.ws.dense_hint) = (certified ? 0 : 1) | ((dense_ran || outcome == 1 || outcome == 2) ? 2 : 0);
    if (mode == 1) L.hdr[LAP_H_CUM + (certified ? 8 : 9)] += 1;
    // re-arm for the next launch (this workgroup is the last reader)
    L.hdr[LAP_H_NEDGES] = 0; L.hdr[LAP_H_VIOL] = 0; L.hdr[LAP_H_BAD] = 0; L.hdr[LAP_H_SOLVE] = 5; L.hdr[LAP_H_SPVIOL] = 0; L.hdr[LAP_H_MODE] = 2;
    L.hdr[LAP_H_DENSE] = 0; L.hdr[LAP_H_DONE] = 0; L.hdr[LAP_H_CERT] = 0; *L.cmaxkey = 0ull;
    // (LAP_H_VERDICT / LAP_H_EMU carry their chain's sequence number since round 6: a stale value is recognised, and the emulation's kernel
    //  of a certified frame may still be about to read the verdict -- they are not re-armed)
}

// ---- provisional commits (round 6; mot_dev.h: ProvRec, lap_kernels.hip: lap_try_provisional) ---------------------------------------
// The patch step: `asg` is what an order-exact emulation returned for the provisionally committed frame.  It must be the committed matching M
// (lap.colOfRow) or M with the columns of rows A and B exchanged; in the second case the two tracks take over what their shadow slots hold --
// model, alpha, response, position, scale, flags: the state of "adopted the other detection", with or without the predict in between -- and,
// if a predict launch ran in between, the shadow items' predicted boxes.  Anything else latches a device error (never seen; the certificate's
// argument excludes it).  One 1024-thread workgroup.
template <typename AT>
__device__ void prov_apply(const AssocArgs& a, const LifeArgs& life, const AT* asg, int nR, bbox_t* pred_cur, bool by_dense)
{
    const int tid = threadIdx.x;
    const LapWs& L = a.ws.lap;
    const ProvRec* rec = life.prov.rec;
    const DLState& D = life.S;
    const KcfPool& kp = life.kp;
    const int np = min(rec->npairs, MOT_PROV_PAIRS);
    bool bad = false, in_pair = false; unsigned swapmask = 0;
    for (int i = 0; i < np; i++) {
        const ProvPair pr = rec->pr[i];
        const int gA = (int)asg[pr.rowA], gB = (int)asg[pr.rowB];
        const bool same = gA == pr.colA && gB == pr.colB, swapped = gA == pr.colB && gB == pr.colA;
        if (!(same || swapped)) bad = true;
        if (swapped) swapmask |= 1u << i;
        if (tid == pr.rowA || tid == pr.rowB) in_pair = true;
    }
    if (tid < nR && !in_pair && (int)asg[tid] != (int)L.colOfRow[tid]) bad = true;   // every other row is forced
    const bool anybad = __syncthreads_or(bad) != 0;
    if (swapmask && !anybad) {
        // This copy sits between the emulation's end and the next row scan.  Its first form walked the tracks one by one -- record, model, scalars, each a chain of
        // dependent global round trips of ~1 us -- and cost 10 us per exchanged pair (profiles/r06_timeline_prov.txt: the patch step ended 5 us after the emulation
        // without an exchange, 16 / 26 us with one / two).  Now: lane k moves track k's scalars (all tracks at once), and every
        // track's model loads are in flight before its first store.
        const int tot = MOT_NCHAN * kp.nbins;
        if (tid < 2 * np && ((swapmask >> (tid >> 1)) & 1)) {
            const ProvTrack t = rec->t[tid];
            if (t.sh >= 0) {
                kp.pos[t.slot] = kp.pos[t.sh]; kp.scale[t.slot] = kp.scale[t.sh]; kp.first_update[t.slot] = kp.first_update[t.sh];
                D.pend_det[t.slot] = D.pend_det[t.sh];
                D.bbox[t.newpos] = t.box_alt;                          // tracker_info.bbox = the adopted detection (td.cpp:519-521)
                if (pred_cur) pred_cur[t.newpos] = pred_cur[rec->n_new + (t.sh - life.prov.sh_base)];
            }
        }
#pragma unroll 1
        for (int k = 0; k < 2 * np; k++) {
            if (!((swapmask >> (k >> 1)) & 1)) continue;
            const int shk = rec->t[k].sh, slk = rec->t[k].slot;
            if (shk < 0) continue;
            const float2* xs = kp.xm + (size_t)shk * tot; float2* xd = kp.xm + (size_t)slk * tot;
            const float* as = kp.alpha + (size_t)shk * kp.nbins; const float* rs = kp.response + (size_t)shk * kp.nb;
            for (int i0 = 0; i0 < tot; i0 += 8 * MK_THREADS) {
                float2 v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) { const int i = i0 + q * MK_THREADS + tid; if (i < tot) v[q] = xs[i]; }
                const float av = (i0 == 0 && tid < kp.nbins) ? as[tid] : 0.f;
                // the response map belongs to the last predict: the shadow has one only if a predict launch ran since the clone (else the track keeps its own)
                const float rv = (i0 == 0 && pred_cur && tid < kp.nb) ? rs[tid] : 0.f;
#pragma unroll
                for (int q = 0; q < 8; q++) { const int i = i0 + q * MK_THREADS + tid; if (i < tot) xd[i] = v[q]; }
                if (i0 == 0 && tid < kp.nbins) kp.alpha[(size_t)slk * kp.nbins + tid] = av;
                if (i0 == 0 && pred_cur && tid < kp.nb) kp.response[(size_t)slk * kp.nb + tid] = rv;
            }
            for (int i = tid + MK_THREADS; i < kp.nbins; i += MK_THREADS) kp.alpha[(size_t)slk * kp.nbins + i] = as[i];          // (templates beyond 1024 bins / cells)
            if (pred_cur) for (int i = tid + MK_THREADS; i < kp.nb; i += MK_THREADS) kp.response[(size_t)slk * kp.nb + i] = rs[i];
        }
    }
    if (tid == 0) {
        if (anybad) D.err[5] = rec->seq;                               // sticky: devloop_check reports it
        for (int k = 0; k < 2 * np; k++) if (rec->t[k].sh >= 0) D.pend_det[rec->t[k].sh] = -1;
        *D.loc_count = rec->n_new;                                     // the shadow items leave the predict list
        L.hdr[LAP_H_PROV] = 0; L.hdr[LAP_H_PMODE] = 0;
        if (!anybad) L.hdr[LAP_H_PSTAT + 1] += __popc(swapmask);
        if (by_dense) { L.hdr[LAP_H_PSTAT + 2] += 1; L.hdr[LAP_H_LAST + 15] = 2; L.hdr[LAP_H_CUM + 9] += 1; } else L.hdr[LAP_H_CUM + 8] += 1;
    }
}

// HELP = false: one workgroup does everything (no helper code compiled in: it would cost the hot loops registers).
// HELP = true : launched with 1 + MK_HELPERS workgroups; workgroups 1.. run mk_helper_loop.
// lap_mode: the fast path (lap_kernels.hip) ran in front of this launch; if its certificate holds (lap_certify.h) the
// solver's matching IS the reference's assignment and the emulation is skipped.
// PATCH: the instantiation launched as the patch step of a provisionally committed frame (launch_prov_patch; a kernel of its own so that the
// final kernel of every frame keeps its registers)
template <bool HELP, bool PATCH = false>
__global__ void __launch_bounds__(MK_THREADS) munkres_kernel(AssocArgs a, int want_cost, LifeArgs life, int lap_mode)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char mk_raw[];
    MkShared& S = *reinterpret_cast<MkShared*>(mk_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int nR, nC; bool rowsTrk; resolve_dims(a, nR, nC, rowsTrk);
    double* __restrict__ d = a.ws.dist;
    int* stat = a.ws.status;
    const long long t_begin = wall_clock64();
    const long long c_begin = clock64();
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int nhelp = HELP ? ((int)gridDim.x - 1) / MK_XCDS : 0;                   // 0: everything in this workgroup
    // Workgroups are dealt to the 8 XCDs round-robin by index, so blocks 0, 8, 16, ... share the controller's XCD: the hand-offs
    // then stay behind one L2 (-0.4 us per step 5; placement only affects speed, never correctness).  The other blocks exit.
    // an earlier kernel of the chain (the solver's fused tail / the sparse emulation) has already decided AND committed this frame
    // (lifecycle step included): bookkeeping only.  (The live count, and with it nR / nC, already belong to the next frame.)
    // lap_mode: bit 0 the fast path ran in front of this launch; bit 1 (round 6) its sparse emulation is a kernel of its own on the emulation
    // stream; bit 2 this launch is the PATCH STEP of the chain a.seq (launched behind the next frame's predict, or at a synchronisation point)
    constexpr bool patch = PATCH;
    if (patch) {
        const LapWs& Lp = a.ws.lap;
        if (Lp.hdr[LAP_H_PROV] != (int)a.seq || Lp.hdr[LAP_H_PROV] == 0) return;   // that frame owes nothing (certified, or committed by an emulation): the usual case
        // the emulation's kernel was running when the frame was committed provisionally (lap_try_provisional checked): wait for its end
        for (int spins = 0;; spins++) {
            if (tid == 0) S.flag[6] = tagged_value(__hip_atomic_load(&Lp.hdr[LAP_H_EMU], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT), a.seq, 2);
            __syncthreads();
            const int st = S.flag[6];
            __syncthreads();
            if (st == 2) break;
            if (spins > 50000000) { if (tid == 0) { life.S.err[5] = (int)a.seq; *life.S.loc_count = life.prov.rec->n_new; Lp.hdr[LAP_H_PROV] = 0; } return; }   // ("cannot happen")
            __builtin_amdgcn_s_sleep(8);
        }
        if (Lp.hdr[LAP_H_PMODE] == 1 && !(want_cost & 8)) { prov_apply(a, life, Lp.spAssign, nR, reinterpret_cast<bbox_t*>(a.cost_only), false); return; }
        // the sparse emulation refused (an entry outside its candidate lists could have mattered): the dense emulation below decides the bit
    } else if ((lap_mode & 2) != 0 && (!HELP || blockIdx.x == 0)) {
        // Stream emulation, frame neither certified nor committed provisionally (verdict 2): the emulation's kernel decides -- and commits -- the
        // frame; wait for its end, or, if it has not even started, take the frame away from it (it then returns at once: exactly one writer)
        const LapWs& Lp = a.ws.lap;
        for (int spins = 0;; spins++) {
            if (tid == 0) {
                int go = 1;
                if (tagged_value(__hip_atomic_load(&Lp.hdr[LAP_H_VERDICT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a.seq, 0) == 2) {
                    const int old = __hip_atomic_load(&Lp.hdr[LAP_H_EMU], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    const int st = tagged_value(old, a.seq, 2);
                    if (st == 0) go = atomicCAS(&Lp.hdr[LAP_H_EMU], old, tagged_word(a.seq, 3)) == old ? 1 : 0;
                    else if (st == 1) go = spins > 50000000 ? 1 : 0;
                }
                S.flag[6] = go;
            }
            __syncthreads();
            const int go = S.flag[6];
            __syncthreads();
            if (go) break;
            __builtin_amdgcn_s_sleep(8);
        }
    }
    const bool committed = (lap_mode & 1) && a.ws.lap.hdr[LAP_H_DONE] != 0;
    if (HELP && blockIdx.x > 0) {
        if (blockIdx.x % MK_XCDS != 0) return;
        // ONE decision per helper workgroup.  With the emulation a kernel of its own (lap_mode bit 1) the frame can be committed -- DONE set, the live count
        // behind nR / nC changed -- while this workgroup starts: waves that had read DONE on either side of that store disagreed, wave 0 left, the barriers
        // of the helper loop released without it and the others ran on the LDS contents of whoever held the CU before (round 6: the memory fault of
        // the two-context soak, profiles/r06_prov_soak.log).  Thread 0's reading counts; a helper that goes ahead on the stale side is released by the
        // controller's EXIT granules and has written nothing by then.
        if (tid == 0) { S.flag[0] = committed ? 1 : 0; S.flag[1] = nR; S.flag[4] = nC; }
        __syncthreads();
        const int h_done = S.flag[0], hR = S.flag[1], hC = S.flag[4];
        __syncthreads();
        if (hR > 0 && hC > 0 && !h_done) mk_helper_loop(a, S, hR, hC, (want_cost >> 3) & 1);
        return;
    }
    if (committed) {
        const int mode = a.ws.lap.hdr[LAP_H_MODE];
        __syncthreads();
        if (tid == 0) { lap_final_bookkeeping(a, mode, true); stat[15] = 0; stat[0] = -1; stat[1] = 0; stat[2] = 0; stat[12] = (int)(wall_clock64() - t_begin); }
        if (tid < MK_MAXN) a.linemin[tid] = ~0ull;
        if (HELP && uwave == 0 && a.ws.ctl) {                            // helpers are spinning on the cover-mask granules: release them
            if (lane == 0) ctl_st(a.ws.ctl + CTL_EPOCH, (u64)((unsigned)ctl_ld(a.ws.ctl + CTL_EPOCH) + 1));
            ctl_st(a.ws.ctl + CTL_COV + lane, (u64)MK_TAG_EXIT << 32);
        }
        return;
    }
    if (nR <= 0 || nC <= 0) {
        if (tid == 0) { *a.ws.cost = 0.0; stat[15] = 0; }
        for (int r = tid; r < max(nR, 0); r += MK_THREADS) a.ws.assignment[r] = -1;
        if (life.enabled) { __syncthreads(); dl_lifecycle_body(life.S, life.kp, life.kal, life.trk_pred, life.dets, life.nD, a.ws.assignment, reinterpret_cast<int*>(S.bm) + 4096); }
        return;
    }
    if (!HELP && (want_cost & 2)) mk_fused_cost(a, S, nR, nC, rowsTrk);
    if (tid == 0) stat[15] = 0;                                        // set again only if a helper hand-off times out
    // lap_mode: mk_sparse_kernel left its verdict: 0 = certified unique optimum, 1 = the sparse order-exact emulation ran and
    // mk_postcheck_kernel found no entry outside the candidate lists that could have mattered -- in both cases the assignment
    // is the reference's and the dense emulation below is skipped
    bool certified = false; const short* given = nullptr;
    if (patch) mk_prepare_dense(a, nR, nC, rowsTrk, a.ws.lap.ccost);
    if (lap_mode & 1) {
        const LapWs& L = a.ws.lap;
        const int mode = L.hdr[LAP_H_MODE], spviol = L.hdr[LAP_H_SPVIOL];
        if (mode == 0) { certified = true; given = L.colOfRow; }
        else if (mode == 1 && !spviol) { certified = true; given = L.spAssign; }
        __syncthreads();
        if (tid == 0) lap_final_bookkeeping(a, mode, certified);
        if (!certified && (want_cost & 4)) mk_prepare_dense(a, nR, nC, rowsTrk);   // the chip-wide preparation was not launched
    }
    if (tid < MK_MAXN) a.linemin[tid] = ~0ull;                         // re-arm the line minima for the next launch's assoc_min_kernel (no memset per frame)
    const int wordsR = (nR + 63) >> 6, wordsC = (nC + 63) >> 6;
    const bool perRow = nR <= nC;
    u64* ctl = a.ws.ctl; unsigned myseq = 0;
    const unsigned epoch = (HELP && ctl) ? (unsigned)ctl_ld(ctl + CTL_EPOCH) : 0;
    const int minDim = perRow ? nR : nC;
    int n_s4 = 0, n_s5 = 0, n_sw = 0, n_cov5 = 0, ncu0 = 0; long long t_s3 = 0, t_s5 = 0, t_h0 = 0, t_h1 = 0, t_h2 = 0, t_h3 = 0;     // wave-0 / thread-0 statistics

    for (int i = tid; i < MK_MAXN; i += MK_THREADS) { S.starColOfRow[i] = -1; S.starRowOfCol[i] = -1; S.primeColOfRow[i] = -1; }
    if (tid < MK_MAXW) { S.covR[tid] = 0; S.covC[tid] = 0; S.hzl[tid] = 0; }
    if (tid == 0) S.flag[7] = 0;
    if (tid < 2 * MK_MAXW) { S.taken32[tid] = 0; S.cont32[tid] = 0; }
    if (certified) {                                                   // the unique optimum: nothing to emulate
        if (tid < nR) S.starColOfRow[tid] = given[tid];
        __syncthreads();
    }
    // ---- steps 1 + 2a: initial stars (hungarian.cpp:93-101 / :128-139) ----
    // lines (rows if perRow, else columns) are scanned in order; each takes its first zero whose cross line
    // is still free.  A line whose first zero sits in a cross line with exactly ONE zero ("clean") can be
    // starred out of order: no other line can ever claim that cross line.  Only the contested lines go
    // through the ordered scan (wave 0).
    if (!certified) {
        const int nL = perRow ? nR : nC, nX = perRow ? nC : nR;
        const int W = perRow ? wordsC : wordsR, WX = perRow ? wordsR : wordsC;
        const u64* lineBm = perRow ? a.ws.zr : a.ws.zc;
        const u64* crossBm = perRow ? a.ws.zc : a.ws.zr;
        for (int i = tid; i < nL * W; i += MK_THREADS) S.bm[(i / W) * MK_MAXW + (i % W)] = lineBm[i];
        // zeros per cross line: the bitmap is read once, coalesced (consecutive lanes = consecutive words), and the WX
        // words of a line are summed across lanes (WX is 1..16: segments of 16 lanes never straddle a wave)
        for (int i0 = 0; i0 < nX * 16; i0 += MK_THREADS) {
            const int i = i0 + tid, x = i >> 4, w = i & 15;
            int cnt = (x < nX && w < WX) ? __popcll(crossBm[(size_t)x * WX + w]) : 0;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 16);
            if (w == 0 && x < nX) S.crosscnt[x] = (unsigned short)min(cnt, 65535);
        }
        __syncthreads();
        for (int l = tid; l < nL; l += MK_THREADS) {
            int fz = -1;
            for (int w = 0; w < W; w++) { const u64 m = S.bm[l * MK_MAXW + w]; if (m) { fz = w * 64 + __ffsll((long long)m) - 1; break; } }
            if (fz >= 0) {
                if (S.crosscnt[fz] == 1) {
                    const int row = perRow ? l : fz, col = perRow ? fz : l;
                    S.starColOfRow[row] = (short)col; S.starRowOfCol[col] = (short)row;
                    atomicOr(&S.taken32[fz >> 5], 1u << (fz & 31));
                } else atomicOr(&S.cont32[l >> 5], 1u << (l & 31));
            }
        }
        __syncthreads();
        if (wave == 0) {
            const u64 cont = (lane < MK_MAXW) ? ((u64)S.cont32[2 * lane] | ((u64)S.cont32[2 * lane + 1] << 32)) : 0;
            const int ncont = wave_list_bits(cont, 0, S.clist, lane);
            u64 taken = (lane < W) ? ((u64)S.taken32[2 * lane] | ((u64)S.taken32[2 * lane + 1] << 32)) : 0;
            for (int q0 = 0; q0 < ncont; q0 += 8) {
                u64 m[8]; int ln[8];
#pragma unroll
                for (int k = 0; k < 8; k++) { ln[k] = (q0 + k < ncont) ? S.clist[q0 + k] : -1; m[k] = (lane < W && ln[k] >= 0) ? S.bm[ln[k] * MK_MAXW + lane] : 0; }
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
        {   // covered columns = starred columns (both branches end with exactly that; rows uncovered :138-139)
            const bool has = tid < nC && S.starRowOfCol[tid] >= 0;
            const u64 bal = __ballot(has);
            if (lane == 0) S.covC[wave] = bal;
        }
        __syncthreads();
    }
    int ncov = 0;
    for (int w = 0; w < wordsC; w++) ncov += __popcll(S.covC[w]);
    bool done = certified || (ncov == minDim);                         // step 2b (:216-237)
    if (!done) {
        __syncthreads();
        for (int i = tid; i < nC * wordsR; i += MK_THREADS) S.bm[(i / wordsR) * MK_MAXW + (i % wordsR)] = a.ws.zc[i];
        __syncthreads();
    }
    // hz[c]: column c holds at least one zero (any row).  Exact for every column that is uncovered while no
    // row is covered: those are never-starred columns, whose entries change in step 5 only, where hz is rebuilt.
    if (!done) {
        const bool has = [&] { bool h2 = false; if (tid < nC) for (int w = 0; w < wordsR; w++) h2 |= S.bm[tid * MK_MAXW + w] != 0; return h2; }();
        const u64 bal = __ballot(has);
        if (lane == 0) S.hz[wave] = bal;
        __syncthreads();
    }
    const long long t_init = wall_clock64();
    // cover masks live in wave 0's registers: lane w (<16) holds word w of covC / hz / phaseUnc, lane l holds
    // word (l & 15) of covR (replicated over the four 16-lane quarters); LDS mirrors serve the other waves
    u64 cC = (lane < MK_MAXW) ? S.covC[lane] : 0, cR16 = 0, phaseUnc = 0, hz = (lane < MK_MAXW) ? S.hz[lane] : 0;
    // hzl: columns that may hold a zero in an UNCOVERED row.  Within a phase rows only get covered and zero bits change in step 5 only,
    // so a column the sweep has found empty stays empty until the next step 5 / augmentation: its bit is dropped and later events of
    // the cycle do not test it again (a crowded noisy frame uncovers hundreds of star columns per phase, each a candidate otherwise)
    u64 hzl = hz;
    const u64 vC = (lane < wordsC) ? ((lane == wordsC - 1 && (nC & 63)) ? ((1ull << (nC & 63)) - 1) : ~0ull) : 0;
    bool covRany = false;
    int guard = 0;
    // Do all 17 workgroups share one XCD (they should: blocks 0, 8, 16, ...)?  Every helper reported its XCC id at its
    // start, tens of microseconds ago.  If yes the hand-off stores stay in that XCD's L2 (ctl_stx); if a report is missing
    // or differs, the agent-scope form is used.  The verdict is published before the first cover-mask publish.
    bool fast = false;
    if (HELP && nhelp > 0) {
        if (uwave == 0) {
            const u64 rep = (lane < nhelp) ? ctl_ld(ctl + CTL_XCCTAB + lane) : 0;
            const bool okl = lane >= nhelp || ((unsigned)(rep >> 32) == epoch + 1 && (int)(rep & 0xF) == xcc_id());
            const bool all = __ballot(!okl) == 0;
            if (lane == 0) { ctl_st(ctl + CTL_MODE, (u64)(all ? 1 : 0) | ((u64)(epoch + 1) << 32)); S.flag[5] = all; }
        }
        __syncthreads();
        fast = S.flag[5] != 0;
        if (tid == 0) stat[10] = fast;
    }
    while (!done) {
        const long long t_a = wall_clock64();
        // ========== steps 3 / 4 / 2a / 2b (:240-334, :192-237): wave 0 until a step 5 is needed ==========
        if (wave == 0) {
            int action = 0;                                           // 2: sweep found nothing -> step 5, 3: finished
            int from = 0; bool found_in_sweep = false;                // state of the current column sweep (:249)
            while (action == 0) {
                if (++n_sw > 8 * MK_MAXN * MK_MAXN) { action = 3; break; }   // safety bound, never reached
                int col = -1, row = -1;
                if (!covRany && from == 0) {
                    // fast sweep: no row is covered, so the first uncovered column holding a zero is the hit
                    const u64 hm = (lane < MK_MAXW) ? (hz & ~cC & vC) : 0;
                    col = wave_first_bit(hm, lane, wordsC);
                    if (col < 0) { action = 2; break; }
                    const u64 mw = (lane < wordsR) ? S.bm[col * MK_MAXW + lane] : 0;
                    row = wave_first_bit(mw, lane, wordsR);
                    if (row < 0) { if (lane == (col >> 6)) { hz &= ~(1ull << (col & 63)); hzl &= ~(1ull << (col & 63)); } continue; }   // stale hint
                } else {
                    // general sweep: candidate columns (uncovered, >= from, may hold a zero) in ascending order, four per
                    // LDS read: quarter q of the wave tests column c_q against the uncovered rows
                    const int q = lane >> 4, wd = lane & 15;
                    const int fw = from >> 6;
                    u64 cand = (lane < MK_MAXW) ? (~cC & vC & hzl) : 0;   // hzl is a superset of "has a zero in an uncovered row"
                    if (lane < fw) cand = 0; else if (lane == fw) cand &= ~0ull << (from & 63);
                    {   // the first candidate alone (with the exact mask of the last step 5 it is almost always the hit): one LDS read, no
                        // selection among four columns
                        const int c0 = wave_first_bit(cand, lane, MK_MAXW);
                        if (c0 >= 0) {
                            const u64 m0 = (lane < wordsR) ? (S.bm[c0 * MK_MAXW + lane] & ~cR16) : 0;   // lanes 0..15: cR16 holds word `lane`
                            const u64 bal0 = __ballot(m0 != 0);
                            if (bal0) {
                                const int fl = __ffsll((long long)bal0) - 1;
                                col = c0; row = fl * 64 + __ffsll((long long)readlane64(m0, fl)) - 1;
                            } else if (lane == (c0 >> 6)) { cand &= ~(1ull << (c0 & 63)); hzl &= ~(1ull << (c0 & 63)); }
                        }
                    }
                    while (col < 0) {
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
#pragma unroll
                        for (int j = 0; j < 4; j++)                   // a tested column without a live zero leaves the candidates of this cycle
                            if (cs[j] >= 0 && ((bal >> (16 * j)) & 0xFFFFull) == 0 && lane == (cs[j] >> 6)) hzl &= ~(1ull << (cs[j] & 63));
                        if (bal) {
                            const int fl = __ffsll((long long)bal) - 1;   // lowest quarter = lowest column, lowest word = lowest row
                            const int qq = fl >> 4;
                            col = qq == 0 ? cs[0] : qq == 1 ? cs[1] : qq == 2 ? cs[2] : cs[3];
                            const u64 word = readlane64(m, fl);
                            row = (fl & 15) * 64 + __ffsll((long long)word) - 1;
                            break;
                        }
                        if (cs[3] < 0) break;
                    }
                    if (col < 0) {                                    // end of this sweep (:246-248)
                        if (found_in_sweep) { found_in_sweep = false; from = 0; continue; }
                        action = 2; break;
                    }
                }
                const int sc = S.starColOfRow[row];
                if (lane == 0) S.primeColOfRow[row] = (short)col;      // prime zero (:255)
                if (sc < 0) {
                    // ---------- step 4 (:283-334): augment along the star/prime path ----------
                    n_s4++;
                    int last = col;                                   // the path ends in the one column that gains a star
                    if (lane == 0) {
                        int cr = row, cc = col;
                        for (int it = 0; it <= nR + nC; it++) {
                            const int old_r = S.starRowOfCol[cc];
                            S.starColOfRow[cr] = (short)cc; S.starRowOfCol[cc] = (short)cr;
                            if (old_r < 0) break;
                            cc = S.primeColOfRow[old_r]; cr = old_r;
                            if (cc < 0) break;                        // broken invariant: never for consistent state
                        }
                        last = cc;
                    }
                    last = __builtin_amdgcn_readfirstlane(last);
                    // delete primes (only covered rows and `row` carry one), uncover rows (:324-330)
                    if (lane < MK_MAXW) { u64 t = cR16; while (t) { const int r2 = lane * 64 + __ffsll((long long)t) - 1; S.primeColOfRow[r2] = -1; t &= t - 1; } }
                    if (lane == 0) S.primeColOfRow[row] = -1;
                    cR16 = 0; covRany = false;
                    // step 2a (:198-209): every starred column is covered again; the columns uncovered in this
                    // phase kept a star, and the last column of the path just received its first one
                    cC |= phaseUnc; if (last >= 0 && lane == (last >> 6)) cC |= 1ull << (last & 63);
                    phaseUnc = 0;
                    hzl = hz;                                         // every row is uncovered again; the uncovered columns are never-starred ones (hz is exact for them)
                    if (lane < MK_MAXW) { S.covR[lane] = 0; S.covC[lane] = cC; }
                    int total = 0;
                    for (int w = 0; w < wordsC; w++) total += __popcll(readlane64(cC, w));
                    if (total == minDim) { action = 3; break; }      // step 2b
                    from = 0; found_in_sweep = false;                 // step 3 starts over
                    continue;
                }
                if ((lane & 15) == (row >> 6)) cR16 |= 1ull << (row & 63);                       // :270
                covRany = true;
                if (lane < MK_MAXW) {
                    if (lane == (sc >> 6)) { cC &= ~(1ull << (sc & 63)); phaseUnc |= 1ull << (sc & 63); hzl |= 1ull << (sc & 63); }   // :271 (its zeros are unknown: a candidate)
                    S.covR[lane] = cR16; S.covC[lane] = cC;
                }
                found_in_sweep = true;
                from = col + 1;                                        // the sweep continues with the next column (:273)
            }
            // list of the uncovered columns for step 5 (with helper workgroups it is built after the publish, off the critical path)
            if (action == 2 && !(HELP && nhelp > 0)) { const int ncu = wave_list_bits(~cC & vC, 0, S.list, lane); if (lane == 0) S.flag[1] = ncu; }
            if (lane == 0) S.flag[0] = action;
        }
        __syncthreads();
        const int action = S.flag[0];
        const long long t_b = wall_clock64();
        t_s3 += t_b - t_a;
        if (action == 3) { done = true; break; }
        // ================= step 5 (:337-368): one row per thread =================
        n_s5++;
        if (HELP && nhelp > 0) {
            // ---- step 5 on the helper workgroups: wave 0 publishes the cover masks and waits until every helper has
            // published its partial minimum (the helpers take h = min of the 16 themselves); then all waves poll and merge
            // the new zero bits ----
            ++myseq;
            if (uwave == 0) {
                // the slots re-armed during the previous step 5 must have landed before anybody can act on this publish
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                {   // 64 granules = 32 cover-mask words, one store instruction
                    const int j = lane >> 1;
                    const u64 src = j < MK_MAXW ? S.covR[j] : S.covC[j - MK_MAXW];
                    ctl_stx(ctl + CTL_COV + lane, ((lane & 1) ? src >> 32 : src & 0xFFFFFFFFull) | ((u64)(epoch + myseq) << 32), fast);
                }
                const long long tq0 = wall_clock64(); t_h0 += tq0 - t_b;
                // uncovered columns and covered rows, ascending (for the merge); overlaps the helpers' phase A
                const int ncu_l = wave_list_bits(~cC & vC, 0, S.list, lane);
                const int ncr = wave_list_bits((lane < MK_MAXW) ? S.covR[lane] : 0, 0, S.clist, lane);
                if (lane == 0) { S.flag[1] = ncu_l; S.flag[2] = ncr; }
                int spins = 0; bool lost = false;
                u64 hk;
                for (;;) {                                              // the granules of this step cannot arrive before all 16 partial minima exist
                    hk = (lane < nhelp) ? ctl_ld(ctl + CTL_PARTIAL + ((myseq & 1) * MK_HELPERS + lane) * MK_PARTIAL_STRIDE) : 0;
                    if (!__ballot(hk == MK_HSENT)) break;
                    if (++spins > MK_SPIN_LIMIT) { lost = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                const long long tq1 = wall_clock64(); t_h1 += tq1 - tq0;
                if (lost && lane == 0) { stat[15] = 1; S.flag[7] = 1; }
            }
            __syncthreads();
            if (S.flag[7]) break;                                      // helpers lost: give up (status[15] != 0)
            const long long tq2 = wall_clock64();
            const int ncu = S.flag[1], ncr = S.flag[2];
            if (n_s5 == 1) ncu0 = ncu;
            // ---- collect + merge: every thread polls exactly the tagged granules it merges.  Uncovered columns get their
            // complete new words (BMOUT); covered columns only the covered rows' bits (COVBITS). ----
            {
                const unsigned tag = epoch + myseq;
                const bool ccol = ncr > 0 && tid < nC && ((S.covC[wave] >> lane) & 1);
                const bool wcov = __ballot(ccol) != 0;                  // this wave's column word holds covered columns
                // BMOUT: one round covers 128 uncovered columns (four granule slots per thread: a column's 32 granules are contiguous, so a
                // wave instruction reads two whole columns = 4 lines).  COVBITS (only when a helper reports a zero there): 4 covered rows a round
                bool lostB = false;
                constexpr int MG = 4;                                  // granule slots per thread and round: 4 x 1024 / 32 = 128 uncovered columns a round
                auto merge_rounds = [&](const int rounds, const bool with_bm, const bool with_cov) {
                    for (int rd = 0; rd < rounds && !lostB; rd++) {
                        int bk[MG]; bool hb[MG]; const u64* bp[MG]; u64 bv[MG] = {0, 0, 0, 0};
#pragma unroll
                        for (int j = 0; j < MG; j++) {
                            const int gi = (rd * MG + j) * MK_THREADS + tid;
                            bk[j] = gi >> 5;
                            hb[j] = with_bm && bk[j] < ncu && ((gi & 31) >> 1) < wordsR;
                            bp[j] = ctl + CTL_BMOUT + (size_t)S.list[min(bk[j], ncu - 1)] * MK_MAXW * 2 + (gi & 31);
                        }
                        const int nrow = with_cov ? min(ncr - rd * 4, 4) : 0;   // covered rows of this round (may be <= 0)
                        const bool hc = wcov && lane < 2 * nrow;
                        const u64* cp = ctl + CTL_COVBITS + ((size_t)wave * MK_MAXN + rd * 4) * 2 + lane;
                        u64 cvv = 0;
                        bool mylost = false;
                        for (int spins = 0;; ) {                            // every wave polls its own granules at its own pace
                            bool ok = true;
#pragma unroll
                            for (int j = 0; j < MG; j++) if (hb[j]) bv[j] = ctl_ld(bp[j]);
                            if (hc) cvv = ctl_ld(cp);
#pragma unroll
                            for (int j = 0; j < MG; j++) if (hb[j]) ok &= (unsigned)(bv[j] >> 32) == tag;
                            if (hc) ok &= (unsigned)(cvv >> 32) == tag;
                            if (!__ballot(!ok)) break;
                            if (++spins > MK_SPIN_LIMIT) { mylost = true; break; }
                        }
                        lostB = __syncthreads_or(mylost);
                        if (lostB) break;
#pragma unroll
                        for (int j = 0; j < MG; j++) {
                            const u64 hi = __shfl_down(bv[j], 1);           // lane pairs: even lane = low half, odd lane = high half
                            const u64 word = (bv[j] & 0xFFFFFFFFull) | (hi << 32);
                            const bool mineW = hb[j] && !(lane & 1);
                            if (mineW) {
                                const int gi = (rd * MG + j) * MK_THREADS + tid;
                                S.bm[S.list[bk[j]] * MK_MAXW + ((gi & 31) >> 1)] = word;
                            }
                            // hz ("the column may hold a zero") of the two columns this wave instruction covers: exact again
                            const u64 nzb = __ballot(mineW && word != 0);
                            // ... and whether one of its zeros sits in an uncovered row (the event loop's candidate mask)
                            const u64 nzl = __ballot(mineW && (word & ~S.covR[((((rd * MG + j) * MK_THREADS + tid) & 31) >> 1) & (MK_MAXW - 1)]) != 0);
                            if ((lane & 31) == 0 && hb[j]) {
                                const int c = S.list[bk[j]];
                                unsigned int* wp = reinterpret_cast<unsigned int*>(&S.hz[c >> 6]) + ((c & 63) >> 5);
                                unsigned int* wl = reinterpret_cast<unsigned int*>(&S.hzl[c >> 6]) + ((c & 63) >> 5);
                                const unsigned int bit = 1u << (c & 31);
                                if ((lane ? nzb >> 32 : nzb & 0xFFFFFFFFull) != 0) atomicOr(wp, bit); else atomicAnd(wp, ~bit);
                                if ((lane ? nzl >> 32 : nzl & 0xFFFFFFFFull) != 0) atomicOr(wl, bit); else atomicAnd(wl, ~bit);
                            }
                        }
                        if (wcov && nrow > 0) {
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const u64 lo = readlane64(cvv, 2 * q), hi2 = readlane64(cvv, 2 * q + 1);
                                if (q < nrow && ccol) {
                                    const int rr = S.clist[rd * 4 + q];
                                    const bool z = (((lane < 32) ? lo : hi2) >> (lane & 31)) & 1;
                                    u64& wd = S.bm[tid * MK_MAXW + (rr >> 6)];
                                    wd = z ? (wd | (1ull << (rr & 63))) : (wd & ~(1ull << (rr & 63)));
                                }
                            }
                        }
                    }
                };
                merge_rounds(max((ncu + 32 * MG - 1) / (32 * MG), 1), true, false);
                // (covered rows) x (covered columns): every helper that owns covered columns reports how many of its rows still hold a zero
                // there after + h.  Normally none does: those bits are cleared right here, no granule is read
                if (!lostB && ncr > 0) {
                    if (uwave == 0) {
                        const bool need = lane < nhelp && lane < wordsC;
                        const u64 cw = (lane < MK_MAXW) ? S.covC[lane] : 0;
                        const u64 vw = (lane < wordsC) ? ((lane == wordsC - 1 && (nC & 63)) ? ((1ull << (nC & 63)) - 1) : ~0ull) : 0;
                        const bool want = need && (cw & vw) != 0;
                        u64 sv = 0; int spins = 0; bool lost = false;
                        for (;;) {
                            if (want) sv = ctl_ld(ctl + CTL_COVSUM + lane * MK_PARTIAL_STRIDE);
                            if (!__ballot(want && (unsigned)(sv >> 32) != tag)) break;
                            if (++spins > MK_SPIN_LIMIT) { lost = true; break; }
                            __builtin_amdgcn_s_sleep(1);
                        }
                        const bool exc = __ballot(want && (unsigned)sv != 0) != 0;
                        if (lane == 0) S.flag[4] = lost ? 2 : (exc ? 1 : 0);
                    }
                    __syncthreads();
                    const int cs = S.flag[4];
                    if (cs == 2) lostB = true;
                    else if (cs == 1) merge_rounds((ncr + 3) / 4, false, true);
                    else {
                        // thread -> (column, row word) so that a wave touches consecutive LDS words
                        const int wq = tid & (MK_MAXW - 1);
                        const u64 crw = S.covR[wq];
                        if (crw) for (int c = tid >> 4; c < nC; c += MK_THREADS / MK_MAXW) if ((S.covC[c >> 6] >> (c & 63)) & 1) S.bm[c * MK_MAXW + wq] &= ~crw;
                    }
                }
                if (lostB) { if (tid == 0) { stat[15] = 2; S.flag[7] = 1; } }
                else if (tid < MK_HELPERS) ctl_stx(ctl + CTL_PARTIAL + ((myseq & 1) * MK_HELPERS + tid) * MK_PARTIAL_STRIDE, MK_HSENT, fast);   // every helper that needed the partial minima has used them: re-arm for step + 2
                t_h2 += wall_clock64() - tq2;
            }
            __syncthreads();
            if (S.flag[7]) break;
            const long long tq3 = wall_clock64();
            if (wave == 0) { hz = (lane < MK_MAXW) ? S.hz[lane] : 0; hzl = (lane < MK_MAXW) ? S.hzl[lane] : 0; }   // updated column by column during the merge (every uncovered column is in it)
            t_h3 += wall_clock64() - tq3;
        } else {
            const int ncu = S.flag[1];
            if (n_s5 == 1) ncu0 = ncu;
            const int r = tid;
            const u64 cw = S.covR[wave];                               // wave == 64-row word of this thread's row
            const bool mine = r < nR && !((cw >> lane) & 1);
            double h = DBL_MAX;
            double v[16];
            const size_t rclamp = (size_t)min(r, nR - 1);
            const long long tp0 = wall_clock64();
            // pass 1: h = min over uncovered rows x uncovered columns; 16 columns per round, all loads in flight at once
            // (unconditional loads with clamped indices: a per-element guard would serialise them)
            for (int k0 = 0; k0 < ncu; k0 += 16) {
#pragma unroll
                for (int k = 0; k < 16; k++) v[k] = d[rclamp + (size_t)nR * (unsigned)__builtin_amdgcn_readfirstlane((int)S.list[min(k0 + k, ncu - 1)])];
#pragma unroll
                for (int k = 0; k < 16; k++) if (mine && k0 + k < ncu && v[k] < h) h = v[k];
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_down(h, off); if (o < h) h = o; }
            const long long tp1 = wall_clock64(); t_h0 += tp1 - tp0;
            if (lane == 0) S.red[wave] = h;
            __syncthreads();
            h = S.red[0];
#pragma unroll
            for (int w = 1; w < MK_THREADS / 64; w++) { const double o = S.red[w]; if (o < h) h = o; }
            const long long tp2 = wall_clock64(); t_h1 += tp2 - tp1;
            // (a) uncovered rows x uncovered columns: d -= h; rebuild the uncovered-row bits of the bitmap word
            for (int k0 = 0; k0 < ncu; k0 += 16) {
                if (ncu > 16) {
#pragma unroll
                    for (int k = 0; k < 16; k++) v[k] = d[rclamp + (size_t)nR * (unsigned)__builtin_amdgcn_readfirstlane((int)S.list[min(k0 + k, ncu - 1)])];
                }
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    if (k0 + k < ncu) {
                        const int c = __builtin_amdgcn_readfirstlane((int)S.list[k0 + k]);
                        bool z = false;
                        if (mine) { const double nv = v[k] - h; d[(size_t)r + (size_t)nR * c] = nv; z = fabs(nv) < DBL_EPSILON; }
                        const u64 bal = __ballot(z);
                        if (lane == 0 && wave < wordsR) S.bm[c * MK_MAXW + wave] = (S.bm[c * MK_MAXW + wave] & cw) | (bal & ~cw);
                    }
                }
            }
            t_h2 += wall_clock64() - tp2;
            // (b) covered rows, every column: d += h, then -= h where the column is uncovered (order of :355-364)
            bool anyCov = false;
            for (int w = 0; w < wordsR; w++) anyCov |= S.covR[w] != 0;
            const long long tb0 = wall_clock64();
            if (anyCov) {
                if (tid == 0) n_cov5++;
                __syncthreads();
                for (int w = 0; w < wordsR; w++) {
                    u64 rows = S.covR[w];
                    while (rows) {
                        const int rr = w * 64 + (__ffsll((long long)rows) - 1); rows &= rows - 1;
                        for (int c = tid; c < nC; c += MK_THREADS) {
                            double x = d[(size_t)rr + (size_t)nR * c] + h;
                            if (!((S.covC[c >> 6] >> (c & 63)) & 1)) x -= h;
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
            t_h3 += wall_clock64() - tb0;
            {   // rebuild hz for the uncovered columns (their entries just changed)
                const bool unc = tid < nC && !((S.covC[wave] >> lane) & 1);
                bool has = false, hasl = false;
                if (unc) for (int w = 0; w < wordsR; w++) { const u64 bw = S.bm[tid * MK_MAXW + w]; has |= bw != 0; hasl |= (bw & ~S.covR[w]) != 0; }
                const u64 bal = __ballot(has), ball = __ballot(hasl);
                if (lane == 0) { S.hz[wave] = (S.hz[wave] & S.covC[wave]) | bal; S.hzl[wave] = ball; }
            }
            __syncthreads();
            if (wave == 0) { hz = (lane < MK_MAXW) ? S.hz[lane] : 0; hzl = (lane < MK_MAXW) ? S.hzl[lane] : 0; }
        }
        t_s5 += wall_clock64() - t_b;
        if (++guard > 4 * MK_MAXN * MK_MAXN) break;                    // cannot happen for finite costs
    }
    __syncthreads();
    if (HELP && nhelp > 0 && uwave == 0) {
        if (lane == 0) ctl_st(ctl + CTL_EPOCH, (u64)(epoch + myseq + 1));   // + 1: the tags of a launch without any step 5 (XCCTAB, MODE) are spent too
        ctl_st(ctl + CTL_COV + lane, (u64)MK_TAG_EXIT << 32);
    }
    if (tid == 0) {
        stat[0] = certified ? -1 : n_s4; stat[1] = n_s5; stat[2] = n_sw; stat[3] = n_cov5; stat[14] = ncu0; 
        // step-5 split (thread 0, 100 MHz ticks): helpers: publish / wait minimum / wait update / merge;  one workgroup: pass 1 / reduce / (a) / (b)
        stat[4] = (int)t_h0; stat[5] = (int)t_h1; stat[6] = (int)t_h2; stat[7] = (int)t_h3;
        stat[8] = (int)(t_init - t_begin); stat[9] = (int)t_s3; stat[11] = (int)t_s5; stat[12] = (int)(wall_clock64() - t_begin);
        stat[13] = (int)((clock64() - c_begin) * 100 / max((long long)1, wall_clock64() - t_begin));   // shader MHz during this launch
    }
    // buildassignmentvector (:161-176) + computeassignmentcost (:179-189)
    for (int r = tid; r < nR; r += MK_THREADS) a.ws.assignment[r] = S.starColOfRow[r];
    if (want_cost & 1) {
        double* vals = reinterpret_cast<double*>(S.bm);                // bitmap no longer needed
        __syncthreads();
        for (int r = tid; r < nR; r += MK_THREADS) { const int c = S.starColOfRow[r]; vals[r] = (c >= 0) ? elem_cost(a, r, c, nR, rowsTrk) : 0.0; }
        __syncthreads();
        if (tid == 0) { double cst = 0.0; for (int r = 0; r < nR; r++) if (S.starColOfRow[r] >= 0) cst += vals[r]; *a.ws.cost = cst; }
    }
    if (patch) { __syncthreads(); prov_apply(a, life, a.ws.assignment, nR, reinterpret_cast<bbox_t*>(a.cost_only), true); return; }
    // device-resident loop: the lifecycle step (td.cpp:472-644) runs here instead of in a launch of its own
    if (life.enabled) {
        __syncthreads();                                               // the assignment vector is complete (same workgroup wrote it)
        // a helper hand-off that timed out leaves a partial starring: never commit it to the tracker state -- the frame is
        // dropped (live list and models untouched) and the context latches a device error that every read-back reports
        if (S.flag[7]) { if (tid == 0) { life.S.err[4] = stat[15]; *life.S.upd_count = 0; } }   // the update launch queued behind this kernel becomes a no-op (its lists are last frame's)
        else dl_lifecycle_body(life.S, life.kp, life.kal, life.trk_pred, life.dets, life.nD, a.ws.assignment, reinterpret_cast<int*>(S.bm) + 4096);
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

// Dense-solver arming (host side, scheduling only -- never a result).  h: [0] device-written hint bits (bit 1: a recent launch needed the
// dense solver), [1] hold armed by such a launch, [2] last detection count, [3] change-armed launches in a row that did not need it,
// [4] launches a count change is ignored for, [5] back-off level, [6] hold armed by a count change.  Returns whether this launch submits
// the dense solver's kernels.
bool dense_arming_step(volatile int* h, int nD)
{
    const bool changed = h[2] > 0 && nD > 0 && h[2] != nD;
    if (h[0] & 2) { h[1] = 512; h[3] = 0; h[4] = 0; h[5] = 0; }   // a recent launch needed it: held for 512 launches
    else {
        if (h[1] > 0) h[1] = h[1] - 1;
        if (h[4] > 0) h[4] = h[4] - 1;
        else if (changed && h[6] < 8) h[6] = 8;                  // a count change arms a short hold ...
        if (h[6] > 0) {
            h[6] = h[6] - 1;
            if (h[3] + 1 >= 32) { h[5] = h[5] < 6 ? h[5] + 1 : 6; h[4] = 64 << h[5]; h[3] = 0; h[6] = 0; }   // ... with exponential back-off while nobody needs it
            else h[3] = h[3] + 1;
        }
    }
    if (nD > 0) h[2] = nD;
    return h[1] > 0 || h[6] > 0;
}

hipError_t launch_lap_front(const AssocArgs& a, int gR, int gC, hipStream_t s, hipEvent_t ev_mid, const LifeArgs& life, bool two_block, int mk_batch, const AssocEmu* emu);   // lap_kernels.hip
hipError_t launch_lap_dense(const AssocArgs& a, int gR, int gC, hipStream_t s);                     // lap_dense.hip
hipError_t launch_mk_sparse(const AssocArgs& a, int gR, int gC, hipStream_t s, const LifeArgs& life);   // mk_sparse.hip

hipError_t launch_assoc(const AssocWs& ws, const bbox_t* trk, const int* nT_dev, int nT, const bbox_t* det, int nD,
                        const double* user_dist, int nR, int nC, int want_cost, hipStream_t s, hipEvent_t ev_mid, const LifeArgs* life_in, const AssocEmu* emu, unsigned* seq_out)
{
    LifeArgs life{}; if (life_in) life = *life_in;
    AssocArgs a{};
    // every launch chain gets a sequence number (>= 1; host counter beside the scheduling hints): the protocol words of its kernels carry it
    {
        static unsigned fallback = 0;
        unsigned* ctr = ws.dense_hint ? reinterpret_cast<unsigned*>(ws.dense_hint) + 8 : &fallback;
        *ctr = (*ctr + 1) & 0x3FFFFFFFu; if (*ctr == 0) *ctr = 1;
        a.seq = *ctr;
        if (seq_out) *seq_out = a.seq;
    }
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
    const int lines = maxR > maxC ? maxR : maxC;
    // Assignment fast path (lap_kernels.hip): exact sparse solver + uniqueness certificate in front of the emulation; rows <=
    // columns only (the only shape td.cpp:462-469 produces).  MOT_LAP_FAST=0 switches it off, MOT_LAP_MIN sets the smallest
    // problem (lines) it is used for -- below that the emulation of a whole frame costs less than the three extra launches.
    const mot_impl::EnvSwitches& E = mot_impl::env();                   // the MOT_* switches, read once per process (mot_env.h)
    const int lap_min = E.lap_min < 0 ? MK_LAP_MIN_LINES : E.lap_min;
    const bool lap = ws.lap.ccol && maxR > 0 && maxR <= maxC && lines >= lap_min;
    // scheduling hint left by the final kernel of earlier launches in pinned host memory, read without synchronisation (stale by a
    // frame or two: it only picks grids, never results): the stream keeps needing the dense emulation / the dense solver
    bool hinted_now = false, prep_in_kernel = false;                  // (evaluated behind the countdown update below)
    bool stream_emu = false;
    // small problems: the Munkres workgroup computes cost, minima and bitmaps itself (mk_fused_cost)
    const bool fused = lines <= MK_FUSE_LINES && !lap;
    hipError_t e = hipSuccess;
    if (!fused && maxR > 0 && maxC > 0) {
        // nT_dev: the true nT is <= nT (the host-side upper bound); tiles outside exit early
        const int gR = ((nT_dev ? (nD > nT ? nD : nT) : maxR) + 63) / 64, gC = ((nT_dev ? (nD > nT ? nD : nT) : maxC) + 63) / 64;
        if (lap) {
            // Dense solver (lap_dense.hip) for frames whose far matches defeat the sparse one (detector misses + false positives): its
            // three launches return at once when the sparse solver succeeded, but they are not even submitted unless one of the last
            // 512 launches needed them (the final kernel's hint in pinned host memory, read without synchronisation) or the number
            // of detections has just changed.  MOT_LAP_DENSE=0 never, =1 always.
            const int dense_mode = E.dense_mode;
            bool want_dense = dense_mode == 1;
            if (dense_mode == 2 && ws.dense_hint) {
                volatile int* h = ws.dense_hint;                       // [0] device-written hint bits, [1] host-side countdown, [2] last nD,
                                                                       // [3] change-armed launches in a row that did not need it, [4] launches a count change is ignored for, [5] back-off level, [6] hold armed by a count change
                // a detection count that changes from frame to frame is what detector noise looks like from the host: arm the dense
                // solver at once, so that the first noisy frame of a stream does not have to go through the emulation -- but only for a
                // few launches, and with an exponential back-off when the device keeps reporting that nobody needed it (objects entering
                // and leaving change the count on a clean stream too; round-2 advisor finding).  A launch that DID need it arms 512.
                want_dense = dense_arming_step(h, nD);
            }
            // Box costs without the dense solver in between: the sparse emulation rides in the solver's launch as its second workgroup
            // (speculative start, lap_kernels.hip); MOT_LAP_TWO_BLOCK=0 keeps the separate launch
            const int two_block_on = E.two_block, mk_batch_on = E.mk_batch;   // MOT_LAP_TWO_BLOCK / MOT_MK_BATCH, MOT_MK_LAZY, MOT_MK_TIMING
            const bool two_block = two_block_on && !a.user && !want_dense;
            // device loop with an emulation stream: the emulation leaves the solver's launch for a kernel of its own there (provisional commits)
            stream_emu = two_block && emu && emu->stream && life.enabled;
            a.stream_emu = stream_emu ? 1 : 0; a.det_copy = stream_emu ? emu->det_copy : nullptr;
            if (!stream_emu) life.prov.enabled = 0;
            e = launch_lap_front(a, gR, gC, s, ev_mid, life, two_block, mk_batch_on, stream_emu ? emu : nullptr); if (e != hipSuccess) return e;
            ev_mid = nullptr;
            if (want_dense) { e = launch_lap_dense(a, gR, gC, s); if (e != hipSuccess) return e; }
            if (!two_block) { e = launch_mk_sparse(a, gR, gC, s, life); if (e != hipSuccess) return e; }
            // working matrix + bitmaps for the dense emulation: chip-wide (lazy: every workgroup checks the verdict and leaves) for caller
            // matrices and for streams whose recent frames needed it; otherwise the final kernel prepares them itself if it has to
            hinted_now = ws.dense_hint && ((*reinterpret_cast<volatile int*>(ws.dense_hint) & 1) != 0 || reinterpret_cast<volatile int*>(ws.dense_hint)[1] > 0 || reinterpret_cast<volatile int*>(ws.dense_hint)[6] > 0);
            const int helpers_forced = E.helpers == 1 ? 1 : 0;
            prep_in_kernel = !a.user && !hinted_now && !helpers_forced;
            if (!prep_in_kernel) hipLaunchKernelGGL(assoc_sub_kernel, dim3(gR, gC), dim3(256), 0, s, a, 1);
        }
        else {
            hipLaunchKernelGGL(assoc_min_kernel, dim3(gR, gC), dim3(256), 0, s, a);
            hipLaunchKernelGGL(assoc_sub_kernel, dim3(gR, gC), dim3(256), 0, s, a, 0);
        }
    }
    if (ev_mid) { e = hipEventRecord(ev_mid, s); if (e != hipSuccess) return e; }   // cost kernels submitted, the Munkres kernel comes next
    e = mot_impl::func_lds_once(reinterpret_cast<const void*>(munkres_kernel<false>), (int)sizeof(MkShared)); if (e != hipSuccess) return e;
    e = mot_impl::func_lds_once(reinterpret_cast<const void*>(munkres_kernel<true>), (int)sizeof(MkShared)); if (e != hipSuccess) return e;
    // Step-5 helper workgroups (16 more CUs stream the matrix; two cross-CU hand-offs per step 5) pay off only when
    // step 5 moves a lot of data: dense hard problems beyond ~512 lines.  MOT_MUNKRES_HELPERS=1 forces them on for
    // every problem above 256 lines, =0 off; default: above MK_HELP_MIN lines.
    const int helpers = E.helpers, force_cov = E.force_cov;             // MOT_MUNKRES_HELPERS=2: forced on AND the controller reads the per-row COVBITS granules (test hook)
    // behind the fast path the dense emulation is the rare last resort: one workgroup, not the 1 + 128 workgroup helper grid
    // (whose launch alone costs more than the common case's whole final kernel)
    // ... unless the stream keeps needing it (detector misses + false positives force far matches no tier can certify): the final
    // kernel leaves a hint in pinned host memory, read here without synchronisation (stale by a frame or two: it only picks the grid)
    const bool hinted = hinted_now;                                    // (a noisy stream: the countdown above is armed)
    const bool big = !prep_in_kernel && ws.ctl && (helpers == 1 ? lines > 256 : (helpers == 2 && lines > MK_HELP_MIN && (!lap || hinted)));
    mot_impl::lds_poison(s);                                           // (debug) MOT_LDS_POISON
    const int lap_mode = (lap ? 1 : 0) | (stream_emu ? 2 : 0);
    if (!stream_emu) life.prov.enabled = 0;
    if (big) hipLaunchKernelGGL(munkres_kernel<true>, dim3(1 + MK_XCDS * MK_HELPERS), dim3(MK_THREADS), sizeof(MkShared), s, a, (want_cost & 1) | (force_cov << 3), life, lap_mode);
    else hipLaunchKernelGGL(munkres_kernel<false>, dim3(1), dim3(MK_THREADS), sizeof(MkShared), s, a, (want_cost & 1) | (fused ? 2 : 0) | (prep_in_kernel ? 4 : 0), life, lap_mode);
    return hipGetLastError();
}

// the patch step of chain `seq` (see prov_apply): the final kernel's single-workgroup instantiation in patch mode -- it returns at once unless that
// frame was committed provisionally, applies the sparse emulation's answer, and holds the dense emulation for the case that one refused
hipError_t launch_prov_patch(const AssocWs& ws, const LifeArgs& life, const bbox_t* trk, const bbox_t* det, int nD, unsigned seq, bbox_t* pred_cur, hipStream_t s)
{
    AssocArgs a{};
    a.trk = trk; a.det = det; a.nT_dev = &life.prov.rec->nT; a.nT = life.S.cap; a.nD = nD;    // ProvRec::nT: the live count the frame was associated with
    a.ws = ws; a.linemin = ws.linemin; a.dims = ws.status + 4;
    a.cost_only = reinterpret_cast<double*>(pred_cur);                  // (patch mode: the slot carries the predicted boxes of the launch in between)
    a.seq = seq;
    hipError_t e = mot_impl::func_lds_once(reinterpret_cast<const void*>(munkres_kernel<false, true>), (int)sizeof(MkShared)); if (e != hipSuccess) return e;
    mot_impl::lds_poison(s);
    hipLaunchKernelGGL((munkres_kernel<false, true>), dim3(1), dim3(MK_THREADS), sizeof(MkShared), s, a, 4 | (mot_impl::env().prov == 2 ? 8 : 0), life, 4);   // (bit 3: test hook MOT_PROV=2)
    return hipGetLastError();
}

// test hook (pure host code, no device needed): one step of the dense-solver arming state machine on a caller-owned int[16]
extern "C" int mot_debug_dense_arming(int* h16, int nD) { return (h16 && dense_arming_step(h16, nD)) ? 1 : 0; }
