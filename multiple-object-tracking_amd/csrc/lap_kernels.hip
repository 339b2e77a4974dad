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
//   lap_verify_kernel   the dense pass: reduced
//                       cost r = c - u_i - v_j of EVERY entry must be >= -tol, prices <= 0 and exactly 0 on free
//                       columns (dual feasibility + complementary slackness => optimal), and every entry with r < eps
//                       that is not matched is recorded as an edge "row i could take the column of row i'".
// The Munkres kernel then runs lap_certify (lap_certify.h): the near-tight digraph is acyclic <=> every other
// assignment costs at least eps more => the reference returns exactly this one.  Otherwise (a tie, or any doubt) the
// order-exact emulation runs as before.  lap_model.c (CPU model, test infrastructure) is the CPU model of this file; tests/test_lap_model.py
// fuzzes "certified => equal to the reference" on CPU, tests/test_gpu_parity.py on the device.
#include <hip/hip_ext.h>
#include "assoc_common.h"
#include "lap_certify.h"
#include "lap_grid.h"
#include "dl_lifecycle.h"
#include "mk_sparse_body.h"
#include "mot_env.h"

using namespace assoc;

namespace {

// ---- stage 1 ----------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) lap_rowscan_kernel(AssocArgs a)
{
    int nR, nC; bool rowsTrk; resolve_dims(a, nR, nC, rowsTrk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0) {                        // the housekeeping assoc_min_kernel does
        if (a.ws.ctl) { for (int i = 0; i < 2 * MK_HELPERS; i++) a.ws.ctl[CTL_PARTIAL + i * MK_PARTIAL_STRIDE] = MK_HSENT; for (int i = 0; i < 64; i++) a.ws.ctl[CTL_COV + i] = 0; }
        a.dims[0] = nR; a.dims[1] = nC; a.dims[2] = rowsTrk; a.dims[3] = nR <= nC;
        // first kernel of the chain: the per-frame protocol words start from zero whatever the previous launch left (its final kernel
        // re-arms them, but not on its early-out for an empty side)
        // (the verdict word as THIS chain's "pending": an untagged 0 reads as a newer chain's word to chain 0x3FFFFFFF -- tests/test_emu_protocol_model.py)
        a.ws.lap.hdr[LAP_H_VERDICT] = tagged_word(a.seq, 0); a.ws.lap.hdr[LAP_H_DONE] = 0; a.ws.lap.hdr[LAP_H_CERT] = 0;
    }
    if (a.det_copy && !a.user) {                                       // stream emulation: its kernel reads a private copy of the detection list (24-byte boxes as 4-byte words)
        const int g = blockIdx.x * 256 + threadIdx.x;
        if (g < a.nD * 6) reinterpret_cast<int*>(a.det_copy)[g] = reinterpret_cast<const int*>(a.det)[g];
    }
    const int r = blockIdx.x * 4 + wave;
    const LapWs& L = a.ws.lap;
    // ---- box costs: the workgroup's four rows share one pass over the column boxes -- centroid and class of every column go to
    // LDS once (coalesced 24-byte reads by all 256 threads; every wavefront used to read all nC boxes itself) ----
    __shared__ unsigned s_cxy[MK_MAXN];                                // (cx & 0xFFFF) | cy << 16: |centroid| <= 1400 when the box is "small"
    __shared__ int s_ty[MK_MAXN];
    bool small = true;
    if (!a.user) {
        for (int j = threadIdx.x; j < nC; j += 256) {
            const bbox_t cb = rowsTrk ? a.det[j] : a.trk[j];
            small &= box_small(cb);
            const int cx = (cb.l + cb.r) >> 1, cy = (cb.t + cb.b) >> 1;
            s_cxy[j] = ((unsigned)cx & 0xFFFFu) | ((unsigned)cy << 16);
            s_ty[j] = cb.type;
        }
        small = __syncthreads_and(small) != 0;
    }
    if (r >= nR || nC <= 0) return;                                   // wave-uniform (behind the workgroup's only barrier)
    bbox_t rb = {};
    if (!a.user) rb = rowsTrk ? a.trk[r] : a.det[r];
    // Order by an integer key and evaluate the float64 cost only for the entries that are kept.  Key = squared centroid distance << 10 |
    // column for same-class entries closer than 2048 px, "none" otherwise: ONE wave minimum per kept entry yields distance and column
    // (equal distances: the lowest column), and a row that does not find LAP_K such entries among its columns takes the general form
    // below.  Exact: same-class costs are < 1.0 <= every cross-class cost, and inside a class the cost grows strictly with the squared
    // distance.
    if (!a.user) {
        unsigned ik[MK_MAXN / 64];
        small &= box_small(rb);
        const int rcx = (rb.l + rb.r) >> 1, rcy = (rb.t + rb.b) >> 1;
        unsigned mx0 = 0, mx1 = 0; bool any1 = false;
#pragma unroll
        for (int t = 0; t < MK_MAXN / 64; t++) {
            const int j = t * 64 + lane;
            ik[t] = 0xFFFFFFFFu;
            if (j < nC) {
                const unsigned u = s_cxy[j];
                const int dx = rcx - (int)(short)(u & 0xFFFFu), dy = rcy - ((int)u >> 16);
                const unsigned d2 = (unsigned)(dx * dx + dy * dy);
                const bool pen = s_ty[j] != rb.type;
                if (!pen && d2 < (1u << 22)) ik[t] = (d2 << 10) | (unsigned)j;
                if (pen) { any1 = true; if (d2 > mx1) mx1 = d2; } else if (d2 > mx0) mx0 = d2;
            }
        }
        if (small) {
            unsigned sel = 0xFFFFFFFFu;                                // lane k keeps the k-th smallest
            for (int k = 0; k < LAP_K; k++) {
                unsigned lk = 0xFFFFFFFFu;
#pragma unroll
                for (int t = 0; t < MK_MAXN / 64; t++) lk = ik[t] < lk ? ik[t] : lk;
                const unsigned wm = wave_min_u32_dpp(lk);
                if (lane == k) sel = wm;
                if (wm != 0xFFFFFFFFu) {
#pragma unroll
                    for (int t = 0; t < MK_MAXN / 64; t++) ik[t] = ik[t] == wm ? 0xFFFFFFFFu : ik[t];   // keys are unique
                }
            }
            // lane k < LAP_K: a kept entry must exist unless the row has fewer than k + 1 columns, and lie closer than 1280 px
            const bool near = lane >= LAP_K || (sel != 0xFFFFFFFFu ? (sel >> 10) < (unsigned)MOT_FRAME_W * MOT_FRAME_W : lane >= nC);
            if (!__ballot(!near)) {
                if (lane < LAP_K) {
                    const bool has = sel != 0xFFFFFFFFu;
                    const double cst = has ? cost_of_d2((int)(sel >> 10), false) : DBL_MAX;
                    L.ccol[(size_t)r * LAP_K + lane] = has ? (unsigned short)(sel & 1023u) : (unsigned short)0xFFFF;
                    L.ccost[(size_t)r * LAP_K + lane] = cst;
                    if (lane == 0) a.linemin[r] = dkey(cst);           // hungarian.cpp:69-81
                }
                // largest cost of the row: the farthest entry of each class
                mx0 = ~wave_min_u32_dpp(~mx0); mx1 = ~wave_min_u32_dpp(~mx1);
                const bool w1 = __ballot(any1) != 0;
                if (lane == 0) { double m = cost_of_d2((int)mx0, false); if (w1) { const double m1 = cost_of_d2((int)mx1, true); if (m1 > m) m = m1; } L.u[r] = m; }   // row maximum (L.u is free until the solver writes the duals)
                return;
            }
        }
    }
    // ---- general form: caller-supplied matrix, or boxes the integer ordering does not cover ----
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
        const u64 wm = wave_min_u64_dpp(lk);
        const unsigned mycol = (lk == wm && lk != ~0ull) ? (unsigned)(lt * 64 + lane) : 0xFFFFFFFFu;
        const unsigned wc = wave_min_u32_dpp(mycol);                  // equal cost: the lowest column
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
    mx = wave_max_u64_dpp(mx);
    if (lane == 0) L.u[r] = dunkey(mx);                                // row maximum: reduced by the solver (no same-address atomics)
    if (__ballot(bad) && lane == 0) atomicOr(&L.hdr[LAP_H_BAD], 1);
}

// ---- stage 2 ----------------------------------------------------------------------------------------------------
// One search = one WAVEFRONT: the touched columns of the search live in the registers of lanes 0..LAP_TS-1 (column,
// tentative distance, predecessor), the minimum is a wave reduction, and relaxing a candidate is a ballot over the touched
// columns -- no LDS traffic besides the candidate list of the row being scanned.  Sixteen searches run at a time; every
// wavefront works through its share of the round's free rows on the SAME snapshot (nothing is written to prices or matching
// until the round's barrier), leaves a record (scanned columns with their price decrements, the augmenting path) in LDS,
// and takes locks; after the barrier the searches that hold all their locks commit.
#define LAP_SR 64                          /* searches per round (4 per wavefront) */
struct LapRec {
    double dec[LAP_TS];                  // scanned slots: Delta - d (what the column's price drops by)
    unsigned short col[LAP_TS];          // touched columns
    unsigned short prow[LAP_TS], pcol[LAP_TS];   // augmenting path: row prow[n] takes column pcol[n] ...
    unsigned char pk[LAP_TS];            // ... which is its pk[n]-th candidate
    unsigned char flag[LAP_TS];          // 1 scanned, 2 end column
    int nt, plen, ok, pad;
};
struct LapShared {
    double cc[MK_MAXN * LAP_K];          // candidate costs, [row][k]
    double v[MK_MAXN];                   // column prices
    double mcost[MK_MAXN];               // cost of the row's matched edge
    double red[MK_THREADS / 64];
    LapRec rec[LAP_SR];
    unsigned lock[MK_MAXN];
    unsigned short cj[MK_MAXN * LAP_K];  // candidate columns
    short rowOfCol[MK_MAXN], colOfRow[MK_MAXN];
    unsigned short flist[MK_MAXN];
    int wave_tot[MK_THREADS / 64];
    int flag[8];
    double margins[2];                   // eps, tol for the fused dual check
};
static_assert(sizeof(LapShared) <= MOT_LDS_LIMIT, "lap_solve_kernel LDS");
// fused tail (box costs): the candidate costs `cc` are dead once the solver is done; their 64 KB hold the certificate's scratch
// (edge list + peel flags), the lifecycle step's scratch and a uniform grid over the column boxes
struct LapFused {
    unsigned certify[LAP_EDGES + (2 * (MK_MAXN + 64) + 3) / 4];
    int life[DL_LIFE_SCRATCH_INTS];
    ColGrid grid;
    int ne, viol;
};
static_assert(sizeof(LapFused) <= sizeof(double) * MK_MAXN * LAP_K, "fused tail overlays the candidate costs");
static_assert(LAP_TS <= 32 && LAP_K <= 16 && (LAP_K & (LAP_K - 1)) == 0, "search state: one touched column per lane, candidates in the low lanes");

__device__ __forceinline__ double readlane_f64(double x, int src)   // src wave-uniform
{
    const u64 b = (u64)__double_as_longlong(x);
    return __longlong_as_double((long long)readlane64(b, src));
}

// ---- provisional commit (round 6; mot_dev.h: ProvRec) ----------------------------------------------------------------------------
// Called (workgroup-uniform) when lap_certify found a cycle among the near-tight edges.  ed[0, ne) is the edge list in LDS, alive[] what the
// certificate's forward peel left (nodes with a path INTO a cycle).  Peeling from the other side as well (a node stays only if an alive node
// has an edge into it) until nothing changes leaves the nodes that lie ON a cycle or between two cycles.  If every one of them has exactly one
// remaining target and is that row's only target in turn -- pairs A <-> B -- those two-row cycles are the only cycles: every assignment within
// eps of the optimum is M with the columns of some of the pairs exchanged (lap_certify.h: any other assignment differs from M along cycles
// of near-tight edges or costs at least eps - n tol more), so the reference returns one of those.  The rows are matched in all of them, so the
// lifecycle step (td.cpp:472-644) sees the same counters, deaths and spawns -- only the box / spectrum the tracks of a pair adopt differs.
// M is committed here; those tracks are cloned into the pool's shadow slots, which adopt the OTHER detection, and appended to the predict
// list: the next predict launch computes both alternatives and prov_apply (assoc_kernels.hip) keeps, pair by pair, the one the order-exact
// emulation names.  Refused (-> the frame waits for the emulation, verdict 2) when the core is anything else or has more than
// MOT_PROV_PAIRS pairs, when a free column is part of it, or when the emulation's kernel is not running yet (the patch step waits for it by
// polling: that is only safe for a kernel that is already resident).
__device__ bool lap_try_provisional(const AssocArgs& a, const LifeArgs& life, LapShared& S, LapFused& F, int nR, int nC, bool rowsTrk, int ne)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const LapWs& L = a.ws.lap;
    unsigned* ed = F.certify;
    unsigned char* alive = reinterpret_cast<unsigned char*>(ed + LAP_EDGES);
    unsigned char* mark = alive + MK_MAXN + 64;
    int* tmin = reinterpret_cast<int*>(ed + 2048);                     // (ne <= 1024: the rest of the edge area is free) smallest / largest alive target of a row
    int* tmax = reinterpret_cast<int*>(ed + 4096);
    int* prow = reinterpret_cast<int*>(ed + 6144);                     // [2 * MOT_PROV_PAIRS]: the pairs, (lower row, its partner)
    if (tid < 64) {
        for (int guard = 0; guard < 4 * MK_MAXN; guard++) {
            bool ch = false;
            // (every alive node is the source of some edge -- lap_certify starts from the sources -- so walking the edge list reaches all of them)
            for (int e = lane; e < ne; e += 64) mark[ed[e] >> 16] = 0;
            for (int e = lane; e < ne; e += 64) { const unsigned x = ed[e]; if (alive[x >> 16] && alive[x & 0xFFFF]) mark[x & 0xFFFF] = 1; }   // has an edge coming in
            for (int e = lane; e < ne; e += 64) { const unsigned s = ed[e] >> 16; if (alive[s] && !mark[s]) { alive[s] = 0; ch = true; } }
            for (int e = lane; e < ne; e += 64) mark[ed[e] >> 16] = 0;
            for (int e = lane; e < ne; e += 64) { const unsigned x = ed[e]; if (alive[x >> 16] && alive[x & 0xFFFF]) mark[x >> 16] = 1; }      // has an edge going out
            for (int e = lane; e < ne; e += 64) { const unsigned s = ed[e] >> 16; if (alive[s] && !mark[s]) { alive[s] = 0; ch = true; } }
            if (!__ballot(ch)) break;
        }
        // what is left lies on a cycle or between two cycles.  Disjoint two-row cycles <=> every remaining row has exactly ONE remaining
        // target, and is that row's only target in turn (then no other cycle exists among them: out-degree one)
        for (int r = lane; r <= nR; r += 64) { tmin[r] = 0x7FFFFFFF; tmax[r] = -1; }
        for (int e = lane; e < ne; e += 64) { const unsigned x = ed[e], s = x >> 16, d = x & 0xFFFF; if (alive[s] && alive[d]) { atomicMin(&tmin[s], (int)d); atomicMax(&tmax[s], (int)d); } }
        int np = 0, code = 0, ncore = 0;                               // code: 0 fine so far, 2 a free column is part of the core, 3 too many pairs, 4 not disjoint two-row cycles
        if (alive[nR]) code = 2;
        for (int r0 = 0; r0 < nR; r0 += 64) {
            const int r = r0 + lane;
            const bool al = r < nR && alive[r];
            int p = -1; bool good = false;
            if (al) { p = tmin[r]; good = p == tmax[r] && p >= 0 && p < nR && p != r && alive[p] && tmin[p] == r && tmax[p] == r; }
            if (__ballot(al && !good) && !code) code = 4;
            const unsigned long long lead = __ballot(al && good && r < p);
            ncore += __popcll(__ballot(al));
            for (unsigned long long m = lead; m; m &= m - 1) {
                const int src = __ffsll((long long)m) - 1;
                const int rr = r0 + src, pp = __builtin_amdgcn_readlane(p, src);
                if (np < MOT_PROV_PAIRS) { if (lane == 0) { prow[2 * np] = rr; prow[2 * np + 1] = pp; } }
                else if (!code) code = 3;
                np++;
            }
        }
        if (!code && np == 0) code = 6;
        // the emulation's kernel must be resident already: the patch step polls for its end
        const int emu = tagged_value(__hip_atomic_load(&L.hdr[LAP_H_EMU], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a.seq, 0);
        if (!code && emu != 1) code = 5;
        if (lane == 0) {
            S.flag[2] = code == 0 ? 1 : 0; S.flag[3] = np;
            // (debug, mot_get_lap_stats()[31]) 1 committed provisionally, else why not (codes above; 5 the emulation's kernel had not started, 6 no core row) | core rows << 8 | pairs << 20
            L.hdr[47] = (code ? code : 1) | (ncore << 8) | (np << 20);
        }
    }
    __syncthreads();
    if (!S.flag[2]) return false;
    const DLState& D = life.S;
    const int np = S.flag[3];
    int T[2 * MOT_PROV_PAIRS], Dm[2 * MOT_PROV_PAIRS], slotT[2 * MOT_PROV_PAIRS];
    ProvRec* rec = life.prov.rec;
#pragma unroll
    for (int i = 0; i < MOT_PROV_PAIRS; i++) {
        if (i >= np) { T[2 * i] = T[2 * i + 1] = 0; Dm[2 * i] = Dm[2 * i + 1] = 0; slotT[2 * i] = slotT[2 * i + 1] = -2; continue; }
        const int rA = prow[2 * i], rB = prow[2 * i + 1], cA = S.colOfRow[rA], cB = S.colOfRow[rB];
        T[2 * i] = rowsTrk ? rA : cA; T[2 * i + 1] = rowsTrk ? rB : cB;          // the two tracks of the cycle ...
        Dm[2 * i] = rowsTrk ? cA : rA; Dm[2 * i + 1] = rowsTrk ? cB : rB;        // ... and the detections they adopt in M; the alternative: each adopts the other's
        slotT[2 * i] = D.slot[T[2 * i]]; slotT[2 * i + 1] = D.slot[T[2 * i + 1]];   // (live order of THIS frame: read before the lifecycle step compacts the list)
        if (tid == 0) { rec->pr[i].rowA = rA; rec->pr[i].rowB = rB; rec->pr[i].colA = cA; rec->pr[i].colB = cB; }
    }
    const int nT_now = *D.nlive;
    __syncthreads();
    if (tid < nR) a.ws.assignment[tid] = S.colOfRow[tid];
    __threadfence_block();
    __syncthreads();
    dl_lifecycle_body(D, life.kp, life.kal, life.trk_pred, life.dets, life.nD, a.ws.assignment, F.life);
    __threadfence_block();
    __syncthreads();
    const int n_new = *D.nlive;
    int* posn = S.wave_tot;                                            // where the tracks sit in the new live list (-1: the track died -- nothing to patch for it)
    if (tid < 2 * MOT_PROV_PAIRS) posn[tid] = -1;
    __syncthreads();
    if (tid < n_new) {
        const int sl = D.slot[tid];
#pragma unroll
        for (int k = 0; k < 2 * MOT_PROV_PAIRS; k++) if (sl == slotT[k]) posn[k] = tid;
    }
    __syncthreads();
    const KcfPool& kp = life.kp;
    const int tot = MOT_NCHAN * kp.nbins;
    int nvalid = 0;
    for (int k = 0; k < 2 * np; k++) {
        const int dalt = Dm[k ^ 1];
        if (posn[k] < 0) { if (tid == 0) { rec->t[k].newpos = -1; rec->t[k].slot = slotT[k]; rec->t[k].sh = -1; rec->t[k].det_alt = dalt; } continue; }
        const int sh = life.prov.sh_base + nvalid, sl = slotT[k];
        // the clone: the model as it is BEFORE the pending blend (the lifecycle step only noted which spectrum the slot adopts)
        const float2* xs = kp.xm + (size_t)sl * tot; float2* xd = kp.xm + (size_t)sh * tot;
        for (int i = tid; i < tot; i += MK_THREADS) xd[i] = xs[i];
        for (int i = tid; i < kp.nbins; i += MK_THREADS) kp.alpha[(size_t)sh * kp.nbins + i] = kp.alpha[(size_t)sl * kp.nbins + i];
        if (tid == 0) {
            const bbox_t bb = life.dets[dalt];
            kp.pos[sh] = bb;                                           // tracker_update's bookkeeping for the other detection (kcf.cpp:470-472), as the lifecycle step does it
            kp.scale[sh] = make_float2(((float)(bb.r - bb.l + 1)) / ((float)kp.cols), ((float)(bb.b - bb.t + 1)) / ((float)kp.rows));
            kp.first_update[sh] = kp.first_update[sl];
            D.pend_det[sh] = dalt;
            D.loc_slots[n_new + nvalid] = sh;
            rec->t[k].newpos = posn[k]; rec->t[k].slot = sl; rec->t[k].sh = sh; rec->t[k].det_alt = dalt; rec->t[k].box_alt = bb;
        }
        nvalid++;
    }
    if (tid == 0) {
        rec->seq = (int)a.seq; rec->npairs = np; rec->nvalid = nvalid; rec->n_new = n_new; rec->nT = nT_now;
        *D.loc_count = n_new + nvalid;
        L.hdr[LAP_H_PMODE] = 0; L.hdr[LAP_H_PROV] = (int)a.seq; L.hdr[LAP_H_PSTAT] += 1;
    }
    return true;
}

// fused != 0 (box costs, not a caller matrix): the dual check, the uniqueness certificate and -- device loop, certified frame --
// the lifecycle step run in THIS workgroup right behind the solver instead of in three more launches.  The dual check is
// spatial: prices are <= 0 (checked per column), so the reduced cost of (i, j) is at least c[i][j] - u_i, and only the columns
// whose box lies within (u_i + eps) * 1280 px of row i's -- looked up in a uniform grid over the column boxes -- can be
// near-tight or infeasible; cross-class entries cost >= 1.0 and matter only for a row with u_i + eps >= 1, which scans every
// column.  Same arithmetic per examined entry as lap_verify_kernel (the dense pass stays for caller matrices and behind the
// dense solver).
// Returns (workgroup-uniform) 1 if the frame is certified here (and, in the device loop, committed), else 2.
// PROV: provisional commits compiled in (the single-workgroup kernel of the stream-emulation chain; the two-workgroup launch never commits provisionally)
template <bool PROV>
__device__ int lap_solve_run(const AssocArgs& a, int fused, const LifeArgs& life, unsigned char* lap_raw)
{
    LapShared& S = *reinterpret_cast<LapShared*>(lap_raw);
    int nR, nC; bool rowsTrk; resolve_dims(a, nR, nC, rowsTrk);
    const LapWs& L = a.ws.lap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long t_begin = wall_clock64();
    if (nR <= 0 || nC <= 0 || nR > nC) { if (tid == 0) L.hdr[LAP_H_SOLVE] = 5; return 2; }
    if (L.hdr[LAP_H_BAD]) { if (tid == 0) L.hdr[LAP_H_SOLVE] = 5; return 2; }
    {   // the candidate lists into LDS: nR * LAP_K <= LAP_K * MK_THREADS entries, all of a thread's loads issued before its first store (the rolled loop
        // waited for global memory once per iteration: 8 round trips at 1024 rows)
        double c8[LAP_K]; unsigned short j8[LAP_K];
        const int ntot = nR * LAP_K;
#pragma unroll
        for (int q = 0; q < LAP_K; q++) { const int i = tid + q * MK_THREADS; const bool in = i < ntot; c8[q] = in ? L.ccost[i] : 0.0; j8[q] = in ? L.ccol[i] : (unsigned short)0xFFFF; }
#pragma unroll
        for (int q = 0; q < LAP_K; q++) { const int i = tid + q * MK_THREADS; if (i < ntot) { S.cc[i] = c8[q]; S.cj[i] = j8[q]; } }
    }
    {   // largest cost of the matrix from the row maxima the row scan left in L.u
        const double rm = wave_min_f64_dpp(tid < nR ? -L.u[tid] : 0.0);
        if (lane == 0) S.red[wave] = -rm;
    }
    S.v[tid] = 0.0; S.rowOfCol[tid] = -1; S.colOfRow[tid] = -1; S.lock[tid] = 0xFFFFFFFFu;
    if (tid < 8) S.flag[tid] = 0;
    __syncthreads();
    double cmax_all = S.red[0];
    for (int w = 1; w < MK_THREADS / 64; w++) cmax_all = fmax(cmax_all, S.red[w]);
    // greedy start: every row asks for its cheapest column, the lowest row wins
    if (tid < nR) atomicMin(&S.lock[S.cj[tid * LAP_K]], (unsigned)tid);
    __syncthreads();
    if (tid < nR) { const int j = S.cj[tid * LAP_K]; if (S.lock[j] == (unsigned)tid) { S.rowOfCol[j] = (short)tid; S.colOfRow[tid] = (short)j; S.mcost[tid] = S.cc[tid * LAP_K]; } }
    __syncthreads();
    int rounds = 0, free0 = -1, searches = 0, commits = 0;
    // ---- round 0: every free row tries, all at once (thread = row), the shortest augmenting path that scans at most ONE column: its
    // best column j0 is free (take it), or j0's owner i2 is scanned and the closest column after that -- among the row's other
    // candidates and i2's candidates -- is free (the row takes it, or i2 moves there and the row takes j0).  Same locks as below
    // (lowest row wins), same dual update (the scanned column's price drops by the path length).  Most free rows end here; the
    // wave-per-search rounds below take what is left. ----
    {
        const int i = tid;
        const bool isfree0 = i < nR && S.colOfRow[i] < 0;
        int h_end = -1, h_scan = -1, h_k = 0, h_k0 = 0, h_i2 = -1, h_via = 0; double h_delta = 0.0;
        S.lock[tid] = 0xFFFFFFFFu;
        __syncthreads();
        if (isfree0) {
            double us = DBL_MAX; int k0 = 0;
#pragma unroll
            for (int k = 0; k < LAP_K; k++) { const int j = S.cj[i * LAP_K + k]; if (j != 0xFFFF) { const double x = S.cc[i * LAP_K + k] - S.v[j]; if (x < us) { us = x; k0 = k; } } }
            const int j0 = S.cj[i * LAP_K + k0];
            const int i2 = S.rowOfCol[j0];
            if (i2 < 0) { h_end = j0; h_k = k0; }
            else {
                const double ui2 = S.mcost[i2] - S.v[j0];
                double best = DBL_MAX; int bj = -1, bk = 0, via = 0;
#pragma unroll
                for (int k = 0; k < LAP_K; k++) {                      // the row's other candidates (earlier slots: they win ties)
                    const int j = S.cj[i * LAP_K + k];
                    if (j != 0xFFFF && k != k0) { const double d = (S.cc[i * LAP_K + k] - S.v[j]) - us; if (d < best) { best = d; bj = j; bk = k; via = 0; } }
                }
#pragma unroll
                for (int k = 0; k < LAP_K; k++) {                      // the owner's candidates
                    const int j = S.cj[i2 * LAP_K + k];
                    if (j != 0xFFFF && j != j0) { const double d = (S.cc[i2 * LAP_K + k] - S.v[j]) - ui2; if (d < best) { best = d; bj = j; bk = k; via = 1; } }
                }
                if (bj >= 0 && S.rowOfCol[bj] < 0) { h_end = bj; h_scan = j0; h_k = bk; h_k0 = k0; h_i2 = i2; h_via = via; h_delta = best; }
            }
            if (h_end >= 0) { atomicMin(&S.lock[h_end], (unsigned)i); if (h_scan >= 0) atomicMin(&S.lock[h_scan], (unsigned)i); }
        }
        __syncthreads();
        if (h_end >= 0 && S.lock[h_end] == (unsigned)i && (h_scan < 0 || S.lock[h_scan] == (unsigned)i)) {
            commits++;
            if (h_scan < 0) { S.colOfRow[i] = (short)h_end; S.rowOfCol[h_end] = (short)i; S.mcost[i] = S.cc[i * LAP_K + h_k]; }
            else {
                S.v[h_scan] -= h_delta;
                if (!h_via) { S.colOfRow[i] = (short)h_end; S.rowOfCol[h_end] = (short)i; S.mcost[i] = S.cc[i * LAP_K + h_k]; }
                else {
                    S.colOfRow[h_i2] = (short)h_end; S.rowOfCol[h_end] = (short)h_i2; S.mcost[h_i2] = S.cc[h_i2 * LAP_K + h_k];
                    S.colOfRow[i] = (short)h_scan; S.rowOfCol[h_scan] = (short)i; S.mcost[i] = S.cc[i * LAP_K + h_k0];
                }
            }
        }
        __syncthreads();
    }
    const long long t_init = wall_clock64(); long long t_sr = 0, t_cm = 0;
    for (;;) {
        const long long t_r0 = wall_clock64();
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
        if (++rounds > 8 * MK_MAXN) { if (tid == 0) S.flag[0] = 1; __syncthreads(); break; }
        __syncthreads();
        const int ns = min(nf, LAP_SR);
        searches += ns;
        for (int q = wave; q < ns; q += MK_THREADS / 64) {
            LapRec& R = S.rec[q];
            const int s0 = S.flist[q];
            // the start row's candidates seed the touched list (slot = candidate index).  Every touched column carries the row that owns
            // it (snapshot), loaded when the column is touched, so scanning a column starts without a dependent LDS read
            int tcol = -1, tpred = s0, tpk = lane, trow = -1; double td = DBL_MAX; bool tscan = false;
            double x0 = DBL_MAX;
            if (lane < LAP_K) { const int j = S.cj[s0 * LAP_K + lane]; if (j != 0xFFFF) { tcol = j; x0 = S.cc[s0 * LAP_K + lane] - S.v[j]; trow = S.rowOfCol[j]; } }
            const double us = row16_min_f64(x0);                       // candidates sit in lanes 0..15
            if (tcol >= 0) td = x0 - us;
            int nt = __popcll(__ballot(tcol >= 0));                    // valid candidates are a prefix of the list
            double Delta = 0.0; int jend = -1; bool fail = false;
            for (;;) {
                // closest touched column that is not scanned yet; equal distance: the earlier slot
                const double best = wave_min_f64_dpp((lane < nt && !tscan) ? td : DBL_MAX);
                const u64 bm = __ballot(lane < nt && !tscan && td == best);
                if (!bm || !(best < DBL_MAX)) { fail = true; break; }  // no augmenting path inside the candidate graph
                const int b = __ffsll((long long)bm) - 1;
                Delta = best;
                const int j = __builtin_amdgcn_readlane(tcol, b);
                const int i = __builtin_amdgcn_readlane(trow, b);
                if (i < 0) { jend = b; break; }
                if (lane == b) tscan = true;
                // one round trip: the owner's matched cost, the column's price, the owner's candidate list
                const int kk = lane & (LAP_K - 1);
                const double mc = S.mcost[i], vj = S.v[j];
                const int cj2v = S.cj[i * LAP_K + kk];
                const double ccv = S.cc[i * LAP_K + kk];
                const double ui = mc - vj;
                // second round trip: price and owner of every candidate
                const int cjc = cj2v == 0xFFFF ? 0 : cj2v;
                const double v2 = S.v[cjc];
                const int ow2 = S.rowOfCol[cjc];
                const double ndc = cj2v != 0xFFFF ? best + ((ccv - v2) - ui) : DBL_MAX;   // lane l holds candidate l & 7
                // which touched slot already holds each candidate (mk: per slot, the candidate that matches it; hit: per candidate)
                int mk = -1; unsigned hit = 0;
#pragma unroll
                for (int k = 0; k < LAP_K; k++) {
                    const int j2 = __builtin_amdgcn_readlane(cj2v, k);
                    const bool eq = lane < nt && tcol == j2;           // 0xFFFF (no candidate) never equals a column
                    if (eq) mk = k;
                    if (__ballot(eq)) hit |= 1u << k;
                }
                // a touched, unscanned column improves (the scanned column itself matches its own scanned slot and is skipped here)
                {
                    const int src = (mk < 0 ? 0 : mk) << 2;
                    const u64 nb = (u64)__double_as_longlong(ndc);
                    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)nb), hi = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)(nb >> 32));
                    const double ndm = __longlong_as_double((long long)(((u64)hi << 32) | lo));
                    if (mk >= 0 && !tscan && ndm < td) { td = ndm; tpred = i; tpk = mk; }
                }
                // candidates that are new to this search take the next slots, in candidate order
                unsigned newm = 0;
#pragma unroll
                for (int k = 0; k < LAP_K; k++) if (__builtin_amdgcn_readlane(cj2v, k) != 0xFFFF) newm |= 1u << k;
                newm &= ~hit;
                if (nt + __popc(newm) > LAP_TS) { fail = true; break; }
                while (newm) {
                    const int k = __ffs((int)newm) - 1; newm &= newm - 1;
                    const int fc = __builtin_amdgcn_readlane(cj2v, k), fo = __builtin_amdgcn_readlane(ow2, k);
                    const double fn = readlane_f64(ndc, k);
                    if (lane == nt) { tcol = fc; td = fn; tpred = i; tpk = k; tscan = false; trow = fo; }
                    nt++;
                }
            }
            if (fail) { if (lane == 0) { R.ok = 0; S.flag[0] = 1; } continue; }
            // record: touched columns, price decrements of the scanned ones, locks
            const bool need = lane < nt && (tscan || lane == jend);
            if (lane < nt) { R.col[lane] = (unsigned short)tcol; R.dec[lane] = Delta - td; R.flag[lane] = (unsigned char)((tscan ? 1 : 0) | (lane == jend ? 2 : 0)); }
            if (need) atomicMin(&S.lock[tcol], (unsigned)q);
            // the augmenting path, end column first
            int plen = 0, t = jend;
            for (int guard = 0; guard <= LAP_TS; guard++) {
                const int pi = __builtin_amdgcn_readlane(tpred, t), pkk = __builtin_amdgcn_readlane(tpk, t), pc = __builtin_amdgcn_readlane(tcol, t);
                if (lane == 0) { R.prow[plen] = (unsigned short)pi; R.pcol[plen] = (unsigned short)pc; R.pk[plen] = (unsigned char)pkk; }
                plen++;
                if (pi == s0) break;
                const int pj = S.colOfRow[pi];                         // a tree row's column is a scanned column of this search
                const u64 m = __ballot(lane < nt && tcol == pj);
                if (!m) { plen = -1; break; }
                t = __ffsll((long long)m) - 1;
            }
            if (lane == 0) { R.nt = nt; R.plen = plen; R.ok = plen > 0; if (plen <= 0) S.flag[0] = 2; }
        }
        __syncthreads();
        const long long t_r1 = wall_clock64(); t_sr += t_r1 - t_r0;
        if (S.flag[0]) break;
        // commits: a search that holds the lock of every column it scanned and of its end column
        for (int q = wave; q < ns; q += MK_THREADS / 64) {
            const LapRec& R = S.rec[q];
            const int nt = R.nt;
            const bool need = lane < nt && R.flag[lane] != 0;
            const bool lost = need && S.lock[R.col[lane]] != (unsigned)q;
            if (__ballot(lost)) continue;
            if (lane == 0) commits++;
            if (lane < nt && (R.flag[lane] & 1)) S.v[R.col[lane]] -= R.dec[lane];
            if (lane < R.plen) {
                const int pi = R.prow[lane], pc = R.pcol[lane];
                S.colOfRow[pi] = (short)pc; S.rowOfCol[pc] = (short)pi; S.mcost[pi] = S.cc[pi * LAP_K + R.pk[lane]];
            }
        }
        __syncthreads();
        t_cm += wall_clock64() - t_r1;
    }
    const long long t_loop = wall_clock64();
    __syncthreads();
    const int status = S.flag[0] ? 1 : 0;
    // duals, Gamma = sum_i (c[i][M(i)] - rowmin_i), margins
    double g = 0.0;
    if (!status && tid < nR) {
        const int j = S.colOfRow[tid];
        const double cm = S.mcost[tid];
        L.u[tid] = cm - S.v[j];
        g = cm - S.cc[tid * LAP_K];
    }
#pragma unroll
    for (int off2 = 32; off2 > 0; off2 >>= 1) g += __shfl_xor(g, off2);
    if (lane == 0) S.red[wave] = g;
    L.v[tid] = S.v[tid]; L.colOfRow[tid] = S.colOfRow[tid]; L.rowOfCol[tid] = S.rowOfCol[tid];
    __syncthreads();
    {   // round 0 counted per thread, the rounds per wavefront (lane 0 carries both)
        int c0 = commits;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c0 += __shfl_xor(c0, o);
        if (lane == 0) S.wave_tot[wave] = c0;
    }
    __syncthreads();
    if (tid == 0) {
        double gamma = 0.0; int ctot = 0;
        for (int w = 0; w < MK_THREADS / 64; w++) { gamma += S.red[w]; ctot += S.wave_tot[w]; }
        const double cmax = cmax_all;
        const double mag = cmax + gamma;
        const double n3 = (double)nC * (double)nC * (double)nC;
        L.dhdr[0] = fmax(1e-9, 1e-15 * n3) * mag;                      // eps: see the header of lap_model.c (CPU model, test infrastructure)
        L.dhdr[1] = 1e-12 * mag;                                       // tol
        L.dhdr[2] = gamma; L.dhdr[3] = cmax;
        S.margins[0] = L.dhdr[0]; S.margins[1] = L.dhdr[1];
        L.hdr[LAP_H_SOLVE] = status;
        L.hdr[LAP_H_LAST + 1] = rounds; L.hdr[LAP_H_LAST + 2] = free0; L.hdr[LAP_H_LAST + 3] = searches; L.hdr[LAP_H_LAST + 4] = ctot;
        L.hdr[LAP_H_LAST + 7] = (int)(wall_clock64() - t_begin);
        L.hdr[49] = (int)(t_init - t_begin); L.hdr[50] = (int)t_sr; L.hdr[51] = (int)t_cm; L.hdr[52] = (int)(wall_clock64() - t_loop);   // (debug: phase ticks)
    }
    if (!fused || a.user) return 2;
    // ================= fused tail: dual check (spatial) -> certificate -> lifecycle =================
    __syncthreads();                                                   // margins are in LDS; every read of `cc` lies behind us
    if (status) return 2;                                              // the solver gave up: the dense solver / the emulation decide (LAP_H_CERT stays 0)
    const long long t_tail = wall_clock64();
    LapFused& F = *reinterpret_cast<LapFused*>(S.cc);
    const double eps = S.margins[0], tol = S.margins[1];
    if (tid == 0) { F.ne = 0; F.viol = 0; }
    const bool big = grid_build(F.grid, a, nR, nC, rowsTrk, S.wave_tot);   // (ends with a barrier)
    bbox_t rb = {}; int rcx = 0, rcy = 0;
    if (tid < nR) { rb = rowsTrk ? a.trk[tid] : a.det[tid]; rcx = (rb.l + rb.r) >> 1; rcy = (rb.t + rb.b) >> 1; }
    // ---- rows: every entry that can be infeasible or near-tight ----
    bool viol = false;
    auto examine = [&](int j, double cst, double ui, int mi) {
        if (j == mi) return;
        const double red = (cst - S.v[j]) - ui;
        if (!(red >= -tol)) viol = true;
        else if (red < eps) {
            const int e = atomicAdd(&F.ne, 1);
            const int owner = S.rowOfCol[j];
            if (e < LAP_EDGES) F.certify[e] = ((unsigned)tid << 16) | (unsigned)(owner >= 0 ? owner : nR);
        }
    };
    if (tid < nR) {
        const int mi = S.colOfRow[tid];
        if (mi < 0 || mi >= nC || S.rowOfCol[mi] != tid) viol = true;      // the matching itself: every row owns exactly the column that names it
        else {
            const double ui = S.mcost[tid] - S.v[mi];                     // == L.u[tid]
            const double reach = ui + eps;
            if (big || !(reach < 1.0)) {
                for (int j = 0; j < nC; j++) {
                    const bbox_t cb = rowsTrk ? a.det[j] : a.trk[j];
                    examine(j, rowsTrk ? pair_cost(rb, cb) : pair_cost(cb, rb), ui, mi);
                }
            } else {
                // >= reach * 1280 + 1: a same-class column farther away costs more than reach (cross-class: cost >= 1 > u_i + eps)
                const int Ri = (int)(reach * (double)MOT_FRAME_W) + 2;
                grid_query(F.grid, rcx, rcy, rb.type, Ri, [&](int j, int d2) { examine(j, cost_of_d2(d2, false), ui, mi); });
            }
        }
    }
    // ---- columns: prices <= 0, exactly 0 on free columns; "a free column could take this column's row" ----
    if (tid < nC) {
        const double vc = S.v[tid];
        const int owner = S.rowOfCol[tid];
        if (!(vc <= 0.0) || (owner < 0 && vc != 0.0)) viol = true;
        else if (owner >= 0 && nC > nR && -vc < eps) {
            const int e = atomicAdd(&F.ne, 1);
            if (e < LAP_EDGES) F.certify[e] = ((unsigned)nR << 16) | (unsigned)owner;
        }
    }
    const int anyviol = __syncthreads_or(viol) ? 1 : 0;
    const int ne = F.ne;
    __syncthreads();
    const int reason = lap_certify(L, nR, nC, F.certify, S.flag, true, ne, anyviol, 0);
    __syncthreads();
    if (tid == 0) {
        L.hdr[LAP_H_VIOL] = anyviol; L.hdr[LAP_H_NEDGES] = ne;
        L.hdr[LAP_H_CERT] = reason + 1;
        if (reason == 0) L.hdr[LAP_H_MODE] = 0;
        L.hdr[55] = (int)(wall_clock64() - t_tail);                      // (debug: dual check + certificate ticks)
    }
    if (reason != 0) {
        // a tie -- but if all that ties is ONE pair of rows that could swap their columns, the frame is committed now and the emulation only owes the swap bit
        if (PROV && reason == 4 && life.enabled && life.prov.enabled && a.stream_emu && ne <= 16 * 64 && lap_try_provisional(a, life, S, F, nR, nC, rowsTrk, ne)) {
            if (tid == 0) { L.hdr[LAP_H_DONE] = 1; L.hdr[56] = (int)(wall_clock64() - t_tail); }
            return 3;
        }
        return 2;
    }
    if (!life.enabled) return 1;
    // certified and in the device loop: commit the frame here (td.cpp:472-644); the rest of the chain returns at once
    if (tid < nR) a.ws.assignment[tid] = S.colOfRow[tid];
    __threadfence_block();
    __syncthreads();
    dl_lifecycle_body(life.S, life.kp, life.kal, life.trk_pred, life.dets, life.nD, a.ws.assignment, F.life);
    if (tid == 0) { L.hdr[LAP_H_DONE] = 1; L.hdr[56] = (int)(wall_clock64() - t_tail); }
    return 1;
}

__global__ void __launch_bounds__(MK_THREADS) lap_solve_kernel(AssocArgs a, int fused, LifeArgs life)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lap_raw[];
    const int verdict = lap_solve_run<true>(a, fused, life, lap_raw);
    if (!a.stream_emu) return;
    // stream emulation (round 6): the sparse emulation runs beside this kernel as a kernel of its own and learns the verdict from the same word
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&a.ws.lap.hdr[LAP_H_VERDICT], tagged_word(a.seq, verdict), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// The sparse order-exact emulation as a kernel of its own on the context's EMULATION stream (device loop with provisional commits): started
// behind the row scan, beside the solver's kernel, it may run on while the main stream goes ahead with the next frame's predict.  LAP_H_EMU
// says where it is: the final kernel of an uncommitted frame and the patch step of a provisionally committed one wait for "finished".
template <bool TIMING>
__global__ void __launch_bounds__(MK_THREADS) mk_sparse_stream_kernel(AssocArgs a, LifeArgs life, int mk_batch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lap_raw[];
    __shared__ int go, nT_now;
    int* ew = &a.ws.lap.hdr[LAP_H_EMU];
    if (threadIdx.x == 0) {
        // the live count, read ONCE for the workgroup: the solver's workgroup (main stream) may be committing this frame -- and changing the count -- right now;
        // waves with different ideas of the problem's dimensions would part ways at the first size test (the run itself is then told to stop by the verdict)
        nT_now = a.nT_dev ? *a.nT_dev : a.nT;
        int old = __hip_atomic_load(ew, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
            if (tagged_value(old, a.seq, 3) == 3) { go = 0; break; }   // the final kernel took the frame before this kernel started (or a later chain is running): nothing to do
            const int seen = atomicCAS(ew, old, tagged_word(a.seq, 1));
            if (seen == old) { go = 1; break; }
            old = seen;
        }
    }
    __syncthreads();
    if (!go) return;
    AssocArgs b = a; LifeArgs lf = life;
    b.nT_dev = nullptr; b.nT = nT_now;
    if (a.det_copy) { b.det = a.det_copy; lf.dets = a.det_copy; }
    mk_sparse_run<true, TIMING>(b, mk_batch, 1, lf, lap_raw);
    __threadfence();                                                   // whatever the run published (results, or a whole committed frame) before "finished"
    __syncthreads();
    // "finished" replaces THIS chain's "started" and nothing else (a compare-and-swap, release).  With several contexts on the shared emulation stream a
    // kernel can start so late that its chain's final kernel has long moved on and the NEXT chain's final kernel has claimed its frame in this word: a plain
    // store would wipe that claim out, the next chain's emulation kernel would start after all, and two workgroups would commit one frame (seen in the
    // two-context soak as a memory fault after ~2,000 repetitions: profiles/r06_prov_soak.log)
    if (threadIdx.x == 0) {
        int expect = tagged_word(a.seq, 1);
        (void)__hip_atomic_compare_exchange_strong(ew, &expect, tagged_word(a.seq, 2), __ATOMIC_RELEASE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Two workgroups, one launch (box costs, device loop or host API): workgroup 0 is the solver with its fused tail, workgroup 1 the sparse
// order-exact emulation, started SPECULATIVELY -- it needs nothing from the solver, only the candidate lists of the row scan -- so on a
// tie frame the emulation is ~40 us into its run when the certificate fails, instead of waiting for a launch of its own behind it; on a
// certified frame it is told to stop (it polls the verdict word once per step-5 cycle) and one launch of the chain is gone.  Exactly one
// of the two commits the frame: the solver's workgroup iff it certifies, else the emulation iff its own check accepts the run; the
// emulation publishes nothing before it has read the verdict.
// TIMING: the emulation's instrumented instantiation (MOT_MK_TIMING=1, probe tools only: its clock reads cost the event loop 6 % even when idle)
template <bool TIMING>
__global__ void __launch_bounds__(MK_THREADS) lap_solve2_kernel(AssocArgs a, LifeArgs life, int mk_batch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lap_raw[];
    if (blockIdx.x == 1) { mk_sparse_run<true, TIMING>(a, mk_batch, 1, life, lap_raw); return; }
    const int verdict = lap_solve_run<false>(a, 1, life, lap_raw);
    __threadfence();                                                   // everything this workgroup wrote (duals, header, lifecycle) before the verdict
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&a.ws.lap.hdr[LAP_H_VERDICT], tagged_word(a.seq, verdict), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- stage 3 ----------------------------------------------------------------------------------------------------
// the dense dual check + near-tight edges, 64 x 64 tiles; a tile's column data (boxes, prices, owners) are staged in LDS
// once so that the sqrt chain of an element does not wait on global loads
__global__ void __launch_bounds__(256) lap_verify_kernel(AssocArgs a, int again)
{
    __shared__ bbox_t colb[64]; __shared__ double colv[64]; __shared__ int colo[64]; __shared__ int big;
    int nR, nC; bool rowsTrk; resolve_dims(a, nR, nC, rowsTrk);
    const LapWs& L = a.ws.lap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 64 + lane, c0 = blockIdx.y * 64;
    if (blockIdx.x * 64 >= nR || c0 >= nC) return;
    if (L.hdr[LAP_H_SOLVE] != 0) return;                               // nothing to check: the solver gave up / was not applicable
    if (again && !L.hdr[LAP_H_DENSE]) return;                          // second pass: only behind the dense solver (lap_dense.hip)
    if (threadIdx.x == 0) big = 0;
    __syncthreads();
    if (threadIdx.x < 64 && c0 + (int)threadIdx.x < nC) {
        const int c = c0 + threadIdx.x;
        if (!a.user) { const bbox_t cb = rowsTrk ? a.det[c] : a.trk[c]; colb[threadIdx.x] = cb; if (!box_small(cb)) big = 1; }
        colv[threadIdx.x] = L.v[c]; colo[threadIdx.x] = L.rowOfCol[c];
    }
    const double eps = L.dhdr[0], tol = L.dhdr[1];
    const double ur = r < nR ? L.u[r] : 0.0;
    const int mr = r < nR ? (int)L.colOfRow[r] : -1;
    bbox_t rb = {};
    if (!a.user && r < nR) rb = rowsTrk ? a.trk[r] : a.det[r];
    __syncthreads();
    const bool est_ok = !a.user && !big && box_small(rb);              // float estimate of the cost usable (assoc_common.h)
    const double far = eps + 4.0 * PAIR_COST_F32_ERR;
    bool viol = false;
    if (r < nR) {
#pragma unroll 4
        for (int cc = wave; cc < 64; cc += 4) {
            const int c = c0 + cc;
            if (c >= nC) break;
            if (c == mr) continue;
            double cst;
            if (a.user) cst = a.user[(size_t)r + (size_t)nR * c];
            else {
                int d2; bool pen; if (rowsTrk) pair_d2(rb, colb[cc], d2, pen); else pair_d2(colb[cc], rb, d2, pen);
                // an entry whose estimated reduced cost is far above eps is feasible and not near-tight: nothing to record
                if (est_ok && ((double)cost_of_d2_f32(d2, pen) - colv[cc]) - ur > far) continue;
                cst = rowsTrk ? pair_cost(rb, colb[cc]) : pair_cost(colb[cc], rb);
            }
            const double red = (cst - colv[cc]) - ur;
            if (!(red >= -tol)) viol = true;
            else if (red < eps) {
                const int e = atomicAdd(&L.hdr[LAP_H_NEDGES], 1);
                const int owner = colo[cc];
                if (e < LAP_EDGES) L.edges[e] = ((unsigned)r << 16) | (unsigned)(owner >= 0 ? owner : nR);
            }
        }
    }
    // the matching itself (first column tile only): every row owns exactly the column that names it -- two rows on one column
    // would pass the dual check and the acyclicity certificate, and be committed as two tracks adopting one detection
    if (blockIdx.y == 0 && wave == 0 && r < nR && (mr < 0 || mr >= nC || (int)L.rowOfCol[mr] != r)) viol = true;
    // per column (first row tile only): prices <= 0, exactly 0 on free columns; "a free column could take this column's row"
    if (blockIdx.x == 0 && threadIdx.x < 64 && c0 + (int)threadIdx.x < nC) {
        const double vc = colv[threadIdx.x];
        const int owner = colo[threadIdx.x];
        if (!(vc <= 0.0) || (owner < 0 && vc != 0.0)) viol = true;
        else if (owner >= 0 && nC > nR && -vc < eps) {
            const int e = atomicAdd(&L.hdr[LAP_H_NEDGES], 1);
            if (e < LAP_EDGES) L.edges[e] = ((unsigned)nR << 16) | (unsigned)owner;
        }
    }
    if (__syncthreads_or(viol) && threadIdx.x == 0) atomicOr(&L.hdr[LAP_H_VIOL], 1);
}

} // namespace

// two_block: the caller will NOT launch mk_sparse_kernel behind this (no dense solver in between): the emulation rides in the solver's launch
hipError_t launch_lap_front(const AssocArgs& a, int gR, int gC, hipStream_t s, hipEvent_t ev_mid, const LifeArgs& life, bool two_block, int mk_batch, const AssocEmu* emu)
{
    hipError_t e = mot_impl::func_lds_once(reinterpret_cast<const void*>(lap_solve_kernel), (int)sizeof(LapShared)); if (e != hipSuccess) return e;
    const int lds2i = (int)(sizeof(LapShared) > sizeof(SpShared) ? sizeof(LapShared) : sizeof(SpShared));
    e = mot_impl::func_lds_once(reinterpret_cast<const void*>(lap_solve2_kernel<false>), lds2i); if (e != hipSuccess) return e;
    e = mot_impl::func_lds_once(reinterpret_cast<const void*>(lap_solve2_kernel<true>), lds2i); if (e != hipSuccess) return e;
    // device loop: the detection features of the split update start on the side stream as soon as the predict is done, beside the
    // row scan (behind it: 2.86 instead of 2.94 M updates/s at 1024 tracks, round 2)
    if (ev_mid) { e = hipEventRecord(ev_mid, s); if (e != hipSuccess) return e; }
    if (a.stream_emu && emu) {
        // Stream emulation (round 6): row scan (its completion event rides in its own packet) -> [emulation stream: the sparse emulation, behind
        // that event] beside [this stream: the solver's workgroup].  The emulation's kernel may outlive this chain: whoever needs its result
        // waits for LAP_H_EMU (the final kernel of an uncommitted frame, the patch step of a provisionally committed one).
        e = mot_impl::func_lds_once(reinterpret_cast<const void*>(mk_sparse_stream_kernel<false>), (int)sizeof(SpShared)); if (e != hipSuccess) return e;
        e = mot_impl::func_lds_once(reinterpret_cast<const void*>(mk_sparse_stream_kernel<true>), (int)sizeof(SpShared)); if (e != hipSuccess) return e;
        hipExtLaunchKernelGGL(lap_rowscan_kernel, dim3((gR * 64 + 3) / 4), dim3(256), 0, s, nullptr, emu->ev_rowscan, 0, a);
        e = hipGetLastError(); if (e != hipSuccess) return e;
        e = hipStreamWaitEvent(emu->stream, emu->ev_rowscan, 0); if (e != hipSuccess) return e;
        mot_impl::lds_poison(emu->stream);
        if (mk_batch & SP_TIMING) hipLaunchKernelGGL(mk_sparse_stream_kernel<true>, dim3(1), dim3(MK_THREADS), sizeof(SpShared), emu->stream, a, life, mk_batch);
        else hipLaunchKernelGGL(mk_sparse_stream_kernel<false>, dim3(1), dim3(MK_THREADS), sizeof(SpShared), emu->stream, a, life, mk_batch);
        mot_impl::lds_poison(s);
        hipLaunchKernelGGL(lap_solve_kernel, dim3(1), dim3(MK_THREADS), sizeof(LapShared), s, a, 1, life);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(lap_rowscan_kernel, dim3((gR * 64 + 3) / 4), dim3(256), 0, s, a);
    mot_impl::lds_poison(s);                                           // (debug) MOT_LDS_POISON
    // box costs: solver + dual check + certificate (+ lifecycle) in one workgroup; caller matrices keep the dense dual check
    const int fused = !a.user ? 1 : 0;
    if (fused && two_block) {
        const size_t lds2 = sizeof(LapShared) > sizeof(SpShared) ? sizeof(LapShared) : sizeof(SpShared);
        if (mk_batch & SP_TIMING) hipLaunchKernelGGL(lap_solve2_kernel<true>, dim3(2), dim3(MK_THREADS), lds2, s, a, life, mk_batch);
        else hipLaunchKernelGGL(lap_solve2_kernel<false>, dim3(2), dim3(MK_THREADS), lds2, s, a, life, mk_batch);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(lap_solve_kernel, dim3(1), dim3(MK_THREADS), sizeof(LapShared), s, a, fused, life);
    if (!fused) hipLaunchKernelGGL(lap_verify_kernel, dim3(gR, gC), dim3(256), 0, s, a, 0);
    return hipGetLastError();
}

hipError_t launch_lap_verify_again(const AssocArgs& a, int gR, int gC, hipStream_t s)
{
    hipLaunchKernelGGL(lap_verify_kernel, dim3(gR, gC), dim3(256), 0, s, a, 1);
    return hipGetLastError();
}
