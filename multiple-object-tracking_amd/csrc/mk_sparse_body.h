// mk_sparse_body.h -- the sparse order-exact Munkres emulation as a device function (see mk_sparse.hip for the method), so that it
// can run as a kernel of its own (mk_sparse_kernel) or as the SECOND WORKGROUP of the solver's launch (lap_kernels.hip), where it starts
// speculatively while the solver is still working and learns from a verdict word whether its result is wanted.
#pragma once
#include <stddef.h>
#include "assoc_common.h"
#include "lap_certify.h"
#include "lap_grid.h"
#include "dl_lifecycle.h"

namespace assoc {

// A `volatile` access through a generic pointer is NOT an LDS instruction: hipcc keeps it a FLAT load / store with sc0 sc1 (address-space
// inference skips volatile accesses).  Where a value another wavefront (or another lane, through an LDS atomic) wrote has to be read again, a
// compiler-level fence in front of a plain access does it: the access stays ds_read / ds_write, and the hardware executes one wavefront's LDS
// instructions in order.
#define LDS_REREAD() asm volatile("" ::: "memory")

#define SPK LAP_K
#define SP_TLS 32               /* slots per column in the transposed lists; a column wanted by more rows: not applicable */
#define SP_UQ 64                /* union requests one step 5 can queue; more: the run falls back to the reference's full reset */
#define SP_LAZY_OFF 0x40000000  /* flag in the mk_batch argument: MOT_MK_LAZY=0 */
#define SP_TIMING   0x20000000  /* flag in the mk_batch argument: MOT_MK_TIMING=1 -- the host launches the instrumented instantiation (clock reads and a time-stamp trace for the probe tools) */
#define SP_CX_MAX 16            /* rows with several zeros at the start that the set-up ties together one by one; more: one component */

struct SpShared {
    int lab[MK_MAXN];                        // lazy reset: component label of a column in the zero graph (= smallest column of its component); first: scanned as int4
    unsigned short cj[SPK * MK_MAXN];        // candidate columns, [row][k] (0xFFFF: none): a row's record is one 16-byte read
    double Scol[MK_MAXN];                    // S_j   (member order keeps `tl` 8-byte aligned: it is filled and read as uint2)
    u64 hkey;                                // step 5: order-preserving key of the minimum (LDS atomicMin, one per wavefront)
    u64 covR[MK_MAXW], covC[MK_MAXW], hz[MK_MAXW], hzAll[MK_MAXW];
    unsigned tzero[MK_MAXN];                 // per column: which of its slots hold a zero
    unsigned tlive[MK_MAXN];                 // ... a zero in an UNCOVERED row
    unsigned short tl[SP_TLS * MK_MAXN];     // transposed lists, [column][slot]: (row << 4) | k, rows ascending
    unsigned char pos[SPK * MK_MAXN];        // slot of (row, k) in its column's list, [row][k]
    unsigned short zmask[MK_MAXN];           // zeros of a row as a mask over its candidates
    short starColOfRow[MK_MAXN], starRowOfCol[MK_MAXN], primeColOfRow[MK_MAXN];
    unsigned short clist[MK_MAXN];
    int cnt[MK_MAXN];                        // set-up: fill cursors, then minSimple; event loop: row stamps of a batch (INT_MAX when idle)
    unsigned char starK[MK_MAXN], primeK[MK_MAXN];   // candidate index of a row's starred / primed zero
    unsigned ph32[MK_MAXW * 2];              // batch path: LDS copy of the columns uncovered in this phase
    unsigned dirty32[MK_MAXW * 2];           // columns whose zero masks a step 5 changed (wave 0 refreshes their hz bits)
    unsigned short blist[64];                // batch path: the candidate columns of one batch, ascending
    unsigned uq[2 + SP_UQ];                  // lazy reset: union requests of a step 5: [0] count, [1..] (label << 16 | label) -- count and first request are one 8-byte read
    unsigned char dirtyLab[MK_MAXN];         // ... per label: the component must be re-grown from scratch at the next augmentation
    int wave_tot[MK_THREADS / 64];
    int flag[8];
    unsigned tprof[4];                       // (MOT_MK_TIMING) wave 0's ticks by part: phase start, one-event iterations, batch iterations, augmentations
};
static_assert(sizeof(SpShared) <= MOT_LDS_LIMIT, "mk_sparse_kernel LDS");
static_assert(offsetof(SpShared, tl) % 8 == 0, "transposed lists are accessed as uint2");
static_assert(offsetof(SpShared, lab) % 16 == 0 && offsetof(SpShared, uq) % 8 == 0, "labels are scanned as int4; request count + first request as uint2");
static_assert(offsetof(SpShared, cj) % 16 == 0 && offsetof(SpShared, pos) % 8 == 0 && SPK == 8, "a row's candidate columns / slots are read as one uint4 / uint2");
static_assert(SPK <= 16 && (SPK & (SPK - 1)) == 0 && (SP_TLS & (SP_TLS - 1)) == 0, "candidate index is packed into 4 bits; lane masks");
static_assert(sizeof(SpShared) >= LAP_EDGES * 4 + 2 * (MK_MAXN + 64), "lap_certify scratch");
// after the run the transposed lists are dead: their 64 KB hold the column grid of the fused after-the-fact check and the
// lifecycle step's scratch
struct SpPost { ColGrid grid; int life[DL_LIFE_SCRATCH_INTS]; double red[MK_THREADS / 64]; };
static_assert(sizeof(SpPost) <= sizeof(unsigned short) * SP_TLS * MK_MAXN, "post-check overlay");

__device__ __forceinline__ bool bit_of(const u64* m, int i) { return (m[i >> 6] >> (i & 63)) & 1ull; }
__device__ __forceinline__ void lds_clear_bit64(u64* words, int i) { atomicAnd(reinterpret_cast<unsigned*>(words) + (i >> 5), ~(1u << (i & 31))); }

// post_fused (box costs): the after-the-fact check runs in this workgroup as a disc query per row over a grid of the column boxes
// (same bound per examined entry as mk_postcheck_kernel; an entry that is not examined satisfies it a fortiori, see below), and -- device
// loop -- an accepted run is committed here (lifecycle step), so a tie frame needs no further kernel of the chain.
// SPEC: speculative run beside the solver (second workgroup of its launch).  Nothing is known about the solver's outcome at the start;
// the run polls L.hdr[LAP_H_VERDICT] (0 pending, 1 certified: the solver's workgroup decides and commits the frame, 2 not certified: this
// run's result is wanted) once per step-5 cycle, gives up as soon as it reads 1, and publishes nothing before it has read 2.
// Round 6: the verdict word is TAGGED with the launch chain's sequence number (assoc_common.h: tagged_word), because the speculative run may now be a
// kernel of its own on the emulation stream that outlives its chain's final kernel: a word some other frame left is "nothing yet" if older, "not
// wanted" if newer.  Values: 1 certified (this run is discarded), 2 not certified (this run decides AND commits the frame), 3 not certified, but the
// solver's workgroup has committed its matching provisionally (mot_dev.h: ProvRec): this run only reports -- LAP_H_PMODE, lap.spAssign -- and the
// patch step reads off which of the two optima the reference returns.
__device__ __forceinline__ int sp_verdict(const LapWs& L, unsigned seq) { return tagged_value(__hip_atomic_load(&L.hdr[LAP_H_VERDICT], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT), seq, 1); }
// the in-loop look: only the word's value matters there (nothing the partner wrote is read on its strength -- the run ends by
// sp_wait_verdict's acquire load before it publishes anything), so no cache invalidate and no wait at the load
__device__ __forceinline__ int sp_verdict_peek(const LapWs& L, unsigned seq) { return tagged_value(__hip_atomic_load(&L.hdr[LAP_H_VERDICT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), seq, 1); }

// workgroup-uniform: waits until the solver's workgroup has published its verdict and returns it (1: certified there, this run is discarded)
__device__ inline int sp_wait_verdict(const LapWs& L, int* flag, unsigned seq)
{
    for (int spins = 0;; spins++) {
        if (threadIdx.x == 0) *flag = sp_verdict(L, seq);
        __syncthreads();
        const int v = *flag;
        __syncthreads();
        if (v) return v;
        // the partner never published a verdict ("cannot happen"): this run is NOT committed -- two writers must never be possible.  The
        // verdict word for the final kernel keeps its armed value (2: the dense order-exact emulation decides the frame, or the solver's
        // workgroup if it does certify after all); status word 3 records the time-out (mot_get_lap_stats()[8])
        if (spins > 4000000) { if (threadIdx.x == 0) L.hdr[LAP_H_LAST + 8] = 3; return 1; }
        __builtin_amdgcn_s_sleep(8);
    }
}

template <bool SPEC, bool TIMING>
__device__ void mk_sparse_run(const AssocArgs& a, int mk_batch, int post_fused, const LifeArgs& life, unsigned char* sp_raw)
{
    SpShared& S = *reinterpret_cast<SpShared*>(sp_raw);
    int nR, nC; bool rowsTrk; resolve_dims(a, nR, nC, rowsTrk);
    const LapWs& L = a.ws.lap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long t_begin = wall_clock64();
    if (nR <= 0 || nC <= 0 || nR > nC) { if (tid == 0) L.hdr[LAP_H_MODE] = 2; return; }
    // ---- the fast path's verdict first: a certified unique optimum needs no emulation at all ----
    const int bad = L.hdr[LAP_H_BAD];
    long long t_cert = t_begin;
    if (!SPEC) {
        if (L.hdr[LAP_H_DONE]) return;                                 // certified and committed by the solver's workgroup (fused tail, lap_kernels.hip)
        // the solver's workgroup may already have evaluated the certificate (box costs); behind the dense solver it is evaluated again here
        const int pre = L.hdr[LAP_H_DENSE] ? 0 : L.hdr[LAP_H_CERT];
        const int reason = pre ? pre - 1 : lap_certify(L, nR, nC, reinterpret_cast<unsigned*>(sp_raw), S.flag);
        __syncthreads();
        t_cert = wall_clock64();
        if (reason == 0) { if (tid == 0) L.hdr[LAP_H_MODE] = 0; return; }
    }
    // "this run cannot decide the frame": published -- like everything else -- only once the verdict says the run is wanted
    auto refuse = [&](int st) {
        const int v = SPEC ? sp_wait_verdict(L, &S.flag[6], a.seq) : 2;
        if (v == 1) return;
        if (tid == 0) { if (v == 3) L.hdr[LAP_H_PMODE] = 2; else { L.hdr[LAP_H_MODE] = 2; if (st) L.hdr[LAP_H_LAST + 8] = st; } }
    };
    if (bad) { refuse(0); return; }                                    // negative / non-finite costs: dense emulation
    // ---- set-up: candidate entries (values in registers: only the row's own thread ever touches them), d = c - row minimum
    // (hungarian.cpp:83-89), zero masks, transposed lists ----
    const int r = tid;
    const int wordsC = (nC + 63) >> 6;
    if (tid < 8) S.flag[tid] = 0;
    if (tid == 0) S.uq[0] = 0;
    if (tid < 4) S.tprof[tid] = 0;
    S.cnt[tid] = 0; S.tzero[tid] = 0; S.Scol[tid] = 0.0;
    S.starColOfRow[tid] = -1; S.starRowOfCol[tid] = -1; S.primeColOfRow[tid] = -1;
    if (tid < MK_MAXW) { S.covR[tid] = 0; S.covC[tid] = 0; }
    if (tid < MK_MAXW * 2) S.dirty32[tid] = 0;
    S.lab[tid] = tid; S.dirtyLab[tid] = 1;                             // lazy reset: every column its own component; phase 0 grows everything from scratch
    __syncthreads();
    double dv[SPK]; unsigned short myc[SPK];
    unsigned zm = 0;                                                   // zero bits of the row's candidates
#pragma unroll
    for (int k = 0; k < SPK; k++) { dv[k] = DBL_MAX; myc[k] = 0xFFFF; }
    if (r < nR) {
        // the row's record: all sixteen loads issued before the first use (inside the per-candidate test each cost load waited for its own round trip to HBM:
        // eight dependent latencies, most of the set-up's time); an absent candidate's cost slot holds DBL_MAX and is never used
        double cst[SPK];
#pragma unroll
        for (int k = 0; k < SPK; k++) { myc[k] = L.ccol[(size_t)r * SPK + k]; cst[k] = L.ccost[(size_t)r * SPK + k]; }
        const double rmin = cst[0];
#pragma unroll
        for (int k = 0; k < SPK; k++) {
            if (myc[k] != 0xFFFF) { dv[k] = cst[k] - rmin; atomicAdd(&S.cnt[myc[k]], 1); if (fabs(dv[k]) < DBL_EPSILON) zm |= 1u << k; }
        }
    }
#pragma unroll
    for (int k = 0; k < SPK; k++) S.cj[r * SPK + k] = myc[k];
    S.zmask[r] = (unsigned short)zm;
    __syncthreads();
    {   // transposed lists: fill (any order), then sort each column's list by row.  The longest list bounds a per-column
        // insertion sort (up to 32 entries of one thread: 20 us of dependent LDS round trips in a crowded scene); instead every
        // ENTRY finds its rank by counting the smaller rows of its list -- all reads, then a barrier, then all writes.
        const int mycnt = tid < nC ? S.cnt[tid] : 0;
        if (mycnt > SP_TLS) S.flag[0] = 1;
        __syncthreads();
        if (S.flag[0]) { refuse(2); return; }
        S.cnt[tid] = 0;
        {   // every slot starts as "no entry" (0xFFFF sorts behind every row): 64 KB, 8-byte stores (the array is 8-byte aligned)
            uint2* t8 = reinterpret_cast<uint2*>(S.tl);
            const uint2 ff = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
            for (int i = tid; i < SP_TLS * MK_MAXN * 2 / 8; i += MK_THREADS) t8[i] = ff;
        }
        __syncthreads();
        if (r < nR) {
#pragma unroll
            for (int k = 0; k < SPK; k++) if (myc[k] != 0xFFFF) { const int p = atomicAdd(&S.cnt[myc[k]], 1); S.tl[myc[k] * SP_TLS + p] = (unsigned short)((r << 4) | k); }
        }
        __syncthreads();
        // rank of an entry = entries of its list with a smaller packed value ((row << 4) | k: a row lists a column once, so the order is
        // the row order); four slots per LDS read, empty slots (0xFFFF) never count
        unsigned char rank[SPK];
        if (r < nR) {
#pragma unroll
            for (int k = 0; k < SPK; k++) {
                rank[k] = 0;
                if (myc[k] != 0xFFFF) {
                    const uint2* t2 = reinterpret_cast<const uint2*>(S.tl + myc[k] * SP_TLS);
                    const unsigned me = (unsigned)((r << 4) | k);
                    const int m4 = (S.cnt[myc[k]] + 3) >> 2;
                    int rk = 0;
                    for (int i = 0; i < m4; i++) {
                        const uint2 q = t2[i];
                        rk += ((q.x & 0xFFFFu) < me) + ((q.x >> 16) < me) + ((q.y & 0xFFFFu) < me) + ((q.y >> 16) < me);
                    }
                    rank[k] = (unsigned char)rk;
                }
            }
        }
        __syncthreads();
        if (r < nR) {
#pragma unroll
            for (int k = 0; k < SPK; k++) if (myc[k] != 0xFFFF) {
                S.tl[myc[k] * SP_TLS + rank[k]] = (unsigned short)((r << 4) | k); S.pos[r * SPK + k] = rank[k];
                if ((zm >> k) & 1) atomicOr(&S.tzero[myc[k]], 1u << rank[k]);      // zero masks of the columns, from the entries' side
            }
        }
        __syncthreads();
        S.tlive[tid] = S.tzero[tid];                                   // all rows uncovered
        __syncthreads();
    }
    const long long t_lists = wall_clock64();
    // ---- step 1 (:93-101): rows ascending, each stars its first zero BY COLUMN INDEX whose column is still free.
    // A row with ONE zero ("simple") can only ever want that column: among the simple rows of a column the lowest one gets it, unless
    // a lower row with several zeros took it first.  Only the rows with several zeros ("complex": equal row minima, e.g. two tracks
    // on one centroid) need the ordered pass, and they are few: wavefront 0 walks them in ascending order and asks, per zero column in
    // ascending order, whether a lower simple row claims it (minSimple) or an earlier complex row took it (starRowOfCol). ----
    int* minSimple = S.cnt;                                            // fill cursors are spent
    minSimple[tid] = 0x7FFFFFFF;
    __syncthreads();
    const int nz = __popc(zm);
    int fz = 0xFFFF;
#pragma unroll
    for (int k = 0; k < SPK; k++) if ((zm >> k) & 1) fz = min(fz, (int)myc[k]);
    if (r < nR && nz == 1) atomicMin(&minSimple[fz], r);
    {
        const bool complex_row = r < nR && nz > 1;
        const u64 bal = __ballot(complex_row);
        if (lane == 0) S.wave_tot[wave] = __popcll(bal);
        __syncthreads();
        int off = 0, ncx = 0;
        for (int w = 0; w < MK_THREADS / 64; w++) { const int t = S.wave_tot[w]; if (w < wave) off += t; ncx += t; }
        if (complex_row) S.clist[off + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)r;
        if (tid == 0) S.flag[5] = ncx;                                 // (lazy reset: how many rows start with several zeros)
        __syncthreads();
        if (wave == 0) {                                               // ordered pass over the complex rows: lane = candidate slot
            for (int q = 0; q < ncx; q++) {
                const int rr = S.clist[q]; const unsigned m = S.zmask[rr];
                unsigned key = 0xFFFFu;
                if (lane < SPK && ((m >> lane) & 1)) { const int c = S.cj[rr * SPK + lane]; if (minSimple[c] > rr && S.starRowOfCol[c] < 0) key = (unsigned)c; }
                const unsigned mykey = key;
                key = wave_min_u32_dpp(key);
                if (lane == 0 && key != 0xFFFFu) { S.starColOfRow[rr] = (short)key; S.starRowOfCol[key] = (short)rr; }
                if (mykey == key && key != 0xFFFFu) S.starK[rr] = (unsigned char)lane;
            }
        }
        __syncthreads();
        // simple rows: the lowest claimant of a column stars it unless a complex row holds it
        if (r < nR && nz == 1 && minSimple[fz] == r && S.starRowOfCol[fz] < 0) { S.starColOfRow[r] = (short)fz; S.starRowOfCol[fz] = (short)r; S.starK[r] = (unsigned char)(__ffs((int)zm) - 1); }
        __syncthreads();
        S.cnt[tid] = 0x7FFFFFFF;                                           // from here on: the batch path's row stamps
    }
    {   // step 2a: covered columns = starred columns; hz = hzAll = columns that hold a zero
        const bool has = tid < nC && S.starRowOfCol[tid] >= 0;
        const u64 bal = __ballot(has);
        const u64 hb = __ballot(S.tzero[tid] != 0);
        if (lane == 0) { S.covC[wave] = bal; S.hz[wave] = hb; S.hzAll[wave] = hb; }
    }
    __syncthreads();
    int ncov = 0;
    for (int w = 0; w < wordsC; w++) ncov += __popcll(S.covC[w]);
    bool done = ncov == nR;
    int n_prime = 0, n_s5 = 0, n_aug = 0; long long t_s3 = 0, t_s5 = 0;
    int n_bat = 0, n_seq = 0;                                           // (debug: iterations of the batch / the one-event path)   // (debug split of the event loop)
    const long long t_setup = wall_clock64() - t_begin;
    // Wavefront 0 keeps the 1024-bit masks as 32-bit words, lane l (and its mirror l + 32) holding word l & 31: covered columns /
    // rows, columns uncovered in this phase, columns with a live zero (hzr).  Mirroring the upper half lets
    // every lane store its word to the LDS copies without a branch.
    unsigned* covR32 = reinterpret_cast<unsigned*>(S.covR); unsigned* covC32 = reinterpret_cast<unsigned*>(S.covC); unsigned* hz32 = reinterpret_cast<unsigned*>(S.hz);
    const bool batch_on = (mk_batch & 0xFFFF) != 0;
    const int bthr = (mk_batch & 0xFFFF) > 1 ? (mk_batch & 0xFFFF) : 3;                       // candidate columns in front of the sweep that make a batch worth its fixed cost
    const int l5 = lane & 31;
    const unsigned vC = (l5 * 32 + 32 <= nC) ? ~0u : (l5 * 32 >= nC ? 0u : ((1u << (nC & 31)) - 1u));
    unsigned cC = covC32[l5], cR = 0, ph = 0;
    unsigned cRv = 0, phv = 0;                                         // lazy reset: the covered rows / uncovered star columns of CLEAN components (subsets of cR / ph)
    unsigned hzr = reinterpret_cast<unsigned*>(S.hz)[l5];
    int nstar = ncov;                                                  // starred columns: + 1 per augmentation
    bool hz_dirty = false;                                             // step 5 changed the masks: wave 0 rebuilds hzr
    int status = 0;
    int vseen = 0;                                                     // (wavefront 0, speculative run) the verdict word as last seen
    // ---- LAZY RESET (round 4; CPU model and the argument: mk_sparse_model.c, mks_run, lazy).  Between two step 5s the connected
    // components of the zero graph evolve independently (the sweep visits columns in ascending order and repeats while anything
    // happened: what a component does in pass p depends on its own state only).  The reference uncovers every row after an
    // augmentation and re-grows its whole forest (:324-334): for a component that was grown from scratch in a phase that ran to its
    // end and has seen no zero appear or vanish and no star change since, that re-growth reproduces covers and primes exactly, so
    // it is skipped; everything else is DIRTY and reset.  Labels: lab[c] = smallest column of c's component, flat, merged by this
    // wavefront when a step 5 creates a zero (components never split: exact, just not minimal). ----
    bool lazy_ok = (mk_batch & SP_LAZY_OFF) == 0;
    // TIMING is a template parameter: even switched off at run time the clock reads and their branches cost the event loop 6 % (measured);
    // the instrumented instantiation is launched only under MOT_MK_TIMING=1 (probe tools)
    constexpr bool timing = TIMING;
    // (MOT_MK_TIMING) time stamps of thread 0 along the cycle, in the dense working matrix (unused by this tier): tag << 56 | ticks
    long long* trace = reinterpret_cast<long long*>(a.ws.dist); int tr_n = 1;
    auto tr = [&](int tag) { if (timing && tid == 0 && tr_n < 8190) trace[tr_n++] = ((long long)tag << 56) | (wall_clock64() & 0xFFFFFFFFFFFFFFll); };
    bool scratch_phase = true;                                         // the running phase started from a reset (or is phase 0)
    unsigned char* dirtyLab = S.dirtyLab;
    // merge the components of the columns ca, cb: every column labelled max(la, lb) gets min(la, lb)
    auto lab_union = [&](int ca, int cb) {                             // wavefront 0, uniform arguments
        LDS_REREAD();
        const int la = __builtin_amdgcn_readfirstlane(S.lab[ca]);
        const int lb = __builtin_amdgcn_readfirstlane(S.lab[cb]);
        if (la == lb) return;
        const int lo = min(la, lb), hi = max(la, lb);
        int4* l4 = reinterpret_cast<int4*>(S.lab);
#pragma unroll
        for (int j = 0; j < MK_MAXN / 256; j++) {
            int4 v = l4[j * 64 + lane];
            const bool any = v.x == hi || v.y == hi || v.z == hi || v.w == hi;
            v.x = v.x == hi ? lo : v.x; v.y = v.y == hi ? lo : v.y; v.z = v.z == hi ? lo : v.z; v.w = v.w == hi ? lo : v.w;
            if (any) l4[j * 64 + lane] = v;
        }
        if (lane == 0) dirtyLab[lo] = 1;                               // (only ever called for components that just changed)
    };
    if (wave == 0 && lazy_ok) {                                        // the zeros of the start: a row with several zeros ties their columns together
        const int ncx0 = S.flag[5];
        if (ncx0 > SP_CX_MAX) { for (int i = lane; i < MK_MAXN; i += 64) S.lab[i] = 0; }   // (dense / tie-heavy matrices: one component = the reference's full reset)
        else for (int q = 0; q < ncx0; q++) {
            const int rr = S.clist[q]; const unsigned m = S.zmask[rr];
            const int kf = __ffs((int)m) - 1, cf = S.cj[rr * SPK + kf];
            for (unsigned mm = m & (m - 1); mm; mm &= mm - 1) lab_union(cf, S.cj[rr * SPK + (__ffs((int)mm) - 1)]);
        }
    }
    while (!done) {
        const long long t_a = timing ? wall_clock64() : 0;
        tr(1);
        // ========== steps 3 / 4 / 2a / 2b (:240-334, :192-237): wavefront 0 ==========
        if (wave == 0) {
            // (speculative run) this cycle's look at the verdict word: the load is issued here and consumed at the end of the event phase
            const int vnow = (SPEC && vseen < 2) ? sp_verdict_peek(L, a.seq) : vseen;   // once the result is wanted the word is final: no more looks   (a second look every 8 events inside long phases was measured: -1.5 % overall)
            int action = 0; bool found = false;
            unsigned fm = ~0u;                                         // columns >= `from` (the sweep position, :249)
            if (hz_dirty) {                                            // after a step 5: the columns whose zero masks changed (usually a handful)
                // every lane owns word l5 of the masks (the mirror lane recomputes the same), so no cross-lane traffic is needed.
                // ONE round trip: the dirty word, the union requests' header (count + first request) and the label array
                LDS_REREAD();
                unsigned dw = S.dirty32[l5];
                for (; dw; dw &= dw - 1) {
                    const int b = __ffs((int)dw) - 1, c = l5 * 32 + b;
                    const unsigned bit = 1u << b;
                    hzr = S.tlive[c] ? (hzr | bit) : (hzr & ~bit);
                }
                S.dirty32[l5] = 0;
                hz_dirty = false;
            }
            if (timing && lane == 0) atomicAdd(&S.tprof[0], (unsigned)(wall_clock64() - t_a));
            tr(2);
            // step 4 (:283-334) for the primed, unstarred (row, col); afterwards every row is uncovered again and the sweep restarts
            auto augment = [&](int row, int col, int ke) {
                n_aug++;
                const long long t_g = timing ? wall_clock64() : 0;
                int last = col;
                if (lane == 0) {                                       // two LDS round trips per step of the path: the column's star, then that row's prime
                    int cr = row, cc = col, pk = ke;
                    for (int it = 0; it <= nR + nC; it++) {
                        const int old_r = S.starRowOfCol[cc];
                        S.starColOfRow[cr] = (short)cc; S.starRowOfCol[cc] = (short)cr; S.starK[cr] = (unsigned char)pk;
                        if (old_r < 0) break;
                        cc = S.primeColOfRow[old_r]; pk = S.primeK[old_r]; cr = old_r;
                        if (cc < 0) break;
                    }
                    last = cc;
                }
                last = __builtin_amdgcn_readfirstlane(last);
                scratch_phase = true;
                // LAZY RESET.  cRv / phv: covered rows / uncovered star columns of components that were CLEAN when the last from-scratch
                // phase ended.  Whatever was covered since belongs to a dirty component (events only happen in components that were reset
                // or got a new zero), so only cRv / phv have to be looked up: a member whose component a step 5 or a merge has made dirty
                // in the meantime is reset with the rest.  A column ties every row with a zero in it into one component, so a row that
                // stays covered has no zero in a column touched here.  (Both half-waves hold the same words and do the same, idempotent, work.)
                if (!lazy_ok) { cRv = 0; phv = 0; }                    // the reference's full reset
                else if (__ballot((cRv | phv) != 0)) {
                    for (;;) { LDS_REREAD(); if (S.flag[4] == n_s5) break; __builtin_amdgcn_s_sleep(1); }   // wavefront 1 has merged this cycle's labels
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    LDS_REREAD();
                    if (S.flag[7]) { lazy_ok = false; cRv = 0; phv = 0; }
                    for (unsigned t = cRv; t; t &= t - 1) {
                        const int b = __ffs((int)t) - 1;
                        const int pc = S.primeColOfRow[l5 * 32 + b];
                        if (dirtyLab[S.lab[pc & (MK_MAXN - 1)]]) cRv &= ~(1u << b);
                    }
                    for (unsigned t = phv; t; t &= t - 1) {
                        const int b = __ffs((int)t) - 1;
                        if (dirtyLab[S.lab[l5 * 32 + b]]) phv &= ~(1u << b);
                    }
                }
                // the rows that come back lose their prime (:324-330) and their zeros are live again: tlive == tzero & ~(slots of covered
                // rows), so each zero's slot bit is OR-ed back (fire and forget); the row's record is ONE round trip
                hz32[l5] = hzr;                                        // the LDS copy takes the columns that get live zeros back
                for (unsigned t = cR & ~cRv; t; t &= t - 1) {
                    const int r2 = l5 * 32 + __ffs((int)t) - 1;
                    const unsigned m2 = S.zmask[r2];
                    const uint4 cjr = *reinterpret_cast<const uint4*>(&S.cj[r2 * SPK]);
                    const uint2 psr = *reinterpret_cast<const uint2*>(&S.pos[r2 * SPK]);
                    S.primeColOfRow[r2] = -1;
                    for (unsigned mm = m2; mm; mm &= mm - 1) {
                        const int k = __ffs((int)mm) - 1;
                        const unsigned cw = (k & 4) ? ((k & 2) ? cjr.w : cjr.z) : ((k & 2) ? cjr.y : cjr.x);
                        const int c2 = (int)((k & 1) ? (cw >> 16) : (cw & 0xFFFFu));
                        const unsigned ps = (((k & 4) ? psr.y : psr.x) >> ((k & 3) * 8)) & 0xFFu;
                        atomicOr(&S.tlive[c2], 1u << ps);
                        atomicOr(&hz32[c2 >> 5], 1u << (c2 & 31));
                    }
                }
                cR = cRv; covR32[l5] = cR;
                cC |= ph & ~phv; ph = phv;                             // step 2a for the reset components: their starred columns are covered again
                if (last >= 0 && l5 == (last >> 5)) cC |= 1u << (last & 31);
                S.primeColOfRow[row] = -1;
                LDS_REREAD();
                hzr = hz32[l5];
                fm = ~0u; found = false;
                if (timing && lane == 0) atomicAdd(&S.tprof[3], (unsigned)(wall_clock64() - t_g));
                tr(4);
                return ++nstar == nR;                                  // step 2b
            };
            long long t_it = timing ? wall_clock64() : 0; int it_kind = 0;   // (timing: the previous iteration's kind, 1 one event / 2 batch)
            auto lap_it = [&]() { tr(3 + 8 * it_kind); if (timing) { const long long t = wall_clock64(); if (it_kind && lane == 0) atomicAdd(&S.tprof[it_kind], (unsigned)(t - t_it)); t_it = t; } };
            while (action == 0) {
                lap_it(); it_kind = 0;
                if (++n_prime > 64 * MK_MAXN * MK_MAXN) { action = 4; break; }   // safety, never reached
                unsigned cand = hzr & ~cC & vC & fm;
                unsigned cb = (unsigned)__ballot(cand != 0);           // (the upper half mirrors the lower one)
                if (!cb && found) { found = false; fm = ~0u; cand = hzr & ~cC & vC; cb = (unsigned)__ballot(cand != 0); }   // the sweep found something: one more from column 0 (:246-248)
                if (!cb) { action = 2; break; }
                // three or more candidate columns in front of the sweep?  (non-empty words are counted on the scalar side; the
                // per-lane prefix counts of the batch path come later, only when they are needed)
                int total = 0;
                if (batch_on) {
                    const int nw = __popc(cb);
                    if (nw >= bthr) total = bthr;
                    else {                                             // fewer non-empty words than the threshold: count their candidates
                        for (unsigned t = cb; t; t &= t - 1) total += __popc((unsigned)__builtin_amdgcn_readlane((int)cand, __ffs((int)t) - 1));
                    }
                }
                if (total >= bthr) {
                    n_bat++; it_kind = 2;
                    // ---------- BATCH: up to 64 consecutive events of this sweep at once (lane = event).  Taken together are the events
                    // of the first f candidate columns such that (1) their first uncovered zero rows are pairwise different, (2) no row but
                    // possibly the last one is unstarred, (3) no column uncovered by one of them (its row's star column) that still has a
                    // live zero lies in front of a later one -- then no event changes what a later one of the batch sees, and covers,
                    // primes and live masks end exactly as after the f sequential iterations (CPU model: mks_solve_batched). ----------
                    covC32[l5] = cC; S.ph32[l5] = ph; hz32[l5] = hzr;  // the LDS copies take the updates
                    int before = 0; total = 0;                         // candidates in lower words / in all words (counts of the 32 words, bit-sliced through ballots)
                    {
                        const int pc = lane < 32 ? __popc(cand) : 0;
                        const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
                        for (int bb = 0; bb < 6; bb++) { const u64 mb = __ballot((pc >> bb) & 1); before += __popcll(mb & lt) << bb; total += __popcll(mb) << bb; }
                    }
                    if (lane < 32) { unsigned t = cand; int o = before; while (t && o < 64) { S.blist[o++] = (unsigned short)(lane * 32 + __ffs((int)t) - 1); t &= t - 1; } }
                    const int ncand = min(total, 64);
                    const bool act = lane < ncand;
                    LDS_REREAD();
                    const int col = act ? (int)S.blist[lane] : 0;
                    const unsigned tlv = S.tlive[col];
                    const unsigned e = S.tl[col * SP_TLS + (tlv ? __ffs((int)tlv) - 1 : 0)];
                    if (__ballot(act && tlv == 0)) { action = 4; break; }   // hz out of step with the masks: cannot happen
                    const int row = (int)(e >> 4) & (MK_MAXN - 1), ke = (int)(e & 15);
                    const int sc = S.starColOfRow[row];
                    const unsigned m = S.zmask[row];
                    if (act) atomicMin(&S.cnt[row], lane);                 // the first event that wants a row owns it
                    const unsigned tls = S.tlive[sc >= 0 ? sc : 0];
                    const unsigned psb = 1u << S.pos[row * SPK + (int)(S.starK[row] & (SPK - 1))];   // (meaningless, and unused, for an unstarred row)
                    LDS_REREAD();
                    const int owner = S.cnt[row];
                    const unsigned limit = (act && sc > col && (tls & ~psb)) ? (unsigned)sc : 0x7FFFFFFFu;
                    const unsigned smin = wave_min_u32_dpp(limit);
                    const u64 stop = __ballot(act && ((unsigned)col > smin || owner != lane));
                    const u64 augm = __ballot(act && sc < 0);
                    int f = stop ? __ffsll((long long)stop) - 1 : ncand;
                    const int ia = augm ? __ffsll((long long)augm) - 1 : 64;
                    f = min(f, ia + 1);
                    if (act) S.cnt[row] = 0x7FFFFFFF;
                    if (lane < f) { S.primeColOfRow[row] = (short)col; S.primeK[row] = (unsigned char)ke; }   // :255
                    if (lane < f && sc >= 0) {                         // cover the row (:270), uncover its star's column (:271)
                        atomicOr(&covR32[row >> 5], 1u << (row & 31));
                        atomicAnd(&covC32[sc >> 5], ~(1u << (sc & 31)));
                        atomicOr(&S.ph32[sc >> 5], 1u << (sc & 31));
                        for (unsigned mm = m; mm; mm &= mm - 1) {          // its zeros leave the live masks
                            const int kk = __ffs((int)mm) - 1, c2 = S.cj[row * SPK + kk];
                            const unsigned bitv = 1u << S.pos[row * SPK + kk];
                            if (atomicAnd(&S.tlive[c2], ~bitv) == bitv) atomicAnd(&hz32[c2 >> 5], ~(1u << (c2 & 31)));
                        }
                    }
                    n_prime += f - 1;
                    LDS_REREAD();
                    cR = covR32[l5]; cC = covC32[l5];
                    ph = S.ph32[l5]; hzr = hz32[l5];
                    if (ia < f) {
                        if (augment(__builtin_amdgcn_readlane(row, ia), __builtin_amdgcn_readlane(col, ia), __builtin_amdgcn_readlane(ke, ia))) { action = 3; break; }
                        continue;
                    }
                    found = true;
                    {   // the sweep continues behind the last column taken (:273)
                        const int from = __builtin_amdgcn_readlane(col, f - 1) + 1, fw = from >> 5;
                        fm = l5 < fw ? 0u : (l5 == fw ? (~0u << (from & 31)) : ~0u);
                    }
                    continue;
                }
                const int cw = __ffs((int)cb) - 1;
                const int col = cw * 32 + __ffs(__builtin_amdgcn_readlane((int)cand, cw)) - 1;
                // first uncovered row holding a zero in this column: ONE LDS round trip (every lane issues both loads, no branch between them)
                const unsigned tlv = S.tlive[col];
                const unsigned tle = S.tl[col * SP_TLS + l5];
                const unsigned lv = (unsigned)__builtin_amdgcn_readfirstlane((int)tlv);
                const unsigned ent = (unsigned)__builtin_amdgcn_readlane((int)tle, lv ? __ffs((int)lv) - 1 : 0);   // (before the check: both loads are issued together)
                const int row = (int)(ent >> 4);
                if (lv == 0) { action = 4; break; }                    // hz out of step with the masks: cannot happen
                // the row's star, its zeros and where they sit in their columns' lists: one more round trip, all loads issued together
                const int kk = lane & (SPK - 1);
                const int sc_v = S.starColOfRow[row];
                const unsigned m_v = S.zmask[row];
                const int c2 = S.cj[row * SPK + kk];
                const int ps = S.pos[row * SPK + kk];
                const int sc = __builtin_amdgcn_readfirstlane(sc_v);
                const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)m_v);
                S.primeColOfRow[row] = (short)col;                     // :255 (every lane stores the same value)
                S.primeK[row] = (unsigned char)(ent & 15);
                n_seq++; it_kind = 1;
                if (sc < 0) {
                    if (augment(row, col, (int)(ent & 15))) { action = 3; break; }
                    continue;
                }
                // cover the row (:270): its zeros leave the live masks; uncover its star's column (:271)
                cR |= (l5 == (row >> 5)) ? (1u << (row & 31)) : 0u;
                covR32[l5] = cR;
                if ((m & (m - 1)) == 0) {
                    // the row's only zero is the one in this column (the usual case): its slot is the first live one of the mask just
                    // read, so nothing has to come back from the LDS -- the column is empty iff that was its only live bit
                    if (lane == 0) atomicAnd(&S.tlive[col], ~(lv & (0u - lv)));
                    if ((lv & (lv - 1)) == 0) hzr &= ~((l5 == (col >> 5)) ? (1u << (col & 31)) : 0u);
                } else {
                    bool emptied = false;
                    if (lane < SPK && ((m >> lane) & 1)) { const unsigned bitv = 1u << ps; emptied = atomicAnd(&S.tlive[c2], ~bitv) == bitv; }
                    for (u64 eb = __ballot(emptied); eb; eb &= eb - 1) {   // columns that lost their last live zero (usually none or one)
                        const int ce = __builtin_amdgcn_readlane(c2, __ffsll((long long)eb) - 1);
                        hzr &= ~((l5 == (ce >> 5)) ? (1u << (ce & 31)) : 0u);
                    }
                }
                {
                    const unsigned sb = (l5 == (sc >> 5)) ? (1u << (sc & 31)) : 0u;
                    cC &= ~sb; ph |= sb;
                }
                found = true;
                {   // the sweep continues behind this column (:273)
                    const int from = col + 1, fw = from >> 5;
                    fm = l5 < fw ? 0u : (l5 == fw ? (~0u << (from & 31)) : ~0u);
                }
            }
            lap_it();
            tr(5);
            covC32[l5] = cC;
            if (SPEC) vseen = vnow;
            if (lane == 0) { S.flag[1] = action; S.flag[3] = (lazy_ok && scratch_phase) ? 1 : 0; S.hkey = ~0ull; if (SPEC) S.flag[6] = vnow; }
            if (scratch_phase) { cRv = cR; phv = ph; }                 // a from-scratch phase ran to its end: every component is clean
            scratch_phase = false;
        }
        else if (wave == 1 && n_s5 > 0 && (mk_batch & SP_LAZY_OFF) == 0) {
            // LAZY RESET, label upkeep (beside wavefront 0's event phase, which needs the labels only when an augmentation finds members of
            // clean components: it waits for flag[4] == this cycle then): the zeros the last step 5 created tie components together
            LDS_REREAD();
            if (S.flag[7] == 0) {
                const int nu = __builtin_amdgcn_readfirstlane((int)S.uq[0]);
                if (nu > SP_UQ) { if (lane == 0) S.flag[7] = 1; }      // more than the queue holds: the reference's full reset from here on
                else for (int i = 0; i < nu; i++) {                    // (a request carries the labels its thread saw; a label is a column of its component, so lab[] of it is current)
                    const unsigned q = (unsigned)__builtin_amdgcn_readfirstlane((int)S.uq[1 + i]);
                    lab_union((int)(q >> 16), (int)(q & 0xFFFFu));
                }
                if (nu && lane == 0) S.uq[0] = 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) S.flag[4] = n_s5;
            LDS_REREAD();
        }
        __syncthreads();
        const int action = S.flag[1];
        if (S.flag[3]) S.dirtyLab[tid] = 0;                            // a from-scratch phase ran to its end: every component is clean (marks of this step 5 come behind the next barrier)
        if (SPEC && S.flag[6] == 1) return;                            // certified by the solver's workgroup: this run is not wanted (uniform: read behind the barrier)
        const long long t_b = timing ? wall_clock64() : 0;
        tr(6);
        t_s3 += t_b - t_a;
        if (action == 3) { done = true; break; }
        if (action == 4) { status = 2; break; }
        // ================= step 5 (:337-368) on the candidate entries: thread = row =================
        n_s5++;
        // straight-line code (no per-candidate predication): an absent candidate has the value DBL_MAX, counts as covered and is
        // carried through the update unchanged in effect (DBL_MAX + h == DBL_MAX); "x + 0.0" / "x - 0.0" are exact, so selecting
        // the addend instead of the operation gives the reference's bits (:355-364)
        const bool rc = bit_of(S.covR, r);
        bool unc[SPK];
        double h = DBL_MAX;
        {
            // the eight cover words are loaded together (one LDS round trip, no per-candidate predication: an absent candidate names
            // column 1023's word and its value DBL_MAX makes the answer irrelevant -- DBL_MAX +/- h == DBL_MAX); the minimum is a chain
            // of v_min_f64 (no NaNs here; the sign of a zero minimum cannot matter: h > 0 whenever step 5 runs)
            unsigned cw8[SPK];
#pragma unroll
            for (int k = 0; k < SPK; k++) cw8[k] = covC32[(myc[k] & (MK_MAXN - 1)) >> 5];
#pragma unroll
            for (int k = 0; k < SPK; k++) {
                unc[k] = !((cw8[k] >> (myc[k] & 31)) & 1u);
                const double cand_v = (!rc && unc[k]) ? dv[k] : DBL_MAX;
                h = vmin_f64(cand_v, h);
            }
        }
        {
            const u64 hk = dkey(wave_min_f64_pos(h));
            tr(7);
            if (lane == 0) atomicMin(&S.hkey, hk);
        }
        __syncthreads();
        tr(8);
        h = dunkey(S.hkey);
        if (!(h < DBL_MAX)) { status = 1; break; }                     // the minimum lies outside the candidate lists: not applicable
        unsigned nm = 0;
        const double hr = rc ? h : 0.0;
#pragma unroll
        for (int k = 0; k < SPK; k++) {
            const double x = (dv[k] + hr) - (unc[k] ? h : 0.0);        // :355-358, :361-364
            dv[k] = x;
            nm |= (fabs(x) < DBL_EPSILON) ? (1u << k) : 0u;
        }
        if (nm != zm) {                                                // zero bits that changed: the column-side masks follow
            // (rare: a thread or two per step.  Columns and slots are read from the LDS records, not from the registers: indexing myc[] by a
            // run-time k would unroll this block eight-fold and park its addresses in registers across the whole loop -- they spilled)
            const unsigned short* cjr = &S.cj[r * SPK]; const unsigned char* psr = &S.pos[r * SPK];
#pragma unroll 1
            for (unsigned ch = nm ^ zm; ch; ch &= ch - 1) {
                const int k = __ffs((int)ch) - 1;
                const int c = cjr[k];
                const unsigned bitv = 1u << psr[k];
                if ((nm >> k) & 1) { atomicOr(&S.tzero[c], bitv); if (!rc) atomicOr(&S.tlive[c], bitv); }
                else { atomicAnd(&S.tzero[c], ~bitv); if (!rc) atomicAnd(&S.tlive[c], ~bitv); }
                atomicOr(&S.dirty32[c >> 5], 1u << (c & 31));
                S.dirtyLab[S.lab[c]] = 1;                              // lazy reset: the component continued from an older state
            }
            if ((nm & ~zm) && (nm & (nm - 1))) {                       // a zero appeared in a row that holds several now: its columns are one component
                const unsigned lf = (unsigned)S.lab[cjr[__ffs((int)nm) - 1]];
#pragma unroll 1
                for (unsigned mm = nm & (nm - 1); mm; mm &= mm - 1) {
                    const unsigned lk = (unsigned)S.lab[cjr[__ffs((int)mm) - 1]];
                    if (lk == lf) continue;
                    const unsigned qi = atomicAdd(&S.uq[0], 1u);
                    if (qi < SP_UQ) S.uq[1 + qi] = (lf << 16) | lk;
                }
            }
            zm = nm; S.zmask[r] = (unsigned short)nm;
        }
        if (tid < nC && !bit_of(S.covC, tid)) S.Scol[tid] += h;
        hz_dirty = true;
        tr(9);
        __syncthreads();
        if (timing) t_s5 += wall_clock64() - t_b;
    }
    __syncthreads();
    const int vfin = SPEC ? sp_wait_verdict(L, &S.flag[6], a.seq) : 2;
    if (vfin == 1) return;                                             // certified there: nothing of this run is published
    const bool report_only = vfin == 3;                                // the frame is committed provisionally: this run owes the swap bit, nothing else
    if (tid < nR) L.spAssign[tid] = S.starColOfRow[tid];
    L.spS[tid] = S.Scol[tid];
    if (tid == 0) {
        if (!report_only) L.hdr[LAP_H_MODE] = status == 0 ? 1 : 2;
        else if (status != 0) L.hdr[LAP_H_PMODE] = 2;
        L.hdr[LAP_H_LAST + 8] = status; L.hdr[LAP_H_LAST + 9] = n_aug; L.hdr[LAP_H_LAST + 10] = n_s5; L.hdr[LAP_H_LAST + 11] = n_prime;   // (wave-0 / thread-0 counts)
        L.hdr[LAP_H_LAST + 12] = (int)t_s3; L.hdr[LAP_H_LAST + 13] = (int)t_s5; L.hdr[LAP_H_LAST + 14] = (int)(wall_clock64() - t_begin);
        L.hdr[48] = (int)t_setup; L.hdr[53] = (int)(t_cert - t_begin); L.hdr[54] = (int)(t_lists - t_cert);
        if (timing) trace[0] = tr_n;
        L.hdr[58] = n_bat; L.hdr[59] = n_seq; L.hdr[60] = (int)S.tprof[0]; L.hdr[61] = (int)S.tprof[1]; L.hdr[62] = (int)S.tprof[2]; L.hdr[63] = (int)S.tprof[3];                             // (debug: event-loop split)   // (debug: set-up ticks: total, certificate, candidate + transposed lists)
    }
    if (!post_fused || a.user || status != 0) return;
    // ================= fused after-the-fact check (see the header): every entry OUTSIDE the lists must satisfy
    //     (c[i][j] - rowmin_i) - S_j > margin.
    // Outside entries cost at least lc_i (the row's last candidate) and S_j <= Smax, so a row with (lc_i - rowmin_i) - Smax > margin
    // is done without looking at any column (rounding is monotone: the bound holds for the float64 expressions themselves).  Otherwise
    // only columns with cost <= rowmin_i + Smax + margin can fail: same-class boxes within that radius (grid query; the radius carries
    // a whole pixel = 7.8e-4 cost units of slack), cross-class ones (cost >= 1) only if the radius reaches 1 -- then every column is examined.
    const long long t_post = wall_clock64();
    __syncthreads();                                                   // every read of the transposed lists lies behind us
    SpPost& P = *reinterpret_cast<SpPost*>(S.tl);
    const bool big = grid_build(P.grid, a, nR, nC, rowsTrk, S.wave_tot);
    {
        const double m = wave_min_f64_dpp(tid < nC ? -S.Scol[tid] : 0.0);
        if (lane == 0) P.red[wave] = -m;
    }
    __syncthreads();
    double Smax = P.red[0];
    for (int w = 1; w < MK_THREADS / 64; w++) Smax = fmax(Smax, P.red[w]);
    const double margin = 1e-9 * (1.0 + L.dhdr[3]);                    // as mk_postcheck_kernel
    bool pviol = false;
    if (r < nR && myc[SPK - 1] != 0xFFFF) {                            // (a row with fewer than LAP_K columns has every entry in its list)
        const int lj = myc[SPK - 1];
        const double rmin = L.ccost[(size_t)r * SPK], lc = L.ccost[(size_t)r * SPK + SPK - 1];
        if (!((lc - rmin) - Smax > margin)) {
            const bbox_t rb = rowsTrk ? a.trk[r] : a.det[r];
            auto examine = [&](int j, double cst) {
                const bool outside = cst > lc || (cst == lc && j > lj);
                if (outside && !(cst - rmin - S.Scol[j] > margin)) pviol = true;
            };
            const double reach = rmin + Smax + margin;
            if (big || !(reach < 0.99)) {
                for (int j = 0; j < nC; j++) { const bbox_t cb = rowsTrk ? a.det[j] : a.trk[j]; examine(j, rowsTrk ? pair_cost(rb, cb) : pair_cost(cb, rb)); }
            } else {
                const int Ri = (int)(reach * (double)MOT_FRAME_W) + 2;
                grid_query(P.grid, (rb.l + rb.r) >> 1, (rb.t + rb.b) >> 1, rb.type, Ri, [&](int j, int d2) { examine(j, cost_of_d2(d2, false)); });
            }
        }
    }
    const int spviol = __syncthreads_or(pviol) ? 1 : 0;
    if (report_only) { if (tid == 0) { L.hdr[LAP_H_PMODE] = spviol ? 2 : 1; L.hdr[57] = (int)(wall_clock64() - t_post); } return; }
    if (tid == 0) { L.hdr[LAP_H_SPVIOL] = spviol; L.hdr[57] = (int)(wall_clock64() - t_post); }
    if (spviol || !life.enabled) return;
    // accepted and in the device loop: commit the frame here (td.cpp:472-644); the final kernel only does its bookkeeping
    if (tid < nR) a.ws.assignment[tid] = S.starColOfRow[tid];
    __threadfence_block();
    __syncthreads();
    dl_lifecycle_body(life.S, life.kp, life.kal, life.trk_pred, life.dets, life.nD, a.ws.assignment, P.life);
    if (tid == 0) L.hdr[LAP_H_DONE] = 1;
}


} // namespace assoc
