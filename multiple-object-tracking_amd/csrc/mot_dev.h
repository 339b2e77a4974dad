// mot_dev.h -- shared host/device declarations of the MI355X tracker library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mot_abi.h"

#define MOT_NCHAN 31          // FHOG channels used by KCF (trackers/kcf.cpp:157)
#define MOT_NORI 18           // contrast-sensitive orientations (libhog/gradientMex.cpp:305)
#define MOT_CELL 4            // trackers/kcf.cpp:488
#define MOT_ASSOC_CTL_WORDS (1024 + 4 * 16384)   // assoc_common.h: control words + tagged COVBITS / BMOUT granules
#define MOT_KCF_THREADS 512   // one workgroup per track
#define MOT_KCF_THREADS_SLAB 512   // HBM-slab templates: one workgroup per CU, so the whole CU's wave slots belong to it
#define MOT_LDS_LIMIT (160 * 1024)

struct FastDiv {              // exact n/d for n*d < 2^32
    uint32_t d, m;
    __host__ void init(uint32_t dd) { d = dd; m = (dd <= 1) ? 0u : (uint32_t)(0xFFFFFFFFull / dd + 1ull); }
    __device__ __forceinline__ uint32_t div(uint32_t n) const { return d <= 1 ? n : __umulhi(n, m); }
    __device__ __forceinline__ void divmod(uint32_t n, uint32_t& q, uint32_t& r) const { q = div(n); r = n - q * d; }
};

// One pool = all KCF tracks that share a template size (rows x cols frozen at
// tracker_new, trackers/kcf.cpp:148-152).  State is track-major (SoA of
// per-track contiguous blocks) so a workgroup streams its model with
// consecutive-lane, 8/16-byte accesses.
struct KcfPool {
    // geometry
    int rows, cols;           // patch h, w (column-major, rows fastest)
    int hb, wb;               // f_rows, f_cols
    int fh;                   // hb/2+1
    int nb;                   // hb*wb cells
    int nbins;                // wb*fh half-spectrum bins
    int ldp;                  // LDS patch column stride: roundup4(rows) + 4 (16-byte aligned 4-pixel groups)
    int ng;                   // 4-pixel groups per column
    FastDiv d_rows, d_cols, d_hb, d_fh, d_nbins, d_nb, d_ng;
    float norm;               // 1/(hb*wb*31)  (kcf.cpp:197)
    float eta, lambda;        // kcf.cpp:211-212
    int fhog_mode, fft20;
    // scratch carve (float offsets) and size
    int offA, offB, offC, offT, lds_floats;   // offT: ping-pong buffer of the generic DFT (unused by the 20x20 register FFT)
    int use_lds;              // 1: scratch in LDS, 0: per-workgroup slab in HBM, with region C and a staging area in LDS
    int dft_inplace;          // LDS-resident, direct transforms: both passes in place (no region T in the layout)
    int szC, stage_floats, stage_G;   // !use_lds: LDS floats of region C / of the staging area, channel planes per DFT stage
    // !use_lds, R1-resident mode (r1_lds): LDS = [R1, wave-interleaved | region C in the order tab, tw, red, N, E/resp, zf, tmp | rest].  The gradient /
    // histogram run in stripes of stripe_k cell columns whose Mq / bins live in LDS from offX (= E) on; the DFTs take tile_T channel planes at a time
    // through the same area (offW).  Only the patch, the spectra and the partial correlation sums go through the slab.
    int r1_lds, offR1c, offX, offW, tile_T, stripe_k;
    float* gscratch;          // [grid][lds_floats] when !use_lds
    // state, indexed by slot
    float2* xm;               // [cap][31][nbins]
    float* alpha;             // [cap][nbins]
    bbox_t* pos;              // [cap]   kcf_t::pos  (crop box of the next predict)
    float2* scale;            // [cap]   (scale_horiz, scale_vert)
    int* first_update;        // [cap]
    float* response;          // [cap][nb]
    // constants
    const float* cos_win;     // [nb]
    const float* yf_re;       // [nbins]  only Re(yf) is read by kcf_update_alpha (kcf.cpp:373)
    const float2* tw_r;       // [hb] (cos,sin)(2*pi*j/hb)
    const float2* tw_c;       // [wb]
    const uint16_t* sse_tab;  // [4096] rcp | rsqrt mantissa tables
    // DFTs as f32 MFMA products (HBM-slab templates with hb, wb <= MOT_DFT_MFMA_MAX): the constant operand, stored in the
    // lane order of v_mfma_f32_16x16x4_f32 fragments (one coalesced 256-byte load per fragment)
    const float* mf_rows;     // [ks_r][3][64]: B[k = 4s + lane/16][n = 16nt + lane%16] = (cos, -sin)(2 pi (n/2) k / hb), 0 outside
    const float* mf_cols;     // [3][ks_c][64]: A[x' = 16mt + lane%16][k = 4s + lane/16] = k < wb ? cos(2 pi x' k / wb) : sin(2 pi x' (k - wb) / wb), 0 outside
    const float* mf_cols2;    // [3 mt][3 xt][4 r][cos, sin][64]: A[x' = 16mt + lane%16][x = 16xt + 4(lane/16) + r] (the k order of dft2_mfma)
    int mf;                   // 1: tables present
};
#define MOT_DFT_MFMA_MAX 41    /* line length up to which the MFMA DFT holds its constant fragments in registers */
#define MOT_MF_KS_R ((MOT_DFT_MFMA_MAX + 3) / 4)
#define MOT_MF_KS_C ((2 * MOT_DFT_MFMA_MAX + 3) / 4)
#define MOT_MF_CW_FLOATS (3 * 3 * 4 * 2 * 64)   /* column fragments of the fused transform */

struct KcfLaunch {
    const int* slots;         // [n] pool slot per workgroup (device)
    const int* count;         // optional device count: workgroups >= *count exit
    const uint8_t* frame;     // 1280x720x3 BGR or null
    const float* patches;     // [n][rows*cols] gray patches or null
    const bbox_t* boxes_in;   // update: [n] box to crop at / adopt as pos (null for predict: crop at pos)
    bbox_t* boxes_out;        // predict: [n] predicted boxes
    int clamp;                // apply td.cpp:378-381 to boxes_out
    float* feat_out;          // debug: [n][32][nb] FHOG (optional)
    int feat_windowed;
    long long* dbg;           // debug: phase time stamps of workgroup 0 (100 MHz ticks), or null
    // update kernel, split mode (device-resident loop): the features of every DETECTION box are computed once, while the
    // association runs, and the per-track update only blends them into the model
    float2* spec_out;         // feature-only launch: [n][31][nbins] spectra of boxes_in[item] are written here, no model update
    const float2* det_spec;   // blend launch: spectra written by a feature-only launch ...
    const int* det_index;     // ... and [n] the detection whose spectrum item uses (-1: compute from boxes_in[item] as usual)
    // deferred blend (device loop, split update): the model blend of frame f runs as the prologue of frame f + 1's predict.  pend_det[slot]
    // >= 0: the slot adopts spectrum pend_spec[pend_det[slot]] (written by frame f's feature launch) before it is correlated; reset to -1
    int* pend_det;            // [cap] by SLOT, or null
    const float2* pend_spec;  // spectra of the previous frame's detections
    int grid_stride;          // update kernel: workgroups loop over the items (grid smaller than the device-side count allows)
    int slab_base;            // HBM-slab templates: item i works in slab (slab_base + i) -- a launch that may run beside another KCF launch gets slabs of its own
    // size classes (device loop with per-track template sizes): item i uses pools[cls[i]]; one shared scratch with a common stride
    const KcfPool* pools;     // device table of pool descriptors, or null (single pool passed by value)
    const int* cls;           // [n] class of every item
    int slab_stride;          // floats per slab of the shared HBM scratch (0: the pool's own lds_floats)
    unsigned lds_bytes;       // dynamic LDS of the launch = the largest need of any class
    int r1_any;               // some class runs the R1-resident pipeline (KcfPool::r1_lds): the launch takes the kernels built with it
    int gen_any;              // some LDS-resident class is not 20 x 20 cells: the launch takes the kernels with the direct transforms compiled in
    int ablate;               // (probe build, MOT_KCF_ABLATE) mask of phases to skip -- timing tools only, see kcf_kernels.hip
    // (debug, MOT_TRACE=1) one 8-int record per workgroup into a ring of 16 frames: [frame_no, pos.l, pos.t, arg-max | item << 16, peak bits, pending
    // detection, first_update, new pos.l | new pos.t << 16] for a predict, [frame_no, box.l, box.t, det_index, first, slot, 0, 0] for an update item
    int* trace; int trace_frame, trace_cap;
};

struct KalmanPool {
    double* x;                // [cap][6]
    double* P;                // [cap][36] column-major
};

// ---- association workspace -------------------------------------------------
// assignment fast path (lap_kernels.hip): sparse exact LAP solver + uniqueness certificate
#define LAP_K 8               // candidate columns per row (the K smallest costs)
#define LAP_TS 32             // columns one search may touch
#define LAP_S 128             // concurrent searches per round
#define LAP_EDGES 8192        // near-tight edges the certificate accepts
struct LapWs {
    unsigned short* ccol;     // [1024][LAP_K] candidate columns of every row, ascending cost (0xFFFF: none)
    double* ccost;            // [1024][LAP_K] their costs
    double* v;                // [1024] column prices (<= 0; 0 on unmatched columns)
    double* u;                // [1024] row duals u_i = c[i][M(i)] - v[M(i)]
    short* colOfRow;          // [1024] the solver's matching
    short* rowOfCol;          // [1024]
    unsigned* edges;          // [LAP_EDGES] near-tight edges (row << 16) | (row that owns the column, or nR = "a free column")
    int* hdr;                 // [64] LAP_H_*
    double* dhdr;             // [8]  eps, tol, gamma, cmax
    unsigned long long* cmaxkey;  // order-preserving key of the largest cost (atomicMax)
    short* spAssign;          // [1024] assignment found by the sparse order-exact emulation (mk_sparse.hip)
    double* spS;              // [1024] S_j: what its step 5 passes subtracted from column j in total
};
enum { LAP_H_SOLVE = 0,       // solver status of this launch: 0 ok, 1 gave up, 5 not applicable (negative / non-finite costs)
       LAP_H_NEDGES = 1, LAP_H_VIOL = 2, LAP_H_BAD = 3,
       LAP_H_MODE = 4,        // verdict for the final kernel: 0 certified unique optimum (lap.colOfRow), 1 sparse emulation ran (lap.spAssign, valid
                              // iff LAP_H_SPVIOL == 0), 2 run the dense order-exact emulation
       LAP_H_SPVIOL = 5,      // mk_postcheck_kernel: an entry outside the candidate lists could have mattered
       LAP_H_DENSE = 6,       // lap_dense_kernel ran in this launch (the sparse solver gave up / its prices failed the dense check): the dual check runs again
       LAP_H_DONE = 7,        // the frame is decided AND committed: a kernel of this launch chain has already run the lifecycle step (device loop) --
                              // every later kernel of the chain returns at once; re-armed by the final kernel
       LAP_H_VERDICT = 9,     // two-workgroup solver launch: 0 pending, 1 certified (the solver's workgroup decides and commits), 2 not certified (the
                              // speculative sparse emulation of the second workgroup is wanted); agent-scope release / acquire; re-armed by the final kernel
       LAP_H_EMU = 10,        // stream-emulation protocol (round 6): (chain sequence number << 2) | state of the sparse emulation's kernel on the emulation
                              // stream: 1 started, 2 finished (whatever it publishes is visible), 3 claimed by the final kernel before it could start
       LAP_H_PROV = 11,       // chain sequence number of a PROVISIONALLY committed tie frame whose swap bit is still owed (0: none), see ProvRec
       LAP_H_PMODE = 12,      // ... the emulation's outcome for that frame: 1 accepted (lap.spAssign), 2 refused (the dense emulation decides the bit)
       LAP_H_PSTAT = 37,      // [37] provisional commits, [38] swaps applied, [39] bits decided by the dense emulation (cumulative; mot_get_lap_stats()[21..23])
       LAP_H_CERT = 8,        // 1 + the certificate's outcome when the solver's own workgroup already evaluated it (fused dual check, lap_kernels.hip); 0: not yet
       LAP_H_DSTAT = 42,      // [42..44] dense solver of the most recent launch: settled columns, free rows after the greedy start, device time (10 ns);
                              // [45] launches in which it ran, [46] ... and were then certified (cumulative)
       LAP_H_LAST = 16,       // [16..31] statistics of the most recent launch: status, rounds, free rows, searches, commits, edges, cyclic nodes, device time (10 ns)
       LAP_H_CUM = 32 };      // [32..36] cumulative certificate outcomes (0 certified, 1..4 reasons), [40] sparse emulation accepted, [41] refused -> dense

// ---- provisional commit of a tie frame (round 6, lap_kernels.hip: lap_try_provisional) --------------------------------------------
// When the certificate fails ONLY because of a few disjoint two-row cycles of near-tight edges -- rows A, B could swap their columns at
// (almost) equal cost, every other row is forced -- the reference returns the solver's matching M with some of those swaps applied, and
// nothing else depends on which: the rows are assigned either way, so counters, deaths, spawns, track ids and the order of the live list are
// the same.  The solver's workgroup then commits M at once (lifecycle step included) and clones the tracks of the cycles into SHADOW slots
// that adopt the other detection; the next frame's predict launch computes both alternatives while the order-exact emulation (on a stream
// of its own) is still running, and the patch step (prov_apply, assoc_kernels.hip) copies the shadows over the tracks of every cycle the
// emulation reports as swapped.
#define MOT_PROV_PAIRS 3      /* disjoint two-row cycles one frame may be committed with (seen on the bench stream: 1 in ~77 %, 2 or 3 in the rest of the tie frames) */
#define MOT_SHADOW_SLOTS (2 * MOT_PROV_PAIRS)   /* extra slots at the end of a device-loop KCF pool / predict list / box segment */
struct ProvTrack { int newpos, slot, sh, det_alt; bbox_t box_alt; };   // live position after the lifecycle step (-1: the track died), pool slot, shadow slot, the detection the shadow adopts + its box
struct ProvPair { int rowA, rowB, colA, colB; };   // one cycle in the orientation of the assignment problem: M[rowA] = colA, M[rowB] = colB; its tracks are t[2 i], t[2 i + 1]
struct ProvRec {
    int seq;                  // chain sequence number of the frame (its own copy: LAP_H_PROV is the flag)
    int npairs, nvalid, n_new;   // cycles; shadow items appended to the predict list; live tracks after the lifecycle step
    int nT, pad[3];           // live tracks the frame was associated with (the patch step's dense emulation rebuilds the problem)
    ProvPair pr[MOT_PROV_PAIRS];
    ProvTrack t[2 * MOT_PROV_PAIRS];
};
struct LapProv {
    int enabled;              // 0: no provisional commits (caller matrices, host API, sharded / Kalman / size-class loops, HBM-slab templates, MOT_PROV=0)
    int sh_base;              // first shadow slot of the pool (= its capacity)
    ProvRec* rec;
};

struct AssocWs {
    double* dist;             // [1024*1024] working matrix, column-major
    unsigned long long* zr;   // row-major zero bitmap  [nR][wordsC]
    unsigned long long* zc;   // col-major zero bitmap  [nC][wordsR]
    unsigned long long* linemin; // [1024] order-preserving keys of the row / column minima
    int* assignment;          // [1024]
    double* cost;             // [1]
    int* status;              // [16]: step4, step5, sweeps, -, nR, nC, rowsAreTrackers, perRow, timers
    unsigned long long* ctl;  // control block + result buffers of the step-5 helper workgroups (MOT_ASSOC_CTL_WORDS u64)
    LapWs lap;                // fast-path workspace (lap.ccol == nullptr: not available)
    int* dense_hint;          // pinned host int, written by the final kernel: bit 0 = this launch fell through to the dense emulation, bit 1 = the
                              // sparse solver failed (gave up / infeasible prices): scheduling hints for the NEXT launches -- helper grid, dense
                              // solver kernels -- read by the host without synchronisation, never a result
};

// host-side launchers implemented in the .hip files
// t_start / t_stop (profiling, both or none): the runtime records the KERNEL's own begin / end time stamps in these events
// (hipExtLaunchKernelGGL) -- what rocprofv3 reports for the launch, without the two marker packets an event pair around it adds
hipError_t launch_kcf_predict(const KcfPool& p, const KcfLaunch& l, int n, hipStream_t s, hipEvent_t t_start = nullptr, hipEvent_t t_stop = nullptr);
hipError_t launch_kcf_predict_features(const KcfPool& p, const KcfLaunch& lp, int n_pred, const KcfLaunch& lf, int n_feat, hipStream_t s);   // one launch: predict items, then feature-only items
hipError_t launch_kcf_update(const KcfPool& p, const KcfLaunch& l, int n, hipStream_t s, bool exclusive_cu = false);
hipError_t launch_kcf_fhog_only(const KcfPool& p, const KcfLaunch& l, int n, hipStream_t s);
hipError_t launch_kcf_crop_only(const KcfPool& p, const KcfLaunch& l, int n, float* patch_out, hipStream_t s);
size_t kcf_lds_bytes(const KcfPool& p);
void kcf_pool_layout(KcfPool& p, bool allow_r1 = true, bool allow_inplace = true);

hipError_t launch_kalman_predict(const KalmanPool& p, const int* slots, const int* count, int n, bbox_t* boxes_out, int clamp, hipStream_t s);
hipError_t launch_kalman_update(const KalmanPool& p, const int* slots, const int* count, int n, const bbox_t* boxes, hipStream_t s);
hipError_t launch_kalman_init(const KalmanPool& p, const int* slots, int n, const bbox_t* boxes, hipStream_t s);

// host side of the stream emulation (device loops that commit two-row tie frames provisionally): the emulation stream, the event the row scan's
// launch carries, and where the row scan copies the frame's detection list for the emulation's kernel
struct AssocEmu { hipStream_t stream; hipEvent_t ev_rowscan; bbox_t* det_copy; };
hipError_t launch_assoc(const AssocWs& ws, const bbox_t* trk, const int* nT_dev, int nT, const bbox_t* det, int nD,
                        const double* user_dist, int nR, int nC, int want_cost, hipStream_t s, hipEvent_t ev_mid = nullptr,
                        const struct LifeArgs* life = nullptr, const AssocEmu* emu = nullptr, unsigned* seq_out = nullptr);
// the patch step of a provisionally committed frame (chain `seq`): a no-op unless that frame is still owed its swap bit; trk / det: THAT frame's
// predicted boxes and detection list (the emulation's copy); pred_cur: the boxes the predict launch in between wrote (its shadow items sit behind
// the live tracks), or null when no predict has run since
hipError_t launch_prov_patch(const AssocWs& ws, const struct LifeArgs& life, const bbox_t* trk, const bbox_t* det, int nD, unsigned seq, bbox_t* pred_cur, hipStream_t s);   // life: run the device loop's lifecycle step as the kernel's tail (dl_lifecycle.h)   // ev_mid: recorded between the cost kernels and the Munkres kernel
hipError_t launch_cost_matrix(const bbox_t* trk, int nT, const bbox_t* det, int nD, double* dist_out, hipStream_t s);
