// dl_lifecycle.h -- device-resident live-track list (tracker_info[] of top/td.cpp:312) and the per-frame lifecycle
// step (td.cpp:472-644), shared by mot_devloop.hip and assoc_kernels.hip.
#pragma once
#include "mot_dev.h"

#define DL_MAX_CLASSES 128
#define DL_MAX_WORLD 64           /* ranks a sharded context can address (one lane of a wavefront each in the spawn assignment) */
#define DL_LIFE_SCRATCH_INTS (2048 + 16 + 4 + DL_MAX_CLASSES + DL_MAX_WORLD)

struct DLState {
    int* nlive; unsigned* next_tid; int* nfree; int* free_slots;
    int* slot; unsigned* tid; int* age; int* vis; int* inv; bbox_t* bbox;   // [cap] live list, td.cpp order
    int* rankpos;                 // [cap] index inside the owner's all-gather segment
    // Sharding (round 5): the rank that owns a live track is part of the replicated live list.  A spawning track goes to the rank that owns the
    // FEWEST live tracks at that moment (lowest rank on a tie; detection order) -- round-robin by creation id while nothing dies, and no rank
    // ever owns more than ceil(cap / world) tracks, so a segment of the all-gather holds exactly that many boxes (3 KB at 1024 tracks on 8
    // GPUs) and the predict / update grids of a rank are shard-sized.  (Rounds 1-4: owner = tid % world, which drifts under track
    // churn -- every segment had to hold max_tracks boxes.)
    int* owner;                   // [cap] owning rank of every live track (all zero when world == 1)
    int* loc_slots; int* loc_count;
    int* upd_slots; bbox_t* upd_boxes; int* upd_count;
    int* upd_det;                 // [cap + max_dets] detection whose box an update item adopts (-1: the predicted box, td.cpp:540-560)
    bbox_t* pred;                 // [cap] predicted boxes in live order
    bbox_t* gather;               // [world*spr] all-gather buffer (own segment written by predict)
    int* err;                     // [8]: spawns dropped for template-size mismatch, capacity drops, pool exhausted, all-gather segment overflow,
                                  //      [4] sticky: a Munkres helper hand-off timed out (frame dropped; read-backs return MOT_ERR_DEVICE)
    int cap, max_dets, rank, world, spr, rows, cols, kind;
    // size classes (KCF): per-track template sizes as the reference has them (rows / cols frozen at tracker_new from the spawning
    // detection, kcf.cpp:148-152, td.cpp:626-627): square templates cls_lo .. cls_lo + ncls - 1 px, one pool per class.  ncls <= 1:
    // the single rows x cols template.
    int ncls, cls_lo;
    int* cls;                     // [cap] class of every live track
    int* loc_cls; int* upd_cls;   // class of every predict / update item
    int* nfree_c; int* free_c;    // [ncls] fill of, [ncls][cap] the per-class free slot stacks
    const KcfPool* pools;         // [ncls] device table of pool descriptors
    // deferred blend (KCF, split update, single template): an assigned or spawning track does not enter the update list; the lifecycle
    // step sets its pos / scale (kcf.cpp:470-472) and notes in pend_det[slot] which detection's spectrum the slot adopts, and the NEXT
    // frame's predict kernel blends it into the model before correlating (kcf_predict_body)
    int defer; int* pend_det;
};

__device__ __forceinline__ int block_excl_scan_flag(bool flag, int* wave_tot, int& total)
{   // exclusive prefix count of `flag` over a 1024-thread workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(flag);
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int off = 0; total = 0;
    for (int w = 0; w < 16; w++) { const int t = wave_tot[w]; if (w < wave) off += t; total += t; }
    return off + pre;
}

// builds rankpos[] and this rank's predict list for the CURRENT live list
__device__ inline void dl_build_lists(const DLState& S, int n, int* wave_tot)
{
    const int t = threadIdx.x;
    const int r = (t < n) ? (S.world > 1 ? S.owner[t] : 0) : -1;
    int mine_total = 0;
    for (int rk = 0; rk < S.world; rk++) {
        int total;
        const int pos = block_excl_scan_flag(r == rk, wave_tot, total);
        if (r == rk) {
            S.rankpos[t] = pos;
            if (pos >= S.spr) atomicAdd(&S.err[3], 1);                 // cannot happen: no rank ever owns more than ceil(cap / world) = spr tracks (DLState::owner)
            else if (rk == S.rank) {
                S.loc_slots[pos] = S.slot[t];
                if (S.ncls > 1) S.loc_cls[pos] = S.cls[t];
                if (S.kind == MOT_TRACKER_KALMAN) S.gather[(size_t)S.rank * S.spr + pos] = S.bbox[t];   // predict is in/out (kalman.cpp:112-115)
            }
        }
        if (rk == S.rank) mine_total = total;
    }
    if (t == 0) *S.loc_count = mine_total;
}

// td.cpp:472-644 for one frame, executed by ONE 1024-thread workgroup: assignment scatter, counters, lost rule, stable
// compaction, spawn, next frame's predict list.  `scratch` = 2048 + 16 + 4 ints of LDS.  Called by the stand-alone
// dl_lifecycle_kernel (mot_devloop.hip) and by the tail of the Munkres kernel (assoc_kernels.hip), which saves a launch.
__device__ inline void dl_lifecycle_body(const DLState& S, const KcfPool& kp, const KalmanPool& kal, const bbox_t* trk_pred,
                                         const bbox_t* dets, int nD, const int* assignment, int* scratch)
{
    int* at = scratch; int* ad = scratch + 1024;
    int* wave_tot = scratch + 2048;
    int* cnt = scratch + 2048 + 16;   // [0] update list, [1] free stack top
    int* cntc = scratch + 2048 + 16 + 4;  // size classes: per-class free stack tops
    int* load = scratch + 2048 + 16 + 4 + DL_MAX_CLASSES;   // sharding: live tracks per rank (spawn assignment)
    const bool sharded = S.world > 1;
    const int t = threadIdx.x;
    const int nT = *S.nlive;
    const bool multi = S.kind == MOT_TRACKER_KCF && S.ncls > 1;
    if (t == 0) { cnt[0] = 0; cnt[1] = *S.nfree; }
    if (multi && t < S.ncls) cntc[t] = S.nfree_c[t];
    if (t < DL_MAX_WORLD) load[t] = 0;
    at[t] = -1; ad[t] = -1;
    __syncthreads();
    // td.cpp:472-502 -- scatter of the assignment vector (rows = the smaller side, td.cpp:462-469)
    if (nT > 0 && nD > 0) {
        if (nT < nD) { if (t < nT) { const int j = assignment[t]; at[t] = j; if (j >= 0) ad[j] = t; } }
        else { if (t < nD) { const int i = assignment[t]; if (i >= 0) at[i] = t; ad[t] = i; } }
    }
    __syncthreads();
    // td.cpp:512-582 (counters, update box) and :585-609 (lost rule)
    int slot = -1, age = 0, vis = 0, inv = 0, tcls = 0; unsigned tid = 0; bbox_t bb{}; bool keep = false, mine = false;
    // Round 5: the update list and the free-slot stack are filled in LIVE ORDER (block scans instead of atomic counters) -- which slot a new
    // track receives and where an item sits in the residual-update list no longer depend on the order in which wavefronts reach an atomic,
    // so two runs of a stream leave the same bits in device memory (state dumps can be compared; results never depended on it).
    bool to_upd = false, to_free = false; int j_upd = -1; int own = 0;
    if (t < nT) {
        slot = S.slot[t]; tid = S.tid[t]; age = S.age[t]; vis = S.vis[t]; inv = S.inv[t];
        if (sharded) own = S.owner[t];
        if (multi) tcls = S.cls[t];
        bb = trk_pred[t];
        const int j = at[t];
        if (j >= 0) { bb = dets[j]; vis++; age++; inv = 0; }
        else { age++; inv++; }
        const bool lost = ((age < 10) && (vis * 5 < 3 * age)) || (inv >= 20);
        keep = !lost;
        mine = own == S.rank;
        if (sharded && keep) atomicAdd(&load[own], 1);
        if (mine) {
            if (keep && S.defer && j >= 0) {                               // tracker_update's bookkeeping now, its model blend in the next predict
                S.pend_det[slot] = j;
                kp.pos[slot] = bb;
                kp.scale[slot] = make_float2(((float)(bb.r - bb.l + 1)) / ((float)kp.cols), ((float)(bb.b - bb.t + 1)) / ((float)kp.rows));
            }
            else if (keep) { to_upd = true; j_upd = j; }
            else if (multi) { const int q = atomicAdd(&cntc[tcls], 1); S.free_c[(size_t)tcls * S.cap + q] = slot; }
            else to_free = true;                                           // tracker_delete (td.cpp:599)
        }
    }
    {
        int n_upd, n_free;
        const int upos = block_excl_scan_flag(to_upd, wave_tot, n_upd);
        const int fpos = block_excl_scan_flag(to_free, wave_tot, n_free);
        const int free0 = cnt[1];                                          // (written before the first barrier of this function; rewritten behind the next one)
        if (to_upd) { S.upd_slots[upos] = slot; S.upd_boxes[upos] = bb; S.upd_det[upos] = j_upd; if (multi) S.upd_cls[upos] = tcls; }
        if (to_free) S.free_slots[free0 + fpos] = slot;
        __syncthreads();
        if (t == 0) { cnt[0] = n_upd; cnt[1] = free0 + n_free; }
    }
    int n_keep;
    const int newpos = block_excl_scan_flag(keep, wave_tot, n_keep);
    __syncthreads();
    if (keep) { S.slot[newpos] = slot; S.tid[newpos] = tid; S.age[newpos] = age; S.vis[newpos] = vis; S.inv[newpos] = inv; S.bbox[newpos] = bb; if (multi) S.cls[newpos] = tcls; if (sharded) S.owner[newpos] = own; }
    // td.cpp:612-644 -- spawn a tracker per unassigned detection, in detection order
    bool spawn = false; bbox_t db{}; int scls = 0;
    if (t < nD && ad[t] < 0) {
        db = dets[t];
        spawn = true;
        const int drows = db.b - db.t + 1, dcols = db.r - db.l + 1;
        if (multi) { scls = drows - S.cls_lo; if (drows != dcols || scls < 0 || scls >= S.ncls) { spawn = false; atomicAdd(&S.err[0], 1); } }   // no class for this template size
        else if (S.kind == MOT_TRACKER_KCF && (drows != S.rows || dcols != S.cols)) { spawn = false; atomicAdd(&S.err[0], 1); }
    }
    int n_spawn;
    const int spos = block_excl_scan_flag(spawn, wave_tot, n_spawn);
    const unsigned tid0 = *S.next_tid;
    // slots for the spawning tracks this rank owns, in detection order from the top of the stack (single template; the per-class stacks of
    // the size classes keep their atomic tops)
    const bool in_cap = spawn && n_keep + spos < S.cap;
    // owners of the spawning tracks, in detection order: each goes to the rank with the fewest live tracks so far (lowest rank on a tie).
    // Wavefront 0, lane = rank: one wave-wide arg-min per spawn (frame 0 spawns everything: ~0.1 ms once; a steady frame spawns a handful).
    int* sp_owner = at;                                                // the assignment scatter is spent (read above, behind a barrier)
    if (sharded) {
        __syncthreads();                                               // load[] complete, at[] dead
        if (t < 64) {
            const int n_sp = min(n_spawn, S.cap - n_keep);
            unsigned ld = (t < S.world) ? (unsigned)load[t] : 0xFFFFFFu;
            for (int q = 0; q < n_sp; q++) {
                unsigned key = (ld << 8) | (unsigned)t;                    // (load, rank): the smallest wins
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)key, off); key = o < key ? o : key; }
                const int win = (int)(key & 0xFFu);
                if (t == win) ld++;
                if (t == 0) sp_owner[q] = win;
            }
        }
        __syncthreads();
    }
    const int own_new = (sharded && in_cap) ? sp_owner[spos] : 0;
    const bool m2 = in_cap && own_new == S.rank;
    int n_pop;
    const int ppos = block_excl_scan_flag(m2 && !multi, wave_tot, n_pop);
    const int top0 = cnt[1];
    __syncthreads();
    if (spawn) {
        const int idx = n_keep + spos;
        if (idx < S.cap) {
            const unsigned ntid = tid0 + (unsigned)spos;
            int ns = -1;
            if (m2) {
                if (multi) { const int top = atomicSub(&cntc[scls], 1) - 1; if (top >= 0) ns = S.free_c[(size_t)scls * S.cap + top]; else atomicAdd(&S.err[2], 1); }
                else { const int top = top0 - 1 - ppos; if (top >= 0) ns = S.free_slots[top]; else atomicAdd(&S.err[2], 1); }
            }
            S.slot[idx] = ns; S.tid[idx] = ntid; S.age[idx] = 0; S.vis[idx] = 0; S.inv[idx] = 0; S.bbox[idx] = db;
            if (multi) S.cls[idx] = scls;
            if (sharded) S.owner[idx] = own_new;
            if (ns >= 0) {
                if (S.kind == MOT_TRACKER_KCF) {
                    const KcfPool& sp = multi ? S.pools[scls] : kp;
                    sp.pos[ns] = db; sp.scale[ns] = make_float2(1.f, 1.f); sp.first_update[ns] = 1;     // kcf.cpp:200-210
                    if (S.defer) S.pend_det[ns] = t;                                   // first update (eta = 1, td.cpp:631-640) in the next predict
                    else { const int q = atomicAdd(&cnt[0], 1); S.upd_slots[q] = ns; S.upd_boxes[q] = db; S.upd_det[q] = t; if (multi) S.upd_cls[q] = scls; }   // first update, td.cpp:631-640
                } else {
                    const double v[6] = { (double)db.l, (double)db.t, (double)db.r, (double)db.b, 0.0, 0.0 }; // kalman.cpp:152-157
                    for (int q = 0; q < 6; q++) kal.x[(size_t)ns * 6 + q] = v[q];
                    for (int q = 0; q < 36; q++) kal.P[(size_t)ns * 36 + q] = (q % 6 == q / 6) ? 1e+4 : 0.0;
                }
            }
        } else atomicAdd(&S.err[1], 1);
    }
    __syncthreads();
    if (t == 0 && !multi) cnt[1] = top0 - n_pop;
    __syncthreads();
    int n_new = n_keep + n_spawn; if (n_new > S.cap) n_new = S.cap;
    if (t == 0) {
        *S.nlive = n_new; *S.next_tid = tid0 + (unsigned)(n_new - n_keep);
        *S.upd_count = cnt[0]; *S.nfree = max(cnt[1], 0);
    }
    if (multi && t < S.ncls) S.nfree_c[t] = max(cntc[t], 0);
    __threadfence_block();
    __syncthreads();
    // lists for the next frame's predict
    dl_build_lists(S, n_new, wave_tot);
}


// arguments of the lifecycle step when it runs as the tail of the Munkres kernel
struct LifeArgs {
    int enabled;
    DLState S; KcfPool kp; KalmanPool kal;
    const bbox_t* trk_pred; const bbox_t* dets; int nD;
    LapProv prov;             // provisional commits of two-row tie frames (mot_dev.h)
};
