// mot_devloop.hip -- device-resident tracker-thread iteration (top/td.cpp:344-644).
//
// The whole per-frame loop is enqueued on the context's stream without any host
// synchronisation or copy: the live-track list (tracker_info[] of td.cpp:312),
// its counters and the lifecycle rules live in HBM and are advanced by one
// single-workgroup kernel per frame.  Per frame (one GPU):
//     kcf_predict (1 WG/track) -> assoc_min -> assoc_sub -> munkres (1 WG)
//     -> dl_lifecycle (1 WG) -> kcf_update (1 WG/track)
// With world > 1 the predicted boxes of the local shard are written into this
// rank's segment of an all-gather buffer (mot_step_begin_device), the caller
// runs ONE ncclAllGather, and mot_step_finish_device continues; association and
// lifecycle are replicated (deterministic) on every rank.
#include "mot_ctx.h"

using namespace mot_impl;

namespace {

// split update: detection features start beside the predict when predict + feature workgroups fit the chip at 2 per CU
#define MOT_SPLIT_EARLY_MAX 512
#define MOT_SPLIT_EXCL_MAX 256    // ... and one per CU (no sharing with predict workgroups) while they all fit that way

struct DLState {
    int* nlive; unsigned* next_tid; int* nfree; int* free_slots;
    int* slot; unsigned* tid; int* age; int* vis; int* inv; bbox_t* bbox;   // [cap] live list, td.cpp order
    int* rankpos;                 // [cap] index inside the owner's all-gather segment
    int* loc_slots; int* loc_count;
    int* upd_slots; bbox_t* upd_boxes; int* upd_count;
    int* upd_det;                 // [cap + max_dets] detection whose box an update item adopts (-1: the predicted box, td.cpp:540-560)
    bbox_t* pred;                 // [cap] predicted boxes in live order
    bbox_t* gather;               // [world*spr] all-gather buffer (own segment written by predict)
    int* err;                     // [4]: spawns dropped for template-size mismatch, capacity drops, pool exhausted, -
    int cap, max_dets, rank, world, spr, rows, cols, kind;
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
__device__ void dl_build_lists(const DLState& S, int n, int* wave_tot)
{
    const int t = threadIdx.x;
    const unsigned mytid = (t < n) ? S.tid[t] : 0u;
    const int r = (t < n) ? (int)(mytid % (unsigned)S.world) : -1;
    int mine_total = 0;
    for (int rk = 0; rk < S.world; rk++) {
        int total;
        const int pos = block_excl_scan_flag(r == rk, wave_tot, total);
        if (r == rk) {
            S.rankpos[t] = pos;
            if (rk == S.rank) {
                S.loc_slots[pos] = S.slot[t];
                if (S.kind == MOT_TRACKER_KALMAN) S.gather[(size_t)S.rank * S.spr + pos] = S.bbox[t];   // predict is in/out (kalman.cpp:112-115)
            }
        }
        if (rk == S.rank) mine_total = total;
    }
    if (t == 0) *S.loc_count = mine_total;
}

__global__ void __launch_bounds__(1024) dl_scatter_kernel(DLState S, const bbox_t* gathered)
{   // gathered segments -> live order (world > 1)
    const int n = *S.nlive, t = threadIdx.x;
    if (t < n) {
        const int r = (int)(S.tid[t] % (unsigned)S.world);
        S.pred[t] = gathered[(size_t)r * S.spr + S.rankpos[t]];
    }
}

__global__ void __launch_bounds__(1024) dl_lifecycle_kernel(DLState S, KcfPool kp, KalmanPool kal, const bbox_t* trk_pred,
                                                            const bbox_t* dets, int nD, const int* assignment)
{
    __shared__ int at[1024], ad[1024];
    __shared__ int wave_tot[16];
    __shared__ int cnt[4];            // [0] update list, [1] free stack top
    const int t = threadIdx.x;
    const int nT = *S.nlive;
    if (t == 0) { cnt[0] = 0; cnt[1] = *S.nfree; }
    at[t] = -1; ad[t] = -1;
    __syncthreads();
    // td.cpp:472-502 -- scatter of the assignment vector (rows = the smaller side, td.cpp:462-469)
    if (nT > 0 && nD > 0) {
        if (nT < nD) { if (t < nT) { const int j = assignment[t]; at[t] = j; if (j >= 0) ad[j] = t; } }
        else { if (t < nD) { const int i = assignment[t]; if (i >= 0) at[i] = t; ad[t] = i; } }
    }
    __syncthreads();
    // td.cpp:512-582 (counters, update box) and :585-609 (lost rule)
    int slot = -1, age = 0, vis = 0, inv = 0; unsigned tid = 0; bbox_t bb{}; bool keep = false, mine = false;
    if (t < nT) {
        slot = S.slot[t]; tid = S.tid[t]; age = S.age[t]; vis = S.vis[t]; inv = S.inv[t];
        bb = trk_pred[t];
        const int j = at[t];
        if (j >= 0) { bb = dets[j]; vis++; age++; inv = 0; }
        else { age++; inv++; }
        const bool lost = ((age < 10) && (vis * 5 < 3 * age)) || (inv >= 20);
        keep = !lost;
        mine = ((int)(tid % (unsigned)S.world) == S.rank);
        if (mine) {
            if (keep) { const int q = atomicAdd(&cnt[0], 1); S.upd_slots[q] = slot; S.upd_boxes[q] = bb; S.upd_det[q] = j; }
            else { const int q = atomicAdd(&cnt[1], 1); S.free_slots[q] = slot; }   // tracker_delete (td.cpp:599)
        }
    }
    int n_keep;
    const int newpos = block_excl_scan_flag(keep, wave_tot, n_keep);
    __syncthreads();
    if (keep) { S.slot[newpos] = slot; S.tid[newpos] = tid; S.age[newpos] = age; S.vis[newpos] = vis; S.inv[newpos] = inv; S.bbox[newpos] = bb; }
    // td.cpp:612-644 -- spawn a tracker per unassigned detection, in detection order
    bool spawn = false; bbox_t db{};
    if (t < nD && ad[t] < 0) {
        db = dets[t];
        spawn = true;
        if (S.kind == MOT_TRACKER_KCF && ((db.b - db.t + 1) != S.rows || (db.r - db.l + 1) != S.cols)) { spawn = false; atomicAdd(&S.err[0], 1); }
    }
    int n_spawn;
    const int spos = block_excl_scan_flag(spawn, wave_tot, n_spawn);
    const unsigned tid0 = *S.next_tid;
    __syncthreads();
    if (spawn) {
        const int idx = n_keep + spos;
        if (idx < S.cap) {
            const unsigned ntid = tid0 + (unsigned)spos;
            const bool m2 = ((int)(ntid % (unsigned)S.world) == S.rank);
            int ns = -1;
            if (m2) {
                const int top = atomicSub(&cnt[1], 1) - 1;
                if (top >= 0) ns = S.free_slots[top]; else atomicAdd(&S.err[2], 1);
            }
            S.slot[idx] = ns; S.tid[idx] = ntid; S.age[idx] = 0; S.vis[idx] = 0; S.inv[idx] = 0; S.bbox[idx] = db;
            if (ns >= 0) {
                if (S.kind == MOT_TRACKER_KCF) {
                    kp.pos[ns] = db; kp.scale[ns] = make_float2(1.f, 1.f); kp.first_update[ns] = 1;     // kcf.cpp:200-210
                    const int q = atomicAdd(&cnt[0], 1); S.upd_slots[q] = ns; S.upd_boxes[q] = db; S.upd_det[q] = t;   // first update, td.cpp:631-640
                } else {
                    const double v[6] = { (double)db.l, (double)db.t, (double)db.r, (double)db.b, 0.0, 0.0 }; // kalman.cpp:152-157
                    for (int q = 0; q < 6; q++) kal.x[(size_t)ns * 6 + q] = v[q];
                    for (int q = 0; q < 36; q++) kal.P[(size_t)ns * 36 + q] = (q % 6 == q / 6) ? 1e+4 : 0.0;
                }
            }
        } else atomicAdd(&S.err[1], 1);
    }
    __syncthreads();
    int n_new = n_keep + n_spawn; if (n_new > S.cap) n_new = S.cap;
    if (t == 0) {
        *S.nlive = n_new; *S.next_tid = tid0 + (unsigned)(n_new - n_keep);
        *S.upd_count = cnt[0]; *S.nfree = max(cnt[1], 0);
    }
    __threadfence_block();
    __syncthreads();
    // lists for the next frame's predict
    dl_build_lists(S, n_new, wave_tot);
}

} // namespace

namespace mot_impl {

struct DevLoop {
    DLState S{};
    int pool = -1;
    DevBuf<int> ints; DevBuf<bbox_t> boxes; DevBuf<unsigned> tids;
    bool begun = false; const void* frame = nullptr;
    hipEvent_t ev[8]{}; bool ev_ok = false;
    // split update (KCF): the spectra of all detection boxes are computed on a second, low-priority stream while the
    // association runs; the per-track update then only blends them into the model
    DevBuf<float2> det_spec; hipStream_t side = nullptr; hipEvent_t ev_mid = nullptr, ev_feat = nullptr, ev_upd = nullptr; bool split = false;
    bool feat_early = false;      // this frame's detection features were launched at the start of the frame
};

void devloop_destroy(DevLoop* d)
{
    if (!d) return;
    if (d->ev_ok) for (hipEvent_t e : d->ev) (void)hipEventDestroy(e);
    if (d->ev_mid) (void)hipEventDestroy(d->ev_mid);
    if (d->ev_feat) (void)hipEventDestroy(d->ev_feat);
    if (d->ev_upd) (void)hipEventDestroy(d->ev_upd);
    if (d->side) (void)hipStreamDestroy(d->side);
    delete d;
}

} // namespace mot_impl

namespace {

int devloop_get(mot_ctx* c, DevLoop** out)
{
    if (c->devloop) { *out = c->devloop; return MOT_OK; }
    if (!c->live.empty() || !c->tracks.empty()) return fail(MOT_ERR_STATE, "context already used in host-orchestrated mode; device-resident loop needs a fresh context");
    std::unique_ptr<DevLoop> d(new DevLoop);
    DLState& S = d->S;
    const int cap = c->cfg.max_tracks, md = c->cfg.max_dets;
    S.cap = cap; S.max_dets = md; S.rank = c->cfg.rank; S.world = c->cfg.world; S.spr = c->slots_per_rank; S.kind = c->cfg.tracker_kind;
    S.rows = c->cfg.dev_rows > 0 ? c->cfg.dev_rows : 80; S.cols = c->cfg.dev_cols > 0 ? c->cfg.dev_cols : 80;
    if (S.kind == MOT_TRACKER_KCF) { int rc = get_pool(c, S.rows, S.cols, &d->pool); if (rc) return rc; }
    // one int arena: nlive, next_tid(as tids), nfree, loc_count, upd_count, err[4], then arrays
    const size_t nints = 16 + (size_t)cap * 8 + 2 * (size_t)(cap + md) + 64;
    HIPCHK(d->ints.alloc(nints)); HIPCHK(hipMemsetAsync(d->ints.p, 0, nints * sizeof(int), c->stream));
    HIPCHK(d->tids.alloc((size_t)cap + 4)); HIPCHK(hipMemsetAsync(d->tids.p, 0, (cap + 4) * sizeof(unsigned), c->stream));
    HIPCHK(d->boxes.alloc((size_t)cap * 2 + cap + md + 8)); HIPCHK(hipMemsetAsync(d->boxes.p, 0, d->boxes.n * sizeof(bbox_t), c->stream));
    int* ip = d->ints.p;
    S.nlive = ip; S.nfree = ip + 1; S.loc_count = ip + 2; S.upd_count = ip + 3; S.err = ip + 4; ip += 16;
    S.free_slots = ip; ip += cap; S.slot = ip; ip += cap; S.age = ip; ip += cap; S.vis = ip; ip += cap; S.inv = ip; ip += cap;
    S.rankpos = ip; ip += cap; S.loc_slots = ip; ip += cap; S.upd_slots = ip; ip += cap + md; S.upd_det = ip; ip += cap + md;
    S.next_tid = d->tids.p; S.tid = d->tids.p + 4;
    S.bbox = d->boxes.p; S.pred = S.bbox + cap; S.upd_boxes = S.pred + cap;
    S.gather = c->d_gather.p;
    // free slot stack: all slots of the pool
    std::vector<int> fs(cap);
    for (int i = 0; i < cap; i++) fs[i] = cap - 1 - i;
    HIPCHK(hipMemcpyAsync(S.free_slots, fs.data(), sizeof(int) * cap, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(S.nfree, &cap, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (S.kind == MOT_TRACKER_KCF) {
        c->pools[d->pool]->free_slots.clear();                           // the device owns the pool now
        const char* ev = getenv("MOT_SPLIT_UPDATE");                     // default on; 0 keeps the fused update kernel
        if (!ev || atoi(ev) != 0) {
            const KcfPool& kp = c->pools[d->pool]->dev;
            int lo = 0, hi = 0;
            HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));           // lo = numerically largest = lowest priority
            HIPCHK(hipStreamCreateWithPriority(&d->side, hipStreamNonBlocking, lo));
            HIPCHK(hipEventCreateWithFlags(&d->ev_mid, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&d->ev_feat, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&d->ev_upd, hipEventDisableTiming));
            HIPCHK(hipEventRecord(d->ev_upd, c->stream));
            HIPCHK(d->det_spec.alloc((size_t)md * MOT_NCHAN * kp.nbins));
            d->split = true;
        }
    }
    else c->kal_free.clear();
    c->devloop = d.release();
    *out = c->devloop;
    return MOT_OK;
}

int dl_begin(mot_ctx* c, DevLoop* d, const void* frame_dev, hipEvent_t* ev, const void* dets_dev = nullptr, int nD = 0)
{
    DLState& S = d->S;
    if (d->begun) return fail(MOT_ERR_STATE, "mot_step_begin_device called twice");
    if (S.kind == MOT_TRACKER_KCF && !frame_dev) return fail(MOT_ERR_ARG, "null frame");
    d->frame = frame_dev;
    bbox_t* seg = S.gather + (size_t)S.rank * S.spr;
    if (ev) HIPCHK(hipEventRecord(ev[0], c->stream));
    d->feat_early = false;
    if (d->split && S.kind == MOT_TRACKER_KCF && dets_dev && nD > 0 && nD <= S.max_dets && S.spr + nD <= MOT_SPLIT_EARLY_MAX) {
        // small frames leave most CUs idle during the predict: the detection features run beside it (they only need the
        // frame and the boxes); the spectra buffer is free once the previous frame's update has finished
        HIPCHK(hipStreamWaitEvent(d->side, d->ev_upd, 0));
        KcfLaunch lf{}; lf.frame = (const uint8_t*)frame_dev; lf.boxes_in = (const bbox_t*)dets_dev; lf.spec_out = d->det_spec.p;
        HIPCHK(launch_kcf_update(c->pools[d->pool]->dev, lf, nD, d->side, S.spr + nD <= MOT_SPLIT_EXCL_MAX));   // own CUs beside the predict
        HIPCHK(hipEventRecord(d->ev_feat, d->side));
        d->feat_early = true;
    }
    if (S.kind == MOT_TRACKER_KCF) {
        KcfLaunch l{}; l.slots = S.loc_slots; l.count = S.loc_count; l.frame = (const uint8_t*)frame_dev; l.boxes_out = seg; l.clamp = 1; l.dbg = c->dbg_on ? c->dbg.p : nullptr;
        HIPCHK(launch_kcf_predict(c->pools[d->pool]->dev, l, S.spr, c->stream));
    } else HIPCHK(launch_kalman_predict(c->kal, S.loc_slots, S.loc_count, S.spr, seg, 1, c->stream));
    if (ev) HIPCHK(hipEventRecord(ev[1], c->stream));
    d->begun = true;
    return MOT_OK;
}

int dl_finish(mot_ctx* c, DevLoop* d, const void* gathered, const void* dets_dev, int nD, hipEvent_t* ev)
{
    DLState& S = d->S;
    if (!d->begun) return fail(MOT_ERR_STATE, "mot_step_finish_device without mot_step_begin_device");
    d->begun = false;
    if (nD < 0 || nD > S.max_dets || (nD && !dets_dev)) return fail(MOT_ERR_ARG, "bad detection list (%d, max %d)", nD, S.max_dets);
    const bbox_t* g = gathered ? (const bbox_t*)gathered : S.gather;
    const bbox_t* trk = g;
    if (S.world > 1) { hipLaunchKernelGGL(dl_scatter_kernel, dim3(1), dim3(1024), 0, c->stream, S, g); HIPCHK(hipGetLastError()); trk = S.pred; }
    const bbox_t* dets = (const bbox_t*)dets_dev;
    KcfPool kp{}; if (S.kind == MOT_TRACKER_KCF) kp = c->pools[d->pool]->dev;
    const bool split = d->split && S.kind == MOT_TRACKER_KCF && nD > 0;
    HIPCHK(launch_assoc(c->assoc, trk, S.nlive, S.cap, dets, nD, nullptr, 0, 0, 0, c->stream, (split && !d->feat_early) ? d->ev_mid : nullptr));
    if (split && !d->feat_early) {
        // features of every detection box, on the side stream, from the moment the Munkres kernel has been handed to the
        // dispatcher (so its 17 workgroups are placed first); the frame and the boxes are inputs of this call
        HIPCHK(hipStreamWaitEvent(d->side, d->ev_mid, 0));
        KcfLaunch lf{}; lf.frame = (const uint8_t*)d->frame; lf.boxes_in = dets; lf.spec_out = d->det_spec.p;
        HIPCHK(launch_kcf_update(kp, lf, nD, d->side));
        HIPCHK(hipEventRecord(d->ev_feat, d->side));
    }
    if (ev) HIPCHK(hipEventRecord(ev[2], c->stream));
    hipLaunchKernelGGL(dl_lifecycle_kernel, dim3(1), dim3(1024), 0, c->stream, S, kp, c->kal, trk, dets, nD, c->assoc.assignment);
    HIPCHK(hipGetLastError());
    if (ev) HIPCHK(hipEventRecord(ev[3], c->stream));
    const int upd_max = S.spr + nD;
    if (S.kind == MOT_TRACKER_KCF) {
        KcfLaunch l{}; l.slots = S.upd_slots; l.count = S.upd_count; l.frame = (const uint8_t*)d->frame; l.boxes_in = S.upd_boxes; l.dbg = c->dbg_on ? c->dbg.p + 16 : nullptr;
        if (split) { HIPCHK(hipStreamWaitEvent(c->stream, d->ev_feat, 0)); l.det_spec = d->det_spec.p; l.det_index = S.upd_det; }
        HIPCHK(launch_kcf_update(kp, l, upd_max, c->stream));
        if (split) HIPCHK(hipEventRecord(d->ev_upd, c->stream));
    } else HIPCHK(launch_kalman_update(c->kal, S.upd_slots, S.upd_count, upd_max, S.upd_boxes, c->stream));
    if (ev) HIPCHK(hipEventRecord(ev[4], c->stream));
    return MOT_OK;
}

} // namespace

extern "C" {

int mot_step_begin_device(mot_ctx* c, const void* frame_dev, void** local_boxes_dev, int* slots_per_rank)
{
    if (!c) return fail(MOT_ERR_ARG, "null ctx");
    int rc = ensure_device(c); if (rc) return rc;
    DevLoop* d; rc = devloop_get(c, &d); if (rc) return rc;
    rc = dl_begin(c, d, frame_dev, nullptr); if (rc) return rc;
    if (local_boxes_dev) *local_boxes_dev = d->S.gather + (size_t)d->S.rank * d->S.spr;
    if (slots_per_rank) *slots_per_rank = d->S.spr;
    return MOT_OK;
}

int mot_step_finish_device(mot_ctx* c, const void* gathered_boxes_dev, const void* dets_dev, int nD)
{
    if (!c || !c->devloop) return fail(MOT_ERR_STATE, "mot_step_finish_device without mot_step_begin_device");
    int rc = ensure_device(c); if (rc) return rc;
    return dl_finish(c, c->devloop, gathered_boxes_dev, dets_dev, nD, nullptr);
}

int mot_step_frame_device(mot_ctx* c, const void* frame_dev, const void* dets_dev, int nD)
{
    if (!c) return fail(MOT_ERR_ARG, "null ctx");
    if (c->cfg.world != 1) return fail(MOT_ERR_STATE, "sharded context: use mot_step_begin_device / all-gather / mot_step_finish_device");
    int rc = ensure_device(c); if (rc) return rc;
    DevLoop* d; rc = devloop_get(c, &d); if (rc) return rc;
    rc = dl_begin(c, d, frame_dev, nullptr, dets_dev, nD); if (rc) return rc;
    return dl_finish(c, d, nullptr, dets_dev, nD, nullptr);
}

int mot_profile_frame_device(mot_ctx* c, const void* frame_dev, const void* dets_dev, int nD, float* stage_ms5)
{
    if (!c || !stage_ms5) return fail(MOT_ERR_ARG, "null argument");
    if (c->cfg.world != 1) return fail(MOT_ERR_STATE, "profile on an unsharded context");
    int rc = ensure_device(c); if (rc) return rc;
    DevLoop* d; rc = devloop_get(c, &d); if (rc) return rc;
    if (!d->ev_ok) { for (int i = 0; i < 8; i++) HIPCHK(hipEventCreate(&d->ev[i])); d->ev_ok = true; }
    rc = dl_begin(c, d, frame_dev, d->ev, dets_dev, nD); if (rc) return rc;
    rc = dl_finish(c, d, nullptr, dets_dev, nD, d->ev); if (rc) return rc;
    HIPCHK(hipEventSynchronize(d->ev[4]));
    // [0] predict  [1] cost/min/sub + munkres (see note)  [2] munkres -- reported together in [1], [2]=0  [3] lifecycle  [4] update
    float ms;
    HIPCHK(hipEventElapsedTime(&ms, d->ev[0], d->ev[1])); stage_ms5[0] = ms;
    HIPCHK(hipEventElapsedTime(&ms, d->ev[1], d->ev[2])); stage_ms5[1] = ms; stage_ms5[2] = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, d->ev[2], d->ev[3])); stage_ms5[3] = ms;
    HIPCHK(hipEventElapsedTime(&ms, d->ev[3], d->ev[4])); stage_ms5[4] = ms;
    return MOT_OK;
}

int mot_live_count(mot_ctx* c, int* n_live)
{
    if (!c || !n_live) return fail(MOT_ERR_ARG, "null argument");
    if (!c->devloop) { *n_live = (int)c->live.size(); return MOT_OK; }
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(n_live, c->devloop->S.nlive, sizeof(int), hipMemcpyDeviceToHost));
    return MOT_OK;
}

int mot_live_tracks(mot_ctx* c, bbox_t* boxes, unsigned* tids, int* ages, int* n_live)
{
    if (!c) return fail(MOT_ERR_ARG, "null ctx");
    if (!c->devloop) {
        for (size_t i = 0; i < c->live.size(); i++) { if (boxes) boxes[i] = c->live[i].bbox; if (tids) tids[i] = c->live[i].tid; if (ages) ages[i] = c->live[i].age; }
        if (n_live) *n_live = (int)c->live.size();
        return MOT_OK;
    }
    const DLState& S = c->devloop->S;
    int n = 0;
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(&n, S.nlive, sizeof(int), hipMemcpyDeviceToHost));
    if (n_live) *n_live = n;
    if (n > 0) {
        if (boxes) HIPCHK(hipMemcpy(boxes, S.bbox, sizeof(bbox_t) * n, hipMemcpyDeviceToHost));
        if (tids) HIPCHK(hipMemcpy(tids, S.tid, sizeof(unsigned) * n, hipMemcpyDeviceToHost));
        if (ages) HIPCHK(hipMemcpy(ages, S.age, sizeof(int) * n, hipMemcpyDeviceToHost));
    }
    return MOT_OK;
}

} // extern "C"
