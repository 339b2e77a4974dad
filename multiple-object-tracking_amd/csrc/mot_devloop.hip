// mot_devloop.hip -- device-resident tracker-thread iteration (top/td.cpp:344-644).
//
// The whole per-frame loop is enqueued on the context's stream without any host
// synchronisation or copy: the live-track list (tracker_info[] of td.cpp:312),
// its counters and the lifecycle rules live in HBM and are advanced by one
// single-workgroup kernel per frame.  Per frame (one GPU):
//     kcf_predict (1 WG/track) -> assoc_min -> assoc_sub -> munkres (+ the lifecycle step as its tail,
//     dl_lifecycle.h) -> kcf_update (1 WG/track; split into detection features + blend, see DevLoop)
// With world > 1 the predicted boxes of the local shard are written into this
// rank's segment of an all-gather buffer (mot_step_begin_device), the caller
// runs ONE ncclAllGather, and mot_step_finish_device continues; association and
// lifecycle are replicated (deterministic) on every rank.
#include "mot_ctx.h"
#include "mot_env.h"
#include "dl_lifecycle.h"
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>

// Events that order work between this context's streams on ONE device: no timing, and a DEVICE-scope release when recorded.  The
// default (system-scope) release writes back and invalidates the caches at every record; the consumers here are kernels of the same
// device.  (Timeline: the gap in front of the predict launch went from 10.5 to 5.9 us; the bench moved within its noise.  The two
// sync packets of a frame -- this wait and the record in front of the row scan -- still cost ~13 us together: the fused-update
// variant, which has neither, shows no gap at all between its kernels.)
// (investigation, MOT_EVENT_SYSTEM=1: the default system-scope release instead -- round 5's hunt for a stale cross-stream read)
static unsigned mot_event_flags() { static const unsigned f = (getenv("MOT_EVENT_SYSTEM") && atoi(getenv("MOT_EVENT_SYSTEM"))) ? hipEventDisableTiming : (hipEventDisableTiming | hipEventReleaseToDevice); return f; }
#define MOT_EVENT_FLAGS (mot_event_flags())

using namespace mot_impl;

namespace {

// split update: detection features start beside the predict when predict + feature workgroups fit the chip at 2 per CU
#define MOT_SPLIT_EARLY_MAX 512
#define MOT_SPLIT_EXCL_MAX 256    // ... and one per CU (no sharing with predict workgroups) while they all fit that way

// frame upload as a kernel: pinned host memory is mapped into the device's address space, so the copy stream can pull the frame over
// PCIe with plain 16-byte loads.  (hipMemcpyAsync on a second stream stalled the HOST for ~6.5 ms every 5-15 calls on this stack --
// tools/h2d_frame_probe.py -- which made the host-fed loop 3-10x slower than the resident one; a kernel has no such hiccup.)
__global__ void __launch_bounds__(256) h2d_copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16,
                                                       uint2* __restrict__ dst2, const uint2* __restrict__ src2, size_t n8)
{   // frame: 16-byte words; boxes: 8-byte words (24 bytes each: never a byte beyond the caller's array is read)
    // a small grid (the kernels of the frame in flight keep their CUs) with four loads in flight per thread: ~0.5 MB outstanding, enough
    // for the link's bandwidth-delay product
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], e = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = e;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) dst2[i] = src2[i];
}

__global__ void __launch_bounds__(1024) dl_scatter_kernel(DLState S, const bbox_t* gathered)
{   // gathered segments -> live order (world > 1)
    const int n = *S.nlive, t = threadIdx.x;
    if (t < n) {
        const int r = S.owner[t];
        S.pred[t] = gathered[(size_t)r * S.spr + S.rankpos[t]];
    }
}

} // namespace

namespace mot_impl {

// Auxiliary streams are shared by all device loops of a process on one device (round 6).  Every HIP stream beyond the first few is another
// hardware queue, and this stack slows down sharply once a process keeps more than about five of them busy-or-idle on a device (measured:
// bench.py's host-fed window ran at 1.7 instead of 2.65 M updates/s when the second context brought the count to seven; tools/hostfed_probe.py) --
// so contexts do not create their own side / emulation streams.  Sharing only ADDS ordering between contexts (their launches serialise on the
// shared stream), never removes one: each context still orders its own work with its own events.
struct AuxStreams { int dev; int reserve; hipStream_t side = nullptr; hipStream_t emu = nullptr; int refs = 0; };
static std::mutex g_aux_mu;
static std::vector<AuxStreams> g_aux;

static hipError_t aux_acquire(int dev, int reserve, bool want_emu, hipStream_t* side, hipStream_t* emu)
{
    std::lock_guard<std::mutex> lock(g_aux_mu);
    AuxStreams* a = nullptr;
    for (AuxStreams& x : g_aux) if (x.dev == dev && x.reserve == reserve) a = &x;
    if (!a) { g_aux.push_back(AuxStreams{dev, reserve}); a = &g_aux.back(); }
    if (!a->side) {
        // The feature launch gets a stream whose kernels may not use `reserve` (default 32) of the chip's CUs, so the one-workgroup kernels and the
        // short dense passes of the association chain on the main stream never queue behind detection-feature workgroups (2.44 -> 2.58 M updates/s
        // at 1024 tracks).  0 (or a refusal by the runtime): a low-priority stream over the whole chip, as in round 1.
        bool masked = false;
        if (reserve > 0 && reserve < 256) {
            uint32_t mask[8];
            for (int w = 0; w < 8; w++) mask[w] = 0xFFFFFFFFu;
            for (int b = 0; b < reserve; b++) mask[b >> 5] &= ~(1u << (b & 31));
            masked = hipExtStreamCreateWithCUMask(&a->side, 8, mask) == hipSuccess;
            if (!masked) { (void)hipGetLastError(); a->side = nullptr; }
        }
        if (!masked) {
            int lo = 0, hi = 0;
            hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi); if (e != hipSuccess) return e;   // lo = numerically largest = lowest priority
            e = hipStreamCreateWithPriority(&a->side, hipStreamNonBlocking, lo); if (e != hipSuccess) return e;
        }
    }
    if (want_emu && !a->emu) { hipError_t e = hipStreamCreateWithFlags(&a->emu, hipStreamNonBlocking); if (e != hipSuccess) return e; }
    a->refs++;
    *side = a->side; *emu = want_emu ? a->emu : nullptr;
    return hipSuccess;
}
static void aux_release(int dev, int reserve)
{
    std::lock_guard<std::mutex> lock(g_aux_mu);
    for (AuxStreams& x : g_aux) if (x.dev == dev && x.reserve == reserve && x.refs > 0 && --x.refs == 0) {
        if (x.side) { (void)hipStreamSynchronize(x.side); (void)hipStreamDestroy(x.side); x.side = nullptr; }
        if (x.emu) { (void)hipStreamSynchronize(x.emu); (void)hipStreamDestroy(x.emu); x.emu = nullptr; }
    }
}

struct DevLoop {
    DLState S{};
    int pool = -1;
    DevBuf<int> ints; DevBuf<bbox_t> boxes; DevBuf<unsigned> tids;
    // size classes: pool index of every class, device table of their descriptors, one shared HBM scratch, LDS need of a launch
    std::vector<int> cls_pool; DevBuf<KcfPool> pools_dev; DevBuf<float> shared_scratch; int slab_stride = 0; unsigned lds_bytes = 0; int r1_any = 0, gen_any = 0;
    bool begun = false; const void* frame = nullptr;
    hipEvent_t ev[8]{}; bool ev_ok = false;
    // split update (KCF): the spectra of all detection boxes are computed on a second, low-priority stream while the
    // association runs; the per-track update then only blends them into the model
    // det_spec: three spectra buffers in rotation -- the previous frame's (read by this frame's blend prologue), this frame's, and the
    // NEXT frame's when the caller passes it ahead (mot_step_frame_device_ahead): its detection features are then computed beside this
    // frame's association chain and are ready long before anything needs them.  ev_spec[b]: buffer b has been written (side stream);
    // spec_side[b]: ... by the side stream (else inside a main-stream launch: stream order suffices)
    DevBuf<float2> det_spec; DevBuf<int> pend; size_t spec_stride = 0; unsigned frame_no = 0; bool defer = false;
    int buf_prev = 0, buf_cur = 1; hipEvent_t ev_spec[3]{}; bool spec_side[3] = {false, false, false}; bool have_cur = false;
    const void* pf_frame = nullptr; const void* pf_dets = nullptr; int pf_nD = -1, pf_buf = -1; bool pf_valid = false;
    const void* next_frame = nullptr; const void* next_dets = nullptr; int next_nD = 0;
    const void* begin_dets = nullptr; int begin_nD = -1;             // the list dl_begin computed / adopted this frame's detection features for
    hipStream_t side = nullptr; hipEvent_t ev_mid = nullptr, ev_in = nullptr; bool split = false;
    // mot_step_frame_host: copy stream + two device buffers (frame, detections); up[b]: upload of buffer b done, done[b]: the frame that read it finished
    // host-fed loop (mot_step_frame_host): three device buffers for frame + boxes, an upload event per buffer, and a ring of the chain events of the last
    // three frames (ev_mid points at the current one); ev_mid0 is the event the device-resident calls use
    hipStream_t copy = nullptr; DevBuf<uint8_t> hbuf[3]; DevBuf<bbox_t> dbuf[3]; hipEvent_t ev_up[3]{}, ev_ring[3]{}, ev_mid0 = nullptr; unsigned host_no = 0; bool host_ok = false;
    bool feat_early = false;      // this frame's detection features were launched at the start of the frame
    bool feat_joined = false;     // ... inside the predict launch itself (no side stream, no event to wait for)
    bool mid_by_predict = false;   // this frame's predict launch carries ev_mid as its completion event
    bool want_mid = false, mid_valid = false;   // host-fed loop: every frame records ev_mid in its chain (the next call's feature launch is ordered behind it)
    // (debug) in-loop timing of the predict launch: pairs of events that receive the kernel's own begin / end stamps while the normal
    // step calls run (look-ahead, side stream and all) -- what rocprofv3 reports for the launch in the timed configuration
    std::vector<hipEvent_t> pt; int pt_used = 0;
    DevBuf<int> trace;            // (debug, MOT_TRACE=1) 32 frames x cap x 8 ints: predict records [0, 16), update records [16, 32) (KcfLaunch::trace)
    bool prof_two_call = false;   // (debug) mot_debug_profile_stages: the two-call / sharded step records its stage events (ev[0..5])
    // Provisional commits (round 6; mot_dev.h: ProvRec).  prov: this loop may commit two-row tie frames at once (single-template LDS-resident KCF,
    // deferred blend, one rank; MOT_PROV=0 switches it off).  Then: the sparse emulation runs on `emu` behind the row scan's event ev_rs; the
    // predicted boxes alternate between the two halves of `seg2` by frame parity and the row scan copies the detection list into a half of
    // `det_copy` (the emulation of frame f may still read both while predict(f + 1) writes and the caller reuses its list); patch_owed: a frame
    // has been associated since the last patch step -- the next predict launch is followed by one, every synchronisation point runs one first.
    bool prov = false; hipStream_t emu = nullptr; hipEvent_t ev_rs = nullptr;
    DevBuf<bbox_t> seg2, det_copy; DevBuf<ProvRec> prov_rec;
    bool patch_owed = false; unsigned seq_last = 0; int nD_last = 0, par_last = 0;
    int aux_dev = -1, aux_reserve = 0;   // key of the shared auxiliary streams this loop holds a reference to (aux_acquire)
};

void devloop_destroy(DevLoop* d)
{
    if (!d) return;
    // a detection-feature launch (side stream) or an upload (copy stream) may still be running: nothing is freed under them
    if (d->side) (void)hipStreamSynchronize(d->side);
    if (d->copy) (void)hipStreamSynchronize(d->copy);
    if (d->emu) (void)hipStreamSynchronize(d->emu);
    if (d->ev_rs) (void)hipEventDestroy(d->ev_rs);
    if (d->ev_ok) for (hipEvent_t e : d->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : d->pt) (void)hipEventDestroy(e);
    if (d->ev_in) (void)hipEventDestroy(d->ev_in);
    for (hipEvent_t e : d->ev_spec) if (e) (void)hipEventDestroy(e);
    if (d->aux_dev >= 0) aux_release(d->aux_dev, d->aux_reserve);    // the side / emulation streams are shared (AuxStreams): the last user destroys them
    if (d->host_ok) { d->ev_mid = d->ev_mid0; for (int b = 0; b < 3; b++) { (void)hipEventDestroy(d->ev_up[b]); (void)hipEventDestroy(d->ev_ring[b]); } (void)hipStreamDestroy(d->copy); }
    if (d->ev_mid) (void)hipEventDestroy(d->ev_mid);
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
    S.cap = cap; S.max_dets = md; S.rank = c->cfg.rank; S.world = c->cfg.world; S.kind = c->cfg.tracker_kind;
    // a rank never owns more than ceil(cap / world) tracks (balanced spawn assignment, DLState::owner): that is a segment of the all-gather
    S.spr = S.world > 1 ? (cap + S.world - 1) / S.world : c->slots_per_rank;
    if (S.world > DL_MAX_WORLD) return fail(MOT_ERR_ARG, "world %d exceeds %d ranks", S.world, DL_MAX_WORLD);
    S.rows = c->cfg.dev_rows > 0 ? c->cfg.dev_rows : 80; S.cols = c->cfg.dev_cols > 0 ? c->cfg.dev_cols : 80;
    S.ncls = 1; S.cls_lo = S.rows;
    if (S.kind == MOT_TRACKER_KCF && c->cfg.dev_size_lo > 0) {
        // per-track template sizes: one pool per square size; all descriptors in a device table, one HBM scratch for all
        S.cls_lo = c->cfg.dev_size_lo; S.ncls = c->cfg.dev_size_hi - c->cfg.dev_size_lo + 1;
        if (S.ncls == 1) S.rows = S.cols = S.cls_lo;                     // lo == hi: the single-template path, with THAT template (not dev_rows x dev_cols)
        std::vector<KcfPool> tab((size_t)S.ncls);
        size_t maxf = 0; unsigned maxlds = 0; int use0 = -1;
        for (int k = 0; k < S.ncls; k++) {
            int pi; int rc = get_pool(c, S.cls_lo + k, S.cls_lo + k, &pi, true); if (rc) return rc;
            d->cls_pool.push_back(pi);
            const KcfPool& kp = c->pools[pi]->dev;
            if (use0 < 0) use0 = kp.use_lds; else if (use0 != kp.use_lds) return fail(MOT_ERR_ARG, "template sizes %d..%d straddle the LDS-resident / HBM-slab split", c->cfg.dev_size_lo, c->cfg.dev_size_hi);
            maxf = std::max(maxf, (size_t)kp.lds_floats); maxlds = std::max(maxlds, (unsigned)kcf_lds_bytes(kp)); if (kp.r1_lds) d->r1_any = 1; if (kp.use_lds && !kp.fft20) d->gen_any = 1;
        }
        if (!use0) { HIPCHK(d->shared_scratch.alloc((size_t)(cap + md) * maxf)); for (int pi : d->cls_pool) c->pools[pi]->dev.gscratch = d->shared_scratch.p; }
        for (int k = 0; k < S.ncls; k++) tab[k] = c->pools[d->cls_pool[k]]->dev;
        HIPCHK(d->pools_dev.alloc((size_t)S.ncls)); HIPCHK(hipMemcpy(d->pools_dev.p, tab.data(), sizeof(KcfPool) * S.ncls, hipMemcpyHostToDevice));
        S.pools = d->pools_dev.p; d->slab_stride = (int)maxf; d->lds_bytes = maxlds; d->pool = d->cls_pool[0];
    } else if (S.kind == MOT_TRACKER_KCF) { int rc = get_pool(c, S.rows, S.cols, &d->pool); if (rc) return rc; }
    const bool multi = S.ncls > 1;
    // one int arena: nlive, next_tid(as tids), nfree, loc_count, upd_count, err[4], then arrays
    const size_t nints = 16 + (size_t)cap * 9 + MOT_SHADOW_SLOTS + 2 * (size_t)(cap + md) + 64 + (multi ? (size_t)cap * 2 + (cap + md) + S.ncls + (size_t)S.ncls * cap : 0);
    HIPCHK(d->ints.alloc(nints)); HIPCHK(hipMemsetAsync(d->ints.p, 0, nints * sizeof(int), c->stream));
    HIPCHK(d->tids.alloc((size_t)cap + 4)); HIPCHK(hipMemsetAsync(d->tids.p, 0, (cap + 4) * sizeof(unsigned), c->stream));
    HIPCHK(d->boxes.alloc((size_t)cap * 2 + cap + md + 8)); HIPCHK(hipMemsetAsync(d->boxes.p, 0, d->boxes.n * sizeof(bbox_t), c->stream));
    int* ip = d->ints.p;
    S.nlive = ip; S.nfree = ip + 1; S.loc_count = ip + 2; S.upd_count = ip + 3; S.err = ip + 4; ip += 16;
    S.free_slots = ip; ip += cap; S.slot = ip; ip += cap; S.age = ip; ip += cap; S.vis = ip; ip += cap; S.inv = ip; ip += cap;
    S.rankpos = ip; ip += cap; S.owner = ip; ip += cap; S.loc_slots = ip; ip += cap + MOT_SHADOW_SLOTS; S.upd_slots = ip; ip += cap + md; S.upd_det = ip; ip += cap + md;
    if (multi) { S.cls = ip; ip += cap; S.loc_cls = ip; ip += cap; S.upd_cls = ip; ip += cap + md; S.nfree_c = ip; ip += S.ncls; S.free_c = ip; ip += (size_t)S.ncls * cap; }
    S.next_tid = d->tids.p; S.tid = d->tids.p + 4;
    S.bbox = d->boxes.p; S.pred = S.bbox + cap; S.upd_boxes = S.pred + cap;
    S.gather = c->d_gather.p;
    // free slot stack: all slots of the pool
    std::vector<int> fs(cap);
    for (int i = 0; i < cap; i++) fs[i] = cap - 1 - i;
    HIPCHK(hipMemcpyAsync(S.free_slots, fs.data(), sizeof(int) * cap, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(S.nfree, &cap, sizeof(int), hipMemcpyHostToDevice, c->stream));
    std::vector<int> nfc;
    if (multi) {
        nfc.assign((size_t)S.ncls, cap);
        HIPCHK(hipMemcpyAsync(S.nfree_c, nfc.data(), sizeof(int) * S.ncls, hipMemcpyHostToDevice, c->stream));
        for (int k = 0; k < S.ncls; k++) HIPCHK(hipMemcpyAsync(S.free_c + (size_t)k * cap, fs.data(), sizeof(int) * cap, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    if (S.kind == MOT_TRACKER_KCF) {
        c->pools[d->pool]->free_slots.clear();                           // the device owns the pool now
        for (int pi : d->cls_pool) c->pools[pi]->free_slots.clear();
        const bool split_on = mot_impl::env().split_update != 0;          // MOT_SPLIT_UPDATE: default on; 0 keeps the fused update kernel
        // (size classes: the spectrum a detection is adopted with depends on the adopting track's template size, which is only
        // known after the assignment -- the fused update kernel is used)
        if (split_on && !multi) {
            const KcfPool& kp = c->pools[d->pool]->dev;
            const int rs = mot_impl::env().side_reserve;                    // MOT_SIDE_RESERVE (-1: by template size)
            // HBM-slab templates run one workgroup per CU for hundreds of microseconds: taking CUs away from them costs a
            // second round (256 tracks at 148 x 148: 322 k -> 240 k updates/s), so only LDS-resident templates reserve by default
            const int reserve = rs >= 0 ? rs : (kp.use_lds ? 32 : 0);
            // provisional commits of two-row tie frames (see DevLoop::prov): one rank, LDS-resident template (HBM-slab launches index their slabs by item)
            const bool want_prov = mot_impl::env().defer_blend != 0 && S.world == 1 && kp.use_lds && mot_impl::env().prov;
            HIPCHK(aux_acquire(c->cfg.device, reserve, want_prov, &d->side, &d->emu));
            d->aux_dev = c->cfg.device; d->aux_reserve = reserve;
            HIPCHK(hipEventCreateWithFlags(&d->ev_mid, MOT_EVENT_FLAGS));
            HIPCHK(hipEventCreateWithFlags(&d->ev_in, MOT_EVENT_FLAGS));
            // Deferred blend (default; MOT_DEFER_BLEND=0 restores the blend launch): the model update of frame f rides in frame f + 1's
            // predict kernel.  The spectra of frame f must then outlive the feature launch of frame f + 1: two buffers, by frame parity.
            d->defer = mot_impl::env().defer_blend != 0;
            d->spec_stride = (size_t)md * MOT_NCHAN * kp.nbins;
            HIPCHK(d->det_spec.alloc(d->spec_stride * 3));
            for (int b = 0; b < 3; b++) HIPCHK(hipEventCreateWithFlags(&d->ev_spec[b], MOT_EVENT_FLAGS));
            if (d->defer) {
                HIPCHK(d->pend.alloc((size_t)cap + MOT_SHADOW_SLOTS)); HIPCHK(hipMemsetAsync(d->pend.p, 0xFF, sizeof(int) * (cap + MOT_SHADOW_SLOTS), c->stream));   // on the context's stream: see mot_ctx_create
                S.defer = 1; S.pend_det = d->pend.p;
            }
            d->split = true;
            // provisional commits of two-row tie frames (see DevLoop::prov): one rank, LDS-resident template (HBM-slab launches index their slabs by item)
            if (d->defer && d->emu) {
                HIPCHK(hipEventCreateWithFlags(&d->ev_rs, MOT_EVENT_FLAGS));
                HIPCHK(d->seg2.alloc(2 * ((size_t)cap + MOT_SHADOW_SLOTS))); HIPCHK(hipMemsetAsync(d->seg2.p, 0, sizeof(bbox_t) * d->seg2.n, c->stream));
                HIPCHK(d->det_copy.alloc(2 * (size_t)md)); HIPCHK(hipMemsetAsync(d->det_copy.p, 0, sizeof(bbox_t) * d->det_copy.n, c->stream));
                HIPCHK(d->prov_rec.alloc(1)); HIPCHK(hipMemsetAsync(d->prov_rec.p, 0, sizeof(ProvRec), c->stream));
                d->prov = true;
            }
        }
        if (getenv("MOT_TRACE") && atoi(getenv("MOT_TRACE")) && !multi) { HIPCHK(d->trace.alloc((size_t)32 * cap * 8)); HIPCHK(hipMemsetAsync(d->trace.p, 0xFF, sizeof(int) * d->trace.n, c->stream)); }
    }
    else c->kal_free.clear();
    HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipDeviceSynchronize());   // nothing of the set-up (fills, table uploads) is still in flight when the first frame is enqueued
    c->devloop = d.release();
    *out = c->devloop;
    return MOT_OK;
}

#define MOT_PROV_MIN_DETS 600
int prov_min_dets()
{
    static const int v = [] { const char* e = getenv("MOT_PROV_MIN_DETS"); return e ? atoi(e) : MOT_PROV_MIN_DETS; }();   // (A/B: the detection count from which a frame takes the stream-emulation chain)
    return v;
}
int split_early_max()
{
    const int v = mot_impl::env().split_early_max;
    return v < 0 ? MOT_SPLIT_EARLY_MAX : v;
}

// The patch step of the last associated frame (launch_prov_patch: a no-op on the device unless that frame was committed provisionally).
// pred_cur: the boxes a predict launch has written since (its shadow items behind the live tracks), or null.
int dl_patch(mot_ctx* c, DevLoop* d, bbox_t* pred_cur)
{
    if (!d->prov || !d->patch_owed) return MOT_OK;
    const DLState& S = d->S;
    LifeArgs life{}; life.enabled = 1; life.S = S; life.kp = c->pools[d->pool]->dev; life.kal = c->kal;
    life.prov.enabled = 1; life.prov.sh_base = S.cap; life.prov.rec = d->prov_rec.p;
    const bbox_t* trk = d->seg2.p + (size_t)d->par_last * (S.cap + MOT_SHADOW_SLOTS);
    const bbox_t* det = d->det_copy.p + (size_t)d->par_last * S.max_dets;
    life.trk_pred = trk; life.dets = det; life.nD = d->nD_last;
    HIPCHK(launch_prov_patch(c->assoc, life, trk, det, d->nD_last, d->seq_last, pred_cur, c->stream));
    d->patch_owed = false;
    return MOT_OK;
}

int dl_begin(mot_ctx* c, DevLoop* d, const void* frame_dev, hipEvent_t* ev, const void* dets_dev = nullptr, int nD = 0)
{
    RoctxRange range_("mot.frame.predict");
    DLState& S = d->S;
    if (d->begun) return fail(MOT_ERR_STATE, "mot_step_begin_device called twice");
    if (S.kind == MOT_TRACKER_KCF && !frame_dev) return fail(MOT_ERR_ARG, "null frame");
    d->frame = frame_dev;
    d->frame_no++;
    d->begin_dets = dets_dev; d->begin_nD = nD;
    // this frame's spectra buffer: the one a look-ahead launch of the previous call filled for exactly this frame and detection list,
    // else a free one (then the features are computed within the frame, as before)
    d->have_cur = d->split && d->pf_valid && d->pf_frame == frame_dev && d->pf_dets == dets_dev && d->pf_nD == nD && dets_dev;
    d->buf_cur = d->have_cur ? d->pf_buf : (d->buf_prev + 1) % 3;
    if (d->split && d->pf_valid && !d->have_cur) {
        // the caller announced another frame / list than the one it passes now.  The look-ahead launch of the previous call may still be
        // WRITING its spectra buffer (and the shared HBM slabs) and READING the frame it was announced with: this frame takes the buffer
        // that is neither that one nor the previous frame's, and the context stream waits for the abandoned launch (rare; costs one wait)
        d->buf_cur = 3 - d->buf_prev - d->pf_buf;
        if (d->spec_side[d->pf_buf]) { HIPCHK(hipStreamWaitEvent(c->stream, d->ev_spec[d->pf_buf], 0)); d->spec_side[d->pf_buf] = false; }
    }
    d->pf_valid = false;
    if (!d->have_cur) d->spec_side[d->buf_cur] = false;
    float2* spec_cur = d->det_spec.p + (size_t)d->buf_cur * d->spec_stride;            // this frame's detection spectra
    const float2* spec_prev = d->det_spec.p + (size_t)d->buf_prev * d->spec_stride;     // ... the previous frame's
    const int par = (int)(d->frame_no & 1);
    if (d->prov) S.gather = d->seg2.p + (size_t)par * (S.cap + MOT_SHADOW_SLOTS);   // this frame's predicted boxes: the half the previous frame's emulation is NOT reading
    bbox_t* seg = S.gather + (size_t)S.rank * S.spr;
    const int n_pred = S.spr + (d->prov ? MOT_SHADOW_SLOTS : 0);       // the shadow items of a provisionally committed frame ride at the end of the predict list
    d->feat_early = false;
    const int early_max = split_early_max();
    d->feat_joined = false;
    // workgroups that will actually WORK in a joined / early launch: this rank's share of the tracks (every segment holds cap slots since the
    // ownership tid % world drifts, but a rank owns about cap / world of them; the surplus workgroups of the grid return at once) + detections
    const int own_est = (S.cap + S.world - 1) / S.world;
    const bool early = d->split && !d->have_cur && S.kind == MOT_TRACKER_KCF && dets_dev && nD > 0 && nD <= S.max_dets && own_est + nD <= early_max;
    // the blend prologue of this predict reads the previous frame's spectra: behind the side-stream launch that wrote them
    if (d->split && d->defer && d->spec_side[d->buf_prev]) HIPCHK(hipStreamWaitEvent(c->stream, d->ev_spec[d->buf_prev], 0));
    bool ext_timed = false;                                            // profiling: the predict launch records its own begin / end (below)
    const int joined_on = mot_impl::env().joined_launch;
    if (S.kind == MOT_TRACKER_KCF) {
        KcfLaunch l{}; l.slots = S.loc_slots; l.count = S.loc_count; l.frame = (const uint8_t*)frame_dev; l.boxes_out = seg; l.clamp = 1; l.dbg = c->dbg_on ? c->dbg.p : nullptr;
        if (S.ncls > 1) { l.pools = S.pools; l.cls = S.loc_cls; l.slab_stride = d->slab_stride; l.lds_bytes = d->lds_bytes; l.r1_any = d->r1_any; l.gen_any = d->gen_any; }
        if (d->defer) { l.pend_det = S.pend_det; l.pend_spec = spec_prev; }
        if (d->trace.p) { l.trace = d->trace.p; l.trace_frame = (int)d->frame_no; l.trace_cap = S.cap; }
        KcfLaunch lf{}; lf.frame = (const uint8_t*)frame_dev; lf.boxes_in = (const bbox_t*)dets_dev; lf.spec_out = spec_cur; lf.slab_base = S.cap;
        if (early && joined_on) {
            // small frames leave most CUs idle during the predict: the detection features (they only need the frame and the boxes) ride in
            // the SAME launch as extra workgroups -- no side stream, no events (MOT_JOINED_LAUNCH=0: side-stream launch as in round 2)
            if (ev) HIPCHK(hipEventRecord(ev[0], c->stream));
            HIPCHK(launch_kcf_predict_features(c->pools[d->pool]->dev, l, n_pred, lf, nD, c->stream));
            d->feat_early = true; d->feat_joined = true;
        } else {
            if (early) {
                // the side stream is ordered behind everything the caller has enqueued on the context stream so far (frame upload,
                // detector output): in-order execution makes this event cover the previous frame's update as well
                HIPCHK(hipEventRecord(d->ev_in, c->stream));
                HIPCHK(hipStreamWaitEvent(d->side, d->ev_in, 0));
                HIPCHK(launch_kcf_update(c->pools[d->pool]->dev, lf, nD, d->side, own_est + nD <= MOT_SPLIT_EXCL_MAX));   // own CUs beside the predict
                HIPCHK(hipEventRecord(d->ev_spec[d->buf_cur], d->side)); d->spec_side[d->buf_cur] = true;
                d->feat_early = true;
            }
            // profiling (single template size): the kernel's own begin / end stamps go to ev[0] / ev[1]
            ext_timed = ev && S.ncls <= 1;
            if (ev && !ext_timed) HIPCHK(hipEventRecord(ev[0], c->stream));
            hipEvent_t t0 = ext_timed ? ev[0] : nullptr, t1 = ext_timed ? ev[1] : nullptr;
            if (!ev && S.ncls <= 1 && (size_t)(2 * d->pt_used + 1) < d->pt.size()) { t0 = d->pt[2 * d->pt_used]; t1 = d->pt[2 * d->pt_used + 1]; d->pt_used++; }   // (debug) in-loop timing
            // The chain event (the side stream's feature launch and the host-fed loop's uploads wait for it) rides in the predict launch's own
            // packet as its completion event instead of a record packet of its own behind it: one dispatch gap (~5 us) less in front of the row scan.
            d->mid_by_predict = false;
            if (!t0 && !t1 && S.ncls <= 1 && d->split && d->ev_mid) { t1 = d->ev_mid; d->mid_by_predict = true; }
            HIPCHK(launch_kcf_predict(c->pools[d->pool]->dev, l, n_pred, c->stream, t0, t1));
        }
        // the previous frame's patch step: a no-op unless that frame was committed provisionally -- then it waits for the emulation that has been
        // running beside this predict, and copies the shadow items' results over the two tracks if the reference's optimum is the swapped one
        if (d->prov && d->patch_owed) { int rc = dl_patch(c, d, seg); if (rc) return rc; }
    } else { if (ev) HIPCHK(hipEventRecord(ev[0], c->stream)); HIPCHK(launch_kalman_predict(c->kal, S.loc_slots, S.loc_count, S.spr, seg, 1, c->stream)); }
    if (ev && !ext_timed) HIPCHK(hipEventRecord(ev[1], c->stream));
    d->begun = true;
    return MOT_OK;
}

// release_inputs: the context stream waits, at the end of the frame, for the side-stream feature launch that read THIS frame's image and
// detection list -- so a caller may overwrite both in stream order behind the call (round-3 advisor finding: with the deferred blend only
// the next frame's predict waited for that launch).  The next predict needed that wait anyway (its blend prologue reads the spectra), so
// the wait moves, it is not added.  mot_step_frame_host passes true as well: the reuse of its three buffers is ordered behind chain events
// (ev_ring), and the chain event of frame f - 2 lies behind this wait of frame f - 3 in stream order -- with release_inputs == false a frame whose
// features were computed inside the frame (side stream) could still be read when its buffer is uploaded again.
int dl_finish(mot_ctx* c, DevLoop* d, const void* gathered, const void* dets_dev, int nD, hipEvent_t* ev, bool release_inputs = true)
{
    RoctxRange range_("mot.frame.assoc_update");
    DLState& S = d->S;
    if (!d->begun) return fail(MOT_ERR_STATE, "mot_step_finish_device without mot_step_begin_device");
    // A refused call leaves the frame BEGUN (round-5 advisor finding: the checks used to run behind `begun = false`, so a finish with a wrong list
    // lost the frame -- its predict had run and consumed the pending updates -- and could not be repeated with the right one).
    if (nD < 0 || nD > S.max_dets || (nD && !dets_dev)) return fail(MOT_ERR_ARG, "bad detection list (%d, max %d)", nD, S.max_dets);
    // the two-call form: the spectra of this frame were computed (or adopted from a look-ahead launch) for the list given to the begin call -- the
    // list associated here must be that one (round-4 advisor finding: a different list silently got the other list's spectra)
    if ((d->feat_early || d->have_cur) && (dets_dev != d->begin_dets || nD != d->begin_nD))
        return fail(MOT_ERR_ARG, "mot_step_finish_device: detection list (%p, %d) differs from the one given to the begin call (%p, %d)", dets_dev, nD, d->begin_dets, d->begin_nD);
    d->begun = false;
    const bbox_t* g = gathered ? (const bbox_t*)gathered : S.gather;
    const bbox_t* trk = g;
    if (S.world > 1) { hipLaunchKernelGGL(dl_scatter_kernel, dim3(1), dim3(1024), 0, c->stream, S, g); HIPCHK(hipGetLastError()); trk = S.pred; }
    const bbox_t* dets = (const bbox_t*)dets_dev;
    KcfPool kp{}; if (S.kind == MOT_TRACKER_KCF) kp = c->pools[d->pool]->dev;
    const bool split = d->split && S.kind == MOT_TRACKER_KCF && nD > 0;
    float2* spec_cur = d->det_spec.p + (size_t)d->buf_cur * d->spec_stride;
    const bool feat_here = split && !d->feat_early && !d->have_cur;     // this frame's detection features still have to be computed (side stream, beside the chain)
    // (small frames are better off with the joined predict + feature launch of their own frame than with a side-stream launch ahead)
    const bool ahead = d->split && S.kind == MOT_TRACKER_KCF && d->next_frame && d->next_dets && d->next_nD > 0 && d->next_nD <= S.max_dets &&
                       (S.cap + S.world - 1) / S.world + d->next_nD > split_early_max();
    // the lifecycle step rides in the tail of the Munkres kernel (one launch and one dispatch gap fewer per frame)
    LifeArgs life{}; life.enabled = 1; life.S = S; life.kp = kp; life.kal = c->kal; life.trk_pred = trk; life.dets = dets; life.nD = nD;
    AssocEmu emu{}; const int par = (int)(d->frame_no & 1);
    // Frame by frame: the stream-emulation chain (and with it a provisional commit) only where it pays.  Its fixed costs -- the patch step's launch behind the
    // predict (7 us even as a no-op) and the completion event the row scan carries for the emulation stream (~5.6 us of idle stream behind it) -- are 12 us per
    // frame; what it saves is a predict launch per TIE frame, and ties grow with the square of the density: 1024 detections + 14 %, 768 + 10 %, 640 + 2 %, 512 - 5 %, 256 - 11 %
    // (profiles/r06_prov_threshold.log).  Below MOT_PROV_MIN_DETS detections the frame takes the two-workgroup launch.
    const bool prov_now = d->prov && nD >= prov_min_dets();
    if (prov_now) {
        if (d->patch_owed) { int rc = dl_patch(c, d, nullptr); if (rc) return rc; }   // (two-call form without a predict in between: cannot happen, but never two frames owed)
        life.prov.enabled = mot_impl::env().prov == 4 ? 0 : 1; life.prov.sh_base = S.cap; life.prov.rec = d->prov_rec.p;   // (MOT_PROV=4, bisecting aid: the stream-emulation chain without provisional commits)
        emu.stream = d->emu; emu.ev_rowscan = d->ev_rs; emu.det_copy = d->det_copy.p + (size_t)par * S.max_dets;
    }
    unsigned seq = 0;
    HIPCHK(launch_assoc(c->assoc, trk, S.nlive, S.cap, dets, nD, nullptr, 0, 0, 0, c->stream, ((feat_here || ahead || d->want_mid) && !d->mid_by_predict) ? d->ev_mid : nullptr, &life,
                        prov_now ? &emu : nullptr, &seq));
    if (prov_now) { d->patch_owed = true; d->seq_last = seq; d->nD_last = nD; d->par_last = par; }
    if (prov_now && ev) { int rc = dl_patch(c, d, nullptr); if (rc) return rc; }   // profiled frame: the chain's stage time includes the emulation, as without the overlap
    d->mid_valid = feat_here || ahead || d->want_mid; d->mid_by_predict = false;
    if (feat_here || ahead) HIPCHK(hipStreamWaitEvent(d->side, d->ev_mid, 0));
    if (feat_here) {
        // features of every detection box, on the side stream, from the moment the association chain starts (its one-workgroup kernels
        // leave the chip idle); the frame and the boxes are inputs of this call
        KcfLaunch lf{}; lf.frame = (const uint8_t*)d->frame; lf.boxes_in = dets; lf.spec_out = spec_cur; lf.slab_base = S.cap;
        HIPCHK(launch_kcf_update(kp, lf, nD, d->side));
        HIPCHK(hipEventRecord(d->ev_spec[d->buf_cur], d->side)); d->spec_side[d->buf_cur] = true;
    }
    if (ahead) {
        // look-ahead: the NEXT frame's detection features (they depend on that frame and its boxes only, not on any tracker state), into
        // the third buffer; the next call recognises the frame / list it was computed for
        const int nb = 3 - d->buf_cur - d->buf_prev;                    // the buffer that is neither this frame's nor the previous frame's (the two always differ)
        KcfLaunch lf{}; lf.frame = (const uint8_t*)d->next_frame; lf.boxes_in = (const bbox_t*)d->next_dets; lf.spec_out = d->det_spec.p + (size_t)nb * d->spec_stride; lf.slab_base = S.cap;
        HIPCHK(launch_kcf_update(kp, lf, d->next_nD, d->side));
        HIPCHK(hipEventRecord(d->ev_spec[nb], d->side)); d->spec_side[nb] = true;
        d->pf_frame = d->next_frame; d->pf_dets = d->next_dets; d->pf_nD = d->next_nD; d->pf_buf = nb; d->pf_valid = true;
    }
    d->next_frame = nullptr; d->next_dets = nullptr; d->next_nD = 0;
    if (ev) HIPCHK(hipEventRecord(ev[2], c->stream));
    if (ev) HIPCHK(hipEventRecord(ev[3], c->stream));
    const int upd_max = S.spr + nD;
    if (S.kind == MOT_TRACKER_KCF) {
        KcfLaunch l{}; l.slots = S.upd_slots; l.count = S.upd_count; l.frame = (const uint8_t*)d->frame; l.boxes_in = S.upd_boxes; l.dbg = c->dbg_on ? c->dbg.p + 16 : nullptr;
        if (S.ncls > 1) { l.pools = S.pools; l.cls = S.upd_cls; l.slab_stride = d->slab_stride; l.lds_bytes = d->lds_bytes; l.r1_any = d->r1_any; l.gen_any = d->gen_any; }
        if (split) { l.det_spec = spec_cur; l.det_index = S.upd_det; }
        if (d->trace.p) { l.trace = d->trace.p; l.trace_frame = (int)d->frame_no; l.trace_cap = S.cap; }
        if (d->defer) {
            // only tracks that keep their PREDICTED box (unmatched, not lost: td.cpp:550-581) are left in the update list -- few or none,
            // count known on the device only: a small grid loops over them.  They read no spectra, so this launch does not wait for the
            // feature launch; the NEXT predict does (its blend prologue reads this frame's spectra), see dl_begin.
            l.grid_stride = 1;
            HIPCHK(launch_kcf_update(kp, l, upd_max, c->stream));
        } else {
            if (split && d->spec_side[d->buf_cur]) HIPCHK(hipStreamWaitEvent(c->stream, d->ev_spec[d->buf_cur], 0));   // the blend launch reads this frame's spectra
            HIPCHK(launch_kcf_update(kp, l, upd_max, c->stream));
        }
        if (split) d->buf_prev = d->buf_cur;
    } else HIPCHK(launch_kalman_update(c->kal, S.upd_slots, S.upd_count, upd_max, S.upd_boxes, c->stream));
    if (ev) HIPCHK(hipEventRecord(ev[4], c->stream));
    if (release_inputs && d->split && d->spec_side[d->buf_prev]) {       // (rotated above: buf_prev is this frame's spectra buffer)
        HIPCHK(hipStreamWaitEvent(c->stream, d->ev_spec[d->buf_prev], 0));
        d->spec_side[d->buf_prev] = false;                             // the context stream is ordered behind that launch from here on
    }
    return MOT_OK;
}

} // namespace

namespace mot_impl {
// sticky device-side errors of the device-resident loop (the stream must be idle).  Also drains the side stream: a detection-feature
// launch may still be reading the frame the caller is about to release.
// every synchronisation point and read-back: a provisionally committed frame gets its patch step first (enqueued on the context's stream; it waits
// for the emulation itself), so what the caller reads is the reference's state, never the provisional one
int devloop_flush(mot_ctx* c)
{
    if (!c->devloop || !c->devloop->prov || !c->devloop->patch_owed) return MOT_OK;
    return dl_patch(c, c->devloop, nullptr);
}
int devloop_check(mot_ctx* c)
{
    if (!c->devloop) return MOT_OK;
    if (c->devloop->patch_owed) { int rc = devloop_flush(c); if (rc) return rc; HIPCHK(hipStreamSynchronize(c->stream)); }
    if (c->devloop->side) HIPCHK(hipStreamSynchronize(c->devloop->side));
    if (c->devloop->emu) HIPCHK(hipStreamSynchronize(c->devloop->emu));
    int err[8];
    HIPCHK(hipMemcpy(err, c->devloop->S.err, sizeof err, hipMemcpyDeviceToHost));
    if (err[5]) return fail(MOT_ERR_DEVICE, "provisional commit of chain %d: the order-exact emulation returned neither of the two optima (internal inconsistency)", err[5]);
    if (err[4]) return fail(MOT_ERR_DEVICE, "Munkres helper workgroups timed out (hand-off %d); the frame was dropped", err[4]);
    if (err[3]) return fail(MOT_ERR_DEVICE, "all-gather segment overflow (%d)", err[3]);
    return MOT_OK;
}
} // namespace mot_impl

extern "C" {

int mot_step_begin_device(mot_ctx* c, const void* frame_dev, void** local_boxes_dev, int* slots_per_rank)
{
    if (!c) return fail(MOT_ERR_ARG, "null ctx");
    int rc = ensure_device(c); if (rc) return rc;
    DevLoop* d; rc = devloop_get(c, &d); if (rc) return rc;
    rc = dl_begin(c, d, frame_dev, d->prof_two_call ? d->ev : nullptr); if (rc) return rc;
    if (local_boxes_dev) *local_boxes_dev = d->S.gather + (size_t)d->S.rank * d->S.spr;
    if (slots_per_rank) *slots_per_rank = d->S.spr;
    return MOT_OK;
}

// the same with the frame's detection list known at the start and one frame of look-ahead (see mot_step_frame_device_ahead): a sharded rank
// then computes the NEXT frame's detection features beside this frame's association chain as well, instead of inside the frame
int mot_step_begin_device_ahead(mot_ctx* c, const void* frame_dev, const void* dets_dev, int nD, const void* next_frame_dev, const void* next_dets_dev, int next_nD,
                                void** local_boxes_dev, int* slots_per_rank)
{
    if (!c) return fail(MOT_ERR_ARG, "null ctx");
    int rc = ensure_device(c); if (rc) return rc;
    DevLoop* d; rc = devloop_get(c, &d); if (rc) return rc;
    if (mot_impl::env().lookahead) { d->next_frame = next_frame_dev; d->next_dets = next_dets_dev; d->next_nD = next_nD; }
    rc = dl_begin(c, d, frame_dev, d->prof_two_call ? d->ev : nullptr, dets_dev, nD); if (rc) return rc;
    if (local_boxes_dev) *local_boxes_dev = d->S.gather + (size_t)d->S.rank * d->S.spr;
    if (slots_per_rank) *slots_per_rank = d->S.spr;
    return MOT_OK;
}

int mot_step_finish_device(mot_ctx* c, const void* gathered_boxes_dev, const void* dets_dev, int nD)
{
    if (!c || !c->devloop) return fail(MOT_ERR_STATE, "mot_step_finish_device without mot_step_begin_device");
    int rc = ensure_device(c); if (rc) return rc;
    DevLoop* d = c->devloop;
    if (d->prof_two_call) HIPCHK(hipEventRecord(d->ev[5], c->stream));   // the caller's all-gather lies between ev[1] and this
    return dl_finish(c, d, gathered_boxes_dev, dets_dev, nD, d->prof_two_call ? d->ev : nullptr);
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

// ---- native RCCL leg ----------------------------------------------------------------------------------------------
namespace {
typedef int (*nccl_all_gather_fn)(const void*, void*, size_t, int /* ncclDataType_t */, void* /* ncclComm_t */, hipStream_t);
struct RcclBinding { nccl_all_gather_fn fn = nullptr; std::string why; };
const RcclBinding* rccl_binding()
{
    static const RcclBinding b = [] {                                   // bound once, thread-safe (function-local static); the loader's message is kept with it
        RcclBinding r;
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { const char* m = dlerror(); r.why = m ? m : "dlopen failed"; return r; }
        r.fn = reinterpret_cast<nccl_all_gather_fn>(dlsym(h, "ncclAllGather"));
        if (!r.fn) { const char* m = dlerror(); r.why = m ? m : "symbol ncclAllGather missing"; }
        return r;
    }();
    return &b;
}
} // namespace

int mot_step_frame_sharded_ahead(mot_ctx* c, const void* frame_dev, const void* dets_dev, int nD, const void* next_frame_dev, const void* next_dets_dev, int next_nD, void* nccl_comm);
int mot_step_frame_sharded(mot_ctx* c, const void* frame_dev, const void* dets_dev, int nD, void* nccl_comm)
{
    return mot_step_frame_sharded_ahead(c, frame_dev, dets_dev, nD, nullptr, nullptr, 0, nccl_comm);
}

int mot_step_frame_sharded_ahead(mot_ctx* c, const void* frame_dev, const void* dets_dev, int nD, const void* next_frame_dev, const void* next_dets_dev, int next_nD, void* nccl_comm)
{
    if (!c || !nccl_comm) return fail(MOT_ERR_ARG, "null argument");
    int rc = ensure_device(c); if (rc) return rc;
    const RcclBinding& rb = *rccl_binding();
    nccl_all_gather_fn all_gather = rb.fn;
    if (!all_gather) return fail(MOT_ERR_DEVICE, "librccl.so.1 / ncclAllGather not available: %s", rb.why.c_str());
    DevLoop* d; rc = devloop_get(c, &d); if (rc) return rc;
    if (mot_impl::env().lookahead && next_frame_dev) { d->next_frame = next_frame_dev; d->next_dets = next_dets_dev; d->next_nD = next_nD; }
    rc = dl_begin(c, d, frame_dev, d->prof_two_call ? d->ev : nullptr, dets_dev, nD); if (rc) return rc;
    // the frame's single collective: every rank's segment of predicted boxes, in place (send = recv + rank * count), on the SAME stream
    // as the kernels on both sides of it -- stream order is the only synchronisation
    const DLState& S = d->S;
    const size_t count = (size_t)S.spr * sizeof(bbox_t);
    const int ncclChar = 0;
    const int nr = all_gather(reinterpret_cast<const char*>(S.gather) + (size_t)S.rank * count, S.gather, count, ncclChar, nccl_comm, c->stream);
    if (nr != 0) { d->begun = false; return fail(MOT_ERR_DEVICE, "ncclAllGather failed (ncclResult_t %d)", nr); }
    if (d->prof_two_call) HIPCHK(hipEventRecord(d->ev[5], c->stream));
    return dl_finish(c, d, nullptr, dets_dev, nD, d->prof_two_call ? d->ev : nullptr);
}

int mot_step_frame_host(mot_ctx* c, const uint8_t* host_bgr, const bbox_t* host_dets, int nD)
{
    if (!c || !host_bgr || nD < 0 || (nD && !host_dets)) return fail(MOT_ERR_ARG, "bad argument");
    if (c->cfg.world != 1) return fail(MOT_ERR_STATE, "sharded context: upload the frame yourself and use mot_step_frame_sharded");
    if (nD > c->cfg.max_dets) return fail(MOT_ERR_CAPACITY, "%d detections exceed max_dets %d", nD, c->cfg.max_dets);
    int rc = ensure_device(c); if (rc) return rc;
    DevLoop* d; rc = devloop_get(c, &d); if (rc) return rc;
    const size_t fbytes = (size_t)MOT_FRAME_W * MOT_FRAME_H * 3;
    if (!d->host_ok) {
        HIPCHK(hipStreamCreateWithFlags(&d->copy, hipStreamNonBlocking));
        for (int b = 0; b < 3; b++) {
            HIPCHK(d->hbuf[b].alloc(fbytes)); HIPCHK(d->dbuf[b].alloc((size_t)c->cfg.max_dets));
            HIPCHK(hipEventCreateWithFlags(&d->ev_up[b], MOT_EVENT_FLAGS)); HIPCHK(hipEventCreateWithFlags(&d->ev_ring[b], MOT_EVENT_FLAGS));
        }
        d->ev_mid0 = d->ev_mid;
        d->host_ok = true;
    }
    // Frame f lives in buffer f % 3.  Schedule (the host runs ahead of the GPU, so these orderings decide where things execute):
    //   copy stream:  upload(f)   behind the chain event of frame f - 2  -> beside the association chain of frame f - 2
    //   side stream:  features(f) behind the chain event of frame f - 1 and the upload -> beside the chain of frame f - 1 (one frame of look-ahead
    //                 without the caller's help: the features need only the uploaded frame and boxes; dl_begin recognises the buffer, pf_*)
    //   main stream:  predict(f) behind the upload -- nothing but the predict runs in a predict's window, as in the device-resident loop.
    // The chain event of frame g is recorded behind predict(g), and predict(g) starts behind everything of frame g - 1 (its blend prologue waited for
    // features(g - 1), the residual update of g - 1 precedes it in stream order): so event(f - 2) says buffer (f - 3) % 3 = f % 3 is free, and event(f - 1)
    // says the spectra buffer features(f) overwrites (last read by predict(f - 1)'s blend) is free.
    const unsigned f = d->host_no;
    const int b = (int)(f % 3);
    // upload = copy KERNEL on the copy stream (pinned, device-mapped host memory); pageable / unregistered memory takes the runtime's staged copy
    const void* src_f = host_bgr; const void* src_d = host_dets;
    bool by_kernel = false;
    if (((uintptr_t)host_bgr % 16 == 0) && (!nD || (uintptr_t)host_dets % 8 == 0)) {
        hipPointerAttribute_t af{}, ad{};
        const bool okf = hipPointerGetAttributes(&af, host_bgr) == hipSuccess && af.devicePointer != nullptr;
        const bool okd = !nD || (hipPointerGetAttributes(&ad, host_dets) == hipSuccess && ad.devicePointer != nullptr);
        if (okf && okd) { by_kernel = true; src_f = af.devicePointer; if (nD) src_d = ad.devicePointer; }
        else (void)hipGetLastError();                                  // pageable / unregistered memory: the runtime's staged copy below
    }
    const bool lookahead = d->split && d->S.kind == MOT_TRACKER_KCF && d->S.ncls <= 1 && mot_impl::env().lookahead;
    // Every configuration orders the upload of frame f behind ev_ring[(f - 2) % 3], the chain event of frame f - 2.  That event is recorded behind
    // predict(f - 2), which in stream order lies behind dl_finish(f - 3)'s release_inputs wait for the side-stream launch that read buffer (f - 3) % 3 --
    // so the buffer is free also when a frame's features were computed inside the frame (Kalman, size classes, look-ahead switched off).
    {
        if (f >= 3) HIPCHK(hipStreamWaitEvent(d->copy, d->ev_ring[(f - 2) % 3], 0));
        if (by_kernel) {
            const size_t n16 = fbytes / 16, n8 = (size_t)nD * sizeof(bbox_t) / 8;
            hipLaunchKernelGGL(h2d_copy_kernel, dim3(32), dim3(256), 0, d->copy, reinterpret_cast<uint4*>(d->hbuf[b].p), reinterpret_cast<const uint4*>(src_f), n16,
                               reinterpret_cast<uint2*>(d->dbuf[b].p), reinterpret_cast<const uint2*>(src_d), n8);
            HIPCHK(hipGetLastError());
        } else {
            HIPCHK(hipMemcpyAsync(d->hbuf[b].p, host_bgr, fbytes, hipMemcpyHostToDevice, d->copy));
            if (nD) HIPCHK(hipMemcpyAsync(d->dbuf[b].p, host_dets, sizeof(bbox_t) * nD, hipMemcpyHostToDevice, d->copy));
        }
        HIPCHK(hipEventRecord(d->ev_up[b], d->copy));
        HIPCHK(hipStreamWaitEvent(c->stream, d->ev_up[b], 0));
    }
    d->want_mid = true;                                       // every frame's chain records its event (the uploads two frames on wait for it)
    if (lookahead && nD > 0 && f >= 1 && d->mid_valid && !d->pf_valid && (d->S.cap + d->S.world - 1) / d->S.world + nD > split_early_max()) {
        const int nb = (d->buf_prev + 1) % 3;
        // (order matters: a wait for an already-complete event queued BEHIND the pending one started the launch 45 us after the event fired)
        HIPCHK(hipStreamWaitEvent(d->side, d->ev_up[b], 0));
        HIPCHK(hipStreamWaitEvent(d->side, d->ev_ring[(f - 1) % 3], 0));
        KcfLaunch lf{}; lf.frame = (const uint8_t*)d->hbuf[b].p; lf.boxes_in = d->dbuf[b].p; lf.spec_out = d->det_spec.p + (size_t)nb * d->spec_stride; lf.slab_base = d->S.cap;
        HIPCHK(launch_kcf_update(c->pools[d->pool]->dev, lf, nD, d->side));
        HIPCHK(hipEventRecord(d->ev_spec[nb], d->side)); d->spec_side[nb] = true;
        d->pf_frame = d->hbuf[b].p; d->pf_dets = d->dbuf[b].p; d->pf_nD = nD; d->pf_buf = nb; d->pf_valid = true;
    }
    if (d->want_mid) d->ev_mid = d->ev_ring[b];                        // this frame's chain event
    rc = dl_begin(c, d, d->hbuf[b].p, nullptr, d->dbuf[b].p, nD); if (rc) return rc;
    // a frame whose features were NOT computed ahead launches them inside dl_finish, on the side stream, reading this frame's buffer: the upload that
    // reuses the buffer (three calls on) is behind the chain event of frame f + 1, and predict(f + 1) waits for that launch (release_inputs below)
    rc = dl_finish(c, d, nullptr, d->dbuf[b].p, nD, nullptr, true); if (rc) return rc;
    d->host_no++;
    return MOT_OK;
}

int mot_step_frame_device_ahead(mot_ctx* c, const void* frame_dev, const void* dets_dev, int nD, const void* next_frame_dev, const void* next_dets_dev, int next_nD)
{
    if (!c) return fail(MOT_ERR_ARG, "null ctx");
    if (c->cfg.world != 1) return fail(MOT_ERR_STATE, "sharded context: use mot_step_begin_device / all-gather / mot_step_finish_device");
    int rc = ensure_device(c); if (rc) return rc;
    DevLoop* d; rc = devloop_get(c, &d); if (rc) return rc;
    const int ahead_on = mot_impl::env().lookahead;
    if (ahead_on) { d->next_frame = next_frame_dev; d->next_dets = next_dets_dev; d->next_nD = next_nD; }
    rc = dl_begin(c, d, frame_dev, nullptr, dets_dev, nD); if (rc) return rc;
    return dl_finish(c, d, nullptr, dets_dev, nD, nullptr);
}

// (debug) the predict launch's own duration while the ordinary step calls run: arm n pairs of events, step n frames, read the times
int mot_debug_predict_timing(mot_ctx* c, int n_pairs)
{
    if (!c || n_pairs < 0) return fail(MOT_ERR_ARG, "bad argument");
    int rc = ensure_device(c); if (rc) return rc;
    DevLoop* d; rc = devloop_get(c, &d); if (rc) return rc;
    MOT_SYNC_CTX(c);
    for (hipEvent_t e : d->pt) (void)hipEventDestroy(e);
    d->pt.assign((size_t)2 * n_pairs, nullptr); d->pt_used = 0;
    for (hipEvent_t& e : d->pt) HIPCHK(hipEventCreate(&e));
    return MOT_OK;
}
int mot_debug_predict_times(mot_ctx* c, float* out_ms, int cap, int* n)
{
    if (!c || !c->devloop || !out_ms || !n) return fail(MOT_ERR_ARG, "bad argument");
    DevLoop* d = c->devloop;
    MOT_SYNC_CTX(c);
    int k = 0;
    for (; k < d->pt_used && k < cap; k++) HIPCHK(hipEventElapsedTime(&out_ms[k], d->pt[2 * k], d->pt[2 * k + 1]));
    *n = k; d->pt_used = 0;
    return MOT_OK;
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

// (debug / bench) stage times of the two-call and the sharded step on THIS rank: enable, step a frame, read.  Events on the context's stream:
// [0] predict launch (the kernel's own begin / end stamps)  [1] end of the predict -> start of the finish call: the all-gather (caller's or RCCL)
// [2] association chain incl. scatter and lifecycle  [3] residual update launch.  Reading synchronises the context's stream.
int mot_debug_profile_stages(mot_ctx* c, int enable, float* stage_ms4)
{
    if (!c) return fail(MOT_ERR_ARG, "null ctx");
    int rc = ensure_device(c); if (rc) return rc;
    DevLoop* d; rc = devloop_get(c, &d); if (rc) return rc;
    if (!d->ev_ok) { for (int i = 0; i < 8; i++) HIPCHK(hipEventCreate(&d->ev[i])); d->ev_ok = true; }
    if (stage_ms4 && d->prof_two_call) {
        MOT_SYNC_CTX(c);
        float ms;
        if (hipEventElapsedTime(&ms, d->ev[0], d->ev[1]) != hipSuccess) { (void)hipGetLastError(); return fail(MOT_ERR_STATE, "no profiled frame yet"); }
        stage_ms4[0] = ms;
        HIPCHK(hipEventElapsedTime(&ms, d->ev[1], d->ev[5])); stage_ms4[1] = ms;
        HIPCHK(hipEventElapsedTime(&ms, d->ev[5], d->ev[2])); stage_ms4[2] = ms;
        HIPCHK(hipEventElapsedTime(&ms, d->ev[3], d->ev[4])); stage_ms4[3] = ms;
    }
    d->prof_two_call = enable != 0;
    return MOT_OK;
}

int mot_live_count(mot_ctx* c, int* n_live)
{
    if (!c || !n_live) return fail(MOT_ERR_ARG, "null argument");
    if (!c->devloop) { *n_live = (int)c->live.size(); return MOT_OK; }
    MOT_SYNC_CTX(c);
    int rc = devloop_check(c); if (rc) return rc;
    HIPCHK(hipMemcpy(n_live, c->devloop->S.nlive, sizeof(int), hipMemcpyDeviceToHost));
    return MOT_OK;
}

int mot_overlay_live(mot_ctx* c, void* frame_dev)
{
    if (!c || !frame_dev) return fail(MOT_ERR_ARG, "null argument");
    if (!c->devloop) return fail(MOT_ERR_STATE, "mot_overlay_live needs the device-resident loop (host-orchestrated contexts: mot_overlay_draw)");
    int rc = ensure_device(c); if (rc) return rc;
    rc = devloop_flush(c); if (rc) return rc;                           // the live boxes of a provisionally committed frame are final only behind its patch step
    const DLState& S = c->devloop->S;
    return overlay_run(c, frame_dev, S.bbox, S.tid, S.nlive, S.cap);
}

int mot_live_response(mot_ctx* c, int live_index, float* out, int* f_rows, int* f_cols)
{
    if (!c || !c->devloop) return fail(MOT_ERR_STATE, "mot_live_response needs the device-resident loop");
    const DLState& S = c->devloop->S;
    if (S.kind != MOT_TRACKER_KCF || S.ncls > 1) return fail(MOT_ERR_STATE, "mot_live_response: single-template KCF loop only");
    const KcfPool& p = c->pools[c->devloop->pool]->dev;
    if (f_rows) *f_rows = p.hb; if (f_cols) *f_cols = p.wb;
    if (!out) return MOT_OK;
    MOT_SYNC_CTX(c);
    int n = 0, slot = -1;
    HIPCHK(hipMemcpy(&n, S.nlive, sizeof(int), hipMemcpyDeviceToHost));
    if (live_index < 0 || live_index >= n) return fail(MOT_ERR_ARG, "live index %d out of range (%d live tracks)", live_index, n);
    HIPCHK(hipMemcpy(&slot, S.slot + live_index, sizeof(int), hipMemcpyDeviceToHost));
    if (slot < 0) return fail(MOT_ERR_ARG, "live track %d is owned by another rank", live_index);
    HIPCHK(hipMemcpy(out, p.response + (size_t)slot * p.nb, sizeof(float) * p.nb, hipMemcpyDeviceToHost));
    return MOT_OK;
}

int mot_live_model(mot_ctx* c, int live_index, float* xm_out, float* alpha_out, bbox_t* pos, float* scale2, int* first_update, int* pending_det)
{
    if (!c || !c->devloop) return fail(MOT_ERR_STATE, "mot_live_model needs the device-resident loop");
    const DLState& S = c->devloop->S;
    if (S.kind != MOT_TRACKER_KCF || S.ncls > 1) return fail(MOT_ERR_STATE, "mot_live_model: single-template KCF loop only");
    const KcfPool& p = c->pools[c->devloop->pool]->dev;
    MOT_SYNC_CTX(c);
    int rc = devloop_check(c); if (rc) return rc;                      // drains the side stream too
    int n = 0, slot = -1;
    HIPCHK(hipMemcpy(&n, S.nlive, sizeof(int), hipMemcpyDeviceToHost));
    if (live_index < 0 || live_index >= n) return fail(MOT_ERR_ARG, "live index %d out of range (%d live tracks)", live_index, n);
    HIPCHK(hipMemcpy(&slot, S.slot + live_index, sizeof(int), hipMemcpyDeviceToHost));
    if (slot < 0) return fail(MOT_ERR_ARG, "live track %d is owned by another rank", live_index);
    if (xm_out) HIPCHK(hipMemcpy(xm_out, p.xm + (size_t)slot * MOT_NCHAN * p.nbins, sizeof(float2) * MOT_NCHAN * p.nbins, hipMemcpyDeviceToHost));
    if (alpha_out) HIPCHK(hipMemcpy(alpha_out, p.alpha + (size_t)slot * p.nbins, sizeof(float) * p.nbins, hipMemcpyDeviceToHost));
    if (pos) HIPCHK(hipMemcpy(pos, p.pos + slot, sizeof(bbox_t), hipMemcpyDeviceToHost));
    if (scale2) HIPCHK(hipMemcpy(scale2, p.scale + slot, sizeof(float2), hipMemcpyDeviceToHost));
    if (first_update) HIPCHK(hipMemcpy(first_update, p.first_update + slot, sizeof(int), hipMemcpyDeviceToHost));
    if (pending_det) { *pending_det = -1; if (S.defer && S.pend_det) HIPCHK(hipMemcpy(pending_det, S.pend_det + slot, sizeof(int), hipMemcpyDeviceToHost)); }
    return MOT_OK;
}

// (debug) stream-ordered snapshot of the device loop's small state into caller-owned DEVICE memory -- no host synchronisation, so a soak can
// record every frame of a run that must not be synchronised and read the records back only when the run went wrong (tools/lookahead_soak.py).
// Layout (4-byte words, cap = max_tracks): [0] nlive [1] upd_count [2] loc_count [3] frame_no | tid[cap] | live bbox[cap] (6 words each) |
// predicted boxes of this frame in item order [cap] (6 words each) | slot[cap] | KCF pos by SLOT [cap] (6 words each) | pend_det by slot [cap] |
// first_update by slot [cap] | upd_slots[64] | upd_det[64] | lap header [64].  Returns the number of bytes through *bytes (dst may be null).
int mot_debug_snapshot(mot_ctx* c, void* dst_dev, size_t* bytes)
{
    if (!c) return fail(MOT_ERR_ARG, "null ctx");
    int rc0 = ensure_device(c); if (rc0) return rc0;
    DevLoop* d; rc0 = devloop_get(c, &d); if (rc0) return rc0;
    if (dst_dev) { rc0 = devloop_flush(c); if (rc0) return rc0; }       // (stream-ordered, like the copies below)
    const DLState& S = d->S;
    const size_t cap = (size_t)S.cap;
    const size_t total = 4 * (4 + cap + 6 * cap + 6 * cap + cap + 6 * cap + cap + cap + 64 + 64 + 64);
    if (bytes) *bytes = total;
    if (!dst_dev) return MOT_OK;
    int rc = ensure_device(c); if (rc) return rc;
    char* q = (char*)dst_dev;
    auto put = [&](const void* src, size_t n) -> hipError_t { hipError_t e = src ? hipMemcpyAsync(q, src, n, hipMemcpyDeviceToDevice, c->stream) : hipMemsetAsync(q, 0xFF, n, c->stream); q += n; return e; };
    HIPCHK(put(S.nlive, 4)); HIPCHK(put(S.upd_count, 4)); HIPCHK(put(S.loc_count, 4));
    { HIPCHK(hipMemsetD32Async((hipDeviceptr_t)q, (int)d->frame_no, 1, c->stream)); q += 4; }   // word [3]: frames stepped so far (the value travels in the fill packet: no host source to keep alive)
    HIPCHK(put(S.tid, 4 * cap)); HIPCHK(put(S.bbox, 24 * cap));
    HIPCHK(put(S.gather + (size_t)S.rank * S.spr, 24 * cap));
    HIPCHK(put(S.slot, 4 * cap));
    const bool kcf1 = S.kind == MOT_TRACKER_KCF && S.ncls <= 1;
    const KcfPool* kp = kcf1 ? &c->pools[d->pool]->dev : nullptr;
    HIPCHK(put(kp ? (const void*)kp->pos : nullptr, 24 * cap));
    HIPCHK(put((kp && S.defer) ? (const void*)S.pend_det : nullptr, 4 * cap));
    HIPCHK(put(kp ? (const void*)kp->first_update : nullptr, 4 * cap));
    HIPCHK(put(S.upd_slots, 4 * 64)); HIPCHK(put(S.upd_det, 4 * 64));
    HIPCHK(put(c->assoc.lap.hdr, 4 * 64));
    return MOT_OK;
}

// (debug, MOT_TRACE=1) the per-workgroup records of the last 16 frames' predict and update launches (KcfLaunch::trace): 32 * max_tracks * 8 ints
int mot_debug_trace_read(mot_ctx* c, int* out, size_t cap_ints, size_t* n_ints)
{
    if (!c || !c->devloop) return fail(MOT_ERR_STATE, "no device loop");
    DevLoop* d = c->devloop;
    if (n_ints) *n_ints = d->trace.n;
    if (!out || !d->trace.p) return MOT_OK;
    MOT_SYNC_CTX(c);
    HIPCHK(hipMemcpy(out, d->trace.p, sizeof(int) * std::min(cap_ints, d->trace.n), hipMemcpyDeviceToHost));
    return MOT_OK;
}

// ---- checkpoint / resume of the device-resident loop (state dump / load) -------------------------------------------------
// One flat host record: header, the live list and its counters (the int arena, ids, boxes), the per-slot tracker state (KCF: model, alpha, pos,
// scale, first-update flag, response map, pending detection + the spectra of the last frame's detections, which the next predict blends in first;
// Kalman: x, P).  Loading needs a FRESH context of the same configuration; the loop then continues bit for bit as the saved one would have
// (tests/test_gpu_devloop.py::test_state_save_load_resumes_bit_for_bit).  Single template size only (no size classes).
namespace {
struct StateHeader { unsigned magic, version; int kind, cap, max_dets, rows, cols, world, rank, defer, split, nbins, nb; unsigned frame_no; unsigned long long ints_n, tids_n, boxes_n, spec_n, total; };
const unsigned kStateMagic = 0x4D4F5453u;   // "MOTS"
struct StatePart { void* dev; size_t bytes; };
int state_parts(mot_ctx* c, DevLoop* d, std::vector<StatePart>& parts, StateHeader& h)
{
    const DLState& S = d->S;
    if (S.ncls > 1) return fail(MOT_ERR_STATE, "state save / load: per-track template sizes are not supported");
    memset(&h, 0, sizeof h);
    h.magic = kStateMagic; h.version = 1; h.kind = S.kind; h.cap = S.cap; h.max_dets = S.max_dets; h.rows = S.rows; h.cols = S.cols; h.world = S.world; h.rank = S.rank;
    h.defer = d->defer ? 1 : 0; h.split = d->split ? 1 : 0; h.frame_no = d->frame_no;
    h.ints_n = d->ints.n; h.tids_n = d->tids.n; h.boxes_n = d->boxes.n;
    parts.push_back({d->ints.p, d->ints.n * sizeof(int)}); parts.push_back({d->tids.p, d->tids.n * sizeof(unsigned)}); parts.push_back({d->boxes.p, d->boxes.n * sizeof(bbox_t)});
    if (S.kind == MOT_TRACKER_KCF) {
        PoolHost& ph = *c->pools[d->pool];
        h.nbins = ph.dev.nbins; h.nb = ph.dev.nb;
        parts.push_back({ph.xm.p, ph.xm.n * sizeof(float2)}); parts.push_back({ph.alpha.p, ph.alpha.n * sizeof(float)}); parts.push_back({ph.pos.p, ph.pos.n * sizeof(bbox_t)});
        parts.push_back({ph.scale.p, ph.scale.n * sizeof(float2)}); parts.push_back({ph.first.p, ph.first.n * sizeof(int)}); parts.push_back({ph.response.p, ph.response.n * sizeof(float)});
        if (d->defer) {
            h.spec_n = d->spec_stride;
            parts.push_back({d->pend.p, d->pend.n * sizeof(int)});
            parts.push_back({d->det_spec.p + (size_t)d->buf_prev * d->spec_stride, d->spec_stride * sizeof(float2)});   // the last frame's detection spectra (pending blends read them)
        }
    } else {
        parts.push_back({c->kal_x.p, c->kal_x.n * sizeof(double)}); parts.push_back({c->kal_P.p, c->kal_P.n * sizeof(double)});
        parts.push_back({c->d_gather.p, c->d_gather.n * sizeof(bbox_t)});     // the Kalman predict is in / out on the boxes the lifecycle step left in the segment (kalman.cpp:112-115)
    }
    size_t total = sizeof(StateHeader);
    for (const StatePart& p : parts) total += p.bytes;
    h.total = total;
    return MOT_OK;
}
int quiesce(mot_ctx* c, DevLoop* d)
{
    if (d->begun) return fail(MOT_ERR_STATE, "state save / load between mot_step_begin_device and mot_step_finish_device");
    MOT_SYNC_CTX(c);
    if (d->side) HIPCHK(hipStreamSynchronize(d->side));
    if (d->copy) HIPCHK(hipStreamSynchronize(d->copy));
    return devloop_check(c);
}
} // namespace

int mot_state_save(mot_ctx* c, void* host_buf, size_t cap_bytes, size_t* bytes)
{
    if (!c) return fail(MOT_ERR_ARG, "null ctx");
    int rc = ensure_device(c); if (rc) return rc;
    DevLoop* d; rc = devloop_get(c, &d); if (rc) return rc;
    std::vector<StatePart> parts; StateHeader h;
    rc = state_parts(c, d, parts, h); if (rc) return rc;
    if (bytes) *bytes = (size_t)h.total;
    if (!host_buf) return MOT_OK;
    if (cap_bytes < h.total) return fail(MOT_ERR_CAPACITY, "state record needs %llu bytes, buffer holds %zu", h.total, cap_bytes);
    rc = quiesce(c, d); if (rc) return rc;
    char* q = (char*)host_buf;
    memcpy(q, &h, sizeof h); q += sizeof h;
    for (const StatePart& p : parts) { HIPCHK(hipMemcpy(q, p.dev, p.bytes, hipMemcpyDeviceToHost)); q += p.bytes; }
    return MOT_OK;
}

int mot_state_load(mot_ctx* c, const void* host_buf, size_t bytes)
{
    if (!c || !host_buf || bytes < sizeof(StateHeader)) return fail(MOT_ERR_ARG, "bad argument");
    int rc = ensure_device(c); if (rc) return rc;
    if (c->devloop && c->devloop->frame_no != 0) return fail(MOT_ERR_STATE, "mot_state_load needs a fresh context (this one has already stepped %u frames)", c->devloop->frame_no);
    DevLoop* d; rc = devloop_get(c, &d); if (rc) return rc;
    std::vector<StatePart> parts; StateHeader h, in;
    rc = state_parts(c, d, parts, h); if (rc) return rc;
    memcpy(&in, host_buf, sizeof in);
    if (in.magic != kStateMagic || in.version != 1) return fail(MOT_ERR_ARG, "not a state record of this library version");
    if (in.kind != h.kind || in.cap != h.cap || in.max_dets != h.max_dets || in.rows != h.rows || in.cols != h.cols || in.world != h.world || in.rank != h.rank ||
        in.defer != h.defer || in.split != h.split || in.ints_n != h.ints_n || in.tids_n != h.tids_n || in.boxes_n != h.boxes_n || in.spec_n != h.spec_n || in.nbins != h.nbins || in.total != h.total || bytes < in.total)
        return fail(MOT_ERR_ARG, "state record does not fit this context (tracker kind, capacities, template size, rank / world or frame structure differ)");
    rc = quiesce(c, d); if (rc) return rc;
    // the spectra of the saved frame go into the buffer this context regards as "previous" (a fresh context: buffer 0)
    const char* q = (const char*)host_buf + sizeof in;
    for (const StatePart& p : parts) { HIPCHK(hipMemcpy(p.dev, q, p.bytes, hipMemcpyHostToDevice)); q += p.bytes; }
    HIPCHK(hipDeviceSynchronize());
    d->frame_no = in.frame_no;
    d->pf_valid = false; d->have_cur = false; d->feat_early = false; d->feat_joined = false; d->mid_valid = false; d->mid_by_predict = false;
    for (bool& b : d->spec_side) b = false;
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
    MOT_SYNC_CTX(c);
    int rc = devloop_check(c); if (rc) return rc;
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
