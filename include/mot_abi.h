/* mot_abi.h -- C ABI of the MI355X-native per-frame tracker update.
 *
 * Drop-in boundary for the hot path of huangfcn/multiple-object-tracking
 * (FHOG -> KCF detect/train, Kalman predict/correct, association cost +
 * Munkres).  Plain C: pointers and sizes only, no C++/torch types.
 *
 * Two layers are exported by libmot_amd.so:
 *
 *  (1) the BATCH ABI (this file, prefix mot_): all live tracks of a frame are
 *      processed by one kernel launch per stage.  This is what a maintainer
 *      binds from top/td.cpp's tracker thread (td.cpp:306-748) and what the
 *      synthetic driver / bench / multi-GPU path call.
 *
 *  (2) the reference's own per-object interface with identical C++ linkage
 *      (td.cpp:229-234; implemented in trackers/kcf.cpp:455-491,
 *      trackers/kalman.cpp:131-163, trackers/hungarian/hungarian.cpp:29):
 *          void* tracker_new(bbox_t*);
 *          void  tracker_predict(void*, float* rgb, bbox_t*);
 *          void  tracker_update (void*, float* rgb, bbox_t*);
 *          void  tracker_delete (void*);
 *          void  assignmentoptimal(int*, double*, double*, int, int);
 *      exported (Itanium-mangled, struct tag _bbox_pos_s) by
 *      libmot_dropin_kcf.so / libmot_dropin_kalman.so -- link-time back-end
 *      selection exactly like the reference (yolo3tracker.vcxproj:134-138).
 *      They are batch-of-one wrappers over (1); see include/mot_dropin.hpp.
 *
 * Error convention: the reference's functions are void and never report
 * errors.  The batch ABI returns int: 0 = MOT_OK, negative
 * = error; mot_last_error() gives a message.  There is NO CPU fallback: if no
 * HIP device / kernel image is available every call fails with MOT_ERR_DEVICE.
 */
#ifndef MOT_ABI_H
#define MOT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bit-identical to top/cnntype.h:36-41 (note the field order l,t,b,r; inclusive coords). */
#ifndef MOT_BBOX_T_DEFINED
#define MOT_BBOX_T_DEFINED
typedef struct _bbox_pos_s {
    int l, t, b, r;
    int type;
    float score;
} bbox_t;
#endif

/* Bit-identical to top/cnntype.h:43-47: what the detector thread hands the tracker thread per frame
 * (td.cpp:330-333: `bbox_chain_t* pdetected = bbox_proc[prc_rptr]; ndetected = pdetected->nbox`).
 * 3076 bytes: nbox + 128 boxes. */
#ifndef MOT_BBOX_CHAIN_T_DEFINED
#define MOT_BBOX_CHAIN_T_DEFINED
#define MOT_CHAIN_MAX_BOXES 128
typedef struct _bbox_chain_s {
    int nbox;
    bbox_t bbox[MOT_CHAIN_MAX_BOXES];
} bbox_chain_t;
#endif
#if defined(__cplusplus)
static_assert(sizeof(bbox_t) == 24, "bbox_t must match top/cnntype.h:36-41");
static_assert(sizeof(bbox_chain_t) == 3076, "bbox_chain_t must match top/cnntype.h:43-47");
#elif defined(__STDC_VERSION__) && __STDC_VERSION__ >= 201112L
_Static_assert(sizeof(bbox_t) == 24, "bbox_t must match top/cnntype.h:36-41");
_Static_assert(sizeof(bbox_chain_t) == 3076, "bbox_chain_t must match top/cnntype.h:43-47");
#endif

/* top/cnntype.h:5-6 */
#define MOT_FRAME_W 1280
#define MOT_FRAME_H 720

enum {
    MOT_OK = 0,
    MOT_ERR_ARG = -1,      /* bad argument (null pointer, size out of range, unknown id) */
    MOT_ERR_DEVICE = -2,   /* HIP runtime / no device / kernel launch failure */
    MOT_ERR_CAPACITY = -3, /* more tracks / detections than the context was created for */
    MOT_ERR_STATE = -4     /* call not valid in the current state (e.g. no frame bound) */
};

enum { MOT_TRACKER_KCF = 0, MOT_TRACKER_KALMAN = 1 }; /* trackers/kcf.cpp | trackers/kalman.cpp */

/* FHOG arithmetic flavour.  INTEL_APPROX reproduces the as-compiled reference
 * (libhog/sse.hpp:40-41 rcpps/rsqrtps, bit-exact table emulation); EXACT uses
 * 1/x and 1/sqrt(x). */
enum { MOT_FHOG_INTEL_APPROX = 0, MOT_FHOG_EXACT = 1 };

enum { MOT_FFT_AUTO = 0, MOT_FFT_GENERIC = 1 }; /* AUTO: radix-4x5 register FFT where the size allows */

typedef struct mot_config {
    int device;        /* HIP device ordinal */
    int tracker_kind;  /* MOT_TRACKER_* */
    int fhog_mode;     /* MOT_FHOG_* */
    int fft_mode;      /* MOT_FFT_* */
    int max_tracks;    /* capacity of the live-track list (reference: 256, td.cpp:12) */
    int max_dets;      /* capacity of a detection list (reference: 128, cnntype.h:46) */
    int rank, world;   /* track sharding: this context holds the KCF/Kalman state of the tracks rank `rank` owns.  Device-resident loop: a spawning track goes to the
                        * rank with the fewest live tracks (lowest rank on a tie; replicated, deterministic), so a rank never owns more than
                        * ceil(max_tracks / world) tracks; host-orchestrated mot_step_begin / mot_step_finish: tid % world */
    void* stream;      /* hipStream_t to launch on, or NULL to create a private stream */
    int dev_rows, dev_cols; /* template size of the device-resident loop (mot_step_frame_device); 0 = 80 */
    int dev_size_lo, dev_size_hi; /* device-resident loop with per-track template sizes (the reference freezes rows / cols at tracker_new
                          from the spawning detection, kcf.cpp:148-152, td.cpp:626-627): one pool per SQUARE size in lo..hi (at most 128
                          sizes, all on the same side of the LDS-resident / HBM-slab split); a detection of another size spawns no
                          track (counted).  0, 0 = the single dev_rows x dev_cols template */
    int reserved[2];   /* must be zero */
} mot_config;

typedef struct mot_ctx mot_ctx; /* opaque */

void mot_config_default(mot_config* cfg);
int mot_ctx_create(const mot_config* cfg, mot_ctx** out);
int mot_ctx_destroy(mot_ctx* ctx);
const char* mot_last_error(void);
void* mot_ctx_stream(mot_ctx* ctx); /* the hipStream_t all work is enqueued on */
int mot_ctx_sync(mot_ctx* ctx);

/* ---- frame binding (replaces the cv::Mat data pointer of td.cpp:330,350) ----
 * 1280x720x3 u8 BGR, row stride 3840 (top/drawlib.c:9-10). */
int mot_frame_upload(mot_ctx* ctx, const uint8_t* host_bgr);      /* H2D copy into the context's frame */
int mot_frame_bind_device(mot_ctx* ctx, const void* device_bgr);  /* use a frame already resident in HBM */

/* ---- per-stage batch calls: one launch for n tracks ----------------------
 * ids are handles returned by mot_tracks_new.  boxes are host arrays.
 * mot_tracks_new replaces tracker_new (+ the first tracker_update with eta=1
 * issued by td.cpp:631-640 for KCF; needs a bound frame).
 * mot_predict_batch replaces the loop td.cpp:344-384 (crop+resize, tracker_predict);
 *   clamp != 0 applies td.cpp:378-381 to the returned boxes.
 * mot_update_batch replaces td.cpp:512-582's tracker_update calls (crop at boxes[i]). */
int mot_tracks_new(mot_ctx* ctx, const bbox_t* boxes, int n, int* ids_out);
int mot_predict_batch(mot_ctx* ctx, const int* ids, int n, bbox_t* boxes_out, int clamp);
int mot_update_batch(mot_ctx* ctx, const int* ids, int n, const bbox_t* boxes);
int mot_delete_batch(mot_ctx* ctx, const int* ids, int n);

/* Same stages fed with caller-supplied gray patches instead of the bound frame:
 * patches[i] is the column-major rows x cols float patch the reference passes as
 * `float* rgb` (trackers/kcf.cpp:455-476).  Used by the per-object drop-in layer.
 * Completion: every call copies the caller's patches and boxes before it returns.  A predict returns the boxes and therefore waits
 * for its kernel.  An UPDATE of at most 8 tracks returns with its launch queued on the context's stream (pinned, device-mapped staging
 * in two halves; the next call does not wait for it) -- the model is updated for every later call on this context, which the stream orders behind
 * it, and a device-side failure surfaces at the next call that waits (mot_ctx_sync, any predict).  td.cpp's update loop
 * (td.cpp:512-582: crop + resize on the host, then tracker_update, per object) thereby overlaps its host work with the previous
 * object's kernel. */
int mot_tracks_new_nofirst(mot_ctx* ctx, const bbox_t* boxes, int n, int* ids_out); /* tracker_new only */
int mot_predict_batch_patches(mot_ctx* ctx, const int* ids, int n, const float* const* patches, bbox_t* boxes_out);
int mot_update_batch_patches(mot_ctx* ctx, const int* ids, int n, const float* const* patches, const bbox_t* boxes);

/* ---- association: td.cpp:386-470 + trackers/hungarian/hungarian.cpp:29 ----
 * mot_assign: builds the td.cpp cost matrix on device from the two box lists
 *   (rows = the smaller side exactly as td.cpp:388-457), runs Munkres, returns
 *   assignment[row] = col or -1 (nRows = min(nT,nD) entries... see below) and the cost.
 *   assigned_trackers[nT] / assigned_detected[nD] are the scatter of td.cpp:472-502.
 * mot_assignment_optimal: Munkres on a caller-supplied column-major float64 cost
 *   matrix -- same contract as assignmentoptimal(). */
int mot_assign(mot_ctx* ctx, const bbox_t* trk, int nT, const bbox_t* det, int nD,
               int* assigned_trackers, int* assigned_detected, double* cost_out);
int mot_assignment_optimal(mot_ctx* ctx, int* assignment, double* cost, const double* dist, int nRows, int nCols);
int mot_cost_matrix(mot_ctx* ctx, const bbox_t* trk, int nT, const bbox_t* det, int nD, double* dist_out);

/* ---- whole tracker-thread iteration (td.cpp:344-644) -----------------------
 * Runs predict -> clamp -> cost -> Munkres -> update assigned / unassigned ->
 * delete lost -> spawn, on the bound frame.  Outputs (all optional, host):
 *   predicted[n_before]          clamped predicted boxes, old track order
 *   assigned_trackers[n_before]  detection index or -1, old track order
 *   live_boxes / live_tids       the track list after lifecycle (n returned via n_live)
 * In a sharded context (world > 1) the caller must pass gathered predictions via
 * mot_step_begin / mot_step_finish instead (see below). */
int mot_step_frame(mot_ctx* ctx, const bbox_t* dets, int nD,
                   bbox_t* predicted, int* assigned_trackers, int* n_before,
                   bbox_t* live_boxes, unsigned* live_tids, int* n_live);

/* The same iteration fed the way td.cpp feeds it: the detector's bbox_chain_t of this frame (td.cpp:326-333).
 * nbox outside 0..128 is MOT_ERR_ARG (the reference reads bbox[] unchecked). */
int mot_step_frame_chain(mot_ctx* ctx, const bbox_chain_t* detected,
                         bbox_t* predicted, int* assigned_trackers, int* n_before,
                         bbox_t* live_boxes, unsigned* live_tids, int* n_live);

/* Split form for multi-GPU: begin = predict the local shard into a device
 * buffer laid out for one all-gather; finish = association + update + lifecycle.
 *   mot_step_begin   writes ceil(max_tracks/world) bbox_t slots for this rank to
 *                    *local_boxes_dev (device pointer owned by the context).
 *   mot_step_finish  takes the all-gathered buffer (world * slots bbox_t, device)
 *                    -- for world == 1 pass the pointer returned by begin. */
int mot_step_begin(mot_ctx* ctx, void** local_boxes_dev, int* slots_per_rank);
int mot_step_finish(mot_ctx* ctx, const void* gathered_boxes_dev, const bbox_t* dets, int nD,
                    bbox_t* predicted, int* assigned_trackers, int* n_before,
                    bbox_t* live_boxes, unsigned* live_tids, int* n_live);

/* ---- steady-state device-resident loop (bench / serving) -------------------
 * Same per-frame semantics as mot_step_frame, but detections come from device
 * memory, nothing is copied back and no host synchronisation happens: the
 * frame's kernels are only enqueued.  Lifecycle (delete/spawn) is executed on
 * device as well.  mot_live_count() synchronises and returns the track count.
 * Lifetime of the inputs: frame_dev and dets_dev are read by kernels on the context's stream AND by a detection-feature launch on a
 * second stream of the context; every device-resident step call ends by ordering the context's stream behind that launch, so work the
 * caller enqueues on mot_ctx_stream() AFTER the call (the next frame's upload into the same buffer, an overlay, ...) may overwrite
 * both.  Touching them from any other stream needs mot_ctx_sync() (or an event recorded on the context's stream behind the call). */
int mot_step_frame_device(mot_ctx* ctx, const void* frame_dev, const void* dets_dev /* bbox_t[nD] */, int nD);
/* The same step with one frame of look-ahead: when the NEXT frame and its detection list are already in device memory (the reference's
 * detector thread runs ahead of the tracker thread through 64-slot rings, td.cpp:56-80,218-223), their detection features -- which
 * depend on that frame and its boxes only -- are computed beside THIS frame's association chain.  Results are identical to
 * mot_step_frame_device; the next call must pass the same pointers and count to benefit (anything else is computed as usual).
 * next_frame_dev / next_dets_dev may be null (end of stream).  Both frames' memory must stay valid until their step has executed:
 * the announced frame and list are read from the moment THIS call is enqueued until the call that processes them has ended (in stream
 * order, as above); announcing one frame and then passing another is allowed and costs one extra stream wait. */
int mot_step_frame_device_ahead(mot_ctx* ctx, const void* frame_dev, const void* dets_dev, int nD,
                                const void* next_frame_dev, const void* next_dets_dev, int next_nD);
int mot_step_begin_device(mot_ctx* ctx, const void* frame_dev, void** local_boxes_dev, int* slots_per_rank);
int mot_step_finish_device(mot_ctx* ctx, const void* gathered_boxes_dev, const void* dets_dev, int nD);
/* The same frame step fed from HOST memory (td.cpp:326-333: the tracker thread receives each frame and its detection list from the
 * capture / detector side): the 2.76 MB frame and the boxes are uploaded on a copy stream of the context into one of three device
 * buffers, so the upload of frame f + 2 (and the detection features of frame f + 1) overlap the association of frame f; the call only
 * enqueues.  host_bgr / host_dets should be
 * pinned (hipHostMalloc / cudaHostAlloc'ed) and must stay valid until the frame has been executed (mot_ctx_sync / mot_live_*). */
int mot_step_frame_host(mot_ctx* ctx, const uint8_t* host_bgr, const bbox_t* host_dets, int nD);
/* The sharded frame as ONE native call (multi-GPU hosts written in C/C++, e.g. a td.cpp-style tracker thread per GPU): predict of the
 * local shard, ONE in-place ncclAllGather of the predicted bbox_t segments (RCCL over xGMI; `nccl_comm` is the caller's ncclComm_t
 * for this rank, created with world = cfg.world ranks), replicated association + lifecycle, local updates -- all enqueued on the
 * context's stream, no host synchronisation.  librccl is bound at run time (dlopen), so single-GPU builds and hosts need no RCCL. */
int mot_step_frame_sharded(mot_ctx* ctx, const void* frame_dev, const void* dets_dev, int nD, void* nccl_comm);
/* The sharded step with one frame of look-ahead (as mot_step_frame_device_ahead): every rank computes the NEXT frame's detection features
 * beside this frame's replicated association chain instead of inside the frame.  Same results; the next call must pass the announced
 * pointers and count to benefit.  mot_step_begin_device_ahead is the two-call form (the caller runs the all-gather itself). */
int mot_step_frame_sharded_ahead(mot_ctx* ctx, const void* frame_dev, const void* dets_dev, int nD,
                                 const void* next_frame_dev, const void* next_dets_dev, int next_nD, void* nccl_comm);
/* mot_step_finish_device must then be given the SAME dets_dev and nD as this call (the frame's detection spectra are computed, or adopted
 * from the look-ahead launch, for that list): a different pointer or count is refused with MOT_ERR_ARG. */
int mot_step_begin_device_ahead(mot_ctx* ctx, const void* frame_dev, const void* dets_dev, int nD,
                                const void* next_frame_dev, const void* next_dets_dev, int next_nD, void** local_boxes_dev, int* slots_per_rank);
/* Checkpoint / resume of the device-resident loop (single template size): mot_state_save writes one flat host record -- live list and counters,
 * per-slot tracker state (KCF: model, alpha, pos, scale, flags, response maps, pending detections and the spectra they refer to; Kalman: x, P) --
 * and reports its size through *bytes (host_buf may be null to ask for it); mot_state_load puts it into a FRESH context of the same
 * configuration (tracker kind, capacities, template size, rank / world, frame structure switches), after which the loop continues bit for bit as
 * the saved one would have.  Both synchronise the context (all its streams). */
int mot_state_save(mot_ctx* ctx, void* host_buf, size_t cap_bytes, size_t* bytes);
int mot_state_load(mot_ctx* ctx, const void* host_buf, size_t bytes);
int mot_live_count(mot_ctx* ctx, int* n_live);
int mot_live_tracks(mot_ctx* ctx, bbox_t* boxes, unsigned* tids, int* ages, int* n_live);

/* ---- detector post-processing (detectors/yolo3.cpp:141-356, 490-547) --------------
 * From the three raw output tensors of the YOLOv3 network (device memory, NHWC [grid << s][grid << s][3 * (5 + classes)], s = 0, 1, 2)
 * to the detection list the tracker thread consumes: sigmoid / exp decode against obj_thresh, letterbox correction, per-class NMS
 * (the reference's own exchange sort and suppression loop, order-exact), clamp + validity filter.  The boxes are written to
 * dets_dev_out (at most cap) in the reference's emit order and their number to *n_dev_out (device int), ready for
 * mot_step_frame_device; with host_chain_out the reference's bbox_chain_t is filled as well (synchronises; at most 128 boxes).
 * The network itself stays the caller's (north_star: "detectors/yolo3 stays a stub"). */
typedef struct {            /* bit-identical to top/cnntype.h:49-54 */
    float obj_thresh;
    float nms_thresh;
    int anchors[18];
} yolo3_options_t;
int mot_yolo_postprocess(mot_ctx* ctx, const float* head0_dev, const float* head1_dev, const float* head2_dev, int tensor_h, int tensor_w,
                         int num_classes, int image_h, int image_w, const yolo3_options_t* opt, bbox_t* dets_dev_out, int cap, int* n_dev_out,
                         bbox_chain_t* host_chain_out);
/* Candidates above obj_thresh beyond the workspace (4096) are dropped in arrival order, which the reference's unbounded vector never does
 * (yolo3.cpp:176-201): the count is latched on the device.  Synchronises; MOT_ERR_CAPACITY if any call of this context dropped candidates
 * (mot_yolo_postprocess reports it itself when host_chain_out is given).  dropped_candidates may be NULL. */
int mot_yolo_status(mot_ctx* ctx, int* dropped_candidates);

/* ---- overlay: the tracker thread's drawing step (td.cpp:647-733) ----------------
 * Three nested rectangle outlines per track (drawRect, top/drawlib.c:97-151) in colormap[hashcolor(tid + 1) & 255] (td.cpp:295-304,
 * 620, 655-699), drawn into a 1280x720x3 frame in device memory, later tracks over earlier ones.  mot_overlay_draw takes host
 * arrays (any context); mot_overlay_live draws the live list of the device-resident loop without any copy.  Enqueued on the
 * context's stream; the frame must not be the input of a frame step that is still in flight. */
int mot_overlay_draw(mot_ctx* ctx, void* frame_dev, const bbox_t* boxes, const unsigned* tids, int n);
int mot_overlay_live(mot_ctx* ctx, void* frame_dev);

/* ---- introspection for parity tests --------------------------------------- */
int mot_get_response(mot_ctx* ctx, int id, float* out, int* f_rows, int* f_cols); /* kcf_t::response (kcf.cpp:58) */
int mot_get_model(mot_ctx* ctx, int id, float* xm_out /* 31*f_cols*(f_rows/2+1)*2 */, float* alpha_out);
int mot_get_kalman_state(mot_ctx* ctx, int id, double* x6, double* P36);
int mot_get_pos(mot_ctx* ctx, int id, bbox_t* pos);
/* device-resident loop: kcf_t::response (kcf.cpp:58) of the i-th LIVE track (order of mot_live_tracks) as left by the most recent
 * predict -- the 1e-4 peak check at full track counts */
int mot_live_response(mot_ctx* ctx, int live_index, float* out, int* f_rows, int* f_cols);
/* ... and its model as it is in device memory right now (kcf_t::xf_md, alpha, pos, scale: kcf.cpp:50-62): xm_out 31 * f_cols * (f_rows/2+1)
 * complex, alpha_out f_cols * (f_rows/2+1) floats, scale2 (horizontal, vertical), pending_det = the detection of the last frame whose
 * spectrum the NEXT predict will blend in first (deferred blend; -1: none).  Any output may be null.  Synchronises the context (all streams).
 * State dump for replaying / comparing runs bit by bit (tools/lookahead_soak.py). */
int mot_live_model(mot_ctx* ctx, int live_index, float* xm_out, float* alpha_out, bbox_t* pos, float* scale2, int* first_update, int* pending_det);
/* debug: enable / read the per-phase time stamps (100 MHz ticks) of workgroup 0 of the device-loop KCF kernels:
 * [0] start [1] crop [2] gradient [3] histogram [4] norm [5] channels [6] DFT [7] end */
/* debug: duration of the predict launch itself (the kernel's own begin / end stamps) while the ordinary device-resident step calls run:
 * arm n pairs, step n frames, read the n durations (synchronises).  What rocprofv3 reports for the launch in the timed configuration. */
/* debug: stream-ordered (no host synchronisation) copy of the device loop's small state -- counts, live list, this frame's predicted boxes, KCF
 * pos / pending detection / first-update flag by slot, head of the residual-update list, association header -- into caller-owned device memory;
 * *bytes receives the record size (dst_dev may be null to ask for it).  Layout: csrc/mot_devloop.hip.  For soak tools. */
int mot_debug_snapshot(mot_ctx* ctx, void* dst_dev, size_t* bytes);
/* debug (MOT_TRACE=1 in the environment when the context is created): the per-workgroup records the predict / residual-update launches of the last 16
 * frames left behind -- 8 ints each, [kind 0 predict | 1 update][frame & 15][slot]; layout: csrc/mot_dev.h, KcfLaunch::trace.  *n_ints receives the
 * size (0: tracing off).  One 32-byte store per workgroup: light enough to leave a timing-dependent failure alive (tools/lookahead_soak.py --trace). */
int mot_debug_trace_read(mot_ctx* ctx, int* out, size_t cap_ints, size_t* n_ints);
/* debug / bench: stage times of the two-call (mot_step_begin_device[_ahead] + mot_step_finish_device) and of the sharded step on this rank.
 * enable, step ONE frame, then call again with stage_ms4 to read [0] predict launch [1] all-gather (end of the predict -> finish call)
 * [2] association chain (scatter, row scan, solver / emulation, lifecycle) [3] residual update launch, in ms (HIP events on the context's
 * stream; reading synchronises).  bench.py --gpus N prints them per rank. */
int mot_debug_profile_stages(mot_ctx* ctx, int enable, float* stage_ms4);
int mot_debug_predict_timing(mot_ctx* ctx, int n_pairs);
int mot_debug_predict_times(mot_ctx* ctx, float* out_ms, int cap, int* n);
int mot_debug_kcf_phases(mot_ctx* ctx, int enable, long long* predict8, long long* update8);
/* Host-side scheduling state machine that decides whether a launch also submits the dense LAP solver's kernels (never a result):
 * one step on a caller-owned int[16] (h[0] bit 1 = "a recent launch needed the dense solver", written by the device in the product);
 * returns 1 if this launch would submit them.  Pure host code: callable without a GPU (tests/test_abi_symbols.py). */
int mot_debug_dense_arming(int* h16, int nD);
/* counters of the most recent Munkres launch: [0] step-4 augmentations [1] step-5 updates [2] step-3 sweeps
 * [3] step-5 passes with covered rows; [4..7] step-5 split in 10 ns ticks (helper workgroups: publish, wait for the
 * minimum, wait for the update, merge; one workgroup: pass 1, reduce, uncovered rows, covered rows); [8..12] device time
 * in 10 ns ticks: init, step 3 + 4, -, step 5, total; [13] shader MHz; [14] uncovered columns at the first step 5;
 * [15] 0, or which helper hand-off timed out */
int mot_get_assoc_stats(mot_ctx* ctx, int* out16);
/* assignment fast path in front of the Munkres emulation (exact sparse solver + uniqueness certificate; the reference's
 * assignmentoptimal, hungarian.cpp:29-368, returns the unique optimum whenever there is one):
 * [0..15] most recent launch: [0] 0 = certified (emulation skipped), 1 = solver gave up / not applicable, 2 = dual check
 *   failed, 3 = too many near-tight edges, 4 = tied optima (order-exact emulation ran); [1] solver rounds [2] free rows after
 *   the greedy start [3] searches [4] commits [5] near-tight edges [6] nodes on cycles [7] solver time in 10 ns ticks;
 *   sparse order-exact emulation (runs when not certified): [8] 0 ok / 1 minimum outside the candidate lists / 2 n.a.,
 *   [9] augmentations [10] step-5 passes [11] step-3 events [12..14] 10 ns ticks: steps 3+4, step 5, total;
 *   [15] what decided this launch: 0 certificate, 1 sparse emulation (accepted by its after-the-fact check), 2 dense emulation
 * [16..20] cumulative launch counts of this context by certificate outcome 0..4; [24] sparse emulation accepted, [25] refused;
 * [21..23] device loop, cumulative: tie frames committed PROVISIONALLY (one pair of tracks could swap their detections at equal cost: the
 *   solver's optimum is committed at once, the next predict computes both alternatives for the pair, and the order-exact emulation -- running
 *   beside that predict -- only names the one the reference returns), swaps applied, frames whose bit the dense emulation had to decide;
 * [26..28] dense solver (frames whose far matches defeat the sparse one: detector misses + false positives) in the most recent launch:
 *   settled columns, free rows after the greedy start, time in 10 ns ticks; [29] launches in which it ran, [30] ... and were certified */
int mot_get_lap_stats(mot_ctx* ctx, int* out32);
/* debug: thread 0's time stamps along the sparse emulation's cycles of the most recent launch (MOT_MK_TIMING=1): out[0] = count, then tag << 56 | 10 ns ticks */
int mot_debug_assoc_trace(mot_ctx* ctx, long long* out, int n);
/* FHOG only (libhog/fhog.h:16-38): H[32][w/4][h/4] for one column-major h x w patch. */
int mot_fhog_extract(mot_ctx* ctx, const float* patch, int h, int w, float* H_out, int windowed);
/* The three C helpers of the reference's tracker thread (top/td.cpp:235-261, top/drawlib.c:97-151, 192-240, 542-637) on the device, with the
 * reference's own argument meaning -- HOST pointers in and out, synchronous; the drop-in libraries export them under the reference's names
 * (rgb2Gray, bilinearInterpolationGray, drawRect), so a td.cpp link needs no object of the reference.  Boxes outside the 1280 x 720 frame are
 * MOT_ERR_ARG (the reference reads / writes unchecked). */
int mot_helper_rgb2gray(mot_ctx* ctx, float* pgra, const uint8_t* prgb, int left, int top, int right, int bottom);
int mot_helper_bilinear_gray(mot_ctx* ctx, float* pdst, const float* psrc, int rows_s, int cols_s, int rows_d, int cols_d);
int mot_helper_draw_rect(mot_ctx* ctx, uint8_t* fbuf, int left, int top, int right, int bottom, unsigned RGB);
/* crop + gray + resize only (top/td.cpp:348-364) on the bound frame. */
int mot_crop_patch(mot_ctx* ctx, const bbox_t* box, int rows, int cols, float* patch_out);

/* ---- timing helpers: HIP events on the context's stream --------------------
 * (bench.py measures kernels on the stream they are launched on.) */
int mot_timer_create(mot_ctx* ctx, int n_events);
int mot_timer_record(mot_ctx* ctx, int idx);
int mot_timer_elapsed_ms(mot_ctx* ctx, int idx_start, int idx_stop, float* ms);
/* per-stage device time of the most recent mot_profile_frame_device() call:
 * [0]=predict kernel [1]=cost/init [2]=munkres [3]=scatter/lifecycle [4]=update kernel, in ms */
int mot_profile_frame_device(mot_ctx* ctx, const void* frame_dev, const void* dets_dev, int nD, float* stage_ms5);

#ifdef __cplusplus
}
#endif
#endif /* MOT_ABI_H */
