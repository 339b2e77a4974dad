// mot_ctx.h -- internal host-side types shared by mot_ctx.hip and mot_devloop.hip.
#pragma once
#include "mot_dev.h"
#include "mot_env.h"
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <memory>
#include <string>
#include <vector>

namespace mot_impl {

int fail(int code, const char* fmt, ...);
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return mot_impl::fail(MOT_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

inline bool debug_alloc() { static const bool v = getenv("MOT_DEBUG_ALLOC") != nullptr; return v; }
template <typename T> struct DevBuf {
    T* p = nullptr; size_t n = 0;
    hipError_t alloc(size_t count)
    {
        release(); n = count;
        if (!count) return hipSuccess;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e == hipSuccess && debug_alloc()) fprintf(stderr, "mot alloc %p .. %p  %zu bytes (%zu x %zu)\n", (void*)p, (void*)((char*)p + count * sizeof(T)), count * sizeof(T), count, sizeof(T));   // (debug) MOT_DEBUG_ALLOC=1: maps a fault address to a buffer
        if (e == hipSuccess && poison_byte() >= 0) { e = hipMemset(p, poison_byte(), count * sizeof(T)); if (e == hipSuccess) e = hipDeviceSynchronize(); }   // (debug) MOT_POISON, mot_env.h
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    ~DevBuf() { release(); }
};
template <typename T> struct PinBuf {
    T* p = nullptr; size_t n = 0;
    hipError_t alloc(size_t count) { release(); n = count; return count ? hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault) : hipSuccess; }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; n = 0; }
    ~PinBuf() { release(); }
};

struct PoolHost {
    KcfPool dev{};
    int cap = 0;
    std::vector<int> free_slots;
    DevBuf<float2> xm; DevBuf<float> alpha; DevBuf<bbox_t> pos; DevBuf<float2> scale; DevBuf<int> first; DevBuf<float> response;
    DevBuf<float> cos_win, yf_re; DevBuf<float2> tw_r, tw_c; DevBuf<float> gscratch; DevBuf<float> mf_rows, mf_cols, mf_cols2;
};

struct TrackRec { int kind; int pool; int slot; int rows, cols; bool live; };

// Staging of the per-object calls that bring their own patch (tracker_predict / tracker_update of the drop-in interface, kcf.cpp:455-476: a
// batch of <= 8): pinned, device-mapped, the kernels read and write it in place (zero-copy).  TWO halves with an event each, so that an
// UPDATE can return as soon as its launch is queued -- the caller's patch has been copied, nothing comes back -- while the next call stages
// into the other half: td.cpp's update loop (crop + resize on the host, then tracker_update, per object: td.cpp:512-582) overlaps its host
// work with the previous object's kernel.  A predict returns a box and still waits.  Nothing else uses these buffers.
struct ZcRing {
    static constexpr int kItems = 8;
    PinBuf<int> slots; PinBuf<bbox_t> boxes_a, boxes_b; PinBuf<float> patches;
    int* d_slots = nullptr; bbox_t* d_boxes_a = nullptr; bbox_t* d_boxes_b = nullptr; float* d_patches = nullptr;
    size_t half_floats = 0;                          // patch capacity of one half
    hipEvent_t ev[2] = {nullptr, nullptr}; bool busy[2] = {false, false}; int next = 0;
    bool tried = false, ok = false;
    ~ZcRing() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); }
};

struct LiveInfo {                    // tracker_info_t (td.cpp:271-290)
    int id; unsigned tid; int age, visible, invisible; bbox_t bbox;
};

// roctx ranges around the host-side enqueue of every stage (tracing).  Opt-in (MOT_ROCTX=1): the marker
// library (librocprofiler-sdk-roctx, what `rocprofv3 --marker-trace` records) is dlopen'ed on first use, so the product has
// no link-time dependency on a profiler and pays one predictable branch per stage when tracing is off.
struct RoctxRange {
    explicit RoctxRange(const char* name);
    ~RoctxRange();
    bool on;
};

struct DevLoop;                      // device-resident frame loop state (mot_devloop.hip)
void devloop_destroy(DevLoop*);
struct YoloWs;                       // detector post-processing workspace (yolo_post.hip)
void yolo_destroy(YoloWs*);

} // namespace mot_impl

struct mot_ctx {
    mot_config cfg{};
    hipStream_t stream = nullptr; bool own_stream = false;
    mot_impl::DevBuf<uint8_t> frame_own; const uint8_t* frame = nullptr;
    mot_impl::DevBuf<uint16_t> sse_tab;
    std::vector<std::unique_ptr<mot_impl::PoolHost>> pools;
    KalmanPool kal{}; mot_impl::DevBuf<double> kal_x, kal_P; std::vector<int> kal_free;
    std::vector<mot_impl::TrackRec> tracks;           // id -> record
    // staging (capacity = max_tracks + max_dets)
    int stage_cap = 0;
    mot_impl::DevBuf<int> d_slots; mot_impl::DevBuf<bbox_t> d_boxes_a, d_boxes_b, d_dets; mot_impl::DevBuf<float> d_patches; size_t patches_cap = 0;
    // device-side addresses of the pinned staging buffers (zero-copy path of small batches, run_batch); null when the mapping is not available
    int* zc_slots = nullptr; bbox_t* zc_boxes_a = nullptr; bbox_t* zc_boxes_b = nullptr;
    mot_impl::ZcRing zc_ring;
    mot_impl::PinBuf<int> h_slots; mot_impl::PinBuf<bbox_t> h_boxes_a, h_boxes_b; mot_impl::PinBuf<int> h_assign; mot_impl::PinBuf<int> h_hint; mot_impl::PinBuf<double> h_cost; mot_impl::PinBuf<float> h_patches;
    // association
    AssocWs assoc{}; mot_impl::DevBuf<double> a_dist; mot_impl::DevBuf<unsigned long long> a_zr, a_zc, a_linemin; mot_impl::DevBuf<int> a_assign, a_status; mot_impl::DevBuf<double> a_cost;
    mot_impl::DevBuf<double> a_user; mot_impl::DevBuf<unsigned long long> a_ctl; mot_impl::DevBuf<unsigned char> a_lap;
    // frame loop (td.cpp:306-748), host-orchestrated mode
    std::vector<mot_impl::LiveInfo> live; unsigned next_tid = 0;
    mot_impl::DevBuf<bbox_t> d_gather; int slots_per_rank = 0; bool step_open = false;
    // device-resident mode
    mot_impl::DevLoop* devloop = nullptr;
    mot_impl::YoloWs* yolo = nullptr;
    // overlay (td.cpp:647-733): per-pixel "last track to draw here" stamps, tagged with a per-call epoch
    mot_impl::DevBuf<unsigned> ov_stamp; unsigned ov_epoch = 0; mot_impl::DevBuf<bbox_t> ov_boxes; mot_impl::DevBuf<unsigned> ov_tids;
    // per-call staging of the td.cpp helper entry points (helper_kernels.hip)
    mot_impl::DevBuf<uint8_t> hlp_bytes; mot_impl::DevBuf<float> hlp_f0, hlp_f1;
    // timers / debug
    std::vector<hipEvent_t> events;
    mot_impl::DevBuf<long long> dbg; bool dbg_on = false;
};

namespace mot_impl {
int ensure_device(mot_ctx* c);
int get_pool(mot_ctx* c, int rows, int cols, int* out_idx, bool shared_scratch = false);   // shared_scratch: no HBM slab of its own (the caller points gscratch at a shared one)
int devloop_check(mot_ctx* c);   // mot_devloop.hip
int devloop_flush(mot_ctx* c);   // mot_devloop.hip: the patch step a provisionally committed frame still owes (round 6), enqueued on the context's stream
// the synchronisation of every read-back of device-loop state: never shows the caller a provisional frame
#define MOT_SYNC_CTX(c) do { int rc_ = mot_impl::devloop_flush(c); if (rc_) return rc_; HIPCHK(hipStreamSynchronize((c)->stream)); } while (0)
int overlay_run(mot_ctx* c, void* frame_dev, const bbox_t* boxes_dev, const unsigned* tids_dev, const int* n_dev, int n_max);   // overlay_kernels.hip
}
