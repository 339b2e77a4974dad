// mot_env.h -- host-side process configuration shared by the launchers: the MOT_* environment switches, read ONCE (C++11 function-local
// static: initialisation is thread-safe, the object is immutable afterwards), and the per-DEVICE table of kernel attributes already set
// (hipFuncSetAttribute is per device; one tracker thread per GPU in one process is the advertised deployment).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>

namespace mot_impl {

// Round 6 removed MOT_ZC_ASYNC (per-object updates that wait for their kernel: lost the A/B, profiles/r05_zc_async_ab.log) and MOT_KCF_R1LDS (the round-3
// slab pipeline for sizes that have an R1-resident instance: 535 against 588 k updates/s at 148 px, round 5), and added MOT_PROV (0: frames wait for the
// emulation as in rounds 3-5 -- the A/B and the bit-equality reference of the provisional commits; 2: test hook).  Kept although their off-variant is slower,
// because each is the only way to FORCE a tier / structure the default path reaches rarely: MOT_LAP_TWO_BLOCK=0 (the sparse emulation as a launch of its
// own behind the solver: the structure every stream with the dense solver armed takes), MOT_MK_LAZY=0 (the reference's full reset after every augmentation:
// what the lazy reset falls back to when a step 5 queues more union requests than it holds, mk_sparse_body.h).
// Round 5 removed the switches whose off-variant had lost every measurement of rounds 2-4 and was covered by no test: MOT_LAP_FUSED (chip-wide
// dual / after-the-fact checks for box costs; caller matrices still take them), MOT_FEAT_BEFORE_ROWSCAN, MOT_H2D_MODE (hipMemcpyAsync uploads;
// pageable host memory still falls back to them), MOT_MID_IN_LAUNCH.  What is left is exercised by tests/test_gpu_variants.py.
struct EnvSwitches {
    int lap_min;             // MOT_LAP_FAST=0: fast path off (1 << 30); MOT_LAP_MIN: smallest problem (lines) it is used for; -1: the built-in default
    int dense_mode;          // MOT_LAP_DENSE: 0 never, 1 always, 2 (default) when the stream's recent frames needed it
    int two_block;           // MOT_LAP_TWO_BLOCK=0: the sparse emulation as a launch of its own behind the solver
    int mk_batch;            // MOT_MK_BATCH (low 16 bits; 0: one event per iteration, n > 1: batch threshold) | MOT_MK_LAZY=0 -> 0x40000000 | MOT_MK_TIMING=1 -> 0x20000000
    int helpers;             // MOT_MUNKRES_HELPERS: 0 off, 1 forced on, 2 default (by size); force_cov: =2 test hook
    int force_cov;
    int split_early_max;     // MOT_SPLIT_EARLY_MAX (-1: built-in default)
    int joined_launch;       // MOT_JOINED_LAUNCH=0: side-stream feature launch for small frames too
    int lookahead;           // MOT_LOOKAHEAD=0: mot_step_frame_device_ahead ignores its hint
    int split_update;        // MOT_SPLIT_UPDATE=0: fused update kernel
    int dft_mfma;            // MOT_DFT_MFMA=0: HBM-slab templates use the generic DFT instead of the MFMA products
    int dft_inplace;         // MOT_DFT_INPLACE=0: LDS-resident templates with the direct transforms keep the ping-pong buffer (region T)
    int k80;                 // MOT_KCF_K80: which kernels of an 80 x 80 px pool run with the geometry folded in (bit 0 predict, 1 feature, 2 update; default 7, 0: none);
                             // bit 3 (8; only in builds with -DMOT_KCF_SPARSE_VIEW=1, `make endcf`): the folded copy inside the out-of-line body of the sparse update
                             // kernel as well -- the instantiation hipcc miscompiles (register copies in front of a folded EXEC restore: kcf_update_sparse_run)
    int defer_blend;         // MOT_DEFER_BLEND=0: blend launch of its own
    int side_reserve;        // MOT_SIDE_RESERVE: CUs the side stream may not use (-1: default by template size)
    int prov;                // MOT_PROV=0: no provisional commits of two-row tie frames (round 6): the sparse emulation stays in the solver's launch and the frame waits for it;
                             // =2 (test hook): the patch step ignores the sparse emulation's answer and lets its dense emulation decide the swap bits
                             // =4 (bisecting aid): the stream-emulation chain without provisional commits -- every tie frame is decided and committed by the emulation's kernel
};

inline const EnvSwitches& env()
{
    static const EnvSwitches e = [] {
        auto geti = [](const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; };
        auto off = [](const char* name) { const char* v = getenv(name); return v && atoi(v) == 0; };
        EnvSwitches s{};
        s.lap_min = off("MOT_LAP_FAST") ? (1 << 30) : (getenv("MOT_LAP_MIN") ? (geti("MOT_LAP_MIN", 1) > 1 ? geti("MOT_LAP_MIN", 1) : 1) : -1);
        s.dense_mode = getenv("MOT_LAP_DENSE") ? (geti("MOT_LAP_DENSE", 0) ? 1 : 0) : 2;
        s.two_block = off("MOT_LAP_TWO_BLOCK") ? 0 : 1;
        s.mk_batch = (geti("MOT_MK_BATCH", 1) & 0xFFFF) | (off("MOT_MK_LAZY") ? 0x40000000 : 0) | (geti("MOT_MK_TIMING", 0) ? 0x20000000 : 0);
        s.helpers = getenv("MOT_MUNKRES_HELPERS") ? (geti("MOT_MUNKRES_HELPERS", 0) ? 1 : 0) : 2;
        s.force_cov = geti("MOT_MUNKRES_HELPERS", 0) == 2 ? 1 : 0;
        s.split_early_max = geti("MOT_SPLIT_EARLY_MAX", -1);
        s.joined_launch = off("MOT_JOINED_LAUNCH") ? 0 : 1;
        s.lookahead = off("MOT_LOOKAHEAD") ? 0 : 1;
        s.split_update = off("MOT_SPLIT_UPDATE") ? 0 : 1;
        s.dft_mfma = off("MOT_DFT_MFMA") ? 0 : 1;
        s.k80 = geti("MOT_KCF_K80", 7);                                  // bit 0 predict, 1 feature, 2 update kernels
        s.dft_inplace = off("MOT_DFT_INPLACE") ? 0 : 1;
        s.defer_blend = off("MOT_DEFER_BLEND") ? 0 : 1;
        s.side_reserve = geti("MOT_SIDE_RESERVE", -1);
        s.prov = geti("MOT_PROV", 1);
        return s;
    }();
    return e;
}

// ---- (debug) poisoning: turns reads of memory nobody wrote into deterministic failures --------------------------------------------
// MOT_POISON=<byte>      every device allocation of the library is filled with this byte (hipMalloc hands out recycled pages in a long-lived
//                        process and zero pages in a fresh one: a read-before-write shows up as a flake that depends on the process' history)
// MOT_LDS_POISON=<word>  a scribble kernel overwrites the LDS of every compute unit with this word in front of every kernel launch of the
//                        library (LDS keeps the previous workgroup's contents)
inline int poison_byte()
{
    static const int v = [] { const char* e = getenv("MOT_POISON"); return e ? (int)(strtol(e, nullptr, 0) & 0xFF) : -1; }();
    return v;
}
hipError_t lds_poison_launch(hipStream_t s, unsigned word);            // helper_kernels.hip
inline void lds_poison(hipStream_t s)
{
    static const long long v = [] { const char* e = getenv("MOT_LDS_POISON"); return e ? (long long)(strtoul(e, nullptr, 0) & 0xFFFFFFFFul) | (1ll << 40) : 0ll; }();
    if (v) (void)lds_poison_launch(s, (unsigned)(v & 0xFFFFFFFFll));
}

// hipFuncSetAttribute(fn, MaxDynamicSharedMemorySize, bytes) once per (device, kernel); safe from several threads / devices
inline hipError_t func_lds_once(const void* fn, int bytes)
{
    static std::mutex mu;
    static std::vector<std::pair<int, const void*>> done;
    int dev = 0; hipError_t e = hipGetDevice(&dev); if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    for (const auto& d : done) if (d.first == dev && d.second == fn) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.emplace_back(dev, fn);
    return e;
}

} // namespace mot_impl
