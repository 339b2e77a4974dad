// dropin.cpp -- per-object drop-in layer (see include/mot_dropin.hpp).
// Built twice: -DMOT_DROPIN_KIND=0 (KCF) and =1 (Kalman).
#include "../../include/mot_dropin.hpp"
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>

#ifndef MOT_DROPIN_KIND
#error "define MOT_DROPIN_KIND (0 = KCF, 1 = Kalman)"
#endif

namespace {
mot_ctx* g_ctx = nullptr;
std::once_flag g_once;

[[noreturn]] void die(const char* what)
{
    // The reference interface is void: it has no way to report an error
    // (SURVEY 8b).  Failing loudly beats returning garbage.
    std::fprintf(stderr, "mot_dropin: %s: %s\n", what, mot_last_error());
    std::abort();
}

mot_ctx* ctx()
{
    std::call_once(g_once, [] {
        mot_config cfg; mot_config_default(&cfg);
        cfg.tracker_kind = MOT_DROPIN_KIND;
        if (const char* d = std::getenv("MOT_DEVICE")) cfg.device = std::atoi(d);
        cfg.max_tracks = 256;   // MAX_OBJECTS_PER_FRAME, td.cpp:12
        cfg.max_dets = 256;
        if (mot_ctx_create(&cfg, &g_ctx) != MOT_OK) die("mot_ctx_create");
    });
    return g_ctx;
}

struct Handle { int id; };
} // namespace

void* tracker_new(bbox_t* pbox)
{
    Handle* h = new Handle;
    if (mot_tracks_new_nofirst(ctx(), pbox, 1, &h->id) != MOT_OK) die("tracker_new");
    return h;
}

void tracker_predict(void* ptracker, float* rgb, bbox_t* pbox)
{
    const float* patches[1] = { rgb };
    if (mot_predict_batch_patches(ctx(), &static_cast<Handle*>(ptracker)->id, 1, patches, pbox) != MOT_OK) die("tracker_predict");
}

void tracker_update(void* ptracker, float* rgb, bbox_t* pbox)
{
    const float* patches[1] = { rgb };
    if (mot_update_batch_patches(ctx(), &static_cast<Handle*>(ptracker)->id, 1, patches, pbox) != MOT_OK) die("tracker_update");
}

void tracker_delete(void* ptracker)
{
    Handle* h = static_cast<Handle*>(ptracker);
    if (mot_delete_batch(ctx(), &h->id, 1) != MOT_OK) die("tracker_delete");
    delete h;
}

void assignmentoptimal(int* assignment, double* cost, double* distMatrixIn, int nOfRows, int nOfColumns)
{
    if (mot_assignment_optimal(ctx(), assignment, cost, distMatrixIn, nOfRows, nOfColumns) != MOT_OK) die("assignmentoptimal");
}

// The C helpers of the reference's tracker thread (td.cpp:235-261; the reference implements them in top/drawlib.c): exported under the
// reference's names with C linkage, so that linking td.cpp against this library leaves nothing of the reference's tracker side to compile.
extern "C" void rgb2Gray(float* pgra, uint8_t* prgb, int32_t left, int32_t top, int32_t right, int32_t bottom)
{
    if (mot_helper_rgb2gray(ctx(), pgra, prgb, left, top, right, bottom) != MOT_OK) die("rgb2Gray");
}

extern "C" void bilinearInterpolationGray(float* pdst, const float* psrc, int rows_s, int cols_s, int rows_d, int cols_d)
{
    if (mot_helper_bilinear_gray(ctx(), pdst, psrc, rows_s, cols_s, rows_d, cols_d) != MOT_OK) die("bilinearInterpolationGray");
}

extern "C" void drawRect(uint8_t* fbuf, int32_t left, int32_t top, int32_t right, int32_t bottom, uint32_t RGB)
{
    if (mot_helper_draw_rect(ctx(), fbuf, left, top, right, bottom, RGB) != MOT_OK) die("drawRect");
}
