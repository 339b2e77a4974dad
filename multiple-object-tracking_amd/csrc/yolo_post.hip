// yolo_post.hip -- detector post-processing on device: what detectors/yolo3.cpp does between the network's three
// output tensors and the bbox_chain_t the tracker thread consumes -- decode_netout (:141-201: sigmoid / exp, score = class * objectness
// >= obj_thresh), correct_yolo_boxes (:203-254: letterbox correction, truncation to int), per-class NMS with the reference's own
// exchange sort (:256-356) and the clamp / validity filter of tensorRunB (:519-547).  The boxes land in device memory in the order
// the reference emits them, so they can feed mot_step_frame_device without a host round trip.
//
//   yolo_decode_kernel   chip-wide, one thread per (head, cell, anchor): every class score above the threshold appends a candidate
//                        tagged with its position in the reference's scan order (head, cell, anchor, class)
//   yolo_nms_kernel      one workgroup: candidates back into scan order (rank by counting), letterbox correction, then ONE THREAD PER
//                        CLASS runs the reference's exchange sort and suppression loop as written (their result depends on the
//                        order of equal scores, and on a quirk: the `is_suppressed` vector of do_nms is never reset between classes),
//                        then thread 0 emits class after class
// Float arithmetic follows the source expression by expression; expf is the device library's (the reference's is MSVC's), and the
// reference cannot be compiled here (Windows + TensorFlow headers): this row's parity is UNPINNED -- checked against the oracle
// restatement only.
#include "mot_ctx.h"

#define YOLO_CAND 4096          /* candidates above the threshold (all heads, classes) */
#define YOLO_MAXC 128           /* classes */

struct YoloCand { float x, y, u, w, s; int c; unsigned key; };
struct YoloDet { int xmin, ymin, xmax, ymax, classes; float objectness; };

namespace {

struct YoloArgs {
    const float* head[3]; int gh[3], gw[3]; int anc[18];
    int tensor_h, tensor_w, nc, image_h, image_w; float obj_thresh, nms_thresh;
    YoloCand* cand; int* ncand; YoloDet* det; YoloDet* cls; int* idx; int* sup; int* kept; int* nkept;
    bbox_t* out; int* nout; int cap;
};

__device__ __forceinline__ float sigm(float v) { return 1 / (1 + expf(-v)); }   // yolo3.cpp:170

__global__ void __launch_bounds__(256) yolo_decode_kernel(YoloArgs a)
{
    const int per = 5 + a.nc;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned keybase = 0;
    for (int h = 0; h < 3; h++) {
        const int cells = a.gh[h] * a.gw[h];
        if (t < cells * 3) {
            const int i = t / 3, b = t - 3 * i, row = i / a.gw[h], col = i - row * a.gw[h];
            const float* v = a.head[h] + ((size_t)i * 3 + b) * per;
            const float objectness = sigm(v[4]);
            const int* anchors = a.anc + (h == 0 ? 12 : h == 1 ? 6 : 0);                 // :512-514
            for (int j = 0; j < a.nc; j++) {
                const float scores = sigm(v[5 + j]) * objectness;
                if (scores >= a.obj_thresh) {
                    const int q = atomicAdd(a.ncand, 1);
                    if (q < YOLO_CAND) {
                        YoloCand c;
                        c.x = (col + sigm(v[0])) / a.gw[h];                                // :186-189
                        c.y = (row + sigm(v[1])) / a.gh[h];
                        c.u = anchors[2 * b + 0] * expf(v[2]) / a.tensor_w;
                        c.w = anchors[2 * b + 1] * expf(v[3]) / a.tensor_h;
                        c.s = scores; c.c = j; c.key = keybase + (unsigned)(t * a.nc + j);
                        a.cand[q] = c;
                    }
                }
            }
            return;
        }
        t -= cells * 3; keybase += (unsigned)(cells * 3 * a.nc);
    }
}

__global__ void __launch_bounds__(1024) yolo_nms_kernel(YoloArgs a)
{
    __shared__ int s_cnt[YOLO_MAXC], s_off[YOLO_MAXC + 1];
    const int tid = threadIdx.x;
    const int n = min(*a.ncand, YOLO_CAND);
    // scan order + correct_yolo_boxes (:203-254)
    float new_w, new_h;
    if (((float)a.tensor_w / (float)a.image_w) < ((float)a.tensor_h / (float)a.image_h)) { new_w = (float)a.tensor_w; new_h = roundf((float)a.image_h * a.tensor_w / (float)a.image_w); }
    else { new_h = (float)a.tensor_h; new_w = roundf((float)a.image_w * a.tensor_h / (float)a.image_h); }
    const float x_offset = (float)((a.tensor_w - new_w) / 2.0 / a.tensor_w), x_scale = (float)new_w / a.tensor_w;
    const float y_offset = (float)((a.tensor_h - new_h) / 2.0 / a.tensor_h), y_scale = (float)new_h / a.tensor_h;
    for (int q = tid; q < n; q += blockDim.x) {
        const YoloCand c = a.cand[q];
        int rank = 0;
        for (int k = 0; k < n; k++) rank += a.cand[k].key < c.key;
        const float x = (c.x - x_offset) / x_scale * (float)a.image_w, y = (c.y - y_offset) / y_scale * (float)a.image_h;
        const float w = (c.u) / x_scale * (float)a.image_w, h = (c.w) / y_scale * (float)a.image_h;
        YoloDet d;
        d.xmin = (int)(x - w / 2); d.xmax = (int)(x + w / 2); d.ymin = (int)(y - h / 2); d.ymax = (int)(y + h / 2);
        d.objectness = c.s; d.classes = c.c;
        a.det[rank] = d;
    }
    for (int i = tid; i < YOLO_CAND; i += blockDim.x) a.sup[i] = 0;       // is_suppressed: ONE vector for all classes (see below)
    if (tid < YOLO_MAXC) s_cnt[tid] = 0;
    __threadfence_block();
    __syncthreads();
    // per class: its boxes in scan order (cls[off_c ..]), the exchange sort, the suppression loop -- one thread per class, as written
    for (int q = tid; q < n; q += blockDim.x) atomicAdd(&s_cnt[a.det[q].classes], 1);
    __syncthreads();
    if (tid == 0) { int o = 0; for (int c = 0; c < a.nc; c++) { s_off[c] = o; o += s_cnt[c]; } s_off[a.nc] = o; }
    __syncthreads();
    // The reference's `is_suppressed` is shared by all classes and indexed with class-LOCAL indices: class c reads and sets entries
    // 0 .. m_c - 1, which still carry what classes 0 .. c - 1 set there.  That makes the classes sequentially dependent through
    // sup[]; they are processed in order by thread 0 .. but only the flags cross classes: sorting is independent.  Sort in parallel
    // (thread = class), then suppress class after class.
    if (tid < a.nc) {
        const int c = tid, off = s_off[c]; int m = 0;
        for (int j = 0; j < n; j++) if (a.det[j].classes == c) a.cls[off + m++] = a.det[j];
        int* idx = a.idx + off;
        for (int i = 0; i < m; i++) idx[i] = i;
        for (int i = 0; i < m; i++)                                       // sort :256-277
            for (int j = i + 1; j < m; j++)
                if (a.cls[off + idx[j]].objectness > a.cls[off + idx[i]].objectness) { const int t = idx[i]; idx[i] = idx[j]; idx[j] = t; }
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) {
        int nout = 0;
        for (int c = 0; c < a.nc; c++) {
            const int off = s_off[c], m = s_cnt[c];
            const int* idx = a.idx + off; const YoloDet* cls = a.cls + off;
            for (int i = 0; i < m; i++) {                                  // do_nms :305-333
                if (a.sup[idx[i]]) continue;
                for (int j = i + 1; j < m; j++) {
                    const YoloDet A = cls[idx[j]], B = cls[idx[i]];
                    const float maxX = (float)min(A.xmax, B.xmax), maxY = (float)min(A.ymax, B.ymax);
                    const float minX = (float)max(A.xmin, B.xmin), minY = (float)max(A.ymin, B.ymin);
                    const float oW = maxX - minX + 1, oH = maxY - minY + 1;
                    if ((oW > 0) & (oH > 0)) {
                        const float a1 = (float)((A.xmax - A.xmin + 1) * (A.ymax - A.ymin + 1)), a2 = (float)((B.xmax - B.xmin + 1) * (B.ymax - B.ymin + 1));
                        const float iou = (oW * oH) / (a1 + a2 - oW * oH);
                        if (iou > a.nms_thresh) a.sup[idx[j]] = 1;
                    }
                }
            }
            for (int i = 0; i < m; i++) {                                  // :335-349 + tensorRunB :519-547
                if (a.sup[idx[i]]) continue;
                YoloDet d = cls[idx[i]];
                d.ymin = max(d.ymin, 0); d.xmin = max(d.xmin, 0); d.ymax = min(d.ymax, a.image_h - 1); d.xmax = min(d.xmax, a.image_w - 1);
                if (d.ymin > d.ymax || d.xmin > d.xmax || d.ymin < 0 || d.xmin < 0 || d.xmax >= a.image_w || d.ymax >= a.image_h) continue;
                if (nout < a.cap) { bbox_t o; o.t = d.ymin; o.l = d.xmin; o.b = d.ymax; o.r = d.xmax; o.type = d.classes; o.score = d.objectness; a.out[nout++] = o; }
            }
        }
        *a.nout = nout;
        if (*a.ncand > YOLO_CAND) a.ncand[1] += *a.ncand - YOLO_CAND;      // sticky: candidates dropped for lack of room (which ones is arrival order: the output of this call is not the reference's)
        *a.ncand = 0;                                                      // re-armed for the next call
    }
}

} // namespace

namespace mot_impl {
struct YoloWs {
    DevBuf<YoloCand> cand; DevBuf<YoloDet> det, cls; DevBuf<int> ints;
};
}

extern "C" int mot_yolo_postprocess(mot_ctx* c, const float* head0_dev, const float* head1_dev, const float* head2_dev, int tensor_h, int tensor_w,
                                    int num_classes, int image_h, int image_w, const yolo3_options_t* opt, bbox_t* dets_dev_out, int cap, int* n_dev_out,
                                    bbox_chain_t* host_chain_out)
{
    using namespace mot_impl;
    if (!c || !head0_dev || !head1_dev || !head2_dev || !opt || !dets_dev_out || !n_dev_out) return fail(MOT_ERR_ARG, "null argument");
    if (num_classes < 1 || num_classes > YOLO_MAXC || tensor_h < 32 || tensor_w < 32 || (tensor_h % 32) || (tensor_w % 32) || image_h < 1 || image_w < 1 || cap < 1)
        return fail(MOT_ERR_ARG, "bad geometry (classes 1..%d, tensor size a multiple of 32)", YOLO_MAXC);
    int rc = ensure_device(c); if (rc) return rc;
    if (!c->yolo) {
        c->yolo = new YoloWs;
        HIPCHK(c->yolo->cand.alloc(YOLO_CAND)); HIPCHK(c->yolo->det.alloc(YOLO_CAND)); HIPCHK(c->yolo->cls.alloc(YOLO_CAND));
        HIPCHK(c->yolo->ints.alloc(2 * YOLO_CAND + 16)); HIPCHK(hipMemsetAsync(c->yolo->ints.p, 0, sizeof(int) * (2 * YOLO_CAND + 16), c->stream));
    }
    YoloWs& W = *c->yolo;
    YoloArgs a{};
    a.head[0] = head0_dev; a.head[1] = head1_dev; a.head[2] = head2_dev;
    const int gh = tensor_h / 32, gw = tensor_w / 32;                      // yolo3.cpp:405-406
    for (int s = 0; s < 3; s++) { a.gh[s] = gh << s; a.gw[s] = gw << s; }
    for (int i = 0; i < 18; i++) a.anc[i] = opt->anchors[i];
    a.tensor_h = tensor_h; a.tensor_w = tensor_w; a.nc = num_classes; a.image_h = image_h; a.image_w = image_w;
    a.obj_thresh = opt->obj_thresh; a.nms_thresh = opt->nms_thresh;
    a.cand = W.cand.p; a.det = W.det.p; a.cls = W.cls.p; a.ncand = W.ints.p; a.idx = W.ints.p + 16; a.sup = W.ints.p + 16 + YOLO_CAND;
    a.out = dets_dev_out; a.nout = n_dev_out; a.cap = cap;
    const int threads = 3 * (a.gh[0] * a.gw[0] + a.gh[1] * a.gw[1] + a.gh[2] * a.gw[2]);
    hipLaunchKernelGGL(yolo_decode_kernel, dim3((threads + 255) / 256), dim3(256), 0, c->stream, a);
    hipLaunchKernelGGL(yolo_nms_kernel, dim3(1), dim3(1024), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    if (host_chain_out) {                                                  // the reference's hand-over format (cnntype.h:43-47): at most 128 boxes
        int n = 0, dropped = 0;
        HIPCHK(hipMemcpyAsync(&n, n_dev_out, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(&dropped, W.ints.p + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (dropped) return fail(MOT_ERR_CAPACITY, "detector post-processing: %d candidates above the objectness threshold did not fit (%d at most); raise obj_thresh", dropped, YOLO_CAND);
        if (n > MOT_CHAIN_MAX_BOXES) n = MOT_CHAIN_MAX_BOXES;
        host_chain_out->nbox = n;
        if (n) HIPCHK(hipMemcpy(host_chain_out->bbox, dets_dev_out, sizeof(bbox_t) * n, hipMemcpyDeviceToHost));
    }
    return MOT_OK;
}

// Candidates beyond the workspace (4096 above obj_thresh) are dropped in arrival order -- unlike the reference's unbounded vector
// (yolo3.cpp:176-201), so such a call's output is NOT the reference's.  The count is latched on the device; this query synchronises and
// returns MOT_ERR_CAPACITY if any call since the context was created dropped candidates (mot_yolo_postprocess with host_chain_out
// reports it by itself).  The reference's other corner, an EMPTY candidate list (correct_yolo_boxes then pushes one uninitialised box,
// yolo3.cpp:203-254), is not restated: no candidates here means no detections.
extern "C" int mot_yolo_status(mot_ctx* c, int* dropped_candidates)
{
    using namespace mot_impl;
    if (!c) return fail(MOT_ERR_ARG, "null ctx");
    int dropped = 0;
    if (c->yolo) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipMemcpy(&dropped, c->yolo->ints.p + 1, sizeof(int), hipMemcpyDeviceToHost)); }
    if (dropped_candidates) *dropped_candidates = dropped;
    return dropped ? fail(MOT_ERR_CAPACITY, "detector post-processing dropped %d candidates (%d fit)", dropped, YOLO_CAND) : MOT_OK;
}

namespace mot_impl { void yolo_destroy(YoloWs* w) { delete w; } }
