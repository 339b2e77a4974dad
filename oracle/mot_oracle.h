/* mot_oracle.h -- CPU restatement of the reference tracker hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (the HIP library, the
 * drop-in shims, the Python host mirror) may include, link or call this file.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * as the checker.
 *
 * Parity status: PINNED.  Every function here is checked by tests/ against
 * golden vectors generated in the build container from the reference's own
 * sources compiled unmodified (oracle/Makefile -> oracle/_ref/, fixtures in
 * tests/golden/, generator tests/golden/make_golden.py).  The reference ships
 * no tests or known-answer vectors of its own.
 *
 * All file:line citations are relative to the reference checkout
 * (huangfcn/multiple-object-tracking).
 */
#ifndef MOT_ORACLE_H
#define MOT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* top/cnntype.h:36-41 -- field order l,t,b,r is the reference's. */
typedef struct orc_bbox_s { int l, t, b, r; int type; float score; } orc_bbox_t;

enum { ORC_FHOG_INTEL_APPROX = 0, ORC_FHOG_EXACT = 1 };

/* x86 rcpps / rsqrtps, bit-exact integer model (tools/gen_sse_tables.c). */
float orc_sse_rcp(float x);
float orc_sse_rsqrt(float x);

/* libhog/gradientMex.cpp:47-56 -- pointer to the centre of the 20020-entry LUT. */
const float* orc_acos_table(void);

/* libhog/gradientMex.cpp:59-100 gradMag(I,M,O,h,w,d=1,full=true). Column-major h x w. */
void orc_grad_mag(const float* I, float* M, float* O, int h, int w, int mode);
/* libhog/gradientMex.cpp:112-145 (non-interpolating branch): orientation bin 0..17 per pixel. */
void orc_orient_bins(const float* O, int n, int* bins);
/* libhog/gradientMex.cpp:148-231 gradHist(M,O,R1,h,w,4,18,-1,true): R1[18][wb][hb]. */
void orc_grad_hist(const float* M, const float* O, float* R1, int h, int w);
/* libhog/fhog.h:16-38 FHoG::extract: H[32][wb][hb] (channel 31 stays zero). */
void orc_fhog(const float* I, int h, int w, float* H, int mode);

/* top/drawlib.c:192-240 */
void orc_rgb2gray(float* dst, const uint8_t* frame_bgr, int left, int top, int right, int bottom);
/* top/drawlib.c:542-637 */
void orc_resize_gray(float* dst, const float* src, int hs, int ws, int h, int w);
/* top/td.cpp:346-364: crop at box, resize to (rows, cols) into dst (rows*cols floats).
 * scratch must hold (b-t+1)*(r-l+1) floats. */
void orc_crop_patch(float* dst, float* scratch, const uint8_t* frame_bgr, const orc_bbox_t* box, int rows, int cols);

/* ---- KCF (trackers/kcf.cpp) ---- */
typedef struct orc_kcf orc_kcf;
orc_kcf* orc_kcf_new(const orc_bbox_t* box, int fhog_mode);            /* kcf.cpp:484,146 */
void     orc_kcf_delete(orc_kcf*);
void     orc_kcf_predict(orc_kcf*, const float* patch, orc_bbox_t* out); /* kcf.cpp:455,430 */
void     orc_kcf_update(orc_kcf*, const float* patch, const orc_bbox_t* box); /* kcf.cpp:462,441 */
int      orc_kcf_rows(const orc_kcf*);
int      orc_kcf_cols(const orc_kcf*);
int      orc_kcf_frows(const orc_kcf*);
int      orc_kcf_fcols(const orc_kcf*);
const float* orc_kcf_response(const orc_kcf*);   /* f_rows*f_cols, column-major */
const float* orc_kcf_alpha(const orc_kcf*);      /* f_cols*(f_rows/2+1) */
const float* orc_kcf_xm(const orc_kcf*);         /* 31*f_cols*(f_rows/2+1) complex (re,im) */
const float* orc_kcf_xf(const orc_kcf*);         /* same shape, last spectrum */
const float* orc_kcf_yf(const orc_kcf*);         /* f_cols*(f_rows/2+1) complex */
const float* orc_kcf_labels(const orc_kcf*);     /* f_rows*f_cols */
const float* orc_kcf_coswin(const orc_kcf*);     /* f_rows*f_cols */
const float* orc_kcf_features(const orc_kcf*);   /* 31*f_rows*f_cols, windowed */
void     orc_kcf_get_pos(const orc_kcf*, orc_bbox_t* out);

/* ---- Kalman (trackers/kalman.cpp, include/sigpack/kalman/kalman.h:207-237) ---- */
typedef struct orc_kalman orc_kalman;
orc_kalman* orc_kalman_new(const orc_bbox_t* box);
void     orc_kalman_delete(orc_kalman*);
void     orc_kalman_predict(orc_kalman*, orc_bbox_t* out);
void     orc_kalman_update(orc_kalman*, const orc_bbox_t* box);
void     orc_kalman_get_state(const orc_kalman*, double* x6, double* P36 /* column-major */);

/* ---- association (top/td.cpp:386-457) + Munkres (trackers/hungarian/hungarian.cpp) ---- */
/* dist layout per td.cpp: nT<nD -> rows=trackers, dist[i+nT*j]; else rows=detections, dist[j+nD*i]. */
void orc_cost_matrix(const orc_bbox_t* trk, int nT, const orc_bbox_t* det, int nD, double* dist);
void orc_assignment_optimal(int* assignment, double* cost, const double* dist, int nRows, int nCols);

/* ---- per-frame tracker loop (top/td.cpp:306-748, tracker thread body) ---- */
enum { ORC_TRACKER_KCF = 0, ORC_TRACKER_KALMAN = 1 };
typedef struct orc_mot orc_mot;
orc_mot* orc_mot_new(int kind, int fhog_mode, int max_tracks);
void     orc_mot_delete(orc_mot*);
/* One frame.  Returns the number of live tracks after the frame; fills (optionally)
 * predicted[] (clamped boxes after predict, old track order), assigned_trackers[]
 * (old track order), and boxes/tids of the live tracks after lifecycle. */
int orc_mot_step(orc_mot*, const uint8_t* frame_bgr, const orc_bbox_t* dets, int ndet,
                 orc_bbox_t* predicted, int* assigned_trackers, int* n_before,
                 orc_bbox_t* live_boxes, unsigned* live_tids);
int orc_mot_ntracks(const orc_mot*);
/* access to a live track's tracker object (for response / state checks) */
orc_kcf*    orc_mot_kcf(orc_mot*, int i);
orc_kalman* orc_mot_kalman(orc_mot*, int i);

/* ---- overlay: top/drawlib.c:97-151 drawRect, top/td.cpp:295-304 hashcolor, td.cpp:647-733 three nested
 * outlines per live track in colormap[hashcolor(tid) & 255] (td.cpp:620,655-699).  drawRect writes the bytes R, G, B of its
 * "RGB" argument to memory offsets 0, 1, 2 of a pixel of the BGR frame -- reproduced as is. ---- */
void orc_draw_rect(uint8_t* fbuf, int left, int top, int right, int bottom, uint32_t rgb);
uint32_t orc_hashcolor(uint32_t a);
const uint32_t* orc_colormap(void);                                  /* 256 entries */
void orc_overlay(uint8_t* frame_bgr, const orc_bbox_t* boxes, const unsigned* tids, int n);

/* ---- detector post-processing: detectors/yolo3.cpp:141-356 (decode_netout, correct_yolo_boxes, sort, do_nms) and the
 * emit loop of tensorRunB (:490-530).  PARITY UNPINNED: yolo3.cpp includes <Windows.h> and TensorFlow headers and cannot be compiled
 * here, and the reference holds no test vectors for it -- this restatement follows the source by reading only.  Three heads of raw
 * network output, NHWC [grid_h << s][grid_w << s][3 * (5 + classes)], s = 0, 1, 2, anchors + 12 / + 6 / + 0 (:512-514).
 * Returns the number of boxes written (at most cap). ---- */
typedef struct { float obj_thresh; float nms_thresh; int anchors[18]; } orc_yolo_opt;   /* top/cnntype.h:49-54 */
int orc_yolo_postprocess(const float* head0, const float* head1, const float* head2, int tensor_h, int tensor_w, int num_classes,
                         int image_h, int image_w, const orc_yolo_opt* opt, orc_bbox_t* out, int cap);

#ifdef __cplusplus
}
#endif
#endif
