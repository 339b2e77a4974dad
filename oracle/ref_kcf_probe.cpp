// ref_kcf_probe.cpp -- test infrastructure.  Compiles the reference's
// trackers/kcf.cpp (included by path, never copied) and exposes kcf_t internals.
#include REF_KCF_CPP
extern "C" {
void* refkcf_new(bbox_t* b) { return tracker_new(b); }
void refkcf_predict(void* p, float* g, bbox_t* b) { tracker_predict(p, g, b); }
void refkcf_update(void* p, float* g, bbox_t* b) { tracker_update(p, g, b); }
void refkcf_delete(void* p) { tracker_delete(p); }
const float* refkcf_response(void* p) { return ((kcf_t*)p)->response; }
const float* refkcf_alpha(void* p) { return ((kcf_t*)p)->alpha; }
const float* refkcf_xm(void* p) { return (const float*)((kcf_t*)p)->xf_md; }
const float* refkcf_xf(void* p) { return (const float*)((kcf_t*)p)->xf_fq; }
const float* refkcf_yf(void* p) { return (const float*)((kcf_t*)p)->yf; }
const float* refkcf_features(void* p) { return ((kcf_t*)p)->xf_tm; }
const float* refkcf_labels(void* p) { return ((kcf_t*)p)->labels.memptr(); }
const float* refkcf_coswin(void* p) { return ((kcf_t*)p)->cos_win.memptr(); }
int refkcf_frows(void* p) { return ((kcf_t*)p)->f_rows; }
int refkcf_fcols(void* p) { return ((kcf_t*)p)->f_cols; }
}
