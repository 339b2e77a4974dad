// ref_kalman_probe.cpp -- test infrastructure.  Compiles the reference's
// trackers/kalman.cpp (included by path, never copied) and exposes x / P.
#include REF_KALMAN_CPP
#include <string.h>
extern "C" {
void* refkal_new(bbox_t* b) { return tracker_new(b); }
void refkal_predict(void* p, bbox_t* b) { tracker_predict(p, 0, b); }
void refkal_update(void* p, bbox_t* b) { tracker_update(p, 0, b); }
void refkal_delete(void* p) { tracker_delete(p); }
void refkal_state(void* p, double* x6, double* P36) {
  kalman_tracker_t* k = (kalman_tracker_t*)p;
  arma::mat x = k->pkalman->get_state_vec(); arma::mat P = k->pkalman->get_err_cov();
  memcpy(x6, x.memptr(), 6 * sizeof(double)); memcpy(P36, P.memptr(), 36 * sizeof(double));
}
}
