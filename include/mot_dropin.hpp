// mot_dropin.hpp -- the reference's per-object tracker interface, unchanged.
//
// These are the five free functions top/td.cpp declares at td.cpp:229-234 and
// links from trackers/kcf.cpp:455-491 | trackers/kalman.cpp:131-163 and
// trackers/hungarian/hungarian.cpp:29.  They have C++ linkage in the reference,
// so the replacement libraries export the same Itanium-mangled symbols:
//     _Z11tracker_newP11_bbox_pos_s          _Z14tracker_deletePv
//     _Z15tracker_predictPvPfP11_bbox_pos_s  _Z14tracker_updatePvPfP11_bbox_pos_s
//     _Z17assignmentoptimalPiPdS0_ii
// libmot_dropin_kcf.so and libmot_dropin_kalman.so each define all five (link
// exactly one, as the reference links exactly one of kcf.cpp / kalman.cpp).
// Every call is a batch-of-one over the MI355X batch ABI (mot_abi.h); the
// process-wide context is created on first use on HIP device $MOT_DEVICE (0).
#pragma once
#include "mot_abi.h"

void* tracker_new(bbox_t* pbox);                                   // kcf.cpp:484 | kalman.cpp:148
void tracker_predict(void* ptracker, float* rgb, bbox_t* pbox);    // kcf.cpp:455 | kalman.cpp:131
void tracker_update(void* ptracker, float* rgb, bbox_t* pbox);     // kcf.cpp:462 | kalman.cpp:136
void tracker_delete(void* ptracker);                               // kcf.cpp:478 | kalman.cpp:141
void assignmentoptimal(int* assignment, double* cost, double* distMatrixIn, int nOfRows, int nOfColumns); // hungarian.cpp:29
