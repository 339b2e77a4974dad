// ref_hungarian_probe.cpp -- test infrastructure.  Compiles the reference's
// trackers/hungarian/hungarian.cpp (included by path, never copied).
#include REF_HUNGARIAN_CPP
extern "C" void refhung_assign(int* a, double* cost, double* dist, int nr, int nc) { assignmentoptimal(a, cost, dist, nr, nc); }
