// Host build of csrc/dft_ct.h for tests/test_dft_ct.py: the two short column passes run with a sequential "thread" loop (the passes are
// separated by barriers on the device; inside a pass the work items are independent, step A's in place by ownership).
#include <cstddef>
struct float2 { float x, y; };
#define DFTCT_FN static inline
#include "../multiple-object-tracking_amd/csrc/dft_ct.h"

extern "C" int ct_small_factor(int n) { return dftct_small_factor(n); }
// T: nch planes of n lines of fh complex bins (in / scratch, overwritten by step A); out: same shape; tw: n entries (cos, sin)(2 pi j / n)
extern "C" void ct_cols(float* T, float* out, const float* tw, int n, int N1, int fh, int nch, int nt)
{
    for (int tid = 0; tid < nt; tid++) dftct_cols_a(reinterpret_cast<float2*>(T), reinterpret_cast<const float2*>(tw), n, N1, fh, nch, tid, nt);
    for (int tid = 0; tid < nt; tid++) dftct_cols_c(reinterpret_cast<const float2*>(T), reinterpret_cast<float2*>(out), reinterpret_cast<const float2*>(tw), n, N1, fh, nch, tid, nt);
}
// the interleaving a GPU could produce for step A: every "thread" in REVERSE order, and items visited in two interleaved sweeps
extern "C" void ct_cols_a_reversed(float* T, const float* tw, int n, int N1, int fh, int nch, int nt)
{
    for (int tid = nt - 1; tid >= 0; tid--) dftct_cols_a(reinterpret_cast<float2*>(T), reinterpret_cast<const float2*>(tw), n, N1, fh, nch, tid, nt);
}
