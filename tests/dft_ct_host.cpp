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
// the in-place variant (kcf_kernels.hip, dft2_generic_inplace under MOT_FFT_MIXED): step A, then every item's sums into "registers", a barrier,
// and only then the outputs overwrite the inputs
#include <vector>
extern "C" void ct_cols_inplace(float* S, const float* tw_, int n, int N1, int fh, int nch)
{
    float2* T = reinterpret_cast<float2*>(S); const float2* tw = reinterpret_cast<const float2*>(tw_);
    for (int tid = 0; tid < 64; tid++) dftct_cols_a(T, tw, n, N1, fh, nch, tid, 64);
    const int kb = (fh + 3) >> 2, per = n * kb, total = nch * per, plane = n * fh;
    std::vector<float> acc((size_t)total * 8);
    const float inv_n1 = 1.0f / (float)N1;
    for (int i = 0; i < total; i++) {
        const int ch = i / per, rem = i - ch * per, xp = rem / kb, k0 = 4 * (rem - xp * kb);
        dftct_cols_c_item(T + (size_t)ch * plane, tw, n, N1, n / N1, fh, xp, k0, inv_n1, &acc[(size_t)i * 8]);
    }
    for (int i = 0; i < total; i++) {
        const int ch = i / per, rem = i - ch * per, xp = rem / kb, k0 = 4 * (rem - xp * kb);
        float2* o = T + (size_t)ch * plane + (size_t)xp * fh;
        for (int q = 0; q < 4; q++) if (k0 + q < fh) { o[k0 + q].x = acc[(size_t)i * 8 + 2 * q]; o[k0 + q].y = acc[(size_t)i * 8 + 2 * q + 1]; }
    }
}
// rows: F = lines of ldf floats (n real values each, overwritten by step A), T = lines of fh complex bins
extern "C" int ct_rows_factor(int n) { return dftct_rows_factor(n); }
extern "C" void ct_rows(float* F, float* T, const float* tw, int n, int N1, int fh, int ldf, int lines, int nt)
{
    for (int tid = 0; tid < nt; tid++) dftct_rows_a(F, reinterpret_cast<const float2*>(tw), n, N1, ldf, lines, tid, nt);
    for (int tid = 0; tid < nt; tid++) dftct_rows_c(F, reinterpret_cast<float2*>(T), reinterpret_cast<const float2*>(tw), n, N1, fh, ldf, lines, tid, nt);
}
// the whole forward transform of `nch` planes as fft_forward's general branch strings the passes together under MOT_FFT_MIXED (+ _ROWS):
// B[(ch*wb + x)*ldf + y] real -> rows -> T[(ch*wb + x)*fh + k] -> columns -> out[(ch*wb + x')*fh + k]
extern "C" void ct_forward2d(float* B, float* T, float* out, const float* twr, const float* twc, int hb, int wb, int nch, int r1, int c1, int nt)
{
    const int fh = hb / 2 + 1, ldf = 2 * fh, lines = nch * wb;
    for (int tid = 0; tid < nt; tid++) dftct_rows_a(B, reinterpret_cast<const float2*>(twr), hb, r1, ldf, lines, tid, nt);
    for (int tid = 0; tid < nt; tid++) dftct_rows_c(B, reinterpret_cast<float2*>(T), reinterpret_cast<const float2*>(twr), hb, r1, fh, ldf, lines, tid, nt);
    for (int tid = 0; tid < nt; tid++) dftct_cols_a(reinterpret_cast<float2*>(T), reinterpret_cast<const float2*>(twc), wb, c1, fh, nch, tid, nt);
    for (int tid = 0; tid < nt; tid++) dftct_cols_c(reinterpret_cast<const float2*>(T), reinterpret_cast<float2*>(out), reinterpret_cast<const float2*>(twc), wb, c1, fh, nch, tid, nt);
}
