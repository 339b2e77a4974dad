// dft_ct.h -- the column pass of the general (LDS-resident, not 20 x 20 cells) forward transform as TWO short passes instead of one direct DFT
// (kcf.cpp:178-195 plans an FFT of any size with FFTW; the direct pass costs wb multiply-adds per output, this one N1 + N2 for wb = N1 * N2).
//
//   x[n], n = N2*n1 + n2        X[k], k = k1 + N1*k2        W = exp(-2 pi i / wb)
//   step A (in place):   Y[k1][n2] = W^(n2*k1) * sum_n1 x[N2*n1 + n2] * W^(N2*n1*k1)      stored at line N2*k1 + n2
//   step C (to region B): X[k1 + N1*k2] = sum_n2 Y[k1][n2] * W^(N1*n2*k2)
//
// Step A's work item is (plane, n2, two adjacent bins): it owns the N1 lines {N2*n1 + n2} of its bins -- reads them all into registers, then
// overwrites the same set {N2*k1 + n2} -- so it needs no second buffer and no barrier of its own; N1 is the SMALL factor (2..5), so the item
// holds at most 5 x 2 complex inputs.  Step C is the old direct pass with N2 terms instead of wb and a block of consecutive input lines.
// Both take every twiddle from the line's own table (tw[j] = (cos, sin)(2 pi j / wb), the forward transform conjugates it), no new tables.
// The rows pass (real input) has its two-step form further down (dftct_rows_a / dftct_rows_c).
// Plain pointers and ints only: the same text compiles for the host, where tests/test_dft_ct.py checks it against numpy for every line
// length from 8 to 64 (DFTCT_FN / DFTCT_HOST).  Round 6: the default build (MOT_FFT_MIXED, kcf_kernels.hip); the rows pass still only with -DMOT_FFT_MIXED_ROWS=1 (`make fftvar`).
#pragma once
#ifndef DFTCT_FN
#define DFTCT_FN __device__ __forceinline__
#endif

// the split of a line of n cells: N1 in {2, 3, 4, 5} dividing n that minimises N1 + n / N1; 0 when none divides n or the line is short
DFTCT_FN int dftct_small_factor(int n)
{
    if (n < 8) return 0;
    int best = 0, cost = n;                                            // the direct pass costs n per output
    for (int f = 2; f <= 5; f++) if (n % f == 0 && f + n / f < cost) { cost = f + n / f; best = f; }
    return best;
}

// a / b for small non-negative ints with inv = 1.0f / b computed once per pass (the passes visit a few items per thread: three hardware
// integer divisions per item would cost as much as the multiply-adds the split saves); the float estimate is off by at most one
DFTCT_FN int dftct_div(int a, int b, float inv, int& rem)
{
    int q = (int)((float)a * inv);
    int r = a - q * b;
    if (r < 0) { q--; r += b; } else if (r >= b) { q++; r -= b; }
    rem = r;
    return q;
}

// step A, in place on T[(ch*n + x)*fh + k]
DFTCT_FN void dftct_cols_a(float2* T, const float2* tw, int n, int N1, int fh, int nch, int tid, int nt)
{
    const int N2 = (int)((float)n / (float)N1 + 0.5f), kb = (fh + 1) >> 1, per = N2 * kb, total = nch * per;
    const float inv_per = 1.0f / (float)per, inv_kb = 1.0f / (float)kb;
    for (int i = tid; i < total; i += nt) {
        int rem, kq;
        const int ch = dftct_div(i, per, inv_per, rem), n2 = dftct_div(rem, kb, inv_kb, kq), k0 = 2 * kq, kc = (k0 + 1 < fh) ? k0 + 1 : k0;
        float2* base = T + (size_t)ch * n * fh;
        float2 a[5], b[5];
#pragma unroll
        for (int n1 = 0; n1 < 5; n1++) if (n1 < N1) { const float2* sx = base + (N2 * n1 + n2) * fh; a[n1] = sx[k0]; b[n1] = sx[kc]; }
#pragma unroll
        for (int k1 = 0; k1 < 5; k1++) if (k1 < N1) {
            float ar = 0.f, ai = 0.f, br = 0.f, bi = 0.f; int j = 0;  // j = ((n1 * k1) mod N1) * N2
#pragma unroll
            for (int n1 = 0; n1 < 5; n1++) if (n1 < N1) {
                const float2 w = tw[j]; const float wi = -w.y;          // forward
                ar += a[n1].x * w.x - a[n1].y * wi; ai += a[n1].x * wi + a[n1].y * w.x;
                br += b[n1].x * w.x - b[n1].y * wi; bi += b[n1].x * wi + b[n1].y * w.x;
                j += k1 * N2; if (j >= n) j -= n;
            }
            const float2 w = tw[n2 * k1]; const float wi = -w.y;        // n2 * k1 < N2 * N1
            float2* o = base + (N2 * k1 + n2) * fh;
            float2 ra; ra.x = ar * w.x - ai * wi; ra.y = ar * wi + ai * w.x;
            float2 rb; rb.x = br * w.x - bi * wi; rb.y = br * wi + bi * w.x;
            o[k0] = ra;
            if (k0 + 1 < fh) o[k0 + 1] = rb;
        }
    }
}

// one work item of step C: the sums of output line x' = k1 + N1*k2 of one plane (as step A left it) for the bins k0 .. k0 + 3 (clamped to fh - 1)
// -> acc = re0, im0, re1, im1, re2, im2, re3, im3.  Also the column pass of the in-place variant (kcf_kernels.hip, dft2_generic_inplace), which
// keeps its outputs in registers across a barrier instead of writing to a second buffer.
DFTCT_FN void dftct_cols_c_item(const float2* plane, const float2* tw, int n, int N1, int N2, int fh, int xp, int k0, float inv_n1, float* acc)
{
    const int q1 = (k0 + 1 < fh) ? k0 + 1 : fh - 1, q2 = (k0 + 2 < fh) ? k0 + 2 : fh - 1, q3 = (k0 + 3 < fh) ? k0 + 3 : fh - 1;
    int k1;
    const int k2 = dftct_div(xp, N1, inv_n1, k1);
    const float2* src = plane + (size_t)(N2 * k1) * fh;
    float re0 = 0.f, im0 = 0.f, re1 = 0.f, im1 = 0.f, re2 = 0.f, im2 = 0.f, re3 = 0.f, im3 = 0.f; int j = 0;   // j = ((n2 * k2) mod N2) * N1
    const int step = k2 * N1;
#pragma unroll 2
    for (int n2 = 0; n2 < N2; n2++) {
        const float2 w = tw[j]; const float wi = -w.y;                  // forward
        const float2* sx = src + n2 * fh;
        const float2 a = sx[k0], b = sx[q1], c = sx[q2], d = sx[q3];
        re0 += a.x * w.x - a.y * wi; im0 += a.x * wi + a.y * w.x;
        re1 += b.x * w.x - b.y * wi; im1 += b.x * wi + b.y * w.x;
        re2 += c.x * w.x - c.y * wi; im2 += c.x * wi + c.y * w.x;
        re3 += d.x * w.x - d.y * wi; im3 += d.x * wi + d.y * w.x;
        j += step; if (j >= n) j -= n;
    }
    acc[0] = re0; acc[1] = im0; acc[2] = re1; acc[3] = im1; acc[4] = re2; acc[5] = im2; acc[6] = re3; acc[7] = im3;
}

// step C: T (as step A left it) -> out[(ch*n + x')*fh + k], x' = k1 + N1*k2
DFTCT_FN void dftct_cols_c(const float2* T, float2* out, const float2* tw, int n, int N1, int fh, int nch, int tid, int nt)
{
    const int N2 = (int)((float)n / (float)N1 + 0.5f), kb = (fh + 3) >> 2, per = n * kb, total = nch * per, plane = n * fh;
    const float inv_per = 1.0f / (float)per, inv_kb = 1.0f / (float)kb, inv_n1 = 1.0f / (float)N1;
    for (int i = tid; i < total; i += nt) {
        int rem, kq;
        const int ch = dftct_div(i, per, inv_per, rem), xp = dftct_div(rem, kb, inv_kb, kq), k0 = 4 * kq;
        float acc[8];
        dftct_cols_c_item(T + (size_t)ch * plane, tw, n, N1, N2, fh, xp, k0, inv_n1, acc);
        float2* o = out + (size_t)ch * plane + (size_t)xp * fh;
        float2 r0; r0.x = acc[0]; r0.y = acc[1]; o[k0] = r0;
        if (k0 + 1 < fh) { float2 r; r.x = acc[2]; r.y = acc[3]; o[k0 + 1] = r; }
        if (k0 + 2 < fh) { float2 r; r.x = acc[4]; r.y = acc[5]; o[k0 + 2] = r; }
        if (k0 + 3 < fh) { float2 r; r.x = acc[6]; r.y = acc[7]; o[k0 + 3] = r; }
    }
}

// ---- the ROWS pass (real input, half spectrum out) as two short passes ----
//   x[y], y = N2*n1 + n2 (real)        X[k], k < fh = n/2 + 1
//   step A (in place on the real line): for every n2 the N1-point REAL DFT Y[k1][n2] = sum_n1 x[N2*n1 + n2] W_N1^(n1*k1); N1 real inputs have
//            N1 real degrees of freedom in their spectrum (Y[N1-k1] = conj Y[k1]), so the result is stored half-complex-packed in the item's
//            own N1 slots {N2*s + n2}: slot 0 = Re Y0, slots 2k1-1 / 2k1 = Re / Im Y[k1] for 0 < 2*k1 < N1, last slot = Re Y[N1/2] for even N1
//   step C (to region T): X[k] = sum_n2 Y[k mod N1][n2] * W^(n2*k)   -- the twiddle of step B folded in: the exponent is the direct pass's own
//            (n2 * k mod n), only over N2 terms instead of n.  Complex multiply-adds (4 flops) instead of real ones (2): the pass costs 2 / N1 of
//            the direct one -- worth it for N1 >= 3.
DFTCT_FN int dftct_rows_factor(int n)
{
    if (n < 12) return 0;
    int best = 0; float cost = 1.0f;                                   // relative to the direct pass
    for (int f = 3; f <= 5; f++) if (n % f == 0 && 2.0f / (float)f < cost) { cost = 2.0f / (float)f; best = f; }
    return best;
}

DFTCT_FN void dftct_rows_a(float* F, const float2* tw, int n, int N1, int ldf, int lines, int tid, int nt)
{
    const int N2 = (int)((float)n / (float)N1 + 0.5f), total = lines * N2;
    const float inv_n2 = 1.0f / (float)N2;
    for (int i = tid; i < total; i += nt) {
        int n2;
        const int line = dftct_div(i, N2, inv_n2, n2);
        float* base = F + (size_t)line * ldf + n2;
        float x[5];
#pragma unroll
        for (int n1 = 0; n1 < 5; n1++) if (n1 < N1) x[n1] = base[N2 * n1];
        float s0 = 0.f, alt = 0.f;
#pragma unroll
        for (int n1 = 0; n1 < 5; n1++) if (n1 < N1) { s0 += x[n1]; alt += (n1 & 1) ? -x[n1] : x[n1]; }
        float re[2] = {0.f, 0.f}, im[2] = {0.f, 0.f};
#pragma unroll
        for (int k1 = 1; k1 <= 2; k1++) if (2 * k1 < N1) {
            int j = 0;                                                  // ((n1 * k1) mod N1) * N2
#pragma unroll
            for (int n1 = 0; n1 < 5; n1++) if (n1 < N1) {
                const float2 w = tw[j];
                re[k1 - 1] += x[n1] * w.x; im[k1 - 1] -= x[n1] * w.y;   // forward
                j += k1 * N2; if (j >= n) j -= n;
            }
        }
        base[0] = s0;
#pragma unroll
        for (int k1 = 1; k1 <= 2; k1++) if (2 * k1 < N1) { base[N2 * (2 * k1 - 1)] = re[k1 - 1]; base[N2 * (2 * k1)] = im[k1 - 1]; }
        if ((N1 & 1) == 0) base[N2 * (N1 - 1)] = alt;
    }
}

// step C: item = (group of four lines, bin k), as the direct rows pass; F as step A left it -> T[line*fh + k]
DFTCT_FN void dftct_rows_c(const float* F, float2* T, const float2* tw, int n, int N1, int fh, int ldf, int lines, int tid, int nt)
{
    const int N2 = (int)((float)n / (float)N1 + 0.5f), total = ((lines + 3) >> 2) * fh;
    const float inv_fh = 1.0f / (float)fh, inv_n1 = 1.0f / (float)N1;
    for (int i = tid; i < total; i += nt) {
        int k, k1;
        const int g = dftct_div(i, fh, inv_fh, k);
        (void)dftct_div(k, N1, inv_n1, k1);
        // where Y[k1] sits in the packed slots: real part in slot sa, imaginary part (times sgn) in slot sb, none for the two real lines
        int sa, sb = 0; float sgn = 0.f;
        if (k1 == 0) sa = 0;
        else if (2 * k1 == N1) sa = N1 - 1;
        else if (2 * k1 < N1) { sa = 2 * k1 - 1; sb = 2 * k1; sgn = 1.f; }
        else { const int kk = N1 - k1; sa = 2 * kk - 1; sb = 2 * kk; sgn = -1.f; }
        const int l0 = 4 * g;
        const int l1 = (l0 + 1 < lines) ? l0 + 1 : lines - 1, l2 = (l0 + 2 < lines) ? l0 + 2 : lines - 1, l3 = (l0 + 3 < lines) ? l0 + 3 : lines - 1;
        const float* a0 = F + (size_t)l0 * ldf + N2 * sa; const float* b0 = F + (size_t)l0 * ldf + N2 * sb;
        const float* a1 = F + (size_t)l1 * ldf + N2 * sa; const float* b1 = F + (size_t)l1 * ldf + N2 * sb;
        const float* a2 = F + (size_t)l2 * ldf + N2 * sa; const float* b2 = F + (size_t)l2 * ldf + N2 * sb;
        const float* a3 = F + (size_t)l3 * ldf + N2 * sa; const float* b3 = F + (size_t)l3 * ldf + N2 * sb;
        float re0 = 0.f, im0 = 0.f, re1 = 0.f, im1 = 0.f, re2 = 0.f, im2 = 0.f, re3 = 0.f, im3 = 0.f; int j = 0;
#pragma unroll 2
        for (int n2 = 0; n2 < N2; n2++) {
            const float2 w = tw[j];
            const float r0 = a0[n2], i0 = sgn * b0[n2], r1 = a1[n2], i1 = sgn * b1[n2], r2 = a2[n2], i2 = sgn * b2[n2], r3 = a3[n2], i3 = sgn * b3[n2];
            re0 += r0 * w.x + i0 * w.y; im0 += i0 * w.x - r0 * w.y;     // (r + i i)(cos - i sin)
            re1 += r1 * w.x + i1 * w.y; im1 += i1 * w.x - r1 * w.y;
            re2 += r2 * w.x + i2 * w.y; im2 += i2 * w.x - r2 * w.y;
            re3 += r3 * w.x + i3 * w.y; im3 += i3 * w.x - r3 * w.y;
            j += k; if (j >= n) j -= n;
        }
        float2* o = T + (size_t)l0 * fh + k;
        float2 r; r.x = re0; r.y = im0; o[0] = r;
        if (l0 + 1 < lines) { r.x = re1; r.y = im1; o[fh] = r; }
        if (l0 + 2 < lines) { r.x = re2; r.y = im2; o[2 * fh] = r; }
        if (l0 + 3 < lines) { r.x = re3; r.y = im3; o[3 * fh] = r; }
    }
}
