// kcf_kernels.hip -- fused per-track KCF predict / update kernels for gfx950.
//
// One 512-thread workgroup per track.  All intermediates of the reference's
// per-track pipeline
//     rgb2Gray + bilinearInterpolationGray   (top/drawlib.c:192-240, 542-637)
//     FHoG::extract                          (libhog/fhog.h:16, gradientMex.cpp)
//     31 x r2c 2-D FFT                       (trackers/kcf.cpp:261-267, FFTW)
//     linear correlation / alpha / model     (trackers/kcf.cpp:269-395)
//     c2r + arg-max + box shift              (trackers/kcf.cpp:397-428)
// live in one scratch slab (LDS when it fits in 160 KB, else an HBM slab per
// workgroup); only the u8 crop, the model xm/alpha and the boxes touch HBM.
//
// Numerics: everything up to and including the 31 FHOG channels is computed
// with the reference's exact float operation order and separate roundings
// (-ffp-contract=off) so features are bit-identical to the as-compiled
// reference, including its rcpps/rsqrtps approximations (table emulation).
// The DFTs are free to contract (FFTW's own rounding is not reproducible).
#include "mot_dev.h"
#include "mot_env.h"
#include <hip/hip_ext.h>
#include <type_traits>
#include "bin_thresholds.inc"

#ifndef MOT_HIST_BATCH
#define MOT_HIST_BATCH 0      /* 1: the histogram's eight read-modify-writes of a footprint column resolved in registers (round 5 A/B; see phase_hist) */
#endif
#ifndef MOT_GRAD_SPEC
#define MOT_GRAD_SPEC 1       /* 1: gradient loop instantiated per FHOG flavour (round 5 A/B; see phase_gradmag) */
#endif
#define PI_F 3.14159265f /* libhog/gradientMex.cpp:12 */
// (probe build only, `make ablate` -> libmot_amd_ablate.so) phases of the KCF kernels can be switched off one by one (KcfLaunch::ablate, from
// MOT_KCF_ABLATE): the launch time lost is what the phase costs the LAUNCH under real contention -- results are garbage, timing tools only
#ifdef MOT_KCF_ABLATE
#define ABL(bit) (!((l.ablate >> (bit)) & 1))
#else
#define ABL(bit) true
#endif
/* R1 (18 orientation sums per cell) in LDS is WAVE-INTERLEAVED: bin o of cell c lives at ((c / 64) * 18 + o) * 64 + c % 64, so
 * a lane's read-modify-writes always hit bank = lane whatever the (data-dependent) bin -- cell-major rows made the
 * histogram's scatter 2- to 4-way bank-conflicted.  In the HBM slab R1 is orientation-major (coalesced), o * nb + c. */
#define R1W(c, o) ((((c) >> 6) * MOT_NORI + (o)) * 64 + ((c) & 63))
template <bool SOA> __device__ __forceinline__ int r1_index(int nb, int cell, int o) { return SOA ? o * nb + cell : R1W(cell, o); }

namespace {

__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }

// x86 rcpps / rsqrtps (libhog/sse.hpp:40-41) -- integer model, see tools/gen_sse_tables.c.
// Written select-style (no branches): the table path is computed unconditionally and the special
// cases (NaN, zero/denormal, infinity, negative, underflow) are patched in with conditional moves.
__device__ __forceinline__ float sse_rcp(float x, const uint16_t* tab)
{
    const uint32_t u = f2u(x), s = u & 0x80000000u, e = (u >> 23) & 0xff, m = u & 0x7fffff;
    const int ep = 253 - (int)e;
    uint32_t r = s | ((uint32_t)ep << 23) | ((uint32_t)tab[m >> 12] << 11);
    r = (ep <= 0) ? s : r;                                   // result underflows -> +-0
    r = (e == 0) ? (s | 0x7f800000u) : r;                    // zero / denormal -> +-inf
    r = (e == 0xff) ? (m ? (u | 0x400000u) : s) : r;         // NaN -> quiet NaN, inf -> +-0
    return u2f(r);
}
__device__ __forceinline__ float sse_rsqrt(float x, const uint16_t* tab)
{
    const uint32_t u = f2u(x), s = u & 0x80000000u, e = (u >> 23) & 0xff, m = u & 0x7fffff;
    const int E = (int)e - 127, odd = E & 1;
    const int ep = 126 - ((E - odd) >> 1);                   // odd: 126-(E-1)/2, even: 126-E/2 (E-odd is even: exact shift)
    uint32_t r = ((uint32_t)ep << 23) | ((uint32_t)tab[2048 + odd * 1024 + (m >> 13)] << 11);
    r = (e == 0xff) ? 0u : r;                                // +inf -> 0
    r = s ? 0xffc00000u : r;                                 // negative -> NaN
    r = (e == 0) ? (s | 0x7f800000u) : r;                    // +-0 / denormal -> +-inf
    r = (e == 0xff && m) ? (u | 0x400000u) : r;              // NaN -> quiet NaN
    return u2f(r);
}

// The two call sites of phase_gradmag see a restricted domain: m2 = gx*gx + gy*gy is finite and >= +0 (gray values are
// 0..255), and rcp's argument is min(rsqrt(m2), 1e10f), a positive normal number <= 1e10 -- the sign / NaN / infinity /
// underflow cases of the general emulations above cannot occur there and are dropped (same bits on that domain).
__device__ __forceinline__ float sse_rsqrt_nonneg_finite(float x, const uint16_t* tab)
{
    const uint32_t u = f2u(x), e = u >> 23, m = u & 0x7fffff;          // sign bit is 0
    const int E = (int)e - 127, odd = E & 1;
    const int ep = 126 - ((E - odd) >> 1);
    const uint32_t r = ((uint32_t)ep << 23) | ((uint32_t)tab[2048 + odd * 1024 + (m >> 13)] << 11);
    return u2f((e == 0) ? 0x7f800000u : r);                             // +0 / denormal -> +inf
}
__device__ __forceinline__ float sse_rcp_pos_normal(float x, const uint16_t* tab)
{
    const uint32_t u = f2u(x), e = u >> 23, m = u & 0x7fffff;          // 1 <= e <= 160: 253 - e > 0
    return u2f(((253u - e) << 23) | ((uint32_t)tab[m >> 12] << 11));
}

// drawlib.c:234 -- double arithmetic, one rounding to float at the end.
__device__ __forceinline__ float gray_of(const uint8_t* __restrict__ frame, int y, int x)
{
    y = min(max(y, 0), MOT_FRAME_H - 1);   // the reference reads unchecked; boxes are clamped by the caller
    x = min(max(x, 0), MOT_FRAME_W - 1);
    const uint8_t* p = frame + ((size_t)y * MOT_FRAME_W + x) * 3;
    double B = (double)p[0], G = (double)p[1], R = (double)p[2];
    return (float)(0.144 * B + 0.587 * G + 0.299 * R);
}

struct Ctx {
    int tid, nt;
};

// ---------------------------------------------------------------------------
// Phase 0: crop + gray + (bilinear resize) -> P[c*ldp + r]
// ---------------------------------------------------------------------------
__device__ void phase_crop(const KcfPool& p, const uint8_t* __restrict__ frame, const float* __restrict__ patch,
                           bbox_t box, float* __restrict__ P, uint8_t* __restrict__ raw, int tid, int nt, int raw_cap = 1 << 30,
                           float* __restrict__ gbuf = nullptr, int gcap = 0, long long* dbg = nullptr)
{
    const int rows = p.rows, cols = p.cols, npx = rows * cols;
    // the pad floats below every patch column are read by the gradient of the rounded-up last 4-pixel group; those pixels
    // carry weight 0 in the histogram but must stay finite
    for (int i = tid; i < cols * (p.ldp - rows); i += nt) { const int c = i / (p.ldp - rows), k = i - c * (p.ldp - rows); P[c * p.ldp + rows + k] = 0.0f; }
    if (patch) {
        for (int d = tid; d < npx; d += nt) {
            uint32_t c, r; p.d_rows.divmod((uint32_t)d, c, r);
            P[c * p.ldp + r] = patch[d];
        }
        return;
    }
    int left = box.l, top = box.t, right = box.r, bottom = box.b;
    if (top > bottom) { int t = top; top = bottom; bottom = t; }      // drawlib.c:203-215
    if (left > right) { int t = left; left = right; right = t; }
    const int hs = box.b - box.t + 1, ws = box.r - box.l + 1;          // td.cpp:360-361 (as passed to the resize)
    const int rows_s = bottom - top + 1;                               // rgb2Gray's column stride
    if (hs == rows && ws == cols && rows_s == rows) {
        // identity resize (frac == 0): patch flat index d = c*rows + r <-> pixel (top+r, left+c)
        if (raw && left >= 0 && top >= 0 && left + cols <= MOT_FRAME_W && top + rows <= MOT_FRAME_H) {
            // stage the BGR rows of the crop in LDS with aligned 16-byte loads (all loads of the workgroup in
            // flight at once, one HBM round trip), then convert from LDS
            const size_t g0 = (size_t)(frame + ((size_t)top * MOT_FRAME_W + left) * 3);
            const int mis = (int)(g0 & 15);                            // same for every row: 3840 = 240 * 16
            const int nch = (mis + cols * 3 + 15) >> 4;                // 16-byte chunks per row
            const int stride = nch * 16;
            if (rows * stride <= raw_cap) {
            const uint8_t* fend = frame + (size_t)MOT_FRAME_W * MOT_FRAME_H * 3;
            for (int i = tid; i < rows * nch; i += nt) {
                const int r = i / nch, j = i - r * nch;
                const uint8_t* src = frame + ((size_t)(top + r) * MOT_FRAME_W + left) * 3 - mis + 16 * j;
                uint4 v;
                if (src + 16 <= fend) v = *reinterpret_cast<const uint4*>(src);
                else { uint8_t t8[16]; for (int q = 0; q < 16; q++) t8[q] = (src + q < fend) ? src[q] : 0; v = *reinterpret_cast<uint4*>(t8); }
                *reinterpret_cast<uint4*>(raw + r * stride + 16 * j) = v;
            }
            __syncthreads();
            for (int i = tid; i < npx; i += nt) {
                uint32_t r, c; p.d_cols.divmod((uint32_t)i, r, c);
                const uint8_t* px = raw + r * stride + mis + 3 * c;
                const double B = (double)px[0], G = (double)px[1], R = (double)px[2];
                P[c * p.ldp + r] = (float)(0.144 * B + 0.587 * G + 0.299 * R);   // drawlib.c:234
            }
            return;
            }
        }
        for (int i = tid; i < npx; i += nt) {
            uint32_t r, c; p.d_cols.divmod((uint32_t)i, r, c);         // c fastest: contiguous BGR bytes
            P[c * p.ldp + r] = gray_of(frame, top + (int)r, left + (int)c);
        }
        return;
    }
    // general case, drawlib.c:542-637 called as (heightSource=hs, widthSource=ws, height=rows, width=cols);
    // the source scratch is rgb2Gray's column-major buffer read with flat index y0*ws+x0.
    const float xs = ((float)ws) / ((float)cols);
    const float ys = ((float)hs) / ((float)rows);
    FastDiv drs; // flat source index -> (col, row) of the gray scratch
    const uint32_t urs = (uint32_t)max(rows_s, 1);
    // Two steps like the reference when the gray image of the source box fits the scratch `gbuf` (LDS): every source pixel
    // is converted ONCE (coalesced BGR reads along the frame rows), the four taps of an output pixel are LDS reads.
    // Element s of the scratch is the pixel (top + s % rows_s, left + s / rows_s), s < hs * ws.
    const int nsrc = hs * ws;
    if (gbuf && hs > 0 && ws > 0 && nsrc <= gcap && rows_s == hs) {
        const uint32_t uws = (uint32_t)ws;
#pragma unroll 4
        for (int i = tid; i < nsrc; i += nt) {
            const uint32_t r = (uint32_t)i / uws, c = (uint32_t)i - r * uws;       // frame-row major: consecutive threads, consecutive pixels
            gbuf[c * urs + r] = gray_of(frame, top + (int)r, left + (int)c);
        }
        __syncthreads();
        if (dbg && blockIdx.x == 0 && tid == 0) dbg[16] = wall_clock64();
        for (int d = tid; d < npx; d += nt) {
            uint32_t y, x; p.d_cols.divmod((uint32_t)d, y, x);
            float sy = (float)y * ys; int y0 = (int)sy; float fracy = sy - (float)y0, ifracy = 1.0f - fracy;
            int y1 = y0 + 1; if (y1 >= hs) y1 = y0;
            float sx = (float)x * xs; int x0 = (int)sx; float fracx = sx - (float)x0, ifracx = 1.0f - fracx;
            int x1 = x0 + 1; if (x1 >= ws) x1 = x0;
            const float c1 = gbuf[y0 * ws + x0], c2 = gbuf[y0 * ws + x1], c3 = gbuf[y1 * ws + x0], c4 = gbuf[y1 * ws + x1];
            float l0 = ifracx * c1 + fracx * c2;                           // drawlib.c:625-627
            float l1 = ifracx * c3 + fracx * c4;
            float v = ifracy * l0 + fracy * l1;
            uint32_t c, r; p.d_rows.divmod((uint32_t)d, c, r);             // KCF reads the flat array column-major
            P[c * p.ldp + r] = v;
        }
        return;
    }
    for (int d = tid; d < npx; d += nt) {
        uint32_t y, x; p.d_cols.divmod((uint32_t)d, y, x);             // dst flat = y*width + x
        float sy = (float)y * ys; int y0 = (int)sy; float fracy = sy - (float)y0, ifracy = 1.0f - fracy;
        int y1 = y0 + 1; if (y1 >= hs) y1 = y0;
        float sx = (float)x * xs; int x0 = (int)sx; float fracx = sx - (float)x0, ifracx = 1.0f - fracx;
        int x1 = x0 + 1; if (x1 >= ws) x1 = x0;
        int s1 = y0 * ws + x0, s2 = y0 * ws + x1, s3 = y1 * ws + x0, s4 = y1 * ws + x1;
        float c1 = gray_of(frame, top + (int)((uint32_t)s1 % urs), left + (int)((uint32_t)s1 / urs));
        float c2 = gray_of(frame, top + (int)((uint32_t)s2 % urs), left + (int)((uint32_t)s2 / urs));
        float c3 = gray_of(frame, top + (int)((uint32_t)s3 % urs), left + (int)((uint32_t)s3 / urs));
        float c4 = gray_of(frame, top + (int)((uint32_t)s4 % urs), left + (int)((uint32_t)s4 / urs));
        float l0 = ifracx * c1 + fracx * c2;                           // drawlib.c:625-627
        float l1 = ifracx * c3 + fracx * c4;
        float v = ifracy * l0 + fracy * l1;
        uint32_t c, r; p.d_rows.divmod((uint32_t)d, c, r);             // KCF reads the flat array column-major
        P[c * p.ldp + r] = v;
    }
    (void)drs;
}

// ---------------------------------------------------------------------------
// Phase 1: gradMag (gradientMex.cpp:15-37, 59-100) + orientation quantisation
// (:119-143) -> Mq = M*(1/16) (float), bin (u8)
// ---------------------------------------------------------------------------
// Stripes (R1-resident HBM-slab templates): only the pixel columns [xb, xb + xn) are produced; Mq / bins are then the stripe's VIRTUAL bases
// (stripe buffer - xb * ldp), so the column indexing below and in the histogram is the same for a stripe and a full plane.
template <int UNR = 1>   // iterations the compiler may interleave (R1-resident templates: 8 waves per CU, the table look-ups of one group hide behind the next)
__device__ void phase_gradmag(const KcfPool& p, const float* __restrict__ P, float* __restrict__ Mq,
                              uint8_t* __restrict__ bins, const uint16_t* __restrict__ tab, int tid, int nt, int xb = 0, int xn = -1)
{
    // one thread = 4 vertically adjacent pixels (y0..y0+3) of column x: three aligned 16-byte LDS reads + two scalars.
    // Mq / bins are stored as [x][2 + y] with column stride ldp so the 8-pixel footprint of a cell starts 16-byte aligned.
    const int h = p.rows, w = p.cols, LP = p.ldp, ng = p.ng;
    const int thr0[9] = MOT_BIN_THR0;
    { const int t1[9] = MOT_BIN_THR1; for (int j = 0; j < 9; j++) if (t1[j] != thr0[j]) __builtin_trap(); }   // both sign flags share the thresholds
    if (thr0[4] != 1) __builtin_trap();
    for (int j = 0; j < 4; j++) if (thr0[5 + j] != -thr0[3 - j] + 1) __builtin_trap();                           // the mirror structure the bin count relies on
    const bool approx_rt = (p.fhog_mode == MOT_FHOG_INTEL_APPROX);
    if (xn < 0) xn = w;
    // `approx` is launch-uniform: the loop is instantiated once per flavour (round 5) -- with the run-time test inside, every pixel's two table
    // look-ups sat behind a branch and were waited for one by one (8 dependent LDS round trips per 4-pixel group); now the four pixels of a
    // group are straight-line code and their look-ups are in flight together
    auto group = [&](int it, auto approx_c) {
        constexpr bool approx = decltype(approx_c)::value;
        uint32_t x, kq; p.d_ng.divmod((uint32_t)it, x, kq);
        x += (uint32_t)xb;
        const int y0 = 4 * (int)kq;
        const float* Pc = P + x * LP;
        const float4 c4 = *reinterpret_cast<const float4*>(Pc + y0);
        const float4 l4 = *reinterpret_cast<const float4*>(P + max((int)x - 1, 0) * LP + y0);
        const float4 r4 = *reinterpret_cast<const float4*>(P + min((int)x + 1, w - 1) * LP + y0);
        const float up = Pc[max(y0 - 1, 0)], dn = Pc[min(y0 + 4, h - 1)];
        const float cc[6] = { up, c4.x, c4.y, c4.z, c4.w, dn };
        const float ll[4] = { l4.x, l4.y, l4.z, l4.w }, rr[4] = { r4.x, r4.y, r4.z, r4.w };
        const float rx = (x == 0 || (int)x == w - 1) ? 1.0f : 0.5f;
        float mq[4]; uint32_t bq[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int y = y0 + j;
            // one-sided differences at the borders = the same expression with clamped neighbours and factor 1
            const float ym = (y == 0) ? cc[j + 1] : cc[j];
            const float yp = (y >= h - 1) ? cc[j + 1] : cc[j + 2];
            const float ry = (y == 0 || y >= h - 1) ? 1.0f : 0.5f;
            const float gx = (rr[j] - ll[j]) * rx;
            const float gy = (yp - ym) * ry;
            const float m2 = gx * gx + gy * gy;
            float m = approx ? sse_rsqrt_nonneg_finite(m2, tab) : 1.0f / sqrtf(m2);
            m = (m < 1e10f) ? m : 1e10f;                               // _mm_min_ps(m, 1e10f)
            const float mag = approx ? sse_rcp_pos_normal(m, tab) : 1.0f / m;
            float g = (gx * m) * 10000.0f;
            g = u2f(f2u(g) ^ (f2u(gy) & 0x80000000u));
            const int idx = (int)g;
            // bin = TOP - #{q : idx >= thr[q]}.  The nine thresholds are {-u3, -u2, -u1, -u0, 1, u0 + 1, u1 + 1, u2 + 1, u3 + 1}
            // (checked below), so with a = |idx| and k = #{q : a > u_q} the count is 5 + k for idx >= 1 and 4 - k otherwise:
            // four compares instead of nine
            const int a = idx < 0 ? -idx : idx;
            const int k = (a > -thr0[3]) + (a > -thr0[2]) + (a > -thr0[1]) + (a > -thr0[0]);
            int b = ((gy < 0.0f) ? MOT_BIN_TOP1 : MOT_BIN_TOP0) - ((idx >= 1) ? 5 + k : 4 - k);
            if (b >= 18) b = 0;
            mq[j] = mag * 0.0625f;                                     // norm = 1/bin/bin (:152,132)
            bq[j] = (uint32_t)b;
        }
        float* mo = Mq + x * LP + 2 + y0;                              // 8-byte aligned
        *reinterpret_cast<float2*>(mo) = make_float2(mq[0], mq[1]);
        *reinterpret_cast<float2*>(mo + 2) = make_float2(mq[2], mq[3]);
        uint8_t* bo = bins + x * LP + 2 + y0;                          // 2-byte aligned
        *reinterpret_cast<uint16_t*>(bo) = (uint16_t)(bq[0] | (bq[1] << 8));
        *reinterpret_cast<uint16_t*>(bo + 2) = (uint16_t)(bq[2] | (bq[3] << 8));
    };
#if !MOT_GRAD_SPEC
    {   // (A/B) one loop, the flavour tested per pixel as before round 5
        auto group_rt = [&](int it) { if (approx_rt) group(it, std::true_type{}); else group(it, std::false_type{}); };
        for (int it = tid; it < xn * ng; it += nt) group_rt(it);
    }
#else
    if (approx_rt) {
        if (UNR > 1) {
#pragma unroll UNR
            for (int it = tid; it < xn * ng; it += nt) group(it, std::true_type{});
        } else {
            for (int it = tid; it < xn * ng; it += nt) group(it, std::true_type{});       // the compiler's own choice, as before the stripes existed
        }
    } else {
        for (int it = tid; it < xn * ng; it += nt) group(it, std::false_type{});
    }
#endif
    // the two pad slots above and below every column are read (with weight 0) by the histogram: finite magnitude, bin 0
    for (int i = tid; i < 4 * xn; i += nt) {
        const int x = xb + (i >> 2), k = i & 3, idx = x * LP + (k < 2 ? k : LP - 4 + k);
        Mq[idx] = 0.0f; bins[idx] = 0;
    }
}

// ---------------------------------------------------------------------------
// Phase 2: gradHist, trilinear branch with nearest orientation
// (gradientMex.cpp:183-221) as a GATHER: one thread owns one cell and adds the
// contributions of its <= 8x8 pixel footprint in the reference's order
// (x outer, y inner), so every R1 value is bit-identical.  Then :225-230.
// ---------------------------------------------------------------------------
// NPRE (HBM-slab templates): the footprint columns are loaded four at a time, all in flight at once, because there every
// load is an L2 round trip; the LDS-resident kernels load column by column.
// BATCH (compile-time A/B, -DMOT_HIST_BATCH=1; off): the eight read-modify-writes of a footprint column resolved in registers -- 8 instead of 64
// dependent LDS round trips per cell, at the price of 56 more instructions per column.  Round 5 measured it both ways and it loses both: a launch that
// FILLS the chip (1024 tracks, two workgroups per CU) is bound by what it issues, not by one workgroup's latency (isolated predict launch 88 -> 92 us,
// profiles/r05_kernel_ab_hist_grad.log), and a small joined launch (64 tracks, one workgroup per CU) does not gain either (36.5 against 36.1 us,
// profiles/r05_bench_n64*.json).
template <int NPRE, bool LDSACC = true, bool BATCH = (MOT_HIST_BATCH != 0)>   // LDSACC: the accumulators live in LDS (false only for the FHOG-only test kernel of templates beyond the LDS)
__device__ void phase_hist(const KcfPool& p, const float* __restrict__ Mq, const uint8_t* __restrict__ bins,
                           float* __restrict__ R1, float* __restrict__ scratch, int tid, int nt, int cell_lo = 0, int cell_hi = -1)
{
    const int hb = p.hb, wb = p.wb, nb = p.nb, LP = p.ldp;
    const int h0 = hb * 4, w0 = wb * 4;
    constexpr int GRP = NPRE ? NPRE : 1;                               // footprint columns whose loads are in flight together
    if (cell_hi < 0) cell_hi = nb;
    for (int cell = cell_lo + tid; cell < cell_hi; cell += nt) {
        uint32_t cx, cy; p.d_hb.divmod((uint32_t)cell, cx, cy);
        // HBM-slab templates accumulate in a per-thread LDS scratch (the read-modify-write chain would otherwise run at L2
        // latency) and copy the finished cell out
        float* __restrict__ Rc = NPRE ? scratch + R1W(tid, 0) : R1 + R1W(cell, 0);   // bin o at Rc[o * 64]; always LDS (HBM-slab templates: the staging area)
#pragma unroll
        for (int o = 0; o < MOT_NORI; o++) Rc[o * 64] = 0.0f;
        const int x_lo = max(0, 4 * (int)cx - 2), x_hi = min(w0 - 1, 4 * (int)cx + 5);
        const int y_lo = max(0, 4 * (int)cy - 2), y_hi = min(h0 - 1, 4 * (int)cy + 5);
        // One column of the footprint (<= 8 pixels) per round: all LDS reads of the round are issued together and
        // the eight read-modify-writes are resolved in registers in pixel order (a later pixel of the same
        // orientation takes the running value of the earlier one), then written back in order -- the same
        // sequence of float additions as the reference, without eight dependent LDS round trips.
        // The bilinear weight of footprint pixel (i, j), i/j = 0..7 counted from (4cx-2, 4cy-2), is the exact
        // product wq[i]*wq[j] with wq = {1,3,5,7,7,5,3,1}/8: the reference's ms[] expressions (:192) evaluate
        // to exactly these values (all operands are multiples of 1/8, products multiples of 1/64).
        const float wq[8] = { 0.125f, 0.375f, 0.625f, 0.875f, 0.875f, 0.625f, 0.375f, 0.125f };
        const int x0 = 4 * (int)cx - 2, y0 = 4 * (int)cy - 2;
        float wy[8];
#pragma unroll
        for (int j = 0; j < 8; j++) wy[j] = (y0 + j >= y_lo && y0 + j <= y_hi) ? wq[j] : 0.0f;
#pragma unroll
        for (int i0 = 0; i0 < 8; i0 += GRP) {
            float4 pma[GRP], pmb[GRP]; uint32_t pba[GRP], pbb[GRP];
#pragma unroll
            for (int g = 0; g < GRP; g++) {
                const int x = min(max(x0 + i0 + g, x_lo), x_hi);          // clamped: always a valid column
                const float* mp = Mq + x * LP + 4 * (int)cy;              // = [x][2 + (4cy-2)], 16-byte aligned
                pma[g] = *reinterpret_cast<const float4*>(mp); pmb[g] = *reinterpret_cast<const float4*>(mp + 4);
                pba[g] = *reinterpret_cast<const uint32_t*>(bins + x * LP + 4 * (int)cy);
                pbb[g] = *reinterpret_cast<const uint32_t*>(bins + x * LP + 4 * (int)cy + 4);
            }
#pragma unroll
            for (int g = 0; g < GRP; g++) {
                const int i = i0 + g, x = x0 + i;
                // straight-line code, no per-pixel predication: a pixel outside the footprint's valid range gets weight 0 and
                // adds +0.0f (an exact no-op on the non-negative sums) to whatever bin its clamped column / zeroed pad names
                const float wx = (x >= x_lo && x <= x_hi) ? wq[i] : 0.0f;
                const float4 ma = pma[g], mb = pmb[g];
                const uint32_t ba = pba[g], bb = pbb[g];
                const float mv[8] = { ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w };
                // The eight read-modify-writes of the column in ONE LDS round trip instead of eight dependent ones (round 5; before, every pixel
                // waited for its own ds_read: 64 round trips per cell): all eight sums are read first, the pixel order is then resolved in
                // registers -- a later pixel of the same orientation continues from the running value of the latest earlier one, exactly the
                // sequence of float additions of the sequential form -- and the eight results are stored in pixel order (the LDS executes a
                // wave's accesses in order, so the last store of an orientation is the one that stays, and the next column's reads see it).
                // (ds_add_f32 -- one LDS float add per pixel, no return value, bit-identical sums -- was measured in round 3: the predict
                // launch went from 100 to 215 us; the LDS atomic path is that slow.)
                if constexpr (!BATCH) {
#pragma unroll
                for (int j = 0; j < 8; j++) {                            // sequential form: one dependent LDS round trip per pixel, fewest instructions
                    const int bj = (int)(((j < 4 ? ba : bb) >> (8 * (j & 3))) & 0xffu);
                    Rc[bj * 64] += (wx * wy[j]) * mv[j];
                }
                } else {
                int ad[8]; float wv[8], rv[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    ad[j] = (int)(((j < 4 ? ba : bb) >> (8 * (j & 3))) & 0xffu) * 64;
                    wv[j] = (wx * wy[j]) * mv[j];
                }
#pragma unroll
                for (int j = 0; j < 8; j++) rv[j] = Rc[ad[j]];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    float base = rv[j];
#pragma unroll
                    for (int m = 0; m < j; m++) base = (ad[m] == ad[j]) ? rv[m] : base;   // rv[m] already holds pixel m's result
                    rv[j] = base + wv[j];
                }
#pragma unroll
                for (int j = 0; j < 8; j++) Rc[ad[j]] = rv[j];
                }
            }
        }
        int nmul = ((int)cx == 0) + ((int)cy == 0) + ((int)cx == wb - 1) + ((int)cy == hb - 1);
        if (nmul) {
            const float c = 8.f / 7.f;
            for (int o = 0; o < MOT_NORI; o++) {
                float v = Rc[o * 64];
                for (int k = 0; k < nmul; k++) v *= c;
                Rc[o * 64] = v;
            }
        }
        if (NPRE) {
            // in the slab R1 is orientation-major, R1[o * nb + cell]: a wave's store covers 64 neighbouring cells (cell-major
            // rows of 19 floats cost the texture addresser one cache line per lane and access)
#pragma unroll
            for (int o = 0; o < MOT_NORI; o++) R1[o * nb + cell] = Rc[o * 64];
        }
    }
}

// Phase 3a: E[cell] = sum_o (R1[o]+R1[o+9])^2  (gradientMex.cpp:308-309, 240-241)
template <bool SOA>   // SOA: R1[o * nb + cell] (HBM slab), else wave-interleaved (LDS)
__device__ void phase_energy(const KcfPool& p, const float* __restrict__ R1, float* __restrict__ E, int tid, int nt)
{
    const int nb = p.nb;
    for (int cell = tid; cell < nb; cell += nt) {
        float e = 0.0f;
#pragma unroll
        for (int o = 0; o < 9; o++) {
            float r2 = R1[r1_index<SOA>(nb, cell, o)] + R1[r1_index<SOA>(nb, cell, o + 9)];
            e += r2 * r2;
        }
        E[cell] = e;
    }
}

// Phase 3b: N[(x)*(hb+1)+y] (gradientMex.cpp:242-251)
__device__ void phase_norm(const KcfPool& p, const float* __restrict__ E, float* __restrict__ N, int tid, int nt)
{
    const int hb = p.hb, wb = p.wb, hb1 = hb + 1, wb1 = wb + 1;
    const float eps = 1e-4f / 4 / 4 / 4 / 4 / 4;
    for (int i = tid; i < hb1 * wb1; i += nt) {
        int x = i / hb1, y = i - x * hb1;
        int cx = min(max(x, 1), wb - 1), cy = min(max(y, 1), hb - 1);  // replicated border
        const float* n = E + (cx - 1) * hb + (cy - 1);
        N[i] = 1.0f / sqrtf(n[0] + n[1] + n[hb] + n[hb + 1] + eps);
    }
}

// Phase 4: hogChannels (gradientMex.cpp:256-280, 313-315) x cos window
// (kcf.cpp:249-258).  F[(ch*wb + x)*ldf + y], ldf = 2*fh.
// Channels are produced in two halves (0: channels 0..15, 1: channels 16..30) so the feature / spectrum buffer
// holds 16 planes instead of 31 and two workgroups fit in one CU's LDS.
#define MOT_HALF0 16
template <int HALF, bool SOA>
__device__ void phase_channels(const KcfPool& p, const float* __restrict__ R1, const float* __restrict__ N,
                               float* __restrict__ F, float* __restrict__ feat_out, int feat_windowed, int tid, int nt)
{
    const int hb = p.hb, wb = p.wb, nb = p.nb, hb1 = hb + 1, ldf = 2 * p.fh;
    const float clip = 0.2f, r = .2357f;
    constexpr int C0 = HALF ? MOT_HALF0 : 0, C1 = HALF ? MOT_NCHAN : MOT_HALF0;
    for (int cell = tid; cell < nb; cell += nt) {
        uint32_t x, y; p.d_hb.divmod((uint32_t)cell, x, y);
        const float n0 = N[(x + 1) * hb1 + y + 1], n1 = N[(x + 1) * hb1 + y], n2 = N[x * hb1 + y + 1], n3 = N[x * hb1 + y];
        const float win = p.cos_win[cell];
        float tex0 = 0.f, tex1 = 0.f, tex2 = 0.f, tex3 = 0.f;
        float rlo[9];
#define PUT(ch, val) do { if ((ch) >= C0 && (ch) < C1) { F[(((ch) - C0) * wb + x) * ldf + y] = (val) * win; \
                          if (feat_out) feat_out[(ch) * nb + cell] = feat_windowed ? (val) * win : (val); } } while (0)
#pragma unroll
        for (int o = 0; o < MOT_NORI; o++) {
            if (HALF == 0 && o >= MOT_HALF0) continue;                  // half 0 needs the first 16 sensitive channels only
            const float v = R1[r1_index<SOA>(nb, cell, o)];
            if (o < 9) rlo[o] = v;
            float t0 = v * n0; if (t0 > clip) t0 = clip;
            float t1 = v * n1; if (t1 > clip) t1 = clip;
            float t2 = v * n2; if (t2 > clip) t2 = clip;
            float t3 = v * n3; if (t3 > clip) t3 = clip;
            float hv = 0.0f;
            hv += t0 * .5f; hv += t1 * .5f; hv += t2 * .5f; hv += t3 * .5f;
            PUT(o, hv);
            if (HALF == 1) {
                tex0 += t0 * r; tex1 += t1 * r; tex2 += t2 * r; tex3 += t3 * r;
                if (o >= 9) {
                    const float v2 = rlo[o - 9] + v;                    // R2 (:309)
                    float u0 = v2 * n0; if (u0 > clip) u0 = clip;
                    float u1 = v2 * n1; if (u1 > clip) u1 = clip;
                    float u2 = v2 * n2; if (u2 > clip) u2 = clip;
                    float u3 = v2 * n3; if (u3 > clip) u3 = clip;
                    float hi = 0.0f;
                    hi += u0 * .5f; hi += u1 * .5f; hi += u2 * .5f; hi += u3 * .5f;
                    PUT(18 + (o - 9), hi);
                }
            }
        }
        if (HALF == 1) {
            PUT(27, tex0); PUT(28, tex1); PUT(29, tex2); PUT(30, tex3);
            if (feat_out) feat_out[31 * nb + cell] = 0.0f;
        }
#undef PUT
    }
}

// The same channels for an arbitrary run [c0, c1) of the 31 (R1-resident HBM-slab templates: the DFTs take a few planes at a time).  Every value is
// produced by the statements of phase_channels above, in their order; the run is served in up to three parts -- contrast-sensitive channels (one
// orientation each), contrast-insensitive ones (two orientations each), the four texture channels (all 18) -- so a tile only pays for the
// orientations it stores.  F[((ch - c0)*wb + x)*ldf + y].
// win4: the cosine-window values of the thread's cells tid, tid + nt, ... (<= 4 of them), loaded once by the caller for all tiles.
__device__ __forceinline__ void phase_channels_tile(const KcfPool& p, const float* __restrict__ R1, const float* __restrict__ N,
                                                    float* __restrict__ F, int c0, int c1, float* __restrict__ feat_out, int feat_windowed, int tid, int nt,
                                                    const float (&win4)[4])
{
    const int hb = p.hb, wb = p.wb, nb = p.nb, hb1 = hb + 1, ldf = 2 * p.fh;
    const float clip = 0.2f, r = .2357f;
    const int s_lo = c0, s_hi = min(c1, MOT_NORI);                       // contrast-sensitive part: channel o = orientation o
    const int i_lo = max(c0, MOT_NORI) - MOT_NORI, i_hi = min(c1, 27) - MOT_NORI;   // contrast-insensitive part: channel 18 + j = orientations j, j + 9
    const bool tex = c1 > 27;
    int jc = 0;
#pragma unroll 1
    for (int cell = tid; cell < nb; cell += nt, jc++) {
        uint32_t x, y; p.d_hb.divmod((uint32_t)cell, x, y);
        const float n0 = N[(x + 1) * hb1 + y + 1], n1 = N[(x + 1) * hb1 + y], n2 = N[x * hb1 + y + 1], n3 = N[x * hb1 + y];
        const float win = jc == 0 ? win4[0] : (jc == 1 ? win4[1] : (jc == 2 ? win4[2] : win4[3]));
        const float* Rc = R1 + R1W(cell, 0);                            // orientation o at Rc[o * 64]
        float* Fc = F + x * ldf + y;                                    // channel ch at Fc[(ch - c0) * wb * ldf]
#define PUT(ch, val) do { Fc[((ch) - c0) * wb * ldf] = (val) * win; \
                          if (feat_out) feat_out[(ch) * nb + cell] = feat_windowed ? (val) * win : (val); } while (0)
#pragma unroll 1
        for (int o = s_lo; o < s_hi; o++) {
            const float v = Rc[o * 64];
            float t0 = v * n0; if (t0 > clip) t0 = clip;
            float t1 = v * n1; if (t1 > clip) t1 = clip;
            float t2 = v * n2; if (t2 > clip) t2 = clip;
            float t3 = v * n3; if (t3 > clip) t3 = clip;
            float hv = 0.0f;
            hv += t0 * .5f; hv += t1 * .5f; hv += t2 * .5f; hv += t3 * .5f;
            PUT(o, hv);
        }
#pragma unroll 1
        for (int j = i_lo; j < i_hi; j++) {
            const float v2 = Rc[j * 64] + Rc[(j + 9) * 64];             // R2 (:309)
            float u0 = v2 * n0; if (u0 > clip) u0 = clip;
            float u1 = v2 * n1; if (u1 > clip) u1 = clip;
            float u2 = v2 * n2; if (u2 > clip) u2 = clip;
            float u3 = v2 * n3; if (u3 > clip) u3 = clip;
            float hi = 0.0f;
            hi += u0 * .5f; hi += u1 * .5f; hi += u2 * .5f; hi += u3 * .5f;
            PUT(MOT_NORI + j, hi);
        }
        if (tex) {
            float tex0 = 0.f, tex1 = 0.f, tex2 = 0.f, tex3 = 0.f;
#pragma unroll
            for (int o = 0; o < MOT_NORI; o++) {
                const float v = Rc[o * 64];
                float t0 = v * n0; if (t0 > clip) t0 = clip;
                float t1 = v * n1; if (t1 > clip) t1 = clip;
                float t2 = v * n2; if (t2 > clip) t2 = clip;
                float t3 = v * n3; if (t3 > clip) t3 = clip;
                tex0 += t0 * r; tex1 += t1 * r; tex2 += t2 * r; tex3 += t3 * r;
            }
            if (27 >= c0) PUT(27, tex0);
            if (28 >= c0 && 28 < c1) PUT(28, tex1);
            if (29 >= c0 && 29 < c1) PUT(29, tex2);
            if (30 >= c0 && 30 < c1) PUT(30, tex3);
            if (feat_out && c1 == MOT_NCHAN) feat_out[31 * nb + cell] = 0.0f;
        }
#undef PUT
    }
}

// ---------------------------------------------------------------------------
// DFTs (replace FFTW; kcf.cpp:180-195 layout: n0 = f_cols slow, n1 = f_rows
// fast, half spectrum along n1).  Contraction allowed from here on.
// ---------------------------------------------------------------------------
#pragma clang fp contract(fast)

// generic forward r2c along y: F[(ch*wb+x)*ldf + y] (real) -> T[((ch*wb+x)*fh + k)] complex
__device__ void dft_rows_generic(const KcfPool& p, const float* __restrict__ F, float2* __restrict__ T,
                                 const float2* __restrict__ twr, int nch, int tid, int nt)
{
    const int hb = p.hb, fh = p.fh, ldf = 2 * fh, total = nch * p.wb * fh;
    for (int i = tid; i < total; i += nt) {
        uint32_t row, k; p.d_fh.divmod((uint32_t)i, row, k);
        const float* in = F + row * ldf;
        float re = 0.f, im = 0.f; int j = 0;
        for (int y = 0; y < hb; y++) {
            const float2 w = twr[j]; const float v = in[y];
            re += v * w.x; im -= v * w.y;
            j += (int)k; if (j >= hb) j -= hb;
        }
        T[i] = make_float2(re, im);
    }
}

// generic complex DFT along x (length wb): in[(ch*wb + x)*fh + k] -> out[(ch*wb + x')*fh + k]
template <int SIGN>
__device__ void dft_cols_generic(const KcfPool& p, const float2* __restrict__ in, float2* __restrict__ out,
                                 const float2* __restrict__ twc, int nch, int tid, int nt)
{
    const int wb = p.wb, fh = p.fh, plane = wb * fh, total = nch * plane;
    for (int i = tid; i < total; i += nt) {
        uint32_t ch, b; p.d_nbins.divmod((uint32_t)i, ch, b);
        uint32_t xp, k; p.d_fh.divmod(b, xp, k);
        const float2* src = in + ch * plane + k;
        float re = 0.f, im = 0.f; int j = 0;
#pragma unroll 4
        for (int x = 0; x < wb; x++) {
            const float2 w = twc[j]; const float2 v = src[x * fh];
            const float wi = (SIGN < 0) ? -w.y : w.y;
            re += v.x * w.x - v.y * wi; im += v.x * wi + v.y * w.x;
            j += (int)xp; if (j >= wb) j -= wb;
        }
        out[i] = make_float2(re, im);
    }
}

// The forward transforms of many planes, register-blocked: the twiddle of a row term depends on (k, y) but not on the line, the twiddle of a column
// term on (x', x) but not on k -- so a thread owns four lines (rows pass) / four adjacent bins (columns pass) of one twiddle sequence and pays
// one twiddle fetch and one index update for four complex multiply-adds.  Every output is the same sum in the same order as in the plain loops.
__device__ __forceinline__ void dft_rows_generic4(const KcfPool& p, const float* __restrict__ F, float2* __restrict__ T,
                                  const float2* __restrict__ twr, int nch, int tid, int nt)
{
    const int hb = p.hb, fh = p.fh, ldf = 2 * fh, lines = nch * p.wb, total = ((lines + 3) >> 2) * fh;
    for (int i = tid; i < total; i += nt) {
        uint32_t g, k; p.d_fh.divmod((uint32_t)i, g, k);
        const int l0 = 4 * (int)g;
        const float* in0 = F + l0 * ldf;
        const float* in1 = F + min(l0 + 1, lines - 1) * ldf;
        const float* in2 = F + min(l0 + 2, lines - 1) * ldf;
        const float* in3 = F + min(l0 + 3, lines - 1) * ldf;
        float re0 = 0.f, im0 = 0.f, re1 = 0.f, im1 = 0.f, re2 = 0.f, im2 = 0.f, re3 = 0.f, im3 = 0.f; int j = 0;
#pragma unroll 2
        for (int y = 0; y < hb; y++) {
            const float2 w = twr[j];
            const float v0 = in0[y], v1 = in1[y], v2 = in2[y], v3 = in3[y];
            re0 += v0 * w.x; im0 -= v0 * w.y; re1 += v1 * w.x; im1 -= v1 * w.y;
            re2 += v2 * w.x; im2 -= v2 * w.y; re3 += v3 * w.x; im3 -= v3 * w.y;
            j += (int)k; if (j >= hb) j -= hb;
        }
        float2* o = T + (size_t)l0 * fh + k;
        o[0] = make_float2(re0, im0);
        if (l0 + 1 < lines) o[fh] = make_float2(re1, im1);
        if (l0 + 2 < lines) o[2 * fh] = make_float2(re2, im2);
        if (l0 + 3 < lines) o[3 * fh] = make_float2(re3, im3);
    }
}
__device__ __forceinline__ void dft_cols_generic4(const KcfPool& p, const float2* __restrict__ in, float2* __restrict__ out,
                                  const float2* __restrict__ twc, int nch, int tid, int nt)
{
    const int wb = p.wb, fh = p.fh, plane = wb * fh, kb = (fh + 3) >> 2, per = wb * kb, total = nch * per;
    for (int i = tid; i < total; i += nt) {
        const int ch = i / per, rem = i - ch * per, xp = rem / kb, k0 = 4 * (rem - xp * kb);
        const int k1 = min(k0 + 1, fh - 1), k2 = min(k0 + 2, fh - 1), k3 = min(k0 + 3, fh - 1);
        const float2* src = in + ch * plane;
        float re0 = 0.f, im0 = 0.f, re1 = 0.f, im1 = 0.f, re2 = 0.f, im2 = 0.f, re3 = 0.f, im3 = 0.f; int j = 0;
#pragma unroll 2
        for (int x = 0; x < wb; x++) {
            const float2 w = twc[j]; const float wi = -w.y;             // forward
            const float2* sx = src + x * fh;
            const float2 a = sx[k0], b = sx[k1], c = sx[k2], d = sx[k3];
            re0 += a.x * w.x - a.y * wi; im0 += a.x * wi + a.y * w.x;
            re1 += b.x * w.x - b.y * wi; im1 += b.x * wi + b.y * w.x;
            re2 += c.x * w.x - c.y * wi; im2 += c.x * wi + c.y * w.x;
            re3 += d.x * w.x - d.y * wi; im3 += d.x * wi + d.y * w.x;
            j += xp; if (j >= wb) j -= wb;
        }
        float2* o = out + ch * plane + xp * fh;
        o[k0] = make_float2(re0, im0);
        if (k0 + 1 < fh) o[k0 + 1] = make_float2(re1, im1);
        if (k0 + 2 < fh) o[k0 + 2] = make_float2(re2, im2);
        if (k0 + 3 < fh) o[k0 + 3] = make_float2(re3, im3);
    }
}

// the column pass as two short passes for line lengths with a factor 2..5 (dft_ct.h).  Round 6: the DEFAULT build -- the whole GPU suite ran on it
// (profiles/r06_fftmix_full_suite.log: 133 passed); -DMOT_FFT_MIXED=0 keeps the direct column pass for A/B builds
#ifndef MOT_FFT_MIXED
#define MOT_FFT_MIXED 1
#endif
#ifndef MOT_FFT_MIXED_ROWS               /* the rows pass (real input) in two steps as well: host-checked (tests/test_dft_ct.py), never run on a GPU */
#define MOT_FFT_MIXED_ROWS 0
#endif
#ifndef MOT_FFT_MIXED_INPLACE            /* the in-place variant (72 / 76 px single pools) too: host-checked, never run on a GPU; costs kcf_predict_kernel<5> 544 B/lane of scratch */
#define MOT_FFT_MIXED_INPLACE 0
#endif
#if MOT_FFT_MIXED
#include "dft_ct.h"
#endif

// The same two passes IN PLACE (LDS-resident templates other than 20 x 20 cells, KcfPool::dft_inplace): a line's row spectrum occupies exactly the
// floats of the line (fh complex = ldf floats), a plane's spectrum those of its row spectra -- so every thread first computes ALL its outputs of a
// pass into registers (<= MOT_DFT_INPLACE_ITEMS items of 8 floats), the workgroup meets at a barrier, and only then the outputs overwrite the
// inputs.  No second buffer: region T leaves the layout (72 / 76 px: 90 / 95 -> 68 / 71 KB, two workgroups per CU instead of one).
#define MOT_DFT_INPLACE_ITEMS 3
// (compiled into the single-pool kernels only, kMode 5: its 24 accumulators cost the size-class kernels spills, and a size-class launch takes the
// largest class's LDS anyway; out of line it ran 40 % slower)
__device__ __forceinline__ void dft2_generic_inplace(int hb, int wb, int fh, FastDiv d_fh, float* __restrict__ B, const float2* __restrict__ twr,
                                                     const float2* __restrict__ twc, int nch, int tid, int nt)
{
    const int ldf = 2 * fh, lines = nch * wb, plane = wb * fh;
    float acc[MOT_DFT_INPLACE_ITEMS][8];
    {   // rows: item = (group of four lines, bin k)
        const int total = ((lines + 3) >> 2) * fh;
#pragma unroll
        for (int it = 0; it < MOT_DFT_INPLACE_ITEMS; it++) {
            const int i = tid + it * nt;
            if (i < total) {
                uint32_t g, k; d_fh.divmod((uint32_t)i, g, k);
                const int l0 = 4 * (int)g;
                const float* in0 = B + l0 * ldf;
                const float* in1 = B + min(l0 + 1, lines - 1) * ldf;
                const float* in2 = B + min(l0 + 2, lines - 1) * ldf;
                const float* in3 = B + min(l0 + 3, lines - 1) * ldf;
                float re0 = 0.f, im0 = 0.f, re1 = 0.f, im1 = 0.f, re2 = 0.f, im2 = 0.f, re3 = 0.f, im3 = 0.f; int j = 0;
#pragma unroll 2
                for (int y = 0; y < hb; y++) {
                    const float2 w = twr[j];
                    const float v0 = in0[y], v1 = in1[y], v2 = in2[y], v3 = in3[y];
                    re0 += v0 * w.x; im0 -= v0 * w.y; re1 += v1 * w.x; im1 -= v1 * w.y;
                    re2 += v2 * w.x; im2 -= v2 * w.y; re3 += v3 * w.x; im3 -= v3 * w.y;
                    j += (int)k; if (j >= hb) j -= hb;
                }
                acc[it][0] = re0; acc[it][1] = im0; acc[it][2] = re1; acc[it][3] = im1; acc[it][4] = re2; acc[it][5] = im2; acc[it][6] = re3; acc[it][7] = im3;
            }
        }
        __syncthreads();
        float2* T = reinterpret_cast<float2*>(B);
#pragma unroll
        for (int it = 0; it < MOT_DFT_INPLACE_ITEMS; it++) {
            const int i = tid + it * nt;
            if (i < total) {
                uint32_t g, k; d_fh.divmod((uint32_t)i, g, k);
                const int l0 = 4 * (int)g;
                float2* o = T + (size_t)l0 * fh + k;
                o[0] = make_float2(acc[it][0], acc[it][1]);
                if (l0 + 1 < lines) o[fh] = make_float2(acc[it][2], acc[it][3]);
                if (l0 + 2 < lines) o[2 * fh] = make_float2(acc[it][4], acc[it][5]);
                if (l0 + 3 < lines) o[3 * fh] = make_float2(acc[it][6], acc[it][7]);
            }
        }
        __syncthreads();
    }
    {   // columns: item = (plane, output line x', group of four bins)
        const int kb = (fh + 3) >> 2, per = wb * kb, total = nch * per;
        float2* S = reinterpret_cast<float2*>(B);
#if MOT_FFT_MIXED && MOT_FFT_MIXED_INPLACE
        // wb = n1 * N2: step A of dft_ct.h in place first; the items below then take step C's sums (dftct_cols_c_item) instead of the direct ones
        const int n1 = dftct_small_factor(wb);
        const float inv_n1 = n1 ? 1.0f / (float)n1 : 0.f;
        if (n1) { dftct_cols_a(S, twc, wb, n1, fh, nch, tid, nt); __syncthreads(); }
#endif
#pragma unroll
        for (int it = 0; it < MOT_DFT_INPLACE_ITEMS; it++) {
            const int i = tid + it * nt;
            if (i < total) {
                const int ch = i / per, rem = i - ch * per, xp = rem / kb, k0 = 4 * (rem - xp * kb);
#if MOT_FFT_MIXED && MOT_FFT_MIXED_INPLACE
                if (n1) { dftct_cols_c_item(S + ch * plane, twc, wb, n1, wb / n1, fh, xp, k0, inv_n1, acc[it]); continue; }
#endif
                const int k1 = min(k0 + 1, fh - 1), k2 = min(k0 + 2, fh - 1), k3 = min(k0 + 3, fh - 1);
                const float2* src = S + ch * plane;
                float re0 = 0.f, im0 = 0.f, re1 = 0.f, im1 = 0.f, re2 = 0.f, im2 = 0.f, re3 = 0.f, im3 = 0.f; int j = 0;
#pragma unroll 2
                for (int x = 0; x < wb; x++) {
                    const float2 w = twc[j]; const float wi = -w.y;     // forward
                    const float2* sx = src + x * fh;
                    const float2 a = sx[k0], b = sx[k1], c = sx[k2], d = sx[k3];
                    re0 += a.x * w.x - a.y * wi; im0 += a.x * wi + a.y * w.x;
                    re1 += b.x * w.x - b.y * wi; im1 += b.x * wi + b.y * w.x;
                    re2 += c.x * w.x - c.y * wi; im2 += c.x * wi + c.y * w.x;
                    re3 += d.x * w.x - d.y * wi; im3 += d.x * wi + d.y * w.x;
                    j += xp; if (j >= wb) j -= wb;
                }
                acc[it][0] = re0; acc[it][1] = im0; acc[it][2] = re1; acc[it][3] = im1; acc[it][4] = re2; acc[it][5] = im2; acc[it][6] = re3; acc[it][7] = im3;
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < MOT_DFT_INPLACE_ITEMS; it++) {
            const int i = tid + it * nt;
            if (i < total) {
                const int ch = i / per, rem = i - ch * per, xp = rem / kb, k0 = 4 * (rem - xp * kb);
                float2* o = S + ch * plane + xp * fh;
                o[k0] = make_float2(acc[it][0], acc[it][1]);
                if (k0 + 1 < fh) o[k0 + 1] = make_float2(acc[it][2], acc[it][3]);
                if (k0 + 2 < fh) o[k0 + 2] = make_float2(acc[it][4], acc[it][5]);
                if (k0 + 3 < fh) o[k0 + 3] = make_float2(acc[it][6], acc[it][7]);
            }
        }
        __syncthreads();
    }
}

// ---- DFTs as f32 matrix products on the matrix cores (v_mfma_f32_16x16x4_f32: exact f32 multiply-adds, k-ordered) ----
// A line transform of prime length (37 cells at 148 px) has no butterfly; as a product with the constant twiddle matrix
// it runs at the MFMA rate instead of one multiply-add and one twiddle fetch per VALU slot.  The constant operand comes
// from the pool's fragment-ordered tables and stays in registers for the whole call.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float* smem_base() { extern __shared__ __attribute__((aligned(16))) float smem_b[]; return smem_b; }

// rows: sF[line*ldf + y] (real, y < hb) -> sT[line*ldf + n], n = 2k + (re, im), k < fh
__device__ void dft_rows_mfma(const KcfPool& p, const float* __restrict__ sF, float* __restrict__ sT, int nlines, int tid, int nt_)
{
    const int lane = tid & 63, wave = tid >> 6, nw = nt_ >> 6;
    const int hb = p.hb, ldf = 2 * p.fh, ks = (hb + 3) >> 2, ntl = (ldf + 15) >> 4;
    const int mtiles = (nlines + 15) >> 4, kq = lane >> 4;
    for (int t = 0; t < ntl; t++) {                                    // 16 output floats (8 bins) per pass; its twiddle fragments stay in registers
        const float* tb = p.mf_rows + t * 64 + lane;
        asm volatile("" : "+v"(tb));                                   // keeps the fragment loads of all passes from being hoisted (and spilled) together
        float bw[MOT_MF_KS_R];
#pragma unroll
        for (int s = 0; s < MOT_MF_KS_R; s++) bw[s] = (s < ks) ? tb[s * 3 * 64] : 0.f;
        for (int mt = wave; mt < mtiles; mt += nw) {
            const int line = min(mt * 16 + (lane & 15), nlines - 1);
            const float* in = sF + line * ldf;
            f32x4 acc = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int s = 0; s < MOT_MF_KS_R; s++) {
                if (s < ks) {
                    const int k = 4 * s + kq;
                    const float a = (k < hb) ? in[min(k, hb - 1)] : 0.f;   // the pad floats of a line are not data
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[s], acc, 0, 0, 0);
                }
            }
            const int n = t * 16 + (lane & 15), l0 = mt * 16 + kq * 4;
            if (n < ldf) {
#pragma unroll
                for (int r = 0; r < 4; r++) if (l0 + r < nlines) sT[(l0 + r) * ldf + n] = acc[r];
            }
        }
    }
}

// columns, forward: sT[(ch*wb + x)*ldf + n] -> out[(ch*wb + x')*ldf + n]:
//   out_re = sum_x cos * T_re + sin * T_im,  out_im = sum_x cos * T_im - sin * T_re   (W = cos - i sin)
// as [cos | sin] (wb x 2wb) times [T ; T'] (2wb x ldf), T'[x][n] = n even ? T[x][n + 1] : -T[x][n - 1]
__device__ void dft_cols_mfma(const KcfPool& p, const float* __restrict__ sT, float* __restrict__ out, int g, int tid, int nt_)
{
    const int lane = tid & 63, wave = tid >> 6, nw = nt_ >> 6;
    const int wb = p.wb, ldf = 2 * p.fh, ks = (2 * wb + 3) >> 2, ntl = (ldf + 15) >> 4, mtl = (wb + 15) >> 4;
    const int kq = lane >> 4, units = g * ntl;
    for (int mt = 0; mt < mtl; mt++) {                                 // 16 output positions x' per pass; its twiddle fragments stay in registers
        const float* ta = p.mf_cols + mt * MOT_MF_KS_C * 64 + lane;
        asm volatile("" : "+v"(ta));                                   // see dft_rows_mfma
        float aw[MOT_MF_KS_C];
#pragma unroll
        for (int s = 0; s < MOT_MF_KS_C; s++) aw[s] = (s < ks) ? ta[s * 64] : 0.f;
        for (int u = wave; u < units; u += nw) {
            const int ch = u / ntl, nt = u - ch * ntl;
            const int n = nt * 16 + (lane & 15), nc = min(n, ldf - 1);
            const float* T = sT + ch * wb * ldf;
            f32x4 acc = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int s = 0; s < MOT_MF_KS_C; s++) {
                if (s < ks) {
                    const int k = 4 * s + kq;
                    const bool second = k >= wb;
                    const int x = min(second ? k - wb : k, wb - 1);
                    float v = T[x * ldf + (second ? (nc ^ 1) : nc)];
                    v = (second && (n & 1)) ? -v : v;
                    v = (k < 2 * wb && n < ldf) ? v : 0.f;
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[s], v, acc, 0, 0, 0);
                }
            }
            if (n < ldf) {
                float* o = out + (size_t)ch * wb * ldf + n;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int xp = mt * 16 + kq * 4 + r;
                    if (xp < wb) o[xp * ldf] = acc[r];
                }
            }
        }
    }
}

// Both passes in one: the row spectra of a channel never leave the accumulators.  A 16 x 16 result tile of the row product
// holds T[x = 16xt + 4(lane/16) + r][n = lane%16] in register r -- exactly the B-operand shape of a k-step of the column
// product if that step's k runs over x = 16xt + 4q + r, q = 0..3; the order of the k's of a sum is free, so the column
// product takes the tiles as they are and the CONSTANT operand (pool table mf_cols2) is stored in that k order.  T' (the
// re/im swap of the sine half) is the neighbouring lane's value: one DPP move.
//   F[(c*wb + x)*ldf + y] (LDS) -> out[(c*wb + x')*ldf + n]
__device__ __attribute__((noinline)) void dft2_mfma(const KcfPool& p, int f_off, int cw_off, float* __restrict__ out, int nch, int tid, int nt_)
{
    extern __shared__ __attribute__((aligned(16))) float dft2_smem[];
    const float* __restrict__ F = dft2_smem + f_off;                   // LDS (kept out of line: a register allocation of its own)
    float* __restrict__ cw = dft2_smem + cw_off;                       // LDS copy of the column fragments [mt][xt][r][cos, sin][lane]
    const int lane = tid & 63, wave = tid >> 6, nw = nt_ >> 6, q = lane >> 4, m = lane & 15;
    const int hb = p.hb, wb = p.wb, ldf = 2 * p.fh, ksr = (hb + 3) >> 2, ntl = (ldf + 15) >> 4, xtl = (wb + 15) >> 4;
    for (int i = tid; i < 3 * 3 * 4 * 2 * 64; i += nt_) cw[i] = p.mf_cols2[i];
    __syncthreads();
    const float* cwl = cw + lane;
    for (int t = 0; t < ntl; t++) {                                    // 16 output floats (8 bins) per pass
        const float* tb = p.mf_rows + t * 64 + lane;
        asm volatile("" : "+v"(tb));                                   // keeps the fragment loads of all passes from being hoisted (and spilled) together
        float bw[MOT_MF_KS_R];
#pragma unroll
        for (int s = 0; s < MOT_MF_KS_R; s++) bw[s] = (s < ksr) ? tb[s * 3 * 64] : 0.f;
        const int n = t * 16 + m;
        for (int ch = wave; ch < nch; ch += nw) {
            const float* Fc = F + ch * wb * ldf;
            float v[3][4], v2[3][4];
#pragma unroll
            for (int xt = 0; xt < 3; xt++) {
                f32x4 acc = { 0.f, 0.f, 0.f, 0.f };
                if (xt < xtl) {
                    const int x = xt * 16 + m;
                    const float* in = Fc + min(x, wb - 1) * ldf;
#pragma unroll
                    for (int s = 0; s < MOT_MF_KS_R; s++) {
                        if (s < ksr) {
                            const int k = 4 * s + q;
                            const float a = (x < wb && k < hb) ? in[min(k, hb - 1)] : 0.f;   // pad lines / pad floats are not data
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[s], acc, 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    v[xt][r] = acc[r];
                    const float pv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[r]), 0xB1, 0xF, 0xF, false));   // lane ^ 1
                    v2[xt][r] = (n & 1) ? -pv : pv;
                }
            }
#pragma unroll
            for (int mt = 0; mt < 3; mt++) {
                if (mt < xtl) {
                    f32x4 o = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                    for (int xt = 0; xt < 3; xt++) {
                        if (xt < xtl) {
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                o = __builtin_amdgcn_mfma_f32_16x16x4f32(cwl[(((mt * 3 + xt) * 4 + r) * 2 + 0) * 64], v[xt][r], o, 0, 0, 0);
                                o = __builtin_amdgcn_mfma_f32_16x16x4f32(cwl[(((mt * 3 + xt) * 4 + r) * 2 + 1) * 64], v2[xt][r], o, 0, 0, 0);
                            }
                        }
                    }
                    if (n < ldf) {
                        float* op = out + (size_t)ch * wb * ldf + n;
#pragma unroll
                        for (int r = 0; r < 4; r++) { const int xp = mt * 16 + q * 4 + r; if (xp < wb) op[xp * ldf] = o[r]; }
                    }
                }
            }
        }
    }
}

// The same chain with every trip count a compile-time constant (XTL tiles of 16 lines, KSR k-steps of 4 rows): no exec-mask
// branch around any MFMA, and the 8 * XTL * XTL constant column fragments of a lane are loaded ONCE (from the pool table in
// global memory: no LDS copy, no barrier) and stay in registers over all passes and channels -- the generic version reads them from LDS
// again for every (pass, channel), one LDS read per MFMA.  Same products in the same order: bit-identical output.
template <int XTL, int KSR>
__device__ __attribute__((noinline)) void dft2_mfma_fixed(const KcfPool& p, int f_off, float* __restrict__ out, int nch, int tid, int nt_)
{
    extern __shared__ __attribute__((aligned(16))) float dft2_smem[];
    const float* __restrict__ F = dft2_smem + f_off;
    const int lane = tid & 63, wave = tid >> 6, nw = nt_ >> 6, q = lane >> 4, m = lane & 15;
    const int hb = p.hb, wb = p.wb, ldf = 2 * p.fh, ntl = (ldf + 15) >> 4;
    float cwr[XTL][XTL][4][2];
#pragma unroll
    for (int mt = 0; mt < XTL; mt++)
#pragma unroll
        for (int xt = 0; xt < XTL; xt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                cwr[mt][xt][r][0] = p.mf_cols2[(((mt * 3 + xt) * 4 + r) * 2 + 0) * 64 + lane];
                cwr[mt][xt][r][1] = p.mf_cols2[(((mt * 3 + xt) * 4 + r) * 2 + 1) * 64 + lane];
            }
    for (int t = 0; t < ntl; t++) {
        float bw[KSR];
#pragma unroll
        for (int s = 0; s < KSR; s++) bw[s] = p.mf_rows[t * 64 + lane + s * 3 * 64];
        const int n = t * 16 + m;
        for (int ch = wave; ch < nch; ch += nw) {
            const float* Fc = F + ch * wb * ldf;
            float v[XTL][4], v2[XTL][4];
#pragma unroll
            for (int xt = 0; xt < XTL; xt++) {
                f32x4 acc = { 0.f, 0.f, 0.f, 0.f };
                const int x = xt * 16 + m;
                const float* in = Fc + min(x, wb - 1) * ldf;
                float a[KSR];
#pragma unroll
                for (int s = 0; s < KSR; s++) { const int k = 4 * s + q; a[s] = in[min(k, hb - 1)]; a[s] = (x < wb && k < hb) ? a[s] : 0.f; }   // all LDS reads of the tile first
#pragma unroll
                for (int s = 0; s < KSR; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bw[s], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    v[xt][r] = acc[r];
                    const float pv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[r]), 0xB1, 0xF, 0xF, false));   // lane ^ 1
                    v2[xt][r] = (n & 1) ? -pv : pv;
                }
            }
#pragma unroll
            for (int mt = 0; mt < XTL; mt++) {
                f32x4 o = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                for (int xt = 0; xt < XTL; xt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        o = __builtin_amdgcn_mfma_f32_16x16x4f32(cwr[mt][xt][r][0], v[xt][r], o, 0, 0, 0);
                        o = __builtin_amdgcn_mfma_f32_16x16x4f32(cwr[mt][xt][r][1], v2[xt][r], o, 0, 0, 0);
                    }
                if (n < ldf) {
                    float* op = out + (size_t)ch * wb * ldf + n;
#pragma unroll
                    for (int r = 0; r < 4; r++) { const int xp = mt * 16 + q * 4 + r; if (xp < wb) op[xp * ldf] = o[r]; }
                }
            }
        }
    }
}

// R1-resident HBM-slab templates: channels [c_lo, c_hi) -> windowed feature planes -> spectra, p.tile_T planes at a time through the LDS work area
// (offW on: E, zf, tmp are dead while this runs).  The transform of a plane is the MFMA chain of dft2_mfma_fixed (same products, same order:
// bit-identical spectra); what differs is the work split -- the unit is (plane, pass of 16 output floats), spread over the waves, so a tile of
// 5 planes still feeds all 8 waves -- and that the 8 * XTL * XTL column fragments are loaded once for all tiles of the call.
// out[((ch - c_lo)*wb + x')*ldf + n].
template <int XTL, int KSR>
__device__ __attribute__((noinline)) void spectrum_tiles_r1(const KcfPool& p, const float* __restrict__ N, int c_lo, int c_hi, float* __restrict__ out,
                                                            float* __restrict__ feat_out, int feat_windowed, int tid, int nt_, long long* dbg)
{
    extern __shared__ __attribute__((aligned(16))) float dft2_smem[];
    const float* __restrict__ R1 = dft2_smem + p.offR1c;               // N: the norm matrix (region C, LDS), from the caller's carve()
    float* __restrict__ F = dft2_smem + p.offW;
    const int lane = tid & 63, wave = tid >> 6, nw = nt_ >> 6, q = lane >> 4, m = lane & 15;
    const int hb = p.hb, wb = p.wb, ldf = 2 * p.fh, ntl = (ldf + 15) >> 4;
    float cwr[XTL][XTL][4][2];
#pragma unroll
    for (int mt = 0; mt < XTL; mt++)
#pragma unroll
        for (int xt = 0; xt < XTL; xt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                cwr[mt][xt][r][0] = p.mf_cols2[(((mt * 3 + xt) * 4 + r) * 2 + 0) * 64 + lane];
                cwr[mt][xt][r][1] = p.mf_cols2[(((mt * 3 + xt) * 4 + r) * 2 + 1) * 64 + lane];
            }
    // the row fragments of all (<= 3) passes and the window values of the thread's cells: no global load inside the tile loop
    float bw0[KSR], bw1[KSR], bw2[KSR], win4[4];
#pragma unroll
    for (int s = 0; s < KSR; s++) { bw0[s] = p.mf_rows[lane + s * 3 * 64]; bw1[s] = p.mf_rows[64 + lane + s * 3 * 64]; bw2[s] = ntl > 2 ? p.mf_rows[128 + lane + s * 3 * 64] : 0.f; }
#pragma unroll
    for (int j = 0; j < 4; j++) win4[j] = p.cos_win[min(tid + j * nt_, p.nb - 1)];
#define TSTAMP(i) do { if (dbg && blockIdx.x == 0 && tid == 0 && c0 == 0) dbg[i] = wall_clock64(); } while (0)
    for (int c0 = c_lo; c0 < c_hi; c0 += p.tile_T) {
        const int c1 = min(c0 + p.tile_T, c_hi), nch = c1 - c0;
        TSTAMP(12);
        phase_channels_tile(p, R1, N, F, c0, c1, feat_out, feat_windowed, tid, nt_, win4);
        __syncthreads();
        TSTAMP(13);
        const int units = nch * ntl;
        for (int u = wave; u < units; u += nw) {
            const int t = u / nch, ch = u - t * nch;
            float bw[KSR];
#pragma unroll
            for (int s = 0; s < KSR; s++) bw[s] = t == 0 ? bw0[s] : (t == 1 ? bw1[s] : bw2[s]);
            const int n = t * 16 + m;
            const float* Fc = F + ch * wb * ldf;
            float v[XTL][4], v2[XTL][4];
#pragma unroll
            for (int xt = 0; xt < XTL; xt++) {
                f32x4 acc = { 0.f, 0.f, 0.f, 0.f };
                const int x = xt * 16 + m;
                const float* in = Fc + min(x, wb - 1) * ldf;
                float a[KSR];
#pragma unroll
                for (int s = 0; s < KSR; s++) { const int k = 4 * s + q; a[s] = in[min(k, hb - 1)]; a[s] = (x < wb && k < hb) ? a[s] : 0.f; }
#pragma unroll
                for (int s = 0; s < KSR; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bw[s], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    v[xt][r] = acc[r];
                    const float pv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[r]), 0xB1, 0xF, 0xF, false));   // lane ^ 1
                    v2[xt][r] = (n & 1) ? -pv : pv;
                }
            }
#pragma unroll
            for (int mt = 0; mt < XTL; mt++) {
                f32x4 o = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                for (int xt = 0; xt < XTL; xt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        o = __builtin_amdgcn_mfma_f32_16x16x4f32(cwr[mt][xt][r][0], v[xt][r], o, 0, 0, 0);
                        o = __builtin_amdgcn_mfma_f32_16x16x4f32(cwr[mt][xt][r][1], v2[xt][r], o, 0, 0, 0);
                    }
                if (n < ldf) {
                    float* op = out + (size_t)(c0 - c_lo + ch) * wb * ldf + n;
#pragma unroll
                    for (int r = 0; r < 4; r++) { const int xp = mt * 16 + q * 4 + r; if (xp < wb) op[xp * ldf] = o[r]; }
                }
            }
        }
        TSTAMP(14);
        __syncthreads();
        TSTAMP(15);
    }
#undef TSTAMP
}
// the instances: XTL = tiles of 16 lines (wb), KSR = k-steps of 4 rows (hb)
__device__ __forceinline__ bool r1_spectrum_dispatch(const KcfPool& p, const float* N, int c_lo, int c_hi, float* out, float* fo, int fw, int tid, int nt, long long* dbg)
{
    const int xtl = (p.wb + 15) >> 4, ksr = (p.hb + 3) >> 2;
    if (xtl == 3 && ksr == 10) { spectrum_tiles_r1<3, 10>(p, N, c_lo, c_hi, out, fo, fw, tid, nt, dbg); return true; }
    if (xtl == 3 && ksr == 9) { spectrum_tiles_r1<3, 9>(p, N, c_lo, c_hi, out, fo, fw, tid, nt, dbg); return true; }
    if (xtl == 2 && ksr == 8) { spectrum_tiles_r1<2, 8>(p, N, c_lo, c_hi, out, fo, fw, tid, nt, dbg); return true; }
    if (xtl == 2 && ksr == 7) { spectrum_tiles_r1<2, 7>(p, N, c_lo, c_hi, out, fo, fw, tid, nt, dbg); return true; }
    if (xtl == 2 && ksr == 10) { spectrum_tiles_r1<2, 10>(p, N, c_lo, c_hi, out, fo, fw, tid, nt, dbg); return true; }
    return false;
}

// ---- radix 4x5 prime-factor 20-point transforms, one thread per transform ----
#define C1_5 0.30901699437494742f   /* cos(2pi/5) */
#define C2_5 (-0.80901699437494742f) /* cos(4pi/5) */
#define S1_5 0.95105651629515357f   /* sin(2pi/5) */
#define S2_5 0.58778525229247313f   /* sin(4pi/5) */

// real input x[20] -> X[0..10]; in place on a 22-float row
__device__ __forceinline__ void rfft20_inplace(float* __restrict__ row)
{
    float x[20];
#pragma unroll
    for (int i = 0; i < 10; i++) { float2 v = *reinterpret_cast<const float2*>(row + 2 * i); x[2 * i] = v.x; x[2 * i + 1] = v.y; }
    // stage 1: four real 5-point DFTs over n2 (input n = (5*n1 + 4*n2) mod 20), outputs k2 = 0,1,2
    float y0[4], y1r[4], y1i[4], y2r[4], y2i[4];
#pragma unroll
    for (int n1 = 0; n1 < 4; n1++) {
        const float a0 = x[(5 * n1) % 20], a1 = x[(5 * n1 + 4) % 20], a2 = x[(5 * n1 + 8) % 20], a3 = x[(5 * n1 + 12) % 20], a4 = x[(5 * n1 + 16) % 20];
        const float s14 = a1 + a4, d14 = a1 - a4, s23 = a2 + a3, d23 = a2 - a3;
        y0[n1] = a0 + s14 + s23;
        y1r[n1] = a0 + C1_5 * s14 + C2_5 * s23; y1i[n1] = -(S1_5 * d14 + S2_5 * d23);
        y2r[n1] = a0 + C2_5 * s14 + C1_5 * s23; y2i[n1] = -(S2_5 * d14 - S1_5 * d23);
    }
    float2 X[11];
    {   // k2 = 0 (real): k = 0,5,10
        const float t0 = y0[0] + y0[2], t1 = y0[0] - y0[2], t2 = y0[1] + y0[3], t3 = y0[1] - y0[3];
        X[0] = make_float2(t0 + t2, 0.f); X[10] = make_float2(t0 - t2, 0.f); X[5] = make_float2(t1, -t3);
    }
    {   // k2 = 1: k1=0 -> 16 (=conj 4), k1=1 -> 1, k1=2 -> 6, k1=3 -> 11 (=conj 9)
        const float t0r = y1r[0] + y1r[2], t0i = y1i[0] + y1i[2], t1r = y1r[0] - y1r[2], t1i = y1i[0] - y1i[2];
        const float t2r = y1r[1] + y1r[3], t2i = y1i[1] + y1i[3], t3r = y1r[1] - y1r[3], t3i = y1i[1] - y1i[3];
        X[4] = make_float2(t0r + t2r, -(t0i + t2i));            // conj(Z0)
        X[6] = make_float2(t0r - t2r, t0i - t2i);                // Z2
        X[1] = make_float2(t1r + t3i, t1i - t3r);                // Z1 = t1 - i t3
        X[9] = make_float2(t1r - t3i, -(t1i + t3r));             // conj(Z3), Z3 = t1 + i t3
    }
    {   // k2 = 2: k1=0 -> 12 (=conj 8), k1=1 -> 17 (=conj 3), k1=2 -> 2, k1=3 -> 7
        const float t0r = y2r[0] + y2r[2], t0i = y2i[0] + y2i[2], t1r = y2r[0] - y2r[2], t1i = y2i[0] - y2i[2];
        const float t2r = y2r[1] + y2r[3], t2i = y2i[1] + y2i[3], t3r = y2r[1] - y2r[3], t3i = y2i[1] - y2i[3];
        X[8] = make_float2(t0r + t2r, -(t0i + t2i));            // conj(Z0)
        X[2] = make_float2(t0r - t2r, t0i - t2i);                // Z2
        X[3] = make_float2(t1r + t3i, -(t1i - t3r));             // conj(Z1)
        X[7] = make_float2(t1r - t3i, t1i + t3r);                // Z3
    }
#pragma unroll
    for (int k = 0; k < 11; k++) *reinterpret_cast<float2*>(row + 2 * k) = X[k];
}

// complex 20-point DFT, stride `st` (in float2), in place.  SIGN=-1 forward, +1 inverse (unnormalised)
template <int SIGN>
__device__ __forceinline__ void cfft20_inplace(float2* __restrict__ base, int st)
{
    float2 x[20];
#pragma unroll
    for (int i = 0; i < 20; i++) x[i] = base[i * st];
    float2 Y[4][5];
#pragma unroll
    for (int n1 = 0; n1 < 4; n1++) {
        const float2 a0 = x[(5 * n1) % 20], a1 = x[(5 * n1 + 4) % 20], a2 = x[(5 * n1 + 8) % 20], a3 = x[(5 * n1 + 12) % 20], a4 = x[(5 * n1 + 16) % 20];
        const float2 s14 = make_float2(a1.x + a4.x, a1.y + a4.y), d14 = make_float2(a1.x - a4.x, a1.y - a4.y);
        const float2 s23 = make_float2(a2.x + a3.x, a2.y + a3.y), d23 = make_float2(a2.x - a3.x, a2.y - a3.y);
        Y[n1][0] = make_float2(a0.x + s14.x + s23.x, a0.y + s14.y + s23.y);
        const float2 R1 = make_float2(a0.x + C1_5 * s14.x + C2_5 * s23.x, a0.y + C1_5 * s14.y + C2_5 * s23.y);
        const float2 R2 = make_float2(a0.x + C2_5 * s14.x + C1_5 * s23.x, a0.y + C2_5 * s14.y + C1_5 * s23.y);
        const float2 I1 = make_float2(S1_5 * d14.x + S2_5 * d23.x, S1_5 * d14.y + S2_5 * d23.y);
        const float2 I2 = make_float2(S2_5 * d14.x - S1_5 * d23.x, S2_5 * d14.y - S1_5 * d23.y);
        // forward: Y1 = R1 - i*I1 = (R1.x + I1.y, R1.y - I1.x); inverse: Y1 = R1 + i*I1
        if (SIGN < 0) {
            Y[n1][1] = make_float2(R1.x + I1.y, R1.y - I1.x); Y[n1][4] = make_float2(R1.x - I1.y, R1.y + I1.x);
            Y[n1][2] = make_float2(R2.x + I2.y, R2.y - I2.x); Y[n1][3] = make_float2(R2.x - I2.y, R2.y + I2.x);
        } else {
            Y[n1][1] = make_float2(R1.x - I1.y, R1.y + I1.x); Y[n1][4] = make_float2(R1.x + I1.y, R1.y - I1.x);
            Y[n1][2] = make_float2(R2.x - I2.y, R2.y + I2.x); Y[n1][3] = make_float2(R2.x + I2.y, R2.y - I2.x);
        }
    }
#pragma unroll
    for (int k2 = 0; k2 < 5; k2++) {
        const float2 t0 = make_float2(Y[0][k2].x + Y[2][k2].x, Y[0][k2].y + Y[2][k2].y);
        const float2 t1 = make_float2(Y[0][k2].x - Y[2][k2].x, Y[0][k2].y - Y[2][k2].y);
        const float2 t2 = make_float2(Y[1][k2].x + Y[3][k2].x, Y[1][k2].y + Y[3][k2].y);
        const float2 t3 = make_float2(Y[1][k2].x - Y[3][k2].x, Y[1][k2].y - Y[3][k2].y);
        const float2 Z0 = make_float2(t0.x + t2.x, t0.y + t2.y), Z2 = make_float2(t0.x - t2.x, t0.y - t2.y);
        float2 Z1, Z3;   // forward: Z1 = t1 - i t3, Z3 = t1 + i t3
        if (SIGN < 0) { Z1 = make_float2(t1.x + t3.y, t1.y - t3.x); Z3 = make_float2(t1.x - t3.y, t1.y + t3.x); }
        else { Z1 = make_float2(t1.x - t3.y, t1.y + t3.x); Z3 = make_float2(t1.x + t3.y, t1.y - t3.x); }
        base[((0 * 5 + 16 * k2) % 20) * st] = Z0;
        base[((1 * 5 + 16 * k2) % 20) * st] = Z1;
        base[((2 * 5 + 16 * k2) % 20) * st] = Z2;
        base[((3 * 5 + 16 * k2) % 20) * st] = Z3;
    }
}

// forward 2-D r2c of `nch` feature planes; result S[(ch*wb + x')*fh + k] in region B.
// F lives in region B (row stride ldf = 2*fh floats); T is the ping-pong buffer of the generic path.
template <bool SLAB, bool GEN = true, bool INPL = false>   // GEN false: only the 20 x 20 register FFT is compiled in (kMode 1); INPL: the in-place passes too (kMode 5)
__device__ __forceinline__ void fft_forward(const KcfPool& p, float* __restrict__ regT, float* __restrict__ regB,
                            const float2* __restrict__ twr, const float2* __restrict__ twc, int nch, int tid, int nt,
                            float* __restrict__ stage = nullptr)
{
    if (!SLAB && p.fft20) {
        const int nrows = nch * p.wb;
        for (int r = tid; r < nrows; r += nt) rfft20_inplace(regB + r * 22);
        __syncthreads();
        float2* S = reinterpret_cast<float2*>(regB);
        const int ncol = nch * 11;
        for (int i = tid; i < ncol; i += nt) {
            const int ch = i / 11, k = i - ch * 11;
            cfft20_inplace<-1>(S + ch * 220 + k, 11);
        }
        __syncthreads();
    } else if (SLAB && stage && p.stage_G > 0) {
        // HBM-slab templates: G channel planes at a time through LDS (features in, rows DFT, columns DFT, spectrum out
        // in place: a channel's spectrum occupies exactly the bytes of its feature plane); same arithmetic as below
        const int planeF = p.wb * 2 * p.fh;                           // floats of one feature plane == floats of its half spectrum
        float* sF = stage; float2* sT = reinterpret_cast<float2*>(stage + p.stage_G * planeF);
        for (int c0 = 0; c0 < nch; c0 += p.stage_G) {
            const int g = min(p.stage_G, nch - c0);
            const float2* src = reinterpret_cast<const float2*>(regB + (size_t)c0 * planeF);   // planeF is even
            for (int i = tid; i < g * planeF / 2; i += nt) reinterpret_cast<float2*>(sF)[i] = src[i];
            __syncthreads();
            if (p.mf) {
                dft_rows_mfma(p, sF, reinterpret_cast<float*>(sT), g * p.wb, tid, nt);
                __syncthreads();
                dft_cols_mfma(p, reinterpret_cast<const float*>(sT), regB + (size_t)c0 * planeF, g, tid, nt);
                __syncthreads();
                continue;
            }
#if MOT_FFT_MIXED && MOT_FFT_MIXED_ROWS
            if (const int r1 = dftct_rows_factor(p.hb)) {
                dftct_rows_a(sF, twr, p.hb, r1, 2 * p.fh, g * p.wb, tid, nt);
                __syncthreads();
                dftct_rows_c(sF, sT, twr, p.hb, r1, p.fh, 2 * p.fh, g * p.wb, tid, nt);
            } else
#endif
            dft_rows_generic4(p, sF, sT, twr, g, tid, nt);                // lines longer than MOT_DFT_MFMA_MAX cells (templates beyond 164 px)
            __syncthreads();
#if MOT_FFT_MIXED
            if (const int n1 = dftct_small_factor(p.wb)) {
                dftct_cols_a(sT, twc, p.wb, n1, p.fh, g, tid, nt);
                __syncthreads();
                dftct_cols_c(sT, reinterpret_cast<float2*>(regB) + (size_t)c0 * p.nbins, twc, p.wb, n1, p.fh, g, tid, nt);
                __syncthreads();
                continue;
            }
#endif
            dft_cols_generic4(p, sT, reinterpret_cast<float2*>(regB) + (size_t)c0 * p.nbins, twc, g, tid, nt);
            __syncthreads();
        }
    } else if (!GEN) {
        __builtin_trap();                                              // launch_kcf_* pick mode 3 for every pool that is not 20 x 20 cells
    } else if (!SLAB && p.dft_inplace) {
        if (!INPL) __builtin_trap();                                   // only single-pool launches of such a pool exist (get_pool: size classes keep region T)
        dft2_generic_inplace(p.hb, p.wb, p.fh, p.d_fh, regB, twr, twc, nch, tid, nt);
    } else {
        float2* T = reinterpret_cast<float2*>(regT);
#if MOT_FFT_MIXED && MOT_FFT_MIXED_ROWS
        if (const int r1 = dftct_rows_factor(p.hb)) {                  // hb = r1 * N2: step A in place on the feature lines, step C into region T (dft_ct.h)
            dftct_rows_a(regB, twr, p.hb, r1, 2 * p.fh, nch * p.wb, tid, nt);
            __syncthreads();
            dftct_rows_c(regB, T, twr, p.hb, r1, p.fh, 2 * p.fh, nch * p.wb, tid, nt);
        } else
#endif
        dft_rows_generic4(p, regB, T, twr, nch, tid, nt);
        __syncthreads();
#if MOT_FFT_MIXED
        if (const int n1 = dftct_small_factor(p.wb)) {                 // wb = n1 * N2: step A in place on T, step C into region B (dft_ct.h)
            dftct_cols_a(T, twc, p.wb, n1, p.fh, nch, tid, nt);
            __syncthreads();
            dftct_cols_c(T, reinterpret_cast<float2*>(regB), twc, p.wb, n1, p.fh, nch, tid, nt);
            __syncthreads();
            return;
        }
#endif
        dft_cols_generic4(p, T, reinterpret_cast<float2*>(regB), twc, nch, tid, nt);
        __syncthreads();
    }
}

// inverse c2r of one plane: Z[(x')*fh + k] -> resp[x*hb + y], unnormalised (kcf.cpp:399)
__device__ void fft_inverse_plane(const KcfPool& p, const float2* __restrict__ Z, float2* __restrict__ tmp,
                                  float* __restrict__ resp, const float2* __restrict__ twr,
                                  const float2* __restrict__ twc, int tid, int nt)
{
    const int hb = p.hb, wb = p.wb, fh = p.fh;
    dft_cols_generic<+1>(p, Z, tmp, twc, 1, tid, nt);
    __syncthreads();
    const bool even = (hb % 2 == 0);
    for (int i = tid; i < p.nb; i += nt) {
        uint32_t x, y; p.d_hb.divmod((uint32_t)i, x, y);
        const float2* src = tmp + x * fh;
        float acc = src[0].x; int j = 0;
#pragma unroll 4
        for (int k = 1; k < fh; k++) {
            j += (int)y; if (j >= hb) j -= hb;
            const float2 w = twr[j]; const float2 v = src[k];
            const float term = v.x * w.x - v.y * w.y;
            acc += (even && k == fh - 1) ? term : 2.0f * term;
        }
        resp[i] = acc;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// block-wide arg-max with the reference's scan semantics (kcf.cpp:402-418):
// column-major scan, strict '>' against -99999 => first maximum wins.
// ---------------------------------------------------------------------------
__device__ int block_argmax_first(const float* __restrict__ resp, int n, float* __restrict__ red_v, int* __restrict__ red_i,
                                  int tid, int nt)
{
    float bv = -99999.0f; int bi = -1;
    for (int i = tid; i < n; i += nt) { const float v = resp[i]; if (v > bv) { bv = v; bi = i; } }
    // within a thread indices increase, so '>' keeps the first
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(bv, off); const int oi = __shfl_down(bi, off);
        if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
    }
    const int wave = tid >> 6, nw = (nt + 63) >> 6;
    if ((tid & 63) == 0) { red_v[wave] = bv; red_i[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < nw; w++) {
            const float ov = red_v[w]; const int oi = red_i[w];
            if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
        }
        red_i[0] = bi;
    }
    __syncthreads();
    return red_i[0];
}

template <bool SLAB, bool GEN, bool INPL>
__device__ __attribute__((noinline)) void fft_forward_ool(const KcfPool& p, float* __restrict__ regT, float* __restrict__ regB,
                                                          const float2* __restrict__ twr, const float2* __restrict__ twc, int nch, int tid, int nt, float* __restrict__ stage)
{
    fft_forward<SLAB, GEN, INPL>(p, regT, regB, twr, twc, nch, tid, nt, stage);
}

struct Regions {
    float* A; float* B; float* C; float* T;
    float* E; float* N; float2* twr; float2* twc; float2* zf; float2* tmp; float* resp; float* red_v; int* red_i;
    uint16_t* tab;
};

__device__ __forceinline__ Regions carve(const KcfPool& p, float* base, float* cbase = nullptr, bool r1_order = false)
{
    Regions r;
    r.A = base + p.offA; r.B = base + p.offB; r.C = cbase ? cbase : base + p.offC; r.T = base + p.offT;
    float* c = r.C;
    r.tab = reinterpret_cast<uint16_t*>(c); c += 2048;               // 4096 u16
    r.twr = reinterpret_cast<float2*>(c); c += 2 * p.hb;
    r.twc = reinterpret_cast<float2*>(c); c += 2 * p.wb;
    if (r1_order) {
        // R1-resident HBM-slab templates: what the gradient / histogram stripes may overwrite (E, zf, tmp) comes last, next to the work area
        r.red_v = c; c += 16;
        r.red_i = reinterpret_cast<int*>(c); c += 16;
        r.N = c; c += (p.hb + 1) * (p.wb + 1);
        c = r.C + (p.offX - p.offR1c - MOT_NORI * 64 * ((p.nb + 63) >> 6));
        r.E = c; r.resp = c; c += (p.nb + 3) & ~3;
        r.zf = reinterpret_cast<float2*>(c); c += 2 * p.nbins;
        r.tmp = reinterpret_cast<float2*>(c);
        return r;
    }
    r.zf = reinterpret_cast<float2*>(c); c += 2 * p.nbins;
    r.tmp = reinterpret_cast<float2*>(c); c += 2 * p.nbins;
    r.E = c; r.resp = c; c += p.nb;                                 // the cell energies are dead long before the response exists
    r.N = c; c += (p.hb + 1) * (p.wb + 1);
    r.red_v = c; c += 16;
    r.red_i = reinterpret_cast<int*>(c); c += 16;
    return r;
}

// Everything up to R1 (region A) and the norm matrix: shared by predict / update.
template <bool SLAB, bool LDSR1 = true, bool R1 = false, bool HB = (MOT_HIST_BATCH != 0)>   // HB: phase_hist's BATCH
__device__ void features_prepare(const KcfPool& p, const KcfLaunch& l, int item, bbox_t box, const Regions& r, int tid, int nt, float* stage = nullptr)
{
#define DBG_STAMP(i) do { if (l.dbg && blockIdx.x == 0 && tid == 0) l.dbg[i] = wall_clock64(); } while (0)
    DBG_STAMP(0);
    // stage constants
    for (int i = tid; i < 2048; i += nt) reinterpret_cast<uint32_t*>(r.tab)[i] = reinterpret_cast<const uint32_t*>(p.sse_tab)[i];
    for (int i = tid; i < p.hb; i += nt) r.twr[i] = p.tw_r[i];
    for (int i = tid; i < p.wb; i += nt) r.twc[i] = p.tw_c[i];
    const float* patch = l.patches ? l.patches + (size_t)item * p.rows * p.cols : nullptr;
    if (SLAB && R1 && p.r1_lds) {
        // R1-resident HBM-slab template.  Crop: the gray scratch of a resized box is the whole LDS (the constants above are staged again behind it),
        // the patch goes to the slab.  Then stripes of stripe_k cell columns: gradient of the stripe's pixel columns (+ 2 / + 1 of halo) into LDS,
        // histogram of its cells into the resident R1 -- the additions of a cell are those of phase_hist over a full plane, in the same order.
        float* lds = smem_base();
        const int total = (int)(MOT_LDS_LIMIT / sizeof(float));
        const int nsrc = (box.b - box.t + 1) * (box.r - box.l + 1);
        const bool in_lds = nsrc <= total;
        __syncthreads();                                               // the constants staged above are not read before they are staged again
        phase_crop(p, l.frame, patch, box, r.A, reinterpret_cast<uint8_t*>(lds), tid, nt, total * 4, in_lds ? lds : r.B, in_lds ? total : p.lds_floats - p.offB, l.dbg);
        __syncthreads();
        DBG_STAMP(1);
        for (int i = tid; i < 2048; i += nt) reinterpret_cast<uint32_t*>(r.tab)[i] = reinterpret_cast<const uint32_t*>(p.sse_tab)[i];
        for (int i = tid; i < p.hb; i += nt) r.twr[i] = p.tw_r[i];
        for (int i = tid; i < p.wb; i += nt) r.twc[i] = p.tw_c[i];
        __syncthreads();
        DBG_STAMP(2);
        float* R1l = lds + p.offR1c; float* X = lds + p.offX;
        const int k = p.stripe_k, w0 = p.wb * 4;
        for (int cx0 = 0; cx0 < p.wb; cx0 += k) {
            const int cx1 = min(cx0 + k, p.wb);
            const int xb = max(0, 4 * cx0 - 2), xe = min(w0 - 1, 4 * cx1 + 1);
            float* Mq = X - xb * p.ldp;                                // virtual bases: column x of the plane is column x - xb of the stripe
            uint8_t* bins = reinterpret_cast<uint8_t*>(X + (4 * k + 4) * p.ldp) - xb * p.ldp;
            phase_gradmag<2>(p, r.A, Mq, bins, r.tab, tid, nt, xb, xe - xb + 1);
            __syncthreads();
            if (cx0 == 0) DBG_STAMP(10);
            phase_hist<0, true>(p, Mq, bins, R1l, nullptr, tid, nt, cx0 * p.hb, cx1 * p.hb);
            __syncthreads();
            if (cx0 == 0) DBG_STAMP(11);
        }
        DBG_STAMP(3);
        phase_energy<false>(p, R1l, r.E, tid, nt);
        __syncthreads();
        phase_norm(p, r.E, r.N, tid, nt);
        __syncthreads();
        DBG_STAMP(4);
        return;
    }
    // byte staging area of the crop: region B, or the LDS staging area of an HBM-slab template
    if (stage) {
        // gray scratch of a resized crop: the LDS staging area, or (source boxes beyond 175 x 175 at 148 px) regions B..T of the slab
        const bbox_t sb = box; const int nsrc = (sb.b - sb.t + 1) * (sb.r - sb.l + 1);
        const bool in_lds = nsrc <= p.stage_floats;
        phase_crop(p, l.frame, patch, box, r.A, reinterpret_cast<uint8_t*>(stage), tid, nt, p.stage_floats * 4, in_lds ? stage : r.B, in_lds ? p.stage_floats : p.lds_floats - p.offB);
    }
    else if (ABL(1)) phase_crop(p, l.frame, patch, box, r.A, reinterpret_cast<uint8_t*>(r.B), tid, nt, 1 << 30, r.B, p.offC - p.offB);
    __syncthreads();
    DBG_STAMP(1);
    float* Mq = r.B; uint8_t* bins = reinterpret_cast<uint8_t*>(r.B + p.cols * p.ldp);
    if (ABL(2)) phase_gradmag(p, r.A, Mq, bins, r.tab, tid, nt);
    __syncthreads();
    DBG_STAMP(2);
    if (ABL(3)) phase_hist<SLAB ? 4 : 0, LDSR1, HB>(p, Mq, bins, r.A, stage, tid, nt);       // R1 overlays the patch
    __syncthreads();
    DBG_STAMP(3);
    if (ABL(4)) phase_energy<SLAB>(p, r.A, r.E, tid, nt);
    __syncthreads();
    if (ABL(4)) phase_norm(p, r.E, r.N, tid, nt);
    __syncthreads();
    DBG_STAMP(4);
}

// one half of the channels -> windowed features -> spectrum in region B (overlays Mq / bins, then itself)
template <int HALF, bool SLAB, bool R1 = false, bool GEN = true, bool OOL = false, bool INPL = false>   // OOL: the direct / staged transforms stay out of line
__device__ void half_spectrum(const KcfPool& p, const KcfLaunch& l, int item, const Regions& r, int tid, int nt, bool spectrum, float* stage = nullptr,
                              float* out_override = nullptr)
{
    float* fo = l.feat_out ? l.feat_out + (size_t)item * 32 * p.nb : nullptr;
    if (SLAB && R1 && p.r1_lds) {                                      // R1-resident: a few planes at a time, R1 -> features -> spectra (slab region B, or the caller's buffer)
        if (!r1_spectrum_dispatch(p, r.N, HALF ? MOT_HALF0 : 0, HALF ? MOT_NCHAN : MOT_HALF0, out_override ? out_override : r.B, fo, l.feat_windowed, tid, nt, l.dbg)) __builtin_trap();
        DBG_STAMP(8 + HALF);
        return;
    }
    // HBM-slab templates with MFMA tables: the half's feature planes go straight into the LDS staging area and both DFT
    // passes run from there (no slab round trip of the planes, no row-spectrum buffer)
    const bool fused = SLAB && spectrum && stage && p.mf && p.stage_floats >= MOT_HALF0 * p.wb * 2 * p.fh + MOT_MF_CW_FLOATS;
    if (ABL(5)) phase_channels<HALF, SLAB>(p, r.A, r.N, fused ? stage : r.B, fo, l.feat_windowed, tid, nt);
    __syncthreads();
    DBG_STAMP(8 + HALF);
    if (fused) {
        // 37 x 37 planes (148-px templates, BASELINE configs[4]): the straight-line instance; other HBM-slab sizes: the generic chain
        if (((p.wb + 15) >> 4) == 3 && ((p.hb + 3) >> 2) == 10) dft2_mfma_fixed<3, 10>(p, (int)(stage - smem_base()), r.B, HALF ? (MOT_NCHAN - MOT_HALF0) : MOT_HALF0, tid, nt);
        else dft2_mfma(p, (int)(stage - smem_base()), (int)(stage - smem_base()) + MOT_HALF0 * p.wb * 2 * p.fh, r.B, HALF ? (MOT_NCHAN - MOT_HALF0) : MOT_HALF0, tid, nt);
        __syncthreads();
    }
    else if (spectrum) {
        // the single-pool kernels with the R1-resident pipeline (kMode 2) keep this, for them dead, path out of line: inlined it costs the shared
        // phases registers (148 px: 586 -> 537 k updates/s); everywhere else -- also in the size-class kernels that mix R1-resident and larger
        // templates (kMode 4) -- inlining it is what pays (168 / 200 px: 0.48 / 0.77 -> 0.37 / 0.55 ms per frame)
        if (!ABL(6)) { __syncthreads(); }
        else if (OOL) fft_forward_ool<SLAB, GEN, INPL>(p, r.T, r.B, r.twr, r.twc, HALF ? (MOT_NCHAN - MOT_HALF0) : MOT_HALF0, tid, nt, stage);
        else fft_forward<SLAB, GEN, INPL>(p, r.T, r.B, r.twr, r.twc, HALF ? (MOT_NCHAN - MOT_HALF0) : MOT_HALF0, tid, nt, stage);
    }
}

// kMode 7: the pool is EXACTLY 80 x 80 px (the reference's and BASELINE's template): its geometry enters the kernel as compile-time constants -- the
// copy below is scalarised, so every loop bound, stride and divisor derived from these fields folds (the phases are all inlined into these kernels).
// Same arithmetic on the same values.  Predict and feature kernels only (launch_kcf_update).
__device__ __forceinline__ FastDiv fastdiv_const(uint32_t d) { FastDiv f; f.d = d; f.m = (d <= 1) ? 0u : (uint32_t)(0xFFFFFFFFull / d + 1ull); return f; }
template <int kMode>
__device__ __forceinline__ KcfPool pool_view(const KcfPool& in)
{
    KcfPool q = in;
    if (kMode == 7) {
        q.rows = 80; q.cols = 80; q.hb = 20; q.wb = 20; q.fh = 11; q.nb = 400; q.nbins = 220; q.ldp = 84; q.ng = 20;
        q.d_rows = fastdiv_const(80); q.d_cols = fastdiv_const(80); q.d_hb = fastdiv_const(20); q.d_fh = fastdiv_const(11);
        q.d_nbins = fastdiv_const(220); q.d_nb = fastdiv_const(400); q.d_ng = fastdiv_const(20);
        q.fft20 = 1; q.use_lds = 1; q.dft_inplace = 0; q.r1_lds = 0; q.szC = 0; q.stage_floats = 0; q.stage_G = 0; q.mf = 0;
    }
    return q;
}

template <int kMode, bool kStagger = false, bool kHB = (MOT_HIST_BATCH != 0)>   // kMode 0: HBM slab, 1: LDS with the 20 x 20 register FFT only (80 px: the headline kernels), 2: HBM slab with the R1-resident pipeline compiled in,
                                             // 3: LDS with the direct transforms compiled in as well (size-class launches), 5: as 3 plus their in-place form (single pool), 4: as 2 for size-class launches,
                                             // 7: as 1 for the pool of exactly 80 x 80 px, geometry folded into the code (pool_view) (the staged transforms inline: larger classes run them)
                                             // (kernels of their own: the other modes keep their code and registers)
__device__ __forceinline__ void kcf_predict_body(const KcfPool& pool_in, const KcfLaunch& l, const int item, float* smem)
{
    constexpr bool kLds = (kMode & 1) != 0;
    const KcfPool p = pool_view<kMode>(pool_in);                       // kMode 7: geometry as constants (see pool_view)
    const KcfPool& pc = p;
    float* base = kLds ? smem : p.gscratch + (size_t)(item + l.slab_base) * (l.slab_stride ? l.slab_stride : p.lds_floats);
    const bool r1m = (kMode == 2 || kMode == 4) && p.r1_lds;                                // R1-resident: LDS = [R1 | region C | work area], the crop's scratch is all of it
    float* stage = r1m ? smem : ((!kLds && p.stage_floats > 0) ? smem + p.szC : nullptr);
    const Regions r = carve(p, base, r1m ? smem + MOT_NORI * 64 * ((p.nb + 63) >> 6) : ((!kLds && p.szC > 0) ? smem : nullptr), r1m);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int slot = l.slots[item];
    if (l.dbg && threadIdx.x == 0 && item < 4096) l.dbg[32 + 3 * item] = wall_clock64();
    const bbox_t pos = p.pos[slot];                                    // kcf_t::pos == tracker_info.bbox (td.cpp:351-354)
    // Deferred blend: tracker_update of the PREVIOUS frame (kcf.cpp:441-453: kf, alpha, model lerp) for this track, from the spectrum
    // its adopted detection box got in that frame's feature launch -- the same arithmetic, in the same order, as the blend launch it
    // replaces (kcf_update_body, blend-only path); pure HBM streaming that overlaps the feature phases of the neighbouring workgroups.
    const float2* xm = p.xm + (size_t)slot * MOT_NCHAN * p.nbins;
    const bool pre = kLds && p.nbins <= nt;                            // HBM-slab templates never prefetch (frees the registers for the DFTs)
    const int bpre = min(tid, p.nbins - 1);
    float2 xmr[MOT_HALF0]; float alr = 0.f;                            // 16 planes at a time (register budget for 2 workgroups / CU)
    bool have_model = false;
    const int dj = l.pend_det ? l.pend_det[slot] : -1;
    int first_seen = -1;                                               // (MOT_TRACE) the first_update flag the blend read
    // Workgroups i and i + 256 of a launch share a compute unit (dispatch order on the 8 x 32 CUs): the second of the pair does its
    // blend AFTER the feature phases, so the model streaming of one overlaps the arithmetic of the other instead of all 512 resident
    // workgroups hitting HBM in the launch's first microseconds
    const bool late = kStagger && pre && dj >= 0 && ((blockIdx.x >> 8) & 1);
    auto blend = [&]() {
        const int tot = MOT_NCHAN * p.nbins;
        const float2* dspec = l.pend_spec + (size_t)dj * tot;
        float2* xmw = p.xm + (size_t)slot * tot;
        const int first = p.first_update[slot];
        first_seen = first;
        const float factor = first ? 1.0f : p.eta, keep = 1.0f - factor;       // kcf.cpp:443
        int i_lo = 0;                                               // first element the streaming blend below still has to do
        if (pre) {
            // one thread per bin: the first 16 planes are blended in registers and STAY there for this frame's correlation
            // (no re-read of what was just written); the waves that own no bins stream planes 16..30 at the same time and hand
            // |S|^2 of those planes over in LDS, so kf still adds up in channel order (kcf.cpp:269-304) and every HBM load of the
            // blend is in flight at once -- one memory latency instead of three
            const int wbins = (p.nbins + 63) & ~63, ns = nt - wbins;
            const int half_lo = MOT_HALF0 * p.nbins, nhi = tot - half_lo;
            const bool split = ns >= 128 && nhi <= MOT_HALF0 * ns;
            float* sq = r.B;                                            // free here: before the crop, or between the norm and the channels
            float kf = 0.f, old = 0.f;
            // both roles run the same 16-element code on the same registers: element j of a bin thread is plane j of its bin, element
            // j of a streaming thread is the j-th of its strided share of planes 16..30
            const bool binthr = tid < p.nbins, strthr = split && tid >= wbins;
            const int ebase = binthr ? tid : half_lo + (tid - wbins), estride = binthr ? p.nbins : ns;
            const int elim = binthr ? half_lo : (strthr ? tot : 0);
            float2 sp[MOT_HALF0];
#pragma unroll
            for (int j = 0; j < MOT_HALF0; j++) {
                const int i = min(ebase + j * estride, tot - 1);
                sp[j] = dspec[i]; xmr[j] = first ? make_float2(0.f, 0.f) : xmw[i];
            }
            if (binthr) old = first ? 0.0f : p.alpha[(size_t)slot * p.nbins + tid];
#pragma unroll
            for (int j = 0; j < MOT_HALF0; j++) {
                const int i = ebase + j * estride;
                const float2 a = sp[j]; const float e2 = a.x * a.x + a.y * a.y;
                float2 m = xmr[j]; m.x = keep * m.x + factor * a.x; m.y = keep * m.y + factor * a.y; xmr[j] = m;
                if (i < elim) xmw[i] = m;
                if (binthr) kf = e2 + kf; else if (i < elim) sq[i - half_lo] = e2;
            }
            if (split) __syncthreads();
            if (tid < p.nbins) {
                if (split) {
#pragma unroll
                    for (int ch = MOT_HALF0; ch < MOT_NCHAN; ch++) kf = sq[(ch - MOT_HALF0) * p.nbins + tid] + kf;
                } else {
#pragma unroll
                    for (int ch = MOT_HALF0; ch < MOT_NCHAN; ch++) { const float2 a = dspec[ch * p.nbins + tid]; kf = (a.x * a.x + a.y * a.y) + kf; }
                }
                const float kq = kf * p.norm;
                const float a = p.yf_re[tid] / (kq + p.lambda);
                alr = keep * old + factor * a;
                p.alpha[(size_t)slot * p.nbins + tid] = alr;
            }
            i_lo = split ? tot : half_lo; have_model = true;
        } else {
            for (int b = tid; b < p.nbins; b += nt) {               // kcf_linear_correlation_kf + kcf_update_alpha (kcf.cpp:269-304, 364-378)
                float kf = 0.f;
                for (int ch = 0; ch < MOT_NCHAN; ch++) { const float2 a = dspec[ch * p.nbins + b]; kf = (a.x * a.x + a.y * a.y) + kf; }
                const float kq = kf * p.norm;
                const float a = p.yf_re[b] / (kq + p.lambda);
                const float old = first ? 0.0f : p.alpha[(size_t)slot * p.nbins + b];
                p.alpha[(size_t)slot * p.nbins + b] = keep * old + factor * a;
            }
        }
        for (int i0 = i_lo; i0 < tot; i0 += 8 * nt) {               // kcf_update_xf (kcf.cpp:380-395): 16 loads in flight per thread
            float2 a8[8], m8[8];
#pragma unroll
            for (int j = 0; j < 8; j++) { const int i = min(i0 + tid + j * nt, tot - 1); a8[j] = dspec[i]; m8[j] = first ? make_float2(0.f, 0.f) : xmw[i]; }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int i = i0 + tid + j * nt;
                if (i < tot) { float2 m = m8[j]; m.x = keep * m.x + factor * a8[j].x; m.y = keep * m.y + factor * a8[j].y; xmw[i] = m; }
            }
        }
        __syncthreads();                                           // the blended model is visible to the whole workgroup (loads below)
        if (tid == 0) { l.pend_det[slot] = -1; p.first_update[slot] = 0; }
    };
    if (dj >= 0 && !late && ABL(0)) blend();
    // the model does not depend on this frame: issue its loads now (31 independent 8-byte loads per bin thread),
    // they land while the features are computed
    if (pre && !have_model && !late && ABL(10)) {
#pragma unroll
        for (int ch = 0; ch < MOT_HALF0; ch++) xmr[ch] = xm[ch * p.nbins + bpre];
        alr = p.alpha[(size_t)slot * p.nbins + bpre];
    }
    features_prepare<!kLds, true, (kMode == 2 || kMode == 4), kHB>(pc, l, item, pos, r, tid, nt, stage);
    if (late && ABL(0)) blend();
    // kcf_linear_correlation_zf (kcf.cpp:306-362): zf = sum_c xf_c * conj(xm_c), then * alpha * norm; accumulated over the
    // two channel halves in registers (one thread per bin)
    const float2* S = reinterpret_cast<const float2*>(r.B);
    // partial sums of the first half wait here for the second: zf, or -- R1-resident templates, whose transform tiles run over zf -- slab region T
    float2* zpark = r1m ? reinterpret_cast<float2*>(r.T) : r.zf;
    float zr = 0.f, zi = 0.f;
    half_spectrum<0, !kLds, (kMode == 2 || kMode == 4), (kMode != 1 && kMode != 7), kMode == 2, kMode == 5>(p, l, item, r, tid, nt, true, stage);
    DBG_STAMP(5);
    if (pre) {
        if (tid < p.nbins && ABL(7)) {
#pragma unroll
            for (int ch = 0; ch < MOT_HALF0; ch++) { const float2 a = S[ch * p.nbins + tid]; const float2 m = xmr[ch]; zr += a.x * m.x + a.y * m.y; zi += a.y * m.x - a.x * m.y; }
        }
        if (ABL(10)) {
#pragma unroll
        for (int ch = MOT_HALF0; ch < MOT_NCHAN; ch++) xmr[ch - MOT_HALF0] = xm[ch * p.nbins + bpre];   // second half: lands during its features
        }
    } else if (r1m && p.nbins <= nt) {
        // R1-resident HBM-slab templates with one bin per thread: the partial sums stay in registers
        if (tid < p.nbins) {
            for (int ch = 0; ch < MOT_HALF0; ch++) { const float2 a = S[ch * p.nbins + tid]; const float2 m = xm[ch * p.nbins + tid]; zr += a.x * m.x + a.y * m.y; zi += a.y * m.x - a.x * m.y; }
        }
    } else {
        for (int b = tid; b < p.nbins; b += nt) {      // generic sizes: partial sums parked in zf
            float pr = 0.f, pi = 0.f;
            for (int ch = 0; ch < MOT_HALF0; ch++) { const float2 a = S[ch * p.nbins + b]; const float2 m = xm[ch * p.nbins + b]; pr += a.x * m.x + a.y * m.y; pi += a.y * m.x - a.x * m.y; }
            zpark[b] = make_float2(pr, pi);
        }
    }
    __syncthreads();
    half_spectrum<1, !kLds, (kMode == 2 || kMode == 4), (kMode != 1 && kMode != 7), kMode == 2, kMode == 5>(p, l, item, r, tid, nt, true, stage);
    DBG_STAMP(6);
    if (pre) {
        if (tid < p.nbins) {
            if (ABL(7)) {
#pragma unroll
            for (int ch = MOT_HALF0; ch < MOT_NCHAN; ch++) { const float2 a = S[(ch - MOT_HALF0) * p.nbins + tid]; const float2 m = xmr[ch - MOT_HALF0]; zr += a.x * m.x + a.y * m.y; zi += a.y * m.x - a.x * m.y; }
            }
            r.zf[tid] = make_float2((zr * alr) * p.norm, (zi * alr) * p.norm);
        }
    } else if (r1m && p.nbins <= nt) {
        if (tid < p.nbins) {
            for (int ch = MOT_HALF0; ch < MOT_NCHAN; ch++) { const float2 a = S[(ch - MOT_HALF0) * p.nbins + tid]; const float2 m = xm[ch * p.nbins + tid]; zr += a.x * m.x + a.y * m.y; zi += a.y * m.x - a.x * m.y; }
            const float al = p.alpha[(size_t)slot * p.nbins + tid];
            r.zf[tid] = make_float2((zr * al) * p.norm, (zi * al) * p.norm);
        }
    } else {
        for (int b = tid; b < p.nbins; b += nt) {
            float pr = zpark[b].x, pi = zpark[b].y;
            for (int ch = MOT_HALF0; ch < MOT_NCHAN; ch++) { const float2 a = S[(ch - MOT_HALF0) * p.nbins + b]; const float2 m = xm[ch * p.nbins + b]; pr += a.x * m.x + a.y * m.y; pi += a.y * m.x - a.x * m.y; }
            const float al = p.alpha[(size_t)slot * p.nbins + b];
            r.zf[b] = make_float2((pr * al) * p.norm, (pi * al) * p.norm);
        }
    }
    __syncthreads();
    DBG_STAMP(17);
    if (ABL(8)) fft_inverse_plane(p, r.zf, r.tmp, r.resp, r.twr, r.twc, tid, nt);
    DBG_STAMP(18);
    if (ABL(9)) for (int i = tid; i < p.nb; i += nt) p.response[(size_t)slot * p.nb + i] = r.resp[i];
    const int best = ABL(9) ? block_argmax_first(r.resp, p.nb, r.red_v, r.red_i, tid, nt) : 0;
    DBG_STAMP(19);
    if (tid == 0) {
        int vd = 1, hd = 1;
        if (best >= 0) { hd = best / p.hb + 1; vd = best - (hd - 1) * p.hb + 1; }
        if (vd > p.hb / 2) vd -= p.hb;                                 // kcf.cpp:420-421
        if (hd > p.wb / 2) hd -= p.wb;
        const float2 sc = p.scale[slot];                               // (horiz, vert)
        bbox_t np = pos;
        {
#pragma clang fp contract(off)
            np.t = (int)((float)pos.t + (float)(MOT_CELL * (vd - 1)) * sc.y); // kcf.cpp:424-427
            np.b = (int)((float)pos.b + (float)(MOT_CELL * (vd - 1)) * sc.y);
            np.l = (int)((float)pos.l + (float)(MOT_CELL * (hd - 1)) * sc.x);
            np.r = (int)((float)pos.r + (float)(MOT_CELL * (hd - 1)) * sc.x);
        }
        p.pos[slot] = np;
        bbox_t o = np;                                                 // *pbox = pkcf->pos (kcf.cpp:438)
        if (l.clamp) {                                                 // td.cpp:378-381
            o.l = min(max(o.l, 0), MOT_FRAME_W - 1); o.r = min(max(o.r, 0), MOT_FRAME_W - 1);
            o.t = min(max(o.t, 0), MOT_FRAME_H - 1); o.b = min(max(o.b, 0), MOT_FRAME_H - 1);
        }
        if (l.boxes_out) l.boxes_out[item] = o;
        if (l.trace) {                                                 // (debug, MOT_TRACE) what this workgroup saw and decided
            int* tr = l.trace + ((size_t)(l.trace_frame & 15) * l.trace_cap + slot) * 8;
            tr[0] = l.trace_frame; tr[1] = pos.l; tr[2] = pos.t; tr[3] = (best & 0xFFFF) | (item << 16); tr[4] = best >= 0 ? __float_as_int(r.resp[best]) : 0;
            tr[5] = dj; tr[6] = first_seen; tr[7] = (np.l & 0xFFFF) | (np.t << 16);
        }
        if (l.dbg && item < 4096) {
            unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));   // cu / se / xcc of this workgroup
            unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            l.dbg[32 + 3 * item + 1] = wall_clock64(); l.dbg[32 + 3 * item + 2] = ((long long)xcc << 32) | hw;
        }
    }
    DBG_STAMP(7);
}

template <int kMode>
__global__ void __launch_bounds__((kMode & 1) ? MOT_KCF_THREADS : MOT_KCF_THREADS_SLAB, ((kMode & 1) || MOT_KCF_THREADS_SLAB > 512) ? 4 : 2) kcf_predict_kernel(const KcfPool p, const KcfLaunch l, int n)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int item = blockIdx.x;
    if (item >= n) return;
    if (l.count && item >= *l.count) return;
    kcf_predict_body<kMode, (kMode & 1) != 0>(p, l, item, smem);
}
// size classes (device loop with per-track template sizes, kcf.cpp:148-152): the workgroup's pool descriptor comes from a device
// table, indexed by the class of its track
template <int kMode>
__global__ void __launch_bounds__((kMode & 1) ? MOT_KCF_THREADS : MOT_KCF_THREADS_SLAB, ((kMode & 1) || MOT_KCF_THREADS_SLAB > 512) ? 4 : 2) kcf_predict_multi_kernel(const KcfLaunch l, int n)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int item = blockIdx.x;
    if (item >= n) return;
    if (l.count && item >= *l.count) return;
    const KcfPool p = l.pools[l.cls[item]];
    kcf_predict_body<kMode>(p, l, item, smem);
}

template <int kMode, bool kView = false>
__device__ __forceinline__ void kcf_update_body(const KcfPool& pool_in, const KcfLaunch& l, const int item, float* smem)
{
    constexpr bool kLds = (kMode & 1) != 0;
    const KcfPool p = kView ? pool_view<kMode>(pool_in) : pool_in;     // kView: see kcf_update_sparse_run
    const KcfPool& pc = p;
    float* base = kLds ? smem : p.gscratch + (size_t)(item + l.slab_base) * (l.slab_stride ? l.slab_stride : p.lds_floats);
    const bool r1m = (kMode == 2 || kMode == 4) && p.r1_lds;                                // R1-resident: LDS = [R1 | region C | work area], the crop's scratch is all of it
    float* stage = r1m ? smem : ((!kLds && p.stage_floats > 0) ? smem + p.szC : nullptr);
    const Regions r = carve(p, base, r1m ? smem + MOT_NORI * 64 * ((p.nb + 63) >> 6) : ((!kLds && p.szC > 0) ? smem : nullptr), r1m);
    const int tid = threadIdx.x, nt = blockDim.x;
    const bool feat_only = l.spec_out != nullptr;                      // launch-uniform
    const int slot = feat_only ? 0 : l.slots[item];
    const bbox_t box = l.boxes_in[item];
    const int first = feat_only ? 1 : p.first_update[slot];
    float2* xm = p.xm + (size_t)slot * MOT_NCHAN * p.nbins;
    const int tot = MOT_NCHAN * p.nbins;
    // split mode: this track's detection already has its spectra in HBM (workgroup-uniform)
    const int dj = (!feat_only && l.det_index) ? l.det_index[item] : -1;
    const float2* dspec = dj >= 0 ? l.det_spec + (size_t)dj * tot : nullptr;
    // old model values do not depend on this frame: load them now (<= 16 independent loads per thread)
    const bool pre = kLds && tot <= 16 * nt;
    float2 xold[16];
    if (pre && !dspec) {
#pragma unroll
        for (int j = 0; j < 16; j++) xold[j] = first ? make_float2(0.f, 0.f) : xm[min(tid + j * nt, tot - 1)];
    }
    if (!dspec) features_prepare<!kLds, true, (kMode == 2 || kMode == 4)>(pc, l, item, box, r, tid, nt, stage);
    const float factor = first ? 1.0f : p.eta;                         // kcf.cpp:443
    const float keep = 1.0f - factor;
    const float2* S = reinterpret_cast<const float2*>(r.B);
    float2* tpark = r1m ? reinterpret_cast<float2*>(r.T) : r.tmp;     // see kcf_predict_body
    const int tot0 = MOT_HALF0 * p.nbins;
    float kf = 0.f;                                                    // kcf_linear_correlation_kf (kcf.cpp:269-304), one bin per thread
    // per half: kf partial sums, then kcf_update_xf (kcf.cpp:380-395) for the channels of this half
#define UPDATE_HALF(C0, C1)                                                                                      \
    do {                                                                                                          \
        if (dspec) S = dspec + (size_t)(C0) * p.nbins;                                                            \
        if (feat_only) {                                                                                          \
            float2* so = l.spec_out + (size_t)item * tot + (size_t)(C0) * p.nbins;                                \
            for (int i = tid; i < ((C1) - (C0)) * p.nbins; i += nt) so[i] = S[i];                                 \
            break;                                                                                                \
        }                                                                                                         \
        if (p.nbins <= nt) {                                                                                      \
            if (tid < p.nbins) for (int ch = (C0); ch < (C1); ch++) { const float2 a = S[(ch - (C0)) * p.nbins + tid]; kf = (a.x * a.x + a.y * a.y) + kf; } \
        } else {                                                                                                  \
            for (int b = tid; b < p.nbins; b += nt) { float q = ((C0) == 0) ? 0.f : tpark[b].x;                   \
                for (int ch = (C0); ch < (C1); ch++) { const float2 a = S[(ch - (C0)) * p.nbins + b]; q = (a.x * a.x + a.y * a.y) + q; } \
                tpark[b].x = q; }                                                                                 \
        }                                                                                                         \
        if (pre) {                                                                                                \
            _Pragma("unroll") for (int j = 0; j < 16; j++) {                                                      \
                const int i = tid + j * nt;                                                                       \
                if (i >= (C0) * p.nbins && i < (C1) * p.nbins) {                                                  \
                    const float2 a = S[i - (C0) * p.nbins]; float2 m = xold[j];                                   \
                    m.x = keep * m.x + factor * a.x; m.y = keep * m.y + factor * a.y; xm[i] = m; }                \
            }                                                                                                     \
        } else {                                                                                                  \
            for (int i = (C0) * p.nbins + tid; i < (C1) * p.nbins; i += nt) {                                     \
                const float2 a = S[i - (C0) * p.nbins]; float2 m = first ? make_float2(0.f, 0.f) : xm[i];         \
                m.x = keep * m.x + factor * a.x; m.y = keep * m.y + factor * a.y; xm[i] = m; }                    \
        }                                                                                                         \
    } while (0)
    if (dspec && pre) {
        // blend-only fast path: everything streams from HBM, all loads of a thread in flight at once (same arithmetic
        // and order as the macro below: kf over ch = 0..30, then kcf_update_xf)
        const int bq = min(tid, p.nbins - 1);
#pragma unroll
        for (int ch = 0; ch < MOT_NCHAN; ch++) { const float2 a = dspec[ch * p.nbins + bq]; kf = (a.x * a.x + a.y * a.y) + kf; }
#pragma unroll
        for (int j0 = 0; j0 < 16; j0 += 8) {                          // 16 loads in flight, twice
            float2 a8[8], m8[8];
#pragma unroll
            for (int j = 0; j < 8; j++) { const int i = min(tid + (j0 + j) * nt, tot - 1); a8[j] = dspec[i]; m8[j] = first ? make_float2(0.f, 0.f) : xm[i]; }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int i = tid + (j0 + j) * nt;
                if (i < tot) { float2 m = m8[j]; m.x = keep * m.x + factor * a8[j].x; m.y = keep * m.y + factor * a8[j].y; xm[i] = m; }
            }
        }
    } else {
    if (!dspec) half_spectrum<0, !kLds, (kMode == 2 || kMode == 4), (kMode != 1 && kMode != 7), kMode == 2, kMode == 5>(p, l, item, r, tid, nt, true, stage);
    DBG_STAMP(5);
    UPDATE_HALF(0, MOT_HALF0);
    if (!dspec) {
        __syncthreads();
        half_spectrum<1, !kLds, (kMode == 2 || kMode == 4), (kMode != 1 && kMode != 7), kMode == 2, kMode == 5>(p, l, item, r, tid, nt, true, stage);
    }
    DBG_STAMP(6);
    UPDATE_HALF(MOT_HALF0, MOT_NCHAN);
    }
#undef UPDATE_HALF
    (void)tot0;
    if (feat_only) return;
    // kcf_update_alpha (kcf.cpp:364-378)
    for (int b = tid; b < p.nbins; b += nt) {
        float kq = (p.nbins <= nt) ? kf : tpark[b].x;
        kq = kq * p.norm;
        const float a = p.yf_re[b] / (kq + p.lambda);
        const float old = first ? 0.0f : p.alpha[(size_t)slot * p.nbins + b];
        p.alpha[(size_t)slot * p.nbins + b] = keep * old + factor * a;
    }
    if (tid == 0 && l.trace) {                                         // (debug, MOT_TRACE)
        int* tr = l.trace + ((size_t)(16 + (l.trace_frame & 15)) * l.trace_cap + slot) * 8;
        tr[0] = l.trace_frame; tr[1] = box.l; tr[2] = box.t; tr[3] = dj; tr[4] = first; tr[5] = slot; tr[6] = item; tr[7] = 0;
    }
    if (tid == 0) {                                                    // kcf.cpp:470-472
        p.pos[slot] = box;
        p.scale[slot] = make_float2(((float)(box.r - box.l + 1)) / ((float)p.cols), ((float)(box.b - box.t + 1)) / ((float)p.rows));
        p.first_update[slot] = 0;
    }
    DBG_STAMP(7);
}

template <int kMode>
__global__ void __launch_bounds__((kMode & 1) ? MOT_KCF_THREADS : MOT_KCF_THREADS_SLAB, ((kMode & 1) || MOT_KCF_THREADS_SLAB > 512) ? 4 : 2) kcf_update_kernel(const KcfPool p, const KcfLaunch l, int n)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int item = blockIdx.x;
    if (item >= n) return;
    if (l.count && item >= *l.count) return;
    kcf_update_body<kMode, true>(p, l, item, smem);
}
// Few items, count known on the device only (KcfLaunch::grid_stride): a small grid loops over them.  The device loop launches this every frame for
// a list that is usually EMPTY (tracks that keep their predicted box), so the count test must come before anything else.  Out-of-line callees take
// the pool descriptor by reference, which makes the compiler copy the by-value kernel arguments to private memory in the kernel's ENTRY block (19
// scratch stores per lane: 20 MB and 4 us per empty launch when the body was inlined here); the body is therefore a function of its own that
// receives the descriptors BY VALUE -- the copy happens at the call, behind the test.
// kView stays false here (kMode 7).  With the folded copy of the descriptor (pool_view) inside THIS function hipcc (ROCm 7.2) emits a wrong
// program: behind the divergent loop of the second channel half's transform the exit block holds seven v_mov_b64 -- the undo of a register
// rotation the allocator split around the loop -- IN FRONT OF the EXEC restore (the loop's own restore was folded into the enclosing region's,
// -amdgpu-remove-redundant-endcf), so they run for no lane and the old-model registers xold[7], [9], [11], [13] keep foreign values: a wrong
// model in exactly those slots of every thread (tests/test_gpu_devloop.py::test_device_loop_vs_oracle[0-48-80-9], frame 8, came out two cells
// off).  tools/isa_exec0_scan.py finds the pattern in code objects; over the 12,000 loop exits / joins of the shipped library this function
// was the only hit, and with -mllvm -amdgpu-remove-redundant-endcf=0 it is gone and the results equal the general kernels' (DESIGN.md
// section 6).  The instantiation is compiled only for that demonstration (-DMOT_KCF_SPARSE_VIEW=1, `make endcf`).
template <int kMode, bool kView = false>
__device__ __attribute__((noinline)) void kcf_update_sparse_run(KcfPool p, KcfLaunch l, int cnt)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int item = blockIdx.x; item < cnt; item += gridDim.x) { kcf_update_body<kMode, kView>(p, l, item, smem); __syncthreads(); }
}
#ifndef MOT_KCF_SPARSE_VIEW
#define MOT_KCF_SPARSE_VIEW 0
#endif
#if MOT_KCF_SPARSE_VIEW
// (demonstration builds only, MOT_KCF_K80 bit 3) the folded descriptor inside the out-of-line body as well: see above
template <int kMode>
__global__ void __launch_bounds__((kMode & 1) ? MOT_KCF_THREADS : MOT_KCF_THREADS_SLAB, ((kMode & 1) || MOT_KCF_THREADS_SLAB > 512) ? 4 : 2) kcf_update_sparse_view_kernel(const KcfPool p, const KcfLaunch l, int n)
{
    const int cnt = min(n, l.count ? *l.count : n);
    if ((int)blockIdx.x >= cnt) return;
    kcf_update_sparse_run<kMode, true>(p, l, cnt);
}
#endif
template <int kMode>
__global__ void __launch_bounds__((kMode & 1) ? MOT_KCF_THREADS : MOT_KCF_THREADS_SLAB, ((kMode & 1) || MOT_KCF_THREADS_SLAB > 512) ? 4 : 2) kcf_update_sparse_kernel(const KcfPool p, const KcfLaunch l, int n)
{
    const int cnt = min(n, l.count ? *l.count : n);
    if ((int)blockIdx.x >= cnt) return;
    kcf_update_sparse_run<kMode>(p, l, cnt);
}
template <int kMode>
__global__ void __launch_bounds__((kMode & 1) ? MOT_KCF_THREADS : MOT_KCF_THREADS_SLAB, ((kMode & 1) || MOT_KCF_THREADS_SLAB > 512) ? 4 : 2) kcf_update_multi_kernel(const KcfLaunch l, int n)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int item = blockIdx.x;
    if (item >= n) return;
    if (l.count && item >= *l.count) return;
    const KcfPool p = l.pools[l.cls[item]];
    kcf_update_body<kMode>(p, l, item, smem);
}

// Feature-only launch of the split update (device loop): crop -> FHOG -> 31 spectra of every DETECTION box, written to l.spec_out.
// A kernel of its own: inside kcf_update_kernel the model prefetch and the blend paths it never takes cost it 300 spilled VGPRs
// (1 KB of scratch per lane) -- and this launch is the one that shares the chip with the association chain every frame.
template <int kMode, bool kHB = (MOT_HIST_BATCH != 0)>
__device__ __forceinline__ void kcf_features_body(const KcfPool& pool_in, const KcfLaunch& l, const int item, float* smem)
{
    constexpr bool kLds = (kMode & 1) != 0;
    const KcfPool p = pool_view<kMode>(pool_in);                       // kMode 7: geometry as constants (see pool_view)
    const KcfPool& pc = p;
    float* base = kLds ? smem : p.gscratch + (size_t)(item + l.slab_base) * (l.slab_stride ? l.slab_stride : p.lds_floats);
    const bool r1m = (kMode == 2 || kMode == 4) && p.r1_lds;                                // R1-resident: LDS = [R1 | region C | work area], the crop's scratch is all of it
    float* stage = r1m ? smem : ((!kLds && p.stage_floats > 0) ? smem + p.szC : nullptr);
    const Regions r = carve(p, base, r1m ? smem + MOT_NORI * 64 * ((p.nb + 63) >> 6) : ((!kLds && p.szC > 0) ? smem : nullptr), r1m);
    const int tid = threadIdx.x, nt = blockDim.x;
    const bbox_t box = l.boxes_in[item];
    features_prepare<!kLds, true, (kMode == 2 || kMode == 4), kHB>(pc, l, item, box, r, tid, nt, stage);
    const float2* S = reinterpret_cast<const float2*>(r.B);
    float2* so = l.spec_out + (size_t)item * MOT_NCHAN * p.nbins;
    if (r1m) {                                                         // the transforms store straight into the launch's spectrum buffer
        half_spectrum<0, !kLds, (kMode == 2 || kMode == 4), (kMode != 1 && kMode != 7), kMode == 2, kMode == 5>(p, l, item, r, tid, nt, true, stage, reinterpret_cast<float*>(so));
        half_spectrum<1, !kLds, (kMode == 2 || kMode == 4), (kMode != 1 && kMode != 7), kMode == 2, kMode == 5>(p, l, item, r, tid, nt, true, stage, reinterpret_cast<float*>(so + (size_t)MOT_HALF0 * p.nbins));
        return;
    }
    half_spectrum<0, !kLds, (kMode == 2 || kMode == 4), (kMode != 1 && kMode != 7), kMode == 2, kMode == 5>(p, l, item, r, tid, nt, true, stage);
    for (int i = tid; i < MOT_HALF0 * p.nbins; i += nt) so[i] = S[i];
    __syncthreads();
    half_spectrum<1, !kLds, (kMode == 2 || kMode == 4), (kMode != 1 && kMode != 7), kMode == 2, kMode == 5>(p, l, item, r, tid, nt, true, stage);
    so += (size_t)MOT_HALF0 * p.nbins;
    for (int i = tid; i < (MOT_NCHAN - MOT_HALF0) * p.nbins; i += nt) so[i] = S[i];
}

template <int kMode>
__global__ void __launch_bounds__((kMode & 1) ? MOT_KCF_THREADS : MOT_KCF_THREADS_SLAB, ((kMode & 1) || MOT_KCF_THREADS_SLAB > 512) ? 4 : 2) kcf_features_kernel(const KcfPool p, const KcfLaunch l, int n)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int item = blockIdx.x;
    if (item >= n) return;
    if (l.count && item >= *l.count) return;
    kcf_features_body<kMode>(p, l, item, smem);
}

// Small frames (predict + feature workgroups fit the chip together): ONE launch carries both -- workgroups [0, n_pred) predict the
// tracks, workgroups [n_pred, n_pred + n_feat) compute the detection spectra of the split update.  No side stream, no event pair, no
// cross-stream wait: at 64 tracks those cost more than the kernels' own work.
template <int kMode, bool kHB = (MOT_HIST_BATCH != 0)>   // kHB: phase_hist's BATCH (compile-time A/B)
__global__ void __launch_bounds__((kMode & 1) ? MOT_KCF_THREADS : MOT_KCF_THREADS_SLAB, ((kMode & 1) || MOT_KCF_THREADS_SLAB > 512) ? 4 : 2) kcf_predict_features_kernel(const KcfPool p, const KcfLaunch lp, const KcfLaunch lf, int n_pred, int n_feat)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x < n_pred) {
        const int item = blockIdx.x;
        if (lp.count && item >= *lp.count) return;
        kcf_predict_body<kMode, false, kHB>(p, lp, item, smem);
    } else {
        const int item = (int)blockIdx.x - n_pred;
        if (item >= n_feat) return;
        kcf_features_body<kMode, kHB>(p, lf, item, smem);
    }
}

template <bool kLds>
__global__ void __launch_bounds__(MOT_KCF_THREADS) kcf_fhog_kernel(const KcfPool p, const KcfLaunch l, int n)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int item = blockIdx.x;
    if (item >= n) return;
    float* base = kLds ? smem : p.gscratch + (size_t)(item + l.slab_base) * (l.slab_stride ? l.slab_stride : p.lds_floats);
    const Regions r = carve(p, base);
    bbox_t box = l.boxes_in ? l.boxes_in[item] : bbox_t{0, 0, p.rows - 1, p.cols - 1, 0, 0.f};
    features_prepare<false, kLds>(p, l, item, box, r, threadIdx.x, blockDim.x);
    half_spectrum<0, false>(p, l, item, r, threadIdx.x, blockDim.x, false);
    __syncthreads();
    half_spectrum<1, false>(p, l, item, r, threadIdx.x, blockDim.x, false);
}

template <bool kLds>
__global__ void __launch_bounds__(MOT_KCF_THREADS) kcf_crop_kernel(const KcfPool p, const KcfLaunch l, int n, float* patch_out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int item = blockIdx.x;
    if (item >= n) return;
    float* base = kLds ? smem : p.gscratch + (size_t)(item + l.slab_base) * (l.slab_stride ? l.slab_stride : p.lds_floats);
    const Regions r = carve(p, base);
    const int tid = threadIdx.x, nt = blockDim.x;
    phase_crop(p, l.frame, nullptr, l.boxes_in[item], r.A, reinterpret_cast<uint8_t*>(r.B), tid, nt);
    __syncthreads();
    const int npx = p.rows * p.cols;
    for (int d = tid; d < npx; d += nt) {
        uint32_t c, rr; p.d_rows.divmod((uint32_t)d, c, rr);
        patch_out[(size_t)item * npx + d] = r.A[c * p.ldp + rr];
    }
}

} // namespace

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
void kcf_pool_layout_r1(KcfPool& p, bool allow);
void kcf_pool_layout(KcfPool& p, bool allow_r1, bool allow_inplace)
{
    p.hb = p.rows / MOT_CELL; p.wb = p.cols / MOT_CELL; p.fh = p.hb / 2 + 1;
    p.nb = p.hb * p.wb; p.nbins = p.wb * p.fh;
    p.ldp = ((p.rows + 3) & ~3) + 4;
    p.ng = (p.rows + 3) / 4;
    p.d_ng.init(p.ng);
    p.d_rows.init(p.rows); p.d_cols.init(p.cols); p.d_hb.init(p.hb); p.d_fh.init(p.fh); p.d_nbins.init(p.nbins); p.d_nb.init(p.nb);
    p.norm = (float)(1.0 / (double)((float)(p.wb * p.hb * MOT_NCHAN)));
    p.eta = 0.05f; p.lambda = 0.0001f;
    const int spec = MOT_HALF0 * p.nbins * 2;                         // floats of 16 spectrum planes / padded feature planes
    auto up4 = [](int v) { return (v + 3) & ~3; };
    const int r1f = MOT_NORI * 64 * ((p.nb + 63) / 64);              // R1, wave-interleaved (the slab's orientation-major form is smaller)
    int szA = p.cols * p.ldp; if (r1f > szA) szA = r1f;               // patch, then R1
    int szB = p.cols * p.ldp + (p.cols * p.ldp + 3) / 4 + 8; if (spec > szB) szB = spec;   // Mq + bins in the padded column layout
    const int szC = 2048 + 2 * p.hb + 2 * p.wb + 4 * p.nbins + p.nb + (p.hb + 1) * (p.wb + 1) + 32;   // E and resp share their nb floats
    const int szT = p.fft20 ? 0 : spec;                               // ping-pong buffer of the generic DFT
    p.offA = 0; p.offB = up4(szA); p.offC = p.offB + up4(szB); p.offT = p.offC + up4(szC);
    p.lds_floats = p.offT + up4(szT);
    p.use_lds = ((size_t)p.lds_floats * sizeof(float) <= MOT_LDS_LIMIT) ? 1 : 0;
    // LDS-resident templates with the direct transforms: in place when a thread can hold all its outputs of a pass (dft2_generic_inplace); region T
    // then leaves the layout.  (The split LDS-resident / HBM slab above is decided WITH region T, so no template size changes its path.)
    p.dft_inplace = 0;
    if (p.use_lds && !p.fft20 && allow_inplace && mot_impl::env().dft_inplace) {
        const int lines = MOT_HALF0 * p.wb, items_r = ((lines + 3) / 4) * p.fh, items_c = MOT_HALF0 * p.wb * ((p.fh + 3) / 4);
        // ... and only where it buys a second workgroup per CU (72 / 76 px: predict launch of 1024 tracks 207 -> 156 us); a third one is out of reach
        // (128 registers per lane), and without an occupancy step the two extra barriers cost a few per cent (64 px: 117 -> 127 us, 88 px: 280 -> 289)
        const size_t with_t = (size_t)p.lds_floats * sizeof(float), without_t = (size_t)p.offT * sizeof(float);
        const bool fits = items_r <= MOT_DFT_INPLACE_ITEMS * MOT_KCF_THREADS && items_c <= MOT_DFT_INPLACE_ITEMS * MOT_KCF_THREADS;
        if (fits && MOT_LDS_LIMIT / with_t < 2 && MOT_LDS_LIMIT / without_t >= 2) { p.dft_inplace = 1; p.lds_floats = p.offT; }
    }
    // HBM-slab templates keep region C (tables, twiddles, the single-plane buffers) in LDS, plus a staging area: the
    // per-thread histogram scratch and G channel planes (features + row spectra) of the generic DFT
    p.szC = 0; p.stage_floats = 0; p.stage_G = 0;
    if (!p.use_lds) {
        const int avail = (int)(MOT_LDS_LIMIT / sizeof(float)) - up4(szC);
        const int hist = up4(MOT_KCF_THREADS_SLAB * MOT_NORI);
        const int plane2 = 2 * p.wb * 2 * p.fh;                      // feature plane + its row spectrum, floats
        if (avail >= hist) {
            p.szC = up4(szC);
            int G = avail / plane2; if (G > MOT_HALF0) G = MOT_HALF0;
            p.stage_G = G;
            p.stage_floats = up4(G * plane2 > hist ? G * plane2 : hist);
        }
    }
    kcf_pool_layout_r1(p, allow_r1);
}

// R1-resident mode of an HBM-slab template (see KcfPool): taken when the MFMA transform has an instance for the plane shape and the LDS holds
// R1 + region C + a work area of >= 3 channel planes.  zf / tmp are float2 arrays: offX keeps them 8-byte aligned (up4).
void kcf_pool_layout_r1(KcfPool& p, bool allow)
{
    p.r1_lds = 0; p.offR1c = 0; p.offX = 0; p.offW = 0; p.tile_T = 0; p.stripe_k = 0;
    if (!allow || p.use_lds || p.fft20 || p.hb > MOT_DFT_MFMA_MAX || p.wb > MOT_DFT_MFMA_MAX) return;
    const int xtl = (p.wb + 15) >> 4, ksr = (p.hb + 3) >> 2;
    if (!((xtl == 3 && (ksr == 9 || ksr == 10)) || (xtl == 2 && (ksr == 7 || ksr == 8 || ksr == 10)))) return;
    auto up4 = [](int v) { return (v + 3) & ~3; };
    const int total = (int)(MOT_LDS_LIMIT / sizeof(float));
    const int r1f = MOT_NORI * 64 * ((p.nb + 63) / 64);
    const int head = up4(2048 + 2 * p.hb + 2 * p.wb + 32 + (p.hb + 1) * (p.wb + 1));
    const int tail = up4(p.nb) + 4 * p.nbins;                        // E / resp, zf, tmp
    const int offX = r1f + head;
    const int offW = offX;                                            // the transform's tile area: E, zf, tmp are dead while it runs (partial sums wait in the slab)
    if (offX + tail >= total || p.nb > 4 * MOT_KCF_THREADS_SLAB) return;     // <= 4 cells per thread (window values in registers)
    const int plane = p.wb * 2 * p.fh;
    int T = (total - offW) / plane; if (T > 8) T = 8;
    const int cols_max = (int)(((size_t)(total - offX) * 4) / (5 * (size_t)p.ldp));   // Mq (float) + bin (byte) per pixel
    int k = (cols_max - 4) / 4;
    if (T < 3 || k < 4) return;
    const int ns = (p.wb + k - 1) / k;
    k = (p.wb + ns - 1) / ns;                                         // same number of stripes, evenly wide
    p.r1_lds = 1; p.offR1c = 0; p.offX = offX; p.offW = offW; p.tile_T = T; p.stripe_k = k;
    p.szC = offX + tail - r1f;                                       // carve() takes region C at smem + r1f
    p.stage_floats = total - p.szC; p.stage_G = 0;                   // "staging area" = everything else (kcf_lds_bytes = the whole LDS)
}

size_t kcf_lds_bytes(const KcfPool& p) { return p.use_lds ? (size_t)p.lds_floats * sizeof(float) : (size_t)(p.szC + p.stage_floats) * sizeof(float); }

template <typename K>
static hipError_t set_lds_attr(K kern, size_t bytes)
{
    // once per kernel symbol AND device (mot_env.h: a mutex-guarded table, several tracker threads / devices per process): allow the full
    // 160 KB of a gfx950 CU as dynamic LDS
    if (bytes <= 64 * 1024) return hipSuccess;
    return mot_impl::func_lds_once(reinterpret_cast<const void*>(kern), MOT_LDS_LIMIT);
}

// kernel mode of a launch: LDS-resident template: 7 for the single pool of exactly 80 x 80 px (geometry as constants), 1 when every pool involved
// is 20 x 20 cells (register FFT), else 5 / 3 (GEN: the pool's !fft20, or
// l.gen_any of a size-class launch); HBM slab: 2 with the R1-resident pipeline (the pool's, or l.r1_any), else 0
#define KCF_LAUNCH3(KERN, R1MODE, GENMODE, R1, GEN, K80, GRID, LDSB, STREAM, ...)                                                                     \
    do {                                                                                                                        \
        if (p.use_lds && !(GEN) && (K80)) { hipError_t e_ = set_lds_attr(KERN<7>, LDSB); if (e_ != hipSuccess) return e_;       \
                          hipLaunchKernelGGL(KERN<7>, dim3(GRID), dim3(MOT_KCF_THREADS), LDSB, STREAM, __VA_ARGS__); }          \
        else if (p.use_lds && !(GEN)) { hipError_t e_ = set_lds_attr(KERN<1>, LDSB); if (e_ != hipSuccess) return e_;           \
                          hipLaunchKernelGGL(KERN<1>, dim3(GRID), dim3(MOT_KCF_THREADS), LDSB, STREAM, __VA_ARGS__); }          \
        else if (p.use_lds) { hipError_t e_ = set_lds_attr(KERN<GENMODE>, LDSB); if (e_ != hipSuccess) return e_;               \
                          hipLaunchKernelGGL(KERN<GENMODE>, dim3(GRID), dim3(MOT_KCF_THREADS), LDSB, STREAM, __VA_ARGS__); }    \
        else if (R1)    { hipError_t e_ = set_lds_attr(KERN<R1MODE>, LDSB); if (e_ != hipSuccess) return e_;                    \
                          hipLaunchKernelGGL(KERN<R1MODE>, dim3(GRID), dim3(MOT_KCF_THREADS_SLAB), LDSB, STREAM, __VA_ARGS__); } \
        else            { hipError_t e_ = set_lds_attr(KERN<0>, LDSB); if (e_ != hipSuccess) return e_;                         \
                          hipLaunchKernelGGL(KERN<0>, dim3(GRID), dim3(MOT_KCF_THREADS_SLAB), LDSB, STREAM, __VA_ARGS__); }     \
    } while (0)

#ifdef MOT_KCF_ABLATE
static int g_ablate_mask = [] { const char* e = getenv("MOT_KCF_ABLATE"); return e ? (int)strtol(e, nullptr, 0) : 0; }();
static int ablate_mask() { return g_ablate_mask; }
extern "C" void mot_debug_kcf_ablate(int mask) { g_ablate_mask = mask; }   // probe build only: tools/kcf_ablate.py switches phases off for single frames
#define ABLATE_ARG(L) KcfLaunch L##_abl = L; L##_abl.ablate = ablate_mask(); const KcfLaunch& L##_use = L##_abl
#else
#define ABLATE_ARG(L) const KcfLaunch& L##_use = L
#endif
hipError_t launch_kcf_predict(const KcfPool& p, const KcfLaunch& l_in, int n, hipStream_t s, hipEvent_t t_start, hipEvent_t t_stop)
{
    ABLATE_ARG(l_in); const KcfLaunch& l = l_in_use;
#ifdef MOT_KCF_ABLATE
    // (probe build) MOT_KCF_ONE_PER_CU=1: the launch asks for more than half of a CU's LDS, so ONE workgroup runs per CU -- the occupancy experiment
    static const int one_per_cu = [] { const char* e = getenv("MOT_KCF_ONE_PER_CU"); return e ? atoi(e) : 0; }();
#define LDS_BYTES_OF(P) ((one_per_cu && (P).use_lds && kcf_lds_bytes(P) < MOT_LDS_LIMIT / 2 + 2048) ? (size_t)(MOT_LDS_LIMIT / 2 + 2048) : kcf_lds_bytes(P))
#else
#define LDS_BYTES_OF(P) kcf_lds_bytes(P)
#endif
    mot_impl::lds_poison(s);                                           // (debug) MOT_LDS_POISON
    if (n <= 0) return hipSuccess;
    if (l.pools) {                                                     // size classes: `p` is any pool of the group (use_lds is common to all)
        const size_t ldsm = l.lds_bytes;
        KCF_LAUNCH3(kcf_predict_multi_kernel, 4, 3, l.r1_any, l.gen_any, false, n, ldsm, s, l, n);
        return hipGetLastError();
    }
    const size_t lds = LDS_BYTES_OF(p);
    if ((t_start || t_stop) && p.use_lds) {                            // the launch carries the caller's events in its own packet (timing; the device loop's chain event)
        if (p.fft20 && p.rows == 80 && p.cols == 80 && (mot_impl::env().k80 & 1)) { hipError_t e = set_lds_attr(kcf_predict_kernel<7>, lds); if (e != hipSuccess) return e;
                       hipExtLaunchKernelGGL(kcf_predict_kernel<7>, dim3(n), dim3(MOT_KCF_THREADS), (unsigned)lds, s, t_start, t_stop, 0, p, l, n); }
        else if (p.fft20) { hipError_t e = set_lds_attr(kcf_predict_kernel<1>, lds); if (e != hipSuccess) return e;
                       hipExtLaunchKernelGGL(kcf_predict_kernel<1>, dim3(n), dim3(MOT_KCF_THREADS), (unsigned)lds, s, t_start, t_stop, 0, p, l, n); }
        else { hipError_t e = set_lds_attr(kcf_predict_kernel<5>, lds); if (e != hipSuccess) return e;
               hipExtLaunchKernelGGL(kcf_predict_kernel<5>, dim3(n), dim3(MOT_KCF_THREADS), (unsigned)lds, s, t_start, t_stop, 0, p, l, n); }
        return hipGetLastError();
    }
    if (t_start || t_stop) {
        if (p.r1_lds) { hipError_t e = set_lds_attr(kcf_predict_kernel<2>, lds); if (e != hipSuccess) return e;
                        hipExtLaunchKernelGGL(kcf_predict_kernel<2>, dim3(n), dim3(MOT_KCF_THREADS_SLAB), (unsigned)lds, s, t_start, t_stop, 0, p, l, n); }
        else { hipError_t e = set_lds_attr(kcf_predict_kernel<0>, lds); if (e != hipSuccess) return e;
               hipExtLaunchKernelGGL(kcf_predict_kernel<0>, dim3(n), dim3(MOT_KCF_THREADS_SLAB), (unsigned)lds, s, t_start, t_stop, 0, p, l, n); }
        return hipGetLastError();
    }
    KCF_LAUNCH3(kcf_predict_kernel, 2, 5, p.r1_lds, !p.fft20, (p.rows == 80 && p.cols == 80 && (mot_impl::env().k80 & 1)), n, lds, s, p, l, n);
    return hipGetLastError();
}

hipError_t launch_kcf_predict_features(const KcfPool& p, const KcfLaunch& lp_in, int n_pred, const KcfLaunch& lf_in, int n_feat, hipStream_t s)
{
    ABLATE_ARG(lp_in); const KcfLaunch& lp = lp_in_use; ABLATE_ARG(lf_in); const KcfLaunch& lf = lf_in_use;
    mot_impl::lds_poison(s);                                           // (debug) MOT_LDS_POISON
    if (n_pred + n_feat <= 0) return hipSuccess;
    const size_t lds = kcf_lds_bytes(p);
    KCF_LAUNCH3(kcf_predict_features_kernel, 2, 5, p.r1_lds, !p.fft20, (p.rows == 80 && p.cols == 80 && (mot_impl::env().k80 & 1)), n_pred + n_feat, lds, s, p, lp, lf, n_pred, n_feat);
    return hipGetLastError();
}

hipError_t launch_kcf_update(const KcfPool& p, const KcfLaunch& l_in, int n, hipStream_t s, bool exclusive_cu)
{
    ABLATE_ARG(l_in); const KcfLaunch& l = l_in_use;
    mot_impl::lds_poison(s);                                           // (debug) MOT_LDS_POISON
    if (n <= 0) return hipSuccess;
    if (l.pools) {
        const size_t ldsm = l.lds_bytes;
        KCF_LAUNCH3(kcf_update_multi_kernel, 4, 3, l.r1_any, l.gen_any, false, n, ldsm, s, l, n);
        return hipGetLastError();
    }
    size_t lds = kcf_lds_bytes(p);
    // exclusive_cu: ask for more than half of a CU's LDS so that no second workgroup (of this or of a concurrently running
    // KCF kernel) is placed on the same CU
    if (exclusive_cu && lds < MOT_LDS_LIMIT / 2 + 2048) lds = MOT_LDS_LIMIT / 2 + 2048;
    // (folded geometry, MOT_KCF_K80 bit 2: the direct update kernel takes it; the sparse one below is launched in mode 7 too but keeps the run-time
    // descriptor inside its out-of-line body -- see kcf_update_sparse_run)
    // grid_stride: the workgroups loop over up to n items (device-side count, usually zero): a grid of n / 8 workgroups, 4 .. 128
    const int grid = l.grid_stride ? (n / 8 < 4 ? (n < 4 ? n : 4) : (n / 8 > 128 ? 128 : n / 8)) : n;
    if (l.spec_out) {                                                  // feature-only launch: the lean kernel
        KCF_LAUNCH3(kcf_features_kernel, 2, 5, p.r1_lds, !p.fft20, (p.rows == 80 && p.cols == 80 && (mot_impl::env().k80 & 2)), n, lds, s, p, l, n);
        return hipGetLastError();
    }
#if MOT_KCF_SPARSE_VIEW
    if (l.grid_stride && p.use_lds && p.fft20 && p.rows == 80 && p.cols == 80 && (mot_impl::env().k80 & 8)) {
        hipError_t e_ = set_lds_attr(kcf_update_sparse_view_kernel<7>, lds); if (e_ != hipSuccess) return e_;
        hipLaunchKernelGGL(kcf_update_sparse_view_kernel<7>, dim3(grid), dim3(MOT_KCF_THREADS), lds, s, p, l, n);
        return hipGetLastError();
    }
#endif
    if (l.grid_stride) KCF_LAUNCH3(kcf_update_sparse_kernel, 2, 5, p.r1_lds, !p.fft20, (p.rows == 80 && p.cols == 80 && (mot_impl::env().k80 & 4)), grid, lds, s, p, l, n);
    else KCF_LAUNCH3(kcf_update_kernel, 2, 5, p.r1_lds, !p.fft20, (p.rows == 80 && p.cols == 80 && (mot_impl::env().k80 & 4)), grid, lds, s, p, l, n);
    return hipGetLastError();
}
#undef KCF_LAUNCH3

hipError_t launch_kcf_fhog_only(const KcfPool& p, const KcfLaunch& l, int n, hipStream_t s)
{
    mot_impl::lds_poison(s);                                           // (debug) MOT_LDS_POISON
    if (n <= 0) return hipSuccess;
    const size_t lds = kcf_lds_bytes(p);
    if (p.use_lds) {
        hipError_t e = set_lds_attr(kcf_fhog_kernel<true>, lds); if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kcf_fhog_kernel<true>, dim3(n), dim3(MOT_KCF_THREADS), lds, s, p, l, n);
    } else hipLaunchKernelGGL(kcf_fhog_kernel<false>, dim3(n), dim3(MOT_KCF_THREADS), 0, s, p, l, n);
    return hipGetLastError();
}

hipError_t launch_kcf_crop_only(const KcfPool& p, const KcfLaunch& l, int n, float* patch_out, hipStream_t s)
{
    mot_impl::lds_poison(s);                                           // (debug) MOT_LDS_POISON
    if (n <= 0) return hipSuccess;
    const size_t lds = kcf_lds_bytes(p);
    if (p.use_lds) {
        hipError_t e = set_lds_attr(kcf_crop_kernel<true>, lds); if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kcf_crop_kernel<true>, dim3(n), dim3(MOT_KCF_THREADS), lds, s, p, l, n, patch_out);
    } else hipLaunchKernelGGL(kcf_crop_kernel<false>, dim3(n), dim3(MOT_KCF_THREADS), 0, s, p, l, n, patch_out);
    return hipGetLastError();
}
