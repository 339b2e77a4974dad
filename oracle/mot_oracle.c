/* mot_oracle.c -- CPU restatement of the reference tracker hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see mot_oracle.h).  Parity status: PINNED against
 * golden vectors produced by the reference's own sources (tests/golden/).
 *
 * Compile with -ffp-contract=off -fno-fast-math: the reference is built for
 * plain SSE2 (no FMA) and FHOG parity is bit-exact only when every float
 * multiply and add is rounded separately.
 *
 * Citations are file:line in huangfcn/multiple-object-tracking.
 */
#include "mot_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#include "sse_tables.inc" /* MOT_SSE_RCP_TAB / MOT_SSE_RSQ_TAB (generated data) */

#define ORC_PI 3.14159265f /* libhog/gradientMex.cpp:12 */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ------------------------------------------------------------------------- */
/* x86 rcpps / rsqrtps integer model (libhog/sse.hpp:40-41 RCP / RCPSQRT).    */
/* ------------------------------------------------------------------------- */
float orc_sse_rcp(float x)
{
    uint32_t u = f2u(x), s = u & 0x80000000u, e = (u >> 23) & 0xff, m = u & 0x7fffff;
    if (e == 0xff) return u2f(m ? (u | 0x400000u) : s);
    if (e == 0) return u2f(s | 0x7f800000u);
    int ep = 253 - (int)e;
    if (ep <= 0) return u2f(s);
    return u2f(s | ((uint32_t)ep << 23) | ((uint32_t)MOT_SSE_RCP_TAB[m >> 12] << 11));
}

float orc_sse_rsqrt(float x)
{
    uint32_t u = f2u(x), s = u & 0x80000000u, e = (u >> 23) & 0xff, m = u & 0x7fffff;
    if (e == 0xff && m) return u2f(u | 0x400000u);
    if (e == 0) return u2f(s | 0x7f800000u);
    if (s) return u2f(0xffc00000u);
    if (e == 0xff) return 0.0f;
    int E = (int)e - 127, odd = E & 1;
    int ep = odd ? 126 - (E - 1) / 2 : 126 - E / 2;
    return u2f(((uint32_t)ep << 23) | ((uint32_t)MOT_SSE_RSQ_TAB[odd * 1024 + (m >> 13)] << 11));
}

/* ------------------------------------------------------------------------- */
/* FHOG (libhog/)                                                             */
/* ------------------------------------------------------------------------- */

/* gradientMex.cpp:47-56.  acos() on a float argument resolves to acosf in the
 * reference's C++ translation unit. */
const float* orc_acos_table(void)
{
    enum { n = 10000, b = 10 };
    static float a[n * 2 + b * 2];
    static int init = 0;
    float* a1 = a + n + b;
    if (init) return a1;
    for (int i = -n - b; i < -n; i++) a1[i] = ORC_PI;
    for (int i = -n; i < n; i++) a1[i] = acosf((float)i / (float)n);
    for (int i = n; i < n + b; i++) a1[i] = 0;
    for (int i = -n - b; i < n / 10; i++)
        if (a1[i] > ORC_PI - 1e-6f) a1[i] = ORC_PI - 1e-6f;
    init = 1;
    return a1;
}

/* gradientMex.cpp:15-37 (grad1) + :59-100 (gradMag), d=1, full=true. */
void orc_grad_mag(const float* I, float* M, float* O, int h, int w, int mode)
{
    const float* acost = orc_acos_table();
    for (int x = 0; x < w; x++) {
        /* column pointers for Gx: previous / next column, one-sided at borders (:18-19) */
        const float* Ic = I + (size_t)x * h;
        const float* Ip = Ic - h;
        const float* In = Ic + h;
        float rx = 0.5f;
        if (x == 0) { rx = 1.0f; Ip = Ic; }
        else if (x == w - 1) { rx = 1.0f; In = Ic; }
        if (w == 1) { Ip = Ic; In = Ic; rx = 1.0f; }
        for (int y = 0; y < h; y++) {
            float gx = (In[y] - Ip[y]) * rx;                       /* :21,24 */
            float gy;                                              /* :27-35 */
            if (h == 1) gy = 0.0f;
            else if (y == 0) gy = (Ic[1] - Ic[0]) * 1.0f;
            else if (y == h - 1) gy = (Ic[h - 1] - Ic[h - 2]) * 1.0f;
            else gy = (Ic[y + 1] - Ic[y - 1]) * 0.5f;
            float m2 = gx * gx + gy * gy;                          /* :74 */
            float m;                                               /* :83 */
            if (mode == ORC_FHOG_INTEL_APPROX) m = orc_sse_rsqrt(m2);
            else m = 1.0f / sqrtf(m2);
            if (!(m < 1e10f)) m = 1e10f;                           /* _mm_min_ps(a,b): b unless a<b */
            float mag = (mode == ORC_FHOG_INTEL_APPROX) ? orc_sse_rcp(m) : 1.0f / m; /* :84 */
            float g = (gx * m) * 10000.0f;                         /* :85 */
            g = u2f(f2u(g) ^ (f2u(gy) & 0x80000000u));             /* :86 */
            M[(size_t)x * h + y] = mag;                            /* :88 */
            float o = acost[(int)g];                               /* :90 */
            if (gy < 0) o += ORC_PI;                               /* :93-96 */
            O[(size_t)x * h + y] = o;
        }
    }
}

/* gradientMex.cpp:119,129-143: nOrients=18, full=true, non-interpolating. */
void orc_orient_bins(const float* O, int n, int* bins)
{
    const float oMult = (float)18 / (2 * ORC_PI);
    for (int i = 0; i < n; i++) {
        float o = O[i] * oMult;
        int o0 = (int)(o + .5f);
        if (o0 >= 18) o0 = 0;
        bins[i] = o0;
    }
}

/* gradientMex.cpp:148-231 with bin=4, nOrients=18, softBin=-1 (trilinear branch
 * :183-221 with nearest orientation), full=true.  R1 must be zeroed by caller. */
void orc_grad_hist(const float* M, const float* O, float* R1, int h, int w)
{
    const int bin = 4, nOrients = 18;
    const int hb = h / bin, wb = w / bin, h0 = hb * bin, w0 = wb * bin, nb = wb * hb;
    const float sInv = 1 / (float)bin, sInv2 = 1 / (float)bin / (float)bin;
    int* bins = (int*)malloc(sizeof(int) * (size_t)(h > 0 ? h : 1));
    const float init = (0 + .5f) * sInv - 0.5f;                    /* :187 */
    float xb = init;
    for (int x = 0; x < w0; x++) {
        orc_orient_bins(O + (size_t)x * h, h0, bins);              /* :159 */
        const int hasLf = xb >= 0;                                 /* :188 */
        const int xb0 = hasLf ? (int)xb : -1;
        const int hasRt = xb0 < wb - 1;
        const float xd = xb - xb0;                                 /* :189 */
        xb += sInv;
        float yb = init;
        for (int y = 0; y < h0; y++) {
            int yb0;
            if (y < bin / 2) yb0 = -1;                             /* :195-196 */
            else yb0 = (int)yb;                                    /* :202,215 */
            const float yd = yb - yb0;                             /* :191 */
            yb += sInv;
            const float xyd = xd * yd;
            const float ms0 = 1 - xd - yd + xyd, ms1 = yd - xyd, ms2 = xd - xyd, ms3 = xyd; /* :192 */
            const float m0 = M[(size_t)x * h + y] * sInv2;         /* :132,143 */
            float* H0 = R1 + (size_t)bins[y] * nb;
            const int top = yb0 >= 0;          /* cell row yb0 exists */
            const int bot = yb0 < hb - 1;      /* cell row yb0+1 exists (always true for leading rows) */
            if (hasLf) {
                if (top) H0[xb0 * hb + yb0] += ms0 * m0;           /* :203,216 */
                if (bot) H0[xb0 * hb + yb0 + 1] += ms1 * m0;       /* :197,203 */
            }
            if (hasRt) {
                if (top) H0[(xb0 + 1) * hb + yb0] += ms2 * m0;     /* :204,217 */
                if (bot) H0[(xb0 + 1) * hb + yb0 + 1] += ms3 * m0; /* :198,204 */
            }
        }
    }
    free(bins);
    /* :225-230 boundary bins get 8/7 (corners twice) */
    const float c = 8.f / 7.f;
    for (int o = 0; o < nOrients; o++) {
        float* H = R1 + (size_t)o * nb;
        for (int y = 0; y < hb; y++) H[0 * hb + y] *= c;
        for (int x = 0; x < wb; x++) H[x * hb + 0] *= c;
        for (int y = 0; y < hb; y++) H[(wb - 1) * hb + y] *= c;
        for (int x = 0; x < wb; x++) H[x * hb + hb - 1] *= c;
    }
}

/* gradientMex.cpp:236-253 for hb,wb >= 2: the eight replication statements reduce
 * to clamping the index into the normalised interior [1..wb-1] x [1..hb-1]. */
static void orc_hog_norm_matrix(const float* R2, int nOrients, int hb, int wb, float* N)
{
    const int hb1 = hb + 1, wb1 = wb + 1;
    const float eps = 1e-4f / 4 / 4 / 4 / 4 / 4;                   /* :238 (bin=4) */
    float* E = (float*)calloc((size_t)hb1 * wb1, sizeof(float));
    for (int o = 0; o < nOrients; o++)
        for (int x = 0; x < wb; x++)
            for (int y = 0; y < hb; y++) {
                float v = R2[(size_t)o * wb * hb + x * hb + y];
                E[(x + 1) * hb1 + (y + 1)] += v * v;               /* :241 */
            }
    for (int x = 0; x < wb - 1; x++)
        for (int y = 0; y < hb - 1; y++) {
            const float* n = E + (x + 1) * hb1 + (y + 1);
            N[(x + 1) * hb1 + (y + 1)] = 1 / sqrtf(n[0] + n[1] + n[hb1] + n[hb1 + 1] + eps); /* :243 */
        }
    for (int x = 0; x < wb1; x++)
        for (int y = 0; y < hb1; y++) {
            int cx = x < 1 ? 1 : (x > wb - 1 ? wb - 1 : x);
            int cy = y < 1 ? 1 : (y > hb - 1 ? hb - 1 : y);
            if (cx != x || cy != y) N[x * hb1 + y] = N[cx * hb1 + cy]; /* :244-251 */
        }
    free(E);
}

/* gradientMex.cpp:256-280 types 1 and 2 */
static void orc_hog_channels(float* H, const float* R, const float* N, int hb, int wb, int nOrients, float clip, int type)
{
    const float r = .2357f;
    const int nb = wb * hb, hb1 = hb + 1;
    for (int o = 0; o < nOrients; o++)
        for (int x = 0; x < wb; x++)
            for (int y = 0; y < hb; y++) {
                const float v = R[(size_t)o * nb + x * hb + y];
                /* N1[y-blk], blk = 0,1,hb1,hb1+1 -> N[x+1][y+1], N[x+1][y], N[x][y+1], N[x][y] */
                const float nn[4] = { N[(x + 1) * hb1 + y + 1], N[(x + 1) * hb1 + y], N[x * hb1 + y + 1], N[x * hb1 + y] };
                for (int c = 0; c < 4; c++) {
                    float t = v * nn[c];
                    if (t > clip) t = clip;
                    if (type == 1) H[(size_t)o * nb + x * hb + y] += t * .5f;     /* :271-272 */
                    else H[(size_t)c * nb + x * hb + y] += t * r;                 /* :275-276 */
                }
            }
}

/* fhog.h:16-38 + gradientMex.cpp:298-317 */
void orc_fhog(const float* I, int h, int w, float* H, int mode)
{
    const int hb = h / 4, wb = w / 4, nb = hb * wb, nbo = nb * 9;
    float* M = (float*)malloc(sizeof(float) * (size_t)h * w * 2);
    float* O = M + (size_t)h * w;
    orc_grad_mag(I, M, O, h, w, mode);
    memset(H, 0, sizeof(float) * (size_t)nb * 32);                 /* fhog.h:31 */
    float* R1 = (float*)calloc((size_t)nb * 18, sizeof(float));
    orc_grad_hist(M, O, R1, h, w);                                 /* :305 */
    float* R2 = (float*)calloc((size_t)nb * 9, sizeof(float));
    for (int o = 0; o < 9; o++)
        for (int x = 0; x < nb; x++) R2[o * nb + x] = R1[o * nb + x] + R1[(o + 9) * nb + x]; /* :308-309 */
    float* N = (float*)calloc((size_t)(hb + 1) * (wb + 1), sizeof(float));
    orc_hog_norm_matrix(R2, 9, hb, wb, N);                         /* :311 */
    orc_hog_channels(H + nbo * 0, R1, N, hb, wb, 18, 0.2f, 1);     /* :313 */
    orc_hog_channels(H + nbo * 2, R2, N, hb, wb, 9, 0.2f, 1);      /* :314 */
    orc_hog_channels(H + nbo * 3, R1, N, hb, wb, 18, 0.2f, 2);     /* :315 */
    free(N); free(R2); free(R1); free(M);
}

/* ------------------------------------------------------------------------- */
/* crop / gray / resize (top/drawlib.c)                                       */
/* ------------------------------------------------------------------------- */
void orc_rgb2gray(float* dst, const uint8_t* frame, int left, int top, int right, int bottom)
{
    if (top > bottom) { int t = top; top = bottom; bottom = t; }   /* drawlib.c:203-215 */
    if (left > right) { int t = left; left = right; right = t; }
    const int cols = right - left + 1, rows = bottom - top + 1;
    for (int r = 0; r < rows; r++) {
        const uint8_t* p = frame + (size_t)(top + r) * 3840 + (size_t)left * 3; /* PIXEL_AT :9 */
        for (int c = 0; c < cols; c++) {
            uint8_t B = p[0], G = p[1], R = p[2];
            p += 3;
            dst[(size_t)c * rows + r] = (float)(0.144 * B + 0.587 * G + 0.299 * R); /* :234 */
        }
    }
}

void orc_resize_gray(float* dst, const float* src, int hs, int ws, int h, int w)
{
    const float xs = ((float)ws) / ((float)w);                     /* drawlib.c:551-552 */
    const float ys = ((float)hs) / ((float)h);
    for (int y = 0; y < h; y++) {
        float sy = y * ys;                                         /* :598 */
        int y0 = (int)sy;
        float fracy = sy - y0, ifracy = 1.0f - fracy;
        int y1 = y0 + 1;
        if (y1 >= hs) y1 = y0;
        for (int x = 0; x < w; x++) {
            float sx = x * xs;                                     /* :573 */
            int x0 = (int)sx;
            float fracx = sx - x0, ifracx = 1.0f - fracx;
            int x1 = x0 + 1;
            if (x1 >= ws) x1 = x0;
            float c1 = src[y0 * ws + x0], c2 = src[y0 * ws + x1];
            float c3 = src[y1 * ws + x0], c4 = src[y1 * ws + x1];
            float l0 = ifracx * c1 + fracx * c2;                   /* :625-627 */
            float l1 = ifracx * c3 + fracx * c4;
            dst[(size_t)y * w + x] = ifracy * l0 + fracy * l1;
        }
    }
}

/* top/td.cpp:348-364: rgb2Gray into scratch, then resize with
 * (heightSource,widthSource,height,width) = (rows_s, cols_s, rows_d, cols_d). */
void orc_crop_patch(float* dst, float* scratch, const uint8_t* frame, const orc_bbox_t* box, int rows, int cols)
{
    orc_rgb2gray(scratch, frame, box->l, box->t, box->r, box->b);
    orc_resize_gray(dst, scratch, box->b - box->t + 1, box->r - box->l + 1, rows, cols);
}

/* ------------------------------------------------------------------------- */
/* DFT helpers.  The reference calls FFTW 3.3.5 single precision (binary only,
 * bin/libfftw3f-3.dll; call sites kcf.cpp:134-143,180-195,265,399).  Restated
 * here as the definition of the transform (direct DFT, double accumulation,
 * rounded to float on output): layout n0=f_cols (slow) x n1=f_rows (fast),
 * half spectrum along n1, inverse unnormalised.                              */
/* ------------------------------------------------------------------------- */
typedef struct { int n; double* c; double* s; } orc_tw;
static void tw_init(orc_tw* t, int n)
{
    t->n = n;
    t->c = (double*)malloc(sizeof(double) * (size_t)n);
    t->s = (double*)malloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < n; i++) {
        double a = 2.0 * 3.14159265358979323846 * (double)i / (double)n;
        t->c[i] = cos(a); t->s[i] = sin(a);
    }
    /* exact values at the quadrant points */
    if (n % 4 == 0) { t->c[n / 4] = 0; t->s[n / 4] = 1; t->c[3 * n / 4] = 0; t->s[3 * n / 4] = -1; }
    if (n % 2 == 0) { t->c[n / 2] = -1; t->s[n / 2] = 0; }
}
static void tw_free(orc_tw* t) { free(t->c); free(t->s); }

/* forward r2c 2-D: in[n0][n1] real -> out[n0][n1/2+1] complex (interleaved float) */
static void dft_r2c_2d(const float* in, float* out, int n0, int n1, const orc_tw* t0, const orc_tw* t1, double* work)
{
    const int nh = n1 / 2 + 1;
    double* Y = work; /* n0*nh*2 */
    for (int c = 0; c < n0; c++)
        for (int k = 0; k < nh; k++) {
            double re = 0, im = 0;
            for (int r = 0; r < n1; r++) {
                int j = (int)(((long)k * r) % n1);
                double v = in[c * n1 + r];
                re += v * t1->c[j]; im -= v * t1->s[j];
            }
            Y[(c * nh + k) * 2] = re; Y[(c * nh + k) * 2 + 1] = im;
        }
    for (int cp = 0; cp < n0; cp++)
        for (int k = 0; k < nh; k++) {
            double re = 0, im = 0;
            for (int c = 0; c < n0; c++) {
                int j = (int)(((long)cp * c) % n0);
                double yr = Y[(c * nh + k) * 2], yi = Y[(c * nh + k) * 2 + 1];
                double wr = t0->c[j], wi = -t0->s[j];
                re += yr * wr - yi * wi; im += yr * wi + yi * wr;
            }
            out[(cp * nh + k) * 2] = (float)re; out[(cp * nh + k) * 2 + 1] = (float)im;
        }
}

/* inverse c2r 2-D, unnormalised: in[n0][n1/2+1] complex -> out[n0][n1] real */
static void dft_c2r_2d(const float* in, float* out, int n0, int n1, const orc_tw* t0, const orc_tw* t1, double* work)
{
    const int nh = n1 / 2 + 1;
    double* Y = work;
    for (int c = 0; c < n0; c++)
        for (int k = 0; k < nh; k++) {
            double re = 0, im = 0;
            for (int cp = 0; cp < n0; cp++) {
                int j = (int)(((long)cp * c) % n0);
                double xr = in[(cp * nh + k) * 2], xi = in[(cp * nh + k) * 2 + 1];
                double wr = t0->c[j], wi = t0->s[j];
                re += xr * wr - xi * wi; im += xr * wi + xi * wr;
            }
            Y[(c * nh + k) * 2] = re; Y[(c * nh + k) * 2 + 1] = im;
        }
    for (int c = 0; c < n0; c++)
        for (int r = 0; r < n1; r++) {
            double acc = Y[(c * nh) * 2];
            for (int k = 1; k < nh; k++) {
                int j = (int)(((long)k * r) % n1);
                double yr = Y[(c * nh + k) * 2], yi = Y[(c * nh + k) * 2 + 1];
                double term = yr * t1->c[j] - yi * t1->s[j];
                if ((n1 % 2 == 0) && k == n1 / 2) acc += term; else acc += 2.0 * term;
            }
            out[c * n1 + r] = (float)acc;
        }
}

/* ------------------------------------------------------------------------- */
/* KCF (trackers/kcf.cpp)                                                     */
/* ------------------------------------------------------------------------- */
struct orc_kcf {
    int rows, cols, f_rows, f_cols, f_chan, cell, mode;
    float *xf_tm, *xf_fq, *xf_md, *yf, *zf, *kf, *alpha, *response, *labels, *cos_win, *hog;
    float norm;
    orc_bbox_t pos;
    float scale_vert, scale_horiz;
    int first_update; float factor, lamda;
    orc_tw t0, t1; double* work;
};

static void kcf_labels(float* out, float sigma, int rows, int cols) /* kcf.cpp:96-122 + circshift :78-94 */
{
    float* xv = (float*)malloc(sizeof(float) * rows);
    float* yv = (float*)malloc(sizeof(float) * cols);
    int rx0 = -rows / 2, ry0 = -cols / 2;
    float sigma_s_inv = (float)(1.0 / (double)(sigma * sigma));    /* :104 */
    for (int i = 0; i < rows; i++) { int x = rx0 + i; xv[i] = (float)exp(-0.5 * x * x * (double)sigma_s_inv); }
    for (int j = 0; j < cols; j++) { int y = ry0 + j; yv[j] = (float)exp(-0.5 * y * y * (double)sigma_s_inv); }
    for (int j = 0; j < cols; j++) {
        int jj = (j + ry0) % cols; if (jj < 0) jj += cols;
        for (int i = 0; i < rows; i++) {
            int ii = (i + rx0) % rows; if (ii < 0) ii += rows;
            out[jj * rows + ii] = xv[i] * yv[j];                   /* :115 outer product, :91 shift */
        }
    }
    free(xv); free(yv);
}

static void hann_f(float* h, int N) /* include/sigpack/window/window.h:34-48,83-89 */
{
    const double PI_2 = 6.28318530717958647692;
    for (int i = 0; i < N; i++) {
        double ha = 0.5 - 0.5 * cos(1.0 * PI_2 * i / (N - 1)) + 0.0 * cos(2.0 * PI_2 * i / (N - 1))
                    - 0.0 * cos(3.0 * PI_2 * i / (N - 1)) + 0.0 * cos(4.0 * PI_2 * i / (N - 1));
        h[i] = (float)ha;
    }
}

orc_kcf* orc_kcf_new(const orc_bbox_t* box, int mode)
{
    orc_kcf* k = (orc_kcf*)calloc(1, sizeof(orc_kcf));
    k->rows = box->b - box->t + 1; k->cols = box->r - box->l + 1;  /* kcf.cpp:148-149 */
    k->cell = 4; k->f_rows = k->rows / 4; k->f_cols = k->cols / 4; k->f_chan = 31; k->mode = mode;
    const int nf = k->f_rows * k->f_cols, nh = k->f_cols * (k->f_rows / 2 + 1);
    k->hog = (float*)calloc((size_t)nf * 32, sizeof(float));
    k->xf_tm = (float*)calloc((size_t)nf * 32, sizeof(float));
    k->xf_fq = (float*)calloc((size_t)nh * 2 * 32, sizeof(float));
    k->xf_md = (float*)calloc((size_t)nh * 2 * 32, sizeof(float)); /* :175 zeroed */
    k->yf = (float*)calloc((size_t)nh * 2, sizeof(float));
    k->zf = (float*)calloc((size_t)nh * 2, sizeof(float));
    k->kf = (float*)calloc((size_t)nh * 2, sizeof(float));
    k->alpha = (float*)calloc((size_t)nh, sizeof(float));           /* :174 */
    k->response = (float*)calloc((size_t)nf, sizeof(float));
    k->labels = (float*)calloc((size_t)nf, sizeof(float));
    k->cos_win = (float*)calloc((size_t)nf, sizeof(float));
    k->norm = (float)(1.0 / (double)((float)(k->f_cols * k->f_rows * k->f_chan))); /* :197 */
    k->scale_vert = 1.0f; k->scale_horiz = 1.0f; k->pos = *box;    /* :200-202 */
    tw_init(&k->t0, k->f_cols); tw_init(&k->t1, k->f_rows);
    k->work = (double*)malloc(sizeof(double) * (size_t)nh * 2);
    kcf_labels(k->labels, 0.7289f, k->f_rows, k->f_cols);          /* :205 */
    {                                                              /* :124-130 */
        float* hy = (float*)malloc(sizeof(float) * k->f_rows);
        float* hx = (float*)malloc(sizeof(float) * k->f_cols);
        hann_f(hy, k->f_rows); hann_f(hx, k->f_cols);
        for (int j = 0; j < k->f_cols; j++)
            for (int i = 0; i < k->f_rows; i++) k->cos_win[j * k->f_rows + i] = hy[i] * hx[j];
        free(hy); free(hx);
    }
    dft_r2c_2d(k->labels, k->yf, k->f_cols, k->f_rows, &k->t0, &k->t1, k->work); /* :207 */
    k->first_update = 1; k->factor = 0.05f; k->lamda = 0.0001f;    /* :210-212 */
    return k;
}

void orc_kcf_delete(orc_kcf* k)
{
    if (!k) return;
    free(k->hog); free(k->xf_tm); free(k->xf_fq); free(k->xf_md); free(k->yf); free(k->zf); free(k->kf);
    free(k->alpha); free(k->response); free(k->labels); free(k->cos_win); free(k->work);
    tw_free(&k->t0); tw_free(&k->t1); free(k);
}

static void kcf_features_fft(orc_kcf* k, const float* patch)       /* kcf.cpp:245-267 */
{
    const int nf = k->f_rows * k->f_cols, nh = k->f_cols * (k->f_rows / 2 + 1);
    orc_fhog(patch, k->rows, k->cols, k->hog, k->mode);
    for (int l = 0; l < k->f_chan; l++)
        for (int i = 0; i < nf; i++) k->xf_tm[l * nf + i] = k->hog[l * nf + i] * k->cos_win[i];
    for (int l = 0; l < k->f_chan; l++)
        dft_r2c_2d(k->xf_tm + (size_t)l * nf, k->xf_fq + (size_t)l * nh * 2, k->f_cols, k->f_rows, &k->t0, &k->t1, k->work);
}

void orc_kcf_predict(orc_kcf* k, const float* patch, orc_bbox_t* out)
{
    const int nh = k->f_cols * (k->f_rows / 2 + 1);
    kcf_features_fft(k, patch);
    /* kcf.cpp:306-362 */
    for (int i = 0; i < nh; i++) { k->zf[2 * i] = 0; k->zf[2 * i + 1] = 0; }
    for (int l = 0; l < k->f_chan; l++) {
        const float* a = k->xf_fq + (size_t)l * nh * 2; const float* b = k->xf_md + (size_t)l * nh * 2;
        for (int i = 0; i < nh; i++) {
            float ia = a[2 * i], qa = a[2 * i + 1], ib = b[2 * i], qb = b[2 * i + 1];
            float ic = ia * ib + qa * qb, qc = qa * ib - ia * qb;
            if (l == 0) { k->zf[2 * i] = ic; k->zf[2 * i + 1] = qc; }
            else { k->zf[2 * i] += ic; k->zf[2 * i + 1] += qc; }
        }
    }
    for (int i = 0; i < nh; i++) {
        k->zf[2 * i] = (k->zf[2 * i] * k->alpha[i]) * k->norm;
        k->zf[2 * i + 1] = (k->zf[2 * i + 1] * k->alpha[i]) * k->norm;
    }
    /* kcf.cpp:397-428 */
    dft_c2r_2d(k->zf, k->response, k->f_cols, k->f_rows, &k->t0, &k->t1, k->work);
    float max_val = -99999.0f; int vd = 1, hd = 1;
    const float* p = k->response;
    for (int j = 1; j <= k->f_cols; j++)
        for (int i = 1; i <= k->f_rows; i++) { if (p[0] > max_val) { max_val = p[0]; vd = i; hd = j; } ++p; }
    if (vd > k->f_rows / 2) vd -= k->f_rows;
    if (hd > k->f_cols / 2) hd -= k->f_cols;
    k->pos.t = (int)((float)k->pos.t + (float)(k->cell * (vd - 1)) * k->scale_vert);   /* :424-427 */
    k->pos.b = (int)((float)k->pos.b + (float)(k->cell * (vd - 1)) * k->scale_vert);
    k->pos.l = (int)((float)k->pos.l + (float)(k->cell * (hd - 1)) * k->scale_horiz);
    k->pos.r = (int)((float)k->pos.r + (float)(k->cell * (hd - 1)) * k->scale_horiz);
    *out = k->pos;                                                 /* :438 */
}

void orc_kcf_update(orc_kcf* k, const float* patch, const orc_bbox_t* box)
{
    const int nh = k->f_cols * (k->f_rows / 2 + 1);
    k->pos = *box;                                                 /* kcf.cpp:470-472 */
    k->scale_horiz = ((float)(box->r - box->l + 1)) / ((float)k->cols);
    k->scale_vert = ((float)(box->b - box->t + 1)) / ((float)k->rows);
    float factor = k->first_update ? 1.0f : k->factor;             /* :443 */
    k->first_update = 0;
    kcf_features_fft(k, patch);
    /* :269-304 */
    for (int l = 0; l < k->f_chan; l++) {
        const float* a = k->xf_fq + (size_t)l * nh * 2;
        for (int i = 0; i < nh; i++) {
            float v = (a[2 * i] * a[2 * i]) + (a[2 * i + 1] * a[2 * i + 1]);
            if (l == 0) k->kf[2 * i] = v; else k->kf[2 * i] = v + k->kf[2 * i];
        }
    }
    for (int i = 0; i < nh; i++) k->kf[2 * i] = k->kf[2 * i] * k->norm;
    /* :364-378 */
    for (int i = 0; i < nh; i++) {
        float a = k->yf[2 * i] / (k->kf[2 * i] + k->lamda);
        k->alpha[i] = (1 - factor) * k->alpha[i] + factor * a;
    }
    /* :380-395 */
    for (int i = 0; i < nh * k->f_chan; i++) {
        k->xf_md[2 * i] = (1 - factor) * k->xf_md[2 * i] + factor * k->xf_fq[2 * i];
        k->xf_md[2 * i + 1] = (1 - factor) * k->xf_md[2 * i + 1] + factor * k->xf_fq[2 * i + 1];
    }
}

int orc_kcf_rows(const orc_kcf* k) { return k->rows; }
int orc_kcf_cols(const orc_kcf* k) { return k->cols; }
int orc_kcf_frows(const orc_kcf* k) { return k->f_rows; }
int orc_kcf_fcols(const orc_kcf* k) { return k->f_cols; }
const float* orc_kcf_response(const orc_kcf* k) { return k->response; }
const float* orc_kcf_alpha(const orc_kcf* k) { return k->alpha; }
const float* orc_kcf_xm(const orc_kcf* k) { return k->xf_md; }
const float* orc_kcf_xf(const orc_kcf* k) { return k->xf_fq; }
const float* orc_kcf_yf(const orc_kcf* k) { return k->yf; }
const float* orc_kcf_labels(const orc_kcf* k) { return k->labels; }
const float* orc_kcf_coswin(const orc_kcf* k) { return k->cos_win; }
const float* orc_kcf_features(const orc_kcf* k) { return k->xf_tm; }
void orc_kcf_get_pos(const orc_kcf* k, orc_bbox_t* out) { *out = k->pos; }

/* ------------------------------------------------------------------------- */
/* Kalman (trackers/kalman.cpp; sp::KF include/sigpack/kalman/kalman.h)       */
/* All matrices column-major (armadillo), float64.                            */
/* ------------------------------------------------------------------------- */
struct orc_kalman { double x[6], P[36], A[36], H[24], Q[36], R[16]; };

#define IX(r, c, nr) ((c) * (nr) + (r))
static void matmul(double* C, const double* A, const double* B, int m, int k, int n, int tB)
{   /* C[m x n] = A[m x k] * (tB ? B^T : B); B is [k x n] or, if tB, [n x k] */
    for (int j = 0; j < n; j++)
        for (int i = 0; i < m; i++) {
            double acc = 0;
            for (int p = 0; p < k; p++) acc += A[IX(i, p, m)] * (tB ? B[IX(j, p, n)] : B[IX(p, j, k)]);
            C[IX(i, j, m)] = acc;
        }
}

orc_kalman* orc_kalman_new(const orc_bbox_t* box)                   /* kalman.cpp:148-162, 29-97 */
{
    orc_kalman* k = (orc_kalman*)calloc(1, sizeof(orc_kalman));
    static const double A[6][6] = { {1,0,0,0,1,0},{0,1,0,0,0,1},{0,0,1,0,1,0},{0,0,0,1,0,1},{0,0,0,0,1,0},{0,0,0,0,0,1} };
    static const double Qb[6][6] = { {.25,0,0,0,.5,0},{0,.25,0,0,0,.5},{0,0,.25,0,.5,0},{0,0,0,.25,0,.5},{.5,0,.5,0,1,0},{0,.5,0,.5,0,1} };
    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) {
        k->A[IX(r, c, 6)] = A[r][c];
        k->Q[IX(r, c, 6)] = 1e-2 * Qb[r][c];                       /* :84 */
        k->P[IX(r, c, 6)] = (r == c) ? 1e+4 : 0.0;                 /* :90 */
    }
    for (int r = 0; r < 4; r++) for (int c = 0; c < 6; c++) k->H[IX(r, c, 4)] = (r == c) ? 1.0 : 0.0;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) k->R[IX(r, c, 4)] = (r == c) ? 512.0 : 0.0; /* :87 */
    k->x[0] = box->l; k->x[1] = box->t; k->x[2] = box->r; k->x[3] = box->b; k->x[4] = 0; k->x[5] = 0;
    return k;
}
void orc_kalman_delete(orc_kalman* k) { free(k); }

void orc_kalman_predict(orc_kalman* k, orc_bbox_t* out)             /* kalman.h:207-211, kalman.cpp:105-116 */
{
    double xn[6], T[36], Pn[36];
    matmul(xn, k->A, k->x, 6, 6, 1, 0);
    memcpy(k->x, xn, sizeof xn);
    matmul(T, k->A, k->P, 6, 6, 6, 0);                             /* (A*P)*A' glue_times_meat.hpp:763-766 */
    matmul(Pn, T, k->A, 6, 6, 6, 1);
    for (int i = 0; i < 36; i++) k->P[i] = Pn[i] + k->Q[i];
    out->l = (int)k->x[0]; out->t = (int)k->x[1]; out->r = (int)k->x[2]; out->b = (int)k->x[3];
}

static void inv4(double* out, const double* m)                      /* adjugate / determinant (auxlib_meat.hpp:182-) */
{
    double inv[16];
    inv[0] = m[5]*m[10]*m[15] - m[5]*m[11]*m[14] - m[9]*m[6]*m[15] + m[9]*m[7]*m[14] + m[13]*m[6]*m[11] - m[13]*m[7]*m[10];
    inv[4] = -m[4]*m[10]*m[15] + m[4]*m[11]*m[14] + m[8]*m[6]*m[15] - m[8]*m[7]*m[14] - m[12]*m[6]*m[11] + m[12]*m[7]*m[10];
    inv[8] = m[4]*m[9]*m[15] - m[4]*m[11]*m[13] - m[8]*m[5]*m[15] + m[8]*m[7]*m[13] + m[12]*m[5]*m[11] - m[12]*m[7]*m[9];
    inv[12] = -m[4]*m[9]*m[14] + m[4]*m[10]*m[13] + m[8]*m[5]*m[14] - m[8]*m[6]*m[13] - m[12]*m[5]*m[10] + m[12]*m[6]*m[9];
    inv[1] = -m[1]*m[10]*m[15] + m[1]*m[11]*m[14] + m[9]*m[2]*m[15] - m[9]*m[3]*m[14] - m[13]*m[2]*m[11] + m[13]*m[3]*m[10];
    inv[5] = m[0]*m[10]*m[15] - m[0]*m[11]*m[14] - m[8]*m[2]*m[15] + m[8]*m[3]*m[14] + m[12]*m[2]*m[11] - m[12]*m[3]*m[10];
    inv[9] = -m[0]*m[9]*m[15] + m[0]*m[11]*m[13] + m[8]*m[1]*m[15] - m[8]*m[3]*m[13] - m[12]*m[1]*m[11] + m[12]*m[3]*m[9];
    inv[13] = m[0]*m[9]*m[14] - m[0]*m[10]*m[13] - m[8]*m[1]*m[14] + m[8]*m[2]*m[13] + m[12]*m[1]*m[10] - m[12]*m[2]*m[9];
    inv[2] = m[1]*m[6]*m[15] - m[1]*m[7]*m[14] - m[5]*m[2]*m[15] + m[5]*m[3]*m[14] + m[13]*m[2]*m[7] - m[13]*m[3]*m[6];
    inv[6] = -m[0]*m[6]*m[15] + m[0]*m[7]*m[14] + m[4]*m[2]*m[15] - m[4]*m[3]*m[14] - m[12]*m[2]*m[7] + m[12]*m[3]*m[6];
    inv[10] = m[0]*m[5]*m[15] - m[0]*m[7]*m[13] - m[4]*m[1]*m[15] + m[4]*m[3]*m[13] + m[12]*m[1]*m[7] - m[12]*m[3]*m[5];
    inv[14] = -m[0]*m[5]*m[14] + m[0]*m[6]*m[13] + m[4]*m[1]*m[14] - m[4]*m[2]*m[13] - m[12]*m[1]*m[6] + m[12]*m[2]*m[5];
    inv[3] = -m[1]*m[6]*m[11] + m[1]*m[7]*m[10] + m[5]*m[2]*m[11] - m[5]*m[3]*m[10] - m[9]*m[2]*m[7] + m[9]*m[3]*m[6];
    inv[7] = m[0]*m[6]*m[11] - m[0]*m[7]*m[10] - m[4]*m[2]*m[11] + m[4]*m[3]*m[10] + m[8]*m[2]*m[7] - m[8]*m[3]*m[6];
    inv[11] = -m[0]*m[5]*m[11] + m[0]*m[7]*m[9] + m[4]*m[1]*m[11] - m[4]*m[3]*m[9] - m[8]*m[1]*m[7] + m[8]*m[3]*m[5];
    inv[15] = m[0]*m[5]*m[10] - m[0]*m[6]*m[9] - m[4]*m[1]*m[10] + m[4]*m[2]*m[9] + m[8]*m[1]*m[6] - m[8]*m[2]*m[5];
    double det = m[0]*inv[0] + m[1]*inv[4] + m[2]*inv[8] + m[3]*inv[12];
    for (int i = 0; i < 16; i++) out[i] = inv[i] / det;
}

void orc_kalman_update(orc_kalman* k, const orc_bbox_t* box)        /* kalman.h:225-237, kalman.cpp:118-128 */
{
    double z[4] = { (double)box->l, (double)box->t, (double)box->r, (double)box->b };
    double HP[24], S[16], Si[16], PHt[24], K[24], Hx[4], err[4], KH[36], Jf[36], T[36], JPJ[36], KR[24], KRK[36];
    matmul(HP, k->H, k->P, 4, 6, 6, 0);
    matmul(S, HP, k->H, 4, 6, 4, 1);
    for (int i = 0; i < 16; i++) S[i] += k->R[i];
    inv4(Si, S);
    matmul(PHt, k->P, k->H, 6, 6, 4, 1);
    matmul(K, PHt, Si, 6, 4, 4, 0);                                 /* K = (P*H')*inv(S) */
    matmul(Hx, k->H, k->x, 4, 6, 1, 0);
    for (int i = 0; i < 4; i++) err[i] = z[i] - Hx[i];
    for (int i = 0; i < 6; i++) { double a = 0; for (int j = 0; j < 4; j++) a += K[IX(i, j, 6)] * err[j]; k->x[i] += a; }
    matmul(KH, K, k->H, 6, 4, 6, 0);
    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) Jf[IX(r, c, 6)] = ((r == c) ? 1.0 : 0.0) - KH[IX(r, c, 6)];
    matmul(T, Jf, k->P, 6, 6, 6, 0);
    matmul(JPJ, T, Jf, 6, 6, 6, 1);
    matmul(KR, K, k->R, 6, 4, 4, 0);
    matmul(KRK, KR, K, 6, 4, 6, 1);
    for (int i = 0; i < 36; i++) k->P[i] = JPJ[i] + KRK[i];
}

void orc_kalman_get_state(const orc_kalman* k, double* x6, double* P36)
{
    if (x6) memcpy(x6, k->x, sizeof(double) * 6);
    if (P36) memcpy(P36, k->P, sizeof(double) * 36);
}

/* ------------------------------------------------------------------------- */
/* association cost (top/td.cpp:386-457)                                      */
/* ------------------------------------------------------------------------- */
static double pair_cost(const orc_bbox_t* a, const orc_bbox_t* d)
{
    int cxi = (a->l + a->r) >> 1, cyi = (a->t + a->b) >> 1;        /* td.cpp:407-411 */
    int cxj = (d->l + d->r) >> 1, cyj = (d->t + d->b) >> 1;
    double dist = 0.0;
    dist += sqrt((double)((cxi - cxj) * (cxi - cxj) + (cyi - cyj) * (cyi - cyj))) * (1.0 / ((double)1280)); /* :413, SCREEN_DIS :50 */
    if (a->type != d->type) dist += 1.0;                           /* :415-418 */
    return dist;
}

void orc_cost_matrix(const orc_bbox_t* trk, int nT, const orc_bbox_t* det, int nD, double* dist)
{
    double* p = dist;
    if (nT < nD) { for (int j = 0; j < nD; j++) for (int i = 0; i < nT; i++) *p++ = pair_cost(&trk[i], &det[j]); }
    else { for (int i = 0; i < nT; i++) for (int j = 0; j < nD; j++) *p++ = pair_cost(&trk[i], &det[j]); }
}

/* ------------------------------------------------------------------------- */
/* Munkres (trackers/hungarian/hungarian.cpp), restated as an explicit state   */
/* machine instead of mutually recursive steps; same scan orders and the same  */
/* floating-point updates.                                                     */
/* ------------------------------------------------------------------------- */
void orc_assignment_optimal(int* assignment, double* cost, const double* distIn, int nR, int nC)
{
    const int nE = nR * nC;
    *cost = 0;
    for (int r = 0; r < nR; r++) assignment[r] = -1;               /* :36-40 */
    if (nE == 0) return;
    double* d = (double*)malloc(sizeof(double) * (size_t)nE);
    memcpy(d, distIn, sizeof(double) * (size_t)nE);
    unsigned char* covC = (unsigned char*)calloc((size_t)nC, 1);
    unsigned char* covR = (unsigned char*)calloc((size_t)nR, 1);
    unsigned char* star = (unsigned char*)calloc((size_t)nE, 1);
    unsigned char* prime = (unsigned char*)calloc((size_t)nE, 1);
    unsigned char* nstar = (unsigned char*)calloc((size_t)nE, 1);
    int minDim;
    if (nR <= nC) {                                                /* :65-102 */
        minDim = nR;
        for (int r = 0; r < nR; r++) {
            double mn = d[r];
            for (int c = 1; c < nC; c++) if (d[r + nR * c] < mn) mn = d[r + nR * c];
            for (int c = 0; c < nC; c++) d[r + nR * c] -= mn;
        }
        for (int r = 0; r < nR; r++)
            for (int c = 0; c < nC; c++)
                if (fabs(d[r + nR * c]) < DBL_EPSILON && !covC[c]) { star[r + nR * c] = 1; covC[c] = 1; break; }
    } else {                                                       /* :103-141 */
        minDim = nC;
        for (int c = 0; c < nC; c++) {
            double mn = d[nR * c];
            for (int r = 1; r < nR; r++) if (d[r + nR * c] < mn) mn = d[r + nR * c];
            for (int r = 0; r < nR; r++) d[r + nR * c] -= mn;
        }
        for (int c = 0; c < nC; c++)
            for (int r = 0; r < nR; r++)
                if (fabs(d[r + nR * c]) < DBL_EPSILON && !covR[r]) { star[r + nR * c] = 1; covC[c] = 1; covR[r] = 1; break; }
        for (int r = 0; r < nR; r++) covR[r] = 0;
    }
    enum { S2A, S2B, S3, S5, DONE } st = S2B;
    while (st != DONE) {
        switch (st) {
        case S2A:                                                  /* :192-213 */
            for (int c = 0; c < nC; c++)
                for (int r = 0; r < nR; r++) if (star[r + nR * c]) { covC[c] = 1; break; }
            st = S2B; break;
        case S2B: {                                                /* :216-237 */
            int n = 0;
            for (int c = 0; c < nC; c++) if (covC[c]) n++;
            st = (n == minDim) ? DONE : S3; break; }
        case S3: {                                                 /* :240-280 */
            int zerosFound = 1, jumped = 0;
            while (zerosFound && !jumped) {
                zerosFound = 0;
                for (int c = 0; c < nC && !jumped; c++) {
                    if (covC[c]) continue;
                    for (int r = 0; r < nR; r++) {
                        if (covR[r] || !(fabs(d[r + nR * c]) < DBL_EPSILON)) continue;
                        prime[r + nR * c] = 1;
                        int sc; for (sc = 0; sc < nC; sc++) if (star[r + nR * sc]) break;
                        if (sc == nC) {                            /* step 4 :283-334 */
                            memcpy(nstar, star, (size_t)nE);
                            nstar[r + nR * c] = 1;
                            int starCol = c, starRow;
                            for (starRow = 0; starRow < nR; starRow++) if (star[starRow + nR * starCol]) break;
                            while (starRow < nR) {
                                nstar[starRow + nR * starCol] = 0;
                                int pr = starRow, pc;
                                for (pc = 0; pc < nC; pc++) if (prime[pr + nR * pc]) break;
                                nstar[pr + nR * pc] = 1;
                                starCol = pc;
                                for (starRow = 0; starRow < nR; starRow++) if (star[starRow + nR * starCol]) break;
                            }
                            memset(prime, 0, (size_t)nE);
                            memcpy(star, nstar, (size_t)nE);
                            memset(covR, 0, (size_t)nR);
                            st = S2A; jumped = 1;
                        } else {
                            covR[r] = 1; covC[sc] = 0; zerosFound = 1;
                        }
                        break; /* leaves the row loop only; the column sweep continues (:273) */
                    }
                }
            }
            if (!jumped) st = S5;
            break; }
        case S5: {                                                 /* :337-368 */
            double h = DBL_MAX;
            for (int r = 0; r < nR; r++) if (!covR[r])
                for (int c = 0; c < nC; c++) if (!covC[c]) { double v = d[r + nR * c]; if (v < h) h = v; }
            for (int r = 0; r < nR; r++) if (covR[r]) for (int c = 0; c < nC; c++) d[r + nR * c] += h;
            for (int c = 0; c < nC; c++) if (!covC[c]) for (int r = 0; r < nR; r++) d[r + nR * c] -= h;
            st = S3; break; }
        default: break;
        }
    }
    for (int r = 0; r < nR; r++)                                    /* :161-176 */
        for (int c = 0; c < nC; c++) if (star[r + nR * c]) { assignment[r] = c; break; }
    for (int r = 0; r < nR; r++) if (assignment[r] >= 0) *cost += distIn[r + nR * assignment[r]]; /* :179-189 */
    free(d); free(covC); free(covR); free(star); free(prime); free(nstar);
}

/* ------------------------------------------------------------------------- */
/* Tracker-thread frame loop (top/td.cpp:306-748)                             */
/* ------------------------------------------------------------------------- */
typedef struct {
    unsigned tid; void* trk; int age, visible, invisible; orc_bbox_t bbox; int rows, cols;
} orc_info;
struct orc_mot { int kind, mode, cap, n; unsigned next_tid; orc_info* info; float* gray; float* scratch; };

orc_mot* orc_mot_new(int kind, int mode, int cap)
{
    orc_mot* m = (orc_mot*)calloc(1, sizeof(orc_mot));
    m->kind = kind; m->mode = mode; m->cap = cap;
    m->info = (orc_info*)calloc((size_t)cap, sizeof(orc_info));
    m->gray = (float*)malloc(sizeof(float) * 1280 * 720);
    m->scratch = (float*)malloc(sizeof(float) * 1280 * 720);
    return m;
}
void orc_mot_delete(orc_mot* m)
{
    if (!m) return;
    for (int i = 0; i < m->n; i++) { if (m->kind == ORC_TRACKER_KCF) orc_kcf_delete((orc_kcf*)m->info[i].trk); else orc_kalman_delete((orc_kalman*)m->info[i].trk); }
    free(m->info); free(m->gray); free(m->scratch); free(m);
}
int orc_mot_ntracks(const orc_mot* m) { return m->n; }
orc_kcf* orc_mot_kcf(orc_mot* m, int i) { return (orc_kcf*)m->info[i].trk; }
orc_kalman* orc_mot_kalman(orc_mot* m, int i) { return (orc_kalman*)m->info[i].trk; }

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

int orc_mot_step(orc_mot* m, const uint8_t* frame, const orc_bbox_t* dets, int nD,
                 orc_bbox_t* predicted, int* assigned_trackers_out, int* n_before,
                 orc_bbox_t* live_boxes, unsigned* live_tids)
{
    const int nT = m->n;
    if (n_before) *n_before = nT;
    /* predict (td.cpp:344-384) */
    for (int i = 0; i < nT; i++) {
        orc_info* t = &m->info[i];
        if (m->kind == ORC_TRACKER_KCF) {
            orc_crop_patch(m->gray, m->scratch, frame, &t->bbox, t->rows, t->cols);
            orc_kcf_predict((orc_kcf*)t->trk, m->gray, &t->bbox);
        } else orc_kalman_predict((orc_kalman*)t->trk, &t->bbox);
        t->bbox.l = clampi(t->bbox.l, 0, 1279); t->bbox.r = clampi(t->bbox.r, 0, 1279);
        t->bbox.t = clampi(t->bbox.t, 0, 719); t->bbox.b = clampi(t->bbox.b, 0, 719);
        if (predicted) predicted[i] = t->bbox;
    }
    /* cost + assignment (td.cpp:386-502) */
    int* at = (int*)malloc(sizeof(int) * (size_t)(nT + 1));
    int* ad = (int*)malloc(sizeof(int) * (size_t)(nD + 1));
    for (int i = 0; i < nT; i++) at[i] = -1;
    for (int j = 0; j < nD; j++) ad[j] = -1;
    if (nT && nD) {
        orc_bbox_t* tb = (orc_bbox_t*)malloc(sizeof(orc_bbox_t) * (size_t)nT);
        for (int i = 0; i < nT; i++) tb[i] = m->info[i].bbox;
        double* dist = (double*)malloc(sizeof(double) * (size_t)nT * nD);
        int* asg = (int*)malloc(sizeof(int) * (size_t)(nT > nD ? nT : nD));
        double cost;
        orc_cost_matrix(tb, nT, dets, nD, dist);
        if (nT < nD) {
            orc_assignment_optimal(asg, &cost, dist, nT, nD);
            for (int i = 0; i < nT; i++) { int j = asg[i]; at[i] = j; if (j >= 0) ad[j] = i; }
        } else {
            orc_assignment_optimal(asg, &cost, dist, nD, nT);
            for (int j = 0; j < nD; j++) { int i = asg[j]; if (i >= 0) at[i] = j; ad[j] = i; }
        }
        free(tb); free(dist); free(asg);
    }
    if (assigned_trackers_out) for (int i = 0; i < nT; i++) assigned_trackers_out[i] = at[i];
    /* update assigned (td.cpp:512-547) */
    for (int i = 0; i < nT; i++) {
        int j = at[i]; if (j < 0) continue;
        orc_info* t = &m->info[i];
        if (m->kind == ORC_TRACKER_KCF) {
            orc_crop_patch(m->gray, m->scratch, frame, &dets[j], t->rows, t->cols);
            orc_kcf_update((orc_kcf*)t->trk, m->gray, &dets[j]);
        } else orc_kalman_update((orc_kalman*)t->trk, &dets[j]);
        t->bbox = dets[j]; t->visible++; t->age++; t->invisible = 0;
    }
    /* update unassigned (td.cpp:550-582) */
    for (int i = 0; i < nT; i++) {
        if (at[i] >= 0) continue;
        orc_info* t = &m->info[i];
        t->age++; t->invisible++;
        if (m->kind == ORC_TRACKER_KCF) {
            orc_crop_patch(m->gray, m->scratch, frame, &t->bbox, t->rows, t->cols);
            orc_kcf_update((orc_kcf*)t->trk, m->gray, &t->bbox);
        } else orc_kalman_update((orc_kalman*)t->trk, &t->bbox);
    }
    /* delete lost (td.cpp:585-609) */
    int n = 0;
    for (int i = 0; i < nT; i++) {
        orc_info* t = &m->info[i];
        int lost = ((t->age < 10) && (t->visible * 5 < 3 * t->age)) || (t->invisible >= 20);
        if (!lost) { if (n != i) m->info[n] = *t; ++n; }
        else { if (m->kind == ORC_TRACKER_KCF) orc_kcf_delete((orc_kcf*)t->trk); else orc_kalman_delete((orc_kalman*)t->trk); }
    }
    m->n = n;
    /* spawn (td.cpp:612-644) */
    for (int j = 0; j < nD; j++) {
        if (ad[j] >= 0) continue;
        if (m->n >= m->cap) break; /* the reference has no bound check (tracker_info[256]); the oracle stops at capacity */
        orc_info* t = &m->info[m->n];
        memset(t, 0, sizeof *t);
        t->tid = m->next_tid++;
        t->bbox = dets[j];
        t->rows = dets[j].b - dets[j].t + 1; t->cols = dets[j].r - dets[j].l + 1;
        if (m->kind == ORC_TRACKER_KCF) {
            t->trk = orc_kcf_new(&dets[j], m->mode);
            orc_rgb2gray(m->gray, frame, dets[j].l, dets[j].t, dets[j].r, dets[j].b);
            orc_kcf_update((orc_kcf*)t->trk, m->gray, &t->bbox);
        } else t->trk = orc_kalman_new(&dets[j]);
        m->n++;
    }
    free(at); free(ad);
    for (int i = 0; i < m->n; i++) { if (live_boxes) live_boxes[i] = m->info[i].bbox; if (live_tids) live_tids[i] = m->info[i].tid; }
    return m->n;
}

/* ---------------------------------------------------------------------------
 * overlay: drawRect (top/drawlib.c:97-151), hashcolor (top/td.cpp:295-304), the colour table and the drawing loop of the
 * tracker thread (td.cpp:647-733).  Pixels outside the 1280 x 720 frame are skipped (the reference writes unchecked).
 * ------------------------------------------------------------------------- */
#define ORC_FRAME_W 1280   /* top/cnntype.h:5-6 */
#define ORC_FRAME_H 720
static void orc_put(uint8_t* fbuf, int y, int x, uint8_t R, uint8_t G, uint8_t B)
{
    if (x < 0 || x >= ORC_FRAME_W || y < 0 || y >= ORC_FRAME_H) return;
    uint8_t* p = fbuf + ((size_t)y * ORC_FRAME_W + x) * 3;            /* PIXEL_AT, drawlib.c:9 */
    p[0] = R; p[1] = G; p[2] = B;                                      /* drawlib.c:136,147: R first, whatever the frame's channel order */
}

void orc_draw_rect(uint8_t* fbuf, int left, int top, int right, int bottom, uint32_t rgb)
{
    const uint8_t R = (rgb >> 16) & 0xff, G = (rgb >> 8) & 0xff, B = rgb & 0xff;   /* :106-108 */
    if (top > bottom) { int t = top; top = bottom; bottom = t; }     /* :112-124 */
    if (left > right) { int t = left; left = right; right = t; }
    for (int x = left; x <= right; x++) { orc_put(fbuf, top, x, R, G, B); orc_put(fbuf, bottom, x, R, G, B); }   /* :132-137 */
    for (int y = top; y <= bottom; y++) { orc_put(fbuf, y, left, R, G, B); orc_put(fbuf, y, right, R, G, B); }   /* :145-151 */
}

uint32_t orc_hashcolor(uint32_t a)
{   /* td.cpp:295-304 */
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

const uint32_t* orc_colormap(void)
{   /* td.cpp:655-697: the 16 system colours, the 6 x 6 x 6 cube on levels {00,5f,87,af,d7,ff}, 24 greys 08 + 10 k -- with the two
     * greys the reference spells 0x606060 and 0x666666 (entries 241, 242) */
    static uint32_t map[256]; static int done = 0;
    if (!done) {
        static const uint32_t sys[16] = { 0x000000, 0x800000, 0x008000, 0x808000, 0x000080, 0x800080, 0x008080, 0xc0c0c0,
                                          0x808080, 0xff0000, 0x00ff00, 0xffff00, 0x0000ff, 0xff00ff, 0x00ffff, 0xffffff };
        static const uint32_t lv[6] = { 0x00, 0x5f, 0x87, 0xaf, 0xd7, 0xff };
        for (int i = 0; i < 16; i++) map[i] = sys[i];
        for (int r = 0; r < 6; r++) for (int g = 0; g < 6; g++) for (int b = 0; b < 6; b++) map[16 + 36 * r + 6 * g + b] = (lv[r] << 16) | (lv[g] << 8) | lv[b];
        for (int k = 0; k < 24; k++) { const uint32_t v = 8 + 10 * k; map[232 + k] = (v << 16) | (v << 8) | v; }
        map[241] = 0x606060; map[242] = 0x666666;
        done = 1;
    }
    return map;
}

void orc_overlay(uint8_t* frame, const orc_bbox_t* boxes, const unsigned* tids, int n)
{
    const uint32_t* cm = orc_colormap();
    for (int j = 0; j < n; j++) {                                      /* td.cpp:647-733 */
        const uint32_t color = cm[orc_hashcolor(tids[j] + 1u) & 255];  /* td.cpp:619-620: tid = tracker_id++; color = hashcolor(tracker_id) & 255, i.e. of tid + 1; :699 */
        const orc_bbox_t b = boxes[j];
        orc_draw_rect(frame, b.l, b.t, b.r, b.b, color);
        orc_draw_rect(frame, b.l + 1, b.t + 1, b.r - 1, b.b - 1, color);
        orc_draw_rect(frame, b.l + 2, b.t + 2, b.r - 2, b.b - 2, color);
    }
}

/* ---------------------------------------------------------------------------
 * detector post-processing (detectors/yolo3.cpp:141-356, 490-530), float arithmetic as written there.  PARITY UNPINNED (see header).
 * ------------------------------------------------------------------------- */
typedef struct { float x, y, u, w; int c; float s; } orc_pre;          /* predecode_t, yolo3.cpp:95-99 */
typedef struct { int xmin, ymin, xmax, ymax, classes; float objectness; } orc_det;   /* detection_t, :101-108 */

static int orc_decode(orc_pre* boxes, int n, int cap, const float* out, const int* anchors, float obj_thresh, int tensor_h, int tensor_w,
                      int grid_h, int grid_w, int nc)
{   /* decode_netout :141-201 */
    const int per = 5 + nc, nb_box = 3 * per;
    for (int i = 0; i < grid_h * grid_w; i++) {
        const int row = i / grid_w, col = i % grid_w;
        const float* cell = out + (size_t)i * nb_box;
        for (int b = 0; b < 3; b++) {
            const float* v = cell + b * per;
            const float objectness = 1 / (1 + expf(-v[4]));
            for (int j = 0; j < nc; j++) {
                const float scores = (1 / (1 + expf(-v[5 + j]))) * objectness;
                if (scores >= obj_thresh && n < cap) {
                    orc_pre p;
                    p.x = (col + (1 / (1 + expf(-v[0])))) / grid_w;
                    p.y = (row + (1 / (1 + expf(-v[1])))) / grid_h;
                    p.u = anchors[2 * b + 0] * expf(v[2]) / tensor_w;
                    p.w = anchors[2 * b + 1] * expf(v[3]) / tensor_h;
                    p.s = scores; p.c = j;
                    boxes[n++] = p;
                }
            }
        }
    }
    return n;
}

int orc_yolo_postprocess(const float* head0, const float* head1, const float* head2, int tensor_h, int tensor_w, int nc,
                         int image_h, int image_w, const orc_yolo_opt* opt, orc_bbox_t* out, int cap)
{
    const int CAND = 16384;
    orc_pre* pre = (orc_pre*)malloc(sizeof(orc_pre) * CAND);
    orc_det* cb = (orc_det*)malloc(sizeof(orc_det) * CAND);
    orc_det* cls = (orc_det*)malloc(sizeof(orc_det) * CAND);
    int* idx = (int*)malloc(sizeof(int) * CAND); int* sup = (int*)malloc(sizeof(int) * CAND);
    const int grid_h = tensor_h / 32, grid_w = tensor_w / 32;            /* :405-406 */
    int n = 0;
    n = orc_decode(pre, n, CAND, head0, opt->anchors + 12, opt->obj_thresh, tensor_h, tensor_w, grid_h << 0, grid_w << 0, nc);   /* :512-514 */
    n = orc_decode(pre, n, CAND, head1, opt->anchors + 6, opt->obj_thresh, tensor_h, tensor_w, grid_h << 1, grid_w << 1, nc);
    n = orc_decode(pre, n, CAND, head2, opt->anchors + 0, opt->obj_thresh, tensor_h, tensor_w, grid_h << 2, grid_w << 2, nc);
    /* correct_yolo_boxes :203-254 (an empty list yields one zero box there; it is dropped again by the emit loop's class filter) */
    float new_w, new_h;
    if (((float)tensor_w / (float)image_w) < ((float)tensor_h / (float)image_h)) { new_w = (float)tensor_w; new_h = roundf((float)image_h * tensor_w / (float)image_w); }
    else { new_h = (float)tensor_h; new_w = roundf((float)image_w * tensor_h / (float)image_h); }
    for (int i = 0; i < n; i++) {
        const float x_offset = (float)((tensor_w - new_w) / 2.0 / tensor_w), x_scale = (float)new_w / tensor_w;
        const float y_offset = (float)((tensor_h - new_h) / 2.0 / tensor_h), y_scale = (float)new_h / tensor_h;
        const float x = (pre[i].x - x_offset) / x_scale * (float)image_w, y = (pre[i].y - y_offset) / y_scale * (float)image_h;
        const float w = (pre[i].u) / x_scale * (float)image_w, h = (pre[i].w) / y_scale * (float)image_h;
        cb[i].xmin = (int)(x - w / 2); cb[i].xmax = (int)(x + w / 2); cb[i].ymin = (int)(y - h / 2); cb[i].ymax = (int)(y + h / 2);
        cb[i].objectness = pre[i].s; cb[i].classes = pre[i].c;
    }
    /* do_nms :279-356 + emit :519-547.  Quirk kept: `is_suppressed` (:286) is declared OUTSIDE the class loop and only ever grows by
     * push_back(0), while it is indexed with the class-local indices 0..m-1 -- so a flag set while an earlier class was processed is
     * still set for the box with the same local index of a later class. */
    int nout = 0;
    for (int i = 0; i < CAND; i++) sup[i] = 0;
    for (int c = 0; c < nc; c++) {
        int m = 0;
        for (int j = 0; j < n; j++) if (cb[j].classes == c) cls[m++] = cb[j];
        for (int i = 0; i < m; i++) idx[i] = i;
        for (int i = 0; i < m; i++)                                     /* sort :256-277: in-place exchange, strict '>' */
            for (int j = i + 1; j < m; j++)
                if (cls[idx[j]].objectness > cls[idx[i]].objectness) { const int t = idx[i]; idx[i] = idx[j]; idx[j] = t; }
        for (int i = 0; i < m; i++) {
            if (sup[idx[i]]) continue;
            for (int j = i + 1; j < m; j++) {
                const orc_det* A = &cls[idx[j]]; const orc_det* B = &cls[idx[i]];
                const float maxX = (float)(A->xmax < B->xmax ? A->xmax : B->xmax), maxY = (float)(A->ymax < B->ymax ? A->ymax : B->ymax);
                const float minX = (float)(A->xmin > B->xmin ? A->xmin : B->xmin), minY = (float)(A->ymin > B->ymin ? A->ymin : B->ymin);
                const float oW = maxX - minX + 1, oH = maxY - minY + 1;
                if ((oW > 0) & (oH > 0)) {
                    const float a1 = (float)((A->xmax - A->xmin + 1) * (A->ymax - A->ymin + 1)), a2 = (float)((B->xmax - B->xmin + 1) * (B->ymax - B->ymin + 1));
                    const float iou = (oW * oH) / (a1 + a2 - oW * oH);
                    if (iou > opt->nms_thresh) sup[idx[j]] = 1;
                }
            }
        }
        for (int i = 0; i < m; i++) {
            if (sup[idx[i]]) continue;
            orc_det d = cls[idx[i]];
            d.ymin = d.ymin > 0 ? d.ymin : 0; d.xmin = d.xmin > 0 ? d.xmin : 0;                     /* :523-526 */
            d.ymax = d.ymax < image_h - 1 ? d.ymax : image_h - 1; d.xmax = d.xmax < image_w - 1 ? d.xmax : image_w - 1;
            if (d.ymin > d.ymax || d.xmin > d.xmax || d.ymin < 0 || d.xmin < 0 || d.xmax >= image_w || d.ymax >= image_h) continue;   /* :528-536 */
            if (nout < cap) { out[nout].t = d.ymin; out[nout].l = d.xmin; out[nout].b = d.ymax; out[nout].r = d.xmax; out[nout].type = d.classes; out[nout].score = d.objectness; nout++; }
        }
    }
    free(pre); free(cb); free(cls); free(idx); free(sup);
    return nout;
}
