// helper_kernels.hip -- the three C helpers of the reference's tracker thread (top/td.cpp:235-261, implemented there by top/drawlib.c)
// on the device, behind the reference's own signatures (host pointers in, host pointers out), so that a td.cpp link needs no reference
// object besides the drop-in library (csrc/dropin.cpp exports rgb2Gray / bilinearInterpolationGray / drawRect on top of these):
//   * rgb2Gray (drawlib.c:192-240): BGR bytes of a box -> gray floats, COLUMN-major (pgra[c * rows + r]), 0.144 B + 0.587 G + 0.299 R in
//     double, one rounding to float;
//   * bilinearInterpolationGray (drawlib.c:542-637): row-major bilinear resize with float weights -- td.cpp:357-364 feeds it the
//     column-major patch with (rows, cols) in the (height, width) places, the reference's row / column quirk, reproduced by the caller;
//   * drawRect (drawlib.c:97-151): one rectangle outline, bytes R, G, B at offsets 0, 1, 2 of a pixel.
// Per call: the touched bytes go up, one small kernel runs, the result comes back (the per-object interface is a compatibility layer, not
// the fast path: the batch / device-resident entry points fuse crop, gray and resize into the KCF kernels and never leave the device).
// Compiled with -ffp-contract=off (Makefile): every multiply and add rounds separately, like the reference's scalar code.
#include "mot_ctx.h"

using namespace mot_impl;

namespace {

// (debug, MOT_LDS_POISON) every compute unit's LDS is overwritten with `word`: 160 KB per workgroup = one workgroup per CU at a time, four
// rounds over the 256 CUs; each workgroup lingers a few microseconds so that the dispatcher spreads the round over the chip
__global__ void __launch_bounds__(256) lds_poison_kernel(unsigned word)
{
    extern __shared__ __attribute__((aligned(16))) unsigned lds_poison_smem[];
    volatile unsigned* q = lds_poison_smem;
    for (int i = threadIdx.x; i < (int)(MOT_LDS_LIMIT / 4); i += 256) q[i] = word;
    __syncthreads();
    for (int k = 0; k < 4; k++) __builtin_amdgcn_s_sleep(127);
}

__global__ void __launch_bounds__(256) helper_gray_kernel(const uint8_t* __restrict__ crop, float* __restrict__ out, int rows, int cols)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols) return;
    const int r = i / cols, c = i - r * cols;
    const uint8_t* p = crop + ((size_t)r * cols + c) * 3;
    const uint8_t B = p[0], G = p[1], R = p[2];                       // drawlib.c:230-232
    out[(size_t)c * rows + r] = (float)(0.144 * B + 0.587 * G + 0.299 * R);   // :234-235, pdst[r], pdst += rows
}

__global__ void __launch_bounds__(256) helper_bilinear_kernel(const float* __restrict__ src, float* __restrict__ dst, int hs, int ws, int h, int w)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    const float xs = ((float)ws) / ((float)w), ys = ((float)hs) / ((float)h);   // drawlib.c:551-552
    const float sx = x * xs; const int x0 = (int)sx;                  // :574-575
    const float fracx = sx - x0, ifracx = 1.0f - fracx;               // :578-579
    int x1 = x0 + 1; if (x1 >= ws) x1 = x0;                            // :581-585
    const float sy = y * ys; const int y0 = (int)sy;                  // :600-601
    const float fracy = sy - y0, ifracy = 1.0f - fracy;
    int y1 = y0 + 1; if (y1 >= hs) y1 = y0;
    const float c1 = src[y0 * ws + x0], c2 = src[y0 * ws + x1], c3 = src[y1 * ws + x0], c4 = src[y1 * ws + x1];   // :623-626
    const float l0 = ifracx * c1 + fracx * c2, l1 = ifracx * c3 + fracx * c4;   // :630-631
    dst[i] = ifracy * l0 + fracy * l1;                                // :632-634
}

// the outline's pixels in a compact strip layout: [top row | bottom row | left column | right column], 3 bytes each
__global__ void __launch_bounds__(256) helper_rect_kernel(uint8_t* __restrict__ strips, int n_px, uint8_t R, uint8_t G, uint8_t B)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_px) return;
    strips[i * 3 + 0] = R; strips[i * 3 + 1] = G; strips[i * 3 + 2] = B;   // drawlib.c:136-137, 147-148
}

} // namespace

namespace mot_impl {
hipError_t lds_poison_launch(hipStream_t s, unsigned word)
{
    hipError_t e = func_lds_once(reinterpret_cast<const void*>(lds_poison_kernel), MOT_LDS_LIMIT); if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lds_poison_kernel, dim3(1024), dim3(256), MOT_LDS_LIMIT, s, word);
    return hipGetLastError();
}
}

extern "C" {

int mot_helper_rgb2gray(mot_ctx* c, float* pgra, const uint8_t* prgb, int left, int top, int right, int bottom)
{
    if (!c || !pgra || !prgb) return fail(MOT_ERR_ARG, "null argument");
    int rc = ensure_device(c); if (rc) return rc;
    if (top > bottom) { const int t = top; top = bottom; bottom = t; }  // drawlib.c:203-215
    if (left > right) { const int t = left; left = right; right = t; }
    if (left < 0 || top < 0 || right >= MOT_FRAME_W || bottom >= MOT_FRAME_H) return fail(MOT_ERR_ARG, "box outside the %d x %d frame (the reference reads unchecked)", MOT_FRAME_W, MOT_FRAME_H);
    const int rows = bottom - top + 1, cols = right - left + 1;
    const size_t nb = (size_t)rows * cols * 3, nf = (size_t)rows * cols;
    if (c->hlp_bytes.n < nb) HIPCHK(c->hlp_bytes.alloc(nb));
    if (c->hlp_f0.n < nf) HIPCHK(c->hlp_f0.alloc(nf));
    HIPCHK(hipMemcpy2DAsync(c->hlp_bytes.p, (size_t)cols * 3, prgb + ((size_t)top * MOT_FRAME_W + left) * 3, (size_t)MOT_FRAME_W * 3, (size_t)cols * 3, rows, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(helper_gray_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, c->stream, c->hlp_bytes.p, c->hlp_f0.p, rows, cols);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(pgra, c->hlp_f0.p, nf * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MOT_OK;
}

int mot_helper_bilinear_gray(mot_ctx* c, float* pdst, const float* psrc, int rows_s, int cols_s, int rows_d, int cols_d)
{
    if (!c || !pdst || !psrc || rows_s < 1 || cols_s < 1 || rows_d < 1 || cols_d < 1) return fail(MOT_ERR_ARG, "bad argument");
    int rc = ensure_device(c); if (rc) return rc;
    const size_t ns = (size_t)rows_s * cols_s, nd = (size_t)rows_d * cols_d;
    if (c->hlp_f0.n < ns) HIPCHK(c->hlp_f0.alloc(ns));
    if (c->hlp_f1.n < nd) HIPCHK(c->hlp_f1.alloc(nd));
    HIPCHK(hipMemcpyAsync(c->hlp_f0.p, psrc, ns * sizeof(float), hipMemcpyHostToDevice, c->stream));
    // the reference's parameter names are (heightSource, widthSource, height, width): whatever the caller puts there is taken as such
    hipLaunchKernelGGL(helper_bilinear_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, c->stream, c->hlp_f0.p, c->hlp_f1.p, rows_s, cols_s, rows_d, cols_d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(pdst, c->hlp_f1.p, nd * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MOT_OK;
}

int mot_helper_draw_rect(mot_ctx* c, uint8_t* fbuf, int left, int top, int right, int bottom, unsigned RGB)
{
    if (!c || !fbuf) return fail(MOT_ERR_ARG, "null argument");
    int rc = ensure_device(c); if (rc) return rc;
    if (top > bottom) { const int t = top; top = bottom; bottom = t; }  // drawlib.c:112-124
    if (left > right) { const int t = left; left = right; right = t; }
    if (left < 0 || top < 0 || right >= MOT_FRAME_W || bottom >= MOT_FRAME_H) return fail(MOT_ERR_ARG, "rectangle outside the %d x %d frame (the reference writes unchecked)", MOT_FRAME_W, MOT_FRAME_H);
    const int w = right - left + 1, h = bottom - top + 1, n_px = 2 * w + 2 * h;
    if (c->hlp_bytes.n < (size_t)n_px * 3) HIPCHK(c->hlp_bytes.alloc((size_t)n_px * 3));
    hipLaunchKernelGGL(helper_rect_kernel, dim3((unsigned)((n_px + 255) / 256)), dim3(256), 0, c->stream, c->hlp_bytes.p, n_px, (uint8_t)((RGB >> 16) & 0xff), (uint8_t)((RGB >> 8) & 0xff), (uint8_t)(RGB & 0xff));
    HIPCHK(hipGetLastError());
    uint8_t* d = c->hlp_bytes.p;
    const size_t pitch = (size_t)MOT_FRAME_W * 3;
    HIPCHK(hipMemcpyAsync(fbuf + ((size_t)top * MOT_FRAME_W + left) * 3, d, (size_t)w * 3, hipMemcpyDeviceToHost, c->stream));                    // top row    (:133-138)
    HIPCHK(hipMemcpyAsync(fbuf + ((size_t)bottom * MOT_FRAME_W + left) * 3, d + (size_t)w * 3, (size_t)w * 3, hipMemcpyDeviceToHost, c->stream));   // bottom row
    HIPCHK(hipMemcpy2DAsync(fbuf + ((size_t)top * MOT_FRAME_W + left) * 3, pitch, d + (size_t)2 * w * 3, 3, 3, h, hipMemcpyDeviceToHost, c->stream));              // left column (:144-150)
    HIPCHK(hipMemcpy2DAsync(fbuf + ((size_t)top * MOT_FRAME_W + right) * 3, pitch, d + (size_t)(2 * w + h) * 3, 3, 3, h, hipMemcpyDeviceToHost, c->stream));       // right column
    HIPCHK(hipStreamSynchronize(c->stream));
    return MOT_OK;
}

} // extern "C"
