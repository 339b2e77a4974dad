// overlay_kernels.hip -- the tracker thread's drawing step on device: three nested rectangle outlines per live
// track (top/td.cpp:647-733) drawn by drawRect (top/drawlib.c:97-151) in colormap[hashcolor(tid + 1) & 255] (td.cpp:295-304, 620,
// 655-699), straight into the BGR frame in HBM.
//
// The reference draws track after track, so where outlines overlap the LAST track wins.  All three outlines of a track have
// one colour, so only the track order matters: pass 1 stamps every outline pixel with max(track index) (atomicMax on a
// per-pixel word, tagged with a per-call epoch so the buffer is never cleared), pass 2 paints a pixel from the track whose
// stamp it carries.  One workgroup per track, threads along the three perimeters.  drawRect's quirks are kept: corners are
// normalised by swapping (a 3 x 3 box's innermost outline has left > right), and the bytes R, G, B of the colour go to memory
// offsets 0, 1, 2 of the pixel whatever the frame's channel order.  Pixels outside the frame are skipped (the reference writes
// unchecked).
#include "mot_ctx.h"

namespace {

__device__ __forceinline__ unsigned hashcolor(unsigned a)
{   // td.cpp:295-304
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

__device__ __forceinline__ unsigned colormap(unsigned i)
{   // td.cpp:655-697: 16 system colours, the 6 x 6 x 6 cube, 24 greys -- with the reference's own spelling of entries 241 and 242
    i &= 255u;
    if (i < 16) {
        const unsigned sys[16] = { 0x000000, 0x800000, 0x008000, 0x808000, 0x000080, 0x800080, 0x008080, 0xc0c0c0,
                                   0x808080, 0xff0000, 0x00ff00, 0xffff00, 0x0000ff, 0xff00ff, 0x00ffff, 0xffffff };
        return sys[i];
    }
    if (i < 232) {
        const unsigned lv[6] = { 0x00, 0x5f, 0x87, 0xaf, 0xd7, 0xff };
        const unsigned k = i - 16, r = k / 36, g = (k / 6) % 6, b = k % 6;
        return (lv[r] << 16) | (lv[g] << 8) | lv[b];
    }
    if (i == 241) return 0x606060;
    if (i == 242) return 0x666666;
    const unsigned v = 8 + 10 * (i - 232);
    return (v << 16) | (v << 8) | v;
}

// visits the outline pixels of drawRect(left, top, right, bottom): f(y, x)
template <typename F>
__device__ __forceinline__ void outline(int left, int top, int right, int bottom, int tid, int nt, F&& f)
{
    if (top > bottom) { const int t = top; top = bottom; bottom = t; }  // drawlib.c:112-124
    if (left > right) { const int t = left; left = right; right = t; }
    const int w = right - left + 1, h = bottom - top + 1;
    for (int i = tid; i < 2 * w + 2 * h; i += nt) {
        int y, x;
        if (i < w) { y = top; x = left + i; }
        else if (i < 2 * w) { y = bottom; x = left + (i - w); }
        else if (i < 2 * w + h) { y = top + (i - 2 * w); x = left; }
        else { y = top + (i - 2 * w - h); x = right; }
        if ((unsigned)x < (unsigned)MOT_FRAME_W && (unsigned)y < (unsigned)MOT_FRAME_H) f(y, x);
    }
}

template <int PASS>
__global__ void __launch_bounds__(256) overlay_kernel(uint8_t* __restrict__ frame, unsigned* __restrict__ stamp, const bbox_t* __restrict__ boxes,
                                                      const unsigned* __restrict__ tids, const int* __restrict__ n_dev, int n, unsigned epoch)
{
    const int j = blockIdx.x;
    if (j >= (n_dev ? *n_dev : n)) return;
    const bbox_t b = boxes[j];
    const unsigned prio = (epoch << 11) | (unsigned)(j + 1);
    const unsigned color = colormap(hashcolor(tids[j] + 1u));            // td.cpp:619-620: tid = tracker_id++; color = hashcolor(tracker_id) -- the id AFTER the increment
    const uint8_t R = (color >> 16) & 0xff, G = (color >> 8) & 0xff, B = color & 0xff;
#pragma unroll
    for (int k = 0; k < 3; k++) {                                      // td.cpp:701-731
        outline(b.l + k, b.t + k, b.r - k, b.b - k, threadIdx.x, blockDim.x, [&](int y, int x) {
            const size_t px = (size_t)y * MOT_FRAME_W + x;
            if (PASS == 0) atomicMax(&stamp[px], prio);
            else if (stamp[px] == prio) { uint8_t* p = frame + px * 3; p[0] = R; p[1] = G; p[2] = B; }   // drawlib.c:136,147
        });
    }
}

} // namespace

namespace mot_impl {
int overlay_run(mot_ctx* c, void* frame_dev, const bbox_t* boxes_dev, const unsigned* tids_dev, const int* n_dev, int n_max)
{
    if (n_max <= 0) return MOT_OK;
    const size_t npx = (size_t)MOT_FRAME_W * MOT_FRAME_H;
    if (!c->ov_stamp.p) { HIPCHK(c->ov_stamp.alloc(npx)); HIPCHK(hipMemsetAsync(c->ov_stamp.p, 0, npx * sizeof(unsigned), c->stream)); c->ov_epoch = 0; }
    if (++c->ov_epoch >= (1u << 21)) { HIPCHK(hipMemsetAsync(c->ov_stamp.p, 0, npx * sizeof(unsigned), c->stream)); c->ov_epoch = 1; }
    hipLaunchKernelGGL(overlay_kernel<0>, dim3(n_max), dim3(256), 0, c->stream, (uint8_t*)frame_dev, c->ov_stamp.p, boxes_dev, tids_dev, n_dev, n_max, c->ov_epoch);
    hipLaunchKernelGGL(overlay_kernel<1>, dim3(n_max), dim3(256), 0, c->stream, (uint8_t*)frame_dev, c->ov_stamp.p, boxes_dev, tids_dev, n_dev, n_max, c->ov_epoch);
    HIPCHK(hipGetLastError());
    return MOT_OK;
}
} // namespace mot_impl
