// event_scope_probe.hip -- does a consumer kernel on stream B, ordered behind a producer kernel on stream A by an event, always see the producer's
// writes?  Round 5: the device loop's detection spectra go from the side stream's feature launch to the main stream's predict launch exactly this
// way, and a soak found about one run in 10^4 in which a few dozen tracks had blended stale spectra -- always where the consumer starts the moment
// the producer ends (frames 0 -> 1).  This probe isolates the mechanism from the library: per iteration the producer writes a fresh pattern
// to a buffer (default 56 MB: one spectra buffer), the consumer checks it and counts stale words; event flags and stream kinds are parameters.
//   hipcc -O3 --offload-arch=gfx950 -o tools/event_scope_probe tools/event_scope_probe.hip && ./tools/event_scope_probe [iters] [MB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void __launch_bounds__(512) produce(unsigned* buf, size_t n, unsigned pat)
{   // one workgroup per 55 KB "detection", like the feature launch: coalesced 8-byte stores
    const size_t per = 13640;                                         // words per item (31 * 220 complex)
    const size_t base = (size_t)blockIdx.x * per;
    for (size_t i = threadIdx.x; i < per && base + i < n; i += blockDim.x) buf[base + i] = pat ^ (unsigned)(base + i);
}
__global__ void __launch_bounds__(512) consume(const unsigned* buf, size_t n, unsigned pat, unsigned* bad, unsigned* first_bad)
{
    const size_t per = 13640;
    const size_t base = (size_t)blockIdx.x * per;
    unsigned nb = 0;
    for (size_t i = threadIdx.x; i < per && base + i < n; i += blockDim.x) if (buf[base + i] != (pat ^ (unsigned)(base + i))) nb++;
    if (nb) { atomicAdd(bad, nb); atomicMin(first_bad, (unsigned)blockIdx.x); }
}
__global__ void __launch_bounds__(256) hammer(uint4* buf, size_t n16, unsigned salt)
{   // a third stream keeps the chip and the memory system busy (the soak's second context)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = buf[i]; v.x += salt; buf[i] = v; }
}
__global__ void spin(unsigned* sink, int n) { unsigned v = 0; for (int i = 0; i < n; i++) v = v * 1664525u + 1013904223u; if (v == 12345u) *sink = v; }

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const size_t mb = argc > 2 ? (size_t)atoi(argv[2]) : 56;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;                    // 1: frames 0 -> 1 pattern -- the producer stream runs a SECOND producer (another buffer) right behind the
                                                                      // first, so the consumer executes beside it; and a third stream hammers the memory system
    const size_t n = mb * 1024 * 1024 / 4;
    const int grid = (int)((n + 13639) / 13640);
    unsigned *buf, *bad, *sink; unsigned* first_bad;
    CHK(hipMalloc((void**)&buf, n * 4)); CHK(hipMalloc((void**)&bad, 4)); CHK(hipMalloc((void**)&first_bad, 4)); CHK(hipMalloc((void**)&sink, 4));
    CHK(hipMemset(buf, 0, n * 4));
    unsigned* buf2 = nullptr; uint4* hbuf = nullptr; hipStream_t sh = nullptr;
    if (mode) { CHK(hipMalloc((void**)&buf2, n * 4)); CHK(hipMalloc((void**)&hbuf, (size_t)256 << 20)); CHK(hipMemset(hbuf, 0, (size_t)256 << 20)); CHK(hipStreamCreateWithFlags(&sh, hipStreamNonBlocking)); }
    struct Variant { const char* name; unsigned flags; bool masked; };
    const Variant vs[] = { { "event: DisableTiming|ReleaseToDevice, producer on a CU-masked stream", hipEventDisableTiming | hipEventReleaseToDevice, true },
                           { "event: DisableTiming|ReleaseToDevice, producer on a plain stream", hipEventDisableTiming | hipEventReleaseToDevice, false },
                           { "event: DisableTiming (system-scope release), producer on a CU-masked stream", hipEventDisableTiming, true },
                           { "event: default flags, producer on a plain stream", 0u, false } };
    for (const Variant& v : vs) {
        hipStream_t sa, sb; hipEvent_t ev, back;
        if (v.masked) { uint32_t mask[8]; for (int w = 0; w < 8; w++) mask[w] = 0xFFFFFFFFu; mask[0] = 0; CHK(hipExtStreamCreateWithCUMask(&sa, 8, mask)); }
        else CHK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
        CHK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
        CHK(hipEventCreateWithFlags(&ev, v.flags)); CHK(hipEventCreateWithFlags(&back, hipEventDisableTiming | hipEventReleaseToDevice));
        CHK(hipMemset(bad, 0, 4)); CHK(hipMemset(first_bad, 0xFF, 4));
        unsigned total_bad = 0; int bad_iters = 0;
        for (int it = 0; it < iters; it++) {
            const unsigned pat = 0x9E3779B9u * (unsigned)(it + 1);
            if (it) CHK(hipStreamWaitEvent(sa, back, 0));               // the producer may overwrite the buffer only behind the previous check
            hipLaunchKernelGGL(produce, dim3(grid), dim3(512), 0, sa, buf, n, pat);
            CHK(hipEventRecord(ev, sa));
            if (mode) {
                hipLaunchKernelGGL(produce, dim3(grid), dim3(512), 0, sa, buf2, n, ~pat);          // the look-ahead launch: runs beside the consumer
                if ((it & 3) == 0) hipLaunchKernelGGL(hammer, dim3(1024), dim3(256), 0, sh, hbuf, ((size_t)256 << 20) / 16, (unsigned)it);
            }
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, sb, sink, 200 + (it % 7) * 300);   // something short in front of the wait, as the chain is
            CHK(hipStreamWaitEvent(sb, ev, 0));
            hipLaunchKernelGGL(consume, dim3(grid), dim3(512), 0, sb, buf, n, pat, bad, first_bad);
            CHK(hipEventRecord(back, sb));
            if ((it & 255) == 255) {
                CHK(hipStreamSynchronize(sb));
                unsigned b; CHK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
                if (b != total_bad) { bad_iters++; total_bad = b; }
            }
        }
        CHK(hipStreamSynchronize(sb)); CHK(hipStreamSynchronize(sa));
        unsigned b, fb; CHK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&fb, first_bad, 4, hipMemcpyDeviceToHost));
        if (mode) CHK(hipStreamSynchronize(sh));
        printf("%-84s iterations %d  stale words %u  (batches of 256 iterations with stale words: %d, lowest item %d)\n", v.name, iters, b, bad_iters + (b != total_bad), b ? (int)fb : -1);
        CHK(hipEventDestroy(ev)); CHK(hipEventDestroy(back)); CHK(hipStreamDestroy(sa)); CHK(hipStreamDestroy(sb));
    }
    return 0;
}
