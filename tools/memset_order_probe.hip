// memset_order_probe.hip -- is hipMemset() of device memory synchronous to the host, and how is it ordered against a NON-BLOCKING stream?
// Round 5: the device loop's set-up filled its pending-detection array with hipMemset(ptr, 0xFF, ...) and then started enqueuing frames on the
// context's non-blocking stream; about one run in 10^4 behaved exactly as if that fill had executed AFTER the first frame's lifecycle kernel.
//   hipcc -O3 --offload-arch=gfx950 -o tools/memset_order_probe tools/memset_order_probe.hip && ./tools/memset_order_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
__global__ void spin(unsigned long long ticks, unsigned* sink) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) { } if (ticks == 1) *sink = 1; }
__global__ void store(int* p, int n, int v) { for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = v; }
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    int* buf; unsigned* sink; CHK(hipMalloc((void**)&buf, 4096)); CHK(hipMalloc((void**)&sink, 4));
    hipStream_t nb; CHK(hipStreamCreateWithFlags(&nb, hipStreamNonBlocking));
    CHK(hipDeviceSynchronize());
    // (1) host-side duration of hipMemset on an idle device, 4 KB and 64 MB
    for (size_t bytes : { (size_t)4096, (size_t)64 << 20 }) {
        int* big; CHK(hipMalloc((void**)&big, bytes)); CHK(hipDeviceSynchronize());
        const double t0 = now_ms(); CHK(hipMemset(big, 0xFF, bytes)); const double t1 = now_ms(); CHK(hipDeviceSynchronize()); const double t2 = now_ms();
        printf("idle device: hipMemset(%zu bytes) returned after %.3f ms, the device was idle %.3f ms later\n", bytes, t1 - t0, t2 - t1);
        CHK(hipFree(big));
    }
    // (2) the null stream is busy for 50 ms: does hipMemset wait for it?  is a non-blocking stream's kernel ordered behind the fill?
    int late = 0, runs = 20;
    for (int r = 0; r < runs; r++) {
        CHK(hipMemset(buf, 0, 4096)); CHK(hipDeviceSynchronize());
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 5000000ull, sink);        // 50 ms at 100 MHz on the NULL stream
        const double t0 = now_ms(); CHK(hipMemset(buf, 0xFF, 4096)); const double t1 = now_ms();
        hipLaunchKernelGGL(store, dim3(1), dim3(256), 0, nb, buf, 1024, 7);        // the "lifecycle step": writes 7 on the non-blocking stream
        CHK(hipStreamSynchronize(nb)); const double t2 = now_ms();
        CHK(hipDeviceSynchronize());
        int h[4]; CHK(hipMemcpy(h, buf, sizeof h, hipMemcpyDeviceToHost));
        if (r == 0) printf("busy null stream: hipMemset(4096) returned after %.3f ms; the non-blocking stream's kernel was done %.3f ms after that; final word %d (7 = kernel last, -1 = the FILL landed last)\n", t1 - t0, t2 - t1, h[0]);
        late += h[0] == -1;
    }
    printf("busy null stream: the fill overwrote the non-blocking stream's later write in %d of %d runs\n", late, runs);
    return 0;
}
