// ubench_latency.hip -- latency of the primitives the one-workgroup association kernels are built from (gfx950), measured with
// s_memrealtime (100 MHz) over N dependent repetitions.  build: hipcc -O3 --offload-arch=gfx950 -o ubench_latency tools/ubench_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define N 2000
__device__ __forceinline__ long long rt() { return wall_clock64(); }

__global__ void __launch_bounds__(1024) k_lat(long long* out, int* gmem, int nwaves_active)
{
    __shared__ int chase[1024];
    __shared__ unsigned long long slot;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    chase[tid] = (tid * 7 + 13) & 1023;
    if (tid == 0) slot = ~0ull;
    __syncthreads();
    long long t0, t1; int acc = 0;
    // 1. s_memrealtime, dependent
    if (wave == 0) { t0 = rt(); long long x = 0; for (int i = 0; i < N; i++) { x += rt() & 1; } t1 = rt(); if (lane == 0) { out[0] = t1 - t0; out[20] = x; } }
    __syncthreads();
    // 2. s_memtime
    if (wave == 0) { t0 = rt(); long long x = 0; for (int i = 0; i < N; i++) { x += clock64() & 1; } t1 = rt(); if (lane == 0) { out[1] = t1 - t0; out[21] = x; } }
    __syncthreads();
    // 3. LDS pointer chase, wave 0 alone
    if (wave == 0) { t0 = rt(); int p = lane; for (int i = 0; i < N; i++) p = chase[p]; t1 = rt(); acc += p; if (lane == 0) out[2] = t1 - t0; }
    __syncthreads();
    // 4. LDS pointer chase, all 16 waves at once
    { t0 = rt(); int p = tid; for (int i = 0; i < N; i++) p = chase[p]; t1 = rt(); acc += p; if (tid == 0) out[3] = t1 - t0; }
    __syncthreads();
    // 5. LDS atomic with return, chain (wave 0, lane 0 only executes)
    if (tid == 0) { t0 = rt(); unsigned long long v = 5; for (int i = 0; i < N; i++) v = atomicMin(&slot, v + i) | 1; t1 = rt(); acc += (int)v; out[4] = t1 - t0; }
    __syncthreads();
    // 6. barrier, all waves
    { t0 = rt(); for (int i = 0; i < N; i++) __syncthreads(); t1 = rt(); if (tid == 0) out[5] = t1 - t0; }
    // 7. barrier + one LDS write/read hand-off per iteration (wave 0 writes, all read)
    { t0 = rt(); int v = 0; for (int i = 0; i < N; i++) { if (tid == 0) chase[0] = i + v; __syncthreads(); v += chase[0]; __syncthreads(); } t1 = rt(); acc += v; if (tid == 0) out[6] = t1 - t0; }
    // 8. global relaxed agent-scope load chain (wave 0)
    if (wave == 0) { t0 = rt(); int p = 0; for (int i = 0; i < N; i++) p = __hip_atomic_load(&gmem[p & 63], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); t1 = rt(); acc += p; if (lane == 0) out[7] = t1 - t0; }
    __syncthreads();
    // 9. wave64 dependent VALU chain (integer mad), wave 0 alone: cycles per dependent instruction
    if (wave == 0) { t0 = rt(); int x = lane; for (int i = 0; i < N; i++) { x = x * 3 + 1; x = x ^ (x >> 3); x = x * 5 + 7; x = x ^ (x >> 2); } t1 = rt(); acc += x; if (lane == 0) out[8] = t1 - t0; }
    __syncthreads();
    // 10. readlane / ballot / ff1 chain (the event loop's column search), wave 0
    if (wave == 0) { t0 = rt(); unsigned m = 0x80000000u >> (lane & 31); int c = 0; for (int i = 0; i < N; i++) { const unsigned long long b = __ballot((m >> (c & 7)) != 0); const int w = __ffsll((long long)b) - 1; const int word = __builtin_amdgcn_readlane((int)m, w); c += __ffs(word) + w; } t1 = rt(); acc += c; if (lane == 0) out[9] = t1 - t0; }
    __syncthreads();
    // 11. f64 add chain, all 16 waves (issue-bound: 4 waves per SIMD)
    { t0 = rt(); double x = tid; for (int i = 0; i < N; i++) { x = x + 1.5; x = x - 0.25; } t1 = rt(); acc += (int)x; if (tid == 0) out[10] = t1 - t0; }
    __syncthreads();
    // 12. f64 add chain, wave 0 alone
    if (wave == 0) { t0 = rt(); double x = tid; for (int i = 0; i < N; i++) { x = x + 1.5; x = x - 0.25; } t1 = rt(); acc += (int)x; if (lane == 0) out[11] = t1 - t0; }
    __syncthreads();
    // 13. LDS: 2 independent loads + dependent load (a typical event round trip pair), wave 0
    if (wave == 0) { t0 = rt(); int p = lane; for (int i = 0; i < N; i++) { const int a = chase[p], b = chase[(p + 64) & 1023]; p = chase[(a + b) & 1023]; } t1 = rt(); acc += p; if (lane == 0) out[12] = t1 - t0; }
    __syncthreads();
    // 14. LDS atomicOr without return + store + dependent load (in-order LDS queue), wave 0
    if (wave == 0) { unsigned* u = reinterpret_cast<unsigned*>(chase); t0 = rt(); int p = lane; for (int i = 0; i < N; i++) { atomicOr(&u[(p + 5) & 1023], 0u); u[512 + lane] = u[512 + lane]; p = chase[p]; } t1 = rt(); acc += p; if (lane == 0) out[13] = t1 - t0; }
    __syncthreads();
    if (acc == 0x7fffffff) out[30] = acc;
}

int main()
{
    long long* out; int* g;
    hipMalloc(&out, 64 * sizeof(long long)); hipMalloc(&g, 64 * sizeof(int)); hipMemset(g, 0, 64 * sizeof(int)); hipMemset(out, 0, 64 * sizeof(long long));
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_lat, dim3(1), dim3(1024), 0, 0, out, g, 16);
    hipDeviceSynchronize();
    long long h[64]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    const char* names[] = {"s_memrealtime (wall_clock64)", "s_memtime (clock64)", "LDS dependent read, 1 wave", "LDS dependent read, 16 waves", "LDS atomicMin u64 with return, 1 lane",
        "__syncthreads, 16 waves", "2 x __syncthreads + LDS hand-off", "global relaxed agent load, dependent", "4 dependent int VALU ops, 1 wave", "ballot+ffs+readlane+ffs chain",
        "2 dependent f64 adds, 16 waves", "2 dependent f64 adds, 1 wave", "LDS 2 loads -> dependent load", "LDS atomicOr + store + dependent load"};
    for (int i = 0; i < 14; i++) printf("%-45s %8.1f ns per iteration\n", names[i], (double)h[i] * 10.0 / N);
    return 0;
}
