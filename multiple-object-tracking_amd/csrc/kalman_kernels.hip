// kalman_kernels.hip -- 6-state / 4-measurement float64 Kalman tracker,
// one 64-lane wavefront per track (trackers/kalman.cpp:29-128,
// include/sigpack/kalman/kalman.h:207-237).
//
// Lane (i + 6*j) owns element (i,j) of every 6x6 product; operands are staged
// in LDS (one 3 KB slab per wavefront).  Products accumulate k = 0..5 in
// order with separate multiply/add roundings, matching the oracle.
#include "mot_dev.h"
#include "mot_env.h"

namespace {

#define KIX(r, c, nr) ((c) * (nr) + (r))

struct KalmanConst { double A[36], H[24], Q[36], R[16]; };

__device__ __forceinline__ void kalman_consts(double* A, double* H, double* Q, double* R, int lane)
{
    // kalman.cpp:55-90 (column-major)
    if (lane < 36) {
        const int r = lane % 6, c = lane / 6;
        double a = (r == c) ? 1.0 : 0.0;
        if ((r == 0 || r == 2) && c == 4) a = 1.0;
        if ((r == 1 || r == 3) && c == 5) a = 1.0;
        A[lane] = a;
        double q = 0.0;
        if (r == c) q = (r < 4) ? 0.25 : 1.0;
        else {
            const int lo = r < c ? r : c, hi = r < c ? c : r;
            if (hi == 4 && (lo == 0 || lo == 2)) q = 0.5;
            if (hi == 5 && (lo == 1 || lo == 3)) q = 0.5;
        }
        Q[lane] = 1e-2 * q;
    }
    if (lane < 24) { const int r = lane % 4, c = lane / 4; H[lane] = (r == c) ? 1.0 : 0.0; }
    if (lane < 16) { const int r = lane % 4, c = lane / 4; R[lane] = (r == c) ? 512.0 : 0.0; }
}

// C[m x n] = A[m x k] * (tB ? B^T : B), executed by lanes < m*n
__device__ __forceinline__ void wmatmul(double* C, const double* A, const double* B, int m, int k, int n, bool tB, int lane)
{
    if (lane < m * n) {
        const int i = lane % m, j = lane / m;
        double acc = 0.0;
        for (int p = 0; p < k; p++) acc += A[KIX(i, p, m)] * (tB ? B[KIX(j, p, n)] : B[KIX(p, j, k)]);
        C[lane] = acc;
    }
}

__device__ __forceinline__ double det3(const double* m, int r0, int r1, int r2, int c0, int c1, int c2)
{
#define M4(r, c) m[KIX(r, c, 4)]
    return M4(r0, c0) * (M4(r1, c1) * M4(r2, c2) - M4(r1, c2) * M4(r2, c1))
         - M4(r0, c1) * (M4(r1, c0) * M4(r2, c2) - M4(r1, c2) * M4(r2, c0))
         + M4(r0, c2) * (M4(r1, c0) * M4(r2, c1) - M4(r1, c1) * M4(r2, c0));
#undef M4
}

constexpr int KWS = 400; // doubles of LDS per wavefront

__global__ void __launch_bounds__(256) kalman_predict_kernel(KalmanPool p, const int* slots, const int* count, int n, bbox_t* boxes_out, int clamp)
{
    __shared__ double ws_all[4 * KWS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + wave;
    double* ws = ws_all + wave * KWS;
    double *A = ws, *H = A + 36, *Q = H + 24, *R = Q + 36, *X = R + 16, *P = X + 8, *T = P + 36, *Xn = T + 36;
    const bool active = item < n && (!count || item < *count);
    kalman_consts(A, H, Q, R, lane);
    int slot = 0;
    if (active) {
        slot = slots[item];
        if (lane < 6) X[lane] = p.x[(size_t)slot * 6 + lane];
        if (lane < 36) P[lane] = p.P[(size_t)slot * 36 + lane];
    }
    __syncthreads();
    if (active) wmatmul(Xn, A, X, 6, 6, 1, false, lane);             // x = A*x        (kalman.h:209)
    if (active) wmatmul(T, A, P, 6, 6, 6, false, lane);              // (A*P)*A' + Q   (kalman.h:210)
    __syncthreads();
    double pn = 0.0;
    if (active && lane < 36) {
        const int i = lane % 6, j = lane / 6;
        double acc = 0.0;
        for (int q = 0; q < 6; q++) acc += T[KIX(i, q, 6)] * A[KIX(j, q, 6)];
        pn = acc + Q[lane];
        p.P[(size_t)slot * 36 + lane] = pn;
    }
    if (active && lane < 6) p.x[(size_t)slot * 6 + lane] = Xn[lane];
    if (active && lane == 0 && boxes_out) {
        bbox_t o = boxes_out[item];                                   // predict writes only l,t,r,b (kalman.cpp:112-115)
        o.l = (int)Xn[0]; o.t = (int)Xn[1]; o.r = (int)Xn[2]; o.b = (int)Xn[3];
        if (clamp) {
            o.l = min(max(o.l, 0), MOT_FRAME_W - 1); o.r = min(max(o.r, 0), MOT_FRAME_W - 1);
            o.t = min(max(o.t, 0), MOT_FRAME_H - 1); o.b = min(max(o.b, 0), MOT_FRAME_H - 1);
        }
        boxes_out[item] = o;
    }
}

__global__ void __launch_bounds__(256) kalman_update_kernel(KalmanPool p, const int* slots, const int* count, int n, const bbox_t* boxes)
{
    __shared__ double ws_all[4 * KWS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + wave;
    double* ws = ws_all + wave * KWS;
    double *A = ws, *H = A + 36, *Q = H + 24, *R = Q + 36, *X = R + 16, *P = X + 8, *T = P + 36, *U = T + 36,
           *S = U + 36, *Si = S + 16, *K = Si + 16, *Jf = K + 24, *E = Jf + 36, *KR = E + 8, *V = KR + 24;
    const bool active = item < n && (!count || item < *count);
    kalman_consts(A, H, Q, R, lane);
    int slot = 0;
    if (active) {
        slot = slots[item];
        if (lane < 6) X[lane] = p.x[(size_t)slot * 6 + lane];
        if (lane < 36) P[lane] = p.P[(size_t)slot * 36 + lane];
    }
    __syncthreads();
    if (active) wmatmul(T, H, P, 4, 6, 6, false, lane);              // H*P
    __syncthreads();
    if (active && lane < 16) {                                        // S = H*P*H' + R
        const int i = lane % 4, j = lane / 4; double acc = 0.0;
        for (int q = 0; q < 6; q++) acc += T[KIX(i, q, 4)] * H[KIX(j, q, 4)];
        S[lane] = acc + R[lane];
    }
    if (active) wmatmul(U, P, H, 6, 6, 4, true, lane);               // P*H'
    __syncthreads();
    if (active && lane < 16) {                                        // inv(S): adjugate / det (auxlib::inv_tiny)
        const int i = lane % 4, j = lane / 4;                         // element (i,j) of the inverse = cofactor(j,i)/det
        int rr[3], cc[3]; int a = 0, b = 0;
        for (int q = 0; q < 4; q++) { if (q != j) rr[a++] = q; if (q != i) cc[b++] = q; }
        const double cof = det3(S, rr[0], rr[1], rr[2], cc[0], cc[1], cc[2]) * (((i + j) & 1) ? -1.0 : 1.0);
        const double det = S[KIX(0, 0, 4)] * det3(S, 1, 2, 3, 1, 2, 3) - S[KIX(0, 1, 4)] * det3(S, 1, 2, 3, 0, 2, 3)
                         + S[KIX(0, 2, 4)] * det3(S, 1, 2, 3, 0, 1, 3) - S[KIX(0, 3, 4)] * det3(S, 1, 2, 3, 0, 1, 2);
        Si[lane] = cof / det;
    }
    if (active && lane < 4) {                                         // z - H*x
        const bbox_t z = boxes[item];
        const double zz[4] = { (double)z.l, (double)z.t, (double)z.r, (double)z.b };
        double acc = 0.0;
        for (int q = 0; q < 6; q++) acc += H[KIX(lane, q, 4)] * X[q];
        E[lane] = zz[lane] - acc;
    }
    __syncthreads();
    if (active) wmatmul(K, U, Si, 6, 4, 4, false, lane);             // K = (P*H')*inv(S)
    __syncthreads();
    if (active && lane < 6) {                                         // x += K*err
        double acc = 0.0;
        for (int q = 0; q < 4; q++) acc += K[KIX(lane, q, 6)] * E[q];
        p.x[(size_t)slot * 6 + lane] = X[lane] + acc;
    }
    if (active && lane < 36) {                                        // Jf = I - K*H
        const int i = lane % 6, j = lane / 6; double acc = 0.0;
        for (int q = 0; q < 4; q++) acc += K[KIX(i, q, 6)] * H[KIX(q, j, 4)];
        Jf[lane] = ((i == j) ? 1.0 : 0.0) - acc;
    }
    if (active) wmatmul(KR, K, R, 6, 4, 4, false, lane);             // K*R
    __syncthreads();
    if (active) wmatmul(T, Jf, P, 6, 6, 6, false, lane);             // Jf*P
    if (active) wmatmul(V, KR, K, 6, 4, 6, true, lane);              // K*R*K'
    __syncthreads();
    if (active && lane < 36) {                                        // P = Jf*P*Jf' + K*R*K'
        const int i = lane % 6, j = lane / 6; double acc = 0.0;
        for (int q = 0; q < 6; q++) acc += T[KIX(i, q, 6)] * Jf[KIX(j, q, 6)];
        p.P[(size_t)slot * 36 + lane] = acc + V[lane];
    }
}

__global__ void kalman_init_kernel(KalmanPool p, const int* slots, int n, const bbox_t* boxes)
{
    const int item = blockIdx.x, lane = threadIdx.x;
    if (item >= n) return;
    const int slot = slots[item];
    const bbox_t b = boxes[item];
    if (lane < 6) {                                                   // kalman.cpp:152-157
        const double v[6] = { (double)b.l, (double)b.t, (double)b.r, (double)b.b, 0.0, 0.0 };
        p.x[(size_t)slot * 6 + lane] = v[lane];
    }
    if (lane < 36) p.P[(size_t)slot * 36 + lane] = (lane % 6 == lane / 6) ? 1e+4 : 0.0; // kalman.cpp:90
}

} // namespace

hipError_t launch_kalman_predict(const KalmanPool& p, const int* slots, const int* count, int n, bbox_t* boxes_out, int clamp, hipStream_t s)
{
    mot_impl::lds_poison(s);                                           // (debug) MOT_LDS_POISON
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(kalman_predict_kernel, dim3((n + 3) / 4), dim3(256), 0, s, p, slots, count, n, boxes_out, clamp);
    return hipGetLastError();
}
hipError_t launch_kalman_update(const KalmanPool& p, const int* slots, const int* count, int n, const bbox_t* boxes, hipStream_t s)
{
    mot_impl::lds_poison(s);                                           // (debug) MOT_LDS_POISON
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(kalman_update_kernel, dim3((n + 3) / 4), dim3(256), 0, s, p, slots, count, n, boxes);
    return hipGetLastError();
}
hipError_t launch_kalman_init(const KalmanPool& p, const int* slots, int n, const bbox_t* boxes, hipStream_t s)
{
    mot_impl::lds_poison(s);                                           // (debug) MOT_LDS_POISON
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(kalman_init_kernel, dim3(n), dim3(64), 0, s, p, slots, n, boxes);
    return hipGetLastError();
}
