// assoc_common.h -- shared device helpers of the association kernels (assoc_kernels.hip)
#pragma once
#include "mot_dev.h"
#include <float.h>

namespace assoc {

#define MK_MAXN 1024
#define MK_MAXW 16                  // 64-bit words per bitmap line
#define MK_THREADS 1024

typedef unsigned long long u64;

// ---- step-5 helper protocol (munkres_kernel<true> launched with 1 + MK_HELPERS workgroups) --------------------
// Control block of u64 words; every access is a relaxed agent-scope atomic (sc1 load / store) and flags follow G16 of
// the HIP guide (payload stores drained with s_waitcnt vmcnt(0) before the flag store of the same wave).  Words that
// are polled sit on 128-byte lines of their own.
//   COV       64 granules {32 bits, 32-bit tag}: the 16 row and 16 column cover-mask words of the current step 5,
//             two granules per word; written by ONE store instruction of the controller's wave 0, polled by the 64
//             lanes of each helper's wave 0.  Tag MK_TAG_EXIT: the controller is done.
//   XCCTAB    every helper reports the XCD it runs on; MODE: the controller's verdict "all 17 workgroups share one XCD"
//             (then the hand-off stores are plain and stay in that XCD's L2, ctl_stx)
//   EPOCH     tag base: tags are EPOCH + step, unique across launches (the controller advances it when it is done)
//   PARTIAL   [step & 1][g]: helper g's minimum key over (uncovered rows) x (its uncovered columns); MK_HSENT = not
//             yet, MK_KEY_NONE = no such element.  Every helper with work in the update phase reads all 16 and takes
//             h = their minimum; the controller re-arms them when the step's zero bits are in
//   BMOUT     [column][row word][2]: new zero bits of an uncovered column as two granules; a column's 32 granules are
//             staged in LDS and written by one store instruction (256 contiguous bytes), not as 32 scalar writes
//   COVBITS   [column word][i-th covered row][2]: zero bits of (covered row) x (64 covered columns), two granules
//   COVSUM    [g]: how many of helper g's COVBITS rows hold a zero at all (normally none: such an entry just grew by h > 0); the
//             controller reads the COVBITS granules only if some count is not 0
// A granule is an aligned 8-byte word written by ONE store and carrying its own tag, so none of these hand-offs needs
// a separate flag, an arrival counter or a store drain (guide: R2 granule, "needs no ordering at all"); the re-armed
// words are drained by the controller one step before they are used again.
#define MK_HELPERS 16
#define CTL_EPOCH 32
#define CTL_MODE 33                    /* {same-XCD flag, tag = EPOCH + 1}: written by the controller before its first publish */
#define CTL_XCCTAB 48                  /* 16 granules {XCC id of helper g, tag = EPOCH + 1} */
#define CTL_COV 64                     /* 64 granules */
#define CTL_PARTIAL 128                /* [2][16] words, each on a 128-byte line of its own (16 writers) */
#define MK_PARTIAL_STRIDE 16
#define CTL_COVSUM 640                 /* [16] granules {zeros among (covered rows) x (helper g's covered columns) after the step, tag}, one 128-byte line each */
#define CTL_COVBITS 1024               /* [16 column words][1024 rows][2]: a helper's granules are contiguous */
#define CTL_BMOUT (1024 + 2 * 16384)   /* [1024 cols][16 row words][2] */
#define MK_CTL_WORDS (1024 + 4 * 16384)
#define MK_TAG_EXIT 0xFFFFFFFFu
#define MK_HSENT 0xFFFFFFFFFFFFFFFFull /* a NaN pattern: h is always finite; as a key: above every cost's key */
#define MK_KEY_NONE 0xFFFFFFFFFFFFFFFEull
static_assert(MK_CTL_WORDS == MOT_ASSOC_CTL_WORDS, "control block size");
__device__ __forceinline__ u64 ctl_ld(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ctl_st(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Hand-off store.  `same_xcd`: producer and every consumer were VERIFIED (HW_REG_XCC_ID) to sit on one XCD, i.e. behind one
// L2: a plain 8-byte store keeps the line in that L2 and the consumers' sc1 loads (L1 bypass) are served from it, instead of
// the write-through + memory-side read of the agent-scope form.
__device__ __forceinline__ void ctl_stx(u64* p, u64 v, bool same_xcd)
{
    if (same_xcd) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int xcc_id() { int x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xF; }

__device__ __forceinline__ u64 dkey(double v) { u64 b = (u64)__double_as_longlong(v); return (b >> 63) ? ~b : (b | 0x8000000000000000ull); }
__device__ __forceinline__ double dunkey(u64 k) { u64 b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k; return __longlong_as_double((long long)b); }

// td.cpp:407-419
__device__ __forceinline__ double pair_cost(const bbox_t a, const bbox_t d)
{
    const int cxi = (a.l + a.r) >> 1, cyi = (a.t + a.b) >> 1;
    const int cxj = (d.l + d.r) >> 1, cyj = (d.t + d.b) >> 1;
    double dist = 0.0;
    dist += sqrt((double)((cxi - cxj) * (cxi - cxj) + (cyi - cyj) * (cyi - cyj))) * (1.0 / ((double)MOT_FRAME_W));
    if (a.type != d.type) dist += 1.0;
    return dist;
}

// The same cost from the integer squared centroid distance d2 and the class flag (what pair_cost evaluates: 0.0 + sqrt(d2) * (1/1280),
// then + 1.0 on a class mismatch), and a float estimate of it whose absolute error stays below PAIR_COST_F32_ERR (d2 < 2^24 is
// exact in float, sqrtf and the product round to 1 ulp each, values < 2.2): dense passes use the estimate to discard the entries
// that are far from any threshold and evaluate the float64 form only for the few that are close.
#define PAIR_COST_F32_ERR 1e-6
__device__ __forceinline__ void pair_d2(const bbox_t a, const bbox_t d, int& d2, bool& pen)
{
    const int cxi = (a.l + a.r) >> 1, cyi = (a.t + a.b) >> 1;
    const int cxj = (d.l + d.r) >> 1, cyj = (d.t + d.b) >> 1;
    d2 = (cxi - cxj) * (cxi - cxj) + (cyi - cyj) * (cyi - cyj);
    pen = a.type != d.type;
}
__device__ __forceinline__ double cost_of_d2(int d2, bool pen)
{
    double dist = 0.0;
    dist += sqrt((double)d2) * (1.0 / ((double)MOT_FRAME_W));
    if (pen) dist += 1.0;
    return dist;
}
__device__ __forceinline__ float cost_of_d2_f32(int d2, bool pen) { return sqrtf((float)d2) * (1.0f / (float)MOT_FRAME_W) + (pen ? 1.0f : 0.0f); }
// the estimate needs d2 < 2^24 (exact in float): centroids within +-1400 of the origin give |dx|, |dy| <= 2800.  Tracker boxes are
// clamped to the 1280 x 720 frame (td.cpp:378-381) and detections lie in it; anything else takes the float64 form throughout.
__device__ __forceinline__ bool box_small(const bbox_t b) { return (unsigned)(b.l + 1400) <= 2800u && (unsigned)(b.r + 1400) <= 2800u && (unsigned)(b.t + 1400) <= 2800u && (unsigned)(b.b + 1400) <= 2800u; }

struct AssocArgs {
    const bbox_t* trk; const bbox_t* det; const int* nT_dev; int nT; int nD;
    const double* user; int userR, userC;
    AssocWs ws;
    u64* linemin;        // [MK_MAXN] keys
    int* dims;           // [4]: nR, nC, rowsAreTrackers, minIsPerRow
    double* cost_only;   // assoc_cost_kernel output
    // round 6: every launch chain has a sequence number (host counter, >= 1); the protocol words between the solver's workgroup, the sparse
    // emulation and the final kernel carry it in their upper bits, so a word left by another frame is never mistaken for this frame's
    unsigned seq;
    int stream_emu;      // the sparse emulation runs as a kernel of its own on the emulation stream (device loop, provisional commits possible)
    bbox_t* det_copy;    // stream_emu: the row scan copies the detection list here; the emulation's kernel reads this copy (the caller may overwrite its list behind the call)
};

// tagged protocol word: (seq << 2) | value.  Value of `word` for chain `seq`: its low bits if the tag is this chain's, `newer` if a LATER chain
// has written the word since (this chain's reader is obsolete), 0 (nothing yet) if the word is older
__device__ __forceinline__ int tagged_value(int word, unsigned seq, int newer)
{
    const unsigned t = (unsigned)word >> 2, s = seq & 0x3FFFFFFFu;
    if (t == s) return word & 3;
    return ((t - s) & 0x3FFFFFFFu) < 0x20000000u ? newer : 0;
}
__device__ __forceinline__ int tagged_word(unsigned seq, int v) { return (int)(((seq & 0x3FFFFFFFu) << 2) | (unsigned)v); }

__device__ __forceinline__ void resolve_dims(const AssocArgs& a, int& nR, int& nC, bool& rowsTrk)
{
    if (a.user) { nR = a.userR; nC = a.userC; rowsTrk = true; return; }
    const int nT = a.nT_dev ? *a.nT_dev : a.nT;
    if (nT < a.nD) { nR = nT; nC = a.nD; rowsTrk = true; }            // td.cpp:388,462-465
    else { nR = a.nD; nC = nT; rowsTrk = false; }
}

__device__ __forceinline__ double elem_cost(const AssocArgs& a, int r, int c, int nR, bool rowsTrk)
{
    if (a.user) return a.user[(size_t)r + (size_t)nR * c];
    return rowsTrk ? pair_cost(a.trk[r], a.det[c]) : pair_cost(a.trk[c], a.det[r]);
}

__device__ __forceinline__ u64 readlane64(u64 v, int src)          // src must be wave-uniform
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
    return ((u64)hi << 32) | lo;
}

// ---- wave reductions on the DPP path (hipcc lowers __shfl_xor to ds_bpermute: an LDS round trip per step and half) ----
// four v_mov_dpp steps reduce inside every row of 16 lanes (all 16 lanes end up with the row's result), the four row results are
// combined through v_readlane: the result is wave-uniform
template <int CTRL> __device__ __forceinline__ unsigned dpp_mov32(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false); }
template <int CTRL> __device__ __forceinline__ u64 dpp_mov64(u64 v) { return ((u64)dpp_mov32<CTRL>((unsigned)(v >> 32)) << 32) | dpp_mov32<CTRL>((unsigned)v); }
#define DPP_QUAD_XOR1 0xB1      /* quad_perm [1,0,3,2] */
#define DPP_QUAD_XOR2 0x4E      /* quad_perm [2,3,0,1] */
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140
__device__ __forceinline__ u64 row16_min_u64(u64 v)
{
    u64 o;
    o = dpp_mov64<DPP_QUAD_XOR1>(v); v = o < v ? o : v;
    o = dpp_mov64<DPP_QUAD_XOR2>(v); v = o < v ? o : v;
    o = dpp_mov64<DPP_ROW_HALF_MIRROR>(v); v = o < v ? o : v;
    o = dpp_mov64<DPP_ROW_MIRROR>(v); v = o < v ? o : v;
    return v;
}
__device__ __forceinline__ u64 wave_min_u64_dpp(u64 v)
{
    v = row16_min_u64(v);
    const u64 a = readlane64(v, 0), b = readlane64(v, 16), c = readlane64(v, 32), d = readlane64(v, 48);
    const u64 ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
__device__ __forceinline__ u64 wave_max_u64_dpp(u64 v) { return ~wave_min_u64_dpp(~v); }
__device__ __forceinline__ unsigned wave_min_u32_dpp(unsigned v)
{
    unsigned o;
    o = dpp_mov32<DPP_QUAD_XOR1>(v); v = o < v ? o : v;
    o = dpp_mov32<DPP_QUAD_XOR2>(v); v = o < v ? o : v;
    o = dpp_mov32<DPP_ROW_HALF_MIRROR>(v); v = o < v ? o : v;
    o = dpp_mov32<DPP_ROW_MIRROR>(v); v = o < v ? o : v;
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16),
                   c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    const unsigned ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
// minimum of doubles (any sign, no NaNs): through the order-preserving key
__device__ __forceinline__ double wave_min_f64_dpp(double x) { return dunkey(wave_min_u64_dpp(dkey(x))); }
__device__ __forceinline__ double row16_min_f64(double x) { return dunkey(row16_min_u64(dkey(x))); }
// v_min_f64 as one instruction (the C++ '<' select costs a compare and two moves, fmin() adds canonicalising moves): NaN-free callers only
__device__ __forceinline__ double vmin_f64(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int CTRL> __device__ __forceinline__ double dpp_mov64(double x)
{
    const u64 u = (u64)__double_as_longlong(x);
    const unsigned lo = dpp_mov32<CTRL>((unsigned)u), hi = dpp_mov32<CTRL>((unsigned)(u >> 32));
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}
// wave-wide minimum of NaN-free doubles with v_min_f64 at every stage (three instructions per stage instead of eight through the key)
__device__ __forceinline__ double wave_min_f64_pos(double x)
{
    x = vmin_f64(dpp_mov64<DPP_QUAD_XOR1>(x), x);
    x = vmin_f64(dpp_mov64<DPP_QUAD_XOR2>(x), x);
    x = vmin_f64(dpp_mov64<DPP_ROW_HALF_MIRROR>(x), x);
    x = vmin_f64(dpp_mov64<DPP_ROW_MIRROR>(x), x);
    const u64 u = (u64)__double_as_longlong(x);
    const double a = __longlong_as_double((long long)readlane64(u, 0)), b = __longlong_as_double((long long)readlane64(u, 16)),
                 c = __longlong_as_double((long long)readlane64(u, 32)), d = __longlong_as_double((long long)readlane64(u, 48));
    const double ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

__device__ __forceinline__ int wave_first_bit(u64 m, int lane, int nwords)
{   // lanes < nwords hold words of a line; returns index of the first set bit, or -1 (wave-uniform)
    const u64 bal = __ballot(lane < nwords && m != 0);
    if (!bal) return -1;
    const int w = __ffsll((long long)bal) - 1;
    const u64 word = readlane64(m, w);
    return w * 64 + (__ffsll((long long)word) - 1);
}

// ascending list of the set bits of a <=1024-bit mask held one word per lane, restricted to bits >= from;
// wave 0 only, returns the count (wave-uniform)
__device__ __forceinline__ int wave_list_bits(u64 w, int from, unsigned short* out, int lane)
{
    const int fw = from >> 6;
    if (lane < fw) w = 0;
    else if (lane == fw) w &= ~0ull << (from & 63);
    const int cnt = __popcll(w);
    int pre = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(pre, off); if (lane >= off) pre += t; }
    const int total = __builtin_amdgcn_readlane(pre, 63);
    int pos = pre - cnt;
    while (w) { const int b = __ffsll((long long)w) - 1; out[pos++] = (unsigned short)(lane * 64 + b); w &= w - 1; }
    return total;
}


} // namespace assoc

