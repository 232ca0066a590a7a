// rd_kernel.hip -- batched mode-decision RD evaluation on gfx950: one wave64 workgroup per transform block runs
//   residual -> forward 2-D transform -> SATD -> quantize/dequantize (+eob) -> coefficient-domain distortion
//   -> inverse 2-D transform + reconstruction -> pixel-domain SSE
// i.e. one iteration of the reference's tx_type_search body (Source/Lib/Codec/product_coding_loop.c:4764-4934).
//
// Mapping: the W x H block lives in LDS as int32; a lane owns one column (then one row) of the separable
// transform and keeps the whole 1-D vector in registers -- the butterfly networks are fully unrolled at compile
// time from their structure (see dct_* below), so every register index and every cosine is a constant.  All
// arithmetic is the reference's integer arithmetic (32-bit wrapping products, 64-bit rounding: half_btf,
// Codec/inv_transforms.h:264-285), hence bit-exact.  Elementwise phases (quantizer, distortions) stride the block
// across the 64 lanes and reduce with shuffles.  Memory traffic per block is the algorithmic minimum: source and
// prediction read once, outputs written once (HBM-bound by design).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>
#include "svt_hip_internal.h"
#include "../../include/svt_hip_dsp.h"

namespace {

typedef unsigned long long u64;
typedef long long          i64;

__constant__ int32_t c_cospi[4][64]; // cospi_arr(bit), bit 10..13 (round(cos(j*pi/128) * 2^bit)), filled at init
// svt_aom_eb_av1_sinpi_arr_data rows for cos_bit 10..13 (Codec/inv_transforms.c:3228-3234)
__constant__ int32_t c_sinpi[4][5] = {{0, 330, 621, 836, 951}, {0, 660, 1241, 1672, 1901}, {0, 1321, 2482, 3344, 3803}, {0, 2642, 4964, 6689, 7606}};

struct TxGeom { uint8_t w, h; };
__host__ __device__ constexpr int tx_wide(int s) { constexpr uint8_t t[19] = {4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64}; return t[s]; }
__host__ __device__ constexpr int tx_high(int s) { constexpr uint8_t t[19] = {4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16}; return t[s]; }
__host__ __device__ constexpr int ilog2c(int n) { return n <= 1 ? 0 : 1 + ilog2c(n >> 1); }
__host__ __device__ constexpr int brevc(int v, int bits) { int r = 0; for (int i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i); return r; }

// fwd_txfm_shift_ls (Codec/transforms.h:27-45), fwd_cos_bit_col/row (:47-50), inv shifts (Codec/inv_transforms.c:17-35),
// av1_get_tx_scale_tab (Codec/full_loop.h:53)
__device__ const int8_t  c_fwd_shift[19][3] = {{2, 0, 0},  {2, -1, 0}, {2, -2, 0}, {2, -4, 0}, {0, -2, -2}, {2, -1, 0}, {2, -1, 0}, {2, -2, 0}, {2, -2, 0}, {2, -4, 0},
                                               {2, -4, 0}, {0, -2, -2}, {2, -4, -2}, {2, -1, 0}, {2, -1, 0}, {2, -2, 0}, {2, -2, 0}, {0, -2, 0}, {2, -4, 0}};
__device__ const int8_t  c_fwd_cos_col[5][5] = {{13, 13, 13, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 13, 12, 13}, {0, 13, 13, 12, 13}, {0, 0, 13, 12, 13}};
__device__ const int8_t  c_fwd_cos_row[5][5] = {{13, 13, 12, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 12, 13, 12}, {0, 12, 13, 12, 11}, {0, 0, 12, 11, 10}};
__device__ const int8_t  c_inv_shift0[19]    = {0, -1, -2, -2, -2, 0, 0, -1, -1, -1, -1, -1, -1, -1, -1, -2, -2, -2, -2};
__device__ const uint8_t c_log_scale[19]     = {0, 0, 0, 1, 2, 0, 0, 0, 0, 1, 1, 2, 2, 0, 0, 0, 0, 1, 1};
// 1-D kernel of the column (vertical) / row (horizontal) pass per TxType: 0 DCT, 1 ADST, 2 FLIPADST, 3 identity (vtx_tab/htx_tab)
__device__ const uint8_t c_vtx[16] = {0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3};
__device__ const uint8_t c_htx[16] = {0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2};

// ((int64)a * b) >> sh (0 < sh < 32) for operands that fit 24 signed bits, low 32 bits of the result: three full-rate instructions
// (v_mul_i32_i24, v_mul_hi_i32_i24, v_alignbit_b32) instead of the quarter-rate 32 x 32 -> 64 multiply
__device__ __forceinline__ int32_t mul24_shr(int32_t a, int32_t b, int sh) {
    int32_t hi;
    asm("v_mul_hi_i32_i24 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    return (int32_t)__builtin_amdgcn_alignbit((uint32_t)hi, (uint32_t)__mul24(a, b), (uint32_t)sh);
}
__device__ __forceinline__ int32_t rshift64(i64 v, int bit) { return (int32_t)((v + ((i64)1 << (bit - 1))) >> bit); }
// half_btf of the reference (Codec/transforms.h / inv_transforms.h): two 32-bit wrapping products, summed and rounded in
// 64 bits.  MUL == 1 (inverse transforms): both operands of every product fit 24 signed bits -- cos weights < 2^13, data
// < 2^19 behind the reference's own stage clamps (clamp_value to bd + 8 / 16..18 bits) -- so the full-rate v_mul_i32_i24
// returns the same low 32 bits as the quarter-rate v_mul_lo_u32.  Forward transforms keep the 32-bit multiply: their
// data range depends on the caller's samples.
// MUL == 2: the caller has bounded the data so that |a| + |b| < 2^18 (weights <= 2^13): neither product wraps and their sum with the rounding
// term stays below 2^31, so the whole butterfly is three full-rate 32-bit instructions with the reference's exact result.
template <int MUL> __device__ __forceinline__ int32_t hbtf(int32_t w0, int32_t a, int32_t w1, int32_t b, int bit) {
    if constexpr (MUL == 2) return (__mul24(w1, b) + (__mul24(w0, a) + (1 << (bit - 1)))) >> bit; // two v_mad_i32_i24 + a shift
    i64 s;
    if constexpr (MUL == 1) s = (i64)__mul24(w0, a) + (i64)__mul24(w1, b);
    else s = (i64)(int32_t)((uint32_t)w0 * (uint32_t)a) + (i64)(int32_t)((uint32_t)w1 * (uint32_t)b);
    return (int32_t)((s + ((i64)1 << (bit - 1))) >> bit);
}
// clamp_value of a SUM or DIFFERENCE inside an inverse pass.  The reference widens to 64 bits before it clamps; here the operands are at most
// 20-bit numbers -- the pass inputs are clamped to bd + 8 / 16 bits on entry, every butterfly output is clamped again, and a half_btf output is
// below sqrt(2) times its inputs -- so the 32-bit sum is the same number and the clamp is one v_med3_i32.
__device__ __forceinline__ int32_t clamp32(int32_t v, int bit) {
    const int32_t hi = (1 << (bit - 1)) - 1, lo = -(1 << (bit - 1));
    return v < lo ? lo : (v > hi ? hi : v);
}
__device__ __forceinline__ int32_t clampv(i64 v, int bit) {
    const i64 hi = ((i64)1 << (bit - 1)) - 1, lo = -((i64)1 << (bit - 1));
    return (int32_t)(v < lo ? lo : (v > hi ? hi : v));
}
__device__ __forceinline__ int32_t clampv(int32_t v, int bit) { return clamp32(v, bit); } // a 32-bit value against a range of at most 32 bits: the same number
__device__ __forceinline__ int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }

// ---------------------------------------------------------------------------------------------------------
// DCT flow graph, by structure.  x[M..2M) is the odd part of a 2M-point DCT:
//   rotation stage k (k = 1..log2(M)-1): lanes in the middle half of every group of 2t (t = M >> k) are rotated
//     with their mirror image (3M-1-p); angle (1 + 4*brev(group)) * (64 >> k); symmetric 2x2 blocks, so the same
//     stage serves both directions;
//   butterfly stage k: groups of t, even groups sum-first, odd groups difference-first;
//   final stage: lane M+i with 2M-1-i, angle 64 - (2*brev(i)+1)*(32/M).
// CLAMP < 0: forward transform (no clamps); otherwise the reference's clamp_value(stage_range) of the inverse.
// ---------------------------------------------------------------------------------------------------------
template <int M, int K, int MUL> __device__ __forceinline__ void odd_rot(int32_t *x, const int32_t *c, int bit) {
    constexpr int t = M >> K;
    if constexpr (K == 1) {
#pragma unroll
        for (int j = 0; j < M / 4; j++) {
            const int p = M + M / 4 + j, m = 3 * M - 1 - p;
            const int32_t a = x[p], b = x[m];
            x[p] = hbtf<MUL>(-c[32], a, c[32], b, bit);
            x[m] = hbtf<MUL>(c[32], b, c[32], a, bit);
        }
    } else {
#pragma unroll
        for (int g = 0; g < (1 << (K - 2)); g++) {
            const int A = (1 + 4 * brevc(g, K - 2)) * (64 >> K), B = 64 - A, base = M + g * 2 * t;
#pragma unroll
            for (int j = 0; j < t / 2; j++) {
                const int p = base + t / 2 + j, m = 3 * M - 1 - p;
                const int32_t a = x[p], b = x[m];
                x[p] = hbtf<MUL>(-c[A], a, c[B], b, bit);
                x[m] = hbtf<MUL>(c[A], b, c[B], a, bit);
            }
#pragma unroll
            for (int j = 0; j < t / 2; j++) {
                const int p = base + t + j, m = 3 * M - 1 - p;
                const int32_t a = x[p], b = x[m];
                x[p] = hbtf<MUL>(-c[B], a, -c[A], b, bit);
                x[m] = hbtf<MUL>(c[B], b, -c[A], a, bit);
            }
        }
    }
}
template <int M, int K, int CLAMP> __device__ __forceinline__ void odd_bfly(int32_t *x) {
    constexpr int t = M >> K;
#pragma unroll
    for (int g = 0; g < M / t; g++) {
#pragma unroll
        for (int j = 0; j < t / 2; j++) {
            const int i0 = M + g * t + j, i1 = M + g * t + t - 1 - j;
            const int32_t lo = x[i0], hi = x[i1];
            int32_t s, d;
            if constexpr (CLAMP < 0) { s = wadd(lo, hi); d = (g & 1) ? wsub(hi, lo) : wsub(lo, hi); }
            else { s = clamp32(lo + hi, CLAMP); d = clamp32((g & 1) ? hi - lo : lo - hi, CLAMP); }
            x[i0] = (g & 1) ? d : s;
            x[i1] = (g & 1) ? s : d;
        }
    }
}
template <int M, bool INV, int MUL = INV ? 1 : 0> __device__ __forceinline__ void odd_final(int32_t *x, const int32_t *c, int bit) {
#pragma unroll
    for (int i = 0; i < M / 2; i++) {
        const int A = 64 - (2 * brevc(i, ilog2c(M)) + 1) * (32 / M), B = 64 - A, p = M + i, m = 2 * M - 1 - i;
        const int32_t a = x[p], b = x[m];
        if constexpr (!INV) { x[p] = hbtf<MUL>(c[A], a, c[B], b, bit); x[m] = hbtf<MUL>(c[A], b, -c[B], a, bit); }
        else { x[p] = hbtf<MUL>(c[A], a, -c[B], b, bit); x[m] = hbtf<MUL>(c[B], a, c[A], b, bit); }
    }
}
template <int M, int K, int FM> struct OddFwd {
    static __device__ __forceinline__ void run(int32_t *x, const int32_t *c, int bit) {
        if constexpr (K < ilog2c(M)) {
            odd_rot<M, K, FM>(x, c, bit);
            odd_bfly<M, K, -1>(x);
            OddFwd<M, K + 1, FM>::run(x, c, bit);
        }
    }
};
template <int M, int K, int CLAMP, int IM> struct OddInv {
    static __device__ __forceinline__ void run(int32_t *x, const int32_t *c, int bit) {
        if constexpr (K >= 1) {
            odd_bfly<M, K, CLAMP>(x);
            odd_rot<M, K, IM>(x, c, bit);
            OddInv<M, K - 1, CLAMP, IM>::run(x, c, bit);
        }
    }
};
// FM (forward multiply mode): 0 = 32-bit wrapping products (any input), 1 = full-rate 24-bit multiplies -- exact whenever every node of the
// pass fits 24 signed bits, which the callers establish from the block's largest residual (see fwd_mul24_safe)
template <int N, int FM> __device__ __forceinline__ void fdct_core(int32_t *x, const int32_t *c, int bit) {
    if constexpr (N == 2) {
        const int32_t a = x[0], b = x[1];
        x[0] = hbtf<FM>(c[32], a, c[32], b, bit);
        x[1] = hbtf<FM>(-c[32], b, c[32], a, bit);
    } else {
        constexpr int M = N / 2;
#pragma unroll
        for (int i = 0; i < M; i++) { const int32_t a = x[i], b = x[N - 1 - i]; x[i] = wadd(a, b); x[N - 1 - i] = wsub(a, b); }
        fdct_core<M, FM>(x, c, bit);
        OddFwd<M, 1, FM>::run(x, c, bit);
        odd_final<M, false, FM>(x, c, bit);
    }
}
// IM (inverse multiply mode): 1 = 24-bit products summed in 64 bits (any clamped input), 2 = the three-instruction butterflies, exact while
// every node of the pass is below 2^18 (weights <= 2^12): the callers measure the pass input
template <int N, int CLAMP, int IM> __device__ __forceinline__ void idct_core(int32_t *x, const int32_t *c, int bit) {
    if constexpr (N == 2) {
        const int32_t a = x[0], b = x[1];
        x[0] = hbtf<IM>(c[32], a, c[32], b, bit);
        x[1] = hbtf<IM>(c[32], a, -c[32], b, bit);
    } else {
        constexpr int M = N / 2;
        odd_final<M, true, IM>(x, c, bit);
        OddInv<M, ilog2c(M) - 1, CLAMP, IM>::run(x, c, bit);
        idct_core<M, CLAMP, IM>(x, c, bit);
#pragma unroll
        for (int i = 0; i < M; i++) { const int32_t a = x[i], b = x[N - 1 - i]; x[i] = clamp32(a + b, CLAMP); x[N - 1 - i] = clamp32(a - b, CLAMP); }
    }
}
template <int N> __device__ __forceinline__ void permute_brev(int32_t *x) { // out[k] = in[brev(k)]: an involution, swap pairs
#pragma unroll
    for (int k = 0; k < N; k++) {
        const int r = brevc(k, ilog2c(N));
        if (r > k) { const int32_t t = x[k]; x[k] = x[r]; x[r] = t; }
    }
}

// ---- ADST 8 / 16 (svt_av1_fadst8/16_new, iadst8/16_new) and ADST 4 ----
template <int N> __host__ __device__ constexpr int adst_perm(int k) { // P_N[2j] = P_{N/2}[j], P_N[2j+1] = N-1-P_{N/2}[j], P_2 = {0,1}
    if constexpr (N == 2) return k;
    else return (k & 1) ? N - 1 - adst_perm<N / 2>(k >> 1) : adst_perm<N / 2>(k >> 1);
}
template <int N, int HH, int MUL> __device__ __forceinline__ void adst_rot(int32_t *x, const int32_t *c, int bit) {
#pragma unroll
    for (int b = 0; b < N; b += 2 * HH) {
        if constexpr (HH == 2) {
            const int32_t a = x[b + 2], d = x[b + 3];
            x[b + 2] = hbtf<MUL>(c[32], a, c[32], d, bit);
            x[b + 3] = hbtf<MUL>(c[32], a, -c[32], d, bit);
        } else {
#pragma unroll
            for (int j = 0; j < HH / 4; j++) {
                const int A = (4 * j + 1) * (64 / HH), B = 64 - A, p = b + HH + 2 * j, q = b + HH + HH / 2 + 2 * j;
                int32_t a = x[p], d = x[p + 1];
                x[p]     = hbtf<MUL>(c[A], a, c[B], d, bit);
                x[p + 1] = hbtf<MUL>(c[B], a, -c[A], d, bit);
                a = x[q]; d = x[q + 1];
                x[q]     = hbtf<MUL>(-c[B], a, c[A], d, bit);
                x[q + 1] = hbtf<MUL>(c[A], a, c[B], d, bit);
            }
        }
    }
}
template <int N, int HH, int CLAMP> __device__ __forceinline__ void adst_bfly(int32_t *x) {
#pragma unroll
    for (int b = 0; b < N; b += 2 * HH)
#pragma unroll
        for (int j = 0; j < HH; j++) {
            const int32_t a = x[b + j], d = x[b + j + HH];
            if constexpr (CLAMP < 0) { x[b + j] = wadd(a, d); x[b + j + HH] = wsub(a, d); }
            else { x[b + j] = clamp32(a + d, CLAMP); x[b + j + HH] = clamp32(a - d, CLAMP); }
        }
}
template <int N, int MUL> __device__ __forceinline__ void adst_last(int32_t *x, const int32_t *c, int bit) {
#pragma unroll
    for (int j = 0; j < N / 2; j++) {
        const int A = (4 * j + 1) * (64 / (2 * N)), B = 64 - A;
        const int32_t a = x[2 * j], d = x[2 * j + 1];
        x[2 * j]     = hbtf<MUL>(c[A], a, c[B], d, bit);
        x[2 * j + 1] = hbtf<MUL>(c[B], a, -c[A], d, bit);
    }
}
template <int N, int HH, int FM> struct AdstFwd {
    static __device__ __forceinline__ void run(int32_t *x, const int32_t *c, int bit) {
        if constexpr (HH < N) { adst_rot<N, HH, FM>(x, c, bit); adst_bfly<N, HH, -1>(x); AdstFwd<N, HH * 2, FM>::run(x, c, bit); }
    }
};
template <int N, int HH, int CLAMP, int IM> struct AdstInv {
    static __device__ __forceinline__ void run(int32_t *x, const int32_t *c, int bit) {
        if constexpr (HH >= 2) { adst_bfly<N, HH, CLAMP>(x); adst_rot<N, HH, IM>(x, c, bit); AdstInv<N, HH / 2, CLAMP, IM>::run(x, c, bit); }
    }
};
template <int N, int FM> __device__ __forceinline__ void fadst(int32_t *x, const int32_t *c, int bit) {
    int32_t y[N];
#pragma unroll
    for (int k = 0; k < N; k++) { const int32_t v = x[adst_perm<N>(k)]; y[k] = (__builtin_popcount(k) & 1) ? (int32_t)(0u - (uint32_t)v) : v; }
    AdstFwd<N, 2, FM>::run(y, c, bit);
    adst_last<N, FM>(y, c, bit);
#pragma unroll
    for (int j = 0; j < N / 2; j++) { x[2 * j] = y[2 * j + 1]; x[2 * j + 1] = y[N - 2 - 2 * j]; }
}
template <int N, int CLAMP, int IM> __device__ __forceinline__ void iadst(int32_t *x, const int32_t *c, int bit) {
    int32_t y[N];
#pragma unroll
    for (int j = 0; j < N / 2; j++) { y[2 * j + 1] = x[2 * j]; y[N - 2 - 2 * j] = x[2 * j + 1]; }
    adst_last<N, IM>(y, c, bit);
    AdstInv<N, N / 2, CLAMP, IM>::run(y, c, bit);
#pragma unroll
    for (int k = 0; k < N; k++) x[adst_perm<N>(k)] = (__builtin_popcount(k) & 1) ? (int32_t)(0u - (uint32_t)y[k]) : y[k];
}
#define MUL32(a, b) ((int32_t)((uint32_t)(a) * (uint32_t)(b)))
__device__ __forceinline__ void adst4(int32_t *x, int bit, bool inverse) { // transforms.c:1415-1503, inv_transforms.c:722-806
    const int32_t *s = c_sinpi[bit - 10];
    const int32_t x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
    if (!(x0 | x1 | x2 | x3)) return;
    if (!inverse) {
        const int32_t s7 = wsub(wadd(x0, x1), x3);
        const int32_t a0 = wadd(wadd(MUL32(s[1], x0), MUL32(s[2], x1)), MUL32(s[4], x3)), a1 = MUL32(s[3], s7);
        const int32_t a2 = wadd(wsub(MUL32(s[4], x0), MUL32(s[1], x1)), MUL32(s[2], x3)), a3 = MUL32(s[3], x2);
        x[0] = rshift64(wadd(a0, a3), bit); x[1] = rshift64(a1, bit); x[2] = rshift64(wsub(a2, a3), bit); x[3] = rshift64(wadd(wsub(a2, a0), a3), bit);
    } else {
        const int32_t s7 = wadd(wsub(x0, x2), x3);
        const int32_t a0 = wadd(wadd(MUL32(s[1], x0), MUL32(s[4], x2)), MUL32(s[2], x3));
        const int32_t a1 = wsub(wsub(MUL32(s[2], x0), MUL32(s[1], x2)), MUL32(s[4], x3)), a3 = MUL32(s[3], x1), a2 = MUL32(s[3], s7);
        x[0] = rshift64(wadd(a0, a3), bit); x[1] = rshift64(wadd(a1, a3), bit); x[2] = rshift64(a2, bit); x[3] = rshift64(wsub(wadd(a0, a1), a3), bit);
    }
}
template <int N> __device__ __forceinline__ void identity(int32_t *x) { // transforms.c:2205-2236, inv_transforms.c:2331-2363
#pragma unroll
    for (int i = 0; i < N; i++) {
        if constexpr (N == 4) x[i] = rshift64((i64)x[i] * 5793, 12);
        else if constexpr (N == 8) x[i] = (int32_t)((uint32_t)x[i] * 2u);
        else if constexpr (N == 16) x[i] = rshift64((i64)x[i] * 2 * 5793, 12);
        else if constexpr (N == 32) x[i] = (int32_t)((uint32_t)x[i] * 4u);
        else x[i] = rshift64((i64)x[i] * 4 * 5793, 12);
    }
}
// 1-D dispatch: type 0 DCT, 1/2 ADST (flips are applied by the 2-D passes), 3 identity.  Wave-uniform switch.
template <int N, int FM> __device__ __forceinline__ void fwd_1d(int32_t *x, int type, int bit) {
    const int32_t *c = c_cospi[bit - 10];
    if (type == 3) identity<N>(x);
    else if (type == 0) { fdct_core<N, FM>(x, c, bit); permute_brev<N>(x); }
    else if constexpr (N == 4) adst4(x, bit, false);
    else if constexpr (N <= 16) fadst<N, FM>(x, c, bit);
}
// A block's forward transform may use the 24-bit multiplies when |residual| <= 4095: a node of a 1-D pass is a sum of at most N inputs of that
// pass with weights of magnitude <= 1, so with the up-shift of at most 2 in front of the column pass and the down-shifts between the passes
// (fwd_txfm_shift_ls, transforms.h:27-45) no node of any of the 19 sizes exceeds 64 * 65,520 < 2^22; the cosine weights are < 2^14.  Any
// 8- / 10-bit picture satisfies it; samples outside the bit depth (the reference accepts any uint16) take the 32-bit path.
// The three-instruction butterflies (hbtf<2>) need every node of the pass below 2^17: N x the largest input of the pass (measured over the wave).
__device__ __forceinline__ bool pass_fits_17_bits(uint32_t wave_max_abs_input, int n) { return (unsigned long long)wave_max_abs_input * (unsigned)n < (1u << 17); }
template <int N, int CLAMP, int IM> __device__ __forceinline__ void inv_1d(int32_t *x, int type) {
    const int32_t *c = c_cospi[2]; // INV_COS_BIT = 12
    if (type == 3) identity<N>(x);
    else if (type == 0) { permute_brev<N>(x); idct_core<N, CLAMP, IM>(x, c, 12); }
    else if constexpr (N == 4) adst4(x, 12, true);
    else if constexpr (N <= 16) iadst<N, CLAMP, IM>(x, c, 12);
}
// every node of an inverse pass is a sum of at most N pass inputs with weights of magnitude <= 1 (the stage clamps only shrink it): below 2^18
// when N x the largest |input| is
__device__ __forceinline__ bool ipass_fits_18_bits(uint32_t wave_max_abs_input, int n) { return (unsigned long long)wave_max_abs_input * (unsigned)n < (1u << 18); }
// largest magnitude of a vector: the largest and the smallest element are tracked instead (one v_max3_i32 / v_min3_i32 per PAIR of elements
// each; |x| first would be two instructions per element before the max)
struct HiLo {
    int32_t hi = 0, lo = 0;
    __device__ __forceinline__ void take(int32_t a, int32_t b) { hi = max(max(hi, a), b); lo = min(min(lo, a), b); }
    __device__ __forceinline__ void take(int32_t a) { hi = max(hi, a); lo = min(lo, a); }
    __device__ __forceinline__ uint32_t max_abs() const { return (uint32_t)max(hi, -lo); }
};
template <int N> __device__ __forceinline__ uint32_t vec_max_abs(const int32_t *x) {
    HiLo m;
    if constexpr (N % 2 == 0) {
#pragma unroll
        for (int i = 0; i < N; i += 2) m.take(x[i], x[i + 1]);
    } else {
#pragma unroll
        for (int i = 0; i < N; i++) m.take(x[i]);
    }
    return m.max_abs();
}
// SAFE32: the caller knows |x| < 2^30 (a pass that took the bounded butterflies, or an inverse pass, whose outputs are clamped to <= 18 bits):
// the rounding add cannot wrap and the reference's 64-bit round_shift is the same two 32-bit instructions
template <int N, bool SAFE32 = false> __device__ __forceinline__ void shift_vec(int32_t *x, int sh) { // svt_av1_round_shift_array_c(x, N, -sh)
    if (sh < 0) {
#pragma unroll
        for (int i = 0; i < N; i++) x[i] = SAFE32 ? ((x[i] + (1 << (-sh - 1))) >> -sh) : rshift64(x[i], -sh);
    } else if (sh > 0) {
#pragma unroll
        for (int i = 0; i < N; i++) x[i] = (int32_t)((uint32_t)x[i] << sh);
    }
}

__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
    return v;
}

struct RdParams {
    SvtHipRdBatchDesc d;
    const int16_t    *iscan[3]; // default, row (V_*), column (H_*) scans of this tx_size
    int               tx_size;
};

template <typename T> __device__ __forceinline__ int ldpix(const void *p, size_t i) { return (int)static_cast<const T *>(p)[i]; }

// reductions over the LW lanes (a power of two) that work on one block
template <int LW> __device__ __forceinline__ u64 seg_sum_u64(u64 v) {
#pragma unroll
    for (int o = LW / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int LW> __device__ __forceinline__ uint32_t seg_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = LW / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int LW> __device__ __forceinline__ uint32_t seg_max_u32(uint32_t v) {
#pragma unroll
    for (int o = LW / 2; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
    return v;
}

__host__ __device__ constexpr int rd_lanes_per_block(int ts) { return tx_wide(ts) > tx_high(ts) ? tx_wide(ts) : tx_high(ts); }
__host__ __device__ constexpr int rd_blocks_per_wave(int ts) { return 64 / rd_lanes_per_block(ts); }

// =========================================================================================================
// One wave64 per workgroup; the wave holds 64 / max(W, H) transform blocks side by side (lane = block * LW + l), so
// the 1-D passes keep every lane busy for all sizes.  A block lives in ONE LDS array with row pitch W + 1 dwords:
// column reads (stride 1) and row reads (stride W + 1) are both bank-conflict free.  The passes run in place --
// a wave executes its LDS loads before the stores that follow them in program order -- and the packed
// (top-left 32x32) coefficients of the 64-point sizes are compacted in place as well, which is what lets nine
// 64x64 workgroups share a CU's LDS.
// =========================================================================================================
// resident waves per SIMD the register allocation aims for: enough workgroups in flight that a picture's blocks of one size
// run as a single round (e.g. 32x32: 4,020 two-block workgroups on 256 CUs x 4 SIMDs x 4 waves)
#ifndef SVT_RD_WAVES_64
#define SVT_RD_WAVES_64 2
#endif
#ifndef SVT_RD_WAVES_32
#define SVT_RD_WAVES_32 4
#endif
#ifndef SVT_RD_RESIDUAL_AHEAD
#define SVT_RD_RESIDUAL_AHEAD 8 /* runs of 8 samples a lane has in flight while it forms the residual */
#endif
#ifndef SVT_RD_SCAN_AHEAD
#define SVT_RD_SCAN_AHEAD 16 /* coefficients per lane whose scan positions are fetched ahead of the quantizer loop */
#endif
#ifndef SVT_RD_WAVES_16
#define SVT_RD_WAVES_16 4
#endif
__host__ __device__ constexpr int rd_waves_per_simd(int ts) {
    return rd_lanes_per_block(ts) == 64 ? SVT_RD_WAVES_64 : rd_lanes_per_block(ts) == 32 ? SVT_RD_WAVES_32 : SVT_RD_WAVES_16;
}
template <int TS, int BD> __global__ void __launch_bounds__(64, rd_waves_per_simd(TS)) rd_tx_kernel(const RdParams p) {
    constexpr int W = tx_wide(TS), H = tx_high(TS), WP = W > 32 ? 32 : W, HP = H > 32 ? 32 : H, NP = WP * HP;
    constexpr int LW = rd_lanes_per_block(TS), BPW = rd_blocks_per_wave(TS);
    constexpr bool kShift32 = LW != 32; // bounded passes round-shift in 32 bits (measured: 64x64 -4 %; the 32-lane layout, at its 128-register limit, spills on it: +12 %)
    constexpr int PA = W + 1, PB = WP + 1; // row pitches (dwords) of the full block and of the packed coefficients
    constexpr int ROW_CLAMP = BD == 8 ? 16 : 18, COL_CLAMP = 16; // svt_av1_gen_inv_stage_range, inv_transforms.c:42-80
    constexpr bool RECT = (W == 2 * H || H == 2 * W);
    using Pix = typename std::conditional<BD == 8, uint8_t, uint16_t>::type;
    __shared__ int32_t lds[BPW][H * PA];
    const int      lane = threadIdx.x, blk = lane / LW, l = lane % LW;
    const uint32_t job   = blockIdx.x * BPW + blk;
    const bool     valid = job < p.d.n_jobs; // lanes of a missing block run along on job 0 and write nothing
    int32_t *A = lds[blk];
    const SvtHipTxJob jb = p.d.jobs[valid ? job : 0];
    const int tt = jb.tx_type & 15, vt = c_vtx[tt], ht = c_htx[tt];
    const bool ud = (vt == 2), lr = (ht == 2);
    const Pix *src  = static_cast<const Pix *>(p.d.src) + jb.src_offset;
    const Pix *pred = static_cast<const Pix *>(p.d.pred) + jb.pred_offset;
    const int8_t *fsh = c_fwd_shift[TS];
    const int bit_col = c_fwd_cos_col[ilog2c(W) - 2][ilog2c(H) - 2], bit_row = c_fwd_cos_row[ilog2c(W) - 2][ilog2c(H) - 2];

    // residual (svt_residual_kernel8bit / 16bit): int16 arithmetic as in the reference
    // runs of RUN samples per (unaligned) vector load: the planes' offsets and strides are the caller's
    constexpr int RUN = W < 8 ? W : 8, RPR = W / RUN; // run length, runs per row
    typedef Pix __attribute__((ext_vector_type(RUN), aligned(sizeof(Pix)))) RunU;
    uint32_t rmax = 0; // largest |residual| this lane produced
    {
        HiLo rhl;
        // a lane's runs are fetched kResidualAhead at a time, all of a group in flight before the first is used (the loop's trip count depends
        // on the lane: left to itself every run is a round trip of its own)
        constexpr int NIT = (RPR * H + LW - 1) / LW, kResidualAhead = NIT < SVT_RD_RESIDUAL_AHEAD ? NIT : SVT_RD_RESIDUAL_AHEAD;
        for (int it0 = 0; it0 < NIT; it0 += kResidualAhead) {
            RunU sv[kResidualAhead], pv[kResidualAhead];
#pragma unroll
            for (int j = 0; j < kResidualAhead; j++) {
                const int i = l + (it0 + j) * LW;
                if (it0 + j < NIT && i < RPR * H) {
                    const int r = i / RPR, c = (i - r * RPR) * RUN;
                    sv[j] = *reinterpret_cast<const RunU *>(src + (size_t)r * p.d.src_stride + c);
                    pv[j] = *reinterpret_cast<const RunU *>(pred + (size_t)r * p.d.pred_stride + c);
                }
            }
#pragma unroll
            for (int j = 0; j < kResidualAhead; j++) {
                const int i = l + (it0 + j) * LW;
                if (it0 + j < NIT && i < RPR * H) {
                    const int r = i / RPR, c = (i - r * RPR) * RUN;
#pragma unroll
                    for (int k = 0; k < RUN; k++) {
                        const int32_t d = (int16_t)((int16_t)sv[j][k] - (int16_t)pv[j][k]);
                        A[r * PA + c + k] = d;
                        rhl.take(d);
                    }
                }
            }
        }
        rmax = rhl.max_abs();
    }
    // one answer for the wave (all its blocks): every residual small enough for the 24-bit multiplies of the forward passes?
    // one answer for the wave (all its blocks): small enough data for the three-instruction butterflies?  The column pass sees the residual
    // shifted up by fsh[0] (0 or 2)
    const bool fast_col = pass_fits_17_bits((uint32_t)__builtin_amdgcn_readfirstlane((int)seg_max_u32<64>(rmax)) << (fsh[0] > 0 ? fsh[0] : 0), H);
    __syncthreads();
    // forward columns (av1_tranform_two_d_core_c, transforms.c:2287-2308)
    uint32_t cmax = 0; // largest |column output| of this lane
    if (l < W) {
        int32_t x[H];
#pragma unroll
        for (int r = 0; r < H; r++) x[r] = A[(ud ? H - 1 - r : r) * PA + l];
        shift_vec<H>(x, fsh[0]);
        if (fast_col) { fwd_1d<H, 2>(x, vt, bit_col); shift_vec<H, kShift32>(x, fsh[1]); } else { fwd_1d<H, 0>(x, vt, bit_col); shift_vec<H>(x, fsh[1]); } // wave-uniform
        const int oc = lr ? W - 1 - l : l;
#pragma unroll
        for (int r = 0; r < H; r++) A[r * PA + oc] = x[r];
        cmax = vec_max_abs<H>(x);
    }
    const bool fast_row = pass_fits_17_bits((uint32_t)__builtin_amdgcn_readfirstlane((int)seg_max_u32<64>(cmax)), W); // the row pass's input
    __syncthreads();
    // forward rows (:2310-2323)
    uint32_t comax = 0; // largest |coefficient| of this lane
    if (l < H) {
        int32_t x[W];
#pragma unroll
        for (int c = 0; c < W; c++) x[c] = A[l * PA + c];
        if (fast_row) { fwd_1d<W, 2>(x, ht, bit_row); shift_vec<W, kShift32>(x, fsh[2]); } else { fwd_1d<W, 0>(x, ht, bit_row); shift_vec<W>(x, fsh[2]); }
        if constexpr (RECT) {
#pragma unroll
            for (int c = 0; c < W; c++) x[c] = rshift64((i64)x[c] * 5793, 12);
        }
#pragma unroll
        for (int c = 0; c < W; c++) A[l * PA + c] = x[c];
        comax = vec_max_abs<W>(x);
    }
    // coefficients below 2^16 in every block of the wave: the quantizer's products fit 24-bit multiplies (see the loop below)
    const bool q24 = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg_max_u32<64>(comax)) < (1u << 16);
    __syncthreads();
    // 64-point sizes keep the top-left 32x32 (svt_handle_transform*_c, transforms.c:2374-2505)
    // partial-frequency shapes (av1_estimate_transform_N2 / _N4 / _ONLY_DC, transforms.c:2633-2948): the pruned 1-D kernels
    // of the reference produce the full transform's low-frequency outputs; everything else is zero and no energy is
    // attributed to the discarded frequencies (svt_handle_transform*_N2_N4_c, :2514-2543)
    const int pf = jb.pf_shape & 3;
    const int keep_w = pf == 3 ? 1 : (W >> pf), keep_h = pf == 3 ? 1 : (H >> pf);
    u64 tq = 0;
    if (pf == 0) if constexpr (W > 32 || H > 32) {
        for (int i = l; i < W * H; i += LW) {
            const int r = i / W, c = i - r * W;
            if (r >= HP || c >= WP) { const int32_t v = A[r * PA + c]; tq += (u64)((i64)v * v); }
        }
        tq = seg_sum_u64<LW>(tq);
    }
    if constexpr (W > 32 || H > 32) __syncthreads(); // the compaction below overwrites discarded coefficients
    // SATD, quantize, coefficient-domain distortion over the kept NP coefficients (packed index rc = r*WP + c).  The
    // dequantized value replaces the coefficient at the packed position r*PB + c <= r*PA + c: a later iteration never
    // reads what an earlier one overwrote (its reads start beyond the earlier iteration's writes), and within one
    // iteration the wave's loads precede its stores.
    const SvtHipQuantRow q = p.d.quant_rows[jb.quant_row];
    const int     log_scale = p.d.quant_kind == 2 ? 0 : c_log_scale[TS]; // quant_kind 2: the TPL dispenser's plain svt_av1_quantize_fp call (src_ops_process.c:225-249)
    const int16_t *iscan    = p.iscan[(tt >= 10) ? ((tt & 1) ? 2 : 1) : 0];
    const uint8_t *qm = (tt < 9) ? p.d.qmatrix : nullptr, *iqm = (tt < 9) ? p.d.iqmatrix : nullptr; // IS_2D_TRANSFORM, full_loop.c:1606-1608
    uint32_t satd = 0, eob = 0, qsum = 0; // qsum: sum of min(|qcoeff|, 63) = svt_av1_compute_cul_level's running sum up to its clamp
    int32_t  dc_q = 0;                    // lane 0 of a block: the quantized DC coefficient (rc == 0 is its first iteration)
    u64      dres = 0, dpred = 0;
    int32_t *co_out = (p.d.coeff && valid) ? p.d.coeff + (size_t)job * NP : nullptr;
    int32_t *q_out  = (p.d.qcoeff && valid) ? p.d.qcoeff + (size_t)job * NP : nullptr;
    int32_t *dq_out = (p.d.dqcoeff && valid) ? p.d.dqcoeff + (size_t)job * NP : nullptr;
    const int32_t zb_c[2]  = {log_scale ? ((q.zbin[0] + (1 << (log_scale - 1))) >> log_scale) : q.zbin[0], log_scale ? ((q.zbin[1] + (1 << (log_scale - 1))) >> log_scale) : q.zbin[1]};
    const int32_t rnd_c[2] = {log_scale ? ((q.round[0] + (1 << (log_scale - 1))) >> log_scale) : q.round[0], log_scale ? ((q.round[1] + (1 << (log_scale - 1))) >> log_scale) : q.round[1]};
    // The common case as a loop of its own (wave-uniform test): "b" quantizer, flat matrix, every coefficient of the wave below 2^16, the whole
    // block kept, nothing but the quantized coefficients asked for.  A lane's first coefficient may be the DC one (its constants are
    // selected per lane); all later ones are AC, whose constants stay in scalar registers.  Every product has 24-bit operands (see the
    // general loop below), |coeff - dqcoeff| <= max(|coeff|, dequant) < 2^16: the squares are exact in 32 bits.
    const bool fast_q = q24 && !qm && p.d.quant_kind == 0 && !co_out && !dq_out && __all(pf == 0); // (co_out / dq_out: null for every job or none)
    if (fast_q) {
        static_assert(NP % LW == 0, "every lane of a block walks the same number of coefficients");
        // The scan positions of a lane's first kScanAhead coefficients (the low frequencies: where the non-zero levels are) are fetched ahead,
        // unconditionally, two to a register.  A load behind the `qv != 0` branch waits with s_waitcnt vmcnt(0) -- which on this ISA also
        // waits for the coefficient STORES of the iterations before it: the loop stalled on store completion exactly where it has work.
        constexpr int kScanAhead = NP / LW < SVT_RD_SCAN_AHEAD ? ((NP / LW) & ~1) : SVT_RD_SCAN_AHEAD;
        static_assert(kScanAhead >= 2 && kScanAhead % 2 == 0 && NP / LW >= kScanAhead, "ahead of the loop");
        uint32_t scan2[kScanAhead / 2];
#pragma unroll
        for (int k = 0; k < kScanAhead; k += 2) scan2[k / 2] = (uint32_t)(uint16_t)iscan[l + LW * k] | ((uint32_t)(uint16_t)iscan[l + LW * (k + 1)] << 16);
        auto one = [&](int rc, int32_t zb, int32_t rnd, int32_t quant, int32_t qshift, int32_t deq, int ahead = -1) {
            const int r = rc / WP, c = rc - r * WP; // WP is a power of two
            const int32_t co = A[r * PA + c], sign = co >> 31, a = (co ^ sign) - sign;
            satd += (uint32_t)a;
            int32_t t = a + rnd;
            if (BD == 8) t = t > 32767 ? 32767 : t;
            const int32_t tmp = mul24_shr(t, quant, 11) + (t << 5);
            int32_t qv = mul24_shr(tmp, qshift, 21 - log_scale);
            qv = a >= zb ? qv : 0;
            const int32_t dq = __mul24(qv, deq) >> log_scale;
            const int32_t qs = (qv ^ sign) - sign, dqs = (dq ^ sign) - sign;
            uint32_t e;
            if (ahead >= 0) e = qv ? ((scan2[ahead >> 1] >> (16 * (ahead & 1))) & 0xFFFFu) + 1u : 0u; // (compile-time `ahead`)
            else e = qv ? (uint32_t)iscan[rc] + 1u : 0u;
            eob = e > eob ? e : eob;
            qsum += (uint32_t)(qv > 63 ? 63 : qv);
            const int32_t dd = a - dq; // == |coeff - dqcoeff|: both carry the coefficient's sign
            dres += (u64)(uint32_t)__mul24(dd, dd);
            dpred += (u64)(uint32_t)__umul24((uint32_t)a, (uint32_t)a);
            A[r * PB + c] = dqs;
            if (q_out) q_out[rc] = qs;
            return qs;
        };
        const int ac0 = l != 0;
        dc_q = one(l, zb_c[ac0], rnd_c[ac0], q.quant[ac0], q.quant_shift[ac0], q.dequant[ac0], 0);
        const int32_t zb1 = zb_c[1], rnd1 = rnd_c[1], quant1 = q.quant[1], qshift1 = q.quant_shift[1], deq1 = q.dequant[1];
#pragma unroll
        for (int k = 1; k < kScanAhead; k++) one(l + LW * k, zb1, rnd1, quant1, qshift1, deq1, k);
        if constexpr (LW >= 32) {
            // two coefficients per trip: two independent chains for the scheduler.
            // Measured 64x64 0.718 -> 0.707 ms, 32x32 0.623 -> 0.611; at 16x16 the two extra registers cross an occupancy step (0.540 -> 0.564).
            static_assert((NP / LW - kScanAhead) % 2 == 0, "pairs");
            for (int rc = l + kScanAhead * LW; rc < NP; rc += 2 * LW) {
                one(rc, zb1, rnd1, quant1, qshift1, deq1);
                one(rc + LW, zb1, rnd1, quant1, qshift1, deq1);
            }
        } else {
#pragma unroll 4
            for (int rc = l + kScanAhead * LW; rc < NP; rc += LW) one(rc, zb1, rnd1, quant1, qshift1, deq1);
        }
    } else
    for (int rc = l; rc < NP; rc += LW) {
        const int r = rc / WP, c = rc - r * WP, ac = rc != 0;
        const bool kept = pf == 0 || (c < keep_w && r < keep_h);
        const int32_t co = kept ? A[r * PA + c] : 0, sign = co < 0 ? -1 : 0, a = (co ^ sign) - sign;
        satd += (uint32_t)a;
        int32_t qv = 0, dq = 0;
        const int32_t wt = qm ? qm[rc] : 32, iwt = qm ? iqm[rc] : 32; // AOM_QM_BITS = 5
        if (q24 && !qm) { // either quantizer with the flat matrix while |coeff| < 2^16 (wave-uniform): every product has 24-bit operands
            // "b": t = |coeff| + round < 2^17; (t << 5) * quant >> 16 == t * quant >> 11; tmp = that + (t << 5) lies in t * [16, 48) < 2^23
            if (p.d.quant_kind == 0) {
                if (a >= zb_c[ac]) {
                    int32_t t = a + rnd_c[ac];
                    if (BD == 8) t = t > 32767 ? 32767 : t;
                    const int32_t tmp = mul24_shr(t, q.quant[ac], 11) + (t << 5);
                    qv = mul24_shr(tmp, q.quant_shift[ac], 21 - log_scale);
                    dq = __mul24(qv, (int32_t)q.dequant[ac]) >> log_scale;
                }
            } else { // "fp": (|coeff| + round_fp) * quant_fp >> (16 - log_scale)
                const bool keep = (a << (1 + log_scale)) >= q.dequant[ac];
                if (keep) {
                    int32_t t = a + (log_scale ? ((q.round_fp[ac] + (1 << (log_scale - 1))) >> log_scale) : q.round_fp[ac]);
                    if (BD == 8) t = t < -32768 ? -32768 : (t > 32767 ? 32767 : t);
                    qv = mul24_shr(t, q.quant_fp[ac], 16 - log_scale);
                    dq = __mul24(qv, (int32_t)q.dequant[ac]) >> log_scale;
                }
            }
        } else if (p.d.quant_kind == 0 && !qm) { // the same "b" quantizer with the flat matrix (wt = iwt = 32), in 32-bit arithmetic:
            // |coeff| < 2^24 for any int16 residual (forward gain <= N per pass, minus the stage shifts), so (|coeff| + round) << 5
            // and its products' high parts fit 32 bits; every intermediate equals the reference's 64-bit value
            const int32_t zb = zb_c[ac];
            if (a >= zb) {
                int32_t t = a + rnd_c[ac];
                if (BD == 8) t = t > 32767 ? 32767 : t;
                const int32_t tw  = t << 5;
                const int32_t tmp = (int32_t)(((i64)tw * q.quant[ac]) >> 16) + tw;
                qv = (int32_t)(((i64)tmp * q.quant_shift[ac]) >> (16 - log_scale + 5));
                dq = (qv * (int32_t)q.dequant[ac]) >> log_scale;
            }
        } else if (p.d.quant_kind == 0) { // svt_aom_quantize_b_c_ii / svt_aom_highbd_quantize_b_c (full_loop.c:29-79,149-198)
            const int32_t zb = log_scale ? ((q.zbin[ac] + (1 << (log_scale - 1))) >> log_scale) : q.zbin[ac];
            if ((i64)a * wt >= ((i64)zb << 5)) {
                i64 t = (i64)a + (log_scale ? ((q.round[ac] + (1 << (log_scale - 1))) >> log_scale) : q.round[ac]);
                if (BD == 8) t = t < -32768 ? -32768 : (t > 32767 ? 32767 : t);
                t *= wt;
                qv = (int32_t)(((((t * q.quant[ac]) >> 16) + t) * q.quant_shift[ac]) >> (16 - log_scale + 5));
                const int32_t deq = ((int32_t)q.dequant[ac] * iwt + 16) >> 5;
                dq = (qv * deq) >> log_scale;
            }
        } else if (!qm) { // quantize_fp_helper_c / highbd_quantize_fp_helper_c without matrices (full_loop.c:282-343,387-452)
            const bool keep = (BD == 8) ? (((i64)a << (1 + log_scale)) >= (int32_t)q.dequant[ac]) : ((a << (1 + log_scale)) >= q.dequant[ac]);
            if (keep) {
                i64 t = (i64)a + (log_scale ? ((q.round_fp[ac] + (1 << (log_scale - 1))) >> log_scale) : q.round_fp[ac]);
                if (BD == 8) t = t < -32768 ? -32768 : (t > 32767 ? 32767 : t);
                qv = (int32_t)((t * q.quant_fp[ac]) >> (16 - log_scale));
                dq = (qv * (int32_t)q.dequant[ac]) >> log_scale;
            }
        } else { // the same helpers' matrix branch
            if ((i64)a * wt >= ((int32_t)q.dequant[ac] << (5 - (1 + log_scale)))) {
                i64 t = (i64)a + (log_scale ? ((q.round_fp[ac] + (1 << (log_scale - 1))) >> log_scale) : q.round_fp[ac]);
                if (BD == 8) t = t < -32768 ? -32768 : (t > 32767 ? 32767 : t);
                qv = (int32_t)((t * q.quant_fp[ac] * wt) >> (16 - log_scale + 5));
                const int32_t deq = ((int32_t)q.dequant[ac] * iwt + 16) >> 5;
                dq = (qv * deq) >> log_scale;
            }
        }
        const int32_t qs = (qv ^ sign) - sign, dqs = (dq ^ sign) - sign;
        if (qv) { const uint32_t e = (uint32_t)iscan[rc] + 1; eob = e > eob ? e : eob; }
        qsum += (uint32_t)(qv > 63 ? 63 : qv);
        if (rc == 0) dc_q = qs;
        const i64 dd = (i64)co - dqs;
        dres += (u64)(dd * dd);
        dpred += (u64)((i64)co * co);
        A[r * PB + c] = dqs; // packed dequantized coefficients feed the inverse transform
        if (co_out) co_out[rc] = co;
        if (q_out) q_out[rc] = qs;
        if (dq_out) dq_out[rc] = dqs;
    }
    satd  = seg_sum_u32<LW>(satd);
    qsum  = seg_sum_u32<LW>(qsum);
    eob   = seg_max_u32<LW>(eob);
    dres  = seg_sum_u64<LW>(dres);
    dpred = seg_sum_u64<LW>(dpred);
    __syncthreads();
    // inverse rows (inv_txfm2d_add_c, inv_transforms.c:2497-2511): discarded frequencies are zero
    int32_t xr[W];
    uint32_t irmax = 0; // largest |row-pass input| of this lane
    if (l < H) {
#pragma unroll
        for (int c = 0; c < W; c++) {
            int32_t v = (l < HP && c < WP) ? A[l * PB + c] : 0;
            if constexpr (RECT) v = rshift64((i64)v * 2896, 12);
            xr[c] = clampv(v, BD + 8);
        }
        irmax = vec_max_abs<W>(xr);
    }
    const bool fast_irow = ipass_fits_18_bits((uint32_t)__builtin_amdgcn_readfirstlane((int)seg_max_u32<64>(irmax)), W); // wave-uniform
    uint32_t icmax = 0; // largest |row-pass output|: the column pass's input
    if (l < H) {
        if (fast_irow) inv_1d<W, ROW_CLAMP, 2>(xr, ht); else inv_1d<W, ROW_CLAMP, 1>(xr, ht);
        shift_vec<W, kShift32>(xr, c_inv_shift0[TS]);
#pragma unroll
        for (int c = 0; c < W; c++) A[l * PA + c] = xr[c];
        icmax = vec_max_abs<W>(xr);
    }
    const bool fast_icol = ipass_fits_18_bits((uint32_t)__builtin_amdgcn_readfirstlane((int)seg_max_u32<64>(icmax)), H);
    __syncthreads();
    // inverse columns + reconstruction + SSE (:2513-2534; svt_spatial_full_distortion_kernel / 16-bit variant)
    u64 sse = 0;
    if (l < W) {
        int32_t x[H];
        const int ic = lr ? W - 1 - l : l;
#pragma unroll
        for (int r = 0; r < H; r++) x[r] = (r < HP) ? clampv(A[r * PA + ic], BD + 6 > 16 ? BD + 6 : 16) : 0; // rows >= 32 of a 64-row block are exactly zero after the row pass: constants, so the network prunes itself
        if (fast_icol) inv_1d<H, COL_CLAMP, 2>(x, vt); else inv_1d<H, COL_CLAMP, 1>(x, vt);
        shift_vec<H, kShift32>(x, -4);
        // residual (with the vertical flip undone) back to LDS, row-major: the reconstruction below moves whole runs
#pragma unroll
        for (int r = 0; r < H; r++) A[r * PA + l] = x[ud ? H - 1 - r : r];
    }
    __syncthreads();
    {
        Pix *rec = (p.d.recon && valid) ? static_cast<Pix *>(p.d.recon) + jb.pred_offset : nullptr;
        // every run's prediction and source samples are requested BEFORE the first reconstruction run is stored: vector memory operations
        // complete in issue order on this ISA, so a load issued behind a store waits for the store's acknowledgement as well
        constexpr int NIT = (RPR * H + LW - 1) / LW; // runs per lane
        RunU pv[NIT], sv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int i = l + it * LW;
            if (i < RPR * H) {
                const int r = i / RPR, c = (i - r * RPR) * RUN;
                pv[it] = *reinterpret_cast<const RunU *>(pred + (size_t)r * p.d.pred_stride + c);
                sv[it] = *reinterpret_cast<const RunU *>(src + (size_t)r * p.d.src_stride + c);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int i = l + it * LW;
            if (i < RPR * H) {
                const int r = i / RPR, c = (i - r * RPR) * RUN;
                RunU out;
#pragma unroll
                for (int k = 0; k < RUN; k++) {
                    int v = (int)pv[it][k] + A[r * PA + c + k];
                    v = v < 0 ? 0 : (v > (1 << BD) - 1 ? (1 << BD) - 1 : v);
                    out[k] = (Pix)v;
                    const int e = (int)sv[it][k] - v;
                    sse += (u64)((uint32_t)e * (uint32_t)e); // |e| < 2^16: the low 32 bits of the (wrapping) product are the square, whatever the sign
                }
                if (rec) *reinterpret_cast<RunU *>(rec + (size_t)r * p.d.pred_stride + c) = out;
            }
        }
    }
    sse = seg_sum_u64<LW>(sse);
    if (l == 0 && valid) {
        p.d.eob[job]  = (uint16_t)eob;
        p.d.satd[job] = satd;
        p.d.dist_coeff[2 * (size_t)job]     = dres;
        p.d.dist_coeff[2 * (size_t)job + 1] = dpred;
        p.d.three_quad_energy[job] = tq;
        p.d.sse[job]  = sse;
        // svt_av1_compute_cul_level (full_loop.c:1449-1466): coefficients past eob are zero, so the sum over the scan equals the sum over
        // the block; set_dc_sign (:1338-1343)
        if (p.d.cul_level) p.d.cul_level[job] = (uint8_t)((qsum > 63 ? 63 : qsum) | (dc_q < 0 ? 64 : 0)) + (uint8_t)(dc_q > 0 ? 128 : 0);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Inverse transform + reconstruction alone (inv_txfm2d_add_c, inv_transforms.c:2459-2535; svt_av1_inv_txfm2d_add_{W}x{H}_c
// :2545-2716): the tail of rd_tx_kernel fed with caller-supplied dequantized coefficients (packed min(W,32) x min(H,32), as
// the reference's own inverse entries take them).  Same wave layout and LDS use as rd_tx_kernel.
// ---------------------------------------------------------------------------------------------------------
struct InvParams {
    SvtHipInvTxBatchDesc d;
};
template <int TS, int BD, typename Pix> __global__ void __launch_bounds__(64, rd_waves_per_simd(TS)) inv_tx_kernel(const InvParams p) {
    constexpr int W = tx_wide(TS), H = tx_high(TS), WP = W > 32 ? 32 : W, HP = H > 32 ? 32 : H, NP = WP * HP;
    constexpr int LW = rd_lanes_per_block(TS), BPW = rd_blocks_per_wave(TS);
    constexpr bool kShift32 = LW != 32;
    constexpr int PA = W + 1, PB = WP + 1;
    constexpr int ROW_CLAMP = BD == 8 ? 16 : 18, COL_CLAMP = 16;
    constexpr bool RECT = (W == 2 * H || H == 2 * W);
    __shared__ int32_t lds[BPW][H * PA];
    const int      lane = threadIdx.x, blk = lane / LW, l = lane % LW;
    const uint32_t job   = blockIdx.x * BPW + blk;
    const bool     valid = job < p.d.n_jobs;
    int32_t *A = lds[blk];
    const SvtHipTxJob jb = p.d.jobs[valid ? job : 0];
    const int tt = jb.tx_type & 15, vt = c_vtx[tt], ht = c_htx[tt];
    const bool ud = (vt == 2), lr = (ht == 2);
    const Pix *pred = static_cast<const Pix *>(p.d.pred) + jb.pred_offset;
    const int32_t *dq = p.d.dqcoeff + (size_t)(valid ? job : 0) * NP;
    for (int rc = l; rc < NP; rc += LW) { const int r = rc / WP, c = rc - r * WP; A[r * PB + c] = dq[rc]; }
    __syncthreads();
    int32_t xr[W];
    uint32_t irmax = 0; // largest |row-pass input| of this lane
    if (l < H) { // inverse rows (:2497-2511)
#pragma unroll
        for (int c = 0; c < W; c++) {
            int32_t v = (l < HP && c < WP) ? A[l * PB + c] : 0;
            if constexpr (RECT) v = rshift64((i64)v * 2896, 12);
            xr[c] = clampv(v, BD + 8);
        }
        irmax = vec_max_abs<W>(xr);
    }
    const bool fast_irow = ipass_fits_18_bits((uint32_t)__builtin_amdgcn_readfirstlane((int)seg_max_u32<64>(irmax)), W); // wave-uniform
    uint32_t icmax = 0; // largest |row-pass output|: the column pass's input
    if (l < H) {
        if (fast_irow) inv_1d<W, ROW_CLAMP, 2>(xr, ht); else inv_1d<W, ROW_CLAMP, 1>(xr, ht);
        shift_vec<W, kShift32>(xr, c_inv_shift0[TS]);
#pragma unroll
        for (int c = 0; c < W; c++) A[l * PA + c] = xr[c];
        icmax = vec_max_abs<W>(xr);
    }
    const bool fast_icol = ipass_fits_18_bits((uint32_t)__builtin_amdgcn_readfirstlane((int)seg_max_u32<64>(icmax)), H);
    __syncthreads();
    if (l < W) { // inverse columns (:2513-2534)
        int32_t x[H];
        const int ic = lr ? W - 1 - l : l;
#pragma unroll
        for (int r = 0; r < H; r++) x[r] = (r < HP) ? clampv(A[r * PA + ic], BD + 6 > 16 ? BD + 6 : 16) : 0;
        if (fast_icol) inv_1d<H, COL_CLAMP, 2>(x, vt); else inv_1d<H, COL_CLAMP, 1>(x, vt);
        shift_vec<H, kShift32>(x, -4);
#pragma unroll
        for (int r = 0; r < H; r++) A[r * PA + l] = x[ud ? H - 1 - r : r];
    }
    __syncthreads();
    if (valid) {
        Pix *rec = static_cast<Pix *>(p.d.recon) + jb.src_offset;
        for (int i = l; i < W * H; i += LW) {
            const int r = i / W, c = i - r * W;
            int v = (int)pred[(size_t)r * p.d.pred_stride + c] + A[r * PA + c];
            v = v < 0 ? 0 : (v > (1 << BD) - 1 ? (1 << BD) - 1 : v);
            rec[(size_t)r * p.d.recon_stride + c] = (Pix)v;
        }
    }
}

template <int BD, typename Pix> int launch_inv(SvtHipContext *ctx, const InvParams &p) {
    const dim3 b(64);
#define CASE(S) case S: hipLaunchKernelGGL((inv_tx_kernel<S, BD, Pix>), dim3((p.d.n_jobs + rd_blocks_per_wave(S) - 1) / rd_blocks_per_wave(S)), b, 0, ctx->stream, p); break;
    switch (p.d.tx_size) {
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18)
    default: return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "tx_size %d", p.d.tx_size);
    }
#undef CASE
    SVT_HIP_CHECK(ctx, hipGetLastError());
    return SVT_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Forward transform alone (av1_tranform_two_d_core_c, transforms.c:2259-2324; svt_av1_fwd_txfm2d_{W}x{H}{,_N2,_N4}_c): the
// head of rd_tx_kernel on a caller-supplied int16 residual, writing the FULL W x H coefficient array the per-size pointers
// return (the 64-point sizes included: packing to 32 x 32 is svt_handle_transform*'s job in the reference, :2374-2505).
// ---------------------------------------------------------------------------------------------------------
struct FwdParams {
    SvtHipFwdTxBatchDesc d;
};
template <int TS> __global__ void __launch_bounds__(64, rd_waves_per_simd(TS)) fwd_tx_kernel(const FwdParams p) {
    constexpr int W = tx_wide(TS), H = tx_high(TS);
    constexpr int LW = rd_lanes_per_block(TS), BPW = rd_blocks_per_wave(TS);
    constexpr int PA = W + 1;
    constexpr bool RECT = (W == 2 * H || H == 2 * W);
    __shared__ int32_t lds[BPW][H * PA];
    const int      lane = threadIdx.x, blk = lane / LW, l = lane % LW;
    const uint32_t job   = blockIdx.x * BPW + blk;
    const bool     valid = job < p.d.n_jobs;
    int32_t *A = lds[blk];
    const SvtHipTxJob jb = p.d.jobs[valid ? job : 0];
    const int tt = jb.tx_type & 15, vt = c_vtx[tt], ht = c_htx[tt];
    const bool ud = (vt == 2), lr = (ht == 2);
    const int16_t *res = p.d.residual + jb.src_offset;
    const int8_t *fsh = c_fwd_shift[TS];
    const int bit_col = c_fwd_cos_col[ilog2c(W) - 2][ilog2c(H) - 2], bit_row = c_fwd_cos_row[ilog2c(W) - 2][ilog2c(H) - 2];
    uint32_t rmax = 0;
    for (int i = l; i < W * H; i += LW) {
        const int r = i / W, c = i - r * W;
        const int32_t d = res[(size_t)r * p.d.residual_stride + c];
        A[r * PA + c] = d;
        rmax = max(rmax, (uint32_t)(d < 0 ? -d : d));
    }
    // one answer for the wave (all its blocks): small enough data for the three-instruction butterflies?  The column pass sees the residual
    // shifted up by fsh[0] (0 or 2)
    const bool fast_col = pass_fits_17_bits((uint32_t)__builtin_amdgcn_readfirstlane((int)seg_max_u32<64>(rmax)) << (fsh[0] > 0 ? fsh[0] : 0), H); // see kFwdMul24MaxResidual
    __syncthreads();
    uint32_t cmax = 0;
    if (l < W) { // columns (:2287-2308)
        int32_t x[H];
#pragma unroll
        for (int r = 0; r < H; r++) x[r] = A[(ud ? H - 1 - r : r) * PA + l];
        shift_vec<H>(x, fsh[0]);
        if (fast_col) fwd_1d<H, 2>(x, vt, bit_col); else fwd_1d<H, 0>(x, vt, bit_col); // wave-uniform
        shift_vec<H>(x, fsh[1]);
        const int oc = lr ? W - 1 - l : l;
#pragma unroll
        for (int r = 0; r < H; r++) { A[r * PA + oc] = x[r]; cmax = max(cmax, (uint32_t)(x[r] < 0 ? -x[r] : x[r])); }
    }
    const bool fast_row = pass_fits_17_bits((uint32_t)__builtin_amdgcn_readfirstlane((int)seg_max_u32<64>(cmax)), W); // the row pass's input
    __syncthreads();
    if (l < H) { // rows (:2310-2323)
        int32_t x[W];
#pragma unroll
        for (int c = 0; c < W; c++) x[c] = A[l * PA + c];
        if (fast_row) fwd_1d<W, 2>(x, ht, bit_row); else fwd_1d<W, 0>(x, ht, bit_row);
        shift_vec<W>(x, fsh[2]);
        if constexpr (RECT) {
#pragma unroll
            for (int c = 0; c < W; c++) x[c] = rshift64((i64)x[c] * 5793, 12);
        }
#pragma unroll
        for (int c = 0; c < W; c++) A[l * PA + c] = x[c];
    }
    __syncthreads();
    if (valid) { // the partial-frequency entries (_N2 / _N4, transforms.c:5202-5425,6769-6990) keep the top-left half / quarter and store zeros elsewhere
        const int pf = jb.pf_shape & 3;
        const int keep_w = pf == 3 ? 1 : (W >> pf), keep_h = pf == 3 ? 1 : (H >> pf);
        int32_t *out = p.d.coeff + (size_t)job * W * H;
        for (int i = l; i < W * H; i += LW) { const int r = i / W, c = i - r * W; out[i] = (c < keep_w && r < keep_h) ? A[r * PA + c] : 0; }
    }
}

int launch_fwd(SvtHipContext *ctx, const FwdParams &p) {
    const dim3 b(64);
#define CASE(S) case S: hipLaunchKernelGGL((fwd_tx_kernel<S>), dim3((p.d.n_jobs + rd_blocks_per_wave(S) - 1) / rd_blocks_per_wave(S)), b, 0, ctx->stream, p); break;
    switch (p.d.tx_size) {
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18)
    default: return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "tx_size %d", p.d.tx_size);
    }
#undef CASE
    SVT_HIP_CHECK(ctx, hipGetLastError());
    return SVT_HIP_OK;
}

template <int BD> int launch_size(SvtHipContext *ctx, const RdParams &p) {
    const dim3 b(64);
#define CASE(S) case S: hipLaunchKernelGGL((rd_tx_kernel<S, BD>), dim3((p.d.n_jobs + rd_blocks_per_wave(S) - 1) / rd_blocks_per_wave(S)), b, 0, ctx->stream, p); break;
    switch (p.tx_size) {
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18)
    default: return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "tx_size %d", p.tx_size);
    }
#undef CASE
    SVT_HIP_CHECK(ctx, hipGetLastError());
    return SVT_HIP_OK;
}

// host-side scan order generator (av1_scan_orders, Codec/coefficients.h:2197): diagonal scan for 2-D types and IDTX
// (zig-zag on square blocks, single-direction diagonals on rectangular ones), row scan for V_*, column scan for H_*
int scan_order(int tx_size, int tx_type, int16_t *scan, int16_t *iscan) {
    const int w = tx_wide(tx_size) > 32 ? 32 : tx_wide(tx_size), h = tx_high(tx_size) > 32 ? 32 : tx_high(tx_size), n = w * h;
    int k = 0;
    if (tx_type >= 10 && (tx_type & 1) == 0) { for (int i = 0; i < n; i++) scan[k++] = (int16_t)i; }
    else if (tx_type >= 11) { for (int c = 0; c < w; c++) for (int r = 0; r < h; r++) scan[k++] = (int16_t)(r * w + c); }
    else
        for (int d = 0; d < w + h - 1; d++) {
            const int down = (w < h) ? 1 : (w > h) ? 0 : (d & 1);
            for (int i = 0; i <= d; i++) { const int r = down ? i : d - i, c = d - r; if (r < h && c < w) scan[k++] = (int16_t)(r * w + c); }
        }
    for (int i = 0; i < n; i++) iscan[scan[i]] = (int16_t)i;
    return n;
}

} // namespace

extern "C" {

int svt_hip_tx_size_wide(int tx_size) { return (tx_size >= 0 && tx_size < 19) ? tx_wide(tx_size) : 0; }
int svt_hip_tx_size_high(int tx_size) { return (tx_size >= 0 && tx_size < 19) ? tx_high(tx_size) : 0; }

int svt_hip_scan_order(int tx_size, int tx_type, int16_t *scan, int16_t *iscan) {
    if (tx_size < 0 || tx_size >= 19 || tx_type < 0 || tx_type >= 16 || !scan || !iscan) return 0;
    return scan_order(tx_size, tx_type, scan, iscan);
}

int svt_hip_rd_batch(SvtHipContext *ctx, const SvtHipRdBatchDesc *d) {
    if (!ctx || !d) return SVT_HIP_ERR_BAD_PARAM;
    if (d->n_jobs == 0) return SVT_HIP_OK;
    if ((d->bit_depth != 8 && d->bit_depth != 10) || d->quant_kind > 2)
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "bit_depth %u / quant_kind %u", d->bit_depth, d->quant_kind);
    if (!d->src || !d->pred || !d->jobs || !d->quant_rows || !d->eob || !d->satd || !d->dist_coeff || !d->three_quad_energy || !d->sse)
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "a mandatory pointer of the RD batch is null");
    if ((d->qmatrix == nullptr) != (d->iqmatrix == nullptr)) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "qmatrix and iqmatrix go together");
    hipSetDevice(ctx->device);
    RdParams p;
    p.d       = *d;
    p.tx_size = d->tx_size;
    if (p.tx_size < 0 || p.tx_size >= 19) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "tx_size %d", p.tx_size);
    for (int k = 0; k < 3; k++) p.iscan[k] = ctx->iscan_dev + ((size_t)p.tx_size * 3 + k) * 1024;
    return d->bit_depth == 8 ? launch_size<8>(ctx, p) : launch_size<10>(ctx, p);
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------
// Full-pel motion-compensated prediction from the ME results: for every 16x16 PU of every b64, copy the 16x16
// block of the reference plane displaced by that PU's best full-pel MV (MeContext.p_sb_best_mv layout as written
// by svt_hip_me_picture: [b64][list][ref][85], 16x16 PUs at n_idx 5..20 in quad-tree order).  Coordinates are
// clamped into the picture (edge replication).  This is the integer-MV case of inter prediction -- the only part
// of prediction this path needs to feed the RD kernels; sub-pel interpolation is out of scope.
// ---------------------------------------------------------------------------------------------------------
namespace {
// one workgroup per 64x64 block; a thread copies runs of 8 samples (one 8- or 16-byte store; the source run starts at an
// arbitrary sample, so its load is unaligned) and falls back to per-sample clamping where the run touches a picture edge
template <typename Pix> __device__ __forceinline__ void fullpel_pred_b64(const Pix *ref, uint32_t ref_stride, int width, int height, const uint32_t *sb_best_mv,
                                                                         int list, int ref_idx, uint32_t w64, int bx, int by, Pix *pred, uint32_t pred_stride,
                                                                         int vec_ok) {
    typedef Pix __attribute__((ext_vector_type(8), aligned(sizeof(Pix)))) RunU; // unaligned run of 8 samples
    typedef Pix __attribute__((ext_vector_type(8))) Run;
    const uint32_t b = (uint32_t)bx + (uint32_t)by * w64;
    const uint32_t *mvs = sb_best_mv + ((size_t)b * 8 + list * 4 + ref_idx) * 85 + 5; // the 16 16x16 PUs, quad-tree order
    for (int seg = threadIdx.x; seg < 64 * 8; seg += 256) {
        const int ly = seg >> 3, lx = (seg & 7) * 8, x = bx * 64 + lx, y = by * 64 + ly;
        if (x >= width || y >= height) continue;
        const int qx = lx >> 4, qy = ly >> 4;
        const uint32_t mv = mvs[(qx & 1) | ((qy & 1) << 1) | ((qx >> 1) << 2) | ((qy >> 1) << 3)];
        const int mvx = (int16_t)(mv & 0xFFFF), mvy = (int16_t)(mv >> 16);
        int sy = y + mvy;
        sy = sy < 0 ? 0 : (sy > height - 1 ? height - 1 : sy);
        const int sx = x + mvx;
        const Pix *srow = ref + (size_t)sy * ref_stride;
        Pix       *drow = pred + (size_t)y * pred_stride + x;
        if (vec_ok && sx >= 0 && sx + 8 <= width && x + 8 <= width) {
            *reinterpret_cast<Run *>(drow) = *reinterpret_cast<const RunU *>(srow + sx);
        } else {
            for (int i = 0; i < 8 && x + i < width; i++) {
                int cx = sx + i;
                cx = cx < 0 ? 0 : (cx > width - 1 ? width - 1 : cx);
                drow[i] = srow[cx];
            }
        }
    }
}

template <typename Pix> __global__ void __launch_bounds__(256) fullpel_pred_kernel(const Pix *ref, uint32_t ref_stride, int width, int height,
                                                                                   const uint32_t *sb_best_mv, int list, int ref_idx, uint32_t w64, int by64_first,
                                                                                   Pix *pred, uint32_t pred_stride, int vec_ok) {
    fullpel_pred_b64<Pix>(ref, ref_stride, width, height, sb_best_mv, list, ref_idx, w64, (int)blockIdx.x, by64_first + (int)blockIdx.y, pred, pred_stride, vec_ok);
}

// several pictures in one launch (blockIdx.z = picture): the per-picture launches of a row band are launch-bound
struct PredBatch {
    SvtHipPredJob job[SVT_HIP_PRED_MAX_JOBS];
};
template <typename Pix> __global__ void __launch_bounds__(256) fullpel_pred_batch_kernel(const PredBatch pb, uint32_t ref_stride, int width, int height, uint32_t w64,
                                                                                         uint32_t pred_stride, int vec_ok) {
    const SvtHipPredJob &j = pb.job[blockIdx.z];
    if (blockIdx.y >= j.b64_row_count) return;
    fullpel_pred_b64<Pix>(static_cast<const Pix *>(j.ref), ref_stride, width, height, j.sb_best_mv, j.list, j.ref_idx, w64, (int)blockIdx.x,
                          (int)(j.b64_row_start + blockIdx.y), static_cast<Pix *>(j.pred), pred_stride, vec_ok);
}
} // namespace

extern "C" int svt_hip_fullpel_pred(SvtHipContext *ctx, const void *ref, uint32_t ref_stride, uint32_t width, uint32_t height, uint8_t bit_depth,
                                    const uint32_t *sb_best_mv, uint8_t list, uint8_t ref_idx, uint32_t b64_row_start, uint32_t b64_row_count, void *pred,
                                    uint32_t pred_stride) {
    if (!ctx || !ref || !sb_best_mv || !pred || list > 1 || ref_idx > 3 || (bit_depth != 8 && bit_depth != 10) || !width || !height)
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "svt_hip_fullpel_pred: bad argument");
    hipSetDevice(ctx->device);
    const uint32_t w64 = (width + 63) / 64, h64 = (height + 63) / 64;
    if (b64_row_start >= h64) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "svt_hip_fullpel_pred: b64_row_start %u >= %u", b64_row_start, h64);
    if (b64_row_count == 0 || b64_row_start + b64_row_count > h64) b64_row_count = h64 - b64_row_start;
    const dim3 g(w64, b64_row_count), blk(256);
    const size_t bpp = bit_depth == 8 ? 1 : 2;
    const int vec_ok = ((reinterpret_cast<uintptr_t>(pred) % (8 * bpp)) == 0 && pred_stride % 8 == 0) ? 1 : 0; // aligned 8-sample stores
    if (bit_depth == 8)
        hipLaunchKernelGGL(fullpel_pred_kernel<uint8_t>, g, blk, 0, ctx->stream, static_cast<const uint8_t *>(ref), ref_stride, (int)width, (int)height,
                           sb_best_mv, (int)list, (int)ref_idx, w64, (int)b64_row_start, static_cast<uint8_t *>(pred), pred_stride, vec_ok);
    else
        hipLaunchKernelGGL(fullpel_pred_kernel<uint16_t>, g, blk, 0, ctx->stream, static_cast<const uint16_t *>(ref), ref_stride, (int)width, (int)height,
                           sb_best_mv, (int)list, (int)ref_idx, w64, (int)b64_row_start, static_cast<uint16_t *>(pred), pred_stride, vec_ok);
    SVT_HIP_CHECK(ctx, hipGetLastError());
    return SVT_HIP_OK;
}

extern "C" int svt_hip_fullpel_pred_batch(SvtHipContext *ctx, uint32_t ref_stride, uint32_t width, uint32_t height, uint8_t bit_depth, uint32_t pred_stride,
                                          uint32_t n_jobs, const SvtHipPredJob *jobs) {
    if (!ctx) return SVT_HIP_ERR_BAD_PARAM;
    if (!jobs || n_jobs == 0 || n_jobs > SVT_HIP_PRED_MAX_JOBS || (bit_depth != 8 && bit_depth != 10) || !width || !height)
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "svt_hip_fullpel_pred_batch: bad argument (1..%d jobs)", SVT_HIP_PRED_MAX_JOBS);
    const uint32_t w64 = (width + 63) / 64, h64 = (height + 63) / 64;
    const size_t   bpp = bit_depth == 8 ? 1 : 2;
    PredBatch pb;
    memset(&pb, 0, sizeof(pb));
    uint32_t rows_max = 0;
    int      vec_ok   = pred_stride % 8 == 0;
    for (uint32_t i = 0; i < n_jobs; i++) {
        SvtHipPredJob j = jobs[i];
        if (!j.ref || !j.sb_best_mv || !j.pred || j.list > 1 || j.ref_idx > 3 || j.b64_row_start >= h64)
            return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "svt_hip_fullpel_pred_batch: job %u: null pointer, list / ref_idx or b64_row_start %u out of range", i,
                                j.b64_row_start);
        if (j.b64_row_count == 0 || j.b64_row_start + j.b64_row_count > h64) j.b64_row_count = h64 - j.b64_row_start;
        rows_max = j.b64_row_count > rows_max ? j.b64_row_count : rows_max;
        vec_ok   = vec_ok && (reinterpret_cast<uintptr_t>(j.pred) % (8 * bpp)) == 0;
        pb.job[i] = j;
    }
    hipSetDevice(ctx->device);
    const dim3 g(w64, rows_max, n_jobs), blk(256);
    if (bit_depth == 8) hipLaunchKernelGGL(fullpel_pred_batch_kernel<uint8_t>, g, blk, 0, ctx->stream, pb, ref_stride, (int)width, (int)height, w64, pred_stride, vec_ok);
    else hipLaunchKernelGGL(fullpel_pred_batch_kernel<uint16_t>, g, blk, 0, ctx->stream, pb, ref_stride, (int)width, (int)height, w64, pred_stride, vec_ok);
    SVT_HIP_CHECK(ctx, hipGetLastError());
    return SVT_HIP_OK;
}

extern "C" int svt_hip_inv_txfm_batch(SvtHipContext *ctx, const SvtHipInvTxBatchDesc *d) {
    if (!ctx || !d) return SVT_HIP_ERR_BAD_PARAM;
    if (d->n_jobs == 0) return SVT_HIP_OK;
    if ((d->bit_depth != 8 && d->bit_depth != 10) || (d->sample_bytes != 1 && d->sample_bytes != 2) || (d->bit_depth == 10 && d->sample_bytes != 2))
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "inverse batch: bit_depth %u with %u-byte samples", d->bit_depth, d->sample_bytes);
    if (d->tx_size >= SVT_HIP_TX_SIZES_ALL) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "tx_size %u", d->tx_size);
    if (!d->pred || !d->recon || !d->jobs || !d->dqcoeff) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "a pointer of the inverse batch is null");
    hipSetDevice(ctx->device);
    InvParams p;
    p.d = *d;
    if (d->bit_depth == 10) return launch_inv<10, uint16_t>(ctx, p);
    return d->sample_bytes == 1 ? launch_inv<8, uint8_t>(ctx, p) : launch_inv<8, uint16_t>(ctx, p);
}

extern "C" int svt_hip_fwd_txfm_batch(SvtHipContext *ctx, const SvtHipFwdTxBatchDesc *d) {
    if (!ctx || !d) return SVT_HIP_ERR_BAD_PARAM;
    if (d->n_jobs == 0) return SVT_HIP_OK;
    if (d->tx_size >= SVT_HIP_TX_SIZES_ALL) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "tx_size %u", d->tx_size);
    if (!d->residual || !d->jobs || !d->coeff) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "a pointer of the forward batch is null");
    hipSetDevice(ctx->device);
    FwdParams p;
    p.d = *d;
    return launch_fwd(ctx, p);
}

// The cosine table (constant memory of this device's copy of the code object) and the inverse scan orders, made once per
// context at svt_hip_context_create: no lazily initialised process-global state (several threads, several GPUs per process).
int svt_hip_rd_tables_init(SvtHipContext *ctx) {
    int32_t cosp[4][64];
    for (int b = 0; b < 4; b++)
        for (int j = 0; j < 64; j++) cosp[b][j] = (int32_t)(cos(3.14159265358979323846 * j / 128.0) * (double)(1 << (10 + b)) + 0.5);
    SVT_HIP_CHECK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_cospi), cosp, sizeof(cosp)));
    const size_t bytes = sizeof(int16_t) * 19 * 3 * 1024;
    int16_t *tab = static_cast<int16_t *>(calloc(1, bytes)), scan[1024];
    if (!tab) return svt_hip_fail(ctx, SVT_HIP_ERR_NO_MEMORY, "iscan tables");
    for (int s = 0; s < 19; s++)
        for (int kind = 0; kind < 3; kind++) scan_order(s, kind == 0 ? 0 : (kind == 1 ? 10 : 11), scan, tab + ((size_t)s * 3 + kind) * 1024);
    int rc = SVT_HIP_OK;
    if (hipMalloc(reinterpret_cast<void **>(&ctx->iscan_dev), bytes) != hipSuccess) rc = svt_hip_fail(ctx, SVT_HIP_ERR_NO_MEMORY, "iscan tables");
    else if (hipMemcpy(ctx->iscan_dev, tab, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = svt_hip_fail(ctx, SVT_HIP_ERR_LAUNCH, "iscan tables: copy failed");
    free(tab);
    return rc;
}

void svt_hip_rd_tables_free(SvtHipContext *ctx) {
    if (ctx->iscan_dev) hipFree(ctx->iscan_dev);
    ctx->iscan_dev = nullptr;
}
