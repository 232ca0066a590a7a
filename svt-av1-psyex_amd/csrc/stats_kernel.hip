// stats_kernel.hip -- batched block statistics on gfx950: SAD, SSE, variance and Hadamard SATD of (source - reference)
// blocks, plus the stand-alone exhaustive SAD search (svt_sad_loop_kernel) and Hadamard transforms behind the
// pointer-level `_hip` entries of include/svt_hip_leaf.h.
//
// Reference functions restated (Source/Lib):
//   svt_nxm_sad_kernel_helper_c / svt_aom_sad_16b_kernel_c          C_DEFAULT/compute_sad_c.c:20-56,209
//   svt_sad_loop_kernel_c                                           C_DEFAULT/compute_sad_c.c:58-101
//   svt_spatial_full_distortion_kernel_c / svt_full_distortion_kernel16_bits_c / svt_aom_sse_c / svt_aom_highbd_sse_c
//                                                                   C_DEFAULT/picture_operators_c.c:65-83, Codec/pic_operators.c:174-197,
//                                                                   Codec/enc_inter_prediction.c:559-583
//   svt_aom_variance{W}x{H}_c / svt_aom_variance_highbd_c           C_DEFAULT/variance.c:256-296
//   svt_aom_hadamard_{4x4,8x8,16x16,32x32}_c, svt_aom_satd_c        C_DEFAULT/picture_operators_c.c:176-326, Codec/common_dsp_rtcd.c:70-77
//   hadamard_path_c                                                 Codec/enc_mode_config.c:2151-2217
//
// One wave64 works through kJobsPerWave consecutive jobs (a wave per 8x8 block is launch-bound: 43k single-job waves of a 1080p picture
// took as long as 11k four-job waves).  Pixel statistics: a lane takes 4 neighbouring samples at a time (one 4- or 8-byte load per plane;
// 8-bit: v_sad_u8 and three v_dot4_u32_u8 give SAD, sum and sum of squares; 10-bit: packed 16-bit differences, v_sad_u16, v_dot2_i32_i16),
// lane sums reduced across the wave by DPP.  hadamard_path: <= 32x32 tiles staged in LDS; 16x16 / 32x32 tiles run their 8x8 cores on the matrix
// cores (exact in f16 operands) and the reference's truncating 16 / 32 combines on the VALU, 4x4 / 8x8 tiles are VALU butterflies with the
// reference's 16-bit intermediates.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <stdarg.h>
#include "svt_hip_internal.h"
#include "leaf_guard.h"
#include "../../include/svt_hip_spy_rd.h"
#include "../../include/svt_hip_dsp.h"
#include "../../include/svt_hip_leaf.h"

namespace {

typedef unsigned long long u64;
typedef long long          i64;

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// 32-bit sum over the wave by DPP (no LDS crossbar traffic): the total lands in lane 63, returned to every lane as a scalar
__device__ __forceinline__ uint32_t wave_sum_dpp(uint32_t v) {
#define SVT_SUM_DPP(ctrl, rmask) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xF, true);
    SVT_SUM_DPP(0x111, 0xF) // row_shr:1
    SVT_SUM_DPP(0x112, 0xF) // row_shr:2
    SVT_SUM_DPP(0x114, 0xF) // row_shr:4
    SVT_SUM_DPP(0x118, 0xF) // row_shr:8: lane 15 of every row holds the row's sum
#undef SVT_SUM_DPP
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// ---- Hadamard building blocks (same operation order and intermediate widths as the reference) -----------------
__device__ __forceinline__ void had_col4(const int16_t *s, int st, int16_t *o) {
    const int16_t b0 = (int16_t)((s[0] + s[st]) >> 1), b1 = (int16_t)((s[0] - s[st]) >> 1);
    const int16_t b2 = (int16_t)((s[2 * st] + s[3 * st]) >> 1), b3 = (int16_t)((s[2 * st] - s[3 * st]) >> 1);
    o[0] = (int16_t)(b0 + b2);
    o[1] = (int16_t)(b1 + b3);
    o[2] = (int16_t)(b0 - b2);
    o[3] = (int16_t)(b1 - b3);
}
__device__ __forceinline__ void had_col8(const int16_t *s, int st, int16_t *o) {
    int16_t b[8], c[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        b[2 * i]     = (int16_t)(s[2 * i * st] + s[(2 * i + 1) * st]);
        b[2 * i + 1] = (int16_t)(s[2 * i * st] - s[(2 * i + 1) * st]);
    }
#pragma unroll
    for (int g = 0; g < 2; g++) {
        c[4 * g + 0] = (int16_t)(b[4 * g + 0] + b[4 * g + 2]);
        c[4 * g + 1] = (int16_t)(b[4 * g + 1] + b[4 * g + 3]);
        c[4 * g + 2] = (int16_t)(b[4 * g + 0] - b[4 * g + 2]);
        c[4 * g + 3] = (int16_t)(b[4 * g + 1] - b[4 * g + 3]);
    }
    // output slots of c[i] + c[i+4] and c[i] - c[i+4]
    o[0] = (int16_t)(c[0] + c[4]); o[2] = (int16_t)(c[0] - c[4]);
    o[7] = (int16_t)(c[1] + c[5]); o[6] = (int16_t)(c[1] - c[5]);
    o[3] = (int16_t)(c[2] + c[6]); o[1] = (int16_t)(c[2] - c[6]);
    o[4] = (int16_t)(c[3] + c[7]); o[5] = (int16_t)(c[3] - c[7]);
}

// Every wave of these kernels works on LDS tiles of its own: a wave's LDS instructions execute in order, so what orders a lane's reads behind
// another lane's writes is a compiler-level barrier only (a workgroup barrier would also tie together waves that share a workgroup but not a job).
__device__ __forceinline__ void stats_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kResPitch = 34; // int16 row pitch of the residual tile in LDS (17 dwords: rows land on different banks)

struct HadLds {
    int16_t res[32 * kResPitch];
    int16_t t[1024];
    int32_t c[1024];
};

// Hadamard of the n x n residual tile in L.res (n = 4, 8, 16, 32); coefficients in L.c in the reference's order.
// One wave; every step ends with a barrier.
__device__ void hadamard_tile(HadLds &L, int n, int lane) {
    if (n == 4) {
        if (lane < 4) had_col4(L.res + lane, kResPitch, L.t + 4 * lane);
        stats_wave_sync();
        if (lane < 4) {
            int16_t o[4];
            had_col4(L.t + lane, 4, o);
            for (int k = 0; k < 4; k++) L.c[4 * lane + k] = o[k];
        }
        stats_wave_sync();
        return;
    }
    const int nb = n >> 3; // 8x8 sub-blocks per side
    // coefficient base of the 8x8 sub-block at (by, bx): 16x16 blocks in raster order, 8x8 blocks in raster order inside
    auto base_of = [&](int by, int bx) {
        if (n == 8) return 0;
        if (n == 16) return 64 * (2 * by + bx);
        return 256 * (2 * (by >> 1) + (bx >> 1)) + 64 * (2 * (by & 1) + (bx & 1));
    };
    for (int it = lane; it < nb * nb * 8; it += 64) { // first pass: columns of every 8x8 sub-block
        const int i = it & 7, sb = it >> 3, by = sb / nb, bx = sb - by * nb;
        had_col8(L.res + (8 * by) * kResPitch + 8 * bx + i, kResPitch, L.t + base_of(by, bx) + 8 * i);
    }
    stats_wave_sync();
    for (int it = lane; it < nb * nb * 8; it += 64) { // second pass: rows of the intermediate
        const int i = it & 7, sb = it >> 3, by = sb / nb, bx = sb - by * nb;
        const int base = base_of(by, bx);
        int16_t   o[8];
        had_col8(L.t + base + i, 8, o);
        for (int k = 0; k < 8; k++) L.c[base + 8 * i + k] = o[k];
    }
    stats_wave_sync();
    auto combine = [&](int32_t *c, int cn, int shift, int i) {
        const int32_t a0 = c[i], a1 = c[cn + i], a2 = c[2 * cn + i], a3 = c[3 * cn + i];
        const int32_t b0 = (a0 + a1) >> shift, b1 = (a0 - a1) >> shift, b2 = (a2 + a3) >> shift, b3 = (a2 - a3) >> shift;
        c[i] = b0 + b2; c[cn + i] = b1 + b3; c[2 * cn + i] = b0 - b2; c[3 * cn + i] = b1 - b3;
    };
    if (n >= 16) {
        const int n16 = (n == 16) ? 1 : 4;
        for (int it = lane; it < n16 * 64; it += 64) combine(L.c + 256 * (it >> 6), 64, 1, it & 63);
        stats_wave_sync();
    }
    if (n == 32) {
        for (int it = lane; it < 256; it += 64) combine(L.c, 256, 2, it);
        stats_wave_sync();
    }
}

// ---- hadamard_path's 16x16 / 32x32 tiles on the matrix cores --------------------------------------------------------------
// The 8x8 cores of a 16x16 block are Y = H16 * (X * H16) with H16 = diag(H8, H8): two v_mfma_f32_16x16x16_f16.  Exact: 9-bit residuals and
// +-1 weights are f16 values, |X * H16| <= 8 * 255 = 2040 < 2048 is still an f16 integer, the f32 sums stay below 2^15; and the first
// product's C layout (lane holds rows k0 .. k0 + 3 of column r) is the second product's B layout.  The reference's 8-point transform is
// the same set of Walsh rows in another order without sign changes (every row starts with +1), so combining the four 8x8 blocks of a 16x16
// element by element (picture_operators_c.c:270-297: (a0 +- a1) >> 1 ...) and the four 16x16 blocks of a 32x32 (:299-326, >> 2) gives the
// reference's coefficients in a permuted order -- and SATD is a sum.  (tools/ubench/hadamard_mfma.hip measures the 8x8 core alone.)
typedef _Float16 had_half4 __attribute__((ext_vector_type(4)));
typedef float    had_float4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ had_half4 had16_weights(int lane) { // H16[r][k0 + j] = H16[k0 + j][r]
    const int r = lane & 15, k0 = 4 * (lane >> 4);
    had_half4 h;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int k = k0 + j;
        h[j] = ((k >> 3) != (r >> 3)) ? (_Float16)0 : ((__builtin_popcount((k & 7) & (r & 7)) & 1) ? (_Float16)-1 : (_Float16)1);
    }
    return h;
}

// the four 8x8 Hadamards of the 16x16 block of L.res at (16 by, 16 bx): this lane's coefficients Y[k0 + j][r] of 8x8 block (k0 / 8, r / 8)
__device__ __forceinline__ void had8x4_mfma(const HadLds &L, int by, int bx, int lane, had_half4 h, int32_t out[4]) {
    const int      r = lane & 15, k0 = 4 * (lane >> 4);
    const int16_t *x = L.res + (16 * by + r) * kResPitch + 16 * bx + k0;
    had_half4 a;
#pragma unroll
    for (int j = 0; j < 4; j++) a[j] = (_Float16)x[j];
    had_float4 t = {0, 0, 0, 0};
    t = __builtin_amdgcn_mfma_f32_16x16x16f16(a, h, t, 0, 0, 0); // T = X * H16: lane holds T[k0 + j][r]
    had_half4 tb;
#pragma unroll
    for (int j = 0; j < 4; j++) tb[j] = (_Float16)t[j];
    had_float4 y = {0, 0, 0, 0};
    y = __builtin_amdgcn_mfma_f32_16x16x16f16(h, tb, y, 0, 0, 0); // Y = H16 * T: lane holds Y[k0 + j][r], 8x8 block (k0 / 8, r / 8)
#pragma unroll
    for (int j = 0; j < 4; j++) out[j] = (int)y[j];
}

// the 16x16 block of L.res at (16 by, 16 bx): this lane's four 16x16-Hadamard coefficients (svt_aom_hadamard_16x16_c arithmetic)
__device__ __forceinline__ void had16_mfma(const HadLds &L, int by, int bx, int lane, had_half4 h, int32_t out[4]) {
    int32_t y[4];
    had8x4_mfma(L, by, bx, lane, h, y);
    const bool right = (lane & 8) != 0, bottom = (lane & 32) != 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int v = y[j], p8 = __shfl_xor(v, 8, 64), p32 = __shfl_xor(v, 32, 64), p40 = __shfl_xor(v, 40, 64);
        // the element's four 8x8 blocks in raster order
        const int l0 = right ? p8 : v, r0 = right ? v : p8;       // this lane's row of blocks: left, right
        const int l1 = right ? p40 : p32, r1 = right ? p32 : p40; // the other row of blocks
        const int a0 = bottom ? l1 : l0, a1 = bottom ? r1 : r0, a2 = bottom ? l0 : l1, a3 = bottom ? r0 : r1;
        const int b0 = (a0 + a1) >> 1, b1 = (a0 - a1) >> 1, b2 = (a2 + a3) >> 1, b3 = (a2 - a3) >> 1;
        out[j] = right ? (bottom ? b1 - b3 : b1 + b3) : (bottom ? b0 - b2 : b0 + b2);
    }
}

// sum of the absolute Hadamard coefficients of the n x n residual tile in L.res, n = 16 or 32 (this lane's share)
__device__ __forceinline__ uint32_t hadamard_satd_mfma(const HadLds &L, int n, int lane) {
    const had_half4 h = had16_weights(lane);
    uint32_t s = 0;
    if (n == 16) {
        int32_t o[4];
        had16_mfma(L, 0, 0, lane, h, o);
#pragma unroll
        for (int j = 0; j < 4; j++) s += (uint32_t)(o[j] < 0 ? -o[j] : o[j]);
        return s;
    }
    int32_t o0[4], o1[4], o2[4], o3[4];
    had16_mfma(L, 0, 0, lane, h, o0); had16_mfma(L, 0, 1, lane, h, o1); had16_mfma(L, 1, 0, lane, h, o2); had16_mfma(L, 1, 1, lane, h, o3);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int b0 = (o0[j] + o1[j]) >> 2, b1 = (o0[j] - o1[j]) >> 2, b2 = (o2[j] + o3[j]) >> 2, b3 = (o2[j] - o3[j]) >> 2;
        const int c0 = b0 + b2, c1 = b1 + b3, c2 = b0 - b2, c3 = b1 - b3;
        s += (uint32_t)(c0 < 0 ? -c0 : c0) + (uint32_t)(c1 < 0 ? -c1 : c1) + (uint32_t)(c2 < 0 ? -c2 : c2) + (uint32_t)(c3 < 0 ? -c3 : c3);
    }
    return s;
}

// A block of a plane, optionally seen through the 2-tap bilinear interpolation of svt_aom_sub_pixel_variance{W}x{H}_c
// (C_DEFAULT/variance.c:28-75,308-318; taps {128 - 16k, 16k}, filter.h:39-48): horizontal pass into 16 bit, vertical pass
// back to the pixel range, each with a rounding shift by FILTER_BITS = 7.  Evaluated on the fly (4 cached reads per
// sample); a neighbour is only read when its tap is non-zero.
template <typename Pix> struct View {
    const Pix *p;
    uint32_t   stride;
    int        fx1, fy1; // second taps (0 = no interpolation in that direction)
    __device__ __forceinline__ int at(int y, int x) const {
        const Pix *q = p + (size_t)y * stride + x;
        if ((fx1 | fy1) == 0) return (int)q[0];
        const int fx0 = 128 - fx1, fy0 = 128 - fy1;
        const int m0 = ((int)q[0] * fx0 + (fx1 ? (int)q[1] * fx1 : 0) + 64) >> 7;
        if (!fy1) return (m0 * fy0 + 64) >> 7;
        const int m1 = ((int)q[stride] * fx0 + (fx1 ? (int)q[stride + 1] * fx1 : 0) + 64) >> 7;
        return (m0 * fy0 + m1 * fy1 + 64) >> 7;
    }
    __device__ __forceinline__ View sub(int y, int x) const { View v = *this; v.p = p + (size_t)y * stride + x; return v; }
};

// ---- PSYEX psy-RD energy (Codec/psy_rd.c:64-274) -----------------------------------------------------------------
// 8-bit: the reference's packed 2 x 16-bit Hadamard never overflows a half on pixel data, so it equals the plain
// unnormalised 2-D Hadamard.  10-bit: its 4-point butterflies keep 32-bit temporaries, so only the low half of the
// packed 2 x 32-bit values survives each butterfly; restated bit for bit (see oracle/stats_oracle.c for the derivation).
__device__ __forceinline__ void had8_inplace(int32_t *v) { // unnormalised 8-point Hadamard of v[0..7]
#pragma unroll
    for (int len = 1; len < 8; len <<= 1)
#pragma unroll
        for (int i = 0; i < 8; i += 2 * len)
#pragma unroll
            for (int j = i; j < i + len; j++) { const int32_t a = v[j], b = v[j + len]; v[j] = a + b; v[j + len] = a - b; }
}
__device__ __forceinline__ u64 pack32(int32_t x0, int32_t x1) { return (u64)(i64)(x0 + x1) + ((u64)(i64)(x0 - x1) << 32); }
__device__ __forceinline__ void bfly_low(u64 d[4], u64 s0, u64 s1, u64 s2, u64 s3) {
    const uint32_t t0 = (uint32_t)(s0 + s1), t1 = (uint32_t)(s0 - s1), t2 = (uint32_t)(s2 + s3), t3 = (uint32_t)(s2 - s3);
    d[0] = (uint32_t)(t0 + t2); d[1] = (uint32_t)(t1 + t3); d[2] = (uint32_t)(t0 - t2); d[3] = (uint32_t)(t1 - t3);
}
__device__ __forceinline__ u64 abs_halves(u64 a) { const u64 m = (a >> 31) & 0x100000001ull, s = (m << 32) - m; return (a + s) ^ s; }
__device__ __forceinline__ u64 fold_halves(u64 b) { return (uint32_t)b + (b >> 32); }

// energy of one n x n tile (n = 8 or 4) of plane p: Hadamard sum - (sum of pixels >> 2)
template <typename Pix> __device__ int32_t psy_tile_energy(const View<Pix> &pv, int n) {
    i64 sum = 0, had;
    if (sizeof(Pix) == 1) {
        int32_t m[8][8];
        if (n == 8) {
#pragma unroll
            for (int y = 0; y < 8; y++) {
#pragma unroll
                for (int x = 0; x < 8; x++) { m[y][x] = (int32_t)pv.at(y, x); sum += m[y][x]; }
                had8_inplace(m[y]);
            }
            i64 acc = 0;
#pragma unroll
            for (int x = 0; x < 8; x++) {
                int32_t c[8];
#pragma unroll
                for (int y = 0; y < 8; y++) c[y] = m[y][x];
                had8_inplace(c);
#pragma unroll
                for (int y = 0; y < 8; y++) acc += c[y] < 0 ? -c[y] : c[y];
            }
            had = (acc + 2) >> 2;
        } else {
            int32_t q[4][4];
            i64     acc = 0;
#pragma unroll
            for (int y = 0; y < 4; y++) {
#pragma unroll
                for (int x = 0; x < 4; x++) { q[y][x] = (int32_t)pv.at(y, x); sum += q[y][x]; }
                const int32_t a = q[y][0] + q[y][1], b = q[y][0] - q[y][1], c = q[y][2] + q[y][3], d = q[y][2] - q[y][3];
                q[y][0] = a + c; q[y][1] = b + d; q[y][2] = a - c; q[y][3] = b - d;
            }
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const int32_t a = q[0][x] + q[1][x], b = q[0][x] - q[1][x], c = q[2][x] + q[3][x], d = q[2][x] - q[3][x];
                const int32_t o0 = a + c, o1 = b + d, o2 = a - c, o3 = b - d;
                acc += (o0 < 0 ? -o0 : o0) + (o1 < 0 ? -o1 : o1) + (o2 < 0 ? -o2 : o2) + (o3 < 0 ? -o3 : o3);
            }
            had = acc >> 1;
        }
    } else {
        u64 t[8][4], hs = 0;
        if (n == 8) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                int32_t r[8];
#pragma unroll
                for (int x = 0; x < 8; x++) { r[x] = (int32_t)pv.at(i, x); sum += r[x]; }
                bfly_low(t[i], pack32(r[0], r[1]), pack32(r[2], r[3]), pack32(r[4], r[5]), pack32(r[6], r[7]));
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                u64 a[8], b = 0;
                bfly_low(a, t[0][i], t[1][i], t[2][i], t[3][i]);
                bfly_low(a + 4, t[4][i], t[5][i], t[6][i], t[7][i]);
#pragma unroll
                for (int k = 0; k < 4; k++) b += abs_halves(a[k] + a[k + 4]) + abs_halves(a[k] - a[k + 4]);
                hs += fold_halves(b);
            }
            had = (i64)((hs + 2) >> 2);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int32_t r[4];
#pragma unroll
                for (int x = 0; x < 4; x++) { r[x] = (int32_t)pv.at(i, x); sum += r[x]; }
                const u64 b0 = pack32(r[0], r[1]), b1 = pack32(r[2], r[3]);
                t[i][0] = b0 + b1; t[i][1] = b0 - b1;
            }
#pragma unroll
            for (int i = 0; i < 2; i++) {
                u64 a[4];
                bfly_low(a, t[0][i], t[1][i], t[2][i], t[3][i]);
                hs += fold_halves(abs_halves(a[0]) + abs_halves(a[1]) + abs_halves(a[2]) + abs_halves(a[3]));
            }
            had = (i64)(hs >> 1);
        }
    }
    return (int32_t)(had - (sum >> 2));
}

struct StatsParams {
    SvtHipBlockStatsDesc d;
    uint32_t n_front; // workgroups [0, n_front) of the launch take the regions (d.pyramids), the rest the flat jobs
    uint32_t jpw;     // flat jobs per wave: kJobsPerWave, or 1 when the batch is too small to fill the chip that way (a wave's jobs run one after the other)
};

constexpr int kJobsPerWave = 4; // a multiple of 4, at most 16 (the psy prefix sum runs inside one DPP row).  Measured: 16 jobs per wave bring the 2160p psy batch from 0.29 to 0.21 ms but the 1080p statistics batch from 0.065 to 0.22 ms (sixteen 64x64 blocks in a row make a long, lonely wave)
typedef uint32_t U32U __attribute__((aligned(1)));
typedef uint32_t U64U __attribute__((ext_vector_type(2), aligned(2)));
typedef short    short2v __attribute__((ext_vector_type(2)));

// SAD, sum and sum of squares of the differences of 4 neighbouring samples, added to the lane's running sums
__device__ __forceinline__ void quad_stats(const uint8_t *s, const uint8_t *r, uint32_t &sad, int32_t &sum, uint32_t &sq) {
    const uint32_t a = *reinterpret_cast<const U32U *>(s), b = *reinterpret_cast<const U32U *>(r);
    sad = __builtin_amdgcn_sad_u8(a, b, sad);
    sum += (int32_t)__builtin_amdgcn_sad_u8(a, 0u, 0u) - (int32_t)__builtin_amdgcn_sad_u8(b, 0u, 0u);
    // sum (a - b)^2 = a.a + b.b - 2 a.b on the packed-byte dot product
    sq += __builtin_amdgcn_udot4(a, a, __builtin_amdgcn_udot4(b, b, 0u, false), false) - 2u * __builtin_amdgcn_udot4(a, b, 0u, false);
}
__device__ __forceinline__ void quad_stats(const uint16_t *s, const uint16_t *r, uint32_t &sad, int32_t &sum, uint32_t &sq) {
    const U64U a = *reinterpret_cast<const U64U *>(s), b = *reinterpret_cast<const U64U *>(r);
    const short2v one = {1, 1};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const uint32_t x = k ? a.y : a.x, y = k ? b.y : b.x;
        const short2v  d = __builtin_bit_cast(short2v, x) - __builtin_bit_cast(short2v, y); // 10-bit samples: no wrap
        sad = __builtin_amdgcn_sad_u16(x, y, sad);
        sum = __builtin_amdgcn_sdot2(d, one, sum, false);
        sq  = (uint32_t)__builtin_amdgcn_sdot2(d, d, (int)sq, false);
    }
}

// residuals of 4 neighbouring samples -> 4 x int16 in the LDS tile
__device__ __forceinline__ void quad_residual(const uint8_t *s, const uint8_t *r, int16_t *dst) {
    const uint32_t a = *reinterpret_cast<const U32U *>(s), b = *reinterpret_cast<const U32U *>(r);
#pragma unroll
    for (int k = 0; k < 4; k++) dst[k] = (int16_t)((int)((a >> (8 * k)) & 0xFF) - (int)((b >> (8 * k)) & 0xFF));
}
__device__ __forceinline__ void quad_residual(const uint16_t *s, const uint16_t *r, int16_t *dst) {
#pragma unroll
    for (int k = 0; k < 4; k++) dst[k] = (int16_t)((int16_t)s[k] - (int16_t)r[k]);
}

// a non-negative 64-bit value by a block's sample count (a power of two for every AV1 block shape)
__device__ __forceinline__ i64 div_by_area(i64 v, int area) { return (area & (area - 1)) ? v / area : v >> (31 - __builtin_clz(area)); }

// the per-block outputs that follow from the SAD, the sum and the sum of squares of the differences (one lane)
__device__ __forceinline__ void write_pixel_outputs(const StatsParams &p, uint32_t job, int w, int h, uint32_t sad, int32_t sum, u64 sse) {
    const uint32_t sq32 = (uint32_t)sse; // the reference's 32-bit accumulator (variance.c) wraps the same way
    if (p.d.sad) p.d.sad[job] = sad;
    if (p.d.sse) p.d.sse[job] = sse;
    if (p.d.var_sse) p.d.var_sse[job] = sq32;
    if (p.d.variance) p.d.variance[job] = sq32 - (uint32_t)div_by_area((i64)sum * sum, w * h);
    if (p.d.variance10 || p.d.var_sse10) { // highbd_10_variance (svt_psnr.c:160-177): rounding shifts, then the clamped variance
        const uint32_t sse10 = (uint32_t)((sse + 8) >> 4);
        const i64      sum10 = ((i64)sum + 2) >> 2;
        const i64      var   = (i64)sse10 - div_by_area(sum10 * sum10, w * h);
        if (p.d.var_sse10) p.d.var_sse10[job] = sse10;
        if (p.d.variance10) p.d.variance10[job] = var >= 0 ? (uint32_t)var : 0u;
    }
    // svt_spatial_full_distortion_kernel_facade (picture_operators_c.c:115-174)
    if (p.d.facade_dist)
        p.d.facade_dist[job] = (u64)svt_hip_spy_rd_bias_inline((i64)sse, (uint32_t)w, (uint32_t)h, p.d.pred_mode[job], p.d.compound_type[job],
                                                               p.d.temporal_layer_index, p.d.psy_rd, p.d.spy_rd);
}

template <typename Pix>
__device__ __forceinline__ void block_stats_job(const StatsParams &p, HadLds &L, const uint32_t job, const int lane, const bool with_satd, uint32_t &o_sad, int32_t &o_sum, u64 &o_sse) {
    const SvtHipBlockJob jb = p.d.jobs[job];
    const int w = jb.width, h = jb.height;
    const View<Pix> src = {static_cast<const Pix *>(p.d.src) + jb.src_offset, p.d.src_stride, 16 * (jb.subpel_x & 7), 16 * (jb.subpel_y & 7)};
    const View<Pix> ref = {static_cast<const Pix *>(p.d.ref) + jb.ref_offset, p.d.ref_stride, 0, 0};
    uint32_t sad = 0, sq = 0; // a lane sees at most 128 * 128 / 64 samples: its sum of squares stays below 2^32
    int32_t  sum = 0;
    if (!(src.fx1 | src.fy1) && !(w & 3)) { // uniform: plain blocks, 4 samples per lane and step
        const int   wq = w >> 2;
        const float rq = __builtin_amdgcn_rcpf((float)wq);
        for (int i = lane; i < wq * h; i += 64) {
            const int r = (int)(((float)i + 0.5f) * rq), c = 4 * (i - r * wq);
            quad_stats(src.p + (size_t)r * src.stride + c, ref.p + (size_t)r * ref.stride + c, sad, sum, sq);
        }
    } else {
        const float rw = __builtin_amdgcn_rcpf((float)w);
        for (int i = lane; i < w * h; i += 64) {
            const int r = (int)(((float)i + 0.5f) * rw), c = i - r * w; // exact for i < 2^21
            const int d = src.at(r, c) - ref.at(r, c);
            sad += (uint32_t)(d < 0 ? -d : d);
            sum += d;
            sq += (uint32_t)(d * d);
        }
    }
    sad = wave_sum_dpp(sad);
    sum = (int32_t)wave_sum_dpp((uint32_t)sum);
    const u64      sse  = (u64)wave_sum_dpp(sq & 0xFFFFu) + ((u64)wave_sum_dpp(sq >> 16) << 16);
    o_sad = sad; o_sum = sum; o_sse = sse; // the outputs derived from them are written by the caller, one lane per job of the wave
    if (p.d.satd && with_satd) { // hadamard_path_c: square blocks, <= 32x32 tiles
        uint32_t satd = 0;
        const int n = w < 32 ? w : 32;
        if (w == h && (w == 4 || w == 8 || w == 16 || w == 32 || w == 64 || w == 128)) {
            for (int ty = 0; ty < h; ty += n)
                for (int tx = 0; tx < w; tx += n) {
                    if (!(src.fx1 | src.fy1)) { // uniform: 4 samples per lane and step (n = 4 .. 32: powers of two)
                        const int sh = n == 4 ? 0 : n == 8 ? 1 : n == 16 ? 2 : 3;
                        for (int i = lane; i < (n * n) >> 2; i += 64) {
                            const int r = i >> sh, c = 4 * (i - (r << sh));
                            quad_residual(src.p + (size_t)(ty + r) * src.stride + tx + c, ref.p + (size_t)(ty + r) * ref.stride + tx + c, &L.res[r * kResPitch + c]);
                        }
                    } else
                        for (int i = lane; i < n * n; i += 64) {
                            const int r = i / n, c = i - r * n;
                            L.res[r * kResPitch + c] = (int16_t)((int16_t)src.at(ty + r, tx + c) - (int16_t)ref.at(ty + r, tx + c));
                        }
                    stats_wave_sync();
                    if (n >= 16) satd += hadamard_satd_mfma(L, n, lane); // uniform
                    else {
                        hadamard_tile(L, n, lane);
                        for (int i = lane; i < n * n; i += 64) { const int32_t v = L.c[i]; satd += (uint32_t)(v < 0 ? -v : v); }
                    }
                    stats_wave_sync();
                }
        }
        satd = wave_sum(satd);
        if (lane == 0) p.d.satd[job] = satd;
    }
}

struct FlatLds { // one wave's tiles
    HadLds   L;
    uint32_t tile0[kJobsPerWave + 1], esum[kJobsPerWave]; // psy: first tile of each job in the wave's tile sequence, energy sums
};
// jobs [wave * kJobsPerWave, ...) of the flat list, by one wave
template <typename Pix> __device__ __forceinline__ void block_stats_flat(const StatsParams &p, FlatLds &F, const uint32_t wave, const int lane) {
    HadLds   &L = F.L;
    uint32_t *tile0 = F.tile0, *esum = F.esum;
    const uint32_t j0 = wave * p.jpw, j1 = j0 + p.jpw < p.d.n_jobs ? j0 + p.jpw : p.d.n_jobs;
    const int      nj = (int)(j1 - j0);
    // Four consecutive plain 8x8 blocks share one matrix-core tile: their residuals side by side in the LDS tile, one pair of MFMAs, four SATDs
    uint32_t quad8 = 0; // bit g: jobs 4g .. 4g + 3 of the wave
    if (p.d.satd) { // uniform
        bool plain8 = false;
        if (lane < nj) {
            const SvtHipBlockJob jb = p.d.jobs[j0 + lane];
            plain8 = jb.width == 8 && jb.height == 8 && !(jb.subpel_x & 7) && !(jb.subpel_y & 7);
        }
        const u64 m = __ballot(plain8);
#pragma unroll
        for (int g = 0; g < kJobsPerWave / 4; g++) quad8 |= (((m >> (4 * g)) & 0xF) == 0xF) ? 1u << g : 0u;
    }
    uint32_t my_sad = 0;
    int32_t  my_sum = 0;
    u64      my_sse = 0;
    for (int k = 0; k < nj; k++) {
        uint32_t sad;
        int32_t  sum;
        u64      sse;
        block_stats_job<Pix>(p, L, j0 + k, lane, !((quad8 >> (k >> 2)) & 1), sad, sum, sse);
        if (lane == k) { my_sad = sad; my_sum = sum; my_sse = sse; }
        stats_wave_sync(); // the LDS tile is reused by the next job
    }
    // svt_psy_distortion{,_hbd}: one lane per 8x8 (or 4x4) tile, the tiles of the wave's jobs side by side
    u64 my_e = 0;
    const bool psy = p.d.psy_energy || p.d.psy_dist || (p.d.psy_sse && p.d.psy_rd > 0.0);
    if (psy) { // uniform
        int nt = 0;
        if (lane < nj) {
            const SvtHipBlockJob jb = p.d.jobs[j0 + lane];
            const int n = (jb.width >= 8 && jb.height >= 8) ? 8 : 4; // the reference's loops: i < height; i += n
            nt = ((jb.width + n - 1) / n) * ((jb.height + n - 1) / n);
        }
        int incl = nt; // inclusive prefix over lanes 0 .. 15
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true); // row_shr:1, zero fill
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);
        static_assert(kJobsPerWave <= 16, "the prefix sum runs inside one DPP row");
        if (lane < kJobsPerWave) { tile0[lane + 1] = (uint32_t)incl; esum[lane] = 0; }
        if (lane == 0) tile0[0] = 0;
        stats_wave_sync();
        const int total = (int)tile0[kJobsPerWave];
        for (int t = lane; t < total; t += 64) {
            int k = 0;
#pragma unroll
            for (int i = 1; i < kJobsPerWave; i++) k += t >= (int)tile0[i] ? 1 : 0; // broadcast reads
            const int tl = t - (int)tile0[k];
            const SvtHipBlockJob jb = p.d.jobs[j0 + k];
            const int n = (jb.width >= 8 && jb.height >= 8) ? 8 : 4, ntx = (jb.width + n - 1) / n;
            const int ty = tl / ntx, tx = tl - ty * ntx;
            const View<Pix> src = {static_cast<const Pix *>(p.d.src) + jb.src_offset, p.d.src_stride, 16 * (jb.subpel_x & 7), 16 * (jb.subpel_y & 7)};
            const View<Pix> ref = {static_cast<const Pix *>(p.d.ref) + jb.ref_offset, p.d.ref_stride, 0, 0};
            const int32_t a = psy_tile_energy<Pix>(src.sub(ty * n, tx * n), n);
            const int32_t b = psy_tile_energy<Pix>(ref.sub(ty * n, tx * n), n);
            atomicAdd(&esum[k], (uint32_t)(a > b ? a - b : b - a)); // a job's sum stays below 2^32: 256 tiles x 64 x 64 x 1023
        }
        stats_wave_sync();
        const u64 e = lane < kJobsPerWave ? esum[lane] : 0;
        my_e = sizeof(Pix) == 1 ? e >> 1 : e << 2;
    }
    if (lane < nj) { // lane k finishes job k of the wave
        const uint32_t job = j0 + lane;
        const SvtHipBlockJob jb = p.d.jobs[job];
        write_pixel_outputs(p, job, jb.width, jb.height, my_sad, my_sum, my_sse);
        if (psy) {
            if (p.d.psy_energy) p.d.psy_energy[job] = my_e;
            if (p.d.psy_dist) p.d.psy_dist[job] = (u64)((double)my_e * p.d.psy_rd); // get_svt_psy_full_dist, psy_rd.c:277-293
            if (p.d.psy_sse) p.d.psy_sse[job] = my_sse + (u64)((double)my_e * p.d.psy_rd); // svt_spatial_psy_distortion_kernel_c, picture_operators_c.c:85-112
        } else if (p.d.psy_sse) p.d.psy_sse[job] = my_sse; // psy_rd <= 0: the plain SSE
    }
    for (int g = 0; g < kJobsPerWave / 4; g++)
        if ((quad8 >> g) & 1) { // uniform
            const int q = lane >> 4, r = (lane & 15) >> 1, c = 4 * (lane & 1); // block, row, first column of this lane's 4 samples
            const SvtHipBlockJob jb = p.d.jobs[j0 + 4 * g + q];
            quad_residual(static_cast<const Pix *>(p.d.src) + jb.src_offset + (size_t)r * p.d.src_stride + c, static_cast<const Pix *>(p.d.ref) + jb.ref_offset + (size_t)r * p.d.ref_stride + c,
                          &L.res[(8 * (q >> 1) + r) * kResPitch + 8 * (q & 1) + c]);
            stats_wave_sync();
            int32_t y[4];
            had8x4_mfma(L, 0, 0, lane, had16_weights(lane), y);
            uint32_t sv = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) sv += (uint32_t)(y[j] < 0 ? -y[j] : y[j]);
            // block (k0 / 8, r / 8) = (lane bit 5, lane bit 3): sum over the other four lane bits
            sv += __shfl_xor(sv, 1, 64); sv += __shfl_xor(sv, 2, 64); sv += __shfl_xor(sv, 4, 64); sv += __shfl_xor(sv, 16, 64);
            if ((lane & 23) == 0) p.d.satd[j0 + 4 * g + 2 * (lane >> 5) + ((lane >> 3) & 1)] = sv;
            stats_wave_sync(); // the tile is rewritten by the next group
        }
}

// ---- hierarchical block statistics: one workgroup <-> one 64x64 region.  The region's samples are read once for the pixel statistics (and once
// more, as residuals into the LDS tiles, for hadamard_path); the sums of the nested blocks are DPP reductions.  Output slot of nested block z:
// out0 + z with z = 0 (64x64), 1 + raster (32x32), 5 + raster (16x16), 21 + raster (8x8).
__device__ __forceinline__ uint32_t quad_sum_u32(uint32_t v) { // sum over the 4 lanes of a quad, in all of them
    v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
    return v + (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t row16_sum_of_quads(uint32_t v) { // v uniform inside each quad: sum of the 4 quads of a 16-lane row, in all 16 lanes
    v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141 /* row_half_mirror */, 0xF, 0xF, true);
    return v + (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140 /* row_mirror */, 0xF, 0xF, true);
}
// Without hadamard_path (the psy / facade batches): ONE wave per region, lane <-> one 8x8 block in Morton order (one psy tile per lane: every
// lane busy in the tile transforms), 16x16 = quad, 32x32 = 16-lane row, 64x64 = the wave.
template <typename Pix> __device__ __forceinline__ void block_stats_pyramid1(const StatsParams &p, const uint32_t reg, const int lane) {
    const uint32_t out0 = p.d.pyramid_out_base + SVT_HIP_PYRAMID_BLOCKS * reg;
    const SvtHipBlockJob jb = p.d.pyramids[reg];
    const Pix *src = static_cast<const Pix *>(p.d.src) + jb.src_offset, *ref = static_cast<const Pix *>(p.d.ref) + jb.ref_offset;
    // lane -> 8x8 block coordinates: bits x0 y0 x1 y1 x2 y2
    const int bx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), by = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    uint32_t sad = 0, sq = 0;
    int32_t  sum = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const Pix *s = src + (size_t)(8 * by + r) * p.d.src_stride + 8 * bx, *q = ref + (size_t)(8 * by + r) * p.d.ref_stride + 8 * bx;
        quad_stats(s, q, sad, sum, sq);
        quad_stats(s + 4, q + 4, sad, sum, sq);
    }
    const bool psy = p.d.psy_energy || p.d.psy_dist || (p.d.psy_sse && p.d.psy_rd > 0.0);
    uint32_t e8 = 0; // |energy(src tile) - energy(ref tile)| of this lane's 8x8 tile
    if (psy) { // uniform
        const View<Pix> sv = {src + (size_t)(8 * by) * p.d.src_stride + 8 * bx, p.d.src_stride, 0, 0}, rv = {ref + (size_t)(8 * by) * p.d.ref_stride + 8 * bx, p.d.ref_stride, 0, 0};
        const int32_t a = psy_tile_energy<Pix>(sv, 8), b = psy_tile_energy<Pix>(rv, 8);
        e8 = (uint32_t)(a > b ? a - b : b - a);
    }
    const uint32_t sad16 = quad_sum_u32(sad), sq16 = quad_sum_u32(sq), e16 = quad_sum_u32(e8);
    const int32_t  sum16 = (int32_t)quad_sum_u32((uint32_t)sum);
    const uint32_t sad32 = row16_sum_of_quads(sad16), sq32 = row16_sum_of_quads(sq16), e32 = row16_sum_of_quads(e16); // a 32x32 block: 1024 x 1023^2 < 2^32
    const int32_t  sum32 = (int32_t)row16_sum_of_quads((uint32_t)sum16);
    const uint32_t sad64 = wave_sum_dpp(sad), e64lo = wave_sum_dpp(e8 & 0xFFFFu), e64hi = wave_sum_dpp(e8 >> 16);
    const int32_t  sum64 = (int32_t)wave_sum_dpp((uint32_t)sum);
    const u64      sse64 = (u64)wave_sum_dpp(sq & 0xFFFFu) + ((u64)wave_sum_dpp(sq >> 16) << 16), e64 = (u64)e64lo + ((u64)e64hi << 16);
    auto emit = [&](uint32_t slot, int n, uint32_t a_sad, int32_t a_sum, u64 a_sse, u64 a_e) {
        write_pixel_outputs(p, slot, n, n, a_sad, a_sum, a_sse);
        const u64 e = sizeof(Pix) == 1 ? a_e >> 1 : a_e << 2;
        if (psy) {
            if (p.d.psy_energy) p.d.psy_energy[slot] = e;
            if (p.d.psy_dist) p.d.psy_dist[slot] = (u64)((double)e * p.d.psy_rd);
            if (p.d.psy_sse) p.d.psy_sse[slot] = a_sse + (u64)((double)e * p.d.psy_rd);
        } else if (p.d.psy_sse) p.d.psy_sse[slot] = a_sse;
    };
    emit(out0 + 21 + 8 * by + bx, 8, sad, sum, sq, e8);
    if ((lane & 3) == 0) emit(out0 + 5 + 4 * (by >> 1) + (bx >> 1), 16, sad16, sum16, sq16, e16);
    if ((lane & 15) == 0) emit(out0 + 1 + 2 * (by >> 2) + (bx >> 2), 32, sad32, sum32, sq32, e32);
    if (lane == 0) emit(out0, 64, sad64, sum64, sse64, e64);
}

// With hadamard_path -- four waves per region: wave q <-> 32x32 quadrant q (raster), lane <-> (8x8 block of the quadrant in Morton order, pair of rows): the 8x8 sums are
// quad sums, the 16x16 sums 16-lane row sums, the 32x32 sums wave sums, the 64x64 sums meet in LDS.  Each wave stages its own quadrant's
// residuals once for hadamard_path's 8x8 / 16x16 / 32x32 SATDs.
struct PyrLds {
    int16_t  res[4][32 * kResPitch];
    uint32_t sad[4], e_lo[4], e_hi[4], satd[4];
    int32_t  sum[4];
    u64      sse[4];
};
template <typename Pix> __device__ __forceinline__ void block_stats_pyramid4(const StatsParams &p, PyrLds &S, const uint32_t reg) {
    const int      lane = threadIdx.x & 63, q = threadIdx.x >> 6, qy = q >> 1, qx = q & 1;
    const uint32_t out0 = p.d.pyramid_out_base + SVT_HIP_PYRAMID_BLOCKS * reg;
    const SvtHipBlockJob jb = p.d.pyramids[reg];
    const Pix *src = static_cast<const Pix *>(p.d.src) + jb.src_offset + (size_t)(32 * qy) * p.d.src_stride + 32 * qx;
    const Pix *ref = static_cast<const Pix *>(p.d.ref) + jb.ref_offset + (size_t)(32 * qy) * p.d.ref_stride + 32 * qx;
    // lane = blk * 4 + row pair; blk bits x0 y0 x1 y1 -> 8x8 block (bx, by) of the quadrant
    const int blk = lane >> 2, rp = lane & 3, bx = (blk & 1) | ((blk >> 1) & 2), by = ((blk >> 1) & 1) | ((blk >> 2) & 2);
    uint32_t sad = 0, sq = 0;
    int32_t  sum = 0;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const Pix *sp = src + (size_t)(8 * by + 2 * rp + r) * p.d.src_stride + 8 * bx, *qp = ref + (size_t)(8 * by + 2 * rp + r) * p.d.ref_stride + 8 * bx;
        quad_stats(sp, qp, sad, sum, sq);
        quad_stats(sp + 4, qp + 4, sad, sum, sq);
    }
    const bool psy = p.d.psy_energy || p.d.psy_dist || (p.d.psy_sse && p.d.psy_rd > 0.0);
    uint32_t e8 = 0; // |energy(src tile) - energy(ref tile)| of the block's 8x8 tile: computed by the block's first lane, zero in the others
    if (psy && rp == 0) {
        const View<Pix> sv = {src + (size_t)(8 * by) * p.d.src_stride + 8 * bx, p.d.src_stride, 0, 0}, rv = {ref + (size_t)(8 * by) * p.d.ref_stride + 8 * bx, p.d.ref_stride, 0, 0};
        const int32_t a = psy_tile_energy<Pix>(sv, 8), b = psy_tile_energy<Pix>(rv, 8);
        e8 = (uint32_t)(a > b ? a - b : b - a);
    }
    // the tree: 8x8 = quad, 16x16 = 16-lane row, 32x32 = the wave, 64x64 = the four waves
    const uint32_t sad8 = quad_sum_u32(sad), sq8 = quad_sum_u32(sq), e8s = quad_sum_u32(e8);
    const int32_t  sum8 = (int32_t)quad_sum_u32((uint32_t)sum);
    const uint32_t sad16 = row16_sum_of_quads(sad8), sq16 = row16_sum_of_quads(sq8), e16 = row16_sum_of_quads(e8s);
    const int32_t  sum16 = (int32_t)row16_sum_of_quads((uint32_t)sum8);
    const uint32_t sad32 = wave_sum_dpp(sad), sq32 = wave_sum_dpp(sq), e32lo = wave_sum_dpp(e8 & 0xFFFFu), e32hi = wave_sum_dpp(e8 >> 16); // 1024 x 1023^2 < 2^32
    const int32_t  sum32 = (int32_t)wave_sum_dpp((uint32_t)sum);
    auto emit = [&](uint32_t slot, int n, uint32_t a_sad, int32_t a_sum, u64 a_sse, u64 a_e) {
        write_pixel_outputs(p, slot, n, n, a_sad, a_sum, a_sse);
        const u64 e = sizeof(Pix) == 1 ? a_e >> 1 : a_e << 2;
        if (psy) {
            if (p.d.psy_energy) p.d.psy_energy[slot] = e;
            if (p.d.psy_dist) p.d.psy_dist[slot] = (u64)((double)e * p.d.psy_rd);
            if (p.d.psy_sse) p.d.psy_sse[slot] = a_sse + (u64)((double)e * p.d.psy_rd);
        } else if (p.d.psy_sse) p.d.psy_sse[slot] = a_sse;
    };
    if (rp == 0) emit(out0 + 21 + 8 * (4 * qy + by) + 4 * qx + bx, 8, sad8, sum8, sq8, e8s);
    if ((lane & 15) == 0) emit(out0 + 5 + 4 * (2 * qy + (by >> 1)) + 2 * qx + (bx >> 1), 16, sad16, sum16, sq16, e16);
    if (lane == 0) {
        emit(out0 + 1 + q, 32, sad32, sum32, sq32, (u64)e32lo + ((u64)e32hi << 16));
        S.sad[q] = sad32; S.sum[q] = sum32; S.sse[q] = sq32; S.e_lo[q] = e32lo; S.e_hi[q] = e32hi;
    }
    uint32_t s32 = 0;
    if (p.d.satd) { // uniform; 8-bit planes (checked by the host): hadamard_path of the quadrant's nested blocks from one staged residual
        const had_half4 h = had16_weights(lane);
        HadLds &L = *reinterpret_cast<HadLds *>(S.res[q]); // only .res is touched by the matrix-core path
        for (int i = lane; i < 256; i += 64) {
            const int r = i >> 3, c = 4 * (i & 7);
            quad_residual(src + (size_t)r * p.d.src_stride + c, ref + (size_t)r * p.d.ref_stride + c, &L.res[r * kResPitch + c]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier(); // a wave's own LDS writes are ordered before its later reads; only this wave touches its tile
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        int32_t  o[4][4];
        uint32_t sv8[4], sv16[4];
#pragma unroll
        for (int sb = 0; sb < 4; sb++) { // the quadrant's 16x16 blocks: their four 8x8 SATDs, then their own (the sums are reduced together below)
            int32_t y[4];
            had8x4_mfma(L, sb >> 1, sb & 1, lane, h, y);
            sv8[sb] = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) sv8[sb] += (uint32_t)(y[j] < 0 ? -y[j] : y[j]);
            had16_mfma(L, sb >> 1, sb & 1, lane, h, o[sb]);
            sv16[sb] = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) sv16[sb] += (uint32_t)(o[sb][j] < 0 ? -o[sb][j] : o[sb][j]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { // svt_aom_hadamard_32x32_c's combine of the four 16x16 transforms (picture_operators_c.c:299-326), as hadamard_satd_mfma
            const int b0 = (o[0][j] + o[1][j]) >> 2, b1 = (o[0][j] - o[1][j]) >> 2, b2 = (o[2][j] + o[3][j]) >> 2, b3 = (o[2][j] - o[3][j]) >> 2;
            const int c0 = b0 + b2, c1 = b1 + b3, c2 = b0 - b2, c3 = b1 - b3;
            s32 += (uint32_t)(c0 < 0 ? -c0 : c0) + (uint32_t)(c1 < 0 ? -c1 : c1) + (uint32_t)(c2 < 0 ? -c2 : c2) + (uint32_t)(c3 < 0 ? -c3 : c3);
        }
#pragma unroll
        for (int sb = 0; sb < 4; sb++) {
            // 8x8 block (k0 / 8, r / 8) = (lane bit 5, lane bit 3) of the 16x16: sum over the other four lane bits
            uint32_t sv = sv8[sb];
            sv += __shfl_xor(sv, 1, 64); sv += __shfl_xor(sv, 2, 64); sv += __shfl_xor(sv, 4, 64); sv += __shfl_xor(sv, 16, 64);
            if ((lane & 23) == 0) p.d.satd[out0 + 21 + 8 * (4 * qy + 2 * (sb >> 1) + (lane >> 5)) + 4 * qx + 2 * (sb & 1) + ((lane >> 3) & 1)] = sv;
            const uint32_t s16 = wave_sum_dpp(sv16[sb]);
            if (lane == 0) p.d.satd[out0 + 5 + 4 * (2 * qy + (sb >> 1)) + 2 * qx + (sb & 1)] = s16;
        }
        s32 = wave_sum_dpp(s32);
        if (lane == 0) { p.d.satd[out0 + 1 + q] = s32; S.satd[q] = s32; }
    }
    __syncthreads();
    if (threadIdx.x == 0) { // the 64x64 block
        u64 sse = 0, e = 0;
        uint32_t sd = 0, st = 0;
        int32_t  sm = 0;
        for (int k = 0; k < 4; k++) { sse += S.sse[k]; e += (u64)S.e_lo[k] + ((u64)S.e_hi[k] << 16); sd += S.sad[k]; sm += S.sum[k]; st += S.satd[k]; }
        emit(out0, 64, sd, sm, sse, e);
        if (p.d.satd) p.d.satd[out0] = st; // hadamard_path_c: a 64x64 block is four 32x32 tiles
    }
}

// One launch per batch: the regions first, the flat jobs behind them (two launches on one stream would run one after the other, and a batch of one
// picture is a few microseconds of work per kernel).
// without hadamard_path: a wave per workgroup -- region `blockIdx.x`, or wave (blockIdx.x - n_pyramids) of the flat list
template <typename Pix> __global__ void __launch_bounds__(64) block_stats_kernel(const StatsParams p) {
    __shared__ FlatLds F;
    if (blockIdx.x < p.n_front) block_stats_pyramid1<Pix>(p, blockIdx.x, threadIdx.x);
    else block_stats_flat<Pix>(p, F, blockIdx.x - p.n_front, threadIdx.x);
}
// with hadamard_path: four waves per workgroup -- region `blockIdx.x`, or four waves of the flat list (a workgroup takes one branch as a whole:
// the region's barrier is reached by all of its waves)
union Stats4Lds {
    PyrLds  S;
    FlatLds F[4];
};
template <typename Pix> __global__ void __launch_bounds__(256) block_stats4_kernel(const StatsParams p) {
    __shared__ Stats4Lds U;
    if (blockIdx.x < p.n_front) block_stats_pyramid4<Pix>(p, U.S, blockIdx.x);
    else {
        const uint32_t wave = (blockIdx.x - p.n_front) * 4 + (threadIdx.x >> 6);
        if (wave * p.jpw < p.d.n_jobs) block_stats_flat<Pix>(p, U.F[threadIdx.x >> 6], wave, threadIdx.x & 63);
    }
}

// ---- svt_sad_loop_kernel: one thread per search position, first minimum in raster order through a 64-bit key ------
struct SadLoopParams {
    const uint8_t *src, *ref;
    uint32_t       src_stride, ref_stride, block_height, block_width, src_stride_raw;
    int            sa_w, sa_h, skip_even;
    u64           *best; // initialised to (0xffffff << 32) | 0xffffffff
};
__global__ void __launch_bounds__(256) sad_loop_kernel(const SadLoopParams p) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    u64       key = ~0ull;
    if (idx < p.sa_w * p.sa_h) {
        const int ys = idx / p.sa_w, xs = idx - ys * p.sa_w;
        if (!(p.skip_even && !(ys & 1))) {
            const uint8_t *r0 = p.ref + (size_t)ys * p.src_stride_raw + xs;
            uint32_t       s  = 0;
            for (uint32_t r = 0; r < p.block_height; r++)
                for (uint32_t c = 0; c < p.block_width; c++) {
                    const int d = (int)p.src[r * p.src_stride + c] - (int)r0[r * p.ref_stride + c];
                    s += (uint32_t)(d < 0 ? -d : d);
                }
            key = ((u64)s << 32) | ((u64)(uint32_t)ys << 16) | (uint32_t)xs;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const u64 t = __shfl_xor(key, o, 64); key = t < key ? t : key; }
    if ((threadIdx.x & 63) == 0 && key != ~0ull) atomicMin(p.best, key);
}

// ---- stand-alone Hadamard (coefficients out) ------------------------------------------------------------------------
__global__ void __launch_bounds__(64) hadamard_kernel(const int16_t *src, int stride, int n, int32_t *coeff) {
    __shared__ HadLds L;
    const int lane = threadIdx.x;
    for (int i = lane; i < n * n; i += 64) { const int r = i / n, c = i - r * n; L.res[r * kResPitch + c] = src[r * stride + c]; }
    __syncthreads();
    hadamard_tile(L, n, lane);
    for (int i = lane; i < n * n; i += 64) coeff[i] = L.c[i];
}

__global__ void __launch_bounds__(64) satd_kernel(const int32_t *coeff, int n, int *out) {
    int acc = 0;
    for (int i = threadIdx.x; i < n; i += 64) { const int v = coeff[i]; acc += v < 0 ? -v : v; }
    acc = wave_sum(acc);
    if (threadIdx.x == 0) *out = acc;
}

// svt_av1_compute_cul_level (full_loop.c:1449-1466): min(63, sum over the first eob scan positions of |q|) + the DC sign bits.  A lane's terms
// are clamped to 63 each, which leaves the clamped total unchanged (and keeps the sum far from overflow, like the reference's early exit).
__global__ void __launch_bounds__(64) cul_level_kernel(const int16_t *scan, const int32_t *q, int eob, uint8_t *out) {
    uint32_t acc = 0;
    for (int c = threadIdx.x; c < eob; c += 64) { const int32_t v = q[scan[c]]; const uint32_t a = (uint32_t)(v < 0 ? -v : v); acc += a > 63u ? 63u : a; }
    acc = wave_sum(acc);
    if (threadIdx.x == 0) { const int32_t dc = q[0]; *out = (uint8_t)((acc > 63u ? 63u : acc) + (dc < 0 ? 64u : (dc > 0 ? 128u : 0u))); }
}

// svt_av1_fwht4x4 (transforms.c:3099-3151): 4-point reversible Walsh-Hadamard on columns, then on the rows of the intermediate; lane i < 4 owns
// column i in both passes (the second pass reads the transposed intermediate through LDS).  64-bit temporaries like the reference.
__device__ __forceinline__ void wht4(i64 a, i64 b, i64 c, i64 d, i64 o[4]) {
    a += b; d -= c;
    const i64 e = (a - d) >> 1;
    b = e - b; c = e - c; a -= c; d += b;
    o[0] = a; o[1] = c; o[2] = d; o[3] = b;
}
__global__ void __launch_bounds__(64) fwht4x4_kernel(const int16_t *in, uint32_t stride, int32_t *out) {
    __shared__ int32_t t[16];
    const int i = threadIdx.x;
    i64 o[4];
    if (i < 4) { wht4(in[i], in[stride + i], in[2 * stride + i], in[3 * stride + i], o); for (int k = 0; k < 4; k++) t[4 * i + k] = (int32_t)o[k]; }
    __syncthreads();
    if (i < 4) { wht4(t[i], t[4 + i], t[8 + i], t[12 + i], o); for (int k = 0; k < 4; k++) out[4 * k + i] = (int32_t)(o[k] * 4); }
}

// ---- coefficient-domain distortion (svt_full_distortion_kernel32_bits_c / _cbf_zero32_bits_c, pic_operators.c:150-222;
// svt_av1_block_error_c, common_dsp_rtcd.c:79-91): out[0] = sum (coeff - recon)^2 (recon == nullptr: 0), out[1] = sum coeff^2
// wrap32: svt_av1_block_error_c squares with SQR() on `int` operands -- a 32-bit wrapping product, widened afterwards
__global__ void __launch_bounds__(256) coeff_dist_kernel(const int32_t *coeff, uint32_t cstride, const int32_t *recon, uint32_t rstride, int w, int h, u64 *out,
                                                         int wrap32) {
    u64 d = 0, e = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < w * h; i += gridDim.x * 256) {
        const int r = i / w, c = i - r * w;
        const i64 a = coeff[(size_t)r * cstride + c];
        if (wrap32) {
            const uint32_t t = (uint32_t)a - (uint32_t)(recon ? recon[(size_t)r * rstride + c] : 0), ua = (uint32_t)a;
            d += (u64)(i64)(int32_t)(t * t);
            e += (u64)(i64)(int32_t)(ua * ua);
            continue;
        }
        if (recon) { const i64 t = a - recon[(size_t)r * rstride + c]; d += (u64)(t * t); }
        e += (u64)(a * a);
    }
    d = wave_sum(d); e = wave_sum(e);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], d); atomicAdd(&out[1], e); }
}

// ---- residual (svt_residual_kernel8bit_c / 16bit_c, pic_operators.c:101-148): int16 arithmetic like the reference
template <typename Pix> __global__ void __launch_bounds__(256) residual_kernel(const Pix *in, uint32_t in_stride, const Pix *pred, uint32_t pred_stride, int16_t *res,
                                                                               uint32_t res_stride, int w, int h) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < w * h; i += gridDim.x * 256) {
        const int r = i / w, c = i - r * w;
        res[(size_t)r * res_stride + c] = (int16_t)((int16_t)in[(size_t)r * in_stride + c] - (int16_t)pred[(size_t)r * pred_stride + c]);
    }
}

// ---- process-global state of the pointer-level entries -----------------------------------------------------------
SvtHipContext *g_leaf_ctx = nullptr;
std::mutex     g_leaf_mutex; // the reference calls its kernels from many threads; these entries serialise on one stream
// previous kernels of the slots svt_hip_install_rtcd wrote to, by this library's symbol name
struct LeafPrev { char symbol[96]; void *prev; void **slot; };
LeafPrev   g_leaf_prev[640];
int        g_leaf_nprev = 0;
std::mutex g_leaf_prev_mutex;
std::atomic<unsigned long long> g_leaf_fallbacks{0}, g_leaf_unhandled{0};
std::atomic<int>                g_leaf_inject{0};
char       g_leaf_msg[SVT_HIP_ERR_BYTES] = "";
thread_local int t_leaf_depth = 0;

} // namespace

extern "C" {

int svt_hip_block_stats_batch(SvtHipContext *ctx, const SvtHipBlockStatsDesc *d) {
    if (!ctx || !d) return SVT_HIP_ERR_BAD_PARAM;
    if (d->n_jobs == 0 && d->n_pyramids == 0) return SVT_HIP_OK;
    if (d->bit_depth != 8 && d->bit_depth != 10) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "bit_depth %u", d->bit_depth);
    if (!d->src || !d->ref || (d->n_jobs && !d->jobs) || (d->n_pyramids && !d->pyramids))
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "a mandatory pointer of the block-stats batch is null");
    if (d->satd && d->bit_depth != 8) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "hadamard_path works on 8-bit input (enc_mode_config.c:2186)");
    if ((d->variance10 || d->var_sse10) && d->bit_depth != 10) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "variance10 is defined on 10-bit planes");
    if (d->facade_dist && (!d->pred_mode || !d->compound_type || d->temporal_layer_index > 5))
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "facade_dist needs pred_mode, compound_type and temporal_layer_index <= 5 (got %u)", d->temporal_layer_index);
    hipSetDevice(ctx->device);
    StatsParams p;
    p.d = *d;
    // a wave works through its jobs one after the other: four per wave (shared matrix-core tiles for 8x8 blocks, fewer waves) only when that
    // still gives every CU several waves -- a one-picture batch's edge jobs are otherwise the launch's critical path
    // (hadamard_path batches only: the psy / facade batches finish a wave's four jobs side by side, one tile per lane -- measured 32 us with four
    // jobs per wave, 42 us with one, on the 2160p batch's 3,720 edge jobs)
    p.jpw = (d->satd && d->n_jobs < (uint32_t)ctx->num_cus * 8u * kJobsPerWave) ? 1 : kJobsPerWave;
    const uint32_t flat = (d->n_jobs + p.jpw - 1) / p.jpw; // waves of the flat list
    p.n_front = d->n_pyramids;
    if (d->satd && d->n_pyramids) // (8-bit planes: checked above)
        hipLaunchKernelGGL(block_stats4_kernel<uint8_t>, dim3(d->n_pyramids + (flat + 3) / 4), dim3(256), 0, ctx->stream, p);
    else if (d->bit_depth == 8) hipLaunchKernelGGL(block_stats_kernel<uint8_t>, dim3(d->n_pyramids + flat), dim3(64), 0, ctx->stream, p);
    else hipLaunchKernelGGL(block_stats_kernel<uint16_t>, dim3(d->n_pyramids + flat), dim3(64), 0, ctx->stream, p);
    SVT_HIP_CHECK(ctx, hipGetLastError());
    return SVT_HIP_OK;
}

int svt_hip_leaf_bind(SvtHipContext *ctx) {
    std::lock_guard<std::mutex> lock(g_leaf_mutex);
    g_leaf_ctx = ctx;
    return SVT_HIP_OK;
}

// The `_hip` entry that takes the place of the reference's function pointer `name`: the exported symbol <name>_hip of this library
// (the pointer-level entries carry the reference's pointer names), or the one the short alias table names where the reference's
// pointer and its `_c` body are called differently.
static const void *rtcd_lookup_symbol(const char *name, char *sym, size_t sym_bytes) {
    static const struct { const char *pointer, *symbol; } alias[] = {
        {"svt_nxm_sad_kernel", "svt_nxm_sad_kernel_helper_hip"},          // aom_dsp_rtcd.h:125 -> svt_nxm_sad_kernel_helper_c
        {"svt_aom_quantize_b", "svt_aom_quantize_b_hip"},                 // -> svt_aom_quantize_b_c_ii
        {"svt_aom_sad_16b_kernel", "svt_aom_sad_16b_kernel_hip"},
    };
    if (!name || !*name || strlen(name) > 200) return nullptr;
    Dl_info info;
    if (!dladdr(reinterpret_cast<const void *>(&svt_hip_leaf_bind), &info) || !info.dli_fname) return nullptr;
    void *self = dlopen(info.dli_fname, RTLD_NOW | RTLD_NOLOAD);
    if (!self) return nullptr;
    snprintf(sym, sym_bytes, "%s_hip", name);
    for (const auto &a : alias)
        if (!strcmp(a.pointer, name)) snprintf(sym, sym_bytes, "%s", a.symbol);
    const void *fn = dlsym(self, sym);
    dlclose(self);
    return fn;
}
const void *svt_hip_rtcd_lookup(const char *name) {
    char sym[256];
    return rtcd_lookup_symbol(name, sym, sizeof(sym));
}

// Stores this library's entries into the slots and keeps what each slot held before -- the encoder's own kernel -- as the entry's way
// out: a `_hip` entry that cannot run (no bound context, a device error) calls it with the same arguments (leaf_guard.h).  No context is
// bound: until svt_hip_leaf_bind() every call goes to the previous kernels.
int svt_hip_rtcd_store(const SvtHipRtcdSlot *slots, uint32_t n_slots, uint32_t *n_skipped) {
    if (!slots && n_slots) return SVT_HIP_ERR_BAD_PARAM;
    uint32_t skipped = 0;
    for (uint32_t i = 0; i < n_slots; i++)
        if (!slots[i].slot) return svt_hip_fail(nullptr, SVT_HIP_ERR_BAD_PARAM, "rtcd slot %u (%s): null address", i, slots[i].name ? slots[i].name : "?");
    std::lock_guard<std::mutex> lock(g_leaf_prev_mutex);
    for (uint32_t i = 0; i < n_slots; i++) {
        char sym[256];
        const void *fn = rtcd_lookup_symbol(slots[i].name, sym, sizeof(sym));
        if (!fn) { skipped++; continue; }
        void *prev = *slots[i].slot;
        if (prev == fn) continue; // installed already: keep the previous kernel recorded then
        int k = 0;
        while (k < g_leaf_nprev && strcmp(g_leaf_prev[k].symbol, sym)) k++;
        if (k == g_leaf_nprev) {
            if (g_leaf_nprev == (int)(sizeof(g_leaf_prev) / sizeof(g_leaf_prev[0]))) return svt_hip_fail(nullptr, SVT_HIP_ERR_NO_MEMORY, "rtcd: too many slots");
            snprintf(g_leaf_prev[k].symbol, sizeof(g_leaf_prev[k].symbol), "%s", sym);
            g_leaf_nprev++;
        }
        g_leaf_prev[k].prev = prev;
        g_leaf_prev[k].slot = slots[i].slot;
        *slots[i].slot = const_cast<void *>(fn);
    }
    if (n_skipped) *n_skipped = skipped;
    return SVT_HIP_OK;
}

// What svt_aom_setup_rtcd_internal (Codec/aom_dsp_rtcd.c:188, called at Globals/enc_handle.c:1444-1445) does for a SIMD flavour: assign
// this backend's entries into the encoder's function pointers.  `slots[i].slot` is the ADDRESS of the encoder's pointer variable
// `slots[i].name`.  Names this library has no entry for are left as they are (the encoder keeps its own kernel there) and counted in
// *n_skipped.  Binds `ctx` for the pointer-level entries (they have no context argument).  Without a context nothing is touched: the
// encoder keeps its dispatch.  Call it before init_fn_ptr() (Codec/av1me.c:31, enc_handle.c:1460), which copies pointer VALUES into
// svt_aom_mefn_ptr[].
int svt_hip_install_rtcd(SvtHipContext *ctx, const SvtHipRtcdSlot *slots, uint32_t n_slots, uint32_t *n_skipped) {
    if (!ctx || (!slots && n_slots)) return SVT_HIP_ERR_BAD_PARAM;
    if (int rc = svt_hip_rtcd_store(slots, n_slots, n_skipped)) return rc;
    return svt_hip_leaf_bind(ctx);
}

// Puts the previous kernels back into the slots svt_hip_install_rtcd / svt_hip_rtcd_store wrote to (an encoder that gives the device up).
int svt_hip_uninstall_rtcd(const SvtHipRtcdSlot *slots, uint32_t n_slots) {
    if (!slots && n_slots) return SVT_HIP_ERR_BAD_PARAM;
    std::lock_guard<std::mutex> lock(g_leaf_prev_mutex);
    for (uint32_t i = 0; i < n_slots; i++) {
        if (!slots[i].slot) continue;
        for (int k = 0; k < g_leaf_nprev; k++)
            if (g_leaf_prev[k].slot == slots[i].slot && g_leaf_prev[k].prev) { *slots[i].slot = g_leaf_prev[k].prev; break; }
    }
    return SVT_HIP_OK;
}

// Calls the previous kernels served since the last call (`fallbacks`), calls that failed with no previous kernel to go to (`unhandled`:
// their outputs are zero / untouched), and the last failure's text.  Any pointer may be null.  Returns the sum of both counts (saturated).
int svt_hip_leaf_status(unsigned long long *fallbacks, unsigned long long *unhandled, char *message, size_t message_bytes) {
    const unsigned long long f = g_leaf_fallbacks.exchange(0), u = g_leaf_unhandled.exchange(0);
    if (fallbacks) *fallbacks = f;
    if (unhandled) *unhandled = u;
    if (message && message_bytes) {
        std::lock_guard<std::mutex> lock(g_leaf_prev_mutex);
        snprintf(message, message_bytes, "%s", g_leaf_msg);
    }
    return (int)((f + u) > 0x7fffffffull ? 0x7fffffff : (f + u));
}

// Testing aid: while on, every pointer-level entry behaves as if the device had failed.
void svt_hip_leaf_inject_failure(int on) { g_leaf_inject.store(on ? 1 : 0); }

} // extern "C"

// ---------------------------------------------------------------------------------------------------------
// Pointer-level entries: host pointers in, host results out, synchronous.  Each call stages the few rows it
// needs into the context's scratch buffer, launches, and copies the result back -- a validation / drop-in path,
// three PCIe round trips per call; production goes through the batched entries.  An entry that cannot run (no bound context, a
// device error) hands the call to the kernel the encoder had in the slot before the installer (leaf_guard.h): fail closed, never abort.
// ---------------------------------------------------------------------------------------------------------
// (leaf_guard.h) the helpers of the pointer-level entries: failures throw, the entry's handler hands the call to the previous kernel
[[noreturn]] void leaf_fail(const char *fmt, ...) {
    LeafFailure f;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(f.what, sizeof(f.what), fmt, ap);
    va_end(ap);
    throw f;
}
SvtHipContext *leaf_ctx() {
    if (g_leaf_inject.load()) leaf_fail("injected failure (svt_hip_leaf_inject_failure)");
    if (!g_leaf_ctx) leaf_fail("a _hip leaf kernel was called with no context bound (svt_hip_leaf_bind / svt_hip_install_rtcd)");
    return g_leaf_ctx;
}
void leaf_check(SvtHipContext *, hipError_t e, const char *what) {
    if (e != hipSuccess) leaf_fail("%s failed in a _hip leaf kernel: %s", what, hipGetErrorString(e));
}
// device staging area: [0, bytes) carved by the caller.  The pointer-level entries run one at a time (leaf_mutex) on the context
// stream; lane 0's result buffer is theirs alone (the asynchronous entries use none, the synchronous ones borrow other lanes).
uint8_t *leaf_scratch(SvtHipContext *ctx, size_t bytes) {
    void *pp = nullptr;
    if (svt_hip_scratch(ctx, &ctx->lane[0], bytes, &pp) != SVT_HIP_OK) leaf_fail("out of device memory in a _hip leaf kernel (%zu bytes)", bytes);
    return static_cast<uint8_t *>(pp);
}
std::mutex &leaf_mutex() { return g_leaf_mutex; }
int         leaf_depth() { return t_leaf_depth; }
LeafEnter::LeafEnter() { t_leaf_depth++; }
LeafEnter::~LeafEnter() { t_leaf_depth--; }
const void *leaf_previous(const char *symbol) {
    std::lock_guard<std::mutex> lock(g_leaf_prev_mutex);
    for (int k = 0; k < g_leaf_nprev; k++)
        if (!strcmp(g_leaf_prev[k].symbol, symbol)) return g_leaf_prev[k].prev;
    return nullptr;
}
static void leaf_note(const char *symbol, const LeafFailure &f, const char *how) {
    std::lock_guard<std::mutex> lock(g_leaf_prev_mutex);
    snprintf(g_leaf_msg, sizeof(g_leaf_msg), "%s: %s -- %s", symbol, f.what, how);
    snprintf(svt_hip_err_buf(), SVT_HIP_ERR_BYTES, "%s", g_leaf_msg); // svt_hip_last_error() of the calling thread
}
void leaf_note_fallback(const char *symbol, const LeafFailure &f) {
    g_leaf_fallbacks++;
    leaf_note(symbol, f, "the call went to the kernel the encoder had installed before");
}
void leaf_note_unhandled(const char *symbol, const LeafFailure &f) {
    g_leaf_unhandled++;
    leaf_note(symbol, f, "no previous kernel is known for this entry: nothing was computed");
    fprintf(stderr, "libsvthip: %s\n", g_leaf_msg);
}

namespace {

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// copies `rows` rows of `row_bytes` bytes (host stride `stride_bytes`) to the device, packed with the same stride
void upload_rows(SvtHipContext *ctx, void *dst, const void *src, size_t stride_bytes, size_t rows, size_t row_bytes) {
    if (rows == 0) return;
    leaf_check(ctx, hipMemcpyAsync(dst, src, (rows - 1) * stride_bytes + row_bytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
}

struct StatsOut { uint32_t sad, variance, var_sse, satd; u64 sse, psy_energy, psy_dist; uint32_t variance10, var_sse10; };

// one (src, ref) block through block_stats_kernel
StatsOut leaf_stats(const void *src, size_t src_stride, const void *ref, size_t ref_stride, int w, int h, int bit_depth, bool want_satd,
                    bool want_psy = false, double psy_rd = 0.0, int xo = 0, int yo = 0) {
    std::lock_guard<std::mutex> lock(g_leaf_mutex);
    SvtHipContext *ctx = leaf_ctx();
    hipSetDevice(ctx->device);
    const size_t bpp = bit_depth == 8 ? 1 : 2;
    const int sh = h + (yo ? 1 : 0), sw = w + (xo ? 1 : 0); // the interpolation reads one more row / column
    const size_t sb = align256(((size_t)sh - 1) * src_stride * bpp + (size_t)sw * bpp), rb = align256(((size_t)h - 1) * ref_stride * bpp + (size_t)w * bpp);
    uint8_t *base = leaf_scratch(ctx, sb + rb + 512);
    uint8_t *d_src = base, *d_ref = base + sb, *d_job = d_ref + rb, *d_out = d_job + 256;
    upload_rows(ctx, d_src, src, src_stride * bpp, sh, (size_t)sw * bpp);
    upload_rows(ctx, d_ref, ref, ref_stride * bpp, h, (size_t)w * bpp);
    SvtHipBlockJob job;
    memset(&job, 0, sizeof(job));
    job.width = (uint8_t)w; job.height = (uint8_t)h; job.subpel_x = (uint8_t)xo; job.subpel_y = (uint8_t)yo;
    leaf_check(ctx, hipMemcpyAsync(d_job, &job, sizeof(job), hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    SvtHipBlockStatsDesc d;
    memset(&d, 0, sizeof(d));
    d.bit_depth = (uint8_t)bit_depth; d.n_jobs = 1; d.src_stride = (uint32_t)src_stride; d.ref_stride = (uint32_t)ref_stride;
    d.src = d_src; d.ref = d_ref; d.jobs = reinterpret_cast<const SvtHipBlockJob *>(d_job);
    StatsOut *o = reinterpret_cast<StatsOut *>(d_out);
    d.sad = &o->sad; d.variance = &o->variance; d.var_sse = &o->var_sse; d.sse = reinterpret_cast<uint64_t *>(&o->sse); d.satd = want_satd ? &o->satd : nullptr;
    if (bit_depth == 10) { d.variance10 = &o->variance10; d.var_sse10 = &o->var_sse10; }
    if (want_psy) { d.psy_rd = psy_rd; d.psy_energy = reinterpret_cast<uint64_t *>(&o->psy_energy); d.psy_dist = reinterpret_cast<uint64_t *>(&o->psy_dist); }
    if (svt_hip_block_stats_batch(ctx, &d) != SVT_HIP_OK) leaf_fail("%s", svt_hip_err_buf());
    StatsOut out;
    memset(&out, 0, sizeof(out));
    leaf_check(ctx, hipMemcpyAsync(&out, d_out, sizeof(out), hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    return out;
}

} // namespace

extern "C" {

void svt_sad_loop_kernel_hip(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t block_height,
                             uint32_t block_width, uint64_t *best_sad, int16_t *x_search_center, int16_t *y_search_center,
                             uint32_t src_stride_raw, uint8_t skip_search_line, int16_t search_area_width, int16_t search_area_height) LEAF_TRY
    std::lock_guard<std::mutex> lock(g_leaf_mutex);
    SvtHipContext *ctx = leaf_ctx();
    hipSetDevice(ctx->device);
    *best_sad = 0xffffff;
    if (search_area_width <= 0 || search_area_height <= 0 || block_height == 0 || block_width == 0) return;
    const size_t sb = align256(((size_t)block_height - 1) * src_stride + block_width);
    const size_t ref_rows_bytes = ((size_t)search_area_height - 1) * src_stride_raw + ((size_t)block_height - 1) * ref_stride + block_width + search_area_width - 1;
    const size_t rb = align256(ref_rows_bytes);
    uint8_t *base = leaf_scratch(ctx, sb + rb + 256);
    uint8_t *d_src = base, *d_ref = base + sb;
    u64     *d_best = reinterpret_cast<u64 *>(d_ref + rb);
    leaf_check(ctx, hipMemcpyAsync(d_src, src, ((size_t)block_height - 1) * src_stride + block_width, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(d_ref, ref, ref_rows_bytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    const u64 init = (0xffffffull << 32) | 0xffffffffull;
    leaf_check(ctx, hipMemcpyAsync(d_best, &init, sizeof(init), hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    SadLoopParams p;
    p.src = d_src; p.ref = d_ref; p.src_stride = src_stride; p.ref_stride = ref_stride; p.block_height = block_height; p.block_width = block_width;
    p.src_stride_raw = src_stride_raw; p.sa_w = search_area_width; p.sa_h = search_area_height;
    p.skip_even = (block_width == 16 && block_height <= 16 && skip_search_line) ? 1 : 0;
    p.best = d_best;
    const int npos = (int)search_area_width * (int)search_area_height;
    hipLaunchKernelGGL(sad_loop_kernel, dim3((npos + 255) / 256), dim3(256), 0, ctx->stream, p);
    leaf_check(ctx, hipGetLastError(), "sad_loop_kernel launch");
    u64 best = 0;
    leaf_check(ctx, hipMemcpyAsync(&best, d_best, sizeof(best), hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    if ((uint32_t)(best >> 32) < 0xffffff) { // strict `<` against the initial value, like the reference
        *best_sad        = best >> 32;
        *x_search_center = (int16_t)(best & 0xFFFF);
        *y_search_center = (int16_t)((best >> 16) & 0xFFFF);
    }
LEAF_CATCH(svt_sad_loop_kernel_hip, src, src_stride, ref, ref_stride, block_height, block_width, best_sad, x_search_center, y_search_center, src_stride_raw, skip_search_line, search_area_width, search_area_height)

uint32_t svt_nxm_sad_kernel_helper_hip(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride, uint32_t height, uint32_t width) LEAF_TRY
    return leaf_stats(src, src_stride, ref, ref_stride, (int)width, (int)height, 8, false).sad;
LEAF_CATCH(svt_nxm_sad_kernel_helper_hip, src, src_stride, ref, ref_stride, height, width)

uint32_t svt_aom_sad_16b_kernel_hip(uint16_t *src, uint32_t src_stride, uint16_t *ref, uint32_t ref_stride, uint32_t height, uint32_t width) LEAF_TRY
    return leaf_stats(src, src_stride, ref, ref_stride, (int)width, (int)height, 10, false).sad;
LEAF_CATCH(svt_aom_sad_16b_kernel_hip, src, src_stride, ref, ref_stride, height, width)

unsigned int svt_aom_variance_hip(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride, int width, int height, unsigned int *sse) LEAF_TRY
    const StatsOut o = leaf_stats(src, (size_t)src_stride, ref, (size_t)ref_stride, width, height, 8, false);
    *sse = o.var_sse;
    return o.variance;
LEAF_CATCH(svt_aom_variance_hip, src, src_stride, ref, ref_stride, width, height, sse)

unsigned int svt_aom_sub_pixel_variance_hip(const uint8_t *src, int src_stride, int xoffset, int yoffset, const uint8_t *ref, int ref_stride, int width,
                                            int height, unsigned int *sse) LEAF_TRY
    const StatsOut o = leaf_stats(src, (size_t)src_stride, ref, (size_t)ref_stride, width, height, 8, false, false, 0.0, xoffset & 7, yoffset & 7);
    *sse = o.var_sse;
    return o.variance;
LEAF_CATCH(svt_aom_sub_pixel_variance_hip, src, src_stride, xoffset, yoffset, ref, ref_stride, width, height, sse)

#define SVT_HIP_VAR(W, H)                                                                                                             \
    unsigned int svt_aom_variance##W##x##H##_hip(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride, unsigned int *sse) LEAF_TRY \
        return svt_aom_variance_hip(src, src_stride, ref, ref_stride, W, H, sse);                                                       \
    LEAF_CATCH(svt_aom_variance##W##x##H##_hip, src, src_stride, ref, ref_stride, sse)                                                  \
    unsigned int svt_aom_sub_pixel_variance##W##x##H##_hip(const uint8_t *src, int src_stride, int xoffset, int yoffset, const uint8_t *ref, \
                                                           int ref_stride, unsigned int *sse) LEAF_TRY                               \
        return svt_aom_sub_pixel_variance_hip(src, src_stride, xoffset, yoffset, ref, ref_stride, W, H, sse);                           \
    LEAF_CATCH(svt_aom_sub_pixel_variance##W##x##H##_hip, src, src_stride, xoffset, yoffset, ref, ref_stride, sse)
SVT_HIP_VAR(4, 4) SVT_HIP_VAR(4, 8) SVT_HIP_VAR(4, 16) SVT_HIP_VAR(8, 4) SVT_HIP_VAR(8, 8) SVT_HIP_VAR(8, 16) SVT_HIP_VAR(8, 32)
SVT_HIP_VAR(16, 4) SVT_HIP_VAR(16, 8) SVT_HIP_VAR(16, 16) SVT_HIP_VAR(16, 32) SVT_HIP_VAR(16, 64) SVT_HIP_VAR(32, 8) SVT_HIP_VAR(32, 16)
SVT_HIP_VAR(32, 32) SVT_HIP_VAR(32, 64) SVT_HIP_VAR(64, 16) SVT_HIP_VAR(64, 32) SVT_HIP_VAR(64, 64) SVT_HIP_VAR(64, 128) SVT_HIP_VAR(128, 64)
SVT_HIP_VAR(128, 128)
#undef SVT_HIP_VAR

int64_t svt_aom_sse_hip(const uint8_t *a, int a_stride, const uint8_t *b, int b_stride, int width, int height) LEAF_TRY
    return (int64_t)leaf_stats(a, (size_t)a_stride, b, (size_t)b_stride, width, height, 8, false).sse;
LEAF_CATCH(svt_aom_sse_hip, a, a_stride, b, b_stride, width, height)

// svt_aom_highbd_sse (aom_dsp_rtcd.h:56; enc_inter_prediction.c:559-570): the uint8_t pointers ARE the uint16_t pointers (plain cast there)
int64_t svt_aom_highbd_sse_hip(const uint8_t *a8, int a_stride, const uint8_t *b8, int b_stride, int width, int height) LEAF_TRY
    return (int64_t)leaf_stats(reinterpret_cast<const uint16_t *>(a8), (size_t)a_stride, reinterpret_cast<const uint16_t *>(b8), (size_t)b_stride, width, height, 10,
                               false).sse;
LEAF_CATCH(svt_aom_highbd_sse_hip, a8, a_stride, b8, b_stride, width, height)

uint64_t svt_spatial_full_distortion_kernel_hip(uint8_t *input, uint32_t input_offset, uint32_t input_stride, uint8_t *recon, int32_t recon_offset,
                                                uint32_t recon_stride, uint32_t area_width, uint32_t area_height) LEAF_TRY
    return leaf_stats(input + input_offset, input_stride, recon + recon_offset, recon_stride, (int)area_width, (int)area_height, 8, false).sse;
LEAF_CATCH(svt_spatial_full_distortion_kernel_hip, input, input_offset, input_stride, recon, recon_offset, recon_stride, area_width, area_height)

uint64_t svt_full_distortion_kernel16_bits_hip(uint8_t *input, uint32_t input_offset, uint32_t input_stride, uint8_t *recon, int32_t recon_offset,
                                               uint32_t recon_stride, uint32_t area_width, uint32_t area_height) LEAF_TRY
    return leaf_stats(reinterpret_cast<uint16_t *>(input) + input_offset, input_stride, reinterpret_cast<uint16_t *>(recon) + recon_offset, recon_stride,
                      (int)area_width, (int)area_height, 10, false).sse;
LEAF_CATCH(svt_full_distortion_kernel16_bits_hip, input, input_offset, input_stride, recon, recon_offset, recon_stride, area_width, area_height)

uint64_t svt_hip_spy_rd_bias(uint64_t sse, uint32_t area_width, uint32_t area_height, uint8_t mode, uint8_t compound_type, uint8_t temporal_layer_index,
                             double psy_rd, uint8_t spy_rd) {
    return (uint64_t)svt_hip_spy_rd_bias_inline((int64_t)sse, area_width, area_height, mode, compound_type, temporal_layer_index, psy_rd, spy_rd);
}

uint64_t svt_spatial_full_distortion_kernel_facade_hip(uint8_t *input, uint32_t input_offset, uint32_t input_stride, uint8_t *recon, int32_t recon_offset,
                                                       uint32_t recon_stride, uint32_t area_width, uint32_t area_height, bool hbd_md, uint8_t mode,
                                                       uint8_t compound_type, uint8_t temporal_layer_index, double psy_rd, uint8_t spy_rd) LEAF_TRY
    const uint64_t sse = hbd_md ? svt_full_distortion_kernel16_bits_hip(input, input_offset, input_stride, recon, recon_offset, recon_stride, area_width, area_height)
                                : svt_spatial_full_distortion_kernel_hip(input, input_offset, input_stride, recon, recon_offset, recon_stride, area_width, area_height);
    return svt_hip_spy_rd_bias(sse, area_width, area_height, mode, compound_type, temporal_layer_index, psy_rd, spy_rd);
LEAF_CATCH(svt_spatial_full_distortion_kernel_facade_hip, input, input_offset, input_stride, recon, recon_offset, recon_stride, area_width, area_height, hbd_md, mode, compound_type, temporal_layer_index, psy_rd, spy_rd)

uint64_t svt_spatial_psy_distortion_kernel_hip(uint8_t *input, uint32_t input_offset, uint32_t input_stride, uint8_t *recon, int32_t recon_offset,
                                               uint32_t recon_stride, uint32_t area_width, uint32_t area_height, double psy_rd) LEAF_TRY
    const StatsOut o = leaf_stats(input + input_offset, input_stride, recon + recon_offset, recon_stride, (int)area_width, (int)area_height, 8, false,
                                  psy_rd > 0.0, psy_rd);
    return o.sse + (psy_rd > 0.0 ? o.psy_dist : 0);
LEAF_CATCH(svt_spatial_psy_distortion_kernel_hip, input, input_offset, input_stride, recon, recon_offset, recon_stride, area_width, area_height, psy_rd)

uint64_t svt_psy_distortion_hip(const uint8_t *input, uint32_t input_stride, const uint8_t *recon, uint32_t recon_stride, uint32_t width, uint32_t height) LEAF_TRY
    return leaf_stats(input, input_stride, recon, recon_stride, (int)width, (int)height, 8, false, true, 0.0).psy_energy;
LEAF_CATCH(svt_psy_distortion_hip, input, input_stride, recon, recon_stride, width, height)
uint64_t svt_psy_distortion_hbd_hip(const uint16_t *input, uint32_t input_stride, const uint16_t *recon, uint32_t recon_stride, uint32_t width, uint32_t height) LEAF_TRY
    return leaf_stats(input, input_stride, recon, recon_stride, (int)width, (int)height, 10, false, true, 0.0).psy_energy;
LEAF_CATCH(svt_psy_distortion_hbd_hip, input, input_stride, recon, recon_stride, width, height)
uint64_t get_svt_psy_full_dist_hip(const void *s, uint32_t so, uint32_t sp, const void *r, uint32_t ro, uint32_t rp, uint32_t w, uint32_t h, uint8_t is_hbd,
                                   double psy_rd) LEAF_TRY
    if (is_hbd == 1)
        return leaf_stats(static_cast<const uint16_t *>(s) + so, sp, static_cast<const uint16_t *>(r) + ro, rp, (int)w, (int)h, 10, false, true, psy_rd).psy_dist;
    return leaf_stats(static_cast<const uint8_t *>(s) + so, sp, static_cast<const uint8_t *>(r) + ro, rp, (int)w, (int)h, 8, false, true, psy_rd).psy_dist;
LEAF_CATCH(get_svt_psy_full_dist_hip, s, so, sp, r, ro, rp, w, h, is_hbd, psy_rd)

uint32_t svt_hip_hadamard_path(const uint8_t *input, uint32_t input_stride, const uint8_t *pred, uint32_t pred_stride, uint32_t block_size_wide) LEAF_TRY
    return leaf_stats(input, input_stride, pred, pred_stride, (int)block_size_wide, (int)block_size_wide, 8, true).satd;
LEAF_CATCH(svt_hip_hadamard_path, input, input_stride, pred, pred_stride, block_size_wide)

static void leaf_hadamard(const int16_t *src_diff, ptrdiff_t src_stride, int32_t *coeff, int n) {
    std::lock_guard<std::mutex> lock(g_leaf_mutex);
    SvtHipContext *ctx = leaf_ctx();
    hipSetDevice(ctx->device);
    const size_t sb = align256((((size_t)n - 1) * (size_t)src_stride + n) * 2);
    uint8_t *base = leaf_scratch(ctx, sb + (size_t)n * n * 4);
    leaf_check(ctx, hipMemcpyAsync(base, src_diff, (((size_t)n - 1) * (size_t)src_stride + n) * 2, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    hipLaunchKernelGGL(hadamard_kernel, dim3(1), dim3(64), 0, ctx->stream, reinterpret_cast<const int16_t *>(base), (int)src_stride, n, reinterpret_cast<int32_t *>(base + sb));
    leaf_check(ctx, hipGetLastError(), "hadamard_kernel launch");
    leaf_check(ctx, hipMemcpyAsync(coeff, base + sb, (size_t)n * n * 4, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}
void svt_aom_hadamard_4x4_hip(const int16_t *src_diff, ptrdiff_t src_stride, int32_t *coeff) LEAF_TRY leaf_hadamard(src_diff, src_stride, coeff, 4); LEAF_CATCH(svt_aom_hadamard_4x4_hip, src_diff, src_stride, coeff)
void svt_aom_hadamard_8x8_hip(const int16_t *src_diff, ptrdiff_t src_stride, int32_t *coeff) LEAF_TRY leaf_hadamard(src_diff, src_stride, coeff, 8); LEAF_CATCH(svt_aom_hadamard_8x8_hip, src_diff, src_stride, coeff)
void svt_aom_hadamard_16x16_hip(const int16_t *src_diff, ptrdiff_t src_stride, int32_t *coeff) LEAF_TRY leaf_hadamard(src_diff, src_stride, coeff, 16); LEAF_CATCH(svt_aom_hadamard_16x16_hip, src_diff, src_stride, coeff)
void svt_aom_hadamard_32x32_hip(const int16_t *src_diff, ptrdiff_t src_stride, int32_t *coeff) LEAF_TRY leaf_hadamard(src_diff, src_stride, coeff, 32); LEAF_CATCH(svt_aom_hadamard_32x32_hip, src_diff, src_stride, coeff)

int svt_aom_satd_hip(const int32_t *coeff, int length) LEAF_TRY
    std::lock_guard<std::mutex> lock(g_leaf_mutex);
    SvtHipContext *ctx = leaf_ctx();
    hipSetDevice(ctx->device);
    if (length <= 0) return 0;
    const size_t cb = align256((size_t)length * 4);
    uint8_t *base = leaf_scratch(ctx, cb + 256);
    leaf_check(ctx, hipMemcpyAsync(base, coeff, (size_t)length * 4, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    hipLaunchKernelGGL(satd_kernel, dim3(1), dim3(64), 0, ctx->stream, reinterpret_cast<const int32_t *>(base), length, reinterpret_cast<int *>(base + cb));
    leaf_check(ctx, hipGetLastError(), "satd_kernel launch");
    int out = 0;
    leaf_check(ctx, hipMemcpyAsync(&out, base + cb, sizeof(out), hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    return out;
LEAF_CATCH(svt_aom_satd_hip, coeff, length)

// svt_av1_compute_cul_level (aom_dsp_rtcd.h:904): the prototype carries no array length; scan[0 .. eob) and the coefficients those
// positions (and position 0) name are what the reference reads, so that is what travels
uint8_t svt_av1_compute_cul_level_hip(const int16_t *const scan, const int32_t *const quant_coeff, uint16_t *eob) LEAF_TRY
    std::lock_guard<std::mutex> lock(g_leaf_mutex);
    SvtHipContext *ctx = leaf_ctx();
    hipSetDevice(ctx->device);
    const int n = *eob;
    int       top = 0;
    for (int c = 0; c < n; c++) top = scan[c] > top ? scan[c] : top;
    const size_t sb = align256((size_t)(n ? n : 1) * 2), qb = align256((size_t)(top + 1) * 4);
    uint8_t *base = leaf_scratch(ctx, sb + qb + 256);
    if (n) leaf_check(ctx, hipMemcpyAsync(base, scan, (size_t)n * 2, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(base + sb, quant_coeff, (size_t)(top + 1) * 4, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    hipLaunchKernelGGL(cul_level_kernel, dim3(1), dim3(64), 0, ctx->stream, reinterpret_cast<const int16_t *>(base), reinterpret_cast<const int32_t *>(base + sb), n, base + sb + qb);
    leaf_check(ctx, hipGetLastError(), "cul_level_kernel launch");
    uint8_t out = 0;
    leaf_check(ctx, hipMemcpyAsync(&out, base + sb + qb, 1, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    return out;
LEAF_CATCH(svt_av1_compute_cul_level_hip, scan, quant_coeff, eob)

// svt_av1_fwht4x4 (aom_dsp_rtcd.h:208)
void svt_av1_fwht4x4_hip(int16_t *input, int32_t *output, uint32_t stride) LEAF_TRY
    std::lock_guard<std::mutex> lock(g_leaf_mutex);
    SvtHipContext *ctx = leaf_ctx();
    hipSetDevice(ctx->device);
    const size_t ib = align256((3 * (size_t)stride + 4) * 2);
    uint8_t *base = leaf_scratch(ctx, ib + 256);
    leaf_check(ctx, hipMemcpyAsync(base, input, (3 * (size_t)stride + 4) * 2, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    hipLaunchKernelGGL(fwht4x4_kernel, dim3(1), dim3(64), 0, ctx->stream, reinterpret_cast<const int16_t *>(base), stride, reinterpret_cast<int32_t *>(base + ib));
    leaf_check(ctx, hipGetLastError(), "fwht4x4_kernel launch");
    leaf_check(ctx, hipMemcpyAsync(output, base + ib, 16 * 4, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
LEAF_CATCH(svt_av1_fwht4x4_hip, input, output, stride)

// get_hvs_modulation_factor (psy_rd.c:295-307): host arithmetic
double svt_hip_hvs_modulation_factor(double psy_rd, int is_islice, uint8_t temporal_layer_index) {
    if (is_islice) return psy_rd * 0.4;
    if (temporal_layer_index == 0) return psy_rd * 0.75;
    if (temporal_layer_index == 1) return psy_rd * 0.9;
    if (temporal_layer_index == 2) return psy_rd * 0.95;
    return psy_rd;
}


// svt_aom_sad{W}x{H} and the four-reference form (aom_dsp_rtcd.h:267-347; macros at C_DEFAULT/compute_sad_c.c:117-207)
#define SVT_HIP_SAD(W, H)                                                                                                              \
    uint32_t svt_aom_sad##W##x##H##_hip(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride) LEAF_TRY              \
        return leaf_stats(src, (size_t)src_stride, ref, (size_t)ref_stride, W, H, 8, false).sad;                                        \
    LEAF_CATCH(svt_aom_sad##W##x##H##_hip, src, src_stride, ref, ref_stride)                                                           \
    void svt_aom_sad##W##x##H##x4d_hip(const uint8_t *src, int src_stride, const uint8_t *const ref[], int ref_stride, uint32_t *sad_array) LEAF_TRY \
        for (int i = 0; i < 4; i++) sad_array[i] = leaf_stats(src, (size_t)src_stride, ref[i], (size_t)ref_stride, W, H, 8, false).sad; \
    LEAF_CATCH(svt_aom_sad##W##x##H##x4d_hip, src, src_stride, ref, ref_stride, sad_array)
SVT_HIP_SAD(4, 4) SVT_HIP_SAD(4, 8) SVT_HIP_SAD(4, 16) SVT_HIP_SAD(8, 4) SVT_HIP_SAD(8, 8) SVT_HIP_SAD(8, 16) SVT_HIP_SAD(8, 32)
SVT_HIP_SAD(16, 4) SVT_HIP_SAD(16, 8) SVT_HIP_SAD(16, 16) SVT_HIP_SAD(16, 32) SVT_HIP_SAD(16, 64) SVT_HIP_SAD(32, 8) SVT_HIP_SAD(32, 16)
SVT_HIP_SAD(32, 32) SVT_HIP_SAD(32, 64) SVT_HIP_SAD(64, 16) SVT_HIP_SAD(64, 32) SVT_HIP_SAD(64, 64) SVT_HIP_SAD(64, 128) SVT_HIP_SAD(128, 64)
SVT_HIP_SAD(128, 128)
#undef SVT_HIP_SAD

// svt_aom_highbd_10_variance{W}x{H} (aom_dsp_rtcd.h:546-568): the uint8_t pointers carry uint16_t addresses >> 1 (CONVERT_TO_SHORTPTR)
#define SVT_HIP_VAR10(W, H)                                                                                                           \
    unsigned int svt_aom_highbd_10_variance##W##x##H##_hip(const uint8_t *src8, int src_stride, const uint8_t *ref8, int ref_stride, unsigned int *sse) LEAF_TRY \
        const StatsOut o = leaf_stats(reinterpret_cast<const uint16_t *>(reinterpret_cast<uintptr_t>(src8) << 1), (size_t)src_stride,  \
                                      reinterpret_cast<const uint16_t *>(reinterpret_cast<uintptr_t>(ref8) << 1), (size_t)ref_stride, W, H, 10, false); \
        *sse = o.var_sse10;                                                                                                            \
        return o.variance10;                                                                                                           \
    LEAF_CATCH(svt_aom_highbd_10_variance##W##x##H##_hip, src8, src_stride, ref8, ref_stride, sse)
SVT_HIP_VAR10(4, 4) SVT_HIP_VAR10(4, 8) SVT_HIP_VAR10(4, 16) SVT_HIP_VAR10(8, 4) SVT_HIP_VAR10(8, 8) SVT_HIP_VAR10(8, 16) SVT_HIP_VAR10(8, 32)
SVT_HIP_VAR10(16, 4) SVT_HIP_VAR10(16, 8) SVT_HIP_VAR10(16, 16) SVT_HIP_VAR10(16, 32) SVT_HIP_VAR10(16, 64) SVT_HIP_VAR10(32, 8) SVT_HIP_VAR10(32, 16)
SVT_HIP_VAR10(32, 32) SVT_HIP_VAR10(32, 64) SVT_HIP_VAR10(64, 16) SVT_HIP_VAR10(64, 32) SVT_HIP_VAR10(64, 64) SVT_HIP_VAR10(64, 128) SVT_HIP_VAR10(128, 64)
SVT_HIP_VAR10(128, 128)
#undef SVT_HIP_VAR10

uint32_t svt_aom_variance_highbd_hip(const uint16_t *a, int a_stride, const uint16_t *b, int b_stride, int w, int h, uint32_t *sse) LEAF_TRY
    const StatsOut o = leaf_stats(a, (size_t)a_stride, b, (size_t)b_stride, w, h, 10, false);
    *sse = o.var_sse;
    return o.variance;
LEAF_CATCH(svt_aom_variance_highbd_hip, a, a_stride, b, b_stride, w, h, sse)

static void leaf_coeff_dist(const int32_t *coeff, uint32_t cstride, const int32_t *recon, uint32_t rstride, uint32_t w, uint32_t h, uint64_t out[2], int wrap32 = 0) {
    std::lock_guard<std::mutex> lock(g_leaf_mutex);
    SvtHipContext *ctx = leaf_ctx();
    hipSetDevice(ctx->device);
    out[0] = out[1] = 0;
    if (!w || !h) return;
    const size_t cb = align256((((size_t)h - 1) * cstride + w) * 4), rb = recon ? align256((((size_t)h - 1) * rstride + w) * 4) : 0;
    uint8_t *base = leaf_scratch(ctx, cb + rb + 256);
    leaf_check(ctx, hipMemcpyAsync(base, coeff, (((size_t)h - 1) * cstride + w) * 4, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    if (recon) leaf_check(ctx, hipMemcpyAsync(base + cb, recon, (((size_t)h - 1) * rstride + w) * 4, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    u64 *d_out = reinterpret_cast<u64 *>(base + cb + rb);
    leaf_check(ctx, hipMemsetAsync(d_out, 0, 16, ctx->stream), "hipMemsetAsync");
    const int n = (int)(w * h), grid = n < 256 * 64 ? (n + 255) / 256 : 64;
    hipLaunchKernelGGL(coeff_dist_kernel, dim3(grid), dim3(256), 0, ctx->stream, reinterpret_cast<const int32_t *>(base), cstride,
                       recon ? reinterpret_cast<const int32_t *>(base + cb) : nullptr, rstride, (int)w, (int)h, d_out, wrap32);
    leaf_check(ctx, hipGetLastError(), "coeff_dist_kernel launch");
    leaf_check(ctx, hipMemcpyAsync(out, d_out, 16, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}
void svt_full_distortion_kernel32_bits_hip(int32_t *coeff, uint32_t coeff_stride, int32_t *recon_coeff, uint32_t recon_coeff_stride, uint64_t distortion_result[2],
                                           uint32_t area_width, uint32_t area_height) LEAF_TRY
    leaf_coeff_dist(coeff, coeff_stride, recon_coeff, recon_coeff_stride, area_width, area_height, distortion_result);
LEAF_CATCH(svt_full_distortion_kernel32_bits_hip, coeff, coeff_stride, recon_coeff, recon_coeff_stride, distortion_result, area_width, area_height)
void svt_full_distortion_kernel_cbf_zero32_bits_hip(int32_t *coeff, uint32_t coeff_stride, uint64_t distortion_result[2], uint32_t area_width, uint32_t area_height) LEAF_TRY
    uint64_t o[2];
    leaf_coeff_dist(coeff, coeff_stride, nullptr, 0, area_width, area_height, o);
    distortion_result[0] = o[1]; // DIST_CALC_RESIDUAL = DIST_CALC_PREDICTION = sum coeff^2 (pic_operators.c:202-222)
    distortion_result[1] = o[1];
LEAF_CATCH(svt_full_distortion_kernel_cbf_zero32_bits_hip, coeff, coeff_stride, distortion_result, area_width, area_height)
int64_t svt_av1_block_error_hip(const int32_t *coeff, const int32_t *dqcoeff, intptr_t block_size, int64_t *ssz) LEAF_TRY
    uint64_t o[2];
    leaf_coeff_dist(coeff, (uint32_t)block_size, dqcoeff, (uint32_t)block_size, (uint32_t)block_size, 1, o, 1);
    *ssz = (int64_t)o[1];
    return (int64_t)o[0];
LEAF_CATCH(svt_av1_block_error_hip, coeff, dqcoeff, block_size, ssz)

} // extern "C"
template <typename Pix> static void leaf_residual(const Pix *in, uint32_t in_stride, const Pix *pred, uint32_t pred_stride, int16_t *res, uint32_t res_stride, uint32_t w, uint32_t h) {
    std::lock_guard<std::mutex> lock(g_leaf_mutex);
    SvtHipContext *ctx = leaf_ctx();
    hipSetDevice(ctx->device);
    if (!w || !h) return;
    const size_t ib = align256((((size_t)h - 1) * in_stride + w) * sizeof(Pix)), pb = align256((((size_t)h - 1) * pred_stride + w) * sizeof(Pix));
    const size_t rbytes = (((size_t)h - 1) * res_stride + w) * 2;
    uint8_t *base = leaf_scratch(ctx, ib + pb + align256(rbytes));
    leaf_check(ctx, hipMemcpyAsync(base, in, (((size_t)h - 1) * in_stride + w) * sizeof(Pix), hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(base + ib, pred, (((size_t)h - 1) * pred_stride + w) * sizeof(Pix), hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    // the rows between the block's columns belong to the caller: bring them over so that the copy back leaves them unchanged
    leaf_check(ctx, hipMemcpyAsync(base + ib + pb, res, rbytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    const int n = (int)(w * h), grid = n < 256 * 64 ? (n + 255) / 256 : 64;
    hipLaunchKernelGGL(residual_kernel<Pix>, dim3(grid), dim3(256), 0, ctx->stream, reinterpret_cast<const Pix *>(base), in_stride, reinterpret_cast<const Pix *>(base + ib),
                       pred_stride, reinterpret_cast<int16_t *>(base + ib + pb), res_stride, (int)w, (int)h);
    leaf_check(ctx, hipGetLastError(), "residual_kernel launch");
    leaf_check(ctx, hipMemcpyAsync(res, base + ib + pb, rbytes, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}
extern "C" {
void svt_residual_kernel8bit_hip(uint8_t *input, uint32_t input_stride, uint8_t *pred, uint32_t pred_stride, int16_t *residual, uint32_t residual_stride,
                                 uint32_t area_width, uint32_t area_height) LEAF_TRY
    leaf_residual<uint8_t>(input, input_stride, pred, pred_stride, residual, residual_stride, area_width, area_height);
LEAF_CATCH(svt_residual_kernel8bit_hip, input, input_stride, pred, pred_stride, residual, residual_stride, area_width, area_height)
void svt_residual_kernel16bit_hip(uint16_t *input, uint32_t input_stride, uint16_t *pred, uint32_t pred_stride, int16_t *residual, uint32_t residual_stride,
                                  uint32_t area_width, uint32_t area_height) LEAF_TRY
    leaf_residual<uint16_t>(input, input_stride, pred, pred_stride, residual, residual_stride, area_width, area_height);
LEAF_CATCH(svt_residual_kernel16bit_hip, input, input_stride, pred, pred_stride, residual, residual_stride, area_width, area_height)

// svt_aom_estimate_transform (Codec/transforms.c:3158-3225) without the pcs / ctx arguments (they only select the lossless WHT):
// int16 residual -> packed coefficients (min(W,32) x min(H,32)) + the energy of the discarded frequencies, through the fused
// RD kernel (a uint16 plane holding the residual's bit pattern against an all-zero prediction reproduces the residual exactly)
int svt_hip_estimate_transform(int16_t *residual, uint32_t residual_stride, int32_t *coeff, int tx_size, uint64_t *three_quad_energy, int tx_type, int pf_shape) {
    if (tx_size < 0 || tx_size >= SVT_HIP_TX_SIZES_ALL || tx_type < 0 || tx_type >= SVT_HIP_TX_TYPES || pf_shape < 0 || pf_shape > 3 || !residual || !coeff) return SVT_HIP_ERR_BAD_PARAM;
    try {
    std::lock_guard<std::mutex> lock(g_leaf_mutex);
    SvtHipContext *ctx = leaf_ctx();
    hipSetDevice(ctx->device);
    const int W = svt_hip_tx_size_wide(tx_size), H = svt_hip_tx_size_high(tx_size), NP = (W > 32 ? 32 : W) * (H > 32 ? 32 : H);
    const size_t sb = align256((((size_t)H - 1) * residual_stride + W) * 2), zb = align256((size_t)W * H * 2);
    uint8_t *base = leaf_scratch(ctx, sb + zb + 256 + 256 + 256 + align256((size_t)NP * 4));
    uint8_t *d_src = base, *d_zero = d_src + sb, *d_job = d_zero + zb, *d_row = d_job + 256, *d_out = d_row + 256, *d_coeff = d_out + 256;
    leaf_check(ctx, hipMemcpyAsync(d_src, residual, (((size_t)H - 1) * residual_stride + W) * 2, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemsetAsync(d_zero, 0, zb, ctx->stream), "hipMemsetAsync");
    SvtHipTxJob job;
    memset(&job, 0, sizeof(job));
    job.tx_type = (uint8_t)tx_type; job.pf_shape = (uint8_t)pf_shape;
    SvtHipQuantRow row;
    memset(&row, 0, sizeof(row));
    for (int k = 0; k < 2; k++) { row.zbin[k] = 32767; row.quant[k] = 1; row.quant_shift[k] = 1; row.dequant[k] = 1; } // quantizer output unused
    leaf_check(ctx, hipMemcpyAsync(d_job, &job, sizeof(job), hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(d_row, &row, sizeof(row), hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    SvtHipRdBatchDesc d;
    memset(&d, 0, sizeof(d));
    d.bit_depth = 10; d.quant_kind = 0; d.tx_size = (uint8_t)tx_size; d.n_jobs = 1; d.src_stride = residual_stride; d.pred_stride = (uint32_t)W;
    d.src = d_src; d.pred = d_zero; d.jobs = reinterpret_cast<const SvtHipTxJob *>(d_job); d.quant_rows = reinterpret_cast<const SvtHipQuantRow *>(d_row); d.n_quant_rows = 1;
    d.eob = reinterpret_cast<uint16_t *>(d_out); d.satd = reinterpret_cast<uint32_t *>(d_out + 8); d.dist_coeff = reinterpret_cast<uint64_t *>(d_out + 16);
    d.three_quad_energy = reinterpret_cast<uint64_t *>(d_out + 32); d.sse = reinterpret_cast<uint64_t *>(d_out + 40); d.coeff = reinterpret_cast<int32_t *>(d_coeff);
    if (svt_hip_rd_batch(ctx, &d) != SVT_HIP_OK) leaf_fail("%s", svt_hip_err_buf());
    leaf_check(ctx, hipMemcpyAsync(coeff, d_coeff, (size_t)NP * 4, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    uint64_t tq = 0;
    leaf_check(ctx, hipMemcpyAsync(&tq, d_out + 32, 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    if (three_quad_energy) *three_quad_energy = tq;
    return SVT_HIP_OK;
    } catch (const LeafFailure &f) { return svt_hip_fail(nullptr, SVT_HIP_ERR_LAUNCH, "svt_hip_estimate_transform: %s", f.what); }
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------
// The 8x8-based SAD pyramid of the integer search as pointer-level entries (aom_dsp_rtcd.h:842-855; bodies
// Codec/motion_estimation.c:98-425).  In production these live inside svt_hip_me_b64_kernel; here one small launch per call.
// ---------------------------------------------------------------------------------------------------------
namespace {

struct ExtSadParams {
    const uint8_t *src, *ref;
    uint32_t       src_stride, ref_stride, mv;
    int            n16, npos, sub_sad;
    uint32_t      *best8, *best16, *mv8, *mv16, *sad16, *sad8;
};

__device__ __forceinline__ uint32_t mv_plus_x(uint32_t mv, int k) {
    const int16_t x = (int16_t)((int16_t)(mv & 0xFFFF) + (int16_t)k);
    return (mv & 0xFFFF0000u) | (uint16_t)x;
}

// lane = 8x8 block in the order of the best arrays: (16x16 in the reference's PU order) * 4 + quadrant (motion_estimation.c:341)
__global__ void __launch_bounds__(64) ext_sad_8x8_16x16_kernel(const ExtSadParams p) {
    const int  lane = threadIdx.x, z16 = lane >> 2, q = lane & 3;
    const bool live = z16 < p.n16;
    const int  z2r[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15}; // raster <-> PU order of the 16x16s (its own inverse)
    const int  b   = p.n16 == 16 ? z2r[z16 & 15] : 0;
    const int  row = (b >> 2) * 16 + (q >> 1) * 8, col = (b & 3) * 16 + (q & 1) * 8;
    uint32_t best = live ? p.best8[lane] : 0, bmv = live ? p.mv8[lane] : 0;
    uint32_t b16 = (live && q == 0) ? p.best16[z16] : 0, m16 = (live && q == 0) ? p.mv16[z16] : 0;
    for (int k = 0; k < p.npos; k++) {
        uint32_t v = 0;
        if (live) {
            const int step = p.sub_sad ? 2 : 1; // svt_aom_compute8x4_sad_kernel_c on every other row, doubled (:42-91,105-136)
            for (int r = 0; r < 8; r += step)
                for (int c = 0; c < 8; c++) {
                    const int d = (int)p.src[(size_t)(row + r) * p.src_stride + col + c] - (int)p.ref[(size_t)(row + r) * p.ref_stride + col + c + k];
                    v += (uint32_t)(d < 0 ? -d : d);
                }
            if (p.sub_sad) v <<= 1;
            if (v < best) { best = v; bmv = mv_plus_x(p.mv, k); }
            if (p.sad8 && p.npos == 1) p.sad8[lane] = v;
        }
        uint32_t total = v + __shfl_xor(v, 1, 64);
        total += __shfl_xor(total, 2, 64);
        if (live && q == 0) {
            p.sad16[z16 * p.npos + k] = total;
            if (total < b16) { b16 = total; m16 = mv_plus_x(p.mv, k); }
        }
    }
    if (live) { p.best8[lane] = best; p.mv8[lane] = bmv; }
    if (live && q == 0) { p.best16[z16] = b16; p.mv16[z16] = m16; }
}

// svt_ext_{eight_,}sad_calculation_32x32_64x64 (:171-205,369-425): sums of four 16x16 SADs per 32x32, of four 32x32 per 64x64
__global__ void ext_sad_32x32_64x64_kernel(const uint32_t *sad16, int npos, uint32_t mv, uint32_t *best32, uint32_t *best64, uint32_t *mv32, uint32_t *mv64,
                                           uint32_t *sad32) {
    if (threadIdx.x != 0) return;
    for (int k = 0; k < npos; k++) {
        uint32_t total = 0;
        for (int q = 0; q < 4; q++) {
            const uint32_t s = sad16[(4 * q) * npos + k] + sad16[(4 * q + 1) * npos + k] + sad16[(4 * q + 2) * npos + k] + sad16[(4 * q + 3) * npos + k];
            sad32[q * npos + k] = s;
            if (s < best32[q]) { best32[q] = s; mv32[q] = mv_plus_x(mv, k); }
            total += s;
        }
        if (total < best64[0]) { best64[0] = total; mv64[0] = mv_plus_x(mv, k); }
    }
}

__global__ void fill_u32_kernel(uint32_t *p, uint32_t n, uint32_t v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

void leaf_ext_8x8_16x16(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride, uint32_t mv, uint32_t *best8, uint32_t *best16,
                        uint32_t *mv8, uint32_t *mv16, uint32_t *sad16, uint32_t *sad8, bool sub_sad, int n16, int npos) {
    std::lock_guard<std::mutex> lock(g_leaf_mutex);
    SvtHipContext *ctx = leaf_ctx();
    hipSetDevice(ctx->device);
    const int    side = n16 == 16 ? 64 : 16, n8 = n16 * 4;
    const size_t sbytes = ((size_t)side - 1) * src_stride + side, rbytes = ((size_t)side - 1) * ref_stride + side + npos - 1;
    const size_t sb = align256(sbytes), rb = align256(rbytes);
    uint8_t *base = leaf_scratch(ctx, sb + rb + 4096);
    uint32_t *d_u = reinterpret_cast<uint32_t *>(base + sb + rb); // best8[64] best16[16] mv8[64] mv16[16] sad16[128] sad8[64]
    leaf_check(ctx, hipMemcpyAsync(base, src, sbytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(base + sb, ref, rbytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(d_u, best8, n8 * 4, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(d_u + 64, best16, n16 * 4, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(d_u + 80, mv8, n8 * 4, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(d_u + 144, mv16, n16 * 4, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    ExtSadParams p;
    p.src = base; p.ref = base + sb; p.src_stride = src_stride; p.ref_stride = ref_stride; p.mv = mv; p.n16 = n16; p.npos = npos; p.sub_sad = sub_sad ? 1 : 0;
    p.best8 = d_u; p.best16 = d_u + 64; p.mv8 = d_u + 80; p.mv16 = d_u + 144; p.sad16 = d_u + 160; p.sad8 = sad8 ? d_u + 288 : nullptr;
    hipLaunchKernelGGL(ext_sad_8x8_16x16_kernel, dim3(1), dim3(64), 0, ctx->stream, p);
    leaf_check(ctx, hipGetLastError(), "ext_sad_8x8_16x16_kernel launch");
    leaf_check(ctx, hipMemcpyAsync(best8, d_u, n8 * 4, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(best16, d_u + 64, n16 * 4, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(mv8, d_u + 80, n8 * 4, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(mv16, d_u + 144, n16 * 4, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(sad16, d_u + 160, (size_t)n16 * npos * 4, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    if (sad8) leaf_check(ctx, hipMemcpyAsync(sad8, d_u + 288, n8 * 4, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}

void leaf_ext_32x32_64x64(const uint32_t *sad16, int npos, uint32_t mv, uint32_t *best32, uint32_t *best64, uint32_t *mv32, uint32_t *mv64, uint32_t *sad32) {
    std::lock_guard<std::mutex> lock(g_leaf_mutex);
    SvtHipContext *ctx = leaf_ctx();
    hipSetDevice(ctx->device);
    uint32_t *d_u = reinterpret_cast<uint32_t *>(leaf_scratch(ctx, 2048)); // sad16[128] best32[4] best64[1] mv32[4] mv64[1] sad32[32]
    leaf_check(ctx, hipMemcpyAsync(d_u, sad16, (size_t)16 * npos * 4, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(d_u + 128, best32, 16, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(d_u + 132, best64, 4, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(d_u + 136, mv32, 16, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(d_u + 140, mv64, 4, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    hipLaunchKernelGGL(ext_sad_32x32_64x64_kernel, dim3(1), dim3(64), 0, ctx->stream, d_u, npos, mv, d_u + 128, d_u + 132, d_u + 136, d_u + 140, d_u + 144);
    leaf_check(ctx, hipGetLastError(), "ext_sad_32x32_64x64_kernel launch");
    leaf_check(ctx, hipMemcpyAsync(best32, d_u + 128, 16, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(best64, d_u + 132, 4, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(mv32, d_u + 136, 16, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(mv64, d_u + 140, 4, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(sad32, d_u + 144, (size_t)4 * npos * 4, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}

} // namespace

extern "C" {

void svt_ext_all_sad_calculation_8x8_16x16_hip(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t mv, uint32_t *p_best_sad_8x8,
                                               uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8, uint32_t *p_best_mv16x16, uint32_t p_eight_sad16x16[16][8],
                                               uint32_t p_eight_sad8x8[64][8], bool sub_sad) LEAF_TRY
    (void)p_eight_sad8x8; // left untouched, like the C body (motion_estimation.c:335-362 never stores to it)
    leaf_ext_8x8_16x16(src, src_stride, ref, ref_stride, mv, p_best_sad_8x8, p_best_sad_16x16, p_best_mv8x8, p_best_mv16x16, &p_eight_sad16x16[0][0], nullptr,
                       sub_sad, 16, 8);
LEAF_CATCH(svt_ext_all_sad_calculation_8x8_16x16_hip, src, src_stride, ref, ref_stride, mv, p_best_sad_8x8, p_best_sad_16x16, p_best_mv8x8, p_best_mv16x16, p_eight_sad16x16, p_eight_sad8x8, sub_sad)

void svt_ext_sad_calculation_8x8_16x16_hip(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t *p_best_sad_8x8,
                                           uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8, uint32_t *p_best_mv16x16, uint32_t mv, uint32_t *p_sad16x16,
                                           uint32_t *p_sad8x8, bool sub_sad) LEAF_TRY
    leaf_ext_8x8_16x16(src, src_stride, ref, ref_stride, mv, p_best_sad_8x8, p_best_sad_16x16, p_best_mv8x8, p_best_mv16x16, p_sad16x16, p_sad8x8, sub_sad, 1, 1);
LEAF_CATCH(svt_ext_sad_calculation_8x8_16x16_hip, src, src_stride, ref, ref_stride, p_best_sad_8x8, p_best_sad_16x16, p_best_mv8x8, p_best_mv16x16, mv, p_sad16x16, p_sad8x8, sub_sad)

void svt_ext_eight_sad_calculation_32x32_64x64_hip(uint32_t p_sad16x16[16][8], uint32_t *p_best_sad_32x32, uint32_t *p_best_sad_64x64, uint32_t *p_best_mv32x32,
                                                   uint32_t *p_best_mv64x64, uint32_t mv, uint32_t p_sad32x32[4][8]) LEAF_TRY
    leaf_ext_32x32_64x64(&p_sad16x16[0][0], 8, mv, p_best_sad_32x32, p_best_sad_64x64, p_best_mv32x32, p_best_mv64x64, &p_sad32x32[0][0]);
LEAF_CATCH(svt_ext_eight_sad_calculation_32x32_64x64_hip, p_sad16x16, p_best_sad_32x32, p_best_sad_64x64, p_best_mv32x32, p_best_mv64x64, mv, p_sad32x32)

void svt_ext_sad_calculation_32x32_64x64_hip(uint32_t *p_sad16x16, uint32_t *p_best_sad_32x32, uint32_t *p_best_sad_64x64, uint32_t *p_best_mv32x32,
                                             uint32_t *p_best_mv64x64, uint32_t mv, uint32_t *p_sad32x32) LEAF_TRY
    leaf_ext_32x32_64x64(p_sad16x16, 1, mv, p_best_sad_32x32, p_best_sad_64x64, p_best_mv32x32, p_best_mv64x64, p_sad32x32);
LEAF_CATCH(svt_ext_sad_calculation_32x32_64x64_hip, p_sad16x16, p_best_sad_32x32, p_best_sad_64x64, p_best_mv32x32, p_best_mv64x64, mv, p_sad32x32)

void svt_initialize_buffer_32bits_hip(uint32_t *pointer, uint32_t count128, uint32_t count32, uint32_t value) LEAF_TRY
    const uint32_t n = count128 * 4 + count32; // me_sad_calculation.c:14-17
    if (!n) return;
    std::lock_guard<std::mutex> lock(g_leaf_mutex);
    SvtHipContext *ctx = leaf_ctx();
    hipSetDevice(ctx->device);
    uint32_t *d = reinterpret_cast<uint32_t *>(leaf_scratch(ctx, (size_t)n * 4));
    hipLaunchKernelGGL(fill_u32_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d, n, value);
    leaf_check(ctx, hipGetLastError(), "fill_u32_kernel launch");
    leaf_check(ctx, hipMemcpyAsync(pointer, d, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
LEAF_CATCH(svt_initialize_buffer_32bits_hip, pointer, count128, count32, value)

} // extern "C"

// ---------------------------------------------------------------------------------------------------------
// Quantizers as pointer-level entries (aom_dsp_rtcd.h:244-263; bodies Codec/full_loop.c:29-79,149-198 ("b"), :282-474 ("fp")).
// In production they are a stage of rd_tx_kernel; here one launch over a caller-supplied coefficient array.
// ---------------------------------------------------------------------------------------------------------
namespace {

struct QuantLeafParams {
    const int32_t *coeff;
    int32_t       *qcoeff, *dqcoeff;
    uint32_t      *eob;
    const int16_t *iscan;
    const uint8_t *qm, *iqm; // null = flat (1 << AOM_QM_BITS)
    int            n, log_scale, hbd, fp;
    int16_t        zbin[2], round[2], quant[2], quant_shift[2], dequant[2]; // [0] = DC, [1] = AC; round / quant are the fp rows when fp
};

__global__ void __launch_bounds__(256) quantize_leaf_kernel(const QuantLeafParams p) {
    __shared__ uint32_t s_eob;
    if (threadIdx.x == 0) s_eob = 0;
    __syncthreads();
    const int ls = p.log_scale;
    uint32_t  eob = 0;
    for (int rc = threadIdx.x; rc < p.n; rc += 256) {
        const int     ac = rc != 0;
        const int32_t co = p.coeff[rc], sign = co < 0 ? -1 : 0, a = (co ^ sign) - sign;
        const int32_t wt = p.qm ? p.qm[rc] : 32, iwt = p.iqm ? p.iqm[rc] : 32; // AOM_QM_BITS = 5
        const int32_t rnd = ls ? ((p.round[ac] + (1 << (ls - 1))) >> ls) : p.round[ac];
        int32_t qv = 0, dq = 0;
        if (!p.fp) { // svt_aom_quantize_b_c_ii / svt_aom_highbd_quantize_b_c
            const int32_t zb = ls ? ((p.zbin[ac] + (1 << (ls - 1))) >> ls) : p.zbin[ac];
            if ((i64)a * wt >= ((i64)zb << 5)) {
                i64 t = (i64)a + rnd;
                if (!p.hbd) t = t < -32768 ? -32768 : (t > 32767 ? 32767 : t);
                t *= wt;
                qv = (int32_t)(((((t * p.quant[ac]) >> 16) + t) * p.quant_shift[ac]) >> (16 - ls + 5));
                dq = (qv * (((int32_t)p.dequant[ac] * iwt + 16) >> 5)) >> ls;
            }
        } else if (!p.qm && !p.iqm) { // quantize_fp_helper_c / highbd_quantize_fp_helper_c, flat
            const bool keep = p.hbd ? ((a << (1 + ls)) >= p.dequant[ac]) : (((i64)a << (1 + ls)) >= (int32_t)p.dequant[ac]);
            if (keep) {
                i64 t = (i64)a + rnd;
                if (!p.hbd) t = t < -32768 ? -32768 : (t > 32767 ? 32767 : t);
                qv = (int32_t)((t * p.quant[ac]) >> (16 - ls));
                dq = (qv * (int32_t)p.dequant[ac]) >> ls;
            }
        } else if ((i64)a * wt >= ((int32_t)p.dequant[ac] << (5 - (1 + ls)))) { // the helpers' matrix branch
            i64 t = (i64)a + rnd;
            if (!p.hbd) t = t < -32768 ? -32768 : (t > 32767 ? 32767 : t);
            qv = (int32_t)((t * p.quant[ac] * wt) >> (16 - ls + 5));
            dq = (qv * (((int32_t)p.dequant[ac] * iwt + 16) >> 5)) >> ls;
        }
        p.qcoeff[rc]  = (qv ^ sign) - sign;
        p.dqcoeff[rc] = (dq ^ sign) - sign;
        if (qv) { const uint32_t e = (uint32_t)p.iscan[rc] + 1; eob = e > eob ? e : eob; }
    }
    atomicMax(&s_eob, eob);
    __syncthreads();
    if (threadIdx.x == 0) *p.eob = s_eob;
}

void leaf_quantize(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round, const int16_t *quant, const int16_t *quant_shift, int32_t *qcoeff,
                   int32_t *dqcoeff, const int16_t *dequant, uint16_t *eob, const int16_t *iscan, const uint8_t *qm, const uint8_t *iqm, int log_scale, int hbd,
                   int fp) {
    std::lock_guard<std::mutex> lock(g_leaf_mutex);
    SvtHipContext *ctx = leaf_ctx();
    hipSetDevice(ctx->device);
    *eob = 0;
    if (n <= 0) return;
    const size_t cb = align256((size_t)n * 4), ib = align256((size_t)n * 2), mb = align256((size_t)n);
    uint8_t *base = leaf_scratch(ctx, 3 * cb + ib + 2 * mb + 256);
    QuantLeafParams p;
    memset(&p, 0, sizeof(p));
    p.coeff = reinterpret_cast<int32_t *>(base); p.qcoeff = reinterpret_cast<int32_t *>(base + cb); p.dqcoeff = reinterpret_cast<int32_t *>(base + 2 * cb);
    p.iscan = reinterpret_cast<int16_t *>(base + 3 * cb);
    uint8_t *d_qm = base + 3 * cb + ib, *d_iqm = d_qm + mb;
    p.eob = reinterpret_cast<uint32_t *>(d_iqm + mb);
    leaf_check(ctx, hipMemcpyAsync(base, coeff, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(base + 3 * cb, iscan, (size_t)n * 2, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    if (qm) { leaf_check(ctx, hipMemcpyAsync(d_qm, qm, (size_t)n, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync"); p.qm = d_qm; }
    if (iqm) { leaf_check(ctx, hipMemcpyAsync(d_iqm, iqm, (size_t)n, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync"); p.iqm = d_iqm; }
    p.n = (int)n; p.log_scale = log_scale; p.hbd = hbd; p.fp = fp;
    for (int k = 0; k < 2; k++) { // MacroblockPlane rows: [0] = DC, [1..7] = AC
        p.zbin[k] = zbin ? zbin[k] : 0; p.round[k] = round[k]; p.quant[k] = quant[k]; p.quant_shift[k] = quant_shift ? quant_shift[k] : 0; p.dequant[k] = dequant[k];
    }
    hipLaunchKernelGGL(quantize_leaf_kernel, dim3(1), dim3(256), 0, ctx->stream, p);
    leaf_check(ctx, hipGetLastError(), "quantize_leaf_kernel launch");
    uint32_t e = 0;
    leaf_check(ctx, hipMemcpyAsync(qcoeff, p.qcoeff, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(dqcoeff, p.dqcoeff, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(&e, p.eob, 4, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    *eob = (uint16_t)e;
}

} // namespace

extern "C" {

#define SVT_HIP_QUANT_B(NAME, HBD)                                                                                                                        \
    void NAME(const int32_t *coeff_ptr, intptr_t n_coeffs, const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr,                 \
              const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr, const int16_t *dequant_ptr, uint16_t *eob_ptr, const int16_t *scan, \
              const int16_t *iscan, const uint8_t *qm_ptr, const uint8_t *iqm_ptr, const int32_t log_scale) LEAF_TRY                                     \
        (void)scan;                                                                                                                                      \
        leaf_quantize(coeff_ptr, n_coeffs, zbin_ptr, round_ptr, quant_ptr, quant_shift_ptr, qcoeff_ptr, dqcoeff_ptr, dequant_ptr, eob_ptr, iscan, qm_ptr, iqm_ptr, \
                      log_scale, HBD, 0);                                                                                                                \
    LEAF_CATCH(NAME, coeff_ptr, n_coeffs, zbin_ptr, round_ptr, quant_ptr, quant_shift_ptr, qcoeff_ptr, dqcoeff_ptr, dequant_ptr, eob_ptr, scan, iscan, qm_ptr, iqm_ptr, log_scale)
SVT_HIP_QUANT_B(svt_aom_quantize_b_hip, 0)
SVT_HIP_QUANT_B(svt_aom_highbd_quantize_b_hip, 1)
SVT_HIP_QUANT_B(svt_av1_quantize_b_qm_hip, 0)
SVT_HIP_QUANT_B(svt_av1_highbd_quantize_b_qm_hip, 1)
#undef SVT_HIP_QUANT_B

#define SVT_HIP_QUANT_FP(NAME, LS)                                                                                                                       \
    void NAME(const int32_t *coeff_ptr, intptr_t n_coeffs, const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr,                 \
              const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr, const int16_t *dequant_ptr, uint16_t *eob_ptr, const int16_t *scan, \
              const int16_t *iscan) LEAF_TRY                                                                                                             \
        (void)scan;                                                                                                                                      \
        leaf_quantize(coeff_ptr, n_coeffs, zbin_ptr, round_ptr, quant_ptr, quant_shift_ptr, qcoeff_ptr, dqcoeff_ptr, dequant_ptr, eob_ptr, iscan, nullptr, nullptr, \
                      LS, 0, 1);                                                                                                                         \
    LEAF_CATCH(NAME, coeff_ptr, n_coeffs, zbin_ptr, round_ptr, quant_ptr, quant_shift_ptr, qcoeff_ptr, dqcoeff_ptr, dequant_ptr, eob_ptr, scan, iscan)
SVT_HIP_QUANT_FP(svt_av1_quantize_fp_hip, 0)
SVT_HIP_QUANT_FP(svt_av1_quantize_fp_32x32_hip, 1)
SVT_HIP_QUANT_FP(svt_av1_quantize_fp_64x64_hip, 2)
#undef SVT_HIP_QUANT_FP

void svt_av1_quantize_fp_qm_hip(const int32_t *coeff_ptr, intptr_t n_coeffs, const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr,
                                const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr, const int16_t *dequant_ptr, uint16_t *eob_ptr,
                                const int16_t *scan, const int16_t *iscan, const uint8_t *qm_ptr, const uint8_t *iqm_ptr, int16_t log_scale) LEAF_TRY
    (void)scan;
    leaf_quantize(coeff_ptr, n_coeffs, zbin_ptr, round_ptr, quant_ptr, quant_shift_ptr, qcoeff_ptr, dqcoeff_ptr, dequant_ptr, eob_ptr, iscan, qm_ptr, iqm_ptr, log_scale, 0, 1);
LEAF_CATCH(svt_av1_quantize_fp_qm_hip, coeff_ptr, n_coeffs, zbin_ptr, round_ptr, quant_ptr, quant_shift_ptr, qcoeff_ptr, dqcoeff_ptr, dequant_ptr, eob_ptr, scan, iscan, qm_ptr, iqm_ptr, log_scale)
void svt_av1_highbd_quantize_fp_hip(const int32_t *coeff_ptr, intptr_t n_coeffs, const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr,
                                    const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr, const int16_t *dequant_ptr, uint16_t *eob_ptr,
                                    const int16_t *scan, const int16_t *iscan, int16_t log_scale) LEAF_TRY
    (void)scan;
    leaf_quantize(coeff_ptr, n_coeffs, zbin_ptr, round_ptr, quant_ptr, quant_shift_ptr, qcoeff_ptr, dqcoeff_ptr, dequant_ptr, eob_ptr, iscan, nullptr, nullptr, log_scale, 1, 1);
LEAF_CATCH(svt_av1_highbd_quantize_fp_hip, coeff_ptr, n_coeffs, zbin_ptr, round_ptr, quant_ptr, quant_shift_ptr, qcoeff_ptr, dqcoeff_ptr, dequant_ptr, eob_ptr, scan, iscan, log_scale)
void svt_av1_highbd_quantize_fp_qm_hip(const int32_t *coeff_ptr, intptr_t n_coeffs, const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr,
                                       const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr, const int16_t *dequant_ptr, uint16_t *eob_ptr,
                                       const int16_t *scan, const int16_t *iscan, const uint8_t *qm_ptr, const uint8_t *iqm_ptr, int16_t log_scale) LEAF_TRY
    (void)scan;
    leaf_quantize(coeff_ptr, n_coeffs, zbin_ptr, round_ptr, quant_ptr, quant_shift_ptr, qcoeff_ptr, dqcoeff_ptr, dequant_ptr, eob_ptr, iscan, qm_ptr, iqm_ptr, log_scale, 1, 1);
LEAF_CATCH(svt_av1_highbd_quantize_fp_qm_hip, coeff_ptr, n_coeffs, zbin_ptr, round_ptr, quant_ptr, quant_shift_ptr, qcoeff_ptr, dqcoeff_ptr, dequant_ptr, eob_ptr, scan, iscan, qm_ptr, iqm_ptr, log_scale)

} // extern "C"

// ---------------------------------------------------------------------------------------------------------
// svt_av1_inv_txfm2d_add_{W}x{H} (common_dsp_rtcd.h:100-141; bodies Codec/inv_transforms.c:2459-2716) as pointer-level
// entries: uint16 planes for either bit depth, separate read / write pointers, packed coefficients for the 64-point sizes.
// ---------------------------------------------------------------------------------------------------------
namespace {
void leaf_inv_txfm(const int32_t *input, const uint16_t *out_r, int32_t stride_r, uint16_t *out_w, int32_t stride_w, int tx_type, int tx_size, int bd) {
    std::lock_guard<std::mutex> lock(g_leaf_mutex);
    SvtHipContext *ctx = leaf_ctx();
    hipSetDevice(ctx->device);
    if ((bd != 8 && bd != 10) || tx_size < 0 || tx_size >= SVT_HIP_TX_SIZES_ALL || tx_type < 0 || tx_type >= SVT_HIP_TX_TYPES || stride_r <= 0 || stride_w <= 0) {
        leaf_fail("svt_av1_inv_txfm2d_add_hip: unsupported bd %d / tx_size %d / tx_type %d", bd, tx_size, tx_type);
    }
    const int W = svt_hip_tx_size_wide(tx_size), H = svt_hip_tx_size_high(tx_size), NP = (W > 32 ? 32 : W) * (H > 32 ? 32 : H);
    const size_t cb = align256((size_t)NP * 4), rbytes = (((size_t)H - 1) * stride_r + W) * 2, wbytes = (size_t)W * H * 2;
    uint8_t *base = leaf_scratch(ctx, cb + align256(rbytes) + align256(wbytes) + 256);
    uint8_t *d_co = base, *d_pred = d_co + cb, *d_rec = d_pred + align256(rbytes), *d_job = d_rec + align256(wbytes);
    leaf_check(ctx, hipMemcpyAsync(d_co, input, (size_t)NP * 4, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(d_pred, out_r, rbytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    SvtHipTxJob job;
    memset(&job, 0, sizeof(job));
    job.tx_type = (uint8_t)tx_type;
    leaf_check(ctx, hipMemcpyAsync(d_job, &job, sizeof(job), hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    SvtHipInvTxBatchDesc d;
    memset(&d, 0, sizeof(d));
    d.bit_depth = (uint8_t)bd; d.sample_bytes = 2; d.tx_size = (uint8_t)tx_size; d.n_jobs = 1; d.pred_stride = (uint32_t)stride_r; d.recon_stride = (uint32_t)W;
    d.pred = d_pred; d.recon = d_rec; d.jobs = reinterpret_cast<const SvtHipTxJob *>(d_job); d.dqcoeff = reinterpret_cast<const int32_t *>(d_co);
    if (svt_hip_inv_txfm_batch(ctx, &d) != SVT_HIP_OK) leaf_fail("%s", svt_hip_err_buf());
    leaf_check(ctx, hipMemcpy2DAsync(out_w, (size_t)stride_w * 2, d_rec, (size_t)W * 2, (size_t)W * 2, H, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy2DAsync");
    leaf_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}
} // namespace

extern "C" {
// TxType and TxSize are one-byte (ATTRIBUTE_PACKED) enums in the reference (definitions.h): uint8_t is the same ABI
#define SVT_HIP_INV_SQ(W, H, TS)                                                                                                                       \
    void svt_av1_inv_txfm2d_add_##W##x##H##_hip(const int32_t *input, uint16_t *output_r, int32_t stride_r, uint16_t *output_w, int32_t stride_w, uint8_t tx_type, \
                                                int32_t bd) LEAF_TRY                                                                                  \
        leaf_inv_txfm(input, output_r, stride_r, output_w, stride_w, tx_type, TS, bd);                                                                \
    LEAF_CATCH(svt_av1_inv_txfm2d_add_##W##x##H##_hip, input, output_r, stride_r, output_w, stride_w, tx_type, bd)
SVT_HIP_INV_SQ(4, 4, 0) SVT_HIP_INV_SQ(8, 8, 1) SVT_HIP_INV_SQ(16, 16, 2) SVT_HIP_INV_SQ(32, 32, 3) SVT_HIP_INV_SQ(64, 64, 4)
#undef SVT_HIP_INV_SQ
#define SVT_HIP_INV_RECT(W, H, TS)                                                                                                                     \
    void svt_av1_inv_txfm2d_add_##W##x##H##_hip(const int32_t *input, uint16_t *output_r, int32_t stride_r, uint16_t *output_w, int32_t stride_w, uint8_t tx_type, \
                                                uint8_t tx_size, int32_t eob, int32_t bd) LEAF_TRY                                                    \
        /* eob only lets the reference skip zero rows; the result does not depend on it */                                                            \
        leaf_inv_txfm(input, output_r, stride_r, output_w, stride_w, tx_type, TS, bd);                                                                \
    LEAF_CATCH(svt_av1_inv_txfm2d_add_##W##x##H##_hip, input, output_r, stride_r, output_w, stride_w, tx_type, tx_size, eob, bd)
SVT_HIP_INV_RECT(8, 16, 7) SVT_HIP_INV_RECT(16, 8, 8) SVT_HIP_INV_RECT(16, 32, 9) SVT_HIP_INV_RECT(32, 16, 10) SVT_HIP_INV_RECT(32, 64, 11) SVT_HIP_INV_RECT(64, 32, 12)
SVT_HIP_INV_RECT(8, 32, 15) SVT_HIP_INV_RECT(32, 8, 16) SVT_HIP_INV_RECT(16, 64, 17) SVT_HIP_INV_RECT(64, 16, 18)
#undef SVT_HIP_INV_RECT
#define SVT_HIP_INV_SMALL(W, H, TS)                                                                                                                    \
    void svt_av1_inv_txfm2d_add_##W##x##H##_hip(const int32_t *input, uint16_t *output_r, int32_t stride_r, uint16_t *output_w, int32_t stride_w, uint8_t tx_type, \
                                                uint8_t tx_size, int32_t bd) LEAF_TRY                                                                 \
        leaf_inv_txfm(input, output_r, stride_r, output_w, stride_w, tx_type, TS, bd);                                                                \
    LEAF_CATCH(svt_av1_inv_txfm2d_add_##W##x##H##_hip, input, output_r, stride_r, output_w, stride_w, tx_type, tx_size, bd)
SVT_HIP_INV_SMALL(4, 8, 5) SVT_HIP_INV_SMALL(8, 4, 6) SVT_HIP_INV_SMALL(4, 16, 13) SVT_HIP_INV_SMALL(16, 4, 14)
#undef SVT_HIP_INV_SMALL
} // extern "C"

// ---------------------------------------------------------------------------------------------------------
// svt_av1_fwd_txfm2d_{W}x{H}{,_N2,_N4} (aom_dsp_rtcd.h / aom_dsp_rtcd.c:421-487; bodies Codec/transforms.c) as pointer-level
// entries: the full W x H coefficient array, like the reference's per-size pointers (bd only selects stage ranges there).
// ---------------------------------------------------------------------------------------------------------
namespace {
void leaf_fwd_txfm(const int16_t *input, int32_t *output, uint32_t stride, int tx_type, int tx_size, int pf_shape) {
    std::lock_guard<std::mutex> lock(g_leaf_mutex);
    SvtHipContext *ctx = leaf_ctx();
    hipSetDevice(ctx->device);
    if (tx_type < 0 || tx_type >= SVT_HIP_TX_TYPES) leaf_fail("svt_av1_fwd_txfm2d_hip: tx_type %d", tx_type);
    const int W = svt_hip_tx_size_wide(tx_size), H = svt_hip_tx_size_high(tx_size);
    const size_t rbytes = (((size_t)H - 1) * stride + W) * 2, obytes = (size_t)W * H * 4;
    uint8_t *base = leaf_scratch(ctx, align256(rbytes) + align256(obytes) + 256);
    uint8_t *d_res = base, *d_out = d_res + align256(rbytes), *d_job = d_out + align256(obytes);
    leaf_check(ctx, hipMemcpyAsync(d_res, input, rbytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    SvtHipTxJob job;
    memset(&job, 0, sizeof(job));
    job.tx_type = (uint8_t)tx_type; job.pf_shape = (uint8_t)pf_shape;
    leaf_check(ctx, hipMemcpyAsync(d_job, &job, sizeof(job), hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    SvtHipFwdTxBatchDesc d;
    memset(&d, 0, sizeof(d));
    d.tx_size = (uint8_t)tx_size; d.n_jobs = 1; d.residual_stride = stride; d.residual = reinterpret_cast<const int16_t *>(d_res);
    d.jobs = reinterpret_cast<const SvtHipTxJob *>(d_job); d.coeff = reinterpret_cast<int32_t *>(d_out);
    if (svt_hip_fwd_txfm_batch(ctx, &d) != SVT_HIP_OK) leaf_fail("%s", svt_hip_err_buf());
    leaf_check(ctx, hipMemcpyAsync(output, d_out, obytes, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}
} // namespace

extern "C" {
#define SVT_HIP_FWD(W, H, TS)                                                                                                                   \
    void svt_av1_fwd_txfm2d_##W##x##H##_hip(int16_t *input, int32_t *output, uint32_t stride, uint8_t tx_type, uint8_t bd) LEAF_TRY                   \
        leaf_fwd_txfm(input, output, stride, tx_type, TS, 0);                                                                         \
    LEAF_CATCH(svt_av1_fwd_txfm2d_##W##x##H##_hip, input, output, stride, tx_type, bd)                                                                                                                                           \
    void svt_av1_fwd_txfm2d_##W##x##H##_N2_hip(int16_t *input, int32_t *output, uint32_t stride, uint8_t tx_type, uint8_t bd) LEAF_TRY                \
        leaf_fwd_txfm(input, output, stride, tx_type, TS, 1);                                                                         \
    LEAF_CATCH(svt_av1_fwd_txfm2d_##W##x##H##_N2_hip, input, output, stride, tx_type, bd)                                                                                                                                           \
    void svt_av1_fwd_txfm2d_##W##x##H##_N4_hip(int16_t *input, int32_t *output, uint32_t stride, uint8_t tx_type, uint8_t bd) LEAF_TRY                \
        leaf_fwd_txfm(input, output, stride, tx_type, TS, 2);                                                                         \
    LEAF_CATCH(svt_av1_fwd_txfm2d_##W##x##H##_N4_hip, input, output, stride, tx_type, bd)
SVT_HIP_FWD(4, 4, 0) SVT_HIP_FWD(8, 8, 1) SVT_HIP_FWD(16, 16, 2) SVT_HIP_FWD(32, 32, 3) SVT_HIP_FWD(64, 64, 4) SVT_HIP_FWD(4, 8, 5) SVT_HIP_FWD(8, 4, 6)
SVT_HIP_FWD(8, 16, 7) SVT_HIP_FWD(16, 8, 8) SVT_HIP_FWD(16, 32, 9) SVT_HIP_FWD(32, 16, 10) SVT_HIP_FWD(32, 64, 11) SVT_HIP_FWD(64, 32, 12) SVT_HIP_FWD(4, 16, 13)
SVT_HIP_FWD(16, 4, 14) SVT_HIP_FWD(8, 32, 15) SVT_HIP_FWD(32, 8, 16) SVT_HIP_FWD(16, 64, 17) SVT_HIP_FWD(64, 16, 18)
#undef SVT_HIP_FWD
} // extern "C"

// ---------------------------------------------------------------------------------------------------------
// svt_handle_transform{16x64,32x64,64x16,64x32,64x64}{,_N2_N4} (aom_dsp_rtcd.c:440-449; Codec/transforms.c:2374-2543): energy of
// the frequencies a 64-point size discards + in-place packing of the kept 32-wide rows, on a host coefficient array.
// ---------------------------------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) handle_transform_kernel(const int32_t *in, int w, int h, int with_energy, int32_t *packed, u64 *energy) {
    __shared__ u64 part[4];
    const int wp = w > 32 ? 32 : w, hp = h > 32 ? 32 : h;
    u64 e = 0;
    for (int i = threadIdx.x; i < w * h; i += 256) {
        const int r = i / w, c = i - r * w;
        const int32_t v = in[i];
        if (r < hp && c < wp) packed[r * wp + c] = v;
        else if (with_energy) e += (u64)((i64)v * v);
    }
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor((unsigned long long)e, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) *energy = part[0] + part[1] + part[2] + part[3];
}

uint64_t leaf_handle_transform(int32_t *output, int w, int h, int with_energy) {
    std::lock_guard<std::mutex> lock(g_leaf_mutex);
    SvtHipContext *ctx = leaf_ctx();
    hipSetDevice(ctx->device);
    const int    wp = w > 32 ? 32 : w, hp = h > 32 ? 32 : h;
    const size_t ib = align256((size_t)w * h * 4), pb = align256((size_t)wp * hp * 4);
    uint8_t *base = leaf_scratch(ctx, ib + pb + 256);
    leaf_check(ctx, hipMemcpyAsync(base, output, (size_t)w * h * 4, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    hipLaunchKernelGGL(handle_transform_kernel, dim3(1), dim3(256), 0, ctx->stream, reinterpret_cast<const int32_t *>(base), w, h, with_energy,
                       reinterpret_cast<int32_t *>(base + ib), reinterpret_cast<u64 *>(base + ib + pb));
    leaf_check(ctx, hipGetLastError(), "handle_transform_kernel launch");
    uint64_t e = 0;
    // only the 64-wide sizes are re-packed (rows of 64 -> rows of 32 at the front of the array; what lies behind keeps its content)
    if (w == 64) leaf_check(ctx, hipMemcpyAsync(output, base + ib, (size_t)wp * hp * 4, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(&e, base + ib + pb, 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    return e;
}
} // namespace

extern "C" {
uint64_t svt_handle_transform16x64_hip(int32_t *output) LEAF_TRY return leaf_handle_transform(output, 16, 64, 1); LEAF_CATCH(svt_handle_transform16x64_hip, output)
uint64_t svt_handle_transform32x64_hip(int32_t *output) LEAF_TRY return leaf_handle_transform(output, 32, 64, 1); LEAF_CATCH(svt_handle_transform32x64_hip, output)
uint64_t svt_handle_transform64x16_hip(int32_t *output) LEAF_TRY return leaf_handle_transform(output, 64, 16, 1); LEAF_CATCH(svt_handle_transform64x16_hip, output)
uint64_t svt_handle_transform64x32_hip(int32_t *output) LEAF_TRY return leaf_handle_transform(output, 64, 32, 1); LEAF_CATCH(svt_handle_transform64x32_hip, output)
uint64_t svt_handle_transform64x64_hip(int32_t *output) LEAF_TRY return leaf_handle_transform(output, 64, 64, 1); LEAF_CATCH(svt_handle_transform64x64_hip, output)
uint64_t svt_handle_transform16x64_N2_N4_hip(int32_t *output) LEAF_TRY (void)output; return 0; LEAF_CATCH(svt_handle_transform16x64_N2_N4_hip, output) // the reference's bodies are empty too (transforms.c:2514-2521)
uint64_t svt_handle_transform32x64_N2_N4_hip(int32_t *output) LEAF_TRY (void)output; return 0; LEAF_CATCH(svt_handle_transform32x64_N2_N4_hip, output)
uint64_t svt_handle_transform64x16_N2_N4_hip(int32_t *output) LEAF_TRY return leaf_handle_transform(output, 64, 16, 0); LEAF_CATCH(svt_handle_transform64x16_N2_N4_hip, output)
uint64_t svt_handle_transform64x32_N2_N4_hip(int32_t *output) LEAF_TRY return leaf_handle_transform(output, 64, 32, 0); LEAF_CATCH(svt_handle_transform64x32_N2_N4_hip, output)
uint64_t svt_handle_transform64x64_N2_N4_hip(int32_t *output) LEAF_TRY return leaf_handle_transform(output, 64, 64, 0); LEAF_CATCH(svt_handle_transform64x64_N2_N4_hip, output)
} // extern "C"
