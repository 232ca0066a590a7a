/* stats_oracle.c -- CPU restatement of the batched block statistics (svt_hip_block_stats_batch), composed from the leaf
 * restatements of dsp_oracle.c / me_oracle.c.  TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg; never linked into or called by the product path.
 *
 * Follows: svt_nxm_sad_kernel_helper_c / svt_aom_sad_16b_kernel_c (C_DEFAULT/compute_sad_c.c:20-56,209),
 * svt_spatial_full_distortion_kernel_c / svt_full_distortion_kernel16_bits_c (picture_operators_c.c:65-83,
 * Codec/pic_operators.c:174-197), svt_aom_variance{W}x{H}_c (C_DEFAULT/variance.c:256-296), hadamard_path_c
 * (Codec/enc_mode_config.c:2151-2217). */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include "../include/svt_hip_dsp.h"

uint32_t orc_nxm_sad(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride, uint32_t height, uint32_t width);
uint32_t orc_sad_16b(const uint16_t *src, uint32_t src_stride, const uint16_t *ref, uint32_t ref_stride, uint32_t height, uint32_t width);
uint64_t orc_spatial_sse8(const uint8_t *a, uint32_t a_off, uint32_t a_stride, const uint8_t *b, int32_t b_off, uint32_t b_stride, uint32_t w, uint32_t h);
uint64_t orc_spatial_sse16(const uint16_t *a, uint32_t a_off, uint32_t a_stride, const uint16_t *b, int32_t b_off, uint32_t b_stride, uint32_t w, uint32_t h);
uint32_t orc_variance8(const uint8_t *a, int a_stride, const uint8_t *b, int b_stride, int w, int h, uint32_t *sse);
uint32_t orc_variance16(const uint16_t *a, int a_stride, const uint16_t *b, int b_stride, int w, int h, uint32_t *sse);
void     orc_residual8(const uint8_t *in, uint32_t in_stride, const uint8_t *pred, uint32_t pred_stride, int16_t *res, uint32_t res_stride, uint32_t w, uint32_t h);
void     orc_hadamard_4x4(const int16_t *src, ptrdiff_t stride, int32_t *coeff);
void     orc_hadamard_8x8(const int16_t *src, ptrdiff_t stride, int32_t *coeff);
void     orc_hadamard_16x16(const int16_t *src, ptrdiff_t stride, int32_t *coeff);
void     orc_hadamard_32x32(const int16_t *src, ptrdiff_t stride, int32_t *coeff);
int      orc_satd(const int32_t *coeff, int n);

/* hadamard_path_c, enc_mode_config.c:2151-2217, with the Buf2D arguments flattened (8-bit input / prediction, square block) */
uint32_t orc_hadamard_path(const uint8_t *input, uint32_t input_stride, const uint8_t *pred, uint32_t pred_stride, uint32_t bsize_wide) {
    const uint32_t n = bsize_wide < 32 ? bsize_wide : 32;
    int16_t        res[32 * 32];
    int32_t        coeff[32 * 32];
    uint32_t       cost = 0;
    for (uint32_t row = 0; row < bsize_wide; row += n)
        for (uint32_t col = 0; col < bsize_wide; col += n) {
            orc_residual8(input + (size_t)row * input_stride + col, input_stride, pred + (size_t)row * pred_stride + col, pred_stride, res, n, n, n);
            switch (n) {
            case 4: orc_hadamard_4x4(res, n, coeff); break;
            case 8: orc_hadamard_8x8(res, n, coeff); break;
            case 16: orc_hadamard_16x16(res, n, coeff); break;
            default: orc_hadamard_32x32(res, n, coeff); break;
            }
            cost += (uint32_t)orc_satd(coeff, (int)(n * n));
        }
    return cost;
}

/* PSYEX psy-RD energy term, Codec/psy_rd.c:64-168 (8-bit) and :171-274 (10-bit).
 * 8-bit: the reference evaluates the Hadamard sums two columns at a time in packed 2 x 16-bit integers; with pixel
 * inputs (against its all-zero buffer) no packed half can overflow -- sum |c| over 8 coefficients <=
 * sqrt(8 * 64 * 64 * 255^2) < 2^16 -- so the packed arithmetic equals the plain unnormalised 2-D Hadamard:
 * sa8d = (sum |H8 X H8'| + 2) >> 2, satd4 = sum |H4 X H4'| >> 1.
 * 10-bit: the reference packs 2 x 32 bits into 64, but its 4-point butterflies (the HADAMARD4 macro used at
 * psy_rd.c:189,192-193,220) keep their temporaries in 32 bits: only the low half survives every butterfly, zero-extended,
 * and the per-half absolute value (psy_rd.c:54-59) then sees carries / borrows of the 64-bit sums in the high half.
 * That arithmetic is part of the encoder's behaviour, so it is restated bit for bit below (pack32 / bfly_low / abs_halves).
 * The block's "energy" is the Hadamard sum minus (sum of pixels) >> 2; the distortion sums |energy(input) -
 * energy(recon)| over the 8x8 tiles (4x4 tiles when a side is below 8) and scales by >> 1 (8-bit) or << 2 (10-bit). */
static void had1d(int32_t *v, int n, int stride) {
    for (int len = 1; len < n; len <<= 1)
        for (int i = 0; i < n; i += 2 * len)
            for (int j = i; j < i + len; j++) {
                const int32_t a = v[j * stride], b = v[(j + len) * stride];
                v[j * stride] = a + b; v[(j + len) * stride] = a - b;
            }
}
static uint64_t pack32(int32_t x0, int32_t x1) { return (uint64_t)(int64_t)(x0 + x1) + ((uint64_t)(int64_t)(x0 - x1) << 32); }
static void bfly_low(uint64_t d[4], uint64_t s0, uint64_t s1, uint64_t s2, uint64_t s3) { /* 4-point butterfly carried out in 32 bits */
    const uint32_t t0 = (uint32_t)(s0 + s1), t1 = (uint32_t)(s0 - s1), t2 = (uint32_t)(s2 + s3), t3 = (uint32_t)(s2 - s3);
    d[0] = (uint32_t)(t0 + t2); d[1] = (uint32_t)(t1 + t3); d[2] = (uint32_t)(t0 - t2); d[3] = (uint32_t)(t1 - t3);
}
static uint64_t abs_halves(uint64_t a) { /* |.| of the two 32-bit halves, in the borrow-compensating form of the reference */
    const uint64_t m = (a >> 31) & 0x100000001ull, s = (m << 32) - m;
    return (a + s) ^ s;
}
static uint64_t fold_halves(uint64_t b) { return (uint32_t)b + (b >> 32); }
static int64_t psy_had16(const uint16_t *p, uint32_t stride, int n) {
    uint64_t t[8][4], sum = 0;
    if (n == 8) {
        for (int i = 0; i < 8; i++) {
            const uint16_t *r = p + (size_t)i * stride;
            bfly_low(t[i], pack32(r[0], r[1]), pack32(r[2], r[3]), pack32(r[4], r[5]), pack32(r[6], r[7]));
        }
        for (int i = 0; i < 4; i++) {
            uint64_t a[8], b = 0;
            bfly_low(a, t[0][i], t[1][i], t[2][i], t[3][i]);
            bfly_low(a + 4, t[4][i], t[5][i], t[6][i], t[7][i]);
            for (int k = 0; k < 4; k++) b += abs_halves(a[k] + a[k + 4]) + abs_halves(a[k] - a[k + 4]);
            sum += fold_halves(b);
        }
        return (int64_t)((sum + 2) >> 2);
    }
    for (int i = 0; i < 4; i++) {
        const uint16_t *r = p + (size_t)i * stride;
        const uint64_t b0 = pack32(r[0], r[1]), b1 = pack32(r[2], r[3]);
        t[i][0] = b0 + b1; t[i][1] = b0 - b1;
    }
    for (int i = 0; i < 2; i++) {
        uint64_t a[4];
        bfly_low(a, t[0][i], t[1][i], t[2][i], t[3][i]);
        sum += fold_halves(abs_halves(a[0]) + abs_halves(a[1]) + abs_halves(a[2]) + abs_halves(a[3]));
    }
    return (int64_t)(sum >> 1);
}
static int32_t psy_tile_energy(const void *pix, int is16, uint32_t stride, int n) {
    int32_t m[64];
    int64_t sum = 0, acc = 0, had;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            const int32_t v = is16 ? ((const uint16_t *)pix)[(size_t)y * stride + x] : ((const uint8_t *)pix)[(size_t)y * stride + x];
            m[y * n + x] = v;
            sum += v;
        }
    if (is16) had = psy_had16((const uint16_t *)pix, stride, n);
    else {
        for (int y = 0; y < n; y++) had1d(m + y * n, n, 1);
        for (int x = 0; x < n; x++) had1d(m + x, n, n);
        for (int i = 0; i < n * n; i++) acc += m[i] < 0 ? -m[i] : m[i];
        had = n == 8 ? (acc + 2) >> 2 : acc >> 1;
    }
    return (int32_t)(had - (sum >> 2));
}
uint64_t orc_psy_distortion(const void *input, uint32_t input_stride, const void *recon, uint32_t recon_stride, uint32_t width, uint32_t height,
                            int is16) {
    const int n   = (width >= 8 && height >= 8) ? 8 : 4;
    const size_t bpp = is16 ? 2 : 1;
    uint64_t total = 0;
    for (uint32_t i = 0; i < height; i += n)
        for (uint32_t j = 0; j < width; j += n) {
            const int32_t a = psy_tile_energy((const uint8_t *)input + ((size_t)i * input_stride + j) * bpp, is16, input_stride, n);
            const int32_t b = psy_tile_energy((const uint8_t *)recon + ((size_t)i * recon_stride + j) * bpp, is16, recon_stride, n);
            total += (uint64_t)(a > b ? a - b : b - a);
        }
    return is16 ? total << 2 : total >> 1;
}

/* The 2-tap bilinear interpolation in front of svt_aom_sub_pixel_variance{W}x{H}_c (C_DEFAULT/variance.c:28-75,308-318):
 * horizontal pass on h + 1 rows into 16-bit, vertical pass back to the pixel type; taps {128 - 16k, 16k} (filter.h:39-48),
 * rounding shift by FILTER_BITS = 7.  Samples are widened to uint16 so that one routine serves both bit depths. */
static void bilinear_block(const void *src, int is16, uint32_t stride, int w, int h, int xo, int yo, uint16_t *out /* [h][w] */) {
    uint16_t *mid = malloc(sizeof(uint16_t) * (size_t)(h + 1) * w);
    const int fx0 = 128 - 16 * xo, fx1 = 16 * xo, fy0 = 128 - 16 * yo, fy1 = 16 * yo;
    for (int y = 0; y < h + 1; y++)
        for (int x = 0; x < w; x++) {
            const int a0 = is16 ? ((const uint16_t *)src)[(size_t)y * stride + x] : ((const uint8_t *)src)[(size_t)y * stride + x];
            /* with a zero second tap the reference still multiplies the neighbour by 0: no read is needed for the value */
            const int a1 = fx1 ? (is16 ? ((const uint16_t *)src)[(size_t)y * stride + x + 1] : ((const uint8_t *)src)[(size_t)y * stride + x + 1]) : 0;
            mid[y * w + x] = (uint16_t)((a0 * fx0 + a1 * fx1 + 64) >> 7);
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) out[y * w + x] = (uint16_t)((mid[y * w + x] * fy0 + (fy1 ? mid[(y + 1) * w + x] * fy1 : 0) + 64) >> 7);
    free(mid);
}

uint32_t orc_sub_pixel_variance8(const uint8_t *a, int a_stride, int xoffset, int yoffset, const uint8_t *b, int b_stride, int w, int h, uint32_t *sse) {
    uint16_t *t16 = malloc(sizeof(uint16_t) * (size_t)w * h);
    uint8_t  *t8  = malloc((size_t)w * h);
    bilinear_block(a, 0, (uint32_t)a_stride, w, h, xoffset, yoffset, t16);
    for (int i = 0; i < w * h; i++) t8[i] = (uint8_t)t16[i];
    const uint32_t v = orc_variance8(t8, w, b, b_stride, w, h, sse);
    free(t16); free(t8);
    return v;
}

/* svt_aom_highbd_10_variance{W}x{H}_c = highbd_variance64 + highbd_10_variance + HIGHBD_VAR (Codec/svt_psnr.c:139-177):
 * 64-bit sums of the 10-bit differences, both rounded back to the 8-bit scale, variance clamped at zero. */
uint32_t orc_highbd_10_variance(const uint16_t *a, int a_stride, const uint16_t *b, int b_stride, int w, int h, uint32_t *sse) {
    uint64_t sse_long = 0;
    int64_t  sum_long = 0;
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            const int diff = (int)a[(size_t)i * a_stride + j] - (int)b[(size_t)i * b_stride + j];
            sum_long += diff;
            sse_long += (uint32_t)(diff * diff);
        }
    *sse                = (uint32_t)((sse_long + 8) >> 4);
    const int     sum   = (int)((sum_long + 2) >> 2);
    const int64_t var   = (int64_t)*sse - ((int64_t)sum * sum) / (w * h);
    return var >= 0 ? (uint32_t)var : 0;
}

/* svt_spatial_full_distortion_kernel_facade's spy-rd biases (C_DEFAULT/picture_operators_c.c:130-171) on an SSE already computed.
 * Enumerators: PredictionMode Codec/definitions.h:1126-1162 (DC 0, V 1, H 2, SMOOTH 9..11, PAETH 12, first inter 13, compound
 * 17..24), CompoundType :1197-1202 (AVERAGE 0, DISTWTD 1, WEDGE 2, DIFFWTD 3). */
int64_t orc_spy_rd_facade(int64_t dist, uint32_t w, uint32_t h, uint8_t mode, uint8_t comp, uint8_t tli, double psy_rd, uint8_t spy_rd) {
    static const uint8_t intra_weights[6] = {8, 8, 9, 10, 11, 12};
    const int blurry_intra = mode == 0 || (mode >= 9 && mode <= 11), neutral_intra = mode == 1 || mode == 2 || mode == 12;
    const int is_intra = mode < 13, is_compound = mode >= 17 && mode < 25;
    if (spy_rd != 1) return dist;
    if (blurry_intra) {
        if (psy_rd == 0.0) dist = dist * 5 / 4;
    } else if (neutral_intra)
        dist = dist * 9 / 8;
    else if (is_compound) {
        if (comp == 0 || comp == 1) dist = dist * 5 / 4;
        else if (comp == 3) dist = dist * 9 / 8;
    }
    if (is_intra) {
        if (tli >= 2) dist = dist * intra_weights[tli] / 8;
        if (w == 64 && h == 64) dist = dist * 3 / 2;
        else if (w * h <= 32 * 32) dist = dist * 17 / 16;
    }
    return dist;
}

int orc_block_stats_batch(const SvtHipBlockStatsDesc *d) {
    if (!d || (d->bit_depth != 8 && d->bit_depth != 10) || !d->src || !d->ref || !d->jobs) return 2;
    if (d->satd && d->bit_depth != 8) return 2;
    if ((d->variance10 || d->var_sse10) && d->bit_depth != 10) return 2;
    for (uint32_t j = 0; j < d->n_jobs; j++) {
        const SvtHipBlockJob jb = d->jobs[j];
        const int w = jb.width, h = jb.height;
        uint32_t sad, var, vsse;
        uint64_t sse;
        const int subpel = (jb.subpel_x | jb.subpel_y) & 7;
        uint32_t  s_stride = d->src_stride;
        uint16_t *f16 = NULL;
        uint8_t  *f8  = NULL;
        if (subpel) { /* every statistic then sees the interpolated source block */
            f16 = malloc(sizeof(uint16_t) * (size_t)w * h);
            bilinear_block(d->bit_depth == 8 ? (const void *)((const uint8_t *)d->src + jb.src_offset) : (const void *)((const uint16_t *)d->src + jb.src_offset),
                           d->bit_depth != 8, d->src_stride, w, h, jb.subpel_x & 7, jb.subpel_y & 7, f16);
            if (d->bit_depth == 8) { f8 = malloc((size_t)w * h); for (int i = 0; i < w * h; i++) f8[i] = (uint8_t)f16[i]; }
            s_stride = (uint32_t)w;
        }
        if (d->bit_depth == 8) {
            const uint8_t *s = subpel ? f8 : (const uint8_t *)d->src + jb.src_offset, *r = (const uint8_t *)d->ref + jb.ref_offset;
            sad = orc_nxm_sad(s, s_stride, r, d->ref_stride, (uint32_t)h, (uint32_t)w);
            sse = orc_spatial_sse8(s, 0, s_stride, r, 0, d->ref_stride, (uint32_t)w, (uint32_t)h);
            var = orc_variance8(s, (int)s_stride, r, (int)d->ref_stride, w, h, &vsse);
            if (d->satd) {
                const int sq = w == h && (w == 4 || w == 8 || w == 16 || w == 32 || w == 64 || w == 128);
                d->satd[j] = sq ? orc_hadamard_path(s, s_stride, r, d->ref_stride, (uint32_t)w) : 0;
            }
        } else {
            const uint16_t *s = subpel ? f16 : (const uint16_t *)d->src + jb.src_offset, *r = (const uint16_t *)d->ref + jb.ref_offset;
            sad = orc_sad_16b(s, s_stride, r, d->ref_stride, (uint32_t)h, (uint32_t)w);
            sse = orc_spatial_sse16(s, 0, s_stride, r, 0, d->ref_stride, (uint32_t)w, (uint32_t)h);
            var = orc_variance16(s, (int)s_stride, r, (int)d->ref_stride, w, h, &vsse);
            if (d->variance10 || d->var_sse10) {
                uint32_t s10;
                const uint32_t v10 = orc_highbd_10_variance(s, (int)s_stride, r, (int)d->ref_stride, w, h, &s10);
                if (d->variance10) d->variance10[j] = v10;
                if (d->var_sse10) d->var_sse10[j] = s10;
            }
        }
        if (d->psy_energy || d->psy_dist) {
            if ((w & 3) || (h & 3)) return 2;
            const size_t bpp = d->bit_depth == 8 ? 1 : 2;
            const void *ps = subpel ? (d->bit_depth == 8 ? (const void *)f8 : (const void *)f16) : (const void *)((const uint8_t *)d->src + (size_t)jb.src_offset * bpp);
            const uint64_t e = orc_psy_distortion(ps, s_stride,
                                                  (const uint8_t *)d->ref + (size_t)jb.ref_offset * bpp, d->ref_stride, (uint32_t)w, (uint32_t)h, d->bit_depth != 8);
            if (d->psy_energy) d->psy_energy[j] = e;
            if (d->psy_dist) d->psy_dist[j] = (uint64_t)((double)e * d->psy_rd); /* get_svt_psy_full_dist, psy_rd.c:277-293 */
        }
        if (d->psy_sse) { /* svt_spatial_psy_distortion_kernel_c, picture_operators_c.c:85-112 */
            uint64_t psy = 0;
            if (d->psy_rd > 0.0) {
                if ((w & 3) || (h & 3)) return 2;
                const size_t bpp = d->bit_depth == 8 ? 1 : 2;
                const void *ps = subpel ? (d->bit_depth == 8 ? (const void *)f8 : (const void *)f16) : (const void *)((const uint8_t *)d->src + (size_t)jb.src_offset * bpp);
                psy = (uint64_t)((double)orc_psy_distortion(ps, s_stride, (const uint8_t *)d->ref + (size_t)jb.ref_offset * bpp, d->ref_stride, (uint32_t)w,
                                                            (uint32_t)h, d->bit_depth != 8) * d->psy_rd);
            }
            d->psy_sse[j] = sse + psy;
        }
        if (d->facade_dist) {
            if (!d->pred_mode || !d->compound_type || d->temporal_layer_index > 5) return 2;
            d->facade_dist[j] = (uint64_t)orc_spy_rd_facade((int64_t)sse, (uint32_t)w, (uint32_t)h, d->pred_mode[j], d->compound_type[j],
                                                            d->temporal_layer_index, d->psy_rd, d->spy_rd);
        }
        if (d->sad) d->sad[j] = sad;
        if (d->sse) d->sse[j] = sse;
        if (d->variance) d->variance[j] = var;
        if (d->var_sse) d->var_sse[j] = vsse;
        free(f16); free(f8);
    }
    return 0;
}

size_t orc_sizeof_stats(int what) { return what == 0 ? sizeof(SvtHipBlockStatsDesc) : sizeof(SvtHipBlockJob); }
