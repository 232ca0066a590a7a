/*
 * dsp_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement of the mode-decision RD leaf kernels of the reference
 * (residual, SSE / spatial distortion, coefficient-domain distortion, block error, SATD, Hadamard, variance,
 * the "b" and "fp" quantizers).  Parity checker only; pinned against the reference's `_c` functions in
 * tests/test_dsp_oracle_vs_ref.py and tests/golden/dsp_*.npz.  Citations are relative to Source/Lib/.
 */
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define QM_BITS 5 /* AOM_QM_BITS */

/* svt_residual_kernel8bit_c / svt_residual_kernel16bit_c: Codec/pic_operators.c:101-148 */
void orc_residual8(const uint8_t *in, uint32_t in_stride, const uint8_t *pred, uint32_t pred_stride, int16_t *res,
                   uint32_t res_stride, uint32_t w, uint32_t h) {
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++)
            res[y * res_stride + x] = (int16_t)((int16_t)in[y * in_stride + x] - (int16_t)pred[y * pred_stride + x]);
}
void orc_residual16(const uint16_t *in, uint32_t in_stride, const uint16_t *pred, uint32_t pred_stride, int16_t *res,
                    uint32_t res_stride, uint32_t w, uint32_t h) {
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++)
            res[y * res_stride + x] = (int16_t)((int16_t)in[y * in_stride + x] - (int16_t)pred[y * pred_stride + x]);
}

/* svt_spatial_full_distortion_kernel_c: C_DEFAULT/picture_operators_c.c:65-83;
 * svt_full_distortion_kernel16_bits_c: Codec/pic_operators.c:174-197 (offsets in samples) */
uint64_t orc_spatial_sse8(const uint8_t *a, uint32_t a_off, uint32_t a_stride, const uint8_t *b, int32_t b_off, uint32_t b_stride,
                          uint32_t w, uint32_t h) {
    uint64_t acc = 0;
    a += a_off;
    b += b_off;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const int64_t d = (int64_t)a[y * a_stride + x] - b[y * b_stride + x];
            acc += (uint64_t)(d * d);
        }
    return acc;
}
uint64_t orc_spatial_sse16(const uint16_t *a, uint32_t a_off, uint32_t a_stride, const uint16_t *b, int32_t b_off,
                           uint32_t b_stride, uint32_t w, uint32_t h) {
    uint64_t acc = 0;
    a += a_off;
    b += b_off;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const int64_t d = (int64_t)a[y * a_stride + x] - b[y * b_stride + x];
            acc += (uint64_t)(d * d);
        }
    return acc;
}

/* svt_full_distortion_kernel32_bits_c / _cbf_zero32_bits_c: Codec/pic_operators.c:150-172,202-222 */
void orc_full_distortion32(const int32_t *coeff, uint32_t c_stride, const int32_t *recon, uint32_t r_stride, uint64_t out[2], uint32_t w,
                           uint32_t h) {
    uint64_t res = 0, pred = 0;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const int64_t c = coeff[y * c_stride + x], d = c - recon[y * r_stride + x];
            res += (uint64_t)(d * d);
            pred += (uint64_t)(c * c);
        }
    out[0] = res;
    out[1] = pred;
}
void orc_full_distortion32_cbf_zero(const int32_t *coeff, uint32_t c_stride, uint64_t out[2], uint32_t w, uint32_t h) {
    uint64_t pred = 0;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const int64_t c = coeff[y * c_stride + x];
            pred += (uint64_t)(c * c);
        }
    out[0] = out[1] = pred;
}

/* svt_av1_block_error_c: Codec/common_dsp_rtcd.c:79-91.  SQR() multiplies two `int`s: the square is a
 * 32-bit (wrapping) product that is then widened. */
int64_t orc_block_error(const int32_t *coeff, const int32_t *dqcoeff, intptr_t n, int64_t *ssz) {
    int64_t err = 0, sq = 0;
    for (intptr_t i = 0; i < n; i++) {
        const uint32_t d = (uint32_t)coeff[i] - (uint32_t)dqcoeff[i], c = (uint32_t)coeff[i];
        err += (int32_t)(d * d);
        sq += (int32_t)(c * c);
    }
    *ssz = sq;
    return err;
}

/* svt_aom_satd_c: Codec/common_dsp_rtcd.c:70-77 */
int orc_satd(const int32_t *coeff, int n) {
    int acc = 0;
    for (int i = 0; i < n; i++) acc += abs(coeff[i]);
    return acc;
}

/* Hadamard 4/8/16/32: C_DEFAULT/picture_operators_c.c:176-326.  The 1-D passes keep 16-bit intermediates;
 * 4-point passes halve after the first butterfly; 16 and 32 combine four sub-transforms with >>1 / >>2. */
static void had_col4(const int16_t *s, ptrdiff_t st, int16_t *o) {
    const int16_t b0 = (int16_t)((s[0] + s[st]) >> 1), b1 = (int16_t)((s[0] - s[st]) >> 1);
    const int16_t b2 = (int16_t)((s[2 * st] + s[3 * st]) >> 1), b3 = (int16_t)((s[2 * st] - s[3 * st]) >> 1);
    o[0] = (int16_t)(b0 + b2);
    o[1] = (int16_t)(b1 + b3);
    o[2] = (int16_t)(b0 - b2);
    o[3] = (int16_t)(b1 - b3);
}
void orc_hadamard_4x4(const int16_t *src, ptrdiff_t stride, int32_t *coeff) {
    int16_t t[16], u[16];
    for (int i = 0; i < 4; i++) had_col4(src + i, stride, t + 4 * i);
    for (int i = 0; i < 4; i++) had_col4(t + i, 4, u + 4 * i);
    for (int i = 0; i < 16; i++) coeff[i] = u[i];
}
static void had_col8(const int16_t *s, ptrdiff_t st, int16_t *o) {
    int16_t b[8], c[8];
    for (int i = 0; i < 4; i++) {
        b[2 * i]     = (int16_t)(s[2 * i * st] + s[(2 * i + 1) * st]);
        b[2 * i + 1] = (int16_t)(s[2 * i * st] - s[(2 * i + 1) * st]);
    }
    for (int g = 0; g < 2; g++) {
        c[4 * g + 0] = (int16_t)(b[4 * g + 0] + b[4 * g + 2]);
        c[4 * g + 1] = (int16_t)(b[4 * g + 1] + b[4 * g + 3]);
        c[4 * g + 2] = (int16_t)(b[4 * g + 0] - b[4 * g + 2]);
        c[4 * g + 3] = (int16_t)(b[4 * g + 1] - b[4 * g + 3]);
    }
    static const uint8_t plus[4] = {0, 7, 3, 4}, minus[4] = {2, 6, 1, 5}; /* output slots of c[i]+c[i+4], c[i]-c[i+4] */
    for (int i = 0; i < 4; i++) {
        o[plus[i]]  = (int16_t)(c[i] + c[i + 4]);
        o[minus[i]] = (int16_t)(c[i] - c[i + 4]);
    }
}
void orc_hadamard_8x8(const int16_t *src, ptrdiff_t stride, int32_t *coeff) {
    int16_t t[64], u[64];
    for (int i = 0; i < 8; i++) had_col8(src + i, stride, t + 8 * i);
    for (int i = 0; i < 8; i++) had_col8(t + i, 8, u + 8 * i);
    for (int i = 0; i < 64; i++) coeff[i] = u[i];
}
static void had_combine(int32_t *coeff, int n, int shift) {
    for (int i = 0; i < n; i++) {
        const int32_t a0 = coeff[i], a1 = coeff[n + i], a2 = coeff[2 * n + i], a3 = coeff[3 * n + i];
        const int32_t b0 = (a0 + a1) >> shift, b1 = (a0 - a1) >> shift, b2 = (a2 + a3) >> shift, b3 = (a2 - a3) >> shift;
        coeff[i]         = b0 + b2;
        coeff[n + i]     = b1 + b3;
        coeff[2 * n + i] = b0 - b2;
        coeff[3 * n + i] = b1 - b3;
    }
}
void orc_hadamard_16x16(const int16_t *src, ptrdiff_t stride, int32_t *coeff) {
    for (int i = 0; i < 4; i++) orc_hadamard_8x8(src + (i >> 1) * 8 * stride + (i & 1) * 8, stride, coeff + 64 * i);
    had_combine(coeff, 64, 1);
}
void orc_hadamard_32x32(const int16_t *src, ptrdiff_t stride, int32_t *coeff) {
    for (int i = 0; i < 4; i++) orc_hadamard_16x16(src + (i >> 1) * 16 * stride + (i & 1) * 16, stride, coeff + 256 * i);
    had_combine(coeff, 256, 2);
}

/* svt_aom_variance{W}x{H}_c: C_DEFAULT/variance.c:256-284 (variance_c + VAR macro): sse - sum^2/(w*h) with a
 * 64-bit product and truncating division, returned as uint32; svt_aom_variance_highbd_c: :278-296 */
uint32_t orc_variance8(const uint8_t *a, int a_stride, const uint8_t *b, int b_stride, int w, int h, uint32_t *sse) {
    int      sum = 0;
    uint32_t sq  = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int d = a[y * a_stride + x] - b[y * b_stride + x];
            sum += d;
            sq += (uint32_t)(d * d);
        }
    *sse = sq;
    return (uint32_t)(sq - (uint32_t)(((int64_t)sum * sum) / (w * h)));
}
uint32_t orc_variance16(const uint16_t *a, int a_stride, const uint16_t *b, int b_stride, int w, int h, uint32_t *sse) {
    int      sum = 0;
    uint32_t sq  = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int d = a[y * a_stride + x] - b[y * b_stride + x];
            sum += d;
            sq += (uint32_t)(d * d);
        }
    *sse = sq;
    return (uint32_t)(sq - ((int64_t)sum * sum) / (w * h));
}

/* ------------------------------------------------------------------------------------------------
 * Quantizers.  `tab` = {zbin[2], round[2], quant[2], quant_shift[2], dequant[2]} (DC, AC) as the reference
 * passes them from MacroblockPlane / Dequants (Codec/full_loop.c:1627-1685).
 * ------------------------------------------------------------------------------------------------ */
static int32_t rpot(int32_t v, int n) { return n ? (v + (1 << (n - 1))) >> n : v; } /* ROUND_POWER_OF_TWO */

/* svt_aom_quantize_b_c_ii (lbd, :29-79) and svt_aom_highbd_quantize_b_c (:149-198) */
void orc_quantize_b(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round, const int16_t *quant,
                    const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff, const int16_t *dequant, uint16_t *eob_out,
                    const int16_t *scan, const uint8_t *qm, const uint8_t *iqm, int log_scale, int highbd) {
    const int32_t zb[2] = {rpot(zbin[0], log_scale), rpot(zbin[1], log_scale)};
    intptr_t      eob = -1, last = n;
    memset(qcoeff, 0, (size_t)n * sizeof(*qcoeff));
    memset(dqcoeff, 0, (size_t)n * sizeof(*dqcoeff));
    if (!highbd) /* pre-scan from the end: trailing coefficients inside the zero bin are dropped */
        for (intptr_t i = n - 1; i >= 0; i--) {
            const int     rc = scan[i];
            const int32_t wt = qm ? qm[rc] : (1 << QM_BITS), c = coeff[rc] * wt;
            if (c < zb[rc != 0] * (1 << QM_BITS) && c > -zb[rc != 0] * (1 << QM_BITS))
                last--;
            else
                break;
        }
    for (intptr_t i = 0; i < last; i++) {
        const int     rc = scan[i], ac = rc != 0;
        const int32_t c = coeff[rc], sign = c < 0 ? -1 : 0, a = (c ^ sign) - sign;
        const int32_t wt = qm ? qm[rc] : (1 << QM_BITS), iwt = iqm ? iqm[rc] : (1 << QM_BITS);
        if (highbd) {
            const int32_t cw = c * wt;
            if (!(cw >= zb[ac] * (1 << QM_BITS) || cw <= -zb[ac] * (1 << QM_BITS)))
                continue;
        } else if (!(a * wt >= (zb[ac] << QM_BITS)))
            continue;
        int64_t t = a + rpot(round[ac], log_scale);
        if (!highbd)
            t = t < INT16_MIN ? INT16_MIN : (t > INT16_MAX ? INT16_MAX : t);
        t *= wt;
        const int32_t q   = (int32_t)(((((t * quant[ac]) >> 16) + t) * quant_shift[ac]) >> (16 - log_scale + QM_BITS));
        const int32_t deq = (dequant[ac] * iwt + (1 << (QM_BITS - 1))) >> QM_BITS;
        const int32_t dq  = (q * deq) >> log_scale;
        qcoeff[rc]        = (q ^ sign) - sign;
        dqcoeff[rc]       = (dq ^ sign) - sign;
        if (q)
            eob = i;
    }
    *eob_out = (uint16_t)(eob + 1);
}

/* quantize_fp_helper_c (lbd, :282-343) and highbd_quantize_fp_helper_c (:387-452) */
void orc_quantize_fp(const int32_t *coeff, intptr_t n, const int16_t *round, const int16_t *quant, int32_t *qcoeff, int32_t *dqcoeff,
                     const int16_t *dequant, uint16_t *eob_out, const int16_t *scan, const uint8_t *qm, const uint8_t *iqm,
                     int log_scale, int highbd) {
    const int rnd[2] = {rpot(round[0], log_scale), rpot(round[1], log_scale)};
    int       eob    = -1;
    memset(qcoeff, 0, (size_t)n * sizeof(*qcoeff));
    memset(dqcoeff, 0, (size_t)n * sizeof(*dqcoeff));
    for (int i = 0; i < n; i++) {
        const int     rc = scan[i], ac = rc != 0;
        const int     c = coeff[rc], sign = c < 0 ? -1 : 0;
        int64_t       a = (c ^ sign) - sign;
        int           q = 0;
        if (!qm && !iqm) {
            if (highbd ? (((int)a << (1 + log_scale)) >= dequant[ac]) : ((a << (1 + log_scale)) >= (int32_t)dequant[ac])) {
                a += rnd[ac];
                if (!highbd)
                    a = a < INT16_MIN ? INT16_MIN : (a > INT16_MAX ? INT16_MAX : a);
                q = (int)((a * quant[ac]) >> (16 - log_scale));
                if (q || highbd) {
                    const int32_t dq = (q * dequant[ac]) >> log_scale;
                    qcoeff[rc]       = (q ^ sign) - sign;
                    dqcoeff[rc]      = (dq ^ sign) - sign;
                }
            }
        } else {
            const int wt = qm ? qm[rc] : (1 << QM_BITS), iwt = iqm ? iqm[rc] : (1 << QM_BITS);
            const int deq = (dequant[ac] * iwt + (1 << (QM_BITS - 1))) >> QM_BITS;
            if (a * wt >= (dequant[ac] << (QM_BITS - (1 + log_scale)))) {
                a += rnd[ac];
                if (!highbd)
                    a = a < INT16_MIN ? INT16_MIN : (a > INT16_MAX ? INT16_MAX : a);
                q = highbd ? (int)((a * quant[ac] * wt) >> (16 - log_scale + QM_BITS)) : (int)((a * wt * quant[ac]) >> (16 - log_scale + QM_BITS));
                const int32_t dq = (q * deq) >> log_scale;
                qcoeff[rc]       = (q ^ sign) - sign;
                dqcoeff[rc]      = (dq ^ sign) - sign;
            }
        }
        if (q)
            eob = i;
    }
    *eob_out = (uint16_t)(eob + 1);
}

/* svt_av1_compute_cul_level_c + set_dc_sign: Codec/full_loop.c:1338-1343,1449-1466 (COEFF_CONTEXT_BITS = 6, cabac_context_model.h:435) */
uint8_t orc_compute_cul_level(const int16_t *scan, const int32_t *quant_coeff, const uint16_t *eob) {
    int32_t cul = 0;
    for (int32_t c = 0; c < *eob; c++) {
        const int32_t v = quant_coeff[scan[c]];
        cul += v < 0 ? -v : v;
        if (cul >= 63) break; /* the clamp below makes the early exit invisible */
    }
    cul = cul < 63 ? cul : 63;
    if (quant_coeff[0] < 0) cul |= 1 << 6;
    else if (quant_coeff[0] > 0) cul += 2 << 6;
    return (uint8_t)cul;
}

/* svt_av1_fwht4x4_c: Codec/transforms.c:3099-3151 (lossless blocks only): the 4-point reversible Walsh-Hadamard, columns then
 * rows, 64-bit temporaries, outputs scaled by UNIT_QUANT_FACTOR = 4 (transforms.h:25-26).  Output order per pass: a, c, d, b. */
static void wht4(int64_t a, int64_t b, int64_t c, int64_t d, int64_t o[4]) {
    a += b;
    d -= c;
    const int64_t e = (a - d) >> 1;
    b = e - b;
    c = e - c;
    a -= c;
    d += b;
    o[0] = a; o[1] = c; o[2] = d; o[3] = b;
}
void orc_fwht4x4(const int16_t *input, int32_t *output, uint32_t stride) {
    int32_t t[16];
    int64_t o[4];
    for (int i = 0; i < 4; i++) { /* column i of the input -> row i of the intermediate */
        wht4(input[i], input[stride + i], input[2 * stride + i], input[3 * stride + i], o);
        for (int k = 0; k < 4; k++) t[4 * i + k] = (int32_t)o[k];
    }
    for (int i = 0; i < 4; i++) { /* column i of the intermediate -> column i of the output */
        wht4(t[i], t[4 + i], t[8 + i], t[12 + i], o);
        for (int k = 0; k < 4; k++) output[4 * k + i] = (int32_t)(o[k] * 4);
    }
}

/* get_hvs_modulation_factor: Codec/psy_rd.c:295-307 */
double orc_hvs_modulation_factor(double psy_rd, int is_islice, uint8_t temporal_layer_index) {
    if (is_islice) return psy_rd * 0.4;
    if (temporal_layer_index == 0) return psy_rd * 0.75;
    if (temporal_layer_index == 1) return psy_rd * 0.9;
    if (temporal_layer_index == 2) return psy_rd * 0.95;
    return psy_rd;
}
