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

int orc_block_stats_batch(const SvtHipBlockStatsDesc *d) {
    if (!d || (d->bit_depth != 8 && d->bit_depth != 10) || !d->src || !d->ref || !d->jobs) return 2;
    if (d->satd && d->bit_depth != 8) return 2;
    for (uint32_t j = 0; j < d->n_jobs; j++) {
        const SvtHipBlockJob jb = d->jobs[j];
        const int w = jb.width, h = jb.height;
        uint32_t sad, var, vsse;
        uint64_t sse;
        if (d->bit_depth == 8) {
            const uint8_t *s = (const uint8_t *)d->src + jb.src_offset, *r = (const uint8_t *)d->ref + jb.ref_offset;
            sad = orc_nxm_sad(s, d->src_stride, r, d->ref_stride, (uint32_t)h, (uint32_t)w);
            sse = orc_spatial_sse8(s, 0, d->src_stride, r, 0, d->ref_stride, (uint32_t)w, (uint32_t)h);
            var = orc_variance8(s, (int)d->src_stride, r, (int)d->ref_stride, w, h, &vsse);
            if (d->satd) {
                const int sq = w == h && (w == 4 || w == 8 || w == 16 || w == 32 || w == 64 || w == 128);
                d->satd[j] = sq ? orc_hadamard_path(s, d->src_stride, r, d->ref_stride, (uint32_t)w) : 0;
            }
        } else {
            const uint16_t *s = (const uint16_t *)d->src + jb.src_offset, *r = (const uint16_t *)d->ref + jb.ref_offset;
            sad = orc_sad_16b(s, d->src_stride, r, d->ref_stride, (uint32_t)h, (uint32_t)w);
            sse = orc_spatial_sse16(s, 0, d->src_stride, r, 0, d->ref_stride, (uint32_t)w, (uint32_t)h);
            var = orc_variance16(s, (int)d->src_stride, r, (int)d->ref_stride, w, h, &vsse);
        }
        if (d->sad) d->sad[j] = sad;
        if (d->sse) d->sse[j] = sse;
        if (d->variance) d->variance[j] = var;
        if (d->var_sse) d->var_sse[j] = vsse;
    }
    return 0;
}

size_t orc_sizeof_stats(int what) { return what == 0 ? sizeof(SvtHipBlockStatsDesc) : sizeof(SvtHipBlockJob); }
