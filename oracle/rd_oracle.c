/*
 * rd_oracle.c -- TEST INFRASTRUCTURE.  Chains the oracle's leaf kernels in the order the reference's tx_type_search
 * loop body calls them (Codec/product_coding_loop.c:4764-4934): residual -> forward transform (+ 64-point repack)
 * -> SATD -> quantize -> coefficient-domain distortion -> inverse transform + recon -> pixel-domain SSE.
 * Parity checker and cpu_baseline for svt_hip_rd_batch; every leaf is pinned in tests/test_dsp_oracle_vs_ref.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/svt_hip_dsp.h"

void     orc_residual8(const uint8_t *, uint32_t, const uint8_t *, uint32_t, int16_t *, uint32_t, uint32_t, uint32_t);
void     orc_residual16(const uint16_t *, uint32_t, const uint16_t *, uint32_t, int16_t *, uint32_t, uint32_t, uint32_t);
void     orc_fwd_txfm2d(const int16_t *, int32_t *, uint32_t, int, int);
uint64_t orc_handle_transform(int32_t *, int);
int      orc_satd(const int32_t *, int);
void     orc_quantize_b(const int32_t *, intptr_t, const int16_t *, const int16_t *, const int16_t *, const int16_t *, int32_t *, int32_t *,
                        const int16_t *, uint16_t *, const int16_t *, const uint8_t *, const uint8_t *, int, int);
void     orc_quantize_fp(const int32_t *, intptr_t, const int16_t *, const int16_t *, int32_t *, int32_t *, const int16_t *, uint16_t *,
                         const int16_t *, const uint8_t *, const uint8_t *, int, int);
void     orc_full_distortion32(const int32_t *, uint32_t, const int32_t *, uint32_t, uint64_t[2], uint32_t, uint32_t);
void     orc_inv_txfm2d_add(const int32_t *, const uint16_t *, int32_t, uint16_t *, int32_t, int, int, int);
int      orc_scan_order(int, int, int16_t *, int16_t *);
int      orc_tx_size_wide(int);
int      orc_tx_size_high(int);
uint8_t  orc_compute_cul_level(const int16_t *, const int32_t *, const uint16_t *);

static const uint8_t k_log_scale[19] = {0, 0, 0, 1, 2, 0, 0, 0, 0, 1, 1, 2, 2, 0, 0, 0, 0, 1, 1}; /* av1_get_tx_scale_tab, full_loop.h:53 */

/* Host-memory mirror of svt_hip_rd_batch: every pointer in `d` is a host pointer. */
int orc_rd_batch(const SvtHipRdBatchDesc *d) {
    const int ts = d->tx_size, W = orc_tx_size_wide(ts), H = orc_tx_size_high(ts), WP = W > 32 ? 32 : W, HP = H > 32 ? 32 : H, NP = WP * HP;
    const int bd = d->bit_depth, hbd = bd != 8;
    int16_t  *res  = malloc(sizeof(int16_t) * W * H), scan[3][1024], iscan[1024];
    int32_t  *co   = malloc(sizeof(int32_t) * W * H), *q = malloc(sizeof(int32_t) * NP), *dq = malloc(sizeof(int32_t) * NP);
    uint16_t *p16  = malloc(sizeof(uint16_t) * W * H), *r16 = malloc(sizeof(uint16_t) * W * H);
    orc_scan_order(ts, 0, scan[0], iscan);
    orc_scan_order(ts, 10, scan[1], iscan);
    orc_scan_order(ts, 11, scan[2], iscan);
    for (uint32_t j = 0; j < d->n_jobs; j++) {
        const SvtHipTxJob    *jb = &d->jobs[j];
        const SvtHipQuantRow *qr = &d->quant_rows[jb->quant_row];
        const int             tt = jb->tx_type & 15, kind = tt >= 10 ? ((tt & 1) ? 2 : 1) : 0;
        if (hbd)
            orc_residual16((const uint16_t *)d->src + jb->src_offset, d->src_stride, (const uint16_t *)d->pred + jb->pred_offset, d->pred_stride, res, W, W, H);
        else
            orc_residual8((const uint8_t *)d->src + jb->src_offset, d->src_stride, (const uint8_t *)d->pred + jb->pred_offset, d->pred_stride, res, W, W, H);
        orc_fwd_txfm2d(res, co, W, tt, ts);
        /* partial-frequency shapes (av1_estimate_transform_N2 / _N4 / _ONLY_DC, transforms.c:2633-2948): the pruned 1-D
         * kernels produce the full transform's low-frequency outputs, everything else is zeroed (:5266-5270, :6830-6834,
         * :2936-2946) and no energy is attributed to the discarded quadrants (svt_handle_transform*_N2_N4_c, :2514-2543) */
        if (jb->pf_shape)
            for (int r = 0; r < H; r++)
                for (int c = 0; c < W; c++) {
                    const int keep = jb->pf_shape == 3 ? (r == 0 && c == 0) : (c < (W >> jb->pf_shape) && r < (H >> jb->pf_shape));
                    if (!keep) co[r * W + c] = 0;
                }
        d->three_quad_energy[j] = orc_handle_transform(co, ts);
        if (jb->pf_shape) d->three_quad_energy[j] = 0;
        d->satd[j]              = (uint32_t)orc_satd(co, NP);
        const uint8_t *qm = tt < 9 ? d->qmatrix : NULL, *iqm = tt < 9 ? d->iqmatrix : NULL; /* IS_2D_TRANSFORM, full_loop.c:1606-1608 */
        if (d->quant_kind == 0)
            orc_quantize_b(co, NP, qr->zbin, qr->round, qr->quant, qr->quant_shift, q, dq, qr->dequant, &d->eob[j], scan[kind], qm, iqm, k_log_scale[ts], hbd);
        else
            orc_quantize_fp(co, NP, qr->round_fp, qr->quant_fp, q, dq, qr->dequant, &d->eob[j], scan[kind], qm, iqm,
                            d->quant_kind == 2 ? 0 : k_log_scale[ts] /* 2: the TPL dispenser's plain svt_av1_quantize_fp, src_ops_process.c:225-249 */, hbd);
        if (d->cul_level) d->cul_level[j] = orc_compute_cul_level(scan[kind], q, &d->eob[j]); /* full_loop.c:1832-1836 */
        orc_full_distortion32(co, WP, dq, WP, &d->dist_coeff[2 * (size_t)j], WP, HP);
        for (int r = 0; r < H; r++)
            for (int c = 0; c < W; c++)
                p16[r * W + c] = hbd ? ((const uint16_t *)d->pred)[jb->pred_offset + (size_t)r * d->pred_stride + c]
                                     : ((const uint8_t *)d->pred)[jb->pred_offset + (size_t)r * d->pred_stride + c];
        orc_inv_txfm2d_add(dq, p16, W, r16, W, tt, ts, bd);
        uint64_t sse = 0;
        for (int r = 0; r < H; r++)
            for (int c = 0; c < W; c++) {
                const int s = hbd ? ((const uint16_t *)d->src)[jb->src_offset + (size_t)r * d->src_stride + c]
                                  : ((const uint8_t *)d->src)[jb->src_offset + (size_t)r * d->src_stride + c];
                const int64_t e = (int64_t)s - r16[r * W + c];
                sse += (uint64_t)(e * e);
                if (d->recon) {
                    if (hbd) ((uint16_t *)d->recon)[jb->pred_offset + (size_t)r * d->pred_stride + c] = r16[r * W + c];
                    else ((uint8_t *)d->recon)[jb->pred_offset + (size_t)r * d->pred_stride + c] = (uint8_t)r16[r * W + c];
                }
            }
        d->sse[j] = sse;
        if (d->coeff) memcpy(d->coeff + (size_t)j * NP, co, sizeof(int32_t) * NP);
        if (d->qcoeff) memcpy(d->qcoeff + (size_t)j * NP, q, sizeof(int32_t) * NP);
        if (d->dqcoeff) memcpy(d->dqcoeff + (size_t)j * NP, dq, sizeof(int32_t) * NP);
    }
    free(res); free(co); free(q); free(dq); free(p16); free(r16);
    return 0;
}
size_t orc_sizeof_dsp(int what) { return what == 0 ? sizeof(SvtHipRdBatchDesc) : what == 1 ? sizeof(SvtHipTxJob) : sizeof(SvtHipQuantRow); }
