/*
 * svt_hip_leaf.h -- pointer-level entries: the reference's kernel prototypes with a `_hip` suffix, to be installed
 * by assignment into the rtcd function pointers (Source/Lib/Codec/aom_dsp_rtcd.h, common_dsp_rtcd.h) after
 * svt_aom_setup_rtcd_internal (Globals/enc_handle.c:1444-1445).  Host pointers in, host results out, synchronous:
 * every call stages its rows into device memory, launches, and copies the result back.  A validation / drop-in
 * path -- one call is microseconds of CPU work but three PCIe round trips here; production goes through the batched
 * entries (svt_hip_me_picture, svt_hip_rd_batch, svt_hip_block_stats_batch).
 *
 * The leaf kernels of the reference have no error channel and no context argument: bind a context once with svt_hip_leaf_bind() (the
 * installer does).  FAIL CLOSED: the reference's dispatch always leaves a working kernel in every pointer (Codec/aom_dsp_rtcd.c:38-48,73-99);
 * an entry here that cannot run -- no bound context, a device error -- therefore calls, with the same arguments, the kernel the encoder had
 * in the slot before svt_hip_install_rtcd / svt_hip_rtcd_store overwrote it (the encoder's own SIMD or `_c` body), and keeps the failure
 * for svt_hip_leaf_status() / svt_hip_last_error() of the calling thread.  Nothing aborts.  An entry that was called directly (no installer,
 * hence no previous kernel) records the failure and returns zero / leaves its outputs alone.
 * The binding is PROCESS-GLOBAL and the entries run one at a time under one lock on the bound context's stream: this layer is the
 * validation / drop-in path and is single-stream by construction; concurrency lives in the batched entries.
 */
#ifndef SVT_HIP_LEAF_H
#define SVT_HIP_LEAF_H

#include <stddef.h>
#include <stdbool.h>
#include <stdint.h>
#include "svt_hip_me.h"

#ifdef __cplusplus
extern "C" {
#endif

int svt_hip_leaf_bind(SvtHipContext *ctx); /* NULL unbinds */

/* Installer: what svt_aom_setup_rtcd_internal (Codec/aom_dsp_rtcd.c:188; called at Globals/enc_handle.c:1444-1445) does for one SIMD
 * flavour.  A slot names one of the encoder's rtcd function pointers (the identifier in aom_dsp_rtcd.h / common_dsp_rtcd.h, e.g.
 * "svt_sad_loop_kernel") and gives the ADDRESS of that pointer variable; svt_hip_install_rtcd stores this library's entry with the same
 * prototype there (<name>_hip) and binds `ctx` for the pointer-level entries.  Names without an entry here are left untouched (the
 * encoder keeps its own kernel) and counted in *n_skipped.  Run it after the encoder's own set-up and before init_fn_ptr()
 * (Codec/av1me.c:31, enc_handle.c:1460: it copies pointer values).  svt_hip_rtcd_lookup returns the entry alone (NULL: none). */
typedef struct SvtHipRtcdSlot {
    const char *name;
    void      **slot;
} SvtHipRtcdSlot;
int         svt_hip_install_rtcd(SvtHipContext *ctx, const SvtHipRtcdSlot *slots, uint32_t n_slots, uint32_t *n_skipped);
const void *svt_hip_rtcd_lookup(const char *name);
/* The installer's first half alone: stores the entries and records the slots' previous kernels, binds no context -- every call goes to
 * the previous kernels until svt_hip_leaf_bind().  svt_hip_uninstall_rtcd puts the previous kernels back into the slots. */
int         svt_hip_rtcd_store(const SvtHipRtcdSlot *slots, uint32_t n_slots, uint32_t *n_skipped);
int         svt_hip_uninstall_rtcd(const SvtHipRtcdSlot *slots, uint32_t n_slots);
/* Since the last call: calls served by previous kernels, calls that failed with nowhere to go, the last failure's text (any pointer may
 * be null); returns the sum of the two counts.  svt_hip_leaf_inject_failure(1) makes every entry fail as if the device had (testing). */
int         svt_hip_leaf_status(unsigned long long *fallbacks, unsigned long long *unhandled, char *message, size_t message_bytes);
void        svt_hip_leaf_inject_failure(int on);

/* svt_sad_loop_kernel (aom_dsp_rtcd.h:779; C_DEFAULT/compute_sad_c.c:58-101) */
void svt_sad_loop_kernel_hip(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t block_height,
                             uint32_t block_width, uint64_t *best_sad, int16_t *x_search_center, int16_t *y_search_center,
                             uint32_t src_stride_raw, uint8_t skip_search_line, int16_t search_area_width, int16_t search_area_height);
/* svt_nxm_sad_kernel_helper_c (compute_sad_c.c:209), svt_aom_sad_16b_kernel_c (:39-56) */
uint32_t svt_nxm_sad_kernel_helper_hip(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride, uint32_t height, uint32_t width);
uint32_t svt_aom_sad_16b_kernel_hip(uint16_t *src, uint32_t src_stride, uint16_t *ref, uint32_t ref_stride, uint32_t height, uint32_t width);

/* svt_aom_variance{W}x{H} (aom_dsp_rtcd.h:470-560; C_DEFAULT/variance.c:256-296) */
unsigned int svt_aom_variance_hip(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride, int width, int height, unsigned int *sse);
#define SVT_HIP_DECL_VAR(W, H) unsigned int svt_aom_variance##W##x##H##_hip(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride, unsigned int *sse);
SVT_HIP_DECL_VAR(4, 4) SVT_HIP_DECL_VAR(4, 8) SVT_HIP_DECL_VAR(4, 16) SVT_HIP_DECL_VAR(8, 4) SVT_HIP_DECL_VAR(8, 8) SVT_HIP_DECL_VAR(8, 16)
SVT_HIP_DECL_VAR(8, 32) SVT_HIP_DECL_VAR(16, 4) SVT_HIP_DECL_VAR(16, 8) SVT_HIP_DECL_VAR(16, 16) SVT_HIP_DECL_VAR(16, 32) SVT_HIP_DECL_VAR(16, 64)
SVT_HIP_DECL_VAR(32, 8) SVT_HIP_DECL_VAR(32, 16) SVT_HIP_DECL_VAR(32, 32) SVT_HIP_DECL_VAR(32, 64) SVT_HIP_DECL_VAR(64, 16) SVT_HIP_DECL_VAR(64, 32)
SVT_HIP_DECL_VAR(64, 64) SVT_HIP_DECL_VAR(64, 128) SVT_HIP_DECL_VAR(128, 64) SVT_HIP_DECL_VAR(128, 128)
#undef SVT_HIP_DECL_VAR

/* svt_aom_sub_pixel_variance{W}x{H} (aom_dsp_rtcd.h; C_DEFAULT/variance.c:308-318): xoffset / yoffset in 1/8 samples */
unsigned int svt_aom_sub_pixel_variance_hip(const uint8_t *src, int src_stride, int xoffset, int yoffset, const uint8_t *ref, int ref_stride, int width,
                                            int height, unsigned int *sse);
#define SVT_HIP_DECL_SPVAR(W, H) \
    unsigned int svt_aom_sub_pixel_variance##W##x##H##_hip(const uint8_t *src, int src_stride, int xoffset, int yoffset, const uint8_t *ref, int ref_stride, unsigned int *sse);
SVT_HIP_DECL_SPVAR(4, 4) SVT_HIP_DECL_SPVAR(4, 8) SVT_HIP_DECL_SPVAR(4, 16) SVT_HIP_DECL_SPVAR(8, 4) SVT_HIP_DECL_SPVAR(8, 8) SVT_HIP_DECL_SPVAR(8, 16)
SVT_HIP_DECL_SPVAR(8, 32) SVT_HIP_DECL_SPVAR(16, 4) SVT_HIP_DECL_SPVAR(16, 8) SVT_HIP_DECL_SPVAR(16, 16) SVT_HIP_DECL_SPVAR(16, 32) SVT_HIP_DECL_SPVAR(16, 64)
SVT_HIP_DECL_SPVAR(32, 8) SVT_HIP_DECL_SPVAR(32, 16) SVT_HIP_DECL_SPVAR(32, 32) SVT_HIP_DECL_SPVAR(32, 64) SVT_HIP_DECL_SPVAR(64, 16) SVT_HIP_DECL_SPVAR(64, 32)
SVT_HIP_DECL_SPVAR(64, 64) SVT_HIP_DECL_SPVAR(64, 128) SVT_HIP_DECL_SPVAR(128, 64) SVT_HIP_DECL_SPVAR(128, 128)
#undef SVT_HIP_DECL_SPVAR

/* The 8x8-based SAD pyramid of the integer search (aom_dsp_rtcd.h:842-855; Codec/motion_estimation.c:98-425, me_sad_calculation.c:14) */
void svt_ext_all_sad_calculation_8x8_16x16_hip(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t mv, uint32_t *p_best_sad_8x8,
                                               uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8, uint32_t *p_best_mv16x16, uint32_t p_eight_sad16x16[16][8],
                                               uint32_t p_eight_sad8x8[64][8], bool sub_sad);
void svt_ext_eight_sad_calculation_32x32_64x64_hip(uint32_t p_sad16x16[16][8], uint32_t *p_best_sad_32x32, uint32_t *p_best_sad_64x64,
                                                   uint32_t *p_best_mv32x32, uint32_t *p_best_mv64x64, uint32_t mv, uint32_t p_sad32x32[4][8]);
void svt_ext_sad_calculation_8x8_16x16_hip(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t *p_best_sad_8x8,
                                           uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8, uint32_t *p_best_mv16x16, uint32_t mv, uint32_t *p_sad16x16,
                                           uint32_t *p_sad8x8, bool sub_sad);
void svt_ext_sad_calculation_32x32_64x64_hip(uint32_t *p_sad16x16, uint32_t *p_best_sad_32x32, uint32_t *p_best_sad_64x64, uint32_t *p_best_mv32x32,
                                             uint32_t *p_best_mv64x64, uint32_t mv, uint32_t *p_sad32x32);
void svt_initialize_buffer_32bits_hip(uint32_t *pointer, uint32_t count128, uint32_t count32, uint32_t value);

/* Quantizers (aom_dsp_rtcd.h:244-263; Codec/full_loop.c:29-79,149-198 "b", :282-474 "fp"): TranLow = int32_t, QmVal = uint8_t; the
 * table rows are MacroblockPlane rows ([0] = DC, [1] = AC); eob from iscan (the inverse of scan, as every av1_scan_orders entry is) */
#define SVT_HIP_DECL_QUANT_B(NAME)                                                                                                                    \
    void NAME(const int32_t *coeff_ptr, intptr_t n_coeffs, const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr,              \
              const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr, const int16_t *dequant_ptr, uint16_t *eob_ptr,               \
              const int16_t *scan, const int16_t *iscan, const uint8_t *qm_ptr, const uint8_t *iqm_ptr, const int32_t log_scale);
SVT_HIP_DECL_QUANT_B(svt_aom_quantize_b_hip) SVT_HIP_DECL_QUANT_B(svt_aom_highbd_quantize_b_hip)
SVT_HIP_DECL_QUANT_B(svt_av1_quantize_b_qm_hip) SVT_HIP_DECL_QUANT_B(svt_av1_highbd_quantize_b_qm_hip)
#undef SVT_HIP_DECL_QUANT_B
#define SVT_HIP_DECL_QUANT_FP(NAME)                                                                                                                   \
    void NAME(const int32_t *coeff_ptr, intptr_t n_coeffs, const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr,              \
              const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr, const int16_t *dequant_ptr, uint16_t *eob_ptr,               \
              const int16_t *scan, const int16_t *iscan);
SVT_HIP_DECL_QUANT_FP(svt_av1_quantize_fp_hip) SVT_HIP_DECL_QUANT_FP(svt_av1_quantize_fp_32x32_hip) SVT_HIP_DECL_QUANT_FP(svt_av1_quantize_fp_64x64_hip)
#undef SVT_HIP_DECL_QUANT_FP
void svt_av1_quantize_fp_qm_hip(const int32_t *coeff_ptr, intptr_t n_coeffs, const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr,
                                const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr, const int16_t *dequant_ptr, uint16_t *eob_ptr,
                                const int16_t *scan, const int16_t *iscan, const uint8_t *qm_ptr, const uint8_t *iqm_ptr, int16_t log_scale);
void svt_av1_highbd_quantize_fp_hip(const int32_t *coeff_ptr, intptr_t n_coeffs, const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr,
                                    const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr, const int16_t *dequant_ptr, uint16_t *eob_ptr,
                                    const int16_t *scan, const int16_t *iscan, int16_t log_scale);
void svt_av1_highbd_quantize_fp_qm_hip(const int32_t *coeff_ptr, intptr_t n_coeffs, const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr,
                                       const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr, const int16_t *dequant_ptr, uint16_t *eob_ptr,
                                       const int16_t *scan, const int16_t *iscan, const uint8_t *qm_ptr, const uint8_t *iqm_ptr, int16_t log_scale);

/* svt_av1_fwd_txfm2d_{W}x{H}{,_N2,_N4} (aom_dsp_rtcd.c:421-487; Codec/transforms.c:2259-2631,5202-5425,6769-6990): int16 residual ->
 * the full W x H coefficient array (the _N2 / _N4 entries keep the top-left half / quarter per dimension, zeros elsewhere) */
#define SVT_HIP_DECL_FWD(W, H)                                                                                            \
    void svt_av1_fwd_txfm2d_##W##x##H##_hip(int16_t *input, int32_t *output, uint32_t stride, uint8_t tx_type, uint8_t bd);    \
    void svt_av1_fwd_txfm2d_##W##x##H##_N2_hip(int16_t *input, int32_t *output, uint32_t stride, uint8_t tx_type, uint8_t bd); \
    void svt_av1_fwd_txfm2d_##W##x##H##_N4_hip(int16_t *input, int32_t *output, uint32_t stride, uint8_t tx_type, uint8_t bd);
SVT_HIP_DECL_FWD(4, 4) SVT_HIP_DECL_FWD(8, 8) SVT_HIP_DECL_FWD(16, 16) SVT_HIP_DECL_FWD(32, 32) SVT_HIP_DECL_FWD(64, 64) SVT_HIP_DECL_FWD(4, 8) SVT_HIP_DECL_FWD(8, 4)
SVT_HIP_DECL_FWD(8, 16) SVT_HIP_DECL_FWD(16, 8) SVT_HIP_DECL_FWD(16, 32) SVT_HIP_DECL_FWD(32, 16) SVT_HIP_DECL_FWD(32, 64) SVT_HIP_DECL_FWD(64, 32)
SVT_HIP_DECL_FWD(4, 16) SVT_HIP_DECL_FWD(16, 4) SVT_HIP_DECL_FWD(8, 32) SVT_HIP_DECL_FWD(32, 8) SVT_HIP_DECL_FWD(16, 64) SVT_HIP_DECL_FWD(64, 16)
#undef SVT_HIP_DECL_FWD

/* svt_handle_transform* (aom_dsp_rtcd.c:440-449; Codec/transforms.c:2374-2543): energy of the discarded frequencies of a 64-point size,
 * and the kept 32-wide rows packed to the front of the array in place */
uint64_t svt_handle_transform16x64_hip(int32_t *output);
uint64_t svt_handle_transform32x64_hip(int32_t *output);
uint64_t svt_handle_transform64x16_hip(int32_t *output);
uint64_t svt_handle_transform64x32_hip(int32_t *output);
uint64_t svt_handle_transform64x64_hip(int32_t *output);
uint64_t svt_handle_transform16x64_N2_N4_hip(int32_t *output);
uint64_t svt_handle_transform32x64_N2_N4_hip(int32_t *output);
uint64_t svt_handle_transform64x16_N2_N4_hip(int32_t *output);
uint64_t svt_handle_transform64x32_N2_N4_hip(int32_t *output);
uint64_t svt_handle_transform64x64_N2_N4_hip(int32_t *output);

/* svt_av1_inv_txfm2d_add_{W}x{H} (common_dsp_rtcd.h:100-141; Codec/inv_transforms.c:2459-2716): uint16 planes, bd 8 or 10, the three prototype
 * forms of the reference (squares; rectangles with tx_size + eob; the 4-wide / 4-high ones with tx_size).  TxType / TxSize are one-byte enums. */
#define SVT_HIP_DECL_INV_SQ(W, H) void svt_av1_inv_txfm2d_add_##W##x##H##_hip(const int32_t *input, uint16_t *output_r, int32_t stride_r, uint16_t *output_w, int32_t stride_w, uint8_t tx_type, int32_t bd);
SVT_HIP_DECL_INV_SQ(4, 4) SVT_HIP_DECL_INV_SQ(8, 8) SVT_HIP_DECL_INV_SQ(16, 16) SVT_HIP_DECL_INV_SQ(32, 32) SVT_HIP_DECL_INV_SQ(64, 64)
#undef SVT_HIP_DECL_INV_SQ
#define SVT_HIP_DECL_INV_RECT(W, H) void svt_av1_inv_txfm2d_add_##W##x##H##_hip(const int32_t *input, uint16_t *output_r, int32_t stride_r, uint16_t *output_w, int32_t stride_w, uint8_t tx_type, uint8_t tx_size, int32_t eob, int32_t bd);
SVT_HIP_DECL_INV_RECT(8, 16) SVT_HIP_DECL_INV_RECT(16, 8) SVT_HIP_DECL_INV_RECT(16, 32) SVT_HIP_DECL_INV_RECT(32, 16) SVT_HIP_DECL_INV_RECT(32, 64)
SVT_HIP_DECL_INV_RECT(64, 32) SVT_HIP_DECL_INV_RECT(8, 32) SVT_HIP_DECL_INV_RECT(32, 8) SVT_HIP_DECL_INV_RECT(16, 64) SVT_HIP_DECL_INV_RECT(64, 16)
#undef SVT_HIP_DECL_INV_RECT
#define SVT_HIP_DECL_INV_SMALL(W, H) void svt_av1_inv_txfm2d_add_##W##x##H##_hip(const int32_t *input, uint16_t *output_r, int32_t stride_r, uint16_t *output_w, int32_t stride_w, uint8_t tx_type, uint8_t tx_size, int32_t bd);
SVT_HIP_DECL_INV_SMALL(4, 8) SVT_HIP_DECL_INV_SMALL(8, 4) SVT_HIP_DECL_INV_SMALL(4, 16) SVT_HIP_DECL_INV_SMALL(16, 4)
#undef SVT_HIP_DECL_INV_SMALL

/* svt_aom_sse (aom_dsp_rtcd.h:53), svt_spatial_full_distortion_kernel / svt_full_distortion_kernel16_bits (common_dsp_rtcd.h:164-168) */
int64_t  svt_aom_sse_hip(const uint8_t *a, int a_stride, const uint8_t *b, int b_stride, int width, int height);
int64_t  svt_aom_highbd_sse_hip(const uint8_t *a8, int a_stride, const uint8_t *b8, int b_stride, int width, int height); /* a8 / b8: uint16_t samples (enc_inter_prediction.c:562) */
uint64_t svt_spatial_full_distortion_kernel_hip(uint8_t *input, uint32_t input_offset, uint32_t input_stride, uint8_t *recon, int32_t recon_offset,
                                                uint32_t recon_stride, uint32_t area_width, uint32_t area_height);
uint64_t svt_full_distortion_kernel16_bits_hip(uint8_t *input, uint32_t input_offset, uint32_t input_stride, uint8_t *recon, int32_t recon_offset,
                                               uint32_t recon_stride, uint32_t area_width, uint32_t area_height);

/* PSYEX facades (common_dsp_rtcd.h:165-166; bodies C_DEFAULT/picture_operators_c.c:85-174).  PredictionMode and CompoundType are
 * ATTRIBUTE_PACKED (one byte) enums in the reference (definitions.h:1126,1197): uint8_t here is the same ABI. */
uint64_t svt_spatial_psy_distortion_kernel_hip(uint8_t *input, uint32_t input_offset, uint32_t input_stride, uint8_t *recon, int32_t recon_offset,
                                               uint32_t recon_stride, uint32_t area_width, uint32_t area_height, double psy_rd);
uint64_t svt_spatial_full_distortion_kernel_facade_hip(uint8_t *input, uint32_t input_offset, uint32_t input_stride, uint8_t *recon,
                                                       int32_t recon_offset, uint32_t recon_stride, uint32_t area_width, uint32_t area_height,
                                                       bool hbd_md, uint8_t mode, uint8_t compound_type, uint8_t temporal_layer_index,
                                                       double psy_rd, uint8_t spy_rd);

/* svt_aom_hadamard_NxN (common_dsp_rtcd.h:1075-1085), svt_aom_satd (aom_dsp_rtcd.h:209) */
void svt_aom_hadamard_4x4_hip(const int16_t *src_diff, ptrdiff_t src_stride, int32_t *coeff);
void svt_aom_hadamard_8x8_hip(const int16_t *src_diff, ptrdiff_t src_stride, int32_t *coeff);
void svt_aom_hadamard_16x16_hip(const int16_t *src_diff, ptrdiff_t src_stride, int32_t *coeff);
void svt_aom_hadamard_32x32_hip(const int16_t *src_diff, ptrdiff_t src_stride, int32_t *coeff);
int  svt_aom_satd_hip(const int32_t *coeff, int length);
/* svt_av1_compute_cul_level (aom_dsp_rtcd.h:904, Codec/full_loop.c:1449-1466), svt_av1_fwht4x4 (aom_dsp_rtcd.h:208, Codec/transforms.c:3099-3151) */
uint8_t svt_av1_compute_cul_level_hip(const int16_t *const scan, const int32_t *const quant_coeff, uint16_t *eob);
void    svt_av1_fwht4x4_hip(int16_t *input, int32_t *output, uint32_t stride);
/* hadamard_path_c (Codec/enc_mode_config.c:2151-2217) with the Buf2D arguments flattened: 8-bit input and prediction,
 * square block of `block_size_wide` (4..128) */
uint32_t svt_hip_hadamard_path(const uint8_t *input, uint32_t input_stride, const uint8_t *pred, uint32_t pred_stride, uint32_t block_size_wide);

/* svt_aom_sad{W}x{H} / svt_aom_sad{W}x{H}x4d (aom_dsp_rtcd.h:267-347; C_DEFAULT/compute_sad_c.c:117-207) */
#define SVT_HIP_DECL_SAD(W, H)                                                                                          \
    uint32_t svt_aom_sad##W##x##H##_hip(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride);          \
    void svt_aom_sad##W##x##H##x4d_hip(const uint8_t *src, int src_stride, const uint8_t *const ref[], int ref_stride, uint32_t *sad_array);
SVT_HIP_DECL_SAD(4, 4) SVT_HIP_DECL_SAD(4, 8) SVT_HIP_DECL_SAD(4, 16) SVT_HIP_DECL_SAD(8, 4) SVT_HIP_DECL_SAD(8, 8) SVT_HIP_DECL_SAD(8, 16)
SVT_HIP_DECL_SAD(8, 32) SVT_HIP_DECL_SAD(16, 4) SVT_HIP_DECL_SAD(16, 8) SVT_HIP_DECL_SAD(16, 16) SVT_HIP_DECL_SAD(16, 32) SVT_HIP_DECL_SAD(16, 64)
SVT_HIP_DECL_SAD(32, 8) SVT_HIP_DECL_SAD(32, 16) SVT_HIP_DECL_SAD(32, 32) SVT_HIP_DECL_SAD(32, 64) SVT_HIP_DECL_SAD(64, 16) SVT_HIP_DECL_SAD(64, 32)
SVT_HIP_DECL_SAD(64, 64) SVT_HIP_DECL_SAD(64, 128) SVT_HIP_DECL_SAD(128, 64) SVT_HIP_DECL_SAD(128, 128)
#undef SVT_HIP_DECL_SAD
/* svt_aom_variance_highbd_c (C_DEFAULT/variance.c:278-296) */
uint32_t svt_aom_variance_highbd_hip(const uint16_t *a, int a_stride, const uint16_t *b, int b_stride, int w, int h, uint32_t *sse);
/* svt_aom_highbd_10_variance{W}x{H} (aom_dsp_rtcd.h:546-568, bodies Codec/svt_psnr.c:139-177): src / ref are CONVERT_TO_BYTEPTR'd
 * uint16_t pointers, as the reference's callers pass them (av1me.c:31-172) */
#define SVT_HIP_DECL_VAR10(W, H) unsigned int svt_aom_highbd_10_variance##W##x##H##_hip(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride, unsigned int *sse);
SVT_HIP_DECL_VAR10(4, 4) SVT_HIP_DECL_VAR10(4, 8) SVT_HIP_DECL_VAR10(4, 16) SVT_HIP_DECL_VAR10(8, 4) SVT_HIP_DECL_VAR10(8, 8) SVT_HIP_DECL_VAR10(8, 16)
SVT_HIP_DECL_VAR10(8, 32) SVT_HIP_DECL_VAR10(16, 4) SVT_HIP_DECL_VAR10(16, 8) SVT_HIP_DECL_VAR10(16, 16) SVT_HIP_DECL_VAR10(16, 32) SVT_HIP_DECL_VAR10(16, 64)
SVT_HIP_DECL_VAR10(32, 8) SVT_HIP_DECL_VAR10(32, 16) SVT_HIP_DECL_VAR10(32, 32) SVT_HIP_DECL_VAR10(32, 64) SVT_HIP_DECL_VAR10(64, 16) SVT_HIP_DECL_VAR10(64, 32)
SVT_HIP_DECL_VAR10(64, 64) SVT_HIP_DECL_VAR10(64, 128) SVT_HIP_DECL_VAR10(128, 64) SVT_HIP_DECL_VAR10(128, 128)
#undef SVT_HIP_DECL_VAR10

/* coefficient-domain distortion (common_dsp_rtcd.h:160-161, aom_dsp_rtcd.h:212), residual (common_dsp_rtcd.h:157,170) */
void    svt_full_distortion_kernel32_bits_hip(int32_t *coeff, uint32_t coeff_stride, int32_t *recon_coeff, uint32_t recon_coeff_stride,
                                              uint64_t distortion_result[2], uint32_t area_width, uint32_t area_height);
void    svt_full_distortion_kernel_cbf_zero32_bits_hip(int32_t *coeff, uint32_t coeff_stride, uint64_t distortion_result[2], uint32_t area_width, uint32_t area_height);
int64_t svt_av1_block_error_hip(const int32_t *coeff, const int32_t *dqcoeff, intptr_t block_size, int64_t *ssz);
void    svt_residual_kernel8bit_hip(uint8_t *input, uint32_t input_stride, uint8_t *pred, uint32_t pred_stride, int16_t *residual, uint32_t residual_stride,
                                    uint32_t area_width, uint32_t area_height);
void    svt_residual_kernel16bit_hip(uint16_t *input, uint32_t input_stride, uint16_t *pred, uint32_t pred_stride, int16_t *residual, uint32_t residual_stride,
                                     uint32_t area_width, uint32_t area_height);
/* svt_aom_estimate_transform (Codec/transforms.c:3158-3225) minus its pcs / ctx arguments: residual -> packed coefficients
 * (min(W,32) x min(H,32)) and the energy of the discarded frequencies; pf_shape 0..3 = DEFAULT / N2 / N4 / ONLY_DC */
int svt_hip_estimate_transform(int16_t *residual, uint32_t residual_stride, int32_t *coeff, int tx_size, uint64_t *three_quad_energy, int tx_type, int pf_shape);

/* PSYEX psy-RD term (Codec/psy_rd.h:23-33): integer energies, and the fp64-scaled form */
uint64_t svt_psy_distortion_hip(const uint8_t *input, uint32_t input_stride, const uint8_t *recon, uint32_t recon_stride, uint32_t width, uint32_t height);
uint64_t svt_psy_distortion_hbd_hip(const uint16_t *input, uint32_t input_stride, const uint16_t *recon, uint32_t recon_stride, uint32_t width, uint32_t height);
uint64_t get_svt_psy_full_dist_hip(const void *s, uint32_t so, uint32_t sp, const void *r, uint32_t ro, uint32_t rp, uint32_t w, uint32_t h, uint8_t is_hbd,
                                   double psy_rd);

#ifdef __cplusplus
}
#endif
#endif /* SVT_HIP_LEAF_H */
