/* svt_hip_spy_rd.h -- the PSYEX "spy-rd" mode biases of svt_spatial_full_distortion_kernel_facade
 * (reference: Source/Lib/C_DEFAULT/picture_operators_c.c:130-171) as one inline function shared by the host entry and
 * the device kernel (the oracle keeps its own restatement: oracle/stats_oracle.c).  Enumerator values: Codec/definitions.h:1126-1162 (PredictionMode),
 * :1197-1202 (CompoundType). */
#ifndef SVT_HIP_SPY_RD_H
#define SVT_HIP_SPY_RD_H
#include <stdint.h>

#ifdef __HIPCC__
#define SVT_HIP_HD __host__ __device__
#else
#define SVT_HIP_HD
#endif

enum {
    SVT_HIP_DC_PRED = 0, SVT_HIP_V_PRED = 1, SVT_HIP_H_PRED = 2, SVT_HIP_SMOOTH_PRED = 9, SVT_HIP_SMOOTH_V_PRED = 10,
    SVT_HIP_SMOOTH_H_PRED = 11, SVT_HIP_PAETH_PRED = 12, SVT_HIP_INTRA_MODE_END = 13 /* NEARESTMV */,
    SVT_HIP_COMP_INTER_MODE_START = 17 /* NEAREST_NEARESTMV */, SVT_HIP_COMP_INTER_MODE_END = 25 /* MB_MODE_COUNT */,
    SVT_HIP_COMPOUND_AVERAGE = 0, SVT_HIP_COMPOUND_DISTWTD = 1, SVT_HIP_COMPOUND_WEDGE = 2, SVT_HIP_COMPOUND_DIFFWTD = 3
};

static inline SVT_HIP_HD int64_t svt_hip_spy_rd_bias_inline(int64_t dist, uint32_t area_width, uint32_t area_height, uint8_t mode,
                                                            uint8_t compound_type, uint8_t temporal_layer_index, double psy_rd,
                                                            uint8_t spy_rd) {
    if (spy_rd != 1) return dist; /* "only enable the tweaks when full spy-rd is active" */
    if (mode == SVT_HIP_DC_PRED || mode == SVT_HIP_SMOOTH_PRED || mode == SVT_HIP_SMOOTH_V_PRED || mode == SVT_HIP_SMOOTH_H_PRED) {
        if (psy_rd == 0.0) dist = (dist * 5) / 4;
    } else if (mode == SVT_HIP_H_PRED || mode == SVT_HIP_V_PRED || mode == SVT_HIP_PAETH_PRED) {
        dist = (dist * 9) / 8;
    } else if (mode >= SVT_HIP_COMP_INTER_MODE_START && mode < SVT_HIP_COMP_INTER_MODE_END) {
        if (compound_type == SVT_HIP_COMPOUND_AVERAGE || compound_type == SVT_HIP_COMPOUND_DISTWTD) dist = (dist * 5) / 4;
        else if (compound_type == SVT_HIP_COMPOUND_DIFFWTD) dist = (dist * 9) / 8;
    }
    if (mode < SVT_HIP_INTRA_MODE_END) {
        if (temporal_layer_index >= 2) dist = (dist * (int64_t)(7 + (temporal_layer_index > 5 ? 5 : temporal_layer_index))) / 8; /* weights {8,8,9,10,11,12} */
        if (area_width == 64 && area_height == 64) dist = (dist * 3) / 2;
        else if (area_width * area_height <= 32 * 32) dist = (dist * 17) / 16;
    }
    return dist;
}
#endif /* SVT_HIP_SPY_RD_H */
