/*
 * txfm_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement of the reference's integer AV1 forward / inverse 2-D
 * transforms (Codec/transforms.c:50-2631, Codec/inv_transforms.c:94-2716).  Parity checker only; pinned against
 * the reference's svt_av1_transform_two_d_* / svt_av1_fwd_txfm2d_* / svt_av1_inv_txfm2d_add_* `_c` functions for
 * all 19 sizes x the types each allows in tests/test_dsp_oracle_vs_ref.py (where the reference build exists), and
 * through the RD chain's fixture tests/golden/rd_chain.npz (tests/test_rd_golden.py: outputs of the reference's `_c`
 * chain for all 19 sizes, which travel to the GPU box).
 *
 * The reference spells every butterfly network out stage by stage.  Here the same flow graphs are expressed by
 * their structure: a DCT of size N is one add/sub butterfly, a DCT of size N/2 on the sums and an "odd part" on
 * the differences (alternating rotation and butterfly stages whose angles follow a bit-reversal rule), followed
 * by a bit-reversal permutation; the inverse runs the transposed graph with the reference's clamps.  Every
 * rotation is the reference's half_btf(): two 32-bit wrapping products, summed in 64 bits, rounded, shifted.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int32_t g_cospi[7][64]; /* cospi_arr(bit)[j] = round(cos(j*pi/128) * 2^bit), bit = 10..16 */
static int     g_init = 0;
/* svt_aom_eb_av1_sinpi_arr_data (Codec/inv_transforms.c:3228-3234): AV1 ADST4 constants per cos_bit 10..16 */
static const int32_t k_sinpi[7][5] = {{0, 330, 621, 836, 951},       {0, 660, 1241, 1672, 1901},     {0, 1321, 2482, 3344, 3803},
                                      {0, 2642, 4964, 6689, 7606},   {0, 5283, 9929, 13377, 15212}, {0, 10566, 19858, 26755, 30424},
                                      {0, 21133, 39716, 53510, 60849}};

extern double cos(double);
static void   init_tables(void) {
    if (g_init) return;
    for (int b = 0; b < 7; b++)
        for (int j = 0; j < 64; j++) {
            const double v = cos(3.14159265358979323846 * j / 128.0) * (double)(1 << (10 + b));
            g_cospi[b][j]  = (int32_t)(v + 0.5);
        }
    g_init = 1;
}
const int32_t *orc_cospi(int bit) { init_tables(); return g_cospi[bit - 10]; }

static inline int32_t round_shift64(int64_t v, int bit) { return (int32_t)((v + ((int64_t)1 << (bit - 1))) >> bit); }

/* half_btf, Codec/inv_transforms.h:264-285 */
static inline int32_t hbtf(int32_t w0, int32_t a, int32_t w1, int32_t b, int bit) {
    const int64_t s = (int64_t)(int32_t)((uint32_t)w0 * (uint32_t)a) + (int64_t)(int32_t)((uint32_t)w1 * (uint32_t)b);
    return (int32_t)((s + ((int64_t)1 << (bit - 1))) >> bit);
}
/* clamp_value, Codec/inv_transforms.c:86-92 */
static inline int32_t clampv(int64_t v, int bit) {
    if (bit <= 0) return (int32_t)v;
    const int64_t hi = ((int64_t)1 << (bit - 1)) - 1, lo = -((int64_t)1 << (bit - 1));
    return (int32_t)(v < lo ? lo : (v > hi ? hi : v));
}
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static int brev(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}
static int ilog2(int n) { int l = 0; while ((1 << l) < n) l++; return l; }

/* ------------------------------------------------------------------------------------------------
 * DCT.  x[M .. 2M) is the odd part of a 2M-point DCT.  clamp_bit < 0: forward (no clamps).
 * ------------------------------------------------------------------------------------------------ */
/* rotation stage k of the odd part (k = 1 .. log2(M)-1); identical in both directions (symmetric 2x2 blocks) */
static void odd_rot(int32_t *x, int M, int k, const int32_t *c, int bit) {
    const int t = M >> k; /* rotated lanes sit in groups of 2t, in the middle half of each group */
    if (k == 1) {
        for (int j = 0; j < M / 4; j++) {
            const int p = M + M / 4 + j, m = 3 * M - 1 - p;
            const int32_t a = x[p], b = x[m];
            x[p] = hbtf(-c[32], a, c[32], b, bit);
            x[m] = hbtf(c[32], b, c[32], a, bit);
        }
        return;
    }
    const int groups = 1 << (k - 2); /* groups in the lower half; their mirror images form the upper half */
    for (int g = 0; g < groups; g++) {
        const int A = (1 + 4 * brev(g, k - 2)) * (64 >> k), B = 64 - A, base = M + g * 2 * t;
        for (int j = 0; j < t / 2; j++) { /* first half of the middle */
            const int p = base + t / 2 + j, m = 3 * M - 1 - p;
            const int32_t a = x[p], b = x[m];
            x[p] = hbtf(-c[A], a, c[B], b, bit);
            x[m] = hbtf(c[A], b, c[B], a, bit);
        }
        for (int j = 0; j < t / 2; j++) { /* second half of the middle */
            const int p = base + t + j, m = 3 * M - 1 - p;
            const int32_t a = x[p], b = x[m];
            x[p] = hbtf(-c[B], a, -c[A], b, bit);
            x[m] = hbtf(c[B], b, -c[A], a, bit);
        }
    }
}
/* butterfly stage k of the odd part: groups of t = M >> k, even groups sum-first, odd groups difference-first */
static void odd_bfly(int32_t *x, int M, int k, int clamp_bit) {
    const int t = M >> k;
    for (int g = 0; g < (M / t); g++) {
        const int b = M + g * t;
        for (int j = 0; j < t / 2; j++) {
            const int32_t lo = x[b + j], hi = x[b + t - 1 - j];
            int32_t s = wadd(lo, hi), d = (g & 1) ? wsub(hi, lo) : wsub(lo, hi);
            if (clamp_bit >= 0) {
                s = clampv((int64_t)lo + hi, clamp_bit);
                d = clampv((g & 1) ? (int64_t)hi - lo : (int64_t)lo - hi, clamp_bit);
            }
            if (g & 1) { x[b + j] = d; x[b + t - 1 - j] = s; }
            else       { x[b + j] = s; x[b + t - 1 - j] = d; }
        }
    }
}
static void odd_final(int32_t *x, int M, const int32_t *c, int bit, int inverse) {
    const int L = ilog2(M);
    for (int i = 0; i < M / 2; i++) {
        const int A = 64 - (2 * brev(i, L) + 1) * (32 / M), B = 64 - A, p = M + i, m = 2 * M - 1 - i;
        const int32_t a = x[p], b = x[m];
        if (!inverse) {
            x[p] = hbtf(c[A], a, c[B], b, bit);
            x[m] = hbtf(c[A], b, -c[B], a, bit);
        } else {
            x[p] = hbtf(c[A], a, -c[B], b, bit);
            x[m] = hbtf(c[B], a, c[A], b, bit);
        }
    }
}
static void fdct_core(int32_t *x, int N, const int32_t *c, int bit) {
    if (N == 2) {
        const int32_t a = x[0], b = x[1];
        x[0] = hbtf(c[32], a, c[32], b, bit);
        x[1] = hbtf(-c[32], b, c[32], a, bit);
        return;
    }
    const int M = N / 2, L = ilog2(M);
    for (int i = 0; i < M; i++) {
        const int32_t a = x[i], b = x[N - 1 - i];
        x[i]         = wadd(a, b);
        x[N - 1 - i] = wsub(a, b);
    }
    fdct_core(x, M, c, bit);
    for (int k = 1; k < L; k++) {
        odd_rot(x, M, k, c, bit);
        odd_bfly(x, M, k, -1);
    }
    odd_final(x, M, c, bit, 0);
}
static void idct_core(int32_t *x, int N, const int32_t *c, int bit, int clamp_bit) {
    if (N == 2) {
        const int32_t a = x[0], b = x[1];
        x[0] = hbtf(c[32], a, c[32], b, bit);
        x[1] = hbtf(c[32], a, -c[32], b, bit);
        return;
    }
    const int M = N / 2, L = ilog2(M);
    odd_final(x, M, c, bit, 1);
    for (int k = L - 1; k >= 1; k--) {
        odd_bfly(x, M, k, clamp_bit);
        odd_rot(x, M, k, c, bit);
    }
    idct_core(x, M, c, bit, clamp_bit);
    for (int i = 0; i < M; i++) {
        const int32_t a = x[i], b = x[N - 1 - i];
        x[i]         = clampv((int64_t)a + b, clamp_bit);
        x[N - 1 - i] = clampv((int64_t)a - b, clamp_bit);
    }
}
static void fdct(const int32_t *in, int32_t *out, int N, int bit) {
    int32_t x[64];
    const int32_t *c = orc_cospi(bit);
    memcpy(x, in, sizeof(int32_t) * N);
    fdct_core(x, N, c, bit);
    for (int k = 0; k < N; k++) out[k] = x[brev(k, ilog2(N))];
}
static void idct(const int32_t *in, int32_t *out, int N, int bit, int clamp_bit) {
    int32_t x[64];
    const int32_t *c = orc_cospi(bit);
    for (int k = 0; k < N; k++) x[brev(k, ilog2(N))] = in[k];
    idct_core(x, N, c, bit, clamp_bit);
    memcpy(out, x, sizeof(int32_t) * N);
}

/* ------------------------------------------------------------------------------------------------
 * ADST 8 / 16 (svt_av1_fadst8_new, fadst16_new, iadst8_new, iadst16_new) and ADST 4
 * ------------------------------------------------------------------------------------------------ */
static void adst_perm(int N, int *P) { /* P_N[2j] = P_{N/2}[j], P_N[2j+1] = N-1-P_{N/2}[j], P_2 = {0,1} */
    if (N == 2) { P[0] = 0; P[1] = 1; return; }
    int Q[32];
    adst_perm(N / 2, Q);
    for (int j = 0; j < N / 2; j++) { P[2 * j] = Q[j]; P[2 * j + 1] = N - 1 - Q[j]; }
}
/* rotation acting on the upper half of every group of 2h lanes (h = 2, 4, 8, ...) */
static void adst_rot(int32_t *x, int N, int h, const int32_t *c, int bit) {
    for (int b = 0; b < N; b += 2 * h) {
        if (h == 2) {
            const int32_t a = x[b + 2], d = x[b + 3];
            x[b + 2] = hbtf(c[32], a, c[32], d, bit);
            x[b + 3] = hbtf(c[32], a, -c[32], d, bit);
            continue;
        }
        for (int j = 0; j < h / 4; j++) {
            const int A = (4 * j + 1) * (64 / h), B = 64 - A;
            int p = b + h + 2 * j;
            int32_t a = x[p], d = x[p + 1];
            x[p]     = hbtf(c[A], a, c[B], d, bit);
            x[p + 1] = hbtf(c[B], a, -c[A], d, bit);
            p = b + h + h / 2 + 2 * j;
            a = x[p]; d = x[p + 1];
            x[p]     = hbtf(-c[B], a, c[A], d, bit);
            x[p + 1] = hbtf(c[A], a, c[B], d, bit);
        }
    }
}
static void adst_bfly(int32_t *x, int N, int h, int clamp_bit) { /* x[j] +- x[j+h] inside groups of 2h */
    for (int b = 0; b < N; b += 2 * h)
        for (int j = 0; j < h; j++) {
            const int32_t a = x[b + j], d = x[b + j + h];
            if (clamp_bit >= 0) { x[b + j] = clampv((int64_t)a + d, clamp_bit); x[b + j + h] = clampv((int64_t)a - d, clamp_bit); }
            else                { x[b + j] = wadd(a, d); x[b + j + h] = wsub(a, d); }
        }
}
static void adst_last(int32_t *x, int N, const int32_t *c, int bit) {
    for (int j = 0; j < N / 2; j++) {
        const int A = (4 * j + 1) * (64 / (2 * N)), B = 64 - A;
        const int32_t a = x[2 * j], d = x[2 * j + 1];
        x[2 * j]     = hbtf(c[A], a, c[B], d, bit);
        x[2 * j + 1] = hbtf(c[B], a, -c[A], d, bit);
    }
}
static void fadst(const int32_t *in, int32_t *out, int N, int bit) {
    int32_t x[16]; int P[16];
    const int32_t *c = orc_cospi(bit);
    adst_perm(N, P);
    for (int k = 0; k < N; k++) x[k] = (__builtin_popcount(k) & 1) ? (int32_t)(0u - (uint32_t)in[P[k]]) : in[P[k]];
    for (int h = 2; h < N; h *= 2) { adst_rot(x, N, h, c, bit); adst_bfly(x, N, h, -1); }
    adst_last(x, N, c, bit);
    for (int j = 0; j < N / 2; j++) { out[2 * j] = x[2 * j + 1]; out[2 * j + 1] = x[N - 2 - 2 * j]; }
}
static void iadst(const int32_t *in, int32_t *out, int N, int bit, int clamp_bit) {
    int32_t x[16]; int P[16];
    const int32_t *c = orc_cospi(bit);
    adst_perm(N, P);
    for (int j = 0; j < N / 2; j++) { x[2 * j + 1] = in[2 * j]; x[N - 2 - 2 * j] = in[2 * j + 1]; }
    adst_last(x, N, c, bit);
    for (int h = N / 2; h >= 2; h /= 2) { adst_bfly(x, N, h, clamp_bit); adst_rot(x, N, h, c, bit); }
    for (int k = 0; k < N; k++) out[P[k]] = (__builtin_popcount(k) & 1) ? (int32_t)(0u - (uint32_t)x[k]) : x[k];
}
/* svt_av1_fadst4_new (transforms.c:1415-1503) / svt_av1_iadst4_new (inv_transforms.c:722-806): 32-bit wrapping */
#define MUL(a, b) ((int32_t)((uint32_t)(a) * (uint32_t)(b)))
static void fadst4(const int32_t *in, int32_t *out, int bit) {
    const int32_t *s = k_sinpi[bit - 10];
    const int32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3];
    if (!(x0 | x1 | x2 | x3)) { out[0] = out[1] = out[2] = out[3] = 0; return; }
    const int32_t s7 = wsub(wadd(x0, x1), x3);
    const int32_t a0 = wadd(wadd(MUL(s[1], x0), MUL(s[2], x1)), MUL(s[4], x3));
    const int32_t a1 = MUL(s[3], s7);
    const int32_t a2 = wadd(wsub(MUL(s[4], x0), MUL(s[1], x1)), MUL(s[2], x3));
    const int32_t a3 = MUL(s[3], x2);
    out[0] = round_shift64(wadd(a0, a3), bit);
    out[1] = round_shift64(a1, bit);
    out[2] = round_shift64(wsub(a2, a3), bit);
    out[3] = round_shift64(wadd(wsub(a2, a0), a3), bit);
}
static void iadst4(const int32_t *in, int32_t *out, int bit) {
    const int32_t *s = k_sinpi[bit - 10];
    const int32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3];
    if (!(x0 | x1 | x2 | x3)) { out[0] = out[1] = out[2] = out[3] = 0; return; }
    const int32_t s7 = wadd(wsub(x0, x2), x3);
    const int32_t a0 = wadd(wadd(MUL(s[1], x0), MUL(s[4], x2)), MUL(s[2], x3));
    const int32_t a1 = wsub(wsub(MUL(s[2], x0), MUL(s[1], x2)), MUL(s[4], x3));
    const int32_t a3 = MUL(s[3], x1), a2 = MUL(s[3], s7);
    out[0] = round_shift64(wadd(a0, a3), bit);
    out[1] = round_shift64(wadd(a1, a3), bit);
    out[2] = round_shift64(a2, bit);
    out[3] = round_shift64(wsub(wadd(a0, a1), a3), bit);
}
/* identity kernels: transforms.c:2205-2236, inv_transforms.c:2331-2363 */
static void identity(const int32_t *in, int32_t *out, int N) {
    for (int i = 0; i < N; i++) switch (N) {
        case 4: out[i] = round_shift64((int64_t)in[i] * 5793, 12); break;
        case 8: out[i] = (int32_t)((uint32_t)in[i] * 2u); break;
        case 16: out[i] = round_shift64((int64_t)in[i] * 2 * 5793, 12); break;
        case 32: out[i] = (int32_t)((uint32_t)in[i] * 4u); break;
        default: out[i] = round_shift64((int64_t)in[i] * 4 * 5793, 12); break;
        }
}

/* 1-D kernel selector: type 0 DCT, 1 ADST, 2 FLIPADST (same kernel; flips are applied in 2-D), 3 identity */
static void fwd_1d(const int32_t *in, int32_t *out, int N, int type, int bit) {
    if (type == 3) identity(in, out, N);
    else if (type == 0) fdct(in, out, N, bit);
    else if (N == 4) fadst4(in, out, bit);
    else fadst(in, out, N, bit);
}
static void inv_1d(const int32_t *in, int32_t *out, int N, int type, int bit, int clamp_bit) {
    if (type == 3) identity(in, out, N);
    else if (type == 0) idct(in, out, N, bit, clamp_bit);
    else if (N == 4) iadst4(in, out, bit);
    else iadst(in, out, N, bit, clamp_bit);
}

/* ------------------------------------------------------------------------------------------------
 * 2-D configuration (svt_aom_transform_config transforms.c:2344-2360, svt_av1_get_inv_txfm_cfg
 * inv_transforms.c:2436-2458) and cores (av1_tranform_two_d_core_c :2259-2324, inv_txfm2d_add_c :2459-2535)
 * ------------------------------------------------------------------------------------------------ */
static const uint8_t k_tx_w[19] = {4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64};
static const uint8_t k_tx_h[19] = {4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16};
/* column (vertical) / row (horizontal) 1-D type of each TxType: vtx_tab / htx_tab, inv_transforms.h:52-87 */
static const uint8_t k_vtx[16] = {0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3};
static const uint8_t k_htx[16] = {0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2};
static const int8_t  k_fwd_shift[19][3] = {{2, 0, 0},   {2, -1, 0},  {2, -2, 0},  {2, -4, 0},  {0, -2, -2}, {2, -1, 0}, {2, -1, 0},
                                           {2, -2, 0},  {2, -2, 0},  {2, -4, 0},  {2, -4, 0},  {0, -2, -2}, {2, -4, -2}, {2, -1, 0},
                                           {2, -1, 0},  {2, -2, 0},  {2, -2, 0},  {0, -2, 0},  {2, -4, 0}};
static const int8_t  k_fwd_cos_col[5][5] = {{13, 13, 13, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 13, 12, 13}, {0, 13, 13, 12, 13}, {0, 0, 13, 12, 13}};
static const int8_t  k_fwd_cos_row[5][5] = {{13, 13, 12, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 12, 13, 12}, {0, 12, 13, 12, 11}, {0, 0, 12, 11, 10}};
static const int8_t  k_inv_shift0[19] = {0, -1, -2, -2, -2, 0, 0, -1, -1, -1, -1, -1, -1, -1, -1, -2, -2, -2, -2}; /* [1] is always -4 */

int orc_tx_size_wide(int tx_size) { return k_tx_w[tx_size]; }
int orc_tx_size_high(int tx_size) { return k_tx_h[tx_size]; }

static void shift_arr(int32_t *a, int n, int sh) { /* svt_av1_round_shift_array_c(arr, n, -sh) */
    if (sh == 0) return;
    if (sh < 0) for (int i = 0; i < n; i++) a[i] = round_shift64(a[i], -sh);
    else for (int i = 0; i < n; i++) a[i] = (int32_t)((uint32_t)a[i] * (1u << sh));
}

/* svt_av1_transform_two_d_* / svt_av1_fwd_txfm2d_*_c: full W x H int32 output (row-major, stride W) */
void orc_fwd_txfm2d(const int16_t *input, int32_t *output, uint32_t stride, int tx_type, int tx_size) {
    const int W = k_tx_w[tx_size], H = k_tx_h[tx_size], wi = ilog2(W) - 2, hi = ilog2(H) - 2;
    const int vt = k_vtx[tx_type], ht = k_htx[tx_type], ud = (vt == 2), lr = (ht == 2);
    const int cb_col = k_fwd_cos_col[wi][hi], cb_row = k_fwd_cos_row[wi][hi];
    const int8_t *sh = k_fwd_shift[tx_size];
    int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * W * H), tin[64], tout[64];
    for (int c = 0; c < W; c++) {
        for (int r = 0; r < H; r++) tin[r] = input[(ud ? H - 1 - r : r) * stride + c];
        shift_arr(tin, H, sh[0]);
        fwd_1d(tin, tout, H, vt, cb_col);
        shift_arr(tout, H, sh[1]);
        for (int r = 0; r < H; r++) buf[r * W + (lr ? W - 1 - c : c)] = tout[r];
    }
    const int rect = (W == 2 * H || H == 2 * W);
    for (int r = 0; r < H; r++) {
        fwd_1d(buf + r * W, output + r * W, W, ht, cb_row);
        shift_arr(output + r * W, W, sh[2]);
        if (rect) for (int c = 0; c < W; c++) output[r * W + c] = round_shift64((int64_t)output[r * W + c] * 5793, 12);
    }
    free(buf);
}

/* svt_av1_inv_txfm2d_add_{WxH}_c: `input` holds min(W,32) x min(H,32) coefficients (the reference's packed
 * layout for 64-point sizes); recon = clip(pred + residual) in `bd` bits, read and written as uint16. */
void orc_inv_txfm2d_add(const int32_t *input, const uint16_t *out_r, int32_t stride_r, uint16_t *out_w, int32_t stride_w, int tx_type,
                        int tx_size, int bd) {
    const int W = k_tx_w[tx_size], H = k_tx_h[tx_size], Wp = W > 32 ? 32 : W, Hp = H > 32 ? 32 : H;
    const int vt = k_vtx[tx_type], ht = k_htx[tx_type], ud = (vt == 2), lr = (ht == 2);
    const int row_clamp = bd == 8 ? 16 : (bd == 10 ? 18 : 20), col_clamp = bd == 12 ? 18 : 16; /* svt_av1_gen_inv_stage_range */
    const int rect = (W == 2 * H || H == 2 * W);
    int32_t *buf = (int32_t *)calloc((size_t)W * H, sizeof(int32_t)), tin[64], tout[64];
    for (int r = 0; r < H; r++) {
        for (int c = 0; c < W; c++) {
            const int32_t v = (r < Hp && c < Wp) ? input[r * Wp + c] : 0;
            tin[c] = rect ? round_shift64((int64_t)v * 2896, 12) : v;
            tin[c] = clampv(tin[c], bd + 8);
        }
        inv_1d(tin, buf + r * W, W, ht, 12, row_clamp);
        shift_arr(buf + r * W, W, k_inv_shift0[tx_size]);
    }
    for (int c = 0; c < W; c++) {
        for (int r = 0; r < H; r++) tin[r] = clampv(buf[r * W + (lr ? W - 1 - c : c)], bd + 6 > 16 ? bd + 6 : 16);
        inv_1d(tin, tout, H, vt, 12, col_clamp);
        shift_arr(tout, H, -4);
        for (int r = 0; r < H; r++) {
            const int64_t v = (int64_t)out_r[r * stride_r + c] + tout[ud ? H - 1 - r : r];
            const int64_t hi = (1 << bd) - 1;
            out_w[r * stride_w + c] = (uint16_t)(v < 0 ? 0 : (v > hi ? hi : v));
        }
    }
    free(buf);
}

/* svt_handle_transform{64x64,64x32,32x64,64x16,16x64}_c (transforms.c:2374-2505): energy of the discarded
 * frequencies, then repack the kept min(W,32) x min(H,32) block contiguously.  No-op (returns 0) for other sizes. */
uint64_t orc_handle_transform(int32_t *coeff, int tx_size) {
    const int W = k_tx_w[tx_size], H = k_tx_h[tx_size], Wp = W > 32 ? 32 : W, Hp = H > 32 ? 32 : H;
    uint64_t e = 0;
    if (W <= 32 && H <= 32) return 0;
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++)
            if (r >= Hp || c >= Wp) e += (uint64_t)((int64_t)coeff[r * W + c] * coeff[r * W + c]);
    if (Wp != W)
        for (int r = 1; r < Hp; r++) memmove(coeff + r * Wp, coeff + r * W, sizeof(int32_t) * Wp);
    return e;
}

/* Scan orders (av1_scan_orders[tx_size][tx_type], Codec/coefficients.h:2197): positions index the kept
 * min(W,32) x min(H,32) block, row-major.  2-D types and IDTX use the diagonal scan (zig-zag for square blocks,
 * single-direction diagonals for rectangular ones), V_* types the row scan, H_* types the column scan. */
int orc_scan_order(int tx_size, int tx_type, int16_t *scan, int16_t *iscan) {
    int w = k_tx_w[tx_size] > 32 ? 32 : k_tx_w[tx_size], h = k_tx_h[tx_size] > 32 ? 32 : k_tx_h[tx_size];
    /* 64-point sizes reuse the scan of the 32-capped shape; 16x64 / 64x16 use 16x32 / 32x16 */
    const int n = w * h;
    int k = 0;
    if (tx_type >= 10 && (tx_type & 1) == 0) { /* V_DCT, V_ADST, V_FLIPADST: row by row */
        for (int i = 0; i < n; i++) scan[k++] = (int16_t)i;
    } else if (tx_type >= 11) { /* H_*: column by column */
        for (int c = 0; c < w; c++)
            for (int r = 0; r < h; r++) scan[k++] = (int16_t)(r * w + c);
    } else {
        for (int d = 0; d < w + h - 1; d++) {
            /* direction of travel along anti-diagonal d: tall blocks top->bottom, wide blocks bottom->top,
             * square blocks alternate (odd diagonals top->bottom) */
            const int down = (w < h) ? 1 : (w > h) ? 0 : (d & 1);
            for (int i = 0; i <= d; i++) {
                const int r = down ? i : d - i, c = d - r;
                if (r < h && c < w) scan[k++] = (int16_t)(r * w + c);
            }
        }
    }
    for (int i = 0; i < n; i++) iscan[scan[i]] = (int16_t)i;
    return n;
}
