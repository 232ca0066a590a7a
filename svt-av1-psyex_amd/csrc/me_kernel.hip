// me_kernel.hip -- open-loop motion estimation for one picture on gfx950 (MI355X, CDNA4).
//
// ONE WAVE owns one 64x64 block (b64) at a time and runs the whole per-block pipeline of svt_aom_motion_estimation_b64
// (reference: Source/Lib/Codec/motion_estimation.c:3076-3153) on the device: zero-MV SADs, pre-HME, HME level 0/1/2,
// search-centre selection, reference pruning, the 8x8-based integer search for 85 square PUs, candidate construction and the
// per-block distortion scalars.  Workgroups are single waves: there is no workgroup barrier anywhere in the kernel, the
// stages of a block are ordered by the wave's own program order, and a CU hides the latencies of one block's dependent
// stages (global loads of the search windows, LDS round trips of the control code) behind the 7 other blocks resident on it.
//
// Mapping to the hardware:
//   * persistent waves pull b64 jobs from eight band queues in HBM (one per XCD; a wave drains the queue of the XCD it
//     runs on first, so neighbouring blocks -- whose search windows overlap -- share that XCD's L2), then steal.
//   * every search stages its reference window once into the wave's LDS arena with 16-byte coalesced loads; the next
//     search's window is already in flight (in registers) while the current one is evaluated.  The source block lives in LDS
//     (HME) or in registers (integer search).
//   * SADs use v_qsad_pk_u16_u8: one instruction = 4 search positions x 4 pixels; row-subsampled like the
//     reference (SUB_SAD_SEARCH).  Arg-min keeps the reference's "first minimum in raster order" rule through a
//     (sad, y, x) lexicographic key, reduced across the wave's lanes.
//   * integer search: the wave <-> 4 consecutive search positions, lane <-> one 8x8 PU (quad-tree lane order), 16x16 /
//     32x32 / 64x64 sums by DPP reductions, 85 running bests in registers.
//   * the data-dependent control logic (search-area sizing, early exits, pruning) runs between the parallel phases on
//     state kept in the wave's LDS slice; launch parameters are scalar (SGPR) values.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "me_kernel.h"

// Diagnostic build only (-DSVT_HIP_ME_PROFILE): lane 0 accumulates shader-clock deltas per phase (private array) and adds
// them to queue_head[16 + 2*i] (u64) when the wave retires.  Never defined in the shipped library.
#ifdef SVT_HIP_ME_PROFILE
// the sums live behind the wave's LDS slice (48 x u32, added with ds_add_u32: a profiling point costs one s_memtime and one LDS atomic)
struct Prof { uint32_t *acc; unsigned long long last; int step; };
#define SVT_HIP_ME_PROFILE_LDS (48 * 4)
#define PROF_DECL Prof prof_; prof_.acc = reinterpret_cast<uint32_t *>(g_lds + lay.total); if (threadIdx.x < 48) prof_.acc[threadIdx.x] = 0; prof_.last = __builtin_readcyclecounter(); prof_.step = 0; Prof *prof = &prof_
#define PROF_STEP(s) (prof->step = (s))
#define PROFS(i) PROF((i) + prof->step) /* per-stage slots: 24.. plan, 32.. staging, 40.. evaluation */
#define PROF(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&prof->acc[i], (uint32_t)(t_ - prof->last)); prof->last = t_; } while (0)
#define PROF_FLUSH(ptr) do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); if (threadIdx.x < 48) atomicAdd((unsigned long long *)(ptr) + threadIdx.x, (unsigned long long)prof->acc[threadIdx.x]); } while (0)
#define PROF_PARAM , Prof *prof
#define PROF_ARG , prof
#else
#define PROF_DECL do { } while (0)
#define PROF(i) do { } while (0)
#define PROFS(i) do { } while (0)
#define PROF_STEP(s) do { } while (0)
#define PROF_FLUSH(ptr) do { } while (0)
#define SVT_HIP_ME_PROFILE_LDS 0
#define PROF_PARAM
#define PROF_ARG
#endif

namespace {

constexpr int kThreads  = 64;    // one wave per block
constexpr int kMaxReq   = 32;    // searches per stage (4 HME regions x 8 refs)
constexpr int kWinBytes = SVT_HIP_ME_WIN_BYTES; // LDS window arena of a wave
constexpr int kSrc64Pitch = 80, kSrc32Pitch = 48, kSrc16Pitch = 16; // LDS row pitches of the source views: block rows 2 apart land on different banks
constexpr int kNarrowMaxPos = 32; // searches with at most this many positions are split by block row instead

// z_to_raster, motion_estimation.c:2520-2531: n_idx (quad-tree order) -> raster-within-depth PU index
__device__ const uint8_t c_z_to_raster[85] = {
    0,  1,  2,  3,  4,  5,  6,  9,  10, 7,  8,  11, 12, 13, 14, 17, 18, 15, 16, 19, 20, 21, 22, 29, 30, 23, 24, 31, 32,
    37, 38, 45, 46, 39, 40, 47, 48, 25, 26, 33, 34, 27, 28, 35, 36, 41, 42, 49, 50, 43, 44, 51, 52, 53, 54, 61, 62, 55,
    56, 63, 64, 69, 70, 77, 78, 71, 72, 79, 80, 57, 58, 65, 66, 59, 60, 67, 68, 73, 74, 81, 82, 75, 76, 83, 84};
// me_idx_85_8x8_to_16x16_conversion / me_idx_16x16_to_parent_32x32_conversion, definitions.h:2613-2632
__device__ const uint8_t c_8x8_to_16x16[64] = {5,  5,  6,  6,  7,  7,  8,  8,  5,  5,  6,  6,  7,  7,  8,  8,  9,  9,  10, 10, 11, 11,
                                               12, 12, 9,  9,  10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 13, 13, 14, 14,
                                               15, 15, 16, 16, 17, 17, 18, 18, 19, 19, 20, 20, 17, 17, 18, 18, 19, 19, 20, 20};
__device__ const uint8_t c_16x16_to_32x32[16] = {1, 1, 2, 2, 1, 1, 2, 2, 3, 3, 4, 4, 3, 3, 4, 4};

typedef unsigned long long u64;

// Launch parameters live in the constant address space: uniform, read-only -> scalar loads, values in SGPRs
#define SVT_CONST_AS __attribute__((address_space(4)))
typedef const SVT_CONST_AS MeKernelParams CParams;
typedef const SVT_CONST_AS DevPlane       CPlane;

struct Req { // one svt_sad_loop_kernel call (compute_sad_c.c:58-101)
    const uint8_t *win;   // reference sample of search index (0,0), block row 0
    uint32_t       stride;
    int16_t        sa_w, sa_h;
    uint8_t        bw, bh, rs, level; // block width, effective rows, plane rows per block row, source view
    uint8_t        skip_even, done; // done: the result is already in St.req_key[] (taken from the dense pre-pass): no tile is planned for it
    int16_t        pad1;
};

struct SearchGeo { int ox, oy, sa_w, sa_h; }; // origin (displacement of search index (0, 0)) and size of a search after clipping

// One tile of a stage's searches: a rectangle of a request's search area whose reference window fits the arena.  The tiles of a stage are
// planned once, lane-parallel (plan_tiles), into St.tile[]; the load / store / evaluation passes of a round read their entries back as
// wave-uniform values.
constexpr int kMaxTiles = 40;
struct __attribute__((aligned(16))) TileEnt {
    const uint8_t *g0;       // global address of the window's first sample (search position (x0, y0), block row 0): LDS byte 0 of the tile
    uint32_t stride;
    int16_t  x0, y0;         // sub-area of the request's search area
    int16_t  w, h;
    uint16_t pitch, nvec;    // LDS row pitch (multiple of 16); 16-byte vectors of the window: (pitch / 16) * rows, rows contiguous in LDS
    uint8_t  req, flags;     // request index; level | skip_even << 2 | narrow << 3 | last << 4
    uint8_t  bw, bh;
    uint8_t  rs, kd, km, pad; // 64 = kd * (pitch / 16) + km: how (row, vector) advance from one 64-vector pass to the next
};
static_assert(sizeof(TileEnt) == 32, "TileEnt is read as two 16-byte words");

struct MeReq { // integer search of one reference (open_loop_me_fullpel_search_sblock, motion_estimation.c:781-817)
    const uint8_t *pix0; // reference sample co-located with the block's top-left, MV (0,0)
    uint32_t       stride;
    int16_t        ox, oy, sa_w, sa_h; // search area origin (MV of index (0,0)) and size
    uint8_t        li, ri, probe, pad;
    int16_t        min_x, max_x, min_y, max_y; // displacements of pix0 that stay inside the padded plane
};

struct PreHme {
    uint16_t sa_w, sa_h;
    int16_t  col, row;
    uint32_t sad;
    uint8_t  valid, pad[3];
};

struct St { // per-block state (subset of MeContext, me_context.h:366-509)
    uint32_t org_x, org_y, b64_w, b64_h, b64_index;
    uint32_t zz_sad[2][4];
    uint32_t sr_divisor[2][4];
    uint32_t hme_sad_final[2][4];
    u64      hme_sad64[2][4]; // SearchResults.hme_sad (needs 64 bit after me_prune_ref for pruned refs)
    int16_t  hme_sc_x[2][4], hme_sc_y[2][4];
    uint8_t  do_ref[2][4];
    PreHme   prehme[2][4][2];
    uint8_t  performed_phme[2][4][2];
    int16_t  hx[3][2][4][2][2], hy[3][2][4][2][2]; // [level][list][ref][sr_w][sr_h]
    uint32_t hs[3][2][4][2][2];
    SvtHipSearchAreaMinMax hme_l0_sa;
    int16_t  me_cx[2][4], me_cy[2][4];   // integer-search centre (local x/y_search_center of integer_search_b64)
    uint8_t  ph_req[2][4][2];            // request index + 1 of a pushed search, 0 = none
    uint8_t  l0_req[2][4];
    uint8_t  lvl_req[2][4][2][2];
    uint8_t  c00_req[2][4];
    // the searches of the current stage and their results
    int      nreq;
    union { // never live together: the requests of the HME stages and of check_00_center / the integer searches of the probe and main stages
        Req      req[kMaxReq];
        struct { MeReq me[8], me_probe[8]; };
    };
    u64      req_key[kMaxReq];
    TileEnt  tile[kMaxTiles];       // the tiles of the current stage's searches (plan_tiles)
    uint32_t sadbuf[kNarrowMaxPos]; // per-position sums of a narrow (row-split) search
    int      nme, nprobe;
    uint32_t me_dist[85];
    uint8_t  cand0[88];  // candidate 0 of every PU (row order), for perform_gm_detection
    uint32_t red[8];
    int      job, tf_exit;
};

// A wave's LDS slice: St, the three source views, best_sad[n_slot][85] and best_mv[n_slot][85] (MeContext.p_sb_best_sad /
// p_sb_best_mv of the (list, reference) pairs the launch's pictures search: n_slot = the largest count among them), the window
// arena.  When every picture of the launch sub-samples both the HME and the integer search (SUB_SAD_SEARCH: every other row),
// the source views keep their even rows only (cshift = 1): row r of a view lives at row r >> cshift.
struct LdsLayout { uint32_t src16, src32, src64, bsad, bmv, dense, win, total; };
// Kernel modes.  kMeFull: the whole per-block pipeline in one wave (svt_hip_me_b64_kernel).  The STAGED form (launches with a dense pre-pass)
// cuts the same pipeline at its searches into small kernels -- each with the register / LDS budget of its own part, hence many more blocks in
// flight per CU than the one-wave-does-everything form, whose time goes into the latencies of a long dependent chain:
//   kMeMid1  zero-MV SADs, pre-HME / level-0 results from the pre-pass, level-1 requests     (control only)
//   kMeS1    the level-1 searches (direct form)
//   kMeMid2  level-1 results -> level-2 requests                    (control only)
//   kMeS2    the level-2 searches (direct form)
//   kMeTail  level-2 results, search centres, check-00, probe, integer search, pruning, candidates, outputs
// A block's state (the head of St) and its search requests / results travel through HBM between them.  A block whose level-1 / level-2
// searches do not qualify for the direct form (more than 32 positions, an empty area) -- or whose pre-HME / level-0 searches the pre-pass did
// not make -- is DEFERRED: the one-kernel form makes it from scratch at the end of the launch (list mode).  (Round 3 first gave such blocks
// search kernels of their own with the general staged form; once the direct form took requests with different areas -- edge blocks -- those
// kernels were empty launches, 7 us each.)
constexpr int kMeFull = 0, kMeMid1 = 1, kMeS1 = 2, kMeMid2 = 4, kMeS2 = 5, kMeTail = 7; // (3 and 6 were the staged-form search kernels)
#ifndef SVT_HIP_ME_LIST_WAVES_PER_CU
#define SVT_HIP_ME_LIST_WAVES_PER_CU 2
#endif
constexpr uint32_t kListWavesPerCu = SVT_HIP_ME_LIST_WAVES_PER_CU;
constexpr uint32_t kStagedMinJobs = 2048; // blocks of a launch from which the staged form is used (measured at 4,080 blocks -- a rank's share at 8 GPUs: 0.58 ms staged, 0.59 ms one-kernel; at 8,160: 0.75 / 0.80)
__host__ __device__ constexpr bool me_is_search(int m) { return m == kMeS1 || m == kMeS2; }
constexpr int kDirectWinBytes = 5120; // the direct searches' arena: a step's 64 x 5 window pieces, then the per-position sums [kMaxReq][kNarrowMaxPos] u32
__host__ __device__ constexpr LdsLayout lds_layout(int n_slot, int cshift, int dense = 1, int mode = 0) {
    const bool v16 = mode == kMeFull, v32 = mode == kMeFull || mode == kMeS1, v64 = mode == kMeFull || mode == kMeS2 || mode == kMeTail;
    const bool best = mode == kMeFull || mode == kMeTail, slots = dense && (mode == kMeFull || mode == kMeMid1);
    const uint32_t arena = (mode == kMeMid1 || mode == kMeMid2) ? 0u : (mode == kMeS1 || mode == kMeS2) ? (uint32_t)kDirectWinBytes : (uint32_t)kWinBytes;
    LdsLayout l = {};
    l.src16 = (uint32_t)((sizeof(St) + 15) & ~(size_t)15);
    l.src32 = l.src16 + (v16 ? (uint32_t)((16 >> cshift) * kSrc16Pitch) : 0u);
    l.src64 = l.src32 + (v32 ? (uint32_t)((32 >> cshift) * kSrc32Pitch) : 0u);
    l.bsad  = l.src64 + (v64 ? (uint32_t)((64 >> cshift) * kSrc64Pitch) : 0u);
    l.bmv   = l.bsad + (best ? (uint32_t)n_slot * 85 * 4 : 0u);
    l.dense = (l.bmv + (best ? (uint32_t)n_slot * 85 * 4 : 0u) + 15) & ~15u; // the block's slots of the dense pre-pass: n_slot x SVT_HIP_ME_DENSE_KINDS x 16 bytes
    l.win   = l.dense + (slots ? (uint32_t)n_slot * SVT_HIP_ME_DENSE_KINDS * 16u : 0u);
    l.total = l.win + arena;
    return l;
}
extern __shared__ __attribute__((aligned(16))) uint8_t g_lds[]; // the wave's LDS slice
struct Shared { // byte offsets into g_lds (pointers kept in a struct would lose the LDS address space and turn into flat accesses)
    St       &st;
    uint32_t  src64, src32, src16, win;
    int       cshift;
};
#define LDS(off) (g_lds + (off))

// Orders the wave's own LDS traffic for the compiler: what one lane stored before this point, every lane may load after it.  The
// hardware executes a wave's LDS instructions in order, so no instruction is needed -- only the compiler must not move memory
// operations across (a lane reading what another lane wrote is invisible to its single-thread view of the program).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    return v;
}
__device__ __forceinline__ u64 wave_sum64(u64 v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += (u64)__shfl_xor((unsigned long long)v, o, 64);
    return v;
}

__host__ __device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__host__ __device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__host__ __device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// svt_aom_get_scaled_picture_distance, motion_estimation.c:1239-1243
__host__ __device__ __forceinline__ uint32_t scaled_distance(uint32_t dist) { return (dist * 5) / 8 + ((dist % 8) ? 1 : 0); }

// (P = the kernel's constant-address-space view of the parameter block, or the launcher's host copy)
template <class P> __host__ __device__ __forceinline__ uint32_t ref_distance(P &p, int li, int ri) {
    long long d = (long long)p.desc.picture_number - (long long)p.desc.ref_picture_number[li][ri];
    return (uint16_t)(int16_t)(d < 0 ? -d : d);
}

// "correct the search area if it is not on the reference picture" (motion_estimation.c:838-866, 1585-1627,
// 1455-1499): the low edge moves the origin only; the high edge moves the origin, then crops the size.
__device__ __forceinline__ void clip_axis(int org, int &origin, int &size, int pad, int dim) {
    if (org + origin < -pad) origin = -pad - org;
    if (org + origin > dim - 1) origin -= (org + origin) - (dim - 1);
    if (org + origin + size > dim) size = imax(1, size - ((org + origin + size) - dim));
}

__device__ __forceinline__ const uint8_t *plane_at(CPlane &pl, int x, int y) {
    return pl.base + (long long)(pl.org_y + y) * pl.stride + (pl.org_x + x);
}

// ---------------------------------------------------------------------------------------------
// Batched SAD searches
// ---------------------------------------------------------------------------------------------

// exact k / d for k < 2^21 without an integer division: (k + 0.5) * rcp(d) is off by < 2^-22 relative, the fraction of
// (k + 0.5) / d stays 0.5 / d away from the integers
__device__ __forceinline__ float rcp_of(uint32_t d) { return __builtin_amdgcn_rcpf((float)d); }
__device__ __forceinline__ uint32_t div_by_rcp(uint32_t k, float r) { return (uint32_t)(((float)k + 0.5f) * r); }

// LDS row pitch of a staged window (its first byte is search position x0: the rows are fetched with unaligned 16-byte loads): room for
// the widest read of the last octet, and an odd multiple of 16 bytes so that consecutive rows start 4 (mod 8) banks apart
__device__ __forceinline__ uint32_t row_pitch(int w, int bw) {
    uint32_t p = (uint32_t)(((w + 7) & ~7) + bw + 4 + 15) & ~15u;
    return (p & 16u) ? p : p + 16u;
}

// bytes of LDS a tile of w x h positions needs
__device__ __forceinline__ uint32_t tile_bytes(const Req &r, int w, int h) {
    return row_pitch(w, r.bw) * (uint32_t)(h - 1 + (r.bh - 1) * r.rs + 1);
}




__device__ __forceinline__ const uint8_t *src_view(const Shared &sh, int level) {
    return LDS(level == 2 ? sh.src64 : (level == 1 ? sh.src32 : sh.src16));
}

// Packed SAD of 4 neighbouring positions over block rows [r0, r1): 4 x u32 sums in out[].  The block's source rows
// come from LDS (same address in every lane: broadcast reads), the reference rows from the staged window.  NDW = block
// width in dwords; a whole row is fetched before its qsads issue so the LDS reads overlap instead of serialising.
template <int NDW>
__device__ __forceinline__ void quad_sad_rows(const uint8_t *src, int src_pitch, int srs, const uint8_t *wrow0, int pitch, int rs, int r0, int r1,
                                              uint32_t out[4]) {
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    constexpr int kRowsPerFlush = 64 / NDW; // 64 qsads x 4 x 255 < 65536: the packed 16-bit lanes cannot overflow before a flush
    for (int rb = r0; rb < r1; rb += kRowsPerFlush) {
        const int re = rb + kRowsPerFlush < r1 ? rb + kRowsPerFlush : r1;
        u64       acc = 0;
        for (int r = rb; r < re; r++) {
            const uint32_t *s = reinterpret_cast<const uint32_t *>(src + r * srs * src_pitch);
            const uint32_t *w = reinterpret_cast<const uint32_t *>(wrow0 + r * rs * pitch);
            uint32_t sv[NDW], wv[NDW + 1];
            if constexpr (NDW % 4 == 0) { // source rows are 16-byte aligned in LDS: whole-vector reads
#pragma unroll
                for (int j = 0; j < NDW; j += 4) {
                    const uint4 q = *reinterpret_cast<const uint4 *>(s + j);
                    sv[j] = q.x; sv[j + 1] = q.y; sv[j + 2] = q.z; sv[j + 3] = q.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < NDW; j++) sv[j] = s[j];
            }
#pragma unroll
            for (int j = 0; j <= NDW; j++) wv[j] = w[j];
#pragma unroll
            for (int j = 0; j < NDW; j++) acc = __builtin_amdgcn_qsad_pk_u16_u8(((u64)wv[j + 1] << 32) | wv[j], sv[j], acc);
        }
        a0 += (uint32_t)(acc & 0xFFFF); a1 += (uint32_t)((acc >> 16) & 0xFFFF);
        a2 += (uint32_t)((acc >> 32) & 0xFFFF); a3 += (uint32_t)(acc >> 48);
    }
    out[0] = a0; out[1] = a1; out[2] = a2; out[3] = a3;
}

// block widths without a fixed-size path (right-edge blocks of pictures whose width is not a multiple of 64): rare, out of line
__device__ __attribute__((noinline)) uint4 quad_sad_generic(const uint8_t *src, int src_pitch, int srs, const uint8_t *wrow0, int pitch, int rs, int bw, int r0, int r1) {
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if ((bw & 3) == 0) { // other multiples of 4 (right-edge blocks): generic dword loop
        const int nd = bw >> 2;
        for (int r = r0; r < r1; r++) {
            const uint32_t *s = reinterpret_cast<const uint32_t *>(src + r * srs * src_pitch);
            const uint32_t *w = reinterpret_cast<const uint32_t *>(wrow0 + r * rs * pitch);
            u64      acc = 0; // at most 16 qsads per row: no overflow within a row
            uint32_t lo  = w[0];
            for (int j = 0; j < nd; j++) {
                const uint32_t hi = w[j + 1];
                acc = __builtin_amdgcn_qsad_pk_u16_u8(((u64)hi << 32) | lo, s[j], acc);
                lo  = hi;
            }
            a0 += (uint32_t)(acc & 0xFFFF); a1 += (uint32_t)((acc >> 16) & 0xFFFF);
            a2 += (uint32_t)((acc >> 32) & 0xFFFF); a3 += (uint32_t)(acc >> 48);
        }
    } else { // widths that are not a multiple of 4 (right-edge blocks of odd picture widths): byte loop
        for (int r = r0; r < r1; r++) {
            const uint8_t *s = src + r * srs * src_pitch;
            const uint8_t *w = wrow0 + r * rs * pitch;
            for (int c = 0; c < bw; c++) {
                const int sv = s[c];
                a0 += (uint32_t)iabs(sv - (int)w[c]);     a1 += (uint32_t)iabs(sv - (int)w[c + 1]);
                a2 += (uint32_t)iabs(sv - (int)w[c + 2]); a3 += (uint32_t)iabs(sv - (int)w[c + 3]);
            }
        }
    }
    return make_uint4(a0, a1, a2, a3);
}

__device__ __forceinline__ void quad_sad(const uint8_t *src, int src_pitch, int srs, const uint8_t *wrow0, int pitch, int rs, int bw, int r0,
                                         int r1, uint32_t out[4]) {
    switch (bw) { // wave-uniform for all practical batches (one block width per HME level)
    case 16: quad_sad_rows<4>(src, src_pitch, srs, wrow0, pitch, rs, r0, r1, out); return;
    case 32: quad_sad_rows<8>(src, src_pitch, srs, wrow0, pitch, rs, r0, r1, out); return;
    case 64: quad_sad_rows<16>(src, src_pitch, srs, wrow0, pitch, rs, r0, r1, out); return;
    case 8: quad_sad_rows<2>(src, src_pitch, srs, wrow0, pitch, rs, r0, r1, out); return;
    case 4: quad_sad_rows<1>(src, src_pitch, srs, wrow0, pitch, rs, r0, r1, out); return;
    default: break;
    }
    const uint4 v = quad_sad_generic(src, src_pitch, srs, wrow0, pitch, rs, bw, r0, r1);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}

// Wide tiles evaluate 8 neighbouring positions per item: the second quad starts one dword further, so a row costs NDW + 2
// window reads (and NDW source reads) for 2 * NDW qsads instead of 2 * NDW + 2 (and 2 * NDW).
template <int NDW>
__device__ __forceinline__ void oct_sad_rows(const uint8_t *src, int src_pitch, int srs, const uint8_t *wrow0, int pitch, int rs, int r0, int r1, uint32_t out[8]) {
    uint32_t a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int kRowsPerFlush = 64 / NDW;
    constexpr int CH = NDW > 8 ? 8 : NDW; // a row goes through the registers in chunks of at most 8 dwords (a 64-pixel row at once costs 34 registers)
    for (int rb = r0; rb < r1; rb += kRowsPerFlush) {
        const int re = rb + kRowsPerFlush < r1 ? rb + kRowsPerFlush : r1;
        u64       acc0 = 0, acc1 = 0;
        for (int r = rb; r < re; r++) {
            const uint32_t *s = reinterpret_cast<const uint32_t *>(src + r * srs * src_pitch);
            const uint32_t *w = reinterpret_cast<const uint32_t *>(wrow0 + r * rs * pitch);
#pragma unroll
            for (int c0 = 0; c0 < NDW; c0 += CH) {
                uint32_t sv[CH], wv[CH + 2];
                if constexpr (CH % 4 == 0) { // source rows are 16-byte aligned in LDS: whole-vector reads
#pragma unroll
                    for (int j = 0; j < CH; j += 4) {
                        const uint4 q = *reinterpret_cast<const uint4 *>(s + c0 + j);
                        sv[j] = q.x; sv[j + 1] = q.y; sv[j + 2] = q.z; sv[j + 3] = q.w;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < CH; j++) sv[j] = s[c0 + j];
                }
                if constexpr (CH % 2 == 0) { // octets start on 8-byte boundaries of the arena: 8-byte window reads
#pragma unroll
                    for (int j = 0; j < CH + 2; j += 2) {
                        const uint2 q = *reinterpret_cast<const uint2 *>(w + c0 + j);
                        wv[j] = q.x; wv[j + 1] = q.y;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < CH + 2; j++) wv[j] = w[c0 + j];
                }
#pragma unroll
                for (int j = 0; j < CH; j++) {
                    acc0 = __builtin_amdgcn_qsad_pk_u16_u8(((u64)wv[j + 1] << 32) | wv[j], sv[j], acc0);
                    acc1 = __builtin_amdgcn_qsad_pk_u16_u8(((u64)wv[j + 2] << 32) | wv[j + 1], sv[j], acc1);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) { a[i] += (uint32_t)((acc0 >> (16 * i)) & 0xFFFF); a[4 + i] += (uint32_t)((acc1 >> (16 * i)) & 0xFFFF); }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = a[i];
}

__device__ __forceinline__ void oct_sad(const uint8_t *src, int src_pitch, int srs, const uint8_t *wrow0, int pitch, int rs, int bw, int r0, int r1, uint32_t out[8]) {
    switch (bw) {
    case 16: oct_sad_rows<4>(src, src_pitch, srs, wrow0, pitch, rs, r0, r1, out); return;
    case 32: oct_sad_rows<8>(src, src_pitch, srs, wrow0, pitch, rs, r0, r1, out); return;
    case 64: oct_sad_rows<16>(src, src_pitch, srs, wrow0, pitch, rs, r0, r1, out); return;
    case 8: oct_sad_rows<2>(src, src_pitch, srs, wrow0, pitch, rs, r0, r1, out); return;
    case 4: oct_sad_rows<1>(src, src_pitch, srs, wrow0, pitch, rs, r0, r1, out); return;
    default: break;
    }
    quad_sad(src, src_pitch, srs, wrow0, pitch, rs, bw, r0, r1, out);         // other widths: two quads
    quad_sad(src, src_pitch, srs, wrow0 + 4, pitch, rs, bw, r0, r1, out + 4);
}

// ---- the search engine: the requests of a stage, one after the other, each as one or more tiles through the LDS arena ----
typedef uint32_t V4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) V4 GV4; // plane addresses come out of LDS-resident structs, which hides their address
                                                        // space from the compiler: say "global" (a flat load also counts against the LDS counter)
constexpr int kVecPerLane = (kWinBytes / 16 + 63) / 64; // 16-byte vectors a lane moves for a full arena

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); } // wave-uniform value -> SGPR
__device__ __forceinline__ const uint8_t *uni_ptr(const uint8_t *p) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    return reinterpret_cast<const uint8_t *>(((uintptr_t)uni((uint32_t)(a >> 32)) << 32) | uni((uint32_t)a));
}

// minimum over the wave's 64 lanes, in every lane's return value (SGPR): butterfly inside the 16-lane rows by DPP (min is idempotent, so the
// mirror patterns serve), two row broadcasts, lane 63 holds the result -- no LDS traffic
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#define SVT_MIN_DPP(ctrl, rmask) { const uint32_t t_ = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xF, false); v = t_ < v ? t_ : v; }
    SVT_MIN_DPP(0xB1, 0xF)  // quad_perm [1,0,3,2]
    SVT_MIN_DPP(0x4E, 0xF)  // quad_perm [2,3,0,1]
    SVT_MIN_DPP(0x141, 0xF) // row_half_mirror
    SVT_MIN_DPP(0x140, 0xF) // row_mirror
    SVT_MIN_DPP(0x142, 0xA) // row_bcast:15 -> rows 1, 3
    SVT_MIN_DPP(0x143, 0xC) // row_bcast:31 -> rows 2, 3
#undef SVT_MIN_DPP
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// sum over the wave's 64 lanes (mod 2^32), in every lane's return value (SGPR), by DPP: no LDS traffic
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#define SVT_SUM_DPP(ctrl, rmask, bc) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xF, bc);
    SVT_SUM_DPP(0x111, 0xF, true)  // row_shr:1 (zero fill)
    SVT_SUM_DPP(0x112, 0xF, true)  // row_shr:2
    SVT_SUM_DPP(0x114, 0xF, true)  // row_shr:4
    SVT_SUM_DPP(0x118, 0xF, true)  // row_shr:8: lane 15 of a row holds the row's sum
    SVT_SUM_DPP(0x142, 0xA, false) // row_bcast:15 -> rows 1, 3
    SVT_SUM_DPP(0x143, 0xC, false) // row_bcast:31 -> rows 2, 3
#undef SVT_SUM_DPP
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// the wave's smallest (sad, pos) pair as sad << 32 | pos -- the reference's "first minimum in raster order": the smallest SAD, then among
// its lanes the smallest position word (y << 16 | x); ~0 when no lane holds a result
__device__ __forceinline__ u64 wave_min_key(uint32_t sad, uint32_t pos) {
    const uint32_t smin = wave_min_u32(sad);
    const uint32_t pmin = wave_min_u32(sad == smin ? pos : 0xFFFFFFFFu);
    return ((u64)smin << 32) | pmin;
}

// A tile entry in scalar registers.  Every field is wave-uniform.
struct TileGeo {
    const uint8_t *g0;
    uint32_t stride;
    int      req;
    int      x0, y0, w, h;
    int      pitch, nvec;
    int      last;         // last tile of its request
    int      bw, bh, rs, level, skip_even, narrow;
    int      kd, km;
};

// inclusive prefix sum over lanes 0 .. 31 (rows 0 and 1 of the wave) by DPP: no LDS traffic
__device__ __forceinline__ int scan32(int v) {
#define SVT_SCAN_DPP(ctrl, rmask, bc) v += __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xF, bc);
    SVT_SCAN_DPP(0x111, 0xF, true)  // row_shr:1 (zero fill)
    SVT_SCAN_DPP(0x112, 0xF, true)  // row_shr:2
    SVT_SCAN_DPP(0x114, 0xF, true)  // row_shr:4
    SVT_SCAN_DPP(0x118, 0xF, true)  // row_shr:8
    SVT_SCAN_DPP(0x142, 0xA, false) // row_bcast:15 -> row 1 (and 3)
#undef SVT_SCAN_DPP
    return v;
}

// Plans the tiles of st.req[0 .. nreq) (nreq <= 32), lane <-> request: a request is cut on a fixed (tw x th) grid chosen so that any of its
// tiles fits the arena (raster order over the grid); tile T of the stage's sequence goes to st.tile[T - t0] for T in [t0, t0 + kMaxTiles).
// Returns the stage's tile count.
__device__ __forceinline__ int plan_tiles(St &st, int nreq, int t0) {
    const int lane = threadIdx.x;
    int cnt = 0, tw = 8, th = 1, ntx = 1;
    Req r = {};
    if (lane < nreq) {
        r = st.req[lane];
        if (r.done) {
            // nothing to evaluate
        } else if (r.sa_w > 0 && r.sa_h > 0) {
            tw = r.sa_w; th = r.sa_h;
            while (th > 1 && tile_bytes(r, tw, th) > (uint32_t)kWinBytes) th = (th + 1) >> 1;
            while (tw > 8 && tile_bytes(r, tw, th) > (uint32_t)kWinBytes) tw = ((tw >> 1) + 7) & ~7;
            ntx = (int)div_by_rcp((uint32_t)(r.sa_w + tw - 1), rcp_of((uint32_t)tw));
            cnt = ntx * (int)div_by_rcp((uint32_t)(r.sa_h + th - 1), rcp_of((uint32_t)th));
        } else if (t0 == 0)
            st.req_key[lane] = (0xffffffull << 32) | 0xffffffffull; // an empty search area: no position evaluated
    }
    const int incl = scan32(cnt), base = incl - cnt - t0;
    const int total = __builtin_amdgcn_readlane(incl, 31);
    const float ntx_rcp = rcp_of((uint32_t)ntx);
    const int narrow = (r.sa_w * r.sa_h <= kNarrowMaxPos) ? 1 : 0;
    for (int k = imax(0, -base); k < cnt && base + k < kMaxTiles; k++) { // divergent: a lane writes its own request's tiles
        const int ty = (int)div_by_rcp((uint32_t)k, ntx_rcp), tx = k - ty * ntx;
        const int x0 = tx * tw, y0 = ty * th, w = imin(tw, r.sa_w - x0), h = imin(th, r.sa_h - y0);
        const int pitch = (int)row_pitch(w, r.bw), vpr = pitch >> 4;
        const int kd = (int)div_by_rcp(64u, rcp_of((uint32_t)vpr));
        TileEnt e;
        e.g0 = r.win + x0 + (long long)y0 * r.stride; e.stride = r.stride;
        e.x0 = (int16_t)x0; e.y0 = (int16_t)y0; e.w = (int16_t)w; e.h = (int16_t)h;
        e.pitch = (uint16_t)pitch; e.nvec = (uint16_t)(vpr * (h - 1 + (r.bh - 1) * r.rs + 1));
        e.req = (uint8_t)lane; e.flags = (uint8_t)(r.level | (r.skip_even << 2) | (narrow << 3) | ((k == cnt - 1) << 4));
        e.bw = r.bw; e.bh = r.bh; e.rs = r.rs; e.kd = (uint8_t)kd; e.km = (uint8_t)(64 - kd * vpr); e.pad = 0;
        st.tile[base + k] = e;
    }
    return total;
}

// tile entry i -> scalar registers (two 16-byte broadcast reads)
__device__ __forceinline__ TileGeo read_tile(const St &st, int i) {
    const uint4 a = *reinterpret_cast<const uint4 *>(&st.tile[i]);
    const uint4 b = *(reinterpret_cast<const uint4 *>(&st.tile[i]) + 1);
    const uint32_t a0 = uni(a.x), a1 = uni(a.y), a2 = uni(a.z), a3 = uni(a.w), b0 = uni(b.x), b1 = uni(b.y), b2 = uni(b.z), b3 = uni(b.w);
    TileGeo t;
    t.g0 = reinterpret_cast<const uint8_t *>(((uintptr_t)a1 << 32) | a0); t.stride = a2;
    t.x0 = (int16_t)(a3 & 0xFFFF); t.y0 = (int16_t)(a3 >> 16); t.w = (int16_t)(b0 & 0xFFFF); t.h = (int16_t)(b0 >> 16);
    t.pitch = (int)(b1 & 0xFFFF); t.nvec = (int)(b1 >> 16);
    t.req = (int)(b2 & 0xFF);
    const int f = (int)((b2 >> 8) & 0xFF);
    t.level = f & 3; t.skip_even = (f >> 2) & 1; t.narrow = (f >> 3) & 1; t.last = (f >> 4) & 1;
    t.bw = (int)((b2 >> 16) & 0xFF); t.bh = (int)(b2 >> 24);
    t.rs = (int)(b3 & 0xFF); t.kd = (int)((b3 >> 8) & 0xFF); t.km = (int)((b3 >> 16) & 0xFF);
    return t;
}

// the tile's window, global -> registers: lane l owns vectors l, l + 64, ... of the tile's flattened (row, vector) space; they go into the
// register slots S0 .. S0 + NS - 1 (compile-time numbers: a runtime slot index would send the array to scratch memory).  The rows start at
// an arbitrary byte: unaligned 16-byte loads (position x0 lands on LDS byte 0 of its row, so a tile has no dead leading positions)
typedef uint32_t V4U __attribute__((ext_vector_type(4), aligned(1)));
typedef const __attribute__((address_space(1))) V4U GV4U;
constexpr int kSmallVec = 128; // a tile of at most this many vectors takes two slots: four such tiles share a round
template <int S0, int NS>
__device__ __forceinline__ void tile_load(const TileGeo &t, V4 (&v)[kVecPerLane]) {
    // Addresses are the uniform window base (SGPR pair) + a 32-bit byte offset per lane, advanced incrementally from pass to pass: no 64-bit
    // vector arithmetic.  Offsets grow with the vector number, so clamping to the last vector's offset keeps the lanes past the end inside
    // the window (their data is never stored).
    const int      lane = threadIdx.x, vpr = t.pitch >> 4, nrows = (int)div_by_rcp((uint32_t)t.nvec, rcp_of((uint32_t)vpr));
    const int      row0 = (int)div_by_rcp((uint32_t)lane, rcp_of((uint32_t)vpr));
    int            c    = lane - row0 * vpr; // vector `lane` = (row0, c); later passes advance by (kd, km)
    uint32_t       off  = (uint32_t)__mul24(row0, (int)t.stride) + (uint32_t)c * 16u;
    const uint32_t last = (uint32_t)__mul24(nrows - 1, (int)t.stride) + (uint32_t)(vpr - 1) * 16u;
    const uint32_t step = (uint32_t)__mul24(t.kd, (int)t.stride) + (uint32_t)t.km * 16u, wrap = t.stride - (uint32_t)vpr * 16u;
    const uintptr_t base = reinterpret_cast<uintptr_t>(t.g0);
#pragma unroll
    for (int j = 0; j < NS; j++)
        if (j * 64 < t.nvec) { // uniform
            v[S0 + j] = __builtin_bit_cast(V4, *reinterpret_cast<GV4U *>(base + (uint64_t)umin(off, last)));
            off += step; c += t.km;
            if (c >= vpr) { c -= vpr; off += wrap; }
        }
}
template <int S0, int NS>
__device__ __forceinline__ void tile_store(const Shared &sh, int nvec, int lds_vec, const V4 (&v)[kVecPerLane]) {
    const int lane = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NS; j++)
        if (j * 64 < nvec && lane + j * 64 < nvec) *reinterpret_cast<V4 *>(&LDS(sh.win)[(lds_vec + lane + j * 64) * 16]) = v[S0 + j];
}

// every position of the tile in the arena; returns the tile's best key (sad << 32 | y << 16 | x), ~0 when no position was evaluated
__device__ __forceinline__ u64 tile_eval(const Shared &sh, const TileGeo &t, int lds_off PROF_PARAM) {
    St            &st   = sh.st;
    const int      lane = threadIdx.x;
    const uint8_t *src  = src_view(sh, t.level);
    const uint8_t *win  = LDS(sh.win) + lds_off;
    const int      sp   = (t.level == 2) ? kSrc64Pitch : (t.level == 1 ? kSrc32Pitch : kSrc16Pitch);
    const int      srs  = t.rs >> sh.cshift; // source row step: the views keep every (1 << cshift)-th row
    uint32_t bsad_ = 0xFFFFFFFFu, bpos_ = 0xFFFFFFFFu;
    if (!t.narrow) { // lane <-> 8 neighbouring positions of one search row, the whole block
        const int   ng = (t.w + 7) >> 3, nitems = ng * t.h;
        const float ng_rcp = rcp_of((uint32_t)ng);
        // A lane visits its positions in raster order, so its first minimum is the minimum of (sad << 12 | visit number): one 32-bit key
        // (sad < 2^20: 64 x 64 x 255; at most 512 items x 8 positions per lane)
        uint32_t best = 0xFFFFFFFFu;
        int      seq  = 0;
        for (int it = lane; it < nitems; it += kThreads, seq += 8) {
            const int y = (int)div_by_rcp((uint32_t)it, ng_rcp), g = it - y * ng;
            if (t.skip_even && !((t.y0 + y) & 1)) continue;
            uint32_t s8[8];
            oct_sad(src, sp, srs, win + y * t.pitch + 8 * g, t.pitch, t.rs, t.bw, 0, t.bh, s8);
            uint32_t k[8];
#pragma unroll
            for (int i = 0; i < 8; i++) k[i] = (s8[i] << 12) + (uint32_t)i;
            if (t.w & 7) { // uniform: only then an octet can hang over the end of the search row
#pragma unroll
                for (int i = 0; i < 8; i++) k[i] = (8 * g + i < t.w) ? k[i] : 0xFFFFF000u;
            }
            const uint32_t m = umin(umin(umin(k[0], k[1]), k[2]), umin(umin(umin(k[3], k[4]), k[5]), umin(k[6], k[7]))); // v_min3_u32
            const uint32_t cand = m + (uint32_t)seq;
            best = cand < best ? cand : best;
        }
        PROF(30);
        if ((best >> 12) != 0xFFFFFu) { // decode the visit number: iteration, position in the octet
            const int sq = (int)(best & 0xFFFu), it = lane + (sq >> 3) * kThreads;
            const int y = (int)div_by_rcp((uint32_t)it, ng_rcp), g = it - y * ng;
            bsad_ = best >> 12;
            bpos_ = ((uint32_t)(t.y0 + y) << 16) | (uint32_t)(t.x0 + 8 * g + (sq & 7));
        }
    } else { // few positions: lane <-> 8 positions x a slice of the block rows (as many slices as keep the wave's lanes busy), summed per position in LDS
        if (lane < kNarrowMaxPos) st.sadbuf[lane] = 0;
        wave_sync();
        const int ng = (t.w + 7) >> 3, per = ng * t.h;
        int       slices = imin(t.bh, imax(1, (int)uni(div_by_rcp(64u, rcp_of((uint32_t)per)))));
        const int rows_per = (int)uni(div_by_rcp((uint32_t)(t.bh + slices - 1), rcp_of((uint32_t)slices)));
        slices = (int)uni(div_by_rcp((uint32_t)(t.bh + rows_per - 1), rcp_of((uint32_t)rows_per)));
        const int   nitems = per * slices;
        const float ng_rcp = rcp_of((uint32_t)ng), h_rcp = rcp_of((uint32_t)t.h);
        for (int it = lane; it < nitems; it += kThreads) {
            const int q = (int)div_by_rcp((uint32_t)it, ng_rcp), g = it - q * ng;
            const int slice = (int)div_by_rcp((uint32_t)q, h_rcp), y = q - slice * t.h;
            if (t.skip_even && !((t.y0 + y) & 1)) continue;
            uint32_t s8[8];
            oct_sad(src, sp, srs, win + y * t.pitch + 8 * g, t.pitch, t.rs, t.bw, slice * rows_per, imin(slice * rows_per + rows_per, t.bh), s8);
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (8 * g + i < t.w) atomicAdd(&st.sadbuf[y * t.w + 8 * g + i], s8[i]);
        }
        wave_sync();
        PROF(23);
        if (lane < t.w * t.h) {
            const int y = (int)(((float)lane + 0.5f) * __frcp_rn((float)t.w)), x = lane - y * t.w; // exact: w, lane <= 32
            if (!(t.skip_even && !((t.y0 + y) & 1))) { bsad_ = st.sadbuf[lane]; bpos_ = ((uint32_t)(t.y0 + y) << 16) | (uint32_t)(t.x0 + x); }
        }
    }
    return wave_min_key(bsad_, bpos_);
}

// The next round of the stage's tile sequence, starting at table entry i (n entries in the table): either ONE tile of any size that fits the
// arena (all register slots), or up to four small tiles (at most kSmallVec vectors each: slots 2j, 2j + 1) while they fit the arena
// together.  Sends the round's windows on their way into the register slots; returns the number of tiles.
__device__ __forceinline__ int load_round(const St &st, int i, int n, V4 (&v)[kVecPerLane], int (&nvo)[4]) {
    if (i >= n) return 0;
    // lanes 0 .. 3 look at the sizes of entries i .. i + 3 (one LDS round trip for the whole decision)
    const int lane = threadIdx.x;
    const uint32_t nv = (lane < 4 && i + lane < n) ? (uint32_t)st.tile[i + lane].nvec : 0xFFFFu;
    const int n0 = __builtin_amdgcn_readlane((int)nv, 0), n1 = __builtin_amdgcn_readlane((int)nv, 1), n2 = __builtin_amdgcn_readlane((int)nv, 2),
              n3 = __builtin_amdgcn_readlane((int)nv, 3);
    nvo[0] = n0; nvo[1] = n1; nvo[2] = n2; nvo[3] = n3;
    if (n0 > kSmallVec) {
        tile_load<0, kVecPerLane>(read_tile(st, i), v);
        return 1;
    }
    constexpr int kArenaVec = kWinBytes / 16;
    int nt = 1;
    if (n1 <= kSmallVec && n0 + n1 <= kArenaVec) {
        nt = 2;
        if (n2 <= kSmallVec && n0 + n1 + n2 <= kArenaVec) {
            nt = 3;
            if (n3 <= kSmallVec && n0 + n1 + n2 + n3 <= kArenaVec) nt = 4;
        }
    }
    tile_load<0, 2>(read_tile(st, i), v);
    if (nt > 1) tile_load<2, 2>(read_tile(st, i + 1), v);
    if (nt > 2) tile_load<4, 2>(read_tile(st, i + 2), v);
    if (nt > 3) tile_load<6, 2>(read_tile(st, i + 3), v);
    return nt;
}
static_assert(kVecPerLane >= 8, "four small tiles of two slots each");

// The wave runs st.req[0 .. nreq) (nreq >= 1) to completion; results in st.req_key[] = (sad << 32 | y << 16 | x), or
// (0xffffff << 32 | 0xffffffff) when no position was evaluated.  The searches go through the arena in ROUNDS of as many windows as it
// holds (the eight level-0 windows of a short-distance picture are two rounds, a level-2 window is a round of its own); while a round is
// evaluated out of the arena, the next round's windows are already on their way into registers.
__device__ __forceinline__ void run_searches(Shared &sh PROF_PARAM) {
    St        &st   = sh.st;
    const int  nreq = (int)uni((uint32_t)st.nreq);
    V4         v[kVecPerLane];
    u64        key = (0xffffffull << 32) | 0xffffffffull;
    for (int t0 = 0;; t0 += kMaxTiles) { // one pass unless the stage has more than kMaxTiles tiles
        const int total = plan_tiles(st, nreq, t0), n = imin(total - t0, kMaxTiles);
        wave_sync(); // the table
        int nv[4] = {0, 0, 0, 0}, nvn[4] = {0, 0, 0, 0};
        int i = 0, nt = load_round(st, 0, n, v, nv);
        PROFS(24);
        while (nt) {
            { // slots are compile-time: a single big tile sits in all of them, small tile j in slots 2j, 2j + 1
                if (nv[0] > kSmallVec) tile_store<0, kVecPerLane>(sh, nv[0], 0, v);
                else {
                    tile_store<0, 2>(sh, nv[0], 0, v);
                    if (nt > 1) tile_store<2, 2>(sh, nv[1], nv[0], v);
                    if (nt > 2) tile_store<4, 2>(sh, nv[2], nv[0] + nv[1], v);
                    if (nt > 3) tile_store<6, 2>(sh, nv[3], nv[0] + nv[1] + nv[2], v);
                }
            }
            wave_sync();
            PROFS(32);
            const int nn = load_round(st, i + nt, n, v, nvn);
            PROF(22);
            int lds_vec = 0;
            for (int j = 0; j < nt; j++) {
                const TileGeo t = read_tile(st, i + j);
                const u64 k = tile_eval(sh, t, lds_vec * 16 PROF_ARG);
                key = k < key ? k : key;
                lds_vec += t.nvec;
                if (t.last) {
                    if (threadIdx.x == 0) st.req_key[t.req] = key;
                    key = (0xffffffull << 32) | 0xffffffffull;
                }
            }
            wave_sync(); // the arena is free again
            PROFS(40);
            i += nt;
            nt = nn;
#pragma unroll
            for (int j = 0; j < 4; j++) nv[j] = nvn[j];
        }
        if (t0 + kMaxTiles >= total) break;
        wave_sync(); // the table is rewritten
    }
}

// A stage whose searches are all SMALL and alike (the HME level-1 / level-2 refinements: 8 x 3 positions around eight centres on a 32 x 32 /
// 64 x 64 block) skips the arena: the octets of positions of ALL its searches are spread over the lanes together with slices of the block
// rows, and a lane fetches its window rows straight from global memory (a row is NDW + 2 dwords: two or five unaligned 16-byte loads, the
// next row on its way while the current one is evaluated) -- no tile plan, no staging rounds, one pass for the whole stage.  Per-position
// sums meet in the (idle) arena by LDS atomics; one arg-min per search.  Returns false (nothing done) when the stage does not qualify.
// One pass of a direct stage.  A lane owns one item (an octet of positions x a slice of block rows) and needs, per block row, the NDW + 2
// window dwords under it: NV 16-byte pieces of ONE window row.  Fetched by their owner the pieces of a step lie in 64 different rows -- every
// lane of every load instruction in a cache line of its own, and the texture addresser, which takes them one line at a time, becomes the
// stage's bound (measured: TA busy 85 % of the level-2 kernel's time).  So the wave fetches a step's 64 x NV pieces in LINEAR order instead
// (piece n = owner n / NV, part n % NV: the NV lanes of a row side by side in one or two lines), hands them over through the arena, and
// every owner reads its row back with NV ds_read_b128 (lane stride NV x 16 bytes, NV odd: conflict-free).  The next step's pieces are on
// their way while the current one is evaluated.  `wp` = the owner's first window row (idle lanes: rows <= 0), `wstep` uniform.
template <int NDW, int DEPTH>
__device__ __forceinline__ void direct_rows(uint8_t *arena, const uint8_t *src, int sp, int srs, const uint8_t *wp, long long wstep, int e0, int rows, int rows_max, uint32_t out[8]) {
    constexpr int NV = (NDW + 2 + 3) / 4, kRowsPerFlush = 64 / NDW;
    static_assert(NV & 1, "owners read their pieces back without bank conflicts");
    static_assert(DEPTH >= 1 && DEPTH <= 3, "steps in flight");
    const int lane = threadIdx.x;
    const uint8_t *pp[NV]; // where this lane fetches piece (lane + 64 k) of the next step to request
    int            pn[NV]; // steps its owner makes
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const int n = lane + 64 * k, owner = n / NV, part = n - owner * NV;
        const u64 base = ((u64)(uint32_t)__shfl((int)(uint32_t)((uintptr_t)wp >> 32), owner, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)(uintptr_t)wp, owner, 64);
        pp[k] = reinterpret_cast<const uint8_t *>((uintptr_t)base) + 16 * part;
        pn[k] = __shfl(rows, owner, 64);
    }
    uint32_t a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    V4U v[DEPTH][NV];
    auto request = [&](V4U (&r)[NV], int step) { // the pieces of `step` (requests are made in step order: pp walks along)
#pragma unroll
        for (int k = 0; k < NV; k++) {
            if (step < pn[k]) r[k] = *reinterpret_cast<GV4U *>(reinterpret_cast<uintptr_t>(pp[k]));
            pp[k] += wstep;
        }
    };
#pragma unroll
    for (int dpt = 0; dpt < DEPTH; dpt++) {
#pragma unroll
        for (int k = 0; k < NV; k++) v[dpt][k] = V4U{0, 0, 0, 0};
        request(v[dpt], dpt);
    }
    u64 acc0 = 0, acc1 = 0;
    auto step = [&](V4U (&r)[NV], int t) {
#pragma unroll
        for (int k = 0; k < NV; k++) *reinterpret_cast<V4U *>(arena + 16 * (lane + 64 * k)) = r[k];
        wave_sync();
        request(r, t + DEPTH);
        if (t < rows) {
            const uint32_t *s = reinterpret_cast<const uint32_t *>(src + (e0 + t) * srs * sp);
            uint32_t sv[NDW], wv[4 * NV];
#pragma unroll
            for (int j = 0; j < NDW; j += 4) {
                const uint4 q = *reinterpret_cast<const uint4 *>(s + j);
                sv[j] = q.x; sv[j + 1] = q.y; sv[j + 2] = q.z; sv[j + 3] = q.w;
            }
#pragma unroll
            for (int k = 0; k < NV; k++) {
                const uint4 q = *reinterpret_cast<const uint4 *>(arena + 16 * (lane * NV + k));
                wv[4 * k] = q.x; wv[4 * k + 1] = q.y; wv[4 * k + 2] = q.z; wv[4 * k + 3] = q.w;
            }
#pragma unroll
            for (int j = 0; j < NDW; j++) {
                acc0 = __builtin_amdgcn_qsad_pk_u16_u8(((u64)wv[j + 1] << 32) | wv[j], sv[j], acc0);
                acc1 = __builtin_amdgcn_qsad_pk_u16_u8(((u64)wv[j + 2] << 32) | wv[j + 1], sv[j], acc1);
            }
        }
        if ((t + 1) % kRowsPerFlush == 0 || t + 1 == rows_max) { // uniform: the 16-bit sums hold kRowsPerFlush rows
#pragma unroll
            for (int i = 0; i < 4; i++) { a[i] += (uint32_t)((acc0 >> (16 * i)) & 0xFFFF); a[4 + i] += (uint32_t)((acc1 >> (16 * i)) & 0xFFFF); }
            acc0 = acc1 = 0;
        }
        wave_sync(); // the arena is rewritten
    };
    for (int t = 0; t < rows_max; t += DEPTH) { // uniform
        step(v[0], t);
        if (DEPTH >= 2 && t + 1 < rows_max) step(v[DEPTH >= 2 ? 1 : 0], t + 1);
        if (DEPTH >= 3 && t + 2 < rows_max) step(v[DEPTH >= 3 ? 2 : 0], t + 2);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = a[i];
}

// do the stage's requests st.req[0 .. nreq) qualify for run_small_searches_direct?  (wave-uniform answer)  Block shape, row step, source view
// and stride are the stage's; the search AREAS may differ from request to request (an edge block's are clipped one by one) as long as each is
// small and all their octet items fit one slice of the wave.  `q_total`: the items of a slice.
__device__ __forceinline__ bool small_direct_ok(const St &st, int *q_total = nullptr) {
    const int lane = threadIdx.x, nreq = (int)uni((uint32_t)st.nreq);
    const Req &r0 = st.req[0];
    const int  bw = (int)uni(r0.bw), bh = (int)uni(r0.bh), rs = (int)uni(r0.rs), level = (int)uni(r0.level), stride = (int)uni((uint32_t)r0.stride);
    bool ok = true;
    int  per = 0;
    if (lane < nreq) {
        const Req &r = st.req[lane];
        const int  w = r.sa_w, h = r.sa_h;
        ok  = w > 0 && h > 0 && w * h <= kNarrowMaxPos && r.bw == bw && r.bh == bh && r.rs == rs && r.level == level && (int)r.stride == stride && !r.skip_even && !r.done;
        per = ((w + 7) >> 3) * h;
    }
    if (!__all(ok) || (bw != 32 && bw != 64) || nreq > kMaxReq || nreq <= 0) return false;
    const int q = (int)wave_sum_u32((uint32_t)per);
    if (q_total) *q_total = q;
    return q <= kThreads;
}
template <int DEPTH = 1> __device__ __forceinline__ bool run_small_searches_direct(Shared &sh PROF_PARAM) {
    St       &st   = sh.st;
    const int lane = threadIdx.x, nreq = (int)uni((uint32_t)st.nreq);
    // few positions per search, whole-vector source rows, one block shape
    int Q = 0; // octet items of one slice
    if (!small_direct_ok(st, &Q)) return false;
    Q = (int)uni((uint32_t)Q);
    const Req &r0 = st.req[0];
    const int  bw = (int)uni(r0.bw), bh = (int)uni(r0.bh), rs = (int)uni(r0.rs), level = (int)uni(r0.level);
    // where a request's items start inside a slice: exclusive prefix of its item count over the requests (lane <-> request), parked in the
    // requests' key slots until the keys themselves are written
    {
        int per = 0;
        if (lane < nreq) per = ((st.req[lane].sa_w + 7) >> 3) * st.req[lane].sa_h;
        int incl = per;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) { const int t = __shfl_up(incl, o, 64); if ((lane & 31) >= o) incl += t; } // (kMaxReq = 32 requests)
        if (lane < nreq) st.req_key[lane] = (u64)(uint32_t)(incl - per);
    }
    wave_sync();
    int       S        = imin(bh, (int)uni(div_by_rcp((uint32_t)kThreads, rcp_of((uint32_t)Q))));
    const int rows_per = (int)uni(div_by_rcp((uint32_t)(bh + S - 1), rcp_of((uint32_t)S)));
    S                  = (int)uni(div_by_rcp((uint32_t)(bh + rows_per - 1), rcp_of((uint32_t)rows_per)));
    uint32_t *sad = reinterpret_cast<uint32_t *>(LDS(sh.win)); // [nreq][kNarrowMaxPos], once the rows have passed through the arena
    const uint8_t *src = src_view(sh, level);
    const int      sp  = (level == 2) ? kSrc64Pitch : (level == 1 ? kSrc32Pitch : kSrc16Pitch), srs = rs >> sh.cshift;
    const long long wstep = (long long)rs * (long long)(int)uni((uint32_t)r0.stride);
    const bool mine = lane < Q * S;
    int req = 0, y = 0, g = 0, e0 = 0, rows = 0, w = 1;
    const uint8_t *wp = nullptr;
    if (mine) {
        const int slice = (int)div_by_rcp((uint32_t)lane, rcp_of((uint32_t)Q)), q = lane - slice * Q;
        for (int r = 1; r < nreq; r++) req += q >= (int)(uint32_t)st.req_key[r] ? 1 : 0; // (broadcast reads; the prefix is monotone)
        const Req &r = st.req[req];
        w = r.sa_w;
        const int ng = (w + 7) >> 3, ql = q - (int)(uint32_t)st.req_key[req];
        y = (int)div_by_rcp((uint32_t)ql, rcp_of((uint32_t)ng)); g = ql - y * ng;
        e0   = slice * rows_per;
        rows = imin(e0 + rows_per, bh) - e0;
        wp   = r.win + (long long)y * r.stride + (long long)e0 * wstep + 8 * g;
    }
    uint32_t s8[8];
    PROF(26);
    if (bw == 64) direct_rows<16, DEPTH>(LDS(sh.win), src, sp, srs, wp, wstep, e0, rows, rows_per, s8);
    else direct_rows<8, DEPTH>(LDS(sh.win), src, sp, srs, wp, wstep, e0, rows, rows_per, s8);
    PROF(27);
    for (int p = lane; p < nreq * kNarrowMaxPos; p += kThreads) sad[p] = 0;
    wave_sync();
    if (mine) {
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (8 * g + i < w) atomicAdd(&sad[req * kNarrowMaxPos + y * w + 8 * g + i], s8[i]);
    }
    wave_sync();
    PROF(23);
    for (int rq = 0; rq < nreq; rq++) { // lane <-> position in raster order: the first minimum is the smallest (sad << 6 | lane) (sad < 2^20)
        const int      wr = (int16_t)uni((uint32_t)st.req[rq].sa_w), hr = (int16_t)uni((uint32_t)st.req[rq].sa_h);
        const uint32_t k32 = wave_min_u32(lane < wr * hr ? (sad[rq * kNarrowMaxPos + lane] << 6) | (uint32_t)lane : 0xFFFFFFFFu);
        const int      wl = (int)(k32 & 63u), wy = (int)uni(div_by_rcp((uint32_t)wl, rcp_of((uint32_t)wr))), wx = wl - wy * wr;
        if (lane == 0) st.req_key[rq] = ((u64)(k32 >> 6) << 32) | ((uint32_t)wy << 16) | (uint32_t)wx;
    }
    wave_sync(); // the arena is free again
    return true;
}

// A result of the dense pre-pass (me_dense.inl): the pre-HME strips and the HME level-0 regions of every block are searched ahead of the
// per-block kernel by a kernel of their own -- their windows depend on the block's position and the picture distance only, never on
// earlier results -- which leaves one 16-byte slot per (block, searched reference, kind): the search's key and the geometry it was made
// for.  A stage that is about to push such a search takes the key when the geometry is the one it wants (always, by construction: both
// sides call prehme_geometry / hme_level_geometry; the compare is what makes a slot the pre-pass did not fill -- an edge block, a
// configuration it does not cover -- fall back to the search itself).
constexpr int kDenseKinds = SVT_HIP_ME_DENSE_KINDS; // 0, 1: pre-HME strips; 2 + 2 h + w: level-0 regions
__device__ __forceinline__ uint32_t dense_pack(int a, int b) { return (uint32_t)(uint16_t)a | ((uint32_t)(uint16_t)b << 16); }
__device__ __forceinline__ void dense_take(St &st, const MeDenseSlot *dl, int idx, int req, const SearchGeo &g, uint32_t &n_hit, uint32_t &n_miss) {
    if (!dl) return;
    const MeDenseSlot s = dl[idx];
    if (s.org == dense_pack(g.ox, g.oy) && s.size == dense_pack(g.sa_w, g.sa_h)) {
        st.req[req].done = 1;
        st.req_key[req]  = s.key == ~0ull ? ((0xffffffull << 32) | 0xffffffffull) : s.key; // no position evaluated
        n_hit++;
    } else
        n_miss++;
}

// slot < 0: append (serial pushes by lane 0); otherwise the caller owns st.req[slot] and sets st.nreq itself (lane-parallel pushes)
__device__ __forceinline__ void push_req(St &st, const uint8_t *win, uint32_t stride, int sa_w, int sa_h, int bw, int bh, int rs,
                                         int level, int skip, int slot = -1) {
    Req &r      = st.req[slot >= 0 ? slot : st.nreq++];
    r.win       = win;
    r.stride    = stride;
    r.sa_w      = (int16_t)sa_w;
    r.sa_h      = (int16_t)sa_h;
    r.bw        = (uint8_t)bw;
    r.bh        = (uint8_t)bh;
    r.rs        = (uint8_t)rs;
    r.level     = (uint8_t)level;
    r.skip_even = (uint8_t)(skip && bw == 16 && bh <= 16);
    r.done      = 0;
}

// The svt_sad_loop_kernel call made by the HME levels and pre-HME (e.g. motion_estimation.c:891-909)
__device__ __forceinline__ void push_hme_req(St &st, CParams &p, int level, CPlane &rp, int org_x, int org_y,
                                             int bw, int bh, int ox, int oy, int sa_w, int sa_h, int skip, int slot = -1) {
    const int full = (p.cfg.hme_search_method == 1);
    push_req(st, plane_at(rp, org_x + ox, org_y + oy), rp.stride, sa_w, sa_h, bw, full ? bh : (bh >> 1), full ? 1 : 2, level, skip, slot);
}

// ---------------------------------------------------------------------------------------------
// Integer search: 85 square PUs per position
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t dpp_quad_sum(uint32_t v) {
    // sum over the 4 lanes of a quad, result in all 4 lanes
    uint32_t t = v + (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
    return t + (uint32_t)__builtin_amdgcn_mov_dpp((int)t, 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, true);
}

__device__ __forceinline__ uint32_t sum16_of_quads(uint32_t v) {
    // v is uniform inside each quad; returns the sum of the 4 quads of each 16-lane row, in all 16 lanes
    uint32_t t = v + (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141 /* row_half_mirror */, 0xF, 0xF, true);
    return t + (uint32_t)__builtin_amdgcn_mov_dpp((int)t, 0x140 /* row_mirror */, 0xF, 0xF, true);
}

__device__ __forceinline__ void upd(u64 &best, uint32_t sad, uint32_t ord) {
    const u64 k = ((u64)sad << 32) | ord;
    best        = k < best ? k : best;
}

// the wave: the window whose top-left sample is pix0 displaced by (wx0, wy0), sample by sample, coordinates clamped into the
// padded plane.  Rare (see run_me_searches): kept out of line so that it costs the common path no registers.
// (its arguments are values: a request passed by reference would have to live in scratch memory on the common path as well)
__device__ __attribute__((noinline)) void stage_clamped(uint8_t *win, const uint8_t *pix0, uint32_t stride, int min_x, int max_x, int min_y, int max_y, int wx0, int wy0,
                                                        int pitch, int rows) {
    for (int i = threadIdx.x; i < pitch * rows; i += kThreads) {
        const int row = i / pitch, cb = i - row * pitch;
        const int x = imin(imax(wx0 + cb, min_x), max_x), y = imin(imax(wy0 + row, min_y), max_y);
        win[row * pitch + cb] = pix0[x + (long long)y * stride];
    }
}

// the wave: integer search for the refs in list[0 .. count).  `merge` semantics follow the reference: strict
// `<` against what is already in best_sad (initial MAX_SAD_VALUE, or the probe's result).
//
// Lane <-> (row group g = lane / 16, 16x16 PU pu = lane % 16 in quad-tree order): a step evaluates FOUR search rows (one per group) x four
// columns (one v_qsad_pk_u16_u8 quad).  A lane adds up the four 8x8 SADs of its 16x16 block itself (plain adds), the 32x32 sums are quad
// sums, the 64x64 sums 16-lane row sums (DPP): 3 x 4 adds + 4 DPP per position instead of 6 DPP per position and lane.  The running bests
// are packed keys (sad << 12 | position inside the tile, raster order): one v_lshl_or + one v_min per PU size and position -- the first
// minimum in raster order wins, like the reference's strict `<` -- against three instructions for a compare and two selects.  Measured on
// the bench launch (one-wave-per-8x8 layout before): ~20 VALU instructions per position and wave -> ~8.
constexpr int kMeInFlight = 4; // window vectors a lane has in flight while staging a tile (8 measured the same: 35 more spilled registers pay for the saved round trip)
constexpr int kMeOrdBits = 12; // positions of one tile: at most 4096 (the tile sizing below); 64x64 SADs stay below 2^20
__device__ __forceinline__ void run_me_searches(Shared &sh, CParams &p, const MeReq *list, int count, uint32_t *bsad, uint32_t *bmv, int r0n PROF_PARAM) {
    const int lane = threadIdx.x;
    const int sub  = (p.cfg.me_search_method == 0);
    const int nrow = sub ? 4 : 8, rstep = sub ? 2 : 1; // rows of an 8x8 block that are compared
    const int g = lane >> 4, pu = lane & 15;
    const int bx = (pu & 1) | ((pu >> 1) & 2), by = ((pu >> 1) & 1) | ((pu >> 2) & 2); // the 16x16 block: bits x0 y0 x1 y1
    const uint8_t *srow = &LDS(sh.src64)[bx * 16]; // this lane's source rows: row r of the 64x64 block at ((r >> cshift) * kSrc64Pitch)
    for (int mi = 0; mi < count; mi++) {
        MeReq m;
        { // uniform: through SGPRs
            const MeReq &mm = list[mi];
            m.pix0 = uni_ptr(mm.pix0); m.stride = uni(mm.stride);
            m.ox = (int16_t)uni((uint32_t)mm.ox); m.oy = (int16_t)uni((uint32_t)mm.oy); m.sa_w = (int16_t)uni((uint32_t)mm.sa_w); m.sa_h = (int16_t)uni((uint32_t)mm.sa_h);
            m.li = (uint8_t)uni(mm.li); m.ri = (uint8_t)uni(mm.ri); m.probe = 0; m.pad = 0;
            m.min_x = (int16_t)uni((uint32_t)mm.min_x); m.max_x = (int16_t)uni((uint32_t)mm.max_x); m.min_y = (int16_t)uni((uint32_t)mm.min_y); m.max_y = (int16_t)uni((uint32_t)mm.max_y);
        }
        u64 b8[4] = {~0ull, ~0ull, ~0ull, ~0ull}, b16 = ~0ull, b32 = ~0ull, b64 = ~0ull; // (sad << 32 | position in the search area), per PU of this lane
        // tile the search area by rows (and columns) so that the window fits the arena and a tile's positions the keys
        const int W = m.sa_w, H = m.sa_h;
        int       tw = W, th = H;
        auto me_pitch = [](int shift, int ww) { return ((shift + ww - 1 + 64 + 15) & ~15) + 16; };
        auto wbytes = [&](int ww, int hh) { return (uint32_t)me_pitch(0, ww) * (uint32_t)(hh - 1 + 64); };
        while (th > 1 && (wbytes(tw, th) > (uint32_t)kWinBytes || tw * th > (1 << kMeOrdBits))) th = (th + 1) >> 1;
        while (tw > 8 && (wbytes(tw, th) > (uint32_t)kWinBytes || tw * th > (1 << kMeOrdBits))) tw = ((tw >> 1) + 7) & ~7;
        for (int y0 = 0; y0 < H; y0 += th)
            for (int x0 = 0; x0 < W; x0 += tw) {
                const int w = imin(tw, W - x0), h = imin(th, H - y0);
                const uint8_t *gwin  = m.pix0 + (m.ox + x0) + (long long)(m.oy + y0) * m.stride;
                const int pitch = me_pitch(0, w); // 16-byte rows (fetched with unaligned 16-byte loads: position x0 sits on LDS byte 0 of its row)
                const int rows        = h - 1 + 64;
                const int vec_per_row = pitch >> 4;
                const float vpr_rcp   = rcp_of((uint32_t)vec_per_row);
                wave_sync(); // previous tile fully consumed
                // The reference's 1-point probe uses the unclipped search centre (motion_estimation.c:1391-1406): a centre far outside
                // the picture would take these reads past the padded plane (undefined in the reference; a fault here).  Such a tile
                // -- uniform test -- is staged sample by sample with coordinates clamped to the plane's edge instead.
                // (The vector loads of the fast path run up to 15 bytes before and 31 bytes behind the samples they need: the planes
                // carry that much slack around every row, pictures.hip.)
                const int wx0 = m.ox + x0, wy0 = m.oy + y0;
                if (m.ox + x0 >= m.min_x && m.ox + x0 + w - 1 + 63 <= m.max_x && wy0 >= m.min_y && wy0 + rows - 1 <= m.max_y) {
                    const int nvec = vec_per_row * rows;
                    // kMeInFlight independent loads per lane before the first is waited for
                    for (int b0 = 0; b0 < nvec; b0 += kMeInFlight * kThreads) { // uniform
                        const int base = b0 + lane;
                        int  k[kMeInFlight];
                        V4   v[kMeInFlight];
#pragma unroll
                        for (int j = 0; j < kMeInFlight; j++) {
                            if (b0 + j * kThreads < nvec) { // uniform: whole instructions beyond the window are skipped (lanes beyond it inside one repeat the last vector)
                                k[j] = imin(base + j * kThreads, nvec - 1);
                                const int row = (int)div_by_rcp((uint32_t)k[j], vpr_rcp), c = k[j] - row * vec_per_row;
                                const V4U q4 = *reinterpret_cast<GV4U *>(reinterpret_cast<uintptr_t>(gwin + (long long)row * m.stride + c * 16));
                                v[j] = V4{q4.x, q4.y, q4.z, q4.w};
                            }
                        }
#pragma unroll
                        for (int j = 0; j < kMeInFlight; j++)
                            if (b0 + j * kThreads < nvec) *reinterpret_cast<V4 *>(&LDS(sh.win)[k[j] * 16]) = v[j];
                    }
                } else {
                    stage_clamped(LDS(sh.win), m.pix0, m.stride, m.min_x, m.max_x, m.min_y, m.max_y, wx0, wy0, pitch, rows);
                }
                wave_sync();
                PROF(46); // (diagnostic build: window staged)
                const int ng = (w + 3) >> 2;
                uint32_t k8[4] = {~0u, ~0u, ~0u, ~0u}, k16 = ~0u, k32 = ~0u, k64 = ~0u; // running bests of the tile: sad << 12 | y * w + x
                const int kshift = kMeOrdBits + sub; // (a sub-sampled SAD counts twice)
                for (int y = 0; y < h; y += 4) { // uniform: four search rows per step, one per lane group
                    const bool live = y + g < h;
                    const int  yy   = live ? y + g : h - 1; // (a dead group reads rows of the tile, its sums go nowhere)
                    const uint8_t *wrow = &LDS(sh.win)[(yy + by * 16) * pitch + bx * 16];
                    const uint32_t ord0 = (uint32_t)(yy * w);
                    for (int gq = 0; gq < ng; gq++) { // uniform: a quad of columns
                        const uint8_t *wp = wrow + 4 * gq;
                        u64 acc[4] = {0, 0, 0, 0}; // the block's four 8x8 SADs (x0 y0 order), four positions each
#pragma unroll
                        for (int half = 0; half < 2; half++) // the upper / lower pair of 8x8 blocks
#pragma unroll
                            for (int k = 0; k < 8; k++) {
                                if (k < nrow) { // uniform
                                    const int r = 8 * half + k * rstep; // row of the 16x16 block
                                    const uint32_t *wr = reinterpret_cast<const uint32_t *>(wp + r * pitch);
                                    const uint32_t d0 = wr[0], d1 = wr[1], d2 = wr[2], d3 = wr[3], d4 = wr[4];
                                    const uint4 sv = *reinterpret_cast<const uint4 *>(srow + ((by * 16 + r) >> sh.cshift) * kSrc64Pitch);
                                    acc[2 * half] = __builtin_amdgcn_qsad_pk_u16_u8(((u64)d1 << 32) | d0, sv.x, acc[2 * half]);
                                    acc[2 * half] = __builtin_amdgcn_qsad_pk_u16_u8(((u64)d2 << 32) | d1, sv.y, acc[2 * half]);
                                    acc[2 * half + 1] = __builtin_amdgcn_qsad_pk_u16_u8(((u64)d3 << 32) | d2, sv.z, acc[2 * half + 1]);
                                    acc[2 * half + 1] = __builtin_amdgcn_qsad_pk_u16_u8(((u64)d4 << 32) | d3, sv.w, acc[2 * half + 1]);
                                }
                            }
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int x = 4 * gq + i;
                            uint32_t v8[4];
#pragma unroll
                            for (int j = 0; j < 4; j++) v8[j] = (uint32_t)((acc[j] >> (16 * i)) & 0xFFFF);
                            const uint32_t v16 = (v8[0] + v8[1]) + (v8[2] + v8[3]);
                            const uint32_t v32 = dpp_quad_sum(v16);
                            const uint32_t v64 = sum16_of_quads(v32);
                            if (x < w && live) { // x: wave-uniform; live: whole 16-lane rows (the DPP sums above never leave a row)
                                const uint32_t ord = ord0 + (uint32_t)x;
#pragma unroll
                                for (int j = 0; j < 4; j++) k8[j] = umin32(k8[j], (v8[j] << kshift) | ord);
                                k16 = umin32(k16, (v16 << kshift) | ord);
                                k32 = umin32(k32, (v32 << kshift) | ord);
                                k64 = umin32(k64, (v64 << kshift) | ord);
                            }
                        }
                    }
                }
                PROF(47); // (diagnostic build: positions evaluated)
                // the four row groups meet; tiles merge by (sad, position in the search area)
                const float w_rcp = rcp_of((uint32_t)w);
                auto fold = [&](u64 &best, uint32_t k) {
                    k = umin32(k, (uint32_t)__shfl_xor((int)k, 16, 64));
                    k = umin32(k, (uint32_t)__shfl_xor((int)k, 32, 64));
                    if (k != ~0u) {
                        const uint32_t ordl = k & ((1u << kMeOrdBits) - 1u), ty = div_by_rcp(ordl, w_rcp), tx = ordl - ty * (uint32_t)w;
                        upd(best, k >> kMeOrdBits, (uint32_t)((y0 + (int)ty) * W + (x0 + (int)tx)));
                    }
                };
#pragma unroll
                for (int j = 0; j < 4; j++) fold(b8[j], k8[j]);
                fold(b16, k16); fold(b32, k32); fold(b64, k64);
            }
        // the wave's bests -> this reference's rows of best_sad / best_mv, PU index in the reference's n_idx order: 0 = 64x64, 1..4, 5..20, 21..84.
        // Each PU has one owner lane (of group 0: after the fold every group holds the same keys), so the compare-and-store below is race-free.
        uint32_t *rs_ = bsad + ((m.li ? r0n : 0) + m.ri) * 85, *rm_ = bmv + ((m.li ? r0n : 0) + m.ri) * 85;
        auto merge = [&](int n, u64 k) {
            const uint32_t sad = (uint32_t)(k >> 32);
            if (k != ~0ull && sad < rs_[n]) {
                const uint32_t ord = (uint32_t)k;
                const int      yy = (int)(ord / (uint32_t)W), xx = (int)(ord - (uint32_t)yy * (uint32_t)W);
                rs_[n] = sad;
                rm_[n] = ((uint32_t)(m.oy + yy) << 16) | (uint16_t)(m.ox + xx);
            }
        };
        if (g == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) merge(21 + 4 * pu + j, b8[j]);
            merge(5 + pu, b16);
            if ((pu & 3) == 0) merge(1 + (pu >> 2), b32);
            if (pu == 0) merge(0, b64);
        }
        wave_sync();
    }
}

// ---------------------------------------------------------------------------------------------
// lane-0 control logic (restates the scalar parts of motion_estimation.c; see per-function citations)
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ bool searched(CParams &p, int li) { return p.desc.temporal_layer_index > 0 || li == 0; }

__device__ void set_hme_all(St &st, CParams &p, int lvl, int li, int ri, int x, int y, uint32_t sad) {
    for (int h = 0; h < p.cfg.num_hme_sa_h; h++)
        for (int w = 0; w < p.cfg.num_hme_sa_w; w++) {
            st.hx[lvl][li][ri][w][h] = (int16_t)x;
            st.hy[lvl][li][ri][w][h] = (int16_t)y;
            st.hs[lvl][li][ri][w][h] = sad;
        }
}

// get_hme_l0_search_area, motion_estimation.c:1800-1868
// Works on a copy of the block's total level-0 area (the reference divides its context's copy in place and restores it after
// each reference, motion_estimation.c:1948-1953,2031-2034: every reference starts from the same base).  (mvx, mvy) = list 0 / reference 0's
// level-0 result of region (0, 0), read only when both reduce_hme_l0_sr thresholds are set.
template <class P> __host__ __device__ __forceinline__ void hme_l0_area(P &p, int li, int ri, uint32_t dist, int mvx, int mvy, int &sa_w, int &sa_h) {
    auto &c = p.cfg;
    struct { SvtHipSearchAreaMinMax hme_l0_sa; } st;
    st.hme_l0_sa.sa_min.width = (uint16_t)c.hme_l0_sa.sa_min.width; st.hme_l0_sa.sa_min.height = (uint16_t)c.hme_l0_sa.sa_min.height;
    st.hme_l0_sa.sa_max.width = (uint16_t)c.hme_l0_sa.sa_max.width; st.hme_l0_sa.sa_max.height = (uint16_t)c.hme_l0_sa.sa_max.height;
    if (c.enable_me_sr_adjustment && c.distance_based_hme_resizing) {
        int is_hor = 1, is_ver = 1, is_still = 0;
        if (c.reduce_hme_l0_sr_th_min && c.reduce_hme_l0_sr_th_max && (li || ri)) {
            is_ver   = iabs(mvx) < c.reduce_hme_l0_sr_th_min && iabs(mvy) > c.reduce_hme_l0_sr_th_max;
            is_hor   = iabs(mvx) > c.reduce_hme_l0_sr_th_max && iabs(mvy) < c.reduce_hme_l0_sr_th_min;
            is_still = iabs(mvx) < c.reduce_hme_l0_sr_th_min * 3 && iabs(mvy) < c.reduce_hme_l0_sr_th_min * 3;
        }
        int xo = is_hor ? 1 : 2, yo = is_ver ? 1 : 2;
        if (c.enable_me_sr_adjustment == 2 && is_still) xo = yo = 4;
        st.hme_l0_sa.sa_min.width  = (uint16_t)(st.hme_l0_sa.sa_min.width / (xo + ri));
        st.hme_l0_sa.sa_min.height = (uint16_t)(st.hme_l0_sa.sa_min.height / (yo + ri));
        st.hme_l0_sa.sa_max.width  = (uint16_t)(st.hme_l0_sa.sa_max.width / (xo + ri));
        st.hme_l0_sa.sa_max.height = (uint16_t)(st.hme_l0_sa.sa_max.height / (yo + ri));
    }
    const int f = (int)scaled_distance(dist);
    int       w = (int16_t)(st.hme_l0_sa.sa_min.width / c.num_hme_sa_w);
    w           = (int16_t)imin(((w * f) + 15) & ~15, ((st.hme_l0_sa.sa_max.width / c.num_hme_sa_w) + 15) & ~15);
    int h       = (int16_t)(st.hme_l0_sa.sa_min.height / c.num_hme_sa_h);
    h           = (int16_t)imin(h * f, st.hme_l0_sa.sa_max.height / c.num_hme_sa_h);
    sa_w        = w;
    sa_h        = h;
}
__device__ void hme_l0_search_area(const St &cst, CParams &p, int li, int ri, uint32_t dist, int &sa_w, int &sa_h) {
    hme_l0_area(p, li, ri, dist, cst.hx[0][0][0][0][0], cst.hy[0][0][0][0][0], sa_w, sa_h);
}

// ---- search geometry shared by the per-block kernel and the dense pre-pass (me_dense.inl): origin (displacement of search index (0, 0)
// from the co-located block) and size of a search after the reference's clipping to the padded plane ----
// pre-HME strip `sri` of (list, reference): prehme_b64 (motion_estimation.c:1722-1796) sizes it from the picture distance, prehme_core
// (:1568-1666) centres and clips it; org_x / org_y = the block's full-resolution origin
template <class P> __host__ __device__ __forceinline__ void prehme_size(P &p, int li, int ri, int sri, int &sa_w, int &sa_h) {
    auto &c = p.cfg;
    const uint32_t f = scaled_distance(ref_distance(p, li, ri));
    sa_w = (int16_t)(uint16_t)imin((int)(c.prehme_sa_cfg[sri].sa_min.width * f), c.prehme_sa_cfg[sri].sa_max.width);
    sa_h = (int16_t)(uint16_t)imin((int)(c.prehme_sa_cfg[sri].sa_min.height * f), c.prehme_sa_cfg[sri].sa_max.height);
}
__device__ __forceinline__ SearchGeo prehme_geometry(CParams &p, int li, int ri, int sri, uint32_t org_x, uint32_t org_y) {
    int sa_w, sa_h;
    prehme_size(p, li, ri, sri, sa_w, sa_h);
    CPlane &rp = p.ref[li][ri].lvl[0];
    const int ox16 = (int16_t)org_x >> 2, oy16 = (int16_t)org_y >> 2;
    int ox = -(int16_t)(sa_w >> 1), oy = -(int16_t)(sa_h >> 1);
    clip_axis(ox16, ox, sa_w, rp.org_x - 1, rp.width);
    clip_axis(oy16, oy, sa_h, rp.org_y - 1, rp.height);
    SearchGeo g = {ox, oy, sa_w, sa_h};
    return g;
}

// hme_level_0/1/2 geometry (motion_estimation.c:820-1113); org_x / org_y = the block's origin on the level's plane
__device__ __forceinline__ SearchGeo hme_level_geometry(CParams &p, int level, CPlane &rp, int org_x, int org_y, int sa_w, int sa_h, int cx, int cy, int sr_w, int sr_h) {
    sa_w = (int16_t)((sa_w + 7) & ~7);
    int pad_w, pad_h, ox, oy;
    if (level == 2) { pad_w = pad_h = 63; } else { pad_w = rp.org_x - 1; pad_h = rp.org_y - 1; }
    if (level == 0) {
        ox = -(int16_t)((sa_w * p.cfg.num_hme_sa_w) >> 1) + (int16_t)(sa_w * sr_w);
        oy = -(int16_t)((sa_h * p.cfg.num_hme_sa_h) >> 1) + (int16_t)(sa_h * sr_h);
    } else {
        ox = -(sa_w >> 1) + cx;
        oy = -(sa_h >> 1) + cy;
    }
    clip_axis(org_x, ox, sa_w, pad_w, rp.width);
    sa_w = (sa_w < 8) ? sa_w : (sa_w & ~7);
    clip_axis(org_y, oy, sa_h, pad_h, rp.height);
    SearchGeo g = {ox, oy, sa_w, sa_h};
    return g;
}

struct HmeGeom { int16_t ox, oy; };

// hme_level_0/1/2: pushes the search and returns its origin
__device__ HmeGeom push_hme_level(St &st, CParams &p, int level, CPlane &rp, int org_x, int org_y, int bw, int bh,
                                  int sa_w, int sa_h, int cx, int cy, int sr_w, int sr_h, int slot = -1) {
    const SearchGeo sg = hme_level_geometry(p, level, rp, org_x, org_y, sa_w, sa_h, cx, cy, sr_w, sr_h);
    push_hme_req(st, p, level, rp, org_x, org_y, bw, bh, sg.ox, sg.oy, sg.sa_w, sg.sa_h, 0, slot);
    HmeGeom g = {(int16_t)sg.ox, (int16_t)sg.oy};
    return g;
}

__device__ __forceinline__ void key_to_result(u64 key, int full, uint32_t &sad, int &x, int &y) {
    sad = (uint32_t)(key >> 32);
    if ((uint32_t)key == 0xffffffffu) { x = 0; y = 0; } // no position evaluated (known divergence)
    else { y = (int)((key >> 16) & 0xFFFF); x = (int)(key & 0xFFFF); }
    if (!full) sad *= 2;
}

// zero-MV style SAD request: svt_nxm_sad_kernel on every other row (get_zz_sad, motion_estimation.c:1667-1689)
__device__ __forceinline__ void push_zz_req(St &st, CPlane &rp, int dx, int dy, int slot = -1) {
    push_req(st, plane_at(rp, (int)st.org_x + dx, (int)st.org_y + dy), rp.stride, 1, 1, (int)st.b64_w, (int)st.b64_h >> 1, 2, 2, 0, slot);
}

} // namespace


// =================================================================================================
// The kernel
// =================================================================================================
// a wave in the control code between two searches asks for issue priority over the waves that are evaluating
#ifndef SVT_ME_CTRL_PRIO
#define SVT_ME_CTRL_PRIO 2
#endif
#define CTRL_PRIO(x) __builtin_amdgcn_s_setprio(x)

// `launch_flags` bit 0: jobs come from the list of deferred blocks (kMeFull at the end of a staged launch).
constexpr size_t kPersistBytes = (offsetof(St, req) + 15) & ~(size_t)15; // the head of St that travels between the kernels of a staged launch
template <int MODE>
__device__ __forceinline__ void me_b64_body(const MeBatchHeader *__restrict__ ghdr, const MeKernelParams *__restrict__ gparams, const uint32_t launch_flags) {
    St     &st = *reinterpret_cast<St *>(g_lds);
    // launch parameters: read-only, uniform addresses -> scalar loads through the constant cache, values in SGPRs
    typedef const SVT_CONST_AS MeBatchHeader CHeader;
    CHeader  &hdr = *(CHeader *)ghdr;
    const int tid = threadIdx.x; // the lane: a workgroup is one wave
    // MeContext.p_sb_best_sad / p_sb_best_mv rows of the (list, reference) pairs in use, behind the fixed part of the LDS slice
    const int       cshift = (int)hdr.cshift;
    const bool      has_dense = hdr.dense != nullptr; // uniform
    const LdsLayout lay    = lds_layout((int)hdr.n_slot, cshift, has_dense ? 1 : 0, MODE);
    uint32_t *const bsad = reinterpret_cast<uint32_t *>(g_lds + lay.bsad);
    uint32_t *const bmv  = reinterpret_cast<uint32_t *>(g_lds + lay.bmv);
    const MeDenseSlot *const dense_lds = has_dense ? reinterpret_cast<const MeDenseSlot *>(g_lds + lay.dense) : nullptr;
    uint32_t n_hit = 0, n_miss = 0; // searches taken from / not found in the dense pre-pass (per lane; summed when the wave retires)
    Shared sh = {st, lay.src64, lay.src32, lay.src16, lay.win, cshift};

    // XCD-aware work pull: queue q holds a contiguous band of b64 rows; start with this XCD's own band
    uint32_t xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    int queue_probe = 0; // only lane 0's copy is used
    // every kernel of a staged launch drains the same ranges with counters of its own; the list kernels walk a list filled by an earlier kernel
    // The one-kernel form balances its long, uneven blocks through the band queues (a returning device-scope atomic per block: nothing next to
    // ~100 us of work).  The short kernels of a staged launch cannot afford one per job -- measured: ~0.1 us per wave and kernel, every one of
    // them serialised at the memory side -- so their waves take a contiguous share of the job range instead (neighbouring blocks, whose windows
    // overlap, still follow each other on one CU); only the few waves of the list kernels pull their (rare) jobs with an atomic.
    constexpr int kList = 0;
    const bool list_mode = (launch_flags & 1u) != 0;
    const uint32_t n_total = hdr.job_base[hdr.n_pictures];
    // Wave i of a staged kernel walks band (i % 8) of the launch -- the band its XCD would pull from in the one-kernel form, workgroups being dealt
    // round-robin over the XCDs -- with the band's other waves, interleaved: what the chip works on at one time is a run of neighbouring blocks,
    // whose windows overlap in that XCD's L2.
    uint32_t share_next = 0, share_end = 0, share_step = 1;
    if (MODE != kMeFull && !list_mode) {
        if (gridDim.x >= SVT_HIP_ME_QUEUES) {
            const uint32_t q = blockIdx.x % SVT_HIP_ME_QUEUES;
            share_step = (gridDim.x - q + SVT_HIP_ME_QUEUES - 1) / SVT_HIP_ME_QUEUES;
            share_next = hdr.queue_begin[q] + blockIdx.x / SVT_HIP_ME_QUEUES;
            share_end  = hdr.queue_begin[q + 1];
        } else {
            share_step = gridDim.x; share_next = blockIdx.x; share_end = n_total;
        }
    }
    auto fetch_job = [&]() {
        int job = -1;
        if (list_mode) {
            // (a look before the pull: the lists are empty or short, and a returning atomic per idle wave on one word is ~0.1 us each, in series)
            const uint32_t n = __hip_atomic_load(&hdr.queue_head[SVT_HIP_ME_LIST_COUNT(kList)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_load(&hdr.queue_head[SVT_HIP_ME_LIST_CURSOR(kList)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n) return job;
            const uint32_t k = atomicAdd(&hdr.queue_head[SVT_HIP_ME_LIST_CURSOR(kList)], 1u);
            if (k < n) job = (int)hdr.lists[kList][k];
            return job;
        }
        if (MODE != kMeFull) {
            if (share_next < share_end) { job = (int)share_next; share_next += share_step; }
            return job;
        }
        while (queue_probe < SVT_HIP_ME_QUEUES) {
            const int      q  = (int)((xcc + queue_probe) & 7);
            const uint32_t lo = hdr.queue_begin[q], hi = hdr.queue_begin[q + 1];
            if (lo < hi) {
                const uint32_t k = atomicAdd(&hdr.queue_head[q], 1u);
                if (lo + k < hi) { job = (int)(lo + k); break; }
            }
            queue_probe++;
        }
        return job;
    };
    // a block's travelling state and requests (staged launches)
    auto stage_base = [&](int gjob) { return hdr.stage + (size_t)gjob * SVT_HIP_ME_STAGE_BYTES; };
    auto export_state = [&](int gjob) {
        uint4 *dst = reinterpret_cast<uint4 *>(stage_base(gjob));
        const uint4 *src = reinterpret_cast<const uint4 *>(g_lds);
        for (int i = tid; i < (int)(kPersistBytes / 16); i += kThreads) dst[i] = src[i];
    };
    auto import_state = [&](int gjob) {
        const uint4 *src = reinterpret_cast<const uint4 *>(stage_base(gjob));
        uint4 *dst = reinterpret_cast<uint4 *>(g_lds);
        for (int i = tid; i < (int)(kPersistBytes / 16); i += kThreads) dst[i] = src[i];
    };
    static_assert(kPersistBytes + kMaxReq * (sizeof(Req) + sizeof(u64)) <= SVT_HIP_ME_STAGE_BYTES, "a block's record holds its state, its requests and their results");
    auto export_reqs = [&](int gjob) { // st.req[0 .. nreq) (nreq travels with the state)
        uint4 *dst = reinterpret_cast<uint4 *>(stage_base(gjob) + kPersistBytes);
        const uint4 *src = reinterpret_cast<const uint4 *>(st.req);
        static_assert(sizeof(Req) == 24 && (kMaxReq * sizeof(Req)) % 16 == 0, "requests are copied as 16-byte words");
        for (int i = tid; i < (int)(kMaxReq * sizeof(Req) / 16); i += kThreads) dst[i] = src[i];
    };
    auto import_reqs = [&](int gjob) {
        const uint4 *src = reinterpret_cast<const uint4 *>(stage_base(gjob) + kPersistBytes);
        uint4 *dst = reinterpret_cast<uint4 *>(st.req);
        for (int i = tid; i < (int)(kMaxReq * sizeof(Req) / 16); i += kThreads) dst[i] = src[i];
    };
    auto keys_of = [&](int gjob) { return reinterpret_cast<u64 *>(stage_base(gjob) + kPersistBytes + kMaxReq * sizeof(Req)); };
    PROF_DECL;

    for (;;) {
        // ---- fetch the next b64 job (fetching ahead was measured slower: it defeats the queues' load balancing) -------
        int gjob = 0;
        if (tid == 0) gjob = fetch_job();
        gjob = __builtin_amdgcn_readfirstlane(gjob);
        if (gjob < 0) break; // the exit condition every wave reaches: all queues empty
        PROF(0);
        // picture of this job (uniform): its parameter block is read with scalar loads
        int pic = 0;
        while (pic + 1 < (int)hdr.n_pictures && (uint32_t)gjob >= hdr.job_base[pic + 1]) pic++;
        pic = __builtin_amdgcn_readfirstlane(pic);
        CParams &p = ((CParams *)gparams)[pic];
        auto &c = p.cfg;
        auto &d = p.desc;
        const int full_hme = (c.hme_search_method == 1);
        const int nl       = d.num_of_list_to_search;
        const bool mctf    = c.me_type == 1; // ME_MCTF: no reference pruning, unscaled distance, HME-SAD early exit, search-level results only
        const int job      = gjob - (int)hdr.job_base[pic];
        const uint32_t bxi = (uint32_t)job % p.w64, byi = p.row0 + (uint32_t)job / p.w64;
        const uint32_t b   = bxi + byi * p.w64;
        const int r0n      = d.num_of_ref_pic_to_search[0]; // row of (list, ref) in best_sad / best_mv: (list ? r0n : 0) + ref
#define BEST_SAD(li, ri) (bsad + (((li) ? r0n : 0) + (ri)) * 85)
#define BEST_MV(li, ri) (bmv + (((li) ? r0n : 0) + (ri)) * 85)
        const int n_rows   = r0n + (nl > 1 ? d.num_of_ref_pic_to_search[1] : 0);
        uint32_t *const jflag = hdr.job_flags ? hdr.job_flags + gjob : nullptr; // staged launches: how the block travels
        if constexpr (MODE != kMeFull && MODE != kMeMid1) {
            const uint32_t f = __hip_atomic_load(jflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (f & SVT_HIP_ME_JOB_DEFERRED) continue; // uniform: the whole-pipeline kernel makes this block at the end of the launch
        }
        if constexpr (me_is_search(MODE)) {
            // ---- search-only kernels: the block's requests in, their keys out ------------------------------------------------
            constexpr int lvl = MODE == kMeS1 ? 1 : 2;
            const int ox = (int)(bxi * 64), oy = (int)(byi * 64), rstep = 1 << cshift;
            PROF(24);
            import_reqs(gjob);
            if (tid == 0) st.nreq = *reinterpret_cast<const int *>(stage_base(gjob) + offsetof(St, nreq));
            if constexpr (lvl == 1) {
                for (int i = tid; i < (32 >> cshift) * 2; i += kThreads) {
                    const int row = i >> 1, cc = i & 1;
                    uint4 v; memcpy(&v, plane_at(p.cur.lvl[1], (ox >> 1) + cc * 16, (oy >> 1) + row * rstep), 16);
                    *reinterpret_cast<uint4 *>(&LDS(sh.src32)[row * kSrc32Pitch + cc * 16]) = v;
                }
            } else {
                for (int i = tid; i < (64 >> cshift) * 4; i += kThreads) {
                    const int row = i >> 2, cc = i & 3;
                    uint4 v; memcpy(&v, plane_at(p.cur.lvl[2], ox + cc * 16, oy + row * rstep), 16);
                    *reinterpret_cast<uint4 *>(&LDS(sh.src64)[row * kSrc64Pitch + cc * 16]) = v;
                }
            }
            wave_sync();
            PROF(25);
            if (st.nreq) { // uniform
                (void)run_small_searches_direct<SVT_HIP_ME_SEARCH_DEPTH>(sh PROF_ARG); // (the Mid kernel routed the block here because it qualifies)
            }
            wave_sync();
            u64 *keys = keys_of(gjob);
            if (tid < kMaxReq) keys[tid] = st.req_key[tid];
            wave_sync();
            continue;
        }

        // ---- block setup (me_process.c:183-214; motion_estimation.c:3090-3105, init_me_hme_data :3010-3071) ---
        if (tid == 0) {
            st.b64_index = b;
            st.org_x = bxi * 64; st.org_y = byi * 64;
            st.b64_w = (uint32_t)(d.aligned_width - st.org_x) < 64 ? d.aligned_width - st.org_x : 64;
            st.b64_h = (uint32_t)(d.aligned_height - st.org_y) < 64 ? d.aligned_height - st.org_y : 64;
            st.hme_l0_sa.sa_min.width = (uint16_t)c.hme_l0_sa.sa_min.width; st.hme_l0_sa.sa_min.height = (uint16_t)c.hme_l0_sa.sa_min.height;
            st.hme_l0_sa.sa_max.width = (uint16_t)c.hme_l0_sa.sa_max.width; st.hme_l0_sa.sa_max.height = (uint16_t)c.hme_l0_sa.sa_max.height;
            st.nreq = 0; st.nme = 0; st.nprobe = 0;
        }
        for (int i = tid; i < 3 * 2 * 4 * 2 * 2; i += kThreads) { (&st.hx[0][0][0][0][0])[i] = 0; (&st.hy[0][0][0][0][0])[i] = 0; (&st.hs[0][0][0][0][0])[i] = 0; }
        if constexpr (MODE == kMeFull || MODE == kMeTail)
            for (int i = tid; i < n_rows * 85; i += kThreads) { bmv[i] = 0; bsad[i] = SVT_HIP_MAX_SAD_VALUE; }
        if (tid < 8) {
            const int li = tid >> 2, ri = tid & 3;
            st.do_ref[li][ri] = 1; st.hme_sad64[li][ri] = 0xFFFFFFFFull; st.sr_divisor[li][ri] = 1; st.zz_sad[li][ri] = ~0u;
            st.hme_sc_x[li][ri] = st.hme_sc_y[li][ri] = 0;
            for (int sri = 0; sri < 2; sri++) {
                PreHme &ph = st.prehme[li][ri][sri];
                ph.valid = 0; ph.col = ph.row = 0; ph.sad = 0; ph.sa_w = ph.sa_h = 0;
                st.performed_phme[li][ri][sri] = 0;
            }
        }
        if ((MODE == kMeFull || MODE == kMeMid1) && has_dense && tid < n_rows * kDenseKinds) // this block's slots of the dense pre-pass (read by the pre-HME / level-0 stages)
            reinterpret_cast<uint4 *>(g_lds + lay.dense)[tid] = reinterpret_cast<const uint4 *>(hdr.dense)[((size_t)gjob * hdr.n_slot) * kDenseKinds + tid];
        // init_zz_sad (motion_estimation.c:2382-2437) rides on the set-up: while a lane holds a 16-byte piece of the source block it fetches the
        // same piece of every searched reference and adds up the zero-MV SAD of the even rows (get_zz_sad, :1667-1689) -- no staging, no
        // search rounds for single positions.  Sums in zz_sum[k], k = the searched (list, reference) pairs in the reference's loop order.
        const bool zz_on = (MODE == kMeFull || MODE == kMeMid1) && (c.me_early_exit_th || c.me_safe_limit_zz_th); // uniform
        const int  nzz   = zz_on ? r0n + ((nl > 1 && d.temporal_layer_index > 0) ? d.num_of_ref_pic_to_search[1] : 0) : 0;
        uint32_t   zz_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        { // source views -> LDS (every row, or the even rows only: cshift)
            const int ox = (int)(bxi * 64), oy = (int)(byi * 64), rstep = 1 << cshift;
            const int bw64 = imin(64, (int)d.aligned_width - ox), bh64 = imin(64, (int)d.aligned_height - oy);
            if constexpr (MODE != kMeMid2)
            for (int i = tid; i < (64 >> cshift) * 4; i += kThreads) {
                const int row = i >> 2, cc = i & 3;
                uint4 v; memcpy(&v, plane_at(p.cur.lvl[2], ox + cc * 16, oy + row * rstep), 16);
                if constexpr (MODE != kMeMid1) *reinterpret_cast<uint4 *>(&LDS(sh.src64)[row * kSrc64Pitch + cc * 16]) = v;
                const int y = row * rstep, x = cc * 16;
                if (nzz && !(y & 1) && y < bh64 && x < bw64) { // block widths are multiples of 8: a piece is whole or half inside
                    const bool whole = x + 8 < bw64;
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        if (k < nzz) { // uniform
                            const int li = k < r0n ? 0 : 1, ri = k < r0n ? k : k - r0n;
                            uint4 w; memcpy(&w, plane_at(p.ref[li][ri].lvl[2], ox + x, oy + y), 16);
                            uint32_t a = __builtin_amdgcn_sad_u8(v.x, w.x, zz_sum[k]);
                            a = __builtin_amdgcn_sad_u8(v.y, w.y, a);
                            if (whole) { a = __builtin_amdgcn_sad_u8(v.z, w.z, a); a = __builtin_amdgcn_sad_u8(v.w, w.w, a); }
                            zz_sum[k] = a;
                        }
                }
            }
            if constexpr (MODE == kMeFull) {
                for (int i = tid; i < (32 >> cshift) * 2; i += kThreads) {
                    const int row = i >> 1, cc = i & 1;
                    uint4 v; memcpy(&v, plane_at(p.cur.lvl[1], (ox >> 1) + cc * 16, (oy >> 1) + row * rstep), 16);
                    *reinterpret_cast<uint4 *>(&LDS(sh.src32)[row * kSrc32Pitch + cc * 16]) = v;
                }
                for (int i = tid; i < (16 >> cshift); i += kThreads) {
                    uint4 v; memcpy(&v, plane_at(p.cur.lvl[0], ox >> 2, (oy >> 2) + i * rstep), 16);
                    *reinterpret_cast<uint4 *>(&LDS(sh.src16)[i * kSrc16Pitch]) = v;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < nzz) { // uniform
                const uint32_t t = wave_sum_u32(zz_sum[k]);
                if (tid == 0) st.req_key[k] = (u64)t << 32; // where zz_post looks for the result of search k
            }
        wave_sync();

        if constexpr (MODE == kMeMid2 || MODE == kMeTail) { // the block's state as the previous kernel left it, and the keys of the searches made in between
            import_state(gjob);
            const u64 *keys = keys_of(gjob);
            if (tid < kMaxReq) st.req_key[tid] = keys[tid];
            wave_sync();
        }
        PROF(1);
        // The stages below run as one loop around a SINGLE inlined copy of run_searches / run_me_searches (the kernel
        // must stay small enough for the instruction cache shared by two CUs): each stage has a "pre" part that
        // pushes its searches and a "post" part that folds the results into the block state.
        // ---- init_zz_sad (motion_estimation.c:2382-2437) ------------------------------------------------
        auto zz_pre = [&]() {
            if (tid == 0) st.nreq = 0; // the SADs came with the block set-up
        };
        auto zz_post = [&]() { // lane <-> (list, reference); the reference's serial loops become wave reductions
            const int  li = (tid >> 2) & 1, ri = tid & 3;
            const bool valid = tid < 8 && li < nl && ri < d.num_of_ref_pic_to_search[li];
            const bool srch  = valid && searched(p, li);
            const u64  smask = __ballot(srch);
            uint32_t   z = 0xFFFFFFFFu;
            if (srch) {
                const int k = __popcll(smask & ((1ull << tid) - 1ull)); // the searched pairs in the reference's loop order
                z = (uint32_t)(st.req_key[k] >> 32) << 1;
                z = (z * 64 * 64) / (st.b64_w * st.b64_h);
                st.zz_sad[li][ri] = z;
            }
            const uint32_t best = wave_min_u32(z);
            if (d.temporal_layer_index > 0 && best < c.zz_sad_th && valid && ri >= 1)
                if ((uint32_t)((z - best) * 100) > (uint32_t)(c.zz_sad_pct * best)) st.do_ref[li][ri] = 0; // (every list is searched when the layer is above 0)
            if (c.me_safe_limit_zz_th) {
                wave_sync();
                const bool limit = d.hierarchical_levels > 0 && nl == 2 && d.temporal_layer_index >= d.hierarchical_levels &&
                    d.similar_brightness_refs && st.zz_sad[0][0] < c.me_safe_limit_zz_th && st.zz_sad[1][0] < c.me_safe_limit_zz_th;
                if (limit && valid && ri >= 1) st.do_ref[li][ri] = 0;
            }
        };
        // ---- prehme_b64 (motion_estimation.c:1693-1796) ---------------------------------------------------
        // with l1_early_exit, list 1 looks at list 0's results: one batch per list then
        auto prehme_pre = [&](int bi) { // lane <-> (list, reference, strip) in the reference's loop order; request numbers = prefix count of the pushing lanes
            const int  l_lo = c.prehme_l1_early_exit ? bi : 0, l_hi = c.prehme_l1_early_exit ? bi + 1 : nl;
            const int  li = (tid >> 3) & 1, ri = (tid >> 1) & 3, sri = tid & 1;
            const bool mine = tid < 16 && li >= l_lo && li < l_hi && ri < d.num_of_ref_pic_to_search[li];
            bool       push = false;
            if (mine) {
                st.ph_req[li][ri][sri] = 0;
                if (searched(p, li)) {
                    PreHme &ph = st.prehme[li][ri][sri];
                    bool    handled = false;
                    // check_prehme_early_exit (:1693-1720)
                    if (c.me_early_exit_th && st.zz_sad[li][ri] < c.me_early_exit_th) { ph.col = ph.row = 0; ph.sad = 0; ph.valid = 1; handled = true; }
                    if (!handled && c.prehme_l1_early_exit) {
                        const PreHme &q = st.prehme[0][ri][sri];
                        if (li == 1 && q.valid && (q.sad < 32 * 32 || (iabs(q.col) < 16 && iabs(q.row) < 16))) {
                            ph.col = (int16_t)-q.col; ph.row = (int16_t)-q.row; ph.sad = q.sad; ph.valid = 1; handled = true;
                        }
                    }
                    if (!handled && !st.do_ref[li][ri]) { ph.col = ph.row = 0; ph.sad = 0xFFFFFFFFu; handled = true; }
                    push = !handled;
                }
            }
            const u64 mask = __ballot(push);
            const int slot = __popcll(mask & ((1ull << tid) - 1ull));
            if (push) {
                // prehme_core (:1568-1666)
                PreHme &ph = st.prehme[li][ri][sri];
                CPlane &rp = p.ref[li][ri].lvl[0];
                const int ox16 = (int16_t)st.org_x >> 2, oy16 = (int16_t)st.org_y >> 2;
                const SearchGeo sg = prehme_geometry(p, li, ri, sri, st.org_x, st.org_y);
                push_hme_req(st, p, 0, rp, ox16, oy16, (int)st.b64_w >> 2, (int)st.b64_h >> 2, sg.ox, sg.oy, sg.sa_w, sg.sa_h, c.prehme_skip_search_line, slot);
                st.ph_req[li][ri][sri] = (uint8_t)(slot + 1);
                dense_take(st, dense_lds, (((li ? r0n : 0) + ri) * kDenseKinds + sri), slot, sg, n_hit, n_miss);
                ph.col = (int16_t)sg.ox; ph.row = (int16_t)sg.oy; // search origin until the result is folded in
                st.performed_phme[li][ri][sri] = 1;
            }
            if (tid == 0) st.nreq = __popcll(mask);
        };
        auto prehme_post = [&](int bi) {
            const int  l_lo = c.prehme_l1_early_exit ? bi : 0, l_hi = c.prehme_l1_early_exit ? bi + 1 : nl;
            const int  li = (tid >> 3) & 1, ri = (tid >> 1) & 3, sri = tid & 1;
            const bool mine = tid < 16 && li >= l_lo && li < l_hi && ri < d.num_of_ref_pic_to_search[li];
            const int  k = mine ? st.ph_req[li][ri][sri] : 0;
            if (k) {
                PreHme &ph = st.prehme[li][ri][sri];
                uint32_t sad; int x, y;
                key_to_result(st.req_key[k - 1], full_hme, sad, x, y);
                ph.sad = sad;
                ph.col = (int16_t)((int16_t)(x + ph.col) * 4);
                ph.row = (int16_t)((int16_t)(y + ph.row) * 4);
                ph.valid = 1;
            }
        };
        auto prehme_final = [&]() { // lane <-> (list, reference)
            wave_sync(); // the strips' results
            const int  li = (tid >> 2) & 1, ri = tid & 3;
            const bool valid = tid < 8 && li < nl && ri < d.num_of_ref_pic_to_search[li];
            uint32_t   m = 0xFFFFFFFFu;
            if (valid) {
                if (searched(p, li)) {
                    m = st.prehme[li][ri][0].sad < st.prehme[li][ri][1].sad ? st.prehme[li][ri][0].sad : st.prehme[li][ri][1].sad;
                } else { // list 1 of a base-layer picture mirrors list 0 (the reference writes list 1's entries here)
                    for (int sri = 0; sri < 2; sri++) {
                        st.prehme[1][ri][sri].col = (int16_t)-st.prehme[0][ri][sri].col;
                        st.prehme[1][ri][sri].row = (int16_t)-st.prehme[0][ri][sri].row;
                        st.prehme[1][ri][sri].sad = st.prehme[0][ri][sri].sad;
                    }
                }
            }
            const uint32_t best_sad = wave_min_u32(m);
            if (d.temporal_layer_index > 0 && best_sad < c.phme_sad_th && valid && ri >= 1 && st.do_ref[li][ri]) // (every list is searched when the layer is above 0)
                if ((uint32_t)((m - best_sad) * 100) > (uint32_t)(c.phme_sad_pct * best_sad)) st.do_ref[li][ri] = 0;
        };
        // ---- hme_level0_b64 (motion_estimation.c:1906-2036) ------------------------------------------
        // get_hme_l0_search_area reads list0/ref0's level-0 result only when both thresholds are set
        const bool l0_dep = c.enable_me_sr_adjustment && c.distance_based_hme_resizing && c.reduce_hme_l0_sr_th_min && c.reduce_hme_l0_sr_th_max;
        // Levels 0 / 1 / 2 run their lane-0 style bookkeeping on 32 lanes of wave 0 instead: lane = (list, reference, region) in
        // the reference's loop order (list, ref, h, w), so that request numbers (prefix count of the pushing lanes) match the
        // serial order.  A lane only touches its own region's hx / hy / hs entries.
        auto region_of = [&](int lane, int &li, int &ri, int &h, int &w) { li = lane >> 4; ri = (lane >> 2) & 3; h = (lane >> 1) & 1; w = lane & 1; };
        auto set_region = [&](int lvl, int li, int ri, int w, int h, int x, int y, uint32_t sad) {
            st.hx[lvl][li][ri][w][h] = (int16_t)x; st.hy[lvl][li][ri][w][h] = (int16_t)y; st.hs[lvl][li][ri][w][h] = sad;
        };
        auto l0_pre = [&](int bi) {
            const bool dep = l0_dep;
            if (tid < 64) {
                int li, ri, h, w;
                region_of(tid, li, ri, h, w);
                bool push = false;
                int  sa_w = 0, sa_h = 0;
                const bool mine = tid < 32 && li < nl && ri < d.num_of_ref_pic_to_search[li] && !(dep && ((li == 0 && ri == 0) != (bi == 0)));
                if (mine) {
                    if (h == 0 && w == 0) st.l0_req[li][ri] = 0;
                    bool handled = false;
                    if (c.me_early_exit_th && st.zz_sad[li][ri] < (c.me_early_exit_th >> 2)) { set_region(0, li, ri, w, h, 0, 0, 0); handled = true; }
                    if (!handled && c.prev_me_stage_based_exit_th) {
                        const int sri = st.prehme[li][ri][0].sad <= st.prehme[li][ri][1].sad ? 0 : 1;
                        if (st.performed_phme[li][ri][sri] && st.prehme[li][ri][sri].sad < (c.prev_me_stage_based_exit_th >> 4)) {
                            set_region(0, li, ri, w, h, st.prehme[li][ri][sri].col, st.prehme[li][ri][sri].row, st.prehme[li][ri][sri].sad);
                            handled = true;
                        }
                    }
                    if (!handled && !st.do_ref[li][ri]) { set_region(0, li, ri, w, h, 0, 0, 0xFFFFFFFFu); handled = true; }
                    if (!handled && searched(p, li)) {
                        hme_l0_search_area(st, p, li, ri, ref_distance(p, li, ri), sa_w, sa_h);
                        push = true;
                    }
                }
                const u64 mask = __ballot(push);
                const int slot = __popcll(mask & ((1ull << tid) - 1ull));
                if (push) {
                    const HmeGeom g = push_hme_level(st, p, 0, p.ref[li][ri].lvl[0], (int16_t)st.org_x >> 2, (int16_t)st.org_y >> 2, (int)st.b64_w >> 2,
                                                     (int)st.b64_h >> 2, sa_w, sa_h, 0, 0, w, h, slot);
                    st.hx[0][li][ri][w][h] = g.ox; st.hy[0][li][ri][w][h] = g.oy;
                    if (h == 0 && w == 0) st.l0_req[li][ri] = (uint8_t)(slot + 1);
                    const SearchGeo sg = {g.ox, g.oy, st.req[slot].sa_w, st.req[slot].sa_h};
                    dense_take(st, dense_lds, (((li ? r0n : 0) + ri) * kDenseKinds + 2 + h * 2 + w), slot, sg, n_hit, n_miss);
                }
                if (tid == 0) st.nreq = __popcll(mask);
            }
        };
        auto l0_post = [&](int bi) {
            const bool dep = l0_dep;
            if (tid < 64) {
                int li, ri, h, w;
                region_of(tid, li, ri, h, w);
                const bool mine = tid < 32 && li < nl && ri < d.num_of_ref_pic_to_search[li] && !(dep && ((li == 0 && ri == 0) != (bi == 0)));
                const int  k0   = mine ? st.l0_req[li][ri] : 0;
                if (k0) {
                    uint32_t sad; int x, y;
                    key_to_result(st.req_key[k0 - 1 + h * 2 + w], full_hme, sad, x, y);
                    st.hs[0][li][ri][w][h] = sad;
                    st.hx[0][li][ri][w][h] = (int16_t)((int16_t)(x + st.hx[0][li][ri][w][h]) * 4);
                    st.hy[0][li][ri][w][h] = (int16_t)((int16_t)(y + st.hy[0][li][ri][w][h]) * 4);
                }
                wave_sync(); // the four regions of a reference are in LDS now
                if (k0 && h == 0 && w == 0 && c.prehme_enable) {
                    // get_worst_quadrant (:1872-1901): the last compare does not raise the max
                    int ww = 0, wh = 0; uint32_t mx = 0;
                    if (st.hs[0][li][ri][0][0] > mx) { mx = st.hs[0][li][ri][0][0]; ww = 0; wh = 0; }
                    if (st.hs[0][li][ri][1][0] > mx) { mx = st.hs[0][li][ri][1][0]; ww = 1; wh = 0; }
                    if (st.hs[0][li][ri][0][1] > mx) { mx = st.hs[0][li][ri][0][1]; ww = 0; wh = 1; }
                    if (st.hs[0][li][ri][1][1] > mx) { ww = 1; wh = 1; }
                    const int sri = st.prehme[li][ri][0].sad <= st.prehme[li][ri][1].sad ? 0 : 1;
                    if (st.prehme[li][ri][sri].sad < st.hs[0][li][ri][ww][wh]) {
                        st.hs[0][li][ri][ww][wh] = st.prehme[li][ri][sri].sad;
                        st.hx[0][li][ri][ww][wh] = st.prehme[li][ri][sri].col;
                        st.hy[0][li][ri][ww][wh] = st.prehme[li][ri][sri].row;
                    }
                }
            }
        };
        // ---- hme_level1_b64 / hme_level2_b64 (motion_estimation.c:2041-2177) ----------------------------
        auto lvl_pre = [&](int lvl) {
            if (tid < 64) {
                int li, ri, h, w;
                region_of(tid, li, ri, h, w);
                bool push = false;
                if (tid < 32) st.lvl_req[li][ri][w][h] = 0;
                if (tid < 32 && li < nl && ri < d.num_of_ref_pic_to_search[li] && searched(p, li)) {
                    bool live = true;
                    if (lvl == 1) {
                        if (c.me_early_exit_th && st.zz_sad[li][ri] < (c.me_early_exit_th >> 2)) { set_region(1, li, ri, w, h, 0, 0, 0); live = false; }
                        else if (!st.do_ref[li][ri]) { set_region(1, li, ri, w, h, 0, 0, 0xFFFFFFFFu); live = false; }
                    }
                    if (live) {
                        const uint32_t exit_th = c.prev_me_stage_based_exit_th >> (lvl == 1 ? 5 : 2);
                        if (c.prev_me_stage_based_exit_th && st.hs[lvl - 1][li][ri][w][h] < exit_th)
                            set_region(lvl, li, ri, w, h, st.hx[lvl - 1][li][ri][w][h], st.hy[lvl - 1][li][ri][w][h], st.hs[lvl - 1][li][ri][w][h]);
                        else push = true;
                    }
                }
                const u64 mask = __ballot(push);
                const int slot = __popcll(mask & ((1ull << tid) - 1ull));
                if (push) {
                    HmeGeom g;
                    if (lvl == 1)
                        g = push_hme_level(st, p, 1, p.ref[li][ri].lvl[1], (int16_t)st.org_x >> 1, (int16_t)st.org_y >> 1, (int)st.b64_w >> 1,
                                           (int)st.b64_h >> 1, (int16_t)c.hme_l1_sa.width, (int16_t)c.hme_l1_sa.height,
                                           st.hx[0][li][ri][w][h] >> 1, st.hy[0][li][ri][w][h] >> 1, 0, 0, slot);
                    else
                        g = push_hme_level(st, p, 2, p.ref[li][ri].lvl[2], (int16_t)st.org_x, (int16_t)st.org_y, (int)st.b64_w, (int)st.b64_h,
                                           (int16_t)c.hme_l2_sa.width, (int16_t)c.hme_l2_sa.height, st.hx[1][li][ri][w][h],
                                           st.hy[1][li][ri][w][h], 0, 0, slot);
                    st.lvl_req[li][ri][w][h] = (uint8_t)(slot + 1);
                    st.hx[lvl][li][ri][w][h] = g.ox; st.hy[lvl][li][ri][w][h] = g.oy;
                }
                if (tid == 0) st.nreq = __popcll(mask);
            }
        };
        auto lvl_post = [&](int lvl) {
            if (tid < 32) {
                int li, ri, h, w;
                region_of(tid, li, ri, h, w);
                const int scale = (lvl == 1) ? 2 : 1;
                const int k = (li < nl && ri < d.num_of_ref_pic_to_search[li]) ? st.lvl_req[li][ri][w][h] : 0;
                if (k) {
                    uint32_t sad; int x, y;
                    key_to_result(st.req_key[k - 1], full_hme, sad, x, y);
                    st.hs[lvl][li][ri][w][h] = sad;
                    st.hx[lvl][li][ri][w][h] = (int16_t)((int16_t)(x + st.hx[lvl][li][ri][w][h]) * scale);
                    st.hy[lvl][li][ri][w][h] = (int16_t)((int16_t)(y + st.hy[lvl][li][ri][w][h]) * scale);
                }
            }
        };
        // ---- set_final_seach_centre_sb (:2182-2380), hme_prune_ref_and_adjust_sr (:2477-2518) -------------
        auto centre = [&]() { // lane <-> (list, reference)
            {
                const int  li = (tid >> 2) & 1, ri = tid & 3;
                const bool valid = tid < 8 && li < nl && ri < d.num_of_ref_pic_to_search[li];
                const bool srch  = valid && searched(p, li);
                int16_t sx = 0, sy = 0;
                u64     hme_sad = 0;
                if (srch && c.enable_hme_flag) {
                    int16_t cx = 0, cy = 0;
                    int lvl = -1;
                    if (c.enable_hme_level0_flag && !c.enable_hme_level1_flag && !c.enable_hme_level2_flag) lvl = 0;
                    if (c.enable_hme_level1_flag && !c.enable_hme_level2_flag) lvl = 1;
                    if (c.enable_hme_level2_flag) lvl = 2;
                    if (lvl >= 0) {
                        cx = st.hx[lvl][li][ri][0][0]; cy = st.hy[lvl][li][ri][0][0]; hme_sad = st.hs[lvl][li][ri][0][0];
                        int w = 1;
                        for (int h = 0; h < c.num_hme_sa_h; h++) {
                            for (; w < c.num_hme_sa_w; w++)
                                if (st.hs[lvl][li][ri][w][h] < hme_sad) {
                                    cx = st.hx[lvl][li][ri][w][h]; cy = st.hy[lvl][li][ri][w][h]; hme_sad = st.hs[lvl][li][ri][w][h];
                                }
                            w = 0;
                        }
                    }
                    sx = cx; sy = cy;
                }
                // the reference's local hme_sad survives across the loop's iterations: a pair that is not searched (list 1 of a base-layer
                // picture) is left with the value of the last searched one in loop order -- list 0's last reference
                const int last0 = d.num_of_ref_pic_to_search[0] - 1;
                const u64 carry = ((u64)(uint32_t)__shfl((int)(uint32_t)(hme_sad >> 32), last0, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)hme_sad, last0, 64);
                if (valid && !srch) hme_sad = carry;
                if (valid) { st.hme_sc_x[li][ri] = sx; st.hme_sc_y[li][ri] = sy; st.hme_sad64[li][ri] = hme_sad; }
                // ME_MCTF leaves after HME when list0/ref0 already matches (motion_estimation.c:3109-3113)
                if (tid == 0) st.tf_exit = mctf && hme_sad < (u64)d.tf_me_exit_th;
            }
            if (c.enable_hme_flag && !mctf) { // hme_prune_ref_and_adjust_sr (skipped for ME_MCTF, :3103,3115): lane <-> (list, reference)
                wave_sync();
                const int  li = (tid >> 2) & 1, ri = tid & 3;
                const bool slot = tid < 8; // the slots beyond the searched references hold the initial 0xFFFFFFFF / centre (0, 0): never the minimum
                const u64  hs = slot ? st.hme_sad64[li][ri] : ~0ull;
                const uint16_t th = c.prune_ref_if_hme_sad_dev_bigger_than_th;
                if (c.enable_me_hme_ref_pruning && th != 0xFFFF) {
                    const u64 best = (u64)wave_min_u32((uint32_t)(hs < 0xFFFFFFFFull ? hs : 0xFFFFFFFFull)); // hme_sad is a 32-bit quantity here
                    if (slot && ri >= 1 && (hs - best) * 100 > (u64)th * best) st.do_ref[li][ri] = 0;
                }
                if (c.enable_me_sr_adjustment && slot) {
                    if (iabs(st.hme_sc_x[li][ri]) <= c.reduce_me_sr_based_on_mv_length_th && iabs(st.hme_sc_y[li][ri]) <= c.reduce_me_sr_based_on_mv_length_th &&
                        hs < c.stationary_hme_sad_abs_th)
                        st.sr_divisor[li][ri] = c.stationary_me_sr_divisor;
                    else if (hs < c.reduce_me_sr_based_on_hme_sad_abs_th)
                        st.sr_divisor[li][ri] = c.me_sr_divisor_for_low_hme_sad;
                }
                wave_sync(); // lane 0 reads what lanes 1 .. 7 wrote
            }
        };
        // ---- integer_search_b64 (motion_estimation.c:1249-1516) --------------------------------------------
        // Refs are independent except through p_sb_best_sad[0][0][0] when enable_me_sr_adjustment == 2 (and the
        // zz early exit is off); then list0/ref0 is finished first.  Per group: (a) check_00_center SADs,
        // (b) search-area sizing and the 1-point probe for the 8x8-variance test, (c) final window, full search.
        const bool me_dep = (!c.me_early_exit_th) && c.enable_me_sr_adjustment == 2;
        auto c00_pre = [&](int gi) { // lane <-> (list, reference)
            const bool dep = me_dep;
            const int  li = (tid >> 2) & 1, ri = tid & 3;
            const bool mine = tid < 8 && li < nl && ri < d.num_of_ref_pic_to_search[li] && !(dep && ((li == 0 && ri == 0) != (gi == 0)));
            bool    push = false;
            int16_t cx = 0, cy = 0;
            if (mine) {
                st.c00_req[li][ri] = 0;
                cx = st.hme_sc_x[li][ri]; cy = st.hme_sc_y[li][ri];
                st.me_cx[li][ri] = cx; st.me_cy[li][ri] = cy;
                push = !(c.me_early_exit_th || !st.do_ref[li][ri]) && (cx != 0 || cy != 0) && d.is_ref;
            }
            const u64 mask = __ballot(push);
            const int slot = 2 * __popcll(mask & ((1ull << tid) - 1ull));
            if (push) { // check_00_center (:1139-1206)
                CPlane &rp = p.ref[li][ri].lvl[2];
                const int ox = (int16_t)st.org_x, oy = (int16_t)st.org_y;
                if (ox + cx < -63) cx = (int16_t)(-63 - ox);
                if (ox + cx > rp.width - 1) cx = (int16_t)(cx - ((ox + cx) - (rp.width - 1)));
                if (oy + cy < -63) cy = (int16_t)(-63 - oy);
                if (oy + cy > rp.height - 1) cy = (int16_t)(cy - ((oy + cy) - (rp.height - 1)));
                st.me_cx[li][ri] = cx; st.me_cy[li][ri] = cy;
                push_zz_req(st, rp, 0, 0, slot);
                push_zz_req(st, rp, cx, cy, slot + 1);
                st.c00_req[li][ri] = (uint8_t)(slot + 2); // index of the second request + 1
            }
            if (tid == 0) st.nreq = 2 * __popcll(mask);
        };
        auto probe_pre = [&](int gi) { // lane <-> (list, reference); list entries in the reference's loop order (prefix count of the lanes that search)
            const bool dep = me_dep;
            const int  li = (tid >> 2) & 1, ri = tid & 3;
            const bool mine = tid < 8 && li < nl && ri < d.num_of_ref_pic_to_search[li] && !(dep && ((li == 0 && ri == 0) != (gi == 0))) && st.do_ref[li][ri];
            MeReq m = {};
            if (mine) {
                CPlane &rp = p.ref[li][ri].lvl[2];
                int16_t  cx = st.me_cx[li][ri], cy = st.me_cy[li][ri];
                const uint32_t dist = mctf ? (uint16_t)ref_distance(p, li, ri) : (uint16_t)scaled_distance(ref_distance(p, li, ri)); // :1299-1302
                int16_t sa_w = (int16_t)imin((int)(c.me_sa.sa_min.width * dist), c.me_sa.sa_max.width);
                int16_t sa_h = (int16_t)imin((int)(c.me_sa.sa_min.height * dist), c.me_sa.sa_max.height);
                if (c.mv_sa_adj_enabled && (!c.mv_sa_adj_nearest_ref_only || ri == 0)) {
                    if (iabs(st.hme_sc_x[li][ri]) > c.mv_sa_adj_mv_size_th) sa_w = (int16_t)(sa_w * c.mv_sa_adj_sa_multiplier);
                    if (iabs(st.hme_sc_y[li][ri]) > c.mv_sa_adj_mv_size_th) sa_h = (int16_t)(sa_h * c.mv_sa_adj_sa_multiplier);
                }
                { const uint32_t q = (uint32_t)(int)sa_w / st.sr_divisor[li][ri]; sa_w = (int16_t)(((q > 1u ? q : 1u) + 7) & ~7u); }
                { const uint32_t q = (uint32_t)(int)sa_h / st.sr_divisor[li][ri]; sa_h = (int16_t)(q > 3u ? q : 3u); }
                const int16_t h0 = sa_h, w0 = sa_w;
                u64 best_hme_sad = ~0ull;
                if (c.me_early_exit_th) {
                    if (st.zz_sad[li][ri] < c.me_early_exit_th / 6) sa_w = sa_h = 1;
                } else {
                    int accurate = 1;
                    const int k2 = st.c00_req[li][ri];
                    if (k2) {
                        const uint32_t zero = (uint32_t)(st.req_key[k2 - 2] >> 32) << 1;
                        const uint32_t hme  = (uint32_t)(st.req_key[k2 - 1] >> 32) << 1;
                        if (zero <= hme) cx = cy = 0; // MIN(zero_cost, hme_cost) == zero_cost
                        best_hme_sad = hme;
                        if (cx == 0 && cy == 0) accurate = 0;
                    }
                    if (c.enable_me_sr_adjustment == 2) {
                        if ((accurate && best_hme_sad < 24 * 24) || (d.is_ref && st.hme_sad64[li][ri] < 24 * 24)) sa_h = (int16_t)(sa_h / 2);
                        if ((li || ri) && BEST_SAD(0, 0)[0] < 5000 && sa_h == h0 && sa_w == w0) { sa_h = (int16_t)(sa_h >> 1); sa_w = (int16_t)(sa_w >> 1); }
                    }
                }
                m.pix0 = plane_at(rp, (int)st.org_x, (int)st.org_y); m.stride = rp.stride;
                m.min_x = (int16_t)(-rp.org_x - (int)st.org_x); m.max_x = (int16_t)(rp.width + rp.org_x - 1 - (int)st.org_x);
                m.min_y = (int16_t)(-rp.org_y - (int)st.org_y); m.max_y = (int16_t)(rp.height + rp.org_y - 1 - (int)st.org_y);
                m.li = (uint8_t)li; m.ri = (uint8_t)ri; m.pad = 0;
                m.sa_w = sa_w; m.sa_h = sa_h; // provisional size, finalised after the probe
                m.ox = cx; m.oy = cy;         // the search centre until then
                m.probe = (c.me_8x8_var_enabled && sa_w * sa_h > 24) ? 1 : 0;
            }
            const bool probe = mine && m.probe;
            const u64  mask = __ballot(mine), pmask = __ballot(probe), below = (1ull << tid) - 1ull;
            if (mine) st.me[__popcll(mask & below)] = m;
            if (probe) { MeReq pr = m; pr.sa_w = pr.sa_h = 1; st.me_probe[__popcll(pmask & below)] = pr; }
            if (tid == 0) { st.nme = __popcll(mask); st.nprobe = __popcll(pmask); }
        };
        auto main_pre = [&]() {
            if (tid < 64) { // wave 0: the 8x8-SAD variance is a wave reduction, lane 0 keeps the result
                const int pic_w = (int16_t)d.aligned_width, pic_h = (int16_t)d.aligned_height;
                for (int i = 0; i < st.nme; i++) {
                    const MeReq m = st.me[i];
                    int16_t sa_w = m.sa_w, sa_h = m.sa_h;
                    const int cx = m.ox, cy = m.oy;
                    if (m.probe) { // :1391-1439 -- only one point was searched: 64x64 SAD == sum of the 8x8 SADs
                        const uint32_t mean = BEST_SAD(m.li, m.ri)[0] / 64;
                        const int32_t  dd   = (int32_t)BEST_SAD(m.li, m.ri)[21 + tid] - (int32_t)mean;
                        const uint32_t var  = wave_sum((uint32_t)(dd * dd)) / 64;
                        if (var > c.me_sr_mult2_th) { sa_w = (int16_t)((imax(1, sa_w * 3 / 2) + 7) & ~7); sa_h = (int16_t)imax(1, sa_h * 3 / 2); }
                        if (var < c.me_sr_div4_th) { sa_w = (int16_t)((imax(1, sa_w >> 2) + 7) & ~7); sa_h = (int16_t)imax(3, imax(1, sa_h >> 2)); }
                        else if (var < c.me_sr_div2_th) { sa_w = (int16_t)((imin(sa_w, sa_w >> 1) + 7) & ~7); sa_h = (int16_t)imax(3, imin(sa_h, sa_h >> 1)); }
                    }
                    int ox = (int16_t)(cx - (sa_w >> 1)), oy = (int16_t)(cy - (sa_h >> 1)), w = sa_w, h = sa_h;
                    clip_axis((int16_t)st.org_x, ox, w, 63, pic_w);
                    w = (w < 8) ? w : (w & ~7);
                    clip_axis((int16_t)st.org_y, oy, h, 63, pic_h);
                    if (tid == 0) { MeReq &o = st.me[i]; o.ox = (int16_t)ox; o.oy = (int16_t)oy; o.sa_w = (int16_t)w; o.sa_h = (int16_t)h; }
                }
            }
        };

        enum { kZz, kPrehme, kL0, kL1, kL2, kC00, kProbe, kMain, kEnd };
#ifdef SVT_ME_ABLATE // diagnostic builds (instruction counting; results are wrong): 1 = no stages, 2 = stop behind HME level 2, 3 = no outputs
        if (SVT_ME_ABLATE == 1) continue;
#endif
        const int n_prehme = c.prehme_l1_early_exit ? nl : 1, n_l0 = l0_dep ? 2 : 1, n_group = me_dep ? 2 : 1;
        if constexpr (MODE == kMeMid1 || MODE == kMeMid2) {
            // The control-only kernels of a staged launch: the one-kernel form's stage loop, written out for the stages they own (same
            // order, same barriers).  A pre-HME / level-0 search the pre-pass did not make defers the block to the whole-pipeline kernel.
            const bool hme_on = c.enable_hme_flag;
            bool defer = false;
            uint32_t flags = 0;
            if constexpr (MODE == kMeMid1) {
                auto pending = [&]() { return __any(tid < st.nreq && !st.req[tid].done); };
                if (c.me_early_exit_th || c.me_safe_limit_zz_th) { zz_pre(); wave_sync(); zz_post(); wave_sync(); }
                if (c.prehme_enable) {
                    for (int b2 = 0; b2 < n_prehme; b2++) {
                        prehme_pre(b2); wave_sync();
                        if (pending()) { defer = true; break; }
                        prehme_post(b2);
                        if (b2 + 1 < n_prehme) wave_sync();
                    }
                    if (!defer) { prehme_final(); wave_sync(); }
                }
                if (!defer && hme_on && c.enable_hme_level0_flag)
                    for (int b2 = 0; b2 < n_l0; b2++) {
                        l0_pre(b2); wave_sync();
                        if (pending()) { defer = true; break; }
                        l0_post(b2); wave_sync();
                    }
                if (!defer) {
                    if (hme_on && c.enable_hme_level1_flag) lvl_pre(1);
                    else if (tid == 0) st.nreq = 0;
                    wave_sync();
                }
            } else {
                flags = __hip_atomic_load(jflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (hme_on && c.enable_hme_level1_flag) { lvl_post(1); wave_sync(); }
                if (hme_on && c.enable_hme_level2_flag) lvl_pre(2);
                else if (tid == 0) st.nreq = 0;
                wave_sync();
            }
            if (!defer && st.nreq && !small_direct_ok(st)) defer = true; // (uniform) the coming level's searches need the general form
            if (defer) {
                if (tid == 0) {
                    *jflag = SVT_HIP_ME_JOB_DEFERRED;
                    hdr.lists[0][atomicAdd(&hdr.queue_head[SVT_HIP_ME_LIST_COUNT(0)], 1u)] = (uint32_t)gjob;
                }
            } else {
                export_state(gjob);
                export_reqs(gjob);
                if (tid == 0) *jflag = flags;
            }
            wave_sync();
            continue;
        }
        if constexpr (MODE == kMeTail) {
            if (c.enable_hme_flag && c.enable_hme_level2_flag) { lvl_post(2); wave_sync(); }
        }
        int step = MODE == kMeTail ? kC00 : kZz, bi = 0; // uniform: derived from launch parameters only
        PROF(2);
        while (step != kEnd) {
            bool run = true;
            CTRL_PRIO(SVT_ME_CTRL_PRIO); // the stretches between the searches are on the block's critical path: ask for issue priority
            switch (step) {
            case kZz: run = c.me_early_exit_th || c.me_safe_limit_zz_th; if (run) zz_pre(); break;
            case kPrehme: run = c.prehme_enable; if (run) prehme_pre(bi); break;
            case kL0: run = c.enable_hme_flag && c.enable_hme_level0_flag; if (run) l0_pre(bi); break;
            case kL1: run = c.enable_hme_flag && c.enable_hme_level1_flag; if (run) lvl_pre(1); break;
            case kL2: run = c.enable_hme_flag && c.enable_hme_level2_flag; if (run) lvl_pre(2); break;
            case kC00: if (bi == 0) centre(); c00_pre(bi); break;
            case kProbe: probe_pre(bi); break;
            default: main_pre(); break;
            }
#ifdef SVT_ME_ABLATE
            if (SVT_ME_ABLATE == 2 && step == kC00) break;
            if (SVT_ME_ABLATE >= 10 && step == SVT_ME_ABLATE - 10) break; // stop in front of stage (SVT_ME_ABLATE - 10): 11 = after zz, 12 = after pre-HME, 13 = after L0, 14 = after L1
            if (SVT_ME_ABLATE == 4 && run && step < kProbe) { step = step == kPrehme ? kL0 : step + 1; bi = 0; continue; } // no HME searches, no post
            if (SVT_ME_ABLATE == 5 && run && step < kProbe) st.nreq = 0;
#endif
            if (run) {
                CTRL_PRIO(0);
                wave_sync();
                PROF(step == kMain ? 20 : 6 + step);
                if (step == kC00 && bi == 0 && st.tf_exit) { step = kEnd; continue; } // uniform: LDS value read after the barrier
                if (step < kProbe) {
                    PROF_STEP(step);
                    if (st.nreq) { // uniform (LDS value read after the barrier)
                        if (!run_small_searches_direct(sh PROF_ARG)) run_searches(sh PROF_ARG);
                    }
                } else {
                    const bool   probe = step == kProbe;
                    const MeReq *list  = probe ? st.me_probe : st.me;
                    const int    count = probe ? st.nprobe : st.nme;
                    run_me_searches(sh, p, list, count, bsad, bmv, r0n PROF_ARG);
                }
                PROF(step == kProbe ? 3 : step == kMain ? 4 : 19);
            }
            CTRL_PRIO(SVT_ME_CTRL_PRIO);
            switch (step) { // fold the results in, pick the next stage
            case kZz: if (run) zz_post(); step = kPrehme; break;
            case kPrehme:
                if (run) prehme_post(bi);
                if (run && bi + 1 < n_prehme) { bi++; break; }
                if (run) prehme_final();
                step = kL0; bi = 0;
                break;
            case kL0:
                if (run) l0_post(bi);
                if (run && bi + 1 < n_l0) { bi++; break; }
                step = kL1; bi = 0;
                break;
            case kL1: if (run) lvl_post(1); step = kL2; break;
            case kL2: if (run) lvl_post(2); step = kC00; break;
            case kC00: step = kProbe; break;
            case kProbe: step = kMain; break;
            default:
                if (bi + 1 < n_group) { bi++; step = kC00; } else step = kEnd;
                break;
            }
            CTRL_PRIO(0);
            wave_sync();
            PROF(5);
        }


#ifdef SVT_ME_ABLATE
        if (SVT_ME_ABLATE == 2 || SVT_ME_ABLATE == 3 || SVT_ME_ABLATE >= 10) continue;
#endif
        // ---- me_prune_ref (motion_estimation.c:1522-1565) ----------------------------------------------------
        if (c.enable_hme_flag && c.enable_me_hme_ref_pruning && !mctf) {
            for (int r = 0; r < 8; r++) { // per reference: sum of its 64 8x8 SADs
                const int li = r >> 2, ri = r & 3;
                if (li >= nl || ri >= d.num_of_ref_pic_to_search[li]) continue;
                const u64 t = st.do_ref[li][ri] ? (u64)wave_sum(BEST_SAD(li, ri)[21 + (tid & 63)]) : (u64)SVT_HIP_MAX_SAD_VALUE * 64;
                if ((tid & 63) == 0) st.hme_sad64[li][ri] = t;
            }
            wave_sync();
            {
                const uint16_t th = c.prune_ref_if_me_sad_dev_bigger_than_th;
                if (th != 0xFFFF) { // lane <-> (list, reference); every value is below 2^32 (64 x MAX_SAD_VALUE, or the initial 0xFFFFFFFF)
                    const int  li = (tid >> 2) & 1, ri = tid & 3;
                    const bool slot = tid < 8;
                    const u64  hs = slot ? st.hme_sad64[li][ri] : ~0ull;
                    const u64  best = (u64)wave_min_u32((uint32_t)(hs < 0xFFFFFFFFull ? hs : 0xFFFFFFFFull));
                    if (slot && ri >= 1 && (hs - best) * 100 > (u64)th * best) st.do_ref[li][ri] = 0;
                }
            }
            wave_sync();
        }

        PROF(13);
        // ---- construct_me_candidate_array* (motion_estimation.c:2532-2836), one thread per PU ----------------
        // (not for ME_MCTF, :3126: the temporal filter consumes p_sb_best_sad / p_sb_best_mv)
        {
          if (!mctf) {
            const uint32_t n_pu = p.n_pu;
            uint8_t  *o_total = p.res.total_me_candidate_index + (size_t)b * n_pu;
            uint32_t *o_mv    = p.res.me_mv_array + (size_t)b * n_pu * d.max_refs;
            uint8_t  *o_cand  = p.res.me_candidate_array + (size_t)b * n_pu * d.max_cand;
            const int r0 = d.num_of_ref_pic_to_search[0], r1 = d.num_of_ref_pic_to_search[1];
            auto use_pu = [&](int nn) { return d.enable_me_16x16 ? (d.enable_me_8x8 || nn < 21) : nn < 5; };
            auto pack   = [](unsigned dir, unsigned i0, unsigned i1, unsigned l0, unsigned l1) {
                return (uint8_t)((dir & 3) | ((i0 & 3) << 2) | ((i1 & 3) << 4) | ((l0 & 1) << 6) | ((l1 & 1) << 7));
            };
            // The reference writes these arrays only partially (malloc'ed): every row starts from zero.  The block's rows
            // are assembled in LDS (the window arena is idle here) and leave as coalesced stores.
            uint32_t *l_mv    = reinterpret_cast<uint32_t *>(LDS(sh.win));
            uint8_t  *l_cand  = LDS(sh.win) + 4 * n_pu * d.max_refs;
            uint8_t  *l_total = l_cand + n_pu * d.max_cand;
            const int stage_dwords = (int)(4 * n_pu * d.max_refs + n_pu * d.max_cand + n_pu + 3) >> 2;
            static_assert(kWinBytes >= 85 * (4 * SVT_HIP_MAX_LISTS * SVT_HIP_MAX_REFS + 32 + 1), "the result rows of one block fit the arena");
            for (int i = tid; i < stage_dwords; i += kThreads) reinterpret_cast<uint32_t *>(LDS(sh.win))[i] = 0;
            wave_sync();
            for (int n = tid; n < d.max_number_of_pus_per_sb; n += kThreads) {
                const int use = use_pu(n);
                const int row = (n > 4) ? c_z_to_raster[n] : n; // == pu below (c_z_to_raster is the identity on 0..4)
                uint8_t first = 0; // candidate 0 of this PU, kept for perform_gm_detection
                auto put_cand = [&](int idx, uint8_t v) { l_cand[row * d.max_cand + idx] = v; if (idx == 0) first = v; };
                uint32_t  nls = nl;
                if (r0 == 1 && r1 == 0) { // construct_me_candidate_array_single_ref
                    const int pu   = c_z_to_raster[n];
                    st.me_dist[pu] = BEST_SAD(0, 0)[n];
                    if (use) l_total[pu] = 1;
                    if (st.do_ref[0][0] && use) { put_cand(0, pack(0, 0, 0, 0, 0)); l_mv[pu * d.max_refs] = BEST_MV(0, 0)[n]; }
                } else if (r0 == 1 && r1 == 1) { // construct_me_candidate_array_mrp_off
                    const int     pu = c_z_to_raster[n];
                    const uint8_t d0 = st.do_ref[0][0], d1 = (nls == 1) ? 0 : st.do_ref[1][0];
                    if (nls < 2 || !st.do_ref[1][0]) nls = 1;
                    const uint32_t prune_th = (d0 && d1) ? (uint32_t)c.prune_me_candidates_th : 0;
                    uint8_t  blk[2] = {d0, d1};
                    uint8_t  off = 0;
                    const uint32_t s0 = BEST_SAD(0, 0)[n], s1 = BEST_SAD(1, 0)[n];
                    const uint32_t best = (d0 && d1) ? (s0 < s1 ? s0 : s1) : (d0 ? s0 : s1);
                    st.me_dist[pu] = best;
                    if (use) l_total[pu] = 1;
                    int min_list = -1;
                    if (c.use_best_unipred_cand_only && blk[0] && blk[1]) min_list = s0 < s1 ? 0 : 1;
                    for (uint32_t li = 0; li < nls && (use || off == 0); li++) {
                        if (!blk[li]) continue;
                        if (prune_th > 0) {
                            const uint32_t dev = (BEST_SAD(li, 0)[n] - best) * 100;
                            if (dev > best * prune_th) { blk[li] = 0; continue; }
                        }
                        if (min_list != -1 && min_list != (int)li) {
                            if (use) l_mv[pu * d.max_refs + (li ? d.max_l0 : 0)] = BEST_MV(li, 0)[n];
                            continue;
                        }
                        if (use) {
                            put_cand(off, pack(li, 0, 0, li == 0 ? li : 24, li == 1 ? li : 24));
                            l_mv[pu * d.max_refs + (li ? d.max_l0 : 0)] = BEST_MV(li, 0)[n];
                        }
                        off++;
                    }
                    if (blk[0] && blk[1] && use) { put_cand(off, pack(2, 0, 0, 0, 1)); l_total[pu] = (uint8_t)(off + 1); }
                } else { // construct_me_candidate_array
                    const int pu = (n > 4) ? c_z_to_raster[n] : n;
                    uint8_t   off = 0;
                    uint8_t   blk[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
                    const uint32_t prune_th = (uint32_t)c.prune_me_candidates_th;
                    uint32_t       best     = ~0u;
                    for (uint32_t li = 0; li < nls; li++)
                        for (int ri = 0; ri < d.num_of_ref_pic_to_search[li]; ri++) {
                            blk[li][ri] = st.do_ref[li][ri];
                            if (blk[li][ri]) best = BEST_SAD(li, ri)[n] < best ? BEST_SAD(li, ri)[n] : best;
                        }
                    st.me_dist[pu] = best;
                    for (uint32_t li = 0; li < nls && (use || off == 0); li++)
                        for (int ri = 0; ri < d.num_of_ref_pic_to_search[li] && (use || off == 0); ri++) {
                            if (!blk[li][ri]) continue;
                            if (prune_th > 0) {
                                const uint32_t dev = (BEST_SAD(li, ri)[n] - best) * 100;
                                if (dev > best * prune_th) { blk[li][ri] = 0; continue; }
                            }
                            if (use) {
                                put_cand(off, pack(li, ri, ri, li == 0 ? li : 24, li == 1 ? li : 24));
                                l_mv[pu * d.max_refs + (li ? d.max_l0 : 0) + ri] = BEST_MV(li, ri)[n];
                            }
                            off++;
                        }
                    if (nls == 2 && use) {
                        for (int a = 0; a < r0; a++)
                            for (int bb = 0; bb < r1; bb++) {
                                if (d.only_l_bwd && (a > 0 || bb > 0)) continue;
                                if (blk[0][a] && blk[1][bb]) put_cand(off++, pack(2, a, bb, 0, 1));
                            }
                        if (!d.only_l_bwd) {
                            for (int a = 1; a < r0; a++)
                                if (blk[0][0] && blk[0][a]) put_cand(off++, pack(2, 0, a, 0, 0));
                            if (r1 == 3 && blk[1][0] && blk[1][2]) put_cand(off++, pack(2, 0, 2, 1, 1));
                        }
                    }
                    if (use) l_total[pu] = off;
                }
                if (row < 88) st.cand0[row] = first;
            }
            wave_sync();
            for (int i = tid; i < (int)(n_pu * d.max_refs); i += kThreads) o_mv[i] = l_mv[i];
            for (int i = tid; i < (int)(n_pu * d.max_cand); i += kThreads) o_cand[i] = l_cand[i];
            for (int i = tid; i < (int)n_pu; i += kThreads) o_total[i] = l_total[i];

            PROF(14);
            // ---- compute_distortion (:2964-3008) + perform_gm_detection (:2838-2961) ---------------------------
            if (tid < 64) { // wave 0: the sums are wave reductions, lane 0 writes
                const uint32_t v8  = st.me_dist[21 + tid];
                const uint32_t d8  = wave_sum(v8);
                const uint32_t d16 = wave_sum(tid < 16 ? st.me_dist[5 + tid] : 0u);
                const uint32_t d32 = wave_sum(tid < 4 ? st.me_dist[1 + tid] : 0u);
                const u64       mean = d8 / 64;
                const long long dd   = (long long)v8 - (long long)mean;
                const u64       ssq  = wave_sum64((u64)(dd * dd));
                // perform_gm_detection (:2838-2961): lane i classifies PU i of the list; the per-(list, ref, component, sign)
                // counts of the reference's cnt[] are popcounts of ballots
                uint8_t stationary = 0, allow_gm = 0;
                if (d.gm_enabled) { // uniform
                    const int low = d.input_resolution <= 2;
                    const int nn  = low ? 64 : 16;
                    int  refk = 0;
                    bool neg_x = false, pos_x = false, neg_y = false, pos_y = false, still_l = false;
                    if (tid < nn) {
                        int idx = (low ? 21 : 5) + tid;
                        if (low && !d.enable_me_8x8) {
                            if (idx >= 21) idx = c_8x8_to_16x16[idx - 21];
                            if (!d.enable_me_16x16 && idx >= 5) idx = c_16x16_to_32x32[idx - 5];
                        } else if (!low && !d.enable_me_16x16 && idx >= 5) idx = c_16x16_to_32x32[idx - 5];
                        const uint8_t  cb  = st.cand0[idx];
                        const unsigned dir = cb & 3;
                        const unsigned li  = (dir == 0 || dir == 2) ? ((cb >> 6) & 1) : ((cb >> 7) & 1);
                        const unsigned ri  = (dir == 0 || dir == 2) ? ((cb >> 2) & 3) : ((cb >> 4) & 3);
                        const u64 a = d.picture_number, bb = d.ref_picture_number[li][ri];
                        int th;
                        if (low) { const int dist = (uint16_t)iabs((int)(int16_t)((a > bb ? a : bb) - (a < bb ? a : bb))); th = d.gm_use_distance_based_active_th ? imax(dist >> 1, 4) : 4; }
                        else     { const int dist = (uint16_t)iabs((int)(int16_t)(a - bb)); th = d.gm_use_distance_based_active_th ? imax(dist * 16, 32) : 32; }
                        const uint32_t mv = BEST_MV(li, ri)[idx];
                        const int mx = (int)(int16_t)(mv & 0xFFFF) << 2, my = (int)(int16_t)(mv >> 16) << 2;
                        refk  = (int)(li * 4 + ri);
                        neg_x = mx < -th; pos_x = !neg_x && mx > th;
                        neg_y = my < -th; pos_y = !neg_y && my > th;
                        const int sth = low ? 0 : 4;
                        still_l = iabs(mx) <= sth && iabs(my) <= sth;
                    }
                    const uint32_t tot = (uint32_t)nn;
                    if ((uint32_t)__popcll(__ballot(still_l)) > (tot * 5) / 100) stationary = 1;
                    const u64 bnx = __ballot(neg_x), bpx = __ballot(pos_x), bny = __ballot(neg_y), bpy = __ballot(pos_y);
                    for (int k = 0; k < 8; k++) { // every (list, ref) a candidate can name
                        const u64 m = __ballot(tid < nn && refk == k);
                        if ((uint32_t)__popcll(m & bnx) > tot / 2 || (uint32_t)__popcll(m & bpx) > tot / 2 || (uint32_t)__popcll(m & bny) > tot / 2 ||
                            (uint32_t)__popcll(m & bpy) > tot / 2)
                            allow_gm = 1;
                    }
                }
              if (tid == 0) {
                const uint32_t pix = st.b64_w * st.b64_h;
                p.res.me_8x8_cost_variance[b] = (uint32_t)(ssq / 64);
                p.res.rc_me_distortion[b]     = d.input_resolution <= 2 ? d8 : d16;
                p.res.me_64x64_distortion[b]  = (st.me_dist[0] * 4096u) / pix;
                p.res.me_32x32_distortion[b]  = (d32 * 4096u) / pix;
                p.res.me_16x16_distortion[b]  = (d16 * 4096u) / pix;
                p.res.me_8x8_distortion[b]    = (d8 * 4096u) / pix;
                p.res.stationary_block_present_sb[b] = stationary;
                p.res.rc_me_allow_gm[b]              = allow_gm;
              }
            }
          }
            PROF(15);
            // ---- optional search-level results ------------------------------------------------------------------
            if (p.res.sb_best_sad || p.res.sb_best_mv) // the slots of the (list, reference) pairs the picture searches; the others are left alone
                for (int i = tid; i < n_rows * 85; i += kThreads) {
                    const int row = i / 85, nn = i - row * 85, li = row < r0n ? 0 : 1, ri = row < r0n ? row : row - r0n;
                    const bool   ok = st.do_ref[li][ri] != 0;
                    const size_t o  = (size_t)b * 680 + (size_t)(li * 4 + ri) * 85 + nn;
                    if (p.res.sb_best_sad) p.res.sb_best_sad[o] = ok ? bsad[i] : SVT_HIP_MAX_SAD_VALUE;
                    if (p.res.sb_best_mv) p.res.sb_best_mv[o] = ok ? bmv[i] : 0;
                }
            if (tid < 8) {
                const int li = tid >> 2, ri = tid & 3;
                const bool live = li < nl && ri < d.num_of_ref_pic_to_search[li];
                if (p.res.do_ref) p.res.do_ref[(size_t)b * 8 + tid] = live ? st.do_ref[li][ri] : 0;
                if (p.res.hme_sad) p.res.hme_sad[(size_t)b * 8 + tid] = live ? (uint32_t)st.hme_sad64[li][ri] : 0;
                if (p.res.hme_sc) {
                    p.res.hme_sc[((size_t)b * 8 + tid) * 2]     = live ? st.hme_sc_x[li][ri] : 0;
                    p.res.hme_sc[((size_t)b * 8 + tid) * 2 + 1] = live ? st.hme_sc_y[li][ri] : 0;
                }
            }
        }
        wave_sync();
        PROF(16);
    }
#ifdef SVT_HIP_ME_PROFILE
#ifndef SVT_HIP_ME_PROFILE_MODE
#define SVT_HIP_ME_PROFILE_MODE 0 /* which kernel of the chain adds its phase sums (one table per diagnostic build) */
#endif
    if (MODE == SVT_HIP_ME_PROFILE_MODE) PROF_FLUSH(hdr.queue_head + 16);
#endif
    if ((MODE == kMeFull || MODE == kMeMid1) && has_dense && hdr.count_dense) { // counters of the context's diagnostics entry (svt_hip_me_dense_counters): uniform
        const uint32_t h = wave_sum_u32(n_hit), m = wave_sum_u32(n_miss);
        if (tid == 0) {
            atomicAdd(reinterpret_cast<unsigned long long *>(hdr.queue_head + SVT_HIP_ME_COUNTER_WORD), (unsigned long long)h);
            atomicAdd(reinterpret_cast<unsigned long long *>(hdr.queue_head + SVT_HIP_ME_COUNTER_WORD) + 1, (unsigned long long)m);
        }
    }
}

#undef BEST_SAD
#undef BEST_MV

#define SVT_ME_KERNEL(NAME, MODE, WAVES)                                                                                                              \
    extern "C" __global__ void __launch_bounds__(64, WAVES) NAME(const MeBatchHeader *__restrict__ ghdr, const MeKernelParams *__restrict__ gparams,   \
                                                                 const uint32_t launch_flags) {                                                       \
        me_b64_body<MODE>(ghdr, gparams, launch_flags);                                                                                               \
    }
SVT_ME_KERNEL(svt_hip_me_b64_kernel, kMeFull, SVT_HIP_ME_WAVES_PER_SIMD)
SVT_ME_KERNEL(svt_hip_me_mid1_kernel, kMeMid1, SVT_HIP_ME_MID_WAVES_PER_SIMD)
SVT_ME_KERNEL(svt_hip_me_s1_kernel, kMeS1, SVT_HIP_ME_SEARCH_WAVES_PER_SIMD)
SVT_ME_KERNEL(svt_hip_me_mid2_kernel, kMeMid2, SVT_HIP_ME_MID_WAVES_PER_SIMD)
SVT_ME_KERNEL(svt_hip_me_s2_kernel, kMeS2, SVT_HIP_ME_SEARCH_WAVES_PER_SIMD)
SVT_ME_KERNEL(svt_hip_me_tail_kernel, kMeTail, SVT_HIP_ME_TAIL_WAVES_PER_SIMD)
#undef SVT_ME_KERNEL

#include "me_dense.inl"

size_t svt_hip_me_kernel_lds_bytes(void) { return lds_layout(SVT_HIP_MAX_LISTS * SVT_HIP_MAX_REFS, 0, 1).total + SVT_HIP_ME_PROFILE_LDS; }

#include "svt_hip_internal.h"

// Host launcher: zero the lane's band queues, copy the header + parameter blocks to HBM through one block of the lane's
// pinned ring (stream ordered: the previous launch on this lane has consumed the device block before the copy lands; the ring
// lets SVT_HIP_PARAM_RING launches be enqueued ahead before the host has to wait) and enqueue the persistent waves on the
// lane's stream.  The caller holds the lane (lane 0: ctx->async_mu).
int svt_hip_me_launch(SvtHipContext *ctx, SvtHipLane *lane, const MeKernelParams *params, const uint32_t *n_jobs, uint32_t n_pictures) {
    if (n_pictures == 0 || n_pictures > SVT_HIP_ME_MAX_PICTURES) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "%u pictures in one ME launch (1..%d)", n_pictures, SVT_HIP_ME_MAX_PICTURES);
    MeBatchHeader hdr;
    memset(&hdr, 0, sizeof(hdr));
    hdr.n_pictures = n_pictures;
    int all_sub = 1;
    for (uint32_t i = 0; i < n_pictures; i++) {
        hdr.job_base[i + 1] = hdr.job_base[i] + n_jobs[i];
        const uint32_t rows = (uint32_t)params[i].desc.num_of_ref_pic_to_search[0] + (params[i].desc.num_of_list_to_search > 1 ? (uint32_t)params[i].desc.num_of_ref_pic_to_search[1] : 0u);
        hdr.n_slot = rows > hdr.n_slot ? rows : hdr.n_slot; // rows of best_sad / best_mv a wave keeps in LDS
        if (params[i].cfg.hme_search_method != 0 || params[i].cfg.me_search_method != 0) all_sub = 0;
    }
    hdr.cshift = (uint32_t)all_sub; // every search of the launch reads even source rows only: the source views keep just those
    for (uint32_t i = n_pictures; i < SVT_HIP_ME_MAX_PICTURES; i++) hdr.job_base[i + 1] = hdr.job_base[n_pictures];
    const uint32_t total = hdr.job_base[n_pictures];
    if (total == 0) return SVT_HIP_OK;
    // eight contiguous ranges of the job space, one queue each: neighbouring blocks share reference windows -> same XCD L2.
    for (int q = 0; q <= SVT_HIP_ME_QUEUES; q++) hdr.queue_begin[q] = (uint32_t)(((uint64_t)total * q) / SVT_HIP_ME_QUEUES);
    hdr.queue_head = lane->queue_head;
    SVT_HIP_CHECK(ctx, hipMemsetAsync(lane->queue_head, 0, SVT_HIP_ME_QUEUES * sizeof(uint32_t), lane->stream));
    // the dense pre-pass: entries, the slot buffer (one slot per job, searched reference and kind; 0xFF = "not filled")
    static thread_local MeDenseEntry entries[SVT_HIP_ME_DENSE_MAX_ENTRIES];
    uint32_t n_entries = 0, n_units = 0;
    if (ctx->me_dense) n_entries = dense_plan(params, n_jobs, n_pictures, entries, &n_units, (uint32_t)ctx->num_cus * 4u * SVT_HIP_ME_DENSE_WAVES);
    auto grow = [&](void **buf, size_t *have, size_t need) -> int { // a lane's device buffers grow on demand (earlier launches of the lane may still read the old one)
        if (need <= *have) return SVT_HIP_OK;
        if (*buf) {
            SVT_HIP_CHECK(ctx, hipStreamSynchronize(lane->stream));
            (void)hipFree(*buf);
            *buf = nullptr; *have = 0;
        }
        if (hipMalloc(buf, need) != hipSuccess) return svt_hip_fail(ctx, SVT_HIP_ERR_NO_MEMORY, "hipMalloc(%zu) failed (ME launch buffers)", need);
        *have = need;
        return SVT_HIP_OK;
    };
    if (n_units) {
        const size_t need = (size_t)total * hdr.n_slot * SVT_HIP_ME_DENSE_KINDS * sizeof(MeDenseSlot);
        if (int rc = grow(&lane->dense, &lane->dense_bytes, need)) return rc;
        SVT_HIP_CHECK(ctx, hipMemsetAsync(lane->dense, 0xFF, need, lane->stream));
        hdr.dense = static_cast<MeDenseSlot *>(lane->dense);
        hdr.n_dense_entries = n_entries; hdr.n_dense_units = n_units;
        hdr.count_dense = ctx->me_counting ? 1u : 0u;
    }
    // With a pre-pass the per-block pipeline runs STAGED: small kernels cut at its searches, a block's state travelling through HBM between them
    // (a launch of few blocks is bound by latency, not by throughput: the chain of nine short kernels costs it more than it gains)
    const bool staged = n_units && (ctx->me_staged == 2 || (ctx->me_staged == 1 && total >= kStagedMinJobs));
    if (staged) {
        const size_t stage_bytes = (size_t)total * SVT_HIP_ME_STAGE_BYTES, words = (size_t)total * sizeof(uint32_t);
        if (int rc = grow(&lane->stage, &lane->stage_bytes, stage_bytes + 2 * words)) return rc;
        uint8_t *base = static_cast<uint8_t *>(lane->stage);
        hdr.stage     = base;
        hdr.job_flags = reinterpret_cast<uint32_t *>(base + stage_bytes);
        hdr.lists[0] = reinterpret_cast<uint32_t *>(base + stage_bytes + words); // the deferred blocks
        SVT_HIP_CHECK(ctx, hipMemsetAsync(lane->queue_head + 128, 0, SVT_HIP_ME_QUEUE_BLOCK_BYTES - 128 * sizeof(uint32_t), lane->stream)); // the staged kernels' counters and the lists
    }
    if (!ctx->me_attr_set) { // per context = per device; racing first calls set the same value
        const void *kernels[] = {reinterpret_cast<const void *>(svt_hip_me_b64_kernel),  reinterpret_cast<const void *>(svt_hip_me_mid1_kernel), reinterpret_cast<const void *>(svt_hip_me_s1_kernel),
                                 reinterpret_cast<const void *>(svt_hip_me_mid2_kernel), reinterpret_cast<const void *>(svt_hip_me_s2_kernel),   reinterpret_cast<const void *>(svt_hip_me_tail_kernel)};
        for (const void *k : kernels) SVT_HIP_CHECK(ctx, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)svt_hip_me_kernel_lds_bytes()));
        ctx->me_attr_set = true;
    }
    const int slot = lane->ring_next;
    lane->ring_next = (slot + 1) % SVT_HIP_PARAM_RING;
    SVT_HIP_CHECK(ctx, hipEventSynchronize(lane->params_copied[slot])); // the copy that last read this pinned block has run
    uint8_t *host = lane->params_host[slot], *dev = lane->params_dev;
    memcpy(host, &hdr, sizeof(hdr));
    memcpy(host + SVT_HIP_ME_HEADER_BYTES, params, sizeof(MeKernelParams) * n_pictures);
    size_t block_bytes = SVT_HIP_ME_HEADER_BYTES + sizeof(MeKernelParams) * n_pictures;
    if (n_units) { // the entry table travels in the same block, behind the parameter blocks' full extent
        memcpy(host + SVT_HIP_ME_ENTRIES_OFFSET, entries, sizeof(MeDenseEntry) * n_entries);
        block_bytes = SVT_HIP_ME_ENTRIES_OFFSET + sizeof(MeDenseEntry) * n_entries;
    }
    SVT_HIP_CHECK(ctx, hipMemcpyAsync(dev, host, block_bytes, hipMemcpyHostToDevice, lane->stream));
    SVT_HIP_CHECK(ctx, hipEventRecord(lane->params_copied[slot], lane->stream));
    const MeBatchHeader  *d_hdr = reinterpret_cast<const MeBatchHeader *>(dev);
    const MeKernelParams *d_par = reinterpret_cast<const MeKernelParams *>(dev + SVT_HIP_ME_HEADER_BYTES);
    // svt_hip_context_set_me_timing: an event in front of every kernel of the chain (me_mark[kernel]) and one behind the last (me_mark[9])
    const bool timing = ctx->me_timing;
    lane->me_marked = 0;
    auto mark = [&](int kernel) { // kernel < 0: the closing event
        if (!timing) return;
        const int i = kernel < 0 ? SVT_HIP_ME_CHAIN_KERNELS : kernel;
        if (!lane->me_mark[i]) hipEventCreate(&lane->me_mark[i]);
        hipEventRecord(lane->me_mark[i], lane->stream);
        if (kernel >= 0) lane->me_marked |= 1u << kernel;
    };
    if (n_units) {
        mark(0);
        hipLaunchKernelGGL(svt_hip_me_dense_kernel, dim3(n_units), dim3(64), 0, lane->stream, d_hdr, d_par, reinterpret_cast<const MeDenseEntry *>(dev + SVT_HIP_ME_ENTRIES_OFFSET));
        SVT_HIP_CHECK(ctx, hipGetLastError());
    }
    // persistent waves: as many per CU as the kernel's LDS slice and the register budget of its launch bounds keep resident
    auto launch = [&](void (*kernel)(const MeBatchHeader *, const MeKernelParams *, uint32_t), int mode, uint32_t waves_per_simd, uint32_t flags) {
        const size_t lds = ((size_t)lds_layout((int)hdr.n_slot, (int)hdr.cshift, hdr.dense ? 1 : 0, mode).total + SVT_HIP_ME_PROFILE_LDS + 127) & ~(size_t)127;
        uint32_t per_cu = (uint32_t)((160u * 1024u) / lds);
        if (per_cu > 4u * waves_per_simd) per_cu = 4u * waves_per_simd;
        if (ctx->me_waves_per_cu && per_cu > ctx->me_waves_per_cu) per_cu = ctx->me_waves_per_cu; // the caller leaves room for kernels of another stream
        uint32_t grid = (uint32_t)ctx->num_cus * per_cu;
        if ((flags & 1u) && grid > (uint32_t)ctx->num_cus * kListWavesPerCu) grid = (uint32_t)ctx->num_cus * kListWavesPerCu; // list kernels: their lists are short (edge blocks)
        if (grid > total) grid = total;
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), lds, lane->stream, d_hdr, d_par, flags);
    };
    if (staged) {
        mark(1); launch(svt_hip_me_mid1_kernel, kMeMid1, SVT_HIP_ME_MID_WAVES_PER_SIMD, 0u);
        mark(2); launch(svt_hip_me_s1_kernel, kMeS1, SVT_HIP_ME_SEARCH_WAVES_PER_SIMD, 0u);
        mark(3); launch(svt_hip_me_mid2_kernel, kMeMid2, SVT_HIP_ME_MID_WAVES_PER_SIMD, 0u);
        mark(4); launch(svt_hip_me_s2_kernel, kMeS2, SVT_HIP_ME_SEARCH_WAVES_PER_SIMD, 0u);
        mark(5); launch(svt_hip_me_tail_kernel, kMeTail, SVT_HIP_ME_TAIL_WAVES_PER_SIMD, 0u);
        mark(6); launch(svt_hip_me_b64_kernel, kMeFull, SVT_HIP_ME_WAVES_PER_SIMD, 1u); // the deferred blocks (usually none: the waves find an empty list and leave)
    } else {
        mark(6); launch(svt_hip_me_b64_kernel, kMeFull, SVT_HIP_ME_WAVES_PER_SIMD, 0u);
    }
    mark(-1);
    SVT_HIP_CHECK(ctx, hipGetLastError());
    return SVT_HIP_OK;
}

static const char *const kChainKernelNames[SVT_HIP_ME_CHAIN_KERNELS] = {"svt_hip_me_dense_kernel", "svt_hip_me_mid1_kernel", "svt_hip_me_s1_kernel",  "svt_hip_me_mid2_kernel",
                                                                        "svt_hip_me_s2_kernel",    "svt_hip_me_tail_kernel", "svt_hip_me_b64_kernel"};
extern "C" const char *svt_hip_me_chain_kernel_name(int i) { return i >= 0 && i < SVT_HIP_ME_CHAIN_KERNELS ? kChainKernelNames[i] : nullptr; }

extern "C" int svt_hip_context_set_me_counting(SvtHipContext *ctx, int on) {
    if (!ctx) return SVT_HIP_ERR_BAD_PARAM;
    ctx->me_counting = on != 0;
    return SVT_HIP_OK;
}

extern "C" int svt_hip_context_set_me_timing(SvtHipContext *ctx, int on) {
    if (!ctx) return SVT_HIP_ERR_BAD_PARAM;
    ctx->me_timing = on != 0;
    return SVT_HIP_OK;
}

// the kernels of the last launch enqueued through lane 0 (the context stream)
extern "C" int svt_hip_me_launch_times(SvtHipContext *ctx, float ms[SVT_HIP_ME_CHAIN_KERNELS]) {
    if (!ctx || !ms) return SVT_HIP_ERR_BAD_PARAM;
    SvtHipLane &l = ctx->lane[0];
    hipSetDevice(ctx->device);
    SVT_HIP_CHECK(ctx, hipStreamSynchronize(l.stream));
    for (int i = 0; i < SVT_HIP_ME_CHAIN_KERNELS; i++) {
        ms[i] = 0.f;
        if (!(l.me_marked >> i & 1u)) continue;
        // the event behind kernel i is the next one recorded: the one in front of the next kernel that ran, or the closing event
        int j = i + 1;
        while (j < SVT_HIP_ME_CHAIN_KERNELS && !(l.me_marked >> j & 1u)) j++;
        SVT_HIP_CHECK(ctx, hipEventElapsedTime(&ms[i], l.me_mark[i], l.me_mark[j]));
    }
    return SVT_HIP_OK;
}

extern "C" int svt_hip_context_set_me_staged(SvtHipContext *ctx, int on) {
    if (!ctx) return SVT_HIP_ERR_BAD_PARAM;
    if (on < 0 || on > 2) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "svt_hip_context_set_me_staged: %d is not 0 (off), 1 (large launches) or 2 (every launch)", on);
    ctx->me_staged = on;
    return SVT_HIP_OK;
}

extern "C" int svt_hip_context_set_me_dense(SvtHipContext *ctx, int on) {
    if (!ctx) return SVT_HIP_ERR_BAD_PARAM;
    ctx->me_dense = on != 0;
    return SVT_HIP_OK;
}

// Diagnostics: searches the per-block kernel took from the dense pre-pass / pushed searches it found no slot for, summed over the lanes
// since the last call (the call waits for the lanes' streams).
extern "C" int svt_hip_me_dense_counters(SvtHipContext *ctx, unsigned long long out[2]) {
    if (!ctx || !out) return SVT_HIP_ERR_BAD_PARAM;
    out[0] = out[1] = 0;
    hipSetDevice(ctx->device);
    for (int i = 0; i < SVT_HIP_LANES; i++) {
        SvtHipLane &l = ctx->lane[i];
        if (!l.ready) continue;
        unsigned long long v[2];
        SVT_HIP_CHECK(ctx, hipStreamSynchronize(l.stream));
        SVT_HIP_CHECK(ctx, hipMemcpy(v, l.queue_head + SVT_HIP_ME_COUNTER_WORD, sizeof(v), hipMemcpyDeviceToHost));
        SVT_HIP_CHECK(ctx, hipMemset(l.queue_head + SVT_HIP_ME_COUNTER_WORD, 0, sizeof(v)));
        out[0] += v[0]; out[1] += v[1];
    }
    return SVT_HIP_OK;
}

extern "C" int svt_hip_context_set_me_waves_per_cu(SvtHipContext *ctx, uint32_t waves) {
    if (!ctx) return SVT_HIP_ERR_BAD_PARAM;
    ctx->me_waves_per_cu = waves;
    return SVT_HIP_OK;
}
