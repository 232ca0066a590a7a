// hadamard_mfma.hip -- VERDICT r01 item 10: is the 8x8 Hadamard core (svt_aom_hadamard_8x8 + svt_aom_satd, the inner step of hadamard_path_c and of
// the 16x16 / 32x32 Hadamards) worth moving to the matrix cores?  Two kernels over the same int16 residual blocks (9-bit values, 128 B per block,
// contiguous), each writing the block's SATD:
//   valu : 8 lanes per block, a lane owns one row: column butterflies across lanes by DPP-style shuffles, row butterflies in registers
//   mfma : 4 blocks = one 16x16 tile; Y = H16 * (X * H16) with H16 = diag(H8, H8) as two v_mfma_f32_16x16x16_f16 -- exact: 9-bit inputs and
//          +-1 weights are exact f16 values, partial sums stay below 2^15 in f32, and the first product's C layout is the second's B layout.
// Both are checked against a host Hadamard.  "stream": blocks come from HBM once (128 B per block); "resident": 8 MB of blocks 64 times, every wave
// re-reading its own kilobyte (cache hits) -- the rate the arithmetic alone allows.
// Build: hipcc --offload-arch=gfx950 -O3 hadamard_mfma.hip -o hadamard_mfma ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float    float4v __attribute__((ext_vector_type(4)));
typedef short    short4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xor_lane(int v, int m) { return __shfl_xor(v, m, 64); }

// ---- VALU: lane = (block, row) ----
__global__ void __launch_bounds__(256) had8_valu(const int16_t *res, uint32_t *satd, uint32_t n_blocks, int repeat) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int rep = 0; rep < repeat; rep++)
        for (uint32_t b0 = wave * 8; b0 < n_blocks; b0 += n_waves * 8) {
            const uint32_t blk = b0 + (lane >> 3), row = lane & 7;
            const uint4    raw = *reinterpret_cast<const uint4 *>(res + (size_t)blk * 64 + row * 8);
            int v[8] = {(int16_t)(raw.x & 0xFFFF), (int16_t)(raw.x >> 16), (int16_t)(raw.y & 0xFFFF), (int16_t)(raw.y >> 16),
                        (int16_t)(raw.z & 0xFFFF), (int16_t)(raw.z >> 16), (int16_t)(raw.w & 0xFFFF), (int16_t)(raw.w >> 16)};
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) { // columns: butterflies between rows (lanes)
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int p = xor_lane(v[k], m);
                    v[k] = (row & m) ? p - v[k] : v[k] + p;
                }
            }
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) { // rows: in registers
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (!(k & m)) { const int a = v[k], b = v[k | m]; v[k] = a + b; v[k | m] = a - b; }
            }
            uint32_t s = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) s += (uint32_t)(v[k] < 0 ? -v[k] : v[k]);
            s += xor_lane(s, 1); s += xor_lane(s, 2); s += xor_lane(s, 4);
            if (row == 0) satd[blk] = s;
        }
}

// ---- MFMA: 4 blocks per 16x16 tile ----
__global__ void __launch_bounds__(256) had8_mfma(const int16_t *res, uint32_t *satd, uint32_t n_blocks, int repeat) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t r = lane & 15, k0 = 4 * (lane >> 4);
    half4 h; // H16[r][k0 + j] = H16[k0 + j][r] (symmetric): the A operand of the second product and the B operand of the first
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t k = k0 + j;
        h[j] = ((k >> 3) != (r >> 3)) ? (_Float16)0 : ((__builtin_popcount((k & 7) & (r & 7)) & 1) ? (_Float16)-1 : (_Float16)1);
    }
    // this lane's 4 samples of tile row r, columns k0 .. k0 + 3: block (r / 8, k0 / 8), its row r % 8, columns k0 % 8 ..
    const uint32_t in_off = ((r >> 3) * 2 + (k0 >> 3)) * 64 + (r & 7) * 8 + (k0 & 7);
    for (int rep = 0; rep < repeat; rep++)
        for (uint32_t b0 = wave * 4; b0 < n_blocks; b0 += n_waves * 4) {
            const short4v x = *reinterpret_cast<const short4v *>(res + (size_t)b0 * 64 + in_off);
            half4 a;
#pragma unroll
            for (int j = 0; j < 4; j++) a[j] = (_Float16)x[j];
            float4v t = {0, 0, 0, 0};
            t = __builtin_amdgcn_mfma_f32_16x16x16f16(a, h, t, 0, 0, 0); // T = X * H16: lane holds T[k0 + j][r]
            half4 tb;
#pragma unroll
            for (int j = 0; j < 4; j++) tb[j] = (_Float16)t[j]; // |T| <= 8 * 255: exact in f16
            float4v y = {0, 0, 0, 0};
            y = __builtin_amdgcn_mfma_f32_16x16x16f16(h, tb, y, 0, 0, 0); // Y = H16 * T: lane holds Y[k0 + j][r]
            uint32_t s = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) s += (uint32_t)__builtin_fabsf(y[j]);
            // the block of Y[k0 + j][r] is (k0 / 8, r / 8): sum over the 8 lanes of r % 8 and the two lane groups sharing k0 / 8
            s += xor_lane(s, 1); s += xor_lane(s, 2); s += xor_lane(s, 4); s += xor_lane(s, 16);
            if ((lane & 23) == 0) satd[b0 + (k0 >> 3) * 2 + (r >> 3)] = s;
        }
}

static uint32_t host_satd(const int16_t *b) {
    int v[64];
    for (int i = 0; i < 64; i++) v[i] = b[i];
    for (int pass = 0; pass < 2; pass++)
        for (int line = 0; line < 8; line++)
            for (int m = 1; m < 8; m <<= 1)
                for (int k = 0; k < 8; k++)
                    if (!(k & m)) {
                        int *p = pass ? &v[line * 8] : &v[line];
                        const int st = pass ? 1 : 8, a0 = p[k * st], a1 = p[(k | m) * st];
                        p[k * st] = a0 + a1; p[(k | m) * st] = a0 - a1;
                    }
    uint32_t s = 0;
    for (int i = 0; i < 64; i++) s += (uint32_t)abs(v[i]);
    return s;
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const uint32_t n_stream = 2u << 20, n_res = 64u << 10; // 256 MB / 8 MB of blocks (a wave re-reads its own 1 KB: cache hits)
    std::vector<int16_t> host((size_t)n_stream * 64);
    uint32_t seed = 12345;
    for (auto &x : host) { seed = seed * 1664525u + 1013904223u; x = (int16_t)((int)((seed >> 9) % 511) - 255); }
    for (int i = 0; i < 64; i++) host[i] = (i & 1) ? -255 : 255; // an extreme block
    int16_t *d_res; uint32_t *d_out;
    (void)hipMalloc(&d_res, host.size() * 2);
    (void)hipMalloc(&d_out, (size_t)n_stream * 4);
    (void)hipMemcpy(d_res, host.data(), host.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<uint32_t> out(n_stream), want(4096);
    for (uint32_t b = 0; b < 4096; b++) want[b] = host_satd(&host[(size_t)b * 64]);
    struct { const char *name; void (*k)(const int16_t *, uint32_t *, uint32_t, int); } ks[2] = {{"valu", had8_valu}, {"mfma", had8_mfma}};
    for (auto &kk : ks) {
        (void)hipMemset(d_out, 0xff, (size_t)n_stream * 4);
        hipLaunchKernelGGL(kk.k, dim3(cus * 8), dim3(256), 0, 0, d_res, d_out, n_stream, 1);
        (void)hipMemcpy(out.data(), d_out, (size_t)n_stream * 4, hipMemcpyDeviceToHost);
        uint32_t bad = 0;
        for (uint32_t b = 0; b < 4096; b++) bad += out[b] != want[b];
        for (uint32_t b = n_stream - 64; b < n_stream; b++) bad += out[b] != host_satd(&host[(size_t)b * 64]);
        printf("%s: %u mismatches against the host Hadamard (4160 blocks checked)\n", kk.name, bad);
        for (int mode = 0; mode < 2; mode++) {
            const uint32_t n = mode ? n_res : n_stream;
            const int repeat = mode ? 64 : 1;
            float ms = 0, best = 1e9f;
            for (int it = 0; it < 5; it++) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(kk.k, dim3(cus * 8), dim3(256), 0, 0, d_res, d_out, n, repeat);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double blocks = (double)n * repeat;
            printf("  %-8s %s: %.3f ms for %.1f M blocks = %.1f G blocks/s = %.2f TB/s of residual bytes\n", kk.name, mode ? "resident" : "stream", best, blocks / 1e6,
                   blocks / best / 1e6, blocks * 128 / best / 1e9);
        }
    }
    return 0;
}
