// qsad_peak.hip -- measures the chip's issue rate of v_qsad_pk_u16_u8 (the packed SAD instruction the ME / DG kernels
// are built on): the "packed-SAD VALU peak" of SURVEY §8d.  One |a-b| evaluation = one byte pair; one lane-instruction = 16.
// Build: hipcc --offload-arch=gfx950 -O3 qsad_peak.hip -o qsad_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long u64;

template <int NACC>
__global__ void __launch_bounds__(256) qsad_loop(u64 *out, u64 seed, int iters) {
    u64 acc[NACC];
    u64 w = seed + threadIdx.x * 0x9E3779B97F4A7C15ull;
    uint32_t s = (uint32_t)(seed >> 7) ^ threadIdx.x;
#pragma unroll
    for (int k = 0; k < NACC; k++) acc[k] = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < NACC; k++) acc[k] = __builtin_amdgcn_qsad_pk_u16_u8(w + k, s, acc[k]);
        asm volatile("" : "+v"(w), "+v"(s));
    }
    u64 r = 0;
#pragma unroll
    for (int k = 0; k < NACC; k++) r ^= acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// same instruction count of v_sad_u8 (4 |a-b| per lane-instruction) and v_msad_u8 for comparison
__global__ void __launch_bounds__(256) sad_u8_loop(uint32_t *out, uint32_t seed, int iters) {
    uint32_t acc[8], w = seed + threadIdx.x, s = seed ^ 0x55aa11u;
    for (int k = 0; k < 8; k++) acc[k] = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = __builtin_amdgcn_sad_u8(w + k, s, acc[k]);
        asm volatile("" : "+v"(w), "+v"(s));
    }
    uint32_t r = 0;
    for (int k = 0; k < 8; k++) r ^= acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    u64 *out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
        const int grid = cus * wg_per_cu;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            qsad_loop<8><<<grid, 256>>>(out, 1234567ull, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double lane_instr = (double)grid * 256 * iters * 8;
        printf("qsad_pk_u16_u8  %d waves/SIMD: %.3f ms  %.2f T lane-instr/s  %.1f T|a-b|/s  (%.2f cycles per wave-instr per SIMD at 2.4 GHz)\n", wg_per_cu, ms,
               lane_instr / ms / 1e9, lane_instr * 16 / ms / 1e9, (double)ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wg_per_cu));
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            sad_u8_loop<<<grid, 256>>>((uint32_t *)out, 12345u, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        hipEventElapsedTime(&ms, e0, e1);
        printf("sad_u8          %d waves/SIMD: %.3f ms  %.2f T lane-instr/s  %.1f T|a-b|/s  (%.2f cycles per wave-instr per SIMD at 2.4 GHz)\n", wg_per_cu, ms,
               lane_instr / ms / 1e9, lane_instr * 4 / ms / 1e9, (double)ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wg_per_cu));
    }
    return 0;
}
