// qsad_peak.hip -- measures the chip's issue rate of v_qsad_pk_u16_u8 (the packed SAD instruction the ME / DG kernels
// are built on): the "packed-SAD VALU peak" of SURVEY §8d.  One |a-b| evaluation = one byte pair; one lane-instruction = 16.
// The loop body is nothing but qsads on independent accumulators (operands stay in registers).
// Build: hipcc --offload-arch=gfx950 -O3 qsad_peak.hip -o qsad_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long u64;

__global__ void __launch_bounds__(256) qsad_loop(u64 *out, u64 seed, int iters) {
    u64 acc[8], w[8];
    uint32_t s = (uint32_t)(seed >> 7) ^ threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; k++) { acc[k] = 0; w[k] = seed * (k + 3) + threadIdx.x * 0x9E3779B97F4A7C15ull; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = __builtin_amdgcn_qsad_pk_u16_u8(w[k], s, acc[k]);
        asm volatile("" : "+v"(s));
    }
    u64 r = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) r ^= acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ void __launch_bounds__(256) sad_u8_loop(uint32_t *out, uint32_t seed, int iters) {
    uint32_t acc[8], w[8], s = seed ^ 0x55aa11u;
    for (int k = 0; k < 8; k++) { acc[k] = 0; w[k] = seed * (k + 5) + threadIdx.x; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = __builtin_amdgcn_sad_u8(w[k], s, acc[k]);
        asm volatile("" : "+v"(s));
    }
    uint32_t r = 0;
    for (int k = 0; k < 8; k++) r ^= acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    u64 *out;
    (void)hipMalloc(&out, (size_t)cus * 8 * 256 * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
        const int grid = cus * wg_per_cu;
        float ms;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            qsad_loop<<<grid, 256>>>(out, 1234567ull, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
        }
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double lane_instr = (double)grid * 256 * iters * 8;
        printf("v_qsad_pk_u16_u8  %d waves/SIMD: %.3f ms  %.2f T lane-instr/s  %.1f T|a-b|/s  (%.2f cycles per wave-instr per SIMD at 2.4 GHz)\n", wg_per_cu, ms,
               lane_instr / ms / 1e9, lane_instr * 16 / ms / 1e9, (double)ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wg_per_cu));
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            sad_u8_loop<<<grid, 256>>>((uint32_t *)out, 12345u, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
        }
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("v_sad_u8          %d waves/SIMD: %.3f ms  %.2f T lane-instr/s  %.1f T|a-b|/s  (%.2f cycles per wave-instr per SIMD at 2.4 GHz)\n", wg_per_cu, ms,
               lane_instr / ms / 1e9, lane_instr * 4 / ms / 1e9, (double)ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wg_per_cu));
    }
    return 0;
}
