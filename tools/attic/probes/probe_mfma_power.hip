// probe_mfma_power.hip — what does the matrix pipe sustain at the package's power limit, by MFMA shape, dtype and operand data?
// Every wave issues independent MFMAs back to back (4 accumulators), 1 or 2 waves per SIMD, operands constant or pseudo-random per lane.
// Reported: TFLOP/s for the chip.  (Round 5: random fp16 operands pull v_mfma_f32_32x32x16_f16 down to 1.65 GHz = 1.72 PFLOP/s.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
#define ITERS 2000
template <int SHAPE, bool BF16>
__global__ __launch_bounds__(512) void k(float* out, unsigned seed) {
    h8 a, b; b8 ab, bb;
    unsigned x = seed ^ (threadIdx.x * 747796405u) ^ (blockIdx.x * 2891336453u);
    for (int i = 0; i < 8; ++i) {
        float va = 1.0f, vb = 0.5f;
        if (seed) { x = x * 1664525u + 1013904223u; va = ((int)((x >> 9) & 1023) - 512) * (1.0f / 512.f); vb = ((int)((x >> 19) & 1023) - 512) * (1.0f / 512.f); }
        a[i] = (_Float16)va; b[i] = (_Float16)vb; ab[i] = (__bf16)va; bb[i] = (__bf16)vb;
    }
    float r = 0.f;
    if (SHAPE == 32) {
        f16v acc[4]; for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (BF16) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[n], 0, 0, 0);
                else acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
            }
        }
        for (int n = 0; n < 4; ++n) r += acc[n][0] + acc[n][9];
    } else {
        f4v acc[8]; for (int n = 0; n < 8; ++n) for (int i = 0; i < 4; ++i) acc[n][i] = 0.f;
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                if (BF16) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[n], 0, 0, 0);
                else acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[n], 0, 0, 0);
            }
        }
        for (int n = 0; n < 8; ++n) r += acc[n][0] + acc[n][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int SHAPE, bool BF16> void run(float* d, int threads, unsigned seed) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<SHAPE, BF16><<<256, threads>>>(d, seed);
    (void)hipEventRecord(e0); k<SHAPE, BF16><<<256, threads>>>(d, seed); k<SHAPE, BF16><<<256, threads>>>(d, seed); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 2;
    const double flops_per_mfma = SHAPE == 32 ? 32.0 * 32 * 16 * 2 : 16.0 * 16 * 32 * 2;
    const double n = (double)ITERS * (SHAPE == 32 ? 4 : 8) * 256 * (threads / 64);
    printf("%s %s  waves/SIMD %d  operands %-8s: %7.1f us  %7.1f TFLOP/s\n", SHAPE == 32 ? "32x32x16" : "16x16x32", BF16 ? "bf16" : "f16 ", threads / 256,
           seed ? "random" : "constant", ms * 1e3, n * flops_per_mfma / (ms * 1e-3) / 1e12);
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 512 * 4);
    for (unsigned seed : {0u, 12345u})
        for (int threads : {256, 512}) {
            run<32, false>(d, threads, seed); run<32, true>(d, threads, seed); run<16, false>(d, threads, seed); run<16, true>(d, threads, seed);
        }
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
}
