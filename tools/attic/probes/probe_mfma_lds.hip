// probe_mfma_lds.hip — how fast can 8-wave blocks (2 per CU) run the scoring inner loop  [ds_read_b128 -> v_mfma_32x32x16] ?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define TILES 64
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }
// MODE 0: reads sunk next to each MFMA (compiler default)   1: 8 reads pinned ahead of 8 MFMAs   2: no LDS (register operands)
// MODE 3: like 1 with two independent accumulators (2 blocks of 32 keys interleaved)
template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 8 ? 4 : 2)) void k(float* out, unsigned long long* clk) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    __shared__ __attribute__((aligned(16))) char lds[32768];
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    for (int i = threadIdx.x; i < 32768 / 4; i += NW * 64) ((float*)lds)[i] = (float)(i & 1023) * 1e-3f;
    __syncthreads();
    h8 bq[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) { u32x4 w = {0x2c002c00u + lane, 0x2e002a00u + kk, 0x2c003000u, 0x28002c00u + lane * 3}; bq[kk] = __builtin_bit_cast(h8, w); }
    float total = 0.f;
    for (int t = 0; t < TILES; ++t) {
        if (MODE == 3) {
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                f16v a0, a1;
                for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
                u32x4 f0[8], f1[8];
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) { f0[kk] = *(const u32x4*)(lds + lds_off(kp * 64 + l31, kk * 2 + half)); f1[kk] = *(const u32x4*)(lds + lds_off(kp * 64 + 32 + l31, kk * 2 + half)); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) { a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, f0[kk]), bq[kk], a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, f1[kk]), bq[kk], a1, 0, 0, 0); }
                __builtin_amdgcn_sched_barrier(0);
                total += a0[0] + a1[5];
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                f16v acc;
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                if (MODE == 0) {
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        const u32x4 raw = *(const u32x4*)(lds + lds_off(kb * 32 + l31, kk * 2 + half));
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, raw), bq[kk], acc, 0, 0, 0);
                    }
                } else if (MODE == 1) {
                    u32x4 fr[8];
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) fr[kk] = *(const u32x4*)(lds + lds_off(kb * 32 + l31, kk * 2 + half));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, fr[kk]), bq[kk], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bq[(kk + kb) & 7], bq[kk], acc, 0, 0, 0);
                }
                total += acc[0] + acc[7];
            }
        }
        __syncthreads();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = total;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}
template <int MODE, int NW> void run(const char* name, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 512 * (8 / NW);
    unsigned long long* clk; hipMalloc(&clk, 16); unsigned long long h[2];
    k<MODE, NW><<<blocks, NW * 64>>>(d, clk);
    hipEventRecord(e0); k<MODE, NW><<<blocks, NW * 64>>>(d, clk); hipEventRecord(e1); hipEventSynchronize(e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * NW * TILES * 32;
    printf("%-58s %.1f us  -> %.1f cycles@2.4GHz per MFMA per SIMD, %.0f TFLOP/s | block0: %.0f shader cycles in %.1f us = %.0f MHz\n", name, ms * 1e3, ms * 1e-3 * 2.4e9 / (mfmas / 1024), mfmas * 32768.0 / (ms * 1e-3) / 1e12, (double)h[0], h[1] / 100.0, (double)h[0] / (h[1] / 100.0));
}
int main() {
    float* d; hipMalloc(&d, 4096 * 512 * 4);
    run<2, 8>("8 waves: MFMA only (register operands)", d);
    run<0, 8>("8 waves: ds_read_b128 + MFMA, compiler order", d);
    run<1, 8>("8 waves: 8 reads pinned ahead of 8 MFMAs", d);
    run<3, 8>("8 waves: 16 reads ahead, two accumulators", d);
    run<2, 4>("4 waves: MFMA only (register operands)", d);
    run<0, 4>("4 waves: ds_read_b128 + MFMA, compiler order", d);
    run<1, 4>("4 waves: 8 reads pinned ahead of 8 MFMAs", d);
    run<3, 4>("4 waves: 16 reads ahead, two accumulators", d);
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
}
