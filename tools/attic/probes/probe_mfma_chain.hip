// probe_mfma_chain.hip — issue cadence of v_mfma_f32_32x32x16_f16 for ONE wave per SIMD:
//   (a) one dependent chain of 8, (b) two interleaved chains of 8 (shared A operand, as in the scoring kernels),
//   (c) as (b) with the A fragments re-read from LDS one block ahead, (d) as (c) plus a VALU epilogue on the results.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define ITERS 2000
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const h8* in, float* out) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
    for (int i = threadIdx.x; i < 64 * 1024 / 4; i += 256) ((float*)lds)[i] = (float)(i & 1023) * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    h8 b0[8], b1[8], a[8];
    for (int i = 0; i < 8; ++i) { b0[i] = in[lane + 64 * i]; b1[i] = in[lane + 64 * (i + 8)]; a[i] = in[lane + 64 * (i + 16)]; }
    float sum = 0.f;
    u4 fr[2][8];
    for (int i = 0; i < 8; ++i) fr[0][i] = *(const u4*)(lds + i * 1024 + lane * 16);
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f16v acc0, acc1;
            for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 0) {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], b0[kk], acc0, 0, 0, 0);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], b1[kk], acc1, 0, 0, 0);
            } else if (MODE == 1) {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], b0[kk], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], b1[kk], acc1, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, fr[kb][kk]), b0[kk], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, fr[kb][kk]), b1[kk], acc1, 0, 0, 0);
                }
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) fr[kb ^ 1][kk] = *(const u4*)(lds + ((it + kb) & 3) * 8192 + kk * 1024 + lane * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 3) {
                float s0 = 0, s1 = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    s0 += __builtin_amdgcn_exp2f(acc0[i] * 1e-3f - 3.f);
                    s1 += __builtin_amdgcn_exp2f(acc1[i] * 1e-3f - 3.f);
                }
                sum += s0 + s1;
            } else {
                sum += acc0[0] + acc1[5];
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}
template <int MODE> void run(const char* name, const h8* in, float* d) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256, 256>>>(in, d);
    hipEventRecord(a); k<MODE><<<256, 256>>>(in, d); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-58s %.3f ms  %.1f cycles per MFMA at 2.0 GHz\n", name, ms, ms * 1e-3 * 2.0e9 / (ITERS * 2 * 16));
}
int main() {
    h8* in; float* d; hipMalloc(&in, 64 * 24 * 16); hipMemset(in, 0, 64 * 24 * 16); hipMalloc(&d, 256 * 256 * 4);
    run<0>("two dependent chains of 8, one after the other", in, d);
    run<1>("two interleaved chains (shared A in registers)", in, d);
    run<2>("interleaved chains, A from LDS one block ahead", in, d);
    run<3>("... plus an exp2 epilogue on the 32 results", in, d);
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
}
