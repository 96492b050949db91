// probe_mfma_dep.hip — issue cadence of v_mfma_f32_32x32x16_f16 chains, pinned with inline assembly:
//   dep8      : one chain of 8 dependent MFMAs (acc -> acc), distinct A/B operands, accumulator re-zeroed per chain
//   inter2x8  : two chains of 8, strictly alternating (acc0, acc1, acc0, ...)
//   indep16   : 16 MFMAs into 16 different accumulators (no dependency at all)
// for 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define ITERS 3000
#define MF(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MF0(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b))
template <int MODE>
__global__ void k(const h8* in, float* out) {
    const int lane = threadIdx.x & 63;
    h8 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[lane + 64 * i]; b[i] = in[lane + 64 * (i + 8)]; }
    float sum = 0.f;
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) {
            f16v c0, c1;
            MF0(c0, a[0], b[0]); MF(c0, a[1], b[1]); MF(c0, a[2], b[2]); MF(c0, a[3], b[3]); MF(c0, a[4], b[4]); MF(c0, a[5], b[5]); MF(c0, a[6], b[6]); MF(c0, a[7], b[7]);
            MF0(c1, a[0], b[7]); MF(c1, a[1], b[6]); MF(c1, a[2], b[5]); MF(c1, a[3], b[4]); MF(c1, a[4], b[3]); MF(c1, a[5], b[2]); MF(c1, a[6], b[1]); MF(c1, a[7], b[0]);
            sum += c0[0] + c1[3];
        } else if (MODE == 1) {
            f16v c0, c1;
            MF0(c0, a[0], b[0]); MF0(c1, a[0], b[7]); MF(c0, a[1], b[1]); MF(c1, a[1], b[6]); MF(c0, a[2], b[2]); MF(c1, a[2], b[5]); MF(c0, a[3], b[3]); MF(c1, a[3], b[4]);
            MF(c0, a[4], b[4]); MF(c1, a[4], b[3]); MF(c0, a[5], b[5]); MF(c1, a[5], b[2]); MF(c0, a[6], b[6]); MF(c1, a[6], b[1]); MF(c0, a[7], b[7]); MF(c1, a[7], b[0]);
            sum += c0[0] + c1[3];
        } else {
            f16v c[8];
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) MF0(c[i], a[i], b[(i + r) & 7]);
#pragma unroll
                for (int i = 0; i < 8; ++i) sum += c[i][i];
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
template <int MODE> void run(const char* name, const h8* in, float* d) {
    for (int threads : {256, 512}) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        k<MODE><<<256, threads>>>(in, d);
        hipEventRecord(a); k<MODE><<<256, threads>>>(in, d); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-10s waves/SIMD=%d  %.3f ms  %.1f ns per MFMA per SIMD  (= %.1f cycles at 2.35 GHz)\n", name, threads / 256, ms,
               ms * 1e6 / (ITERS * 16.0 * (threads / 256)), ms * 1e6 / (ITERS * 16.0 * (threads / 256)) * 2.35);
    }
}
int main() {
    h8* in; float* d; hipMalloc(&in, 64 * 16 * 16); hipMemset(in, 0, 64 * 16 * 16); hipMalloc(&d, 256 * 512 * 4);
    run<0>("dep8", in, d);
    run<1>("inter2x8", in, d);
    run<2>("indep16", in, d);
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
}
