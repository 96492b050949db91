// probe_overlap.hip — can MFMA (matrix pipe) and VALU work of DIFFERENT waves on the SAME SIMD overlap on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define ITERS 400
// mode bits per wave group (waves 0-3 = group 0, waves 4-7 = group 1): 1 = MFMA phase work, 2 = VALU phase work
__device__ __forceinline__ void mfma_phase(f16v& acc, h8 a, h8 b) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void valu_phase(float (&x)[16]) {
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.25f);   // 96 independent-ish VALU
}
template <int M0, int M1, bool ALT>
__global__ __launch_bounds__(512) void k(float* out) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mode = wave < 4 ? M0 : M1;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f16v acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float x[16]; for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
    for (int it = 0; it < ITERS; ++it) {
        if (ALT) {  // every wave alternates MFMA and VALU phases; group 1 starts with the VALU phase (anti-phase)
            if (mode == 1) { mfma_phase(acc, a, b); x[0] += acc[0]; valu_phase(x); }
            else { valu_phase(x); mfma_phase(acc, a, b); x[0] += acc[0]; }
        } else {
            if (mode & 1) mfma_phase(acc, a, b);
            if (mode & 2) valu_phase(x);
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i] + acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// in-wave interleave: 8 x [1 MFMA + NV independent VALU] per iteration, pinned with sched_barrier
template <int NV, bool DO_MFMA, bool DO_VALU>
__global__ __launch_bounds__(512) void kin(float* out) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f16v acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float x[16]; for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (DO_MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (DO_VALU) {
#pragma unroll
                for (int i = 0; i < NV; ++i) x[(m * NV + i) & 15] = __builtin_fmaf(x[(m * NV + i) & 15], 1.0001f, 0.25f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i] + acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, bool DM, bool DV> float runin(float* d, int threads) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kin<NV, DM, DV><<<256, threads>>>(d);
    hipEventRecord(e0); kin<NV, DM, DV><<<256, threads>>>(d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}
template <int M0, int M1, bool ALT> float run(float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<M0, M1, ALT><<<256, 512>>>(d);
    hipEventRecord(e0); k<M0, M1, ALT><<<256, 512>>>(d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    printf("one 512-thread block per CU: waves w and w+4 share a SIMD.  times in us for %d iterations\n", ITERS);
    printf("A  both groups MFMA only        : %.1f\n", run<1, 1, false>(d));
    printf("B  both groups VALU only        : %.1f\n", run<2, 2, false>(d));
    printf("C  group0 MFMA, group1 VALU     : %.1f   (overlap => ~max(A,B)/2.. , no overlap => (A+B)/2)\n", run<1, 2, false>(d));
    printf("D  every wave MFMA then VALU (in phase)   : %.1f\n", run<1, 1, true>(d));
    printf("E  group0 MFMA->VALU, group1 VALU->MFMA   : %.1f\n", run<1, 2, true>(d));
    printf("F  one group only, MFMA only (4 waves)    : %.1f\n", run<1, 0, false>(d));
    printf("G  one group only, VALU only (4 waves)    : %.1f\n", run<2, 0, false>(d));
    for (int threads : {256, 512}) {
        printf("in-wave interleave, %d waves/SIMD:  NV=6: mfma-only %.1f  valu-only %.1f  both %.1f |  NV=12: mfma-only %.1f valu-only %.1f both %.1f\n",
               threads / 256, runin<6, true, false>(d, threads), runin<6, false, true>(d, threads), runin<6, true, true>(d, threads),
               runin<12, true, false>(d, threads), runin<12, false, true>(d, threads), runin<12, true, true>(d, threads));
    }
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
}
