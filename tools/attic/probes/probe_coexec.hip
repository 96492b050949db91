// probe_coexec.hip — WHICH VALU instructions execute beside a running v_mfma_f32_32x32x16_f16 on one gfx950 SIMD?
// Round 1 (probe_overlap) found that v_fma_f32 does not overlap with the matrix pipe at all; probe_pipe found that the scoring
// epilogue's mix (conversions, mixed-precision fma, exp2, adds) overlaps to about two thirds.  This probe takes the instruction
// forms one at a time: a dependent chain of 8 MFMAs per iteration, NV filler instructions of ONE form after every MFMA
// (independent registers, pinned with sched_barrier), 1 and 2 waves per SIMD.  Reported per form: matrix only, fillers only,
// both, and the share of the shorter part that was hidden: (m + v - both) / min(m, v).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define ITERS 400

enum Form { FMA_VVV, FMA_VSS, MUL_VV, ADD_VV, ADD_VS, EXP, CVTPK, MIX_VS0, MIX_VSV, MIX_VVV, MAX3, PKFMA16, PKMUL16, MAX_VV, MOV, CVT_F32_F16, NFORMS };
static const char* names[] = {"v_fma_f32 v,v,v", "v_fma_f32 v,s,s", "v_mul_f32 v,v", "v_add_f32 v,v", "v_add_f32 s,v", "v_exp_f32", "v_cvt_pk_f16_f32 v,v",
                              "v_fma_mix_f32 v,s,0", "v_fma_mix_f32 v,s,v", "v_fma_mix_f32 v,v,v", "v_max3_f32 v,v,v", "v_pk_fma_f16 v,v,v", "v_pk_mul_f16 v,s",
                              "v_max_f32 v,v", "v_mov_b32", "v_cvt_f32_f16"};

template <int F> __device__ __forceinline__ void filler(float& d, float a, float b, float s) {
    if (F == FMA_VVV) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d) : "v"(a), "v"(b));
    if (F == FMA_VSS) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(d) : "s"(s));
    if (F == MUL_VV) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(d) : "v"(a));
    if (F == ADD_VV) asm volatile("v_add_f32 %0, %0, %1" : "+v"(d) : "v"(a));
    if (F == ADD_VS) asm volatile("v_add_f32 %0, %1, %0" : "+v"(d) : "s"(s));
    if (F == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(d));
    if (F == CVTPK) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(d) : "v"(a));
    if (F == MIX_VS0) asm volatile("v_fma_mix_f32 %0, %0, %1, 0 op_sel_hi:[1,0,0]" : "+v"(d) : "s"(s));
    if (F == MIX_VSV) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(d) : "s"(s), "v"(b));
    if (F == MIX_VVV) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(d) : "v"(a), "v"(b));
    if (F == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(d) : "v"(a), "v"(b));
    if (F == PKFMA16) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(d) : "v"(a), "v"(b));
    if (F == PKMUL16) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(d) : "s"(s));
    if (F == MAX_VV) asm volatile("v_max_f32 %0, %0, %1" : "+v"(d) : "v"(a));
    if (F == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(a));
    if (F == CVT_F32_F16) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(d));
}

template <int F, int NV, bool DO_MFMA, bool DO_VALU>
__global__ __launch_bounds__(512) void k(float* out, float s) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f16v acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float x[16]; for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i * 0.01f;
    float c0 = 1.0001f + threadIdx.x * 1e-9f, c1 = 0.25f;
    asm volatile("" : "+v"(c0), "+v"(c1));
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (DO_MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (DO_VALU) {
#pragma unroll
                for (int i = 0; i < NV; ++i) filler<F>(x[(m * NV + i) & 15], c0, c1, s);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = 0; for (int i = 0; i < 16; ++i) r += x[i] + acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int F, int NV, bool DM, bool DV> float run(float* d, int threads) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<F, NV, DM, DV><<<256, threads>>>(d, 1.0001f);
    hipEventRecord(e0); k<F, NV, DM, DV><<<256, threads>>>(d, 1.0001f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}
template <int F, int NV> void form(float* d) {
    for (int threads : {256, 512}) {
        const float m = run<F, NV, true, false>(d, threads), v = run<F, NV, false, true>(d, threads), both = run<F, NV, true, true>(d, threads);
        printf("%-22s NV=%2d waves/SIMD %d : mfma %6.1f  valu %6.1f  both %6.1f us   hidden %.2f of the shorter part\n", names[F], NV, threads / 256, m, v, both,
               (m + v - both) / (m < v ? m : v));
    }
}
template <int F> void forms(float* d) { form<F, 4>(d); form<F, 8>(d); }
int main() {
    float* d; (void)hipMalloc(&d, 256 * 512 * 4);
    forms<FMA_VVV>(d); forms<FMA_VSS>(d); forms<MUL_VV>(d); forms<ADD_VV>(d); forms<ADD_VS>(d); forms<EXP>(d); forms<CVTPK>(d);
    forms<MIX_VS0>(d); forms<MIX_VSV>(d); forms<MIX_VVV>(d); forms<MAX3>(d); forms<PKFMA16>(d); forms<PKMUL16>(d); forms<MAX_VV>(d); forms<MOV>(d);
    forms<CVT_F32_F16>(d);
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
}
