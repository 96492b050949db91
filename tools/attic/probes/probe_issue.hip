// probe_issue.hip — issue cost of individual gfx950 instructions used by the scoring epilogues, relative to v_fma_f32
// (4 cycles per wave64 instruction on a 16-lane SIMD).  Every test runs 8 independent dependency chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 4000
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define OP_FMA(i)    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
#define OP_EXP(i)    asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
#define OP_MIX32(i)  asm volatile("v_fma_mix_f32 %0, %0, %1, %1 op_sel_hi:[1,0,0]" : "+v"(x[i]) : "v"(c));
#define OP_MIXLO(i)  asm volatile("v_fma_mixlo_f16 %0, %0, %1, 0 op_sel_hi:[1,0,0]" : "+v"(x[i]) : "v"(c));
#define OP_MIXHI(i)  asm volatile("v_fma_mixhi_f16 %0, %0, %1, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x[i]) : "v"(c));
#define OP_CVTPK(i)  asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
#define OP_CVT16(i)  asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(x[i]));
#define OP_CVT32(i)  asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(x[i]));
#define OP_PKMAX(i)  asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(x[i]) : "v"(c));
#define OP_PKADD(i)  asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(cc));
#define OP_PKMUL(i)  asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(cc));
#define OP_MAX3(i)   asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
#define OP_SUB(i)    asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
#define OP_MUL(i)    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
#define OP_NOP(i)    asm volatile("s_nop 0");
#define OP_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(c));
#define OP_LOG(i)    asm volatile("v_log_f32 %0, %0" : "+v"(x[i]));
#define OP_RCP(i)    asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
#define OP_EXP16(i)  asm volatile("v_exp_f16 %0, %0" : "+v"(x[i]));
#define OP_LDEXP(i)  asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
#define OP_PKFMA16(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
#define OP_PKFMA32(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y[i]) : "v"(cc));

#define OP_FMA_S(i)   asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "s"(sc));
#define OP_MUL_S(i)   asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i]) : "s"(sc));
#define OP_MIX32_S(i) asm volatile("v_fma_mix_f32 %0, %0, %1, %0 op_sel_hi:[1,0,0]" : "+v"(x[i]) : "s"(sc));
#define OP_MIXLO_S(i) asm volatile("v_fma_mixlo_f16 %0, %0, %1, 0 op_sel_hi:[1,0,0]" : "+v"(x[i]) : "s"(sc));
#define OP_MUL_LIT(i) asm volatile("v_mul_f32 %0, 0x3fb8aa3b, %0" : "+v"(x[i]));
#define OP_FMA_LIT(i) asm volatile("v_fma_f32 %0, %0, 0x3fb8aa3b, %0" : "+v"(x[i]));
#define OP_CND_S(i)   asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "s"(mask));
#define OP_CMP(i)     asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(x[i]), "v"(c) : "vcc");
#define OP_MFMA(i)    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+v"(acc[i & 1]) : "v"(ab));
#define OP_FMA3(i)    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[(i + 1) & 7]), "v"(x[(i + 2) & 7]), "v"(x[(i + 3) & 7]));
#define OP_DSR(i)     asm volatile("ds_read_b128 %0, %1" : "=v"(q[i & 1]) : "v"(ldsaddr));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define KERNEL(NAME, OP)                                                                      \
    __global__ void k_##NAME(float* out, float seed) {                                        \
        float x[8]; f2 y[8]; float c = seed * 0.5f; f2 cc = {c, c};                           \
        const float sc = __builtin_amdgcn_readfirstlane(seed * 0.25f);                        \
        const unsigned long mask = __builtin_amdgcn_ballot_w64(threadIdx.x & 1);              \
        f16v acc[2] = {}; h8 ab = {}; u4 q[2] = {}; const int ldsaddr = (threadIdx.x & 63) * 16; \
        for (int i = 0; i < 8; ++i) { x[i] = seed + i; y[i] = f2{seed + i, seed - i}; }       \
        for (int it = 0; it < ITERS; ++it) { REP8(OP) REP8(OP) REP8(OP) REP8(OP) }            \
        float s = acc[0][0] + acc[1][3] + q[0][0] + q[1][1]; for (int i = 0; i < 8; ++i) s += x[i] + y[i][0] + y[i][1]; \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                       \
    }
KERNEL(fma, OP_FMA) KERNEL(exp, OP_EXP) KERNEL(mix32, OP_MIX32) KERNEL(mixlo, OP_MIXLO) KERNEL(mixhi, OP_MIXHI)
KERNEL(cvtpk, OP_CVTPK) KERNEL(cvt16, OP_CVT16) KERNEL(cvt32, OP_CVT32) KERNEL(pkmax, OP_PKMAX) KERNEL(pkadd, OP_PKADD)
KERNEL(pkmul, OP_PKMUL) KERNEL(max3, OP_MAX3) KERNEL(sub, OP_SUB) KERNEL(mul, OP_MUL) KERNEL(nop, OP_NOP) KERNEL(cndmask, OP_CNDMASK)
KERNEL(fma_s, OP_FMA_S) KERNEL(mul_s, OP_MUL_S) KERNEL(mix32_s, OP_MIX32_S) KERNEL(mixlo_s, OP_MIXLO_S) KERNEL(mul_lit, OP_MUL_LIT)
KERNEL(cnd_s, OP_CND_S) KERNEL(cmp, OP_CMP) KERNEL(mfma, OP_MFMA) KERNEL(fma3, OP_FMA3) KERNEL(dsr, OP_DSR)
KERNEL(log, OP_LOG) KERNEL(rcp, OP_RCP) KERNEL(exp16, OP_EXP16) KERNEL(ldexp, OP_LDEXP) KERNEL(pkfma16, OP_PKFMA16) KERNEL(pkfma32, OP_PKFMA32)

static double base_ms[2];
template <typename K> void run(const char* name, K kern, float* d, bool is_base = false) {
    int idx = 0;
    for (int threads : {256, 512}) {  // 1 and 2 waves per SIMD, one block per CU
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        kern<<<256, threads>>>(d, 1.0f);
        hipEventRecord(a); kern<<<256, threads>>>(d, 1.0f); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (is_base) base_ms[idx] = ms;
        printf("%-10s waves/SIMD=%d  %.3f ms   %.2f x v_fma_f32  (= %.1f cycles / wave-instruction)\n", name, threads / 256, ms,
               ms / base_ms[idx], 4.0 * ms / base_ms[idx]);
        ++idx;
    }
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    run("fma", k_fma, d, true);
#define RUN(N) run(#N, k_##N, d);
    RUN(exp) RUN(log) RUN(rcp) RUN(exp16) RUN(mix32) RUN(mixlo) RUN(mixhi) RUN(cvtpk) RUN(cvt16) RUN(cvt32) RUN(pkmax) RUN(pkadd) RUN(pkmul)
    RUN(fma_s) RUN(mul_s) RUN(mix32_s) RUN(mixlo_s) RUN(mul_lit) RUN(cnd_s) RUN(cmp) RUN(fma3) RUN(mfma) RUN(dsr)
    RUN(pkfma16) RUN(pkfma32) RUN(max3) RUN(sub) RUN(mul) RUN(ldexp) RUN(cndmask) RUN(nop)
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
}
