// probe_valu.hip — VALU issue-rate probe on gfx950: cycles per wave-instruction for the ops of the scoring rounding chain.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 16
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef __bf16 b2v __attribute__((ext_vector_type(2)));
#define ITERS 2000
template <int OP>
__global__ void k(float* out, float seed) {
    float x[N];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = seed + i + threadIdx.x * 0.001f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (OP == 0) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);
            if (OP == 1) { _Float16 h = (_Float16)x[i]; x[i] = (float)h + 1.0f; }              // cvt16 + cvt32 + add
            if (OP == 2) x[i] = __builtin_amdgcn_exp2f(x[i]) * 0.5f;                            // exp + mul
            if (OP == 3) { _Float16 h = (_Float16)x[i]; _Float16 h2 = (_Float16)((float)h * 0.0883f); x[i] = (float)h2 + 1.0f; }  // full chain + add
            if (OP == 4) { __bf16 h = (__bf16)x[i]; __bf16 h2 = (__bf16)((float)h * 0.0883f); x[i] = (float)h2 + 1.0f; }
            if (OP == 8) { _Float16 h = (_Float16)x[i]; float f = (float)h; asm volatile("" : "+v"(f)); float d = f * 0.0883f; asm volatile("" : "+v"(d)); _Float16 h2 = (_Float16)d; float v = (float)h2; asm volatile("" : "+v"(v)); x[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(v, 1.44f, -3.f)) + 1.0f; }
            if (OP == 6 && (i & 1) == 0) { f2 a = {x[i], x[i + 1]}; h2v h = __builtin_convertvector(a, h2v); f2 f = __builtin_convertvector(h, f2); f2 d = f * 0.0883f; h2v hh = __builtin_convertvector(d, h2v); f2 v = __builtin_convertvector(hh, f2); f2 y = v * 1.44f - 3.f; f2 e = {__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])}; e = e + 1.0f; x[i] = e[0]; x[i + 1] = e[1]; }
            if (OP == 7 && (i & 1) == 0) { f2 a = {x[i], x[i + 1]}; b2v h = __builtin_convertvector(a, b2v); f2 f = __builtin_convertvector(h, f2); f2 d = f * 0.0883f; b2v hh = __builtin_convertvector(d, b2v); f2 v = __builtin_convertvector(hh, f2); f2 y = v * 1.44f - 3.f; f2 e = {__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])}; e = e + 1.0f; x[i] = e[0]; x[i + 1] = e[1]; }
            if (OP == 9 && (i & 1) == 0) { f2 a = {x[i], x[i + 1]}; h2v h = __builtin_convertvector(a, h2v); f2 f = __builtin_convertvector(h, f2); f2 d = f * 0.0883f; h2v hh = __builtin_convertvector(d, h2v); f2 v = __builtin_convertvector(hh, f2); f2 y = v - 3.f; y = y - 0.25f; x[i] = fmaxf(y[0], x[i]) ; x[i + 1] = fmaxf(y[1], x[i+1]); }
            if (OP == 5) { _Float16 h = (_Float16)x[i]; _Float16 h2 = (_Float16)((float)h * 0.0883f); float v = (float)h2; x[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(v, 1.44f, -3.f)) + 1.0f; } // chain + fma + exp + add
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, int insts, float* d) {
    for (int wpb : {256, 1024}) {  // 1 wave/SIMD and 4 waves/SIMD per CU (one block per CU)
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        k<OP><<<256, wpb>>>(d, 1.0f);
        hipEventRecord(a); k<OP><<<256, wpb>>>(d, 1.0f); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double wave_insts = (double)ITERS * N * insts;                    // per wave
        double cyc = ms * 1e-3 * 2.4e9;                                    // assume 2.4 GHz
        printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz, %d instr/elem)\n", name, wpb / 256, ms,
               cyc / (wave_insts * (wpb / 256)), insts);
    }
}
int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    run<0>("v_fma_f32", 1, d);
    run<1>("cvt_f16+cvt_f32+add", 3, d);
    run<2>("exp2+mul", 2, d);
    run<3>("f16 chain (4)+add", 5, d);
    run<4>("bf16 chain+add", 5, d);
    run<5>("f16 chain+fma+exp+add", 7, d);
    run<8>("f16 nomix chain+fma+exp+add", 8, d);
    run<6>("PACKED f16 chain+fma+exp+add (per elem)", 1, d);
    run<7>("PACKED bf16 chain+fma+exp+add (per elem)", 1, d);
    run<9>("PACKED f16 chain+2sub+max (per elem)", 1, d);
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
}
