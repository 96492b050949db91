// probe_pipe.hip — what does one gfx950 SIMD make of the scoring inner stream?  32 MFMA (32x32x16 f16) "slots" per iteration,
// round-robin over NACC independent accumulators, each slot followed by the real epilogue instruction blocks of kvz_score.hip
// (cvt_pk x2 + 8-instruction rounding-chain block on even slots, 4 x exp + 4 x add on odd slots: 9 VALU per slot on average).
// Reports cycles (s_memtime) per slot for: MFMA only / fillers only / interleaved / burst-then-fillers, 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define ITERS 300

__device__ __forceinline__ void fillA(float a0, float a1, float a2, float a3, float rcp, float l2e, float nm, float (&arg)[4]) {
    unsigned xa = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{a0, a1}, h2v));
    unsigned xb = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{a2, a3}, h2v));
    asm volatile("v_fma_mixlo_f16 %[xa], %[xa], %[r], 0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %[xb], %[xb], %[r], 0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %[xa], %[xa], %[r], 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %[xb], %[xb], %[r], 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %[g0], %[xa], %[l2e], %[nm] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %[g2], %[xb], %[l2e], %[nm] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %[g1], %[xa], %[l2e], %[nm] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %[g3], %[xb], %[l2e], %[nm] op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : [xa] "+v"(xa), [xb] "+v"(xb), [g0] "=&v"(arg[0]), [g1] "=&v"(arg[1]), [g2] "=&v"(arg[2]), [g3] "=&v"(arg[3])
        : [r] "s"(rcp), [l2e] "v"(l2e), [nm] "v"(nm));
}
__device__ __forceinline__ void fillB(const float (&arg)[4], float& p0, float& p1) {
    float e0, e1, e2, e3;
    asm volatile("v_exp_f32 %[e0], %[a0]\n\tv_exp_f32 %[e1], %[a1]\n\tv_exp_f32 %[e2], %[a2]\n\tv_exp_f32 %[e3], %[a3]\n\t"
        "v_add_f32 %[p0], %[p0], %[e0]\n\tv_add_f32 %[p1], %[p1], %[e1]\n\tv_add_f32 %[p0], %[p0], %[e2]\n\tv_add_f32 %[p1], %[p1], %[e3]"
        : [e0] "=&v"(e0), [e1] "=&v"(e1), [e2] "=&v"(e2), [e3] "=&v"(e3), [p0] "+v"(p0), [p1] "+v"(p1)
        : [a0] "v"(arg[0]), [a1] "v"(arg[1]), [a2] "v"(arg[2]), [a3] "v"(arg[3]));
}
// MODE 0 MFMA only, 1 fillers only, 2 interleaved (one slot = MFMA + filler block), 3 burst (32 MFMA, then 32 filler blocks)
// FILL: filler blocks per slot x 2 (2 = the real load: A on even, B on odd slots; 1 = half of it; 4 = double)
template <int NACC, int MODE, int FILL, int BAR = 0>
__global__ __launch_bounds__(512) void kp(float* out, unsigned long long* cyc, float rcp, unsigned seed) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f16v acc[4];
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    float src[16]; for (int i = 0; i < 16; ++i) src[i] = threadIdx.x * 0.01f + i;
    float p0 = 0.f, p1 = 0.f, arg[4] = {0.f, 0.f, 0.f, 0.f};
    const float l2e = 1.4426950408889634f, nm = -3.f;
    __shared__ __attribute__((aligned(16))) char lds[65536];
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    for (int o = threadIdx.x * 16; o < 65536; o += blockDim.x * 16) {
        u4 w = u4{0x3c003c00u, 0x3c003c00u, 0x38003800u, 0x34003400u};
        if (seed) {   // random fp16 values of magnitude 0.5 .. 2 with random signs and mantissas: realistic toggling on the LDS, MFMA and VALU paths
            unsigned x = seed ^ (o * 2654435761u) ^ (blockIdx.x * 40503u);
            for (int j = 0; j < 4; ++j) { x = x * 1664525u + 1013904223u; unsigned r = x >> 8; w[j] = (r & 0x83ff83ffu) | 0x38003800u | ((r >> 3) & 0x04000400u); }
        }
        *(u4*)(lds + o) = w;
    }
    if (seed) { unsigned x = seed ^ (threadIdx.x * 747796405u); for (int i = 0; i < 8; ++i) { x = x * 1664525u + 1013904223u; b[i] = (_Float16)(((int)((x >> 9) & 1023) - 512) * (1.0f / 512.f)); a[i] = (_Float16)(((int)((x >> 19) & 1023) - 512) * (1.0f / 512.f)); } }
    __syncthreads();
    u4 fr[2][8];
    const int fo = (threadIdx.x & 63) * 16;
    for (int k = 0; k < 8; ++k) { fr[0][k] = *(u4*)(lds + fo + k * 1024); fr[1][k] = *(u4*)(lds + fo + 8192 + k * 1024); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 3) {
#pragma unroll
            for (int s = 0; s < 32; ++s) acc[s % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[s % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < (MODE >= 4 ? 0 : 32); ++s) {
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 0 || MODE == 2) acc[s % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[s % NACC], 0, 0, 0);
            if (MODE != 0) {
#pragma unroll
                for (int f = 0; f < FILL; f += 2) {
                    if ((s & 1) == 0 || FILL >= 4) fillA(src[(4 * s) & 15], src[(4 * s + 1) & 15], src[(4 * s + 2) & 15], src[(4 * s + 3) & 15], rcp, l2e, nm, arg);
                    if ((s & 1) == 1 || FILL >= 4) fillB(arg, p0, p1);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 7 || MODE == 8) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                f16v& accn = acc[(st + 1) & 1];
                const f16v& accc = acc[st & 1];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    __builtin_amdgcn_sched_barrier(0);
                    accn = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, fr[st & 1][s]), b, accn, 0, 0, 0);
                    const int q = (s >> 1) * 4;
                    if ((s & 1) == 0) fillA(accc[q], accc[q + 1], accc[q + 2], accc[q + 3], rcp, l2e, nm, arg);
                    else fillB(arg, p0, p1);
                    if (s == 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (MODE == 7) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) fr[(st + 1) & 1][k] = *(u4*)(lds + fo + ((it * 4 + st) & 3) * 16384 + k * 1024);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (BAR == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (BAR == 2 && (threadIdx.x >> 8) == (it & 1)) { for (int z = 0; z < 30; ++z) asm volatile("s_nop 7"); }   // one half of the block falls behind, alternating
        } else if (MODE >= 4) {
            // the kernel's real shape: steps of 8 slots; the chain of a step accumulates into one set while the fillers read the
            // 16 values of the OTHER set (produced by the previous step's chain).  MODE 5: the values are first copied out of the
            // accumulator registers (16 v_mov), MODE 6: the fillers read plain registers (control)
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                f16v& accn = acc[(st + 1) & 1];
                const f16v& accc = acc[st & 1];
                float cp[16];
                if (MODE == 5) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) { cp[i] = accc[i]; asm volatile("" : "+v"(cp[i])); }
                }
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    __builtin_amdgcn_sched_barrier(0);
                    accn = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, accn, 0, 0, 0);
                    const int q = (s >> 1) * 4;
                    if ((s & 1) == 0) {
                        if (MODE == 4) fillA(accc[q], accc[q + 1], accc[q + 2], accc[q + 3], rcp, l2e, nm, arg);
                        else if (MODE == 5) fillA(cp[q], cp[q + 1], cp[q + 2], cp[q + 3], rcp, l2e, nm, arg);
                        else fillA(src[q], src[q + 1], src[q + 2], src[q + 3], rcp, l2e, nm, arg);
                    } else {
                        fillB(arg, p0, p1);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sacc = p0 + p1 + arg[0];
    for (int n = 0; n < 4; ++n) sacc += acc[n][0] + acc[n][7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sacc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
static unsigned g_seed = 0;
template <int NACC, int MODE, int FILL, int BAR = 0> void run(const char* what, float* d, unsigned long long* c, int threads) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kp<NACC, MODE, FILL, BAR><<<256, threads>>>(d, c, 0.0883883461f, g_seed);
    hipEventRecord(e0); kp<NACC, MODE, FILL, BAR><<<256, threads>>>(d, c, 0.0883883461f, g_seed); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    const double slots = (double)ITERS * 32;
    printf("%-34s nacc %d  waves/SIMD %d : %7.1f ns/slot/wave  %6.1f memtime-ticks/slot  (SIMD: %.1f ns per slot-of-any-wave)\n", what, NACC, threads / 256,
           ms * 1e6 / slots, (double)cy / slots, ms * 1e6 / slots / (threads / 256));
}
int main(int argc, char** argv) {
    if (argc > 1) g_seed = (unsigned)atoi(argv[1]);
    printf("operand data: %s\n", g_seed ? "random" : "constant");
    float* d; hipMalloc(&d, 256 * 512 * 4);
    unsigned long long* c; hipMalloc(&c, 8);
    for (int threads : {256, 512}) {
        if (threads == 256) {
            run<1, 0, 2>("mfma only", d, c, 256); run<2, 0, 2>("mfma only", d, c, 256); run<4, 0, 2>("mfma only", d, c, 256);
            run<1, 1, 2>("fillers only (9/slot)", d, c, 256); run<1, 1, 1>("fillers only (4.5/slot)", d, c, 256);
            run<1, 2, 2>("interleaved, 9 fill/slot", d, c, 256); run<2, 2, 2>("interleaved, 9 fill/slot", d, c, 256); run<4, 2, 2>("interleaved, 9 fill/slot", d, c, 256);
            run<4, 2, 1>("interleaved, 4.5 fill/slot", d, c, 256); run<2, 2, 1>("interleaved, 4.5 fill/slot", d, c, 256); run<4, 2, 4>("interleaved, 18 fill/slot", d, c, 256);
            run<1, 3, 2>("burst 32 mfma then fillers", d, c, 256); run<4, 3, 2>("burst 32 mfma then fillers", d, c, 256);
        } else {
            run<1, 0, 2>("mfma only", d, c, 512); run<4, 0, 2>("mfma only", d, c, 512);
            run<1, 1, 2>("fillers only (9/slot)", d, c, 512);
            run<1, 2, 2>("interleaved, 9 fill/slot", d, c, 512); run<2, 2, 2>("interleaved, 9 fill/slot", d, c, 512); run<4, 2, 2>("interleaved, 9 fill/slot", d, c, 512);
            run<4, 2, 1>("interleaved, 4.5 fill/slot", d, c, 512);
            run<1, 3, 2>("burst 32 mfma then fillers", d, c, 512); run<4, 3, 2>("burst 32 mfma then fillers", d, c, 512);
        }
        run<2, 6, 2>("pipelined, fillers on plain regs", d, c, threads);
        run<2, 4, 2>("pipelined, fillers read other acc", d, c, threads);
        run<2, 5, 2>("pipelined, other acc copied first", d, c, threads);
        run<2, 8, 2>("pipelined, A operands in 2 reg sets", d, c, threads);
        run<2, 7, 2>("pipelined, + 8 ds_read_b128 / step", d, c, threads);
        run<2, 7, 2, 1>("  the same + s_barrier / 4 steps", d, c, threads);
        run<2, 8, 2, 1>("  2 reg sets + s_barrier / 4 steps", d, c, threads);
    }
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
}
