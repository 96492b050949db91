// probe_stage.hip — L2 -> LDS staging throughput per CU on gfx950: LDS-DMA (global_load_lds_dwordx4) versus
// global_load_dwordx4 + ds_write_b128, for an L2-resident source (every block re-reads the same 1 MiB).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define TILE 32768
#define ITERS 400

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const char* src, float* out, int ntiles) {
    __shared__ __attribute__((aligned(16))) char lds[2 * TILE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = THREADS / 64, PER = TILE / 1024 / NW;
    float acc = 0;
    for (int it = 0; it < ITERS; ++it) {
        const char* tile = src + (size_t)((it * 7 + blockIdx.x) % ntiles) * TILE;
        char* buf = lds + (it & 1) * TILE;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int ci = i * NW + wave;
                __builtin_amdgcn_global_load_lds((gptr_t)(tile + ci * 1024 + lane * 16), (lptr_t)(buf + ci * 1024), 16, 0, 0);
            }
        } else {
            u4 r[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) r[i] = *(const u4*)(tile + (i * NW + wave) * 1024 + lane * 16);
#pragma unroll
            for (int i = 0; i < PER; ++i) *(u4*)(buf + (i * NW + wave) * 1024 + lane * 16) = r[i];
        }
        __syncthreads();
        acc += *(float*)(buf + threadIdx.x * 4);
    }
    out[blockIdx.x * THREADS + threadIdx.x] = acc;
}
template <int MODE, int THREADS> void run(const char* name, const char* src, float* out, int blocks) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE, THREADS><<<blocks, THREADS>>>(src, out, 32);
    hipEventRecord(a); k<MODE, THREADS><<<blocks, THREADS>>>(src, out, 32); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double bytes = (double)blocks * ITERS * TILE;
    printf("%-34s blocks=%d threads=%d: %.3f ms  %.2f TB/s  (%.1f B/clk/CU at 2.0 GHz, 256 CUs)\n", name, blocks, THREADS, ms, bytes / ms * 1e-9,
           bytes / (ms * 1e-3) / 256 / 2.0e9);
}
int main() {
    char* src; float* out; hipMalloc(&src, 32 * TILE); hipMemset(src, 1, 32 * TILE); hipMalloc(&out, 1024 * 512 * 4);
    for (int blocks : {256, 512}) {
        run<0, 256>("LDS-DMA dwordx4, 4 waves", src, out, blocks);
        run<1, 256>("global_load + ds_write, 4 waves", src, out, blocks);
        run<0, 512>("LDS-DMA dwordx4, 8 waves", src, out, blocks);
        run<1, 512>("global_load + ds_write, 8 waves", src, out, blocks);
    }
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
}
