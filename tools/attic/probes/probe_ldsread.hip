// probe_ldsread.hip — ds_read_b128 throughput for the fragment access pattern of the scoring kernels (swizzled 256-B rows,
// lanes 0-31 = rows, lanes 32-63 = the odd 16-B chunk) versus a linear pattern; 4 waves per CU, one per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define ITERS 2000
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out) {
    __shared__ __attribute__((aligned(16))) char lds[96 * 1024];
    for (int i = threadIdx.x; i < 96 * 1024 / 4; i += 256) ((float*)lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    uint32_t addr[8];
    for (int kk = 0; kk < 8; ++kk) {
        if (MODE == 0) addr[kk] = l31 * 256 + (((kk * 2 + half) ^ (l31 & 15)) << 4);      // kernel pattern (D = 128)
        if (MODE == 1) addr[kk] = kk * 1024 + lane * 16;                                  // linear
        if (MODE == 2) addr[kk] = l31 * 256 + ((kk * 2 + half) << 4);                     // unswizzled rows (conflicts)
        if (MODE == 3) addr[kk] = (lane & 15) * 256 + ((((lane >> 4) + kk * 4) ^ (lane & 15)) << 4);  // 16 rows x 4 chunks per instruction
    }
    u4 acc = {0, 0, 0, 0};
    for (int it = 0; it < ITERS; ++it) {
        const int off = (it & 7) * 8192;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            u4 v = *(const u4*)(lds + addr[kk] + off);
            acc += v;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <int MODE> void run(const char* name, float* d) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256, 256>>>(d);
    hipEventRecord(a); k<MODE><<<256, 256>>>(d); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double bytes_per_cu = (double)ITERS * 8 * 4 * 1024;
    printf("%-44s %.3f ms  %.1f B/clk/CU at 2.0 GHz  (%.1f cycles per wave-instruction per CU)\n", name, ms, bytes_per_cu / (ms * 1e-3 * 2.0e9),
           ms * 1e-3 * 2.0e9 / (ITERS * 8 * 4));
}
int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    run<0>("kernel pattern (swizzled rows, half=chunk)", d);
    run<1>("linear", d);
    run<2>("unswizzled rows", d);
    run<3>("16 rows x 4 chunks per instruction", d);
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
}
