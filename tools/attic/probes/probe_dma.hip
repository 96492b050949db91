// probe_dma.hip — what does ONE wave pay to issue the four LDS-DMA pieces of a key tile (gfx950)?
// 8 waves per block, one block per CU; every iteration = the staging of one 32-KiB tile in one of several styles, then FILL x 4 fp32
// FMAs (stand-in for the epilogue, long enough to cover the load latency), then vmcnt(0) + barrier.  Reported: s_memtime ticks per iteration minus the no-staging baseline.
//   style 1: per piece  s_mov m0 / s_nop / global_load_lds_dwordx4 v, s[base_i]          (what kvz_score.hip does)
//   style 2: one s_mov m0 per tile, pieces differ by the instruction offset (LDS address AND global address move by it)
//   style 3: global_load_dwordx4 -> VGPR, ds_write_b128
// Style 2 is verified: the tile read back from LDS must equal the source.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lptr_t;
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define TILE 32768
#define ITERS 200

template <int STYLE, int FILL>
__global__ __launch_bounds__(512) void k(const char* src, unsigned long long* ticks, uint32_t* check) {
    __shared__ __attribute__((aligned(16))) char lds[2 * TILE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)lds;
    float f0 = lane, f1 = 1.0f, f2 = 0.5f, f3 = 2.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        const char* tile = src + (size_t)((it * 7 + blockIdx.x) % 32) * TILE;
        const uint32_t buf = lds0 + (it & 1) * TILE;
        if (STYLE == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint64_t b = (uint64_t)(uintptr_t)(tile + (i * 8 + wave) * 1024);
                const uint64_t bs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                                    (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
                const uint32_t la = __builtin_amdgcn_readfirstlane(buf + (i * 8 + wave) * 1024);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"((uint32_t)(lane * 16)), "s"(bs), "s"(la) : "memory");
            }
        } else if (STYLE == 2) {
            // wave w owns the 4 KiB  [w*4096, w*4096 + 4096)  of the tile: one M0, four instruction offsets
            const uint64_t b = (uint64_t)(uintptr_t)(tile + wave * 4096);
            const uint64_t bs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                                (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
            const uint32_t la = __builtin_amdgcn_readfirstlane(buf + wave * 4096);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %0, %1\n\t"
                         "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                         "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                         "global_load_lds_dwordx4 %0, %1 offset:3072" ::"v"((uint32_t)(lane * 16)), "s"(bs), "s"(la) : "memory");
        } else if (STYLE == 3) {
            u4 r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[i]) : "v"(tile + (i * 8 + wave) * 1024 + lane * 16) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i)
                asm volatile("ds_write_b128 %0, %1" ::"v"(buf + (i * 8 + wave) * 1024 + lane * 16), "v"(r[i]) : "memory");
        }
#pragma unroll
        for (int i = 0; i < FILL; ++i) {
            asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(1.0001f), "v"(0.001f));
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
    if (blockIdx.x == 0 && STYLE != 0) {  // the last tile staged: compare with the source
        const int it = ITERS - 1;
        const char* tile = src + (size_t)((it * 7) % 32) * TILE;
        const char* buf = lds + (it & 1) * TILE;
        uint32_t bad = 0;
        for (int o = threadIdx.x * 4; o < TILE; o += 512 * 4) bad += *(const uint32_t*)(buf + o) != *(const uint32_t*)(tile + o);
        atomicAdd(check, bad);
    }
    if (f0 + f1 + f2 + f3 == 12345.f) ticks[0] = 0;
}
template <int STYLE, int FILL> double run(const char* src, unsigned long long* ticks, uint32_t* check, const char* name, double base) {
    hipMemset(check, 0, 4);
    k<STYLE, FILL><<<256, 512>>>(src, ticks, check);
    k<STYLE, FILL><<<256, 512>>>(src, ticks, check);
    hipDeviceSynchronize();
    unsigned long long h[2048]; uint32_t bad;
    hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(&bad, check, 4, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 2048; ++i) s += h[i];
    s /= 2048.0 * ITERS;
    printf("%-44s fill %3d x4 FMA: %8.1f ticks/iteration  (+%.1f over no staging)  mismatches %u\n", name, FILL, s, s - base, bad);
    return s;
}
int main() {
    char* src; unsigned long long* ticks; uint32_t* check;
    hipMalloc(&src, 32 * TILE); hipMalloc(&ticks, 2048 * 8); hipMalloc(&check, 4);
    uint32_t* h = (uint32_t*)malloc(32 * TILE); for (int i = 0; i < 32 * TILE / 4; ++i) h[i] = i * 2654435761u; hipMemcpy(src, h, 32 * TILE, hipMemcpyHostToDevice);
    double b;
    b = run<0, 0>(src, ticks, check, "no staging", 0);
    run<1, 0>(src, ticks, check, "m0 per piece (current)", b);
    run<2, 0>(src, ticks, check, "one m0, instruction offsets", b);
    run<3, 0>(src, ticks, check, "global_load + ds_write", b);
    b = run<0, 100>(src, ticks, check, "no staging", 0);
    run<1, 100>(src, ticks, check, "m0 per piece (current)", b);
    run<2, 100>(src, ticks, check, "one m0, instruction offsets", b);
    run<3, 100>(src, ticks, check, "global_load + ds_write", b);
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
}
