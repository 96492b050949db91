// probe_stream.hip - what a split-key decode kernel can pull out of COLD HBM on gfx950 (round 4).
// 28 "layers" of 80 MB (K and V of 157 k keys x 256 B), each read exactly once per pass, one launch per layer, launches back to
// back: the access structure of kvz_attn.hip's split kernel without its arithmetic (the loaded words are xor-reduced).
//   PAT 0: K rows the way the MFMA A operand wants them (lane = key row, 16 B at quad*16 + kk*64: 64-byte pieces of 16 rows per
//          instruction) + V contiguous;   PAT 1: everything contiguous (1 KiB per wave instruction)
//   DEPTH: tiles (32 keys = 16 KiB of K+V per wave) requested ahead of the one being consumed
//   blocks x waves: 176 / 256 x 8 (one block per CU), 512 x 4 and 512 x 8 (two per CU)
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/probe_stream.hip -o tools/probe_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
constexpr int ROWB = 256, KT = 32;

template <int PAT, int DEPTH, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void rd(const char* __restrict__ k, const char* __restrict__ v, int keys, int chunk, uint32_t* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, quad = lane >> 4;
    const int c0 = blockIdx.x * chunk, c1 = min(keys, c0 + chunk);
    u4 kr[DEPTH + 1][8], vr[DEPTH + 1][8];
    auto load = [&](int s, int t0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (PAT == 0) {
                const int key = min(t0 + (i >> 2) * 16 + l15, keys - 1);
                kr[s][i] = *(const u4*)(k + (size_t)key * ROWB + quad * 16 + (i & 3) * 64);
            } else {
                const int c = i * 64 + lane;
                kr[s][i] = *(const u4*)(k + (size_t)min(t0 + c / 16, keys - 1) * ROWB + (c % 16) * 16);
            }
            const int c = i * 64 + lane;
            vr[s][i] = *(const u4*)(v + (size_t)min(t0 + c / 16, keys - 1) * ROWB + (c % 16) * 16);
        }
    };
    uint32_t acc = 0;
    const int step = WAVES * KT;
    int t = c0 + wave * KT;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) if (t + d * step < c1) load(d, t + d * step);
    int slot = 0;
    for (; t < c1; t += step) {
        const int tn = t + DEPTH * step;
        // (ring of DEPTH + 1 register sets, fully unrolled dispatch)
#pragma unroll
        for (int s = 0; s <= DEPTH; ++s)
            if (s == slot) {
                if (tn < c1) load((s + DEPTH) % (DEPTH + 1), tn);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc ^= kr[s][i][0] ^ kr[s][i][3] ^ vr[s][i][1] ^ vr[s][i][2];
            }
        slot = (slot + 1) % (DEPTH + 1);
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

template <int PAT, int DEPTH, int WAVES>
static void run(const char* name, char** K, char** V, int L, int keys, int blocks, uint32_t* out) {
    const int chunk = ((keys + blocks - 1) / blocks + 127) / 128 * 128;
    const int nb = (keys + chunk - 1) / chunk;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int l = 0; l < L; ++l) rd<PAT, DEPTH, WAVES><<<nb, WAVES * 64>>>(K[l], V[l], keys, chunk, out);
    hipEventRecord(a);
    for (int rep = 0; rep < 3; ++rep)
        for (int l = 0; l < L; ++l) rd<PAT, DEPTH, WAVES><<<nb, WAVES * 64>>>(K[l], V[l], keys, chunk, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = 2.0 * keys * ROWB, us = ms * 1e3 / (3 * L);
    printf("%-46s blocks %3d (chunk %4d) x %d waves: %6.1f us per layer = %5.2f TB/s\n", name, nb, chunk, WAVES, us, bytes / us * 1e-6);
}

int main() {
    const int L = 28, keys = 157200;
    char* K[L]; char* V[L]; uint32_t* out;
    for (int l = 0; l < L; ++l) { hipMalloc(&K[l], (size_t)keys * ROWB); hipMalloc(&V[l], (size_t)keys * ROWB); hipMemset(K[l], l + 1, (size_t)keys * ROWB); hipMemset(V[l], l + 3, (size_t)keys * ROWB); }
    hipMalloc(&out, 4096 * 4);
    for (int blocks : {176, 252, 512}) {
        run<0, 1, 8>("MFMA-operand K rows, 1 tile ahead", K, V, L, keys, blocks, out);
        run<1, 1, 8>("contiguous, 1 tile ahead", K, V, L, keys, blocks, out);
        run<0, 2, 8>("MFMA-operand K rows, 2 tiles ahead", K, V, L, keys, blocks, out);
        run<1, 2, 8>("contiguous, 2 tiles ahead", K, V, L, keys, blocks, out);
        run<1, 3, 8>("contiguous, 3 tiles ahead", K, V, L, keys, blocks, out);
    }
    for (int blocks : {252, 512, 1024}) {
        run<0, 1, 4>("MFMA-operand K rows, 1 ahead, 4-wave blocks", K, V, L, keys, blocks, out);
        run<1, 2, 4>("contiguous, 2 ahead, 4-wave blocks", K, V, L, keys, blocks, out);
    }
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
}
