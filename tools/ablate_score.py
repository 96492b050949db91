#!/usr/bin/env python
"""Build ablated variants of kvz_score.hip (pass A) into tools/ab/*.so — time attribution experiment, not product code."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "kvzip_amd/csrc/kvz_score.hip")).read()
CS = os.path.join(ROOT, "kvzip_amd/csrc")

def rep(s, old, new):
    assert old in s, old
    return s.replace(old, new)

def nofrag(s):
    return rep(s, "            if (kb + 1 < SC_TILE / 32) load_frags(fr[(kb + 1) & 1], kb + 1);",
               "            if (kb + 1 < SC_TILE / 32) { for (int kk = 0; kk < C::KK; ++kk) fr[(kb + 1) & 1][kk] = fr[kb & 1][kk] + 1u; }")
def nostage(s):
    return rep(s, "        if (linear) stage_tile_linear<D, NWAVES>(dst, kh + (int64_t)(kv0 + off) * C::ROW_BYTES, lane_off, wave);\n        else stage_keys_gather",
               "        if (t > 1) return;\n        if (linear) stage_tile_linear<D, NWAVES>(dst, kh + (int64_t)(kv0 + off) * C::ROW_BYTES, lane_off, wave);\n        else stage_keys_gather")
def noepi(s):
    a = s.index("        uint32_t xp[8];\n        float av[16];\n        const int rel = rows.limit[g]")
    b = s.index("        l_run[g] += ps0 + ps1;  // (masked keys")
    return s[:a] + "        float ps0 = acc[0], ps1 = acc[7];\n" + s[b:]
def nomfma(s):
    return rep(s, "                    if (!skip[g]) acc[g] = Mfma32<T>::mfma(__builtin_bit_cast(v8, fr[kb & 1][kk]), bq[g][kk], acc[g]);",
               "                    if (!skip[g]) { acc[g][2 * kk] += __builtin_bit_cast(float, fr[kb & 1][kk][0]); acc[g][2 * kk + 1] += __builtin_bit_cast(float, fr[kb & 1][kk][1]); }")
def nobarrier(s):
    return rep(s, "        block_barrier();  // ... everybody's has, and nobody reads the current buffer any more", "        // (no barrier)")
variants = {"base": lambda s: s, "nofrag": nofrag, "nostage": nostage, "noepi": noepi, "nomfma": nomfma, "nobarrier": nobarrier,
            "noepi_nomfma": lambda s: nomfma(noepi(s)), "nofrag_nostage": lambda s: nostage(nofrag(s)),
            "compute_only": lambda s: nobarrier(nostage(nofrag(s))), "mfma_only": lambda s: noepi(nobarrier(nostage(nofrag(s)))),
            "epi_only": lambda s: nomfma(nobarrier(nostage(nofrag(s))))}
def trace(s):
    """Light in-kernel timeline of pass A: three s_memtime stamps per tile (tile start, arrival at the hand-over barrier,
    release from it), consumed only at the end of the tile so that the fragment prefetch is not disturbed."""
    s = rep(s, "namespace kvz {\n\ntypedef _Float16 h8", "namespace kvz {\n__device__ unsigned long long g_trace[16 * 8 * 96];\n\ntypedef _Float16 h8")
    s = rep(s, "    Item cur = item_from(0);\n",
            "    const bool trace = (blockIdx.x % 16 == 5 && lane == 0);\n"
            "    unsigned long long* tr = g_trace + ((blockIdx.x / 16) * 8 + wave) * 96;\n"
            "    int tp = 0;\n"
            "    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;\n"
            "    Item cur = item_from(0);\n")
    s = rep(s, "        stage_wait();     // my part of everything in flight (the next tile, the next item's query rows) has landed\n        block_barrier();  // ... everybody's has, and nobody reads the current buffer any more\n",
            "        stage_wait();\n        ts1 = __builtin_amdgcn_s_memtime();\n        block_barrier();\n        ts2 = __builtin_amdgcn_s_memtime();\n")
    s = rep(s, "    while (true) {\n        if ((t * SC_TILE + SC_TILE - 1) > diag0) tile_body(std::true_type{});",
            "    while (true) {\n        ts0 = __builtin_amdgcn_s_memtime();\n        if ((t * SC_TILE + SC_TILE - 1) > diag0) tile_body(std::true_type{});")
    s = rep(s, "        pbuf ^= 1;\n        ahead = ahead_next;\n",
            "        if (trace && tp < 30) { tr[3 * tp] = ts0; tr[3 * tp + 1] = ts1; tr[3 * tp + 2] = ts2; ++tp; }\n        pbuf ^= 1;\n        ahead = ahead_next;\n")
    s += """
extern "C" int kvz_debug_read_trace(unsigned long long* host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(kvz::g_trace), bytes);
}
"""
    return s
variants["trace"] = trace
def gantt(s):
    """Start / end wall-clock stamp (100 MHz), shader-clock ticks, item and tile count of every pass A block."""
    s = rep(s, "namespace kvz {\n\ntypedef _Float16 h8", "namespace kvz {\n__device__ unsigned long long g_trace[4 * 8192];\n\ntypedef _Float16 h8")
    s = rep(s, "    Item cur = item_from(0);\n", "    const int bid = blockIdx.x; int n_items = 0, n_tiles = 0;\n"
            "    if (threadIdx.x == 0) { g_trace[4 * bid] = wall_clock64(); g_trace[4 * bid + 1] = 0; g_trace[4 * bid + 2] = __builtin_amdgcn_s_memtime(); }\n"
            "    Item cur = item_from(0);\n")
    s = rep(s, "        if (!valid(nxt)) break;\n", "        ++n_items; n_tiles += cur.t_hi - cur.t_lo;\n        if (!valid(nxt)) break;\n")
    s = rep(s, "        start_item();\n    }\n}", "        start_item();\n    }\n"
            "    if (threadIdx.x == 0) { g_trace[4 * bid + 1] = wall_clock64(); g_trace[4 * bid + 2] = __builtin_amdgcn_s_memtime() - g_trace[4 * bid + 2]; g_trace[4 * bid + 3] = ((unsigned long long)n_items << 32) | n_tiles; }\n}")
    s += """
extern "C" int kvz_debug_read_trace(unsigned long long* host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(kvz::g_trace), bytes);
}
"""
    return s
variants["gantt"] = gantt
variants["fragafter"] = lambda s: rep(s, """            if (kb + 1 < SC_TILE / 32) load_frags(fr[(kb + 1) & 1], kb + 1);
            else turnover();""", """            __builtin_amdgcn_sched_barrier(0);
            if (kb + 1 < SC_TILE / 32) load_frags(fr[(kb + 1) & 1], kb + 1);
            else turnover();""")
def prio(s):
    """raise the wave priority while the matrix chain is issued (pass A and pass B)"""
    s = rep(s, """            __builtin_amdgcn_sched_barrier(0);
            // the chains of different row groups are independent and are issued ALTERNATELY""", """            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(3);
            // the chains of different row groups are independent and are issued ALTERNATELY""")
    s = rep(s, """            if (kb + 1 < SC_TILE / 32) load_frags(fr[(kb + 1) & 1], kb + 1);
            else turnover();
            __builtin_amdgcn_sched_barrier(0);""", """            __builtin_amdgcn_s_setprio(0);
            if (kb + 1 < SC_TILE / 32) load_frags(fr[(kb + 1) & 1], kb + 1);
            else turnover();
            __builtin_amdgcn_sched_barrier(0);""")
    s = rep(s, """#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) acc = Mfma32<T>::mfma(ak[kk], __builtin_bit_cast(v8, fr[kk]), acc);
            if (kb + 1 < SC_TILE / 32) load_frags(fr, buf, kb + 1);""", """            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) acc = Mfma32<T>::mfma(ak[kk], __builtin_bit_cast(v8, fr[kk]), acc);
            __builtin_amdgcn_s_setprio(0);
            if (kb + 1 < SC_TILE / 32) load_frags(fr, buf, kb + 1);""")
    return s
variants["prio"] = prio
def prio_epi(s):
    """the opposite: raise the priority during the VALU epilogue"""
    s = rep(s, """            else turnover();
            __builtin_amdgcn_sched_barrier(0);
            auto epi = """, """            else turnover();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(3);
            auto epi = """)
    s = rep(s, """            if constexpr (PA_RG > 1) epi(std::integral_constant<int, PA_RG - 1>{});""", """            if constexpr (PA_RG > 1) epi(std::integral_constant<int, PA_RG - 1>{});
            __builtin_amdgcn_s_setprio(0);""")
    return s
variants["prio_epi"] = prio_epi
variants["pb8x4"] = lambda s: rep(s, "#define KVZ_PB_WAVES 4\n#define KVZ_PB_OCC 2", "#define KVZ_PB_WAVES 8\n#define KVZ_PB_OCC 4")
variants["pb4x3"] = lambda s: rep(s, "#define KVZ_PB_WAVES 4\n#define KVZ_PB_OCC 2", "#define KVZ_PB_WAVES 4\n#define KVZ_PB_OCC 3")
variants["pb8x2"] = lambda s: rep(s, "#define KVZ_PB_WAVES 4\n#define KVZ_PB_OCC 2", "#define KVZ_PB_WAVES 8\n#define KVZ_PB_OCC 2")
def split2(s):
    """two independent accumulators per row group (even / odd k-steps), summed before the epilogue"""
    return rep(s, """#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk)
#pragma unroll
                for (int g = 0; g < PA_RG; ++g)
                    if (!skip[g]) acc[g] = Mfma32<T>::mfma(__builtin_bit_cast(v8, fr[kb & 1][kk]), bq[g][kk], acc[g]);""",
               """            f16v acc2[PA_RG];
#pragma unroll
            for (int g = 0; g < PA_RG; ++g)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc2[g][i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < C::KK; kk += 2)
#pragma unroll
                for (int g = 0; g < PA_RG; ++g)
                    if (!skip[g]) {
                        acc[g] = Mfma32<T>::mfma(__builtin_bit_cast(v8, fr[kb & 1][kk]), bq[g][kk], acc[g]);
                        acc2[g] = Mfma32<T>::mfma(__builtin_bit_cast(v8, fr[kb & 1][kk + 1]), bq[g][kk + 1], acc2[g]);
                    }
#pragma unroll
            for (int g = 0; g < PA_RG; ++g) acc[g] += acc2[g];""")
variants["split2"] = split2
variants["w4rg2"] = lambda s: rep(s, "#define KVZ_PA_WAVES 8\n#define KVZ_PA_RG 1", "#define KVZ_PA_WAVES 4\n#define KVZ_PA_RG 2")
variants["ks3"] = lambda s: rep(s, "#define KVZ_KSPLIT_TILES 4", "#define KVZ_KSPLIT_TILES 3")
variants["ks5"] = lambda s: rep(s, "#define KVZ_KSPLIT_TILES 4", "#define KVZ_KSPLIT_TILES 5")
variants["ks6"] = lambda s: rep(s, "#define KVZ_KSPLIT_TILES 4", "#define KVZ_KSPLIT_TILES 6")
variants["ks2"] = lambda s: rep(s, "#define KVZ_KSPLIT_TILES 4", "#define KVZ_KSPLIT_TILES 2")
variants["ks8"] = lambda s: rep(s, "#define KVZ_KSPLIT_TILES 4", "#define KVZ_KSPLIT_TILES 8")
os.makedirs(os.path.join(ROOT, "tools/ab"), exist_ok=True)
objs = [os.path.join(CS, o) for o in ("kvz_api.o", "kvz_select.o", "kvz_compact.o", "kvz_attn.o")]
only = sys.argv[1:]
for name, fn in variants.items():
    if only and name not in only: continue
    src = os.path.join(ROOT, "tools/ab", f"score_{name}.hip")
    open(src, "w").write(fn(SRC))
    obj = src.replace(".hip", ".o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", CS, "-c", src, "-o", obj])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, obj, "-o",
                           os.path.join(ROOT, "tools/ab", f"lib_{name}.so")])
    os.remove(obj)
    print("built", name)
