#!/usr/bin/env python
"""Build ablated variants of kvz_score.hip (pass A) into tools/ab/*.so — time attribution experiment, not product code."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "kvzip_amd/csrc/kvz_score.hip")).read()
CS = os.path.join(ROOT, "kvzip_amd/csrc")

def rep(s, old, new):
    assert old in s, old
    return s.replace(old, new)

EXP0 = "ps0 += __builtin_amdgcn_exp2f(__builtin_fmaf(pair_lo<T>(xp[p]), L2E, -ml2_run));"
EXP1 = "ps1 += __builtin_amdgcn_exp2f(__builtin_fmaf(pair_hi<T>(xp[p]), L2E, -ml2_run));"
variants = {
    "base": lambda s: s,
    "noexp": lambda s: rep(rep(s, EXP0, "ps0 += __builtin_fmaf(pair_lo<T>(xp[p]), L2E, -ml2_run);"), EXP1,
                           "ps1 += __builtin_fmaf(pair_hi<T>(xp[p]), L2E, -ml2_run);"),
    "noexpfma": lambda s: rep(rep(s, EXP0, "ps0 += __builtin_bit_cast(float, xp[p]);"), EXP1, ""),
    "nomfma": lambda s: rep(s, "for (int kk = 0; kk < C::KK; ++kk) acc = Mfma32<T>::mfma(__builtin_bit_cast(v8, fr[kk]), bq[kk], acc);",
                            "for (int kk = 0; kk < C::KK; ++kk) { acc[2 * kk] += __builtin_bit_cast(float, fr[kk][0]); acc[2 * kk + 1] += __builtin_bit_cast(float, fr[kk][1]); }"),
    "nostage": lambda s: rep(s, "if (t + 1 < t_hi) stage(next_buf, t + 1);", ""),
    "nochainasm": lambda s: rep(s, 'asm(KVZ_MIX_ALL\n                "v_pk_max_f16 %[m0], %0, %1', 'asm("s_nop 0\\n\\t"\n                "v_pk_max_f16 %[m0], %0, %1'),
}
variants["fixedtile"] = lambda s: rep(s, "if (t + 1 < t_hi) stage(next_buf, t + 1);", "if (t + 1 < t_hi) stage(next_buf, t_lo);")
variants["noepi_nomfma"] = lambda s: variants["nomfma"](variants["nochainasm"](variants["noexpfma"](s)))
variants["noepi_fixedtile"] = lambda s: variants["fixedtile"](variants["nochainasm"](variants["noexpfma"](s)))
def trace(s):
    """In-kernel timeline of persistent pass A: s_memtime stamps (lane 0 of every wave of a few blocks).
    Per tile 13 stamps: [after-mfma-issue, after-epilogue] x 3, then for the 4th block: after-mfma, after stage_wait,
    after barrier, after stage issue, after frag prefetch issue, after epilogue."""
    s = rep(s, "namespace kvz {\n\ntypedef _Float16 h8", "namespace kvz {\n__device__ unsigned long long g_trace[16 * 4 * 160];\n\ntypedef _Float16 h8")
    s = rep(s, "    // ---- first item: static",
            "    const bool trace = (blockIdx.x % 32 == 5 && lane == 0);\n"
            "    unsigned long long* tr = g_trace + ((blockIdx.x / 32) * 4 + wave) * 160;\n"
            "    int tp = 0;\n"
            "#define STAMP() do { if (trace && tp < 158) tr[2 + tp++] = __builtin_amdgcn_s_memtime(); } while (0)\n"
            "    if (trace) { tr[0] = 0; tr[1] = wall_clock64(); }\n"
            "    STAMP();\n"
            "    // ---- first item: static")
    s = rep(s, "                if (kb + 1 < SC_TILE / 32) load_frags(fr, cur_index, kb + 1);\n                else turnover(cur, cur_index ^ 1, t);\n",
            "                STAMP();\n                if (kb + 1 < SC_TILE / 32) load_frags(fr, cur_index, kb + 1);\n                else turnover(cur, cur_index ^ 1, t);\n")
    s = rep(s, "            stage_wait();     // my part of the tile in flight has landed\n            __syncthreads();  // ... everybody's has, and nobody reads `cur` any more\n",
            "            stage_wait(); STAMP();\n            __syncthreads(); STAMP();\n")
    s = rep(s, "                if (t + 2 < t_hi) stage(cur, kh, t + 2);\n                load_frags(fr, nxt_index, 0);\n",
            "                if (t + 2 < t_hi) stage(cur, kh, t + 2);\n                STAMP();\n                load_frags(fr, nxt_index, 0);\n                STAMP();\n")
    s = rep(s, "                    else epilogue(acc, k0, std::false_type{});\n                }\n            }\n        };",
            "                    else epilogue(acc, k0, std::false_type{});\n                }\n                STAMP();\n            }\n        };")
    s += """
extern "C" int kvz_debug_read_trace(unsigned long long* host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(kvz::g_trace), bytes);
}
"""
    return s
variants["trace"] = trace
def gantt(s):
    """Start / end wall-clock stamp (100 MHz), HW id and item count of every persistent pass A block."""
    s = rep(s, "namespace kvz {\n\ntypedef _Float16 h8", "namespace kvz {\n__device__ unsigned long long g_trace[4 * 8192];\n\ntypedef _Float16 h8")
    s = rep(s, "    // ---- first item ----\n", "    const int bid = blockIdx.x; int n_items = 0, n_tiles = 0;\n"
            "    if (threadIdx.x == 0) { g_trace[4 * bid] = wall_clock64(); g_trace[4 * bid + 1] = 0; g_trace[4 * bid + 2] = __builtin_amdgcn_s_memtime(); }\n"
            "    // ---- first item ----\n")
    s = rep(s, "        if (next_id >= nitems) break;\n        it = nit;", "        ++n_items; n_tiles += t_hi - t_lo;\n        if (next_id >= nitems) break;\n        it = nit;")
    s = rep(s, "        for (int kk = 0; kk < C::KK; ++kk) bq[kk] = bq_next[kk];\n    }\n}",
            "        for (int kk = 0; kk < C::KK; ++kk) bq[kk] = bq_next[kk];\n    }\n"
            "    if (threadIdx.x == 0) { g_trace[4 * bid + 1] = wall_clock64(); g_trace[4 * bid + 2] = __builtin_amdgcn_s_memtime() - g_trace[4 * bid + 2]; g_trace[4 * bid + 3] = ((unsigned long long)n_items << 32) | n_tiles; }\n}")
    s += """
extern "C" int kvz_debug_read_trace(unsigned long long* host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(kvz::g_trace), bytes);
}
"""
    return s
variants["gantt"] = gantt
W = 'asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); '
variants["w_turn"] = lambda s: rep(s, "                load_frags(fr, nxt_index, 0);", "                " + W + "load_frags(fr, nxt_index, 0);")
variants["w_start"] = lambda s: rep(s, "        load_frags(fr, 0, 0);", "        " + W + "load_frags(fr, 0, 0);")
variants["w_both"] = lambda s: variants["w_turn"](variants["w_start"](s))
variants["w_kb"] = lambda s: rep(s, "                if (kb + 1 < SC_TILE / 32) load_frags(fr, cur_index, kb + 1);", "                if (kb + 1 < SC_TILE / 32) { " + W + "load_frags(fr, cur_index, kb + 1); }")
def cfrags(s):
    """compiler-visible fragment reads (the LDS-DMA alias wait comes back) - race bisection"""
    a = s.index("template <int D>\n__device__ static inline void frag_load(")
    b = s.index("template <int KK>\n__device__ static inline void frag_wait(")
    new = """template <int D>
__device__ static inline void frag_load(u32x4 (&fr)[D / 16], const FragAddr<D>& fa, int byte_off) {
    typedef const __attribute__((address_space(3))) u32x4* lp;
#pragma unroll
    for (int kk = 0; kk < D / 16; ++kk) fr[kk] = *(lp)(uintptr_t)(fa.a[kk] + byte_off);
}
"""
    return s[:a] + new + s[b:]
variants["cfrags"] = cfrags
variants["qturn"] = lambda s: rep(rep(s, "                stage(buf0, head_keys(nit.h), nit.t_lo);\n", "                stage(buf0, head_keys(nit.h), nit.t_lo);\n                nrow = row_of(nit);\n                load_q(bq_next, nrow.qp);\n"),
                                  "        if (next_id < nitems) {\n            nrow = row_of(nit);\n            load_q(bq_next, nrow.qp);\n        }\n", "")
variants["ks2"] = lambda s: rep(s, "#define KVZ_KSPLIT_TILES 8", "#define KVZ_KSPLIT_TILES 2")
variants["ks4"] = lambda s: rep(s, "#define KVZ_KSPLIT_TILES 8", "#define KVZ_KSPLIT_TILES 4")
variants["noepi"] = lambda s: variants["nochainasm"](variants["noexpfma"](s))
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
