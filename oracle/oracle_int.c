/*
 * oracle_int.c — CPU ORACLE (test infrastructure, never linked into the product path).
 *
 * Plain-C restatement of the INTEGER / BYTE parts of the reference path, used by tests/ as a second,
 * independent checker next to oracle/kvzip_oracle.py and by bench.py's cpu_baseline leg:
 *   orc_threshold          attention/score.py:88-102   (_threshold)
 *   orc_full_mask          attention/kvcache.py:140-150 (_get_valid)
 *   orc_compact            attention/kvcache.py:152-185 (prepare_init, one layer)
 *   orc_update_flatten     csrc/csrc/cuda_api.cu:24-65  (update_flatten_view_kernel offsets)
 *
 * Scores are 16-bit patterns (fp16 / bf16).  Parity: pinned through tests/test_oracle_c.py, which checks
 * every function against the golden vectors generated from the reference's own Python.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* value of a 16-bit float pattern as a C float */
static float half_to_float(uint16_t h, int is_bf16) {
    uint32_t u;
    if (is_bf16) {
        u = (uint32_t)h << 16;
    } else {
        uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu;
        if (exp == 0) {
            if (man == 0) u = sign;
            else {  /* subnormal: normalise */
                int e = -1;
                do { man <<= 1; ++e; } while (!(man & 0x400u));
                u = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
            }
        } else if (exp == 31) u = sign | 0x7F800000u | (man << 13);
        else u = sign | ((exp + 112u) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static int cmp_desc(const void* a, const void* b) {
    float x = *(const float*)a, y = *(const float*)b;
    return (x < y) - (x > y);
}

/* _threshold: sorted descending, n = max((int)(len*ratio) - 1, 0), thres = sorted[n], valid = score > thres.
 * returns 0; *thres_out receives the threshold (0 when ratio >= 1). */
int orc_threshold(const uint16_t* scores, int64_t n, double ratio, int is_bf16, uint8_t* valid, double* thres_out) {
    if (!(ratio < 1.0)) {
        memset(valid, 1, (size_t)n);
        *thres_out = 0.0;
        return 0;
    }
    float* vals = (float*)malloc((size_t)n * sizeof(float));
    if (!vals) return -1;
    for (int64_t i = 0; i < n; ++i) vals[i] = half_to_float(scores[i], is_bf16);
    qsort(vals, (size_t)n, sizeof(float), cmp_desc);
    int64_t idx = (int64_t)((double)n * ratio) - 1;
    if (idx < 0) idx = 0;
    const float thres = vals[idx];
    free(vals);
    for (int64_t i = 0; i < n; ++i) valid[i] = half_to_float(scores[i], is_bf16) > thres;
    *thres_out = (double)thres;
    return 0;
}

/* _get_valid for one head row: ones(sink) ++ valid[0:N] ++ ones(klen - sink - N) */
void orc_full_mask(const uint8_t* valid_row, int sink, int N, int klen, uint8_t* full) {
    for (int p = 0; p < klen; ++p) full[p] = (p < sink || p >= sink + N) ? 1 : (valid_row[p - sink] != 0);
}

/* prepare_init for one layer: k, v [Hkv, klen, row_bytes] -> flat [sum len, row_bytes]; len_k[Hkv], cu_len_k[Hkv+1].
 * returns the number of rows written. */
int64_t orc_compact(const uint8_t* k, const uint8_t* v, const uint8_t* valid /*[Hkv,N]*/, int Hkv, int N, int sink,
                    int klen, int row_bytes, uint8_t* k_out, uint8_t* v_out, int32_t* len_k, int32_t* cu_len_k,
                    int32_t* max_len_k) {
    int64_t rows = 0;
    int mx = 0;
    cu_len_k[0] = 0;
    for (int h = 0; h < Hkv; ++h) {
        int cnt = 0;
        for (int p = 0; p < klen; ++p) {
            const int keep = (p < sink || p >= sink + N) ? 1 : (valid[(int64_t)h * N + (p - sink)] != 0);
            if (!keep) continue;
            memcpy(k_out + rows * row_bytes, k + ((int64_t)h * klen + p) * row_bytes, (size_t)row_bytes);
            memcpy(v_out + rows * row_bytes, v + ((int64_t)h * klen + p) * row_bytes, (size_t)row_bytes);
            ++rows;
            ++cnt;
        }
        len_k[h] = cnt;
        cu_len_k[h + 1] = cu_len_k[h] + cnt;
        if (cnt > mx) mx = cnt;
    }
    *max_len_k = mx;
    return rows;
}

/* update_flatten_view: out = cat_h( cache[cu[h] : cu[h]+headlens[h]], state[h*t : (h+1)*t] ) */
void orc_update_flatten(const uint8_t* cache, const uint8_t* state, const int32_t* headlens, const int32_t* cu_headlens,
                        int Hkv, int t, int row_bytes, uint8_t* out) {
    for (int h = 0; h < Hkv; ++h) {
        const int64_t src = cu_headlens[h];
        const int64_t dst = src + (int64_t)h * t;
        const int64_t ins = (int64_t)cu_headlens[h + 1] + (int64_t)h * t;
        memcpy(out + dst * row_bytes, cache + src * row_bytes, (size_t)headlens[h] * row_bytes);
        memcpy(out + ins * row_bytes, state + (int64_t)h * t * row_bytes, (size_t)t * row_bytes);
    }
}
