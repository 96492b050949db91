/*
 * kvzip_hip.h — C ABI of the MI355X (gfx950) KV-eviction hot path.
 *
 * This is the drop-in boundary for the path named by BASELINE.json:north_star:
 *   KV importance scoring -> threshold / top-k selection -> KV compaction ->
 *   O(1) append -> variable-length post-prune attention.
 *
 * Every entry point is `extern "C"`, takes raw DEVICE pointers plus sizes, an
 * explicit HIP stream as its last argument, never allocates, never synchronises
 * the device and returns 0 on success or a negative KVZ_E* code (the message is
 * available from kvz_last_error()).  The caller owns every buffer.
 *
 * Each declaration cites the reference interface it replaces
 * (paths are relative to snu-mllab/KVzip @ 2026-03-13).
 */
#ifndef KVZIP_HIP_H
#define KVZIP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KVZ_ABI_VERSION 6

/* element type of K/V/Q/score tensors (reference: csrc/csrc/static_switch.h:3-12) */
#define KVZ_F16 0
#define KVZ_BF16 1

/* error codes */
#define KVZ_OK 0
#define KVZ_EINVAL (-1)   /* bad argument (shape, dtype, alignment, null pointer) */
#define KVZ_EWORKSPACE (-2) /* workspace too small */
#define KVZ_ELAUNCH (-3)  /* HIP launch / runtime error */
#define KVZ_EUNSUPPORTED (-4) /* unsupported head_dim / group size */

typedef void* kvz_stream_t; /* hipStream_t */

int kvz_abi_version(void);
/* thread-local, NUL-terminated description of the last non-zero return */
const char* kvz_last_error(void);

/* Optional measurement hook (no reference counterpart; the reference times with torch.cuda.synchronize() +
 * wall clock, utils/func.py:52-79): when enabled, the dominant kernels are bracketed by hipEvents recorded on
 * the launch stream.  `period` = 0 switches it off, 1 brackets every launch, n brackets every n-th launch of each
 * kernel (an event pair costs ~2.5 us of stream time per bracketed launch).  kvz_prof_read synchronises on the recorded
 * events and returns the accumulated time and the number of BRACKETED launches of kernel `name` ("score_rowstat",
 * "score_colmax", "compact_gather", "select", "varlen_attn"). */
void kvz_prof_enable(int period);
void kvz_prof_reset(void);
int kvz_prof_read(const char* name, double* total_ms, int64_t* count);

/* ------------------------------------------------------------------------- *
 * a1  KV importance scoring          reference: attention/score.py:36-65
 *     (+ causal mask                 reference: attention/score.py:67-85)
 *
 * For one layer and one scoring chunk, for every KV head h:
 *   keys  = K[h, 0:sink] ++ K[h, start:end] ++ K[h, klen-q_len:klen]
 *   A     = half( half(Q_h . keys^T  [fp32 accumulate]) / float(sqrt(D)) )
 *   A[:, :, -q_len:] causally masked (key j of the repeat block visible to query i iff j <= i)
 *   P     = softmax(A, dim=keys)  (fp32 internally, rounded once to half)
 *   out[h, 0:end-start] = max over (group g, query i) of P[g, i, sink:sink+end-start]
 *
 * q   : [Hkv*G, q_len, D]   rows contiguous (D), head stride q_head_stride elements
 * k   : [Hkv, klen, D]      rows contiguous (D), head stride k_head_stride elements
 * out : [Hkv, end-start]    half (same dtype as q/k), head stride out_head_stride elements
 * ws  : kvz_score_workspace_bytes(...) bytes of device scratch.  ONE size for every path the call can take (the knob score_prune and the
 *       shape decide at launch time): partial row statistics and the column slices of the two-pass kernels, plus - always, so that a
 *       buffer sized once serves both - the buffers of the pruned call: the per-group maxima u (Hkv x ceil(m/32) x groups x 128 bytes), n_r
 *       per row, group bounds, the work-item queue and the per-group lists of candidate keys (Hkv x groups x (1 + 32 ceil(m/32)) x 4 bytes).
 *       At the headline shape (Hkv 4, G 7, q 2026, m 2000): 31 MB, of which 29 MB belong to the pruned call.
 * D in {64, 128}.
 * ------------------------------------------------------------------------- */
size_t kvz_score_workspace_bytes(int Hkv, int G, int q_len, int m, int sink);
int kvz_score_chunk(const void* q, int64_t q_head_stride,
                    const void* k, int64_t k_head_stride, int klen,
                    int sink, int start, int end, int q_len,
                    int Hkv, int G, int D, int dtype,
                    void* out, int64_t out_head_stride,
                    void* ws, size_t ws_bytes, kvz_stream_t stream);

/* Asynchronous form of a1 (no reference counterpart: the reference scores on the forward pass's own stream, attention/attn.py:53-54).
 * The scores of a layer are a side product that nothing consumes before prune(), so the call runs on a SIDE stream:
 *   record(ready[slot], caller); wait(side, ready[slot]); kvz_score_chunk(..., side); record(done[slot], side)
 * and kvz_async_wait(handle, slot, stream) orders `stream` behind the scoring calls still pending in `slot` (slot < 0: all).
 * A context is a set of host-side events (no device memory); slots are the caller's (typically one per layer).
 * side == caller degenerates to kvz_score_chunk on that stream.  A handle is driven from ONE host thread at a time (the per-slot
 * state is not locked; creation / destruction of handles is thread-safe). */
int kvz_async_create(int n_slots);   /* -> handle >= 0, or a negative KVZ_E* code */
int kvz_async_destroy(int handle);
int kvz_async_wait(int handle, int slot, kvz_stream_t stream);
int kvz_score_chunk_async(int handle, int slot, kvz_stream_t caller, kvz_stream_t side,
                          const void* q, int64_t q_head_stride,
                          const void* k, int64_t k_head_stride, int klen,
                          int sink, int start, int end, int q_len,
                          int Hkv, int G, int D, int dtype,
                          void* out, int64_t out_head_stride,
                          void* ws, size_t ws_bytes);

/* The same call with the row slices of pass B merged by atomics into a LOG buffer and no finalize launch: `log_out` is
 * [Hkv, log_head_stride] uint32 = bit patterns of the fp32 log-scores  max_r (x - m_r - log l_r)  (<= 0, so the maximum is an
 * unsigned minimum of the patterns), pre-filled with kvz_score_log_fill (-inf).  kvz_score_finalize_log turns a whole buffer
 * (all layers, all chunks) into the 16-bit scores of  attention/score.py:59-63  at once; entries never scored leave `out`
 * untouched.  Same values as kvz_score_chunk, one launch less per (layer, chunk). */
int kvz_score_chunk_log(const void* q, int64_t q_head_stride, const void* k, int64_t k_head_stride, int klen,
                        int sink, int start, int end, int q_len, int Hkv, int G, int D, int dtype,
                        uint32_t* log_out, int64_t log_head_stride, void* ws, size_t ws_bytes, kvz_stream_t stream);
int kvz_score_chunk_async_log(int handle, int slot, kvz_stream_t caller, kvz_stream_t side,
                              const void* q, int64_t q_head_stride, const void* k, int64_t k_head_stride, int klen,
                              int sink, int start, int end, int q_len, int Hkv, int G, int D, int dtype,
                              uint32_t* log_out, int64_t log_head_stride, void* ws, size_t ws_bytes);
/* f2 ("compute the score inside the scoring forward's attention kernel"): the column-maximum pass alone, on row
 * statistics that kvz_flash_fwd_window produced from the forward's own QK^T tiles (attention/attn.py:53-54 duplicates the QK^T of
 * :75-89 in the reference).  stats: [Hkv, stats_head_stride] float2 = (m_r, l'_r), row g*q_len + i.  Output: the log buffer only. */
int kvz_score_from_stats_log(const void* q, int64_t q_head_stride, const void* k, int64_t k_head_stride, int klen,
                             int sink, int start, int end, int q_len, int Hkv, int G, int D, int dtype,
                             const float* stats, int64_t stats_head_stride,
                             uint32_t* log_out, int64_t log_head_stride, kvz_stream_t stream);
int kvz_score_from_stats_async_log(int handle, int slot, kvz_stream_t caller, kvz_stream_t side,
                                   const void* q, int64_t q_head_stride, const void* k, int64_t k_head_stride, int klen,
                                   int sink, int start, int end, int q_len, int Hkv, int G, int D, int dtype,
                                   const float* stats, int64_t stats_head_stride,
                                   uint32_t* log_out, int64_t log_head_stride);
int kvz_score_log_fill(uint32_t* log, int64_t n, kvz_stream_t stream);
int kvz_score_finalize_log(const uint32_t* log, int64_t n, void* out, int dtype, kvz_stream_t stream);
/* kvz_score_finalize_log AND the first pass of kvz_select_threshold in one launch: select_ws (kvz_select_workspace_bytes() bytes) is
 * cleared and receives the histogram of the top 11 bits of the order key of every entry of `out` as it stands afterwards (entries
 * never scored contribute the value `out` already holds).  Follow with kvz_select_threshold_prehist on exactly these n values. */
int kvz_score_finalize_log_hist(const uint32_t* log, int64_t n, void* out, int dtype, void* select_ws, size_t select_ws_bytes,
                                kvz_stream_t stream);

/* One host call for the scoring pass of a layer:  update() of the repeat chunk's K,V into the DENSE cache (attention/kvcache.py:75-78,
 * kvz_dense_append on the CALLER's stream, where the forward's own attention reads the rows next) followed by kvz_score_chunk_async_log
 * on the side stream with k = k_cache, klen = fill + t.  Before the append the caller's stream is ordered behind the previous scoring
 * call of the same slot (it read the rows that are about to be overwritten).  Arguments as in the two calls it replaces.
 * (Round 4 measured the alternative - no append launch, the scoring kernels copy the rows themselves: slower, see
 * profiles/r4_ab_in_kernel_append.txt; the append launch co-resides with the scoring kernels of the side streams and costs nothing.) */
int kvz_update_score_async_log(int handle, int slot, kvz_stream_t caller, kvz_stream_t side,
                               void* k_cache, void* v_cache, int64_t cache_head_stride, int fill,
                               const void* k_state, const void* v_state,
                               int64_t ks_head_stride, int64_t ks_row_stride, int64_t vs_head_stride, int64_t vs_row_stride, int t,
                               const void* q, int64_t q_head_stride, int sink, int start, int end, int q_len,
                               int Hkv, int G, int D, int dtype,
                               uint32_t* log_out, int64_t log_head_stride, void* ws, size_t ws_bytes);

/* Pipelined tail of the asynchronous log calls (knob score_prune = 6; no reference counterpart - the reference's scores are complete when
 * attention/score.py:36-65 returns).  With a workspace of at least THREE times kvz_score_workspace_bytes() the asynchronous log entry
 * points (kvz_score_chunk_async_log, kvz_update_score_async_log) leave the bounds phase of a call to the next call on the same
 * workspace and side stream, and its candidate-key phase to the call after that: one launch behind the row-statistics kernel runs the
 * three phases of three consecutive calls side by side.  The log scores of a call are therefore complete only after two further calls
 * on its workspace - or after kvz_score_tail_flush(ws), which launches what is pending on the stream those calls used (returns 1 when
 * something was launched, 0 when nothing was pending, a negative KVZ_E* code on error).  kvz_score_tail_flush_async does the same and
 * then records the done-event of `slot` on `side`, so that kvz_async_wait(handle, slot | -1, stream) covers the flushed phases.
 * Until then the caller keeps q, k and the workspace of the pending calls alive and unchanged (the ctx rows of k and all of q are
 * read again by the candidate-key phase).  Smaller workspaces, other knob values and the synchronous entry points never defer. */
int kvz_score_tail_flush(const void* ws);
int kvz_score_tail_flush_async(int handle, int slot, const void* ws, kvz_stream_t side);

/* ------------------------------------------------------------------------- *
 * a4  global-threshold selection     reference: attention/score.py:88-102
 *
 *   flat   = scores viewed as n values (any leading shape)
 *   idx    = max((int64)((double)n * ratio) - 1, 0)
 *   thres  = idx-th largest value (0-based, duplicates counted)
 *   valid  = scores > thres            (strict: ties at thres are evicted)
 *   ratio >= 1  ->  valid all ones, thres = 0
 *
 * scores       : n half values
 * valid_out    : n bytes (0/1), same flat order
 * row_counts   : optional int32[n / row_len]  number of valid entries per row of row_len
 * thres_dev    : float[1] on device;  kept_dev: int64[1] on device (number of valid entries)
 * ws           : kvz_select_workspace_bytes() bytes
 * ------------------------------------------------------------------------- */
size_t kvz_select_workspace_bytes(void);
int kvz_select_threshold(const void* scores, int64_t n, double ratio, int dtype,
                         uint8_t* valid_out,
                         int64_t row_len, int32_t* row_counts,
                         float* thres_dev, int64_t* kept_dev,
                         void* ws, size_t ws_bytes, kvz_stream_t stream);
/* The same selection when `ws` already holds the first histogram of exactly these scores (kvz_score_finalize_log_hist): two launches
 * (low-bits histogram, mask) instead of three.  Same results, bit for bit.  A workspace whose histogram does NOT belong to these
 * scores (the wanted rank lies beyond its total) gives *thres_dev = NaN and an all-false mask - never a threshold read from
 * uninitialised memory; a caller that sees NaN without NaN scores falls back to kvz_select_threshold (EvictCache._select does). */
int kvz_select_threshold_prehist(const void* scores, int64_t n, double ratio, int dtype,
                                 uint8_t* valid_out,
                                 int64_t row_len, int32_t* row_counts,
                                 float* thres_dev, int64_t* kept_dev,
                                 void* ws, size_t ws_bytes, kvz_stream_t stream);

/* ------------------------------------------------------------------------- *
 * a5  per-(layer,head) top-k         reference: attention/score.py:104-120
 *
 *   for every row of row_len scores keep exactly k = (int64)((double)row_len*ratio)
 *   entries: all values greater than the k-th largest, plus the lowest-index
 *   entries equal to it until k are kept (torch.topk's tie order is unspecified;
 *   on tie-free rows the result is identical).
 * ------------------------------------------------------------------------- */
int kvz_select_topk_rows(const void* scores, int64_t rows, int64_t row_len, int64_t k,
                         int dtype, uint8_t* valid_out, int32_t* row_counts,
                         kvz_stream_t stream);

/* ------------------------------------------------------------------------- *
 * a17  head-level selection          reference: model/wrapper.py:40-58 feeding attention/score.py:88-102
 *
 * The reference expands the [L, Hkv] head scores to [L, 1, Hkv, N] and runs the global threshold on L*Hkv*N values.
 * Every head value appears N times, so  thres = head value of rank  max((int64)((double)(rows*N)*ratio) - 1, 0) / N
 * (descending, duplicates counted)  and  valid_heads[r] = head_scores[r] > thres.  Nothing of size N is touched.
 *   head_scores : rows = L*Hkv half values      valid_heads : rows bytes (0/1)
 *   row_counts  : optional int32[rows] = N for a kept head, 0 otherwise
 *   kept_dev    : int64[1] = N * number of kept heads (what the expanded mask would sum to)
 *   ratio >= 1  -> all ones, thres = 0
 * ------------------------------------------------------------------------- */
int kvz_select_heads(const void* head_scores, int rows, int64_t N, double ratio, int dtype,
                     uint8_t* valid_heads, int32_t* row_counts, float* thres_dev, int64_t* kept_dev,
                     kvz_stream_t stream);

/* f4  head-score production          reference: test.py:22-25  (torch.stack(kv.score).squeeze().amax(-1))
 *   out[r] = max over the row_len scores of row r (16-bit patterns, compared as numbers)  */
int kvz_rowmax16(const void* scores, int64_t rows, int64_t row_len, int dtype, void* out, kvz_stream_t stream);

/* ------------------------------------------------------------------------- *
 * a8+a9  compaction plan + gather    reference: attention/kvcache.py:140-185
 *
 * Plan (all layers at once).  For row r = layer*Hkv + h:
 *   full mask   = ones(sink) ++ valid[r, 0:N] ++ ones(klen - sink - N)
 *   len_k[r]    = popcount(full mask)
 *   seg_start[layer, h] = sum_{h'<h} (len_k[layer,h'] + slack)      (slack = 0 -> packed,
 *                          identical to the reference's cu_len_k[:-1])
 *   cu_len_k[layer, 0:Hkv+1] = [0, cumsum(len_k[layer])]            (reference layout)
 *   max_len_k[layer]   = max_h len_k[layer,h]
 *   tile_base[r, t]    = number of kept tokens of row r before token tile t (tile = KVZ_COMPACT_TILE)
 * ------------------------------------------------------------------------- */
#define KVZ_COMPACT_TILE 1024
size_t kvz_compact_plan_bytes(int layers, int Hkv, int klen); /* bytes for tile_base */
int kvz_compact_plan(const uint8_t* valid, int layers, int Hkv, int N, int sink, int klen,
                     int slack,
                     int32_t* len_k, int32_t* cu_len_k, int32_t* seg_start, int32_t* max_len_k,
                     int32_t* tile_base, kvz_stream_t stream);

/* Gather for one layer: order-preserving, head-major.
 *   k_out[seg_start[h] + j, :] = k[h, src_j, :]   for the j-th kept token of head h (same for v)
 * k, v       : [Hkv, klen, D] rows contiguous, head stride in_head_stride elements
 * valid      : this layer's [Hkv, N] bytes;  tile_base: this layer's [Hkv, ntiles]
 * k_out,v_out: [total_rows, D]
 */
int kvz_compact_layer(const void* k, const void* v, int64_t in_head_stride,
                      const uint8_t* valid, const int32_t* tile_base, const int32_t* seg_start,
                      int Hkv, int N, int sink, int klen, int D, int elem_bytes,
                      void* k_out, void* v_out, kvz_stream_t stream);

/* Batched gather over all layers in ONE launch.  k_ptrs/v_ptrs/k_out_ptrs/v_out_ptrs are DEVICE
 * arrays of `layers` pointers; valid/tile_base/seg_start are the full plan arrays. */
int kvz_compact_layers(const void* const* k_ptrs, const void* const* v_ptrs, int64_t in_head_stride,
                       const uint8_t* valid, const int32_t* tile_base, const int32_t* seg_start,
                       int layers, int Hkv, int N, int sink, int klen, int D, int elem_bytes,
                       void* const* k_out_ptrs, void* const* v_out_ptrs, kvz_stream_t stream);

/* Head-level variants of the plan and the batched gather (a17): `valid_heads` holds ONE byte per (layer, head); the full
 * mask of a row is  ones(sink) ++ (valid_heads[r] ? ones(N) : zeros(N)) ++ ones(klen - sink - N),  i.e. a kept head moves
 * all of its rows and a dropped head only its sink rows.  Same outputs and layout as kvz_compact_plan / kvz_compact_layers. */
int kvz_compact_plan_heads(const uint8_t* valid_heads, int layers, int Hkv, int N, int sink, int klen,
                           int slack,
                           int32_t* len_k, int32_t* cu_len_k, int32_t* seg_start, int32_t* max_len_k,
                           int32_t* tile_base, kvz_stream_t stream);
int kvz_compact_layers_heads(const void* const* k_ptrs, const void* const* v_ptrs, int64_t in_head_stride,
                             const uint8_t* valid_heads, const int32_t* tile_base, const int32_t* seg_start,
                             int layers, int Hkv, int N, int sink, int klen, int D, int elem_bytes,
                             void* const* k_out_ptrs, void* const* v_out_ptrs, kvz_stream_t stream);

/* ------------------------------------------------------------------------- *
 * a11  update_flatten_view           reference: csrc/csrc/cuda_api.cu:15-111
 *
 * Reference-exact out-of-place rebuild:
 *   out = cat_h( cache[cu_headlens[h] : cu_headlens[h]+headlens[h]] , state[h*t:(h+1)*t] )
 * cache [sum, D], state [Hkv*t, D], out [sum_h headlens[h] + Hkv*t, D]
 * ------------------------------------------------------------------------- */
int kvz_update_flatten_view(const void* cache, const void* state,
                            const int32_t* headlens, const int32_t* cu_headlens,
                            int Hkv, int t, int D, int elem_bytes,
                            void* out, kvz_stream_t stream);

/* a10 (MI355X layout)  O(t) in-place append into per-head slack:
 *   cache[seg_start[h] + base_len[h] + len_offset + i, :] = state[h, i, :]    i in [0,t)
 * base_len is the device-resident len_k of the pruned cache; len_offset is the reference's host-side
 * info["offset"][layer] (attention/kvcache.py:58), so no device-side bookkeeping is needed per token.
 * k_state / v_state are [Hkv, t, D] views with arbitrary head and row strides (elements; D contiguous), so K after
 * RoPE and V straight out of the projection can be appended without a copy.  Both K and V in one launch. */
int kvz_append_inplace(void* k_cache, void* v_cache,
                       const void* k_state, const void* v_state,
                       int64_t k_head_stride, int64_t k_row_stride, int64_t v_head_stride, int64_t v_row_stride,
                       const int32_t* seg_start, const int32_t* base_len, int len_offset,
                       int Hkv, int t, int D, int elem_bytes, kvz_stream_t stream);

/* a10 before pruning (reference attention/kvcache.py:75-78: torch.cat along the sequence): append t rows per head to the DENSE
 * cache in place.  k_cache / v_cache are [Hkv, capacity, D] with head stride cache_head_stride elements; the new rows go to
 * rows fill .. fill+t of every head.  k_state / v_state as in kvz_append_inplace.  No device-resident metadata. */
int kvz_dense_append(void* k_cache, void* v_cache, int64_t cache_head_stride, int fill,
                     const void* k_state, const void* v_state,
                     int64_t k_head_stride, int64_t k_row_stride, int64_t v_head_stride, int64_t v_row_stride,
                     int Hkv, int t, int D, int elem_bytes, kvz_stream_t stream);

/* ------------------------------------------------------------------------- *
 * a13  variable-length attention     reference call site: attention/attn.py:56-73
 *      (flash_attn_varlen_func, flash-attn 2.7.4.post1, un-vendored third party)
 *
 * Every KV head is one ragged "sequence"; its G query heads are MQA heads.
 *   q   : [Hkv*q_len, G, D]   (row = h*q_len + i)
 *   k,v : [rows, D]; head h owns rows k_start[h] .. k_start[h]+len_h,  len_h = k_len[h] + k_len_offset
 *   out : [Hkv*q_len, G, D]
 *   causal (bottom-right aligned): query i sees key j iff j <= i + (len_h - q_len)
 *   P = softmax(q.k^T * scale) in fp32, out = P.v rounded to half
 * ------------------------------------------------------------------------- */
size_t kvz_varlen_attn_workspace_bytes(int Hkv, int G, int q_len, int D, int max_len_k);
/* k_meta_host (optional, HOST pointer, may be NULL): [k_start[0..Hkv), k_len[0..Hkv)] as the caller knows them on the host (the
 * cache object does: one D2H copy per prune).  With it the kernel gets the head segments as launch arguments and does not start
 * with a dependent load of the device arrays.  Up to 64 heads; ignored beyond.
 * ws: kvz_varlen_attn_workspace_bytes(...) bytes of scratch (partial results of the key ranges, or of the key splits of the
 * multi-row kernel); no initialisation needed, the size does not depend on max_len_k.
 * k_len_offset_dev (optional, device int32): added to k_len_offset INSIDE the kernels.  A generation step captured in a HIP graph
 * is replayed with unchanged arguments, so the part of the appended-token count that changes from token to token lives on the
 * device (kvz_add_i32 advances it as the last node of the step); decode calls only (q_len*G <= 64 rows). */
int kvz_varlen_attn(const void* q, const void* k, const void* v,
                    const int32_t* k_start, const int32_t* k_len, int k_len_offset, const int32_t* k_len_offset_dev,
                    const int32_t* k_meta_host,
                    int Hkv, int G, int q_len, int D, int max_len_k,
                    float scale, int causal, int dtype,
                    void* out, void* ws, size_t ws_bytes, kvz_stream_t stream);

/* a10 + a13 fused (decode step, one new token):  kvz_append_inplace(t = 1) followed by kvz_varlen_attn(q_len = 1) in ONE
 * launch.  Replaces the pair
 *   past_key_value.update(...)   reference attention/attn.py:44-48  -> csrc/csrc/cuda_api.cu:68-111
 *   flash_attn_varlen_func(...)  reference attention/attn.py:61-71
 * for the generation loop.  k_state / v_state: [Hkv, 1, D] views (head stride in elements, D contiguous).  The row is
 * written at  k_start[h] + k_len[h] + k_len_offset  (k_len_offset = tokens appended before this call) and the attention
 * runs over k_len[h] + k_len_offset + 1 keys; max_len_k must cover that.  Results are bit-identical to the two-call
 * sequence. */
int kvz_varlen_attn_append(const void* q, void* k_cache, void* v_cache,
                           const void* k_state, const void* v_state,
                           int64_t k_state_head_stride, int64_t v_state_head_stride,
                           const int32_t* k_start, const int32_t* k_len, int k_len_offset, const int32_t* k_len_offset_dev,
                           const int32_t* k_meta_host,
                           int Hkv, int G, int D, int max_len_k, float scale, int dtype,
                           void* out, void* ws, size_t ws_bytes, kvz_stream_t stream);

/* The dense causal forward of a SCORING pass (reference attention/attn.py:75-89 with kv.get_score set): kvz_flash_fwd on the
 * 32-row kernel, which additionally applies the scoring rounding chain (attention/score.py:57-61) to the accumulators of the key
 * tiles that belong to  sink ++ [win_start, win_end) ++ repeat chunk  and writes the per-row softmax statistics
 * win_stats [Hkv, win_stats_head_stride] float2 (m_r, l'_r; row g*q_len + i) for kvz_score_from_stats_log.  Head segments by
 * value (k_meta_host: starts then lengths, the lengths include the q_len rows of the repeat chunk).  KVZ_EUNSUPPORTED when the
 * shape is not one the 32-row kernel takes (head_dim 128, enough row blocks): the caller then scores with kvz_score_chunk. */
int kvz_flash_fwd_window(const void* q, int64_t q_stride_head, int64_t q_stride_group, int64_t q_stride_pos,
                         const void* k, const void* v, const int32_t* k_meta_host,
                         int Hkv, int G, int q_len, int D, float scale, int dtype,
                         void* out, int64_t o_stride_head, int64_t o_stride_group, int64_t o_stride_pos,
                         int win_sink, int win_start, int win_end, float* win_stats, int64_t win_stats_head_stride,
                         kvz_stream_t stream);

/* *p += delta on the stream (the device-side token counter of a captured generation step, see kvz_varlen_attn). */
int kvz_add_i32(int32_t* p, int delta, kvz_stream_t stream);

/* f2 + a13 with q_len > 1: causal GQA attention of R = q_len*G query rows per KV head over that head's key segment, keys
 * walked once per 128-row block (LDS-shared 64-key tiles, MFMA 16x16x32, online softmax).  Replaces
 *   flash_attn_func(query, key, value, causal=True)            reference attention/attn.py:75-89 (dense pre-prune forward;
 *                                                              head h owns rows h*capacity .. of the dense cache), and
 *   flash_attn_varlen_func(..., max_seqlen_q = q_len > 1)      reference attention/attn.py:61-71 (first generation step on a
 *                                                              pruned cache; kvz_varlen_attn forwards here when q_len*G > 16).
 * Element (h, g, i) of the query sits at  q + h*q_stride_head + g*q_stride_group + i*q_stride_pos  (same for out), so both
 * [Hkv*q_len, G, D] and [H, q_len, D] are addressed without a re-layout copy; D contiguous.  k, v: [rows, D]; head h owns rows
 * k_start[h] .. +k_len[h]+k_len_offset (device arrays, or k_meta_host = {start[Hkv], len[Hkv]} by value, up to 64 heads; either
 * may be NULL if the other is given).  Causal mask aligned bottom-right: position i sees keys j <= i + len_h - q_len.
 * lse_out: NULL or float [Hkv, q_len*G] (row i*G+g): natural-log LSE of the scaled logits, -inf for rows that see no key.
 * ws: NULL, or kvz_flash_workspace_bytes(...) bytes (no initialisation needed).  The keys of a head are split over several blocks
 * whose partial results a second launch merges whenever whole (head, row tile) blocks would leave CUs idle - few query rows, or a
 * number of row tiles that does not fill the last round of blocks; that needs the workspace (up to 68 MB at head_dim 128).
 * Without one every block walks all keys of its head (correct, slower in those cases). */
size_t kvz_flash_workspace_bytes(int Hkv, int G, int q_len, int D);
int kvz_flash_fwd(const void* q, int64_t q_stride_head, int64_t q_stride_group, int64_t q_stride_pos,
                  const void* k, const void* v,
                  const int32_t* k_start, const int32_t* k_len, int k_len_offset, const int32_t* k_meta_host,
                  int Hkv, int G, int q_len, int D, float scale, int causal, int dtype,
                  void* out, int64_t o_stride_head, int64_t o_stride_group, int64_t o_stride_pos,
                  float* lse_out, void* ws, size_t ws_bytes, kvz_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* KVZIP_HIP_H */
