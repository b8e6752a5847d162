/*
 * asd_hip.h -- C ABI of libasd_hip.so: the MI355X (gfx950) draft-verify / accept /
 * optimal-stopping hot path behind the reference's Python API.
 *
 * The reference (sa2shun/adaptive-speculative-decoding) is 100 % Python and has no FFI of its
 * own, so every entry point below cites the reference *Python* symbol whose
 * arithmetic it replaces.  The reference-side binding a maintainer would add is a ctypes stub;
 * it is shown in INTEGRATION.md and shipped as asd_amd/_binding.py.
 *
 * Conventions
 *   - every function returns an asd_status (0 = ok, negative = error); nothing throws, nothing
 *     allocates device memory, nothing synchronises the device or the stream;
 *   - every pointer except the ones marked "host" is a DEVICE pointer on the current HIP
 *     device; `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - there is no CPU mode: the library only launches gfx950 kernels.  The CPU restatement
 *     lives in oracle/ and is test infrastructure only.
 *   - rows: a "row" is one verified position r = b*K + k of sequence b, draft position k.
 */
#ifndef ASD_HIP_H
#define ASD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASD_VERSION_MAJOR 0
#define ASD_VERSION_MINOR 3
#define ASD_VERSION_PATCH 0

typedef enum asd_status {
    ASD_OK = 0,
    ASD_ERR_INVALID_ARG = -1, /* NULL pointer, negative size, length mismatch (reference: ValueError, dp_solver.py:34-35) */
    ASD_ERR_UNSUPPORTED = -2, /* K > 64, L > 16, unknown dtype, ... */
    ASD_ERR_WORKSPACE = -3,   /* workspace too small or misaligned */
    ASD_ERR_HIP = -4,         /* a HIP runtime call failed (launch error, no device) */
    ASD_ERR_ALIGNMENT = -5    /* an operand pointer violates its element alignment */
} asd_status;

typedef enum asd_dtype {
    ASD_DTYPE_F32 = 0,
    ASD_DTYPE_BF16 = 1,
    ASD_DTYPE_F16 = 2
} asd_dtype;

/* limits enforced by the launchers */
#define ASD_MAX_DRAFT_LEN 64   /* K: one ballot word per sequence */
#define ASD_MAX_STAGES 16      /* L: tiers in the DP rule */
#define ASD_MAX_SPLITS 64      /* vocab splits per row inside one launch */
#define ASD_MAX_MLP_DIM 1024   /* predictor input / hidden width */
#define ASD_NUM_LP_STATS 5     /* mean, std, min, q25, median */

/* version = major*10000 + minor*100 + patch */
int asd_version(void);
/* host: static string for a status code */
const char* asd_status_string(int status);
/* host: number of compute units of HIP device `device` (cached); <0 = asd_status */
int asd_device_cu_count(int device);

/* ------------------------------------------------------------------------------------------
 * A5 / A6  token-level verify + accept.
 *
 * Replaces the per-token loop `probs = F.softmax(score[0]); logprob = torch.log(probs[token_id])`
 * of src/training/generate_training_data.py:128-136 (one token at a time, 3 launches + a D2H
 * sync each) with ONE streaming pass over [B,K,V], and adds the speculative-sampling test the
 * north star names (no reference symbol exists for it):
 *
 *   lse[b,k]    = log sum_v exp(logits[b,k,v])                (logits / T with asd_verify_accept_ex)
 *   lp_t[b,k]   = logits[b,k,tok[b,k]] - lse[b,k]          (tok outside [0,V) => -inf)
 *   accept[b,k] = log(u[b,k]) <= lp_t[b,k] - lp_d[b,k]     (== u <= min(1, p_t/p_d))
 *   bits[b]     = sum_k accept[b,k] << k
 *   n_acc[b]    = number of leading 1 bits of bits[b] (length of the accepted prefix)
 *
 * logits: [B*K rows][V] of `dtype`, consecutive rows `ld_row` ELEMENTS apart (ld_row >= V).
 * workspace: >= asd_verify_accept_workspace_bytes(B,K,V,dtype) bytes, 256-byte aligned, and
 * zero-initialised ONCE with asd_workspace_init before its first use; the kernel leaves it
 * ready for the next call (tickets are reset by the last arriver), so one workspace serves any
 * number of stream-ordered calls.  Two calls that may run CONCURRENTLY need two workspaces.
 * ---------------------------------------------------------------------------------------- */
size_t asd_verify_accept_workspace_bytes(int B, int K, int V, int dtype);
int asd_workspace_init(void* workspace, size_t workspace_bytes, void* stream);

/* Sticky status of a workspace (verify, residual-sample and draft-sample workspaces alike): its FIRST 32-bit word.  The
 * kernels hand results across workgroups through self-tagging words of the workspace and wait for them with a bounded poll;
 * a word that never arrives poisons what depended on it (lp_target = NaN and reject; score = NaN, k* = L - 1, stop = 0;
 * tok = -1, lp = NaN) -- never a silently wrong value -- and or-s ASD_WS_LOST_HANDOFF into this word.  The word is cleared
 * only by asd_workspace_init.  A workspace whose status is non-zero MUST be re-initialised before its next use: the late
 * word was never handed back empty, so the workspace is no longer all-zero between calls.  (The reference's error
 * convention on this path: log, count, re-raise -- src/serving/pipeline.py:133-136; SURVEY.md section 8b: "return 0 on
 * success, negative asd_status on error; never throw".  A launch cannot return what its kernel finds out later: the status
 * word is that return channel.)
 * asd_workspace_status copies the word to *status_host and WAITS for `stream` (the one synchronising call of this library);
 * a caller that synchronises anyway may instead read the word from the buffer itself. */
#define ASD_WS_LOST_HANDOFF 0x1u
int asd_workspace_status(const void* workspace, uint32_t* status_host /*host, out*/, void* stream);

int asd_verify_accept(const void* logits, int dtype, int64_t ld_row,
                      const int32_t* tok /*[B,K]*/, const float* lp_draft /*[B,K]*/,
                      const float* u /*[B,K]*/, int B, int K, int V,
                      float* lp_target /*[B,K] out*/, uint8_t* accept /*[B,K] out*/,
                      int32_t* n_acc /*[B] out*/, uint64_t* accept_bits /*[B] out, may be NULL*/,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Options of asd_verify_accept_ex (host struct; NULL = all defaults). */
typedef struct asd_verify_options {
    float inv_temperature; /* > 0.  Logits are multiplied by it INSIDE the streaming pass (it only changes the
                              per-element FMA constant), i.e. the test is run on softmax(logits / T) without a
                              separate scaling pass over [B,K,V].  1.0f = logits as given.  The reference samples
                              at temperature 0.7 (generate_training_data.py:110-119, pipeline.py:94). */
    int splits;            /* launch geometry for tuning sweeps: workgroups per row in [1,ASD_MAX_SPLITS]; 0 = heuristic */
    int threads;           /* 256 | 512 | 1024 lanes per workgroup; 0 = heuristic */
    int unroll;            /* tile size in KiB a wave claims at a time: 2 | 3 | 4 | 8; 0 = heuristic */
    int nontemporal;       /* 0 | 1; -1 = heuristic */
} asd_verify_options;

int asd_verify_accept_ex(const void* logits, int dtype, int64_t ld_row,
                         const int32_t* tok, const float* lp_draft, const float* u,
                         int B, int K, int V,
                         float* lp_target, uint8_t* accept, int32_t* n_acc, uint64_t* accept_bits,
                         void* workspace, size_t workspace_bytes,
                         const asd_verify_options* opt /*host, may be NULL*/, void* stream);

/* N1, the A14 half: the verify pass that also emits, per verified position, what the missing
 * FeatureExtractor of docs/guides/RESEARCH_PROTOCOL.md:378-400 derives from per-token log-prob lists on the host:
 *   row_max_lp[b,k]  = max_v log softmax(logits[b,k]/T)[v]   (the doc's np.max(lp); free: it is -ln s of the running pair)
 *   row_entropy[b,k] = -sum_v p_v ln p_v  of that softmax      (the doc's -sum exp(lp) * lp taken over the WHOLE vocabulary
 *                      instead of the top-k list vLLM returns; one more FMA per element in the streaming loop)
 * Either may be NULL (both NULL == asd_verify_accept_ex with inv_temperature).  With row_entropy the launch is one
 * workgroup per row whatever the batch and K <= 32 (ASD_ERR_UNSUPPORTED otherwise).  A row without any finite logit
 * reports NaN for both. */
int asd_verify_accept_stats(const void* logits, int dtype, int64_t ld_row,
                            const int32_t* tok, const float* lp_draft, const float* u, int B, int K, int V,
                            float* lp_target, uint8_t* accept, int32_t* n_acc, uint64_t* accept_bits,
                            float* row_max_lp /*[B,K] out, may be NULL*/, float* row_entropy /*[B,K] out, may be NULL*/,
                            void* workspace, size_t workspace_bytes, float inv_temperature, void* stream);

/* Vocab-sharded target (lm_head split over ranks): each rank reduces its [B,K,V_shard] slice,
 * whose first column is global vocab id `v_offset`, to msg[b,k,:] = (m2, s, g) with
 *   sum_v exp(x_v) = s * 2^m2   (log2 domain: m2 = log2(e) * max_v x up to rounding, which cancels),
 *   g = x[tok - v_offset] if the token lies in this shard, else -inf.
 * One all-gather of msg ([B,K,3] f32 per rank) is the only exchange step. */
int asd_lse_partial(const void* logits_shard, int dtype, int64_t ld_row,
                    const int32_t* tok /*[B,K] GLOBAL ids*/, int B, int K, int V_shard,
                    int64_t v_offset, float inv_temperature, float* msg /*[B,K,3] out*/,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Combine the all-gathered partials (fixed shard order => identical on every rank) and run the
 * acceptance test.  msg_all: [n_shards][B][K][3]. */
int asd_accept_from_partials(const float* msg_all, int n_shards,
                             const float* lp_draft, const float* u, int B, int K, float inv_temperature,
                             float* lp_target, uint8_t* accept, int32_t* n_acc,
                             uint64_t* accept_bits, void* stream);

/* ------------------------------------------------------------------------------------------
 * N2  lm_head projection fused with the verify pass.  The reference materialises
 * logits = lm_head(hidden) and then gathers log-probs from them
 * (src/training/generate_training_data.py:128-136 via model.generate(output_scores=True));
 * here  logits[b,k,v] = sum_d hidden[b,k,d] * weight[v,d]  live only in MFMA accumulators (f32)
 * and are reduced on chip to the same (m2, s, g) partials as asd_lse_partial, one per 128-column
 * vocabulary block, then combined and tested exactly like asd_accept_from_partials.
 * hidden [B*K][ld_h], weight [V][ld_w] (the nn.Linear / HF lm_head layout), both bf16 or both f16 (dtype =
 * ASD_DTYPE_BF16 | ASD_DTYPE_F16 -- the reference loads its models in fp16, generate_training_data.py:79-85; f32:
 * ASD_ERR_UNSUPPORTED), D % 64 == 0, ld % 8 == 0, 16-byte aligned bases.
 * The log-sum-exp is over UNROUNDED f32 logits, i.e. closer to the exact product than the
 * bf16-materialised path: results agree with asd_verify_accept on bf16(hidden @ weight.T) to the
 * bf16 rounding of the logits, not bit for bit (tests/test_gpu_lm_head.py states the tolerance).
 * Workspace: asd_lm_head_verify_workspace_bytes(B, K, V), no initialisation needed. */
size_t asd_lm_head_verify_workspace_bytes(int B, int K, int V);
int asd_lm_head_verify(const void* hidden, int64_t ld_h, const void* weight, int64_t ld_w, int dtype, int D,
                       const int32_t* tok, const float* lp_draft, const float* u, int B, int K, int V,
                       float inv_temperature, float* lp_target, uint8_t* accept, int32_t* n_acc,
                       uint64_t* accept_bits, void* workspace, size_t workspace_bytes, void* stream);
/* The same call with the row arg-max: argmax_out[b,k] = argmin-id argmax_v logits[b,k,v] (ties -> lowest id,
 * NaN logits never win, -1 if the row has no logit > -inf; may be NULL).  greedy != 0 replaces the sampling
 * test by greedy verification, accept[b,k] = (tok[b,k] == argmax_out[b,k]); lp_draft and u are then unused
 * (may be NULL) and lp_target is still the draft token's log-prob under the target.  With the arg-max a
 * hidden-state tier needs no logits row for greedy decoding: the correction token after the accepted
 * prefix is argmax_out[b, n_acc[b]], the bonus token the arg-max of an extra row. */
int asd_lm_head_verify_ex(const void* hidden, int64_t ld_h, const void* weight, int64_t ld_w, int dtype, int D,
                          const int32_t* tok, const float* lp_draft, const float* u, int B, int K, int V,
                          float inv_temperature, int greedy, float* lp_target, uint8_t* accept, int32_t* n_acc,
                          uint64_t* accept_bits, int32_t* argmax_out /*[B,K] out, may be NULL*/,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Optional one-time re-layout of an lm_head matrix for the three calls above and below (like asd_mlp_pack_weights for
 * the predictor): [V][ld_w] bf16 -> tile-major [ceil(V/256)][D/64][256 rows][64 columns], rows past V zero.  A column
 * block's 64-deep reduction step is then ONE contiguous 32 KiB run instead of 256 lines that sit D*2 bytes apart (each
 * in another DRAM page).  Pass the packed image as `weight` with ld_w = 0; results are bit-identical to the unpacked call.
 * For a vocabulary shard (asd_lm_head_partial) pack the shard's rows.  packed: asd_lm_head_packed_bytes(V, D) bytes. */
size_t asd_lm_head_packed_bytes(int V, int D);
int asd_lm_head_pack_weights(const void* weight, int64_t ld_w, int dtype, int V, int D, void* packed /*out*/,
                             size_t packed_bytes, void* stream);

/* Tensor-parallel lm_head (weight split over ranks along the vocabulary): this rank's [V_shard, D] slice,
 * whose row 0 is global vocabulary id v_offset, reduced straight to the asd_lse_partial message
 * msg[b,k,:] = (m2, s, g); all-gather the messages and finish with asd_accept_from_partials.  The
 * logits of the shard are never formed; [B,K,3] floats per rank is the only exchange. */
int asd_lm_head_partial(const void* hidden, int64_t ld_h, const void* weight_shard, int64_t ld_w, int dtype, int D,
                        const int32_t* tok /*[B,K] GLOBAL ids*/, int B, int K, int V_shard, int64_t v_offset,
                        float inv_temperature, float* msg /*[B,K,3] out*/, void* workspace, size_t workspace_bytes,
                        void* stream);

/* ------------------------------------------------------------------------------------------
 * X3 (callers' side of the path)  y[M][N] = x[M][D] . w[N][D]^T (+ bias[N]): the nn.Linear projections of the
 * decoder layers the reference runs through transformers / vLLM (third party there: src/serving/real_model_pipeline.py:135,
 * src/models/stage.py) -- the step that produces the hidden states asd_lm_head_verify consumes and, for the draft tier, the
 * logits asd_draft_sample consumes.  The lm_head kernels' main loops with a STORE epilogue; narrow matrices are cut into
 * reduction slices whose f32 partials meet in `workspace` and are added in slice order (bit-reproducible).
 * x, w, bias (may be NULL), y: all bf16 or all f16 (dtype); f32 accumulation, ONE rounding at the store.
 * D % 64 == 0, N % 4 == 0, x / w 16-byte aligned with ld % 8 == 0, y 8-byte aligned with ld_y % 4 == 0.
 * ld_w == 0: w is the tile-major image asd_lm_head_pack_weights(w, ld, dtype, N, D, ...) writes (asd_lm_head_packed_bytes(N, D)
 * bytes: N * D * 2 when N % 256 == 0, so a matrix can be re-laid in its own storage); same results, bit for bit.
 * workspace: asd_linear_workspace_bytes(M, N, D) bytes (no initialisation needed; may be shared by calls on one stream). */
size_t asd_linear_workspace_bytes(int M, int N, int D);
int asd_linear(const void* x, int64_t ld_x, const void* w, int64_t ld_w, const void* bias, int dtype, int M, int N, int D,
               void* y /*[M][ld_y] out*/, int64_t ld_y, void* workspace, size_t workspace_bytes, void* stream);
/* ... + residual[M][ld_res] (may be NULL; may alias y: y = y + x . w^T, the decoder layer's residual connection) */
int asd_linear_ex(const void* x, int64_t ld_x, const void* w, int64_t ld_w, const void* bias, const void* residual,
                  int64_t ld_res, int dtype, int M, int N, int D, void* y /*[M][ld_y] out*/, int64_t ld_y, void* workspace,
                  size_t workspace_bytes, void* stream);
/* Building block of asd_decoder_forward: as asd_linear_ex, but a plan with reduction slices stops after the product --
 * *k_slices > 1 on return means y was NOT written: the f32 partials lie in `workspace` as [k_slices][M][N] and the caller's
 * next kernel adds them in slice order, then the bias, then the residual (the order asd_linear_ex uses: same bits).
 * *k_slices == 1: y is complete. */
int asd_linear_partial(const void* x, int64_t ld_x, const void* w, int64_t ld_w, const void* bias, const void* residual,
                       int64_t ld_res, int dtype, int M, int N, int D, void* y, int64_t ld_y, void* workspace,
                       size_t workspace_bytes, void* stream, int* k_slices /*out*/);
/* host: the number of reduction slices the launcher uses for this product (1 = unsliced; 0 = unsupported shape) */
int asd_linear_slices(int M, int N, int D);

/* ------------------------------------------------------------------------------------------
 * X3, continued: the rest of a decoder layer around the projections, for the M = B * T positions a tier is fed in one pass
 * (row m = b * T + t), over a per-sequence KV cache.  bf16 only; head_dim 128 (every Qwen2.5 shape).
 *   asd_rmsnorm        out = x * rsqrt(mean(x^2) + eps) * weight, f32 arithmetic, one rounding; D % 4 == 0, D <= 8192
 *   asd_rope_kv_store  qkv [M][ld] = q heads | k heads | v heads: rotary embedding (half-split pairs (i, i + 64), angle
 *                      pos[m] * inv_freq[i]) of q IN PLACE and of k INTO the cache; v into the TRANSPOSED cache
 *                        k_cache  [cache rows][KVH][t_max][128]      vt_cache [cache rows][KVH][128][t_max]
 *                      cache row of sequence b: rows[b] (rows == NULL: b); pos is clamped to [0, t_max - 1]; t_max % 32 == 0
 *   asd_attn_ragged    out[m][head] = softmax_j<=pos[m] (q . k_j / sqrt(128)) v_j over the cache of sequence m / T (grouped
 *                      queries: head / (H / KVH) picks the kv head); MFMA flash-decode, both cache operands read in operand
 *                      layout straight from global memory (that is what the transposed V cache is for)
 *   asd_silu_mul       act[m][i] = silu(gate_up[m][i]) * gate_up[m][I + i]
 *   asd_decoder_forward  n_layers x (rmsnorm, qkv, rope + cache write, attention, o + residual, rmsnorm, gate | up, silu * up,
 *                      down + residual) on x [M][ld_x] in place -- nine launches per layer, all issued by this one call.
 * Entries of the caches past a sequence's committed length are rewritten before they are read (KV rollback = a length
 * update, asd_commit_step).  CONTRACT of the caches (X3 is plumbing around the path, third party in the reference):
 *   - they must hold FINITE values from the start (zero-fill them once): asd_attn_ragged masks keys beyond pos by giving them
 *     probability 0, but still feeds the whole 32-key tile through the MFMA, and 0 * NaN = NaN;
 *   - positions are clamped into [0, t_max - 1] by BOTH kernels; slot t_max - 1 is therefore a TRASH slot (the padding
 *     behind a ragged feed lands there, several rows may write it): size t_max at least one slot beyond the longest real
 *     sequence (serving/hierarchy.py does) and never pass a negative position for a row whose result is used. */
typedef struct asd_layer {
    const void* ln1_w;      /* [hidden] */
    const void* qkv_w;      /* [hidden + 2 * KVH * 128][hidden]: q | k | v projection rows */
    const void* qkv_b;      /* [hidden + 2 * KVH * 128] or NULL */
    const void* o_w;        /* [hidden][hidden] */
    const void* ln2_w;      /* [hidden] */
    const void* gate_up_w;  /* [2 * intermediate][hidden]: gate rows, then up rows */
    const void* down_w;     /* [hidden][intermediate] */
    void* k_cache;
    void* vt_cache;
    int weights_packed;     /* != 0: the four matrices are tile-major images (asd_lm_head_pack_weights of each, see asd_linear) */
} asd_layer_t;
typedef struct asd_decoder_shape {
    int hidden, heads, kv_heads, head_dim, intermediate;
    float rms_eps;
    const float* inv_freq;  /* [64] f32, device: theta^(-2 i / 128) */
    int t_max;              /* positions per cache row */
} asd_decoder_shape_t;
int asd_rmsnorm(const void* x, int64_t ld_x, const void* weight, float eps, int dtype, int M, int D, void* out /*[M][ld_out]*/,
                int64_t ld_out, void* stream);
int asd_rope_kv_store(void* qkv /*[B*T][ld_qkv] in/out*/, int64_t ld_qkv, const int32_t* pos /*[B*T]*/,
                      const int32_t* rows /*[B] or NULL*/, const float* inv_freq /*[64]*/, int dtype, int B, int T, int H, int KVH,
                      int head_dim, void* k_cache, void* vt_cache, int t_max, void* stream);
int asd_attn_ragged(const void* qkv, int64_t ld_qkv, const void* k_cache, const void* vt_cache, const int32_t* pos,
                    const int32_t* rows, int dtype, int B, int T, int H, int KVH, int head_dim, int t_max,
                    void* out /*[B*T][ld_out], H * 128 wide*/, int64_t ld_out, void* stream);
int asd_silu_mul(const void* gate_up /*[M][ld_gu], 2 I wide*/, int64_t ld_gu, int dtype, int M, int I, void* act /*[M][ld_act]*/,
                 int64_t ld_act, void* stream);
size_t asd_decoder_scratch_bytes(const asd_decoder_shape_t* shape, int M);
/* final_norm_w / normed_out (both or neither): the model's last RMSNorm, applied to the final x into normed_out [B*T][ld_normed]
 * (fused with the last layer's residual connection).  Where a projection is cut into reduction slices, the kernel that follows
 * it adds the slices' partials itself (asd_linear_partial), so a layer is nine launches whatever the plans are. */
int asd_decoder_forward(const asd_layer_t* layers, int n_layers, const asd_decoder_shape_t* shape, void* x /*[B*T][ld_x] in/out*/,
                        int64_t ld_x, const int32_t* pos, const int32_t* rows, int B, int T, const void* final_norm_w,
                        void* normed_out, int64_t ld_normed, void* scratch /*256-byte aligned*/, size_t scratch_bytes,
                        void* stream);

/* ------------------------------------------------------------------------------------------
 * N3  commit / KV rollback bookkeeping of one token-level step, on the device.
 * The reference's src/serving/cache_manager.py:149-190 `truncate_at_stage` trims a dict of strings;
 * with a per-sequence KV cache the rollback after a rejection is a length update.  Row b of the token
 * buffer `out_tokens` ([B][ld_out] i32) receives tok[b, 0..n_acc[b]) followed by drawn[b] (the residual /
 * bonus token of asd_residual_sample) at index seq_len[b]; seq_len[b] += n_acc[b] + 1, clamped to
 * max_len (tokens past max_len are dropped); n_commit[b] (may be NULL) = tokens appended.  n_acc is
 * clamped to [0, K].  No host synchronisation is needed to learn n_acc. */
int asd_commit_step(const int32_t* tok /*[B,K]*/, const int32_t* n_acc /*[B]*/, const int32_t* drawn /*[B]*/,
                    int B, int K, int32_t* seq_len /*[B] in/out*/, int32_t* out_tokens /*[B][ld_out]*/,
                    int64_t ld_out, int32_t* n_commit /*[B] out, may be NULL*/, int32_t max_len, void* stream);

/* ------------------------------------------------------------------------------------------
 * A7  log-prob statistics: features [5..9] of extract_features,
 * src/training/generate_training_data.py:166-175 -- np.mean, np.std (population), np.min,
 * np.percentile(.,25) (linear interpolation), np.median, all in float64 like numpy.
 * lp: [B rows][<=K valid] f32, rows `ld` elements apart; n_valid[b] in [0,K] (NULL => K).
 * n_valid[b]==0 => five zeros (reference: `features.extend([0.0]*5)`, :174-175).
 * stats: [B,5] f64.
 * ---------------------------------------------------------------------------------------- */
int asd_logprob_stats(const float* lp, int64_t ld, const int32_t* n_valid, int B, int K,
                      double* stats /*[B,5] out*/, void* stream);

/* ------------------------------------------------------------------------------------------
 * A8  MinimalQualityPredictor.forward in eval mode, src/minimal_adaptive_decoder.py:38-49:
 *   score = sigmoid(W2 . relu(W1 x + b1) + b2)          (Dropout(0.1) == identity)
 * packed_w (f32): W1T [in_dim][hidden] (= net.0.weight transposed), b1 [hidden],
 *                 W2 [hidden] (= net.3.weight[0]), b2 [1]   => in_dim*hidden + 2*hidden + 1.
 * asd_mlp_pack_weights (host pointers) builds that buffer from the state_dict tensors.
 * ---------------------------------------------------------------------------------------- */
size_t asd_mlp_packed_floats(int in_dim, int hidden);
int asd_mlp_pack_weights(const float* w1 /*host [hidden,in_dim]*/, const float* b1 /*host*/,
                         const float* w2 /*host [hidden]*/, const float* b2 /*host [1]*/,
                         int in_dim, int hidden, float* packed /*host out*/);
int asd_mlp_predict(const float* x /*[B rows][in_dim], rows ldx apart*/, int64_t ldx,
                    const float* packed_w, int B, int in_dim, int hidden,
                    float* score /*[B] out*/, void* stream);

/* A11  stop test of MinimalAdaptiveDecoder.decode, src/minimal_adaptive_decoder.py:153-164:
 *   stage[b] = first s in [0,L) with (double)score[b] >= theta[s], or s == L-1. */
int asd_threshold_stop(const float* score /*[B]*/, const double* theta /*[L]*/, int B, int L,
                       int32_t* stage /*[B] out*/, void* stream);

/* A2  bayesian_adjustment, src/algorithms/dp_solver.py:106-130 (f64, no FMA contraction):
 *   a = n_obs*p + alpha;  b = n_obs*(1-p) + beta;  out = a / (a + b) */
int asd_bayes_adjust(const double* p /*[n]*/, int64_t n_obs, double alpha, double beta, int n,
                     double* out /*[n] out*/, void* stream);

/* A1  optimal_stopping_rule, src/algorithms/dp_solver.py:12-71, batched over B requests.
 *   p: [B,L] f64; C: [L] f64 shared by the batch; lam f64.
 *   risk_adjustment != 0 applies bayesian_adjustment(p, n_obs=100, alpha, beta) first (:38-39).
 *   k_star: [B] i32; J: [B,L+1] f64 (may be NULL).  Bit-exact with CPython float arithmetic. */
int asd_optimal_stopping(const double* p, const double* C, double lam, int B, int L,
                         int risk_adjustment, double alpha, double beta,
                         int32_t* k_star, double* J, void* stream);

/* A3  compute_expected_cost, dp_solver.py:74-103: sum(C[:k+1]) + lam*(1 - prod(p[:k+1])),
 * for a per-request stopping stage k[b] in [0,L). */
int asd_expected_cost(const double* p /*[B,L]*/, const double* C /*[L]*/, double lam,
                      const int32_t* k /*[B]*/, int B, int L, double* cost /*[B] out*/,
                      void* stream);

/* N4  the DP rule over a GRID of lambda values in one launch: what a lambda controller
 * (src/algorithms/optimizer.py:47-205 LambdaOptimizer, :261-335 GridSearchOptimizer) needs per candidate
 * when requests are described by their predicted stage probabilities.  For every lam[g] and request b:
 *   k_star[g,b]  optimal_stopping_rule(p[b], C, lam[g], risk_adjustment, alpha, beta)      (dp_solver.py:12-71)
 *   cost[g,b]    sum(C[:k*+1]),   p_ok[g,b] = prod(p[b,:k*+1])     (the two terms of compute_expected_cost,
 *                dp_solver.py:74-103: expected cost = cost + lam * (1 - p_ok)).  cost / p_ok may be NULL. */
int asd_lambda_sweep(const double* p /*[B,L]*/, const double* C /*[L]*/, const double* lam /*[G]*/,
                     int B, int L, int G, int risk_adjustment, double alpha, double beta,
                     int32_t* k_star /*[G,B]*/, double* cost /*[G,B]*/, double* p_ok /*[G,B]*/, void* stream);

/* A10  OptimalStoppingTheory.derive_optimal_policy, src/theory/optimal_stopping.py:45-82
 * (+ _compute_improvement_probability :84-91).  HOST pointers: an O(n) f64 recursion run once
 * per set_lambda; theta[n] (theta[n-1] = 0), V[n+1] (may be NULL). */
int asd_derive_thresholds(const double* q, const double* c, int n, double lam,
                          double* theta, double* V);

/* ------------------------------------------------------------------------------------------
 * Fused post-verify epilogue (first form): one launch per tier step does
 *   stats   = logprob_stats(lp[b, :n_valid[b]])                      (A7)
 *   x       = feat[b,:] with x[stats_col .. stats_col+5) = (f32)stats (stats_col < 0: untouched)
 *   score   = mlp(x)                                                  (A8)
 *   p_adj   = risk_adjustment ? bayes(score, n_obs, alpha, beta) : score   (A2, pipeline.py:234-238)
 *   p_hist[b, stage_idx] = p_adj            (p_hist: [B,L] f64, columns < stage_idx are inputs)
 *   k_star  = optimal_stopping_rule(p_hist[b,:n_dp], C[:n_dp], lam)   (A1)  n_dp = prefix ? stage_idx+1 : L
 *             (columns > stage_idx are read as given: the caller pre-fills priors, 1.0 for the last)
 *   theta != NULL additionally:  thr_stop[b] = (double)score >= theta[stage_idx] || stage_idx == L-1   (A11)
 *   stop[b] = (k_star == stage_idx)
 * Outputs that are NULL are skipped.  feat rows are ldf apart and are NOT modified.
 * ---------------------------------------------------------------------------------------- */
int asd_predictor_stop(const float* lp /*[B,K]*/, int64_t ld_lp, const int32_t* n_valid, int K,
                       const float* feat /*[B,in_dim]*/, int64_t ldf, int stats_col,
                       const float* packed_w, int in_dim, int hidden,
                       int risk_adjustment, int64_t n_obs, double alpha, double beta,
                       double* p_hist /*[B,L] in/out*/, const double* C /*[L]*/, double lam,
                       int L, int stage_idx, int prefix_rule,
                       const double* theta /*[L] or NULL*/, int B,
                       float* score /*[B]*/, int32_t* k_star /*[B]*/, uint8_t* stop /*[B]*/,
                       uint8_t* thr_stop /*[B]*/, double* stats /*[B,5]*/, void* stream);

/* ------------------------------------------------------------------------------------------
 * The token a speculative step COMMITS after its accepted prefix (no reference symbol).
 * For sequence b with j = n_acc[b]:
 *   j <  K : w(v) = max(0, softmax(t_logits[b,j]/T)(v) - softmax(d_logits[b,j]/T)(v))   (residual distribution;
 *            if it is empty, i.e. p_t <= p_d everywhere, w = softmax(t_logits[b,j]/T))
 *   j >= K : w(v) = softmax(bonus_logits[b]/T)(v)      (all drafted tokens accepted; bonus_logits NULL => token -1)
 *   token[b] = min { v : sum_{v' <= v} w(v') > r[b] * sum_v w(v) }        (inverse CDF in vocabulary order)
 * t_logits / d_logits: [B*K rows][V], bonus_logits: [B rows][V], all of `dtype`, rows ld_* ELEMENTS apart,
 * 16-byte aligned and a whole number of 16-byte vectors (ASD_ERR_ALIGNMENT otherwise).  r: [B] uniforms in [0,1).
 * workspace: >= asd_residual_sample_workspace_bytes(B, V, dtype), 256-byte aligned, zero-initialised ONCE with asd_workspace_init
 * (since 0.2: up to 64 sequences every sequence's two rows are spread over 2 ... 32 workgroups inside ONE launch, whose single-writer /
 * single-reader mailboxes live in the workspace and are handed back empty by every call; larger batches use the multi-launch /
 * one-workgroup-per-sequence forms, which need no initialisation).  A hand-off that never arrives poisons the sequence (token -1).
 * ---------------------------------------------------------------------------------------- */
size_t asd_residual_sample_workspace_bytes(int B, int V, int dtype);
int asd_residual_sample(const void* t_logits, int64_t ld_t, const void* d_logits, int64_t ld_d,
                        const void* bonus_logits, int64_t ld_b, int dtype,
                        const int32_t* n_acc /*[B]*/, const float* r /*[B]*/, int B, int K, int V,
                        float inv_temperature, int32_t* token /*[B] out*/,
                        void* workspace, size_t workspace_bytes, void* stream);

/* asd_residual_sample for drafts that were drawn with nucleus (top-p) truncation (asd_draft_sample below): the draft
 * distribution at row (b,k) is softmax(d_logits[b,k]/T) restricted to logits >= d_threshold[b,k] and renormalised --
 * the distribution the drafted token was actually drawn from, so the residual max(0, p_t - p_d) stays exact.
 * d_threshold: [B,K] f32 (asd_draft_sample's nucleus_logit per drafted position); NULL == asd_residual_sample. */
int asd_residual_sample_ex(const void* t_logits, int64_t ld_t, const void* d_logits, int64_t ld_d,
                           const void* bonus_logits, int64_t ld_b, int dtype,
                           const int32_t* n_acc, const float* r, int B, int K, int V,
                           float inv_temperature, const float* d_threshold /*[B,K] or NULL*/, int32_t* token,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * X1  the draft tier's per-step token proposal (north star piece (i)).  The reference delegates it to
 * HF `model.generate(do_sample=True, temperature=0.7, top_p=0.9)` (src/training/generate_training_data.py:110-119;
 * third-party arithmetic, parity unpinned); torch needs log_softmax + multinomial + gather over [B,V] per token.
 * One call per drafted position, for row b (one next-token logits row per sequence):
 *   q(v)   = softmax(logits[b]/T)(v) for v in the nucleus N_b, 0 outside, renormalised over N_b
 *   N_b    = { v : logits[b,v] >= x*_b },  x*_b = the LARGEST logit value with  sum_{logits[b,v] >= x*} softmax(.)(v) >= top_p
 *            (HF's TopPLogitsWarper set, with every token tied at the boundary value kept);  top_p outside (0,1): all v
 *   tok[b] = min { v in N_b : sum_{v' in N_b, v' <= v} q(v') > r[b] }          (inverse CDF in vocabulary order, r in [0,1))
 *   lp[b]  = log q(tok[b])                                       (the lp_draft the verify step needs; may be NULL)
 *   nucleus_logit[b] = x*_b (-inf without truncation; may be NULL): with it asd_residual_sample_ex reconstructs q exactly.
 * logits: [B rows][V] of `dtype`, rows ld ELEMENTS apart, 16-byte aligned, a whole number of 16-byte vectors.
 * Rows must not contain NaN / +inf.  ONE launch; V*sizeof(elem) <= 2 MiB.
 * Geometry: up to 128 rows, every row is spread over G = 2 ... 32 workgroups (~one per compute unit) that keep their part
 * of the row in registers and meet through single-writer / single-reader mailbox words in `workspace`; more rows, or a NULL /
 * short workspace: one streaming 1024-lane workgroup per row.  The per-tile partial sums are canonical and folded in a
 * fixed order, so tok / lp / nucleus_logit do NOT depend on the geometry (lp and nucleus_logit bit for bit; the token
 * wherever the draw is more than a float rounding away from a tile boundary of the CDF).
 * workspace: asd_draft_sample_workspace_bytes(B,V,dtype) bytes (~25 MB), 256-byte aligned, zero-initialised ONCE with
 * asd_workspace_init; every call hands it back all-zero, so one workspace serves any number of stream-ordered calls of any
 * B' <= B (two calls that may run concurrently need two).  A hand-off that never arrives (bounded wait) POISONS the row --
 * tok = -1, lp = nucleus_logit = NaN -- it is never guessed.
 * ---------------------------------------------------------------------------------------- */
size_t asd_draft_sample_workspace_bytes(int B, int V, int dtype);
int asd_draft_sample(const void* logits, int64_t ld, int dtype, const float* r /*[B]*/, int B, int V,
                     float inv_temperature, float top_p, int32_t* tok /*[B] out*/, float* lp /*[B] out, may be NULL*/,
                     float* nucleus_logit /*[B] out, may be NULL*/, void* workspace, size_t workspace_bytes, void* stream);

/* N1, second form: asd_verify_accept with the epilogue of asd_predictor_stop run INSIDE the same
 * launch by the wave that completes each sequence (lp = the kernel's own lp_target, all K
 * positions valid).  ONE launch per tier step instead of two, at every batch size: with one workgroup
 * per row (B*K >= CUs) every row hands its lp_t to the sequence's finisher through a self-tagging
 * workspace slot; with split rows (B*K < CUs, e.g. B = 8) the finisher wave already holds all K lp_t.
 * The in-kernel form covers the reference's 64->32->1 predictor (src/minimal_adaptive_decoder.py:38-49) and -- with one
 * workgroup per row -- the 256->128->1 predictor its server instantiates (QualityPredictor(feature_dim=256),
 * src/serving/server.py:168; docs/guides/RESEARCH_PROTOCOL.md:315-364), for hierarchies of <= 4 tiers and draft lengths
 * <= 16; any other predictor shape, depth, draft length or forced geometry is served by the same call as two launches with
 * identical results (bit-identical: tests/test_gpu_predictor.py).  Parameters: those of asd_verify_accept followed by those
 * of asd_predictor_stop (without lp / n_valid / K / B); _ex adds asd_verify_options (inv_temperature:
 * the tiers verify at T = 0.7, pipeline.py:94).  Replaces the per-stage sequence
 * predictor.predict -> bayesian_adjustment -> optimal_stopping_rule of src/serving/pipeline.py:225-261. */
int asd_verify_accept_fused(const void* logits, int dtype, int64_t ld_row,
                            const int32_t* tok, const float* lp_draft, const float* u, int B, int K, int V,
                            float* lp_target, uint8_t* accept, int32_t* n_acc, uint64_t* accept_bits,
                            void* workspace, size_t workspace_bytes,
                            const float* feat, int64_t ldf, int stats_col,
                            const float* packed_w, int in_dim, int hidden,
                            int risk_adjustment, int64_t n_obs, double alpha, double beta,
                            double* p_hist, const double* C, double lam, int L, int stage_idx, int prefix_rule,
                            const double* theta,
                            float* score, int32_t* k_star, uint8_t* stop, uint8_t* thr_stop, double* stats,
                            void* stream);
int asd_verify_accept_fused_ex(const void* logits, int dtype, int64_t ld_row,
                               const int32_t* tok, const float* lp_draft, const float* u, int B, int K, int V,
                               float* lp_target, uint8_t* accept, int32_t* n_acc, uint64_t* accept_bits,
                               void* workspace, size_t workspace_bytes,
                               const float* feat, int64_t ldf, int stats_col,
                               const float* packed_w, int in_dim, int hidden,
                               int risk_adjustment, int64_t n_obs, double alpha, double beta,
                               double* p_hist, const double* C, double lam, int L, int stage_idx, int prefix_rule,
                               const double* theta,
                               float* score, int32_t* k_star, uint8_t* stop, uint8_t* thr_stop, double* stats,
                               const asd_verify_options* opt /*host, may be NULL*/, void* stream);

#ifdef ASD_TEST_HOOKS
/* ------------------------------------------------------------------------------------------
 * TEST HOOKS -- not part of the product library.  They are process-global switches (not thread-safe next to the
 * reference's 100-thread caller, src/serving/pipeline.py:83), so libasd_hip.so does not contain them at all: they exist in
 * the separate TEST build of the same sources (-DASD_TEST_HOOKS -> lib/libasd_hip_test.so, build.py), which the test-suite
 * loads next to the product library for the few tests that need them (kernels.test_hooks()).
 * ---------------------------------------------------------------------------------------- */
/* force the reduction slices of asd_linear* (0 = the launcher's own choice); returns the previous value */
int asd_debug_force_linear_slices(int k_slices);
/* 256 < M <= 288 as one 288-row block (1, default) or as 256 + 32 rows (0); returns the previous value */
int asd_debug_linear_tall(int on);
/* force the workgroups per sequence of the following asd_residual_sample[_ex] calls (1 ... 32; -1 = never the group form;
 * 0 = heuristic) */
int asd_debug_residual_groups(int groups);
/* force the workgroups per row of the following asd_draft_sample calls (1, 2, 4 ... 32; -1 = the one-workgroup streaming
 * form; 0 = heuristic).  Results must not depend on it. */
int asd_debug_draft_groups(int groups);
/* fault injection: in the asd_draft_sample calls that follow, partner workgroup g >= 1 of row b (index = b * G + g, G = the
 * launch's workgroups per row) does not publish its tile pairs.  The leader's bounded wait (~0.5 s) must end in tok = -1 /
 * lp = nucleus_logit = NaN for that row and ASD_WS_LOST_HANDOFF in the workspace's status word.  index < 0 = off. */
int asd_debug_draft_withhold(int index);
/* fault injection: in the verify launches that follow, the workgroup with linear index row * S + split (S = the launch's
 * splits per row) does not publish its hand-off slot.  The kernel's bounded wait must then POISON the affected outputs --
 * lp_target = NaN and accept = 0 for the row (split rows), score = NaN / k_star = L - 1 / stop = 0 for the sequence
 * (in-kernel epilogue) -- and raise ASD_WS_LOST_HANDOFF in the workspace's status word; never return a plausible wrong
 * value.  index < 0 switches the hook off. */
int asd_debug_verify_withhold(int index);
#endif /* ASD_TEST_HOOKS */

#ifdef __cplusplus
}
#endif
#endif /* ASD_HIP_H */
