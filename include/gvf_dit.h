/*
 * gvf_dit.h -- C ABI of the MI355X (gfx950) kernels behind the DiT denoise step.
 *
 * Reference seams these replace:
 *   model/attention/full_attn.py:74-140   scaled_dot_product_attention(qkv | q,kv | q,k,v)   -> gvf_attn_fwd_bf16
 *   model/attention/modules.py:8-15       MultiHeadRMSNorm (fused into the attention prologue) -> gamma_q / gamma_k
 *   model/attention/modules.py:112-146    the nn.Linear projections around it                 -> gvf_gemm_bf16
 *   model/dit.py:128-138, 246-277         FeedForwardNet, adaLN modulate / gate / residual     -> gvf_gemm_bf16 epilogues,
 *                                                                                               gvf_layernorm_modulate_bf16
 * Numerics: 16-bit contraction operands, accumulation fp32 (MFMA 16x16x32 / 32x32x16), softmax / LayerNorm /
 * RMSNorm / residual stream in fp32 -- the placement the reference gets from torch.autocast.
 * The operand type is a run-time argument `dtype` of every entry point below: GVF_DT_F16 (what the reference
 * runs: accelerate mixed_precision='fp16', inference_dpm_latent.py:122-125) or GVF_DT_BF16 (what BASELINE.json
 * names); same MFMA rate, same layouts.  The `*_bf16` names are the round-1/2 entry points, kept as wrappers
 * with dtype = GVF_DT_BF16.  "16-bit" below means the type `dtype` selects.
 * Conventions as in gvf_rast.h: device pointers, caller-owned buffers, explicit stream, int status.
 */
#ifndef GVF_DIT_H
#define GVF_DIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GVF_DT_BF16 0
#define GVF_DT_F16  1

/* gvf_gemm epilogues:  acc[m][n] = sum_k A[m][k] * W[n][k]  (+ bias[n]) */
#define GVF_EPI_STORE_BF16   0   /* C bf16 [M][ldc]  = acc                                  */
#define GVF_EPI_GELU_BF16    1   /* C bf16           = gelu_tanh(acc)      (mlp.0)          */
#define GVF_EPI_STORE_F32    2   /* C f32  [M][ldc]  = acc                                  */
#define GVF_EPI_RESID_F32    3   /* C f32 (in place) += gate[m / rows_per_group][n] * acc   (gate null -> 1): x = x + g*h */
#define GVF_EPI_GEGLU_16     4   /* C 16-bit [M][N/2] = r16(v) * gelu_erf(r16(g)): GEGLU (model/autoencoder.py:90-93) of a projection whose rows (and bias)
                                    come in 64-row groups of 32 value rows then their 32 gate rows (output column 32 b + i <- rows 64 b + i and
                                    64 b + 32 + i); N % 64 == 0; bit-identical to the store epilogue followed by gvf_geglu */

/* C = A W^T (+bias): A bf16 [M][lda] row-major, W bf16 [N][ldw] row-major (nn.Linear layout),
 * K a multiple of 64 (pad activations and weights), lda/ldw multiples of 8, bias f32 [N] or null.
 * gate f32: row g of `gate` (leading dimension gate_ld) applies to rows [g*rows_per_group, ...). */
int gvf_gemm(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc,
             int M, int N, int K, int epilogue, const float* gate, int gate_ld, int rows_per_group, void* stream);
int gvf_gemm_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc,
                  int M, int N, int K, int epilogue, const float* gate, int gate_ld, int rows_per_group,
                  void* stream);

/* ---- LayerNorm folded into the GEMMs around it (the DiT block: x = x + g * h, then LN(x) feeds the next projection) ----
 * gvf_gemm_bf16_resid_stats = gvf_gemm_bf16 with GVF_EPI_RESID_F32 that ALSO writes, per row of the updated fp32 stream,
 * gvf_gemm_stats_parts(N) partial (sum, sum of squares) pairs (one per 64-column slice; row_stats f32 [M][parts][2]; N a
 * multiple of 128).  gvf_gemm_ln_bf16 consumes them:  C = epilogue( (LN(X) * s + t) W^T + bias )  with X the fp32 stream
 * (ldx floats, K = its width <= 1024), LN over the K columns from the partial statistics (eps), s / t the optional affine
 * (ln_w, ln_b) and / or adaLN (1 + scale[g], shift[g]; g = row / rows_per_group, rows_per_group a multiple of 128) terms.
 * The bf16 operand is the rounded normalised value -- exactly what gvf_layernorm_modulate_bf16 would have written --
 * but the 37.7 MB LayerNorm pass and its launch are gone.  epilogue: STORE_BF16, GELU_BF16 or STORE_F32. */
int gvf_gemm_stats_parts(int N);
int gvf_gemm_resid_stats(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, float* C, int ldc,
                         int M, int N, int K, const float* gate, int gate_ld, int rows_per_group, float* row_stats,
                         void* stream);
int gvf_gemm_ln(int dtype, const float* X, int ldx, const float* row_stats, int n_part, float eps, const float* ln_w,
                const float* ln_b, const float* shift, const float* scale, int mod_ld, int rows_per_group,
                const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N, int K, int epilogue,
                void* stream);
int gvf_gemm_bf16_resid_stats(const void* A, int lda, const void* W, int ldw, const float* bias, float* C, int ldc,
                              int M, int N, int K, const float* gate, int gate_ld, int rows_per_group, float* row_stats,
                              void* stream);
int gvf_gemm_ln_bf16(const float* X, int ldx, const float* row_stats, int n_part, float eps, const float* ln_w,
                     const float* ln_b, const float* shift, const float* scale, int mod_ld, int rows_per_group,
                     const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N, int K, int epilogue,
                     void* stream);

/* ---- row-block kernel: one launch per sub-layer boundary of the DiT block (model_channels C = 512 only) -------------------------
 * Replaces, in ONE launch over 48-row blocks of the fp32 stream x [M][512] (M a multiple of 48), this run of the reference's
 * ModulatedSparseTransformerCrossBlock._forward (model/dit.py:236-277) and DiT._forward (model/dit.py:449-480):
 *     x   += gate1 * (A W1^T + b1)                  the projection closing a sub-layer (to_out, or input_layer for A = the input)
 *     hb   = bf16(LN(x) * mul1 + add1)              the LayerNorm opening the next one (ln1: affine and / or adaLN modulate)
 *   [ x   += gate_m * (gelu(hb Wfc1^T + b_fc1) Wfc2^T + b_fc2);  hb = bf16(LN(x) * mul2 + add2) ]      when hidden != 0: the whole MLP
 *     out3 = epi3(hb W3^T + b3)                     the first projection of the next sub-layer (to_qkv / to_q), N3 a multiple of 512
 * or, with N3 == 0, hb_out = hb (rows of the last LayerNorm, bf16 [M][512]) for a projection the kernel does not cover
 * (final_layer.linear).  The normalised rows and the MLP's hidden units never leave the CU.
 * `w` is ONE weight stream in MFMA-fragment order, the segments back to back in the order the kernel consumes them:
 *   W1:  gvf_rowblock_pack_weight (nn.Linear weight bf16 [N][K], N a multiple of 512, K padded to a multiple of 128 ->
 *        gvf_rowblock_packed_bytes(N, K) bytes),
 *   MLP: gvf_rowblock_pack_mlp (mlp.0 weight [hidden][512] and mlp.2 weight [512][hidden] -> 2 * hidden * 512 * 2 bytes, interleaved
 *        per 512 hidden units; hidden a multiple of 512, <= 2048)           -- only when hidden != 0,
 *   W3:  gvf_rowblock_pack_weight                                            -- only when N3 != 0.
 * K1 = 0 skips the closing projection (x already holds the sub-layer's result; `a`, b1, gate1 NULL, no W1 segment in `w`).  With the MLP and
 * neither N3 nor hb_out the launch ends after the MLP's update of x (LayerNorm ln2 is skipped: final_layer reads the stream itself).
 * K1 (<= 512, multiple of 128) is the PADDED depth of W1; A is bf16 [M][lda] with lda >= K1.  gate / shift / scale: f32, row g =
 * row / rows_per_group of leading dimension mod_ld (rows_per_group a multiple of 48); NULL = no gate / no modulate.
 * Rounding points are those of the unfused launches: bf16 operands, fp32 accumulation, fp32 stream, LayerNorm in fp32,
 * bf16 hidden units. */
typedef struct gvf_rowblock_ln {
    const float* ln_w; const float* ln_b;      /* [512] or both NULL */
    const float* shift; const float* scale;    /* or both NULL */
} gvf_rowblock_ln;
typedef struct gvf_rowblock_args {
    const void* a; int32_t lda; int32_t K1; const void* w; const float* b1;
    float* x; int32_t M; int32_t C;
    const float* x_in; int32_t x_in_period;    /* optional: residual read from x_in[(row / rows_per_group) * period + (row % rows_per_group) % period]
                                                   (f32 [groups * period][512]) instead of x: input_layer on top of the position embedding */
    /* optional, K1 == 0: x (or x_in) += in_x[row] w_in^T + in_b in fp32 before the LayerNorm -- input_layer (model/dit.py:455-460);
       in_x f32 [M][in_cin], in_wt = the weight TRANSPOSED f32 [in_cin][512], in_cin a multiple of 4, <= 16 */
    const float* in_x; const float* in_wt; const float* in_b; int32_t in_cin;
    const float* gate1; gvf_rowblock_ln ln1;
    int32_t mod_ld; int32_t rows_per_group; float eps;
    const float* b_fc1; const float* b_fc2; int32_t hidden; const float* gate_m; gvf_rowblock_ln ln2;
    const float* b3; void* out3; int32_t N3; int32_t epi3;
    void* hb_out;
    /* optional, N3 = 1536 (to_qkv of the spatial self attention, head_dim 32): q (pass 0) goes to out3 as bf16 [M][512]; k and v go straight
       into the tiled K / V^T images of gvf_attn_tiled_fwd_bf16 -- bit-identical to gvf_attn_pack_kv_bf16(k_scale, gamma_k) on the row-major
       projection, which is then never written.  Key sets = runs of kv_L rows (kv_L a multiple of 64, M a multiple of kv_L).
       kv_group_rows > 0: the stream is laid out in groups of rows_per_group rows (a multiple of 48) of which only the first
       kv_group_rows (a multiple of kv_L) are tokens -- a sample whose T*N is not a multiple of 48, padded; the padding rows are
       computed like any other row but write no keys, and key sets count tokens only (set = group * kv_group_rows / kv_L + ...). */
    void* k_tiles; void* v_tiles; int32_t kv_L; float k_scale; const float* gamma_k;
    int32_t kv_group_rows;
    int32_t dtype;                             /* GVF_DT_BF16 / GVF_DT_F16: type of a, the packed weights, out3, hb_out and the K / V^T tiles */
    /* optional temporal section, t_frames > 0 (no MLP section, K1 > 0, N3 > 0): the launch also runs the block's temporal self attention
       (model/dit.py:255-261 -> model/attention/modules.py:119-140, heads of 32) between ln1 and the last projection:
           hb = ln1(x);  [q | k | v] = hb Wqkv^T + t_b_qkv;  o = softmax_frames(rms(q) rms(k)^T * t_scale) v  per token and head;
           x += t_gate * (o Wout^T + t_b_out);  hb = t_ln(x);  out3 = hb W3^T + b3
       with the weight stream W1 | Wqkv (3 passes of 512, its row blocks in the order V | Q | K: the v pass runs first; t_b_qkv keeps the
       module's order [b_q | b_k | b_v]) | Wout | W3.  A group (sample) is the frame-major stream (T, N, C) with
       t_stride = N: a token's frames are rows t_stride apart.  A workgroup owns 48 / t_frames tokens (t_frames must divide 48), a group
       ceil(N / (48 / t_frames)) workgroups = rows_per_group rows: when that is more than t_frames * N, the rows behind the tokens are the
       group's padding (the phantom tokens of its last workgroup; same layout as kv_group_rows).  t_gamma_q / t_gamma_k: MultiHeadRMSNorm
       gains f32 [512] (both or neither). */
    int32_t t_frames; int32_t t_stride;
    const float* t_b_qkv; const float* t_gamma_q; const float* t_gamma_k; float t_scale;
    const float* t_b_out; const float* t_gate; gvf_rowblock_ln t_ln;
} gvf_rowblock_args;
/* layout of gvf_rowblock_args as compiled: {sizeof, offsetof x, in_x, gate1, mod_ld, b_fc1, ln2, b3, hb_out, k_tiles, gamma_k, kv_group_rows,
   dtype, t_frames, t_b_qkv, t_scale, t_ln}; returns the count */
int gvf_rowblock_args_layout(int32_t* out, int n);
int64_t gvf_rowblock_packed_bytes(int N, int K);
int gvf_rowblock_pack_weight(const void* w_bf16, int ldw, int N, int K, void* packed, void* stream);
int gvf_rowblock_pack_mlp(const void* w_fc1_bf16, const void* w_fc2_bf16, int hidden, void* packed, void* stream);
int gvf_rowblock_fused(const gvf_rowblock_args* args, void* stream);            /* operand type = args->dtype */
int gvf_rowblock_fused_bf16(const gvf_rowblock_args* args, void* stream);       /* args->dtype ignored: bf16 */

/* softmax(q k^T * scale) v, head_dim 32 or 64, no mask, scale > 0.  Batch index = (outer, inner); every tensor
 * takes 4 strides in elements {outer, inner, seq, head}: element (o,i,l,h,c) sits at
 * o*s[0] + i*s[1] + l*s[2] + h*s[3] + c, so the q/k/v slices of a packed qkv / kv projection, a K/V set
 * shared by all `inner` entries (stride 0) and the (B,T,N,.)<->(B,N,T,.) view of the temporal attention
 * need no copies.  v_transposed != 0: v is stored [.., head][d][key] (keys contiguous, v_strides[2] = d
 * stride, rows padded with finite values to a multiple of 64 keys) -- the layout of the DiT's
 * step-invariant cross-attention cache.
 * gamma_q / gamma_k: f32 [H][head_dim] MultiHeadRMSNorm gains (x <- normalize(x) * gamma * sqrt(head_dim)) or null. */
int gvf_attn_fwd(int dtype, const void* q, const void* k, const void* v, void* out,
                 int n_outer, int n_inner, int Lq, int Lk, int H, int head_dim,
                 const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                 const int64_t* o_strides, int v_transposed,
                 const float* gamma_q, const float* gamma_k, float scale, void* stream);
int gvf_attn_fwd_bf16(const void* q, const void* k, const void* v, void* out,
                      int n_outer, int n_inner, int Lq, int Lk, int H, int head_dim,
                      const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                      const int64_t* o_strides, int v_transposed,
                      const float* gamma_q, const float* gamma_k, float scale, void* stream);

/* Variable-length batch over packed token lists (the reference's sparse attention seam,
 * model/sparse_attention/full_attn.py:189-210: flash_attn_varlen_* / xformers BlockDiagonalMask): sequence s
 * owns query rows [cu_seqlens_q[s], cu_seqlens_q[s+1]) and key rows [cu_seqlens_k[s], cu_seqlens_k[s+1]) of
 * the packed tensors; strides as above with strides[0] (outer) normally 0.  cu_seqlens: device int32 [n_seqs+1]. */
int gvf_attn_varlen_fwd(int dtype, const void* q, const void* k, const void* v, void* out, int n_seqs,
                        const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, int max_Lq, int max_Lk,
                        int H, int head_dim,
                        const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                        const int64_t* o_strides, const float* gamma_q, const float* gamma_k, float scale,
                        void* stream);
int gvf_attn_varlen_fwd_bf16(const void* q, const void* k, const void* v, void* out, int n_seqs,
                             const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, int max_Lq, int max_Lk,
                             int H, int head_dim,
                             const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                             const int64_t* o_strides, const float* gamma_q, const float* gamma_k, float scale,
                             void* stream);

/* The plain projection C = r16(A W^T + bias) on 256 x 256 x 64 tiles, one wave per SIMD (csrc/gemm256.hip; opt-in: faster than gvf_gemm's
 * 128-wide kernel on cache-hot operands, not inside the VAE decode).  gvf_gemm256_eligible: M, N multiples of 256, K of 64, 16-byte rows;
 * GVF_GEMM256=1 makes gvf_gemm take it for the store epilogue when the output has at least 256 tiles.  A, W, C 16-bit row-major, bias f32 [N]
 * or null, pointers 16-byte aligned. */
int gvf_gemm256_eligible(int M, int N, int K, int lda, int ldw, int ldc);
int gvf_gemm256(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N, int K,
                void* stream);

/* The same projection on 256 x 256 x 64 tiles with EIGHT waves, two per SIMD (csrc/gemm8.hip, round 6): 790-900 TFLOP/s on the motion VAE's large
 * projections where gvf_gemm's 128-wide kernel reaches 560-640.  epilogue: GVF_EPI_STORE_BF16 (C 16-bit [M][N]) or GVF_EPI_GEGLU_16 (C 16-bit
 * [M][N/2], the interleaved value / gate convention of gvf_gemm) on 256-wide tiles, or GVF_EPI_RESID_F32 WITHOUT a gate (C f32 [M][N] += acc + bias)
 * on 192 x 192 tiles -- one tile per CU for the VAE's 12 288 x 768 residual projections, or GVF_EPI_STORE_F32 (C f32 [M][N] = acc + bias, 256-wide
 * tiles, ANY M: the hoisted condition projections of DiT.prepare_conditions).  gvf_gemm8_eligible: the tile it would use (256 / 192) or 0:
 * M (except GVF_EPI_STORE_F32), N multiples of the tile, K of 64, 16-byte rows.  gvf_gemm takes it by itself for eligible calls with at least one tile per CU
 * (GVF_GEMM8=0: off, =2: from one tile on). */
int gvf_gemm8_eligible(int M, int N, int K, int lda, int ldw, int ldc, int epilogue);
int gvf_gemm8(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N, int K,
              int epilogue, void* stream);

/* ---- cross attention against a pre-tiled, step-invariant K/V cache (csrc/attn_xt.hip), head_dim 32 ----------------
 * The two cross attentions of the DiT block (model/dit.py:263-270 -> model/attention/full_attn.py:74-140) read keys /
 * values that depend on the conditions only.  gvf_attn_pack_kv_bf16 stores them ONCE per condition set in the image
 * the attention workgroups stage into LDS: per (set, head) ceil(L / 64) tiles of 64 keys, 4 KiB of K (pre-multiplied by
 * k_scale = softmax_scale * log2(e) in fp32 before the rounding to bf16; optional MultiHeadRMSNorm gain gamma_k
 * f32 [H][32]) and 4 KiB of V^T each.  kv: f32 (kv_is_f32 != 0) or bf16 rows, row (set * L + key), leading dimension ld
 * (elements); K of head h at columns [k_col0 + 32 h, +32), V at [v_col0 + 32 h, +32).
 * k_tiles / v_tiles: n_sets * H * ceil(L / 64) * 4096 bytes each, 16-byte aligned. */
int gvf_attn_pack_kv(int dtype, const void* kv, int kv_is_f32, int64_t ld, int k_col0, int v_col0, int n_sets, int L, int H,
                     float k_scale, const float* gamma_k, void* k_tiles, void* v_tiles, void* stream);
int gvf_attn_pack_kv_bf16(const void* kv, int kv_is_f32, int64_t ld, int k_col0, int v_col0, int n_sets, int L, int H,
                          float k_scale, const float* gamma_k, void* k_tiles, void* v_tiles, void* stream);
/* The same with the keys of every (set, head) stored in a caller-chosen ORDER: key_order int32 [n_sets][H][L], slot j of (set, head) holds
 * source row key_order[set][head][j] (a permutation of 0 .. L-1; null = identity).  softmax(q k^T) v does not depend on the order of the
 * keys; the fp16 path of gvf_attn_tiled_fwd does, for SPEED: a query's shift is its best score against the FIRST key tile, and a later key that
 * beats it by 2^16 sends the workgroup to the exact pass -- with the largest-norm keys first (attention sinks / artefact tokens are
 * high-norm keys) the first tile's best is almost always within that window of the row's (DiT.prepare_conditions orders its caches so). */
/* The order DiT.prepare_conditions uses: key_order [n_sets][H][L] <- the n_first largest-norm keys of every (set, head) first, the others behind
 * them, both groups in context order (ties at the threshold: the earlier keys).  kv: f32 rows as for gvf_attn_pack_kv (K of head h at columns
 * [k_col0 + 32 h, +32), 16-byte aligned rows); L <= 8192. */
int gvf_attn_key_order(const float* kv, int64_t ld, int k_col0, int n_sets, int L, int H, int n_first, int32_t* key_order, void* stream);
int gvf_attn_pack_kv_ordered(int dtype, const void* kv, int kv_is_f32, int64_t ld, int k_col0, int v_col0, int n_sets, int L, int H,
                             float k_scale, const float* gamma_k, const int32_t* key_order, void* k_tiles, void* v_tiles, void* stream);
/* The two above for n_groups row sets of ONE shape in one launch each (DiT.prepare_conditions: the 12 blocks' to_kv products of a context --
 * model/dit.py:257-262 runs them block by block; they depend on the conditions alone): group g reads kv + g * group_stride (elements; e.g. the
 * next [rows][ld] matrix, or the next column band of one wide matrix), its gain gamma_k + g * H * 32, and writes key_order + g * n_sets * H * L,
 * k_tiles / v_tiles + g * n_sets * H * ceil(L / 64) * 4096 bytes.  Bit-identical to n_groups single calls.  n_groups <= 65535. */
int gvf_attn_key_order_groups(const float* kv, int64_t ld, int64_t group_stride, int n_groups, int k_col0, int n_sets, int L, int H, int n_first,
                              int32_t* key_order, void* stream);
int gvf_attn_pack_kv_groups(int dtype, const void* kv, int kv_is_f32, int64_t ld, int64_t group_stride, int n_groups, int k_col0, int v_col0,
                            int n_sets, int L, int H, float k_scale, const float* gamma_k, const int32_t* key_order, void* k_tiles, void* v_tiles,
                            void* stream);

/* out = softmax(q k^T * scale) v over such a cache (the scale is inside k_tiles).  q / out: bf16, strides
 * {outer, inner, seq, head} in elements as for gvf_attn_fwd_bf16; (outer, inner) reads K/V set
 * outer * kv_set_stride_outer + inner * kv_set_stride_inner (inner stride 0: one set shared by all frames of a sample).
 * gamma_q: optional MultiHeadRMSNorm gain of q, f32 [H][32].  out_is_f32 != 0: out is float with the same element strides
 * (the kernel's arithmetic without the final rounding to bf16; used by the parity tests).
 * Numerics: P = exp2(s) without the running maximum while every query's denominator stays inside [2^-100, 2^100];
 * a workgroup with a query outside recomputes its 256 queries with the exact online softmax (force_exact != 0: always).
 * GVF_DT_F16: every query's scores are shifted by its maximum over the first key tile (through the MFMA accumulator's
 * initial value), so that exp2 stays inside fp16's range; an overflow (a later score 2^16 above that) fails the same guard.
 * force_exact is a set of flags: GVF_ATTN_FORCE_EXACT (1) = always the exact path; GVF_ATTN_SCORES_BOUNDED (2) = the caller vouches that
 * no log2-domain score q . k' exceeds 15.5 (RMS-normalised q and k with known gains: 32 max_d |gamma_q gamma_k| scale log2 e): the fp16
 * kernel then skips the shift (exp2 of such a score fits fp16) and is the bf16 kernel with the other MFMA opcode; a broken promise
 * overflows to inf -- and a query whose scores all sit below -15 underflows -- into the same guard and is recomputed exactly.  Ignored for bf16.
 * fallback_counter (optional, device int32): += 1 per workgroup that took the exact path. */
#define GVF_ATTN_FORCE_EXACT 1
#define GVF_ATTN_SCORES_BOUNDED 2
int gvf_attn_tiled_fwd(int dtype, const void* q, const void* k_tiles, const void* v_tiles, void* out, int n_outer, int n_inner,
                       int Lq, int Lk, int H, const int64_t* q_strides, const int64_t* o_strides,
                       int64_t kv_set_stride_outer, int64_t kv_set_stride_inner, const float* gamma_q,
                       int out_is_f32, int force_exact, int32_t* fallback_counter, void* stream);
int gvf_attn_tiled_fwd_bf16(const void* q, const void* k_tiles, const void* v_tiles, void* out, int n_outer, int n_inner,
                            int Lq, int Lk, int H, const int64_t* q_strides, const int64_t* o_strides,
                            int64_t kv_set_stride_outer, int64_t kv_set_stride_inner, const float* gamma_q,
                            int out_is_f32, int force_exact, int32_t* fallback_counter, void* stream);
/* The same launch, which also TOUCHES [prefetch, prefetch + prefetch_bytes) once (one dword per 128-byte line, spread over the workgroups,
 * read into an LDS landing zone and discarded): the weights of the launch that FOLLOWS on the stream -- in the DiT block every attention is
 * followed by a row-block launch whose 1-6 MB weight stream is new to the caches (107 MB of weights cycle through per denoise step) --
 * arrive in the Infinity Cache while this launch, which is bound by its matrix / vector pipes and leaves the memory system idle, runs.
 * The result does not depend on it.  prefetch_bytes = 0: exactly gvf_attn_tiled_fwd. */
int gvf_attn_tiled_fwd_pf(int dtype, const void* q, const void* k_tiles, const void* v_tiles, void* out, int n_outer, int n_inner,
                          int Lq, int Lk, int H, const int64_t* q_strides, const int64_t* o_strides,
                          int64_t kv_set_stride_outer, int64_t kv_set_stride_inner, const float* gamma_q,
                          int out_is_f32, int force_exact, int32_t* fallback_counter, const void* prefetch, int64_t prefetch_bytes, void* stream);

/* ---- the same for head_dim 64 and key sets of <= 512 keys (csrc/attn_xt64.hip) --------------------------------------
 * The decoder cross attention of the motion VAE (model/autoencoder.py:557-577: every static Gaussian queries the 512 latents of a
 * frame through 12 heads of 64).  gvf_attn_pack_kv64 writes, per (set, head), ceil(L / 64) tiles of 8 KiB of K (pre-multiplied by
 * k_scale = softmax_scale * log2(e) in fp32 before the one rounding) and 8 KiB of V^T in the order the MFMA fragments are read;
 * kv rows as for gvf_attn_pack_kv with K of head h at columns [k_col0 + 64 h, +64), V at [v_col0 + 64 h, +64).
 * k_tiles / v_tiles: n_sets * H * ceil(L / 64) * 8192 bytes each, 16-byte aligned.
 * gvf_attn_tiled64_fwd: out = softmax(q k^T * scale) v, q / out 16-bit with strides {outer, inner, seq, head} in elements (multiples
 * of 8; q, out 16-byte aligned), K/V set of (outer, inner) = outer * kv_set_stride_outer + inner * kv_set_stride_inner; Lk <= 512
 * (the whole key set of a (set, head) is copied into LDS once per 2048 queries).  Numerics as gvf_attn_tiled_fwd: P = exp2(s) without
 * the running maximum behind the [2^-100, 2^100] denominator guard (GVF_DT_F16: per-query shift, guard [2^-6, ...)); a wave with a
 * query outside recomputes its 64 queries with the exact online softmax (force_exact & GVF_ATTN_FORCE_EXACT: always).
 * fallback_counter (optional, device int32): += 1 per 64-query wave pass that took the exact path. */
int gvf_attn_pack_kv64(int dtype, const void* kv, int kv_is_f32, int64_t ld, int k_col0, int v_col0, int n_sets, int L, int H,
                       float k_scale, void* k_tiles, void* v_tiles, void* stream);
int gvf_attn_tiled64_fwd(int dtype, const void* q, const void* k_tiles, const void* v_tiles, void* out, int n_outer, int n_inner,
                         int Lq, int Lk, int H, const int64_t* q_strides, const int64_t* o_strides,
                         int64_t kv_set_stride_outer, int64_t kv_set_stride_inner, int force_exact, int32_t* fallback_counter,
                         void* stream);

/* The same launch with the projection that follows the attention folded into its epilogue (the decoder's to_out o to_outputs, one
 * <= 16 x H*64 matrix: model/autoencoder.py:575-577, 606-608): instead of the H*64-wide 16-bit output rows it writes, per head, the 16 fp32
 * partial products  part[set = outer * n_inner + inner][head][q][m] = sum_d W[m][64 head + d] * r16(o[q][64 head + d])  (r16 = the rounding the
 * stored output would have had) -- half the bytes, and the GEMM that re-read them is gone.  fold_frags: gvf_attn_fold_pack's image of W
 * (16-bit row-major [n_out <= 16][ld >= H*64]; H * 4096 bytes, 16-byte aligned).  gvf_attn_fold_reduce then writes
 * out[set][q][0 .. n_out) = bias + sum over heads (ascending: deterministic), out rows / sets out_row_stride / out_set_stride floats apart. */
int gvf_attn_fold_pack(int dtype, const void* w, int ld, int n_out, int H, void* fold_frags, void* stream);
int gvf_attn_tiled64_fold_fwd(int dtype, const void* q, const void* k_tiles, const void* v_tiles, const void* fold_frags, float* part,
                              int n_outer, int n_inner, int Lq, int Lk, int H, const int64_t* q_strides,
                              int64_t kv_set_stride_outer, int64_t kv_set_stride_inner, int force_exact, int32_t* fallback_counter,
                              void* stream);
int gvf_attn_fold_reduce(const float* part, const float* bias, float* out, int n_sets, int H, int Lq, int n_out,
                         int64_t out_set_stride, int64_t out_row_stride, void* stream);

/* out_bf16[r][:] = LN(x[r][:]) (eps, no affine) then either  * ln_w + ln_b  (affine LayerNorm, norm3/4)
 * or  * (1 + scale[g]) + shift[g]  (adaLN, g = r / rows_per_group; shift/scale rows have stride mod_ld),
 * x f32 [rows][C]; C a multiple of 256 (<= 1024) takes the register-resident fast path. */
int gvf_layernorm_modulate(int dtype, const float* x, void* out16, int rows, int C, float eps,
                           const float* ln_w, const float* ln_b,
                           const float* shift, const float* scale, int mod_ld, int rows_per_group,
                           void* stream);
int gvf_layernorm_modulate_bf16(const float* x, void* out_bf16, int rows, int C, float eps,
                                const float* ln_w, const float* ln_b,
                                const float* shift, const float* scale, int mod_ld, int rows_per_group,
                                void* stream);

/* dst bf16 [rows][ld_dst] = act(src f32 [rows][cols]) with zero padding of columns cols..ld_dst-1.
 * act: 0 = identity, 1 = SiLU. */
int gvf_cast_pad(int dtype, const float* src, int ld_src, void* dst, int ld_dst, int64_t rows, int cols, int act,
                 void* stream);
int gvf_cast_pad_bf16(const float* src, int ld_src, void* dst, int ld_dst, int64_t rows, int cols, int act,
                      void* stream);

/* Two-term bf16 expansion of an fp32 matrix, laid out along K so that ONE plain bf16 gvf_gemm (fp32 accumulation, fp32 output) computes the
 * fp32-class product of the step-invariant condition projections (model/dit.py:464-465 image_cond_proj / static_cond_proj and every block's
 * to_kv(context), model/attention/modules.py:134-143):  hi = bf16(x), lo = bf16(x - hi);
 *   mode 0 (activations):  dst [rows][3 Kp] = [ hi | lo | hi ]        mode 1 (nn.Linear weights):  dst [rows][3 Kp] = [ hi | hi | lo ]
 * Kp = cols rounded up to 64, zero padded.  gemm(dst_a, dst_w) = a w^T - a_lo w_lo^T: relative 2^-16 per term (the K / V cache it feeds is
 * rounded to 16 bits once, 2^-9 .. 2^-12), every output row summed in one fixed order -- a sample's numbers do not depend on its batch. */
int gvf_split3_bf16(const float* src, int64_t ld_src, void* dst, int64_t rows, int cols, int mode, void* stream);

/* ---- the small projections of the denoise step in FP32 (csrc/elem.hip) ----------------------------------------------------------------
 * 0.3 % of the step's FLOPs and more than a third of its bf16 error (their results multiply or feed everything else), so they do not go
 * through the bf16 matrix pipe:
 *   gvf_dit_timestep_embed_f32: out_silu[b] = silu(W2 silu(W0 [cos | sin](t_b f) + b0) + b2), f32 [B][C]   (model/dit.py:59-100, 217-225)
 *   gvf_dit_modulation_f32:     out[b][n] = w[n] . s[b] + bias[n]: every block's adaLN vectors in one GEMV   (model/dit.py:217-225, 298-303)
 *   gvf_dit_input_layer_f32:    out[row] = pos[(row / rows_per_group) * pos_period + (row % rows_per_group) % pos_period] + w_t^T x[row] + bias
 *                               (w_t = the weight TRANSPOSED, f32 [Cin][C])
 *                               (model/dit.py:455-460: input_layer(x) + the position embedding broadcast over the frames; pos may be NULL)
 *   gvf_dit_final_layer_f32:    out[row] = w (LayerNorm(x[row]) * (1 + scale[g]) + shift[g]) + bias              (model/dit.py:298-303)
 * Weights are the nn.Linear fp32 weights as stored ([out][in], contiguous).  C <= 512 (a multiple of 4) for the last two; Cin <= 24; Cout <= 32. */
int gvf_dit_timestep_embed_f32(const float* t, int B, int freq_dim, float max_period, const float* w0, const float* b0, const float* w2,
                               const float* b2, int C, float* out_silu, float* t_emb, void* stream);
int gvf_dit_modulation_f32(const float* s, int B, int C, const float* w, const float* bias, int N, float* out, void* stream);
int gvf_dit_input_layer_f32(const float* x, int M, int Cin, const float* w_t, const float* bias, const float* pos, int pos_period,
                            int rows_per_group, int C, float* out, void* stream);
int gvf_dit_final_layer_f32(const float* x, int M, int C, float eps, const float* shift, const float* scale, int mod_ld, int rows_per_group,
                            const float* w, const float* bias, int Cout, float* out, void* stream);

/* TimestepEmbedder and the SiLU in front of the adaLN projections in one launch (model/dit.py:59-100, 217-225):
 *   t_freq = [cos(t f_i) | sin(t f_i)], f_i = max_period^(-i / (freq_dim/2));  t_emb = W2 silu(W0 t_freq + b0) + b2;
 *   out[b][0..C) = bf16(silu(t_emb[b])), zero-padded to ld_out columns -- the A operand of the GEMM that computes every block's
 * modulation vectors.  t: f32 [B]; w0: bf16 [C][ldw0 >= freq_dim], w2: bf16 [C][ldw2 >= C] (nn.Linear layout); t_emb: optional f32
 * [B][C] copy of the embedding.  freq_dim even, <= 1024; C <= 1024.  Operands are rounded to bf16 where the unfused launches
 * (gvf_cast_pad_bf16 + gvf_gemm_bf16) round them. */
int gvf_dit_timestep_embed_bf16(const float* t, int B, int freq_dim, float max_period, const void* w0_bf16, int ldw0, const float* b0,
                                const void* w2_bf16, int ldw2, const float* b2, int C, void* out_bf16, int ld_out, float* t_emb,
                                void* stream);

/* ---- DPM-Solver state updates (csrc/dpm.hip): the tensor arithmetic of the reference's model/dpmsolver.py steps as single launches ----
 * The schedule coefficients are host floats (the solver keeps its times on the host); x, model outputs and results are contiguous f32 of one shape.
 * Every product / sum / quotient is rounded to fp32 on its own, in the order the reference's expressions evaluate (no fused multiply-add).
 * gvf_dpm_x0       x0 = (x - sigma noise) / alpha                                   data_prediction_fn, model/dpmsolver.py:450-461
 * gvf_dpm_lincomb  out = ((a x) + (b m0)) + (c m1)    (m1 null: (a x) + (b m0))      dpm_solver_first_update :564-609, the intermediate state of
 *                  singlestep_dpm_solver_second_update :611-690, multistep_dpm_solver_second_update :813-869 (regrouped by tensor)
 * gvf_dpm_second_err   the closing launch of an adaptive order-2 step (dpmsolver++, solver_type "dpmsolver"), :1013-1019:
 *                  x_lower = (a x) - (b m);  x_higher = ((a x) - (b m)) - (c (m1 - m));  delta = max(atol, rtol max(|x_lower|, |x_prev|));
 *                  E[0] = max over the n_samples samples of sqrt(mean(((x_higher - x_lower) / delta)^2)) -- a device float the caller reads back
 *                  (the step-size test is the solver's one synchronisation point).  scratch: gvf_dpm_err_scratch_doubles(...) doubles, block sums in a
 *                  fixed order (deterministic).  n_samples <= 65535. */
int gvf_dpm_x0(const float* x, const float* noise, float sigma, float alpha, float* x0, int64_t n, void* stream);
int gvf_dpm_lincomb(const float* x, const float* m0, const float* m1, float a, float b, float c, float* out, int64_t n, void* stream);
int64_t gvf_dpm_err_scratch_doubles(int n_samples, int64_t n_per_sample);
int gvf_dpm_second_err(const float* x, const float* m, const float* m1, const float* x_prev, float a, float b, float c, float atol, float rtol,
                       float* x_lower, float* x_higher, int n_samples, int64_t n_per_sample, double* scratch, float* E, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GVF_DIT_H */
