/*
 * gvf_vae.h -- C ABI of the MI355X (gfx950) kernels that only the motion-VAE decoder needs
 * (SURVEY.md section 8f NEXT #1; the GEMMs, LayerNorm and attention come from gvf_dit.h).
 *
 * Reference seams these replace:
 *   model/autoencoder.py:90-93     GEGLU.forward: x, gates = chunk(2); x * F.gelu(gates) (erf GELU)  -> gvf_geglu_bf16
 *   model/autoencoder.py:392-394   gs_embedding = Linear(14, dim) + LayerNorm(no affine),
 *                  :250-301        position_encoding = PointEmbed(dim) + LayerNorm(no affine),
 *                  :561, :80-81    their sum, then the decoder PreNorm LayerNorm              -> gvf_vae_query_embed_bf16
 * `dtype` = GVF_DT_BF16 / GVF_DT_F16 of gvf_dit.h: the 16-bit type of the GEMM / attention operands these kernels read and write (the
 * reference decodes under accelerate's fp16 autocast, inference_dpm_latent.py:256-257); the `*_bf16*` names are the round-1/2 entry points.
 * Conventions as in gvf_rast.h: device pointers, caller-owned buffers, explicit stream, int status.
 */
#ifndef GVF_VAE_H
#define GVF_VAE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* out bf16 [rows][ld_out] (first F columns) = in[r][c] * gelu_erf(in[r][F + c]);  in bf16 [rows][ld_in], F % 8 == 0,
 * ld_in / ld_out multiples of 8. */
int gvf_geglu(int dtype, const void* in16, int ld_in, void* out16, int ld_out, int64_t rows, int F, void* stream);
int gvf_geglu_bf16(const void* in_bf16, int ld_in, void* out_bf16, int ld_out, int64_t rows, int F, void* stream);

/* out bf16 [P][C] = LN_pre( LN_emb(q W^T + b) + LN_emb(point_embed(q[:, :3])) ), LayerNorms without affine; the two
 * embedding norms use eps_embed (nn.LayerNorm default 1e-5, :393-394), the PreNorm one eps_prenorm (1e-6, :77):
 * queries f32 [P][qdim] (qdim >= 3, <= 16), W f32 [C][qdim], bias f32 [C], omega f32 [C/6];
 * point_embed(p) = concat over axis a of [sin(p_a * omega), cos(p_a * omega)].  C % 6 == 0, C <= 1024.
 * All arithmetic fp32 (the reference's autocast would round the 14 -> C Linear to bf16). */
int gvf_vae_query_embed_bf16(const float* queries, int qdim, const float* W, const float* bias, const float* omega,
                             void* out_bf16, int64_t P, int C, float eps_embed, float eps_prenorm, void* stream);

/* The same embedding for the ENCODER (model/autoencoder.py:520-524: input_embedding(delta) + position_encoding(xyz)):
 * additionally writes the embedding itself, out_embed_f32 [P][C] = LN_emb(q W^T + b) + LN_emb(point_embed(q[:, :3]))
 * (may be null), which the encoder's residual stream starts from; out_bf16 is its PreNorm-normalised GEMM operand.
 * The encoder passes rows [xyz | delta] with zero weights on the xyz columns. */
int gvf_vae_embed(int dtype, const float* queries, int qdim, const float* W, const float* bias, const float* omega,
                  void* out16, float* out_embed_f32, int64_t P, int C, float eps_embed, float eps_prenorm, void* stream);
int gvf_vae_embed_bf16_f32(const float* queries, int qdim, const float* W, const float* bias, const float* omega,
                           void* out_bf16, float* out_embed_f32, int64_t P, int C, float eps_embed, float eps_prenorm,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GVF_VAE_H */
