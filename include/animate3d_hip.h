/*
 * animate3d_hip.h — C-ABI of libanimate3d_hip.so (gfx950 / MI355X).
 *
 * The drop-in boundary of this project is the Python call surface of the reference's
 * `MVUNetMotionModel.forward` (animatediff/models/unet_motion_mv_model.py:633-867); the
 * host side (animate3d_amd/unet.py) mirrors that class.  Everything arithmetic below that
 * surface is executed by the entry points declared here: plain pointers, sizes and a HIP
 * stream, no torch types.  Each entry point names the reference call it replaces.
 *
 * Conventions
 *   - activations / weights: bf16 (uint16_t storage), row-major, last dim contiguous
 *   - bias / norm affine / statistics: fp32
 *   - images are NHWC, flattened to rows: row = ((b*H + y)*W + x), image batch order is
 *     the reference's (b n f) order (unet_motion_mv_model.py:767)
 *   - ld* = leading dimension (elements between consecutive rows)
 *   - every function returns 0 on success, a negative A3D_E* code on bad arguments, or a
 *     positive hipError_t from the launch.  Nothing here allocates or synchronises.
 */
#ifndef ANIMATE3D_HIP_H
#define ANIMATE3D_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* a3d_stream_t; /* hipStream_t */

enum {
  A3D_OK = 0,
  A3D_EINVAL = -1,      /* shape / alignment precondition violated */
  A3D_EUNSUPPORTED = -2 /* e.g. head_dim not in {40, 80, 160} */
};

/* dtype codes for the two layout-conversion entry points */
enum { A3D_F32 = 0, A3D_BF16 = 1, A3D_F16 = 2 };

/* Library / build identification ("animate3d_hip gfx950 <n kernels>"). */
const char* a3d_version(void);

/* ------------------------------------------------------------------------------------
 * Row addressing for the attention kernels.  Sequence position s of attention group g
 * lives at tensor row
 *     row(g, s) = (g / gdiv) * ga + (g % gdiv) * gb + (s / seg_len) * seg_stride + s % seg_len
 * This expresses the reference's einops regroupings as addressing instead of copies:
 *   "(b n f) l c -> (b f) (n l) c"   (attention_processor.py:54,340,557):
 *        g = b*F + f ; gdiv = F, ga = n*F*L, gb = L, seg_len = L, seg_stride = F*L
 *   first-frame K/V (attention_processor.py:389-397): same with gb = 0
 *   text / IP tokens shared by the F frames of a video (unet_motion_mv_model.py:754,763):
 *        g = image index ; gdiv = F, ga = T, gb = 0, seg_len = T
 * ------------------------------------------------------------------------------------ */
typedef struct a3d_rowmap {
  int64_t gdiv, ga, gb;
  int64_t seg_len, seg_stride;
  int64_t ld; /* elements per row of the underlying [rows, ld] tensor */
} a3d_rowmap;

/* Y[M,N] = alpha * (X[M,K] · W[N,K]^T + bias[N] + rowbias[m / rb_div][N]) + beta * R[M,N]
 * Replaces every nn.Linear / 1x1 nn.Conv2d on the path (diffusers Attention.to_q/k/v/out,
 * the processors' to_*_i2v / to_*_sp / to_*_ip Linear layers attention_processor.py:162-166,
 * 322-323,490-493; Transformer2D/Temporal proj_in/out; FeedForward; TimestepEmbedding;
 * ResnetBlock2D.time_emb_proj / conv_shortcut) with the residual add, AlphaBlender mix
 * (attention_processor.py:709) and time-embedding broadcast fused as epilogues.
 * bias, rowbias, R may be NULL.  Requires K % 64 == 0, N % 4 == 0, 16-byte aligned rows.
 * `flags` (also of a3d_gemm_geglu / a3d_conv3x3; 0 = defaults) is a per-call launch parameter — the library keeps no tuning state:
 *   bits 0-7   A3D_GEMM_RESERVED_CUS: compute units the persistent kernel's grid leaves free for concurrently running kernels
 *              (the sharded path passes the CUs an in-flight RCCL all-gather needs; at least 32 CUs are always used);
 *   bits 8-9   kernel choice: 0 = automatic (persistent 256-row-tile kernel for the big token matrices; round 6: the LDS-DMA ring kernel
 *              with 128-row tiles for the small ones — UNet levels 2 / 3 of a multi-GPU rank, the 4D-SDS shape —; 128 x 128 register-staged
 *              tiles for ragged shapes), A3D_GEMM_TILE128 = always the register-staged 128 x 128 kernel, A3D_GEMM_RING = the ring kernel
 *              whenever the shape allows it (a3d_gemm only; falls back to the automatic choice otherwise), A3D_GEMM_DIRECT = the persistent
 *              kernel storing straight from the accumulator layout (a3d_gemm only).  All kernels walk K in the same
 *              order with the same epilogue arithmetic: bit-identical results (A/B measurements and the parity tests).  Other bits must be
 *              zero (A3D_EINVAL). */
#define A3D_GEMM_RESERVED_CUS_MASK 0xff
#define A3D_GEMM_KERNEL_MASK 0x300
#define A3D_GEMM_TILE128 0x100
#define A3D_GEMM_RING 0x200
#define A3D_GEMM_DIRECT 0x300   /* the persistent kernel with its direct (LDS-free) epilogue where the shape allows it: measurement of round 6 */
int a3d_gemm_bf16(a3d_stream_t stream, const void* X, int64_t ldx, const void* W, int64_t ldw,
                  const float* bias, const void* rowbias, int64_t rb_div, const void* R, int64_t ldr,
                  void* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, float alpha, float beta, int flags);

/* a3d_gemm with a caller-owned workspace for SPLIT-K (round 5).  The persistent kernel works on 256-row tiles: at the small token
 * matrices of UNet levels 2 / 3 (M <= 8192 at the 4D-SDS shape, animatemv_guidance.py:339-346) a launch is 32-160 tiles on 256 compute
 * units.  With a workspace the K-tiles of one output tile are dealt to up to 16 work items that leave fp32 partial accumulators in `ws`;
 * a second kernel adds them in index order (deterministic) and applies the epilogue.  Results agree with a3d_gemm to fp32 summation
 * order (NOT bit for bit: callers that compare runs bit for bit use a3d_gemm).  Two calls: first with ws_needed != NULL — nothing is
 * launched, *ws_needed receives the bytes a split launch of exactly this call would use (0: the shape does not split, call a3d_gemm) —
 * then with ws (16-byte aligned, >= that many bytes) and ws_needed == NULL.  The library still never allocates. */
int a3d_gemm_ws_bf16(a3d_stream_t stream, const void* X, int64_t ldx, const void* W, int64_t ldw,
                     const float* bias, const void* rowbias, int64_t rb_div, const void* R, int64_t ldr,
                     void* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, float alpha, float beta, int flags,
                     void* ws, int64_t ws_bytes, int64_t* ws_needed);

/* Y[M,N] = [X | X2] W^T + bias with the A operand in two pieces: contraction columns [0, K1) are read from X (row stride ldx), [K1, K) from
 * X2 (row stride ldx2) — the 1x1 shortcut convolution of an up-block ResnetBlock2D over torch.cat([hidden, skip], dim=1)
 * (unet_motion_mv_model.py:826-827 / diffusers CrossAttnUpBlockMotion) without materialising the concatenation (round 5).  Served by the
 * persistent kernel only: A3D_EUNSUPPORTED (M % 256 != 0, too few tiles, ldx / ldx2 / ldw not multiples of 64) means "concatenate and call
 * a3d_gemm".  K1 % 64 == 0, K % 64 == 0, N % 8 == 0, 16-byte aligned rows.  Bit-identical to a3d_gemm on the concatenated operand. */
int a3d_gemm2_bf16(a3d_stream_t stream, const void* X, int64_t ldx, const void* X2, int64_t ldx2, int64_t K1,
                   const void* W, int64_t ldw, const float* bias, void* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, int flags);

/* Same contraction with an fp32 result: Y[M,N] (float, row stride ldy floats) = alpha * (X W^T + bias).  For logits that must
 * not be rounded to bf16: the single-head 512-wide self-attention of the VAE mid block (diffusers AutoencoderKL, used by
 * pipeline.py:566-579 decode_latents) computes S = Q K^T / sqrt(512) with this, a3d_softmax_rows_f32_bf16, and two more GEMMs.
 * Requires K % 64 == 0, N % 8 == 0, 16-byte aligned rows. */
int a3d_gemm_f32out_bf16(a3d_stream_t stream, const void* X, int64_t ldx, const void* W, int64_t ldw,
                         const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, float alpha);

/* Fused feed-forward input projection + GEGLU (diffusers FeedForward.net[0] = GEGLU: proj, chunk(2), h * gelu(gate)):
 *   Y[M, N2/2][m, j] = (X·Wh^T + bh)[m, j] * gelu_erf((X·Wg^T + bg)[m, j])
 * W / bias rows must be INTERLEAVED in blocks of 32: rows [64b, 64b+32) = h rows [32b, 32b+32),
 * rows [64b+32, 64b+64) = gate rows [32b, 32b+32), so h and gate of one output column land in the same
 * wave tile and the 2x wider intermediate never goes to HBM.  N2 % 64 == 0, K % 64 == 0. */
int a3d_gemm_geglu_bf16(a3d_stream_t stream, const void* X, int64_t ldx, const void* W, int64_t ldw,
                        const float* bias, void* Y, int64_t ldy, int64_t M, int64_t N2, int64_t K, int flags);

/* 3x3 convolution, padding 1, NHWC, as an implicit GEMM (K = 9*Cin):
 *   Y[b, yo, xo, co] = bias[co] + rowbias[(row) / rb_div][co] + R[...]
 *                      + sum_{ky,kx,ci} X[b, yo*stride+ky-1, xo*stride+kx-1, ci] * Wp[co][ky][kx][ci]
 * up2x bit 0 first applies nearest 2x upsampling to X (diffusers Upsample2D) by addressing; bits 1 / 2 crop the upsampled
 * image by its last row / column, which is what F.interpolate(size = (2H-1, 2W-1), mode = "nearest") yields: the forced
 * upsample size of latents that are not multiples of 8 (unet_motion_mv_model.py:690-698, 831-837).
 * Replaces cuDNN conv2d in diffusers ResnetBlock2D.conv1/conv2, Downsample2D, Upsample2D and
 * conv_out (unet_motion_mv_model.py:271,859).  Requires Cin % 64 == 0, Cout % 4 == 0. */
int a3d_conv3x3_bf16(a3d_stream_t stream, const void* X, const void* Wp, const float* bias,
                     const void* rowbias, int64_t rb_div, const void* R, void* Y,
                     int B, int H, int W, int Cin, int Cout, int stride, int up2x, int flags);

/* a3d_conv3x3 with a split-K workspace (see a3d_gemm_ws_bf16): the 8 x 8 and 4 x 4 feature maps of the mid block (K = 9 Cin = 11 520 ... 23 040 over
 * 128 or fewer output tiles; cuDNN under nn.Conv2d picks a split-K algorithm there too).  Work items hold whole 64-channel slices. */
int a3d_conv3x3_ws_bf16(a3d_stream_t stream, const void* X, const void* Wp, const float* bias,
                        const void* rowbias, int64_t rb_div, const void* R, void* Y,
                        int B, int H, int W, int Cin, int Cout, int stride, int up2x, int flags,
                        void* ws, int64_t ws_bytes, int64_t* ws_needed);

/* softmax(Q K^T * scale) V per (group, head), flash-style (no score matrix in memory).
 *   O[row_o(g,s), h*D + d] = out_scale * attn(...)   (+ previous O contents if accumulate)
 * Replaces xformers.ops.memory_efficient_attention at attention_processor.py:103,233,268,
 * 405,416,656.  head_dim in {40, 80, 160} (the SD1.5 UNet levels) and 64 (CLIP text tower of pipeline.py:345-524).  K and V share
 * kmap.  `accumulate` is a per-call flag word: bit 0 (A3D_ATTN_ACCUMULATE): add to the previous O contents; bit 1
 * (A3D_ATTN_CAUSAL): causal mask (key s visible to queries >= s of its group; head_dim 64 / 160 only) — transformers'
 * CLIPTextModel causal attention; bit 2 (A3D_ATTN_EXACT): skip the max-free first pass of the LDS-DMA staged kernels (head_dim
 * 40 / 80, long aligned K/V) and run their exact running-maximum pass directly — the pass a workgroup otherwise re-runs after an
 * overflow; bit 3 (A3D_ATTN_PLAIN): use the generic kernel that serves the short / ragged shapes (parity tests, A/B timing).
 * Results agree to rounding in every mode. */
#define A3D_ATTN_ACCUMULATE 1
#define A3D_ATTN_CAUSAL 2
#define A3D_ATTN_EXACT 4
#define A3D_ATTN_PLAIN 8
int a3d_flash_attn_bf16(a3d_stream_t stream, const void* Q, const void* K, const void* V, void* O,
                        const a3d_rowmap* qmap, const a3d_rowmap* kmap, const a3d_rowmap* omap,
                        int groups, int heads, int head_dim, int64_t q_len, int64_t kv_len,
                        float scale, float out_scale, int accumulate);

/* a3d_flash_attn plus diagnostics of the LDS-DMA staged kernels (head_dim 40 / 80 on long aligned K/V): `counters` = 3 caller-zeroed
 * device words; [0] += workgroups sent straight to the exact pass by the fp16 spread vote, [1] += workgroups that discarded a max-free
 * result and re-ran exactly (overflow), [2] += workgroups launched.  Launches of the other kernels leave the words alone.  The reference has
 * no counterpart (xformers.ops.memory_efficient_attention, attention_processor.py:405,416,656, always keeps a running maximum); this
 * exists so that parity tests on peaked softmax inputs can assert how much of a launch left the fast path.  Same results as a3d_flash_attn. */
int a3d_flash_attn_counted_bf16(a3d_stream_t stream, const void* Q, const void* K, const void* V, void* O,
                                const a3d_rowmap* qmap, const a3d_rowmap* kmap, const a3d_rowmap* omap,
                                int groups, int heads, int head_dim, int64_t q_len, int64_t kv_len,
                                float scale, float out_scale, int accumulate, unsigned int* counters);

/* Two key sets in one launch, each with its own softmax:
 *   O = out_scale * attn(Q, K, V) + out_scale2 * attn(Q, K2, V2)   (+ previous O contents if accumulate)
 * Replaces the text-token attention, the per-adapter image-token attention and `hidden_states = hidden_states + scale * ip` of the
 * IPAdapter processor (attention_processor.py:233, 254-283) with ONE pass over Q and O.  head_dim 40 or 80 (A3D_EUNSUPPORTED otherwise:
 * the caller then issues two a3d_flash_attn calls, the second with accumulate). */
int a3d_flash_attn2_bf16(a3d_stream_t stream, const void* Q, const void* K, const void* V, const void* K2, const void* V2, void* O,
                         const a3d_rowmap* qmap, const a3d_rowmap* kmap, const a3d_rowmap* kmap2, const a3d_rowmap* omap,
                         int groups, int heads, int head_dim, int64_t q_len, int64_t kv_len, int64_t kv_len2,
                         float scale, float out_scale, float out_scale2, int accumulate);

/* Temporal (AnimateDiff) self-attention over the F frames of every (video, pixel, head):
 * rows of Q/K/V/O are ((v*F + f)*L + l).  Replaces the unfused
 * get_attention_scores + torch.bmm of attention_processor.py:630-636.  frames <= 32. */
int a3d_temporal_attn_bf16(a3d_stream_t stream, const void* Q, const void* K, const void* V, int64_t ldqkv,
                           void* O, int64_t ldo, int videos, int frames, int64_t L, int heads,
                           int head_dim, float scale);

/* The same attention when the FRAMES of a video are sharded over ranks (frame-parallel denoise step): this
 * rank holds queries (and writes outputs) only for frames [q_f0, q_f0 + q_frames), rows ((v*q_frames + f - q_f0)*L + l); K / V
 * are the all-gathered tensors, kv_frames_per_block frames per rank block: frame f of video v sits at row
 * (f / kv_frames_per_block) * kv_block_stride + (v*kv_frames_per_block + f % kv_frames_per_block)*L + l.
 * Per-query arithmetic is that of a3d_temporal_attn_bf16 (reference: attention_processor.py:630-636 over all F frames). */
int a3d_temporal_attn_sharded_bf16(a3d_stream_t stream, const void* Q, int64_t ldq, const void* K, const void* V, int64_t ldkv,
                                   void* O, int64_t ldo, int videos, int frames, int64_t L, int heads,
                                   int head_dim, float scale, int q_f0, int q_frames, int kv_frames_per_block,
                                   int64_t kv_block_stride);

/* GroupNorm over `rows` consecutive rows x (C/groups) channels per instance, optional SiLU.
 * 2-D GroupNorm: B = images, rows = H*W.  The motion module's 3-D GroupNorm over
 * (C/32, F, H, W) (diffusers TransformerTemporalModel.norm): B = videos, rows = F*H*W.
 * ws: fp32 scratch of a3d_group_norm_ws_floats(B, rows, groups) elements. */
int64_t a3d_group_norm_ws_floats(int B, int64_t rows, int groups);
int a3d_group_norm_bf16(a3d_stream_t stream, const void* X, void* Y, const float* gamma, const float* beta,
                        float* ws, int B, int64_t rows, int C, int groups, float eps, int silu);

/* a3d_group_norm_bf16 over a TWO-SOURCE input: channels [0, Ca) of a row come from Xa ([B*rows, Ca]), channels [Ca, Ca + Cb) from Xb
 * ([B*rows, Cb]); Y is the normalised [B*rows, Ca + Cb] tensor.  norm1 of an up-block ResnetBlock2D over torch.cat([hidden, skip], dim=1)
 * (diffusers CrossAttnUpBlockMotion / UpBlockMotion, unet_motion_mv_model.py:826-827) without the concatenation pass (round 5).  A group may
 * straddle the two sources (1280 + 640 channels in 32 groups of 60); Ca % 8 == 0.  ws as for a3d_group_norm_bf16 with C = Ca + Cb. */
int a3d_group_norm2_bf16(a3d_stream_t stream, const void* Xa, int Ca, const void* Xb, int Cb, void* Y, const float* gamma,
                         const float* beta, float* ws, int B, int64_t rows, int groups, float eps, int silu);

/* The two halves of a3d_group_norm_bf16 for a norm whose instance is spread over ranks (the motion module's 3-D GroupNorm of
 * diffusers TransformerTemporalModel.norm under frame sharding): `sums` receives fp64 [B][groups][2] = (sum, sum of squares)
 * of this rank's rows; the caller all-reduces them over the frame group, forms (mean, rstd) as fp32 [B][groups][2] and hands
 * them to the apply half.  ws as above. */
int a3d_group_norm_sums_bf16(a3d_stream_t stream, const void* X, float* ws, double* sums, int B, int64_t rows, int C, int groups);
int a3d_group_norm_apply_bf16(a3d_stream_t stream, const void* X, void* Y, const float* gamma, const float* beta,
                              const float* stats, int B, int64_t rows, int C, int groups, int silu);

/* LayerNorm over C, with the positional-embedding adds of the motion-module processor fused:
 *   Y1[m] = LN(X[m]) + pe1[(m / pe1_div) % pe1_mod]     (pe1 may be NULL)
 *   Y2[m] = LN(X[m]) + pe2[(m / pe2_div) % pe2_mod]     (Y2 may be NULL)
 * (time_pos_embed attention_processor.py:583-584; SinePositionalEncoding2D :561-563). */
int a3d_layer_norm_bf16(a3d_stream_t stream, const void* X, void* Y1, void* Y2, const float* gamma,
                        const float* beta, int64_t M, int C, float eps,
                        const void* pe1, int64_t pe1_div, int64_t pe1_mod,
                        const void* pe2, int64_t pe2_div, int64_t pe2_mod);

/* Y[m, j] = X[m, j] * gelu_erf(X[m, N + j])   (diffusers GEGLU). N % 8 == 0. */
int a3d_geglu_bf16(a3d_stream_t stream, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t M, int64_t N);

/* Y = X * sigmoid(X), n elements (n % 8 == 0). */
int a3d_silu_bf16(a3d_stream_t stream, const void* X, void* Y, int64_t n);

/* Elementwise activation on n elements (n % 8 == 0): mode 0 SiLU, 1 QuickGELU x * sigmoid(1.702 x) (transformers CLIPTextModel of
 * pipeline.py:345-524 encode_prompt), 2 exact GELU (CLIP ViT-H image encoder of pipeline.py:527-538 encode_image). */
int a3d_activation_bf16(a3d_stream_t stream, const void* X, void* Y, int64_t n, int mode);

/* Y[m] = [A[m, 0:Ca] | B[m, 0:Cb]]  (torch.cat([x, skip], dim=1) of the up blocks). */
int a3d_concat_bf16(a3d_stream_t stream, const void* A, int64_t Ca, const void* Bsrc, int64_t Cb, void* Y, int64_t M);

/* diffusers Timesteps(dim, flip_sin_to_cos=True, shift 0): Y[v] = [cos(t_v f_j) | sin(t_v f_j)],
 * fp32 math, bf16 out (unet_motion_mv_model.py:723-728). */
int a3d_timestep_embed_bf16(a3d_stream_t stream, const float* t, void* Y, int V, int dim);

/* sample [V, C, F, H, W] (dtype code) -> im2col rows [(V F) H W, 64] bf16 with
 * k = (ky*3 + kx)*C + c for k < 9*C, zero above (3x3, pad 1).  Folds the
 * permute/reshape of unet_motion_mv_model.py:767 and conv_in's patch gather. 9*C <= 64. */
int a3d_im2col_in(a3d_stream_t stream, const void* sample, int dtype, void* Y, int V, int C, int F, int H, int W);

/* rows [(V F) H W, C] bf16 -> [V, C, F, H, W] in `dtype` (unet_motion_mv_model.py:862). */
int a3d_unpack_out(a3d_stream_t stream, const void* X, void* Y, int dtype, int V, int C, int F, int H, int W);

/* Row softmax of fp32 logits X[M, N] (row stride ldx floats) -> bf16 probabilities Y[M, N] (row stride ldy elements);
 * the softmax of diffusers' AttnProcessor in the VAE mid block.  N % 4 == 0. */
int a3d_softmax_rows_f32_bf16(a3d_stream_t stream, const float* X, int64_t ldx, void* Y, int64_t ldy, int64_t M, int64_t N);

/* Planar fp32 channel mix y[b, o, :] = scale * sum_c w[o, c] x[b, c, :] + bias[o] for <= 8 channels: the VAE's 1x1
 * post_quant_conv together with the 1 / scaling_factor of pipeline.py:567 (decode_latents). */
int a3d_channel_mix_f32(a3d_stream_t stream, const float* X, const float* W, const float* bias, float* Y,
                        int B, int Cin, int Cout, int64_t HW, float scale);

/* CFG combine + DDIM step + first-frame re-pin, one elementwise pass (pipeline.py:1023-1031):
 *   eps = e_uncond + s (e_text - e_uncond) ; x0 = (x - sqrt(1-a_t) eps) / sqrt(a_t)
 *   x_prev = sqrt(a_prev) x0 + sqrt(1 - a_prev) eps ; frame 0 <- first_frame
 * fp32 tensors [n, C, F, H, W]; eps_pair is [2n, C, F, H, W] in (uncond, text) order. */
int a3d_cfg_ddim_step_f32(a3d_stream_t stream, const float* eps_pair, const float* x, const float* first_frame,
                          float* x_prev, int64_t n, int C, int F, int64_t HW, float guidance,
                          float alpha_t, float alpha_prev);


/* ---------------------------------------------------------------------------------------------------------------------
 * Training path: the backward kernels behind `loss.backward()` of train.py:576-590 and the fused
 * optimiser step of train.py:583-596.  The matrix products of the backward pass (dgrad = dY W, wgrad = dY^T X) reuse
 * a3d_gemm / a3d_conv3x3 on transposed operands; what is declared here is everything else.
 * --------------------------------------------------------------------------------------------------------------------- */

/* Flash attention backward (the xformers memory_efficient_attention backward of attention_processor.py:103-105, 233-235, 405-418,
 * 656-658, 691-693).  Same row maps as a3d_flash_attn; dO rows through `domap`, dQ through `dqmap`, dK / dV through `dkmap`
 * (group index of the K/V maps: the first of the `q_per_kv` consecutive query groups that read that K/V, e.g. the F frames of a
 * video in the first-frame branch; groups % q_per_kv == 0).  dO is the gradient of the attention output scaled by `do_scale`
 * (the forward's out_scale).  lse2 / delta: fp32 scratch [groups * heads * q_len] each (log2-sum-exp and sum_k P dP per query
 * row, recomputed here: the forward keeps nothing).  dQ may be NULL (no query gradient wanted) or dK == dV == NULL (keys / values
 * of frozen text / image tokens); bit 0 of accumulate adds into dQ / dK / dV, bit 1: see a3d_flash_attn_lse below.  head_dim 40 / 64 / 80 / 160. */
int a3d_flash_attn_bwd_bf16(a3d_stream_t stream, const void* Q, const void* K, const void* V, const void* dO,
                            void* dQ, void* dK, void* dV, float* lse2, float* delta,
                            const a3d_rowmap* qmap, const a3d_rowmap* kmap, const a3d_rowmap* domap,
                            const a3d_rowmap* dqmap, const a3d_rowmap* dkmap,
                            int groups, int heads, int head_dim, int64_t q_len, int64_t kv_len, int q_per_kv,
                            float scale, float do_scale, int accumulate);

/* The statistics of the backward without its statistics pass (a third of the attention backward's time at head_dim 40), for callers that
 * keep the forward's output: a3d_flash_attn_lse is a3d_flash_attn that also returns lse2 [groups][heads][q_len] (log2 of the softmax
 * denominator per query, out of the kernel's own row sums: one float store per query and head; xformers keeps the same tensor for its
 * backward); a3d_attn_delta computes delta[g][h][q] = sum_d dO[q][h, d] * O[q][h, d] (dO rows through `domap`, O rows through `omap`,
 * O = that forward's un-accumulated output).  a3d_flash_attn_bwd with bit 1 of `accumulate` set then reads lse2 / delta instead of
 * recomputing them. */
int a3d_flash_attn_lse_bf16(a3d_stream_t stream, const void* Q, const void* K, const void* V, void* O,
                            const a3d_rowmap* qmap, const a3d_rowmap* kmap, const a3d_rowmap* omap,
                            int groups, int heads, int head_dim, int64_t q_len, int64_t kv_len,
                            float scale, float out_scale, int accumulate, float* lse2);
int a3d_attn_delta_bf16(a3d_stream_t stream, const void* dO, const void* O, const a3d_rowmap* domap, const a3d_rowmap* omap,
                        float* delta, int groups, int heads, int head_dim, int64_t q_len);

/* Temporal attention backward (attention_processor.py:630-636 under autograd): rows ((v*F + f)*L + l); Q / K / V share the row
 * stride ldqkv, dQ / dK / dV share ldd (e.g. the three column ranges of one [rows, 3C] gradient buffer).  frames <= 32. */
int a3d_temporal_attn_bwd_bf16(a3d_stream_t stream, const void* Q, const void* K, const void* V, int64_t ldqkv,
                               const void* dO, int64_t lddo, void* dQ, void* dK, void* dV, int64_t ldd,
                               int videos, int frames, int64_t L, int heads, int head_dim, float scale);

/* LayerNorm backward (diffusers BasicTransformerBlock.norm1/2/3): dX [M, C]; dgamma / dbeta fp32 [C] (both or neither;
 * accumulate == 0 zeroes them first; fp32 atomics: run-to-run equal to rounding).  C <= 2048; C = 320 / 640 / 1280 with 16-byte
 * aligned X, dY, dX, gamma take the 16-byte-access kernel (8 / 16 / 32 lanes per row). */
int a3d_layer_norm_bwd_bf16(a3d_stream_t stream, const void* X, const void* dY, const float* gamma, void* dX,
                            float* dgamma, float* dbeta, int64_t M, int C, float eps, int accumulate);

/* GroupNorm (+ fused SiLU) backward on channel-last [B][rows][C] (ResnetBlock2D.norm1/2, Transformer2DModel.norm, the 3-D norm of
 * TransformerTemporalModel): stats fp32 [B][groups][2] = (mean, rstd) as a3d_group_norm_apply takes them; ws fp32 [B*C*2] scratch;
 * dgamma / dbeta fp32 [C], ADDED to (both or neither).  C % 8 == 0, C <= 8192, X / dY / dX 16-byte aligned. */
int a3d_group_norm_bwd_bf16(a3d_stream_t stream, const void* X, const void* dY, const float* gamma, const float* beta,
                            const float* stats, void* dX, float* ws, float* dgamma, float* dbeta,
                            int B, int64_t rows, int C, int groups, int silu);

/* GEGLU backward on the interleaved projection P [M, 2N] of a3d_gemm_geglu (column blocks [32 h | 32 gate]):
 * dP = [dY * gelu(gate) | dY * h * gelu'(gate)] in the same layout.  N % 32 == 0. */
int a3d_geglu_bwd_bf16(a3d_stream_t stream, const void* P, int64_t ldp, const void* dY, int64_t lddy, void* dP, int64_t lddp,
                       int64_t M, int64_t N);

/* Y[c][r] = X[r][c] (r < rows, c < cols), Y[c][rows .. rows_pad) = 0: operands of the weight-gradient GEMM dW = dY^T X, whose
 * contraction (the token rows) must be a multiple of 64, and W^T of the trainable weights for dX = dY W.  16-byte accesses when cols,
 * rows_pad, ldx and ldy are multiples of 8 and both pointers 16-byte aligned; any shape otherwise. */
int a3d_transpose_bf16(a3d_stream_t stream, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t rows, int64_t cols,
                       int64_t rows_pad);

/* Weight gradient of a Linear: dW[N, K] (fp32, row stride lddw floats) (+)= alpha * dY[M, N]^T X[M, K], both operands in their natural
 * row-major layout.  The token axis M is split over the grid (the output is small, the contraction long): partial tiles go to the
 * fp32 workspace `ws` (a3d_wgrad_ws_floats(M, N, K) floats, 16-byte aligned) and a second kernel sums them.  accumulate != 0 adds
 * to dW.  N % 8 == 0, K % 8 == 0, 16-byte aligned rows. */
int64_t a3d_wgrad_ws_floats(int64_t M, int64_t N, int64_t K);
int a3d_wgrad_bf16(a3d_stream_t stream, const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW, int64_t lddw,
                   float* ws, int64_t M, int64_t N, int64_t K, float alpha, int accumulate);

/* out[c] (+)= alpha * sum_r X[r][c]  (bias gradients), fp32 out (fp32 atomics over the row split).  cols % 320 == 0 with ldx % 8 == 0 and
 * a 16-byte aligned X reads 16 bytes per lane; any shape otherwise. */
int a3d_colsum_bf16(a3d_stream_t stream, const void* X, int64_t ldx, int64_t rows, int64_t cols, float* out, float alpha, int accumulate);

/* Y = a X + b Y over n elements (n % 8 == 0): sum of the gradients of two branches. */
int a3d_axpby_bf16(a3d_stream_t stream, const void* X, void* Y, int64_t n, float a, float b);

/* dgrad helper of the stride-2 down-sampling conv: Z [B, H, W, C] = dY [B, Ho, Wo, C] at the even positions, zero elsewhere
 * (Ho = (H-1)/2 + 1); the input gradient then is a stride-1 a3d_conv3x3 of Z with the flipped, transposed weight. */
int a3d_zero_insert2x_bf16(a3d_stream_t stream, const void* dY, void* Z, int B, int H, int W, int C);

/* Backward of the nearest up-sampling in front of the up-sampler conv: dX [B, H, W, C] = sum of the <= 4 positions of
 * dU [B, He, We, C] each input pixel was copied to (He = 2H or 2H-1, likewise We: the forced sizes of unet_motion_mv_model.py:831-837). */
int a3d_upsample2x_bwd_bf16(a3d_stream_t stream, const void* dU, void* dX, int B, int H, int W, int He, int We, int C);

/* Optimiser over ONE flat fp32 buffer holding every trainable parameter (and flat gradient / moment buffers of the same length):
 *   a3d_sqnorm_f32     out[0] (+)= sum g^2
 *   a3d_clip_ctrl_f32  ctrl[0] = inv_loss_scale * min(1, max_norm / (norm + 1e-6)) (torch.nn.utils.clip_grad_norm_, train.py:586,593),
 *                      ctrl[1] = 1 if the norm is not finite (GradScaler: skip the step), ctrl[2] = the unscaled norm
 *   a3d_adamw_f32      torch.optim.AdamW step (decoupled weight decay) on g * ctrl[0], skipped when ctrl[1] != 0; ctrl may be NULL. */
int a3d_sqnorm_f32(a3d_stream_t stream, const float* g, int64_t n, float* out, int accumulate);
int a3d_clip_ctrl_f32(a3d_stream_t stream, const float* sqnorm, float max_norm, float inv_loss_scale, float* ctrl);
int a3d_adamw_f32(a3d_stream_t stream, float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, float bias_corr1, float bias_corr2, const float* ctrl);

/* ---------------------------------------------------------------------------------------------------------------------
 * fp16-storage twins.  Every entry point above that reads or writes 16-bit activations / weights exists a second time with
 * IEEE fp16 as the storage type (same signature, same semantics, same fp32 accumulation / statistics / softmax; the MFMA is
 * v_mfma_f32_32x32x16_f16): the dtype the reference's 4D-SDS caller runs the UNet in (animatemv_guidance.py:339-346 casts
 * the model and every input to fp16; BASELINE.json configs 4 and 5).  Built from the same sources with -DA3D_STORAGE_F16
 * (animate3d_amd/build.py).  a3d_im2col_in_f16 / a3d_unpack_out_f16: the CALLER tensor's dtype is still the `dtype` argument,
 * the activation side is fp16.
 * --------------------------------------------------------------------------------------------------------------------- */
int a3d_gemm_f16(a3d_stream_t stream, const void* X, int64_t ldx, const void* W, int64_t ldw,
                  const float* bias, const void* rowbias, int64_t rb_div, const void* R, int64_t ldr,
                  void* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, float alpha, float beta, int flags);
int a3d_gemm_ws_f16(a3d_stream_t stream, const void* X, int64_t ldx, const void* W, int64_t ldw,
                    const float* bias, const void* rowbias, int64_t rb_div, const void* R, int64_t ldr,
                    void* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, float alpha, float beta, int flags,
                    void* ws, int64_t ws_bytes, int64_t* ws_needed);
int a3d_gemm2_f16(a3d_stream_t stream, const void* X, int64_t ldx, const void* X2, int64_t ldx2, int64_t K1,
                  const void* W, int64_t ldw, const float* bias, void* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, int flags);
int a3d_gemm_f32out_f16(a3d_stream_t stream, const void* X, int64_t ldx, const void* W, int64_t ldw,
                         const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, float alpha);
int a3d_gemm_geglu_f16(a3d_stream_t stream, const void* X, int64_t ldx, const void* W, int64_t ldw,
                        const float* bias, void* Y, int64_t ldy, int64_t M, int64_t N2, int64_t K, int flags);
int a3d_conv3x3_f16(a3d_stream_t stream, const void* X, const void* Wp, const float* bias,
                     const void* rowbias, int64_t rb_div, const void* R, void* Y,
                     int B, int H, int W, int Cin, int Cout, int stride, int up2x, int flags);
int a3d_conv3x3_ws_f16(a3d_stream_t stream, const void* X, const void* Wp, const float* bias,
                       const void* rowbias, int64_t rb_div, const void* R, void* Y,
                       int B, int H, int W, int Cin, int Cout, int stride, int up2x, int flags,
                       void* ws, int64_t ws_bytes, int64_t* ws_needed);
int a3d_flash_attn_f16(a3d_stream_t stream, const void* Q, const void* K, const void* V, void* O,
                        const a3d_rowmap* qmap, const a3d_rowmap* kmap, const a3d_rowmap* omap,
                        int groups, int heads, int head_dim, int64_t q_len, int64_t kv_len,
                        float scale, float out_scale, int accumulate);
int a3d_flash_attn_counted_f16(a3d_stream_t stream, const void* Q, const void* K, const void* V, void* O,
                               const a3d_rowmap* qmap, const a3d_rowmap* kmap, const a3d_rowmap* omap,
                               int groups, int heads, int head_dim, int64_t q_len, int64_t kv_len,
                               float scale, float out_scale, int accumulate, unsigned int* counters);
int a3d_flash_attn2_f16(a3d_stream_t stream, const void* Q, const void* K, const void* V, const void* K2, const void* V2, void* O,
                        const a3d_rowmap* qmap, const a3d_rowmap* kmap, const a3d_rowmap* kmap2, const a3d_rowmap* omap,
                        int groups, int heads, int head_dim, int64_t q_len, int64_t kv_len, int64_t kv_len2,
                        float scale, float out_scale, float out_scale2, int accumulate);
int a3d_temporal_attn_f16(a3d_stream_t stream, const void* Q, const void* K, const void* V, int64_t ldqkv,
                           void* O, int64_t ldo, int videos, int frames, int64_t L, int heads,
                           int head_dim, float scale);
int a3d_temporal_attn_sharded_f16(a3d_stream_t stream, const void* Q, int64_t ldq, const void* K, const void* V, int64_t ldkv,
                                   void* O, int64_t ldo, int videos, int frames, int64_t L, int heads,
                                   int head_dim, float scale, int q_f0, int q_frames, int kv_frames_per_block,
                                   int64_t kv_block_stride);
int a3d_group_norm_f16(a3d_stream_t stream, const void* X, void* Y, const float* gamma, const float* beta,
                        float* ws, int B, int64_t rows, int C, int groups, float eps, int silu);
int a3d_group_norm2_f16(a3d_stream_t stream, const void* Xa, int Ca, const void* Xb, int Cb, void* Y, const float* gamma,
                        const float* beta, float* ws, int B, int64_t rows, int groups, float eps, int silu);
int a3d_group_norm_sums_f16(a3d_stream_t stream, const void* X, float* ws, double* sums, int B, int64_t rows, int C, int groups);
int a3d_group_norm_apply_f16(a3d_stream_t stream, const void* X, void* Y, const float* gamma, const float* beta,
                              const float* stats, int B, int64_t rows, int C, int groups, int silu);
int a3d_layer_norm_f16(a3d_stream_t stream, const void* X, void* Y1, void* Y2, const float* gamma,
                        const float* beta, int64_t M, int C, float eps,
                        const void* pe1, int64_t pe1_div, int64_t pe1_mod,
                        const void* pe2, int64_t pe2_div, int64_t pe2_mod);
int a3d_geglu_f16(a3d_stream_t stream, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t M, int64_t N);
int a3d_silu_f16(a3d_stream_t stream, const void* X, void* Y, int64_t n);
int a3d_activation_f16(a3d_stream_t stream, const void* X, void* Y, int64_t n, int mode);
int a3d_concat_f16(a3d_stream_t stream, const void* A, int64_t Ca, const void* Bsrc, int64_t Cb, void* Y, int64_t M);
int a3d_timestep_embed_f16(a3d_stream_t stream, const float* t, void* Y, int V, int dim);
int a3d_im2col_in_f16(a3d_stream_t stream, const void* sample, int dtype, void* Y, int V, int C, int F, int H, int W);
int a3d_unpack_out_f16(a3d_stream_t stream, const void* X, void* Y, int dtype, int V, int C, int F, int H, int W);
int a3d_softmax_rows_f32_f16(a3d_stream_t stream, const float* X, int64_t ldx, void* Y, int64_t ldy, int64_t M, int64_t N);
int a3d_flash_attn_bwd_f16(a3d_stream_t stream, const void* Q, const void* K, const void* V, const void* dO,
                            void* dQ, void* dK, void* dV, float* lse2, float* delta,
                            const a3d_rowmap* qmap, const a3d_rowmap* kmap, const a3d_rowmap* domap,
                            const a3d_rowmap* dqmap, const a3d_rowmap* dkmap,
                            int groups, int heads, int head_dim, int64_t q_len, int64_t kv_len, int q_per_kv,
                            float scale, float do_scale, int accumulate);
int a3d_flash_attn_lse_f16(a3d_stream_t stream, const void* Q, const void* K, const void* V, void* O,
                           const a3d_rowmap* qmap, const a3d_rowmap* kmap, const a3d_rowmap* omap,
                           int groups, int heads, int head_dim, int64_t q_len, int64_t kv_len,
                           float scale, float out_scale, int accumulate, float* lse2);
int a3d_attn_delta_f16(a3d_stream_t stream, const void* dO, const void* O, const a3d_rowmap* domap, const a3d_rowmap* omap,
                       float* delta, int groups, int heads, int head_dim, int64_t q_len);
int a3d_temporal_attn_bwd_f16(a3d_stream_t stream, const void* Q, const void* K, const void* V, int64_t ldqkv,
                               const void* dO, int64_t lddo, void* dQ, void* dK, void* dV, int64_t ldd,
                               int videos, int frames, int64_t L, int heads, int head_dim, float scale);
int a3d_layer_norm_bwd_f16(a3d_stream_t stream, const void* X, const void* dY, const float* gamma, void* dX,
                            float* dgamma, float* dbeta, int64_t M, int C, float eps, int accumulate);
int a3d_group_norm_bwd_f16(a3d_stream_t stream, const void* X, const void* dY, const float* gamma, const float* beta,
                            const float* stats, void* dX, float* ws, float* dgamma, float* dbeta,
                            int B, int64_t rows, int C, int groups, int silu);
int a3d_geglu_bwd_f16(a3d_stream_t stream, const void* P, int64_t ldp, const void* dY, int64_t lddy, void* dP, int64_t lddp,
                       int64_t M, int64_t N);
int a3d_transpose_f16(a3d_stream_t stream, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t rows, int64_t cols,
                       int64_t rows_pad);
int a3d_colsum_f16(a3d_stream_t stream, const void* X, int64_t ldx, int64_t rows, int64_t cols, float* out, float alpha, int accumulate);
int a3d_axpby_f16(a3d_stream_t stream, const void* X, void* Y, int64_t n, float a, float b);
int a3d_zero_insert2x_f16(a3d_stream_t stream, const void* dY, void* Z, int B, int H, int W, int C);
int a3d_upsample2x_bwd_f16(a3d_stream_t stream, const void* dU, void* dX, int B, int H, int W, int He, int We, int C);
int a3d_wgrad_f16(a3d_stream_t stream, const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW, int64_t lddw,
                   float* ws, int64_t M, int64_t N, int64_t K, float alpha, int accumulate);

#ifdef __cplusplus
}
#endif
#endif /* ANIMATE3D_HIP_H */
