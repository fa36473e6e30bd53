// Temporal (AnimateDiff) self-attention over the F <= 32 frames of every (video, pixel, head).
//
// The reference materialises [(b n h w) heads, F, F] scores with baddbmm + softmax + bmm
// (attention_processor.py:630-636).  ~0.05 % of the step's FLOPs: the kernel is bound by the three reads +
// one write of the token tensors, so it is organised around full-line HBM accesses:
//   * a workgroup owns PIX consecutive pixels x one 320-channel slab (640 B per token row) x all F frames;
//     the Q, K, V slabs are copied to LDS with 16-byte loads (40 lanes cover one 640-B row segment);
//   * one thread owns one query (pixel, 40-dim slice, frame i): scores against the F keys with
//     v_dot2c_f32_bf16 on packed bf16 pairs (keys are LDS broadcasts across the query lanes), slices of one
//     head (D = 80 / 160) are summed with lane shuffles, softmax in registers, then P·V for the thread's own
//     40-dim slice and five 16-byte stores.
// Every token byte is read from HBM exactly once (the first version re-read K/V F times through L1).
#include "common.h"

#ifdef A3D_STORAGE_F16
extern int g_a3d_ta_pix;     // one set of tuning knobs for both builds (defined in the bf16 object)
#else
int g_a3d_ta_pix = 1;        // pixels per workgroup at <= 16 frames (a3d_tune_flash(10 + pix), diagnostics; 1 measured fastest)
#endif

namespace {

constexpr int SLAB = 320;            // channels per workgroup (one 640-byte row segment)
constexpr int SL = 40;               // dims per thread slice
constexpr int NSL = SLAB / SL;       // 8 slices per pixel

struct TAParams {
  const uint16_t* Q; const uint16_t* K; const uint16_t* V; int64_t ld;      // ld: row stride of K and V
  int64_t ldq;       // row stride of Q
  uint16_t* O; int64_t ldo;
  int videos, frames; int64_t L; int heads; float scale_log2;
  int64_t npix;      // videos * L
  // frame-sharded call (a3d_temporal_attn_sharded_bf16): Q / O hold only frames [q_f0, q_f0 + q_frames) of every video, rows
  // ((v*q_frames + f - q_f0)*L + l); K / V hold all frames as kv_fpr frames per rank block: frame f of video v at row
  // (f / kv_fpr) * kv_rs + (v*kv_fpr + f % kv_fpr)*L + l.  Unsharded: q_f0 = 0, q_frames = kv_fpr = frames, kv_rs = 0.
  int q_f0, q_frames, kv_fpr; int64_t kv_rs;
};

A3D_DEV float dot2(uint32_t a, uint32_t b, float c) {
#ifdef A3D_STORAGE_F16
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2_t, a), __builtin_bit_cast(h16x2_t, b), c, false);          // v_dot2_f32_f16
#else
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(h16x2_t, a), __builtin_bit_cast(h16x2_t, b), c, false);   // v_dot2c_f32_bf16
#endif
}

// FP = frames padded to 16 or 32 (register array size), PIX = pixels per workgroup, DP = 40-dim slices per head
template <int FP, int PIX, int DP>
__global__ __launch_bounds__(PIX * NSL * FP) void temporal_attn_kernel(const TAParams p) {
  constexpr int ROWB = PIX * SLAB + 8;               // LDS elements per frame (+16 B: de-alias the frame stride)
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];   // [3][F][ROWB]
  const int F = p.frames;
  const int tid = threadIdx.x;
  const int64_t pix0 = (int64_t)blockIdx.x * PIX;
  const int c0 = blockIdx.y * SLAB;

  // ---- stage Q, K, V slabs: item = (tensor, frame, pixel, 16-byte chunk).  When frames == FP (the model's 16) the trip
  //      count and every divisor are compile-time: all loads are issued back to back, then all LDS writes
  constexpr int chunks = SLAB / 8;                       // 40 chunks per row segment
  auto src_of = [&](int it, int Fd) -> const uint16_t* {
    const int ch = it % chunks;
    const int px = (it / chunks) % PIX;
    const int f = (it / (chunks * PIX)) % Fd;
    const int ten = it / (chunks * PIX * Fd);
    int64_t pix = pix0 + px;
    if (pix >= p.npix) pix = p.npix - 1;
    const int64_t v = pix / p.L, l = pix % p.L;
    if (ten == 0) {
      int fq = f - p.q_f0;                       // frames this rank has no query for: any valid row (the lanes are idle later)
      if (fq < 0 || fq >= p.q_frames) fq = 0;
      return p.Q + ((v * p.q_frames + fq) * p.L + l) * p.ldq + c0 + ch * 8;
    }
    const int64_t row = (int64_t)(f / p.kv_fpr) * p.kv_rs + (v * p.kv_fpr + f % p.kv_fpr) * p.L + l;
    return (ten == 1 ? p.K : p.V) + row * p.ld + c0 + ch * 8;
  };
  auto dst_of = [&](int it, int Fd) -> uint16_t* {
    const int ch = it % chunks;
    const int px = (it / chunks) % PIX;
    const int f = (it / (chunks * PIX)) % Fd;
    const int ten = it / (chunks * PIX * Fd);
    return smem + ((size_t)ten * Fd + f) * ROWB + px * SLAB + ch * 8;
  };
  if (F == FP) {
    constexpr int NT = PIX * NSL * FP, ITEMS = 3 * FP * PIX * chunks, NIT = (ITEMS + NT - 1) / NT;
    u32x4_t stg[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int it = tid + k * NT;
      if (ITEMS % NT == 0 || it < ITEMS) stg[k] = *reinterpret_cast<const u32x4_t*>(src_of(it, FP));
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int it = tid + k * NT;
      if (ITEMS % NT == 0 || it < ITEMS) *reinterpret_cast<u32x4_t*>(dst_of(it, FP)) = stg[k];
    }
  } else {
    const int items = 3 * F * PIX * chunks;
    for (int it = tid; it < items; it += blockDim.x) *reinterpret_cast<u32x4_t*>(dst_of(it, F)) = *reinterpret_cast<const u32x4_t*>(src_of(it, F));
  }
  __syncthreads();

  // ---- one thread = one query (pixel px, slice sl, frame i)
  const int sl = tid % NSL;            // slices of one head are adjacent lanes (shuffle partners xor 1, xor 2)
  const int i = (tid / NSL) % FP;
  const int px = tid / (FP * NSL);
  const bool active = i >= p.q_f0 && i < p.q_f0 + p.q_frames && pix0 + px < p.npix;
  const int fi = i < F ? i : F - 1;
  const uint16_t* qrow = smem + ((size_t)0 * F + fi) * ROWB + px * SLAB + sl * SL;
  const uint16_t* kbase = smem + ((size_t)1 * F) * ROWB + px * SLAB + sl * SL;
  const uint16_t* vbase = smem + ((size_t)2 * F) * ROWB + px * SLAB + sl * SL;
  uint32_t q[SL / 2];
#pragma unroll
  for (int c = 0; c < SL / 8; ++c) {
    const u32x4_t w = *reinterpret_cast<const u32x4_t*>(qrow + c * 8);
    q[4 * c] = w[0]; q[4 * c + 1] = w[1]; q[4 * c + 2] = w[2]; q[4 * c + 3] = w[3];
  }
  float s[FP];
#pragma unroll
  for (int j = 0; j < FP; ++j) {
    float acc = 0.f;
    if (j < F) {
#pragma unroll
      for (int c = 0; c < SL / 8; ++c) {
        const u32x4_t w = *reinterpret_cast<const u32x4_t*>(kbase + (size_t)j * ROWB + c * 8);
        acc = dot2(q[4 * c], w[0], acc); acc = dot2(q[4 * c + 1], w[1], acc);
        acc = dot2(q[4 * c + 2], w[2], acc); acc = dot2(q[4 * c + 3], w[3], acc);
      }
    }
    s[j] = acc;
  }
  // slices of the same head are adjacent lanes: sum the partial dot products
  if constexpr (DP >= 2) {
#pragma unroll
    for (int j = 0; j < FP; ++j) s[j] += __shfl_xor(s[j], 1);
  }
  if constexpr (DP >= 4) {
#pragma unroll
    for (int j = 0; j < FP; ++j) s[j] += __shfl_xor(s[j], 2);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < FP; ++j) if (j < F) mx = fmaxf(mx, s[j]);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < FP; ++j) {
    s[j] = (j < F) ? __builtin_amdgcn_exp2f((s[j] - mx) * p.scale_log2) : 0.f;
    sum += s[j];
  }
  const float inv = 1.f / sum;
  float o[SL];
#pragma unroll
  for (int d = 0; d < SL; ++d) o[d] = 0.f;
#pragma unroll
  for (int j = 0; j < FP; ++j) {
    if (j < F) {
      const float pj = s[j];
#pragma unroll
      for (int c = 0; c < SL / 8; ++c) {
        const u32x4_t w = *reinterpret_cast<const u32x4_t*>(vbase + (size_t)j * ROWB + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[8 * c + 2 * e] = fmaf(pj, lo16(w[e]), o[8 * c + 2 * e]);
          o[8 * c + 2 * e + 1] = fmaf(pj, hi16(w[e]), o[8 * c + 2 * e + 1]);
        }
      }
    }
  }
  if (active) {
    const int64_t pix = pix0 + px;
    const int64_t v = pix / p.L, l = pix % p.L;
    uint16_t* dst = p.O + ((v * p.q_frames + (i - p.q_f0)) * p.L + l) * p.ldo + c0 + sl * SL;
#pragma unroll
    for (int c = 0; c < SL / 8; ++c) {
      u32x4_t ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = pack16(o[8 * c + 2 * e] * inv, o[8 * c + 2 * e + 1] * inv);
      *reinterpret_cast<u32x4_t*>(dst + c * 8) = ov;
    }
  }
}

template <int FP, int PIX, int DP>
int launch(hipStream_t s, const TAParams& p, int C) {
  const int64_t nblk = (p.npix + PIX - 1) / PIX;
  if (nblk > 0x7fffffffLL) return A3D_EINVAL;
  const size_t lds = (size_t)3 * p.frames * (PIX * SLAB + 8) * sizeof(uint16_t);
  static uint64_t attr_done = 0;
  if (int rc = a3d_once_per_device(attr_done, [] {
        return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_attn_kernel<FP, PIX, DP>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 3 * FP * (PIX * SLAB + 8) * 2); })) return rc;
  temporal_attn_kernel<FP, PIX, DP><<<dim3((unsigned)nblk, (unsigned)(C / SLAB)), dim3(PIX * NSL * FP), lds, s>>>(p);
  return a3d_launch_status();
}

template <int DP>
int launch_dp(hipStream_t s, const TAParams& p, int C) {
  // 1 pixel per workgroup (128 threads, <= 31 KB LDS, 5 workgroups per CU) measured 22-28 % faster than 2 pixels: more
  // independent load / compute phases in flight per CU (profiles/r1_microbench_gemm_conv_misc.log)
  if (p.frames <= 16 && g_a3d_ta_pix == 2) return launch<16, 2, DP>(s, p, C);   // 256 threads, <= 62 KB LDS
  if (p.frames <= 16) return launch<16, 1, DP>(s, p, C);
  return launch<32, 1, DP>(s, p, C);                          // 256 threads, <= 62 KB LDS
}

}  // namespace

static int temporal_attn_launch(a3d_stream_t stream, const void* Q, int64_t ldq, const void* K, const void* V, int64_t ldqkv, void* O, int64_t ldo,
                                int videos, int frames, int64_t L, int heads, int head_dim, float scale,
                                int q_f0, int q_frames, int kv_fpr, int64_t kv_rs) {
  if (!Q || !K || !V || !O || videos <= 0 || frames <= 0 || frames > 32 || L <= 0 || heads <= 0) return A3D_EINVAL;
  if (q_f0 < 0 || q_frames <= 0 || q_f0 + q_frames > frames || kv_fpr <= 0 || frames % kv_fpr != 0 || kv_rs < 0) return A3D_EINVAL;
  if (ldqkv % 8 || ldo % 8 || ldq % 8) return A3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(V) |
       reinterpret_cast<uintptr_t>(O)) & 15u) return A3D_EINVAL;
  const int C = heads * head_dim;
  if (C % SLAB != 0) return A3D_EUNSUPPORTED;
  TAParams p{};
  p.Q = (const uint16_t*)Q; p.K = (const uint16_t*)K; p.V = (const uint16_t*)V; p.ld = ldqkv; p.ldq = ldq;
  p.O = (uint16_t*)O; p.ldo = ldo; p.videos = videos; p.frames = frames; p.L = L; p.heads = heads;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.npix = (int64_t)videos * L;
  p.q_f0 = q_f0; p.q_frames = q_frames; p.kv_fpr = kv_fpr; p.kv_rs = kv_rs;
  hipStream_t s = (hipStream_t)stream;
  switch (head_dim) {
    case 40: return launch_dp<1>(s, p, C);
    case 80: return launch_dp<2>(s, p, C);
    case 160: return launch_dp<4>(s, p, C);
    default: return A3D_EUNSUPPORTED;
  }
}

extern "C" int A3D_FN(a3d_temporal_attn)(a3d_stream_t stream, const void* Q, const void* K, const void* V, int64_t ldqkv,
                                      void* O, int64_t ldo, int videos, int frames, int64_t L, int heads,
                                      int head_dim, float scale) {
  return temporal_attn_launch(stream, Q, ldqkv, K, V, ldqkv, O, ldo, videos, frames, L, heads, head_dim, scale, 0, frames, frames, 0);
}

extern "C" int A3D_FN(a3d_temporal_attn_sharded)(a3d_stream_t stream, const void* Q, int64_t ldq, const void* K, const void* V, int64_t ldkv,
                                              void* O, int64_t ldo, int videos, int frames, int64_t L, int heads,
                                              int head_dim, float scale, int q_f0, int q_frames, int kv_frames_per_block,
                                              int64_t kv_block_stride) {
  return temporal_attn_launch(stream, Q, ldq, K, V, ldkv, O, ldo, videos, frames, L, heads, head_dim, scale, q_f0, q_frames,
                              kv_frames_per_block, kv_block_stride);
}
