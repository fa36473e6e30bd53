// Temporal (AnimateDiff) self-attention over the F <= 32 frames of every (video, pixel, head).
//
// The reference materialises [(b n h w) heads, F, F] scores with baddbmm + softmax + bmm
// (attention_processor.py:630-636).  Here one thread owns one query (video, pixel, head, frame):
// F scores live in registers, K and V rows are read in 16-byte chunks straight from the
// [(v f) l, C] token tensors (row stride L*ld between frames) and are shared through L1/L2 by
// the F threads of the same pixel, which sit in adjacent lanes.  ~0.05 % of the step's FLOPs;
// the kernel is bound by the three reads + one write of the token tensors.
#include "common.h"

namespace {

struct TAParams {
  const uint16_t* Q; const uint16_t* K; const uint16_t* V; int64_t ld;
  uint16_t* O; int64_t ldo;
  int videos, frames; int64_t L; int heads; float scale_log2;
  int64_t total;   // videos * L * frames
};

template <int FMAX, int D>
__global__ __launch_bounds__(256) void temporal_attn_kernel(const TAParams p) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= p.total) return;
  const int F = p.frames;
  const int i = (int)(idx % F);
  const int64_t pix = idx / F;           // v * L + l
  const int64_t v = pix / p.L, l = pix % p.L;
  const int head = blockIdx.y;
  const int64_t row0 = (v * F) * p.L + l;             // frame 0 of this pixel
  const int64_t fstride = p.L * p.ld;                 // elements between frames
  const uint16_t* qp = p.Q + (row0 + (int64_t)i * p.L) * p.ld + head * D;
  const uint16_t* kp = p.K + row0 * p.ld + head * D;
  const uint16_t* vp = p.V + row0 * p.ld + head * D;

  float s[FMAX];
#pragma unroll
  for (int j = 0; j < FMAX; ++j) s[j] = 0.f;
#pragma unroll 1
  for (int c = 0; c < D / 8; ++c) {
    const u32x4_t qv = *reinterpret_cast<const u32x4_t*>(qp + c * 8);
    float qf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[e] = (e & 1) ? hi_bf(qv[e >> 1]) : lo_bf(qv[e >> 1]);
#pragma unroll
    for (int j = 0; j < FMAX; ++j) {
      if (j < F) {
        const u32x4_t kv = *reinterpret_cast<const u32x4_t*>(kp + j * fstride + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[j] = fmaf(qf[e], (e & 1) ? hi_bf(kv[e >> 1]) : lo_bf(kv[e >> 1]), s[j]);
      }
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < FMAX; ++j) if (j < F) mx = fmaxf(mx, s[j]);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < FMAX; ++j) {
    s[j] = (j < F) ? __builtin_amdgcn_exp2f((s[j] - mx) * p.scale_log2) : 0.f;
    sum += s[j];
  }
  const float inv = 1.f / sum;
  uint16_t* op = p.O + (row0 + (int64_t)i * p.L) * p.ldo + head * D;
#pragma unroll 1
  for (int c = 0; c < D / 8; ++c) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
    for (int j = 0; j < FMAX; ++j) {
      if (j < F) {
        const u32x4_t vv = *reinterpret_cast<const u32x4_t*>(vp + j * fstride + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(s[j], (e & 1) ? hi_bf(vv[e >> 1]) : lo_bf(vv[e >> 1]), o[e]);
      }
    }
    u32x4_t ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = pack2bf(o[2 * e] * inv, o[2 * e + 1] * inv);
    *reinterpret_cast<u32x4_t*>(op + c * 8) = ov;
  }
}

template <int D>
int launch_d(hipStream_t s, const TAParams& p) {
  const int64_t nblk = (p.total + 255) / 256;
  if (nblk > 0x7fffffffLL || p.heads > 65535) return A3D_EINVAL;
  const dim3 grid((unsigned)nblk, (unsigned)p.heads), block(256);
  if (p.frames <= 4) temporal_attn_kernel<4, D><<<grid, block, 0, s>>>(p);
  else if (p.frames <= 8) temporal_attn_kernel<8, D><<<grid, block, 0, s>>>(p);
  else if (p.frames <= 16) temporal_attn_kernel<16, D><<<grid, block, 0, s>>>(p);
  else temporal_attn_kernel<32, D><<<grid, block, 0, s>>>(p);
  return a3d_launch_status();
}

}  // namespace

extern "C" int a3d_temporal_attn_bf16(a3d_stream_t stream, const void* Q, const void* K, const void* V, int64_t ldqkv,
                                      void* O, int64_t ldo, int videos, int frames, int64_t L, int heads,
                                      int head_dim, float scale) {
  if (!Q || !K || !V || !O || videos <= 0 || frames <= 0 || frames > 32 || L <= 0 || heads <= 0) return A3D_EINVAL;
  if (ldqkv % 8 || ldo % 8) return A3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(V) |
       reinterpret_cast<uintptr_t>(O)) & 15u) return A3D_EINVAL;
  TAParams p{};
  p.Q = (const uint16_t*)Q; p.K = (const uint16_t*)K; p.V = (const uint16_t*)V; p.ld = ldqkv;
  p.O = (uint16_t*)O; p.ldo = ldo; p.videos = videos; p.frames = frames; p.L = L; p.heads = heads;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.total = (int64_t)videos * L * frames;
  hipStream_t s = (hipStream_t)stream;
  switch (head_dim) {
    case 40: return launch_d<40>(s, p);
    case 80: return launch_d<80>(s, p);
    case 160: return launch_d<160>(s, p);
    default: return A3D_EUNSUPPORTED;
  }
}
