// Temporal (AnimateDiff) self-attention over the F <= 32 frames of every (video, pixel, head).
//
// The reference materialises [(b n h w) heads, F, F] scores with baddbmm + softmax + bmm
// (attention_processor.py:630-636).  ~0.05 % of the step's FLOPs: the kernel is bound by the three reads +
// one write of the token tensors, so it is organised around full-line HBM accesses:
//   * a workgroup owns one pixel x one 320-channel slab (640 B per token row) x all F frames;
//     the Q, K, V slabs are copied to LDS with 16-byte loads (40 lanes cover one 640-B row segment);
//   * scores and outputs per (pixel, head) run on the matrix cores (temporal_attn_mfma_kernel below; the round-1 kernel did
//     ~1 000 v_dot2 / v_fma per thread and reached 2.9-3.2 TB/s).
// Every token byte is read from HBM exactly once (the first version re-read K/V F times through L1).
#include "common.h"


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

// MFMA 16x16x16 (the K = 16 form: its B operand holds 4 k per 16-lane row, which is exactly how the 16x16 result tile of the score
// product leaves a lane — P feeds the second product without any cross-lane movement).  Lane l supplies A[i = l&15][k = 4*(l>>4) .. +3] and
// B[k = 4*(l>>4) .. +3][j = l&15]; result register r of lane l is D[i = 4*(l>>4) + r][j = l&15].
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
A3D_DEV f32x4_t mfma16k16(const u32x2_t& a, const u32x2_t& b, const f32x4_t& c) {
#ifdef A3D_STORAGE_F16
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_t, a), __builtin_bit_cast(f16x4_t, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_t, a), __builtin_bit_cast(s16x4_t, b), c, 0, 0, 0);
#endif
}

// Matrix-core version (round 3): the same staging (every token byte read from HBM once), then per (pixel, head) S^T = K·Q^T and
// O^T = V^T·P^T on v_mfma_f32_16x16x16 — 6 MFMAs per head at 16 frames / head_dim 40 instead of ~1 000 v_dot2 / v_fma per thread, which
// left the kernel at 3.3 TB/s.  One workgroup = one pixel x one 320-channel slab x all frames; a wave owns the heads
// wave, wave + NW, ...; query i and its 4 keys per 16-lane row sit in lane (i, l>>4): row maximum / sum = 3 in-lane ops + two
// lane exchanges (xor 16, xor 32); V^T fragments come out of LDS transposed by ds_read_b64_tr_b16; the normalised O tile
// is parked in the (dead) Q rows of the head and the slab leaves with 16-byte stores.
template <int FP, int D>
__global__ __launch_bounds__(NSL * FP) void temporal_attn_mfma_kernel(const TAParams p) {
  constexpr int ROWB = SLAB + 8;                      // 656-byte rows: the 16 frame rows of one read spread over all banks
  constexpr int FB = FP / 16;                        // 16-frame blocks
  constexpr int KS = (D + 15) / 16;                  // k-steps of the score product = 16-dim blocks of O
  constexpr int NH = SLAB / D;                       // heads in the slab
  constexpr int NT = NSL * FP, NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];   // [3][F][ROWB]
  const int F = p.frames;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t pix = blockIdx.x < p.npix ? (int64_t)blockIdx.x : p.npix - 1;
  const int c0 = blockIdx.y * SLAB;
  const int64_t v = pix / p.L, l = pix % p.L;
  constexpr int chunks = SLAB / 8;

  // ---- stage the Q, K, V slabs: item = (tensor, frame, 16-byte chunk)
  auto src_of = [&](int it) -> const uint16_t* {
    const int ch = it % chunks;
    const int f = (it / chunks) % F;
    const int ten = it / (chunks * F);
    if (ten == 0) {
      int fq = f - p.q_f0;                       // frames this rank has no query for: any valid row
      if (fq < 0 || fq >= p.q_frames) fq = 0;
      return p.Q + ((v * p.q_frames + fq) * p.L + l) * p.ldq + c0 + ch * 8;
    }
    const int64_t row = (int64_t)(f / p.kv_fpr) * p.kv_rs + (v * p.kv_fpr + f % p.kv_fpr) * p.L + l;
    return (ten == 1 ? p.K : p.V) + row * p.ld + c0 + ch * 8;
  };
  auto dst_of = [&](int it) -> uint16_t* {
    const int ch = it % chunks;
    const int f = (it / chunks) % F;
    const int ten = it / (chunks * F);
    return smem + ((size_t)ten * F + f) * ROWB + ch * 8;
  };
  const int items = 3 * F * chunks;
  if (F == FP) {
    constexpr int ITEMS = 3 * FP * chunks, NIT = (ITEMS + NT - 1) / NT;
    u32x4_t stg[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int it = tid + k * NT;
      if (ITEMS % NT == 0 || it < ITEMS) stg[k] = *reinterpret_cast<const u32x4_t*>(src_of(it));
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int it = tid + k * NT;
      if (ITEMS % NT == 0 || it < ITEMS) *reinterpret_cast<u32x4_t*>(dst_of(it)) = stg[k];
    }
  } else {
    for (int it = tid; it < items; it += NT) *reinterpret_cast<u32x4_t*>(dst_of(it)) = *reinterpret_cast<const u32x4_t*>(src_of(it));
  }
  __syncthreads();

  const int i16 = lane & 15, k4 = lane >> 4;
  uint16_t* const Qs = smem;
  const uint16_t* const Ks = smem + (size_t)F * ROWB;
  const uint16_t* const Vs = smem + (size_t)2 * F * ROWB;
  for (int h = wid; h < NH; h += NW) {
    const int hc = h * D;
    // ---- scores, transposed: S^T[key j][query i]; block (jb, ib)
    f32x4_t sT[FB][FB];
    u32x2_t qfr[FB][KS];
#pragma unroll
    for (int ib = 0; ib < FB; ++ib) {
      const int fi = 16 * ib + i16 < F ? 16 * ib + i16 : F - 1;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        u32x2_t w = *reinterpret_cast<const u32x2_t*>(Qs + (size_t)fi * ROWB + hc + 16 * ks + 4 * k4);
        if (16 * ks + 4 * k4 >= D) w = u32x2_t{0u, 0u};          // contraction padding (head_dim 40: dims 40..47)
        qfr[ib][ks] = w;
      }
    }
#pragma unroll
    for (int jb = 0; jb < FB; ++jb) {
      const int fj = 16 * jb + i16 < F ? 16 * jb + i16 : F - 1;
      u32x2_t kfr[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        kfr[ks] = *reinterpret_cast<const u32x2_t*>(Ks + (size_t)fj * ROWB + hc + 16 * ks + 4 * k4);
        if (16 * ks + 4 * k4 >= D) kfr[ks] = u32x2_t{0u, 0u};      // (behind the last head these lanes would read the uninitialised row padding)
      }
#pragma unroll
      for (int ib = 0; ib < FB; ++ib) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = mfma16k16(kfr[ks], qfr[ib][ks], acc);
        sT[jb][ib] = acc;
      }
    }
    // ---- softmax over the keys of query i = 16 ib + i16: this lane holds keys 16 jb + 4 k4 + r
    u32x2_t pfr[FB][FB];
    float inv[FB];
#pragma unroll
    for (int ib = 0; ib < FB; ++ib) {
      float mx = -INFINITY;
#pragma unroll
      for (int jb = 0; jb < FB; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = 16 * jb + 4 * k4 + r < F;
          sT[jb][ib][r] = ok ? sT[jb][ib][r] * p.scale_log2 : -INFINITY;
          mx = fmaxf(mx, sT[jb][ib][r]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = 0.f;
#pragma unroll
      for (int jb = 0; jb < FB; ++jb) {
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { e[r] = __builtin_amdgcn_exp2f(sT[jb][ib][r] - mx); sum += e[r]; }
        pfr[jb][ib] = u32x2_t{pack16(e[0], e[1]), pack16(e[2], e[3])};
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      inv[ib] = 1.f / sum;
    }
    // ---- O^T[d][i] = sum_j V^T[d][j] P^T[j][i] per 16-dim block; parked in the Q rows of this head (dead now)
#pragma unroll
    for (int db = 0; db < KS; ++db) {
      u32x2_t vfr[FB];
#pragma unroll
      for (int jb = 0; jb < FB; ++jb) {
        const int fj = 16 * jb + 4 * k4 + (i16 >> 2) < F ? 16 * jb + 4 * k4 + (i16 >> 2) : F - 1;      // masked keys carry P = 0
        vfr[jb] = lds_tr16_b64(Vs + (size_t)fj * ROWB + hc + 16 * db + 4 * (i16 & 3));
      }
#pragma unroll
      for (int ib = 0; ib < FB; ++ib) {
        f32x4_t o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jb = 0; jb < FB; ++jb) o = mfma16k16(vfr[jb], pfr[jb][ib], o);
        const int fi = 16 * ib + i16;
        if (16 * db + 4 * k4 < D && fi < F) {
          const u32x2_t w = {pack16(o[0] * inv[ib], o[1] * inv[ib]), pack16(o[2] * inv[ib], o[3] * inv[ib])};
          *reinterpret_cast<u32x2_t*>(Qs + (size_t)fi * ROWB + hc + 16 * db + 4 * k4) = w;
        }
      }
    }
  }
  __syncthreads();
  // ---- the O slab of this rank's frames leaves with 16-byte stores
  if ((int64_t)blockIdx.x < p.npix) {
    for (int it = tid; it < p.q_frames * chunks; it += NT) {
      const int ch = it % chunks, fq = it / chunks;
      const u32x4_t w = *reinterpret_cast<const u32x4_t*>(Qs + (size_t)(p.q_f0 + fq) * ROWB + ch * 8);
      *reinterpret_cast<u32x4_t*>(p.O + ((v * p.q_frames + fq) * p.L + l) * p.ldo + c0 + ch * 8) = w;
    }
  }
}

template <int FP, int D>
int launch_mfma(hipStream_t s, const TAParams& p, int C) {
  if (p.npix > 0x7fffffffLL) return A3D_EINVAL;
  const size_t lds = (size_t)3 * p.frames * (SLAB + 8) * sizeof(uint16_t);
  static uint64_t attr_done = 0;
  if (int rc = a3d_once_per_device(attr_done, [] {
        return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_attn_mfma_kernel<FP, D>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 3 * FP * (SLAB + 8) * 2); })) return rc;
  temporal_attn_mfma_kernel<FP, D><<<dim3((unsigned)p.npix, (unsigned)(C / SLAB)), dim3(NSL * FP), lds, s>>>(p);
  return a3d_launch_status();
}

template <int DP>
int launch_dp(hipStream_t s, const TAParams& p, int C) {
  if (p.frames <= 16) return launch_mfma<16, 40 * DP>(s, p, C);
  return launch_mfma<32, 40 * DP>(s, p, C);
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
