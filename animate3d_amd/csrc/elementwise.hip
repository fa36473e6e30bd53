// Memory-bound helpers of the denoise step (gfx950): GEGLU, SiLU, channel concat, timestep
// sinusoid, input im2col / output unpack (layout folds), CFG + DDIM epilogue.
// All use 16-byte accesses and grid-stride loops capped at 2048 workgroups.
#include "common.h"

namespace {

inline unsigned grid_for(int64_t items, int block = 256) {
  int64_t g = (items + block - 1) / block;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (unsigned)g;
}

__global__ void geglu_kernel(const uint16_t* X, int64_t ldx, uint16_t* Y, int64_t ldy, int64_t M, int64_t N8) {
  const int64_t total = M * N8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / N8, c = i % N8;
    const u32x4_t h = *reinterpret_cast<const u32x4_t*>(X + m * ldx + c * 8);
    const u32x4_t gt = *reinterpret_cast<const u32x4_t*>(X + m * ldx + N8 * 8 + c * 8);
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float hv = (j & 1) ? hi16(h[j >> 1]) : lo16(h[j >> 1]);
      const float gv = (j & 1) ? hi16(gt[j >> 1]) : lo16(gt[j >> 1]);
      y[j] = hv * (0.5f * gv * (1.f + erff(gv * 0.70710678118654752f)));
    }
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack16(y[2 * j], y[2 * j + 1]);
    *reinterpret_cast<u32x4_t*>(Y + m * ldy + c * 8) = o;
  }
}

__global__ void silu_kernel(const uint16_t* X, uint16_t* Y, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(X + i * 8);
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = lo16(v[j]), b = hi16(v[j]);
      o[j] = pack16(a / (1.f + __expf(-a)), b / (1.f + __expf(-b)));
    }
    *reinterpret_cast<u32x4_t*>(Y + i * 8) = o;
  }
}

// mode 1: QuickGELU x * sigmoid(1.702 x) (CLIP text tower); mode 2: exact GELU (erf; CLIP ViT-H vision tower)
template <int MODE>
__global__ void act_kernel(const uint16_t* X, uint16_t* Y, int64_t n8) {
  auto f = [](float a) { return MODE == 1 ? a / (1.f + __expf(-1.702f * a)) : 0.5f * a * (1.f + erff(a * 0.70710678118654752f)); };
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(X + i * 8);
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack16(f(lo16(v[j])), f(hi16(v[j])));
    *reinterpret_cast<u32x4_t*>(Y + i * 8) = o;
  }
}

__global__ void concat_kernel(const uint16_t* A, int64_t Ca8, const uint16_t* B, int64_t Cb8, uint16_t* Y, int64_t M) {
  const int64_t C8 = Ca8 + Cb8, total = M * C8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / C8, c = i % C8;
    const uint16_t* src = (c < Ca8) ? A + (m * Ca8 + c) * 8 : B + (m * Cb8 + (c - Ca8)) * 8;
    st_stream(Y + i * 8, ld_stream(src));
  }
}

__global__ void timestep_kernel(const float* t, uint16_t* Y, int V, int dim) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V * half) return;
  const int v = i / half, j = i % half;
  const float freq = expf(-9.210340371976184f * (float)j / (float)half);   // ln(10000)
  const float a = t[v] * freq;
  Y[(int64_t)v * dim + j] = f2h(cosf(a));
  Y[(int64_t)v * dim + half + j] = f2h(sinf(a));
}

A3D_DEV float ld_f32(const float* p, int64_t i) { return p[i]; }
A3D_DEV float ld_f32(const uint16_t* p, int64_t i) { return bfbits2f(p[i]); }      // caller tensor in bf16
A3D_DEV float ld_f32(const _Float16* p, int64_t i) { return (float)p[i]; }
A3D_DEV void st_f32(float* p, int64_t i, float v) { p[i] = v; }
A3D_DEV void st_f32(uint16_t* p, int64_t i, float v) { p[i] = f2bfbits(v); }       // caller tensor in bf16
A3D_DEV void st_f32(_Float16* p, int64_t i, float v) { p[i] = (_Float16)v; }

// one thread per output pixel (v, f, y, x): gathers the 3x3xC patch from [V, C, F, H, W]
template <typename T>
__global__ void im2col_in_kernel(const T* S, uint16_t* Y, int V, int C, int F, int H, int W) {
  const int64_t total = (int64_t)V * F * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const int f = (int)((i / ((int64_t)W * H)) % F);
    const int v = (int)(i / ((int64_t)W * H * F));
    const int K = 9 * C;
    uint16_t* dst = Y + i * 64;
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8) {
      float fv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k8 * 8 + j;
        float val = 0.f;
        if (k < K) {
          const int tap = k / C, c = k - tap * C;
          const int ky = tap / 3, kx = tap - ky * 3;
          const int yy = y + ky - 1, xx = x + kx - 1;
          if (yy >= 0 && yy < H && xx >= 0 && xx < W)
            val = ld_f32(S, ((((int64_t)v * C + c) * F + f) * H + yy) * W + xx);
        }
        fv[j] = val;
      }
      u32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = pack16(fv[2 * j], fv[2 * j + 1]);
      *reinterpret_cast<u32x4_t*>(dst + k8 * 8) = o;
    }
  }
}

template <typename T>
__global__ void unpack_out_kernel(const uint16_t* X, T* Y, int V, int C, int F, int H, int W) {
  const int64_t total = (int64_t)V * C * F * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const int f = (int)((i / ((int64_t)W * H)) % F);
    const int c = (int)((i / ((int64_t)W * H * F)) % C);
    const int v = (int)(i / ((int64_t)W * H * F * C));
    const int64_t row = (((int64_t)v * F + f) * H + y) * W + x;
    st_f32(Y, i, h2f(X[row * C + c]));
  }
}

__global__ void cfg_ddim_kernel(const float* eps_pair, const float* x, const float* first, float* x_prev,
                                int64_t n, int C, int F, int64_t HW, float guidance, float sa_t, float s1a_t,
                                float sa_p, float s1a_p) {
  const int64_t per = (int64_t)C * F * HW, total = n * per;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t hw = i % HW;
    const int f = (int)((i / HW) % F);
    const int64_t vc = i / (HW * F);     // v*C + c
    if (f == 0) { x_prev[i] = first[vc * HW + hw]; continue; }
    const float eu = eps_pair[i], et = eps_pair[total + i];
    const float e = eu + guidance * (et - eu);
    const float x0 = (x[i] - s1a_t * e) / sa_t;
    x_prev[i] = sa_p * x0 + s1a_p * e;
  }
}

}  // namespace

namespace {

// row softmax of fp32 logits [M, N] -> bf16 probabilities (one wave per row, three passes over the L2-resident row)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* X, int64_t ldx, uint16_t* Y, int64_t ldy, int64_t M, int N) {
  const int lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float* x = X + m * ldx;
  float mx = -INFINITY;
  for (int c = lane * 4; c < N; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(x + c);
    mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int c = lane * 4; c < N; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(x + c);
    sum += __expf(v.x - mx) + __expf(v.y - mx) + __expf(v.z - mx) + __expf(v.w - mx);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = 1.f / sum;
  for (int c = lane * 4; c < N; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(x + c);
    u32x2_t o;
    o[0] = pack16(__expf(v.x - mx) * inv, __expf(v.y - mx) * inv);
    o[1] = pack16(__expf(v.z - mx) * inv, __expf(v.w - mx) * inv);
    *reinterpret_cast<u32x2_t*>(Y + m * ldy + c) = o;
  }
}

// y[b, o, p] = scale * (sum_c w[o, c] x[b, c, p]) + bias[o] on planar fp32 images with a handful of channels (VAE post_quant_conv)
__global__ void channel_mix_kernel(const float* X, const float* Wm, const float* bias, float* Y, int B, int Cin, int Cout, int64_t HW, float scale) {
  const int64_t total = (int64_t)B * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / HW, px = i % HW;
    float xin[8];
    for (int c = 0; c < Cin; ++c) xin[c] = X[(b * Cin + c) * HW + px];
    for (int o = 0; o < Cout; ++o) {
      float acc = 0.f;
      for (int c = 0; c < Cin; ++c) acc = fmaf(Wm[o * Cin + c], xin[c], acc);
      Y[(b * Cout + o) * HW + px] = fmaf(scale, acc, bias ? bias[o] : 0.f);
    }
  }
}

}  // namespace

#ifdef A3D_STORAGE_F16
#define A3D_SOFTMAX_ROWS a3d_softmax_rows_f32_f16
#define A3D_IM2COL_IN a3d_im2col_in_f16
#define A3D_UNPACK_OUT a3d_unpack_out_f16
#else
#define A3D_SOFTMAX_ROWS a3d_softmax_rows_f32_bf16
#define A3D_IM2COL_IN a3d_im2col_in
#define A3D_UNPACK_OUT a3d_unpack_out
#endif

extern "C" int A3D_SOFTMAX_ROWS(a3d_stream_t stream, const float* X, int64_t ldx, void* Y, int64_t ldy, int64_t M, int64_t N) {
  if (!X || !Y || M <= 0 || N <= 0 || N % 4 != 0 || N > 0x7fffffffLL || ldx % 4 != 0 || ldy % 4 != 0) return A3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(X) & 15u) || (reinterpret_cast<uintptr_t>(Y) & 7u)) return A3D_EINVAL;
  const int64_t nblk = (M + 3) / 4;
  if (nblk > 0x7fffffffLL) return A3D_EINVAL;
  softmax_rows_kernel<<<dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream>>>(X, ldx, (uint16_t*)Y, ldy, M, (int)N);
  return a3d_launch_status();
}

#ifndef A3D_STORAGE_F16
extern "C" int a3d_channel_mix_f32(a3d_stream_t stream, const float* X, const float* W, const float* bias, float* Y,
                                   int B, int Cin, int Cout, int64_t HW, float scale) {
  if (!X || !W || !Y || B <= 0 || Cin <= 0 || Cin > 8 || Cout <= 0 || Cout > 8 || HW <= 0) return A3D_EINVAL;
  channel_mix_kernel<<<dim3(grid_for((int64_t)B * HW)), dim3(256), 0, (hipStream_t)stream>>>(X, W, bias, Y, B, Cin, Cout, HW, scale);
  return a3d_launch_status();
}
#endif

#ifndef A3D_STORAGE_F16
extern "C" const char* a3d_version(void) { return "animate3d_hip gfx950 r6 (bf16 + fp16 storage)"; }
#endif

extern "C" int A3D_FN(a3d_geglu)(a3d_stream_t stream, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t M, int64_t N) {
  if (!X || !Y || M <= 0 || N <= 0 || N % 8 || ldx % 8 || ldy % 8) return A3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15u) return A3D_EINVAL;
  geglu_kernel<<<grid_for(M * (N / 8)), 256, 0, (hipStream_t)stream>>>((const uint16_t*)X, ldx, (uint16_t*)Y, ldy, M, N / 8);
  return a3d_launch_status();
}

extern "C" int A3D_FN(a3d_silu)(a3d_stream_t stream, const void* X, void* Y, int64_t n) {
  if (!X || !Y || n <= 0 || n % 8) return A3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15u) return A3D_EINVAL;
  silu_kernel<<<grid_for(n / 8), 256, 0, (hipStream_t)stream>>>((const uint16_t*)X, (uint16_t*)Y, n / 8);
  return a3d_launch_status();
}

extern "C" int A3D_FN(a3d_activation)(a3d_stream_t stream, const void* X, void* Y, int64_t n, int mode) {
  if (!X || !Y || n <= 0 || n % 8 != 0 || mode < 0 || mode > 2) return A3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15u) return A3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) silu_kernel<<<grid_for(n / 8), 256, 0, s>>>((const uint16_t*)X, (uint16_t*)Y, n / 8);
  else if (mode == 1) act_kernel<1><<<grid_for(n / 8), 256, 0, s>>>((const uint16_t*)X, (uint16_t*)Y, n / 8);
  else act_kernel<2><<<grid_for(n / 8), 256, 0, s>>>((const uint16_t*)X, (uint16_t*)Y, n / 8);
  return a3d_launch_status();
}

extern "C" int A3D_FN(a3d_concat)(a3d_stream_t stream, const void* A, int64_t Ca, const void* Bsrc, int64_t Cb, void* Y, int64_t M) {
  if (!A || !Bsrc || !Y || M <= 0 || Ca <= 0 || Cb <= 0 || Ca % 8 || Cb % 8) return A3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(Bsrc) | reinterpret_cast<uintptr_t>(Y)) & 15u) return A3D_EINVAL;
  concat_kernel<<<grid_for(M * ((Ca + Cb) / 8)), 256, 0, (hipStream_t)stream>>>((const uint16_t*)A, Ca / 8, (const uint16_t*)Bsrc, Cb / 8, (uint16_t*)Y, M);
  return a3d_launch_status();
}

extern "C" int A3D_FN(a3d_timestep_embed)(a3d_stream_t stream, const float* t, void* Y, int V, int dim) {
  if (!t || !Y || V <= 0 || dim <= 0 || dim % 2) return A3D_EINVAL;
  const int total = V * (dim / 2);
  timestep_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(t, (uint16_t*)Y, V, dim);
  return a3d_launch_status();
}

extern "C" int A3D_IM2COL_IN(a3d_stream_t stream, const void* sample, int dtype, void* Y, int V, int C, int F, int H, int W) {
  if (!sample || !Y || V <= 0 || C <= 0 || F <= 0 || H <= 0 || W <= 0 || 9 * C > 64) return A3D_EINVAL;
  const unsigned g = grid_for((int64_t)V * F * H * W);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case A3D_F32: im2col_in_kernel<float><<<g, 256, 0, s>>>((const float*)sample, (uint16_t*)Y, V, C, F, H, W); break;
    case A3D_BF16: im2col_in_kernel<uint16_t><<<g, 256, 0, s>>>((const uint16_t*)sample, (uint16_t*)Y, V, C, F, H, W); break;
    case A3D_F16: im2col_in_kernel<_Float16><<<g, 256, 0, s>>>((const _Float16*)sample, (uint16_t*)Y, V, C, F, H, W); break;
    default: return A3D_EINVAL;
  }
  return a3d_launch_status();
}

extern "C" int A3D_UNPACK_OUT(a3d_stream_t stream, const void* X, void* Y, int dtype, int V, int C, int F, int H, int W) {
  if (!X || !Y || V <= 0 || C <= 0 || F <= 0 || H <= 0 || W <= 0) return A3D_EINVAL;
  const unsigned g = grid_for((int64_t)V * C * F * H * W);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case A3D_F32: unpack_out_kernel<float><<<g, 256, 0, s>>>((const uint16_t*)X, (float*)Y, V, C, F, H, W); break;
    case A3D_BF16: unpack_out_kernel<uint16_t><<<g, 256, 0, s>>>((const uint16_t*)X, (uint16_t*)Y, V, C, F, H, W); break;
    case A3D_F16: unpack_out_kernel<_Float16><<<g, 256, 0, s>>>((const uint16_t*)X, (_Float16*)Y, V, C, F, H, W); break;
    default: return A3D_EINVAL;
  }
  return a3d_launch_status();
}

#ifndef A3D_STORAGE_F16
extern "C" int a3d_cfg_ddim_step_f32(a3d_stream_t stream, const float* eps_pair, const float* x, const float* first_frame,
                                     float* x_prev, int64_t n, int C, int F, int64_t HW, float guidance,
                                     float alpha_t, float alpha_prev) {
  if (!eps_pair || !x || !first_frame || !x_prev || n <= 0 || C <= 0 || F <= 0 || HW <= 0) return A3D_EINVAL;
  if (alpha_t <= 0.f || alpha_t > 1.f || alpha_prev <= 0.f || alpha_prev > 1.f) return A3D_EINVAL;
  cfg_ddim_kernel<<<grid_for(n * C * F * HW), 256, 0, (hipStream_t)stream>>>(
      eps_pair, x, first_frame, x_prev, n, C, F, HW, guidance, sqrtf(alpha_t), sqrtf(1.f - alpha_t), sqrtf(alpha_prev),
      sqrtf(1.f - alpha_prev));
  return a3d_launch_status();
}
#endif
