// Backward / optimiser kernels of the training path (reference train.py:576-601 runs these through torch autograd
// and torch.optim.AdamW).  All HBM-bound elementwise / reduction work: 16-byte accesses, fp32 arithmetic, 16-bit storage (bf16 / fp16
// build); the matrix products of the backward pass reuse the forward GEMM / conv kernels on transposed operands.
#include "common.h"

namespace {

A3D_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ------------------------------------------------------------------ transpose (weights and activations of the wgrad GEMMs)
// Y[c][r] = X[r][c], r < rows, c < cols; Y columns rows .. rows_pad-1 are zero (the wgrad contraction needs K % 64 == 0).
__global__ __launch_bounds__(256) void transpose_kernel(const uint16_t* __restrict__ X, int64_t ldx, uint16_t* __restrict__ Y, int64_t ldy,
                                                         int64_t rows, int64_t cols, int64_t rows_pad) {
  __shared__ uint16_t tile[64][66];
  const int64_t r0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int64_t r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? X[r * ldx + c] : (uint16_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int64_t c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows_pad) Y[c * ldy + r] = tile[tx][i];
  }
}

// The same with 16-byte global accesses on both sides (cols, rows_pad and both leading dimensions multiples of 8): a 64 x 64 tile goes
// into LDS row by row (pitch 66 halves = 33 words: the column gather below spreads over the banks), every thread then gathers 8 rows of
// one column into a 16-byte store.  The per-step transposes of the trainable weights (dX = dY W needs W^T) are up to 10240 x 1280.
__global__ __launch_bounds__(256) void transpose16_kernel(const uint16_t* __restrict__ X, int64_t ldx, uint16_t* __restrict__ Y, int64_t ldy,
                                                           int64_t rows, int64_t cols, int64_t rows_pad) {
  __shared__ uint32_t tile[64][33];
  const int64_t r0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = threadIdx.x + 256 * i, r = q >> 3, cc = (q & 7) * 8;
    u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
    if (r0 + r < rows && c0 + cc < cols) v = *reinterpret_cast<const u32x4_t*>(X + (r0 + r) * ldx + c0 + cc);
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[r][(cc >> 1) + j] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = threadIdx.x + 256 * i, c = q >> 3, rr = (q & 7) * 8;
    if (c0 + c < cols && r0 + rr < rows_pad) {
      uint32_t h[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { const uint32_t w = tile[rr + e][c >> 1]; h[e] = (c & 1) ? (w >> 16) : (w & 0xffffu); }
      u32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = h[2 * j] | (h[2 * j + 1] << 16);
      *reinterpret_cast<u32x4_t*>(Y + (c0 + c) * ldy + r0 + rr) = o;
    }
  }
}

// ------------------------------------------------------------------ column sums (bias gradients): out[c] += alpha * sum_r X[r][c]
__global__ __launch_bounds__(256) void colsum_kernel(const uint16_t* __restrict__ X, int64_t ldx, int64_t rows, int64_t cols,
                                                      int64_t rows_per_block, float* __restrict__ out, float alpha) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + tx;
  const int64_t r_beg = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r_end = min(rows, r_beg + rows_per_block);
  float s = 0.f;
  if (c < cols)
    for (int64_t r = r_beg + ty; r < r_end; r += 4) s += h2f(X[r * ldx + c]);
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < cols) atomicAdd(out + c, alpha * (red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]));
}

// Widths that are multiples of 320 (every bias of the model): TPR lanes own a row of a 40 * TPR column tile, five 16-byte chunks per lane
// (chunk = sub + TPR * i: TPR * 16 contiguous bytes per step), a wave adds 64 / TPR rows per step.  The kernel above reads 2 bytes per lane
// and load (1.7 TB/s on [65536, 320]).
template <int TPR>
__global__ __launch_bounds__(256) void colsum_rows_kernel(const uint16_t* __restrict__ X, int64_t ldx, int64_t rows, int64_t rows_per_block,
                                                           float* __restrict__ out, float alpha) {
  constexpr int RPW = 64 / TPR;
  __shared__ float red[4][40 * TPR];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int sub = lane % TPR, rsel = lane / TPR;
  const int64_t c0 = (int64_t)blockIdx.x * (40 * TPR);
  const int64_t r_beg = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r_end = min(rows, r_beg + rows_per_block);
  float acc[5][8];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
#pragma unroll 2
  for (int64_t r = r_beg + wid * RPW + rsel; r < r_end; r += 4 * RPW) {
    const uint16_t* xr = X + r * ldx + c0;
    u32x4_t v[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) v[i] = ld_stream(xr + (sub + TPR * i) * 8);
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] += (j & 1) ? hi16(v[i][j >> 1]) : lo16(v[i][j >> 1]);
  }
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int o = 32; o >= TPR; o >>= 1) acc[i][j] += __shfl_xor(acc[i][j], o);
      if (rsel == 0) red[wid][(sub + TPR * i) * 8 + j] = acc[i][j];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < 40 * TPR; c += 256) atomicAdd(out + c0 + c, alpha * ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])));
}

template <int TPR>
int launch_colsum_rows(hipStream_t s, const uint16_t* X, int64_t ldx, int64_t rows, int64_t cols, float* out, float alpha) {
  const int64_t bx = cols / (40 * TPR);
  int64_t by = rows / (16 * (64 / TPR));                      // >= 4 steps of every wave per block
  const int64_t cap = bx >= 512 ? 1 : 512 / bx;          // every workgroup ends in one atomic per column
  if (by > cap) by = cap;
  if (by < 1) by = 1;
  const int64_t rpb = (rows + by - 1) / by;
  colsum_rows_kernel<TPR><<<dim3((unsigned)bx, (unsigned)by), dim3(256), 0, s>>>(X, ldx, rows, rpb, out, alpha);
  return a3d_launch_status();
}

// ------------------------------------------------------------------ GEGLU backward on the interleaved projection
// P [M, 2N]: column blocks of 64 = [32 h | 32 gate] (HipOps.interleave_geglu); y = h * gelu(gate);  dP = [dY gelu(gate) | dY h gelu'(gate)]
A3D_DEV float gelu_f(float g) { return 0.5f * g * (1.f + erff(g * 0.70710678118654752f)); }
A3D_DEV float gelu_grad(float g) {
  return 0.5f * (1.f + erff(g * 0.70710678118654752f)) + g * 0.3989422804014327f * __expf(-0.5f * g * g);
}
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const uint16_t* __restrict__ P, int64_t ldp, const uint16_t* __restrict__ dY, int64_t lddy,
                                                         uint16_t* __restrict__ dP, int64_t lddp, int64_t M, int64_t N) {
  const int64_t chunks = N / 8;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * chunks) return;
  const int64_t m = idx / chunks, ch = idx % chunks;
  const int64_t blk = ch / 4, sub = ch % 4;
  const int64_t pc = blk * 64 + sub * 8;
  const u32x4_t hv = *reinterpret_cast<const u32x4_t*>(P + m * ldp + pc);
  const u32x4_t gv = *reinterpret_cast<const u32x4_t*>(P + m * ldp + pc + 32);
  const u32x4_t dv = *reinterpret_cast<const u32x4_t*>(dY + m * lddy + ch * 8);
  u32x4_t dh, dg;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float h0 = lo16(hv[j]), h1 = hi16(hv[j]), g0 = lo16(gv[j]), g1 = hi16(gv[j]), d0 = lo16(dv[j]), d1 = hi16(dv[j]);
    dh[j] = pack16(d0 * gelu_f(g0), d1 * gelu_f(g1));
    dg[j] = pack16(d0 * h0 * gelu_grad(g0), d1 * h1 * gelu_grad(g1));
  }
  *reinterpret_cast<u32x4_t*>(dP + m * lddp + pc) = dh;
  *reinterpret_cast<u32x4_t*>(dP + m * lddp + pc + 32) = dg;
}

// ------------------------------------------------------------------ y = a * x + b * y
__global__ __launch_bounds__(256) void axpby_kernel(const uint16_t* __restrict__ X, uint16_t* __restrict__ Y, int64_t n8, float a, float b) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const u32x4_t x = reinterpret_cast<const u32x4_t*>(X)[i];
  u32x4_t y = u32x4_t{0u, 0u, 0u, 0u};
  if (b != 0.f) y = reinterpret_cast<const u32x4_t*>(Y)[i];
  u32x4_t o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = pack16(a * lo16(x[j]) + b * lo16(y[j]), a * hi16(x[j]) + b * hi16(y[j]));
  reinterpret_cast<u32x4_t*>(Y)[i] = o;
}

// ------------------------------------------------------------------ LayerNorm backward: one wave per row, CPL channels per lane
struct LNBParams {
  const uint16_t* X; const uint16_t* dY; const float* gamma; uint16_t* dX; float* dgamma; float* dbeta;
  int64_t M; int C; float eps;
};
template <int CPL>
__global__ __launch_bounds__(256) void layer_norm_bwd_kernel(const LNBParams p) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int C = p.C;
  const float invC = 1.f / (float)C;
  float gam[CPL], dg[CPL], db[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int c = lane + 64 * k;
    gam[k] = c < C ? p.gamma[c] : 0.f;
    dg[k] = 0.f; db[k] = 0.f;
  }
  for (int64_t m = (int64_t)blockIdx.x * 4 + wid; m < p.M; m += (int64_t)gridDim.x * 4) {
    float x[CPL], dy[CPL];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      x[k] = c < C ? h2f(p.X[m * C + c]) : 0.f;
      dy[k] = c < C ? h2f(p.dY[m * C + c]) : 0.f;
      s += x[k];
    }
    const float mean = wave_sum(s) * invC;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      const float d = c < C ? x[k] - mean : 0.f;
      v += d * d;
    }
    const float rstd = rsqrtf(wave_sum(v) * invC + p.eps);
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      x[k] = c < C ? (x[k] - mean) * rstd : 0.f;            // x-hat
      const float g = dy[k] * gam[k];
      a += g; b += g * x[k];
      dg[k] += dy[k] * x[k]; db[k] += dy[k];
    }
    a = wave_sum(a) * invC; b = wave_sum(b) * invC;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      if (c < C) p.dX[m * C + c] = f2h(rstd * (dy[k] * gam[k] - a - x[k] * b));
    }
  }
  if (p.dgamma) {
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      if (c < C) { atomicAdd(p.dgamma + c, dg[k]); atomicAdd(p.dbeta + c, db[k]); }
    }
  }
}

// Sub-wave rows for the model's widths C = 320 / 640 / 1280 = 40 * LPR (the forward's layout, norms.hip layer_norm_rows_kernel): LPR lanes
// own one row — five 16-byte chunks of X and five of dY per lane, chunk = sub + LPR * i, so a step of the wave reads LPR * 16 contiguous
// bytes per row and ten loads per lane are in flight — and a wave handles 64 / LPR rows per batch.  The one-wave-per-row kernel above
// reads 2 bytes per lane and load (0.8 TB/s at C = 320) and ends in 2 C atomics per wave; here the workgroup reduces its dgamma / dbeta
// partials through LDS first (2 C atomics per workgroup, <= 512 workgroups), and frozen affines (PARAM = false) skip that part.
template <int LPR, bool PARAM>
__global__ __launch_bounds__(256, 2) void layer_norm_bwd_rows_kernel(const LNBParams p) {
  constexpr int RPW = 64 / LPR;
  __shared__ float red[PARAM ? 4 : 1][PARAM ? 2 : 1][PARAM ? 40 * LPR : 1];
  __shared__ __attribute__((aligned(16))) float gsh[PARAM ? 40 * LPR : 4];      // PARAM: gamma is read from LDS (80 accumulator registers per lane)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int sub = lane % LPR, rsel = lane / LPR;
  const int C = p.C;
  float gar[PARAM ? 1 : 5][8], dg[PARAM ? 5 : 1][8], db[PARAM ? 5 : 1][8];
  if constexpr (PARAM) {
    for (int c = threadIdx.x; c < C; c += 256) gsh[c] = p.gamma[c];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) { dg[i][j] = 0.f; db[i][j] = 0.f; }
    __syncthreads();
  } else {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int c = sub + LPR * i;
      const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + c * 8), g1 = *reinterpret_cast<const float4*>(p.gamma + c * 8 + 4);
      gar[i][0] = g0.x; gar[i][1] = g0.y; gar[i][2] = g0.z; gar[i][3] = g0.w; gar[i][4] = g1.x; gar[i][5] = g1.y; gar[i][6] = g1.z; gar[i][7] = g1.w;
    }
  }
  auto gamma8 = [&](int i, float (&g)[8]) {
    if constexpr (PARAM) {
      const float4 g0 = *reinterpret_cast<const float4*>(gsh + (sub + LPR * i) * 8), g1 = *reinterpret_cast<const float4*>(gsh + (sub + LPR * i) * 8 + 4);
      g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = gar[i][j];
    }
  };
  const float inv_c = 1.f / (float)C;
  const int64_t nbatch = (p.M + RPW - 1) / RPW;
#pragma unroll 1
  for (int64_t bt = (int64_t)blockIdx.x * 4 + wid; bt < nbatch; bt += (int64_t)gridDim.x * 4) {
    const int64_t m = bt * RPW + rsel;
    const bool ok = m < p.M;
    const int64_t mm = ok ? m : p.M - 1;
    u32x4_t xr[5], dr[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) xr[i] = ld_stream(p.X + mm * C + (sub + LPR * i) * 8);
#pragma unroll
    for (int i = 0; i < 5; ++i) dr[i] = ld_stream(p.dY + mm * C + (sub + LPR * i) * 8);
    float x[5][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) { x[i][j] = (j & 1) ? hi16(xr[i][j >> 1]) : lo16(xr[i][j >> 1]); sum += x[i][j]; }
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum * inv_c;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) { x[i][j] -= mean; sq += x[i][j] * x[i][j]; }
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = rsqrtf(sq * inv_c + p.eps);
    const float okf = ok ? 1.f : 0.f;
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      float g[8];
      gamma8(i, g);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dy = (j & 1) ? hi16(dr[i][j >> 1]) : lo16(dr[i][j >> 1]);
        x[i][j] *= rstd;                                       // x-hat
        const float gy = dy * g[j];
        a += gy; b += gy * x[i][j];
        if constexpr (PARAM) { dg[i][j] = fmaf(dy * okf, x[i][j], dg[i][j]); db[i][j] = fmaf(dy, okf, db[i][j]); }
      }
    }
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    a *= inv_c; b *= inv_c;
    if (ok) {
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        float g[8];
        gamma8(i, g);
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j)         // dY is decoded again from its packed registers instead of keeping 40 products alive
          o[j] = pack16(rstd * (lo16(dr[i][j]) * g[2 * j] - a - x[i][2 * j] * b), rstd * (hi16(dr[i][j]) * g[2 * j + 1] - a - x[i][2 * j + 1] * b));
        st_stream(p.dX + m * C + (sub + LPR * i) * 8, o);
      }
    }
  }
  if constexpr (PARAM) {
    // rows of the wave (lanes with the same sub), then the four waves through LDS, then one atomic per channel and workgroup
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int o = 32; o >= LPR; o >>= 1) { dg[i][j] += __shfl_xor(dg[i][j], o); db[i][j] += __shfl_xor(db[i][j], o); }
        if (rsel == 0) { red[wid][0][(sub + LPR * i) * 8 + j] = dg[i][j]; red[wid][1][(sub + LPR * i) * 8 + j] = db[i][j]; }
      }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      atomicAdd(p.dgamma + c, (red[0][0][c] + red[1][0][c]) + (red[2][0][c] + red[3][0][c]));
      atomicAdd(p.dbeta + c, (red[0][1][c] + red[1][1][c]) + (red[2][1][c] + red[3][1][c]));
    }
  }
}

template <int LPR>
int launch_ln_bwd_rows(hipStream_t s, const LNBParams& p) {
  const int64_t nbatch = (p.M + 64 / LPR - 1) / (64 / LPR);
  int64_t blocks = (nbatch + 3) / 4;
  if (p.dgamma) {
    if (blocks > 512) blocks = 512;
    layer_norm_bwd_rows_kernel<LPR, true><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(p);
  } else {
    if (blocks > 4096) blocks = 4096;
    layer_norm_bwd_rows_kernel<LPR, false><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(p);
  }
  return a3d_launch_status();
}

// ------------------------------------------------------------------ GroupNorm (+SiLU) backward, channel-last [B][rows][C]
struct GNBParams {
  const uint16_t* X; const uint16_t* dY; const float* gamma; const float* beta; const float* stats;   // stats [B][groups][2] = mean, rstd
  uint16_t* dX; float* ws;      // ws [B][C][2]: per-channel sums of g and g * x-hat (zeroed by the entry point)
  float* dgamma; float* dbeta;
  int B; int64_t rows; int C, groups, cg, silu; int64_t rows_per_block;
};
A3D_DEV float gn_upstream(float dy, float xh, float gam, float bet, int silu) {
  if (!silu) return dy;
  const float z = fmaf(xh, gam, bet);
  const float sg = 1.f / (1.f + __expf(-z));
  return dy * sg * (1.f + z * (1.f - sg));
}
// Both passes: a workgroup is rpb rows x (C / 8) 16-byte chunks (rpb = max(1, 256 / (C / 8)): 240 threads at C = 320 / 640 / 960), thread =
// (row slot, chunk) with the chunk fixed, so that its eight channels' statistics and affine parameters are loop invariants and a row step of
// the workgroup reads rpb * C * 2 contiguous bytes of X and of dY.
struct GNChan { float mean[8], rstd[8], gam[8], bet[8]; };
A3D_DEV void gn_chan_load(const GNBParams& p, int b, int c0, GNChan& k) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int grp = (c0 + e) / p.cg;
    k.mean[e] = p.stats[((int64_t)b * p.groups + grp) * 2];
    k.rstd[e] = p.stats[((int64_t)b * p.groups + grp) * 2 + 1];
    k.gam[e] = p.gamma[c0 + e];
    k.bet[e] = p.beta[c0 + e];
  }
}
// pass 1: per-(b, channel) sums of g and g * x-hat over the workgroup's rows.  Every thread leaves its 8 x 2 partial sums in LDS
// ([row slot][C][2], dynamic: blockDim * 64 bytes), then consecutive threads add up the row slots of consecutive ws entries: the 2 C
// atomics of a workgroup go out as whole cache lines (one lane per 64-byte segment, as the thread = chunk layout would issue them, is
// what made the first version of this kernel 2.5 x slower than the 2-byte-load kernel it replaced).
__global__ __launch_bounds__(1024) void gn_bwd_sums_kernel(const GNBParams p) {
  extern __shared__ float red[];
  const int nch = p.C / 8;
  const int ch = threadIdx.x % nch, rr = threadIdx.x / nch, rpb = blockDim.x / nch;
  const int b = blockIdx.y, c0 = ch * 8;
  GNChan k;
  gn_chan_load(p, b, c0, k);
  const int64_t r_beg = (int64_t)blockIdx.x * p.rows_per_block, r_end = min(p.rows, r_beg + p.rows_per_block);
  const uint16_t* x = p.X + ((int64_t)b * p.rows) * p.C + c0;
  const uint16_t* dy = p.dY + ((int64_t)b * p.rows) * p.C + c0;
  float sa[8], sb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sa[e] = 0.f; sb[e] = 0.f; }
#pragma unroll 2
  for (int64_t r = r_beg + rr; r < r_end; r += rpb) {
    const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(x + r * p.C), dv = *reinterpret_cast<const u32x4_t*>(dy + r * p.C);   // (pass 2 reads both again)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (((e & 1) ? hi16(xv[e >> 1]) : lo16(xv[e >> 1])) - k.mean[e]) * k.rstd[e];
      const float g = gn_upstream((e & 1) ? hi16(dv[e >> 1]) : lo16(dv[e >> 1]), xh, k.gam[e], k.bet[e], p.silu);
      sa[e] += g; sb[e] = fmaf(g, xh, sb[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) *reinterpret_cast<float2*>(red + ((int64_t)rr * p.C + c0 + e) * 2) = make_float2(sa[e], sb[e]);
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * p.C; t += blockDim.x) {
    float v = red[t];
    for (int q = 1; q < rpb; ++q) v += red[q * 2 * p.C + t];
    atomicAdd(p.ws + (int64_t)b * 2 * p.C + t, v);
  }
}
// pass 2: dx = rstd * (g gamma - s1/n - x-hat s2/n) with the group sums s1 = sum_c gamma_c A_c, s2 = sum_c gamma_c B_c
__global__ __launch_bounds__(1024) void gn_bwd_apply_kernel(const GNBParams p) {
  __shared__ float gs[2][64];
  const int b = blockIdx.y;
  if ((int)threadIdx.x < p.groups) {
    float s1 = 0.f, s2 = 0.f;
    for (int c = threadIdx.x * p.cg; c < ((int)threadIdx.x + 1) * p.cg; ++c) {
      const float gam = p.gamma[c];
      s1 += gam * p.ws[((int64_t)b * p.C + c) * 2];
      s2 += gam * p.ws[((int64_t)b * p.C + c) * 2 + 1];
    }
    const float inv_n = 1.f / ((float)p.rows * (float)p.cg);
    gs[0][threadIdx.x] = s1 * inv_n; gs[1][threadIdx.x] = s2 * inv_n;
  }
  __syncthreads();
  const int nch = p.C / 8;
  const int ch = threadIdx.x % nch, rr = threadIdx.x / nch, rpb = blockDim.x / nch;
  const int c0 = ch * 8;
  GNChan k;
  gn_chan_load(p, b, c0, k);
  float g1[8], g2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { g1[e] = gs[0][(c0 + e) / p.cg]; g2[e] = gs[1][(c0 + e) / p.cg]; }
  const int64_t r_beg = (int64_t)blockIdx.x * p.rows_per_block, r_end = min(p.rows, r_beg + p.rows_per_block);
  const int64_t base = ((int64_t)b * p.rows) * p.C + c0;
#pragma unroll 2
  for (int64_t r = r_beg + rr; r < r_end; r += rpb) {
    const u32x4_t xv = ld_stream(p.X + base + r * p.C), dv = ld_stream(p.dY + base + r * p.C);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (((e & 1) ? hi16(xv[e >> 1]) : lo16(xv[e >> 1])) - k.mean[e]) * k.rstd[e];
      const float g = gn_upstream((e & 1) ? hi16(dv[e >> 1]) : lo16(dv[e >> 1]), xh, k.gam[e], k.bet[e], p.silu);
      o[e] = k.rstd[e] * (g * k.gam[e] - g1[e] - xh * g2[e]);
    }
    u32x4_t ov;
#pragma unroll
    for (int j = 0; j < 4; ++j) ov[j] = pack16(o[2 * j], o[2 * j + 1]);
    st_stream(p.dX + base + r * p.C, ov);
  }
}
__global__ __launch_bounds__(256) void gn_bwd_param_kernel(const GNBParams p) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= p.C) return;
  float a = 0.f, bb = 0.f;
  for (int b = 0; b < p.B; ++b) { a += p.ws[((int64_t)b * p.C + c) * 2]; bb += p.ws[((int64_t)b * p.C + c) * 2 + 1]; }
  p.dbeta[c] += a; p.dgamma[c] += bb;
}

// ------------------------------------------------------------------ conv helpers
// stride-2 conv dgrad = stride-1 conv of the zero-stuffed gradient: Z[b][2oy][2ox] = dY[b][oy][ox], zero elsewhere
__global__ __launch_bounds__(256) void zero_insert_kernel(const uint16_t* __restrict__ dY, uint16_t* __restrict__ Z, int B, int H, int W, int Ho, int Wo, int C) {
  const int chunks = C / 8;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * H * W * chunks) return;
  const int ch = (int)(idx % chunks);
  const int64_t pix = idx / chunks;
  const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
  u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
  if ((x & 1) == 0 && (y & 1) == 0 && (y >> 1) < Ho && (x >> 1) < Wo)
    v = *reinterpret_cast<const u32x4_t*>(dY + (((int64_t)b * Ho + (y >> 1)) * Wo + (x >> 1)) * C + ch * 8);
  *reinterpret_cast<u32x4_t*>(Z + pix * C + ch * 8) = v;
}
// nearest-2x upsample backward: dX[b][y][x] = sum of dU[b][2y+dy][2x+dx] inside the (possibly forced 2H-1 / 2W-1) extent
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const uint16_t* __restrict__ dU, uint16_t* __restrict__ dX, int B, int H, int W, int He, int We, int C) {
  const int chunks = C / 8;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * H * W * chunks) return;
  const int ch = (int)(idx % chunks);
  const int64_t pix = idx / chunks;
  const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int uy = 2 * y + dy, ux = 2 * x + dx;
      if (uy < He && ux < We) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(dU + (((int64_t)b * He + uy) * We + ux) * C + ch * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[2 * j] += lo16(v[j]); acc[2 * j + 1] += hi16(v[j]); }
      }
    }
  u32x4_t o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = pack16(acc[2 * j], acc[2 * j + 1]);
  *reinterpret_cast<u32x4_t*>(dX + pix * C + ch * 8) = o;
}

// ------------------------------------------------------------------ temporal attention backward
// One workgroup = one pixel x one 320-channel slab x all F frames (as the forward); thread = (40-dim slice, frame).
constexpr int SLAB = 320, SL = 40, NSL = SLAB / SL;
struct TABParams {
  const uint16_t* Q; const uint16_t* K; const uint16_t* V; int64_t ld;
  const uint16_t* dO; int64_t lddo;
  uint16_t* dQ; uint16_t* dK; uint16_t* dV; int64_t ldd;
  int frames; int64_t L; float scale, scale_log2; int64_t npix;
};
template <int FP, int DP>
__global__ __launch_bounds__(NSL * FP) void temporal_attn_bwd_kernel(const TABParams p) {
  constexpr int ROWB = SLAB + 8;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];     // [4][F][ROWB] then float Pm[NSL][F][F], dSm[NSL][F][F]
  const int F = p.frames;
  const int tid = threadIdx.x;
  const int64_t pix = blockIdx.x;
  const int c0 = blockIdx.y * SLAB;
  const int64_t v = pix / p.L, l = pix % p.L;
  float* const Pm = reinterpret_cast<float*>(smem + (size_t)4 * F * ROWB);
  float* const dSm = Pm + (size_t)NSL * F * F;
  constexpr int chunks = SLAB / 8;
  for (int it = tid; it < 4 * F * chunks; it += NSL * FP) {
    const int ch = it % chunks, f = (it / chunks) % F, ten = it / (chunks * F);
    const int64_t row = (v * F + f) * p.L + l;
    const uint16_t* src = ten == 0 ? p.Q + row * p.ld : ten == 1 ? p.K + row * p.ld : ten == 2 ? p.V + row * p.ld : p.dO + row * p.lddo;
    *reinterpret_cast<u32x4_t*>(smem + ((size_t)ten * F + f) * ROWB + ch * 8) = *reinterpret_cast<const u32x4_t*>(src + c0 + ch * 8);
  }
  __syncthreads();
  const int sl = tid % NSL, i = tid / NSL;
  const bool act = i < F;
  const int fi = act ? i : F - 1;
  auto slab = [&](int ten, int f) { return smem + ((size_t)ten * F + f) * ROWB + sl * SL; };
  float q[SL], go[SL];
#pragma unroll
  for (int d = 0; d < SL; ++d) { q[d] = h2f(slab(0, fi)[d]); go[d] = h2f(slab(3, fi)[d]); }
  float s[FP], dp[FP];
#pragma unroll
  for (int j = 0; j < FP; ++j) {
    float a = 0.f, b = 0.f;
    if (j < F) {
      const uint16_t* kr = slab(1, j); const uint16_t* vr = slab(2, j);
#pragma unroll
      for (int d = 0; d < SL; ++d) { a = fmaf(q[d], h2f(kr[d]), a); b = fmaf(go[d], h2f(vr[d]), b); }
    }
    s[j] = a; dp[j] = b;
  }
  if constexpr (DP >= 2) {
#pragma unroll
    for (int j = 0; j < FP; ++j) { s[j] += __shfl_xor(s[j], 1); dp[j] += __shfl_xor(dp[j], 1); }
  }
  if constexpr (DP >= 4) {
#pragma unroll
    for (int j = 0; j < FP; ++j) { s[j] += __shfl_xor(s[j], 2); dp[j] += __shfl_xor(dp[j], 2); }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < FP; ++j) if (j < F) mx = fmaxf(mx, s[j]);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < FP; ++j) { s[j] = (j < F) ? __builtin_amdgcn_exp2f((s[j] - mx) * p.scale_log2) : 0.f; sum += s[j]; }
  const float inv = 1.f / sum;
  float delta = 0.f;
#pragma unroll
  for (int j = 0; j < FP; ++j) { s[j] *= inv; delta = fmaf(s[j], dp[j], delta); }
  float dq[SL];
#pragma unroll
  for (int d = 0; d < SL; ++d) dq[d] = 0.f;
#pragma unroll
  for (int j = 0; j < FP; ++j) {
    if (j < F) {
      const float ds = s[j] * (dp[j] - delta) * p.scale;
      if (act) { Pm[((size_t)sl * F + i) * F + j] = s[j]; dSm[((size_t)sl * F + i) * F + j] = ds; }
      const uint16_t* kr = slab(1, j);
#pragma unroll
      for (int d = 0; d < SL; ++d) dq[d] = fmaf(ds, h2f(kr[d]), dq[d]);
    }
  }
  const int64_t row = (v * F + fi) * p.L + l;
  if (act) {
    uint16_t* dst = p.dQ + row * p.ldd + c0 + sl * SL;
#pragma unroll
    for (int c = 0; c < SL / 8; ++c) {
      u32x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack16(dq[8 * c + 2 * e], dq[8 * c + 2 * e + 1]);
      *reinterpret_cast<u32x4_t*>(dst + c * 8) = o;
    }
  }
  __syncthreads();
  // thread (slice, key frame j = i): dK_j = sum_i dS_ij Q_i, dV_j = sum_i P_ij dO_i
  float dk[SL], dv[SL];
#pragma unroll
  for (int d = 0; d < SL; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
  if (act) {
    for (int qi = 0; qi < F; ++qi) {
      const float ds = dSm[((size_t)sl * F + qi) * F + i], pp = Pm[((size_t)sl * F + qi) * F + i];
      const uint16_t* qr = slab(0, qi); const uint16_t* gr = slab(3, qi);
#pragma unroll
      for (int d = 0; d < SL; ++d) { dk[d] = fmaf(ds, h2f(qr[d]), dk[d]); dv[d] = fmaf(pp, h2f(gr[d]), dv[d]); }
    }
    uint16_t* dkd = p.dK + row * p.ldd + c0 + sl * SL;
    uint16_t* dvd = p.dV + row * p.ldd + c0 + sl * SL;
#pragma unroll
    for (int c = 0; c < SL / 8; ++c) {
      u32x4_t a, b;
#pragma unroll
      for (int e = 0; e < 4; ++e) { a[e] = pack16(dk[8 * c + 2 * e], dk[8 * c + 2 * e + 1]); b[e] = pack16(dv[8 * c + 2 * e], dv[8 * c + 2 * e + 1]); }
      *reinterpret_cast<u32x4_t*>(dkd + c * 8) = a;
      *reinterpret_cast<u32x4_t*>(dvd + c * 8) = b;
    }
  }
}

template <int FP, int DP>
int launch_tab(hipStream_t s, const TABParams& p, int C) {
  if (p.npix > 0x7fffffffLL) return A3D_EINVAL;
  const size_t lds = (size_t)4 * p.frames * (SLAB + 8) * sizeof(uint16_t) + (size_t)2 * NSL * p.frames * p.frames * sizeof(float);
  static uint64_t attr_done = 0;
  if (int rc = a3d_once_per_device(attr_done, [] {
        return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_attn_bwd_kernel<FP, DP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        4 * FP * (SLAB + 8) * 2 + 2 * NSL * FP * FP * 4); })) return rc;
  temporal_attn_bwd_kernel<FP, DP><<<dim3((unsigned)p.npix, (unsigned)(C / SLAB)), dim3(NSL * FP), lds, s>>>(p);
  return a3d_launch_status();
}
template <int DP>
int launch_tab_dp(hipStream_t s, const TABParams& p, int C) {
  return p.frames <= 16 ? launch_tab<16, DP>(s, p, C) : launch_tab<32, DP>(s, p, C);
}

#ifndef A3D_STORAGE_F16
// ------------------------------------------------------------------ optimiser (fp32 master parameters in one flat buffer)
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const float v = g[i]; s = fmaf(v, v, s); }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}
// ctrl[0] = factor applied to every gradient (1 / loss_scale, times the clip coefficient), ctrl[1] = 1 when the step must be skipped
// (non-finite gradients: GradScaler semantics of train.py:583-590), ctrl[2] = gradient norm after unscaling
__global__ void clip_ctrl_kernel(const float* __restrict__ sq, float max_norm, float inv_loss_scale, float* __restrict__ ctrl) {
  const float norm = sqrtf(sq[0]) * inv_loss_scale;
  const bool finite = isfinite(norm);
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(1.f, max_norm / (norm + 1e-6f));        // torch.nn.utils.clip_grad_norm_
  ctrl[0] = finite ? inv_loss_scale * coef : 0.f;
  ctrl[1] = finite ? 0.f : 1.f;
  ctrl[2] = norm;
}
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                     int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2,
                                                     const float* __restrict__ ctrl) {
  const float gs = ctrl ? ctrl[0] : 1.f;
  if (ctrl && ctrl[1] != 0.f) return;                                       // skipped step: parameters and moments untouched
  const float step = lr / bc1, rs = rsqrtf(bc2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * gs;
    float pi = p[i] * (1.f - lr * wd);                                      // decoupled weight decay (torch.optim.AdamW)
    const float mi = fmaf(b1, m[i], (1.f - b1) * gi);
    const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
    pi -= step * mi / (sqrtf(vi) * rs + eps);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}
#endif

}  // namespace

extern "C" int A3D_FN(a3d_transpose)(a3d_stream_t stream, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t rows, int64_t cols, int64_t rows_pad) {
  if (!X || !Y || rows <= 0 || cols <= 0 || rows_pad < rows || ldx < cols || ldy < rows_pad) return A3D_EINVAL;
  const int64_t bx = (rows_pad + 63) / 64, by = (cols + 63) / 64;
  if (bx > 0x7fffffffLL || by > 65535) return A3D_EINVAL;
  const bool wide = cols % 8 == 0 && rows_pad % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 &&
                    ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15u) == 0;
  if (wide) transpose16_kernel<<<dim3((unsigned)bx, (unsigned)by), dim3(256), 0, (hipStream_t)stream>>>((const uint16_t*)X, ldx, (uint16_t*)Y, ldy, rows, cols, rows_pad);
  else transpose_kernel<<<dim3((unsigned)bx, (unsigned)by), dim3(256), 0, (hipStream_t)stream>>>((const uint16_t*)X, ldx, (uint16_t*)Y, ldy, rows, cols, rows_pad);
  return a3d_launch_status();
}

extern "C" int A3D_FN(a3d_colsum)(a3d_stream_t stream, const void* X, int64_t ldx, int64_t rows, int64_t cols, float* out, float alpha, int accumulate) {
  if (!X || !out || rows <= 0 || cols <= 0 || ldx < cols) return A3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) { if (hipError_t e = hipMemsetAsync(out, 0, (size_t)cols * sizeof(float), s); e != hipSuccess) return (int)e; }
  if (cols % 320 == 0 && ldx % 8 == 0 && (reinterpret_cast<uintptr_t>(X) & 15u) == 0) {
    const uint16_t* x = (const uint16_t*)X;
    if (cols % 2560 == 0) return launch_colsum_rows<64>(s, x, ldx, rows, cols, out, alpha);
    if (cols % 1280 == 0) return launch_colsum_rows<32>(s, x, ldx, rows, cols, out, alpha);
    if (cols % 640 == 0) return launch_colsum_rows<16>(s, x, ldx, rows, cols, out, alpha);
    return launch_colsum_rows<8>(s, x, ldx, rows, cols, out, alpha);
  }
  const int64_t bx = (cols + 63) / 64;
  int64_t by = (rows + 255) / 256; if (by > 512) by = 512;
  const int64_t rpb = (rows + by - 1) / by;
  colsum_kernel<<<dim3((unsigned)bx, (unsigned)by), dim3(256), 0, s>>>((const uint16_t*)X, ldx, rows, cols, rpb, out, alpha);
  return a3d_launch_status();
}

extern "C" int A3D_FN(a3d_geglu_bwd)(a3d_stream_t stream, const void* P, int64_t ldp, const void* dY, int64_t lddy, void* dP, int64_t lddp, int64_t M, int64_t N) {
  if (!P || !dY || !dP || M <= 0 || N <= 0 || N % 32 != 0 || ldp % 8 || lddy % 8 || lddp % 8) return A3D_EINVAL;
  const int64_t n = M * (N / 8);
  geglu_bwd_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>((const uint16_t*)P, ldp, (const uint16_t*)dY, lddy, (uint16_t*)dP, lddp, M, N);
  return a3d_launch_status();
}

extern "C" int A3D_FN(a3d_axpby)(a3d_stream_t stream, const void* X, void* Y, int64_t n, float a, float b) {
  if (!X || !Y || n <= 0 || n % 8 != 0) return A3D_EINVAL;
  axpby_kernel<<<dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>((const uint16_t*)X, (uint16_t*)Y, n / 8, a, b);
  return a3d_launch_status();
}

extern "C" int A3D_FN(a3d_layer_norm_bwd)(a3d_stream_t stream, const void* X, const void* dY, const float* gamma, void* dX,
                                           float* dgamma, float* dbeta, int64_t M, int C, float eps, int accumulate) {
  if (!X || !dY || !gamma || !dX || M <= 0 || C <= 0 || C > 2048 || (dgamma == nullptr) != (dbeta == nullptr)) return A3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (dgamma && !accumulate) {
    if (hipError_t e = hipMemsetAsync(dgamma, 0, (size_t)C * sizeof(float), s); e != hipSuccess) return (int)e;
    if (hipError_t e = hipMemsetAsync(dbeta, 0, (size_t)C * sizeof(float), s); e != hipSuccess) return (int)e;
  }
  LNBParams p{(const uint16_t*)X, (const uint16_t*)dY, gamma, (uint16_t*)dX, dgamma, dbeta, M, C, eps};
  const bool al16 = ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(dX) | reinterpret_cast<uintptr_t>(gamma)) & 15u) == 0;
  if (al16 && C == 320) return launch_ln_bwd_rows<8>(s, p);
  if (al16 && C == 640) return launch_ln_bwd_rows<16>(s, p);
  if (al16 && C == 1280) return launch_ln_bwd_rows<32>(s, p);
  int64_t blocks = (M + 3) / 4; if (blocks > 2048) blocks = 2048;
  const int cpl = (C + 63) / 64;
  if (cpl <= 5) layer_norm_bwd_kernel<5><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(p);
  else if (cpl <= 10) layer_norm_bwd_kernel<10><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(p);
  else if (cpl <= 20) layer_norm_bwd_kernel<20><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(p);
  else layer_norm_bwd_kernel<32><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(p);
  return a3d_launch_status();
}

extern "C" int A3D_FN(a3d_group_norm_bwd)(a3d_stream_t stream, const void* X, const void* dY, const float* gamma, const float* beta,
                                           const float* stats, void* dX, float* ws, float* dgamma, float* dbeta,
                                           int B, int64_t rows, int C, int groups, int silu) {
  if (!X || !dY || !gamma || !beta || !stats || !dX || !ws || B <= 0 || rows <= 0 || C <= 0 || groups <= 0 || groups > 64 || C % groups != 0 || C % 8 != 0)
    return A3D_EINVAL;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return A3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipError_t e = hipMemsetAsync(ws, 0, (size_t)B * C * 2 * sizeof(float), s); e != hipSuccess) return (int)e;
  GNBParams p{(const uint16_t*)X, (const uint16_t*)dY, gamma, beta, stats, (uint16_t*)dX, ws, dgamma, dbeta, B, rows, C, groups, C / groups, silu, 0};
  if (B > 65535 || C / 8 > 1024) return A3D_EINVAL;
  const int nch = C / 8, rpb = nch >= 256 ? 1 : 256 / nch;
  const unsigned threads = (unsigned)(nch * rpb);
  // ~2048 workgroups over the B samples, each at least 8 row steps long
  int64_t bx = (2048 + B - 1) / B;
  const int64_t most = (rows + 8 * rpb - 1) / (8 * rpb);
  if (bx > most) bx = most;
  if (bx < 1) bx = 1;
  p.rows_per_block = (rows + bx - 1) / bx;
  bx = (rows + p.rows_per_block - 1) / p.rows_per_block;
  gn_bwd_sums_kernel<<<dim3((unsigned)bx, (unsigned)B), dim3(threads), (size_t)threads * 64, s>>>(p);
  gn_bwd_apply_kernel<<<dim3((unsigned)bx, (unsigned)B), dim3(threads), 0, s>>>(p);
  if (dgamma) gn_bwd_param_kernel<<<dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s>>>(p);
  return a3d_launch_status();
}

extern "C" int A3D_FN(a3d_zero_insert2x)(a3d_stream_t stream, const void* dY, void* Z, int B, int H, int W, int C) {
  if (!dY || !Z || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 != 0) return A3D_EINVAL;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t n = (int64_t)B * H * W * (C / 8);
  zero_insert_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>((const uint16_t*)dY, (uint16_t*)Z, B, H, W, Ho, Wo, C);
  return a3d_launch_status();
}

extern "C" int A3D_FN(a3d_upsample2x_bwd)(a3d_stream_t stream, const void* dU, void* dX, int B, int H, int W, int He, int We, int C) {
  if (!dU || !dX || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 != 0) return A3D_EINVAL;
  if ((He != 2 * H && He != 2 * H - 1) || (We != 2 * W && We != 2 * W - 1)) return A3D_EINVAL;
  const int64_t n = (int64_t)B * H * W * (C / 8);
  upsample_bwd_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>((const uint16_t*)dU, (uint16_t*)dX, B, H, W, He, We, C);
  return a3d_launch_status();
}

extern "C" int A3D_FN(a3d_temporal_attn_bwd)(a3d_stream_t stream, const void* Q, const void* K, const void* V, int64_t ldqkv,
                                              const void* dO, int64_t lddo, void* dQ, void* dK, void* dV, int64_t ldd,
                                              int videos, int frames, int64_t L, int heads, int head_dim, float scale) {
  if (!Q || !K || !V || !dO || !dQ || !dK || !dV || videos <= 0 || frames <= 0 || frames > 32 || L <= 0 || heads <= 0) return A3D_EINVAL;
  const int C = heads * head_dim;
  if (C % SLAB != 0 || ldqkv % 8 || lddo % 8 || ldd % 8 || ldqkv < C || lddo < C || ldd < C) return A3D_EINVAL;
  TABParams p{(const uint16_t*)Q, (const uint16_t*)K, (const uint16_t*)V, ldqkv, (const uint16_t*)dO, lddo,
              (uint16_t*)dQ, (uint16_t*)dK, (uint16_t*)dV, ldd, frames, L, scale, scale * 1.4426950408889634f, (int64_t)videos * L};
  hipStream_t s = (hipStream_t)stream;
  switch (head_dim) {
    case 40: return launch_tab_dp<1>(s, p, C);
    case 80: return launch_tab_dp<2>(s, p, C);
    case 160: return launch_tab_dp<4>(s, p, C);
    default: return A3D_EUNSUPPORTED;
  }
}

#ifndef A3D_STORAGE_F16
extern "C" int a3d_sqnorm_f32(a3d_stream_t stream, const float* g, int64_t n, float* out, int accumulate) {
  if (!g || !out || n <= 0) return A3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) { if (hipError_t e = hipMemsetAsync(out, 0, sizeof(float), s); e != hipSuccess) return (int)e; }
  int64_t blocks = (n + 2047) / 2048; if (blocks > 2048) blocks = 2048;
  sqnorm_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(g, n, out);
  return a3d_launch_status();
}

extern "C" int a3d_clip_ctrl_f32(a3d_stream_t stream, const float* sqnorm, float max_norm, float inv_loss_scale, float* ctrl) {
  if (!sqnorm || !ctrl) return A3D_EINVAL;
  clip_ctrl_kernel<<<dim3(1), dim3(1), 0, (hipStream_t)stream>>>(sqnorm, max_norm, inv_loss_scale, ctrl);
  return a3d_launch_status();
}

extern "C" int a3d_adamw_f32(a3d_stream_t stream, float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, float bias_corr1, float bias_corr2, const float* ctrl) {
  if (!p || !g || !m || !v || n <= 0 || bias_corr1 <= 0.f || bias_corr2 <= 0.f) return A3D_EINVAL;
  int64_t blocks = (n + 1023) / 1024; if (blocks > 4096) blocks = 4096;
  adamw_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, ctrl);
  return a3d_launch_status();
}
#endif
