// GEMM and implicit-GEMM 3x3 convolution for gfx950 (bf16 in, fp32 accumulate, bf16 out).
//
//   Y[M,N] = alpha * (A[M,K] · W[N,K]^T + bias[N] + rowbias[m / rb_div][N]) + beta * R[M,N]
//
// A is either a dense row-major matrix (Linear / 1x1 conv) or the virtual im2col matrix of an
// NHWC image batch (3x3 conv, pad 1, stride 1|2, optional nearest-2x upsample folded into the
// addressing).  One 256-thread workgroup (4 waves, 2x2) owns a 128x128 output tile; each wave
// owns 64x64 = 2x2 MFMA 32x32x16 tiles.  K advances in steps of 64 through a double-buffered,
// padded LDS image (row stride 144 B = 9 x 16-B slots, odd => conflict-free ds_read_b128);
// the next K-tile is fetched global->registers while the current one feeds the MFMAs, one
// barrier per K-step.  The MFMA is issued as D^T = W · A^T so that every lane ends up with 4
// consecutive output columns per register quad (8-byte stores, bias/residual as 8-byte loads).
// The 1-D grid is remapped so that each XCD walks a contiguous range of tiles (tiles that
// share the same A rows are neighbours => A is fetched from HBM once per XCD, W stays in L2).
#include "gemm_common.h"

namespace {

constexpr int BM = 128, BN = 128;
// K-step BKT = 64 (two LDS stages = 72 KB, 2 workgroups per CU) for long contractions, BKT = 32 (40 KB, 3 per CU:
// more independent workgroups to cover the per-tile load latency and epilogue) for the short K = 320 / 640 GEMMs.
template <int BKT> struct TileCfg {
  static constexpr int LROW = BKT + 8;                  // LDS row stride in elements: odd number of 16-B slots
  static constexpr int TILE_ELEMS = 128 * LROW;         // one operand tile
  static constexpr int STAGE_BYTES = 2 * 2 * TILE_ELEMS * 2;   // [buf][A|W]
  static constexpr int EPI_BYTES = 4 * 32 * 68 * 4;     // epilogue staging, one 32-row half per wave at a time
  static constexpr int SMEM_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
  static constexpr int CPR = BKT / 8;                   // 16-byte chunks per tile row
  static constexpr int RPT = 256 / CPR;                 // rows covered by one pass of the 256 threads
  static constexpr int NPASS = 128 / RPT;               // chunks per thread per operand
};



// CONV: 0 = dense A, 1 = 3x3 conv gather (pad 1, stride 1|2), 2 = 3x3 conv over a nearest-2x upsampled input
template <int CONV, int EPI, int BKT, bool RES>
__global__ __launch_bounds__(256, (BKT == 64 ? 2 : 3)) void gemm_kernel(const GemmParams p) {
  using TC = TileCfg<BKT>;
  constexpr int LROW = TC::LROW, TILE_ELEMS = TC::TILE_ELEMS, NPASS = TC::NPASS, RPT = TC::RPT, BK = BKT;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, g = lane >> 5;

  const int64_t nblk = p.tiles_n * p.tiles_m;
  const int64_t lid = xcd_remap(blockIdx.x, nblk);
  const int64_t tile_n = lid % p.tiles_n, tile_m = lid / p.tiles_n;
  const int64_t m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- staging assignment: thread owns 16-byte chunk column kc of rows srow + 32*i
  const int kc = tid % TC::CPR;
  const int srow = tid / TC::CPR;

  // per-row source bookkeeping for A
  const uint16_t* a_ptr[NPASS];   // dense: row pointer (+kc*8); conv: unused
  int a_b[CONV == 2 ? NPASS : 1], a_y[CONV == 2 ? NPASS : 1], a_x[CONV == 2 ? NPASS : 1];   // up2x conv: output pixel coordinates
  bool a_ok[NPASS];
  int a_mask[NPASS];
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    int64_t m = m0 + srow + RPT * i;
    a_ok[i] = m < p.M;
    if (m >= p.M) m = p.M - 1;
    if constexpr (CONV != 0) {
      const int hw = p.Ho * p.Wo;
      const int b = (int)(m / hw);
      const int rem = (int)(m - (int64_t)b * hw);
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      if constexpr (CONV == 2) {
        a_b[i] = b; a_y[i] = oy; a_x[i] = ox;
        a_ptr[i] = nullptr; a_mask[i] = 0;
      } else {
        // tap (0,0) source position and a 9-bit in-bounds mask: per K-tile only a uniform tap offset is added
        const int y0 = oy * p.stride - 1, x0 = ox * p.stride - 1;
        a_ptr[i] = p.X + (((int64_t)b * p.H + y0) * p.Wd + x0) * p.Cin + kc * 8;
        int mask = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int yy = y0 + t / 3, xx = x0 + t % 3;
          if (a_ok[i] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd) mask |= 1 << t;
        }
        a_mask[i] = mask;
      }
    } else {
      a_ptr[i] = p.X + m * p.ldx + kc * 8;
      a_mask[i] = 0;
    }
  }
  const uint16_t* w_ptr[NPASS];
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    int64_t n = n0 + srow + RPT * i;
    if (n >= p.N) n = p.N - 1;
    w_ptr[i] = p.W + n * p.ldw + kc * 8;
  }

  struct RegTile { u32x4_t a[NPASS], w[NPASS]; };
  RegTile rt0, rt1;     // two K-tiles in flight (prefetch distance 2)
  auto load_tile = [&](int64_t k0, RegTile& rt) {
    u32x4_t (&ra)[NPASS] = rt.a; u32x4_t (&rw)[NPASS] = rt.w;
    if constexpr (CONV != 0) {
      {   // the persistent kernel's K walk (gemm_pp.hip): nine taps of one 64-channel slice, then the next slice; k0 becomes the
          // column of W ([tap][Cin]) that the walk visits at linear position k0 — same accumulation order, bit-identical results
        const int kt64 = (int)(k0 >> 6);
        const int c64 = kt64 / 9, t = kt64 - 9 * c64;
        k0 = (int64_t)t * p.Cin + c64 * 64 + (k0 & 63);
      }
      const int tap = (int)(k0 / p.Cin);
      const int ci0 = (int)(k0 - (int64_t)tap * p.Cin);
      const int ky = tap / 3, kx = tap - ky * 3;
      if constexpr (CONV == 1) {       // uniform tap offset + per-row bit test
        const int64_t toff = ((int64_t)ky * p.Wd + kx) * p.Cin + ci0;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
          if ((a_mask[i] >> tap) & 1) ra[i] = *reinterpret_cast<const u32x4_t*>(a_ptr[i] + toff);
          else ra[i] = u32x4_t{0u, 0u, 0u, 0u};
        }
      } else {                         // nearest-2x upsample folded into the address (3 convs per step)
        const int He = p.He, We = p.We;          // 2H x 2W, or one less when the caller forces the output size
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
          const int yy = a_y[i] + ky - 1, xx = a_x[i] + kx - 1;
          const bool ok = a_ok[i] && yy >= 0 && yy < He && xx >= 0 && xx < We;
          if (ok) {
            const uint16_t* src = p.X + (((int64_t)a_b[i] * p.H + (yy >> 1)) * p.Wd + (xx >> 1)) * p.Cin + ci0 + kc * 8;
            ra[i] = *reinterpret_cast<const u32x4_t*>(src);
          } else {
            ra[i] = u32x4_t{0u, 0u, 0u, 0u};
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NPASS; ++i) ra[i] = *reinterpret_cast<const u32x4_t*>(a_ptr[i] + k0);
    }
#pragma unroll
    for (int i = 0; i < NPASS; ++i) rw[i] = *reinterpret_cast<const u32x4_t*>(w_ptr[i] + k0);
  };
  auto store_tile = [&](int buf, const RegTile& rt) {
    const u32x4_t (&ra)[NPASS] = rt.a; const u32x4_t (&rw)[NPASS] = rt.w;
    uint16_t* As = smem + (buf * 2 + 0) * TILE_ELEMS;
    uint16_t* Ws = smem + (buf * 2 + 1) * TILE_ELEMS;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      *reinterpret_cast<u32x4_t*>(As + (srow + RPT * i) * LROW + kc * 8) = ra[i];
      *reinterpret_cast<u32x4_t*>(Ws + (srow + RPT * i) * LROW + kc * 8) = rw[i];
    }
  };

  f32x16_t acc[2][2];   // [tn][tm]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // residual rows this lane will need in the epilogue (row = 8*j + lane/8 of each 32-row half, 8 columns at
  // 8*(lane&7)): requested during the LAST K-step so their HBM latency hides under its MFMAs
  u32x4_t rres[RES ? 4 : 1];
  const int ecc = lane & 7;
  const int64_t en = n0 + wn * 64 + 8 * ecc;
  const bool epf = RES && (EPI == EPI_LINEAR) && p.R != nullptr && p.vec16 && en + 8 <= p.N;
  auto prefetch_residual = [&](int tm) {      // half tm = 0 during the last K-step, half 1 while half 0 is written out
    if constexpr (RES) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int64_t m = m0 + wm * 64 + tm * 32 + 8 * j + (lane >> 3);
        if (m >= p.M) m = p.M - 1;
        rres[j] = *reinterpret_cast<const u32x4_t*>(p.R + m * p.ldr + en);
      }
    }
  };

  const int64_t nk = p.K / BK;
  auto compute = [&](int cur) {
    const uint16_t* As = smem + (cur * 2 + 0) * TILE_ELEMS;
    const uint16_t* Ws = smem + (cur * 2 + 1) * TILE_ELEMS;
    // fragments of k-step ks+1 are requested before the MFMAs of k-step ks are issued (register double buffer)
    u32x4_t fw[2][2], fa[2][2];
    auto load_frags = [&](int slot, int ks) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        fw[slot][t] = *reinterpret_cast<const u32x4_t*>(Ws + (wn * 64 + t * 32 + l31) * LROW + ks * 16 + g * 8);
        fa[slot][t] = *reinterpret_cast<const u32x4_t*>(As + (wm * 64 + t * 32 + l31) * LROW + ks * 16 + g * 8);
      }
    };
    load_frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      if (ks + 1 < BK / 16) load_frags((ks + 1) & 1, ks + 1);
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) acc[tn][tm] = mfma32(fw[ks & 1][tn], fa[ks & 1][tm], acc[tn][tm]);
    }
  };

  // K pipeline: LDS is double-buffered, and TWO further K-tiles are in flight in registers (rt0 / rt1), so a
  // tile's global loads have two K-steps of MFMA time to land (one step was shorter than the HBM/L2 latency).
  load_tile(0, rt0);
  store_tile(0, rt0);
  if (nk > 1) load_tile(1 * BK, rt0);
  if (nk > 2) load_tile(2 * BK, rt1);
  __syncthreads();
  for (int64_t kt = 0;;) {
    // even step: LDS[0] = tile kt, rt0 = tile kt+1, rt1 = tile kt+2 (in flight)
    if (kt + 1 >= nk && epf) prefetch_residual(0);
    compute(0);
    if (kt + 1 < nk) store_tile(1, rt0);
    if (kt + 3 < nk) load_tile((kt + 3) * BK, rt0);
    __syncthreads();
    if (++kt >= nk) break;
    // odd step: LDS[1] = tile kt, rt1 = tile kt+1, rt0 = tile kt+2 (in flight)
    if (kt + 1 >= nk && epf) prefetch_residual(0);
    compute(1);
    if (kt + 1 < nk) store_tile(0, rt1);
    if (kt + 3 < nk) load_tile((kt + 3) * BK, rt1);
    __syncthreads();
    if (++kt >= nk) break;
  }

  // ---- epilogue.  Each wave transposes its 64x64 fp32 sub-tile through a private LDS region (row stride 68
  //      floats: conflict-free ds_write_b128 from the MFMA layout) and re-reads it row-major, so that every lane
  //      owns 8 consecutive output columns: bias / rowbias / residual / output are 16-byte accesses and one wave
  //      instruction touches 8 full 128-byte row segments (the direct MFMA-layout epilogue issued 8-byte stores
  //      to 32 different rows per instruction and was store-issue bound at K = 320).
  constexpr int SROW = 68;
  float* const stg = reinterpret_cast<float*>(smem) + wid * (32 * SROW);     // one 32-row half (tm) at a time
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
  if (tm == 1) __syncthreads();                                              // half 0 fully read before it is overwritten
#pragma unroll
  for (int tn = 0; tn < 2; ++tn)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 v;
      v.x = acc[tn][tm][4 * q]; v.y = acc[tn][tm][4 * q + 1]; v.z = acc[tn][tm][4 * q + 2]; v.w = acc[tn][tm][4 * q + 3];
      *reinterpret_cast<float4*>(stg + l31 * SROW + tn * 32 + 8 * q + 4 * g) = v;
    }
  __syncthreads();
  const int64_t mbase = m0 + wm * 64 + tm * 32;

  if constexpr (EPI == EPI_GEGLU) {
    // columns [0,32) of the wave's sub-tile are h, [32,64) the matching gates (weight rows are interleaved on
    // the host): out[m][j] = (h + b_h) * gelu_erf(gate + b_g); 4 lanes x 8 columns per output row.
    const int cc = lane & 3;
    const int64_t nh = n0 + wn * 64 + 8 * cc;                  // column of h in the interleaved N space
    const int64_t oc = (n0 + wn * 64) / 2 + 8 * cc;            // output column
    if (nh + 32 < p.N + 0 && oc + 8 <= p.N / 2) {
      float bh[8], bg[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { bh[e] = p.bias ? p.bias[nh + e] : 0.f; bg[e] = p.bias ? p.bias[nh + 32 + e] : 0.f; }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = 16 * j + (lane >> 2);
        const int64_t m = mbase + row;
        if (m >= p.M) continue;
        const float4 h0 = *reinterpret_cast<const float4*>(stg + row * SROW + 8 * cc);
        const float4 h1 = *reinterpret_cast<const float4*>(stg + row * SROW + 8 * cc + 4);
        const float4 g0 = *reinterpret_cast<const float4*>(stg + row * SROW + 32 + 8 * cc);
        const float4 g1 = *reinterpret_cast<const float4*>(stg + row * SROW + 32 + 8 * cc + 4);
        const float hv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
        const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        float y[8];
        geglu8(hv, gv, bh, bg, y);
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack16(y[2 * e], y[2 * e + 1]);
        *reinterpret_cast<u32x4_t*>(p.Y + m * p.ldy + oc) = o;
      }
    }
  } else {
    const int cc = lane & 7;
    const int64_t n = n0 + wn * 64 + 8 * cc;
    if (n < p.N) {
      const bool full = (n + 8 <= p.N) && p.vec16;            // else: N % 8 == 4 tail or unaligned rows -> 8-byte halves
      float bv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[e] = (p.bias && n + e < p.N) ? p.bias[n + e] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = 8 * j + (lane >> 3);
        const int64_t m = mbase + row;
        if (m >= p.M) continue;
        const float4 a0 = *reinterpret_cast<const float4*>(stg + row * SROW + 8 * cc);
        const float4 a1 = *reinterpret_cast<const float4*>(stg + row * SROW + 8 * cc + 4);
        float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bv[e];
        const uint16_t* rb = p.rowbias ? p.rowbias + (m / p.rb_div) * p.N + n : nullptr;
        const uint16_t* rr = p.R ? p.R + m * p.ldr + n : nullptr;
        uint16_t* yy = p.Y + m * p.ldy + n;
        if (full) {
          if (rb) {
            const u32x4_t t = *reinterpret_cast<const u32x4_t*>(rb);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] += lo16(t[e]); v[2 * e + 1] += hi16(t[e]); }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = epi_scale(v[e], p.alpha);
          if constexpr (RES) {
            if (rr) {
              const u32x4_t t = rres[j];            // prefetched (epf is true whenever this branch is taken)
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[2 * e] = epi_axpy(v[2 * e], p.beta, lo16(t[e])); v[2 * e + 1] = epi_axpy(v[2 * e + 1], p.beta, hi16(t[e])); }
            }
          }
          if (p.out_f32) {
            float* yf = reinterpret_cast<float*>(p.Y) + m * p.ldy + n;
            *reinterpret_cast<float4*>(yf) = float4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<float4*>(yf + 4) = float4{v[4], v[5], v[6], v[7]};
          } else {
            u32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pack16(v[2 * e], v[2 * e + 1]);
            *reinterpret_cast<u32x4_t*>(yy) = o;
          }
        } else {
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            if (n + 4 * hh + 4 > p.N) continue;
            float w[4] = {v[4 * hh], v[4 * hh + 1], v[4 * hh + 2], v[4 * hh + 3]};
            if (rb) {
              const u32x2_t t = *reinterpret_cast<const u32x2_t*>(rb + 4 * hh);
              w[0] += lo16(t[0]); w[1] += hi16(t[0]); w[2] += lo16(t[1]); w[3] += hi16(t[1]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = epi_scale(w[e], p.alpha);
            if (rr) {
              const u32x2_t t = *reinterpret_cast<const u32x2_t*>(rr + 4 * hh);
              w[0] = epi_axpy(w[0], p.beta, lo16(t[0])); w[1] = epi_axpy(w[1], p.beta, hi16(t[0]));
              w[2] = epi_axpy(w[2], p.beta, lo16(t[1])); w[3] = epi_axpy(w[3], p.beta, hi16(t[1]));
            }
            u32x2_t o;
            o[0] = pack16(w[0], w[1]);
            o[1] = pack16(w[2], w[3]);
            *reinterpret_cast<u32x2_t*>(yy + 4 * hh) = o;
          }
        }
      }
    }
  }
  if (tm == 0 && epf) prefetch_residual(1);
  }   // tm halves
}


// =====================================================================================================================
// The big token matrices (levels 0-2 of the UNet: M = 32768 ... 524288, 97 % of the GEMM / conv FLOPs) take the persistent
// 256 x (NB*64) kernel of gemm_pp.hip (one 512-thread workgroup per CU, LDS-DMA staged K-tiles, ping-pong main loop).
// returns -1000 when the shape is not eligible (caller falls back to the 128x128 kernel)
constexpr int PBM = 256;
constexpr int PERSIST_MIN_FILL = 50;     // minimum average CU fill (per cent) of the persistent grid's rounds: at 50 % (level 3, 128 tiles)
                                         // it still ties or beats the 128x128 kernel by 3-10 % (profiles/README.md, round 1)
// Tile width and split-K factor of a persistent launch.  Returns false when the shape is not the persistent kernel's.  Cost model of one
// candidate (nb, S): rounds of the grid x (K-tiles per item + a fixed ~12 K-tiles of epilogue / turn-around) x tile width; S > 1 (only with
// a workspace, EPI_LINEAR, 16-bit output) must beat S = 1 by 15 % and leave >= 9 K-tiles per item, conv items hold whole 64-channel slices.
struct PPPlan { int nb, S, cus; int64_t tiles_m, tiles_n; };
constexpr int SPLITK_MAX = 16;
#ifndef A3D_EXP_SPLITK_MINK
#define A3D_EXP_SPLITK_MINK 9          // fewest K-tiles a split-K work item may hold (measurement builds override it)
#endif
template <int CONV, int EPI>
bool plan_persist(const GemmParams& p, int flags, bool allow_split, PPPlan& out) {
  if (a3d_gemm_kernel_of(flags) == A3D_GEMM_TILE128 || p.out_f32) return false;
  static int cus_of[64] = {0};
  const int dev = a3d_current_device();
  if (cus_of[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    cus_of[dev] = n > 0 ? n : 256;
  }
  // the persistent grid leaves the caller's reserved CUs free (the sharded path while an RCCL all-gather is in flight: its
  // kernels need CUs of their own to overlap with the GEMMs; animate3d_amd/parallel.py)
  const int reserved = a3d_gemm_reserved_cus_of(flags);
  const int cus = cus_of[dev] - reserved > 32 ? cus_of[dev] - reserved : 32;
  if (!p.vec16 || p.K % 64 != 0 || p.M % PBM != 0 || (p.rowbias && p.rb_div % PBM != 0)) return false;
  // 32-bit DMA offsets
  if (p.ldw % 64 != 0 || (CONV == 0 && p.ldx % 64 != 0)) return false;
  if (CONV == 0 && (uint64_t)p.ldx * 16u >= (1ull << 31)) return false;
  if ((uint64_t)p.ldw * 16u >= (1ull << 31)) return false;
  if (CONV != 0 && ((uint64_t)p.B * p.H * p.Wd + 2u * p.Wd + 2u) * (uint64_t)p.Cin * 2u >= (1ull << 32)) return false;
  const int64_t tm = (p.M + PBM - 1) / PBM;
  const int nk = (int)(p.K / 64);
  const int unit = CONV ? 9 : 1;                       // K-tiles an item may be cut at
  // the PLAN is made for the whole chip whatever the caller reserves: the split factor fixes the order of the K sum, and a launch's result must
  // not depend on how many CUs an in-flight all-gather was given (a smaller grid walks the same work items; tests/test_unet_gpu.py)
  const int cus_plan = cus_of[dev];
  constexpr int OVH = 12;
  int64_t best = -1, best1 = -1;
  int bnb = 0, bS = 1, bnb1 = 0;
  for (int nb = 5; nb >= 4; --nb) {
    if (EPI == EPI_GEGLU && nb == 5) continue;                  // (h | gate pairs: 64-column blocks)
    if (p.N % (nb * 64) != 0) continue;
    const int64_t tiles = tm * (p.N / (nb * 64));
    for (int S = 1; S <= (allow_split && EPI == EPI_LINEAR ? SPLITK_MAX : 1); ++S) {
      if ((nk / unit) % S != 0 || (S > 1 && nk / S < A3D_EXP_SPLITK_MINK)) continue;
      const int64_t items = tiles * S;
      const int64_t rounds = (items + cus_plan - 1) / cus_plan;
      if (items * 100 < rounds * cus_plan * PERSIST_MIN_FILL) continue;             // average fill of the rounds (per cent)
      if (S > 1 && p.ws != nullptr && items * nb * 65536 > p.ws_bytes) continue;    // (workspace too small for this factor)
      // both tile widths divide N (1280, 2560, 3840 ...): the 256 x 320 tile stages fewer operand bytes per FLOP and is the default (8 % handicap
      // for the narrower one), but when the grid is only a round or two (level 3: M = 8192) the narrower tile can fill more CUs
      const int64_t cost = rounds * (nk / S + OVH) * nb * (nb == 4 ? 108 : 100) + (S > 1 ? 100 * nb : 0);
      if (S == 1 && (best1 < 0 || cost < best1)) { best1 = cost; bnb1 = nb; }
      if (best < 0 || cost < best) { best = cost; bnb = nb; bS = S; }
    }
  }
  if (best < 0) return false;
  if (bS > 1 && best1 >= 0 && best * 100 > best1 * 85) { bnb = bnb1; bS = 1; }      // split-K has to be worth its reduce pass
  out.nb = bnb; out.S = bS; out.cus = cus;
  out.tiles_m = tm; out.tiles_n = p.N / (bnb * 64);
  return true;
}

// Tile width of a ring-kernel launch (gemm_ring.hip).  A lone workgroup per CU is bound by the CU's LDS-DMA fill rate, so a launch costs
// rounds x (128 + BN) per K-tile: the widest tile that keeps the number of rounds lowest wins.  The choice may depend on anything (CU
// reservation included): every width walks K in the same order, results are bit-identical.
struct RingPlan { int nb, cus, cus_plan; int64_t cost; };
#ifndef A3D_RING_VS_PP_PCT
#define A3D_RING_VS_PP_PCT 75           // a persistent-kernel tile costs ~0.75 x (256 + BN) of the ring kernel's units per K-tile and round (measured: at equal
                                        // CU fill the two tie at M = 8192, N = 1280 although the ring tile is 128 rows; profiles/r6_microbench_smallm_ring.log)
#endif
constexpr int RING_VS_PP_PCT = A3D_RING_VS_PP_PCT;
inline bool plan_ring(const GemmParams& p, int flags, RingPlan& out) {
  if (a3d_gemm_kernel_of(flags) == A3D_GEMM_TILE128 || a3d_gemm_kernel_of(flags) == A3D_GEMM_DIRECT || p.out_f32 || !p.vec16 || p.X2 != nullptr) return false;
  if (p.M % 128 != 0 || p.K % 64 != 0 || p.K < 256 || p.ldx % 64 != 0 || p.ldw % 64 != 0) return false;
  if ((uint64_t)p.ldx * 64u >= (1ull << 32) || (uint64_t)p.ldw * 256u >= (1ull << 32)) return false;      // 32-bit piece offsets: 3 x 16 ldx, 9 x 16 ldw bytes
  if (p.rowbias && p.rb_div % 128 != 0) return false;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, a3d_current_device()) != hipSuccess || n <= 0) n = 256;
  const int reserved = a3d_gemm_reserved_cus_of(flags);
  out.cus_plan = n;
  out.cus = n - reserved > 32 ? n - reserved : 32;
  out.nb = 0; out.cost = 0;
  for (int nb : {5, 4, 2}) {
    if (p.N % (64 * nb) != 0) continue;
    const int64_t tiles = (p.M / 128) * (p.N / (64 * nb));
    const int64_t rounds = (tiles + out.cus - 1) / out.cus;
    const int64_t cost = rounds * (128 + 64 * nb);
    if (out.nb == 0 || cost < out.cost) { out.nb = nb; out.cost = cost; }
  }
  return out.nb != 0;
}

template <int CONV, int EPI>
int try_launch_persist(hipStream_t stream, GemmParams& p, int flags) {
  PPPlan pl;
  if (!plan_persist<CONV, EPI>(p, flags, p.ws != nullptr, pl)) return -1000;
  p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n;
  p.ksplit = pl.S; p.nk_item = (int)(p.K / 64) / pl.S;
  p.direct = (CONV == 0 && EPI == EPI_LINEAR && pl.S == 1 && p.X2 == nullptr && a3d_gemm_kernel_of(flags) == A3D_GEMM_DIRECT) ? 1 : 0;
  return A3D_FN(a3d_launch_gemm_pp)(CONV, EPI, pl.nb, stream, p, pl.cus);
}


template <int CONV, int EPI, int BKT, bool RES>
int launch_res(hipStream_t stream, GemmParams& p, int64_t nblk) {
  using TC = TileCfg<BKT>;
  static uint64_t attr_done = 0;
  if (int rc = a3d_once_per_device(attr_done, [] {
        return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<CONV, EPI, BKT, RES>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, TC::SMEM_BYTES); })) return rc;
  gemm_kernel<CONV, EPI, BKT, RES><<<dim3((unsigned)nblk), dim3(256), TC::SMEM_BYTES, stream>>>(p);
  return a3d_launch_status();
}

template <int CONV, int EPI, int BKT>
int launch_bk(hipStream_t stream, GemmParams& p, int64_t nblk) {
  if constexpr (EPI == EPI_LINEAR) {
    if (p.R != nullptr) return launch_res<CONV, EPI, BKT, true>(stream, p, nblk);
  }
  return launch_res<CONV, EPI, BKT, false>(stream, p, nblk);
}

template <int CONV, int EPI = EPI_LINEAR>
int launch(hipStream_t stream, GemmParams& p, int flags) {
  if (flags & ~(A3D_GEMM_RESERVED_CUS_MASK | A3D_GEMM_KERNEL_MASK)) return A3D_EINVAL;
  if (a3d_gemm_kernel_of(flags) > A3D_GEMM_DIRECT) return A3D_EINVAL;
  if constexpr (CONV == 0 && EPI == EPI_LINEAR) {
    // small token matrices: the LDS-DMA ring kernel (gemm_ring.hip) when the persistent grid would be under-filled or lose to it
    RingPlan rp;
    if (plan_ring(p, flags, rp)) {
      PPPlan pl;
      bool ring = true;
      if (a3d_gemm_kernel_of(flags) != A3D_GEMM_RING && plan_persist<CONV, EPI>(p, flags, p.ws != nullptr, pl)) {
        const int64_t tiles = pl.tiles_m * pl.tiles_n * pl.S;
        const int64_t pp_cost = ((tiles + rp.cus_plan - 1) / rp.cus_plan) * (256 + 64 * pl.nb) * RING_VS_PP_PCT;
        ring = pl.S == 1 && rp.cost * 100 < pp_cost;
      }
      if (ring) {
        p.tiles_m = p.M / 128; p.tiles_n = p.N / (64 * rp.nb);
        return A3D_FN(a3d_launch_gemm_ring)(rp.nb, stream, p, rp.cus);
      }
    }
  }
  {
    const int rc = try_launch_persist<CONV, EPI>(stream, p, flags);
    if (rc != -1000) return rc;
  }
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  const int64_t nblk = p.tiles_m * p.tiles_n;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return A3D_EINVAL;
  // measured on MI355X (profiles/r1_microbench_gemm_conv_v4.log): K-step 32 (3-4 workgroups per CU) wins for dense
  // K <= 640 and whenever the grid is under ~3 workgroups per CU; K-step 64 wins elsewhere (all 3x3 convs)
  const bool small = (CONV == 0 && p.K <= 640) || nblk < 768;
  if (small) return launch_bk<CONV, EPI, 32>(stream, p, nblk);
  return launch_bk<CONV, EPI, 64>(stream, p, nblk);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

namespace {
// workspace query of the *_ws entry points: the bytes a split-K launch of this call would use (0: the call does not split)
template <int CONV>
int64_t splitk_ws_bytes(const GemmParams& p, int flags) {
  PPPlan pl;
  GemmParams q = p;
  q.ws = nullptr; q.ws_bytes = 0;
  if (flags & ~(A3D_GEMM_RESERVED_CUS_MASK | A3D_GEMM_KERNEL_MASK)) return 0;
  if (!plan_persist<CONV, EPI_LINEAR>(q, flags, true, pl) || pl.S <= 1) return 0;
  return pl.tiles_m * pl.tiles_n * pl.S * pl.nb * 65536;
}
}  // namespace

static int gemm_entry(a3d_stream_t stream, const void* X, int64_t ldx, const void* W, int64_t ldw,
                      const float* bias, const void* rowbias, int64_t rb_div, const void* R, int64_t ldr,
                      void* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, float alpha, float beta, int flags,
                      void* ws, int64_t ws_bytes, int64_t* ws_needed) {
  if (!X || !W || !Y || M <= 0 || N <= 0 || K <= 0) return A3D_EINVAL;
  if (K % 64 != 0 || N % 4 != 0) return A3D_EINVAL;
  if (ldx % 8 != 0 || ldw % 8 != 0 || ldy % 4 != 0 || (R && ldr % 4 != 0)) return A3D_EINVAL;
  if (!aligned16(X) || !aligned16(W) || (reinterpret_cast<uintptr_t>(Y) & 7u) || (R && (reinterpret_cast<uintptr_t>(R) & 7u)))
    return A3D_EINVAL;
  if (bias && (reinterpret_cast<uintptr_t>(bias) & 15u)) return A3D_EINVAL;
  if (rowbias && (rb_div <= 0 || (reinterpret_cast<uintptr_t>(rowbias) & 7u))) return A3D_EINVAL;
  GemmParams p{};
  p.X = (const uint16_t*)X; p.ldx = ldx; p.W = (const uint16_t*)W; p.ldw = ldw;
  p.bias = bias; p.rowbias = (const uint16_t*)rowbias; p.rb_div = rowbias ? rb_div : 1;
  p.R = (const uint16_t*)R; p.ldr = ldr; p.Y = (uint16_t*)Y; p.ldy = ldy;
  p.M = M; p.N = N; p.K = K; p.alpha = alpha; p.beta = beta;
  p.vec16 = (ldy % 8 == 0) && aligned16(Y) && (!R || (ldr % 8 == 0 && aligned16(R))) && (!rowbias || (N % 8 == 0 && aligned16(rowbias)));
  if (ws_needed) { *ws_needed = splitk_ws_bytes<0>(p, flags); return 0; }
  if (ws && ws_bytes > 0 && aligned16(ws)) { p.ws = (float*)ws; p.ws_bytes = ws_bytes; }
  return launch<0>((hipStream_t)stream, p, flags);
}

extern "C" int A3D_FN(a3d_gemm)(a3d_stream_t stream, const void* X, int64_t ldx, const void* W, int64_t ldw,
                             const float* bias, const void* rowbias, int64_t rb_div, const void* R, int64_t ldr,
                             void* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, float alpha, float beta, int flags) {
  return gemm_entry(stream, X, ldx, W, ldw, bias, rowbias, rb_div, R, ldr, Y, ldy, M, N, K, alpha, beta, flags, nullptr, 0, nullptr);
}

extern "C" int A3D_FN(a3d_gemm_ws)(a3d_stream_t stream, const void* X, int64_t ldx, const void* W, int64_t ldw,
                                const float* bias, const void* rowbias, int64_t rb_div, const void* R, int64_t ldr,
                                void* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, float alpha, float beta, int flags,
                                void* ws, int64_t ws_bytes, int64_t* ws_needed) {
  return gemm_entry(stream, X, ldx, W, ldw, bias, rowbias, rb_div, R, ldr, Y, ldy, M, N, K, alpha, beta, flags, ws, ws_bytes, ws_needed);
}

// Y = [X | X2] W^T + bias with the A operand in two pieces (columns [0, K1) from X, [K1, K) from X2): persistent kernel only — the caller
// concatenates and calls a3d_gemm when this returns A3D_EUNSUPPORTED (small or ragged M, unaligned rows)
extern "C" int A3D_FN(a3d_gemm2)(a3d_stream_t stream, const void* X, int64_t ldx, const void* X2, int64_t ldx2, int64_t K1,
                              const void* W, int64_t ldw, const float* bias, void* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, int flags) {
  if (!X || !X2 || !W || !Y || M <= 0 || N <= 0 || K <= 0 || K1 <= 0 || K1 >= K) return A3D_EINVAL;
  if (K % 64 != 0 || K1 % 64 != 0 || N % 8 != 0) return A3D_EINVAL;
  if (ldx % 8 != 0 || ldx2 % 8 != 0 || ldw % 8 != 0 || ldy % 8 != 0) return A3D_EINVAL;
  if (!aligned16(X) || !aligned16(X2) || !aligned16(W) || !aligned16(Y)) return A3D_EINVAL;
  if (bias && (reinterpret_cast<uintptr_t>(bias) & 15u)) return A3D_EINVAL;
  if (flags & ~(A3D_GEMM_RESERVED_CUS_MASK | A3D_GEMM_KERNEL_MASK)) return A3D_EINVAL;
  if (ldx2 % 64 != 0 || (uint64_t)ldx2 * 16u >= (1ull << 31)) return A3D_EUNSUPPORTED;      // (32-bit DMA offsets, as for X)
  GemmParams p{};
  p.X = (const uint16_t*)X; p.ldx = ldx; p.X2 = (const uint16_t*)X2; p.ldx2 = ldx2; p.K1 = K1;
  p.W = (const uint16_t*)W; p.ldw = ldw; p.bias = bias; p.rb_div = 1; p.Y = (uint16_t*)Y; p.ldy = ldy;
  p.M = M; p.N = N; p.K = K; p.alpha = 1.f; p.beta = 0.f; p.vec16 = 1;
  const int rc = try_launch_persist<0, EPI_LINEAR>((hipStream_t)stream, p, flags);
  return rc == -1000 ? A3D_EUNSUPPORTED : rc;
}

extern "C" int A3D_FN(a3d_gemm_f32out)(a3d_stream_t stream, const void* X, int64_t ldx, const void* W, int64_t ldw,
                                    const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, float alpha) {
  if (!X || !W || !Y || M <= 0 || N <= 0 || K <= 0) return A3D_EINVAL;
  if (K % 64 != 0 || N % 8 != 0 || ldx % 8 != 0 || ldw % 8 != 0 || ldy % 4 != 0) return A3D_EINVAL;
  if (!aligned16(X) || !aligned16(W) || !aligned16(Y)) return A3D_EINVAL;
  if (bias && (reinterpret_cast<uintptr_t>(bias) & 15u)) return A3D_EINVAL;
  GemmParams p{};
  p.X = (const uint16_t*)X; p.ldx = ldx; p.W = (const uint16_t*)W; p.ldw = ldw;
  p.bias = bias; p.rb_div = 1; p.Y = (uint16_t*)Y; p.ldy = ldy;
  p.M = M; p.N = N; p.K = K; p.alpha = alpha; p.beta = 0.f; p.vec16 = 1; p.out_f32 = 1;
  return launch<0>((hipStream_t)stream, p, 0);
}

static int conv_entry(a3d_stream_t stream, const void* X, const void* Wp, const float* bias,
                      const void* rowbias, int64_t rb_div, const void* R, void* Y,
                      int B, int H, int W, int Cin, int Cout, int stride, int up2x, int flags,
                      void* ws, int64_t ws_bytes, int64_t* ws_needed) {
  if (!X || !Wp || !Y || B <= 0 || H <= 0 || W <= 0) return A3D_EINVAL;
  if (Cin % 64 != 0 || Cout % 4 != 0 || (stride != 1 && stride != 2)) return A3D_EINVAL;
  if (up2x < 0 || up2x > 7 || (up2x > 1 && !(up2x & 1)) || (up2x && stride != 1)) return A3D_EINVAL;
  if (!aligned16(X) || !aligned16(Wp) || (reinterpret_cast<uintptr_t>(Y) & 7u)) return A3D_EINVAL;
  if (bias && (reinterpret_cast<uintptr_t>(bias) & 15u)) return A3D_EINVAL;
  if (rowbias && rb_div <= 0) return A3D_EINVAL;
  // up2x bit 0: nearest-2x upsample in front of the conv; bits 1 / 2: the upsampled image is cropped by its last row /
  // column (nearest interpolation to the forced size 2H-1 / 2W-1 of unet_motion_mv_model.py:831-837 is exactly that)
  const int He = up2x ? 2 * H - ((up2x >> 1) & 1) : H, We = up2x ? 2 * W - ((up2x >> 2) & 1) : W;
  if (He <= 0 || We <= 0) return A3D_EINVAL;
  GemmParams p{};
  p.X = (const uint16_t*)X; p.ldx = Cin; p.W = (const uint16_t*)Wp; p.ldw = (int64_t)9 * Cin;
  p.bias = bias; p.rowbias = (const uint16_t*)rowbias; p.rb_div = rowbias ? rb_div : 1;
  p.R = (const uint16_t*)R; p.ldr = Cout; p.Y = (uint16_t*)Y; p.ldy = Cout;
  p.B = B; p.H = H; p.Wd = W; p.Cin = Cin; p.stride = stride; p.up = up2x ? 1 : 0; p.He = He; p.We = We;
  p.Ho = (He + 2 - 3) / stride + 1; p.Wo = (We + 2 - 3) / stride + 1;
  p.M = (int64_t)B * p.Ho * p.Wo; p.N = Cout; p.K = (int64_t)9 * Cin;
  p.alpha = 1.f; p.beta = 1.f;
  p.vec16 = (Cout % 8 == 0) && aligned16(Y) && (!R || aligned16(R)) && (!rowbias || aligned16(rowbias));
  if (ws_needed) { *ws_needed = up2x ? splitk_ws_bytes<2>(p, flags) : splitk_ws_bytes<1>(p, flags); return 0; }
  if (ws && ws_bytes > 0 && aligned16(ws)) { p.ws = (float*)ws; p.ws_bytes = ws_bytes; }
  return up2x ? launch<2>((hipStream_t)stream, p, flags) : launch<1>((hipStream_t)stream, p, flags);
}

extern "C" int A3D_FN(a3d_conv3x3)(a3d_stream_t stream, const void* X, const void* Wp, const float* bias,
                                const void* rowbias, int64_t rb_div, const void* R, void* Y,
                                int B, int H, int W, int Cin, int Cout, int stride, int up2x, int flags) {
  return conv_entry(stream, X, Wp, bias, rowbias, rb_div, R, Y, B, H, W, Cin, Cout, stride, up2x, flags, nullptr, 0, nullptr);
}

extern "C" int A3D_FN(a3d_conv3x3_ws)(a3d_stream_t stream, const void* X, const void* Wp, const float* bias,
                                   const void* rowbias, int64_t rb_div, const void* R, void* Y,
                                   int B, int H, int W, int Cin, int Cout, int stride, int up2x, int flags,
                                   void* ws, int64_t ws_bytes, int64_t* ws_needed) {
  return conv_entry(stream, X, Wp, bias, rowbias, rb_div, R, Y, B, H, W, Cin, Cout, stride, up2x, flags, ws, ws_bytes, ws_needed);
}

extern "C" int A3D_FN(a3d_gemm_geglu)(a3d_stream_t stream, const void* X, int64_t ldx, const void* W, int64_t ldw,
                                   const float* bias, void* Y, int64_t ldy, int64_t M, int64_t N2, int64_t K, int flags) {
  if (!X || !W || !Y || M <= 0 || N2 <= 0 || K <= 0) return A3D_EINVAL;
  if (K % 64 != 0 || N2 % 64 != 0 || ldx % 8 != 0 || ldw % 8 != 0 || ldy % 8 != 0) return A3D_EINVAL;
  if (!aligned16(X) || !aligned16(W) || !aligned16(Y)) return A3D_EINVAL;
  GemmParams p{};
  p.X = (const uint16_t*)X; p.ldx = ldx; p.W = (const uint16_t*)W; p.ldw = ldw;
  p.bias = bias; p.rb_div = 1; p.Y = (uint16_t*)Y; p.ldy = ldy;
  p.M = M; p.N = N2; p.K = K; p.alpha = 1.f; p.beta = 0.f; p.vec16 = 1;
  return launch<0, EPI_GEGLU>((hipStream_t)stream, p, flags);
}
