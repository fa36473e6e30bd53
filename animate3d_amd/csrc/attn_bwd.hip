// Flash attention BACKWARD for gfx950 (training path, SURVEY.md §8 f4: the xformers memory_efficient_attention backward that
// train.py:576-590 runs through autograd).  Same row-map addressing as the forward (flash_attn.hip), head dims 40 / 80 / 160,
// 16-bit storage (bf16 / fp16 build), fp32 scores, statistics and accumulators, no score matrix in memory.
//
// Three passes over the same tile structure (one template, MODE):
//   MODE_STATS  per query row: lse2 = log2 sum_k 2^(s_qk * scale * log2 e)  and  delta = sum_k P_qk dP_qk  (= rowsum(dO ∘ O), but
//               computed from P and dP so that the forward output need not be kept: the IP-adapter call accumulates several
//               attentions into one buffer).
//   MODE_DQ     dQ = scale * dS K,            dS = P ∘ (dP - delta),  P = 2^(s' - lse2),  dP = dO V^T
//   MODE_DKV    dK = scale * dS^T Q,  dV = P^T dO   summed over the `q_per_kv` consecutive query groups that share one K/V
//               group (first-frame attention: the F frames of a video read frame 0's keys, attention_processor.py:409-418).
//
// Structure: a 256-thread workgroup = 4 waves; every wave owns 32 COLUMNS of the score tile for the whole kernel (queries in
// STATS / DQ, keys in DKV) and holds their two B operands in registers; the ROW side (keys, resp. queries) streams through LDS in
// tiles of BR rows.  Both score-shaped products of a tile use v_mfma_f32_32x32x16: A = row-side images [row][d] read with the
// forward's row permutation (kperm), so that the 16 results of a lane are two runs of 8 consecutive rows and P / dS, packed to
// 16 bits, ARE the B operand of the gradient products (contraction over rows) without any cross-lane movement; their A operand
// is a transposed image [d][row] of the row-side tensor, written while staging (4 rows x 8 dims per thread, v_perm_b32).
// First version of the training path: single-buffered LDS, div/mod row addressing on every tile — correct first, tuned later.
#include "common.h"

namespace {

constexpr int MODE_DQ = 0, MODE_DKV = 1, MODE_STATS = 2;

struct BwdParams {
  const uint16_t* Q; const uint16_t* K; const uint16_t* V; const uint16_t* dO;
  uint16_t* dQ; uint16_t* dK; uint16_t* dV;
  float* lse2; float* delta;              // [groups][heads][q_len]
  a3d_rowmap qm, km, dom, dqm, dkm;       // rows of Q, K|V, dO, dQ, dK|dV
  int heads; int q_len, kv_len; int q_per_kv;
  float scale, scale_log2, do_scale; int accumulate;
};

A3D_DEV int64_t map_row(const a3d_rowmap& m, int64_t g, int64_t s) {
  return (g / m.gdiv) * m.ga + (g % m.gdiv) * m.gb + (s / m.seg_len) * m.seg_stride + (s % m.seg_len);
}

// row (within a 32-row sub-tile) that feeds MFMA A-row i: result register r of a lane in half g then is row 16*(r>>3) + 8*g + (r&7)
A3D_DEV int kperm(int i) {
  const int j = i & 3, g = (i >> 2) & 1, b = i >> 3;
  return 16 * (b >> 1) + 8 * g + 4 * (b & 1) + j;
}

template <int D, int MODE, int NU>
__global__ __launch_bounds__(256, 1) void attn_bwd_kernel(const BwdParams p) {
  constexpr int BR = 32 * NU;              // rows per LDS tile
  constexpr int DK = (D + 15) / 16 * 16;   // contraction length of the score products, zero padded
  constexpr int KS = DK / 16;
  constexpr int MT = (D + 31) / 32;        // 32-row tiles of the transposed gradients
  constexpr int NROW = DK + 8;             // natural image row stride (elements): odd number of 16-B slots
  constexpr int TROW = BR + 8;             // transposed image row stride
  constexpr int DCH = D / 8;               // 16-byte chunks per row
  constexpr int N_ELEMS = BR * NROW, T_ELEMS = MT * 32 * TROW;
  constexpr int NT = (MODE == MODE_STATS) ? 0 : (MODE == MODE_DQ ? 1 : 2);
  static_assert((NROW / 8) % 2 == 1 && (TROW / 8) % 2 == 1, "LDS row strides must be an odd number of 16-B slots");

  __shared__ __attribute__((aligned(16))) uint16_t smem[2 * N_ELEMS + (NT > 0 ? NT : 1) * T_ELEMS];
  __shared__ __attribute__((aligned(16))) float rstat[2][BR];
  uint16_t* const N1 = smem;
  uint16_t* const N2 = smem + N_ELEMS;
  uint16_t* const T1 = smem + 2 * N_ELEMS;
  uint16_t* const T2 = T1 + (NT > 1 ? T_ELEMS : 0);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, g = lane >> 5;
  const int head = blockIdx.x % p.heads;
  const int ct = blockIdx.x / p.heads;
  const int64_t hoff = (int64_t)head * D;
  const int64_t grp_c = (MODE == MODE_DKV) ? (int64_t)blockIdx.y * p.q_per_kv : (int64_t)blockIdx.y;   // group the column maps see
  const int clen = (MODE == MODE_DKV) ? p.kv_len : p.q_len;
  const int rlen = (MODE == MODE_DKV) ? p.q_len : p.kv_len;

  // zero the contraction padding of the natural images once (staging never writes it)
  if constexpr (DK > D) {
    for (int i = tid; i < 2 * BR * (DK - D); i += 256) {
      const int b = i / (BR * (DK - D)), rem = i % (BR * (DK - D));
      smem[b * N_ELEMS + (rem / (DK - D)) * NROW + D + rem % (DK - D)] = 0;
    }
  }

  // ---- column operands (B fragments): lane (column l31, half g) holds X[col][16*ks + 8*g .. +7]
  const int col = ct * 128 + wid * 32 + l31;
  const bool col_ok = col < clen;
  const int colc = col_ok ? col : clen - 1;
  u32x4_t cA[KS], cB[KS];
  {
    const uint16_t* a_src; const uint16_t* b_src;
    if constexpr (MODE == MODE_DKV) {
      const int64_t row = map_row(p.km, grp_c, colc);
      a_src = p.K + row * p.km.ld + hoff; b_src = p.V + row * p.km.ld + hoff;
    } else {
      a_src = p.Q + map_row(p.qm, grp_c, colc) * p.qm.ld + hoff;
      b_src = p.dO + map_row(p.dom, grp_c, colc) * p.dom.ld + hoff;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = 16 * ks + 8 * g;
      if (d0 < D) {
        cA[ks] = *reinterpret_cast<const u32x4_t*>(a_src + d0);
        cB[ks] = *reinterpret_cast<const u32x4_t*>(b_src + d0);
      } else {
        cA[ks] = u32x4_t{0u, 0u, 0u, 0u};
        cB[ks] = u32x4_t{0u, 0u, 0u, 0u};
      }
    }
  }

  f32x16_t acc1[MT], acc2[MODE == MODE_DKV ? MT : 1];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[mt][r] = 0.f;
  if constexpr (MODE == MODE_DKV) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[mt][r] = 0.f;
  }
  float m_run = -INFINITY, l_run = 0.f, t_run = 0.f;     // MODE_STATS
  float lse_c = 0.f, dl_c = 0.f;                          // MODE_DQ: statistics of this lane's query
  if constexpr (MODE == MODE_DQ) {
    const int64_t si = ((int64_t)blockIdx.y * p.heads + head) * p.q_len + colc;
    lse_c = p.lse2[si]; dl_c = p.delta[si];
  }

  const int nrow_off = kperm(l31) * NROW + 8 * g;
  const int trow_off = l31 * TROW + 8 * g;

  // transposed staging of 4 rows x 8 dims: word j of a row holds dims 2j (lo) and 2j+1 (hi)
  auto store_t = [&](uint16_t* T, int ch, int rq, const u32x4_t (&w)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32x2_t even, odd;
      even[0] = __builtin_amdgcn_perm(w[1][j], w[0][j], 0x05040100u);
      even[1] = __builtin_amdgcn_perm(w[3][j], w[2][j], 0x05040100u);
      odd[0] = __builtin_amdgcn_perm(w[1][j], w[0][j], 0x07060302u);
      odd[1] = __builtin_amdgcn_perm(w[3][j], w[2][j], 0x07060302u);
      *reinterpret_cast<u32x2_t*>(T + (ch * 8 + 2 * j) * TROW + rq * 4) = even;
      *reinterpret_cast<u32x2_t*>(T + (ch * 8 + 2 * j + 1) * TROW + rq * 4) = odd;
    }
  };

  const int ngrp_r = (MODE == MODE_DKV) ? p.q_per_kv : 1;
  for (int gi = 0; gi < ngrp_r; ++gi) {
    const int64_t grp_r = (MODE == MODE_DKV) ? grp_c + gi : grp_c;      // group the row maps see
    for (int r0 = 0; r0 < rlen; r0 += BR) {
      // ---- stage the row tile: natural images of both row-side tensors, transposed image(s) for the gradient products
      for (int c = tid; c < BR * DCH; c += 256) {
        const int r = c / DCH, ch = c % DCH;
        int s = r0 + r; if (s >= rlen) s = rlen - 1;
        const uint16_t *s1, *s2;
        if constexpr (MODE == MODE_DKV) {
          s1 = p.Q + map_row(p.qm, grp_r, s) * p.qm.ld; s2 = p.dO + map_row(p.dom, grp_r, s) * p.dom.ld;
        } else {
          const int64_t row = map_row(p.km, grp_r, s);
          s1 = p.K + row * p.km.ld; s2 = p.V + row * p.km.ld;
        }
        *reinterpret_cast<u32x4_t*>(N1 + r * NROW + ch * 8) = *reinterpret_cast<const u32x4_t*>(s1 + hoff + ch * 8);
        *reinterpret_cast<u32x4_t*>(N2 + r * NROW + ch * 8) = *reinterpret_cast<const u32x4_t*>(s2 + hoff + ch * 8);
      }
      if constexpr (MODE != MODE_STATS) {
        for (int it = tid; it < (BR / 4) * DCH; it += 256) {
          const int rq = it / DCH, ch = it % DCH;
          u32x4_t w1[4], w2[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            int s = r0 + 4 * rq + j; if (s >= rlen) s = rlen - 1;
            if constexpr (MODE == MODE_DKV) {
              w1[j] = *reinterpret_cast<const u32x4_t*>(p.Q + map_row(p.qm, grp_r, s) * p.qm.ld + hoff + ch * 8);
              w2[j] = *reinterpret_cast<const u32x4_t*>(p.dO + map_row(p.dom, grp_r, s) * p.dom.ld + hoff + ch * 8);
            } else {
              w1[j] = *reinterpret_cast<const u32x4_t*>(p.K + map_row(p.km, grp_r, s) * p.km.ld + hoff + ch * 8);
            }
          }
          store_t(T1, ch, rq, w1);
          if constexpr (MODE == MODE_DKV) store_t(T2, ch, rq, w2);
        }
      }
      if constexpr (MODE == MODE_DKV) {
        if (tid < 2 * BR) {
          int s = r0 + (tid % BR); if (s >= rlen) s = rlen - 1;
          const int64_t si = (grp_r * p.heads + head) * (int64_t)p.q_len + s;
          rstat[tid / BR][tid % BR] = (tid < BR) ? p.lse2[si] : p.delta[si];
        }
      }
      __syncthreads();

#pragma unroll
      for (int u = 0; u < NU; ++u) {
        // ---- the two score-shaped products of this 32-row sub-tile
        f32x16_t s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const u32x4_t a1 = *reinterpret_cast<const u32x4_t*>(N1 + 32 * u * NROW + nrow_off + 16 * ks);
          const u32x4_t a2 = *reinterpret_cast<const u32x4_t*>(N2 + 32 * u * NROW + nrow_off + 16 * ks);
          s = mfma32(a1, cA[ks], s);
          dp = mfma32(a2, cB[ks], dp);
        }
        const int rbase = r0 + 32 * u + 8 * g;          // row of register r: rbase + 16*(r>>3) + (r&7)
        if constexpr (MODE == MODE_STATS) {
          float sv[16];
          float mx = -INFINITY;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            sv[r] = (rbase + 16 * (r >> 3) + (r & 7) < rlen) ? s[r] * p.scale_log2 : -INFINITY;
            mx = fmaxf(mx, sv[r]);
          }
          mx = fmaxf(mx, __shfl_xor(mx, 32));
          const float m_new = fmaxf(m_run, mx);            // finite: sub-tile 0 of every tile holds a valid row
          const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
          l_run *= alpha; t_run *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(sv[r] - m_new);
            l_run += e;
            t_run = fmaf(e, dp[r], t_run);
          }
          m_run = m_new;
        } else {
          float pv[16], ds[16];
          if constexpr (MODE == MODE_DQ) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const bool ok = rbase + 16 * (r >> 3) + (r & 7) < rlen;
              pv[r] = ok ? __builtin_amdgcn_exp2f(fmaf(s[r], p.scale_log2, -lse_c)) : 0.f;
              ds[r] = pv[r] * fmaf(dp[r], p.do_scale, -dl_c);
            }
          } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const float4 la = *reinterpret_cast<const float4*>(&rstat[0][32 * u + 16 * h + 8 * g]);
              const float4 lb = *reinterpret_cast<const float4*>(&rstat[0][32 * u + 16 * h + 8 * g + 4]);
              const float4 da = *reinterpret_cast<const float4*>(&rstat[1][32 * u + 16 * h + 8 * g]);
              const float4 db = *reinterpret_cast<const float4*>(&rstat[1][32 * u + 16 * h + 8 * g + 4]);
              const float ls[8] = {la.x, la.y, la.z, la.w, lb.x, lb.y, lb.z, lb.w};
              const float dl[8] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const int r = 8 * h + e;
                const bool ok = rbase + 16 * h + e < rlen;
                pv[r] = ok ? __builtin_amdgcn_exp2f(fmaf(s[r], p.scale_log2, -ls[e])) : 0.f;
                ds[r] = pv[r] * fmaf(dp[r], p.do_scale, -dl[e]);
              }
            }
          }
          u32x4_t dsf[2], pf[2];
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              dsf[h][j] = pack16(ds[8 * h + 2 * j], ds[8 * h + 2 * j + 1]);
              pf[h][j] = pack16(pv[8 * h + 2 * j], pv[8 * h + 2 * j + 1]);
            }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const u32x4_t t1 = *reinterpret_cast<const u32x4_t*>(T1 + 32 * mt * TROW + trow_off + 32 * u + 16 * h);
              acc1[mt] = mfma32(t1, dsf[h], acc1[mt]);                      // dQ^T += K^T dS^T   /   dK^T += Q^T dS
              if constexpr (MODE == MODE_DKV) {
                const u32x4_t t2 = *reinterpret_cast<const u32x4_t*>(T2 + 32 * mt * TROW + trow_off + 32 * u + 16 * h);
                acc2[mt] = mfma32(t2, pf[h], acc2[mt]);                     // dV^T += dO^T P
              }
            }
        }
      }
      __syncthreads();
    }
  }

  // ---- results: lane holds column l31; register r = 4*qd + j of tile mt is dim d = 32*mt + 8*qd + 4*g + j
  if constexpr (MODE == MODE_STATS) {
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float t_tot = t_run + __shfl_xor(t_run, 32);
    if (g == 0 && col_ok) {
      const int64_t si = ((int64_t)blockIdx.y * p.heads + head) * p.q_len + col;
      p.lse2[si] = m_run + __builtin_amdgcn_logf(l_tot);       // v_log_f32 = log2
      p.delta[si] = p.do_scale * t_tot / l_tot;
    }
  } else {
    if (!col_ok) return;
    auto write = [&](uint16_t* base, const a3d_rowmap& m, const f32x16_t (&acc)[MT], float mul) __attribute__((always_inline)) {
      uint16_t* dst = base + map_row(m, grp_c, col) * m.ld + hoff;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int d0 = 32 * mt + 8 * qd + 4 * g;
          if (d0 < D) {
            float v0 = acc[mt][4 * qd] * mul, v1 = acc[mt][4 * qd + 1] * mul, v2 = acc[mt][4 * qd + 2] * mul, v3 = acc[mt][4 * qd + 3] * mul;
            u32x2_t* o = reinterpret_cast<u32x2_t*>(dst + d0);
            if (p.accumulate) {
              const u32x2_t old = *o;
              v0 += lo16(old[0]); v1 += hi16(old[0]); v2 += lo16(old[1]); v3 += hi16(old[1]);
            }
            u32x2_t w;
            w[0] = pack16(v0, v1); w[1] = pack16(v2, v3);
            *o = w;
          }
        }
    };
    if constexpr (MODE == MODE_DQ) {
      write(p.dQ, p.dqm, acc1, p.scale);
    } else {
      write(p.dK, p.dkm, acc1, p.scale);
      write(p.dV, p.dkm, acc2, p.do_scale);
    }
  }
}

template <int D, int MODE, int NU>
int launch(hipStream_t s, const BwdParams& p, int groups_y) {
  const int clen = (MODE == MODE_DKV) ? p.kv_len : p.q_len;
  const int64_t ctiles = (clen + 127) / 128;
  if (ctiles * p.heads > 0x7fffffffLL || groups_y > 65535) return A3D_EINVAL;
  attn_bwd_kernel<D, MODE, NU><<<dim3((unsigned)(ctiles * p.heads), (unsigned)groups_y), dim3(256), 0, s>>>(p);
  return a3d_launch_status();
}

template <int MODE>
int dispatch(hipStream_t s, const BwdParams& p, int head_dim, int groups_y) {
  switch (head_dim) {
    case 40: return launch<40, MODE, 2>(s, p, groups_y);
    case 64: return launch<64, MODE, 2>(s, p, groups_y);
    case 80: return launch<80, MODE, 2>(s, p, groups_y);
    case 160: return launch<160, MODE, 1>(s, p, groups_y);
    default: return A3D_EUNSUPPORTED;
  }
}

bool map_ok(const a3d_rowmap* m, int head_dim, int heads) {
  return m && m->gdiv > 0 && m->seg_len > 0 && m->ld >= (int64_t)heads * head_dim && m->ld % 8 == 0;
}

}  // namespace

extern "C" int A3D_FN(a3d_flash_attn_bwd)(a3d_stream_t stream, const void* Q, const void* K, const void* V, const void* dO,
                                           void* dQ, void* dK, void* dV, float* lse2, float* delta,
                                           const a3d_rowmap* qmap, const a3d_rowmap* kmap, const a3d_rowmap* domap,
                                           const a3d_rowmap* dqmap, const a3d_rowmap* dkmap,
                                           int groups, int heads, int head_dim, int64_t q_len, int64_t kv_len, int q_per_kv,
                                           float scale, float do_scale, int accumulate) {
  if (!Q || !K || !V || !dO || !lse2 || !delta) return A3D_EINVAL;
  if (!dQ && !dK) return A3D_EINVAL;
  if ((dK == nullptr) != (dV == nullptr)) return A3D_EINVAL;
  if (groups <= 0 || heads <= 0 || q_len <= 0 || kv_len <= 0 || q_len > 0x7fffffffLL || kv_len > 0x7fffffffLL) return A3D_EINVAL;
  if (q_per_kv <= 0 || groups % q_per_kv != 0) return A3D_EINVAL;
  if (!map_ok(qmap, head_dim, heads) || !map_ok(kmap, head_dim, heads) || !map_ok(domap, head_dim, heads)) return A3D_EINVAL;
  if (dQ && !map_ok(dqmap, head_dim, heads)) return A3D_EINVAL;
  if (dK && !map_ok(dkmap, head_dim, heads)) return A3D_EINVAL;
  BwdParams p;
  p.Q = (const uint16_t*)Q; p.K = (const uint16_t*)K; p.V = (const uint16_t*)V; p.dO = (const uint16_t*)dO;
  p.dQ = (uint16_t*)dQ; p.dK = (uint16_t*)dK; p.dV = (uint16_t*)dV;
  p.lse2 = lse2; p.delta = delta;
  p.qm = *qmap; p.km = *kmap; p.dom = *domap;
  p.dqm = dQ ? *dqmap : *qmap; p.dkm = dK ? *dkmap : *kmap;
  p.heads = heads; p.q_len = (int)q_len; p.kv_len = (int)kv_len; p.q_per_kv = q_per_kv;
  p.scale = scale; p.scale_log2 = scale * 1.4426950408889634f; p.do_scale = do_scale; p.accumulate = accumulate;
  hipStream_t s = (hipStream_t)stream;
  if (int rc = dispatch<MODE_STATS>(s, p, head_dim, groups)) return rc;
  if (dQ) { if (int rc = dispatch<MODE_DQ>(s, p, head_dim, groups)) return rc; }
  if (dK) { if (int rc = dispatch<MODE_DKV>(s, p, head_dim, groups / q_per_kv)) return rc; }
  return A3D_OK;
}
