// Flash attention forward for gfx950: O = softmax(Q K^T * scale) V per (group, head), bf16 in/out,
// fp32 scores/accumulators, no score matrix in memory.  Head dims 40 / 80 / 160 (SD1.5 levels).
//
// Row addressing goes through a3d_rowmap, so the reference's "(b n f) l c -> (b f) (n l) c"
// regrouping, the first-frame K/V selection of the I2V branch and the per-video text/IP tokens
// are all just different maps over the same [rows, C] tensors (no rearrange copies).
//
// Structure (one 256-thread workgroup = 4 waves; each wave owns 32 query rows; KV tiles of BKV keys):
//   * S^T = K · Q^T with v_mfma_f32_32x32x16_bf16: A = K rows from LDS (16-B reads, padded rows),
//     B = Q^T held in registers for the whole kernel.  The result layout gives every lane ONE
//     query (lane&31) and 16 keys per 32-key sub-tile, so row max is in-lane plus one exchange
//     with lane^32 — no LDS, no butterfly.
//   * The K row that feeds MFMA row i is permuted (kperm) so that the 8 scores a lane holds in
//     registers 8j..8j+7 are 8 CONSECUTIVE keys: P^T then is directly the B operand of
//     O^T = V^T · P^T (no cross-lane movement), and the matching A operand is one 16-B read of
//     a V^T image in LDS.  V is transposed while it is staged (4 keys x 8 dims per thread,
//     8-byte LDS writes).
//   * O^T accumulators keep the query in lane&31 too, so the online-softmax rescale is a plain
//     per-lane multiply — and it is skipped (wave-uniform branch) whenever no running max grew.
//   * Row sums come out of the matrix pipe: for D = 40 / 80 the V^T image has spare rows (O^T is
//     computed in 32-row tiles), one of them holds ones, so O^T[row D] = sum_k P — the same
//     bf16-rounded P that multiplies V, rescaled together with O.
//   * K/V images are double-buffered in LDS: tile t+1 is written (from registers filled during
//     the previous iteration) and tile t+2 is requested from HBM/L2 before tile t's MFMAs start;
//     one barrier per tile.  Per-thread source pointers advance incrementally (no div/mod, no
//     64-bit multiplies in the loop); only the last tile carries clamp + mask code.
//   * grid.x = heads * q_tiles with the head fastest: with 8 heads block b runs on XCD b%8 =
//     head, so all q-tiles of one (group, head) share one XCD's L2 copy of that K/V.
//   ALIGNED = a KV tile never straddles a row-map segment (seg_len % BKV == 0, or one segment);
//   the generic variant (small low-resolution levels only) recomputes rows with 32-bit div/mod.
#include <type_traits>

#include "common.h"

namespace {

constexpr int BQ = 128;

struct AttnParams {
  const uint16_t* Q; const uint16_t* K; const uint16_t* V; uint16_t* O;
  a3d_rowmap qm, km, om;
  int heads; int q_len, kv_len;
  float scale_log2, out_scale; int accumulate;
  float thr_raw;     // lazy-max threshold in raw score units
};

A3D_DEV int64_t map_row(const a3d_rowmap& m, int64_t g, int64_t s) {
  return (g / m.gdiv) * m.ga + (g % m.gdiv) * m.gb + (s / m.seg_len) * m.seg_stride + (s % m.seg_len);
}

// K row (within a 32-row sub-tile) that feeds MFMA A-row i: chosen so that result register r of a
// lane in half g is key 16*(r>>3) + 8*g + (r&7).
A3D_DEV int kperm(int i) {
  const int j = i & 3, g = (i >> 2) & 1, b = i >> 3;
  return 16 * (b >> 1) + 8 * g + 4 * (b & 1) + j;
}

// VAR bits: 1 = per-32-key-sub-tile online softmax (lets QK^T of the next sub-tile / PV of the previous one run
//           under the softmax VALU work), 2 = lazy running max (exchange + rescale only when a score exceeds the
//           running max by more than thr_raw), 4 = s_setprio(1) around MFMA groups.
constexpr int V_SUB = 1, V_LAZY = 2, V_PRIO = 4;

template <int D, int BKV, bool ALIGNED, int VAR>
__global__ __launch_bounds__(256, 2) void flash_attn_kernel(const AttnParams p) {
  constexpr int NU = BKV / 32;             // 32-key sub-tiles per KV tile
  constexpr int VROW = BKV + 8;            // V^T image row stride (elements): odd number of 16-B slots
  constexpr int DK = (D + 15) / 16 * 16;   // contraction length of Q K^T, zero padded
  constexpr int KS = DK / 16;
  constexpr int MT = (D + 31) / 32;        // 32-row tiles of O^T
  constexpr int KROW = DK + 8;             // K image row stride (elements); (DK+8)/8 is odd for 40/80/160
  constexpr int DCH = D / 8;               // 16-byte chunks per row
  constexpr int KCHUNKS = BKV * DCH;       // K staging chunks per tile
  constexpr int KPT = (KCHUNKS + 255) / 256;
  constexpr int VITEMS = (BKV / 4) * DCH;  // V staging items (4 keys x 8 dims)
  constexpr int VPT = (VITEMS + 255) / 256;
  constexpr bool ONES = MT * 32 > D;       // spare V^T row available for the row sums
  constexpr int KS_ELEMS = BKV * KROW, VT_ELEMS = MT * 32 * VROW;
  static_assert((KROW / 8) % 2 == 1 && (VROW / 8) % 2 == 1, "LDS row strides must be an odd number of 16-B slots");

  __shared__ __attribute__((aligned(16))) uint16_t smem[2 * (KS_ELEMS + VT_ELEMS)];
  uint16_t* const Ks0 = smem;
  uint16_t* const Vt0 = smem + 2 * KS_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, g = lane >> 5;
  const int head = blockIdx.x % p.heads;
  const int qt = blockIdx.x / p.heads;
  const int64_t grp = blockIdx.y;
  const int64_t hoff = (int64_t)head * D;

  // one-time LDS init: zero the contraction padding of both K images, ones row of both V^T images
  if constexpr (DK > D) {
    for (int i = tid; i < 2 * BKV * (DK - D); i += 256) {
      const int b = i / (BKV * (DK - D)), rem = i % (BKV * (DK - D));
      Ks0[b * KS_ELEMS + (rem / (DK - D)) * KROW + D + rem % (DK - D)] = 0;
    }
  }
  if constexpr (ONES) {
    for (int i = tid; i < 2 * BKV; i += 256) Vt0[(i / BKV) * VT_ELEMS + D * VROW + (i % BKV)] = 0x3F80;   // bf16 1.0
  }

  // ---- Q^T fragments: lane (q = l31, half g) holds Q[q][16*ks + 8*g .. +7]
  const int q_idx = qt * BQ + wid * 32 + l31;
  const bool q_ok = q_idx < p.q_len;
  const int64_t q_row = map_row(p.qm, grp, q_ok ? q_idx : p.q_len - 1);
  u32x4_t qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int d0 = 16 * ks + 8 * g;
    if (d0 < D) qf[ks] = *reinterpret_cast<const u32x4_t*>(p.Q + q_row * p.qm.ld + hoff + d0);
    else qf[ks] = u32x4_t{0u, 0u, 0u, 0u};
  }

  // ---- K/V staging: per-thread source pointers, advanced tile by tile
  const int64_t ld = p.km.ld;
  const int64_t kgbase = (grp / p.km.gdiv) * p.km.ga + (grp % p.km.gdiv) * p.km.gb;
  const uint16_t* const Kh = p.K + hoff;
  const uint16_t* const Vh = p.V + hoff;
  const uint32_t seg_len = (uint32_t)p.km.seg_len;
  const int64_t tile_step = (int64_t)BKV * ld;                                  // elements per tile
  const int64_t wrap_step = (p.km.seg_stride - p.km.seg_len) * ld;             // extra jump at a segment end
  int kr[KPT], kc[KPT], vq[VPT], vc[VPT];
  const uint16_t* kptr[KPT];
  const uint16_t* vptr[VPT];
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    const int c = tid + 256 * i;
    kr[i] = c / DCH; kc[i] = c % DCH;
    kptr[i] = Kh + (kgbase + kr[i]) * ld + kc[i] * 8;
  }
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int it = tid + 256 * i;
    vq[i] = it / DCH; vc[i] = it % DCH;
    vptr[i] = Vh + (kgbase + vq[i] * 4) * ld + vc[i] * 8;
  }
  uint32_t seg_off = 0;     // offset of the NEXT tile to load inside its segment (ALIGNED path)
  int next_kv0 = 0;         // first key of the next tile to load

  auto row_generic = [&](int s) -> int64_t {      // clamp + 32-bit div/mod (generic path, tails)
    if (s >= p.kv_len) s = p.kv_len - 1;
    const uint32_t seg = (uint32_t)s / seg_len;
    return kgbase + (int64_t)seg * p.km.seg_stride + ((uint32_t)s - seg * seg_len);
  };

  u32x4_t kreg[KPT];
  u32x4_t vreg[VPT][4];
  auto load_kv = [&](auto tail_c) {
    constexpr bool TAIL = decltype(tail_c)::value;
    if constexpr (ALIGNED && !TAIL) {
#pragma unroll
      for (int i = 0; i < KPT; ++i)
        if (tid + 256 * i < KCHUNKS) kreg[i] = *reinterpret_cast<const u32x4_t*>(kptr[i]);
#pragma unroll
      for (int i = 0; i < VPT; ++i)
        if (tid + 256 * i < VITEMS) {
#pragma unroll
          for (int r = 0; r < 4; ++r) vreg[i][r] = *reinterpret_cast<const u32x4_t*>(vptr[i] + r * ld);
        }
      // advance to the next tile (wave-uniform wrap test)
      seg_off += BKV;
      int64_t step = tile_step;
      if (seg_off >= seg_len) { step += wrap_step; seg_off = 0; }
#pragma unroll
      for (int i = 0; i < KPT; ++i) kptr[i] += step;
#pragma unroll
      for (int i = 0; i < VPT; ++i) vptr[i] += step;
    } else {
#pragma unroll
      for (int i = 0; i < KPT; ++i)
        if (tid + 256 * i < KCHUNKS)
          kreg[i] = *reinterpret_cast<const u32x4_t*>(Kh + row_generic(next_kv0 + kr[i]) * ld + kc[i] * 8);
#pragma unroll
      for (int i = 0; i < VPT; ++i)
        if (tid + 256 * i < VITEMS) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            vreg[i][r] = *reinterpret_cast<const u32x4_t*>(Vh + row_generic(next_kv0 + vq[i] * 4 + r) * ld + vc[i] * 8);
        }
    }
    next_kv0 += BKV;
  };
  auto store_kv = [&](int buf) {
    uint16_t* const Ks = Ks0 + buf * KS_ELEMS;
    uint16_t* const Vt = Vt0 + buf * VT_ELEMS;
#pragma unroll
    for (int i = 0; i < KPT; ++i)
      if (tid + 256 * i < KCHUNKS) *reinterpret_cast<u32x4_t*>(Ks + kr[i] * KROW + kc[i] * 8) = kreg[i];
#pragma unroll
    for (int i = 0; i < VPT; ++i)
      if (tid + 256 * i < VITEMS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // word j of each row holds dims 2j (lo) and 2j+1 (hi)
          const uint32_t w0 = vreg[i][0][j], w1 = vreg[i][1][j], w2 = vreg[i][2][j], w3 = vreg[i][3][j];
          u32x2_t even, odd;
          even[0] = (w0 & 0xffffu) | (w1 << 16);
          even[1] = (w2 & 0xffffu) | (w3 << 16);
          odd[0] = (w0 >> 16) | (w1 & 0xffff0000u);
          odd[1] = (w2 >> 16) | (w3 & 0xffff0000u);
          *reinterpret_cast<u32x2_t*>(Vt + (vc[i] * 8 + 2 * j) * VROW + vq[i] * 4) = even;
          *reinterpret_cast<u32x2_t*>(Vt + (vc[i] * 8 + 2 * j + 1) * VROW + vq[i] * 4) = odd;
        }
      }
  };

  f32x16_t oacc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[mt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int krow_off = kperm(l31) * KROW + 8 * g;
  const int vrow_off = l31 * VROW + 8 * g;

  auto compute = [&](int buf, int kv0, auto tail_c) {
    constexpr bool TAIL = decltype(tail_c)::value;
    const uint16_t* const Ks = Ks0 + buf * KS_ELEMS + krow_off;
    const uint16_t* const Vt = Vt0 + buf * VT_ELEMS + vrow_off;
    // ---- S^T = K · Q^T for the NU 32-key sub-tiles
    f32x16_t sacc[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[u][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u32x4_t kf = *reinterpret_cast<const u32x4_t*>(Ks + 32 * u * KROW + 16 * ks);
        sacc[u] = mfma32(kf, qf[ks], sacc[u]);
      }
    }
    if constexpr (TAIL) {      // keys past kv_len
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kv0 + 32 * u + 16 * (r >> 3) + 8 * g + (r & 7) >= p.kv_len) sacc[u][r] = -INFINITY;
    }
    if constexpr ((VAR & V_SUB) == 0) {
      // ---- whole-tile online softmax (log2 domain); the query lives in lane&31, its other half in lane^32
      float mx = sacc[0][0];
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[u][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (__any(mx > m_run)) {           // some running max grew: rescale everything held at the old max
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2);
        m_run = m_new;
        if constexpr (!ONES) l_run *= alpha;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[mt][r] *= alpha;
      }
      const float mneg = -m_run * p.scale_log2;
      u32x4_t pf[NU][2];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        float pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pv[r] = __builtin_amdgcn_exp2f(fmaf(sacc[u][r], p.scale_log2, mneg));
          if constexpr (!ONES) l_run += pv[r];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < 4; ++j) pf[u][h][j] = pack2bf(pv[8 * h + 2 * j], pv[8 * h + 2 * j + 1]);
      }
      // ---- O^T += V^T · P^T   (row D of V^T is all ones: O^T[D] accumulates the row sums)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const u32x4_t vf = *reinterpret_cast<const u32x4_t*>(Vt + 32 * mt * VROW + 32 * u + 16 * h);
            oacc[mt] = mfma32(vf, pf[u][h], oacc[mt]);
          }
    } else {
      // ---- one online-softmax step per 32-key sub-tile: sub-tile u's VALU work has the QK^T MFMAs of the later
      //      sub-tiles and the PV MFMAs of sub-tile u-1 in flight underneath it.
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        float mx = sacc[u][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[u][r]);
        bool grow;
        if constexpr ((VAR & V_LAZY) != 0) grow = mx > m_run + p.thr_raw;      // this lane's 16 keys only; the vote covers the pair
        else { mx = fmaxf(mx, __shfl_xor(mx, 32)); grow = mx > m_run; }
        if (__any(grow)) {
          if constexpr ((VAR & V_LAZY) != 0) mx = fmaxf(mx, __shfl_xor(mx, 32));
          const float m_new = fmaxf(m_run, mx);
          const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2);
          m_run = m_new;
          if constexpr (!ONES) l_run *= alpha;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[mt][r] *= alpha;
        }
        const float mneg = -m_run * p.scale_log2;
        float pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pv[r] = __builtin_amdgcn_exp2f(fmaf(sacc[u][r], p.scale_log2, mneg));
          if constexpr (!ONES) l_run += pv[r];
        }
        u32x4_t pf[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < 4; ++j) pf[h][j] = pack2bf(pv[8 * h + 2 * j], pv[8 * h + 2 * j + 1]);
        if constexpr ((VAR & V_PRIO) != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const u32x4_t vf = *reinterpret_cast<const u32x4_t*>(Vt + 32 * mt * VROW + 32 * u + 16 * h);
            oacc[mt] = mfma32(vf, pf[h], oacc[mt]);
          }
        if constexpr ((VAR & V_PRIO) != 0) __builtin_amdgcn_s_setprio(0);
      }
    }
  };

  const int ntiles = (p.kv_len + BKV - 1) / BKV;
  const bool has_tail = (p.kv_len % BKV) != 0;
  auto load_tile = [&](int t) {     // t = index of the tile being requested
    if (has_tail && t == ntiles - 1) load_kv(std::true_type{}); else load_kv(std::false_type{});
  };

  // ---- prologue: tile 0 into buffer 0, tile 1 in flight
  load_tile(0);
  store_kv(0);
  if (ntiles > 1) load_tile(1);
  __syncthreads();
  const int nfast = has_tail ? ntiles - 1 : ntiles;     // tiles the incremental-pointer path may load
  int t = 0;
  if constexpr (ALIGNED) {
    for (; t + 2 < nfast; ++t) {                         // steady state: no clamp, no mask, no div/mod
      store_kv((t + 1) & 1);                             // registers hold tile t+1 (requested one iteration ago)
      load_kv(std::false_type{});                        // tile t+2
      compute(t & 1, t * BKV, std::false_type{});
      __syncthreads();
    }
  }
  for (; t < ntiles - 1; ++t) {                          // generic / last iterations
    store_kv((t + 1) & 1);
    if (t + 2 < ntiles) load_tile(t + 2);
    compute(t & 1, t * BKV, std::false_type{});
    __syncthreads();
  }
  if (has_tail) compute((ntiles - 1) & 1, (ntiles - 1) * BKV, std::true_type{});
  else compute((ntiles - 1) & 1, (ntiles - 1) * BKV, std::false_type{});

  // ---- finalize: lane holds O[q = l31][d = 32*mt + 8*qd + 4*g + j]
  float l_tot;
  if constexpr (ONES) {
    constexpr int LM = D / 32, LR = ((D % 32) / 8) * 4;       // O^T row D sits in half g = 0, register LR of tile LM
    static_assert((D % 32) % 8 == 0, "row D must map to half 0");
    l_tot = __shfl(oacc[LM][LR], l31);
  } else {
    l_tot = l_run + __shfl_xor(l_run, 32);
  }
  const float inv = p.out_scale / l_tot;
  if (q_ok) {
    uint16_t* orow = p.O + map_row(p.om, grp, q_idx) * p.om.ld + hoff;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int d = 32 * mt + 8 * qd + 4 * g;
        if (d < D) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = oacc[mt][4 * qd + j] * inv;
          if (p.accumulate) {
            const u32x2_t prev = *reinterpret_cast<const u32x2_t*>(orow + d);
            v[0] += lo_bf(prev[0]); v[1] += hi_bf(prev[0]); v[2] += lo_bf(prev[1]); v[3] += hi_bf(prev[1]);
          }
          u32x2_t o;
          o[0] = pack2bf(v[0], v[1]);
          o[1] = pack2bf(v[2], v[3]);
          *reinterpret_cast<u32x2_t*>(orow + d) = o;
        }
      }
  }
}

bool map_ok(const a3d_rowmap* m, int head_dim) {
  return m && m->gdiv > 0 && m->seg_len > 0 && m->ld > 0 && m->ld % 8 == 0 && head_dim % 8 == 0;
}

int g_flash_variant = 0;   // a3d_tune_flash(); 0 = default

template <int D, int BKV, int VAR>
void launch(bool aligned, dim3 grid, hipStream_t s, const AttnParams& p) {
  if (aligned) flash_attn_kernel<D, BKV, true, VAR><<<grid, dim3(256), 0, s>>>(p);
  else flash_attn_kernel<D, BKV, false, 0><<<grid, dim3(256), 0, s>>>(p);
}

}  // namespace

extern "C" int a3d_tune_flash(int variant) {
  if (variant < 0 || variant > 5) return A3D_EINVAL;
  g_flash_variant = variant;
  return A3D_OK;
}

extern "C" int a3d_flash_attn_bf16(a3d_stream_t stream, const void* Q, const void* K, const void* V, void* O,
                                   const a3d_rowmap* qmap, const a3d_rowmap* kmap, const a3d_rowmap* omap,
                                   int groups, int heads, int head_dim, int64_t q_len, int64_t kv_len,
                                   float scale, float out_scale, int accumulate) {
  if (!Q || !K || !V || !O || groups <= 0 || heads <= 0 || q_len <= 0 || kv_len <= 0) return A3D_EINVAL;
  if (!map_ok(qmap, head_dim) || !map_ok(kmap, head_dim) || !map_ok(omap, head_dim)) return A3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(V)) & 15u) return A3D_EINVAL;
  if (reinterpret_cast<uintptr_t>(O) & 7u) return A3D_EINVAL;
  if (groups > 65535 || q_len > 0x3fffffffLL || kv_len > 0x3fffffffLL || kmap->seg_len > 0x3fffffffLL) return A3D_EINVAL;
  AttnParams p{};
  p.Q = (const uint16_t*)Q; p.K = (const uint16_t*)K; p.V = (const uint16_t*)V; p.O = (uint16_t*)O;
  p.qm = *qmap; p.km = *kmap; p.om = *omap;
  p.heads = heads; p.q_len = (int)q_len; p.kv_len = (int)kv_len;
  p.scale_log2 = scale * 1.4426950408889634f; p.out_scale = out_scale; p.accumulate = accumulate;
  const int q_tiles = (int)((q_len + BQ - 1) / BQ);
  p.thr_raw = 3.0f / p.scale_log2;        // lazy max: tolerate P up to 2^3
  const int var = g_flash_variant;
  const bool big = (var == 4 || var == 5) && head_dim == 40;
  const int bkv = head_dim == 160 ? 32 : (big ? 128 : 64);
  const bool aligned = (kmap->seg_len % bkv == 0) || (kv_len <= kmap->seg_len);
  const dim3 grid((unsigned)(heads * q_tiles), (unsigned)groups);
  hipStream_t s = (hipStream_t)stream;
  switch (head_dim) {
    case 40:
      switch (var) {
        case 1: launch<40, 64, V_SUB>(aligned, grid, s, p); break;
        case 2: launch<40, 64, V_SUB | V_LAZY>(aligned, grid, s, p); break;
        case 3: launch<40, 64, V_SUB | V_LAZY | V_PRIO>(aligned, grid, s, p); break;
        case 4: launch<40, 128, V_SUB | V_LAZY>(aligned, grid, s, p); break;
        case 5: launch<40, 128, V_SUB | V_LAZY | V_PRIO>(aligned, grid, s, p); break;
        default: launch<40, 64, 0>(aligned, grid, s, p); break;
      }
      break;
    case 80:
      switch (var) {
        case 0: launch<80, 64, 0>(aligned, grid, s, p); break;
        case 1: launch<80, 64, V_SUB>(aligned, grid, s, p); break;
        case 3: case 5: launch<80, 64, V_SUB | V_LAZY | V_PRIO>(aligned, grid, s, p); break;
        default: launch<80, 64, V_SUB | V_LAZY>(aligned, grid, s, p); break;
      }
      break;
    case 160:
      if (var == 0) launch<160, 32, 0>(aligned, grid, s, p); else launch<160, 32, V_SUB | V_LAZY>(aligned, grid, s, p);
      break;
    default: return A3D_EUNSUPPORTED;
  }
  return a3d_launch_status();
}
