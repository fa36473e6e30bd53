// Flash attention forward for gfx950: O = softmax(Q K^T * scale) V per (group, head), bf16 in/out,
// fp32 scores/accumulators, no score matrix in memory.  Head dims 40 / 80 / 160 (SD1.5 levels).
//
// Row addressing goes through a3d_rowmap, so the reference's "(b n f) l c -> (b f) (n l) c"
// regrouping, the first-frame K/V selection of the I2V branch and the per-video text/IP tokens
// are all just different maps over the same [rows, C] tensors (no rearrange copies).
//
// Structure (one 256-thread workgroup = 4 waves; each wave owns 32 query rows, KV tiles of 64):
//   * S^T = K · Q^T with v_mfma_f32_32x32x16_bf16: A = K rows from LDS (16-B reads, padded rows),
//     B = Q^T held in registers for the whole kernel.  The result layout gives every lane ONE
//     query (lane&31) and 16 keys, so row max / row sum are in-lane plus one exchange with
//     lane^32 — no LDS, no butterfly.
//   * The K row that feeds MFMA row i is permuted (kperm) so that the 8 scores a lane holds in
//     registers 8j..8j+7 are 8 CONSECUTIVE keys: P^T then is directly the B operand of
//     O^T = V^T · P^T (no cross-lane movement), and the matching A operand is one 16-B read of
//     a V^T image in LDS.  V is transposed while it is staged (4 keys x 8 dims per thread,
//     8-byte LDS writes).
//   * O^T accumulators keep the query in lane&31 too, so the online-softmax rescale is a plain
//     per-lane multiply.
//   * K/V tile t+1 is fetched global->registers while tile t is being consumed.
//   * grid.x = heads * q_tiles with the head fastest: with 8 heads block b runs on XCD b%8 =
//     head, so all q-tiles of one (group, head) share one XCD's L2 copy of that K/V.
#include "common.h"

namespace {

constexpr int BQ = 128;

struct AttnParams {
  const uint16_t* Q; const uint16_t* K; const uint16_t* V; uint16_t* O;
  a3d_rowmap qm, km, om;
  int heads; int64_t q_len, kv_len;
  float scale_log2, out_scale; int accumulate;
  int q_tiles; int kv_aligned;
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

template <int D, int BKV>
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
  static_assert((KROW / 8) % 2 == 1, "K row stride must be an odd number of 16-B slots");

  __shared__ __attribute__((aligned(16))) uint16_t Ks[BKV * KROW];
  __shared__ __attribute__((aligned(16))) uint16_t Vt[MT * 32 * VROW];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, g = lane >> 5;
  const int head = blockIdx.x % p.heads;
  const int qt = blockIdx.x / p.heads;
  const int64_t grp = blockIdx.y;
  const int64_t hoff = (int64_t)head * D;

  // zero the contraction padding of the K image once (columns D..DK-1 are never staged)
  if constexpr (DK > D) {
    for (int i = tid; i < BKV * (DK - D); i += 256) {
      const int r = i / (DK - D), c = i % (DK - D);
      Ks[r * KROW + D + c] = 0;
    }
  }

  // ---- Q^T fragments: lane (q = l31, half g) holds Q[q][16*ks + 8*g .. +7]
  const int64_t q_idx = (int64_t)qt * BQ + wid * 32 + l31;
  const bool q_ok = q_idx < p.q_len;
  const int64_t q_row = map_row(p.qm, grp, q_ok ? q_idx : p.q_len - 1);
  u32x4_t qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int d0 = 16 * ks + 8 * g;
    if (d0 < D) qf[ks] = *reinterpret_cast<const u32x4_t*>(p.Q + q_row * p.qm.ld + hoff + d0);
    else qf[ks] = u32x4_t{0u, 0u, 0u, 0u};
  }

  // ---- K/V row addressing.  Fast path (kv_aligned): a 64-key tile never straddles a segment, so
  //      the tile's first row is wave-uniform and advanced incrementally; otherwise 32-bit
  //      div/mod per staged row (only the small low-resolution levels take that path).
  const int64_t kgbase = (grp / p.km.gdiv) * p.km.ga + (grp % p.km.gdiv) * p.km.gb;
  const uint32_t kseg_len = (uint32_t)p.km.seg_len;
  int64_t tile_base = kgbase;      // row of the tile's first key (aligned path)
  uint32_t tile_off = 0;           // its offset inside the segment
  const uint16_t* Kh = p.K + hoff;
  const uint16_t* Vh = p.V + hoff;
  auto kv_row = [&](int64_t kv0, int r) -> int64_t {
    int64_t s = kv0 + r;
    if (s >= p.kv_len) s = p.kv_len - 1;
    if (p.kv_aligned) return tile_base + (s - kv0);
    const uint32_t su = (uint32_t)s;
    const uint32_t seg = su / kseg_len;
    return kgbase + (int64_t)seg * p.km.seg_stride + (su - seg * kseg_len);
  };

  // ---- staging registers
  u32x4_t kreg[KPT];
  u32x4_t vreg[VPT][4];
  auto load_kv = [&](int64_t t) {
    const int64_t kv0 = t * BKV;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int c = tid + 256 * i;
      if (c < KCHUNKS) {
        const int r = c / DCH, ch = c % DCH;
        kreg[i] = *reinterpret_cast<const u32x4_t*>(Kh + kv_row(kv0, r) * p.km.ld + ch * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int it = tid + 256 * i;
      if (it < VITEMS) {
        const int qd = it / DCH, ch = it % DCH;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          vreg[i][r] = *reinterpret_cast<const u32x4_t*>(Vh + kv_row(kv0, qd * 4 + r) * p.km.ld + ch * 8);
      }
    }
    // advance the uniform tile cursor
    tile_off += BKV;
    tile_base += BKV;
    if (tile_off >= kseg_len) { tile_base += p.km.seg_stride - (int64_t)tile_off; tile_off = 0; }
  };
  auto store_kv = [&]() {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int c = tid + 256 * i;
      if (c < KCHUNKS) {
        const int r = c / DCH, ch = c % DCH;
        *reinterpret_cast<u32x4_t*>(Ks + r * KROW + ch * 8) = kreg[i];
      }
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int it = tid + 256 * i;
      if (it < VITEMS) {
        const int qd = it / DCH, ch = it % DCH;
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // word j of each row holds dims 2j (lo) and 2j+1 (hi)
          const uint32_t w0 = vreg[i][0][j], w1 = vreg[i][1][j], w2 = vreg[i][2][j], w3 = vreg[i][3][j];
          u32x2_t even, odd;
          even[0] = (w0 & 0xffffu) | (w1 << 16);
          even[1] = (w2 & 0xffffu) | (w3 << 16);
          odd[0] = (w0 >> 16) | (w1 & 0xffff0000u);
          odd[1] = (w2 >> 16) | (w3 & 0xffff0000u);
          *reinterpret_cast<u32x2_t*>(Vt + (ch * 8 + 2 * j) * VROW + qd * 4) = even;
          *reinterpret_cast<u32x2_t*>(Vt + (ch * 8 + 2 * j + 1) * VROW + qd * 4) = odd;
        }
      }
    }
  };

  f32x16_t oacc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[mt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int64_t ntiles = (p.kv_len + BKV - 1) / BKV;
  const int krow = kperm(l31);
  load_kv(0);

  for (int64_t t = 0; t < ntiles; ++t) {
    __syncthreads();          // everyone is done reading the previous tile
    store_kv();
    __syncthreads();
    if (t + 1 < ntiles) load_kv(t + 1);

    // ---- S^T = K · Q^T for the two 32-key sub-tiles
    f32x16_t sacc[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[u][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u32x4_t kf = *reinterpret_cast<const u32x4_t*>(Ks + (32 * u + krow) * KROW + 16 * ks + 8 * g);
        sacc[u] = mfma32(kf, qf[ks], sacc[u]);
      }
    }
    // ---- mask keys past kv_len (only the last tile can have any)
    const int64_t kv0 = t * BKV;
    if (kv0 + BKV > p.kv_len) {
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t kv = kv0 + 32 * u + 16 * (r >> 3) + 8 * g + (r & 7);
          if (kv >= p.kv_len) sacc[u][r] = -INFINITY;
        }
    }
    // ---- online softmax (log2 domain); the query lives in lane&31, its other half in lane^32
    float mx = sacc[0][0];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[u][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2);
    const float mneg = -m_new * p.scale_log2;
    m_run = m_new;
    float psum = 0.f;
    u32x4_t pf[NU][2];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pv[r] = __builtin_amdgcn_exp2f(fmaf(sacc[u][r], p.scale_log2, mneg));
        psum += pv[r];
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) pf[u][h][j] = pack2bf(pv[8 * h + 2 * j], pv[8 * h + 2 * j + 1]);
    }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[mt][r] *= alpha;
    // ---- O^T += V^T · P^T
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const u32x4_t vf = *reinterpret_cast<const u32x4_t*>(Vt + (32 * mt + l31) * VROW + 32 * u + 16 * h + 8 * g);
          oacc[mt] = mfma32(vf, pf[u][h], oacc[mt]);
        }
  }

  // ---- finalize: lane holds O[q = l31][d = 32*mt + 8*qd + 4*g + j]
  const float l_tot = l_run + __shfl_xor(l_run, 32);
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

}  // namespace

extern "C" int a3d_flash_attn_bf16(a3d_stream_t stream, const void* Q, const void* K, const void* V, void* O,
                                   const a3d_rowmap* qmap, const a3d_rowmap* kmap, const a3d_rowmap* omap,
                                   int groups, int heads, int head_dim, int64_t q_len, int64_t kv_len,
                                   float scale, float out_scale, int accumulate) {
  if (!Q || !K || !V || !O || groups <= 0 || heads <= 0 || q_len <= 0 || kv_len <= 0) return A3D_EINVAL;
  if (!map_ok(qmap, head_dim) || !map_ok(kmap, head_dim) || !map_ok(omap, head_dim)) return A3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(V)) & 15u) return A3D_EINVAL;
  if (reinterpret_cast<uintptr_t>(O) & 7u) return A3D_EINVAL;
  if (groups > 65535) return A3D_EINVAL;
  AttnParams p{};
  p.Q = (const uint16_t*)Q; p.K = (const uint16_t*)K; p.V = (const uint16_t*)V; p.O = (uint16_t*)O;
  p.qm = *qmap; p.km = *kmap; p.om = *omap;
  p.heads = heads; p.q_len = q_len; p.kv_len = kv_len;
  p.scale_log2 = scale * 1.4426950408889634f; p.out_scale = out_scale; p.accumulate = accumulate;
  p.q_tiles = (int)((q_len + BQ - 1) / BQ);
  const int bkv = head_dim == 160 ? 32 : 64;
  p.kv_aligned = (kmap->seg_len % bkv == 0 || kv_len <= kmap->seg_len) ? 1 : 0;
  if (kv_len > 0x7fffffffLL || kmap->seg_len > 0x7fffffffLL) return A3D_EINVAL;
  const dim3 grid((unsigned)(heads * p.q_tiles), (unsigned)groups), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (head_dim) {
    case 40: flash_attn_kernel<40, 64><<<grid, block, 0, s>>>(p); break;
    case 80: flash_attn_kernel<80, 64><<<grid, block, 0, s>>>(p); break;
    case 160: flash_attn_kernel<160, 32><<<grid, block, 0, s>>>(p); break;
    default: return A3D_EUNSUPPORTED;
  }
  return a3d_launch_status();
}
