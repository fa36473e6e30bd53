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
// 16 bits, ARE the B operand of the gradient products (contraction over rows) without any cross-lane movement; their A operand,
// the row-side tensor with the row index as contraction, is read from the SAME natural image by ds_read_b64_tr_b16 (round 2 staged a
// second, transposed image per tensor: every row tile was loaded from global twice and transposed with v_perm_b32 + 8-byte LDS writes).
// The loads of the next row tile are issued into registers before the current tile's MFMAs and written to LDS after them; a wave owns
// one or two 32-column sub-tiles (CT) that share every row-side LDS fragment.
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
// the same map with the group part hoisted: the staging loops evaluate it for every row of every tile, and a 64-bit
// div / mod pair per row cost more than the tile's MFMAs in the first version (profiles/README.md, training section)
struct GroupRows { int64_t base; int64_t seg_stride; uint32_t seg_len; };
A3D_DEV GroupRows group_rows(const a3d_rowmap& m, int64_t g) {
  GroupRows r;
  r.base = (g / m.gdiv) * m.ga + (g % m.gdiv) * m.gb;
  r.seg_stride = m.seg_stride;
  r.seg_len = m.seg_len > 0x7fffffffLL ? 0x7fffffffu : (uint32_t)m.seg_len;     // sequence positions are < 2^31
  return r;
}
A3D_DEV int64_t row_of(const GroupRows& a, int s) {
  const uint32_t seg = (uint32_t)s / a.seg_len;
  return a.base + (int64_t)seg * a.seg_stride + (int64_t)((uint32_t)s - seg * a.seg_len);
}

// row (within a 32-row sub-tile) that feeds MFMA A-row i: result register r of a lane in half g then is row 16*(r>>3) + 8*g + (r&7)
A3D_DEV int kperm(int i) {
  const int j = i & 3, g = (i >> 2) & 1, b = i >> 3;
  return 16 * (b >> 1) + 8 * g + 4 * (b & 1) + j;
}

// LDS-DMA: 64 lanes x 16 B -> LDS[lds_dst + 16 lane] under a wave-uniform lane mask; source = scalar base + per-lane byte offset.  Not counted by
// the compiler: s_waitcnt vmcnt by hand (flash_common.h has the same helper for the forward kernels).
A3D_DEV void bwd_glds16_m(uint32_t voff, const void* sbase, uint32_t lds_dst, uint64_t mask) {
  unsigned keep;
  uint64_t ex;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, %5\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep), "=&s"(ex) : "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst)), "s"(mask) : "memory");
}
A3D_DEV const uint16_t* bwd_scalar(const uint16_t* ptr) {      // wave-uniform by construction; say so
  const uint64_t a = (uint64_t)(uintptr_t)ptr;
  return (const uint16_t*)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a));
}

// DMA (round 6): the row tiles go global -> LDS by LDS-DMA instead of through registers.  Timing ablations (profiles/r6_attn_bwd_ablations.log)
// put a quarter of a head_dim-40 backward into the row-tile hand-over — 2 x NPT global loads per thread issued before the tile's MFMAs,
// 2 x NPT ds_write_b128 behind them, their waits — and nothing into the barrier itself.  With DMA the next tile is requested at the START of an
// iteration into the other buffer (free since the barrier that ended the previous iteration) and has the whole tile's compute time to land;
// the iteration ends with s_waitcnt vmcnt(0) + the same one barrier.  Same LDS image (padded natural rows; pad / contraction-padding slots are
// never written: those lanes are masked off), same double buffer, so the occupancy is unchanged.  Launches whose row tiles are consecutive
// rows (ALIGNED) and whose row count is a multiple of the tile (no clamped rows) — every training shape of the UNet — take it.
template <int D, int MODE, int NU, int CT, bool ALIGNED, int OCC, bool DMA = false>
__global__ __launch_bounds__(256, OCC) void attn_bwd_kernel(const BwdParams p) {
  static_assert(!DMA || ALIGNED, "DMA staging needs consecutive rows per tile");
  constexpr int BR = 32 * NU;              // rows per LDS tile
  constexpr int DK = (D + 15) / 16 * 16;   // contraction length of the score products, zero padded
  constexpr int KS = DK / 16;
  constexpr int MT = (D + 31) / 32;        // 32-row tiles of the transposed gradients
  constexpr int NROW = DK + 8;             // natural image row stride (elements): odd number of 16-B slots
  constexpr int DCH = D / 8;               // 16-byte chunks per row
  constexpr int N_ELEMS = BR * NROW;
  constexpr int NCH = BR * DCH, NPT = (NCH + 255) / 256;            // natural staging: 16-byte chunks, per thread
  constexpr int WCOLS = 32 * CT;           // columns per wave
  static_assert((NROW / 8) % 2 == 1, "the LDS row stride must be an odd number of 16-B slots");

  // + 64 elements: the transposing reads of the last d block run up to 24 columns past a row's end (results never stored)
  // Two buffers of [natural image of tensor a | of tensor b]: tile t+1 is written while tile t is being read, one barrier per tile.
  __shared__ __attribute__((aligned(16))) uint16_t smem[4 * N_ELEMS + 64];
  __shared__ __attribute__((aligned(16))) float rstat_all[2][2][BR];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, g = lane >> 5;
  const int head = blockIdx.x % p.heads;
  const int ctile = blockIdx.x / p.heads;
  const int64_t hoff = (int64_t)head * D;
  const int64_t grp_c = (MODE == MODE_DKV) ? (int64_t)blockIdx.y * p.q_per_kv : (int64_t)blockIdx.y;   // group the column maps see
  const int clen = (MODE == MODE_DKV) ? p.kv_len : p.q_len;
  const int rlen = (MODE == MODE_DKV) ? p.q_len : p.kv_len;

  // zero the contraction padding of the natural images once (staging never writes it)
  if constexpr (DK > D) {
    for (int i = tid; i < 4 * BR * (DK - D); i += 256) {
      const int b = i / (BR * (DK - D)), rem = i % (BR * (DK - D));
      smem[b * N_ELEMS + (rem / (DK - D)) * NROW + D + rem % (DK - D)] = 0;
    }
  }

  // ---- column operands (B fragments): lane (column l31 of sub-tile c, half g) holds X[col][16*ks + 8*g .. +7]
  int col[CT]; bool col_ok[CT];
  u32x4_t cA[CT][KS], cB[CT][KS];
  float lse_c[CT], dl_c[CT];                                  // MODE_DQ: statistics of this lane's queries
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    col[c] = ctile * (4 * WCOLS) + wid * WCOLS + 32 * c + l31;
    col_ok[c] = col[c] < clen;
    const int colc = col_ok[c] ? col[c] : clen - 1;
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
        cA[c][ks] = *reinterpret_cast<const u32x4_t*>(a_src + d0);
        cB[c][ks] = *reinterpret_cast<const u32x4_t*>(b_src + d0);
      } else {
        cA[c][ks] = u32x4_t{0u, 0u, 0u, 0u};
        cB[c][ks] = u32x4_t{0u, 0u, 0u, 0u};
      }
    }
    lse_c[c] = 0.f; dl_c[c] = 0.f;
    if constexpr (MODE == MODE_DQ) {
      const int64_t si = ((int64_t)blockIdx.y * p.heads + head) * p.q_len + colc;
      lse_c[c] = p.lse2[si]; dl_c[c] = p.delta[si];
    }
  }

  f32x16_t acc1[CT][MT], acc2[MODE == MODE_DKV ? CT : 1][MT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc1[c][mt][r] = 0.f;
        if constexpr (MODE == MODE_DKV) acc2[c][mt][r] = 0.f;
      }
  float m_run[CT], l_run[CT], t_run[CT];                     // MODE_STATS
#pragma unroll
  for (int c = 0; c < CT; ++c) { m_run[c] = -INFINITY; l_run[c] = 0.f; t_run[c] = 0.f; }

  const int nrow_off = kperm(l31) * NROW + 8 * g;
  // transposing reads: lane (16-lane group q4 = lane >> 4, i16) addresses row 8 g + (i16 >> 2) (+ 4 for the second read), columns 16 (q4 & 1) + 4 (i16 & 3) .. + 3
  const int tr_off = (8 * g + ((lane & 15) >> 2)) * NROW + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  // ---- DMA lanes: 16-byte slot s of a tensor's tile image is row s / SPR, chunk s % SPR (chunks >= DCH: padding, never written);
  // instruction j covers slots 64 j .. 64 j + 63, wave w issues j = w, w + 4, ...
  constexpr int SPR = NROW / 8, NSLOT = BR * SPR, NI = (NSLOT + 63) / 64, NIW = (NI + 3) / 4;
  uint32_t voffA[DMA ? NIW : 1], voffB[DMA ? NIW : 1];
  uint64_t dmask[DMA ? NIW : 1];
  if constexpr (DMA) {
    const int64_t lda_ = (MODE == MODE_DKV) ? p.qm.ld : p.km.ld, ldb_ = (MODE == MODE_DKV) ? p.dom.ld : p.km.ld;
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
      const int slot = 64 * (wid + 4 * i) + lane;
      const int row = slot / SPR, ch = slot % SPR;
      const bool ok = slot < NSLOT && ch < DCH;
      voffA[i] = ok ? (uint32_t)((row * lda_ + ch * 8) * 2) : 0u;
      voffB[i] = ok ? (uint32_t)((row * ldb_ + ch * 8) * 2) : 0u;
      dmask[i] = __builtin_amdgcn_ballot_w64(ok);
    }
  }

  // ---- row-tile staging through registers: the loads of tile t+1 are issued before tile t's MFMAs, the LDS writes after them
  const uint16_t* const src_a = ((MODE == MODE_DKV) ? p.Q : p.K) + hoff;      // row-side tensors: (Q, dO) in DKV, (K, V) otherwise
  const uint16_t* const src_b = ((MODE == MODE_DKV) ? p.dO : p.V) + hoff;
  const int64_t ld_a = (MODE == MODE_DKV) ? p.qm.ld : p.km.ld;
  const int64_t ld_b = (MODE == MODE_DKV) ? p.dom.ld : p.km.ld;
  const int ntiles = (rlen + BR - 1) / BR;
  const int total_tiles = ntiles * ((MODE == MODE_DKV) ? p.q_per_kv : 1);
  u32x4_t n1[NPT], n2[NPT];
  float stat_r = 0.f;
  const float inv_do_scale = 1.f / p.do_scale;
  auto load_tile = [&](int t) __attribute__((always_inline)) {
    const int64_t grp_r = (MODE == MODE_DKV) ? grp_c + t / ntiles : grp_c;      // group the row maps see
    const int r0 = (t % ntiles) * BR;
    const GroupRows ra = (MODE == MODE_DKV) ? group_rows(p.qm, grp_r) : group_rows(p.km, grp_r);
    const GroupRows rb = (MODE == MODE_DKV) ? group_rows(p.dom, grp_r) : ra;
    // ALIGNED: a tile never straddles a map segment (seg_len % BR == 0), so its rows are consecutive: one division per tile
    const int64_t tile_a = ALIGNED ? row_of(ra, r0) : 0, tile_b = ALIGNED ? ((MODE == MODE_DKV) ? row_of(rb, r0) : tile_a) : 0;
    auto rows_of = [&](int s, int64_t& row_a, int64_t& row_b) __attribute__((always_inline)) {
      if (s >= rlen) s = rlen - 1;
      if constexpr (ALIGNED) {
        row_a = tile_a + (s - r0); row_b = tile_b + (s - r0);
      } else {
        row_a = row_of(ra, s); row_b = (MODE == MODE_DKV) ? row_of(rb, s) : row_a;
      }
    };
    if constexpr (DMA) {
      // (tile t goes into buffer t & 1: the caller guarantees its last readers have passed a barrier)
      const uint16_t* const ba = bwd_scalar(src_a + tile_a * ld_a);
      const uint16_t* const bb = bwd_scalar(src_b + tile_b * ld_b);
      const uint32_t l1 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(smem + (t & 1) * 2 * N_ELEMS);
      const uint32_t l2 = l1 + (uint32_t)N_ELEMS * 2u;
#pragma unroll
      for (int i = 0; i < NIW; ++i) {
        const int j = wid + 4 * i;
        if (j < NI) {
          bwd_glds16_m(voffA[i], ba, l1 + (uint32_t)j * 1024u, dmask[i]);
          bwd_glds16_m(voffB[i], bb, l2 + (uint32_t)j * 1024u, dmask[i]);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NPT; ++i) {
        const int c = tid + 256 * i;
        if (NCH % 256 == 0 || c < NCH) {
          const int r = c / DCH, ch = c % DCH;
          int64_t row_a, row_b;
          rows_of(r0 + r, row_a, row_b);
          n1[i] = *reinterpret_cast<const u32x4_t*>(src_a + row_a * ld_a + ch * 8);
          n2[i] = *reinterpret_cast<const u32x4_t*>(src_b + row_b * ld_b + ch * 8);
        }
      }
    }
    if constexpr (MODE == MODE_DKV) {
      if (tid < 2 * BR) {
        int s = r0 + (tid % BR); if (s >= rlen) s = rlen - 1;
        const int64_t si = (grp_r * p.heads + head) * (int64_t)p.q_len + s;
        // delta enters as dP's starting value: dP_raw - delta / do_scale (do_scale is applied once, to the finished gradients)
        stat_r = (tid < BR) ? p.lse2[si] : -p.delta[si] * inv_do_scale;
      }
    }
  };
  auto store_tile = [&](int buf) __attribute__((always_inline)) {
    uint16_t* const S1 = smem + buf * 2 * N_ELEMS;
    uint16_t* const S2 = S1 + N_ELEMS;
    if constexpr (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of the tile have landed (the barrier that follows publishes them)
    } else {
#pragma unroll
      for (int i = 0; i < NPT; ++i) {
        const int c = tid + 256 * i;
        if (NCH % 256 == 0 || c < NCH) {
          const int r = c / DCH, ch = c % DCH;
          *reinterpret_cast<u32x4_t*>(S1 + r * NROW + ch * 8) = n1[i];
          *reinterpret_cast<u32x4_t*>(S2 + r * NROW + ch * 8) = n2[i];
        }
      }
    }
    if constexpr (MODE == MODE_DKV) {
      if (tid < 2 * BR) rstat_all[buf][tid / BR][tid % BR] = stat_r;
    }
  };

  // Round 6.  (a) VALU diet of the gradient passes: dP's accumulator STARTS at -delta / do_scale (a lane constant in the dQ pass, one LDS
  // value per row in the dK|dV pass), so dS = P (dP_raw - delta / do_scale) is one multiply — do_scale is applied once, to the finished dQ / dK —,
  // and the select that zeroes P for rows beyond the sequence only exists in the instantiation of a ragged last tile: 2.5-3 plain VALU
  // instructions per score + the exp, where there were 5.5-6.  (b) Software pipeline inside a row tile: the two score-shaped products of
  // sub-tile u + 1 are issued before the exp / dS arithmetic of sub-tile u, whose gradient products in turn run under the arithmetic of u + 1.
  // (the second score set is 32 registers: the pipeline exists where it fits the 256-register budget of two workgroups per CU without
  // spilling — the dQ pass up to head_dim 64; the dK|dV pass (235 registers before) and head_dim 80 keep the lean arithmetic only)
  constexpr bool PIPE = (MODE == MODE_DQ) && D <= 64 && NU > 1;
  f32x16_t sS[PIPE ? 2 : 1][CT], sP[PIPE ? 2 : 1][CT];
  auto score_products = [&](const uint16_t* N1, const uint16_t* N2, const float (*rstat)[BR], int u, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      if constexpr (MODE == MODE_DQ) {
        const float d0 = -dl_c[c] * inv_do_scale;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sS[slot][c][r] = 0.f; sP[slot][c][r] = d0; }
      } else if constexpr (MODE == MODE_DKV) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float4 da = *reinterpret_cast<const float4*>(&rstat[1][32 * u + 16 * h + 8 * g]);
          const float4 db = *reinterpret_cast<const float4*>(&rstat[1][32 * u + 16 * h + 8 * g + 4]);
          sP[slot][c][8 * h] = da.x; sP[slot][c][8 * h + 1] = da.y; sP[slot][c][8 * h + 2] = da.z; sP[slot][c][8 * h + 3] = da.w;
          sP[slot][c][8 * h + 4] = db.x; sP[slot][c][8 * h + 5] = db.y; sP[slot][c][8 * h + 6] = db.z; sP[slot][c][8 * h + 7] = db.w;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sS[slot][c][r] = 0.f;
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) { sS[slot][c][r] = 0.f; sP[slot][c][r] = 0.f; }
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4_t a1 = *reinterpret_cast<const u32x4_t*>(N1 + 32 * u * NROW + nrow_off + 16 * ks);
      const u32x4_t a2 = *reinterpret_cast<const u32x4_t*>(N2 + 32 * u * NROW + nrow_off + 16 * ks);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        sS[slot][c] = mfma32(a1, cA[c][ks], sS[slot][c]);
        sP[slot][c] = mfma32(a2, cB[c][ks], sP[slot][c]);
      }
    }
  };
  // exp / dS arithmetic and the gradient products of sub-tile u (scores in slot); RAGGED: rows beyond rlen exist in this tile
  auto gradient_step = [&](const uint16_t* N1, const uint16_t* N2, const float (*rstat)[BR], int r0, int u, int slot, auto ragged_c) __attribute__((always_inline)) {
    constexpr bool RAGGED = decltype(ragged_c)::value;
    const int rbase = r0 + 32 * u + 8 * g;          // row of register r: rbase + 16*(r>>3) + (r&7)
    float ls[16];                                   // MODE_DKV: log-sum-exp of the 16 rows this lane holds
    if constexpr (MODE == MODE_DKV) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 la = *reinterpret_cast<const float4*>(&rstat[0][32 * u + 16 * h + 8 * g]);
        const float4 lb = *reinterpret_cast<const float4*>(&rstat[0][32 * u + 16 * h + 8 * g + 4]);
        ls[8 * h] = la.x; ls[8 * h + 1] = la.y; ls[8 * h + 2] = la.z; ls[8 * h + 3] = la.w;
        ls[8 * h + 4] = lb.x; ls[8 * h + 5] = lb.y; ls[8 * h + 6] = lb.z; ls[8 * h + 7] = lb.w;
      }
    }
    u32x4_t dsf[CT][2], pf[CT][2];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      float pv[16], ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float lse = (MODE == MODE_DQ) ? lse_c[c] : ls[r];
        pv[r] = __builtin_amdgcn_exp2f(fmaf(sS[slot][c][r], p.scale_log2, -lse));
        if constexpr (RAGGED) { if (!(rbase + 16 * (r >> 3) + (r & 7) < rlen)) pv[r] = 0.f; }
        ds[r] = pv[r] * sP[slot][c][r];
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dsf[c][h][j] = pack16(ds[8 * h + 2 * j], ds[8 * h + 2 * j + 1]);
          pf[c][h][j] = pack16(pv[8 * h + 2 * j], pv[8 * h + 2 * j + 1]);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // A = X^T [d = 32 mt + l31][rows 32 u + 16 h + 8 g .. + 7] straight from the natural image: two transposing reads of 4 rows each
        const uint16_t* const tsrc = N1 + (32 * u + 16 * h) * NROW + 32 * mt + tr_off;
        const u32x2_t f1a = lds_tr16_b64(tsrc), f1b = lds_tr16_b64(tsrc + 4 * NROW);
        const u32x4_t f1 = {f1a[0], f1a[1], f1b[0], f1b[1]};
#pragma unroll
        for (int c = 0; c < CT; ++c) acc1[c][mt] = mfma32(f1, dsf[c][h], acc1[c][mt]);        // dQ^T += K^T dS^T   /   dK^T += Q^T dS
        if constexpr (MODE == MODE_DKV) {
          const uint16_t* const tsrc2 = N2 + (32 * u + 16 * h) * NROW + 32 * mt + tr_off;
          const u32x2_t f2a = lds_tr16_b64(tsrc2), f2b = lds_tr16_b64(tsrc2 + 4 * NROW);
          const u32x4_t f2 = {f2a[0], f2a[1], f2b[0], f2b[1]};
#pragma unroll
          for (int c = 0; c < CT; ++c) acc2[c][mt] = mfma32(f2, pf[c][h], acc2[c][mt]);       // dV^T += dO^T P
        }
      }
  };

  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < total_tiles; ++t) {
    const int r0 = (t % ntiles) * BR;
    const uint16_t* const N1 = smem + (t & 1) * 2 * N_ELEMS;
    const uint16_t* const N2 = N1 + N_ELEMS;
    const float (*const rstat)[BR] = rstat_all[t & 1];
#if !defined(A3D_EXP_BWD_NOSTAGE)
    if (t + 1 < total_tiles) load_tile(t + 1);
#endif
    if constexpr (MODE == MODE_STATS) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        score_products(N1, N2, rstat, u, 0);
        const int rbase = r0 + 32 * u + 8 * g;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          float sv[16];
          float mx = -INFINITY;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            sv[r] = (rbase + 16 * (r >> 3) + (r & 7) < rlen) ? sS[0][c][r] * p.scale_log2 : -INFINITY;
            mx = fmaxf(mx, sv[r]);
          }
          mx = fmaxf(mx, __shfl_xor(mx, 32));
          const float m_new = fmaxf(m_run[c], mx);            // finite: sub-tile 0 of every tile holds a valid row
          const float alpha = __builtin_amdgcn_exp2f(m_run[c] - m_new);
          l_run[c] *= alpha; t_run[c] *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(sv[r] - m_new);
            l_run[c] += e;
            t_run[c] = fmaf(e, sP[0][c][r], t_run[c]);
          }
          m_run[c] = m_new;
        }
      }
    } else {
      const bool ragged = r0 + BR > rlen;                     // workgroup-uniform
      if constexpr (PIPE) score_products(N1, N2, rstat, 0, 0);
      static_for<NU>([&](auto u_c) __attribute__((always_inline)) {
        constexpr int u = decltype(u_c)::value;
        constexpr int slot = PIPE ? (u & 1) : 0;
        if constexpr (PIPE) { if constexpr (u + 1 < NU) score_products(N1, N2, rstat, u + 1, (u + 1) & 1); }
        else score_products(N1, N2, rstat, u, 0);
        if constexpr (D >= 160) {               // (512 registers in use there: one body, with the select)
          gradient_step(N1, N2, rstat, r0, u, slot, std::true_type{});
        } else {
          if (ragged) gradient_step(N1, N2, rstat, r0, u, slot, std::true_type{});
          else gradient_step(N1, N2, rstat, r0, u, slot, std::false_type{});
        }
      });
    }
#ifdef A3D_EXP_BWD_SINGLEBUF      // A/B build (python -m animate3d_amd.build --experiment A3D_EXP_BWD_SINGLEBUF): round 2's flow, two barriers per tile
    __syncthreads();
    if (t + 1 < total_tiles) store_tile((t + 1) & 1);
    __syncthreads();
#elif defined(A3D_EXP_BWD_NOSTAGE)      // timing ablations (results wrong): no row-tile loads / LDS stores after tile 0; == 2: no barrier either
    if (t == 0) store_tile(1);
#if A3D_EXP_BWD_NOSTAGE != 2
    __syncthreads();
#endif
#else
    // tile t+1 goes into the other buffer: its last readers (tile t-1) passed the barrier of the previous iteration
    if (t + 1 < total_tiles) store_tile((t + 1) & 1);
    __syncthreads();
#endif
  }

  // ---- results: lane holds column l31 of each sub-tile; register r = 4*qd + j of tile mt is dim d = 32*mt + 8*qd + 4*g + j
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    if constexpr (MODE == MODE_STATS) {
      const float l_tot = l_run[c] + __shfl_xor(l_run[c], 32);
      const float t_tot = t_run[c] + __shfl_xor(t_run[c], 32);
      if (g == 0 && col_ok[c]) {
        const int64_t si = ((int64_t)blockIdx.y * p.heads + head) * p.q_len + col[c];
        p.lse2[si] = m_run[c] + __builtin_amdgcn_logf(l_tot);       // v_log_f32 = log2
        p.delta[si] = p.do_scale * t_tot / l_tot;
      }
    } else {
      if (!col_ok[c]) continue;
      auto write = [&](uint16_t* base, const a3d_rowmap& m, const f32x16_t (&acc)[MT], float mul) __attribute__((always_inline)) {
        uint16_t* dst = base + map_row(m, grp_c, col[c]) * m.ld + hoff;
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
        write(p.dQ, p.dqm, acc1[c], p.scale * p.do_scale);
      } else {
        write(p.dK, p.dkm, acc1[c], p.scale * p.do_scale);
        write(p.dV, p.dkm, acc2[c], p.do_scale);
      }
    }
  }
}

template <int D, int MODE, int NU, int CT>
int launch(hipStream_t s, const BwdParams& p, int groups_y) {
  const int clen = (MODE == MODE_DKV) ? p.kv_len : p.q_len;
  const int64_t ctiles = (clen + 128 * CT - 1) / (128 * CT);
  if (ctiles * p.heads > 0x7fffffffLL || groups_y > 65535) return A3D_EINVAL;
  const dim3 grid((unsigned)(ctiles * p.heads), (unsigned)groups_y);
  // row side: (Q, dO) maps in DKV, the K/V map otherwise
  constexpr int BR = 32 * NU;
  const bool aligned = (MODE == MODE_DKV) ? (p.qm.seg_len % BR == 0 && p.dom.seg_len % BR == 0) : (p.km.seg_len % BR == 0);
  // workgroups per CU the register budget is set for: two wherever the kernel fits 256 registers without spilling
  // (measured, round 3: three workgroups per CU for the 168-register dQ pass and 128-row tiles at head_dim 40 change nothing: 2.33-2.44 ms
  // per level-0 backward in every variant, profiles/README.md)
  constexpr int OCC = (CT == 1 && (D <= 64 || (D == 80 && MODE != MODE_DKV))) ? 2 : 1;
  const int rlen = (MODE == MODE_DKV) ? p.q_len : p.kv_len;
  // LDS-DMA staging: consecutive rows per tile, no clamped rows, 32-bit lane offsets, 16-byte aligned rows (checked by the entry point)
  const int64_t ld_max = p.qm.ld > p.km.ld ? (p.qm.ld > p.dom.ld ? p.qm.ld : p.dom.ld) : (p.km.ld > p.dom.ld ? p.km.ld : p.dom.ld);
#ifdef A3D_EXP_BWD_NODMA
  const bool dma = false;
#else
  const bool dma = MODE != MODE_STATS && aligned && rlen % BR == 0 && (int64_t)BR * ld_max * 2 < (1ll << 31);
#endif
  if (dma) attn_bwd_kernel<D, MODE, NU, CT, true, OCC, true><<<grid, dim3(256), 0, s>>>(p);
  else if (aligned) attn_bwd_kernel<D, MODE, NU, CT, true, OCC><<<grid, dim3(256), 0, s>>>(p);
  else attn_bwd_kernel<D, MODE, NU, CT, false, OCC><<<grid, dim3(256), 0, s>>>(p);
  return a3d_launch_status();
}

// One 32-column sub-tile per wave and two workgroups per CU: measured faster than two sub-tiles at one workgroup per CU (whose single
// wave per SIMD has nothing to overlap its LDS / exp / staging latencies with); the CT = 2 instantiation stays available for tuning.
template <int MODE>
int dispatch(hipStream_t s, const BwdParams& p, int head_dim, int groups_y) {
  switch (head_dim) {
#ifdef A3D_EXP_BWD_NU2
    case 40: return launch<40, MODE, 2, 1>(s, p, groups_y);      // measurement build: round 5's 64-row tiles
#else
    // round 6: 128-row tiles at head_dim 40 (57 KB of LDS, still two workgroups per CU): half the barriers per score, 2.28 -> 2.12 ms per
    // level-0 backward (profiles/r6_microbench_attn_bwd.log); head_dim 80 would drop to one workgroup per CU (90 KB) and stays at 64 rows
    case 40: return launch<40, MODE, 4, 1>(s, p, groups_y);
#endif
    case 64: return launch<64, MODE, 2, 1>(s, p, groups_y);
    case 80: return launch<80, MODE, 2, 1>(s, p, groups_y);
    case 160: return launch<160, MODE, 1, 1>(s, p, groups_y);
    default: return A3D_EUNSUPPORTED;
  }
}

// delta[g][h][q] = sum_d dO[row(q)][h D + d] * O[row(q)][h D + d]: the row sums  sum_k P_qk dP_qk  of the statistics pass taken from the
// forward's output instead (dO V^T P^T = dO · (P V) = dO · O / out_scale, times do_scale = out_scale), for callers that kept O and the
// forward's log-sum-exp (a3d_flash_attn_lse).  One thread per (query, head); the heads of a row are adjacent threads.
template <int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(const uint16_t* __restrict__ dO, const uint16_t* __restrict__ O, const a3d_rowmap dom,
                                                         const a3d_rowmap om, float* __restrict__ delta, int heads, int q_len) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t grp = blockIdx.y;
  if (t >= (int64_t)q_len * heads) return;
  const int q = (int)(t / heads), h = (int)(t % heads);
  const u32x4_t* a = reinterpret_cast<const u32x4_t*>(dO + map_row(dom, grp, q) * dom.ld + (int64_t)h * D);
  const u32x4_t* b = reinterpret_cast<const u32x4_t*>(O + map_row(om, grp, q) * om.ld + (int64_t)h * D);
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < D / 8; ++i) {
    const u32x4_t x = a[i], y = b[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = fmaf(lo16(x[e]), lo16(y[e]), fmaf(hi16(x[e]), hi16(y[e]), acc));
  }
  delta[(grp * heads + h) * q_len + q] = acc;
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
  // accumulate: bit 0 = add to dQ / dK / dV; bit 1 = lse2 and delta hold the statistics already (a3d_flash_attn_lse + a3d_attn_delta):
  // the statistics pass (a third of the backward's time at head_dim 40) is skipped
  p.scale = scale; p.scale_log2 = scale * 1.4426950408889634f; p.do_scale = do_scale; p.accumulate = accumulate & 1;
  hipStream_t s = (hipStream_t)stream;
  if (!((accumulate >> 1) & 1)) {
    if (int rc = dispatch<MODE_STATS>(s, p, head_dim, groups)) return rc;
  }
  if (dQ) { if (int rc = dispatch<MODE_DQ>(s, p, head_dim, groups)) return rc; }
  if (dK) { if (int rc = dispatch<MODE_DKV>(s, p, head_dim, groups / q_per_kv)) return rc; }
  return A3D_OK;
}

extern "C" int A3D_FN(a3d_attn_delta)(a3d_stream_t stream, const void* dO, const void* O, const a3d_rowmap* domap, const a3d_rowmap* omap,
                                       float* delta, int groups, int heads, int head_dim, int64_t q_len) {
  if (!dO || !O || !delta || groups <= 0 || groups > 65535 || heads <= 0 || q_len <= 0 || q_len > 0x7fffffffLL) return A3D_EINVAL;
  if (!map_ok(domap, head_dim, heads) || !map_ok(omap, head_dim, heads)) return A3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(dO) | reinterpret_cast<uintptr_t>(O)) & 15u) return A3D_EINVAL;
  const int64_t n = q_len * heads;
  const dim3 grid((unsigned)((n + 255) / 256), (unsigned)groups);
  hipStream_t s = (hipStream_t)stream;
  switch (head_dim) {
    case 40: attn_delta_kernel<40><<<grid, dim3(256), 0, s>>>((const uint16_t*)dO, (const uint16_t*)O, *domap, *omap, delta, heads, (int)q_len); break;
    case 64: attn_delta_kernel<64><<<grid, dim3(256), 0, s>>>((const uint16_t*)dO, (const uint16_t*)O, *domap, *omap, delta, heads, (int)q_len); break;
    case 80: attn_delta_kernel<80><<<grid, dim3(256), 0, s>>>((const uint16_t*)dO, (const uint16_t*)O, *domap, *omap, delta, heads, (int)q_len); break;
    case 160: attn_delta_kernel<160><<<grid, dim3(256), 0, s>>>((const uint16_t*)dO, (const uint16_t*)O, *domap, *omap, delta, heads, (int)q_len); break;
    default: return A3D_EUNSUPPORTED;
  }
  return a3d_launch_status();
}
