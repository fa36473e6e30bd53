// Flash attention forward for gfx950: O = softmax(Q K^T * scale) V per (group, head), bf16 in/out,
// fp32 scores/accumulators, no score matrix in memory.  Head dims 40 / 80 / 160 (SD1.5 levels).
//
// Row addressing goes through a3d_rowmap, so the reference's "(b n f) l c -> (b f) (n l) c"
// regrouping, the first-frame K/V selection of the I2V branch and the per-video text/IP tokens
// are all just different maps over the same [rows, C] tensors (no rearrange copies).
//
// Structure (one 256-thread workgroup = 4 waves; each wave owns QT x 32 query rows; KV tiles of BKV keys):
//   * S^T = K · Q^T with v_mfma_f32_32x32x16_bf16: A = K rows from LDS (16-B reads, padded rows), B = Q^T held
//     in registers for the whole kernel; one K fragment feeds the MFMAs of all QT query sub-tiles.  The result
//     layout gives every lane ONE query per sub-tile (lane&31) and 16 keys per 32-key sub-tile.
//   * The K row that feeds MFMA row i is permuted (kperm) so that the 8 scores a lane holds in registers
//     8j..8j+7 are 8 CONSECUTIVE keys: P^T then is directly the B operand of O^T = V^T · P^T (no cross-lane
//     movement), and the matching A operand is one 16-B read of a V^T image in LDS (V is transposed while it is
//     staged: 4 keys x 8 dims per thread, v_perm_b32 + 8-byte LDS writes).
//   * The measured limiter of the first versions was VALU issue (~13 VALU per MFMA), so the softmax is stripped
//     to one v_exp + half a v_cvt_pk + half a v_max3 per score:
//       - Q is pre-multiplied by scale*log2(e) when its fragments are loaded (once per kernel);
//       - the running-max offset is subtracted INSIDE the matrix pipe: for D = 40 the contraction is padded to
//         48 anyway, so K gets a constant-1 column and Q carries -m in that slot (OFS_PAD); for D = 80 the
//         offset enters as the MFMA's C operand (OFS_ACC).  P = exp2(S') needs no fma;
//       - the max is lazy: the offset only moves (exchange with lane^32, rescale O, re-base S') when some score
//         exceeds it by more than 2^6 (LAZY_THR) — one wave vote per tile on the common path;
//       - row sums come out of the matrix pipe: a spare V^T row holds ones, so O^T[row D] = sum_k P.
//   * O^T accumulators keep the query in lane&31 too, so the (rare) rescale is a plain per-lane multiply.
//   * K/V images are double-buffered in LDS: tile t+1 is written (from registers filled during the previous
//     iteration) and tile t+2 is requested from HBM/L2 before tile t's MFMAs start; one barrier per tile.
//     Per-thread source pointers advance incrementally; only the last tile carries clamp + mask code.
//   * grid.x = heads * q_tiles with the head fastest: with 8 heads block b runs on XCD b%8 = head, so all
//     q-tiles of one (group, head) share one XCD's L2 copy of that K/V.
//   ALIGNED = a KV tile never straddles a row-map segment (seg_len % BKV == 0, or one segment); the generic
//   variant (small low-resolution levels only) recomputes rows with 32-bit div/mod.
#include <type_traits>
#include <utility>

#include "common.h"
#include "flash_common.h"


namespace {

// VAR (tuning experiments, a3d_tune_flash): 1 = s_setprio(1) around MFMA groups, 2 = V fragments read before the
// exps of their sub-tile, 4 = sched_group_barrier pattern {1 MFMA, 4 TRANS, 2 VALU} over the exp/PV section.
// NM = 1 (bf16 storage, OFS_PAD / OFS_ACC, aligned launches without a tail, >= 3 tiles): first a max-free pass — the offset is the exact
// maximum of the first tile + 40 and never moves (bf16 P has fp32's exponent range; fp32 accumulation), so the per-score VALU work is
// one v_exp and half a v_cvt_pk; the row sums out of the matrix pipe are checked once at the end and a workgroup whose sums
// left [0, 2^100) re-runs with the exact lazy running maximum (flash_attn_dm.hip has the long form of the argument).
// TWO: a second key set (p.K2 / V2 / km2 / kv_len2 / out_scale2) is attended to after the first, with its own softmax; the first
// result waits in registers and one sum is stored (IPAdapter processor: text tokens + image tokens, attention_processor.py:233, 268-283).
template <int D, int BKV, int QT, int OFS, bool ALIGNED, int VAR, int NM = 0, bool TWO = false>
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
  constexpr int BQW = 32 * QT, BQ = 4 * BQW;
  constexpr int KS_PAD = D / 16, G_PAD = (D % 16) / 8;    // fragment slot of contraction index D (OFS_PAD)
  static_assert((KROW / 8) % 2 == 1 && (VROW / 8) % 2 == 1, "LDS row strides must be an odd number of 16-B slots");
  static_assert(OFS != OFS_PAD || (DK > D && D % 8 == 0), "OFS_PAD needs a spare contraction slot");

  __shared__ __attribute__((aligned(16))) uint16_t smem[2 * (KS_ELEMS + VT_ELEMS)];
  uint16_t* const Ks0 = smem;
  uint16_t* const Vt0 = smem + 2 * KS_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, g = lane >> 5;
  // head fastest: with 8 heads block b runs on XCD b % 8 = head and all query tiles of a (group, head) share one L2 copy of its K/V
  // (a query-tile-major order for the short-K/V cross attention measured no difference: that launch is bound by its per-workgroup
  // prologue, not by the Q / O lines shared between XCDs; profiles/README.md, round 4)
  const int head = blockIdx.x % p.heads;
  const int qt = blockIdx.x / p.heads;
  const int64_t grp = blockIdx.y;
  const int64_t hoff = (int64_t)head * D;

  // one-time LDS init: contraction padding of both K images (OFS_PAD: column D = 1.0), ones row of both V^T images
  if constexpr (DK > D) {
    for (int i = tid; i < 2 * BKV * (DK - D); i += 256) {
      const int b = i / (BKV * (DK - D)), rem = i % (BKV * (DK - D));
      const int c = rem % (DK - D);
      Ks0[b * KS_ELEMS + (rem / (DK - D)) * KROW + D + c] = (OFS == OFS_PAD && c == 0) ? ONE16 : 0;
    }
  }
  if constexpr (ONES) {
    for (int i = tid; i < 2 * BKV; i += 256) Vt0[(i / BKV) * VT_ELEMS + D * VROW + (i % BKV)] = ONE16;   // bf16 1.0
  }

  // ---- Q^T fragments: lane (q = l31, half g) holds Q[q][16*ks + 8*g .. +7] for each of its QT queries
  int q_idx[QT];
  bool q_ok[QT];
  u32x4_t qf[QT][KS];
#pragma unroll
  for (int qs = 0; qs < QT; ++qs) {
    q_idx[qs] = qt * BQ + wid * BQW + qs * 32 + l31;
    q_ok[qs] = q_idx[qs] < p.q_len;
    const int64_t q_row = map_row(p.qm, grp, q_ok[qs] ? q_idx[qs] : p.q_len - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = 16 * ks + 8 * g;
      if (d0 < D) {
        u32x4_t w = *reinterpret_cast<const u32x4_t*>(p.Q + q_row * p.qm.ld + hoff + d0);
        if constexpr (OFS != OFS_FMA) {      // fold scale * log2(e) into Q once
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = pack16(lo16(w[j]) * p.scale_log2, hi16(w[j]) * p.scale_log2);
        }
        qf[qs][ks] = w;
      } else {
        qf[qs][ks] = u32x4_t{0u, 0u, 0u, 0u};
      }
    }
  }

#ifdef A3D_STORAGE_F16
  constexpr bool TRY_NOMAX = false;
#else
  constexpr bool TRY_NOMAX = NM != 0;
#endif
  static_assert(NM == 0 || (OFS != OFS_FMA && ALIGNED && ONES), "the max-free pass needs the offset inside the matrix pipe and MFMA row sums");
  // One complete pass over the keys; returns false when the max-free result must be discarded.
  f32x16_t osave[TWO ? QT : 1][TWO ? MT : 1];      // normalised result of the first key set
  auto pass = [&](auto nm_c, auto set_c) __attribute__((always_inline)) -> bool {
  constexpr bool NOMAX = decltype(nm_c)::value;
  constexpr bool SET1 = decltype(set_c)::value;       // this pass reads the second key set
  const uint16_t* const Kp = SET1 ? p.K2 : p.K;
  const uint16_t* const Vp = SET1 ? p.V2 : p.V;
  const a3d_rowmap& kmp = SET1 ? p.km2 : p.km;
  const int kvl = SET1 ? p.kv_len2 : p.kv_len;
  const float oscale = SET1 ? p.out_scale2 : p.out_scale;
  if constexpr (OFS == OFS_PAD) {
#pragma unroll
    for (int qs = 0; qs < QT; ++qs)
      if (g == G_PAD) qf[qs][KS_PAD][0] = 0u;
  }
  // ---- K/V staging: per-thread source pointers, advanced tile by tile
  const int64_t ld = kmp.ld;
  const int64_t kgbase = map_group_base(kmp, grp);
  const uint16_t* const Kh = Kp + hoff;
  const uint16_t* const Vh = Vp + hoff;
  const uint32_t seg_len = (uint32_t)kmp.seg_len;
  const int64_t tile_step = (int64_t)BKV * ld;                                  // elements per tile
  const int64_t wrap_step = (kmp.seg_stride - kmp.seg_len) * ld;             // extra jump at a segment end
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
    if (s >= kvl) s = kvl - 1;
    const uint32_t seg = (uint32_t)s / seg_len;
    return kgbase + (int64_t)seg * kmp.seg_stride + ((uint32_t)s - seg * seg_len);
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
        for (int j = 0; j < 4; ++j) {   // word j of each key row holds dims 2j (lo half) and 2j+1 (hi half)
          const uint32_t w0 = vreg[i][0][j], w1 = vreg[i][1][j], w2 = vreg[i][2][j], w3 = vreg[i][3][j];
          u32x2_t even, odd;            // v_perm_b32: selector bytes 0-3 pick from the 2nd operand, 4-7 from the 1st
          even[0] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);
          even[1] = __builtin_amdgcn_perm(w3, w2, 0x05040100u);
          odd[0] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);
          odd[1] = __builtin_amdgcn_perm(w3, w2, 0x07060302u);
          *reinterpret_cast<u32x2_t*>(Vt + (vc[i] * 8 + 2 * j) * VROW + vq[i] * 4) = even;
          *reinterpret_cast<u32x2_t*>(Vt + (vc[i] * 8 + 2 * j + 1) * VROW + vq[i] * 4) = odd;
        }
      }
  };

  f32x16_t oacc[QT][MT];
#pragma unroll
  for (int qs = 0; qs < QT; ++qs)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qs][mt][r] = 0.f;
  float m_off[QT];          // OFS_PAD/ACC: offset currently subtracted from the (scaled) scores; OFS_FMA: running max
  float l_run = 0.f;        // only when !ONES
  f32x16_t minit[OFS == OFS_ACC ? QT : 1];
#pragma unroll
  for (int qs = 0; qs < QT; ++qs) m_off[qs] = (OFS == OFS_FMA) ? -INFINITY : 0.f;
  if constexpr (OFS == OFS_ACC) {
#pragma unroll
    for (int qs = 0; qs < QT; ++qs)
#pragma unroll
      for (int r = 0; r < 16; ++r) minit[qs][r] = 0.f;
  }
  bool first = true;        // wave-uniform: no offset chosen yet
  const int krow_off = kperm(l31) * KROW + 8 * g;
  const int vrow_off = l31 * VROW + 8 * g;

  // first_c: 2 = the wave-uniform `first` flag decides (exact pass), 1 = this is the first tile, 0 = it is not (max-free pass)
  auto compute = [&](int buf, int kv0, auto tail_c, auto first_c) {
    constexpr bool TAIL = decltype(tail_c)::value;
    constexpr int FM = decltype(first_c)::value;
    const uint16_t* const Ks = Ks0 + buf * KS_ELEMS + krow_off;
    const uint16_t* const Vt = Vt0 + buf * VT_ELEMS + vrow_off;
    // ---- S'^T = K · Q'^T (- offset) for the NU 32-key sub-tiles; one K fragment feeds all QT query sub-tiles
    f32x16_t sacc[QT][NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
      for (int qs = 0; qs < QT; ++qs) {
        if constexpr (OFS == OFS_ACC) sacc[qs][u] = minit[qs];
        else {
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[qs][u][r] = 0.f;
        }
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u32x4_t kf = *reinterpret_cast<const u32x4_t*>(Ks + 32 * u * KROW + 16 * ks);
        if constexpr ((VAR & 1) != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int qs = 0; qs < QT; ++qs) sacc[qs][u] = mfma32(kf, qf[qs][ks], sacc[qs][u]);
        if constexpr ((VAR & 1) != 0) __builtin_amdgcn_s_setprio(0);
      }
    }
    if constexpr (TAIL) {      // keys past kv_len
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kv0 + 32 * u + 16 * (r >> 3) + 8 * g + (r & 7) >= kvl) {
#pragma unroll
            for (int qs = 0; qs < QT; ++qs) sacc[qs][u][r] = -INFINITY;
          }
    }
    if (p.causal) {            // wave-uniform flag: keys after the query's own position
#pragma unroll
      for (int qs = 0; qs < QT; ++qs)
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kv0 + 32 * u + 16 * (r >> 3) + 8 * g + (r & 7) > q_idx[qs]) sacc[qs][u][r] = -INFINITY;
    }

    if constexpr (OFS == OFS_FMA && QT > 1) {
      // ---- D = 80 with two query sub-tiles per wave: scores are raw; running max in raw units, lazy by LAZY_THR / scale_log2
#pragma unroll
      for (int qs = 0; qs < QT; ++qs) {
        float mx = sacc[qs][0][0];
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[qs][u][r]);
        if (__any(mx > m_off[qs] + LAZY_THR / p.scale_log2)) {
          mx = fmaxf(mx, __shfl_xor(mx, 32));
          const float m_new = fmaxf(m_off[qs], mx);
          const float alpha = __builtin_amdgcn_exp2f((m_off[qs] - m_new) * p.scale_log2);
          m_off[qs] = m_new;
          if constexpr (!ONES) l_run *= alpha;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qs][mt][r] *= alpha;
        }
        const float mneg = -m_off[qs] * p.scale_log2;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[qs][u][r] = fmaf(sacc[qs][u][r], p.scale_log2, mneg);
      }
    } else if constexpr (OFS == OFS_FMA) {
      // ---- D = 160: scores are raw; running max in raw units, lazy by LAZY_THR / scale_log2
      float mx = sacc[0][0][0];
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[0][u][r]);
      if (__any(mx > m_off[0] + LAZY_THR / p.scale_log2)) {
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_off[0], mx);
        const float alpha = __builtin_amdgcn_exp2f((m_off[0] - m_new) * p.scale_log2);
        m_off[0] = m_new;
        l_run *= alpha;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[0][mt][r] *= alpha;
      }
      const float mneg = -m_off[0] * p.scale_log2;
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[0][u][r] = fmaf(sacc[0][u][r], p.scale_log2, mneg);
    } else {
      // ---- lazy offset update: the common path is max + compare + one wave vote per query sub-tile; the slow
      //      path re-bases S', rescales O and moves the offset.  Max-free pass: only the first tile sets an offset.
      if constexpr (FM != 0) {
      const bool isfirst = (FM == 1) ? true : first;
#pragma unroll
      for (int qs = 0; qs < QT; ++qs) {
        float mx = sacc[qs][0][0];
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[qs][u][r]);
        if (__any(isfirst || mx > LAZY_THR)) {
          const float mxp = fmaxf(mx, __shfl_xor(mx, 32));
          float delta = isfirst ? mxp + (NOMAX ? 40.f : 0.f) : fmaxf(mxp, 0.f);
          float new_off = m_off[qs] + delta;
          if constexpr (OFS == OFS_PAD) { new_off = round16(new_off); delta = new_off - m_off[qs]; }
          m_off[qs] = new_off;
#pragma unroll
          for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[qs][u][r] -= delta;
          if (!isfirst) {
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            if constexpr (!ONES) l_run *= alpha;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int r = 0; r < 16; ++r) oacc[qs][mt][r] *= alpha;
          }
          if constexpr (OFS == OFS_PAD) {
            if (g == G_PAD) qf[qs][KS_PAD][0] = pack16(-new_off, 0.f);     // contraction slot D carries -offset
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) minit[qs][r] = -new_off;
          }
        }
      }
      }
      first = false;
    }

    // ---- P = exp2(S'), O^T += V^T · P^T   (row D of V^T is all ones: O^T[D] accumulates the row sums)
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      u32x4_t vfr[(VAR & 2) ? MT : 1][2];
      if constexpr ((VAR & 2) != 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int h = 0; h < 2; ++h) vfr[mt][h] = *reinterpret_cast<const u32x4_t*>(Vt + 32 * mt * VROW + 32 * u + 16 * h);
      }
      u32x4_t pf[QT][2];
#pragma unroll
      for (int qs = 0; qs < QT; ++qs) {
        float pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pv[r] = __builtin_amdgcn_exp2f(sacc[qs][u][r]);
          if constexpr (!ONES) l_run += pv[r];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < 4; ++j) pf[qs][h][j] = pack16(pv[8 * h + 2 * j], pv[8 * h + 2 * j + 1]);
      }
      if constexpr ((VAR & 1) != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          u32x4_t vf;
          if constexpr ((VAR & 2) != 0) vf = vfr[mt][h];
          else vf = *reinterpret_cast<const u32x4_t*>(Vt + 32 * mt * VROW + 32 * u + 16 * h);
#pragma unroll
          for (int qs = 0; qs < QT; ++qs) oacc[qs][mt] = mfma32(vf, pf[qs][h], oacc[qs][mt]);
        }
      if constexpr ((VAR & 1) != 0) __builtin_amdgcn_s_setprio(0);
    }
    if constexpr ((VAR & 4) != 0) {
#pragma unroll
      for (int i = 0; i < NU * MT * 2 * QT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x400, 4, 0);   // 4 TRANS (v_exp)
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // 2 VALU (v_cvt_pk)
      }
    }
  };

  const int ntiles = (kvl + BKV - 1) / BKV;
  const bool has_tail = (kvl % BKV) != 0;
  auto load_tile = [&](int t) {     // t = index of the tile being requested
    if (has_tail && t == ntiles - 1) load_kv(std::true_type{}); else load_kv(std::false_type{});
  };

  // ---- prologue: tile 0 into buffer 0, tile 1 in flight
  load_tile(0);
  store_kv(0);
  if (ntiles > 1) load_tile(1);
  __syncthreads();
  const int nfast = has_tail ? ntiles - 1 : ntiles;     // tiles the incremental-pointer path may load
  constexpr std::integral_constant<int, NOMAX ? 0 : 2> FM_LOOP{};
  int t = 0;
  if constexpr (ALIGNED) {
    if constexpr (NOMAX) {                               // peeled first tile (the launcher guarantees >= 3 tiles, no tail)
      store_kv(1);
      load_kv(std::false_type{});
      compute(0, 0, std::false_type{}, std::integral_constant<int, 1>{});
      __syncthreads();
      t = 1;
    }
    for (; t + 2 < nfast; ++t) {                         // steady state: no clamp, no mask, no div/mod
      store_kv((t + 1) & 1);                             // registers hold tile t+1 (requested one iteration ago)
      load_kv(std::false_type{});                        // tile t+2
      compute(t & 1, t * BKV, std::false_type{}, FM_LOOP);
      __syncthreads();
    }
  }
  for (; t < ntiles - 1; ++t) {                          // generic / last iterations
    store_kv((t + 1) & 1);
    if (t + 2 < ntiles) load_tile(t + 2);
    compute(t & 1, t * BKV, std::false_type{}, FM_LOOP);
    __syncthreads();
  }
  if (has_tail) compute((ntiles - 1) & 1, (ntiles - 1) * BKV, std::true_type{}, FM_LOOP);
  else compute((ntiles - 1) & 1, (ntiles - 1) * BKV, std::false_type{}, FM_LOOP);

  // ---- finalize: lane holds O[q = l31][d = 32*mt + 8*qd + 4*g + j] for each query sub-tile
  float inv[QT], l_fin[QT];
  bool bad = false;
#pragma unroll
  for (int qs = 0; qs < QT; ++qs) {
    float l_tot;
    if constexpr (ONES) {
      constexpr int LM = D / 32, LR = ((D % 32) / 8) * 4;       // O^T row D sits in half g = 0, register LR of tile LM
      static_assert((D % 32) % 8 == 0, "row D must map to half 0");
      l_tot = __shfl(oacc[qs][LM][LR], l31);
    } else {
      l_tot = l_run + __shfl_xor(l_run, 32);
    }
    bad = bad || !(l_tot < 1.2676506e30f) || !(l_tot > 0.f);
    inv[qs] = oscale / l_tot;
    l_fin[qs] = l_tot;
  }
  if constexpr (NOMAX) {
    if (__syncthreads_or(bad ? 1 : 0)) return false;       // (every wave is also done with the LDS images)
  }
  if constexpr (!TWO) {
    if (p.lse != nullptr) {      // training: log2 of the softmax denominator (m_off: raw-score maximum in the fma form, else the log2-domain offset)
#pragma unroll
      for (int qs = 0; qs < QT; ++qs)
        if (q_ok[qs] && g == 0)
          p.lse[((int64_t)grp * p.heads + head) * p.q_len + q_idx[qs]] =
              __builtin_amdgcn_logf(l_fin[qs]) + (OFS == OFS_FMA ? m_off[qs] * p.scale_log2 : m_off[qs]);
    }
  }
  if constexpr (TWO && !SET1) {
    if (p.K2 != nullptr) {                                 // first key set of two: keep the normalised result in registers
#pragma unroll
      for (int qs = 0; qs < QT; ++qs)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) osave[qs][mt][r] = oacc[qs][mt][r] * inv[qs];
      return true;
    }
  }
#pragma unroll
  for (int qs = 0; qs < QT; ++qs) {
    if (q_ok[qs]) {
      uint16_t* orow = p.O + map_row(p.om, grp, q_idx[qs]) * p.om.ld + hoff;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int d = 32 * mt + 8 * qd + 4 * g;
          if (d < D) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[j] = oacc[qs][mt][4 * qd + j] * inv[qs];
              if constexpr (TWO && SET1) v[j] += osave[qs][mt][4 * qd + j];
            }
            if (p.accumulate) {
              const u32x2_t prev = *reinterpret_cast<const u32x2_t*>(orow + d);
              v[0] += lo16(prev[0]); v[1] += hi16(prev[0]); v[2] += lo16(prev[1]); v[3] += hi16(prev[1]);
            }
            u32x2_t o;
            o[0] = pack16(v[0], v[1]);
            o[1] = pack16(v[2], v[3]);
            *reinterpret_cast<u32x2_t*>(orow + d) = o;
          }
        }
    }
  }
  return true;
  };    // pass

  if constexpr (TRY_NOMAX) {
    if (!pass(std::true_type{}, std::false_type{})) pass(std::false_type{}, std::false_type{});
  } else {
    pass(std::false_type{}, std::false_type{});
  }
  if constexpr (TWO) {
    if (p.K2 != nullptr) {
      __syncthreads();                                     // the LDS images are re-used
      pass(std::false_type{}, std::true_type{});
    }
  }
}

bool map_ok(const a3d_rowmap* m, int head_dim) {
  return m && m->gdiv > 0 && m->seg_len > 0 && m->ld > 0 && m->ld % 8 == 0 && head_dim % 8 == 0;
}


template <int D, int BKV, int QT, int OFS, int VAR = 0, int NM = 0>
void launch(bool aligned, int groups, hipStream_t s, const AttnParams& p) {
  const int q_tiles = (p.q_len + 128 * QT - 1) / (128 * QT);
  const dim3 grid((unsigned)(p.heads * q_tiles), (unsigned)groups);
  if (aligned) flash_attn_kernel<D, BKV, QT, OFS, true, VAR, NM><<<grid, dim3(256), 0, s>>>(p);
  else flash_attn_kernel<D, BKV, QT, OFS, false, 0><<<grid, dim3(256), 0, s>>>(p);
}

template <int D, int BKV, int OFS>
void launch_two(bool aligned, int groups, hipStream_t s, const AttnParams& p) {
  const int q_tiles = (p.q_len + 127) / 128;
  const dim3 grid((unsigned)(p.heads * q_tiles), (unsigned)groups);
  if (aligned) flash_attn_kernel<D, BKV, 1, OFS, true, 0, 0, true><<<grid, dim3(256), 0, s>>>(p);
  else flash_attn_kernel<D, BKV, 1, OFS, false, 0, 0, true><<<grid, dim3(256), 0, s>>>(p);
}

}  // namespace

static int flash_attn_impl(a3d_stream_t stream, const void* Q, const void* K, const void* V, void* O,
                           const a3d_rowmap* qmap, const a3d_rowmap* kmap, const a3d_rowmap* omap,
                           int groups, int heads, int head_dim, int64_t q_len, int64_t kv_len,
                           float scale, float out_scale, int accumulate, float* lse, unsigned int* counters = nullptr) {
  if (!Q || !K || !V || !O || groups <= 0 || heads <= 0 || q_len <= 0 || kv_len <= 0) return A3D_EINVAL;
  if (!map_ok(qmap, head_dim) || !map_ok(kmap, head_dim) || !map_ok(omap, head_dim)) return A3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(V)) & 15u) return A3D_EINVAL;
  if (reinterpret_cast<uintptr_t>(O) & 7u) return A3D_EINVAL;
  if (groups > 65535 || q_len > 0x3fffffffLL || kv_len > 0x3fffffffLL || kmap->seg_len > 0x3fffffffLL) return A3D_EINVAL;
  AttnParams p{};
  p.Q = (const uint16_t*)Q; p.K = (const uint16_t*)K; p.V = (const uint16_t*)V; p.O = (uint16_t*)O;
  p.qm = *qmap; p.km = *kmap; p.om = *omap;
  p.heads = heads; p.q_len = (int)q_len; p.kv_len = (int)kv_len;
  if (accumulate & ~(A3D_ATTN_ACCUMULATE | A3D_ATTN_CAUSAL | A3D_ATTN_EXACT | A3D_ATTN_PLAIN)) return A3D_EINVAL;
  p.scale_log2 = scale * 1.4426950408889634f; p.out_scale = out_scale; p.accumulate = accumulate & 1; p.causal = (accumulate >> 1) & 1;
  const bool exact = (accumulate & A3D_ATTN_EXACT) != 0, plain = (accumulate & A3D_ATTN_PLAIN) != 0;
  p.lse = lse;
  p.counters = counters;
  if (p.causal && head_dim != 64 && head_dim != 160) return A3D_EUNSUPPORTED;     // offered on the raw-score (fma) kernels only
  const int bkv = head_dim == 160 ? 32 : 64;
  const bool aligned = (kmap->seg_len % bkv == 0) || (kv_len <= kmap->seg_len);
  hipStream_t s = (hipStream_t)stream;
  switch (head_dim) {
    case 40:
      if (q_len <= 128) { launch<40, 64, 1, OFS_PAD>(aligned, groups, s, p); break; }
      // LDS-DMA staged kernel (flash_attn_dm.hip): the default for the long aligned shapes in both storage types (max-free first
      // pass + P·V through the 16x16x32 MFMA; fp16 storage with a sampled offset and fp16's narrower window, DM_BIAS there);
      // A3D_ATTN_EXACT runs its exact pass only, A3D_ATTN_PLAIN the generic kernel below.
      if (!plain && aligned && kv_len % 64 == 0 && kv_len >= 256 && q_len >= 256) {      // (a workgroup covers 512 queries)
        if (int rc = A3D_FN(a3d_launch_flash_dm)(exact ? 4 : 5, groups, s, p)) return rc;
        break;
      }
      launch<40, 64, 2, OFS_PAD, 0>(aligned, groups, s, p);
      break;
    case 80:
      // LDS-DMA staged kernel (flash_attn_dm80.hip): the default from 512 keys (both storage types).  Otherwise: long sequences take
      // two query sub-tiles per wave and 32-key tiles — every K / V^T fragment read feeds two MFMAs, which relieves the LDS port that
      // bounds the one-sub-tile kernel (+4-8 %, profiles/r2_microbench_flash80_ab.log); short ones keep one sub-tile per wave.
      if (!plain && aligned && kv_len % 64 == 0 && kv_len >= 512 && q_len >= 256) {
        if (int rc = A3D_FN(a3d_launch_flash_dm80)(exact ? 0 : 1, groups, s, p)) return rc;
        break;
      }
      if (q_len >= 2048 && kv_len >= 2048) { launch<80, 32, 2, OFS_FMA>(aligned, groups, s, p); break; }
      launch<80, 64, 1, OFS_ACC>(aligned, groups, s, p);
      break;
    case 64: launch<64, 64, 1, OFS_FMA>(aligned, groups, s, p); break;        // CLIP text tower (12 heads of 64)
    case 160:
#ifndef A3D_EXP_R5_PATHS      // (measurement build: the generic kernel as in rounds 1-5)
      // LDS-DMA staged kernel on eight waves (flash_attn_dm160.hip, round 6): 64-key tiles from 256 keys, both storage types
      // (16-byte stores: O and its row pitch must be 16-byte multiples — the entry point itself only asks for 8)
      // (no lower bound on q_len: the choice must depend on the KEY side only — a view-sharded rank sees a quarter of the queries against the same gathered
      // keys, 64 at level 3, and has to reproduce the unsharded launch bit for bit; a workgroup with few valid queries costs what any workgroup costs)
      if (!plain && !p.causal && kv_len % 64 == 0 && kv_len >= 256 && ((kmap->seg_len % 64 == 0) || (kv_len <= kmap->seg_len)) &&
          (reinterpret_cast<uintptr_t>(O) & 15u) == 0 && omap->ld % 8 == 0) {
        if (int rc = A3D_FN(a3d_launch_flash_dm160)(exact ? 0 : 1, groups, s, p)) return rc;
        break;
      }
#endif
      launch<160, 32, 1, OFS_FMA>(aligned, groups, s, p);
      break;
    default: return A3D_EUNSUPPORTED;
  }
  return a3d_launch_status();
}

extern "C" int A3D_FN(a3d_flash_attn)(a3d_stream_t stream, const void* Q, const void* K, const void* V, void* O,
                                   const a3d_rowmap* qmap, const a3d_rowmap* kmap, const a3d_rowmap* omap,
                                   int groups, int heads, int head_dim, int64_t q_len, int64_t kv_len,
                                   float scale, float out_scale, int accumulate) {
  return flash_attn_impl(stream, Q, K, V, O, qmap, kmap, omap, groups, heads, head_dim, q_len, kv_len, scale, out_scale, accumulate, nullptr);
}

// Training forward: the same attention, and per query the log2 of its softmax denominator (lse2 [groups][heads][q_len] floats) —
// what a3d_flash_attn_bwd's statistics pass would recompute.  The row sum comes out of the kernel's own accumulators, so this costs one
// float store per query and head.
extern "C" int A3D_FN(a3d_flash_attn_lse)(a3d_stream_t stream, const void* Q, const void* K, const void* V, void* O,
                                       const a3d_rowmap* qmap, const a3d_rowmap* kmap, const a3d_rowmap* omap,
                                       int groups, int heads, int head_dim, int64_t q_len, int64_t kv_len,
                                       float scale, float out_scale, int accumulate, float* lse2) {
  if (!lse2) return A3D_EINVAL;
  return flash_attn_impl(stream, Q, K, V, O, qmap, kmap, omap, groups, heads, head_dim, q_len, kv_len, scale, out_scale, accumulate, lse2);
}

// Diagnostics: a3d_flash_attn that also books, per launch of an LDS-DMA staged kernel (head_dim 40 / 80, long aligned K/V), how many
// workgroups left the max-free fast path — counters[0] += sent to the exact pass by the fp16 spread vote, [1] += exact re-runs after an
// overflow, [2] += workgroups launched (device words, caller-zeroed; other kernels leave them alone).  Same results as a3d_flash_attn.
extern "C" int A3D_FN(a3d_flash_attn_counted)(a3d_stream_t stream, const void* Q, const void* K, const void* V, void* O,
                                           const a3d_rowmap* qmap, const a3d_rowmap* kmap, const a3d_rowmap* omap,
                                           int groups, int heads, int head_dim, int64_t q_len, int64_t kv_len,
                                           float scale, float out_scale, int accumulate, unsigned int* counters) {
  if (!counters || (reinterpret_cast<uintptr_t>(counters) & 3u)) return A3D_EINVAL;
  return flash_attn_impl(stream, Q, K, V, O, qmap, kmap, omap, groups, heads, head_dim, q_len, kv_len, scale, out_scale, accumulate, nullptr, counters);
}

// Two key sets in one launch: O = out_scale * softmax(Q K^T * scale) V + out_scale2 * softmax(Q K2^T * scale) V2 (+ previous O if
// accumulate).  Replaces the text attention + per-adapter image-token attention + `hidden += scale * ip` sequence of the IPAdapter
// processor (attention_processor.py:233, 254-283) by one pass over Q and O.  head_dim 40 / 80 (others: A3D_EUNSUPPORTED, call twice).
extern "C" int A3D_FN(a3d_flash_attn2)(a3d_stream_t stream, const void* Q, const void* K, const void* V, const void* K2, const void* V2, void* O,
                                    const a3d_rowmap* qmap, const a3d_rowmap* kmap, const a3d_rowmap* kmap2, const a3d_rowmap* omap,
                                    int groups, int heads, int head_dim, int64_t q_len, int64_t kv_len, int64_t kv_len2,
                                    float scale, float out_scale, float out_scale2, int accumulate) {
  if (!Q || !K || !V || !K2 || !V2 || !O || groups <= 0 || heads <= 0 || q_len <= 0 || kv_len <= 0 || kv_len2 <= 0) return A3D_EINVAL;
  if (!map_ok(qmap, head_dim) || !map_ok(kmap, head_dim) || !map_ok(kmap2, head_dim) || !map_ok(omap, head_dim)) return A3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(V) | reinterpret_cast<uintptr_t>(K2) |
       reinterpret_cast<uintptr_t>(V2)) & 15u) return A3D_EINVAL;
  if (reinterpret_cast<uintptr_t>(O) & 7u) return A3D_EINVAL;
  if (groups > 65535 || q_len > 0x3fffffffLL || kv_len > 0x3fffffffLL || kv_len2 > 0x3fffffffLL || kmap->seg_len > 0x3fffffffLL ||
      kmap2->seg_len > 0x3fffffffLL) return A3D_EINVAL;
  if (head_dim != 40 && head_dim != 80) return A3D_EUNSUPPORTED;
  AttnParams p{};
  p.Q = (const uint16_t*)Q; p.K = (const uint16_t*)K; p.V = (const uint16_t*)V; p.O = (uint16_t*)O;
  p.K2 = (const uint16_t*)K2; p.V2 = (const uint16_t*)V2;
  p.qm = *qmap; p.km = *kmap; p.km2 = *kmap2; p.om = *omap;
  p.heads = heads; p.q_len = (int)q_len; p.kv_len = (int)kv_len; p.kv_len2 = (int)kv_len2;
  if (accumulate & ~(A3D_ATTN_ACCUMULATE | A3D_ATTN_PLAIN)) return A3D_EINVAL;
  p.scale_log2 = scale * 1.4426950408889634f; p.out_scale = out_scale; p.out_scale2 = out_scale2; p.accumulate = accumulate & 1; p.causal = 0;
  const bool aligned = ((kmap->seg_len % 64 == 0) || (kv_len <= kmap->seg_len)) && ((kmap2->seg_len % 64 == 0) || (kv_len2 <= kmap2->seg_len));
  hipStream_t s = (hipStream_t)stream;
  if (head_dim == 40 && !(accumulate & A3D_ATTN_PLAIN)) {      // level 0: the register-resident cross-attention kernel (cross_attn.hip)
    const int rc = A3D_FN(a3d_launch_cross_attn40)(groups, s, p);
    if (rc != A3D_EUNSUPPORTED) return rc;
  }
  if (head_dim == 40) launch_two<40, 64, OFS_PAD>(aligned, groups, s, p);
  else launch_two<80, 64, OFS_ACC>(aligned, groups, s, p);
  return a3d_launch_status();
}
