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

// tuning knobs shared by the bf16 and the fp16 build of this file (defined once, in the bf16 object)
#ifdef A3D_STORAGE_F16
extern int g_flash_variant;
#else
int g_flash_variant = 0;   // a3d_tune_flash(): 0 = default dispatch, 5 = plain kernel only, 16 = ping-pong instead of interleaved (A/B timing)
#endif

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
  const int64_t kgbase = (grp / kmp.gdiv) * kmp.ga + (grp % kmp.gdiv) * kmp.gb;
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

// =====================================================================================================
// Ping-pong variant for the level-0 multi-view attentions (D = 40, long aligned K/V).
//
// PMC profile of the kernel above (QT = 2): matrix pipe busy 50 %, VALU issue 58 %, and the two nearly
// serialised — the two resident waves of a SIMD come from different workgroups and drift into the same phase.
// Here one 512-thread workgroup puts TWO of its waves on every SIMD (wave w and w+4) and keeps them half a
// tile apart with workgroup barriers, so that at any time one of them is in its matrix block
// [PV(t-1), QK^T(t)] and the other in its vector block [softmax(t), K/V staging]:
//
//     barrier interval      group A (waves 0-3)            group B (waves 4-7)
//        (#1,#2)             softmax(0), stage tile 1        PV(-), QK(0)
//        (#2,#3)             PV(0), QK(1)                    softmax(0), stage tile 2
//        (#3,#4)             softmax(1), stage tile 2        PV(0), QK(1)          ...
//
// Each group stages half of every K/V tile (A: tile t+1, B: tile t+2 during the vector block of tile t), K is
// double- and V triple-buffered in LDS; all 8 waves (512 queries) share one staged copy, which also halves the
// LDS write traffic and staging VALU per FLOP again.  Per-query arithmetic is identical to the kernel above.
// ABL (timing ablations only, results are wrong): 1 = no exp, 2 = no max / lazy check, 4 = no K/V staging after the
// prologue, 8 = no QK^T MFMAs, 16 = no PV MFMAs.
template <int D, int ABL = 0>
__global__ __launch_bounds__(512, 2) void flash_attn_pp_kernel(const AttnParams p) {
  static_assert(D == 40, "ping-pong variant is instantiated for head_dim 40");
  constexpr int BKV = 64, NU = 2, QT = 2, VROW = BKV + 8;
  constexpr int DK = 48, KS = 3, MT = 2, KROW = DK + 8, DCH = D / 8;
  constexpr int KS_ELEMS = BKV * KROW, VT_ELEMS = MT * 32 * VROW;
  constexpr int KSHARE = BKV * DCH / 2, VSHARE = (BKV / 4) * DCH / 2;      // 160 K chunks, 40 V items per group
  constexpr int KS_PAD = D / 16, G_PAD = (D % 16) / 8;
  constexpr int BQ = 512;

  __shared__ __attribute__((aligned(16))) uint16_t smem[2 * KS_ELEMS + 3 * VT_ELEMS];
  uint16_t* const Ks0 = smem;
  uint16_t* const Vt0 = smem + 2 * KS_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int grp_w = wid >> 2, wq = wid & 3, gt = tid & 255;
  const int l31 = lane & 31, g = lane >> 5;
  const int head = blockIdx.x % p.heads;
  const int qt = blockIdx.x / p.heads;
  const int64_t grp = blockIdx.y;
  const int64_t hoff = (int64_t)head * D;

  for (int i = tid; i < 2 * BKV * (DK - D); i += 512) {
    const int b = i / (BKV * (DK - D)), rem = i % (BKV * (DK - D));
    const int c = rem % (DK - D);
    Ks0[b * KS_ELEMS + (rem / (DK - D)) * KROW + D + c] = (c == 0) ? ONE16 : 0;
  }
  for (int i = tid; i < 3 * BKV; i += 512) Vt0[(i / BKV) * VT_ELEMS + D * VROW + (i % BKV)] = ONE16;

  // ---- Q^T fragments (pre-scaled), two 32-query sub-tiles per wave
  int q_idx[QT];
  bool q_ok[QT];
  u32x4_t qf[QT][KS];
#pragma unroll
  for (int qs = 0; qs < QT; ++qs) {
    q_idx[qs] = qt * BQ + grp_w * 256 + wq * 64 + qs * 32 + l31;
    q_ok[qs] = q_idx[qs] < p.q_len;
    const int64_t q_row = map_row(p.qm, grp, q_ok[qs] ? q_idx[qs] : p.q_len - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = 16 * ks + 8 * g;
      if (d0 < D) {
        u32x4_t w = *reinterpret_cast<const u32x4_t*>(p.Q + q_row * p.qm.ld + hoff + d0);
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = pack16(lo16(w[j]) * p.scale_log2, hi16(w[j]) * p.scale_log2);
        qf[qs][ks] = w;
      } else {
        qf[qs][ks] = u32x4_t{0u, 0u, 0u, 0u};
      }
    }
  }

  // ---- this thread's share of every K/V tile
  const int64_t ld = p.km.ld;
  const int64_t kgbase = (grp / p.km.gdiv) * p.km.ga + (grp % p.km.gdiv) * p.km.gb;
  const uint32_t seg_len = (uint32_t)p.km.seg_len;
  const int64_t tile_step = (int64_t)BKV * ld;
  const int64_t wrap_step = (p.km.seg_stride - p.km.seg_len) * ld;
  // K chunks on the group's first 160 threads; V items on its last wave, key-quad fastest across lanes so that the
  // 8-byte V^T writes of a 16-lane group fall on 16 different key columns (conflict-free; dim-fastest was 5-way)
  const bool has_k = gt < KSHARE, has_v = gt >= 192 && gt < 192 + VSHARE;
  const int kci = grp_w * KSHARE + (has_k ? gt : 0);
  const int vit = has_v ? gt - 192 : 0;                       // 0..39 within the group
  const int kr = kci / DCH, kc = kci % DCH;
  const int vq = grp_w * (VSHARE / DCH) + vit % (VSHARE / DCH), vc = vit / (VSHARE / DCH);
  const uint16_t* kptr = p.K + hoff + (kgbase + kr) * ld + kc * 8;
  const uint16_t* vptr = p.V + hoff + (kgbase + vq * 4) * ld + vc * 8;
  uint32_t seg_off = 0;
  u32x4_t kreg, vreg[4];
  auto load_share = [&]() {            // reads the tile the pointers stand on, then advances them by one tile
    if (has_k) kreg = *reinterpret_cast<const u32x4_t*>(kptr);
    if (has_v) {
#pragma unroll
      for (int r = 0; r < 4; ++r) vreg[r] = *reinterpret_cast<const u32x4_t*>(vptr + r * ld);
    }
    seg_off += BKV;
    int64_t step = tile_step;
    if (seg_off >= seg_len) { step += wrap_step; seg_off = 0; }
    kptr += step; vptr += step;
  };
  auto store_share = [&](int tile) {
    uint16_t* const Ks = Ks0 + (tile & 1) * KS_ELEMS;
    uint16_t* const Vt = Vt0 + (tile % 3) * VT_ELEMS;
    if (has_k) *reinterpret_cast<u32x4_t*>(Ks + kr * KROW + kc * 8) = kreg;
    if (has_v) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t w0 = vreg[0][j], w1 = vreg[1][j], w2 = vreg[2][j], w3 = vreg[3][j];
        u32x2_t even, odd;
        even[0] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);
        even[1] = __builtin_amdgcn_perm(w3, w2, 0x05040100u);
        odd[0] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);
        odd[1] = __builtin_amdgcn_perm(w3, w2, 0x07060302u);
        *reinterpret_cast<u32x2_t*>(Vt + (vc * 8 + 2 * j) * VROW + vq * 4) = even;
        *reinterpret_cast<u32x2_t*>(Vt + (vc * 8 + 2 * j + 1) * VROW + vq * 4) = odd;
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
  float m_off[QT] = {0.f, 0.f};
  const int krow_off = kperm(l31) * KROW + 8 * g;
  const int vrow_off = l31 * VROW + 8 * g;
  f32x16_t sacc[QT][NU];
  u32x4_t pf[QT][NU][2];

  u32x4_t va[4], vb[4];
  auto prefetch_v = [&](int tile) {          // V^T fragments of `tile` (complete in LDS two phases before they are used)
    const uint16_t* const Vt = Vt0 + (tile % 3) * VT_ELEMS + vrow_off;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        va[mt * 2 + h] = *reinterpret_cast<const u32x4_t*>(Vt + 32 * mt * VROW + 16 * h);
        vb[mt * 2 + h] = *reinterpret_cast<const u32x4_t*>(Vt + 32 * mt * VROW + 32 + 16 * h);
      }
  };
  // Matrix block of tile t: PV(t-1) then QK^T(t).  LDS fragment reads are software-pipelined by hand (the
  // compiler otherwise sinks every ds_read next to its MFMA and exposes ~150 cycles of LDS latency per fragment):
  // two batches of V fragments are in flight before the first MFMA, each K batch is requested one MFMA batch early;
  // sched_barrier(0) pins the read groups above the MFMA groups.
  auto matrix_block = [&](int t, auto with_pv_c) {
    constexpr bool WITH_PV = decltype(with_pv_c)::value;
    const uint16_t* const Ks = Ks0 + (t & 1) * KS_ELEMS + krow_off;
    u32x4_t ka[KS], kb[KS];
    auto load_k = [&](u32x4_t (&kk)[KS], int u) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) kk[ks] = *reinterpret_cast<const u32x4_t*>(Ks + 32 * u * KROW + 16 * ks);
    };
    auto pv_batch = [&](u32x4_t (&vv)[4], int u) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int qs = 0; qs < QT; ++qs) {
            if constexpr ((ABL & 16) == 0) oacc[qs][mt] = mfma32(vv[mt * 2 + h], pf[qs][u][h], oacc[qs][mt]);
            else { oacc[qs][mt][0] += __uint_as_float(vv[mt * 2 + h][0] ^ pf[qs][u][h][0]); }
          }
    };
    auto qk_batch = [&](u32x4_t (&kk)[KS], int u) {
#pragma unroll
      for (int qs = 0; qs < QT; ++qs)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[qs][u][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int qs = 0; qs < QT; ++qs) {
          if constexpr ((ABL & 8) == 0) sacc[qs][u] = mfma32(kk[ks], qf[qs][ks], sacc[qs][u]);
          else { sacc[qs][u][ks] += __uint_as_float(kk[ks][0] ^ qf[qs][ks][0]) * 1e-30f; }
        }
    };
    if constexpr (WITH_PV) {       // va / vb were requested at the end of the previous vector block (before the barrier)
      pv_batch(va, 0);
      load_k(ka, 0);
      __builtin_amdgcn_sched_barrier(0);
      pv_batch(vb, 1);
      load_k(kb, 1);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      load_k(ka, 0);
      load_k(kb, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    qk_batch(ka, 0);
    __builtin_amdgcn_sched_barrier(0);
    qk_batch(kb, 1);
  };
  auto softmax = [&](bool first) {
#pragma unroll
    for (int qs = 0; qs < QT; ++qs) {
      float mx = sacc[qs][0][0];
      if constexpr ((ABL & 2) == 0) {       // four independent max chains (a single chain is latency-bound)
        float m4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) m4[c] = sacc[qs][c & 1][c >> 1];
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) m4[(r & 1) * 2 + u] = fmaxf(m4[(r & 1) * 2 + u], sacc[qs][u][r]);
        mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
      }
      if (__any(first || ((ABL & 2) == 0 && mx > LAZY_THR))) {
        const float mxp = fmaxf(mx, __shfl_xor(mx, 32));
        float delta = first ? mxp : fmaxf(mxp, 0.f);
        const float new_off = round16(m_off[qs] + delta);
        delta = new_off - m_off[qs];
        m_off[qs] = new_off;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[qs][u][r] -= delta;
        if (!first) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qs][mt][r] *= alpha;
        }
        if (g == G_PAD) qf[qs][KS_PAD][0] = pack16(-new_off, 0.f);
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        float e[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) e[r] = (ABL & 1) ? sacc[qs][u][r] : __builtin_amdgcn_exp2f(sacc[qs][u][r]);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < 4; ++j) pf[qs][u][h][j] = pack16(e[8 * h + 2 * j], e[8 * h + 2 * j + 1]);
        if constexpr ((ABL & 32) != 0) {     // experiment: pair every two v_exp with the v_cvt_pk of the previous pair
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
          }
        }
      }
    }
  };

  const int nt = p.kv_len / BKV;              // launcher guarantees kv_len % BKV == 0 and nt >= 2
  // ---- prologue: both groups stage their share of tile 0; B also stages its share of tile 1
  load_share();
  store_share(0);
  if (grp_w == 1) { load_share(); store_share(1); }
  __syncthreads();                            // #0: tile 0 complete
  if (grp_w == 1) __syncthreads();            // #1: B idles through A's first matrix block

  for (int t = 0; t < nt; ++t) {
    const int nxt = t + 1 + grp_w;            // tile this group stages during the vector block of tile t
    // ---- matrix block
    if (nxt < nt && (ABL & 4) == 0) load_share();
    if (t > 0) matrix_block(t, std::true_type{}); else matrix_block(t, std::false_type{});
    __syncthreads();
    // ---- vector block
    softmax(t == 0);
    if (nxt < nt && (ABL & 4) == 0) store_share(nxt);
    prefetch_v(t);                              // for PV(t) right after the barrier
    __syncthreads();
  }
  if (grp_w == 0) __syncthreads();            // balance B's idle barrier
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int qs = 0; qs < QT; ++qs) {
        oacc[qs][mt] = mfma32(va[mt * 2 + h], pf[qs][0][h], oacc[qs][mt]);
        oacc[qs][mt] = mfma32(vb[mt * 2 + h], pf[qs][1][h], oacc[qs][mt]);
      }

  // ---- finalize
#pragma unroll
  for (int qs = 0; qs < QT; ++qs) {
    constexpr int LM = D / 32, LR = ((D % 32) / 8) * 4;
    const float l_tot = __shfl(oacc[qs][LM][LR], l31);
    const float inv = p.out_scale / l_tot;
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
            for (int j = 0; j < 4; ++j) v[j] = oacc[qs][mt][4 * qd + j] * inv;
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
}

// =====================================================================================================
// Interleaved (software-pipelined) variant for the level-0 multi-view attentions (D = 40, long aligned K/V).
//
// Issue-rate measurements on MI355X (tools/ubench_exp.hip, profiles/README.md): a wave that runs only MFMAs next to a
// wave that runs only softmax VALU on the same SIMD (what the ping-pong kernel above arranges) reaches 1.46 PFLOP/s of
// matrix work beside 6.5 T exp/s; two waves per SIMD that EACH carry a fine-grained mix {1 MFMA, 2-3 v_exp, 1 v_cvt_pk,
// 1-2 v_max3} reach 1.9-2.1 PFLOP/s beside 8.5-9.4 T exp/s.  So here every wave is its own pipeline over 32-key
// sub-tiles j = 0, 1, 2, ...: in step j the matrix pipe computes S(j+1) = K·Q^T and O += V^T·P(j-1) while the VALU turns
// S(j) into P(j) (v_exp + v_cvt_pk) and takes the row maximum of S(j+1); the two instruction streams are independent
// inside a step and are emitted in alternating groups pinned with sched_barrier(0), one MFMA per group.
//
//   step j:   MFMA  | QK^T(j+1) (3·QT)                 | PV(j-1) (4·QT)                          |
//             VALU  | exp/cvt of S(j), first part       | exp/cvt of S(j), rest; max of S(j+1)    | vote
//
// Lazy offset as in the kernels above (the offset rides in the spare contraction slot, P may reach 2^6); the vote of
// step j covers S(j+1), i.e. it precedes the exponentiation of S(j+1).  When it fires (rare) everything still relative
// to the old offset is folded into O first — P(j) is accumulated at once and cleared so that the regular PV(j) of the
// next step adds zero — then O is rescaled, S(j+1) re-based and the Q slot rewritten.
// K/V: all waves of the workgroup share one staged copy of every tile (registers -> LDS, one tile ahead); K rows as
// in the kernels above (permuted reads, padded contraction with the constant-1 column), V row-major with a 96-element
// pitch and its constant-1 "dimension" 40, transposed on the way out of LDS by ds_read_b64_tr_b16 (conflict-free at
// this pitch).  K is triple-, V quadruple-buffered: one barrier per 64-key tile.
constexpr int IL_KROW = 56, IL_VPITCH = 96, IL_NKB = 3, IL_NVB = 4;
constexpr int IL_DUMP_OFF = IL_NKB * 64 * IL_KROW + IL_NVB * 64 * IL_VPITCH;        // element offset of the staging dump area
constexpr int IL_SMEM_BYTES = (IL_DUMP_OFF + IL_NKB * 64 * IL_KROW + 64 * 8) * 2;  // dump: 1 KB reachable under any K buffer offset

// ABL (timing ablations, -DA3D_ABLATIONS builds only, results wrong): 1 = no v_exp, 2 = no row max / vote, 4 = no K/V staging after the
// prologue, 8 = no per-tile barrier, 16 = no QK^T MFMAs, 32 = no PV MFMAs, 64 = fragments read from LDS once, 128 = no v_cvt_pk
template <int QT, int NW, int ABL = 0>
__global__ __launch_bounds__(NW * 64, NW / 4) void flash_attn_il_kernel(const AttnParams p) {
  constexpr int D = 40, BKV = 64, KS = 3, MT = 2, KROW = IL_KROW, VPITCH = IL_VPITCH, DCH = 5;
  constexpr int NT = NW * 64;
  constexpr int KS_ELEMS = BKV * KROW, V_ELEMS = BKV * VPITCH;
  constexpr int KS_PAD = 2, G_PAD = 1;                    // fragment slot of contraction index 40
  constexpr int BQ = NW * 32 * QT;
  constexpr int NCH = 2 * BKV * DCH;                      // staging chunks per tile: 320 of K, then 320 of V
  constexpr int CPT = (NCH + NT - 1) / NT;
  constexpr int NEXP = 16 * QT, NCVT = 8 * QT;

  extern __shared__ __attribute__((aligned(16))) uint16_t il_smem[];
  uint16_t* const Ks0 = il_smem;
  uint16_t* const Vs0 = il_smem + IL_NKB * KS_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, g = lane >> 5;
  const int head = blockIdx.x % p.heads;
  const int qt = blockIdx.x / p.heads;
  const int64_t grp = blockIdx.y;
  const int64_t hoff = (int64_t)head * D;

  // one-time LDS init: K columns 40..47 = (1, 0, ..), V "dimensions" 40..47 = (1, 0, ..), 48..63 = 0
  for (int i = tid; i < IL_NKB * BKV * 8; i += NT) Ks0[(i >> 3) * KROW + D + (i & 7)] = ((i & 7) == 0) ? ONE16 : 0;
  for (int i = tid; i < IL_NVB * BKV * 24; i += NT) Vs0[(i / 24) * VPITCH + D + (i % 24)] = ((i % 24) == 0) ? ONE16 : 0;

  // ---- Q^T fragments (pre-scaled by scale * log2 e)
  u32x4_t qf[QT][KS];
#pragma unroll
  for (int qs = 0; qs < QT; ++qs) {
    const int q_idx = qt * BQ + wid * 32 * QT + qs * 32 + l31;
    const int64_t q_row = map_row(p.qm, grp, q_idx < p.q_len ? q_idx : p.q_len - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = 16 * ks + 8 * g;
      if (d0 < D) {
        u32x4_t w = *reinterpret_cast<const u32x4_t*>(p.Q + q_row * p.qm.ld + hoff + d0);
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = pack16(lo16(w[j]) * p.scale_log2, hi16(w[j]) * p.scale_log2);
        qf[qs][ks] = w;
      } else {
        qf[qs][ks] = u32x4_t{0u, 0u, 0u, 0u};
      }
    }
  }

  // ---- K/V staging: chunk c = tid + NT*i of a tile (c < 320: K row c/5, 16-byte piece c%5; else V); wave-uniform roles
  const int64_t ld = p.km.ld;
  const int64_t kgbase = (grp / p.km.gdiv) * p.km.ga + (grp % p.km.gdiv) * p.km.gb;
  const uint32_t seg_len = (uint32_t)p.km.seg_len;
  const int64_t tile_step = (int64_t)BKV * ld;
  const int64_t wrap_step = (p.km.seg_stride - p.km.seg_len) * ld;
  // Every thread moves exactly CPT chunks per tile and the code is branch-free (a branch inside a pipeline step lets the
  // optimiser sink the step's v_exp below it): threads whose last chunk does not exist re-read their first chunk and park
  // it in a dump area behind the buffers.
  const uint16_t* src[CPT];
  int dst[CPT];                 // LDS element offset from il_smem, buffer offset excluded
  bool isk[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = tid + NT * i;
    const int c_wave = __builtin_amdgcn_readfirstlane(wid) * 64 + NT * i;      // first chunk of this wave: roles are wave-uniform
    const bool act = c_wave < NCH;
    isk[i] = c_wave < BKV * DCH;
    const int cc = isk[i] ? c : (act ? c - BKV * DCH : 0);
    const int row = cc / DCH, ch = cc % DCH;
    src[i] = act ? (isk[i] ? p.K : p.V) + hoff + (kgbase + row) * ld + ch * 8 : src[0];
    dst[i] = isk[i] ? row * KROW + ch * 8 : (act ? IL_NKB * KS_ELEMS + row * VPITCH + ch * 8 : IL_DUMP_OFF + lane * 8);
    if (!act) isk[i] = true;      // dump: no per-tile buffer offset (kbuf_off is applied; the dump area allows for it)
  }
  uint32_t seg_off = 0;
  u32x4_t sreg[CPT];
  auto load_tile = [&]() __attribute__((always_inline)) {           // reads the tile the pointers stand on, then advances them
#pragma unroll
    for (int i = 0; i < CPT; ++i) sreg[i] = *reinterpret_cast<const u32x4_t*>(src[i]);
    seg_off += BKV;
    int64_t stp = tile_step;
    if (seg_off >= seg_len) { stp += wrap_step; seg_off = 0; }
#pragma unroll
    for (int i = 0; i < CPT; ++i) src[i] += stp;
  };
  auto store_tile = [&](int tile) __attribute__((always_inline)) {
    const int kb = (tile % IL_NKB) * KS_ELEMS, vb = (tile % IL_NVB) * V_ELEMS;
#pragma unroll
    for (int i = 0; i < CPT; ++i) *reinterpret_cast<u32x4_t*>(il_smem + (isk[i] ? kb : vb) + dst[i]) = sreg[i];
  };

  f32x16_t oacc[QT][MT];
#pragma unroll
  for (int qs = 0; qs < QT; ++qs)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qs][mt][r] = 0.f;
  float m_off[QT];
#pragma unroll
  for (int qs = 0; qs < QT; ++qs) m_off[qs] = 0.f;
  const int krow_off = kperm(l31) * KROW + 8 * g;
  // V^T fragment by two transposing reads: the 16-lane group (lane >> 4) covers dims 16*(q4 & 1) .. +15 of the 32-row
  // tile and key half q4 >> 1 (= g); lane i of the group addresses key i/4, dims 4*(i%4) .. +3 and receives dim i
  const int i16 = lane & 15, q4 = lane >> 4;
  const int vlane_off = (8 * (q4 >> 1) + (i16 >> 2)) * VPITCH + 16 * (q4 & 1) + 4 * (i16 & 3);

  f32x16_t sA[QT], sB[QT];
  u32x4_t pA[QT][2], pB[QT][2];
#pragma unroll
  for (int qs = 0; qs < QT; ++qs)
#pragma unroll
    for (int h = 0; h < 2; ++h) { pA[qs][h] = u32x4_t{0u, 0u, 0u, 0u}; pB[qs][h] = u32x4_t{0u, 0u, 0u, 0u}; }

  auto read_v = [&](u32x4_t (&vf)[MT][2], const uint16_t* Vsub) __attribute__((always_inline)) {      // Vsub: V buffer + 32-key sub-tile + vlane_off
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const u32x2_t lo = lds_tr16_b64(Vsub + (16 * h) * VPITCH + 32 * mt);
        const u32x2_t hi = lds_tr16_b64(Vsub + (16 * h + 4) * VPITCH + 32 * mt);
        vf[mt][h] = u32x4_t{lo[0], lo[1], hi[0], hi[1]};
      }
  };

  // K fragments: during step j kf holds K sub-tile j+1 (the first MFMAs of the step need it at once); it is re-read for
  // the NEXT step (K sub-tile j+2) right behind the last QK^T MFMA, so no step starts by waiting for LDS.  The V
  // fragments are first needed ~200 cycles into the step and are requested at its start.  Everything a step reads was
  // staged at least one barrier earlier.
  u32x4_t kf[KS];
  // One pipeline step (see the header): Knext / Vsub / Vflush point at the sub-tiles of S(j+2), P(j-1) and P(j).
  auto step = [&](auto do_qk_c, auto do_pv_c, auto pf_k_c, f32x16_t (&sCur)[QT], f32x16_t (&sNext)[QT], u32x4_t (&pCur)[QT][2],
                  u32x4_t (&pPrev)[QT][2], const uint16_t* Knext, const uint16_t* Vsub, const uint16_t* Vflush, auto&& hook) __attribute__((always_inline)) {
    constexpr bool DO_QK = decltype(do_qk_c)::value, DO_PV = decltype(do_pv_c)::value, PF_K = decltype(pf_k_c)::value;
    constexpr int NQK = DO_QK ? KS * QT : 0, NPV = DO_PV ? 2 * MT * QT : 0, NS = NQK + NPV;
    constexpr int NMAX = (DO_QK && (ABL & 2) == 0) ? 8 * QT : 0;                        // v_max3 over S(j+1): two chains of 4 per query sub-tile
    constexpr int MAX_S0 = DO_PV ? NQK + 1 : NS;                    // first slot that may carry max ops (S(j+1) complete)
    constexpr int MAX_SLOTS = (NS - MAX_S0) < 4 ? (NS - MAX_S0) : 4;  // front-loaded: the vote must be ready well before the step ends
    constexpr int VOTE_SLOT = MAX_S0 + MAX_SLOTS + 1;               // combine + compare + ballot here (if such a slot exists)
    auto prefetch_k = [&]() __attribute__((always_inline)) {
      if constexpr (PF_K && (ABL & 64) == 0) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = *reinterpret_cast<const u32x4_t*>(Knext + 16 * ks);
      }
    };
    u32x4_t vf[MT][2];
    u32x2_t vh[MT][2][2];                 // halves of the V^T fragments, requested one per early slot
    auto read_vh = [&](auto i_c) __attribute__((always_inline)) {
      constexpr int i = decltype(i_c)::value, mt = i / 4, h = (i / 2) % 2, rr = i % 2;
      vh[mt][h][rr] = lds_tr16_b64(Vsub + (16 * h + 4 * rr) * VPITCH + 32 * mt);
    };
    constexpr bool SPREAD_V = DO_PV && DO_QK && (ABL & 64) == 0;
    if constexpr (DO_PV && !DO_QK && (ABL & 64) == 0) read_v(vf, Vsub);
    if constexpr ((ABL & 64) != 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) { vf[mt][0] = kf[0]; vf[mt][1] = kf[1]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    float e[NEXP];
    float mx[QT][2];
    auto do_max = [&](auto k_c) __attribute__((always_inline)) {         // max op k: link k / (2 QT) of chain (k / QT) % 2 of query sub-tile k % QT
      constexpr int k = decltype(k_c)::value;                            // (consecutive ops belong to different chains)
      constexpr int qs = k % QT, c = (k / QT) % 2, i = k / (2 * QT), b = 8 * c;
      if constexpr (i == 0) mx[qs][c] = vmax3(sNext[qs][b], sNext[qs][b + 1], sNext[qs][b + 2]);
      else if constexpr (i < 3) mx[qs][c] = vmax3(mx[qs][c], sNext[qs][b + 2 * i + 1], sNext[qs][b + 2 * i + 2]);
      else mx[qs][c] = vmax3(mx[qs][c], sNext[qs][b + 7], mx[qs][c]);
    };
    auto do_cvt = [&](auto c_c) __attribute__((always_inline)) {
      constexpr int c = decltype(c_c)::value;
      constexpr int qs = c / 8, h = (c / 4) % 2, jj = c % 4;
      if constexpr ((ABL & 128) == 0) pCur[qs][h][jj] = pack16(e[2 * c], e[2 * c + 1]);
      else pCur[qs][h][jj] = __float_as_uint(e[2 * c]) ^ __float_as_uint(e[2 * c + 1]);
    };
    uint64_t ballot = 0;
    auto vote = [&]() __attribute__((always_inline)) {
      bool need = false;
#pragma unroll
      for (int qs = 0; qs < QT; ++qs) {
        mx[qs][0] = vmax3(mx[qs][0], mx[qs][1], mx[qs][1]);
        need = need || (mx[qs][0] > LAZY_THR);
      }
      ballot = __ballot(need);
    };
    static_for<NS>([&](auto s_c) __attribute__((always_inline)) {
      constexpr int s = decltype(s_c)::value;
      if constexpr (s < NQK) {
        constexpr int ks = s / QT, qs = s % QT;
        if constexpr (ks == 0) {
          f32x16_t z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.f;
          if constexpr ((ABL & 16) == 0) sNext[qs] = mfma32(kf[ks], qf[qs][ks], z);
          else sNext[qs][0] = __uint_as_float(kf[ks][0] ^ qf[qs][ks][0]) * 1e-30f;
        } else {
          if constexpr ((ABL & 16) == 0) sNext[qs] = mfma32(kf[ks], qf[qs][ks], sNext[qs]);
          else sNext[qs][ks] = __uint_as_float(kf[ks][1] ^ qf[qs][ks][1]) * 1e-30f;
        }
      } else {
        constexpr int i = s - NQK, mt = i / (2 * QT), h = (i / QT) % 2, qs = i % QT;
        if constexpr ((ABL & 32) == 0) oacc[qs][mt] = mfma32(vf[mt][h], pPrev[qs][h], oacc[qs][mt]);
        else oacc[qs][mt][h] += __uint_as_float(vf[mt][h][0] ^ pPrev[qs][h][0]);
      }
      if constexpr (s == NQK - 1) prefetch_k();                    // last QK^T MFMA issued: K fragments for the next step
      if constexpr (SPREAD_V && s < NQK) {
        constexpr int R0 = 8 * s / NQK, R1 = 8 * (s + 1) / NQK;
        static_for<R1 - R0>([&](auto r_c) __attribute__((always_inline)) { read_vh(std::integral_constant<int, R0 + decltype(r_c)::value>{}); });
        if constexpr (s == NQK - 1) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int h = 0; h < 2; ++h) vf[mt][h] = u32x4_t{vh[mt][h][0][0], vh[mt][h][0][1], vh[mt][h][1][0], vh[mt][h][1][1]};
        }
      }
      hook(s_c);
      constexpr int E0 = NEXP * s / NS, E1 = NEXP * (s + 1) / NS;
      static_for<E1 - E0>([&](auto x_c) __attribute__((always_inline)) {
        constexpr int x = E0 + decltype(x_c)::value;
        if constexpr ((ABL & 1) == 0) e[x] = __builtin_amdgcn_exp2f(sCur[x / 16][x % 16]);
        else e[x] = sCur[x / 16][x % 16];
      });
      constexpr int C0 = (s == 0) ? 0 : (NEXP * (s - 1) / NS) / 2, C1 = E0 / 2;      // pairs completed by earlier slots
      static_for<C1 - C0>([&](auto c_c) __attribute__((always_inline)) { do_cvt(std::integral_constant<int, C0 + decltype(c_c)::value>{}); });
      if constexpr (NMAX > 0 && MAX_SLOTS > 0 && s >= MAX_S0 && s < MAX_S0 + MAX_SLOTS) {
        constexpr int M0 = NMAX * (s - MAX_S0) / MAX_SLOTS, M1 = NMAX * (s - MAX_S0 + 1) / MAX_SLOTS;
        static_for<M1 - M0>([&](auto k_c) __attribute__((always_inline)) { do_max(std::integral_constant<int, M0 + decltype(k_c)::value>{}); });
      }
      if constexpr (NMAX > 0 && MAX_SLOTS > 0 && s == VOTE_SLOT && VOTE_SLOT < NS) vote();
      __builtin_amdgcn_sched_barrier(0);
    });
    {   // tail: the last pairs, and all max ops when no PV slot followed the QK^T slots
      constexpr int C0 = (NS == 0) ? 0 : (NEXP * (NS - 1) / NS) / 2;
      static_for<NCVT - C0>([&](auto c_c) __attribute__((always_inline)) { do_cvt(std::integral_constant<int, C0 + decltype(c_c)::value>{}); });
      if constexpr (NMAX > 0 && MAX_SLOTS == 0)
        static_for<NMAX>([&](auto k_c) __attribute__((always_inline)) { do_max(k_c); });
      if constexpr (NMAX > 0 && (MAX_SLOTS == 0 || VOTE_SLOT >= NS)) vote();
    }
    if constexpr (DO_QK && (ABL & 2) == 0) {
      if (ballot != 0) {
        // fold P(j) into O now (exact: it is relative to the old offset), clear it, then move the offset
        u32x4_t vfl[MT][2];
        read_v(vfl, Vflush);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int qs = 0; qs < QT; ++qs) oacc[qs][mt] = mfma32(vfl[mt][h], pCur[qs][h], oacc[qs][mt]);
#pragma unroll
        for (int qs = 0; qs < QT; ++qs) {
#pragma unroll
          for (int h = 0; h < 2; ++h) pCur[qs][h] = u32x4_t{0u, 0u, 0u, 0u};
          const float mxp = fmaxf(mx[qs][0], __shfl_xor(mx[qs][0], 32));
          float delta = fmaxf(mxp, 0.f);
          const float new_off = round16(m_off[qs] + delta);
          delta = new_off - m_off[qs];
          m_off[qs] = new_off;
          const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
          for (int r = 0; r < 16; ++r) sNext[qs][r] -= delta;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qs][mt][r] *= alpha;
          if (g == G_PAD) qf[qs][KS_PAD][0] = pack16(-new_off, 0.f);
        }
      }
    }
  };

  const int nt = p.kv_len / BKV;              // launcher guarantees kv_len % 64 == 0, nt >= 4, aligned segments
  // ---- prologue: tiles 0 and 1 in LDS, tile 2 in registers
  load_tile(); store_tile(0);
  load_tile(); store_tile(1);
  load_tile();
  __syncthreads();
  {   // S(0) and the first offset
    const uint16_t* const Ksub = Ks0 + krow_off;
#pragma unroll
    for (int qs = 0; qs < QT; ++qs) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sA[qs][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        sA[qs] = mfma32(*reinterpret_cast<const u32x4_t*>(Ksub + 16 * ks), qf[qs][ks], sA[qs]);
      float mx = sA[qs][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sA[qs][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float new_off = round16(mx);
      m_off[qs] = new_off;
#pragma unroll
      for (int r = 0; r < 16; ++r) sA[qs][r] -= new_off;
      if (g == G_PAD) qf[qs][KS_PAD][0] = pack16(-new_off, 0.f);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kf[ks] = *reinterpret_cast<const u32x4_t*>(Ksub + 32 * KROW + 16 * ks);    // K sub-tile 1
  }

  auto iteration = [&](int t, auto first_c, auto last_c, auto store_c, auto load_c) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
    constexpr bool STORE = decltype(store_c)::value, LOAD = decltype(load_c)::value;
    // staging rides in the VALU groups of the even step: tile t+2 (in registers since the previous iteration) goes to LDS
    // early, tile t+3 is requested behind it
    auto stage_hook = [&](auto s_c) __attribute__((always_inline)) {
      constexpr int s = decltype(s_c)::value;
      if constexpr ((ABL & 4) == 0) {
        if constexpr (STORE && s == 1) store_tile(t + 2);
        if constexpr (LOAD && s == 2 && (ABL & 256) == 0) load_tile();
      }
    };
    auto odd_hook = [&](auto s_c) __attribute__((always_inline)) {      // experiment (ABL bit 256): request the next tile from the odd step
      constexpr int s = decltype(s_c)::value;
      if constexpr ((ABL & 4) == 0 && (ABL & 256) != 0 && LOAD && s == 1) load_tile();
    };
    const uint16_t* const Kn = Ks0 + ((t + 1) % IL_NKB) * KS_ELEMS + krow_off;
    const uint16_t* const Vp = Vs0 + ((t + IL_NVB - 1) % IL_NVB) * V_ELEMS + vlane_off;
    const uint16_t* const Vt = Vs0 + (t % IL_NVB) * V_ELEMS + vlane_off;
    // even step j = 2t:  S(2t+1) from K(t) keys 32..63 (in kf), O += V(t-1)[32..63] P(2t-1), P(2t) from S(2t);
    //                    fetches K(t+1) keys 0..31 for the odd step
    step(std::true_type{}, std::integral_constant<bool, !FIRST>{}, std::integral_constant<bool, !LAST>{}, sA, sB, pA, pB,
         Kn, Vp + 32 * VPITCH, Vt, stage_hook);
    // odd step j = 2t+1: S(2t+2) from K(t+1) keys 0..31 (in kf), O += V(t)[0..31] P(2t), P(2t+1) from S(2t+1);
    //                    fetches K(t+1) keys 32..63 for the next even step
    step(std::integral_constant<bool, !LAST>{}, std::true_type{}, std::integral_constant<bool, !LAST>{}, sB, sA, pB, pA,
         Kn + 32 * KROW, Vt, Vt + 32 * VPITCH, odd_hook);
    if constexpr (!LAST && (ABL & 8) == 0) {
      if constexpr ((ABL & 512) != 0) {      // experiment: LDS writes drained, global loads left in flight across the barrier
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      } else {
        __syncthreads();
      }
    }
  };
  constexpr std::true_type Y{};
  constexpr std::false_type N{};
  iteration(0, Y, N, Y, Y);
  for (int t = 1; t < nt - 3; ++t) iteration(t, N, N, Y, Y);
  iteration(nt - 3, N, N, Y, N);              // tile nt-1 goes to LDS, nothing left to request
  iteration(nt - 2, N, N, N, N);
  iteration(nt - 1, N, Y, N, N);
  {   // O += V(nt-1)[32..63] P(2nt-1)
    u32x4_t vf[MT][2];
    read_v(vf, Vs0 + ((nt - 1) % IL_NVB) * V_ELEMS + vlane_off + 32 * VPITCH);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int qs = 0; qs < QT; ++qs) oacc[qs][mt] = mfma32(vf[mt][h], pB[qs][h], oacc[qs][mt]);
  }

  // ---- finalize (as in the kernels above)
#pragma unroll
  for (int qs = 0; qs < QT; ++qs) {
    constexpr int LM = D / 32, LR = ((D % 32) / 8) * 4;
    const float l_tot = __shfl(oacc[qs][LM][LR], l31);
    const float inv = p.out_scale / l_tot;
    const int q_idx = qt * BQ + wid * 32 * QT + qs * 32 + l31;
    if (q_idx < p.q_len) {
      uint16_t* orow = p.O + map_row(p.om, grp, q_idx) * p.om.ld + hoff;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int d = 32 * mt + 8 * qd + 4 * g;
          if (d < D) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = oacc[qs][mt][4 * qd + j] * inv;
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
}

template <int QT, int NW, int ABL = 0>
int launch_il(int groups, hipStream_t s, const AttnParams& p) {
  static uint64_t attr_done = 0;
  if (int rc = a3d_once_per_device(attr_done, [] {
        return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_attn_il_kernel<QT, NW, ABL>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, IL_SMEM_BYTES); })) return rc;
  constexpr int BQ = NW * 32 * QT;
  const int q_tiles = (p.q_len + BQ - 1) / BQ;
  flash_attn_il_kernel<QT, NW, ABL><<<dim3((unsigned)(p.heads * q_tiles), (unsigned)groups), dim3(NW * 64), IL_SMEM_BYTES, s>>>(p);
  return a3d_launch_status();
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

extern int g_a3d_ta_pix;      // temporal_attn.hip

#ifndef A3D_STORAGE_F16
extern "C" int a3d_tune_flash(int variant) {
  if (variant == 11 || variant == 12 || variant == 14) { g_a3d_ta_pix = variant - 10; return A3D_OK; }
  if (variant == 8 || variant == 17 || (variant >= 20 && variant <= 43)) { g_flash_variant = variant; return A3D_OK; }     // head dim 80: force two / one query sub-tile per wave
#ifdef A3D_ABLATIONS
  if ((variant < 0 || variant > 7) && variant != 13 && variant != 15 && variant != 16 && variant != 19 && (variant < 1000 || variant >= 2024)) return A3D_EINVAL;
#else
  if (variant != 0 && variant != 5 && variant != 6 && variant != 7 && variant != 13 && variant != 15 && variant != 16 && variant != 19) return A3D_EINVAL;
#endif
  g_flash_variant = variant;
  return A3D_OK;
}
#endif

static int flash_attn_impl(a3d_stream_t stream, const void* Q, const void* K, const void* V, void* O,
                           const a3d_rowmap* qmap, const a3d_rowmap* kmap, const a3d_rowmap* omap,
                           int groups, int heads, int head_dim, int64_t q_len, int64_t kv_len,
                           float scale, float out_scale, int accumulate, float* lse) {
  if (!Q || !K || !V || !O || groups <= 0 || heads <= 0 || q_len <= 0 || kv_len <= 0) return A3D_EINVAL;
  if (!map_ok(qmap, head_dim) || !map_ok(kmap, head_dim) || !map_ok(omap, head_dim)) return A3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(V)) & 15u) return A3D_EINVAL;
  if (reinterpret_cast<uintptr_t>(O) & 7u) return A3D_EINVAL;
  if (groups > 65535 || q_len > 0x3fffffffLL || kv_len > 0x3fffffffLL || kmap->seg_len > 0x3fffffffLL) return A3D_EINVAL;
  AttnParams p{};
  p.Q = (const uint16_t*)Q; p.K = (const uint16_t*)K; p.V = (const uint16_t*)V; p.O = (uint16_t*)O;
  p.qm = *qmap; p.km = *kmap; p.om = *omap;
  p.heads = heads; p.q_len = (int)q_len; p.kv_len = (int)kv_len;
  p.scale_log2 = scale * 1.4426950408889634f; p.out_scale = out_scale; p.accumulate = accumulate & 1; p.causal = (accumulate >> 1) & 1;
  p.lse = lse;
  if (p.causal && head_dim != 64 && head_dim != 160) return A3D_EUNSUPPORTED;     // offered on the raw-score (fma) kernels only
  const bool no_lse = lse == nullptr;      // the log-sum-exp output exists in the LDS-DMA kernels and in the plain kernel only
  const int bkv = head_dim == 160 ? 32 : 64;
  const bool aligned = (kmap->seg_len % bkv == 0) || (kv_len <= kmap->seg_len);
  hipStream_t s = (hipStream_t)stream;
  switch (head_dim) {
    case 40:
      if (q_len <= 128) { launch<40, 64, 1, OFS_PAD>(aligned, groups, s, p); break; }
#ifdef A3D_ABLATIONS
      if (g_flash_variant >= 1000) {      // timing ablations / experiments of the interleaved kernel: 1000 + ABL bits
        const int a = g_flash_variant - 1000;
        const bool w4 = false;
        int rc = A3D_EUNSUPPORTED;
#define A3D_IL_ABL(X) if (a == X) rc = w4 ? launch_il<2, 4, X>(groups, s, p) : launch_il<2, 8, X>(groups, s, p);
        A3D_IL_ABL(0) A3D_IL_ABL(1) A3D_IL_ABL(2) A3D_IL_ABL(4) A3D_IL_ABL(8) A3D_IL_ABL(12) A3D_IL_ABL(16) A3D_IL_ABL(32) A3D_IL_ABL(48)
        A3D_IL_ABL(64) A3D_IL_ABL(76) A3D_IL_ABL(129) A3D_IL_ABL(131) A3D_IL_ABL(207) A3D_IL_ABL(124)
        A3D_IL_ABL(256) A3D_IL_ABL(512) A3D_IL_ABL(768)
#undef A3D_IL_ABL
        if (rc != A3D_OK) return rc;
        break;
      }
#endif
      // LDS-DMA staged kernel (flash_attn_dm.hip).  bf16 storage: the default for the long aligned shapes (flags 5: max-free first
      // pass + P·V through the 16x16x32 MFMA); a3d_tune_flash(20 + flags) forces a flag set, 19 forces the interleaved kernel below.
      // fp16 storage runs the same flag set with a sampled offset and fp16's narrower window (flash_attn_dm.hip, DM_BIAS).
      {
        const bool long_aligned = aligned && kv_len % 64 == 0 && kv_len >= 256 && q_len >= 256;      // (a workgroup covers 512 queries)
        int dm_flags = (g_flash_variant >= 20 && g_flash_variant <= 35) ? g_flash_variant - 20 : -1;
        if (g_flash_variant == 0) dm_flags = 5;
        if (long_aligned && dm_flags >= 0) {
          if (int rc = A3D_FN(a3d_launch_flash_dm)(dm_flags, groups, s, p)) return rc;
          break;
        }
      }
      // interleaved kernel (default for the long aligned shapes): 8 waves x 64 queries; A/B variants 7 = 4 waves x 64, 13 = 4 x 128,
      // 15 = 8 x 32; 16 = the ping-pong kernel instead
      if (no_lse && g_flash_variant != 5 && g_flash_variant != 16 && aligned && kv_len % 64 == 0 && kv_len >= 256 && q_len >= 512) {
        int rc = 0;
        if (g_flash_variant == 7) rc = launch_il<2, 4>(groups, s, p);
        else if (g_flash_variant == 13) rc = launch_il<4, 4>(groups, s, p);
        else if (g_flash_variant == 15) rc = launch_il<1, 8>(groups, s, p);
        else rc = launch_il<2, 8>(groups, s, p);
        if (rc != A3D_OK) return rc;
        break;
      }
      if (no_lse && g_flash_variant != 5 && aligned && kv_len % 64 == 0 && kv_len >= 128 && q_len >= 512) {
        const int q_tiles = (int)((q_len + 511) / 512);
        const dim3 grid((unsigned)(heads * q_tiles), (unsigned)groups);
        switch (g_flash_variant) {
#ifdef A3D_ABLATIONS   // timing ablations: WRONG results by construction; only in -DA3D_ABLATIONS builds (profiles/README.md)
          case 1: flash_attn_pp_kernel<40, 1><<<grid, dim3(512), 0, s>>>(p); break;
          case 2: flash_attn_pp_kernel<40, 3><<<grid, dim3(512), 0, s>>>(p); break;
          case 3: flash_attn_pp_kernel<40, 4><<<grid, dim3(512), 0, s>>>(p); break;
          case 4: flash_attn_pp_kernel<40, 32><<<grid, dim3(512), 0, s>>>(p); break;
#endif
          default: flash_attn_pp_kernel<40, 0><<<grid, dim3(512), 0, s>>>(p); break;
        }
        break;
      }
      launch<40, 64, 2, OFS_PAD, 0>(aligned, groups, s, p);
      break;
    case 80:
      // long sequences (level 1 of the 512-px configurations): two query sub-tiles per wave, 32-key tiles — every K / V^T
      // fragment read feeds two MFMAs, which relieves the LDS port that bounds the one-sub-tile kernel (+4-8 %,
      // profiles/r2_microbench_flash80_ab.log); short ones keep one sub-tile per wave (more workgroups).  a3d_tune_flash(8 | 17) forces either.
      // LDS-DMA staged kernel (flash_attn_dm80.hip): the default from 512 tokens (both storage types); a3d_tune_flash(42) forces it (43: its exact pass only),
      // 8 / 17 force the kernels below
      {
        const bool long_aligned = aligned && kv_len % 64 == 0 && kv_len >= 256;
        int dm_flags = g_flash_variant == 42 ? 1 : (g_flash_variant == 43 ? 0 : -1);
        if (g_flash_variant == 0 && kv_len >= 512 && q_len >= 256) dm_flags = 1;
        if (long_aligned && dm_flags >= 0) {
          if (int rc = A3D_FN(a3d_launch_flash_dm80)(dm_flags, groups, s, p)) return rc;
          break;
        }
      }
#ifndef A3D_STORAGE_F16
      // bf16, long aligned sequences: max-free first pass with the offset in the MFMA's C operand (a3d_tune_flash(40); 41: one query sub-tile)
      if (g_flash_variant == 40 && aligned && kv_len % 32 == 0 && kv_len >= 128) { launch<80, 32, 2, OFS_ACC, 0, 1>(true, groups, s, p); break; }
      if (g_flash_variant == 41 && aligned && kv_len % 64 == 0 && kv_len >= 256) { launch<80, 64, 1, OFS_ACC, 0, 1>(true, groups, s, p); break; }
#endif
      if (g_flash_variant == 8 || (g_flash_variant != 17 && q_len >= 2048 && kv_len >= 2048)) { launch<80, 32, 2, OFS_FMA>(aligned, groups, s, p); break; }
      launch<80, 64, 1, OFS_ACC>(aligned, groups, s, p);
      break;
    case 64: launch<64, 64, 1, OFS_FMA>(aligned, groups, s, p); break;        // CLIP text tower (12 heads of 64)
    case 160: launch<160, 32, 1, OFS_FMA>(aligned, groups, s, p); break;
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
  p.scale_log2 = scale * 1.4426950408889634f; p.out_scale = out_scale; p.out_scale2 = out_scale2; p.accumulate = accumulate & 1; p.causal = 0;
  const bool aligned = ((kmap->seg_len % 64 == 0) || (kv_len <= kmap->seg_len)) && ((kmap2->seg_len % 64 == 0) || (kv_len2 <= kmap2->seg_len));
  hipStream_t s = (hipStream_t)stream;
  if (head_dim == 40) launch_two<40, 64, OFS_PAD>(aligned, groups, s, p);
  else launch_two<80, 64, OFS_ACC>(aligned, groups, s, p);
  return a3d_launch_status();
}
