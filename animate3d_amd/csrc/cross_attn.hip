// Cross attention against a handful of tokens (IPAdapter processor, attention_processor.py:233, 254-283: 77 text tokens + 4 IP-Adapter
// image tokens, each key set with its own softmax, O = attn(Q, K, V) + s2 * attn(Q, K2, V2)), head_dim 40, level 0 of the UNet.
//
// Why a kernel of its own (round 4).  The work is ~0.1 FLOP per byte: Q in, O out.  The generic kernel (flash_attn_kernel, one workgroup
// = one head x 128 queries) ran this launch at 1.0 TB/s of algorithmic bytes: per 128 queries of ONE head it initialises LDS padding,
// stages two K/V tiles through LDS behind barriers and drains its accumulators — 32 768 such workgroups at level 0 whose prologue and
// epilogue dominate, and a token row's eight 80-byte head slices are touched by eight different workgroups.  Here
//   * a workgroup = `heads` waves (wave = head) x up to 1 024 queries of one group: the 8 waves walk the same token rows at the same
//     time, so every 640-byte row of Q is read and of O is written in full lines by one CU;
//   * K and V^T of the wave's head live in REGISTERS as MFMA operands for the whole workgroup (3 + 1 key tiles of 32: <= 96 + 32 keys):
//     no LDS, no barrier, no per-tile staging — per 32 queries a wave issues 3 x 16-byte loads, 28 MFMAs, ~60 v_exp and 5 x 8-byte stores;
//   * exact softmax per key set (row maximum over the <= 96 + 32 scores a lane holds for its query), row sums through a ones row of V^T
//     as in the other attention kernels, fp32 accumulation, one rounding of the summed result.
// Layouts follow flash_attn_kernel: S^T = K·Q^T with v_mfma_f32_32x32x16 (lane = query, kperm makes 8 consecutive result registers 8
// consecutive keys, so P^T is directly the B operand of O^T = V^T·P^T).
#include "flash_common.h"

namespace {

constexpr int XA_D = 40, XA_KS = 3, XA_MT = 2;       // contraction 40 -> 48 (zero padded), O^T rows 41 -> 64 (dims, ones row, zeros)

// KT1 / KT2: 32-key tiles of the first / second key set
template <int KT1, int KT2>
__global__ __launch_bounds__(512, 2) void cross_attn40_kernel(const AttnParams p, const int q_per_wg) {
  const int lane = threadIdx.x & 63;
  const int head = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (head >= p.heads) return;                        // (no barriers in this kernel)
  const int l31 = lane & 31, g = lane >> 5;
  const int64_t grp = blockIdx.y;
  const int64_t hoff = (int64_t)head * XA_D;
  const int q_begin = blockIdx.x * q_per_wg;
  const int q_end = q_begin + q_per_wg < p.q_len ? q_begin + q_per_wg : p.q_len;

  // a key set lies inside ONE row-map segment here (checked by the launcher): row(key) = row(0) + key
  const int64_t kb1 = map_row(p.km, grp, 0), kb2 = map_row(p.km2, grp, 0);
  const int64_t qb = map_row(p.qm, grp, 0), ob = map_row(p.om, grp, 0);      // likewise the queries of a group
  // ---- K fragments (A operand of S^T = K·Q^T): lane (row l31 -> key 32 kt + kperm(l31), half g) holds dims 16 ks + 8 g .. + 7
  //      Contraction slot 40 (dims 40..47 are padding) masks the keys past the end of the set inside the matrix pipe: Q carries 1.0 there,
  //      K carries 0 for a real key and -30000 for a missing one (its score then is ~-30000 log2 units: exp2 gives an exact 0), so the
  //      softmax needs no per-score compare / select.
  auto load_k = [&](const uint16_t* K, int64_t kb, int64_t ld, int kv_len, int kt, int ks) -> u32x4_t {
    const int key = 32 * kt + kperm(l31);
    const int d0 = 16 * ks + 8 * g;
    if (d0 == XA_D) return u32x4_t{key < kv_len ? 0u : (uint32_t)f2h(-30000.f), 0u, 0u, 0u};
    if (key < kv_len && d0 < XA_D) return *reinterpret_cast<const u32x4_t*>(K + (kb + key) * ld + hoff + d0);
    return u32x4_t{0u, 0u, 0u, 0u};
  };
  // ---- V^T fragments (A operand of O^T = V^T·P^T): lane (row l31 -> O^T row 32 mt + l31, half g) holds keys 32 kt + 16 hh + 8 g .. + 7
  //      of that row: dims < 40 from V, row 40 = ones (row sums), rows 41.. = zeros
  auto load_vt = [&](const uint16_t* V, int64_t kb, int64_t ld, int kv_len, int mt, int kt, int hh) -> u32x4_t {
    const int d = 32 * mt + l31;
    uint16_t e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int key = 32 * kt + 16 * hh + 8 * g + j;
      uint16_t v = d == XA_D ? ONE16 : (uint16_t)0;
      if (d < XA_D && key < kv_len) v = V[(kb + key) * ld + hoff + d];
      e[j] = v;
    }
    u32x4_t r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = (uint32_t)e[2 * j] | ((uint32_t)e[2 * j + 1] << 16);
    return r;
  };
  u32x4_t kf1[KT1][XA_KS], vt1[XA_MT][KT1][2], kf2[KT2][XA_KS], vt2[XA_MT][KT2][2];
#pragma unroll
  for (int kt = 0; kt < KT1; ++kt) {
#pragma unroll
    for (int ks = 0; ks < XA_KS; ++ks) kf1[kt][ks] = load_k(p.K, kb1, p.km.ld, p.kv_len, kt, ks);
#pragma unroll
    for (int mt = 0; mt < XA_MT; ++mt)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) vt1[mt][kt][hh] = load_vt(p.V, kb1, p.km.ld, p.kv_len, mt, kt, hh);
  }
#pragma unroll
  for (int kt = 0; kt < KT2; ++kt) {
#pragma unroll
    for (int ks = 0; ks < XA_KS; ++ks) kf2[kt][ks] = load_k(p.K2, kb2, p.km2.ld, p.kv_len2, kt, ks);
#pragma unroll
    for (int mt = 0; mt < XA_MT; ++mt)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) vt2[mt][kt][hh] = load_vt(p.V2, kb2, p.km2.ld, p.kv_len2, mt, kt, hh);
  }

  // one key set for the 32 queries whose (pre-scaled) Q^T fragments are qf: returns the normalised O^T tiles scaled by `oscale`
  auto attend = [&](auto kt_c, const u32x4_t (&kf)[decltype(kt_c)::value][XA_KS], const u32x4_t (&vt)[XA_MT][decltype(kt_c)::value][2], int kv_len,
                    const u32x4_t (&qf)[XA_KS], float oscale, f32x16_t (&o)[XA_MT]) __attribute__((always_inline)) {
    constexpr int KT = decltype(kt_c)::value;
    f32x16_t s[KT];
    float mx = -1e30f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < XA_KS; ++ks) s[kt] = mfma32(kf[kt][ks], qf[ks], s[kt]);
#pragma unroll
      // (missing keys sit at ~-30000: they never win.  Plain fmaxf, not the v_max3 inline-asm helper: the compiler pads the MFMA-result ->
      // VALU-read hazard only for instructions it can see, and an unpadded read right behind the MFMA returns a stale maximum — any maximum
      // gives a correct softmax, so every parity test passed, but the bits changed from run to run)
      for (int r = 0; r < 16; r += 2) mx = fmaxf(mx, fmaxf(s[kt][r], s[kt][r + 1]));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
#pragma unroll
    for (int mt = 0; mt < XA_MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[mt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        u32x4_t pf;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          pf[jj] = pack16(__builtin_amdgcn_exp2f(s[kt][8 * hh + 2 * jj] - mx), __builtin_amdgcn_exp2f(s[kt][8 * hh + 2 * jj + 1] - mx));
#pragma unroll
        for (int mt = 0; mt < XA_MT; ++mt) o[mt] = mfma32(vt[mt][kt][hh], pf, o[mt]);
      }
    // row sum = O^T row 40 = tile 1, row 8 = register 4 of the lanes of half 0 (query l31); only the registers that hold dims < 40 are scaled
    const float inv = oscale / __shfl(o[1][4], l31);
#pragma unroll
    for (int r = 0; r < 16; ++r) o[0][r] *= inv;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[1][r] *= inv;
  };

  // Q^T fragments of the 32 queries from q0 (raw: scaling happens after the prefetch has landed); slot 40 of half 1 = 1.0 (the mask slot)
  auto load_q = [&](int q0, u32x4_t (&raw)[XA_KS]) __attribute__((always_inline)) {
    int qi = q0 + l31;
    qi = qi < q_end ? qi : q_end - 1;
    const uint16_t* qp = p.Q + (qb + qi) * p.qm.ld + hoff;
#pragma unroll
    for (int ks = 0; ks < XA_KS; ++ks) {
      const int d0 = 16 * ks + 8 * g;
      raw[ks] = d0 < XA_D ? *reinterpret_cast<const u32x4_t*>(qp + d0) : u32x4_t{(uint32_t)ONE16, 0u, 0u, 0u};
    }
  };
  // Two sub-tiles of Q are in flight ahead of the one being attended to (a CU holds one workgroup = 8 waves: the bytes in flight,
  // not the arithmetic, set this kernel's speed)
  u32x4_t qn0[XA_KS], qn1[XA_KS];
  load_q(q_begin, qn0);
  load_q(q_begin + 32, qn1);
  for (int q0 = q_begin; q0 < q_end; q0 += 32) {
    const int q_idx = q0 + l31;
    const bool q_ok = q_idx < q_end;
    u32x4_t qf[XA_KS];
#pragma unroll
    for (int ks = 0; ks < XA_KS; ++ks) {
      const int d0 = 16 * ks + 8 * g;
      qf[ks] = qn0[ks];
      qn0[ks] = qn1[ks];
      if (d0 < XA_D) {
#pragma unroll
        for (int j = 0; j < 4; ++j) qf[ks][j] = pack16(lo16(qf[ks][j]) * p.scale_log2, hi16(qf[ks][j]) * p.scale_log2);
      }
    }
    load_q(q0 + 64, qn1);                               // (clamped to the last query row past the end)
    f32x16_t o1[XA_MT], o2[XA_MT];
    attend(std::integral_constant<int, KT1>{}, kf1, vt1, p.kv_len, qf, p.out_scale, o1);
    attend(std::integral_constant<int, KT2>{}, kf2, vt2, p.kv_len2, qf, p.out_scale2, o2);
    // The two halves of a wave hold alternate 4-column groups of one row (half 0: dims 8 k .. + 3, half 1: 8 k + 4 .. + 7).  One
    // v_permlane32_swap per packed word hands each half 8 CONSECUTIVE columns of two groups, so a row leaves with two 16-byte stores and
    // one 8-byte store per lane instead of five 8-byte ones (the stores, not the arithmetic, were the longest part of a sub-tile).
    u32x2_t w[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int mt = k >> 2, qd = k & 3;
      w[k][0] = pack16(o1[mt][4 * qd] + o2[mt][4 * qd], o1[mt][4 * qd + 1] + o2[mt][4 * qd + 1]);
      w[k][1] = pack16(o1[mt][4 * qd + 2] + o2[mt][4 * qd + 2], o1[mt][4 * qd + 3] + o2[mt][4 * qd + 3]);
    }
    if (q_ok) {
      uint16_t* orow = p.O + (ob + q_idx) * p.om.ld + hoff;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {          // groups 2 pr, 2 pr + 1: half 0 ends up with dims 16 pr .. + 7, half 1 with 16 pr + 8 .. + 15
        u32x4_t v;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const auto r = __builtin_amdgcn_permlane32_swap(w[2 * pr][c], w[2 * pr + 1][c], false, false);
          v[c] = r[0];
          v[2 + c] = r[1];
        }
        *reinterpret_cast<u32x4_t*>(orow + 16 * pr + 8 * g) = v;
      }
      *reinterpret_cast<u32x2_t*>(orow + 32 + 4 * g) = w[4];
    }
  }
}

}  // namespace

// returns A3D_EUNSUPPORTED when the shape is not this kernel's (the caller then takes the generic two-key-set kernel): head_dim 40,
// <= 8 heads, <= 96 + 32 keys, each key set inside ONE row-map segment, no accumulation into O, O rows 16-byte aligned (the kernel
// stores 16 bytes per lane at head offsets of 80 bytes: base and row pitch must both be multiples of 16 bytes — a3d_flash_attn2 itself only asks for 8)
int A3D_FN(a3d_launch_cross_attn40)(int groups, hipStream_t s, const AttnParams& p) {
  if (p.heads > 8 || p.kv_len > 96 || p.kv_len2 > 32 || p.accumulate || p.lse) return A3D_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(p.O) & 15u) || (p.om.ld % 8) != 0) return A3D_EUNSUPPORTED;
  if (p.kv_len > p.km.seg_len || p.kv_len2 > p.km2.seg_len || p.q_len > p.qm.seg_len || p.q_len > p.om.seg_len || p.q_len < 256) return A3D_EUNSUPPORTED;
  // up to 1 024 queries per workgroup: the K / V^T register images are built once per workgroup (96 two-byte gathers per lane)
  int q_per_wg = p.q_len >= 2048 ? 1024 : (p.q_len >= 1024 ? 512 : 256);
  const unsigned blocks = (unsigned)((p.q_len + q_per_wg - 1) / q_per_wg);
  cross_attn40_kernel<3, 1><<<dim3(blocks, (unsigned)groups), dim3(512), 0, s>>>(p, q_per_wg);
  return a3d_launch_status();
}
