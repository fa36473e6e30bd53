// Level-0 multi-view / first-frame attention (head_dim 40, long aligned K/V) with LDS-DMA staging and a max-free softmax.
// Replaces xformers.ops.memory_efficient_attention at attention_processor.py:405, 416, 656 for the 16 384-key level-0 shapes
// (the launch that is half of the denoise step).  Same software pipeline over 32-key sub-tiles as flash_attn_il_kernel
// (flash_attn.hip): in step j the matrix pipe computes S(j+1) = K·Q^T and O += V^T·P(j-1) while the VALU turns S(j) into P(j).
// What is different, and why (profiles/README.md, round 3):
//
//  * K/V staging by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write_b128 (13 LDS cycles each, bank
//    conflicts on the 112-byte K pitch), no per-thread pointer arithmetic: the tile bases are SGPRs advanced by SALU, the
//    per-lane part is one 32-bit offset.  A 64-key tile is 640 chunks of 16 B = 10 wave-instructions; every wave issues
//    exactly two per tile (one full, one with 16 lanes), so the hand-counted wait before the per-tile barrier is vmcnt(2).
//    Tiles are requested three ahead into a ring of 8 buffers; LDS reads stay in flight across the barrier.
//  * Dense LDS images (80-byte rows, nothing padded, nothing initialised per tile).  K rows are read with 16-byte reads:
//    5 r mod 16 is a bijection, so any 16 rows that differ mod 16 are conflict-free.  V rows are stored in a 4x4-transposed
//    order inside every 16-key group (the DMA source address does the permutation) so that the four keys one
//    ds_read_b64_tr_b16 group touches lie 4 rows = 80 dwords = 16 banks apart.  The constant parts of the operands — the
//    1.0 in contraction slot 40 of K that carries the softmax offset, the ones "dimension" 40 of V that produces the row
//    sums, the zero padding up to 48 / 64 — are not part of the images: the lanes that would read them point at a small
//    constant region instead (all lanes of a read that share an address are one broadcast).
//  * NO running row maximum.  P = exp2(S - m) is stored in 16 bits and accumulated in fp32; the result does not depend on m as long
//    as nothing overflows or vanishes.  m is fixed per query before the key loop (maximum of 32 sample scores + DM_BIAS, carried in
//    contraction slot 40 of Q) and nothing is checked inside the loop; at the end the row sum that the matrix pipe produces anyway
//    (O^T row 40) must be positive, finite and < 2^100, otherwise the workgroup discards its result and re-runs with the exact
//    running maximum (run_exact: the arithmetic of flash_attn_kernel).  bf16 storage: P has fp32's exponent range, the window around
//    the estimate is ~2^100 wide either way and the first 32 keys are sample enough.  fp16 storage: P holds 2^-24 .. 2^16, see DM_BIAS.
#include "flash_common.h"

namespace {

constexpr int DM_ROWB = 80;                        // bytes per K / V row in LDS
constexpr int DM_UNITB = 32 * DM_ROWB;             // one 32-key sub-tile of K or of V
constexpr int DM_TILEB = 4 * DM_UNITB;             // [K keys 0..63 | V keys 0..63 (rows permuted)]
constexpr int DM_VOFF = 2 * DM_UNITB;              // V image inside a tile buffer
constexpr int DM_RING = 8;
constexpr int DM_CK = DM_RING * DM_TILEB;          // K constant chunk (16 B): 1.0 in contraction slot 40, zeros in 41..47
constexpr int DM_CV = DM_CK + 80;                  // V constant region: (1,0,0,0) pieces at DM_CV + {0, 80, 1280, 1360}, zero elsewhere; its first
                                                   // bank is 4 mod 8 (mod 16), where no data lane of the same read lands
constexpr int DM_CV_BYTES = 1408;
constexpr int DM_SAMPLE = DM_CV + DM_CV_BYTES;     // one 32-key K sub-tile of sample keys (fp16 storage: the offset estimate)
constexpr int DM_SMEM_BYTES = DM_SAMPLE + DM_UNITB;
constexpr float DM_L_BAD = 1.2676506e30f;          // 2^100: beyond this the max-free result is not trusted
// Max-free offset = maximum of 32 sample scores + DM_BIAS (log2 units).  P is stored in 16 bits: bf16 has fp32's exponent range, so any
// row maximum within ~2^100 of the estimate works and the first 32 keys are sample enough.  fp16 holds 2^-24 .. 2^16: the window is
// [estimate - 4, estimate + 20) — the sample is 32 keys spread evenly over the whole key range (one extra 2.5-KB DMA and 3 MFMAs per
// query sub-tile), the bias is small so that the row maximum stays a normal number, and a row that still overflows (P = inf, the
// row sum is not finite) sends its workgroup to the exact pass like any other overflow.  Round 6: the bias grows with the sample's
// spread (flash_common.h: f16_sampled_bias — up to 12, the window then ends 28 units above the sample maximum), and a row whose sample
// predicts an overflow even so never starts the max-free pass (the workgroup votes).
#ifdef A3D_STORAGE_F16
constexpr float DM_BIAS = F16_BIAS;
constexpr bool DM_SAMPLED = true;
#else
constexpr float DM_BIAS = 40.f;
constexpr bool DM_SAMPLED = false;
#endif

extern __shared__ __attribute__((aligned(16))) uint8_t dm_smem[];

A3D_DEV u32x4_t dm_lds128(uint32_t off) { return *reinterpret_cast<const u32x4_t*>(dm_smem + off); }
A3D_DEV u32x2_t dm_ldstr(uint32_t off) { return lds_tr16_b64(reinterpret_cast<const uint16_t*>(dm_smem + off)); }

// FLAGS: 1 = max-free softmax with exact re-run (bf16 storage only), 2 = static s_setprio(1) for the second-dispatched half
//        of the workgroup (MI355X_MICROARCH.md "static priority for the younger half"), 4 = O^T += V^T·P^T through
//        v_mfma_f32_16x16x32: O^T has 41 useful rows (40 dims + the ones row), i.e. 48 in 16-row tiles instead of 64 in
//        32-row tiles: 24 instead of 28 MFMA-equivalents per 64 x 64 tile.  The 32x32 score tile keeps a query in lane & 31,
//        the 16x16x32 B operand wants it in lane & 15: one v_permlane16_swap per packed pair of P moves the odd 16-lane rows
//        of the first key half against the even rows of the second (16 per 64 x 64 tile).
// QT: query sub-tiles of 32 per wave.  2 = 8 waves x 64 queries (two waves per SIMD, up to 256 registers); 1 = 16 waves x 32 queries
//     (four waves per SIMD, 128 registers: twice the LDS fragment reads per MFMA, but four instruction streams per SIMD to
//     overlap matrix and vector work and to hide LDS latency and barrier skew).  Both stage one copy of K/V for 512 queries.
template <int FLAGS, int QT>
__global__ __launch_bounds__(1024 / QT, 4 / QT) void flash_attn_dm_kernel(const AttnParams p) {
  constexpr int D = 40, KS = 3, MT = 2, NW = 16 / QT, NT = 64 * NW, BQ = 512;
  constexpr int NDMA = QT;                                // LDS-DMA instructions per wave and tile (QT = 1: waves 0..9 one each)
  constexpr int KS_PAD = 2, G_PAD = 1;                    // fragment slot of contraction index 40
  constexpr int NEXP = 16 * QT, NCVT = 8 * QT;
  constexpr bool PV16 = (FLAGS & 4) != 0;
  constexpr bool TRY_NOMAX = (FLAGS & 1) != 0;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int w = __builtin_amdgcn_readfirstlane(wid);
  const int l31 = lane & 31, g = lane >> 5, i16 = lane & 15, q4 = lane >> 4;
  const int head = blockIdx.x % p.heads;
  const int qt = blockIdx.x / p.heads;
  const int64_t grp = blockIdx.y;
  const int64_t hoff = (int64_t)head * D;
  const uint32_t lds0 = fa_lds_addr(dm_smem);

  // one-time LDS init: the constant region
  for (int i = tid; i < (DM_SMEM_BYTES - DM_CK) / 2; i += NT) {
    const int vb = 2 * i - (DM_CV - DM_CK);      // byte offset inside the V constant region
    const bool one = (i == 0) || vb == 0 || vb == 80 || vb == 1280 || vb == 1360;
    reinterpret_cast<uint16_t*>(dm_smem + DM_CK)[i] = one ? ONE16 : (uint16_t)0;
  }

  // ---- Q^T fragments (pre-scaled by scale * log2 e); slot 40 (lanes of half 1, fragment 2, word 0) carries -offset
  u32x4_t qf[QT][KS];
#pragma unroll
  for (int qs = 0; qs < QT; ++qs) {
    const int q_idx = qt * BQ + wid * 32 * QT + qs * 32 + l31;
    const int64_t q_row = map_row(p.qm, grp, q_idx < p.q_len ? q_idx : p.q_len - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = 16 * ks + 8 * g;
      if (d0 < D) {
        u32x4_t wq = *reinterpret_cast<const u32x4_t*>(p.Q + q_row * p.qm.ld + hoff + d0);
#pragma unroll
        for (int j = 0; j < 4; ++j) wq[j] = pack16(lo16(wq[j]) * p.scale_log2, hi16(wq[j]) * p.scale_log2);
        qf[qs][ks] = wq;
      } else {
        qf[qs][ks] = u32x4_t{0u, 0u, 0u, 0u};
      }
    }
  }

  // ---- DMA lanes.  Chunk slot s of a tile buffer (16 B at byte 16 s): s < 320 is K row s / 5, piece s % 5; s >= 320 is V
  // physical row (s - 320) / 5, which holds key 16 (r / 16) + 4 (r & 3) + ((r >> 2) & 3).  Wave w issues slots 64 w .. + 63
  // (instruction A: K for w < 5, V otherwise) and 512 + 16 w .. + 15 (instruction B, 16 lanes: V).
  const int64_t ld = p.km.ld;
  const int64_t kgbase = map_group_base(p.km, grp);
  const uint32_t seg_len = (uint32_t)p.km.seg_len;
  const int64_t tile_step = (int64_t)64 * ld;
  const int64_t wrap_step = (p.km.seg_stride - p.km.seg_len) * ld;
  auto slot_src = [&](int slot) -> uint32_t {
    const bool isk = slot < 320;
    const int s2 = isk ? slot : slot - 320;
    const int prow = s2 / 5, c = s2 % 5;
    int key = prow;
    if (!isk) {
      if constexpr (PV16) {     // row = 16 half + 8 b4 + 2 j + rr  holds key 16 b4 + 8 half + 4 rr + j of its 32-key sub-tile
        const int pr = prow & 31;
        key = (prow & ~31) | (16 * ((pr >> 3) & 1) + 8 * (pr >> 4) + 4 * (pr & 1) + ((pr >> 1) & 3));
      } else {
        key = (prow & ~15) | (4 * (prow & 3) + ((prow >> 2) & 3));
      }
    }
    return (uint32_t)(((int64_t)key * ld + c * 8) * 2);
  };
  const uint32_t voffA = slot_src(w < 10 ? 64 * w + lane : lane);
  const uint32_t voffB = slot_src(QT == 2 ? 512 + 16 * w + i16 : 0);
  const uint64_t maskA = w < 10 ? ~0ull : 0ull;
  // sample sub-tile (DM_SAMPLED): slot s = 64 w + lane < 160 is piece s % 5 of sample row s / 5 = key (s / 5) * (kv_len / 32)
  uint32_t voffS = 0;
  if constexpr (DM_SAMPLED) {
    const int ss = (64 * w + lane) % 160;
    const int64_t key = (int64_t)(ss / 5) * (p.kv_len / 32);
    voffS = (uint32_t)((map_seq(p.km, key) * ld + (ss % 5) * 8) * 2);
  }
  const uint64_t maskS = w < 2 ? ~0ull : (w == 2 ? 0xffffffffull : 0ull);

  // ---- fragment addressing (byte offsets into dm_smem; the tile / sub-tile offset is added per step, scaled by 0 for the
  // lanes that read constants)
  const uint32_t klane = (uint32_t)(kperm(l31) * DM_ROWB + 16 * g);             // K fragments 0, 1: + 32 ks
  const uint32_t klane2 = g ? (uint32_t)DM_CK : klane + 64u;                    // K fragment 2: dims 32..39 | constant chunk
  uint32_t kmul = g ? 0u : 1u;
  asm volatile("" : "+v"(kmul));      // opaque: keeps  lane + mul * offset  one v_mad_u32_u24 (the compiler otherwise builds mov + cndmask + add)
  const int c4 = i16 & 3;
  // 32x32 PV: a 16-lane group (q4) reads keys 8 (q4 >> 1) + 4 rr + (i16 >> 2) of a 16-key half, dims 16 (q4 & 1) + 4 c4 of a 32-row tile;
  //           rows are stored 4x4-transposed inside every 16-key group.
  // 16x16 PV: a group reads keys [0, 16, 8, 24][q4] + 4 rr + (i16 >> 2), dims 4 c4 of a 16-row tile; row = 16 (q4 >> 1) + 8 (q4 & 1) + 2 (i16 >> 2) + rr
  //           (the eight keys of a 32-lane half land on eight rows of one parity: 8-dword windows on all 64 banks).
  const uint32_t vrow = PV16 ? (uint32_t)((16 * (q4 >> 1) + 8 * (q4 & 1) + 2 * (i16 >> 2)) * DM_ROWB)
                             : (uint32_t)((4 * (i16 >> 2) + 2 * (q4 >> 1)) * DM_ROWB);
  const uint32_t vlane0 = PV16 ? vrow + (uint32_t)(8 * c4) : vrow + (uint32_t)(2 * (16 * (q4 & 1) + 4 * c4));      // O^T rows 0..31: all data
  const bool v1_data = PV16 ? c4 < 2 : ((q4 & 1) == 0 && c4 < 2);               // O^T rows 32..: dims 32..39 | ones | zeros
  const bool v1_one = PV16 ? c4 == 2 : ((q4 & 1) == 0 && c4 == 2);
  const uint32_t vlane1 = v1_data ? vrow + (uint32_t)(2 * (32 + 4 * c4)) : (v1_one ? (uint32_t)DM_CV : (uint32_t)(DM_CV + 8));
  uint32_t vmul = v1_data ? 1u : 0u;
  asm volatile("" : "+v"(vmul));

  if constexpr ((FLAGS & 2) != 0) {
    if (w >= 4) __builtin_amdgcn_s_setprio(1);
  }

  // ---- pieces shared by the two passes
  const uint16_t* gA = nullptr;
  const uint16_t* gB = nullptr;
  uint32_t seg_off = 0;
  auto dma_reset = [&]() __attribute__((always_inline)) {
    gA = dm_scalar((w < 5 ? p.K : p.V) + hoff + kgbase * ld);
    gB = dm_scalar(p.V + hoff + kgbase * ld);
    seg_off = 0;
  };
  auto dma_advance = [&]() __attribute__((always_inline)) {
    seg_off += 64;
    int64_t stp = tile_step;
    if (seg_off >= seg_len) { stp += wrap_step; seg_off = 0; }
    gA += stp; gB += stp;
  };
  auto dma_a = [&](int tile) __attribute__((always_inline)) {
    if constexpr (QT == 2) {
      dm_glds16(voffA, gA, lds0 + (uint32_t)((tile & (DM_RING - 1)) * DM_TILEB + 1024 * w));
    } else {
      dm_glds16_m(voffA, gA, lds0 + (uint32_t)((tile & (DM_RING - 1)) * DM_TILEB + 1024 * (w < 10 ? w : 0)), maskA);
      dma_advance();
    }
  };
  auto dma_b = [&](int tile) __attribute__((always_inline)) {      // second instruction of a tile (QT = 2); then the bases move on
    if constexpr (QT == 2) {
      dm_glds16_q(voffB, gB, lds0 + (uint32_t)((tile & (DM_RING - 1)) * DM_TILEB + 8192 + 256 * w));
      dma_advance();
    }
  };
  f32x16_t oacc[QT][MT];       // 32x32 PV: O^T tiles [query sub-tile][32 rows]
  f32x4_t oacc16[3][2 * QT];   // 16x16 PV: O^T tiles [16 rows][16 queries]; rows 4 q4 + r, query 16 nb + i16
  u32x4_t kf[KS];
  u32x4_t vf[MT][2];           // V^T fragments of the sub-tile whose P is multiplied next (32x32 PV: [row tile][key half])
  u32x4_t vf16[3];             // (16x16 PV: [row tile], 32 keys each)
  auto read_k = [&](uint32_t koff) __attribute__((always_inline)) {          // K fragments of the sub-tile at byte offset koff
    const uint32_t a = klane + koff, a2 = __umul24(kmul, koff) + klane2;
    kf[0] = dm_lds128(a); kf[1] = dm_lds128(a + 32); kf[2] = dm_lds128(a2);
  };
  // V^T fragment half i of the sub-tile at a0 / a1 (a1: the row tile that holds the constants)
  auto read_vhalf = [&](auto i_c, uint32_t a0, uint32_t a1) __attribute__((always_inline)) {
    constexpr int i = decltype(i_c)::value;
    if constexpr (PV16) {
      constexpr int mb = i / 2, rr = i % 2;
      const u32x2_t t = dm_ldstr((mb == 2 ? a1 : a0 + 32 * mb) + rr * DM_ROWB);
      vf16[mb][2 * rr] = t[0];
      vf16[mb][2 * rr + 1] = t[1];
    } else {
      constexpr int mt = i / 4, h = (i / 2) % 2, rr = i % 2;
      const u32x2_t t = dm_ldstr((mt ? a1 : a0) + (16 * h + rr) * DM_ROWB);
      vf[mt][h][2 * rr] = t[0];
      vf[mt][h][2 * rr + 1] = t[1];
    }
  };
  constexpr int NVH = PV16 ? 6 : 8;          // fragment halves per sub-tile
  auto read_v = [&](uint32_t voff) __attribute__((always_inline)) {
    const uint32_t a0 = vlane0 + voff, a1 = __umul24(vmul, voff) + vlane1;
    static_for<NVH>([&](auto i_c) __attribute__((always_inline)) { read_vhalf(i_c, a0, a1); });
  };
  // PV MFMA instruction i of a sub-tile (32x32: 8 instructions of 32 cycles; 16x16: 12 of 16 cycles); P as packed by finish_p
  constexpr int NPVI = PV16 ? 6 * QT : 4 * QT;
  auto pv_mfma = [&](auto i_c, u32x4_t (&P)[QT][2]) __attribute__((always_inline)) {
    constexpr int i = decltype(i_c)::value;
    if constexpr (PV16) {
      constexpr int mb = i / (2 * QT), nb = i % (2 * QT);
      oacc16[mb][nb] = mfma16(vf16[mb], P[nb / 2][nb % 2], oacc16[mb][nb]);
    } else {
      constexpr int mt = i / (2 * QT), h = (i / QT) % 2, qs = i % QT;
      oacc[qs][mt] = mfma32(vf[mt][h], P[qs][h], oacc[qs][mt]);
    }
  };
  // 16x16 PV: turn the packed probabilities of query sub-tile qs, pair jj (P[qs][h][jj] = keys 16 h + 8 g + 2 jj, + 1 of query l31) into the
  // B operands of query blocks 2 qs (in P[qs][0]) and 2 qs + 1 (in P[qs][1]): 16-lane row q4 then holds keys [0, 16, 8, 24][q4] + 2 jj, + 1
  auto swap_p = [&](auto k_c, u32x4_t (&P)[QT][2]) __attribute__((always_inline)) {
    constexpr int k = decltype(k_c)::value, qs = k / 4, jj = k % 4;
    const auto r = __builtin_amdgcn_permlane16_swap(P[qs][0][jj], P[qs][1][jj], false, false);
    P[qs][0][jj] = r[0];
    P[qs][1][jj] = r[1];
  };
  auto clear_o = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int qs = 0; qs < QT; ++qs) {
      if (g == G_PAD) qf[qs][KS_PAD][0] = 0u;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[qs][mt][r] = 0.f;
    }
#pragma unroll
    for (int mb = 0; mb < 3; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2 * QT; ++nb) oacc16[mb][nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  };
  const int nt = p.kv_len / 64;               // launcher guarantees kv_len % 64 == 0, nt >= 4, aligned segments
  [[maybe_unused]] const float cmax_sds = f16_expected_max_sds(p.kv_len);
  auto prologue_dma = [&]() __attribute__((always_inline)) {      // (sample sub-tile,) tiles 0, 1, 2 requested; all but tile 2 complete
    dma_reset();
    if constexpr (DM_SAMPLED && TRY_NOMAX)
      dm_glds16_m(voffS, dm_scalar(p.K + hoff + kgbase * ld), lds0 + (uint32_t)(DM_SAMPLE + 1024 * (w < 3 ? w : 0)), maskS);
    dma_a(0); dma_b(0);
    dma_a(1); dma_b(1);
    dma_a(2); dma_b(2);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "i"(NDMA) : "memory");
  };
  // first offset of a query: exact maximum of its scores against the 32-key sub-tile at koff (+ bias); leaves the re-based scores in s
  // (fp16 max-free pass: also returns whether the spread of the sample scores predicts an overflow of fp16's window)
  auto first_scores = [&](f32x16_t (&s)[QT], float (&m_off)[QT], float bias, uint32_t koff = 0u) __attribute__((always_inline)) -> bool {
    read_k(koff);
    bool wide = false;
#pragma unroll
    for (int qs = 0; qs < QT; ++qs) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[qs][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s[qs] = mfma32(kf[ks], qf[qs][ks], s[qs]);
      float mx = s[qs][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[qs][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float b = bias;
      if constexpr (DM_SAMPLED) {
        if (bias != 0.f) {                 // (the exact pass starts from the plain maximum)
          float sm = 0.f, sq = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) { sm += s[qs][r]; sq = fmaf(s[qs][r], s[qs][r], sq); }
          sm += __shfl_xor(sm, 32); sq += __shfl_xor(sq, 32);
          const float mean = sm * (1.f / 32.f);
          bool wq;
          b = f16_sampled_bias(mx, mean, sq * (1.f / 32.f) - mean * mean, cmax_sds, wq);
          wide = wide || wq;
        }
      }
      const float new_off = round16(mx + b);
      m_off[qs] = new_off;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[qs][r] -= new_off;
      if (g == G_PAD) qf[qs][KS_PAD][0] = pack16(-new_off, 0.f);
    }
    return wide;
  };
  // row sums (O^T row 40): 32x32 layout: register 4 of the second row tile, half 0 (query l31); 16x16 layout: register 0 of row
  // tile 2 in lanes 32..47 (query i16 of block nb).  Returns 1 / sum scaled for the store; `bad` = sum unusable (max-free pass).
  constexpr int NINV = PV16 ? 2 * QT : QT;
  float l_row[NINV];
  auto row_sums = [&](float (&inv)[NINV]) __attribute__((always_inline)) -> bool {
    bool bad = false;
#pragma unroll
    for (int i = 0; i < NINV; ++i) {
      float l_tot;
      if constexpr (PV16) l_tot = __shfl(oacc16[2][i][0], 32 + i16);
      else l_tot = __shfl(oacc[i][1][4], l31);
      bad = bad || !(l_tot < DM_L_BAD) || !(l_tot > 0.f);
      inv[i] = p.out_scale / l_tot;
      l_row[i] = l_tot;
    }
    return bad;
  };
  // training: log2 of the softmax denominator per query = offset + log2(row sum); the offset of query l31 of sub-tile qs is what the
  // lanes of half G_PAD carry (negated, 16 bits) in contraction slot 40 of their Q fragment
  auto store_lse = [&]() __attribute__((always_inline)) {
    if (p.lse == nullptr) return;
#pragma unroll
    for (int i = 0; i < NINV; ++i) {
      const int qs = PV16 ? i / 2 : i, ql = PV16 ? 16 * (i % 2) + i16 : l31;
      const float noff = lo16((uint32_t)__shfl((int)qf[qs][KS_PAD][0], 32 * G_PAD + ql));
      const int q_idx = qt * BQ + wid * 32 * QT + 32 * qs + ql;
      if (q_idx < p.q_len && (PV16 ? q4 == 0 : g == 0))
        p.lse[((int64_t)grp * p.heads + head) * p.q_len + q_idx] = __builtin_amdgcn_logf(l_row[i]) - noff;
    }
  };
  auto store4 = [&](uint16_t* dst, float v0, float v1, float v2, float v3) __attribute__((always_inline)) {
    if (p.accumulate) {
      const u32x2_t prev = *reinterpret_cast<const u32x2_t*>(dst);
      v0 += lo16(prev[0]); v1 += hi16(prev[0]); v2 += lo16(prev[1]); v3 += hi16(prev[1]);
    }
    u32x2_t o;
    o[0] = pack16(v0, v1);
    o[1] = pack16(v2, v3);
    *reinterpret_cast<u32x2_t*>(dst) = o;
  };
  auto store_out = [&](const float (&inv)[NINV]) __attribute__((always_inline)) {
    if constexpr (PV16) {      // lane holds O[q = 16 nb + i16][d = 16 mb + 4 q4 + r]
#pragma unroll
      for (int nb = 0; nb < 2 * QT; ++nb) {
        const int q_idx = qt * BQ + wid * 32 * QT + 16 * nb + i16;
        if (q_idx < p.q_len) {
          uint16_t* orow = p.O + map_row(p.om, grp, q_idx) * p.om.ld + hoff;
#pragma unroll
          for (int mb = 0; mb < 3; ++mb) {
            const int d = 16 * mb + 4 * q4;
            if (d < D) store4(orow + d, oacc16[mb][nb][0] * inv[nb], oacc16[mb][nb][1] * inv[nb], oacc16[mb][nb][2] * inv[nb], oacc16[mb][nb][3] * inv[nb]);
          }
        }
      }
    } else {                   // lane holds O[q = l31][d = 32*mt + 8*qd + 4*g + j] for each query sub-tile
#pragma unroll
      for (int qs = 0; qs < QT; ++qs) {
        const int q_idx = qt * BQ + wid * 32 * QT + qs * 32 + l31;
        if (q_idx < p.q_len) {
          uint16_t* orow = p.O + map_row(p.om, grp, q_idx) * p.om.ld + hoff;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              const int d = 32 * mt + 8 * qd + 4 * g;
              if (d < D) store4(orow + d, oacc[qs][mt][4 * qd] * inv[qs], oacc[qs][mt][4 * qd + 1] * inv[qs], oacc[qs][mt][4 * qd + 2] * inv[qs], oacc[qs][mt][4 * qd + 3] * inv[qs]);
            }
        }
      }
    }
  };

  // ================================================================================================================
  // Max-free pass (bf16 storage): branch-free software pipeline.  Returns false when the result must be discarded.
  // ================================================================================================================
  auto run_fast = [&]() __attribute__((always_inline)) -> bool {
    clear_o();
    f32x16_t sA[QT], sB[QT];
    u32x4_t pA[QT][2], pB[QT][2];
#pragma unroll
    for (int qs = 0; qs < QT; ++qs)
#pragma unroll
      for (int h = 0; h < 2; ++h) { pA[qs][h] = u32x4_t{0u, 0u, 0u, 0u}; pB[qs][h] = u32x4_t{0u, 0u, 0u, 0u}; }

    // One pipeline step j: the matrix pipe does O += V^T(j-1)·P(j-1) (8 MFMAs, fragments vf read during step j-1), then
    // S(j+1) = K(j+1)·Q^T (6 MFMAs, fragments read in slot 3); the VALU turns S(j) into P(j) over all 14 slots.  With the PV
    // block first, S(j+1) is born when half of S(j) is already dead: 48 instead of 64 score registers (the QK-first order
    // of flash_attn_il_kernel does not fit 256 registers here without spilling, and a spill reload is a vmcnt(0) that drains
    // the DMA queue).  kOff / vOff: LDS byte offsets of K sub-tile j+1 and V sub-tile j.
    auto step = [&](auto do_qk_c, auto do_pv_c, f32x16_t (&sCur)[QT], f32x16_t (&sNext)[QT], u32x4_t (&pCur)[QT][2],
                    u32x4_t (&pPrev)[QT][2], uint32_t kOff, uint32_t vOff, auto&& hook) __attribute__((always_inline)) {
      constexpr bool DO_QK = decltype(do_qk_c)::value, DO_PV = decltype(do_pv_c)::value;
      constexpr int PPS = PV16 ? 2 : 1;                                   // PV instructions per slot (a slot = 32 matrix-pipe cycles)
      constexpr int NPV = DO_PV ? NPVI / PPS : 0, NQK = DO_QK ? KS * QT : 0, NS = NPV + NQK;
      constexpr int KSLOT = NPV >= 4 ? 3 : 0, VS0 = NPV + 1;              // K fragments early in the PV block; V halves VPS per slot from VS0
      constexpr int VPS = DO_QK ? (NVH + NQK - 2) / (NQK - 1) : 1;
      const uint32_t va0 = vlane0 + vOff, va1 = __umul24(vmul, vOff) + vlane1;
      float e[NEXP];
      auto do_cvt = [&](auto c_c) __attribute__((always_inline)) {
        constexpr int c = decltype(c_c)::value;
        constexpr int qs = c / 8, h = (c / 4) % 2, jj = c % 4;
        pCur[qs][h][jj] = pack16(e[2 * c], e[2 * c + 1]);
      };
      // conversions complete before slot s ends: pairs [0, cdone(s)); a swap needs pair qs*8 + 4 + jj and waits one more slot (the
      // permlane reads two wait states after a VALU write)
      auto cdone = [](int s) constexpr { return s < 0 ? 0 : (NEXP * s / NS) / 2; };
      __builtin_amdgcn_sched_barrier(0);
      static_for<NS>([&](auto s_c) __attribute__((always_inline)) {
        constexpr int s = decltype(s_c)::value;
        if constexpr (s < NPV) {
          static_for<PPS>([&](auto i_c) __attribute__((always_inline)) { pv_mfma(std::integral_constant<int, PPS * s + decltype(i_c)::value>{}, pPrev); });
        } else {
          constexpr int i = s - NPV, ks = i / QT, qs = i % QT;
          if constexpr (ks == 0) {
            f32x16_t z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            sNext[qs] = mfma32(kf[ks], qf[qs][ks], z);
          } else {
            sNext[qs] = mfma32(kf[ks], qf[qs][ks], sNext[qs]);
          }
        }
        if constexpr (DO_QK && DO_PV && s == KSLOT) read_k(kOff);
        if constexpr (DO_QK && s >= VS0 && VPS * (s - VS0) < NVH) {
          static_for<VPS>([&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = VPS * (s - VS0) + decltype(i_c)::value;
            if constexpr (i < NVH) read_vhalf(std::integral_constant<int, i>{}, va0, va1);
          });
        }
        hook(s_c);
        constexpr int E0 = NEXP * s / NS, E1 = NEXP * (s + 1) / NS;
        static_for<E1 - E0>([&](auto x_c) __attribute__((always_inline)) {
          constexpr int x = E0 + decltype(x_c)::value;
          e[x] = __builtin_amdgcn_exp2f(sCur[x / 16][x % 16]);
        });
        constexpr int C0 = cdone(s - 1), C1 = cdone(s);
        static_for<C1 - C0>([&](auto c_c) __attribute__((always_inline)) { do_cvt(std::integral_constant<int, C0 + decltype(c_c)::value>{}); });
        if constexpr (PV16) {
          static_for<4 * QT>([&](auto k_c) __attribute__((always_inline)) {
            constexpr int k = decltype(k_c)::value, need = (k / 4) * 8 + 4 + (k % 4);
            if constexpr (need < cdone(s - 1) && !(need < cdone(s - 2))) swap_p(k_c, pCur);
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      {
        constexpr int C0 = cdone(NS - 1);
        static_for<NCVT - C0>([&](auto c_c) __attribute__((always_inline)) { do_cvt(std::integral_constant<int, C0 + decltype(c_c)::value>{}); });
        if constexpr (PV16) {
          static_for<4 * QT>([&](auto k_c) __attribute__((always_inline)) {
            constexpr int k = decltype(k_c)::value, need = (k / 4) * 8 + 4 + (k % 4);
            if constexpr (!(need < cdone(NS - 2))) swap_p(k_c, pCur);
          });
        }
      }
    };

    prologue_dma();
    {
      float m_off[QT];
      if constexpr (DM_SAMPLED) {
        const bool wide = first_scores(sA, m_off, DM_BIAS, (uint32_t)DM_SAMPLE);      // offset from the sample keys (now in Q's pad slot)
        // a VOTE, not an OR (round 5): the variance of 32 samples scatters by +-25 %, so at a true spread well inside the window (score sd 3:
        // variance 19 of 28) 2-3 % of the queries still read above the threshold and an OR over the workgroup's 512 queries sent EVERY workgroup
        // to the 20 % slower exact pass (profiles/r5_flash_score_spread.log).  The workgroup goes exact right away when more than a quarter of
        // its queries predict an overflow; a row that does overflow in the max-free pass is still caught by its row sum at the end.
        if (__syncthreads_count(wide ? 1 : 0) * 4 > NT) { dm_count(p, 0); return false; }
        read_k(0u);                                                 // ... then S(0) of keys 0..31 under it
#pragma unroll
        for (int qs = 0; qs < QT; ++qs) {
#pragma unroll
          for (int r = 0; r < 16; ++r) sA[qs][r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) sA[qs] = mfma32(kf[ks], qf[qs][ks], sA[qs]);
        }
      } else {
        first_scores(sA, m_off, DM_BIAS);
      }
      read_k((uint32_t)DM_UNITB);               // K(0) keys 32..63 for step 0
    }

    auto iteration = [&](int t, auto first_c, auto last_c, auto dma_c) __attribute__((always_inline)) {
      constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value, DMA = decltype(dma_c)::value;
      auto even_hook = [&](auto s_c) __attribute__((always_inline)) {
        if constexpr (DMA && decltype(s_c)::value == 2) dma_a(t + 3);
      };
      auto odd_hook = [&](auto s_c) __attribute__((always_inline)) {
        if constexpr (DMA && decltype(s_c)::value == 2) dma_b(t + 3);
      };
      const uint32_t tb = (uint32_t)((t & (DM_RING - 1)) * DM_TILEB);
      const uint32_t tn = (uint32_t)(((t + 1) & (DM_RING - 1)) * DM_TILEB);
      // even step j = 2t:  O += V(t-1)[32..63] P(2t-1), S(2t+1) from K(t) keys 32..63, P(2t) from S(2t); reads V(t)[0..31]
      step(std::true_type{}, std::integral_constant<bool, !FIRST>{}, sA, sB, pA, pB, tb + DM_UNITB, tb + DM_VOFF, even_hook);
      // odd step j = 2t+1: O += V(t)[0..31] P(2t), S(2t+2) from K(t+1) keys 0..31, P(2t+1) from S(2t+1); reads V(t)[32..63]
      step(std::integral_constant<bool, !LAST>{}, std::true_type{}, sB, sA, pB, pA, tn, tb + DM_VOFF + DM_UNITB, odd_hook);
      if constexpr (!LAST) {
        // tile t+2 (requested one iteration ago) must be complete for everybody; tile t+3's two requests may stay in flight.
        // LDS reads stay in flight too: no buffer is re-used within four iterations.
#ifdef A3D_ABLATIONS
        if constexpr ((FLAGS & 16) != 0) { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(NDMA) : "memory"); } else      // timing ablation: no barrier (results may be wrong)
#endif
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "i"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      }
    };
    constexpr std::true_type Y{};
    constexpr std::false_type N{};
    iteration(0, Y, N, Y);
    for (int t = 1; t < nt - 3; ++t) iteration(t, N, N, Y);
    iteration(nt - 3, N, N, N);
    iteration(nt - 2, N, N, N);
    iteration(nt - 1, N, Y, N);
    {   // O += V(nt-1)[32..63] P(2nt-1)
      read_v((uint32_t)(((nt - 1) & (DM_RING - 1)) * DM_TILEB + DM_VOFF + DM_UNITB));
      static_for<NPVI>([&](auto i_c) __attribute__((always_inline)) { pv_mfma(i_c, pB); });
    }
    float inv[NINV];
    const bool bad = row_sums(inv);
    if (__syncthreads_or(bad ? 1 : 0)) { dm_count(p, 1); return false; }          // (also: every wave is done with the LDS images)
    store_out(inv);
    store_lse();
    return true;
  };

  // ================================================================================================================
  // Exact pass: running maximum per 32-key sub-tile, lazy by LAZY_THR, un-pipelined (the arithmetic of flash_attn_kernel).
  // Only runs after an overflow of the max-free pass (or always, in builds / flag sets without it).
  // ================================================================================================================
  auto run_exact = [&]() __attribute__((always_inline)) {
    clear_o();
    float m_off[QT];
    f32x16_t sc[QT];
    prologue_dma();
    first_scores(sc, m_off, 0.f);
    for (int t = 0; t < nt; ++t) {
      const bool more = t + 3 < nt;
      if (more) { dma_a(t + 3); dma_b(t + 3); }
      const uint32_t tb = (uint32_t)((t & (DM_RING - 1)) * DM_TILEB);
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        if (t > 0 || sub > 0) {
          read_k(tb + sub * DM_UNITB);
#pragma unroll
          for (int qs = 0; qs < QT; ++qs) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[qs][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) sc[qs] = mfma32(kf[ks], qf[qs][ks], sc[qs]);
            float mx = sc[qs][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[qs][r]);
            if (__any(mx > LAZY_THR)) {
              const float delta0 = fmaxf(fmaxf(mx, __shfl_xor(mx, 32)), 0.f);
              const float new_off = round16(m_off[qs] + delta0);
              const float delta = new_off - m_off[qs];
              m_off[qs] = new_off;
              const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
              for (int r = 0; r < 16; ++r) sc[qs][r] -= delta;
              if constexpr (PV16) {          // the query of O block 2 qs + a, lane i16 is the score tile's lane 16 a + i16
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                  const float alpha_o = __shfl(alpha, 16 * a + i16);
#pragma unroll
                  for (int mb = 0; mb < 3; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) oacc16[mb][2 * qs + a][r] *= alpha_o;
                }
              } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                  for (int r = 0; r < 16; ++r) oacc[qs][mt][r] *= alpha;
              }
              if (g == G_PAD) qf[qs][KS_PAD][0] = pack16(-new_off, 0.f);
            }
          }
        }
        u32x4_t pf[QT][2];
#pragma unroll
        for (int qs = 0; qs < QT; ++qs)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
              pf[qs][h][jj] = pack16(__builtin_amdgcn_exp2f(sc[qs][8 * h + 2 * jj]), __builtin_amdgcn_exp2f(sc[qs][8 * h + 2 * jj + 1]));
        if constexpr (PV16) static_for<4 * QT>([&](auto k_c) __attribute__((always_inline)) { swap_p(k_c, pf); });
        read_v(tb + DM_VOFF + sub * DM_UNITB);
        static_for<NPVI>([&](auto i_c) __attribute__((always_inline)) { pv_mfma(i_c, pf); });
      }
      if (t + 1 < nt) {
        if (more) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "i"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
    float inv[NINV];
    row_sums(inv);
    store_out(inv);
    store_lse();
  };

  __syncthreads();        // constant region written
  dm_count(p, 2);
  if constexpr (TRY_NOMAX) {
    if (!run_fast()) run_exact();
  } else {
    run_exact();
  }
}

template <int FLAGS, int QT>
int launch_dm(int groups, hipStream_t s, const AttnParams& p) {
  static uint64_t attr_done = 0;
  if (int rc = a3d_once_per_device(attr_done, [] {
        return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_attn_dm_kernel<FLAGS, QT>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, DM_SMEM_BYTES); })) return rc;
  const int q_tiles = (p.q_len + 511) / 512;
  flash_attn_dm_kernel<FLAGS, QT><<<dim3((unsigned)(p.heads * q_tiles), (unsigned)groups), dim3(1024 / QT), DM_SMEM_BYTES, s>>>(p);
  return a3d_launch_status();
}

}  // namespace

// flags: 5 = max-free first pass + exact re-run on overflow (the default), 4 = exact pass only; both with P·V through the 16x16x32 MFMA and
// 8 waves x 64 queries (the other flag sets of flash_attn_dm_kernel were round-3 A/B variants: profiles/README.md).  Shapes: head_dim 40,
// kv_len % 64 == 0, kv_len >= 256, aligned segments (checked by the caller).
int A3D_FN(a3d_launch_flash_dm)(int flags, int groups, hipStream_t s, const AttnParams& p) {
  switch (flags) {
    case 4: return launch_dm<4, 2>(groups, s, p);
    case 5: return launch_dm<5, 2>(groups, s, p);
    default: return A3D_EINVAL;
  }
}
