// Level-2 / level-3 multi-view and first-frame attention (head_dim 160, aligned K/V from 256 keys, any number of queries): the LDS-DMA design of
// flash_attn_dm80.hip — global_load_lds staging, PV-first software pipeline, max-free 16-bit softmax with an exact re-run — at head_dim 160,
// on EIGHT waves (two per SIMD, 256 registers each).  Replaces xformers.ops.memory_efficient_attention at attention_processor.py:405, 416,
// 656 for the 1 024-key level-2 shapes, which ran on the generic register-staged kernel (flash_attn.hip) at 0.22 of the matrix peak for six
// rounds; round 5's port to FOUR waves with up to 512 registers each (profiles/r5_flash_dw160_one_wave_per_simd.patch) measured equal to it:
// one wave per SIMD cannot hide its own LDS / VALU work under its MFMAs.
//
// What makes the working set of a wave (32 queries) fit 256 registers at this head size — O^T 5 x 16, two score tiles 2 x 16, P 2 x 8,
// Q^T 40 — although K (40) and V^T (40) fragments of a 32-key sub-tile alone would take it to ~280:
//  * K and V^T fragments are read from LDS ONE PIPELINE SLOT AT A TIME, five slots ahead of the MFMA that consumes them (a slot = one
//    32 x 32 x 16 MFMA = 32 matrix-pipe cycles): never more than ten 4-register fragments of the two kinds together are live;
//  * no ones-row for the row sums (a sixth O^T tile: 16 registers and 20 % more P·V work): the softmax denominators are summed on the VALU,
//    which is idle at this head size (56 plain instructions per 20 MFMAs); overflow / underflow of the 16-bit P is then checked on the
//    finished accumulators (fp16: a P = inf turns every dimension of its row non-finite) and on the sum;
//  * the softmax offset is subtracted on the VALU (v_sub in front of v_exp) instead of a 16-register C operand: the contraction (160 = 10 x 16)
//    has no spare slot.
// A 64-key tile is 40 KB (K 64 x 320 B, V the same): a ring of THREE tiles; tile t + 2 is requested at the start of iteration t — every wave
// issues five full LDS-DMA instructions — and must have landed when the iteration ends (vmcnt(0) + the one barrier per tile): an iteration is
// 40 MFMAs per wave = 2 560 matrix-pipe cycles per SIMD, more than an HBM round trip.
// 320-byte rows = 80 dwords = 16 banks: K rows r and r + 4 share banks, so the 16-byte chunk index of a K row is XORed with (row >> 2) & 3 on
// the DMA source address and again on the fragment address (conflict-free 16-lane groups of ds_read_b128); V rows stay in natural order —
// the four keys one ds_read_b64_tr_b16 group touches are four consecutive rows = 16 banks apart.
#include "flash_common.h"

namespace {

constexpr int F_D = 160, F_KS = 10, F_MT = 5;
constexpr int F_ROWB = 2 * F_D;                    // bytes per K / V row in LDS
constexpr int F_CPR = F_D / 8;                     // 16-byte chunks per row
constexpr int F_UNITB = 32 * F_ROWB;               // one 32-key sub-tile of K or of V
constexpr int F_KB = 2 * F_UNITB;                  // K (or V) image of a 64-key tile
constexpr int F_TILEB = 2 * F_KB;                  // [K keys 0..63 (chunks swizzled) | V keys 0..63]
constexpr int F_RING = 3;
constexpr int F_NDMA = 5;                          // LDS-DMA instructions per wave and tile: 2 560 chunks / 64 lanes / 8 waves
constexpr int F_SAMPLE = F_RING * F_TILEB;         // one 32-key K sub-tile of sample keys (fp16 storage: the offset estimate)
constexpr float F_L_HI = 1.2676506e30f;            // 2^100: beyond this the max-free result is not trusted
#ifdef A3D_STORAGE_F16
constexpr float F_BIAS = F16_BIAS;
constexpr bool F_SAMPLED = true;
constexpr float F_L_LO = 1.220703125e-4f;          // 2^-13: a row sum below it has no normal fp16 P at all
constexpr int F_VOFF = F_SAMPLE + F_UNITB;
#else
constexpr float F_BIAS = 40.f;
constexpr bool F_SAMPLED = false;
constexpr float F_L_LO = 7.8886091e-31f;           // 2^-100
constexpr int F_VOFF = F_SAMPLE;
#endif
// The five DMA source offsets of a lane live in LDS after the prologue ([instruction][thread], 10 KB), not in registers: the key loop runs at ~250
// of its 256 registers, and a spilled offset comes back through scratch_load + s_waitcnt vmcnt(0) — i.e. behind the LDS-DMA pieces issued just
// before it, a full memory round trip inside every iteration (measured: 1 097 -> 1 381 us at 4 096 keys when two of them spilled).  An LDS read
// one pipeline slot ahead of its DMA instruction costs nothing.
constexpr int F_SMEM_BYTES = F_VOFF + F_NDMA * 512 * 4;
A3D_DEV int f_kswz(int row) { return (row >> 2) & 3; }      // K chunk swizzle of a row

extern __shared__ __attribute__((aligned(16))) uint8_t f_smem[];
A3D_DEV u32x4_t f_lds128(uint32_t off) { return *reinterpret_cast<const u32x4_t*>(f_smem + off); }
A3D_DEV u32x2_t f_ldstr(uint32_t off) { return lds_tr16_b64(reinterpret_cast<const uint16_t*>(f_smem + off)); }

// FLAGS: 1 = max-free first pass
template <int FLAGS>
__global__ __launch_bounds__(512, 2) void flash_attn_dm160_kernel(const AttnParams p) {
  constexpr int D = F_D, KS = F_KS, MT = F_MT, NT = 512, BQ = 256;
  constexpr int NEXP = 16, NCVT = 8, NDMA = F_NDMA;
  constexpr bool TRY_NOMAX = (FLAGS & 1) != 0;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int w = __builtin_amdgcn_readfirstlane(wid);
  const int l31 = lane & 31, g = lane >> 5, i16 = lane & 15, q4 = lane >> 4;
  const int head = blockIdx.x % p.heads;
  const int qt = blockIdx.x / p.heads;
  const int64_t grp = blockIdx.y;
  const int64_t hoff = (int64_t)head * D;
  const uint32_t lds0 = fa_lds_addr(f_smem);

  // ---- DMA lanes.  Chunk slot s of a tile buffer (16 B at byte 16 s): s < 1280 is K row s / 20, position s % 20 = chunk ^ ((row >> 2) & 3);
  // s >= 1280 is V row (s - 1280) / 20, chunk (s - 1280) % 20.  Instruction i of wave w covers slots 64 (8 i + w) .. + 63: K for 8 i + w < 20.
  const int64_t ld = p.km.ld;
  const int64_t kgbase = map_group_base(p.km, grp);
  const uint32_t seg_len = (uint32_t)p.km.seg_len;
  const int64_t tile_step = (int64_t)64 * ld;
  const int64_t wrap_step = (p.km.seg_stride - p.km.seg_len) * ld;
  auto slot_src = [&](int slot) -> uint32_t {
    const bool isk = slot < 64 * F_CPR;
    const int s2 = isk ? slot : slot - 64 * F_CPR;
    const int row = s2 / F_CPR, cpos = s2 % F_CPR;
    const int c = isk ? (cpos ^ f_kswz(row)) : cpos;
    return (uint32_t)(((int64_t)row * ld + c * 8) * 2);
  };
  uint32_t voff[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) voff[i] = slot_src(64 * (8 * i + w) + lane);
#pragma unroll
  for (int i = 0; i < NDMA; ++i) *reinterpret_cast<uint32_t*>(f_smem + F_VOFF + (i * 512 + tid) * 4) = voff[i];
  // (the reader rebuilds its lane index from mbcnt instead of keeping a table address live: in the fp16 build that register was the next one spilled)
  auto tab = [&](int I) __attribute__((always_inline)) -> uint32_t {
    const uint32_t ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return *reinterpret_cast<const uint32_t*>(f_smem + (ln * 4u + (uint32_t)(F_VOFF + I * 2048 + w * 256)));
  };
  // sample sub-tile (F_SAMPLED): slot s < 640 is chunk s % 20 (swizzled like every K row) of sample row s / 20 = key (s / 20) * (kv_len / 32);
  // instruction i of wave w covers slots 64 (8 i + w) ..: waves 0..7 one each, waves 0 and 1 a second one
  uint32_t voffS[2] = {0u, 0u};
  if constexpr (F_SAMPLED) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ss = (64 * (8 * i + w) + lane) % (32 * F_CPR), prow = ss / F_CPR;
      const int64_t key = (int64_t)prow * (p.kv_len / 32);
      voffS[i] = (uint32_t)((map_seq(p.km, key) * ld + ((ss % F_CPR) ^ f_kswz(prow)) * 8) * 2);
    }
  }

  // ---- fragment addressing (byte offsets into f_smem; the tile / sub-tile offset is added per step)
  const int krow = kperm(l31);
  // K fragment of contraction step ks: chunk 2 ks + g, swizzled: 4 (ks >> 1) + ((2 (ks & 1) + g) ^ swz) for a 2-bit swizzle
  const uint32_t krowb = (uint32_t)(krow * F_ROWB);
  const uint32_t kc_even = krowb + (uint32_t)(16 * (g ^ f_kswz(krow))), kc_odd = krowb + (uint32_t)(16 * ((2 + g) ^ f_kswz(krow)));
  // V^T fragment (O^T tile mt, key half h): two transposing reads rr of 4 keys each: lane reads dims 32 mt + 16 (q4 & 1) + 4 (i16 & 3) .. + 3
  // of key 16 h + 8 (q4 >> 1) + 4 rr + (i16 >> 2)
  const uint32_t vlane = (uint32_t)((8 * (q4 >> 1) + (i16 >> 2)) * F_ROWB + 2 * (16 * (q4 & 1) + 4 * (i16 & 3)));

  const uint16_t* gK = nullptr;
  const uint16_t* gV = nullptr;
  uint32_t seg_off = 0;
  auto dma_reset = [&]() __attribute__((always_inline)) {
    gK = dm_scalar(p.K + hoff + kgbase * ld);
    gV = dm_scalar(p.V + hoff + kgbase * ld);
    seg_off = 0;
  };
  auto tile_base = [&](int tile) __attribute__((always_inline)) -> uint32_t { return (uint32_t)((tile % F_RING) * F_TILEB); };
  // instruction I of this wave's share of a tile; the last one moves the bases on
  auto dma_i = [&](int tile, auto i_c, uint32_t vo) __attribute__((always_inline)) {
    constexpr int I = decltype(i_c)::value;
    const bool isk = 8 * I + w < F_CPR;               // wave-uniform
    dm_glds16(vo, isk ? gK : gV, lds0 + tile_base(tile) + 1024u * (uint32_t)(8 * I + w));
    if constexpr (I == NDMA - 1) {
      seg_off += 64;
      int64_t stp = tile_step;
      if (seg_off >= seg_len) { stp += wrap_step; seg_off = 0; }
      gK += stp; gV += stp;
    }
  };
  auto dma_tile = [&](int tile) __attribute__((always_inline)) {          // (offsets from the LDS table: written by this thread itself, no barrier needed)
    static_for<NDMA>([&](auto i_c) __attribute__((always_inline)) { dma_i(tile, i_c, tab(decltype(i_c)::value)); });
  };
  // (sample sub-tile,) tiles 0 and 1: requested before anything else of the prologue — the Q rows are fetched under them —, complete for
  // everybody after prologue_wait
  auto prologue_issue = [&]() __attribute__((always_inline)) {
    dma_reset();
    if constexpr (F_SAMPLED && TRY_NOMAX) {
      dm_glds16(voffS[0], dm_scalar(p.K + hoff + kgbase * ld), lds0 + (uint32_t)(F_SAMPLE + 1024 * w));
      dm_glds16_m(voffS[1], dm_scalar(p.K + hoff + kgbase * ld), lds0 + (uint32_t)(F_SAMPLE + 1024 * (8 + (w < 2 ? w : 0))), w < 2 ? ~0ull : 0ull);
    }
    dma_tile(0);
    dma_tile(1);
  };
  auto prologue_wait = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); };
  prologue_issue();

  // ---- Q^T fragments (pre-scaled by scale * log2 e)
  const int q_idx = qt * BQ + wid * 32 + l31;
  u32x4_t qf[KS];
  {
    const int64_t q_row = map_row(p.qm, grp, q_idx < p.q_len ? q_idx : p.q_len - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4_t wq = *reinterpret_cast<const u32x4_t*>(p.Q + q_row * p.qm.ld + hoff + 16 * ks + 8 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j) wq[j] = pack16(lo16(wq[j]) * p.scale_log2, hi16(wq[j]) * p.scale_log2);
      qf[ks] = wq;
    }
  }


  f32x16_t oacc[MT];
  u32x4_t kf[KS];
  u32x4_t vf[MT][2];
  float off = 0.f;               // softmax offset of this lane's query (log2 units)
  float lsum = 0.f;              // this lane's share of the row sum (its 16 keys of every sub-tile)
  // (ke / ko: this lane's even / odd-step chunk address of the sub-tile, made opaque by the caller: the optimiser otherwise hoists one address
  // register per fragment and tile-buffer constant out of the key loop — 60 registers and their spills — instead of folding the constants
  // into the reads' offset fields)
  auto read_k1 = [&](auto ks_c, uint32_t ke, uint32_t ko) __attribute__((always_inline)) {
    constexpr int ks = decltype(ks_c)::value;
    kf[ks] = f_lds128(((ks & 1) ? ko : ke) + 64 * (ks >> 1));
  };
  auto read_k = [&](uint32_t koff) __attribute__((always_inline)) {
    uint32_t ke = kc_even + koff, ko = kc_odd + koff;
    asm volatile("" : "+v"(ke), "+v"(ko));
    static_for<KS>([&](auto ks_c) __attribute__((always_inline)) { read_k1(ks_c, ke, ko); });
  };
  // V^T fragment i of a sub-tile in the order the P·V MFMAs consume them: key half h = i / MT, O^T tile mt = i % MT
  auto read_v1 = [&](auto i_c, uint32_t va) __attribute__((always_inline)) {
    constexpr int i = decltype(i_c)::value, h = i / MT, mt = i % MT;
    const u32x2_t t0 = f_ldstr(va + 64 * mt + (16 * h) * F_ROWB), t1 = f_ldstr(va + 64 * mt + (16 * h + 4) * F_ROWB);
    vf[mt][h] = u32x4_t{t0[0], t0[1], t1[0], t1[1]};
  };
  auto read_v = [&](uint32_t voff_) __attribute__((always_inline)) {
    uint32_t va = vlane + voff_;
    asm volatile("" : "+v"(va));
    static_for<2 * MT>([&](auto i_c) __attribute__((always_inline)) { read_v1(i_c, va); });
  };
  auto pv_mfma = [&](auto i_c, u32x4_t (&P)[2]) __attribute__((always_inline)) {
    constexpr int i = decltype(i_c)::value, h = i / MT, mt = i % MT;       // the two MFMAs of one accumulator are five issues apart
    oacc[mt] = mfma32(vf[mt][h], P[h], oacc[mt]);
  };
  auto clear_o = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[mt][r] = 0.f;
    lsum = 0.f;
  };
  const int nt = p.kv_len / 64;               // launcher guarantees kv_len % 64 == 0, nt >= 4, aligned segments
  // first offset: exact maximum of the query's scores against the 32-key sub-tile at koff (+ bias); leaves the raw scores in s
  // (fp16 max-free pass: the bias follows the spread of the sample scores; returns whether even so an overflow is predicted)
  auto first_scores = [&](f32x16_t& s, float bias, uint32_t koff) __attribute__((always_inline)) -> bool {
    read_k(koff);
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) s = mfma32(kf[ks], qf[ks], s);
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    bool wide = false;
    if constexpr (F_SAMPLED) {
      if (bias != 0.f) {
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sm += s[r]; sq = fmaf(s[r], s[r], sq); }
        sm += __shfl_xor(sm, 32); sq += __shfl_xor(sq, 32);
        const float mean = sm * (1.f / 32.f);
        bias = f16_sampled_bias(mx, mean, sq * (1.f / 32.f) - mean * mean, f16_expected_max_sds(p.kv_len), wide);
      }
    }
    off = mx + bias;
    return wide;
  };
  auto finish = [&](bool check) __attribute__((always_inline)) -> bool {      // row sums, range checks of the max-free pass, stores
    const float l_tot = lsum + __shfl_xor(lsum, 32);
    if (check) {
      bool bad = !(l_tot < F_L_HI) || !(l_tot > F_L_LO);
#ifdef A3D_STORAGE_F16
      // a P that overflowed fp16 (inf) makes every dimension of its query's row inf or NaN
      float chk = 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) chk += fabsf(oacc[mt][r]);
      bad = bad || !(chk < 3.0e38f);
#endif
      if (__syncthreads_or(bad ? 1 : 0)) { dm_count(p, 1); return false; }          // (also: every wave is done with the LDS images)
    }
    const float inv = p.out_scale / l_tot;
    if (p.lse != nullptr && q_idx < p.q_len && g == 0)      // training: log2 of the softmax denominator
      p.lse[((int64_t)grp * p.heads + head) * p.q_len + q_idx] = __builtin_amdgcn_logf(l_tot) + off;
    if (q_idx < p.q_len) {      // lane holds O[q = l31][d = 32*mt + 8*qd + 4*g + j]
      uint16_t* orow = p.O + map_row(p.om, grp, q_idx) * p.om.ld + hoff;
      if (p.accumulate) {       // (workgroup-uniform) read-modify-write in the accumulators' own layout: 8-byte accesses
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const int d = 32 * mt + 8 * qd + 4 * g;
            const u32x2_t prev = *reinterpret_cast<const u32x2_t*>(orow + d);
            u32x2_t o;
            o[0] = pack16(oacc[mt][4 * qd] * inv + lo16(prev[0]), oacc[mt][4 * qd + 1] * inv + hi16(prev[0]));
            o[1] = pack16(oacc[mt][4 * qd + 2] * inv + lo16(prev[1]), oacc[mt][4 * qd + 3] * inv + hi16(prev[1]));
            *reinterpret_cast<u32x2_t*>(orow + d) = o;
          }
      } else {
        // The two halves of a wave hold alternate 4-dim groups of one row (half 0: dims 8 k .. + 3, half 1: 8 k + 4 .. + 7): one v_permlane32_swap
        // per packed word hands each half 8 CONSECUTIVE dims of two groups — ten 16-byte stores per lane instead of twenty 8-byte ones (at
        // 1 024 keys the store tail of the epilogue is a tenth of a workgroup's time: the stores are issue-bound, MI355X_MICROARCH.md)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            u32x4_t v;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const uint32_t w0 = pack16(oacc[mt][8 * pr + 2 * c] * inv, oacc[mt][8 * pr + 2 * c + 1] * inv);
              const uint32_t w1 = pack16(oacc[mt][8 * pr + 4 + 2 * c] * inv, oacc[mt][8 * pr + 4 + 2 * c + 1] * inv);
              const auto r = __builtin_amdgcn_permlane32_swap(w0, w1, false, false);
              v[c] = r[0];
              v[2 + c] = r[1];
            }
            *reinterpret_cast<u32x4_t*>(orow + 32 * mt + 16 * pr + 8 * g) = v;
          }
      }
    }
    return true;
  };

  // ================================================================================================================
  // Max-free pass: branch-free software pipeline
  // ================================================================================================================
  auto run_fast = [&]() __attribute__((always_inline)) -> bool {
    clear_o();
    f32x16_t sA, sB;
    u32x4_t pA[2], pB[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) { pA[h] = u32x4_t{0u, 0u, 0u, 0u}; pB[h] = u32x4_t{0u, 0u, 0u, 0u}; }

    // Step j (20 slots of one MFMA each): slots 0..9: O += V^T(j-1)·P(j-1) (fragments read during step j-1), slots 10..19: S(j+1) = K(j+1)·Q^T;
    // the VALU turns S(j) (sCur) into P(j) (pCur) over all slots.  kOff / vOff: LDS byte offsets of K sub-tile j+1 and V sub-tile j.
    // K fragment ks is read in slot 5 + ks (five slots ahead of its MFMA), V^T fragment i of sub-tile j in slot 10 + i (consumed in slot i of step j+1).
    // A step without the P·V block (the very first) finds its first five K fragments read already; one without the QK block (the very last) reads no V.
    auto step = [&](auto do_qk_c, auto do_pv_c, f32x16_t& sCur, f32x16_t& sNext, u32x4_t (&pCur)[2], u32x4_t (&pPrev)[2],
                    uint32_t kOff, uint32_t vOff, auto&& hook) __attribute__((always_inline)) {
      constexpr bool DO_QK = decltype(do_qk_c)::value, DO_PV = decltype(do_pv_c)::value;
      constexpr int NPV = DO_PV ? 2 * MT : 0, NQK = DO_QK ? KS : 0, NS = NPV + NQK;
      constexpr int KAHEAD = 5;
      uint32_t va = vlane + vOff, ke = kc_even + kOff, ko = kc_odd + kOff;
      asm volatile("" : "+v"(va), "+v"(ke), "+v"(ko));
      float e[NEXP];
      auto do_cvt = [&](auto c_c) __attribute__((always_inline)) {
        constexpr int c = decltype(c_c)::value, h = c / 4, jj = c % 4;
        pCur[h][jj] = pack16(e[2 * c], e[2 * c + 1]);
      };
      auto cdone = [](int s) constexpr { return s < 0 ? 0 : (NEXP * s / NS) / 2; };
      __builtin_amdgcn_sched_barrier(0);
      static_for<NS>([&](auto s_c) __attribute__((always_inline)) {
        constexpr int s = decltype(s_c)::value;
        if constexpr (s < NPV) {
          pv_mfma(s_c, pPrev);
        } else {
          constexpr int ks = s - NPV;
          if constexpr (ks == 0) {
            f32x16_t z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            sNext = mfma32(kf[0], qf[0], z);
          } else {
            sNext = mfma32(kf[ks], qf[ks], sNext);
          }
        }
        if constexpr (DO_QK) {          // (a step without the P·V block finds fragments 0 .. KAHEAD-1 read already)
          constexpr int ks_rd = s - (NPV - KAHEAD);
          if constexpr (ks_rd >= (DO_PV ? 0 : KAHEAD) && ks_rd < KS) read_k1(std::integral_constant<int, ks_rd>{}, ke, ko);
        }
        if constexpr (DO_QK && s >= NPV) read_v1(std::integral_constant<int, s - NPV>{}, va);
        hook(s_c);
        constexpr int E0 = NEXP * s / NS, E1 = NEXP * (s + 1) / NS;
        static_for<E1 - E0>([&](auto x_c) __attribute__((always_inline)) {
          constexpr int x = E0 + decltype(x_c)::value;
          e[x] = __builtin_amdgcn_exp2f(sCur[x] - off);
          lsum += e[x];
        });
        constexpr int C0 = cdone(s - 1), C1 = cdone(s);
        static_for<C1 - C0>([&](auto c_c) __attribute__((always_inline)) { do_cvt(std::integral_constant<int, C0 + decltype(c_c)::value>{}); });
        __builtin_amdgcn_sched_barrier(0);
      });
      {
        constexpr int C0 = cdone(NS - 1);
        static_for<NCVT - C0>([&](auto c_c) __attribute__((always_inline)) { do_cvt(std::integral_constant<int, C0 + decltype(c_c)::value>{}); });
      }
    };

    prologue_wait();
    {
      if constexpr (F_SAMPLED) {
        const bool wide = first_scores(sA, F_BIAS, (uint32_t)F_SAMPLE);       // offset from the sample keys
        // a vote, not an OR (flash_attn_dm.hip): the workgroup goes exact right away when more than a quarter of its queries predict an overflow
        if (__syncthreads_count(wide ? 1 : 0) * 4 > NT) { dm_count(p, 0); return false; }
        read_k(0u);                                                // ... then S(0) of keys 0..31
#pragma unroll
        for (int r = 0; r < 16; ++r) sA[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) sA = mfma32(kf[ks], qf[ks], sA);
      } else {
        first_scores(sA, F_BIAS, 0u);
      }
      {                                        // K(0) keys 32..63 for step 0: its first five fragments (the step reads the others itself)
        uint32_t ke = kc_even + (uint32_t)F_UNITB, ko = kc_odd + (uint32_t)F_UNITB;
        asm volatile("" : "+v"(ke), "+v"(ko));
        static_for<5>([&](auto ks_c) __attribute__((always_inline)) { read_k1(ks_c, ke, ko); });
      }
    }

    auto iteration = [&](int t, auto first_c, auto last_c, auto dma_c_) __attribute__((always_inline)) {
      constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value, DMA = decltype(dma_c_)::value;
      // tile t + 2 goes into the buffer tile t - 1 left at the barrier that ended the previous iteration
      // (a lane's source offset is read from the LDS table one slot ahead of its DMA instruction)
      [[maybe_unused]] uint32_t vo = 0u;
      auto even_hook = [&](auto s_c) __attribute__((always_inline)) {
        constexpr int s = decltype(s_c)::value;
        if constexpr (DMA && s % 3 == 0 && s < 9) vo = tab(s / 3);
        if constexpr (DMA && s % 3 == 1 && s < 9) dma_i(t + 2, std::integral_constant<int, s / 3>{}, vo);
      };
      auto odd_hook = [&](auto s_c) __attribute__((always_inline)) {
        constexpr int s = decltype(s_c)::value;
        if constexpr (DMA && s % 3 == 0 && s < 6) vo = tab(3 + s / 3);
        if constexpr (DMA && s % 3 == 1 && s < 6) dma_i(t + 2, std::integral_constant<int, 3 + s / 3>{}, vo);
      };
      const uint32_t tb = tile_base(t), tn = tile_base(t + 1);
      // even step j = 2t:  O += V(t-1)[32..63] P(2t-1), S(2t+1) from K(t) keys 32..63, P(2t) from S(2t); reads V(t)[0..31]
      step(std::true_type{}, std::integral_constant<bool, !FIRST>{}, sA, sB, pA, pB, tb + F_UNITB, tb + F_KB, even_hook);
      // odd step j = 2t+1: O += V(t)[0..31] P(2t), S(2t+2) from K(t+1) keys 0..31, P(2t+1) from S(2t+1); reads V(t)[32..63]
      step(std::integral_constant<bool, !LAST>{}, std::true_type{}, sB, sA, pB, pA, tn, tb + F_KB + F_UNITB, odd_hook);
      if constexpr (!LAST) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // tile t + 2 has landed for everybody; tile t is free
    };
    constexpr std::true_type Y{};
    constexpr std::false_type N{};
    if (nt > 2) {
      iteration(0, Y, N, Y);
      for (int t = 1; t < nt - 2; ++t) iteration(t, N, N, Y);
      iteration(nt - 2, N, N, N);
    } else {
      iteration(0, Y, N, N);
    }
    iteration(nt - 1, N, Y, N);
    {   // O += V(nt-1)[32..63] P(2nt-1)
      read_v(tile_base(nt - 1) + F_KB + F_UNITB);
      static_for<2 * MT>([&](auto i_c) __attribute__((always_inline)) { pv_mfma(i_c, pB); });
    }
    return finish(true);
  };

  // ================================================================================================================
  // Exact pass: lazy running maximum per 32-key sub-tile, un-pipelined (after an overflow of the max-free pass, or always)
  // ================================================================================================================
  auto run_exact = [&](auto rerun_c) __attribute__((always_inline)) {
    clear_o();
    f32x16_t sc;
    if constexpr (decltype(rerun_c)::value) prologue_issue();      // (every wave has left the LDS images: the vote / the row-sum check were barriers)
    prologue_wait();
    first_scores(sc, 0.f, 0u);
    for (int t = 0; t < nt; ++t) {
      if (t + 2 < nt) dma_tile(t + 2);
      const uint32_t tb = tile_base(t);
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        if (t > 0 || sub > 0) {
          read_k(tb + sub * F_UNITB);
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) sc = mfma32(kf[ks], qf[ks], sc);
          float mx = sc[0];
#pragma unroll
          for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
          if (__any(mx - off > LAZY_THR)) {
            const float delta = fmaxf(fmaxf(mx, __shfl_xor(mx, 32)) - off, 0.f);
            off += delta;
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            lsum *= alpha;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int r = 0; r < 16; ++r) oacc[mt][r] *= alpha;
          }
        }
        u32x4_t pf[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const float e0 = __builtin_amdgcn_exp2f(sc[8 * h + 2 * jj] - off), e1 = __builtin_amdgcn_exp2f(sc[8 * h + 2 * jj + 1] - off);
            lsum += e0 + e1;
            pf[h][jj] = pack16(e0, e1);
          }
        read_v(tb + F_KB + sub * F_UNITB);
        static_for<2 * MT>([&](auto i_c) __attribute__((always_inline)) { pv_mfma(i_c, pf); });
      }
      if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    finish(false);
  };

  dm_count(p, 2);
  if constexpr (TRY_NOMAX) {
    if (!run_fast()) run_exact(std::true_type{});
  } else {
    run_exact(std::false_type{});
  }
}

template <int FLAGS>
int launch_dm160(int groups, hipStream_t s, const AttnParams& p) {
  static uint64_t attr_done = 0;
  if (int rc = a3d_once_per_device(attr_done, [] {
        return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_attn_dm160_kernel<FLAGS>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, F_SMEM_BYTES); })) return rc;
  const int q_tiles = (p.q_len + 255) / 256;
  flash_attn_dm160_kernel<FLAGS><<<dim3((unsigned)(p.heads * q_tiles), (unsigned)groups), dim3(512), F_SMEM_BYTES, s>>>(p);
  return a3d_launch_status();
}

}  // namespace

// flags: 0 = exact pass only, 1 = max-free first pass.  Shapes: head_dim 160, kv_len % 64 == 0, kv_len >= 256 (nt >= 4 is not needed: >= 2),
// aligned 64-key tiles (checked by the caller).
int A3D_FN(a3d_launch_flash_dm160)(int flags, int groups, hipStream_t s, const AttnParams& p) {
  switch (flags) {
    case 0: return launch_dm160<0>(groups, s, p);
    case 1: return launch_dm160<1>(groups, s, p);
    default: return A3D_EINVAL;
  }
}
