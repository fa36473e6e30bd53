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
//  * bf16 storage only: NO row maximum.  P = exp2(S - m) is stored in bf16, whose exponent range is fp32's, and accumulated in
//    fp32; the result does not depend on m as long as nothing overflows.  m is the exact maximum of the first 32 keys; after
//    that the kernel only watches the row sums that the matrix pipe produces anyway (O^T row 40): once per 64-key tile one
//    compare per query sub-tile; when a sum passes 2^30 the offset moves by log2(sum) (same rare path as the exact kernel:
//    the pending P is folded in first).  A probability that still overflows (a score more than ~2^7 log2 units above
//    everything seen before it, inside one check interval) makes the row sum non-finite or > 2^100: the workgroup then
//    discards its result and re-runs with the exact running maximum (the interleaved kernel's arithmetic).  The exact path
//    is also what the fp16 build always runs (fp16 P overflows at 2^16).
#include "flash_common.h"

namespace {

constexpr int DM_ROWB = 80;                        // bytes per K / V row in LDS
constexpr int DM_UNITB = 32 * DM_ROWB;             // one 32-key sub-tile of K or of V
constexpr int DM_TILEB = 4 * DM_UNITB;             // [K keys 0..63 | V keys 0..63 (rows permuted)]
constexpr int DM_VOFF = 2 * DM_UNITB;              // V image inside a tile buffer
constexpr int DM_RING = 8;
constexpr int DM_CK = DM_RING * DM_TILEB;          // K constant chunk (16 B): 1.0 in contraction slot 40, zeros in 41..47
constexpr int DM_CV = DM_CK + 64;                  // V constant region: (1,0,0,0) pieces at DM_CV + {0, 80, 1280, 1360}, zero elsewhere
constexpr int DM_CV_BYTES = 1408;
constexpr int DM_SMEM_BYTES = DM_CV + DM_CV_BYTES;
constexpr float DM_L_BAD = 1.2676506e30f;          // 2^100: beyond this the max-free result is not trusted
constexpr float DM_BIAS = 40.f;                    // max-free offset = maximum of the first 32 scores + 40 (log2 units)

// LDS-DMA: 64 lanes x 16 B, lane i -> LDS[lds_dst + 16 i]; source = scalar base + per-lane byte offset.  Not counted by the
// compiler: s_waitcnt vmcnt by hand.
A3D_DEV void dm_glds16(uint32_t voff, const void* sbase, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}
// the same for lanes 0..15 only; the exec mask is switched inside the statement (a compiler-visible branch in a pipeline step
// lets the optimiser sink the step's v_exp below it)
A3D_DEV void dm_glds16_q(uint32_t voff, const void* sbase, uint32_t lds_dst) {
  unsigned keep;
  uint64_t ex;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, 0xffff\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep), "=&s"(ex) : "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}
A3D_DEV const uint16_t* dm_scalar(const uint16_t* ptr) {      // wave-uniform by construction; say so
  const uint64_t a = (uint64_t)(uintptr_t)ptr;
  return (const uint16_t*)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a));
}

extern __shared__ __attribute__((aligned(16))) uint8_t dm_smem[];

A3D_DEV u32x4_t dm_lds128(uint32_t off) { return *reinterpret_cast<const u32x4_t*>(dm_smem + off); }
A3D_DEV u32x2_t dm_ldstr(uint32_t off) { return lds_tr16_b64(reinterpret_cast<const uint16_t*>(dm_smem + off)); }

// FLAGS: 1 = max-free softmax with exact re-run (bf16 storage only), 2 = static s_setprio(1) for the second-dispatched half
//        of the workgroup (MI355X_MICROARCH.md "static priority for the younger half")
template <int FLAGS>
__global__ __launch_bounds__(512, 2) void flash_attn_dm_kernel(const AttnParams p) {
  constexpr int D = 40, QT = 2, KS = 3, MT = 2, NT = 512, BQ = 512;
  constexpr int KS_PAD = 2, G_PAD = 1;                    // fragment slot of contraction index 40
  constexpr int NEXP = 16 * QT, NCVT = 8 * QT;
#ifdef A3D_STORAGE_F16
  constexpr bool TRY_NOMAX = false;
#else
  constexpr bool TRY_NOMAX = (FLAGS & 1) != 0;
#endif

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
    const int vb = 2 * i - 64;      // byte offset inside the V constant region
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
  const int64_t kgbase = (grp / p.km.gdiv) * p.km.ga + (grp % p.km.gdiv) * p.km.gb;
  const uint32_t seg_len = (uint32_t)p.km.seg_len;
  const int64_t tile_step = (int64_t)64 * ld;
  const int64_t wrap_step = (p.km.seg_stride - p.km.seg_len) * ld;
  auto slot_src = [&](int slot) -> uint32_t {
    const bool isk = slot < 320;
    const int s2 = isk ? slot : slot - 320;
    const int prow = s2 / 5, c = s2 % 5;
    const int key = isk ? prow : ((prow & ~15) | (4 * (prow & 3) + ((prow >> 2) & 3)));
    return (uint32_t)(((int64_t)key * ld + c * 8) * 2);
  };
  const uint32_t voffA = slot_src(64 * w + lane);
  const uint32_t voffB = slot_src(512 + 16 * w + i16);

  // ---- fragment addressing (byte offsets into dm_smem; the tile / sub-tile offset is added per step, scaled by 0 for the
  // lanes that read constants)
  const uint32_t klane = (uint32_t)(kperm(l31) * DM_ROWB + 16 * g);             // K fragments 0, 1: + 32 ks
  const uint32_t klane2 = g ? (uint32_t)DM_CK : klane + 64u;                    // K fragment 2: dims 32..39 | constant chunk
  const uint32_t kmul = g ? 0u : 1u;
  const int c4 = i16 & 3;
  const uint32_t vrow = (uint32_t)((4 * (i16 >> 2) + 2 * (q4 >> 1)) * DM_ROWB);
  const uint32_t vlane0 = vrow + (uint32_t)(2 * (16 * (q4 & 1) + 4 * c4));      // O^T rows 0..31: all data
  const bool v1_data = (q4 & 1) == 0 && c4 < 2;                                 // O^T rows 32..63: dims 32..39 | ones | zeros
  const uint32_t vlane1 = v1_data ? vrow + (uint32_t)(2 * (32 + 4 * c4)) : (((q4 & 1) == 0 && c4 == 2) ? (uint32_t)DM_CV : (uint32_t)(DM_CV + 8));
  const uint32_t vmul = v1_data ? 1u : 0u;

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
  auto dma_a = [&](int tile) __attribute__((always_inline)) {
    dm_glds16(voffA, gA, lds0 + (uint32_t)((tile & (DM_RING - 1)) * DM_TILEB + 1024 * w));
  };
  auto dma_b = [&](int tile) __attribute__((always_inline)) {      // second instruction of a tile; then the bases move on
    dm_glds16_q(voffB, gB, lds0 + (uint32_t)((tile & (DM_RING - 1)) * DM_TILEB + 8192 + 256 * w));
    seg_off += 64;
    int64_t stp = tile_step;
    if (seg_off >= seg_len) { stp += wrap_step; seg_off = 0; }
    gA += stp; gB += stp;
  };
  f32x16_t oacc[QT][MT];
  u32x4_t kf[KS];
  u32x4_t vf[MT][2];          // V^T fragments of the sub-tile whose P is multiplied next
  auto read_k = [&](uint32_t koff) __attribute__((always_inline)) {          // K fragments of the sub-tile at byte offset koff
    const uint32_t a = klane + koff, a2 = klane2 + kmul * koff;
    kf[0] = dm_lds128(a); kf[1] = dm_lds128(a + 32); kf[2] = dm_lds128(a2);
  };
  auto read_v = [&](uint32_t voff) __attribute__((always_inline)) {
    const uint32_t a0 = vlane0 + voff, a1 = vlane1 + vmul * voff;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const u32x2_t lo = dm_ldstr((mt ? a1 : a0) + (16 * h) * DM_ROWB);
        const u32x2_t hi = dm_ldstr((mt ? a1 : a0) + (16 * h + 1) * DM_ROWB);
        vf[mt][h] = u32x4_t{lo[0], lo[1], hi[0], hi[1]};
      }
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
  };
  const int nt = p.kv_len / 64;               // launcher guarantees kv_len % 64 == 0, nt >= 4, aligned segments
  auto prologue_dma = [&]() __attribute__((always_inline)) {      // tiles 0, 1, 2 requested; 0 and 1 complete
    dma_reset();
    dma_a(0); dma_b(0);
    dma_a(1); dma_b(1);
    dma_a(2); dma_b(2);
    asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
  };
  // first offset of a query: exact maximum of its first 32 scores (+ bias); leaves the re-based scores in s
  auto first_scores = [&](f32x16_t (&s)[QT], float (&m_off)[QT], float bias) __attribute__((always_inline)) {
    read_k(0u);
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
      const float new_off = round16(mx + bias);
      m_off[qs] = new_off;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[qs][r] -= new_off;
      if (g == G_PAD) qf[qs][KS_PAD][0] = pack16(-new_off, 0.f);
    }
  };
  auto store_out = [&](const float (&inv)[QT]) __attribute__((always_inline)) {
    // lane holds O[q = l31][d = 32*mt + 8*qd + 4*g + j] for each query sub-tile
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
            if (d < D) {
              float v[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = oacc[qs][mt][4 * qd + j] * inv[qs];
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
      constexpr int NPV = DO_PV ? 2 * MT * QT : 0, NQK = DO_QK ? KS * QT : 0, NS = NPV + NQK;
      constexpr int KSLOT = 3, VS0 = NPV + 1;
      const uint32_t va0 = vlane0 + vOff, va1 = vlane1 + vmul * vOff;
      float e[NEXP];
      auto do_cvt = [&](auto c_c) __attribute__((always_inline)) {
        constexpr int c = decltype(c_c)::value;
        constexpr int qs = c / 8, h = (c / 4) % 2, jj = c % 4;
        pCur[qs][h][jj] = pack16(e[2 * c], e[2 * c + 1]);
      };
      auto read_vh = [&](auto i_c) __attribute__((always_inline)) {      // half i of the 8 V^T fragment halves of sub-tile j
        constexpr int i = decltype(i_c)::value, mt = i / 4, h = (i / 2) % 2, rr = i % 2;
        const u32x2_t t = dm_ldstr((mt ? va1 : va0) + (16 * h + rr) * DM_ROWB);
        vf[mt][h][2 * rr] = t[0];
        vf[mt][h][2 * rr + 1] = t[1];
      };
      __builtin_amdgcn_sched_barrier(0);
      static_for<NS>([&](auto s_c) __attribute__((always_inline)) {
        constexpr int s = decltype(s_c)::value;
        if constexpr (s < NPV) {
          constexpr int mt = s / (2 * QT), h = (s / QT) % 2, qs = s % QT;
          oacc[qs][mt] = mfma32(vf[mt][h], pPrev[qs][h], oacc[qs][mt]);
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
        if constexpr (DO_QK && s >= VS0 && s < VS0 + 4) {
          read_vh(std::integral_constant<int, 2 * (s - VS0)>{});
          read_vh(std::integral_constant<int, 2 * (s - VS0) + 1>{});
        }
        hook(s_c);
        constexpr int E0 = NEXP * s / NS, E1 = NEXP * (s + 1) / NS;
        static_for<E1 - E0>([&](auto x_c) __attribute__((always_inline)) {
          constexpr int x = E0 + decltype(x_c)::value;
          e[x] = __builtin_amdgcn_exp2f(sCur[x / 16][x % 16]);
        });
        constexpr int C0 = (s == 0) ? 0 : (NEXP * (s - 1) / NS) / 2, C1 = E0 / 2;
        static_for<C1 - C0>([&](auto c_c) __attribute__((always_inline)) { do_cvt(std::integral_constant<int, C0 + decltype(c_c)::value>{}); });
        __builtin_amdgcn_sched_barrier(0);
      });
      {
        constexpr int C0 = (NEXP * (NS - 1) / NS) / 2;
        static_for<NCVT - C0>([&](auto c_c) __attribute__((always_inline)) { do_cvt(std::integral_constant<int, C0 + decltype(c_c)::value>{}); });
      }
    };

    prologue_dma();
    {
      float m_off[QT];
      first_scores(sA, m_off, DM_BIAS);
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
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
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
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int qs = 0; qs < QT; ++qs) oacc[qs][mt] = mfma32(vf[mt][h], pB[qs][h], oacc[qs][mt]);
    }
    float inv[QT];
    bool bad = false;
#pragma unroll
    for (int qs = 0; qs < QT; ++qs) {
      const float l_tot = __shfl(oacc[qs][1][4], l31);        // O^T row 40 = register 4 of the second tile, half 0
      bad = bad || !(l_tot < DM_L_BAD) || !(l_tot > 0.f);
      inv[qs] = p.out_scale / l_tot;
    }
    if (__syncthreads_or(bad ? 1 : 0)) return false;          // (also: every wave is done with the LDS images)
    store_out(inv);
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
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[qs][mt][r] *= alpha;
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
        read_v(tb + DM_VOFF + sub * DM_UNITB);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int qs = 0; qs < QT; ++qs) oacc[qs][mt] = mfma32(vf[mt][h], pf[qs][h], oacc[qs][mt]);
      }
      if (t + 1 < nt) {
        if (more) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
    float inv[QT];
#pragma unroll
    for (int qs = 0; qs < QT; ++qs) inv[qs] = p.out_scale / __shfl(oacc[qs][1][4], l31);
    store_out(inv);
  };

  __syncthreads();        // constant region written
  if constexpr (TRY_NOMAX) {
    if (!run_fast()) run_exact();
  } else {
    run_exact();
  }
}

template <int FLAGS>
int launch_dm(int groups, hipStream_t s, const AttnParams& p) {
  static uint64_t attr_done = 0;
  if (int rc = a3d_once_per_device(attr_done, [] {
        return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_attn_dm_kernel<FLAGS>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, DM_SMEM_BYTES); })) return rc;
  const int q_tiles = (p.q_len + 511) / 512;
  flash_attn_dm_kernel<FLAGS><<<dim3((unsigned)(p.heads * q_tiles), (unsigned)groups), dim3(512), DM_SMEM_BYTES, s>>>(p);
  return a3d_launch_status();
}

}  // namespace

// flags: see flash_attn_dm_kernel.  Shapes: head_dim 40, kv_len % 64 == 0, kv_len >= 256, aligned segments (checked by the caller).
int A3D_FN(a3d_launch_flash_dm)(int flags, int groups, hipStream_t s, const AttnParams& p) {
  switch (flags) {
    case 0: return launch_dm<0>(groups, s, p);
    case 1: return launch_dm<1>(groups, s, p);
    case 2: return launch_dm<2>(groups, s, p);
    case 3: return launch_dm<3>(groups, s, p);
    default: return A3D_EINVAL;
  }
}
