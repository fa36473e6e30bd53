// Persistent GEMM / implicit-GEMM 3x3 convolution with a four-stage LDS ring (gfx950) — a round-3 EXPERIMENT, not the default:
// a3d_tune_gemm(9) / (10) select it, bit-identical to the shipped kernels (same K order, same epilogue; tests/test_hip_kernels_gpu.py).
//
// Same tile and wave layout as gemm_persist_kernel (gemm_conv.hip): one 512-thread workgroup per CU, 256 x (NB*64) output tile,
// 8 waves as 4(M) x 2(N), LDS-DMA operand staging, the epilogue of gemm_common.h.  What changes is the operand pipeline.
// Measured on MI355X (profiles/r3_microbench_gemmscale.log): with two 64-wide K-tile stages a single CU runs one K-tile per
// 0.88 us, 32 CUs one per 0.97 us, all 256 CUs together one per 2.3-2.6 us.  The hypothesis tested here: every workgroup asks for
// its next 72 KB at the same barrier and has one K-tile of compute to hide (latency + burst / bandwidth) behind, so a deeper,
// smoother request stream should close the gap.  K advances in half-tiles of 32, four stages of 36 KB:
//   * DMA for half-tile h+4 is issued as soon as stage h is free (3 half-tiles = 1.5 K-tiles of compute ahead of its use);
//     with SPREAD the five DMA instructions of a stage are interleaved with the MFMAs of one half-tile of compute;
//   * the barrier of half-tile h sits in the MIDDLE of its compute: before it every wave has pulled its last fragments of
//     stage h into registers (so the stage can be refilled) and has seen its own DMA pieces of stage h+1 land; after it the
//     fragments of h+1 are requested while the second k-step of h still feeds the matrix pipe — no LDS latency is exposed;
//   * waits are counted (s_waitcnt vmcnt(N) with N = DMA instructions this wave issued after the ones it needs: exactly NI per
//     stage, so N is a compile-time constant per pipeline state), never vmcnt(0) in the steady state.
// Result: faster on an idle chip (0.80-0.84 us per K-tile on one CU) and 5-15 % SLOWER at full occupancy (2.5-2.8 us): the
// full-chip period is set by a shared resource that scales with the number of requests, not by exposed latency — 64-byte row
// segments double the request count of the 128-byte ones.  The two-stage kernel stays the default (profiles/README.md, round 3).
// LDS image of a stage: X rows (256 x 64 B) then W rows (NB*64 x 64 B); a DMA instruction writes 16 rows x 64 B lane-linearly;
// 16-byte chunk c of row r sits at slot c ^ ((r >> 2) & 3) (applied on the per-lane SOURCE address and on the fragment read
// address): the 16 lanes of a ds_read_b128 group then cover all 64 banks exactly once.
// At a tile boundary the two stages consumed last hold the epilogue's transposition buffers (4 waves each), the next tile's
// first two half-tiles are already in flight in the other two.
#include "gemm_common.h"

namespace {

constexpr int RBM = 256;
template <int NB> struct RCfg {
  static constexpr int BN = NB * 64;
  static constexpr int XB = RBM * 64;                  // X half-tile: 256 rows x 32 k x 2 B
  static constexpr int WB = BN * 64;
  static constexpr int EPI_WAVE = 32 * 68 * 4;         // one wave's transposition buffer
  static constexpr int STG = 36864;                    // stage stride: X + W for NB = 5, padded for NB = 4 (4 x EPI_WAVE must fit)
  static constexpr int BIAS_OFF = 4 * STG;             // [2 tile parities][bias fp32 @0 | rowbias 16-bit @1280]
  static constexpr int BIAS_STRIDE = 2048;
  static constexpr int SMEM = 4 * STG + 2 * BIAS_STRIDE;
  static constexpr int NI = (NB == 5) ? 5 : 4;         // DMA instructions per wave per half-tile (exact: the counted waits rely on it)
  static_assert(XB + WB <= STG && 4 * EPI_WAVE <= STG, "stage layout");
};

// LDS-DMA of lanes 0..31 only (8 rows x 64 B); exec is switched inside the statement so the instruction count per stage stays exact
A3D_DEV void glds16_s_lo32(uint32_t voff, const void* sbase, uint32_t lds_dst) {
  unsigned keep;
  uint64_t ex;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, 0xffffffff\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep), "=&s"(ex) : "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}

template <int CONV, int EPI, int NB, bool RES, bool SPREAD>
__global__ __launch_bounds__(512, 1) void gemm_ring_kernel(const GemmParams p) {
  using RC = RCfg<NB>;
  constexpr int NI = RC::NI;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  char* const smem_b = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, g = lane >> 5;
  const int lr = lane >> 2, pos = lane & 3;            // DMA role: row within the 16-row piece, 16-byte slot
  const uint32_t lds0 = lds_addr(smem);

  const int64_t ntiles = p.tiles_m * p.tiles_n;
  const int64_t G = gridDim.x;
  int64_t t = xcd_remap(blockIdx.x, G);
  if (t >= ntiles) return;
  const int nk2 = (int)(p.K / 32);                     // half-tiles per output tile (>= 2)

  // fragment reads: lane (l31, g) reads logical chunk 2*ks + g of row (.. + l31), stored at chunk ^ ((row >> 2) & 3); block base
  // rows are multiples of 32, so offset(ks) = koff0 ^ (ks << 5)
  const uint32_t koff0 = (uint32_t)((g ^ ((l31 >> 2) & 3)) << 4);
  const uint32_t xrd = (uint32_t)(wm * 64 + l31) * 64u;
  const int wblk = (NB == 5) ? wn * 4 : wn * NB;       // column blocks of wave column wn: see gemm_persist_kernel
  const int wblk_last = (NB == 5) ? 8 + wn : wn * NB + NB - 1;
  const uint32_t wrd = (uint32_t)RC::XB + (uint32_t)(wblk * 32 + l31) * 64u;
  const uint32_t wrd_last = (uint32_t)RC::XB + (uint32_t)(wblk_last * 32 + l31) * 64u;

  // ---- DMA sources: a piece is 16 consecutive tile rows; lane (lr, pos) fetches chunk pos ^ ((lr >> 2) & 3) of row lr of the piece
  const uint32_t swz = (uint32_t)((pos ^ ((lr >> 2) & 3)) << 4);
  const uint32_t vx0 = (uint32_t)(lr * p.ldx * 2) + swz;
  const uint32_t vw0 = (uint32_t)(lr * p.ldw * 2) + swz;
  // NB = 5: W rows 256 .. 319 go as eight 8-row half pieces (lanes 0..31), wave w takes rows 256 + 8 w + lr: ((8 w + lr) >> 2) & 3
  const uint32_t vw1 = (uint32_t)(lr * p.ldw * 2) + (uint32_t)((pos ^ (((lr >> 2) + 2 * (wid & 1)) & 3)) << 4);
  // conv: byte offset of the row's tap-(0,0) pixel from (X - cbias) and 9-bit in-image tap masks of this lane's two rows
  uint32_t aoff[CONV ? 2 : 1];
  uint32_t amask = 0;
  const int64_t cbias = CONV ? ((int64_t)p.Wd + 1) * p.Cin : 0;

  // ---- load cursor: walks the stream of half-tiles (tile after tile) ahead of the compute cursor
  int64_t ld_t = t, ld_m0 = 0, ld_n0 = 0;
  int ld_par = 0, ld_kh = 0, ld_h = 0;                 // bias-area parity, half-tile within the tile, stream index of the next issue
  int ld_k = 0, itap = 0, ici0 = 0;
  bool ld_more = true;
  auto setup_tile = [&](int64_t tt) {
    const int64_t tile_n = tt % p.tiles_n, tile_m = tt / p.tiles_n;
    ld_m0 = tile_m * RBM; ld_n0 = tile_n * RC::BN;
    ld_kh = 0; ld_k = 0; itap = 0; ici0 = 0;
    ld_par ^= 1;
    if constexpr (CONV != 0) {
      amask = 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = (wid * 2 + i) * 16 + lr;
        const int64_t m = ld_m0 + r;
        const int hw = p.Ho * p.Wo;
        const int b = (int)(m / hw);
        const int rem = (int)(m - (int64_t)b * hw);
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        uint32_t mask = 0;
        if constexpr (CONV == 2) {
          // nearest-2x upsample folded into the gather (gemm_persist_kernel): tap (ky, kx) of output pixel (oy, ox) reads source
          // pixel (sy0 + ((ky + ey) >> 1), sx0 + ((kx + ex) >> 1)), sy0 = (oy - 1) >> 1, ey = 1 for even oy (likewise x)
          const int sy0 = (oy - 1) >> 1, sx0 = (ox - 1) >> 1;
          aoff[i] = (uint32_t)(((((int64_t)b * p.H + sy0) * p.Wd + sx0) * p.Cin + cbias) * 2) + swz;
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) {
            const int yy = oy + tp / 3 - 1, xx = ox + tp % 3 - 1;
            if (yy >= 0 && yy < p.He && xx >= 0 && xx < p.We) mask |= 1u << tp;
          }
          amask |= (mask << (9 * i)) | ((uint32_t)(~oy & 1) << (18 + 2 * i)) | ((uint32_t)(~ox & 1) << (19 + 2 * i));
        } else {
          const int y0 = oy * p.stride - 1, x0 = ox * p.stride - 1;
          aoff[i] = (uint32_t)(((((int64_t)b * p.H + y0) * p.Wd + x0) * p.Cin + cbias) * 2) + swz;
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) {
            const int yy = y0 + tp / 3, xx = x0 + tp % 3;
            if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd) mask |= 1u << tp;
          }
          amask |= mask << (9 * i);
        }
      }
    }
  };
  // The next half-tile of the stream goes out as issue_begin(), NI x issue_piece(j), issue_end(): exactly NI DMA instructions per
  // wave (+ the tile's bias vectors with its first half-tile).  SPREAD interleaves the pieces with the MFMAs of one half-tile of
  // compute; otherwise they go back to back right after the barrier that frees their stage.
  uint32_t is_dst = 0;
  auto issue_begin = [&]() {
    is_dst = lds0 + (uint32_t)(ld_h & 3) * RC::STG;
    if (ld_kh == 0) {
      const uint32_t bdst = lds0 + (uint32_t)RC::BIAS_OFF + (uint32_t)ld_par * RC::BIAS_STRIDE;
      if (p.bias) {
        if (wid == 0) glds16_s((uint32_t)lane * 16u, p.bias + ld_n0, bdst);
        if (NB == 5 && wid == 1) { if (lane < 16) glds16_s((uint32_t)lane * 16u, p.bias + ld_n0 + 256, bdst + 1024u); }
      }
      if (EPI == EPI_LINEAR && p.rowbias && wid == 2) {
        if (lane < RC::BN / 8) glds16_s((uint32_t)lane * 16u, scalar_ptr(p.rowbias + (ld_m0 / p.rb_div) * p.N + ld_n0), bdst + 1280u);
      }
    }
  };
  auto issue_piece = [&](int j) {        // j is a constant wherever this is called: 0, 1 = X pieces, 2, 3 = W pieces, 4 = W half piece (NB = 5)
    if (j < 2) {
      const int pc = wid * 2 + j;
      if constexpr (CONV != 0) {
        const int ky = itap / 3, kx = itap - ky * 3;
        const uint16_t* xb = CONV == 2 ? p.X + (ici0 - cbias) : p.X + ((int64_t)(ky * p.Wd + kx) * p.Cin + ici0 - cbias);
        uint32_t vo = aoff[j];
        if constexpr (CONV == 2) {
          const uint32_t rowb = (uint32_t)(p.Wd * p.Cin * 2), colb = (uint32_t)(p.Cin * 2);
          const uint32_t yE = (uint32_t)((ky + 1) >> 1) * rowb, yO = (uint32_t)(ky >> 1) * rowb;
          const uint32_t xE = (uint32_t)((kx + 1) >> 1) * colb, xO = (uint32_t)(kx >> 1) * colb;
          vo += (((amask >> (18 + 2 * j)) & 1u) ? yE : yO) + (((amask >> (19 + 2 * j)) & 1u) ? xE : xO);
        }
        // out-of-image tap: this lane's 16 bytes come from the zero page (one instruction per piece either way)
        const char* src = ((amask >> (itap + 9 * j)) & 1u) ? reinterpret_cast<const char*>(xb) + vo
                                                           : reinterpret_cast<const char*>(g_zero_page) + pos * 16;
        glds16_v(src, is_dst + (uint32_t)pc * 1024u);
      } else {
        glds16_s(vx0, p.X + (ld_m0 + pc * 16) * p.ldx + ld_k, is_dst + (uint32_t)pc * 1024u);
      }
    } else if (j < 4) {
      const int pc = wid * 2 + (j - 2);
      glds16_s(vw0, p.W + (ld_n0 + pc * 16) * p.ldw + ld_k, is_dst + (uint32_t)RC::XB + (uint32_t)pc * 1024u);
    } else {
      glds16_s_lo32(vw1, p.W + (ld_n0 + 256 + wid * 8) * p.ldw + ld_k, is_dst + (uint32_t)RC::XB + 16384u + (uint32_t)wid * 512u);
    }
  };
  auto issue_end = [&]() {
    if constexpr (CONV != 0) {
      ici0 += 32;
      if (ici0 >= p.Cin) { ici0 = 0; ++itap; }
    }
    ld_k += 32;
    ++ld_h;
    if (++ld_kh == nk2) {
      ld_t += G;
      if (ld_t < ntiles) setup_tile(ld_t);
      else ld_more = false;
    }
  };
  auto issue_next = [&]() {
    issue_begin();
#pragma unroll
    for (int j = 0; j < NI; ++j) issue_piece(j);
    issue_end();
  };

  f32x16_t acc[NB][2];   // [tn][tm]
  u32x4_t fx[2][2], fw[2][NB];          // slot 0: k-step 0 of a half-tile, slot 1: k-step 1
  auto load_frags = [&](int slot, int stage, int ks) {
    const char* base = smem_b + stage * RC::STG;
    const uint32_t ko = koff0 ^ (uint32_t)(ks << 5);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) fx[slot][tm] = *reinterpret_cast<const u32x4_t*>(base + xrd + tm * 2048 + ko);
#pragma unroll
    for (int tn = 0; tn < NB - 1; ++tn) fw[slot][tn] = *reinterpret_cast<const u32x4_t*>(base + wrd + tn * 2048 + ko);
    fw[slot][NB - 1] = *reinterpret_cast<const u32x4_t*>(base + wrd_last + ko);
  };
  auto mfma_step = [&](int slot) {
#pragma unroll
    for (int tn = 0; tn < NB; ++tn)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) acc[tn][tm] = mfma32(fw[slot][tn], fx[slot][tm], acc[tn][tm]);
  };
  // SPREAD: the MFMAs of one k-step with DMA pieces of the stage being filled issued between them (phase 0: the k-step right after
  // the barrier carries pieces 0 .. NI - 3, phase 1: the next half-tile's first k-step carries the last two)
  auto mfma_step_dma = [&](int slot, int phase, bool on) {
#pragma unroll
    for (int tn = 0; tn < NB; ++tn)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        acc[tn][tm] = mfma32(fw[slot][tn], fx[slot][tm], acc[tn][tm]);
        const int q = tn * 2 + tm;
        int pc = -1;
        if (NB == 5) pc = phase == 0 ? (q == 1 ? 0 : q == 4 ? 1 : q == 7 ? 2 : -1) : (q == 2 ? 3 : q == 6 ? 4 : -1);
        else pc = phase == 0 ? (q == 1 ? 0 : q == 5 ? 1 : -1) : (q == 1 ? 2 : q == 5 ? 3 : -1);
        if (pc >= 0) {
          __builtin_amdgcn_sched_barrier(0);
          if (on) issue_piece(pc);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
  };
  // VMEM stores one wave issues per epilogue (newer than the DMA of the next tile's first two half-tiles, older than the two
  // stages issued right after the epilogue)
  constexpr int NST = (EPI == EPI_GEGLU) ? 2 * 2 * ((NB + 1) / 2) : 2 * (4 * (NB / 2) + 2 * (NB & 1));
  // wait until this wave's DMA pieces of half-tile `target` have landed: `newer` = stages issued after it (each exactly NI
  // instructions; anything else issued after it only makes the wait stricter), `stores` = the epilogue's stores sit in between
  auto wait_landed = [&](int newer, bool stores) {
    if (stores && newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(2 * NI + NST) : "memory");
    else if (!stores && newer >= 3) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(3 * NI) : "memory");
    else if (!stores && newer == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(2 * NI) : "memory");
    else if (!stores && newer == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  setup_tile(t);
  for (int i = 0; i < 4 && ld_more; ++i) issue_next();
  wait_landed(ld_h - 1, false);
  __builtin_amdgcn_s_barrier();
  int h = 0;                            // stream index of the half-tile being computed
  load_frags(0, 0, 0);
  bool first_tile = true;
  int cur_par = 1;                      // bias-area parity of the tile being computed (the load cursor flips ld_par per tile, starting at 1)
  for (;;) {
    const int64_t tile_n = t % p.tiles_n, tile_m = t / p.tiles_n;
    const int64_t m0 = tile_m * RBM, n0 = tile_n * RC::BN;
    const int64_t tnext = t + G;
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    bool pend = false;                  // SPREAD: a stage whose last two DMA pieces are still to be issued
    for (int kh = 0; kh < nk2; ++kh, ++h) {
      load_frags(1, h & 3, 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (SPREAD) {
        mfma_step_dma(0, 1, pend);
        if (pend) { issue_end(); pend = false; }
      } else {
        mfma_step(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // own DMA pieces of h+1 landed, own fragment reads of stage h complete ...
      wait_landed(ld_h - h - 2, kh == 0 && !first_tile && p.vm_counted);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();     // ... everybody's: stage h may be refilled, stage h+1 may be read
      const bool go = kh < nk2 - 2 && ld_more && ld_h <= h + 4;
      if constexpr (SPREAD) {
        if (go) issue_begin();
      } else {
        if (go) issue_next();
      }
      if (kh + 1 < nk2) load_frags(0, (h + 1) & 3, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (SPREAD) {
        mfma_step_dma(1, 0, go);
        pend = go;
      } else {
        mfma_step(1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: the two stages consumed last (h-1: waves 0-3, h-2: waves 4-7) hold the transposition buffers
    persist_epilogue<EPI, NB, RES>(p, acc, reinterpret_cast<float*>(smem_b + ((h - 1 - (wid >> 2)) & 3) * RC::STG + (wid & 3) * RC::EPI_WAVE),
                                   reinterpret_cast<const float*>(smem_b + RC::BIAS_OFF + cur_par * RC::BIAS_STRIDE),
                                   reinterpret_cast<const uint16_t*>(smem_b + RC::BIAS_OFF + cur_par * RC::BIAS_STRIDE + 1280),
                                   m0, n0, wm, wblk, wblk_last, lane);
    if (tnext >= ntiles) break;
    __builtin_amdgcn_s_barrier();       // every wave is done with its transposition buffer: the two stages may be refilled
    while (ld_more && ld_h < h + 4) issue_next();
    load_frags(0, h & 3, 0);            // the next tile's first half-tile landed before the barrier of the last iteration
    t = tnext;
    first_tile = false;
    cur_par ^= 1;
  }
}

template <int CONV, int EPI, int NB, bool RES, bool SPREAD = false>
int launch_ring(hipStream_t stream, const GemmParams& p, int cus) {
  using RC = RCfg<NB>;
  if constexpr (!SPREAD && CONV == 0) {
    if (p.ring_spread) return launch_ring<CONV, EPI, NB, RES, true>(stream, p, cus);
  }
  static uint64_t attr_done = 0;
  if (int rc = a3d_once_per_device(attr_done, [] {
        return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring_kernel<CONV, EPI, NB, RES, SPREAD>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, RC::SMEM); })) return rc;
  const int64_t ntiles = p.tiles_m * p.tiles_n;
  const unsigned grid = (unsigned)(ntiles < cus ? ntiles : cus);
  gemm_ring_kernel<CONV, EPI, NB, RES, SPREAD><<<dim3(grid), dim3(512), RC::SMEM, stream>>>(p);
  return a3d_launch_status();
}

template <int CONV>
int launch_ring_conv(int epi, int nb, hipStream_t stream, const GemmParams& p, int cus) {
  if (epi == EPI_GEGLU) {
    if constexpr (CONV == 0) { if (nb == 4) return launch_ring<0, EPI_GEGLU, 4, false>(stream, p, cus); }
    return A3D_EUNSUPPORTED;
  }
  if (nb == 5) return p.R ? launch_ring<CONV, EPI_LINEAR, 5, true>(stream, p, cus) : launch_ring<CONV, EPI_LINEAR, 5, false>(stream, p, cus);
  if (nb == 4) return p.R ? launch_ring<CONV, EPI_LINEAR, 4, true>(stream, p, cus) : launch_ring<CONV, EPI_LINEAR, 4, false>(stream, p, cus);
  return A3D_EUNSUPPORTED;
}

}  // namespace

// p.tiles_m / tiles_n / vm_counted filled by the caller (try_launch_persist, gemm_conv.hip), which also checked eligibility
int A3D_FN(a3d_launch_gemm_ring)(int conv, int epi, int nb, hipStream_t stream, const GemmParams& p, int cus) {
  if (conv == 0) return launch_ring_conv<0>(epi, nb, stream, p, cus);
  if (conv == 1) return launch_ring_conv<1>(epi, nb, stream, p, cus);
  return launch_ring_conv<2>(epi, nb, stream, p, cus);
}
