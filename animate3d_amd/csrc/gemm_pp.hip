// Persistent GEMM / implicit-GEMM 3x3 convolution for the big token matrices (levels 0-2 of the UNet: M = 32768 ... 524288).
//
// One 512-thread workgroup per CU (8 waves as 4(M) x 2(N), each wave owns 64 x NB*32 outputs = 2 x NB MFMA 32x32x16 tiles, NB = 5 =>
// BN = 320: every channel count of the model is a multiple of 320, and a full-width N tile means A is read from HBM exactly once at
// level 0) walks a strided list of 256 x (NB*64) output tiles.  Operand K-tiles (64 wide) go global -> LDS directly with
// global_load_lds_dwordx4 (no staging registers, no ds_write), two LDS stages.  The LDS image is lane-linear per DMA instruction
// (8 rows x 128 B), so the bank swizzle (16-byte chunk index ^= (row >> 1) & 7, conflict-free for the 16-lane groups of ds_read_b128)
// is applied on the per-lane SOURCE address and again on the fragment read address.  The K-tile stream runs across tile boundaries:
// the first K-tile of the next tile is in flight while the epilogue of the current one drains through the LDS stage that was just
// consumed.  3x3 convolutions use the same kernel: the A "row" pointer is the tap-(0,0) pixel, out-of-image taps read a zero page.
//
// PING-PONG main loop (round 4).  Rounds 1-3 ran the eight waves in lockstep: wait for the DMA, barrier, nine DMA issues per wave,
// fragment reads, 40 MFMAs — the matrix pipe idled while every wave queued its DMA requests and waited for its first fragments.
// Here the waves form two groups (waves 0-3 / 4-7 = one wave per SIMD each) that run ONE BARRIER APART: while group A issues the
// MFMAs of a k-phase, group B issues its fragment reads and its share of the next K-tile's DMA, then they swap (the 8-phase /
// ping-pong schedule of the CDNA4 guide, "256^2 8-phase template", on this kernel's 256 x 320 tile and 32x32x16 MFMA): +4-10 % at
// K >= 1280, equal at K = 320 / 640 and for the convolutions, bit-identical outputs (profiles/README.md, round 4).  Per phase and group:
//
//     L section:  ds_read_b128 fragments of this phase | LDS-DMA pieces of the next K-tile | s_waitcnt lgkmcnt(0)
//     s_barrier (B1)
//     M section:  s_setprio 1 | 10 (NB = 5) MFMAs of one k-step (16 of K) | s_setprio 0
//     s_barrier (B2)
//
// Group 1 executes one extra barrier in front of every tile, so its L section coincides with group 0's M section and vice
// versa; group 0 executes one extra barrier behind the tile's last phase, so both groups run the epilogue together.
// Hazards (P(i) = i-th barrier instance of the workgroup; group 0's phase j has B1 = P(2j+1), B2 = P(2j+2); group 1's one later):
//   * fragment reads complete (lgkmcnt(0)) before the reading wave's B1, so after the barrier that follows the last phase of a
//     K-tile for BOTH groups its stage may be refilled: the next-but-one K-tile's DMA starts in the first L section of the next K-tile;
//   * a wave's DMA pieces are retired by its own s_waitcnt vmcnt(0) in front of the last barrier instance before the first read
//     of that stage: group 0 waits at the end of the last M section of the K-tile (before B2), group 1 at the end of its last L
//     section (before B1) — the same barrier instance;
//   * the epilogue's LDS transposition buffers alias the stage consumed last; every wave is out of the epilogue before group 0
//     passes B1 of the next tile's first phase, so the first K-tile of a tile issues its DMA one phase later than the others.
#include "gemm_common.h"

namespace {

constexpr int PBM = 256;
template <int NB> struct PPCfg {
  static constexpr int BN = NB * 64;
  static constexpr int XBYTES = PBM * 128;
  static constexpr int WBYTES = BN * 128;
  static constexpr int EPI_BYTES = 8 * 32 * 68 * 4;
  static constexpr int STAGE = (XBYTES + WBYTES) > EPI_BYTES ? (XBYTES + WBYTES) : EPI_BYTES;
  static constexpr int BIAS_OFF = 2 * STAGE;
  static constexpr int BIAS_STRIDE = 2048;
  static constexpr int SMEM = 2 * STAGE + 2 * BIAS_STRIDE;
};

// Tile order: groups of PP_GM rows of tiles, column-major inside a group, so that the 32 consecutive tile ids an XCD works on at
// any time are an 8 x 4 block (8 A panels + 4 W panels per K-tile through that XCD's L2) instead of one row of up to 32 different
// W panels — the wide projections (N >= 2560: 8-40 column tiles) otherwise stream all of W through every L2 once per row of tiles.
#ifndef A3D_PP_GM
#define A3D_PP_GM 8
#endif
constexpr int PP_GM = A3D_PP_GM;
A3D_DEV void pp_tile_coords(int64_t t, int64_t tiles_m, int64_t tiles_n, int64_t& tile_m, int64_t& tile_n) {
  const int64_t gsz = PP_GM * tiles_n;
  const int64_t blk = t / gsz;
  const int64_t first = blk * PP_GM;
  const int64_t gm = tiles_m - first < PP_GM ? tiles_m - first : PP_GM;
  const int64_t within = t - blk * gsz;
  tile_n = within / gm;
  tile_m = first + (within - tile_n * gm);
}

A3D_DEV void pp_barrier() {
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// Four phases (k-steps of 16) per K-tile; the 4 + NB DMA pieces a wave issues per K-tile go three per phase into phases 0-2 (measured
// against five-then-the-rest and against two k-steps per phase: profiles/r4_microbench_pp.log).
// CONV: 0 = dense A, 1 = 3x3 conv gather (pad 1, stride 1|2), 2 = 3x3 conv over a nearest-2x upsampled input
// SPLIT: split-K work items (GemmParams::ksplit > 1) — a separate instantiation: the item bookkeeping and the fp32 partial stores cost the
// unsplit kernels registers they do not have (the conv instantiations sit at 256)
// TWO: two-source A operand (GemmParams::X2; dense only) — its own instantiation for the same reason
// DIRECT: W rows staged in the permuted order of direct_epilogue (gemm_common.h) and that epilogue instead of the LDS transposition (dense linear
// epilogues only; A3D_GEMM_DIRECT in the call's flags word)
template <int CONV, int EPI, int NB, bool RES, bool SPLIT = false, bool TWO = false, bool DIRECT = false>
__global__ __launch_bounds__(512, 1) void gemm_pp_kernel(const GemmParams p) {
  static_assert(!TWO || CONV == 0, "the two-source A operand is a dense-GEMM feature");
  static_assert(!DIRECT || (CONV == 0 && EPI == EPI_LINEAR && !SPLIT && !TWO), "the direct epilogue is a dense linear-epilogue feature");
  using PC = PPCfg<NB>;
  constexpr int PH = 1, NPH = 4;              // k-steps per phase, phases per K-tile
  constexpr int NP = 4 + NB;                  // DMA pieces per wave and K-tile: X 0..3, W 0..NB-1
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  char* const smem_b = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int grp = wid >> 2;                            // ping-pong group
  const int l31 = lane & 31, g = lane >> 5;
  const int lr = lane >> 3, pos = lane & 7;
  const uint32_t lds0 = lds_addr(smem);

  // work items: one per output tile, or (split-K) `ksplit` consecutive items per tile, item s of a tile contracting K-tiles [s nk, (s + 1) nk)
  const int S = SPLIT ? p.ksplit : 1;
  const int64_t ntiles = p.tiles_m * p.tiles_n * S;
  const int64_t G = gridDim.x;
  int64_t t = xcd_remap(blockIdx.x, G);
  if (t >= ntiles) return;
  const int nk = SPLIT ? p.nk_item : (int)(p.K / 64);

  const uint32_t koff0 = (uint32_t)((g ^ ((l31 >> 1) & 7)) << 4);
  const uint32_t xrd = (uint32_t)(wm * 64 + l31) * 128u;
  const int wblk = (NB == 5) ? wn * 4 : wn * NB;
  const int wblk_last = (NB == 5) ? 8 + wn : wn * NB + NB - 1;
  const uint32_t wrd = (uint32_t)PC::XBYTES + (uint32_t)(wblk * 32 + l31) * 128u;
  const uint32_t wrd_last = (uint32_t)PC::XBYTES + (uint32_t)(wblk_last * 32 + l31) * 128u;

  const uint32_t vx0 = (uint32_t)(lr * p.ldx * 2 + ((pos ^ (lr >> 1)) << 4));
  // DIRECT: LDS row 8 pc + lr of the W image (MFMA row 8 b + 4 g + c of its 32-block: b = pc & 3, lr = 4 g + c) holds W row
  // 32 (pc >> 2) + 16 g + 4 b + c: the lane part of the source offset is (16 (lr >> 2) + (lr & 3)) rows, the piece part wro[i]
  const uint32_t vw0 = DIRECT ? (uint32_t)((16 * (lr >> 2) + (lr & 3)) * p.ldw * 2 + ((pos ^ (lr >> 1)) << 4))
                              : (uint32_t)(lr * p.ldw * 2 + ((pos ^ (lr >> 1)) << 4));
  uint32_t wro[DIRECT ? NB : 1];
  if constexpr (DIRECT) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int pc = wid * NB + i;
      wro[i] = (uint32_t)((32 * (pc >> 2) + 4 * (pc & 3)) * p.ldw * 2);
    }
  }
  uint32_t aoff[CONV ? 4 : 1];
  uint32_t amask[CONV ? 2 : 1];
  // scalar copies of the tap masks: bit set <=> ALL 8 pixels of the piece are inside the image for that tap.  Such a (piece, tap) — 90 % of them
  // at 64 x 64, 80 % at 32 x 32 — is one saddr + 32-bit-offset DMA like a dense piece; only pieces that touch the border select between
  // the image and the zero page per lane (7 VALU instructions each: the conv's L section carried 46 more VALU instructions per K-tile than
  // the dense kernel's, profiles/r5_conv_pmc.md)
  uint32_t sok[CONV ? 2 : 1];
  const int64_t cbias = CONV ? ((int64_t)p.Wd + 1) * p.Cin : 0;
  int64_t ld_m0 = 0, ld_n0 = 0;
  int ld_par = 0;
  int ik0 = 0, itap = 0, ikx = 0, iky = 0;
  // 3x3 conv: byte address of (tap, channel slice) of the K-tile requested next relative to a lane's tap-(0,0) offset aoff[], advanced
  // incrementally with the K walk (round 5: recomputing it from (tap / 3, tap % 3, slice) in each of the three DMA phases of a K-tile
  // was ~17 dependent SALU instructions per phase, and the zero page's address came back through the GOT — s_getpc + s_load_dwordx2 +
  // s_waitcnt lgkmcnt(0) BEHIND the phase's fragment reads — three times per K-tile: the L section was longer than the M section
  // it hides under).  The zero page's address is pinned in a scalar pair once per kernel.
  uint64_t xck = 0;
  uint64_t zpage = 0;
  if constexpr (CONV != 0) {
    zpage = (uint64_t)(uintptr_t)g_zero_page;
    asm volatile("" : "+s"(zpage));
  }
  // Scalar DMA bases of this wave's first X / W piece for the K-tile requested next, advanced by 128 bytes per K-tile; the other
  // pieces are a 32-bit offset away (i * 8 rows).  (Recomputing (row0 + 8 pc) * ld + k0 per piece cost ~20 dependent SALU
  // instructions per piece — 180 per wave and K-tile — in front of the first MFMA of every K-tile.)
  uint64_t xk = 0, wk = 0, rbk = 0;
  const uint32_t sx8 = (uint32_t)(p.ldx * 16), sw8 = (uint32_t)(p.ldw * 16);      // bytes between two pieces (8 rows)
  // second A source (TWO): its own per-lane offset, piece stride and running base; the K-tile at position ik0 >= K1 reads it
  uint64_t xk2 = 0;
  const uint32_t vx0b = TWO ? (uint32_t)(lr * p.ldx2 * 2 + ((pos ^ (lr >> 1)) << 4)) : 0u;
  const uint32_t sx8b = TWO ? (uint32_t)(p.ldx2 * 16) : 0u;
  auto setup_tile = [&](int64_t tt) {
    int64_t tile_n, tile_m;
    const int64_t tile_id = SPLIT ? (int64_t)((uint32_t)tt / (uint32_t)S) : tt;
    // K offset of this item in elements of a row of X / W (conv: whole 64-channel slices = 9 K-tiles each, the launcher makes nk a multiple of 9)
    const int64_t kofs = SPLIT ? (CONV ? ((int64_t)(tt - tile_id * S) * nk / 9) * 64 : (int64_t)(tt - tile_id * S) * nk * 64) : 0;
    pp_tile_coords(tile_id, p.tiles_m, p.tiles_n, tile_m, tile_n);
    ld_m0 = tile_m * PBM; ld_n0 = tile_n * PC::BN;
    ik0 = 0; itap = 0; ikx = 0; iky = 0;
    ld_par ^= 1;
    if constexpr (CONV != 0) xck = (uint64_t)(uintptr_t)p.X - (uint64_t)cbias * 2u + (uint64_t)kofs * 2u;
    if constexpr (CONV == 0) xk = (uint64_t)(uintptr_t)(p.X + (ld_m0 + wid * 32) * p.ldx + kofs);
    if constexpr (TWO) xk2 = (uint64_t)(uintptr_t)(p.X2 + (ld_m0 + wid * 32) * p.ldx2);
    wk = (uint64_t)(uintptr_t)(p.W + (ld_n0 + (DIRECT ? 0 : wid * (NB * 8))) * p.ldw + kofs);
    if (EPI == EPI_LINEAR && p.rowbias) rbk = (uint64_t)(uintptr_t)(p.rowbias + (ld_m0 / p.rb_div) * p.N + ld_n0);
    if constexpr (CONV != 0) {
      amask[0] = 0; amask[1] = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = (wid * 4 + i) * 8 + lr;
        const int slot = pos ^ ((r >> 1) & 7);
        // 32-bit divisions: the launcher admits conv inputs below 4 GB only (32-bit DMA offsets), so M < 2^25
        const uint32_t m = (uint32_t)ld_m0 + (uint32_t)r;
        const uint32_t hw = (uint32_t)(p.Ho * p.Wo);
        const int b = (int)(m / hw);
        const int rem = (int)(m - (uint32_t)b * hw);
        const int oy = (int)((uint32_t)rem / (uint32_t)p.Wo), ox = rem - oy * p.Wo;
        uint32_t mask = 0;
        if constexpr (CONV == 2) {
          const int sy0 = (oy - 1) >> 1, sx0 = (ox - 1) >> 1;
          aoff[i] = (uint32_t)(((((int64_t)b * p.H + sy0) * p.Wd + sx0) * p.Cin + cbias) * 2 + slot * 16);
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) {
            const int yy = oy + tp / 3 - 1, xx = ox + tp % 3 - 1;
            if (yy >= 0 && yy < p.He && xx >= 0 && xx < p.We) mask |= 1u << tp;
          }
          amask[i >> 1] |= (mask << (9 * (i & 1))) | ((uint32_t)(~oy & 1) << (18 + 2 * (i & 1))) | ((uint32_t)(~ox & 1) << (19 + 2 * (i & 1)));
        } else {
          const int y0 = oy * p.stride - 1, x0 = ox * p.stride - 1;
          aoff[i] = (uint32_t)(((((int64_t)b * p.H + y0) * p.Wd + x0) * p.Cin + cbias) * 2 + slot * 16);
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) {
            const int yy = y0 + tp / 3, xx = x0 + tp % 3;
            if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd) mask |= 1u << tp;
          }
          amask[i >> 1] |= mask << (9 * (i & 1));
        }
      }
      sok[0] = 0; sok[1] = 0;
#pragma unroll
      for (int bit = 0; bit < 18; ++bit) {
        if (__builtin_amdgcn_ballot_w64(((amask[0] >> bit) & 1u) != 0u) == ~0ull) sok[0] |= 1u << bit;
        if (__builtin_amdgcn_ballot_w64(((amask[1] >> bit) & 1u) != 0u) == ~0ull) sok[1] |= 1u << bit;
      }
    }
  };
  // DMA pieces [A, B) of the K-tile at the running position (ik0 / itap / xck) into stage buf; the position moves on with the last piece
  auto issue_pieces = [&](int buf, auto a_c, auto b_c) __attribute__((always_inline)) {
    constexpr int A = decltype(a_c)::value, B = decltype(b_c)::value;
    if constexpr (A >= B) return;
    const uint32_t dst = lds0 + (uint32_t)buf * PC::STAGE;
    if constexpr (A == 0) {
      if (ik0 == 0) {
        const uint32_t bdst = lds0 + (uint32_t)PC::BIAS_OFF + (uint32_t)ld_par * PC::BIAS_STRIDE;
        if (p.bias) {
          if (wid == 0) glds16_s((uint32_t)lane * 16u, p.bias + ld_n0, bdst);
          if (NB == 5 && wid == 1) { if (lane < 16) glds16_s((uint32_t)lane * 16u, p.bias + ld_n0 + 256, bdst + 1024u); }
        }
        if (EPI == EPI_LINEAR && p.rowbias && wid == 2) {
          if (lane < PC::BN / 8) glds16_s((uint32_t)lane * 16u, (const void*)(uintptr_t)rbk, bdst + 1280u);
        }
      }
    }
    if constexpr (CONV != 0) {
      const int ky = iky, kx = ikx;
      const uint32_t rowb = (uint32_t)(p.Wd * p.Cin * 2), colb = (uint32_t)(p.Cin * 2);
      const uint32_t yE = (uint32_t)((ky + 1) >> 1) * rowb, yO = (uint32_t)(ky >> 1) * rowb;
      const uint32_t xE = (uint32_t)((kx + 1) >> 1) * colb, xO = (uint32_t)(kx >> 1) * colb;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < A || i >= B) continue;
        const uint32_t d = dst + (uint32_t)(wid * 4 + i) * 1024u;
        const uint32_t mk = amask[i >> 1];
        uint32_t vo = aoff[i];
        if constexpr (CONV == 2) vo += (((mk >> (18 + 2 * (i & 1))) & 1u) ? yE : yO) + (((mk >> (19 + 2 * (i & 1))) & 1u) ? xE : xO);
        const int bit = itap + 9 * (i & 1);
        if ((sok[i >> 1] >> bit) & 1u) {
          glds16_s(vo, (const void*)(uintptr_t)xck, d);          // whole piece inside the image: scalar base + per-lane 32-bit offset
        } else {
          // one DMA instruction with per-lane 64-bit sources: in-image taps read X, the others the zero page (the two exec-masked
          // instructions of the lockstep kernel cost two branches per piece in the L section)
          const uint64_t src = ((mk >> bit) & 1u) ? xck + vo : zpage;
          glds16_v((const void*)(uintptr_t)src, d);
        }
      }
    } else {
      const bool sec = TWO && ik0 >= (int)p.K1;              // wave-uniform: this K-tile comes from the second source
      const uint32_t vxs = sec ? vx0b : vx0, sxs = sec ? sx8b : sx8;
      const uint64_t xks = sec ? xk2 : xk;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < A || i >= B) continue;
        const int pc = wid * 4 + i;
        glds16_s(vxs ^ (uint32_t)((i & 1) << 6), (const void*)(uintptr_t)(xks + (uint64_t)(uint32_t)(i * sxs)), dst + (uint32_t)pc * 1024u);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (4 + i < A || 4 + i >= B) continue;
      const int pc = wid * NB + i;
      const uint32_t wpo = DIRECT ? wro[DIRECT ? i : 0] : (uint32_t)(i * sw8);
      glds16_s(vw0 ^ (uint32_t)((pc & 1) << 6), (const void*)(uintptr_t)(wk + (uint64_t)wpo), dst + (uint32_t)PC::XBYTES + (uint32_t)pc * 1024u);
    }
    if constexpr (B == NP) {
      if constexpr (CONV != 0) {
        // K walk of the implicit GEMM: all nine taps of one 64-channel slice, then the next slice (W rows are [tap][Cin]: a tap is
        // Cin elements away).  Tap-major order touched a tile's whole input window (6 image rows x Cin) between two uses of the same
        // line — ~5 MB per XCD at level 0, more than its L2; this order re-reads a 32 KB slice nine times in a row.  gemm_kernel walks
        // the same order (conv_k0), so the two kernels stay bit-identical.
        const uint64_t cin2 = (uint64_t)(uint32_t)p.Cin * 2u;
        wk += cin2;
        ++itap;
        if (++ikx == 3) { ikx = 0; ++iky; }
        if (itap == 9) {                                  // next 64-channel slice, back to tap (0, 0)
          itap = 0; iky = 0; wk -= 9u * cin2 - 128u;
          if constexpr (CONV == 1) xck -= (2u * (uint64_t)(uint32_t)p.Wd + 2u) * cin2;
          xck += 128u;
        } else if constexpr (CONV == 1) {
          xck += ikx == 0 ? ((uint64_t)(uint32_t)p.Wd - 2u) * cin2 : cin2;
        }
      } else {
        if (TWO && ik0 >= (int)p.K1) xk2 += 128; else xk += 128;
        wk += 128;
      }
      ik0 += 64;
    }
  };
  // DMA share of phase ph of a K-tile; `late` = first K-tile of a tile (nothing in phase 0: the epilogue buffers; then five, then the rest)
  auto issue_phase = [&](int buf, auto ph_c, bool late) __attribute__((always_inline)) {
    constexpr int ph = decltype(ph_c)::value;
    using I0 = std::integral_constant<int, 0>;
    using I3 = std::integral_constant<int, 3>;
    using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>;
    using IN = std::integral_constant<int, NP>;
    if (!late) {
      if constexpr (ph == 0) issue_pieces(buf, I0{}, I3{});
      if constexpr (ph == 1) issue_pieces(buf, I3{}, I6{});
      if constexpr (ph == 2) issue_pieces(buf, I6{}, IN{});
    } else {
      if constexpr (ph == 1) issue_pieces(buf, I0{}, I5{});
      if constexpr (ph == 2) issue_pieces(buf, I5{}, IN{});
    }
  };

  f32x16_t acc[NB][2];
  u32x4_t fx[PH][2], fw[PH][NB];
  auto load_frags = [&](int buf, int ks0) __attribute__((always_inline)) {
    const char* xs = smem_b + buf * PC::STAGE + xrd;
    const char* ws = smem_b + buf * PC::STAGE + wrd;
    const char* wl = smem_b + buf * PC::STAGE + wrd_last;
#pragma unroll
    for (int s = 0; s < PH; ++s) {
      const uint32_t ko = koff0 ^ (uint32_t)((ks0 + s) << 5);
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) fx[s][tm] = *reinterpret_cast<const u32x4_t*>(xs + tm * 4096 + ko);
#pragma unroll
      for (int tn = 0; tn < NB - 1; ++tn) fw[s][tn] = *reinterpret_cast<const u32x4_t*>(ws + tn * 4096 + ko);
      fw[s][NB - 1] = *reinterpret_cast<const u32x4_t*>(wl + ko);
    }
  };
  auto mfma_phase = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < PH; ++s)
#pragma unroll
      for (int tn = 0; tn < NB; ++tn)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) acc[tn][tm] = mfma32(fw[s][tn], fx[s][tm], acc[tn][tm]);
  };

  // the epilogue reads the per-tile bias (fp32) and rowbias (16-bit) images from LDS unconditionally: a launch without one of them
  // zero-fills both parities of the image once (+0.0 is what the 128 x 128 kernel adds for an absent bias, too)
  if (!p.bias || !(EPI == EPI_LINEAR && p.rowbias)) {
    uint32_t* const bz = reinterpret_cast<uint32_t*>(smem_b + PC::BIAS_OFF);
    for (int i = tid; i < 2 * PC::BIAS_STRIDE / 4; i += 512) {
      const int o = (i * 4) % PC::BIAS_STRIDE;                 // byte offset inside one parity: [0, 1280) bias, [1280, 1920) rowbias
      if ((o < 1280 && !p.bias) || (o >= 1280 && !(EPI == EPI_LINEAR && p.rowbias))) bz[i] = 0u;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    pp_barrier();
  }
  // prologue: the first K-tile of the first tile, complete for everybody
  setup_tile(t);
  issue_pieces(0, std::integral_constant<int, 0>{}, std::integral_constant<int, NP>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  pp_barrier();
  int buf = 0;
  for (;;) {
    int64_t tile_n, tile_m;
    pp_tile_coords(SPLIT ? (int64_t)((uint32_t)t / (uint32_t)S) : t, p.tiles_m, p.tiles_n, tile_m, tile_n);
    const int64_t m0 = tile_m * PBM, n0 = tile_n * PC::BN;
    const int64_t tnext = t + G;
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int cur_par = ld_par;
    if (grp) pp_barrier();                         // group 1 runs one barrier behind group 0
    for (int kt = 0; kt < nk; ++kt) {
      const bool last = kt + 1 >= nk;
      const bool more = !last || tnext < ntiles;   // a K-tile follows this one (this tile's next or the next tile's first)
      const bool late = kt == 0;
      static_for<NPH>([&](auto ph_c) __attribute__((always_inline)) {
        constexpr int ph = decltype(ph_c)::value;
        // ---- L section
        load_frags(buf, ph * PH);
        if (more) {
          if constexpr (ph == 0 || ph == 1) {
            // the next tile's first K-tile: set the tile up right before its first piece is issued
            if (last && (late ? ph == 1 : ph == 0)) setup_tile(tnext);
          }
          issue_phase(buf ^ 1, ph_c, late);
          if constexpr (ph == NPH - 1) { if (grp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pp_barrier();                              // B1
        // ---- M section
        __builtin_amdgcn_s_setprio(1);
        mfma_phase();
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ph == NPH - 1) { if (more && !grp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        pp_barrier();                              // B2
      });
      buf ^= 1;
    }
    if (!grp) pp_barrier();                        // group 0 waits for group 1's last phase: both run the epilogue together

    if constexpr (SPLIT) {
      // split-K item: the fp32 accumulators leave in register order (1 KB per store instruction, no LDS); splitk_reduce_kernel reads them back
      float* const wsi = p.ws + ((t * 8 + wid) * (NB * 8)) * 256 + lane * 4;
#pragma unroll
      for (int tn = 0; tn < NB; ++tn)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(wsi + ((tn * 2 + tm) * 4 + q) * 256) =
                float4{acc[tn][tm][4 * q], acc[tn][tm][4 * q + 1], acc[tn][tm][4 * q + 2], acc[tn][tm][4 * q + 3]};
    } else if constexpr (DIRECT) {
      direct_epilogue<NB, RES>(p, acc, reinterpret_cast<const float*>(smem_b + PC::BIAS_OFF + cur_par * PC::BIAS_STRIDE),
                               reinterpret_cast<const uint16_t*>(smem_b + PC::BIAS_OFF + cur_par * PC::BIAS_STRIDE + 1280),
                               m0, n0, wm, wblk, wblk_last, lane);
    } else {
      persist_epilogue<EPI, NB, RES>(p, acc, reinterpret_cast<float*>(smem_b + (buf ^ 1) * PC::STAGE) + wid * (32 * 68),
                                     reinterpret_cast<const float*>(smem_b + PC::BIAS_OFF + cur_par * PC::BIAS_STRIDE),
                                     reinterpret_cast<const uint16_t*>(smem_b + PC::BIAS_OFF + cur_par * PC::BIAS_STRIDE + 1280),
                                     m0, n0, wm, wblk, wblk_last, lane);
    }
    if (tnext >= ntiles) break;
    t = tnext;
  }
}

// Second half of a split-K launch: per output tile, add the `ksplit` fp32 slices in index order (deterministic) and apply the epilogue's
// arithmetic — ((acc + bias) + rowbias) * alpha, then + beta * R, one rounding — in the layout the accumulators were stored in: thread =
// (wave, lane) of the producing workgroup, 8-byte stores of 4 consecutive columns.
template <int NB>
__global__ __launch_bounds__(512) void splitk_reduce_kernel(const GemmParams p) {
  using PC = PPCfg<NB>;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, g = lane >> 5;
  const int wblk = (NB == 5) ? wn * 4 : wn * NB;
  const int wblk_last = (NB == 5) ? 8 + wn : wn * NB + NB - 1;
  const int S = p.ksplit;
  int64_t tile_n, tile_m;
  pp_tile_coords(blockIdx.x, p.tiles_m, p.tiles_n, tile_m, tile_n);
  const int64_t m0 = tile_m * PBM, n0 = tile_n * PC::BN;
  const float* const w0 = p.ws + (((int64_t)blockIdx.x * S * 8 + wid) * (NB * 8)) * 256 + lane * 4;
  const int64_t item_stride = (int64_t)8 * NB * 8 * 256;
  // grid.y = NB: one workgroup per (tile, 32-column block of every wave) — a launch is 32-160 tiles, far too few workgroups to pull fp32 partials
  // at HBM rate with one workgroup per tile (58 us per reduce at level 3, 1.8 TB/s)
  const int tn = blockIdx.y;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int64_t m = m0 + wm * 64 + tm * 32 + l31;
      const int64_t nb0 = n0 + (tn < NB - 1 ? wblk + tn : wblk_last) * 32 + 4 * g;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* src = w0 + ((tn * 2 + tm) * 4 + q) * 256;
        float4 a = *reinterpret_cast<const float4*>(src);
        for (int s = 1; s < S; ++s) {
          const float4 b = *reinterpret_cast<const float4*>(src + s * item_stride);
          a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        const int64_t n = nb0 + 8 * q;
        float v[4] = {a.x, a.y, a.z, a.w};
        if (p.bias) {
          const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (p.rowbias) {
          const u32x2_t tb = *reinterpret_cast<const u32x2_t*>(p.rowbias + (m / p.rb_div) * p.N + n);
          v[0] += lo16(tb[0]); v[1] += hi16(tb[0]); v[2] += lo16(tb[1]); v[3] += hi16(tb[1]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = epi_scale(v[e], p.alpha);
        if (p.R) {
          const u32x2_t tr = *reinterpret_cast<const u32x2_t*>(p.R + m * p.ldr + n);
          v[0] = epi_axpy(v[0], p.beta, lo16(tr[0])); v[1] = epi_axpy(v[1], p.beta, hi16(tr[0]));
          v[2] = epi_axpy(v[2], p.beta, lo16(tr[1])); v[3] = epi_axpy(v[3], p.beta, hi16(tr[1]));
        }
        u32x2_t o;
        o[0] = pack16(v[0], v[1]);
        o[1] = pack16(v[2], v[3]);
        *reinterpret_cast<u32x2_t*>(p.Y + m * p.ldy + n) = o;
      }
    }
}

template <int CONV, int EPI, int NB, bool RES, bool SPLIT = false, bool TWO = false, bool DIRECT = false>
int launch_pp(hipStream_t stream, const GemmParams& p, int cus) {
  using PC = PPCfg<NB>;
  static uint64_t attr_done = 0;
  if (int rc = a3d_once_per_device(attr_done, [] {
        return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<CONV, EPI, NB, RES, SPLIT, TWO, DIRECT>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, PC::SMEM); })) return rc;
  const int64_t ntiles = p.tiles_m * p.tiles_n * (SPLIT ? p.ksplit : 1);
  const unsigned grid = (unsigned)(ntiles < cus ? ntiles : cus);
  gemm_pp_kernel<CONV, EPI, NB, RES, SPLIT, TWO, DIRECT><<<dim3(grid), dim3(512), PC::SMEM, stream>>>(p);
  if (int rc = a3d_launch_status()) return rc;
  if constexpr (SPLIT) {
    splitk_reduce_kernel<NB><<<dim3((unsigned)(p.tiles_m * p.tiles_n), NB), dim3(512), 0, stream>>>(p);
    return a3d_launch_status();
  }
  return 0;
}

template <int CONV>
int launch_pp_conv(int epi, int nb, hipStream_t stream, const GemmParams& p, int cus) {
  if (epi == EPI_GEGLU) {
    if constexpr (CONV == 0) { if (nb == 4) return launch_pp<0, EPI_GEGLU, 4, false>(stream, p, cus); }
    return A3D_EUNSUPPORTED;
  }
  if constexpr (CONV == 0) {
    if (p.X2 != nullptr) {     // two-source A operand: plain linear epilogue (bias), no split
      if (p.R || p.ksplit > 1) return A3D_EUNSUPPORTED;
      if (nb == 5) return launch_pp<0, EPI_LINEAR, 5, false, false, true>(stream, p, cus);
      if (nb == 4) return launch_pp<0, EPI_LINEAR, 4, false, false, true>(stream, p, cus);
      return A3D_EUNSUPPORTED;
    }
  }
  if (p.ksplit > 1) {        // (the residual is applied by the reduce kernel)
    if (nb == 5) return launch_pp<CONV, EPI_LINEAR, 5, false, true>(stream, p, cus);
    if (nb == 4) return launch_pp<CONV, EPI_LINEAR, 4, false, true>(stream, p, cus);
    return A3D_EUNSUPPORTED;
  }
  if constexpr (CONV == 0) {
    if (p.direct) {            // direct epilogue (A3D_GEMM_DIRECT)
      if (nb == 5) return p.R ? launch_pp<0, EPI_LINEAR, 5, true, false, false, true>(stream, p, cus) : launch_pp<0, EPI_LINEAR, 5, false, false, false, true>(stream, p, cus);
      if (nb == 4) return p.R ? launch_pp<0, EPI_LINEAR, 4, true, false, false, true>(stream, p, cus) : launch_pp<0, EPI_LINEAR, 4, false, false, false, true>(stream, p, cus);
      return A3D_EUNSUPPORTED;
    }
  }
  if (nb == 5) return p.R ? launch_pp<CONV, EPI_LINEAR, 5, true>(stream, p, cus) : launch_pp<CONV, EPI_LINEAR, 5, false>(stream, p, cus);
  if (nb == 4) return p.R ? launch_pp<CONV, EPI_LINEAR, 4, true>(stream, p, cus) : launch_pp<CONV, EPI_LINEAR, 4, false>(stream, p, cus);
  return A3D_EUNSUPPORTED;
}

}  // namespace

int A3D_FN(a3d_launch_gemm_pp)(int conv, int epi, int nb, hipStream_t stream, const GemmParams& p, int cus) {
  if (conv == 0) return launch_pp_conv<0>(epi, nb, stream, p, cus);
  if (conv == 1) return launch_pp_conv<1>(epi, nb, stream, p, cus);
  return launch_pp_conv<2>(epi, nb, stream, p, cus);
}
