// Dense GEMM for the SMALL token matrices (round 6): M = 1024 ... 16384 rows — UNet levels 2 / 3 of a multi-GPU rank, the whole 4D-SDS
// shape, the training step's level-1-3 linears — where the persistent 256 x 320 kernel (gemm_pp.hip) has too few tiles to fill the chip and
// the shapes used to fall back to the register-staged 128 x 128 kernel (gemm_conv.hip), whose two K-tiles of prefetch per workgroup
// leave a lone workgroup per CU waiting one HBM latency per two K-tiles: 1.5 us per K-tile at M <= 4096, 105-350 TFLOP/s
// (profiles/r6_microbench_smallm_before.log).
//
// Same K order and epilogue arithmetic (bit-identical to both other kernels), but the operands go
// global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers) into a RING of four 32 KB K-tile stages, three K-tiles
// (96 KB per CU) in flight ahead of the one being multiplied, and the K-tile stream runs across the tiles of a persistent workgroup (the
// next tile's first K-tiles land during the epilogue).  256 threads = 4 waves as 2 (M) x 2 (N), one per SIMD, each 64 x 64 = 2 x 2 MFMA
// 32x32x16 tiles; fragments double-buffered in registers (the reads of k-step s + 1 are issued before the MFMAs of k-step s).
// LDS image and swizzle as in gemm_pp.hip (lane-linear 8-row DMA pieces, 16-byte chunk index ^= (row >> 1) & 7 on the source address
// and on the fragment read).  One barrier per K-tile; waits are COUNTED (vmcnt(8 x K-tiles issued behind the one needed)): loads retire
// in order, so "everything but the newest n" always covers the pieces of the K-tile about to be read, whatever else (bias pieces, the
// previous epilogue's stores) is in the queue.
//
// Hazards.  Stage s = (global K-tile index) & 3.  Iteration G: wait for own pieces of K-tile G | barrier (a: everybody's pieces of G have
// landed; b: everybody is done reading stage (G - 1) & 3) | request K-tile G + 3 into stage (G + 3) & 3 = (G - 1) & 3 | fragments + MFMAs
// of stage G & 3.  Tile end: barrier (everybody is done with the last stage), epilogue staged through that stage (4 x 8704 bytes = one
// stage incl. its 2 KB pad), whose refill is requested after the next iteration's barrier.
#include "gemm_common.h"

namespace {

// Tile widths (round 6, second step).  What bounds a lone workgroup per CU is the CU's LDS-DMA fill rate, ~60-90 GB/s: a K-tile of a
// 128 x 128 tile (32 KB) takes ~0.5 us whatever the matrix pipe could do, so the time of a launch is rounds x (128 + BN) x K / rate and
// the best tile is the WIDEST one that still gives every CU at most one tile: NB = 2 (128 x 128, ring of 4), NB = 4 (128 x 256, ring of 3),
// NB = 5 (128 x 320, ring of 2) — chosen per shape by plan_ring (gemm_conv.hip); the wave layout for NB = 5 is gemm_pp.hip's (four
// contiguous 32-column blocks per wave plus one of the last two).
constexpr int RBM = 128;
constexpr int R_XBYTES = RBM * 128;
template <int NB> struct RingCfg {
  static constexpr int BN = NB * 64;
  static constexpr int WBYTES = BN * 128;
  static constexpr int EPI = 4 * 32 * 68 * 4;                  // 34 816: the epilogue's four wave-private fp32 buffers
  static constexpr int STAGE = (R_XBYTES + WBYTES) > EPI ? (R_XBYTES + WBYTES) : EPI;
  static constexpr int RING = NB == 2 ? 4 : (NB == 4 ? 3 : 2);
  static constexpr int AHEAD = RING - 1;
  static constexpr int BIAS_OFF = RING * STAGE;
  static constexpr int BIAS_STRIDE = 2048;                     // per parity: fp32 bias [0, 1280), 16-bit rowbias row [1280, 1920)
  static constexpr int SMEM = BIAS_OFF + 2 * BIAS_STRIDE;
  static constexpr int NPW = 2 * NB;                           // W pieces per wave and K-tile (X: 4)
  static constexpr int NP = 4 + NPW;
  static_assert(SMEM <= 160 * 1024, "LDS");
};

A3D_DEV void ring_barrier() {
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <int NB, bool RES>
__global__ __launch_bounds__(256, 1) void gemm_ring_kernel(const GemmParams p) {
  using RC = RingCfg<NB>;
  constexpr int RBN = RC::BN, R_STAGE = RC::STAGE, R_RING = RC::RING, R_AHEAD = RC::AHEAD, R_BIAS_OFF = RC::BIAS_OFF, R_BIAS_STRIDE = RC::BIAS_STRIDE;
  constexpr int NPW = RC::NPW, NP = RC::NP;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  char* const smem_b = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, g = lane >> 5;
  const int lr = lane >> 3, pos = lane & 7;
  const uint32_t lds0 = lds_addr(smem);

  const int64_t ntiles = p.tiles_m * p.tiles_n;
  const int64_t G = gridDim.x;
  const int64_t t_first = xcd_remap(blockIdx.x, G);
  if (t_first >= ntiles) return;
  const int nk = (int)(p.K / 64);
  const int64_t my_tiles = (ntiles - t_first + G - 1) / G;
  const int64_t total_kt = my_tiles * nk;                       // K-tiles this workgroup consumes

  // fragment read offsets (see gemm_pp.hip)
  const uint32_t koff0 = (uint32_t)((g ^ ((l31 >> 1) & 7)) << 4);
  const uint32_t xrd = (uint32_t)(wm * 64 + l31) * 128u;
  const int wblk = (NB == 5) ? wn * 4 : wn * NB;
  const int wblk_last = (NB == 5) ? 8 + wn : wn * NB + NB - 1;
  const uint32_t wrd = (uint32_t)R_XBYTES + (uint32_t)(wblk * 32 + l31) * 128u;
  const uint32_t wrd_last = (uint32_t)R_XBYTES + (uint32_t)(wblk_last * 32 + l31) * 128u;
  // DMA source offsets of a lane inside a piece (8 rows x 128 B, chunk index swizzled by (row >> 1) & 7; row = 8 i + lr)
  const uint32_t vx0 = (uint32_t)(lr * p.ldx * 2 + ((pos ^ (lr >> 1)) << 4));
  const uint32_t vw0 = (uint32_t)(lr * p.ldw * 2 + ((pos ^ (lr >> 1)) << 4));
  const uint32_t sx8 = (uint32_t)(p.ldx * 16), sw8 = (uint32_t)(p.ldw * 16);      // bytes between two pieces (8 rows)

  // ---- request stream: K-tile `iq` (global index over this workgroup's tiles) of tile `it`
  int64_t it = t_first;                  // tile the next requested K-tile belongs to
  int ik = 0;                            // its K-tile index inside that tile
  int64_t iq = 0;                        // global index of the next requested K-tile
  int ipar = 0;                          // bias image parity of tile `it`
  uint64_t xk = 0, wk = 0;
  auto tile_mn = [&](int64_t t, int64_t& tm_, int64_t& tn_) { tn_ = t % p.tiles_n; tm_ = t / p.tiles_n; };
  auto setup_request = [&]() {
    int64_t tm_, tn_;
    tile_mn(it, tm_, tn_);
    xk = (uint64_t)(uintptr_t)(p.X + (tm_ * RBM + wid * 32) * p.ldx);
    wk = (uint64_t)(uintptr_t)(p.W + (tn_ * RBN + wid * (NPW * 8)) * p.ldw);
  };
  uint32_t istage = 0;                   // ring slot of the next requested K-tile (iq % R_RING)
  auto request = [&]() __attribute__((always_inline)) {       // one K-tile: NP pieces per wave (+ the tile's bias / rowbias rows with its first)
    const uint32_t dst = lds0 + istage * (uint32_t)R_STAGE;
    istage = istage + 1 == (uint32_t)R_RING ? 0u : istage + 1;
    if (ik == 0) {
      int64_t tm_, tn_;
      tile_mn(it, tm_, tn_);
      const uint32_t bdst = lds0 + (uint32_t)R_BIAS_OFF + (uint32_t)ipar * R_BIAS_STRIDE;
      if (p.bias) {
        if (wid == 0) { if (lane < (RBN < 256 ? RBN / 4 : 64)) glds16_s((uint32_t)lane * 16u, p.bias + tn_ * RBN, bdst); }
        if (NB == 5 && wid == 2) { if (lane < 16) glds16_s((uint32_t)lane * 16u, p.bias + tn_ * RBN + 256, bdst + 1024u); }
      }
      if (p.rowbias && wid == 1) {
        if (lane < RBN / 8) glds16_s((uint32_t)lane * 16u, p.rowbias + ((tm_ * RBM) / p.rb_div) * p.N + tn_ * RBN, bdst + 1280u);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      glds16_s(vx0 ^ (uint32_t)((i & 1) << 6), (const void*)(uintptr_t)(xk + (uint64_t)(uint32_t)(i * sx8)), dst + (uint32_t)(wid * 4 + i) * 1024u);
#pragma unroll
    for (int i = 0; i < NPW; ++i)
      glds16_s(vw0 ^ (uint32_t)((i & 1) << 6), (const void*)(uintptr_t)(wk + (uint64_t)(uint32_t)(i * sw8)),
               dst + (uint32_t)R_XBYTES + (uint32_t)(wid * NPW + i) * 1024u);
    xk += 128; wk += 128;
    ++iq;
    if (++ik == nk) { ik = 0; it += G; ipar ^= 1; if (iq < total_kt) setup_request(); }
  };

  // launches without a bias / rowbias: the epilogue reads the images unconditionally (zero-filled once)
  if (!p.bias || !p.rowbias) {
    uint32_t* const bz = reinterpret_cast<uint32_t*>(smem_b + R_BIAS_OFF);
    for (int i = tid; i < 2 * R_BIAS_STRIDE / 4; i += 256) {
      const int o = (i * 4) % R_BIAS_STRIDE;
      if ((o < 1280 && !p.bias) || (o >= 1280 && !p.rowbias)) bz[i] = 0u;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ring_barrier();
  }

  setup_request();
#pragma unroll 1
  for (int i = 0; i < R_AHEAD; ++i) { if (iq < total_kt) request(); }

  f32x16_t acc[NB][2];
  u32x4_t fx[2][2], fw[2][NB];
  auto load_frags = [&](uint32_t stage_off, int ks, int fb) __attribute__((always_inline)) {
    const char* xs = smem_b + stage_off + xrd;
    const char* ws = smem_b + stage_off + wrd;
    const char* wl = smem_b + stage_off + wrd_last;
    const uint32_t ko = koff0 ^ (uint32_t)(ks << 5);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) fx[fb][tm] = *reinterpret_cast<const u32x4_t*>(xs + tm * 4096 + ko);
#pragma unroll
    for (int tn = 0; tn < NB - 1; ++tn) fw[fb][tn] = *reinterpret_cast<const u32x4_t*>(ws + tn * 4096 + ko);
    fw[fb][NB - 1] = *reinterpret_cast<const u32x4_t*>(wl + ko);
  };

  int64_t cq = 0;                         // global index of the K-tile being consumed
  uint32_t cstage = 0;                    // its ring slot
  int cpar = 0;
  for (int64_t t = t_first; t < ntiles; t += G) {
    int64_t tile_m, tile_n;
    tile_mn(t, tile_m, tile_n);
    const int64_t m0 = tile_m * RBM, n0 = tile_n * RBN;
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    uint32_t last_off = 0;
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt, ++cq) {
      // K-tiles requested behind the one needed now: their pieces may stay in flight
      const int64_t behind = iq - 1 - cq;
      if (R_AHEAD >= 3 && behind >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(2 * NP) : "memory");
      else if (R_AHEAD >= 2 && behind >= 1) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(NP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ring_barrier();
      if (iq < total_kt) request();
      const uint32_t stage = cstage * (uint32_t)R_STAGE;
      last_off = stage;
      cstage = cstage + 1 == (uint32_t)R_RING ? 0u : cstage + 1;
      load_frags(stage, 0, 0);
      static_for<4>([&](auto s_c) __attribute__((always_inline)) {
        constexpr int s = decltype(s_c)::value;
        if constexpr (s + 1 < 4) load_frags(stage, s + 1, (s + 1) & 1);
        if constexpr (s + 1 < 4) asm volatile("s_waitcnt lgkmcnt(%0)" :: "i"(2 + NB) : "memory");      // the reads of k-step s have returned (in order)
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tn = 0; tn < NB; ++tn)
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) acc[tn][tm] = mfma32(fw[s & 1][tn], fx[s & 1][tm], acc[tn][tm]);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    ring_barrier();                        // every wave is done with the last stage: the epilogue stages through it
    persist_epilogue<EPI_LINEAR, NB, RES>(p, acc, reinterpret_cast<float*>(smem_b + last_off) + wid * (32 * 68),
                                          reinterpret_cast<const float*>(smem_b + R_BIAS_OFF + cpar * R_BIAS_STRIDE),
                                          reinterpret_cast<const uint16_t*>(smem_b + R_BIAS_OFF + cpar * R_BIAS_STRIDE + 1280),
                                          m0, n0, wm, wblk, wblk_last, lane);
    cpar ^= 1;
  }
}

template <int NB, bool RES>
int launch_ring(hipStream_t stream, const GemmParams& p, int cus) {
  using RC = RingCfg<NB>;
  static uint64_t attr_done = 0;
  if (int rc = a3d_once_per_device(attr_done, [] {
        return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring_kernel<NB, RES>), hipFuncAttributeMaxDynamicSharedMemorySize, RC::SMEM); }))
    return rc;
  const int64_t ntiles = p.tiles_m * p.tiles_n;
  const unsigned grid = (unsigned)(ntiles < cus ? ntiles : cus);
  gemm_ring_kernel<NB, RES><<<dim3(grid), dim3(256), RC::SMEM, stream>>>(p);
  return a3d_launch_status();
}

}  // namespace

// caller (gemm_conv.hip) has checked the shape: M % 128 == 0, N % (64 nb) == 0, K % 64 == 0, K >= 256, 16-byte accesses, 32-bit DMA offsets,
// and filled tiles_m = M / 128, tiles_n = N / (64 nb)
int A3D_FN(a3d_launch_gemm_ring)(int nb, hipStream_t stream, const GemmParams& p, int cus) {
  switch (nb) {
    case 2: return p.R ? launch_ring<2, true>(stream, p, cus) : launch_ring<2, false>(stream, p, cus);
    case 4: return p.R ? launch_ring<4, true>(stream, p, cus) : launch_ring<4, false>(stream, p, cus);
    case 5: return p.R ? launch_ring<5, true>(stream, p, cus) : launch_ring<5, false>(stream, p, cus);
    default: return A3D_EINVAL;
  }
}
