// Two-workgroups-per-CU persistent GEMM for the short-K projections (K = 320 / 640: level-0 / level-1 linears, fused GEGLU).
//
// Why a second kernel: with 5-10 K-tiles per output tile the epilogue (transposition through LDS, bias / GELU arithmetic, 160-KB
// of stores per CU) is 25-40 % of a tile's time in gemm_persist_kernel / gemm_pp_kernel, and their one workgroup per CU runs it
// with the matrix pipe idle (ablations in profiles/README.md, round 4: without any store the K = 320, N = 1280 launch still takes
// 0.48 ms against 0.34 ms of matrix work).  Here a CU holds TWO independent 256-thread workgroups (4 waves each, one per SIMD,
// 256 registers per wave), each walking its own list of 128 x (NB*64) tiles: nothing synchronises them, so they drift half a
// tile apart and one's epilogue runs under the other's MFMAs.  Price: the W K-tile is staged once per 128 rows instead of once
// per 256 (94 instead of 142 FLOP per staged byte) and the K-step is 32 (two 28-KB stages per workgroup must fit 80 KB) — which is
// why the long-K shapes stay on the 256-row kernels.
//
// Per workgroup: 4 waves as 2(M) x 2(N), wave tile 64 x NB*32 = 2 x NB MFMA 32x32x16 tiles (the accumulator layout and the
// epilogue of the 256-row kernels: persist_epilogue, gemm_common.h).  K-steps of 32: [X 128 rows | W BN rows] x 64 bytes go
// global -> LDS by LDS-DMA (16 rows x 64 B per wave-instruction; the 16-byte chunk index is XOR-ed with (row >> 2) & 3 on the
// source address and on the fragment read, conflict-free for ds_read_b128), two stages, one barrier per K-step.  No prefetch across
// tiles: the other workgroup of the CU covers prologue and epilogue.
#include "gemm_common.h"

namespace {

constexpr int DBM = 128;
template <int NB> struct DCfg {
  static constexpr int BN = NB * 64;
  static constexpr int XBYTES = DBM * 64;
  static constexpr int WBYTES = BN * 64;
  static constexpr int STAGE = XBYTES + WBYTES;
  static constexpr int EPI_BYTES = 4 * 32 * 68 * 4;                    // per-wave 32 x 68 fp32 transposition buffers (over both stages)
  static constexpr int BIAS_OFF = 2 * STAGE > EPI_BYTES ? 2 * STAGE : EPI_BYTES;
  static constexpr int SMEM = BIAS_OFF + 2048;                         // [bias fp32 @0 | rowbias 16-bit @1280]
};

template <int EPI, int NB, bool RES>
__global__ __launch_bounds__(256, 2) void gemm_duo_kernel(const GemmParams p) {
  using DC = DCfg<NB>;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  char* const smem_b = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, g = lane >> 5;
  const int lr = lane >> 2, pos = lane & 3;             // DMA role: row within the 16-row piece, 16-byte slot
  const uint32_t lds0 = lds_addr(smem);

  const int64_t ntiles = p.tiles_m * p.tiles_n;
  const int64_t G = gridDim.x;
  int64_t t = xcd_remap(blockIdx.x, G);
  if (t >= ntiles) return;
  const int nk = (int)(p.K / 32);

  // fragment reads: row (.. + l31), logical chunk 2 * ks + g of the 32-wide K-step, stored at chunk ^ ((row >> 2) & 3)
  const uint32_t sw = (uint32_t)((l31 >> 2) & 3);
  const uint32_t koffA = ((uint32_t)g ^ sw) << 4, koffB = ((uint32_t)(2 + g) ^ sw) << 4;      // ks = 0 / 1
  const uint32_t xrd = (uint32_t)(wm * 64 + l31) * 64u;
  const int wblk = (NB == 5) ? wn * 4 : wn * NB;
  const int wblk_last = (NB == 5) ? 8 + wn : wn * NB + NB - 1;
  const uint32_t wrd = (uint32_t)DC::XBYTES + (uint32_t)(wblk * 32 + l31) * 64u;
  const uint32_t wrd_last = (uint32_t)DC::XBYTES + (uint32_t)(wblk_last * 32 + l31) * 64u;

  const uint32_t vx0 = (uint32_t)(lr * p.ldx * 2 + ((pos ^ ((lr >> 2) & 3)) << 4));
  const uint32_t vw0 = (uint32_t)(lr * p.ldw * 2 + ((pos ^ ((lr >> 2) & 3)) << 4));
  const uint32_t sx16 = (uint32_t)(p.ldx * 32), sw16 = (uint32_t)(p.ldw * 32);               // bytes between two pieces (16 rows)
  uint64_t xk = 0, wk = 0;                     // scalar sources of this wave's first X / W piece of the K-step requested next

  f32x16_t acc[NB][2];
  u32x4_t fx[2][2], fw[2][NB];
  for (;;) {
    const int64_t tile_n = t % p.tiles_n, tile_m = t / p.tiles_n;
    const int64_t m0 = tile_m * DBM, n0 = tile_n * DC::BN;
    xk = (uint64_t)(uintptr_t)(p.X + (m0 + wid * 32) * p.ldx);
    wk = (uint64_t)(uintptr_t)(p.W + (n0 + wid * (NB * 16)) * p.ldw);
    auto issue = [&](int buf) __attribute__((always_inline)) {
      const uint32_t dst = lds0 + (uint32_t)buf * DC::STAGE;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        glds16_s(vx0, (const void*)(uintptr_t)(xk + (uint64_t)(uint32_t)(i * sx16)), dst + (uint32_t)(wid * 2 + i) * 1024u);
#pragma unroll
      for (int i = 0; i < NB; ++i)
        glds16_s(vw0, (const void*)(uintptr_t)(wk + (uint64_t)(uint32_t)(i * sw16)), dst + (uint32_t)DC::XBYTES + (uint32_t)(wid * NB + i) * 1024u);
      xk += 64; wk += 64;
    };
    {   // per-tile epilogue vectors ride along with the first K-step
      const uint32_t bdst = lds0 + (uint32_t)DC::BIAS_OFF;
      if (p.bias) {
        if (wid == 0) glds16_s((uint32_t)lane * 16u, p.bias + n0, bdst);
        if (NB == 5 && wid == 1) { if (lane < 16) glds16_s((uint32_t)lane * 16u, p.bias + n0 + 256, bdst + 1024u); }
      }
      if (EPI == EPI_LINEAR && p.rowbias && wid == 2) {
        if (lane < DC::BN / 8) glds16_s((uint32_t)lane * 16u, p.rowbias + (m0 / p.rb_div) * p.N + n0, bdst + 1280u);
      }
    }
    issue(0);
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of K-step kt have landed ...
      asm volatile("s_barrier" ::: "memory");                         // ... everybody's have, and stage buf^1 is no longer read
      if (kt + 1 < nk) issue(buf ^ 1);
      const char* xs = smem_b + buf * DC::STAGE + xrd;
      const char* ws = smem_b + buf * DC::STAGE + wrd;
      const char* wl = smem_b + buf * DC::STAGE + wrd_last;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const uint32_t ko = s ? koffB : koffA;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) fx[s][tm] = *reinterpret_cast<const u32x4_t*>(xs + tm * 2048 + ko);
#pragma unroll
        for (int tn = 0; tn < NB - 1; ++tn) fw[s][tn] = *reinterpret_cast<const u32x4_t*>(ws + tn * 2048 + ko);
        fw[s][NB - 1] = *reinterpret_cast<const u32x4_t*>(wl + ko);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int tn = 0; tn < NB; ++tn)
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) acc[tn][tm] = mfma32(fw[s][tn], fx[s][tm], acc[tn][tm]);
    }
    asm volatile("s_barrier" ::: "memory");                           // both stages are free: they become the transposition buffers

    persist_epilogue<EPI, NB, RES>(p, acc, reinterpret_cast<float*>(smem_b) + wid * (32 * 68),
                                   reinterpret_cast<const float*>(smem_b + DC::BIAS_OFF),
                                   reinterpret_cast<const uint16_t*>(smem_b + DC::BIAS_OFF + 1280),
                                   m0, n0, wm, wblk, wblk_last, lane);
    t += G;
    if (t >= ntiles) break;
    asm volatile("s_barrier" ::: "memory");                           // every wave is out of its transposition buffer (and of the bias vectors)
  }
}

template <int EPI, int NB, bool RES>
int launch_duo(hipStream_t stream, const GemmParams& p, int cus) {
  using DC = DCfg<NB>;
  static uint64_t attr_done = 0;
  if (int rc = a3d_once_per_device(attr_done, [] {
        return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_duo_kernel<EPI, NB, RES>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, DC::SMEM); })) return rc;
  const int64_t ntiles = p.tiles_m * p.tiles_n;
  const int64_t slots = 2 * (int64_t)cus;
  const unsigned grid = (unsigned)(ntiles < slots ? ntiles : slots);
  gemm_duo_kernel<EPI, NB, RES><<<dim3(grid), dim3(256), DC::SMEM, stream>>>(p);
  return a3d_launch_status();
}

}  // namespace

// Dense A only.  The caller (try_launch_persist, gemm_conv.hip) has checked M % 128 == 0, K % 32 == 0, N % (nb * 64) == 0 and the
// 32-bit DMA offsets, and has filled tiles_m (128-row tiles) / tiles_n.
int A3D_FN(a3d_launch_gemm_duo)(int epi, int nb, hipStream_t stream, const GemmParams& p, int cus) {
  if (epi == EPI_GEGLU) {
    if (nb == 4) return launch_duo<EPI_GEGLU, 4, false>(stream, p, cus);
    return A3D_EUNSUPPORTED;
  }
  if (nb == 5) return p.R ? launch_duo<EPI_LINEAR, 5, true>(stream, p, cus) : launch_duo<EPI_LINEAR, 5, false>(stream, p, cus);
  if (nb == 4) return p.R ? launch_duo<EPI_LINEAR, 4, true>(stream, p, cus) : launch_duo<EPI_LINEAR, 4, false>(stream, p, cus);
  return A3D_EUNSUPPORTED;
}
