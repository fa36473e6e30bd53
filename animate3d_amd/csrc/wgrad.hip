// Weight gradient dW[N, K] = alpha * dY^T X of every trainable Linear (training path; reference: autograd of the
// nn.Linear layers inside `motion_modules.` / `i2v.`, train.py:576-590).
//
// Shape of the problem on this model: the OUTPUT is small (320^2 .. 1280 x 5120) and the CONTRACTION is the token axis
// (M = 4 096 .. 65 536 rows).  A GEMM kernel that tiles the output gets 9 .. 400 workgroups with a contraction loop thousands of
// steps long — the first version of the training path ran it that way (on transposed copies of dY and X) and spent 25 % of the step
// on <= 9 of 256 CUs.  Here the token axis is split over the grid instead (split-K): a workgroup owns a 128 x 128 output tile and
// a slice of M, reads dY and X in their natural row-major layout (no transposed copies) and writes its partial tile to a workspace
// [splits][N][K]; a second, bandwidth-trivial kernel sums the splits.  (Adding the partial tiles into the result with global atomics
// was measured first: the splits of one tile finish together and ~60 of them serialise on the same L2 lines — the epilogue took
// 20 x longer than the products, profiles/README.md.)
//
// Both MFMA operands need the token index as the contraction, i.e. transposed images dY^T [n][m], X^T [k][m]: they are built while
// staging (4 rows x 8 columns per thread, v_perm_b32 + 8-byte LDS writes — the forward attention's V^T path), double-buffered in
// LDS with register prefetch of the next 32 rows; one barrier per step.  4 waves = 2 x 2, each 64 x 64 of the tile.
#include "gemm_common.h"      // LDS-DMA helpers (glds16_v, lds_addr), the zero page


namespace {

struct WgParams {
  const uint16_t* dY; int64_t lddy; const uint16_t* X; int64_t ldx;
  float* ws;
  int64_t M; int N, K; int tiles_k; int tiles; int splits; int64_t rows_per_split; float alpha;
};

// ---- The split-K tile with LDS-DMA staging (round 3).  The round-2 kernel (register staging) spent its time between the MFMAs: per 32 token
// rows a wave has 8 MFMAs (256 cycles) against global loads it waits for one step ahead, 16 v_perm + 8 ds_write_b64 of the
// transposition and a barrier (130-200 TFLOP/s).  Here both operand tiles go global -> LDS untouched, row-major [token][128 columns]
// (global_load_lds_dwordx4, 4 rows of 256 B per instruction, exactly four instructions per wave and stage so the wait is a counted
// vmcnt(8) with two stages in flight behind the one being read; ring of four 16-KB stages = 64 KB, two workgroups per CU), and the
// transposition the MFMA operands need (contraction = token index) happens on the way OUT of LDS: ds_read_b64_tr_b16 hands lane i of
// a 16-lane group column i of a 4-row x 16-column block.  16-byte chunk c of row r sits at c ^ ((r & 3) << 2): the four rows a
// 16-lane group reads then occupy four different 64-byte windows (conflict-free for both groups of a 32-lane half).  Columns past
// N / K and rows past the slice come from a zero page (per-lane source select: the instruction count stays exact).
constexpr int WD_BR = 32;                     // token rows per stage
constexpr int WD_ROWB = 256;                  // bytes per tile row (128 columns)
constexpr int WD_OPB = WD_BR * WD_ROWB;       // one operand tile of a stage
constexpr int WD_STG = 2 * WD_OPB;            // [dY rows | X rows]
constexpr int WD_RING = 4;
constexpr int WD_SMEM = WD_RING * WD_STG;

__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(const WgParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t wd_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5, i16 = lane & 15, q4 = lane >> 4;
  const int wn = wid >> 1, wk = wid & 1;
  const int tile = blockIdx.x, split = blockIdx.y;
  const int n0 = (tile / p.tiles_k) * 128, k0 = (tile % p.tiles_k) * 128;
  const int64_t m_beg = (int64_t)split * p.rows_per_split;
  const int64_t m_end = min(p.M, m_beg + p.rows_per_split);
  const uint32_t lds0 = lds_addr(wd_smem);

  // ---- DMA role: waves 0, 1 stage dY rows 16 (w & 1) .. + 15 of a stage, waves 2, 3 the X rows; lane = (row lr of a 4-row piece, LDS chunk pos)
  const bool is_a = wid < 2;
  const uint16_t* const src = is_a ? p.dY : p.X;
  const int64_t ld = is_a ? p.lddy : p.ldx;
  const int lr = lane >> 4, pos = lane & 15;
  const int col = (is_a ? n0 : k0) + 8 * (pos ^ (lr << 2));
  const bool col_ok = col < (is_a ? p.N : p.K);
  const int row0 = (wid & 1) * 16 + lr;                          // + 4 i for piece i
  const uint32_t dst0 = (uint32_t)((is_a ? 0 : WD_OPB) + (wid & 1) * 16 * WD_ROWB);
  const char* const zp = reinterpret_cast<const char*>(g_zero_page);
  auto issue = [&](int64_t s) __attribute__((always_inline)) {
    const int64_t m0 = m_beg + s * WD_BR;
    const uint32_t dst = lds0 + (uint32_t)((s & (WD_RING - 1)) * WD_STG) + dst0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m0 + row0 + 4 * i;
      const char* a = (col_ok && m < m_end) ? reinterpret_cast<const char*>(src + m * ld + col) : zp;
      glds16_v(a, dst + (uint32_t)(i * 4 * WD_ROWB));
    }
  };

  // ---- fragment addressing: lane (group q4, i16) reads 8 bytes of token row 8 g + 4 rr + (i16 >> 2) (+ 16 ks) at columns
  // block + 16 (q4 & 1) + 4 (i16 & 3) .. + 3; after the transposing read it owns column block + l31, rows 8 g + 4 rr .. + 3
  const int rsub = i16 >> 2;
  auto lane_off = [&](int colblk) -> uint32_t {
    const int c = colblk + 16 * (q4 & 1) + 4 * (i16 & 3);
    return (uint32_t)((8 * g + rsub) * WD_ROWB + (((c >> 3) ^ (rsub << 2)) << 4) + (c & 7) * 2);
  };
  uint32_t a_off[2], b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    a_off[i] = lane_off(wn * 64 + 32 * i);
    b_off[i] = (uint32_t)WD_OPB + lane_off(wk * 64 + 32 * i);
  }

  f32x16_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int64_t nsteps = (m_end - m_beg + WD_BR - 1) / WD_BR;
  issue(0); issue(1); issue(2);
  for (int64_t s = 0; s < nsteps; ++s) {
    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");      // stage s landed for everybody; everybody is done reading stage s - 1
    issue(s + 3);                                                       // ... whose slot stage s + 3 takes (past the slice: zero page)
    const uint8_t* const T = wd_smem + (s & (WD_RING - 1)) * WD_STG;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4_t af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x2_t a0 = lds_tr16_b64(reinterpret_cast<const uint16_t*>(T + a_off[i] + (16 * ks) * WD_ROWB));
        const u32x2_t a1 = lds_tr16_b64(reinterpret_cast<const uint16_t*>(T + a_off[i] + (16 * ks + 4) * WD_ROWB));
        const u32x2_t b0 = lds_tr16_b64(reinterpret_cast<const uint16_t*>(T + b_off[i] + (16 * ks) * WD_ROWB));
        const u32x2_t b1 = lds_tr16_b64(reinterpret_cast<const uint16_t*>(T + b_off[i] + (16 * ks + 4) * WD_ROWB));
        af[i] = u32x4_t{a0[0], a0[1], a1[0], a1[1]};
        bf[i] = u32x4_t{b0[0], b0[1], b1[0], b1[1]};
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = mfma32(af[a], bf[b], acc[a][b]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may land after the workgroup has given its LDS back

  float* const out = p.ws + (int64_t)split * p.N * p.K;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int kcol = k0 + wk * 64 + b * 32 + l31;
      if (kcol < p.K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int nrow = n0 + wn * 64 + a * 32 + mfma_row(r, g);
          if (nrow < p.N) out[(int64_t)nrow * p.K + kcol] = acc[a][b][r];
        }
      }
    }
}

// dW (+)= alpha * sum over splits
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dW, int64_t lddw, int N, int K,
                                                            int splits, float alpha, int accumulate) {
  const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;             // one float4 of the [N, K] result
  const int64_t total4 = (int64_t)N * K / 4;
  if (i4 >= total4) return;
  float4 s = {0.f, 0.f, 0.f, 0.f};
  for (int sp = 0; sp < splits; ++sp) {
    const float4 v = reinterpret_cast<const float4*>(ws + (int64_t)sp * N * K)[i4];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const int64_t e = i4 * 4;
  float* o = dW + (e / K) * lddw + (e % K);
  if (accumulate) { o[0] += alpha * s.x; o[1] += alpha * s.y; o[2] += alpha * s.z; o[3] += alpha * s.w; }
  else { o[0] = alpha * s.x; o[1] = alpha * s.y; o[2] = alpha * s.z; o[3] = alpha * s.w; }
}

}  // namespace

// splits of the token axis for a given problem (also the workspace size): ~768 workgroups (3 per CU), >= 1 024 token rows each
static void wgrad_plan(int64_t M, int64_t N, int64_t K, int64_t* splits, int64_t* rows_per_split) {
  const int64_t tiles = ((N + 127) / 128) * ((K + 127) / 128);
  int64_t sp = (768 + tiles - 1) / tiles;
  const int64_t max_splits = (M + 1023) / 1024;
  if (sp > max_splits) sp = max_splits;
  if (sp < 1) sp = 1;
  int64_t rps = (M + sp - 1) / sp;
  rps = (rps + WD_BR - 1) / WD_BR * WD_BR;
  *splits = (M + rps - 1) / rps;
  *rows_per_split = rps;
}

#ifndef A3D_STORAGE_F16
extern "C" int64_t a3d_wgrad_ws_floats(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int64_t sp, rps;
  wgrad_plan(M, N, K, &sp, &rps);
  return sp * N * K;
}
#endif

extern "C" int A3D_FN(a3d_wgrad)(a3d_stream_t stream, const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW, int64_t lddw,
                                  float* ws, int64_t M, int64_t N, int64_t K, float alpha, int accumulate) {
  if (!dY || !X || !dW || !ws || M <= 0 || N <= 0 || K <= 0 || N % 8 != 0 || K % 8 != 0 || lddy % 8 != 0 || ldx % 8 != 0) return A3D_EINVAL;
  if (lddy < N || ldx < K || lddw < K || N > 0x7fffffffLL || K > 0x7fffffffLL) return A3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(dY) & 15u) || (reinterpret_cast<uintptr_t>(X) & 15u) || (reinterpret_cast<uintptr_t>(ws) & 15u)) return A3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int64_t tiles_k = (K + 127) / 128, tiles = ((N + 127) / 128) * tiles_k;
  int64_t splits, rps;
  wgrad_plan(M, N, K, &splits, &rps);
  if (tiles > 0x7fffffffLL || splits > 65535) return A3D_EINVAL;
  WgParams p{(const uint16_t*)dY, lddy, (const uint16_t*)X, ldx, ws, M, (int)N, (int)K, (int)tiles_k, (int)tiles, (int)splits, rps, alpha};
  wgrad_dma_kernel<<<dim3((unsigned)tiles, (unsigned)splits), dim3(256), WD_SMEM, s>>>(p);
  const int64_t total4 = N * K / 4;
  wgrad_reduce_kernel<<<dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s>>>(ws, dW, lddw, (int)N, (int)K, (int)splits, alpha, accumulate);
  return a3d_launch_status();
}
