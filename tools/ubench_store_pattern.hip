// VERDICT r5 item 4a, measured in isolation: what does the OUTPUT side of a 256 x 320 GEMM tile cost with
//   (A) the shipped epilogue's store pattern — the MFMA-layout accumulators rounded to 16 bits, transposed through a wave-private LDS
//       buffer, one lane owning 8 consecutive columns: 16-byte stores, 8 lanes per 128-byte row segment (gemm_common.h: persist_epilogue), and
//   (B) the proposed direct pattern — W rows permuted inside every 32-block so that register r of a lane is output column 16 g + r of ONE row:
//       no LDS at all, two global_store_dwordx4 per lane and 32 x 32 MFMA tile, every lane of a store instruction on a different row.
// One 512-thread workgroup per CU walks tiles of a [M, N] 16-bit output exactly as gemm_pp.hip does (8 waves as 4 (M) x 2 (N), wave = 64 rows
// x 160 columns = 2 x 5 MFMA tiles); the "accumulators" are synthetic registers, there is no main loop: the time is the store pattern's alone
// (plus, for A, the LDS transposition it needs).  If B alone is not clearly faster than A, moving bias / residual into B's layout cannot win
// the 10 % the experiment was asked to show (kill criterion: N = 1280, K = 320 at 0.58 ms against 0.64).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_store_pattern tools/ubench_store_pattern.hip && tools/ubench_store_pattern
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t pack16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// MODE 0: pattern A (LDS transposition, coalesced 16-byte stores); MODE 1: pattern B (direct row-per-lane 16-byte stores);
// MODE 2: pattern B with 8-byte stores (round 4's rejected form, for reference)
template <int MODE>
__global__ __launch_bounds__(512, 1) void store_kernel(uint16_t* __restrict__ Y, int64_t ldy, int tiles_m, int tiles_n, float seed) {
  __shared__ __attribute__((aligned(16))) char stg[8 * 8192];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, g = lane >> 5;
  float acc[5][2][16];
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = seed * (float)(tid + 16 * (2 * a + b) + r);
  char* const sb = stg + wid * 8192;
  for (int t = blockIdx.x; t < tiles_m * tiles_n; t += gridDim.x) {
    const int64_t m0 = (int64_t)(t / tiles_n) * 256, n0 = (int64_t)(t % tiles_n) * 320;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int64_t mbase = m0 + wm * 64 + tm * 32;
      if constexpr (MODE == 0) {
        // passes of 64 / 64 / 32 columns as in persist_epilogue: stage rounded 16-bit values (8-byte LDS writes in the MFMA layout: a lane holds
        // 4 consecutive columns of row l31 per register quad), read back 16 bytes per lane, store
#pragma unroll
        for (int ps = 0; ps < 3; ++ps) {
          const int ncol = ps < 2 ? 64 : 32;
          const int col0 = (ps < 2 ? wn * 4 + 2 * ps : 8 + wn) * 32;
          char* const buf = sb + (ps & 1) * 4096;
#pragma unroll
          for (int tl = 0; tl < 2; ++tl) {
            if (32 * tl >= ncol) continue;
            const int tn = ps < 2 ? 2 * ps + tl : 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              u32x2 o;
              o[0] = pack16(acc[tn][tm][4 * q], acc[tn][tm][4 * q + 1]);
              o[1] = pack16(acc[tn][tm][4 * q + 2], acc[tn][tm][4 * q + 3]);
              *reinterpret_cast<u32x2*>(buf + l31 * 128 + 16 * ((4 * tl + q) ^ (l31 & 7)) + 8 * (g ^ ((l31 >> 3) & 1))) = o;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const int lpr = ncol / 8, cc = lane & (lpr - 1);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (j * (64 / lpr) >= 32) continue;
            const int row = (64 / lpr) * j + lane / lpr;
            const u32x4 o = *reinterpret_cast<const u32x4*>(buf + row * 128 + 16 * (cc ^ (row & 7)));
            *reinterpret_cast<u32x4*>(Y + (mbase + row) * ldy + n0 + col0 + 8 * cc) = o;
          }
        }
      } else {
        // direct: lane (l31, g) owns row mbase + l31, columns blk * 32 + 16 g + [0, 16) of every 32-column block of its wave
        uint16_t* const yrow = Y + (mbase + l31) * ldy + n0;
#pragma unroll
        for (int tn = 0; tn < 5; ++tn) {
          const int blk = tn < 4 ? wn * 4 + tn : 8 + wn;
          uint16_t* const dst = yrow + blk * 32 + 16 * g;
          if constexpr (MODE == 1) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              u32x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = pack16(acc[tn][tm][8 * h + 2 * e], acc[tn][tm][8 * h + 2 * e + 1]);
              *reinterpret_cast<u32x4*>(dst + 8 * h) = o;
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              u32x2 o;
              o[0] = pack16(acc[tn][tm][4 * q], acc[tn][tm][4 * q + 1]);
              o[1] = pack16(acc[tn][tm][4 * q + 2], acc[tn][tm][4 * q + 3]);
              *reinterpret_cast<u32x2*>(dst + 4 * q) = o;
            }
          }
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 5; ++a) acc[a][0][0] += 1.0f;      // (keeps the tile loop from being hoisted)
  }
}

template <int MODE>
float run(uint16_t* Y, int64_t M, int64_t N, int reps) {
  const int tiles_m = (int)(M / 256), tiles_n = (int)(N / 320);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  store_kernel<MODE><<<256, 512>>>(Y, N, tiles_m, tiles_n, 0.001f);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0);
    store_kernel<MODE><<<256, 512>>>(Y, N, tiles_m, tiles_n, 0.001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const int64_t M = 524288;
  uint16_t* Y;
  CHECK(hipMalloc(&Y, (size_t)M * 1280 * 2));
  printf("output side of a 256 x 320 GEMM tile alone (no main loop, no operands), one 512-thread workgroup per CU, best of 10; M = %lld\n", (long long)M);
  for (int64_t N : {1280LL, 960LL, 320LL}) {
    const double bytes = (double)M * N * 2;
    const float a = run<0>(Y, M, N, 10), b = run<1>(Y, M, N, 10), c = run<2>(Y, M, N, 10);
    printf("N = %4lld: (A) LDS-transposed, coalesced 16-byte stores %.3f ms = %.2f TB/s | (B) direct row-per-lane 16-byte stores %.3f ms = %.2f TB/s | "
           "(B8) direct 8-byte stores %.3f ms = %.2f TB/s\n", (long long)N, a, bytes / a / 1e9, b, bytes / b / 1e9, c, bytes / c / 1e9);
  }
  hipFree(Y);
  return 0;
}
