// Issue-rate micro-benchmark behind the "exp ceiling" quoted in DESIGN.md / bench.py (gfx950):
// how many cycles one wave64 v_exp_f32 occupies a SIMD, alone, next to v_cvt_pk_bf16_f32, next to MFMAs of the same
// wave, and next to an MFMA-only partner wave on the same SIMD (the situation of the ping-pong attention kernel).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_exp tools/ubench_exp.hip && tools/ubench_exp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: 16 independent v_exp_f32 per iteration
// MODE 1: 16 v_exp_f32 + 8 v_cvt_pk_bf16_f32 (the softmax body of the attention kernel)
// MODE 2: 16 v_exp_f32 + 8 v_cvt_pk + 7 MFMA 32x32x16 in the same wave (28 MFMA per 64 exp, as in the D = 40 kernel)
// MODE 3: waves 0-3 of a 512-thread block run MFMAs only, waves 4-7 run MODE 1 (one of each per SIMD)
// MODE 4: 7 MFMA only (matrix-pipe reference)
// MODE 5: 16 v_max3_f32 (plain full-rate VALU reference)
// MODE 6: the work of MODE 2 hand-interleaved: {1 MFMA, 2-3 v_exp, 1 v_cvt_pk} groups pinned with sched_barrier(0)
// MODE 7: MODE 6 + 11 v_max3_f32 per iteration (the row-max share of the attention kernel)
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  const int wid = threadIdx.x >> 6;
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = -1.0f - 0.001f * (threadIdx.x + i);
  f32x16_t acc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  bf16x8_t a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * threadIdx.x); b[i] = (__bf16)(0.002f * i); }
  unsigned pk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float mx[4] = {0.f, 0.f, 0.f, 0.f};
  const bool mfma_wave = (MODE == 4) || (MODE == 3 && wid < 4) || (MODE == 2);
  const bool exp_wave = (MODE == 0 || MODE == 1 || MODE == 2) || (MODE == 3 && wid >= 4);
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 5) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(v[(i + 2) & 15]));
    }
    if (MODE == 6 || MODE == 7) {
      // 7 MFMA slots; 16 exps + 8 cvts (+ 11 max3) spread over them.  cvt i consumes exps 2i, 2i+1 of the PREVIOUS slot.
#pragma unroll
      for (int sl = 0; sl < 7; ++sl) {
        acc[sl & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[sl & 1], 0, 0, 0);
        const int e0 = (16 * sl) / 7, e1 = (16 * (sl + 1)) / 7;
#pragma unroll
        for (int i = e0; i < e1; ++i) asm volatile("v_exp_f32 %0, %1" : "=v"(v[i]) : "v"(v[i]));
        const int c0 = (8 * sl) / 7, c1 = (8 * (sl + 1)) / 7;
#pragma unroll
        for (int i = c0; i < c1; ++i) {
          const int j = (i + 7) & 7;      // pairs finished in an earlier slot
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[j]) : "v"(v[2 * j]), "v"(v[2 * j + 1]));
        }
        if (MODE == 7) {
          const int m0 = (11 * sl) / 7, m1 = (11 * (sl + 1)) / 7;
#pragma unroll
          for (int i = m0; i < m1; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx[i & 3]) : "v"(v[i]), "v"(v[i + 4]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (exp_wave) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %1" : "=v"(v[i]) : "v"(v[i]));
      if (MODE != 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(v[2 * i]), "v"(v[2 * i + 1]));
      }
    }
    if (mfma_wave) {
#pragma unroll
      for (int i = 0; i < 7; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 1], 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  float s = acc[0][0] + acc[1][3] + mx[0] + mx[1] + mx[2] + mx[3];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += __uint_as_float(pk[i]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wid] = t1 - t0;
}

template <int MODE>
int run(const char* name, int threads, int iters, double exps_per_iter, double mfma_per_iter) {
  const int blocks = 256;
  float* out; long long* cyc;
  CHECK(hipMalloc(&out, sizeof(float) * blocks * 512));
  CHECK(hipMalloc(&cyc, sizeof(long long) * blocks * 8));
  CHECK(hipMemset(cyc, 0, sizeof(long long) * blocks * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<MODE><<<blocks, threads>>>(out, cyc, iters);                  // warm-up
  CHECK(hipEventRecord(e0));
  k<MODE><<<blocks, threads>>>(out, cyc, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> h(blocks * 8);
  CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 8, hipMemcpyDeviceToHost));
  const int waves = threads / 64;
  double lo = 0, hi = 0; int nlo = 0, nhi = 0;
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < waves; ++w) { if (w < 4) { lo += h[b * 8 + w]; ++nlo; } else { hi += h[b * 8 + w]; ++nhi; } }
  lo /= nlo; if (nhi) hi /= nhi;
  printf("%-62s threads %3d: %8.3f ms | waves 0-3: %8.1f clk/iter", name, threads, ms, lo / iters);
  if (nhi) printf(" | waves 4-7: %8.1f clk/iter", hi / iters);
  if (exps_per_iter > 0) {
    const double exp_waves = (MODE == 3) ? 4.0 : waves;
    const double chip_rate = exps_per_iter * 64.0 * exp_waves * blocks * iters / (ms * 1e-3);
    printf(" | %6.2f T exp/s chip", chip_rate / 1e12);
  }
  if (mfma_per_iter > 0) {
    const double mf_waves = (MODE == 3) ? 4.0 : waves;
    printf(" | %7.1f TFLOP/s MFMA", mfma_per_iter * 32768.0 * mf_waves * blocks * iters / (ms * 1e-3) / 1e12);
  }
  printf("\n");
  CHECK(hipFree(out)); CHECK(hipFree(cyc));
  return 0;
}

int main() {
  const int it = 20000;
  run<5>("16 v_max3_f32", 256, it, 0, 0);
  run<5>("16 v_max3_f32", 512, it, 0, 0);
  run<0>("16 v_exp_f32", 256, it, 16, 0);
  run<0>("16 v_exp_f32", 512, it, 16, 0);
  run<1>("16 v_exp_f32 + 8 v_cvt_pk_bf16_f32", 256, it, 16, 0);
  run<1>("16 v_exp_f32 + 8 v_cvt_pk_bf16_f32", 512, it, 16, 0);
  run<4>("7 MFMA 32x32x16 bf16", 256, it, 0, 7);
  run<4>("7 MFMA 32x32x16 bf16", 512, it, 0, 7);
  run<2>("16 v_exp + 8 v_cvt_pk + 7 MFMA, same wave", 256, it, 16, 7);
  run<2>("16 v_exp + 8 v_cvt_pk + 7 MFMA, same wave", 512, it, 16, 7);
  run<6>("7 x {MFMA, 2-3 v_exp, 1 v_cvt_pk} hand-interleaved", 256, it, 16, 7);
  run<6>("7 x {MFMA, 2-3 v_exp, 1 v_cvt_pk} hand-interleaved", 512, it, 16, 7);
  run<7>("7 x {MFMA, 2-3 v_exp, 1 v_cvt_pk, 1-2 v_max3} hand-interleaved", 256, it, 16, 7);
  run<7>("7 x {MFMA, 2-3 v_exp, 1 v_cvt_pk, 1-2 v_max3} hand-interleaved", 512, it, 16, 7);
  run<3>("waves 0-3: 7 MFMA | waves 4-7: 16 v_exp + 8 v_cvt_pk (per SIMD)", 512, it, 16, 7);
  return 0;
}
