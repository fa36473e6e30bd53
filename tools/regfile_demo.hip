// Evidence for profiles/README.md (round 5): how many registers a wave of a 512-thread workgroup can hold on gfx950.
// A CU has 4 SIMDs x 512 registers per lane (VGPR + AGPR, one unified file).  A 512-thread workgroup is 8 waves = 2 waves per SIMD, and
// all of them must be resident at once, so each wave gets at most 512 / 2 = 256 registers — arch and accumulator registers TOGETHER.
// `two_acc_sets<512>` keeps two 160-register accumulator tiles live (the "second accumulator set in the unused AGPRs" of the round-4
// write-up) and spills; `two_acc_sets<256>` (4 waves = one per SIMD, 512 registers each) holds them.
//   python tools/resource_usage.py tools/regfile_demo.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int THREADS>
__global__ __launch_bounds__(THREADS, 1) void two_acc_sets(const bf16x8* a, const bf16x8* b, float* out, int n) {
  f32x16 acc0[10], acc1[10];
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[i][r] = 0.f; acc1[i][r] = 0.f; }
  for (int k = 0; k < n; ++k) {
    const bf16x8 x = a[k * THREADS + threadIdx.x], y = b[k * THREADS + threadIdx.x];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      acc0[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc0[i], 0, 0, 0);
      acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, acc1[i], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[((i * 16 + r) * 2) * THREADS + threadIdx.x] = acc0[i][r], out[((i * 16 + r) * 2 + 1) * THREADS + threadIdx.x] = acc1[i][r];
}
template __global__ void two_acc_sets<512>(const bf16x8*, const bf16x8*, float*, int);
template __global__ void two_acc_sets<256>(const bf16x8*, const bf16x8*, float*, int);
