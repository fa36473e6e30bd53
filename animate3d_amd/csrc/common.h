// Shared device helpers for the gfx950 kernels of libanimate3d_hip.so.
// Wave = 64 lanes; MFMA shape used throughout: v_mfma_f32_32x32x16_{bf16,f16}.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

#include "../../include/animate3d_hip.h"

typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

#define A3D_DEV static __device__ __forceinline__

// ---- the 16-bit activation / weight type of this build of the kernels
// Every kernel source is compiled twice (animate3d_amd/build.py): as is for bf16 storage and with -DA3D_STORAGE_F16 for IEEE fp16
// storage (the dtype the reference's 4D-SDS caller runs the UNet in, animatemv_guidance.py:339-346).  Arithmetic is the same
// in both: fp32 accumulation, statistics and softmax; only the load / store conversions, the MFMA opcode
// (v_mfma_f32_32x32x16_bf16 / _f16), the packed dot product of the temporal attention and the bit pattern of 1.0 differ.
// A3D_FN(name) appends the storage suffix to an entry-point name: a3d_gemm_bf16 / a3d_gemm_f16.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#ifdef A3D_STORAGE_F16
#define A3D_FN(base) base##_f16
typedef _Float16 h16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2_t __attribute__((ext_vector_type(2)));
constexpr uint16_t ONE16 = 0x3C00;       // 1.0
A3D_DEV float h2f(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// v_cvt_pk_f16_f32: round-to-nearest-even, two values per instruction
A3D_DEV uint32_t pack16(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h16x2_t));
}
A3D_DEV float lo16(uint32_t w) { return (float)__builtin_bit_cast(h16x2_t, w)[0]; }
A3D_DEV float hi16(uint32_t w) { return (float)__builtin_bit_cast(h16x2_t, w)[1]; }
#else
#define A3D_FN(base) base##_bf16
typedef __bf16 h16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 h16x2_t __attribute__((ext_vector_type(2)));
constexpr uint16_t ONE16 = 0x3F80;       // 1.0
// bf16 <-> f32 (round-to-nearest-even on the way down, as torch does)
A3D_DEV float h2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// v_cvt_pk_bf16_f32: hardware round-to-nearest-even, two values per instruction
A3D_DEV uint32_t pack16(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h16x2_t));
}
A3D_DEV float lo16(uint32_t w) { return __uint_as_float(w << 16); }
A3D_DEV float hi16(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
#endif
// 16-byte accesses of the streaming (memory-bound) kernels — GroupNorm, LayerNorm, concat: tensors read or written once per launch
// and larger than the caches (336 MB at level 0), marked non-temporal (global_load / global_store ... nt).  Measured against plain
// accesses (profiles/r4_microbench_stream_variants.log): GroupNorm 0.220 -> 0.196 ms, two-output LayerNorm 0.254 -> 0.233 (with the
// prefetch in layer_norm_rows_kernel), concat 0.40 -> 0.35-0.36; -1..-2 ms per denoise step.
A3D_DEV u32x4_t ld_stream(const void* ptr) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(ptr)); }
A3D_DEV void st_stream(void* ptr, u32x4_t v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(ptr)); }
A3D_DEV uint16_t f2h(float f) { return (uint16_t)(pack16(f, 0.f) & 0xffffu); }
// caller-side bf16 tensors at the boundary kernels (im2col_in / unpack_out): always bf16, whatever the storage type of the build
A3D_DEV float bfbits2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
A3D_DEV uint16_t f2bfbits(float f) {
  typedef __bf16 bfx2 __attribute__((ext_vector_type(2)));
  const f32x2_t v = {f, 0.f};
  return (uint16_t)(__builtin_bit_cast(uint32_t, __builtin_convertvector(v, bfx2)) & 0xffffu);
}

// MFMA 32x32x16 (bf16 or fp16 inputs, fp32 accumulate).  Lane l supplies A[i = l&31][k-slots of group l>>5] and
// B[k-slots of group l>>5][j = l&31] (8 bf16 each); the result register r of lane l is
// D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31].
A3D_DEV f32x16_t mfma32(const u32x4_t& a, const u32x4_t& b, const f32x16_t& c) {
#ifdef A3D_STORAGE_F16
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8_t, a), __builtin_bit_cast(h16x8_t, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(h16x8_t, a), __builtin_bit_cast(h16x8_t, b), c, 0, 0, 0);
#endif
}
// MFMA 16x16x32 (bf16 or fp16 inputs, fp32 accumulate).  Lane l supplies A[i = l&15][k = 8*(l>>4) .. +7] and B[k = 8*(l>>4) .. +7][j = l&15];
// result register r of lane l is D[i = 4*(l>>4) + r][j = l&15].
typedef float f32x4_t __attribute__((ext_vector_type(4)));
A3D_DEV f32x4_t mfma16(const u32x4_t& a, const u32x4_t& b, const f32x4_t& c) {
#ifdef A3D_STORAGE_F16
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8_t, a), __builtin_bit_cast(h16x8_t, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(h16x8_t, a), __builtin_bit_cast(h16x8_t, b), c, 0, 0, 0);
#endif
}
// row index inside a 32x32 MFMA result for register r of a lane in half g = lane>>5
// ds_read_b64_tr_b16: within a 16-lane group, lanes 4 j .. 4 j + 3 address row j (8 bytes = 4 columns each); lane i receives column i of
// the four rows — the transposing LDS read behind every contraction-over-rows MFMA operand
typedef short v4i16_t __attribute__((ext_vector_type(4)));
A3D_DEV u32x2_t lds_tr16_b64(const uint16_t* ptr) {
  return __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16_t*)ptr));
}

// compile-time unrolled loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <int N, typename F, int... I>
A3D_DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
A3D_DEV void static_for(F&& f) { static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

A3D_DEV int mfma_row(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

// bijective XCD-aware remap of a 1-D grid: block b runs on XCD b % 8; give every XCD a
// contiguous range of logical ids so neighbouring tiles share that XCD's L2.
A3D_DEV int64_t xcd_remap(int64_t bid, int64_t nblk) {
  const int64_t q = nblk / 8, r = nblk % 8;
  const int64_t xcd = bid % 8, idx = bid / 8;
  const int64_t base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

A3D_DEV uint32_t xcd_remap32(uint32_t bid, uint32_t nblk) {      // the same for grids below 2^32 (no 64-bit divisions)
  const uint32_t q = nblk / 8u, r = nblk % 8u;
  const uint32_t xcd = bid % 8u, idx = bid / 8u;
  const uint32_t base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
  return base + idx;
}

// Per-device one-time state (dynamic-LDS attribute, CU count): one process may drive several GPUs, so the "done" flags
// are bit masks / tables indexed by the current HIP device, not process-wide booleans.
static inline int a3d_current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) d = 0;
  return d;
}
template <typename F>
static inline int a3d_once_per_device(uint64_t& done_mask, F&& fn) {
  const int d = a3d_current_device();
  if ((done_mask >> d) & 1ull) return 0;
  const int rc = fn();
  if (rc == 0) done_mask |= (1ull << d);
  return rc;
}

static inline int a3d_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? A3D_OK : (int)e;
}
