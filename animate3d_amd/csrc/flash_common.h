// Shared pieces of the flash-attention translation units (flash_attn.hip: plain / ping-pong / interleaved kernels and the
// C-ABI entry point; flash_attn_dm.hip: the LDS-DMA staged level-0 kernel).
#pragma once
#include <type_traits>
#include <utility>

#include "common.h"

struct AttnParams {
  const uint16_t* Q; const uint16_t* K; const uint16_t* V; uint16_t* O;
  a3d_rowmap qm, km, om;
  int heads; int q_len, kv_len;
  float scale_log2, out_scale; int accumulate;
  int causal;      // key s may only be seen by queries >= s of the same group (CLIP text tower); generic kernel only
  // second key set (a3d_flash_attn2: text tokens + IP-Adapter image tokens in one launch): O = out_scale * attn(Q, K, V) +
  // out_scale2 * attn(Q, K2, V2), each with its own softmax.  K2 == nullptr: single key set.
  const uint16_t* K2; const uint16_t* V2; a3d_rowmap km2; int kv_len2; float out_scale2;
  // training (a3d_flash_attn_lse): per query log2 sum_k 2^(s_qk * scale * log2 e), [groups][heads][q_len] floats, as the backward's
  // statistics pass would compute it (attn_bwd.hip); nullptr: not wanted
  float* lse;
  // diagnostics (a3d_flash_attn_counted): device words the LDS-DMA staged kernels add to — [0] workgroups that skipped the max-free pass on
  // the spread predictor's vote (fp16 storage), [1] workgroups that discarded a max-free result and re-ran exactly (overflow), [2] workgroups
  // launched.  nullptr (every other entry point): nothing is counted.  Touched on the cold paths and once at kernel start only.
  unsigned int* counters;
};

namespace {

constexpr int OFS_FMA = 0, OFS_PAD = 1, OFS_ACC = 2;
constexpr float LAZY_THR = 6.0f;     // log2 units: P may reach 2^6 before the offset moves


// Row maps in 32-bit arithmetic: groups (<= 65 535) and sequence positions (< 2^30) are checked by the entry points; a divisor beyond 2^31 is
// larger than any dividend, so clamping it leaves quotient 0 / remainder = dividend.  (A 64-bit division is a ~100-instruction routine on
// this chip and every workgroup's prologue had five to seven of them: ~1 us of a 25-50 us workgroup at 1 024 keys.)
A3D_DEV uint32_t map_clamp32(int64_t d) { return d > 0x7fffffffLL ? 0x7fffffffu : (uint32_t)d; }
A3D_DEV int64_t map_group_base(const a3d_rowmap& m, int64_t g) {
  const uint32_t g32 = (uint32_t)g, gd = map_clamp32(m.gdiv), gq = g32 / gd, gr = g32 - gq * gd;
  return (int64_t)gq * m.ga + (int64_t)gr * m.gb;
}
A3D_DEV int64_t map_seq(const a3d_rowmap& m, int64_t s) {      // row of sequence position s relative to its group's base
  const uint32_t s32 = (uint32_t)s, sl = map_clamp32(m.seg_len), sq = s32 / sl, sr = s32 - sq * sl;
  return (int64_t)sq * m.seg_stride + (int64_t)sr;
}
A3D_DEV int64_t map_row(const a3d_rowmap& m, int64_t g, int64_t s) { return map_group_base(m, g) + map_seq(m, s); }

// K row (within a 32-row sub-tile) that feeds MFMA A-row i: chosen so that result register r of a
// lane in half g is key 16*(r>>3) + 8*g + (r&7).
A3D_DEV int kperm(int i) {
  const int j = i & 3, g = (i >> 2) & 1, b = i >> 3;
  return 16 * (b >> 1) + 8 * g + 4 * (b & 1) + j;
}

A3D_DEV float round16(float x) { return lo16(pack16(x, 0.f)); }

// fp16 storage, max-free pass of the LDS-DMA kernels: where to put the window of P = exp2(S - offset) — fp16 holds 2^-24 .. 2^16 — from
// the statistics of 32 sample scores of a query (maximum mx, mean, variance; log2 units).  Rounds 3-5: offset = mx + 4, i.e. the window
// ends 20 units above the sample maximum; for Gaussian scores the maximum of 16 384 sits ~1.85 sd (+- 0.6 sd) above the maximum of 32, so
// at score sd 3 (natural units: 4.3 log2 units) 1-2 % of the rows overflowed, which is every 512-query workgroup: the whole launch fell
// back to the exact pass (+13-19 %: profiles/r5_flash_score_spread.log).  Round 6: the offset is LIFTED towards the expected row maximum
// mean + c(kv_len) sd (c = the expected maximum of kv_len standard normals: 3.2 at 1 024 keys, 3.9 at 16 384), by at most F16_LIFT_MAX so
// that the sample maximum itself stays a normal fp16 number (2^-(4 + 8)): a distribution with lighter tails than the prediction loses
// nothing but the subnormal tail of the tail.  `wide`: even the lifted window is predicted to overflow (3.7 sigma of the prediction's
// own scatter: 0.6 sd) — the workgroup votes on it (> 25 % of its queries -> exact pass at once).
constexpr float F16_BIAS = 4.f, F16_LIFT_MAX = 8.f, F16_TOP = 16.f;
A3D_DEV float f16_expected_max_sds(int kv_len) {      // E[max of n standard normals], n = 2^8 .. 2^16: 2.85 + 0.175 (log2 n - 8)
  return 2.85f + 0.175f * (float)(31 - __builtin_clz((unsigned)(kv_len < 256 ? 256 : kv_len)) - 8);
}
A3D_DEV float f16_sampled_bias(float mx, float mean, float var, float cmax, bool& wide) {
  const float sd = __builtin_amdgcn_sqrtf(fmaxf(var, 0.f));
  const float lift = fminf(fmaxf(fmaf(cmax, sd, mean) - mx, 0.f), F16_LIFT_MAX);
  wide = !(fmaf(cmax + 2.2f, sd, mean) <= mx + F16_BIAS + lift + F16_TOP);      // (NaN statistics: wide)
  return F16_BIAS + lift;
}


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
// the same under a wave-uniform lane mask (0 = this wave has no share of the tile)
A3D_DEV void dm_glds16_m(uint32_t voff, const void* sbase, uint32_t lds_dst, uint64_t mask) {
  unsigned keep;
  uint64_t ex;
  mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(mask >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mask);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, %5\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep), "=&s"(ex) : "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst)), "s"(mask) : "memory");
}
A3D_DEV const uint16_t* dm_scalar(const uint16_t* ptr) {      // wave-uniform by construction; say so
  const uint64_t a = (uint64_t)(uintptr_t)ptr;
  return (const uint16_t*)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a));
}

// one thread of the workgroup books an event of the diagnostics counters (cold paths only)
A3D_DEV void dm_count(const AttnParams& p, int which) {
#ifndef A3D_EXP_R5_PATHS          // (measurement build: the round-5 kernels without the counters, for the same-box A/B of the level-0 launch)
  if (p.counters != nullptr && threadIdx.x == 0) atomicAdd(p.counters + which, 1u);
#endif
}

A3D_DEV uint32_t fa_lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

}  // namespace

// flash_attn_dm.hip / flash_attn_dm80.hip (one definition per storage type; internal to the library: hidden visibility)
__attribute__((visibility("hidden"))) int A3D_FN(a3d_launch_flash_dm)(int flags, int groups, hipStream_t s, const AttnParams& p);
__attribute__((visibility("hidden"))) int A3D_FN(a3d_launch_flash_dm80)(int flags, int groups, hipStream_t s, const AttnParams& p);
__attribute__((visibility("hidden"))) int A3D_FN(a3d_launch_flash_dm160)(int flags, int groups, hipStream_t s, const AttnParams& p);
// cross_attn.hip: two short key sets (text + IP tokens), head_dim 40; A3D_EUNSUPPORTED = not this kernel's shape
__attribute__((visibility("hidden"))) int A3D_FN(a3d_launch_cross_attn40)(int groups, hipStream_t s, const AttnParams& p);
