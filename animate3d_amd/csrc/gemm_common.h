// Shared pieces of the GEMM / implicit-GEMM convolution kernels (gemm_conv.hip: 128 x 128 tiles and the dispatch; gemm_pp.hip: persistent
// 256 x (NB*64) tiles): launch parameters, epilogue arithmetic, LDS-DMA helpers and the LDS-transposing epilogue of the persistent tiles.
#pragma once
#include "common.h"

struct GemmParams {
  const uint16_t* X; int64_t ldx;
  const uint16_t* W; int64_t ldw;
  const float* bias;
  const uint16_t* rowbias; int64_t rb_div;
  const uint16_t* R; int64_t ldr;
  uint16_t* Y; int64_t ldy;
  int64_t M, N, K;
  float alpha, beta;
  int vec16;            // output / residual / rowbias rows allow 16-byte accesses
  int out_f32;          // 128x128 kernel only: Y is float (attention logits of the VAE mid block must not be rounded to bf16)
  int abl;              // -DA3D_ABLATIONS builds only (timing experiments, results wrong): bit 0 = every tile stores to output rows 0..255
                        // (writes stay in L2), bit 1 = no output stores
  // conv geometry (CONV only)
  int B, H, Wd, Cin, Ho, Wo, stride, up, He, We;   // He x We: extent of the (virtual) upsampled image of the up2x conv
  int64_t tiles_n, tiles_m;
  // split-K of the persistent kernel (small M, long K: levels 2 / 3 leave most of the chip idle at one 256-row tile per item): the K-tiles
  // are dealt to `ksplit` work items per output tile, each writes its fp32 accumulators to `ws` in register order, a second kernel adds
  // the slices in index order and applies the epilogue arithmetic (splitk_reduce_kernel, gemm_pp.hip).  ksplit <= 1: off.
  int ksplit, nk_item;   // K-tiles (64 wide) per work item
  float* ws; int64_t ws_bytes;
  // two-source A operand of the dense persistent kernel (a3d_gemm2: the 1x1 shortcut convolution of an up-block ResNet reads [hidden | skip]
  // without a torch.cat): columns [0, K1) of the contraction come from X (row stride ldx), [K1, K) from X2 (row stride ldx2).  X2 == nullptr: off.
  const uint16_t* X2; int64_t ldx2; int64_t K1;
  int direct;           // persistent dense kernel: the direct epilogue (A3D_GEMM_DIRECT; gemm_common.h: direct_epilogue)
};

constexpr int EPI_LINEAR = 0, EPI_GEGLU = 1;

// the per-call `flags` word of a3d_gemm / a3d_gemm_geglu / a3d_conv3x3 (include/animate3d_hip.h)
static inline int a3d_gemm_kernel_of(int flags) { return flags & A3D_GEMM_KERNEL_MASK; }
static inline int a3d_gemm_reserved_cus_of(int flags) { return flags & A3D_GEMM_RESERVED_CUS_MASK; }

// alpha * v and + beta * r with the roundings pinned (no compiler-chosen fma contraction), so that every kernel variant
// produces bit-identical outputs for any alpha (the AlphaBlender mix uses alpha = sigmoid(mix_factor))
A3D_DEV float epi_scale(float v, float alpha) { return __fmul_rn(v, alpha); }
A3D_DEV float epi_axpy(float v, float beta, float r) { return __fmaf_rn(beta, r, v); }

// erf-GELU with Abramowitz-Stegun 7.1.26 (|erf error| < 1.5e-7, far below bf16 resolution): 2 transcendentals +
// ~10 VALU instead of libm erff's ~25-instruction polynomial ladder — this runs in a GEMM epilogue.
A3D_DEV float gelu_erf(float gte) {
  const float x = fabsf(gte) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = __builtin_amdgcn_exp2f(-x * x * 1.4426950408889634f);
  const float erfx = fmaf(-poly, e, 1.0f);                     // erf(|g|/sqrt2)
  const float hg = __fmul_rn(0.5f, gte);            // 0.5 g (1 + erf): roundings pinned, identical in every kernel variant
  // erf is odd: hg * erf(g/sqrt2) = |hg| * erf(|g|/sqrt2) exactly (same product, same fma rounding as the sign-select form it replaces:
  // two VALU per value less in an epilogue that is as long as the K = 320 main loop)
  return __fmaf_rn(fabsf(hg), erfx, hg);
}

// The same arithmetic on a pair of values held as a 64-bit register pair: the polynomial runs on v_pk_fma_f32 / v_pk_mul_f32 without the
// register shuffling the compiler's own pairing of scalar code needs (4 v_mov per value in the fused GEGLU epilogue).  Per element the
// operations and roundings are exactly gelu_erf's.
A3D_DEV f32x2_t splat2(float v) { return f32x2_t{v, v}; }
A3D_DEV f32x2_t gelu_erf2(f32x2_t g) {
  const f32x2_t x = __builtin_elementwise_abs(g) * 0.70710678118654752f;
  const f32x2_t den = __builtin_elementwise_fma(splat2(0.3275911f), x, splat2(1.0f));
  const f32x2_t t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  f32x2_t poly = __builtin_elementwise_fma(splat2(1.061405429f), t, splat2(-1.453152027f));
  poly = __builtin_elementwise_fma(poly, t, splat2(1.421413741f));
  poly = __builtin_elementwise_fma(poly, t, splat2(-0.284496736f));
  poly = __builtin_elementwise_fma(poly, t, splat2(0.254829592f));
  poly = poly * t;
  const f32x2_t a = (-x) * x * 1.4426950408889634f;
  const f32x2_t e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
  const f32x2_t erfx = __builtin_elementwise_fma(-poly, e, splat2(1.0f));
  const f32x2_t hg = g * 0.5f;
  return __builtin_elementwise_fma(__builtin_elementwise_abs(hg), erfx, hg);
}
// y[0..7] = (h + b_h) * gelu(g + b_g) on four register pairs
A3D_DEV void geglu8(const float (&hv)[8], const float (&gv)[8], const float (&bh)[8], const float (&bg)[8], float (&y)[8]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f32x2_t gg = f32x2_t{gv[2 * e], gv[2 * e + 1]} + f32x2_t{bg[2 * e], bg[2 * e + 1]};
    const f32x2_t hh = f32x2_t{hv[2 * e], hv[2 * e + 1]} + f32x2_t{bh[2 * e], bh[2 * e + 1]};
    const f32x2_t yy = hh * gelu_erf2(gg);
    y[2 * e] = yy[0]; y[2 * e + 1] = yy[1];
  }
}

static __device__ u32x4_t g_zero_page[4];      // 64 zero bytes: source of out-of-image conv taps (device globals are zero-filled)

A3D_DEV uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// LDS-DMA: 64 lanes x 16 B -> LDS[lds_dst + 16 * lane]; the compiler does not count these (s_waitcnt vmcnt by hand)
A3D_DEV void glds16_v(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}
A3D_DEV void glds16_s(uint32_t voff, const void* sbase, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}
// a wave-uniform pointer the compiler computed on the vector ALU (64-bit divisions): move it to scalar registers for an "s" operand
A3D_DEV const uint16_t* scalar_ptr(const uint16_t* ptr) {
  const uint64_t a = (uint64_t)(uintptr_t)ptr;
  return (const uint16_t*)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a));
}
// LDS reads whose ISSUE position matters (software-pipelined operand fetches): volatile keeps instruction selection from sinking them next to
// their use; the explicit LDS address space keeps them ds_read instructions (a volatile access through a generic pointer stays a flat load)
A3D_DEV u32x4_t lds_vload128(const void* p) {
  return *(const volatile __attribute__((address_space(3))) u32x4_t*)(uintptr_t)lds_addr(p);
}
A3D_DEV u32x2_t lds_vload64(const void* p) {
  return *(const volatile __attribute__((address_space(3))) u32x2_t*)(uintptr_t)lds_addr(p);
}
A3D_DEV void wave_lds_fence() {          // orders this wave's LDS writes before its later LDS reads (LDS executes a wave's ops in order)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- epilogue of one 256 x (NB*64) tile held as acc[tn][tm] by 8 waves (4(M) x 2(N); wave (wm, wn) owns rows wm*64 .. +63 and the
//      32-column blocks wblk .. wblk + NB - 2 and wblk_last).  stg: this wave's private 32 x 68 fp32 LDS buffer; bias_lds / rowbias_lds:
//      the tile's bias (fp32, tile columns) and rowbias row (16-bit) as the main loop's DMA left them in LDS.
template <int EPI, int NB, bool RES>
A3D_DEV void persist_epilogue(const GemmParams& p, f32x16_t (&acc)[NB][2], float* const stg, const float* const bias_lds,
                              const uint16_t* const rowbias_lds, const int64_t m0, const int64_t n0, const int wm, const int wblk,
                              const int wblk_last, const int lane) {
  const int l31 = lane & 31, g = lane >> 5;
  // ---- epilogue: per 32-row half and <= 64-column pass, transpose through a wave-private LDS buffer so that each
  //      lane owns 8 consecutive output columns (16-byte bias / rowbias / residual / output accesses)
  constexpr int SROW = 68;
  constexpr int NP = (NB + 1) / 2;                       // passes of <= 64 columns per 32-row half
  u32x4_t rres[RES ? 2 : 1][RES ? 4 : 1];
  auto pass_cols = [&](int ps) { return (2 * ps + 1 < NB) ? 64 : 32; };
  auto pass_col0 = [&](int ps) { return ((2 * ps + 1 < NB) ? wblk + 2 * ps : wblk_last) * 32; };   // first tile column of pass ps
  auto load_res = [&](int pi, int slot) {
    if constexpr (RES) {
      const int tm = pi / NP, ps = pi % NP;
      const int64_t mbase = m0 + wm * 64 + tm * 32;
      const int64_t nbase = n0 + pass_col0(ps);
      if (pass_cols(ps) == 64) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t m = mbase + 8 * j + (lane >> 3);
          rres[slot][j] = *reinterpret_cast<const u32x4_t*>(p.R + m * p.ldr + nbase + 8 * (lane & 7));
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int64_t m = mbase + 16 * j + (lane >> 2);
          rres[slot][j] = *reinterpret_cast<const u32x4_t*>(p.R + m * p.ldr + nbase + 8 * (lane & 3));
        }
      }
    }
  };
  if constexpr (RES) load_res(0, 0);
  // Epilogues without a residual (and the fused GEGLU) apply bias / rowbias / alpha (GEGLU: the whole h * gelu(gate)) in the MFMA
  // layout — a lane holds 4 consecutive columns of one row per register quad, the per-column vectors are broadcast LDS reads —
  // and send ROUNDED 16-bit values through the transposition buffer: a quarter (GEGLU) / half of the fp32 staging traffic, whose
  // ds_write_b128 rate (~80 B/clk per CU) made the transposition the longest part of a K = 320 tile's epilogue.  Arithmetic and
  // rounding per element are exactly those of the fp32-staged path below (kept for residual epilogues: one rounding after the add).
  if constexpr (EPI == EPI_GEGLU || !RES) {
    // Staged rows are unpadded (64 / 128 bytes) and swizzled instead: the 16-byte chunk index is XORed with row bits and the two
    // 8-byte halves of a chunk trade places on rows with bit 3 set.  ds_write_b64 is served 16 consecutive lanes (= rows, same
    // column) at a time over 32 banks and ds_read_b128 in the lane groups {0-3,12-15,20-27} ... over 64: both come out
    // conflict-free (the padded layout of before was 2-way on the stores and overlapped rows on the loads: SQ_LDS_BANK_CONFLICT
    // of a K = 640 launch was a quarter of its LDS cycles, profiles/r4_gemm_pp_pmc_sq.md).
    constexpr int RS = (EPI == EPI_GEGLU) ? 64 : 128;
    char* const stg16 = reinterpret_cast<char*>(stg);
    const int wsw = (EPI == EPI_GEGLU) ? ((l31 >> 1) & 3) : (l31 & 7);      // chunk swizzle of the row this lane stores
    const int whalf = 8 * (g ^ ((l31 >> 3) & 1));
    auto unswap = [](u32x4_t o, int row) {
      const bool sw = row & 8;
      u32x4_t r;
      r[0] = sw ? o[2] : o[0]; r[1] = sw ? o[3] : o[1]; r[2] = sw ? o[0] : o[2]; r[3] = sw ? o[1] : o[3];
      return r;
    };
    // Round 5: the epilogue was LDS-LATENCY bound, not bandwidth bound: the compiler issued {broadcast read of a bias quad, s_waitcnt lgkmcnt(0),
    // ~10 VALU, ds_write_b64} eight times per pass and {ds_read_b128, s_waitcnt, global_store} four times — twelve LDS round trips in series,
    // 6 passes per tile = the ~8 k cycles the K = 320 ablations attribute to "the epilogue's own instruction stream" (profiles/README.md).
    // Now (a) the bias / rowbias operands of quad i + 1 are requested before quad i is computed (8 more live registers: all the epilogue's
    // first pass has to spare next to a 160-register accumulator), (b) a pass's transposed rows are read back in one batch before its stores,
    // (c) consecutive passes alternate between two 4 KB staging buffers, so a pass's read-back / stores and the next pass's arithmetic overlap
    // (one wave-level fence per pass instead of two).  Arithmetic and rounding per element are unchanged (bit-identical outputs).
    constexpr int BUF2 = 32 * RS;                       // second staging buffer of this wave (2 x 4 KB <= its 8.5 KB region)
#pragma unroll
    for (int pi = 0; pi < 2 * NP; ++pi) {
      const int tm = pi / NP, ps = pi % NP;
      const int ncol = pass_cols(ps);
      const int64_t mbase = m0 + wm * 64 + tm * 32;
      const int64_t nbase = n0 + pass_col0(ps);
      char* const sb = stg16 + (pi & 1) * BUF2;
      if constexpr (EPI == EPI_GEGLU) {
        // operands of one step (register quads 2 q2, 2 q2 + 1 of h and of the gate): four float4 broadcast reads
        u32x4_t bq[2][4];
        auto fetch = [&](int q2, int slot) __attribute__((always_inline)) {       // (the LDS bias image is zero-filled when there is no bias)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int c0 = 8 * (2 * q2 + h) + 4 * g;
            bq[slot][2 * h] = lds_vload128(bias_lds + pass_col0(ps) + c0);      // volatile: see the linear epilogue below
            bq[slot][2 * h + 1] = lds_vload128(bias_lds + pass_col0(ps) + 32 + c0);
          }
        };
        fetch(0, 0);
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {                      // register quads 2 q2, 2 q2 + 1: columns 16 q2 + 4 g + {0..3} and + 8
          if (q2 + 1 < 2) fetch(q2 + 1, (q2 + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);                    // (the scheduler otherwise sinks the reads next to their use)
          float hv[8], gv[8], bh[8], bg[8], y[8];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const u32x4_t b0 = bq[q2 & 1][2 * h], b1 = bq[q2 & 1][2 * h + 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) { bh[4 * h + j] = __uint_as_float(b0[j]); bg[4 * h + j] = __uint_as_float(b1[j]); }
#pragma unroll
            for (int j = 0; j < 4; ++j) { hv[4 * h + j] = acc[2 * ps][tm][4 * (2 * q2 + h) + j]; gv[4 * h + j] = acc[2 * ps + 1][tm][4 * (2 * q2 + h) + j]; }
          }
          geglu8(hv, gv, bh, bg, y);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            u32x2_t o;
            o[0] = pack16(y[4 * h], y[4 * h + 1]);
            o[1] = pack16(y[4 * h + 2], y[4 * h + 3]);
            *reinterpret_cast<u32x2_t*>(sb + l31 * RS + 16 * ((2 * q2 + h) ^ wsw) + whalf) = o;
          }
        }
        wave_lds_fence();
        const int cc = lane & 3;
        const int64_t oc = nbase / 2 + 8 * cc;
        u32x4_t ob[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int row = 16 * j + (lane >> 2);
          ob[j] = *reinterpret_cast<const u32x4_t*>(sb + row * RS + 16 * (cc ^ ((row >> 1) & 3)));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int row = 16 * j + (lane >> 2);
          const int64_t m = mbase + row;
          const u32x4_t o = unswap(ob[j], row);
#ifdef A3D_ABLATIONS
          if (p.abl & 2) { asm volatile("" :: "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3])); continue; }
          *reinterpret_cast<u32x4_t*>(p.Y + ((p.abl & 1) ? (m & 255) : m) * p.ldy + oc) = o;
#else
          *reinterpret_cast<u32x4_t*>(p.Y + m * p.ldy + oc) = o;
#endif
        }
      } else {
        constexpr int NQ = 8;                                   // quads of a 64-column pass (tl, q); a 32-column pass uses the first four
        const int nq = (2 * ps + 1 < NB) ? 8 : 4;
        u32x4_t bq[3];
        u32x2_t tq[3];
        auto fetch = [&](int idx, int slot) __attribute__((always_inline)) {
          const int c0 = (idx >> 2) * 32 + 8 * (idx & 3) + 4 * g;   // column within the pass
          // unconditional: the kernel zero-fills the LDS bias / rowbias images once when the launch has none (a branch per operand and
          // quad made the compiler wait for the read it had just issued: s_waitcnt merges its counters at every join)
          // volatile: these reads provably do not alias the staging stores, so nothing but the volatile ordering (against the
          // sched_barrier below) keeps instruction selection from sinking them next to their use, one exposed LDS round trip per quad
          bq[slot] = lds_vload128(bias_lds + pass_col0(ps) + c0);
          tq[slot] = lds_vload64(rowbias_lds + pass_col0(ps) + c0);
        };
        // a launch with neither operand (the bias-free Q|K|V projections: the largest N of the step) skips the reads: they are broadcast
        // ds_read_b128, 8 LDS cycles each whatever they deliver, and the LDS — shared by the eight waves that run the epilogue at the
        // same time — is what the epilogue is bound by (round 5: pipelining the reads changed nothing, leaving them out is worth 3 %)
        auto quads = [&](auto with_c) __attribute__((always_inline)) {
          constexpr bool WITH = decltype(with_c)::value;
          if constexpr (WITH) { fetch(0, 0); fetch(1, 1); }
#pragma unroll
          for (int idx = 0; idx < NQ; ++idx) {
            if (idx >= nq) continue;
            const int tl = idx >> 2, q = idx & 3;
            const int tn = 2 * ps + tl;
            if constexpr (WITH) {
              if (idx + 2 < nq) fetch(idx + 2, (idx + 2) % 3);      // two quads ahead: an LDS round trip is ~2 quads of arithmetic
              __builtin_amdgcn_sched_barrier(0);                    // (the scheduler otherwise sinks the reads next to their use)
            }
            float v[4] = {acc[tn][tm][4 * q], acc[tn][tm][4 * q + 1], acc[tn][tm][4 * q + 2], acc[tn][tm][4 * q + 3]};
            if constexpr (WITH) {
              const u32x4_t b = bq[idx % 3];
              v[0] += __uint_as_float(b[0]); v[1] += __uint_as_float(b[1]); v[2] += __uint_as_float(b[2]); v[3] += __uint_as_float(b[3]);
              const u32x2_t tb = tq[idx % 3];
              v[0] += lo16(tb[0]); v[1] += hi16(tb[0]); v[2] += lo16(tb[1]); v[3] += hi16(tb[1]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = epi_scale(v[e], p.alpha);
            u32x2_t o;
            o[0] = pack16(v[0], v[1]);
            o[1] = pack16(v[2], v[3]);
            *reinterpret_cast<u32x2_t*>(sb + l31 * RS + 16 * ((4 * tl + q) ^ wsw) + whalf) = o;
            if constexpr (WITH) __builtin_amdgcn_sched_barrier(0);
          }
        };
        if (p.bias || p.rowbias) quads(std::true_type{});
        else quads(std::false_type{});
        wave_lds_fence();
        const int lpr = ncol / 8;
        const int cc = lane & (lpr - 1);
        u32x4_t ob[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j * (64 / lpr) >= 32) continue;
          const int row = (64 / lpr) * j + lane / lpr;
          ob[j] = *reinterpret_cast<const u32x4_t*>(sb + row * RS + 16 * (cc ^ (row & 7)));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j * (64 / lpr) >= 32) continue;
          const int row = (64 / lpr) * j + lane / lpr;
          const int64_t m = mbase + row;
          const u32x4_t o = unswap(ob[j], row);
#ifdef A3D_ABLATIONS
          if (p.abl & 2) { asm volatile("" :: "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3])); continue; }
          *reinterpret_cast<u32x4_t*>(p.Y + ((p.abl & 1) ? (m & 255) : m) * p.ldy + nbase + 8 * cc) = o;
#else
          *reinterpret_cast<u32x4_t*>(p.Y + m * p.ldy + nbase + 8 * cc) = o;
#endif
        }
      }
      // (no second fence: the next pass writes the OTHER staging buffer; the one after next is separated from this pass's reads by that pass's fence)
    }
    return;
  }
#pragma unroll
  for (int pi = 0; pi < 2 * NP; ++pi) {
    const int tm = pi / NP, ps = pi % NP;
    const int ncol = pass_cols(ps);
    if constexpr (RES) { if (pi + 1 < 2 * NP) load_res(pi + 1, (pi + 1) & 1); }
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
      const int tn = 2 * ps + tl;
      if (tn < NB) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 v;
          v.x = acc[tn][tm][4 * q]; v.y = acc[tn][tm][4 * q + 1]; v.z = acc[tn][tm][4 * q + 2]; v.w = acc[tn][tm][4 * q + 3];
          *reinterpret_cast<float4*>(stg + l31 * SROW + tl * 32 + 8 * q + 4 * g) = v;
        }
      }
    }
    wave_lds_fence();
    const int64_t mbase = m0 + wm * 64 + tm * 32;
    const int64_t nbase = n0 + pass_col0(ps);               // first column of this pass

    if constexpr (EPI == EPI_GEGLU) {
      // NB is even here: columns [0,32) of the pass are h, [32,64) the matching gates
      const int cc = lane & 3;
      const int64_t oc = nbase / 2 + 8 * cc;
      float bh[8], bg[8];
      {
        const float* bl = bias_lds + (pass_col0(ps) + 8 * cc);
#pragma unroll
        for (int e = 0; e < 8; ++e) { bh[e] = bl[e]; bg[e] = bl[32 + e]; }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = 16 * j + (lane >> 2);
        const int64_t m = mbase + row;
        const float4 h0 = *reinterpret_cast<const float4*>(stg + row * SROW + 8 * cc);
        const float4 h1 = *reinterpret_cast<const float4*>(stg + row * SROW + 8 * cc + 4);
        const float4 g0 = *reinterpret_cast<const float4*>(stg + row * SROW + 32 + 8 * cc);
        const float4 g1 = *reinterpret_cast<const float4*>(stg + row * SROW + 32 + 8 * cc + 4);
        const float hv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
        const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        float y[8];
        geglu8(hv, gv, bh, bg, y);
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack16(y[2 * e], y[2 * e + 1]);
#ifdef A3D_ABLATIONS
        if (p.abl & 2) { asm volatile("" :: "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3])); continue; }
        *reinterpret_cast<u32x4_t*>(p.Y + ((p.abl & 1) ? (m & 255) : m) * p.ldy + oc) = o;
#else
        *reinterpret_cast<u32x4_t*>(p.Y + m * p.ldy + oc) = o;
#endif
      }
    } else {
      const int lpr = ncol / 8;                                // lanes per row: 8 (64 columns) or 4 (32 columns)
      const int cc = lane & (lpr - 1);
      const int64_t n = nbase + 8 * cc;
      const int ncl = pass_col0(ps) + 8 * cc;                   // column within the tile
      float bv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[e] = bias_lds[ncl + e];          // (zero-filled images when the launch has no bias / rowbias)
      const u32x4_t tb = *reinterpret_cast<const u32x4_t*>(rowbias_lds + ncl);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j * (64 / lpr) >= 32) continue;                    // 32-column pass: two row groups of 16
        const int row = (64 / lpr) * j + lane / lpr;
        const int64_t m = mbase + row;
        const float4 a0 = *reinterpret_cast<const float4*>(stg + row * SROW + 8 * cc);
        const float4 a1 = *reinterpret_cast<const float4*>(stg + row * SROW + 8 * cc + 4);
        float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bv[e];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] += lo16(tb[e]); v[2 * e + 1] += hi16(tb[e]); }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = epi_scale(v[e], p.alpha);
        if constexpr (RES) {
          const u32x4_t tr = rres[pi & 1][j];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[2 * e] = epi_axpy(v[2 * e], p.beta, lo16(tr[e])); v[2 * e + 1] = epi_axpy(v[2 * e + 1], p.beta, hi16(tr[e])); }
        }
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack16(v[2 * e], v[2 * e + 1]);
#ifdef A3D_ABLATIONS
        if (p.abl & 2) { asm volatile("" :: "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3])); continue; }
        *reinterpret_cast<u32x4_t*>(p.Y + ((p.abl & 1) ? (m & 255) : m) * p.ldy + n) = o;
#else
        *reinterpret_cast<u32x4_t*>(p.Y + m * p.ldy + n) = o;
#endif
      }
    }
    wave_lds_fence();
  }
}

// ---- Direct epilogue (round 6): no LDS transposition.  The kernel stages W so that MFMA row i = 8 b + 4 g + c of a
//      32-row block is output column 16 g + 4 b + c of that block (a permutation of the DMA SOURCE rows, gemm_pp.hip: issue_pieces): register
//      r of a lane is then column 16 g + r of ONE output row (l31) — 16 consecutive columns, two 16-byte stores per lane and 32 x 32 tile; bias
//      (fp32, broadcast LDS reads), rowbias and the residual are read in the same layout.  Arithmetic per element exactly as in
//      persist_epilogue (((acc + bias) + rowbias) * alpha, then + beta * r, one rounding): bit-identical outputs.
template <int NB, bool RES>
A3D_DEV void direct_epilogue(const GemmParams& p, f32x16_t (&acc)[NB][2], const float* const bias_lds, const uint16_t* const rowbias_lds,
                             const int64_t m0, const int64_t n0, const int wm, const int wblk, const int wblk_last, const int lane) {
  const int l31 = lane & 31, g = lane >> 5;
  const bool with = p.bias || p.rowbias;            // launch-uniform: the bias-free Q|K|V projections skip the operand reads
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int64_t m = m0 + wm * 64 + tm * 32 + l31;
    uint16_t* const yrow = p.Y + m * p.ldy + n0;
    u32x4_t rr[RES ? NB : 1][2];
    if constexpr (RES) {
      const uint16_t* const rrow = p.R + m * p.ldr + n0;
#pragma unroll
      for (int tn = 0; tn < NB; ++tn) {
        const int col = (tn < NB - 1 ? wblk + tn : wblk_last) * 32 + 16 * g;
        rr[tn][0] = *reinterpret_cast<const u32x4_t*>(rrow + col);
        rr[tn][1] = *reinterpret_cast<const u32x4_t*>(rrow + col + 8);
      }
    }
#pragma unroll
    for (int tn = 0; tn < NB; ++tn) {
      const int col = (tn < NB - 1 ? wblk + tn : wblk_last) * 32 + 16 * g;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[tn][tm][r];
      if (with) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const u32x4_t b = lds_vload128(bias_lds + col + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * q + e] += __uint_as_float(b[e]);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const u32x4_t tb = lds_vload128(rowbias_lds + col + 8 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[8 * h + 2 * e] += lo16(tb[e]); v[8 * h + 2 * e + 1] += hi16(tb[e]); }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = epi_scale(v[r], p.alpha);
      if constexpr (RES) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[8 * h + 2 * e] = epi_axpy(v[8 * h + 2 * e], p.beta, lo16(rr[tn][h][e]));
            v[8 * h + 2 * e + 1] = epi_axpy(v[8 * h + 2 * e + 1], p.beta, hi16(rr[tn][h][e]));
          }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack16(v[8 * h + 2 * e], v[8 * h + 2 * e + 1]);
        *reinterpret_cast<u32x4_t*>(yrow + col + 8 * h) = o;
      }
    }
  }
}

// gemm_pp.hip: the persistent 256 x (nb * 64) tile kernel (conv 0 | 1 | 2, epi EPI_*, nb 4 | 5); the caller (try_launch_persist,
// gemm_conv.hip) has filled tiles_m / tiles_n (and ksplit / nk_item / ws for a split-K launch) and checked the shape
__attribute__((visibility("hidden"))) int A3D_FN(a3d_launch_gemm_pp)(int conv, int epi, int nb, hipStream_t stream, const GemmParams& p, int cus);
// gemm_ring.hip: the LDS-DMA ring kernel for small dense GEMMs (128 x (64 nb) tiles, nb = 2 | 4 | 5; EPI_LINEAR); the caller has checked the
// shape and filled tiles_m / tiles_n
__attribute__((visibility("hidden"))) int A3D_FN(a3d_launch_gemm_ring)(int nb, hipStream_t stream, const GemmParams& p, int cus);
