// GroupNorm (+SiLU) and LayerNorm (+positional-embedding adds) for NHWC rows, gfx950.
// Both are HBM-bound: 16-byte accesses, one read pass for statistics + one read/write pass.
#include "common.h"

namespace {

constexpr int GN_ROWS_PER_BLOCK = 128;   // rows of one instance handled by one workgroup

// threads: x = 16-byte channel chunk (C/8 of them), y = row lane
struct GNParams {
  const uint16_t* X; uint16_t* Y; const float* gamma; const float* beta;
  // two-source input (a3d_group_norm2: the up blocks' torch.cat([hidden, skip], 1) never materialised): channels [0, C1) come from X (row
  // stride C1), channels [C1, C) from Xb (row stride C - C1); Xb == nullptr: one source with row stride C
  const uint16_t* Xb; int C1;
  float* partial;   // [B][nchunk][groups][2]
  float* stats;     // [B][groups][2]  (mean, rstd)
  double* sums;     // [B][groups][2]  (sum, sum of squares): written INSTEAD of stats when non-null (split call, frame-sharded 3-D norm)
  int B; int64_t rows; int C; int groups; int cg; int nchunk; float eps; int silu;
};

__global__ void gn_partial_kernel(const GNParams p) {
  // dynamic LDS: per-thread partials [ny][c8][4] = (sum_lo, sq_lo, sum_hi, sq_hi); reduced in a fixed order
  // (deterministic: no atomics), one thread per group.
  extern __shared__ float spart[];
  const int c8 = threadIdx.x, ry = threadIdx.y, nx = blockDim.x, ny = blockDim.y;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int64_t r0 = (int64_t)chunk * GN_ROWS_PER_BLOCK;
  const int64_t r1 = (r0 + GN_ROWS_PER_BLOCK < p.rows) ? r0 + GN_ROWS_PER_BLOCK : p.rows;
  const int ch0 = c8 * 8;
  const int g_lo = ch0 / p.cg;                              // a chunk touches at most 2 groups (cg >= 4)
  const int split = (g_lo + 1) * p.cg - ch0;                // first `split` channels belong to g_lo
  float s_lo = 0.f, q_lo = 0.f, s_hi = 0.f, q_hi = 0.f;
  const bool second = p.Xb != nullptr && ch0 >= p.C1;
  const int64_t ldsrc = p.Xb == nullptr ? p.C : (second ? p.C - p.C1 : p.C1);
  const uint16_t* base = (second ? p.Xb + (ch0 - p.C1) : p.X + ch0) + ((int64_t)b * p.rows) * ldsrc;
  auto accum = [&](const u32x4_t& v) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (j & 1) ? hi16(v[j >> 1]) : lo16(v[j >> 1]);
      if (j < split) { s_lo += f; q_lo += f * f; } else { s_hi += f; q_hi += f * f; }
    }
  };
  int64_t r = r0 + ry;
  for (; r + 3 * ny < r1; r += 4 * ny) {          // four independent row loads in flight per thread, accumulated in row order
    u32x4_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ld_stream(base + (r + (int64_t)k * ny) * ldsrc);
#pragma unroll
    for (int k = 0; k < 4; ++k) accum(v[k]);
  }
  for (; r < r1; r += ny) accum(ld_stream(base + r * ldsrc));
  float* mine = spart + ((size_t)ry * nx + c8) * 4;
  mine[0] = s_lo; mine[1] = q_lo; mine[2] = s_hi; mine[3] = q_hi;
  __syncthreads();
  const int t = ry * nx + c8;
  if (t < p.groups) {
    const int grp = t;
    const int cfirst = (grp * p.cg) / 8, clast = ((grp + 1) * p.cg - 1) / 8;    // chunks touching this group
    float s = 0.f, q = 0.f;
    for (int y = 0; y < ny; ++y)
      for (int c = cfirst; c <= clast; ++c) {
        const float* src = spart + ((size_t)y * nx + c) * 4;
        const int lo_grp = (c * 8) / p.cg;
        if (lo_grp == grp) { s += src[0]; q += src[1]; } else { s += src[2]; q += src[3]; }
      }
    float* out = p.partial + (((int64_t)b * p.nchunk + chunk) * p.groups + grp) * 2;
    out[0] = s; out[1] = q;
  }
}

__global__ void gn_finalize_kernel(const GNParams p) {
  // one block per instance b: thread (grp, sub) sums chunks sub, sub + nsub, ... (a single thread per group walked all
  // nchunk partials as one dependent load chain: 512 chunks = 0.1 ms of pure latency for the per-video 3-D norm), then
  // the nsub partial sums of a group are added in a fixed order (deterministic) in fp64.
  extern __shared__ double sred[];          // [nsub][groups][2]
  const int b = blockIdx.x, grp = threadIdx.x, sub = threadIdx.y, nsub = blockDim.y;
  double s = 0.0, q = 0.0;
  if (grp < p.groups) {
    const float* in = p.partial + (int64_t)b * p.nchunk * p.groups * 2 + grp * 2;
    for (int c = sub; c < p.nchunk; c += nsub) {
      const float2 v = *reinterpret_cast<const float2*>(in + (int64_t)c * p.groups * 2);
      s += v.x; q += v.y;
    }
    sred[((size_t)sub * p.groups + grp) * 2] = s;
    sred[((size_t)sub * p.groups + grp) * 2 + 1] = q;
  }
  __syncthreads();
  if (sub != 0 || grp >= p.groups) return;
  s = 0.0; q = 0.0;
  for (int k = 0; k < nsub; ++k) { s += sred[((size_t)k * p.groups + grp) * 2]; q += sred[((size_t)k * p.groups + grp) * 2 + 1]; }
  if (p.sums) {
    p.sums[((int64_t)b * p.groups + grp) * 2 + 0] = s;
    p.sums[((int64_t)b * p.groups + grp) * 2 + 1] = q;
    return;
  }
  const double n = (double)p.rows * p.cg;
  const double mean = s / n;
  double var = q / n - mean * mean;
  if (var < 0.0) var = 0.0;
  p.stats[((int64_t)b * p.groups + grp) * 2 + 0] = (float)mean;
  p.stats[((int64_t)b * p.groups + grp) * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
}

__global__ void gn_apply_kernel(const GNParams p) {
  const int c8 = threadIdx.x, ry = threadIdx.y, ny = blockDim.y;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int64_t r0 = (int64_t)chunk * GN_ROWS_PER_BLOCK;
  const int64_t r1 = (r0 + GN_ROWS_PER_BLOCK < p.rows) ? r0 + GN_ROWS_PER_BLOCK : p.rows;
  const int ch0 = c8 * 8;
  float sc[8], sh[8];       // y = x * sc + sh
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = ch0 + j, grp = ch / p.cg;
    const float mean = p.stats[((int64_t)b * p.groups + grp) * 2], rstd = p.stats[((int64_t)b * p.groups + grp) * 2 + 1];
    const float ga = p.gamma[ch], be = p.beta[ch];
    sc[j] = rstd * ga;
    sh[j] = be - mean * rstd * ga;
  }
  const int64_t off = ((int64_t)b * p.rows) * p.C + ch0;
  const bool second = p.Xb != nullptr && ch0 >= p.C1;
  const int64_t ldsrc = p.Xb == nullptr ? p.C : (second ? p.C - p.C1 : p.C1);
  const uint16_t* const src = (second ? p.Xb + (ch0 - p.C1) : p.X + ch0) + ((int64_t)b * p.rows) * ldsrc;
  auto apply = [&](const u32x4_t& v, int64_t r) {
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f[j] = ((j & 1) ? hi16(v[j >> 1]) : lo16(v[j >> 1])) * sc[j] + sh[j];
      if (p.silu) f[j] = f[j] / (1.f + __expf(-f[j]));
    }
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack16(f[2 * j], f[2 * j + 1]);
    st_stream(p.Y + off + r * p.C, o);
  };
  int64_t r = r0 + ry;
  for (; r + 3 * ny < r1; r += 4 * ny) {          // four independent row loads in flight per thread
    u32x4_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ld_stream(src + (r + (int64_t)k * ny) * ldsrc);
#pragma unroll
    for (int k = 0; k < 4; ++k) apply(v[k], r + (int64_t)k * ny);
  }
  for (; r < r1; r += ny) apply(ld_stream(src + r * ldsrc), r);
}

// ---------------- GroupNorm in ONE launch and ONE read of the tensor (round 6): a workgroup owns a slab of `sg` whole groups of one instance
// for ALL its rows and keeps it in registers between the statistics and the apply phase (up to 512 threads x GNF_RMAX 16-byte chunks =
// 172 KB).  Covers the 2-D GroupNorms of levels 1-3 (80-320-channel slabs; level 0 too at a 32 x 32 latent) and the per-video 3-D norm
// of level 3; the three-launch path above stays for the instances that do not fit (level 0 at a 64 x 64 latent: 4 096 rows, the per-video
// norms of levels 0-2).  Traffic 1 read + 1 write instead of 2 + 1; no workspace; and the small instances of a multi-GPU rank (16 images x 256
// rows x 1280 channels was 3 launches of 32 workgroups whose threads each walked 128 rows serially: 72 us) become one launch.
// Statistics: per-thread fp32 sums over <= GNF_RMAX rows in row order, fixed-order LDS reduction, fp64 combine (deterministic; does not
// depend on B, so a batch slice reproduces the full batch bit for bit).
constexpr int GNF_RMAX = 21;             // largest instantiation: 512 threads x 21 chunks = 172 KB per workgroup (at 41 — level 0's 4 096 rows x 40
                                         // channels — the compiler spills 155 of 256 registers)
template <bool STREAM, int RMAX>
__global__ __launch_bounds__(512) void gn_fused_kernel(const GNParams p, const int sg, const int nx, const int ny, const int nslab) {
  extern __shared__ float sm[];                 // [ny][nx][4] thread partials | [8][nx][4] stage-A sums | [sg][2] (mean, rstd)
  const int tid = threadIdx.x;
  const int ry = tid / nx, c8 = tid - ry * nx;
  const bool act = ry < ny;
  const uint32_t lid = (uint32_t)xcd_remap32(blockIdx.x, gridDim.x);      // neighbouring slabs of an image share 128-byte lines: same XCD, same L2
  const int b = (int)(lid / (uint32_t)nslab), slab = (int)(lid - (uint32_t)b * (uint32_t)nslab);
  const int chs = slab * sg * p.cg;             // first channel of the slab
  const int ch0 = chs + c8 * 8;
  const int g_lo = (c8 * 8) / p.cg;             // (local group index) a chunk touches at most 2 groups
  const int split = (g_lo + 1) * p.cg - c8 * 8;
  const bool second = p.Xb != nullptr && ch0 >= p.C1;
  const int64_t ldsrc = p.Xb == nullptr ? p.C : (second ? p.C - p.C1 : p.C1);
  const uint16_t* const src = (second ? p.Xb + (ch0 - p.C1) : p.X + ch0) + ((int64_t)b * p.rows) * ldsrc;
  // STREAM: full-line slabs (non-temporal accesses); slabs narrower than a 128-byte line share lines with their neighbours: plain accesses keep them in L2
  // this thread's rows are row0, row0 + ny, ...: nv of them (<= RMAX); 32-bit element offsets (an instance that fits the registers is < 2^31
  // bytes).  Branch-free loads: iterations beyond nv re-read the thread's last valid row and are zeroed.
  const uint32_t nrows = (uint32_t)p.rows, step = (uint32_t)ny;
  uint32_t nv = (act && (uint32_t)ry < nrows) ? (nrows - (uint32_t)ry + step - 1u) / step : 0u;
  const uint32_t rbase = nv ? (uint32_t)ry : 0u;
  const uint16_t* const tsrc = src + rbase * (uint32_t)ldsrc;
  const uint32_t sstride = step * (uint32_t)ldsrc;
  u32x4_t v[RMAX];
  {
    const uint32_t last = nv ? nv - 1u : 0u;
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      const uint32_t idx = (uint32_t)i < last ? (uint32_t)i : last;
      v[i] = STREAM ? ld_stream(tsrc + idx * sstride) : *reinterpret_cast<const u32x4_t*>(tsrc + idx * sstride);
    }
  }
#pragma unroll
  for (int i = 0; i < RMAX; ++i) {
    const bool ok = (uint32_t)i < nv;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[i][e] = ok ? v[i][e] : 0u;
  }
  // per-column sums over the thread's rows in row order (rows beyond the instance are zeros: they add nothing), then the columns of the
  // chunk go to its one or two groups
  float cs[8], cq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { cs[j] = 0.f; cq[j] = 0.f; }
#pragma unroll
  for (int i = 0; i < RMAX; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (j & 1) ? hi16(v[i][j >> 1]) : lo16(v[i][j >> 1]);
      cs[j] += f; cq[j] += f * f;
    }
    __builtin_amdgcn_sched_barrier(0);          // row by row: the scheduler otherwise converts every row up front (8 extra registers per row)
  }
  float s_lo = 0.f, q_lo = 0.f, s_hi = 0.f, q_hi = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < split) { s_lo += cs[j]; q_lo += cq[j]; } else { s_hi += cs[j]; q_hi += cq[j]; }
  }
  // the apply phase converts the raw 16-bit words again: without this fence the compiler keeps all 8 * RMAX converted floats alive
  // across the reduction (common subexpressions of the two phases) and spills ~480 registers
#pragma unroll
  for (int i = 0; i < RMAX; ++i) asm volatile("" : "+v"(v[i]));
  asm volatile("" : "+v"(nv));                  // (likewise anything derived from the row count)
  float* const part = sm;
  float* const stA = sm + (size_t)ny * nx * 4;
  float* const st = stA + (size_t)8 * nx * 4;
  if (act) { float* mine = part + ((size_t)ry * nx + c8) * 4; mine[0] = s_lo; mine[1] = q_lo; mine[2] = s_hi; mine[3] = q_hi; }
  __syncthreads();
  if (tid < 8 * nx) {                           // stage A: thread (y0, c) adds rows y0, y0 + 8, ... of chunk column c
    const int y0 = tid / nx, c = tid - y0 * nx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 1
    for (int y = y0; y < ny; y += 8) { const float* q = part + ((size_t)y * nx + c) * 4; a0 += q[0]; a1 += q[1]; a2 += q[2]; a3 += q[3]; }
    float* o = stA + ((size_t)y0 * nx + c) * 4; o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
  }
  __syncthreads();
  if (tid < sg) {                               // stage B: one thread per group, fp64 combine
    const int cfirst = (tid * p.cg) / 8, clast = ((tid + 1) * p.cg - 1) / 8;
    double s = 0.0, q = 0.0;
#pragma unroll 1
    for (int y0 = 0; y0 < 8; ++y0)
#pragma unroll 1
      for (int c = cfirst; c <= clast; ++c) {
        const float* a = stA + ((size_t)y0 * nx + c) * 4;
        if ((c * 8) / p.cg == tid) { s += a[0]; q += a[1]; } else { s += a[2]; q += a[3]; }
      }
    const double n = (double)p.rows * p.cg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    st[2 * tid] = (float)mean;
    st[2 * tid + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
  }
  __syncthreads();
  if (!act) return;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int gl = (c8 * 8 + j) / p.cg;
    const float mean = st[2 * gl], rstd = st[2 * gl + 1];
    const float ga = p.gamma[ch0 + j], be = p.beta[ch0 + j];
    sc[j] = rstd * ga;
    sh[j] = be - mean * rstd * ga;
  }
  uint16_t* const tdst = p.Y + ((int64_t)b * p.rows + rbase) * p.C + ch0;
  const uint32_t dstride = step * (uint32_t)p.C;
#pragma unroll
  for (int i = 0; i < RMAX; ++i) {
    if ((uint32_t)i < nv) {
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f[j] = ((j & 1) ? hi16(v[i][j >> 1]) : lo16(v[i][j >> 1])) * sc[j] + sh[j];
        if (p.silu) f[j] = f[j] * __builtin_amdgcn_rcpf(1.f + __expf(-f[j]));      // (v_rcp_f32: 1 ulp; the IEEE division is ~20 instructions per element)
      }
      u32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = pack16(f[2 * j], f[2 * j + 1]);
      if constexpr (STREAM) st_stream(tdst + (uint32_t)i * dstride, o); else *reinterpret_cast<u32x4_t*>(tdst + (uint32_t)i * dstride) = o;
    }
    __builtin_amdgcn_sched_barrier(0);          // row by row (registers: the data alone are 4 RMAX)
  }
}

// slab / thread plan of gn_fused_kernel; false: the instance does not fit (three-launch path)
struct GNFPlan { int sg, nx, ny, threads, nslab, iters; };
inline bool gn_fused_plan(int64_t rows, int C, int groups, GNFPlan& out) {
#ifdef A3D_EXP_R5_PATHS
  return false;          // measurement build (tools/microbench.py, A3D_LIB=...): the round-5 three-launch GroupNorm for every instance
#endif
  const int cg = C / groups;
  int best_sg = 0;
  for (int sg = 1; sg <= groups; ++sg) {                 // widest slab (<= 640 bytes per row) that fits the registers of 512 threads
    if (groups % sg != 0 || (sg * cg) % 8 != 0) continue;
    const int nx = sg * cg / 8;
    if (nx > 64 || sg * cg * 2 > 640) break;
    const int ny = 512 / nx;
    if ((rows + ny - 1) / ny <= GNF_RMAX) best_sg = sg;
  }
  if (best_sg == 0) return false;
  out.sg = best_sg; out.nx = best_sg * cg / 8; out.nslab = groups / best_sg;
  out.threads = 512;
  {                                                      // 256 threads when they still hold the slab in <= 21 chunks each
    const int ny = 256 / out.nx;
    if (ny >= 1 && (rows + ny - 1) / ny <= 21 && 8 * out.nx <= 256) out.threads = 256;
  }
  out.ny = out.threads / out.nx;
  out.iters = (int)((rows + out.ny - 1) / out.ny);
  return 8 * out.nx <= out.threads && out.sg <= out.threads;
}

template <bool STREAM>
void gn_fused_launch(const GNParams& p, const GNFPlan& pl, int B, size_t smem, hipStream_t s) {
  const dim3 grid((unsigned)(B * pl.nslab)), block(pl.threads);
  if (pl.iters <= 11) gn_fused_kernel<STREAM, 11><<<grid, block, smem, s>>>(p, pl.sg, pl.nx, pl.ny, pl.nslab);
  else gn_fused_kernel<STREAM, 21><<<grid, block, smem, s>>>(p, pl.sg, pl.nx, pl.ny, pl.nslab);
}

// ---------------- LayerNorm: one wave per row, up to 3 16-byte chunks per lane (C <= 1536)
struct LNParams {
  const uint16_t* X; uint16_t* Y1; uint16_t* Y2; const float* gamma; const float* beta;
  int64_t M; int C; float eps;
  const uint16_t* pe1; int64_t pe1_div, pe1_mod;
  const uint16_t* pe2; int64_t pe2_div, pe2_mod;
};

__global__ __launch_bounds__(256) void layer_norm_kernel(const LNParams p) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t m = (int64_t)blockIdx.x * 4 + wid;
  if (m >= p.M) return;
  const int nch = p.C / 8;
  float f[3][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const u32x4_t v = *reinterpret_cast<const u32x4_t*>(p.X + m * p.C + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { f[i][j] = (j & 1) ? hi16(v[j >> 1]) : lo16(v[j >> 1]); sum += f[i][j]; }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)p.C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = f[i][j] - mean; sq += d * d; }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) sq += __shfl_xor(sq, o);
  const float rstd = rsqrtf(sq / (float)p.C + p.eps);
  const uint16_t* e1 = p.pe1 ? p.pe1 + ((m / p.pe1_div) % p.pe1_mod) * p.C : nullptr;
  const uint16_t* e2 = (p.Y2 && p.pe2) ? p.pe2 + ((m / p.pe2_div) % p.pe2_mod) * p.C : nullptr;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      float y[8];
      const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + c * 8), g1 = *reinterpret_cast<const float4*>(p.gamma + c * 8 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(p.beta + c * 8), b1 = *reinterpret_cast<const float4*>(p.beta + c * 8 + 4);
      const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = (f[i][j] - mean) * rstd * ga[j] + be[j];
      {
        float z[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = y[j];
        if (e1) {
          const u32x4_t ev = *reinterpret_cast<const u32x4_t*>(e1 + c * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) z[j] += (j & 1) ? hi16(ev[j >> 1]) : lo16(ev[j >> 1]);
        }
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack16(z[2 * j], z[2 * j + 1]);
        *reinterpret_cast<u32x4_t*>(p.Y1 + m * p.C + c * 8) = o;
      }
      if (p.Y2) {
        float z[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = y[j];
        if (e2) {
          const u32x4_t ev = *reinterpret_cast<const u32x4_t*>(e2 + c * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) z[j] += (j & 1) ? hi16(ev[j >> 1]) : lo16(ev[j >> 1]);
        }
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack16(z[2 * j], z[2 * j + 1]);
        *reinterpret_cast<u32x4_t*>(p.Y2 + m * p.C + c * 8) = o;
      }
    }
  }
}

// Sub-wave rows for the model's widths C = 320 / 640 / 1280 = 40 * LPR: LPR lanes own one row (five 16-byte chunks per
// lane, chunk = sub + LPR * i => LPR * 16 contiguous bytes per step), a wave normalises 64 / LPR rows at once and loops
// over LN_BATCH row batches with gamma / beta held in registers.  The one-wave-per-row kernel above leaves 24 of 64 lanes
// idle at C = 320 and has a single 640-byte load in flight per wave (3.3 TB/s); here every lane has five loads in flight.
constexpr int LN_BATCH = 4;
template <int LPR>
__global__ __launch_bounds__(256) void layer_norm_rows_kernel(const LNParams p) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int sub = lane % LPR, rsel = lane / LPR;
  float ga[5][8], be[5][8];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int c = sub + LPR * i;
    const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + c * 8), g1 = *reinterpret_cast<const float4*>(p.gamma + c * 8 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(p.beta + c * 8), b1 = *reinterpret_cast<const float4*>(p.beta + c * 8 + 4);
    ga[i][0] = g0.x; ga[i][1] = g0.y; ga[i][2] = g0.z; ga[i][3] = g0.w; ga[i][4] = g1.x; ga[i][5] = g1.y; ga[i][6] = g1.z; ga[i][7] = g1.w;
    be[i][0] = b0.x; be[i][1] = b0.y; be[i][2] = b0.z; be[i][3] = b0.w; be[i][4] = b1.x; be[i][5] = b1.y; be[i][6] = b1.z; be[i][7] = b1.w;
  }
  const float inv_c = 1.f / (float)p.C;
  auto row_of = [&](int bt) { return (((int64_t)blockIdx.x * 4 + wid) * LN_BATCH + bt) * RPW + rsel; };
  auto load_rows = [&](int bt, u32x4_t (&dst)[5]) {
    const int64_t m = row_of(bt);
    const uint16_t* xr = p.X + (m < p.M ? m : p.M - 1) * p.C;
#pragma unroll
    for (int i = 0; i < 5; ++i) dst[i] = ld_stream(xr + (sub + LPR * i) * 8);
  };
  u32x4_t nxt[5];
  load_rows(0, nxt);
#pragma unroll 1
  for (int bt = 0; bt < LN_BATCH; ++bt) {
    const int64_t m = row_of(bt);
    const bool ok = m < p.M;
    const int64_t mm = ok ? m : p.M - 1;
    u32x4_t raw[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) raw[i] = nxt[i];
    if (bt + 1 < LN_BATCH) load_rows(bt + 1, nxt);      // the next batch's rows are in flight under this batch's arithmetic and stores
    float f[5][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) { f[i][j] = (j & 1) ? hi16(raw[i][j >> 1]) : lo16(raw[i][j >> 1]); sum += f[i][j]; }
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum * inv_c;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = f[i][j] - mean; sq += d * d; }
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = rsqrtf(sq * inv_c + p.eps);
    const uint16_t* e1 = p.pe1 ? p.pe1 + ((mm / p.pe1_div) % p.pe1_mod) * p.C : nullptr;
    const uint16_t* e2 = (p.Y2 && p.pe2) ? p.pe2 + ((mm / p.pe2_div) % p.pe2_mod) * p.C : nullptr;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int c = sub + LPR * i;
      float y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = (f[i][j] - mean) * rstd * ga[i][j] + be[i][j];
      {
        float z[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = y[j];
        if (e1) {
          const u32x4_t ev = *reinterpret_cast<const u32x4_t*>(e1 + c * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) z[j] += (j & 1) ? hi16(ev[j >> 1]) : lo16(ev[j >> 1]);
        }
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack16(z[2 * j], z[2 * j + 1]);
        if (ok) st_stream(p.Y1 + m * p.C + c * 8, o);
      }
      if (p.Y2) {
        float z[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = y[j];
        if (e2) {
          const u32x4_t ev = *reinterpret_cast<const u32x4_t*>(e2 + c * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) z[j] += (j & 1) ? hi16(ev[j >> 1]) : lo16(ev[j >> 1]);
        }
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack16(z[2 * j], z[2 * j + 1]);
        if (ok) st_stream(p.Y2 + m * p.C + c * 8, o);
      }
    }
  }
}

template <int LPR>
int launch_ln_rows(hipStream_t s, const LNParams& p) {
  const int64_t rows_per_block = (int64_t)4 * LN_BATCH * (64 / LPR);
  const int64_t nblk = (p.M + rows_per_block - 1) / rows_per_block;
  if (nblk > 0x7fffffffLL) return A3D_EINVAL;
  layer_norm_rows_kernel<LPR><<<dim3((unsigned)nblk), dim3(256), 0, s>>>(p);
  return a3d_launch_status();
}


inline int gn_nchunk(int64_t rows) { return (int)((rows + GN_ROWS_PER_BLOCK - 1) / GN_ROWS_PER_BLOCK); }

}  // namespace

#ifndef A3D_STORAGE_F16
extern "C" int64_t a3d_group_norm_ws_floats(int B, int64_t rows, int groups) {
  return (int64_t)B * gn_nchunk(rows) * groups * 2 + (int64_t)B * groups * 2;
}
#endif

// mode 0: whole GroupNorm; 1: statistics only (raw fp64 sums); 2: apply only (stats = [B][groups][2] mean, rstd)
static int group_norm_launch(int mode, a3d_stream_t stream, const void* X, void* Y, const float* gamma, const float* beta,
                             float* ws, double* sums, const float* stats_in, int B, int64_t rows, int C, int groups, float eps, int silu,
                             const void* Xb = nullptr, int C1 = 0) {
  if (!X || B <= 0 || rows <= 0 || C <= 0 || groups <= 0) return A3D_EINVAL;
  if (Xb && (C1 <= 0 || C1 >= C || C1 % 8 != 0 || (reinterpret_cast<uintptr_t>(Xb) & 15u))) return A3D_EINVAL;
  if (mode != 1 && (!Y || !gamma || !beta)) return A3D_EINVAL;
  if ((mode != 2 && !ws) || (mode == 1 && !sums) || (mode == 2 && !stats_in)) return A3D_EINVAL;
  if (C % 8 != 0 || C % groups != 0 || C / 8 > 1024 || groups > 1024) return A3D_EINVAL;
  // a 16-byte chunk (8 channels) must touch at most 2 groups: channels-per-group 4 (chunk = exactly 2 groups) or >= 7
  // (8 channels starting anywhere span <= 2 groups of >= 7); 5 and 6 can straddle 3 groups and are rejected
  if (const int cg = C / groups; cg < 4 || cg == 5 || cg == 6) return A3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15u) return A3D_EINVAL;
  if (B > 65535) return A3D_EINVAL;
  GNParams p{};
  p.X = (const uint16_t*)X; p.Y = (uint16_t*)Y; p.gamma = gamma; p.beta = beta;
  p.Xb = (const uint16_t*)Xb; p.C1 = C1;
  p.B = B; p.rows = rows; p.C = C; p.groups = groups; p.cg = C / groups; p.nchunk = gn_nchunk(rows);
  p.eps = eps; p.silu = silu;
  p.partial = ws; p.stats = mode == 2 ? const_cast<float*>(stats_in) : ws + (int64_t)B * p.nchunk * groups * 2;
  p.sums = mode == 1 ? sums : nullptr;
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) {
    GNFPlan pl;
    if (gn_fused_plan(rows, C, groups, pl) && (int64_t)B * pl.nslab <= 0x7fffffffLL && rows * (int64_t)C * 2 < (1ll << 31)) {
      const size_t smem = ((size_t)pl.ny * pl.nx * 4 + (size_t)8 * pl.nx * 4 + (size_t)2 * pl.sg) * sizeof(float);
      const bool full_lines = (pl.sg * p.cg * 2) % 128 == 0 && Xb == nullptr;      // (every slab then starts on a line: C % (sg cg) == 0)
      if (full_lines) gn_fused_launch<true>(p, pl, B, smem, s);
      else gn_fused_launch<false>(p, pl, B, smem, s);
      return a3d_launch_status();
    }
  }
  const int c8 = C / 8;
  int ny = 256 / c8; if (ny < 1) ny = 1; if (ny > 32) ny = 32;
  if (c8 * ny < groups) return A3D_EINVAL;
  const dim3 block(c8, ny), grid(p.nchunk, B);
  if (mode != 2) {
    gn_partial_kernel<<<grid, block, (size_t)c8 * ny * 4 * sizeof(float), s>>>(p);
    const int gx = (groups + 31) / 32 * 32;
    int nsub = 1024 / gx; if (nsub > 32) nsub = 32; if (nsub > p.nchunk) nsub = p.nchunk; if (nsub < 1) nsub = 1;
    gn_finalize_kernel<<<dim3(B), dim3(gx, nsub), (size_t)nsub * groups * 2 * sizeof(double), s>>>(p);
  }
  if (mode != 1) gn_apply_kernel<<<grid, block, 0, s>>>(p);
  return a3d_launch_status();
}

extern "C" int A3D_FN(a3d_group_norm)(a3d_stream_t stream, const void* X, void* Y, const float* gamma, const float* beta,
                                   float* ws, int B, int64_t rows, int C, int groups, float eps, int silu) {
  return group_norm_launch(0, stream, X, Y, gamma, beta, ws, nullptr, nullptr, B, rows, C, groups, eps, silu);
}

extern "C" int A3D_FN(a3d_group_norm2)(a3d_stream_t stream, const void* Xa, int Ca, const void* Xb, int Cb, void* Y, const float* gamma,
                                    const float* beta, float* ws, int B, int64_t rows, int groups, float eps, int silu) {
  if (!Xb || Ca <= 0 || Cb <= 0) return A3D_EINVAL;
  return group_norm_launch(0, stream, Xa, Y, gamma, beta, ws, nullptr, nullptr, B, rows, Ca + Cb, groups, eps, silu, Xb, Ca);
}

extern "C" int A3D_FN(a3d_group_norm_sums)(a3d_stream_t stream, const void* X, float* ws, double* sums, int B, int64_t rows, int C, int groups) {
  return group_norm_launch(1, stream, X, nullptr, nullptr, nullptr, ws, sums, nullptr, B, rows, C, groups, 0.f, 0);
}

extern "C" int A3D_FN(a3d_group_norm_apply)(a3d_stream_t stream, const void* X, void* Y, const float* gamma, const float* beta,
                                         const float* stats, int B, int64_t rows, int C, int groups, int silu) {
  return group_norm_launch(2, stream, X, Y, gamma, beta, nullptr, nullptr, stats, B, rows, C, groups, 0.f, silu);
}

extern "C" int A3D_FN(a3d_layer_norm)(a3d_stream_t stream, const void* X, void* Y1, void* Y2, const float* gamma,
                                   const float* beta, int64_t M, int C, float eps,
                                   const void* pe1, int64_t pe1_div, int64_t pe1_mod,
                                   const void* pe2, int64_t pe2_div, int64_t pe2_mod) {
  if (!X || !Y1 || !gamma || !beta || M <= 0 || C <= 0 || C % 8 != 0 || C > 1536) return A3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y1) | reinterpret_cast<uintptr_t>(Y2)) & 15u) return A3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15u) return A3D_EINVAL;
  if ((pe1 && (pe1_div <= 0 || pe1_mod <= 0)) || (pe2 && (pe2_div <= 0 || pe2_mod <= 0))) return A3D_EINVAL;
  LNParams p{};
  p.X = (const uint16_t*)X; p.Y1 = (uint16_t*)Y1; p.Y2 = (uint16_t*)Y2; p.gamma = gamma; p.beta = beta;
  p.M = M; p.C = C; p.eps = eps;
  p.pe1 = (const uint16_t*)pe1; p.pe1_div = pe1 ? pe1_div : 1; p.pe1_mod = pe1 ? pe1_mod : 1;
  p.pe2 = (const uint16_t*)pe2; p.pe2_div = pe2 ? pe2_div : 1; p.pe2_mod = pe2 ? pe2_mod : 1;
  // the model's widths always take the sub-wave-rows kernel (independent of M: a batch slice must reproduce the full batch
  // bit for bit, tests/test_unet_gpu.py::test_full_size_batch_independence); other widths (text tokens, C = 768): one wave per row
  if (C == 320) return launch_ln_rows<8>((hipStream_t)stream, p);
  if (C == 640) return launch_ln_rows<16>((hipStream_t)stream, p);
  if (C == 1280) return launch_ln_rows<32>((hipStream_t)stream, p);
  const int64_t nblk = (M + 3) / 4;
  if (nblk > 0x7fffffffLL) return A3D_EINVAL;
  layer_norm_kernel<<<dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream>>>(p);
  return a3d_launch_status();
}
