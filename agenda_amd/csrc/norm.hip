// GroupNorm(+SiLU) and LayerNorm for NHWC bf16 activations (gfx950).
// Replaces torch group_norm/silu/layer_norm reached through diffusers ResnetBlock2D /
// Transformer2DModel / BasicTransformerBlock.
// HBM-bound: 16-B vector loads, fp32 statistics, two-level deterministic reduction
// (per-block partials -> fp64 finalize), affine folded into per-(image,channel) scale/shift.
#include "kernels.h"
#include <cstdlib>
typedef __attribute__((ext_vector_type(2))) float f32x2;

// ws layout (floats): [0, B*nchunk*groups*2) block partials ; then B*C scale ; then B*C shift
// Thread t owns channel vector (t % nvec) and pixel-row (t / nvec) of its block's pixel chunk, so no
// index division happens inside the streaming loops.
struct GnGeom { int nvec, PR, ppb, nchunk; };
static GnGeom gn_geom(int B, int C, int HW) {
  GnGeom g; g.nvec = C / 8; g.PR = 512 / g.nvec;
  constexpr int nblk = 256;   // one 512-thread block per CU measured best (kbench gn: 0.96 -> 0.88 ms per forward vs 512; 128 / 64: +10 / +21 %)
  const long long target = ((long long)HW * B + nblk - 1) / nblk;    // pixels per block for ~nblk blocks
  int it = (int)((target + g.PR - 1) / g.PR); if (it < 1) it = 1; if (it > 16) it = 16;
  g.ppb = it * g.PR; g.nchunk = (HW + g.ppb - 1) / g.ppb;
  return g;
}

__global__ __launch_bounds__(512) void gn_stats_kernel(const bf16_t* __restrict__ x0, const bf16_t* __restrict__ x1,
                                                       int C0, int C1, int HW, int groups, int ppb, int nchunk,
                                                       float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int C = C0 + C1, nvec = C >> 3;
  const int PR = 512 / nvec;
  const int tid = threadIdx.x;
  const int vcol = tid % nvec, prow = tid / nvec;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int p0 = chunk * ppb, p1 = min(HW, p0 + ppb);
  float s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
  if (prow < PR) {
    const int ch = vcol * 8;
    const bf16_t* base; int Cs, cc;
    if (ch < C0) { base = x0; Cs = C0; cc = ch; } else { base = x1; Cs = C1; cc = ch - C0; }
    base += (long long)b * HW * Cs + cc;
    int px = p0 + prow;
    for (; px + 3 * PR < p1; px += 4 * PR) {          // 4 independent loads in flight
      s16x8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *(const s16x8*)(base + (long long)(px + u * PR) * Cs);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = bf2f((bf16_t)v[u][e]); s[e] += f; ss[e] += f * f; }
    }
    for (; px < p1; px += PR) {
      const s16x8 v = *(const s16x8*)(base + (long long)px * Cs);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float f = bf2f((bf16_t)v[e]); s[e] += f; ss[e] += f * f; }
    }
  }
  // reduce over pixel-rows through LDS: [PR][C] sums and sumsqs
  float* ls = (float*)smem;
  float* lss = ls + PR * C;
  if (prow < PR) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { ls[prow * C + vcol * 8 + e] = s[e]; lss[prow * C + vcol * 8 + e] = ss[e]; }
  }
  __syncthreads();
  const int cpg = C / groups;
  // 8 threads per group: strided partial sums then a shuffle reduce
  const int g = tid >> 3, l = tid & 7;
  float a = 0.f, q = 0.f;
  if (g < groups) {
    const int n = PR * cpg;
    for (int i = l; i < n; i += 8) { const int r = i / cpg, cch = g * cpg + (i - r * cpg); a += ls[r * C + cch]; q += lss[r * C + cch]; }
  }
  for (int o = 4; o >= 1; o >>= 1) { a += __shfl_xor(a, o); q += __shfl_xor(q, o); }
  if (g < groups && l == 0) {
    float* o = part + (((long long)b * nchunk + chunk) * groups + g) * 2;
    o[0] = a; o[1] = q;
  }
}

// apply: every block first reduces the per-chunk partials of ITS image (deterministic fixed order,
// fp64) into group mean/rstd in LDS -- the "finalize" step costs ~nchunk*groups*8 B of L2 reads per
// block and saves a dependent kernel launch -- then streams y = silu?(x*scale + shift).
__global__ __launch_bounds__(512) void gn_apply_kernel(const bf16_t* __restrict__ x0, const bf16_t* __restrict__ x1,
                                                       int C0, int C1, int HW, int ppb, int nchunk, int groups, float eps,
                                                       const float* __restrict__ part,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int silu, bf16_t* __restrict__ y) {
  __shared__ float smean[64], srstd[64];
  const int C = C0 + C1, nvec = C >> 3;
  const int PR = 512 / nvec;
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  {
    const int g = tid >> 3, l = tid & 7;
    double a = 0.0, q = 0.0;
    if (g < groups) {
      for (int ck = l; ck < nchunk; ck += 8) {
        const float* pp = part + (((long long)b * nchunk + ck) * groups + g) * 2;
        a += (double)pp[0]; q += (double)pp[1];
      }
    }
    for (int o = 4; o >= 1; o >>= 1) { a += __shfl_xor(a, o); q += __shfl_xor(q, o); }
    if (g < groups && l == 0) {
      const double n = (double)HW * (C / groups);
      const double mean = a / n;
      double var = q / n - mean * mean;
      if (var < 0) var = 0;
      smean[g] = (float)mean;
      srstd[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
  const int vcol = tid % nvec, prow = tid / nvec;
  if (prow >= PR) return;
  const int p0 = blockIdx.x * ppb, p1 = min(HW, p0 + ppb);
  const int ch = vcol * 8;
  const bf16_t* base; int Cs, cc;
  if (ch < C0) { base = x0; Cs = C0; cc = ch; } else { base = x1; Cs = C1; cc = ch - C0; }
  base += (long long)b * HW * Cs + cc;
  bf16_t* yb = y + (long long)b * HW * C + ch;
  const int cpg = C / groups;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int gg = (ch + e) / cpg;
    sc[e] = srstd[gg] * gamma[ch + e];
    sh[e] = beta[ch + e] - smean[gg] * sc[e];
  }
  auto one = [&](const s16x8 v, int px) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float r = fmaf(bf2f((bf16_t)v[e]), sc[e], sh[e]); o[e] = silu ? silu_f(r) : r; }
    u32x4 pk;
    pk[0] = pack_bf2(o[0], o[1]); pk[1] = pack_bf2(o[2], o[3]); pk[2] = pack_bf2(o[4], o[5]); pk[3] = pack_bf2(o[6], o[7]);
    *(u32x4*)(yb + (long long)px * C) = pk;
  };
  int px = p0 + prow;
  for (; px + 3 * PR < p1; px += 4 * PR) {
    s16x8 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *(const s16x8*)(base + (long long)(px + u * PR) * Cs);
#pragma unroll
    for (int u = 0; u < 4; ++u) one(v[u], px + u * PR);
  }
  for (; px < p1; px += PR) one(*(const s16x8*)(base + (long long)px * Cs), px);
}

// GroupNorm whose statistics pass is already done: the igemm launches that produced x0 / x1 left per-(M tile, channel)
// partial sums (igemm_epilogue.h, colstat_out).  grid (pixel chunks, images, 4 channel slices); the 8 waves of a block
// first reduce the partials of the slice's groups (one wave per group, fixed order, fp64), then stream
// y = silu?(x * scale + shift) over the slice's channels.  One kernel, one read + one write of the activation.
#define GN_SLICES 4
__global__ __launch_bounds__(512) void gn_apply_part_kernel(const bf16_t* __restrict__ x0, const bf16_t* __restrict__ x1,
                                                            int C0, int C1, int HW, int ppb, int groups, float eps,
                                                            const float* __restrict__ part0, const float* __restrict__ part1, int bm0, int bm1,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            int silu, bf16_t* __restrict__ y) {
  __shared__ float smean[8], srstd[8];
  const int C = C0 + C1, cpg = C / groups, gps = groups / GN_SLICES, Cs = C / GN_SLICES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, slice = blockIdx.z;
  // The launch is a chain of memory round trips (partial sums -> gamma / beta -> activation -> stores) on a few hundred blocks: request
  // the affine parameters and the first eight activation vectors of this lane BEFORE the statistics are reduced -- none of them depends
  // on the statistics, only the arithmetic does.  Ragged ends are clamped loads and predicated stores (no branch around a load).
  const int nvec = Cs >> 3, PR = 512 / nvec;
  const int vcol = tid % nvec, prow_raw = tid / nvec;
  const bool active = prow_raw < PR;
  const int prow = active ? prow_raw : 0;
  const int p0 = blockIdx.x * ppb, p1 = min(HW, p0 + ppb);
  const int ch = slice * Cs + vcol * 8;
  const bf16_t* base; int Csrc, cc;
  if (ch < C0) { base = x0; Csrc = C0; cc = ch; } else { base = x1; Csrc = C1; cc = ch - C0; }
  base += (long long)b * HW * Csrc + cc;
  bf16_t* yb = y + (long long)b * HW * C + ch;
  const f32x4 g0 = *(const f32x4*)(gamma + ch), g1 = *(const f32x4*)(gamma + ch + 4);
  const f32x4 b0 = *(const f32x4*)(beta + ch), b1 = *(const f32x4*)(beta + ch + 4);
  s16x8 v[8];
  {
    const int px = p0 + prow;
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int q = px + u * PR < p1 ? px + u * PR : p1 - 1; v[u] = *(const s16x8*)(base + (long long)q * Csrc); }
  }
  if (wave < gps) {
    const int g = slice * gps + wave;
    const int nt0 = HW / bm0, nt1 = C1 ? HW / bm1 : 0;
    const int ntm = nt0 > nt1 ? nt0 : nt1;
    double a = 0.0, q = 0.0;
    // eight independent 8-byte loads in flight per lane and round: one at a time, a 64 x 64 map's 320 .. 960 partial sums per group
    // cost the block 5 .. 15 dependent L2 round trips (3 - 10 us) before its first activation byte moved; same summation order
    const int n = cpg * ntm;                           // < 2^16 (launcher): idx / cpg through a float reciprocal is exact there
    const float rcpg = 1.0f / (float)cpg;
    const int nt1c = nt1 > 0 ? nt1 : 1;
    for (int i0 = lane; i0 < n; i0 += 64 * 8) {
      f32x2 pv[8]; bool pok[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {                      // clamped addresses, unconditional loads, masked adds: no branch, no integer division
        const int idx = i0 + 64 * u, idc = idx < n ? idx : n - 1;
        const int tile = (int)(((float)idc + 0.5f) * rcpg), chn = g * cpg + (idc - tile * cpg);
        const bool in0 = chn < C0;
        const int nt = in0 ? nt0 : nt1c, tc = tile < nt ? tile : nt - 1;
        const float* src = in0 ? part0 + (((long long)b * nt0 + tc) * C0 + chn) * 2 : part1 + (((long long)b * nt1c + tc) * C1 + (chn - C0)) * 2;
        pv[u] = *(const f32x2*)src;
        pok[u] = idx < n && tile < (in0 ? nt0 : nt1);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a += pok[u] ? (double)pv[u][0] : 0.0; q += pok[u] ? (double)pv[u][1] : 0.0; }
    }
    for (int o = 32; o >= 1; o >>= 1) { a += __shfl_xor(a, o); q += __shfl_xor(q, o); }
    if (lane == 0) {
      const double nn = (double)HW * cpg;
      const double mean = a / nn;
      double var = q / nn - mean * mean;
      if (var < 0) var = 0;
      smean[wave] = (float)mean; srstd[wave] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
  if (!active) return;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int gl = (ch + e) / cpg - slice * gps;
    const float ga = e < 4 ? g0[e & 3] : g1[e & 3], be = e < 4 ? b0[e & 3] : b1[e & 3];
    sc[e] = srstd[gl] * ga;
    sh[e] = be - smean[gl] * sc[e];
  }
  auto one = [&](const s16x8 vv, int px) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float r = fmaf(bf2f((bf16_t)vv[e]), sc[e], sh[e]); o[e] = silu ? silu_f(r) : r; }
    u32x4 pk;
    pk[0] = pack_bf2(o[0], o[1]); pk[1] = pack_bf2(o[2], o[3]); pk[2] = pack_bf2(o[4], o[5]); pk[3] = pack_bf2(o[6], o[7]);
    *(u32x4*)(yb + (long long)px * C) = pk;
  };
  for (int px = p0 + prow; px < p1; px += 8 * PR) {
    if (px != p0 + prow) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int q = px + u * PR < p1 ? px + u * PR : p1 - 1; v[u] = *(const s16x8*)(base + (long long)q * Csrc); }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) if (px + u * PR < p1) one(v[u], px + u * PR);
  }
}

// true when launch_groupnorm can use the producers' partial sums for this problem
static bool gn_part_ok(const GroupNormP& p) {
  const int C = p.C0 + p.C1;
  if (!p.part0 || p.bm0 <= 0 || (p.C1 && (!p.part1 || p.bm1 <= 0))) return false;
  if (p.groups % GN_SLICES || p.groups / GN_SLICES > 8 || C % (8 * GN_SLICES) || C % p.groups) return false;
  if ((C / GN_SLICES) % (C / p.groups)) return false;                 // a slice holds whole groups
  if ((long long)(C / p.groups) * (p.HW / p.bm0) >= 65536 || (p.C1 && (long long)(C / p.groups) * (p.HW / p.bm1) >= 65536)) return false;   // the kernel's float-reciprocal index split
  if (p.HW % p.bm0 || (p.C1 && p.HW % p.bm1) || (p.C0 % 8)) return false;
  return (C / GN_SLICES) / 8 <= 512;
}

int launch_groupnorm(const GroupNormP& p, hipStream_t st) {
  const int C = p.C0 + p.C1;
  const int nvec = C / 8;
  if (C % 8 || p.C0 % 8 || nvec > 512 || C % p.groups || p.groups > 64) {
    agd_set_error("groupnorm: unsupported C0=%d C1=%d groups=%d", p.C0, p.C1, p.groups); return -1;
  }
  if (gn_part_ok(p)) {
    const int nvs = C / GN_SLICES / 8, PR = 512 / nvs;
    // pixel chunks per channel slice: ~128 for the 64x64 maps (UNet batch 8), ~64 below (tools/kb_gn.py: 256 -> 128: 16.9 -> 15.4 us at 64x64
    // C = 320, 64: 10.6 -> 9.4 at 32x32 C = 640, 8.4 -> 6.9 at 16x16 C = 1280; 32 and 512 are 20-50 % slower): every block pays the
    // statistics finalize once, so fewer, fatter blocks win until the grid stops covering the CUs
    const int tdiv = (long long)p.HW * p.B >= 32768 ? 128 : 64;
    const long long target = ((long long)p.HW * p.B + tdiv - 1) / tdiv;
    int it = (int)((target + PR - 1) / PR); if (it < 1) it = 1; if (it > 64) it = 64;
    const int ppb = it * PR, nchunk = (p.HW + ppb - 1) / ppb;
    hipLaunchKernelGGL(gn_apply_part_kernel, dim3(nchunk, p.B, GN_SLICES), dim3(512), 0, st, p.x0, p.x1, p.C0, p.C1, p.HW, ppb, p.groups, p.eps,
                       p.part0, p.part1, p.bm0, p.bm1 > 0 ? p.bm1 : 1, p.gamma, p.beta, p.silu, p.y);
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  const GnGeom g = gn_geom(p.B, C, p.HW);
  float* part = p.ws;
  const int lds = 2 * g.PR * C * 4;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(g.nchunk, p.B), dim3(512), lds, st, p.x0, p.x1, p.C0, p.C1, p.HW, p.groups, g.ppb, g.nchunk, part);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(g.nchunk, p.B), dim3(512), 0, st, p.x0, p.x1, p.C0, p.C1, p.HW, g.ppb, g.nchunk, p.groups, p.eps,
                     part, p.gamma, p.beta, p.silu, p.y);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// required workspace floats for a groupnorm call (host helper)
long long groupnorm_ws_floats(int B, int C, int HW, int groups) {
  const GnGeom g = gn_geom(B, C, HW);
  return (long long)B * g.nchunk * groups * 2 + 2LL * B * C;
}

// ---------------------------------------------------------------------------------------
// LayerNorm: LPR lanes per row (16 -> 4 rows per wave for C <= 1280, else 64), values held in
// registers, two-pass variance, 16-B loads/stores.
// ---------------------------------------------------------------------------------------
template <int LPR, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                        const float* __restrict__ g, const float* __restrict__ bta,
                                                        int rows, int C, float eps) {
  constexpr int RPB = 256 / LPR;                    // rows per block
  const int l = threadIdx.x % LPR;
  const int row = blockIdx.x * RPB + threadIdx.x / LPR;
  const bool rok = row < rows;
  const int nvec = C >> 3;
  float v[MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int vi = l + k * LPR;
    if (rok && vi < nvec) {
      const s16x8 r = *(const s16x8*)(x + (long long)row * C + vi * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[k][e] = bf2f((bf16_t)r[e]); s += v[k][e]; }
    }
  }
#pragma unroll
  for (int o = LPR / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int vi = l + k * LPR;
    if (rok && vi < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[k][e] - mean; q += d * d; }
    }
  }
#pragma unroll
  for (int o = LPR / 2; o >= 1; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int vi = l + k * LPR;
    if (rok && vi < nvec) {
      float gm[8], bt[8], o8[8];
      *(f32x4*)&gm[0] = *(const f32x4*)(g + vi * 8); *(f32x4*)&gm[4] = *(const f32x4*)(g + vi * 8 + 4);
      *(f32x4*)&bt[0] = *(const f32x4*)(bta + vi * 8); *(f32x4*)&bt[4] = *(const f32x4*)(bta + vi * 8 + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] = (v[k][e] - mean) * rstd * gm[e] + bt[e];
      u32x4 pk;
      pk[0] = pack_bf2(o8[0], o8[1]); pk[1] = pack_bf2(o8[2], o8[3]); pk[2] = pack_bf2(o8[4], o8[5]); pk[3] = pack_bf2(o8[6], o8[7]);
      *(u32x4*)(y + (long long)row * C + vi * 8) = pk;
    }
  }
}

int launch_layernorm(const bf16_t* x, bf16_t* y, const float* g, const float* b, int rows, int C, float eps, hipStream_t st) {
  if (C % 8 || C > 2048) { agd_set_error("layernorm: unsupported C=%d", C); return -1; }
  // few rows (16x16 / 8x8 feature maps): one wave per row so the grid still covers the 256 CUs
  const bool few = rows < 16 * 512;
  if (few || C > 1280) hipLaunchKernelGGL((layernorm_kernel<64, 4>), dim3((rows + 3) / 4), dim3(256), 0, st, x, y, g, b, rows, C, eps);
  else if (C <= 640) hipLaunchKernelGGL((layernorm_kernel<16, 5>), dim3((rows + 15) / 16), dim3(256), 0, st, x, y, g, b, rows, C, eps);
  else hipLaunchKernelGGL((layernorm_kernel<16, 10>), dim3((rows + 15) / 16), dim3(256), 0, st, x, y, g, b, rows, C, eps);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// GroupNorm folded into the projection behind it (Transformer2DModel: norm -> proj_in, no activation in between).
//   proj_in(GN(x))[n] = sum_c W[n][c] (gamma_c rstd_g (x_c - mu_g) + beta_c) + bias[n]
//                     = sum_c Wb[n][c] x_c  +  (bias[n] + sum_c W[n][c] beta_c - sum_c Wb[n][c] mu_g(c)),   Wb = bf16(W gamma rstd)
// per IMAGE (the statistics are per image and group): the GEMM then reads the RAW activation with image i's matrix and adds image i's
// row -- the separate read + write of the activation by gn_apply_part disappears.  The mean term uses the ROUNDED Wb, so what is
// rounded once to bf16 is W gamma rstd and nothing is amplified by |mu| / sigma.  grid (N / 32, B): a block first reduces the
// image's 32 group statistics from the producer's partial sums (fp64, fixed order), then 8 waves x 4 rows each.
// ---------------------------------------------------------------------------------------
#define GNF_ROWS 16
__global__ __launch_bounds__(512) void gn_fold_weight_kernel(const float* __restrict__ part, int bm, int HW, int C, int groups, float eps,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const bf16_t* __restrict__ W, const float* __restrict__ bias, int N,
                                                             bf16_t* __restrict__ Wb, float* __restrict__ rowadd, int frag_ni) {
  extern __shared__ float gsm[];                   // [groups] mean, [groups] rstd, [C] scale, [C] mean per channel, [C] beta
  float* smean = gsm; float* srstd = gsm + groups; float* sa = gsm + 2 * groups; float* smu = sa + C; float* sbeta = smu + C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, cpg = C / groups, nt = HW / bm;
  // every launch of this kernel is a chain of dependent L2 round trips: all loads of a phase are issued before the first is used.
  // The rows' weight vectors first (their latency runs under the statistics phase): wave w owns rows w and w + 8 of the block's 16
  const int nvec = C >> 3;
  constexpr int VPL = 2;                           // vectors per lane and row (C <= 1024)
  s16x8 w8[2][VPL];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int n = blockIdx.x * GNF_ROWS + wave + 8 * r;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int v = lane + 64 * k;
      w8[r][k] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (n < N && v < nvec) w8[r][k] = *(const s16x8*)(W + (long long)n * C + v * 8);
    }
  }
  // statistics: a 16-lane quarter of a wave per group (group = 8 * round + ... below), its cpg * nt partial sums in one batch of loads
  const int qd = lane >> 4, ql = lane & 15;
  const float inv_cpg = 1.0f / (float)cpg;
  for (int g0 = 0; g0 < groups; g0 += 32) {
    const int g = g0 + wave * 4 + qd;
    const int n = cpg * nt;
    double a = 0.0, q = 0.0;
    for (int i0 = ql; i0 < n; i0 += 16 * 24) {
      f32x2 v[24];
#pragma unroll
      for (int u = 0; u < 24; ++u) {
        const int idx = i0 + 16 * u;
        v[u] = f32x2{0.f, 0.f};
        if (g < groups && idx < n) { const int tile = fast_udiv(idx, cpg, inv_cpg), ch = g * cpg + (idx - tile * cpg); v[u] = *(const f32x2*)(part + (((long long)b * nt + tile) * C + ch) * 2); }
      }
#pragma unroll
      for (int u = 0; u < 24; ++u) { a += (double)v[u][0]; q += (double)v[u][1]; }
    }
    for (int o = 8; o >= 1; o >>= 1) { a += __shfl_xor(a, o); q += __shfl_xor(q, o); }
    if (ql == 0 && g < groups) {
      const double cnt = (double)HW * cpg;
      const double mean = a / cnt;
      double var = q / cnt - mean * mean;
      if (var < 0) var = 0;
      smean[g] = (float)mean; srstd[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 512) { const int g = fast_udiv(c, cpg, inv_cpg); sa[c] = srstd[g] * gamma[c]; smu[c] = smean[g]; sbeta[c] = beta[c]; }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int n = blockIdx.x * GNF_ROWS + wave + 8 * r;
    if (n >= N) break;                               // wave-uniform
    bf16_t* wo = Wb + ((long long)b * N + n) * C;
    float accb = 0.f, accm = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int v = lane + 64 * k;
      if (v < nvec) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = v * 8 + e;
          const float w = bf2f((bf16_t)w8[r][k][e]);
          const float wb = bf2f(f2bf(w * sa[c]));
          o[e] = wb;
          accb += w * sbeta[c];
          accm += wb * smu[c];
        }
        u32x4 pk;
        pk[0] = pack_bf2(o[0], o[1]); pk[1] = pack_bf2(o[2], o[3]); pk[2] = pack_bf2(o[4], o[5]); pk[3] = pack_bf2(o[6], o[7]);
        if (frag_ni > 0) {
          // MFMA fragment order (tblock.hip launch_frag_order_w, KC = C): the 16 bytes (row n, k-vector v) are lane (v & 3) * 16 + rho of fragment
          // (column range n / (16 NI), k-step v / 4, tile j), rho = 4 q' + r' with n % (16 NI) = q' * 4 NI + 4 j + r'
          const int wr = frag_ni * 16, nqi = n / wr, cl = n - nqi * wr, qp = cl / (4 * frag_ni), rem = cl - qp * 4 * frag_ni, j = rem >> 2, rp = rem & 3;
          const long long fi = (((long long)nqi * (C >> 5) + (v >> 2)) * frag_ni + j) * 64 + (v & 3) * 16 + 4 * qp + rp;
          *(u32x4*)(Wb + (long long)b * N * C + fi * 8) = pk;
        } else *(u32x4*)(wo + v * 8) = pk;
      }
    }
    for (int o = 32; o >= 1; o >>= 1) { accb += __shfl_xor(accb, o); accm += __shfl_xor(accm, o); }
    if (lane == 0) rowadd[(long long)b * N + n] = (bias ? bias[n] : 0.f) + accb - accm;
  }
}

int launch_gn_fold_weight(const float* part, int bm, int B, int HW, int C, int groups, float eps, const float* gamma, const float* beta,
                          const bf16_t* W, const float* bias, int N, bf16_t* Wb, float* rowadd, hipStream_t st, int frag_ni) {
  if (frag_ni > 0 && (N % (16 * frag_ni) || C % 32)) { agd_set_error("gn_fold_weight: fragment order needs N %% %d == 0 and C %% 32 == 0", 16 * frag_ni); return -1; }
  if (!part || bm < 1 || HW % bm || C % groups || (C & 7) || B < 1 || N < 1) { agd_set_error("gn_fold_weight: bad shape (HW %d bm %d C %d groups %d)", HW, bm, C, groups); return -1; }
  const size_t lds = (size_t)(2 * groups + 3 * C) * sizeof(float);
  if (C > 1024 || lds > 48 * 1024) { agd_set_error("gn_fold_weight: C = %d too wide", C); return -1; }
  hipLaunchKernelGGL(gn_fold_weight_kernel, dim3((N + GNF_ROWS - 1) / GNF_ROWS, B), dim3(512), lds, st, part, bm, HW, C, groups, eps, gamma, beta, W, bias, N, Wb, rowadd, frag_ni);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
