// GroupNorm(+SiLU) and LayerNorm for NHWC bf16 activations (gfx950).
// Replaces torch group_norm/silu/layer_norm reached through diffusers ResnetBlock2D /
// Transformer2DModel / BasicTransformerBlock.
// HBM-bound: 16-B vector loads, fp32 statistics, two-level deterministic reduction
// (per-block partials -> fp64 finalize), affine folded into per-(image,channel) scale/shift.
#include "kernels.h"

// ws layout (floats): [0, B*nchunk*32*2) block partials ; then B*C scale ; then B*C shift
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
    for (int px = p0 + prow; px < p1; px += PR) {
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
  // one thread per group: sum over its channels and pixel-rows (small)
  if (tid < groups) {
    float a = 0.f, q = 0.f;
    for (int r = 0; r < PR; ++r)
      for (int cch = tid * cpg; cch < (tid + 1) * cpg; ++cch) { a += ls[r * C + cch]; q += lss[r * C + cch]; }
    float* o = part + (((long long)b * nchunk + chunk) * groups + tid) * 2;
    o[0] = a; o[1] = q;
  }
}

__global__ void gn_finalize_kernel(const float* __restrict__ part, int nchunk, int groups, int C, int HW, float eps,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ scale, float* __restrict__ shift) {
  // block = one image; thread t: group g = t / 8, lane-in-group l = t % 8 strides over chunks
  __shared__ float smean[64], srstd[64];
  const int b = blockIdx.x, tid = threadIdx.x;
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
  __syncthreads();
  const int cpg = C / groups;
  for (int cch = tid; cch < C; cch += blockDim.x) {
    const int gg = cch / cpg;
    const float sc = srstd[gg] * gamma[cch];
    scale[(long long)b * C + cch] = sc;
    shift[(long long)b * C + cch] = beta[cch] - smean[gg] * sc;
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x0, const bf16_t* __restrict__ x1,
                                                       int C0, int C1, int HW, long long nvec_total,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       int silu, bf16_t* __restrict__ y) {
  const int C = C0 + C1, nvec = C >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec_total; i += (long long)gridDim.x * 256) {
    const long long px = i / nvec;            // global pixel index (b*HW + p)
    const int vc = (int)(i - px * nvec);
    const int ch = vc * 8;
    const int b = (int)(px / HW);
    const bf16_t* src = (ch < C0) ? x0 + px * C0 + ch : x1 + px * C1 + (ch - C0);
    const s16x8 v = *(const s16x8*)src;
    const f32x4 sc0 = *(const f32x4*)(scale + (long long)b * C + ch), sc1 = *(const f32x4*)(scale + (long long)b * C + ch + 4);
    const f32x4 sh0 = *(const f32x4*)(shift + (long long)b * C + ch), sh1 = *(const f32x4*)(shift + (long long)b * C + ch + 4);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = bf2f((bf16_t)v[e]);
      float r = f * (e < 4 ? sc0[e] : sc1[e - 4]) + (e < 4 ? sh0[e] : sh1[e - 4]);
      o[e] = silu ? silu_f(r) : r;
    }
    u32x4 pk;
    pk[0] = pack_bf2(o[0], o[1]); pk[1] = pack_bf2(o[2], o[3]); pk[2] = pack_bf2(o[4], o[5]); pk[3] = pack_bf2(o[6], o[7]);
    *(u32x4*)(y + px * C + ch) = pk;
  }
}

int launch_groupnorm(const GroupNormP& p, hipStream_t st) {
  const int C = p.C0 + p.C1;
  const int nvec = C / 8;
  if (C % 8 || p.C0 % 8 || nvec > 512 || C % p.groups || p.groups > 64) {
    agd_set_error("groupnorm: unsupported C0=%d C1=%d groups=%d", p.C0, p.C1, p.groups); return -1;
  }
  const int PR = 512 / nvec;
  const int ppb = PR * 8;
  const int nchunk = (p.HW + ppb - 1) / ppb;
  float* part = p.ws;
  float* scale = part + (long long)p.B * nchunk * p.groups * 2;
  float* shift = scale + (long long)p.B * C;
  const int lds = 2 * PR * C * 4;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, p.B), dim3(512), lds, st, p.x0, p.x1, p.C0, p.C1, p.HW, p.groups, ppb, nchunk, part);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(p.B), dim3(512), 0, st, part, nchunk, p.groups, C, p.HW, p.eps, p.gamma, p.beta, scale, shift);
  const long long nv = (long long)p.B * p.HW * nvec;
  const int grid = (int)((nv + 255) / 256 < 4096 ? (nv + 255) / 256 : 4096);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(grid), dim3(256), 0, st, p.x0, p.x1, p.C0, p.C1, p.HW, nv, scale, shift, p.silu, p.y);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// required workspace floats for a groupnorm call (host helper)
extern "C" long long agd_groupnorm_ws_floats(int B, int C, int HW, int groups) {
  const int nvec = C / 8; const int PR = 512 / nvec; const int ppb = PR * 8;
  const int nchunk = (HW + ppb - 1) / ppb;
  return (long long)B * nchunk * groups * 2 + 2LL * B * C;
}

// ---------------------------------------------------------------------------------------
// LayerNorm: one wave per row, values held in registers (C <= 2048), two-pass variance.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                        const float* __restrict__ g, const float* __restrict__ bta,
                                                        int rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = C >> 3;
  float v[4][8];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int vi = lane + k * 64;
    if (vi < nvec) {
      const s16x8 r = *(const s16x8*)(x + (long long)row * C + vi * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[k][e] = bf2f((bf16_t)r[e]); s += v[k][e]; }
    }
  }
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int vi = lane + k * 64;
    if (vi < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[k][e] - mean; q += d * d; }
    }
  }
  for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int vi = lane + k * 64;
    if (vi < nvec) {
      float o8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] = (v[k][e] - mean) * rstd * g[vi * 8 + e] + bta[vi * 8 + e];
      u32x4 pk;
      pk[0] = pack_bf2(o8[0], o8[1]); pk[1] = pack_bf2(o8[2], o8[3]); pk[2] = pack_bf2(o8[4], o8[5]); pk[3] = pack_bf2(o8[6], o8[7]);
      *(u32x4*)(y + (long long)row * C + vi * 8) = pk;
    }
  }
}

int launch_layernorm(const bf16_t* x, bf16_t* y, const float* g, const float* b, int rows, int C, float eps, hipStream_t st) {
  if (C % 8 || C > 2048) { agd_set_error("layernorm: unsupported C=%d", C); return -1; }
  hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, y, g, b, rows, C, eps);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
