// Layout / elementwise / heat-map kernels (HBM-bound byte movers) for the AGenDA path on gfx950.
#include "kernels.h"

// ---------------------------------------------------------------------------------------
// Weight re-layout: torch [N][Cin][taps] fp32 -> [N][tap][Cpad] bf16 (K = tap-major, channel
// minor, zero-padded to Cpad).  geglu_bn>0: permute rows so every group of geglu_bn (= 16) rows holds
// [8 values | 8 gates] (ff.net.0.proj: first N/2 rows are values, last N/2 are gates): a lane of the igemm
// epilogue owns 16 consecutive rows of one pixel = 8 output channels with their gates.
// ---------------------------------------------------------------------------------------
__global__ void convert_weight_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int N, int Cin, int taps,
                                      int Cpad, int geglu_bn) {
  const long long total = (long long)N * taps * Cpad;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const int tap = (int)((i / Cpad) % taps);
    const int n = (int)(i / ((long long)Cpad * taps));
    int src_n = n;
    if (geglu_bn > 0) {
      const int half = geglu_bn / 2, j = n / geglu_bn, wi = n % geglu_bn;
      src_n = (wi < half) ? j * half + wi : N / 2 + j * half + (wi - half);
    }
    float v = 0.f;
    if (c < Cin) v = w[((long long)src_n * Cin + c) * taps + tap];
    out[i] = f2bf(v);
  }
}
int launch_convert_weight(const float* w, bf16_t* out, int N, int Cin, int taps, int Cpad, int geglu_bn, hipStream_t st) {
  const long long total = (long long)N * taps * Cpad;
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(convert_weight_kernel, dim3(grid), dim3(256), 0, st, w, out, N, Cin, taps, Cpad, geglu_bn);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = f2bf(x[i]);
}
__global__ void bf16_to_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = bf2f(x[i]);
}
static inline int grid_for(long long n) { long long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }
int launch_f32_to_bf16(const float* x, bf16_t* y, long long n, hipStream_t st) {
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, y, n);
  HIP_CHECK_RET(hipGetLastError()); return 0;
}
int launch_bf16_to_f32(const bf16_t* x, float* y, long long n, hipStream_t st) {
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, y, n);
  HIP_CHECK_RET(hipGetLastError()); return 0;
}

// latents NCHW fp32 [B][C][HW] -> NHWC bf16 [dup*B][HW][Cpad] (zero channel pad), * scale
__global__ void prep_latents_kernel(const float* __restrict__ lat, bf16_t* __restrict__ out, int B, int C, int HW, int Cpad,
                                    int dup, float scale) {
  const long long total = (long long)dup * B * HW * Cpad;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const long long px = i / Cpad;
    const int p = (int)(px % HW);
    const int b = (int)((px / HW) % B);
    out[i] = f2bf(c < C ? lat[((long long)b * C + c) * HW + p] * scale : 0.f);
  }
}
int launch_prep_latents(const float* lat, bf16_t* out, int B, int C, int HW, int Cpad, int dup, float scale, hipStream_t st) {
  hipLaunchKernelGGL(prep_latents_kernel, dim3(grid_for((long long)dup * B * HW * Cpad)), dim3(256), 0, st, lat, out, B, C, HW, Cpad, dup, scale);
  HIP_CHECK_RET(hipGetLastError()); return 0;
}

// Upsample2D.conv as four 2x2 phase convs on the un-upsampled map (IgemmP::ups4): output pixel (2 i + a, 2 j + b) of conv3x3(nearest2x(x)) sees source rows {i - 1 + a, i + a}: for a = 0
// tap kh = 0 on the upper row and kh = 1, 2 on the lower one, for a = 1 kh = 0, 1 on the upper and kh = 2 on the lower (columns alike).  [Cout][9][Cin] -> [4 Cout][4][Cin]: row (2 a + b) Cout + co,
// tap 2 r + s = the fp32 sum of the coinciding taps, rounded to bf16 once.
__global__ void upsample_phase_weight_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ out, int Cout, int Cin) {
  const long long total = (long long)4 * Cout * 4 * Cin;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cin);
    const int lt = (int)((i / Cin) % 4);
    const int n = (int)(i / ((long long)4 * Cin));
    const int ph = n / Cout, co = n - ph * Cout, a = ph >> 1, b = ph & 1, r = lt >> 1, sx = lt & 1;
    const int kh0 = a == 0 ? (r == 0 ? 0 : 1) : (r == 0 ? 0 : 2), kh1 = a == 0 ? (r == 0 ? 0 : 2) : (r == 0 ? 1 : 2);
    const int kw0 = b == 0 ? (sx == 0 ? 0 : 1) : (sx == 0 ? 0 : 2), kw1 = b == 0 ? (sx == 0 ? 0 : 2) : (sx == 0 ? 1 : 2);
    float acc = 0.f;
    for (int kh = kh0; kh <= kh1; ++kh)
      for (int kw = kw0; kw <= kw1; ++kw) acc += bf2f(w[((long long)co * 9 + kh * 3 + kw) * Cin + c]);
    out[i] = f2bf(acc);
  }
}
int launch_upsample_phase_weight(const bf16_t* w, bf16_t* out, int Cout, int Cin, hipStream_t st) {
  hipLaunchKernelGGL(upsample_phase_weight_kernel, dim3(grid_for((long long)16 * Cout * Cin)), dim3(256), 0, st, w, out, Cout, Cin);
  HIP_CHECK_RET(hipGetLastError()); return 0;
}

// diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin], fp32
__global__ void timestep_embed_kernel(float t, float* __restrict__ out, int dim) {
  const int half = dim / 2;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= half) return;
  const float freq = expf(-logf(10000.0f) * (float)k / (float)half);
  const float a = t * freq;
  out[k] = cosf(a);
  out[half + k] = sinf(a);
}
int launch_timestep_embed(float t, float* out, int dim, hipStream_t st) {
  hipLaunchKernelGGL(timestep_embed_kernel, dim3((dim / 2 + 255) / 256), dim3(256), 0, st, t, out, dim);
  HIP_CHECK_RET(hipGetLastError()); return 0;
}

// Small-M linear (time embedding MLP, per-resnet time_emb_proj): out[m][n] = act(W[n].act_in(x[m]) + b[n]),  M <= 8.
// The block stages act_in(x) in LDS once (the activation is applied M*K times, not M*K*N times); one wave per output column
// (several columns in turn when N is large), W streamed once with 16-B loads, fp32 accumulate in a fixed order.
// MMAX rows at most: 8 (one forward's time embedding) or 25 (agd_denoise embeds all of its timesteps up front: 50 steps = two sweeps of the 50 MB stacked time_emb_proj
// matrix instead of seven, VERDICT r4 item 8 -- the fp32 rows stay fp32, 25 x 1280 floats are 125 KB of LDS)
template <int MMAX>
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ x, const bf16_t* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           int M, int N, int K, int silu_in, int silu_out, int cpw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xs = (float*)smem;                        // [M][K]
  for (int i = threadIdx.x; i < M * K; i += 256) { const float v = x[i]; xs[i] = silu_in ? silu_f(v) : v; }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int nb = (blockIdx.x * 4 + (threadIdx.x >> 6)) * cpw;
  for (int c = 0; c < cpw; ++c) {
    const int n = nb + c;
    if (n >= N) return;
    float acc[MMAX];
#pragma unroll
    for (int m = 0; m < MMAX; ++m) acc[m] = 0.f;
    for (int k0 = lane * 8; k0 < K; k0 += 512) {
      const s16x8 wv = *(const s16x8*)(W + (long long)n * K + k0);
      float wf[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) wf[e] = bf2f((bf16_t)wv[e]);
#pragma unroll
      for (int m = 0; m < MMAX; ++m) {
        if (m < M) {
          const f32x4 x0 = *(const f32x4*)&xs[m * K + k0], x1 = *(const f32x4*)&xs[m * K + k0 + 4];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[m] += wf[e] * x0[e];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[m] += wf[4 + e] * x1[e];
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MMAX; ++m) {
      float a = acc[m];
      for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o);
      if (lane == 0 && m < M) {
        a += bias ? bias[n] : 0.f;
        out[(long long)m * N + n] = silu_out ? silu_f(a) : a;
      }
    }
  }
}
int launch_small_linear(const float* x, const bf16_t* W, const float* bias, float* out, int M, int N, int K, int silu_in,
                        int silu_out, hipStream_t st) {
  if (M < 1 || M > 25 || (K & 7) || (size_t)M * K * 4 > 160 * 1024) { agd_set_error("small_linear: M=%d K=%d unsupported", M, K); return -1; }
  const int cpw = N > 4096 ? (N + 4095) / 4096 : 1;            // ~1024 blocks at most
  const size_t lds = (size_t)M * K * 4;
  if (M <= 8 && lds <= 65536) {
    hipLaunchKernelGGL(small_linear_kernel<8>, dim3((N + 4 * cpw - 1) / (4 * cpw)), dim3(256), lds, st, x, W, bias, out, M, N, K, silu_in, silu_out, cpw);
  } else {
    static std::atomic<bool> attr[AGD_MAX_DEVICES] = {};
    int dev = 0; HIP_CHECK_RET(hipGetDevice(&dev));
    if (dev < 0 || dev >= AGD_MAX_DEVICES) { agd_set_error("small_linear: device ordinal %d out of range", dev); return -1; }
    if (!attr[dev]) { HIP_CHECK_RET(hipFuncSetAttribute((const void*)small_linear_kernel<25>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr[dev] = true; }
    hipLaunchKernelGGL(small_linear_kernel<25>, dim3((N + 4 * cpw - 1) / (4 * cpw)), dim3(256), lds, st, x, W, bias, out, M, N, K, silu_in, silu_out, cpw);
  }
  HIP_CHECK_RET(hipGetLastError()); return 0;
}

// CFG combine + DDIM step (eta=0), in place on fp32 NCHW latents.
// eps NHWC fp32 [2B][HW][ldc]: rows [0,B) unconditional, [B,2B) conditional.
__global__ void cfg_ddim_kernel(const float* __restrict__ eps, int ldc, float* __restrict__ lat, int B, int C, int HW,
                                float guidance, float sa_t, float s1a_t, float sa_p, float s1a_p, int vpred) {
  const long long total = (long long)B * C * HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const int b = (int)(i / ((long long)HW * C));
    const float eu = eps[((long long)b * HW + p) * ldc + c];
    const float ec = eps[((long long)(b + B) * HW + p) * ldc + c];
    const float mo = eu + guidance * (ec - eu);
    const float x = lat[i];
    float x0, e;
    if (!vpred) { x0 = (x - s1a_t * mo) / sa_t; e = mo; }
    else { x0 = sa_t * x - s1a_t * mo; e = sa_t * mo + s1a_t * x; }
    lat[i] = sa_p * x0 + s1a_p * e;
  }
}
int launch_cfg_ddim(const float* eps, int ldc, float* lat, int B, int C, int HW, float guidance, float a_t, float a_p,
                    int vpred, hipStream_t st) {
  hipLaunchKernelGGL(cfg_ddim_kernel, dim3(grid_for((long long)B * C * HW)), dim3(256), 0, st, eps, ldc, lat, B, C, HW, guidance,
                     sqrtf(a_t), sqrtf(1.f - a_t), sqrtf(a_p), sqrtf(1.f - a_p), vpred);
  HIP_CHECK_RET(hipGetLastError()); return 0;
}

// CFG combine + one linear-multistep (PNDM / PLMS) update on fp32 NCHW latents:
//   e0 = eps_u + g (eps_c - eps_u);  e = w0 e0 + w1 h1 + w2 h2 + w3 h3;  lat = a * src + b * e;  store (if given) = e0
// h1..h3: CFG-combined eps of earlier model evaluations (NCHW); src: the sample the step starts from (lat itself, or the
// sample saved at the first evaluation for PLMS's second call).
__global__ void cfg_plms_kernel(const float* __restrict__ eps, int ldc, float* __restrict__ lat, const float* __restrict__ src,
                                const float* __restrict__ h1, const float* __restrict__ h2, const float* __restrict__ h3,
                                float* __restrict__ store, int B, int C, int HW, float guidance, float w0, float w1, float w2, float w3,
                                float a, float b) {
  const long long total = (long long)B * C * HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const int bb = (int)(i / ((long long)HW * C));
    const float eu = eps[((long long)bb * HW + p) * ldc + c];
    const float ec = eps[((long long)(bb + B) * HW + p) * ldc + c];
    const float e0 = eu + guidance * (ec - eu);
    float e = w0 * e0;
    if (h1) e += w1 * h1[i];
    if (h2) e += w2 * h2[i];
    if (h3) e += w3 * h3[i];
    const float x = src[i];
    if (store) store[i] = e0;
    lat[i] = a * x + b * e;
  }
}
int launch_cfg_plms(const float* eps, int ldc, float* lat, const float* src, const float* h1, const float* h2, const float* h3, float* store,
                    int B, int C, int HW, float guidance, const float* w, float a, float b, hipStream_t st) {
  hipLaunchKernelGGL(cfg_plms_kernel, dim3(grid_for((long long)B * C * HW)), dim3(256), 0, st, eps, ldc, lat, src, h1, h2, h3, store, B, C, HW,
                     guidance, w[0], w[1], w[2], w[3], a, b);
  HIP_CHECK_RET(hipGetLastError()); return 0;
}

// diffusers post-process: (x/2+0.5).clamp(0,1) -> round-half-even(255 x) -> uint8, NHWC
__global__ void image_u8_kernel(const float* __restrict__ x, int ldc, unsigned char* __restrict__ out, long long npix, int C) {
  const long long total = npix * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long px = i / C; const int c = (int)(i % C);
    float v = x[px * ldc + c] * 0.5f + 0.5f;
    v = fminf(fmaxf(v, 0.f), 1.f);
    out[i] = (unsigned char)rintf(v * 255.0f);
  }
}
int launch_image_u8(const float* x, int ldc, unsigned char* out, long long npix, int C, hipStream_t st) {
  hipLaunchKernelGGL(image_u8_kernel, dim3(grid_for(npix * C)), dim3(256), 0, st, x, ldc, out, npix, C);
  HIP_CHECK_RET(hipGetLastError()); return 0;
}

__global__ void nchw_from_nhwc_kernel(const float* __restrict__ x, int ldc, float* __restrict__ out, int B, int C, int HW) {
  const long long total = (long long)B * C * HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW); const int c = (int)((i / HW) % C); const int b = (int)(i / ((long long)HW * C));
    out[i] = x[((long long)b * HW + p) * ldc + c];
  }
}
int launch_nchw_from_nhwc_f32(const float* x, int ldc, float* out, int B, int C, int HW, hipStream_t st) {
  hipLaunchKernelGGL(nchw_from_nhwc_kernel, dim3(grid_for((long long)B * C * HW)), dim3(256), 0, st, x, ldc, out, B, C, HW);
  HIP_CHECK_RET(hipGetLastError()); return 0;
}

// Row softmax fp32 -> bf16 (VAE mid-block single-head attention, N x N scores)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, bf16_t* __restrict__ p, int cols) {
  __shared__ float red[8];
  const long long row = blockIdx.x;
  const float* sr = s + row * cols;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < cols; i += 256) mx = fmaxf(mx, sr[i]);
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int i = threadIdx.x; i < cols; i += 256) sum += __expf(sr[i] - mx);
  for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
  if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = sum;
  __syncthreads();
  sum = red[4] + red[5] + red[6] + red[7];
  const float inv = 1.f / sum;
  for (int i = threadIdx.x; i < cols; i += 256) p[row * cols + i] = f2bf(__expf(sr[i] - mx) * inv);
}
int launch_softmax_rows(const float* s, bf16_t* p, int rows, int cols, hipStream_t st) {
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, st, s, p, cols);
  HIP_CHECK_RET(hipGetLastError()); return 0;
}

// ---------------------------------------------------------------------------------------
// Heat-map aggregation.  PyTorch upsample_bicubic2d semantics (align_corners=False, A=-0.75):
// src = scale*(dst+0.5)-0.5, taps clamped to the border.
// ---------------------------------------------------------------------------------------
AGD_DEV float cc1(float x) { const float A = -0.75f; return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
AGD_DEV float cc2(float x) { const float A = -0.75f; return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
AGD_DEV void cubic_coeffs(float t, float* c) { c[0] = cc2(t + 1.f); c[1] = cc1(t); c[2] = cc1(1.f - t); c[3] = cc2(2.f - t); }

AGD_DEV float bicubic_at(const float* __restrict__ m, int side, int S, int oy, int ox) {
  if (side == S) return m[oy * side + ox];
  const float scale = (float)side / (float)S;
  const float ry = scale * (oy + 0.5f) - 0.5f, rx = scale * (ox + 0.5f) - 0.5f;
  const float fy = floorf(ry), fx = floorf(rx);
  const int iy = (int)fy, ix = (int)fx;
  float cy[4], cx[4];
  cubic_coeffs(ry - fy, cy);
  cubic_coeffs(rx - fx, cx);
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int yy = min(max(iy - 1 + i, 0), side - 1);
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int xx = min(max(ix - 1 + j, 0), side - 1);
      r += m[yy * side + xx] * cx[j];
    }
    acc += r * cy[i];
  }
  return acc;
}

struct HeatLayers { HeatLayer l[24]; int n; };

// daam compute_global_heat_map: mean over every (layer, head) accumulator of clamp(bicubic(acc), 0)
__global__ void daam_global_kernel(HeatLayers L, int total_maps, int T, int S, int img, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * S * S) return;
  const int ox = i % S, oy = (i / S) % S, t = i / (S * S);
  float sum = 0.f;
  for (int li = 0; li < L.n; ++li) {
    const HeatLayer& hl = L.l[li];
    for (int h = 0; h < hl.heads; ++h) {
      const float* m = hl.acc + img * hl.img_stride + h * hl.head_stride + (long long)t * hl.side * hl.side;
      sum += fmaxf(bicubic_at(m, hl.side, S, oy, ox), 0.f);
    }
  }
  out[i] = sum / (float)total_maps;
}
int launch_daam_global(const HeatLayer* layers, int n_layers, int total_maps, int T, int S, int img, float* out, hipStream_t st) {
  if (n_layers > 24) { agd_set_error("daam_global: too many layers"); return -1; }
  HeatLayers L; L.n = n_layers;
  for (int i = 0; i < n_layers; ++i) L.l[i] = layers[i];
  hipLaunchKernelGGL(daam_global_kernel, dim3((T * S * S + 255) / 256), dim3(256), 0, st, L, total_maps, T, S, img, out);
  HIP_CHECK_RET(hipGetLastError()); return 0;
}

// hook.py compute_global_heat_map, streamed: sum[b][t] += clamp(bicubic(map[b][t]), 0) per recorded call
__global__ void hook_accum_kernel(const float* __restrict__ map, int BT, int side, int S, float* __restrict__ sum) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)BT * S * S) return;
  const int ox = (int)(i % S), oy = (int)((i / S) % S);
  const long long bt = i / ((long long)S * S);
  sum[i] += fmaxf(bicubic_at(map + bt * side * side, side, S, oy, ox), 0.f);
}
int launch_hook_accum(const float* map, int B, int T, int side, int S, float* sum, hipStream_t st) {
  const long long n = (long long)B * T * S * S;
  hipLaunchKernelGGL(hook_accum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, map, B * T, side, S, sum);
  HIP_CHECK_RET(hipGetLastError()); return 0;
}

__global__ void scale_kernel(float* x, long long n, float s) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x[i] *= s;
}
int launch_scale(float* x, long long n, float s, hipStream_t st) {
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, n, s);
  HIP_CHECK_RET(hipGetLastError()); return 0;
}

// ---------------------------------------------------------------------------------------
// Export path on device, bit-exact with the reference's host code:
//   data_generation.py:82-84  (x - min) / (max - min + 1e-8) * 255 -> astype(uint8) (truncation)
//   data_generation.py:60,85  PIL Image.resize (default BICUBIC): Pillow Resample.c 8-bit path --
//       22-bit fixed-point coefficients (computed on the host in double, like Pillow), horizontal
//       pass then vertical pass, each rounded through clip8
//   postprocess_heatmap.py:44-48  stack [obj, fg, 255 - bg]
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void heatmap_u8_kernel(const float* __restrict__ hm, int npix, unsigned char* __restrict__ out) {
  __shared__ float smn[4], smx[4];
  const float* x = hm + (long long)blockIdx.x * npix;
  float mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < npix; i += 256) { const float v = x[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
  for (int o = 32; o >= 1; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
  mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
  const float den = (mx - mn) + 1e-8f;                       // float32 arithmetic, as numpy does on a float32 array
  for (int i = threadIdx.x; i < npix; i += 256) {
    const float v = __fmul_rn(__fdiv_rn(__fsub_rn(x[i], mn), den), 255.0f);   // no contraction: match numpy op by op
    out[(long long)blockIdx.x * npix + i] = (unsigned char)(int)v;            // truncation toward zero
  }
}
int launch_heatmap_u8(const float* hm, int n, int npix, unsigned char* out, hipStream_t st) {
  if (n <= 0 || npix <= 0) return 0;                           // images-only runs request no word maps: nothing to launch
  hipLaunchKernelGGL(heatmap_u8_kernel, dim3(n), dim3(256), 0, st, hm, npix, out);
  HIP_CHECK_RET(hipGetLastError()); return 0;
}

// one resampling pass along `axis_len` (stride `astride` elements between taps); `inner` = elements per tap group
__global__ void pil_resample_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                    const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                    long long n_outer, int in_len, int out_len, int inner) {
  // layout: [outer][len][inner]
  const long long total = n_outer * out_len * inner;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % inner);
    const int o = (int)((i / inner) % out_len);
    const long long ou = i / ((long long)inner * out_len);
    const int xmin = bounds[2 * o], cnt = bounds[2 * o + 1];
    const unsigned char* src = in + (ou * in_len + xmin) * inner + c;
    const int* k = kk + (long long)o * ksize;
    int ss = 1 << 21;                                        // 1 << (PRECISION_BITS - 1)
    for (int t = 0; t < cnt; ++t) ss += (int)src[(long long)t * inner] * k[t];
    ss >>= 22;
    out[i] = (unsigned char)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
  }
}
int launch_pil_resample(const unsigned char* in, unsigned char* out, const int* bounds, const int* kk, int ksize,
                        long long n_outer, int in_len, int out_len, int inner, hipStream_t st) {
  const long long total = n_outer * out_len * inner;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(pil_resample_kernel, dim3(grid_for(total)), dim3(256), 0, st, in, out, bounds, kk, ksize, n_outer, in_len, out_len, inner);
  HIP_CHECK_RET(hipGetLastError()); return 0;
}

__global__ void stack_heatmaps_kernel(const unsigned char* __restrict__ obj, const unsigned char* __restrict__ fg,
                                      const unsigned char* __restrict__ bg, long long npix, unsigned char* __restrict__ rgb,
                                      unsigned char* __restrict__ inv) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
    const unsigned char ib = (unsigned char)(255 - bg[i]);
    rgb[3 * i] = obj[i]; rgb[3 * i + 1] = fg[i]; rgb[3 * i + 2] = ib;
    if (inv) inv[i] = ib;
  }
}
int launch_stack_heatmaps(const unsigned char* obj, const unsigned char* fg, const unsigned char* bg, long long npix,
                          unsigned char* rgb, unsigned char* inv, hipStream_t st) {
  if (npix <= 0) return 0;
  hipLaunchKernelGGL(stack_heatmaps_kernel, dim3(grid_for(npix)), dim3(256), 0, st, obj, fg, bg, npix, rgb, inv);
  HIP_CHECK_RET(hipGetLastError()); return 0;
}

// CLIP embeddings: out[b*T+t] = token_embedding[ids[b][t]] + position_embedding[t]   (bf16 tables, fp32 add)
__global__ void embed_gather_kernel(const int* __restrict__ ids, const bf16_t* __restrict__ tok, const bf16_t* __restrict__ pos,
                                    bf16_t* __restrict__ out, int BT, int T, int H, int vocab_cap) {
  const int nv = H >> 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)BT * nv; i += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(i / nv), v = (int)(i % nv);
    int id = ids[row]; id = id < 0 ? 0 : (id >= vocab_cap ? vocab_cap - 1 : id);
    const s16x8 a = *(const s16x8*)(tok + (long long)id * H + v * 8);
    const s16x8 b = *(const s16x8*)(pos + (long long)(row % T) * H + v * 8);
    u32x4 pk;
#pragma unroll
    for (int e = 0; e < 4; ++e) pk[e] = pack_bf2(bf2f((bf16_t)a[2 * e]) + bf2f((bf16_t)b[2 * e]), bf2f((bf16_t)a[2 * e + 1]) + bf2f((bf16_t)b[2 * e + 1]));
    *(u32x4*)(out + (long long)row * H + v * 8) = pk;
  }
}
int launch_embed_gather(const int* ids, const bf16_t* tok, const bf16_t* pos, bf16_t* out, int B, int T, int H, int vocab_cap, hipStream_t st) {
  hipLaunchKernelGGL(embed_gather_kernel, dim3(grid_for((long long)B * T * (H / 8))), dim3(256), 0, st, ids, tok, pos, out, B * T, T, H, vocab_cap);
  HIP_CHECK_RET(hipGetLastError()); return 0;
}

// ---------------------------------------------------------------------------------------
// LayerNorm folded into the GEMM it feeds (model.hip transformer()): W'[n][k] = bf16(W[n][k] gamma[k]),
// colsum[r] = sum_k W'[n][k], bias'[r] = bias[r] + sum_k beta[k] W[n][k]   (r = the row's ORIGINAL index: GEGLU
// projections are stored in [8 values | 8 gates] row groups, the epilogue indexes bias / colsum in original order)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_fold_weight_kernel(const bf16_t* __restrict__ W, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ bias, int N, int K,
                                                            int geglu_bn, bf16_t* __restrict__ Wf, float* __restrict__ cs, float* __restrict__ bf) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float a = 0.f, b = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float w = bf2f(W[(long long)n * K + k]);
    const bf16_t wp = f2bf(w * gamma[k]);
    Wf[(long long)n * K + k] = wp;
    a += bf2f(wp); b += beta[k] * w;
  }
  for (int o = 32; o >= 1; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
  if (lane == 0) {
    int r = n;
    if (geglu_bn > 0) { const int half = geglu_bn / 2, j = n / geglu_bn, wi = n % geglu_bn; r = (wi < half) ? j * half + wi : N / 2 + j * half + (wi - half); }
    cs[r] = a; bf[r] = (bias ? bias[r] : 0.f) + b;
  }
}
int launch_ln_fold_weight(const bf16_t* W, const float* gamma, const float* beta, const float* bias, int N, int K, int geglu_bn, bf16_t* Wf,
                          float* cs, float* bf, hipStream_t st) {
  hipLaunchKernelGGL(ln_fold_weight_kernel, dim3((N + 3) / 4), dim3(256), 0, st, W, gamma, beta, bias, N, K, geglu_bn, Wf, cs, bf);
  HIP_CHECK_RET(hipGetLastError()); return 0;
}
