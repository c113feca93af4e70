// Training-mode side of the processor seam: what the reference's token fine-tuning does with
// hook.py's recorded maps (reference data_generation/finetune_sd_token.py:1040-1069) and the backward pass of one
// cross-attention call (hook.py:91-120) with respect to its two inputs, so that the attention regulariser can reach
// `hidden_states` and `encoder_hidden_states` (the learned token embeddings).
//
//  * hook_headmean_kernel  -- hook.py:55 `maps.mean(dim=1)`: per-head probabilities -> head-mean map, fixed summation
//                             order (the inference recorder used float atomics across heads before: not reproducible)
//  * attn_reg_loss_kernel  -- finetune_sd_token.py:1046-1066: min-max + L1 normalisation of the object / foreground /
//                             background token maps, the two L1 terms, and their analytic gradient w.r.t. the map
//  * attn_bwd_kernel       -- recompute P = softmax(scale Q K^T); dP = dMap / H (+ dO V^T); dS = P o (dP - rowsum(dP o P));
//                             dQ = scale dS K; per-tile partials of dK = scale dS^T Q and dV = P^T dO
//  * kv_grad_reduce_kernel -- ordered sum of the partials -> [B][T][dK | dV] bf16 (the layout of the fused to_k/to_v rows)
//
// These are fp32 VALU kernels: 77 keys, a few launches per training step; clarity and reproducibility over speed.
#include "kernels.h"

// ---------------------------------------------------------------------------------------
// heads [Bp][H][T][N] fp32 -> map [Bp][T][N] = mean over heads (h = 0, 1, ... in order)
// ---------------------------------------------------------------------------------------
__global__ void hook_headmean_kernel(const float* __restrict__ heads, int H, long long TN, long long total, float* __restrict__ map) {
  const float inv = 1.0f / (float)H;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / TN, r = i - b * TN;
    const float* p = heads + b * H * TN + r;
    float s = 0.f;
    for (int h = 0; h < H; ++h) s += p[h * TN];
    map[i] = s * inv;
  }
}
int launch_hook_headmean(const float* heads, int Bp, int H, int T, int N, float* map, hipStream_t st) {
  const long long TN = (long long)T * N, total = TN * Bp;
  if (total <= 0) return 0;
  const long long g = (total + 255) / 256;
  hipLaunchKernelGGL(hook_headmean_kernel, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, st, heads, H, TN, total, map);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// Attention regulariser (finetune_sd_token.py:1046-1066), one block per sample of ONE recorded map [B][T][P]:
//   o^ = (o - min o) / (max o - min o + 1e-8)          (:1050, object token map)
//   r  = (1 - o^) / sum(1 - o^) ;  o~ = o^ / sum(o^)   (:1051-1053)
//   f  = f^ / sum(f^) , b = b^ / sum(b^)               (:1055-1062, foreground / background token maps)
//   bg_loss = coef * mean |r - b| ; fg_loss = coef * mean |o~ - f|   (:1064-1065; coef = reg_weight / #samples with an object)
// loss_out[b] = {bg, fg}; dmap (optional, pre-zeroed by the launcher) receives d(bg + fg)/dmap on the three rows, including
// the paths through min / max (torch spreads those evenly over ties).  idx < 0 on obj => the sample is skipped (:1048).
// ---------------------------------------------------------------------------------------
__device__ float block_reduce(float v, int op, float* red) {     // op 0 sum, 1 min, 2 max ; 256 threads
  for (int o = 32; o >= 1; o >>= 1) {
    const float w = __shfl_xor(v, o);
    v = op == 0 ? v + w : (op == 1 ? fminf(v, w) : fmaxf(v, w));
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < 4; ++i) r = op == 0 ? r + red[i] : (op == 1 ? fminf(r, red[i]) : fmaxf(r, red[i]));
  return r;
}

struct RowStat { float mn, mx, D, S; };   // min, max, max - min + eps, sum of the min-max normalised row

__device__ RowStat row_stat(const float* x, int P, float* red) {
  float mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < P; i += 256) { mn = fminf(mn, x[i]); mx = fmaxf(mx, x[i]); }
  RowStat s; s.mn = block_reduce(mn, 1, red); s.mx = block_reduce(mx, 2, red);
  s.D = (s.mx - s.mn) + 1e-8f;
  float a = 0.f;
  for (int i = threadIdx.x; i < P; i += 256) a += (x[i] - s.mn) / s.D;
  s.S = block_reduce(a, 0, red);
  return s;
}

// adds to `grow` the gradient through x^ = (x - mn) / D given h = dL/dx^ (evaluated by `hfn(i)`), incl. the min / max paths
template <typename HFn>
__device__ void add_minmax_grad(const float* x, int P, const RowStat& s, HFn hfn, float* grow, float* red) {
  float hs = 0.f, hx = 0.f, nmin = 0.f, nmax = 0.f;
  for (int i = threadIdx.x; i < P; i += 256) {
    const float h = hfn(i), xh = (x[i] - s.mn) / s.D;
    hs += h; hx += h * xh;
    nmin += x[i] == s.mn ? 1.f : 0.f; nmax += x[i] == s.mx ? 1.f : 0.f;
  }
  hs = block_reduce(hs, 0, red); hx = block_reduce(hx, 0, red);
  nmin = block_reduce(nmin, 0, red); nmax = block_reduce(nmax, 0, red);
  const float gmin = (hx - hs) / s.D, gmax = -hx / s.D;            // dL/dmin = sum h (x^ - 1) / D ; dL/dmax = -sum h x^ / D
  for (int i = threadIdx.x; i < P; i += 256) {
    float g = hfn(i) / s.D;
    if (x[i] == s.mn) g += gmin / nmin;
    if (x[i] == s.mx) g += gmax / nmax;
    grow[i] += g;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void attn_reg_loss_kernel(const float* __restrict__ map, int T, int P, const int* __restrict__ obj_idx,
                                                            const int* __restrict__ fg_idx, const int* __restrict__ bg_idx, float coef,
                                                            float* __restrict__ loss_out, float* __restrict__ dmap) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const int io = obj_idx[b], ifg = fg_idx[b], ibg = bg_idx[b];
  if (io < 0 || io >= T || ifg < 0 || ifg >= T || ibg < 0 || ibg >= T) {        // no object in this sample (:1048)
    if (threadIdx.x == 0) { loss_out[2 * b] = 0.f; loss_out[2 * b + 1] = 0.f; }
    return;
  }
  const float* o = map + ((long long)b * T + io) * P;
  const float* f = map + ((long long)b * T + ifg) * P;
  const float* g = map + ((long long)b * T + ibg) * P;
  const RowStat so = row_stat(o, P, red), sf = row_stat(f, P, red), sg = row_stat(g, P, red);
  const float U = (float)P - so.S;                                               // sum(1 - o^)
  const float c = coef / (float)P;                                               // torch.mean over the P pixels
  auto oh = [&](int i) { return (o[i] - so.mn) / so.D; };
  auto r_ = [&](int i) { return (1.f - oh(i)) / U; };
  auto ot = [&](int i) { return oh(i) / so.S; };
  auto fn = [&](int i) { return ((f[i] - sf.mn) / sf.D) / sf.S; };
  auto bn = [&](int i) { return ((g[i] - sg.mn) / sg.D) / sg.S; };
  auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
  float lb = 0.f, lf = 0.f, gr_r = 0.f, gr_o = 0.f, gr_f = 0.f, gr_b = 0.f;      // losses ; sum g.n for each normalised map
  for (int i = threadIdx.x; i < P; i += 256) {
    const float db = r_(i) - bn(i), df = ot(i) - fn(i);
    lb += fabsf(db); lf += fabsf(df);
    gr_r += c * sgn(db) * r_(i); gr_b += -c * sgn(db) * bn(i);
    gr_o += c * sgn(df) * ot(i); gr_f += -c * sgn(df) * fn(i);
  }
  lb = block_reduce(lb, 0, red); lf = block_reduce(lf, 0, red);
  gr_r = block_reduce(gr_r, 0, red); gr_b = block_reduce(gr_b, 0, red); gr_o = block_reduce(gr_o, 0, red); gr_f = block_reduce(gr_f, 0, red);
  if (threadIdx.x == 0) { loss_out[2 * b] = c * lb; loss_out[2 * b + 1] = c * lf; }
  if (!dmap) return;
  // n = x^ / S: dL/dx^_q = (g_q - sum_p g_p n_p) / S ; r = (1 - o^) / U: dL/do^_q = -(g_q - sum g r) / U
  auto h_o = [&](int i) { return (c * sgn(ot(i) - fn(i)) - gr_o) / so.S - (c * sgn(r_(i) - bn(i)) - gr_r) / U; };
  auto h_f = [&](int i) { return (-c * sgn(ot(i) - fn(i)) - gr_f) / sf.S; };
  auto h_b = [&](int i) { return (-c * sgn(r_(i) - bn(i)) - gr_b) / sg.S; };
  // rows may coincide (n_object_embedding = 0 makes obj == fg): accumulate one after the other, in this fixed order
  add_minmax_grad(o, P, so, h_o, dmap + ((long long)b * T + io) * P, red);
  add_minmax_grad(f, P, sf, h_f, dmap + ((long long)b * T + ifg) * P, red);
  add_minmax_grad(g, P, sg, h_b, dmap + ((long long)b * T + ibg) * P, red);
}

int launch_attn_reg_loss(const float* map, int B, int T, int P, const int* obj_idx, const int* fg_idx, const int* bg_idx, float coef,
                         float* loss_out, float* dmap, hipStream_t st) {
  if (B <= 0) return 0;
  if (dmap && hipMemsetAsync(dmap, 0, (size_t)B * T * P * 4, st) != hipSuccess) { agd_set_error("attn_reg_loss: memset"); return -1; }
  hipLaunchKernelGGL(attn_reg_loss_kernel, dim3(B), dim3(256), 0, st, map, T, P, obj_idx, fg_idx, bg_idx, coef, loss_out, dmap);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// Cross-attention backward for one (batch row, head, 64-query tile) per single-wave block.
// ---------------------------------------------------------------------------------------
struct AttnBwdP {
  const bf16_t* q; int ldq; long long sq;          // [B][N][C] (+ head * D)
  const bf16_t* k; const bf16_t* v; int ldkv; long long skv;   // [B][T][2C]: k at +0, v at +C
  const bf16_t* dout; int ldo; long long so;       // dO [B][N][C] or NULL
  const float* dmap; int b0;                       // [B - b0][T][N] or NULL: gradient of the head-mean map of batch rows >= b0
  bf16_t* dq;                                      // [B][N][C]
  float* dk_part; float* dv_part;                  // [B*H][ntiles][T][D]
  int B, H, N, T, ntiles; float scale;
};

template <int D>
__global__ __launch_bounds__(64) void attn_bwd_kernel(const AttnBwdP p) {
  constexpr int TP = 97;                            // row pitch of the [64][T] fp32 arrays (T <= 96), odd -> conflict-free rows
  constexpr int QP = D + 2;                         // bf16 row pitch of the Q / dO tiles (4-byte aligned, breaks the power of two)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Ps = (float*)smem;                         // [64][TP]
  float* Ds = Ps + 64 * TP;                         // [64][TP]  dP, then dS
  bf16_t* Ks = (bf16_t*)(Ds + 64 * TP);             // [T][D]
  bf16_t* Vs = Ks + 96 * D;
  bf16_t* Qs = Vs + 96 * D;                         // [64][QP]
  bf16_t* Os = Qs + 64 * QP;                        // [64][QP]
  const int tid = threadIdx.x;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int q0 = blockIdx.x * 64, n = q0 + tid;
  const bool ok = n < p.N;
  const bool has_do = p.dout != nullptr, has_map = p.dmap != nullptr && b >= p.b0;
  const int T = p.T;
  const bf16_t* kb = p.k + b * p.skv + h * D;
  const bf16_t* vb = p.v + b * p.skv + h * D;
  for (int i = tid; i < T * D; i += 64) { const int t = i / D, d = i - t * D; Ks[i] = kb[(long long)t * p.ldkv + d]; Vs[i] = vb[(long long)t * p.ldkv + d]; }
  for (int i = tid; i < 64 * D; i += 64) {
    const int r = i / D, d = i - r * D; const bool rok = q0 + r < p.N;
    Qs[r * QP + d] = rok ? p.q[b * p.sq + (long long)(q0 + r) * p.ldq + h * D + d] : (bf16_t)0;
    Os[r * QP + d] = (rok && has_do) ? p.dout[b * p.so + (long long)(q0 + r) * p.ldo + h * D + d] : (bf16_t)0;
  }
  __syncthreads();
  const bf16_t* qr = Qs + tid * QP;
  const bf16_t* orow = Os + tid * QP;
  float* pr = Ps + tid * TP;
  float* dr = Ds + tid * TP;
  // 1. scores, softmax (hook.py:108)
  float mx = -INFINITY;
  for (int t = 0; t < T; ++t) {
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += bf2f(qr[d]) * bf2f(Ks[t * D + d]);
    s *= p.scale; pr[t] = s; mx = fmaxf(mx, s);
  }
  float sum = 0.f;
  for (int t = 0; t < T; ++t) { const float e = __expf(pr[t] - mx); pr[t] = e; sum += e; }
  const float inv = 1.0f / sum;
  // 2. dP = dMap / H (hook.py:55 head mean; :48-49 only the kept batch rows) + dO V^T (hook.py:114) ; rowdot = sum dP P
  float rowdot = 0.f;
  const float invh = 1.0f / (float)p.H;
  for (int t = 0; t < T; ++t) {
    const float pt = pr[t] * inv; pr[t] = pt;
    float dp = 0.f;
    if (has_map && ok) dp = p.dmap[((long long)(b - p.b0) * T + t) * p.N + n] * invh;
    if (has_do) { float a = 0.f; for (int d = 0; d < D; ++d) a += bf2f(orow[d]) * bf2f(Vs[t * D + d]); dp += a; }
    dr[t] = dp; rowdot += dp * pt;
  }
  // 3. dS = P o (dP - rowdot)   (softmax backward)
  for (int t = 0; t < T; ++t) dr[t] = ok ? pr[t] * (dr[t] - rowdot) : 0.f;
  if (!ok) for (int t = 0; t < T; ++t) pr[t] = 0.f;
  // 4. dQ = scale dS K
  if (ok) {
    bf16_t* dq = p.dq + b * p.sq + (long long)n * p.ldq + h * D;
    for (int d0 = 0; d0 < D; d0 += 8) {
      float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int t = 0; t < T; ++t) {
        const float ds = dr[t];
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += ds * bf2f(Ks[t * D + d0 + e]);
      }
      u32x4 pk;
      pk[0] = pack_bf2(a[0] * p.scale, a[1] * p.scale); pk[1] = pack_bf2(a[2] * p.scale, a[3] * p.scale);
      pk[2] = pack_bf2(a[4] * p.scale, a[5] * p.scale); pk[3] = pack_bf2(a[6] * p.scale, a[7] * p.scale);
      *(u32x4*)(dq + d0) = pk;
    }
  }
  __syncthreads();
  // 5. per-tile partials: dK[t][d] = scale sum_n dS[n][t] Q[n][d] ; dV[t][d] = sum_n P[n][t] dO[n][d]
  float* dkp = p.dk_part + ((long long)bh * p.ntiles + blockIdx.x) * T * D;
  float* dvp = p.dv_part + ((long long)bh * p.ntiles + blockIdx.x) * T * D;
  for (int i = tid; i < T * D; i += 64) {
    const int t = i / D, d = i - t * D;
    float ak = 0.f, av = 0.f;
    for (int r = 0; r < 64; ++r) { ak += Ds[r * TP + t] * bf2f(Qs[r * QP + d]); av += Ps[r * TP + t] * bf2f(Os[r * QP + d]); }
    dkp[i] = ak * p.scale; dvp[i] = av;
  }
}

// dkv[b][t][h*D + d] (dK) and [C + h*D + d] (dV) = ordered sum over the query tiles
__global__ void kv_grad_reduce_kernel(const float* __restrict__ dk_part, const float* __restrict__ dv_part, int H, int T, int D, int ntiles,
                                      long long total, bf16_t* __restrict__ dkv) {
  const int C = H * D;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int d = (int)(i % D); const int t = (int)((i / D) % T); const long long bh = i / ((long long)D * T);
    const int b = (int)(bh / H), h = (int)(bh % H);
    float ak = 0.f, av = 0.f;
    for (int j = 0; j < ntiles; ++j) {
      const long long o = ((bh * ntiles + j) * T + t) * D + d;
      ak += dk_part[o]; av += dv_part[o];
    }
    bf16_t* row = dkv + ((long long)b * T + t) * 2 * C;
    row[h * D + d] = f2bf(ak); row[C + h * D + d] = f2bf(av);
  }
}

template <int D>
static int launch_bwd_d(const AttnBwdP& p, hipStream_t st) {
  constexpr int lds = 2 * 64 * 97 * 4 + 2 * 96 * D * 2 + 2 * 64 * (D + 2) * 2;
  auto kfn = attn_bwd_kernel<D>;
  if (lds > 65536) {
    static std::atomic<bool> attr[AGD_MAX_DEVICES] = {};
    int dev = 0; HIP_CHECK_RET(hipGetDevice(&dev));
    if (dev < 0 || dev >= AGD_MAX_DEVICES) { agd_set_error("attn_bwd: device ordinal %d out of range", dev); return -1; }
    if (!attr[dev]) { HIP_CHECK_RET(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr[dev] = true; }
  }
  hipLaunchKernelGGL(kfn, dim3(p.ntiles, p.B * p.H), dim3(64), lds, st, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// q [B][N][C], kv [B][T][2C], dout [B][N][C] or NULL (all bf16), dmap fp32 [B - b0][T][N] or NULL
// -> dq [B][N][C] bf16, dkv [B][T][2C] bf16; part: workspace of 2 * B*H * ntiles * T * D floats
int launch_attention_backward(const bf16_t* q, const bf16_t* kv, const bf16_t* dout, const float* dmap, int b0, int B, int H, int D, int N, int T,
                              float scale, bf16_t* dq, bf16_t* dkv, float* part, hipStream_t st) {
  if (T > 96 || T < 1) { agd_set_error("attention backward: %d keys unsupported (1..96)", T); return -1; }
  const int C = H * D;
  AttnBwdP p{};
  p.q = q; p.ldq = C; p.sq = (long long)N * C; p.k = kv; p.v = kv + C; p.ldkv = 2 * C; p.skv = (long long)T * 2 * C;
  p.dout = dout; p.ldo = C; p.so = (long long)N * C; p.dmap = dmap; p.b0 = b0; p.dq = dq;
  p.B = B; p.H = H; p.N = N; p.T = T; p.ntiles = (N + 63) / 64; p.scale = scale;
  p.dk_part = part; p.dv_part = part + (long long)B * H * p.ntiles * T * D;
  int rc;
  switch (D) {
    case 32: rc = launch_bwd_d<32>(p, st); break;
    case 40: rc = launch_bwd_d<40>(p, st); break;
    case 64: rc = launch_bwd_d<64>(p, st); break;
    case 80: rc = launch_bwd_d<80>(p, st); break;
    case 128: rc = launch_bwd_d<128>(p, st); break;
    case 160: rc = launch_bwd_d<160>(p, st); break;
    default: agd_set_error("attention backward: unsupported head dim %d", D); return -1;
  }
  if (rc) return rc;
  const long long total = (long long)B * H * T * D;
  hipLaunchKernelGGL(kv_grad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p.dk_part, p.dv_part, H, T, D, p.ntiles, total, dkv);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
long long attention_backward_ws_floats(int B, int H, int D, int N, int T) { return 2LL * B * H * ((N + 63) / 64) * T * D; }

// bf16 [R][Cc] -> [Cc][R]  (weights transposed once per layer for the input-gradient GEMMs)
__global__ void transpose_bf16_kernel(const bf16_t* __restrict__ in, int R, int Cc, bf16_t* __restrict__ out) {
  __shared__ bf16_t tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += 8) { const int r = by + j, c = bx + threadIdx.x; tile[j][threadIdx.x] = (r < R && c < Cc) ? in[(long long)r * Cc + c] : (bf16_t)0; }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) { const int c = bx + j, r = by + threadIdx.x; if (c < Cc && r < R) out[(long long)c * R + r] = tile[threadIdx.x][j]; }
}
int launch_transpose_bf16(const bf16_t* in, int R, int Cc, bf16_t* out, hipStream_t st) {
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((Cc + 31) / 32, (R + 31) / 32), dim3(32, 8), 0, st, in, R, Cc, out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
