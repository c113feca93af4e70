// Cross-attention of the C = 1280 transformer blocks (16 x 16 and 8 x 8 maps) against PER-IMAGE PRE-MULTIPLIED context matrices.
//
// Replaces on the reference path: attn2 of BasicTransformerBlock through the processor seam -- the op sequence of
// reference data_generation/hook.py:91-120 (to_q, head split, softmax(scale q k^T), P v, head merge, to_out) behind diffusers' norm2,
// with the recorder's side output (hook.py:110-112 / daam's per-(layer, head) sums).
//
// The text context is fixed for a whole prompt batch, so everything that depends on it alone is multiplied out ONCE per
// agd_set_context (DESIGN section 4 'attn2 against pre-multiplied context matrices'):
//     S[m][(h,t)]  = scale * sum_d q[m][h,d] k[t][h,d],  q = LN(x) Wq^T
//                  = rstd_m (sum_c x[m][c] K''[(h,t)][c] - mu_m cs[(h,t)]) + bs[(h,t)]
//         K''[(h,t)][c] = gamma[c] scale sum_d k[t][h,d] Wq[(h,d)][c]      cs = row sums of the bf16 K''      bs[(h,t)] = scale sum_d k[t][h,d] (Wq beta)[(h,d)]
//     out[m][n]    = sum_h sum_t P_h[m][t] (sum_d v[t][h,d] Wo[n][(h,d)]) + bo[n] + x[m][n]
//                  = sum_(h,t) P[m][(h,t)] V''[n][(h,t)] + bo[n] + x[m][n]
// to_q, the attention and to_out become TWO GEMMs (K = 1280 -> N = 8 x 80, K = 8 x 80 -> N = 1280) with 2.3 x fewer MACs than the three launches
// they replace at d = 160, no q / o round trip, and no 19 us latency chain of a 77-key attention kernel.  The softmax and the recorder's
// read-modify-write sit in the first GEMM's epilogue: a lane of the swapped-operand MFMA tile holds 20 consecutive tokens of one pixel, the four
// lanes of a pixel one head's 80 (padded) tokens, so a row's max / sum are in-lane + two shuffles, and the recorder rows [token][pixel] receive
// 16 consecutive pixels (64 B) per token and store instruction.
//
// Kernels (gfx950 only):
//   premul_gemm_kernel    the context products K'' and V'' (once per prompt batch; small strided GEMMs straight from global memory)
//   premul_rowsum_kernel  cs and bs
//   xattn_s_kernel        S GEMM (LDS-DMA ring as igemm.hip, per-image weights) + folded LayerNorm + softmax + recorder + P (bf16)
// The second GEMM is an ordinary igemm launch with per-image weights (IgemmP::w_per_image) on 64 x 160 tiles.
#include "kernels.h"

typedef __attribute__((ext_vector_type(2))) float f32x2_;
typedef u32x4 __attribute__((aligned(8))) u32x4_a8_;

// ---------------------------------------------------------------------------------------
// out[z][m][n] = bf16(alpha * colscale[n] * sum_k A[z][m][k] B[z][n][k]),  z = (b, h);  rows m >= Mv and columns n >= Nv are written as zeros
// (the padded token rows / columns).  K % 8 == 0, 16-byte aligned rows.  One wave = a 16 x 80 strip, a workgroup 64 x 80.
// ---------------------------------------------------------------------------------------
struct PremulP {
  const bf16_t* A; long long sAb, sAh; int lda;
  const bf16_t* B; long long sBb, sBh; int ldb;
  bf16_t* O; long long sOb, sOh; int ldo;
  int M, N, K, Mv, Nv, H;
  float alpha; const float* colscale;
};

__global__ __launch_bounds__(256) void premul_gemm_kernel(const PremulP p) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int z = blockIdx.z, b = z / p.H, h = z - b * p.H;
  const int m0 = blockIdx.x * 64 + wid * 16, n0 = blockIdx.y * 80;
  if (m0 >= p.M) return;
  const bf16_t* A = p.A + b * p.sAb + h * p.sAh;
  const bf16_t* B = p.B + b * p.sBb + h * p.sBh;
  bf16_t* O = p.O + b * p.sOb + h * p.sOh;
  const int fr = lane & 15, kc = (lane >> 4) * 8;
  const int am = m0 + fr;
  const bool a_ok = am < p.Mv;
  f32x4 acc[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < p.K; k0 += 32) {
    const bool k_ok = k0 + kc < p.K;                   // K % 8 == 0: the last 32-deep step may be ragged (head dim 80 = 2 x 32 + 16)
    bf16x8 a = {};
    if (a_ok && k_ok) a = *(const bf16x8*)(A + (long long)am * p.lda + k0 + kc);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int bn = n0 + j * 16 + fr;
      bf16x8 w = {};
      if (bn < p.Nv && k_ok) w = *(const bf16x8*)(B + (long long)bn * p.ldb + k0 + kc);
      // operands swapped: acc[r] = out[m = lane & 15][n = 16 j + 4 (lane >> 4) + r] -- four consecutive columns of one row per lane
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, acc[j], 0, 0, 0);
    }
  }
  const int q = lane >> 4;
  if (am < p.M) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int n = n0 + j * 16 + 4 * q;
      if (n + 4 <= p.N) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float cs = p.colscale ? p.colscale[n + r] : 1.0f; v[r] = (a_ok && n + r < p.Nv) ? acc[j][r] * p.alpha * cs : 0.f; }
        u32x2 pk; pk[0] = pack_bf2(v[0], v[1]); pk[1] = pack_bf2(v[2], v[3]);
        *(u32x2*)(O + (long long)am * p.ldo + n) = pk;
      }
    }
  }
}

// cs[row] = sum_c bf16 K''[row][c];  bs[row] = scale * sum_d k[b][t][h D + d] wqb[h D + d];  row = (b, h, t) with t < TP (rows t >= T: zeros)
__global__ __launch_bounds__(256) void premul_rowsum_kernel(const bf16_t* __restrict__ kpp, int C, const bf16_t* __restrict__ kv, int ldkv, long long skv,
                                                            const float* __restrict__ wqb, int H, int D, int T, int TP, float scale, float* __restrict__ cs,
                                                            float* __restrict__ bs, int rows) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int t = row % TP, bh = row / TP, h = bh % H, b = bh / H;
  float s = 0.f, d_ = 0.f;
  if (t < T) {
    const bf16_t* r = kpp + (long long)row * C;
    for (int c = lane * 8; c < C; c += 512) {
      const u32x4 v = *(const u32x4*)(r + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) s += __uint_as_float(v[e] << 16) + __uint_as_float(v[e] & 0xFFFF0000u);
    }
    const bf16_t* kr = kv + b * skv + (long long)t * ldkv + h * D;
    for (int d = lane; d < D; d += 64) d_ += bf2f(kr[d]) * wqb[h * D + d];
  }
  for (int o = 32; o >= 1; o >>= 1) { s += __shfl_xor(s, o); d_ += __shfl_xor(d_, o); }
  if (lane == 0) { cs[row] = s; bs[row] = d_ * scale; }
}

// out[j] = sum_c W[j][c] v[c]  (Wq . norm2.bias at load time), one wave per row
__global__ __launch_bounds__(256) void matvec_bf16_kernel(const bf16_t* __restrict__ W, const float* __restrict__ v, float* __restrict__ out, int N, int K) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  float a = 0.f;
  for (int c = lane; c < K; c += 64) a += bf2f(W[(long long)row * K + c]) * v[c];
  for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o);
  if (lane == 0) out[row] = a;
}
int launch_matvec_bf16(const bf16_t* W, const float* v, float* out, int N, int K, hipStream_t st) {
  hipLaunchKernelGGL(matvec_bf16_kernel, dim3((N + 3) / 4), dim3(256), 0, st, W, v, out, N, K);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// per-row (sum, sum of squares) of a bf16 [M][C] activation: the one-slot form of the LayerNorm-fold producers' row statistics (single-op entry points only:
// in the walk the GEMM that writes the residual stream emits them)
__global__ __launch_bounds__(256) void rowstat_bf16_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, int M, int C) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float s = 0.f, q = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = bf2f(x[(long long)row * C + c]); s += v; q += v * v; }
  for (int o = 32; o >= 1; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
  if (lane == 0) { out[(long long)row * 2] = s; out[(long long)row * 2 + 1] = q; }
}
int launch_rowstat_bf16(const bf16_t* x, float* out, int M, int C, hipStream_t st) {
  hipLaunchKernelGGL(rowstat_bf16_kernel, dim3((M + 3) / 4), dim3(256), 0, st, x, out, M, C);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int launch_xattn_premul(const XattnPremulP& x, hipStream_t st) {
  const int C = x.C, H = x.H, D = C / H, TP = XATTN_TP;
  if (C % H || D % 8 || x.T < 1 || x.T > TP || (C % 80) || (C % 64)) { agd_set_error("xattn_premul: C %d heads %d tokens %d", C, H, x.T); return -1; }
  // K''[b][(h,t)][c] = gamma[c] scale sum_d k[b][t][h D + d] WqT[c][h D + d]
  PremulP k{};
  k.A = x.kv; k.sAb = x.skv; k.sAh = D; k.lda = x.ldkv;
  k.B = x.wqT; k.sBb = 0; k.sBh = D; k.ldb = C;
  k.O = x.kpp; k.sOb = (long long)H * TP * C; k.sOh = (long long)TP * C; k.ldo = C;
  k.M = TP; k.N = C; k.K = D; k.Mv = x.T; k.Nv = C; k.H = H; k.alpha = x.scale; k.colscale = x.gamma;
  hipLaunchKernelGGL(premul_gemm_kernel, dim3((TP + 63) / 64, C / 80, x.B * H), dim3(256), 0, st, k);
  HIP_CHECK_RET(hipGetLastError());
  // V''[b][n][(h,t)] = sum_d Wo[n][h D + d] v[b][t][h D + d]
  PremulP v{};
  v.A = x.wo; v.sAb = 0; v.sAh = D; v.lda = C;
  v.B = x.kv + C; v.sBb = x.skv; v.sBh = D; v.ldb = x.ldkv;
  v.O = x.vpp; v.sOb = (long long)C * H * TP; v.sOh = TP; v.ldo = H * TP;
  v.M = C; v.N = TP; v.K = D; v.Mv = C; v.Nv = x.T; v.H = H; v.alpha = 1.0f; v.colscale = nullptr;
  hipLaunchKernelGGL(premul_gemm_kernel, dim3(C / 64, 1, x.B * H), dim3(256), 0, st, v);
  HIP_CHECK_RET(hipGetLastError());
  const int rows = x.B * H * TP;
  hipLaunchKernelGGL(premul_rowsum_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x.kpp, C, x.kv, x.ldkv, x.skv, x.wqb, H, D, x.T, TP, x.scale, x.kcs, x.kbs, rows);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// S GEMM + folded LayerNorm + softmax + recorder.  One workgroup = 64 rows of one image x one head (80 padded tokens), K = C.
// Main loop as igemm.hip: A (raw residual-stream rows) and B (this image's K'' rows of this head) by LDS-DMA into a 4-stage ring, one
// barrier per 64-deep K step, counted vmcnt, 128-B LDS rows XOR-swizzled on the source side, MFMA operand roles swapped (D = W . X^T) with the
// weight rows permuted inside the tile (tile j, row 4q + r <-> column 20 q + 4 j + r) so that lane (q, pixel) owns tokens 20 q .. 20 q + 19.
// 4 waves, each 16 rows x 80 columns (5 accumulator tiles).
// ---------------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ void xs_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

__global__ __launch_bounds__(256) void xattn_s_kernel(const XattnSP p) {
  constexpr int BM = 64, BN = XATTN_TP, NI = 5, STAGES = 4;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;   // 8 KB + 10 KB
  constexpr int LPS = 5;                                   // LDS-DMA instructions per stage and wave: 2 A pieces + 3 B pieces (waves 2, 3: the third is dead)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  char* const sink = smem + STAGES * STAGE;                // 4 x 1 KiB: where the dead pieces land
  float* const lnst = (float*)(sink + 4096);               // [64] (mean, rstd)
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  // grid (images, row tiles of an image x heads): workgroups are dealt round-robin over the XCDs, so with the image on blockIdx.x every workgroup of an image lands on
  // ONE XCD (8 images: exactly one each) and that XCD's L2 holds the image's K'' (1.6 MB at C = 1280) once -- with the row tile on x each XCD pulled four images' matrices
  const int img = blockIdx.x, head = (int)blockIdx.y % p.H;
  const int m0 = img * p.HW + ((int)blockIdx.y / p.H) * BM;
  const bf16_t* baseA = p.x;
  const bf16_t* baseW = p.kpp + ((long long)img * p.H + head) * BN * p.C;

  // per-lane gather offsets (fixed for the kernel; the K advance is an SGPR offset)
  const int lrow = lane >> 3;
  unsigned avoff[2], bvoff[3];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (i * 4 + wid) * 8 + lrow;              // tile row this lane fetches; swizzle key = row & 7 = lrow
    avoff[i] = (unsigned)(((long long)(m0 + row) * p.C + ((lane & 7) ^ lrow) * 8) * 2);
  }
  bool b_live[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int pc = i * 4 + wid;                            // 8-row piece of the 80-row B tile
    b_live[i] = pc < BN / 8;
    const int row = (b_live[i] ? pc : 0) * 8 + lrow;
    const int key = (row & 3) | (((row / (4 * NI)) & 1) << 2);      // the fragment row that reads this weight row (permuted), mod 8
    bvoff[i] = (unsigned)(((long long)row * p.C + ((lane & 7) ^ key) * 8) * 2);
  }
  auto issue = [&](int slot, unsigned koff, bool live) {
    char* sA = ring + slot * STAGE;
    char* sB = sA + A_BYTES;
    const unsigned nr = live ? 0x7FFFFFF0u : 0u;
#pragma unroll
    for (int i = 0; i < 2; ++i) bufdma16(baseA, sA + (i * 4 + wid) * 1024, avoff[i], koff, nr);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (b_live[i]) bufdma16(baseW, sB + (i * 4 + wid) * 1024, bvoff[i], koff, nr);
      else bufdma16(baseW, sink + wid * 1024, bvoff[i], koff, 0u);          // keeps the per-wave vmcnt arithmetic uniform
    }
  };

  const int nk = p.C >> 6;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) issue(s, (unsigned)s * 128u, s < nk);
  // folded LayerNorm: (mean, rstd) of the tile's rows from the producer's per-N-tile partial sums, summed beside the prologue's DMA (as igemm.hip);
  // the epilogue reads them behind the K loop's barriers
  if (threadIdx.x < BM) {
    const int m = m0 + (int)threadIdx.x;
    float S = 0.f, Q = 0.f;
    for (int k = 0; k < p.ln_slots; ++k) { const f32x2_ v = *(const f32x2_*)(p.ln_stats + ((long long)m * p.ln_slots + k) * 2); S += v[0]; Q += v[1]; }
    const float mu = S * p.ln_invC;
    float var = Q * p.ln_invC - mu * mu; var = var < 0.f ? 0.f : var;
    *(f32x2_*)(lnst + threadIdx.x * 2) = f32x2_{mu, rsqrtf(var + p.ln_eps)};
  }

  f32x4 acc[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15;
  int foffA[2], foffB[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int sw = (((kk << 2) + (lane >> 4)) ^ (lane & 7)) << 4;
    foffA[kk] = frow * 128 + sw;
    foffB[kk] = (frow >> 2) * (4 * NI * 128) + (frow & 3) * 128 + sw;
  }
  for (int ks = 0; ks < nk; ++ks) {
    xs_wait_vmcnt<(STAGES - 2) * LPS>();
    asm volatile("s_barrier" ::: "memory");
    const int nxt = ks + STAGES - 1;
    const char* sA = ring + (ks % STAGES) * STAGE + wid * 16 * 128;
    const char* sB = ring + (ks % STAGES) * STAGE + A_BYTES;
    bf16x8 a0 = *(const bf16x8*)(sA + foffA[0]), a1 = *(const bf16x8*)(sA + foffA[1]);
    bf16x8 b0[NI], b1[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) { b0[j] = *(const bf16x8*)(sB + j * 512 + foffB[0]); b1[j] = *(const bf16x8*)(sB + j * 512 + foffB[1]); }
    issue(nxt % STAGES, (unsigned)nxt * 128u, nxt < nk);   // the slot step ks - 1 released (every wave is past this step's barrier)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j], a0, acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], a1, acc[j], 0, 0, 0);
  }
  // (the tail steps issued their pieces through zero-record descriptors: the hardware still writes zeros to LDS -- let them land)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: lane (q, px) holds S of pixel m = m0 + 16 wid + px for tokens 20 q + 4 j + r
  const int q = lane >> 4, px = lane & 15;
  const int row_t = wid * 16 + px, m = m0 + row_t;
  const f32x2_ st = *(const f32x2_*)(lnst + row_t * 2);
  const float mu = st[0], rstd = st[1];
  const float* cs = p.kcs + ((long long)img * p.H + head) * BN + q * 20;
  const float* bs = p.kbs + ((long long)img * p.H + head) * BN + q * 20;
  float s[20];
  float mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const f32x4 c4 = *(const f32x4*)(cs + 4 * j), b4 = *(const f32x4*)(bs + 4 * j);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = 4 * j + r;
      const float v = rstd * (acc[j][r] - mu * c4[r]) + b4[r];
      s[e] = (q * 20 + e < p.T) ? v : -3.0e38f;
      mx = fmaxf(mx, s[e]);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = xhalf_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < 20; ++e) { s[e] = (q * 20 + e < p.T) ? __builtin_amdgcn_exp2f((s[e] - mx) * 1.44269504088896340736f) : 0.f; sum += s[e]; }
  sum += __shfl_xor(sum, 16); sum = xhalf_sum(sum);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int e = 0; e < 20; ++e) s[e] *= inv;
  // recorder: rec[img - b0][head][t][pixel] += P (each (pixel, token) belongs to exactly one lane of the grid: plain read-modify-write, run-to-run identical)
  if (p.rec && img >= p.rec_b0) {
    float* rp = p.rec + (long long)(img - p.rec_b0) * p.rec_img_stride + (long long)head * p.rec_head_stride + (m - img * p.HW);
    float old[20];
#pragma unroll
    for (int e = 0; e < 20; ++e) { const int t = q * 20 + e; old[e] = (t < p.rec_T) ? rp[(long long)t * p.HW] : 0.f; }
#pragma unroll
    for (int e = 0; e < 20; ++e) { const int t = q * 20 + e; if (t < p.rec_T) rp[(long long)t * p.HW] = old[e] + s[e]; }
  }
  bf16_t* op = p.P + (long long)m * (p.H * BN) + head * BN + q * 20;       // 40 bytes per lane, 8-byte aligned
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    u32x4 pk;
    pk[0] = pack_bf2(s[8 * c], s[8 * c + 1]); pk[1] = pack_bf2(s[8 * c + 2], s[8 * c + 3]);
    pk[2] = pack_bf2(s[8 * c + 4], s[8 * c + 5]); pk[3] = pack_bf2(s[8 * c + 6], s[8 * c + 7]);
    *(u32x4_a8_*)(op + 8 * c) = pk;
  }
  u32x2 pk2; pk2[0] = pack_bf2(s[16], s[17]); pk2[1] = pack_bf2(s[18], s[19]);
  *(u32x2*)(op + 16) = pk2;
}

int launch_xattn_s(const XattnSP& p, hipStream_t st) {
  if (p.M % 64 || p.HW % 64 || p.M % p.HW || p.C % 64 || p.T < 1 || p.T > XATTN_TP || !p.ln_stats || p.ln_slots < 1) {
    agd_set_error("xattn_s: M %d HW %d C %d T %d slots %d", p.M, p.HW, p.C, p.T, p.ln_slots); return -1;
  }
  if ((long long)p.M * p.C * 2 >= (1LL << 31) || (long long)XATTN_TP * p.C * 2 >= (1LL << 31)) { agd_set_error("xattn_s: 32-bit offsets"); return -1; }
  constexpr int lds = 4 * (64 + XATTN_TP) * 128 + 4096 + 64 * 8;
  static std::atomic<bool> attr[AGD_MAX_DEVICES] = {};
  int dev = 0; HIP_CHECK_RET(hipGetDevice(&dev));
  if (dev < 0 || dev >= AGD_MAX_DEVICES) { agd_set_error("xattn_s: device ordinal %d out of range", dev); return -1; }
  if (!attr[dev]) { HIP_CHECK_RET(hipFuncSetAttribute((const void*)xattn_s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr[dev] = true; }
  hipLaunchKernelGGL(xattn_s_kernel, dim3(p.M / p.HW, (p.HW / 64) * p.H), dim3(256), lds, st, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
