// igemm epilogue, straight from the accumulator registers (no LDS round trip, no barrier).
//
// The main loop issues its MFMAs with the operand roles swapped -- D = W_tile . X_tile^T -- so a lane's accumulator
// registers hold CONSECUTIVE OUTPUT CHANNELS of ONE pixel (16x16x32 C/D map: col = lane & 15 = pixel, row =
// 4 * (lane >> 4) + reg = weight row), and the weight rows of the wave's N range are assigned to the 16-row MFMA tiles
// as   tile j, row 4q + r  <->  wave-local column q * 4NI + 4j + r   (a free permutation: the fragment read address is
// per lane).  Lane (q, pixel) therefore owns the 4*NI consecutive channels  q*4NI .. q*4NI + 4NI-1  of its pixel in
// every pixel tile i: bias / time-embedding row / residual / GEGLU / activation are lane-local and the result leaves as
// 16-byte (8-byte for the 160-wide tile) row chunks -- the same store instructions the LDS-staged epilogue issued, minus
// 64 KB of LDS writes + reads and two barriers per tile (the epilogue was a third of a K = 320 tile's LDS time).
// Split-K writes its fp32 slab rows the same way.
#pragma once
#include "kernels.h"

// LayerNorm-fold consumer: per-row mean / rstd of the A rows from the producer's per-N-tile partial sums [M][slots] float2.
// The slot loop is OUTSIDE the (unrolled) row loop, so the MI loads of one slot are independent and in flight together: with the
// rows outside, every row paid `slots` dependent L2 round trips one after the other (8-row wave tiles: +17 us on an L0 qkv launch).
// Summation order per row is unchanged (slot 0, 1, ...).  lds_stats != nullptr: the kernel prologue already left (mean, rstd) of
// the tile's rows in LDS (igemm8p.h).
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef u32x4 __attribute__((aligned(8))) u32x4_a8;      // 16-byte access at an 8-byte aligned address
template <int MI>
AGD_DEV void ln_row_stats(const IgemmP& p, int mrow0, int row0_tile, const float* lds_stats, float (&lmu)[MI], float (&lrs)[MI]) {
  if (lds_stats) {
#pragma unroll
    for (int i = 0; i < MI; ++i) { const f32x2 v = *(const f32x2*)(lds_stats + (row0_tile + i * 16) * 2); lmu[i] = v[0]; lrs[i] = v[1]; }
    return;
  }
  float S[MI], Q[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) { S[i] = 0.f; Q[i] = 0.f; }
  for (int k = 0; k < p.ln_slots; ++k) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = mrow0 + i * 16;
      if (m < p.M) { const f32x2 v = *(const f32x2*)(p.ln_stats + ((long long)m * p.ln_slots + k) * 2); S[i] += v[0]; Q[i] += v[1]; }
    }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const float mu = S[i] * p.ln_invC;
    float var = Q[i] * p.ln_invC - mu * mu; var = var < 0.f ? 0.f : var;
    lmu[i] = mu; lrs[i] = rsqrtf(var + p.ln_eps);
  }
}

// FAST_ONLY: the launcher guarantees that every lane's channel run is whole and aligned (N, ldo, ldr multiples of the run): the
// element-by-element path for ragged tiles is compiled out (igemm8p.h: with 128 accumulators per lane it would put them in scratch)
// The waves of a workgroup that hold no output (igemm_kernel's second K group) pass exactly the workgroup barriers igemm_epilogue passes --
// KEEP IN STEP with the __syncthreads() calls below (all of them sit at function level behind wave-uniform kernel arguments).
AGD_DEV void igemm_epilogue_ghost(const IgemmP& p) {
  if (p.colstat_out) { __syncthreads(); __syncthreads(); __syncthreads(); }
  if (p.rowstat_out) { __syncthreads(); __syncthreads(); }
}

template <int BM, int BN, int WM, int WN, int GEGLU, int SPLITK, int FAST_ONLY = 0>
AGD_DEV void igemm_epilogue(const IgemmP& p, f32x4 (&acc)[BM / WM / 16][BN / WN / 16], char* smem, int lane, int wm, int wn, int m0,
                            int n0, int tn, int bz, const float* lds_stats = nullptr) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MI = WTM / 16, NI = WTN / 16;
  static_assert(!GEGLU || NI == 4, "GEGLU epilogue: a lane's 16 columns must be one [8 values | 8 gates] group");
  constexpr int CA = 4 * NI;                        // accumulator columns per lane and pixel
  constexpr int CW = GEGLU ? CA / 2 : CA;           // output channels per lane and pixel
  constexpr int SV = (CW % 8 == 0) ? 8 : 4;         // elements per vector access (16 B of bf16, else 8 B)
  const int q = lane >> 4, px = lane & 15;
  const int HWo = p.Hout * p.Wout;
  const float inv_hwo = 1.0f / (float)HWo;          // rowadd image index (M < 2^24 is checked by the launcher)
  const int Nout = GEGLU ? p.N / 2 : p.N;
  const int ncol = n0 + wn * WTN + q * CA;          // first accumulator column of this lane (global)
  const int no = GEGLU ? ncol / 2 : ncol;           // first output channel ([8 values | 8 gates] per 16 columns)
  const int mrow0 = m0 + wm * WTM + px;

  if constexpr (SPLITK) {
    float* part = p.splitk_ws + ((long long)blockIdx.z * p.M) * p.N;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = mrow0 + i * 16;
      if (m >= p.M) continue;
      float* op = part + (long long)m * p.N + no;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = no + 4 * j;
        if (n + 4 <= p.N) *(f32x4*)(op + 4 * j) = acc[i][j] * p.alpha;          // N % 4 == 0 (launcher)
        else for (int r = 0; r < 4; ++r) if (n + r < p.N) op[4 * j + r] = acc[i][j][r] * p.alpha;
      }
    }
    return;
  } else {
    // producer side of the GroupNorm statistics (p.colstat_out): the wave's bf16-rounded output tile goes to LDS as well, then
    // lane = column sums the rows -- costs registers for nothing else; the ring is free once every wave has left the K loop
    bf16_t* ctile = nullptr;
    if (p.colstat_out) {
      __syncthreads();
      ctile = (bf16_t*)smem + (wm * WN + wn) * (WTM * WTN);
    }
    // producer side of the LayerNorm fold: per-row (sum, sum of squares) of the bf16-rounded outputs, this lane's share
    float rs[MI], rq[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) { rs[i] = 0.f; rq[i] = 0.f; }
    const bool full = no + CW <= Nout;
    const bool fast = FAST_ONLY ? full : (full && (no % SV) == 0 && (p.ldo % (p.out_f32 ? 4 : SV)) == 0 && (!p.residual || (p.ldr % SV) == 0));
    if (fast) {
      // residual chunks first: their latency overlaps the bias loads and the arithmetic.  Tall wave tiles (MI = 8, igemm8p.h) keep
      // only RA rows in flight -- all eight would cost 64 registers beside the 128 accumulators -- and refill a slot once it is consumed
      constexpr int RA = MI > 4 ? 2 : MI;
      unsigned rres[RA][CW / 2];
      auto load_res = [&](int i, int slot) {
        const int m = mrow0 + i * 16;
        if (m < p.M) {
          const bf16_t* rp = p.residual + bz * p.sR + (long long)m * p.ldr + no;
          if constexpr (SV == 4 && CW % 8 == 4) {
            // the 160-wide tile's 20 channels per lane = 40 bytes, 8-byte aligned: 16 + 16 + 8-byte accesses instead of five 8-byte ones (the
            // row-per-lane epilogue is load / store ISSUE bound, MI355X_MICROARCH.md 'attention epilogue store tail')
#pragma unroll
            for (int c = 0; c < CW / 8; ++c) { const u32x4 t = *(const u32x4_a8*)(rp + 8 * c); rres[slot][4 * c] = t[0]; rres[slot][4 * c + 1] = t[1]; rres[slot][4 * c + 2] = t[2]; rres[slot][4 * c + 3] = t[3]; }
            const u32x2 t = *(const u32x2*)(rp + CW - 4); rres[slot][CW / 2 - 2] = t[0]; rres[slot][CW / 2 - 1] = t[1];
          } else {
#pragma unroll
          for (int c = 0; c < CW / SV; ++c) {
            if constexpr (SV == 8) { const u32x4 t = *(const u32x4*)(rp + 8 * c); rres[slot][4 * c] = t[0]; rres[slot][4 * c + 1] = t[1]; rres[slot][4 * c + 2] = t[2]; rres[slot][4 * c + 3] = t[3]; }
            else { const u32x2 t = *(const u32x2*)(rp + 4 * c); rres[slot][2 * c] = t[0]; rres[slot][2 * c + 1] = t[1]; }
          }
          }
        }
      };
      if (p.residual) {
#pragma unroll
        for (int i = 0; i < RA; ++i) load_res(i, i);
      }
      float hb[CW], hg[GEGLU ? CW : 1];
      // LayerNorm folded into this GEMM (A rows are raw, W carries gamma): per-row mean / rstd from the producer's partial sums,
      // out = rstd (acc - mean colsum[n]) + bias'[n]
      const bool lnf = p.ln_stats != nullptr;
      float lmu[MI], lrs[MI], cs[CW], csg[GEGLU ? CW : 1];
      if (lnf) {
        ln_row_stats<MI>(p, mrow0, wm * WTM + px, lds_stats, lmu, lrs);
#pragma unroll
        for (int c = 0; c < CW / 4; ++c) *(f32x4*)&cs[4 * c] = *(const f32x4*)(p.ln_cs + no + 4 * c);
        if constexpr (GEGLU) {
#pragma unroll
          for (int c = 0; c < CW / 4; ++c) *(f32x4*)&csg[4 * c] = *(const f32x4*)(p.ln_cs + Nout + no + 4 * c);
        }
      }
#pragma unroll
      for (int e = 0; e < CW; ++e) hb[e] = 0.f;
      if (p.bias && p.bias_mode == 1) {
#pragma unroll
        for (int c = 0; c < CW / 4; ++c) *(f32x4*)&hb[4 * c] = *(const f32x4*)(p.bias + no + 4 * c);
      }
      if constexpr (GEGLU) {
#pragma unroll
        for (int c = 0; c < CW / 4; ++c) *(f32x4*)&hg[4 * c] = *(const f32x4*)(p.bias + Nout + no + 4 * c);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = mrow0 + i * 16;
        if (m >= p.M) {
          if (ctile) { for (int e = 0; e < CA; ++e) ctile[(i * 16 + px) * WTN + q * CA + e] = 0; }
          continue;
        }
        float v[CW];
        if constexpr (GEGLU) {
          if (lnf) {
#pragma unroll
            for (int e = 0; e < CW; ++e)
              v[e] = (lrs[i] * (acc[i][e >> 2][e & 3] - lmu[i] * cs[e]) + hb[e]) *
                     gelu_erf_f(lrs[i] * (acc[i][NI / 2 + (e >> 2)][e & 3] - lmu[i] * csg[e]) + hg[e]);
          } else {
#pragma unroll
            for (int e = 0; e < CW; ++e)
              v[e] = (acc[i][e >> 2][e & 3] * p.alpha + hb[e]) * gelu_erf_f(acc[i][NI / 2 + (e >> 2)][e & 3] * p.alpha + hg[e]);
          }
        } else {
          const float bm = p.bias_mode == 2 ? p.bias[m] : 0.f;
          if (lnf) {
#pragma unroll
            for (int e = 0; e < CW; ++e) v[e] = lrs[i] * (acc[i][e >> 2][e & 3] - lmu[i] * cs[e]) + hb[e];
          } else {
#pragma unroll
            for (int e = 0; e < CW; ++e) v[e] = acc[i][e >> 2][e & 3] * p.alpha + hb[e] + bm;
          }
          if (p.rowadd) {
            const float* ra = p.rowadd + (long long)fast_udiv(m, HWo, inv_hwo) * p.rowadd_ld + no;
#pragma unroll
            for (int c = 0; c < CW / 4; ++c) {
              const f32x4 t = *(const f32x4*)(ra + 4 * c);
              v[4 * c] += t[0]; v[4 * c + 1] += t[1]; v[4 * c + 2] += t[2]; v[4 * c + 3] += t[3];
            }
          }
        }
        if (p.residual) {
#pragma unroll
          for (int e = 0; e < CW; e += 2) {
            v[e] += __uint_as_float(rres[i % RA][e >> 1] << 16);
            v[e + 1] += __uint_as_float(rres[i % RA][e >> 1] & 0xFFFF0000u);
          }
          if (i + RA < MI) load_res(i + RA, i % RA);
        }
        if (p.act == 1) {
#pragma unroll
          for (int e = 0; e < CW; ++e) v[e] = silu_f(v[e]);
        } else if (p.act == 2) {
#pragma unroll
          for (int e = 0; e < CW; ++e) v[e] = v[e] / (1.0f + __expf(-1.702f * v[e]));
        } else if (p.act == 3) {
#pragma unroll
          for (int e = 0; e < CW; ++e) v[e] = gelu_erf_f(v[e]);
        }
        if constexpr (!FAST_ONLY) {      // the 8-phase kernels are never LayerNorm-statistics producers (their launcher refuses): 16 registers less
          if (p.rowstat_out) {
#pragma unroll
            for (int e = 0; e < CW; ++e) { const float r = bf2f(f2bf(v[e])); rs[i] += r; rq[i] += r * r; }
          }
        }
        if (p.out_f32) {
          float* op = (float*)p.out + bz * p.sO + (long long)m * p.ldo + no;
#pragma unroll
          for (int c = 0; c < CW / 4; ++c) *(f32x4*)(op + 4 * c) = f32x4{v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]};
        } else {
          long long orow = m; int ocol = no;
          if (p.ups4) {      // phase conv of an upsampling conv: source pixel (b, i, j), phase (a, b') = no / Cout -> output pixel (b, 2 i + a, 2 j + b') of the 2x map, channel no % Cout
            const int ph = no / p.ups4, bimg = fast_udiv(m, HWo, inv_hwo), rem = m - bimg * HWo, ii = rem / p.Wout, jj = rem - ii * p.Wout;
            orow = ((long long)bimg * 2 * p.Hout + 2 * ii + (ph >> 1)) * (2 * p.Wout) + 2 * jj + (ph & 1);
            ocol = no - ph * p.ups4;
          }
          bf16_t* op = (bf16_t*)p.out + bz * p.sO + orow * p.ldo + ocol;
          bf16_t* lp = ctile ? ctile + (i * 16 + px) * WTN + q * CA : nullptr;
          if constexpr (SV == 4 && CW % 8 == 4) {
#pragma unroll
            for (int c = 0; c < CW / 8; ++c) {
              u32x4 pk;
              pk[0] = pack_bf2(v[8 * c], v[8 * c + 1]); pk[1] = pack_bf2(v[8 * c + 2], v[8 * c + 3]);
              pk[2] = pack_bf2(v[8 * c + 4], v[8 * c + 5]); pk[3] = pack_bf2(v[8 * c + 6], v[8 * c + 7]);
              *(u32x4_a8*)(op + 8 * c) = pk;
              if (lp) *(u32x4_a8*)(lp + 8 * c) = pk;
            }
            u32x2 pk; pk[0] = pack_bf2(v[CW - 4], v[CW - 3]); pk[1] = pack_bf2(v[CW - 2], v[CW - 1]);
            *(u32x2*)(op + CW - 4) = pk;
            if (lp) *(u32x2*)(lp + CW - 4) = pk;
          } else
#pragma unroll
          for (int c = 0; c < CW / SV; ++c) {
            if constexpr (SV == 8) {
              u32x4 pk;
              pk[0] = pack_bf2(v[8 * c], v[8 * c + 1]); pk[1] = pack_bf2(v[8 * c + 2], v[8 * c + 3]);
              pk[2] = pack_bf2(v[8 * c + 4], v[8 * c + 5]); pk[3] = pack_bf2(v[8 * c + 6], v[8 * c + 7]);
              *(u32x4*)(op + 8 * c) = pk;
              if (lp) *(u32x4*)(lp + 8 * c) = pk;
            } else {
              u32x2 pk;
              pk[0] = pack_bf2(v[4 * c], v[4 * c + 1]); pk[1] = pack_bf2(v[4 * c + 2], v[4 * c + 3]);
              *(u32x2*)(op + 4 * c) = pk;
              if (lp) *(u32x2*)(lp + 4 * c) = pk;
            }
          }
        }
      }
    } else if constexpr (!FAST_ONLY) {
      // ragged / unaligned tiles (conv_out's 4 channels, N tails): element by element
      float lmu[MI], lrs[MI];
      if (p.ln_stats) ln_row_stats<MI>(p, mrow0, wm * WTM + px, lds_stats, lmu, lrs);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = mrow0 + i * 16;
        if (ctile) { for (int e = 0; e < CA; ++e) ctile[(i * 16 + px) * WTN + q * CA + e] = 0; }
        if (m >= p.M) continue;
        const int img = p.rowadd ? fast_udiv(m, HWo, inv_hwo) : 0;
#pragma unroll
        for (int e = 0; e < CW; ++e) {
          const int n = no + e;
          if (n >= Nout) continue;
          float x;
          if constexpr (GEGLU) {
            float xv = acc[i][e >> 2][e & 3] * p.alpha, xg = acc[i][NI / 2 + (e >> 2)][e & 3] * p.alpha;
            if (p.ln_stats) { xv = lrs[i] * (xv - lmu[i] * p.ln_cs[n]); xg = lrs[i] * (xg - lmu[i] * p.ln_cs[Nout + n]); }
            x = (xv + p.bias[n]) * gelu_erf_f(xg + p.bias[Nout + n]);
          } else {
            x = acc[i][e >> 2][e & 3] * p.alpha;
            if (p.ln_stats) x = lrs[i] * (x - lmu[i] * p.ln_cs[n]);
            if (p.bias_mode == 1) x += p.bias[n]; else if (p.bias_mode == 2) x += p.bias[m];
            if (p.rowadd) x += p.rowadd[(long long)img * p.rowadd_ld + n];
          }
          if (p.residual) x += bf2f(p.residual[bz * p.sR + (long long)m * p.ldr + n]);
          if (p.act == 1) x = silu_f(x); else if (p.act == 2) x = x / (1.0f + __expf(-1.702f * x)); else if (p.act == 3) x = gelu_erf_f(x);
          if (p.rowstat_out) { const float r = bf2f(f2bf(x)); rs[i] += r; rq[i] += r * r; }
          if (ctile) ctile[(i * 16 + px) * WTN + q * CA + e] = f2bf(x);
          if (p.out_f32) ((float*)p.out)[bz * p.sO + (long long)m * p.ldo + n] = x;
          else ((bf16_t*)p.out)[bz * p.sO + (long long)m * p.ldo + n] = f2bf(x);
        }
      }
    }
    if (p.colstat_out) {          // wave-uniform (kernel argument): per-(tile, channel) sum and sum of squares over the tile's rows
      constexpr int NT = WM * WN * 64;
      __syncthreads();                                 // tiles written
      float* cst = (float*)((bf16_t*)smem + WM * WN * WTM * WTN);   // [WM][BN][2]
      for (int col = lane; col < WTN; col += 64) {
        float S = 0.f, Q = 0.f;
        for (int r = 0; r < WTM; ++r) { const float x = bf2f(ctile[r * WTN + col]); S += x; Q += x * x; }
        cst[(wm * BN + wn * WTN + col) * 2] = S; cst[(wm * BN + wn * WTN + col) * 2 + 1] = Q;
      }
      __syncthreads();
      const int tm_ = m0 / BM;
      for (int t = threadIdx.x; t < BN; t += NT) {     // fixed order over the tile's row waves: reproducible
        const int n = n0 + t;
        if (n < p.N) {
          float S = 0.f, Q = 0.f;
#pragma unroll
          for (int w = 0; w < WM; ++w) { S += cst[(w * BN + t) * 2]; Q += cst[(w * BN + t) * 2 + 1]; }
          float* op = p.colstat_out + ((long long)tm_ * p.N + n) * 2;
          op[0] = S; op[1] = Q;
        }
      }
    }
    if (!FAST_ONLY && p.rowstat_out) {          // wave-uniform (kernel argument)
      constexpr int NT = WM * WN * 64;
#pragma unroll
      for (int i = 0; i < MI; ++i) {                   // the 4 lanes (q) that share a pixel row
        rs[i] += __shfl_xor(rs[i], 16); rs[i] += __shfl_xor(rs[i], 32);
        rq[i] += __shfl_xor(rq[i], 16); rq[i] += __shfl_xor(rq[i], 32);
      }
      __syncthreads();                                 // every wave has left the K loop: the LDS ring is free
      float* stg = (float*)smem;                       // [WN][BM][2]
      if (q == 0) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int row = wm * WTM + i * 16 + px;
          stg[(wn * BM + row) * 2] = rs[i]; stg[(wn * BM + row) * 2 + 1] = rq[i];
        }
      }
      __syncthreads();
      for (int r = threadIdx.x; r < BM; r += NT) {     // fixed order over the waves of the tile: reproducible
        const int m = m0 + r;
        if (m < p.M) {
          float S = 0.f, Q = 0.f;
#pragma unroll
          for (int w = 0; w < WN; ++w) { S += stg[(w * BM + r) * 2]; Q += stg[(w * BM + r) * 2 + 1]; }
          float* op = p.rowstat_out + ((long long)m * p.rowstat_slots + tn) * 2;
          op[0] = S; op[1] = Q;
        }
      }
    }
  }
}
