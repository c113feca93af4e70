// Shared igemm epilogue: accumulators -> LDS (fp32) -> whole 16-B row chunks with bias / time-embedding
// row add / residual (prefetched) / GEGLU / SiLU fused, one rounding to bf16; split-K writes fp32 slabs.
#pragma once
#include "kernels.h"

template <int BM, int BN, int WM, int WN, int GEGLU, int SPLITK>
AGD_DEV void igemm_epilogue(const IgemmP& p, f32x4 (&acc)[BM / WM / 16][BN / WN / 16], char* smem, int tid, int lane,
                            int wm, int wn, int m0, int n0, int tn, int bz) {
  constexpr int NT = WM * WN * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MI = WTM / 16, NI = WTN / 16;
  const int HWo = p.Hout * p.Wout;
  const float inv_hwo_e = 1.0f / (float)HWo;        // rowadd image index (M < 2^24 is checked by the launcher)
  // ---- epilogue: acc -> LDS fp32 [BM][BN] -> coalesced 16-B row chunks -----------------------
  // Residual chunks are prefetched into registers before the LDS round trip so their HBM latency
  // overlaps the staging; all trip counts are compile-time.
  constexpr int OW = GEGLU ? BN / 2 : BN;           // output columns produced by this tile
  constexpr int CPR = OW / 8;                       // 8-column chunks per output row
  constexpr int EP_IT = (BM * CPR + NT - 1) / NT;
  const int Nout = GEGLU ? p.N / 2 : p.N;
  const int no0 = GEGLU ? tn * (BN / 2) : n0;

  auto stage_acc = [&]() {
    __syncthreads();
    float* stg_ = (float*)smem;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          stg_[(wm * WTM + i * 16 + (lane >> 4) * 4 + r) * BN + wn * WTN + j * 16 + (lane & 15)] = acc[i][j][r] * p.alpha;
    __syncthreads();
  };
  const float* stg = (const float*)smem;

  if constexpr (SPLITK) {
    stage_acc();
    float* part = p.splitk_ws + ((long long)blockIdx.z * p.M) * p.N;
#pragma unroll
    for (int it = 0; it < EP_IT; ++it) {
      const int idx = tid + it * NT;
      const int r = idx / CPR, cc = idx - r * CPR;
      const int m = m0 + r, no = no0 + cc * 8;
      const int nvalid = (Nout - no) < 8 ? (Nout - no) : 8;
      if (idx < BM * CPR && m < p.M && nvalid > 0) {
        const float* sp = stg + r * BN + cc * 8;
        float* op = part + (long long)m * p.N + no;
        if (nvalid == 8) { *(f32x4*)op = *(const f32x4*)sp; *(f32x4*)(op + 4) = *(const f32x4*)(sp + 4); }
        else for (int e = 0; e < nvalid; ++e) op[e] = sp[e];
      }
    }
    return;
  } else {
    const bool vec_all = ((p.ldo & 7) == 0) && (!p.residual || (p.ldr & 7) == 0);
    s16x8 rres[EP_IT];
    if (p.residual && vec_all) {
#pragma unroll
      for (int it = 0; it < EP_IT; ++it) {
        const int idx = tid + it * NT;
        const int r = idx / CPR, cc = idx - r * CPR;
        const int m = m0 + r, no = no0 + cc * 8;
        if (idx < BM * CPR && m < p.M && no + 8 <= Nout) rres[it] = *(const s16x8*)(p.residual + bz * p.sR + (long long)m * p.ldr + no);
      }
    }
    // when the thread count is a multiple of the chunks per row, each thread keeps ONE column chunk for all its
    // rows: per-column epilogue operands (bias vectors) are loaded once, before the staging barriers
    constexpr bool FIXED_CC = (NT % CPR) == 0;
    float hb[8], hg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { hb[e] = 0.f; hg[e] = 0.f; }
    if constexpr (FIXED_CC) {
      const int no_ = no0 + (tid % CPR) * 8;
      if (p.bias && p.bias_mode != 2 && no_ + 8 <= Nout) {
        *(f32x4*)&hb[0] = *(const f32x4*)(p.bias + no_); *(f32x4*)&hb[4] = *(const f32x4*)(p.bias + no_ + 4);
        if constexpr (GEGLU) { *(f32x4*)&hg[0] = *(const f32x4*)(p.bias + Nout + no_); *(f32x4*)&hg[4] = *(const f32x4*)(p.bias + Nout + no_ + 4); }
      }
    }
    stage_acc();
#pragma unroll
    for (int it = 0; it < EP_IT; ++it) {
      const int idx = tid + it * NT;
      const int r = idx / CPR, cc = idx - r * CPR;
      const int m = m0 + r, no = no0 + cc * 8;
      const int nvalid = (Nout - no) < 8 ? (Nout - no) : 8;
      if (idx >= BM * CPR || m >= p.M || nvalid <= 0) continue;
      const bool vec_ok = vec_all && nvalid == 8;
      float v[8];
      const float* sp = stg + r * BN + cc * 8;
      *(f32x4*)&v[0] = *(const f32x4*)sp;
      *(f32x4*)&v[4] = *(const f32x4*)(sp + 4);
      if constexpr (GEGLU) {
        float g[8];
        *(f32x4*)&g[0] = *(const f32x4*)(sp + BN / 2);
        *(f32x4*)&g[4] = *(const f32x4*)(sp + BN / 2 + 4);
        if constexpr (!FIXED_CC) {
          *(f32x4*)&hb[0] = *(const f32x4*)(p.bias + no); *(f32x4*)&hb[4] = *(const f32x4*)(p.bias + no + 4);
          *(f32x4*)&hg[0] = *(const f32x4*)(p.bias + Nout + no); *(f32x4*)&hg[4] = *(const f32x4*)(p.bias + Nout + no + 4);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (v[e] + hb[e]) * gelu_erf_f(g[e] + hg[e]);
      } else {
        if (p.bias_mode == 1) {
          if (nvalid == 8) {
            if constexpr (!FIXED_CC) { *(f32x4*)&hb[0] = *(const f32x4*)(p.bias + no); *(f32x4*)&hb[4] = *(const f32x4*)(p.bias + no + 4); }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += hb[e];
          } else {
            for (int e = 0; e < nvalid; ++e) v[e] += p.bias[no + e];
          }
        } else if (p.bias_mode == 2) {
          const float bm = p.bias[m];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bm;
        }
        if (p.rowadd) {
          const float* ra = p.rowadd + (long long)fast_udiv(m, HWo, inv_hwo_e) * p.rowadd_ld + no;
          if (nvalid == 8) {
            float rv[8];
            *(f32x4*)&rv[0] = *(const f32x4*)ra; *(f32x4*)&rv[4] = *(const f32x4*)(ra + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rv[e];
          } else {
            for (int e = 0; e < nvalid; ++e) v[e] += ra[e];
          }
        }
      }
      if (p.residual) {
        if (vec_ok) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bf2f((bf16_t)rres[it][e]);
        } else {
          const bf16_t* rp = p.residual + bz * p.sR + (long long)m * p.ldr + no;
          for (int e = 0; e < nvalid; ++e) v[e] += bf2f(rp[e]);
        }
      }
      if (p.act == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
      } else if (p.act == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] / (1.0f + __expf(-1.702f * v[e]));
      } else if (p.act == 3) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_erf_f(v[e]);
      }
      if (p.out_f32) {
        float* op = (float*)p.out + bz * p.sO + (long long)m * p.ldo + no;
        if (((p.ldo & 3) == 0) && nvalid == 8) {
          *(f32x4*)op = *(f32x4*)&v[0];
          *(f32x4*)(op + 4) = *(f32x4*)&v[4];
        } else {
          for (int e = 0; e < nvalid; ++e) op[e] = v[e];
        }
      } else {
        bf16_t* op = (bf16_t*)p.out + bz * p.sO + (long long)m * p.ldo + no;
        if (vec_ok) {
          u32x4 pk;
          pk[0] = pack_bf2(v[0], v[1]); pk[1] = pack_bf2(v[2], v[3]);
          pk[2] = pack_bf2(v[4], v[5]); pk[3] = pack_bf2(v[6], v[7]);
          *(u32x4*)op = pk;
        } else {
          for (int e = 0; e < nvalid; ++e) op[e] = f2bf(v[e]);
        }
      }
    }
  }
}
