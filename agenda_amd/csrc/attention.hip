// Flash attention for gfx950 with fused cross-attention probability recording (DAAM / hook.py).
//
// Replaces: the attention-processor body of reference data_generation/hook.py:104-115
// (head_to_batch_dim, get_attention_scores = softmax(scale*QK^T), bmm(P,V), batch_to_head_dim)
// and, fused into the same kernel, the recorder side channel: daam's per-(layer,head) time-summed
// maps or hook.py:28-56 `_unravel_attn` head-mean maps (row H2).
// The [B*H, N, 77] probability tensor of hook.py:108 is never materialised in HBM.
//
// Formulation (CDNA4, 64-wide waves, v_mfma_f32_32x32x16_bf16):
//   S^T[key][query] = K . Q^T   (A = K rows from LDS, B = Q fragments held in registers)
//     -> accumulator has the QUERY on the lane and the keys in registers, so the softmax row
//        max/sum are in-lane reductions + one cross-half shuffle;
//   O^T[d][query]  += V^T . P^T (A = V^T via ds_read_b64_tr_b16 from a row-major V tile,
//                                B = the S^T accumulator registers converted to bf16 in place);
//   recorded probabilities are stored token-major [T][h*w]: for a fixed register the 32 lanes
//   of a half-wave hold 32 consecutive pixels -> 128-B coalesced read-modify-write.
#include "kernels.h"
#include <type_traits>

#ifdef AGD_EXPERIMENTS
int g_attn_qb = 1;   // queries per wave / 32 for the long self-attention shapes (tools/ only)
extern "C" __attribute__((visibility("default"))) void agd_set_attn_qb(int v) { g_attn_qb = v; }
#else
constexpr int g_attn_qb = 1;
#endif

template <int D> struct AttnCfg {
  static constexpr int KSTEPS = (D + 15) / 16;           // QK^T k-steps (d padded to 16)
  static constexpr int DBLK = (D + 31) / 32;             // PV output d-blocks (d padded to 32)
  static constexpr int KCH = KSTEPS * 2;                 // 16-B chunks per K row incl. zero pad
  static constexpr int KPITCH = ((KCH | 1)) * 16;        // odd #chunks -> conflict-free b128 row reads
  static constexpr int VPITCH = ((DBLK | 1)) * 64;       // odd multiple of 64 B -> conflict-free tr reads
  static constexpr int CH = D / 8;                       // real 16-B chunks per row
  // A spare (padded) V column exists when D is not a multiple of 32: column D is set to 1.0 so the
  // PV MFMA also produces the softmax row sum (row D of O^T) -- no VALU adds for the denominator.
  static constexpr bool ONES = (DBLK * 32 > D);
  // A spare (padded) QK^T k-column exists when D is not a multiple of 16: K's pad column D is set to 1.0 and Q's
  // to -m (the running reference max), so the MFMA itself subtracts the max -- no VALU op and no C-operand tuple.
  static constexpr bool QPAD = (KSTEPS * 16 > D);
};

#define DEFER_THR 8.0f   // log2 units: rescale O only when a row max grows by more than 2^8

// AMASK: additive attention mask [B][Nk] (diffusers `attention_mask` after prepare_attention_mask, hook.py:92,108:
// broadcast over heads and queries) -- a separate instantiation so the production kernels carry none of it.
// RECORD: 0 = plain flash attention; 1 = single key tile, probabilities added into per-(image, head) rows (DAAM) or written per
// head (hook.py mode); 2 = single key tile, the block walks ALL heads of its (batch row, query tile) and adds the head SUM of the
// probabilities into per-image rows once -- the DAAM layers at latent resolution, where the aggregation is linear in the heads
// (bicubic to the same size is the identity and clamp(min=0) cannot fire), so 1/8 of the accumulator traffic.
template <int D, int KB, int QB, int RECORD, int AMASK = 0>
#if defined(AGD_EXPERIMENTS) && defined(EXP_ATTN_LB)      // tools/: occupancy experiments on the plain flash kernels
#define ATTN_LB(def) ((!RECORD && !AMASK) ? EXP_ATTN_LB : (def))
#else
#define ATTN_LB(def) (def)
#endif
__global__ __launch_bounds__(256, ATTN_LB((AMASK ? 1 : (QB == 1 && D <= 64 && !RECORD) ? 3 : (QB == 1 && D <= 80) ? 2 : 1))) void attn_kernel(const AttnP p) {
  using C = AttnCfg<D>;
  constexpr int KEYS = KB * 32;
  constexpr int KSTEPS = C::KSTEPS, DBLK = C::DBLK, KPITCH = C::KPITCH, VPITCH = C::VPITCH, CH = C::CH;
  constexpr bool ONES = C::ONES && !RECORD;
  constexpr bool QPAD = C::QPAD && !RECORD;              // -m through the K-dim pad column
  constexpr bool CNEG = !QPAD && !RECORD;                // -m through the MFMA C operand
  constexpr int NCHUNK = KEYS * CH;                      // 16-B chunks per K (or V) tile
  constexpr int LD_IT = (NCHUNK + 255) / 256;
  constexpr int STAGE = KEYS * (KPITCH + VPITCH);
  constexpr int NSTAGE = (RECORD == 1) ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int c = lane & 31, hh = lane >> 5;
  // XCD-aware 1-D grid: blocks are dealt round-robin over the 8 XCDs, so give every (batch, head) pair --
  // whose query tiles all stream the same K/V -- to ONE XCD (its L2 then serves K/V after the first tile).
  // RECORD 2: one block per (batch row, group of rec_hpb heads, query tile), the group's heads inside the block
  const int HPB = (RECORD == 2) ? p.H / p.rec_hpb : p.H;
  const int nqt = p.nqt, nbh = p.B * HPB;
  int bh, qt;
  {
    const int id = blockIdx.x, x = id & 7, j = id >> 3;
    const int per = (nbh + 7) >> 3;                    // (b,h) pairs per XCD group
    const int lb = j / nqt;
    qt = j - lb * nqt;
    bh = x * per + lb;                                 // may exceed nbh for the last groups: exit
  }
  if (bh >= nbh) return;
  const int hgrp = bh % HPB;                             // RECORD 2: head group; else the head
  int head = (RECORD == 2) ? hgrp * p.rec_hpb : hgrp;
  const int hend = (RECORD == 2) ? head + p.rec_hpb : head + 1;
  const int b = bh / HPB;
  const int q0 = qt * (128 * QB) + wid * (32 * QB);
  const bf16_t* kp = p.k + b * p.sk + head * D;
  const bf16_t* vp = p.v + b * p.sv + head * D;

  // pad chunks (never overwritten by tile loads): K pad = 0; V pad = 0 except column D = 1.0 (ONES)
#pragma unroll
  for (int sidx = 0; sidx < NSTAGE; ++sidx) {
    char* sK = smem + sidx * STAGE;
    char* sV = sK + KEYS * KPITCH;
    if constexpr (C::KCH > CH) {
      for (int i = tid; i < KEYS * (C::KCH - CH); i += 256) {
        const int r = i / (C::KCH - CH), cc = CH + i % (C::KCH - CH);
        *(u32x4*)(sK + r * KPITCH + cc * 16) = u32x4{(QPAD && cc == CH) ? 0x3F80u : 0u, 0, 0, 0};
      }
    }
    if constexpr (DBLK * 4 > CH) {
      for (int i = tid; i < KEYS * (DBLK * 4 - CH); i += 256) {
        const int r = i / (DBLK * 4 - CH), cc = CH + i % (DBLK * 4 - CH);
        *(u32x4*)(sV + r * VPITCH + cc * 16) = u32x4{(ONES && cc == CH) ? 0x3F80u : 0u, 0, 0, 0};
      }
    }
  }

  const float qscale = p.scale * 1.44269504088896340736f;   // exp(x) = exp2(x*log2e)
  // Q fragments: lane (c, hh) holds Q[q0 + 32*qb + c][16s + 8hh .. +8]
  bf16x8 qf[QB][KSTEPS];
  u32x4 qnext[RECORD == 2 ? KSTEPS : 1];                 // RECORD 2: the next head's Q fragments, fetched under this head's work
  auto load_q = [&]() {
    const bf16_t* qp = p.q + b * p.sq + head * D;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const int qrow = q0 + qb * 32 + c;
      const bool qok = qrow < p.Nq;
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        const int d0 = 16 * s + 8 * hh;
        u32x4 v = u32x4{0, 0, 0, 0};
        if (qok && d0 < D) v = *(const u32x4*)(qp + (long long)qrow * p.ldq + d0);
        qf[qb][s] = __builtin_bit_cast(bf16x8, v);
        if constexpr (!RECORD) {
          // fold scale*log2(e) into Q once (fp32 multiply, RNE back to bf16): the QK^T accumulator then holds
          // log2-domain logits and the per-score v_fma in the softmax disappears (that kernel is VALU-bound).
#pragma unroll
          for (int j = 0; j < 8; ++j) qf[qb][s][j] = (__bf16)((float)qf[qb][s][j] * qscale);
        }
      }
    }
  };
  auto prefetch_q = [&](int h) {                          // RECORD 2 (QB = 1)
    const bf16_t* qp = p.q + b * p.sq + h * D;
    const int qrow = q0 + c;
#pragma unroll
    for (int s = 0; s < (RECORD == 2 ? KSTEPS : 1); ++s) {
      const int d0 = 16 * s + 8 * hh;
      qnext[s] = u32x4{0, 0, 0, 0};
      if (qrow < p.Nq && d0 < D) qnext[s] = *(const u32x4*)(qp + (long long)qrow * p.ldq + d0);
    }
  };
  load_q();

  // K/V tile prefetch through buffer descriptors: per-lane 32-bit offset fixed for the kernel, the tile
  // advance in an SGPR soffset; rows >= Nk fall outside num_records and read as zeros.
  u32x4 kreg[LD_IT], vreg[LD_IT];
  unsigned kvoff[LD_IT], vvoff[LD_IT];
#pragma unroll
  for (int it = 0; it < LD_IT; ++it) {
    const int idx = tid + it * 256;
    const int r = idx / CH, cc = idx - r * CH;
    kvoff[it] = (idx < NCHUNK) ? (unsigned)((r * p.ldk + cc * 8) * 2) : 0x80000000u;
    vvoff[it] = (idx < NCHUNK) ? (unsigned)((r * p.ldv + cc * 8) * 2) : 0x80000000u;
  }
  const unsigned k_bytes = (unsigned)(((long long)(p.Nk - 1) * p.ldk + D) * 2), v_bytes = (unsigned)(((long long)(p.Nk - 1) * p.ldv + D) * 2);
  const auto krs = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, k_bytes, 0x00020000);
  const auto vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vp, 0, v_bytes, 0x00020000);
  auto gload = [&](int tile) {
    const unsigned ks_ = __builtin_amdgcn_readfirstlane((unsigned)(tile * KEYS * p.ldk * 2));
    const unsigned vs_ = __builtin_amdgcn_readfirstlane((unsigned)(tile * KEYS * p.ldv * 2));
#pragma unroll
    for (int it = 0; it < LD_IT; ++it) {
      kreg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, kvoff[it], ks_, 0));
      vreg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, vvoff[it], vs_, 0));
    }
  };
  // RECORD 2: the K/V slice of head h (one tile: Nk <= KEYS); same per-lane offsets, the head shifts the descriptor base
  auto gload_head = [&](int h) {
    const auto krh = __builtin_amdgcn_make_buffer_rsrc((void*)(p.k + b * p.sk + h * D), 0, k_bytes, 0x00020000);
    const auto vrh = __builtin_amdgcn_make_buffer_rsrc((void*)(p.v + b * p.sv + h * D), 0, v_bytes, 0x00020000);
#pragma unroll
    for (int it = 0; it < LD_IT; ++it) {
      kreg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(krh, kvoff[it], 0, 0));
      vreg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(vrh, vvoff[it], 0, 0));
    }
  };
  auto lstore = [&](int stage) {
    char* sK = smem + stage * STAGE;
    char* sV = sK + KEYS * KPITCH;
#pragma unroll
    for (int it = 0; it < LD_IT; ++it) {
      const int idx = tid + it * 256;
      const int r = idx / CH, cc = idx - r * CH;
      if (idx < NCHUNK) {
        *(u32x4*)(sK + r * KPITCH + cc * 16) = kreg[it];
        *(u32x4*)(sV + r * VPITCH + cc * 16) = vreg[it];
      }
    }
  };

  f32x16 oacc[QB][DBLK];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int i = 0; i < DBLK; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) oacc[qb][i][j] = 0.f;
  float m_run[QB], l_run[QB];
  // non-RECORD: -m_run replicated over a 16-register tuple = the C operand of the first QK^T MFMA, so the
  // accumulator comes out as (logit - m) and exp2 applies to it directly.  Rewritten only on a (rare) rescale.
  f32x16 negm[CNEG ? QB : 1];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = RECORD ? -INFINITY : 0.f; l_run[qb] = 0.f;
    if constexpr (CNEG) {
#pragma unroll
      for (int j = 0; j < 16; ++j) negm[qb][j] = 0.f;
    }
  }
  const float sc = qscale;
  f32x16 pacc[RECORD == 2 ? KB : 1];                     // RECORD 2: head sum of this lane's probabilities
  const bool recb = RECORD == 2 && p.record_mode == 3 && b >= p.rec_b0;
  if constexpr (RECORD == 2) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int j = 0; j < 16; ++j) pacc[kb][j] = 0.f;
  }

  const int ntiles = RECORD ? 1 : (p.Nk + KEYS - 1) / KEYS;   // RECORD: host guarantees Nk <= KEYS
  const bool ragged = (p.Nk % KEYS) != 0 || p.causal || AMASK;   // causal / additive mask: every tile takes the masked path
  gload(0);
  lstore(0);
  __syncthreads();

  // V^T fragment addressing for ds_read_b64_tr_b16: lane supplies row (key) q, 4 columns at 4*pp
  const int gi = lane & 15;
  const int tr_row = (gi >> 2) + 4 * hh;                 // + 16*s (+8 second half) + 32*kb
  const int tr_col = ((lane >> 4) & 1) * 16 + (gi & 3) * 4;  // + 32*db   (elements)

  auto tile_body = [&](int t, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const int cur = (RECORD == 1) ? 0 : (RECORD == 2) ? ((t - (hend - p.rec_hpb)) & 1) : (t & 1);   // RECORD 2: t is the head
    const char* sK = smem + cur * STAGE;
    const char* sV = sK + KEYS * KPITCH;
    if constexpr (RECORD == 2) { if (t + 1 < hend) { gload_head(t + 1); prefetch_q(t + 1); } }
    else if (t + 1 < ntiles) gload(t + 1);
    // ---- S^T = K . Q^T ----
    f32x16 sacc[QB][KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        const bf16x8 a = *(const bf16x8*)(sK + (kb * 32 + c) * KPITCH + (2 * s + hh) * 16);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          if (s == 0) { const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        sacc[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[qb][s], CNEG ? negm[qb] : z, 0, 0, 0); }
          else sacc[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[qb][s], sacc[qb][kb], 0, 0, 0);
        }
      }
    }
    // ---- online softmax (query on the lane); keys beyond Nk masked on the last tile only ----
    if constexpr (MASK) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = (RECORD ? 0 : t * KEYS) + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            if (key >= p.Nk || (p.causal && key > q0 + qb * 32 + c)) sacc[qb][kb][i] = -INFINITY;
            else if constexpr (AMASK) {
              // RECORD keeps raw logits (scaled below): pre-divide so that (s + a/scale)*scale = s*scale + a
              const float a = p.mask[(long long)b * p.Nk + key];
              sacc[qb][kb][i] += RECORD ? a / p.scale : a * 1.44269504088896340736f;
            }
          }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float mx = sacc[qb][0][0];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[qb][kb][i]);
      if constexpr (RECORD) {
        mx = xhalf_max(mx) * sc;
        m_run[qb] = mx;                                   // single tile: no running state to rescale
        if constexpr (RECORD == 2) l_run[qb] = 0.f;       // per head
        const float nm = -mx;
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[qb][kb][i], sc, nm));
            sacc[qb][kb][i] = pv;
            rs += pv;
          }
        l_run[qb] += rs;
      } else {
        // sacc already holds (logit - m_run) in log2 units.  Deferred rescale: only when some score of this
        // tile exceeds the reference by 2^DEFER_THR (or on the first tile, which fixes the reference).
        if (t == 0 || !__all(mx <= DEFER_THR)) {
          const float mxf = xhalf_max(mx);
          float delta = (t == 0) ? mxf : fmaxf(mxf, 0.f);
          if (!(delta > -INFINITY)) delta = 0.f;
          if constexpr (QPAD) {                             // the reference lives in a bf16 Q slot: round it
            const float mb = (float)(__bf16)(m_run[qb] + delta);
            delta = mb - m_run[qb];
            constexpr int pc = D - 16 * (KSTEPS - 1);       // pad column within the last k-step
            if (hh == pc / 8) qf[qb][KSTEPS - 1][pc % 8] = (__bf16)(-mb);
          }
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          m_run[qb] += delta;
          l_run[qb] *= alpha;
#pragma unroll
          for (int db = 0; db < DBLK; ++db)
#pragma unroll
            for (int j = 0; j < 16; ++j) oacc[qb][db][j] *= alpha;
#pragma unroll
          for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[qb][kb][i] -= delta;
          if constexpr (CNEG) {
            const float nm = -m_run[qb];
#pragma unroll
            for (int j = 0; j < 16; ++j) negm[qb][j] = nm;
          }
        }
        if constexpr (CNEG) asm volatile("" : "+v"(negm[qb]));   // keep the tuple resident (no per-tile re-splat)
        if constexpr (ONES) {
#pragma unroll
          for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[qb][kb][i] = __builtin_amdgcn_exp2f(sacc[qb][kb][i]);
        } else {
          float rs = 0.f;
#pragma unroll
          for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const float pv = __builtin_amdgcn_exp2f(sacc[qb][kb][i]);
              sacc[qb][kb][i] = pv;
              rs += pv;
            }
          l_run[qb] += rs;
        }
      }
    }

    if constexpr (RECORD == 2) {
      if (recb) {
        const float inv = 1.0f / xhalf_sum(l_run[0]);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i) pacc[kb][i] += sacc[0][kb][i] * inv;
      }
    } else if constexpr (RECORD) {
      // single-tile case (host guarantees Nk <= KEYS): probabilities are final here.
      // All old values are loaded first, then added, then stored: the RMW chains overlap.
      if (p.record_mode != 0 && b >= p.rec_b0 && (q0 + c) < p.Nq) {
        const float lt = xhalf_sum(l_run[0]);
        const float inv = 1.0f / lt;
        const int img = b - p.rec_b0;
        if (p.record_mode == 1) {
          // buffer RMW: wave-uniform base (this image/head slice), one 32-bit per-lane offset; the
          // hardware range check (num_records = rec_T*Nq*4 B) drops token rows >= rec_T.
          float* base = p.rec + img * p.rec_img_stride + head * p.rec_head_stride;
          const unsigned nbytes = (unsigned)p.rec_T * (unsigned)p.Nq * 4u;
          const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, nbytes, 0x00020000);
          const unsigned rowb = (unsigned)p.Nq * 4u;
          const unsigned voff0 = (unsigned)(q0 + c) * 4u + (unsigned)(4 * hh) * rowb;
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) {
            float old[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const unsigned off = voff0 + (unsigned)(kb * 32 + (i & 3) + 8 * (i >> 2)) * rowb;
              old[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const unsigned off = voff0 + (unsigned)(kb * 32 + (i & 3) + 8 * (i >> 2)) * rowb;
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, old[i] + sacc[0][kb][i] * inv), rsrc, off, 0, 0);
            }
          }
        }
      }
    }

    // ---- O^T += V^T . P^T ----
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 pf[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[qb][j] = (__bf16)sacc[qb][kb][8 * s + j];
        const char* vb = sV + (kb * 32 + 16 * s + tr_row) * VPITCH + tr_col * 2;
#pragma unroll
        for (int db = 0; db < DBLK; ++db) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + db * 64));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + 8 * VPITCH + db * 64));
          const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
          const u32x4 a4 = u32x4{lo2[0], lo2[1], hi2[0], hi2[1]};
#pragma unroll
          for (int qb = 0; qb < QB; ++qb)
            oacc[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a4), pf[qb], oacc[qb][db], 0, 0, 0);
        }
      }
    if ((RECORD == 2) ? (t + 1 < hend) : (t + 1 < ntiles)) {
      lstore(cur ^ 1);      // the other stage was last read in iteration t-1 (all waves passed its barrier)
      __syncthreads();
    }
  };
  // ---- finalize + store O[query][d] of the current head ----
  auto store_o = [&]() {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float lt;
      if constexpr (ONES) {
        constexpr int ld = D % 32;
        const float lv = oacc[qb][D / 32][4 * (ld / 8)];     // row D of O^T lives in lanes 0..31 (hh = 0)
        lt = __shfl(lv, c);
      } else {
        lt = xhalf_sum(l_run[qb]);
      }
      const float inv = 1.0f / lt;
      const int qrow = q0 + qb * 32 + c;
      if (qrow < p.Nq) {
        bf16_t* op = p.o + b * p.so + (long long)qrow * p.ldo + head * D;
#pragma unroll
        for (int db = 0; db < DBLK; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int d0 = db * 32 + 8 * g + 4 * hh;
            if (d0 < D) {
              u32x2 pk;
              pk[0] = pack_bf2(oacc[qb][db][4 * g + 0] * inv, oacc[qb][db][4 * g + 1] * inv);
              pk[1] = pack_bf2(oacc[qb][db][4 * g + 2] * inv, oacc[qb][db][4 * g + 3] * inv);
              *(u32x2*)(op + d0) = pk;
            }
          }
      }
    }
  };
  {
    using T_ = std::integral_constant<bool, true>; using F_ = std::integral_constant<bool, false>;
    if constexpr (RECORD == 2) {
      const int h0 = head;
      for (; head < hend; ++head) {                    // K/V of the first head are staged; tile_body prefetches head + 1
        if (head > h0) {
#pragma unroll
          for (int s = 0; s < KSTEPS; ++s) qf[0][s] = __builtin_bit_cast(bf16x8, qnext[s]);
#pragma unroll
          for (int db = 0; db < DBLK; ++db)
#pragma unroll
            for (int j = 0; j < 16; ++j) oacc[0][db][j] = 0.f;
        }
        tile_body(head, T_{});
        store_o();
      }
      // one read-modify-write of the per-image rows with the head sum: all loads first, then the stores
      if (recb && (q0 + c) < p.Nq) {
        float* base = p.rec + (long long)(b - p.rec_b0) * p.rec_img_stride + hgrp * p.rec_head_stride;
        const unsigned nbytes = (unsigned)p.rec_T * (unsigned)p.Nq * 4u;       // token rows >= rec_T are dropped by the range check
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, nbytes, 0x00020000);
        const unsigned rowb = (unsigned)p.Nq * 4u;
        const unsigned voff0 = (unsigned)(q0 + c) * 4u + (unsigned)(4 * hh) * rowb;
        float old[KB][16];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            old[kb][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff0 + (unsigned)(kb * 32 + (i & 3) + 8 * (i >> 2)) * rowb, 0, 0));
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, old[kb][i] + pacc[kb][i]), rsrc,
                                                  voff0 + (unsigned)(kb * 32 + (i & 3) + 8 * (i >> 2)) * rowb, 0, 0);
      }
      return;
    } else if constexpr (RECORD) {
      tile_body(0, T_{});                              // single (always key-masked) tile
    } else {
      if (p.causal || AMASK) { for (int t = 0; t < ntiles - 1; ++t) tile_body(t, T_{}); }
      else { for (int t = 0; t < ntiles - 1; ++t) tile_body(t, F_{}); }
      if (ragged) tile_body(ntiles - 1, T_{}); else tile_body(ntiles - 1, F_{});
    }
  }
  store_o();
}


template <int D, int KB, int QB, int RECORD, int AMASK = 0>
static int launch_attn_t(const AttnP& p, hipStream_t st) {
  using C = AttnCfg<D>;
  constexpr int lds = (RECORD == 1 ? 1 : 2) * KB * 32 * (C::KPITCH + C::VPITCH);
  AttnP pp = p;
  pp.nqt = (p.Nq + 128 * QB - 1) / (128 * QB);
  const int per = (p.B * (RECORD == 2 ? p.H / p.rec_hpb : p.H) + 7) / 8;
  dim3 grid(8 * per * pp.nqt);
  auto kfn = attn_kernel<D, KB, QB, RECORD, AMASK>;
  if (lds > 65536) {                                   // per (instantiation, device) latch
    static std::atomic<bool> attr[AGD_MAX_DEVICES] = {};
    int dev = 0; HIP_CHECK_RET(hipGetDevice(&dev));
    if (dev < 0 || dev >= AGD_MAX_DEVICES) { agd_set_error("attention: device ordinal %d out of range", dev); return -1; }
    if (!attr[dev]) { HIP_CHECK_RET(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr[dev] = true; }
  }
  hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, pp);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

template <int D>
static int launch_attn_d(const AttnP& p, hipStream_t st) {
  if (p.record_mode != 0) {
    if (p.Nk > 96) { agd_set_error("attention: recording needs Nk <= 96 (got %d)", p.Nk); return -1; }
    if (p.record_mode == 3) {                    // head-summed rows: the block walks the heads
      if (p.mask) { agd_set_error("attention: head-summed recording does not take an attention mask"); return -1; }
      if (p.rec_hpb < 1 || p.H % p.rec_hpb) { agd_set_error("attention: rec_hpb %d does not divide %d heads", p.rec_hpb, p.H); return -1; }
      return launch_attn_t<D, 3, 1, 2>(p, st);
    }
    return p.mask ? launch_attn_t<D, 3, 1, 1, 1>(p, st) : launch_attn_t<D, 3, 1, 1>(p, st);
  }
  if (p.mask) {                                  // processor-seam calls with an attention_mask (never on the SD hot path)
    if (p.causal) { agd_set_error("attention: additive mask + causal unsupported"); return -1; }
    return (p.Nk <= 96 && p.Nk > 64) ? launch_attn_t<D, 3, 1, 0, 1>(p, st) : launch_attn_t<D, 2, 1, 0, 1>(p, st);
  }
  if (p.Nk <= 96 && p.Nk > 64) return launch_attn_t<D, 3, 1, 0>(p, st);
  if constexpr (D <= 80) { if (p.Nq >= 1024 && g_attn_qb == 2) return launch_attn_t<D, 2, 2, 0>(p, st); }
#if defined(AGD_EXPERIMENTS) && defined(EXP_ATTN_KB)
  if (p.Nk >= 1024) return launch_attn_t<D, EXP_ATTN_KB, 1, 0>(p, st);
#endif
  return launch_attn_t<D, 2, 1, 0>(p, st);
}

int launch_attention(const AttnP& p, hipStream_t st) {
  if ((p.ldq | p.ldk | p.ldv | p.ldo) & 3) { agd_set_error("attention: row pitches must be multiples of 4 elements"); return -1; }
  if ((p.ldq | p.ldk | p.ldv) & 7) { agd_set_error("attention: q/k/v pitches must be multiples of 8 elements"); return -1; }
  switch (p.D) {
    case 32: return launch_attn_d<32>(p, st);
    case 40: return launch_attn_d<40>(p, st);
    case 64: return launch_attn_d<64>(p, st);
    case 80: return launch_attn_d<80>(p, st);
    case 128: return launch_attn_d<128>(p, st);
    case 160: return launch_attn_d<160>(p, st);
    default: agd_set_error("attention: unsupported head dim %d", p.D); return -1;
  }
}
