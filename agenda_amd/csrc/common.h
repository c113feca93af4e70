// Common device/host helpers for the agenda_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

typedef unsigned short bf16_t;  // raw bf16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define AGD_DEV __device__ __forceinline__

AGD_DEV float bf2f(bf16_t b) { return __uint_as_float(((unsigned int)b) << 16); }
AGD_DEV bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
AGD_DEV unsigned int pack_bf2(float lo, float hi) {
  return (unsigned int)f2bf(lo) | ((unsigned int)f2bf(hi) << 16);
}
AGD_DEV float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the bf16 output step): 1 rcp + 1 exp
// instead of libm erff's ~40-instruction polynomial ladder -- the GEGLU epilogue evaluates it 32x per thread.
AGD_DEV float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896340736f);
  const float r = 1.0f - p * t * e;
  return x < 0.f ? -r : r;
}
// gelu(x) = x * Phi(x) with the same A&S 7.1.26 erf, rearranged to 13 VALU + 2 transcendental ops per element:
// q = 1 - Phi(|x|) = (0.5 * poly(t)) * t * exp(-x^2 / 2),  t = 1 / (1 + (p / sqrt 2) |x|);  gelu = max(x, 0) - |x| * q
// (x >= 0: x - x q = x Phi(x);  x < 0: -|x| q = x (1 - Phi(|x|)) = x Phi(x)).  The GEGLU epilogue runs it 32x per thread.
AGD_DEV float gelu_erf_f(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.2316418878f, ax, 1.0f));
  float p = fmaf(0.5307027145f, t, -0.7265760135f);
  p = fmaf(p, t, 0.7107068705f);
  p = fmaf(p, t, -0.142248368f);
  p = fmaf(p, t, 0.127414796f);
  const float e = __builtin_amdgcn_exp2f(ax * ax * -0.72134752044448170368f);
  const float q = p * t * e;
  return fmaxf(x, 0.f) - ax * q;
}

// value held by lane (l ^ 32) combined with own: one VALU v_permlane32_swap instead of a ds_bpermute round trip
AGD_DEV float xhalf_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
AGD_DEV float xhalf_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// global -> LDS direct (LDS-DMA), 16 B per lane; LDS destination = wave-uniform base + lane*16.
AGD_DEV void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// LDS-DMA through a buffer descriptor: `base` must be wave-uniform; `voff` is the per-lane byte offset
// (>= 0x7FFFFFF0 -> out of range -> the hardware writes zeros: conv padding / tile tails for free);
// `soff` is a wave-uniform (SGPR) byte offset.  LDS destination = wave-uniform base + lane*16.
AGD_DEV void bufdma16(const void* base, void* lds_wave_base, unsigned voff, unsigned soff, unsigned nrec = 0x7FFFFFF0u) {
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, nrec, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

// x / d for 0 <= x < 2^24 and d >= 1 without the ~40-instruction integer-division sequence: float estimate + one
// correction step each way (the estimate is off by at most 1).  Used for the im2col row -> (image, y, x) split.
AGD_DEV int fast_udiv(int x, int d, float inv_d) {
  int q = (int)((float)x * inv_d);
  int r = x - q * d;
  if (r < 0) { --q; r += d; }
  if (r >= d) ++q;
  return q;
}

// XCD-aware bijective block remap: consecutive logical ids land on the same XCD (8 XCDs,
// round-robin dispatch), so tiles that share an operand panel share an L2.  Speed only.
AGD_DEV int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Tile (tm, tn) of a logical block id (xcd_remap: the ids [c q, (c + 1) q), q = tiles / 8, share one XCD's L2).  xb_m > 0: the launcher cut the tile grid into eight
// xb_m x xb_n blocks, one per XCD, chosen so that the XCD's working set (xb_m A panels + xb_n W panels) is smallest -- tools/ubench/l2_stride.hip: a CU takes in
// 113 GB/s while its XCD's working set fits the 4 MB L2 and 79 GB/s once it does not; a row of tiles (the A-major / W-major walks) is the WORST shape for that.
AGD_DEV void tile_of(int bid, int tiles_m, int tiles_n, int wmajor, int xb_m, int xb_n, int& tm, int& tn) {
  if (xb_m > 0) {
    const int q = xb_m * xb_n, c = bid / q, j = bid - c * q;
    const int nbn = tiles_n / xb_n, cm = c / nbn, cn = c - cm * nbn;
    const int jm = j / xb_n;
    tm = cm * xb_m + jm; tn = cn * xb_n + (j - jm * xb_n);
  } else if (wmajor) { tm = bid % tiles_m; tn = bid / tiles_m; }
  else { tn = bid % tiles_n; tm = bid / tiles_n; }
}

#define HIP_CHECK_RET(expr)                                                      \
  do {                                                                           \
    hipError_t _e = (expr);                                                      \
    if (_e != hipSuccess) { agd_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); return -1; } \
  } while (0)

extern "C" void agd_set_error(const char* fmt, ...);
