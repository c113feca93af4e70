// Internal launcher interface between the C-ABI/runtime (model.hip) and the HIP kernels.
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------------------
// Implicit GEMM:  out[M][N] = epilogue( alpha * sum_k A[m][k] * W[n][k] )
//   A is gathered on the fly from one or two NHWC bf16 sources (channel concat), with optional
//   nearest-2x upsample of the input and stride/pad (3x3 or 1x1 taps).  k = tap*(C0+C1) + c.
//   Linear layers are the KS=1, stride=1 case with Hin*Win = tokens.
// ---------------------------------------------------------------------------------------
#define AGD_MAX_DEVICES 16
struct SplitKWs { float* p = nullptr; size_t cap = 0; };   // split-K fp32 partial slabs, owned by the caller (one per ctx)

struct IgemmP {
  // ResnetBlock2D.conv_shortcut folded into conv2's launch (row-halo kernel, unsplit): after the 3x3 groups the K loop walks the 1x1 shortcut's channel chunks over
  // its own sources (the block's raw input, possibly a skip concat) with the weight rows' trailing sc_C0 + sc_C1 columns; K = 9 (C0 + C1) + sc_C0 + sc_C1
  const bf16_t* sc0 = nullptr; const bf16_t* sc1 = nullptr; int sc_C0 = 0, sc_C1 = 0;
  // Upsample2D.conv (3x3 on the nearest-2x upsampled map) as four 2x2 phase convs on the un-upsampled map in ONE launch: ups4 = Cout; ksize = 2, N = 4 Cout, the weight
  // rows [phase a b][co] hold the taps that coincide pre-summed; a tile's phase = n0 / ups4 sets its padding (1 - a, 1 - b) and its output rows (2 i + a, 2 j + b)
  int ups4 = 0;
  const bf16_t* src0; const bf16_t* src1;
  int C0, C1;
  int Hin, Win, Hout, Wout;
  int ksize, stride, pad, up;       // ksize 1|3 ; up 1|2
  const bf16_t* W;                  // [N][ksize*ksize*(C0+C1)]
  const float* bias; int bias_mode; // 0 none, 1 per-n, 2 per-m
  const float* rowadd; int rowadd_ld;  // [image][N] added per output row's image (m / (Hout*Wout))
  const bf16_t* residual; int ldr;  // [M][Nout]
  void* out; int out_f32; int ldo;  // Nout = geglu ? N/2 : N
  int M, N, K;
  float alpha;
  int geglu;                        // 1: tile columns are [val half | gate half] -> val*gelu(gate)
  int act;                          // 0 none, 1 silu, 2 quick_gelu x*sigmoid(1.702x), 3 gelu(erf) on the output
  int batch;                        // grid.y
  long long sA0, sA1, sW, sO, sR;   // per-batch element strides
  const bf16_t* zero_page;          // >= 256 B of zeros
  SplitKWs* ws;                     // caller's split-K workspace (grown on demand); NULL: a per-device default
  float* splitk_ws;                 // set by the launcher: fp32 partial slabs [S][M][N]
  // LayerNorm folded into the GEMMs around it (model.hip transformer()):
  //  producer side: rowstat_out [M][rowstat_slots] float2 = per-row (sum, sum of squares) of the bf16-rounded outputs of each N tile
  //  consumer side: A rows are RAW (un-normalised), W = W.diag(gamma); out = rstd (acc - mean colsum[n]) + bias[n] with
  //                 mean / rstd from ln_stats [M][ln_slots] float2, colsum = ln_cs [N] (GEGLU: values then gates, like bias)
  int colstat_rows;                 // rows of one image (M tiles must not straddle images)
  float* colstat_out;               // GroupNorm statistics from the producer: [M tiles][N] float2 = per-channel (sum, sum of squares) of each M tile's bf16 outputs
  float* rowstat_out; int rowstat_slots;
  const float* ln_stats; int ln_slots; const float* ln_cs; float ln_invC, ln_eps;
  int* cfg_out;                     // host pointer: igemm_query() -- report {BM, BN, splits} instead of launching
  int w_per_image;                  // 1x1 launches only: image i (= m / (Hout*Wout)) multiplies with W + i * sW (GroupNorm folded into the weights, model.hip transformer());
                                    // M tiles must not straddle images
  int wmajor;                       // set by the launcher: 1 = consecutive tiles share the weight panel (W-major walk), else the A panel
  int warm;                         // caller: 1 = cold weights expected; the launcher keeps it only for W-major launches (in-kernel warm-up of the XCD's W slice)
  int halo;                         // caller: 1 = 3x3 stride-1 launches may take the row-halo kernel (igemm_halo.h)
  int p8;                           // caller: 1 = launches with enough 256-row tiles take the 8-wave / 8-phase kernel (igemm8p.h); 2 / 3 force its 256- / 160-wide tile, 4 = any legal tile (benches, tests)
  // split-K launches whose output a GroupNorm(+SiLU) reads next (resnet conv1 -> norm2): the slab-sum pass normalises as well -- one
  // workgroup per (image, group) sums the slabs, applies the epilogue, takes the group's statistics from the bf16-rounded values it
  // holds in registers and writes gn_y = silu?(gn(out)); `out` itself is written only with gn_keep_out.  The launcher sets *gn_fused
  // (host) to 1 when it took this form; otherwise the caller runs the GroupNorm kernels as before.
  const float* gn_gamma; const float* gn_beta; bf16_t* gn_y; int gn_groups; float gn_eps; int gn_silu; int gn_keep_out; int* gn_fused;
  int kg2;                          // caller: 1 = unsplit 1x1 launches on 64-row tiles (one workgroup per CU) run two K groups of waves per workgroup (igemm_kernel KG = 2)
  int smap;                         // caller: 1 = 3x3 stride-1 convs on 8 x 8 maps take the whole-images-resident kernel (igemm_smap.h)
  int wreg, wreg_mmin, wreg_mmax;   // caller: bit 0 = plain / bit 1 = GEGLU 1x1 launches with wreg_mmin <= M <= wreg_mmax take the weight-streaming kernel when Wfrag is set; bit 2 = two K groups of waves where the launch is at most one workgroup per CU
  const bf16_t* Wfrag; int wfrag_ni; // the same matrix in MFMA fragment order for the weight-streaming kernel (igemm_wreg.h): column ranges of wfrag_ni x 16, KC = K
  int xb_m, xb_n;                   // set by the launcher: the tile grid is cut into eight xb_m x xb_n blocks, one per XCD (tile_of, common.h); 0 = the A-major / W-major walk
  int xcd_block;                    // caller: 1 = allow that
  int pc;                           // caller: producer / consumer kernel (igemm_pc.h) -- bit 0 = the 1x1 launches on 64 x 160 tiles (16 x 16 / 8 x 8 maps), bit 1 = 3x3 convs of the 16 x 16 maps
  int stagger;                      // timing experiments only: start delay of the CU's second workgroup, x1024 cycles
  int dbg;                          // timing experiments only (builds with -DAGD_EXPERIMENTS)
};
int launch_igemm(const IgemmP& p, hipStream_t st);
int igemm_query(const IgemmP& p, int* cfg3);       // {BM, BN, K splits} launch_igemm would use; launches nothing

// ---------------------------------------------------------------------------------------
// Attention (flash, swapped-QK^T formulation).  Q [B][Nq][ldq] (+head*D), K/V [B][Nk][ldk].
// record_mode: 0 none; 1 DAAM: acc[img][head][t][pix] += P for batch rows >= rec_b0;
//              (hook.py mode records per-head rows the same way into a per-call buffer; the ordered head mean is train.hip's)
// ---------------------------------------------------------------------------------------
struct AttnP {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o;
  int ldq, ldk, ldv, ldo;           // row pitches in elements
  long long sq, sk, sv, so;         // per-batch strides in elements
  int B, H, D, Nq, Nk;
  float scale;
  int record_mode; int rec_b0;      // 0 none, 1 per-(image, head) rows += P, 2 hook.py (per-head rows = P), 3 per-image rows += sum over heads of P;
                                    // rec_b0 = first batch row that records (B/2 for CFG inference, 0 for train)
  float* rec; long long rec_img_stride; long long rec_head_stride;  // DAAM: [img][head][T][Nq]
  int rec_T;                        // number of token rows to record (<= Nk)
  int rec_hpb;                      // record_mode 3: heads summed per block (divides H); rows [img][H / rec_hpb][T][Nq]
  int nqt;                          // set by the launcher: query tiles per (batch, head)
  int causal;                       // 1: key j attends only to queries i >= j (CLIP text encoder)
  const float* mask;                // additive fp32 [B][Nk] (broadcast over heads and queries) or NULL
};
int launch_attention(const AttnP& p, hipStream_t st);

// ---------------------------------------------------------------------------------------
// Fused row-panel kernels of the transformer blocks at C = 320 (tblock.hip)
// ---------------------------------------------------------------------------------------
// norm3 -> GEGLU projection -> ff.net.2 + residual in one launch; the LayerNorm is folded (W1 = W diag(gamma), cs1 = its column sums,
// b1 = bias + W beta: values then gates, un-permuted hidden index), its row statistics are taken from the rows themselves
struct FFusedP {
  const bf16_t* h; bf16_t* out;     // [M][C] residual stream in / out (may alias: a workgroup reads its 128 rows before it writes them)
  const bf16_t* w1f;                // GEGLU projection in fragment order (launch_frag_order_w1)
  const float* cs1; const float* b1;
  const bf16_t* w2f;                // ff.net.2 [C][4C] in fragment order (launch_frag_order_w, NI = C / 64, KC = 128)
  const float* b2;                  // [C]
  int M; float ln_eps;
  // optional proj_out stage behind it (wpf != NULL): pout = (h + FF(h)) . Wp^T + bp + xres; `out` is then not written.
  // colstat (optional): [M / 128][C] float2 per-(tile, channel) sums of the bf16 outputs for the GroupNorm that reads pout
  const bf16_t* wpf; const float* bp; const bf16_t* xres; bf16_t* pout; float* colstat;
  int xres_rows;                    // > 0: xres holds xres_rows rows, output row m adds row m % xres_rows (CFG-shared prefix)
  int premul;                       // 1 (with wpf): w2f holds Wp W2 and bp holds Wp b2 + bp (pre-multiplied at load time); the proj_out stage adds Wp . h on top of GEMM2's sums
};
int launch_ff_fused(const FFusedP& p, int C, hipStream_t st);
// norm2 -> to_q -> cross-attention (77 keys, recorder optional) -> to_out + bias + residual in one launch; 8 heads of 40
struct AttnChainP {
  const bf16_t* h; bf16_t* out;     // [M][C] residual stream in / out (may alias)
  const float* gamma; const float* beta; float ln_eps;     // norm2
  const bf16_t* wqf;                // attn2.to_q [C][C] in fragment order (launch_frag_order_w, NI = C / 64, KC = C)
  const bf16_t* wof; const float* bo;                      // attn2.to_out.0
  const bf16_t* kv; int ldkv; long long skv;               // projected context [B][T][ldkv]: K at column 0, V at column C
  int M, HW, T; float scale;
  int record; float* rec; long long rec_img_stride, rec_head_stride; int rec_T, rec_b0, rec_hpb;   // as AttnP record_mode 3 (rec_hpb in {1, 2, 4, 8})
  float* rowstat_out;               // optional: [M] float2 (sum, sum of squares) of the bf16 outputs (LayerNorm-fold producer, one slot)
  // optional prologue (o1 != NULL): h1 = o1 . Wo1^T + bo1 + h first (attn1.to_out + residual); h1 goes to `out` (!= h) and is the chain's input
  const bf16_t* o1; const bf16_t* wo1f; const float* bo1;
  int rows32;                       // C = 640: allow 32-row panels (twice the workgroups) where 64-row panels leave CUs idle
  int src_rows;                     // > 0: the INPUT tensors (h, o1) hold src_rows rows and output row m reads input row m % src_rows -- the CFG-shared prefix's
                                    // duplication happens here instead of in copy launches (needs out != h)
};
int launch_attn_chain(const AttnChainP& p, int C, int heads, hipStream_t st);
// proj_in (GroupNorm folded into per-image matrices, launch_gn_fold_weight with frag_ni = C / 64) -> h (stored) -> norm1 -> fused q / k / v
// projection in one launch; h [M][C], qkv [M][3C]
struct QkvChainP {
  const bf16_t* x;                  // [M][C] raw block input
  const bf16_t* wbf; long long wb_stride; const float* rowadd;   // per-image proj_in matrices in fragment order (+ image * wb_stride elements), per-image rows [B][C]
  bf16_t* h;                        // [M][C] residual stream out
  const float* gamma; const float* beta; float ln_eps;           // norm1
  const bf16_t* wqkvf;              // attn1 [3C][C] (q rows, k rows, v rows) in fragment order (NI = C / 64, KC = C)
  bf16_t* qkv;                      // [M][3C]
  int M, HW;
  long long rowadd_stride;          // elements between the images' rows of `rowadd` (0: one row -- the plain proj_in bias)
  // gn_part != nullptr: x is the RAW activation and the kernel applies the transformer's GroupNorm itself -- statistics from the producer's
  // per-(M tile, channel) partial sums [B][HW / gn_bm][C][2], rows normalised (and rounded to bf16, as gn_apply would) inside the LDS panel;
  // wbf is then the plain proj_in matrix in fragment order (wb_stride 0) and rowadd its bias
  const float* gn_part; int gn_bm, gn_groups; float gn_eps; const float* gn_gamma; const float* gn_beta;
  int rows64;                       // C = 320: 64-row panels on four waves, two workgroups co-resident per CU (tblock_fuse bit 11)
  int sched2;                       // round 6's schedule (qkv_chain2_kernel, tblock_fuse bit 12): same results bit for bit
};
int launch_qkv_chain(const QkvChainP& p, int C, hipStream_t st);
int launch_frag_order_w1(const bf16_t* src, bf16_t* dst, int C, int HID, hipStream_t st);
int launch_frag_order_w(const bf16_t* src, bf16_t* dst, int N, int K, int NI, int KC, hipStream_t st);

// ---------------------------------------------------------------------------------------
// attn2 of the C = 1280 blocks against per-image pre-multiplied context matrices (xattn_pre.hip)
// ---------------------------------------------------------------------------------------
#define XATTN_TP 80                 // token rows per head in the pre-multiplied matrices (77 padded; rows >= T are zeros)
// once per prompt batch: K''[b][(h,t)][c] = gamma[c] scale sum_d k[b][t][h,d] Wq[(h,d)][c] (bf16), cs = its row sums, bs[(h,t)] = scale k[t][h,:] . (Wq beta)[h,:],
// V''[b][n][(h,t)] = sum_d Wo[n][(h,d)] v[b][t][h,d] (bf16)
struct XattnPremulP {
  const bf16_t* kv; int ldkv; long long skv;      // projected context [B][T][ldkv]: K at column 0, V at column C
  const bf16_t* wqT;                              // attn2.to_q transposed: [c][(h,d)]
  const bf16_t* wo;                               // attn2.to_out.0 [n][(h,d)]
  const float* gamma; const float* wqb;           // norm2.weight [C]; Wq . norm2.bias [C]
  int B, T, C, H; float scale;
  bf16_t* kpp; float* kcs; float* kbs; bf16_t* vpp;   // [B][H TP][C], [B][H TP], [B][H TP], [B][C][H TP]
};
int launch_xattn_premul(const XattnPremulP& p, hipStream_t st);
// per forward: P[m][(h,t)] = softmax_t(rstd_m (x[m] . K''[(h,t)] - mu_m cs[(h,t)]) + bs[(h,t)]) (bf16), recorder rows [img - rec_b0][head][t][pixel] += P
struct XattnSP {
  const bf16_t* x;                                // [M][C] raw residual stream (norm2 folded)
  const float* ln_stats; int ln_slots; float ln_invC, ln_eps;   // per-row (sum, sum of squares) partials of x from its producer: [M][slots] float2
  const bf16_t* kpp; const float* kcs; const float* kbs;
  bf16_t* P;                                      // [M][H TP]
  int M, HW, C, H, T;
  float* rec; long long rec_img_stride, rec_head_stride; int rec_T, rec_b0;   // optional recorder: per-(image, head) rows [T][HW]
};
int launch_xattn_s(const XattnSP& p, hipStream_t st);
int launch_matvec_bf16(const bf16_t* W, const float* v, float* out, int N, int K, hipStream_t st);      // out[j] = sum_c W[j][c] v[c]
int launch_rowstat_bf16(const bf16_t* x, float* out, int M, int C, hipStream_t st);                    // [M] float2 (sum, sum of squares) per row

// ---------------------------------------------------------------------------------------
// Norms
// ---------------------------------------------------------------------------------------
struct GroupNormP {
  const bf16_t* x0; const bf16_t* x1; int C0, C1;   // optional channel-concat input
  bf16_t* y;                                         // [B][HW][C0+C1]
  const float* gamma; const float* beta;
  int B, HW, groups; float eps; int silu;
  float* ws;                                         // workspace: B*C*2 (sums) + B*C*2 (scale/shift)
  // per-channel partial sums emitted by the igemm launches that produced x0 / x1 ([HW*B / bm tiles][Cs] float2, bm rows per
  // tile): when given for every source, the statistics pass over the activation is skipped (one kernel instead of two)
  const float* part0; const float* part1; int bm0, bm1;
};
int launch_groupnorm(const GroupNormP& p, hipStream_t st);
// GroupNorm (no activation) folded into the 1x1 projection that follows it: per image, Wb = W . diag(rstd_g gamma) (bf16) and
// rowadd = bias + W beta - Wb mu (fp32), statistics from the producer's per-(tile, channel) partial sums [HW / bm tiles][C] float2
int launch_gn_fold_weight(const float* part, int bm, int B, int HW, int C, int groups, float eps, const float* gamma, const float* beta,
                          const bf16_t* W, const float* bias, int N, bf16_t* Wb, float* rowadd, hipStream_t st, int frag_ni = 0);   // frag_ni > 0: Wb in MFMA fragment order (tblock.hip)
long long groupnorm_ws_floats(int B, int C, int HW, int groups);   // workspace floats a launch_groupnorm call needs
int launch_layernorm(const bf16_t* x, bf16_t* y, const float* g, const float* b, int rows, int C, float eps, hipStream_t st);

// ---------------------------------------------------------------------------------------
// Misc elementwise / layout kernels
// ---------------------------------------------------------------------------------------
int launch_convert_weight(const float* w, bf16_t* out, int N, int Cin, int taps, int Cpad, int geglu_bn, hipStream_t st);
int launch_f32_to_bf16(const float* x, bf16_t* y, long long n, hipStream_t st);
int launch_bf16_to_f32(const bf16_t* x, float* y, long long n, hipStream_t st);
bool igemm_can_fuse_shortcut(const IgemmP& p);
int launch_upsample_phase_weight(const bf16_t* w, bf16_t* out, int Cout, int Cin, hipStream_t st);   // [Cout][9][Cin] -> [4 Cout][4][Cin]
int launch_prep_latents(const float* lat_nchw, bf16_t* out_nhwc, int B, int C, int HW, int Cpad, int dup, float scale, hipStream_t st);
int launch_timestep_embed(float t, float* out, int dim, hipStream_t st);
int launch_small_linear(const float* x, const bf16_t* W, const float* bias, float* out, int M, int N, int K, int silu_in,
                        int silu_out, hipStream_t st);
int launch_cfg_ddim(const float* eps_nhwc, int ldc, float* lat_nchw, int B, int C, int HW, float guidance,
                    float a_t, float a_p, int vpred, hipStream_t st);
int launch_cfg_plms(const float* eps_nhwc, int ldc, float* lat, const float* src, const float* h1, const float* h2, const float* h3,
                    float* store, int B, int C, int HW, float guidance, const float* w4, float a, float b, hipStream_t st);
int launch_ln_fold_weight(const bf16_t* W, const float* gamma, const float* beta, const float* bias, int N, int K, int geglu_bn, bf16_t* Wf,
                          float* colsum, float* bias_f, hipStream_t st);
int launch_image_u8(const float* x_nhwc, int ldc, unsigned char* out, long long npix, int C, hipStream_t st);
int launch_nchw_from_nhwc_f32(const float* x, int ldc, float* out, int B, int C, int HW, hipStream_t st);
int launch_softmax_rows(const float* s, bf16_t* p, int rows, int cols, hipStream_t st);

// Heat maps
struct HeatLayer { const float* acc; int side; int heads; long long img_stride; long long head_stride; };
int launch_daam_global(const HeatLayer* layers, int n_layers, int total_maps, int T, int S, int img, float* out, hipStream_t st);
int launch_hook_accum(const float* map, int B, int T, int side, int S, float* sum, hipStream_t st);
int launch_scale(float* x, long long n, float s, hipStream_t st);

// export path (bit-exact with numpy min-max/astype and PIL Image.resize BICUBIC on uint8)
int launch_heatmap_u8(const float* hm, int n, int npix, unsigned char* out, hipStream_t st);
int launch_pil_resample(const unsigned char* in, unsigned char* out, const int* bounds, const int* kk, int ksize,
                        long long n_outer, int in_len, int out_len, int inner, hipStream_t st);
int launch_stack_heatmaps(const unsigned char* obj, const unsigned char* fg, const unsigned char* bg, long long npix,
                          unsigned char* rgb, unsigned char* inv, hipStream_t st);

// training-mode seam (train.hip)
int launch_hook_headmean(const float* heads, int Bp, int H, int T, int N, float* map, hipStream_t st);
int launch_attn_reg_loss(const float* map, int B, int T, int P, const int* obj_idx, const int* fg_idx, const int* bg_idx, float coef,
                         float* loss_out, float* dmap, hipStream_t st);
int launch_attention_backward(const bf16_t* q, const bf16_t* kv, const bf16_t* dout, const float* dmap, int b0, int B, int H, int D, int N, int T,
                              float scale, bf16_t* dq, bf16_t* dkv, float* part, hipStream_t st);
long long attention_backward_ws_floats(int B, int H, int D, int N, int T);
int launch_transpose_bf16(const bf16_t* in, int R, int Cc, bf16_t* out, hipStream_t st);

int launch_embed_gather(const int* ids, const bf16_t* tok, const bf16_t* pos, bf16_t* out, int B, int T, int H, int vocab_cap, hipStream_t st);
