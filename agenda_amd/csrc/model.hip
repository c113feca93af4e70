// Host runtime + C-ABI of libagenda_hip.so (see include/agenda_hip.h for the contract).
// Owns: bf16 re-laid weights, an activation arena, cross-attention K/V caches, heat-map
// accumulators; walks the SD UNet / VAE-decoder graphs as stream-ordered HIP kernel launches.
#include "../../include/agenda_hip.h"
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
static thread_local char g_err[2048] = "";
// C ABI entry points are the only default-visibility symbols of the library (built with -fvisibility=hidden)
#define AGD_API extern "C" __attribute__((visibility("default")))
extern "C" void agd_set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
#define CK(expr) do { if ((expr) != 0) return -1; } while (0)
#define FAIL(...) do { agd_set_error(__VA_ARGS__); return -1; } while (0)

struct WMat { bf16_t* w = nullptr; int N = 0, Cin = 0, Cpad = 0, taps = 0;
              int sc_cols = 0;                                   // conv2 with its block's conv_shortcut appended: rows are [taps x Cpad | sc_cols] (igemm_halo.h shortcut loop)
              bf16_t* wfrag = nullptr; int wfrag_ni = 0; };   // the matrix once more in MFMA fragment order (igemm_wreg.h), column ranges of wfrag_ni x 16
// cpart: per-(M tile, channel) partial sums the producing igemm launch leaves for a following GroupNorm
// ([B*H*W / cpart_bm][C] float2; cpart_bm = 0: none were produced -> the GroupNorm runs its own statistics pass)
// normed / normed_gamma: a GroupNorm of this activation that its producer already applied (split-K slab pass, igemm.hip splitk_reduce_gn_kernel):
// the consumer whose norm weights are `normed_gamma` reads `normed` instead of running the GroupNorm kernels
struct Act { bf16_t* p = nullptr; int B = 0, H = 0, W = 0, C = 0; float* cpart = nullptr; int cpart_bm = 0;
             bf16_t* normed = nullptr; const float* normed_gamma = nullptr;
             long long n() const { return (long long)B * H * W * C; } };
// the GroupNorm that will read a resnet's output next, when it reads that tensor alone (no concat): lets a split-K conv2 normalise in its slab pass
struct NextGn { const float* gamma = nullptr; const float* beta = nullptr; float eps = 0.f; int silu = 0; };

struct Arena {
  char* base = nullptr; size_t cap = 0, off = 0, peak = 0;
  void* alloc(size_t bytes) {
    size_t a = (off + 255) & ~(size_t)255;
    if (a + bytes > cap) { agd_set_error("activation arena exhausted: need %zu + %zu > %zu bytes (raise workspace_bytes)", a, bytes, cap); return nullptr; }
    off = a + bytes; if (off > peak) peak = off;
    return base + a;
  }
  size_t mark() const { return off; }
  void release(size_t m) { off = m; }
};

enum { PC_CONV3 = 0, PC_GEMM, PC_ATTN_SELF, PC_ATTN_CROSS, PC_GN, PC_LN, PC_ELEM, PC_HEAT, PC_VAE_ATTN, PC_OTHER, PC_TOUCH };
static const char* kClassNames[AGD_N_CLASSES] = {"igemm_conv3x3", "igemm_linear_1x1", "attn_self_flash", "attn_cross_daam",
                                                 "groupnorm", "layernorm", "elementwise", "heatmap", "vae_attn_softmax", "other", "weight_touch"};

// device buffer that only grows (capacity in bytes); the old block is freed when a larger one is needed
struct DBuf {
  void* p = nullptr; size_t cap = 0;
  template <typename T> T* as() const { return (T*)p; }
  int ensure(size_t bytes) {
    if (p && bytes <= cap) return 0;
    if (p) { (void)hipDeviceSynchronize(); (void)hipFree(p); p = nullptr; cap = 0; }
    const size_t want = bytes ? bytes : 256;
    if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; agd_set_error("hipMalloc(%zu) failed", want); return -1; }
    cap = want;
    return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct XLayer {
  std::string name; int C = 0, heads = 0, level = 0; bool mid = false;
  WMat wkv; DBuf kvb; bf16_t* kv = nullptr; DBuf accb; float* acc = nullptr; int acc_side = 0;
  int acc_heads = 0;                                  // slices in acc per image: `heads`, or fewer = head-group sums (layers at latent resolution)
  DBuf wqTb, wkvTb, woTb; WMat wqT, wkvT, woT;        // transposed weights for the input-gradient GEMMs (built on first backward)
  // attn2 against per-image pre-multiplied context matrices (xattn_pre.hip; the C = 1280 blocks): load-time parts (to_q transposed, Wq . norm2.bias)
  // and the per-prompt-batch products K'' / cs / bs / V'' (agd_set_context)
  bf16_t* pm_wqT = nullptr; float* pm_wqb = nullptr; bool pm_ready = false;
  DBuf pm_kppb, pm_vppb, pm_csb; bf16_t* pm_kpp = nullptr; bf16_t* pm_vpp = nullptr; float* pm_kcs = nullptr; float* pm_kbs = nullptr;
};

struct ProfEv { int cls; double flops, bytes; hipEvent_t a, b; };   // algorithmic flop / HBM bytes of the launch

struct agd_ctx {
  int device = 0; agd_config cfg{}; std::string err;
  std::unordered_map<std::string, WMat> W;
  std::unordered_map<std::string, float*> V;
  std::unordered_map<std::string, int> Vn;
  std::vector<void*> owned;
  Arena arena; bf16_t* zero_page = nullptr;
  float* stage = nullptr; size_t stage_bytes = 0;
  bool finalized = false;
  // time embedding
  WMat tproj_all; float* tproj_bias = nullptr; float* tproj_out = nullptr; int tproj_total = 0;
  const float* tproj_cur = nullptr;                 // time_emb_proj outputs of the forward being walked
  int tproj_cur_ld = 0;                             // 0: one timestep for every image; tproj_total: one row per image (agd_unet_forward_ts)
  float* tsteps_buf = nullptr; int tsteps_cap = 0;   // agd_denoise: embeddings of ALL steps, computed up front
  std::unordered_map<std::string, int> tproj_off;
  float* temb_buf = nullptr;   // [dim | 4dim | 4dim] fp32 scratch
  float* t_dev = nullptr;
  // cross attention
  std::vector<XLayer> xl; std::unordered_map<std::string, int> xl_idx;
  DBuf ctxb; bf16_t* ctx_bf16 = nullptr; int ctx_B2 = 0, ctx_T = 0;
  // recorder
  int rec_mode = 0, rec_is_train = 0, rec_T_cfg = 0, rec_T = 0, rec_B = 0, rec_L = 0;   // rec_T: rows the accumulators were sized for (set by record_reset)
  DBuf hook_sumb, hook_scratchb, hook_headsb, hook_storeb;     // hook_heads: per-head probabilities of one call; hook_store: kept per-call maps (train mode)
  struct HookRec { size_t off; int Bp, T, N; };
  std::vector<HookRec> hook_recs; size_t hook_store_used = 0;
  DBuf bwd_wsb;                                                 // attention-backward workspace
  float* hook_sum = nullptr; float* hook_scratch = nullptr; int hook_count = 0, hook_Bp = 0, hook_T = 0;
  // denoise scratch
  DBuf latb, epsb, vae_imgb, plmsb;
  bf16_t* lat_bf16 = nullptr; float* eps_nhwc = nullptr;
  SplitKWs splitk;                                    // split-K partial slabs of this ctx (stream-ordered reuse)
  int opt_cfg_share = 1;                              // agd_set_option("cfg_shared_prefix")
  int opt_ln_fold = 1;                                // agd_set_option("ln_fold"): LayerNorm folded into the GEMMs around it
  int opt_gn_fused = 1;                               // agd_set_option("gn_fused_stats"): GroupNorm statistics from the producing igemm's epilogue
  int opt_warm = 3;                                   // agd_set_option("weight_warm"): in-kernel cold-weight warm-up: 1 = W-major launches (per-XCD slices), 3 = A-major launches too
  int opt_gn_proj_fold = 1;                           // agd_set_option("gn_proj_fold"): the transformers' GroupNorm folded into per-image proj_in matrices (1: C <= 320, 2: C <= 640)
  int opt_p8 = 1;                                     // agd_set_option("igemm8p"): 256-row 8-wave / 8-phase igemm for launches with enough tiles (igemm8p.h)
  int opt_halo = 1;                                   // agd_set_option("conv_halo"): 3x3 stride-1 convs through the row-halo kernel (igemm_halo.h)
  int opt_tb_fuse = 255 | 512 | 1024 | 2048 | 4096;                 // agd_set_option("tblock_fuse"): fused row-panel kernels of the C = 320 transformer blocks (tblock.hip): bit 0 = feed-forward (bit 3: + proj_out),
                                                      // bit 1 = attn2 chain (bit 2: + attn1.to_out in front of it), bit 4 = proj_in -> norm1 -> qkv, bit 5 = the attn2 chain for the C = 640 blocks too, bit 6 = the CFG-shared prefix's duplication inside the fused kernels, bit 7 = the GroupNorm applied inside the bit-4 launch (no fold launch), bit 8 (off) = the bit-4 launch with the GroupNorm inside for the C = 640 blocks too, bit 9 = ff.net.2 / proj_out pre-multiplied inside the feed-forward launch, bit 10 = 32-row panels for the C = 640 attn2 chain,
                                                      // round 6: bit 11 = the bit-4 launch on 64-row panels (two co-resident four-wave workgroups per CU), bit 12 = on qkv_chain2_kernel's schedule for the C = 640 blocks, bit 9 = ff.net.2 / proj_out pre-multiplied inside the feed-forward kernel, bit 10 = 32-row panels for the C = 640 attn2 chain where 64-row panels would fill half the chip
  int opt_ups4 = 7; /* see agd_set_option */                                   // agd_set_option("upsample_phases"): the UNet's nearest-2x upsampling convs as four 2x2 phase convs on the un-upsampled map (one launch, 4/9 of the MACs)
  int opt_ffproj = 1;                                 // agd_set_option("ff_proj_fuse"): ff.net.2 and proj_out as ONE GEMM with the pre-multiplied matrix [Wp W2 | Wp] over [hidden | h] (blocks whose feed-forward is not the fused row-panel kernel)
  int opt_sc_fuse = 3;                                // agd_set_option("shortcut_fuse"): a UNet resnet's 1x1 conv_shortcut runs as extra K of its conv2 launch where that is an unsplit row-halo launch
  int opt_wreg = 3;                                   // agd_set_option("wreg_mask"): weight-streaming kernel (igemm_wreg.h) for bit 0: the C = 1280 GEGLU at 1024 <= M <= 4096 (the 16 x 16 blocks), bit 1: proj_in / proj_out of the C = 640 blocks
  int opt_kg2 = 1;                                    // agd_set_option("igemm_kgroups"): two K groups of waves per workgroup on the one-workgroup-per-CU 1x1 launches of the small maps
  int opt_side = 0;                                   // agd_set_option("side_stream"): a resnet's 1x1 conv_shortcut runs on a second stream beside norm1 / conv1 / norm2
  hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int opt_smap = 1;                                   // agd_set_option("conv_smap"): 3x3 convs of the 8 x 8 maps through the whole-images-resident kernel (igemm_smap.h)
  int opt_reduce_gn = 1;                              // agd_set_option("reduce_gn"): split-K slab sum + the GroupNorm that reads it as one launch (igemm.hip splitk_reduce_gn_kernel)
  int opt_xcd_block = 1;                              // agd_set_option("xcd_block"): the igemm tile grid cut into one block per XCD that minimises the XCD's L2 working set (igemm.hip pick_xcd_block)
  int opt_pc = 49;                                    // agd_set_option("igemm_pc"): producer / consumer igemm (igemm_pc.h) -- bit 0: 1x1 launches on 64 x 160 tiles, bit 1: 3x3 convs of the 16 x 16 maps;
                                                      // bit 4: the row-halo producer / consumer kernel (igemm_pch.h) for 3x3 convs whose 128 x 160 tiles (x K slices) fit one wave of workgroups;
                                                      // bit 5: the 1x1 launches of one 128 x 160 tile per CU (M = 8192, N = 640) on igemm_pc.h's 128-row form
  int opt_xpre = 1;                                   // agd_set_option("attn2_premul"): attn2 of the C = 1280 blocks as two GEMMs against per-image pre-multiplied context matrices (xattn_pre.hip)
  int opt_touch = 3;                                  // agd_set_option("weight_touch"): n > 0 = stream 1x1 weight matrices of >= n MB through the caches right before their launch
  unsigned* touch_sink = nullptr;
  // profiling
  bool prof_on = false; std::vector<ProfEv> prof; std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
  long long launches[AGD_N_CLASSES] = {0};
};

static int fail_ctx(agd_ctx* c) { if (c) c->err = g_err; return -1; }
#define API_CK(c, expr) do { if ((expr) != 0) return fail_ctx(c); } while (0)

struct ProfScope {
  agd_ctx* c; hipStream_t st; bool on; size_t idx;
  ProfScope(agd_ctx* c_, hipStream_t st_, int cls, double flops, double bytes = 0) : c(c_), st(st_), on(c_ && c_->prof_on), idx(0) {
    if (c) c->launches[cls]++;
    if (!on) return;
    auto get = [&]() { if (c->ev_used == c->ev_pool.size()) { hipEvent_t e; hipEventCreate(&e); c->ev_pool.push_back(e); } return c->ev_pool[c->ev_used++]; };
    ProfEv pe; pe.cls = cls; pe.flops = flops; pe.bytes = bytes; pe.a = get(); pe.b = get();
    hipEventRecord(pe.a, st);
    idx = c->prof.size(); c->prof.push_back(pe);
  }
  ~ProfScope() { if (on) hipEventRecord(c->prof[idx].b, st); }
};

// frees a dmalloc'd block before agd_destroy (load-time scratch); the caller has synchronised
static void dfree(agd_ctx* c, void* p) {
  if (!p) return;
  if (c) { auto it = std::find(c->owned.begin(), c->owned.end(), p); if (it != c->owned.end()) c->owned.erase(it); }
  (void)hipFree(p);
}
template <typename T> static T* dmalloc(agd_ctx* c, size_t n) {
  void* p = nullptr;
  if (hipMalloc(&p, n * sizeof(T) ? n * sizeof(T) : 256) != hipSuccess) { agd_set_error("hipMalloc(%zu) failed", n * sizeof(T)); return nullptr; }
  if (c) c->owned.push_back(p);
  return (T*)p;
}

// stats: the activation will be written by an igemm launch and read by a GroupNorm -> room for the producer's partial sums
static Act alloc_act(agd_ctx* c, int B, int H, int W, int C, bool stats = false) {
  Act a; a.B = B; a.H = H; a.W = W; a.C = C;
  a.p = (bf16_t*)c->arena.alloc((size_t)a.n() * 2);
  if (a.p && stats && c->opt_gn_fused && ((long long)H * W) % 64 == 0) {
    a.cpart = (float*)c->arena.alloc((size_t)((long long)B * H * W / 64) * C * 2 * sizeof(float));     // worst case: 64-row tiles
    if (!a.cpart) a.p = nullptr;
  }
  return a;
}

// ---------------------------------------------------------------------------------------
// op wrappers
// ---------------------------------------------------------------------------------------
struct GemmOpt {
  const float* bias = nullptr; const float* rowadd = nullptr; int rowadd_ld = 0;
  const bf16_t* residual = nullptr; int ldr = 0; int geglu = 0; int act = 0; int out_f32 = 0; int ldo = 0;
  int stride = 1, up = 1; float alpha = 1.f;
  int* query_cfg = nullptr;                                                              // igemm_query only: {BM, BN, splits}
  Act* out_act = nullptr;                                                                // output feeds a GroupNorm: leave per-channel partial sums in it
  int rows_per_image = 0;                                                                // for out_act on linear-shaped launches (B=1, W=M): rows of one image
  float* rowstat_out = nullptr; int rowstat_slots = 0;                                   // producer side of the LayerNorm fold
  const float* ln_stats = nullptr; int ln_slots = 0; const float* ln_cs = nullptr; float ln_invC = 0.f, ln_eps = 0.f;   // consumer side
  int warm = 0;                 // benches: request the in-kernel cold-weight warm-up (the walk sets it through the ctx option)
  int halo = 0;                 // benches: allow the row-halo 3x3 kernel (the walk sets it through the ctx option)
  int want_rowstat = 0;         // igemm_query only: the launch will be a LayerNorm row-statistics producer (changes the kernel family)
  int p8 = 0;                   // benches: 1 = allow the 8-phase kernel, 2 / 3 = force its 256- / 160-wide tile (the walk sets it through the ctx option)
  int w_per_image = 0;          // 1x1 launches: image i multiplies with w.w + i * N * K (GroupNorm folded into per-image matrices)
  // the GroupNorm(+SiLU) that reads this launch's output next, for split-K launches (IgemmP::gn_y): *gn_fused = 1 when the slab-sum pass did it
  const float* gn_gamma = nullptr; const float* gn_beta = nullptr; bf16_t* gn_y = nullptr; int gn_groups = 0; float gn_eps = 0.f; int gn_silu = 0, gn_keep_out = 0; int* gn_fused = nullptr;
  int kg2 = 0;                  // benches / tests: two K groups of waves per workgroup on the 64-row 1x1 tiles (the walk sets it through the ctx option)
  int smap = 0;                 // benches / tests: allow the 8 x 8 whole-images-resident 3x3 kernel (the walk sets it through the ctx option)
  int xcd_block = 0;            // benches / tests: XCD-aware tile blocks (IgemmP::xcd_block; the walk sets it through the ctx option)
  int pc = 0;                   // benches / tests: producer / consumer kernel mask (IgemmP::pc; the walk sets it through the ctx option)
  int wreg = 0;                 // benches / tests: weight-streaming kernel for this launch when the WMat carries a fragment-order copy (bit 0 plain, bit 1 GEGLU)
  const bf16_t* sc0 = nullptr; const bf16_t* sc1 = nullptr; int sc_C0 = 0, sc_C1 = 0;   // the block's 1x1 conv_shortcut as extra K of this 3x3 launch (WMat::sc_cols)
  int ups4 = 0;                 // > 0: phase-decomposed upsampling conv (IgemmP::ups4 = Cout; ksize 2, the merged [4 Cout][4][Cin] matrix, hout / wout = the input size)
  int* can_fuse_sc = nullptr;   // query only: *can_fuse_sc = 1 when this launch could take a shortcut that way (nothing is launched)
  int pad = -1;                 // -1: 1 for 3x3, 0 for 1x1
  int hout = 0, wout = 0;       // >0: override (asymmetric (0,1,0,1) padding of the VAE encoder's stride-2 convs)
};
static const int kTextExtraRows = 256;   // room for tokenizer.add_tokens() (learned tokens)

// streams `n16` 16-byte words through the memory hierarchy and keeps nothing
__global__ __launch_bounds__(256) void touch_kernel(const u32x4* __restrict__ p, long long n16, unsigned* __restrict__ sink) {
  unsigned a = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) { const u32x4 v = p[i]; a |= v[0] ^ v[1] ^ v[2] ^ v[3]; }
  if (a == 0x9E3779B9u) *sink = a;
}
static int run_conv(agd_ctx* c, hipStream_t st, const bf16_t* s0, int C0, const bf16_t* s1, int C1, int B, int Hin, int Win,
                    const WMat& w, int ksize, void* out, const GemmOpt& o, const bf16_t* zero_page) {
  IgemmP p{};
  p.src0 = s0; p.src1 = s1; p.C0 = C0; p.C1 = C1; p.Hin = Hin; p.Win = Win;
  p.ksize = ksize; p.stride = o.stride; p.pad = o.pad >= 0 ? o.pad : (ksize == 3 ? 1 : 0); p.up = o.up;
  p.Hout = o.hout > 0 ? o.hout : (Hin * o.up + 2 * p.pad - ksize) / o.stride + 1;
  p.Wout = o.wout > 0 ? o.wout : (Win * o.up + 2 * p.pad - ksize) / o.stride + 1;
  p.W = w.w; p.bias = o.bias; p.bias_mode = o.bias ? 1 : 0; p.rowadd = o.rowadd; p.rowadd_ld = o.rowadd_ld;
  p.residual = o.residual; p.N = w.N; p.K = w.taps * w.Cpad; p.M = B * p.Hout * p.Wout; p.ups4 = o.ups4;
  if (o.sc0) {
    if (w.sc_cols != o.sc_C0 + o.sc_C1 || w.sc_cols < 64) FAIL("conv: weight carries %d shortcut columns, the launch %d + %d", w.sc_cols, o.sc_C0, o.sc_C1);
    p.sc0 = o.sc0; p.sc1 = o.sc1; p.sc_C0 = o.sc_C0; p.sc_C1 = o.sc_C1; p.K += w.sc_cols;
  }
  const int nout = o.geglu ? w.N / 2 : w.N;
  p.ldr = o.ldr ? o.ldr : nout; p.out = out; p.out_f32 = o.out_f32; p.ldo = o.ldo ? o.ldo : nout;
  p.alpha = o.alpha; p.geglu = o.geglu; p.act = o.act; p.batch = 1; p.zero_page = zero_page; p.ws = c ? &c->splitk : nullptr;
  p.rowstat_out = o.rowstat_out; p.rowstat_slots = o.rowstat_slots;
  p.ln_stats = o.ln_stats; p.ln_slots = o.ln_slots; p.ln_cs = o.ln_cs; p.ln_invC = o.ln_invC; p.ln_eps = o.ln_eps;
  if (o.w_per_image) { p.w_per_image = 1; p.sW = (long long)w.N * w.taps * w.Cpad; }
  if (o.wreg && w.wfrag) { p.Wfrag = w.wfrag; p.wfrag_ni = w.wfrag_ni; p.wreg = o.wreg; p.wreg_mmin = 1; p.wreg_mmax = 1 << 30; }
  if (o.gn_y && c && c->opt_reduce_gn) { p.gn_gamma = o.gn_gamma; p.gn_beta = o.gn_beta; p.gn_y = o.gn_y; p.gn_groups = o.gn_groups; p.gn_eps = o.gn_eps; p.gn_silu = o.gn_silu;
                                         p.gn_keep_out = o.gn_keep_out; p.gn_fused = o.gn_fused; }
  p.halo = (c && c->opt_halo) || o.halo;
  p.smap = (c && c->opt_smap) || o.smap;
  p.kg2 = (c && c->opt_kg2) || o.kg2;
  p.pc = (c ? c->opt_pc : 0) | o.pc;
  p.xcd_block = (c ? c->opt_xcd_block : 0) | o.xcd_block;
  p.p8 = o.p8 < 0 ? 0 : o.p8 ? o.p8 : c ? c->opt_p8 : 0;          // before any igemm_query: the tile shape depends on it (-1: not for this launch)
  if (o.query_cfg) { if (o.want_rowstat && !p.rowstat_out) { p.rowstat_out = (float*)16; p.rowstat_slots = 0; } return igemm_query(p, o.query_cfg); }   // (nothing is launched)
  if (o.out_act) {
    o.out_act->cpart_bm = 0;
    if (o.out_act->cpart && !o.out_f32 && !o.geglu) {
      int cfg[3] = {0, 0, 0};
      if (igemm_query(p, cfg) != 0) {                      // (a shortcut-fusion query on a launch that cannot take the shortcut: "no", not an error)
        if (o.can_fuse_sc) { *o.can_fuse_sc = 0; return 0; }
        return -1;
      }
      const int rpi = o.rows_per_image > 0 ? o.rows_per_image : p.Hout * p.Wout;
      // split-K launches keep the separate statistics kernel: a reduce pass that also emits partial sums was measured
      // slower than the two it replaces (+5.7 ms per batch, tools/ab_option.py)
      if (cfg[2] == 1 && cfg[0] > 0 && rpi % cfg[0] == 0) { p.colstat_out = o.out_act->cpart; p.colstat_rows = rpi; o.out_act->cpart_bm = cfg[0]; }
    }
  }
  // (asked AFTER the column-statistics decision above: the query must see the launch exactly as it will run -- a statistics producer never splits K,
  //  which can change the tile the launcher picks)
  if (o.can_fuse_sc) { *o.can_fuse_sc = igemm_can_fuse_shortcut(p) ? 1 : 0; return 0; }
  if (w.taps != ksize * ksize || w.Cpad != C0 + C1) FAIL("conv: weight [N=%d taps=%d Cpad=%d] does not match input C=%d+%d ksize=%d", w.N, w.taps, w.Cpad, C0, C1, ksize);
  // algorithmic HBM bytes: every input pixel / weight read once, the output written once (+ the residual read)
  const double in_b = 2.0 * B * Hin * Win * (double)(C0 + C1), w_b = 2.0 * p.N * (double)p.K;
  const double out_b = (double)p.M * nout * (o.out_f32 ? 4.0 : 2.0) + (o.residual ? 2.0 * p.M * nout : 0.0);
  // W-major launches (weights >= 1.5x the activations) warm their weights inside the kernel; the launcher's rule restated here
  const bool wmaj = ksize == 1 && w_b > 1.5 * in_b && w_b >= (double)(1 << 20);
  if (c && c->opt_warm) p.warm = c->opt_warm == 1 ? 1 : 3;      // 1: W-major launches only; 3: also A-major launches with >= 1 MB of weights (the launcher decides)
  if (o.warm) p.warm = o.warm;
  if (c && c->opt_touch > 0 && ksize == 1 && p.M <= 8192 && w_b >= c->opt_touch * 1e6 && !(c->opt_warm && wmaj) && c->opt_warm != 3) {
    // the weights arrive cold (1.7 GB per forward against 256 MB of Infinity Cache): a full-rate streaming read in front of the launch
    // costs less than the tile-by-tile cold misses inside it (tools/kb_cold.py; in situ 575.5 -> 572.3 ms per batch, tools/ab_option.py;
    // touching the 3x3 matrices of the 16x16 maps as well gave the gain back)
    if (!c->touch_sink) c->touch_sink = dmalloc<unsigned>(c, 64);
    if (c->touch_sink) { ProfScope pt(c, st, PC_TOUCH, 0, w_b);
      hipLaunchKernelGGL(touch_kernel, dim3(1024), dim3(256), 0, st, (const u32x4*)w.w, (long long)(w_b / 16), c->touch_sink); }
  }
  ProfScope ps(c, st, ksize != 1 ? PC_CONV3 : PC_GEMM, 2.0 * p.M * (double)p.N * p.K, in_b + w_b + out_b + 2.0 * B * Hin * Win * (double)(o.sc_C0 + o.sc_C1));
  return launch_igemm(p, st);
}

static int run_gn(agd_ctx* c, hipStream_t st, const bf16_t* x0, int C0, const bf16_t* x1, int C1, int B, int HW,
                  const float* gamma, const float* beta, int groups, float eps, int silu, bf16_t* y,
                  const Act* a0 = nullptr, const Act* a1 = nullptr) {
  GroupNormP g{}; g.x0 = x0; g.x1 = x1; g.C0 = C0; g.C1 = C1; g.y = y; g.gamma = gamma; g.beta = beta;
  g.B = B; g.HW = HW; g.groups = groups; g.eps = eps; g.silu = silu;
  if (c->opt_gn_fused && a0 && a0->cpart_bm > 0 && (!x1 || (a1 && a1->cpart_bm > 0))) {      // every source brought its partial sums
    g.part0 = a0->cpart; g.bm0 = a0->cpart_bm;
    if (x1) { g.part1 = a1->cpart; g.bm1 = a1->cpart_bm; }
  }
  const long long wsf = groupnorm_ws_floats(B, C0 + C1, HW, groups);
  const size_t mk = c->arena.mark();
  g.ws = (float*)c->arena.alloc((size_t)wsf * 4);
  if (!g.ws) return -1;
  ProfScope ps(c, st, PC_GN, 0, 4.0 * B * HW * (double)(C0 + C1));       // one read + one write of the activation
  const int rc = launch_groupnorm(g, st);
  c->arena.release(mk);   // stream-ordered: later kernels that reuse this memory run after the norm
  return rc;
}

static const WMat* getW(agd_ctx* c, const std::string& k) {
  auto it = c->W.find(k);
  if (it == c->W.end()) { agd_set_error("missing weight '%s'", k.c_str()); return nullptr; }
  return &it->second;
}
static const float* getV(agd_ctx* c, const std::string& k) {
  auto it = c->V.find(k);
  if (it == c->V.end()) { agd_set_error("missing tensor '%s'", k.c_str()); return nullptr; }
  return it->second;
}
#define GETW(var, key) const WMat* var = getW(c, key); if (!var) return -1
#define GETV(var, key) const float* var = getV(c, key); if (!var) return -1

// ResnetBlock2D (norm1-silu-conv1 (+temb) - norm2-silu-conv2 + shortcut)
static int resnet(agd_ctx* c, hipStream_t st, const std::string& pre, const Act& x0, const Act* x1, int Cout, float eps,
                  bool has_temb, int groups, Act& out, const NextGn* next = nullptr) {
  const int B = x0.B, H = x0.H, Wd = x0.W, HW = H * Wd;
  const int C1 = x1 ? x1->C : 0, Cin = x0.C + C1;
  out = alloc_act(c, B, H, Wd, Cout, true); if (!out.p) return -1;
  bf16_t* out_normed = nullptr;
  if (next && next->gamma && c->opt_reduce_gn) { out_normed = (bf16_t*)c->arena.alloc((size_t)out.n() * 2); if (!out_normed) return -1; }   // lives as long as `out`
  const size_t mk = c->arena.mark();
  // the 1x1 conv_shortcut depends on the block's input alone: with the option on it runs on a second stream beside norm1 / conv1 / norm2
  // (those launches are one workgroup per CU or fewer on the small maps); unsplit launches only -- the split-K workspace is the main stream's
  bf16_t* side_out = nullptr;
  const bool has_sc = c->W.count(pre + "conv_shortcut.weight") != 0;
  if (has_sc && c->opt_side && (c->opt_side == 1 || HW <= c->opt_side)) {        // 1: every level; else: maps of at most that many pixels
    GETW(ws, pre + "conv_shortcut.weight"); GETV(bs, pre + "conv_shortcut.bias");
    GemmOpt os; os.bias = bs; int cfg[3] = {0, 0, 0}; os.query_cfg = cfg;
    CK(run_conv(c, st, x0.p, x0.C, x1 ? x1->p : nullptr, C1, B, H, Wd, *ws, 1, nullptr, os, c->zero_page));
    if (cfg[2] == 1) {
      if (!c->side) {
        if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) FAIL("side stream");
      }
      side_out = (bf16_t*)c->arena.alloc((size_t)B * HW * Cout * 2); if (!side_out) return -1;
      os.query_cfg = nullptr;
      if (hipEventRecord(c->ev_fork, st) != hipSuccess || hipStreamWaitEvent(c->side, c->ev_fork, 0) != hipSuccess) FAIL("side stream fork");
      CK(run_conv(c, c->side, x0.p, x0.C, x1 ? x1->p : nullptr, C1, B, H, Wd, *ws, 1, side_out, os, c->zero_page));
      if (hipEventRecord(c->ev_join, c->side) != hipSuccess) FAIL("side stream join");
    }
  }
  Act n1 = alloc_act(c, B, H, Wd, Cin); if (!n1.p) return -1;
  GETV(g1, pre + "norm1.weight"); GETV(b1, pre + "norm1.bias");
  const bf16_t* n1p = n1.p;
  if (!x1 && x0.normed && x0.normed_gamma == g1) n1p = x0.normed;      // the producer's slab pass already applied this very norm
  else CK(run_gn(c, st, x0.p, x0.C, x1 ? x1->p : nullptr, C1, B, HW, g1, b1, groups, eps, 1, n1.p, &x0, x1));
  Act h = alloc_act(c, B, H, Wd, Cout, true); if (!h.p) return -1;
  GETW(w1, pre + "conv1.weight"); GETV(cb1, pre + "conv1.bias");
  GemmOpt o1; o1.bias = cb1; o1.out_act = &h;
  if (has_temb) {
    auto it = c->tproj_off.find(pre);
    if (it == c->tproj_off.end()) FAIL("no time_emb_proj for %s", pre.c_str());
    o1.rowadd = c->tproj_cur + it->second; o1.rowadd_ld = c->tproj_cur_ld;
  }
  Act n2 = alloc_act(c, B, H, Wd, Cout); if (!n2.p) return -1;
  GETV(g2, pre + "norm2.weight"); GETV(b2, pre + "norm2.bias");
  // conv1's output is read by norm2 and by nothing else: a split-K launch's slab-sum pass normalises straight into n2 (h is not written)
  int gn_done = 0;
  o1.gn_gamma = g2; o1.gn_beta = b2; o1.gn_y = n2.p; o1.gn_groups = groups; o1.gn_eps = eps; o1.gn_silu = 1; o1.gn_keep_out = 0; o1.gn_fused = &gn_done;
  CK(run_conv(c, st, n1p, Cin, nullptr, 0, B, H, Wd, *w1, 3, h.p, o1, c->zero_page));
  if (!gn_done) CK(run_gn(c, st, h.p, Cout, nullptr, 0, B, HW, g2, b2, groups, eps, 1, n2.p, &h));
  const bf16_t* res = x0.p;
  // conv_shortcut as extra K of conv2 (igemm_halo.h shortcut loop): one launch, the shortcut's output never exists -- where conv2 is an unsplit row-halo launch
  bool fuse_sc = false;
  if (has_sc && !side_out && (c->opt_sc_fuse & (H == 8 && Wd == 8 ? 2 : 1)) && c->W.count(pre + "conv2.sc")) {
    GETW(w2q, pre + "conv2.sc");
    int can = 0; GemmOpt q; q.can_fuse_sc = &can; q.sc0 = x0.p; q.sc_C0 = x0.C; q.sc1 = x1 ? x1->p : nullptr; q.sc_C1 = C1;
    q.out_act = &out;                                    // the launch below as it will run (its output leaves GroupNorm partial sums)
    CK(run_conv(c, st, n2.p, Cout, nullptr, 0, B, H, Wd, *w2q, 3, nullptr, q, c->zero_page));
    fuse_sc = can != 0;
  }
  if (fuse_sc) {
    GETW(w2s, pre + "conv2.sc"); GETV(cbs, pre + "conv2.sc.bias");
    GemmOpt o2; o2.bias = cbs; o2.out_act = &out; o2.sc0 = x0.p; o2.sc_C0 = x0.C; o2.sc1 = x1 ? x1->p : nullptr; o2.sc_C1 = C1;
    int gn2_done = 0;
    if (out_normed) {      // (as below: a split-K launch's slab pass also applies the GroupNorm that reads this output next)
      o2.gn_gamma = next->gamma; o2.gn_beta = next->beta; o2.gn_y = out_normed; o2.gn_groups = groups; o2.gn_eps = next->eps; o2.gn_silu = next->silu;
      o2.gn_keep_out = 1; o2.gn_fused = &gn2_done;
    }
    CK(run_conv(c, st, n2.p, Cout, nullptr, 0, B, H, Wd, *w2s, 3, out.p, o2, c->zero_page));
    if (gn2_done) { out.normed = out_normed; out.normed_gamma = next->gamma; }
    c->arena.release(mk);
    return 0;
  }
  if (side_out) {
    if (hipStreamWaitEvent(st, c->ev_join, 0) != hipSuccess) FAIL("side stream wait");
    res = side_out;
  } else if (has_sc) {
    GETW(ws, pre + "conv_shortcut.weight"); GETV(bs, pre + "conv_shortcut.bias");
    GemmOpt os; os.bias = bs;
    CK(run_conv(c, st, x0.p, x0.C, x1 ? x1->p : nullptr, C1, B, H, Wd, *ws, 1, h.p, os, c->zero_page));  // h is free again
    res = h.p;
  } else if (x1 || x0.C != Cout) {
    FAIL("resnet %s: channel change %d->%d without conv_shortcut", pre.c_str(), Cin, Cout);
  }
  GETW(w2, pre + "conv2.weight"); GETV(cb2, pre + "conv2.bias");
  GemmOpt o2; o2.bias = cb2; o2.residual = res; o2.out_act = &out;
  int gn2_done = 0;
  if (out_normed) {      // the next GroupNorm reads this output alone: a split-K launch's slab pass writes the output AND its normalised copy
    o2.gn_gamma = next->gamma; o2.gn_beta = next->beta; o2.gn_y = out_normed; o2.gn_groups = groups; o2.gn_eps = next->eps; o2.gn_silu = next->silu;
    o2.gn_keep_out = 1; o2.gn_fused = &gn2_done;
  }
  CK(run_conv(c, st, n2.p, Cout, nullptr, 0, B, H, Wd, *w2, 3, out.p, o2, c->zero_page));
  if (gn2_done) { out.normed = out_normed; out.normed_gamma = next->gamma; }
  c->arena.release(mk);
  return 0;
}

static int run_attention(agd_ctx* c, hipStream_t st, int cls, AttnP& a) {
  const double fl = 4.0 * a.B * a.H * (double)a.Nq * a.Nk * a.D;
  // q read + o written + k/v read once per (batch, head); the recorder's read-modify-write of its fp32 rows on top
  double by = 2.0 * a.B * a.H * (double)a.D * (2.0 * a.Nq + 2.0 * a.Nk);
  if (a.record_mode == 1) by += 8.0 * (a.B - a.rec_b0) * a.H * (double)a.rec_T * a.Nq;
  else if (a.record_mode == 2 || a.record_mode == 3) by += 8.0 * (a.B - a.rec_b0) * (double)a.rec_T * a.Nq;
  ProfScope ps(c, st, cls, fl, by);
  return launch_attention(a, st);
}

// cross-attention attn2 body shared by the UNet walk and the agd_cross_attn seam
static int cross_attention(agd_ctx* c, hipStream_t st, XLayer& xl, const bf16_t* q, int B2, int N, bf16_t* out, bool record,
                           const float* mask = nullptr) {
  const int C = xl.C, D = C / xl.heads, T = c->ctx_T;
  AttnP a{}; a.q = q; a.k = xl.kv; a.v = xl.kv + C; a.o = out;
  a.ldq = C; a.ldk = 2 * C; a.ldv = 2 * C; a.ldo = C;
  a.sq = (long long)N * C; a.sk = (long long)T * 2 * C; a.sv = a.sk; a.so = a.sq;
  a.B = B2; a.H = xl.heads; a.D = D; a.Nq = N; a.Nk = T; a.scale = 1.0f / sqrtf((float)D);
  a.record_mode = 0; a.mask = mask;
  const int side = (int)lrintf(sqrtf((float)N));
  bool hook_call = false;
  if (record && c->rec_mode == 1 && !xl.mid && xl.acc) {
    // daam: factor = latent_side / side; recorded iff factor != 8 (mid block); conditional half only
    if (c->rec_L / side != 8 && side == xl.acc_side && B2 / 2 == c->rec_B) {
      a.rec_b0 = B2 / 2; a.rec = xl.acc; a.rec_T = c->rec_T;
      a.rec_head_stride = (long long)c->rec_T * N; a.rec_img_stride = a.rec_head_stride * xl.acc_heads;
      if (xl.acc_heads < xl.heads) {                 // latent-resolution layer: head-group sums (attention.hip RECORD 2)
        if (mask) FAIL("attention_mask together with daam recording at latent resolution is not supported");
        a.record_mode = 3; a.rec_hpb = xl.heads / xl.acc_heads;
      } else a.record_mode = 1;
    }
  } else if (record && c->rec_mode == 2 && c->hook_scratch) {
    const int b0 = c->rec_is_train ? 0 : B2 / 2;
    if (B2 - b0 == c->hook_Bp) {
      if (T != c->hook_T || side > c->rec_L) FAIL("hook recorder was reset for %d tokens / latent side %d but this call has %d tokens / side %d: call clear() (agd_record_reset) after changing the context", c->hook_T, c->rec_L, T, side);
      // per-head probabilities of this call (plain read-modify-write rows, like the DAAM accumulators), then an ORDERED
      // head mean (hook.py:55): reproducible run to run, unlike float atomics across the head workgroups
      const size_t nh = (size_t)c->hook_Bp * xl.heads * T * N;
      CK(c->hook_headsb.ensure(nh * 4));
      if (hipMemsetAsync(c->hook_headsb.p, 0, nh * 4, st) != hipSuccess) FAIL("memset hook heads");
      a.record_mode = 1; a.rec_b0 = b0; a.rec = c->hook_headsb.as<float>(); a.rec_T = T;
      a.rec_head_stride = (long long)T * N; a.rec_img_stride = a.rec_head_stride * xl.heads;
      hook_call = true;
    }
  }
  CK(run_attention(c, st, PC_ATTN_CROSS, a));
  if (hook_call) {
    ProfScope ps(c, st, PC_HEAT, 0);
    CK(launch_hook_headmean(c->hook_headsb.as<float>(), c->hook_Bp, xl.heads, T, N, c->hook_scratch, st));
    if (c->rec_is_train) {
      // hook.py:110-112 `self.cross_attn_maps.append(maps)`: training reads every per-call map (finetune_sd_token.py:1043-1045),
      // so they are kept (device memory, grown geometrically, earlier maps carried over)
      const size_t bytes = (size_t)c->hook_Bp * T * N * 4, need = c->hook_store_used + bytes;
      if (need > c->hook_storeb.cap) {
        DBuf bigger; CK(bigger.ensure(need * 2 > ((size_t)64 << 20) ? need * 2 : ((size_t)64 << 20)));
        if (c->hook_store_used && hipMemcpyAsync(bigger.p, c->hook_storeb.p, c->hook_store_used, hipMemcpyDeviceToDevice, st) != hipSuccess) FAIL("hook store copy");
        hipStreamSynchronize(st);
        c->hook_storeb.release(); c->hook_storeb = bigger;
      }
      if (hipMemcpyAsync((char*)c->hook_storeb.p + c->hook_store_used, c->hook_scratch, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) FAIL("hook store");
      c->hook_recs.push_back({c->hook_store_used, c->hook_Bp, T, N});
      c->hook_store_used = need;
    }
    CK(launch_hook_accum(c->hook_scratch, c->hook_Bp, T, side, c->rec_L, c->hook_sum, st));
    c->hook_count++;
  }
  return 0;
}

// Transformer2DModel with one BasicTransformerBlock
// the second half of a [2B'] activation := its first half (CFG: both halves saw identical inputs so far)
static int dup_half(agd_ctx* c, hipStream_t st, bf16_t* p, long long half_elems) {
  ProfScope ps(c, st, PC_ELEM, 0);
  if (hipMemcpyAsync(p + half_elems, p, (size_t)half_elems * 2, hipMemcpyDeviceToDevice, st) != hipSuccess) FAIL("dup_half copy failed");
  return 0;
}

// dup != 0 (first transformer of a CFG forward, agd_denoise only): x holds B' = batch/2 images whose unconditional and
// conditional rows are still IDENTICAL (same latents, same timestep; the text context enters at attn2).  GroupNorm,
// proj_in, norm1 and the self-attention run once on B' rows; the result is duplicated right before the first
// cross-attention and the block returns 2B' rows.  Bit-identical to running both halves (every op here is
// row- or image-local), at half the cost for the most expensive attention call of the forward.
static int transformer(agd_ctx* c, hipStream_t st, const std::string& pre, const Act& x, int heads, int groups, Act& out, int dup = 0) {
  const int Bs = x.B;                                  // batch of the shared part
  int B = dup ? 2 * x.B : x.B;
  const int HW = x.H * x.W, C = x.C;
  int M = Bs * HW;                                     // rows until the duplication point, B*HW after it
  out = alloc_act(c, B, x.H, x.W, C, true); if (!out.p) return -1;
  const size_t mk = c->arena.mark();
  const std::string t = pre + "transformer_blocks.0.";
  Act n = alloc_act(c, B, x.H, x.W, C); if (!n.p) return -1;
  GETV(gg, pre + "norm.weight"); GETV(gb, pre + "norm.bias");
  // The transformer's GroupNorm has no activation behind it: where the producer of x left its per-tile channel sums, the norm is folded
  // into per-image proj_in matrices (norm.hip gn_fold_weight_kernel) and proj_in reads the raw x -- no read + write of the activation by
  // a GroupNorm kernel.  Default C <= 320 (the 64 x 64 maps: the fold launch takes 8.5 us against the 17 us apply pass; in situ 522.4 -> 521.3 ms
  // per batch, tools/ab_option.py); at C = 640 the fold (16.8 us, 6.5 MB of matrices) costs more than the 10.8 us pass it replaces.
  const bool gfold = c->opt_gn_proj_fold && c->opt_gn_fused && C <= (c->opt_gn_proj_fold >= 2 ? 640 : 320) && x.cpart && x.cpart_bm > 0 && HW % 128 == 0 && HW % x.cpart_bm == 0;
  // C = 640 (the 32 x 32 maps): no fold, but the qkv chain kernel (tblock.hip) takes the raw rows and applies the GroupNorm itself -- the apply launch disappears
  // (bit 8, default off: measured 71 us against 65 for the three launches it replaces -- 128 workgroups each streaming 3.3 MB of weights)
  const bool qkv640 = !gfold && C == 640 && (c->opt_tb_fuse & 16) && (c->opt_tb_fuse & 256) && (c->opt_tb_fuse & 128) && c->opt_gn_fused && x.cpart && x.cpart_bm > 0 &&
                      HW % 64 == 0 && HW % x.cpart_bm == 0 && groups <= 32 && !(x.normed && x.normed_gamma == gg) && !dup &&
                      c->W.count(pre + "proj_in.frag") && c->W.count(t + "attn1.qkv.frag");
  if (!gfold && !qkv640) {
    if (x.normed && x.normed_gamma == gg) n.p = x.normed;          // the producing resnet's slab pass already applied this norm
    else CK(run_gn(c, st, x.p, C, nullptr, 0, Bs, HW, gg, gb, groups, 1e-6f, 0, n.p, &x));
  }
  Act h = alloc_act(c, B, x.H, x.W, C); if (!h.p) return -1;
  Act ln = n;  // reuse (only the unfolded path normalises into it)
  bf16_t* qkv = (bf16_t*)c->arena.alloc((size_t)B * HW * 3 * C * 2); if (!qkv) return -1;
  bf16_t* att = (bf16_t*)c->arena.alloc((size_t)B * HW * C * 2); if (!att) return -1;
  const bf16_t* xres = x.p;                            // residual of proj_out
  // LayerNorm folded into the GEMMs around it (opt "ln_fold"): the GEMM that writes h also emits per-row (sum, sum of
  // squares) of its bf16 outputs per N tile; the GEMM that would read LayerNorm(h) reads h itself with W diag(gamma) and
  // finishes  rstd (acc - mean colsum) + (bias + W beta)  in its epilogue.  The three LayerNorm launches (and their
  // read + write of the activation) per block disappear; h is still rounded to bf16 exactly once.
  const bool fold = c->opt_ln_fold == 1 || (c->opt_ln_fold == 2 && C <= 320) || (c->opt_ln_fold == 3 && C <= 640);
  const float lneps = 1e-5f;
  float* stats = nullptr; int slots = 0;               // row statistics of the current h
  // h_out = A . W^T (+ bias, + residual): writes h and, when folding, its row statistics
  // (shaped: launched as [images][H][W] instead of one row of M pixels -- the per-image forms of the epilogue need the image of a row)
  auto produce = [&](const bf16_t* A, int K, const WMat& w, GemmOpt o, bf16_t* hout, bool shaped = false, bool want_stats = true) -> int {
    const int gb_ = shaped ? M / HW : 1, gh_ = shaped ? x.H : 1, gw_ = shaped ? x.W : M;
    stats = nullptr; slots = 0;
    if (fold && want_stats) {
      int cfg[3] = {0, 0, 0}; GemmOpt qo = o; qo.query_cfg = cfg; qo.want_rowstat = 1;
      CK(run_conv(c, st, A, K, nullptr, 0, gb_, gh_, gw_, w, 1, hout, qo, c->zero_page));
      slots = (w.N + cfg[1] - 1) / cfg[1];
      stats = (float*)c->arena.alloc((size_t)B * HW * slots * 2 * sizeof(float)); if (!stats) return -1;   // B*HW rows: room for the CFG duplicate
      o.rowstat_out = stats; o.rowstat_slots = slots;
    }
    return run_conv(c, st, A, K, nullptr, 0, gb_, gh_, gw_, w, 1, hout, o, c->zero_page);
  };
  // out = LayerNorm(h) . W^T (+ bias) [GEGLU]: folded, or the LayerNorm kernel followed by the plain GEMM
  auto consume = [&](const std::string& lnkey, const std::string& wkey, const float* bias, int geglu, bf16_t* outp) -> int {
    if (fold) {
      const std::string k = wkey + ".lnfold";
      GETW(wf, k); GETV(cs, k + ".cs"); GETV(bf, k + ".bias");
      GemmOpt o; o.bias = bf; o.geglu = geglu; o.ln_stats = stats; o.ln_slots = slots; o.ln_cs = cs; o.ln_invC = 1.0f / (float)C; o.ln_eps = lneps;
      if (geglu && C == 1280 && M >= 1024 && M <= 4096 && (c->opt_wreg & 1) && wf->wfrag) o.wreg = 2;
      return run_conv(c, st, h.p, C, nullptr, 0, 1, 1, M, *wf, 1, outp, o, c->zero_page);
    }
    GETV(g, lnkey + ".weight"); GETV(b, lnkey + ".bias");
    { ProfScope ps(c, st, PC_LN, 0, 4.0 * M * (double)C); CK(launch_layernorm(h.p, ln.p, g, b, M, C, lneps, st)); }
    GETW(w, wkey); GemmOpt o; o.bias = bias; o.geglu = geglu;
    return run_conv(c, st, ln.p, C, nullptr, 0, 1, 1, M, *w, 1, outp, o, c->zero_page);
  };
  bool qkv_done = false;
  { GETW(w, pre + "proj_in.weight"); GETV(b, pre + "proj_in.bias");
    if (gfold) {
      if (w->taps != 1 || w->Cpad != C) FAIL("gn_proj_fold: proj_in weight [N=%d taps=%d Cpad=%d] is not a 1x1 over %d channels", w->N, w->taps, w->Cpad, C);
      const bool qkv_fuse = (c->opt_tb_fuse & 16) && C == 320 && w->N == C && c->W.count(t + "attn1.qkv.frag");
      // bit 7: the qkv chain kernel applies the GroupNorm itself (statistics from the partial sums, rows normalised in its LDS panel): no fold launch
      const bool gn_inside = qkv_fuse && (c->opt_tb_fuse & 128) && groups <= 32 && c->W.count(pre + "proj_in.frag");
      bf16_t* wb = nullptr; float* radd = nullptr;
      if (!gn_inside) {
        wb = (bf16_t*)c->arena.alloc((size_t)Bs * w->N * C * 2);
        radd = (float*)c->arena.alloc((size_t)Bs * w->N * sizeof(float));
        if (!wb || !radd) return -1;
        ProfScope ps(c, st, PC_GN, 0, 2.0 * Bs * (double)w->N * C * 2);
        CK(launch_gn_fold_weight(x.cpart, x.cpart_bm, Bs, HW, C, groups, 1e-6f, gg, gb, w->w, b, w->N, wb, radd, st, qkv_fuse ? C / 64 : 0));
      }
      if (qkv_fuse) {      // proj_in -> h -> norm1 -> q / k / v in one launch (tblock.hip); norm1's statistics come from the rows themselves
        GETW(fqkv, t + "attn1.qkv.frag"); GETV(g1, t + "norm1.weight"); GETV(b1_, t + "norm1.bias");
        QkvChainP qp{}; qp.x = x.p; qp.wbf = wb; qp.wb_stride = (long long)w->N * C; qp.rowadd = radd; qp.rowadd_stride = C; qp.h = h.p; qp.gamma = g1; qp.beta = b1_; qp.ln_eps = lneps;
        qp.wqkvf = fqkv->w; qp.qkv = qkv; qp.M = M; qp.HW = HW;
        qp.rows64 = (c->opt_tb_fuse & 2048) && HW % 64 == 0 ? 1 : 0;
        qp.sched2 = (c->opt_tb_fuse & 4096) ? 1 : 0;
        if (gn_inside) {
          GETW(fpi, pre + "proj_in.frag");
          if (!b) FAIL("proj_in without bias");
          qp.wbf = fpi->w; qp.wb_stride = 0; qp.rowadd = b; qp.rowadd_stride = 0;
          qp.gn_part = x.cpart; qp.gn_bm = x.cpart_bm; qp.gn_groups = groups; qp.gn_eps = 1e-6f; qp.gn_gamma = gg; qp.gn_beta = gb;
        }
        stats = nullptr; slots = 0;
        ProfScope ps(c, st, PC_GEMM, 8.0 * M * (double)C * C, 2.0 * M * (double)C * 5.0 + 2.0 * (Bs + 3.0) * C * (double)C);
        CK(launch_qkv_chain(qp, C, st));
        qkv_done = true;
      } else {
      WMat wi = *w; wi.w = wb;
      GemmOpt o; o.rowadd = radd; o.rowadd_ld = w->N; o.w_per_image = 1;
      CK(produce(x.p, C, wi, o, h.p, true));
      }
    } else if (qkv640) {
      if (w->taps != 1 || w->Cpad != C || w->N != C || !b) FAIL("qkv chain: proj_in weight [N=%d taps=%d Cpad=%d] is not a biased 1x1 over %d channels", w->N, w->taps, w->Cpad, C);
      GETW(fpi, pre + "proj_in.frag"); GETW(fqkv, t + "attn1.qkv.frag"); GETV(g1, t + "norm1.weight"); GETV(b1_, t + "norm1.bias");
      QkvChainP qp{}; qp.x = x.p; qp.wbf = fpi->w; qp.wb_stride = 0; qp.rowadd = b; qp.rowadd_stride = 0; qp.h = h.p; qp.gamma = g1; qp.beta = b1_; qp.ln_eps = lneps;
      qp.wqkvf = fqkv->w; qp.qkv = qkv; qp.M = M; qp.HW = HW;
      qp.gn_part = x.cpart; qp.gn_bm = x.cpart_bm; qp.gn_groups = groups; qp.gn_eps = 1e-6f; qp.gn_gamma = gg; qp.gn_beta = gb;
      qp.sched2 = (c->opt_tb_fuse & 4096) ? 1 : 0;
      stats = nullptr; slots = 0;
      ProfScope ps(c, st, PC_GEMM, 8.0 * M * (double)C * C, 2.0 * M * (double)C * 5.0 + 2.0 * 4.0 * C * (double)C);
      CK(launch_qkv_chain(qp, C, st));
      qkv_done = true;
    } else {
      GemmOpt o; o.bias = b;
      if (C == 640 && (c->opt_wreg & 2)) o.wreg = 1;
      CK(produce(n.p, C, *w, o, h.p));
    }
  }
  // fused row-panel kernels of this block (tblock.hip), where their shape is built: C = 320, 8 heads, <= 96 keys, whole 128-row tiles per
  // image; the hook.py recorder (per-head maps of every call) keeps the kernel chain
  const bool ff_fused = (c->opt_tb_fuse & 1) && fold && C == 320 && c->W.count(t + "ff.w1.frag");
  bool xpre_ready = false;                               // the pre-multiplied attn2 form is ready for this block (agd_set_context built its products)
  bool xpre_rec = false;                                 // ... and it records into the DAAM accumulators (daam's rule, as cross_attention())
  { auto itx = c->xl_idx.find(t + "attn2");              // the ONE predicate of that form: the chain kernel steps aside exactly where it holds
    if (itx != c->xl_idx.end()) {
      const XLayer& xq = c->xl[itx->second];
      xpre_rec = c->rec_mode == 1 && !xq.mid && xq.acc && c->rec_L / x.H != 8 && x.H == xq.acc_side && x.W == x.H && B / 2 == c->rec_B;
      xpre_ready = xq.pm_ready && c->opt_xpre && fold && HW % 64 == 0 && c->rec_mode != 2 && c->ctx_T <= XATTN_TP && !dup && (!xpre_rec || xq.acc_heads == xq.heads);
    } }
  const bool chain_fuse = (c->opt_tb_fuse & 2) && (C == 320 || (C == 640 && (c->opt_tb_fuse & 32))) && heads == 8 && HW % (C == 320 ? 128 : 64) == 0 && c->ctx_T <= 96 &&
                          c->rec_mode != 2 && c->W.count(t + "attn2.to_q.frag") && !xpre_ready;
  // CFG-shared prefix with the fused kernels behind it: the duplication of the B' rows happens INSIDE them (the chain reads input row m % M', the feed-forward's
  // proj_out stage adds block-input row m % M'): no copy launches, and attn1.to_out joins the chain here too
  const bool lazy_dup = dup && (c->opt_tb_fuse & 64) && chain_fuse && C == 320 && ff_fused && (c->opt_tb_fuse & 8) && c->W.count(pre + "proj_out.frag");
  const bool chain_pre = chain_fuse && (c->opt_tb_fuse & 4) && (!dup || lazy_dup) && c->W.count(t + "attn1.to_out.frag");     // attn1.to_out + residual inside the chain launch
  // --- self attention ---
  { if (!qkv_done) CK(consume(t + "norm1", t + "attn1.qkv", nullptr, 0, qkv));
    AttnP a{}; a.q = qkv; a.k = qkv + C; a.v = qkv + 2 * C; a.o = att;
    a.ldq = a.ldk = a.ldv = 3 * C; a.ldo = C; a.sq = a.sk = a.sv = (long long)HW * 3 * C; a.so = (long long)HW * C;
    a.B = Bs; a.H = heads; a.D = C / heads; a.Nq = HW; a.Nk = HW; a.scale = 1.0f / sqrtf((float)(C / heads));
    CK(run_attention(c, st, PC_ATTN_SELF, a));
    GETW(wo, t + "attn1.to_out.0.weight"); GETV(bo, t + "attn1.to_out.0.bias");
    GemmOpt oo; oo.bias = bo; oo.residual = h.p;
    // (the fused attn2 chain takes norm2's statistics from the rows themselves; with bit 2 it also starts at this very GEMM)
    if (!chain_pre) CK(produce(att, C, *wo, oo, h.p, false, !chain_fuse)); }
  const int Mshared = M;                               // rows of the shared part
  if (dup && lazy_dup) M = B * HW;
  else if (dup) {                                      // the halves diverge from here on (text context)
    CK(dup_half(c, st, h.p, (long long)M * C));
    if (fold && stats) { ProfScope ps(c, st, PC_ELEM, 0);
      if (hipMemcpyAsync(stats + (size_t)M * slots * 2, stats, (size_t)M * slots * 2 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) FAIL("dup stats copy failed"); }
    Act x2 = alloc_act(c, B, x.H, x.W, C); if (!x2.p) return -1;
    { ProfScope ps(c, st, PC_ELEM, 0);
      if (hipMemcpyAsync(x2.p, x.p, (size_t)M * C * 2, hipMemcpyDeviceToDevice, st) != hipSuccess) FAIL("dup copy failed"); }
    CK(dup_half(c, st, x2.p, (long long)M * C));
    xres = x2.p;
    M = B * HW;
  }
  // --- cross attention (the processor seam) ---
  bool chain_done = false;
  { auto it = c->xl_idx.find(t + "attn2");
    if (it == c->xl_idx.end()) FAIL("cross-attn layer %s not registered", (t + "attn2").c_str());
    if (c->ctx_B2 != B) FAIL("context batch %d != unet batch %d (call agd_set_context)", c->ctx_B2, B);
    XLayer& xl = c->xl[it->second];
    // one launch for norm2 -> to_q -> attention (+ DAAM record) -> to_out + residual (tblock.hip)
    if (chain_fuse) {
      GETW(fq, t + "attn2.to_q.frag"); GETW(fo, t + "attn2.to_out.frag");
      GETV(g2, t + "norm2.weight"); GETV(b2_, t + "norm2.bias"); GETV(bo, t + "attn2.to_out.0.bias");
      const int T = c->ctx_T, D = C / heads;
      AttnChainP ap{}; ap.h = h.p; ap.out = h.p; ap.gamma = g2; ap.beta = b2_; ap.ln_eps = lneps; ap.wqf = fq->w; ap.wof = fo->w; ap.bo = bo;
      ap.kv = xl.kv; ap.ldkv = 2 * C; ap.skv = (long long)T * 2 * C; ap.M = M; ap.HW = HW; ap.T = T; ap.scale = 1.0f / sqrtf((float)D);
      double rec_bytes = 0;
      if (c->rec_mode == 1 && !xl.mid && xl.acc && c->rec_L / x.H != 8 && x.H == xl.acc_side && x.W == x.H && B / 2 == c->rec_B) {   // daam's rule, as cross_attention()
        ap.record = 1; ap.rec = xl.acc; ap.rec_b0 = B / 2; ap.rec_T = c->rec_T; ap.rec_hpb = heads / xl.acc_heads;
        ap.rec_head_stride = (long long)c->rec_T * HW; ap.rec_img_stride = ap.rec_head_stride * xl.acc_heads;
        rec_bytes = 8.0 * (B - ap.rec_b0) * xl.acc_heads * (double)c->rec_T * HW;
      }
      if (chain_pre) {                                   // h1 = attn1.to_out(att) + h is computed (and stored) by the chain itself: into a second buffer
        GETW(f1o, t + "attn1.to_out.frag"); GETV(bo1, t + "attn1.to_out.0.bias");
        bf16_t* h2 = (bf16_t*)c->arena.alloc((size_t)M * C * 2); if (!h2) return -1;
        ap.o1 = att; ap.wo1f = f1o->w; ap.bo1 = bo1; ap.out = h2;
      } else if (lazy_dup) { bf16_t* h2 = (bf16_t*)c->arena.alloc((size_t)M * C * 2); if (!h2) return -1; ap.out = h2; }
      if (lazy_dup) ap.src_rows = Mshared;
      ap.rows32 = (c->opt_tb_fuse & 1024) ? 1 : 0;       // bit 10: 32-row panels for the C = 640 chain where 64-row panels fill half the chip
      if (fold && !ff_fused) {                           // the GEGLU consumer of the LayerNorm fold reads one slot of row statistics
        slots = 1; stats = (float*)c->arena.alloc((size_t)M * 2 * sizeof(float)); if (!stats) return -1;
        ap.rowstat_out = stats;
      }
      ProfScope ps(c, st, PC_ATTN_CROSS, (chain_pre ? 6.0 : 4.0) * M * (double)C * C + 4.0 * B * heads * (double)HW * T * D,
                   (chain_pre ? 6.0 : 4.0) * M * (double)C + (chain_pre ? 6.0 : 4.0) * C * (double)C + 4.0 * B * (double)T * C + rec_bytes);
      CK(launch_attn_chain(ap, C, heads, st));
      if (ap.out != ap.h) h.p = ap.out;                  // the residual stream continues in the second buffer
      chain_done = true;
    }
  }
  // attn2 of the C = 1280 blocks against the per-image pre-multiplied context matrices (xattn_pre.hip): S GEMM + softmax + recorder, then the output GEMM
  // with per-image weights -- two launches instead of to_q, the attention kernel and to_out; the hook.py recorder (per-call head means) keeps the kernel chain
  bool xpre_done = false;
  if (!chain_done) {
    auto it = c->xl_idx.find(t + "attn2"); XLayer& xl = c->xl[it->second];
    const bool rec_daam = xpre_rec;
    if (xpre_ready && stats && slots > 0) {               // (norm2's row statistics come from attn1.to_out's epilogue: the chain kernel was not chosen, so it wrote them)
      GETV(bo, t + "attn2.to_out.0.bias");
      const int HT = heads * XATTN_TP;
      bf16_t* P = (bf16_t*)c->arena.alloc((size_t)M * HT * 2); if (!P) return -1;
      XattnSP sp{}; sp.x = h.p; sp.ln_stats = stats; sp.ln_slots = slots; sp.ln_invC = 1.0f / (float)C; sp.ln_eps = lneps;
      sp.kpp = xl.pm_kpp; sp.kcs = xl.pm_kcs; sp.kbs = xl.pm_kbs; sp.P = P; sp.M = M; sp.HW = HW; sp.C = C; sp.H = heads; sp.T = c->ctx_T;
      double rec_bytes = 0;
      if (rec_daam) {
        sp.rec = xl.acc; sp.rec_b0 = B / 2; sp.rec_T = c->rec_T; sp.rec_head_stride = (long long)c->rec_T * HW; sp.rec_img_stride = sp.rec_head_stride * xl.acc_heads;
        rec_bytes = 8.0 * (B - sp.rec_b0) * heads * (double)c->rec_T * HW;
      }
      { ProfScope ps(c, st, PC_ATTN_CROSS, 2.0 * M * (double)HT * C, 2.0 * M * (double)C + 2.0 * B * (double)HT * C + 2.0 * M * (double)HT + rec_bytes);
        CK(launch_xattn_s(sp, st)); }
      WMat wv; wv.w = xl.pm_vpp; wv.N = C; wv.Cin = HT; wv.Cpad = HT; wv.taps = 1;
      GemmOpt oo; oo.bias = bo; oo.residual = h.p; oo.w_per_image = 1;
      CK(produce(P, HT, wv, oo, h.p, true, !ff_fused));
      xpre_done = true;
    }
  }
  if (!chain_done && !xpre_done)
  { bf16_t* q = qkv;
    CK(consume(t + "norm2", t + "attn2.to_q.weight", nullptr, 0, q));
    auto it = c->xl_idx.find(t + "attn2");
    CK(cross_attention(c, st, c->xl[it->second], q, B, HW, att, true));
    GETW(wo, t + "attn2.to_out.0.weight"); GETV(bo, t + "attn2.to_out.0.bias");
    GemmOpt oo; oo.bias = bo; oo.residual = h.p;
    CK(produce(att, C, *wo, oo, h.p, false, !ff_fused)); }
  // --- GEGLU feed-forward ---
  bool proj_done = false;
  if (ff_fused) {
    // one launch: norm3 (folded) -> GEGLU -> ff.net.2 + residual; the 4C-wide hidden activation stays in LDS (tblock.hip)
    GETW(f1, t + "ff.w1.frag"); GETW(f2, t + "ff.w2.frag");
    const std::string k = t + "ff.net.0.proj.weight.lnfold";
    GETV(cs, k + ".cs"); GETV(bf, k + ".bias"); GETV(b2, t + "ff.net.2.bias");
    FFusedP fp{}; fp.h = h.p; fp.out = h.p; fp.w1f = f1->w; fp.cs1 = cs; fp.b1 = bf; fp.w2f = f2->w; fp.b2 = b2; fp.M = M; fp.ln_eps = lneps;
    if ((c->opt_tb_fuse & 8) && c->W.count(pre + "proj_out.frag")) {      // proj_out + residual (+ the next GroupNorm's partial sums) behind it, same launch
      GETW(fpw, pre + "proj_out.frag"); GETV(bp, pre + "proj_out.bias");
      fp.wpf = fpw->w; fp.bp = bp; fp.xres = xres; fp.pout = out.p;
      if ((c->opt_tb_fuse & 512) && c->W.count(t + "ff.w2p.frag")) {      // ff.net.2 and proj_out pre-multiplied (as ff_proj_fuse does for the other blocks): no intermediate h3
        GETW(f2p, t + "ff.w2p.frag"); GETV(bcp, pre + "ffproj.bias");
        fp.w2f = f2p->w; fp.bp = bcp; fp.premul = 1;
      }
      if (lazy_dup) fp.xres_rows = Mshared;             // xres is still the B'-row block input
      out.cpart_bm = 0;
      if (out.cpart && c->opt_gn_fused && HW % 128 == 0) { fp.colstat = out.cpart; out.cpart_bm = 128; }
      proj_done = true;
    }
    ProfScope ps(c, st, PC_GEMM, 2.0 * M * (double)C * (proj_done ? 13.0 : 12.0) * C, (proj_done ? 8.0 : 4.0) * M * (double)C + 2.0 * (proj_done ? 13.0 : 12.0) * C * (double)C);
    CK(launch_ff_fused(fp, C, st));
  } else
  { bf16_t* ff = (bf16_t*)c->arena.alloc((size_t)M * 4 * C * 2); if (!ff) return -1;
    GETV(b1, t + "ff.net.0.proj.bias");
    CK(consume(t + "norm3", t + "ff.net.0.proj.weight", b1, 1, ff));
    if (c->opt_ffproj && c->W.count(pre + "ffproj.weight")) {
      // proj_out(ff.net.2(g) + h) + x = [Wp W2 | Wp] . [g | h] + (Wp b2 + bp) + x: one launch over the channel concat of the hidden activation and the residual stream
      // with the matrix pre-multiplied at load time -- h3 is never formed, the proj_out launch (its 5 - 10 MB output pass and epilogue) disappears
      GETW(wc, pre + "ffproj.weight"); GETV(bc, pre + "ffproj.bias");
      GemmOpt o; o.bias = bc; o.residual = xres; o.out_act = &out; o.rows_per_image = HW;
      CK(run_conv(c, st, ff, 4 * C, h.p, C, 1, 1, M, *wc, 1, out.p, o, c->zero_page));
      proj_done = true;
    } else {
    GETW(w2, t + "ff.net.2.weight"); GETV(b2, t + "ff.net.2.bias");
    GemmOpt o2; o2.bias = b2; o2.residual = h.p;
    CK(run_conv(c, st, ff, 4 * C, nullptr, 0, 1, 1, M, *w2, 1, h.p, o2, c->zero_page)); } }
  if (!proj_done)
  { GETW(w, pre + "proj_out.weight"); GETV(b, pre + "proj_out.bias"); GemmOpt o; o.bias = b; o.residual = xres; o.out_act = &out; o.rows_per_image = HW;
    if (C == 640 && (c->opt_wreg & 2)) o.wreg = 1;
    CK(run_conv(c, st, h.p, C, nullptr, 0, 1, 1, M, *w, 1, out.p, o, c->zero_page)); }
  c->arena.release(mk);
  return 0;
}

// time embeddings of n <= 25 timesteps (inference: one timestep serves every batch row); scratch = n * 9 * dim floats,
// out = [n][tproj_total]: every resnet's time_emb_proj(silu(temb)) from one stacked matrix
static int time_embed(agd_ctx* c, hipStream_t st, const float* ts, int n, float* scratch, float* out) {
  const int dim = c->cfg.block_out_channels[0], td = dim * 4;
  ProfScope ps(c, st, PC_ELEM, 0);
  float* e0 = scratch; float* e1 = e0 + (size_t)n * dim; float* e2 = e1 + (size_t)n * td;
  for (int i = 0; i < n; ++i) CK(launch_timestep_embed(ts[i], e0 + (size_t)i * dim, dim, st));
  GETW(w1, "unet.time_embedding.linear_1.weight"); GETV(b1, "unet.time_embedding.linear_1.bias");
  GETW(w2, "unet.time_embedding.linear_2.weight"); GETV(b2, "unet.time_embedding.linear_2.bias");
  CK(launch_small_linear(e0, w1->w, b1, e1, n, td, w1->Cpad, 0, 1, st));     // linear_1 + SiLU
  CK(launch_small_linear(e1, w2->w, b2, e2, n, td, w2->Cpad, 0, 0, st));     // linear_2 -> temb
  CK(launch_small_linear(e2, c->tproj_all.w, c->tproj_bias, out, n, c->tproj_total, td, 1, 0, st));
  return 0;
}

// x: [B2][L*L][64] bf16 (latent channels zero-padded) -> eps [B2][L*L][out_channels] fp32 NHWC
// tproj_row: this timestep's time_emb_proj outputs when the caller computed them up front (agd_denoise), else nullptr
// cfg_shared: rows [0,B2/2) and [B2/2,B2) of xin are identical (agd_denoise): share everything ahead of the first attn2
// tproj_ld: 0 = tproj_row serves every image; tproj_total = tproj_row holds one row per image (per-sample timesteps, training)
static int unet_walk(agd_ctx* c, hipStream_t st, const bf16_t* xin, int B2, int L, float t, float* eps_out,
                     const float* tproj_row = nullptr, bool cfg_shared = false, int tproj_ld = 0) {
  const agd_config& g = c->cfg;
  const int nl = g.n_levels, G = g.norm_num_groups;
  const std::string u = "unet.";
  c->arena.release(0);
  c->tproj_cur_ld = tproj_row ? tproj_ld : 0;
  if (tproj_row) c->tproj_cur = tproj_row;
  else { CK(time_embed(c, st, &t, 1, c->temb_buf, c->tproj_out)); c->tproj_cur = c->tproj_out; }
  std::vector<Act> skips;
  const bool shared = cfg_shared && (B2 % 2) == 0 && g.down_cross[0] && c->opt_cfg_share;
  const int Bh = shared ? B2 / 2 : B2;
  Act h = alloc_act(c, B2, L, L, g.block_out_channels[0], true); if (!h.p) return -1;
  { GETW(w, u + "conv_in.weight"); GETV(b, u + "conv_in.bias"); GemmOpt o; o.bias = b; o.out_act = &h;
    CK(run_conv(c, st, xin, 64, nullptr, 0, Bh, L, L, *w, 3, h.p, o, c->zero_page)); }
  if (shared) {                                                             // the skip connection needs all B2 rows (and their partial sums)
    CK(dup_half(c, st, h.p, (long long)Bh * L * L * h.C));
    if (h.cpart_bm > 0) {
      const size_t nb = (size_t)((long long)Bh * L * L / h.cpart_bm) * h.C * 2 * sizeof(float);
      if (hipMemcpyAsync((char*)h.cpart + nb, h.cpart, nb, hipMemcpyDeviceToDevice, st) != hipSuccess) FAIL("dup partials copy failed");
    }
  }
  skips.push_back(h);
  for (int i = 0; i < nl; ++i) {
    const int co = g.block_out_channels[i];
    for (int j = 0; j < g.layers_per_block; ++j) {
      const bool first = shared && i == 0 && j == 0;
      Act hin = h; if (first) hin.B = Bh;
      // the GroupNorm that reads this resnet's output alone: the block's transformer, the next resnet of an attention-free level, or mid_block
      NextGn ng; std::string nk; 
      if (g.down_cross[i]) { nk = u + "down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j) + ".norm."; ng.eps = 1e-6f; ng.silu = 0; }
      else if (j + 1 < g.layers_per_block) { nk = u + "down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j + 1) + ".norm1."; ng.eps = 1e-5f; ng.silu = 1; }
      else if (i == nl - 1) { nk = u + "mid_block.resnets.0.norm1."; ng.eps = 1e-5f; ng.silu = 1; }
      if (!nk.empty()) { auto ig = c->V.find(nk + "weight"), ib = c->V.find(nk + "bias"); if (ig != c->V.end() && ib != c->V.end()) { ng.gamma = ig->second; ng.beta = ib->second; } }
      Act r; CK(resnet(c, st, u + "down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + ".", hin, nullptr, co, 1e-5f, true, G, r, first ? nullptr : &ng));
      h = r;
      if (g.down_cross[i]) {
        Act a; CK(transformer(c, st, u + "down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j) + ".", h, g.num_heads[i], G, a, first ? 1 : 0));
        h = a;
      }
      skips.push_back(h);
    }
    if (i != nl - 1) {
      const std::string k = u + "down_blocks." + std::to_string(i) + ".downsamplers.0.conv.";
      GETW(w, k + "weight"); GETV(b, k + "bias");
      Act d = alloc_act(c, B2, h.H / 2, h.W / 2, co, true); if (!d.p) return -1;
      GemmOpt o; o.bias = b; o.stride = 2; o.out_act = &d;
      CK(run_conv(c, st, h.p, co, nullptr, 0, B2, h.H, h.W, *w, 3, d.p, o, c->zero_page));
      h = d; skips.push_back(h);
    }
  }
  { const int cm = g.block_out_channels[nl - 1];
    NextGn ngm; { auto ig = c->V.find(u + "mid_block.attentions.0.norm.weight"), ib = c->V.find(u + "mid_block.attentions.0.norm.bias");
                  if (ig != c->V.end() && ib != c->V.end()) { ngm.gamma = ig->second; ngm.beta = ib->second; ngm.eps = 1e-6f; ngm.silu = 0; } }
    Act r; CK(resnet(c, st, u + "mid_block.resnets.0.", h, nullptr, cm, 1e-5f, true, G, r, &ngm)); h = r;
    Act a; CK(transformer(c, st, u + "mid_block.attentions.0.", h, g.num_heads[nl - 1], G, a)); h = a;
    Act r2; CK(resnet(c, st, u + "mid_block.resnets.1.", h, nullptr, cm, 1e-5f, true, G, r2)); h = r2; }
  for (int i = 0; i < nl; ++i) {
    const int lvl = nl - 1 - i, co = g.block_out_channels[lvl];
    for (int j = 0; j < g.layers_per_block + 1; ++j) {
      Act sk = skips.back(); skips.pop_back();
      NextGn ngu;
      if (g.down_cross[lvl]) { const std::string nk = u + "up_blocks." + std::to_string(i) + ".attentions." + std::to_string(j) + ".norm.";
        auto ig = c->V.find(nk + "weight"), ib = c->V.find(nk + "bias"); if (ig != c->V.end() && ib != c->V.end()) { ngu.gamma = ig->second; ngu.beta = ib->second; ngu.eps = 1e-6f; ngu.silu = 0; } }
      Act r; CK(resnet(c, st, u + "up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + ".", h, &sk, co, 1e-5f, true, G, r, &ngu));
      h = r;
      if (g.down_cross[lvl]) {
        Act a; CK(transformer(c, st, u + "up_blocks." + std::to_string(i) + ".attentions." + std::to_string(j) + ".", h, g.num_heads[lvl], G, a));
        h = a;
      }
    }
    if (i != nl - 1) {
      const std::string k = u + "up_blocks." + std::to_string(i) + ".upsamplers.0.conv.";
      GETW(w, k + "weight"); GETV(b, k + "bias");
      Act d = alloc_act(c, B2, h.H * 2, h.W * 2, co, true); if (!d.p) return -1;
      if ((c->opt_ups4 & 1) && c->W.count(k + "phases") && (long long)B2 * h.H * h.W >= ((c->opt_ups4 & 4) ? 512 : 2048)) {
        // conv3x3(nearest2x(x)) = four 2x2 convs on x, one per output phase, with the taps that coincide pre-summed (misc.hip upsample_phase_weight_kernel): 4/9 of the MACs
        GETW(w4, k + "phases"); GETV(b4, k + "phases.bias");
        GemmOpt o; o.bias = b4; o.out_act = &d; o.ups4 = co; o.hout = h.H; o.wout = h.W; o.ldo = co; o.pad = 0;
        if (c->opt_ups4 & 8) o.p8 = -1;                    // (A/B: keep the phase convs on the 4-wave kernel)
        CK(run_conv(c, st, h.p, co, nullptr, 0, B2, h.H, h.W, *w4, 2, d.p, o, c->zero_page));
      } else {
      GemmOpt o; o.bias = b; o.up = 2; o.out_act = &d;
      CK(run_conv(c, st, h.p, co, nullptr, 0, B2, h.H, h.W, *w, 3, d.p, o, c->zero_page));
      }
      h = d;
    }
  }
  { Act n = alloc_act(c, B2, L, L, h.C); if (!n.p) return -1;
    GETV(gg, u + "conv_norm_out.weight"); GETV(gb, u + "conv_norm_out.bias");
    CK(run_gn(c, st, h.p, h.C, nullptr, 0, B2, L * L, gg, gb, G, 1e-5f, 1, n.p, &h));
    GETW(w, u + "conv_out.weight"); GETV(b, u + "conv_out.bias");
    GemmOpt o; o.bias = b; o.out_f32 = 1; o.ldo = g.out_channels;
    CK(run_conv(c, st, n.p, h.C, nullptr, 0, B2, L, L, *w, 3, eps_out, o, c->zero_page)); }
  return 0;
}

// AutoencoderKL mid-block attention: single head over N = L*L tokens, C channels, through batched GEMMs
static int vae_mid_attention(agd_ctx* c, hipStream_t st, const std::string& a, int G, int B, int L, int top, Act& h) {
    const int N = L * L, C = top, M = B * N;
    Act out = alloc_act(c, B, L, L, C, true); if (!out.p) return -1;
    const size_t mk = c->arena.mark();
    Act n = alloc_act(c, B, L, L, C); if (!n.p) return -1;
    GETV(gg, a + "group_norm.weight"); GETV(gb, a + "group_norm.bias");
    CK(run_gn(c, st, h.p, C, nullptr, 0, B, N, gg, gb, G, 1e-6f, 0, n.p, &h));
    bf16_t* q = (bf16_t*)c->arena.alloc((size_t)M * C * 2); bf16_t* k = (bf16_t*)c->arena.alloc((size_t)M * C * 2);
    bf16_t* vT = (bf16_t*)c->arena.alloc((size_t)M * C * 2); bf16_t* att = (bf16_t*)c->arena.alloc((size_t)M * C * 2);
    float* S = (float*)c->arena.alloc((size_t)B * N * N * 4); bf16_t* P = (bf16_t*)c->arena.alloc((size_t)B * N * N * 2);
    if (!q || !k || !vT || !att || !S || !P) return -1;
    GETW(wq, a + "to_q.weight"); GETV(bq, a + "to_q.bias"); GETW(wk, a + "to_k.weight"); GETV(bk, a + "to_k.bias");
    GETW(wv, a + "to_v.weight"); GETV(bv, a + "to_v.bias"); GETW(wo, a + "to_out.0.weight"); GETV(bo, a + "to_out.0.bias");
    { GemmOpt o; o.bias = bq; CK(run_conv(c, st, n.p, C, nullptr, 0, 1, 1, M, *wq, 1, q, o, c->zero_page)); }
    { GemmOpt o; o.bias = bk; CK(run_conv(c, st, n.p, C, nullptr, 0, 1, 1, M, *wk, 1, k, o, c->zero_page)); }
    { // V^T[b] = Wv . X[b]^T  -> [C][N], bias per output row
      IgemmP p{}; p.src0 = wv->w; p.C0 = C; p.Hin = 1; p.Win = C; p.Hout = 1; p.Wout = C; p.ksize = 1; p.stride = 1; p.up = 1;
      p.W = n.p; p.bias = bv; p.bias_mode = 2; p.out = vT; p.ldo = N; p.ldr = N; p.M = C; p.N = N; p.K = C; p.alpha = 1.f;
      p.batch = B; p.sA0 = 0; p.sW = (long long)N * C; p.sO = (long long)C * N; p.zero_page = c->zero_page;
      ProfScope ps(c, st, PC_GEMM, 2.0 * B * C * (double)N * C); CK(launch_igemm(p, st)); }
    { // S[b] = scale * Q[b] K[b]^T  (fp32)
      IgemmP p{}; p.src0 = q; p.C0 = C; p.Hin = 1; p.Win = N; p.Hout = 1; p.Wout = N; p.ksize = 1; p.stride = 1; p.up = 1;
      p.W = k; p.out = S; p.out_f32 = 1; p.ldo = N; p.ldr = N; p.M = N; p.N = N; p.K = C; p.alpha = 1.0f / sqrtf((float)C);
      p.batch = B; p.sA0 = (long long)N * C; p.sW = (long long)N * C; p.sO = (long long)N * N; p.zero_page = c->zero_page;
      ProfScope ps(c, st, PC_VAE_ATTN, 2.0 * B * N * (double)N * C); CK(launch_igemm(p, st)); }
    { ProfScope ps(c, st, PC_VAE_ATTN, 0); CK(launch_softmax_rows(S, P, B * N, N, st)); }
    { // O[b] = P[b] V[b]  via V^T as the [N=C][K=N] operand
      IgemmP p{}; p.src0 = P; p.C0 = N; p.Hin = 1; p.Win = N; p.Hout = 1; p.Wout = N; p.ksize = 1; p.stride = 1; p.up = 1;
      p.W = vT; p.out = att; p.ldo = C; p.ldr = C; p.M = N; p.N = C; p.K = N; p.alpha = 1.f;
      p.batch = B; p.sA0 = (long long)N * N; p.sW = (long long)C * N; p.sO = (long long)N * C; p.zero_page = c->zero_page;
      ProfScope ps(c, st, PC_VAE_ATTN, 2.0 * B * N * (double)N * C); CK(launch_igemm(p, st)); }
    { GemmOpt o; o.bias = bo; o.residual = h.p; o.out_act = &out; o.rows_per_image = N; CK(run_conv(c, st, att, C, nullptr, 0, 1, 1, M, *wo, 1, out.p, o, c->zero_page)); }
    c->arena.release(mk);
    h = out;
  return 0;
}

// AutoencoderKL.decode: z [B][L*L][64] bf16 (already divided by scaling factor) -> fp32 NHWC [B][8L*8L][ldo=4]
static int vae_walk(agd_ctx* c, hipStream_t st, const bf16_t* zin, int B, int L, float* img_out) {
  const agd_config& g = c->cfg;
  const int nl = g.vae_n_levels, G = g.vae_norm_num_groups;
  const std::string v = "vae.";
  c->arena.release(0);
  const int top = g.vae_block_out_channels[nl - 1];
  // post_quant_conv (1x1, 4->4) written into a zeroed 64-channel buffer so conv_in sees padded input
  bf16_t* pq = (bf16_t*)c->arena.alloc((size_t)B * L * L * 64 * 2); if (!pq) return -1;
  if (hipMemsetAsync(pq, 0, (size_t)B * L * L * 64 * 2, st) != hipSuccess) FAIL("memset pq");
  { GETW(w, v + "post_quant_conv.weight"); GETV(b, v + "post_quant_conv.bias"); GemmOpt o; o.bias = b; o.ldo = 64;
    CK(run_conv(c, st, zin, 64, nullptr, 0, B, L, L, *w, 1, pq, o, c->zero_page)); }
  Act h = alloc_act(c, B, L, L, top, true); if (!h.p) return -1;
  { GETW(w, v + "decoder.conv_in.weight"); GETV(b, v + "decoder.conv_in.bias"); GemmOpt o; o.bias = b; o.out_act = &h;
    CK(run_conv(c, st, pq, 64, nullptr, 0, B, L, L, *w, 3, h.p, o, c->zero_page)); }
  { Act r; CK(resnet(c, st, v + "decoder.mid_block.resnets.0.", h, nullptr, top, 1e-6f, false, G, r)); h = r; }
  CK(vae_mid_attention(c, st, v + "decoder.mid_block.attentions.0.", G, B, L, top, h));
  { Act r; CK(resnet(c, st, v + "decoder.mid_block.resnets.1.", h, nullptr, top, 1e-6f, false, G, r)); h = r; }
  for (int i = 0; i < nl; ++i) {
    const int co = g.vae_block_out_channels[nl - 1 - i];
    for (int j = 0; j < g.vae_layers_per_block + 1; ++j) {
      Act r; CK(resnet(c, st, v + "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + ".", h, nullptr, co, 1e-6f, false, G, r));
      h = r;
    }
    if (i != nl - 1) {
      const std::string k = v + "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv.";
      GETW(w, k + "weight"); GETV(b, k + "bias");
      Act d = alloc_act(c, B, h.H * 2, h.W * 2, co, true); if (!d.p) return -1;
      if ((c->opt_ups4 & 2) && c->W.count(k + "phases")) {              // four 2x2 phase convs on the un-upsampled map (as in the UNet walk)
        GETW(w4, k + "phases"); GETV(b4, k + "phases.bias");
        GemmOpt o; o.bias = b4; o.out_act = &d; o.ups4 = co; o.hout = h.H; o.wout = h.W; o.ldo = co; o.pad = 0;
        if (c->opt_ups4 & 8) o.p8 = -1;
        CK(run_conv(c, st, h.p, co, nullptr, 0, B, h.H, h.W, *w4, 2, d.p, o, c->zero_page));
      } else {
      GemmOpt o; o.bias = b; o.up = 2; o.out_act = &d;
      CK(run_conv(c, st, h.p, co, nullptr, 0, B, h.H, h.W, *w, 3, d.p, o, c->zero_page));
      }
      h = d;
    }
  }
  { Act n = alloc_act(c, B, h.H, h.W, h.C); if (!n.p) return -1;
    GETV(gg, v + "decoder.conv_norm_out.weight"); GETV(gb, v + "decoder.conv_norm_out.bias");
    CK(run_gn(c, st, h.p, h.C, nullptr, 0, B, h.H * h.W, gg, gb, G, 1e-6f, 1, n.p, &h));
    GETW(w, v + "decoder.conv_out.weight"); GETV(b, v + "decoder.conv_out.bias");
    GemmOpt o; o.bias = b; o.out_f32 = 1; o.ldo = 4;
    CK(run_conv(c, st, n.p, h.C, nullptr, 0, B, h.H, h.W, *w, 3, img_out, o, c->zero_page)); }
  return 0;
}

// AutoencoderKL.encode: x [B][S*S][64] bf16 (3 image channels zero-padded) -> moments fp32 NHWC [B][L*L][2*lc]
static int vae_encode_walk(agd_ctx* c, hipStream_t st, const bf16_t* xin, int B, int S_, float* moments) {
  const agd_config& g = c->cfg;
  const int nl = g.vae_n_levels, G = g.vae_norm_num_groups, lc = g.vae_latent_channels;
  const std::string v = "vae.";
  c->arena.release(0);
  Act h = alloc_act(c, B, S_, S_, g.vae_block_out_channels[0], true); if (!h.p) return -1;
  { GETW(w, v + "encoder.conv_in.weight"); GETV(b, v + "encoder.conv_in.bias"); GemmOpt o; o.bias = b; o.out_act = &h;
    CK(run_conv(c, st, xin, 64, nullptr, 0, B, S_, S_, *w, 3, h.p, o, c->zero_page)); }
  for (int i = 0; i < nl; ++i) {
    const int co = g.vae_block_out_channels[i];
    for (int j = 0; j < g.vae_layers_per_block; ++j) {
      Act r; CK(resnet(c, st, v + "encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + ".", h, nullptr, co, 1e-6f, false, G, r));
      h = r;
    }
    if (i != nl - 1) {   // Downsample2D(padding=0) after F.pad(x, (0,1,0,1)): taps beyond the bottom/right edge read zeros
      const std::string k = v + "encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv.";
      GETW(w, k + "weight"); GETV(b, k + "bias");
      Act d = alloc_act(c, B, h.H / 2, h.W / 2, co, true); if (!d.p) return -1;
      GemmOpt o; o.bias = b; o.stride = 2; o.pad = 0; o.hout = h.H / 2; o.wout = h.W / 2; o.out_act = &d;
      CK(run_conv(c, st, h.p, co, nullptr, 0, B, h.H, h.W, *w, 3, d.p, o, c->zero_page));
      h = d;
    }
  }
  const int top = g.vae_block_out_channels[nl - 1], L = h.H;
  { Act r; CK(resnet(c, st, v + "encoder.mid_block.resnets.0.", h, nullptr, top, 1e-6f, false, G, r)); h = r; }
  CK(vae_mid_attention(c, st, v + "encoder.mid_block.attentions.0.", G, B, L, top, h));
  { Act r; CK(resnet(c, st, v + "encoder.mid_block.resnets.1.", h, nullptr, top, 1e-6f, false, G, r)); h = r; }
  Act n = alloc_act(c, B, L, L, top); if (!n.p) return -1;
  GETV(gg, v + "encoder.conv_norm_out.weight"); GETV(gb, v + "encoder.conv_norm_out.bias");
  CK(run_gn(c, st, h.p, top, nullptr, 0, B, L * L, gg, gb, G, 1e-6f, 1, n.p, &h));
  // conv_out (top -> 2*lc) into a zeroed 64-channel buffer, then quant_conv 1x1 (2*lc -> 2*lc) in fp32 out
  bf16_t* co64 = (bf16_t*)c->arena.alloc((size_t)B * L * L * 64 * 2); if (!co64) return -1;
  if (hipMemsetAsync(co64, 0, (size_t)B * L * L * 64 * 2, st) != hipSuccess) FAIL("memset enc");
  { GETW(w, v + "encoder.conv_out.weight"); GETV(b, v + "encoder.conv_out.bias"); GemmOpt o; o.bias = b; o.ldo = 64;
    CK(run_conv(c, st, n.p, top, nullptr, 0, B, L, L, *w, 3, co64, o, c->zero_page)); }
  { GETW(w, v + "quant_conv.weight"); GETV(b, v + "quant_conv.bias"); GemmOpt o; o.bias = b; o.out_f32 = 1; o.ldo = 2 * lc;
    CK(run_conv(c, st, co64, 64, nullptr, 0, B, L, L, *w, 1, moments, o, c->zero_page)); }
  return 0;
}

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
AGD_API const char* agd_version(void) { return "agenda_hip 0.1 (gfx950)"; }
AGD_API const char* agd_last_error(agd_ctx* c) { return (c && !c->err.empty()) ? c->err.c_str() : g_err; }
AGD_API const char* agd_profile_class_name(int cls) { return (cls >= 0 && cls < AGD_N_CLASSES) ? kClassNames[cls] : ""; }

AGD_API agd_ctx* agd_create(int device_id, const agd_config* cfg) {
  if (!cfg || cfg->struct_size != (int)sizeof(agd_config)) { agd_set_error("agd_create: bad config (struct_size %d != %zu)", cfg ? cfg->struct_size : -1, sizeof(agd_config)); return nullptr; }
  if (cfg->n_levels < 1 || cfg->n_levels > AGD_MAX_LEVELS || cfg->vae_n_levels < 1 || cfg->vae_n_levels > AGD_MAX_LEVELS) { agd_set_error("agd_create: bad level count"); return nullptr; }
  for (int i = 0; i < cfg->n_levels; ++i)
    if (cfg->block_out_channels[i] % 64) { agd_set_error("agd_create: UNet channels must be multiples of 64"); return nullptr; }
  for (int i = 0; i < cfg->vae_n_levels; ++i)
    if (cfg->vae_block_out_channels[i] % 64) { agd_set_error("agd_create: VAE channels must be multiples of 64"); return nullptr; }
  if (cfg->cross_attention_dim % 64) { agd_set_error("agd_create: cross_attention_dim must be a multiple of 64"); return nullptr; }
  if (hipSetDevice(device_id) != hipSuccess) { agd_set_error("hipSetDevice(%d) failed", device_id); return nullptr; }
  agd_ctx* c = new agd_ctx(); c->device = device_id; c->cfg = *cfg;
  const size_t ws = cfg->workspace_bytes > 0 ? (size_t)cfg->workspace_bytes : ((size_t)8 << 30);
  if (hipMalloc((void**)&c->arena.base, ws) != hipSuccess) { agd_set_error("arena hipMalloc(%zu) failed", ws); delete c; return nullptr; }
  c->arena.cap = ws;
  c->zero_page = dmalloc<bf16_t>(c, 2048);
  c->t_dev = dmalloc<float>(c, 64);
  if (!c->zero_page || !c->t_dev) { delete c; return nullptr; }
  hipMemset(c->zero_page, 0, 4096);
  return c;
}

AGD_API void agd_destroy(agd_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  for (void* p : c->owned) hipFree(p);
  for (auto& xl : c->xl) { xl.kvb.release(); xl.accb.release(); }
  c->ctxb.release(); c->hook_sumb.release(); c->hook_scratchb.release(); c->hook_headsb.release(); c->hook_storeb.release(); c->bwd_wsb.release();
  for (auto& xl : c->xl) { xl.wqTb.release(); xl.wkvTb.release(); xl.woTb.release(); xl.pm_kppb.release(); xl.pm_vppb.release(); xl.pm_csb.release(); }
  c->latb.release(); c->epsb.release(); c->vae_imgb.release(); c->plmsb.release();
  if (c->side) { hipStreamDestroy(c->side); hipEventDestroy(c->ev_fork); hipEventDestroy(c->ev_join); }
  if (c->splitk.p) hipFree(c->splitk.p);
  if (c->arena.base) hipFree(c->arena.base);
  if (c->stage) hipFree(c->stage);
  if (c->tsteps_buf) hipFree(c->tsteps_buf);
  for (auto e : c->ev_pool) hipEventDestroy(e);
  delete c;
}

static bool ends_with(const std::string& s, const char* suf) { const size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }

AGD_API int agd_load_tensor(agd_ctx* c, const char* name, const void* ptr, int dtype, int ndim, const long long* shape) {
  if (!c || !name || !ptr || !shape) { agd_set_error("agd_load_tensor: null argument"); return fail_ctx(c); }
  if (dtype != 0) { agd_set_error("agd_load_tensor: only float32 (dtype 0) supported"); return fail_ctx(c); }
  hipSetDevice(c->device);
  long long n = 1; for (int i = 0; i < ndim; ++i) n *= shape[i];
  const size_t bytes = (size_t)n * 4;
  if (bytes > c->stage_bytes) {
    if (c->stage) hipFree(c->stage);
    if (hipMalloc((void**)&c->stage, bytes) != hipSuccess) { c->stage = nullptr; c->stage_bytes = 0; agd_set_error("stage alloc failed"); return fail_ctx(c); }
    c->stage_bytes = bytes;
  }
  if (hipMemcpy(c->stage, ptr, bytes, hipMemcpyDefault) != hipSuccess) { agd_set_error("copy of '%s' failed", name); return fail_ctx(c); }
  const std::string k(name);
  if (ndim == 1) {
    float* d = dmalloc<float>(c, (size_t)n); if (!d) return fail_ctx(c);
    hipMemcpy(d, c->stage, bytes, hipMemcpyDeviceToDevice);
    c->V[k] = d; c->Vn[k] = (int)n;
  } else if (ndim == 2 || ndim == 4) {
    WMat w; w.N = (int)shape[0]; w.Cin = (int)shape[1]; w.taps = ndim == 4 ? (int)(shape[2] * shape[3]) : 1;
    if (w.taps != 1 && w.taps != 9) { agd_set_error("'%s': only 1x1 / 3x3 kernels", name); return fail_ctx(c); }
    w.Cpad = (w.Cin + 63) / 64 * 64;
    const bool is_tok_emb = ends_with(k, "token_embedding.weight");
    w.w = dmalloc<bf16_t>(c, (size_t)(w.N + (is_tok_emb ? kTextExtraRows : 0)) * w.taps * w.Cpad); if (!w.w) return fail_ctx(c);
    if (is_tok_emb) hipMemset(w.w + (size_t)w.N * w.Cpad, 0, (size_t)kTextExtraRows * w.Cpad * 2);
    const int geglu_bn = ends_with(k, "ff.net.0.proj.weight") ? 16 : 0;   // [8 values | 8 gates] per 16 rows (igemm_epilogue.h)
    if (geglu_bn && (w.N % 256)) { agd_set_error("'%s': GEGLU projection rows %d not a multiple of 256", name, w.N); return fail_ctx(c); }
    API_CK(c, launch_convert_weight(c->stage, w.w, w.N, w.Cin, w.taps, w.Cpad, geglu_bn, 0));
    hipDeviceSynchronize();
    c->W[k] = w;
  } else { agd_set_error("'%s': unsupported ndim %d", name, ndim); return fail_ctx(c); }
  return 0;
}

static int concat_rows(agd_ctx* c, const std::vector<const WMat*>& parts, WMat& out) {
  out = WMat(); out.Cin = parts[0]->Cin; out.Cpad = parts[0]->Cpad; out.taps = parts[0]->taps;
  for (auto* p : parts) { if (p->Cpad != out.Cpad || p->taps != out.taps) FAIL("concat_rows: mismatched K"); out.N += p->N; }
  out.w = dmalloc<bf16_t>(c, (size_t)out.N * out.taps * out.Cpad); if (!out.w) return -1;
  size_t off = 0;
  for (auto* p : parts) { const size_t nb = (size_t)p->N * p->taps * p->Cpad; hipMemcpy(out.w + off, p->w, nb * 2, hipMemcpyDeviceToDevice); off += nb; }
  return 0;
}

AGD_API int agd_finalize(agd_ctx* c) {
  if (!c) return -1;
  hipSetDevice(c->device);
  const agd_config& g = c->cfg;
  // ---- enumerate transformer blocks: fused QKV + cross K/V weights, recorder layers (daam order: up, down, mid)
  std::vector<std::pair<std::string, int>> tf;   // (prefix, level)
  const int nl = g.n_levels;
  for (int i = 0; i < nl; ++i) { const int lvl = nl - 1 - i;
    if (g.down_cross[lvl]) for (int j = 0; j < g.layers_per_block + 1; ++j) tf.push_back({"unet.up_blocks." + std::to_string(i) + ".attentions." + std::to_string(j) + ".", lvl}); }
  for (int i = 0; i < nl; ++i)
    if (g.down_cross[i]) for (int j = 0; j < g.layers_per_block; ++j) tf.push_back({"unet.down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j) + ".", i});
  tf.push_back({"unet.mid_block.attentions.0.", nl - 1});
  for (auto& pr : tf) {
    const std::string t = pr.first + "transformer_blocks.0.";
    const WMat* q = getW(c, t + "attn1.to_q.weight"); const WMat* k = getW(c, t + "attn1.to_k.weight"); const WMat* v = getW(c, t + "attn1.to_v.weight");
    if (!q || !k || !v) return fail_ctx(c);
    WMat qkv; API_CK(c, concat_rows(c, {q, k, v}, qkv)); c->W[t + "attn1.qkv"] = qkv;
    const WMat* ck = getW(c, t + "attn2.to_k.weight"); const WMat* cv = getW(c, t + "attn2.to_v.weight");
    if (!ck || !cv) return fail_ctx(c);
    XLayer xl; xl.name = t + "attn2"; xl.C = q->N; xl.level = pr.second; xl.heads = g.num_heads[pr.second];
    xl.mid = pr.first.find("mid_block") != std::string::npos;
    API_CK(c, concat_rows(c, {ck, cv}, xl.wkv));
    if (xl.C % xl.heads) { agd_set_error("%s: C %d not divisible by heads %d", xl.name.c_str(), xl.C, xl.heads); return fail_ctx(c); }
    // pre-multiplied attn2 (xattn_pre.hip) where it saves work: H x 80 padded token columns <= C / 2, i.e. head dim >= 160 (SD-1.x: the C = 1280 blocks)
    { const WMat* wq2 = getW(c, t + "attn2.to_q.weight"); const WMat* wo2 = getW(c, t + "attn2.to_out.0.weight");
      auto be2 = c->V.find(t + "norm2.bias");
      const int C2 = xl.C, D2 = C2 / xl.heads;
      // (option bit 1, off by default: also where the columns equal C -- head dim 80, the C = 640 blocks: no fewer MACs, but three full-chip launches instead of the
      //  half-chip chain kernel)
      if (wq2 && wo2 && be2 != c->V.end() && D2 % 8 == 0 && C2 % 160 == 0 && C2 % 64 == 0 && (xl.heads * XATTN_TP) % 64 == 0 && xl.heads * XATTN_TP <= C2 &&
          wq2->taps == 1 && wq2->N == C2 && wq2->Cpad == C2 && wo2->taps == 1 && wo2->N == C2 && wo2->Cpad == C2) {
        xl.pm_wqT = dmalloc<bf16_t>(c, (size_t)C2 * C2); xl.pm_wqb = dmalloc<float>(c, C2);
        if (!xl.pm_wqT || !xl.pm_wqb) return fail_ctx(c);
        API_CK(c, launch_transpose_bf16(wq2->w, C2, C2, xl.pm_wqT, 0));
        API_CK(c, launch_matvec_bf16(wq2->w, be2->second, xl.pm_wqb, C2, C2, 0));       // (Wq beta)[(h,d)]
      } }
    c->xl_idx[xl.name] = (int)c->xl.size(); c->xl.push_back(xl);
    // LayerNorm folded into the three GEMMs it feeds: W' = W diag(gamma), colsum(W'), bias' = bias + W beta
    struct Fold { const char* w; const char* bias; const char* ln; int geglu; };
    const Fold folds[3] = {{"attn1.qkv", nullptr, "norm1", 0}, {"attn2.to_q.weight", nullptr, "norm2", 0}, {"ff.net.0.proj.weight", "ff.net.0.proj.bias", "norm3", 16}};
    for (const Fold& f : folds) {
      const WMat* w = getW(c, t + f.w); const float* ga = getV(c, t + f.ln + ".weight"); const float* be = getV(c, t + f.ln + ".bias");
      if (!w || !ga || !be) return fail_ctx(c);
      const float* b0 = f.bias ? getV(c, t + f.bias) : nullptr;
      if (f.bias && !b0) return fail_ctx(c);
      if (w->taps != 1 || w->Cpad != w->Cin) { agd_set_error("%s: cannot fold LayerNorm (padded K)", (t + f.w).c_str()); return fail_ctx(c); }
      WMat wf = *w; wf.w = dmalloc<bf16_t>(c, (size_t)w->N * w->Cpad);
      float* cs = dmalloc<float>(c, w->N); float* bf = dmalloc<float>(c, w->N);
      if (!wf.w || !cs || !bf) return fail_ctx(c);
      API_CK(c, launch_ln_fold_weight(w->w, ga, be, b0, w->N, w->Cpad, f.geglu, wf.w, cs, bf, 0));
      const std::string k = t + f.w + ".lnfold";
      c->W[k] = wf; c->V[k + ".cs"] = cs; c->V[k + ".bias"] = bf; c->Vn[k + ".cs"] = w->N; c->Vn[k + ".bias"] = w->N;
    }
    if (q->N == 1280) {                                 // the C = 1280 GEGLU matrix once more in igemm_wreg.h's fragment order (option wreg_mask bit 0)
      auto it = c->W.find(t + "ff.net.0.proj.weight.lnfold");
      if (it != c->W.end() && it->second.taps == 1 && it->second.N % 256 == 0) {
        WMat& wm_ = it->second;
        wm_.wfrag = dmalloc<bf16_t>(c, (size_t)wm_.N * wm_.Cpad); if (!wm_.wfrag) return fail_ctx(c);
        API_CK(c, launch_frag_order_w(wm_.w, wm_.wfrag, wm_.N, wm_.Cpad, 4, wm_.Cpad, 0)); wm_.wfrag_ni = 4;
      }
    }
    // ff.net.2 and proj_out pre-multiplied: [Wp W2 | Wp] (rows of 5 C) and Wp b2 + bp, for the blocks whose feed-forward runs as separate launches
    { const WMat* w2 = getW(c, t + "ff.net.2.weight"); const WMat* wp = getW(c, pr.first + "proj_out.weight");
      auto b2 = c->V.find(t + "ff.net.2.bias"); auto bp = c->V.find(pr.first + "proj_out.bias");
      if (w2 && wp && b2 != c->V.end() && bp != c->V.end() && w2->taps == 1 && wp->taps == 1 && wp->N == q->N && wp->Cpad == q->N && w2->N == q->N && w2->Cpad == 4 * q->N) {
        const int C = q->N;
        bf16_t* w2t = dmalloc<bf16_t>(c, (size_t)4 * C * C);
        WMat wc = *wp; wc.N = C; wc.taps = 1; wc.Cpad = 5 * C; wc.Cin = 5 * C; wc.wfrag = nullptr; wc.wfrag_ni = 0; wc.sc_cols = 0;
        wc.w = dmalloc<bf16_t>(c, (size_t)C * 5 * C); float* bc = dmalloc<float>(c, C);
        if (!w2t || !wc.w || !bc) return fail_ctx(c);
        API_CK(c, launch_transpose_bf16(w2->w, C, 4 * C, w2t, 0));                       // W2 [C][4C] -> [4C][C]
        { WMat wt; wt.w = w2t; wt.N = 4 * C; wt.Cin = C; wt.Cpad = C; wt.taps = 1;        // (Wp W2)[n][k] = sum_j Wp[n][j] W2T[k][j]: Wp's rows as the activation rows
          GemmOpt o; o.ldo = 5 * C;
          API_CK(c, run_conv(c, 0, wp->w, C, nullptr, 0, 1, 1, C, wt, 1, wc.w, o, c->zero_page)); }
        if (hipMemcpy2D(wc.w + 4 * C, (size_t)5 * C * 2, wp->w, (size_t)C * 2, (size_t)C * 2, C, hipMemcpyDeviceToDevice) != hipSuccess) { agd_set_error("finalize: ff/proj matrix assembly failed"); return fail_ctx(c); }
        std::vector<unsigned short> hw((size_t)C * C); std::vector<float> hb2(C), hbp(C);
        if (hipMemcpy(hw.data(), wp->w, (size_t)C * C * 2, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(hb2.data(), b2->second, C * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(hbp.data(), bp->second, C * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { agd_set_error("finalize: ff/proj bias read failed"); return fail_ctx(c); }
        for (int n = 0; n < C; ++n) {
          double a = hbp[n];
          for (int j = 0; j < C; ++j) { unsigned u = (unsigned)hw[(size_t)n * C + j] << 16; float f; memcpy(&f, &u, 4); a += (double)f * hb2[j]; }
          hbp[n] = (float)a;
        }
        if (hipMemcpy(bc, hbp.data(), C * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { agd_set_error("finalize: ff/proj bias write failed"); return fail_ctx(c); }
        c->W[pr.first + "ffproj.weight"] = wc; c->V[pr.first + "ffproj.bias"] = bc; c->Vn[pr.first + "ffproj.bias"] = C;
        if (C == 320) {                                   // the fused feed-forward kernel's form of it: Wp W2 alone, in ff.net.2's fragment order
          bf16_t* tmpc = dmalloc<bf16_t>(c, (size_t)C * 4 * C); WMat f2p = *w2; f2p.wfrag = nullptr; f2p.wfrag_ni = 0;
          f2p.w = dmalloc<bf16_t>(c, (size_t)C * 4 * C);
          if (!tmpc || !f2p.w) return fail_ctx(c);
          if (hipMemcpy2D(tmpc, (size_t)4 * C * 2, wc.w, (size_t)5 * C * 2, (size_t)4 * C * 2, C, hipMemcpyDeviceToDevice) != hipSuccess) { agd_set_error("finalize: Wp W2 copy failed"); return fail_ctx(c); }
          API_CK(c, launch_frag_order_w(tmpc, f2p.w, C, 4 * C, C / 64, 128, 0));
          c->W[t + "ff.w2p.frag"] = f2p;
          hipDeviceSynchronize(); dfree(c, tmpc);          // load-time scratch
        }
        hipDeviceSynchronize(); dfree(c, w2t);
      } }
    // fused row-panel kernels (tblock.hip, C = 320 blocks): the matrices once more in MFMA fragment order
    if (q->N == 320 || q->N == 640) {                 // (C = 640: the attn2 chain only -- a wave's GEMM tile is 80 columns whatever C: NI = 5)
      const int C = q->N;
      const WMat* w1 = getW(c, t + "ff.net.0.proj.weight.lnfold"); const WMat* w2 = getW(c, t + "ff.net.2.weight");
      if (!w1 || !w2) return fail_ctx(c);
      if (C == 320 && w1->N == 8 * C && w1->Cpad == C && w2->N == C && w2->Cpad == 4 * C && w2->taps == 1) {
        WMat f1 = *w1, f2 = *w2;
        f1.w = dmalloc<bf16_t>(c, (size_t)w1->N * C); f2.w = dmalloc<bf16_t>(c, (size_t)C * 4 * C);
        if (!f1.w || !f2.w) return fail_ctx(c);
        API_CK(c, launch_frag_order_w1(w1->w, f1.w, C, 4 * C, 0));
        API_CK(c, launch_frag_order_w(w2->w, f2.w, C, 4 * C, C / 64, 128, 0));
        c->W[t + "ff.w1.frag"] = f1; c->W[t + "ff.w2.frag"] = f2;
        const WMat* wp = getW(c, pr.first + "proj_out.weight");
        if (!wp) return fail_ctx(c);
        if (wp->N == C && wp->Cpad == C && wp->taps == 1) {
          WMat fp_ = *wp; fp_.w = dmalloc<bf16_t>(c, (size_t)C * C); if (!fp_.w) return fail_ctx(c);
          API_CK(c, launch_frag_order_w(wp->w, fp_.w, C, C, C / 64, C, 0));
          c->W[pr.first + "proj_out.frag"] = fp_;
        }

      }
      if (C == 640) {                                   // proj_in / proj_out once more in igemm_wreg.h's fragment order (option wreg_mask bit 1)
        for (const char* nm : {"proj_in.weight", "proj_out.weight"}) {
          auto it = c->W.find(pr.first + nm); if (it == c->W.end()) { agd_set_error("finalize: missing weight '%s%s'", pr.first.c_str(), nm); return fail_ctx(c); }
          WMat& wm_ = it->second;
          if (wm_.taps == 1 && wm_.N % 128 == 0 && wm_.Cpad % 64 == 0) {
            wm_.wfrag = dmalloc<bf16_t>(c, (size_t)wm_.N * wm_.Cpad); if (!wm_.wfrag) return fail_ctx(c);
            API_CK(c, launch_frag_order_w(wm_.w, wm_.wfrag, wm_.N, wm_.Cpad, 2, wm_.Cpad, 0)); wm_.wfrag_ni = 2;
          }
        }
      }
      { const WMat* wi = getW(c, pr.first + "proj_in.weight");
        if (!wi) return fail_ctx(c);
        if (wi->N == C && wi->Cpad == C && wi->taps == 1) {
          WMat fi = *wi; fi.w = dmalloc<bf16_t>(c, (size_t)C * C); if (!fi.w) return fail_ctx(c);
          API_CK(c, launch_frag_order_w(wi->w, fi.w, C, C, 5, C, 0));
          c->W[pr.first + "proj_in.frag"] = fi;
        } }
      const WMat* wq = getW(c, t + "attn2.to_q.weight"); const WMat* wo = getW(c, t + "attn2.to_out.0.weight");
      if (!wq || !wo) return fail_ctx(c);
      if (wq->N == C && wq->Cpad == C && wq->taps == 1 && wo->N == C && wo->Cpad == C && wo->taps == 1 && xl.heads == 8) {
        WMat fq = *wq, fo = *wo;
        fq.w = dmalloc<bf16_t>(c, (size_t)C * C); fo.w = dmalloc<bf16_t>(c, (size_t)C * C);
        if (!fq.w || !fo.w) return fail_ctx(c);
        API_CK(c, launch_frag_order_w(wq->w, fq.w, C, C, 5, C, 0));
        API_CK(c, launch_frag_order_w(wo->w, fo.w, C, C, 5, C, 0));
        c->W[t + "attn2.to_q.frag"] = fq; c->W[t + "attn2.to_out.frag"] = fo;
        { const WMat* wqkv = getW(c, t + "attn1.qkv"); if (!wqkv) return fail_ctx(c);
          if (wqkv->N == 3 * C && wqkv->Cpad == C && wqkv->taps == 1) {
            WMat fqkv = *wqkv; fqkv.w = dmalloc<bf16_t>(c, (size_t)3 * C * C); if (!fqkv.w) return fail_ctx(c);
            API_CK(c, launch_frag_order_w(wqkv->w, fqkv.w, 3 * C, C, 5, C, 0));
            c->W[t + "attn1.qkv.frag"] = fqkv;
          } }
        const WMat* wo1 = getW(c, t + "attn1.to_out.0.weight");
        if (!wo1) return fail_ctx(c);
        if (wo1->N == C && wo1->Cpad == C && wo1->taps == 1) {
          WMat f1o = *wo1; f1o.w = dmalloc<bf16_t>(c, (size_t)C * C); if (!f1o.w) return fail_ctx(c);
          API_CK(c, launch_frag_order_w(wo1->w, f1o.w, C, C, 5, C, 0));
          c->W[t + "attn1.to_out.frag"] = f1o;
        }
      }
    }
  }
  // ---- the UNet's upsampling convs once more as the merged phase matrices [4 Cout][4 taps][Cin] (+ the bias four times)
  { std::vector<std::string> keys;
    const std::string tail = "upsamplers.0.conv.weight";
    for (auto& kv : c->W) if (kv.first.size() > tail.size() && kv.first.compare(kv.first.size() - tail.size(), tail.size(), tail) == 0)
      keys.push_back(kv.first.substr(0, kv.first.size() - 6));                 // "... .conv." (UNet and VAE decoder)
    for (const std::string& k : keys) {
      const WMat* w = getW(c, k + "weight"); auto bi = c->V.find(k + "bias");
      if (!w || bi == c->V.end()) return fail_ctx(c);
      if (w->taps != 9 || (w->Cpad & 63) || (w->N % 160 && w->N % 128)) continue;
      WMat w4 = *w; w4.N = 4 * w->N; w4.taps = 4; w4.wfrag = nullptr; w4.wfrag_ni = 0; w4.sc_cols = 0;
      w4.w = dmalloc<bf16_t>(c, (size_t)w4.N * 4 * w->Cpad); float* b4 = dmalloc<float>(c, w4.N);
      if (!w4.w || !b4) return fail_ctx(c);
      API_CK(c, launch_upsample_phase_weight(w->w, w4.w, w->N, w->Cpad, 0));
      for (int ph = 0; ph < 4; ++ph)
        if (hipMemcpy(b4 + (size_t)ph * w->N, bi->second, w->N * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess) { agd_set_error("finalize: phase bias copy failed"); return fail_ctx(c); }
      c->W[k + "phases"] = w4; c->V[k + "phases.bias"] = b4; c->Vn[k + "phases.bias"] = w4.N;
    } }
  // ---- UNet resnets with a conv_shortcut: conv2's matrix once more with the shortcut's columns appended to every row, and the two biases summed
  { std::vector<std::string> pres;
    const std::string tail = "conv_shortcut.weight";
    for (auto& kv : c->W) if (kv.first.compare(0, 5, "unet.") == 0 && kv.first.size() > tail.size() && kv.first.compare(kv.first.size() - tail.size(), tail.size(), tail) == 0)
      pres.push_back(kv.first.substr(0, kv.first.size() - tail.size()));
    for (const std::string& pre : pres) {
      const WMat* w2 = getW(c, pre + "conv2.weight"); const WMat* ws = getW(c, pre + "conv_shortcut.weight");
      auto b2 = c->V.find(pre + "conv2.bias"); auto bs = c->V.find(pre + "conv_shortcut.bias");
      if (!w2 || !ws || b2 == c->V.end() || bs == c->V.end()) return fail_ctx(c);
      if (w2->taps != 9 || ws->taps != 1 || w2->N != ws->N || (ws->Cpad & 63) || (w2->Cpad & 63)) continue;
      const size_t k2 = (size_t)9 * w2->Cpad, ks = (size_t)ws->Cpad;
      WMat f = *w2; f.sc_cols = (int)ks; f.wfrag = nullptr; f.wfrag_ni = 0;
      f.w = dmalloc<bf16_t>(c, (size_t)f.N * (k2 + ks)); float* fb = dmalloc<float>(c, f.N);
      if (!f.w || !fb) return fail_ctx(c);
      if (hipMemcpy2D(f.w, (k2 + ks) * 2, w2->w, k2 * 2, k2 * 2, f.N, hipMemcpyDeviceToDevice) != hipSuccess ||
          hipMemcpy2D(f.w + k2, (k2 + ks) * 2, ws->w, ks * 2, ks * 2, f.N, hipMemcpyDeviceToDevice) != hipSuccess) { agd_set_error("finalize: shortcut weight concat failed"); return fail_ctx(c); }
      std::vector<float> ha(f.N), hb(f.N);
      if (hipMemcpy(ha.data(), b2->second, f.N * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(hb.data(), bs->second, f.N * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { agd_set_error("finalize: bias read failed"); return fail_ctx(c); }
      for (int i = 0; i < f.N; ++i) ha[i] += hb[i];
      if (hipMemcpy(fb, ha.data(), f.N * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { agd_set_error("finalize: bias write failed"); return fail_ctx(c); }
      c->W[pre + "conv2.sc"] = f; c->V[pre + "conv2.sc.bias"] = fb; c->Vn[pre + "conv2.sc.bias"] = f.N;
    } }
  // ---- all time_emb_proj stacked into one [sum Cout][4*dim] matrix
  { std::vector<const WMat*> parts; std::vector<std::string> pres; int total = 0;
    for (auto& kv : c->W) if (ends_with(kv.first, "time_emb_proj.weight")) pres.push_back(kv.first.substr(0, kv.first.size() - strlen("time_emb_proj.weight")));
    std::sort(pres.begin(), pres.end());
    for (auto& p : pres) { const WMat* w = getW(c, p + "time_emb_proj.weight"); parts.push_back(w); c->tproj_off[p] = total; total += w->N; }
    if (!parts.empty()) {
      API_CK(c, concat_rows(c, parts, c->tproj_all)); c->tproj_total = total;
      c->tproj_bias = dmalloc<float>(c, total); c->tproj_out = dmalloc<float>(c, total);
      if (!c->tproj_bias || !c->tproj_out) return fail_ctx(c);
      for (auto& p : pres) { const float* b = getV(c, p + "time_emb_proj.bias"); if (!b) return fail_ctx(c);
        hipMemcpy(c->tproj_bias + c->tproj_off[p], b, (size_t)c->Vn[p + "time_emb_proj.bias"] * 4, hipMemcpyDeviceToDevice); }
    }
    const int dim = g.block_out_channels[0];
    c->temb_buf = dmalloc<float>(c, (size_t)dim * 9); if (!c->temb_buf) return fail_ctx(c); }
  // ---- CLIP text encoder: fused q/k/v projection per layer
  for (int l = 0; l < g.text_layers; ++l) {
    const std::string a = "text.encoder.layers." + std::to_string(l) + ".self_attn.";
    const WMat* q = getW(c, a + "q_proj.weight"); const WMat* k = getW(c, a + "k_proj.weight"); const WMat* v = getW(c, a + "v_proj.weight");
    const float* bq = getV(c, a + "q_proj.bias"); const float* bk = getV(c, a + "k_proj.bias"); const float* bv = getV(c, a + "v_proj.bias");
    if (!q || !k || !v || !bq || !bk || !bv) return fail_ctx(c);
    WMat qkv; API_CK(c, concat_rows(c, {q, k, v}, qkv)); c->W[a + "qkv.weight"] = qkv;
    float* b = dmalloc<float>(c, (size_t)3 * q->N); if (!b) return fail_ctx(c);
    hipMemcpy(b, bq, (size_t)q->N * 4, hipMemcpyDeviceToDevice); hipMemcpy(b + q->N, bk, (size_t)q->N * 4, hipMemcpyDeviceToDevice);
    hipMemcpy(b + 2 * q->N, bv, (size_t)q->N * 4, hipMemcpyDeviceToDevice);
    c->V[a + "qkv.bias"] = b; c->Vn[a + "qkv.bias"] = 3 * q->N;
  }
  hipDeviceSynchronize();
  c->finalized = true;
  return 0;
}

static hipStream_t S(void* s) { return (hipStream_t)s; }
static int need_final(agd_ctx* c) { if (!c) { agd_set_error("null ctx"); return -1; } if (!c->finalized) { agd_set_error("agd_finalize not called"); return -1; } hipSetDevice(c->device); return 0; }

AGD_API int agd_set_context(agd_ctx* c, const float* ctx_emb, int batch2, int tokens, void* stream) {
  API_CK(c, need_final(c));
  hipStream_t st = S(stream);
  const int Dc = c->cfg.cross_attention_dim;
  if (tokens > 96) { agd_set_error("set_context: tokens %d > 96 unsupported", tokens); return fail_ctx(c); }
  if (batch2 < 1 || tokens < 1) { agd_set_error("set_context: batch2 %d tokens %d", batch2, tokens); return fail_ctx(c); }
  // buffers only ever grow (capacity-tracked); a shorter/smaller context reuses the same blocks
  API_CK(c, c->ctxb.ensure((size_t)batch2 * tokens * Dc * 2)); c->ctx_bf16 = c->ctxb.as<bf16_t>();
  for (auto& xl : c->xl) { API_CK(c, xl.kvb.ensure((size_t)batch2 * tokens * 2 * xl.C * 2)); xl.kv = xl.kvb.as<bf16_t>(); }
  c->ctx_B2 = batch2; c->ctx_T = tokens;
  API_CK(c, launch_f32_to_bf16(ctx_emb, c->ctx_bf16, (long long)batch2 * tokens * Dc, st));
  for (auto& xl : c->xl) {
    GemmOpt o;
    API_CK(c, run_conv(c, st, c->ctx_bf16, Dc, nullptr, 0, 1, 1, batch2 * tokens, xl.wkv, 1, xl.kv, o, c->zero_page));
    // the context products of the pre-multiplied attn2 form, once per prompt batch (xattn_pre.hip)
    xl.pm_ready = false;
    if (c->opt_xpre && xl.pm_wqT && tokens <= XATTN_TP && ((c->opt_xpre & 2) || xl.heads * XATTN_TP * 2 <= xl.C)) {
      const std::string t = xl.name.substr(0, xl.name.size() - 5);          // "...transformer_blocks.0."
      const WMat* wo2 = getW(c, t + "attn2.to_out.0.weight"); const float* g2 = getV(c, t + "norm2.weight");
      if (!wo2 || !g2) { agd_set_error("%s: pre-multiplied form without attn2.to_out.0.weight / norm2.weight", xl.name.c_str()); return fail_ctx(c); }
      const size_t HT = (size_t)xl.heads * XATTN_TP;
      API_CK(c, xl.pm_kppb.ensure((size_t)batch2 * HT * xl.C * 2)); API_CK(c, xl.pm_vppb.ensure((size_t)batch2 * HT * xl.C * 2)); API_CK(c, xl.pm_csb.ensure((size_t)batch2 * HT * 2 * sizeof(float)));
      xl.pm_kpp = xl.pm_kppb.as<bf16_t>(); xl.pm_vpp = xl.pm_vppb.as<bf16_t>(); xl.pm_kcs = xl.pm_csb.as<float>(); xl.pm_kbs = xl.pm_kcs + (size_t)batch2 * HT;
      XattnPremulP pm{}; pm.kv = xl.kv; pm.ldkv = 2 * xl.C; pm.skv = (long long)tokens * 2 * xl.C; pm.wqT = xl.pm_wqT; pm.wo = wo2->w; pm.gamma = g2; pm.wqb = xl.pm_wqb;
      pm.B = batch2; pm.T = tokens; pm.C = xl.C; pm.H = xl.heads; pm.scale = 1.0f / sqrtf((float)(xl.C / xl.heads));
      pm.kpp = xl.pm_kpp; pm.kcs = xl.pm_kcs; pm.kbs = xl.pm_kbs; pm.vpp = xl.pm_vpp;
      { ProfScope ps(c, st, PC_ATTN_CROSS, 8.0 * batch2 * (double)HT * xl.C * (xl.C / xl.heads) / 2.0, 4.0 * batch2 * (double)HT * xl.C);
        API_CK(c, launch_xattn_premul(pm, st)); }
      xl.pm_ready = true;
    }
  }
  return 0;
}

static int ensure_lat(agd_ctx* c, int B2, int L) {
  const size_t need = (size_t)B2 * L * L * 64;
  const int oc = c->cfg.out_channels > 4 ? c->cfg.out_channels : 4;
  CK(c->latb.ensure(need * 2)); CK(c->epsb.ensure((size_t)B2 * L * L * oc * 4));
  c->lat_bf16 = c->latb.as<bf16_t>(); c->eps_nhwc = c->epsb.as<float>();
  return 0;
}

static int embed_all_timesteps(agd_ctx* c, hipStream_t st, const float* timesteps, int n, const float** out);
AGD_API int agd_unet_forward(agd_ctx* c, const float* sample, int batch2, int L, float timestep, float* out, void* stream) {
  API_CK(c, need_final(c));
  hipStream_t st = S(stream);
  API_CK(c, ensure_lat(c, batch2, L));
  const int Cl = c->cfg.in_channels;
  { ProfScope ps(c, st, PC_ELEM, 0); API_CK(c, launch_prep_latents(sample, c->lat_bf16, batch2, Cl, L * L, 64, 1, 1.0f, st)); }
  API_CK(c, unet_walk(c, st, c->lat_bf16, batch2, L, timestep, c->eps_nhwc));
  { ProfScope ps(c, st, PC_ELEM, 0); API_CK(c, launch_nchw_from_nhwc_f32(c->eps_nhwc, c->cfg.out_channels, out, batch2, c->cfg.out_channels, L * L, st)); }
  return 0;
}

// the training call `unet(noisy, timesteps[bsz], encoder_hidden_states)` (finetune_sd_token.py:1027): one timestep PER IMAGE
// (host array of batch2 floats): every image gets its own time-embedding row in the resnets' row add
AGD_API int agd_unet_forward_ts(agd_ctx* c, const float* sample, int batch2, int L, const float* timesteps, float* out, void* stream) {
  API_CK(c, need_final(c));
  if (!timesteps || batch2 < 1) { agd_set_error("unet_forward_ts: bad arguments"); return fail_ctx(c); }
  hipStream_t st = S(stream);
  API_CK(c, ensure_lat(c, batch2, L));
  const int Cl = c->cfg.in_channels;
  const float* tp_all = nullptr;
  API_CK(c, embed_all_timesteps(c, st, timesteps, batch2, &tp_all));
  { ProfScope ps(c, st, PC_ELEM, 0); API_CK(c, launch_prep_latents(sample, c->lat_bf16, batch2, Cl, L * L, 64, 1, 1.0f, st)); }
  API_CK(c, unet_walk(c, st, c->lat_bf16, batch2, L, timesteps[0], c->eps_nhwc, tp_all, false, c->tproj_total));
  { ProfScope ps(c, st, PC_ELEM, 0); API_CK(c, launch_nchw_from_nhwc_f32(c->eps_nhwc, c->cfg.out_channels, out, batch2, c->cfg.out_channels, L * L, st)); }
  return 0;
}

// eps strides for cfg_ddim: NCHW eps handled by transposing through the NHWC kernel's stride form
AGD_API int agd_cfg_ddim_step(agd_ctx* c, const float* eps, float* latents, int batch, int L, float guidance, float alpha_t,
                                 float alpha_prev, void* stream) {
  API_CK(c, need_final(c));
  hipStream_t st = S(stream);
  API_CK(c, ensure_lat(c, 2 * batch, L));
  // NCHW eps -> NHWC scratch (tiny), then the fused kernel
  const int C = c->cfg.out_channels, HW = L * L;
  // reuse nchw_from_nhwc in the reverse direction: treat eps as "NHWC with ldc=HW" is not valid; do an explicit pass
  float* tmp = c->eps_nhwc;
  // out[b][p][c] = eps[b][c][p]  == nchw_from_nhwc with (C,HW) swapped
  API_CK(c, launch_nchw_from_nhwc_f32(eps, HW, tmp, 2 * batch, HW, C, st));
  ProfScope ps(c, st, PC_ELEM, 0);
  API_CK(c, launch_cfg_ddim(tmp, C, latents, batch, C, HW, guidance, alpha_t, alpha_prev, c->cfg.prediction_type, st));
  return 0;
}

// time embeddings of every model evaluation of a denoise loop, 8 timesteps per launch (the stacked time_emb_proj matrix is
// ~50 MB of weights: streamed ceil(n/8) times instead of once per step); returns [n][tproj_total] in *out
static int embed_all_timesteps(agd_ctx* c, hipStream_t st, const float* timesteps, int n, const float** out) {
  const int dim0 = c->cfg.block_out_channels[0];
  const size_t per_step = (size_t)c->tproj_total + (size_t)9 * dim0;         // (scratch: 9 dim floats per row of a chunk; chunk <= n rows at a time)
  if (c->tsteps_cap < n) {
    if (c->tsteps_buf) { hipDeviceSynchronize(); hipFree(c->tsteps_buf); }
    c->tsteps_buf = nullptr; c->tsteps_cap = 0;
    if (hipMalloc((void**)&c->tsteps_buf, per_step * n * sizeof(float)) != hipSuccess) FAIL("denoise: time-embedding buffer alloc failed");
    c->tsteps_cap = n;
  }
  float* tp_all = c->tsteps_buf;                                  // [n][tproj_total]
  float* tscratch = c->tsteps_buf + (size_t)c->tproj_total * n;
  const int chunk = (size_t)25 * 4 * dim0 * 4 <= 160 * 1024 ? 25 : 8;      // rows per launch: the 4 dim-wide fp32 rows of a chunk sit in LDS (misc.hip small_linear_kernel)
  for (int s0 = 0; s0 < n; s0 += chunk) {
    const int m = n - s0 < chunk ? n - s0 : chunk;
    CK(time_embed(c, st, timesteps + s0, m, tscratch, tp_all + (size_t)s0 * c->tproj_total));
  }
  *out = tp_all;
  return 0;
}

AGD_API int agd_denoise(agd_ctx* c, float* latents, int batch, int L, int n_steps, const float* timesteps, const float* alpha_t,
                           const float* alpha_prev, float guidance, void* stream) {
  API_CK(c, need_final(c));
  hipStream_t st = S(stream);
  const int B2 = 2 * batch, Cl = c->cfg.in_channels, HW = L * L;
  API_CK(c, ensure_lat(c, B2, L));
  if (c->ctx_B2 != B2) { agd_set_error("denoise: context batch %d != 2*batch %d", c->ctx_B2, B2); return fail_ctx(c); }
  const float* tp_all = nullptr;                                   // all timesteps are known up front: embed them now
  API_CK(c, embed_all_timesteps(c, st, timesteps, n_steps, &tp_all));
  for (int s = 0; s < n_steps; ++s) {
    { ProfScope ps(c, st, PC_ELEM, 0); API_CK(c, launch_prep_latents(latents, c->lat_bf16, batch, Cl, HW, 64, 2, 1.0f, st)); }
    API_CK(c, unet_walk(c, st, c->lat_bf16, B2, L, timesteps[s], c->eps_nhwc, tp_all + (size_t)s * c->tproj_total, true));
    { ProfScope ps(c, st, PC_ELEM, 0);
      API_CK(c, launch_cfg_ddim(c->eps_nhwc, c->cfg.out_channels, latents, batch, Cl, HW, guidance, alpha_t[s], alpha_prev[s], c->cfg.prediction_type, st)); }
  }
  return 0;
}

// The denoise loop under the reference's ACTUAL scheduler: `pipeline(prompt, num_inference_steps=20)` at
// data_generation.py:59 runs the checkpoint's PNDMScheduler (skip_prk_steps, i.e. PLMS) -- n_evals = steps + 1 model
// evaluations, the second timestep evaluated twice [upstream-knowledge: diffusers 0.21.2 PNDMScheduler.step_plms].
// Per evaluation i the host passes the UNet timestep and the two coefficients of `_get_prev_sample`
// (prev = sample_coeff[i] * sample + eps_coeff[i] * model_output); the linear-multistep weights are applied here:
//   i = 0: e0                       (the sample is kept: evaluation 1 restarts from it)
//   i = 1: (e1 + e0) / 2 from the KEPT sample (e1 is not added to the history)
//   then : (3 e - h1) / 2 ; (23 e - 16 h1 + 5 h2) / 12 ; (55 e - 59 h1 + 37 h2 - 9 h3) / 24
AGD_API int agd_denoise_plms(agd_ctx* c, float* latents, int batch, int L, int n_evals, const float* timesteps, const float* sample_coeff,
                             const float* eps_coeff, float guidance, void* stream) {
  API_CK(c, need_final(c));
  hipStream_t st = S(stream);
  if (c->cfg.prediction_type != 0) { agd_set_error("denoise_plms: epsilon prediction only"); return fail_ctx(c); }
  if (n_evals < 2) { agd_set_error("denoise_plms: needs >= 2 model evaluations (got %d)", n_evals); return fail_ctx(c); }
  const int B2 = 2 * batch, Cl = c->cfg.in_channels, HW = L * L;
  API_CK(c, ensure_lat(c, B2, L));
  if (c->ctx_B2 != B2) { agd_set_error("denoise: context batch %d != 2*batch %d", c->ctx_B2, B2); return fail_ctx(c); }
  const size_t n1 = (size_t)batch * Cl * HW;
  API_CK(c, c->plmsb.ensure(n1 * 5 * sizeof(float)));            // 4 history slots + the kept sample
  float* hist[4]; for (int k = 0; k < 4; ++k) hist[k] = c->plmsb.as<float>() + n1 * k;
  float* kept = c->plmsb.as<float>() + n1 * 4;
  const float* tp_all = nullptr;
  API_CK(c, embed_all_timesteps(c, st, timesteps, n_evals, &tp_all));
  int n_hist = 0, head = 0;                                       // hist[(head - 1 - k) & 3] = k-th newest stored eps
  for (int i = 0; i < n_evals; ++i) {
    { ProfScope ps(c, st, PC_ELEM, 0); API_CK(c, launch_prep_latents(latents, c->lat_bf16, batch, Cl, HW, 64, 2, 1.0f, st)); }
    API_CK(c, unet_walk(c, st, c->lat_bf16, B2, L, timesteps[i], c->eps_nhwc, tp_all + (size_t)i * c->tproj_total, true));
    float w[4] = {1.f, 0.f, 0.f, 0.f};
    const float* h[3] = {nullptr, nullptr, nullptr};
    const float* src = latents; float* store = nullptr;
    if (i == 1) {                                                 // PLMS second call: average with e0, restart from the kept sample
      w[0] = 0.5f; w[1] = 0.5f; h[0] = hist[(head - 1) & 3]; src = kept;
    } else {
      if (i == 0 && hipMemcpyAsync(kept, latents, n1 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) { agd_set_error("plms: keep sample"); return fail_ctx(c); }
      store = hist[head & 3];
      for (int k = 0; k < 3 && k < n_hist; ++k) h[k] = hist[(head - 1 - k) & 3];
      if (n_hist == 1) { w[0] = 1.5f; w[1] = -0.5f; }
      else if (n_hist == 2) { w[0] = 23.f / 12.f; w[1] = -16.f / 12.f; w[2] = 5.f / 12.f; }
      else if (n_hist >= 3) { w[0] = 55.f / 24.f; w[1] = -59.f / 24.f; w[2] = 37.f / 24.f; w[3] = -9.f / 24.f; }
    }
    { ProfScope ps(c, st, PC_ELEM, 0);
      API_CK(c, launch_cfg_plms(c->eps_nhwc, c->cfg.out_channels, latents, src, h[0], h[1], h[2], store, batch, Cl, HW, guidance, w,
                                sample_coeff[i], eps_coeff[i], st)); }
    if (store) { ++head; if (n_hist < 3) ++n_hist; }
  }
  return 0;
}

AGD_API int agd_vae_decode(agd_ctx* c, const float* latents, int batch, int L, unsigned char* out_u8, float* out_f32, void* stream) {
  API_CK(c, need_final(c));
  hipStream_t st = S(stream);
  API_CK(c, ensure_lat(c, batch, L));
  const int S8 = L << (c->cfg.vae_n_levels - 1);
  const long long npix = (long long)batch * S8 * S8;
  { ProfScope ps(c, st, PC_ELEM, 0);
    API_CK(c, launch_prep_latents(latents, c->lat_bf16, batch, c->cfg.vae_latent_channels, L * L, 64, 1, 1.0f / c->cfg.vae_scaling_factor, st)); }
  API_CK(c, c->vae_imgb.ensure((size_t)npix * 4 * 4));     // grows once per (batch, side); stream-ordered reuse, no sync
  float* img = c->vae_imgb.as<float>();
  int rc = vae_walk(c, st, c->lat_bf16, batch, L, img);
  if (rc == 0 && out_u8) { ProfScope ps(c, st, PC_ELEM, 0); rc = launch_image_u8(img, 4, out_u8, npix, c->cfg.vae_out_channels, st); }
  if (rc == 0 && out_f32) {
    // [npix][4] -> [npix][3] : "NCHW from NHWC" with HW=1 does exactly that per pixel
    rc = launch_nchw_from_nhwc_f32(img, 4, out_f32, (int)npix, c->cfg.vae_out_channels, 1, st);
  }
  return rc ? fail_ctx(c) : 0;
}

// ---- options --------------------------------------------------------------------------
AGD_API int agd_set_option(agd_ctx* c, const char* name, int value) {
  if (!c || !name) { agd_set_error("set_option: null argument"); return fail_ctx(c); }
  if (!strcmp(name, "cfg_shared_prefix")) { c->opt_cfg_share = value != 0; return 0; }
  if (!strcmp(name, "ln_fold")) { c->opt_ln_fold = value; return 0; }       // 0 off, 1 on; 2 / 3: only blocks with C <= 320 / 640 (A/B)
  if (!strcmp(name, "gn_fused_stats")) { c->opt_gn_fused = value != 0; return 0; }
  if (!strcmp(name, "weight_touch")) { c->opt_touch = value < 0 ? 0 : value; return 0; }
  if (!strcmp(name, "weight_warm")) { c->opt_warm = value; return 0; }
  if (!strcmp(name, "conv_halo")) { c->opt_halo = value != 0; return 0; }
  if (!strcmp(name, "gn_proj_fold")) { c->opt_gn_proj_fold = value < 0 ? 0 : value; return 0; }   // 0 off, 1: blocks with C <= 320, 2: C <= 640 (A/B)
  if (!strcmp(name, "tblock_fuse")) { c->opt_tb_fuse = value < 0 ? 0 : value; return 0; }   // bit 0: fused feed-forward, bit 1: fused attn2 chain (C = 320 blocks)
  if (!strcmp(name, "reduce_gn")) { c->opt_reduce_gn = value != 0; return 0; }
  if (!strcmp(name, "xcd_block")) { c->opt_xcd_block = value != 0; return 0; }
  if (!strcmp(name, "igemm_pc")) { c->opt_pc = value < 0 ? 0 : value; return 0; }
  if (!strcmp(name, "attn2_premul")) { c->opt_xpre = value < 0 ? 0 : value; return 0; }     // takes effect at the next agd_set_context (the products are built there)
  if (!strcmp(name, "conv_smap")) { c->opt_smap = value != 0; return 0; }
  if (!strcmp(name, "side_stream")) { c->opt_side = value < 0 ? 0 : value; return 0; }
  if (!strcmp(name, "igemm_kgroups")) { c->opt_kg2 = value != 0; return 0; }
  if (!strcmp(name, "upsample_phases")) { c->opt_ups4 = value & 15; return 0; }      // bit 0: the UNet's upsamplers from 16 x 16 maps up, bit 1: the VAE decoder's, bit 2: the UNet's 8 x 8 -> 16 x 16 one too
  if (!strcmp(name, "ff_proj_fuse")) { c->opt_ffproj = value != 0; return 0; }
  if (!strcmp(name, "shortcut_fuse")) { c->opt_sc_fuse = value & 3; return 0; }      // bit 0: row-halo launches (64 x 64 .. 16 x 16 maps), bit 1: the 8 x 8 whole-images launches
  if (!strcmp(name, "wreg_mask")) { c->opt_wreg = value & 3; return 0; }
  if (!strcmp(name, "igemm8p")) { c->opt_p8 = value < 0 ? 0 : value; return 0; }   // 0 off, 1 on (the launcher decides per launch); tests: 2 / 3 / 4 force the 256-wide / 160-wide / any legal tile
  agd_set_error("set_option: unknown option '%s'", name);
  return fail_ctx(c);
}

// ---- recorder -------------------------------------------------------------------------
AGD_API int agd_record_config(agd_ctx* c, int mode, int is_train, int rec_tokens) {
  if (!c) return -1;
  if (mode < 0 || mode > 2) { agd_set_error("record_config: mode %d", mode); return fail_ctx(c); }
  c->rec_mode = mode; c->rec_is_train = is_train; c->rec_T_cfg = rec_tokens;   // takes effect at the next agd_record_reset
  return 0;
}

// hook.py recorder state for `rows` kept batch rows (B' of hook.py:48-55) at latent side L: running sum, per-call scratch
static int hook_reset_rows(agd_ctx* c, int rows, int L, hipStream_t st) {
  const int Tc = c->ctx_T > 0 ? c->ctx_T : c->cfg.max_tokens;
  const size_t n = (size_t)rows * Tc * L * L;
  CK(c->hook_sumb.ensure(n * 4)); CK(c->hook_scratchb.ensure(n * 4));
  c->hook_sum = c->hook_sumb.as<float>(); c->hook_scratch = c->hook_scratchb.as<float>();
  c->hook_Bp = rows; c->hook_T = Tc;
  if (hipMemsetAsync(c->hook_sum, 0, n * 4, st) != hipSuccess) FAIL("memset hook");
  c->hook_count = 0; c->hook_recs.clear(); c->hook_store_used = 0;
  return 0;
}

// `hooker.clear()` with the number of recorded batch rows given explicitly (training calls the UNet without CFG: any batch)
AGD_API int agd_hook_reset(agd_ctx* c, int rows, int L, void* stream) {
  API_CK(c, need_final(c));
  if (c->rec_mode != 2) { agd_set_error("hook_reset: recorder is not in hook.py mode (agd_record_config(2, ..))"); return fail_ctx(c); }
  if (rows < 1 || L < 1) { agd_set_error("hook_reset: rows %d latent side %d", rows, L); return fail_ctx(c); }
  API_CK(c, hook_reset_rows(c, rows, L, S(stream)));
  c->rec_B = c->rec_is_train ? (rows + 1) / 2 : rows; c->rec_L = L;
  return 0;
}

AGD_API int agd_record_reset(agd_ctx* c, int batch, int L, void* stream) {
  API_CK(c, need_final(c));
  hipStream_t st = S(stream);
  if (batch < 1 || L < 1) { agd_set_error("record_reset: batch %d latent side %d", batch, L); return fail_ctx(c); }
  const int T = c->rec_T_cfg > 0 ? c->rec_T_cfg : c->cfg.max_tokens;
  c->rec_T = T;
  if (c->rec_mode == 1) {
    for (auto& xl : c->xl) {
      if (xl.mid) continue;
      const int side = L >> xl.level;
      if (side < 1 || (L / side) == 8) { xl.acc = nullptr; xl.acc_side = 0; continue; }
      // capacity-tracked: a larger batch / token count / side than the block was allocated for reallocates it
      // daam averages clamp(bicubic(map), 0) over every (layer, head) map.  At latent resolution the resize is the identity and the
      // clamp cannot fire (sums of probabilities), so heads are summed in the recording kernel: down to 1/8 of the state and traffic
      xl.acc_heads = xl.heads;
      if (side == L) {       // heads per workgroup: as many as still leave >= 256 workgroups (UNet batch x head groups x 128-query tiles)
        int hpb = xl.heads;
        while (hpb > 1 && hpb % 2 == 0 && (long long)2 * batch * (xl.heads / hpb) * ((side * side + 127) / 128) < 256) hpb /= 2;
        xl.acc_heads = xl.heads / hpb;
      }
      const size_t n = (size_t)batch * xl.acc_heads * T * side * side;
      API_CK(c, xl.accb.ensure(n * 4)); xl.acc = xl.accb.as<float>(); xl.acc_side = side;
      if (hipMemsetAsync(xl.acc, 0, n * 4, st) != hipSuccess) { agd_set_error("memset acc"); return fail_ctx(c); }
    }
  } else if (c->rec_mode == 2) {
    API_CK(c, hook_reset_rows(c, c->rec_is_train ? 2 * batch : batch, L, st));   // `batch` = images; UNet batch is 2*batch under CFG
  }
  c->rec_B = batch; c->rec_L = L;
  return 0;
}

AGD_API int agd_daam_global(agd_ctx* c, int img, int rows, float* out, void* stream) {
  API_CK(c, need_final(c));
  hipStream_t st = S(stream);
  std::vector<HeatLayer> hl; int total = 0;
  for (auto& xl : c->xl) {
    if (!xl.acc || xl.mid) continue;
    HeatLayer h; h.acc = xl.acc; h.side = xl.acc_side; h.heads = xl.acc_heads;
    h.head_stride = (long long)c->rec_T * xl.acc_side * xl.acc_side; h.img_stride = h.head_stride * xl.acc_heads;
    hl.push_back(h); total += xl.heads;                            // the mean is over (layer, head) maps either way
  }
  if (hl.empty() || c->rec_mode != 1) { agd_set_error("No heat maps found. Did you forget to call `with trace(...)`?"); c->err = g_err; return -2; }
  if (rows > c->rec_T || img >= c->rec_B) { agd_set_error("daam_global: rows %d > recorded %d or img %d >= %d", rows, c->rec_T, img, c->rec_B); return fail_ctx(c); }
  { ProfScope ps(c, st, PC_HEAT, 0); API_CK(c, launch_daam_global(hl.data(), (int)hl.size(), total, rows, c->rec_L, img, out, st)); }
  return 0;
}

AGD_API int agd_hook_count(agd_ctx* c) { return c ? c->hook_count : 0; }

AGD_API int agd_hook_last_map(agd_ctx* c, float* out, int n_query, void* stream) {
  API_CK(c, need_final(c));
  hipStream_t st = S(stream);
  if (c->hook_count == 0 || !c->hook_scratch) { agd_set_error("No heat maps found."); c->err = g_err; return -2; }
  const size_t n = (size_t)c->hook_Bp * c->hook_T * n_query;
  if (hipMemcpyAsync(out, c->hook_scratch, n * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) { agd_set_error("hook_last_map copy"); return fail_ctx(c); }
  return 0;
}

AGD_API int agd_hook_global(agd_ctx* c, float* out, void* stream) {
  API_CK(c, need_final(c));
  hipStream_t st = S(stream);
  if (c->hook_count == 0 || !c->hook_sum) { agd_set_error("No heat maps found."); c->err = g_err; return -2; }
  const long long n = (long long)c->hook_Bp * c->hook_T * c->rec_L * c->rec_L;
  hipMemcpyAsync(out, c->hook_sum, (size_t)n * 4, hipMemcpyDeviceToDevice, st);
  API_CK(c, launch_scale(out, n, 1.0f / (float)c->hook_count, st));
  return 0;
}

// The processor seam: one `Attention` module call of the UNet, hook.py:83-122 in full.
//   layer "...attn2" + ctx_emb        -> cross-attention (is_cross, hook.py:95-99), feeds the recorder when record != 0
//   layer "...attn1" + ctx_emb NULL   -> self-attention (encoder_hidden_states = hidden_states), records nothing
//   attn_mask: additive fp32 [batch2][keys] or NULL (hook.py:92 prepare_attention_mask -> hook.py:108 get_attention_scores)
// Output is to_out[0](attention) + bias (to_out[1] is Dropout(0)); the residual is the caller's (BasicTransformerBlock).
AGD_API int agd_attn_processor(agd_ctx* c, const char* layer, const float* hidden, const float* ctx_emb, const float* attn_mask,
                               int batch2, int n_query, int tokens, float* out, int record, void* stream) {
  API_CK(c, need_final(c));
  hipStream_t st = S(stream);
  if (!layer || !hidden || !out || batch2 < 1 || n_query < 1) { agd_set_error("attn_processor: bad argument"); return fail_ctx(c); }
  std::string name(layer);
  if (name.compare(0, 5, "unet.") != 0) name = "unet." + name;
  const bool is_attn2 = ends_with(name, "attn2"), is_attn1 = ends_with(name, "attn1");
  if (!is_attn1 && !is_attn2) { agd_set_error("attn_processor: unknown layer '%s' (neither an attn1 nor an attn2 module)", layer); return fail_ctx(c); }
  const std::string t = name.substr(0, name.size() - 5);
  auto it = c->xl_idx.find(t + "attn2");
  if (it == c->xl_idx.end()) { agd_set_error("attn_processor: unknown layer '%s'", layer); return fail_ctx(c); }
  XLayer& xl = c->xl[it->second];
  const int C = xl.C, M = batch2 * n_query;
  if (is_attn2) {
    // cross-attention needs the text context (a self-attention call on attn2 would feed C-wide rows to a ctx_dim-wide to_k)
    if (ctx_emb) API_CK(c, agd_set_context(c, ctx_emb, batch2, tokens, stream));
    else if (c->ctx_T <= 0) { agd_set_error("attn_processor: attn2 called without encoder_hidden_states and no context set"); return fail_ctx(c); }
    if (c->ctx_B2 != batch2) { agd_set_error("attn_processor: context batch %d != %d", c->ctx_B2, batch2); return fail_ctx(c); }
  } else if (ctx_emb) {
    agd_set_error("attn_processor: attn1 is the self-attention module (encoder_hidden_states must be NULL)"); return fail_ctx(c);
  }
  c->arena.release(0);
  bf16_t* x = (bf16_t*)c->arena.alloc((size_t)M * C * 2);
  bf16_t* q = (bf16_t*)c->arena.alloc((size_t)M * (is_attn2 ? 1 : 3) * C * 2);
  bf16_t* att = (bf16_t*)c->arena.alloc((size_t)M * C * 2);
  if (!x || !q || !att) return fail_ctx(c);
  API_CK(c, launch_f32_to_bf16(hidden, x, (long long)M * C, st));
  const std::string an = t + (is_attn2 ? "attn2." : "attn1.");
  const WMat* wo = getW(c, an + "to_out.0.weight"); const float* bo = getV(c, an + "to_out.0.bias");
  if (!wo || !bo) return fail_ctx(c);
  if (is_attn2) {
    const WMat* wq = getW(c, an + "to_q.weight"); if (!wq) return fail_ctx(c);
    { GemmOpt o; API_CK(c, run_conv(c, st, x, C, nullptr, 0, 1, 1, M, *wq, 1, q, o, c->zero_page)); }
    API_CK(c, cross_attention(c, st, xl, q, batch2, n_query, att, record != 0, attn_mask));
  } else {
    const WMat* wqkv = getW(c, t + "attn1.qkv"); if (!wqkv) return fail_ctx(c);
    { GemmOpt o; API_CK(c, run_conv(c, st, x, C, nullptr, 0, 1, 1, M, *wqkv, 1, q, o, c->zero_page)); }
    AttnP a{}; a.q = q; a.k = q + C; a.v = q + 2 * C; a.o = att;
    a.ldq = a.ldk = a.ldv = 3 * C; a.ldo = C; a.sq = a.sk = a.sv = (long long)n_query * 3 * C; a.so = (long long)n_query * C;
    a.B = batch2; a.H = xl.heads; a.D = C / xl.heads; a.Nq = n_query; a.Nk = n_query; a.scale = 1.0f / sqrtf((float)(C / xl.heads));
    a.mask = attn_mask;
    API_CK(c, run_attention(c, st, PC_ATTN_SELF, a));
  }
  { GemmOpt o; o.bias = bo; o.out_f32 = 1; API_CK(c, run_conv(c, st, att, C, nullptr, 0, 1, 1, M, *wo, 1, out, o, c->zero_page)); }
  return 0;
}

// ---- training-mode seam -------------------------------------------------------------------
AGD_API int agd_hook_num_maps(agd_ctx* c) { return c ? (int)c->hook_recs.size() : 0; }

// hook.py:110-112: the k-th map appended since the last clear() (train mode keeps them all): dims = {B', T, n_query}
AGD_API int agd_hook_map_dims(agd_ctx* c, int k, int* dims) {
  if (!c || !dims || k < 0 || k >= (int)c->hook_recs.size()) { agd_set_error("hook_map: no recorded map %d (have %d; is_train keeps per-call maps)", k, c ? (int)c->hook_recs.size() : 0); return fail_ctx(c); }
  dims[0] = c->hook_recs[k].Bp; dims[1] = c->hook_recs[k].T; dims[2] = c->hook_recs[k].N;
  return 0;
}
AGD_API int agd_hook_map(agd_ctx* c, int k, float* out, void* stream) {
  API_CK(c, need_final(c));
  if (k < 0 || k >= (int)c->hook_recs.size()) { agd_set_error("hook_map: no recorded map %d (have %d)", k, (int)c->hook_recs.size()); return fail_ctx(c); }
  const auto& r = c->hook_recs[k];
  if (hipMemcpyAsync(out, (const char*)c->hook_storeb.p + r.off, (size_t)r.Bp * r.T * r.N * 4, hipMemcpyDeviceToDevice, S(stream)) != hipSuccess) { agd_set_error("hook_map copy"); return fail_ctx(c); }
  return 0;
}

// finetune_sd_token.py:1046-1066 for ONE recorded map [B][T][P] (P = h*w): loss_out [B][2] = {bg, fg} terms of each sample
// (already times coef = reg_weight / #samples with an object), dmap [B][T][P] (may be NULL) = their gradient w.r.t. the map.
// obj/fg/bg_idx: device int [B], obj < 0 skips the sample (no object in the image, :1048).
AGD_API int agd_op_attn_reg_loss(const float* map, int B, int T, int P, const int* obj_idx, const int* fg_idx, const int* bg_idx, float coef,
                                 float* loss_out, float* dmap, void* stream) {
  if (!map || !obj_idx || !fg_idx || !bg_idx || !loss_out || B < 0 || T < 1 || P < 1) { agd_set_error("attn_reg_loss: bad argument"); return -1; }
  CK(launch_attn_reg_loss(map, B, T, P, obj_idx, fg_idx, bg_idx, coef, loss_out, dmap, S(stream)));
  return 0;
}

static int transposed(agd_ctx* c, hipStream_t st, const WMat& w, DBuf& buf, WMat& out) {
  if (out.w) return 0;
  const int K = w.taps * w.Cpad;
  CK(buf.ensure((size_t)w.N * K * 2));
  CK(launch_transpose_bf16(w.w, w.N, K, buf.as<bf16_t>(), st));
  out.w = buf.as<bf16_t>(); out.N = K; out.Cin = w.N; out.Cpad = w.N; out.taps = 1;
  return 0;
}

// Backward of one cross-attention call of the seam (hook.py:91-120) w.r.t. its inputs:
//   d_out  [B2, N, C] fp32 or NULL: gradient of the returned hidden_states
//   d_map  [B', T, N] fp32 or NULL: gradient of the recorded head-mean map (hook.py:55; B' = B2 in train mode, the
//          conditional half B2/2 otherwise, hook.py:48-49) -- e.g. from agd_op_attn_reg_loss
//   -> d_hidden [B2, N, C] and d_ctx [B2, T, ctx_dim] fp32 (either may be NULL)
// P is recomputed from Q = to_q(hidden) and the cached K/V of `ctx_emb` (bf16 operands, fp32 arithmetic, like the forward).
AGD_API int agd_attn_processor_backward(agd_ctx* c, const char* layer, const float* hidden, const float* ctx_emb, const float* d_out,
                                        const float* d_map, int is_train, int batch2, int n_query, int tokens, float* d_hidden, float* d_ctx,
                                        void* stream) {
  API_CK(c, need_final(c));
  hipStream_t st = S(stream);
  if (!layer || !hidden || (!d_out && !d_map) || batch2 < 1 || n_query < 1) { agd_set_error("attn_processor_backward: bad argument"); return fail_ctx(c); }
  std::string name(layer);
  if (name.compare(0, 5, "unet.") != 0) name = "unet." + name;
  auto it = c->xl_idx.find(name);
  if (it == c->xl_idx.end() || !ends_with(name, "attn2")) { agd_set_error("attn_processor_backward: unknown cross-attention layer '%s'", layer); return fail_ctx(c); }
  XLayer& xl = c->xl[it->second];
  if (ctx_emb) API_CK(c, agd_set_context(c, ctx_emb, batch2, tokens, stream));
  if (c->ctx_B2 != batch2 || c->ctx_T < 1) { agd_set_error("attn_processor_backward: context batch %d != %d", c->ctx_B2, batch2); return fail_ctx(c); }
  const int C = xl.C, H = xl.heads, D = C / H, T = c->ctx_T, M = batch2 * n_query, Dc = c->cfg.cross_attention_dim;
  const std::string t = xl.name.substr(0, xl.name.size() - 5);
  const WMat* wq = getW(c, t + "attn2.to_q.weight"); const WMat* wo = getW(c, t + "attn2.to_out.0.weight");
  if (!wq || !wo) return fail_ctx(c);
  API_CK(c, transposed(c, st, *wq, xl.wqTb, xl.wqT)); API_CK(c, transposed(c, st, xl.wkv, xl.wkvTb, xl.wkvT)); API_CK(c, transposed(c, st, *wo, xl.woTb, xl.woT));
  API_CK(c, c->bwd_wsb.ensure((size_t)attention_backward_ws_floats(batch2, H, D, n_query, T) * 4));
  c->arena.release(0);
  bf16_t* x = (bf16_t*)c->arena.alloc((size_t)M * C * 2); bf16_t* q = (bf16_t*)c->arena.alloc((size_t)M * C * 2);
  bf16_t* dy = d_out ? (bf16_t*)c->arena.alloc((size_t)M * C * 2) : nullptr; bf16_t* dO = d_out ? (bf16_t*)c->arena.alloc((size_t)M * C * 2) : nullptr;
  bf16_t* dq = (bf16_t*)c->arena.alloc((size_t)M * C * 2); bf16_t* dkv = (bf16_t*)c->arena.alloc((size_t)batch2 * T * 2 * C * 2);
  if (!x || !q || !dq || !dkv || (d_out && (!dy || !dO))) return fail_ctx(c);
  API_CK(c, launch_f32_to_bf16(hidden, x, (long long)M * C, st));
  { GemmOpt o; API_CK(c, run_conv(c, st, x, C, nullptr, 0, 1, 1, M, *wq, 1, q, o, c->zero_page)); }                  // Q = to_q(hidden), hook.py:93
  if (d_out) {                                                                                                       // dO = d_out . Wo  (hook.py:118)
    API_CK(c, launch_f32_to_bf16(d_out, dy, (long long)M * C, st));
    GemmOpt o; API_CK(c, run_conv(c, st, dy, C, nullptr, 0, 1, 1, M, xl.woT, 1, dO, o, c->zero_page));
  }
  API_CK(c, launch_attention_backward(q, xl.kv, dO, d_map, is_train ? 0 : batch2 / 2, batch2, H, D, n_query, T, 1.0f / sqrtf((float)D), dq, dkv,
                                      c->bwd_wsb.as<float>(), st));
  if (d_hidden) { GemmOpt o; o.out_f32 = 1; API_CK(c, run_conv(c, st, dq, C, nullptr, 0, 1, 1, M, xl.wqT, 1, d_hidden, o, c->zero_page)); }          // dX = dQ . Wq
  if (d_ctx) { GemmOpt o; o.out_f32 = 1; API_CK(c, run_conv(c, st, dkv, 2 * C, nullptr, 0, 1, 1, batch2 * T, xl.wkvT, 1, d_ctx, o, c->zero_page)); } // dCtx = [dK | dV] . Wkv
  (void)Dc;
  return 0;
}

AGD_API int agd_cross_attn(agd_ctx* c, const char* layer, const float* hidden, const float* ctx_emb, int batch2, int n_query,
                           int tokens, float* out, int record, void* stream) {
  return agd_attn_processor(c, layer, hidden, ctx_emb, nullptr, batch2, n_query, tokens, out, record, stream);
}

// ---- profiling ------------------------------------------------------------------------
AGD_API int agd_profile_begin(agd_ctx* c) {
  if (!c) return -1;
  c->prof.clear(); c->ev_used = 0; c->prof_on = true;
  for (int i = 0; i < AGD_N_CLASSES; ++i) c->launches[i] = 0;
  return 0;
}
// per class: Sum of HIP-event ms, algorithmic flop and HBM bytes, launches, and roof_ms = Sum over launches of
// max(flop / mfma_peak, bytes / hbm_peak) -- the time the launch's BINDING roof allows (short-K GEMMs are HBM-bound)
AGD_API int agd_profile_end_ex(agd_ctx* c, double mfma_peak_flops, double hbm_peak_bytes, double* ms, double* flops, double* bytes,
                               double* roof_ms, double* roof_ms_hbm_bound, long long* launches) {
  if (!c || !ms || !flops || !bytes || !roof_ms || !launches) return -1;
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  for (int i = 0; i < AGD_N_CLASSES; ++i) { ms[i] = 0; flops[i] = 0; bytes[i] = 0; roof_ms[i] = 0; launches[i] = c->launches[i]; if (roof_ms_hbm_bound) roof_ms_hbm_bound[i] = 0; }
  for (auto& pe : c->prof) {
    float t = 0; hipEventElapsedTime(&t, pe.a, pe.b);
    ms[pe.cls] += t; flops[pe.cls] += pe.flops; bytes[pe.cls] += pe.bytes;
    const double tm = mfma_peak_flops > 0 ? pe.flops / mfma_peak_flops * 1e3 : 0, th = hbm_peak_bytes > 0 ? pe.bytes / hbm_peak_bytes * 1e3 : 0;
    roof_ms[pe.cls] += tm > th ? tm : th;
    if (roof_ms_hbm_bound && th > tm) roof_ms_hbm_bound[pe.cls] += th;
  }
  c->prof_on = false; c->prof.clear(); c->ev_used = 0;
  return 0;
}
AGD_API int agd_profile_end(agd_ctx* c, double* ms, double* flops, long long* launches) {
  double by[AGD_N_CLASSES], rf[AGD_N_CLASSES];
  return agd_profile_end_ex(c, 0, 0, ms, flops, by, rf, nullptr, launches);
}

// ---------------------------------------------------------------------------------------
// single-op entry points (fp32 in/out; temp device buffers per call; used by parity tests)
// ---------------------------------------------------------------------------------------
struct Tmp {
  std::vector<void*> v;
  template <typename T> T* get(size_t n) { void* p = nullptr; if (hipMalloc(&p, n * sizeof(T) + 256) != hipSuccess) { agd_set_error("tmp alloc failed"); return nullptr; } v.push_back(p); return (T*)p; }
  ~Tmp() { for (void* p : v) hipFree(p); }
};
static bf16_t* op_zero_page() {
  static bf16_t* z = nullptr;
  if (!z) { hipMalloc((void**)&z, 4096); hipMemset(z, 0, 4096); }
  return z;
}

// NCHW fp32 <-> NHWC bf16 (padded) helpers built from the library kernels
static int to_nhwc_bf16(const float* x, bf16_t* y, int B, int C, int HW, int Cpad, hipStream_t st) { return launch_prep_latents(x, y, B, C, HW, Cpad, 1, 1.0f, st); }

AGD_API int agd_op_conv2d_ex(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W, int Cout,
                                int ksize, int stride, int pad, int upsample, int flags, void* stream);
AGD_API int agd_op_conv2d(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W, int Cout,
                             int ksize, int stride, int pad, int upsample, void* stream) {
  return agd_op_conv2d_ex(x, w, bias, y, B, Cin, H, W, Cout, ksize, stride, pad, upsample, 0, stream);
}
// flags bit 0: 3x3 stride-1 launches take the row-halo kernel (igemm_halo.h) where it applies; bits 1..3: the 8-phase kernel
// (igemm8p.h) -- 2 = where the launcher would pick it, 4 / 8 = force its 256- / 160-wide tile
AGD_API int agd_op_conv2d_ex(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W, int Cout,
                                int ksize, int stride, int pad, int upsample, int flags, void* stream) {
  hipStream_t st = S(stream); Tmp tmp;
  if (pad != (ksize == 3 ? 1 : 0)) { agd_set_error("op_conv2d: pad must be 1 for 3x3, 0 for 1x1"); return -1; }
  const int Cpad = (Cin + 63) / 64 * 64, taps = ksize * ksize, up = upsample ? 2 : 1;
  const int Ho = (H * up + 2 * pad - ksize) / stride + 1, Wo = (W * up + 2 * pad - ksize) / stride + 1;
  bf16_t* xb = tmp.get<bf16_t>((size_t)B * H * W * Cpad); bf16_t* wb = tmp.get<bf16_t>((size_t)Cout * taps * Cpad);
  float* yn = tmp.get<float>((size_t)B * Ho * Wo * Cout);
  if (!xb || !wb || !yn) return -1;
  CK(to_nhwc_bf16(x, xb, B, Cin, H * W, Cpad, st));
  CK(launch_convert_weight(w, wb, Cout, Cin, taps, Cpad, 0, st));
  WMat wm; wm.w = wb; wm.N = Cout; wm.Cin = Cin; wm.Cpad = Cpad; wm.taps = taps;
  GemmOpt o; o.bias = bias; o.stride = stride; o.up = up; o.out_f32 = 1; o.halo = flags & 1; o.p8 = (flags & 4) ? 2 : (flags & 8) ? 3 : (flags & 2) ? 1 : 0;
  o.smap = (flags & 16) ? 1 : 0;
  o.pc = ((flags >> 7) & 15) | ((flags >> 8) & 48); // the producer / consumer kernels (igemm_pc.h, igemm_pch.h): IgemmP::pc mask bits 0..3 in bits 7..10, bits 4 / 5 in bits 12 / 13
  o.xcd_block = (flags >> 11) & 1;                   // XCD-aware tile blocks (igemm.hip pick_xcd_block)
  if ((flags & 64) && upsample && ksize == 3 && stride == 1) {     // the upsampling conv as four 2x2 phase convs (IgemmP::ups4); bf16 output (that form's only one), widened afterwards
    bf16_t* w4 = tmp.get<bf16_t>((size_t)4 * Cout * 4 * Cpad); float* b4 = tmp.get<float>((size_t)4 * Cout); bf16_t* yb = tmp.get<bf16_t>((size_t)B * Ho * Wo * Cout);
    if (!w4 || !b4 || !yb) return -1;
    CK(launch_upsample_phase_weight(wb, w4, Cout, Cpad, st));
    for (int ph = 0; ph < 4; ++ph) {
      if (bias) { if (hipMemcpyAsync(b4 + (size_t)ph * Cout, bias, Cout * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) { agd_set_error("op_conv2d: bias copy"); return -1; } }
      else if (hipMemsetAsync(b4 + (size_t)ph * Cout, 0, Cout * sizeof(float), st) != hipSuccess) { agd_set_error("op_conv2d: bias clear"); return -1; }
    }
    WMat wm4; wm4.w = w4; wm4.N = 4 * Cout; wm4.Cin = Cin; wm4.Cpad = Cpad; wm4.taps = 4;
    GemmOpt o4; o4.bias = b4; o4.ups4 = Cout; o4.hout = H; o4.wout = W; o4.ldo = Cout; o4.pad = 0; o4.p8 = o.p8;
    CK(run_conv(nullptr, st, xb, Cpad, nullptr, 0, B, H, W, wm4, 2, yb, o4, op_zero_page()));
    CK(launch_bf16_to_f32(yb, yn, (long long)B * Ho * Wo * Cout, st));
  } else
  CK(run_conv(nullptr, st, xb, Cpad, nullptr, 0, B, H, W, wm, ksize, yn, o, op_zero_page()));
  CK(launch_nchw_from_nhwc_f32(yn, Cout, y, B, Cout, Ho * Wo, st));
  hipStreamSynchronize(st);
  return 0;
}

AGD_API int agd_op_linear(const float* x, const float* w, const float* bias, const float* residual, float* y, int M, int K, int N,
                             int geglu, void* stream) {
  hipStream_t st = S(stream); Tmp tmp;
  const int flags = geglu; geglu &= 1;            // bit 0: GEGLU; bits 1..3: the 8-phase kernel, as in agd_op_conv2d_ex
  if (K % 64) { agd_set_error("op_linear: K must be a multiple of 64"); return -1; }
  const int Nout = geglu ? N / 2 : N;
  bf16_t* xb = tmp.get<bf16_t>((size_t)M * K); bf16_t* wb = tmp.get<bf16_t>((size_t)N * K);
  bf16_t* rb = residual ? tmp.get<bf16_t>((size_t)M * Nout) : nullptr;
  if (!xb || !wb || (residual && !rb)) return -1;
  CK(launch_f32_to_bf16(x, xb, (long long)M * K, st));
  CK(launch_convert_weight(w, wb, N, K, 1, K, geglu ? 16 : 0, st));
  if (residual) CK(launch_f32_to_bf16(residual, rb, (long long)M * Nout, st));
  WMat wm; wm.w = wb; wm.N = N; wm.Cin = K; wm.Cpad = K; wm.taps = 1;
  GemmOpt o; o.bias = bias; o.residual = rb; o.geglu = geglu; o.out_f32 = 1; o.p8 = (flags & 4) ? 2 : (flags & 8) ? 3 : (flags & 2) ? 1 : 0;
  o.kg2 = (flags & 32) ? 1 : 0;                      // two K groups of waves per workgroup where the launcher's 64-row unsplit tiles apply
  o.pc = ((flags >> 7) & 15) | ((flags >> 8) & 48); // the producer / consumer kernel (igemm_pc.h): IgemmP::pc mask bits 0..3 in bits 7..10, bits 4 / 5 in bits 12 / 13
  o.xcd_block = (flags >> 11) & 1;                   // XCD-aware tile blocks (igemm.hip pick_xcd_block)
  if (flags & 16) {                                  // the weight-streaming kernel (igemm_wreg.h); bf16 output (that kernel's only form), widened afterwards
    const int ni = geglu ? 4 : 2;
    if (N % (ni * 64)) { agd_set_error("op_linear: the weight-streaming kernel needs N %% %d == 0", ni * 64); return -1; }
    wm.wfrag = tmp.get<bf16_t>((size_t)N * K); bf16_t* yb = tmp.get<bf16_t>((size_t)M * Nout); if (!wm.wfrag || !yb) return -1;
    CK(launch_frag_order_w(wb, wm.wfrag, N, K, ni, K, st));
    wm.wfrag_ni = ni; o.wreg = (flags & 64) ? 7 : 3; o.out_f32 = 0;
    CK(run_conv(nullptr, st, xb, K, nullptr, 0, 1, 1, M, wm, 1, yb, o, op_zero_page()));
    CK(launch_bf16_to_f32(yb, y, (long long)M * Nout, st));
    hipStreamSynchronize(st);
    return 0;
  }
  CK(run_conv(nullptr, st, xb, K, nullptr, 0, 1, 1, M, wm, 1, y, o, op_zero_page()));
  hipStreamSynchronize(st);
  return 0;
}

AGD_API int agd_op_groupnorm(const float* x, const float* gamma, const float* beta, float* y, int B, int C, int HW, int groups,
                                float eps, int silu, void* stream) {
  hipStream_t st = S(stream); Tmp tmp;
  bf16_t* xb = tmp.get<bf16_t>((size_t)B * HW * C); bf16_t* yb = tmp.get<bf16_t>((size_t)B * HW * C);
  float* ws = tmp.get<float>((size_t)groupnorm_ws_floats(B, C, HW, groups)); float* yf = tmp.get<float>((size_t)B * HW * C);
  if (!xb || !yb || !ws || !yf) return -1;
  CK(to_nhwc_bf16(x, xb, B, C, HW, C, st));
  GroupNormP g{}; g.x0 = xb; g.C0 = C; g.y = yb; g.gamma = gamma; g.beta = beta; g.B = B; g.HW = HW; g.groups = groups; g.eps = eps; g.silu = silu; g.ws = ws;
  CK(launch_groupnorm(g, st));
  CK(launch_bf16_to_f32(yb, yf, (long long)B * HW * C, st));
  CK(launch_nchw_from_nhwc_f32(yf, C, y, B, C, HW, st));
  hipStreamSynchronize(st);
  return 0;
}

// conv3x3 (+bias) -> GroupNorm(+SiLU) as the graph walk chains them: with fused != 0 the conv launch leaves per-channel partial
// sums and the GroupNorm skips its statistics pass; with fused == 0 the two-kernel GroupNorm runs on the same conv output.
// y_nchw fp32 [B, Cout, H, W].  (Test entry point for the producer-statistics path.)
AGD_API int agd_op_conv_groupnorm(const float* x, const float* w, const float* bias, const float* gamma, const float* beta, float* y, int B,
                                  int Cin, int H, int W, int Cout, int groups, float eps, int silu, int fused, void* stream) {
  hipStream_t st = S(stream); Tmp tmp;
  const int Cpad = (Cin + 63) / 64 * 64, HW = H * W;
  bf16_t* xb = tmp.get<bf16_t>((size_t)B * HW * Cpad); bf16_t* wb = tmp.get<bf16_t>((size_t)Cout * 9 * Cpad);
  bf16_t* hb = tmp.get<bf16_t>((size_t)B * HW * Cout); bf16_t* yb = tmp.get<bf16_t>((size_t)B * HW * Cout);
  float* part = tmp.get<float>((size_t)((long long)B * HW / 64 + 1) * Cout * 2); float* yf = tmp.get<float>((size_t)B * HW * Cout);
  float* ws = tmp.get<float>((size_t)groupnorm_ws_floats(B, Cout, HW, groups));
  if (!xb || !wb || !hb || !yb || !part || !yf || !ws) return -1;
  CK(to_nhwc_bf16(x, xb, B, Cin, HW, Cpad, st));
  CK(launch_convert_weight(w, wb, Cout, Cin, 9, Cpad, 0, st));
  IgemmP p{};
  p.src0 = xb; p.C0 = Cpad; p.Hin = H; p.Win = W; p.Hout = H; p.Wout = W; p.ksize = 3; p.stride = 1; p.pad = 1; p.up = 1;
  p.W = wb; p.bias = bias; p.bias_mode = bias ? 1 : 0; p.N = Cout; p.K = 9 * Cpad; p.M = B * HW; p.ldr = Cout; p.out = hb; p.ldo = Cout;
  p.alpha = 1.f; p.batch = 1; p.zero_page = op_zero_page();
  p.p8 = (fused & 4) ? 2 : (fused & 8) ? 3 : (fused & 2) ? 1 : 0; fused &= 1;      // bits 1..3: the 8-phase kernel, as in agd_op_conv2d_ex
  int bm = 0;
  if (fused) {
    int cfg[3] = {0, 0, 0};
    CK(igemm_query(p, cfg));
    bm = cfg[0];
    if (cfg[2] != 1 || bm < 1 || HW % bm || HW % 64) { agd_set_error("op_conv_groupnorm: shape not eligible for producer statistics (tile %d, splits %d)", cfg[0], cfg[2]); return -1; }
    p.colstat_out = part; p.colstat_rows = HW;
  }
  CK(launch_igemm(p, st));
  GroupNormP g{}; g.x0 = hb; g.C0 = Cout; g.y = yb; g.gamma = gamma; g.beta = beta; g.B = B; g.HW = HW; g.groups = groups; g.eps = eps; g.silu = silu; g.ws = ws;
  if (fused) { g.part0 = part; g.bm0 = bm; }
  CK(launch_groupnorm(g, st));
  CK(launch_bf16_to_f32(yb, yf, (long long)B * HW * Cout, st));
  CK(launch_nchw_from_nhwc_f32(yf, Cout, y, B, Cout, HW, st));
  hipStreamSynchronize(st);
  return 0;
}

AGD_API int agd_op_layernorm(const float* x, const float* gamma, const float* beta, float* y, int rows, int C, float eps, void* stream) {
  hipStream_t st = S(stream); Tmp tmp;
  bf16_t* xb = tmp.get<bf16_t>((size_t)rows * C); bf16_t* yb = tmp.get<bf16_t>((size_t)rows * C);
  if (!xb || !yb) return -1;
  CK(launch_f32_to_bf16(x, xb, (long long)rows * C, st));
  CK(launch_layernorm(xb, yb, gamma, beta, rows, C, eps, st));
  CK(launch_bf16_to_f32(yb, y, (long long)rows * C, st));
  hipStreamSynchronize(st);
  return 0;
}

// y = x + ff.net.2(GEGLU(ff.net.0(LayerNorm(x)))) through the fused row-panel kernel (tblock.hip); w1 [8C][C] (values then gates),
// b1 [8C], w2 [C][4C], b2 [C], x / y [M][C] fp32 (x is rounded to bf16 first, as the residual stream is stored)
AGD_API int agd_op_ff_fused(const float* x, const float* gamma, const float* beta, const float* w1, const float* b1, const float* w2,
                            const float* b2, float* y, int M, int C, float eps, void* stream) {
  hipStream_t st = S(stream); Tmp tmp;
  if (C != 320) { agd_set_error("op_ff_fused: C = %d (320 only)", C); return -1; }
  const int H8 = 8 * C, H4 = 4 * C;
  bf16_t* xb = tmp.get<bf16_t>((size_t)M * C); bf16_t* yb = tmp.get<bf16_t>((size_t)M * C);
  bf16_t* w1b = tmp.get<bf16_t>((size_t)H8 * C); bf16_t* w1l = tmp.get<bf16_t>((size_t)H8 * C); bf16_t* w1f = tmp.get<bf16_t>((size_t)H8 * C);
  bf16_t* w2b = tmp.get<bf16_t>((size_t)C * H4); bf16_t* w2f = tmp.get<bf16_t>((size_t)C * H4);
  float* cs = tmp.get<float>(H8); float* bf = tmp.get<float>(H8);
  if (!xb || !yb || !w1b || !w1l || !w1f || !w2b || !w2f || !cs || !bf) return -1;
  CK(launch_f32_to_bf16(x, xb, (long long)M * C, st));
  CK(launch_convert_weight(w1, w1b, H8, C, 1, C, 16, st));
  CK(launch_ln_fold_weight(w1b, gamma, beta, b1, H8, C, 16, w1l, cs, bf, st));
  CK(launch_frag_order_w1(w1l, w1f, C, H4, st));
  CK(launch_convert_weight(w2, w2b, C, H4, 1, H4, 0, st));
  CK(launch_frag_order_w(w2b, w2f, C, H4, C / 64, 128, st));
  FFusedP fp{}; fp.h = xb; fp.out = yb; fp.w1f = w1f; fp.cs1 = cs; fp.b1 = bf; fp.w2f = w2f; fp.b2 = b2; fp.M = M; fp.ln_eps = eps;
  CK(launch_ff_fused(fp, C, st));
  CK(launch_bf16_to_f32(yb, y, (long long)M * C, st));
  hipStreamSynchronize(st);
  return 0;
}

// y = x + to_out(attention(to_q(LayerNorm(x)), k, v)) through the fused row-panel kernel (tblock.hip): x / y [B * HW][C] fp32, wq / wo
// [C][C], bo [C], kv [B][T][2C] (projected context: K columns then V columns); probs_sum (optional) [B][T][HW] = probabilities summed
// over the heads (the recorder's head-group form, all images recording)
AGD_API int agd_op_attn_chain(const float* x, const float* gamma, const float* beta, const float* wq, const float* kv, const float* wo,
                              const float* bo, float* y, float* probs_sum, int B, int HW, int T, int C, int heads, float eps, void* stream) {
  hipStream_t st = S(stream); Tmp tmp;
  const long long M = (long long)B * HW;
  bf16_t* xb = tmp.get<bf16_t>((size_t)M * C); bf16_t* yb = tmp.get<bf16_t>((size_t)M * C); bf16_t* kvb = tmp.get<bf16_t>((size_t)B * T * 2 * C);
  bf16_t* wqb = tmp.get<bf16_t>((size_t)C * C); bf16_t* wqf = tmp.get<bf16_t>((size_t)C * C);
  bf16_t* wob = tmp.get<bf16_t>((size_t)C * C); bf16_t* wof = tmp.get<bf16_t>((size_t)C * C);
  if (!xb || !yb || !kvb || !wqb || !wqf || !wob || !wof) return -1;
  if (C != 320 && C != 640) { agd_set_error("op_attn_chain: C %d", C); return -1; }
  CK(launch_f32_to_bf16(x, xb, M * C, st));
  CK(launch_f32_to_bf16(kv, kvb, (long long)B * T * 2 * C, st));
  CK(launch_convert_weight(wq, wqb, C, C, 1, C, 0, st)); CK(launch_frag_order_w(wqb, wqf, C, C, 5, C, st));
  CK(launch_convert_weight(wo, wob, C, C, 1, C, 0, st)); CK(launch_frag_order_w(wob, wof, C, C, 5, C, st));
  AttnChainP ap{}; ap.h = xb; ap.out = yb; ap.gamma = gamma; ap.beta = beta; ap.ln_eps = eps; ap.wqf = wqf; ap.wof = wof; ap.bo = bo;
  ap.rows32 = (heads >> 8) & 1; heads &= 255;            // (bit 8 of `heads`: the 32-row panel form of the C = 640 kernel, tests)
  ap.kv = kvb; ap.ldkv = 2 * C; ap.skv = (long long)T * 2 * C; ap.M = (int)M; ap.HW = HW; ap.T = T; ap.scale = 1.0f / sqrtf((float)(C / heads));
  if (probs_sum) {
    if (hipMemsetAsync(probs_sum, 0, (size_t)B * T * HW * 4, st) != hipSuccess) { agd_set_error("memset probs"); return -1; }
    ap.record = 1; ap.rec = probs_sum; ap.rec_b0 = 0; ap.rec_T = T; ap.rec_hpb = heads; ap.rec_head_stride = (long long)T * HW; ap.rec_img_stride = (long long)T * HW;
  }
  CK(launch_attn_chain(ap, C, heads, st));
  CK(launch_bf16_to_f32(yb, y, M * C, st));
  hipStreamSynchronize(st);
  return 0;
}

// x + to_out(attention(to_q(LayerNorm(x)), k, v)) through the pre-multiplied form (xattn_pre.hip): x [B][HW][C] fp32, wq / wo [C][C], bo [C], kv [B][T][2C];
// probs (optional) [B][heads][T][HW] = every head's probabilities (the recorder's per-(image, head) rows, all images recording)
AGD_API int agd_op_xattn_premul(const float* x, const float* gamma, const float* beta, const float* wq, const float* kv, const float* wo,
                                const float* bo, float* y, float* probs, int B, int HW, int T, int C, int heads, float eps, void* stream) {
  hipStream_t st = S(stream); Tmp tmp;
  const long long M = (long long)B * HW;
  if (heads < 1 || C % heads || (C / heads) % 8 || C % 160 || C % 64 || HW % 64 || T < 1 || T > XATTN_TP || (heads * XATTN_TP) % 64) { agd_set_error("op_xattn_premul: C %d heads %d HW %d T %d", C, heads, HW, T); return -1; }
  const size_t HT = (size_t)heads * XATTN_TP;
  bf16_t* xb = tmp.get<bf16_t>((size_t)M * C); bf16_t* yb = tmp.get<bf16_t>((size_t)M * C); bf16_t* kvb = tmp.get<bf16_t>((size_t)B * T * 2 * C);
  bf16_t* wqb = tmp.get<bf16_t>((size_t)C * C); bf16_t* wqT = tmp.get<bf16_t>((size_t)C * C); bf16_t* wob = tmp.get<bf16_t>((size_t)C * C);
  float* wqbeta = tmp.get<float>(C); float* rst = tmp.get<float>((size_t)M * 2);
  bf16_t* kpp = tmp.get<bf16_t>((size_t)B * HT * C); bf16_t* vpp = tmp.get<bf16_t>((size_t)B * HT * C); float* kcs = tmp.get<float>((size_t)B * HT * 2);
  bf16_t* P = tmp.get<bf16_t>((size_t)M * HT);
  if (!xb || !yb || !kvb || !wqb || !wqT || !wob || !wqbeta || !rst || !kpp || !vpp || !kcs || !P) return -1;
  CK(launch_f32_to_bf16(x, xb, M * C, st));
  CK(launch_f32_to_bf16(kv, kvb, (long long)B * T * 2 * C, st));
  CK(launch_convert_weight(wq, wqb, C, C, 1, C, 0, st)); CK(launch_transpose_bf16(wqb, C, C, wqT, st));
  CK(launch_convert_weight(wo, wob, C, C, 1, C, 0, st));
  CK(launch_matvec_bf16(wqb, beta, wqbeta, C, C, st));
  CK(launch_rowstat_bf16(xb, rst, (int)M, C, st));
  XattnPremulP pm{}; pm.kv = kvb; pm.ldkv = 2 * C; pm.skv = (long long)T * 2 * C; pm.wqT = wqT; pm.wo = wob; pm.gamma = gamma; pm.wqb = wqbeta;
  pm.B = B; pm.T = T; pm.C = C; pm.H = heads; pm.scale = 1.0f / sqrtf((float)(C / heads)); pm.kpp = kpp; pm.kcs = kcs; pm.kbs = kcs + (size_t)B * HT; pm.vpp = vpp;
  CK(launch_xattn_premul(pm, st));
  XattnSP sp{}; sp.x = xb; sp.ln_stats = rst; sp.ln_slots = 1; sp.ln_invC = 1.0f / (float)C; sp.ln_eps = eps; sp.kpp = kpp; sp.kcs = pm.kcs; sp.kbs = pm.kbs; sp.P = P;
  sp.M = (int)M; sp.HW = HW; sp.C = C; sp.H = heads; sp.T = T;
  if (probs) {
    if (hipMemsetAsync(probs, 0, (size_t)B * heads * T * HW * 4, st) != hipSuccess) { agd_set_error("memset probs"); return -1; }
    sp.rec = probs; sp.rec_b0 = 0; sp.rec_T = T; sp.rec_head_stride = (long long)T * HW; sp.rec_img_stride = sp.rec_head_stride * heads;
  }
  CK(launch_xattn_s(sp, st));
  WMat wv; wv.w = vpp; wv.N = C; wv.Cin = (int)HT; wv.Cpad = (int)HT; wv.taps = 1;
  GemmOpt oo; oo.bias = bo; oo.residual = xb; oo.w_per_image = 1;
  CK(run_conv(nullptr, st, P, (int)HT, nullptr, 0, B, 1, HW, wv, 1, yb, oo, op_zero_page()));
  CK(launch_bf16_to_f32(yb, y, M * C, st));
  hipStreamSynchronize(st);
  return 0;
}

AGD_API int agd_op_attention(const float* q, const float* k, const float* v, float* o, int B, int H, int D, int Nq, int Nk,
                                float scale, float* probs_out, void* stream) {
  hipStream_t st = S(stream); Tmp tmp;
  const int C = H * D;
  bf16_t* qb = tmp.get<bf16_t>((size_t)B * Nq * C); bf16_t* kb = tmp.get<bf16_t>((size_t)B * Nk * C);
  bf16_t* vb = tmp.get<bf16_t>((size_t)B * Nk * C); bf16_t* ob = tmp.get<bf16_t>((size_t)B * Nq * C);
  if (!qb || !kb || !vb || !ob) return -1;
  CK(launch_f32_to_bf16(q, qb, (long long)B * Nq * C, st));
  CK(launch_f32_to_bf16(k, kb, (long long)B * Nk * C, st));
  CK(launch_f32_to_bf16(v, vb, (long long)B * Nk * C, st));
  AttnP a{}; a.q = qb; a.k = kb; a.v = vb; a.o = ob; a.ldq = a.ldk = a.ldv = a.ldo = C;
  a.sq = (long long)Nq * C; a.so = a.sq; a.sk = (long long)Nk * C; a.sv = a.sk;
  a.B = B; a.H = H; a.D = D; a.Nq = Nq; a.Nk = Nk; a.scale = scale;
  if (probs_out) {
    if (hipMemsetAsync(probs_out, 0, (size_t)B * H * Nk * Nq * 4, st) != hipSuccess) { agd_set_error("memset probs"); return -1; }
    a.record_mode = 1; a.rec_b0 = 0; a.rec = probs_out; a.rec_T = Nk;
    a.rec_head_stride = (long long)Nk * Nq; a.rec_img_stride = a.rec_head_stride * H;
  }
  CK(launch_attention(a, st));
  CK(launch_bf16_to_f32(ob, o, (long long)B * Nq * C, st));
  hipStreamSynchronize(st);
  return 0;
}

// as agd_op_attention with the probabilities summed over the heads: probs_sum_out [B][Nk][Nq] (attention.hip RECORD 2)
AGD_API int agd_op_attention_headsum(const float* q, const float* k, const float* v, float* o, int B, int H, int D, int Nq, int Nk,
                                     float scale, float* probs_sum_out, void* stream) {
  hipStream_t st = S(stream); Tmp tmp;
  const int C = H * D;
  if (!probs_sum_out) { agd_set_error("attention_headsum: probs_sum_out is required"); return -1; }
  bf16_t* qb = tmp.get<bf16_t>((size_t)B * Nq * C); bf16_t* kb = tmp.get<bf16_t>((size_t)B * Nk * C);
  bf16_t* vb = tmp.get<bf16_t>((size_t)B * Nk * C); bf16_t* ob = tmp.get<bf16_t>((size_t)B * Nq * C);
  if (!qb || !kb || !vb || !ob) return -1;
  CK(launch_f32_to_bf16(q, qb, (long long)B * Nq * C, st));
  CK(launch_f32_to_bf16(k, kb, (long long)B * Nk * C, st));
  CK(launch_f32_to_bf16(v, vb, (long long)B * Nk * C, st));
  AttnP a{}; a.q = qb; a.k = kb; a.v = vb; a.o = ob; a.ldq = a.ldk = a.ldv = a.ldo = C;
  a.sq = (long long)Nq * C; a.so = a.sq; a.sk = (long long)Nk * C; a.sv = a.sk;
  a.B = B; a.H = H; a.D = D; a.Nq = Nq; a.Nk = Nk; a.scale = scale;
  if (hipMemsetAsync(probs_sum_out, 0, (size_t)B * Nk * Nq * 4, st) != hipSuccess) { agd_set_error("memset probs"); return -1; }
  a.record_mode = 3; a.rec_hpb = H; a.rec_b0 = 0; a.rec = probs_sum_out; a.rec_T = Nk; a.rec_img_stride = (long long)Nk * Nq;
  CK(launch_attention(a, st));
  CK(launch_bf16_to_f32(ob, o, (long long)B * Nq * C, st));
  hipStreamSynchronize(st);
  return 0;
}

AGD_API int agd_op_bicubic_clamp_mean(const float* maps, int n_maps, int T, int side, int S_, float* out, void* stream) {
  hipStream_t st = S(stream);
  // n_maps accumulators of [T][side][side]: treat as one layer with n_maps "heads"
  HeatLayer h; h.acc = maps; h.side = side; h.heads = n_maps; h.head_stride = (long long)T * side * side; h.img_stride = 0;
  CK(launch_daam_global(&h, 1, n_maps, T, S_, 0, out, st));
  hipStreamSynchronize(st);
  return 0;
}

#ifdef AGD_EXPERIMENTS   // the micro-benchmark entry points exist only in the experiments library (make exp): tools/kb*.py
// ---------------------------------------------------------------------------------------
// kernel micro-benchmarks (random bf16 operands; HIP-event timing on the launch stream)
// ---------------------------------------------------------------------------------------
__global__ void fill_random_bf16(bf16_t* p, long long n, unsigned seed, float scale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = f2bf(((float)(h & 0xFFFF) / 32768.0f - 1.0f) * scale);
  }
}
static void fill_rand(bf16_t* p, long long n, unsigned seed, float scale) {
  hipLaunchKernelGGL(fill_random_bf16, dim3(4096), dim3(256), 0, 0, p, n, seed, scale);
}

AGD_API int agd_bench_conv(int B, int H, int W, int C0, int C1, int Cout, int ksize, int stride, int up, int geglu,
                              int with_residual, int iters, double* ms_out) {
  Tmp tmp;
  const int Ctot = C0 + C1, taps = ksize * ksize, pad = ksize == 3 ? 1 : 0;
  const int Ho = (H * up + 2 * pad - ksize) / stride + 1, Wo = (W * up + 2 * pad - ksize) / stride + 1;
  const long long M = (long long)B * Ho * Wo;
  const int Nout = (geglu & 1) ? Cout / 2 : Cout;
  bf16_t* x0 = tmp.get<bf16_t>((size_t)B * H * W * C0); bf16_t* x1 = C1 ? tmp.get<bf16_t>((size_t)B * H * W * C1) : nullptr;
  bf16_t* w = tmp.get<bf16_t>((size_t)Cout * taps * Ctot); bf16_t* y = tmp.get<bf16_t>((size_t)M * Nout);
  bf16_t* r = with_residual ? tmp.get<bf16_t>((size_t)M * Nout) : nullptr;
  float* bias = tmp.get<float>(Cout);
  if (!x0 || !w || !y || !bias || (C1 && !x1) || (with_residual && !r)) return -1;
  fill_rand(x0, (long long)B * H * W * C0, 1, 1.0f); if (x1) fill_rand(x1, (long long)B * H * W * C1, 2, 1.0f);
  fill_rand(w, (long long)Cout * taps * Ctot, 3, 0.05f); if (r) fill_rand(r, M * Nout, 4, 1.0f);
  hipMemset(bias, 0, Cout * 4);
  WMat wm; wm.w = w; wm.N = Cout; wm.Cin = Ctot; wm.Cpad = Ctot; wm.taps = taps;
  // geglu bit 1 = GEGLU; bit 2 = also emit LayerNorm row statistics (producer); bit 4 = LayerNorm-folded consumer epilogue
  const int mode = geglu; geglu &= 1;
  GemmOpt o; o.bias = bias; o.stride = stride; o.up = up; o.geglu = geglu; o.residual = r; o.halo = (mode & 8) ? 1 : 0;
  o.p8 = (mode & 32) ? 2 : (mode & 64) ? 3 : (mode & 16) ? 1 : 0;
  o.smap = (mode & 256) ? 1 : 0;
  o.kg2 = (mode & 512) ? 1 : 0;
  o.pc = ((mode >> 11) & 15) | ((mode >> 12) & 48);  // producer / consumer kernels (igemm_pc.h, igemm_pch.h): IgemmP::pc mask bits 0..3 in bits 11..14, bit 4 in bit 16
  o.xcd_block = (mode >> 15) & 1;                    // XCD-aware tile blocks
  if (mode & 128) {                                  // weight-streaming kernel (igemm_wreg.h): the matrix once more in fragment order
    const int ni = geglu ? 4 : 2;
    wm.wfrag = tmp.get<bf16_t>((size_t)Cout * taps * Ctot); if (!wm.wfrag) return -1;
    if (taps != 1 || C1) { agd_set_error("bench: wreg is for plain 1x1 launches"); return -1; }
    CK(launch_frag_order_w(w, wm.wfrag, Cout, Ctot, ni, Ctot, 0));
    wm.wfrag_ni = ni; o.wreg = (mode & 1024) ? 7 : 3;
  }
  float* stats = nullptr; float* cs = nullptr;
  if (mode & 6) {
    int cfg[3] = {0, 0, 0}; GemmOpt qo = o; qo.query_cfg = cfg; qo.want_rowstat = (mode & 2) ? 1 : 0;
    CK(run_conv(nullptr, 0, x0, C0, x1, C1, B, H, W, wm, ksize, y, qo, op_zero_page()));
    const int slots = (mode & 2) ? (Cout + cfg[1] - 1) / cfg[1] : 2;
    stats = tmp.get<float>((size_t)M * slots * 2); cs = tmp.get<float>(Cout);
    if (!stats || !cs) return -1;
    hipMemset(stats, 0, (size_t)M * slots * 2 * 4); hipMemset(cs, 0, Cout * 4);
    if (mode & 2) { o.rowstat_out = stats; o.rowstat_slots = slots; }
    if (mode & 4) { o.ln_stats = stats; o.ln_slots = slots; o.ln_cs = cs; o.ln_invC = 1.0f / Ctot; o.ln_eps = 1e-5f; }
  }
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) CK(run_conv(nullptr, 0, x0, C0, x1, C1, B, H, W, wm, ksize, y, o, op_zero_page()));
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) CK(run_conv(nullptr, 0, x0, C0, x1, C1, B, H, W, wm, ksize, y, o, op_zero_page()));
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float t = 0; hipEventElapsedTime(&t, a, b);
  *ms_out = t / iters;
  hipEventDestroy(a); hipEventDestroy(b);
  return 0;
}

// One 1x1 / 3x3 launch with COLD weights, as inside the UNet walk (1.7 GB of weights per forward against 256 MB of Infinity Cache):
// every iteration first overwrites a 1 GiB scratch (evicts L2 + Infinity Cache), rewrites the activation (hot, as after its
// producer), then times [optional streaming touch of the weight matrix] + the launch.  warm: 0 = cold weights, 1 = touch then launch
// (both timed), 2 = weights left hot (no flush of them: the touch runs untimed).  ms_out = mean of the timed regions.
AGD_API int agd_bench_conv_cold(int B, int H, int W, int C0, int Cout, int ksize, int geglu, int with_residual, int warm, int iters, double* ms_out) {
  Tmp tmp;
  const int taps = ksize * ksize;
  const long long M = (long long)B * H * W;
  const int Nout = geglu ? Cout / 2 : Cout;
  const size_t xn = (size_t)M * C0, wn = (size_t)Cout * taps * C0, flush_bytes = (size_t)1 << 30;
  bf16_t* x0 = tmp.get<bf16_t>(xn); bf16_t* xs = tmp.get<bf16_t>(xn); bf16_t* w = tmp.get<bf16_t>(wn); bf16_t* y = tmp.get<bf16_t>((size_t)M * Nout);
  bf16_t* r = with_residual ? tmp.get<bf16_t>((size_t)M * Nout) : nullptr;
  float* bias = tmp.get<float>(Cout); char* scratch = tmp.get<char>(flush_bytes); unsigned* sink = tmp.get<unsigned>(64);
  if (!x0 || !xs || !w || !y || !bias || !scratch || !sink || (with_residual && !r)) return -1;
  fill_rand(xs, (long long)xn, 1, 1.0f); fill_rand(w, (long long)wn, 3, 0.05f); if (r) fill_rand(r, M * Nout, 4, 1.0f);
  hipMemset(bias, 0, Cout * 4);
  WMat wm; wm.w = w; wm.N = Cout; wm.Cin = C0; wm.Cpad = C0; wm.taps = taps;
  GemmOpt o; o.bias = bias; o.geglu = geglu; o.residual = r; o.warm = warm == 3 ? 3 : 0;      // warm 3: cold weights, in-kernel warm-up
  o.halo = 1; o.p8 = 1; o.smap = 1;                                                             // the walk's dispatch options (ctx defaults)
  // the intervening layer (warm >= 4): conv3x3 640 -> 640 on 8 x 32 x 32
  bf16_t* ix = tmp.get<bf16_t>((size_t)8 * 1024 * 640); bf16_t* iy = tmp.get<bf16_t>((size_t)8 * 1024 * 640); bf16_t* iw = tmp.get<bf16_t>((size_t)640 * 9 * 640);
  if (!ix || !iy || !iw) return -1;
  fill_rand(ix, (long long)8 * 1024 * 640, 11, 1.0f); fill_rand(iw, (long long)640 * 9 * 640, 12, 0.05f);
  WMat iwm; iwm.w = iw; iwm.N = 640; iwm.Cin = 640; iwm.Cpad = 640; iwm.taps = 9;
  GemmOpt io; io.bias = bias;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  double tot = 0;
  for (int i = 0; i < iters + 1; ++i) {
    hipMemsetAsync(scratch, i, flush_bytes, 0);
    hipMemcpyAsync(x0, xs, xn * 2, hipMemcpyDeviceToDevice, 0);
    if (r) hipMemcpyAsync(y, r, (size_t)M * Nout * 2, hipMemcpyDeviceToDevice, 0);       // touches the residual / output lines
    if (warm == 2 || warm >= 4) hipLaunchKernelGGL(touch_kernel, dim3(1024), dim3(256), 0, 0, (const u32x4*)w, (long long)(wn * 2 / 16), sink);
    // warm 4 / 5 / 6: the touch is followed by 1 / 2 / 4 intervening launches of a typical layer (an L1 3x3 conv: ~100 MB through the
    // caches each) before the timed launch: does the Infinity Cache still hold the matrix?
    for (int k = 0; k < (warm == 4 ? 1 : warm == 5 ? 2 : warm == 6 ? 4 : 0); ++k) CK(run_conv(nullptr, 0, ix, 640, nullptr, 0, 8, 32, 32, iwm, 3, iy, io, op_zero_page()));
    hipEventRecord(a, 0);
    if (warm == 1) hipLaunchKernelGGL(touch_kernel, dim3(1024), dim3(256), 0, 0, (const u32x4*)w, (long long)(wn * 2 / 16), sink);
    CK(run_conv(nullptr, 0, x0, C0, nullptr, 0, B, H, W, wm, ksize, y, o, op_zero_page()));
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float t = 0; hipEventElapsedTime(&t, a, b);
    if (i > 0) tot += t;
  }
  *ms_out = tot / iters;
  hipEventDestroy(a); hipEventDestroy(b);
  return 0;
}

// fused row-panel kernels (tblock.hip) at C = 320: kind 0 = feed-forward, 1 = attn2 chain (B images of HW tokens, 77 keys, the upper half of the
// images recording head-summed probabilities, as in a CFG forward)
AGD_API int agd_bench_tblock(int kind, int B, int HW, int iters, double* ms_out) {
  Tmp tmp;
  const int C = (kind == 4 || kind == 5) ? 640 : 320, T = 77; const long long M = (long long)B * HW;      // kind 4 / 5: the attn2 chain at C = 640 (plain / from attn1.to_out)
  if (kind == 4 || kind == 5) kind = kind == 4 ? 1 : 3;
  bf16_t* h = tmp.get<bf16_t>((size_t)M * C); bf16_t* o = tmp.get<bf16_t>((size_t)M * C);
  bf16_t* w1 = tmp.get<bf16_t>((size_t)8 * C * C); bf16_t* w1f = tmp.get<bf16_t>((size_t)8 * C * C);
  bf16_t* w2 = tmp.get<bf16_t>((size_t)4 * C * C); bf16_t* w2f = tmp.get<bf16_t>((size_t)4 * C * C);
  bf16_t* kv = tmp.get<bf16_t>((size_t)B * T * 2 * C);
  const int rslices = C == 640 ? 8 : 1;                 // recorder slices per image: per-head rows below latent resolution, one head-summed slice at it
  float* vec = tmp.get<float>((size_t)16 * C + 64); float* rec = tmp.get<float>((size_t)B * rslices * T * HW);
  if (!h || !o || !w1 || !w1f || !w2 || !w2f || !kv || !vec || !rec) return -1;
  fill_rand(h, M * C, 1, 1.0f); fill_rand(w1, 8LL * C * C, 2, 0.05f); fill_rand(w2, 4LL * C * C, 3, 0.03f); fill_rand(kv, (long long)B * T * 2 * C, 4, 1.0f);
  hipMemset(vec, 0, ((size_t)16 * C + 64) * 4); hipMemset(rec, 0, (size_t)B * rslices * T * HW * 4);
  FFusedP fp{}; AttnChainP ap{};
  if (kind == 0) {
    CK(launch_frag_order_w1(w1, w1f, C, 4 * C, 0)); CK(launch_frag_order_w(w2, w2f, C, 4 * C, C / 64, 128, 0));
    fp.h = h; fp.out = o; fp.w1f = w1f; fp.cs1 = vec; fp.b1 = vec + 8 * C; fp.w2f = w2f; fp.b2 = vec; fp.M = (int)M; fp.ln_eps = 1e-5f;
  } else {
    CK(launch_frag_order_w(w1, w1f, C, C, 5, C, 0)); CK(launch_frag_order_w(w2, w2f, C, C, 5, C, 0));
    ap.h = h; ap.out = o; ap.gamma = vec; ap.beta = vec; ap.ln_eps = 1e-5f; ap.wqf = w1f; ap.wof = w2f; ap.bo = vec; ap.kv = kv; ap.ldkv = 2 * C; ap.skv = (long long)T * 2 * C;
    ap.M = (int)M; ap.HW = HW; ap.T = T; ap.scale = 1.0f / sqrtf((float)(C / 8));
    ap.record = 1; ap.rec = rec; ap.rec_b0 = B / 2; ap.rec_T = T; ap.rec_hpb = 8 / rslices; ap.rec_head_stride = (long long)T * HW; ap.rec_img_stride = (long long)rslices * T * HW;
  }
  bf16_t* o2 = nullptr; float* cst = nullptr;
  if (kind == 2) {                                   // feed-forward + proj_out stage (+ column statistics)
    o2 = tmp.get<bf16_t>((size_t)M * C); cst = tmp.get<float>((size_t)(M / 128 + 1) * C * 2); if (!o2 || !cst) return -1;
    bf16_t* wpf = tmp.get<bf16_t>((size_t)C * C); if (!wpf) return -1;
    CK(launch_frag_order_w(w2, wpf, C, C, C / 64, C, 0));          // (any matrix will do)
    CK(launch_frag_order_w1(w1, w1f, C, 4 * C, 0)); CK(launch_frag_order_w(w2, w2f, C, 4 * C, C / 64, 128, 0));
    fp.h = h; fp.out = o; fp.w1f = w1f; fp.cs1 = vec; fp.b1 = vec + 8 * C; fp.w2f = w2f; fp.b2 = vec; fp.M = (int)M; fp.ln_eps = 1e-5f;
    fp.wpf = wpf; fp.bp = vec; fp.xres = o2; fp.pout = o; fp.colstat = cst;
  }
  if (kind == 3) {                                   // attn2 chain starting at attn1.to_out
    o2 = tmp.get<bf16_t>((size_t)M * C); bf16_t* wo1f = tmp.get<bf16_t>((size_t)C * C); if (!o2 || !wo1f) return -1;
    fill_rand(o2, M * C, 9, 1.0f);
    CK(launch_frag_order_w(w2, wo1f, C, C, 5, C, 0));
    ap.o1 = o2; ap.wo1f = wo1f; ap.bo1 = vec;
  }
  QkvChainP qp{};
  const bool qkvk = kind >= 6 && kind <= 9;            // (8 / 9: the same on round 6's schedule) the block head (GroupNorm inside -> proj_in -> norm1 -> q / k / v); 7: two co-resident 64-row workgroups per CU
  if (qkvk) {
    bf16_t* qkv = tmp.get<bf16_t>((size_t)M * 3 * C); float* part = tmp.get<float>((size_t)B * (HW / 128) * C * 2); float* gv = tmp.get<float>((size_t)4 * C);
    if (!qkv || !part || !gv) return -1;
    hipMemset(part, 0x3C, (size_t)B * (HW / 128) * C * 2 * 4); hipMemset(gv, 0x3C, (size_t)4 * C * 4);    // 0x3C3C3C3C = 0.0115f: finite, non-zero everywhere
    CK(launch_frag_order_w(w2, w2f, C, C, 5, C, 0)); CK(launch_frag_order_w(w1, w1f, 3 * C, C, 5, C, 0));
    qp.x = h; qp.wbf = w2f; qp.wb_stride = 0; qp.rowadd = gv; qp.rowadd_stride = 0; qp.h = o; qp.gamma = gv + C; qp.beta = gv + 2 * C; qp.ln_eps = 1e-5f;
    qp.wqkvf = w1f; qp.qkv = qkv; qp.M = (int)M; qp.HW = HW; qp.rows64 = kind == 7 || kind == 9; qp.sched2 = kind >= 8;
    qp.gn_part = part; qp.gn_bm = 128; qp.gn_groups = 32; qp.gn_eps = 1e-6f; qp.gn_gamma = gv + 3 * C; qp.gn_beta = gv;
  }
  auto run = [&]() { return qkvk ? launch_qkv_chain(qp, C, 0) : (kind == 0 || kind == 2) ? launch_ff_fused(fp, C, 0) : launch_attn_chain(ap, C, 8, 0); };
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) CK(run());
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) CK(run());
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float t = 0; hipEventElapsedTime(&t, a, b);
  *ms_out = t / iters;
  hipEventDestroy(a); hipEventDestroy(b);
  return 0;
}

AGD_API int agd_bench_attention(int B, int H, int D, int Nq, int Nk, int record, int iters, double* ms_out) {
  Tmp tmp;
  const int C = H * D;
  const bool self = (Nq == Nk);
  // self: packed qkv rows [3C]; cross: q [C], kv [2C]
  bf16_t* q = tmp.get<bf16_t>((size_t)B * Nq * (self ? 3 * C : C)); bf16_t* kv = self ? nullptr : tmp.get<bf16_t>((size_t)B * Nk * 2 * C);
  bf16_t* o = tmp.get<bf16_t>((size_t)B * Nq * C);
  float* rec = record ? tmp.get<float>((size_t)B * H * Nk * Nq) : nullptr;     // record: 1 per-head rows, 2 head-summed rows
  if (!q || !o || (!self && !kv) || (record && !rec)) return -1;
  fill_rand(q, (long long)B * Nq * (self ? 3 * C : C), 5, 1.0f); if (kv) fill_rand(kv, (long long)B * Nk * 2 * C, 6, 1.0f);
  if (rec) hipMemset(rec, 0, (size_t)B * H * Nk * Nq * 4);
  AttnP a{};
  if (self) { a.q = q; a.k = q + C; a.v = q + 2 * C; a.ldq = a.ldk = a.ldv = 3 * C; a.sq = a.sk = a.sv = (long long)Nq * 3 * C; }
  else { a.q = q; a.k = kv; a.v = kv + C; a.ldq = C; a.ldk = a.ldv = 2 * C; a.sq = (long long)Nq * C; a.sk = a.sv = (long long)Nk * 2 * C; }
  a.o = o; a.ldo = C; a.so = (long long)Nq * C; a.B = B; a.H = H; a.D = D; a.Nq = Nq; a.Nk = Nk; a.scale = 1.0f / sqrtf((float)D);
  // record: 1 per-head rows; 2, 3, 4 = head-group sums with 8, 4, 2 heads per workgroup
  if (record) { a.record_mode = record >= 2 ? 3 : 1; a.rec_hpb = record >= 2 ? (16 >> record < H ? 16 >> record : H) : 0; a.rec_b0 = B / 2; a.rec = rec;
                a.rec_T = Nk; a.rec_head_stride = (long long)Nk * Nq; a.rec_img_stride = a.rec_head_stride * (record >= 2 ? H / a.rec_hpb : H); }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) CK(launch_attention(a, 0));
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) CK(launch_attention(a, 0));
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float t = 0; hipEventElapsedTime(&t, e0, e1);
  *ms_out = t / iters;
  hipEventDestroy(e0); hipEventDestroy(e1);
  return 0;
}

// GroupNorm(+SiLU) of a [B][HW][C0 (+ C1 concatenated)] activation; fused_stats != 0 takes the one-kernel path that reads the
// producing igemm launches' per-(128-row tile, channel) partial sums (here: arbitrary values -- timing only)
AGD_API int agd_bench_groupnorm_ex(int B, int HW, int C0, int C1, int fused_stats, int iters, double* ms_out) {
  Tmp tmp;
  const int C = C0 + C1;
  if (B < 1 || HW < 1 || C0 < 8 || C1 < 0 || iters < 1 || !ms_out) { agd_set_error("bench_groupnorm: bad arguments"); return -1; }
  bf16_t* x0 = tmp.get<bf16_t>((size_t)B * HW * C0); bf16_t* x1 = C1 ? tmp.get<bf16_t>((size_t)B * HW * C1) : nullptr;
  bf16_t* y = tmp.get<bf16_t>((size_t)B * HW * C);
  float* ws = tmp.get<float>((size_t)groupnorm_ws_floats(B, C, HW, 32)); float* g = tmp.get<float>(2 * C);
  if (!x0 || (C1 && !x1) || !y || !ws || !g) return -1;
  fill_rand(x0, (long long)B * HW * C0, 7, 1.0f); if (C1) fill_rand(x1, (long long)B * HW * C1, 8, 1.0f);
  hipMemset(g, 0, 2 * C * 4);
  GroupNormP p{}; p.x0 = x0; p.x1 = x1; p.C0 = C0; p.C1 = C1; p.y = y; p.gamma = g; p.beta = g + C; p.B = B; p.HW = HW; p.groups = 32; p.eps = 1e-5f; p.silu = 1; p.ws = ws;
  if (fused_stats) {
    if (HW % 128) { agd_set_error("bench_groupnorm: fused_stats needs HW %% 128 == 0"); return -1; }
    const size_t n0 = (size_t)B * (HW / 128) * C0 * 2, n1 = (size_t)B * (HW / 128) * C1 * 2;
    float* p0 = tmp.get<float>(n0); float* p1 = C1 ? tmp.get<float>(n1) : nullptr;
    if (!p0 || (C1 && !p1)) return -1;
    hipMemset(p0, 0, n0 * 4); if (C1) hipMemset(p1, 0, n1 * 4);
    p.part0 = p0; p.part1 = p1; p.bm0 = 128; p.bm1 = C1 ? 128 : 0;
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) CK(launch_groupnorm(p, 0));
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) CK(launch_groupnorm(p, 0));
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float t = 0; hipEventElapsedTime(&t, e0, e1);
  *ms_out = t / iters;
  hipEventDestroy(e0); hipEventDestroy(e1);
  return 0;
}
AGD_API int agd_bench_groupnorm(int B, int HW, int C, int iters, double* ms_out) { return agd_bench_groupnorm_ex(B, HW, C, 0, 0, iters, ms_out); }
#endif  // AGD_EXPERIMENTS


// ---------------------------------------------------------------------------------------
// export entry points
// ---------------------------------------------------------------------------------------
namespace {
struct PilCoeffs { int in = 0, out = 0, ksize = 0; int* bounds = nullptr; int* kk = nullptr; };
double pil_bicubic(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}
// Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc (PRECISION_BITS = 22), cached per (in, out)
const PilCoeffs* pil_coeffs(int in_size, int out_size) {
  static std::vector<PilCoeffs> cache;
  for (auto& c : cache) if (c.in == in_size && c.out == out_size) return &c;
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  std::vector<int> bounds(2 * out_size), kk((size_t)out_size * ksize, 0);
  std::vector<double> w(ksize);
  const double ss = 1.0 / filterscale;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5); if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5); if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) { w[x] = pil_bicubic((x + xmin - center + 0.5) * ss); ww += w[x]; }
    for (int x = 0; x < xmax; ++x) {
      const double v = ww != 0.0 ? w[x] / ww : w[x];
      kk[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << 22)) : (int)(0.5 + v * (1 << 22));
    }
    bounds[2 * xx] = xmin; bounds[2 * xx + 1] = xmax;
  }
  PilCoeffs c; c.in = in_size; c.out = out_size; c.ksize = ksize;
  if (hipMalloc((void**)&c.bounds, bounds.size() * 4) != hipSuccess || hipMalloc((void**)&c.kk, kk.size() * 4) != hipSuccess) return nullptr;
  hipMemcpy(c.bounds, bounds.data(), bounds.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(c.kk, kk.data(), kk.size() * 4, hipMemcpyHostToDevice);
  cache.push_back(c);
  return &cache.back();
}
}  // namespace

AGD_API int agd_op_heatmap_u8(const float* hm, int n, int npix, unsigned char* out, void* stream) {
  CK(launch_heatmap_u8(hm, n, npix, out, S(stream)));
  return 0;
}

// in [n][H][W][C] uint8 -> out [n][oh][ow][C], == PIL Image.resize((ow, oh)) per image (default BICUBIC)
AGD_API int agd_op_resize_u8_pil(const unsigned char* in, int n, int H, int W, int C, int oh, int ow, unsigned char* out, void* stream) {
  hipStream_t st = S(stream);
  const unsigned char* src = in;
  unsigned char* tmp = nullptr;
  if (W != ow) {                                            // horizontal pass first
    const PilCoeffs* ch = pil_coeffs(W, ow); if (!ch) FAIL("resize: coefficient upload failed");
    unsigned char* dst = out;
    if (H != oh) { if (hipMalloc((void**)&tmp, (size_t)n * H * ow * C) != hipSuccess) FAIL("resize: tmp alloc"); dst = tmp; }
    CK(launch_pil_resample(src, dst, ch->bounds, ch->kk, ch->ksize, (long long)n * H, W, ow, C, st));
    src = dst;
  }
  if (H != oh) {                                            // then vertical
    const PilCoeffs* cv = pil_coeffs(H, oh); if (!cv) FAIL("resize: coefficient upload failed");
    CK(launch_pil_resample(src, out, cv->bounds, cv->kk, cv->ksize, n, H, oh, ow * C, st));
  } else if (W == ow) {
    hipMemcpyAsync(out, in, (size_t)n * H * W * C, hipMemcpyDeviceToDevice, st);
  }
  hipStreamSynchronize(st);
  if (tmp) hipFree(tmp);
  return 0;
}

AGD_API int agd_op_stack_heatmaps(const unsigned char* obj, const unsigned char* fg, const unsigned char* bg, long long npix,
                                     unsigned char* rgb, unsigned char* inv, void* stream) {
  CK(launch_stack_heatmaps(obj, fg, bg, npix, rgb, inv, S(stream)));
  return 0;
}


// ---------------------------------------------------------------------------------------
// CLIP text encoder: `pipeline.text_encoder(input_ids)[0]`
// ---------------------------------------------------------------------------------------
AGD_API int agd_text_set_embedding_row(agd_ctx* c, int token_id, const float* row) {
  API_CK(c, need_final(c));
  const WMat* te = getW(c, "text.embeddings.token_embedding.weight"); if (!te) return fail_ctx(c);
  if (token_id < 0 || token_id >= te->N + kTextExtraRows) { agd_set_error("text: token id %d out of range (vocab %d + %d)", token_id, te->N, kTextExtraRows); return fail_ctx(c); }
  Tmp tmp; float* d = tmp.get<float>(te->Cpad); if (!d) return fail_ctx(c);
  if (hipMemcpy(d, row, (size_t)te->Cin * 4, hipMemcpyDefault) != hipSuccess) { agd_set_error("text: row copy failed"); return fail_ctx(c); }
  API_CK(c, launch_f32_to_bf16(d, te->w + (size_t)token_id * te->Cpad, te->Cin, 0));
  hipDeviceSynchronize();
  return 0;
}

AGD_API int agd_text_encode(agd_ctx* c, const int* input_ids, int B, int T, float* out, void* stream) {
  API_CK(c, need_final(c));
  hipStream_t st = S(stream);
  const agd_config& g = c->cfg;
  if (g.text_layers <= 0) { agd_set_error("text encoder not configured"); return fail_ctx(c); }
  const int H = g.text_hidden, heads = g.text_heads, D = H / heads, M = B * T;
  if (T > g.text_max_pos || T > 96) { agd_set_error("text: %d tokens unsupported (max %d)", T, g.text_max_pos < 96 ? g.text_max_pos : 96); return fail_ctx(c); }
  const std::string tx = "text.";
  const WMat* te = getW(c, tx + "embeddings.token_embedding.weight"); const WMat* pe = getW(c, tx + "embeddings.position_embedding.weight");
  if (!te || !pe) return fail_ctx(c);
  c->arena.release(0);
  int* ids = (int*)c->arena.alloc((size_t)M * 4);
  bf16_t* x = (bf16_t*)c->arena.alloc((size_t)M * H * 2); bf16_t* h = (bf16_t*)c->arena.alloc((size_t)M * H * 2);
  bf16_t* qkv = (bf16_t*)c->arena.alloc((size_t)M * 3 * H * 2); bf16_t* att = (bf16_t*)c->arena.alloc((size_t)M * H * 2);
  bf16_t* ff = (bf16_t*)c->arena.alloc((size_t)M * g.text_intermediate * 2);
  if (!ids || !x || !h || !qkv || !att || !ff) return fail_ctx(c);
  if (hipMemcpyAsync(ids, input_ids, (size_t)M * 4, hipMemcpyDefault, st) != hipSuccess) { agd_set_error("text: ids copy failed"); return fail_ctx(c); }
  API_CK(c, launch_embed_gather(ids, te->w, pe->w, x, B, T, H, te->N + kTextExtraRows, st));
  for (int l = 0; l < g.text_layers; ++l) {
    const std::string L = tx + "encoder.layers." + std::to_string(l) + ".";
    const float* g1 = getV(c, L + "layer_norm1.weight"); const float* b1 = getV(c, L + "layer_norm1.bias");
    const float* g2 = getV(c, L + "layer_norm2.weight"); const float* b2 = getV(c, L + "layer_norm2.bias");
    const WMat* wqkv = getW(c, L + "self_attn.qkv.weight"); const float* bqkv = getV(c, L + "self_attn.qkv.bias");
    const WMat* wo = getW(c, L + "self_attn.out_proj.weight"); const float* bo = getV(c, L + "self_attn.out_proj.bias");
    const WMat* w1 = getW(c, L + "mlp.fc1.weight"); const float* bf1 = getV(c, L + "mlp.fc1.bias");
    const WMat* w2 = getW(c, L + "mlp.fc2.weight"); const float* bf2 = getV(c, L + "mlp.fc2.bias");
    if (!g1 || !b1 || !g2 || !b2 || !wqkv || !bqkv || !wo || !bo || !w1 || !bf1 || !w2 || !bf2) return fail_ctx(c);
    API_CK(c, launch_layernorm(x, h, g1, b1, M, H, g.text_eps, st));
    { GemmOpt o; o.bias = bqkv; API_CK(c, run_conv(c, st, h, H, nullptr, 0, 1, 1, M, *wqkv, 1, qkv, o, c->zero_page)); }
    { AttnP a{}; a.q = qkv; a.k = qkv + H; a.v = qkv + 2 * H; a.o = att; a.ldq = a.ldk = a.ldv = 3 * H; a.ldo = H;
      a.sq = a.sk = a.sv = (long long)T * 3 * H; a.so = (long long)T * H; a.B = B; a.H = heads; a.D = D; a.Nq = T; a.Nk = T;
      a.scale = 1.0f / sqrtf((float)D); a.causal = 1;
      API_CK(c, run_attention(c, st, PC_OTHER, a)); }
    { GemmOpt o; o.bias = bo; o.residual = x; API_CK(c, run_conv(c, st, att, H, nullptr, 0, 1, 1, M, *wo, 1, x, o, c->zero_page)); }
    API_CK(c, launch_layernorm(x, h, g2, b2, M, H, g.text_eps, st));
    { GemmOpt o; o.bias = bf1; o.act = g.text_act == 0 ? 2 : 3; API_CK(c, run_conv(c, st, h, H, nullptr, 0, 1, 1, M, *w1, 1, ff, o, c->zero_page)); }
    { GemmOpt o; o.bias = bf2; o.residual = x; API_CK(c, run_conv(c, st, ff, g.text_intermediate, nullptr, 0, 1, 1, M, *w2, 1, x, o, c->zero_page)); }
  }
  { const float* gf = getV(c, tx + "final_layer_norm.weight"); const float* bfn = getV(c, tx + "final_layer_norm.bias");
    if (!gf || !bfn) return fail_ctx(c);
    API_CK(c, launch_layernorm(x, h, gf, bfn, M, H, g.text_eps, st));
    API_CK(c, launch_bf16_to_f32(h, out, (long long)M * H, st)); }
  return 0;
}


// `vae.encode(image).latent_dist` moments: image fp32 NCHW [B,3,S,S] in [-1,1] -> mean, logvar fp32 NCHW [B,lc,L,L]
AGD_API int agd_vae_encode(agd_ctx* c, const float* image, int batch, int side, float* mean_out, float* logvar_out, void* stream) {
  API_CK(c, need_final(c));
  hipStream_t st = S(stream);
  const int lc = c->cfg.vae_latent_channels, L = side >> (c->cfg.vae_n_levels - 1);
  if (side % (1 << (c->cfg.vae_n_levels - 1)) || side % 8) { agd_set_error("vae_encode: side %d not divisible", side); return fail_ctx(c); }
  Tmp tmp;
  bf16_t* x = tmp.get<bf16_t>((size_t)batch * side * side * 64); float* mom = tmp.get<float>((size_t)batch * L * L * 2 * lc);
  float* mom_nchw = tmp.get<float>((size_t)batch * L * L * 2 * lc);
  if (!x || !mom || !mom_nchw) return fail_ctx(c);
  API_CK(c, launch_prep_latents(image, x, batch, c->cfg.vae_out_channels, side * side, 64, 1, 1.0f, st));
  API_CK(c, vae_encode_walk(c, st, x, batch, side, mom));
  API_CK(c, launch_nchw_from_nhwc_f32(mom, 2 * lc, mom_nchw, batch, 2 * lc, L * L, st));
  for (int b = 0; b < batch; ++b) {
    hipMemcpyAsync(mean_out + (size_t)b * lc * L * L, mom_nchw + (size_t)b * 2 * lc * L * L, (size_t)lc * L * L * 4, hipMemcpyDeviceToDevice, st);
    hipMemcpyAsync(logvar_out + (size_t)b * lc * L * L, mom_nchw + ((size_t)b * 2 + 1) * lc * L * L, (size_t)lc * L * L * 4, hipMemcpyDeviceToDevice, st);
  }
  hipStreamSynchronize(st);
  return 0;
}
