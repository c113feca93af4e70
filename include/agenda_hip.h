/* agenda_hip.h -- C ABI of libagenda_hip.so: the MI355X (gfx950) implementation of the AGenDA
 * data-generation hot path (Stable-Diffusion UNet denoise loop + DAAM cross-attention heat maps +
 * VAE decode).
 *
 * The reference (humansensinglab/AGenDA) has no FFI: the path sits behind three PYTHON surfaces
 * (SURVEY.md §8b).  Each entry point below names the reference interface it stands under:
 *   - diffusers `StableDiffusionPipeline.__call__`   reference data_generation/data_generation.py:59
 *   - `daam.trace(pipe)` / `compute_global_heat_map`  reference data_generation/data_generation.py:57,64
 *   - diffusers attention-processor protocol          reference data_generation/hook.py:83-122
 * The Python shim in agenda_amd/ binds these with ctypes.
 *
 * Conventions: plain pointers + sizes, no torch types.  Every call returns 0 on success, nonzero on
 * error (message via agd_last_error).  Tensors are caller-owned DEVICE pointers unless noted; the
 * library owns only its weights, workspace and heat-map accumulators.  `stream` is a hipStream_t
 * (NULL = default stream); calls are stream-ordered with no hidden syncs except where noted.
 * One ctx per GPU per process; calls on one ctx must be serialised by the caller.
 */
#ifndef AGENDA_HIP_H
#define AGENDA_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct agd_ctx agd_ctx;

#define AGD_MAX_LEVELS 8

/* Architecture description == the fields of unet/config.json + vae/config.json that
 * `StableDiffusionPipeline.from_pretrained` (data_generation.py:30) reads. */
typedef struct agd_config {
  int struct_size;                 /* sizeof(agd_config), ABI guard */
  int in_channels, out_channels, n_levels;
  int block_out_channels[AGD_MAX_LEVELS];
  int down_cross[AGD_MAX_LEVELS];  /* 1 = CrossAttnDownBlock2D at this level */
  int num_heads[AGD_MAX_LEVELS];
  int layers_per_block, cross_attention_dim, use_linear_projection, norm_num_groups;
  int vae_latent_channels, vae_out_channels, vae_n_levels;
  int vae_block_out_channels[AGD_MAX_LEVELS];
  int vae_layers_per_block, vae_norm_num_groups;
  float vae_scaling_factor;
  int max_tokens;                  /* 77 */
  int prediction_type;             /* 0 epsilon, 1 v_prediction */
  long long workspace_bytes;       /* activation arena; 0 = default (8 GiB) */
  /* CLIP text encoder (`pipeline.text_encoder`, data_generation.py:47-52); text_layers == 0: none loaded */
  int text_hidden, text_layers, text_heads, text_intermediate, text_vocab, text_max_pos;
  int text_act;                    /* 0 quick_gelu (SD-1.x CLIP ViT-L/14), 1 gelu (OpenCLIP, SD-2.x) */
  float text_eps;
} agd_config;

/* ---- lifetime ------------------------------------------------------------------------- */
agd_ctx* agd_create(int device_id, const agd_config* cfg);
void agd_destroy(agd_ctx* ctx);
const char* agd_last_error(agd_ctx* ctx);          /* ctx may be NULL (last global error) */

/* ---- weights: names are the diffusers state-dict keys, UNet keys prefixed "unet.", VAE keys
 * "vae." (`pipeline.unet` / `pipeline.vae`, data_generation.py:30).  `ptr` may be host or
 * device memory; dtype 0 = float32.  Synchronous. */
int agd_load_tensor(agd_ctx* ctx, const char* name, const void* ptr, int dtype, int ndim, const long long* shape);
int agd_finalize(agd_ctx* ctx);                    /* after the last agd_load_tensor */

/* ---- text context: `encoder_hidden_states` [2B, T, ctx_dim] fp32, rows [0,B) unconditional,
 * [B,2B) conditional (CFG order assumed by hook.py:48-49).  Projects K/V of every attn2 once. */
int agd_set_context(agd_ctx* ctx, const float* ctx_emb, int batch2, int tokens, void* stream);

/* ---- `text_encoder(input_ids)[0]` (CLIPTextModel.last_hidden_state): ids int32 [B, T] (host or device) ->
 * out fp32 [B, T, text_hidden].  Weights: "text." + transformers state-dict key.  Rows of the token embedding can
 * be (over)written -- the learned-token injection of data_generation.py:45-52 -- ids up to text_vocab + 255. */
int agd_text_encode(agd_ctx* ctx, const int* input_ids, int batch, int tokens, float* out, void* stream);
int agd_text_set_embedding_row(agd_ctx* ctx, int token_id, const float* row);

/* ---- `unet(sample, t, encoder_hidden_states).sample`: sample/out fp32 NCHW [B2,4,L,L] */
int agd_unet_forward(agd_ctx* ctx, const float* sample, int batch2, int latent_side, float timestep, float* out,
                     void* stream);
/* the same with one timestep PER IMAGE (host array of batch2 floats): the training call `unet(noisy_latents, timesteps, ehs)` of
 * finetune_sd_token.py:1027, where `timesteps` is a [bsz] tensor */
int agd_unet_forward_ts(agd_ctx* ctx, const float* sample, int batch2, int latent_side, const float* timesteps, float* out, void* stream);

/* ---- CFG combine + DDIM (eta 0) step on fp32 NCHW latents [B,4,L,L] in place;
 * eps is [2B,4,L,L] NCHW.  (`scheduler.step` inside pipeline.__call__) */
int agd_cfg_ddim_step(agd_ctx* ctx, const float* eps, float* latents, int batch, int latent_side, float guidance,
                      float alpha_t, float alpha_prev, void* stream);

/* ---- the whole denoise loop of `pipeline(prompt, num_inference_steps=..)` on device:
 * latents [B,4,L,L] fp32 in/out; per-step timesteps and alpha_cumprod (t, prev) from the host
 * scheduler.  Heat-map recording follows agd_record_config. */
int agd_denoise(agd_ctx* ctx, float* latents, int batch, int latent_side, int n_steps, const float* timesteps,
                const float* alpha_t, const float* alpha_prev, float guidance, void* stream);

/* ---- the same loop under the scheduler the reference actually runs: data_generation.py:59 calls the pipeline with the
 * checkpoint's default PNDMScheduler (skip_prk_steps: PLMS) x 20.  n_evals = steps + 1 model evaluations; per evaluation the
 * UNet timestep and the two `_get_prev_sample` coefficients (prev = sample_coeff * sample + eps_coeff * model_output) come
 * from the host scheduler (agenda_amd/scheduler.py PNDMScheduler); the multistep weights are applied on the device. */
int agd_denoise_plms(agd_ctx* ctx, float* latents, int batch, int latent_side, int n_evals, const float* timesteps,
                     const float* sample_coeff, const float* eps_coeff, float guidance, void* stream);

/* ---- `vae.decode(latents / scaling_factor)` + image post-process.
 * out_u8: [B, 8L, 8L, 3] uint8 (may be NULL); out_f32: [B, 8L, 8L, 3] fp32 in [-1,1] (may be NULL) */
int agd_vae_decode(agd_ctx* ctx, const float* latents, int batch, int latent_side, unsigned char* out_u8,
                   float* out_f32, void* stream);

/* ---- `vae.encode(image).latent_dist` (img2img front end; anchors: finetune_sd.py:764-765):
 * image fp32 NCHW [B,3,S,S] in [-1,1] -> mean and logvar fp32 NCHW [B,4,S/8,S/8].  Syncs. */
int agd_vae_encode(agd_ctx* ctx, const float* image, int batch, int side, float* mean_out, float* logvar_out, void* stream);

/* ---- per-ctx options (no environment variables steer the library).  "cfg_shared_prefix" (default 1): agd_denoise runs the
 * layers ahead of the first cross-attention once for the identical unconditional / conditional halves (bit-identical to 0).
 * "ln_fold", "gn_fused_stats" (default 1): LayerNorm / GroupNorm statistics produced by the GEMM that writes the activation.
 * "gn_proj_fold" (default 1; 0 off, 2: also C = 640): the GroupNorm in front of a transformer's proj_in (no activation in between) is
 * folded into per-image proj_in matrices where C <= 320 and the producer left its statistics; proj_in then reads the un-normalised activation.
 * "weight_touch" (default 3; 0 off): 1x1 weight matrices of at least that many MB are streamed through the caches by a read-only
 * kernel right in front of the launch that uses them (the UNet's 1.7 GB of weights never stay in the 256 MB Infinity Cache).
 * "conv_halo" (default 1): 3x3 stride-1 convolutions run the row-halo kernel (one LDS image of the tile's pixel rows serves the three
 * horizontal taps) where its geometry applies.
 * "weight_warm" (default 3; 0 off): the first workgroups of a launch stream its weight matrix through the caches before their main
 * loops -- 1: only launches whose weights outweigh their activations (each XCD its own slice, into its L2); 3: every launch with
 * >= 1 MB of weights and 1024 <= M <= 32768 (the touch launches then disappear).
 * "igemm8p" (default 1): launches with enough 256-row tiles (wide 1x1 projections, 3x3 convs with N a multiple of 256, the
 * upsampling convs) run the 8-wave / 8-phase implicit-GEMM kernel (igemm8p.h); 0 = the 4-wave kernels everywhere; tests: 2 / 3 / 4
 * force its 256-wide / 160-wide / any legal tile.
 * "tblock_fuse" (default 1791): fused row-panel kernels of the transformer blocks at C = 320 (tblock.hip) -- bit 0: norm3 -> GEGLU ->
 * ff.net.2 + residual as one launch, bit 1: norm2 -> to_q -> cross-attention (+ recorder) -> to_out + residual as one launch, bit 2: that
 * launch starts at attn1.to_out + residual, bit 3: the feed-forward launch ends with proj_out + residual (+ the next GroupNorm's sums), bit 4: proj_in (GroupNorm folded) -> norm1 ->
 * q / k / v projections as one launch, bit 5: the attn2 chain (bits 1, 2) for the C = 640 blocks of the 32 x 32 maps as well (64-row panels), bit 6: under `cfg_shared_prefix` the duplication of the shared rows happens inside the fused kernels (no copy launches), bit 7: the bit-4 launch applies the transformer's GroupNorm itself (statistics from the producer's partial sums, rows normalised in its LDS panel) instead of reading per-image folded matrices from a fold launch, bit 8 (not in the default: measured slower): that launch, GroupNorm inside, for the C = 640 blocks too, bit 9: inside the feed-forward launch (bits 0, 3) ff.net.2 and proj_out are pre-multiplied (Wp W2 at load time; the proj_out stage adds Wp . h on the same accumulators -- no intermediate h3), bit 10: the attn2 chain of the C = 640 blocks on 32-row panels where 64-row panels give fewer than 200 workgroups (M = 8192: 256 workgroups instead of 128).
 * "reduce_gn" (default 1): the slab-sum pass of a split-K conv also applies the GroupNorm (+ SiLU) that reads its output next (conv1 -> norm2 of a
 * ResnetBlock2D; conv2 -> the following module's norm where that reads this output alone); 0 = separate statistics / apply launches.
 * "conv_smap" (default 1): 3x3 convs of the 8 x 8 maps run the whole-images-resident kernel (igemm_smap.h).
 * "upsample_phases" (default 7; bit 0: the UNet's from 16 x 16 maps up, bit 1: the VAE decoder's, bit 2: the UNet's 8 x 8 -> 16 x 16 one): nearest-2x upsampling convs run as four 2x2 phase convs on the un-upsampled map in one launch (taps that coincide pre-summed at load
 * time, pixel-shuffled output rows: 4/9 of the MACs); 0 = the 3x3 conv with the upsample folded into its gather.
 * "ff_proj_fuse" (default 1): ff.net.2 (+ residual) and proj_out (+ block residual) as one GEMM with the pre-multiplied matrix [Wp W2 | Wp] over [hidden | h], in the transformer
 * blocks whose feed-forward is not the fused row-panel kernel; 0 = two launches.
 * "shortcut_fuse" (default 3): a UNet ResnetBlock2D's 1x1 conv_shortcut runs as extra K of its conv2 launch (the shortcut's output is never stored) -- bit 0: where conv2 is a
 * row-halo launch (the 64 x 64 .. 16 x 16 maps), bit 1: the 8 x 8 whole-images launches; 0 = its own launch, added as conv2's residual.
 * "wreg_mask" (default 3): the weight-streaming kernel (igemm_wreg.h: weight fragments straight to registers) for bit 0 = the GEGLU projection of the C = 1280 blocks at 16 x 16,
 * bit 1 = proj_in / proj_out of the C = 640 transformer blocks.
 * "igemm_kgroups" (default 1): the unsplit 1x1 launches on 64 x 64 tiles (8 x 8 maps: at most one workgroup per CU) run two K groups of four waves per workgroup.
 * "igemm_pc" (default 49): producer / consumer implicit GEMM (csrc/igemm_pc.h: loader waves issue the ring's LDS-DMA pieces, consumer waves read fragments one K step ahead and run the MFMAs) --
 * bit 0: the 1x1 launches on 64 x 160 tiles (one workgroup per CU: the 16 x 16 / 8 x 8 maps), bits 1 - 3 (measured slower, off): 3x3 convs of the 16 x 16 / 8 x 8 maps,
 * bit 4: the row-halo producer / consumer kernel (csrc/igemm_pch.h) for the 3x3 stride-1 convs whose 128 x 160 tiles (x K slices) make at most one workgroup per CU (the
 * 32 x 32 and 16 x 16 maps at UNet batch 8), fused conv_shortcut included; same tiles and summation order as the row-halo kernel: bit-identical;
 * bit 5: the plain 1x1 launches of one 128 x 160 tile per CU (M = 8192, N = 640: ff.net.2 + proj_out of the 32 x 32 blocks) on igemm_pc.h's 128-row form; bit-identical.
 * "xcd_block" (default 1): the igemm tile grid is cut into one a x b block of tiles per XCD, (a, b) minimising the XCD's L2 working set, instead of tiles / 8 consecutive tiles of the
 * A-major / W-major walk (same tiles, same results bit for bit).
 * "attn2_premul" (default 1, bit 1 -- also the head-dim-80 blocks -- off: a tie; read at the next agd_set_context): attn2 of the blocks with head dim >= 160 (SD-1.x: C = 1280, the 16 x 16 and 8 x 8 maps) runs against per-image
 * PRE-MULTIPLIED context matrices built once per agd_set_context (csrc/xattn_pre.hip; hook.py:91-120): S = LN(h) . K'' with K'' = gamma scale (k Wq), softmax + recorder in that GEMM's
 * epilogue, out = P . V'' + bo + h with V'' = Wo v -- two launches instead of to_q, the attention kernel and to_out; 0 = the kernel chain.  The hook.py recorder keeps the chain.
 * "side_stream" (default 0; measured slower, kept for the A/B): a resnet's 1x1 conv_shortcut runs on a second stream beside norm1 / conv1 / norm2. */
int agd_set_option(agd_ctx* ctx, const char* name, int value);

/* ---- heat-map recording (daam.trace / hook.py UNetCrossAttentionHooker state)
 * mode 0 off; 1 DAAM (per-layer/head time sums, mid block excluded, conditional half);
 * 2 HOOK (hook.py: head-mean per call, every attn2 incl. mid; is_train=1 keeps all batch rows).
 * rec_tokens: token rows recorded (<= tokens; rows beyond len(prompt)+2 are never read by daam); takes effect at the
 * next agd_record_reset, which (re)sizes the accumulators for it. */
int agd_record_config(agd_ctx* ctx, int mode, int is_train, int rec_tokens);
int agd_record_reset(agd_ctx* ctx, int batch, int latent_side, void* stream);   /* hooker.clear() / new trace */
/* daam `compute_global_heat_map()` for image `img`: out [rows, S, S] fp32 (rows <= rec_tokens). Stream-ordered. */
int agd_daam_global(agd_ctx* ctx, int img, int rows, float* out, void* stream);
/* hook.py `compute_global_heat_map()`: out [B', T, S, S]; returns -2 if nothing was recorded. Stream-ordered. */
int agd_hook_global(agd_ctx* ctx, float* out, void* stream);
int agd_hook_count(agd_ctx* ctx);
/* the map hook.py:110-112 appends for the most recent recorded call: out [B', T, h, w] (h*w = n_query). Stream-ordered. */
int agd_hook_last_map(agd_ctx* ctx, float* out, int n_query, void* stream);

/* ---- the processor seam, hook.py:83-122 in full: one call of the diffusers attention-processor protocol
 * `proc(attn, hidden_states, encoder_hidden_states=None, attention_mask=None)` for the UNet `Attention` module
 * `layer` (module path, e.g. "down_blocks.0.attentions.0.transformer_blocks.0.attn2" or "...attn1").
 *   ctx_emb != NULL on an attn2 module: cross-attention (is_cross_attn, hook.py:94-99); record != 0 feeds the recorder
 *   ctx_emb == NULL on an attn1 module: self-attention (encoder_hidden_states = hidden_states); records nothing
 *   attn_mask: additive fp32 [B2, keys] (hook.py:92 `prepare_attention_mask`, broadcast over heads and queries) or NULL
 * hidden/out fp32 [B2, N, C]; ctx_emb fp32 [B2, T, ctx_dim].  Returns to_out[0](softmax(scale QK^T + mask) V) + bias. */
int agd_attn_processor(agd_ctx* ctx, const char* layer, const float* hidden, const float* ctx_emb, const float* attn_mask,
                       int batch2, int n_query, int tokens, float* out, int record, void* stream);
/* the same for an attn2 module without a mask (kept for callers of the round-1 ABI) */
int agd_cross_attn(agd_ctx* ctx, const char* layer, const float* hidden, const float* ctx_emb, int batch2,
                   int n_query, int tokens, float* out, int record, void* stream);

/* ---- training-mode seam (what finetune_sd_token.py does with hook.py's recorder)
 * With agd_record_config(2, is_train = 1) every recorded attn2 call keeps its head-mean map [B', T, n_query] on the device
 * (hook.py:110-112 `cross_attn_maps.append`), in call order, until the next agd_record_reset (`hooker.clear()`,
 * finetune_sd_token.py:1024,1069). */
int agd_hook_reset(agd_ctx* ctx, int rows, int latent_side, void* stream);   /* clear() with B' given explicitly (no CFG pairing) */
int agd_hook_num_maps(agd_ctx* ctx);
int agd_hook_map_dims(agd_ctx* ctx, int k, int* dims3);                 /* {B', T, n_query} of the k-th kept map */
int agd_hook_map(agd_ctx* ctx, int k, float* out, void* stream);        /* out [B', T, n_query] fp32; stream-ordered */
/* The attention regulariser of finetune_sd_token.py:1046-1066 on ONE recorded map [B, T, P = h*w]: per sample the object /
 * foreground / background token rows are min-max normalised (+1e-8), L1-normalised, and compared:
 *   bg = coef * mean|(1 - o^)/sum(1 - o^) - b~| ,  fg = coef * mean|o^/sum(o^) - f~|      (coef = reg_weight / #object samples)
 * loss_out [B][2] = {bg, fg}; dmap [B, T, P] (may be NULL) = d(bg + fg)/d map (min/max paths included, as torch autograd).
 * obj/fg/bg_idx: device int [B]; obj_idx < 0 skips the sample (:1048).  Stream-ordered. */
int agd_op_attn_reg_loss(const float* map, int B, int T, int P, const int* obj_idx, const int* fg_idx, const int* bg_idx, float coef,
                         float* loss_out, float* dmap, void* stream);
/* Backward of one cross-attention call of the seam (hook.py:91-120) w.r.t. its inputs: given d_out [B2, N, C] (gradient of the
 * returned hidden_states; may be NULL) and/or d_map [B', T, N] (gradient of the recorded map; B' = B2 if is_train else B2/2;
 * may be NULL) -> d_hidden [B2, N, C] and d_ctx [B2, T, ctx_dim] fp32 (either may be NULL).  ctx_emb NULL = the cached context. */
int agd_attn_processor_backward(agd_ctx* ctx, const char* layer, const float* hidden, const float* ctx_emb, const float* d_out,
                                const float* d_map, int is_train, int batch2, int n_query, int tokens, float* d_hidden, float* d_ctx,
                                void* stream);

/* ---- single-op entry points (fp32 in/out, converted to the bf16 NHWC compute layout inside);
 * used by the parity tests, mirror torch.nn.functional signatures the oracle uses. */
int agd_op_conv2d(const float* x_nchw, const float* w, const float* bias, float* y_nchw, int B, int Cin, int H, int W,
                  int Cout, int ksize, int stride, int pad, int upsample, void* stream);
/* flags bit 0: 3x3 stride-1 launches take the row-halo kernel where it applies; bit 4: 8 x 8 maps take the whole-images-resident kernel
   (igemm_smap.h); bits 1..3: the 256-row 8-wave / 8-phase kernel --
   2 = where the launcher would pick it, 4 / 8 = force its 256- / 160-wide tile (agd_op_linear: the same bits in `geglu`, bit 0 = GEGLU,
   bit 4 = the weight-streaming 1x1 kernel, igemm_wreg.h: the matrix in fragment order straight to registers, bf16 output) */
int agd_op_conv2d_ex(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W, int Cout,
                     int ksize, int stride, int pad, int upsample, int flags, void* stream);
int agd_op_linear(const float* x, const float* w, const float* bias, const float* residual, float* y, int M, int K,
                  int N, int geglu, void* stream);
/* BasicTransformerBlock's feed-forward tail as ONE launch (tblock.hip): y = x + ff.net.2(GEGLU(ff.net.0(norm3(x)))), the op sequence
 * diffusers runs behind reference data_generation/data_generation.py:59; x / y [M][C] fp32, w1 [8C][C] (value rows then gate rows),
 * b1 [8C], w2 [C][4C], b2 [C]; C = 320 (the 64 x 64 maps' blocks) */
int agd_op_ff_fused(const float* x, const float* gamma, const float* beta, const float* w1, const float* b1, const float* w2,
                    const float* b2, float* y, int M, int C, float eps, void* stream);
/* BasicTransformerBlock's attn2 section as ONE launch (tblock.hip): y = x + to_out(softmax(scale to_q(norm2(x)) k^T) v), the processor body of
 * reference data_generation/hook.py:91-120 with the LayerNorm in front and the residual behind it; x / y [B * HW][C] fp32, kv [B][T][2C] =
 * the projected context (to_k columns then to_v columns), probs_sum (optional) [B][T][HW] = the probabilities summed over the heads;
 * C = 320 (HW a multiple of 128) or C = 640 (HW a multiple of 64), 8 heads, T <= 96 */
int agd_op_attn_chain(const float* x, const float* gamma, const float* beta, const float* wq, const float* kv, const float* wo,
                      const float* bo, float* y, float* probs_sum, int B, int HW, int T, int C, int heads, float eps, void* stream);
/* the same function through the pre-multiplied form of the C = 1280 blocks (csrc/xattn_pre.hip; hook.py:91-120 behind norm2): the context products
 * K'' = gamma scale (k Wq), V'' = Wo v are built first, then S = LN-folded x K''^T -> softmax -> P V''^T + bo + x as two GEMMs.
 * probs (may be NULL): [B][heads][T][HW], every head's probabilities (the recorder's per-(image, head) rows).  Needs HW % 64 == 0, T <= 80,
 * head dim % 8 == 0, C % 160 == 0 and C % 64 == 0 (i.e. C a multiple of 320), heads x 80 a multiple of 64. */
int agd_op_xattn_premul(const float* x, const float* gamma, const float* beta, const float* wq, const float* kv, const float* wo,
                        const float* bo, float* y, float* probs, int B, int HW, int T, int C, int heads, float eps, void* stream);
int agd_op_groupnorm(const float* x_nchw, const float* gamma, const float* beta, float* y_nchw, int B, int C, int HW,
                     int groups, float eps, int silu, void* stream);
/* conv3x3(+bias) -> GroupNorm(+SiLU), chained as the graph walk chains them; fused != 0: the conv launch emits the per-channel
 * partial sums and the GroupNorm skips its statistics pass (the production path), fused == 0: the two-kernel GroupNorm. */
int agd_op_conv_groupnorm(const float* x_nchw, const float* w, const float* bias, const float* gamma, const float* beta, float* y_nchw,
                          int B, int Cin, int H, int W, int Cout, int groups, float eps, int silu, int fused, void* stream);
int agd_op_layernorm(const float* x, const float* gamma, const float* beta, float* y, int rows, int C, float eps,
                     void* stream);
/* q [B,Nq,H*D], k/v [B,Nk,H*D] fp32 -> o [B,Nq,H*D]; probs_out (may be NULL): [B,H,Nk,Nq] fp32, needs Nk<=96 */
int agd_op_attention(const float* q, const float* k, const float* v, float* o, int B, int H, int D, int Nq, int Nk,
                     float scale, float* probs_out, void* stream);
/* same, the probabilities summed over the heads (the recording form of the daam layers at latent resolution): probs_sum_out [B,Nk,Nq] */
int agd_op_attention_headsum(const float* q, const float* k, const float* v, float* o, int B, int H, int D, int Nq, int Nk,
                             float scale, float* probs_sum_out, void* stream);
int agd_op_bicubic_clamp_mean(const float* maps, int n_maps, int T, int side, int S, float* out, void* stream);

/* ---- export path on device, bit-exact with the reference's host code:
 * heat map fp32 [n][npix] -> uint8 by per-map min-max (+1e-8), x255, truncation   (data_generation.py:82-84) */
int agd_op_heatmap_u8(const float* hm, int n, int npix, unsigned char* out, void* stream);
/* uint8 [n][H][W][C] -> [n][oh][ow][C], identical to PIL `Image.resize((ow, oh))` (default BICUBIC) per image
 * (data_generation.py:60,85).  Syncs. */
int agd_op_resize_u8_pil(const unsigned char* in, int n, int H, int W, int C, int oh, int ow, unsigned char* out, void* stream);
/* rgb [npix][3] = [obj, fg, 255 - bg]; inv [npix] = 255 - bg (may be NULL)        (postprocess_heatmap.py:44-48) */
int agd_op_stack_heatmaps(const unsigned char* obj, const unsigned char* fg, const unsigned char* bg, long long npix,
                          unsigned char* rgb, unsigned char* inv, void* stream);

/* ---- per-kernel-class timing (HIP events on the launch stream) */
#define AGD_N_CLASSES 11
int agd_profile_begin(agd_ctx* ctx);
int agd_profile_end(agd_ctx* ctx, double* ms, double* flops, long long* launches);  /* arrays of AGD_N_CLASSES; syncs */
/* the same plus, per class, the algorithmic HBM bytes and roof_ms = Sum over launches of max(flop / mfma_peak_flops,
 * bytes / hbm_peak_bytes) [ms]: the time each launch's BINDING roof allows (bench.py's per-class roofline fractions);
 * roof_ms_hbm_bound (may be NULL): the part of roof_ms that came from launches whose binding roof is HBM. */
int agd_profile_end_ex(agd_ctx* ctx, double mfma_peak_flops, double hbm_peak_bytes, double* ms, double* flops, double* bytes,
                       double* roof_ms, double* roof_ms_hbm_bound, long long* launches);
const char* agd_profile_class_name(int cls);

#ifdef AGD_EXPERIMENTS
/* Only in the experiments library (`make -C agenda_amd/csrc exp` -> agenda_amd/libagenda_hip_exp.so); the product library exports none of these. */
/* ---- kernel micro-benchmarks (tools/kbench.py): random bf16 operands, HIP-event timing of `iters` launches -> ms per launch */
int agd_bench_conv(int B, int H, int W, int C0, int C1, int Cout, int ksize, int stride, int up, int geglu, int with_residual,
                   int iters, double* ms_out);
int agd_bench_attention(int B, int H, int D, int Nq, int Nk, int record, int iters, double* ms_out);
/* one launch with cold weights (caches flushed per iteration); warm: 0 cold, 1 streaming touch of the weights timed with the launch, 2 weights hot */
int agd_bench_conv_cold(int B, int H, int W, int C0, int Cout, int ksize, int geglu, int with_residual, int warm, int iters, double* ms_out);
int agd_bench_groupnorm(int B, int HW, int C, int iters, double* ms_out);
int agd_bench_groupnorm_ex(int B, int HW, int C0, int C1, int fused_stats, int iters, double* ms_out);
#endif /* AGD_EXPERIMENTS */

const char* agd_version(void);

#ifdef __cplusplus
}
#endif
#endif
