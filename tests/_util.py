"""Test helpers (no reference content): a tiny character-level CLIP tokenizer directory in the on-disk format
`transformers.CLIPTokenizer.from_pretrained` reads (vocab.json + merges.txt), so `tokenizer/` loading, learned-token
injection and `compute_token_merge_indices` run against the real tokenizer CLASS offline (no BPE vocabulary ships with
this repo or the image)."""
import json
import os


def _bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(2 ** 8):
        if b not in bs:
            bs.append(b)
            cs.append(2 ** 8 + n)
            n += 1
    return [chr(c) for c in cs]


def write_tiny_clip_tokenizer(path) -> int:
    """Writes vocab.json / merges.txt (no merges: every word is spelled in characters); returns the vocabulary size (514)."""
    os.makedirs(path, exist_ok=True)
    chars = _bytes_to_unicode()
    vocab = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    with open(os.path.join(path, "vocab.json"), "w") as f:
        json.dump(vocab, f)
    with open(os.path.join(path, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n")
    return len(vocab)


def write_tiny_checkpoint(path, cfg, unet_sd, vae_sd, text_sd=None, scheduler="PNDMScheduler", text_hidden=64):
    """A diffusers-layout checkpoint directory (what `save_pretrained` writes, reference finetune_sd_token.py:164-187) for a
    small config: unet/ vae/ scheduler/ [+ text_encoder/ + tokenizer/].  Returns the tokenizer vocabulary size (or None)."""
    import torch
    from safetensors.torch import save_file
    for sub in ("unet", "vae", "scheduler"):
        os.makedirs(os.path.join(path, sub), exist_ok=True)
    uc = {"in_channels": 4, "out_channels": 4, "block_out_channels": list(cfg.unet.block_out_channels),
          "down_block_types": ["CrossAttnDownBlock2D" if c else "DownBlock2D" for c in cfg.unet.down_cross],
          "layers_per_block": cfg.unet.layers_per_block, "attention_head_dim": list(cfg.unet.num_heads),
          "cross_attention_dim": cfg.unet.cross_attention_dim, "use_linear_projection": False, "norm_num_groups": 32,
          "sample_size": cfg.default_sample_size}
    vc = {"latent_channels": 4, "out_channels": 3, "block_out_channels": list(cfg.vae.block_out_channels),
          "layers_per_block": cfg.vae.layers_per_block, "norm_num_groups": 32, "scaling_factor": cfg.vae.scaling_factor}
    json.dump(uc, open(os.path.join(path, "unet", "config.json"), "w"))
    json.dump(vc, open(os.path.join(path, "vae", "config.json"), "w"))
    json.dump({"_class_name": scheduler, "num_train_timesteps": 1000, "beta_start": 0.00085, "beta_end": 0.012, "steps_offset": 1,
               "set_alpha_to_one": False, "prediction_type": "epsilon", "skip_prk_steps": True},
              open(os.path.join(path, "scheduler", "scheduler_config.json"), "w"))
    save_file({k: t.contiguous() for k, t in unet_sd.items()}, os.path.join(path, "unet", "diffusion_pytorch_model.safetensors"))
    save_file({k: t.contiguous() for k, t in vae_sd.items()}, os.path.join(path, "vae", "diffusion_pytorch_model.safetensors"))
    if text_sd is None:
        return None
    n_vocab = write_tiny_clip_tokenizer(os.path.join(path, "tokenizer"))
    os.makedirs(os.path.join(path, "text_encoder"), exist_ok=True)
    json.dump({"hidden_size": text_hidden, "num_hidden_layers": 1, "num_attention_heads": 1, "intermediate_size": 2 * text_hidden,
               "vocab_size": n_vocab, "max_position_embeddings": 77, "hidden_act": "quick_gelu", "layer_norm_eps": 1e-5},
              open(os.path.join(path, "text_encoder", "config.json"), "w"))
    tsd = {"text_model." + k: t.contiguous() for k, t in text_sd.items()}
    tsd["text_model.embeddings.position_ids"] = torch.arange(77)[None].float()
    save_file(tsd, os.path.join(path, "text_encoder", "model.safetensors"))
    return n_vocab
