"""CLIP text encoder on device vs `transformers.CLIPTextModel` -- the implementation the
reference runs through `pipeline.text_encoder` (requirements.txt:17), random-initialised (no checkpoints offline)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _hf_model(tc, sd):
    from transformers import CLIPTextConfig, CLIPTextModel
    cfg = CLIPTextConfig(hidden_size=tc.hidden_size, intermediate_size=tc.intermediate_size,
                         num_hidden_layers=tc.num_hidden_layers, num_attention_heads=tc.num_attention_heads,
                         max_position_embeddings=tc.max_position_embeddings, vocab_size=tc.vocab_size,
                         hidden_act=tc.hidden_act, layer_norm_eps=tc.layer_norm_eps)
    m = CLIPTextModel(cfg).eval()
    own = m.state_dict()
    pref = "text_model." if any(k.startswith("text_model.") for k in own) else ""
    m.load_state_dict({pref + k: v for k, v in sd.items()}, strict=False)
    return m


@pytest.mark.parametrize("layers,hidden,heads,inter,act", [(2, 128, 2, 256, "quick_gelu"), (12, 768, 12, 3072, "quick_gelu"),
                                                            (3, 256, 4, 1024, "gelu")])
def test_text_encoder_matches_transformers_clip(layers, hidden, heads, inter, act):
    from agenda_amd import StableDiffusionPipeline, config, synthetic
    cfg = config.tiny(cross_dim=hidden if hidden % 64 == 0 else 64)
    cfg.text = config.TextConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                                 intermediate_size=inter, vocab_size=1000, hidden_act=act)
    tsd = synthetic.make_text_weights(cfg, seed=5)
    pipe = StableDiffusionPipeline(cfg, synthetic.make_unet_weights(cfg), synthetic.make_vae_weights(cfg),
                                   text_sd=tsd, workspace_bytes=1 << 30)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 1000, (3, 77), generator=g)
    ids[:, 0] = 998
    ids[:, 40:] = 999                                         # padded tail, like CLIP's EOS padding
    with torch.no_grad():
        want = _hf_model(cfg.text, tsd)(input_ids=ids).last_hidden_state
    got = pipe.engine.text_encode(ids).cpu()
    assert got.shape == want.shape
    err = float((got - want).abs().max() / want.abs().max())
    assert err < 2.0 ** -5, err
    rms = float(((got - want) ** 2).mean().sqrt() / (want ** 2).mean().sqrt())
    assert rms < 2.0 ** -6, rms        # bf16 activations through up to 12 residual layers: measured 0.008 at ViT-L/14 size
    pipe.engine.close()


def test_learned_token_injection_reaches_the_device():
    """data_generation.py:45-52 flow on the device encoder: add tokens, resize, overwrite rows, encode."""
    from agenda_amd import StableDiffusionPipeline, config, synthetic
    from agenda_amd.generation import inject_learned_tokens
    cfg = config.tiny(cross_dim=128)
    cfg.text = config.TextConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=600)
    tsd = synthetic.make_text_weights(cfg, seed=6)
    pipe = StableDiffusionPipeline(cfg, synthetic.make_unet_weights(cfg), synthetic.make_vae_weights(cfg), text_sd=tsd,
                                   workspace_bytes=1 << 30)
    prompt = "An aerial view image with new_token_v0 cars in new_token_v2 New Zealand"
    for w in prompt.lower().split():
        pipe.tokenizer.convert_tokens_to_ids(w + "</w>")      # stable ids below the base vocab
    emb = {"new_token_v0": torch.randn(128) * 0.05, "new_token_v2": torch.randn(128) * 0.05}
    pipe.tokenizer.vocab = {k: v for k, v in pipe.tokenizer.vocab.items()}
    base = len(pipe.tokenizer)
    # the tokenizer stand-in numbers added tokens after its own words; place them after the encoder's base vocab
    pipe.tokenizer.vocab.update({f"<pad{i}>": i for i in range(base, 600)})
    ids = inject_learned_tokens(pipe, emb, ["new_token_v0", "new_token_v2"])
    assert ids == [600, 601]
    e1 = pipe.text_encoder([prompt])
    sd2 = dict(tsd)
    sd2["embeddings.token_embedding.weight"] = torch.cat([tsd["embeddings.token_embedding.weight"],
                                                          torch.stack([emb["new_token_v0"], emb["new_token_v2"]])])
    cfg2 = config.TextConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=602)
    tid = torch.tensor([pipe.tokenizer.encode(prompt)])
    with torch.no_grad():
        want = _hf_model(cfg2, sd2)(input_ids=tid).last_hidden_state
    assert float((e1 - want).abs().max() / want.abs().max()) < 2.0 ** -5
    # and the pipeline consumes it: prompt -> context -> UNet runs
    out = pipe([prompt], num_inference_steps=1, output_type="latent")
    assert torch.isfinite(out.latents).all()
    pipe.engine.close()
