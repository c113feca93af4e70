"""BASELINE config 4's code path end to end on ONE rank, in the reference's on-disk formats (no real fine-tuned checkpoint
exists offline, so the checkpoint is a small synthetic one written in the diffusers layout of finetune_sd_token.py:164-187
plus a `learned_embeds.bin` in the format of finetune_sd_token.py:156-161):
  from_pretrained (the checkpoint's own PNDM scheduler) -> learned-token selection + injection (data_generation.py:33-54)
  -> seed loop with DAAM recording (:56-64) -> per-token heat maps (:70-86) -> postprocess_heatmap.py stacking (:36-50)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_config4_learned_token_heatmaps_to_stacked_rgb(tmp_path):
    from PIL import Image
    from _util import write_tiny_checkpoint
    from agenda_amd import config, generation, postprocess, synthetic
    cfg = config.tiny()
    cfg.text = config.TextConfig(hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=128, vocab_size=514)
    u, v = synthetic.make_unet_weights(cfg, 11, bias_std=0.05), synthetic.make_vae_weights(cfg, 12, bias_std=0.05, with_encoder=True)
    ck = tmp_path / "ckpt"
    n_vocab = write_tiny_checkpoint(str(ck), cfg, u, v, synthetic.make_text_weights(cfg, 3))
    assert n_vocab == 514
    g = torch.Generator().manual_seed(0)
    embeds = {f"new_token_v{i}": torch.randn(64, generator=g) * 0.02 for i in range(3)}          # finetune_sd_token.py:156-161
    torch.save(embeds, tmp_path / "learned_embeds.bin")
    out = tmp_path / "Synthetic"
    # README: template with init tokens (cars, Utah, New Zealand) selects new_token_v0 and new_token_v2 (data_generation.py:39-43)
    generation.main(["--save-dir", str(out), "--pretrained-model-path", str(ck), "--learnable-tokens-embedding-path", str(tmp_path / "learned_embeds.bin"),
                     "--prompt", "An aerial view image with {} cars in {} New Zealand", "--initialize_token", "cars", "Utah", "New Zealand",
                     "--store_learnable_token_heatmaps", "--word_token_heatmaps", "view", "--num-images", "3", "--batch-size", "2",
                     "--num-inference-steps", "3", "--image-size", "56"])
    want = ["0.png", "1.png", "2.png"]
    dirs = sorted(os.listdir(out))
    assert dirs == ["daam_new_token_v0_heatmaps", "daam_new_token_v2_heatmaps", "daam_view_heatmaps", "images"], dirs
    for d in dirs:
        assert sorted(os.listdir(out / d)) == want
    hm = np.asarray(Image.open(out / "daam_new_token_v0_heatmaps" / "1.png"))
    assert hm.shape == (56, 56) and hm.dtype == np.uint8 and hm.max() > 200 and hm.min() < 50       # min-max normalised map
    # the stacked detector input (postprocess_heatmap.py CLI): object = learned car token, fg = "view", bg = learned domain token
    n = postprocess.main(["--save-dir", str(out), "--object-heatmap-path", "daam_new_token_v0_heatmaps", "--fg-heatmap-path",
                          "daam_view_heatmaps", "--bg-heatmap-path", "daam_new_token_v2_heatmaps"])
    assert n == 3
    for f in want:
        o, fg, bg = (np.asarray(Image.open(out / d / f)) for d in ("daam_new_token_v0_heatmaps", "daam_view_heatmaps", "daam_new_token_v2_heatmaps"))
        st = np.asarray(Image.open(out / "daam_stack_heatmaps" / f))
        np.testing.assert_array_equal(st, np.stack([o, fg, 255 - bg], -1))                           # postprocess_heatmap.py:44-48
        np.testing.assert_array_equal(np.asarray(Image.open(out / "daam_inv_heatmaps" / f)), 255 - bg)
    # the learned rows really reached the device text encoder: a different embedding file changes the token's heat map
    embeds2 = dict(embeds); embeds2["new_token_v0"] = embeds["new_token_v0"] + 0.5
    torch.save(embeds2, tmp_path / "learned_embeds2.bin")
    out2 = tmp_path / "Synthetic2"
    generation.main(["--save-dir", str(out2), "--pretrained-model-path", str(ck), "--learnable-tokens-embedding-path", str(tmp_path / "learned_embeds2.bin"),
                     "--prompt", "An aerial view image with {} cars in {} New Zealand", "--initialize_token", "cars", "Utah", "New Zealand",
                     "--store_learnable_token_heatmaps", "--num-images", "1", "--num-inference-steps", "3", "--image-size", "56"])
    a = np.asarray(Image.open(out / "daam_new_token_v0_heatmaps" / "0.png")).astype(int)
    b = np.asarray(Image.open(out2 / "daam_new_token_v0_heatmaps" / "0.png")).astype(int)
    assert np.abs(a - b).max() > 0
