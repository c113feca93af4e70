"""CPU-only tests: host logic that mirrors the reference driver, the oracle's closed-form
properties, and that the C-ABI library loads and exports every symbol include/agenda_hip.h
declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- C ABI ---------------------------------------------------------------------------------
def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "agenda_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"#ifdef AGD_EXPERIMENTS.*?#endif", "", txt, flags=re.S)      # micro-benchmark entry points: experiments library only
    return sorted(set(re.findall(r"\b(agd_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_expected_entry_points():
    syms = _declared_symbols()
    for must in ("agd_create", "agd_destroy", "agd_last_error", "agd_load_tensor", "agd_finalize", "agd_set_context",
                 "agd_unet_forward", "agd_denoise", "agd_vae_decode", "agd_cross_attn", "agd_daam_global", "agd_hook_global"):
        assert must in syms


def test_library_loads_and_exports_every_declared_symbol():
    so = os.path.join(ROOT, "agenda_amd", "libagenda_hip.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(so)
    for s in _declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/agenda_hip.h but not exported"
    lib.agd_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.agd_version()


def test_library_exports_nothing_the_header_does_not_declare():
    """The C ABI is exactly include/agenda_hip.h: the library is built with -fvisibility=hidden, so no helper, experiment
    knob or internal launcher leaks out as an `agd_*` symbol."""
    import subprocess
    so = os.path.join(ROOT, "agenda_amd", "libagenda_hip.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith("agd_")}
    assert exported == set(_declared_symbols()), exported ^ set(_declared_symbols())


def test_python_binding_covers_the_header():
    from agenda_amd import _lib
    assert set(_declared_symbols()) == set(_lib.EXPORTS)
    _lib.load()


def test_config_struct_matches_header_size():
    """agd_create rejects a mismatched struct_size; keep the ctypes mirror in sync with the header."""
    from agenda_amd import _lib
    n_int = 4 + 3 * 8 + 4 + 3 + 8 + 2      # ints before the float
    expect = (n_int + 1 + 2) * 4           # + float + max_tokens + prediction_type
    expect = (expect + 7) // 8 * 8 + 8     # align + long long
    expect += 8 * 4                        # text encoder: 7 ints + float
    assert ctypes.sizeof(_lib.AgdConfig) == expect


def test_product_path_refuses_cpu():
    from agenda_amd import StableDiffusionPipeline, config, _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.AgendaHipError):
        StableDiffusionPipeline.from_synthetic("tiny")


def test_product_never_imports_oracle():
    for dp, _, fns in os.walk(os.path.join(ROOT, "agenda_amd")):
        for fn in fns:
            if fn.endswith(".py"):
                src = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src, fn


# ---- scheduler -----------------------------------------------------------------------------
def test_ddim_scheduler_matches_oracle_restatement():
    from agenda_amd.scheduler import DDIMScheduler
    from oracle import sd_oracle as O
    s, o = DDIMScheduler(), O.DDIM()
    for n in (10, 20, 50):
        ts = s.set_timesteps(n)
        assert list(ts) == list(o.set_timesteps(n))
        a_t, a_p = s.step_coeffs()
        for i, t in enumerate(ts):
            at, ap = o.coeffs(int(t))
            assert abs(a_t[i] - at) < 1e-7 and abs(a_p[i] - ap) < 1e-7
    ts = s.set_timesteps(50)
    assert ts[0] == 981 and ts[-1] == 1 and len(ts) == 50           # leading spacing + steps_offset=1
    assert abs(float(s.alphas_cumprod[0]) - 0.99915) < 1e-5


def test_ddim_step_with_zero_eps_is_pure_rescale():
    from oracle import sd_oracle as O
    o = O.DDIM(); o.set_timesteps(10)
    x = torch.randn(2, 4, 8, 8)
    t = int(o.timesteps[3])
    a_t, a_p = o.coeffs(t)
    y = o.step(torch.zeros_like(x), t, x)
    torch.testing.assert_close(y, x * (a_p / a_t) ** 0.5)


# ---- tokenizer / token merge / driver semantics (reference data_generation.py) ----------------
def test_pndm_scheduler_program_matches_oracle_restatement():
    """The host PNDM/PLMS program (timesteps + the two `_get_prev_sample` coefficients per model evaluation) against the
    oracle's step-by-step restatement of diffusers' `step_plms`, driven with a known eps sequence: same samples."""
    from agenda_amd.scheduler import PNDMScheduler
    from oracle import sd_oracle as O
    n = 20
    s = PNDMScheduler()
    ts = s.set_timesteps(n)
    o = O.PNDM()
    assert list(ts) == list(o.set_timesteps(n)) and len(ts) == n + 1 and ts[1] == ts[2] == 901 and ts[0] == 951 and ts[-1] == 1
    tsf, a, b = s.plms_program()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 8, 8, generator=g)
    eps = [torch.randn(2, 4, 8, 8, generator=g) for _ in ts]
    want = x.clone()
    for t, e in zip(ts, eps):
        want = o.step(e, int(t), want)
    # the device applies: i=0 e0 ; i=1 (e1+e0)/2 from the kept sample ; then the Adams-Bashforth weights on the history
    got, kept, hist = x.clone(), None, []
    for i, e in enumerate(eps):
        if i == 1:
            got = float(a[i]) * kept + float(b[i]) * (0.5 * e + 0.5 * hist[-1])
            continue
        if i == 0:
            kept = got.clone()
        h = hist[-3:]
        w = {0: [1.0], 1: [1.5, -0.5], 2: [23 / 12, -16 / 12, 5 / 12], 3: [55 / 24, -59 / 24, 37 / 24, -9 / 24]}[len(h)]
        comb = w[0] * e + sum(wk * hk for wk, hk in zip(w[1:], reversed(h)))
        got = float(a[i]) * got + float(b[i]) * comb
        hist.append(e)
    assert float((got - want).abs().max()) < 1e-4 * float(want.abs().max())
    with pytest.raises(ValueError):
        PNDMScheduler(prediction_type="v_prediction")


def test_compute_token_merge_indices():
    from agenda_amd.text import SimpleTokenizer
    from agenda_amd.trace import compute_token_merge_indices
    from oracle import sd_oracle as O
    tok = SimpleTokenizer()
    prompt = "An aerial view image with cars in Utah"
    idx, _ = compute_token_merge_indices(tok, prompt, "cars")
    assert idx == [6]                                                 # 0-based position 5, +1 for SOS
    assert O.compute_token_merge_indices(tok.tokenize, prompt, "cars")[0] == idx
    idx, _ = compute_token_merge_indices(tok, "a red car and a blue car", "car")
    assert idx == [3, 7]                                              # every occurrence is merged
    with pytest.raises(ValueError, match="not found in prompt"):
        compute_token_merge_indices(tok, prompt, "boat")


def test_learned_token_selection_follows_reference_rule():
    """data_generation.py:36-43 and README example: template with init tokens cars/Utah/New Zealand."""
    from agenda_amd.generation import select_learned_tokens
    from oracle import sd_oracle as O
    tpl = "An aerial view image with {} cars in {} New Zealand"
    learned = ["new_token_v0", "new_token_v1", "new_token_v2"]
    init = ["cars", "Utah", "New Zealand"]
    words = ["cars"]
    new, w, prompt = select_learned_tokens(tpl, init, learned, words, store_learnable=True)
    assert new == ["new_token_v0", "new_token_v2"]
    assert w is words and words == ["cars", "new_token_v0", "new_token_v2"]     # aliasing + in-place append
    assert prompt == "An aerial view image with new_token_v0 cars in new_token_v2 New Zealand"
    assert O.select_learned_tokens(tpl, init, learned, ["cars"], True)[0] == new


def test_learned_token_injection_and_tokenization():
    from agenda_amd.text import SimpleTokenizer, SyntheticTextEncoder
    from agenda_amd.generation import inject_learned_tokens
    from agenda_amd.trace import compute_token_merge_indices

    class P:
        pass
    p = P(); p.tokenizer = SimpleTokenizer(); p.text_encoder = SyntheticTextEncoder(p.tokenizer, 64)
    emb = {"new_token_v0": torch.full((64,), 0.5), "new_token_v2": torch.full((64,), -0.25)}
    ids = inject_learned_tokens(p, emb, ["new_token_v0", "new_token_v2"])
    w = p.text_encoder.get_input_embeddings().weight
    assert torch.all(w.data[ids[0]] == 0.5) and torch.all(w.data[ids[1]] == -0.25)
    prompt = "An aerial view image with new_token_v0 cars in new_token_v2 New Zealand"
    assert compute_token_merge_indices(p.tokenizer, prompt, "new_token_v0")[0] == [6]
    ctx = p.text_encoder([prompt])
    assert ctx.shape == (1, 77, 64)


def test_heatmap_export_truncates_not_rounds():
    from agenda_amd.generation import export_heatmap_u8, stack_heatmaps
    from oracle import sd_oracle as O
    hm = np.array([[0.0, 0.999], [0.5, 1.0]], dtype=np.float32)
    u8 = export_heatmap_u8(hm)
    assert u8.dtype == np.uint8 and u8[0, 0] == 0 and u8[1, 1] == 255      # fp32: 1 + 1e-8 == 1
    assert u8[0, 1] == 254 and u8[1, 0] == 127                              # 254.745 / 127.5 TRUNCATE (round would give 255 / 128)
    np.testing.assert_array_equal(u8, O.export_heatmap_u8(hm))
    obj, fg, bg = (np.full((2, 2), v, np.uint8) for v in (10, 20, 30))
    rgb, inv = stack_heatmaps(obj, fg, bg)
    assert rgb.shape == (2, 2, 3) and tuple(rgb[0, 0]) == (10, 20, 225) and inv[0, 0] == 225
    r2, i2 = O.stack_heatmaps(obj, fg, bg)
    np.testing.assert_array_equal(rgb, r2)


def test_seed_sharding_covers_every_seed_once():
    from agenda_amd.generation import shard_seeds
    for world in (1, 2, 3, 8):
        allseeds = sorted(s for r in range(world) for s in shard_seeds(37, r, world))
        assert allseeds == list(range(37))


# ---- oracle closed-form properties (parity-unpinned parts) -------------------------
def test_oracle_bicubic_scale1_is_identity_and_softmax_rows_sum_to_one():
    from oracle import sd_oracle as O
    m = torch.rand(1, 3, 16, 16)
    torch.testing.assert_close(O.hooker_global_heat_map([m], 16), m)
    q, k = torch.randn(4, 10, 8), torch.randn(4, 7, 8)
    p = O.attention_scores(q, k, 8 ** -0.5)
    torch.testing.assert_close(p.sum(-1), torch.ones(4, 10))


def test_oracle_daam_recorder_rules():
    from oracle import sd_oracle as O
    rec = O.DaamRecorder(latent_area=64 * 64, context_size=77)
    heads, B = 2, 1
    p = torch.rand(2 * B * heads, 1024, 77).softmax(-1)
    rec(p, heads, "down_blocks.1.attentions.0.transformer_blocks.0.attn2")
    assert len(rec.acc) == heads                                   # one accumulator per head
    rec(p, heads, "down_blocks.1.attentions.0.transformer_blocks.0.attn2")
    key = (2, "down_blocks.1.attentions.0.transformer_blocks.0.attn2", 0)
    want = 2 * p[2:].reshape(B, heads, 1024, 77)[:, 0].permute(0, 2, 1).reshape(B, 77, 32, 32)
    torch.testing.assert_close(rec.acc[key], want)                 # conditional half, summed over time
    rec(torch.rand(4, 64, 77), heads, "mid_block.attentions.0.transformer_blocks.0.attn2")
    rec(torch.rand(4, 64, 77), heads, "up_blocks.0.x")             # factor 8 -> skipped
    rec(torch.rand(4, 1024, 1024), heads, "down_blocks.1.self")    # not 77 keys -> skipped
    assert len(rec.acc) == heads
    g = rec.compute_global_heat_map(n_rows=14)
    assert g.shape == (1, 14, 64, 64) and float(g.min()) >= 0


def test_oracle_tiny_pipeline_runs_and_is_deterministic():
    from agenda_amd import config, synthetic
    from oracle import sd_oracle as O
    cfg = config.tiny()
    u, v = synthetic.make_unet_weights(cfg), synthetic.make_vae_weights(cfg)
    ctx, lat = synthetic.make_context(cfg, 1), synthetic.make_latents(cfg, [3], 16)
    a = O.generate(u, v, cfg, ctx, lat, 2)
    b = O.generate(u, v, cfg, ctx, lat, 2)
    assert a[0].shape == (1, 128, 128, 3) and a[0].dtype == np.uint8
    np.testing.assert_array_equal(a[0], b[0])


def test_param_inventory_counts():
    from agenda_amd import config
    n = sum(int(np.prod(s)) for s in config.unet_param_shapes(config.sd15().unet).values())
    assert abs(n - 859.52e6) < 0.05e6                               # SD-1.x UNet: 859.5 M parameters
    names = config.cross_attn_layer_names(config.sd15().unet)
    assert len(names) == 16 and names[-1].startswith("mid_block")   # 9 up + 6 down + 1 mid attn2


def test_save_outputs_and_postprocess_roundtrip(tmp_path):
    """Host export path + postprocess CLI produce the reference's directory layout and stacked RGB."""
    from PIL import Image
    from agenda_amd.generation import save_outputs, export_heatmap_u8
    from agenda_amd import postprocess
    rng = np.random.default_rng(1)
    imgs = rng.integers(1, 256, size=(2, 64, 64, 3), dtype=np.uint8)
    imgs[1] = 0                                                      # all-black image is skipped (data_generation.py:61-62)
    hms = rng.random((2, 3, 16, 16), dtype=np.float32)
    words = ["cars", "new_token_v0", "new_token_v2"]
    save_outputs(str(tmp_path), [7, 8], imgs, hms, words, 28)
    assert sorted(os.listdir(tmp_path / "images")) == ["7.png"]
    for wi, w in enumerate(words):
        got = np.asarray(Image.open(tmp_path / f"daam_{w}_heatmaps" / "7.png"))
        want = np.asarray(Image.fromarray(export_heatmap_u8(hms[0, wi])).resize((28, 28)))
        np.testing.assert_array_equal(got, want)
        assert not (tmp_path / f"daam_{w}_heatmaps" / "8.png").exists()
    n = postprocess.main(["--save-dir", str(tmp_path), "--object-heatmap-path", "daam_cars_heatmaps",
                          "--fg-heatmap-path", "daam_new_token_v0_heatmaps", "--bg-heatmap-path", "daam_new_token_v2_heatmaps"])
    assert n == 1
    rgb = np.asarray(Image.open(tmp_path / "daam_stack_heatmaps" / "7.png"))
    o, f, b = (np.asarray(Image.open(tmp_path / f"daam_{w}_heatmaps" / "7.png")) for w in words)
    np.testing.assert_array_equal(rgb, np.stack([o, f, 255 - b], -1))
    np.testing.assert_array_equal(np.asarray(Image.open(tmp_path / "daam_inv_heatmaps" / "7.png")), 255 - b)


def test_pndm_without_skip_prk_steps_is_refused_not_silently_plms(tmp_path):
    """diffusers' PNDMScheduler defaults skip_prk_steps to False (Runge-Kutta warm-up steps); only the SD configuration (True = PLMS)
    is implemented, and a scheduler config that says otherwise must not run PLMS silently."""
    from agenda_amd.config import SchedulerConfig
    from agenda_amd.scheduler import PNDMScheduler
    PNDMScheduler.from_config(SchedulerConfig())                       # SD's own: skip_prk_steps = True
    with pytest.raises(ValueError, match="skip_prk_steps"):
        PNDMScheduler.from_config(SchedulerConfig(skip_prk_steps=False))
