"""Driver for the generation hot loop, mirroring reference data_generation/data_generation.py:26-86
(token selection, per-seed image + per-word DAAM heat-map export), batched and seed-sharded
across ranks: images are independent per seed, so rank r takes seeds
s = r (mod world); the only exchange is one all_gather of uint8 images + fp32 word heat maps per
batch (RCCL over xGMI on GPUs; gloo in the CPU tests)."""
from __future__ import annotations

import argparse
import os
from typing import List, Optional, Sequence

import numpy as np
import torch


def select_learned_tokens(prompt_template: str, initialize_token: Sequence[str], learned: Sequence[str],
                          word_token_heatmaps: Optional[List[str]], store_learnable: bool):
    """data_generation.py:36-43,54 -- a learned token is used iff its init word is a substring of the
    UNFORMATTED template; the heat-map word list aliases the CLI list and is appended in place."""
    words = word_token_heatmaps if word_token_heatmaps is not None else []
    new_tokens = []
    for t, n in zip(initialize_token, learned):
        if t in prompt_template:
            if store_learnable:
                words.append(n)
            new_tokens.append(n)
    return new_tokens, words, prompt_template.format(*new_tokens)


def inject_learned_tokens(pipe, embeds_dict, new_tokens):
    """data_generation.py:45-52."""
    if not new_tokens:
        return []
    emb = torch.stack([embeds_dict[t] for t in new_tokens])
    pipe.tokenizer.add_tokens(list(new_tokens))
    ids = pipe.tokenizer.convert_tokens_to_ids(list(new_tokens))
    pipe.text_encoder.resize_token_embeddings(len(pipe.tokenizer))
    with torch.no_grad():
        w = pipe.text_encoder.get_input_embeddings().weight
        w.data[ids] = emb.to(w.dtype)
    return ids


def shard_seeds(num_images: int, rank: int, world: int) -> List[int]:
    return list(range(rank, num_images, world))


def export_heatmap_u8(hm: np.ndarray) -> np.ndarray:
    """data_generation.py:82-84: min-max (+1e-8), x255, truncating uint8 cast."""
    hm = np.asarray(hm, dtype=np.float32)
    hm = (hm - hm.min()) / (hm.max() - hm.min() + 1e-8) * 255
    return hm.astype(np.uint8)


def stack_heatmaps(obj: np.ndarray, fg: np.ndarray, bg: np.ndarray):
    """postprocess_heatmap.py:44-48."""
    inv = 255 - bg
    return np.stack([obj, fg, inv], axis=-1), inv


def generate_batch(pipe, seeds: Sequence[int], words: Sequence[str], prompt: Optional[str] = None,
                   prompt_embeds: Optional[torch.Tensor] = None, num_inference_steps: int = 50,
                   guidance_scale: float = 7.5, height: Optional[int] = None, rec_tokens: Optional[int] = None,
                   word_rows: Optional[Sequence[Sequence[int]]] = None):
    """One hot-path pass: len(seeds) images + per-word DAAM maps.
    Returns (uint8 images [B,H,W,3] on GPU, fp32 heat maps [B, n_words, S, S] on GPU)."""
    from .trace import trace
    from . import synthetic
    B = len(seeds)
    side = height or pipe.cfg.default_sample_size * pipe.vae_scale_factor
    L = side // pipe.vae_scale_factor
    lat = synthetic.make_latents(pipe.cfg, seeds, L)      # CPU generator per image seed (data_generation.py:58)
    with trace(pipe, rec_tokens=rec_tokens) as trc:
        if prompt_embeds is None:
            out = pipe([prompt] * B, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                       latents=lat, height=side, width=side, output_type="pt")
        else:
            out = pipe(prompt_embeds=prompt_embeds, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                       latents=lat, height=side, width=side, output_type="pt")
        hms = []
        for i in range(B):
            g = trc.compute_global_heat_map(prompt=prompt, image_index=i)
            if word_rows is not None:
                # integer (view) indexing: a python list index makes torch upload an index tensor with a BLOCKING pageable copy, i.e. a host
                # sync at the end of every batch
                hms.append(torch.stack([torch.stack([g.heat_maps[int(i)] for i in r]).mean(0) for r in word_rows]))
            elif words:
                hms.append(torch.stack([g.compute_word_heat_map(w).heatmap for w in words]))
            else:
                hms.append(g.heat_maps[:0])
    return out.images, torch.stack(hms)


class PendingGather:
    """Handle of an all_gather posted with `gather_outputs(..., async_op=True)`: the collectives run on the backend's own stream
    while the caller enqueues the next batch; `wait()` returns what the blocking call returns."""

    def __init__(self, works, finish):
        self._works, self._finish, self._out = works, finish, None

    def wait(self):
        if self._out is None:
            for w in self._works:
                w.wait()
            self._out = self._finish()
        return self._out


def gather_outputs(images: torch.Tensor, heatmaps: torch.Tensor, seeds: Optional[Sequence[int]] = None, max_batch: Optional[int] = None,
                   async_op: bool = False, global_seeds: Optional[Sequence[int]] = None):
    """The final exchange step: literally ONE all_gather per batch.  Every rank packs its seed ids, images and
    heat maps into one byte buffer ([ids int64 | images | heat maps], sections 16-byte aligned); one collective moves it; the
    result is unpacked as views of the gathered buffer.

    Ranks may hold different batch sizes (the last, ragged round of `shard_seeds`; even zero images): every rank pads its
    sections to `max_batch` rows (id -1), the padding is dropped after the collective, and rows come back ordered by seed.
    `max_batch` must be the same on every rank (defaults to the local batch: the equal-batch case of bench.py).
    Returns (images, heatmaps) when `seeds` is None (equal full batches: rank-interleaved = the global seed order of
    `shard_seeds`, a pure view permutation), else (seeds, images, heatmaps).  With `global_seeds` (the sorted seeds of ALL ranks in
    this round, which callers of `shard_seeds` know on the host) nothing here synchronises with the host: rows are ordered by a
    masked argsort on the device and sliced by len(global_seeds); without it the seed list is read back from the gathered ids
    (one device -> host copy).  No-op for world size 1.

    Contract of `global_seeds`: it must be EXACTLY the sorted union of the seeds the ranks packed this round (what `shard_seeds`
    hands each rank, derived from one place -- `round_seeds` in `generation.main`).  The sync-free path cannot check that: a list
    that disagrees with the packed ids (a rank that dropped an image, a different `max_batch`) would pair seeds with the wrong rows
    silently.  Set AGD_GATHER_CHECK=1 to verify it (one device -> host copy per round: gathered ids == global_seeds, the next id the
    padding sentinel); the ragged tests run with it on."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        res = (images, heatmaps) if seeds is None else (list(seeds), images, heatmaps)
        return PendingGather([], lambda: res) if async_op else res
    w = dist.get_world_size()
    b = images.shape[0]
    mb = int(max_batch) if max_batch is not None else b
    if b > mb:
        raise ValueError(f"local batch {b} > max_batch {mb}")
    if seeds is None and b != mb:
        raise ValueError("gather_outputs without seeds is the equal-batch form: pass seeds (and max_batch) for ragged rounds")
    dev = images.device
    ishape, hshape = tuple(images.shape[1:]), tuple(heatmaps.shape[1:])
    ib = images[:0].new_empty((1,) + ishape).numel() * images.element_size()        # bytes per image row
    hb = heatmaps[:0].new_empty((1,) + hshape).numel() * heatmaps.element_size()
    a16 = lambda n: (n + 15) & ~15
    o_img = a16(mb * 8)
    o_hm = o_img + a16(mb * ib)
    total = o_hm + a16(mb * hb)
    buf = torch.zeros(total, dtype=torch.uint8, device=dev)
    ids = buf[:mb * 8].view(torch.int64)
    ids.fill_(-1)
    if b:
        if seeds is not None:
            ids[:b] = torch.as_tensor(list(seeds), dtype=torch.int64).to(dev, non_blocking=True)
        else:
            ids[:b] = torch.arange(b, device=dev) * w + dist.get_rank()             # seed = rank + world * local_index
        buf[o_img:o_img + b * ib] = images.contiguous().view(torch.uint8).reshape(-1)
        buf[o_hm:o_hm + b * hb] = heatmaps.contiguous().view(torch.uint8).reshape(-1)
    # the collective form is chosen up front, identically on every rank (never retry a collective after one failed: the other
    # ranks would not join the second one): RCCL takes the flat form, other backends (gloo rehearsals) the list form
    if dist.get_backend() == "nccl":
        out = torch.empty(w * total, dtype=torch.uint8, device=dev)
        works = [dist.all_gather_into_tensor(out, buf, async_op=True)]
        gathered = lambda: out.view(w, total)
    else:
        parts = [torch.empty_like(buf) for _ in range(w)]
        works = [dist.all_gather(parts, buf, async_op=True)]
        gathered = lambda: torch.stack(parts)

    def finish():
        g = gathered()                                                              # [world][total] bytes
        gs = g[:, :mb * 8].contiguous().view(torch.int64).reshape(w * mb)
        gi = g[:, o_img:o_img + mb * ib].contiguous().view(images.dtype).reshape((w, mb) + ishape)
        gh = g[:, o_hm:o_hm + mb * hb].contiguous().view(heatmaps.dtype).reshape((w, mb) + hshape)
        if seeds is None:                       # equal full batches: seed = rank + world * local index -> transpose, no sort, no sync
            return gi.transpose(0, 1).reshape((w * mb,) + ishape), gh.transpose(0, 1).reshape((w * mb,) + hshape)
        gi, gh = gi.reshape((w * mb,) + ishape), gh.reshape((w * mb,) + hshape)
        key = torch.where(gs >= 0, gs, torch.full_like(gs, torch.iinfo(torch.int64).max))
        order = torch.argsort(key)
        if global_seeds is not None:            # the caller knows every rank's seeds of this round: no host sync at all
            n = len(global_seeds)
            if os.environ.get("AGD_GATHER_CHECK"):      # debug: the caller's list against what the ranks really packed (host sync)
                got = key[order].tolist()
                sentinel = torch.iinfo(torch.int64).max
                if got[:n] != [int(s_) for s_ in global_seeds] or (n < len(got) and got[n] != sentinel):
                    raise RuntimeError(f"gather_outputs: global_seeds {list(global_seeds)} disagree with the gathered ids "
                                       f"{[g_ for g_ in got if g_ != sentinel]}")
            return list(global_seeds), gi[order[:n]], gh[order[:n]]
        srt = key[order].tolist()               # (device -> host: the seed list is part of the result)
        n = sum(1 for s_ in srt if s_ != torch.iinfo(torch.int64).max)
        return srt[:n], gi[order[:n]], gh[order[:n]]

    pend = PendingGather(works, finish)
    return pend if async_op else pend.wait()


def save_outputs(save_dir: str, seeds, images_u8, heatmaps, words, image_size: int, stack_words=None, exported: bool = False):
    """data_generation.py:60-62,66-86: resize, skip all-black, images/ + daam_<word>_heatmaps/ PNGs.
    CUDA tensors take the device export path (agenda_amd/export.py: min-max -> uint8 -> PIL-exact bicubic resize on
    the GPU, one D2H copy of the finished buffers); numpy inputs take the reference's literal host code.  Both
    produce identical bytes (tests/test_export.py).  `stack_words=(obj, fg, bg)` additionally writes
    daam_stack_heatmaps/ + daam_inv_heatmaps/ as postprocess_heatmap.py:44-50 would."""
    from PIL import Image
    os.makedirs(os.path.join(save_dir, "images"), exist_ok=True)
    if exported:                                # already the final payloads (export_batch ran before the gather)
        small, hm = (t.cpu().numpy() if torch.is_tensor(t) else np.asarray(t) for t in (images_u8, heatmaps))
    elif torch.is_tensor(images_u8) and images_u8.is_cuda:
        from . import export
        small, hm = export.export_batch(images_u8, heatmaps, image_size)
        small, hm = small.cpu().numpy(), hm.cpu().numpy()
    else:
        images_u8, heatmaps = np.asarray(images_u8), np.asarray(heatmaps)
        small = np.stack([np.asarray(Image.fromarray(im).resize((image_size, image_size))) for im in images_u8])
        hm = np.stack([np.stack([np.asarray(Image.fromarray(export_heatmap_u8(h)).resize((image_size, image_size)))
                                 for h in hs]) if len(hs) else np.zeros((0, image_size, image_size), np.uint8) for hs in heatmaps])
    for i, seed in enumerate(seeds):
        if np.max(small[i]) < 1e-5:             # NSFW content filter (black image), data_generation.py:61-62
            continue
        Image.fromarray(small[i]).save(os.path.join(save_dir, "images", f"{seed}.png"))
        for wi, word in enumerate(words):
            d = os.path.join(save_dir, "daam_" + word + "_heatmaps")
            os.makedirs(d, exist_ok=True)
            Image.fromarray(hm[i, wi]).save(os.path.join(d, f"{seed}.png"))
        if stack_words is not None:
            o, f, b = (hm[i, list(words).index(w)] for w in stack_words)
            rgb, inv = stack_heatmaps(o, f, b)
            for sub, arr in (("daam_stack_heatmaps", rgb), ("daam_inv_heatmaps", inv)):
                os.makedirs(os.path.join(save_dir, sub), exist_ok=True)
                Image.fromarray(arr).save(os.path.join(save_dir, sub, f"{seed}.png"))


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Image and attention map generation (MI355X).")
    p.add_argument("--save-dir", type=str, default="Data/Synthetic")
    p.add_argument("--pretrained-model-path", type=str, default=None, help="diffusers-layout checkpoint; omit for synthetic weights")
    p.add_argument("--learnable-tokens-embedding-path", type=str, default=None)
    p.add_argument("--prompt", type=str, default="An aerial view image with {} cars in {} Utah")
    p.add_argument("--initialize_token", type=str, default=["cars", "Utah", "New Zealand"], nargs="+")
    p.add_argument("--word_token_heatmaps", type=str, default=None, nargs="+")
    p.add_argument("--store_learnable_token_heatmaps", action="store_true")
    p.add_argument("--num-images", type=int, default=10000)
    p.add_argument("--image-size", type=int, default=112)
    p.add_argument("--batch-size", type=int, default=4)
    p.add_argument("--num-inference-steps", type=int, default=20)
    p.add_argument("--synthetic-config", type=str, default="sd15", help="architecture for synthetic weights when no checkpoint is given")
    p.add_argument("--stack", type=str, default=None, nargs=3, metavar=("OBJ", "FG", "BG"),
                   help="also write daam_stack_heatmaps/ + daam_inv_heatmaps/ for these three words (postprocess_heatmap.py)")
    p.add_argument("--scheduler", type=str, default=None, choices=["DDIMScheduler", "PNDMScheduler"],
                   help="default: the checkpoint's scheduler/scheduler_config.json (PNDM for SD-1.4, as the reference runs it); "
                        "DDIMScheduler for synthetic weights")
    p.add_argument("--no-gather", action="store_true",
                   help="multi-GPU: every rank writes its own files instead of the final all_gather to rank 0")
    return p.parse_args(argv)


def main(argv=None):
    import torch.distributed as dist
    from . import StableDiffusionPipeline
    args = parse_args(argv)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if "AGD_FORCE_DEVICE" in os.environ:          # rehearsal of the multi-rank path on a 1-GPU box (ranks share one card)
        local = int(os.environ["AGD_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    own_group = False
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("AGD_DIST_BACKEND", "nccl"))          # "nccl" = RCCL on ROCm
        own_group = True
    pipe = (StableDiffusionPipeline.from_pretrained(args.pretrained_model_path, device=local, scheduler=args.scheduler)
            if args.pretrained_model_path else
            StableDiffusionPipeline.from_synthetic(args.synthetic_config, device=local, scheduler=args.scheduler or "DDIMScheduler"))
    embeds = torch.load(args.learnable_tokens_embedding_path) if args.learnable_tokens_embedding_path else {}
    if embeds:
        new_tokens, words, prompt = select_learned_tokens(args.prompt, args.initialize_token, list(embeds.keys()),
                                                         args.word_token_heatmaps, args.store_learnable_token_heatmaps)
        inject_learned_tokens(pipe, embeds, new_tokens)
    else:       # no learned-token file (the reference requires one): plain prompt, placeholders dropped
        words = args.word_token_heatmaps if args.word_token_heatmaps is not None else []
        prompt = " ".join(args.prompt.replace("{}", " ").split())
    seeds = shard_seeds(args.num_images, rank, world)
    gather = world > 1 and not args.no_gather
    # every rank joins every round (the collective count must match): ranks whose shard ran out contribute zero rows
    per_rank = (args.num_images + world - 1) // world
    rounds = (per_rank + args.batch_size - 1) // args.batch_size if gather else (len(seeds) + args.batch_size - 1) // args.batch_size
    S = args.image_size
    for r in range(rounds):
        chunk = seeds[r * args.batch_size:(r + 1) * args.batch_size]
        if chunk:
            imgs, hms = generate_batch(pipe, chunk, words, prompt=prompt, num_inference_steps=args.num_inference_steps)
        if not gather:
            save_outputs(args.save_dir, chunk, imgs, hms, words, S, stack_words=args.stack)
            continue
        # export on the producing GPU (resize 512 -> S, min-max -> uint8 -> resize), then ONE gather of the finished
        # payloads (S*S*3 + S*S per word bytes per image instead of the full-size tensors); rank 0 writes the files
        from . import export
        dev = torch.device("cuda", local)
        if chunk:
            small, hm8 = export.export_batch(imgs, hms, S)
        else:
            small = torch.zeros(0, S, S, 3, dtype=torch.uint8, device=dev)
            hm8 = torch.zeros(0, len(words), S, S, dtype=torch.uint8, device=dev)
        if dist.get_backend() == "gloo":
            small, hm8 = small.cpu(), hm8.cpu()
        round_seeds = sorted(s_ for rk in range(world) for s_ in shard_seeds(args.num_images, rk, world)[r * args.batch_size:(r + 1) * args.batch_size])
        all_seeds, small, hm8 = gather_outputs(small, hm8, seeds=chunk, max_batch=args.batch_size, global_seeds=round_seeds)
        if rank == 0:
            save_outputs(args.save_dir, all_seeds, small, hm8, words, S, stack_words=args.stack, exported=True)
    if world > 1:
        dist.barrier()
        if own_group:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
