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
