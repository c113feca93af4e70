"""Stack three grayscale DAAM heat maps into the RGB input the detectors consume -- same CLI and on-disk result as
reference data_generation/postprocess_heatmap.py:8-50 (`[obj, fg, 255 - bg]` + the inverted background map).
The reference pairs files by the *unsorted* `os.listdir` order of the three folders (postprocess_heatmap.py:32-36),
which is only right when all three listings enumerate identically; here files are paired by NAME (the seed), which
gives the same result whenever the reference's pairing is correct and stays correct when it would not be."""
from __future__ import annotations

import argparse
import os

import numpy as np
from PIL import Image

from .generation import stack_heatmaps


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Stack attention map.")
    p.add_argument("--save-dir", type=str, default="Data/Synthetic")
    p.add_argument("--object-heatmap-path", type=str, default=None)
    p.add_argument("--fg-heatmap-path", type=str, default=None)
    p.add_argument("--bg-heatmap-path", type=str, default=None)
    p.add_argument("--stack-heatmap-save-path", type=str, default="daam_stack_heatmaps")
    p.add_argument("--inv-heatmap-save-path", type=str, default="daam_inv_heatmaps")
    return p.parse_args(argv)


def main(argv=None):
    a = parse_args(argv)
    obj_d, fg_d, bg_d = (os.path.join(a.save_dir, x) for x in (a.object_heatmap_path, a.fg_heatmap_path, a.bg_heatmap_path))
    out_d, inv_d = os.path.join(a.save_dir, a.stack_heatmap_save_path), os.path.join(a.save_dir, a.inv_heatmap_save_path)
    os.makedirs(out_d, exist_ok=True)
    os.makedirs(inv_d, exist_ok=True)
    names = sorted(set(os.listdir(obj_d)) & set(os.listdir(fg_d)) & set(os.listdir(bg_d)))
    for n in names:
        o, f, b = (np.asarray(Image.open(os.path.join(d, n))) for d in (obj_d, fg_d, bg_d))
        rgb, inv = stack_heatmaps(o, f, b)
        Image.fromarray(rgb).save(os.path.join(out_d, n))
        Image.fromarray(inv).save(os.path.join(inv_d, n))
    return len(names)


if __name__ == "__main__":
    main()
