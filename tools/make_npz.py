#!/usr/bin/env python3
"""Dumps a directory of 32x32 PNG / JPEG images (e.g. the van den Oord et al. downsampled-ImageNet archives
train_32x32.tar / valid_32x32.tar that TFDS `downsampled_imagenet/32x32` serves to the reference, ldm/dataset.py:187-199)
into the .npz that `--config.data.dataset=npz:<file>` reads: uint8 `images` [N, 32, 32, 3] in sorted file-name order
(the order TFDS yields the validation split in is the archive order = sorted names).

    mkdir valid_32x32 && tar -xf valid_32x32.tar -C valid_32x32
    python tools/make_npz.py valid_32x32 imagenet32_oord_valid.npz [--limit N]
"""
import argparse
import os

import numpy as np


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("src", help="directory with the images (searched recursively)")
    ap.add_argument("out", help="output .npz")
    ap.add_argument("--limit", type=int, default=0)
    a = ap.parse_args()
    from PIL import Image
    names = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(a.src) for f in fs
                   if f.lower().endswith((".png", ".jpg", ".jpeg")))
    if a.limit:
        names = names[:a.limit]
    if not names:
        raise SystemExit(f"no images under {a.src}")
    out = np.empty((len(names), 32, 32, 3), dtype=np.uint8)
    for i, n in enumerate(names):
        im = np.asarray(Image.open(n).convert("RGB"))
        if im.shape != (32, 32, 3):
            raise SystemExit(f"{n}: shape {im.shape}, expected 32x32x3")
        out[i] = im
    np.savez(a.out, images=out)
    print(f"{len(names)} images -> {a.out}")


if __name__ == "__main__":
    main()
