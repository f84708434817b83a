"""Writes seed files for tools/fuzz_host.cpp: PNGs of every colour type / depth / interlacing, baseline and
progressive JPEGs with and without subsampling (PIL encodes them), and binary glTF files from the test
suite's own writer.  usage: make_fuzz_seeds.py OUTDIR"""
import os
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
rng = np.random.default_rng(11)
rgb = rng.integers(0, 256, (24, 20, 3), dtype=np.uint8)
rgb[:, :, 1] = np.linspace(0, 255, 20, dtype=np.uint8)[None, :]
rgba = np.dstack([rgb, rng.integers(0, 256, (24, 20, 1), dtype=np.uint8)])
Image.fromarray(rgb).save(f"{out}/rgb.png")
Image.fromarray(rgba).save(f"{out}/rgba.png")
Image.fromarray(rgb[:, :, 0]).save(f"{out}/grey.png")
Image.fromarray(rgb).convert("P").save(f"{out}/palette.png")
Image.fromarray((rgb[:, :, 0].astype(np.uint16) * 257)).save(f"{out}/grey16.png")
Image.fromarray(rgb[:, :, 0] > 127).save(f"{out}/bits1.png")
Image.fromarray(rgb).save(f"{out}/base.jpg", quality=85)
Image.fromarray(rgb).save(f"{out}/prog.jpg", quality=70, progressive=True)
Image.fromarray(rgb).save(f"{out}/s444.jpg", quality=90, subsampling=0)
Image.fromarray(rgb).save(f"{out}/s420.jpg", quality=60, subsampling=2)
Image.fromarray(rgb[:, :, 0]).save(f"{out}/grey.jpg", quality=80)
import glb_scenes
glb_scenes.room_glb(f"{out}/room_small.glb", nu=8, nv=4, tex=8, seed=5)
print("seeds in", out, sorted(os.listdir(out)))
