"""Shared description of the committed golden fixtures (tests/golden/*.npz)."""
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = {
    "cornell_quads": ("cornell_quads", {}, 48, 36, 4, 4, 2),
    "cornell_knot": ("cornell_knot", dict(nu=96, nv=16), 48, 36, 4, 4, 2),
    "textured_room": ("textured_room", dict(nu=48, nv=12, tex_size=32), 48, 36, 4, 4, 2),
    "mixed": ("mixed", dict(nu=64, nv=12), 48, 36, 3, 1, 3),
    "prism": ("prism", {}, 40, 40, 4, 4, 2),
    "interior": ("interior", dict(nu=40, nv=10, tex_size=32), 48, 36, 4, 4, 2),
    "unlit_knot": ("unlit_knot", dict(nu=64, nv=12), 40, 30, 3, 4, 2),
    # binary glTF input: tests/golden/room.glb is data written once by glb_scenes.room_glb(nu=32, nv=10, tex=16)
    "glb_room": ("glb:{golden}/room.glb", {}, 48, 36, 4, 4, 2),
}

# Spectral integrator in hero-wavelength mode (the reference's HERO_WAVELENGTH_ENABLE 1): image only,
# in hero_<case>.npz.  Same tuple as above.
HERO_CASES = {
    "prism": ("prism", {}, 40, 40, 4, 4, 2),
    "mixed": ("mixed", dict(nu=64, nv=12), 48, 36, 3, 1, 3),
}


def scene_name(case):
    return CASES[case][0].replace("{golden}", HERE)


def load(case):
    return np.load(os.path.join(HERE, f"{case}.npz"))
