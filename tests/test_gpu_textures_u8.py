"""GPU: 8-bit textures kept as RGBA8 words on the device (RT_OPT_TEXTURE_U8, default on).

The reference's loader turns every 8-bit image into one fp32 word per channel, value / 255.0f
(src/MeshData.cpp:556-587), and the shader fetches those words one by one (src/RayScene.cpp:337-380).  At
commit the library stores a four-channel texture whose words are all such values as one word per texel and
returns the same floats through a table, so the image must not change by a single bit -- against the
checker, which reads the uploaded fp32 words, and against the library with the option off."""

import numpy as np
import pytest

from raytracer_amd import capi
from oracle_binding import OracleScene

pytestmark = pytest.mark.gpu

SCENES = [
    ("textured_room", dict(nu=64, nv=16, tex_size=32)),
    ("interior", dict(nu=48, nv=12, tex_size=32)),
    ("fuzz", dict(nu=300, nv=5, seed=21)),
]


def _render(scene, option, rtype, width=64, height=48):
    cam = capi.make_camera(width=width, height=height, bounces=4, ray_per_pixel=4)
    be = capi.Backend(0)
    try:
        be.set_option(capi.RT_OPT_TEXTURE_U8, option)
        be.upload_scene(scene)
        be.resize(width, height)
        be.set_frame(cam, scene.sky, rtype)
        be.render(1, 2, 1)
        image, _ = be.readback()
        info = {k: be.info(v) for k, v in (("uploaded", capi.RT_INFO_TEXTURE_BYTES_UPLOADED), ("device", capi.RT_INFO_TEXTURE_BYTES_DEVICE),
                                          ("rgba8", capi.RT_INFO_TEXTURES_RGBA8), ("fp32", capi.RT_INFO_TEXTURES_FP32))}
    finally:
        be.close()
    return cam, image, info


@pytest.mark.parametrize("rtype", [capi.RT_TYPE_RGB, capi.RT_TYPE_SPECTRAL])
@pytest.mark.parametrize("name,kw", SCENES)
def test_rgba8_textures_do_not_change_the_image(name, kw, rtype):
    scene = capi.Scene(name, **kw)
    cam, packed, info = _render(scene, 1, rtype)
    _, plain, plain_info = _render(scene, 0, rtype)
    want, _ = OracleScene(scene).render(cam, rtype, 1, 2, 1)
    assert np.array_equal(packed, want, equal_nan=True), f"{int(np.any(packed != want, axis=-1).sum())} pixels differ from the checker"
    assert np.array_equal(plain, want, equal_nan=True)
    assert plain_info["rgba8"] == 0 and plain_info["device"] == plain_info["uploaded"]
    assert info["rgba8"] > 0, "no texture of this scene was stored with 8-bit channels: the test exercises nothing"
    assert info["device"] < info["uploaded"]


def test_interior_textures_take_a_quarter_of_the_bytes():
    scene = capi.Scene("interior", nu=48, nv=12, tex_size=64)
    _, _, info = _render(scene, 1, capi.RT_TYPE_RGB, 32, 24)
    assert info["fp32"] == 0 and info["rgba8"] >= 4
    assert info["device"] < 0.36 * info["uploaded"]      # three-channel textures shrink to a third, four-channel ones to a quarter


def test_a_texel_that_is_not_an_eight_bit_value_keeps_its_texture_in_fp32():
    """One word of one texture moved off the k / 255 grid: that texture stays fp32 (the others are still
    packed), and the image follows the changed word exactly as the checker's does."""
    scene = capi.Scene("textured_room", nu=64, nv=16, tex_size=32)
    _, _, before = _render(scene, 1, capi.RT_TYPE_RGB)
    tex = scene.streams[5]
    # the first texture's header is at word 0: width, height, channels
    w, h, c = (int(x) for x in tex[:3].view(np.int32))
    assert c == 4
    texels = tex[5:5 + w * h * c]
    texels[:] = np.nextafter(texels, np.float32(2.0))      # every word one ulp off the grid
    cam, image, after = _render(scene, 1, capi.RT_TYPE_RGB)
    want, _ = OracleScene(scene).render(cam, capi.RT_TYPE_RGB, 1, 2, 1)
    assert np.array_equal(image, want, equal_nan=True)
    assert after["rgba8"] == before["rgba8"] - 1 and after["fp32"] == before["fp32"] + 1


def test_three_channel_textures_read_the_word_behind_their_texel_as_uploaded():
    """The reference's alpha fetch reads the word after the rgb triple of a three-channel texture (the next
    texel's red, or what follows the texture).  The packed form keeps both: the next red in the texel's fourth
    byte, the word behind the last texel as uploaded.  The random scenes use three-channel base-colour textures."""
    for seed in (33, 34, 35):
        scene = capi.Scene("fuzz", nu=400, nv=6, seed=seed)
        cam, image, info = _render(scene, 1, capi.RT_TYPE_RGB)
        want, _ = OracleScene(scene).render(cam, capi.RT_TYPE_RGB, 1, 2, 1)
        assert np.array_equal(image, want, equal_nan=True)
        assert info["rgba8"] > 0
