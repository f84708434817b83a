"""PNG / PFM writers of the host layer, read back with independent decoders."""
import numpy as np
import pytest

from raytracer_amd import capi


def test_png_round_trip_is_flipped_to_top_down(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    path = tmp_path / "frame.png"
    capi.write_png(path, img)
    back = np.asarray(Image.open(path))
    assert back.shape == img.shape and np.array_equal(back, img[::-1])
    with pytest.raises(capi.RtError):
        capi.write_png(tmp_path / "no_such_dir" / "x.png", img)


def test_pfm_holds_the_scaled_colour_channels(tmp_path):
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 40, (9, 14, 4)).astype(np.float32)
    path = tmp_path / "frame.pfm"
    capi.write_pfm(path, img, scale=0.25)
    raw = path.read_bytes()
    head = raw.split(b"\n", 3)
    assert head[0] == b"PF" and head[1] == b"14 9" and float(head[2]) < 0      # negative scale = little endian
    data = np.frombuffer(head[3], dtype="<f4").reshape(9, 14, 3)
    assert np.array_equal(data, img[..., :3] * np.float32(0.25))


def test_own_png_is_readable_by_the_gltf_texture_decoder(tmp_path):
    """Writer and loader agree: a PNG written here comes back as k/255 texels through a .glb."""
    from glb_writer import GlbBuilder
    from glb_scenes import quad
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (6, 5, 4), dtype=np.uint8)
    png = tmp_path / "t.png"
    capi.write_png(png, img)
    b = GlbBuilder()
    t = b.texture(png.read_bytes())
    m = b.material(pbrMetallicRoughness={"baseColorTexture": {"index": t}})
    p, n, uv, i = quad([0, 0, 0], [1, 0, 0], [0, 1, 0])
    b.node(mesh=b.mesh([b.primitive(p, i, m, normals=n, uvs=uv)]), root=True)
    sc = capi.Scene("glb:" + b.write(tmp_path / "t.glb"))
    texels = sc.streams[5][5:].reshape(6, 5, 4)
    assert np.array_equal(texels, img[::-1].astype(np.float32) / np.float32(255))
