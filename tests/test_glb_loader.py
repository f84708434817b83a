"""CPU tests of binary glTF ingestion (MeshData::LoadMeshData, reference src/MeshData.cpp:419-590).

The reference ships no .glb file or loader fixture (its models are in an archive that is
not unpacked), so the expected values here are restated from the loader's source: parity
of this row is unpinned against the reference and pinned against that restatement only.
"""
import numpy as np
import pytest

from glb_scenes import quad, room_glb
from glb_writer import FLOAT, U8, U16, U32, GlbBuilder, png_bytes, trs_matrix
from raytracer_amd import capi
from test_host_layer import decode_model, i32


def load(path):
    return capi.Scene("glb:" + str(path))


def model_verts(sc, m=0):
    d = decode_model(sc, m)
    info = i32(sc.streams[3]).reshape(-1, 4)
    end = info[m + 1][3] if m + 1 < sc.models_count else sc.streams[4].size
    return d, sc.streams[4][d["vert_at"]: end].reshape(-1, 8)


def tri_set(verts, ident):
    """Triangles as a sorted list of rounded corner positions (independent of leaf order)."""
    return sorted(tuple(np.round(verts[t[:3], :3].reshape(-1), 5)) for t in ident)


def test_missing_or_malformed_file_yields_no_scene(tmp_path):
    with pytest.raises(capi.RtError):
        load(tmp_path / "absent.glb")
    bad = tmp_path / "bad.glb"
    bad.write_bytes(b"glTF" + b"\x02\0\0\0" + b"\x20\0\0\0" + b"\xff" * 20)
    with pytest.raises(capi.RtError):
        load(bad)
    junk = tmp_path / "junk.glb"
    junk.write_bytes(b"not a glb at all, just text")
    with pytest.raises(capi.RtError):
        load(junk)


def test_node_transforms_and_index_types(tmp_path):
    """world = parent * (T * R * S); positions transformed, normals by the inverse transpose;
    u8/u16/u32 indices; a `matrix` node; a non-indexed primitive adds vertices but no triangles."""
    b = GlbBuilder()
    m = b.material(pbrMetallicRoughness={"baseColorFactor": [0.5, 0.5, 0.5, 1]})
    rng = np.random.default_rng(1)
    expect_pos, expect_nrm, expect_tris = [], [], []

    def add(world, index_type, base, normals=True, indexed=True):
        p = rng.uniform(-1, 1, (6, 3)).astype(np.float32)
        n = rng.normal(size=(6, 3))
        n = (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(np.float32)
        idx = np.array([0, 1, 2, 3, 4, 5, 0, 2, 4])
        prim = b.primitive(p, idx if indexed else None, m, normals=n if normals else None, index_type=index_type)
        expect_pos.append((world[:3, :3] @ p.T.astype(float)).T + world[:3, 3])
        nm = np.linalg.inv(world[:3, :3]).T
        nn = (nm @ (n if normals else np.tile([0, 1, 0], (6, 1))).T.astype(float)).T
        expect_nrm.append(nn / np.linalg.norm(nn, axis=1, keepdims=True))
        if indexed:
            expect_tris.extend((idx.reshape(-1, 3) + base).tolist())
        return prim

    s, c = np.sin(0.4), np.cos(0.4)
    parent = trs_matrix([0.5, -0.25, 2.0], [0, s, 0, c], [2, 1, 0.5])
    child = trs_matrix([0, 1, 0], [s, 0, 0, c], [1, 3, 1])
    mat = np.array([[0, -1, 0, 0.3], [1, 0, 0, 0.1], [0, 0, 2, -0.7], [0, 0, 0, 1]], float)
    mesh_a = b.mesh([add(parent, U8, 0), add(parent, U16, 6, normals=False)])
    mesh_b = b.mesh([add(parent @ child, U32, 12), add(parent @ child, U16, 18, indexed=False)])
    mesh_c = b.mesh([add(mat, U16, 24)])
    nb = b.node(mesh=mesh_b, translation=[0, 1, 0], rotation=[s, 0, 0, c], scale=[1, 3, 1])
    b.node(mesh=mesh_a, children=[nb], root=True, translation=[0.5, -0.25, 2.0], rotation=[0, s, 0, c], scale=[2, 1, 0.5])
    b.node(mesh=mesh_c, root=True, matrix=mat.T.reshape(-1).tolist())   # column-major
    sc = load(b.write(tmp_path / "xf.glb"))

    assert sc.models_count == 1 and sc.light_sources_count == 0
    d, verts = model_verts(sc)
    assert verts.shape[0] == 30 and d["idents"] == 12
    assert np.allclose(verts[:, :3], np.concatenate(expect_pos), atol=2e-6)
    assert np.allclose(verts[:, 3:6], np.concatenate(expect_nrm), atol=2e-6)
    assert np.array_equal(verts[:, 6:8], np.zeros((30, 2), np.float32))      # no TEXCOORD_0
    got = sorted(tuple(t) for t in d["ident"][:, :3].tolist())
    assert got == sorted(tuple(t) for t in expect_tris)
    assert (d["ident"][:, 3] == 1).all()                                     # scene material 0 is the default one


@pytest.mark.parametrize("uv_type,stride,pad", [(FLOAT, None, 0), (FLOAT, 20, 4), (U8, 4, 0), (U16, None, 0), (U16, 8, 8)])
def test_texcoord_formats_and_stride(tmp_path, uv_type, stride, pad):
    b = GlbBuilder()
    m = b.material()
    p, n, _, i = quad([0, 0, 0], [1, 0, 0], [0, 1, 0])
    uv = np.array([[0.0, 1.0], [0.2, 0.4], [0.6, 0.8], [1.0, 0.0]])
    b.node(mesh=b.mesh([b.primitive(p, i, m, normals=n, uvs=uv, uv_type=uv_type, uv_stride=stride, uv_pad_front=pad)]), root=True)
    _, verts = model_verts(load(b.write(tmp_path / "uv.glb")))
    if uv_type == FLOAT:
        expect = uv.astype(np.float32)
    elif uv_type == U8:
        expect = np.round(uv * 255).astype(np.float32) / np.float32(255)
    else:
        expect = np.round(uv * 65535).astype(np.float32) / np.float32(65535)
    assert np.array_equal(verts[:, 6:8], expect)


def test_materials_follow_the_reference_mapping(tmp_path):
    b = GlbBuilder()
    tex = [b.texture(png_bytes(np.full((2, 2, 3), 10 * k, np.uint8))) for k in range(6)]
    b.material()                                                             # 0: all defaults
    b.material(alphaMode="MASK",                                             # 1: cutoff default .5
               pbrMetallicRoughness={"baseColorFactor": [0.1, 0.2, 0.3, 0.4], "metallicFactor": 0.25, "roughnessFactor": 0.75,
                                     "baseColorTexture": {"index": tex[1]}, "metallicRoughnessTexture": {"index": tex[2]}},
               normalTexture={"index": tex[0], "scale": 0.5}, occlusionTexture={"index": tex[5], "strength": 0.3},
               emissiveTexture={"index": tex[3]}, emissiveFactor=[0.5, 0.25, 0.125])
    b.material(alphaMode="BLEND", alphaCutoff=0.9,                           # 2: BLEND ignores the cutoff
               extensions={"KHR_materials_volume": {"thicknessFactor": 0.6},
                           "KHR_materials_transmission": {"transmissionFactor": 0.8, "transmissionTexture": {"index": tex[4]}},
                           "KHR_materials_ior": {"ior": 1.33},
                           "KHR_materials_emissive_strength": {"emissiveStrength": 0.0}})
    b.material(alphaMode="MASK", alphaCutoff=0.3)                            # 3
    prims = []
    for k in range(4):
        p, n, uv, i = quad([k, 0, 0], [0.5, 0, 0], [0, 0.5, 0])
        prims.append(b.primitive(p, i, k, normals=n, uvs=uv))
    p, n, uv, i = quad([9, 0, 0], [0.5, 0, 0], [0, 0.5, 0])
    prims.append(b.primitive(p, i, None, normals=n))                         # no material -> model material 0
    b.node(mesh=b.mesh(prims), root=True)
    sc = load(b.write(tmp_path / "mat.glb"))

    mats = sc.streams[2].reshape(-1, 23)
    assert mats.shape[0] == 1 + 4
    refs = i32(mats[:, :6])           # normal, baseColor, metallicRoughness, emissive, transmission, occlusion
    f = mats[:, 6:]                   # emission rgb, strength, albedo rgb, subsurface, roughness, metallic, specular,
    #                                   specularTint, transmission, ior, alphaCut, normalScale, occlusionStrength
    header = 5 + 2 * 2 * 4            # every 2x2 image is decoded to RGBA
    assert list(refs[1]) == [-1] * 6
    assert np.allclose(f[1], [0, 0, 0, 1, 1, 1, 1, 0, 1, 1, 0.5, 0, 0, 1.5, 0, 1, 1])
    assert list(refs[2]) == [tex[0] * header, tex[1] * header, tex[2] * header, tex[3] * header, -1, tex[5] * header]
    assert np.allclose(f[2], [0.5, 0.25, 0.125, 1, 0.1, 0.2, 0.3, 0, 0.75, 0.25, 0.5, 0, 0, 1.5, 0.5, 0.5, 0.3])
    assert list(refs[3]) == [-1, -1, -1, -1, tex[4] * header, -1]
    assert np.allclose(f[3], [0, 0, 0, 0, 1, 1, 1, 0.6, 1, 1, 0.5, 0, 0.8, 1.33, 0, 1, 1])
    assert np.isclose(f[4][14], 0.3)

    d, verts = model_verts(sc)
    by_x = {int(round(float(verts[t[0], 0]))): t for t in d["ident"]}
    assert [int(by_x[k][3]) for k in (0, 1, 2, 3, 9)] == [1, 2, 3, 4, 1]
    flags = {k: bool(by_x[k][4] != 0) for k in by_x}   # the flag word is a float 0/1; any non-zero bit pattern is true
    assert flags == {0: False, 1: True, 2: False, 3: False, 9: False}        # a transmission texture alone does not count


def test_emissive_primitives_become_the_leading_light_set(tmp_path):
    b = GlbBuilder()
    plain = b.material(pbrMetallicRoughness={"baseColorFactor": [0.7, 0.7, 0.7, 1]})
    lamp = b.material(emissiveFactor=[1, 1, 1], extensions={"KHR_materials_emissive_strength": {"emissiveStrength": 4.0}})
    dark = b.material(emissiveFactor=[0, 0, 0], extensions={"KHR_materials_emissive_strength": {"emissiveStrength": 4.0}})
    glow = b.material(emissiveFactor=[1, 0, 0])   # emissive but no strength extension: stays ordinary geometry
    quads = [(plain, 0), (lamp, 1), (dark, 2), (glow, 3), (lamp, 4)]
    prims = []
    for mat, x in quads:
        p, n, uv, i = quad([x, 0, 0], [0.5, 0, 0], [0, 0.5, 0])
        prims.append(b.primitive(p, i, mat, normals=n))
    b.node(mesh=b.mesh(prims), root=True)
    sc = load(b.write(tmp_path / "lights.glb"))
    assert sc.light_sources_count == 1
    d, verts = model_verts(sc)
    assert d["lights"] == 4 and d["idents"] == 10
    # light vertices are moved to the front: x = 1.. and 4.. first, then 0, 2, 3
    assert [int(np.floor(x)) for x in verts[::4, 0]] == [1, 4, 0, 2, 3]
    # the light-identifier copy keeps its model-local material index: only `identifiers` are
    # re-based onto the scene's material list (src/RayScene.cpp:135-138)
    assert d["light"][:, :3].max() < 8 and (d["light"][:, 3] == lamp).all()
    assert tri_set(verts, d["light"]) == tri_set(verts, [t for t in d["ident"] if t[3] == 1 + lamp])
    mats = sc.streams[2].reshape(-1, 23)
    assert np.allclose(mats[1 + lamp][6:10], [1, 1, 1, 4]) and np.allclose(mats[1 + glow][6:10], [1, 0, 0, 1])


@pytest.mark.parametrize("mode", ["RGBA", "RGB", "L", "LA", "P"])
def test_png_textures_match_an_independent_decoder(tmp_path, mode):
    from PIL import Image
    import io
    rng = np.random.default_rng(5)
    w, h = 13, 7
    if mode == "P":
        img = Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8), "RGB").quantize(17)
    else:
        ch = {"RGBA": 4, "RGB": 3, "L": 1, "LA": 2}[mode]
        arr = rng.integers(0, 255, (h, w, ch), dtype=np.uint8)
        arr[2:5] = np.arange(w)[None, :, None] * 9 + np.arange(3)[:, None, None]   # smooth rows pick other PNG filters
        img = Image.fromarray(arr[..., 0] if ch == 1 else arr, mode)
    buf = io.BytesIO()
    img.save(buf, format="PNG", optimize=True)
    expect = np.asarray(img.convert("RGBA"), np.uint8).astype(np.float32) / np.float32(255)

    b = GlbBuilder()
    t = b.texture(buf.getvalue(), wrap_s=33648, with_sampler=True)
    m = b.material(pbrMetallicRoughness={"baseColorTexture": {"index": t}})
    p, n, uv, i = quad([0, 0, 0], [1, 0, 0], [0, 1, 0])
    b.node(mesh=b.mesh([b.primitive(p, i, m, normals=n, uvs=uv)]), root=True)
    sc = load(b.write(tmp_path / "tex.glb"))
    tex = sc.streams[5]
    assert list(i32(tex[:5])) == [w, h, 4, 33648, 10497]
    assert np.array_equal(tex[5:].reshape(h, w, 4), expect)


def test_room_scene_has_the_expected_structure(tmp_path):
    sc = load(room_glb(tmp_path / "room.glb", nu=12, nv=6))
    assert sc.models_count == 1 and sc.light_sources_count == 1 and sc.object_count == 0
    st = sc.model_stats(0)
    assert st["triangles"] == 10 + 2 + 2 * 12 * 6 + 2 and st["light_triangles"] == 2 and st["valid"] == 1
    d, verts = model_verts(sc)
    # the panel (child of the room node, translated by +1 in y) hangs just under the ceiling
    assert np.allclose(verts[:4, 1], 1.999, atol=1e-6)
    # every position lies in the room's bounds
    assert verts[:, 0].min() >= -1 - 1e-6 and verts[:, 0].max() <= 1 + 1e-6 and verts[:, 1].min() >= -1e-6 and verts[:, 1].max() <= 2 + 1e-6


def test_two_files_make_two_models(tmp_path):
    a = room_glb(tmp_path / "a.glb", nu=8, nv=4)
    b = GlbBuilder()
    m = b.material(pbrMetallicRoughness={"baseColorFactor": [0.2, 0.3, 0.9, 1], "metallicFactor": 0.0})
    p, n, uv, i = quad([-0.3, 0.01, -0.6], [0.6, 0, 0], [0, 0.4, 0.2])
    b.node(mesh=b.mesh([b.primitive(p, i, m, normals=n)]), root=True)
    sc = capi.Scene("glb:" + a + ";" + b.write(tmp_path / "b.glb"))
    assert sc.models_count == 2 and sc.light_sources_count == 1
    mats = sc.streams[2].reshape(-1, 23)
    assert mats.shape[0] == 1 + 6 + 1
    d1 = decode_model(sc, 1)
    assert (d1["ident"][:, 3] == 7).all()      # second file's material 0 -> scene material 7


JPEG_CASES = {
    "444_q92": dict(quality=92, subsampling=0),
    "422_q85": dict(quality=85, subsampling=1),
    "420_q90": dict(quality=90, subsampling=2),
    "420_q50_restart": dict(quality=50, subsampling=2, restart_marker_blocks=3),
    "grey_q80": dict(quality=80),
    "444_optimised": dict(quality=95, subsampling=0, optimize=True),
}


@pytest.mark.parametrize("case", sorted(JPEG_CASES))
def test_jpeg_textures_are_close_to_an_independent_decoder(tmp_path, case):
    """Baseline JPEG: decoders may differ by a few code values (inverse DCT rounding, chroma
    upsampling filter, colour conversion); stb_image, which the reference uses, is not
    available to compare with, so the bar is closeness to libjpeg (through PIL)."""
    from PIL import Image
    import io
    rng = np.random.default_rng(11)
    w, h = 37, 29                      # not multiples of the 8- or 16-pixel MCU
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(xx / 5.0), 128 + 100 * np.cos(yy / 4.0), (xx * 7 + yy * 3) % 256], -1)
    img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
    pil = Image.fromarray(img, "RGB")
    if case.startswith("grey"):
        pil = pil.convert("L")
    buf = io.BytesIO()
    pil.save(buf, format="JPEG", **JPEG_CASES[case])
    expect = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGBA"), np.int32)

    b = GlbBuilder()
    t = b.texture(buf.getvalue())
    b.doc["images"][0]["mimeType"] = "image/jpeg"
    m = b.material(pbrMetallicRoughness={"baseColorTexture": {"index": t}})
    p, n, uv, i = quad([0, 0, 0], [1, 0, 0], [0, 1, 0])
    b.node(mesh=b.mesh([b.primitive(p, i, m, normals=n, uvs=uv)]), root=True)
    tex = load(b.write(tmp_path / "jpeg.glb")).streams[5]
    assert list(i32(tex[:3])) == [w, h, 4]
    got = np.round(tex[5:].reshape(h, w, 4) * 255).astype(np.int32)
    diff = np.abs(got - expect)
    assert diff[..., 3].max() == 0
    assert diff.max() <= 4 and diff.mean() <= 0.6, (diff.max(), diff.mean())


def _texture_of(tmp_path, image_bytes, name="img.glb"):
    b = GlbBuilder()
    t = b.texture(image_bytes)
    m = b.material(pbrMetallicRoughness={"baseColorTexture": {"index": t}})
    p, n, uv, i = quad([0, 0, 0], [1, 0, 0], [0, 1, 0])
    b.node(mesh=b.mesh([b.primitive(p, i, m, normals=n, uvs=uv)]), root=True)
    tex = load(b.write(tmp_path / name)).streams[5]
    w, h, ch = (int(v) for v in i32(tex[:3]))
    return w, h, ch, np.round(tex[5:].reshape(h, w, 4) * 255).astype(np.int32) if w * h else None


PROGRESSIVE_CASES = {
    "444": dict(quality=90, subsampling=0),
    "422": dict(quality=80, subsampling=1),
    "420": dict(quality=92, subsampling=2),
    "420_restart": dict(quality=60, subsampling=2, restart_marker_blocks=2),
    "grey": dict(quality=85),
    "444_q100": dict(quality=100, subsampling=0),
    "420_q15": dict(quality=15, subsampling=2),
}


@pytest.mark.parametrize("case", sorted(PROGRESSIVE_CASES))
def test_progressive_jpeg_decodes_to_the_same_texels_as_its_sequential_twin(tmp_path, case):
    """A progressive file carries the same quantised coefficients as the sequential file of the same
    encoder settings, only in another order (DC first, spectral bands, successive-approximation
    refinements, end-of-band runs): both must decode to identical texels, and close to libjpeg's."""
    from PIL import Image
    import io
    rng = np.random.default_rng(23)
    w, h = 83, 61
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 110 * np.sin(xx / 6.0 + yy / 9.0), 128 + 100 * np.cos(yy / 3.0), (xx * 5 + yy * 11) % 256], -1)
    img = np.clip(img + rng.normal(0, 10, img.shape), 0, 255).astype(np.uint8)
    img[20:40, 30:60] = rng.integers(0, 255, (20, 30, 3))        # busy patch: long refinement runs
    pil = Image.fromarray(img, "RGB")
    if case == "grey":
        pil = pil.convert("L")
    streams = {}
    for progressive in (False, True):
        buf = io.BytesIO()
        pil.save(buf, format="JPEG", progressive=progressive, **PROGRESSIVE_CASES[case])
        streams[progressive] = buf.getvalue()
    assert b"\xff\xc2" in streams[True] and b"\xff\xc2" not in streams[False]
    ws, hs, _, sequential = _texture_of(tmp_path, streams[False], "seq.glb")
    wp, hp, _, progressive = _texture_of(tmp_path, streams[True], "prog.glb")
    assert (ws, hs, wp, hp) == (w, h, w, h)
    assert np.array_equal(progressive, sequential)
    expect = np.asarray(Image.open(io.BytesIO(streams[True])).convert("RGBA"), np.int32)
    diff = np.abs(progressive - expect)
    assert diff.max() <= 5 and diff.mean() <= 0.7, (diff.max(), diff.mean())


def test_truncated_progressive_jpeg_does_not_crash(tmp_path):
    from PIL import Image
    import io
    rng = np.random.default_rng(2)
    buf = io.BytesIO()
    Image.fromarray(rng.integers(0, 255, (40, 40, 3), dtype=np.uint8), "RGB").save(buf, format="JPEG", progressive=True, quality=80)
    whole = buf.getvalue()
    for cut in (len(whole) // 3, len(whole) // 2, len(whole) - 3):
        w, h, _, texels = _texture_of(tmp_path, whole[:cut], "cut.glb")
        assert (w, h) in ((40, 40), (0, 0))      # either what the scans so far give, or refused


PNG_FORMS = [
    # (colour type, bit depth, channels)
    (0, 1, 1), (0, 2, 1), (0, 4, 1), (0, 8, 1), (0, 16, 1),
    (2, 8, 3), (2, 16, 3), (3, 1, 1), (3, 2, 1), (3, 4, 1), (3, 8, 1),
    (4, 8, 2), (4, 16, 2), (6, 8, 4), (6, 16, 4),
]


def _png_expectation(samples, color_type, depth, palette, trns):
    h, w, ch = samples.shape
    to8 = (lambda v: v >> 8) if depth == 16 else (lambda v: v * {1: 255, 2: 85, 4: 17, 8: 1}[depth])
    out = np.full((h, w, 4), 255, np.int32)
    if color_type == 3:
        out[..., :3] = np.asarray(palette)[samples[..., 0]]
        if trns is not None:
            alpha = np.full(256, 255)
            alpha[:len(trns)] = list(trns)
            out[..., 3] = alpha[samples[..., 0]]
    elif color_type in (0, 4):
        out[..., :3] = to8(samples[..., :1])
        if color_type == 4:
            out[..., 3] = to8(samples[..., 1])
    else:
        out[..., :3] = to8(samples[..., :3])
        if color_type == 6:
            out[..., 3] = to8(samples[..., 3])
    return out


@pytest.mark.parametrize("interlace", [False, True])
@pytest.mark.parametrize("form", PNG_FORMS, ids=lambda f: "type%d_%dbit" % f[:2])
def test_png_of_every_colour_type_depth_and_interlace(tmp_path, form, interlace):
    """Every colour type / bit depth pair of the PNG standard, plain and Adam7-interlaced, every row
    filter: texels are the samples scaled to 8 bits (16-bit: the high byte)."""
    from glb_writer import raw_png
    from PIL import Image
    import io
    color_type, depth, ch = form
    rng = np.random.default_rng(color_type * 100 + depth)
    w, h = 19, 11                       # not multiples of 8: ragged Adam7 passes and partial bytes
    samples = rng.integers(0, 2 ** depth, (h, w, ch))
    samples[3:6] = (np.arange(w)[None, :, None] * 3 + np.arange(3)[:, None, None]) % (2 ** depth)
    palette = rng.integers(0, 256, (2 ** depth, 3)) if color_type == 3 else None
    trns = bytes(rng.integers(0, 256, 2 ** depth // 2 + 1).tolist()) if color_type == 3 else None
    for filter_type in range(5):
        stream = raw_png(samples, color_type, depth, interlace, palette, trns, filter_type)
        gw, gh, gch, got = _texture_of(tmp_path, stream)
        assert (gw, gh, gch) == (w, h, 4)
        assert np.array_equal(got, _png_expectation(samples, color_type, depth, palette, trns)), filter_type
    if depth == 8:      # the writer above against an independent reader
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(stream)).convert("RGBA"), np.int32), got)


@pytest.mark.parametrize("interlace", [False, True])
@pytest.mark.parametrize("color_type,depth", [(0, 2), (0, 8), (0, 16), (2, 8), (2, 16)])
def test_png_colour_key_makes_matching_pixels_transparent(tmp_path, color_type, depth, interlace):
    from glb_writer import raw_png
    rng = np.random.default_rng(9)
    ch = 1 if color_type == 0 else 3
    samples = rng.integers(0, 2 ** depth, (9, 10, ch))
    key = samples[4, 7].copy()
    samples[1, 2] = key
    near = key.copy()
    near[-1] ^= 1                       # differs in the lowest bit only: not transparent, even at 16 bits
    samples[6, 3] = near
    trns = b"".join(int(v).to_bytes(2, "big") for v in key)
    _, _, _, got = _texture_of(tmp_path, raw_png(samples, color_type, depth, interlace, trns=trns))
    expect = _png_expectation(samples, color_type, depth, None, None)
    expect[..., 3] = np.where((samples == key).all(-1), 0, 255)
    assert expect[1, 2, 3] == 0 and expect[4, 7, 3] == 0 and expect[6, 3, 3] == 255
    assert np.array_equal(got, expect)


@pytest.mark.parametrize("size", [(1, 1), (2, 3), (5, 1), (1, 9), (8, 8), (9, 17)])
def test_interlaced_png_with_empty_passes(tmp_path, size):
    """Images narrower or shorter than the Adam7 lattice have passes without any pixel."""
    from glb_writer import raw_png
    w, h = size
    samples = np.random.default_rng(w * 31 + h).integers(0, 256, (h, w, 4))
    gw, gh, _, got = _texture_of(tmp_path, raw_png(samples, 6, 8, True, filter_type=4))
    assert (gw, gh) == (w, h) and np.array_equal(got, samples)


def test_png_outside_the_standard_is_refused(tmp_path):
    from glb_writer import raw_png
    samples = np.zeros((4, 4, 3), np.int64)
    for stream in (raw_png(samples, 2, 4),                       # RGB has no 4-bit form
                   raw_png(samples[..., :1], 3, 16, palette=np.zeros((2, 3))),
                   raw_png(samples, 2, 8)[:-40],                 # data cut short
                   raw_png(samples, 2, 8, filter_type=0).replace(b"IDAT", b"IDAX")):
        w, h, _, _ = _texture_of(tmp_path, stream)
        assert (w, h) == (0, 0)


# ---- hostile files: the loader is the trust boundary for external data --------------------------
def _two_triangles():
    pos = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
    return pos, [0, 1, 2, 0, 2, 3]


def _load_or_none(path):
    try:
        return load(path)
    except capi.RtError:
        return None


def test_position_accessor_must_be_packed_float3(tmp_path):
    """POSITION and NORMAL are read as tightly packed float3.  An accessor declared SCALAR, VEC2 or
    byte-typed passes a per-element bounds check but would be over-read by up to 3x: such a primitive
    contributes nothing (and nothing is read past the binary chunk)."""
    pos, idx = _two_triangles()
    for kind, component in (("SCALAR", FLOAT), ("VEC2", FLOAT), ("VEC3", U8), ("VEC4", FLOAT)):
        g = GlbBuilder()
        prim = g.primitive(pos, idx)
        acc = g.doc["accessors"][prim["attributes"]["POSITION"]]
        acc["type"], acc["componentType"] = kind, component
        acc["count"] = len(g.bin)            # as many one-byte elements as the buffer holds
        g.node(mesh=g.mesh([prim]), root=True)
        sc = _load_or_none(g.write(tmp_path / f"pos_{kind}_{component}.glb"))
        assert sc is None or sc.model_stats(0)["triangles"] == 0


def test_normal_accessor_of_the_wrong_shape_is_ignored(tmp_path):
    pos, idx = _two_triangles()
    g = GlbBuilder()
    prim = g.primitive(pos, idx, normals=np.tile([0, 0, 1.0], (4, 1)))
    g.doc["accessors"][prim["attributes"]["NORMAL"]]["type"] = "SCALAR"     # 4 floats declared, 12 would be read
    g.node(mesh=g.mesh([prim]), root=True)
    sc = load(g.write(tmp_path / "nrm.glb"))
    d, verts = model_verts(sc)
    assert sc.model_stats(0)["triangles"] == 2
    assert np.allclose(verts[:, 3:6], [0, 1, 0])                             # the loader's default normal


@pytest.mark.parametrize("index_type", [U8, U16, U32])
def test_indices_beyond_the_vertex_count_drop_their_triangle(tmp_path, index_type):
    pos, _ = _two_triangles()
    big = {U8: 200, U16: 60000, U32: 0xFFFFFFF0}[index_type]
    g = GlbBuilder()
    g.node(mesh=g.mesh([g.primitive(pos, [0, 1, 2, 0, 2, big, 4, 1, 2, 0, 2, 3], index_type=index_type)]), root=True)
    sc = load(g.write(tmp_path / "idx.glb"))
    d, verts = model_verts(sc)
    assert sc.model_stats(0)["triangles"] == 2                               # the two triangles with valid corners
    assert d["ident"][:, :3].max() < 4


def test_index_accessor_of_the_wrong_shape_or_type_is_refused(tmp_path):
    pos, idx = _two_triangles()
    for kind, component in (("VEC3", U16), ("SCALAR", FLOAT)):
        g = GlbBuilder()
        prim = g.primitive(pos, idx)
        acc = g.doc["accessors"][prim["indices"]]
        acc["type"], acc["componentType"] = kind, component
        g.node(mesh=g.mesh([prim]), root=True)
        sc = _load_or_none(g.write(tmp_path / f"idx_{kind}.glb"))
        assert sc is None or sc.model_stats(0)["triangles"] == 0


@pytest.mark.parametrize("field,value", [("count", 1e30), ("count", -5), ("byteOffset", 1e18), ("byteOffset", -4)])
def test_accessor_fields_out_of_range_are_refused(tmp_path, field, value):
    pos, idx = _two_triangles()
    g = GlbBuilder()
    prim = g.primitive(pos, idx)
    g.doc["accessors"][prim["attributes"]["POSITION"]][field] = value
    g.node(mesh=g.mesh([prim]), root=True)
    sc = _load_or_none(g.write(tmp_path / "range.glb"))
    assert sc is None or sc.model_stats(0)["triangles"] == 0


def test_strided_view_whose_packed_extent_leaves_the_buffer_is_refused(tmp_path):
    """byteStride 4 makes every 12-byte element start inside the chunk while `count` packed float3
    would not fit: the packed read must not happen."""
    pos, idx = _two_triangles()
    g = GlbBuilder()
    prim = g.primitive(pos, idx)
    acc = g.doc["accessors"][prim["attributes"]["POSITION"]]
    g.doc["bufferViews"][acc["bufferView"]]["byteStride"] = 4
    acc["count"] = (len(g.bin) - 12) // 4 + 1
    g.node(mesh=g.mesh([prim]), root=True)
    sc = _load_or_none(g.write(tmp_path / "stride.glb"))
    assert sc is None or sc.model_stats(0)["triangles"] == 0


def test_jpeg_header_claiming_a_huge_image_leaves_the_texture_empty(tmp_path):
    import struct
    sof = b"\xff\xc0" + struct.pack(">HBHHB", 17, 8, 65535, 65535, 3) + bytes([1, 0x11, 0, 2, 0x11, 0, 3, 0x11, 0])
    fake = b"\xff\xd8" + sof + b"\xff\xd9"
    pos, idx = _two_triangles()
    g = GlbBuilder()
    tex = g.texture(fake)
    mat = g.material(pbrMetallicRoughness={"baseColorTexture": {"index": tex}})
    g.node(mesh=g.mesh([g.primitive(pos, idx, material=mat, uvs=pos[:, :2])]), root=True)
    sc = load(g.write(tmp_path / "huge.glb"))
    assert sc.model_stats(0)["triangles"] == 2
    assert sc.streams[5].size < 64                                           # header words at most, no texels


def test_headers_within_the_size_limit_but_without_the_data_allocate_nothing(tmp_path):
    """A few header bytes claiming 16384 x 16384 (inside the readers' size limit) must not cost gigabytes:
    deflate expands at most 1032 : 1 and a JPEG block needs at least one bit, so both readers refuse a
    stream too short for its header before sizing any buffer from it."""
    import resource
    import struct
    import zlib

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xffffffff)
    png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 16384, 16384, 16, 6, 0, 0, 0)) +
           chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b""))
    sof = b"\xff\xc2" + struct.pack(">HBHHB", 17, 8, 16384, 16384, 3) + bytes([1, 0x22, 0, 2, 0x11, 0, 3, 0x11, 0])
    jpeg = b"\xff\xd8" + sof + b"\xff\xd9"
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    for name, stream in (("png", png), ("jpeg", jpeg)):
        pos, idx = _two_triangles()
        g = GlbBuilder()
        mat = g.material(pbrMetallicRoughness={"baseColorTexture": {"index": g.texture(stream)}})
        g.node(mesh=g.mesh([g.primitive(pos, idx, material=mat, uvs=pos[:, :2])]), root=True)
        sc = load(g.write(tmp_path / f"claim_{name}.glb"))
        assert sc.model_stats(0)["triangles"] == 2
        assert sc.streams[5].size < 64
    grown_kib = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before
    assert grown_kib < 256 * 1024, f"peak resident size grew by {grown_kib >> 10} MiB"
