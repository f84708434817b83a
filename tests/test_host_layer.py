"""CPU tests of the C++ host layer (wire format, BVH invariants) and of the C-ABI surface."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from raytracer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def i32(words):
    return np.asarray(words, dtype=np.float32).view(np.int32)


def test_abi_exports_every_declared_symbol():
    for header, lib, declared in (("rt_hip.h", capi.hip, capi.HIP_SYMBOLS), ("rt_host.h", capi.host, capi.HOST_SYMBOLS)):
        text = open(os.path.join(ROOT, "include", header)).read()
        names = set(re.findall(r"\b(rth?_[a-z0-9_]+)\s*\(", text))
        assert names == set(declared), (header, names ^ set(declared))
        for n in declared:
            assert hasattr(lib, n), n


def test_abi_rejects_bad_arguments_without_a_gpu():
    hip = capi.hip
    assert hip.rt_create(None, 0) == -1
    assert hip.rt_upload(None, 1, None, 0) == -1
    assert hip.rt_render(None, 1, 1, 1) == -1
    assert hip.rt_strerror(0) == b"ok" and hip.rt_strerror(-5) == b"inconsistent scene streams"
    ctx = C.c_void_p()
    rc = hip.rt_create(C.byref(ctx), 0)
    if rc == 0:  # a GPU is present
        assert hip.rt_upload(ctx, 9, None, 0) == -1
        assert hip.rt_render(ctx, 1, 1, 1) == -2   # nothing committed yet
        hip.rt_destroy(ctx)
    else:
        assert rc == -4 and not ctx.value  # RT_ERR_NO_DEVICE: the product path fails loudly, no fallback


def test_primitive_stream_layout():
    """type, material, payload per primitive (reference src/RayScene.cpp:55-60)."""
    sc = capi.Scene("three_color")
    w, wi = sc.streams[1], i32(sc.streams[1])
    # sphere(6 words), quad(12), sphere, sphere, sphere
    assert w.size == 6 + 12 + 6 * 3
    assert wi[0] == 0 and wi[1] == 1                        # Sphere, material 1 (0 is the default material)
    assert np.allclose(w[2:6], [0, 0, 1, 0.12])
    assert wi[6] == 1 and wi[7] == 2                        # Quad, material 2
    assert np.allclose(w[8:18], [-5, -0.1, 0, 10, 0, 0, 0, 0, 10, 0])
    assert wi[18] == 0 and wi[19] == 3
    # 23 words per material, 1 default + 5 added
    assert sc.streams[2].size == 23 * 6
    m3 = sc.streams[2][23 * 3: 23 * 4]
    assert list(i32(m3[:6])) == [-1] * 6
    assert np.allclose(m3[6:10], [1, 0.2, 0.2, 140])        # emission colour + strength of the red light
    assert np.allclose(m3[10:13], [1, 1, 1]) and m3[14] == 1.0 and m3[19] == 1.5   # albedo, roughness, ior


def test_cornell_builder_matches_reference_layout():
    sc = capi.Scene("cornell_quads")
    wi, w = i32(sc.streams[1]), sc.streams[1]
    # light quad first, 12 words each; q of the light = pos + ((2-.6)/2, 2-1e-4, 2/2-.6)
    assert wi[0] == 1 and wi[1] == 1
    assert np.allclose(w[2:5], [-1 + 0.7, 0 + 2 - 1e-4, -1 + 0.4], atol=1e-6)
    assert np.allclose(w[5:11], [0.6, 0, 0, 0, 0, 0.6])
    # 7th quad (front) is cull-faced
    front = 12 * 6
    assert wi[front] == 1 and w[front + 11] == 1.0
    # red then green wall albedo
    mats = sc.streams[2].reshape(-1, 23)
    assert np.allclose(mats[2, 10:13], [.65, .05, .05]) and np.allclose(mats[3, 10:13], [.12, .45, .15])


def decode_model(sc, m):
    info = i32(sc.streams[3]).reshape(-1, 4)[m]
    idents, lights, nodes, loc = (int(x) for x in info)
    words = sc.streams[4]
    at = loc
    light = i32(words[at: at + lights * 5]).reshape(-1, 5); at += lights * 5
    node_w = words[at: at + nodes * 9].reshape(-1, 9); at += nodes * 9
    ident = i32(words[at: at + idents * 5]).reshape(-1, 5).copy(); ident_f = words[at: at + idents * 5].reshape(-1, 5); at += idents * 5
    return dict(idents=idents, lights=lights, nodes=nodes, loc=loc, light=light, node=node_w, ident=ident, ident_f=ident_f, vert_at=at)


@pytest.mark.parametrize("bvh", ["reference", "sah"])
@pytest.mark.parametrize("name,kw", [("cornell_knot", dict(nu=64, nv=12)), ("mixed", dict(nu=48, nv=8)), ("textured_room", dict(nu=32, nv=8, tex_size=16))])
def test_model_blob_and_bvh_invariants(name, kw, bvh):
    sc = capi.Scene(name, bvh=bvh, **kw)
    words = sc.streams[4]
    seen_non_light = False
    expected_loc = 0
    for m in range(sc.models_count):
        d = decode_model(sc, m)
        stats = None
        # light-bearing models come first in the info stream (src/RayScene.cpp:62-86)
        if d["lights"] == 0:
            seen_non_light = True
        else:
            assert not seen_non_light
        assert d["loc"] == expected_loc
        nverts_words = (i32(sc.streams[3]).reshape(-1, 4)[m + 1][3] if m + 1 < sc.models_count else words.size) - d["vert_at"]
        assert nverts_words % 8 == 0
        verts = words[d["vert_at"]: d["vert_at"] + nverts_words].reshape(-1, 8)
        expected_loc = d["vert_at"] + nverts_words

        node = d["node"]
        left, right, leaf = i32(node[:, 6]), i32(node[:, 7]), node[:, 8] != 0
        covered = np.zeros(d["idents"], bool)
        depth = {0: 1}
        for n in range(d["nodes"]):
            if leaf[n]:
                assert 1 <= right[n] - left[n] <= 2                      # leaf <= 2 triangles (src/BVHTree.cpp:18)
                assert not covered[left[n]: right[n]].any()
                covered[left[n]: right[n]] = True
                for t in range(left[n], right[n]):                        # containment (src/TraceableObject.cpp:138-143)
                    p = verts[d["ident"][t, :3], :3]
                    assert (p >= node[n, 0:3]).all() and (p <= node[n, 3:6]).all()
            else:
                assert left[n] == n + 1 and right[n] > left[n]            # pre-order layout (src/BVHTree.cpp:40-45)
                depth[left[n]] = depth[right[n]] = depth[n] + 1
                # Note: a child's box need NOT lie inside its parent's.  Every union pads
                # again, so the pad a box ends up with depends on where its extreme
                # triangle sits in the range order, which nth_element changes.
        assert covered.all()
        assert max(depth.values()) <= 32
        assert d["nodes"] == 2 * int(leaf.sum()) - 1
        if bvh == "reference":
            assert int(leaf.sum()) == 1 << int(np.ceil(np.log2(max(d["idents"], 2) / 2)))   # halving until <= 2 per leaf
        # light identifiers index the FIRST vertices of the model (src/MeshData.cpp:444-454)
        if d["lights"]:
            assert d["light"][:, :3].max() < 4 * d["lights"]


def test_texture_stream_and_material_offsets():
    sc = capi.Scene("textured_room", nu=16, nv=6, tex_size=16)
    tex = sc.streams[5]
    ti = i32(tex)
    # 5 textures: 16x16x4, 16x16x3, 16x16x3, 8x8x3, 16x8x3, each with a 5-word header
    sizes = [(16, 16, 4), (16, 16, 3), (16, 16, 3), (8, 8, 3), (16, 8, 3)]
    at, offsets = 0, []
    for (w, h, c) in sizes:
        offsets.append(at)
        assert list(ti[at: at + 3]) == [w, h, c]
        texels = tex[at + 5: at + 5 + w * h * c]
        assert texels.min() >= 0 and texels.max() <= 1
        assert np.array_equal(np.round(texels * 255) / np.float32(255), texels)   # k/255.0f like an 8-bit source
        at += 5 + w * h * c
    assert at == tex.size
    assert list(ti[offsets[1] + 3: offsets[1] + 5]) == [33071, 33648]
    mats = sc.streams[2].reshape(-1, 23)
    refs = i32(mats[:, :6])
    # material 0 default; model materials follow with texture numbers turned into header offsets
    assert list(refs[0]) == [-1] * 6
    lamp, floor = refs[1], refs[2]
    assert lamp[3] == offsets[3]
    assert floor[1] == offsets[0] and floor[2] == offsets[1] and floor[0] == offsets[2]
    # hasTextures flag is derived from the material (transmission texture alone does not count)
    d = decode_model(sc, 0)
    flags = d["ident_f"][:, 4]
    mat_of = d["ident"][:, 3]
    has = np.array([(refs[m][[0, 1, 2, 3, 5]] != -1).any() for m in mat_of])
    assert np.array_equal(flags != 0, has)


def test_sky_colour_is_gamma_expanded():
    sc = capi.Scene("roughness_demo")   # default sky (0.5, 0.7, 1.0) ^ 2.2
    assert np.allclose(sc.sky, np.array([0.5, 0.7, 1.0]) ** 2.2, rtol=1e-6)
    assert np.allclose(capi.Scene("cornell_knot", nu=8, nv=4).sky, 0)


def test_full_size_bench_scene_shape():
    """The dragon-class stand-in: 872,448 triangles, median-split depth 20, 1,048,575 nodes."""
    sc = capi.Scene("cornell_knot", nu=2048, nv=213)
    st = sc.model_stats(1)
    assert st["triangles"] == 872448 and st["vertices"] == 436224
    assert st["nodes"] == 2 * st["leaves"] - 1 == 1048575
    assert st["max_height"] == 20 and st["max_leaf_tris"] == 2 and st["valid"] == 1


def test_sah_trees_render_the_same_image_with_fewer_visits():
    """The opt-in SAH build changes only which nodes are opened: same closest hits (ties aside),
    hence the same paths and the same image from the same seeds."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import OracleScene
    cam = capi.make_camera(width=64, height=48, bounces=3, ray_per_pixel=1)
    out = {}
    for mode in ("reference", "sah"):
        sc = capi.Scene("mixed", nu=96, nv=16, bvh=mode)
        orc = OracleScene(sc)
        img, st = orc.render(cam, 0, 1, 2, 1)
        _, _, t, u, v, _ = orc.trace_primary(cam, frame=1)
        out[mode] = (img, st, t)
    assert np.array_equal(out["reference"][2], out["sah"][2])
    differing = int(np.any(out["reference"][0] != out["sah"][0], axis=-1).sum())
    assert differing <= 2                                   # exact-tie pixels only, if any
    assert out["sah"][1]["node_visits"] < 0.85 * out["reference"][1]["node_visits"]
    assert out["sah"][1]["rays"] in range(out["reference"][1]["rays"] - 8, out["reference"][1]["rays"] + 9)


def test_headers_compile_as_c99_and_link_from_c(tmp_path):
    """The boundary is a C ABI: both public headers must be plain C, and a C program must link against the libraries."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "abi.c"
    src.write_text(
        "#include <stddef.h>\n#include <stdint.h>\n#include \"rt_hip.h\"\n#include \"rt_host.h\"\n"
        "int main(void) {\n"
        "    rt_ctx *ctx = NULL;\n"
        "    int rc = rt_create(&ctx, 0);\n"
        "    if (rc == RT_OK) { rc = rt_display_rgba8_end(ctx, 0) == RT_ERR_STATE ? RT_OK : 1; rt_destroy(ctx); return rc; }\n"
        "    return rc == RT_ERR_NO_DEVICE ? 0 : 1;   /* no GPU: refused, nothing else */\n"
        "}\n")
    lib = os.path.join(ROOT, "raytracer_amd", "lib")
    exe = tmp_path / "abi"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", lib, "-lrt_host", "-lrt_hip", "-Wl,-rpath," + lib], check=True)
    assert subprocess.run([str(exe)]).returncode == 0
