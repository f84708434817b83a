"""Synthetic .glb scenes shared by the loader tests (CPU) and the render-parity tests (GPU)."""
import numpy as np

from glb_writer import FLOAT, U8, U16, U32, GlbBuilder, png_bytes, torus


def quad(p0, e1, e2):
    p0, e1, e2 = (np.asarray(x, float) for x in (p0, e1, e2))
    n = np.cross(e1, e2)
    n /= np.linalg.norm(n)
    pos = np.array([p0, p0 + e1, p0 + e1 + e2, p0 + e2])
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], float)
    return pos, np.tile(n, (4, 1)), uv, np.array([[0, 1, 2], [0, 2, 3]])


def checker_rgba(size, seed):
    rng = np.random.default_rng(seed)
    img = np.zeros((size, size, 4), np.uint8)
    yy, xx = np.mgrid[0:size, 0:size]
    cell = ((xx // 4) + (yy // 4)) % 2
    img[..., :3] = np.where(cell[..., None] == 1, rng.integers(120, 255, 3), rng.integers(10, 90, 3))
    img[..., :3] += rng.integers(0, 8, (size, size, 3)).astype(np.uint8)
    img[..., 3] = np.where(((xx // 2) + (yy // 2)) % 3 == 0, 40, 255)   # holes for the alpha cut
    return img


def room_glb(path, nu=24, nv=10, tex=16, seed=3):
    """Open-front room (z = -1 open towards the default camera) with an emissive ceiling panel,
    a glass torus under a rotated + scaled parent node and an alpha-cut textured card."""
    b = GlbBuilder()
    white = b.material(pbrMetallicRoughness={"baseColorFactor": [0.8, 0.8, 0.8, 1], "metallicFactor": 0.0, "roughnessFactor": 0.9})
    lamp = b.material(emissiveFactor=[1.0, 0.9, 0.8], extensions={"KHR_materials_emissive_strength": {"emissiveStrength": 15.0}},
                      pbrMetallicRoughness={"metallicFactor": 0.0})
    red = b.material(pbrMetallicRoughness={"baseColorFactor": [0.65, 0.05, 0.05, 1], "metallicFactor": 0.0})
    green = b.material(pbrMetallicRoughness={"baseColorFactor": [0.12, 0.45, 0.15, 1], "metallicFactor": 0.0})
    t_card = b.texture(png_bytes(checker_rgba(tex, seed)), wrap_s=33071, wrap_t=33648)
    card = b.material(alphaMode="MASK", alphaCutoff=0.4, doubleSided=True,
                      pbrMetallicRoughness={"baseColorTexture": {"index": t_card}, "metallicFactor": 0.0, "roughnessFactor": 0.6})
    glass = b.material(pbrMetallicRoughness={"baseColorFactor": [0.95, 0.97, 0.95, 1], "metallicFactor": 0.0, "roughnessFactor": 0.05},
                       extensions={"KHR_materials_transmission": {"transmissionFactor": 1.0}, "KHR_materials_ior": {"ior": 1.45}})

    # room in a [-1,1]^3 frame, placed by the node at y = 1
    walls = [
        (quad([-1, -1, -1], [0, 0, 2], [2, 0, 0]), white),    # floor, normal +y
        (quad([-1, 1, -1], [2, 0, 0], [0, 0, 2]), white),     # ceiling, normal -y
        (quad([-1, -1, 1], [0, 2, 0], [2, 0, 0]), white),     # back, normal -z
        (quad([-1, -1, -1], [0, 2, 0], [0, 0, 2]), red),      # left, normal +x
        (quad([1, -1, -1], [0, 0, 2], [0, 2, 0]), green),     # right, normal -x
    ]
    prims = [b.primitive(p, i, m, normals=n, uvs=uv, index_type=U8) for (p, n, uv, i), m in walls]
    room = b.mesh(prims)
    p, n, uv, i = quad([-0.35, 0.999, -0.35], [0.7, 0, 0], [0, 0, 0.7])
    panel = b.mesh([b.primitive(p, i, lamp, normals=n)])
    panel_node = b.node(mesh=panel)
    b.node(mesh=room, children=[panel_node], root=True, translation=[0, 1, 0])

    p, n, uv, i = torus(nu, nv)
    ring = b.mesh([b.primitive(p, i.reshape(-1), glass, normals=n, uvs=uv, index_type=U32 if len(p) > 65535 else U16)])
    ring_node = b.node(mesh=ring, scale=[1.2, 1.2, 1.2])
    s = np.sin(0.35), np.cos(0.35)
    b.node(children=[ring_node], root=True, translation=[0.25, 0.75, 0.1], rotation=[s[0], 0, 0, s[1]])

    p, n, uv, i = quad([-0.8, 0.05, -0.1], [0.7, 0, -0.3], [0, 0.9, 0])
    b.node(mesh=b.mesh([b.primitive(p, i, card, normals=n, uvs=uv, uv_type=U16, uv_stride=8, uv_pad_front=8)]), root=True)
    return b.write(path)
