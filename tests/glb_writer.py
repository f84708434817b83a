"""Writes small binary glTF 2.0 files for the loader tests (test helper, not product code)."""
import io
import json
import struct

import numpy as np

FLOAT, U8, U16, U32 = 5126, 5121, 5123, 5125
_DTYPE = {FLOAT: np.float32, U8: np.uint8, U16: np.uint16, U32: np.uint32}


class GlbBuilder:
    def __init__(self):
        self.bin = bytearray()
        self.doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": []}], "nodes": [], "meshes": [],
                    "materials": [], "accessors": [], "bufferViews": [], "buffers": [{"byteLength": 0}]}

    # ---- raw storage ------------------------------------------------------------------
    def view(self, payload, stride=None):
        while len(self.bin) % 4:
            self.bin.append(0)
        view = {"buffer": 0, "byteOffset": len(self.bin), "byteLength": len(payload)}
        if stride:
            view["byteStride"] = stride
        self.bin += payload
        self.doc["bufferViews"].append(view)
        return len(self.doc["bufferViews"]) - 1

    def accessor(self, array, component, kind, stride=None, pad_front=0):
        """`array` is (count, n); with `stride` each element is padded to that many bytes and
        the accessor gets a byteOffset of pad_front inside its view."""
        a = np.ascontiguousarray(array, dtype=_DTYPE[component])
        count = a.shape[0]
        if stride:
            elem = a.itemsize * (a.shape[1] if a.ndim > 1 else 1)
            rows = np.zeros((count, stride), np.uint8)
            rows[:, :elem] = a.reshape(count, -1).view(np.uint8)
            payload = bytes(pad_front) + rows.tobytes()
        else:
            payload = bytes(pad_front) + a.tobytes()
        v = self.view(payload, stride)
        acc = {"bufferView": v, "componentType": component, "count": count, "type": kind}
        if pad_front:
            acc["byteOffset"] = pad_front
        self.doc["accessors"].append(acc)
        return len(self.doc["accessors"]) - 1

    # ---- scene graph --------------------------------------------------------------------
    def primitive(self, positions, indices, material=None, normals=None, uvs=None, index_type=U16, uv_type=FLOAT, uv_stride=None,
                  uv_pad_front=0):
        attrs = {"POSITION": self.accessor(positions, FLOAT, "VEC3")}
        if normals is not None:
            attrs["NORMAL"] = self.accessor(normals, FLOAT, "VEC3")
        if uvs is not None:
            if uv_type == U8:
                uvs = np.round(np.asarray(uvs) * 255)
            elif uv_type == U16:
                uvs = np.round(np.asarray(uvs) * 65535)
            attrs["TEXCOORD_0"] = self.accessor(uvs, uv_type, "VEC2", uv_stride, uv_pad_front)
        prim = {"attributes": attrs, "mode": 4}
        if indices is not None:
            prim["indices"] = self.accessor(np.asarray(indices).reshape(-1, 1), index_type, "SCALAR")
        if material is not None:
            prim["material"] = material
        return prim

    def mesh(self, primitives):
        self.doc["meshes"].append({"primitives": primitives})
        return len(self.doc["meshes"]) - 1

    def node(self, mesh=None, children=None, root=False, **trs):
        n = dict(trs)
        if mesh is not None:
            n["mesh"] = mesh
        if children:
            n["children"] = list(children)
        self.doc["nodes"].append(n)
        idx = len(self.doc["nodes"]) - 1
        if root:
            self.doc["scenes"][0]["nodes"].append(idx)
        return idx

    def material(self, **fields):
        self.doc["materials"].append(fields)
        return len(self.doc["materials"]) - 1

    def texture(self, png_bytes, wrap_s=None, wrap_t=None, with_sampler=True):
        self.doc.setdefault("images", []).append({"bufferView": self.view(png_bytes), "mimeType": "image/png"})
        tex = {"source": len(self.doc["images"]) - 1}
        if with_sampler:
            s = {}
            if wrap_s is not None:
                s["wrapS"] = wrap_s
            if wrap_t is not None:
                s["wrapT"] = wrap_t
            self.doc.setdefault("samplers", []).append(s)
            tex["sampler"] = len(self.doc["samplers"]) - 1
        self.doc.setdefault("textures", []).append(tex)
        return len(self.doc["textures"]) - 1

    # ---- container ------------------------------------------------------------------------
    def tobytes(self):
        while len(self.bin) % 4:
            self.bin.append(0)
        self.doc["buffers"][0]["byteLength"] = len(self.bin)
        js = json.dumps(self.doc, separators=(",", ":")).encode()
        js += b" " * (-len(js) % 4)
        total = 12 + 8 + len(js) + 8 + len(self.bin)
        out = io.BytesIO()
        out.write(struct.pack("<III", 0x46546C67, 2, total))
        out.write(struct.pack("<II", len(js), 0x4E4F534A))
        out.write(js)
        out.write(struct.pack("<II", len(self.bin), 0x004E4942))
        out.write(bytes(self.bin))
        return out.getvalue()

    def write(self, path):
        with open(path, "wb") as f:
            f.write(self.tobytes())
        return str(path)


def png_bytes(array, mode=None, **save_kw):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(np.asarray(array, np.uint8), mode).save(buf, format="PNG", **save_kw)
    return buf.getvalue()


def raw_png(samples, color_type, bit_depth, interlace=False, palette=None, trns=None, filter_type=0):
    """A PNG stream written from scratch: `samples` is (h, w, channels) of integers below
    2**bit_depth, rows top-down.  Covers what an imaging library will not write: low bit depths,
    16-bit colour, Adam7 interlacing, colour-key transparency, a chosen row filter (0..4)."""
    import struct
    import zlib
    samples = np.asarray(samples)
    h, w, ch = samples.shape

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))

    def pack_rows(sub):
        rows = []
        bpp = max(1, ch * bit_depth // 8)
        prev = None
        for r in sub:
            flat = r.reshape(-1)
            if bit_depth == 16:
                line = flat.astype(">u2").tobytes()
            elif bit_depth == 8:
                line = flat.astype(np.uint8).tobytes()
            else:
                bits = np.zeros(((len(flat) * bit_depth + 7) // 8) * 8, np.uint8)
                for k in range(bit_depth):
                    bits[np.arange(len(flat)) * bit_depth + k] = (flat >> (bit_depth - 1 - k)) & 1
                line = np.packbits(bits).tobytes()
            cur = np.frombuffer(line, np.uint8).astype(np.int32)
            up = prev if prev is not None else np.zeros_like(cur)
            left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
            if len(cur) <= bpp:
                left = np.zeros_like(cur)
            upleft = np.concatenate([np.zeros(bpp, np.int32), up[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
            if filter_type == 0:
                pred = np.zeros_like(cur)
            elif filter_type == 1:
                pred = left
            elif filter_type == 2:
                pred = up
            elif filter_type == 3:
                pred = (left + up) // 2
            else:
                pa, pb, pc = np.abs(up - upleft), np.abs(left - upleft), np.abs(left + up - 2 * upleft)
                pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, upleft))
            rows.append(bytes([filter_type]) + ((cur - pred) & 255).astype(np.uint8).tobytes())
            prev = cur
        return b"".join(rows)

    if interlace:
        lattice = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
        raw = b"".join(pack_rows(samples[y0::dy, x0::dx]) for x0, y0, dx, dy in lattice
                       if samples[y0::dy, x0::dx].size)
    else:
        raw = pack_rows(samples)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, color_type, 0, 0, 1 if interlace else 0))
    if palette is not None:
        out += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    half = len(raw) // 2 if len(raw) > 8 else len(raw)
    packed = zlib.compress(raw, 6)
    out += chunk(b"IDAT", packed[:len(packed) // 2]) + chunk(b"IDAT", packed[len(packed) // 2:])   # split across two chunks
    return out + chunk(b"IEND", b"")


def quat_matrix(q):
    """Rotation matrix of a unit quaternion given as glTF stores it, (x, y, z, w)."""
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def trs_matrix(translation=(0, 0, 0), rotation=(0, 0, 0, 1), scale=(1, 1, 1)):
    m = np.eye(4)
    m[:3, :3] = quat_matrix(rotation) @ np.diag(scale)
    m[:3, 3] = translation
    return m


def torus(nu, nv, R=0.35, r=0.12, center=(0, 0, 0)):
    """Indexed torus: (positions, normals, uvs, triangles)."""
    u = np.arange(nu) * (2 * np.pi / nu)
    v = np.arange(nv) * (2 * np.pi / nv)
    U, V = np.meshgrid(u, v, indexing="ij")
    n = np.stack([np.cos(V) * np.cos(U), np.sin(V), np.cos(V) * np.sin(U)], -1)
    p = np.stack([R * np.cos(U), np.zeros_like(U), R * np.sin(U)], -1) + r * n + np.asarray(center)
    uv = np.stack([U / (2 * np.pi), V / (2 * np.pi)], -1)
    tris = []
    for i in range(nu):
        for j in range(nv):
            a, b = i * nv + j, ((i + 1) % nu) * nv + j
            c, d = i * nv + (j + 1) % nv, ((i + 1) % nu) * nv + (j + 1) % nv
            tris += [(a, b, d), (a, d, c)]
    return p.reshape(-1, 3), n.reshape(-1, 3), uv.reshape(-1, 2), np.asarray(tris)
