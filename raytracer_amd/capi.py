"""ctypes bindings of ``include/rt_hip.h`` (backend) and ``include/rt_host.h`` (host layer).

Everything here is plumbing: argument marshalling, error codes to exceptions,
numpy views of the serialised streams.  No arithmetic of the render path
happens in Python.
"""

import ctypes as C
import os

import numpy as np

_LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")

RT_SLOT_OBJECTS, RT_SLOT_MATERIALS, RT_SLOT_MODEL_INFO, RT_SLOT_MODELS, RT_SLOT_TEXTURES = 1, 2, 3, 4, 5
RT_TYPE_RGB, RT_TYPE_SPECTRAL = 0, 1
RT_OK, RT_ERR_INVALID, RT_ERR_STATE, RT_ERR_HIP, RT_ERR_NO_DEVICE, RT_ERR_SCENE, RT_ERR_NOMEM = 0, -1, -2, -3, -4, -5, -6
RT_OPT_SHADOW_EARLY_OUT, RT_OPT_REFILL_LANES, RT_OPT_INNER_SHARE, RT_OPT_MAX_PATHS, RT_OPT_TILE_MAJOR, RT_OPT_SPECTRAL_PDF_AS_WRITTEN = 1, 2, 3, 4, 5, 6
RT_OPT_WIDE_BVH = 7
RT_OPT_HERO_WAVELENGTH = 8
RT_OPT_NEE = 9
RT_OPT_STRATIFIED = 10
RT_OPT_PIPELINE = 11
RT_OPT_TEXTURE_U8 = 12
RT_OPT_FAST_SHADE = 13
RT_OPT_TREE_BUILD = 14
RT_TREE_BUILD_HOST, RT_TREE_BUILD_DEVICE = 0, 1
RT_INFO_TEXTURE_BYTES_UPLOADED, RT_INFO_TEXTURE_BYTES_DEVICE, RT_INFO_TEXTURES_RGBA8, RT_INFO_TEXTURES_FP32, RT_INFO_TREE_MODE = 1, 2, 3, 4, 5

# Every symbol include/rt_hip.h declares.
HIP_SYMBOLS = [
    "rt_create", "rt_destroy", "rt_upload", "rt_commit_scene", "rt_resize", "rt_set_frame", "rt_render", "rt_reserve",
    "rt_reset_accum", "rt_sync", "rt_readback", "rt_display_rgba8", "rt_display_rgba8_noalloc", "rt_display_rgba8_begin", "rt_display_rgba8_end", "rt_restore_accum", "rt_accum_device_ptr", "rt_bind_accum", "rt_add_frames",
    "rt_trace_primary", "rt_eval_colour", "rt_get_stats", "rt_reset_stats", "rt_set_counters", "rt_set_option", "rt_set_tile_shard", "rt_strerror", "rt_last_hip_error",
    "rt_comm_unique_id", "rt_comm_init", "rt_reduce", "rt_get_info", "rt_build_id",
]
# Every symbol include/rt_host.h declares.
HOST_SYMBOLS = [
    "rth_scene_build", "rth_scene_free", "rth_scene_counts", "rth_scene_sky", "rth_scene_stream",
    "rth_scene_model_stats", "rth_camera_make", "rth_engine_create", "rth_engine_render", "rth_engine_frame",
    "rth_engine_free", "rth_write_png", "rth_write_pfm",
]


def build_id():
    """Hash of the sources and flags librt_hip.so was built from."""
    return hip.rt_build_id().decode()


class RtCamera(C.Structure):
    _fields_ = [
        ("fov", C.c_float),
        ("resolution", C.c_float * 2),
        ("position", C.c_float * 3),
        ("forward", C.c_float * 3),
        ("right", C.c_float * 3),
        ("up", C.c_float * 3),
        ("bounces", C.c_int32),
        ("rayPerPixel", C.c_int32),
    ]


class RtStats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64),
        ("camera_samples", C.c_uint64),
        ("node_visits", C.c_uint64),
        ("tri_tests", C.c_uint64),
        ("shaded_hits", C.c_uint64),
        ("trace_ms", C.c_double),
        ("total_ms", C.c_double),
        ("trace_launches", C.c_uint64),
        ("max_stack", C.c_uint64),
        ("retraced", C.c_uint64),
        ("wave_steps", C.c_uint64),
        ("active_lane_steps", C.c_uint64),
        ("leaf_visits", C.c_uint64),
        ("inner_phase_steps", C.c_uint64),
        ("refills", C.c_uint64),
        ("shadow_rays", C.c_uint64),
    ]


class RtError(RuntimeError):
    pass


def _load(name):
    path = os.path.join(_LIB_DIR, name)
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `make` at the repository root (there is no CPU fallback)")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


hip = _load("librt_hip.so")
host = _load("librt_host.so")

_vp, _i, _u32, _sz, _f = C.c_void_p, C.c_int, C.c_uint32, C.c_size_t, C.c_float
_fp, _i32p = C.POINTER(C.c_float), C.POINTER(C.c_int32)

hip.rt_create.argtypes = [C.POINTER(_vp), _i]
hip.rt_destroy.argtypes = [_vp]
hip.rt_destroy.restype = None
hip.rt_upload.argtypes = [_vp, _i, _vp, _sz]
hip.rt_commit_scene.argtypes = [_vp, _i, _i, _i]
hip.rt_resize.argtypes = [_vp, _i, _i]
hip.rt_set_frame.argtypes = [_vp, C.POINTER(RtCamera), _fp, _i]
hip.rt_render.argtypes = [_vp, _u32, _u32, _u32]
hip.rt_reserve.argtypes = [_vp, _u32]
hip.rt_reset_accum.argtypes = [_vp]
hip.rt_sync.argtypes = [_vp]
hip.rt_readback.argtypes = [_vp, _vp, _sz, C.POINTER(_u32)]
hip.rt_display_rgba8.argtypes = [_vp, _vp, _sz]
hip.rt_display_rgba8_noalloc.argtypes = [_vp, _vp, _sz]
hip.rt_display_rgba8_begin.argtypes = [_vp, _i, _vp, _sz]
hip.rt_display_rgba8_end.argtypes = [_vp, _i]
hip.rt_restore_accum.argtypes = [_vp, _vp, _sz, _u32]
hip.rt_accum_device_ptr.argtypes = [_vp, C.POINTER(_vp), C.POINTER(_sz)]
hip.rt_bind_accum.argtypes = [_vp, _vp, _sz]
hip.rt_add_frames.argtypes = [_vp, _u32]
hip.rt_trace_primary.argtypes = [_vp, _u32, _vp, _vp, _vp, _vp, _vp]
hip.rt_eval_colour.argtypes = [_vp, _i, _vp, _sz, _vp]
hip.rt_get_stats.argtypes = [_vp, C.POINTER(RtStats)]
hip.rt_reset_stats.argtypes = [_vp]
hip.rt_set_counters.argtypes = [_vp, _i]
hip.rt_set_option.argtypes = [_vp, _i, _i]
hip.rt_set_tile_shard.argtypes = [_vp, _u32, _u32]
hip.rt_comm_unique_id.argtypes = [_vp, _sz]
hip.rt_comm_init.argtypes = [_vp, _i, _i, _vp, _sz]
hip.rt_reduce.argtypes = [_vp, _i]
hip.rt_get_info.argtypes = [_vp, _i, C.POINTER(C.c_uint64)]
hip.rt_build_id.argtypes = []
hip.rt_build_id.restype = C.c_char_p
hip.rt_strerror.argtypes = [_i]
hip.rt_strerror.restype = C.c_char_p
hip.rt_last_hip_error.argtypes = [_vp]
hip.rt_last_hip_error.restype = C.c_char_p

host.rth_scene_build.argtypes = [C.c_char_p, _i32p, _i]
host.rth_scene_build.restype = _vp
host.rth_scene_free.argtypes = [_vp]
host.rth_scene_free.restype = None
host.rth_scene_counts.argtypes = [_vp, _i32p, _i32p, _i32p]
host.rth_scene_sky.argtypes = [_vp, _fp]
host.rth_scene_stream.argtypes = [_vp, _i, C.POINTER(_fp), C.POINTER(_sz)]
host.rth_scene_model_stats.argtypes = [_vp, _i, _i32p]
host.rth_camera_make.argtypes = [C.POINTER(RtCamera), _fp, _f, _f, _f, _i, _i, _i, _i]
host.rth_engine_create.argtypes = [C.c_char_p, _i32p, _i, _fp, _f, _f, _f, _i, _i, _i, _i, _i, _i]
host.rth_engine_create.restype = _vp
host.rth_engine_render.argtypes = [_vp, _i]
host.rth_engine_frame.argtypes = [_vp, _vp, _sz]
host.rth_engine_free.argtypes = [_vp]
host.rth_engine_free.restype = None
host.rth_write_png.argtypes = [C.c_char_p, _vp, _i, _i]
host.rth_write_pfm.argtypes = [C.c_char_p, _vp, _i, _i, _f]


def comm_unique_id():
    buf = C.create_string_buffer(128)
    rc = hip.rt_comm_unique_id(buf, 128)
    if rc != 0:
        raise RtError(hip.rt_strerror(rc).decode())
    return buf.raw


def write_png(path, rgba8):
    """Writes an (H, W, 4) uint8 frame (row 0 = bottom row, as Backend.display_rgba8 returns it)."""
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 4 or host.rth_write_png(str(path).encode(), a.ctypes.data_as(_vp), a.shape[1], a.shape[0]) != 0:
        raise RtError(f"cannot write {path}")


def write_pfm(path, rgba, scale=1.0):
    """Writes the colour channels of an (H, W, 4) float32 frame (row 0 = bottom row) times `scale`."""
    a = np.ascontiguousarray(rgba, dtype=np.float32)
    if a.ndim != 3 or a.shape[2] != 4 or host.rth_write_pfm(str(path).encode(), a.ctypes.data_as(_vp), a.shape[1], a.shape[0], scale) != 0:
        raise RtError(f"cannot write {path}")


def _iparams(params):
    vals = [int(p) for p in params]
    return (C.c_int32 * len(vals))(*vals)


def make_camera(position=(0.0, 1.0, -2.2), yaw=90.0, pitch=0.0, fov=50.0, width=64, height=64, bounces=4, ray_per_pixel=1):
    """RayCamera + updateDirection() -> the uniform block (defaults: the reference's main.cpp camera)."""
    cam = RtCamera()
    pos = (C.c_float * 3)(*position)
    rc = host.rth_camera_make(C.byref(cam), pos, yaw, pitch, fov, width, height, bounces, ray_per_pixel)
    if rc != 0:
        raise RtError(hip.rt_strerror(rc).decode())
    return cam


class Scene:
    """A named scene built by the C++ host layer; exposes the five serialised streams."""

    def __init__(self, name, nu=0, nv=0, tex_size=0, seed=0, bvh=None):
        """bvh: None = process default (the reference's median split), "sah" = opt-in SAH trees, "reference"."""
        self.name = name
        self._h = None
        mode = {None: 0, "sah": 1, "reference": 2}[bvh]
        self._h = host.rth_scene_build(name.encode(), _iparams([nu, nv, tex_size, seed, mode]), 5)
        if not self._h:
            raise RtError(f"unknown scene {name!r}")
        oc, mc, lc = C.c_int32(), C.c_int32(), C.c_int32()
        host.rth_scene_counts(self._h, C.byref(oc), C.byref(mc), C.byref(lc))
        self.object_count, self.models_count, self.light_sources_count = oc.value, mc.value, lc.value
        sky = (C.c_float * 3)()
        host.rth_scene_sky(self._h, sky)
        self.sky = np.array(list(sky), dtype=np.float32)
        self.streams = {}
        for slot in range(1, 6):
            ptr, nbytes = _fp(), _sz()
            host.rth_scene_stream(self._h, slot, C.byref(ptr), C.byref(nbytes))
            n = nbytes.value // 4
            self.streams[slot] = np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.zeros(0, np.float32)

    def model_stats(self, model):
        out = (C.c_int32 * 10)()
        rc = host.rth_scene_model_stats(self._h, model, out)
        if rc != 0:
            raise RtError(hip.rt_strerror(rc).decode())
        keys = ["nodes", "triangles", "vertices", "light_triangles", "min_height", "max_height", "leaves", "min_leaf_tris", "max_leaf_tris", "valid"]
        return dict(zip(keys, list(out)))

    def close(self):
        if self._h:
            host.rth_scene_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


class Backend:
    """Thin object wrapper over an ``rt_ctx``; raises RtError on any non-zero status."""

    def __init__(self, device=0):
        self._ctx = _vp()
        self._check(hip.rt_create(C.byref(self._ctx), device), creating=True)
        self.width = self.height = 0

    def _check(self, rc, creating=False):
        if rc != 0:
            detail = "" if creating else hip.rt_last_hip_error(self._ctx).decode()
            raise RtError(f"{hip.rt_strerror(rc).decode()} ({rc}) {detail}")

    def upload_scene(self, scene):
        for slot in range(1, 6):
            w = np.ascontiguousarray(scene.streams[slot], dtype=np.float32)
            self._check(hip.rt_upload(self._ctx, slot, w.ctypes.data_as(_vp), w.nbytes))
        self._check(hip.rt_commit_scene(self._ctx, scene.object_count, scene.models_count, scene.light_sources_count))

    def resize(self, width, height):
        self._check(hip.rt_resize(self._ctx, width, height))
        self.width, self.height = width, height

    def set_frame(self, cam, sky, rtype=RT_TYPE_RGB):
        skyv = (C.c_float * 3)(*[float(s) for s in sky])
        self._check(hip.rt_set_frame(self._ctx, C.byref(cam), skyv, rtype))

    def render(self, first_frame=1, n_frames=1, stride=1):
        self._check(hip.rt_render(self._ctx, first_frame, n_frames, stride))

    def reserve(self, n_frames):
        """Allocates the path buffers a render call of n_frames needs, ahead of the call."""
        self._check(hip.rt_reserve(self._ctx, n_frames))

    def sync(self):
        self._check(hip.rt_sync(self._ctx))

    def reset_accum(self):
        self._check(hip.rt_reset_accum(self._ctx))

    def readback(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        frames = _u32()
        self._check(hip.rt_readback(self._ctx, out.ctypes.data_as(_vp), out.size, C.byref(frames)))
        return out, frames.value

    def display_rgba8(self):
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        self._check(hip.rt_display_rgba8(self._ctx, out.ctypes.data_as(_vp), out.size))
        return out

    def display_rgba8_noalloc(self):
        """The display pass through the library's fixed staging area (no device allocation in the call)."""
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        self._check(hip.rt_display_rgba8_noalloc(self._ctx, out.ctypes.data_as(_vp), out.size))
        return out

    def display_begin(self, slot, ptr, nbytes):
        """Queues the display pass and its copy to `ptr` (pinned host or device memory of H*W*4 bytes) and returns."""
        self._check(hip.rt_display_rgba8_begin(self._ctx, slot, _vp(ptr), nbytes))

    def display_end(self, slot):
        """Blocks until the buffer given to the slot's last display_begin holds the frame."""
        self._check(hip.rt_display_rgba8_end(self._ctx, slot))

    def restore_accum(self, rgba_sum, frames):
        a = np.ascontiguousarray(rgba_sum, dtype=np.float32)
        self._check(hip.rt_restore_accum(self._ctx, a.ctypes.data_as(_vp), a.size, frames))

    def accum_device_ptr(self):
        ptr, n = _vp(), _sz()
        self._check(hip.rt_accum_device_ptr(self._ctx, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def bind_accum(self, dev_ptr, floats):
        self._check(hip.rt_bind_accum(self._ctx, _vp(dev_ptr), floats))

    def add_frames(self, frames):
        self._check(hip.rt_add_frames(self._ctx, frames))

    def trace_primary(self, frame=1):
        n = self.width * self.height
        model = np.empty(n, np.int32)
        ident = np.empty(n, np.int32)
        t, u, v = (np.empty(n, np.float32) for _ in range(3))
        self._check(hip.rt_trace_primary(self._ctx, frame, *(a.ctypes.data_as(_vp) for a in (model, ident, t, u, v))))
        shape = (self.height, self.width)
        return model.reshape(shape), ident.reshape(shape), t.reshape(shape), u.reshape(shape), v.reshape(shape)

    def eval_colour(self, fn, values=None, items=1):
        """Parity hook: the device's colour functions/tables (rt_eval_colour) on rows of inputs."""
        width_out = (3, 1, 1, 3, 193)[fn]
        a = None
        if values is not None:
            a = np.ascontiguousarray(values, dtype=np.float32)
            items = a.shape[0] if a.ndim > 1 else a.size
        out = np.empty((items, width_out), np.float32)
        self._check(hip.rt_eval_colour(self._ctx, fn, a.ctypes.data_as(_vp) if a is not None else None, items, out.ctypes.data_as(_vp)))
        return out

    def stats(self):
        s = RtStats()
        self._check(hip.rt_get_stats(self._ctx, C.byref(s)))
        return {k: getattr(s, k) for k, _ in RtStats._fields_}

    def reset_stats(self):
        self._check(hip.rt_reset_stats(self._ctx))

    def set_tile_shard(self, offset, stride):
        self._check(hip.rt_set_tile_shard(self._ctx, offset, stride))

    def comm_init(self, rank, world, unique_id):
        """Joins an RCCL communicator of `world` contexts (collective); `unique_id` from comm_unique_id() on rank 0."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(hip.rt_comm_init(self._ctx, rank, world, buf, 128))

    def reduce(self, root=0):
        self._check(hip.rt_reduce(self._ctx, root))

    def set_option(self, option, value):
        self._check(hip.rt_set_option(self._ctx, option, int(value)))

    def info(self, what):
        value = C.c_uint64(0)
        self._check(hip.rt_get_info(self._ctx, what, C.byref(value)))
        return int(value.value)

    def set_counters(self, enabled):
        self._check(hip.rt_set_counters(self._ctx, 1 if enabled else 0))

    def close(self):
        if self._ctx:
            hip.rt_destroy(self._ctx)
            self._ctx = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
