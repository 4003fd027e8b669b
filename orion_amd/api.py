"""ctypes binding of the HIP library (``liborion_hip.so``, C ABI in ``include/orion_hip.h``)
and a host-side mirror of the reference's render entry point.

Nothing here computes: ray queries and rendering run in the HIP kernels, and a
missing library is an error, never a fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import host

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORION_HIP_LIB selects another build of the same library (A/B measurements)
LIB_PATH = os.environ.get("ORION_HIP_LIB") or os.path.join(_HERE, "liborion_hip.so")

INVALID = 0xFFFFFFFF

# OrionHit
HIT_DTYPE = np.dtype([("mesh", "<u4"), ("tri", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])
assert HIT_DTYPE.itemsize == 20

TEX_RGB_F32, TEX_RGB_U8, TEX_RGBA_U8, TEX_GRAY_U8 = 0, 1, 2, 3

KERNEL_NAMES = ["closest_hit", "shadow", "shade", "generate", "resolve", "any_hit", "tail"]


class _Texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32), ("texels", C.c_void_p)]


class _Material(C.Structure):
    _fields_ = [("tex_ambient", C.c_uint32), ("tex_diffuse", C.c_uint32), ("tex_specular", C.c_uint32),
                ("tex_emissive", C.c_uint32), ("shininess", C.c_float), ("emissive_color", C.c_float * 3)]


class _Mesh(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("n_vertices", C.c_uint32), ("indices", C.c_void_p), ("n_indices", C.c_uint32),
                ("nodes", C.c_void_p), ("n_nodes", C.c_uint32), ("tri_order", C.c_void_p), ("material", C.c_uint32)]


class _Light(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("color", C.c_float * 3), ("intensity", C.c_float)]


class _Camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("front", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3)]


class _RenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp_begin", C.c_uint32), ("spp_end", C.c_uint32),
                ("light_samples", C.c_uint32), ("max_depth", C.c_uint32), ("seed", C.c_uint64),
                ("n_lights", C.c_uint32), ("lights", C.c_void_p),
                ("tile_x0", C.c_uint32), ("tile_y0", C.c_uint32), ("tile_x1", C.c_uint32), ("tile_y1", C.c_uint32)]


class _KernelStats(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("items", C.c_uint64), ("ms", C.c_double)]


class OrionError(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded HIP library.  Raises if it has not been built: there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OrionError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        try:
            # torch ships its own copy of the HIP runtime; importing it first makes this process use that
            # one copy for both (loading the system runtime first leaves torch without a device)
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.orion_hip_last_error.restype = C.c_char_p
        L.orion_hip_scene_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                             C.c_int, C.POINTER(C.c_void_p)]
        L.orion_hip_scene_destroy.argtypes = [C.c_void_p]
        L.orion_hip_scene_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.orion_hip_closest_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orion_hip_any_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orion_hip_closest_hit_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orion_hip_render.argtypes = [C.c_void_p, C.POINTER(_Camera), C.POINTER(_RenderParams), C.c_void_p]
        L.orion_hip_camera_vectors.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.POINTER(_Camera)]
        L.orion_hip_profile_enable.argtypes = [C.c_void_p, C.c_int]
        L.orion_hip_profile_read.argtypes = [C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _check(rc: int) -> None:
    if rc != 0:
        raise OrionError(f"orion_hip error {rc}: {lib().orion_hip_last_error().decode(errors='replace')}")


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _ptr(x) -> int:
    return x.data_ptr() if _is_torch(x) else x.ctypes.data


def camera_vectors(eye, look_at, up, y_view: float, aspect: float) -> np.ndarray:
    """RayTracer::calculateCameraVectors (orion/raytracer.cpp:212-238) -> camera12 = eye3 front3 right3 up3."""
    e = np.ascontiguousarray(eye, np.float32); l = np.ascontiguousarray(look_at, np.float32); u = np.ascontiguousarray(up, np.float32)
    cam = _Camera()
    _check(lib().orion_hip_camera_vectors(e.ctypes.data, l.ctypes.data, u.ctypes.data, y_view, aspect, C.byref(cam)))
    return np.array(list(cam.eye) + list(cam.front) + list(cam.right) + list(cam.up), np.float32)


def _array_ptr(x, what: str, shape, torch_dtype: str, np_dtype, device: Optional[int] = None) -> int:
    """Address of a caller's array after checking what the library takes on trust: element type, C-contiguity,
    shape (None = any extent) and, for torch tensors, the device.  Raises OrionError on a mismatch -- the
    kernels would otherwise read or write out of bounds."""
    if _is_torch(x):
        import torch
        ok = x.dtype == getattr(torch, torch_dtype) and x.is_contiguous() and x.is_cuda
        if ok and device is not None and device >= 0:
            ok = x.device.index == device
        nbytes_shape = tuple(x.shape)
    else:
        ok = isinstance(x, np.ndarray) and x.dtype == np_dtype and x.flags["C_CONTIGUOUS"]
        nbytes_shape = tuple(x.shape) if isinstance(x, np.ndarray) else ()
    if ok and shape is not None:
        ok = len(nbytes_shape) == len(shape) and all(s is None or s == t for s, t in zip(shape, nbytes_shape))
    if not ok:
        kind = "OrionHit" if getattr(np.dtype(np_dtype), "names", None) else np.dtype(np_dtype).name
        where = "" if device is None or device < 0 else f" on cuda:{device}"
        raise OrionError(f"{what}: expected a C-contiguous {kind} array of shape {shape}{where}, "
                         f"got {type(x).__name__} {getattr(x, 'dtype', '?')} {nbytes_shape}")
    return _ptr(x)


def _settle(*tensors) -> None:
    """The library runs on streams of its own and its calls return when their work is done, but it
    knows nothing of the torch streams that may still be writing (or reading) the device buffers
    handed to it: wait for the current torch stream before the call."""
    for t in tensors:
        if t is not None and _is_torch(t):
            import torch
            torch.cuda.current_stream(t.device).synchronize()
            return


def mesh_descs(model: host.Model):
    """The OrionMeshDesc array of a model and the arrays it points into (keep them alive)."""
    keep: List[np.ndarray] = []
    meshes = (_Mesh * max(1, len(model.meshes)))()
    for i, m in enumerate(model.meshes):
        v = np.ascontiguousarray(m.vertices, np.float32); ix = np.ascontiguousarray(m.indices, np.uint32)
        nd = np.ascontiguousarray(m.nodes); to = np.ascontiguousarray(m.tri_order, np.uint32)
        assert nd.dtype == host.NODE_DTYPE
        keep += [v, ix, nd, to]
        d = meshes[i]
        d.vertices, d.n_vertices, d.indices, d.n_indices = v.ctypes.data, len(v), ix.ctypes.data, len(ix)
        d.nodes, d.n_nodes, d.tri_order, d.material = nd.ctypes.data, len(nd), to.ctypes.data, m.material
    return meshes, keep


class Scene:
    """A model flattened and resident in HBM (orion_hip_scene_create)."""

    def __init__(self, model: host.Model, device: int = -1):
        L = lib()
        self.model = model
        if device < 0:
            try:
                import torch
                device = torch.cuda.current_device() if torch.cuda.is_available() else -1
            except ImportError:
                pass
        self.device = device
        keep: List[np.ndarray] = []
        # one texture set (ambient, diffuse, specular, emissive) per material
        tex_descs, mat_descs = [], (_Material * max(1, len(model.materials)))()
        for k, m in enumerate(model.materials):
            ids = []
            for t in m.textures():
                d = np.ascontiguousarray(t.data)
                if d.dtype == np.uint8:
                    fmt = {1: TEX_GRAY_U8, 3: TEX_RGB_U8, 4: TEX_RGBA_U8}[d.shape[2]]
                else:
                    d = np.ascontiguousarray(d, np.float32); fmt = TEX_RGB_F32
                keep.append(d)
                ids.append(len(tex_descs))
                tex_descs.append((d.shape[1], d.shape[0], fmt, d.ctypes.data))
            md = mat_descs[k]
            md.tex_ambient, md.tex_diffuse, md.tex_specular, md.tex_emissive = ids
            md.shininess = m.shininess
            md.emissive_color[:] = [float(x) for x in m.emissive]
        textures = (_Texture * max(1, len(tex_descs)))()
        for i, (w, h, f, p) in enumerate(tex_descs):
            textures[i].width, textures[i].height, textures[i].format, textures[i].texels = w, h, f, p
        meshes, mesh_keep = mesh_descs(model)
        keep += mesh_keep
        h = C.c_void_p()
        _check(L.orion_hip_scene_create(C.byref(meshes), len(model.meshes), C.byref(mat_descs), len(model.materials),
                                        C.byref(textures), len(tex_descs), device, C.byref(h)))
        self.h = h

    def close(self) -> None:
        if getattr(self, "h", None):
            lib().orion_hip_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name: str, value: int) -> None:
        _check(lib().orion_hip_scene_set_option(self.h, name.encode(), value))

    # -- ray queries ----------------------------------------------------------------
    def closest_hit(self, org, dir, out=None):
        """TracedModel::intersect for n rays.  numpy arrays (host) or torch CUDA tensors (device)."""
        if _is_torch(org):
            import torch
            n = org.shape[0]
            if out is None:
                out = torch.empty((n, 5), dtype=torch.int32, device=org.device)
            po = _array_ptr(org, "closest_hit: org", (n, 3), "float32", np.float32, self.device)
            pd = _array_ptr(dir, "closest_hit: dir", (n, 3), "float32", np.float32, self.device)
            ph = _array_ptr(out, "closest_hit: out", (n, 5), "int32", np.int32, self.device)
            _settle(org)
            _check(lib().orion_hip_closest_hit(self.h, po, pd, n, ph))
            return out
        o = np.ascontiguousarray(org, np.float32).reshape(-1, 3); d = np.ascontiguousarray(dir, np.float32).reshape(-1, 3)
        if o.shape != d.shape:
            raise OrionError(f"closest_hit: {o.shape[0]} origins, {d.shape[0]} directions")
        if out is None:
            out = np.zeros(len(o), HIT_DTYPE)
        ph = _array_ptr(out, "closest_hit: out", (len(o),), "int32", HIT_DTYPE)
        _check(lib().orion_hip_closest_hit(self.h, o.ctypes.data, d.ctypes.data, len(o), ph))
        return out

    def closest_hit_noalloc(self, org, dir, out):
        """orion_hip_closest_hit_noalloc: torch CUDA tensors only, the caller brings the output ([n, 5] int32)."""
        import torch
        n = org.shape[0]
        po = _array_ptr(org, "closest_hit_noalloc: org", (n, 3), "float32", np.float32, self.device)
        pd = _array_ptr(dir, "closest_hit_noalloc: dir", (n, 3), "float32", np.float32, self.device)
        ph = _array_ptr(out, "closest_hit_noalloc: out", (n, 5), "int32", np.int32, self.device)
        _settle(org)
        _check(lib().orion_hip_closest_hit_noalloc(self.h, po, pd, n, ph))
        return out

    def any_hit(self, org, dir, out=None):
        if _is_torch(org):
            import torch
            n = org.shape[0]
            if out is None:
                out = torch.empty(n, dtype=torch.uint8, device=org.device)
            po = _array_ptr(org, "any_hit: org", (n, 3), "float32", np.float32, self.device)
            pd = _array_ptr(dir, "any_hit: dir", (n, 3), "float32", np.float32, self.device)
            ph = _array_ptr(out, "any_hit: out", (n,), "uint8", np.uint8, self.device)
            _settle(org)
            _check(lib().orion_hip_any_hit(self.h, po, pd, n, ph))
            return out
        o = np.ascontiguousarray(org, np.float32).reshape(-1, 3); d = np.ascontiguousarray(dir, np.float32).reshape(-1, 3)
        if o.shape != d.shape:
            raise OrionError(f"any_hit: {o.shape[0]} origins, {d.shape[0]} directions")
        if out is None:
            out = np.zeros(len(o), np.uint8)
        ph = _array_ptr(out, "any_hit: out", (len(o),), "uint8", np.uint8)
        _check(lib().orion_hip_any_hit(self.h, o.ctypes.data, d.ctypes.data, len(o), ph))
        return out

    # -- rendering --------------------------------------------------------------------
    def render(self, camera12, width: int, height: int, spp_begin: int, spp_end: int, light_samples: int,
               max_depth: int, seed: int = 1, lights=None, tile: Optional[Tuple[int, int, int, int]] = None, accum=None):
        """Adds samples [spp_begin, spp_end) into ``accum`` ([H,W,4]: rgb sums + sample count).

        ``accum`` may be a numpy array or a torch CUDA tensor; a zeroed numpy array is made if omitted."""
        cam = _Camera()
        c = np.ascontiguousarray(camera12, np.float32).reshape(12)
        cam.eye[:], cam.front[:], cam.right[:], cam.up[:] = c[0:3].tolist(), c[3:6].tolist(), c[6:9].tolist(), c[9:12].tolist()
        lights = np.zeros((0, 7), np.float32) if lights is None else np.ascontiguousarray(lights, np.float32).reshape(-1, 7)
        p = _RenderParams()
        p.width, p.height, p.spp_begin, p.spp_end = width, height, spp_begin, spp_end
        p.light_samples, p.max_depth, p.seed = light_samples, max_depth, seed
        p.n_lights, p.lights = len(lights), lights.ctypes.data if len(lights) else None
        if tile is not None:
            p.tile_x0, p.tile_y0, p.tile_x1, p.tile_y1 = tile
        if accum is None:
            accum = np.zeros((height, width, 4), np.float32)
        pa = _array_ptr(accum, "render: accum", (height, width, 4), "float32", np.float32, self.device)
        _settle(accum)
        _check(lib().orion_hip_render(self.h, C.byref(cam), C.byref(p), pa))
        return accum

    # -- profiling ----------------------------------------------------------------------
    def profile(self, enable: bool) -> None:
        _check(lib().orion_hip_profile_enable(self.h, 1 if enable else 0))

    def profile_read(self) -> dict:
        st = (_KernelStats * len(KERNEL_NAMES))()
        _check(lib().orion_hip_profile_read(self.h, C.byref(st)))
        return {KERNEL_NAMES[i]: {"launches": int(st[i].launches), "items": int(st[i].items), "ms": float(st[i].ms)}
                for i in range(len(KERNEL_NAMES))}


@dataclass
class Frame:
    """What RayTracer::traceRTC derives from an .rtc before its pixel loop (orion/raytracer.cpp:26-64)."""
    rtc: host.Rtc
    model: host.Model
    camera12: np.ndarray


def load_frame(rtc_path: str, load_textures: bool = True, strategy: int = host.MEDIAN) -> Frame:
    """``strategy``: how the trees are split (host.MEDIAN, the reference's default; host.MIDDLE; host.SAH --
    SBVH::Strategy, orion/avx/sbvh.hpp:16-42)."""
    rtc = host.parse_rtc(rtc_path)
    obj = os.path.join(os.path.dirname(rtc_path), rtc.obj_file)
    model = host.load_model(obj, strategy=strategy, load_textures=load_textures)
    cam = camera_vectors(rtc.view_point, rtc.look_at, rtc.vector_up, rtc.y_view, rtc.aspect_ratio)
    return Frame(rtc, model, cam)


def trace_rtc(rtc_path: str, samples: int, light_samples: int, seed: int = 1, device: int = -1,
              spp_range: Optional[Tuple[int, int]] = None) -> np.ndarray:
    """RayTracer::traceRTC up to the image save (orion/raytracer.cpp:19-88): returns the [H,W,3] radiance
    image, already scaled by 1/samples.  ``spp_range`` restricts the call to a shard of the samples, in which
    case the un-normalised [H,W,4] accumulator is returned for the caller to reduce."""
    fr = load_frame(rtc_path)
    sc = Scene(fr.model, device)
    try:
        b, e = spp_range if spp_range is not None else (0, samples)
        acc = sc.render(fr.camera12, fr.rtc.xres, fr.rtc.yres, b, e, light_samples, max(0, fr.rtc.recursion_level),
                        seed=seed, lights=fr.rtc.lights)
    finally:
        sc.close()
    if spp_range is not None:
        return acc
    return acc[:, :, :3] * np.float32(1.0 / samples)
