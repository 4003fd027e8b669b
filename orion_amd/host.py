"""ctypes binding of the C++ host layer (``liborion_host.so``): .rtc parsing,
OBJ/MTL import and the SBVH build.  Produces plain numpy arrays in the layouts
``include/orion_hip.h`` takes.  No GPU code is touched here.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborion_host.so")

MEDIAN, MIDDLE, SAH = 0, 1, 2

# OrionBvhNode / orion_amd::BvhNode: lo3 hi3 is_leaf begin end child0 child1
NODE_DTYPE = np.dtype([("lo", "<f4", 3), ("hi", "<f4", 3), ("is_leaf", "<u4"),
                       ("begin", "<u4"), ("end", "<u4"), ("child", "<u4", 2)])
assert NODE_DTYPE.itemsize == 44


class _Rtc(C.Structure):
    _fields_ = [("recursion_level", C.c_int32), ("xres", C.c_int32), ("yres", C.c_int32),
                ("n_lights", C.c_int32), ("aspect_ratio", C.c_float),
                ("view_point", C.c_float * 3), ("look_at", C.c_float * 3), ("vector_up", C.c_float * 3),
                ("y_view", C.c_float), ("obj_file", C.c_char * 512), ("texture_file", C.c_char * 512)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(_LIB_PATH)
        L.orion_host_model_load.restype = C.c_void_p
        L.orion_host_model_load.argtypes = [C.c_char_p, C.c_int]
        L.orion_host_model_from_arrays.restype = C.c_void_p
        L.orion_host_model_from_arrays.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int]
        L.orion_host_model_free.argtypes = [C.c_void_p]
        for name in ("orion_host_model_mesh_count", "orion_host_model_material_count"):
            getattr(L, name).restype = C.c_uint32
            getattr(L, name).argtypes = [C.c_void_p]
        L.orion_host_mesh_sizes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        for name in ("orion_host_mesh_vertices", "orion_host_mesh_indices", "orion_host_mesh_nodes", "orion_host_mesh_tri_order"):
            getattr(L, name).restype = C.c_void_p
            getattr(L, name).argtypes = [C.c_void_p, C.c_uint32]
        L.orion_host_mesh_name.restype = C.c_char_p
        L.orion_host_mesh_name.argtypes = [C.c_void_p, C.c_uint32]
        L.orion_host_material_get.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orion_host_material_string.restype = C.c_char_p
        L.orion_host_material_string.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        L.orion_host_parse_rtc.argtypes = [C.c_char_p, C.POINTER(_Rtc), C.c_void_p, C.c_uint32]
        L.orion_host_write_rtc.argtypes = [C.c_char_p, C.POINTER(_Rtc), C.c_void_p]
        _lib = L
    return _lib


@dataclass
class Rtc:
    """Contents of an .rtc file (orion/rtc_parser.hpp:11-31)."""
    obj_file: str = ""
    texture_file: str = ""
    recursion_level: int = 0
    xres: int = 0
    yres: int = 0
    aspect_ratio: float = 0.0
    view_point: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    look_at: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    vector_up: np.ndarray = field(default_factory=lambda: np.array([0, 1, 0], np.float32))
    y_view: float = 1.0
    lights: np.ndarray = field(default_factory=lambda: np.zeros((0, 7), np.float32))  # pos3 col3 intensity


def parse_rtc(path: str) -> Rtc:
    raw = _Rtc()
    room = 64
    lights = np.zeros((room, 7), np.float32)
    lib().orion_host_parse_rtc(path.encode(), C.byref(raw), lights.ctypes.data, room)
    if raw.n_lights > room:          # any number of L lines (orion/rtc_parser.cpp:120-127): read again with room for all
        room = raw.n_lights
        lights = np.zeros((room, 7), np.float32)
        lib().orion_host_parse_rtc(path.encode(), C.byref(raw), lights.ctypes.data, room)
    return Rtc(obj_file=raw.obj_file.decode(), texture_file=raw.texture_file.decode(),
               recursion_level=raw.recursion_level, xres=raw.xres, yres=raw.yres,
               aspect_ratio=raw.aspect_ratio,
               view_point=np.array(raw.view_point[:], np.float32), look_at=np.array(raw.look_at[:], np.float32),
               vector_up=np.array(raw.vector_up[:], np.float32), y_view=raw.y_view,
               lights=lights[:max(0, min(room, raw.n_lights))].copy())


def write_rtc(path: str, rtc: Rtc) -> None:
    raw = _Rtc()
    raw.recursion_level, raw.xres, raw.yres = rtc.recursion_level, rtc.xres, rtc.yres
    raw.n_lights = len(rtc.lights)
    raw.aspect_ratio = rtc.aspect_ratio
    raw.view_point[:] = [float(x) for x in rtc.view_point]
    raw.look_at[:] = [float(x) for x in rtc.look_at]
    raw.vector_up[:] = [float(x) for x in rtc.vector_up]
    raw.y_view = rtc.y_view
    raw.obj_file = rtc.obj_file.encode()
    raw.texture_file = rtc.texture_file.encode()
    lights = np.ascontiguousarray(rtc.lights, np.float32).reshape(-1, 7)
    lib().orion_host_write_rtc(path.encode(), C.byref(raw), lights.ctypes.data)


@dataclass
class Texture:
    """Row-major texels; ``data`` is float32 [h,w,3] or uint8 [h,w,c], c in {1,3,4}."""
    data: np.ndarray

    @staticmethod
    def solid(rgb) -> "Texture":
        return Texture(np.asarray(rgb, np.float32).reshape(1, 1, 3).copy())

    def as_float_rgb(self) -> np.ndarray:
        d = self.data
        if d.dtype == np.uint8:
            f = d.astype(np.float32) / np.float32(255.0)
            if f.shape[2] == 1:
                f = np.repeat(f, 3, axis=2)
            return np.ascontiguousarray(f[:, :, :3])
        return np.ascontiguousarray(d, np.float32)


@dataclass
class Material:
    """Phong parameters + the four texture channels the tracer reads (orion/material.hpp:12-33)."""
    name: str = ""
    ambient: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    diffuse: np.ndarray = field(default_factory=lambda: np.full(3, 0.6, np.float32))
    specular: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    emissive: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    shininess: float = 0.0
    opacity: float = 1.0
    map_diffuse: str = ""
    map_specular: str = ""
    map_bump: str = ""
    map_ambient: str = ""
    # Optional image textures; None means "the solid colour" (Material::texturesFromSolidColor).
    tex_ambient: Optional[Texture] = None
    tex_diffuse: Optional[Texture] = None
    tex_specular: Optional[Texture] = None
    tex_emissive: Optional[Texture] = None

    def textures(self) -> List[Texture]:
        """ambient, diffuse, specular, emissive -- each an image or the 1x1 solid colour."""
        return [self.tex_ambient or Texture.solid(self.ambient),
                self.tex_diffuse or Texture.solid(self.diffuse),
                self.tex_specular or Texture.solid(self.specular),
                self.tex_emissive or Texture.solid(self.emissive)]

    def solid14(self) -> np.ndarray:
        return np.concatenate([self.ambient, self.diffuse, self.specular, self.emissive,
                               [self.shininess, self.opacity]]).astype(np.float32)


@dataclass
class Mesh:
    name: str
    vertices: np.ndarray   # [n,8] float32: position3 normal3 uv2
    indices: np.ndarray    # [3*t] uint32
    nodes: np.ndarray      # NODE_DTYPE
    tri_order: np.ndarray  # [t] uint32
    material: int

    @property
    def n_triangles(self) -> int:
        return len(self.indices) // 3


@dataclass
class Model:
    meshes: List[Mesh]
    materials: List[Material]
    directory: str = ""

    @property
    def n_triangles(self) -> int:
        return sum(m.n_triangles for m in self.meshes)


def _np_from(ptr, dtype, count):
    if count == 0 or not ptr:
        return np.zeros(0, dtype)
    buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


def _extract(h) -> Model:
    L = lib()
    meshes = []
    for i in range(L.orion_host_model_mesh_count(h)):
        sz = np.zeros(4, np.uint32)
        L.orion_host_mesh_sizes(h, i, sz.ctypes.data)
        nv, ni, nn, mat = (int(x) for x in sz)
        meshes.append(Mesh(
            name=(L.orion_host_mesh_name(h, i) or b"").decode(errors="replace"),
            vertices=_np_from(L.orion_host_mesh_vertices(h, i), np.float32, nv * 8).reshape(nv, 8),
            indices=_np_from(L.orion_host_mesh_indices(h, i), np.uint32, ni),
            nodes=_np_from(L.orion_host_mesh_nodes(h, i), NODE_DTYPE, nn),
            tri_order=_np_from(L.orion_host_mesh_tri_order(h, i), np.uint32, ni // 3),
            material=mat))
    mats = []
    for i in range(L.orion_host_model_material_count(h)):
        f = np.zeros(14, np.float32)
        L.orion_host_material_get(h, i, f.ctypes.data)
        s = [(L.orion_host_material_string(h, i, k) or b"").decode(errors="replace") for k in range(5)]
        mats.append(Material(name=s[0], ambient=f[0:3].copy(), diffuse=f[3:6].copy(), specular=f[6:9].copy(),
                             emissive=f[9:12].copy(), shininess=float(f[12]), opacity=float(f[13]),
                             map_diffuse=s[1], map_specular=s[2], map_bump=s[3], map_ambient=s[4]))
    directory = (L.orion_host_material_string(h, 0, 5) or b"").decode() if mats else ""
    return Model(meshes, mats, directory)


def load_model(obj_path: str, strategy: int = MEDIAN, load_textures: bool = False) -> Model:
    """Import an OBJ (+MTL) and build one SBVH per mesh (orion/model.cpp:19-193, orion/mesh.hpp:62-71).

    With ``load_textures`` the map_Kd / map_Ks images are decoded (Pillow) and installed the way
    the reference does it: the diffuse map is also used as the specular map whenever a specular
    map is named (orion/model.cpp:184-187).
    """
    L = lib()
    h = L.orion_host_model_load(obj_path.encode(), strategy)
    try:
        model = _extract(h)
    finally:
        L.orion_host_model_free(h)
    if load_textures:
        cache = {}
        for m in model.materials:
            if m.map_diffuse:
                p = os.path.join(model.directory, m.map_diffuse)
                if p not in cache:
                    try:
                        cache[p] = load_image(p)
                    except OSError as e:
                        # The reference prints and goes on with a texture that has no texels (orion/texture.hpp:35-38);
                        # fetching from it is undefined there.  Here the material keeps its solid colour.
                        print(f"orion_amd: cannot load texture {p}: {e}", file=sys.stderr)
                        cache[p] = None
                if cache[p] is None:
                    continue
                m.tex_diffuse = cache[p]
                if m.map_specular:
                    m.tex_specular = cache[p]
    return model


def load_image(path: str) -> Texture:
    """Decode an image to uint8 texels, keeping 1, 3 or 4 channels (orion/texture.hpp:33-70).  PNG and
    baseline JPEG go through the host layer's own readers (the reference's texels, bit for bit);
    anything they refuse, and every other format, goes through Pillow."""
    low = path.lower()
    try:
        if low.endswith(".png"):
            t = load_png(path)
            if t.data.shape[2] != 2:
                return t
        elif low.endswith((".jpg", ".jpeg")):
            return load_jpeg(path)
    except RuntimeError:
        pass
    from PIL import Image
    im = Image.open(path)
    if im.mode not in ("L", "RGB", "RGBA"):
        im = im.convert("RGBA" if "A" in im.mode else "RGB")
    a = np.asarray(im, np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return Texture(np.ascontiguousarray(a))


def _load_native(kind: int, path: str) -> Texture:
    L = lib()
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    buf = C.POINTER(C.c_uint8)()
    err = C.create_string_buffer(256)
    L.orion_host_load_image.argtypes = [C.c_int, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(C.POINTER(C.c_uint8)), C.c_char_p, C.c_int]
    L.orion_host_free.argtypes = [C.c_void_p]
    if L.orion_host_load_image(kind, path.encode(), C.byref(w), C.byref(h), C.byref(c), C.byref(buf), err, 256) != 0:
        raise RuntimeError(f"{path}: {err.value.decode()}")
    try:
        a = np.ctypeslib.as_array(buf, shape=(h.value, w.value, c.value)).copy()
    finally:
        L.orion_host_free(buf)
    return Texture(a)


def load_png(path: str) -> Texture:
    """Decode a PNG with the C++ host layer's own reader (what the orion_rt tool uses); raises on failure."""
    return _load_native(0, path)


def load_jpeg(path: str) -> Texture:
    """Decode a baseline JPEG with the C++ host layer's reader, which follows the arithmetic of the
    decoder the reference uses (texels identical to the reference's); raises on failure."""
    return _load_native(1, path)


def model_from_arrays(vertices: np.ndarray, indices: np.ndarray, strategy: int = MEDIAN,
                      material: Optional[Material] = None) -> Model:
    """One-mesh model from caller arrays (synthetic scenes)."""
    L = lib()
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 8)
    ix = np.ascontiguousarray(indices, np.uint32).reshape(-1)
    h = L.orion_host_model_from_arrays(v.ctypes.data, len(v), ix.ctypes.data, len(ix), strategy)
    try:
        model = _extract(h)
    finally:
        L.orion_host_model_free(h)
    if material is not None:
        model.materials[0] = material
    return model


def merge_models(models: List[Model]) -> Model:
    """Concatenate meshes of several models into one scene (mesh order preserved)."""
    meshes, mats = [], []
    for md in models:
        base = len(mats)
        mats.extend(md.materials)
        for m in md.meshes:
            meshes.append(Mesh(m.name, m.vertices, m.indices, m.nodes, m.tri_order, m.material + base))
    return Model(meshes, mats, models[0].directory if models else "")
