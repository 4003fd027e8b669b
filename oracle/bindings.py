"""ctypes bindings for the oracle (liborion_oracle.so) and the reference harness
(_ref/liborion_ref.so).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(_HERE, "liborion_oracle.so")
REF_LIB = os.path.join(_HERE, "_ref", "liborion_ref.so")
REF_NATIVE_LIB = os.path.join(_HERE, "_ref", "liborion_ref_native.so")

INVALID = 0xFFFFFFFF


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class _MeshDesc(C.Structure):
    _fields_ = [("verts", C.c_void_p), ("n_verts", C.c_uint32), ("idx", C.c_void_p), ("n_idx", C.c_uint32),
                ("tex_rgb", C.c_void_p * 4), ("tex_w", C.c_int32 * 4), ("tex_h", C.c_int32 * 4),
                ("shininess", C.c_float), ("emissive_color", C.c_float * 3)]


class _RenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp_begin", C.c_uint32), ("spp_end", C.c_uint32),
                ("light_samples", C.c_uint32), ("max_depth", C.c_uint32), ("seed", C.c_uint64),
                ("n_lights", C.c_uint32), ("lights", C.c_void_p),
                ("tile_x0", C.c_uint32), ("tile_y0", C.c_uint32), ("tile_x1", C.c_uint32), ("tile_y1", C.c_uint32),
                ("rng_kind", C.c_int32), ("accum_mode", C.c_int32), ("rsqrt_mode", C.c_int32), ("threads", C.c_int32)]


_oracle = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        L = C.CDLL(ORACLE_LIB)
        L.oracle_scene_create.restype = C.c_void_p
        L.oracle_scene_create.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        L.oracle_scene_destroy.argtypes = [C.c_void_p]
        L.oracle_mesh_tree_size.restype = C.c_uint32
        L.oracle_mesh_tree_size.argtypes = [C.c_void_p, C.c_uint32]
        L.oracle_mesh_area.restype = C.c_float
        L.oracle_mesh_area.argtypes = [C.c_void_p, C.c_uint32]
        L.oracle_mesh_tree_dump.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_mesh_intersect.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32] + [C.c_void_p] * 4
        L.oracle_closest_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_any_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
        L.oracle_mesh_attributes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.oracle_normalize.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        L.oracle_mesh_random_points.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p]
        L.oracle_color_brdf.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
        L.oracle_color_phong.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
        L.oracle_material_fetch.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
        L.oracle_reflect.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.oracle_xoroshiro_u64.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_xoroshiro_floats.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_philox.argtypes = [C.c_uint32] * 4 + [C.c_uint64, C.c_void_p]
        L.oracle_u01.restype = C.c_float
        L.oracle_u01.argtypes = [C.c_uint32]
        L.oracle_camera_vectors.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_int]
        L.oracle_render.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_RenderParams), C.c_void_p, C.c_void_p]
        L.oracle_max_threads.restype = C.c_int
        L.wide_model_create.restype = C.c_void_p
        L.wide_model_create.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p]
        L.wide_model_destroy.argtypes = [C.c_void_p]
        L.wide_model_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.wide_model_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.wide_model_closest_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _oracle = L
    return _oracle


def _rays(org, dir):
    o = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(dir, np.float32).reshape(-1, 3)
    assert o.shape == d.shape
    return o, d


class OracleScene:
    """The CPU restatement over a ``orion_amd.host.Model`` (it builds its own BVHs)."""

    def __init__(self, model, strategy: int = 0):
        L = oracle_lib()
        self.model = model
        self._keep = []
        descs = (_MeshDesc * len(model.meshes))()
        for i, m in enumerate(model.meshes):
            v = np.ascontiguousarray(m.vertices, np.float32)
            ix = np.ascontiguousarray(m.indices, np.uint32)
            mat = model.materials[m.material]
            tex = [t.as_float_rgb() for t in mat.textures()]
            self._keep += [v, ix] + tex
            d = descs[i]
            d.verts, d.n_verts, d.idx, d.n_idx = v.ctypes.data, len(v), ix.ctypes.data, len(ix)
            for k in range(4):
                d.tex_rgb[k] = tex[k].ctypes.data
                d.tex_w[k] = tex[k].shape[1]
                d.tex_h[k] = tex[k].shape[0]
            d.shininess = mat.shininess
            d.emissive_color[:] = [float(x) for x in mat.emissive]
        self.h = L.oracle_scene_create(C.byref(descs), len(model.meshes), strategy)
        self.n_meshes = len(model.meshes)

    def __del__(self):
        if getattr(self, "h", None):
            oracle_lib().oracle_scene_destroy(self.h)
            self.h = None

    def tree(self, mesh: int):
        L = oracle_lib()
        n = L.oracle_mesh_tree_size(self.h, mesh)
        nt = self.model.meshes[mesh].n_triangles
        boxes = np.zeros((n, 6), np.float32); meta = np.zeros((n, 3), np.uint32); perm = np.zeros(nt, np.uint32)
        L.oracle_mesh_tree_dump(self.h, mesh, _p(boxes), _p(meta), _p(perm))
        return boxes, meta, perm

    def area(self, mesh: int) -> float:
        return oracle_lib().oracle_mesh_area(self.h, mesh)

    def mesh_intersect(self, mesh, org, dir, t_in=None):
        o, d = _rays(org, dir)
        n = len(o)
        t = np.full(n, np.inf, np.float32) if t_in is None else np.ascontiguousarray(t_in, np.float32).copy()
        u = np.zeros(n, np.float32); v = np.zeros(n, np.float32); tri = np.zeros(n, np.uint32)
        oracle_lib().oracle_mesh_intersect(self.h, mesh, _p(o), _p(d), n, _p(t), _p(u), _p(v), _p(tri))
        return tri, t, u, v

    def closest_hit(self, org, dir, threads: int = 1, count: bool = False):
        o, d = _rays(org, dir)
        n = len(o)
        ids = np.zeros((n, 2), np.uint32); tuv = np.zeros((n, 3), np.float32)
        counters = np.zeros(2, np.uint64)
        oracle_lib().oracle_closest_hit(self.h, _p(o), _p(d), n, _p(ids), _p(tuv), _p(counters) if count else None, threads)
        return (ids, tuv, counters) if count else (ids, tuv)

    def any_hit(self, org, dir, threads: int = 1):
        o, d = _rays(org, dir)
        occ = np.zeros(len(o), np.uint8)
        oracle_lib().oracle_any_hit(self.h, _p(o), _p(d), len(o), _p(occ), threads)
        return occ

    def attributes(self, mesh, tri, u, v):
        tri = np.ascontiguousarray(tri, np.uint32); u = np.ascontiguousarray(u, np.float32); v = np.ascontiguousarray(v, np.float32)
        out = np.zeros((len(tri), 5), np.float32)
        oracle_lib().oracle_mesh_attributes(self.h, mesh, _p(tri), _p(u), _p(v), len(tri), _p(out))
        return out

    def random_points(self, mesh, seed, n):
        out = np.zeros((n, 4), np.float32)
        oracle_lib().oracle_mesh_random_points(self.h, mesh, seed, n, _p(out))
        return out

    def color_brdf(self, mesh, rows18, rsqrt_mode=0):
        a = np.ascontiguousarray(rows18, np.float32).reshape(-1, 18)
        out = np.zeros((len(a), 3), np.float32)
        oracle_lib().oracle_color_brdf(self.h, mesh, _p(a), len(a), _p(out), rsqrt_mode)
        return out

    def color_phong(self, mesh, rows18, rsqrt_mode=0):
        a = np.ascontiguousarray(rows18, np.float32).reshape(-1, 18)
        out = np.zeros((len(a), 3), np.float32)
        oracle_lib().oracle_color_phong(self.h, mesh, _p(a), len(a), _p(out), rsqrt_mode)
        return out

    def material_fetch(self, mesh, which, uv):
        a = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        out = np.zeros((len(a), 3), np.float32)
        oracle_lib().oracle_material_fetch(self.h, mesh, which, _p(a), len(a), _p(out))
        return out

    def render(self, camera12, width, height, spp_begin, spp_end, light_samples, max_depth, seed=1,
               lights=None, tile=None, rng_kind=1, accum_mode=1, rsqrt_mode=0, threads=0,
               accum: Optional[np.ndarray] = None, count: bool = False):
        """camera12 = eye3, front3, right3, up3.  Returns accum [H,W,4] (rgb sums + sample count)."""
        L = oracle_lib()
        cam = np.ascontiguousarray(camera12, np.float32).reshape(12)
        lights = np.zeros((0, 7), np.float32) if lights is None else np.ascontiguousarray(lights, np.float32).reshape(-1, 7)
        p = _RenderParams()
        p.width, p.height, p.spp_begin, p.spp_end = width, height, spp_begin, spp_end
        p.light_samples, p.max_depth, p.seed = light_samples, max_depth, seed
        p.n_lights, p.lights = len(lights), lights.ctypes.data
        x0, y0, x1, y1 = tile if tile is not None else (0, 0, width, height)
        p.tile_x0, p.tile_y0, p.tile_x1, p.tile_y1 = x0, y0, x1, y1
        p.rng_kind, p.accum_mode, p.rsqrt_mode = rng_kind, accum_mode, rsqrt_mode
        p.threads = threads if threads > 0 else L.oracle_max_threads()
        if accum is None:
            accum = np.zeros((height, width, 4), np.float32)
        counters = np.zeros(2, np.uint64)
        L.oracle_render(self.h, _p(cam), C.byref(p), _p(accum), _p(counters) if count else None)
        return (accum, counters) if count else accum


class WideModel:
    """CPU model of the product's wide-tree walk over a ``host.Model`` (trees from the product's host builder)."""

    STATS = ["nodes", "own_fail", "leaves", "tris", "stale", "candidates", "max_stack", "max_excess_pads", "not_taken"]
    NOT_TAKEN = 0xFFFFFFFE

    def __init__(self, model):
        from orion_amd import api
        L = oracle_lib()
        descs, self._keep = api.mesh_descs(model)
        why = C.create_string_buffer(256)
        self.h = L.wide_model_create(C.byref(descs), len(model.meshes), why)
        self.why_not = why.value.decode()
        if self.h:
            info = np.zeros(3, np.uint32); finfo = np.zeros(2, np.float32)
            L.wide_model_info(self.h, _p(info), _p(finfo))
            self.n_nodes, self.stack_need, self.n_virtual = int(info[0]), int(info[1]), int(info[2])
            self.pad, self.org_bound = float(finfo[0]), float(finfo[1])

    def __del__(self):
        if getattr(self, "h", None):
            oracle_lib().wide_model_destroy(self.h)
            self.h = None

    def nodes(self):
        """(n_nodes x 16 words as uploaded, n_nodes x 4 triangle counts below each slot)."""
        words = np.zeros((self.n_nodes, 16), np.uint32); below = np.zeros((self.n_nodes, 4), np.uint32)
        oracle_lib().wide_model_nodes(C.c_void_p(self.h), _p(words), _p(below))
        return words, below

    def closest_hit(self, org, dir, cull=True, threads=1, kappa=0.0):
        o, d = _rays(org, dir)
        n = len(o)
        ids = np.zeros((n, 2), np.uint32); tuv = np.zeros((n, 3), np.float32); st = np.zeros(108, np.float64)
        oracle_lib().wide_model_closest_hit(self.h, _p(o), _p(d), n, int(cull), kappa, _p(ids), _p(tuv), _p(st), threads)
        out = dict(zip(self.STATS, st[:9].tolist()))
        out["depth_hist"] = st[9:41].copy(); out["steps_hist"] = st[41:105].copy()
        out["danger_visits"], out["danger_slots"], out["slots_seen"] = st[105:108].tolist()
        return ids, tuv, out


def normalize(xyz, rsqrt_mode=0):
    a = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3).copy()
    oracle_lib().oracle_normalize(_p(a), len(a), rsqrt_mode)
    return a


def reflect(rows6):
    a = np.ascontiguousarray(rows6, np.float32).reshape(-1, 6)
    out = np.zeros((len(a), 3), np.float32)
    oracle_lib().oracle_reflect(_p(a), len(a), _p(out))
    return out


def xoroshiro_u64(seed, n, jumps=0):
    out = np.zeros(n, np.uint64)
    oracle_lib().oracle_xoroshiro_u64(seed, jumps, n, _p(out))
    return out


def xoroshiro_floats(seed, n, jumps=0):
    out = np.zeros(n, np.float32)
    oracle_lib().oracle_xoroshiro_floats(seed, jumps, n, _p(out))
    return out


def philox(c0, c1, c2, c3, seed):
    out = np.zeros(4, np.uint32)
    oracle_lib().oracle_philox(c0, c1, c2, c3, seed, _p(out))
    return out


def u01(word: int) -> float:
    return float(oracle_lib().oracle_u01(word))


def camera_vectors(eye, look_at, up, y_view, aspect, rsqrt_mode=0):
    """Returns camera12 = eye3, front3, right3, up3 (the layout render() takes)."""
    e = np.ascontiguousarray(eye, np.float32); l = np.ascontiguousarray(look_at, np.float32); u = np.ascontiguousarray(up, np.float32)
    out = np.zeros(9, np.float32)
    oracle_lib().oracle_camera_vectors(_p(e), _p(l), _p(u), y_view, aspect, _p(out), rsqrt_mode)
    return np.concatenate([e, out[0:3], out[6:9], out[3:6]]).astype(np.float32)


# ---------------------------------------------------------------------------
# Reference harness (development container only)
# ---------------------------------------------------------------------------
def have_ref(path: str = REF_LIB) -> bool:
    return os.path.exists(path)


_refs = {}


def ref_lib(path: str = REF_LIB):
    if path not in _refs:
        L = C.CDLL(path)
        L.ref_mesh_create.restype = C.c_void_p
        L.ref_mesh_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
        L.ref_mesh_destroy.argtypes = [C.c_void_p]
        L.ref_mesh_set_texture.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
        L.ref_mesh_id.restype = C.c_uint32
        L.ref_mesh_id.argtypes = [C.c_void_p]
        L.ref_mesh_area.restype = C.c_float
        L.ref_mesh_area.argtypes = [C.c_void_p]
        L.ref_mesh_tree_size.restype = C.c_uint32
        L.ref_mesh_tree_size.argtypes = [C.c_void_p]
        L.ref_mesh_tree_dump.argtypes = [C.c_void_p] * 4
        L.ref_mesh_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32] + [C.c_void_p] * 4
        L.ref_model_intersect.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
        L.ref_mesh_attributes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.ref_normalize.argtypes = [C.c_void_p, C.c_uint32]
        L.ref_mesh_random_points.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
        L.ref_color_brdf.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.ref_color_phong.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.ref_material_fetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
        L.ref_reflect.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.ref_xoroshiro_u64.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
        L.ref_xoroshiro_floats.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
        L.ref_parse_rtc.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32]
        L.ref_camera_vectors.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.ref_max_threads.restype = C.c_int
        _refs[path] = L
    return _refs[path]


class RefScene:
    """The reference's own TracedMesh objects, one per mesh of a ``host.Model``."""

    def __init__(self, model, strategy: int = 0, lib_path: str = REF_LIB, texture_files: Optional[dict] = None):
        self.L = ref_lib(lib_path)
        self.model = model
        self.meshes: List[int] = []
        for i, m in enumerate(model.meshes):
            v = np.ascontiguousarray(m.vertices, np.float32)
            ix = np.ascontiguousarray(m.indices, np.uint32)
            s = model.materials[m.material].solid14()
            h = self.L.ref_mesh_create(_p(v), len(v), _p(ix), len(ix), _p(s), strategy)
            self.meshes.append(h)
            for (mi, ttype), path in (texture_files or {}).items():
                if mi == i:
                    assert self.L.ref_mesh_set_texture(h, ttype, path.encode()) == 0
        self._arr = (C.c_void_p * len(self.meshes))(*self.meshes)

    def __del__(self):
        for h in getattr(self, "meshes", []):
            self.L.ref_mesh_destroy(h)
        self.meshes = []

    def tree(self, mesh: int):
        h = self.meshes[mesh]
        n = self.L.ref_mesh_tree_size(h)
        nt = self.model.meshes[mesh].n_triangles
        boxes = np.zeros((n, 6), np.float32); meta = np.zeros((n, 3), np.uint32); perm = np.zeros(nt, np.uint32)
        self.L.ref_mesh_tree_dump(h, _p(boxes), _p(meta), _p(perm))
        return boxes, meta, perm

    def area(self, mesh: int) -> float:
        return self.L.ref_mesh_area(self.meshes[mesh])

    def mesh_intersect(self, mesh, org, dir, t_in=None):
        o, d = _rays(org, dir)
        n = len(o)
        t = np.full(n, np.inf, np.float32) if t_in is None else np.ascontiguousarray(t_in, np.float32).copy()
        u = np.zeros(n, np.float32); v = np.zeros(n, np.float32); tri = np.zeros(n, np.uint32)
        self.L.ref_mesh_intersect(self.meshes[mesh], _p(o), _p(d), n, _p(t), _p(u), _p(v), _p(tri))
        return tri, t, u, v

    def closest_hit(self, org, dir, threads: int = 1):
        o, d = _rays(org, dir)
        n = len(o)
        ids = np.zeros((n, 2), np.uint32); tuv = np.zeros((n, 3), np.float32)
        self.L.ref_model_intersect(self._arr, len(self.meshes), _p(o), _p(d), n, _p(ids), _p(tuv), threads)
        return ids, tuv

    def attributes(self, mesh, tri, u, v):
        tri = np.ascontiguousarray(tri, np.uint32); u = np.ascontiguousarray(u, np.float32); v = np.ascontiguousarray(v, np.float32)
        out = np.zeros((len(tri), 5), np.float32)
        self.L.ref_mesh_attributes(self.meshes[mesh], _p(tri), _p(u), _p(v), len(tri), _p(out))
        return out

    def random_points(self, mesh, seed, n):
        out = np.zeros((n, 4), np.float32)
        self.L.ref_mesh_random_points(self.meshes[mesh], seed, n, _p(out))
        return out

    def color_brdf(self, mesh, rows18):
        a = np.ascontiguousarray(rows18, np.float32).reshape(-1, 18)
        out = np.zeros((len(a), 3), np.float32)
        self.L.ref_color_brdf(self.meshes[mesh], _p(a), len(a), _p(out))
        return out

    def color_phong(self, mesh, rows18):
        a = np.ascontiguousarray(rows18, np.float32).reshape(-1, 18)
        out = np.zeros((len(a), 3), np.float32)
        self.L.ref_color_phong(self.meshes[mesh], _p(a), len(a), _p(out))
        return out

    def material_fetch(self, mesh, which, uv):
        a = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        out = np.zeros((len(a), 3), np.float32)
        self.L.ref_material_fetch(self.meshes[mesh], which, _p(a), len(a), _p(out))
        return out


def ref_normalize(xyz, lib_path: str = REF_LIB):
    a = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3).copy()
    ref_lib(lib_path).ref_normalize(_p(a), len(a))
    return a


def ref_reflect(rows6):
    a = np.ascontiguousarray(rows6, np.float32).reshape(-1, 6)
    out = np.zeros((len(a), 3), np.float32)
    ref_lib().ref_reflect(_p(a), len(a), _p(out))
    return out


def ref_xoroshiro_u64(seed, n, jumps=0):
    out = np.zeros(n, np.uint64)
    ref_lib().ref_xoroshiro_u64(seed, jumps, n, _p(out))
    return out


def ref_xoroshiro_floats(seed, n, jumps=0):
    out = np.zeros(n, np.float32)
    ref_lib().ref_xoroshiro_floats(seed, jumps, n, _p(out))
    return out


def ref_parse_rtc(path: str):
    ints = np.zeros(4, np.int32); floats = np.zeros(11, np.float32); lights = np.zeros((64, 7), np.float32)
    buf = C.create_string_buffer(512)
    ref_lib().ref_parse_rtc(path.encode(), _p(ints), _p(floats), _p(lights), 64, buf, 512)
    return ints, floats, lights[:ints[3]].copy(), buf.value.decode()


def ref_camera_vectors(eye, look_at, up, y_view, aspect):
    e = np.ascontiguousarray(eye, np.float32); l = np.ascontiguousarray(look_at, np.float32); u = np.ascontiguousarray(up, np.float32)
    out = np.zeros(9, np.float32)
    ref_lib().ref_camera_vectors(_p(e), _p(l), _p(u), y_view, aspect, _p(out))
    return np.concatenate([e, out[0:3], out[6:9], out[3:6]]).astype(np.float32)
