"""ctypes binding of librt_hip.so (include/rt_hip.h).  There is no CPU fallback: if the library is
missing or no MI355X is visible, the calls raise."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_hip.so")

# every function include/rt_hip.h declares
EXPORTS = ["rt_scene_create", "rt_scene_set_camera", "rt_scene_set_views", "rt_render", "rt_render_device", "rt_render_device_noalloc",
           "rt_noalloc_scratch_bytes", "rt_shard_bytes",
           "rt_assemble_shards", "rt_last_ray_counts", "rt_last_list_use", "rt_last_render_ms", "rt_last_trace_ms", "rt_set_kernel_timing",
           "rt_last_kernel_ms", "rt_scene_destroy",
           "rt_debug_set_queue_capacity", "rt_debug_last_regrown", "rt_last_error",
           "rt_multi_create", "rt_multi_set_camera", "rt_multi_set_views", "rt_multi_render", "rt_multi_render_views", "rt_multi_last_ray_counts",
           "rt_multi_last_render_ms", "rt_multi_destroy"]
# every function include/rt_host.h declares
HOST_EXPORTS = ["rt_host_scene_parse", "rt_host_scene_load", "rt_host_scene_desc", "rt_host_scene_camera",
                "rt_host_scene_resize_screen", "rt_host_scene_turn_camera", "rt_host_scene_set_camera_pose",
                "rt_host_scene_get_camera_pose", "rt_host_scene_find_material", "rt_host_scene_find_mesh",
                "rt_host_scene_add_instance", "rt_host_scene_add_mesh", "rt_host_scene_destroy", "rt_host_write_png"]

# RtMeshLoadFn of include/rt_host.h
RELEASE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t)
MESH_LOAD_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                           C.POINTER(RELEASE_FN))

_LIB = None


class RtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__("rt_hip error %d: %s" % (code, message))
        self.code = code


def _share_torch_hip_runtime():
    """PyTorch's ROCm wheel ships its own libamdhip64.so / libhsa-runtime64.so.  A process that first
    loads the system copies (through librt_hip.so) and later imports torch ends up with two ROCm
    runtimes, and the second one to initialise may find "no ROCm-capable device".  When torch is
    installed, its copy is therefore loaded first; librt_hip.so's dependency on libamdhip64.so.7 then
    resolves to that already loaded object, whichever of the two is imported first by the caller."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing - build it with __graft_entry__.build() or make -C rust_raytracer_amd/csrc"
                               % LIB_PATH)
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        vp, u32 = C.c_void_p, C.c_uint32
        L.rt_scene_create.argtypes = [C.POINTER(abi.RtSceneDesc), C.c_int, C.POINTER(vp)]
        L.rt_scene_set_camera.argtypes = [vp, C.POINTER(C.c_float), u32, u32]
        L.rt_scene_set_views.argtypes = [vp, C.POINTER(C.c_float), u32, u32, u32]
        L.rt_render.argtypes = [vp, C.POINTER(abi.RtRenderParams), vp, C.POINTER(abi.RtAux)]
        L.rt_render_device.argtypes = [vp, C.POINTER(abi.RtRenderParams), vp]
        L.rt_render_device_noalloc.argtypes = [vp, C.POINTER(abi.RtRenderParams), vp]
        L.rt_noalloc_scratch_bytes.argtypes = [C.POINTER(C.c_uint64)]
        L.rt_shard_bytes.argtypes = [vp, C.POINTER(abi.RtRenderParams), C.POINTER(C.c_uint64)]
        L.rt_assemble_shards.argtypes = [vp, C.POINTER(abi.RtRenderParams), vp, vp]
        L.rt_last_ray_counts.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.rt_last_list_use.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.rt_last_render_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.rt_last_trace_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.rt_set_kernel_timing.argtypes = [vp, C.c_int]
        L.rt_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        L.rt_scene_destroy.argtypes = [vp]
        L.rt_scene_destroy.restype = None
        L.rt_debug_set_queue_capacity.argtypes = [vp, u32]
        L.rt_debug_last_regrown.argtypes = [vp, C.POINTER(u32)]
        L.rt_last_error.restype = C.c_char_p
        L.rt_multi_create.argtypes = [C.POINTER(abi.RtSceneDesc), C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
        L.rt_multi_set_camera.argtypes = [vp, C.POINTER(C.c_float), u32, u32]
        L.rt_multi_set_views.argtypes = [vp, C.POINTER(C.c_float), u32, u32, u32]
        L.rt_multi_render.argtypes = [vp, C.POINTER(abi.RtRenderParams), vp]
        L.rt_multi_render_views.argtypes = [vp, C.POINTER(abi.RtRenderParams), C.POINTER(C.c_float), u32, u32, vp]
        L.rt_multi_last_ray_counts.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.rt_multi_last_render_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.rt_multi_destroy.argtypes = [vp]
        L.rt_multi_destroy.restype = None
        fp = C.POINTER(C.c_float)
        L.rt_host_scene_parse.argtypes = [C.c_char_p, C.c_size_t, MESH_LOAD_FN, vp, C.POINTER(vp)]
        L.rt_host_scene_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(vp)]
        L.rt_host_scene_desc.argtypes = [vp, C.POINTER(abi.RtSceneDesc)]
        L.rt_host_scene_camera.argtypes = [vp, fp, C.POINTER(u32), C.POINTER(u32)]
        L.rt_host_scene_resize_screen.argtypes = [vp, u32, u32]
        L.rt_host_scene_turn_camera.argtypes = [vp] + [C.c_float] * 4
        L.rt_host_scene_set_camera_pose.argtypes = [vp, fp, fp]
        L.rt_host_scene_get_camera_pose.argtypes = [vp, fp, fp]
        L.rt_host_scene_find_material.argtypes = [vp, C.c_char_p]
        L.rt_host_scene_find_mesh.argtypes = [vp, C.c_char_p]
        L.rt_host_scene_add_instance.argtypes = [vp, u32, fp, fp, fp, C.c_int32]
        L.rt_host_scene_add_mesh.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_size_t, u32, C.c_int]
        L.rt_host_scene_destroy.argtypes = [vp]
        L.rt_host_scene_destroy.restype = None
        _LIB = L
    return _LIB


def check(code: int):
    if code != abi.RT_OK:
        raise RtError(code, lib().rt_last_error().decode())


class DeviceScene:
    """Owner of one RtScene handle (a scene resident on one GPU)."""

    def __init__(self, desc: abi.RtSceneDesc, device_id: int = 0):
        self._h = C.c_void_p()
        check(lib().rt_scene_create(C.byref(desc), device_id, C.byref(self._h)))
        self.width = self.height = 0
        self.n_views = 1

    def close(self):
        if getattr(self, "_h", None):
            lib().rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_camera(self, screen_to_world, width: int, height: int):
        m = np.ascontiguousarray(np.asarray(screen_to_world, dtype=np.float32).reshape(16))
        check(lib().rt_scene_set_camera(self._h, m.ctypes.data_as(C.POINTER(C.c_float)), width, height))
        self.width, self.height, self.n_views = width, height, 1

    def set_views(self, screen_to_world, width: int, height: int):
        """Several cameras rendered by one call (rt_scene_set_views): `screen_to_world` is [n_views, 16]; outputs gain
        a leading view axis."""
        m = np.ascontiguousarray(np.asarray(screen_to_world, dtype=np.float32).reshape(-1, 16))
        check(lib().rt_scene_set_views(self._h, m.ctypes.data_as(C.POINTER(C.c_float)), m.shape[0], width, height))
        self.width, self.height, self.n_views = width, height, m.shape[0]

    @staticmethod
    def _params(max_depth=10, fmt=abi.RT_FORMAT_RGB8, shard_index=0, shard_count=1, stream=None, mode=abi.RT_MODE_RAYTRACE):
        return abi.RtRenderParams(max_depth, fmt, shard_index, shard_count, mode, 0, stream)

    def render(self, max_depth: int = 10, channels: int = 3, aux: bool = False, depth_map: bool = False):
        """Whole frame to host memory: dict(rgb[H,W,channels], counters, and with aux prim_id / instance_id / flags / hit_t);
        after set_views every array has a leading view axis."""
        h, w = self.height, self.width
        lead = (self.n_views,) if self.n_views > 1 else ()
        rgb = np.zeros(lead + (h, w, channels), dtype=np.uint8)
        p = self._params(max_depth, abi.RT_FORMAT_RGBA8 if channels == 4 else abi.RT_FORMAT_RGB8,
                         mode=abi.RT_MODE_DEPTH_MAP if depth_map else abi.RT_MODE_RAYTRACE)
        a = abi.RtAux()
        out = {"rgb": rgb}
        if aux:
            out["prim_id"] = np.zeros(lead + (h, w), dtype=np.uint32)
            out["instance_id"] = np.zeros(lead + (h, w), dtype=np.uint32)
            out["flags"] = np.zeros(lead + (h, w), dtype=np.uint8)
            out["hit_t"] = np.zeros(lead + (h, w), dtype=np.float32)
            a.hit_t = out["hit_t"].ctypes.data_as(C.POINTER(C.c_float))
            a.prim_id = out["prim_id"].ctypes.data_as(C.POINTER(C.c_uint32))
            a.instance_id = out["instance_id"].ctypes.data_as(C.POINTER(C.c_uint32))
            a.flags = out["flags"].ctypes.data_as(C.POINTER(C.c_uint8))
        check(lib().rt_render(self._h, C.byref(p), rgb.ctypes.data_as(C.c_void_p), C.byref(a)))
        out["counters"] = np.array(a.counters, dtype=np.uint64)
        return out

    def shard_bytes(self, shard_count: int, channels: int = 3) -> int:
        """Bytes of one view's shard (or frame when shard_count <= 1); a render writes n_views of them."""
        p = self._params(10, abi.RT_FORMAT_RGBA8 if channels == 4 else abi.RT_FORMAT_RGB8, 0, shard_count)
        n = C.c_uint64()
        check(lib().rt_shard_bytes(self._h, C.byref(p), C.byref(n)))
        return int(n.value)

    def render_device(self, d_out: int, max_depth: int = 10, channels: int = 3, shard_index: int = 0,
                      shard_count: int = 1, stream: int | None = None, noalloc: bool = False):
        """Asynchronous render into device memory at address `d_out` (see rt_render_device); `noalloc` selects
        rt_render_device_noalloc, whose work buffers live in the library's static scratch array."""
        p = self._params(max_depth, abi.RT_FORMAT_RGBA8 if channels == 4 else abi.RT_FORMAT_RGB8, shard_index,
                         shard_count, stream)
        fn = lib().rt_render_device_noalloc if noalloc else lib().rt_render_device
        check(fn(self._h, C.byref(p), C.c_void_p(d_out)))

    def assemble_shards(self, d_gathered: int, d_frame: int, shard_count: int, channels: int = 3,
                        stream: int | None = None):
        p = self._params(10, abi.RT_FORMAT_RGBA8 if channels == 4 else abi.RT_FORMAT_RGB8, 0, shard_count, stream)
        check(lib().rt_assemble_shards(self._h, C.byref(p), C.c_void_p(d_gathered), C.c_void_p(d_frame)))

    def last_ray_counts(self):
        c = (C.c_uint64 * abi.RT_N_COUNTERS)()
        check(lib().rt_last_ray_counts(self._h, c))
        return np.array(c, dtype=np.uint64)

    def last_list_use(self) -> dict:
        """rt_last_list_use: bytes held by the handle's per-frame buffers, entries per depth its lists can hold and entries
        the fullest depth of the last frame held (counting a frame also lets the handle size its next buffers by it)."""
        u = (C.c_uint64 * abi.RT_N_LIST_USE)()
        check(lib().rt_last_list_use(self._h, u))
        return {name: int(u[k]) for k, name in enumerate(abi.LIST_USE_NAMES)}

    def last_render_ms(self) -> float:
        ms = C.c_float()
        check(lib().rt_last_render_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def last_trace_ms(self) -> float:
        ms = C.c_float()
        check(lib().rt_last_trace_ms(self._h, C.byref(ms)))
        return float(ms.value)

    LEVEL_KERNELS = ("rt_setup", "rt_traverse", "rt_shade")

    def set_kernel_timing(self, enable: bool):
        check(lib().rt_set_kernel_timing(self._h, 1 if enable else 0))

    def last_kernel_ms(self) -> dict:
        """{kernel name: (summed ms, launches)} of the last frame rendered in kernel timing mode."""
        ms = (C.c_float * len(self.LEVEL_KERNELS))()
        n = (C.c_uint32 * len(self.LEVEL_KERNELS))()
        check(lib().rt_last_kernel_ms(self._h, ms, n))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(self.LEVEL_KERNELS)}

    def debug_set_queue_capacity(self, capacity: int):
        check(lib().rt_debug_set_queue_capacity(self._h, capacity))

    def debug_last_regrown(self) -> int:
        n = C.c_uint32()
        check(lib().rt_debug_last_regrown(self._h, C.byref(n)))
        return int(n.value)


class MultiDeviceScene:
    """Owner of one RtMulti: the scene replicated on several GPUs of this process, frames tile-sharded over them
    (rt_multi_* of include/rt_hip.h)."""

    def __init__(self, desc: abi.RtSceneDesc, device_ids):
        ids = (C.c_int * len(device_ids))(*device_ids)
        self._h = C.c_void_p()
        check(lib().rt_multi_create(C.byref(desc), ids, len(device_ids), C.byref(self._h)))
        self.width = self.height = 0
        self.n_views = 1

    def close(self):
        if getattr(self, "_h", None):
            lib().rt_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_views(self, screen_to_world, width: int, height: int):
        m = np.ascontiguousarray(np.asarray(screen_to_world, dtype=np.float32).reshape(-1, 16))
        check(lib().rt_multi_set_views(self._h, m.ctypes.data_as(C.POINTER(C.c_float)), m.shape[0], width, height))
        self.width, self.height, self.n_views = width, height, m.shape[0]

    def set_camera(self, screen_to_world, width: int, height: int):
        self.set_views(np.asarray(screen_to_world, dtype=np.float32).reshape(1, 16), width, height)

    def render(self, max_depth: int = 10, channels: int = 3):
        lead = (self.n_views,) if self.n_views > 1 else ()
        rgb = np.zeros(lead + (self.height, self.width, channels), dtype=np.uint8)
        p = abi.RtRenderParams(max_depth, abi.RT_FORMAT_RGBA8 if channels == 4 else abi.RT_FORMAT_RGB8, 0, 0, 0, 0, None)
        check(lib().rt_multi_render(self._h, C.byref(p), rgb.ctypes.data_as(C.c_void_p)))
        c = (C.c_uint64 * abi.RT_N_COUNTERS)()
        check(lib().rt_multi_last_ray_counts(self._h, c))
        ms = C.c_float()
        check(lib().rt_multi_last_render_ms(self._h, C.byref(ms)))
        return {"rgb": rgb, "counters": np.array(c, dtype=np.uint64), "ms": float(ms.value)}

    def render_views(self, screen_to_world, views_per_pass: int, max_depth: int = 10, channels: int = 3):
        """rt_multi_render_views: any number of cameras at the current screen size, `views_per_pass` at a time, the
        exchange of one pass overlapped with the tracing of the next.  Returns the frames [n, H, W, channels]."""
        m = np.ascontiguousarray(np.asarray(screen_to_world, dtype=np.float32).reshape(-1, 16))
        rgb = np.zeros((m.shape[0], self.height, self.width, channels), dtype=np.uint8)
        p = abi.RtRenderParams(max_depth, abi.RT_FORMAT_RGBA8 if channels == 4 else abi.RT_FORMAT_RGB8, 0, 0, 0, 0, None)
        check(lib().rt_multi_render_views(self._h, C.byref(p), m.ctypes.data_as(C.POINTER(C.c_float)), m.shape[0], views_per_pass,
                                          rgb.ctypes.data_as(C.c_void_p)))
        return rgb


def write_png(path: str, pixels: np.ndarray):
    """native_app's `canvas.write_png(path)` (write_png.rs:8-30) for an (H, W, 3 or 4) uint8 canvas."""
    h, w, c = pixels.shape
    data = np.ascontiguousarray(pixels, dtype=np.uint8)
    L = lib()
    L.rt_host_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    check(L.rt_host_write_png(path.encode(), data.ctypes.data, w, h, c))


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class HostScene:
    """Owner of one RtHostScene: a lib_raytracer::Scene built by the library's own loader."""

    def __init__(self, handle):
        self._h = handle
        self._keep = []

    @classmethod
    def load(cls, json_path: str, model_dir: str = ""):
        """Parser::parse_json with native_app's FilesystemMeshLoader."""
        h = C.c_void_p()
        check(lib().rt_host_scene_load(json_path.encode(), model_dir.encode(), C.byref(h)))
        return cls(h)

    @classmethod
    def parse(cls, json_text: str, mesh_loader=None):
        """Parser::parse_json with `mesh_loader(file_name) -> bytes | str | None` as the MeshLoader."""
        h = C.c_void_p()
        kept = []

        def load(_user, file_name, out_data, out_len, out_release):
            data = mesh_loader(file_name.decode()) if mesh_loader else None
            if data is None:
                return 1
            if isinstance(data, str):
                data = data.encode()
            buf = C.create_string_buffer(data, len(data))
            kept.append(buf)
            out_data[0] = C.cast(buf, C.c_void_p)
            out_len[0] = len(data)
            out_release[0] = RELEASE_FN(0)
            return 0

        text = json_text.encode()
        check(lib().rt_host_scene_parse(text, len(text), MESH_LOAD_FN(load), None, C.byref(h)))
        return cls(h)

    def close(self):
        if getattr(self, "_h", None):
            lib().rt_host_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def desc(self) -> abi.RtSceneDesc:
        d = abi.RtSceneDesc()
        check(lib().rt_host_scene_desc(self._h, C.byref(d)))
        return d

    def camera(self):
        m = (C.c_float * 16)()
        w, h = C.c_uint32(), C.c_uint32()
        check(lib().rt_host_scene_camera(self._h, m, C.byref(w), C.byref(h)))
        return np.array(m, dtype=np.float32), int(w.value), int(h.value)

    def resize_screen(self, width: int, height: int):
        check(lib().rt_host_scene_resize_screen(self._h, width, height))

    def turn_camera(self, begin_x, begin_y, end_x, end_y):
        check(lib().rt_host_scene_turn_camera(self._h, begin_x, begin_y, end_x, end_y))

    def set_camera_pose(self, position=None, orientation=None):
        check(lib().rt_host_scene_set_camera_pose(self._h, _f3(position) if position is not None else None,
                                                  _f3(orientation) if orientation is not None else None))

    def camera_pose(self):
        p, o = (C.c_float * 3)(), (C.c_float * 3)()
        check(lib().rt_host_scene_get_camera_pose(self._h, p, o))
        return np.array(p, dtype=np.float32), np.array(o, dtype=np.float32)

    def find_material(self, name: str) -> int:
        return lib().rt_host_scene_find_material(self._h, name.encode())

    def find_mesh(self, name: str) -> int:
        return lib().rt_host_scene_find_mesh(self._h, name.encode())

    def add_instance(self, mesh: int, position, orientation, scale, material_override: int = -1):
        check(lib().rt_host_scene_add_instance(self._h, mesh, _f3(position), _f3(orientation), _f3(scale),
                                               material_override))

    def add_mesh(self, name: str, obj_text, material: int, counter_clockwise: bool = True):
        data = obj_text.encode() if isinstance(obj_text, str) else obj_text
        check(lib().rt_host_scene_add_mesh(self._h, name.encode(), data, len(data), material, int(counter_clockwise)))
