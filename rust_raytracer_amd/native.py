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
EXPORTS = ["rt_scene_create", "rt_scene_set_camera", "rt_render", "rt_render_device", "rt_shard_bytes",
           "rt_assemble_shards", "rt_last_ray_counts", "rt_last_render_ms", "rt_scene_destroy",
           "rt_debug_set_queue_capacity", "rt_debug_last_spilled", "rt_last_error"]

_LIB = None


class RtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__("rt_hip error %d: %s" % (code, message))
        self.code = code


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing - build it with __graft_entry__.build() or make -C rust_raytracer_amd/csrc"
                               % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, u32 = C.c_void_p, C.c_uint32
        L.rt_scene_create.argtypes = [C.POINTER(abi.RtSceneDesc), C.c_int, C.POINTER(vp)]
        L.rt_scene_set_camera.argtypes = [vp, C.POINTER(C.c_float), u32, u32]
        L.rt_render.argtypes = [vp, C.POINTER(abi.RtRenderParams), vp, C.POINTER(abi.RtAux)]
        L.rt_render_device.argtypes = [vp, C.POINTER(abi.RtRenderParams), vp]
        L.rt_shard_bytes.argtypes = [vp, C.POINTER(abi.RtRenderParams), C.POINTER(C.c_uint64)]
        L.rt_assemble_shards.argtypes = [vp, C.POINTER(abi.RtRenderParams), vp, vp]
        L.rt_last_ray_counts.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.rt_last_render_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.rt_scene_destroy.argtypes = [vp]
        L.rt_scene_destroy.restype = None
        L.rt_debug_set_queue_capacity.argtypes = [vp, u32]
        L.rt_debug_last_spilled.argtypes = [vp, C.POINTER(u32)]
        L.rt_last_error.restype = C.c_char_p
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
        self.width, self.height = width, height

    @staticmethod
    def _params(max_depth=10, fmt=abi.RT_FORMAT_RGB8, shard_index=0, shard_count=1, stream=None):
        return abi.RtRenderParams(max_depth, fmt, shard_index, shard_count, stream)

    def render(self, max_depth: int = 10, channels: int = 3, aux: bool = False):
        """Whole frame to host memory: dict(rgb[H,W,channels], counters, and with aux prim_id / instance_id / flags)."""
        h, w = self.height, self.width
        rgb = np.zeros((h, w, channels), dtype=np.uint8)
        p = self._params(max_depth, abi.RT_FORMAT_RGBA8 if channels == 4 else abi.RT_FORMAT_RGB8)
        a = abi.RtAux()
        out = {"rgb": rgb}
        if aux:
            out["prim_id"] = np.zeros((h, w), dtype=np.uint32)
            out["instance_id"] = np.zeros((h, w), dtype=np.uint32)
            out["flags"] = np.zeros((h, w), dtype=np.uint8)
            a.prim_id = out["prim_id"].ctypes.data_as(C.POINTER(C.c_uint32))
            a.instance_id = out["instance_id"].ctypes.data_as(C.POINTER(C.c_uint32))
            a.flags = out["flags"].ctypes.data_as(C.POINTER(C.c_uint8))
        check(lib().rt_render(self._h, C.byref(p), rgb.ctypes.data_as(C.c_void_p), C.byref(a)))
        out["counters"] = np.array(a.counters, dtype=np.uint64)
        return out

    def shard_bytes(self, shard_count: int, channels: int = 3) -> int:
        p = self._params(10, abi.RT_FORMAT_RGBA8 if channels == 4 else abi.RT_FORMAT_RGB8, 0, shard_count)
        n = C.c_uint64()
        check(lib().rt_shard_bytes(self._h, C.byref(p), C.byref(n)))
        return int(n.value)

    def render_device(self, d_out: int, max_depth: int = 10, channels: int = 3, shard_index: int = 0,
                      shard_count: int = 1, stream: int | None = None):
        """Asynchronous render into device memory at address `d_out` (see rt_render_device)."""
        p = self._params(max_depth, abi.RT_FORMAT_RGBA8 if channels == 4 else abi.RT_FORMAT_RGB8, shard_index,
                         shard_count, stream)
        check(lib().rt_render_device(self._h, C.byref(p), C.c_void_p(d_out)))

    def assemble_shards(self, d_gathered: int, d_frame: int, shard_count: int, channels: int = 3,
                        stream: int | None = None):
        p = self._params(10, abi.RT_FORMAT_RGBA8 if channels == 4 else abi.RT_FORMAT_RGB8, 0, shard_count, stream)
        check(lib().rt_assemble_shards(self._h, C.byref(p), C.c_void_p(d_gathered), C.c_void_p(d_frame)))

    def last_ray_counts(self):
        c = (C.c_uint64 * abi.RT_N_COUNTERS)()
        check(lib().rt_last_ray_counts(self._h, c))
        return np.array(c, dtype=np.uint64)

    def last_render_ms(self) -> float:
        ms = C.c_float()
        check(lib().rt_last_render_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def debug_set_queue_capacity(self, capacity: int):
        check(lib().rt_debug_set_queue_capacity(self._h, capacity))

    def debug_last_spilled(self) -> int:
        n = C.c_uint32()
        check(lib().rt_debug_last_spilled(self._h, C.byref(n)))
        return int(n.value)
