"""Frame-level renderer with the interface of the reference's GpuRenderer
(web_app/wasm/src/gpu_renderer.rs:44-68, 280, 356-364): construct it from a scene file, call
``render`` per frame, ``resize_screen`` / ``turn_camera`` in between.  The scene lives on one
MI355X; only the camera matrix crosses to the device per frame."""
from __future__ import annotations

import numpy as np

from . import meshes, native


class HipRenderer:
    def __init__(self, scene_path: str, model_dir: str | None = None, width: int | None = None,
                 height: int | None = None, device_id: int = 0, max_depth: int = 10):
        """GpuRenderer::new(width, height, asset_store, scene_file_name), gpu_renderer.rs:70-92: parse
        the scene, resize to (width, height) when given, upload it."""
        if model_dir is None:
            model_dir = meshes.ensure_model_dir()
        self.scene = native.HostScene.load(scene_path, model_dir)
        if width and height:
            self.scene.resize_screen(width, height)
        self.device = native.DeviceScene(self.scene.desc(), device_id)
        self.max_depth = max_depth
        self._push_camera()

    def _push_camera(self):
        m, w, h = self.scene.camera()
        self.device.set_camera(m, w, h)

    @property
    def size(self):
        return self.device.width, self.device.height

    def render(self, canvas_u8: np.ndarray | None = None, channels: int = 4) -> np.ndarray:
        """GpuRenderer::render(&mut self, canvas_u8), gpu_renderer.rs:280-354: fills an RGBA8 canvas,
        row 0 = top (channels=3 gives native_app's RGB8 canvas instead)."""
        out = self.device.render(max_depth=self.max_depth, channels=channels)["rgb"]
        if canvas_u8 is not None:
            np.copyto(canvas_u8.reshape(out.shape), out)
            return canvas_u8
        return out

    def resize_screen(self, width: int, height: int):
        """gpu_renderer.rs:356-359 -> Scene::resize_screen"""
        self.scene.resize_screen(width, height)
        self._push_camera()

    def turn_camera(self, drag_begin_x, drag_begin_y, drag_end_x, drag_end_y):
        """gpu_renderer.rs:361-364 -> Scene::turn_camera"""
        self.scene.turn_camera(drag_begin_x, drag_begin_y, drag_end_x, drag_end_y)
        self._push_camera()

    def close(self):
        self.device.close()
        self.scene.close()
