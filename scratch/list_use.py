"""Prints what the lists of the headline frame hold per depth (RT_DEBUG_LISTS) on the first-guess buffers and on the measured ones."""
import os, sys
os.environ["RT_DEBUG_LISTS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rust_raytracer_amd import native
W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from rust_raytracer_amd import meshes
models = meshes.ensure_model_dir("/tmp/list_use_models", levels={"bunny.obj": 4, "santa.obj": 4, "sphere_high.obj": 3, "sphere_low.obj": 1})
host = native.HostScene.load(os.path.join(root, "scenes", "reflective_bunnies.json"), models)
host.resize_screen(W, H)
m, w, h = host.camera()
dev = native.DeviceScene(host.desc(), 0)
dev.set_camera(m, w, h)
out = torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda")
for k in range(3):
    dev.render_device(out.data_ptr())
    print(k, dev.last_list_use(), flush=True)
