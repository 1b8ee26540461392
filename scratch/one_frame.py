import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
root = os.environ.get('GRAFT_REPO_ROOT', '.')
import numpy as np
from rust_raytracer_amd import meshes, native
md = meshes.ensure_model_dir()
host = native.HostScene.load(os.path.join(root, 'scenes/reflective_bunnies.json'), md)
host.resize_screen(1920, 1080)
dev = native.DeviceScene(host.desc(), 0)
m, w, h = host.camera(); dev.set_camera(m, w, h)
for _ in range(3):
    r = dev.render(max_depth=10)
print(dev.last_trace_ms())
