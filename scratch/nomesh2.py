import sys, os, json
sys.path.insert(0, '.')
import numpy as np
from rust_raytracer_amd import meshes, native
native.LIB_PATH = os.path.abspath('scratch/librt_hip_stats.so')
scene = json.load(open('scenes/reflective_bunnies.json'))
scene['meshes'] = []; scene['mesh_instances'] = []
host = native.HostScene.parse(json.dumps(scene))
host.resize_screen(1920, 1080)
dev = native.DeviceScene(host.desc(), 0)
m, w, h = host.camera(); dev.set_camera(m, w, h)
for d in (1, 2, 5, 10):
    for _ in range(3): r = dev.render(max_depth=d)
    print('max_depth', d, 'trace_ms %.3f frame_ms %.3f' % (dev.last_trace_ms(), dev.last_render_ms()))
