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
prev = np.zeros(5, dtype=np.int64)
for d in range(1, 11):
    for _ in range(2): r = dev.render(max_depth=d)
    c = r['counters'][:5].astype(np.int64)
    print('depth', d-1, 'main', int(c[1]+c[3]+c[4]-prev[1]-prev[3]-prev[4]), 'shadow', int(c[2]-prev[2]), 'trace_ms %.3f frame_ms %.3f' % (dev.last_trace_ms(), dev.last_render_ms()))
    prev = c
