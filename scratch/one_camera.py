# One frame at drag camera k of bench.py (for rocprofv3 traces): python3 scratch/one_camera.py <k> [frames]
import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from rust_raytracer_amd import meshes, native
k = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
host = native.HostScene.load('/root/repo/scenes/reflective_bunnies.json', meshes.ensure_model_dir())
host.resize_screen(1920, 1080)
if k:
    step = 12.0 * ((k + 1) // 2)
    host.turn_camera(0.0, 0.0, step if k % 2 else -step, 0.0)
m, w, h = host.camera()
dev = native.DeviceScene(host.desc(), 0)
dev.set_camera(m, w, h)
for _ in range(n):
    r = dev.render(max_depth=10, aux=True)
c = r['counters']
print('camera', k, 'rays', int(c[0]), 'trace ms', dev.last_trace_ms(), 'counters', [int(x) for x in c])
