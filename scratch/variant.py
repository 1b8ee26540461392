import sys, os
sys.path.insert(0, '.')
from rust_raytracer_amd import meshes, native
if len(sys.argv) > 1: native.LIB_PATH = os.path.abspath(sys.argv[1])
md = meshes.ensure_model_dir()
host = native.HostScene.load('scenes/reflective_bunnies.json', md)
host.resize_screen(1920, 1080)
dev = native.DeviceScene(host.desc(), 0)
m, w, h = host.camera(); dev.set_camera(m, w, h)
ts = []
dev.set_kernel_timing(True)
for _ in range(6):
    dev.render(max_depth=10)
    k = dev.last_kernel_ms()
    ts.append((dev.last_trace_ms(),) + tuple(k[n][0] for n in dev.LEVEL_KERNELS))
b = min(ts)
print(sys.argv[1:], 'trace %.3f' % b[0], ' '.join('%s %.3f' % (n[3:], v) for n, v in zip(dev.LEVEL_KERNELS, b[1:])))
