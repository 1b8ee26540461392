"""What the size of the scene's walk data costs the BVH walk: the 8x8 (or 16x16) instance grid at 4K with its instances sharing
1, 2, 4 ... meshes of the same size.  Same rays, same instance boxes, same tree shapes up to the meshes' jitter - only the bytes
behind them grow.  Usage: python scratch/grid_locality.py [grid] [level]"""
import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from rust_raytracer_amd import native, synthetic
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 8
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for meshes in (1, 4, 32) if grid == 8 else (1, 16, 128):
    d = '/tmp/grid_locality_%d' % meshes
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, 'grid.json')
    open(path, 'w').write(synthetic.instance_grid_scene(d, grid, meshes, level, 3840, 2160))
    host = native.HostScene.load(path, d)
    dev = native.DeviceScene(host.desc(), 0)
    m, w, h = host.camera(); dev.set_camera(m, w, h)
    buf = torch.zeros(w * h * 3, dtype=torch.uint8, device='cuda')
    dev.set_kernel_timing(True)
    for _ in range(3):
        dev.render_device(buf.data_ptr(), max_depth=10); c = dev.last_ray_counts()
    ms = dev.last_kernel_ms()
    print('grid %dx%d, %3d meshes: rays %d, walk items visited %d nodes %d triangle tests | %s' % (
        grid, grid, meshes, c[0], c[13], c[14], ' '.join('%s %.3f ms' % (k, v[0]) for k, v in ms.items())), flush=True)
    dev.close(); host.close()
