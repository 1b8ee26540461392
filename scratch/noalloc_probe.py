import os, sys
sys.path.insert(0, '/root/repo')
from rust_raytracer_amd import native
host = native.HostScene.load('/root/repo/scenes/cornell_box.json', '')
print("creating", flush=True)
dev = native.DeviceScene(host.desc(), 0)
print("created", flush=True)
