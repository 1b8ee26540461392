# Writes the scenes of synthetic.mixed_scene for seeds [a, b) as files, one directory per seed, for tests/emu/emu_render:
#   python tests/emu/gen_mixed.py <a> <b> <out_dir>     ->  <out_dir>/<seed>/scene.json (+ .depth, meshes)
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from rust_raytracer_amd import synthetic

a, b, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
for seed in range(a, b):
    d = os.path.join(out, str(seed))
    os.makedirs(d, exist_ok=True)
    text, depth, what = synthetic.mixed_scene(d, seed)
    open(os.path.join(d, "scene.json"), "w").write(text)
    open(os.path.join(d, "scene.json.depth"), "w").write(str(depth))
