# Scenes of the fuzz generators through the kernels on the CPU (emu_render, under AddressSanitizer) and against the oracle:
#   python tests/emu/fuzz_emu.py random|spoiled|mixed <a> <b> [work_dir]
# Every seed gets a directory of its own; the run stops at the first sanitizer report or difference.
import sys, os, json, subprocess, tempfile
sys.path.insert(0, '/root/repo')
from rust_raytracer_amd import synthetic
HERE = os.path.dirname(os.path.abspath(__file__))
kind, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
work = sys.argv[4] if len(sys.argv) > 4 else tempfile.mkdtemp(prefix='emu_' + kind + '_')
depth = {'random': 6, 'spoiled': 5, 'mixed': 5}[kind]
refused = 0
for seed in range(a, b):
    d = os.path.join(work, str(seed))
    os.makedirs(d, exist_ok=True)
    scene = os.path.join(d, 'scene.json')
    if kind == 'random':
        open(scene, 'w').write(synthetic.random_scene(d, seed))
    elif kind == 'spoiled':
        open(scene, 'w').write(synthetic.spoiled_scene(d, seed)[0])
    else:
        subprocess.run([sys.executable, os.path.join(HERE, 'gen_mixed.py'), str(seed), str(seed + 1), work], check=True)
    r = subprocess.run([os.path.join(HERE, 'emu_render'), d, str(depth), scene, os.path.join(d, 'out')], capture_output=True, text=True)
    if r.returncode != 0:
        if 'BVH builder' in r.stderr or 'BVH builder' in r.stdout:
            refused += 1; continue                      # the reference panics on such a mesh as well
        print('seed', seed, 'emu_render failed:\n', r.stderr[-3000:]); sys.exit(1)
    c = subprocess.run([sys.executable, os.path.join(HERE, 'compare.py'), d, str(depth), scene, os.path.join(d, 'out')], capture_output=True, text=True)
    if c.returncode != 0:
        print('seed', seed, c.stdout[-500:], c.stderr[-500:]); sys.exit(1)
    if seed % 50 == 0: print('at seed', seed, flush=True)
print(kind, 'seeds', a, b, 'clean; refused', refused)
