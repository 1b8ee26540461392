# Runs one of the scratch/fuzz_*.py scripts with the device replaced by emu_render (the kernels on the CPU under
# AddressSanitizer): python tests/emu/with_emu.py scratch/fuzz_meshes.py 0 200
# The scripts build their scenes with OracleScene.from_text(text, dir) and call test_gpu_parity.render_both(scene, ...);
# both are wrapped here: the text is written to a file, emu_render renders it, and its dumps come back as the `got` dict.
import sys, os, runpy, subprocess, tempfile
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from oracle import orc
import test_gpu_parity as T
HERE = os.path.dirname(os.path.abspath(__file__))
_from_text = orc.OracleScene.from_text


def from_text(text, model_dir="", *a, **k):
    s = _from_text(text, model_dir, *a, **k)
    s._emu_text, s._emu_dir = text, model_dir
    return s


orc.OracleScene.from_text = staticmethod(from_text)
counter = [0]


def render_both(oscene, max_depth=10, queue_capacity=0):
    work = tempfile.mkdtemp(prefix='emu_scene_')
    path = os.path.join(work, 'scene.json')
    open(path, 'w').write(oscene._emu_text)
    env = dict(os.environ)
    if queue_capacity: env['EMU_QUEUE_CAPACITY'] = str(queue_capacity)
    r = subprocess.run([os.path.join(HERE, 'emu_render'), oscene._emu_dir, str(max_depth), path, os.path.join(work, 'out')],
                       capture_output=True, text=True, env=env)
    if r.returncode != 0:
        print('emu_render failed on scene', path, '\n', r.stderr[-4000:], flush=True)
        sys.exit(1)
    want = oscene.render(max_depth=max_depth)
    h, w = want['rgb'].shape[:2]
    out = os.path.join(work, 'out')
    got = {'rgb': np.fromfile(out + '.rgb', dtype=np.uint8).reshape(h, w, 3),
           'prim_id': np.fromfile(out + '.prim', dtype=np.uint32).reshape(h, w),
           'instance_id': np.fromfile(out + '.inst', dtype=np.uint32).reshape(h, w),
           'flags': np.fromfile(out + '.flags', dtype=np.uint8).reshape(h, w),
           'hit_t': np.fromfile(out + '.t', dtype=np.float32).reshape(h, w),
           'counters': np.fromfile(out + '.counters', dtype=np.uint64), 'regrown': 0}
    for f in os.listdir(work): os.unlink(os.path.join(work, f))
    os.rmdir(work)
    counter[0] += 1
    return got, want


T.render_both = render_both
script = sys.argv[1]
sys.argv = sys.argv[1:]
runpy.run_path(script, run_name='__main__')
print('scenes rendered on the CPU:', counter[0])
