# Compares what emu_render wrote for a scene with the oracle's render of it: python compare.py <model_dir> <depth> <scene.json> <out_prefix> ...
import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from oracle import orc
model_dir, depth = sys.argv[1], int(sys.argv[2])
bad = 0
for path, out in zip(sys.argv[3::2], sys.argv[4::2]):
    d = depth
    if os.path.exists(path + '.depth'): d = int(open(path + '.depth').read())
    if not os.path.exists(out + '.rgb'):
        print(path, 'NO OUTPUT'); bad += 1; continue
    s = orc.OracleScene.from_file(path, model_dir)
    want = s.render(d)
    h, w = want['rgb'].shape[:2]
    rgb = np.fromfile(out + '.rgb', dtype=np.uint8).reshape(h, w, 3)
    prim = np.fromfile(out + '.prim', dtype=np.uint32).reshape(h, w)
    inst = np.fromfile(out + '.inst', dtype=np.uint32).reshape(h, w)
    flags = np.fromfile(out + '.flags', dtype=np.uint8).reshape(h, w)
    t = np.fromfile(out + '.t', dtype=np.float32).reshape(h, w)
    counters = np.fromfile(out + '.counters', dtype=np.uint64)
    problems = []
    if not np.array_equal(prim, want['prim_id']): problems.append('prim ids differ on %d pixels' % (prim != want['prim_id']).sum())
    if not np.array_equal(inst, want['instance_id']): problems.append('instance ids differ')
    if not np.array_equal(flags, want['flags']): problems.append('flags differ on %d pixels' % (flags != want['flags']).sum())
    if not np.array_equal(t, want['hit_t'], equal_nan=True): problems.append('t differs')
    if counters[:5].tolist() != [int(x) for x in want['counters'][:5]]: problems.append('ray counts %s vs %s' % (counters[:5].tolist(), [int(x) for x in want['counters'][:5]]))
    diff = np.abs(rgb.astype(int) - want['rgb'].astype(int)).max(axis=2)
    within = (diff <= 1).mean()
    if within < 0.98: problems.append('only %.4f within 1 LSB' % within)
    print(path, 'OK' if not problems else 'DIFFERS: ' + '; '.join(problems), 'identical %.4f' % (diff == 0).mean(), flush=True)
    bad += bool(problems)
sys.exit(1 if bad else 0)
