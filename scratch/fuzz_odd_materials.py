# Random scenes whose materials and lights get zero, negative-zero, infinite and NaN colours and shininess of 0, below 0 and
# infinity written into the scene description in place: what decides whether a shadow ray is traced at all (shadow_is_moot).
# Device against oracle, with and without the per-pixel flags (render_both).  python scratch/fuzz_odd_materials.py a b
import sys, os, tempfile, random
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import orc
from rust_raytracer_amd import synthetic, abi
import test_gpu_parity as T
a, b = int(sys.argv[1]), int(sys.argv[2])
d = tempfile.mkdtemp()
inf, nan = float('inf'), float('nan')
odd = [0.0, -0.0, inf, -inf, nan, 1e-30, 1e30, 0.5]
bad, skipped = [], 0
for seed in range(a, b):
    rng = random.Random(seed * 104729 + 7)
    s = orc.OracleScene.from_text(synthetic.random_scene(d, seed), d)
    desc = s.desc()
    what = []
    for k in range(desc.n_materials):
        m = desc.materials[k]
        if rng.random() < 0.5:
            for c in range(3):
                if rng.random() < 0.6: m.diffuse[c] = rng.choice(odd)
                if rng.random() < 0.6: m.specular[c] = rng.choice(odd)
            what.append('colours %d' % k)
        if rng.random() < 0.4:
            m.shininess = rng.choice([0.0, -0.0, -2.0, inf, -inf, nan, 1e-3, 1e4]); what.append('shininess %d' % k)
        if rng.random() < 0.15:
            m.emissive[rng.randrange(3)] = rng.choice([inf, nan, -1.0]); what.append('emissive %d' % k)
    for k in range(desc.n_lights):
        L = desc.lights[k]
        if rng.random() < 0.4:
            for c in range(3):
                if rng.random() < 0.6: L.diffuse[c] = rng.choice(odd)
                if rng.random() < 0.6: L.specular[c] = rng.choice(odd)
            what.append('light %d' % k)
    got, want = T.render_both(s, max_depth=5)
    try:
        T.assert_parity(got, want, min_within_1=0.995)
    except AssertionError as e:
        bad.append(seed); print('seed', seed, 'FAILED', what, str(e)[:300], flush=True)
    skipped += int(got['counters'][abi.RT_CTR_SHADOW_NOT_TRACED])
    if seed % 100 == 0: print('at seed', seed, flush=True)
print('seeds', a, b, 'failures', bad, 'shadow rays not traced', skipped)
