# Random and spoiled scenes seen from far away - 1e3 to 1e9 units, perspective with a narrow field of view or orthographic,
# aimed at the scene: the rays' mesh-local origins lose the boxes to rounding (traverse_heavy_pairs), culling boxes are not
# trusted any more, t is tens of ulps coarse.  Device against oracle.  python scratch/fuzz_far_camera.py a b
import sys, os, tempfile, json, random, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import orc
from rust_raytracer_amd import synthetic, abi
import test_gpu_parity as T
a, b = int(sys.argv[1]), int(sys.argv[2])
d = tempfile.mkdtemp()
bad, heavy = [], 0
for seed in range(a, b):
    rng = random.Random(seed * 15485863 + 11)
    pick = rng.random()
    if pick < 0.3:
        scene = json.loads(synthetic.far_origin_scene(d, 4)[0])        # meshes with enough leaves for the leaf-by-leaf path
        scene['camera']['screen_dimensions'] = [48, 32]
    elif pick < 0.65:
        scene = json.loads(synthetic.random_scene(d, seed))
    else:
        scene = json.loads(synthetic.spoiled_scene(d, seed, 64, 40)[0])
    cam = scene['camera']
    dist = 10.0 ** rng.uniform(3, 9)
    yaw, pitch = rng.uniform(0, 360), rng.uniform(-80, 80)
    # the camera looks down -z rotated by Ry(yaw) Rx(pitch): put it `dist` behind the scene's centre along its view direction
    cy, sy, cp, sp = math.cos(math.radians(yaw)), math.sin(math.radians(yaw)), math.cos(math.radians(pitch)), math.sin(math.radians(pitch))
    view = (-sy * cp, sp, -cy * cp)
    centre = (0.0, 0.0, -6.0)
    cam['position'] = {'x': centre[0] - view[0] * dist, 'y': centre[1] - view[1] * dist, 'z': centre[2] - view[2] * dist}
    cam['orientation_degrees'] = {'x': pitch, 'y': yaw, 'z': rng.choice([0.0, rng.uniform(0, 360)])}
    if rng.random() < 0.5:
        cam['use_perspective'] = False; cam['ortho_world_height'] = rng.choice([4.0, 12.0, 40.0])
    else:
        cam['use_perspective'] = True; cam['y_fov_degrees'] = math.degrees(2 * math.atan(rng.choice([3.0, 8.0, 20.0]) / dist))
    cam['z_near'] = 0.1; cam['z_far'] = dist * 2
    s = orc.OracleScene.from_text(json.dumps(scene), d)
    got, want = T.render_both(s, max_depth=4)
    try:
        T.assert_parity(got, want, min_within_1=0.98)
    except AssertionError as e:
        bad.append(seed); print('seed', seed, 'FAILED dist %.3g' % dist, str(e)[:300], flush=True)
    heavy += int(got['counters'][abi.RT_CTR_WALK_HEAVY_PAIRS])
    if seed % 100 == 0: print('at seed', seed, flush=True)
print('seeds', a, b, 'failures', bad, 'pairs tested leaf by leaf', heavy)
