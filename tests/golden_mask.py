"""Which pixels of the reference's render.png can pin the oracle although bunny.obj is missing.

render.png is native_app's picture of res/scenes/og_scene_rust.json (= scenes/reflective_bunnies.json) at 3840x2160.
The scene's four mesh instances use a bunny.obj the reference's repository does not hold, so a pixel is only usable
if nothing in its ray tree - primary ray, shadow rays, reflected and transmitted rays down to the depth limit - can
touch a bunny with a weight that matters.  This module finds those pixels with the oracle alone: every bunny instance is
replaced by a proxy, a closed box (in the mesh's own coordinates) that contains the bunny, once with a material that
emits 4.0 in every channel and reflects nothing, once with one that is perfectly black.  A ray that would meet the
real bunny meets the box first; the bunnies are opaque and what they send back along a ray lies between those two
extremes (their materials: emission + ambient + diffuse + specular + 0.8 x a reflection stay below 4), and a box
shadows whatever the bunny inside it shadows.  A pixel whose 8-bit colour is the same in the scene without meshes and in
both proxy scenes therefore cannot be moved by the real bunnies by more than about one step.  Those pixels, minus a
margin of a few pixels, are the usable ones.

Run as a script (in the container that holds /root/reference) it writes tests/golden/reference_render_bunny_free.png:
render.png with alpha 255 on the usable pixels, alpha 128 on the remaining pixels whose primary ray hits one of the two
glass spheres (bunnies show through them almost everywhere, so these pixels only support a statistical statement:
most of them must still agree), and every other pixel set to (0, 0, 0, 0) - data only, no reference source.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
if os.path.dirname(HERE) not in sys.path:
    sys.path.insert(0, os.path.dirname(HERE))
SCENE = os.path.join(os.path.dirname(HERE), "scenes", "reflective_bunnies.json")
GOLDEN = os.path.join(HERE, "golden", "reference_render_bunny_free.png")
# Mesh-local box around the bunny.  bunny.obj is unknown, so the box was fitted to the picture: starting from
# [-2, 2] x [-1.5, 2.5] x [-2, 2], every face was moved inwards for as long as the pixels left usable still agreed with the
# oracle's render of the scene without meshes (a face that cuts into a bunny, or into the shadow volume of one, uncovers
# thousands of differing pixels at once), then moved back out by 0.06.  The box comes out much larger than a bunny: the
# file's own frame is evidently not centred on the animal.  The box only chooses which pixels are left out; the pixels kept are compared
# with a render that knows nothing about it.
PROXY_BOX = ((-0.31, -0.81, -1.06), (0.5, 2.31, 2.06))
GROW = 4            # pixels the mask is grown by (silhouettes move by a pixel with the last bits of the camera matrices)


def box_obj(lo, hi):
    """OBJ text of the closed box [lo, hi]: 12 triangles, counter-clockwise from outside, outward normals."""
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    v = [(x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)]
    quads = [((0, 3, 2, 1), (0, 0, -1)), ((4, 5, 6, 7), (0, 0, 1)), ((0, 1, 5, 4), (0, -1, 0)),
             ((3, 7, 6, 2), (0, 1, 0)), ((0, 4, 7, 3), (-1, 0, 0)), ((1, 2, 6, 5), (1, 0, 0))]
    out = ["o proxy"] + ["v %.6f %.6f %.6f" % p for p in v] + ["vn %d %d %d" % n for _, n in quads]
    for k, (q, _) in enumerate(quads):
        a, b, c, d = (i + 1 for i in q)
        out.append("f %d//%d %d//%d %d//%d" % (a, k + 1, b, k + 1, c, k + 1))
        out.append("f %d//%d %d//%d %d//%d" % (a, k + 1, c, k + 1, d, k + 1))
    return "\n".join(out) + "\n"


PROXY_EMISSIONS = (4.0, 0.0)       # the two extremes of what a bunny can send back along a ray


def scenes():
    """(scene without meshes, [scene with bright proxies, scene with black proxies], directory holding the proxy's OBJ);
    the scenes as JSON texts."""
    import tempfile
    scene = json.load(open(SCENE))
    bare = dict(scene, meshes=[], mesh_instances=[])
    d = tempfile.mkdtemp(prefix="bunny_proxy_")
    proxies = []
    for e in PROXY_EMISSIONS:
        proxy = json.loads(json.dumps(scene))
        # the Phong sum adds the emission once per light (raytracer.rs:137-153)
        per_light = e / max(len(scene["lights"]), 1)
        proxy["materials"].append({"name": "proxy", "emissive": {"r": per_light, "g": per_light, "b": per_light},
                                   "ambient": {"r": 0, "g": 0, "b": 0}, "diffuse": {"r": 0, "g": 0, "b": 0},
                                   "specular": {"r": 0, "g": 0, "b": 0}, "shininess": 1, "material_type": "Phong"})
        for mesh in proxy["meshes"]:
            mesh["material"] = "proxy"
            with open(os.path.join(d, mesh["file_name"]), "w") as fh:
                fh.write(box_obj(*PROXY_BOX))
        for inst in proxy["mesh_instances"]:
            inst["material_override"] = "proxy"
        proxies.append(json.dumps(proxy))
    return json.dumps(bare), proxies, d


def bunny_free_mask(threads=0):
    """(usable[H, W] bool, the oracle's render of the scene without meshes: dict with rgb, prim_id, flags)."""
    from scipy import ndimage
    from oracle import orc
    bare_text, proxy_texts, d = scenes()
    bare = orc.OracleScene.from_text(bare_text).render(10, threads=threads)
    touched = np.zeros(bare["rgb"].shape[:2], dtype=bool)
    for text in proxy_texts:
        proxy = orc.OracleScene.from_text(text, d).render(10, threads=threads, aux=False)
        touched |= (bare["rgb"] != proxy["rgb"]).any(axis=2)
    touched = ndimage.binary_dilation(touched, iterations=GROW)
    return ~touched, bare


GLASS_SPHERES = (0, 2)          # Scene::spheres entries with the ReflectAndRefract material `transparent`


def glass_sphere_pixels(bare):
    """Pixels whose primary ray hits a glass sphere, from the aux planes of the oracle's render without meshes."""
    prim = bare["prim_id"]
    hit = (bare["flags"] & 1) != 0
    return hit & (prim >> 30 == 1) & np.isin(prim & 0x3FFFFFFF, GLASS_SPHERES)


if __name__ == "__main__":
    from PIL import Image
    ref = np.array(Image.open("/root/reference/render.png").convert("RGB"))
    usable, bare = bunny_free_mask()
    out = np.zeros(ref.shape[:2] + (4,), dtype=np.uint8)
    glass = glass_sphere_pixels(bare) & ~usable
    out[glass, :3] = ref[glass]
    out[glass, 3] = 128
    out[usable, :3] = ref[usable]
    out[usable, 3] = 255
    Image.fromarray(out, "RGBA").save(GOLDEN, optimize=True)
    diff = np.abs(bare["rgb"].astype(int) - ref.astype(int)).max(axis=2)
    print("usable %.1f %% of the picture; of those within 1 LSB of the oracle: %.5f; file %d bytes"
          % (100 * usable.mean(), (diff[usable] <= 1).mean(), os.path.getsize(GOLDEN)))
