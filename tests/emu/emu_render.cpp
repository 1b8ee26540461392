// Renders scene files with the library's kernels running on the CPU (this directory's stand-in for HIP) and writes, per scene,
// <out>.rgb (H*W*3 bytes), <out>.prim / .inst (u32 per pixel), .flags (u8), .t (f32), .counters (15 u64).
//   emu_render <model_dir> <max_depth_default> <scene.json> <out_prefix> [<scene.json> <out_prefix> ...]
// A scene file may carry "emu_max_depth": n at top level (ignored by the loader) - otherwise the default is used.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "../../include/rt_host.h"
extern "C" int rt_debug_set_queue_capacity(RtScene* s, uint32_t capacity);
extern "C" int rt_debug_last_regrown(RtScene* s, uint32_t* out);

static void dump(const std::string& path, const void* p, size_t n) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f || fwrite(p, 1, n, f) != n) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(2); }
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: emu_render model_dir max_depth scene.json out_prefix [...]\n"); return 2; }
    const char* model_dir = argv[1];
    const int default_depth = atoi(argv[2]);
    int failures = 0;
    for (int a = 3; a + 1 < argc; a += 2) {
        const char* path = argv[a];
        const std::string out = argv[a + 1];
        int depth = default_depth;
        {   // optional per-scene depth, written by the generator script next to the scene: <scene>.depth
            FILE* f = fopen((std::string(path) + ".depth").c_str(), "r");
            if (f) { if (fscanf(f, "%d", &depth) != 1) depth = default_depth; fclose(f); }
        }
        RtHostScene* host = nullptr;
        if (rt_host_scene_load(path, model_dir, &host) != RT_OK) { fprintf(stderr, "%s: %s\n", path, rt_last_error()); ++failures; continue; }
        RtSceneDesc desc_value;
        rt_host_scene_desc(host, &desc_value);
        const RtSceneDesc* desc = &desc_value;
        if (getenv("EMU_SHRINK_BVH")) {
            // spoils the tree in place, as synthetic.shrink_bvh_boxes does: every third node's box loses a quarter of its extent
            // on both sides, so child boxes stick out of their parents' and triangles out of their leaves'
            RtBvhNode* nodes = const_cast<RtBvhNode*>(desc->mesh_bvh_nodes);
            for (uint32_t i = 0; i < desc->n_mesh_bvh_nodes; ++i) {
                if (i % 3 != 1) continue;
                for (int c = 0; c < 3; ++c) {
                    const float w = nodes[i].aabb_max[c] - nodes[i].aabb_min[c];
                    const float q = w * 0.25f;
                    nodes[i].aabb_min[c] = nodes[i].aabb_min[c] + q;
                    nodes[i].aabb_max[c] = nodes[i].aabb_max[c] - q;
                }
            }
        }
        RtScene* scene = nullptr;
        if (rt_scene_create(desc, 0, &scene) != RT_OK) { fprintf(stderr, "%s: create: %s\n", path, rt_last_error()); rt_host_scene_destroy(host); ++failures; continue; }
        float m[16]; uint32_t w = 0, h = 0;
        rt_host_scene_camera(host, m, &w, &h);
        {   // optional camera matrix, written next to the scene: <scene>.camera = 16 floats (column-major screen_to_world), width, height
            FILE* f = fopen((std::string(path) + ".camera").c_str(), "r");
            if (f) {
                float mm[16]; unsigned ww = 0, hh = 0;
                bool ok = true;
                for (int k = 0; k < 16; ++k) ok = ok && fscanf(f, "%g", &mm[k]) == 1;
                ok = ok && fscanf(f, "%u %u", &ww, &hh) == 2;
                if (ok) { memcpy(m, mm, sizeof(m)); w = ww; h = hh; }
                fclose(f);
            }
        }
        rt_scene_set_camera(scene, m, w, h);
        if (const char* cap = getenv("EMU_QUEUE_CAPACITY")) rt_debug_set_queue_capacity(scene, (uint32_t)atoi(cap));   // forces overflows, as the tests do
        const size_t n = (size_t)w * h;
        std::vector<uint8_t> rgb(n * 3), flags(n);
        std::vector<uint32_t> prim(n), inst(n);
        std::vector<float> t(n);
        RtAux aux;
        memset(&aux, 0, sizeof(aux));
        aux.prim_id = prim.data(); aux.instance_id = inst.data(); aux.flags = flags.data(); aux.hit_t = t.data();
        RtRenderParams p;
        memset(&p, 0, sizeof(p));
        p.max_depth = (uint32_t)depth; p.output_format = RT_FORMAT_RGB8; p.shard_index = 0; p.shard_count = 1;
        // EMU_MODE picks another entry point for the same picture (colours only, no aux planes):
        //   shards:<n>  n tile shards through rt_render_device, then rt_assemble_shards
        //   views:<n>   n views of the same camera in one pass (rt_scene_set_views); the first is written, all must be equal
        //   noalloc     rt_render_device_noalloc
        //   depthmap    rt_render in depth-map mode
        //   trim        rt_render twice: the second frame's lists are sized by what the first one held (rt_last_list_use) and
        //               must take less memory and give the same picture
        const char* mode = getenv("EMU_MODE");
        int rc = RT_OK;
        bool with_aux = false;
        if (mode && !strncmp(mode, "shards:", 7)) {
            const uint32_t shards = (uint32_t)atoi(mode + 7);
            p.shard_count = shards;
            uint64_t bytes = 0;
            rc = rt_shard_bytes(scene, &p, &bytes);
            std::vector<uint8_t> gathered((size_t)bytes * shards);
            for (uint32_t k = 0; k < shards && rc == RT_OK; ++k) {
                p.shard_index = k;
                rc = rt_render_device(scene, &p, gathered.data() + (size_t)bytes * k);
                uint64_t counters[RT_N_COUNTERS];
                if (rc == RT_OK) rc = rt_last_ray_counts(scene, counters);
            }
            if (rc == RT_OK) rc = rt_assemble_shards(scene, &p, gathered.data(), rgb.data());
        } else if (mode && !strncmp(mode, "views:", 6)) {
            const uint32_t views = (uint32_t)atoi(mode + 6);
            std::vector<float> ms((size_t)16 * views);
            for (uint32_t v = 0; v < views; ++v) memcpy(&ms[16 * v], m, sizeof(m));
            rc = rt_scene_set_views(scene, ms.data(), views, w, h);
            std::vector<uint8_t> all(n * 3 * views);
            if (rc == RT_OK) rc = rt_render(scene, &p, all.data(), nullptr);
            for (uint32_t v = 1; v < views && rc == RT_OK; ++v)
                if (memcmp(all.data(), all.data() + n * 3 * v, n * 3)) { fprintf(stderr, "%s: view %u differs from view 0\n", path, v); rc = -100; }
            memcpy(rgb.data(), all.data(), n * 3);
        } else if (mode && !strcmp(mode, "noalloc")) {
            rc = rt_render_device_noalloc(scene, &p, rgb.data());
            uint64_t counters[RT_N_COUNTERS];
            if (rc == RT_OK) rc = rt_last_ray_counts(scene, counters);
        } else if (mode && !strcmp(mode, "trim")) {
            std::vector<uint8_t> first(n * 3);
            uint64_t before[RT_N_LIST_USE], after[RT_N_LIST_USE];
            rc = rt_render(scene, &p, first.data(), nullptr);
            if (rc == RT_OK) rc = rt_last_list_use(scene, before);
            if (rc == RT_OK) rc = rt_render(scene, &p, rgb.data(), nullptr);
            if (rc == RT_OK) rc = rt_last_list_use(scene, after);
            if (rc == RT_OK) {
                printf("%s: lists %llu -> %llu bytes, ray slots %llu -> %llu for %llu rays\n", path, (unsigned long long)before[RT_LIST_USE_BYTES],
                       (unsigned long long)after[RT_LIST_USE_BYTES], (unsigned long long)before[RT_LIST_USE_RAY_SLOTS],
                       (unsigned long long)after[RT_LIST_USE_RAY_SLOTS], (unsigned long long)after[RT_LIST_USE_RAYS]);
                const bool smaller = after[RT_LIST_USE_BYTES] * 10 < before[RT_LIST_USE_BYTES] * 7 && after[RT_LIST_USE_RAY_SLOTS] >= after[RT_LIST_USE_RAYS] &&
                                     after[RT_LIST_USE_WALK_SLOTS] >= after[RT_LIST_USE_WALK_ITEMS] && after[RT_LIST_USE_HIT_SLOTS] >= after[RT_LIST_USE_HITS];
                if (!smaller) { fprintf(stderr, "%s: the second frame's buffers are not smaller, or shorter than what it held\n", path); rc = -100; }
                else if (memcmp(first.data(), rgb.data(), n * 3)) { fprintf(stderr, "%s: the second frame differs from the first\n", path); rc = -100; }
            }
        } else if (mode && !strcmp(mode, "depthmap")) {
            p.mode = RT_MODE_DEPTH_MAP;
            rc = rt_render(scene, &p, rgb.data(), nullptr);
        } else {
            with_aux = true;
            rc = rt_render(scene, &p, rgb.data(), &aux);
        }
        if (rc != RT_OK) { fprintf(stderr, "%s: render: %s\n", path, rc == -100 ? "frames differ (see above)" : rt_last_error()); ++failures; }
        else {
            dump(out + ".rgb", rgb.data(), rgb.size());
            if (with_aux) {
                dump(out + ".prim", prim.data(), n * 4); dump(out + ".inst", inst.data(), n * 4);
                dump(out + ".flags", flags.data(), n); dump(out + ".t", t.data(), n * 4); dump(out + ".counters", aux.counters, sizeof(aux.counters));
            }
            uint32_t redone = 0;
            rt_debug_last_regrown(scene, &redone);
            printf("%s: %ux%u depth %d mode %s rays %llu, rendered %u time(s)\n", path, w, h, depth, mode ? mode : "plain", (unsigned long long)aux.counters[0], redone + 1);
            fflush(stdout);
        }
        rt_scene_destroy(scene);
        rt_host_scene_destroy(host);
    }
    return failures ? 1 : 0;
}
