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
        RtScene* scene = nullptr;
        if (rt_scene_create(desc, 0, &scene) != RT_OK) { fprintf(stderr, "%s: create: %s\n", path, rt_last_error()); rt_host_scene_destroy(host); ++failures; continue; }
        float m[16]; uint32_t w = 0, h = 0;
        rt_host_scene_camera(host, m, &w, &h);
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
        const int rc = rt_render(scene, &p, rgb.data(), &aux);
        if (rc != RT_OK) { fprintf(stderr, "%s: render: %s\n", path, rt_last_error()); ++failures; }
        else {
            dump(out + ".rgb", rgb.data(), rgb.size()); dump(out + ".prim", prim.data(), n * 4); dump(out + ".inst", inst.data(), n * 4);
            dump(out + ".flags", flags.data(), n); dump(out + ".t", t.data(), n * 4); dump(out + ".counters", aux.counters, sizeof(aux.counters));
            printf("%s: %ux%u depth %d rays %llu\n", path, w, h, depth, (unsigned long long)aux.counters[0]);
            fflush(stdout);
        }
        rt_scene_destroy(scene);
        rt_host_scene_destroy(host);
    }
    return failures ? 1 : 0;
}
