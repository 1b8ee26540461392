// native_app on the HIP path: what native_app/src/main.rs does - parse a scene file, paint it, write a PNG, print the
// two durations - with paint_scene (main.rs:46-66) replaced by rt::HipRenderer.  The reference's three paths are
// constants (main.rs:17-23); here they are arguments with the same defaults.
//
//     native_app [--gpus N | --rehearse-gpus N] [scene.json [model_dir [image.png [max_depth [n_orbit]]]]]
//
// --gpus N renders every frame on the devices 0 .. N-1 of this machine (tiles dealt round-robin, stitched on device 0,
// include/rt_hip.h rt_multi_*); --rehearse-gpus N runs the same schedule with N shards on device 0 alone.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../include/rt_renderer.hpp"

int main(int argc, char** argv) {
    std::vector<int> devices;
    if (argc > 2 && (std::string(argv[1]) == "--gpus" || std::string(argv[1]) == "--rehearse-gpus")) {
        const bool rehearse = std::string(argv[1]) == "--rehearse-gpus";
        const int n = std::atoi(argv[2]);
        if (n < 1) { std::fprintf(stderr, "native_app: %s needs a positive count\n", argv[1]); return 2; }
        for (int i = 0; i < n; ++i) devices.push_back(rehearse ? 0 : i);
        argc -= 2;
        argv += 2;
    }
    const std::string scene_path = argc > 1 ? argv[1] : "res/scenes/og_scene_rust.json";
    const std::string model_dir = argc > 2 ? argv[2] : "res/models";
    const std::string image_path = argc > 3 ? argv[3] : "render.png";
    const uint32_t max_depth = argc > 4 ? (uint32_t)std::atoi(argv[4]) : RT_DEFAULT_MAX_DEPTH;
    using clock = std::chrono::steady_clock;
    try {
        auto t0 = clock::now();
        // the scene file's own screen_dimensions
        rt::HipRenderer renderer = devices.empty() ? rt::HipRenderer(0, 0, scene_path, model_dir) : rt::HipRenderer(0, 0, scene_path, model_dir, devices);
        renderer.set_max_depth(max_depth);
        std::printf("Parsing took %.3fs\n", std::chrono::duration<float>(clock::now() - t0).count());

        std::vector<uint8_t> canvas((size_t)renderer.width() * renderer.height() * 3);
        t0 = clock::now();
        renderer.render_rgb(canvas.data(), canvas.size());
        std::printf("Rendering took %.3fs\n", std::chrono::duration<float>(clock::now() - t0).count());
        std::printf("%llu rays, %.3f ms on %s\n", (unsigned long long)renderer.last_rays(), renderer.last_render_ms(),
                    devices.empty() ? "the device" : (std::to_string(devices.size()) + " devices, exchange and copy included").c_str());

        rt::check(rt_host_write_png(image_path.c_str(), canvas.data(), renderer.width(), renderer.height(), 3));

        // native_app [...] max_depth n_orbit: n_orbit more frames, the camera dragged 40 pixels to the right before each,
        // rendered in one call and written next to the first as <image>.<k>.png
        const uint32_t n_orbit = argc > 5 ? (uint32_t)std::atoi(argv[5]) : 0;
        if (n_orbit) {
            std::vector<float> drags((size_t)n_orbit * 4);
            for (uint32_t k = 0; k < n_orbit; ++k) { drags[4 * k] = 0.0f; drags[4 * k + 1] = 0.0f; drags[4 * k + 2] = 40.0f; drags[4 * k + 3] = 0.0f; }
            std::vector<uint8_t> frames((size_t)n_orbit * canvas.size());
            t0 = clock::now();
            renderer.render_drags_rgb(reinterpret_cast<const float(*)[4]>(drags.data()), n_orbit, frames.data(), frames.size());
            std::printf("Rendering %u more frames took %.3fs\n", n_orbit, std::chrono::duration<float>(clock::now() - t0).count());
            for (uint32_t k = 0; k < n_orbit; ++k)
                rt::check(rt_host_write_png((image_path + "." + std::to_string(k) + ".png").c_str(), frames.data() + (size_t)k * canvas.size(),
                                            renderer.width(), renderer.height(), 3));
        }
    } catch (const rt::Error& e) {
        std::fprintf(stderr, "native_app: %s (code %d)\n", e.what(), e.code);   // the reference panics here
        return 1;
    }
    return 0;
}
