// rt_renderer.hpp - the frame-level renderer of the reference, in C++ over the C ABI of rt_hip.h / rt_host.h.
//
// The reference selects its renderer by constructing a struct that owns the Scene: GpuRenderer
// (web_app/wasm/src/gpu_renderer.rs:44-68 new, :280-354 render, :356-359 resize_screen, :361-364 turn_camera).
// rt::HipRenderer has the same four operations with the same argument meaning.  Where the reference panics
// (parser.rs:151-216 unwrap on a missing mesh or material, gpu_renderer.rs:47-49 on a missing scene) this class
// throws rt::Error carrying rt_last_error()'s text; nothing is caught or retried inside.
//
// Header-only; link against librt_hip.so.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_hip.h"
#include "rt_host.h"

namespace rt {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

inline void check(int rc) {
    if (rc != RT_OK) throw Error(rc, rt_last_error());
}

class HipRenderer {
public:
    // GpuRenderer::new: parse the scene (meshes from model_dir, the FilesystemMeshLoader of
    // native_app/src/filesystem_mesh_loader.rs:11-30), resize its camera to the canvas, upload it.
    // width == 0 keeps the screen_dimensions of the scene file (what native_app does, main.rs:46-48).
    HipRenderer(uint32_t canvas_width, uint32_t canvas_height, const std::string& scene_path, const std::string& model_dir,
                int device_id = 0) {
        check(rt_host_scene_load(scene_path.c_str(), model_dir.c_str(), &host_));
        try {
            if (canvas_width != 0 && canvas_height != 0) check(rt_host_scene_resize_screen(host_, canvas_width, canvas_height));
            RtSceneDesc desc;
            check(rt_host_scene_desc(host_, &desc));
            check(rt_scene_create(&desc, device_id, &scene_));
            push_camera();
        } catch (...) {
            release();
            throw;
        }
    }
    // The same renderer over several GPUs of this machine (rt_multi_*): the scene is replicated, every frame is cut
    // into 32x32 tiles dealt round-robin to the devices and stitched on the first one - the worker pool of
    // web_app/src/main/model.ts:93-129 with GPUs for workers.  The pictures are byte-identical to the one-device ones.
    HipRenderer(uint32_t canvas_width, uint32_t canvas_height, const std::string& scene_path, const std::string& model_dir,
                const std::vector<int>& device_ids) {
        check(rt_host_scene_load(scene_path.c_str(), model_dir.c_str(), &host_));
        try {
            if (canvas_width != 0 && canvas_height != 0) check(rt_host_scene_resize_screen(host_, canvas_width, canvas_height));
            RtSceneDesc desc;
            check(rt_host_scene_desc(host_, &desc));
            check(rt_multi_create(&desc, device_ids.data(), (int)device_ids.size(), &multi_));
            push_camera();
        } catch (...) {
            release();
            throw;
        }
    }
    HipRenderer(const HipRenderer&) = delete;
    HipRenderer& operator=(const HipRenderer&) = delete;
    ~HipRenderer() { release(); }

    uint32_t width() const { return width_; }
    uint32_t height() const { return height_; }

    // GpuRenderer::render(&mut self, canvas_u8: &mut [u8]): RGBA8, row 0 = top, width * height * 4 bytes.
    void render(uint8_t* canvas_u8, size_t canvas_len) { render_into(canvas_u8, canvas_len, RT_FORMAT_RGBA8, 4); }
    // native_app's Canvas layout (canvas.rs:15-46): RGB8, row 0 = top, width * height * 3 bytes.
    void render_rgb(uint8_t* canvas_u8, size_t canvas_len) { render_into(canvas_u8, canvas_len, RT_FORMAT_RGB8, 3); }

    // GpuRenderer::resize_screen(&mut self, width: f32, height: f32)
    void resize_screen(uint32_t width, uint32_t height) {
        check(rt_host_scene_resize_screen(host_, width, height));
        push_camera();
    }
    // GpuRenderer::turn_camera(&mut self, drag_begin x, y, drag_end x, y), pixel coordinates
    void turn_camera(float begin_x, float begin_y, float end_x, float end_y) {
        check(rt_host_scene_turn_camera(host_, begin_x, begin_y, end_x, end_y));
        push_camera();
    }

    // Several frames in one call (rt_scene_set_views): `n` camera drags, each applied to the camera the previous one
    // left (turn_camera semantics, scene.rs:60-97), one RGB8 frame per drag, back to back in `canvases`
    // (n * width * height * 3 bytes).  An orbit of n frames costs markedly less this way than n calls of render_rgb.
    // The reference has no counterpart: its loop would call turn_camera and render n times.
    void render_drags_rgb(const float (*drags)[4], uint32_t n, uint8_t* canvases, size_t canvases_len) {
        if (n == 0 || n > RT_MAX_VIEWS) throw Error(RT_ERR_INVALID_ARGUMENT, "between 1 and " + std::to_string(RT_MAX_VIEWS) + " frames per call");
        if (canvases_len != (size_t)n * width_ * height_ * 3)
            throw Error(RT_ERR_INVALID_ARGUMENT, "canvases of " + std::to_string(canvases_len) + " bytes for " + std::to_string(n) + " frames");
        std::vector<float> views((size_t)n * 16);
        for (uint32_t k = 0; k < n; ++k) {
            check(rt_host_scene_turn_camera(host_, drags[k][0], drags[k][1], drags[k][2], drags[k][3]));
            check(rt_host_scene_camera(host_, &views[(size_t)k * 16], &width_, &height_));
        }
        if (multi_) check(rt_multi_set_views(multi_, views.data(), n, width_, height_));
        else check(rt_scene_set_views(scene_, views.data(), n, width_, height_));
        draw(canvases, RT_FORMAT_RGB8);
        push_camera();                                     // back to one view: the camera after the last drag
    }

    // MAX_RAY_RECURSION_DEPTH (raytracer.rs:9) is a constant of the reference; here it can be set per renderer.
    void set_max_depth(uint32_t depth) { max_depth_ = depth; }
    // primary + secondary rays and milliseconds of the last render (kernels on one device; the whole call on several)
    uint64_t last_rays() const { return last_rays_; }
    float last_render_ms() const { return last_ms_; }

private:
    void push_camera() {
        float m[16];
        check(rt_host_scene_camera(host_, m, &width_, &height_));
        if (multi_) check(rt_multi_set_camera(multi_, m, width_, height_));
        else check(rt_scene_set_camera(scene_, m, width_, height_));
    }
    void draw(uint8_t* canvas, uint32_t format) {
        RtRenderParams p{};
        p.max_depth = max_depth_;
        p.output_format = format;
        if (multi_) {
            check(rt_multi_render(multi_, &p, canvas));
            uint64_t counters[RT_N_COUNTERS];
            check(rt_multi_last_ray_counts(multi_, counters));
            last_rays_ = counters[RT_CTR_RAYS];
            check(rt_multi_last_render_ms(multi_, &last_ms_));
        } else {
            p.shard_count = 1;
            RtAux aux{};
            check(rt_render(scene_, &p, canvas, &aux));
            last_rays_ = aux.counters[RT_CTR_RAYS];
            check(rt_last_render_ms(scene_, &last_ms_));
        }
    }
    void render_into(uint8_t* canvas, size_t len, uint32_t format, uint32_t channels) {
        if (len != (size_t)width_ * height_ * channels)
            throw Error(RT_ERR_INVALID_ARGUMENT, "canvas of " + std::to_string(len) + " bytes for a " + std::to_string(width_) + "x" +
                                                     std::to_string(height_) + " frame of " + std::to_string(channels) + " channels");
        draw(canvas, format);
    }
    void release() {
        if (multi_) rt_multi_destroy(multi_);
        multi_ = nullptr;
        if (scene_) rt_scene_destroy(scene_);
        if (host_) rt_host_scene_destroy(host_);
        scene_ = nullptr;
        host_ = nullptr;
    }

    RtHostScene* host_ = nullptr;
    RtScene* scene_ = nullptr;
    RtMulti* multi_ = nullptr;          // set instead of scene_ by the several-GPU constructor
    uint32_t width_ = 0, height_ = 0;
    uint32_t max_depth_ = RT_DEFAULT_MAX_DEPTH;
    uint64_t last_rays_ = 0;
    float last_ms_ = 0.0f;
};

}  // namespace rt
