// One frame over several GPUs of one node from one process: the rt_multi_* entry points of include/rt_hip.h.
//
// The reference's analogue is its pool of web workers: each worker holds its own copy of the Scene, renders the rows
// y_offset, y_offset + row_jump, ... into its own buffer, and one owner stitches the rows (web_app/src/main/model.ts:93-129,
// web_app/wasm/src/renderer.rs:68-76).  Here every GPU holds the scene (an RtScene handle each), GPU i renders the
// 32x32 tiles t with t % n == i (rt_render_device), one ncclAllGather over RCCL moves the shard buffers, and the first
// GPU turns them into the row-major frame (rt_assemble_shards) and copies it to the caller's canvas.
//
// RCCL is loaded with dlopen when the first multi-device renderer is created, so the library has no load-time
// dependency on it (a process that also uses torch.distributed then shares whichever librccl.so.1 is already loaded).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/rt_hip.h"

extern "C" char* rt_internal_error_buffer(void);

namespace {

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(rt_internal_error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(RT_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// the few RCCL entry points used, bound at run time (signatures as in rccl/rccl.h)
typedef struct ncclComm* ncclComm_t;
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int /* ncclDataType_t */, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
const int kNcclUint8 = 1;          // ncclUint8 / ncclChar in rccl.h's ncclDataType_t

int load_rccl(Rccl* r) {
    static Rccl loaded;
    if (!loaded.lib) {
        void* lib = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!lib) return fail(RT_ERR_DEVICE, "cannot load RCCL (librccl.so.1): %s", dlerror());
        Rccl t;
        t.lib = lib;
        t.CommInitAll = reinterpret_cast<decltype(t.CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        t.CommDestroy = reinterpret_cast<decltype(t.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        t.AllGather = reinterpret_cast<decltype(t.AllGather)>(dlsym(lib, "ncclAllGather"));
        t.GroupStart = reinterpret_cast<decltype(t.GroupStart)>(dlsym(lib, "ncclGroupStart"));
        t.GroupEnd = reinterpret_cast<decltype(t.GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        t.GetErrorString = reinterpret_cast<decltype(t.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!t.CommInitAll || !t.CommDestroy || !t.AllGather || !t.GroupStart || !t.GroupEnd || !t.GetErrorString)
            return fail(RT_ERR_DEVICE, "librccl.so.1 lacks an entry point this library needs");
        loaded = t;
    }
    *r = loaded;
    return RT_OK;
}

}  // namespace

struct RtMulti {
    int n = 0;
    std::vector<int> devices;
    std::vector<RtScene*> scenes;          // the scene, replicated: one handle per device
    std::vector<hipStream_t> streams;
    std::vector<uint8_t*> shard;           // per device: this device's tiles of every view
    std::vector<uint8_t*> gathered;        // per device: every device's shard buffer, rank-major
    std::vector<hipEvent_t> done;          // per device: its shard is rendered (the copy path waits on these)
    uint8_t* frame = nullptr;              // on the first device: the row-major frames
    size_t shard_bytes = 0, frame_bytes = 0;
    bool same_device = false;              // every entry of `devices` names one GPU: copies instead of the collective
    Rccl rccl;
    std::vector<ncclComm_t> comms;
    uint32_t width = 0, height = 0, n_views = 1;
    bool camera_set = false;
    float last_ms = 0.0f;
    uint32_t regrown = 0;
};

static void free_buffers(RtMulti* m) {
    for (int i = 0; i < m->n; ++i) {
        if (!m->shard[i] && !m->gathered[i]) continue;
        (void)hipSetDevice(m->devices[i]);
        if (m->shard[i]) (void)hipFree(m->shard[i]);
        if (m->gathered[i]) (void)hipFree(m->gathered[i]);
        m->shard[i] = m->gathered[i] = nullptr;
    }
    if (m->frame) { (void)hipSetDevice(m->devices[0]); (void)hipFree(m->frame); m->frame = nullptr; }
    m->shard_bytes = m->frame_bytes = 0;
}

extern "C" void rt_multi_destroy(RtMulti* m) {
    if (!m) return;
    for (int i = 0; i < m->n; ++i) {
        if (!m->streams[i]) continue;
        (void)hipSetDevice(m->devices[i]);
        (void)hipStreamSynchronize(m->streams[i]);
    }
    for (ncclComm_t c : m->comms)
        if (c) (void)m->rccl.CommDestroy(c);
    free_buffers(m);
    for (int i = 0; i < m->n; ++i) {
        if (!m->scenes[i] && !m->done[i] && !m->streams[i]) continue;       // nothing was created on this device
        (void)hipSetDevice(m->devices[i]);
        if (m->scenes[i]) rt_scene_destroy(m->scenes[i]);
        if (m->done[i]) (void)hipEventDestroy(m->done[i]);
        if (m->streams[i]) (void)hipStreamDestroy(m->streams[i]);
    }
    delete m;
}

extern "C" int rt_multi_create(const RtSceneDesc* desc, const int* device_ids, int n_devices, RtMulti** out) {
    if (!desc || !device_ids || !out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_create: null argument");
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_create: %d devices; between 1 and 64 are supported", n_devices);
    RtMulti* m = new RtMulti();
    m->n = n_devices;
    m->devices.assign(device_ids, device_ids + n_devices);
    m->scenes.assign(n_devices, nullptr);
    m->streams.assign(n_devices, nullptr);
    m->shard.assign(n_devices, nullptr);
    m->gathered.assign(n_devices, nullptr);
    m->done.assign(n_devices, nullptr);
    bool distinct = true;
    m->same_device = true;
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j) {
            if (device_ids[i] == device_ids[j]) distinct = false; else m->same_device = false;
        }
    if (n_devices == 1) m->same_device = false;
    auto bail = [&](int code) { rt_multi_destroy(m); return code; };
    int n_visible = 0;
    if (hipGetDeviceCount(&n_visible) != hipSuccess || n_visible == 0) return bail(fail(RT_ERR_NO_DEVICE, "no HIP device is visible"));
    for (int i = 0; i < n_devices; ++i)
        if (device_ids[i] < 0 || device_ids[i] >= n_visible)
            return bail(fail(RT_ERR_NO_DEVICE, "rt_multi_create: device %d of %d does not exist", device_ids[i], n_visible));
    if (!distinct && !m->same_device)
        return bail(fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_create: device_ids must be all different (or, to rehearse on one GPU, all the same)"));
    for (int i = 0; i < n_devices; ++i) {
        int rc = rt_scene_create(desc, device_ids[i], &m->scenes[i]);       // validates the device id, replicates the scene
        if (rc) return bail(rc);
        if (hipSetDevice(device_ids[i]) != hipSuccess || hipStreamCreateWithFlags(&m->streams[i], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&m->done[i], hipEventDisableTiming) != hipSuccess)
            return bail(fail(RT_ERR_DEVICE, "rt_multi_create: stream / event creation failed on device %d", device_ids[i]));
    }
    if (!m->same_device) {
        // one communicator per device, all created by this process (ncclCommInitAll); with a single device the
        // collective degenerates to a copy, and the code path stays the one several devices take
        int rc = load_rccl(&m->rccl);
        if (rc) return bail(rc);
        m->comms.assign(n_devices, nullptr);
        int e = m->rccl.CommInitAll(m->comms.data(), n_devices, device_ids);
        if (e != 0) {
            m->comms.clear();
            return bail(fail(RT_ERR_DEVICE, "ncclCommInitAll over %d devices: %s", n_devices, m->rccl.GetErrorString(e)));
        }
    }
    *out = m;
    return RT_OK;
}

extern "C" int rt_multi_set_views(RtMulti* m, const float* screen_to_world, uint32_t n_views, uint32_t width, uint32_t height) {
    if (!m || !screen_to_world) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_set_views: null argument");
    for (int i = 0; i < m->n; ++i) {
        int rc = rt_scene_set_views(m->scenes[i], screen_to_world, n_views, width, height);
        if (rc) return rc;
    }
    m->width = width; m->height = height; m->n_views = n_views;
    m->camera_set = true;
    return RT_OK;
}

extern "C" int rt_multi_set_camera(RtMulti* m, const float screen_to_world[16], uint32_t width, uint32_t height) {
    return rt_multi_set_views(m, screen_to_world, 1, width, height);
}

static int ensure_buffers(RtMulti* m, const RtRenderParams* p) {
    RtRenderParams q = *p;
    q.shard_index = 0;
    q.shard_count = (uint32_t)m->n;
    uint64_t per_view = 0;
    int rc = rt_shard_bytes(m->scenes[0], &q, &per_view);
    if (rc) return rc;
    const size_t shard_bytes = (size_t)per_view * m->n_views;
    const size_t frame_bytes = (size_t)m->width * m->height * (p->output_format == RT_FORMAT_RGBA8 ? 4 : 3) * m->n_views;
    if (shard_bytes == m->shard_bytes && frame_bytes == m->frame_bytes) return RT_OK;
    for (int i = 0; i < m->n; ++i) { HIP_TRY(hipSetDevice(m->devices[i])); HIP_TRY(hipStreamSynchronize(m->streams[i])); }
    free_buffers(m);
    for (int i = 0; i < m->n; ++i) {
        HIP_TRY(hipSetDevice(m->devices[i]));
        HIP_TRY(hipMalloc((void**)&m->shard[i], shard_bytes));
        // only the first device assembles, but an all-gather delivers to everyone
        HIP_TRY(hipMalloc((void**)&m->gathered[i], (m->same_device && i > 0) ? 16 : shard_bytes * m->n));
    }
    HIP_TRY(hipSetDevice(m->devices[0]));
    HIP_TRY(hipMalloc((void**)&m->frame, frame_bytes));
    m->shard_bytes = shard_bytes;
    m->frame_bytes = frame_bytes;
    return RT_OK;
}

extern "C" int rt_multi_render(RtMulti* m, const RtRenderParams* p, uint8_t* out_pixels) {
    if (!m || !p || !out_pixels) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render: null argument");
    if (!m->camera_set) return fail(RT_ERR_INVALID_ARGUMENT, "the camera has not been set (rt_multi_set_camera)");
    if (p->stream) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render: the renderer owns one stream per device; params->stream must be NULL");
    int rc = ensure_buffers(m, p);
    if (rc) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    m->regrown = 0;
    for (;;) {
        // every device renders its tiles
        for (int i = 0; i < m->n; ++i) {
            RtRenderParams q = *p;
            q.shard_index = (uint32_t)i;
            q.shard_count = (uint32_t)m->n;
            q.stream = m->streams[i];
            if ((rc = rt_render_device(m->scenes[i], &q, m->shard[i]))) return rc;
            if (m->same_device) { HIP_TRY(hipSetDevice(m->devices[i])); HIP_TRY(hipEventRecord(m->done[i], m->streams[i])); }
        }
        // the frame's one exchange step
        if (m->same_device) {
            HIP_TRY(hipSetDevice(m->devices[0]));
            for (int i = 0; i < m->n; ++i) {
                HIP_TRY(hipStreamWaitEvent(m->streams[0], m->done[i], 0));
                HIP_TRY(hipMemcpyAsync(m->gathered[0] + (size_t)i * m->shard_bytes, m->shard[i], m->shard_bytes, hipMemcpyDeviceToDevice, m->streams[0]));
            }
        } else {
            int e = m->rccl.GroupStart();
            for (int i = 0; i < m->n && e == 0; ++i) {
                HIP_TRY(hipSetDevice(m->devices[i]));
                e = m->rccl.AllGather(m->shard[i], m->gathered[i], m->shard_bytes, kNcclUint8, m->comms[i], m->streams[i]);
            }
            const int e2 = m->rccl.GroupEnd();
            if (e == 0) e = e2;
            if (e != 0) return fail(RT_ERR_DEVICE, "ncclAllGather of %zu bytes per device: %s", m->shard_bytes, m->rccl.GetErrorString(e));
        }
        // the first device stitches the tiles into frames and hands them over
        RtRenderParams q = *p;
        q.shard_index = 0;
        q.shard_count = (uint32_t)m->n;
        q.stream = m->streams[0];
        if ((rc = rt_assemble_shards(m->scenes[0], &q, m->gathered[0], m->frame))) return rc;
        HIP_TRY(hipSetDevice(m->devices[0]));
        HIP_TRY(hipMemcpyAsync(out_pixels, m->frame, m->frame_bytes, hipMemcpyDeviceToHost, m->streams[0]));
        for (int i = 0; i < m->n; ++i) { HIP_TRY(hipSetDevice(m->devices[i])); HIP_TRY(hipStreamSynchronize(m->streams[i])); }
        // a shard whose ray lists overflowed is incomplete: its handle has longer lists now, render the frame again
        bool overflowed = false;
        uint64_t counters[RT_N_COUNTERS];
        for (int i = 0; i < m->n; ++i) {
            rc = rt_last_ray_counts(m->scenes[i], counters);
            if (rc == RT_ERR_OUT_OF_RANGE) overflowed = true;
            else if (rc) return rc;
        }
        if (!overflowed) break;
        if (++m->regrown > 16) return fail(RT_ERR_OUT_OF_RANGE, "the ray lists keep overflowing");
    }
    m->last_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}

extern "C" int rt_multi_last_ray_counts(RtMulti* m, uint64_t counters[RT_N_COUNTERS]) {
    if (!m || !counters) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_last_ray_counts: null argument");
    for (int k = 0; k < RT_N_COUNTERS; ++k) counters[k] = 0;
    for (int i = 0; i < m->n; ++i) {
        uint64_t c[RT_N_COUNTERS];
        int rc = rt_last_ray_counts(m->scenes[i], c);
        if (rc) return rc;
        for (int k = 0; k < RT_N_COUNTERS; ++k) counters[k] += c[k];
    }
    return RT_OK;
}

extern "C" int rt_multi_last_render_ms(RtMulti* m, float* out_ms) {
    if (!m || !out_ms) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_last_render_ms: null argument");
    *out_ms = m->last_ms;
    return RT_OK;
}
