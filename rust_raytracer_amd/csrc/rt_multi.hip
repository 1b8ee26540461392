// One frame over several GPUs of one node from one process: the rt_multi_* entry points of include/rt_hip.h.
//
// The reference's analogue is its pool of web workers: each worker holds its own copy of the Scene, renders the rows
// y_offset, y_offset + row_jump, ... into its own buffer - all workers at once, one thread each
// (web_app/src/worker/render_worker.ts:71-76) - and one owner stitches the rows (web_app/src/main/model.ts:93-129,
// web_app/wasm/src/renderer.rs:68-76).  Here every GPU holds the scene (an RtScene handle each) and has a host thread of
// its own that issues its launches, so the devices are fed side by side; GPU i renders the 32x32 tiles t with t % n == i
// (rt_render_device), one ncclAllGather over RCCL moves the shard buffers, and the first GPU turns them into the
// row-major frame (rt_assemble_shards) and copies it to the caller's canvas.  Exchange, stitching and copy run on a second
// stream per device and on a second set of buffers, so that with several passes (rt_multi_render_views) pass k + 1 is
// traced while pass k is gathered and handed over.
//
// RCCL is loaded with dlopen when the first multi-device renderer is created, so the library has no load-time
// dependency on it (a process that also uses torch.distributed then shares whichever librccl.so.1 is already loaded).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
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

// The host thread of one device: it makes the device current once and then issues, job by job, what the caller's thread
// hands it (the launches of one shard).  rt_last_error() is thread-local, so a failure's text travels back in `error`.
struct Worker {
    std::thread thread;
    std::mutex lock;
    std::condition_variable wake;
    enum { IDLE, RENDER, QUIT } job = IDLE;
    bool finished = true;
    RtScene* scene = nullptr;
    int device = 0;
    RtRenderParams params{};
    uint8_t* out = nullptr;
    hipEvent_t done = nullptr;
    int rc = RT_OK;
    char error[512] = "";

    void run() {
        (void)hipSetDevice(device);
        for (;;) {
            std::unique_lock<std::mutex> hold(lock);
            wake.wait(hold, [&] { return job != IDLE; });
            if (job == QUIT) return;
            rc = rt_render_device(scene, &params, out);
            if (rc == RT_OK && hipEventRecord(done, (hipStream_t)params.stream) != hipSuccess) {
                rc = RT_ERR_DEVICE;
                snprintf(rt_internal_error_buffer(), 512, "hipEventRecord failed on device %d", device);
            }
            if (rc != RT_OK) { strncpy(error, rt_last_error(), sizeof(error) - 1); error[sizeof(error) - 1] = 0; }
            job = IDLE;
            finished = true;
            wake.notify_all();
        }
    }
    void post(const RtRenderParams& p, uint8_t* dst) {
        std::lock_guard<std::mutex> hold(lock);
        params = p; out = dst; finished = false; job = RENDER;
        wake.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> hold(lock);
        wake.wait(hold, [&] { return finished; });
        return rc;
    }
};

struct RtMulti {
    int n = 0;
    std::vector<int> devices;
    std::vector<RtScene*> scenes;          // the scene, replicated: one handle per device
    std::vector<hipStream_t> streams;      // per device: the launches of its shards
    std::vector<hipStream_t> xstreams;     // per device: exchange, stitching, copy to the host
    // two sets of buffers, used alternately by successive passes
    std::vector<uint8_t*> shard[2];        // per device: this device's tiles of every view of a pass
    std::vector<uint8_t*> gathered[2];     // per device: every device's shard buffer, rank-major
    uint8_t* frame[2] = {nullptr, nullptr};   // on the first device: the row-major frames of a pass
    std::vector<hipEvent_t> done;          // per device: its shard is rendered (the exchange waits on these)
    std::vector<hipEvent_t> taken[2];      // per device and buffer set: the exchange has read the shard buffer
    std::vector<Worker*> workers;          // one issuing thread per device (none for a single device)
    size_t shard_view_bytes = 0;           // one view of one device's shard
    uint32_t buffer_views = 0, buffer_pixel_bytes = 0;      // what the buffers were made for
    bool same_device = false;              // every entry of `devices` names one GPU: copies instead of the collective
    Rccl rccl;
    std::vector<ncclComm_t> comms;
    uint32_t width = 0, height = 0, n_views = 1;
    std::vector<float> views;              // what rt_multi_set_views was given
    bool camera_set = false;
    float last_ms = 0.0f;
    uint32_t regrown = 0;
};

static void free_buffers(RtMulti* m) {
    for (int b = 0; b < 2; ++b) {
        for (int i = 0; i < m->n; ++i) {
            if (!m->shard[b][i] && !m->gathered[b][i]) continue;
            (void)hipSetDevice(m->devices[i]);
            if (m->shard[b][i]) (void)hipFree(m->shard[b][i]);
            if (m->gathered[b][i]) (void)hipFree(m->gathered[b][i]);
            m->shard[b][i] = m->gathered[b][i] = nullptr;
        }
        if (m->frame[b]) { (void)hipSetDevice(m->devices[0]); (void)hipFree(m->frame[b]); m->frame[b] = nullptr; }
    }
    m->shard_view_bytes = 0;
    m->buffer_views = m->buffer_pixel_bytes = 0;
}

// Waits for everything the renderer has enqueued, on every device.  Never fails early: it is also what an error path
// calls before it returns, so that no launch is left running on buffers the caller may free.
static void sync_all(RtMulti* m) {
    for (int i = 0; i < m->n; ++i) {
        if (hipSetDevice(m->devices[i]) != hipSuccess) continue;
        if (m->streams[i]) (void)hipStreamSynchronize(m->streams[i]);
        if (m->xstreams[i]) (void)hipStreamSynchronize(m->xstreams[i]);
    }
}

extern "C" void rt_multi_destroy(RtMulti* m) {
    if (!m) return;
    for (Worker* w : m->workers) {
        if (!w) continue;
        if (w->thread.joinable()) {
            { std::lock_guard<std::mutex> hold(w->lock); w->job = Worker::QUIT; w->wake.notify_all(); }
            w->thread.join();
        }
        delete w;
    }
    sync_all(m);
    for (ncclComm_t c : m->comms)
        if (c) (void)m->rccl.CommDestroy(c);
    free_buffers(m);
    for (int i = 0; i < m->n; ++i) {
        (void)hipSetDevice(m->devices[i]);
        if (m->scenes[i]) rt_scene_destroy(m->scenes[i]);
        if (m->done[i]) (void)hipEventDestroy(m->done[i]);
        for (int b = 0; b < 2; ++b)
            if (m->taken[b][i]) (void)hipEventDestroy(m->taken[b][i]);
        if (m->streams[i]) (void)hipStreamDestroy(m->streams[i]);
        if (m->xstreams[i]) (void)hipStreamDestroy(m->xstreams[i]);
    }
    delete m;
}

extern "C" int rt_multi_create(const RtSceneDesc* desc, const int* device_ids, int n_devices, RtMulti** out) {
    if (!desc || !device_ids || !out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_create: null argument");
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_create: %d devices; between 1 and 64 are supported", n_devices);
    int n_visible = 0;
    if (hipGetDeviceCount(&n_visible) != hipSuccess || n_visible == 0) return fail(RT_ERR_NO_DEVICE, "no HIP device is visible");
    for (int i = 0; i < n_devices; ++i)
        if (device_ids[i] < 0 || device_ids[i] >= n_visible)
            return fail(RT_ERR_NO_DEVICE, "rt_multi_create: device %d of %d does not exist", device_ids[i], n_visible);
    bool distinct = true, same = true;
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j) {
            if (device_ids[i] == device_ids[j]) distinct = false; else same = false;
        }
    if (n_devices == 1) same = false;
    if (!distinct && !same)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_create: device_ids must be all different (or, to rehearse on one GPU, all the same)");
    RtMulti* m = new RtMulti();
    m->n = n_devices;
    m->same_device = same;
    m->devices.assign(device_ids, device_ids + n_devices);
    m->scenes.assign(n_devices, nullptr);
    m->streams.assign(n_devices, nullptr);
    m->xstreams.assign(n_devices, nullptr);
    m->done.assign(n_devices, nullptr);
    for (int b = 0; b < 2; ++b) {
        m->shard[b].assign(n_devices, nullptr);
        m->gathered[b].assign(n_devices, nullptr);
        m->taken[b].assign(n_devices, nullptr);
    }
    auto bail = [&](int code) { rt_multi_destroy(m); return code; };
    for (int i = 0; i < n_devices; ++i) {
        int rc = rt_scene_create(desc, device_ids[i], &m->scenes[i]);       // replicates the scene
        if (rc) return bail(rc);
        if (hipSetDevice(device_ids[i]) != hipSuccess || hipStreamCreateWithFlags(&m->streams[i], hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&m->xstreams[i], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&m->done[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&m->taken[0][i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&m->taken[1][i], hipEventDisableTiming) != hipSuccess)
            return bail(fail(RT_ERR_DEVICE, "rt_multi_create: stream / event creation failed on device %d", device_ids[i]));
    }
    if (n_devices > 1 && !m->same_device) {
        // one communicator per device, all created by this process (ncclCommInitAll).  A single device needs none - its
        // "exchange" is a copy - so a one-device renderer works without RCCL installed.
        int rc = load_rccl(&m->rccl);
        if (rc) return bail(rc);
        m->comms.assign(n_devices, nullptr);
        int e = m->rccl.CommInitAll(m->comms.data(), n_devices, device_ids);
        if (e != 0) {
            m->comms.clear();
            return bail(fail(RT_ERR_DEVICE, "ncclCommInitAll over %d devices: %s", n_devices, m->rccl.GetErrorString(e)));
        }
    }
    if (n_devices > 1) {
        m->workers.assign(n_devices, nullptr);
        for (int i = 0; i < n_devices; ++i) {
            Worker* w = new Worker();
            w->scene = m->scenes[i]; w->device = device_ids[i]; w->done = m->done[i];
            m->workers[i] = w;
            w->thread = std::thread([w] { w->run(); });
        }
    }
    *out = m;
    return RT_OK;
}

static int set_views(RtMulti* m, const float* screen_to_world, uint32_t n_views, uint32_t width, uint32_t height) {
    for (int i = 0; i < m->n; ++i) {
        int rc = rt_scene_set_views(m->scenes[i], screen_to_world, n_views, width, height);
        if (rc) return rc;
    }
    return RT_OK;
}

extern "C" int rt_multi_set_views(RtMulti* m, const float* screen_to_world, uint32_t n_views, uint32_t width, uint32_t height) {
    if (!m || !screen_to_world) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_set_views: null argument");
    int rc = set_views(m, screen_to_world, n_views, width, height);
    if (rc) return rc;
    m->views.assign(screen_to_world, screen_to_world + (size_t)16 * n_views);
    m->width = width; m->height = height; m->n_views = n_views;
    m->camera_set = true;
    return RT_OK;
}

extern "C" int rt_multi_set_camera(RtMulti* m, const float screen_to_world[16], uint32_t width, uint32_t height) {
    return rt_multi_set_views(m, screen_to_world, 1, width, height);
}

static uint32_t pixel_bytes(const RtRenderParams* p) { return p->output_format == RT_FORMAT_RGBA8 ? 4u : 3u; }

// Buffers for passes of up to `pass_views` views (the scene handles know the screen size by now).
static int ensure_buffers(RtMulti* m, const RtRenderParams* p, uint32_t pass_views) {
    RtRenderParams q = *p;
    q.shard_index = 0;
    q.shard_count = (uint32_t)m->n;
    uint64_t per_view = 0;
    int rc = rt_shard_bytes(m->scenes[0], &q, &per_view);
    if (rc) return rc;
    if (per_view == m->shard_view_bytes && pass_views <= m->buffer_views && pixel_bytes(p) == m->buffer_pixel_bytes) return RT_OK;
    sync_all(m);
    free_buffers(m);
    const size_t shard_bytes = (size_t)per_view * pass_views;
    for (int b = 0; b < 2; ++b) {
        for (int i = 0; i < m->n; ++i) {
            HIP_TRY(hipSetDevice(m->devices[i]));
            HIP_TRY(hipMalloc((void**)&m->shard[b][i], shard_bytes));
            // only the first device assembles, but an all-gather delivers to everyone
            HIP_TRY(hipMalloc((void**)&m->gathered[b][i], (m->comms.empty() && i > 0) ? 16 : shard_bytes * m->n));
        }
        HIP_TRY(hipSetDevice(m->devices[0]));
        HIP_TRY(hipMalloc((void**)&m->frame[b], (size_t)m->width * m->height * pixel_bytes(p) * pass_views));
    }
    m->shard_view_bytes = (size_t)per_view;
    m->buffer_views = pass_views;
    m->buffer_pixel_bytes = pixel_bytes(p);
    return RT_OK;
}

// Enqueues one pass - the views the scene handles are set to - on buffer set `b`: shards (every device's own thread
// issues its launches), exchange, stitching, copy of the frames to `out`.  Does not wait for any of it.
static int enqueue_pass(RtMulti* m, const RtRenderParams* p, int b, uint32_t pass_views, uint8_t* out) {
    const size_t shard_bytes = m->shard_view_bytes * pass_views;
    const size_t frame_bytes = (size_t)m->width * m->height * pixel_bytes(p) * pass_views;
    int rc = RT_OK;
    // every device renders its tiles, once the exchange of the pass before last has read this buffer set
    for (int i = 0; i < m->n; ++i) {
        HIP_TRY(hipSetDevice(m->devices[i]));
        HIP_TRY(hipStreamWaitEvent(m->streams[i], m->taken[b][i], 0));
    }
    for (int i = 0; i < m->n; ++i) {
        RtRenderParams q = *p;
        q.shard_index = (uint32_t)i;
        q.shard_count = (uint32_t)m->n;
        q.stream = m->streams[i];
        if (m->workers.empty()) {
            rc = rt_render_device(m->scenes[i], &q, m->shard[b][i]);
            if (rc == RT_OK) { HIP_TRY(hipSetDevice(m->devices[i])); HIP_TRY(hipEventRecord(m->done[i], m->streams[i])); }
        } else {
            m->workers[i]->post(q, m->shard[b][i]);
        }
    }
    for (int i = 0; i < m->n && !m->workers.empty(); ++i) {
        const int r = m->workers[i]->wait();                                 // every worker is waited for, whatever the others returned
        if (r != RT_OK && rc == RT_OK) { rc = r; (void)fail(r, "%s", m->workers[i]->error); }
    }
    if (rc) return rc;
    // the frame's one exchange step, behind the shards
    for (int i = 0; i < m->n; ++i) {
        HIP_TRY(hipSetDevice(m->devices[i]));
        HIP_TRY(hipStreamWaitEvent(m->xstreams[m->comms.empty() ? 0 : i], m->done[i], 0));
    }
    if (m->comms.empty()) {
        HIP_TRY(hipSetDevice(m->devices[0]));
        for (int i = 0; i < m->n; ++i)
            HIP_TRY(hipMemcpyAsync(m->gathered[b][0] + (size_t)i * shard_bytes, m->shard[b][i], shard_bytes, hipMemcpyDeviceToDevice, m->xstreams[0]));
        for (int i = 0; i < m->n; ++i) HIP_TRY(hipEventRecord(m->taken[b][i], m->xstreams[0]));
    } else {
        int e = m->rccl.GroupStart();
        hipError_t he = hipSuccess;
        for (int i = 0; i < m->n && e == 0 && he == hipSuccess; ++i) {
            he = hipSetDevice(m->devices[i]);
            if (he == hipSuccess) e = m->rccl.AllGather(m->shard[b][i], m->gathered[b][i], shard_bytes, kNcclUint8, m->comms[i], m->xstreams[i]);
        }
        const int e2 = m->rccl.GroupEnd();                                    // on every path: a group is never left open
        if (e == 0) e = e2;
        if (he != hipSuccess) return fail(RT_ERR_DEVICE, "hipSetDevice before ncclAllGather: %s", hipGetErrorString(he));
        if (e != 0) return fail(RT_ERR_DEVICE, "ncclAllGather of %zu bytes per device: %s", shard_bytes, m->rccl.GetErrorString(e));
        for (int i = 0; i < m->n; ++i) { HIP_TRY(hipSetDevice(m->devices[i])); HIP_TRY(hipEventRecord(m->taken[b][i], m->xstreams[i])); }
    }
    // the first device stitches the tiles into frames and hands them over
    RtRenderParams q = *p;
    q.shard_index = 0;
    q.shard_count = (uint32_t)m->n;
    q.stream = m->xstreams[0];
    if ((rc = rt_assemble_shards(m->scenes[0], &q, m->gathered[b][0], m->frame[b]))) return rc;
    HIP_TRY(hipSetDevice(m->devices[0]));
    HIP_TRY(hipMemcpyAsync(out, m->frame[b], frame_bytes, hipMemcpyDeviceToHost, m->xstreams[0]));
    return RT_OK;
}

// `n_total` views (matrices in `views`) in passes of at most `per_pass`, buffer sets alternating; everything is waited
// for before this returns, on success and on failure alike.
static int render_passes(RtMulti* m, const RtRenderParams* p, const float* views, uint32_t n_total, uint32_t per_pass, uint8_t* out_pixels) {
    if (p->stream) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render: the renderer owns its streams; params->stream must be NULL");
    if (p->shard_index != 0 || p->shard_count > 1)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render: the renderer deals the tiles itself; params->shard_index / shard_count must be unset (0, and 0 or 1)");
    const auto t0 = std::chrono::steady_clock::now();
    m->regrown = 0;
    const size_t view_bytes = (size_t)m->width * m->height * pixel_bytes(p);
    int rc = RT_OK;
    for (;;) {
        for (uint32_t first = 0, k = 0; first < n_total && rc == RT_OK; first += per_pass, ++k) {
            const uint32_t pass_views = n_total - first < per_pass ? n_total - first : per_pass;
            rc = set_views(m, views + (size_t)16 * first, pass_views, m->width, m->height);
            if (rc == RT_OK && k == 0) rc = ensure_buffers(m, p, pass_views);                  // the first pass is the largest
            if (rc == RT_OK) rc = enqueue_pass(m, p, (int)(k & 1), pass_views, out_pixels + view_bytes * first);
        }
        sync_all(m);
        if (rc) break;
        // a shard whose ray lists overflowed is incomplete: its handle has longer lists now, render everything again
        bool overflowed = false;
        uint64_t counters[RT_N_COUNTERS];
        for (int i = 0; i < m->n; ++i) {
            const int r = rt_last_ray_counts(m->scenes[i], counters);
            if (r == RT_ERR_OUT_OF_RANGE) overflowed = true;
            else if (r && rc == RT_OK) rc = r;
        }
        if (rc || !overflowed) break;
        if (++m->regrown > 16) { rc = fail(RT_ERR_OUT_OF_RANGE, "the ray lists keep overflowing"); break; }
    }
    // the handles go back to the views of rt_multi_set_views
    { const int r = set_views(m, m->views.data(), m->n_views, m->width, m->height); if (rc == RT_OK) rc = r; }
    m->last_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

extern "C" int rt_multi_render(RtMulti* m, const RtRenderParams* p, uint8_t* out_pixels) {
    if (!m || !p || !out_pixels) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render: null argument");
    if (!m->camera_set) return fail(RT_ERR_INVALID_ARGUMENT, "the camera has not been set (rt_multi_set_camera)");
    return render_passes(m, p, m->views.data(), m->n_views, m->n_views, out_pixels);
}

extern "C" int rt_multi_render_views(RtMulti* m, const RtRenderParams* p, const float* screen_to_world, uint32_t n_views,
                                     uint32_t views_per_pass, uint8_t* out_pixels) {
    if (!m || !p || !screen_to_world || !out_pixels) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render_views: null argument");
    if (!m->camera_set) return fail(RT_ERR_INVALID_ARGUMENT, "the screen size has not been set (rt_multi_set_camera)");
    if (n_views == 0) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render_views: no views");
    if (views_per_pass == 0 || views_per_pass > RT_MAX_VIEWS)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render_views: %u views per pass; between 1 and %d are supported", views_per_pass, RT_MAX_VIEWS);
    // (rt_multi_last_ray_counts afterwards reports the last pass)
    return render_passes(m, p, screen_to_world, n_views, views_per_pass, out_pixels);
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
