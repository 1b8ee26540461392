// Runtime of the CPU stand-in for HIP (see hip/hip_runtime.h in this directory): fibers, collectives, a trivial host API.
#include <hip/hip_runtime.h>

#include <sys/mman.h>
#include <ucontext.h>

#include <chrono>
#include <cstdio>
#include <vector>

#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/common_interface_defs.h>
#define EMU_ASAN 1
#else
#define EMU_ASAN 0
#endif

namespace emu {

Dim thread_idx, block_idx, block_dim, grid_dim;

namespace {
constexpr size_t kStackBytes = 256 * 1024;
constexpr unsigned kWave = 64;

enum State { READY, WAIT_WAVE, WAIT_BLOCK, DONE };

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    State state = DONE;
    unsigned tid = 0;
    unsigned long long wave_gen = 0;     // generation of the wave collective this fiber waits for
    unsigned long long block_gen = 0;
    void* fake_stack = nullptr;          // AddressSanitizer's bookkeeping for this fiber's stack while it is switched out
};

struct Wave {
    unsigned long long vals[kWave];
    unsigned long long snap[kWave];
    unsigned long long live = 0, arrived = 0, snap_live = 0;
    unsigned long long gen = 0;
};

std::vector<Fiber> fibers;
std::vector<Wave> waves;
ucontext_t scheduler_ctx;
Fiber* current = nullptr;
const std::function<void()>* body = nullptr;
unsigned block_live = 0, block_arrived = 0;
unsigned long long block_gen = 0;
std::vector<unsigned char> lds_buffer;

void* scheduler_fake_stack = nullptr;
const void* scheduler_stack_bottom = nullptr;      // the thread's own stack, where the scheduler runs
size_t scheduler_stack_size = 0;

void switch_to_scheduler(bool dying) {
    Fiber* f = current;
#if EMU_ASAN
    __sanitizer_start_switch_fiber(dying ? nullptr : &f->fake_stack, scheduler_stack_bottom, scheduler_stack_size);
#endif
    swapcontext(&f->ctx, &scheduler_ctx);
#if EMU_ASAN
    __sanitizer_finish_switch_fiber(f->fake_stack, &scheduler_stack_bottom, &scheduler_stack_size);
#endif
    (void)dying;
}

void complete_wave(Wave& w) {
    memcpy(w.snap, w.vals, sizeof(w.vals));
    w.snap_live = w.live;
    w.arrived = 0;
    ++w.gen;
}

void fiber_main() {
#if EMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &scheduler_stack_bottom, &scheduler_stack_size);
#endif
    (*body)();
    Fiber* f = current;
    f->state = DONE;
    Wave& w = waves[f->tid / kWave];
    w.live &= ~(1ull << (f->tid % kWave));
    if (w.live && (w.arrived & w.live) == w.live) complete_wave(w);       // the others were only waiting for this lane
    --block_live;
    if (block_live && block_arrived == block_live) { block_arrived = 0; ++block_gen; }
    switch_to_scheduler(true);
    abort();                                                               // a finished fiber is never resumed
}

}  // namespace

unsigned int lane() { return current->tid % kWave; }
void* dynamic_lds() { return lds_buffer.data(); }

void sync_block() {
    Fiber* f = current;
    const unsigned long long gen = block_gen;
    if (++block_arrived == block_live) { block_arrived = 0; ++block_gen; return; }
    f->state = WAIT_BLOCK;
    f->block_gen = gen;
    switch_to_scheduler(false);
}

const unsigned long long* wave_exchange(unsigned long long mine, unsigned long long* live_mask) {
    Fiber* f = current;
    Wave& w = waves[f->tid / kWave];
    const unsigned l = f->tid % kWave;
    w.vals[l] = mine;
    w.arrived |= 1ull << l;
    if ((w.arrived & w.live) == w.live) {
        complete_wave(w);
    } else {
        f->state = WAIT_WAVE;
        f->wave_gen = w.gen;
        switch_to_scheduler(false);
    }
    *live_mask = w.snap_live;
    return w.snap;
}

void launch(const std::function<void()>& fn, dim3 grid, dim3 block, size_t lds_bytes) {
    const unsigned n = block.x * block.y * block.z;
    if (fibers.size() < n) {
        const size_t old = fibers.size();
        fibers.resize(n);
        for (size_t i = old; i < n; ++i) {
            fibers[i].stack = (char*)mmap(nullptr, kStackBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (fibers[i].stack == MAP_FAILED) { perror("mmap"); abort(); }
        }
    }
    waves.resize((n + kWave - 1) / kWave);
    lds_buffer.assign(lds_bytes + 64, 0xCD);        // fresh per launch; every block finds garbage, as on the GPU
    body = &fn;
    grid_dim = {grid.x, grid.y, grid.z};
    block_dim = {block.x, block.y, block.z};
    for (unsigned b = 0; b < grid.x * grid.y * grid.z; ++b) {
        block_idx = {b % grid.x, (b / grid.x) % grid.y, b / (grid.x * grid.y)};
        block_live = n; block_arrived = 0;
        for (Wave& w : waves) { w.live = 0; w.arrived = 0; }
        for (unsigned t = 0; t < n; ++t) {
            Fiber& f = fibers[t];
            f.tid = t; f.state = READY;
            waves[t / kWave].live |= 1ull << (t % kWave);
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = kStackBytes;
            f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, fiber_main, 0);
        }
        unsigned long long idle_rounds = 0;
        while (block_live) {
            bool progressed = false;
            for (unsigned t = 0; t < n; ++t) {
                Fiber& f = fibers[t];
                if (f.state == DONE) continue;
                if (f.state == WAIT_WAVE && waves[t / kWave].gen == f.wave_gen) continue;
                if (f.state == WAIT_BLOCK && block_gen == f.block_gen) continue;
                f.state = READY;
                current = &f;
                thread_idx = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
#if EMU_ASAN
                __sanitizer_start_switch_fiber(&scheduler_fake_stack, f.stack, kStackBytes);
#endif
                swapcontext(&scheduler_ctx, &f.ctx);
#if EMU_ASAN
                __sanitizer_finish_switch_fiber(scheduler_fake_stack, nullptr, nullptr);
#endif
                progressed = true;
            }
            if (!progressed && ++idle_rounds > 2) {
                fprintf(stderr, "emu: deadlock in block %u: %u threads alive, %u at the block barrier;", b, block_live, block_arrived);
                for (size_t wv = 0; wv < waves.size(); ++wv)
                    fprintf(stderr, " wave %zu live %016llx arrived %016llx;", wv, waves[wv].live, waves[wv].arrived);
                fprintf(stderr, "\n  (a collective was not reached by every live thread: wave operations must sit in wave-uniform control flow)\n");
                abort();
            }
        }
    }
    current = nullptr;
}

}  // namespace emu

// ---- host API ------------------------------------------------------------------------------------------------------
struct emuStream { int unused; };
struct emuEvent { std::chrono::steady_clock::time_point t; };

static int emu_cus() { const char* e = getenv("EMU_CUS"); int v = e ? atoi(e) : 2; return v > 0 ? v : 2; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = emu_cus(); strcpy(p->name, "cpu emulation"); return hipSuccess; }
hipError_t hipMalloc(void** p, size_t bytes) { *p = malloc(bytes ? bytes : 1); if (*p) memset(*p, 0xCD, bytes); return *p ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned int) { *p = calloc(1, bytes ? bytes : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned int) { *dev = host; return hipSuccess; }
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind) { memmove(dst, src, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) { memmove(dst, src, n); return hipSuccess; }
hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) { *s = new emuStream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned int) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipMemGetInfo(size_t* free_bytes, size_t* total_bytes) { *free_bytes = *total_bytes = (size_t)4 << 30; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emuEvent(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned int) { *e = new emuEvent(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count() + 1e-3f;
    return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "error"; }
