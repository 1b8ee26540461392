// Stand-in for <hip/hip_runtime.h> that lets g++ compile the library's .hip sources for the HOST, so that the kernels can be
// run on the CPU under AddressSanitizer (tests/emu/README.md).  Diagnostic use only: nothing in the package loads it.
//
// Execution model: a launch runs its blocks one after the other; the threads of a block are fibers (ucontext) that run until
// they reach a collective - __syncthreads, or a wave operation (__ballot, __shfl*, readfirstlane, wave_barrier) - and are
// resumed once every thread of the block / lane of the wave that has not returned has arrived at one.  That matches the
// GPU for code that calls its collectives in wave-uniform control flow, which these kernels do.  What it does not give is
// lockstep between two collectives: a lane may run ahead of its neighbours there, so code that hands data from lane to lane
// through LDS needs __builtin_amdgcn_wave_barrier() in between (on the GPU that is a scheduling barrier only).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __align__(n) __attribute__((aligned(n)))
#define __shared__ static                       // one copy per kernel: blocks run one after the other
#define EMU_DYNAMIC_LDS(type, name) type* name = reinterpret_cast<type*>(emu::dynamic_lds())

struct float4 { float x, y, z, w; };
struct uint2 { unsigned int x, y; };
struct dim3 { unsigned int x, y, z; dim3(unsigned int a = 1, unsigned int b = 1, unsigned int c = 1) : x(a), y(b), z(c) {} };
static inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
static inline uint2 make_uint2(unsigned int x, unsigned int y) { uint2 r; r.x = x; r.y = y; return r; }

namespace emu {
struct Dim { unsigned int x, y, z; };
extern Dim thread_idx, block_idx, block_dim, grid_dim;
void* dynamic_lds();
void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t lds_bytes);
void sync_block();
// wave collectives: every live lane contributes `mine` (bit pattern), then reads what it needs from the snapshot
const unsigned long long* wave_exchange(unsigned long long mine, unsigned long long* live_mask);
unsigned int lane();
}
#define threadIdx emu::thread_idx
#define blockIdx emu::block_idx
#define blockDim emu::block_dim
#define gridDim emu::grid_dim

static inline void __syncthreads() { emu::sync_block(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline void __builtin_amdgcn_wave_barrier() { unsigned long long live; (void)emu::wave_exchange(0, &live); }
static inline unsigned long long __ballot(int pred) {
    unsigned long long live;
    const unsigned long long* v = emu::wave_exchange(pred ? 1ull : 0ull, &live);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) if (((live >> l) & 1) && v[l]) m |= 1ull << l;
    return m;
}
template <class T> static inline T emu_from_bits(unsigned long long b) { T t; memcpy(&t, &b, sizeof(T)); return t; }
template <class T> static inline unsigned long long emu_bits(T t) { unsigned long long b = 0; static_assert(sizeof(T) <= 8, ""); memcpy(&b, &t, sizeof(T)); return b; }
template <class T> static inline T __shfl(T v, int src) {
    unsigned long long live;
    const unsigned long long* s = emu::wave_exchange(emu_bits(v), &live);
    src &= 63;
    return ((live >> src) & 1) ? emu_from_bits<T>(s[src]) : v;
}
template <class T> static inline T __shfl_up(T v, unsigned int delta) {
    unsigned long long live;
    const unsigned long long* s = emu::wave_exchange(emu_bits(v), &live);
    const unsigned int l = emu::lane();
    if (l < delta) return v;
    return ((live >> (l - delta)) & 1) ? emu_from_bits<T>(s[l - delta]) : v;
}
template <class T> static inline T __shfl_down(T v, unsigned int delta) {
    unsigned long long live;
    const unsigned long long* s = emu::wave_exchange(emu_bits(v), &live);
    const unsigned int l = emu::lane();
    if (l + delta >= 64) return v;
    return ((live >> (l + delta)) & 1) ? emu_from_bits<T>(s[l + delta]) : v;
}
static inline int __builtin_amdgcn_readfirstlane(int v) {           // (signed, as the compiler's builtin is)
    unsigned long long live;
    const unsigned long long* s = emu::wave_exchange((unsigned int)v, &live);
    return (int)(unsigned int)s[__builtin_ctzll(live)];
}
// DPP operand: the lane reads `v` of another lane of its row.  The controls the kernels use: quad permutes (0x00-0xFF, two
// bits per lane of the quad), row_mirror (0x140), row_half_mirror (0x141).
static inline int __builtin_amdgcn_update_dpp(int old, int v, int ctrl, int, int, bool bound_ctrl) {
    unsigned long long live;
    const unsigned long long* s = emu::wave_exchange((unsigned int)v, &live);
    const unsigned int l = emu::lane();
    unsigned int src;
    if (ctrl >= 0 && ctrl < 0x100) src = (l & ~3u) | (((unsigned int)ctrl >> (2 * (l & 3u))) & 3u);
    else if (ctrl == 0x140) src = (l & ~15u) | (15u - (l & 15u));
    else if (ctrl == 0x141) src = (l & ~7u) | (7u - (l & 7u));
    else abort();
    if ((live >> src) & 1) return (int)(unsigned int)s[src];
    return bound_ctrl ? 0 : old;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline unsigned int __float_as_uint(float f) { unsigned int u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned int u) { float f; memcpy(&f, &u, 4); return f; }
static inline long long __double2ll_rn(double d) { return llrint(d); }
static inline long long clock64() { return 0; }

// atomics: the fibers of a launch run on one OS thread, never inside one of these
template <class T, class U> static inline T atomicAdd(T* p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class U> static inline T atomicSub(T* p, U v) { T o = *p; *p = (T)(o - (T)v); return o; }
template <class T, class U> static inline T atomicMin(T* p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class U> static inline T atomicOr(T* p, U v) { T o = *p; *p = (T)(o | (T)v); return o; }

// ---- host API ------------------------------------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotReady = 600 };
typedef struct emuStream* hipStream_t;
typedef struct emuEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipHostMallocMapped = 2, hipEventDisableTiming = 2 };
struct hipDeviceProp_t { int multiProcessorCount; char name[64]; };

hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipMalloc(void** p, size_t bytes);
template <class T> static inline hipError_t hipMalloc(T** p, size_t bytes) { return hipMalloc((void**)p, bytes); }
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned int flags);
hipError_t hipHostFree(void* p);
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned int flags);
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind k, hipStream_t s);
hipError_t hipMemset(void* p, int v, size_t n);
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned int flags);
hipError_t hipDeviceSynchronize();
hipError_t hipMemGetInfo(size_t* free_bytes, size_t* total_bytes);
// scoped atomic loads: the emulated device is one thread at a time, any scope is the same load
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(ptr, order, scope) __atomic_load_n(ptr, order)
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned int flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
#define HIP_SYMBOL(x) x
template <class T> static inline hipError_t hipGetSymbolAddress(void** p, T& symbol) { *p = (void*)&symbol; return hipSuccess; }
template <class T> static inline hipError_t hipMemcpyFromSymbol(void* dst, const T& symbol, size_t n) { memcpy(dst, &symbol, n); return hipSuccess; }
template <class F> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 2; return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, ...) \
    emu::launch([&]() { kernel(__VA_ARGS__); }, dim3(grid), dim3(block), (size_t)(lds_bytes))
