// Internal header of liblvi_hip.so: HIP error handling, launch context with HIP-event
// kernel timing, and the wave64 / workgroup primitives every kernel file uses.
// gfx950 only: wavefront = 64 lanes, ballots are 64-bit.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lvi_hotpath.h"

namespace lvi {

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
void set_error(const std::string& s);                  // thread-local, read by lvi_last_error()

struct HipError { hipError_t e; const char* what; const char* file; int line; };

#define LVI_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t _e = (call);                                                               \
        if (_e != hipSuccess) throw ::lvi::HipError{_e, #call, __FILE__, __LINE__};           \
    } while (0)

// one record per profiled launch
struct ProfRec { int name_id; double bytes; hipEvent_t a, b; };

struct Profiler {
    bool on = false;
    std::vector<std::string> names;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    // accumulated
    std::vector<int64_t> launches;
    std::vector<double> total_ms, bytes;

    int name_id(const char* n);
    hipEvent_t get_event();
    void collect();          // sync + fold recs into the accumulators
    void reset();
    ~Profiler();
};

struct Ctx {
    hipStream_t stream = nullptr;
    Profiler* prof = nullptr;
};

// RAII bracket around one kernel launch: records HIP events on the launch stream when
// profiling is on (this is what bench.py's roofline numbers come from).
struct LaunchScope {
    const Ctx& c; int id = -1; double bytes; hipEvent_t a = nullptr, b = nullptr;
    LaunchScope(const Ctx& ctx, const char* name, double bytes_alg) : c(ctx), bytes(bytes_alg)
    {
        if (c.prof && c.prof->on) {
            id = c.prof->name_id(name);
            a = c.prof->get_event(); b = c.prof->get_event();
            (void)hipEventRecord(a, c.stream);
        }
    }
    ~LaunchScope()
    {
        if (id >= 0) { (void)hipEventRecord(b, c.stream); c.prof->recs.push_back(ProfRec{id, bytes, a, b}); }
    }
};
#define LVI_LAUNCH(ctx, name, bytes, ...)                       \
    do {                                                        \
        ::lvi::LaunchScope _ls(ctx, name, (double)(bytes));     \
        __VA_ARGS__;                                            \
        LVI_HIP(hipGetLastError());                             \
    } while (0)

// bump allocator over one hipMalloc (all buffers live as long as the handle)
struct Arena {
    char* base = nullptr; size_t size = 0, used = 0;
    void init(size_t bytes) { LVI_HIP(hipMalloc((void**)&base, bytes)); size = bytes; used = 0; }
    template <class T> T* alloc(size_t n)
    {
        size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
        if (used + bytes > size) throw HipError{hipErrorOutOfMemory, "arena exhausted", __FILE__, __LINE__};
        T* p = reinterpret_cast<T*>(base + used); used += bytes; return p;
    }
    void release() { if (base) (void)hipFree(base); base = nullptr; }
};
// sizing pass: same calls, no memory
struct ArenaSizer {
    size_t used = 0;
    template <class T> T* alloc(size_t n) { used += (n * sizeof(T) + 255) & ~size_t(255); return nullptr; }
};

inline int div_up(int a, int b) { return (a + b - 1) / b; }

// Batched launches: one launch sequence serves up to LVI_MAX_BATCH independent scans (SURVEY §7.3-6 i).  Every kernel
// of the lidar path takes the argument blocks of all scans of the batch and picks its own by blockIdx.z, so a batch of S
// scans costs the launches of one scan, and the one-workgroup kernels of the chain (scans, solves) run S workgroups wide.
constexpr int MAX_BATCH = LVI_MAX_BATCH;
template <class T> struct Batch { T a[MAX_BATCH]; };
inline dim3 zdim(dim3 g, int S) { g.z = (unsigned)S; return g; }

#ifdef __HIPCC__
// ---------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------
#define LVI_WAVE 64

// a pointer read from a device-side record has no known address space: loads through it compile to flat_load (LDS-aperture test,
// counted on lgkmcnt together with the LDS traffic).  The arrays of this library live in global memory: load as such.
typedef float lvi_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ lvi_pt ld_global_pt(const lvi_pt* p)
{
    const lvi_f4 v = *(const __attribute__((address_space(1))) lvi_f4*)p;
    lvi_pt o; o.x = v.x; o.y = v.y; o.z = v.z; o.intensity = v.w;
    return o;
}
__device__ __forceinline__ uint8_t ld_global_u8(const uint8_t* p) { return *(const __attribute__((address_space(1))) uint8_t*)p; }
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

template <class T>
__device__ __forceinline__ T wave_incl_scan(T v)
{
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { T t = __shfl_up(v, o, 64); if (l >= o) v += t; }
    return v;
}
template <class T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// 64-bit integer sum over the wavefront through DPP moves (quad swaps, half-row and row mirrors, row broadcasts) instead of
// twelve ds_bpermute round trips: every lane receives the total.  Integer addition commutes: the same value as wave_sum.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_move_i64(long long v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL, ROW_MASK, 0xF, false);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ long long wave_sum_i64(long long v)
{
    v += dpp_move_i64<0xB1, 0xF>(v);              // quad_perm [1,0,3,2]
    v += dpp_move_i64<0x4E, 0xF>(v);              // quad_perm [2,3,0,1]: every lane holds its quad's sum
    v += dpp_move_i64<0x141, 0xF>(v);             // row_half_mirror: ... its half row's
    v += dpp_move_i64<0x140, 0xF>(v);             // row_mirror: ... its row's (16 lanes)
    v += dpp_move_i64<0x142, 0xA>(v);             // row_bcast15 into rows 1 and 3
    v += dpp_move_i64<0x143, 0xC>(v);             // row_bcast31 into rows 2 and 3: lane 63 holds the total
    const int lo = __builtin_amdgcn_readlane((int)v, 63), hi = __builtin_amdgcn_readlane((int)(v >> 32), 63);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// workgroup exclusive scan of one int per thread; BLOCK threads (multiple of 64, <= 1024).
// `ws` = shared scratch of BLOCK/64 + 1 ints.  Returns the exclusive prefix; *total = sum.
template <int BLOCK>
__device__ __forceinline__ int block_excl_scan(int v, int* ws, int* total)
{
    constexpr int NW = BLOCK / 64;
    const int incl = wave_incl_scan(v);
    if (lane_id() == 63) ws[wave_id()] = incl;
    __syncthreads();
    if (threadIdx.x < 64) {
        int w = (threadIdx.x < NW) ? ws[threadIdx.x] : 0;
        int wi = wave_incl_scan(w);
        if (threadIdx.x < NW) ws[threadIdx.x] = wi - w;
        if (threadIdx.x == NW - 1) ws[NW] = wi;
    }
    __syncthreads();
    const int r = ws[wave_id()] + incl - v;
    if (total) *total = ws[NW];
    __syncthreads();
    return r;
}

// the same for 64-bit words (several counters packed into one word are scanned together: one pass of barriers instead of one per counter)
template <int BLOCK>
__device__ __forceinline__ unsigned long long block_excl_scan_u64(unsigned long long v, unsigned long long* ws, unsigned long long* total)
{
    constexpr int NW = BLOCK / 64;
    const unsigned long long incl = wave_incl_scan(v);
    if (lane_id() == 63) ws[wave_id()] = incl;
    __syncthreads();
    if (threadIdx.x < 64) {
        unsigned long long w = (threadIdx.x < NW) ? ws[threadIdx.x] : 0ull;
        unsigned long long wi = wave_incl_scan(w);
        if (threadIdx.x < NW) ws[threadIdx.x] = wi - w;
        if (threadIdx.x == NW - 1) ws[NW] = wi;
    }
    __syncthreads();
    const unsigned long long r = ws[wave_id()] + incl - v;
    if (total) *total = ws[NW];
    __syncthreads();
    return r;
}

// order-preserving float <-> uint encoding for atomicMin/atomicMax on floats
__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// f32 arithmetic that must match the CPU restatement bit for bit.  On ROCm 7.2 the CUDA-style
// __fmul_rn/__fadd_rn are plain operators and __fsqrt_rn is the NATIVE (1-ulp) square root, so
// none of them gives any guarantee: exactness comes from compiling the library with
// -ffp-contract=off (no FMA contraction; plain * + - are then IEEE round-to-nearest) and from
// sqrtf() and '/' being correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }
__device__ __forceinline__ float div_rn(float a, float b) { return a / b; }
__device__ __forceinline__ float sqrt_rn(float a) { return sqrtf(a); }
#endif  // __HIPCC__

}  // namespace lvi
