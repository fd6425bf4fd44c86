// Library identity, error strings and the deferred-reduction registry.
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace {

constexpr int MANY = 48;  // descriptors per batched launch (kernel-argument struct of 48 x 80 bytes + group starts: 4040 of the 4096-byte limit)

// Descriptors that add into the SAME destination form a group (kept in issue order): one block owns a
// 64-element chunk of the destination and runs the group's reductions one after the other, so the
// read-modify-writes of a shared accumulator never race and keep their order.
struct ReduceBatch {
    EqhReduceDesc d[MANY];
    int g_start[MANY + 1];  // group g = descriptors [g_start[g], g_start[g+1])
    int n_groups;
};
static_assert(sizeof(ReduceBatch) <= 4096, "ReduceBatch is passed by value: it must fit the 4 KB kernel-argument segment");

__global__ void __launch_bounds__(1024) eqh_k_reduce_many(ReduceBatch b) {
    __shared__ float s_part[1024];
    int g = 0;
    while (g + 1 < b.n_groups && (int)blockIdx.x >= b.d[b.g_start[g + 1]].first_block) ++g;  // block-uniform
    const int i0 = b.g_start[g], i1 = b.g_start[g + 1];
    const int64_t e0 = (int64_t)((int)blockIdx.x - b.d[i0].first_block) * 64;
    for (int i = i0; i < i1; ++i) eqh_reduce_chunk(b.d[i], e0, 1, s_part);
}

// "wide" reductions (weight-gradient slabs: a handful of slabs of 65 536 elements each): one float4 of the
// destination per thread, the group's slabs added one after the other -- the 64-column x 16-slab-group blocks
// of the kernel above left most of their threads idle on these (33 us for 16 MB; this takes a quarter).
__global__ void __launch_bounds__(256) eqh_k_reduce_wide(ReduceBatch b) {
    int g = 0;
    while (g + 1 < b.n_groups && (int)blockIdx.x >= b.d[b.g_start[g + 1]].first_block) ++g;  // block-uniform
    const int i0 = b.g_start[g], i1 = b.g_start[g + 1];
    const EqhReduceDesc& d0 = b.d[i0];
    const int64_t e = ((int64_t)((int)blockIdx.x - d0.first_block) * 256 + threadIdx.x) * 4;
    if (e >= d0.elems) return;
    float* dst = d0.out0 + (d0.row_len > 0 ? (e / d0.row_len) * d0.out_ld + (e % d0.row_len) : e);
    const float4 old = *reinterpret_cast<const float4*>(dst);      // requested first: it is added last
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    // eight slabs' loads in flight per thread (one at a time -- the loop as first written -- left 11 KB in flight per CU:
    // 1.3 TB/s for the 27 MB of a BASELINE step); the adds keep the slab order
    constexpr int DEEP = 8;
    for (int i = i0; i < i1; ++i) {
        const float* __restrict__ p = b.d[i].slab + e;
        const int n = b.d[i].n_slabs;
        for (int s0 = 0; s0 < n; s0 += DEEP) {
            float4 u[DEEP];
#pragma unroll
            for (int j = 0; j < DEEP; ++j)
                u[j] = s0 + j < n ? *reinterpret_cast<const float4*>(p + (int64_t)(s0 + j) * d0.elems) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < DEEP; ++j)
                if (s0 + j < n) { v.x += u[j].x; v.y += u[j].y; v.z += u[j].z; v.w += u[j].w; }
        }
    }
    *reinterpret_cast<float4*>(dst) = make_float4(old.x + v.x, old.y + v.y, old.z + v.z, old.w + v.w);
}

std::mutex g_mu;
std::unordered_map<hipStream_t, std::vector<EqhReduceDesc>> g_deferred;  // key present = deferral active

}  // namespace

bool eqh_defer_try(hipStream_t stream, const EqhReduceDesc& d) {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_deferred.find(stream);
    if (it == g_deferred.end()) return false;
    it->second.push_back(d);
    return true;
}

extern "C" int eqh_defer_begin(void* stream_) {
    std::lock_guard<std::mutex> lock(g_mu);
    g_deferred[static_cast<hipStream_t>(stream_)];  // creates the (empty) list
    return EQH_OK;
}

extern "C" int eqh_defer_flush(void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    std::vector<EqhReduceDesc> todo;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        auto it = g_deferred.find(stream);
        if (it == g_deferred.end()) return EQH_OK;
        todo.swap(it->second);
        g_deferred.erase(it);
    }
    // group by destination (stable: issue order inside a group); a reduction that shares a destination
    // pointer with a group of a different shape starts a new, later launch
    auto same = [](const EqhReduceDesc& x, const EqhReduceDesc& y) {
        return x.out0 == y.out0 && x.out1 == y.out1 && x.out2 == y.out2 && x.elems == y.elems && x.len0 == y.len0 &&
               x.len1 == y.len1 && x.out_ld == y.out_ld && x.row_len == y.row_len;
    };
    auto touches = [](const EqhReduceDesc& x, const EqhReduceDesc& y) {
        const float* px[3] = {x.out0, x.out1, x.out2};
        const float* py[3] = {y.out0, y.out1, y.out2};
        for (const float* a : px)
            for (const float* c : py)
                if (a != nullptr && a == c) return true;
        return false;
    };
    std::vector<std::vector<EqhReduceDesc>> groups;
    auto is_wide = [](const std::vector<EqhReduceDesc>& g) {
        const EqhReduceDesc& d = g[0];
        int slabs = 0;
        for (const EqhReduceDesc& x : g) slabs += x.n_slabs;
        return d.out1 == nullptr && d.elems >= 4096 && (d.elems & 3) == 0 && (d.row_len & 3) == 0 && (d.out_ld & 3) == 0 &&
               slabs <= 1024 && (((uintptr_t)d.out0 | (uintptr_t)d.slab) & 15) == 0;
    };
    auto launch_kind = [&](bool wide) -> int {
        size_t gi = 0;
        while (gi < groups.size()) {
            ReduceBatch b;
            int n = 0, blocks = 0;
            b.n_groups = 0;
            bool oversize = false;
            while (gi < groups.size()) {
                if (is_wide(groups[gi]) != wide || groups[gi].empty()) { ++gi; continue; }
                if ((int)groups[gi].size() > MANY) { oversize = true; break; }
                if (n + (int)groups[gi].size() > MANY) break;
                b.g_start[b.n_groups++] = n;
                const int per = wide ? 1024 : 64;
                const int chunks = (int)((groups[gi][0].elems + per - 1) / per);
                for (const EqhReduceDesc& d : groups[gi]) {
                    b.d[n] = d;
                    b.d[n].first_block = blocks;
                    ++n;
                }
                blocks += chunks;
                ++gi;
            }
            if (n == 0 && oversize) {  // one group larger than a batch: consecutive launches, in order
                std::vector<EqhReduceDesc>& g = groups[gi];
                b.n_groups = 1;
                b.g_start[0] = 0;
                n = MANY;
                for (int i = 0; i < MANY; ++i) { b.d[i] = g[i]; b.d[i].first_block = 0; }
                const int per = wide ? 1024 : 64;
                blocks = (int)((g[0].elems + per - 1) / per);
                g.erase(g.begin(), g.begin() + MANY);
            }
            b.g_start[b.n_groups] = n;
            if (n == 0 || blocks == 0) continue;
            if (wide)
                hipLaunchKernelGGL(eqh_k_reduce_wide, dim3(blocks), dim3(256), 0, stream, b);
            else
                hipLaunchKernelGGL(eqh_k_reduce_many, dim3(blocks), dim3(1024), 0, stream, b);
            if (hipGetLastError() != hipSuccess) return EQH_ERR_LAUNCH;
        }
        return EQH_OK;
    };
    auto launch_groups = [&]() -> int {
        if (launch_kind(false) != EQH_OK || launch_kind(true) != EQH_OK) return EQH_ERR_LAUNCH;
        groups.clear();
        return EQH_OK;
    };
    for (const EqhReduceDesc& d : todo) {
        bool placed = false, conflict = false;
        for (auto& g : groups) {
            if (same(g[0], d)) { g.push_back(d); placed = true; break; }
            if (touches(g[0], d)) conflict = true;
        }
        if (placed) continue;
        if (conflict && launch_groups() != EQH_OK) return EQH_ERR_LAUNCH;
        groups.push_back({d});
    }
    if (launch_groups() != EQH_OK) return EQH_ERR_LAUNCH;
    return EQH_OK;
}

// Device-side time stamps for measuring kernels INSIDE a replayed hipGraph (where HIP events on the launching
// stream see nothing): a one-thread kernel stores the constant-rate wall clock.  Two stamps around a launch give
// its in-graph duration plus one launch slot, which a back-to-back pair of stamps calibrates (bench.py).
namespace {
__global__ void eqh_k_stamp(unsigned long long* slot) { *slot = wall_clock64(); }
}  // namespace

extern "C" int eqh_stamp(void* slot, void* stream_) {
    if (!slot) return EQH_ERR_ARG;
    hipLaunchKernelGGL(eqh_k_stamp, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream_),
                       static_cast<unsigned long long*>(slot));
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

// Cross-stream trigger INSIDE a replayed graph: eqh_signal_post (a node of the step's hipGraph) bumps a device counter;
// eqh_signal_wait (launched eagerly on another stream) holds that stream until the counter has reached `target` -- the
// index build of the next batch starts when the running step has left its chip-filling front-end kernels, not at the step's
// head (an event cannot be recorded from the middle of a captured graph).  One lane polls with s_sleep between loads and
// leaves after timeout_us whatever happens, so a wait whose post never comes (a graph that was not launched) drains by
// itself; the caller enqueues the posting graph BEFORE the wait.
namespace {
__global__ void eqh_k_signal_post(int* counter) {
    __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void eqh_k_signal_wait(const int* counter, int target, unsigned long long timeout_ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((int)(__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) < 0 &&
           __builtin_amdgcn_s_memrealtime() - t0 < timeout_ticks)
        __builtin_amdgcn_s_sleep(32);
}
}  // namespace

extern "C" int eqh_signal_post(int32_t* counter, void* stream_) {
    if (!counter) return EQH_ERR_ARG;
    hipLaunchKernelGGL(eqh_k_signal_post, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream_), counter);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int eqh_signal_wait(const int32_t* counter, int32_t target, int32_t timeout_us, void* stream_) {
    if (!counter || timeout_us < 1 || timeout_us > 1000000) return EQH_ERR_ARG;
    int64_t khz = 100000;
    int dev = 0, k = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&k, hipDeviceAttributeWallClockRate, dev) == hipSuccess && k > 0)
        khz = k;
    hipLaunchKernelGGL(eqh_k_signal_wait, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream_), counter, (int)target,
                       (unsigned long long)(khz * timeout_us / 1000));
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

// Shader clock the chip holds right now: one wavefront brackets a ~spin_us wait on the constant-rate clock with the
// shader-cycle counter (s_memtime) and the 100 MHz real-time counter (s_memrealtime); out = {d cycles, d ticks}.
namespace {
__global__ void eqh_k_clock_probe(unsigned long long* out, unsigned long long spin_ticks) {
    if (threadIdx.x != 0) return;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < spin_ticks) r1 = __builtin_amdgcn_s_memrealtime();     // every wave leaves after spin_ticks
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    out[0] = c1 - c0;
    out[1] = r1 - r0;
}
}  // namespace

// dst[0 .. n) += src[0 .. n): a gradient computed into a fresh tensor handed to the parameter's persistent accumulator.  Inside a
// deferral window (eqh_defer_begin .. eqh_defer_flush on this stream) it rides the step's one batched reduction launch
// (a slab reduction over ONE slab); outside it is its own launch.  src must stay alive until the flush.
extern "C" int eqh_accumulate(const float* src, float* dst, int64_t n, void* stream_) {
    if (n < 0 || (n > 0 && (!src || !dst))) return EQH_ERR_ARG;
    if (n == 0) return EQH_OK;
    return eqh_reduce_slabs_async(src, 1, n, dst, static_cast<hipStream_t>(stream_), 1);
}

// Events for ordering two streams of ONE device (the trainer's index-prefetch stream and the step's): no timing and no
// system-scope fence -- a default event's release writes the L2 back and invalidates it when it is recorded, at the head of
// every training step in this use.
extern "C" int eqh_event_create(void** event) {
    if (!event) return EQH_ERR_ARG;
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) return EQH_ERR_ARG;
    *event = e;
    return EQH_OK;
}
extern "C" int eqh_event_record(void* event, void* stream_) {
    if (!event) return EQH_ERR_ARG;
    return hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream_)) == hipSuccess ? EQH_OK : EQH_ERR_ARG;
}
extern "C" int eqh_event_wait(void* event, void* stream_) {
    if (!event) return EQH_ERR_ARG;
    return hipStreamWaitEvent(static_cast<hipStream_t>(stream_), static_cast<hipEvent_t>(event), 0) == hipSuccess ? EQH_OK : EQH_ERR_ARG;
}
extern "C" int eqh_event_destroy(void* event) {
    if (!event) return EQH_OK;
    return hipEventDestroy(static_cast<hipEvent_t>(event)) == hipSuccess ? EQH_OK : EQH_ERR_ARG;
}

extern "C" int eqh_clock_probe(void* out, int32_t spin_us, void* stream_) {
    if (!out || spin_us < 1 || spin_us > 10000) return EQH_ERR_ARG;
    int64_t khz = 100000;
    int dev = 0, k = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&k, hipDeviceAttributeWallClockRate, dev) == hipSuccess && k > 0)
        khz = k;
    hipLaunchKernelGGL(eqh_k_clock_probe, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream_),
                       static_cast<unsigned long long*>(out), (unsigned long long)(khz * spin_us / 1000));
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int64_t eqh_wall_clock_khz(void) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return -1;
    return khz;
}

extern "C" int eqh_version(void) { return 2; }

extern "C" const char* eqh_error_string(int code) {
    switch (code) {
        case EQH_OK: return "ok";
        case EQH_ERR_ARG: return "invalid argument (null pointer, negative size or unsupported shape)";
        case EQH_ERR_ALIGN: return "row length not a multiple of 4 floats or pointer not 16-byte aligned";
        case EQH_ERR_RANGE: return "size exceeds int32 indexing";
        case EQH_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown error";
    }
}
