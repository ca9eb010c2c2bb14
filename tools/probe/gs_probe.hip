// gs_probe.hip -- libgsplat_probe.so: tuning probes, built from tools/ and NOT part of libgsplat_hip.so (whose exports
// are exactly what include/gsplat.h declares).  gs_sync_probe / gs_atomic_probe / gs_lds_probe price synchronisation,
// global atomics and LDS histograms on this chip; gs_debug_render_stats re-runs RenderGaussians
// of a context's last frame with per-tile counters.  The latter reads the context behind a gs_ctx handle of
// libgsplat_hip.so through the internal header csrc/gs_ctx.h (same tree, same build) and runs its own copy of the render
// kernels, compiled here with the counters switched on.
//
//     make -C tools/probe        (hipcc --offload-arch=gfx950)
#define GS_RENDER_STATS 1
#include "../../vk3dgaussiansplatting_amd/csrc/gs_render.hip"
#include "../../vk3dgaussiansplatting_amd/csrc/gs_ctx.h"

#include <string>

namespace gs {

// ---- hand-off probe (tuning only, gs_sync_probe): what does it cost to hand a small list from one radix stage to the
//      next -- a kernel boundary per stage, or one persistent launch with a device-wide barrier per stage?  One step =
//      every workgroup reads `dwords` 16-byte words of ITS slice of `a` written in the previous step by ANOTHER
//      workgroup (the neighbour: a real cross-workgroup dependency, like a pass reading what the previous pass
//      scattered) and writes its slice of `b`; a and b swap every step.
__device__ __forceinline__ void probe_step(const uint4* src, uint4* dst, uint32_t wg, uint32_t wgs,
                                           uint32_t per_wg, uint32_t step) {
    const uint32_t from = (wg + 1u) % wgs;                 // the neighbour's slice
    for (uint32_t i = threadIdx.x; i < per_wg; i += blockDim.x) {
        uint4 v = src[(size_t)from * per_wg + i];
        v.x += step; v.y ^= wg;
        dst[(size_t)wg * per_wg + i] = v;
    }
}

__global__ __launch_bounds__(256) void k_probe_step(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t per_wg, uint32_t step) {
    probe_step(src, dst, blockIdx.x, gridDim.x, per_wg, step);
}

// One launch, `steps` steps, a counter barrier between them: every storing wave drains its stores, the workgroup meets,
// one lane releases (agent scope: the XCD's L2 writes its dirty lines back), arrives, polls the counter with relaxed
// agent-scope loads and a sleep, the workgroup meets again and every wave acquires (this CU's L1 is invalidated).  Every spin is
// bounded: a grid that is not wholly resident gives up, flags it and still terminates.
__global__ __launch_bounds__(256) void k_probe_persistent(uint4* a, uint4* b, uint32_t per_wg, uint32_t steps,
                                                          uint32_t* counter, uint32_t* timed_out) {
    uint4* src = a;
    uint4* dst = b;
    for (uint32_t s = 0; s < steps; ++s) {
        probe_step(src, dst, blockIdx.x, gridDim.x, per_wg, s);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            (void)__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t want = (s + 1u) * gridDim.x;
            uint32_t budget = 400000u;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && --budget) __builtin_amdgcn_s_sleep(2);
            if (budget == 0u) *timed_out = 1u;
        }
        __syncthreads();
        // EVERY wave acquires (its CU's vector L1 may hold last step's lines of the slice it is about to read): the
        // barrier a real pass hand-off needs, not only lane 0's
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        uint4* t = src; src = dst; dst = t;
    }
}

// ---- atomic-rate probe (tuning only, gs_atomic_probe): could a radix Scatter feed the NEXT pass's per-group digit counts with
//      global atomics instead of a Count launch?  Workgroup w issues `lines` wave instructions of 16 active lanes, each a
//      non-returning agent-scope add to the 16 consecutive counters of row (w * stride_num / stride_den + k) % rows: with
//      stride 1/16 sixteen neighbouring workgroups meet on a row, as neighbouring source groups of a pass meet on a destination group.
__global__ __launch_bounds__(256) void k_probe_atomics(uint32_t* __restrict__ table, uint32_t rows, uint32_t lines, uint32_t stride_num,
                                                       uint32_t stride_den, uint32_t add) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane >= 16u) return;
    const uint32_t first = (uint32_t)(((uint64_t)blockIdx.x * stride_num) / stride_den);
    for (uint32_t k = wave; k < lines; k += 4u) {
        const uint32_t row = (first + k * 97u) % rows;      // the 16 digits of a group land in 16 runs, i.e. rows far apart
        if (add) (void)__hip_atomic_fetch_add(&table[(size_t)row * 16u + lane], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else table[(size_t)row * 16u + lane] = k;            // the same addresses with plain stores, for reference
    }
}
void launch_probe_atomics(uint32_t* table, uint32_t rows, uint32_t workgroups, uint32_t lines, uint32_t stride_num, uint32_t stride_den,
                          uint32_t add, hipStream_t stream) {
    hipLaunchKernelGGL(k_probe_atomics, dim3(workgroups), dim3(256), 0, stream, table, rows, lines, stride_num, stride_den, add);
}

// ---- How fast are 256-bin histograms in LDS?  (tuning only: the Count of the 8-bit sorter.)  1024 workgroups of 4 waves;
//      every lane makes `reps` x 32 updates with digits from a register generator -- no memory loads -- in one of the forms
//      below, then the counters are summed into out[] so that nothing is optimised away.
//      kind 0 no LDS (generator only) | 1 ds_add, random digit, one histogram per wave | 2 ds_add, address = lane |
//      3 ds_add, three digits | 4 plain read-modify-write of the lane's own column of packed byte counters |
//      5 eight ballots + one ds_add per digit present in the round | 6 like 1 with the returning form
__global__ __launch_bounds__(256) void k_probe_lds(uint32_t* __restrict__ out, uint32_t kind, uint32_t reps) {
    __shared__ uint32_t s_h[4][4096];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t* h = s_h[wave];
    for (uint32_t i = lane; i < 4096u; i += 64u) h[i] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t x = (blockIdx.x * 256u + tid) * 2654435761u + 12345u, acc = 0u;
    for (uint32_t r = 0; r < reps; ++r) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            x = x * 1664525u + 1013904223u;
            const uint32_t d = x >> 24;
            if (kind == 0u) acc ^= d;
            else if (kind == 1u) (void)__hip_atomic_fetch_add(&h[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (kind == 2u) (void)__hip_atomic_fetch_add(&h[lane], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (kind == 3u) (void)__hip_atomic_fetch_add(&h[d % 3u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (kind == 4u) { uint32_t* p = &h[(d >> 2) * 64u + lane]; *p = *p + (1u << (8u * (d & 3u))); }
            else if (kind == 5u) {
                uint32_t m_lo = 0xFFFFFFFFu, m_hi = 0xFFFFFFFFu;
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const int32_t sbit = __builtin_amdgcn_sbfe((int32_t)d, (uint32_t)b, 1u);
                    const uint64_t bal = __ballot(sbit != 0);
                    m_lo &= ~((uint32_t)bal ^ (uint32_t)sbit);
                    m_hi &= ~((uint32_t)(bal >> 32) ^ (uint32_t)sbit);
                }
                const uint64_t same = ((uint64_t)m_hi << 32) | m_lo;
                if (mbcnt(same) == 0u) (void)__hip_atomic_fetch_add(&h[d], (uint32_t)__popcll(same), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else acc += __hip_atomic_fetch_add(&h[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < 4096u; i += 64u) acc += h[i];
    out[blockIdx.x * 256u + tid] = acc;
}
void launch_probe_lds(uint32_t* out, uint32_t kind, uint32_t reps, hipStream_t stream) {
    hipLaunchKernelGGL(k_probe_lds, dim3(1024), dim3(256), 0, stream, out, kind, reps);
}

void launch_probe_step(const void* src, void* dst, uint32_t workgroups, uint32_t per_wg, uint32_t step, hipStream_t stream) {
    hipLaunchKernelGGL(k_probe_step, dim3(workgroups), dim3(256), 0, stream, (const uint4*)src, (uint4*)dst, per_wg, step);
}
void launch_probe_persistent(void* a, void* b, uint32_t workgroups, uint32_t per_wg, uint32_t steps, uint32_t* counter,
                             uint32_t* timed_out, hipStream_t stream) {
    hipLaunchKernelGGL(k_probe_persistent, dim3(workgroups), dim3(256), 0, stream, (uint4*)a, (uint4*)b, per_wg, steps, counter, timed_out);
}


// ---- LDS poison (tools/soak.py): LDS is not cleared between workgroups, so a kernel that reads a slot it never wrote
//      sees whatever the previous tenant left there -- usually a plausible float of the same kernel, which hides the bug.
//      Every CU gets workgroups that fill the 64 KB a static allocation may take with `pattern` (NaNs, all ones).
__global__ __launch_bounds__(256) void k_lds_poison(uint32_t pattern, uint32_t* __restrict__ sink) {
    __shared__ uint32_t s_all[16384];
    for (uint32_t i = threadIdx.x; i < 16384u; i += 256u) s_all[i] = pattern;
    __syncthreads();
    if (s_all[(threadIdx.x * 61u) & 16383u] != pattern) sink[0] = 1u;      // keeps the stores alive
}

} // namespace gs

using namespace gs;

namespace {
int fail(gs_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->last_error = msg;
    return code;
}
#define HIP_TRY(ctx, expr)                                                                       \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return fail((ctx), GS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)
}

extern "C" {

// Re-runs RenderGaussians of the last frame with per-tile
// counters; out = uint32[tiles][8] {list length, splats visited, splats needing exp, clock ticks, entries staged, 0, 0, 0}.
int gs_debug_render_stats(gs_ctx* c, const float* view, const float* proj, const float* cam_pos, uint32_t* out) {
    if (!c || !out || !c->have_frame) return GS_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    const FrameParams fp = make_frame_params(c, view, proj, cam_pos, 0);
    const size_t tiles = (size_t)c->grid_w * c->grid_h;
    uint4* d = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d, tiles * 2 * sizeof(uint4)));
    HIP_TRY(c, hipMemsetAsync(d, 0, tiles * 2 * sizeof(uint4), c->stream));
    launch_render_stats(fp, c->scratch.raster, c->sort.id[c->sorted_index], c->ranges, c->framebuffer, d, c->stream);
    hipError_t e = hipMemcpyAsync(out, d, tiles * 2 * sizeof(uint4), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, GS_ERR_HIP, std::string("gs_debug_render_stats: ") + hipGetErrorString(e));
    return GS_OK;
}

// Tuning only (not declared in gsplat.h): microseconds per dependent step of `workgroups` x 256 threads that each move
// bytes_per_wg bytes written by another workgroup in the step before -- persistent = 0: one kernel launch per step (a
// hipGraph replay of `steps` launches, as the frame replays its radix passes); persistent = 1: ONE launch with a
// device-wide counter barrier (release / acquire at agent scope) between the steps.  timed_out = 1 when the persistent
// grid was not wholly resident and a bounded spin gave up (the number is then meaningless).
int gs_sync_probe(gs_ctx* c, int persistent, uint32_t workgroups, uint32_t steps, uint32_t bytes_per_wg, uint32_t iters,
                  float* us_per_step, uint32_t* timed_out) {
    if (!c || !us_per_step || !timed_out || workgroups == 0 || steps == 0 || iters == 0) return GS_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    const uint32_t per_wg = bytes_per_wg / 16u;
    const size_t bytes = std::max<size_t>(16, (size_t)workgroups * per_wg * 16);
    void *a = nullptr, *b = nullptr;
    uint32_t* flags = nullptr;             // [0] barrier counter, [1] timed out
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipMalloc(&a, bytes);
    if (e == hipSuccess) e = hipMalloc(&b, bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&flags, 8);
    if (e == hipSuccess) e = hipMemsetAsync(a, 1, bytes, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(b, 2, bytes, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(flags, 0, 8, c->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess && !persistent) {
        hipGraph_t graph = nullptr;
        e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed);
        if (e == hipSuccess) {
            for (uint32_t s = 0; s < steps; ++s) launch_probe_step((s & 1u) ? b : a, (s & 1u) ? a : b, workgroups, per_wg, s, c->stream);
            e = hipStreamEndCapture(c->stream, &graph);
        }
        if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
    }
    float ms = 0.0f;
    uint32_t host_flags[2] = {0u, 0u};
    for (uint32_t it = 0; it < iters + 2u && e == hipSuccess; ++it) {          // two warm-up rounds
        if (it == 2u) e = hipEventRecord(e0, c->stream);
        if (persistent) {
            if (e == hipSuccess) e = hipMemsetAsync(flags, 0, 4, c->stream);   // the counter; the flag stays
            launch_probe_persistent(a, b, workgroups, per_wg, steps, flags, flags + 1, c->stream);
        } else if (e == hipSuccess) {
            e = hipGraphLaunch(exec, c->stream);
        }
    }
    if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e == hipSuccess && persistent) e = hipMemcpy(host_flags, flags, 8, hipMemcpyDeviceToHost);
    if (exec) (void)hipGraphExecDestroy(exec);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (flags) (void)hipFree(flags);
    if (e != hipSuccess) return fail(c, GS_ERR_HIP, std::string("gs_sync_probe: ") + hipGetErrorString(e));
    *us_per_step = ms * 1000.0f / (float)iters / (float)steps;
    *timed_out = host_flags[1];
    return GS_OK;
}

// Tuning only: microseconds of one launch of `workgroups` workgroups that each issue `lines` 16-lane atomic adds (add = 1) or plain
// stores (add = 0) to rows of a rows x 16 counter table; stride_num / stride_den workgroups share a starting row.
int gs_atomic_probe(gs_ctx* c, uint32_t workgroups, uint32_t lines, uint32_t rows, uint32_t stride_num, uint32_t stride_den, uint32_t add,
                    uint32_t iters, float* us_per_launch) {
    if (!c || !us_per_launch || !workgroups || !rows || !stride_den || !iters) return GS_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    uint32_t* table = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc((void**)&table, (size_t)rows * 64);
    if (e == hipSuccess) e = hipMemsetAsync(table, 0, (size_t)rows * 64, c->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float ms = 0.0f;
    if (e == hipSuccess) {
        for (int w = 0; w < 3; ++w) launch_probe_atomics(table, rows, workgroups, lines, stride_num, stride_den, add, c->stream);
        e = hipEventRecord(e0, c->stream);
        for (uint32_t i = 0; i < iters; ++i) launch_probe_atomics(table, rows, workgroups, lines, stride_num, stride_den, add, c->stream);
        if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (table) (void)hipFree(table);
    if (e != hipSuccess) return fail(c, GS_ERR_HIP, std::string("gs_atomic_probe: ") + hipGetErrorString(e));
    *us_per_launch = ms * 1000.0f / (float)iters;
    return GS_OK;
}

// tools/soak.py: fills the LDS of every CU with `pattern` on the context's stream (2048 workgroups of 64 KB: every CU hosts
// several in turn, two at a time).
int gs_lds_poison(gs_ctx* c, uint32_t pattern) {
    if (!c) return GS_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    uint32_t* sink = nullptr;
    HIP_TRY(c, hipMalloc((void**)&sink, 4));
    hipLaunchKernelGGL(k_lds_poison, dim3(2048), dim3(256), 0, c->stream, pattern, sink);
    hipError_t e = hipStreamSynchronize(c->stream);
    (void)hipFree(sink);
    if (e != hipSuccess) return fail(c, GS_ERR_HIP, std::string("gs_lds_poison: ") + hipGetErrorString(e));
    return GS_OK;
}

// Tuning only (tools/lds_probe.py): microseconds per launch of k_probe_lds.
int gs_lds_probe(gs_ctx* c, uint32_t kind, uint32_t reps, uint32_t iters, float* us_per_launch) {
    if (!c || !us_per_launch || kind > 6u || !iters || reps > 64u) return GS_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    uint32_t* out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc((void**)&out, (size_t)512 * 512 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float ms = 0.0f;
    if (e == hipSuccess) {
        for (int w = 0; w < 3; ++w) launch_probe_lds(out, kind, reps, c->stream);
        e = hipEventRecord(e0, c->stream);
        for (uint32_t i = 0; i < iters; ++i) launch_probe_lds(out, kind, reps, c->stream);
        if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (out) (void)hipFree(out);
    if (e != hipSuccess) return fail(c, GS_ERR_HIP, std::string("gs_lds_probe: ") + hipGetErrorString(e));
    *us_per_launch = ms * 1000.0f / (float)iters;
    return GS_OK;
}


} // extern "C"
