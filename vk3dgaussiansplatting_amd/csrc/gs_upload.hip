// gs_upload.hip -- one-off conversion of the reference's 336-byte AoS GaussianData records
// (Engine/Graphics/ShaderStructs.h:59-70) to the SoA planes the per-frame kernels read with fully
// coalesced loads.  Runs once per scene (Renderer::initForScene, Renderer.cpp:712-724), chunk by
// chunk through a staging buffer so a 50 M-splat scene never needs a second full-size AoS copy in HBM.
#include "gs_device_utils.h"
#include "gs_internal.h"

namespace gs {

// chunk: [count][84] floats starting at splat `first`; n = total splats (plane stride)
__global__ __launch_bounds__(256) void k_aos_to_soa(const float* __restrict__ chunk, uint32_t first,
                                                     uint32_t count, uint32_t n, SceneBuffers s) {
    __shared__ float tile[64][85];   // 64 records, padded row to dodge bank conflicts
    const uint32_t rec0 = blockIdx.x * 64u;
    const uint32_t nrec = (count - rec0) < 64u ? (count - rec0) : 64u;
    // coalesced read of nrec*84 consecutive floats
    for (uint32_t k = threadIdx.x; k < nrec * 84u; k += 256u)
        tile[k / 84u][k % 84u] = chunk[(size_t)rec0 * 84u + k];
    __syncthreads();
    // coalesced plane writes: lane = record
    const uint32_t r = threadIdx.x & 63u, part = threadIdx.x >> 6;   // 4 waves split the 59 fields
    if (r < nrec) {
        const size_t g = (size_t)first + rec0 + r;
        for (uint32_t fld = part; fld < 59u; fld += 4u) {
            if (fld < 3u) s.pos[(size_t)fld * n + g] = tile[r][0 + fld];
            else if (fld < 6u) s.scale[(size_t)(fld - 3u) * n + g] = tile[r][4 + (fld - 3u)];
            else if (fld < 10u) s.rot[(size_t)(fld - 6u) * n + g] = tile[r][8 + (fld - 6u)];
            else if (fld < 58u) {
                const uint32_t k = fld - 10u, coeff = k / 3u, ch = k % 3u;
                s.sh[(size_t)k * n + g] = tile[r][12 + coeff * 4u + ch];
            } else s.opacity[g] = tile[r][12 + 3];
        }
        if (part == 3u) {
            // |R|_2^2 * max(scale)^2 >= largest eigenvalue of Sigma = (R S)(R S)^T, with R as Common.glsl:17-30
            // builds it from the (not necessarily unit) quaternion
            const float q0 = tile[r][8], x = tile[r][9], y = tile[r][10], z = tile[r][11];
            const float e[9] = {1.0f - 2.0f * y * y - 2.0f * z * z, 2.0f * x * y - 2.0f * q0 * z, 2.0f * x * z + 2.0f * q0 * y,
                                2.0f * x * y + 2.0f * q0 * z, 1.0f - 2.0f * x * x - 2.0f * z * z, 2.0f * y * z - 2.0f * q0 * x,
                                2.0f * x * z - 2.0f * q0 * y, 2.0f * y * z + 2.0f * q0 * x, 1.0f - 2.0f * x * x - 2.0f * y * y};
            // |R|_2^2 <= min(trace, largest absolute row sum) of R R^T (Gershgorin): 1 for a unit quaternion, where the
            // Frobenius norm alone says 3
            float f2 = 0.0f, gersh = 0.0f;
#pragma unroll
            for (int k = 0; k < 9; ++k) f2 += e[k] * e[k];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float row = 0.0f;
#pragma unroll
                for (int j = 0; j < 3; ++j) row += fabsf(e[3 * i] * e[3 * j] + e[3 * i + 1] * e[3 * j + 1] + e[3 * i + 2] * e[3 * j + 2]);
                gersh = fmaxf(gersh, row);
            }
            const float r2 = (gersh < f2 ? gersh : f2) * 1.0001f;      // a NaN takes f2, NaN again: the splat is kept
            const float s0 = fabsf(tile[r][4]), s1 = fabsf(tile[r][5]), s2 = fabsf(tile[r][6]);
            const float sm = fmaxf(s0, fmaxf(s1, s2));
            s.sig2[g] = r2 * sm * sm;
        }
    }
}

// Per wave of the project kernel (64 consecutive splats -- neighbours in space, the arrays are in Morton order): the box
// around the positions and the largest sig2.  A context that renders a subset of the tile rows rejects whole waves
// with it (k_project).  min/max are exact selections, so the box contains every position whatever the order.
__global__ __launch_bounds__(256) void k_block_bounds(uint32_t n, SceneBuffers s) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const bool ok = g < n;
    float v[7];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float p = ok ? s.pos[(size_t)a * n + g] : 0.0f;
        v[a] = ok ? p : 3.0e38f;        // min
        v[3 + a] = ok ? p : -3.0e38f;   // max
    }
    v[6] = ok ? s.sig2[g] : 0.0f;
    // a NaN position or bound must not be lost by fminf/fmaxf: it poisons the record instead (k_project then keeps
    // the wave, every comparison with a NaN being false)
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 7; ++k) bad = bad || (ok && !(v[k] == v[k]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = fminf(v[k], __shfl_xor(v[k], off, 64));
#pragma unroll
        for (int k = 3; k < 7; ++k) v[k] = fmaxf(v[k], __shfl_xor(v[k], off, 64));
    }
    const bool wave_bad = __ballot(bad) != 0ull;
    if ((threadIdx.x & 63) == 0 && blockIdx.x * 256u + (threadIdx.x & ~63u) < n) {
        const float nanv = __builtin_nanf("");
        float4* out = reinterpret_cast<float4*>(s.block_bounds) + (size_t)(g >> 6) * 2;
        out[0] = wave_bad ? make_float4(nanv, nanv, nanv, nanv) : make_float4(v[0], v[1], v[2], v[3]);
        out[1] = wave_bad ? make_float4(nanv, nanv, nanv, nanv) : make_float4(v[4], v[5], v[6], 0.0f);
    }
}

void launch_block_bounds(uint32_t n, const SceneBuffers& s, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_block_bounds, dim3((n + 255u) / 256u), dim3(256), 0, stream, n, s);
}

void launch_aos_to_soa(const float* chunk, uint32_t first, uint32_t count, uint32_t n,
                       const SceneBuffers& s, hipStream_t stream) {
    if (count == 0) return;
    hipLaunchKernelGGL(k_aos_to_soa, dim3((count + 63u) / 64u), dim3(256), 0, stream, chunk, first,
                       count, n, s);
}

} // namespace gs

// ---------------------------------------------------------------------------------------------
// Stream bandwidth probes: the "measured HBM roofline" denominator of north_star (device-to-device
// copy on the same MI355X) and the small-buffer regime the per-pass sort kernels live in.
// ---------------------------------------------------------------------------------------------
namespace gs {

template <typename T>
__global__ __launch_bounds__(256) void k_stream_copy(const T* __restrict__ src, T* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

template <typename T>
__global__ __launch_bounds__(256) void k_stream_read(const T* __restrict__ src, uint32_t* __restrict__ sink, size_t n) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = src[i];
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
#pragma unroll
        for (int k = 0; k < (int)(sizeof(T) / 4); ++k) acc ^= w[k];
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;   // keeps the loads alive, practically never taken
}

// Copy with four 16-byte loads in flight per lane before the first store (the plain grid-stride copy above keeps one):
// what MI355X_MICROARCH.md's 6.3 TB/s "float4 copy" needs -- more bytes in flight per CU.  NT = 1: non-temporal
// stores (the destination is not re-read); NT = 2: non-temporal loads too.
template <int NT>
__global__ __launch_bounds__(256) void k_stream_copy4(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (NT == 2) {
                const unsigned __attribute__((ext_vector_type(4)))* p =
                    reinterpret_cast<const unsigned __attribute__((ext_vector_type(4)))*>(src + i + k * stride);
                const unsigned __attribute__((ext_vector_type(4))) t = __builtin_nontemporal_load(p);
                v[k] = make_uint4(t.x, t.y, t.z, t.w);
            } else {
                v[k] = src[i + k * stride];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (NT >= 1) {
                unsigned __attribute__((ext_vector_type(4))) t = {v[k].x, v[k].y, v[k].z, v[k].w};
                __builtin_nontemporal_store(t, reinterpret_cast<unsigned __attribute__((ext_vector_type(4)))*>(dst + i + k * stride));
            } else {
                dst[i + k * stride] = v[k];
            }
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// Radix-scatter write pattern without the sorting: the buffer is treated as three arrays of n dwords;
// persistent workgroups read tiles of TILE consecutive dwords from each array and write every tile as
// 16 runs of TILE/16 dwords into 16 destination regions (what a 4-bit pass with uniform digits does).
template <int TILE, int SKEW = 0>
__global__ __launch_bounds__(256) void k_scatter_pattern(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n3) {
    const size_t n = n3 / 3;                       // dwords per array
    const size_t tiles = n / TILE;
    constexpr int RUN = TILE / 16;
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const uint32_t* s = src + a * n + t * TILE;
            uint32_t* d = dst + a * n;
#pragma unroll
            for (int r = 0; r < TILE / 256; ++r) {
                const int p = r * 256 + threadIdx.x;          // position inside the tile
                const int digit = p / RUN, off = p % RUN;
                d[(size_t)digit * (n / 16) + t * RUN + off + (size_t)digit * SKEW] = s[p];   // SKEW: misaligned regions
            }
        }
    }
}

void launch_stream_probe(int kind, const void* src, void* dst, size_t bytes, uint32_t blocks, hipStream_t stream) {
    switch (kind) {
        case 4: hipLaunchKernelGGL(k_scatter_pattern<3072>, dim3(blocks), dim3(256), 0, stream, (const uint32_t*)src, (uint32_t*)dst, bytes / 4); return;
        case 5: hipLaunchKernelGGL(k_scatter_pattern<6144>, dim3(blocks), dim3(256), 0, stream, (const uint32_t*)src, (uint32_t*)dst, bytes / 4); return;
        case 6: hipLaunchKernelGGL(k_scatter_pattern<12288>, dim3(blocks), dim3(256), 0, stream, (const uint32_t*)src, (uint32_t*)dst, bytes / 4); return;
        case 7: hipLaunchKernelGGL(k_scatter_pattern<49152>, dim3(blocks), dim3(256), 0, stream, (const uint32_t*)src, (uint32_t*)dst, bytes / 4); return;
        case 8: hipLaunchKernelGGL((k_scatter_pattern<3072, 37>), dim3(blocks), dim3(256), 0, stream, (const uint32_t*)src, (uint32_t*)dst, bytes / 4); return;
        case 9: hipLaunchKernelGGL((k_scatter_pattern<3072, 5>), dim3(blocks), dim3(256), 0, stream, (const uint32_t*)src, (uint32_t*)dst, bytes / 4); return;
        case 10: hipLaunchKernelGGL(k_stream_copy4<0>, dim3(blocks), dim3(256), 0, stream, (const uint4*)src, (uint4*)dst, bytes / 16); return;
        case 11: hipLaunchKernelGGL(k_stream_copy4<1>, dim3(blocks), dim3(256), 0, stream, (const uint4*)src, (uint4*)dst, bytes / 16); return;
        case 12: hipLaunchKernelGGL(k_stream_copy4<2>, dim3(blocks), dim3(256), 0, stream, (const uint4*)src, (uint4*)dst, bytes / 16); return;
        default: break;
    }
    switch (kind) {
        case 0: hipLaunchKernelGGL(k_stream_read<uint4>, dim3(blocks), dim3(256), 0, stream, (const uint4*)src, (uint32_t*)dst, bytes / 16); break;
        case 1: hipLaunchKernelGGL(k_stream_copy<uint4>, dim3(blocks), dim3(256), 0, stream, (const uint4*)src, (uint4*)dst, bytes / 16); break;
        case 2: hipLaunchKernelGGL(k_stream_read<uint32_t>, dim3(blocks), dim3(256), 0, stream, (const uint32_t*)src, (uint32_t*)dst, bytes / 4); break;
        default: hipLaunchKernelGGL(k_stream_copy<uint32_t>, dim3(blocks), dim3(256), 0, stream, (const uint32_t*)src, (uint32_t*)dst, bytes / 4); break;
    }
}

} // namespace gs
