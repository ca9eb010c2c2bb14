// gs_upload.hip -- one-off conversion of the reference's 336-byte AoS GaussianData records
// (Engine/Graphics/ShaderStructs.h:59-70) to the SoA planes the per-frame kernels read with fully
// coalesced loads.  Runs once per scene (Renderer::initForScene, Renderer.cpp:712-724), chunk by
// chunk through a staging buffer so a 50 M-splat scene never needs a second full-size AoS copy in HBM.
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
            // |R|_F^2 * max(scale)^2 >= largest eigenvalue of Sigma = (R S)(R S)^T, with R as Common.glsl:17-30
            // builds it from the (not necessarily unit) quaternion
            const float q0 = tile[r][8], x = tile[r][9], y = tile[r][10], z = tile[r][11];
            const float e[9] = {1.0f - 2.0f * y * y - 2.0f * z * z, 2.0f * x * y - 2.0f * q0 * z, 2.0f * x * z + 2.0f * q0 * y,
                                2.0f * x * y + 2.0f * q0 * z, 1.0f - 2.0f * x * x - 2.0f * z * z, 2.0f * y * z - 2.0f * q0 * x,
                                2.0f * x * z - 2.0f * q0 * y, 2.0f * y * z + 2.0f * q0 * x, 1.0f - 2.0f * x * x - 2.0f * y * y};
            float f2 = 0.0f;
#pragma unroll
            for (int k = 0; k < 9; ++k) f2 += e[k] * e[k];
            const float s0 = fabsf(tile[r][4]), s1 = fabsf(tile[r][5]), s2 = fabsf(tile[r][6]);
            const float sm = fmaxf(s0, fmaxf(s1, s2));
            s.sig2[g] = f2 * sm * sm;
        }
    }
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
