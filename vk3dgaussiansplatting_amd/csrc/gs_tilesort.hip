// gs_tilesort.hip -- alternative back-end behind the GpuSort seam (Engine/Graphics/Sort/GpuSort.h:8-22;
// SURVEY.md 8(f)-4): GS_SORT_TILE_BUCKET.
//
// The contractual sorter (gs_sort.hip) runs P = 12 global 4-bit passes over the whole 64-bit key.
// The key is (tile << 32 | depth) and RenderGaussians only ever looks at one tile's run, so the same
// order is reached with far less HBM traffic by sorting MSD-first:
//   1. the global passes sort by the TILE word only (ceil(bits(T-1)/4) = 4 passes instead of 12;
//      same Count/Scan/Scatter kernels, stable, starting from the canonical emission order);
//   2. FindRanges;
//   3. k_tile_sort: every tile's run (mean ~1.6 k elements at the README shapes) is sorted by the
//      32-bit depth word with a stable LSD radix sort that never leaves the CU: the run is loaded into
//      LDS once, sorted in place by four 8-bit passes (runs up to 4096 elements: every key sits in a
//      register between the read and the write of a pass), and written back once.
// Stable tile sort followed by a stable per-tile depth sort == stable sort by (tile, depth): the
// output (keys, payload order, ranges, pixels) is bit-identical to the contractual path.
// Runs are dispatched by size class (16 / 32 / 64 KB of LDS with single-chunk passes, 160 KB with
// chunked passes); runs beyond that are sorted in place in global memory (same routine, generic pointers).
#include "gs_device_utils.h"
#include "gs_internal.h"

namespace gs {

constexpr int kTsRounds = 4;                 // keys per thread per chunk
constexpr uint32_t kTsBigMax = 9984;         // 1024-thread / 160 KB variant
constexpr uint32_t kTsScratchWords = 16 + 16 + 16 * 16 + 16 * 16 + 4;   // 4-bit passes: hist, base, wcnt, wbase, flags
// 8-bit single-chunk passes: [waves][256] counts/bases + wave totals + flags
constexpr uint32_t ts_scratch8_words(uint32_t threads) { return (threads / 64u) * 256u + 16u; }

// One stable 4-bit pass of a whole workgroup over n (key, id) pairs, src -> dst (LDS or global).
// Returns false (uniformly) if every key has the same digit: nothing was moved.
template <int THREADS>
__device__ __forceinline__ bool wg_radix_pass(const uint32_t* src_key, const uint32_t* src_id,
                                              uint32_t* dst_key, uint32_t* dst_id, uint32_t n,
                                              uint32_t shift, uint32_t* scratch) {
    constexpr int WAVES = THREADS / 64;
    constexpr uint32_t CHUNK = THREADS * kTsRounds;
    uint32_t* s_hist = scratch;              // [16]
    uint32_t* s_base = scratch + 16;         // [16] running first position of every digit
    uint32_t* s_wcnt = scratch + 32;         // [16 waves][16]
    uint32_t* s_wbase = scratch + 32 + 256;  // [16 waves][16]
    uint32_t* s_flag = scratch + 32 + 512;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if (tid < kBins) s_hist[tid] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += THREADS) atomicAdd(&s_hist[(src_key[i] >> shift) & 15u], 1u);
    __syncthreads();
    if (tid < kBins) {
        const uint32_t v = s_hist[tid];
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < kBins; off <<= 1) {
            const uint32_t t = __shfl_up(inc, off, 64);
            if (tid >= off) inc += t;
        }
        s_base[tid] = inc - v;
        const uint64_t same = __ballot(v == n);      // one digit holds everything: identity pass
        if (tid == 0) s_flag[0] = same ? 1u : 0u;
    }
    __syncthreads();
    if (s_flag[0]) return false;

    for (uint32_t c0 = 0; c0 < n; c0 += CHUNK) {
        uint32_t key[kTsRounds], id[kTsRounds], rank[kTsRounds];
        const uint32_t base = c0 + (uint32_t)wave * (kTsRounds * 64) + lane;
        uint32_t cntreg = 0;
#pragma unroll
        for (int r = 0; r < kTsRounds; ++r) {
            const uint32_t idx = base + r * 64;
            const bool ok = idx < n;
            key[r] = ok ? src_key[idx] : 0xFFFFFFFFu;
            id[r] = ok ? src_id[idx] : 0u;
            const uint32_t dg = (key[r] >> shift) & 15u;
            uint64_t mask = __ballot(ok);
#pragma unroll
            for (int b = 0; b < kRadixBits; ++b) {
                const bool bit = (dg >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                mask &= bit ? bal : ~bal;
            }
            mask = ok ? mask : 0ull;
            const uint32_t in_round = mbcnt(mask);
            const uint32_t n_round = (uint32_t)__popcll(mask);
            const uint32_t before = (uint32_t)__shfl((int)cntreg, (int)dg, 64);
            rank[r] = before + in_round;
            const bool leader = ok && in_round == 0u;
            const int dest = leader ? (int)dg : 63;
            const uint32_t recv = (uint32_t)__builtin_amdgcn_ds_permute(dest << 2, (int)n_round);
            cntreg += lane < kBins ? recv : 0u;
        }
        if (lane < kBins) s_wcnt[wave * 16 + lane] = cntreg;
        __syncthreads();
        if (tid < kBins) {
            uint32_t run = s_base[tid];
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                s_wbase[w * 16 + tid] = run;
                run += s_wcnt[w * 16 + tid];
            }
            s_base[tid] = run;               // next chunk continues behind this one
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kTsRounds; ++r) {
            const uint32_t idx = base + r * 64;
            if (idx < n) {
                const uint32_t p = s_wbase[wave * 16 + ((key[r] >> shift) & 15u)] + rank[r];
                dst_key[p] = key[r];
                dst_id[p] = id[r];
            }
        }
        // the next chunk's s_wcnt writes are ordered behind this chunk's s_wbase reads by its
        // first barrier; its s_wbase writes come after that barrier too
    }
    __syncthreads();
    return true;
}

// Single-chunk variant for runs of at most THREADS * ROUNDS keys: the digit totals come from the
// ranking itself (no separate histogram sweep) and a pass needs three barriers.
template <int THREADS, int ROUNDS>
__device__ __forceinline__ bool wg_radix_pass_single(const uint32_t* src_key, const uint32_t* src_id,
                                                     uint32_t* dst_key, uint32_t* dst_id, uint32_t n,
                                                     uint32_t shift, uint32_t* scratch) {
    constexpr int WAVES = THREADS / 64;
    uint32_t* s_wcnt = scratch + 32;
    uint32_t* s_wbase = scratch + 32 + 256;
    uint32_t* s_flag = scratch + 32 + 512;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t key[ROUNDS], id[ROUNDS], rank[ROUNDS];
    const uint32_t base = (uint32_t)wave * (ROUNDS * 64) + lane;
    uint32_t cntreg = 0;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint32_t idx = base + r * 64;
        const bool ok = idx < n;
        key[r] = ok ? src_key[idx] : 0xFFFFFFFFu;
        id[r] = ok ? src_id[idx] : 0u;
        const uint32_t dg = (key[r] >> shift) & 15u;
        uint64_t mask = __ballot(ok);
#pragma unroll
        for (int b = 0; b < kRadixBits; ++b) {
            const bool bit = (dg >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            mask &= bit ? bal : ~bal;
        }
        mask = ok ? mask : 0ull;
        const uint32_t in_round = mbcnt(mask);
        const uint32_t n_round = (uint32_t)__popcll(mask);
        const uint32_t before = (uint32_t)__shfl((int)cntreg, (int)dg, 64);
        rank[r] = before + in_round;
        const bool leader = ok && in_round == 0u;
        const int dest = leader ? (int)dg : 63;
        const uint32_t recv = (uint32_t)__builtin_amdgcn_ds_permute(dest << 2, (int)n_round);
        cntreg += lane < kBins ? recv : 0u;
    }
    if (lane < kBins) s_wcnt[wave * 16 + lane] = cntreg;
    __syncthreads();
    if (tid < kBins) {
        uint32_t c[WAVES];
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { c[w] = s_wcnt[w * 16 + tid]; tot += c[w]; }
        uint32_t inc = tot;
#pragma unroll
        for (int off = 1; off < kBins; off <<= 1) {
            const uint32_t t = __shfl_up(inc, off, 64);
            if (tid >= off) inc += t;
        }
        uint32_t run = inc - tot;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { s_wbase[w * 16 + tid] = run; run += c[w]; }
        const uint64_t same = __ballot(tot == n);      // one digit holds everything: identity pass
        if (tid == 0) s_flag[0] = same ? 1u : 0u;
    }
    __syncthreads();
    const bool identity = s_flag[0] != 0u;
    if (!identity) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t idx = base + r * 64;
            if (idx < n) {
                const uint32_t p = s_wbase[wave * 16 + ((key[r] >> shift) & 15u)] + rank[r];
                dst_key[p] = key[r];
                dst_id[p] = id[r];
            }
        }
    }
    __syncthreads();
    return !identity;
}

// 8-bit digit variant of the single-chunk pass (4 passes over the 32-bit depth word instead of 8): keys of a
// wave that share a digit are found by eight ballots, their leader adds the group's size to the wave's own LDS
// histogram (256 bins; only this wave touches it, rounds run in program order, so the returned running count is
// deterministic) and hands the old value to the group.  scratch8 = [WAVES][256] counts/bases + 8 wave totals + flag.
template <int THREADS, int ROUNDS>
__device__ __forceinline__ bool wg_radix_pass_single8(const uint32_t* src_key, const uint32_t* src_id,
                                                      uint32_t* dst_key, uint32_t* dst_id, uint32_t n,
                                                      uint32_t shift, uint32_t* scratch8) {
    constexpr int WAVES = THREADS / 64;
    uint32_t* s_hist = scratch8;                    // [WAVES][256]
    uint32_t* s_wtot = scratch8 + WAVES * 256;      // [4] totals of the four 64-digit groups
    uint32_t* s_flag = s_wtot + 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < WAVES * 256 / THREADS; ++k) s_hist[k * THREADS + tid] = 0u;
    __syncthreads();

    uint32_t key[ROUNDS], id[ROUNDS], rank[ROUNDS];
    const uint32_t base = (uint32_t)wave * (ROUNDS * 64) + lane;
    uint32_t* my_hist = s_hist + wave * 256;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint32_t idx = base + r * 64;
        const bool ok = idx < n;
        key[r] = ok ? src_key[idx] : 0xFFFFFFFFu;
        id[r] = ok ? src_id[idx] : 0u;
        const uint32_t dg = (key[r] >> shift) & 255u;
        uint64_t mask = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (dg >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            mask &= bit ? bal : ~bal;
        }
        mask = ok ? mask : 0ull;
        const uint32_t in_round = mbcnt(mask);
        uint32_t old = 0u;
        if (ok && in_round == 0u) old = atomicAdd(&my_hist[dg], (uint32_t)__popcll(mask));
        const int leader = ok ? (int)__builtin_ctzll(mask) : lane;
        rank[r] = (uint32_t)__shfl((int)old, leader, 64) + in_round;
    }
    __syncthreads();
    // digit d = tid (THREADS >= 256): totals over the waves, exclusive scan over the 256 digits
    uint32_t c[WAVES];
    uint32_t tot = 0, inc = 0;
    if (tid < 256) {
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { c[w] = s_hist[w * 256 + tid]; tot += c[w]; }
        inc = wave_inclusive_scan(tot);
        if (lane == 63) s_wtot[wave] = inc;
        const uint64_t same = __ballot(tot == n);      // one digit holds everything: identity pass
        if (lane == 0) s_flag[wave] = same ? 1u : 0u;
    }
    __syncthreads();
    if (tid < 256) {
        uint32_t run = inc - tot;
        for (int w = 0; w < wave; ++w) run += s_wtot[w];
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { s_hist[w * 256 + tid] = run; run += c[w]; }
    }
    const bool identity = (s_flag[0] | s_flag[1] | s_flag[2] | s_flag[3]) != 0u;
    __syncthreads();
    if (!identity) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t idx = base + r * 64;
            if (idx < n) {
                const uint32_t p = my_hist[(key[r] >> shift) & 255u] + rank[r];
                dst_key[p] = key[r];
                dst_id[p] = id[r];
            }
        }
    }
    __syncthreads();
    return !identity;
}

// LDS-resident sort of one tile's run of NMIN < n <= NMAX elements.  ROUNDS > 0: single-chunk passes
// (NMAX == THREADS * ROUNDS); ROUNDS == 0: chunked passes (any NMAX that fits LDS).
// workgroup b -> global tile id of the b-th owned tile (FrameParams: rows first_row + k * row_stride)
struct TsTiles { uint32_t grid_w, first_row, row_stride; };
__device__ __forceinline__ uint32_t ts_tile(const TsTiles& m, uint32_t b) {
    const uint32_t k = b / m.grid_w;
    return (m.first_row + k * m.row_stride) * m.grid_w + (b - k * m.grid_w);
}

template <int THREADS, int ROUNDS, uint32_t NMIN, uint32_t NMAX>
__global__ __launch_bounds__(THREADS) void k_tile_sort_lds(const uint32_t* __restrict__ ranges,
                                                            uint32_t* __restrict__ lo,
                                                            uint32_t* __restrict__ id,
                                                            TsTiles tiles) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    static_assert(ROUNDS == 0 || NMAX == (uint32_t)(THREADS * ROUNDS), "single-chunk capacity");
    const uint32_t tile = ts_tile(tiles, blockIdx.x);
    const uint32_t start = ranges[tile * 2 + 0], end = ranges[tile * 2 + 1];
    const uint32_t n = end > start ? end - start : 0u;
    if (n <= NMIN || n > NMAX) return;
    // single-chunk classes sort in place (every key of the run is in registers between the read and the write
    // of a pass): two LDS arrays; the chunked class ping-pongs between two images: four
    constexpr uint32_t IMAGES = ROUNDS > 0 ? 1u : 2u;
    uint32_t* key_a = smem;
    uint32_t* id_a = smem + NMAX;
    uint32_t* key_b = smem + (IMAGES - 1u) * 2u * NMAX;
    uint32_t* id_b = key_b + NMAX;
    uint32_t* scratch = smem + IMAGES * 2u * NMAX;
    const int tid = threadIdx.x;
    for (uint32_t i = tid; i < n; i += THREADS) {
        key_a[i] = lo[start + i];
        id_a[i] = id[start + i];
    }
    __syncthreads();
    bool in_a = true;
    if constexpr (ROUNDS > 0) {
#pragma unroll 1
        for (uint32_t shift = 0; shift < 32u; shift += 8u)
            (void)wg_radix_pass_single8<THREADS, ROUNDS>(key_a, id_a, key_a, id_a, n, shift, scratch);
    } else {
#pragma unroll 1
        for (uint32_t shift = 0; shift < 32u; shift += kRadixBits) {
            const bool moved = in_a ? wg_radix_pass<THREADS>(key_a, id_a, key_b, id_b, n, shift, scratch)
                                    : wg_radix_pass<THREADS>(key_b, id_b, key_a, id_a, n, shift, scratch);
            if (moved) in_a = !in_a;
        }
    }
    const uint32_t* fk = in_a ? key_a : key_b;
    const uint32_t* fi = in_a ? id_a : id_b;
    for (uint32_t i = tid; i < n; i += THREADS) {
        lo[start + i] = fk[i];
        id[start + i] = fi[i];
    }
}

// Oversized runs: the same passes directly on the global ping-pong halves (one workgroup per tile).
__global__ __launch_bounds__(1024) void k_tile_sort_global(const uint32_t* __restrict__ ranges,
                                                           uint32_t* lo, uint32_t* id, uint32_t* lo_alt,
                                                           uint32_t* id_alt, TsTiles tiles) {
    __shared__ uint32_t scratch[kTsScratchWords];
    const uint32_t tile = ts_tile(tiles, blockIdx.x);
    const uint32_t start = ranges[tile * 2 + 0], end = ranges[tile * 2 + 1];
    const uint32_t n = end > start ? end - start : 0u;
    if (n <= kTsBigMax) return;
    bool in_a = true;
#pragma unroll 1
    for (uint32_t shift = 0; shift < 32u; shift += kRadixBits) {
        const bool moved = in_a ? wg_radix_pass<1024>(lo + start, id + start, lo_alt + start, id_alt + start, n, shift, scratch)
                                : wg_radix_pass<1024>(lo_alt + start, id_alt + start, lo + start, id + start, n, shift, scratch);
        if (moved) in_a = !in_a;
    }
    if (!in_a)
        for (uint32_t i = threadIdx.x; i < n; i += 1024) {
            lo[start + i] = lo_alt[start + i];
            id[start + i] = id_alt[start + i];
        }
}

// size classes: runs of 2..1024, ..2048, ..4096 elements sort in place with single-chunk 8-bit passes in 12 / 20 /
// 40 KB of LDS; ..9984 with chunked 4-bit passes ping-ponging in 160 KB; beyond: global.
#define TS_KERNELS(X)                                   \
    X((k_tile_sort_lds<256, 4, 1u, 1024u>), 256, 1024u)   \
    X((k_tile_sort_lds<256, 8, 1024u, 2048u>), 256, 2048u) \
    X((k_tile_sort_lds<512, 8, 2048u, 4096u>), 512, 4096u) \
    X((k_tile_sort_lds<1024, 0, 4096u, kTsBigMax>), 0, kTsBigMax)

static size_t ts_lds_bytes(uint32_t nmax, uint32_t threads = 0) {   // threads != 0: a single-chunk (8-bit, in-place) class
    if (threads) return (2 * (size_t)nmax + ts_scratch8_words(threads)) * sizeof(uint32_t);
    return (4 * (size_t)nmax + kTsScratchWords) * sizeof(uint32_t);
}

int init_tile_sort() {
    hipError_t e = hipSuccess;
#define X(KERNEL, THREADS, NMAX)                                                                 \
    if (e == hipSuccess)                                                                          \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL),                          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ts_lds_bytes(NMAX, THREADS));
    TS_KERNELS(X)
#undef X
    return e == hipSuccess ? 0 : (int)e;
}

// The size classes touch disjoint tiles, so the rare big ones (few workgroups, whole-CU LDS) run on a
// helper stream beside the two small classes that hold most of the work: fork after FindRanges, join
// before RenderGaussians.
void launch_tile_sort(const FrameParams& fp, const uint32_t* ranges, uint32_t* lo, uint32_t* id,
                      uint32_t* lo_alt, uint32_t* id_alt, hipStream_t stream, hipStream_t helper,
                      hipEvent_t fork, hipEvent_t join) {
    const uint32_t num_tiles = fp.rows_owned * fp.grid_w;
    if (num_tiles == 0) return;
    const TsTiles tile0{fp.grid_w, fp.first_row, fp.row_stride};
    const bool split = helper != nullptr && fork != nullptr && join != nullptr;
    hipStream_t big = split ? helper : stream;
    if (split) {
        (void)hipEventRecord(fork, stream);
        (void)hipStreamWaitEvent(helper, fork, 0);
    }
    hipLaunchKernelGGL((k_tile_sort_lds<512, 8, 2048u, 4096u>), dim3(num_tiles), dim3(512), ts_lds_bytes(4096u, 512), big, ranges, lo, id, tile0);
    hipLaunchKernelGGL((k_tile_sort_lds<1024, 0, 4096u, kTsBigMax>), dim3(num_tiles), dim3(1024), ts_lds_bytes(kTsBigMax), big, ranges, lo, id, tile0);
    hipLaunchKernelGGL(k_tile_sort_global, dim3(num_tiles), dim3(1024), 0, big, ranges, lo, id, lo_alt, id_alt, tile0);
    hipLaunchKernelGGL((k_tile_sort_lds<256, 8, 1024u, 2048u>), dim3(num_tiles), dim3(256), ts_lds_bytes(2048u, 256), stream, ranges, lo, id, tile0);
    hipLaunchKernelGGL((k_tile_sort_lds<256, 4, 1u, 1024u>), dim3(num_tiles), dim3(256), ts_lds_bytes(1024u, 256), stream, ranges, lo, id, tile0);
    if (split) {
        (void)hipEventRecord(join, helper);
        (void)hipStreamWaitEvent(stream, join, 0);
    }
}

} // namespace gs
