// gs_sort.hip -- the reference's stable 4-bit LSD radix sort over 64-bit (tile<<32 | depth) keys
// with a 32-bit payload, re-designed for gfx950 (wave64, 256 CUs, HBM-bound).
//
// Reference pipeline per 4-bit pass (RadixSort.cpp:309-642): five dependent dispatches
//   Count -> Reduce -> Scan -> ScanAdd -> Scatter, 64 keys per workgroup, 16-byte uvec4 elements.
// Here the same five stages run in TWO launches per pass over 4096-key tiles ("groups"):
//   k_count    Count  : per-group digit histogram -> table[bin][group]  (RadixSortCount.comp:40-91);
//                       reads only the 4-byte key half the digit lives in (keys are SoA).
//              Reduce : the group's 16 counts are added into seg_sum[bin][group/64] with 16
//                       fire-and-forget device-scope integer atomics (RadixSortReduce.comp:34-72;
//                       integer adds commute, so the result is deterministic).
//   k_scatter  Scan   : prologue -- every workgroup derives its segment's exclusive base from the
//                       (16 x S)-entry seg_sum array, bin-major (RadixSortScan.comp:29-71);
//              ScanAdd: prologue -- exclusive prefix of the group's counts inside its 64-group
//                       segment, read from the L2-resident table (RadixSortScanAdd.comp:34-66);
//              Scatter: wave64 match-mask ranking (stable), LDS-staged local sort, run-wise
//                       coalesced stores (RadixSortScatter.comp:58-171).
// Folding Reduce/Scan/ScanAdd into the neighbours removes three dependent launches per pass
// (36 per frame at 12 passes); each costs ~2 us of launch boundary on MI355X and the single-
// workgroup Scan is the serial tail the reference's README.md:34 complains about.
// Output is bit-identical to a stable sort by the low num_sort_bits of the key.
// Launch grids are sized from the list CAPACITY; workgroups beyond the device-side element count
// (SortParams, the IndirectSetup record) exit at once -- no host read-back inside a frame.
#include "gs_device_utils.h"
#include "gs_internal.h"

namespace gs {

__device__ __forceinline__ uint32_t digit_of(uint32_t word, uint32_t sh) { return (word >> sh) & 15u; }

constexpr int kSortWaves = kSortThreads / 64;

// ---------------------------------------------------------------------------------------------
// Count + Reduce
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSortThreads) void k_count(const SortParams* __restrict__ params,
                                                         const uint32_t* __restrict__ word,
                                                         uint32_t* __restrict__ table,
                                                         uint32_t* __restrict__ seg_sum,
                                                         uint32_t sh) {
    __shared__ uint32_t s_cnt[kSortWaves][kBins];
    const uint32_t e = params->num_elems, G = params->num_groups, S = params->num_segments;
    const uint32_t grp = blockIdx.x;
    if (grp >= G) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < kSortWaves * kBins) (&s_cnt[0][0])[tid] = 0;
    __syncthreads();
    const uint32_t base = grp * kSortTile + (uint32_t)wave * (kSortKeysPerThread * 64) + lane;
    uint32_t w[kSortKeysPerThread];
#pragma unroll
    for (int r = 0; r < kSortKeysPerThread; ++r) {
        const uint32_t idx = base + r * 64;
        w[r] = idx < e ? word[idx] : 0u;
    }
#pragma unroll
    for (int r = 0; r < kSortKeysPerThread; ++r) {
        const uint32_t idx = base + r * 64;
        if (idx < e) atomicAdd(&s_cnt[wave][digit_of(w[r], sh)], 1u);   // LDS, order-free
    }
    __syncthreads();
    if (tid < kBins) {
        uint32_t t = 0;
#pragma unroll
        for (int k = 0; k < kSortWaves; ++k) t += s_cnt[k][tid];
        table[tid * G + grp] = t;                                  // RadixSortCount.comp:89, bin-major
        if (t) atomicAdd(&seg_sum[tid * S + grp / kSegGroups], t); // Reduce
    }
}

// ---------------------------------------------------------------------------------------------
// Scan + ScanAdd (prologue) + Scatter
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSortThreads) void k_scatter(
    const SortParams* __restrict__ params, const uint32_t* __restrict__ in_lo,
    const uint32_t* __restrict__ in_hi, const uint32_t* __restrict__ in_id,
    uint32_t* __restrict__ out_lo, uint32_t* __restrict__ out_hi, uint32_t* __restrict__ out_id,
    const uint32_t* __restrict__ table, const uint32_t* __restrict__ seg_sum, uint32_t shift) {
    __shared__ uint32_t s_lo[kSortTile];
    __shared__ uint32_t s_hi[kSortTile];
    __shared__ uint32_t s_id[kSortTile];
    __shared__ uint32_t s_wcnt[kSortWaves][kBins];
    __shared__ uint32_t s_wbase[kSortWaves][kBins];
    __shared__ uint32_t s_dtot[kBins];   // Scan: total keys per digit over the whole list
    __shared__ uint32_t s_gpre[kBins];   // Scan+ScanAdd: keys of digit d in groups before this one
    __shared__ int32_t s_gbase[kBins];

    const uint32_t e = params->num_elems, G = params->num_groups, S = params->num_segments;
    const uint32_t grp = blockIdx.x;
    if (grp >= G) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool use_hi = shift >= 32u;
    const uint32_t sh = shift & 31u;
    const uint32_t tile_base = grp * kSortTile;
    const uint32_t base = tile_base + (uint32_t)wave * (kSortKeysPerThread * 64) + lane;

    // ---- load (coalesced: each wave-instruction reads 256 contiguous bytes per array)
    uint32_t lo[kSortKeysPerThread], hi[kSortKeysPerThread], id[kSortKeysPerThread];
#pragma unroll
    for (int r = 0; r < kSortKeysPerThread; ++r) {
        const uint32_t idx = base + r * 64;
        const bool ok = idx < e;
        lo[r] = ok ? in_lo[idx] : 0xFFFFFFFFu;
        hi[r] = ok ? in_hi[idx] : 0xFFFFFFFFu;
        id[r] = ok ? in_id[idx] : 0u;
    }

    // ---- Scan + ScanAdd: wave w owns digits 4w..4w+3
    {
        const uint32_t seg = grp / kSegGroups, j = grp - seg * kSegGroups;
#pragma unroll
        for (int q = 0; q < kBins / kSortWaves; ++q) {
            const int d = wave * (kBins / kSortWaves) + q;
            uint32_t tot = 0, pre = 0;
            for (uint32_t s0 = 0; s0 < S; s0 += 64) {
                const uint32_t s = s0 + lane;
                const uint32_t v = s < S ? seg_sum[d * S + s] : 0u;
                tot += v;
                pre += s < seg ? v : 0u;
            }
            const uint32_t gi = seg * kSegGroups + lane;
            pre += ((uint32_t)lane < j && gi < G) ? table[d * G + gi] : 0u;
            tot = wave_reduce_add(tot);
            pre = wave_reduce_add(pre);
            if (lane == 0) { s_dtot[d] = tot; s_gpre[d] = pre; }
        }
    }

    // ---- stable rank inside the wave.  Per round: 4 ballots give every lane the mask of lanes
    //      holding the same digit; lane d (d < 16) keeps the wave's running count of digit d.
    uint32_t rank[kSortKeysPerThread];
    uint32_t cntreg = 0;
#pragma unroll
    for (int r = 0; r < kSortKeysPerThread; ++r) {
        const uint32_t idx = base + r * 64;
        const bool ok = idx < e;
        const uint32_t dg = digit_of(use_hi ? hi[r] : lo[r], sh);
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
        // the first lane of every digit group sends the group's size to counter lane `dg`;
        // everybody else sends to lane 63, which holds no counter
        const bool leader = ok && in_round == 0u;
        const int dest = leader ? (int)dg : 63;
        const uint32_t recv = (uint32_t)__builtin_amdgcn_ds_permute(dest << 2, (int)n_round);
        cntreg += lane < kBins ? recv : 0u;
    }
    if (lane < kBins) s_wcnt[wave][lane] = cntreg;
    __syncthreads();

    // ---- local digit starts, per-wave bases, global base (threads 0..15, one per digit)
    if (tid < kBins) {
        uint32_t c[kSortWaves];
        uint32_t tot = 0;
#pragma unroll
        for (int k = 0; k < kSortWaves; ++k) { c[k] = s_wcnt[k][tid]; tot += c[k]; }
        uint32_t inc = tot, ginc = s_dtot[tid];
#pragma unroll
        for (int off = 1; off < kBins; off <<= 1) {
            const uint32_t t = __shfl_up(inc, off, 64);
            const uint32_t gt = __shfl_up(ginc, off, 64);
            if (tid >= off) { inc += t; ginc += gt; }
        }
        const uint32_t dstart = inc - tot;              // first local position of digit d
        const uint32_t gstart = ginc - s_dtot[tid];     // first global index of digit d
        uint32_t run = dstart;
#pragma unroll
        for (int k = 0; k < kSortWaves; ++k) { s_wbase[k][tid] = run; run += c[k]; }
        s_gbase[tid] = (int32_t)(gstart + s_gpre[tid] - dstart); // global = s_gbase[d] + local pos
    }
    __syncthreads();

    // ---- local sort into LDS
#pragma unroll
    for (int r = 0; r < kSortKeysPerThread; ++r) {
        const uint32_t idx = base + r * 64;
        if (idx < e) {
            const uint32_t dg = digit_of(use_hi ? hi[r] : lo[r], sh);
            const uint32_t p = s_wbase[wave][dg] + rank[r];
            s_lo[p] = lo[r];
            s_hi[p] = hi[r];
            s_id[p] = id[r];
        }
    }
    __syncthreads();

    // ---- run-wise coalesced stores: consecutive local positions of one digit are consecutive
    //      global indices (RadixSortScatter.comp:153-168)
    const uint32_t valid = (e - tile_base) < (uint32_t)kSortTile ? (e - tile_base) : (uint32_t)kSortTile;
#pragma unroll
    for (int r = 0; r < kSortKeysPerThread; ++r) {
        const uint32_t p = (uint32_t)r * kSortThreads + tid;
        if (p < valid) {
            const uint32_t l = s_lo[p], h = s_hi[p];
            const uint32_t d = digit_of(use_hi ? h : l, sh);
            const uint32_t o = (uint32_t)(s_gbase[d] + (int32_t)p);
            out_lo[o] = l;
            out_hi[o] = h;
            out_id[o] = s_id[p];
        }
    }
}

int launch_radix_sort(const SortBuffers& sb, uint32_t capacity, uint32_t num_sort_bits,
                      hipStream_t stream, hipEvent_t* scatter_events) {
    const uint32_t max_groups = (capacity + kSortTile - 1) / kSortTile;
    const uint32_t max_segments = (max_groups + kSegGroups - 1) / kSegGroups;
    const uint32_t passes = (num_sort_bits + kRadixBits - 1) / kRadixBits;
    // one zeroed seg_sum slice per pass (role of gpuClearBuffers, RadixSort.cpp:676-692)
    (void)hipMemsetAsync(sb.seg_sum, 0, (size_t)passes * kBins * max_segments * sizeof(uint32_t), stream);
    int src = 0;
    uint32_t pass = 0;
    for (uint32_t shift = 0; shift < num_sort_bits; shift += kRadixBits, ++pass) { // RadixSort.cpp:309
        const int dst = src ^ 1;
        const uint32_t* word = shift >= 32u ? sb.hi[src] : sb.lo[src];
        uint32_t* seg = sb.seg_sum + (size_t)pass * kBins * max_segments;
        hipLaunchKernelGGL(k_count, dim3(max_groups), dim3(kSortThreads), 0, stream, sb.params,
                           word, sb.table, seg, shift & 31u);
        if (scatter_events) (void)hipEventRecord(scatter_events[2 * pass], stream);
        hipLaunchKernelGGL(k_scatter, dim3(max_groups), dim3(kSortThreads), 0, stream, sb.params,
                           sb.lo[src], sb.hi[src], sb.id[src], sb.lo[dst], sb.hi[dst], sb.id[dst],
                           sb.table, seg, shift);
        if (scatter_events) (void)hipEventRecord(scatter_events[2 * pass + 1], stream);
        src = dst;                                                            // RadixSort.cpp:638-641
    }
    return src;
}

// ---------------------------------------------------------------------------------------------
// helpers for the stand-alone sorter entry points (gs_sort_host / gs_sort_bench)
// ---------------------------------------------------------------------------------------------
__global__ void k_set_sort_params(SortParams* params, uint32_t n) {
    params->counter = n;
    params->num_elems = n;
    params->num_groups = (n + kSortTile - 1) / kSortTile;
    params->num_segments = (params->num_groups + kSegGroups - 1) / kSegGroups;
    params->overflow = 0;
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void k_fill_random_keys(uint32_t* lo, uint32_t* hi, uint32_t* id, uint32_t n,
                                   uint32_t num_tiles, uint64_t seed) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t z = splitmix64(seed + i);
        lo[i] = (uint32_t)z;
        hi[i] = (uint32_t)((z >> 32) % num_tiles);
        id[i] = i;
    }
}

__global__ void k_check_sorted(const uint32_t* lo, const uint32_t* hi, uint32_t n, uint32_t* bad) {
    uint32_t local = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x + 1; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t a = ((uint64_t)hi[i - 1] << 32) | lo[i - 1];
        const uint64_t b = ((uint64_t)hi[i] << 32) | lo[i];
        local += a > b ? 1u : 0u;
    }
    if (local) atomicAdd(bad, local);
}

void launch_set_sort_params(SortParams* params, uint32_t n, hipStream_t stream) {
    hipLaunchKernelGGL(k_set_sort_params, dim3(1), dim3(1), 0, stream, params, n);
}
void launch_fill_random_keys(uint32_t* lo, uint32_t* hi, uint32_t* id, uint32_t n,
                             uint32_t num_tiles, uint64_t seed, hipStream_t stream) {
    hipLaunchKernelGGL(k_fill_random_keys, dim3(2048), dim3(256), 0, stream, lo, hi, id, n,
                       num_tiles, seed);
}
void launch_check_sorted(const uint32_t* lo, const uint32_t* hi, uint32_t n, uint32_t* bad_count,
                         hipStream_t stream) {
    hipLaunchKernelGGL(k_check_sorted, dim3(2048), dim3(256), 0, stream, lo, hi, n, bad_count);
}

} // namespace gs
