// gs_sort.hip -- the reference's stable 4-bit LSD radix sort over 64-bit (tile<<32 | depth) keys
// with a 32-bit payload, re-designed for gfx950 (wave64, 256 CUs, HBM-bound).
//
// Reference pipeline per 4-bit pass (RadixSort.cpp:309-642): five dependent dispatches
//   Count -> Reduce -> Scan -> ScanAdd -> Scatter, 64 keys per workgroup, 16-byte uvec4 elements.
// Here the same five stages run in THREE launches per pass over 2048-key tiles ("groups"):
//   k_count    Count  : per-group digit histogram -> table[bin][group]  (RadixSortCount.comp:40-91);
//                       reads only the 4-byte key half the digit lives in (keys are SoA).
//              Reduce : each of the 1024 persistent workgroups owns a CONTIGUOUS run of groups (one
//                       reduce segment) and stores the run's 16 digit totals to seg_sum[bin][segment]
//                       with plain stores (RadixSortReduce.comp:34-72; device atomics here cost
//                       ~15 us per pass on MI355X, measured).
//   k_scan     Scan   : one workgroup, exclusive scan of the 16 x 1024 segment totals in bin-major
//                       order, in place (RadixSortScan.comp:29-71).
//   k_scatter  ScanAdd: prologue -- exclusive prefix of the group's counts inside its segment,
//                       read from the L2-resident table, plus the segment base
//                       (RadixSortScanAdd.comp:34-66);
//              Scatter: wave64 match-mask ranking (stable), LDS-staged local sort, run-wise
//                       coalesced stores (RadixSortScatter.comp:58-171).
// Count is persistent (1024 workgroups, each walking the groups of its segment and prefetching the next group's
// keys); Scatter launches one workgroup per group (five resident per CU).  Inside a frame the words are narrower
// than the reference's: 16-bit band-relative tile ids when they fit, and depth words that shrink as their digits are
// consumed (see k_scatter); the stand-alone sorter (gs_sort_host) always moves three 32-bit words.
// Output is bit-identical to a stable sort by the low num_sort_bits of the key.
// Launch grids are fixed; the device-side element count (SortParams, the IndirectSetup record)
// bounds every loop -- no host read-back inside a frame.
#include "gs_device_utils.h"
#include "gs_internal.h"

namespace gs {

__device__ __forceinline__ uint32_t digit_of(uint32_t word, uint32_t sh) { return (word >> sh) & 15u; }

constexpr int kSortWaves = kSortThreads / 64;

// ---------------------------------------------------------------------------------------------
// Count + Reduce
// ---------------------------------------------------------------------------------------------
#ifndef GS_SCATTER_GRID
#define GS_SCATTER_GRID 1048576 // cap on k_scatter's grid: above it workgroups walk several groups.  One workgroup per
                                // group (no cap in practice) measured faster than 1024 persistent ones once the
                                // payload shrank: 39.8 / 45.4 / 52.3 us against 45.3 / 50.9 / 58.7 for the 12 / 16 / 20-byte passes
#endif

// Count + Reduce.  Persistent workgroups walk the groups (tiles) with a stride of gridDim and
// prefetch the next group's keys while counting the current one, so HBM never idles between the
// load / count / store phases of a group.  16-byte coalesced loads (order inside the tile is
// irrelevant for a histogram); per-thread counters packed 8 x 8 bit in two 64-bit registers,
// widened to 16-bit fields and summed across the wave with six DPP adds per word; no LDS atomics.
static_assert(kSortKeysPerThread % 4 == 0 && kSortKeysPerThread <= 252, "packed 8-bit counters");
constexpr int kCountVec = kSortKeysPerThread / 4;

// HI16: `word` is an array of 16-bit tile ids (a group of kSortTile keys is kSortTile * 2 bytes: with 8 keys per
// thread exactly one 16-byte load, kept in v[0]).
template <bool HI16>
__device__ __forceinline__ void count_load(const uint32_t* __restrict__ word, uint32_t grp, uint32_t e,
                                           int tid, uint4 (&v)[kCountVec]) {
    const uint32_t tile_base = grp * kSortTile;
    if constexpr (HI16) {
        static_assert(kSortKeysPerThread % 8 == 0, "16-bit count path: 8 keys per 16-byte load");
        constexpr int V16 = kSortKeysPerThread / 8;   // 16-byte loads per thread, kept in v[0..V16)
        const uint16_t* h = reinterpret_cast<const uint16_t*>(word);
        if (tile_base + kSortTile <= e) {
            const uint4* w4 = reinterpret_cast<const uint4*>(h + tile_base);
#pragma unroll
            for (int r = 0; r < V16; ++r) v[r] = w4[r * kSortThreads + tid];
        } else {
#pragma unroll
            for (int r = 0; r < V16; ++r) {
                uint32_t w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t i0 = tile_base + (uint32_t)(r * kSortThreads + tid) * 8u + (uint32_t)q * 2u;
                    const uint32_t a = i0 < e ? h[i0] : 0u, b = i0 + 1 < e ? h[i0 + 1] : 0u;
                    w[q] = a | (b << 16);
                }
                v[r] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        return;
    }
    if (tile_base + kSortTile <= e) {
        const uint4* w4 = reinterpret_cast<const uint4*>(word + tile_base);
#pragma unroll
        for (int r = 0; r < kCountVec; ++r) v[r] = w4[r * kSortThreads + tid];
    } else {   // ragged last tile: element-wise, missing keys marked with an impossible pattern below
#pragma unroll
        for (int r = 0; r < kCountVec; ++r) {
            const uint32_t i0 = tile_base + (uint32_t)(r * kSortThreads + tid) * 4u;
            v[r].x = i0 + 0 < e ? word[i0 + 0] : 0u;
            v[r].y = i0 + 1 < e ? word[i0 + 1] : 0u;
            v[r].z = i0 + 2 < e ? word[i0 + 2] : 0u;
            v[r].w = i0 + 3 < e ? word[i0 + 3] : 0u;
        }
    }
}

// ABLATE is a tuning-only switch (gs_debug_count_bench): bit 1 drops the table store, bit 2 the
// counting itself.  The product always launches ABLATE = 0.
template <int ABLATE, bool HI16 = false>
__global__ __launch_bounds__(kSortThreads) void k_count(const SortParams* __restrict__ params,
                                                         const uint32_t* __restrict__ word,
                                                         uint32_t* __restrict__ table,
                                                         uint32_t* __restrict__ seg_sum,
                                                         uint32_t sh) {
    __shared__ uint32_t s_pack[2][kSortWaves][8];   // per-wave packed totals (two 16-bit counters per word)
    const uint32_t e = params->num_elems, G = params->num_groups, K = params->groups_per_seg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // segment = blockIdx.x owns groups [seg*K, min(seg*K + K, G))
    uint32_t grp = blockIdx.x * K;
    const uint32_t grp_end = (grp + K < G) ? grp + K : G;
    uint32_t seg_total = 0;     // threads 0..15: this segment's total of digit tid
    uint4 nxt[kCountVec];
    if (grp < grp_end) count_load<HI16>(word, grp, e, tid, nxt);
    for (int it = 0; grp < grp_end; ++grp, it ^= 1) {
        uint4 v[kCountVec];
#pragma unroll
        for (int r = 0; r < kCountVec; ++r) v[r] = nxt[r];
        if (grp + 1 < grp_end) count_load<HI16>(word, grp + 1, e, tid, nxt);   // prefetch
        const uint32_t tile_base = grp * kSortTile;
        const bool full = tile_base + kSortTile <= e;
        // per-lane counters: one 4-bit field per digit in a single 64-bit register (a lane sees at most
        // kSortKeysPerThread <= 15 keys of a group), so a key costs one shift and one 64-bit add
        static_assert(kSortKeysPerThread <= 15, "4-bit per-lane digit counters");
        uint64_t c = 0;
        if (ABLATE & 4) {
#pragma unroll
            for (int r = 0; r < kCountVec; ++r) c += v[r].x ^ v[r].y ^ v[r].z ^ v[r].w;
        } else if constexpr (HI16) {
#pragma unroll
            for (int r = 0; r < kSortKeysPerThread / 8; ++r) {
                const uint32_t k[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const uint32_t key = (k[q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
                    const uint32_t d = digit_of(key, sh);
                    const bool ok = full || tile_base + (uint32_t)(r * kSortThreads + tid) * 8u + (uint32_t)q < e;
                    c += (uint64_t)(ok ? 1u : 0u) << (d * 4u);
                }
            }
        } else
#pragma unroll
        for (int r = 0; r < kCountVec; ++r) {
            const uint32_t k[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t d = digit_of(k[q], sh);
                const bool ok = full || tile_base + (uint32_t)(r * kSortThreads + tid) * 4u + q < e;
                c += (uint64_t)(ok ? 1u : 0u) << (d * 4u);
            }
        }
        // widen to 16-bit fields (a wave total is at most 64 * kSortKeysPerThread < 65536): a[j] holds digits
        // j, j+4, j+8, j+12 in its four 16-bit fields
        const uint64_t m = 0x000F000F000F000Full;
        const uint64_t a[4] = {c & m, (c >> 4) & m, (c >> 8) & m, (c >> 12) & m};
        uint32_t w[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            w[2 * q] = wave_sum_to_lane63((uint32_t)a[q]);
            w[2 * q + 1] = wave_sum_to_lane63((uint32_t)(a[q] >> 32));
        }
        if (lane == 63) {
#pragma unroll
            for (int q = 0; q < 8; ++q) s_pack[it][wave][q] = w[q];
        }
        __syncthreads();   // s_pack is double-buffered, so one barrier per group is enough
        if (tid < kBins) {
            // digit d sits in u64 a[d & 3], 16-bit field d >> 2
            const int word = (tid & 3) * 2 + (tid >> 3);
            const int half = (tid >> 2) & 1;
            uint32_t t = 0;
#pragma unroll
            for (int k = 0; k < kSortWaves; ++k) t += (s_pack[it][k][word] >> (16 * half)) & 0xFFFFu;
            if (!(ABLATE & 2) || t == 0xFFFFFFFFu) table[tid * G + grp] = t; // RadixSortCount.comp:89, bin-major
            seg_total += t;
        }
    }
    if (tid < kBins) seg_sum[tid * kSegments + blockIdx.x] = seg_total;   // Reduce (zero for empty segments)
}

// ---------------------------------------------------------------------------------------------
// Scan: exclusive scan of seg_sum[16][kSegments] in bin-major order, in place; one workgroup of 1024
// threads, thread t owns 16*kSegments/1024 consecutive entries (16-byte loads).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_scan(uint32_t* __restrict__ seg_sum) {
    constexpr int PER = kBins * kSegments / 1024;      // consecutive entries per thread
    static_assert(PER % 4 == 0 && PER >= 4, "k_scan: 16-byte loads");
    __shared__ uint32_t s_wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint4* p = reinterpret_cast<uint4*>(seg_sum) + tid * (PER / 4);
    uint4 v[PER / 4];
#pragma unroll
    for (int q = 0; q < PER / 4; ++q) v[q] = p[q];
    uint32_t x[PER];
#pragma unroll
    for (int q = 0; q < PER / 4; ++q) { x[4 * q] = v[q].x; x[4 * q + 1] = v[q].y; x[4 * q + 2] = v[q].z; x[4 * q + 3] = v[q].w; }
    uint32_t run = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) { const uint32_t t = x[q]; x[q] = run; run += t; }
    const uint32_t inc = wave_inclusive_scan(run);
    if (lane == 63) s_wave_tot[wave] = inc;
    __syncthreads();
    uint32_t base = inc - run;
    for (int w = 0; w < wave; ++w) base += s_wave_tot[w];
#pragma unroll
    for (int q = 0; q < PER / 4; ++q)
        p[q] = make_uint4(x[4 * q] + base, x[4 * q + 1] + base, x[4 * q + 2] + base, x[4 * q + 3] + base);
}

// ---------------------------------------------------------------------------------------------
// Scan + ScanAdd (prologue) + Scatter.  Persistent workgroups, next group's keys prefetched into
// registers while the current group is ranked, staged and stored.
// ---------------------------------------------------------------------------------------------
struct ScatterKeys {
    uint32_t lo[kSortKeysPerThread], hi[kSortKeysPerThread], id[kSortKeysPerThread];
};

template <int LO_IN, bool HI16>
__device__ __forceinline__ void scatter_load(const uint32_t* __restrict__ in_lo,
                                             const uint32_t* __restrict__ in_hi,
                                             const uint32_t* __restrict__ in_id, uint32_t base,
                                             uint32_t e, ScatterKeys& k) {
#pragma unroll
    for (int r = 0; r < kSortKeysPerThread; ++r) {   // coalesced: 256 contiguous bytes per wave-instruction
        const uint32_t idx = base + r * 64;
        const bool ok = idx < e;
        if constexpr (LO_IN == 4) k.lo[r] = ok ? in_lo[idx] : 0xFFFFFFFFu;
        else if constexpr (LO_IN == 2) k.lo[r] = ok ? (uint32_t)reinterpret_cast<const uint16_t*>(in_lo)[idx] : 0xFFFFu;
        else k.lo[r] = 0u;
        if constexpr (HI16) k.hi[r] = ok ? (uint32_t)reinterpret_cast<const uint16_t*>(in_hi)[idx] : 0xFFFFu;
        else k.hi[r] = ok ? in_hi[idx] : 0xFFFFFFFFu;
        k.id[r] = ok ? in_id[idx] : 0u;
    }
}

// ScanAdd inputs of one group for this wave's four digits: the lane's share of the per-group counts of the earlier
// groups of the segment, and the scanned segment base.
struct ScanAddRegs {
    uint32_t pre[kBins / kSortWaves], base[kBins / kSortWaves];
};

__device__ __forceinline__ void scan_add_load(const uint32_t* __restrict__ table, const uint32_t* __restrict__ seg_base,
                                              uint32_t grp, uint32_t G, uint32_t K, int wave, int lane, ScanAddRegs& sa) {
    const uint32_t seg = grp / K, j = grp - seg * K;   // j < K (K <= 64 up to 134 M elements)
#pragma unroll
    for (int q = 0; q < kBins / kSortWaves; ++q) {
        const int d = wave * (kBins / kSortWaves) + q;
        uint32_t pre = 0;
        for (uint32_t l0 = 0; l0 < j; l0 += 64)
            pre += l0 + (uint32_t)lane < j ? table[d * G + seg * K + l0 + lane] : 0u;
        sa.pre[q] = pre;
        sa.base[q] = seg_base[d * kSegments + seg];
    }
}

// GS_SCATTER_ABLATE: timing-only builds (tools/build_variants.sh), never shipped.  bit 0: stores go to
// the tile's own range (no scatter pattern); bit 1: no global stores; bit 2: no ranking.
// Measured at E = 13.1 M (MI355X): full 71.7 us; identity stores 55.4; no stores 44.5; no ranking +
// identity stores 48.7; neither 18.3.  An XCD-contiguous walk of the groups (so that neighbouring
// runs meet in one L2) measured 74.5 us -- slower, not kept.
#ifndef GS_SCATTER_ABLATE
#define GS_SCATTER_ABLATE 0
#endif
#ifndef GS_SCATTER_PREFETCH
#define GS_SCATTER_PREFETCH 0   // 1: keep the next group's keys in registers while working on the current one (only
                                // useful with a persistent grid, see GS_SCATTER_GRID)
#endif
#ifndef GS_SCATTER_MINWAVES
#define GS_SCATTER_MINWAVES 5
#endif
// LO_IN / LO_OUT = bytes of the depth word read / written per element (4, 2 or 0).  The stand-alone sorter
// (gs_sort_host) and GS_SORT_TILE_BUCKET use <4, 4>: everything moves.  In a frame the depth word is needed only as a
// sort key -- FindRanges reads the tile words, RenderGaussians the ids, gs_debug_read rebuilds the sorted depth
// words from the ids -- so bits a pass has consumed are dead weight: passes 0-2 run <4, 4>, pass 3 writes only the
// upper half <4, 2>, passes 4-6 sort on that half <2, 2>, pass 7 (last depth digit) does not write it <2, 0>, and the
// tile-word passes run <0, 0>.
// HI16: the tile words are stored as 16-bit ids relative to the band's first tile (any grid of at most 65535 tiles):
// 2 bytes less read and 2 less written per element in every pass.
template <int LO_IN, int LO_OUT, bool HI16>
__global__ __launch_bounds__(kSortThreads, GS_SCATTER_MINWAVES) void k_scatter(
    const SortParams* __restrict__ params, const uint32_t* __restrict__ in_lo,
    const uint32_t* __restrict__ in_hi, const uint32_t* __restrict__ in_id,
    uint32_t* __restrict__ out_lo, uint32_t* __restrict__ out_hi, uint32_t* __restrict__ out_id,
    const uint32_t* __restrict__ table, const uint32_t* __restrict__ seg_base, uint32_t shift) {
    __shared__ uint32_t s_lo[kSortTile];
    __shared__ uint32_t s_hi[kSortTile];
    __shared__ uint32_t s_id[kSortTile];
    __shared__ uint32_t s_wcnt[kSortWaves][kBins];
    __shared__ uint32_t s_wbase[kSortWaves][kBins];
    __shared__ uint32_t s_gpre[kBins];   // ScanAdd: global index of the first key of digit d of this group
    __shared__ int32_t s_gbase[kBins];

    const uint32_t e = params->num_elems, G = params->num_groups, K = params->groups_per_seg;
    uint32_t grp = blockIdx.x;
    if (grp >= G) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool use_hi = shift >= 32u;
    const uint32_t sh = use_hi ? shift - 32u : (LO_IN == 2 ? shift - 16u : shift);   // bit offset inside the stored word
    const uint32_t wave_off = (uint32_t)wave * (kSortKeysPerThread * 64) + lane;

#if GS_SCATTER_PREFETCH
    ScatterKeys nxt;
    scatter_load<LO_IN, HI16>(in_lo, in_hi, in_id, grp * kSortTile + wave_off, e, nxt);
#endif
    ScanAddRegs sa_nxt;
    scan_add_load(table, seg_base, grp, G, K, wave, lane, sa_nxt);

    for (; grp < G; grp += gridDim.x) {
#if GS_SCATTER_PREFETCH
        ScatterKeys k = nxt;
        if (grp + gridDim.x < G)
            scatter_load<LO_IN, HI16>(in_lo, in_hi, in_id, (grp + gridDim.x) * kSortTile + wave_off, e, nxt); // prefetch
#else
        ScatterKeys k;
        scatter_load<LO_IN, HI16>(in_lo, in_hi, in_id, grp * kSortTile + wave_off, e, k);
#endif
        const ScanAddRegs sa = sa_nxt;
        if (grp + gridDim.x < G) scan_add_load(table, seg_base, grp + gridDim.x, G, K, wave, lane, sa_nxt);
        const uint32_t tile_base = grp * kSortTile;
        const uint32_t base = tile_base + wave_off;

        // ---- ScanAdd: keys of digit d in all groups before this one = scanned segment base +
        //      counts of the earlier groups of the same segment (wave w: digits 4w..4w+3).  The table
        //      reads were issued one group ahead (scan_add_load), only the wave sums happen here.
#pragma unroll
        for (int q = 0; q < kBins / kSortWaves; ++q) {
            const int d = wave * (kBins / kSortWaves) + q;
            const uint32_t pre = wave_sum_to_lane63(sa.pre[q]);
            if (lane == 63) s_gpre[d] = pre + sa.base[q];
        }
        // ---- stable rank inside the wave.  Per round: 4 ballots give every lane the mask of lanes
        //      holding the same digit; lane d (d < 16) keeps the wave's running count of digit d.
        uint32_t rank[kSortKeysPerThread];
        uint32_t cntreg = 0;
#pragma unroll
        for (int r = 0; r < kSortKeysPerThread; ++r) {
            const uint32_t idx = base + r * 64;
            const bool ok = idx < e;
            const uint32_t dg = digit_of(use_hi ? k.hi[r] : k.lo[r], sh);
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
#if GS_SCATTER_ABLATE & 4
            rank[r] = (uint32_t)(wave * kSortKeysPerThread + r) * 64u + lane;   // linear position
#endif
        }
        if (lane < kBins) s_wcnt[wave][lane] = cntreg;
        __syncthreads();

        // ---- local digit starts, per-wave bases, global base (threads 0..15, one per digit)
        if (tid < kBins) {
            uint32_t c[kSortWaves];
            uint32_t tot = 0;
#pragma unroll
            for (int w = 0; w < kSortWaves; ++w) { c[w] = s_wcnt[w][tid]; tot += c[w]; }
            uint32_t inc = tot;
#pragma unroll
            for (int off = 1; off < kBins; off <<= 1) {
                const uint32_t t = __shfl_up(inc, off, 64);
                if (tid >= off) inc += t;
            }
            const uint32_t dstart = inc - tot;       // first local position of digit d
            uint32_t run = dstart;
#pragma unroll
            for (int w = 0; w < kSortWaves; ++w) { s_wbase[w][tid] = run; run += c[w]; }
            s_gbase[tid] = (int32_t)(s_gpre[tid] - dstart); // global = s_gbase[d] + local pos
        }
        __syncthreads();

        // ---- local sort into LDS
#pragma unroll
        for (int r = 0; r < kSortKeysPerThread; ++r) {
            const uint32_t idx = base + r * 64;
            if (idx < e) {
                const uint32_t dg = digit_of(use_hi ? k.hi[r] : k.lo[r], sh);
#if GS_SCATTER_ABLATE & 4
                const uint32_t p = rank[r];
#else
                const uint32_t p = s_wbase[wave][dg] + rank[r];
#endif
                if constexpr (LO_IN != 0) s_lo[p] = k.lo[r];
                s_hi[p] = k.hi[r];
                s_id[p] = k.id[r];
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
                const uint32_t l = LO_IN != 0 ? s_lo[p] : 0u, h = s_hi[p];
                const uint32_t d = digit_of(use_hi ? h : l, sh);
#if GS_SCATTER_ABLATE & 8
                const uint32_t o = tile_base + p + 13u + (d & 0u) < e ? tile_base + p + 13u : p;   // contiguous but misaligned
#elif GS_SCATTER_ABLATE & 5
                const uint32_t o = tile_base + p + (d & 0u);
#else
                const uint32_t o = (uint32_t)(s_gbase[d] + (int32_t)p);
#endif
#if GS_SCATTER_ABLATE & 2
                if (l == 0x12345678u && h == 0x9abcdef0u) out_lo[o] = l;   // keeps the pipeline alive, ~never taken
#else
                if constexpr (LO_OUT == 4) out_lo[o] = l;
                else if constexpr (LO_OUT == 2) reinterpret_cast<uint16_t*>(out_lo)[o] = (uint16_t)(LO_IN == 4 ? l >> 16 : l);
                if constexpr (HI16) reinterpret_cast<uint16_t*>(out_hi)[o] = (uint16_t)h;
                else out_hi[o] = h;
                out_id[o] = s_id[p];
#endif
            }
        }
        __syncthreads();   // LDS is reused by the next group
    }
}

int launch_radix_sort(const SortBuffers& sb, uint32_t capacity, uint32_t num_sort_bits,
                      hipStream_t stream, hipEvent_t* scatter_events, uint32_t first_bit,
                      bool drop_depth_payload, bool hi16) {
    const uint32_t max_groups = (capacity + kSortTile - 1) / kSortTile;
    int src = 0;
    uint32_t pass = 0;
    for (uint32_t shift = first_bit; shift < num_sort_bits; shift += kRadixBits, ++pass) { // RadixSort.cpp:309
        const int dst = src ^ 1;
        const bool tile_pass = shift >= 32u;
        const uint32_t* word = tile_pass ? sb.hi[src] : sb.lo[src];
        // 16-bit words: the tile ids of a band (hi16) and, in a frame, the upper half of the depth word once the
        // lower half is consumed (passes 4-7, see k_scatter)
        int cin, cout;
        scatter_depth_bytes(shift, first_bit, drop_depth_payload, &cin, &cout);
        const bool lo16 = !tile_pass && cin == 2;
        const bool word16 = kHi16Supported && ((tile_pass && hi16) || lo16);
        if constexpr (kHi16Supported) {
            if (word16)
                hipLaunchKernelGGL((k_count<0, true>), dim3(kSegments), dim3(kSortThreads), 0, stream, sb.params,
                                   word, sb.table, sb.seg_sum, lo16 ? shift - 16u : shift & 31u);
        }
        if (!word16)
            hipLaunchKernelGGL((k_count<0, false>), dim3(kSegments), dim3(kSortThreads), 0, stream, sb.params,
                               word, sb.table, sb.seg_sum, shift & 31u);
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, stream, sb.seg_sum);
        if (scatter_events) (void)hipEventRecord(scatter_events[2 * pass], stream);
        // bytes of the depth word read / written by this pass (see k_scatter)
        int lo_in, lo_out;
        scatter_depth_bytes(shift, first_bit, drop_depth_payload, &lo_in, &lo_out);
        const uint32_t pgrid = max_groups < (uint32_t)GS_SCATTER_GRID ? max_groups : (uint32_t)GS_SCATTER_GRID;
#define GS_LAUNCH_SCATTER(LO_IN, LO_OUT, HI16)                                                                       \
        hipLaunchKernelGGL((k_scatter<LO_IN, LO_OUT, HI16>), dim3(pgrid), dim3(kSortThreads), 0, stream, sb.params, \
                           sb.lo[src], sb.hi[src], sb.id[src], sb.lo[dst], sb.hi[dst], sb.id[dst],                  \
                           sb.table, sb.seg_sum, shift)
#define GS_LAUNCH_SCATTER_H(LO_IN, LO_OUT) \
        do { if (hi16) GS_LAUNCH_SCATTER(LO_IN, LO_OUT, true); else GS_LAUNCH_SCATTER(LO_IN, LO_OUT, false); } while (0)
        if (lo_in == 4 && lo_out == 4) GS_LAUNCH_SCATTER_H(4, 4);
        else if (lo_in == 4 && lo_out == 2) GS_LAUNCH_SCATTER_H(4, 2);
        else if (lo_in == 2 && lo_out == 2) GS_LAUNCH_SCATTER_H(2, 2);
        else if (lo_in == 2 && lo_out == 0) GS_LAUNCH_SCATTER_H(2, 0);
        else GS_LAUNCH_SCATTER_H(0, 0);
#undef GS_LAUNCH_SCATTER_H
#undef GS_LAUNCH_SCATTER
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
    params->groups_per_seg = (params->num_groups + kSegments - 1) / kSegments;
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

void launch_count_ablate(int ablate, const SortBuffers& sb, uint32_t capacity, uint32_t grid, hipStream_t stream) {
    (void)capacity; (void)grid;
    switch (ablate) {
        case 0: hipLaunchKernelGGL(k_count<0>, dim3(kSegments), dim3(kSortThreads), 0, stream, sb.params, sb.lo[0], sb.table, sb.seg_sum, 0u); break;
        case 3: hipLaunchKernelGGL(k_count<2>, dim3(kSegments), dim3(kSortThreads), 0, stream, sb.params, sb.lo[0], sb.table, sb.seg_sum, 0u); break;
        default: hipLaunchKernelGGL(k_count<6>, dim3(kSegments), dim3(kSortThreads), 0, stream, sb.params, sb.lo[0], sb.table, sb.seg_sum, 0u); break;
    }
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
