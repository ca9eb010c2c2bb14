// gs_sort8.hip -- GS_SORT_RADIX8 / GS_SORT_RADIX8_SPLAT_FIRST: the same stable LSD radix sort over the (tile << 32 | depth)
// keys as gs_sort.hip, with 8-bit digits -- half the passes, each of them the reference's five stages
// (RadixSort.cpp:309-642: Count -> Reduce -> Scan -> ScanAdd -> Scatter) in three launches.  The output of a stable LSD
// sort does not depend on the digit width, so lists, ranges and pixels are the ones of the contractual 4-bit pipeline
// (the A/B slot SURVEY.md 8(f)-4 asks for, behind the GpuSort seam, GpuSort.h:8-22).
//
//   k_count8    Count  : digit histogram of every group of kSort8Tile keys; a wave takes 2048 keys per step with 16-byte
//                        loads of the one word the digit lives in and adds into the group's 256 LDS counters
//                        (RadixSortCount.comp:40-91 with 256 bins; the depth passes add a run of equal digits in
//                        neighbouring lanes -- one splat's keys -- with one atomic, see count8_keys).
//               Reduce + ScanAdd inside a segment: workgroup s owns the contiguous groups of segment s; thread d walks
//                        them and leaves in table[group][d] the keys of digit d in the EARLIER groups of the segment
//                        (group-major: a row is 1 KB, written and later read as one line run) and the segment's
//                        totals in seg_sum[s][d], again one 1 KB row.
//   k_scan8     Scan   : workgroup d scans the 512 segment totals of digit d: seg_base[d][s] = keys of digit d in the
//                        segments before s, totals[d] = keys of digit d in the list (RadixSortScan.comp:29-71).
//   k_scatter8  ScanAdd, the rest: thread d adds seg_base[d][segment] and table[group][d]; the keys of all smaller
//                        digits come from a wave scan over totals[].
//               Scatter: one workgroup per group: wave64 match-mask ranking over the eight digit bits (stable), the
//                        waves' running digit counts in LDS, local sort into LDS, run-wise stores
//                        (RadixSortScatter.comp:58-171).  Groups are dealt to workgroups so that neighbouring groups
//                        share an XCD: a digit run is only 16 keys long on average, and the runs of neighbouring
//                        groups complete each other's cache lines inside one L2.
// Word widths inside a frame as in gs_sort.hip: 16-bit compact tile ids, depth words that shrink once their lower
// half is consumed (pass 0 moves all of it, pass 1 keeps the upper half, pass 2 sorts on that half, pass 3 drops it).
#include "gs_device_utils.h"
#include "gs_internal.h"

#include <type_traits>

namespace gs {

// The keys of a pass are read once: non-temporal loads keep them from displacing the partly written destination lines
// in L2, which neighbouring groups are about to complete (config C's RadixSort 0.590 -> 0.552 ms, config D's 1.59 ->
// 1.33 with the 4-bit passes; DESIGN.md section 4.1.  Non-temporal STORES, or such loads in Count, cost 10-80 %).
#define GS_KEY_LOAD(p) __builtin_nontemporal_load(p)

constexpr int kC8Chunk = 2048;                          // keys a Count wave takes per step
constexpr int kC8Waves = 8;
constexpr int kC8Threads = kC8Waves * 64;
constexpr int kC8MaxK = 32;                             // groups of a segment whose counters sit in LDS at once (32 KB)
static_assert(kSort8TileSmall % kC8Chunk == 0 && kSort8Tile % kC8Chunk == 0, "a group is a whole number of Count steps");

template <bool W16>
struct Count8Regs { uint4 v[W16 ? 4 : 8]; };            // 32 keys per lane

template <bool W16>
__device__ __forceinline__ void count8_load(const uint32_t* __restrict__ word, uint32_t chunk, uint32_t e, int lane,
                                            Count8Regs<W16>& k) {
    constexpr int V = W16 ? 4 : 8;
    constexpr uint32_t PER = W16 ? 8u : 4u;              // keys per 16-byte load
    const uint32_t first = chunk * kC8Chunk;
    if (first + kC8Chunk <= e) {
        const uint4* w4 = W16 ? reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(word) + first)
                              : reinterpret_cast<const uint4*>(word + first);
#pragma unroll
        for (int r = 0; r < V; ++r) k.v[r] = w4[r * 64 + lane];
    } else {   // ragged end of the list: element-wise, keys past the end are skipped by the bounds test of the count
#pragma unroll
        for (int r = 0; r < V; ++r) {
            const uint32_t i0 = first + (uint32_t)(r * 64 + lane) * PER;
            uint32_t w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (W16) {
                    const uint16_t* h = reinterpret_cast<const uint16_t*>(word);
                    const uint32_t i = i0 + 2u * (uint32_t)q;
                    w[q] = (i < e ? (uint32_t)h[i] : 0u) | ((i + 1u < e ? (uint32_t)h[i + 1u] : 0u) << 16);
                } else {
                    w[q] = i0 + (uint32_t)q < e ? word[i0 + q] : 0u;
                }
            }
            k.v[r] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

// RUNS (the passes over the depth word): the keys of one splat carry one depth and lie side by side in the list -- in
// InitSortList's order and, the sort being stable, after every depth pass -- so neighbouring lanes of a wave
// instruction (keys 4 or 8 apart) often hold the same digit, and same-address LDS atomics serialise.  The first lane of
// every run of equal digits (runs end at the 16-lane rows of the DPP shift) adds the run's length for all of them.
template <bool W16, bool FULL, bool RUNS>
__device__ __forceinline__ void count8_keys(const Count8Regs<W16>& k, uint32_t chunk, uint32_t e, int lane, uint32_t sh,
                                            uint32_t mask, uint32_t* hist) {
    constexpr int V = W16 ? 4 : 8;
    constexpr uint32_t PER = W16 ? 8u : 4u;
    const uint32_t first = chunk * kC8Chunk;
#pragma unroll
    for (int r = 0; r < V; ++r) {
        const uint32_t w[4] = {k.v[r].x, k.v[r].y, k.v[r].z, k.v[r].w};
        const uint32_t i0 = first + (uint32_t)(r * 64 + lane) * PER;
#pragma unroll
        for (uint32_t q = 0; q < PER; ++q) {
            const uint32_t key = W16 ? (w[q >> 1] >> (16u * (q & 1u))) & 0xFFFFu : w[q];
            const uint32_t d = (key >> sh) & mask;
            const bool ok = FULL || i0 + q < e;
            if constexpr (RUNS) {
                const uint32_t dd = ok ? d : 0xFFFFFFFFu;               // keys past the end: a run of their own, never added
                const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)~dd, (int)dd, 0x111, 0xf, 0xf, false);   // row_shr:1
                const uint64_t lead = __ballot(prev != dd);             // first lanes of the runs (lanes 0, 16, 32, 48 always)
                const uint64_t above = (lead >> 1) >> lane;
                const uint32_t n = above ? (uint32_t)__builtin_ctzll(above) + 1u : 64u - (uint32_t)lane;
                if (ok && prev != dd)
                    (void)__hip_atomic_fetch_add(&hist[d], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (ok) {
                (void)__hip_atomic_fetch_add(&hist[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_u32
            }
        }
    }
}

template <bool W16, bool RUNS, int TILE>
__global__ __launch_bounds__(kC8Threads) void k_count8(const SortParams* __restrict__ params,
                                                       const uint32_t* __restrict__ word, uint32_t* __restrict__ table,
                                                       uint32_t* __restrict__ seg_sum, uint32_t sh, uint32_t mask) {
    __shared__ uint32_t s_hist[kC8MaxK][kBins8];
    constexpr uint32_t kC8ChunksPerGroup = TILE / kC8Chunk;
    const uint32_t e = params->num_elems;
    const uint32_t G = (e + TILE - 1) / TILE, K = (G + kSegments - 1) / kSegments;
    const uint32_t chunks = (e + kC8Chunk - 1) / kC8Chunk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t grp0 = blockIdx.x * K;
    const uint32_t grp_end = grp0 + K < G ? grp0 + K : G;
    {   // the counters of the groups of one round (the walk below clears what it has read)
        const uint32_t rows = grp_end > grp0 ? (grp_end - grp0 < (uint32_t)kC8MaxK ? grp_end - grp0 : (uint32_t)kC8MaxK) : 0u;
        for (uint32_t i = tid; i < rows * kBins8; i += kC8Threads) (&s_hist[0][0])[i] = 0u;
    }
    __syncthreads();
    uint32_t run = 0;                                    // thread d: keys of digit d in the segment so far
    for (uint32_t b0 = grp0; b0 < grp_end; b0 += kC8MaxK) {   // one round for every list a frame can hold (E <= 67 M)
        const uint32_t b1 = b0 + kC8MaxK < grp_end ? b0 + kC8MaxK : grp_end;
        const uint32_t ch_end = b1 * kC8ChunksPerGroup < chunks ? b1 * kC8ChunksPerGroup : chunks;
        uint32_t ch = b0 * kC8ChunksPerGroup + (uint32_t)wave;
        Count8Regs<W16> cur;
        if (ch < ch_end) count8_load<W16>(word, ch, e, lane, cur);
        while (ch < ch_end) {
            const uint32_t nxt_ch = ch + kC8Waves;
            Count8Regs<W16> nxt;
            if (nxt_ch < ch_end) count8_load<W16>(word, nxt_ch, e, lane, nxt);   // in flight while this step is counted
            uint32_t* hist = s_hist[ch / kC8ChunksPerGroup - b0];
            if (ch * kC8Chunk + kC8Chunk <= e) count8_keys<W16, true, RUNS>(cur, ch, e, lane, sh, mask, hist);
            else count8_keys<W16, false, RUNS>(cur, ch, e, lane, sh, mask, hist);
            if (nxt_ch < ch_end) cur = nxt;
            ch = nxt_ch;
        }
        __syncthreads();
        if (tid < kBins8) {
            for (uint32_t j = b0; j < b1; ++j) {
                const uint32_t t = s_hist[j - b0][tid];
                s_hist[j - b0][tid] = 0u;
                table[(size_t)j * kBins8 + tid] = run;
                run += t;
            }
        }
        __syncthreads();
    }
    if (tid < kBins8) seg_sum[blockIdx.x * kBins8 + tid] = run;   // segment-major: one 1 KB row per workgroup; zero for empty segments
}

// Scan: workgroup d, thread s: the 512 segment totals of digit d -> seg_base[d][s] = keys of digit d in the segments
// before s, and totals[d] = keys of digit d in the whole list (k_scatter8 adds the smaller digits' totals itself).
__global__ __launch_bounds__(kSegments) void k_scan8(const uint32_t* __restrict__ seg_sum, uint32_t* __restrict__ seg_base,
                                                     uint32_t* __restrict__ totals) {
    static_assert(kSegments == 512, "eight waves, one segment per thread");
    __shared__ uint32_t s_w[8];
    const uint32_t d = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t v = seg_sum[tid * kBins8 + d];
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t base = 0u;
    for (int w = 0; w < wave; ++w) base += s_w[w];
    seg_base[d * kSegments + tid] = base + inc - v;
    if (tid == kSegments - 1) totals[d] = base + inc;
}

// ---------------------------------------------------------------------------------------------
// ScanAdd (rest) + Scatter, one workgroup per group of kSort8Tile keys, two barriers.
//   loads      striped as in k_scatter (key r of lane l of wave w = element w*R*64 + r*64 + l of the group); thread d
//              also fetches the global index of the group's first key of digit d (seg_base + table)
//   rank       per round of 64 keys: eight ballots give every lane the mask of lanes with the same digit; the wave's
//              running count of every digit lives in its own 256 LDS words (read by all lanes of the digit, advanced by
//              the first of them; DS operations of one wave execute in order)
//   barrier 1  every wave derives -- redundantly, lane l for digits 4l .. 4l+3 -- the local start of each digit (wave
//              scan over the four waves' totals) and its own base per digit, wave 0 also the global offset per digit
//   stage      keys + payload to their sorted local position in LDS
//   barrier 2
//   store      position p = r*threads + tid read back linearly, global index = offset[digit] + p
// ---------------------------------------------------------------------------------------------
template <int LO_IN, int LO_OUT, bool HI16, bool FULL, int R>
__device__ __forceinline__ void scatter8_group(
    uint32_t e, uint32_t seg, uint32_t grp, const uint32_t* __restrict__ in_lo, const uint32_t* __restrict__ in_hi,
    const uint32_t* __restrict__ in_id, uint32_t* __restrict__ out_lo, uint32_t* __restrict__ out_hi,
    uint32_t* __restrict__ out_id, const uint32_t* __restrict__ table, const uint32_t* __restrict__ seg_base,
    const uint32_t* __restrict__ totals, uint32_t shift, uint32_t mask, uint2* s_slot, typename std::conditional<HI16, uint16_t, uint32_t>::type* s_third,
    uint32_t* s_cnt, uint32_t* s_wbase, uint32_t* s_gpre, uint32_t* s_gofs, uint32_t* s_tot) {
    constexpr int NT = kSort8Threads, W = NT / 64, TILE = NT * R;
    using third_t = typename std::conditional<HI16, uint16_t, uint32_t>::type;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool use_hi = shift >= 32u;
    const uint32_t sh = use_hi ? shift - 32u : (LO_IN == 2 ? shift - 16u : shift);   // bit offset inside the stored word
    const uint32_t tile_base = grp * TILE;
    const uint32_t base = tile_base + (uint32_t)wave * (R * 64) + lane;
    const uint32_t valid = FULL ? (uint32_t)TILE : e - tile_base;

    uint32_t lo[R], hi[R], id[R];
    {
        const uint16_t* lo16 = reinterpret_cast<const uint16_t*>(in_lo) + base;
        const uint16_t* hi16 = reinterpret_cast<const uint16_t*>(in_hi) + base;
        const uint32_t *lo32 = in_lo + base, *hi32 = in_hi + base, *id32 = in_id + base;
#pragma unroll
        for (int r = 0; r < R; ++r) {   // coalesced: 64 consecutive elements per wave-instruction
            const bool ok = FULL || base + r * 64 < e;
            if constexpr (HI16) hi[r] = ok ? (uint32_t)GS_KEY_LOAD(&hi16[r * 64]) : 0xFFFFu;
            else hi[r] = ok ? GS_KEY_LOAD(&hi32[r * 64]) : 0xFFFFFFFFu;
            if constexpr (LO_IN == 4) lo[r] = ok ? GS_KEY_LOAD(&lo32[r * 64]) : 0xFFFFFFFFu;
            else if constexpr (LO_IN == 2) lo[r] = ok ? (uint32_t)GS_KEY_LOAD(&lo16[r * 64]) : 0xFFFFu;
            else lo[r] = 0u;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) id[r] = (FULL || base + r * 64 < e) ? GS_KEY_LOAD(&id32[r * 64]) : 0u;
    }
    // Scan + ScanAdd outputs for this group (L2-resident): keys ahead of the group's first key of digit tid
    uint32_t gpre = 0u, dtot = 0u;
    if (NT == kBins8 || tid < kBins8) {
        gpre = seg_base[tid * kSegments + seg] + table[(size_t)grp * kBins8 + tid];
        dtot = totals[tid];                        // keys of digit tid in the whole list
    }

    // the wave's running digit counts
    uint32_t* wcnt = s_cnt + wave * kBins8;
    *reinterpret_cast<uint4*>(&wcnt[4 * lane]) = make_uint4(0u, 0u, 0u, 0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    uint32_t rank[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool ok = FULL || base + r * 64 < e;
        const uint32_t dg = ((use_hi ? hi[r] : lo[r]) >> sh) & mask;
        uint32_t m_lo = 0xFFFFFFFFu, m_hi = 0xFFFFFFFFu;
        if (!FULL) { const uint64_t v = __ballot(ok); m_lo = (uint32_t)v; m_hi = (uint32_t)(v >> 32); }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (b >= 4 && (mask >> b) == 0u) break;                                     // a last pass of fewer bits (uniform)
            const int32_t sbit = __builtin_amdgcn_sbfe((int32_t)dg, (uint32_t)b, 1u);   // all ones where the bit is set
            const uint64_t bal = __ballot(sbit != 0);
            m_lo &= ~((uint32_t)bal ^ (uint32_t)sbit);                                  // lanes that agree on bit b
            m_hi &= ~((uint32_t)(bal >> 32) ^ (uint32_t)sbit);
        }
        uint64_t same = ((uint64_t)m_hi << 32) | m_lo;
        if (!FULL) same = ok ? same : 0ull;
        const uint32_t in_round = mbcnt(same);
        const uint32_t n_round = (uint32_t)__popcll(same);
        const uint32_t before = wcnt[dg];
        rank[r] = before + in_round;
        if ((FULL || ok) && in_round == 0u) wcnt[dg] = before + n_round;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (NT == kBins8 || tid < kBins8) { s_gpre[tid] = gpre; s_tot[tid] = dtot; }
    __syncthreads();

    // ---- local digit starts (every wave alike), this wave's bases, global offsets (wave 0)
    {
        uint4 tot = make_uint4(0u, 0u, 0u, 0u), mine = tot;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const uint4 c = *reinterpret_cast<const uint4*>(&s_cnt[w * kBins8 + 4 * lane]);
            tot.x += c.x; tot.y += c.y; tot.z += c.z; tot.w += c.w;
            if (w < wave) { mine.x += c.x; mine.y += c.y; mine.z += c.z; mine.w += c.w; }
        }
        const uint32_t lane_tot = tot.x + tot.y + tot.z + tot.w;
        const uint32_t d0 = wave_inclusive_scan(lane_tot) - lane_tot;      // first local position of digit 4 lane
        const uint4 dstart = make_uint4(d0, d0 + tot.x, d0 + tot.x + tot.y, d0 + tot.x + tot.y + tot.z);
        *reinterpret_cast<uint4*>(&s_wbase[wave * kBins8 + 4 * lane]) =
            make_uint4(dstart.x + mine.x, dstart.y + mine.y, dstart.z + mine.z, dstart.w + mine.w);
        if (wave == 0) {
            // Scan over the digits: keys of smaller digits anywhere in the list
            const uint4 t = *reinterpret_cast<const uint4*>(&s_tot[4 * lane]);
            const uint32_t lane_t = t.x + t.y + t.z + t.w;
            const uint32_t b0 = wave_inclusive_scan(lane_t) - lane_t;
            const uint4 g = *reinterpret_cast<const uint4*>(&s_gpre[4 * lane]);
            *reinterpret_cast<uint4*>(&s_gofs[4 * lane]) =                  // global index = s_gofs[digit] + local position
                make_uint4(b0 + g.x - dstart.x, b0 + t.x + g.y - dstart.y, b0 + t.x + t.y + g.z - dstart.z,
                           b0 + t.x + t.y + t.z + g.w - dstart.w);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // ---- local sort into LDS
    const uint32_t* wbase = s_wbase + wave * kBins8;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t dg = ((use_hi ? hi[r] : lo[r]) >> sh) & mask;
        const uint32_t p = wbase[dg] + rank[r];
        if (FULL || base + r * 64 < e) {
            if constexpr (LO_IN == 0) s_slot[p] = make_uint2(id[r], hi[r]);
            else if constexpr (LO_IN == 2 && HI16) s_slot[p] = make_uint2(id[r], lo[r] | (hi[r] << 16));
            else { s_slot[p] = make_uint2(id[r], lo[r]); s_third[p] = (third_t)hi[r]; }
        }
    }
    __syncthreads();

    // ---- stores: consecutive local positions of one digit are consecutive global indices
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t p = (uint32_t)r * NT + tid;
        if (FULL || p < valid) {
            const uint2 sl = s_slot[p];
            uint32_t l, h;
            if constexpr (LO_IN == 0) { l = 0u; h = sl.y; }
            else if constexpr (LO_IN == 2 && HI16) { l = sl.y & 0xFFFFu; h = sl.y >> 16; }
            else { l = sl.y; h = s_third[p]; }
            const uint32_t d = ((use_hi ? h : l) >> sh) & mask;
            const uint32_t o = s_gofs[d] + p;
            if constexpr (LO_OUT == 4) out_lo[o] = l;
            else if constexpr (LO_OUT == 2) reinterpret_cast<uint16_t*>(out_lo)[o] = (uint16_t)(LO_IN == 4 ? l >> 16 : l);
            if constexpr (HI16) reinterpret_cast<uint16_t*>(out_hi)[o] = (uint16_t)h;
            else out_hi[o] = h;
            out_id[o] = sl.x;
        }
    }
}

template <int LO_IN, int LO_OUT, bool HI16, int R>
__global__ __launch_bounds__(kSort8Threads)
void k_scatter8(const SortParams* __restrict__ params, const uint32_t* __restrict__ in_lo,
                const uint32_t* __restrict__ in_hi, const uint32_t* __restrict__ in_id,
                uint32_t* __restrict__ out_lo, uint32_t* __restrict__ out_hi, uint32_t* __restrict__ out_id,
                const uint32_t* __restrict__ table, const uint32_t* __restrict__ seg_base,
                const uint32_t* __restrict__ totals, uint32_t shift, uint32_t mask) {
    constexpr bool kThird = LO_IN == 4 || (LO_IN == 2 && !HI16);
    constexpr int W = kSort8Threads / 64, TILE = kSort8Threads * R;
    __shared__ uint2 s_slot[TILE];
    __shared__ typename std::conditional<HI16, uint16_t, uint32_t>::type s_third[kThird ? TILE : 1];
    __shared__ __attribute__((aligned(16))) uint32_t s_cnt[W * kBins8];
    __shared__ __attribute__((aligned(16))) uint32_t s_wbase[W * kBins8];
    __shared__ __attribute__((aligned(16))) uint32_t s_gpre[kBins8];
    __shared__ __attribute__((aligned(16))) uint32_t s_gofs[kBins8];
    __shared__ __attribute__((aligned(16))) uint32_t s_tot[kBins8];
    const uint32_t e = params->num_elems;
    const uint32_t G = (e + TILE - 1) / TILE, K = (G + kSegments - 1) / kSegments;
    // Workgroups b, b + 8, ... share an XCD (observed placement, speed only): each of the eight takes a contiguous run of
    // the groups, so that the short digit runs of neighbouring groups -- neighbours in the destination too -- meet in
    // one L2 and leave it as whole lines.
    const uint32_t per_xcd = (G + 7u) / 8u;
    bool again = false;
    for (uint32_t vb = blockIdx.x; vb < 8u * per_xcd; vb += gridDim.x) {
        const uint32_t grp = (vb & 7u) * per_xcd + (vb >> 3);
        if (grp >= G) continue;
        if (again) __syncthreads();   // LDS is reused
        again = true;
        const uint32_t seg = grp / K;
        if (grp * TILE + TILE <= e)
            scatter8_group<LO_IN, LO_OUT, HI16, true, R>(e, seg, grp, in_lo, in_hi, in_id, out_lo, out_hi, out_id, table, seg_base,
                                                      totals, shift, mask, s_slot, s_third, s_cnt, s_wbase, s_gpre, s_gofs, s_tot);
        else
            scatter8_group<LO_IN, LO_OUT, HI16, false, R>(e, seg, grp, in_lo, in_hi, in_id, out_lo, out_hi, out_id, table, seg_base,
                                                       totals, shift, mask, s_slot, s_third, s_cnt, s_wbase, s_gpre, s_gofs, s_tot);
    }
}

int launch_radix_sort8(const SortBuffers& sb, uint32_t capacity, uint32_t num_sort_bits, hipStream_t stream,
                       hipEvent_t* scatter_events, uint32_t first_bit, bool drop_depth_payload, bool hi16, float share,
                       int start, uint32_t coarse_pass, const SortParams* params) {
    if (sb.digit_bits != 8u) return -1;           // 4-bit buffers hold 16 x 512 segment words, not 2 x 256 x 512
    if (!params) params = sb.params;
    // Group size by what the list can be expected to hold (the host never reads the element count back): 2048-key groups for
    // short lists -- more workgroups, shorter latency chains -- 4096 for long ones (digit runs twice as long).
    const float bound = (float)capacity * (share < 0.5f ? share : 1.0f);   // a band holds about its share of the capacity
    const bool small = bound < (float)kSort8SmallBelow;
    const uint32_t tile = small ? (uint32_t)kSort8TileSmall : (uint32_t)kSort8Tile;
    uint32_t max_groups = (capacity + tile - 1) / tile;
    if (share < 0.5f) {   // a tile-row band: see launch_radix_sort
        const uint32_t g = (uint32_t)((float)max_groups * 2.0f * share) + 64u;
        max_groups = g < max_groups ? g : max_groups;
    }
    uint32_t* seg_base = sb.seg_sum + (size_t)kBins8 * kSegments;
    int src = start;
    uint32_t pass = 0;
    for (uint32_t shift = first_bit; shift < num_sort_bits; shift += 8u, ++pass) {
        const int dst = src ^ 1;
        const uint32_t bits = num_sort_bits - shift < 8u ? num_sort_bits - shift : 8u;
        const uint32_t mask = (1u << bits) - 1u;
        const bool tile_pass = shift >= 32u;
        const uint32_t* word = tile_pass ? sb.hi[src] : sb.lo[src];
        int lo_in, lo_out;
        scatter_depth_bytes(shift, first_bit, drop_depth_payload, &lo_in, &lo_out, 8u);
        const bool lo16 = !tile_pass && lo_in == 2;
        const bool word16 = (tile_pass && hi16) || lo16;
        // the pass's digit totals (k_scan8 -> k_scatter8): a slab of sb.coarse
        uint32_t* totals = sb.coarse + (size_t)(coarse_pass + pass) * kBins * kCoarse;
        const uint32_t sh = lo16 ? shift - 16u : shift & 31u;
#define GS_LAUNCH_COUNT8(W16, RUNS)                                                                                  \
        do { if (small) hipLaunchKernelGGL((k_count8<W16, RUNS, kSort8TileSmall>), dim3(kSegments), dim3(kC8Threads), 0, stream, \
                                           params, word, sb.table, sb.seg_sum, sh, mask);                            \
             else hipLaunchKernelGGL((k_count8<W16, RUNS, kSort8Tile>), dim3(kSegments), dim3(kC8Threads), 0, stream,  \
                                     params, word, sb.table, sb.seg_sum, sh, mask); } while (0)
        // Runs of equal digits in neighbouring keys: the depth digits wherever splats are replicated into tiles (not in the
        // splat list of the splat-first order, whose passes stop at bit 32), and the top tile digit -- the keys arrive
        // sorted by everything below it, so the long lists of a capture's heavy tiles lie in runs (C-hard: that Count
        // 37 -> 17 us with the runs added once; uniform fog pays 2.5 us for it).
        const bool runs = tile_pass ? shift + 8u >= num_sort_bits : num_sort_bits > 32u;
        if (runs) { if (word16) GS_LAUNCH_COUNT8(true, true); else GS_LAUNCH_COUNT8(false, true); }
        else { if (word16) GS_LAUNCH_COUNT8(true, false); else GS_LAUNCH_COUNT8(false, false); }
#undef GS_LAUNCH_COUNT8
        hipLaunchKernelGGL(k_scan8, dim3(kBins8), dim3(kSegments), 0, stream, sb.seg_sum, seg_base, totals);
        if (scatter_events) (void)hipEventRecord(scatter_events[2 * pass], stream);
#define GS_LAUNCH_SCATTER8_R(LO_IN, LO_OUT, HI16, R)                                                                   \
        hipLaunchKernelGGL((k_scatter8<LO_IN, LO_OUT, HI16, R>), dim3(max_groups), dim3(kSort8Threads), 0, stream, params, \
                           sb.lo[src], sb.hi[src], sb.id[src], sb.lo[dst], sb.hi[dst], sb.id[dst], sb.table, seg_base,    \
                           totals, shift, mask)
#define GS_LAUNCH_SCATTER8(LO_IN, LO_OUT, HI16)                                                                        \
        do { if (small) GS_LAUNCH_SCATTER8_R(LO_IN, LO_OUT, HI16, kSort8KeysSmall);                                   \
             else GS_LAUNCH_SCATTER8_R(LO_IN, LO_OUT, HI16, kSort8KeysPerThread); } while (0)
#define GS_LAUNCH_SCATTER8_H(LO_IN, LO_OUT) \
        do { if (hi16) GS_LAUNCH_SCATTER8(LO_IN, LO_OUT, true); else GS_LAUNCH_SCATTER8(LO_IN, LO_OUT, false); } while (0)
        if (lo_in == 4 && lo_out == 4) GS_LAUNCH_SCATTER8_H(4, 4);
        else if (lo_in == 4 && lo_out == 2) GS_LAUNCH_SCATTER8_H(4, 2);
        else if (lo_in == 2 && lo_out == 2) GS_LAUNCH_SCATTER8_H(2, 2);
        else if (lo_in == 2 && lo_out == 0) GS_LAUNCH_SCATTER8_H(2, 0);
        else GS_LAUNCH_SCATTER8_H(0, 0);
#undef GS_LAUNCH_SCATTER8_H
#undef GS_LAUNCH_SCATTER8
#undef GS_LAUNCH_SCATTER8_R
        if (scatter_events) (void)hipEventRecord(scatter_events[2 * pass + 1], stream);
        src = dst;
    }
    return src;
}

} // namespace gs
