// gs_sort.hip -- the reference's stable 4-bit LSD radix sort over 64-bit (tile<<32 | depth) keys
// with a 32-bit payload, re-designed for gfx950 (wave64, 256 CUs, HBM-bound).
//
// Reference pipeline per 4-bit pass (RadixSort.cpp:309-642): five dependent dispatches
//   Count -> Reduce -> Scan -> ScanAdd -> Scatter, 64 keys per workgroup, 16-byte uvec4 elements.
// Here the same five stages run in TWO launches per pass over 2048-key tiles ("groups"):
//   k_count    Count  : per-group digit histogram, one WAVE per group (RadixSortCount.comp:40-91); reads only the
//                       word the digit lives in (keys are SoA; 2 bytes per key once that word is 16 bits wide).
//              Reduce : workgroup s (512 threads, kSegments = 512 of them) owns reduce segment s = K consecutive
//                       groups; their 16 digit totals go to seg_sum[bin][segment] with plain stores and, one level
//                       up, into 16 x 64 coarse totals with one agent-scope atomic add per digit
//                       (RadixSortReduce.comp:34-72, two levels).
//              ScanAdd, segment-local part: the table gets, per group and digit, the keys of that digit in the EARLIER
//                       groups of the segment (RadixSortScanAdd.comp:34-66).
//   k_scatter  Scan   : prologue -- every workgroup scans the coarse totals itself (DPP row scans) and adds the
//                       segment totals and the table entry ahead of its group (RadixSortScan.comp:29-71 and the rest
//                       of ScanAdd); there is no Scan launch.
//              Scatter: one 256-thread workgroup per group (5-6 resident per CU): wave64 match-mask ranking (stable),
//                       LDS-staged local sort, run-wise coalesced stores (RadixSortScatter.comp:58-171).
// Short lists (at most kFedMaxGroups groups; gs_config.count_launches) sort with ONE k_count launch: every Scatter counts the next
// pass's digit of the keys it stores and feeds the next pass's per-group count rows (k_scatter<.., FED>, "fed counts").
// Inside a frame the words are narrower than the reference's: 16-bit compact tile ids when they fit, and depth words
// that shrink as their digits are consumed (see k_scatter); the stand-alone sorter (gs_sort_host) always moves three
// 32-bit words.  Output is bit-identical to a stable sort by the low num_sort_bits of the key.
// Launch grids are fixed; the device-side element count (SortParams, the IndirectSetup record)
// bounds every loop -- no host read-back inside a frame.
#include "gs_device_utils.h"
#include "gs_internal.h"

#include <type_traits>

namespace gs {

// The keys of a pass are read once: non-temporal loads keep them from displacing the partly written destination lines
// in L2, which neighbouring groups are about to complete (config C's RadixSort 0.590 -> 0.552 ms, config D's 1.59 ->
// 1.33 with the 4-bit passes; DESIGN.md section 4.1.  Non-temporal STORES, or such loads in Count, cost 10-80 %).
#define GS_KEY_LOAD(p) __builtin_nontemporal_load(p)

__device__ __forceinline__ uint32_t digit_of(uint32_t word, uint32_t sh) { return (word >> sh) & 15u; }

constexpr int kSortWaves = kSortThreads / 64;

// ---------------------------------------------------------------------------------------------
// Count + Reduce.  Workgroup s owns reduce segment s = the contiguous groups [s K, s K + K); its eight waves take
// the groups round-robin, ONE WAVE PER GROUP: a lane reads 32 keys of the group with 16-byte loads (order inside a
// group is irrelevant for a histogram), the next group's loads are in flight while the current one is counted, and
// nothing crosses waves until the segment totals at the very end (one barrier per workgroup instead of one per
// group).  Per-lane counters: sixteen 4-bit fields of one 64-bit register per 8 keys (a key = one shift + one add),
// widened into 16-bit fields and summed over the wave with DPP adds; no LDS atomics.
// W16: the word the digit lives in is stored as 16 bits (tile ids of a frame; the upper depth half in passes 4-7).
// ---------------------------------------------------------------------------------------------
// inclusive prefix sum inside each row of 16 lanes (DPP row_shr 1/2/4/8; lanes shifted in from outside the row add 0)
__device__ __forceinline__ uint32_t row16_inclusive_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false);   // row_shr:8
    return v;
}

constexpr int kCountKeysPerLane = kSortTile / 64;   // 32
constexpr int kCountMaxK = 128;                     // groups per segment whose counts are kept in LDS (8 KB)
#ifndef GS_COUNT_WAVES
#define GS_COUNT_WAVES 8
#endif
constexpr int kCountWaves = GS_COUNT_WAVES;         // waves of a Count workgroup = groups of the segment counted side by side
                                                    // (8 measured 1.6 % better than 4 on config C's sort, 16 no better)
constexpr int kCountThreads = kCountWaves * 64;
static_assert(kCountKeysPerLane % 8 == 0, "k_count consumes the group in chunks of 8 keys per lane");

template <bool W16>
struct CountRegs { uint4 v[kCountKeysPerLane / (W16 ? 8 : 4)]; };

template <bool W16>
__device__ __forceinline__ void count_load(const uint32_t* __restrict__ word, uint32_t grp, uint32_t e, int lane,
                                           CountRegs<W16>& k) {
    constexpr int V = kCountKeysPerLane / (W16 ? 8 : 4);
    constexpr uint32_t PER = W16 ? 8u : 4u;          // keys per 16-byte load
    const uint32_t tile_base = grp * kSortTile;
    if (tile_base + kSortTile <= e) {
        const uint4* w4 = W16 ? reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(word) + tile_base)
                              : reinterpret_cast<const uint4*>(word + tile_base);
#pragma unroll
        for (int r = 0; r < V; ++r) k.v[r] = w4[r * 64 + lane];
    } else {   // ragged last group: element-wise; keys past the end are skipped by the bounds test of the count
#pragma unroll
        for (int r = 0; r < V; ++r) {
            const uint32_t i0 = tile_base + (uint32_t)(r * 64 + lane) * PER;
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

// Digit histogram of the wave's group into four 64-bit registers of 16-bit fields: a[j] holds digits j, j+4, j+8, j+12.
template <bool W16, bool FULL>
__device__ __forceinline__ void count_keys(const CountRegs<W16>& k, uint32_t grp, uint32_t e, int lane, uint32_t sh,
                                           uint64_t (&a)[4]) {
    constexpr int V = kCountKeysPerLane / (W16 ? 8 : 4);
    constexpr uint32_t PER = W16 ? 8u : 4u;
    constexpr int CH = W16 ? 1 : 2;                 // 16-byte loads per chunk of 8 keys
    const uint64_t m = 0x000F000F000F000Full;
    const uint32_t tile_base = grp * kSortTile;
#pragma unroll
    for (int ch = 0; ch < V / CH; ++ch) {
        uint64_t c = 0;                             // sixteen 4-bit counters, at most 8 keys each
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int r = ch * CH + u;
            const uint32_t w[4] = {k.v[r].x, k.v[r].y, k.v[r].z, k.v[r].w};
            const uint32_t i0 = tile_base + (uint32_t)(r * 64 + lane) * PER;
#pragma unroll
            for (uint32_t q = 0; q < PER; ++q) {
                const uint32_t key = W16 ? (w[q >> 1] >> (16u * (q & 1u))) & 0xFFFFu : w[q];
                const uint32_t d = digit_of(key, sh);
                const bool ok = FULL || i0 + q < e;
                c += (uint64_t)(ok ? 1u : 0u) << (d * 4u);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] += (c >> (4 * j)) & m;
    }
}

template <bool W16>
__global__ __launch_bounds__(kCountThreads) void k_count(const SortParams* __restrict__ params,
                                                         const uint32_t* __restrict__ word,
                                                         uint32_t* __restrict__ table,
                                                         uint32_t* __restrict__ seg_sum,
                                                         uint32_t* __restrict__ coarse,
                                                         uint32_t sh, uint32_t* __restrict__ fed_rows,
                                                         uint32_t* __restrict__ fed_zero) {
    __shared__ uint32_t s_pack[kCountWaves][8];      // wave-private: packed totals of the group the wave just counted
    __shared__ uint32_t s_hist[kCountMaxK][kBins];  // digit counts of the segment's groups (the first kCountMaxK of them)
    const uint32_t e = params->num_elems, G = params->num_groups, K = params->groups_per_seg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // segment = blockIdx.x owns groups [seg*K, min(seg*K + K, G)); wave w takes groups seg*K + w, + 4, ...
    const uint32_t grp_end = (blockIdx.x * K + K < G) ? blockIdx.x * K + K : G;
    const uint32_t grp0 = blockIdx.x * K;
    uint32_t grp = grp0 + (uint32_t)wave;
    CountRegs<W16> cur;
    if (grp < grp_end) count_load<W16>(word, grp, e, lane, cur);
    while (grp < grp_end) {
        const uint32_t nxt_grp = grp + kCountWaves;
        CountRegs<W16> nxt;
        if (nxt_grp < grp_end) count_load<W16>(word, nxt_grp, e, lane, nxt);   // in flight while this group is counted
        uint64_t a[4] = {0, 0, 0, 0};
        if (grp * kSortTile + kSortTile <= e) count_keys<W16, true>(cur, grp, e, lane, sh, a);
        else count_keys<W16, false>(cur, grp, e, lane, sh, a);
        uint32_t w[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                // a wave total is at most 2048: fits the 16-bit fields
            w[2 * q] = wave_sum_to_lane63((uint32_t)a[q]);
            w[2 * q + 1] = wave_sum_to_lane63((uint32_t)(a[q] >> 32));
        }
        // lane 63 holds the totals; lanes 0..15 pick theirs up through the wave's own LDS words (one wave writes and
        // reads them, DS operations of a wave execute in order: no barrier)
        if (lane == 63) {
#pragma unroll
            for (int q = 0; q < 8; ++q) s_pack[wave][q] = w[q];
        }
        __builtin_amdgcn_wave_barrier();   // no code: keeps the compiler from moving the reads below above the writes
        if (lane < kBins) {
            // digit d sits in u64 a[d & 3], 16-bit field d >> 2
            const int widx = (lane & 3) * 2 + (lane >> 3);
            const int half = (lane >> 2) & 1;
            const uint32_t t = (s_pack[wave][widx] >> (16 * half)) & 0xFFFFu;
            const uint32_t j = grp - grp0;
            if (fed_rows) {                          // fed counts (see k_scatter<.., FED>): the raw row of the group, and
                fed_rows[grp * kBins + lane] = t;    // the row the first Scatter adds into starts from zero
                fed_zero[grp * kBins + lane] = 0u;
            } else if (j < (uint32_t)kCountMaxK) s_hist[j][lane] = t;
            else table[lane * G + grp] = t;          // segments of more than kCountMaxK groups (E > 268 M): raw counts, see below
        }
        __builtin_amdgcn_wave_barrier();   // ... nor the next group's writes above these reads
        if (nxt_grp < grp_end) cur = nxt;
        grp = nxt_grp;
    }
    if (fed_rows) return;            // kernel-uniform: no Reduce / ScanAdd structure in this mode
    __syncthreads();
    // ScanAdd inside the segment (RadixSortScanAdd.comp:34-66), here rather than in every Scatter workgroup: the table
    // gets, per group and digit, the number of keys of that digit in the EARLIER groups of the segment (bin-major,
    // RadixSortCount.comp:89); the segment totals go to the reduce buffer (RadixSortReduce.comp:34-72).
    const uint32_t n_grp = grp_end > grp0 ? grp_end - grp0 : 0u;
    if (n_grp <= (uint32_t)kCountMaxK) {
        // every list below 268 M elements: a row of 16 lanes per digit scans the segment's groups sixteen at a time (DPP
        // row scan + the carry of the chunks before) -- four steps where one thread per digit used to walk K groups, one
        // LDS round trip each (K = 13 at config C, 32 at D, 58 at E)
        if (tid < kBins * 16) {
            const uint32_t d = (uint32_t)tid >> 4, jj = (uint32_t)tid & 15u;
            uint32_t carry = 0;
            for (uint32_t cb = 0; cb < n_grp; cb += 16u) {           // workgroup-uniform trip count
                const uint32_t j = cb + jj;
                const uint32_t t = j < n_grp ? s_hist[j][d] : 0u;
                const uint32_t inc = row16_inclusive_scan(t);
                if (j < n_grp) table[d * G + grp0 + j] = carry + inc - t;
                carry += (uint32_t)__shfl((int)inc, (lane & ~15) | 15, 64);     // the chunk's total: lane 15 of the row
            }
            if (jj == 0u) {
                seg_sum[d * kSegments + blockIdx.x] = carry;   // zero for empty segments
                // Reduce, second level: kCoarse coarse segments of kSegments / kCoarse segments each, summed with one
                // agent-scope atomic add per digit and workgroup (no return value, nothing waits for it; the launch boundary
                // publishes it).  Scatter's prologue scans these 16 x kCoarse totals itself -- there is no Scan launch.
                if (carry != 0u)
                    (void)__hip_atomic_fetch_add(&coarse[d * kCoarse + blockIdx.x / (kSegments / kCoarse)], carry,
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else if (tid < kBins) {        // segments of more than kCountMaxK groups: the counts beyond lie in the table itself
        uint32_t run = 0;
        for (uint32_t j = 0; j < n_grp; ++j) {
            const uint32_t t = j < (uint32_t)kCountMaxK ? s_hist[j][tid] : table[tid * G + grp0 + j];
            table[tid * G + grp0 + j] = run;
            run += t;
        }
        seg_sum[tid * kSegments + blockIdx.x] = run;
        if (run != 0u)
            (void)__hip_atomic_fetch_add(&coarse[tid * kCoarse + blockIdx.x / (kSegments / kCoarse)], run,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------
// ScanAdd (prologue) + Scatter: one workgroup per group of kSortTile keys, two barriers.
//
//   loads      the group's keys and payload, striped (key r of lane l of wave w is element w*512 + r*64 + l of the
//              group: every load instruction of a wave covers 64 consecutive elements), all issued up front
//              together with the ScanAdd inputs
//   ScanAdd    every wave, redundantly, in registers: lane d + 16 q sums the table entries l = q, q+4, ... of digit
//              d for the earlier groups of the segment; two cross-row adds and the scanned segment base give the
//              global index of the group's first key of digit d in lanes d, d+16, d+32, d+48
//   rank       per round of 64 keys: 4 ballots give every lane the mask of lanes holding the same digit (stable
//              rank inside the round = v_mbcnt of it); lane d keeps the wave's running count of digit d
//   barrier 1  the four waves' digit counts meet in LDS; every wave derives, again redundantly in lanes 0..15 (DPP
//              scan inside the 16-lane row), the local start of each digit, its own base and the global offset
//   stage      keys + payload to their sorted local position in LDS, one 8-byte slot per element (plus a third word
//              when 12 or more bytes travel)
//   barrier 2
//   store      position p = r*256 + tid read back linearly; consecutive local positions of one digit are
//              consecutive global indices (RadixSortScatter.comp:153-168): run-wise coalesced stores
// The group count (not the list capacity) bounds the work: surplus workgroups leave at once.
// ---------------------------------------------------------------------------------------------
#ifndef GS_SCATTER_MINWAVES_KEY
#define GS_SCATTER_MINWAVES_KEY 5    // resident workgroups per CU asked of the compiler: passes that carry 12 or more bytes
#endif
#ifndef GS_SCATTER_MINWAVES_SMALL
#define GS_SCATTER_MINWAVES_SMALL 5  // ... passes whose element fits one 8-byte LDS slot
#endif

constexpr uint32_t kSortTileLog2 = 11;
static_assert((1u << kSortTileLog2) == (uint32_t)kSortTile, "destination group of an element = index >> kSortTileLog2");
__device__ __forceinline__ void fed_zero_row(uint32_t* rows, uint32_t grp, int d) { rows[grp * kBins + d] = 0u; }


// LO_IN / LO_OUT = bytes of the depth word read / written per element (4, 2 or 0).  The stand-alone sorter
// (gs_sort_host) and GS_SORT_TILE_BUCKET use <4, 4>: everything moves.  In a frame the depth word is needed only as a
// sort key -- FindRanges reads the tile words, RenderGaussians the ids, gs_debug_read rebuilds the sorted depth
// words from the ids -- so bits a pass has consumed are dead weight: passes 0-2 run <4, 4>, pass 3 writes only the
// upper half <4, 2>, passes 4-6 sort on that half <2, 2>, pass 7 (last depth digit) does not write it <2, 0>, and the
// tile-word passes run <0, 0>.
// HI16: the tile words are 16-bit compact tile ids (at most 65535 owned tiles): 2 bytes less read and 2 less written
// per element in every pass.
// FULL: the group holds kSortTile valid keys (every group but the last): no per-element bounds logic.
// FED: "fed counts" -- the pass has no Count launch of its own.  Every Scatter workgroup of pass p counts, per digit run it
// stores and per destination group the run reaches (at most two: a run is at most one group long), the NEXT digit of the
// keys it stores (LDS atomics while they pass through the store loop) and adds those <= 32 rows of 16 counts to the rows
// of the destination groups, rows[g][16], with agent-scope atomics (16 lanes = one 64-byte request; a destination row
// receives ~16-32 such requests, so no line is a hot spot -- which is why there are no coarser levels: a level that
// sums S groups would take 16 S requests per line).  The prologue of pass p + 1 therefore sums the rows of ALL groups
// itself (those ahead of its own for the prefix, all of them for the digit totals): G x 64 bytes per workgroup out of
// L2, affordable for lists of up to ~1000 groups (a tile-row band of a multi-GPU frame, config A), where a Count launch
// is one fixed ~6-8 us latency chain per pass (profiles/r06_fed_probe.txt, r06_fed_kernels_band8_configC.txt).  Three row sets rotate: pass p reads set p % 3, adds into (p + 1) % 3 and
// clears its own row of (p + 2) % 3; pass 0's rows come from k_count (one launch per sort instead of one per pass).
// In this mode the kernel's `table` argument is the row set read, `seg_sum` the set added into (null in the last
// pass) and `coarse` the set cleared.
template <int LO_IN, int LO_OUT, bool HI16, bool FULL, bool FED>
__device__ __forceinline__ void scatter_group(
    uint32_t e, uint32_t G, uint32_t K, uint32_t grp, const uint32_t* __restrict__ in_lo,
    const uint32_t* __restrict__ in_hi, const uint32_t* __restrict__ in_id,
    uint32_t* __restrict__ out_lo, uint32_t* __restrict__ out_hi, uint32_t* __restrict__ out_id,
    const uint32_t* __restrict__ table, const uint32_t* __restrict__ seg_sum, const uint32_t* __restrict__ coarse,
    uint32_t shift, uint2* s_slot, typename std::conditional<HI16, uint16_t, uint32_t>::type* s_third,
    uint32_t* s_wcnt, uint32_t* s_pre, uint32_t* s_next, uint32_t* s_first) {
    constexpr int R = kSortKeysPerThread;
    // what travels beside the 8-byte slot {id, word}: nothing when the element is id + one 32-bit word (tile-word
    // passes; depth passes whose depth and tile words are both 16 bits wide), else the tile word (s_third)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool use_hi = shift >= 32u;
    const uint32_t sh = use_hi ? shift - 32u : (LO_IN == 2 ? shift - 16u : shift);   // bit offset inside the stored word
    const uint32_t tile_base = grp * kSortTile;
    const uint32_t base = tile_base + (uint32_t)wave * (R * 64) + lane;
    const uint32_t valid = FULL ? (uint32_t)kSortTile : e - tile_base;

    // ---- Scan + ScanAdd inputs (L2-resident outputs of the Count launch) and the group's elements: every load up
    //      front.  Keys of digit d ahead of this group = all keys of smaller digits (totals over the coarse segments)
    //      + digit d in earlier coarse segments + in earlier segments of this coarse segment + in earlier groups of
    //      this segment (RadixSortScan.comp:29-71 and RadixSortScanAdd.comp:34-66, evaluated where they are used).
    constexpr uint32_t kFinePerCoarse = kSegments / kCoarse;
    const uint32_t seg = grp / K;
    const uint32_t cseg = seg / kFinePerCoarse, jf = seg - cseg * kFinePerCoarse;
    const int sd = lane & 15, sq = lane >> 4;
    // thread t: digit t >> 4, coarse segments 4 (t & 15) .. + 3
    static_assert(kCoarse == 64 && kSortThreads == 256, "one 16-byte load per thread covers the coarse totals");
    uint4 cv = make_uint4(0u, 0u, 0u, 0u);
    uint32_t pre = 0u;
    constexpr int kFedBatch = 8;                 // 16-byte row loads a thread keeps in flight: 512 rows per batch
    uint4 fv[FED ? kFedBatch : 1];
    if constexpr (FED) {
        // rows of all groups, first batch: thread t takes digits 4 (t & 3) .. + 3 of rows t >> 2, + 64, ... (a wave
        // instruction = 16 whole rows); issued ahead of the keys (L2 hits: they are back long before the keys are)
        s_next[tid] = 0u; s_next[tid + kSortThreads] = 0u;
        if (tid < kBins) fed_zero_row(const_cast<uint32_t*>(coarse), grp, tid);
        const uint4* __restrict__ rows4 = reinterpret_cast<const uint4*>(table);
#pragma unroll
        for (int k = 0; k < kFedBatch; ++k) {
            const uint32_t r = ((uint32_t)tid >> 2) + 64u * (uint32_t)k;
            fv[k] = r < G ? rows4[r * 4u + ((uint32_t)tid & 3u)] : make_uint4(0u, 0u, 0u, 0u);
        }
    } else {
        cv = reinterpret_cast<const uint4*>(coarse)[tid];
        // the table already holds the group's prefix inside its segment (k_count); lanes of row 0 take it, every row
        // adds its share of the segment totals ahead inside the coarse segment
        pre = sq == 0 ? table[sd * G + grp] : 0u;
        for (uint32_t l = (uint32_t)sq; l < jf; l += 4u) pre += seg_sum[sd * kSegments + cseg * kFinePerCoarse + l];
    }

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

    // ---- per digit: total over all coarse segments, and over those before this group's (row of 16 lanes = one
    //      digit; DPP scan inside the row, lane 15 of the row holds the sums) -> LDS, read after barrier 1
    if constexpr (FED) {
        uint4 fa = make_uint4(0u, 0u, 0u, 0u), fb = make_uint4(0u, 0u, 0u, 0u);    // all groups / the groups ahead of this one
        auto take = [&](const uint4& v, uint32_t r) {     // a thread's rows come in ascending order: the sum ahead of the group is a
            fa.x += v.x; fa.y += v.y; fa.z += v.z; fa.w += v.w;   // snapshot of the running sum, taken as long as the row lies ahead
            if (r < grp) fb = fa;
        };
#pragma unroll
        for (int k = 0; k < kFedBatch; ++k) take(fv[k], ((uint32_t)tid >> 2) + 64u * (uint32_t)k);
        {   // lists of more than 512 groups: further batches (each one L2 round trip)
            const uint4* __restrict__ rows4 = reinterpret_cast<const uint4*>(table);
            for (uint32_t b = 64u * (uint32_t)kFedBatch; b < G; b += 64u * (uint32_t)kFedBatch) {   // kernel-uniform trip count
                uint4 w[kFedBatch];
#pragma unroll
                for (int k = 0; k < kFedBatch; ++k) {
                    const uint32_t r = b + ((uint32_t)tid >> 2) + 64u * (uint32_t)k;
                    w[k] = r < G ? rows4[r * 4u + ((uint32_t)tid & 3u)] : make_uint4(0u, 0u, 0u, 0u);
                }
#pragma unroll
                for (int k = 0; k < kFedBatch; ++k) take(w[k], b + ((uint32_t)tid >> 2) + 64u * (uint32_t)k);
            }
        }
        // lanes l, l + 4, ... of a wave hold partial sums of the same four digits: rotate-and-add inside the rows of 16
        // lanes (DPP row_ror 4, 8), then across the four rows; lanes 0..3 leave the wave's share in LDS
        // (s_pre[wave][0..15] totals, [16..31] ahead of this group), summed over the waves after barrier 1
        auto wsum = [](uint32_t v) {
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);   // row_ror:4
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);   // row_ror:8
            v += (uint32_t)__shfl_xor((int)v, 16, 64);
            v += (uint32_t)__shfl_xor((int)v, 32, 64);
            return v;
        };
        fa.x = wsum(fa.x); fa.y = wsum(fa.y); fa.z = wsum(fa.z); fa.w = wsum(fa.w);
        fb.x = wsum(fb.x); fb.y = wsum(fb.y); fb.z = wsum(fb.z); fb.w = wsum(fb.w);
        if (lane < 4) {
            *reinterpret_cast<uint4*>(&s_pre[wave * 2 * kBins + 4 * lane]) = fa;
            *reinterpret_cast<uint4*>(&s_pre[wave * 2 * kBins + kBins + 4 * lane]) = fb;
        }
    } else {
        const uint32_t c0 = 4u * (uint32_t)(tid & 15);
        const uint32_t all = cv.x + cv.y + cv.z + cv.w;
        const uint32_t before = (c0 < cseg ? cv.x : 0u) + (c0 + 1u < cseg ? cv.y : 0u) + (c0 + 2u < cseg ? cv.z : 0u) +
                                (c0 + 3u < cseg ? cv.w : 0u);
        const uint32_t all_s = row16_inclusive_scan(all), before_s = row16_inclusive_scan(before);
        if ((tid & 15) == 15) { s_pre[tid >> 4] = all_s; s_pre[kBins + (tid >> 4)] = before_s; }
        // fine segments and groups ahead inside this coarse segment (every wave alike, lanes sd + 16 q)
        pre += (uint32_t)__shfl_xor((int)pre, 16, 64);
        pre += (uint32_t)__shfl_xor((int)pre, 32, 64);
    }

    // ---- stable rank inside the wave.  Per round: 4 ballots give every lane the mask of lanes holding the same
    //      digit; lane d (d < 16) keeps the wave's running count of digit d.
    uint32_t rank[R];
    uint32_t cntreg = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool ok = FULL || base + r * 64 < e;
        const uint32_t kw = use_hi ? hi[r] : lo[r];
        const uint32_t dg = digit_of(kw, sh);
        // per digit bit b: S = all ones where the lane's bit is set (one signed bit-field extract), the ballot of the
        // bit, and XNOR(ballot, S) = the lanes that agree with this lane on bit b; AND over the four bits
        uint32_t m_lo = 0xFFFFFFFFu, m_hi = 0xFFFFFFFFu;
        if (!FULL) { const uint64_t v = __ballot(ok); m_lo = (uint32_t)v; m_hi = (uint32_t)(v >> 32); }
#pragma unroll
        for (int b = 0; b < kRadixBits; ++b) {
            const int32_t sbit = __builtin_amdgcn_sbfe((int32_t)kw, sh + (uint32_t)b, 1u);
            const uint64_t bal = __ballot(sbit != 0);
            m_lo &= ~((uint32_t)bal ^ (uint32_t)sbit);
            m_hi &= ~((uint32_t)(bal >> 32) ^ (uint32_t)sbit);
        }
        uint64_t mask = ((uint64_t)m_hi << 32) | m_lo;
        if (!FULL) mask = ok ? mask : 0ull;
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
    if (lane < kBins) s_wcnt[lane * kSortWaves + wave] = cntreg;
    __syncthreads();

    // ---- local digit starts, this wave's bases, global offset: lane d of every row of 16 (all four rows alike)
    uint32_t wbase, gofs;
    {
        static_assert(kSortWaves == 4, "one 16-byte LDS read per digit");
        const uint4 c = *reinterpret_cast<const uint4*>(&s_wcnt[sd * kSortWaves]);
        const uint32_t tot = c.x + c.y + c.z + c.w;
        const uint32_t dstart = row16_inclusive_scan(tot) - tot;        // first local position of digit sd
        wbase = dstart + (wave > 0 ? c.x : 0u) + (wave > 1 ? c.y : 0u) + (wave > 2 ? c.z : 0u);
        // Scan: keys of smaller digits anywhere = exclusive scan of the digit totals
        uint32_t dtot, ahead;
        if constexpr (FED) {
            static_assert(kSortWaves == 4, "four partial sums per digit");
            dtot = (s_pre[sd] + s_pre[2 * kBins + sd]) + (s_pre[4 * kBins + sd] + s_pre[6 * kBins + sd]);
            ahead = (s_pre[kBins + sd] + s_pre[3 * kBins + sd]) + (s_pre[5 * kBins + sd] + s_pre[7 * kBins + sd]);
        } else {
            dtot = s_pre[sd];
            ahead = s_pre[kBins + sd] + pre;
        }
        const uint32_t gpre = (row16_inclusive_scan(dtot) - dtot) + ahead;
        gofs = gpre - dstart;                                            // global = gofs(digit) + local position
        if constexpr (FED) { if (tid < kBins) s_first[tid] = gpre; }     // where the group's run of digit `tid` begins (read after barrier 2)
    }

    // ---- local sort into LDS.  The cross-lane reads run with every lane active (a lane past the end of a ragged
    //      group must still SERVE its wbase to the others: ds_bpermute returns 0 for a masked-off source lane)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t dg = digit_of(use_hi ? hi[r] : lo[r], sh);
        const uint32_t p = (uint32_t)__shfl((int)wbase, (int)dg, 64) + rank[r];
        if (FULL || base + r * 64 < e) {
            if constexpr (LO_IN == 0) s_slot[p] = make_uint2(id[r], hi[r]);
            else if constexpr (LO_IN == 2 && HI16) s_slot[p] = make_uint2(id[r], lo[r] | (hi[r] << 16));
            else { s_slot[p] = make_uint2(id[r], lo[r]); s_third[p] = (typename std::conditional<HI16, uint16_t, uint32_t>::type)hi[r]; }
        }
    }
    __syncthreads();

    // ---- run-wise coalesced stores (positions past `valid` read stale LDS: their digit only feeds a cross-lane
    //      read that must run unmasked, nothing of theirs is stored)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t p = (uint32_t)r * kSortThreads + tid;
        const uint2 sl = s_slot[p];
        uint32_t l, h;
        if constexpr (LO_IN == 0) { l = 0u; h = sl.y; }
        else if constexpr (LO_IN == 2 && HI16) { l = sl.y & 0xFFFFu; h = sl.y >> 16; }
        else { l = sl.y; h = s_third[p]; }
        const uint32_t d = digit_of(use_hi ? h : l, sh);
        const uint32_t o = (uint32_t)__shfl((int)gofs, (int)d, 64) + p;
        if (FULL || p < valid) {
            if constexpr (LO_OUT == 4) out_lo[o] = l;
            else if constexpr (LO_OUT == 2) reinterpret_cast<uint16_t*>(out_lo)[o] = (uint16_t)(LO_IN == 4 ? l >> 16 : l);
            if constexpr (HI16) reinterpret_cast<uint16_t*>(out_hi)[o] = (uint16_t)h;
            else out_hi[o] = h;
            out_id[o] = sl.x;
        }
        if constexpr (FED) {
            if (seg_sum) {   // not the last pass: the key's NEXT digit, counted under (run, destination group)
                // Lanes hold consecutive sorted positions, and in the depth passes the elements of one splat lie side by side
                // with one depth word: neighbouring lanes want the same counter (64 lanes on one LDS address serialise).  So
                // equal neighbours are added once: a lane whose counter differs from its left neighbour's (inside its row of
                // 16 lanes) heads a run and adds the run's length -- the distance to the next head in the ballot.
                const uint32_t ns = shift + (uint32_t)kRadixBits;
                const uint32_t nd = ns >= 32u ? digit_of(h, ns - 32u) : digit_of(l, LO_IN == 2 ? ns - 16u : ns);
                const uint32_t j = (o >> kSortTileLog2) - (s_first[d] >> kSortTileLog2);          // 0 or 1 for a stored element
                const bool stored = FULL || p < valid;
                const uint32_t ctr = stored ? (((d * 2u + j) << kRadixBits) | nd) : 0xFFFFFFFFu;
                const uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFEu, (int)ctr, 0x111, 0xf, 0xf, false);   // row_shr:1
                const bool head = ctr != left;
                const uint64_t heads = __ballot(head);
                const uint64_t after = lane == 63 ? 0ull : heads >> (lane + 1);
                const uint32_t len = after ? (uint32_t)__builtin_ctzll(after) + 1u : 64u - (uint32_t)lane;
                if (head && stored)
                    (void)__hip_atomic_fetch_add(&s_next[ctr], len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    if constexpr (FED) {
        if (seg_sum) {
            __syncthreads();
            uint32_t* __restrict__ next_rows = const_cast<uint32_t*>(seg_sum);
#pragma unroll
            for (int k = 0; k < 2; ++k) {        // 512 counters: 16 runs x 2 destination groups x 16 digits
                const uint32_t idx = (uint32_t)(k * kSortThreads + tid);
                const uint32_t c = s_next[idx];
                const uint32_t dgrp = (s_first[idx >> 5] >> kSortTileLog2) + ((idx >> 4) & 1u);
                if (c != 0u)
                    (void)__hip_atomic_fetch_add(&next_rows[dgrp * kBins + (idx & 15u)], c, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// (the fed kind keeps a batch of count rows in registers beside the keys: four workgroups per CU, 128 VGPRs -- its lists
// are short, a CU never holds more than four of its workgroups anyway)
template <int LO_IN, int LO_OUT, bool HI16, bool FED>
__global__ __launch_bounds__(kSortThreads, FED ? 4 : ((LO_IN == 4 || (LO_IN == 2 && !HI16)) ? GS_SCATTER_MINWAVES_KEY : GS_SCATTER_MINWAVES_SMALL))
void k_scatter(const SortParams* __restrict__ params, const uint32_t* __restrict__ in_lo,
               const uint32_t* __restrict__ in_hi, const uint32_t* __restrict__ in_id,
               uint32_t* __restrict__ out_lo, uint32_t* __restrict__ out_hi, uint32_t* __restrict__ out_id,
               const uint32_t* __restrict__ table, const uint32_t* __restrict__ seg_sum,
               const uint32_t* __restrict__ coarse, uint32_t shift) {
    constexpr bool kThird = LO_IN == 4 || (LO_IN == 2 && !HI16);
    __shared__ uint2 s_slot[kSortTile];
    __shared__ typename std::conditional<HI16, uint16_t, uint32_t>::type s_third[kThird ? kSortTile : 1];
    __shared__ __attribute__((aligned(16))) uint32_t s_wcnt[kBins * kSortWaves];
    __shared__ __attribute__((aligned(16))) uint32_t s_pre[FED ? 2 * kBins * kSortWaves : 2 * kBins];
    __shared__ uint32_t s_next[FED ? 2 * kBins * kBins : 1];      // [run][destination group 0 / 1][next digit]
    __shared__ uint32_t s_first[FED ? kBins : 1];
    const uint32_t e = params->num_elems, G = params->num_groups, K = params->groups_per_seg;
    // one group per workgroup as a rule: the grid is sized from an upper estimate of the element count (the list
    // capacity scaled to the context's share of the tiles) and walks on only if a frame exceeds it
    // Workgroups b, b + 8, ... share an XCD (observed placement, speed only): each of the eight takes a contiguous run of
    // the groups, so that the digit runs of neighbouring groups -- neighbours in the destination too -- meet in one L2.
    const uint32_t per_xcd = (G + 7u) / 8u;
    bool again = false;
    for (uint32_t vb = blockIdx.x; vb < 8u * per_xcd; vb += gridDim.x) {
        const uint32_t grp = (vb & 7u) * per_xcd + (vb >> 3);
        if (grp >= G) continue;
        if (again) __syncthreads();   // LDS is reused
        again = true;
        if (grp * kSortTile + kSortTile <= e)
            scatter_group<LO_IN, LO_OUT, HI16, true, FED>(e, G, K, grp, in_lo, in_hi, in_id, out_lo, out_hi, out_id, table,
                                                          seg_sum, coarse, shift, s_slot, s_third, s_wcnt, s_pre, s_next, s_first);
        else
            scatter_group<LO_IN, LO_OUT, HI16, false, FED>(e, G, K, grp, in_lo, in_hi, in_id, out_lo, out_hi, out_id, table,
                                                           seg_sum, coarse, shift, s_slot, s_third, s_wcnt, s_pre, s_next, s_first);
    }
}

int launch_radix_sort(const SortBuffers& sb, uint32_t capacity, uint32_t num_sort_bits,
                      hipStream_t stream, hipEvent_t* scatter_events, uint32_t first_bit,
                      bool drop_depth_payload, bool hi16, float share, int start, uint32_t coarse_pass,
                      const SortParams* params, uint32_t digit_bits, bool fed) {
    if (sb.digit_bits != digit_bits) return -1;   // table / seg_sum are sized per digit width (alloc_sort)
    if (digit_bits == 8u)
        return launch_radix_sort8(sb, capacity, num_sort_bits, stream, scatter_events, first_bit, drop_depth_payload, hi16,
                                  share, start, coarse_pass, params);
    if (fed && !sb.fed[0]) return -1;
    if (!params) params = sb.params;
    uint32_t max_groups = (capacity + kSortTile - 1) / kSortTile;
    // a context that owns a share of the tiles (tile-row band of a multi-GPU frame) launches Scatter over twice
    // that share of the capacity's groups; k_scatter walks on if a frame should hold more
    if (share < 0.5f) {
        const uint32_t g = (uint32_t)((float)max_groups * 2.0f * share) + 64u;
        max_groups = g < max_groups ? g : max_groups;
    }
    // a fed sort is chosen for short lists (gs_api.cpp: from the element count of the frame before): its grid need not
    // cover more groups than such a list has; k_scatter walks on if this frame holds more
    if (fed && max_groups > 2u * kFedMaxGroups) max_groups = 2u * kFedMaxGroups;
    // sb.coarse (the coarse digit totals of every pass; Count adds into them with atomics) must be zero on entry:
    // k_scan_blocks clears it in a frame, k_set_sort_params for the stand-alone sorter
    int src = start;
    uint32_t pass = 0;
    for (uint32_t shift = first_bit; shift < num_sort_bits; shift += kRadixBits, ++pass) { // RadixSort.cpp:309
        const int dst = src ^ 1;
        const bool tile_pass = shift >= 32u;
        const bool last = shift + kRadixBits >= num_sort_bits;
        const uint32_t* word = tile_pass ? sb.hi[src] : sb.lo[src];
        // 16-bit words: the tile ids of a band (hi16) and, in a frame, the upper half of the depth word once the
        // lower half is consumed (passes 4-7, see k_scatter)
        int cin, cout;
        scatter_depth_bytes(shift, first_bit, drop_depth_payload, &cin, &cout);
        const bool lo16 = !tile_pass && cin == 2;
        const bool word16 = (tile_pass && hi16) || lo16;
        uint32_t* coarse = sb.coarse + (size_t)(coarse_pass + pass) * kBins * kCoarse;   // zeroed above; this pass's Count adds into it
        // fed counts: one Count launch per SORT (the rows of pass 0); pass p reads set p % 3, adds into (p + 1) % 3
        // (nothing in the last pass) and clears (p + 2) % 3
        uint32_t* const rows_in = fed ? sb.fed[pass % 3u] : nullptr;
        uint32_t* const rows_next = fed && !last ? sb.fed[(pass + 1u) % 3u] : nullptr;
        uint32_t* const rows_zero = fed ? sb.fed[(pass + 2u) % 3u] : nullptr;
        if (!fed || pass == 0u) {
            uint32_t* const fr = fed ? rows_in : nullptr;
            uint32_t* const fz = fed ? sb.fed[1] : nullptr;
            if (word16)
                hipLaunchKernelGGL((k_count<true>), dim3(kSegments), dim3(kCountThreads), 0, stream, params,
                                   word, sb.table, sb.seg_sum, coarse, lo16 ? shift - 16u : shift & 31u, fr, fz);
            else
                hipLaunchKernelGGL((k_count<false>), dim3(kSegments), dim3(kCountThreads), 0, stream, params,
                                   word, sb.table, sb.seg_sum, coarse, shift & 31u, fr, fz);
        }
        if (scatter_events) (void)hipEventRecord(scatter_events[2 * pass], stream);
        // bytes of the depth word read / written by this pass (see k_scatter)
        int lo_in, lo_out;
        scatter_depth_bytes(shift, first_bit, drop_depth_payload, &lo_in, &lo_out);
        const uint32_t pgrid = max_groups;
#define GS_LAUNCH_SCATTER(LO_IN, LO_OUT, HI16, FED, T0, T1, T2)                                                        \
        hipLaunchKernelGGL((k_scatter<LO_IN, LO_OUT, HI16, FED>), dim3(pgrid), dim3(kSortThreads), 0, stream, params, \
                           sb.lo[src], sb.hi[src], sb.id[src], sb.lo[dst], sb.hi[dst], sb.id[dst],                  \
                           T0, T1, T2, shift)
#define GS_LAUNCH_SCATTER_F(LO_IN, LO_OUT, HI16) \
        do { if (fed) GS_LAUNCH_SCATTER(LO_IN, LO_OUT, HI16, true, rows_in, rows_next, rows_zero); \
             else GS_LAUNCH_SCATTER(LO_IN, LO_OUT, HI16, false, sb.table, sb.seg_sum, coarse); } while (0)
#define GS_LAUNCH_SCATTER_H(LO_IN, LO_OUT) \
        do { if (hi16) GS_LAUNCH_SCATTER_F(LO_IN, LO_OUT, true); else GS_LAUNCH_SCATTER_F(LO_IN, LO_OUT, false); } while (0)
        if (lo_in == 4 && lo_out == 4) GS_LAUNCH_SCATTER_H(4, 4);
        else if (lo_in == 4 && lo_out == 2) GS_LAUNCH_SCATTER_H(4, 2);
        else if (lo_in == 2 && lo_out == 2) GS_LAUNCH_SCATTER_H(2, 2);
        else if (lo_in == 2 && lo_out == 0) GS_LAUNCH_SCATTER_H(2, 0);
        else GS_LAUNCH_SCATTER_H(0, 0);
#undef GS_LAUNCH_SCATTER_H
#undef GS_LAUNCH_SCATTER_F
#undef GS_LAUNCH_SCATTER
        if (scatter_events) (void)hipEventRecord(scatter_events[2 * pass + 1], stream);
        src = dst;                                                            // RadixSort.cpp:638-641
    }
    return src;
}

// ---------------------------------------------------------------------------------------------
// helpers for the stand-alone sorter entry points (gs_sort_host / gs_sort_bench)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_set_sort_params(SortParams* params, uint32_t* coarse, uint32_t n) {
    for (uint32_t i = threadIdx.x; i < (uint32_t)(kMaxSortPasses * kBins * kCoarse); i += 1024u) coarse[i] = 0u;
    if (threadIdx.x != 0) return;
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

void launch_set_sort_params(SortParams* params, uint32_t* coarse, uint32_t n, hipStream_t stream) {
    hipLaunchKernelGGL(k_set_sort_params, dim3(1), dim3(1024), 0, stream, params, coarse, n);
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
