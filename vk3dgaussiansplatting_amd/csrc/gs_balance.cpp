// gs_balance.cpp -- the band-cutting rule of a sharded frame (see gs_balance.h).  No HIP in this file.
#include "gs_balance.h"

#include "../../include/gsplat.h"

#include <algorithm>
#include <cmath>

namespace gs {

// Contiguous bands whose weights are as equal as whole rows allow -- vk3dgaussiansplatting_amd/dist.py:
// balanced_row_partition, statement for statement (the two are compared in the tests): edge r sits at the row boundary
// whose weight prefix is nearest to r / R of the total; every rank keeps at least one row while there are rows to give.
std::vector<uint32_t> balanced_edges(const std::vector<double>& weights_in, uint32_t world) {
    const uint32_t ty = (uint32_t)weights_in.size();
    std::vector<uint32_t> edges(world + 1u, 0u);
    if (ty == 0u) return edges;
    std::vector<double> w = weights_in;
    double total = 0.0;
    bool usable = true;       // finite and non-negative: a negative weight would make the prefix non-monotone (binary search below)
    for (double x : w) { total += x; usable = usable && std::isfinite(x) && x >= 0.0; }
    if (!usable || total <= 0.0) w.assign(ty, 1.0);
    std::vector<double> prefix(ty + 1u, 0.0);
    for (uint32_t k = 0; k < ty; ++k) prefix[k + 1u] = prefix[k] + w[k];
    for (uint32_t r = 1; r < world; ++r) {
        const double target = prefix[ty] * (double)r / (double)world;
        uint32_t k = (uint32_t)(std::lower_bound(prefix.begin(), prefix.end(), target) - prefix.begin());   // first prefix >= target
        if (k > 0u && target - prefix[k - 1u] <= prefix[std::min(k, ty)] - target) --k;
        const uint32_t lo = std::min(edges[r - 1u] + 1u, ty);
        const uint32_t left = world - r;                                  // ranks still to come
        const uint32_t hi = std::max(lo, ty > left ? ty - left : 0u);
        edges[r] = std::min(std::max(k, lo), hi);
    }
    edges[world] = ty;
    return edges;
}

std::vector<uint32_t> equal_row_edges(uint32_t ty, uint32_t world) {       // dist.tile_row_partition: ceil(Ty / R) rows each
    const uint32_t per = (ty + world - 1u) / world;
    std::vector<uint32_t> e(world + 1u);
    for (uint32_t r = 0; r <= world; ++r) e[r] = std::min(r * per, ty);
    return e;
}

}  // namespace gs

extern "C" int gs_balance_rows(const double* row_weights, uint32_t tiles_y, uint32_t world, uint32_t* edges_out) {
    if (!edges_out || world == 0u || (tiles_y && !row_weights)) return GS_ERR_INVALID;
    const std::vector<uint32_t> e = gs::balanced_edges(std::vector<double>(row_weights, row_weights + tiles_y), world);
    for (uint32_t k = 0; k <= world; ++k) edges_out[k] = e[k];
    return GS_OK;
}
