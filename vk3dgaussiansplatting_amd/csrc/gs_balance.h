// gs_balance.h -- how a sharded frame's tile rows are cut into contiguous bands (gs_dist.cpp: gs_dist_shard_rows, gs_dist_rebalance; the
// export gs_balance_rows).  Pure host arithmetic, no HIP: also built by the sanitizer test (tests/host/sanitize_host.cpp).
#pragma once

#include <cstdint>
#include <vector>

namespace gs {

// world + 1 edges of contiguous bands over weights.size() rows whose weights are as equal as whole rows allow
std::vector<uint32_t> balanced_edges(const std::vector<double>& weights, uint32_t world);
// ceil(Ty / R) rows each (dist.tile_row_partition)
std::vector<uint32_t> equal_row_edges(uint32_t tiles_y, uint32_t world);

}  // namespace gs
