// rp_tile.hpp -- STUB (tile kernels land in the next commit)
#pragma once
#include "rp_device.hpp"
namespace rp {
constexpr int kTileBlock = 256;
struct TilePlan { long long n_tiles; long long total_nt; long long n_orfs; };
struct TileWorkspace { long long *tile_first; void *partials; };
inline TilePlan make_tile_plan(long long n_orfs, long long total_nt) { return TilePlan{1, total_nt, n_orfs}; }
inline size_t workspace_bytes(const TilePlan &) { return 256; }
inline TileWorkspace carve_workspace(void *p, const TilePlan &) { return TileWorkspace{(long long *)p, nullptr}; }
__global__ void k_tile_index(const int64_t *, long long, TilePlan, TileWorkspace) {}
__global__ void k_tile_score(const int32_t *, const int64_t *, long long, TilePlan, TileWorkspace, OrfOutputs, FilterParams) {}
__global__ void k_tile_finalize(const int32_t *, const int64_t *, long long, TilePlan, TileWorkspace, OrfOutputs, FilterParams) {}
}
