// Tile-order helpers shared by the tile-GEMM translation units (gemm.hip, gemm_w128.hip).
#pragma once
#include "common.h"

#define BK 64
#define BM2 256
#define BN2 256

namespace CW_NS {

// Grouped tile order (logical id -> (m-tile, n-tile)): ids walk down GM m-tiles of one n-tile before moving to
// the next n-tile, so the ~32 blocks resident on an XCD share 8 A panels x 4 W panels (< 4 MB L2) instead of
// streaming the whole weight matrix once per m-tile row (measured 24x over-fetch on fc1 with row-major order).
__device__ inline void grouped_tile(int tile, int tiles_m, int tiles_n, int& mt, int& nt, const int GM = 8) {
    const int width = GM * tiles_n;
    const int group = tile / width;
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int in_g = tile - group * width;
    mt = first_m + in_g % gsz;
    nt = in_g / gsz;
}

// XCD-aware, bijective block remap (8 XCDs, block b is dispatched to XCD b % 8): consecutive logical
// tiles, which share an A row-panel, land on the same XCD's L2.
__device__ inline int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    int xcd = bid % nx, slot = bid / nx;
    int q = nwg / nx, r = nwg % nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

}  // namespace CW_NS
