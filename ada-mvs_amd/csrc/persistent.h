// Shared by the persistent kernels (slice_red.hip, slice_red_bf16x3.hip): tile bookkeeping and the
// resident-capacity query.
#pragma once
#include "common.h"

namespace adamvs {

// Tile bookkeeping of the persistent kernels: tile t -> (tile_x, tile_y, b) with the two divisions done as
// multiply-high by constants prepared on the host (t is workgroup-uniform, so this stays on the scalar unit).
struct TileGrid {
  int tiles_x, tiles_y, ntiles;
  unsigned mx, my;       // ceil(2^32 / tiles_x), ceil(2^32 / tiles_y); unused when the divisor is 1
};
__device__ __forceinline__ void tile_coords(const TileGrid& g, int t, int& b, int& tx, int& ty) {
  unsigned r = g.tiles_x == 1 ? (unsigned)t : __umulhi((unsigned)t, g.mx);
  tx = t - (int)r * g.tiles_x;
  unsigned bb = g.tiles_y == 1 ? r : __umulhi(r, g.my);
  ty = (int)r - (int)bb * g.tiles_y;
  b = (int)bb;
}
static int make_tile_grid(TileGrid& g, int tiles_x, int tiles_y, int B) {
  long n = (long)tiles_x * tiles_y * B;
  // exactness of the multiply-high quotient needs t * divisor < 2^32
  if (n <= 0 || n * (tiles_x > tiles_y ? tiles_x : tiles_y) >= (1L << 32)) return set_error(-1, "too many tiles (%ld)", n);
  g.tiles_x = tiles_x; g.tiles_y = tiles_y; g.ntiles = (int)n;
  g.mx = (unsigned)(((1ull << 32) + tiles_x - 1) / tiles_x);
  g.my = (unsigned)(((1ull << 32) + tiles_y - 1) / tiles_y);
  return 0;
}

// workgroups of `kernel` that stay resident per CU (occupancy query, cached per instantiation by the caller)
template <typename K>
static int resident_blocks(K kernel, int threads, size_t lds) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, lds) != hipSuccess || n < 1) n = 1;
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  return n * cus;
}

}  // namespace adamvs
