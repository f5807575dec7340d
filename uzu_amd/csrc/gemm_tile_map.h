// gemm_tile_map.h -- workgroup id -> output tile of the large-tile prefill GEMM (k_gemm128.hip).
// Plain C++ so that tests/test_host_math.py can compile it for the host and check that every tile is visited exactly once.
#pragma once
#include <stdint.h>
#ifndef __HIPCC__
#define UZU_TILE_HD
#else
#define UZU_TILE_HD __host__ __device__
#endif

namespace uzu {
namespace k {

struct TileMap {
    uint32_t TM, TN, S, m_blocks, Q; // super-tile of TM x TN tiles (S = TM TN <= 64), Q super-tiles in all
};
UZU_TILE_HD inline TileMap tile_map(uint32_t m_tiles, uint32_t n_tiles) {
    TileMap t;
    uint32_t tm0 = 1;
    while (tm0 * 2 <= m_tiles && tm0 < 8) tm0 *= 2;
    uint32_t tn = n_tiles / 8 < 64 / tm0 ? n_tiles / 8 : 64 / tm0; // keep at least one super-column per XCD
    t.TN = tn ? tn : 1;
    t.TM = tm0;
    t.S = t.TM * t.TN;
    t.m_blocks = (m_tiles + t.TM - 1) / t.TM;
    t.Q = t.m_blocks * ((n_tiles + t.TN - 1) / t.TN);
    return t;
}

// workgroup `block` of a grid of gemm_grid_x(...) -> tile (m_t, n_t); false = padding workgroup (exits)
UZU_TILE_HD inline bool gemm_tile_of_block(uint32_t block, uint32_t m_tiles, uint32_t n_tiles, uint32_t* m_t, uint32_t* n_t) {
    const TileMap tm = tile_map(m_tiles, n_tiles);
    const uint32_t xcd = block & 7, slot = block >> 3;
    const uint32_t q = xcd + 8 * (slot / tm.S), in = slot % tm.S;
    *m_t = (q % tm.m_blocks) * tm.TM + in % tm.TM;
    *n_t = (q / tm.m_blocks) * tm.TN + in / tm.TM;
    return *m_t < m_tiles && *n_t < n_tiles;
}
UZU_TILE_HD inline uint32_t gemm_grid_x(uint32_t m_tiles, uint32_t n_tiles) {
    const TileMap tm = tile_map(m_tiles, n_tiles);
    return 8 * tm.S * ((tm.Q + 7) / 8);
}

} // namespace k
} // namespace uzu
