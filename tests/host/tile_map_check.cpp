// Host check of uzu_amd/csrc/gemm_tile_map.h: for every (m_tiles, n_tiles) the grid visits each output tile exactly once
// and all workgroups of a super-tile sit on one XCD (block id mod 8).
#include <stdio.h>
#include <vector>

#include "../../uzu_amd/csrc/gemm_tile_map.h"

int main() {
    using namespace uzu::k;
    long checked = 0, bad = 0;
    for (uint32_t m_tiles = 1; m_tiles <= 48; ++m_tiles)
        for (uint32_t n_tiles = 1; n_tiles <= 160; ++n_tiles) {
            const uint32_t grid = gemm_grid_x(m_tiles, n_tiles);
            std::vector<int> seen(m_tiles * n_tiles, 0);
            const TileMap tm = tile_map(m_tiles, n_tiles);
            bool ok = tm.S >= 1 && tm.S <= 64 && grid % 8 == 0;
            for (uint32_t b = 0; b < grid; ++b) {
                uint32_t m_t, n_t;
                if (!gemm_tile_of_block(b, m_tiles, n_tiles, &m_t, &n_t)) continue;
                seen[m_t * n_tiles + n_t]++;
                // the workgroups of a super-tile share the XCD: its index is (block >> 3) / S of that XCD's list
                const uint32_t q = (b & 7) + 8 * ((b >> 3) / tm.S);
                if (m_t / tm.TM != q % tm.m_blocks || n_t / tm.TN != q / tm.m_blocks) ok = false;
            }
            for (int v : seen) ok = ok && v == 1;
            ++checked;
            if (!ok) {
                if (bad++ < 5) printf("BAD m_tiles %u n_tiles %u\n", m_tiles, n_tiles);
            }
        }
    printf("tile_map: %ld shapes, %ld bad\n", checked, bad);
    return bad != 0;
}
