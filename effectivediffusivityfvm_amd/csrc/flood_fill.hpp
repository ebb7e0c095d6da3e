// flood_fill.hpp -- connectivity of the pore space (host side, linear time).
//
// Same result as the reference's FloodFill (Deff2DGPU/Deff2D.cuh:557-713): the cells that
// are not solid (Grid != 1) and cannot be reached from the left column through
// 4-connected non-solid cells -- with wrap-around between the first and last ROW
// (cuh:641-664) but not between columns -- are marked Grid = 2 ("non-participating",
// they get identity rows in DiscretizeMatrix2D_ImpSolid, cuh:750-752).  PathFlag
// reports whether the fill ever visits the last column (cuh:619-621).
//
// The reference keeps its frontier in a std::set<pair<int,int>> (O(n log n), ordered
// pops); the reached set does not depend on the visiting order, so this version uses a
// flat index queue and a byte map: O(n), no allocation per cell.
//
// One reference quirk is reproduced because it changes results: the right-column
// seeding test is written `Domain[indexR == -1]` (cuh:601) and therefore reads
// Domain[0].  By the time it is evaluated Domain[0] is 0 exactly when the top-left
// cell is not solid, so every cell of the last column (solid or not) is put on the
// frontier iff the TOP-LEFT cell is solid -- which also raises PathFlag.
#pragma once
#include <cstdint>
#include <vector>

namespace deff {

inline int flood_fill(unsigned int *grid, int nx, int ny)
{
    const int64_t n = (int64_t)nx * ny;
    enum : uint8_t { OPEN = 0, WALL = 1, SEEN = 2 };
    std::vector<uint8_t> state((size_t)n);
    for (int64_t p = 0; p < n; ++p) state[p] = (grid[p] == 1) ? WALL : OPEN;
    std::vector<int64_t> queue;
    queue.reserve((size_t)n / 4 + 2 * (size_t)ny);
    // a solid cell seeded through the quirk is visited too (it spreads the fill); remember it
    // separately so that `state` keeps telling solids from pores
    const bool top_left_solid = state[0] == WALL;
    for (int r = 0; r < ny; ++r) {
        const int64_t left = (int64_t)r * nx, right = left + nx - 1;
        if (state[left] == OPEN) { state[left] = SEEN; queue.push_back(left); }
        if (top_left_solid) {
            if (state[right] == OPEN) state[right] = SEEN;
            queue.push_back(right);                      // even when solid, cuh:601-603
        }
    }
    bool path = false;
    auto visit = [&](int64_t q) {
        if (state[q] == OPEN) { state[q] = SEEN; queue.push_back(q); }
    };
    for (size_t head = 0; head < queue.size(); ++head) {
        const int64_t p = queue[head];
        const int r = (int)(p / nx), c = (int)(p - (int64_t)r * nx);
        if (c == nx - 1) path = true;
        visit((int64_t)(r == 0 ? ny - 1 : r - 1) * nx + c);      // north, periodic
        visit((int64_t)(r == ny - 1 ? 0 : r + 1) * nx + c);      // south, periodic
        if (c != 0) visit(p - 1);
        if (c != nx - 1) visit(p + 1);
    }
    for (int64_t p = 0; p < n; ++p)
        if (state[p] == OPEN) grid[p] = 2;
    return path ? 1 : 0;
}

}  // namespace deff
