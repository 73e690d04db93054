// fast_levels.h -- the levels of a pyramid buffer as the multi-level FAST kernels see them (fast.hip; built by orb.hip): where each level sits in the buffer
// (x, y, w, h), the first 4-row tile of each level in the grid of the score passes (tile0), and its first row in the row list of the collect passes (row0);
// entry [n] closes both lists.
#pragma once
namespace mi355 {
constexpr int FAST_MAX_LEVELS = 32;
struct FastLevels { int n; int x[FAST_MAX_LEVELS], y[FAST_MAX_LEVELS], w[FAST_MAX_LEVELS], h[FAST_MAX_LEVELS]; int tile0[FAST_MAX_LEVELS + 1], row0[FAST_MAX_LEVELS + 1]; };
}
