// Compact [A B] of the KUKA arm's Euler step (thread-lane setup kernel -> matrix-core backward pass, float handles with >= 512 problems).
//
// [A B] = I + dt [0 I 0; dqdd/d(q, qd, u)]   (utils/integrators.cuh:38-53): rows 0..6 of every column are the constants {1, dt, 0} -- half of the
// 1176 bytes per knot the reference's layout moves from the derivative kernel to the backward pass and back every iteration.  On this path only
// the 7 DYNAMIC rows (state rows 7..13) of the 21 columns are kept: 147 floats = 588 bytes per knot, and the consumer rebuilds the constant
// rows in registers.
//
// Layout: knots are addressed by their global index G = problem * N + knot; 64 consecutive knots form a chunk (= the knots of one wave of the setup
// kernel), and inside a chunk the columns are grouped into the three PIECES in which that kernel completes them (arm_tl_gradient's marks):
//     piece 0: columns 0..3, 7..10    piece 1: columns 4..6, 11..13    piece 2: columns 14..20
//     chunk = [piece 0: 64 knots x 8 columns x 7 rows][piece 1: 64 x 6 x 7][piece 2: 64 x 7 x 7]
// so every flush of the setup kernel writes ONE contiguous, 16-byte aligned run (14336 / 10752 / 12544 bytes) with 16 bytes per lane, and a knot's
// share of a piece is one contiguous run of 224 / 168 / 196 bytes for the backward pass.
// The reference-layout array "AB" stays the API view: pddp_get_array("AB") expands, pddp_set_array("AB") compacts (k_abc_expand / k_abc_compact).
#pragma once

#include "pddp_common.hpp"

namespace pddp {

constexpr int kAbcKnot = 147;                       // floats per knot
constexpr int kAbcChunk = 64 * kAbcKnot;            // floats per chunk of 64 knots
PDDP_HD constexpr int abc_piece(int col) { return col >= 14 ? 2 : ((col % 7) < 4 ? 0 : 1); }
PDDP_HD constexpr int abc_piece_cols(int piece) { return piece == 0 ? 8 : piece == 1 ? 6 : 7; }
PDDP_HD constexpr int abc_piece_off(int piece) { return piece == 0 ? 0 : piece == 1 ? 64 * 56 : 64 * 98; }
PDDP_HD constexpr int abc_col_in_piece(int col) { return col < 4 ? col : col < 7 ? col - 4 : col < 11 ? col - 3 : col < 14 ? col - 8 : col - 14; }
PDDP_HD constexpr int abc_piece_col(int piece, int ci) { return piece == 0 ? (ci < 4 ? ci : ci + 3) : piece == 1 ? (ci < 3 ? ci + 4 : ci + 8) : ci + 14; }
// float index of dynamic row r (state row 7 + r) of column `col` of global knot G
PDDP_HD size_t abc_index(size_t G, int col, int r) {
    const int p = abc_piece(col);
    return (G >> 6) * kAbcChunk + abc_piece_off(p) + ((G & 63) * abc_piece_cols(p) + abc_col_in_piece(col)) * 7 + r;
}
// floats to allocate for `knots` knots (whole chunks + slack for the backward pass's clamped 16-byte over-reads)
PDDP_HD constexpr size_t abc_floats(size_t knots) { return ((knots + 63) / 64) * kAbcChunk + 16; }
// the constant rows: [A B](row, col) for row < 7
PDDP_HD constexpr float abc_const(int row, int col, float dt) { return col == row ? 1.f : (col == row + 7 ? dt : 0.f); }

}  // namespace pddp
