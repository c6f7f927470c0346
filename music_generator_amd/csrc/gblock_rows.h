// Row map of the two side buffers of the fused generator grow block (gblock.hip, gblock_bwd.hip).
//
// Both buffers are tile-major: [position tile of 32 low-resolution positions][row][32 positions],
// float32, GB_ROWS rows per tile.  A row is one channel of one saved tensor at one of the four
// output sub-positions (a, b) of the 2x2 up-sampling; ab = 2a + b.  The activation buffer (written
// by the saving forward) and the gradient buffer (written by the data-gradient kernel) use the
// SAME map, the gradient of a tensor sitting where the tensor itself sits in the other buffer;
// rows GB_X hold the block input (activations) / the output gradient dY (gradients).
#pragma once

constexpr int GB_X = 0;          // activations: input x, 2 rows;  gradients: dY, [ab][2] = 8 rows
constexpr int GB_F = 8;          // featured conv output, 128 rows
constexpr int GB_BR = 136;       // 6 branches x GB_BR_ROWS
constexpr int GB_BR_ROWS = 352;
constexpr int GB_O1 = 0;         //   ConvT(16,(2,1)) output, rows (a, co) = 16a + co       (32)
constexpr int GB_O2 = 32;        //   Dense(32) output, [a][32]                             (64)
constexpr int GB_O3 = 96;        //   ConvT(64,(1,2)) output, [ab][64]                      (256)
constexpr int GB_S = GB_BR + 6 * GB_BR_ROWS;   // 2248: Add output, [ab][128]
constexpr int GB_T1 = GB_S + 512;              // 2760: Dense(256) output, [ab][256]
constexpr int GB_T2 = GB_T1 + 1024;            // 3784: Dense(64) output, [ab][64]
constexpr int GB_T3 = GB_T2 + 256;             // 4040: Dense(8) output, [ab][8]
constexpr int GB_END = GB_T3 + 32;             // 4072
constexpr int GB_ROWS = GB_END + 32;           // 4104: padding so that a 32-row read of a short tensor stays inside

// The bf16 kernels take the map as a type.  GbRowsLean is the map of a tape WITHOUT the ConvT(64,(1,2)) and
// Dense(256) outputs (and without the gradients that arrive at them): 1544 rows per tile instead of 4104; the
// weight-gradient kernels of gblock_lean.hip recompute those tensors from the rows next to them.
struct GbRowsFull {
    static constexpr int X = GB_X, F = GB_F, BR = GB_BR, BR_ROWS = GB_BR_ROWS, O1 = GB_O1, O2 = GB_O2, O3 = GB_O3, S = GB_S,
                         T1 = GB_T1, T2 = GB_T2, T3 = GB_T3, ROWS = GB_ROWS;
    static constexpr bool lean = false;
};
struct GbRowsLean {
    static constexpr int X = 0, F = 8, BR = 136, BR_ROWS = 96, O1 = 0, O2 = 32, O3 = 0 /* not kept */,
                         S = BR + 6 * BR_ROWS /* 712 */, T1 = 0 /* not kept */, T2 = S + 512 /* 1224 */,
                         T3 = T2 + 256 /* 1480 */, ROWS = T3 + 32 + 32 /* 1544 */;
    static constexpr bool lean = true;
};
// inside a kernel templated on the map: the names of the float32 map, bound to the kernel's own map
#define GB_USE_ROWS(RM)                                                                                              \
    constexpr int GB_X = RM::X, GB_F = RM::F, GB_BR = RM::BR, GB_BR_ROWS = RM::BR_ROWS, GB_O1 = RM::O1, GB_O2 = RM::O2, \
                  GB_O3 = RM::O3, GB_S = RM::S, GB_T1 = RM::T1, GB_T2 = RM::T2, GB_T3 = RM::T3, GB_ROWS = RM::ROWS;   \
    (void)GB_X; (void)GB_F; (void)GB_BR; (void)GB_BR_ROWS; (void)GB_O1; (void)GB_O2; (void)GB_O3; (void)GB_S;          \
    (void)GB_T1; (void)GB_T2; (void)GB_T3; (void)GB_ROWS
