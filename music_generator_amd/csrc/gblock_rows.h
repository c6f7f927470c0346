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
