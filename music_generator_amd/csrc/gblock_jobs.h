// Job table of the generator block's weight gradients (add_generator_block, trainer/models.py:358-399):
// which activation rows meet which gradient rows of the tile-major side buffers (row map: gblock_rows.h),
// how many waves share a job and where their partial sums (slabs) sit in the workspace.  Shared by the
// kernels that read every operand from the side buffers (gblock_bwd.hip) and by the ones that recompute the
// operands of the widest layers from their inputs (gblock_lean.hip).
#pragma once
#include "gblock_rows.h"

struct GJob {
    int rowA, rowB;          // first activation / gradient row of sub-position term 0
    int nsub;                // terms contracted besides the positions (sub-positions sharing the weight)
    int strideA, strideB;    // row stride between terms
    int bias;                // 1: also sum the gradient operand over positions
    int nsplit;              // waves sharing the job (slabs)
    int pad;
    long ws_off;             // float offset of the job's slabs in the workspace
};
constexpr int NJOBS = 39;
struct GJobs {
    GJob j[NJOBS];
    int rows;                // rows per position tile of the side buffers the jobs read (GbRowsFull / GbRowsLean)
    int pad;
};

// job numbering (see build_jobs): 0-3 W1, 4-9 D2_k, 10-11 W2, 12 W3, 13 W4, 14-19 D1_k,
// 20-31 K2_k[b], 32-37 K1_k, 38 K0
struct JobShape {
    int ta, tb;
};
__host__ __device__ inline JobShape job_shape(int j) {
    if (j < 10) return {2, 4};
    if (j < 12) return {4, 2};
    if (j < 13) return {2, 1};
    if (j < 20) return {1, 1};
    if (j < 32) return {1, 2};
    if (j < 38) return {4, 1};
    return {1, 4};
}

// launches per accumulator shape: first job, number of jobs
struct ShapeRun {
    int ta, tb, job0, njobs;
    int wg_per_cu;      // resident workgroups per CU at the kernel's register use
};
static const ShapeRun kRuns[7] = {{2, 4, 0, 10, 1}, {4, 2, 10, 2, 1}, {2, 1, 12, 1, 3}, {1, 1, 13, 7, 5},
                                  {1, 2, 20, 12, 3}, {4, 1, 32, 6, 2}, {1, 4, 38, 1, 2}};

// workgroups per job: the jobs of a launch together fill the chip's resident slots exactly once (a
// partly filled second round would idle most CUs), at most 256 per job (1024 slabs)
static inline int grid_x(long ntiles, const ShapeRun& run) {
    long x = (ntiles + 3) / 4;
    long cap = (256L * run.wg_per_cu) / run.njobs;
    if (cap > 256) cap = 256;
    cap &= ~7L;         // a multiple of 8 workgroups per job: lets the kernel keep a slab's jobs on one XCD
    if (x > cap) x = cap;
    return (int)(x < 1 ? 1 : x);
}

// The recomputing kernels (gblock_lean.hip) run 6 branch jobs and 4 tail jobs, one workgroup per CU.  A
// branch job k fills the slabs of jobs 4+k (D2_k) and 20+2k, 21+2k (K2_k); tail job q (output channels
// 64q .. 64q+63 of Dense(256)) fills its tiles of the slabs of jobs 0-3 (W1) and 10-11 (W2).
constexpr int LEAN_BRANCH_JOBS = 6, LEAN_TAIL_JOBS = 4;
static inline int lean_grid_x(long ntiles, int njobs) {
    const ShapeRun run = {0, 0, 0, njobs, 1};
    return grid_x(ntiles, run);
}
static inline bool lean_job(int j) { return j < 12 || (j >= 20 && j < 32); }

// Fill the job table for `ntiles` position tiles; returns the workspace size in floats.  `lean`: side buffers with
// the lean row map; the jobs the recomputing kernels serve (their operands are not on such a tape) get those
// kernels' split counts and no rows.
template <class RM>
static inline void set_job_rows(GJobs& J) {
    auto set = [&](int j, int rowA, int rowB, int nsub, int sA, int sB, int bias) {
        J.j[j] = GJob{rowA, rowB, nsub, sA, sB, bias, 0, 0, 0};
    };
    for (int ab = 0; ab < 2; ++ab)
        for (int bb = 0; bb < 2; ++bb) set(ab * 2 + bb, RM::S + 64 * ab, RM::T1 + 128 * bb, 4, 128, 256, ab == 0);
    for (int k = 0; k < 6; ++k) set(4 + k, RM::BR + k * RM::BR_ROWS + RM::O3, RM::S, 4, 64, 128, k == 0);
    for (int ab = 0; ab < 2; ++ab) set(10 + ab, RM::T1 + 128 * ab, RM::T2, 4, 256, 64, ab == 0);
    set(12, RM::T2, RM::T3, 4, 64, 8, 1);
    set(13, RM::T3, RM::X, 4, 8, 2, 1);
    for (int k = 0; k < 6; ++k) {
        const int br = RM::BR + k * RM::BR_ROWS;
        set(14 + k, br + RM::O1, br + RM::O2, 2, 16, 32, 1);
        for (int b = 0; b < 2; ++b) set(20 + k * 2 + b, br + RM::O2, br + RM::O3 + 64 * b, 2, 32, 128, 1);
        set(32 + k, RM::F, br + RM::O1, 1, 0, 0, 1);
    }
    set(38, RM::X, RM::F, 1, 0, 0, 1);
    J.rows = RM::ROWS;
    J.pad = 0;
}

static inline long build_jobs(GJobs& J, long ntiles, bool lean = false) {
    if (lean) set_job_rows<GbRowsLean>(J);
    else set_job_rows<GbRowsFull>(J);
    long off = 0;
    for (const ShapeRun& run : kRuns) {
        for (int j = run.job0; j < run.job0 + run.njobs; ++j) {
            int nsplit = grid_x(ntiles, run) * 4;
            if (lean && lean_job(j))
                nsplit = lean_grid_x(ntiles, (j < 4 || j == 10 || j == 11) ? LEAN_TAIL_JOBS : LEAN_BRANCH_JOBS) * 4;
            J.j[j].nsplit = nsplit;
            J.j[j].ws_off = off;
            off += (long)nsplit * (run.ta * run.tb * 1024L + run.tb * 64L);
        }
    }
    return off;
}
