// bcast_plan.h -- what broadcast.hip (built-in Ops, ahead of time) and jit.hip (user-defined Ops, hipRTC) share on the
// host: the normalised problem, the kernel choice, and the parameter blocks of bcast_kernels.hip.h.
#pragma once

#include "internal.h"
#include "ops.hip.h"

namespace smhip {
namespace bk {

using namespace dev;

#include "bcast_kernels.hip.h"

constexpr int kLdsVectorsInFlight = 4;  // LDS kernel: 75-78 % of peak; 1: 55-65 %, 2: 73-80 %, 8: 71-74 % (tools/bcast_matrix.py)

// A broadcast problem after normalise(): size-1 dims dropped, jointly dense / jointly broadcast neighbours merged.
struct Plan {
    int ndim;
    int64_t shape[SMHIP_MAX_NDIM], sa[SMHIP_MAX_NDIM], sb[SMHIP_MAX_NDIM];
    size_t n;
};
Plan normalise(const int64_t *shape, const int64_t *sa, const int64_t *sb, int ndim);

// One kernel launch: the body, its compile-time variant, the grid and the parameter block.
struct Launch {
    enum Kind { kRow, kLds, kTile, kGather, kStrided } kind;
    int ia, ib, tx, rows;   // row: INNER_A / INNER_B, lanes per row, rows per lane
    bool ca, cb;            // row: CONST_A / CONST_B
    bool swapped;           // lds: the streamed operand is the Op's right one
    bool vec;               // tile: 16-byte form
    int qb;                 // tile: bytes of one patch row (kTileQBytes, or kTileQBytesWide with the row-major walk, or kTileQBytesShort)
    int ma, mb;             // tile: LDS-mode operands (compile-time in the 16-byte form)
    int w;                  // gather: outputs per lane
                            // strided rows: ia / ib are the inner strides (0..4)
    unsigned grid;
    size_t lds_bytes;
    union Params {
        RowParams row;
        LdsParams lds;
        TileParams tile;
        GatherParams gather;
        StridedParams strided;
        Params() {}
    } p;
};
// `heavy`: the Op is arithmetic-bound (pow).  esz: element size in bytes (4 or 8).
int plan_launch(const Plan &pl, int esz, bool heavy, Launch *L);

}  // namespace bk

// jit.hip: one broadcast launch of a user-defined Op (the variant is compiled by hipRTC on first use)
int jit_launch(int op, int dtype, const bk::Launch &L, const void *a, const void *b, void *out, hipStream_t s);

}  // namespace smhip
