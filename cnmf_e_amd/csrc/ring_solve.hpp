// B2b: the per-pixel ridge solve of fit_ring_model.m:92-108 -- the kernel: set-up of one pixel's window, the gather of its normal equations
// from the block-pair covariance table into MFMA accumulator tiles, rs_solve_core (ring_solve_core.hpp), the store of the p weights.
#pragma once

namespace cnmfe {

// What a load reads for an entry with a missing neighbour (identity rows: no select and no live mask behind the load) comes from a device buffer
// `fillg` = {0, 1} in GLOBAL memory.  Round 3 kept the two values in a `__device__ const` array: the gather then loads through a pointer that is either
// into the table (global) or into the constant address space -- a FLAT pointer, and a flat load counts on lgkmcnt as well as vmcnt, so every
// `s_waitcnt lgkmcnt(0)` behind the two LDS reads that form the NEXT entry's address also waited for the previous entry's table load (84 flat_load_dwordx2
// in k_ring_solve5<6>, each behind a full wait).  With both pointers global the loads are global_load_dwordx2 on vmcnt alone.  Measured in round 4 at the
// headline size together with the per-fit window-code table (k_win_codes, bg.hip): 8.02 -> 7.64 ms (profiles/r04/queued_patches_ab.txt) -- the gather was
// not serialised as badly as the ISA suggested.
#define RS_KNAME k_ring_solve5
#define RS_EXTRA , const double *__restrict__ fillg
#define RS_FILLP fillg
#include "ring_solve_kernel.inc"
#undef RS_KNAME
#undef RS_EXTRA
#undef RS_FILLP

}  // namespace cnmfe
