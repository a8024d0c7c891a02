// patch.h — host side of the patch operator (patch.hip): the per-batch tables.
#pragma once
#include "kernels.h"
#include "symbolic_gpu.h"

namespace remo {

size_t patch_arena_bytes(int64_t nt, int64_t n_max, int kmax);
// Enqueues the table kernels on s (no synchronisation).  flag_and_max: three device ints the caller reads after its next
// synchronisation - [0] != 0: a patch holds more distinct rows than the tables do (use another operator), [1]: the largest
// row count of a patch (PatchOpT::lds_rows), [2]: slots of the boundary slab in use (the caller sets PatchTables::nslot_cap to it).
void build_patch_tables(Arena &ar, hipStream_t s, const DeviceSymbolic &sy, const double *C, int kmax, PatchTables &out, int32_t *flag_and_max);

}  // namespace remo
