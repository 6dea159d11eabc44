// relgraph_dense.h -- what rspmm_kernels.hip's run_plan needs from relgraph_dense.hip (internal to libultra_rspmm.so; the library
// is built with -fvisibility=hidden, nothing here is exported).
#ifndef ULTRA_RELGRAPH_DENSE_H
#define ULTRA_RELGRAPH_DENSE_H

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "ultra_rspmm.h"

namespace ultra_detail {

// one sum-aggregation call over a plan: kind 0 forward, 1 d_input, 2 d_relation (rspmm_kernels.hip's enum Kind)
struct DenseCall {
    const ultra_segments *seg;
    int kind, sum_op, mul_op;
    const float *relation;      // [n_rel, F]
    const float *input;         // [n_src, F]
    const float *grad;          // backward: output_grad [n_dst, F]
    const float *add_rows;      // forward: fused boundary rows; d_input: the gradient to accumulate into (may alias out)
    const int32_t *bnode;       // forward: sparse boundary
    const float *bvec;
    int bdim;
    float *out;
    void *workspace;
    size_t workspace_bytes;
    int64_t gather_rows, gather2_rows, n_rel, F;
};

// compute units the persistent grids of this process are sized for on the current device (the device's count minus
// ultra_rspmm_reserve_cus); implemented in rspmm_kernels.hip, used by the other translation units' launches
int persistent_cus(int *n_cu);
// knob bit 4 of ultra_rspmm_force_general_path: wide lane groups (32 / 64 lanes per row) on inputs small enough to be cache-resident
bool wide_groups_forced();

bool dense_applies(const DenseCall &call);
int dense_launch(const DenseCall &call, hipStream_t stream);

}  // namespace ultra_detail

#endif
