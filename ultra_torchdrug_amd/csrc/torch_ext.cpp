// torch_ext.cpp -- the PyTorch-ROCm C++ extension of the operator boundary: TORCH_LIBRARY(ultra_mi, ...).
//
// The reference binds its rspmm through a JIT-built PyTorch C++ extension (torchdrug.utils.extension.load over
// rspmm.{h,cpp,cu}; /root/reference/README.md:43-45, call sites /root/reference/ultra/layer.py:134-167,336-369).  This
// file is that layer for the MI355X library: dispatcher-registered operators (CUDA dispatch key = HIP under ROCm,
// Autograd key for the raw-CSR operator) that validate tensors, allocate outputs through the caching allocator, take the
// CURRENT stream and call the same C ABI (include/ultra_rspmm.h) the ctypes binding calls.  Host code only: the
// kernels live in libultra_rspmm.so, which this library links.  Built with hipcc against the torch headers
// (csrc/Makefile, target torch); nothing here is generated or translated.
//
// CPU dispatch key (BASELINE config 1 runs with `--gpus null`, /root/reference/README.md:79,90): build_relcsr, rspmm_fwd and
// rspmm_bwd also have host kernels, written here (at::parallel_for; per output element ONE sequential accumulation over
// the row's sorted edges, `y = w * (rel (*|+) x)` rounded before `acc (+|min|max) y` -- this file is compiled with
// -ffp-contract=off).  They are product code: nothing under oracle/ is included or linked; the tests hold them
// bit-equal to the oracle's sequential order.
//
// Operators:
//   ultra_mi::build_relcsr(edge_list, edge_weight?, num_node, num_relation) -> Tensor[]
//       coalesced CSR over destination nodes of torchdrug's (node_in, node_out, relation) edge list:
//       [row_ptr int32 (N + 1), src int32 (E), rel int32 (E), w fp32 (E), edge_of_input int64 (E_in)]
//   ultra_mi::rspmm_fwd(row_ptr, src, rel, w?, relation, input, sum_op, mul_op) -> Tensor          (differentiable)
//   ultra_mi::rspmm_bwd(row_ptr, src, rel, w?, relation, input, output, output_grad, sum_op, mul_op)
//       -> (d_relation, d_input)
//   plan-based forms used by ultra_torchdrug_amd.functional (the plan = the bytes of one `ultra_segments` struct in a
//   CPU uint8 tensor; the device arrays it points to are owned by the Python RelCSR object):
//   ultra_mi::rspmm_plan_fwd(plan, relation, input, add_rows?, boundary_node?, boundary_value?, n_src, sum_op, mul_op) -> Tensor
//   ultra_mi::rspmm_plan_bwd(by_src?, by_rel?, relation, input, output?, output_grad, d_input(a!)?, accumulate, n_src, n_dst,
//       sum_op, mul_op) -> d_relation          (d_input is written -- or accumulated into -- in place)
#include <ATen/ATen.h>
#include <ATen/Parallel.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/autograd.h>
#include <torch/library.h>

#include <cfloat>
#include <cstring>
#include <tuple>
#include <vector>

#include "ultra_rspmm.h"

namespace {

using at::Tensor;
using c10::optional;

void check_status(int status, const char *what) {
    if (status == ULTRA_OK) return;
    if (status == ULTRA_ERR_HIP)
        TORCH_CHECK(false, "libultra_rspmm (", what, "): ", ultra_rspmm_status_string(status), " [hipError_t=",
                    ultra_rspmm_last_hip_error(), "]");
    TORCH_CHECK(false, "libultra_rspmm (", what, "): ", ultra_rspmm_status_string(status));
}

void *current_stream(const Tensor &t) {
    return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream();
}

void check_dense(const Tensor &t, const char *name, at::ScalarType dtype, const Tensor &like) {
    TORCH_CHECK(t.is_cuda(), "ultra_mi: ", name, " must be on an MI355X (HIP) device: this operator has no CPU kernel (only "
                "build_relcsr, rspmm_fwd and rspmm_bwd are registered on the CPU key)");
    TORCH_CHECK(t.scalar_type() == dtype, "ultra_mi: ", name, " has dtype ", t.scalar_type(), ", expected ", dtype);
    TORCH_CHECK(t.device() == like.device(), "ultra_mi: ", name, " is on ", t.device(), ", expected ", like.device());
}

const float *fptr(const optional<Tensor> &t) { return (t.has_value() && t->defined()) ? t->data_ptr<float>() : nullptr; }

// ------------------------------------------------------------------------------------------------ build_relcsr
std::vector<Tensor> build_relcsr(const Tensor &edge_list, const optional<Tensor> &edge_weight, int64_t num_node,
                                 int64_t num_relation) {
    TORCH_CHECK(edge_list.dim() == 2 && edge_list.size(1) == 3, "build_relcsr: edge_list must be (E, 3) rows of "
                "(node_in, node_out, relation), got ", edge_list.sizes());
    check_dense(edge_list, "edge_list", at::kLong, edge_list);
    c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(edge_list.device());
    const int64_t n_in = edge_list.size(0);
    // destination = node_out (the layers aggregate over adjacency.transpose(0, 1): ultra/layer.py:127,328)
    Tensor dst = edge_list.select(1, 1).contiguous(), src = edge_list.select(1, 0).contiguous(),
           rel = edge_list.select(1, 2).contiguous();
    Tensor weight;
    if (edge_weight.has_value() && edge_weight->defined()) {
        check_dense(*edge_weight, "edge_weight", at::kFloat, edge_list);
        TORCH_CHECK(edge_weight->numel() == n_in, "build_relcsr: one weight per edge expected");
        weight = edge_weight->contiguous();
    }
    auto i32 = edge_list.options().dtype(at::kInt);
    Tensor out_row = at::empty({n_in}, i32), out_col = at::empty({n_in}, i32), out_rel = at::empty({n_in}, i32);
    Tensor out_w = at::empty({n_in}, edge_list.options().dtype(at::kFloat));
    Tensor edge_of_input = at::empty({n_in}, edge_list.options());
    Tensor temp = at::empty({(int64_t)ultra_relcsr_coalesce_temp_bytes(n_in)}, edge_list.options().dtype(at::kByte));
    int64_t n_unique = 0;
    int unit = 1;
    check_status(ultra_relcsr_coalesce(dst.data_ptr<int64_t>(), src.data_ptr<int64_t>(), rel.data_ptr<int64_t>(),
                                       weight.defined() ? weight.data_ptr<float>() : nullptr, n_in, num_node, num_node,
                                       num_relation, out_row.data_ptr<int>(), out_col.data_ptr<int>(),
                                       out_rel.data_ptr<int>(), out_w.data_ptr<float>(), edge_of_input.data_ptr<int64_t>(),
                                       &n_unique, &unit, temp.data_ptr(), (size_t)temp.numel(), current_stream(edge_list)),
                 "ultra_relcsr_coalesce");
    out_row = out_row.narrow(0, 0, n_unique);
    Tensor rows = at::arange(num_node + 1, i32);
    Tensor row_ptr = at::searchsorted(out_row, rows, /*out_int32=*/true);
    return {row_ptr, out_col.narrow(0, 0, n_unique).contiguous(), out_rel.narrow(0, 0, n_unique).contiguous(),
            out_w.narrow(0, 0, n_unique).contiguous(), edge_of_input};
}

// ------------------------------------------------------------------------------------------------ raw CSR forward
struct CsrArgs {
    int64_t n_rows, n_edges, n_rel, F;
};

CsrArgs check_csr(const Tensor &row_ptr, const Tensor &src, const Tensor &rel, const optional<Tensor> &w,
                  const Tensor &relation, const Tensor &input, int64_t sum_op, int64_t mul_op) {
    TORCH_CHECK(sum_op >= 0 && sum_op <= 2 && mul_op >= 0 && mul_op <= 1, "ultra_mi: unknown sum/mul operator code");
    TORCH_CHECK(input.dim() == 2 && relation.dim() == 2, "ultra_mi: relation and input must be 2-D");
    TORCH_CHECK(relation.size(1) == input.size(1), "ultra_mi: Expect relation and input to have the same width, but found ",
                relation.size(1), " and ", input.size(1));
    check_dense(input, "input", at::kFloat, input);
    check_dense(relation, "relation", at::kFloat, input);
    check_dense(row_ptr, "row_ptr", at::kInt, input);
    check_dense(src, "src", at::kInt, input);
    check_dense(rel, "rel", at::kInt, input);
    TORCH_CHECK(row_ptr.dim() == 1 && row_ptr.numel() >= 1 && src.dim() == 1 && rel.sizes() == src.sizes(),
                "ultra_mi: row_ptr (N + 1,), src (E,), rel (E,) expected");
    if (w.has_value() && w->defined()) {
        check_dense(*w, "w", at::kFloat, input);
        TORCH_CHECK(w->sizes() == src.sizes(), "ultra_mi: one weight per edge expected");
    }
    return {row_ptr.numel() - 1, src.numel(), relation.size(0), input.size(1)};
}

Tensor rspmm_fwd_hip(const Tensor &row_ptr, const Tensor &src, const Tensor &rel, const optional<Tensor> &w,
                     const Tensor &relation, const Tensor &input, int64_t sum_op, int64_t mul_op) {
    const CsrArgs a = check_csr(row_ptr, src, rel, w, relation, input, sum_op, mul_op);
    TORCH_CHECK(a.F % 4 == 0, "ultra_mi::rspmm_fwd needs 16-byte rows (F % 4 == 0); use a RelCSR plan for other widths");
    c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(input.device());
    Tensor rp = row_ptr.contiguous(), s = src.contiguous(), r = rel.contiguous(), rl = relation.contiguous(),
           x = input.contiguous(), wt;
    if (w.has_value() && w->defined()) wt = w->contiguous();
    Tensor out = at::empty({a.n_rows, a.F}, input.options());
    if (out.numel() == 0) return out;
    check_status(ultra_rspmm_fwd_f32(rp.data_ptr<int>(), s.data_ptr<int>(), r.data_ptr<int>(),
                                     wt.defined() ? wt.data_ptr<float>() : nullptr, rl.data_ptr<float>(),
                                     x.data_ptr<float>(), out.data_ptr<float>(), a.n_rows, a.n_edges, a.n_rel, a.F,
                                     (int)sum_op, (int)mul_op, current_stream(input)),
                 "ultra_rspmm_fwd_f32");
    return out;
}

// ------------------------------------------------------------------------------------------------ raw CSR backward
// One ordered plan built on the fly (the raw-CSR operator keeps no state between calls; callers that train on one graph
// use the cached plans of ultra_torchdrug_amd.RelCSR instead).
struct OwnedPlan {
    ultra_segments seg{};
    std::vector<Tensor> keep;
};

OwnedPlan make_plan(const Tensor &row, const Tensor &node_a, const Tensor &node_b, const Tensor &rel, const Tensor &weight,
                    int64_t n_rows, int64_t n_node_a, int64_t n_rel, bool relation_plan, void *stream) {
    OwnedPlan plan;
    const int64_t E = row.numel();
    const int64_t piece_len = 256, chunk_edges = 32, chunk_rows = 64, slack = 16;
    auto i32 = row.options().dtype(at::kInt);
    const int64_t cap_chunks = n_rows + E / piece_len + 2, cap_long = E / piece_len + 1;
    Tensor chunks = at::empty({cap_chunks, 4}, i32), long_rows = at::empty({cap_long, 3}, i32);
    Tensor packed = at::empty({E + slack}, i32);
    Tensor temp = at::empty({(int64_t)ultra_relcsr_plan_temp_bytes(E, n_rows, piece_len)}, row.options().dtype(at::kByte));
    int64_t counts[4] = {0, 0, 0, 0};
    check_status(ultra_relcsr_plan(row.data_ptr<int>(), node_a.data_ptr<int>(), rel.data_ptr<int>(), E, n_rows, n_node_a,
                                   n_rel, relation_plan ? 1 : 0, 0, 1, chunk_edges, chunk_rows, piece_len,
                                   chunks.data_ptr<int>(), cap_chunks, long_rows.data_ptr<int>(), cap_long,
                                   E > 0 ? packed.data_ptr<int>() : nullptr, slack, counts, temp.data_ptr(),
                                   (size_t)temp.numel(), stream),
                 "ultra_relcsr_plan");
    Tensor node_a_kept = node_a;
    if (counts[3] == 32) node_a_kept = at::cat({node_a, at::zeros({slack}, i32)});
    Tensor weight_kept;
    if (weight.defined()) weight_kept = at::cat({weight, at::ones({slack}, weight.options())});
    ultra_segments &s = plan.seg;
    s.struct_bytes = (uint32_t)sizeof(ultra_segments);
    s.abi_version = (uint32_t)ULTRA_RSPMM_ABI_VERSION;
    s.n_rows = n_rows;
    s.n_edges = E;
    s.row = row.data_ptr<int>();
    s.node_a = node_a_kept.data_ptr<int>();
    s.node_b = node_b.defined() ? node_b.data_ptr<int>() : nullptr;
    s.rel = rel.data_ptr<int>();
    s.weight = weight_kept.defined() ? weight_kept.data_ptr<float>() : nullptr;
    s.n_chunks = counts[0];
    s.chunks = chunks.data_ptr<int>();
    s.n_long_rows = counts[1];
    s.long_rows = long_rows.data_ptr<int>();
    s.n_pieces = counts[2];
    s.piece_len = piece_len;
    s.packed = counts[3] ? reinterpret_cast<const uint32_t *>(packed.data_ptr<int>()) : nullptr;
    s.packed_src_shift = counts[3];
    plan.keep = {row, node_a_kept, rel, chunks, long_rows, packed};
    if (node_b.defined()) plan.keep.push_back(node_b);
    if (weight_kept.defined()) plan.keep.push_back(weight_kept);
    return plan;
}

std::tuple<Tensor, Tensor> rspmm_bwd_hip(const Tensor &row_ptr, const Tensor &src, const Tensor &rel,
                                         const optional<Tensor> &w, const Tensor &relation, const Tensor &input,
                                         const Tensor &output, const Tensor &output_grad, int64_t sum_op, int64_t mul_op) {
    const CsrArgs a = check_csr(row_ptr, src, rel, w, relation, input, sum_op, mul_op);
    check_dense(output_grad, "output_grad", at::kFloat, input);
    TORCH_CHECK(output_grad.dim() == 2 && output_grad.size(0) == a.n_rows && output_grad.size(1) == a.F,
                "ultra_mi::rspmm_bwd: output_grad must be (", a.n_rows, ", ", a.F, ")");
    c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(input.device());
    void *stream = current_stream(input);
    const int64_t n_src = input.size(0), E = a.n_edges;
    Tensor d_relation = at::empty_like(relation, at::MemoryFormat::Contiguous);
    Tensor d_input = at::empty_like(input, at::MemoryFormat::Contiguous);
    if (E == 0 || a.F == 0) return {d_relation.zero_(), d_input.zero_()};
    // destination of every edge, then the two orders the atomic-free backward reduces in (stable sorts keep the CSR
    // order inside a key: (src, dst, rel) for d_input, (rel, dst, src) for d_relation)
    auto i64 = input.options().dtype(at::kLong);
    Tensor counts = (row_ptr.narrow(0, 1, a.n_rows) - row_ptr.narrow(0, 0, a.n_rows)).to(at::kLong);
    Tensor dst = at::repeat_interleave(at::arange(a.n_rows, i64), counts, c10::nullopt, E);
    Tensor src64 = src.to(at::kLong), rel64 = rel.to(at::kLong);
    Tensor weight;
    if (w.has_value() && w->defined()) weight = w->contiguous();
    Tensor order_s = std::get<1>(at::sort(src64, /*stable=*/true, 0, false));
    Tensor order_r = std::get<1>(at::sort(rel64, /*stable=*/true, 0, false));
    auto take = [&](const Tensor &t, const Tensor &order) { return t.index_select(0, order).to(at::kInt).contiguous(); };
    OwnedPlan by_src = make_plan(take(src64, order_s), take(dst, order_s), Tensor(), take(rel64, order_s),
                                 weight.defined() ? weight.index_select(0, order_s) : Tensor(), n_src, a.n_rows, a.n_rel,
                                 false, stream);
    OwnedPlan by_rel = make_plan(take(rel64, order_r), take(src64, order_r), take(dst, order_r), take(rel64, order_r),
                                 weight.defined() ? weight.index_select(0, order_r) : Tensor(), a.n_rel, n_src, a.n_rel,
                                 true, stream);
    const int64_t n_ws = std::max(by_src.seg.n_pieces, by_rel.seg.n_pieces) * a.F;
    Tensor ws = at::empty({std::max<int64_t>(n_ws, 1)}, input.options());
    Tensor rl = relation.contiguous(), x = input.contiguous(), g = output_grad.contiguous(), o = output.contiguous();
    check_status(ultra_rspmm_backward_f32(&by_src.seg, &by_rel.seg, rl.data_ptr<float>(), x.data_ptr<float>(),
                                          o.data_ptr<float>(), g.data_ptr<float>(), d_input.data_ptr<float>(),
                                          d_relation.data_ptr<float>(), ws.data_ptr<float>(), (size_t)n_ws * 4, n_src,
                                          a.n_rows, a.n_rel, a.F, (int)sum_op, (int)mul_op, stream),
                 "ultra_rspmm_backward_f32");
    return {d_relation, d_input};
}

// autograd for the raw-CSR operator (counterpart of torchdrug's RSPMM*Function classes)
class RspmmCsrFunction : public torch::autograd::Function<RspmmCsrFunction> {
   public:
    static Tensor forward(torch::autograd::AutogradContext *ctx, const Tensor &row_ptr, const Tensor &src, const Tensor &rel,
                          const optional<Tensor> &w, const Tensor &relation, const Tensor &input, int64_t sum_op,
                          int64_t mul_op) {
        at::AutoDispatchBelowADInplaceOrView guard;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("ultra_mi::rspmm_fwd", "")
                             .typed<Tensor(const Tensor &, const Tensor &, const Tensor &, const optional<Tensor> &,
                                           const Tensor &, const Tensor &, int64_t, int64_t)>();
        Tensor out = op.call(row_ptr, src, rel, w, relation, input, sum_op, mul_op);
        ctx->save_for_backward({row_ptr, src, rel, (w.has_value() && w->defined()) ? *w : Tensor(), relation, input, out});
        ctx->saved_data["sum_op"] = sum_op;
        ctx->saved_data["mul_op"] = mul_op;
        return out;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext *ctx,
                                                   torch::autograd::variable_list grads) {
        auto saved = ctx->get_saved_variables();
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("ultra_mi::rspmm_bwd", "")
                             .typed<std::tuple<Tensor, Tensor>(const Tensor &, const Tensor &, const Tensor &,
                                                               const optional<Tensor> &, const Tensor &, const Tensor &,
                                                               const Tensor &, const Tensor &, int64_t, int64_t)>();
        optional<Tensor> w;
        if (saved[3].defined()) w = saved[3];
        auto result = op.call(saved[0], saved[1], saved[2], w, saved[4], saved[5], saved[6], grads[0].contiguous(),
                              ctx->saved_data["sum_op"].toInt(), ctx->saved_data["mul_op"].toInt());
        return {Tensor(), Tensor(), Tensor(), Tensor(), std::get<0>(result), std::get<1>(result), Tensor(), Tensor()};
    }
};

Tensor rspmm_fwd_autograd(const Tensor &row_ptr, const Tensor &src, const Tensor &rel, const optional<Tensor> &w,
                          const Tensor &relation, const Tensor &input, int64_t sum_op, int64_t mul_op) {
    return RspmmCsrFunction::apply(row_ptr, src, rel, w, relation, input, sum_op, mul_op);
}

// ------------------------------------------------------------------------------------------------ plan-based forms
const ultra_segments *plan_of(const optional<Tensor> &plan, const char *name) {
    if (!plan.has_value() || !plan->defined()) return nullptr;
    TORCH_CHECK(plan->device().is_cpu() && plan->scalar_type() == at::kByte && plan->is_contiguous() &&
                    plan->numel() == (int64_t)sizeof(ultra_segments),
                "ultra_mi: ", name, " must be the ", sizeof(ultra_segments), " bytes of an ultra_segments struct (CPU uint8)");
    return reinterpret_cast<const ultra_segments *>(plan->data_ptr<uint8_t>());
}

Tensor rspmm_plan_fwd(const Tensor &plan, const Tensor &relation, const Tensor &input, const optional<Tensor> &add_rows,
                      const optional<Tensor> &boundary_node, const optional<Tensor> &boundary_value, int64_t n_src,
                      int64_t sum_op, int64_t mul_op) {
    const ultra_segments *seg = plan_of(plan, "plan");
    TORCH_CHECK(seg != nullptr, "ultra_mi::rspmm_plan_fwd: plan is required");
    check_dense(input, "input", at::kFloat, input);
    check_dense(relation, "relation", at::kFloat, input);
    TORCH_CHECK(input.dim() == 2 && relation.dim() == 2 && input.size(1) == relation.size(1) && input.size(0) == n_src,
                "ultra_mi::rspmm_plan_fwd: relation (R, F) and input (n_src, F) expected");
    c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(input.device());
    const int64_t F = input.size(1), n_rel = relation.size(0);
    Tensor rl = relation.contiguous(), x = input.contiguous();
    Tensor out = at::empty({seg->n_rows, F}, input.options());
    if (out.numel() == 0) return out;
    const size_t ws_bytes = ultra_rspmm_workspace_bytes(seg, F);
    Tensor ws = at::empty({(int64_t)std::max<size_t>(ws_bytes / 4, 1)}, input.options());
    void *stream = current_stream(input);
    if (boundary_node.has_value() && boundary_node->defined()) {
        TORCH_CHECK(boundary_value.has_value() && boundary_value->defined() && !(add_rows.has_value() && add_rows->defined()),
                    "ultra_mi::rspmm_plan_fwd: give the boundary either dense (add_rows) or sparse (node, value)");
        check_dense(*boundary_node, "boundary_node", at::kInt, input);
        check_dense(*boundary_value, "boundary_value", at::kFloat, input);
        Tensor bv = boundary_value->contiguous(), bn = boundary_node->contiguous();
        TORCH_CHECK(bv.dim() == 2 && bv.size(0) == bn.numel() && bv.numel() == F, "ultra_mi: boundary must be (B,), (B, D) with B * D == F");
        check_status(ultra_rspmm_forward_boundary_f32(seg, rl.data_ptr<float>(), x.data_ptr<float>(), bn.data_ptr<int>(),
                                                      bv.data_ptr<float>(), bv.size(1), out.data_ptr<float>(),
                                                      ws.data_ptr<float>(), ws_bytes, n_src, n_rel, F, (int)sum_op,
                                                      (int)mul_op, stream),
                     "ultra_rspmm_forward_boundary_f32");
        return out;
    }
    Tensor add;
    if (add_rows.has_value() && add_rows->defined()) {
        check_dense(*add_rows, "add_rows", at::kFloat, input);
        TORCH_CHECK(add_rows->sizes() == out.sizes(), "ultra_mi: add_rows must have the shape of the output");
        add = add_rows->contiguous();
    }
    check_status(ultra_rspmm_forward_f32(seg, rl.data_ptr<float>(), x.data_ptr<float>(),
                                         add.defined() ? add.data_ptr<float>() : nullptr, out.data_ptr<float>(),
                                         ws.data_ptr<float>(), ws_bytes, n_src, n_rel, F, (int)sum_op, (int)mul_op, stream),
                 "ultra_rspmm_forward_f32");
    return out;
}

// d_input is an OUT argument (schema: Tensor(a!)?): the caller allocates it -- or hands over the gradient the same rows
// already hold from the layer's dense epilogue together with accumulate = true, and the kernels add the edge gradient
// INTO it (they read every element before they write it: no separate add pass, no extra (N, F) tensor).  Returns
// d_relation only, so no input is ever returned as an un-annotated output (dispatcher contract).
Tensor rspmm_plan_bwd(const optional<Tensor> &by_src, const optional<Tensor> &by_rel, const Tensor &relation,
                      const Tensor &input, const optional<Tensor> &output, const Tensor &output_grad,
                      const optional<Tensor> &d_input_out, bool accumulate, int64_t n_src, int64_t n_dst, int64_t sum_op,
                      int64_t mul_op) {
    const ultra_segments *s_src = plan_of(by_src, "by_src"), *s_rel = plan_of(by_rel, "by_rel");
    TORCH_CHECK(sum_op >= 0 && sum_op <= 2 && mul_op >= 0 && mul_op <= 1, "ultra_mi: unknown sum/mul operator code");
    check_dense(input, "input", at::kFloat, input);
    check_dense(relation, "relation", at::kFloat, input);
    check_dense(output_grad, "output_grad", at::kFloat, input);
    TORCH_CHECK(input.dim() == 2 && relation.dim() == 2 && input.size(1) == relation.size(1) && input.size(0) == n_src,
                "ultra_mi::rspmm_plan_bwd: relation (R, F) and input (n_src = ", n_src, ", F) expected, got ", relation.sizes(),
                " and ", input.sizes());
    const int64_t F = input.size(1), n_rel = relation.size(0);
    TORCH_CHECK(output_grad.dim() == 2 && output_grad.size(0) == n_dst && output_grad.size(1) == F,
                "ultra_mi::rspmm_plan_bwd: output_grad must be (", n_dst, ", ", F, "), got ", output_grad.sizes());
    TORCH_CHECK(!s_src || (s_src->n_rows == n_src), "ultra_mi::rspmm_plan_bwd: by_src plan has ", s_src ? s_src->n_rows : 0,
                " rows, n_src is ", n_src);
    TORCH_CHECK(!s_rel || (s_rel->n_rows == n_rel), "ultra_mi::rspmm_plan_bwd: by_rel plan has ", s_rel ? s_rel->n_rows : 0,
                " rows, relation has ", n_rel);
    c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(input.device());
    Tensor rl = relation.contiguous(), x = input.contiguous(), g = output_grad.contiguous(), o;
    if (output.has_value() && output->defined()) {
        check_dense(*output, "output", at::kFloat, input);
        TORCH_CHECK(output->sizes() == output_grad.sizes(), "ultra_mi::rspmm_plan_bwd: output must have the shape of output_grad");
        o = output->contiguous();
    }
    TORCH_CHECK(sum_op == 0 || o.defined(), "ultra_mi::rspmm_plan_bwd: min / max aggregation needs the forward output");
    Tensor d_input, d_relation = s_rel ? at::empty_like(rl) : at::empty({0}, input.options());
    if (s_src) {
        TORCH_CHECK(d_input_out.has_value() && d_input_out->defined(), "ultra_mi::rspmm_plan_bwd: d_input (out) is required with by_src");
        check_dense(*d_input_out, "d_input", at::kFloat, input);
        TORCH_CHECK(d_input_out->sizes() == x.sizes() && d_input_out->is_contiguous(),
                    "ultra_mi::rspmm_plan_bwd: d_input must be a contiguous tensor of the shape of input");
        TORCH_CHECK(!accumulate || sum_op == 0, "ultra_mi::rspmm_plan_bwd: accumulate is for sum aggregation");
        d_input = *d_input_out;
    }
    if (F == 0 || (!s_src && !s_rel)) return d_relation;
    const size_t ws_bytes = std::max(s_src ? ultra_rspmm_workspace_bytes(s_src, F) : 0,
                                     s_rel ? ultra_rspmm_workspace_bytes(s_rel, F) : 0);
    Tensor ws = at::empty({(int64_t)std::max<size_t>(ws_bytes / 4, 1)}, input.options());
    check_status(ultra_rspmm_backward_accumulate_f32(s_src, s_rel, rl.data_ptr<float>(), x.data_ptr<float>(),
                                                     o.defined() ? o.data_ptr<float>() : nullptr, g.data_ptr<float>(),
                                                     (s_src && accumulate) ? d_input.data_ptr<float>() : nullptr,
                                                     d_input.defined() ? d_input.data_ptr<float>() : nullptr,
                                                     s_rel ? d_relation.data_ptr<float>() : nullptr,
                                                     ws.data_ptr<float>(), ws_bytes, n_src, n_dst, n_rel, F, (int)sum_op,
                                                     (int)mul_op, current_stream(input)),
                 "ultra_rspmm_backward_accumulate_f32");
    return d_relation;
}

// ================================================================================================ CPU dispatch key
// (host kernels of the three raw-CSR operators; same schemas, same checks, same results as the HIP ones in the
// reference's summation order)
void check_host(const Tensor &t, const char *name, at::ScalarType dtype) {
    TORCH_CHECK(t.device().is_cpu(), "ultra_mi (CPU): ", name, " is on ", t.device(), "; all tensors of one call must share a device");
    TORCH_CHECK(t.scalar_type() == dtype, "ultra_mi: ", name, " has dtype ", t.scalar_type(), ", expected ", dtype);
}

std::vector<Tensor> build_relcsr_cpu(const Tensor &edge_list, const optional<Tensor> &edge_weight, int64_t num_node,
                                     int64_t num_relation) {
    TORCH_CHECK(edge_list.dim() == 2 && edge_list.size(1) == 3, "build_relcsr: edge_list must be (E, 3) rows of "
                "(node_in, node_out, relation), got ", edge_list.sizes());
    check_host(edge_list, "edge_list", at::kLong);
    const int64_t n_in = edge_list.size(0), n_rel = std::max<int64_t>(num_relation, 1);
    TORCH_CHECK((double)num_node * (double)num_node * (double)n_rel < 9.0e18, "build_relcsr: adjacency too large for a 64-bit key");
    Tensor el = edge_list.contiguous();
    const int64_t *e = el.data_ptr<int64_t>();
    Tensor weight;
    if (edge_weight.has_value() && edge_weight->defined()) {
        check_host(*edge_weight, "edge_weight", at::kFloat);
        TORCH_CHECK(edge_weight->numel() == n_in, "build_relcsr: one weight per edge expected");
        weight = edge_weight->contiguous();
    }
    const float *w_in = weight.defined() ? weight.data_ptr<float>() : nullptr;
    // coalesce(): stable sort by (destination = node_out, source = node_in, relation); duplicates of one triple merge
    // by adding their weights in input order
    Tensor key = at::empty({n_in}, el.options());
    int64_t *k = key.data_ptr<int64_t>();
    for (int64_t i = 0; i < n_in; ++i) {
        const int64_t src = e[3 * i], dst = e[3 * i + 1], rel = e[3 * i + 2];
        TORCH_CHECK(src >= 0 && src < num_node && dst >= 0 && dst < num_node && rel >= 0 && rel < n_rel,
                    "build_relcsr: edge ", i, " = (", src, ", ", dst, ", ", rel, ") is out of range");
        k[i] = (dst * num_node + src) * n_rel + rel;
    }
    auto sorted = at::sort(key, /*stable=*/true, 0, false);
    const int64_t *sk = std::get<0>(sorted).data_ptr<int64_t>(), *order = std::get<1>(sorted).data_ptr<int64_t>();
    auto i32 = el.options().dtype(at::kInt);
    Tensor out_src = at::empty({n_in}, i32), out_rel = at::empty({n_in}, i32), out_w = at::empty({n_in}, el.options().dtype(at::kFloat));
    Tensor row_ptr = at::zeros({num_node + 1}, i32), edge_of_input = at::empty({n_in}, el.options());
    int *ps = out_src.data_ptr<int>(), *pr = out_rel.data_ptr<int>(), *rp = row_ptr.data_ptr<int>();
    float *pw = out_w.data_ptr<float>();
    int64_t *eoi = edge_of_input.data_ptr<int64_t>();
    int64_t m = 0;
    for (int64_t i = 0; i < n_in; ++i) {
        const int64_t o = order[i];
        const float wi = w_in ? w_in[o] : 1.0f;
        if (i > 0 && sk[i] == sk[i - 1]) {
            pw[m - 1] = pw[m - 1] + wi;
        } else {
            ps[m] = (int)e[3 * o];
            pr[m] = (int)e[3 * o + 2];
            pw[m] = wi;
            rp[e[3 * o + 1] + 1] += 1;
            ++m;
        }
        eoi[o] = m - 1;
    }
    for (int64_t v = 0; v < num_node; ++v) rp[v + 1] += rp[v];
    return {row_ptr, out_src.narrow(0, 0, m).contiguous(), out_rel.narrow(0, 0, m).contiguous(),
            out_w.narrow(0, 0, m).contiguous(), edge_of_input};
}

CsrArgs check_csr_cpu(const Tensor &row_ptr, const Tensor &src, const Tensor &rel, const optional<Tensor> &w,
                      const Tensor &relation, const Tensor &input, int64_t sum_op, int64_t mul_op) {
    TORCH_CHECK(sum_op >= 0 && sum_op <= 2 && mul_op >= 0 && mul_op <= 1, "ultra_mi: unknown sum/mul operator code");
    TORCH_CHECK(input.dim() == 2 && relation.dim() == 2, "ultra_mi: relation and input must be 2-D");
    TORCH_CHECK(relation.size(1) == input.size(1), "ultra_mi: Expect relation and input to have the same width, but found ",
                relation.size(1), " and ", input.size(1));
    check_host(input, "input", at::kFloat);
    check_host(relation, "relation", at::kFloat);
    check_host(row_ptr, "row_ptr", at::kInt);
    check_host(src, "src", at::kInt);
    check_host(rel, "rel", at::kInt);
    TORCH_CHECK(row_ptr.dim() == 1 && row_ptr.numel() >= 1 && src.dim() == 1 && rel.sizes() == src.sizes(),
                "ultra_mi: row_ptr (N + 1,), src (E,), rel (E,) expected");
    if (w.has_value() && w->defined()) {
        check_host(*w, "w", at::kFloat);
        TORCH_CHECK(w->sizes() == src.sizes(), "ultra_mi: one weight per edge expected");
    }
    // the kernels trust the CSR: a row pointer that runs backwards or past E, a source outside `input` or a relation outside
    // `relation` would read -- and, in the backward sweep, WRITE -- out of bounds.  O(N + E) host scans, cheap next to the kernel.
    const int64_t n_rows = row_ptr.numel() - 1, n_edges = src.numel();
    {
        const Tensor rp = row_ptr.contiguous(), s = src.contiguous(), r = rel.contiguous();
        const int *p = rp.data_ptr<int>();
        TORCH_CHECK(p[0] == 0 && p[n_rows] == n_edges, "ultra_mi: row_ptr must start at 0 and end at E = ", n_edges, ", found ",
                    p[0], " .. ", p[n_rows]);
        for (int64_t i = 0; i < n_rows; ++i)
            TORCH_CHECK(p[i] <= p[i + 1], "ultra_mi: row_ptr decreases at row ", i, " (", p[i], " > ", p[i + 1], ")");
        const int *sp = s.data_ptr<int>(), *rlp = r.data_ptr<int>();
        int s_lo = 0, s_hi = -1, r_lo = 0, r_hi = -1;
        for (int64_t k = 0; k < n_edges; ++k) {
            s_lo = sp[k] < s_lo ? sp[k] : s_lo; s_hi = sp[k] > s_hi ? sp[k] : s_hi;
            r_lo = rlp[k] < r_lo ? rlp[k] : r_lo; r_hi = rlp[k] > r_hi ? rlp[k] : r_hi;
        }
        TORCH_CHECK(s_lo >= 0 && s_hi < input.size(0), "ultra_mi: src ids must lie in [0, ", input.size(0), "), found ", s_lo,
                    " .. ", s_hi);
        TORCH_CHECK(r_lo >= 0 && r_hi < relation.size(0), "ultra_mi: rel ids must lie in [0, ", relation.size(0), "), found ",
                    r_lo, " .. ", r_hi);
    }
    return {n_rows, n_edges, relation.size(0), input.size(1)};
}

// the operator pair as compile-time constants: the column loops then vectorise (element-wise, no reassociation)
template <int SUM> inline float reduce_op(float acc, float y) {
    if (SUM == 0) return acc + y;
    if (SUM == 1) return y < acc ? y : acc;      // keeps the accumulator unless the message is strictly better
    return y > acc ? y : acc;
}
template <int MUL> inline float message_op(float r, float x) { return MUL == 0 ? r * x : r + x; }

// One task = one destination row x one slab of 32 / 64 / 256 columns, accumulated in a local array (the compiler keeps it in vector
// registers) strictly in the row's edge order -- element-wise vector code, no reassociation, no FMA (-ffp-contract=off; the
// `target` attributes below add no fma feature): the same bits on every ISA level.  The body is stamped three times -- baseline
// x86-64, AVX2, AVX-512 -- because the library is built once and travels to other hosts: the level is picked at run time
// (__builtin_cpu_supports).  (clang's target_clones does not take function templates.)  FB15k237Inductive-v1-shaped relation
// graph (360 nodes, 518 k edges, F = 1 024) on 8 cores: 42 ms per call as round 3 wrote it (256-column slabs accumulated in
// memory, SSE2), 9 ms with AVX-512.
#define ULTRA_DEFINE_ROW_TASKS(NAME, ATTR, SLAB)                                                                                 \
    template <int SUM, int MUL, bool HAS_W>                                                                                  \
    ATTR void NAME(const int *row_ptr, const int *src, const int *rel, const float *w, const float *relation, const float *x, \
                   float *out, int64_t F, int64_t n_slab, int64_t t0, int64_t t1) {                                          \
        constexpr int64_t slab = SLAB;                                                                                       \
        const float identity = SUM == 0 ? 0.0f : (SUM == 1 ? FLT_MAX : -FLT_MAX); /* NaryMin / NaryMax ::zero */              \
        for (int64_t t = t0; t < t1; ++t) {                                                                                  \
            const int64_t v = t / n_slab, f0 = (t % n_slab) * slab;                                                          \
            const int64_t n = std::min<int64_t>(slab, F - f0);                                                               \
            float acc[slab];                                                                                                 \
            for (int64_t j = 0; j < slab; ++j) acc[j] = identity;                                                            \
            const int64_t k0 = row_ptr[v], k1 = row_ptr[v + 1];                                                              \
            if (n == slab) {                                                                                                 \
                for (int64_t k = k0; k < k1; ++k) {                                                                          \
                    const float *__restrict__ xr = x + (int64_t)src[k] * F + f0;                                             \
                    const float *__restrict__ rr = relation + (int64_t)rel[k] * F + f0;                                      \
                    const float wk = HAS_W ? w[k] : 1.0f;                                                                    \
                    for (int64_t j = 0; j < slab; ++j) {                                                                     \
                        const float m = message_op<MUL>(rr[j], xr[j]);                                                       \
                        acc[j] = reduce_op<SUM>(acc[j], HAS_W ? wk * m : m);                                                 \
                    }                                                                                                        \
                }                                                                                                            \
            } else {                                                                                                         \
                for (int64_t k = k0; k < k1; ++k) {                                                                          \
                    const float *__restrict__ xr = x + (int64_t)src[k] * F + f0;                                             \
                    const float *__restrict__ rr = relation + (int64_t)rel[k] * F + f0;                                      \
                    const float wk = HAS_W ? w[k] : 1.0f;                                                                    \
                    for (int64_t j = 0; j < n; ++j) {                                                                        \
                        const float m = message_op<MUL>(rr[j], xr[j]);                                                       \
                        acc[j] = reduce_op<SUM>(acc[j], HAS_W ? wk * m : m);                                                 \
                    }                                                                                                        \
                }                                                                                                            \
            }                                                                                                                \
            float *__restrict__ o = out + v * F + f0;                                                                        \
            for (int64_t j = 0; j < n; ++j) o[j] = acc[j];                                                                   \
        }                                                                                                                    \
    }
// slab = what the level's vector registers hold as accumulators: 8 x 4, 8 x 8, 16 x 16 floats
constexpr int64_t kSlabBase = 32, kSlabAvx2 = 64, kSlabAvx512 = 256;
ULTRA_DEFINE_ROW_TASKS(row_tasks_base, , kSlabBase)
ULTRA_DEFINE_ROW_TASKS(row_tasks_avx2, __attribute__((target("avx2"))), kSlabAvx2)
ULTRA_DEFINE_ROW_TASKS(row_tasks_avx512, __attribute__((target("avx512f"))), kSlabAvx512)
#undef ULTRA_DEFINE_ROW_TASKS

int cpu_isa_level() {       // 0: baseline, 1: AVX2, 2: AVX-512F (ULTRA_CPU_ISA=0|1|2 caps it: A/B runs, tests of every level)
    static const int level = [] {
        int have = __builtin_cpu_supports("avx512f") ? 2 : (__builtin_cpu_supports("avx2") ? 1 : 0);
        if (const char *cap = getenv("ULTRA_CPU_ISA")) have = std::min(have, std::max(0, atoi(cap)));
        return have;
    }();
    return level;
}

template <int SUM, int MUL, bool HAS_W>
void rspmm_rows_cpu(const int *row_ptr, const int *src, const int *rel, const float *w, const float *relation, const float *x,
                    float *out, int64_t n_rows, int64_t F) {
    // a task = one destination row x one slab of columns: a hub row's edges are walked by several threads
    const int isa = cpu_isa_level();
    const int64_t slab = isa == 2 ? kSlabAvx512 : (isa == 1 ? kSlabAvx2 : kSlabBase);
    const int64_t n_slab = (F + slab - 1) / slab;
    at::parallel_for(0, n_rows * n_slab, 1, [&](int64_t t0, int64_t t1) {
        if (isa == 2) row_tasks_avx512<SUM, MUL, HAS_W>(row_ptr, src, rel, w, relation, x, out, F, n_slab, t0, t1);
        else if (isa == 1) row_tasks_avx2<SUM, MUL, HAS_W>(row_ptr, src, rel, w, relation, x, out, F, n_slab, t0, t1);
        else row_tasks_base<SUM, MUL, HAS_W>(row_ptr, src, rel, w, relation, x, out, F, n_slab, t0, t1);
    });
}

Tensor rspmm_fwd_cpu(const Tensor &row_ptr, const Tensor &src, const Tensor &rel, const optional<Tensor> &w,
                     const Tensor &relation, const Tensor &input, int64_t sum_op, int64_t mul_op) {
    const CsrArgs a = check_csr_cpu(row_ptr, src, rel, w, relation, input, sum_op, mul_op);
    Tensor rp = row_ptr.contiguous(), s = src.contiguous(), r = rel.contiguous(), rl = relation.contiguous(),
           x = input.contiguous(), wt;
    if (w.has_value() && w->defined()) wt = w->contiguous();
    Tensor out = at::empty({a.n_rows, a.F}, input.options());
    if (out.numel() == 0) return out;
    const float *wp = wt.defined() ? wt.data_ptr<float>() : nullptr;
#define ULTRA_CPU_FWD(S, M)                                                                                              \
    (wp ? rspmm_rows_cpu<S, M, true>(rp.data_ptr<int>(), s.data_ptr<int>(), r.data_ptr<int>(), wp, rl.data_ptr<float>(),  \
                                     x.data_ptr<float>(), out.data_ptr<float>(), a.n_rows, a.F)                          \
        : rspmm_rows_cpu<S, M, false>(rp.data_ptr<int>(), s.data_ptr<int>(), r.data_ptr<int>(), wp, rl.data_ptr<float>(), \
                                      x.data_ptr<float>(), out.data_ptr<float>(), a.n_rows, a.F))
    switch (sum_op * 2 + mul_op) {
        case 0: ULTRA_CPU_FWD(0, 0); break;
        case 1: ULTRA_CPU_FWD(0, 1); break;
        case 2: ULTRA_CPU_FWD(1, 0); break;
        case 3: ULTRA_CPU_FWD(1, 1); break;
        case 4: ULTRA_CPU_FWD(2, 0); break;
        default: ULTRA_CPU_FWD(2, 1); break;
    }
#undef ULTRA_CPU_FWD
    return out;
}

// Backward on the host: the reference's sequential sweep over the CSR (`d_input[src] += ...`, `d_relation[rel] += ...`
// edge after edge in sorted order), made race-free by giving every thread its own slab of COLUMNS -- each element of
// either gradient is still accumulated by one thread in CSR order.  Per edge and element
//     contribution = ((g * [out == y]) * w) * d(rel (*|+) x)/d{x, rel}        ([.] only for min / max; every tie is fed)
template <int SUM, int MUL, bool HAS_W>
void rspmm_sweep_backward_cpu(const int *row_ptr, const int *src, const int *rel, const float *w, const float *relation,
                              const float *x, const float *out, const float *g, float *d_rel, float *d_x, int64_t n_rows,
                              int64_t F) {
    const int64_t slab = 64, n_slab = (F + slab - 1) / slab;
    at::parallel_for(0, n_slab, 1, [&](int64_t s0, int64_t s1) {
        for (int64_t sl = s0; sl < s1; ++sl) {
            const int64_t f0 = sl * slab, f1 = std::min(F, f0 + slab);
            for (int64_t v = 0; v < n_rows; ++v) {
                const float *__restrict__ gr = g + v * F;
                const float *__restrict__ orow = out ? out + v * F : nullptr;
                for (int64_t k = row_ptr[v]; k < row_ptr[v + 1]; ++k) {
                    const float *__restrict__ xr = x + (int64_t)src[k] * F;
                    const float *__restrict__ rr = relation + (int64_t)rel[k] * F;
                    float *__restrict__ dx = d_x + (int64_t)src[k] * F;
                    float *__restrict__ dr = d_rel + (int64_t)rel[k] * F;
                    const float wk = HAS_W ? w[k] : 1.0f;
                    for (int64_t f = f0; f < f1; ++f) {
                        float gm = gr[f];
                        if (SUM != 0) gm = gm * ((orow[f] == wk * message_op<MUL>(rr[f], xr[f])) ? 1.0f : 0.0f);
                        const float gw = gm * wk;
                        dx[f] = dx[f] + gw * (MUL == 0 ? rr[f] : 1.0f);
                        dr[f] = dr[f] + gw * (MUL == 0 ? xr[f] : 1.0f);
                    }
                }
            }
        }
    });
}

std::tuple<Tensor, Tensor> rspmm_bwd_cpu(const Tensor &row_ptr, const Tensor &src, const Tensor &rel,
                                         const optional<Tensor> &w, const Tensor &relation, const Tensor &input,
                                         const Tensor &output, const Tensor &output_grad, int64_t sum_op, int64_t mul_op) {
    const CsrArgs a = check_csr_cpu(row_ptr, src, rel, w, relation, input, sum_op, mul_op);
    check_host(output_grad, "output_grad", at::kFloat);
    TORCH_CHECK(output_grad.dim() == 2 && output_grad.size(0) == a.n_rows && output_grad.size(1) == a.F,
                "ultra_mi::rspmm_bwd: output_grad must be (", a.n_rows, ", ", a.F, ")");
    if (sum_op != 0) {
        check_host(output, "output", at::kFloat);
        TORCH_CHECK(output.sizes() == output_grad.sizes(), "ultra_mi::rspmm_bwd: output must have the shape of output_grad");
    }
    Tensor d_relation = at::zeros_like(relation, at::MemoryFormat::Contiguous);
    Tensor d_input = at::zeros_like(input, at::MemoryFormat::Contiguous);
    if (a.n_edges == 0 || a.F == 0) return {d_relation, d_input};
    Tensor rp = row_ptr.contiguous(), s = src.contiguous(), r = rel.contiguous(), rl = relation.contiguous(),
           x = input.contiguous(), g = output_grad.contiguous(), o, wt;
    if (sum_op != 0) o = output.contiguous();
    if (w.has_value() && w->defined()) wt = w->contiguous();
    const float *wp = wt.defined() ? wt.data_ptr<float>() : nullptr;
    const float *op = o.defined() ? o.data_ptr<float>() : nullptr;
#define ULTRA_CPU_BWD(S, M)                                                                                                  \
    (wp ? rspmm_sweep_backward_cpu<S, M, true>(rp.data_ptr<int>(), s.data_ptr<int>(), r.data_ptr<int>(), wp,                 \
                                               rl.data_ptr<float>(), x.data_ptr<float>(), op, g.data_ptr<float>(),           \
                                               d_relation.data_ptr<float>(), d_input.data_ptr<float>(), a.n_rows, a.F)       \
        : rspmm_sweep_backward_cpu<S, M, false>(rp.data_ptr<int>(), s.data_ptr<int>(), r.data_ptr<int>(), wp,                \
                                                rl.data_ptr<float>(), x.data_ptr<float>(), op, g.data_ptr<float>(),          \
                                                d_relation.data_ptr<float>(), d_input.data_ptr<float>(), a.n_rows, a.F))
    switch (sum_op * 2 + mul_op) {
        case 0: ULTRA_CPU_BWD(0, 0); break;
        case 1: ULTRA_CPU_BWD(0, 1); break;
        case 2: ULTRA_CPU_BWD(1, 0); break;
        case 3: ULTRA_CPU_BWD(1, 1); break;
        case 4: ULTRA_CPU_BWD(2, 0); break;
        default: ULTRA_CPU_BWD(2, 1); break;
    }
#undef ULTRA_CPU_BWD
    return {d_relation, d_input};
}

}  // namespace

TORCH_LIBRARY(ultra_mi, m) {
    m.def("build_relcsr(Tensor edge_list, Tensor? edge_weight, int num_node, int num_relation) -> Tensor[]");
    m.def("rspmm_fwd(Tensor row_ptr, Tensor src, Tensor rel, Tensor? w, Tensor relation, Tensor input, int sum_op, int mul_op) -> Tensor");
    m.def("rspmm_bwd(Tensor row_ptr, Tensor src, Tensor rel, Tensor? w, Tensor relation, Tensor input, Tensor output, "
          "Tensor output_grad, int sum_op, int mul_op) -> (Tensor, Tensor)");
    m.def("rspmm_plan_fwd(Tensor plan, Tensor relation, Tensor input, Tensor? add_rows, Tensor? boundary_node, "
          "Tensor? boundary_value, int n_src, int sum_op, int mul_op) -> Tensor");
    m.def("rspmm_plan_bwd(Tensor? by_src, Tensor? by_rel, Tensor relation, Tensor input, Tensor? output, Tensor output_grad, "
          "Tensor(a!)? d_input, bool accumulate, int n_src, int n_dst, int sum_op, int mul_op) -> Tensor");
    m.def("abi_version() -> int", []() -> int64_t { return ultra_rspmm_abi_version(); });
}

// "CUDA" is the dispatch key of HIP tensors in a ROCm build of PyTorch
TORCH_LIBRARY_IMPL(ultra_mi, CUDA, m) {
    m.impl("build_relcsr", build_relcsr);
    m.impl("rspmm_fwd", rspmm_fwd_hip);
    m.impl("rspmm_bwd", rspmm_bwd_hip);
}

// host kernels of the same three operators (config 1: `--gpus null`)
TORCH_LIBRARY_IMPL(ultra_mi, CPU, m) {
    m.impl("build_relcsr", build_relcsr_cpu);
    m.impl("rspmm_fwd", rspmm_fwd_cpu);
    m.impl("rspmm_bwd", rspmm_bwd_cpu);
}

// the plan tensor lives on the CPU while the dense operands live on the device: no single backend key fits
TORCH_LIBRARY_IMPL(ultra_mi, CompositeExplicitAutograd, m) {
    m.impl("rspmm_plan_fwd", rspmm_plan_fwd);
    m.impl("rspmm_plan_bwd", rspmm_plan_bwd);
}

TORCH_LIBRARY_IMPL(ultra_mi, Autograd, m) {
    m.impl("rspmm_fwd", rspmm_fwd_autograd);
}
