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
//   ultra_mi::rspmm_plan_bwd(by_src?, by_rel?, relation, input, output?, output_grad, d_input_add?, n_src, n_dst, sum_op, mul_op)
//       -> (d_input, d_relation)
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/autograd.h>
#include <torch/library.h>

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
    TORCH_CHECK(t.is_cuda(), "ultra_mi: ", name, " must be on an MI355X (HIP) device; there is no CPU fallback");
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

std::tuple<Tensor, Tensor> rspmm_plan_bwd(const optional<Tensor> &by_src, const optional<Tensor> &by_rel,
                                          const Tensor &relation, const Tensor &input, const optional<Tensor> &output,
                                          const Tensor &output_grad, const optional<Tensor> &d_input_add, int64_t n_src,
                                          int64_t n_dst, int64_t sum_op, int64_t mul_op) {
    const ultra_segments *s_src = plan_of(by_src, "by_src"), *s_rel = plan_of(by_rel, "by_rel");
    check_dense(input, "input", at::kFloat, input);
    check_dense(relation, "relation", at::kFloat, input);
    check_dense(output_grad, "output_grad", at::kFloat, input);
    c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(input.device());
    const int64_t F = input.size(1), n_rel = relation.size(0);
    Tensor rl = relation.contiguous(), x = input.contiguous(), g = output_grad.contiguous(), o;
    if (output.has_value() && output->defined()) o = output->contiguous();
    // d_input_add: the gradient the same rows receive from the layer's dense epilogue; accumulated IN PLACE (the kernels
    // read every element before they write it), so no separate add pass and no extra (N, F) tensor
    Tensor d_input, d_relation = s_rel ? at::empty_like(rl) : Tensor();
    const float *add_ptr = nullptr;
    if (s_src) {
        if (d_input_add.has_value() && d_input_add->defined()) {
            check_dense(*d_input_add, "d_input_add", at::kFloat, input);
            TORCH_CHECK(d_input_add->sizes() == x.sizes() && d_input_add->is_contiguous(),
                        "ultra_mi::rspmm_plan_bwd: d_input_add must be a contiguous tensor of the shape of input");
            d_input = *d_input_add;
            add_ptr = d_input.data_ptr<float>();
        } else {
            d_input = at::empty_like(x);
        }
    }
    if (F == 0 || (!s_src && !s_rel)) return {d_input, d_relation};
    const size_t ws_bytes = std::max(s_src ? ultra_rspmm_workspace_bytes(s_src, F) : 0,
                                     s_rel ? ultra_rspmm_workspace_bytes(s_rel, F) : 0);
    Tensor ws = at::empty({(int64_t)std::max<size_t>(ws_bytes / 4, 1)}, input.options());
    check_status(ultra_rspmm_backward_accumulate_f32(s_src, s_rel, rl.data_ptr<float>(), x.data_ptr<float>(),
                                                     o.defined() ? o.data_ptr<float>() : nullptr, g.data_ptr<float>(), add_ptr,
                                                     d_input.defined() ? d_input.data_ptr<float>() : nullptr,
                                                     d_relation.defined() ? d_relation.data_ptr<float>() : nullptr,
                                                     ws.data_ptr<float>(), ws_bytes, n_src, n_dst, n_rel, F, (int)sum_op,
                                                     (int)mul_op, current_stream(input)),
                 "ultra_rspmm_backward_accumulate_f32");
    return {d_input.defined() ? d_input : at::empty({0}, input.options()),
            d_relation.defined() ? d_relation : at::empty({0}, input.options())};
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
          "Tensor(a!)? d_input_add, int n_src, int n_dst, int sum_op, int mul_op) -> (Tensor, Tensor)");
    m.def("abi_version() -> int", []() -> int64_t { return ultra_rspmm_abi_version(); });
}

// "CUDA" is the dispatch key of HIP tensors in a ROCm build of PyTorch
TORCH_LIBRARY_IMPL(ultra_mi, CUDA, m) {
    m.impl("build_relcsr", build_relcsr);
    m.impl("rspmm_fwd", rspmm_fwd_hip);
    m.impl("rspmm_bwd", rspmm_bwd_hip);
}

// the plan tensor lives on the CPU while the dense operands live on the device: no single backend key fits
TORCH_LIBRARY_IMPL(ultra_mi, CompositeExplicitAutograd, m) {
    m.impl("rspmm_plan_fwd", rspmm_plan_fwd);
    m.impl("rspmm_plan_bwd", rspmm_plan_bwd);
}

TORCH_LIBRARY_IMPL(ultra_mi, Autograd, m) {
    m.impl("rspmm_fwd", rspmm_fwd_autograd);
}
