// trk_torch_ops.cpp -- the fused rollout as a NATIVE PyTorch dispatcher op (libtrk_torch.so, loaded with torch.ops.load_library).
//
// north star: "exposed to Python through PyTorch-ROCm custom ops with explicit backward kernels so the Robot/Task API ... [is] a
// drop-in for autograd".  custom_ops.py registers the ops from Python (torch.library.custom_op): correct, traceable, but every call
// pays the Python custom-op machinery twice (forward and backward: 172 us for `compute_collision_cost(q).sum().backward()` where the
// kernel takes 9, profiles/r03_bench_task_api.txt).  Here the two ops of that idiom live in C++:
//
//   trk::rollout(q, model, cost_model, w_self, w_obj, w_ws, w_ee, want_pos) -> (cost, gq, link_pos)
//       reference call site: PlanningTask.compute_collision_cost tasks.py:135-137 (FK + three collision fields [+ EE])
//       CUDA(HIP) kernel: trk_rollout_cost_grad / trk_rollout_cost_grad_f16 of libtrk.so (include/trk.h)
//       Autograd: backward = trk::scale_rows_native(saved gq, grad of cost) -- the explicit backward kernel trk_scale_rows; the forward
//                 kernel already produced d cost / d q.  gq and link_pos are by-products (the reference's graph would not differentiate
//                 its own gradient either): they are marked non-differentiable, not given made-up zero gradients.
//       Meta: shapes only (torch.compile / fake tensors).
//   trk::scale_rows_native(g, scale) -> g * scale[..., None]   (scale may be an expanded scalar: `.sum().backward()` hands one down;
//       the Python-registered twin is trk::scale_rows of custom_ops.py)
//
// `model` / `cost_model` are the C handles (TrkModel* / TrkCostModel* as integers): plain ints to the dispatcher and to
// torch.compile, nothing to look up.  The Python objects that own them (ops.ModelHandle / ops.CostHandle) must outlive the call.
//
// torch is plumbing here as everywhere: tensors, streams, the autograd graph.  All arithmetic is in libtrk.so.
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include "../../include/trk.h"

namespace {

using at::Tensor;

void trk_check(int rc, const char* what) {
    TORCH_CHECK(rc == TRK_OK, what, ": ", trk_last_error());
}

trk_stream_t current_stream(const Tensor& t) {
    return reinterpret_cast<trk_stream_t>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}

struct Shapes {
    int64_t batch, horizon, n;
    std::vector<int64_t> lead;
};

Shapes shapes_of(const Tensor& q, int64_t n_dofs) {
    TORCH_CHECK(q.dim() >= 1 && q.size(-1) == n_dofs, "trk::rollout: q has ", q.size(-1), " columns, the model has ", n_dofs, " DOF");
    Shapes s;
    s.lead.assign(q.sizes().begin(), q.sizes().end() - 1);
    if (q.dim() == 3) { s.batch = q.size(0); s.horizon = q.size(1); }
    else { s.batch = q.numel() / n_dofs; s.horizon = 1; }
    s.n = s.batch * s.horizon;
    return s;
}

// ---------------------------------------------------------------------------------------------------------------------------
// trk::rollout -- device implementation
// ---------------------------------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor> rollout_hip(const Tensor& q_in, int64_t model, int64_t cm, double w_self, double w_obj, double w_ws,
                                               double w_ee, bool want_pos) {
    // the handles travel as integers: validate them against libtrk.so's registry of live handles before they are dereferenced
    TORCH_CHECK(trk_handle_kind(reinterpret_cast<const void*>(model)) == 1, "trk::rollout: ", model, " is not a live TrkModel handle (destroyed, or never created)");
    TORCH_CHECK(trk_handle_kind(reinterpret_cast<const void*>(cm)) == 2, "trk::rollout: ", cm, " is not a live TrkCostModel handle (destroyed, or never created)");
    const TrkModel* m = reinterpret_cast<const TrkModel*>(model);
    const TrkCostModel* c = reinterpret_cast<const TrkCostModel*>(cm);
    const bool f16 = q_in.scalar_type() == at::kHalf;
    TORCH_CHECK(f16 || q_in.scalar_type() == at::kFloat, "trk::rollout: q must be float32 or float16");
    const Tensor q = q_in.contiguous();
    const int64_t D = trk_model_n_dofs(m), L = trk_model_n_links(m);
    const Shapes s = shapes_of(q, D);
    auto opt32 = q.options().dtype(at::kFloat);
    std::vector<int64_t> pos_shape = s.lead, g_shape = s.lead;
    pos_shape.push_back(L); pos_shape.push_back(3);
    g_shape.push_back(D);
    Tensor cost = at::empty(s.lead, opt32);
    Tensor gq = at::empty(g_shape, q.options());
    Tensor pos = want_pos ? at::empty(pos_shape, q.options()) : at::empty({0}, q.options());
    const TrkRolloutWeights w{(float)w_self, (float)w_obj, (float)w_ws, (float)w_ee};
    c10::DeviceGuard guard(q.device());      // the generic guard: PyTorch-ROCm registers its HIP implementation under the "cuda" device type
    if (f16)
        trk_check(trk_rollout_cost_grad_f16(m, c, &w, q.data_ptr(), s.batch, s.horizon, want_pos ? pos.data_ptr() : nullptr,
                                            cost.data_ptr<float>(), gq.data_ptr(), TRK_F16, 1.0f, nullptr, current_stream(q)),
                  "trk_rollout_cost_grad_f16");
    else
        trk_check(trk_rollout_cost_grad(m, c, &w, q.data_ptr<float>(), s.batch, s.horizon, want_pos ? pos.data_ptr<float>() : nullptr,
                                        cost.data_ptr<float>(), gq.data_ptr<float>(), nullptr, current_stream(q)),
                  "trk_rollout_cost_grad");
    return {cost, gq, pos};
}

std::tuple<Tensor, Tensor, Tensor> rollout_meta(const Tensor& q, int64_t model, int64_t cm, double, double, double, double, bool want_pos) {
    TORCH_CHECK(trk_handle_kind(reinterpret_cast<const void*>(model)) == 1, "trk::rollout: ", model, " is not a live TrkModel handle");
    const TrkModel* m = reinterpret_cast<const TrkModel*>(model);
    const int64_t D = trk_model_n_dofs(m), L = trk_model_n_links(m);     // host-side queries of the handle: no device work
    const Shapes s = shapes_of(q, D);
    std::vector<int64_t> pos_shape = s.lead, g_shape = s.lead;
    pos_shape.push_back(L); pos_shape.push_back(3);
    g_shape.push_back(D);
    return {at::empty(s.lead, q.options().dtype(at::kFloat)), at::empty(g_shape, q.options()),
            want_pos ? at::empty(pos_shape, q.options()) : at::empty({0}, q.options())};
}

// ---------------------------------------------------------------------------------------------------------------------------
// trk::scale_rows -- out[n, :] = g[n, :] * scale[n]; an expanded scalar (all strides 0) is read as ONE value
// ---------------------------------------------------------------------------------------------------------------------------
Tensor scale_rows_hip(const Tensor& g_in, const Tensor& scale_in) {
    const Tensor g = g_in.contiguous();
    TORCH_CHECK(g.dim() >= 1 && (g.scalar_type() == at::kFloat || g.scalar_type() == at::kHalf), "trk::scale_rows: g must be float32 / float16");
    const int64_t D = g.size(-1), n = D ? g.numel() / D : 0;
    TORCH_CHECK(scale_in.numel() == n && scale_in.scalar_type() == at::kFloat && scale_in.device() == g.device(),
                "trk::scale_rows: scale must be float32 with one element per row of g, on g's device");
    bool scalar = n > 0;
    for (int64_t k = 0; k < scale_in.dim(); ++k) scalar = scalar && (scale_in.size(k) == 1 || scale_in.stride(k) == 0);
    const Tensor scale = scalar ? scale_in : scale_in.contiguous();
    Tensor out = at::empty_like(g);
    if (n == 0) return out;
    c10::DeviceGuard guard(g.device());
    trk_check(trk_scale_rows(g.data_ptr(), scale.data_ptr<float>(), scalar ? 0 : 1, n, (int32_t)D,
                             g.scalar_type() == at::kHalf ? TRK_F16 : TRK_F32, out.data_ptr(), current_stream(g)),
              "trk_scale_rows");
    return out;
}

Tensor scale_rows_meta(const Tensor& g, const Tensor&) { return at::empty_like(g, g.options(), at::MemoryFormat::Contiguous); }

// ---------------------------------------------------------------------------------------------------------------------------
// autograd: one node whose backward is the explicit kernel
// ---------------------------------------------------------------------------------------------------------------------------
struct RolloutFn : public torch::autograd::Function<RolloutFn> {
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const Tensor& q, int64_t model, int64_t cm,
                                                  double w_self, double w_obj, double w_ws, double w_ee, bool want_pos) {
        at::AutoDispatchBelowADInplaceOrView below;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("trk::rollout", "")
                             .typed<std::tuple<Tensor, Tensor, Tensor>(const Tensor&, int64_t, int64_t, double, double, double, double, bool)>();
        auto [cost, gq, pos] = op.call(q, model, cm, w_self, w_obj, w_ws, w_ee, want_pos);
        ctx->save_for_backward({gq});
        ctx->set_materialize_grads(false);
        // gq (d cost / d q itself) and link_pos are by-products of the fused kernel: flagged non-differentiable -- `requires_grad` is
        // False on them, differentiating through them alone raises, no zero gradient is made up (fk_map_collision gives
        // differentiable link positions)
        ctx->mark_non_differentiable({gq, pos});
        return {cost, gq, pos};
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        Tensor out;
        if (grads[0].defined()) {
            static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("trk::scale_rows_native", "").typed<Tensor(const Tensor&, const Tensor&)>();
            out = op.call(ctx->get_saved_variables()[0], grads[0]);
        }
        return {out, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

std::tuple<Tensor, Tensor, Tensor> rollout_autograd(const Tensor& q, int64_t model, int64_t cm, double w_self, double w_obj, double w_ws,
                                                    double w_ee, bool want_pos) {
    auto out = RolloutFn::apply(q, model, cm, w_self, w_obj, w_ws, w_ee, want_pos);
    return {out[0], out[1], out[2]};
}

}  // namespace

TORCH_LIBRARY_FRAGMENT(trk, m) {
    m.def("rollout(Tensor q, int model, int cost_model, float w_self, float w_obj, float w_ws, float w_ee, bool want_pos) -> (Tensor, Tensor, Tensor)");
    m.def("scale_rows_native(Tensor g, Tensor scale) -> Tensor");
}
TORCH_LIBRARY_IMPL(trk, CUDA, m) {          // "CUDA" is the dispatch key of HIP devices in PyTorch-ROCm
    m.impl("rollout", rollout_hip);
    m.impl("scale_rows_native", scale_rows_hip);
}
TORCH_LIBRARY_IMPL(trk, Meta, m) {
    m.impl("rollout", rollout_meta);
    m.impl("scale_rows_native", scale_rows_meta);
}
TORCH_LIBRARY_IMPL(trk, Autograd, m) {
    m.impl("rollout", rollout_autograd);
}
