// crender_torch.cpp — torch C++ extension over the C ABI of include/crender_hip.h.
//
// The binding north_star names ("launches hand-written CDNA4 HIP kernels through a thin
// torch-cpp-extension C-ABI"): every function here unwraps torch-ROCm tensors (device pointer,
// shape and dtype checks), takes torch's CURRENT HIP stream of the tensors' device, and calls the
// extern "C" entry point of libcrender_hip.so with plain pointers and sizes.  No kernel lives
// here and no result is computed here; the C ABI stays the drop-in boundary (the reference-side
// caller this replaces is crender/cy/renderer.py:47-49 -> AdvancedPixelBufferFiller.render_model,
// .pyx:92-104).  Built in-tree by __graft_entry__.build() with torch.utils.cpp_extension.
#include <torch/extension.h>

#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>   // (torch-ROCm tensors report device type "cuda")
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <array>
#include <string>

#include "../../include/crender_hip.h"

namespace {

void check(int status, const char *what)
{
    if (status != CRENDER_OK) {
        const char *msg = crender_last_error();
        throw std::runtime_error(std::string(what) + " failed (code " + std::to_string(status) + "): " +
                                 (msg ? msg : ""));
    }
}

const float *tri_ptr(const at::Tensor &t, const char *name, int64_t T)
{
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous(), name,
                ": expected a contiguous float32 device tensor");
    TORCH_CHECK(t.dim() == 3 && t.size(0) == T && t.size(1) == 3 && t.size(2) == 3, name,
                ": expected shape [T, 3, 3]");
    return t.data_ptr<float>();
}

float *plane_ptr(const at::Tensor &t, const char *name, const at::Device &dev)
{
    TORCH_CHECK(t.is_cuda() && t.device() == dev && t.scalar_type() == at::kFloat && t.is_contiguous(), name,
                ": expected a contiguous float32 tensor on the triangles' device");
    return t.data_ptr<float>();
}

std::array<float, 16> p16(const at::Tensor &P)
{
    TORCH_CHECK(!P.is_cuda() && P.scalar_type() == at::kFloat && P.numel() == 16, "P: expected 16 float32 values on the host");
    const at::Tensor c = P.contiguous();
    std::array<float, 16> out;
    std::copy(c.data_ptr<float>(), c.data_ptr<float>() + 16, out.begin());
    return out;
}

void *current_stream(const at::Device &dev) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream(); }

// replaces: render_model (.pyx:92-104) — crender_render_model on torch's current stream.
// Returns the frame's number on the plan (crender_plan_frame_ticket): what crender_plan_poll_bin_usage
// is asked about later.
int64_t render_model(int64_t plan, const at::Tensor &tri, const at::Tensor &col, const at::Tensor &nrm,
                  const at::Tensor &P, at::Tensor z, at::Tensor color, at::Tensor normal,
                  const c10::optional<at::Tensor> &winner, int64_t flags)
{
    const int64_t T = tri.dim() == 3 ? tri.size(0) : -1;
    const at::Device dev = z.device();
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    int32_t *win = nullptr;
    if (winner.has_value()) {
        TORCH_CHECK(winner->is_cuda() && winner->scalar_type() == at::kInt && winner->is_contiguous(),
                    "winner: expected a contiguous int32 device tensor");
        win = winner->data_ptr<int32_t>();
    }
    const auto Pm = p16(P);
    check(crender_render_model(reinterpret_cast<crender_plan *>(plan), T ? tri_ptr(tri, "tri", T) : nullptr,
                               T ? tri_ptr(col, "col", T) : nullptr, T ? tri_ptr(nrm, "nrm", T) : nullptr, T,
                               Pm.data(), plane_ptr(z, "z", dev), plane_ptr(color, "color", dev),
                               plane_ptr(normal, "normal", dev), win, (unsigned)flags, current_stream(dev)),
          "crender_render_model");
    return (int64_t)crender_plan_frame_ticket(reinterpret_cast<crender_plan *>(plan));
}

// crender_pipeline_bind with tensors
void pipeline_bind(int64_t pipeline, int64_t slot, const at::Tensor &tri, const at::Tensor &col,
                   const at::Tensor &nrm, const at::Tensor &P, at::Tensor z, at::Tensor color,
                   at::Tensor normal, const c10::optional<at::Tensor> &winner, int64_t flags)
{
    const int64_t T = tri.dim() == 3 ? tri.size(0) : -1;
    const at::Device dev = z.device();
    int32_t *win = nullptr;
    if (winner.has_value()) {
        TORCH_CHECK(winner->is_cuda() && winner->scalar_type() == at::kInt && winner->is_contiguous(),
                    "winner: expected a contiguous int32 device tensor");
        win = winner->data_ptr<int32_t>();
    }
    const auto Pm = p16(P);
    check(crender_pipeline_bind(reinterpret_cast<crender_pipeline *>(pipeline), (int)slot, tri_ptr(tri, "tri", T),
                                tri_ptr(col, "col", T), tri_ptr(nrm, "nrm", T), T, Pm.data(),
                                plane_ptr(z, "z", dev), plane_ptr(color, "color", dev),
                                plane_ptr(normal, "normal", dev), win, (unsigned)flags),
          "crender_pipeline_bind");
}

// The per-frame call of the swap chain: the next bound frame on torch's current stream of
// `device_index` (which must be the current device: the library launches there).
void pipeline_submit(int64_t pipeline, int64_t device_index)
{
    check(crender_pipeline_submit(reinterpret_cast<crender_pipeline *>(pipeline),
                                  c10::hip::getCurrentHIPStreamMasqueradingAsCUDA((c10::DeviceIndex)device_index).stream()),
          "crender_pipeline_submit");
}

void pipeline_join(int64_t pipeline, int64_t device_index)
{
    check(crender_pipeline_join(reinterpret_cast<crender_pipeline *>(pipeline),
                                c10::hip::getCurrentHIPStreamMasqueradingAsCUDA((c10::DeviceIndex)device_index).stream()),
          "crender_pipeline_join");
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "torch-tensor front end of libcrender_hip.so (include/crender_hip.h)";
    m.def("abi_version", []() { return crender_abi_version(); });
    m.def("render_model", &render_model, "crender_render_model on torch's current stream");
    m.def("pipeline_bind", &pipeline_bind, "crender_pipeline_bind");
    m.def("pipeline_submit", &pipeline_submit, "crender_pipeline_submit on torch's current stream");
    m.def("pipeline_join", &pipeline_join, "crender_pipeline_join");
}
