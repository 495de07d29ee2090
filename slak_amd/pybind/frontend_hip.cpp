// slak_amd/pybind/frontend_hip.cpp -- the reference's pybind module `_depthwise_conv2d_implicit_gemm_C`
// (cutlass/examples/19_large_depthwise_conv2d_torch_extension/frontend.cpp:3-16, frontend.h:3-10) on top of libslak_hip.so.
// Same six exported names and signatures (+ the bf16 trio), so the reference's own depthwise_conv2d_implicit_gemm.py
// (`import _depthwise_conv2d_implicit_gemm_C as _extension`, :8) runs on it unmodified.  Host-only C++ (no device code here):
// tensors in, the C ABI of include/slak_hip.h underneath, outputs allocated through the torch allocator like the reference
// (torch::empty_like, forward_fp32.cu:206; fp32 dw, backward_filter_fp16.cu:187), the CURRENT stream instead of the null
// stream (convolution.h:243), and an exception instead of exit(EXIT_FAILURE) (forward_fp32.cu:173-192).
#include <torch/extension.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>

#include <algorithm>

#include "../../include/slak_hip.h"

namespace {

int dt(const torch::Tensor& t, const char* what) {
    switch (t.scalar_type()) {
        case torch::kFloat: return SLAK_F32;
        case torch::kHalf: return SLAK_F16;
        case torch::kBFloat16: return SLAK_BF16;
        default: TORCH_CHECK(false, what, ": only float32, float16 and bfloat16 are supported, got ", t.scalar_type());
    }
    return -1;
}

void check_tensor(const torch::Tensor& t, const char* what) {       // forward_fp32.cu:194-196, :203-204
    TORCH_CHECK(t.is_cuda(), what, " must be a CUDA/HIP tensor");
    TORCH_CHECK(t.is_contiguous(), what, " must be contiguous");
}

struct Dims { int N, C, H, W, kh, kw; };
Dims dims(const torch::Tensor& x, const torch::Tensor& w) {
    TORCH_CHECK(x.dim() == 4 && w.dim() == 4 && w.size(1) == 1 && w.size(0) == x.size(1),
                "expected x (N,C,H,W) and depthwise weight (C,1,kh,kw), got ", x.sizes(), " and ", w.sizes());
    return {(int)x.size(0), (int)x.size(1), (int)x.size(2), (int)x.size(3), (int)w.size(2), (int)w.size(3)};
}

torch::Tensor scratch(int op, const Dims& d, const torch::Tensor& like, int dtype) {
    const size_t n = slak_dwconv2d_workspace_bytes(op, d.N, d.C, d.H, d.W, d.kh, d.kw, dtype);
    return torch::empty({(int64_t)std::max<size_t>(n, 16)}, like.options().dtype(torch::kUInt8));
}

void* stream_of(const torch::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

void check_rc(int rc, const char* fn) {
    TORCH_CHECK(rc == SLAK_OK, fn, ": ", slak_status_string(rc), " (", slak_last_hip_error(), ")");
}

torch::Tensor forward(torch::Tensor x, torch::Tensor w) {
    check_tensor(x, "input"); check_tensor(w, "weight");
    const Dims d = dims(x, w);
    c10::hip::HIPGuard guard(x.get_device());
    auto y = torch::empty_like(x);
    auto ws = scratch(0, d, x, dt(x, "input"));
    check_rc(slak_dwconv2d_forward(x.data_ptr(), dt(x, "input"), w.data_ptr(), dt(w, "weight"), y.data_ptr(), dt(y, "output"),
                                   d.N, d.C, d.H, d.W, d.kh, d.kw, ws.data_ptr(), (size_t)ws.numel(), stream_of(x)),
             "slak_dwconv2d_forward");
    return y;
}

torch::Tensor backward_data(torch::Tensor dy, torch::Tensor w) {
    check_tensor(dy, "grad"); check_tensor(w, "weight");
    const Dims d = dims(dy, w);
    c10::hip::HIPGuard guard(dy.get_device());
    auto dx = torch::empty_like(dy);
    auto ws = scratch(1, d, dy, dt(dy, "grad"));
    check_rc(slak_dwconv2d_backward_data(dy.data_ptr(), dt(dy, "grad"), w.data_ptr(), dt(w, "weight"), dx.data_ptr(), dt(dx, "dx"),
                                         d.N, d.C, d.H, d.W, d.kh, d.kw, ws.data_ptr(), (size_t)ws.numel(), stream_of(dy)),
             "slak_dwconv2d_backward_data");
    return dx;
}

torch::Tensor backward_filter(torch::Tensor dy, torch::Tensor x, torch::Tensor w) {
    check_tensor(dy, "grad"); check_tensor(x, "input"); check_tensor(w, "weight");
    const Dims d = dims(x, w);
    TORCH_CHECK(dy.sizes() == x.sizes() && dy.scalar_type() == x.scalar_type(), "grad and input must have the same shape and dtype");
    c10::hip::HIPGuard guard(x.get_device());
    auto dw = torch::empty({d.C, 1, d.kh, d.kw}, x.options().dtype(torch::kFloat));     // always fp32: backward_filter_fp16.cu:187
    auto ws = scratch(2, d, x, dt(x, "input"));
    check_rc(slak_dwconv2d_backward_filter(dy.data_ptr(), dt(dy, "grad"), x.data_ptr(), dt(x, "input"), (float*)dw.data_ptr(),
                                           d.N, d.C, d.H, d.W, d.kh, d.kw, ws.data_ptr(), (size_t)ws.numel(), stream_of(x)),
             "slak_dwconv2d_backward_filter");
    return dw;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    // frontend.cpp:3-16 -- the dtype suffix is kept for the caller's sake; the entry points dispatch on the tensors' dtypes
    m.def("forward_fp32", &forward, "forward_fp32");
    m.def("backward_data_fp32", &backward_data, "backward_data_fp32");
    m.def("backward_filter_fp32", &backward_filter, "backward_filter_fp32");
    m.def("forward_fp16", &forward, "forward_fp16");
    m.def("backward_data_fp16", &backward_data, "backward_data_fp16");
    m.def("backward_filter_fp16", &backward_filter, "backward_filter_fp16");
    m.def("forward_bf16", &forward, "forward_bf16");
    m.def("backward_data_bf16", &backward_data, "backward_data_bf16");
    m.def("backward_filter_bf16", &backward_filter, "backward_filter_bf16");
}
