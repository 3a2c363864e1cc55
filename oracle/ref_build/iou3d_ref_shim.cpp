// Test infrastructure (NOT product code): a C entry point around the REFERENCE's own host implementation of the
// rotated BEV IoU, `boxes_iou_bev_cpu` (pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:232-252), which oracle/ref_build/Makefile
// compiles UNMODIFIED from /root/reference into oracle/_ref/libiou3d_ref.so.  Nothing of the reference is copied: this
// file only declares the reference's function (its header, iou3d_cpu.h:9) and wraps caller memory in tensors.
#include <torch/extension.h>

int boxes_iou_bev_cpu(at::Tensor boxes_a_tensor, at::Tensor boxes_b_tensor, at::Tensor ans_iou_tensor);

extern "C" int ref_boxes_iou_bev_cpu(const float *boxes_a, int n, const float *boxes_b, int m, float *iou) {
    auto opt = torch::TensorOptions().dtype(torch::kFloat32);
    at::Tensor a = torch::from_blob(const_cast<float *>(boxes_a), {n, 7}, opt);
    at::Tensor b = torch::from_blob(const_cast<float *>(boxes_b), {m, 7}, opt);
    at::Tensor o = torch::from_blob(iou, {n, m}, opt);
    return boxes_iou_bev_cpu(a, b, o);
}
