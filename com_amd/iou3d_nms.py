"""Rotated BEV IoU / 3D IoU / NMS with the function names and signatures of the reference's
pcdet/ops/iou3d_nms/iou3d_nms_utils.py:31-116, over the HIP kernels of com_amd/csrc/iou3d.hip.

Boxes are (N, 7) float32 device tensors (x, y, z, dx, dy, dz, heading).  `boxes_bev_iou_cpu` is the reference's
host-side variant for the GT-sampling augmentor (iou3d_nms_utils.py:12-28; COMAug calls it per frame,
database_sampler_v2.py:600-601): numpy arrays / CPU tensors in, the library's host entry point
pcd_boxes_iou_bev_host underneath (no GPU call: usable inside forked DataLoader workers)."""
import ctypes

import numpy as np
import torch

from . import _lib as L


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    """iou3d_nms_utils.py:12-28: (N, 7), (M, 7) CPU boxes -> (N, M) BEV IoU; numpy in -> numpy out, tensor in -> tensor out
    (the reference's check_numpy_to_torch convention).  Bit-identical to the reference's iou3d_cpu.cpp (fixture G13)."""
    is_numpy = isinstance(boxes_a, np.ndarray)
    a = np.ascontiguousarray(boxes_a if isinstance(boxes_a, np.ndarray) else _cpu_array(boxes_a), dtype=np.float32)
    b = np.ascontiguousarray(boxes_b if isinstance(boxes_b, np.ndarray) else _cpu_array(boxes_b), dtype=np.float32)
    assert a.ndim == 2 and b.ndim == 2 and a.shape[1] == 7 and b.shape[1] == 7
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    L.check(L.lib().pcd_boxes_iou_bev_host(a.ctypes.data_as(ctypes.c_void_p), a.shape[0],
                                           b.ctypes.data_as(ctypes.c_void_p), b.shape[0],
                                           out.ctypes.data_as(ctypes.c_void_p)), "pcd_boxes_iou_bev_host")
    return out if is_numpy else torch.from_numpy(out)


def _cpu_array(t):
    assert not t.is_cuda, 'Only support CPU tensors'          # (the reference's own assertion, iou3d_nms_utils.py:22)
    return t.detach().float().numpy()


def _boxes(t):
    if not t.is_cuda:
        raise L.PcdError("iou3d_nms ops need HIP device tensors (there is no CPU fallback)")
    assert t.dim() == 2 and t.shape[1] == 7
    return t.contiguous().float()


def _pairwise(boxes_a, boxes_b, want_iou):
    a, b = _boxes(boxes_a), _boxes(boxes_b)
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    L.check(L.lib().pcd_boxes_overlap_bev(L.ptr(a), a.shape[0], L.ptr(b), b.shape[0], L.ptr(out), int(want_iou),
                                          L.stream_ptr()), "pcd_boxes_overlap_bev")
    return out


def boxes_overlap_bev(boxes_a, boxes_b):
    """(N, M) BEV intersection areas (iou3d_nms_cuda.boxes_overlap_bev_gpu)."""
    return _pairwise(boxes_a, boxes_b, False)


def boxes_iou_bev(boxes_a, boxes_b):
    """iou3d_nms_utils.py:31-46."""
    return _pairwise(boxes_a, boxes_b, True)


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """iou3d_nms_utils.py:49-82: BEV overlap x height overlap / union volume."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a_max = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    a_min = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    b_max = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1)
    b_min = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)
    overlaps_bev = boxes_overlap_bev(boxes_a, boxes_b)
    overlaps_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def nms_sorted(boxes_sorted, thresh, normal=False):
    """Greedy NMS over boxes already sorted by descending score.  Returns (keep int64 [n] on the device, num_keep
    int32 [1] on the device): nothing is read back, so the call can sit inside a captured graph."""
    b = _boxes(boxes_sorted)
    n = b.shape[0]
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=b.device)
    num = torch.zeros((1,), dtype=torch.int32, device=b.device)
    lib = L.lib()
    ws = torch.empty((max(int(lib.pcd_nms_workspace_bytes(n)), 256),), dtype=torch.uint8, device=b.device)
    L.check(lib.pcd_nms_bev(L.ptr(b), n, float(thresh), int(bool(normal)), L.ptr(keep), L.ptr(num), L.ptr(ws),
                            ws.numel(), L.stream_ptr()), "pcd_nms_bev")
    return keep, num


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """iou3d_nms_utils.py:85-100 (the one host read-back is the number of survivors, as in the reference)."""
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    keep, num = nms_sorted(boxes[order], thresh, normal=False)
    return order[keep[:int(num.item())]].contiguous(), None


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """iou3d_nms_utils.py:103-116."""
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    keep, num = nms_sorted(boxes[order], thresh, normal=True)
    return order[keep[:int(num.item())]].contiguous(), None
