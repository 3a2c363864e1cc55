// Stride-parity classes of a strided conv's INPUT rows (spconv.hip: data / weight gradient over classes; colmap.hip:
// class-compact neighbour tables).  A row of class (rz, ry, rx) = ((z + pd) % sd, (y + ph) % sh, (x + pw) % sw) can reach
// an output only through the kernel offsets k with (r - k * dil) % s == 0 on every axis: 1..8 of 27 for k = 3, s = 2.
#pragma once
#include "common.h"

struct ClsTable {
    int ncls;
    int nk[8];
    int k[8][8];   // usable kernel offsets of class c, ascending
};

// PCD_OK, or PCD_ERR_UNSUPPORTED (more than 8 classes, or a class with more than 8 usable offsets)
static inline int make_cls_table(const int *ksize, const int *stride, const int *dil, ClsTable &T) {
    T = ClsTable{};
    T.ncls = stride[0] * stride[1] * stride[2];
    if (T.ncls > 8 || T.ncls <= 0) return PCD_ERR_UNSUPPORTED;
    for (int rz = 0; rz < stride[0]; ++rz)
        for (int ry = 0; ry < stride[1]; ++ry)
            for (int rx = 0; rx < stride[2]; ++rx) {
                const int cls = (rz * stride[1] + ry) * stride[2] + rx;
                int cnt = 0;
                for (int kz = 0; kz < ksize[0]; ++kz)
                    for (int ky = 0; ky < ksize[1]; ++ky)
                        for (int kx = 0; kx < ksize[2]; ++kx) {
                            auto ok = [](int r, int k, int d, int st_) { return (((r - k * d) % st_) + st_) % st_ == 0; };
                            if (ok(rz, kz, dil[0], stride[0]) && ok(ry, ky, dil[1], stride[1]) && ok(rx, kx, dil[2], stride[2])) {
                                if (cnt >= 8) return PCD_ERR_UNSUPPORTED;
                                T.k[cls][cnt++] = (kz * ksize[1] + ky) * ksize[2] + kx;
                            }
                        }
                T.nk[cls] = cnt;
            }
    return PCD_OK;
}
