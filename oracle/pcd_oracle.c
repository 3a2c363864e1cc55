/*
 * pcd_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (com_amd/) never links, imports or calls it.
 *
 * PARITY STATUS
 *   - hard voxelization, SubM / strided rulebooks, sparse-conv arithmetic follow the
 *     *published* algorithm of the third-party `spconv` package (traveller59/spconv, unpinned
 *     by the reference: docker/Dockerfile:55 `pip install spconv-cu102`, docs/INSTALL.md:9,30-33),
 *     whose source is NOT under /root/reference and cannot be imported here.  The reference has
 *     no tests or golden vectors for it  ==> "parity unpinned" for these
 *     functions.  They are cross-pinned instead against torch.nn.functional.conv3d on densified
 *     grids and against the reference's own in-repo torch code (DynamicMeanVFE index math), see
 *     tests/golden/make_golden.py.
 *   - dynamic voxelization + mean follows pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:53-72
 *     and IS pinned against that file run in the build container (fixture G2).
 *
 * Everything is plain C99, float32 arithmetic with contraction disabled (build with
 * -ffp-contract=off) so that floor((p - min) / vsize) is evaluated exactly as IEEE f32
 * sub + div, the way numpy / torch / the spconv C++ voxel generator evaluate it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* tiny open-addressing map  int64 key -> int32 value                                           */
typedef struct {
    int64_t *keys;
    int32_t *vals;
    uint64_t mask;
} orc_map;

static int orc_map_init(orc_map *m, int64_t n) {
    uint64_t cap = 16;
    while (cap < (uint64_t)(2 * n + 2)) cap <<= 1;
    m->keys = (int64_t *)malloc(cap * sizeof(int64_t));
    m->vals = (int32_t *)malloc(cap * sizeof(int32_t));
    if (!m->keys || !m->vals) return -1;
    for (uint64_t i = 0; i < cap; ++i) m->keys[i] = -1;
    m->mask = cap - 1;
    return 0;
}
static void orc_map_free(orc_map *m) {
    free(m->keys);
    free(m->vals);
}
static inline uint64_t orc_hash(int64_t k) {
    uint64_t x = (uint64_t)k;
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    return x;
}
/* returns pointer to the value slot; *found tells whether key already existed */
static inline int32_t *orc_map_slot(orc_map *m, int64_t key, int *found) {
    uint64_t h = orc_hash(key) & m->mask;
    for (;;) {
        if (m->keys[h] == key) {
            *found = 1;
            return &m->vals[h];
        }
        if (m->keys[h] == -1) {
            m->keys[h] = key;
            *found = 0;
            return &m->vals[h];
        }
        h = (h + 1) & m->mask;
    }
}
static inline int32_t orc_map_get(const orc_map *m, int64_t key) {
    uint64_t h = orc_hash(key) & m->mask;
    for (;;) {
        if (m->keys[h] == key) return m->vals[h];
        if (m->keys[h] == -1) return -1;
        h = (h + 1) & m->mask;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* grid size = round((max - min) / vsize)    pcdet/datasets/processor/data_processor.py:127-128 */
void orc_grid_size(const float range[6], const float vsize[3], int grid[3]) {
    for (int j = 0; j < 3; ++j) {
        double g = ((double)range[j + 3] - (double)range[j]) / (double)vsize[j];
        grid[j] = (int)llround(g);
    }
}

/* ------------------------------------------------------------------------------------------ */
/*
 * Hard voxelization.  Called by the reference at pcdet/datasets/processor/data_processor.py:44-60
 * (VoxelGeneratorWrapper.generate -> spconv.utils.VoxelGeneratorV2.generate /
 * Point2VoxelCPU3d.point_to_voxel).  Algorithm = SURVEY.md Appendix A.1 (spconv upstream):
 * point order; coordinate (z,y,x) = floor((p - min)/vsize) in f32; points outside [0,grid) are
 * skipped; voxel ids in first-appearance order; once max_voxels voxels exist further NEW voxels
 * are skipped; the first T points of a voxel are kept, zero padded.
 *
 * points [n, c] f32 (x,y,z,...), outputs sized for max_voxels.  Returns M.
 */
int orc_voxelize_hard(const float *points, int n, int c, const float range[6],
                      const float vsize[3], const int grid[3], int T, int max_voxels,
                      float *voxels, int32_t *coords, int32_t *num_points) {
    orc_map map;
    if (orc_map_init(&map, n) != 0) return -1;
    memset(voxels, 0, (size_t)max_voxels * T * c * sizeof(float));
    memset(num_points, 0, (size_t)max_voxels * sizeof(int32_t));
    int voxel_num = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = points + (size_t)i * c;
        int cc[3];
        int ok = 1;
        for (int j = 0; j < 3; ++j) {
            float d = p[j] - range[j];
            float q = d / vsize[j];
            float f = floorf(q);
            /* int conversion as C++ does it; NaN / huge values fail the range test */
            if (!(f >= 0.0f) || !(f < (float)grid[j])) {
                ok = 0;
                break;
            }
            cc[j] = (int)f;
        }
        if (!ok) continue;
        int64_t key = ((int64_t)cc[2] * grid[1] + cc[1]) * grid[0] + cc[0];
        /* emulate coor_to_voxelidx[z][y][x] == -1 test without registering dropped voxels */
        int32_t v = orc_map_get(&map, key);
        if (v == -1) {
            if (voxel_num >= max_voxels) continue;
            int found;
            int32_t *slot = orc_map_slot(&map, key, &found);
            v = voxel_num++;
            *slot = v;
            coords[v * 3 + 0] = cc[2];
            coords[v * 3 + 1] = cc[1];
            coords[v * 3 + 2] = cc[0];
        }
        int np = num_points[v];
        if (np < T) {
            memcpy(voxels + ((size_t)v * T + np) * c, p, (size_t)c * sizeof(float));
            num_points[v] = np + 1;
        }
    }
    orc_map_free(&map);
    return voxel_num;
}

/* ------------------------------------------------------------------------------------------ */
/*
 * Dynamic voxelization + mean.  pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:53-72.
 * points [n, 1+c] f32 = (batch_idx, x, y, z, ...);  key = b*XYZ + cx*YZ + cy*Z + cz (:57-60; the
 * reference uses int32 there, we use int64 and note that it overflows for b >= 23 on the Waymo
 * grid); unique ascending (:63); mean of the c features (:65); coords (b, z, y, x) (:68-72).
 * out_feat [n, c], out_coords [n, 4] worst case.  Returns number of voxels.
 */
typedef struct {
    int64_t key;
    int32_t idx;
} orc_kv;
static int orc_kv_cmp(const void *a, const void *b) {
    const orc_kv *x = (const orc_kv *)a, *y = (const orc_kv *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
int orc_voxelize_dynamic_mean(const float *points, int n, int c, const float range[6],
                              const float vsize[3], const int grid[3], float *out_feat,
                              int32_t *out_coords, int32_t *out_count) {
    orc_kv *kv = (orc_kv *)malloc((size_t)(n > 0 ? n : 1) * sizeof(orc_kv));
    if (!kv) return -1;
    int m = 0;
    const int64_t sxyz = (int64_t)grid[0] * grid[1] * grid[2];
    const int64_t syz = (int64_t)grid[1] * grid[2];
    const int64_t sz = grid[2];
    for (int i = 0; i < n; ++i) {
        const float *p = points + (size_t)i * (c + 1);
        int cc[3];
        int ok = 1;
        for (int j = 0; j < 3; ++j) {
            float f = floorf((p[1 + j] - range[j]) / vsize[j]);
            if (!(f >= 0.0f) || !(f < (float)grid[j])) {
                ok = 0;
                break;
            }
            cc[j] = (int)f;
        }
        if (!ok) continue;
        kv[m].key = (int64_t)(int)p[0] * sxyz + cc[0] * syz + cc[1] * sz + cc[2];
        kv[m].idx = i;
        ++m;
    }
    qsort(kv, (size_t)m, sizeof(orc_kv), orc_kv_cmp);
    int nv = 0;
    int i = 0;
    while (i < m) {
        int j = i;
        float *f = out_feat + (size_t)nv * c;
        for (int q = 0; q < c; ++q) f[q] = 0.0f;
        while (j < m && kv[j].key == kv[i].key) {
            const float *p = points + (size_t)kv[j].idx * (c + 1) + 1;
            for (int q = 0; q < c; ++q) f[q] += p[q];
            ++j;
        }
        float cnt = (float)(j - i);
        for (int q = 0; q < c; ++q) f[q] = f[q] / cnt;
        int64_t key = kv[i].key;
        out_coords[nv * 4 + 0] = (int32_t)(key / sxyz);
        int32_t cx = (int32_t)((key % sxyz) / syz);
        int32_t cy = (int32_t)((key % syz) / sz);
        int32_t cz = (int32_t)(key % sz);
        out_coords[nv * 4 + 1] = cz;
        out_coords[nv * 4 + 2] = cy;
        out_coords[nv * 4 + 3] = cx;
        if (out_count) out_count[nv] = j - i;
        ++nv;
        i = j;
    }
    free(kv);
    return nv;
}

/* ------------------------------------------------------------------------------------------ */
/*
 * Rulebooks.  Semantics = spconv, call sites
 * pcdet/models/backbones_3d/spconv_backbone.py:12 (SubMConv3d), :14-15 (SparseConv3d).
 * Kernel offsets k = (kd*KH + kh)*KW + kw; in_pos = out_pos*stride - pad + k_idx*dil.
 * indices [n,4] i32 (b,z,y,x).  Canonical form: pairs within each k ascending in input row;
 * strided-conv output rows sorted by linear key ((b*D+z)*H+y)*W+x.
 *
 * Two views of the same rulebook are emitted:
 *   pairs   [K][2][pmax]  (pmax = n_in): pairs[k][0][p] = in row, pairs[k][1][p] = out row, -1 padded
 *   pair_num[K]
 *   nbr_out [K][n_out]: input row feeding output row o through offset k, or -1
 *   nbr_in  [K][n_in] : output row fed by input row i through offset k, or -1
 */
static inline int64_t orc_lin(int b, int z, int y, int x, const int shape[3]) {
    return (((int64_t)b * shape[0] + z) * shape[1] + y) * shape[2] + x;
}

/* Threads of the per-offset loops of the two rulebook builders (bench.py's multi-core CPU baseline; 1 = sequential, the
 * default).  The offsets are independent once the coordinate map is built: every k writes its own rows of pairs / nbr_out /
 * nbr_in / pair_num, so the tables are identical for any thread count. */
static int orc_rb_threads = 1;
void orc_set_rulebook_threads(int t) { orc_rb_threads = t > 1 ? t : 1; }

int orc_rulebook_subm(const int32_t *indices, int n, const int shape[3], const int ksize[3],
                      const int dil[3], int32_t *pairs, int32_t *pair_num, int32_t *nbr_out,
                      int32_t *nbr_in) {
    const int K = ksize[0] * ksize[1] * ksize[2];
    orc_map map;
    if (orc_map_init(&map, n) != 0) return -1;
    for (int i = 0; i < n; ++i) {
        const int32_t *c = indices + (size_t)i * 4;
        int found;
        int32_t *slot = orc_map_slot(&map, orc_lin(c[0], c[1], c[2], c[3], shape), &found);
        if (found) {
            orc_map_free(&map);
            return -2; /* duplicate coordinate */
        }
        *slot = i;
    }
    for (size_t q = 0; q < (size_t)K * 2 * n; ++q) pairs[q] = -1;
    for (size_t q = 0; q < (size_t)K * n; ++q) nbr_out[q] = -1;
    if (nbr_in)
        for (size_t q = 0; q < (size_t)K * n; ++q) nbr_in[q] = -1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(orc_rb_threads) if (orc_rb_threads > 1)
    for (int k = 0; k < K; ++k) {
        int kd = k / (ksize[1] * ksize[2]), kh = (k / ksize[2]) % ksize[1], kw = k % ksize[2];
        int cnt = 0;
        for (int i = 0; i < n; ++i) {
            const int32_t *c = indices + (size_t)i * 4;
            /* out = in + pad - k*dil with pad = (K/2)*dil (SubM forces "same" padding) */
            int oz = c[1] + (ksize[0] / 2 - kd) * dil[0];
            int oy = c[2] + (ksize[1] / 2 - kh) * dil[1];
            int ox = c[3] + (ksize[2] / 2 - kw) * dil[2];
            if (oz < 0 || oz >= shape[0] || oy < 0 || oy >= shape[1] || ox < 0 || ox >= shape[2])
                continue;
            int32_t o = orc_map_get(&map, orc_lin(c[0], oz, oy, ox, shape));
            if (o < 0) continue;
            pairs[((size_t)k * 2 + 0) * n + cnt] = i;
            pairs[((size_t)k * 2 + 1) * n + cnt] = o;
            ++cnt;
            nbr_out[(size_t)k * n + o] = i;
            if (nbr_in) nbr_in[(size_t)k * n + i] = o;
        }
        pair_num[k] = cnt;
    }
    orc_map_free(&map);
    return n;
}

void orc_conv_out_shape(const int in_shape[3], const int ksize[3], const int stride[3],
                        const int pad[3], const int dil[3], int out_shape[3]) {
    for (int d = 0; d < 3; ++d) {
        int num = in_shape[d] + 2 * pad[d] - dil[d] * (ksize[d] - 1) - 1;
        /* floor division, num may be negative */
        int q = num >= 0 ? num / stride[d] : -((-num + stride[d] - 1) / stride[d]);
        out_shape[d] = q + 1;
    }
}

static int orc_i64_cmp(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return x < y ? -1 : (x > y);
}

/* out_cap = capacity (rows) of out_indices / nbr_out; returns n_out or <0 */
int orc_rulebook_conv(const int32_t *indices, int n, const int in_shape[3], const int ksize[3],
                      const int stride[3], const int pad[3], const int dil[3], int out_cap,
                      int32_t *out_indices, int32_t *pairs, int32_t *pair_num, int32_t *nbr_out,
                      int32_t *nbr_in) {
    const int K = ksize[0] * ksize[1] * ksize[2];
    int out_shape[3];
    orc_conv_out_shape(in_shape, ksize, stride, pad, dil, out_shape);
    int64_t *cand = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * K * sizeof(int64_t));
    int64_t *ckey = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * K * sizeof(int64_t)); /* per (i,k) */
    if (!cand || !ckey) return -1;
    size_t nc = 0;
    for (int i = 0; i < n; ++i) {
        const int32_t *c = indices + (size_t)i * 4;
        for (int k = 0; k < K; ++k) {
            int kk[3] = {k / (ksize[1] * ksize[2]), (k / ksize[2]) % ksize[1], k % ksize[2]};
            int o[3];
            int ok = 1;
            for (int d = 0; d < 3; ++d) {
                int t = c[1 + d] + pad[d] - kk[d] * dil[d];
                if (t < 0 || t % stride[d] != 0) {
                    ok = 0;
                    break;
                }
                o[d] = t / stride[d];
                if (o[d] >= out_shape[d]) {
                    ok = 0;
                    break;
                }
            }
            int64_t key = -1;
            if (ok) {
                key = orc_lin(c[0], o[0], o[1], o[2], out_shape);
                cand[nc++] = key;
            }
            ckey[(size_t)i * K + k] = key;
        }
    }
    qsort(cand, nc, sizeof(int64_t), orc_i64_cmp);
    size_t nu = 0;
    for (size_t q = 0; q < nc; ++q)
        if (q == 0 || cand[q] != cand[q - 1]) cand[nu++] = cand[q];
    if ((int64_t)nu > out_cap) {
        free(cand);
        free(ckey);
        return -3;
    }
    orc_map map;
    if (orc_map_init(&map, (int64_t)nu) != 0) return -1;
    const int64_t vol = (int64_t)out_shape[0] * out_shape[1] * out_shape[2];
    for (size_t r = 0; r < nu; ++r) {
        int found;
        *orc_map_slot(&map, cand[r], &found) = (int32_t)r;
        int64_t key = cand[r];
        int64_t rem = key % vol;
        out_indices[r * 4 + 0] = (int32_t)(key / vol);
        out_indices[r * 4 + 1] = (int32_t)(rem / ((int64_t)out_shape[1] * out_shape[2]));
        out_indices[r * 4 + 2] = (int32_t)((rem / out_shape[2]) % out_shape[1]);
        out_indices[r * 4 + 3] = (int32_t)(rem % out_shape[2]);
    }
    for (size_t q = 0; q < (size_t)K * 2 * n; ++q) pairs[q] = -1;
    for (size_t q = 0; q < (size_t)K * out_cap; ++q) nbr_out[q] = -1;
    if (nbr_in)
        for (size_t q = 0; q < (size_t)K * n; ++q) nbr_in[q] = -1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(orc_rb_threads) if (orc_rb_threads > 1)
    for (int k = 0; k < K; ++k) {
        int cnt = 0;
        for (int i = 0; i < n; ++i) {
            int64_t key = ckey[(size_t)i * K + k];
            if (key < 0) continue;
            int32_t o = orc_map_get(&map, key);
            pairs[((size_t)k * 2 + 0) * n + cnt] = i;
            pairs[((size_t)k * 2 + 1) * n + cnt] = o;
            ++cnt;
            nbr_out[(size_t)k * out_cap + o] = i;
            if (nbr_in) nbr_in[(size_t)k * n + i] = o;
        }
        pair_num[k] = cnt;
    }
    orc_map_free(&map);
    free(cand);
    free(ckey);
    return (int)nu;
}

/* ------------------------------------------------------------------------------------------ */
/*
 * Sparse convolution arithmetic, spconv "native" algorithm:
 *   Y[o] = bias + sum_k sum_{(i,o) in pairs[k]} X[i] @ W[k],     W [K][cin][cout] f32
 * gather -> GEMM -> scatter-add per kernel offset.  fp32 throughout.
 * Also the CPU baseline ("port") timed by bench.py.
 */
void orc_conv_fwd(const float *X, int cin, const float *W, const float *bias, const int32_t *pairs,
                  const int32_t *pair_num, int K, int pmax, float *Y, int n_out, int cout) {
    for (int o = 0; o < n_out; ++o)
        for (int c = 0; c < cout; ++c) Y[(size_t)o * cout + c] = bias ? bias[c] : 0.0f;
    for (int k = 0; k < K; ++k) {
        const int32_t *pin = pairs + ((size_t)k * 2 + 0) * pmax;
        const int32_t *pout = pairs + ((size_t)k * 2 + 1) * pmax;
        const float *Wk = W + (size_t)k * cin * cout;
        for (int p = 0; p < pair_num[k]; ++p) {
            const float *x = X + (size_t)pin[p] * cin;
            float *y = Y + (size_t)pout[p] * cout;
            for (int ci = 0; ci < cin; ++ci) {
                const float xv = x[ci];
                const float *w = Wk + (size_t)ci * cout;
                for (int co = 0; co < cout; ++co) y[co] += xv * w[co];
            }
        }
    }
}

/* dX[i] += dY[o] @ W[k]^T ; dW[k] += X[i]^T (x) dY[o] ; dbias = sum_o dY[o]   (A.5) */
void orc_conv_bwd(const float *X, int n_in, int cin, const float *W, const float *dY, int n_out,
                  int cout, const int32_t *pairs, const int32_t *pair_num, int K, int pmax,
                  float *dX, float *dW, float *dbias) {
    memset(dX, 0, (size_t)n_in * cin * sizeof(float));
    memset(dW, 0, (size_t)K * cin * cout * sizeof(float));
    if (dbias) {
        for (int c = 0; c < cout; ++c) dbias[c] = 0.0f;
        for (int o = 0; o < n_out; ++o)
            for (int c = 0; c < cout; ++c) dbias[c] += dY[(size_t)o * cout + c];
    }
    for (int k = 0; k < K; ++k) {
        const int32_t *pin = pairs + ((size_t)k * 2 + 0) * pmax;
        const int32_t *pout = pairs + ((size_t)k * 2 + 1) * pmax;
        const float *Wk = W + (size_t)k * cin * cout;
        float *dWk = dW + (size_t)k * cin * cout;
        for (int p = 0; p < pair_num[k]; ++p) {
            const float *x = X + (size_t)pin[p] * cin;
            float *dx = dX + (size_t)pin[p] * cin;
            const float *dy = dY + (size_t)pout[p] * cout;
            for (int ci = 0; ci < cin; ++ci) {
                const float *w = Wk + (size_t)ci * cout;
                float *dw = dWk + (size_t)ci * cout;
                const float xv = x[ci];
                float acc = 0.0f;
                for (int co = 0; co < cout; ++co) {
                    acc += dy[co] * w[co];
                    dw[co] += xv * dy[co];
                }
                dx[ci] += acc;
            }
        }
    }
}

/*
 * Multi-threaded variants (OpenMP), used ONLY by bench.py's cpu_baseline leg so that the baseline can use the host's
 * cores: the same gather-GEMM-scatter arithmetic, parallel over the pairs of one kernel offset (inside one offset
 * every output row and every input row occurs at most once, so Y / dX rows are written by one thread); dW is
 * accumulated in per-thread buffers and added up in thread order.
 */
#ifdef _OPENMP
#include <omp.h>
#endif
void orc_conv_fwd_mt(const float *X, int cin, const float *W, const float *bias, const int32_t *pairs,
                     const int32_t *pair_num, int K, int pmax, float *Y, int n_out, int cout, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(static)
    for (int o = 0; o < n_out; ++o)
        for (int c = 0; c < cout; ++c) Y[(size_t)o * cout + c] = bias ? bias[c] : 0.0f;
    for (int k = 0; k < K; ++k) {
        const int32_t *pin = pairs + ((size_t)k * 2 + 0) * pmax;
        const int32_t *pout = pairs + ((size_t)k * 2 + 1) * pmax;
        const float *Wk = W + (size_t)k * cin * cout;
        const int P = pair_num[k];
#pragma omp parallel for schedule(static)
        for (int p = 0; p < P; ++p) {
            const float *x = X + (size_t)pin[p] * cin;
            float *y = Y + (size_t)pout[p] * cout;
            for (int ci = 0; ci < cin; ++ci) {
                const float xv = x[ci];
                const float *w = Wk + (size_t)ci * cout;
                for (int co = 0; co < cout; ++co) y[co] += xv * w[co];
            }
        }
    }
}

void orc_conv_bwd_mt(const float *X, int n_in, int cin, const float *W, const float *dY, int n_out, int cout,
                     const int32_t *pairs, const int32_t *pair_num, int K, int pmax, float *dX, float *dW,
                     int threads) {
    (void)n_out;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    const int T = omp_get_max_threads();
#else
    const int T = 1;
#endif
    memset(dX, 0, (size_t)n_in * cin * sizeof(float));
    memset(dW, 0, (size_t)K * cin * cout * sizeof(float));
    float *priv = (float *)calloc((size_t)T * cin * cout, sizeof(float));
    if (!priv) return;
    for (int k = 0; k < K; ++k) {
        const int32_t *pin = pairs + ((size_t)k * 2 + 0) * pmax;
        const int32_t *pout = pairs + ((size_t)k * 2 + 1) * pmax;
        const float *Wk = W + (size_t)k * cin * cout;
        const int P = pair_num[k];
        memset(priv, 0, (size_t)T * cin * cout * sizeof(float));
#pragma omp parallel
        {
#ifdef _OPENMP
            float *dWt = priv + (size_t)omp_get_thread_num() * cin * cout;
#else
            float *dWt = priv;
#endif
#pragma omp for schedule(static)
            for (int p = 0; p < P; ++p) {
                const float *x = X + (size_t)pin[p] * cin;
                float *dx = dX + (size_t)pin[p] * cin;
                const float *dy = dY + (size_t)pout[p] * cout;
                for (int ci = 0; ci < cin; ++ci) {
                    const float *w = Wk + (size_t)ci * cout;
                    float *dw = dWt + (size_t)ci * cout;
                    const float xv = x[ci];
                    float acc = 0.0f;
                    for (int co = 0; co < cout; ++co) {
                        acc += dy[co] * w[co];
                        dw[co] += xv * dy[co];
                    }
                    dx[ci] += acc;
                }
            }
        }
        float *dWk = dW + (size_t)k * cin * cout;
        for (int t = 0; t < T; ++t)
            for (int e = 0; e < cin * cout; ++e) dWk[e] += priv[(size_t)t * cin * cout + e];
    }
    free(priv);
}

/* ------------------------------------------------------------------------------------------ */
/*
 * SparseConvTensor.dense() + HeightCompression view:
 * pcdet/models/backbones_2d/map_to_bev/height_compression.py:20-25 -> out [B, C*D, H, W] with
 * channel index c*D + z.  PointPillarScatter (pointpillar_scatter.py:17-37) is the D == 1 case.
 */
void orc_dense_bev(const float *feat, const int32_t *indices, int n, int c, int B, int D, int H,
                   int W, float *out) {
    memset(out, 0, (size_t)B * c * D * H * W * sizeof(float));
    for (int r = 0; r < n; ++r) {
        const int32_t *id = indices + (size_t)r * 4;
        for (int q = 0; q < c; ++q)
            out[((((size_t)id[0] * c + q) * D + id[1]) * H + id[2]) * W + id[3]] =
                feat[(size_t)r * c + q];
    }
}

/* ------------------------------------------------------------------------------------------ */
/*
 * Rotated BEV overlap / IoU and greedy NMS: restatement of pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:59-252
 * (= the device code iou3d_nms_kernel.cu:27-234) and of the mask + host-loop NMS
 * (iou3d_nms_kernel.cu:267-376, iou3d_nms.cpp:100-130), float32 like the reference.
 * PARITY STATUS: PINNED.  The reference file is compiled unmodified into oracle/_ref/libiou3d_ref.so
 * (oracle/ref_build/Makefile); this restatement reproduces `boxes_iou_bev_cpu` bit for bit on fixture G13 (18 000
 * random + 144 degenerate pairs, tests/test_iou3d.py) and on fresh boxes whenever oracle/_ref is present.
 */
typedef struct { float x, y; } orc_p2;

static float orc_cross3(orc_p2 p1, orc_p2 p2, orc_p2 p0) {          /* iou3d_cpu.cpp:63-65 */
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
static int orc_rect_cross(orc_p2 p1, orc_p2 p2, orc_p2 q1, orc_p2 q2) {   /* :67-73 */
    return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
           fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}
static int orc_in_box2d(const float *box, orc_p2 p) {               /* :75-85 */
    const float MARGIN = 1e-2f;
    float c = cosf(-box[6]), s = sinf(-box[6]);
    float rx = (p.x - box[0]) * c + (p.y - box[1]) * (-s);
    float ry = (p.x - box[0]) * s + (p.y - box[1]) * c;
    return fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN;
}
static int orc_intersection(orc_p2 p1, orc_p2 p0, orc_p2 q1, orc_p2 q0, orc_p2 *ans) {   /* :87-116 */
    const float EPS = 1e-8f;
    if (!orc_rect_cross(p0, p1, q0, q1)) return 0;
    float s1 = orc_cross3(q0, p1, p0), s2 = orc_cross3(p1, q1, p0);
    float s3 = orc_cross3(p0, q1, q0), s4 = orc_cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    float s5 = orc_cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > EPS) {
        ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        float D = a0 * b1 - a1 * b0;
        ans->x = (b0 * c1 - b1 * c0) / D;
        ans->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}
static void orc_corners(const float *box, orc_p2 *c) {              /* :134-162 */
    float hx = box[3] / 2, hy = box[4] / 2;
    float x1 = box[0] - hx, y1 = box[1] - hy, x2 = box[0] + hx, y2 = box[1] + hy;
    float ca = cosf(box[6]), sa = sinf(box[6]);
    float px[4] = {x1, x2, x2, x1}, py[4] = {y1, y1, y2, y2};
    for (int k = 0; k < 4; ++k) {                                    /* rotate_around_center, :118-122 */
        c[k].x = (px[k] - box[0]) * ca + (py[k] - box[1]) * (-sa) + box[0];
        c[k].y = (px[k] - box[0]) * sa + (py[k] - box[1]) * ca + box[1];
    }
    c[4] = c[0];
}
float orc_box_overlap_bev(const float *a, const float *b) {          /* :128-220 */
    orc_p2 ca[5], cb[5], pts[16], centre = {0.f, 0.f};
    int cnt = 0;
    orc_corners(a, ca);
    orc_corners(b, cb);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            orc_p2 x;
            if (orc_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], &x)) {
                centre.x += x.x;
                centre.y += x.y;
                pts[cnt++] = x;
            }
        }
    for (int k = 0; k < 4; ++k) {
        if (orc_in_box2d(a, cb[k])) { centre.x += cb[k].x; centre.y += cb[k].y; pts[cnt++] = cb[k]; }
        if (orc_in_box2d(b, ca[k])) { centre.x += ca[k].x; centre.y += ca[k].y; pts[cnt++] = ca[k]; }
    }
    centre.x /= cnt;
    centre.y /= cnt;
    for (int j = 0; j < cnt - 1; ++j)                                 /* bubble sort, point_cmp :124-126 */
        for (int i = 0; i < cnt - j - 1; ++i)
            if (atan2f(pts[i].y - centre.y, pts[i].x - centre.x) >
                atan2f(pts[i + 1].y - centre.y, pts[i + 1].x - centre.x)) {
                orc_p2 t = pts[i];
                pts[i] = pts[i + 1];
                pts[i + 1] = t;
            }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; ++k) {
        float ax = pts[k].x - pts[0].x, ay = pts[k].y - pts[0].y;
        float bx = pts[k + 1].x - pts[0].x, by = pts[k + 1].y - pts[0].y;
        area += ax * by - ay * bx;
    }
    return fabsf(area) / 2.0f;
}
float orc_iou_bev(const float *a, const float *b) {                  /* :222-229 */
    float sa = a[3] * a[4], sb = b[3] * b[4];
    float so = orc_box_overlap_bev(a, b);
    return so / fmaxf(sa + sb - so, 1e-8f);
}
static float orc_iou_normal(const float *a, const float *b) {        /* iou3d_nms_kernel.cu:312-324 */
    float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    float inter = w * h;
    return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, 1e-8f);
}
void orc_boxes_pairwise_bev(const float *boxes_a, int na, const float *boxes_b, int nb, int want_iou, float *out) {
    for (int i = 0; i < na; ++i)                                     /* iou3d_cpu.cpp:246-250 */
        for (int j = 0; j < nb; ++j)
            out[(size_t)i * nb + j] = want_iou ? orc_iou_bev(boxes_a + (size_t)i * 7, boxes_b + (size_t)j * 7)
                                               : orc_box_overlap_bev(boxes_a + (size_t)i * 7, boxes_b + (size_t)j * 7);
}
/* greedy NMS over boxes sorted by descending score: box i, if still alive, removes every j > i whose IoU with it
 * exceeds thresh -- the fixed point of the reference's bitmask + host loop.  Returns the number kept. */
int orc_nms_bev(const float *boxes, int n, float thresh, int normal, int64_t *keep) {
    unsigned char *dead = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        if (dead[i]) continue;
        keep[kept++] = i;
        for (int j = i + 1; j < n; ++j) {
            if (dead[j]) continue;
            float v = normal ? orc_iou_normal(boxes + (size_t)i * 7, boxes + (size_t)j * 7)
                             : orc_iou_bev(boxes + (size_t)i * 7, boxes + (size_t)j * 7);
            if (v > thresh) dead[j] = 1;
        }
    }
    free(dead);
    return kept;
}
