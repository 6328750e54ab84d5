// norm.hip -- BatchNorm-on-batch-statistics / InstanceNorm fused with the following activation and the residual add.
// HBM-bound streaming kernels: 16-byte vector loads along the NHWC channel axis, per-thread fp32 partials, one LDS
// reduction per block, fp64 combination of the per-block partials in the (tiny) finalize kernels.
// See include/deepliif_hip.h (dl_norm_forward / dl_norm_backward) for semantics and reference citations.
#include "common.h"

struct NormGeom {
    int N, HW, Cp, C;
    int nchunks, ppc;          // chunks per image, pixels per chunk
};

static NormGeom make_geom(const dl_norm_desc *d) {
    NormGeom g;
    g.N = d->N; g.HW = d->H * d->W; g.Cp = d->Cp; g.C = d->C;
    int want = (1024 + d->N - 1) / d->N;
    int maxc = (g.HW + 63) / 64;
    g.nchunks = want < maxc ? want : maxc;
    if (g.nchunks < 1) g.nchunks = 1;
    g.ppc = (g.HW + g.nchunks - 1) / g.nchunks;
    g.nchunks = (g.HW + g.ppc - 1) / g.ppc;
    return g;
}

// forward with externally produced partials (dl_conv_forward's fused statistics): only the chunk count changes
static NormGeom make_geom_fwd(const dl_norm_desc *d) {
    NormGeom g = make_geom(d);
    if (d->ext_nchunks > 0) { g.nchunks = d->ext_nchunks; g.ppc = 0; }
    return g;
}

extern "C" size_t dl_norm_ws_floats(const dl_norm_desc *d) {
    NormGeom g = make_geom(d);
    if (d->ext_nchunks > g.nchunks) g.nchunks = d->ext_nchunks;
    return (size_t)g.N * g.nchunks * 2 * g.Cp + (size_t)4 * g.N * g.Cp + (size_t)2048 * g.Cp + 64;    // partials | chunk sums | c1 | c2 | dy channel-sum partials
}

// Non-temporal accesses (compile-time knob; round-6 A/B of six settings under rocprofv3, tools/gpu_r06_nt.sh, profiles/r06/norm_nt_r06.txt):
//   bit 0 = non-temporal STORES in the apply passes (z / dy are consumed by a conv that starts after the whole tensor is written),
//   bit 1 = non-temporal LOADS in the apply passes (the last read of y / dz in that direction),
//   bit 2 / bit 3 = non-temporal loads of y / dz in the partial-sum passes (faster there, 52 -> 42 us, but the apply pass that re-reads the same
//   tensors right after loses its cache hits, 65 -> 67 us: a wash).
// 3 is the measured best: norm family 82.0 -> 78.3 ms per 4 steps on one stream, whole step +1.5 %.
#ifndef DL_NORM_NT
#define DL_NORM_NT 3
#endif
template <typename T> __device__ __forceinline__ void nld_p(const T *p, float (&v)[8]) { if constexpr ((DL_NORM_NT & 4) != 0) Vec8<T>::load_nt(p, v); else Vec8<T>::load(p, v); }
template <typename T> __device__ __forceinline__ void nld_pd(const T *p, float (&v)[8]) { if constexpr ((DL_NORM_NT & 8) != 0) Vec8<T>::load_nt(p, v); else Vec8<T>::load(p, v); }
template <typename T> __device__ __forceinline__ void nld_a(const T *p, float (&v)[8]) { if constexpr ((DL_NORM_NT & 2) != 0) Vec8<T>::load_nt(p, v); else Vec8<T>::load(p, v); }
template <typename T> __device__ __forceinline__ void nst_a(T *p, const float (&v)[8]) { if constexpr ((DL_NORM_NT & 1) != 0) Vec8<T>::store_nt(p, v); else Vec8<T>::store(p, v); }

// MODE 0: forward statistics  (s1 = sum y, s2 = sum y^2)
// MODE 1: backward reductions (s1 = sum dn, s2 = sum dn * xhat), dn = dz * act'(y*scale+shift)
// ACT (the activation fused behind the norm) is a TEMPLATE parameter of the three streaming kernels: as a runtime argument its
// if / else-if ladder (with tanhf in one arm) was compiled to real branches per element -- ~100 scalar branches per loop iteration,
// which held these kernels at 4.0-4.5 TB/s while the branch-free axpby kernel streams the same tensors at 6.6 TB/s.
template <typename T, int MODE, int ACT>
__global__ void __launch_bounds__(256) norm_partial_kernel(const T *y, int y_ps, const T *dz, int dz_ps, NormGeom g,
                                                           const float *mean, const float *rstd, const float *scale, const float *shift,
                                                           float *part) {
    constexpr int act = ACT;
    __shared__ float red[256 * 17];
    const int tid = threadIdx.x;
    const int n = blockIdx.x / g.nchunks, chunk = blockIdx.x % g.nchunks;
    const int cvec = g.Cp / 8;                         // 16-byte columns per pixel
    const int p0 = chunk * g.ppc, p1 = min(g.HW, p0 + g.ppc);
    for (int cbase = 0; cbase < cvec; cbase += 256) {  // Cp > 2048 never happens; loop kept for generality
        const int tpp = min(cvec - cbase, 256);        // threads per pixel
        const int rows = 256 / tpp;
        const int col = tid % tpp, row = tid / tpp;
        float s1[8], s2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
        if (row < rows) {
            const int c0 = (cbase + col) * 8;
            float mu[8], rs[8], sc[8], sh[8];
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    mu[i] = mean[n * g.Cp + c0 + i]; rs[i] = rstd[n * g.Cp + c0 + i];
                    sc[i] = scale[n * g.Cp + c0 + i]; sh[i] = shift[n * g.Cp + c0 + i];
                }
            }
            for (int p = p0 + row; p < p1; p += rows) {
                const size_t pix = (size_t)n * g.HW + p;
                float v[8];
                nld_p<T>(y + pix * y_ps + c0, v);
                if (MODE == 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { s1[i] += v[i]; s2[i] += v[i] * v[i]; }
                } else {
                    float d[8];
                    nld_pd<T>(dz + pix * dz_ps + c0, d);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float nv = v[i] * sc[i] + sh[i];
                        float dn = d[i];
                        if (act == DL_ACT_RELU) dn = nv > 0.f ? dn : 0.f;
                        else if (act == DL_ACT_LRELU) dn = nv > 0.f ? dn : 0.2f * dn;
                        else if (act == DL_ACT_TANH) { const float t = tanhf(nv); dn *= 1.f - t * t; }
                        s1[i] += dn;
                        s2[i] += dn * (v[i] - mu[i]) * rs[i];
                    }
                }
            }
        }
        // block reduction over `rows` for each column
#pragma unroll
        for (int i = 0; i < 8; ++i) { red[tid * 17 + i] = s1[i]; red[tid * 17 + 8 + i] = s2[i]; }
        __syncthreads();
        if (tid < tpp) {
            float a1[8], a2[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a1[i] = a2[i] = 0.f;
            for (int r = 0; r < rows; ++r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { a1[i] += red[(r * tpp + tid) * 17 + i]; a2[i] += red[(r * tpp + tid) * 17 + 8 + i]; }
            }
            float *o = part + ((size_t)(n * g.nchunks + chunk) * 2) * g.Cp + (cbase + tid) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) { o[i] = a1[i]; o[g.Cp + i] = a2[i]; }
        }
        __syncthreads();
    }
}

// stage A of both finalizers: S[n][0|1][c] = sum over the image's chunks of the per-block partials.
// block = 32 channels x 8 chunk lanes; grid = (Cp/32 rounded up, N)
// With FUSE != 0 (instance scope) the per-(n,c) statistics / backward constants are finalised right here:
//   FUSE 1 (forward) : mean, rstd, scale, shift           FUSE 2 (backward): c1 = S1/HW, c2 = S2/HW
template <int FUSE>
__global__ void __launch_bounds__(1024) norm_chunk_sum_kernel(const float *part, NormGeom g, float *sums, float eps, float *o0, float *o1,
                                                              float *o2, float *o3, const float *gamma, const float *beta) {
    // block = 32 channels x 32 chunk lanes: every thread owns at most nchunks/32 partials and loads them with independent
    // accumulators (the 8-lane version was a chain of 8-16 dependent L2 round trips: 11 us for < 2 MB)
    constexpr int KL = 32;
    __shared__ float red[2][KL][33];
    const int cl = threadIdx.x & 31, kl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl, n = blockIdx.y;
    float s1 = 0.f, s2 = 0.f;
    if (c < g.Cp) {
        float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
        const float *base = part + ((size_t)n * g.nchunks * 2) * g.Cp + c;
        int k = kl;
        for (; k + 3 * KL < g.nchunks; k += 4 * KL) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float *o = base + (size_t)(k + u * KL) * 2 * g.Cp;
                a1[u] += o[0];
                a2[u] += o[g.Cp];
            }
        }
        for (int u = 0; k < g.nchunks; k += KL, ++u) {
            const float *o = base + (size_t)k * 2 * g.Cp;
            a1[u & 3] += o[0];
            a2[u & 3] += o[g.Cp];
        }
        s1 = (a1[0] + a1[1]) + (a1[2] + a1[3]);
        s2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
    }
    red[0][kl][cl] = s1;
    red[1][kl][cl] = s2;
    __syncthreads();
    if (kl == 0 && c < g.Cp) {
        double a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int r = 0; r < KL; ++r) { a1 += (double)red[0][r][cl]; a2 += (double)red[1][r][cl]; }
        sums[((size_t)n * 2) * g.Cp + c] = (float)a1;
        sums[((size_t)n * 2 + 1) * g.Cp + c] = (float)a2;
        if (FUSE == 1) {           // per-(n, c) statistics: InstanceNorm2d (no affine, networks.py:36-37), or BatchNorm2d's affine on the statistics of ONE tile
                                   // (batched inference with per-sample normalisation) -- same formulas as norm_fwd_finalize_kernel
            // with an affine the sums go through their fp32 rounding first, as norm_fwd_finalize_kernel reads them: a tile normalised on its own statistics must come
            // out bit-identical whether it was served as one of a batch (this path) or alone in train() mode (batch scope, N = 1: the finalize kernel)
            const bool affine = gamma != nullptr || beta != nullptr;
            const double m1 = affine ? (double)(float)a1 : a1, m2 = affine ? (double)(float)a2 : a2;
            const double mu = m1 / g.HW;
            double var = m2 / g.HW - mu * mu;
            if (var < 0.0) var = 0.0;
            const float rs = (float)(1.0 / sqrt(var + (double)eps));
            float sc = 0.f, sh = 0.f;
            if (c < g.C) {
                const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
                sc = ga * rs;
                sh = be - (float)mu * sc;
            }
            o0[n * g.Cp + c] = (float)mu; o1[n * g.Cp + c] = rs; o2[n * g.Cp + c] = sc; o3[n * g.Cp + c] = sh;
        } else if (FUSE == 2) {
            o0[n * g.Cp + c] = (float)(a1 / g.HW); o1[n * g.Cp + c] = (float)(a2 / g.HW);
        }
    }
}

// forward finalize: one thread per channel (batch) or per (n, channel) (instance)
__global__ void __launch_bounds__(256) norm_fwd_finalize_kernel(const float *sums, NormGeom g, int scope, float eps, const float *gamma,
                                                                const float *beta, float *running_mean, float *running_var, float momentum,
                                                                float *mean, float *rstd, float *scale, float *shift) {
    const int total = (scope == DL_NORM_BATCH) ? g.Cp : g.N * g.Cp;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = i % g.Cp;
    const int n0 = (scope == DL_NORM_BATCH) ? 0 : i / g.Cp;
    const int n1 = (scope == DL_NORM_BATCH) ? g.N : n0 + 1;
    double s1 = 0.0, s2 = 0.0;
    for (int n = n0; n < n1; ++n) {
        s1 += (double)sums[((size_t)n * 2) * g.Cp + c];
        s2 += (double)sums[((size_t)n * 2 + 1) * g.Cp + c];
    }
    const double cnt = (double)(n1 - n0) * g.HW;
    const double mu = s1 / cnt;
    double var = s2 / cnt - mu * mu;
    if (var < 0.0) var = 0.0;
    const float rs = (float)(1.0 / sqrt(var + (double)eps));
    float sc = 0.f, sh = 0.f;
    if (c < g.C) {
        const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
        sc = ga * rs;
        sh = be - (float)mu * sc;
        if (scope == DL_NORM_BATCH && running_mean && momentum >= 0.f) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
            const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    }
    for (int n = n0; n < n1; ++n) {
        mean[n * g.Cp + c] = (float)mu; rstd[n * g.Cp + c] = rs; scale[n * g.Cp + c] = sc; shift[n * g.Cp + c] = sh;
    }
}

// z = act(y*scale + shift) (+res): each thread owns one 8-channel column of one image (per-channel constants live in
// registers) and walks down the pixels; grid = (pixel blocks, N)
template <typename T, int ACT>
__global__ void __launch_bounds__(256) norm_apply_kernel(const T *y, int y_ps, const float *scale, const float *shift, const T *res, int r_ps,
                                                         T *z, int z_ps, NormGeom g, float *split, int split_ps) {
    constexpr int act = ACT;
    const int cvec = g.Cp / 8;
    const int n = blockIdx.y;
    for (int cbase = 0; cbase < cvec; cbase += 256) {
        const int tpp = min(cvec - cbase, 256), rows = 256 / tpp;
        const int col = threadIdx.x % tpp, row = threadIdx.x / tpp;
        if (row >= rows) continue;
        const int c0 = (cbase + col) * 8;
        float sc[8], sh[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { sc[k] = scale[n * g.Cp + c0 + k]; sh[k] = shift[n * g.Cp + c0 + k]; }
        // 4 pixels per iteration with every load issued before the first use: one 16-byte load in flight per thread capped the
        // kernel at ~4.0 TB/s (32 KB in flight per CU over ~2 us of loaded latency); the arithmetic per element is unchanged
        constexpr int U = 4;
        const int pstep = gridDim.x * rows;
        int p = blockIdx.x * rows + row;
        for (; p + (U - 1) * pstep < g.HW; p += U * pstep) {
            float v[U][8], r[U][8];
#pragma unroll
            for (int u = 0; u < U; ++u) nld_a<T>(y + ((size_t)n * g.HW + p + u * pstep) * y_ps + c0, v[u]);
            if (res) {
#pragma unroll
                for (int u = 0; u < U; ++u) nld_a<T>(res + ((size_t)n * g.HW + p + u * pstep) * r_ps + c0, r[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[u][k] = apply_act(act, v[u][k] * sc[k] + sh[k]);
                if (res) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[u][k] += r[u][k];
                }
                if (z) nst_a<T>(z + ((size_t)n * g.HW + p + u * pstep) * z_ps + c0, v[u]);       // z == NULL: only the split copy is wanted
                if (split) store_split8(split + ((size_t)n * g.HW + p + u * pstep) * split_ps + c0, v[u]);
            }
        }
        for (; p < g.HW; p += pstep) {
            const size_t pix = (size_t)n * g.HW + p;
            float v[8];
            nld_a<T>(y + pix * y_ps + c0, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = apply_act(act, v[k] * sc[k] + sh[k]);
            if (res) {
                float r[8];
                nld_a<T>(res + pix * r_ps + c0, r);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += r[k];
            }
            if (z) nst_a<T>(z + pix * z_ps + c0, v);
            if (split) store_split8(split + pix * split_ps + c0, v);
        }
    }
}

// backward finalize: c1 = S1/m, c2 = S2/m per (n,c); dgamma/dbeta per channel.  One thread per channel, N is small.
__global__ void __launch_bounds__(256) norm_bwd_finalize_kernel(const float *sums, NormGeom g, int scope, float *c1, float *c2,
                                                                float *dgamma, float *dbeta, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.Cp) return;
    double t1 = 0.0, t2 = 0.0;
    for (int n = 0; n < g.N; ++n) {
        const double s1 = (double)sums[((size_t)n * 2) * g.Cp + c], s2 = (double)sums[((size_t)n * 2 + 1) * g.Cp + c];
        t1 += s1; t2 += s2;
        if (scope == DL_NORM_INSTANCE) { c1[n * g.Cp + c] = (float)(s1 / g.HW); c2[n * g.Cp + c] = (float)(s2 / g.HW); }
    }
    if (scope == DL_NORM_BATCH) {
        const double m = (double)g.N * g.HW;
        for (int n = 0; n < g.N; ++n) { c1[n * g.Cp + c] = (float)(t1 / m); c2[n * g.Cp + c] = (float)(t2 / m); }
    }
    if (dgamma && c < g.C) {
        dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)t2;
        dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)t1;
    }
}

// dy = gamma*rstd*(dn - c1 - xhat*c2): same thread->column ownership as norm_apply_kernel
template <typename T, int ACT>
__global__ void __launch_bounds__(256) norm_bwd_apply_kernel(const T *dz, int dz_ps, const T *y, int y_ps, const float *gamma,
                                                             const float *mean, const float *rstd, const float *scale, const float *shift,
                                                             const float *c1, const float *c2, T *dy, int dy_ps, NormGeom g,
                                                             float *bpart, float *split, int split_ps) {
    constexpr int act = ACT;
    __shared__ float bred[256 * 9];
    const int cvec = g.Cp / 8;
    const int n = blockIdx.y;
    for (int cbase = 0; cbase < cvec; cbase += 256) {
        const int tpp = min(cvec - cbase, 256), rows = 256 / tpp;
        const int col = threadIdx.x % tpp, row = threadIdx.x / tpp;
        const int c0 = (cbase + col) * 8;
        float bs[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) bs[k] = 0.f;
        if (row < rows) {
        float sc[8], sh[8], rs[8], mr[8], k1[8], k2[8], gr[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int b = n * g.Cp + c0 + k;
            sc[k] = scale[b]; sh[k] = shift[b]; rs[k] = rstd[b]; mr[k] = -mean[b] * rstd[b]; k1[k] = c1[b]; k2[k] = c2[b];
            const float ga = (c0 + k < g.C) ? (gamma ? gamma[c0 + k] : 1.f) : 0.f;
            gr[k] = ga * rs[k];
        }
        // 4 pixels per iteration, loads first (see norm_apply_kernel); pixels are still consumed in the same order, so the
        // bias partial sums bs[] are accumulated exactly as before
        constexpr int U = 4;
        const int pstep = gridDim.x * rows;
        int p = blockIdx.x * rows + row;
        for (; p + (U - 1) * pstep < g.HW; p += U * pstep) {
            float v[U][8], d[U][8];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t pix = (size_t)n * g.HW + p + u * pstep;
                nld_a<T>(y + pix * y_ps + c0, v[u]);
                nld_a<T>(dz + pix * dz_ps + c0, d[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float o[8];
                const float (&V)[8] = v[u];
                const float (&D)[8] = d[u];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                    const float nv = V[k] * sc[k] + sh[k];
                    float dn = D[k];
                    if (act == DL_ACT_RELU) dn = nv > 0.f ? dn : 0.f;
                    else if (act == DL_ACT_LRELU) dn = nv > 0.f ? dn : 0.2f * dn;
                    else if (act == DL_ACT_TANH) { const float t = tanhf(nv); dn *= 1.f - t * t; }
                    const float xh = V[k] * rs[k] + mr[k];
                    o[k] = gr[k] * (dn - k1[k] - xh * k2[k]);
                    bs[k] += o[k];
                }
                if (dy) nst_a<T>(dy + ((size_t)n * g.HW + p + u * pstep) * dy_ps + c0, o);       // dy == NULL: only the split copy is wanted
                if (split) store_split8(split + ((size_t)n * g.HW + p + u * pstep) * split_ps + c0, o);
            }
        }
        for (; p < g.HW; p += pstep) {
            const size_t pix = (size_t)n * g.HW + p;
            float V[8], D[8], o[8];
            nld_a<T>(y + pix * y_ps + c0, V);
            nld_a<T>(dz + pix * dz_ps + c0, D);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float nv = V[k] * sc[k] + sh[k];
                float dn = D[k];
                if (act == DL_ACT_RELU) dn = nv > 0.f ? dn : 0.f;
                else if (act == DL_ACT_LRELU) dn = nv > 0.f ? dn : 0.2f * dn;
                else if (act == DL_ACT_TANH) { const float t = tanhf(nv); dn *= 1.f - t * t; }
                const float xh = V[k] * rs[k] + mr[k];
                o[k] = gr[k] * (dn - k1[k] - xh * k2[k]);
                bs[k] += o[k];
            }
            if (dy) nst_a<T>(dy + pix * dy_ps + c0, o);
            if (split) store_split8(split + pix * split_ps + c0, o);
        }
        }
        if (bpart) {       // per-block channel sums of dy: the gradient of the conv bias in front of this norm
#pragma unroll
            for (int k = 0; k < 8; ++k) bred[threadIdx.x * 9 + k] = bs[k];
            __syncthreads();
            if (threadIdx.x < tpp) {
                float a[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] = 0.f;
                for (int r = 0; r < rows; ++r)
#pragma unroll
                    for (int k = 0; k < 8; ++k) a[k] += bred[(r * tpp + threadIdx.x) * 9 + k];
                float *o = bpart + ((size_t)(n * gridDim.x + blockIdx.x)) * g.Cp + (cbase + threadIdx.x) * 8;
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = a[k];
            }
            __syncthreads();
        }
    }
}

// out[c] += sum over the per-block partials (block = 32 channels x 32 partial lanes)
__global__ void __launch_bounds__(1024) norm_bias_final_kernel(const float *part, int nparts, int Cp, int C, float *out) {
    __shared__ float red[32][33];
    const int cl = threadIdx.x & 31, kl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < C) {     // 8 independent accumulators: 64 dependent L2 round trips (nparts = 2048) become 8
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int b = kl;
        for (; b + 7 * 32 < nparts; b += 8 * 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += part[(size_t)(b + u * 32) * Cp + c];
        }
        for (int u = 0; b < nparts; b += 32, ++u) acc[u & 7] += part[(size_t)b * Cp + c];
        s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
    red[kl][cl] = s;
    __syncthreads();
    if (kl == 0 && c < C) {
        double a = 0.0;
#pragma unroll
        for (int r = 0; r < 32; ++r) a += (double)red[r][cl];
        out[c] += (float)a;
    }
}

// grid of the apply kernels: (pixel blocks per image, N); ~16 pixels per thread, at least enough blocks to fill 256 CUs
static dim3 apply_grid(const NormGeom &g) {
    const int cvec = g.Cp / 8;
    const int rows = 256 / (cvec < 256 ? cvec : 256);
    int bx = (g.HW + rows * 16 - 1) / (rows * 16);
    // A/B switch for the open launch-shape question (profiles/r01/pmc_norm/README.txt): DL_NORM_GRID_MUL scales the block count
    static const float mul = [] { const char *e = DL_DEV_ENV("DL_NORM_GRID_MUL"); const float v = e ? (float)atof(e) : 1.f; return v > 0.f ? v : 1.f; }();
    const int want = (int)((2048 * mul + g.N - 1) / g.N);
    if (bx > want) bx = want;
    if (bx < 1) bx = 1;
    return dim3(bx, g.N);
}

// instantiate a kernel launch for the activation of the descriptor (DL_ACT_NONE for anything unknown)
#define DL_NORM_ACT_SWITCH(act_, LAUNCH)                  \
    switch (act_) {                                       \
        case DL_ACT_RELU: LAUNCH(DL_ACT_RELU); break;     \
        case DL_ACT_LRELU: LAUNCH(DL_ACT_LRELU); break;   \
        case DL_ACT_TANH: LAUNCH(DL_ACT_TANH); break;     \
        default: LAUNCH(DL_ACT_NONE); break;              \
    }

static int check_desc(const dl_norm_desc *d, const char *who) {
    if (!d) DL_FAIL("%s: null desc", who);
    if (d->N <= 0 || d->H <= 0 || d->W <= 0) DL_FAIL("%s: empty problem (N=%d, %dx%d): nothing to launch", who, d->N, d->H, d->W);
    if (d->Cp % 8 || d->Cp <= 0 || d->C > d->Cp) DL_FAIL("%s: Cp=%d C=%d", who, d->Cp, d->C);
    if (d->y_pstride % 8 || d->z_pstride % 8 || (d->r_pstride % 8)) DL_FAIL("%s: pixel strides must be multiples of 8", who);
    if (d->dtype != DL_F32 && d->dtype != DL_BF16) DL_FAIL("%s: dtype %d", who, d->dtype);
    return 0;
}

extern "C" int dl_norm_forward(const dl_norm_desc *d, const void *y, const float *gamma, const float *beta,
                               float *running_mean, float *running_var, float *mean, float *rstd, float *scale, float *shift,
                               const void *residual, void *z, float *ws, void *z_split, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (check_desc(d, "dl_norm_forward")) return -1;
    if (z_split && d->dtype != DL_F32) DL_FAIL("dl_norm_forward: the split copy belongs to the fp32 (strict) policy");
    if (!y || !mean || !rstd || !scale || !shift || !ws) DL_FAIL("dl_norm_forward: null argument");
    if (!z && !z_split) DL_FAIL("dl_norm_forward: z may only be NULL when z_split receives the result");
    const NormGeom g = make_geom_fwd(d);
    const int pblocks = g.N * g.nchunks;
    if (d->ext_nchunks > 0) {
        // partial sums were produced by the convolution that wrote y
    } else if (d->dtype == DL_F32)
        hipLaunchKernelGGL((norm_partial_kernel<float, 0, DL_ACT_NONE>), dim3(pblocks), dim3(256), 0, stream, (const float *)y, d->y_pstride,
                           (const float *)nullptr, 0, g, nullptr, nullptr, nullptr, nullptr, ws);
    else
        hipLaunchKernelGGL((norm_partial_kernel<bf16_t, 0, DL_ACT_NONE>), dim3(pblocks), dim3(256), 0, stream, (const bf16_t *)y, d->y_pstride,
                           (const bf16_t *)nullptr, 0, g, nullptr, nullptr, nullptr, nullptr, ws);
    DL_CHECK_LAUNCH("dl_norm_forward(stats)");
    float *sums = ws + (size_t)g.N * g.nchunks * 2 * g.Cp;
    if (d->scope == DL_NORM_INSTANCE) {
        // (r06: also with an affine -- the 9-generator inference batch ran 185 norm_fwd_finalize launches of 4.7 us per step next to as many chunk sums)
        hipLaunchKernelGGL(norm_chunk_sum_kernel<1>, dim3((g.Cp + 31) / 32, g.N), dim3(1024), 0, stream, ws, g, sums, d->eps, mean, rstd, scale, shift, gamma, beta);
        DL_CHECK_LAUNCH("dl_norm_forward(chunk sums + finalize)");
    } else {
        hipLaunchKernelGGL(norm_chunk_sum_kernel<0>, dim3((g.Cp + 31) / 32, g.N), dim3(1024), 0, stream, ws, g, sums, d->eps, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        DL_CHECK_LAUNCH("dl_norm_forward(chunk sums)");
        const int ftotal = (d->scope == DL_NORM_BATCH) ? g.Cp : g.N * g.Cp;
        hipLaunchKernelGGL(norm_fwd_finalize_kernel, dim3((ftotal + 255) / 256), dim3(256), 0, stream, sums, g, d->scope, d->eps, gamma, beta,
                           running_mean, running_var, d->momentum, mean, rstd, scale, shift);
        DL_CHECK_LAUNCH("dl_norm_forward(finalize)");
    }
    const dim3 blocks = apply_grid(g);
#define DL_LAUNCH_APPLY_F32(A) hipLaunchKernelGGL((norm_apply_kernel<float, A>), blocks, dim3(256), 0, stream, (const float *)y, d->y_pstride, \
                                                  scale, shift, (const float *)residual, d->r_pstride, (float *)z, d->z_pstride, g, (float *)z_split, d->Cp)
#define DL_LAUNCH_APPLY_BF16(A) hipLaunchKernelGGL((norm_apply_kernel<bf16_t, A>), blocks, dim3(256), 0, stream, (const bf16_t *)y, d->y_pstride, \
                                                   scale, shift, (const bf16_t *)residual, d->r_pstride, (bf16_t *)z, d->z_pstride, g, (float *)nullptr, 0)
    if (d->dtype == DL_F32) { DL_NORM_ACT_SWITCH(d->act, DL_LAUNCH_APPLY_F32) }
    else { DL_NORM_ACT_SWITCH(d->act, DL_LAUNCH_APPLY_BF16) }
    DL_CHECK_LAUNCH("dl_norm_forward(apply)");
    return 0;
}

extern "C" int dl_norm_backward(const dl_norm_desc *d, const void *dz, const void *y, const float *gamma,
                                const float *mean, const float *rstd, const float *scale, const float *shift,
                                void *dy, float *dgamma, float *dbeta, int accumulate_affine, float *dy_chansum, float *ws, void *dy_split,
                                void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (check_desc(d, "dl_norm_backward")) return -1;
    if (dy_split && d->dtype != DL_F32) DL_FAIL("dl_norm_backward: the split copy belongs to the fp32 (strict) policy");
    if (!dz || !y || !mean || !rstd || !scale || !shift || !ws) DL_FAIL("dl_norm_backward: null argument");
    if (!dy && !dy_split) DL_FAIL("dl_norm_backward: dy may only be NULL when dy_split receives the result");
    // ext_nchunks > 0: the producer of dz (dl_conv_forward_bnstats) already left the [N][ext_nchunks][2][Cp] partials at the start of ws
    const bool ext = d->ext_nchunks > 0;
    const NormGeom g = make_geom_fwd(d);
    float *part = ws;
    float *sums = ws + (size_t)g.N * g.nchunks * 2 * g.Cp;
    float *c1 = sums + (size_t)2 * g.N * g.Cp;
    float *c2 = c1 + (size_t)g.N * g.Cp;
    const int pblocks = g.N * g.nchunks;
    // dz uses z_pstride, dy uses r_pstride slot of the desc (documented in ops.py): keep explicit names here
    const int dz_ps = d->z_pstride, dy_ps = d->r_pstride;
#define DL_LAUNCH_BRED_F32(A) hipLaunchKernelGGL((norm_partial_kernel<float, 1, A>), dim3(pblocks), dim3(256), 0, stream, (const float *)y, \
                                                 d->y_pstride, (const float *)dz, dz_ps, g, mean, rstd, scale, shift, part)
#define DL_LAUNCH_BRED_BF16(A) hipLaunchKernelGGL((norm_partial_kernel<bf16_t, 1, A>), dim3(pblocks), dim3(256), 0, stream, (const bf16_t *)y, \
                                                  d->y_pstride, (const bf16_t *)dz, dz_ps, g, mean, rstd, scale, shift, part)
    if (ext) { /* partials are in place */ }
    else if (d->dtype == DL_F32) { DL_NORM_ACT_SWITCH(d->act, DL_LAUNCH_BRED_F32) }
    else { DL_NORM_ACT_SWITCH(d->act, DL_LAUNCH_BRED_BF16) }
    DL_CHECK_LAUNCH("dl_norm_backward(reduce)");
    if (d->scope == DL_NORM_INSTANCE && !dgamma) {
        hipLaunchKernelGGL(norm_chunk_sum_kernel<2>, dim3((g.Cp + 31) / 32, g.N), dim3(1024), 0, stream, part, g, sums, d->eps, c1, c2, nullptr, nullptr, nullptr, nullptr);
        DL_CHECK_LAUNCH("dl_norm_backward(chunk sums + finalize)");
    } else {
        hipLaunchKernelGGL(norm_chunk_sum_kernel<0>, dim3((g.Cp + 31) / 32, g.N), dim3(1024), 0, stream, part, g, sums, d->eps, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        DL_CHECK_LAUNCH("dl_norm_backward(chunk sums)");
        hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3((g.Cp + 255) / 256), dim3(256), 0, stream, sums, g, d->scope, c1, c2, dgamma, dbeta,
                           accumulate_affine);
        DL_CHECK_LAUNCH("dl_norm_backward(finalize)");
    }
    float *bpart = dy_chansum ? c2 + (size_t)g.N * g.Cp : nullptr;
    const dim3 blocks = apply_grid(g);
#define DL_LAUNCH_BAPPLY_F32(A) hipLaunchKernelGGL((norm_bwd_apply_kernel<float, A>), blocks, dim3(256), 0, stream, (const float *)dz, dz_ps, \
                                                   (const float *)y, d->y_pstride, gamma, mean, rstd, scale, shift, c1, c2, (float *)dy, dy_ps, g, bpart, \
                                                   (float *)dy_split, d->Cp)
#define DL_LAUNCH_BAPPLY_BF16(A) hipLaunchKernelGGL((norm_bwd_apply_kernel<bf16_t, A>), blocks, dim3(256), 0, stream, (const bf16_t *)dz, dz_ps, \
                                                    (const bf16_t *)y, d->y_pstride, gamma, mean, rstd, scale, shift, c1, c2, (bf16_t *)dy, dy_ps, g, bpart, \
                                                    (float *)nullptr, 0)
    if (d->dtype == DL_F32) { DL_NORM_ACT_SWITCH(d->act, DL_LAUNCH_BAPPLY_F32) }
    else { DL_NORM_ACT_SWITCH(d->act, DL_LAUNCH_BAPPLY_BF16) }
    DL_CHECK_LAUNCH("dl_norm_backward(apply)");
    if (dy_chansum) {
        hipLaunchKernelGGL(norm_bias_final_kernel, dim3((g.C + 31) / 32), dim3(1024), 0, stream, bpart, (int)(blocks.x * blocks.y), g.Cp, g.C, dy_chansum);
        DL_CHECK_LAUNCH("dl_norm_backward(bias sum)");
    }
    return 0;
}
