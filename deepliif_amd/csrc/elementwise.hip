// elementwise.hip -- activations, axpby, channel copies (torch.cat / slicing), channel sums (bias gradients), GAN / SmoothL1
// losses with their gradients, fused Adam, and the hardware probes.  All HBM-bound: 16-byte vector accesses along the NHWC
// channel axis, grid-stride loops, fixed-order two-stage reductions (deterministic).  See include/deepliif_hip.h.
#include "common.h"

#define EW_BLOCKS(total) ((int)min((size_t)8192, ((size_t)(total) + 255) / 256))

// ------------------------------------------------------------------------------------------- activations
template <typename T>
__global__ void __launch_bounds__(256) act_fwd_kernel(int act, const T *x, int x_ps, T *y, int y_ps, size_t npix, int Cp) {
    const int cvec = Cp / 8;
    const size_t total = npix * cvec;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / cvec;
        const int c0 = (int)(i % cvec) * 8;
        float v[8];
        Vec8<T>::load(x + p * x_ps + c0, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = apply_act(act, v[k]);
        Vec8<T>::store(y + p * y_ps + c0, v);
    }
}
template <typename T>
__global__ void __launch_bounds__(256) act_bwd_kernel(int act, const T *dy, int dy_ps, const T *y, int y_ps, T *dx, int dx_ps, size_t npix,
                                                      int Cp) {
    const int cvec = Cp / 8;
    const size_t total = npix * cvec;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / cvec;
        const int c0 = (int)(i % cvec) * 8;
        float g[8], o[8];
        Vec8<T>::load(dy + p * dy_ps + c0, g);
        Vec8<T>::load(y + p * y_ps + c0, o);
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] *= act_grad_from_output(act, o[k]);
        Vec8<T>::store(dx + p * dx_ps + c0, g);
    }
}
extern "C" int dl_act_forward(int act, int dtype, const void *x, int x_ps, void *y, int y_ps, int64_t npix, int Cp, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !y || Cp % 8 || x_ps % 8 || y_ps % 8) DL_FAIL("dl_act_forward: bad argument");
    const size_t total = (size_t)npix * (Cp / 8);
    if (dtype == DL_F32) hipLaunchKernelGGL(act_fwd_kernel<float>, dim3(EW_BLOCKS(total)), dim3(256), 0, stream, act, (const float *)x, x_ps, (float *)y, y_ps, (size_t)npix, Cp);
    else hipLaunchKernelGGL(act_fwd_kernel<bf16_t>, dim3(EW_BLOCKS(total)), dim3(256), 0, stream, act, (const bf16_t *)x, x_ps, (bf16_t *)y, y_ps, (size_t)npix, Cp);
    DL_CHECK_LAUNCH("dl_act_forward");
    return 0;
}
extern "C" int dl_act_backward(int act, int dtype, const void *dy, int dy_ps, const void *y, int y_ps, void *dx, int dx_ps, int64_t npix,
                               int Cp, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!dy || !y || !dx || Cp % 8 || dy_ps % 8 || y_ps % 8 || dx_ps % 8) DL_FAIL("dl_act_backward: bad argument");
    const size_t total = (size_t)npix * (Cp / 8);
    if (dtype == DL_F32) hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(EW_BLOCKS(total)), dim3(256), 0, stream, act, (const float *)dy, dy_ps, (const float *)y, y_ps, (float *)dx, dx_ps, (size_t)npix, Cp);
    else hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(EW_BLOCKS(total)), dim3(256), 0, stream, act, (const bf16_t *)dy, dy_ps, (const bf16_t *)y, y_ps, (bf16_t *)dx, dx_ps, (size_t)npix, Cp);
    DL_CHECK_LAUNCH("dl_act_backward");
    return 0;
}

// ------------------------------------------------------------------------------------------- attention gate (att_unet.py:108-115)
template <typename T>
__global__ void __launch_bounds__(256) gate_fwd_kernel(const T *x, int x_ps, const T *psi, int psi_ps, T *out, int o_ps, size_t npix, int Cp) {
    const int cvec = Cp / 8;
    const size_t total = npix * cvec;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / cvec;
        const int c0 = (int)(i % cvec) * 8;
        float v[8];
        Vec8<T>::load(x + p * x_ps + c0, v);
        const float a = load1<T>(psi + p * psi_ps);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= a;
        Vec8<T>::store(out + p * o_ps + c0, v);
    }
}
// one thread per (pixel, 8-channel chunk); the cvec = Cp / 8 <= 64 chunks of a pixel sit in adjacent lanes of ONE wave (cvec is a power of
// two and the grid stride is a multiple of 64), so dpsi = sum_c g * x is a butterfly over those lanes
template <typename T>
__global__ void __launch_bounds__(256) gate_bwd_kernel(const T *g, int g_ps, const T *x, int x_ps, const T *psi, int psi_ps, T *dx, int dx_ps, T *dpsi,
                                                       int dpsi_ps, size_t npix, int Cp) {
    const int cvec = Cp / 8;
    const size_t total = npix * cvec;
    const size_t rounded = (total + 63) / 64 * 64;               // whole waves stay converged for the shuffles
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < rounded; i += (size_t)gridDim.x * blockDim.x) {
        const bool live = i < total;
        const size_t p = live ? i / cvec : 0;
        const int c0 = (int)(i % cvec) * 8;
        float gv[8], xv[8];
        float part = 0.f;
        if (live) {
            Vec8<T>::load(g + p * g_ps + c0, gv);
            Vec8<T>::load(x + p * x_ps + c0, xv);
            const float a = load1<T>(psi + p * psi_ps);
#pragma unroll
            for (int k = 0; k < 8; ++k) { part += gv[k] * xv[k]; gv[k] *= a; }
            if (dx) Vec8<T>::store(dx + p * dx_ps + c0, gv);
        }
        for (int o = cvec >> 1; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if (live && c0 == 0) {
            float d[8] = {part, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            Vec8<T>::store(dpsi + p * dpsi_ps, d);
        }
    }
}
extern "C" int dl_gate_forward(int dtype, const void *x, int x_ps, const void *psi, int psi_ps, void *out, int o_ps, int64_t npix, int Cp, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !psi || !out || Cp % 8 || x_ps % 8 || psi_ps % 8 || o_ps % 8 || npix <= 0) DL_FAIL("dl_gate_forward: bad argument");
    const size_t total = (size_t)npix * (Cp / 8);
    if (dtype == DL_F32) hipLaunchKernelGGL(gate_fwd_kernel<float>, dim3(EW_BLOCKS(total)), dim3(256), 0, stream, (const float *)x, x_ps, (const float *)psi, psi_ps, (float *)out, o_ps, (size_t)npix, Cp);
    else hipLaunchKernelGGL(gate_fwd_kernel<bf16_t>, dim3(EW_BLOCKS(total)), dim3(256), 0, stream, (const bf16_t *)x, x_ps, (const bf16_t *)psi, psi_ps, (bf16_t *)out, o_ps, (size_t)npix, Cp);
    DL_CHECK_LAUNCH("dl_gate_forward");
    return 0;
}
extern "C" int dl_gate_backward(int dtype, const void *g, int g_ps, const void *x, int x_ps, const void *psi, int psi_ps, void *dx, int dx_ps, void *dpsi,
                                int dpsi_ps, int64_t npix, int Cp, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!g || !x || !psi || !dpsi || Cp % 8 || g_ps % 8 || x_ps % 8 || psi_ps % 8 || dpsi_ps % 8 || (dx && dx_ps % 8) || npix <= 0) DL_FAIL("dl_gate_backward: bad argument");
    const int cvec = Cp / 8;
    if (cvec > 64 || (cvec & (cvec - 1))) DL_FAIL("dl_gate_backward: Cp=%d must be a power of two <= 512 (one wave reduces the channels of a pixel)", Cp);
    const size_t total = (size_t)npix * cvec;
    if (dtype == DL_F32) hipLaunchKernelGGL(gate_bwd_kernel<float>, dim3(EW_BLOCKS(total)), dim3(256), 0, stream, (const float *)g, g_ps, (const float *)x, x_ps, (const float *)psi, psi_ps, (float *)dx, dx_ps, (float *)dpsi, dpsi_ps, (size_t)npix, Cp);
    else hipLaunchKernelGGL(gate_bwd_kernel<bf16_t>, dim3(EW_BLOCKS(total)), dim3(256), 0, stream, (const bf16_t *)g, g_ps, (const bf16_t *)x, x_ps, (const bf16_t *)psi, psi_ps, (bf16_t *)dx, dx_ps, (bf16_t *)dpsi, dpsi_ps, (size_t)npix, Cp);
    DL_CHECK_LAUNCH("dl_gate_backward");
    return 0;
}

// ------------------------------------------------------------------------------------------- out = alpha*a + beta*b
template <typename T>
__global__ void __launch_bounds__(256) axpby_kernel(float alpha, const T *a, int a_ps, float beta, const T *b, int b_ps, T *out, int o_ps,
                                                    size_t npix, int Cp) {
    const int cvec = Cp / 8;
    const size_t total = npix * cvec;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / cvec;
        const int c0 = (int)(i % cvec) * 8;
        float va[8], vb[8];
        Vec8<T>::load(a + p * a_ps + c0, va);
        if (b) {
            Vec8<T>::load(b + p * b_ps + c0, vb);
#pragma unroll
            for (int k = 0; k < 8; ++k) va[k] = alpha * va[k] + beta * vb[k];
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) va[k] = alpha * va[k];
        }
        Vec8<T>::store(out + p * o_ps + c0, va);
    }
}
extern "C" int dl_axpby(int dtype, float alpha, const void *a, int a_ps, float beta, const void *b, int b_ps, void *out, int o_ps,
                        int64_t npix, int Cp, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!a || !out || Cp % 8 || a_ps % 8 || o_ps % 8 || (b && b_ps % 8)) DL_FAIL("dl_axpby: bad argument");
    const size_t total = (size_t)npix * (Cp / 8);
    if (dtype == DL_F32) hipLaunchKernelGGL(axpby_kernel<float>, dim3(EW_BLOCKS(total)), dim3(256), 0, stream, alpha, (const float *)a, a_ps, beta, (const float *)b, b_ps, (float *)out, o_ps, (size_t)npix, Cp);
    else hipLaunchKernelGGL(axpby_kernel<bf16_t>, dim3(EW_BLOCKS(total)), dim3(256), 0, stream, alpha, (const bf16_t *)a, a_ps, beta, (const bf16_t *)b, b_ps, (bf16_t *)out, o_ps, (size_t)npix, Cp);
    DL_CHECK_LAUNCH("dl_axpby");
    return 0;
}

// ------------------------------------------------------------------------------------------- channel copy (cat / slice)
template <typename T>
__global__ void __launch_bounds__(256) copy_channels_kernel(const T *src, int s_ps, int s_c0, T *dst, int d_ps, int d_c0, size_t npix, int C,
                                                            int accumulate) {
    const size_t total = npix * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / C;
        const int c = (int)(i % C);
        float v = load1<T>(src + p * s_ps + s_c0 + c);
        T *d = dst + p * d_ps + d_c0 + c;
        if (accumulate) v += load1<T>(d);
        store1<T>(d, v);
    }
}
extern "C" int dl_copy_channels(int dtype, const void *src, int s_ps, int s_c0, void *dst, int d_ps, int d_c0, int64_t npix, int C,
                                int accumulate, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!src || !dst || C <= 0) DL_FAIL("dl_copy_channels: bad argument");
    const size_t total = (size_t)npix * C;
    if (dtype == DL_F32) hipLaunchKernelGGL(copy_channels_kernel<float>, dim3(EW_BLOCKS(total)), dim3(256), 0, stream, (const float *)src, s_ps, s_c0, (float *)dst, d_ps, d_c0, (size_t)npix, C, accumulate);
    else hipLaunchKernelGGL(copy_channels_kernel<bf16_t>, dim3(EW_BLOCKS(total)), dim3(256), 0, stream, (const bf16_t *)src, s_ps, s_c0, (bf16_t *)dst, d_ps, d_c0, (size_t)npix, C, accumulate);
    DL_CHECK_LAUNCH("dl_copy_channels");
    return 0;
}

// ------------------------------------------------------------------------------------------- nn.Upsample(scale_factor=2, mode='nearest')
// ResnetGenerator with --upsample resize_conv (networks.py:409-415): y[n, 2h + a, 2w + b, c] = x[n, h, w, c]; backward dx = sum of the 2 x 2 block.
// One thread = one INPUT pixel x 8 channels (16-byte accesses).
template <typename T, int BWD>
__global__ void __launch_bounds__(256) upsample2_kernel(const T *src, int s_ps, T *dst, int d_ps, int N, int H, int W, int Cp) {
    const int cvec = Cp / 8;
    const size_t total = (size_t)N * H * W * cvec;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cvec) * 8;
        size_t p = i / cvec;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int n = (int)(p / H);
        const size_t big = ((size_t)n * 2 * H + 2 * h) * (2 * W) + 2 * w;          // pixel (2h, 2w) of the [N, 2H, 2W] tensor
        const size_t small_ = ((size_t)n * H + h) * W + w;
        float v[8];
        if (BWD) {
            float a[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                Vec8<T>::load(src + (big + (size_t)(q >> 1) * 2 * W + (q & 1)) * s_ps + c8, a);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += a[k];
            }
            Vec8<T>::store(dst + small_ * d_ps + c8, v);
        } else {
            Vec8<T>::load(src + small_ * s_ps + c8, v);
#pragma unroll
            for (int q = 0; q < 4; ++q) Vec8<T>::store(dst + (big + (size_t)(q >> 1) * 2 * W + (q & 1)) * d_ps + c8, v);
        }
    }
}
extern "C" int dl_upsample2_nearest(int dtype, int backward, const void *src, int s_ps, void *dst, int d_ps, int N, int H, int W, int Cp, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!src || !dst || N <= 0 || H <= 0 || W <= 0 || Cp <= 0 || Cp % 8 || s_ps % 8 || d_ps % 8) DL_FAIL("dl_upsample2_nearest: bad argument");
    const size_t total = (size_t)N * H * W * (Cp / 8);
    if (dtype == DL_F32) {
        if (backward) hipLaunchKernelGGL((upsample2_kernel<float, 1>), dim3(EW_BLOCKS(total)), dim3(256), 0, stream, (const float *)src, s_ps, (float *)dst, d_ps, N, H, W, Cp);
        else hipLaunchKernelGGL((upsample2_kernel<float, 0>), dim3(EW_BLOCKS(total)), dim3(256), 0, stream, (const float *)src, s_ps, (float *)dst, d_ps, N, H, W, Cp);
    } else if (dtype == DL_BF16) {
        if (backward) hipLaunchKernelGGL((upsample2_kernel<bf16_t, 1>), dim3(EW_BLOCKS(total)), dim3(256), 0, stream, (const bf16_t *)src, s_ps, (bf16_t *)dst, d_ps, N, H, W, Cp);
        else hipLaunchKernelGGL((upsample2_kernel<bf16_t, 0>), dim3(EW_BLOCKS(total)), dim3(256), 0, stream, (const bf16_t *)src, s_ps, (bf16_t *)dst, d_ps, N, H, W, Cp);
    } else DL_FAIL("dl_upsample2_nearest: dtype %d", dtype);
    DL_CHECK_LAUNCH("dl_upsample2_nearest");
    return 0;
}

// ------------------------------------------------------------------------------------------- per-channel sum over pixels
#define CS_BLOCKS 256
template <typename T>
__global__ void __launch_bounds__(256) channel_sum_partial_kernel(const T *x, int ps, size_t npix, int Cp, float *part) {
    __shared__ float red[256 * 9];
    const int tid = threadIdx.x;
    const int cvec = Cp / 8;
    for (int cbase = 0; cbase < cvec; cbase += 256) {
        const int tpp = min(cvec - cbase, 256), rows = 256 / tpp;
        const int col = tid % tpp, row = tid / tpp;
        float s[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] = 0.f;
        if (row < rows) {
            for (size_t p = (size_t)blockIdx.x * rows + row; p < npix; p += (size_t)gridDim.x * rows) {
                float v[8];
                Vec8<T>::load(x + p * ps + (cbase + col) * 8, v);
#pragma unroll
                for (int k = 0; k < 8; ++k) s[k] += v[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) red[tid * 9 + k] = s[k];
        __syncthreads();
        if (tid < tpp) {
            float a[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = 0.f;
            for (int r = 0; r < rows; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] += red[(r * tpp + tid) * 9 + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) part[(size_t)blockIdx.x * Cp + (cbase + tid) * 8 + k] = a[k];
        }
        __syncthreads();
    }
}
// block = 32 channels x 8 partial lanes
__global__ void __launch_bounds__(256) channel_sum_final_kernel(const float *part, int nblocks, int Cp, int C, float *out, int accumulate) {
    __shared__ float red[8][33];
    const int cl = threadIdx.x & 31, kl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < C)
        for (int b = kl; b < nblocks; b += 8) s += part[(size_t)b * Cp + c];
    red[kl][cl] = s;
    __syncthreads();
    if (kl == 0 && c < C) {
        double a = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) a += (double)red[r][cl];
        out[c] = (accumulate ? out[c] : 0.f) + (float)a;
    }
}
extern "C" int dl_channel_sum(int dtype, const void *x, int ps, int64_t npix, int Cp, int C, float *out, int accumulate, float *ws,
                              void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !out || !ws || Cp % 8 || ps % 8) DL_FAIL("dl_channel_sum: bad argument (ws needs %d*Cp floats)", CS_BLOCKS);
    if (dtype == DL_F32) hipLaunchKernelGGL(channel_sum_partial_kernel<float>, dim3(CS_BLOCKS), dim3(256), 0, stream, (const float *)x, ps, (size_t)npix, Cp, ws);
    else hipLaunchKernelGGL(channel_sum_partial_kernel<bf16_t>, dim3(CS_BLOCKS), dim3(256), 0, stream, (const bf16_t *)x, ps, (size_t)npix, Cp, ws);
    DL_CHECK_LAUNCH("dl_channel_sum(partial)");
    hipLaunchKernelGGL(channel_sum_final_kernel, dim3((C + 31) / 32), dim3(256), 0, stream, ws, CS_BLOCKS, Cp, C, out, accumulate);
    DL_CHECK_LAUNCH("dl_channel_sum(final)");
    return 0;
}

// ------------------------------------------------------------------------------------------- losses
#define LOSS_BLOCKS 512
extern "C" size_t dl_loss_ws_floats(void) { return LOSS_BLOCKS; }

template <typename T>
__global__ void __launch_bounds__(256) loss_kernel(int kind, const T *x, int x_ps, const T *target, int t_ps, float tconst, size_t npix, int C,
                                                   float inv_count, T *grad, int g_ps, float gscale, int Cp, float *part) {
    __shared__ float red[4];
    const size_t total = npix * C;
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / C;
        const int c = (int)(i % C);
        const float v = load1<T>(x + p * x_ps + c);
        const float t = target ? load1<T>(target + p * t_ps + c) : tconst;
        float l, g;
        if (kind == DL_LOSS_BCE_LOGITS) {
            // max(v,0) - v*t + log(1 + exp(-|v|));  d/dv = sigmoid(v) - t
            const float e = expf(-fabsf(v));
            l = fmaxf(v, 0.f) - v * t + log1pf(e);
            const float sig = v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
            g = sig - t;
        } else if (kind == DL_LOSS_MSE) {
            const float d = v - t;
            l = d * d;
            g = 2.f * d;
        } else if (kind == DL_LOSS_LINEAR) {
            l = t * v;                                   // GANLoss('wgangp'): -mean(pred) for real, +mean(pred) for fake; the sign is the "target"
            g = t;
        } else if (kind == DL_LOSS_L1) {
            const float d = v - t;                       // nn.L1Loss: |d|, gradient sign(d) with sign(0) = 0
            l = fabsf(d);
            g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        } else {
            const float d = v - t, ad = fabsf(d);
            l = ad < 1.f ? 0.5f * d * d : ad - 0.5f;
            g = ad < 1.f ? d : (d > 0.f ? 1.f : -1.f);
        }
        acc += l;
        if (grad) store1<T>(grad + p * g_ps + c, g * gscale * inv_count);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    // zero the padded channels of the gradient so downstream vector kernels see clean padding
    if (grad && Cp > C) {
        const size_t totalp = npix * (Cp - C);
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < totalp; i += (size_t)gridDim.x * blockDim.x) {
            const size_t p = i / (Cp - C);
            const int c = C + (int)(i % (Cp - C));
            store1<T>(grad + p * g_ps + c, 0.f);
        }
    }
}
__global__ void loss_final_kernel(const float *part, int n, float inv_count, float *out, float out_scale, int accumulate) {
    // 64 lanes, each the double sum of a contiguous slice of the partials, then lane 0 adds the 64 slice sums in lane order: a fixed summation
    // tree (deterministic), 6 us instead of the 20 us one thread needed for 512 dependent-latency loads (80 launches per step)
    __shared__ double sl[64];
    const int per = (n + 63) / 64;
    double s = 0.0;
    for (int i = threadIdx.x * per; i < min(n, (int)(threadIdx.x + 1) * per); ++i) s += (double)part[i];
    sl[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < 64; ++i) tot += sl[i];
        const float v = (float)(tot * (double)inv_count) * out_scale;
        out[0] = accumulate ? out[0] + v : v;
    }
}
extern "C" int dl_loss_acc(int kind, int dtype, const void *x, int x_ps, const void *target, int t_ps, float tconst, int64_t npix, int C, int Cp,
                           float *loss_out, float out_scale, int accumulate, void *grad, int g_ps, float gscale, float *ws, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !loss_out || !ws || C <= 0 || C > Cp) DL_FAIL("dl_loss: bad argument");
    if (kind < 0 || kind > 4) DL_FAIL("dl_loss: kind %d", kind);
    const float inv = 1.0f / (float)((double)npix * C);
    if (dtype == DL_F32) hipLaunchKernelGGL(loss_kernel<float>, dim3(LOSS_BLOCKS), dim3(256), 0, stream, kind, (const float *)x, x_ps, (const float *)target, t_ps, tconst, (size_t)npix, C, inv, (float *)grad, g_ps, gscale, Cp, ws);
    else hipLaunchKernelGGL(loss_kernel<bf16_t>, dim3(LOSS_BLOCKS), dim3(256), 0, stream, kind, (const bf16_t *)x, x_ps, (const bf16_t *)target, t_ps, tconst, (size_t)npix, C, inv, (bf16_t *)grad, g_ps, gscale, Cp, ws);
    DL_CHECK_LAUNCH("dl_loss");
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, stream, ws, LOSS_BLOCKS, inv, loss_out, out_scale, accumulate);
    DL_CHECK_LAUNCH("dl_loss(final)");
    return 0;
}
extern "C" int dl_loss(int kind, int dtype, const void *x, int x_ps, const void *target, int t_ps, float tconst, int64_t npix, int C, int Cp,
                       float *loss_out, void *grad, int g_ps, float gscale, float *ws, void *stream_) {
    return dl_loss_acc(kind, dtype, x, x_ps, target, t_ps, tconst, npix, C, Cp, loss_out, 1.0f, 0, grad, g_ps, gscale, ws, stream_);
}

// ------------------------------------------------------------------------------------------- KL divergence of two whole-tensor softmaxes
// DeepLIIFKD (DeepLIIFKD_model.py:313-336): KLDivLoss(reduction='batchmean')(LogSoftmax(dim=-1)(x.view(1, 1, -1)), Softmax(dim=-1)(t.view(1, 1, -1)))
//   = sum_j p_j (log p_j - log q_j),  p = softmax(t), q = softmax(x) over ALL real elements of the tensor (the batch dimension of the view is 1),
//   d/dx_j = q_j - p_j.
// With Zx = sum e^x, Zt = sum e^t, A = sum e^t (t - x):  KL = A / Zt - log Zt + log Zx.  Both tensors are generator outputs (tanh, or a
// convex combination of tanh outputs), so |x|, |t| <= 1 and the exponentials need no running maximum.  fp32 block partials in a fixed order, combined in
// double by one thread (deterministic); the gradient kernel reads Zx, Zt from device memory -- no host round trip.
// ws: 3 * LOSS_BLOCKS + 4 floats (dl_kldiv_ws_floats).
extern "C" size_t dl_kldiv_ws_floats(void) { return 3 * LOSS_BLOCKS + 4; }

template <typename T>
__global__ void __launch_bounds__(256) kldiv_partial_kernel(const T *x, int x_ps, const T *t, int t_ps, size_t npix, int C, float *part) {
    __shared__ float red[3][4];
    const size_t total = npix * C;
    float zx = 0.f, zt = 0.f, a = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / C;
        const int c = (int)(i % C);
        const float xv = load1<T>(x + p * x_ps + c), tv = load1<T>(t + p * t_ps + c);
        const float et = expf(tv);
        zx += expf(xv);
        zt += et;
        a += et * (tv - xv);
    }
    zx = wave_sum(zx); zt = wave_sum(zt); a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = zx; red[1][threadIdx.x >> 6] = zt; red[2][threadIdx.x >> 6] = a; }
    __syncthreads();
    if (threadIdx.x < 3) part[threadIdx.x * LOSS_BLOCKS + blockIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}
__global__ void kldiv_final_kernel(const float *part, float *stats, float *out, float out_scale, int accumulate) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double zx = 0.0, zt = 0.0, a = 0.0;
        for (int i = 0; i < LOSS_BLOCKS; ++i) { zx += (double)part[i]; zt += (double)part[LOSS_BLOCKS + i]; a += (double)part[2 * LOSS_BLOCKS + i]; }
        const float v = (float)(a / zt - log(zt) + log(zx)) * out_scale;
        out[0] = accumulate ? out[0] + v : v;
        stats[0] = (float)(1.0 / zx);
        stats[1] = (float)(1.0 / zt);
    }
}
template <typename T>
__global__ void __launch_bounds__(256) kldiv_grad_kernel(const T *x, int x_ps, const T *t, int t_ps, size_t npix, int C, int Cp, const float *stats,
                                                         T *grad, int g_ps, float gscale) {
    const float izx = stats[0], izt = stats[1];
    const size_t total = npix * Cp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / Cp;
        const int c = (int)(i % Cp);
        float g = 0.f;                                            // padded channels: clean zeros for the vector kernels downstream
        if (c < C) g = gscale * (expf(load1<T>(x + p * x_ps + c)) * izx - expf(load1<T>(t + p * t_ps + c)) * izt);
        store1<T>(grad + p * g_ps + c, g);
    }
}
extern "C" int dl_kldiv(int dtype, const void *x, int x_ps, const void *t, int t_ps, int64_t npix, int C, int Cp, float *loss_out, float out_scale,
                        int accumulate, void *grad, int g_ps, float gscale, float *ws, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !t || !loss_out || !ws || C <= 0 || C > Cp || npix <= 0) DL_FAIL("dl_kldiv: bad argument");
    if (dtype != DL_F32 && dtype != DL_BF16) DL_FAIL("dl_kldiv: dtype %d", dtype);
    float *stats = ws + 3 * LOSS_BLOCKS;
    if (dtype == DL_F32) hipLaunchKernelGGL(kldiv_partial_kernel<float>, dim3(LOSS_BLOCKS), dim3(256), 0, stream, (const float *)x, x_ps, (const float *)t, t_ps, (size_t)npix, C, ws);
    else hipLaunchKernelGGL(kldiv_partial_kernel<bf16_t>, dim3(LOSS_BLOCKS), dim3(256), 0, stream, (const bf16_t *)x, x_ps, (const bf16_t *)t, t_ps, (size_t)npix, C, ws);
    DL_CHECK_LAUNCH("dl_kldiv");
    hipLaunchKernelGGL(kldiv_final_kernel, dim3(1), dim3(64), 0, stream, ws, stats, loss_out, out_scale, accumulate);
    DL_CHECK_LAUNCH("dl_kldiv(final)");
    if (grad) {
        if (dtype == DL_F32) hipLaunchKernelGGL(kldiv_grad_kernel<float>, dim3(LOSS_BLOCKS), dim3(256), 0, stream, (const float *)x, x_ps, (const float *)t, t_ps, (size_t)npix, C, Cp, stats, (float *)grad, g_ps, gscale);
        else hipLaunchKernelGGL(kldiv_grad_kernel<bf16_t>, dim3(LOSS_BLOCKS), dim3(256), 0, stream, (const bf16_t *)x, x_ps, (const bf16_t *)t, t_ps, (size_t)npix, C, Cp, stats, (bf16_t *)grad, g_ps, gscale);
        DL_CHECK_LAUNCH("dl_kldiv(grad)");
    }
    return 0;
}

// ------------------------------------------------------------------------------------------- 2x2 max pooling (VGG19 features)
// nn.MaxPool2d(kernel_size=2, stride=2): y[n,ho,wo,c] = max over the 2x2 window; backward routes dy to the FIRST maximum of the window in
// row-major window order (ATen's rule), everything else gets 0 (including a trailing odd row / column).  One thread = one window x 8 channels.
template <typename T>
__global__ void maxpool2_fwd_kernel(const T *__restrict__ x, int x_ps, T *__restrict__ y, int y_ps, int N, int H, int W, int Cp) {
    const int Ho = H / 2, Wo = W / 2, cv = Cp / 8;
    const long long total = (long long)N * Ho * Wo * cv;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cv) * 8;
        long long p = i / cv;
        const int wo = (int)(p % Wo); p /= Wo;
        const int ho = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const T *b = x + (((long long)n * H + 2 * ho) * W + 2 * wo) * x_ps + c8;
        float a[8], q[8];
        Vec8<T>::load(b, a);
        Vec8<T>::load(b + x_ps, q);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = fmaxf(a[k], q[k]);
        Vec8<T>::load(b + (long long)W * x_ps, q);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = fmaxf(a[k], q[k]);
        Vec8<T>::load(b + (long long)W * x_ps + x_ps, q);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = fmaxf(a[k], q[k]);
        Vec8<T>::store(y + (((long long)n * Ho + ho) * Wo + wo) * y_ps + c8, a);
    }
}
template <typename T>
__global__ void maxpool2_bwd_kernel(const T *__restrict__ x, int x_ps, const T *__restrict__ dy, int dy_ps, T *__restrict__ dx, int dx_ps, int N, int H, int W, int Cp) {
    const int Hc = (H + 1) / 2, Wc = (W + 1) / 2, Ho = H / 2, Wo = W / 2, cv = Cp / 8;
    const long long total = (long long)N * Hc * Wc * cv;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cv) * 8;
        long long p = i / cv;
        const int wo = (int)(p % Wc); p /= Wc;
        const int ho = (int)(p % Hc);
        const int n = (int)(p / Hc);
        const bool full = ho < Ho && wo < Wo;
        float z[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) z[k] = 0.f;
        if (!full) {            // trailing odd row / column: no window covers it
            for (int dh = 0; dh < 2; ++dh)
                for (int dw = 0; dw < 2; ++dw) {
                    const int h = 2 * ho + dh, w = 2 * wo + dw;
                    if (h < H && w < W) Vec8<T>::store(dx + (((long long)n * H + h) * W + w) * dx_ps + c8, z);
                }
            continue;
        }
        const T *b = x + (((long long)n * H + 2 * ho) * W + 2 * wo) * x_ps + c8;
        float v[4][8], g[8], m[8];
        Vec8<T>::load(b, v[0]);
        Vec8<T>::load(b + x_ps, v[1]);
        Vec8<T>::load(b + (long long)W * x_ps, v[2]);
        Vec8<T>::load(b + (long long)W * x_ps + x_ps, v[3]);
        Vec8<T>::load(dy + (((long long)n * Ho + ho) * Wo + wo) * dy_ps + c8, g);
        int arg[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            m[k] = v[0][k]; arg[k] = 0;
#pragma unroll
            for (int j = 1; j < 4; ++j)
                if (v[j][k] > m[k]) { m[k] = v[j][k]; arg[k] = j; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = arg[k] == j ? g[k] : 0.f;
            Vec8<T>::store(dx + (((long long)n * H + 2 * ho + (j >> 1)) * W + 2 * wo + (j & 1)) * dx_ps + c8, o);
        }
    }
}
extern "C" int dl_maxpool2_forward(int dtype, const void *x, int x_ps, void *y, int y_ps, int N, int H, int W, int Cp, void *stream) {
    if (N <= 0 || H < 2 || W < 2 || Cp <= 0 || Cp % 8) DL_FAIL("dl_maxpool2_forward: empty problem or bad channel count (N=%d H=%d W=%d Cp=%d)", N, H, W, Cp);
    const long long total = (long long)N * (H / 2) * (W / 2) * (Cp / 8);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == DL_F32) hipLaunchKernelGGL(maxpool2_fwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float *)x, x_ps, (float *)y, y_ps, N, H, W, Cp);
    else hipLaunchKernelGGL(maxpool2_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)x, x_ps, (bf16_t *)y, y_ps, N, H, W, Cp);
    DL_CHECK_LAUNCH("dl_maxpool2_forward");
    return 0;
}
extern "C" int dl_maxpool2_backward(int dtype, const void *x, int x_ps, const void *dy, int dy_ps, void *dx, int dx_ps, int N, int H, int W, int Cp, void *stream) {
    if (N <= 0 || H < 2 || W < 2 || Cp <= 0 || Cp % 8) DL_FAIL("dl_maxpool2_backward: empty problem or bad channel count (N=%d H=%d W=%d Cp=%d)", N, H, W, Cp);
    const long long total = (long long)N * ((H + 1) / 2) * ((W + 1) / 2) * (Cp / 8);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == DL_F32) hipLaunchKernelGGL(maxpool2_bwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float *)x, x_ps, (const float *)dy, dy_ps, (float *)dx, dx_ps, N, H, W, Cp);
    else hipLaunchKernelGGL(maxpool2_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)x, x_ps, (const bf16_t *)dy, dy_ps, (bf16_t *)dx, dx_ps, N, H, W, Cp);
    DL_CHECK_LAUNCH("dl_maxpool2_backward");
    return 0;
}

// ------------------------------------------------------------------------------------------- Adam
// one element of the update.  Shared by the two kernels below with floating-point contraction OFF: they must round identically (a captured step replays
// adam_dev_kernel, the eager step runs adam_kernel; with contraction left to the compiler the two differed by one ulp in most parameters -- r04)
__device__ __forceinline__ void adam_update(float &p, float g, float &m, float &v, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, float gscale) {
#pragma clang fp contract(off)
    const float step_size = lr / bc1;
    const float gi = g * gscale;
    const float mi = b1 * m + (1.f - b1) * gi;
    const float vi = b2 * v + (1.f - b2) * gi * gi;
    m = mi;
    v = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p = p - step_size * (mi / denom);
}
__global__ void __launch_bounds__(256) adam_kernel(float *p, const float *g, float *m, float *v, size_t n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2_sqrt, float gscale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        adam_update(p[i], g[i], m[i], v[i], lr, b1, b2, eps, bc1, bc2_sqrt, gscale);
}
extern "C" int dl_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                            float beta2, float eps, int step, float grad_scale, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step < 1) DL_FAIL("dl_adam_step: bad argument");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_kernel, dim3(EW_BLOCKS(n)), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, (size_t)n, lr, beta1, beta2,
                       eps, bc1, sqrtf(bc2), grad_scale);
    DL_CHECK_LAUNCH("dl_adam_step");
    return 0;
}

// The same update with its per-step scalars read from DEVICE memory: a captured hipGraph of the training step replays fixed kernel arguments, while
// the learning rate (schedulers) and Adam's bias corrections change every step.  hyper = {lr, beta1, beta2, eps, 1 - beta1^t, sqrt(1 - beta2^t),
// grad_scale}, computed by dl_adam_hyper with the expressions of dl_adam_step and copied to the device outside the graph: identical arithmetic,
// bit-identical parameters.
__global__ void __launch_bounds__(256) adam_dev_kernel(float *p, const float *g, float *m, float *v, size_t n, const float *hyper) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], bc1 = hyper[4], bc2_sqrt = hyper[5], gscale = hyper[6];
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        adam_update(p[i], g[i], m[i], v[i], lr, b1, b2, eps, bc1, bc2_sqrt, gscale);
}
extern "C" int dl_adam_hyper(float lr, float beta1, float beta2, float eps, int step, float grad_scale, float *hyper_host) {
    if (!hyper_host || step < 1) DL_FAIL("dl_adam_hyper: bad argument");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = 1.f - powf(beta2, (float)step);
    hyper_host[0] = lr; hyper_host[1] = beta1; hyper_host[2] = beta2; hyper_host[3] = eps;
    hyper_host[4] = bc1; hyper_host[5] = sqrtf(bc2); hyper_host[6] = grad_scale; hyper_host[7] = 0.f;
    return 0;
}
extern "C" int dl_adam_step_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, const float *hyper_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !hyper_dev || n <= 0) DL_FAIL("dl_adam_step_dev: bad argument");
    hipLaunchKernelGGL(adam_dev_kernel, dim3(EW_BLOCKS(n)), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, (size_t)n, hyper_dev);
    DL_CHECK_LAUNCH("dl_adam_step_dev");
    return 0;
}

// ------------------------------------------------------------------------------------------- dropout
// nn.Dropout(0.5) of ResnetBlock / UnetSkipConnectionBlock (networks.py:493-494, 604-605): y = x * keep / (1 - p).  The keep mask
// is a counter-based hash of (seed, element index), so the backward pass regenerates it from the same seed instead of storing it.
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
template <typename T>
__global__ void __launch_bounds__(256) dropout_kernel(const T *x, int x_ps, T *y, int y_ps, size_t npix, int Cp, uint32_t seed, uint32_t thresh,
                                                      float scale) {
    const int cvec = Cp / 8;
    const size_t total = npix * cvec;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / cvec;
        const int c0 = (int)(i % cvec) * 8;
        float v[8];
        Vec8<T>::load(x + p * x_ps + c0, v);
        const uint32_t base = hash32((uint32_t)(i * 8) ^ (uint32_t)((i * 8) >> 32) * 0x9e3779b9U ^ seed);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (hash32(base + 0x9e3779b9U * (uint32_t)(k + 1)) >= thresh) ? v[k] * scale : 0.f;
        Vec8<T>::store(y + p * y_ps + c0, v);
    }
}
extern "C" int dl_dropout(int dtype, const void *x, int x_ps, void *y, int y_ps, int64_t npix, int Cp, float p, uint64_t seed, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !y || Cp % 8 || x_ps % 8 || y_ps % 8 || !(p >= 0.f && p < 1.f)) DL_FAIL("dl_dropout: bad argument");
    const size_t total = (size_t)npix * (Cp / 8);
    const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
    const uint32_t s32 = (uint32_t)seed ^ (uint32_t)(seed >> 32) * 0x85ebca6bU;
    if (dtype == DL_F32) hipLaunchKernelGGL(dropout_kernel<float>, dim3(EW_BLOCKS(total)), dim3(256), 0, stream, (const float *)x, x_ps, (float *)y, y_ps, (size_t)npix, Cp, s32, thresh, 1.f / (1.f - p));
    else hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(EW_BLOCKS(total)), dim3(256), 0, stream, (const bf16_t *)x, x_ps, (bf16_t *)y, y_ps, (size_t)npix, Cp, s32, thresh, 1.f / (1.f - p));
    DL_CHECK_LAUNCH("dl_dropout");
    return 0;
}

// ------------------------------------------------------------------------------------------- narrow-Cout conv helpers
__device__ __forceinline__ int reflect_w(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}
// y[n,h,w,co] = act(bias[co] + sum_kw T[n,h,w+kw-pad,co*KW+kw]); pad channels of y are zeroed.
// One block = SS_PIX consecutive pixels of one image row: the T rows (Tc fp32 channels per pixel) of those pixels plus the
// (KW-1) halo are staged through LDS with coalesced 16-byte loads, then each thread sums its pixel's Cout*KW taps.
#define SS_PIX 256
template <typename T>
__global__ void __launch_bounds__(256) shift_sum_kernel(const float *Tm, int N, int H, int W, int Tc, int Cout, int KW, int pad, int pad_mode,
                                                        const float *bias, int act, T *out, int o_ps, int oCp) {
    extern __shared__ __attribute__((aligned(16))) float tile[];      // [(SS_PIX + KW - 1)][Tc + 1]
    const int segs = (W + SS_PIX - 1) / SS_PIX;
    const int seg = blockIdx.x % segs;
    const size_t row = blockIdx.x / segs;                             // (n*H + h)
    const int w0 = seg * SS_PIX;
    const int span = min(SS_PIX, W - w0) + KW - 1;
    const int ld = Tc + 1;
    // stage: pixel (w0 - pad + i), i in [0, span)
    const int c4 = Tc / 4;
    for (int q = threadIdx.x; q < span * c4; q += blockDim.x) {
        const int i = q / c4, c = (q % c4) * 4;
        int ws = w0 - pad + i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        bool ok = (unsigned)ws < (unsigned)W;
        if (pad_mode == DL_PAD_REFLECT) { ws = reflect_w(ws, W); ok = true; }
        if (ok) v = *reinterpret_cast<const float4 *>(Tm + (row * W + ws) * Tc + c);
        float *d = tile + i * ld + c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    const int w = w0 + threadIdx.x;
    if (threadIdx.x < SS_PIX && w < W) {
        T *o = out + (row * W + w) * o_ps;
        for (int co = 0; co < Cout; ++co) {
            float s = bias ? bias[co] : 0.f;
            for (int kw = 0; kw < KW; ++kw) s += tile[(threadIdx.x + kw) * ld + co * KW + kw];
            store1<T>(o + co, apply_act(act, s));
        }
        for (int c = Cout; c < oCp; ++c) store1<T>(o + c, 0.f);
    }
}
extern "C" int dl_shift_sum(const float *Tm, int N, int H, int W, int Tc, int Cout, int KW, int pad, int pad_mode, const float *bias, int act,
                            int out_dtype, void *out, int o_ps, int oCp, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!Tm || !out || Cout * KW > Tc || Tc % 4) DL_FAIL("dl_shift_sum: bad argument");
    const int segs = (W + SS_PIX - 1) / SS_PIX;
    const size_t blocks = (size_t)N * H * segs;
    const size_t smem = (size_t)(SS_PIX + KW - 1) * (Tc + 1) * sizeof(float);
    if (out_dtype == DL_F32) hipLaunchKernelGGL(shift_sum_kernel<float>, dim3((unsigned)blocks), dim3(256), smem, stream, Tm, N, H, W, Tc, Cout, KW, pad, pad_mode, bias, act, (float *)out, o_ps, oCp);
    else hipLaunchKernelGGL(shift_sum_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), smem, stream, Tm, N, H, W, Tc, Cout, KW, pad, pad_mode, bias, act, (bf16_t *)out, o_ps, oCp);
    DL_CHECK_LAUNCH("dl_shift_sum");
    return 0;
}
// Narrow-Cout transposed conv (the UNet's outermost ConvTranspose2d(2*ngf, 3, k=4, s=2, p=1) + Tanh, networks.py:573-576): every input
// pixel feeds 4 x 4 output positions, so the gather GEMM (4 sub-pixel phases x 4 taps) stages each input pixel 16 times for 3 useful output
// columns: 327 us at 8 x 256^2 x 128 -> 512^2.  Instead T[n,y,x,(ky*4+kx)*Cout+co] = sum_ci x[n,y,x,ci] * W[ci,co,ky,kx] is ONE plain GEMM over the
// input (dl_conv_forward(raw_out) with a 1 x 1 plan: the input is staged once) and this kernel sums the 2 x 2 contributions of every output pixel:
//   out[n,oy,ox,co] = act(bias[co] + sum_{ky,kx : (oy+1-ky), (ox+1-kx) even, in range} T[n,(oy+1-ky)/2,(ox+1-kx)/2,(ky*4+kx)*Cout+co])
template <typename T>
__global__ void __launch_bounds__(256) convt4_gather_kernel(const float *Tm, int N, int H, int W, int Tc, int Cout, const float *bias, int act,
                                                            T *out, int o_ps, int oCp) {
    const int Ho = 2 * H, Wo = 2 * W;
    const size_t npix = (size_t)N * Ho * Wo;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), n = (int)(p / ((size_t)Wo * Ho));
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int ky = ((oy + 1) & 1) + 2 * a;                  // the two kernel rows with (oy + 1 - ky) even
            const int y = (oy + 1 - ky) >> 1;
            if ((unsigned)y >= (unsigned)H) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int kx = ((ox + 1) & 1) + 2 * b;
                const int x = (ox + 1 - kx) >> 1;
                if ((unsigned)x >= (unsigned)W) continue;
                const float *t = Tm + (((size_t)n * H + y) * W + x) * Tc + (ky * 4 + kx) * Cout;
                for (int c = 0; c < Cout; ++c) acc[c] += t[c];
            }
        }
        T *o = out + p * o_ps;
        for (int c = 0; c < Cout; ++c) store1<T>(o + c, apply_act(act, acc[c] + (bias ? bias[c] : 0.f)));
        for (int c = Cout; c < oCp; ++c) store1<T>(o + c, 0.f);
    }
}
extern "C" int dl_convt4_gather(const float *Tm, int N, int H, int W, int Tc, int Cout, const float *bias, int act, int out_dtype, void *out,
                                int o_ps, int oCp, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!Tm || !out || N <= 0 || H <= 0 || W <= 0 || Cout < 1 || Cout > 4 || 16 * Cout > Tc || Cout > oCp) DL_FAIL("dl_convt4_gather: bad argument");
    const size_t npix = (size_t)N * 4 * H * W;
    if (out_dtype == DL_F32) hipLaunchKernelGGL(convt4_gather_kernel<float>, dim3(EW_BLOCKS(npix)), dim3(256), 0, stream, Tm, N, H, W, Tc, Cout, bias, act, (float *)out, o_ps, oCp);
    else hipLaunchKernelGGL(convt4_gather_kernel<bf16_t>, dim3(EW_BLOCKS(npix)), dim3(256), 0, stream, Tm, N, H, W, Tc, Cout, bias, act, (bf16_t *)out, o_ps, oCp);
    DL_CHECK_LAUNCH("dl_convt4_gather");
    return 0;
}
// D[n,h,w,co*KW+kw] = dy[n,h,w-(kw-pad),co]  (zero outside; zero padding only).  One thread per pixel builds its whole
// Dc-channel row in registers and writes it with 16-byte stores.
template <typename T>
__global__ void __launch_bounds__(256) shift_stack_kernel(const T *dy, int dy_ps, int N, int H, int W, int Cout, int KW, int pad, T *D, int Dc) {
    const size_t npix = (size_t)N * H * W;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(p % W);
        const size_t rowbase = p - w;
        T *d = D + p * Dc;
        for (int c0 = 0; c0 < Dc; c0 += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = c0 + k;
                const int co = c / KW, kw = c - co * KW;
                const int ws = w - (kw - pad);
                v[k] = (co < Cout && (unsigned)ws < (unsigned)W) ? load1<T>(dy + (rowbase + ws) * dy_ps + co) : 0.f;
            }
            Vec8<T>::store(d + c0, v);
        }
    }
}
extern "C" int dl_shift_stack(int dtype, const void *dy, int dy_ps, int N, int H, int W, int Cout, int KW, int pad, int pad_mode, void *D, int Dc,
                              void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!dy || !D || Cout * KW > Dc || Dc % 8 || pad_mode != DL_PAD_ZERO) DL_FAIL("dl_shift_stack: bad argument (zero padding only)");
    const size_t npix = (size_t)N * H * W;
    if (dtype == DL_F32) hipLaunchKernelGGL(shift_stack_kernel<float>, dim3(EW_BLOCKS(npix)), dim3(256), 0, stream, (const float *)dy, dy_ps, N, H, W, Cout, KW, pad, (float *)D, Dc);
    else hipLaunchKernelGGL(shift_stack_kernel<bf16_t>, dim3(EW_BLOCKS(npix)), dim3(256), 0, stream, (const bf16_t *)dy, dy_ps, N, H, W, Cout, KW, pad, (bf16_t *)D, Dc);
    DL_CHECK_LAUNCH("dl_shift_stack");
    return 0;
}

// ------------------------------------------------------------------------------------------- probes
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// D[16][16] = A[16][32] * B[32][16] with the fragment mapping the GEMM kernels assume
__global__ void probe_mfma16_kernel(const bf16_t *A, const bf16_t *B, float *D) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    bf16x8_t a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = (short)A[r * 32 + g * 8 + j];          // A[row = r][k = 8g + j]
        b[j] = (short)B[(g * 8 + j) * 16 + r];        // B[k = 8g + j][col = r]
    }
    f32x4_t c = {0.f, 0.f, 0.f, 0.f};
    c = dl_mfma16(a, b, c);
#pragma unroll
    for (int i = 0; i < 4; ++i) D[(g * 4 + i) * 16 + r] = c[i];   // row = 4g + i, col = r
}
extern "C" int dl_probe_mfma16(const uint16_t *a, const uint16_t *b, float *d, void *stream_) {
    hipLaunchKernelGGL(probe_mfma16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, a, b, d);
    DL_CHECK_LAUNCH("dl_probe_mfma16");
    return 0;
}
// src: 64 rows x 16 cols bf16 (row-major).  Lane (m = lane&15, g = lane>>4) issues ds_read_b64_tr_b16 at
// &tile[4g + (m>>2)][(m&3)*4]; dst[lane][j] receives the four returned elements.
__global__ void probe_trread_kernel(const bf16_t *src, bf16_t *dst) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64 * 16];
    for (int i = threadIdx.x; i < 64 * 16; i += 64) tile[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x, m = lane & 15, g = lane >> 4;
    const bf16_t *p = tile + (4 * g + (m >> 2)) * 16 + (m & 3) * 4;
    const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3))) *)p);
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[lane * 4 + j] = (bf16_t)v[j];
}
extern "C" int dl_probe_trread(const uint16_t *src, uint16_t *dst, void *stream_) {
    hipLaunchKernelGGL(probe_trread_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, src, dst);
    DL_CHECK_LAUNCH("dl_probe_trread");
    return 0;
}

// ------------------------------------------------------------------------------------------- reflection-pad backward (fold)
// dst[n,h,w,:] = sum of src[n,hp,wp,:] over every position of the reflection-padded extent (H+2p) x (W+2p) that
// nn.ReflectionPad2d(p) fills from (h,w): the interior copy (h+p, w+p) plus the mirrored border rows / columns
// (padded index p-h for 1 <= h <= p, p+2H-2-h for H-1-p <= h <= H-2; same along w).  At most 3 x 3 sources, summed in a fixed
// order in fp32.  src = the data gradient with respect to the explicitly padded input (dl_conv_forward with the pad-0 plan).
template <typename T>
__global__ void __launch_bounds__(256) reflect_fold_kernel(const T *src, int s_ps, T *dst, int d_ps, int N, int H, int W, int p, int Cp) {
    const int cvec = Cp / 8;
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    const size_t total = (size_t)N * H * W * cvec;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cvec) * 8;
        size_t pix = i / cvec;
        const int w = (int)(pix % W);
        pix /= W;
        const int h = (int)(pix % H), n = (int)(pix / H);
        int hs[3], ws[3], nh = 0, nw = 0;
        hs[nh++] = h + p;
        if (h >= 1 && h <= p) hs[nh++] = p - h;
        if (h <= H - 2 && h >= H - 1 - p) hs[nh++] = p + 2 * H - 2 - h;
        ws[nw++] = w + p;
        if (w >= 1 && w <= p) ws[nw++] = p - w;
        if (w <= W - 2 && w >= W - 1 - p) ws[nw++] = p + 2 * W - 2 - w;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < nh; ++a)
            for (int b = 0; b < nw; ++b) {
                float v[8];
                Vec8<T>::load(src + (((size_t)n * Hp + hs[a]) * Wp + ws[b]) * s_ps + c0, v);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += v[k];
            }
        Vec8<T>::store(dst + (((size_t)n * H + h) * W + w) * d_ps + c0, acc);
    }
}
extern "C" int dl_reflect_fold(int dtype, const void *src, int src_pstride, void *dst, int dst_pstride, int N, int H, int W, int pad, int Cp,
                               void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N <= 0 || H <= 0 || W <= 0) DL_FAIL("dl_reflect_fold: empty problem (N=%d, %dx%d): nothing to launch", N, H, W);
    if (!src || !dst || Cp % 8 || src_pstride % 8 || dst_pstride % 8) DL_FAIL("dl_reflect_fold: bad argument");
    if (pad < 1 || pad >= H || pad >= W) DL_FAIL("dl_reflect_fold: pad=%d must be in [1, min(H, W) - 1] (nn.ReflectionPad2d requires pad < size)", pad);
    const size_t total = (size_t)N * H * W * (Cp / 8);
    if (dtype == DL_F32) hipLaunchKernelGGL(reflect_fold_kernel<float>, dim3(EW_BLOCKS(total)), dim3(256), 0, stream, (const float *)src, src_pstride, (float *)dst, dst_pstride, N, H, W, pad, Cp);
    else if (dtype == DL_BF16) hipLaunchKernelGGL(reflect_fold_kernel<bf16_t>, dim3(EW_BLOCKS(total)), dim3(256), 0, stream, (const bf16_t *)src, src_pstride, (bf16_t *)dst, dst_pstride, N, H, W, pad, Cp);
    else DL_FAIL("dl_reflect_fold: dtype %d", dtype);
    DL_CHECK_LAUNCH("dl_reflect_fold");
    return 0;
}
