// conv_dot.hip -- the PatchGAN's one-channel prediction layer, Conv2d(512, 1, k4, s1, p1) (NLayerDiscriminator, networks.py:655-660): forward and data gradient.
//
// 0.12 GF over 8 MB: not a GEMM.  The gather GEMM ran it on its narrowest tiles (conv_gemm_glds_kernel<256,16,32> with an 18-way split-K + slab reduction: 40 us
// forward; <128,128,64> for the data gradient: 26 us) -- 29 workgroups of latency.  Here a WAVE owns an output pixel and a LANE 8 of the 512 channels:
//   forward        out[n,ho,wo] = bias + sum_t x[n, ho+dh_t, wo+dw_t, :] . w_t       16 coalesced 1 KB loads in flight per wave, 128 FMAs per lane, one wave butterfly;
//   data gradient  dx[n,h,w,:]  = sum_t dy[n, h+dh_t, w+dw_t] * w_t                  16 broadcast loads of one value, 128 FMAs per lane, one coalesced 1 KB store.
// The lane's weights (16 taps x 8 channels) stay in registers as fp32; waves are persistent over the pixels (grid = 4 waves x 1024 workgroups at most).
// Same descriptors and packed weight images as the gather GEMM (forward: row 0 of the image; data gradient: column tap*8 of every row): no host change.
#include "conv_args.h"

struct DotArgs {
    const bf16_t *in;
    const bf16_t *w;
    const float *bias;
    bf16_t *out;
    int N, Hi, Wi, in_ps, Ho, Wo, out_ps, ntaps, w_kstride, act, npix;
    int8_t dh[16], dw[16];
};

__device__ __forceinline__ float dot_act(int act, float v) { return act == DL_ACT_LRELU ? (v > 0.f ? v : 0.2f * v) : (act == DL_ACT_RELU ? (v > 0.f ? v : 0.f) : v); }

// forward: 512 contracted channels (lane l: channels 8l .. 8l+7), ONE real output channel written into an 8-channel padded pixel
__global__ void __launch_bounds__(256) conv_dot_fwd_kernel(const DotArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float w[16][8];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        if (t < a.ntaps) Vec8<bf16_t>::load(a.w + t * 512 + lane * 8, w[t]);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) w[t][e] = 0.f;
        }
    }
    const float bias = a.bias ? a.bias[0] : 0.f;
    const int HW = a.Ho * a.Wo;
    for (int p = blockIdx.x * 4 + wave; p < a.npix; p += gridDim.x * 4) {
        const int n = p / HW, rem = p - n * HW;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        u32x4_t x[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int hi = ho + a.dh[t], wi = wo + a.dw[t];
            x[t] = u32x4_t{0u, 0u, 0u, 0u};
            if (t < a.ntaps && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi)
                x[t] = *reinterpret_cast<const u32x4_t *>(a.in + ((size_t)(n * a.Hi + hi) * a.Wi + wi) * a.in_ps + lane * 8);
        }
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc += h16_lo_f32(x[t][e]) * w[t][2 * e];
                acc += h16_hi_f32(x[t][e]) * w[t][2 * e + 1];
            }
        acc = wave_sum(acc);
        if (lane == 0) {
            u32x4_t o = u32x4_t{0u, 0u, 0u, 0u};
            o[0] = pack2_bf16(dot_act(a.act, acc + bias), 0.f);
            *reinterpret_cast<u32x4_t *>(a.out + (size_t)p * a.out_ps) = o;
        }
    }
}

// data gradient: the gradient tensor has ONE real channel (in an 8-channel padded pixel), 512 output channels (lane l: 8l .. 8l+7).
// The packed image has one ROW per output channel and the real value of tap t at column 8t: gathered lane by lane that is 128 two-byte loads from 8 different
// rows (first version: 68 us, slower than the GEMM it replaced).  The workgroup reads the 512 rows once (every thread two rows of 256 B), drops the
// seven padding columns on the way into LDS ([tap][channel] fp32, 32 KB), and every lane then takes its 16 x 8 weights from there.
__global__ void __launch_bounds__(256) conv_dot_dgrad_kernel(const DotArgs a) {
    __shared__ float wl[16 * 512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = threadIdx.x; r < 512; r += 256) {
        const bf16_t *row = a.w + (size_t)r * a.w_kstride;
#pragma unroll
        for (int t = 0; t < 16; ++t) wl[t * 512 + r] = t < a.ntaps ? bf16_to_f32(row[t * 8]) : 0.f;
    }
    __syncthreads();
    float w[16][8];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) w[t][e] = wl[t * 512 + lane * 8 + e];
    const int HW = a.Ho * a.Wo;
    for (int p = blockIdx.x * 4 + wave; p < a.npix; p += gridDim.x * 4) {
        const int n = p / HW, rem = p - n * HW;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        float g[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int hi = ho + a.dh[t], wi = wo + a.dw[t];
            g[t] = (t < a.ntaps && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi) ? bf16_to_f32(a.in[((size_t)(n * a.Hi + hi) * a.Wi + wi) * a.in_ps]) : 0.f;
        }
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += g[t] * w[t][e];
        Vec8<bf16_t>::store(a.out + (size_t)p * a.out_ps + lane * 8, v);
    }
}

static void dot_fill(DotArgs &da, const ConvArgs &a) {
    memset(&da, 0, sizeof(da));
    da.in = reinterpret_cast<const bf16_t *>(a.in);
    da.w = a.w_hi + a.phase_kbase[0];
    da.bias = a.bias;
    da.out = reinterpret_cast<bf16_t *>(a.out);
    da.N = a.N; da.Hi = a.Hi; da.Wi = a.Wi; da.in_ps = a.in_pstride; da.Ho = a.Ho; da.Wo = a.Wo; da.out_ps = a.out_pstride;
    da.ntaps = a.phase_tap_begin[1] - a.phase_tap_begin[0];
    da.w_kstride = a.w_kstride; da.act = a.act; da.npix = a.N * a.Ho * a.Wo;
    for (int t = 0; t < da.ntaps; ++t) {
        da.dh[t] = (int8_t)(a.taps[a.phase_tap_begin[0] + t] & 0xff);
        da.dw[t] = (int8_t)((a.taps[a.phase_tap_begin[0] + t] >> 8) & 0xff);
    }
}

// common shape: one phase, stride 1 both ways, at most 16 taps, zero padding, bf16, no input activation, epilogue activation none / ReLU / LeakyReLU
static bool dot_common(const ConvArgs &a) {
    const int nt = a.phase_tap_begin[1] - a.phase_tap_begin[0];
    return a.n_phase == 1 && a.in_step == 1 && a.out_step == 1 && !a.raw_out && nt >= 1 && nt <= 16 && a.pad_mode == DL_PAD_ZERO && a.in_act == DL_ACT_NONE &&
           a.bn_y == nullptr && a.Ho == a.Hq && a.Wo == a.Wq && (a.act == DL_ACT_NONE || a.act == DL_ACT_RELU || a.act == DL_ACT_LRELU);
}
// forward: 512 contracted channels, ONE real output channel (bias_n == 1 says so: the rows 1..7 of the padded pixel are zero weights + no bias)
bool dot_fwd_eligible(const ConvArgs &a) { return dot_common(a) && a.Ci == 512 && a.Co == 8 && a.bias != nullptr && a.bias_n == 1; }
// data gradient: ONE real contracted channel (dl_conv_desc.ci_real == 1, checked by the caller), 512 output channels, no bias
bool dot_dgrad_eligible(const ConvArgs &a) { return dot_common(a) && a.Ci == 8 && a.Co == 512 && a.bias == nullptr && a.act == DL_ACT_NONE; }

int launch_conv_dot(const ConvArgs &a, bool fwd, hipStream_t stream) {
    DotArgs da;
    dot_fill(da, a);
    // forward: up to 1024 workgroups of 4 pixel-waves; data gradient: every workgroup first reads the 128 KB weight image -> 256 workgroups, ~8 pixels per wave
    const int wgs = min(fwd ? 1024 : 256, (da.npix + 3) / 4);
    if (fwd) hipLaunchKernelGGL(conv_dot_fwd_kernel, dim3(wgs), dim3(256), 0, stream, da);
    else hipLaunchKernelGGL(conv_dot_dgrad_kernel, dim3(wgs), dim3(256), 0, stream, da);
    DL_CHECK_LAUNCH(fwd ? "dl_conv_forward(dot fwd)" : "dl_conv_forward(dot dgrad)");
    return 0;
}
