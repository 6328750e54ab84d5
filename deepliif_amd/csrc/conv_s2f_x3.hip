// conv_s2f_x3.hip -- conv_s2f_kernel (conv_s2f.hip: the FUSED four-phase tile of the stride-2 layers) for the STRICT policy: fp32 storage, split-bf16 x3
// products (lo_w hi_x + hi_w lo_x + hi_w hi_x), the policy the GPU tests hold to 1e-3 against the reference.  Same layers -- ConvTranspose2d(k3, s2, p1, op1)
// forward (ResnetGenerator up1 / up2, networks.py:400-436), the data gradient of Conv2d(k3 | k4, s2, p1) (down1 / down2, NLayerDiscriminator c2-c4,
// networks.py:576-609 / 638-660) --, same tile (256 phase-grid pixels x 64 output channels x ALL four phases in one workgroup, 8 waves as 4 pixel groups x
// 2 channel groups), same descriptor and packed weights as the 4-phase strict path (conv_gemm_glds_x3_kernel<128, 64>, one workgroup per (tile, PHASE): every
// phase re-stages the input tile a whole grid later -- 549 us on up2, 141 TF/s).
//   * K step = 32 channels: the activations are the producer-written SPLIT COPY (dl_conv_desc.in_split: every group of 8 channels = [8 hi bf16 | 8 lo bf16]),
//     so a pixel row of 32 channels is 128 B = the bf16 kernel's 64 channels and the DMA geometry, the 32 KB input tile per (chunk, DISTINCT offset), the
//     8 KB weight tile per tap (a row = [32 hi | 32 lo] gathered from the two packed images by per-lane source pointers) and the double buffer carry over;
//   * a macro step = two K=16 sub-steps; per phase that has a tap at the offset: 2 sub-steps x 2 pixel blocks x 3 products = 12 MFMAs (32x32x16) per wave;
//   * epilogue: fp32 results leave through LDS in two passes (phases 0-1, then 2-3: [256 px][2 x 64 ch] fp32 = 128 KB, 16-byte chunks XOR-swizzled by the
//     pixel) as 256-byte pieces to the output pixels of each phase; bias / ReLU; fused per-(image, channel) statistics over pixels AND phases.
// Inputs that are not split copies stay on the 4-phase kernel.
#include "conv_args.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((address_space(3))) char lds_char_t;
typedef __attribute__((address_space(3))) const bf16x8_t lds_frag_t;

__device__ __attribute__((aligned(64))) unsigned char g_s2fx3_zero_page[64];

template <int V> struct S3IC { static constexpr int value = V; };

constexpr int S3_MAX_OFF = 9;
struct S2fX3Args {
    ConvArgs a;
    int n_off;
    int off_delta[S3_MAX_OFF];             // element offset (dh * Wi + dw) * in_pstride of the offset (4-byte elements)
    int8_t off_dh[S3_MAX_OFF], off_dw[S3_MAX_OFF];
    int user_k[S3_MAX_OFF][4];             // per offset and phase: weight column base (kbase_p + tap_index * Ci) or -1
};

__device__ __forceinline__ float s3_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
    return v;
}

constexpr int S3_XB = 256 * 128;               // input tile of one (32-channel chunk, offset): 256 pixels x [4 groups x (8 hi | 8 lo)]
constexpr int S3_WB = 64 * 128;                // weight tile of one tap: 64 output channels x [32 hi | 32 lo]
constexpr int S3_BUF = S3_XB + 4 * S3_WB;      // one pipeline buffer: the input tile + up to four taps' weights
constexpr size_t S3_LOOP_LDS = (size_t)2 * S3_BUF;
constexpr size_t S3_EPI_LDS = (size_t)256 * 512 + 256 * sizeof(int) + (size_t)4 * 2 * 64 * sizeof(float);
constexpr size_t S3_LDS = S3_LOOP_LDS > S3_EPI_LDS ? S3_LOOP_LDS : S3_EPI_LDS;
static_assert(S3_LDS <= 160 * 1024, "the whole LDS of a CU");

__global__ void __launch_bounds__(512) conv_s2f_x3_kernel(const S2fX3Args sa) {
    const ConvArgs &a = sa.a;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_char_t *lds = (lds_char_t *)smem_raw;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;                // 64-pixel group, 32-channel group
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % a.tiles_n, tm = bid / a.tiles_n;
    const int nch = a.Ci >> 5;                               // 32-channel chunks
    const int NO = sa.n_off;
    const int T = nch * NO;                                  // macro steps: (chunk, offset), offset fastest

    // ---- staging geometry (as conv_s2f_kernel, byte addresses): LDS rows are 128 B = 8 chunks of 16 B, chunk c of row r at position c ^ ((r >> 1) & 7).
    // activations: chunk 2g + plane of a row = plane (0 hi, 1 lo) of channel group g -- the split copy's own order;  weights: chunk 4*plane + k8
    const int lrow = lane >> 3, lcp = lane & 7;
    const int HWq = a.Hq * a.Wq;
    const char *in = reinterpret_cast<const char *>(a.in);
    const char *zero = reinterpret_cast<const char *>(g_s2fx3_zero_page);
    const char *x_ptr[4];
    unsigned x_mask[4];                                      // bit o: offset o of this pixel lies inside the image
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = (wave * 4 + i) * 8 + lrow;
        const int m = tm * 256 + s;
        const bool ok = m < a.Mtot;
        const int mm = ok ? m : 0;
        const int n = mm / HWq, rem = mm - n * HWq;
        const int hq = rem / a.Wq, wq = rem - hq * a.Wq;
        x_ptr[i] = in + ((size_t)(n * a.Hi + hq) * a.Wi + wq) * (size_t)a.in_pstride * 4 + (lcp ^ ((s >> 1) & 7)) * 16;
        unsigned mk = 0;
        if (ok)
            for (int o = 0; o < NO; ++o) {
                const int hi = hq + sa.off_dh[o], wi = wq + sa.off_dw[o];
                if (((unsigned)hi < (unsigned)a.Hi) && ((unsigned)wi < (unsigned)a.Wi)) mk |= 1u << o;
            }
        x_mask[i] = mk;
    }
    const int wrow = wave * 8 + lrow;                        // output channel inside the 64-channel tile this lane stages
    const int wcl = lcp ^ ((wrow >> 1) & 7);                 // logical chunk of the weight row this lane fetches: plane wcl >> 2, K columns (wcl & 3) * 8 ..
    const char *w_ptr = reinterpret_cast<const char *>((wcl & 4) ? a.w_lo : a.w_hi) + ((size_t)(tn * 64 + wrow) * a.w_kstride + (wcl & 3) * 8) * 2;

    auto stage = [&](int u) __attribute__((always_inline)) {
        const int c = u / NO, o = u - c * NO;
        const int buf = (u & 1) * S3_BUF;
        const ptrdiff_t xd = (ptrdiff_t)sa.off_delta[o] * 4 + c * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = (x_mask[i] >> o) & 1u;
            const char *src = ok ? x_ptr[i] + xd : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(lds + buf + (wave * 4 + i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int kb = sa.user_k[o][p];
            if (kb >= 0)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(w_ptr + (ptrdiff_t)(kb + c * 32) * 2),
                                                 (__attribute__((address_space(3))) void *)(lds + buf + S3_XB + p * S3_WB + wave * 1024), 16, 0, 0);
        }
    };

    // ---- fragment addressing (bytes): lane = (row lr of a 32-row block, K half lh of a 16-wide sub-step s)
    //   activations: logical chunk 4s + 2lh + plane -> ((2lh ^ swz) << 4) ^ (s << 6) ^ (plane << 4);   weights: 4 plane + 2s + lh -> ((lh ^ swz) << 4) ^ (s << 5) ^ (plane << 6)
    const int lr = lane & 31, lh = lane >> 5;
    const int swz = (lr >> 1) & 7;
    const int ax = (wm * 64 + lr) * 128 + (((2 * lh) ^ swz) << 4);                        // + j * 4096 (pixel block)
    const int aw = S3_XB + (wn * 32 + lr) * 128 + ((lh ^ swz) << 4);                      // + p * S3_WB

    f32x16_t acc[4][2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][j][r] = 0.f;

    if (T > 0) stage(0);
    for (int u = 0; u < T; ++u) {
        // my pieces of step u have landed; after the barrier everybody's have, and everybody is done reading buffer (u+1)&1 (step u-1)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (u + 1 < T) stage(u + 1);
        const int c = u / NO, o = u - c * NO;
        (void)c;
        const int buf = (u & 1) * S3_BUF;
        bf16x8_t xh[2][2], xl[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                xh[s][j] = *reinterpret_cast<lds_frag_t *>(lds + buf + (ax ^ (s << 6)) + j * 4096);
                xl[s][j] = *reinterpret_cast<lds_frag_t *>(lds + buf + (ax ^ (s << 6) ^ 16) + j * 4096);
            }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (sa.user_k[o][p] < 0) continue;               // wave-uniform (kernel argument)
            bf16x8_t wh[2], wl[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                wh[s] = *reinterpret_cast<lds_frag_t *>(lds + buf + p * S3_WB + (aw ^ (s << 5)));
                wl[s] = *reinterpret_cast<lds_frag_t *>(lds + buf + p * S3_WB + (aw ^ (s << 5) ^ 64));
            }
            // term-major, small terms first: lo_w hi_x, hi_w lo_x, hi_w hi_x (the order of conv_x3.h)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[p][j] = dl_mfma32(wl[s], xh[s][j], acc[p][j]);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[p][j] = dl_mfma32(wh[s], xl[s][j], acc[p][j]);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[p][j] = dl_mfma32(wh[s], xh[s][j], acc[p][j]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();             // LDS is dead from here on (the epilogue reuses it)

    // ---- epilogue.  acc[p][j][r] = out phase p, pixel = wm*64 + j*32 + lr, channel = wn*32 + 8*(r>>2) + 4*lh + (r&3), fp32.
    // Two passes (phases 2H, 2H + 1): [256 pixels][2 phases x 64 channels] fp32 through LDS (512-byte rows, 16-byte chunk c of row r at c ^ (r & 31):
    // the 32 lanes of a fragment column write 32 different chunks), then 256-byte pieces (one phase of one pixel) out, 16 B per lane.
    lds_char_t *tile = lds;
    __attribute__((address_space(3))) int *rowtab = reinterpret_cast<__attribute__((address_space(3))) int *>(lds + 256 * 512);
    __attribute__((address_space(3))) float *red = reinterpret_cast<__attribute__((address_space(3))) float *>(lds + 256 * 512 + 256 * 4);       // [wm][2][64]
    const bool want_stats = a.stats_part != nullptr;
    if (tid < 256) {
        const int m = tm * 256 + tid;
        int opix = -1;
        if (m < a.Mtot) {
            const int n = m / HWq, rem = m - n * HWq;
            const int hq = rem / a.Wq, wq = rem - hq * a.Wq;
            opix = (n * a.Ho + 2 * hq) * a.Wo + 2 * wq;      // phase (0, 0) pixel; phase (oh, ow) adds oh * Wo + ow
        }
        rowtab[tid] = opix;
    }
    bool live[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) live[j] = tm * 256 + wm * 64 + j * 32 + lr < a.Mtot;
    float *out = reinterpret_cast<float *>(a.out);

    auto pass = [&](auto HH) __attribute__((always_inline)) {
        constexpr int H = decltype(HH)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cl = wn * 32 + q * 8 + lh * 4;             // channel inside the 64-channel tile
            const int co = tn * 64 + cl;
            float bias[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bias[e] = (co + e < a.bias_n) ? a.bias[co + e] : 0.f;
            }
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4_t v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[2 * H + pp][j][q * 4 + e] + bias[e];
                    if (a.act == DL_ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    }
                    const int row = wm * 64 + j * 32 + lr;
                    *reinterpret_cast<__attribute__((address_space(3))) f32x4_t *>(tile + row * 512 + ((((pp * 64 + cl) >> 2) ^ (row & 31)) << 4)) = v;
                    if (want_stats && live[j]) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
                    }
                }
            }
            if (want_stats) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1[e] = s3_row16_sum(s1[e]); s2[e] = s3_row16_sum(s2[e]);
                    s1[e] += __shfl_xor(s1[e], 16, 64); s2[e] += __shfl_xor(s2[e], 16, 64);
                }
                if (lr == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (H == 0) {
                            red[(wm * 2 + 0) * 64 + cl + e] = s1[e];
                            red[(wm * 2 + 1) * 64 + cl + e] = s2[e];
                        } else {       // same lane, same address as in the first pass
                            red[(wm * 2 + 0) * 64 + cl + e] += s1[e];
                            red[(wm * 2 + 1) * 64 + cl + e] += s2[e];
                        }
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int idx = tid; idx < 256 * 32; idx += 512) {
            const int row = idx >> 5, cc = idx & 31;             // 16-byte chunk cc of the row: phase 2H + (cc >> 4), channels (cc & 15) * 4 ..
            const int opix = rowtab[row];
            const int p = 2 * H + (cc >> 4), co = tn * 64 + (cc & 15) * 4;
            if (opix < 0 || co >= a.Co) continue;
            const f32x4_t v = *reinterpret_cast<__attribute__((address_space(3))) const f32x4_t *>(tile + row * 512 + ((cc ^ (row & 31)) << 4));
            *reinterpret_cast<f32x4_t *>(out + ((size_t)opix + (size_t)(p >> 1) * a.Wo + (p & 1)) * a.out_pstride + co) = v;
        }
        __syncthreads();
    };
    pass(S3IC<0>{});
    pass(S3IC<1>{});

    if (want_stats && tid < 64) {
        // every pixel of this tile lies in ONE image (s2f_stats_chunks): one chunk per tile
        const int m0 = tm * 256;
        const int n = m0 / HWq;
        const int chunk = (m0 - n * HWq) >> 8;
        const int co = tn * 64 + tid;
        if (co < a.Co) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { t1 += red[(w * 2 + 0) * 64 + tid]; t2 += red[(w * 2 + 1) * 64 + tid]; }
            float *o = a.stats_part + ((size_t)(n * a.stats_nchunks + chunk) * 2) * a.Co + co;
            o[0] = t1;
            o[a.Co] = t2;
        }
    }
}

static bool s3_phase_order_ok(const ConvArgs &a) {
    for (int p = 0; p < 4; ++p)
        if (a.phase_oh[p] != (p >> 1) || a.phase_ow[p] != (p & 1)) return false;
    return true;
}

// The layers this kernel serves: conv_s2f_kernel's (s2f_eligible, conv_s2f.hip) under the strict policy with a SPLIT-COPY input; Cin a multiple of 32
bool s2f_x3_eligible(const ConvArgs &a) {
    if (!a.in_split) return false;
    if (a.n_phase != 4 || a.splitk != 1 || a.raw_out || a.in_step != 1 || a.out_step != 2 || !s3_phase_order_ok(a)) return false;
    if (a.Ho != 2 * a.Hq || a.Wo != 2 * a.Wq || a.Hi != a.Hq || a.Wi != a.Wq) return false;
    if (a.Ci < 32 || (a.Ci & 31) || a.Co < 64 || (a.Co & 63) || a.pad_mode != DL_PAD_ZERO || a.bn_y != nullptr || a.in_act != DL_ACT_NONE) return false;
    if (a.act != DL_ACT_NONE && a.act != DL_ACT_RELU) return false;
    if ((a.in_pstride & 3) || (a.out_pstride & 3)) return false;
    // the size rule of the bf16 kernel (profiles/r04/s2f_first_look.txt): >= 256 tiles of 256 phase-grid pixels; DL_CONV_S2F=2 lifts it (tests)
    const char *env = dl_switch(DL_SW_CONV_S2F);
    if (a.Mtot < 65536 && !(env && env[0] == '2')) return false;
    int ndist = 0;
    int16_t seen[S3_MAX_OFF + 1];
    for (int t = 0; t < a.phase_tap_begin[4]; ++t) {
        bool dup = false;
        for (int i = 0; i < ndist; ++i) dup |= seen[i] == a.taps[t];
        if (!dup) {
            if (ndist == S3_MAX_OFF) return false;
            seen[ndist++] = a.taps[t];
        }
    }
    for (int p = 0; p < 4; ++p)
        for (int t = a.phase_tap_begin[p]; t < a.phase_tap_begin[p + 1]; ++t)
            for (int t2 = t + 1; t2 < a.phase_tap_begin[p + 1]; ++t2)
                if (a.taps[t] == a.taps[t2]) return false;
    return ndist > 0;
}

int launch_conv_s2f_x3(const ConvArgs &a0, hipStream_t stream) {
    if (!a0.w_lo) DL_FAIL("dl_conv_forward(s2f_x3): the strict policy needs the lo weight plane");
    S2fX3Args sa;
    memset(&sa, 0, sizeof(sa));
    sa.a = a0;
    ConvArgs &a = sa.a;
    a.tiles_m = (a.Mtot + 255) / 256;
    a.tiles_n = a.Co / 64;
    int no = 0;
    int16_t key[S3_MAX_OFF];
    for (int p = 0; p < 4; ++p)
        for (int t = a.phase_tap_begin[p]; t < a.phase_tap_begin[p + 1]; ++t) {
            int o = -1;
            for (int i = 0; i < no; ++i)
                if (key[i] == a.taps[t]) o = i;
            if (o < 0) {
                o = no++;
                key[o] = a.taps[t];
                sa.off_dh[o] = (int8_t)(a.taps[t] & 0xff);
                sa.off_dw[o] = (int8_t)((a.taps[t] >> 8) & 0xff);
                sa.off_delta[o] = ((int)sa.off_dh[o] * a.Wi + (int)sa.off_dw[o]) * a.in_pstride;
                for (int q = 0; q < 4; ++q) sa.user_k[o][q] = -1;
            }
            sa.user_k[o][p] = a.phase_kbase[p] + (t - a.phase_tap_begin[p]) * a.Ci;
        }
    sa.n_off = no;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_s2f_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S3_LDS);
        if (e != hipSuccess) DL_FAIL("dl_conv_forward(s2f_x3): hipFuncSetAttribute(%zu): %s", S3_LDS, hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(a.tiles_m * a.tiles_n, 1);
    hipLaunchKernelGGL(conv_s2f_x3_kernel, grid, dim3(512), S3_LDS, stream, sa);
    DL_CHECK_LAUNCH("dl_conv_forward(s2f_x3)");
    return 0;
}
