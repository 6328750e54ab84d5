// conv_s2f.hip -- the FUSED four-phase tile for the stride-2 layers ("fused four-phase tile"): ConvTranspose2d(k3, s2, p1, op1)
// forward (ResnetGenerator up1 / up2, networks.py:400-436), the data gradient of Conv2d(k3 | k4, s2, p1) (ResnetGenerator down1 / down2,
// NLayerDiscriminator c2-c4, networks.py:576-609 / 638-660): out[n, 2h+oh, 2w+ow, co] = sum over the taps of sub-pixel phase (oh, ow) of
// x[n, h+dh, w+dw, :] . W_p,t[co, :].
//
// The 4-phase gather GEMM (conv_gemm_glds_kernel, one workgroup per (pixel tile, PHASE)) stages the input tile once per (phase, tap) and the
// four phases of a tile run a whole grid apart: PMC on up2 (128 -> 64 @ 256^2) showed 546 MB fetched for a 134 MB input, 250 TF/s.  Here ONE
// workgroup owns a tile of 256 phase-grid pixels x 64 output channels for ALL four phases:
//   * K loop over (64-channel chunk, DISTINCT input offset (dh, dw)): the offset's input tile (256 px x 64 ch, 32 KB) is staged ONCE and feeds
//     every phase that has a tap at that offset (k3: offset (0,0) feeds 4 phases, (0,1) and (1,0) two, (1,1) one; k4: 9 offsets, 16 taps) --
//     4 input tiles per chunk instead of 9 (k3) / 9 instead of 16 (k4); the tap's 64 x 64 weight tile (8 KB) goes next to it;
//   * four accumulator sets (8 waves as 4 pixel groups x 2 channel groups: 64 px x 32 ch x 4 phases = 8 accumulators of
//     v_mfma_f32_32x32x16_bf16 per wave), double-buffered LDS, one barrier per (chunk, offset);
//   * epilogue: bias / ReLU, the four 64-channel results of a pixel go through LDS as one 512-byte row and leave as 128-byte pieces to the
//     2 x 2 output pixels of that phase-grid pixel (256 contiguous bytes per output row when the tensor is 64 channels wide); fused norm
//     statistics (sum over pixels AND phases), one chunk per tile.
// Consumes the SAME descriptor and packed weights as the 4-phase path (n_phase = 4, per-phase tap lists and weight column bases): no host change.
#include "conv_args.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((address_space(3))) char lds_char_t;
typedef __attribute__((address_space(3))) const bf16x8_t lds_frag_t;

__device__ __attribute__((aligned(64))) unsigned char g_s2f_zero_page[64];

template <int V> struct S2IC { static constexpr int value = V; };

constexpr int S2F_MAX_OFF = 9;
struct S2fArgs {
    ConvArgs a;
    int n_off;
    int off_delta[S2F_MAX_OFF];            // element offset (dh * Wi + dw) * in_pstride of the offset
    int8_t off_dh[S2F_MAX_OFF], off_dw[S2F_MAX_OFF];
    int user_k[S2F_MAX_OFF][4];            // per offset and phase: weight column base (kbase_p + tap_index * Ci) or -1
};

__device__ __forceinline__ float s2f_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
    return v;
}

constexpr int S2F_XB = 256 * 128;              // input tile of one (chunk, offset): 256 pixels x 64 channels
constexpr int S2F_WB = 64 * 128;               // weight tile of one tap: 64 output channels x 64 K
constexpr int S2F_BUF = S2F_XB + 4 * S2F_WB;   // one pipeline buffer: the input tile + up to four taps' weights
constexpr size_t S2F_LOOP_LDS = (size_t)2 * S2F_BUF;
constexpr size_t S2F_EPI_LDS = (size_t)256 * 512 + 256 * sizeof(int) + (size_t)4 * 2 * 64 * sizeof(float);
constexpr size_t S2F_LDS = S2F_LOOP_LDS > S2F_EPI_LDS ? S2F_LOOP_LDS : S2F_EPI_LDS;

__global__ void __launch_bounds__(512) conv_s2f_kernel(const S2fArgs sa) {
    const ConvArgs &a = sa.a;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_char_t *lds = (lds_char_t *)smem_raw;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;                // 64-pixel group, 32-channel group
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % a.tiles_n, tm = bid / a.tiles_n;
    const int nch = a.Ci >> 6;
    const int NO = sa.n_off;
    const int T = nch * NO;                                  // macro steps: (chunk, offset), offset fastest

    // ---- staging geometry.  Input tile: instruction i of wave w fills tile rows (w*4 + i)*8 .. +8 (phase-grid pixels, flattened (n, h, w));
    // weights of a tap: wave w fills rows w*8 .. +8 (output channels) of that tap's 64-row tile.  LDS rows are 128 B = 8 chunks of 16 B; chunk c
    // of row r sits at position c ^ ((r >> 1) & 7) (permutation applied to the SOURCE address: the DMA image is lane-linear).
    const int lrow = lane >> 3, lcp = lane & 7;
    const int HWq = a.Hq * a.Wq;
    const bf16_t *in = reinterpret_cast<const bf16_t *>(a.in);
    const bf16_t *zero = reinterpret_cast<const bf16_t *>(g_s2f_zero_page);
    const bf16_t *x_ptr[4];
    unsigned x_mask[4];                                      // bit o: offset o of this pixel lies inside the image
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = (wave * 4 + i) * 8 + lrow;
        const int m = tm * 256 + s;
        const bool ok = m < a.Mtot;
        const int mm = ok ? m : 0;
        const int n = mm / HWq, rem = mm - n * HWq;
        const int hq = rem / a.Wq, wq = rem - hq * a.Wq;
        x_ptr[i] = in + ((size_t)(n * a.Hi + hq) * a.Wi + wq) * (size_t)a.in_pstride + (lcp ^ ((s >> 1) & 7)) * 8;
        unsigned mk = 0;
        if (ok)
            for (int o = 0; o < NO; ++o) {
                const int hi = hq + sa.off_dh[o], wi = wq + sa.off_dw[o];
                if (((unsigned)hi < (unsigned)a.Hi) && ((unsigned)wi < (unsigned)a.Wi)) mk |= 1u << o;
            }
        x_mask[i] = mk;
    }
    const int wrow = wave * 8 + lrow;                        // output channel inside the 64-channel tile this lane stages
    const bf16_t *w_ptr = a.w_hi + (size_t)(tn * 64 + wrow) * a.w_kstride + (lcp ^ ((wrow >> 1) & 7)) * 8;

    // stage macro step u (chunk u / NO, offset u % NO) into buffer u & 1: 4 input pieces + one weight piece per phase that uses the offset
    auto stage = [&](int u) __attribute__((always_inline)) {
        const int c = u / NO, o = u - c * NO;
        const int buf = (u & 1) * S2F_BUF;
        const ptrdiff_t xd = (ptrdiff_t)sa.off_delta[o] + c * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = (x_mask[i] >> o) & 1u;
            const bf16_t *src = ok ? x_ptr[i] + xd : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(lds + buf + (wave * 4 + i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int kb = sa.user_k[o][p];
            if (kb >= 0)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(w_ptr + kb + c * 64),
                                                 (__attribute__((address_space(3))) void *)(lds + buf + S2F_XB + p * S2F_WB + wave * 1024), 16, 0, 0);
        }
    };

    // ---- fragment addressing (bytes): lane = (row lr of a 32-row block, K half lh of a 16-wide sub-step)
    const int lr = lane & 31, lh = lane >> 5;
    const int ax = (wm * 64 + lr) * 128 + ((lh ^ ((lr >> 1) & 7)) << 4);                  // + jb * 4096 (pixel block), ^ (s << 5) (sub-step)
    const int aw = S2F_XB + (wn * 32 + lr) * 128 + ((lh ^ ((lr >> 1) & 7)) << 4);         // + p * S2F_WB

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
        const int buf = (u & 1) * S2F_BUF;
        bf16x8_t xf[4][2];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j) xf[s][j] = *reinterpret_cast<lds_frag_t *>(lds + buf + (ax ^ (s << 5)) + j * 4096);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (sa.user_k[o][p] < 0) continue;               // wave-uniform (kernel argument)
            bf16x8_t wf[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) wf[s] = *reinterpret_cast<lds_frag_t *>(lds + buf + p * S2F_WB + (aw ^ (s << 5)));
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[p][j] = dl_mfma32(wf[s], xf[s][j], acc[p][j]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();             // LDS is dead from here on (the epilogue reuses it)

    // ---- epilogue.  acc[p][j][r] = out phase p, pixel = wm*64 + j*32 + lr, channel = wn*32 + 8*(r>>2) + 4*lh + (r&3)
    lds_char_t *tile = lds;                                  // [256 pixels][4 phases x 64 channels] bf16, 8-byte units XOR-swizzled by the pixel
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
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int cl = wn * 32 + q * 8 + lh * 4;                 // channel inside the 64-channel tile
        const int co = tn * 64 + cl;
        float bias[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bias[e] = (co + e < a.bias_n) ? a.bias[co + e] : 0.f;
        }
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int vch = p * 64 + cl;
            const int unit = (vch >> 2) ^ (((lr & 15) << 1) & 62);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[p][j][q * 4 + e] + bias[e];
                if (a.act == DL_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                u32x2_t pk;
                pk[0] = pack2_bf16(v[0], v[1]);
                pk[1] = pack2_bf16(v[2], v[3]);
                *reinterpret_cast<__attribute__((address_space(3))) u32x2_t *>(tile + (wm * 64 + j * 32 + lr) * 512 + unit * 8) = pk;
                if (want_stats && live[j]) {          // statistics of exactly what is stored (bf16-rounded)
                    const float q0 = h16_lo_f32(pk[0]), q1 = h16_hi_f32(pk[0]);
                    const float q2 = h16_lo_f32(pk[1]), q3 = h16_hi_f32(pk[1]);
                    s1[0] += q0; s2[0] += q0 * q0; s1[1] += q1; s2[1] += q1 * q1;
                    s1[2] += q2; s2[2] += q2 * q2; s1[3] += q3; s2[3] += q3 * q3;
                }
            }
        }
        if (want_stats) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1[e] = s2f_row16_sum(s1[e]); s2[e] = s2f_row16_sum(s2[e]);
                s1[e] += __shfl_xor(s1[e], 16, 64); s2[e] += __shfl_xor(s2[e], 16, 64);
            }
            if (lr == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    red[(wm * 2 + 0) * 64 + cl + e] = s1[e];
                    red[(wm * 2 + 1) * 64 + cl + e] = s2[e];
                }
            }
        }
    }
    __syncthreads();
    bf16_t *out = reinterpret_cast<bf16_t *>(a.out);
#pragma unroll 4
    for (int idx = tid; idx < 256 * 32; idx += 512) {
        const int row = idx >> 5, cc = idx & 31;             // 16-byte chunk cc of the row: phase cc >> 3, channels (cc & 7) * 8 ..
        const int opix = rowtab[row];
        const int p = cc >> 3, co = tn * 64 + (cc & 7) * 8;
        if (opix < 0 || co >= a.Co) continue;
        const int unit = (cc * 2) ^ (((row & 15) << 1) & 62);
        const u32x4_t v = *reinterpret_cast<__attribute__((address_space(3))) const u32x4_t *>(tile + row * 512 + unit * 8);
        *reinterpret_cast<u32x4_t *>(out + ((size_t)opix + (size_t)(p >> 1) * a.Wo + (p & 1)) * a.out_pstride + co) = v;
    }
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

// phase p of a 2 x 2 sub-pixel decomposition writes output pixel (2 hq + p / 2, 2 wq + p % 2)
static bool s2f_phase_order_ok(const ConvArgs &a) {
    for (int p = 0; p < 4; ++p)
        if (a.phase_oh[p] != (p >> 1) || a.phase_ow[p] != (p & 1)) return false;
    return true;
}

// The layers this kernel serves: four sub-pixel phases (out_step 2, in_step 1) in the order (0,0) (0,1) (1,0) (1,1), exact 2x geometry, zero padding,
// Cin a multiple of 64, Co a multiple of 64, bf16 result without split-K / raw accumulators / input activation / fused norm-backward reductions,
// at most 9 distinct input offsets.
bool s2f_eligible(const ConvArgs &a) {
    if (a.n_phase != 4 || a.splitk != 1 || a.raw_out || a.in_step != 1 || a.out_step != 2 || !s2f_phase_order_ok(a)) return false;
    if (a.Ho != 2 * a.Hq || a.Wo != 2 * a.Wq || a.Hi != a.Hq || a.Wi != a.Wq) return false;
    if (a.Ci < 64 || (a.Ci & 63) || a.Co < 64 || (a.Co & 63) || a.pad_mode != DL_PAD_ZERO || a.bn_y != nullptr || a.in_act != DL_ACT_NONE) return false;
    if (a.act != DL_ACT_NONE && a.act != DL_ACT_RELU) return false;
    if (a.epi_old) return false;
    // Size rule (same-box A/B, profiles/r04/s2f_first_look.txt): the tile pays where the phase grid has >= 256 tiles of 256 pixels -- up2 forward 318 -> 216 us,
    // down1 data gradient 266 -> 190, up1 forward 187 -> 154, down2 data gradient 165 -> 137, PatchGAN c2 data gradient 98 -> 77 -- ties at 128 tiles (c3: 62 / 61)
    // and loses at 32 tiles x 8 channel tiles (c4: 53 -> 95 us: one workgroup per CU, the input tile staged by 8 channel tiles).  DL_CONV_S2F=2 lifts the rule (tests).
    const char *env = dl_switch(DL_SW_CONV_S2F);
    if (a.Mtot < 65536 && !(env && env[0] == '2')) return false;
    int ndist = 0;
    int16_t seen[S2F_MAX_OFF + 1];
    for (int t = 0; t < a.phase_tap_begin[4]; ++t) {
        bool dup = false;
        for (int i = 0; i < ndist; ++i) dup |= seen[i] == a.taps[t];
        if (!dup) {
            if (ndist == S2F_MAX_OFF) return false;
            seen[ndist++] = a.taps[t];
        }
    }
    // one tap per (phase, offset)
    for (int p = 0; p < 4; ++p)
        for (int t = a.phase_tap_begin[p]; t < a.phase_tap_begin[p + 1]; ++t)
            for (int t2 = t + 1; t2 < a.phase_tap_begin[p + 1]; ++t2)
                if (a.taps[t] == a.taps[t2]) return false;
    return ndist > 0;
}

// chunks of fused norm statistics this kernel writes: one per 256-pixel tile, tiles must not straddle images
int s2f_stats_chunks(const ConvArgs &a) {
    const int hw = a.Hq * a.Wq;
    return (hw % 256) ? 0 : hw / 256;
}

int launch_conv_s2f(const ConvArgs &a0, hipStream_t stream) {
    S2fArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.a = a0;
    ConvArgs &a = sa.a;
    a.tiles_m = (a.Mtot + 255) / 256;
    a.tiles_n = a.Co / 64;
    // distinct offsets in order of first appearance; per (offset, phase) the weight column base of the tap
    int no = 0;
    int16_t key[S2F_MAX_OFF];
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
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_s2f_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S2F_LDS);
        if (e != hipSuccess) DL_FAIL("dl_conv_forward(s2f): hipFuncSetAttribute(%zu): %s", S2F_LDS, hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(a.tiles_m * a.tiles_n, 1);
    hipLaunchKernelGGL(conv_s2f_kernel, grid, dim3(512), S2F_LDS, stream, sa);
    DL_CHECK_LAUNCH("dl_conv_forward(s2f)");
    return 0;
}
