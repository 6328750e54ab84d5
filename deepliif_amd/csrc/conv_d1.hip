// conv_d1.hip -- the PatchGAN's first layer, Conv2d(6, 64, k4, s2, p1) + LeakyReLU(0.2) (NLayerDiscriminator, networks.py:638-641), forward:
// out[n, ho, wo, co] = act(bias[co] + sum_{kh,kw,ci} x[n, 2ho-1+kh, 2wo-1+kw, ci] W[co, (kh,kw), ci]),  8 (6 real) contracted and 64 output channels.
//
// 6.4 GF over 33 MB of input and 67 MB of output: a streaming layer (64 flop per byte), 15 launches per training step.  The gather GEMM
// (conv_gemm_glds_kernel<128,64,64>) staged a 128-pixel x 64-wide K tile per step for it and ran 58 us = 1.7 TB/s.  Here:
//   * an input pixel is 16 bytes = ONE K-half of v_mfma_f32_32x32x16_bf16: K sub-step s = (kh, kw pair p) reads the pixels 2wo-1+2p and 2wo+2p -- the 64 lanes
//     of a fragment read 2 KB of CONSECUTIVE LDS bytes (no swizzle, no de-interleave); the weights are 8 fragments = 32 VGPRs per wave, loaded once;
//   * a workgroup walks down a strip of output rows of the full 256-pixel width (4 waves = 2 channel halves x 2 pixel halves); input rows 2ho+1 and 2ho+2 feed
//     output row ho (kh 2, 3) and ho+1 (kh 0, 1) from one fragment read: two accumulator sets, every input row staged once (8 KB + two halo pixels);
//   * the epilogue IS the work: bias / LeakyReLU, bf16, 256 x 64 tile transposed through 32 KB of LDS, 128-byte NHWC pixel rows per 8 lanes; 65 KB of LDS and
//     ~230 registers per wave, so TWO workgroups share a CU and one's epilogue hides behind the other's loads and MFMAs.
// Same descriptor and packed weights as the gather GEMM (n_phase = 1, in_step = 2, 16 taps (kh - 1, kw - 1) kh-major): no host change beyond the dispatch.
#include "conv_args.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((address_space(3))) char lds_char_t;
typedef __attribute__((address_space(3))) const bf16x8_t lds_frag_t;

template <int V> struct D1IC { static constexpr int value = V; };

struct D1Args {
    ConvArgs a;
    int R, nstrips;
};

constexpr int D1_W = 256;                                  // output pixels per row handled by one workgroup (the whole row)
constexpr int D1_SLOT = 9 * 1024;                          // one input row: 514 pixels x 16 B = 8 224 B in 9 DMA pieces of 64 pixels
constexpr int D1_TILE = 4 * D1_SLOT;                       // epilogue tile: 256 pixels x 64 channels bf16
constexpr int D1_BIAS = D1_TILE + 256 * 128;
constexpr size_t D1_LDS = (size_t)D1_BIAS + 256;
static_assert(2 * D1_LDS <= 160 * 1024, "two workgroups per CU");

__global__ void __launch_bounds__(256, 2) conv_d1_kernel(const D1Args sa) {
    const ConvArgs &a = sa.a;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_char_t *lds = (lds_char_t *)smem_raw;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int cg = wave & 1, ph = wave >> 1;               // 32-channel half, 128-pixel half

    int b = blockIdx.x;
    const int strip = b % sa.nstrips; b /= sa.nstrips;
    const int tn = b % a.tiles_n;
    const int n = b / a.tiles_n;
    const int ho0 = strip * sa.R;

    // ---- weights: K sub-step s = kh*2 + p covers the taps (kh, 2p) and (kh, 2p+1): lane (co = lr, K half lh) holds tap (kh, 2p + lh), 8 channels
    bf16x8_t W[8];
    {
        const bf16_t *wp = a.w_hi + (size_t)(tn * 64 + cg * 32 + lr) * a.w_kstride + a.phase_kbase[0] + lh * 8;
#pragma unroll
        for (int s = 0; s < 8; ++s) W[s] = *reinterpret_cast<const bf16x8_t *>(wp + s * 16);
    }

    // ---- staging: LDS pixel q of a row slot = image pixel q - 1 (q = 0: the left padding); piece = 64 pixels; wave w issues pieces w, w+4, w+8 (9 pieces)
    const int psb = a.in_pstride * 2;
    const unsigned OOB = 0x80000000u;
    const size_t row_bytes = (size_t)a.Wi * psb;
    const char *in = reinterpret_cast<const char *>(a.in);
    // (the buffer starts one pixel in front of the image, so that pixel q - 1 has the non-negative lane offset q * psb; never dereferenced there: q = 0 is out of range)
    const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(in + (size_t)n * a.Hi * row_bytes - psb), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_none = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(in), 0, 0, 0x00020000);
    unsigned v_off[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int q = 64 * (wave + 4 * k) + lane;
        v_off[k] = (q >= 1 && q <= a.Wi && q < 9 * 64) ? (unsigned)(q * psb) : OOB;
    }
    auto stage_piece = [&](auto Kc, int soff, int slot, const __amdgpu_buffer_rsrc_t rs) __attribute__((always_inline)) {
        constexpr int K = decltype(Kc)::value;
        if (K < 2 || wave == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(lds + slot * D1_SLOT + (wave + 4 * K) * 1024), 16, (int)v_off[K], soff, 0, 0);
    };
    auto stage = [&](int r, int slot) __attribute__((always_inline)) {       // input row r (outside the image: zeros) -> ring slot
        const bool real = r >= 0 && r < a.Hi;
        const __amdgpu_buffer_rsrc_t rs = real ? rsrc_in : rsrc_none;
        const int soff = real ? r * (int)row_bytes : 0;
        stage_piece(D1IC<0>{}, soff, slot, rs); stage_piece(D1IC<1>{}, soff, slot, rs); stage_piece(D1IC<2>{}, soff, slot, rs);
    };

    // ---- fragment address (bytes inside a slot): pixel q = 2 wo + 2p + lh, wo = ph*128 + j*32 + lr   ->   + p*32 + j*1024
    const int a_frag = (ph * 128 + lr) * 32 + lh * 16;

    lds_char_t *tile = lds + D1_TILE;
    __attribute__((address_space(3))) float *bias_l = reinterpret_cast<__attribute__((address_space(3))) float *>(lds + D1_BIAS);
    if (tid < 64) {
        const int co = tn * 64 + tid;
        bias_l[tid] = (a.bias && co < a.bias_n) ? a.bias[co] : 0.f;
    }
    __syncthreads();
    const float slope = a.act == DL_ACT_LRELU ? 0.2f : (a.act == DL_ACT_RELU ? 0.f : 1.f);          // act(v) = v > 0 ? v : slope * v
    const int opb = a.out_pstride * 2;
    const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(a.out) + (size_t)n * a.Ho * a.Wo * opb, 0, 0x7fffffff, 0x00020000);
    const unsigned st_off = (unsigned)((tid >> 3) * opb + (tn * 64 + (tid & 7) * 8) * 2);             // pixel (tid >> 3) + 32 it, 16-byte chunk tid & 7

    f32x16_t accA[4], accB[4];
    auto reset_acc = [&](f32x16_t (&acc)[4]) __attribute__((always_inline)) {
        f32x4_t bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = *reinterpret_cast<__attribute__((address_space(3))) const f32x4_t *>(bias_l + cg * 32 + q * 8 + lh * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = bv[r >> 2][r & 3];
    };
    reset_acc(accA);
    reset_acc(accB);

    // one staged input row with kernel row KHC into `cur` and / or KHN into `nxt` (-1 = none): 2 K sub-steps x 4 pixel blocks
    auto row_mac = [&](auto KHC, auto KHN, int slot, f32x16_t (&cur)[4], f32x16_t (&nxt)[4]) __attribute__((always_inline)) {
        constexpr int khc = decltype(KHC)::value, khn = decltype(KHN)::value;
        bf16x8_t F[2][4];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) F[p][j] = *reinterpret_cast<lds_frag_t *>(lds + slot * D1_SLOT + a_frag + p * 32 + j * 1024);
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (khc >= 0) cur[j] = dl_mfma32(W[(khc < 0 ? 0 : khc) * 2 + p], F[p][j], cur[j]);
                if constexpr (khn >= 0) nxt[j] = dl_mfma32(W[(khn < 0 ? 0 : khn) * 2 + p], F[p][j], nxt[j]);
            }
    };

    // acc[j][q*4 + e] = output channel cg*32 + q*8 + lh*4 + e of pixel ph*128 + j*32 + lr (bias included); afterwards acc = bias
    auto epilogue = [&](f32x16_t (&acc)[4], int ho) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[j][q * 4 + e];
                    v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
                }
                u32x2_t pk;
                pk[0] = pack2_bf16(v[0], v[1]);
                pk[1] = pack2_bf16(v[2], v[3]);
                const int unit = (cg * 8 + q * 2 + lh) ^ ((lr & 7) << 1);
                *reinterpret_cast<__attribute__((address_space(3))) u32x2_t *>(tile + (ph * 128 + j * 32 + lr) * 128 + unit * 8) = pk;
            }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // LDS only: the DMA of the next rows and the stores stay in flight
        const int orow = ho * a.Wo * opb;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int px = (tid >> 3) + 32 * it;
            const int unit = ((tid & 7) * 2) ^ ((px & 7) << 1);
            const u32x4_t v = *reinterpret_cast<__attribute__((address_space(3))) const u32x4_t *>(tile + px * 128 + unit * 8);
            __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_out, (int)st_off, orow + 32 * it * opb, 0);
        }
        reset_acc(acc);
    };

    // ---- pipeline: output row t of the strip (ho = ho0 + t) takes kh 0, 1 from input rows 2ho-1, 2ho (the previous step's rows; the strip's first row: the
    // prelude) and kh 2, 3 from rows 2ho+1, 2ho+2, which also give kh 0, 1 of row ho+1.  Row pair t lives in ring slots 2 (t & 1), 2 (t & 1) + 1.
    const int R = sa.R;
    stage(2 * ho0 - 1, 2);
    stage(2 * ho0, 3);
    stage(2 * ho0 + 1, 0);
    stage(2 * ho0 + 2, 1);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    row_mac(D1IC<0>{}, D1IC<-1>{}, 2, accA, accB);
    row_mac(D1IC<1>{}, D1IC<-1>{}, 3, accA, accB);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");            // slots 2, 3 are about to be refilled
    auto out_row = [&](int t, int s0, f32x16_t (&cur)[4], f32x16_t (&nxt)[4]) __attribute__((always_inline)) {
        const int ho = ho0 + t;
        if (t + 1 < R) { stage(2 * ho + 3, s0 ^ 2); stage(2 * ho + 4, (s0 ^ 2) + 1); }
        row_mac(D1IC<2>{}, D1IC<0>{}, s0, cur, nxt);
        row_mac(D1IC<3>{}, D1IC<1>{}, s0 + 1, cur, nxt);
        __builtin_amdgcn_sched_barrier(0);          // the row's MFMAs stay in front of the epilogue that reads their accumulators
        epilogue(cur, ho);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    for (int t = 0; t < R; t += 2) {                // R is even (d1_strip_rows)
        out_row(t, 0, accA, accB);
        out_row(t + 1, 2, accB, accA);
    }
}

// strip height: an even divisor of Ho that gives the grid about two workgroups per CU
static int d1_strip_rows(const ConvArgs &a) {
    const int per_img = a.N * (a.Co / 64);
    int best = 0;
    for (int R = 2; R <= a.Ho; R += 2) {
        if (a.Ho % R) continue;
        const int wgs = per_img * (a.Ho / R);
        if (best == 0 || wgs >= 480) best = R;
        if (wgs < 480) break;
    }
    return best;
}

// The layer this kernel serves: one phase, input step 2, the sixteen taps (kh - 1, kw - 1) kh-major, zero padding, exactly 8 (padded) contracted channels, output
// channels a multiple of 64, exact 2x geometry with output rows of exactly 256 pixels, bf16, no split-K / raw accumulators / input activation / fused statistics.
bool d1_eligible(const ConvArgs &a) {
    if (a.n_phase != 1 || a.splitk != 1 || a.raw_out || a.in_step != 2 || a.out_step != 1) return false;
    if (a.phase_tap_begin[1] - a.phase_tap_begin[0] != 16 || a.Ci != 8 || a.Co < 64 || (a.Co & 63)) return false;
    if (a.Hi != 2 * a.Ho || a.Wi != 2 * a.Wo || a.Hq != a.Ho || a.Wq != a.Wo || a.Wo != D1_W || (a.Ho & 1)) return false;
    if (a.pad_mode != DL_PAD_ZERO || a.bn_y != nullptr || a.in_act != DL_ACT_NONE || a.epi_old || a.stats_part != nullptr) return false;
    if (a.act != DL_ACT_NONE && a.act != DL_ACT_RELU && a.act != DL_ACT_LRELU) return false;
    if ((size_t)a.Hi * a.Wi * (size_t)a.in_pstride * 2 >= ((size_t)1 << 31) || (size_t)a.Ho * a.Wo * (size_t)a.out_pstride * 2 >= ((size_t)1 << 31)) return false;
    for (int t = 0; t < 16; ++t) {
        const int dh = (int)(int8_t)(a.taps[t] & 0xff), dw = (int)(int8_t)((a.taps[t] >> 8) & 0xff);
        if (dh != t / 4 - 1 || dw != t % 4 - 1) return false;
    }
    return d1_strip_rows(a) > 0;
}

int launch_conv_d1(const ConvArgs &a0, hipStream_t stream) {
    D1Args sa;
    memset(&sa, 0, sizeof(sa));
    sa.a = a0;
    ConvArgs &a = sa.a;
    sa.R = d1_strip_rows(a);
    if (sa.R <= 0) DL_FAIL("dl_conv_forward(d1): no strip height for Ho=%d", a.Ho);
    sa.nstrips = a.Ho / sa.R;
    a.tiles_n = a.Co / 64;
    a.tiles_m = a.N * sa.nstrips;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_d1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)D1_LDS);
        if (e != hipSuccess) DL_FAIL("dl_conv_forward(d1): hipFuncSetAttribute(%zu): %s", D1_LDS, hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(conv_d1_kernel, dim3(a.tiles_m * a.tiles_n), dim3(256), D1_LDS, stream, sa);
    DL_CHECK_LAUNCH("dl_conv_forward(d1)");
    return 0;
}
