// conv_d1g.hip -- the data gradient of the PatchGAN's first layer, Conv2d(6, 64, k4, s2, p1) (NLayerDiscriminator, networks.py:638-641; reached from
// loss_G.backward(), DeepLIIF_model.py:429): dx[n, 2h+ph, 2w+pw, c] = sum over the taps (dh, dw) of sub-pixel phase (ph, pw) of dy[n, h+dh, w+dw, :] . W_p,t[c, :],
// 64 contracted channels, 8 (6 real) output channels.
//
// 6.4 GF over 67 MB of dy and 33 MB of dx.  The gather GEMM ran the four phases as four grids of its narrowest tile (conv_gemm_glds_kernel<256,16,32>, 8 output
// channels in a 16-wide tile): 146 us = 0.7 TB/s.  Here the FOUR PHASES are the M dimension: row m = phase * 8 + channel of v_mfma_f32_32x32x16_bf16, K = the 9
// input offsets (dh, dw) in {-1, 0, 1}^2 x 64 channels with zero blocks where a phase has no tap at an offset (4 of 9 are used: 36 MFMAs per 32 pixels instead of
// 16 -- irrelevant next to the memory time).  The 32 x 576 weight matrix is 36 A fragments = 144 VGPRs per wave, loaded once; a wave owns 32 pixels of a
// 128-pixel segment of dy row h; the dy rows h-1, h, h+1 are read from a ring of four 16.6 KB LDS slots (every row segment staged once, read three times);
// the accumulators leave as 8-byte stores (4 channels of one output pixel per lane and phase), no LDS transpose: dx is a sixth of the traffic.
// Same descriptor and packed weights as the four-phase paths (n_phase = 4, per-phase tap lists and weight column bases): no host change beyond the dispatch.
#include "conv_args.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((address_space(3))) char lds_char_t;
typedef __attribute__((address_space(3))) const bf16x8_t lds_frag_t;

template <int V> struct D1GIC { static constexpr int value = V; };

struct D1gArgs {
    ConvArgs a;
    int R, nstrips, segs;
    int kb[4][3][3];                 // weight column base of phase p's tap at input offset (dh + 1, dw + 1); -1 = the phase has no such tap
};

constexpr int D1G_SLOT = 17 * 1024;                       // one dy row segment: 130 pixels (one halo pixel each side) x 128 B in 17 DMA pieces of 8 pixels (136)
constexpr size_t D1G_LDS = (size_t)4 * D1G_SLOT;
static_assert(2 * D1G_LDS <= 160 * 1024, "two workgroups per CU");

__global__ void __launch_bounds__(256, 2) conv_d1g_kernel(const D1gArgs sa) {
    const ConvArgs &a = sa.a;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_char_t *lds = (lds_char_t *)smem_raw;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;

    int b = blockIdx.x;
    const int strip = b % sa.nstrips; b /= sa.nstrips;
    const int seg = b % sa.segs;
    const int n = b / sa.segs;
    const int h0 = strip * sa.R;
    const int w0 = seg * 128;

    // ---- weights: A fragment (offset o = (dh+1)*3 + (dw+1), 16-channel chunk s): lane (m = lr = phase*8 + c, K half lh) holds W_p,t[c][s*16 + lh*8 ..] or zeros
    bf16x8_t W[9][4];
    {
        const int p = lr >> 3, c = lr & 7;
        const bf16_t *wrow = a.w_hi + (size_t)c * a.w_kstride + lh * 8;
#pragma unroll
        for (int o = 0; o < 9; ++o) {
            const int kb = p == 0 ? sa.kb[0][o / 3][o % 3] : (p == 1 ? sa.kb[1][o / 3][o % 3] : (p == 2 ? sa.kb[2][o / 3][o % 3] : sa.kb[3][o / 3][o % 3]));
#pragma unroll
            for (int s = 0; s < 4; ++s) W[o][s] = kb >= 0 ? *reinterpret_cast<const bf16x8_t *>(wrow + kb + s * 16) : bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    // ---- staging: LDS pixel q of a slot = dy pixel w0 - 1 + q (q = 0 .. 129); piece = 8 pixels x 128 B; wave w issues pieces w, w+4, .. of the 17; chunk c of
    // pixel q sits at position c ^ ((q >> 1) & 7) (applied to the SOURCE address).  buffer_load ... lds, the buffer starts one pixel in front of the image
    const int lrow = lane >> 3, lcp = lane & 7;
    const int psb = a.in_pstride * 2;
    const unsigned OOB = 0x80000000u;
    const size_t row_bytes = (size_t)a.Wi * psb;
    const char *in = reinterpret_cast<const char *>(a.in);
    const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(in + (size_t)n * a.Hi * row_bytes - psb), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_none = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(in), 0, 0, 0x00020000);
    unsigned v_off[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int q = 8 * (wave + 4 * k) + lrow;
        const int gx = w0 - 1 + q;                          // dy pixel
        v_off[k] = (gx >= 0 && gx < a.Wi && q < 130) ? (unsigned)((gx + 1) * psb + ((lcp ^ ((q >> 1) & 7)) << 4)) : OOB;
    }
    auto stage_piece = [&](auto Kc, int soff, int slot, const __amdgpu_buffer_rsrc_t rs) __attribute__((always_inline)) {
        constexpr int K = decltype(Kc)::value;
        if (K < 4 || wave == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(lds + slot * D1G_SLOT + (wave + 4 * K) * 1024), 16, (int)v_off[K], soff, 0, 0);
    };
    auto stage = [&](int r, int slot) __attribute__((always_inline)) {       // dy row r (outside the tensor: zeros) -> ring slot
        const bool real = r >= 0 && r < a.Hi;
        const __amdgpu_buffer_rsrc_t rs = real ? rsrc_in : rsrc_none;
        const int soff = real ? r * (int)row_bytes : 0;
        stage_piece(D1GIC<0>{}, soff, slot, rs); stage_piece(D1GIC<1>{}, soff, slot, rs); stage_piece(D1GIC<2>{}, soff, slot, rs);
        stage_piece(D1GIC<3>{}, soff, slot, rs); stage_piece(D1GIC<4>{}, soff, slot, rs);
    };

    // ---- fragment addressing: the wave's pixel w = w0 + wave*32 + lr; offset dw reads LDS pixel q = wave*32 + lr + 1 + dw; ^ (s << 5) for the 16-channel chunk
    int a_dw[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int q = wave * 32 + lr + d;                   // d = dw + 1
        a_dw[d] = q * 128 + ((lh ^ ((q >> 1) & 7)) << 4);
    }

    // ---- output: lane (pixel lr, K half lh) holds for phase p = r >> 2 the channels 4 lh + (r & 3): 8 bytes of output pixel (2h + ph, 2w + pw)
    const int opb = a.out_pstride * 2;
    const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(a.out) + (size_t)n * a.Ho * a.Wo * opb, 0, 0x7fffffff, 0x00020000);
    const unsigned st_off = (unsigned)(2 * (w0 + wave * 32 + lr) * opb + lh * 8);

    // ---- pipeline: dy row r lives in ring slot (r - h0 + 1) & 3; step h reads rows h-1, h, h+1 and stages row h+2
    const int R = sa.R;
    stage(h0 - 1, 0);
    stage(h0, 1);
    stage(h0 + 1, 2);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    for (int t = 0; t < R; ++t) {
        const int h = h0 + t;
        const int sm = t & 3, s0 = (t + 1) & 3, sp = (t + 2) & 3;          // slots of rows h-1, h, h+1
        if (t + 1 < R) stage(h + 2, (t + 3) & 3);
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int o = 0; o < 9; ++o) {
            const int slot = o / 3 == 0 ? sm : (o / 3 == 1 ? s0 : sp);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8_t f = *reinterpret_cast<lds_frag_t *>(lds + slot * D1G_SLOT + (a_dw[o % 3] ^ (s << 5)));
                acc = dl_mfma32(W[o][s], f, acc);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            u32x2_t pk;
            pk[0] = pack2_bf16(acc[p * 4 + 0], acc[p * 4 + 1]);
            pk[1] = pack2_bf16(acc[p * 4 + 2], acc[p * 4 + 3]);
            __builtin_amdgcn_raw_buffer_store_b64(pk, rsrc_out, (int)st_off, ((2 * h + (p >> 1)) * a.Wo + (p & 1)) * opb, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // vmcnt(4): everything but this step's 4 stores per lane, i.e. the row staged during the step (issued before them), has landed
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// fills kb[p][dh+1][dw+1]; false when the descriptor is not the 2 x 2-phase form of a 4 x 4 stride-2 layer with taps at offsets in {-1, 0, 1}
static bool d1g_tap_table(const ConvArgs &a, int (&kb)[4][3][3]) {
    for (int p = 0; p < 4; ++p) {
        if (a.phase_oh[p] != (p >> 1) || a.phase_ow[p] != (p & 1)) return false;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) kb[p][i][j] = -1;
        const int nt = a.phase_tap_begin[p + 1] - a.phase_tap_begin[p];
        if (nt != 4) return false;
        for (int t = 0; t < nt; ++t) {
            const int16_t tp = a.taps[a.phase_tap_begin[p] + t];
            const int dh = (int)(int8_t)(tp & 0xff), dw = (int)(int8_t)((tp >> 8) & 0xff);
            if (dh < -1 || dh > 1 || dw < -1 || dw > 1 || kb[p][dh + 1][dw + 1] >= 0) return false;
            kb[p][dh + 1][dw + 1] = a.phase_kbase[p] + t * a.Ci;
        }
    }
    return true;
}

static int d1g_strip_rows(const ConvArgs &a) {
    const int per_img = a.N * (a.Wq / 128);
    int best = 0;
    for (int R = 1; R <= a.Hq; ++R) {
        if (a.Hq % R) continue;
        const int wgs = per_img * (a.Hq / R);
        if (best == 0 || wgs >= 480) best = R;
        if (wgs < 480) break;
    }
    return best;
}

// The layer this kernel serves: four sub-pixel phases (out_step 2, in_step 1) of a 4 x 4 stride-2 layer, exact 2x geometry, zero padding, exactly 64 contracted
// channels, 8 (padded) output channels, phase-grid rows that are multiples of 128 pixels, bf16, no bias / activation / split-K / raw accumulators / statistics.
bool d1g_eligible(const ConvArgs &a) {
    if (a.n_phase != 4 || a.splitk != 1 || a.raw_out || a.in_step != 1 || a.out_step != 2) return false;
    if (a.Ho != 2 * a.Hq || a.Wo != 2 * a.Wq || a.Hi != a.Hq || a.Wi != a.Wq || (a.Wq & 127)) return false;
    if (a.Ci != 64 || a.Co != 8 || a.pad_mode != DL_PAD_ZERO || a.bn_y != nullptr || a.in_act != DL_ACT_NONE || a.act != DL_ACT_NONE || a.bias != nullptr ||
        a.stats_part != nullptr)
        return false;
    if ((size_t)a.Hi * a.Wi * (size_t)a.in_pstride * 2 >= ((size_t)1 << 31) || (size_t)a.Ho * a.Wo * (size_t)a.out_pstride * 2 >= ((size_t)1 << 31)) return false;
    int kb[4][3][3];
    return d1g_tap_table(a, kb) && d1g_strip_rows(a) > 0;
}

int launch_conv_d1g(const ConvArgs &a0, hipStream_t stream) {
    D1gArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.a = a0;
    ConvArgs &a = sa.a;
    if (!d1g_tap_table(a, sa.kb)) DL_FAIL("dl_conv_forward(d1g): not the four-phase form of a 4x4 stride-2 layer");
    sa.R = d1g_strip_rows(a);
    sa.nstrips = a.Hq / sa.R;
    sa.segs = a.Wq / 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_d1g_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)D1G_LDS);
        if (e != hipSuccess) DL_FAIL("dl_conv_forward(d1g): hipFuncSetAttribute(%zu): %s", D1G_LDS, hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(conv_d1g_kernel, dim3(a.N * sa.segs * sa.nstrips), dim3(256), D1G_LDS, stream, sa);
    DL_CHECK_LAUNCH("dl_conv_forward(d1g)");
    return 0;
}
