// conv_s2u.hip -- the stride-2 3x3 layers in the UP direction with the layer's WEIGHTS HELD IN REGISTERS: ResnetGenerator up2 forward
// (ConvTranspose2d(128, 64, k3, s2, p1, op1), networks.py:425-436) and the data gradient of down1 (Conv2d(64, 128, k3, s2, p1), networks.py:400-404):
// out[n, 2h+ph, 2w+pw, co] = sum over the taps (dh, dw) of sub-pixel phase (ph, pw) of x[n, h+dh, w+dw, :] . W_p,t[co, :], 128 contracted and 64 output channels.
//
// The mirror image of conv_s2d.hip.  268 MB of output + 134 MB of input are HBM-bound (~75 us); the fused four-phase tile (conv_s2f.hip) re-stages the input tile
// per distinct offset and the weights per tap (177-194 us).  The weights are 9 taps x 64 x 128 bf16 = 147 KB -- the register file of a CU holds them:
//   * the nine (phase, tap) products are UNITS of 32 output channels x 128 contracted channels (32 VGPRs of MFMA A operands each); four waves, one per SIMD:
//     waves 0 / 1 own phase (1,1) -- four taps -- for output channels 0-31 / 32-63 (4 units, 128 VGPRs), waves 2 / 3 own phases (0,0), (0,1), (1,0) -- one,
//     two, two taps -- (5 units, 160 VGPRs).  No weight byte is staged after the prologue; every wave writes disjoint outputs;
//   * a workgroup walks down a strip of input rows of ONE 64-pixel row segment: step h multiplies rows h (taps dh = 0) and h + 1 (dh = 1) into the output
//     rows 2h and 2h + 1 of 128 pixels; every input row segment (65 pixels x 128 channels, 16.6 KB) is staged ONCE and read twice from LDS (as row h + 1 of
//     step h, as row h of step h + 1): one accumulator set, a ring of three slots, one s_waitcnt vmcnt(0) + barrier per step;
//   * LDS rows are 256 B (one pixel), the 16-byte chunks XOR-swizzled by the pixel: the fragment of tap dw reads pixel lr + dw -- consecutive rows, no conflict;
//   * epilogue per step: bias / ReLU, bf16, the 2 x 128-pixel output rows transposed through 32 KB of LDS (the four phases interleave there), whole 128-byte
//     NHWC pixel rows per 8 lanes; fused per-(image, channel) statistics in the store pass (a thread keeps the same 8 channels for all of its pixels).
// 64 MFMAs (waves 0 / 1) and 80 (waves 2 / 3) per step and wave: 2 560 matrix-pipe cycles for 49 KB of HBM traffic -- the kernel is HBM-bound by design.
// Same descriptor and packed weights as the four-phase paths (n_phase = 4, per-phase tap lists and weight column bases): no host change beyond the dispatch.
#include "conv_args.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((address_space(3))) char lds_char_t;
typedef __attribute__((address_space(3))) const bf16x8_t lds_frag_t;

template <int V> struct S2UIC { static constexpr int value = V; };

struct S2uArgs {
    ConvArgs a;
    int R, nstrips, segs;
    int kb[4][2][2];                 // weight column base of phase p's tap at input offset (dh, dw); -1 = the phase has no such tap
};

constexpr int S2U_SLOT = 68 * 256;                    // one input row segment: 65 pixels used, 17 DMA pieces of 4 pixels
constexpr int S2U_TILE = 3 * S2U_SLOT;                // epilogue tile: 2 output rows x 128 pixels x 64 channels bf16
constexpr int S2U_BIAS = S2U_TILE + 2 * 128 * 128;    // the channel tile's bias (64 floats)
constexpr size_t S2U_LDS = (size_t)S2U_BIAS + 256;
static_assert(S2U_LDS <= 160 * 1024, "LDS of one CU");

template <bool STATS, bool RELU>
__global__ void __launch_bounds__(256) conv_s2u_kernel(const S2uArgs sa) {
    const ConvArgs &a = sa.a;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_char_t *lds = (lds_char_t *)smem_raw;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int cg = wave & 1;                            // 32-channel half of the 64-channel tile
    const bool role1 = wave >= 2;                       // waves 2, 3: phases (0,0), (0,1), (1,0); waves 0, 1: phase (1,1)

    int b = blockIdx.x;
    const int strip = b % sa.nstrips; b /= sa.nstrips;
    const int seg = b % sa.segs; b /= sa.segs;
    const int tn = b % a.tiles_n;
    const int n = b / a.tiles_n;
    const int h0 = strip * sa.R;
    const int w0 = seg * 64;

    // ---- the wave's weights: unit u = 32 output channels x 128 contracted channels = 8 K sub-steps of v_mfma_f32_32x32x16_bf16
    //   role 0: u = dh*2 + dw of phase 3                      role 1: u = 0: p0 (0,0) | 1: p1 (0,0) | 2: p1 (0,1) | 3: p2 (0,0) | 4: p2 (1,0)
    bf16x8_t W[5][8];
    {
        const bf16_t *wp = a.w_hi + (size_t)(tn * 64 + cg * 32 + lr) * a.w_kstride + lh * 8;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            int kb;
            if (!role1) kb = u < 4 ? sa.kb[3][u >> 1][u & 1] : -1;
            else kb = u == 0 ? sa.kb[0][0][0] : (u == 1 ? sa.kb[1][0][0] : (u == 2 ? sa.kb[1][0][1] : (u == 3 ? sa.kb[2][0][0] : sa.kb[2][1][0])));
#pragma unroll
            for (int s = 0; s < 8; ++s) W[u][s] = kb >= 0 ? *reinterpret_cast<const bf16x8_t *>(wp + kb + s * 16) : bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    // ---- staging: piece = 4 pixels x 256 B; wave w issues pieces w, w + 4, .. of the 17; chunk c of pixel px sits at position c ^ (px & 15) (applied to the
    // SOURCE address).  buffer_load ... lds: the image is the buffer, row and piece in the scalar offset, ONE lane offset; pixels beyond the row are out of range
    const int ppx = lane >> 4, pch = lane & 15;
    const int psb = a.in_pstride * 2;
    const unsigned OOB = 0x80000000u;
    const size_t row_bytes = (size_t)a.Wi * psb;
    const char *in = reinterpret_cast<const char *>(a.in);
    const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(in + (size_t)n * a.Hi * row_bytes), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_none = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(in), 0, 0, 0x00020000);
    unsigned v_off[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int px = 4 * (wave + 4 * k) + ppx;          // pixel of the segment (0 .. 67)
        v_off[k] = (w0 + px < a.Wi && px <= 64) ? (unsigned)((w0 + px) * psb + ((pch ^ (px & 15)) << 4)) : OOB;
    }
    auto stage_piece = [&](auto Kc, int soff, int slot, const __amdgpu_buffer_rsrc_t rs) __attribute__((always_inline)) {
        constexpr int K = decltype(Kc)::value;
        if (K < 4 || wave == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(lds + slot * S2U_SLOT + (wave + 4 * K) * 1024), 16, (int)v_off[K], soff, 0, 0);
    };
    auto stage = [&](int r, int slot) __attribute__((always_inline)) {          // input row r (a row below the image: zeros) -> ring slot
        const __amdgpu_buffer_rsrc_t rs = r < a.Hi ? rsrc_in : rsrc_none;
        const int soff = r < a.Hi ? r * (int)row_bytes : 0;
        stage_piece(S2UIC<0>{}, soff, slot, rs); stage_piece(S2UIC<1>{}, soff, slot, rs); stage_piece(S2UIC<2>{}, soff, slot, rs);
        stage_piece(S2UIC<3>{}, soff, slot, rs); stage_piece(S2UIC<4>{}, soff, slot, rs);
    };

    // ---- fragment addressing (bytes inside a slot): lane = (pixel lr of a 32-pixel block, K half lh); tap dw reads pixel lr + dw; + j * 8192, ^ (s << 5)
    int a_dw[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) a_dw[d] = (lr + d) * 256 + ((lh ^ ((lr + d) & 15)) << 4);

    // ---- epilogue state
    lds_char_t *tile = lds + S2U_TILE;
    __attribute__((address_space(3))) float *bias_l = reinterpret_cast<__attribute__((address_space(3))) float *>(lds + S2U_BIAS);
    if (tid < 64) {
        const int co = tn * 64 + tid;
        bias_l[tid] = (a.bias && co < a.bias_n) ? a.bias[co] : 0.f;
    }
    __syncthreads();
    constexpr bool relu = RELU;                 // (a compile-time flag: as a run-time one it cost a v_max + v_cndmask per stored value)
    const int opb = a.out_pstride * 2;
    const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(a.out) + (size_t)n * a.Ho * a.Wo * opb, 0, 0x7fffffff, 0x00020000);
    // store pass: thread t owns 16-byte chunk t & 7 (channels (t & 7) * 8 ..) of tile pixels (t >> 3) + 32 it, it = 0..7: tile pixel q = output row q >> 7, pixel q & 127
    const unsigned st_off = (unsigned)((tid >> 3) * opb + (tn * 64 + (tid & 7) * 8) * 2);      // + (32 (it & 3)) pixels, + (it >> 2) output rows: scalar
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }

    f32x16_t acc[3][2];                 // role 0: acc[0] = phase 3;  role 1: acc[p] = phase p
    auto reset_acc = [&]() __attribute__((always_inline)) {
        f32x4_t bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = *reinterpret_cast<__attribute__((address_space(3))) const f32x4_t *>(bias_l + cg * 32 + q * 8 + lh * 4);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][j][r] = bv[r >> 2][r & 3];
    };
    reset_acc();

    // tile write of one phase's accumulators: output row ph of the step, pixel 2 (j*32 + lr) + pw; 8-byte unit (4 channels) XOR-swizzled by the pixel
    auto put_phase = [&](f32x16_t (&ac)[2], int ph, int pw) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = ac[j][q * 4 + e];
                    if constexpr (relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                u32x2_t pk;
                pk[0] = pack2_bf16(v[0], v[1]);
                pk[1] = pack2_bf16(v[2], v[3]);
                const int pix = ph * 128 + 2 * (j * 32 + lr) + pw;
                const int unit = (cg * 8 + q * 2 + lh) ^ ((lr & 7) << 1);
                *reinterpret_cast<__attribute__((address_space(3))) u32x2_t *>(tile + pix * 128 + unit * 8) = pk;
            }
    };
    auto epilogue = [&](int h) __attribute__((always_inline)) {
        if (!role1) put_phase(acc[0], 1, 1);
        else { put_phase(acc[0], 0, 0); put_phase(acc[1], 0, 1); put_phase(acc[2], 1, 0); }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // LDS only: the DMA of the next row stays in flight
        const int orow = (2 * h * a.Wo + 2 * w0) * opb;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int q = (tid >> 3) + 32 * it;                 // tile pixel: output row q >> 7, pixel q & 127
            const int cc = tid & 7;
            const int lrq = (q >> 1) & 31;                      // the lr of the lane that wrote this pixel (pixel = 2 (j*32 + lr) + pw)
            const int unit = (cc * 2) ^ ((lrq & 7) << 1);
            const u32x4_t v = *reinterpret_cast<__attribute__((address_space(3))) const u32x4_t *>(tile + q * 128 + unit * 8);
            __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_out, (int)st_off, orow + (it >> 2) * a.Wo * opb + 32 * (it & 3) * opb, 0);
            if constexpr (STATS) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = h16_lo_f32(v[e]), hi = h16_hi_f32(v[e]);
                    s1[2 * e] += lo; s2[2 * e] += lo * lo; s1[2 * e + 1] += hi; s2[2 * e + 1] += hi * hi;
                }
            }
        }
        reset_acc();
    };

    // ---- one step: rows h (slot sa_) and h + 1 (slot sb_) -> output rows 2h, 2h + 1; the row for step h + 1 (h + 2) is staged piece by piece in the MFMA stream
    auto read_frags0 = [&](bf16x8_t (&F)[8], int sA, int sB, int s) __attribute__((always_inline)) {       // role 0: offsets (0,0) (0,1) (1,0) (1,1) x 2 blocks
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                F[u * 2 + j] = *reinterpret_cast<lds_frag_t *>(lds + ((u >> 1) ? sB : sA) * S2U_SLOT + (a_dw[u & 1] ^ (s << 5)) + j * 8192);
    };
    auto read_frags1 = [&](bf16x8_t (&F)[8], int sA, int sB, int s) __attribute__((always_inline)) {       // role 1: offsets (0,0) (0,1) (1,0) x 2 blocks
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            F[0 + j] = *reinterpret_cast<lds_frag_t *>(lds + sA * S2U_SLOT + (a_dw[0] ^ (s << 5)) + j * 8192);
            F[2 + j] = *reinterpret_cast<lds_frag_t *>(lds + sA * S2U_SLOT + (a_dw[1] ^ (s << 5)) + j * 8192);
            F[4 + j] = *reinterpret_cast<lds_frag_t *>(lds + sB * S2U_SLOT + (a_dw[0] ^ (s << 5)) + j * 8192);
        }
    };
    auto step = [&](int sA, int sB, int soff_dma, int slot_dma, const __amdgpu_buffer_rsrc_t rs) __attribute__((always_inline)) {
        bf16x8_t FA[8], FB[8];
        if (!role1) {
            read_frags0(FA, sA, sB, 0);
            auto sub = [&](auto Sc, bf16x8_t (&Fc)[8], bf16x8_t (&Fn)[8]) __attribute__((always_inline)) {
                constexpr int s = decltype(Sc)::value;
                if constexpr (s + 1 < 8) read_frags0(Fn, sA, sB, s + 1);
                if constexpr (s < 5) stage_piece(S2UIC<(s < 5 ? s : 0)>{}, soff_dma, slot_dma, rs);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[0][j] = dl_mfma32(W[u][s], Fc[u * 2 + j], acc[0][j]);
            };
            sub(S2UIC<0>{}, FA, FB); sub(S2UIC<1>{}, FB, FA); sub(S2UIC<2>{}, FA, FB); sub(S2UIC<3>{}, FB, FA);
            sub(S2UIC<4>{}, FA, FB); sub(S2UIC<5>{}, FB, FA); sub(S2UIC<6>{}, FA, FB); sub(S2UIC<7>{}, FB, FA);
        } else {
            read_frags1(FA, sA, sB, 0);
            auto sub = [&](auto Sc, bf16x8_t (&Fc)[8], bf16x8_t (&Fn)[8]) __attribute__((always_inline)) {
                constexpr int s = decltype(Sc)::value;
                if constexpr (s + 1 < 8) read_frags1(Fn, sA, sB, s + 1);
                if constexpr (s < 5) stage_piece(S2UIC<(s < 5 ? s : 0)>{}, soff_dma, slot_dma, rs);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[0][j] = dl_mfma32(W[0][s], Fc[0 + j], acc[0][j]);
                    acc[1][j] = dl_mfma32(W[1][s], Fc[0 + j], acc[1][j]);
                    acc[2][j] = dl_mfma32(W[3][s], Fc[0 + j], acc[2][j]);
                    acc[1][j] = dl_mfma32(W[2][s], Fc[2 + j], acc[1][j]);
                    acc[2][j] = dl_mfma32(W[4][s], Fc[4 + j], acc[2][j]);
                }
            };
            sub(S2UIC<0>{}, FA, FB); sub(S2UIC<1>{}, FB, FA); sub(S2UIC<2>{}, FA, FB); sub(S2UIC<3>{}, FB, FA);
            sub(S2UIC<4>{}, FA, FB); sub(S2UIC<5>{}, FB, FA); sub(S2UIC<6>{}, FA, FB); sub(S2UIC<7>{}, FB, FA);
        }
    };

    // ---- pipeline: input row h0 + t lives in ring slot t % 3; step t needs rows t and t + 1 and stages row t + 2
    const int R = sa.R;
    stage(h0, 0);
    stage(h0 + 1, 1);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    int sA = 0, sB = 1, sC = 2;
    for (int t = 0; t < R; ++t) {
        const int h = h0 + t;
        const bool more = t + 1 < R;
        const bool real = more && h + 2 < a.Hi;              // (the row below the image is staged as zeros: the out-of-range resource)
        step(sA, sB, real ? (h + 2) * (int)row_bytes : 0, sC, real ? rsrc_in : rsrc_none);
        __builtin_amdgcn_sched_barrier(0);                   // the step's MFMAs stay in front of the epilogue that reads their accumulators
        epilogue(h);
        __builtin_amdgcn_sched_barrier(0);
        // vmcnt(8): everything but this step's 8 stores per thread -- i.e. every DMA piece of the row staged during the step (issued before the stores; gfx9
        // VMEM operations complete in issue order) -- has landed; the stores drain behind the next step's MFMAs instead of in front of the barrier
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int t3 = sA; sA = sB; sB = sC; sC = t3;
    }

    if constexpr (STATS) {
        // thread t holds channels (t & 7)*8 .. +8 summed over its pixels; the 32 threads of a channel group are lanes c, c+8, .., c+56 of the four waves
        __attribute__((address_space(3))) float *red = reinterpret_cast<__attribute__((address_space(3))) float *>(tile);      // [wave][2][64]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t1 = s1[e], t2 = s2[e];
            t1 += __shfl_xor(t1, 8, 64); t2 += __shfl_xor(t2, 8, 64);
            t1 += __shfl_xor(t1, 16, 64); t2 += __shfl_xor(t2, 16, 64);
            t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
            if (lane < 8) { red[(wave * 2 + 0) * 64 + lane * 8 + e] = t1; red[(wave * 2 + 1) * 64 + lane * 8 + e] = t2; }
        }
        __syncthreads();
        if (tid < 64 && tn * 64 + tid < a.Co) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { t1 += red[(w * 2 + 0) * 64 + tid]; t2 += red[(w * 2 + 1) * 64 + tid]; }
            float *o = a.stats_part + ((size_t)(n * a.stats_nchunks + seg * sa.nstrips + strip) * 2) * a.Co + tn * 64 + tid;
            o[0] = t1;
            o[a.Co] = t2;
        }
    }
}

// strip height: a divisor of Hq that gives the grid about one workgroup per CU (each strip re-reads one halo row, so taller is cheaper)
static int s2u_strip_rows(const ConvArgs &a) {
    const int per_img = a.N * (a.Wq / 64) * (a.Co / 64);
    int best = 0;
    for (int R = 1; R <= a.Hq; ++R) {
        if (a.Hq % R) continue;
        const int wgs = per_img * (a.Hq / R);
        if (best == 0 || wgs >= 240) best = R;
        if (wgs < 240) break;
    }
    return best;
}

// fills kb[p][dh][dw]; false when the descriptor is not the 2 x 2-phase form of a 3 x 3 stride-2 layer (phase (ph, pw) has the taps dh in {0 .. ph}, dw in {0 .. pw})
static bool s2u_tap_table(const ConvArgs &a, int (&kb)[4][2][2]) {
    for (int p = 0; p < 4; ++p) {
        if (a.phase_oh[p] != (p >> 1) || a.phase_ow[p] != (p & 1)) return false;
        for (int dh = 0; dh < 2; ++dh)
            for (int dw = 0; dw < 2; ++dw) kb[p][dh][dw] = -1;
        const int nt = a.phase_tap_begin[p + 1] - a.phase_tap_begin[p];
        if (nt != ((p >> 1) + 1) * ((p & 1) + 1)) return false;
        for (int t = 0; t < nt; ++t) {
            const int16_t tp = a.taps[a.phase_tap_begin[p] + t];
            const int dh = (int)(int8_t)(tp & 0xff), dw = (int)(int8_t)((tp >> 8) & 0xff);
            if (dh < 0 || dh > (p >> 1) || dw < 0 || dw > (p & 1) || kb[p][dh][dw] >= 0) return false;
            kb[p][dh][dw] = a.phase_kbase[p] + t * a.Ci;
        }
    }
    return true;
}

// The layers this kernel serves: four sub-pixel phases (out_step 2, in_step 1) of a 3 x 3 stride-2 layer, exact 2x geometry, zero padding, exactly 128 contracted
// channels, output channels a multiple of 64, phase-grid rows that are multiples of 64 pixels, bf16, no split-K / raw accumulators / input activation / fused
// norm-backward reductions.
bool s2u_eligible(const ConvArgs &a) {
    if (a.n_phase != 4 || a.splitk != 1 || a.raw_out || a.in_step != 1 || a.out_step != 2) return false;
    if (a.Ho != 2 * a.Hq || a.Wo != 2 * a.Wq || a.Hi != a.Hq || a.Wi != a.Wq || (a.Wq & 63)) return false;
    if (a.Ci != 128 || a.Co < 64 || (a.Co & 63) || a.pad_mode != DL_PAD_ZERO || a.bn_y != nullptr || a.in_act != DL_ACT_NONE || a.epi_old) return false;
    if (a.act != DL_ACT_NONE && a.act != DL_ACT_RELU) return false;
    if ((size_t)a.Hi * a.Wi * (size_t)a.in_pstride * 2 >= ((size_t)1 << 31) || (size_t)a.Ho * a.Wo * (size_t)a.out_pstride * 2 >= ((size_t)1 << 31)) return false;      // 32-bit offsets
    int kb[4][2][2];
    return s2u_tap_table(a, kb) && s2u_strip_rows(a) > 0;
}

// chunks of fused norm statistics per image: one per workgroup (row segment x strip)
int s2u_stats_chunks(const ConvArgs &a) {
    const int R = s2u_strip_rows(a);
    return R > 0 ? (a.Wq / 64) * (a.Hq / R) : 0;
}

int launch_conv_s2u(const ConvArgs &a0, hipStream_t stream) {
    S2uArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.a = a0;
    ConvArgs &a = sa.a;
    if (!s2u_tap_table(a, sa.kb)) DL_FAIL("dl_conv_forward(s2u): not the four-phase form of a 3x3 stride-2 layer");
    sa.R = s2u_strip_rows(a);
    if (sa.R <= 0) DL_FAIL("dl_conv_forward(s2u): no strip height for Hq=%d", a.Hq);
    sa.nstrips = a.Hq / sa.R;
    sa.segs = a.Wq / 64;
    a.tiles_n = a.Co / 64;
    a.tiles_m = a.N * sa.segs * sa.nstrips;
    if (a.stats_part && a.stats_nchunks != sa.segs * sa.nstrips) DL_FAIL("dl_conv_forward(s2u): statistics chunks %d != %d", a.stats_nchunks, sa.segs * sa.nstrips);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipSuccess;
        const void *fns[4] = {reinterpret_cast<const void *>(conv_s2u_kernel<false, false>), reinterpret_cast<const void *>(conv_s2u_kernel<false, true>),
                              reinterpret_cast<const void *>(conv_s2u_kernel<true, false>), reinterpret_cast<const void *>(conv_s2u_kernel<true, true>)};
        for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)S2U_LDS);
        if (e != hipSuccess) DL_FAIL("dl_conv_forward(s2u): hipFuncSetAttribute(%zu): %s", S2U_LDS, hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(a.tiles_m * a.tiles_n, 1);
    const bool relu = a.act == DL_ACT_RELU;
    if (a.stats_part) {
        if (relu) hipLaunchKernelGGL((conv_s2u_kernel<true, true>), grid, dim3(256), S2U_LDS, stream, sa);
        else hipLaunchKernelGGL((conv_s2u_kernel<true, false>), grid, dim3(256), S2U_LDS, stream, sa);
    } else {
        if (relu) hipLaunchKernelGGL((conv_s2u_kernel<false, true>), grid, dim3(256), S2U_LDS, stream, sa);
        else hipLaunchKernelGGL((conv_s2u_kernel<false, false>), grid, dim3(256), S2U_LDS, stream, sa);
    }
    DL_CHECK_LAUNCH("dl_conv_forward(s2u)");
    return 0;
}
