// conv_w4.hip -- the ResnetBlock conv (3x3, stride 1, image rows of 128 pixels, Cin % 64 == 0, Co % 256 == 0, bf16) on a
// one-wave-per-SIMD tile: forward and data gradient of networks.py:467-513 (the reference's ResnetBlock.conv_block).
//
// Why a second kernel for the dominant shape: the 8-phase kernel (conv_gemm.hip) runs 8 waves of
// 128 px x 64 ch; per 64-wide K step it moves 64 KB global->LDS and reads 192 KB LDS->registers for 2048 matrix-pipe cycles, and
// its two wave groups exchange roles across eight barriers per step -- every phase costs max(MFMA block, partner's load block).
// Here the 256 px x 256 ch tile is computed by FOUR waves of 128 px x 128 ch (16 accumulators of v_mfma_f32_32x32x16_bf16 =
// 256 registers, the whole AGPR half of the 512-entry file; one wave per SIMD):
//   * LDS->register traffic per K step drops to 128 KB (-33 %): a 32-row fragment feeds 4 MFMAs instead of 2;
//   * a K=16 sub-step needs 8 fragments (32 VGPRs) for 16 MFMAs = 512 matrix-pipe cycles, so the fragments are double-buffered
//     in registers and every wave overlaps its OWN ds_reads and DMA issue with its OWN MFMAs: one barrier per K step, no
//     role exchange;
//   * kernel-column reuse: the three kw taps of one (64-channel chunk, kh) read ONE staged slab of 2 image rows x 128 pixels at
//     row offsets dw: 32 KB of activation DMA per THREE K steps, 43 KB per step instead of 64 KB.  The slab has NO pad pixels
//     (they would not fit, below): the one fragment lane per wave that falls off the image row (pixel -1 / 128) is zeroed in
//     registers -- the kw order is a template parameter (FLIP: the data gradient walks dw = +1, 0, -1), so only the two outer
//     taps pay 4 v_cndmask per sub-step;
//   * K order (chunk, kh, kw): the 9 taps of a 64-channel chunk run back to back (the slab of a chunk stays in the XCD's L2).
// LDS: three weight buffers (3 x 32 KB) + two slabs (2 x 32 KB) = 163 840 B = all 160 KB.  Layout [W0 | X0 | X1 | W1 | W2]: the
// off-image fragment rows (slab row -1 / 256) then fall into a neighbouring buffer -- a harmless read, zeroed afterwards.
// Pipeline: K step t = 3 g + kw uses weight buffer kw (static); the DMA of W(t+2) and of slab(g+1) (4 pieces in each of the
// first two steps of group g, issued BEFORE the weight pieces) is issued during step t, so every piece has a whole K step
// (> 2000 matrix-pipe cycles) to land: the step ends with s_waitcnt vmcnt(8) -- the 8 youngest pieces, W(t+2), stay in
// flight -- lgkmcnt(0), s_barrier.  Sub-step 3's MFMAs run after the barrier and cover the first fragment reads of step
// t+1.  The steps past the end re-stage the last group into idle buffers (no branches in the loop).  r04 first version (two
// weight buffers, vmcnt(0) per step, padded slabs): 144.5 us in-step against 157.7 us for the 8-phase kernel
// (profiles/r04/w4_first_look.txt); DMA-only 79.5 us, MFMA-only 103.9 us, no-DMA 118.6 us.
// Epilogue: as tile_epilogue_lds of conv_gemm.hip for the 32x32 accumulator layout -- bias / ReLU, bf16 tile transposed
// through the dead operand LDS, whole NHWC pixel rows of 16 B per lane, fused per-(image, channel) statistics (DPP row sums).
#include "conv_args.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((address_space(3))) char lds_char_t;
typedef __attribute__((address_space(3))) const bf16x8_t lds_frag_t;

__device__ __attribute__((aligned(64))) unsigned char g_w4_zero_page[64];

template <int V> struct W4IC { static constexpr int value = V; };

__device__ __forceinline__ float w4_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
    return v;
}

constexpr int W4_BUF = 256 * 128;                     // one weight buffer (256 output channels x 64 K) = one slab (2 x 128 pixels x 64 channels)
constexpr int W4_X0 = 1 * W4_BUF;                     // slab s at W4_X0 + s * W4_BUF
__host__ __device__ constexpr int w4_wofs(int b) { return b == 0 ? 0 : (b == 1 ? 3 * W4_BUF : 4 * W4_BUF); }
constexpr size_t W4_LOOP_LDS = (size_t)5 * W4_BUF;
constexpr size_t W4_EPI_LDS = (size_t)256 * 512 + 256 * sizeof(int) + (size_t)4 * 2 * 256 * sizeof(float) + 256 * sizeof(float);
constexpr size_t W4_LDS = W4_LOOP_LDS > W4_EPI_LDS ? W4_LOOP_LDS : W4_EPI_LDS;
static_assert(W4_LDS <= 160 * 1024, "the whole LDS of a CU");

// FLIP: kw taps ordered dw = +1, 0, -1 (data gradient) instead of -1, 0, +1.
// VAR bit 0: DMA pieces and fragment reads in SEPARATE MFMA shadows (2 reads after the even MFMAs, 1 piece after the odd ones) instead of one read
//            + one piece per shadow;  bit 1: the barrier of a step sits in the MIDDLE of sub-step 3 (after 8 of its MFMAs) instead of in front of it.
// ABL != 0: timing-only ablations (results wrong by construction): 1 = no DMA in the loop, 2 = DMA only, 3 = MFMAs only, 4 = no K loop (prologue
//           + epilogue), 5 = DMA issued but never waited for
template <bool FLIP, int VAR, int ABL>
__global__ void __launch_bounds__(256) conv_gemm_w4_kernel(const ConvArgs a) {
    constexpr bool SEP = (VAR & 1) != 0, LATE = (VAR & 2) != 0;
    constexpr bool DMA_ON = ABL != 1 && ABL != 3;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_char_t *lds = (lds_char_t *)smem_raw;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % a.tiles_n, tm = bid / a.tiles_n;
    const int nch = a.Ci >> 6;
    const int G = ABL == 4 ? 0 : 3 * nch;       // (chunk, kh) groups of three K steps

    // taps are ordered (kh, kw); dh is constant per kernel row (w4_eligible)
    int dhs[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) dhs[k] = (int)(int8_t)(a.taps[3 * k] & 0xff);

    // ---- staging geometry.  Weights: instruction i of wave w fills buffer rows (w*8 + i)*8 .. +8 (output channels).  Activations: waves 0,1 stage
    // image row 0 of the tile (columns 0-63 / 64-127), waves 2,3 image row 1; instruction i covers 8 pixels.  LDS rows are 128 B = 8 chunks of 16 B;
    // chunk c of row r sits at position c ^ ((r >> 1) & 7): the DMA image is lane-linear, so the permutation is applied to the SOURCE address.
    const int lrow = lane >> 3, lcp = lane & 7;
    const int R = wave >> 1, cbase = (wave & 1) * 64;
    const int HWq = a.Hq * a.Wq;
    const int m0 = tm * 256;
    const int n_img = m0 / HWq;
    const int h0 = (m0 - n_img * HWq) >> 7;
    uint32_t x_off[8], w_off[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int col = cbase + 8 * i + lrow, xrow = R * 128 + col;
        x_off[i] = (uint32_t)(((R * a.Wi + col) * a.in_pstride + (lcp ^ ((xrow >> 1) & 7)) * 8) * 2);
        const int s = (wave * 8 + i) * 8 + lrow;
        w_off[i] = (uint32_t)((s * a.w_kstride + (lcp ^ ((s >> 1) & 7)) * 8) * 2);
    }
    const char *xg = reinterpret_cast<const char *>(a.in) + ((size_t)(n_img * a.Hi + h0) * a.Wi) * (size_t)a.in_pstride * 2;
    const char *wg = reinterpret_cast<const char *>(a.w_hi) + ((size_t)(tn * 256) * a.w_kstride + a.phase_kbase[0]) * 2;
    const char *zero = reinterpret_cast<const char *>(g_w4_zero_page);
    const int x_dst0 = W4_X0 + (R * 128 + cbase) * 128;              // + slab * BUF + i * 1024
    const int w_dst0 = wave * 8 * 1024;                              // + w4_wofs(buf) + i * 1024
    const ptrdiff_t tap_bytes = (ptrdiff_t)a.Ci * 2;                 // one kernel tap further along a packed weight row

    // uniform part of the activation source address of group (chunk c, kernel row kh), and whether that image row exists for this wave
    auto x_group_base = [&](int c, int kh, bool &valid) __attribute__((always_inline)) {
        const int dh = kh == 0 ? dhs[0] : (kh == 1 ? dhs[1] : dhs[2]);
        valid = (unsigned)(h0 + R + dh) < (unsigned)a.Hi;
        return xg + ((ptrdiff_t)dh * a.Wi * a.in_pstride + c * 64) * 2;
    };
    // VAR bit 2 / bit 3: weight / activation pieces by buffer_load ... lds (resource in SGPRs, loop-invariant 32-bit lane offset, the per-step
    // part of the address in the scalar offset) instead of global_load_lds with a 64-bit VALU address add (and two v_cndmask for the zero
    // page) per piece: with one wave per SIMD every VALU instruction next to a DMA piece comes out of the MFMA issue stream.  An image row
    // outside the tensor is fetched through a resource with num_records = 0: out-of-range buffer loads deliver zeros.
    constexpr bool WBUF = (VAR & 4) != 0, XBUF = (VAR & 8) != 0;
    const ptrdiff_t row_bytes = (ptrdiff_t)a.Wi * a.in_pstride * 2;
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wg), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xg - row_bytes), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xg - row_bytes), 0, 0, 0x00020000);
    auto dma_x = [&](auto I, const char *base, bool valid, int slab) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        if constexpr (XBUF) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(valid ? rsrc_x : rsrc_0, (__attribute__((address_space(3))) void *)(lds + x_dst0 + slab * W4_BUF + i * 1024), 16,
                                                     (int)x_off[i], (int)(base - (xg - row_bytes)), 0, 0);
            return;
        }
        const char *src = valid ? base + x_off[i] : zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds + x_dst0 + slab * W4_BUF + i * 1024), 16, 0, 0);
    };
    auto dma_w = [&](auto I, const char *base, auto BUF) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value, buf = decltype(BUF)::value;
        if constexpr (WBUF) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void *)(lds + w_dst0 + w4_wofs(buf) + i * 1024), 16, (int)w_off[i],
                                                     (int)(base - wg), 0, 0);
            return;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + w_off[i]),
                                         (__attribute__((address_space(3))) void *)(lds + w_dst0 + w4_wofs(buf) + i * 1024), 16, 0, 0);
    };

    // ---- fragment addressing (bytes): lane = (row lr of a 32-row block, K half lh of a 16-wide sub-step)
    const int lr = lane & 31, lh = lane >> 5;
    const int aw = (wn * 128 + lr) * 128 + ((lh ^ ((lr >> 1) & 7)) << 4);
    int axk[3];                                                         // activation fragment base of kw = 0, 1, 2 (slab row shift dw = -1, 0, +1 or reversed)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int row = wm * 128 + lr + (FLIP ? 1 - k : k - 1);
        axk[k] = W4_X0 + row * 128 + ((lh ^ ((row >> 1) & 7)) << 4);
    }
    const bool edge_lo = lr == 0, edge_hi = lr == 31;                   // the lane whose pixel is column -1 (block 0, dw = -1) / 128 (block 3, dw = +1)

    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    bf16x8_t FA[8], FB[8];
    if constexpr (ABL == 2 || ABL == 3) {
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            FA[f] = bf16x8_t{(short)(0x3f80 + lane), (short)(0x3f00 + f), 0x3e80, 0x3f81, (short)0xbf80, 0x3f10, 0x3e90, 0x3f91};
            FB[f] = bf16x8_t{(short)(0x3f00 + lane), (short)(0x3f80 + f), 0x3e90, 0x3f01, (short)0xbf00, 0x3f20, 0x3e80, 0x3f11};
        }
    }
    auto read_one = [&](auto FI, int wp, int xp, bf16x8_t (&F)[8]) __attribute__((always_inline)) {
        constexpr int f = decltype(FI)::value;
        if constexpr (ABL == 2 || ABL == 3) return;
        if constexpr (f < 4) F[f] = *reinterpret_cast<lds_frag_t *>(lds + wp + f * 4096);
        else F[f] = *reinterpret_cast<lds_frag_t *>(lds + xp + (f - 4) * 4096);
    };
    auto mma_one = [&](auto MI, const bf16x8_t (&F)[8]) __attribute__((always_inline)) {
        // MFMA 0 of a sub-step multiplies the two fragments that were read LAST (W3, X3): the compiler's wait for them (= for every read of the
        // previous sub-step: LDS returns in order) then sits in front of the sub-step, before any read of the next one is issued.  With the
        // natural order hipcc put an s_waitcnt lgkmcnt(0) BEHIND the first two fresh reads of every sub-step (a full LDS round trip, 4 x per step)
        constexpr int m = decltype(MI)::value, i = 3 - (m >> 2), j = 3 - (m & 3);
        if constexpr (ABL == 2) return;
        acc[i][j] = dl_mfma32(F[i], F[4 + j], acc[i][j]);
    };
    // the fragment lane that fell off the image row reads a neighbouring buffer: zero it (SH = slab shift of the step the fragments belong to)
    auto fix_edge = [&](auto SHc, bf16x8_t (&F)[8]) __attribute__((always_inline)) {
        constexpr int SH = decltype(SHc)::value;
        if constexpr (ABL == 2 || ABL == 3) return;
        if constexpr (SH == 0) { if (edge_lo) F[4] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0}; }
        if constexpr (SH == 2) { if (edge_hi) F[7] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0}; }
    };
    // MFMAs [M0, M1) of a K=16 sub-step on Fc; the fragment reads of the NEXT sub-step (into Fn, bases wp / xp) ride in their shadows when READS;
    // hook(m) is called after MFMA m (DMA issue slots).  Source order = intended issue order; pinned by the sched_group_barriers that follow.
    auto mfma_range = [&](auto M0c, auto M1c, auto READSc, const bf16x8_t (&Fc)[8], bf16x8_t (&Fn)[8], int wp, int xp, auto &&hook) __attribute__((always_inline)) {
        constexpr int M0 = decltype(M0c)::value, M1 = decltype(M1c)::value;
        constexpr bool READS = decltype(READSc)::value != 0;
        constexpr int RBASE = decltype(READSc)::value == 2 ? 8 : 0;      // READS == 2: the reads start at MFMA 8 (second half of a sub-step)
        auto rd = [&](auto K) __attribute__((always_inline)) {           // K-th read of the sub-step: W0 X0 W1 X1 W2 X2 W3 X3
            constexpr int k = decltype(K)::value;
            read_one(W4IC<(k & 1) ? 4 + (k >> 1) : (k >> 1)>{}, wp, xp, Fn);
        };
        auto one = [&](auto MI) __attribute__((always_inline)) {
            constexpr int m = decltype(MI)::value;
            if constexpr (m >= M0 && m < M1) {
                mma_one(MI, Fc);
                if constexpr (READS) {
                    constexpr int q = m - RBASE;
                    if constexpr (RBASE == 8) {                      // second half of a sub-step: two reads in each of the shadows 8..11
                        if constexpr (q >= 0 && q < 4) { rd(W4IC<(q >= 0 && q < 4) ? 2 * q : 0>{}); rd(W4IC<(q >= 0 && q < 4) ? 2 * q + 1 : 0>{}); }
                    } else if constexpr (SEP) {
                        if constexpr (q >= 0 && q < 8 && (q & 1) == 0) { rd(W4IC<(q >= 0 && q < 8) ? q : 0>{}); rd(W4IC<(q >= 0 && q < 8) ? q + 1 : 0>{}); }
                    } else {
                        if constexpr (q >= 0 && q < 8) rd(W4IC<(q >= 0 && q < 8) ? q : 0>{});
                    }
                }
                hook(MI);
            }
        };
        one(W4IC<0>{}); one(W4IC<1>{}); one(W4IC<2>{}); one(W4IC<3>{}); one(W4IC<4>{}); one(W4IC<5>{}); one(W4IC<6>{}); one(W4IC<7>{});
        one(W4IC<8>{}); one(W4IC<9>{}); one(W4IC<10>{}); one(W4IC<11>{}); one(W4IC<12>{}); one(W4IC<13>{}); one(W4IC<14>{}); one(W4IC<15>{});
    };
    // pin the issue order of a range written by mfma_range: per MFMA shadow one MFMA, then its reads / its DMA piece
    auto pin = [&](auto M0c, auto M1c, auto READSc, auto NDc) __attribute__((always_inline)) {
        constexpr int M0 = decltype(M0c)::value, M1 = decltype(M1c)::value, ND = decltype(NDc)::value;
        constexpr bool READS = decltype(READSc)::value != 0;
        constexpr int RBASE = decltype(READSc)::value == 2 ? 8 : 0;
#pragma unroll
        for (int m = M0; m < M1; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            const int q = m - RBASE;
            if (READS && q >= 0 && q < 8) {
                if (RBASE == 8) { if (q < 4) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
                else if (SEP) { if ((q & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
                else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            // DMA slots (see dma_slot below): SEP: after MFMAs 1, 3, 5, 7;  else after MFMAs 1, 4, 7, 10
            const bool slot = SEP ? ((m & 1) == 1 && (m >> 1) < ND) : (m % 3 == 1 && m / 3 < ND);
            if (slot) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
    };
    // index of the DMA piece issued after MFMA m of a sub-step (-1: none)
    auto dma_slot = [](int m) constexpr { return SEP ? (((m & 1) == 1 && (m >> 1) < 4) ? (m >> 1) : -1) : ((m % 3 == 1 && m / 3 < 4) ? m / 3 : -1); };
    auto nohook = [](auto) __attribute__((always_inline)) {};

    // ---- running source pointers: group g and group g + 1 (clamped to the last group)
    int c = 0, kh = 0;
    const char *wgrp = wg, *wgrp_n = wg;            // weights of (c, kh, kw = 0) of this / the next group
    const char *xb_n = xg;
    bool xv_n = false;
    auto advance = [&](int g, int &cn, int &khn) __attribute__((always_inline)) {
        cn = c; khn = kh + 1;
        if (khn == 3) { khn = 0; cn = c + 1; }
        if (g + 1 >= G) { cn = c; khn = kh; }
        wgrp_n = wg + ((ptrdiff_t)(khn * 3) * a.Ci + cn * 64) * 2;
        xb_n = x_group_base(cn, khn, xv_n);
    };

    // ---- prologue: slab of group 0, weights of steps 0 and 1
    if (G > 0) {
        bool v0;
        const char *xb0 = x_group_base(0, 0, v0);
        dma_x(W4IC<0>{}, xb0, v0, 0); dma_x(W4IC<1>{}, xb0, v0, 0); dma_x(W4IC<2>{}, xb0, v0, 0); dma_x(W4IC<3>{}, xb0, v0, 0);
        dma_x(W4IC<4>{}, xb0, v0, 0); dma_x(W4IC<5>{}, xb0, v0, 0); dma_x(W4IC<6>{}, xb0, v0, 0); dma_x(W4IC<7>{}, xb0, v0, 0);
        dma_w(W4IC<0>{}, wg, W4IC<0>{}); dma_w(W4IC<1>{}, wg, W4IC<0>{}); dma_w(W4IC<2>{}, wg, W4IC<0>{}); dma_w(W4IC<3>{}, wg, W4IC<0>{});
        dma_w(W4IC<4>{}, wg, W4IC<0>{}); dma_w(W4IC<5>{}, wg, W4IC<0>{}); dma_w(W4IC<6>{}, wg, W4IC<0>{}); dma_w(W4IC<7>{}, wg, W4IC<0>{});
        const char *w1 = wg + tap_bytes;
        dma_w(W4IC<0>{}, w1, W4IC<1>{}); dma_w(W4IC<1>{}, w1, W4IC<1>{}); dma_w(W4IC<2>{}, w1, W4IC<1>{}); dma_w(W4IC<3>{}, w1, W4IC<1>{});
        dma_w(W4IC<4>{}, w1, W4IC<1>{}); dma_w(W4IC<5>{}, w1, W4IC<1>{}); dma_w(W4IC<6>{}, w1, W4IC<1>{}); dma_w(W4IC<7>{}, w1, W4IC<1>{});
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // slab 0, weights 0 visible (weights 1 still in flight: __syncthreads() would drain them)

    if (G > 0) {
        const int wp = aw + w4_wofs(0), xp = axk[0];
        read_one(W4IC<0>{}, wp, xp, FA); read_one(W4IC<4>{}, wp, xp, FA); read_one(W4IC<1>{}, wp, xp, FA); read_one(W4IC<5>{}, wp, xp, FA);
        read_one(W4IC<2>{}, wp, xp, FA); read_one(W4IC<6>{}, wp, xp, FA); read_one(W4IC<3>{}, wp, xp, FA); read_one(W4IC<7>{}, wp, xp, FA);
    }

    // K step t = 3 g + KW of group g: see the header comment
    auto step = [&](auto KWc, int g) __attribute__((always_inline)) {
        constexpr int KW = decltype(KWc)::value;
        constexpr int SH = FLIP ? 2 - KW : KW, SHN = FLIP ? 2 - (KW + 1) % 3 : (KW + 1) % 3;
        const int xs = (g & 1) * W4_BUF, xsn = ((g + 1) & 1) * W4_BUF;
        const int wcur = aw + w4_wofs(KW), xcur = axk[KW] + xs;
        const int wnext = aw + w4_wofs((KW + 1) % 3);
        const int xnext = KW < 2 ? axk[(KW + 1) % 3] + xs : axk[0] + xsn;
        // W(t+2): KW 0 -> (g, kw 2), KW 1 -> (g+1, kw 0), KW 2 -> (g+1, kw 1); it goes to the buffer step t-1 has just released
        const char *wsrc = KW == 0 ? wgrp + 2 * tap_bytes : (KW == 1 ? wgrp_n : wgrp_n + tap_bytes);
        const int xslab = (g + 1) & 1;
        // DMA pieces of this step in issue order: [4 slab pieces (KW 0: 0-3, KW 1: 4-7)] then 8 weight pieces; 4 per sub-step
        auto piece = [&](auto Pc) __attribute__((always_inline)) {
            constexpr int p = decltype(Pc)::value;
            if constexpr (!DMA_ON) return;
            constexpr int NX = KW < 2 ? 4 : 0;
            if constexpr (p < NX) dma_x(W4IC<(KW == 1 ? 4 : 0) + (p < NX ? p : 0)>{}, xb_n, xv_n, xslab);
            else if constexpr (p - NX < 8) dma_w(W4IC<(p - NX >= 0 && p - NX < 8) ? p - NX : 0>{}, wsrc, W4IC<(KW + 2) % 3>{});
        };
        auto hook_s = [&](auto Sc) __attribute__((always_inline)) {
            return [&](auto MI) __attribute__((always_inline)) {
                constexpr int s = decltype(Sc)::value, m = decltype(MI)::value;
                constexpr int k = dma_slot(m);
                if constexpr (k >= 0) piece(W4IC<(k >= 0 ? s * 4 + k : 0)>{});
            };
        };
        constexpr int NP = KW < 2 ? 12 : 8;                               // pieces of this step
        constexpr int ND0 = 4, ND1 = 4, ND2 = NP - 8;
        // sched_barrier(0) at every sub-step boundary: without it hipcc hoists the edge fix of the NEXT sub-step's fragment up to the read that
        // fetches it (an s_waitcnt lgkmcnt(0) on a fresh read in the middle of the MFMA stream)
        __builtin_amdgcn_sched_barrier(0);
        fix_edge(W4IC<SH>{}, FA);
        mfma_range(W4IC<0>{}, W4IC<16>{}, W4IC<1>{}, FA, FB, wcur ^ (1 << 5), xcur ^ (1 << 5), hook_s(W4IC<0>{}));
        pin(W4IC<0>{}, W4IC<16>{}, W4IC<1>{}, W4IC<DMA_ON ? ND0 : 0>{});
        __builtin_amdgcn_sched_barrier(0);
        fix_edge(W4IC<SH>{}, FB);
        mfma_range(W4IC<0>{}, W4IC<16>{}, W4IC<1>{}, FB, FA, wcur ^ (2 << 5), xcur ^ (2 << 5), hook_s(W4IC<1>{}));
        pin(W4IC<0>{}, W4IC<16>{}, W4IC<1>{}, W4IC<DMA_ON ? ND1 : 0>{});
        __builtin_amdgcn_sched_barrier(0);
        fix_edge(W4IC<SH>{}, FA);
        if constexpr (ND2 > 0) mfma_range(W4IC<0>{}, W4IC<16>{}, W4IC<1>{}, FA, FB, wcur ^ (3 << 5), xcur ^ (3 << 5), hook_s(W4IC<2>{}));
        else mfma_range(W4IC<0>{}, W4IC<16>{}, W4IC<1>{}, FA, FB, wcur ^ (3 << 5), xcur ^ (3 << 5), nohook);
        pin(W4IC<0>{}, W4IC<16>{}, W4IC<1>{}, W4IC<DMA_ON ? ND2 : 0>{});
        __builtin_amdgcn_sched_barrier(0);
        fix_edge(W4IC<SH>{}, FB);
        // everything but the 8 youngest pieces (= W(t+2)) has landed: W(t+1), the slab pieces of this step; this wave's reads of buffer t are complete
        if constexpr (LATE) {
            mfma_range(W4IC<0>{}, W4IC<8>{}, W4IC<0>{}, FB, FA, wnext, xnext, nohook);
            pin(W4IC<0>{}, W4IC<8>{}, W4IC<0>{}, W4IC<0>{});
            __builtin_amdgcn_sched_barrier(0);          // keeps those 8 MFMAs in front of the barrier (register-only instructions may cross an asm)
        }
        if constexpr (ABL == 5) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if constexpr (DMA_ON) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if constexpr (LATE) {
            mfma_range(W4IC<8>{}, W4IC<16>{}, W4IC<2>{}, FB, FA, wnext, xnext, nohook);
            pin(W4IC<8>{}, W4IC<16>{}, W4IC<2>{}, W4IC<0>{});
        } else {
            mfma_range(W4IC<0>{}, W4IC<16>{}, W4IC<1>{}, FB, FA, wnext, xnext, nohook);
            pin(W4IC<0>{}, W4IC<16>{}, W4IC<1>{}, W4IC<0>{});
        }
        (void)SHN;
    };

    for (int g = 0; g < G; ++g) {
        int cn, khn;
        advance(g, cn, khn);
        step(W4IC<0>{}, g);
        step(W4IC<1>{}, g);
        step(W4IC<2>{}, g);
        c = cn; kh = khn; wgrp = wgrp_n;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();             // LDS is dead from here on (the epilogue reuses it)

    // ---- epilogue.  acc[i][j][r] = out[pixel = wm*128 + j*32 + lr][channel = wn*128 + i*32 + 8*(r>>2) + 4*lh + (r&3)]
    lds_char_t *tile = lds;                                                         // [256 pixels][256 channels] bf16, 8-byte units XOR-swizzled by the pixel
    __attribute__((address_space(3))) int *rowtab = reinterpret_cast<__attribute__((address_space(3))) int *>(lds + 256 * 512);
    __attribute__((address_space(3))) float *red = reinterpret_cast<__attribute__((address_space(3))) float *>(lds + 256 * 512 + 256 * 4);       // [wave][2][256]
    // the tile's 256 bias values reach the accumulator pass through LDS (r05): fetched inside the pass, hipcc had made them four guarded dword loads and an
    // s_waitcnt vmcnt(0) per (i, q); 64 registers of preloaded float4s made the allocator spill.  (Isolated 20-launch timings seemed to show a bias costing
    // 9-12 us whatever its fetch; a probe that repeated the SAME launch found 138.8 us first and 122.6 us six measurements later -- the clock drifts for seconds
    // under the power cap, profiles/r05/bias_probe_clock_drift.txt -- so effects below ~10 us are read from the in-step rocprof averages only.)
    __attribute__((address_space(3))) float *bias_lds = reinterpret_cast<__attribute__((address_space(3))) float *>(lds + 256 * 512 + 256 * 4 + 8 * 256 * 4);
    const bool want_stats = a.stats_part != nullptr && a.bn_y == nullptr;
    {
        const int m = m0 + tid;
        int opix = -1;
        if (m < a.Mtot) {
            const int n = m / HWq, rem = m - n * HWq;
            const int hq = rem / a.Wq, wq = rem - hq * a.Wq;
            opix = (n * a.Ho + hq) * a.Wo + wq;
        }
        rowtab[tid] = opix;
        const int cb = tn * 256 + tid;
        bias_lds[tid] = (a.bias && cb < a.bias_n) ? a.bias[cb] : 0.f;
    }
    __syncthreads();
    // ReLU as a COMPILE-TIME flag of two copies of this pass (r05: tested per (i, q, j) it had cut the pass into basic blocks behind scalar branches).
    // The statistics left this pass in r06: per (i, q) they cost eight 16-lane DPP reductions + a cross-half shuffle of values every LANE held for different
    // pixels; the store pass below gives every THREAD the same 8 channels for all of its 32 pixels, so the sums run in registers and meet once.
    auto acc_pass = [&](auto RELUc) __attribute__((always_inline)) {
        constexpr bool RELU = decltype(RELUc)::value != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cl = wn * 128 + i * 32 + q * 8 + lh * 4;           // channel inside the tile
                const f32x4_t bias = *reinterpret_cast<__attribute__((address_space(3))) const f32x4_t *>(bias_lds + cl);
                const int unit = (cl >> 2) ^ (((lr & 15) << 1) & 62);
                lds_char_t *dst = tile + (wm * 128 + lr) * 512 + unit * 8;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][q * 4 + e] + bias[e];
                    if constexpr (RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    }
                    u32x2_t p;
                    p[0] = pack2_bf16(v[0], v[1]);
                    p[1] = pack2_bf16(v[2], v[3]);
                    *reinterpret_cast<__attribute__((address_space(3))) u32x2_t *>(dst + j * 32 * 512) = p;
                }
            }
    };
    const bool want_bn = a.bn_y != nullptr;                 // (never together with an addend: the host asks for the reductions only for a sole contribution)
    const bool want_add = a.add != nullptr && !want_bn;
    u32x4_t pre[16];
    if ((want_add || want_bn) && tn * 256 + (tid & 31) * 8 < a.Co) {
        const int pps = want_add ? a.add_pstride : a.bn_y_pstride;
        const bf16_t *addp = (want_add ? a.add : a.bn_y) + (size_t)(m0 + (tid >> 5)) * pps + tn * 256 + (tid & 31) * 8;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            pre[k] = *reinterpret_cast<const u32x4_t *>(addp + (size_t)(8 * k) * pps);
    }
    if (a.act == DL_ACT_RELU) acc_pass(W4IC<1>{}); else acc_pass(W4IC<0>{});
    __syncthreads();
    bf16_t *out = reinterpret_cast<bf16_t *>(a.out);
    // store pass: thread t owns 16-byte chunk t & 31 (channels (t & 31) * 8 ..) of pixel rows (t >> 5) + 8 k.  Two compile-time options:
    //   ADD   (dl_conv_forward_add): out = bf16(conv + addend) -- the residual block's "gradient so far" is added here instead of by a separate pass over
    //         both tensors (Act.add_grad's axpby: 2 reads + 1 write of 67 MB for every ResnetBlock of the data-gradient chain);
    //   STATS the fused norm statistics of exactly the values that are stored (bf16-rounded).
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    const int cc = tid & 31;
    const int co_t = tn * 256 + cc * 8;
    // (ADD: the thread's 32 addend chunks are PREFETCHED -- rows 0..15 before the accumulator pass, row k + 16 as soon as row k's register is free: fetched
    // inside the loop, four at a time behind their stores, the pass paid one HBM round trip per four rows and cost as much as the axpby it replaced)
    auto store_pass = [&](auto ADDc, auto STATSc, u32x4_t (&pre)[16]) __attribute__((always_inline)) {
        constexpr bool ADD = decltype(ADDc)::value != 0, STATS = decltype(STATSc)::value == 1, BNR = decltype(STATSc)::value == 2;
        static_assert(!(ADD && BNR), "one prefetch array");
        if (co_t >= a.Co) return;
        const bf16_t *addp = ADD ? a.add + (size_t)(m0 + (tid >> 5)) * a.add_pstride + co_t : (BNR ? a.bn_y + (size_t)(m0 + (tid >> 5)) * a.bn_y_pstride + co_t : nullptr);
        const int pre_ps = ADD ? a.add_pstride : a.bn_y_pstride;                                             // w4 shapes: output pixel index = m
        // BNR (dl_conv_forward_bnstats): this conv's output is dz of the layer z = act(norm(y)) in front; S1 = sum dn, S2 = sum dn * xhat of the values stored
        float mr[8], rs[8], sc[8], sh[8];
        const float bn_slope = a.bn_act == DL_ACT_RELU ? 0.f : (a.bn_act == DL_ACT_LRELU ? 0.2f : 1.f);
        if constexpr (BNR) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ci = n_img * a.Co + co_t + i;
                rs[i] = a.bn_rstd[ci]; mr[i] = -a.bn_mean[ci] * rs[i]; sc[i] = a.bn_scale[ci]; sh[i] = a.bn_shift[ci];       // xhat = y * rstd - mean * rstd
            }
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const int row = (tid >> 5) + 8 * k;
            const int unit = (cc * 2) ^ (((row & 15) << 1) & 62);
            u32x4_t v = *reinterpret_cast<__attribute__((address_space(3))) const u32x4_t *>(tile + row * 512 + unit * 8);
            u32x4_t r = u32x4_t{0u, 0u, 0u, 0u};
            if constexpr (ADD || BNR) {
                r = pre[k & 15];
                if (k + 16 < 32) pre[k & 15] = *reinterpret_cast<const u32x4_t *>(addp + (size_t)(8 * (k + 16)) * pre_ps);
            }
            if constexpr (ADD) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = pack2_bf16(h16_lo_f32(v[e]) + h16_lo_f32(r[e]), h16_hi_f32(v[e]) + h16_hi_f32(r[e]));
            }
            {       // (w4 shapes: whole image rows of 128 pixels, an even number of them -- every tile row is a pixel of the tensor)
                *reinterpret_cast<u32x4_t *>(out + (size_t)(m0 + row) * a.out_pstride + co_t) = v;
                if constexpr (STATS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = h16_lo_f32(v[e]), hi = h16_hi_f32(v[e]);
                        s1[2 * e] += lo; s2[2 * e] += lo * lo; s1[2 * e + 1] += hi; s2[2 * e + 1] += hi * hi;
                    }
                }
                if constexpr (BNR) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const unsigned dw = v[i >> 1], yw = r[i >> 1];
                        float dn = (i & 1) ? h16_hi_f32(dw) : h16_lo_f32(dw);
                        const float yy = (i & 1) ? h16_hi_f32(yw) : h16_lo_f32(yw);
                        const float nv = yy * sc[i] + sh[i];
                        dn = nv > 0.f ? dn : dn * bn_slope;          // act'(nv): 1 above zero; below: 0 (ReLU), 0.2 (LeakyReLU), 1 (no activation) -- no branch per element
                        s1[i] += dn;
                        s2[i] += dn * (yy * rs[i] + mr[i]);
                    }
                }
            }
        }
    };
    if (want_bn) store_pass(W4IC<0>{}, W4IC<2>{}, pre);
    else if (want_stats) { if (want_add) store_pass(W4IC<1>{}, W4IC<1>{}, pre); else store_pass(W4IC<0>{}, W4IC<1>{}, pre); }
    else { if (want_add) store_pass(W4IC<1>{}, W4IC<0>{}, pre); else store_pass(W4IC<0>{}, W4IC<0>{}, pre); }
    if (want_stats || want_bn) {
        // the 8 threads of a channel group are lanes c, c + 32 of the four waves: one shuffle, then the waves meet in LDS (fixed order: deterministic)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t1 = s1[e], t2 = s2[e];
            t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
            if (lane < 32) { red[(wave * 2 + 0) * 256 + lane * 8 + e] = t1; red[(wave * 2 + 1) * 256 + lane * 8 + e] = t2; }
        }
        __syncthreads();
        const int chunk = (m0 - n_img * HWq) >> 8;                 // tile inside its image (n_phase == 1)
        const int co = tn * 256 + tid;
        if (co < a.Co) {
            float *o = a.stats_part + ((size_t)(n_img * a.stats_nchunks + chunk) * 2) * a.Co + co;
            o[0] = (red[0 * 256 + tid] + red[2 * 256 + tid]) + (red[4 * 256 + tid] + red[6 * 256 + tid]);
            o[a.Co] = (red[1 * 256 + tid] + red[3 * 256 + tid]) + (red[5 * 256 + tid] + red[7 * 256 + tid]);
        }
    }
}

// taps of kernel row 0 ordered dw = +1, 0, -1 (the data gradient's order)?
static bool w4_flipped(const ConvArgs &a) { return (int8_t)((a.taps[0] >> 8) & 0xff) == 1; }

// The layers this kernel serves: one phase of 9 taps ordered (kh, kw) with dh constant per kernel row and the SAME dw order (-1, 0, +1 or reversed:
// the data gradient) in every row, stride 1, image rows exactly 128 pixels wide (a 256-pixel tile = two whole image rows), zero padding,
// Cin a multiple of 64, Co a multiple of 256, epilogue activation none / ReLU, bf16 result (no split-K / raw accumulators / fused norm-backward reductions).
bool w4_eligible(const ConvArgs &a) {
    if (a.n_phase != 1 || a.splitk != 1 || a.raw_out || a.in_step != 1 || a.out_step != 1 || a.Wq != 128 || a.Wi != 128 || (a.Hq & 1)) return false;
    if (a.Ho != a.Hq || a.Wo != a.Wq || a.Hi != a.Hq) return false;
    if (a.phase_tap_begin[1] - a.phase_tap_begin[0] != 9 || a.phase_tap_begin[0] != 0) return false;
    if (a.Ci < 64 || (a.Ci & 63) || (a.Co & 255) || a.pad_mode != DL_PAD_ZERO || a.in_act != DL_ACT_NONE) return false;
    if (a.act != DL_ACT_NONE && a.act != DL_ACT_RELU) return false;
    if ((size_t)a.Hi * a.Wi * (size_t)a.in_pstride * 2 >= ((size_t)1 << 31) || (size_t)256 * a.w_kstride * 2 >= ((size_t)1 << 31)) return false;   // 32-bit lane offsets
    int seen = 0;
    for (int kh = 0; kh < 3; ++kh) {
        const int dh0 = (int8_t)(a.taps[kh * 3] & 0xff);
        if (dh0 < -1 || dh0 > 1) return false;
        for (int kw = 0; kw < 3; ++kw) {
            const int16_t tp = a.taps[kh * 3 + kw];
            const int dh = (int8_t)(tp & 0xff), dw = (int8_t)((tp >> 8) & 0xff);
            if (dh != dh0 || dw != (int8_t)((a.taps[kw] >> 8) & 0xff)) return false;
        }
        seen |= 1 << (dh0 + 1);
    }
    const int d0 = (int8_t)((a.taps[0] >> 8) & 0xff), d1 = (int8_t)((a.taps[1] >> 8) & 0xff), d2 = (int8_t)((a.taps[2] >> 8) & 0xff);
    if (!((d0 == -1 && d1 == 0 && d2 == 1) || (d0 == 1 && d1 == 0 && d2 == -1))) return false;
    return seen == 7;
}

template <bool FLIP, int VAR, int ABL>
static int launch_w4(const ConvArgs &a, hipStream_t stream) {
    auto kern = conv_gemm_w4_kernel<FLIP, VAR, ABL>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)W4_LDS);
        if (e != hipSuccess) DL_FAIL("dl_conv_forward(w4): hipFuncSetAttribute(%zu): %s", W4_LDS, hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(a.tiles_m * a.tiles_n, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), W4_LDS, stream, a);
    DL_CHECK_LAUNCH("dl_conv_forward(w4)");
    return 0;
}

template <int VAR, int ABL>
static int launch_w4_dir(const ConvArgs &a, hipStream_t stream) {
    return w4_flipped(a) ? launch_w4<true, VAR, ABL>(a, stream) : launch_w4<false, VAR, ABL>(a, stream);
}

#ifndef DL_W4_DEFAULT_VAR
#define DL_W4_DEFAULT_VAR 13
#endif

int launch_conv_w4(const ConvArgs &a0, hipStream_t stream) {
    ConvArgs a = a0;
    a.tiles_m = (a.Mtot + 255) / 256;
    a.tiles_n = a.Co / 256;
#ifndef DL_DEV_SWITCHES
    return launch_w4_dir<DL_W4_DEFAULT_VAR, 0>(a, stream);
#else       // schedule / DMA variants and the timing-only ablations (results WRONG by construction): dev build only
    static const char *abl = DL_DEV_ENV("DL_W4_ABLATE");          // timing-only ablations, on the default variant
    static const char *var = DL_DEV_ENV("DL_W4_VAR");             // schedule / DMA variant (A/B): bits as documented at the kernel
    if (abl && abl[0] >= '1' && abl[0] <= '5') {
        switch (abl[0]) {
            case '1': return launch_w4_dir<DL_W4_DEFAULT_VAR, 1>(a, stream);
            case '2': return launch_w4_dir<DL_W4_DEFAULT_VAR, 2>(a, stream);
            case '3': return launch_w4_dir<DL_W4_DEFAULT_VAR, 3>(a, stream);
            case '4': return launch_w4_dir<DL_W4_DEFAULT_VAR, 4>(a, stream);
            default: return launch_w4_dir<DL_W4_DEFAULT_VAR, 5>(a, stream);
        }
    }
    const int v = var ? atoi(var) : DL_W4_DEFAULT_VAR;
    // default 13 = separated shadows + weights and activations by buffer_load lds (same-box A/B, profiles/r04/w4_dma_variants.txt: 138.3 / 137.4 us per
    // launch in the training step with global_load_lds (1), 136.9 / 137.1 with the weights on buffer loads (5), 135.2 / 135.7 with both (13))
    switch (v) {
        case 0: return launch_w4_dir<0, 0>(a, stream);
        case 3: return launch_w4_dir<3, 0>(a, stream);
        case 1: return launch_w4_dir<1, 0>(a, stream);
        case 5: return launch_w4_dir<5, 0>(a, stream);
        default: return launch_w4_dir<13, 0>(a, stream);
    }
#endif
}


// ------------------------------------------------------------------------------------------------------------------
// dl_probe_mfma_sustained: the denominator of the roofline line measured instead of quoted.  One workgroup per CU slot, four
// waves, each with the register set of conv_gemm_w4_kernel: 16 accumulators of v_mfma_f32_32x32x16_bf16 and 8 + 8 operand fragments.  The
// fragments are loaded ONCE from `data` (random bf16 from the caller: the matrix cores' power draw, and with it the clock the chip sustains,
// depends on the operand bits -- zero-filled operands run ~25 % faster than random ones on this part) and then `iters` x 64 MFMAs run
// register-resident: no LDS, no DMA, no barrier.  flops = blocks * 4 waves * iters * 64 * 32768; the caller times it with events.
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) probe_mfma_sustained_kernel(const bf16_t *data, int iters, float *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8_t F[16];
#pragma unroll
    for (int f = 0; f < 16; ++f)
        F[f] = *reinterpret_cast<const bf16x8_t *>(data + ((size_t)((blockIdx.x * 4 + wave) * 16 + f) * 64 + lane) * 8);
    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = dl_mfma32(F[i + 4 * (s & 1)], F[8 + j + 4 * (s >> 1)], acc[i][j]);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 12345.678f) sink[0] = t;          // keeps the accumulators alive; practically never true
}

extern "C" size_t dl_probe_mfma_sustained_elems(int blocks) { return (size_t)blocks * 4 * 16 * 64 * 8; }

extern "C" int dl_probe_mfma_sustained(const void *data, int blocks, int iters, float *sink, void *stream_) {
    if (!data || !sink || blocks <= 0 || iters <= 0) DL_FAIL("dl_probe_mfma_sustained: bad arguments");
    hipLaunchKernelGGL(probe_mfma_sustained_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, reinterpret_cast<const bf16_t *>(data), iters, sink);
    DL_CHECK_LAUNCH("dl_probe_mfma_sustained");
    return 0;
}
