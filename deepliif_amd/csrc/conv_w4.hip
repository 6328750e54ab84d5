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
//   * kernel-column reuse: the three kw taps of one (64-channel chunk, kh) read ONE staged slab of 2 x (1 + 128 + 1) pixel rows
//     at row offsets dw + 1 (pad columns zeroed once), so the activations cost 32 KB of DMA per THREE K steps: 43 KB per step
//     instead of 64 KB;
//   * K order (chunk, kh, kw): the 9 taps of a 64-channel chunk run back to back (the slab of a chunk stays in the XCD's L2).
// Pipeline per K step t (W double buffer, X slab double buffer):  sub-steps 0..2 issue the ds_reads of the next sub-step
// before their MFMAs and carry the DMA of W(t+1) [+ a third of slab(g+1)]; then s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier
// (the DMA was issued >= 1000 cycles earlier); sub-step 3's MFMAs run AFTER the barrier and cover the first fragment reads
// of step t+1.  The steps past the end re-stage the last step into the idle buffers (no branches in the loop).
// Epilogue: as tile_epilogue_lds of conv_gemm.hip for the 32x32 accumulator layout -- bias / activation, bf16 tile transposed
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

constexpr int W4_XB = 2 * 130 * 128;      // one activation slab: 2 image rows x (pad + 128 + pad) pixels x 64 channels (bytes)
constexpr int W4_WB = 256 * 128;          // one weight buffer: 256 output channels x 64 K (bytes)
constexpr int W4_XS = 0, W4_WS = 2 * W4_XB;
constexpr size_t W4_LOOP_LDS = (size_t)2 * W4_XB + 2 * W4_WB;
constexpr size_t W4_EPI_LDS = (size_t)256 * 512 + 256 * sizeof(int) + (size_t)2 * 2 * 256 * sizeof(float);
constexpr size_t W4_LDS = W4_LOOP_LDS > W4_EPI_LDS ? W4_LOOP_LDS : W4_EPI_LDS;

// ABL != 0: timing-only ablations (results wrong by construction): 1 = no DMA in the loop, 2 = DMA only, 3 = MFMAs only
template <int ABL, int SCHED>
__global__ void __launch_bounds__(256) conv_gemm_w4_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_char_t *lds = (lds_char_t *)smem_raw;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % a.tiles_n, tm = bid / a.tiles_n;
    const int nch = a.Ci >> 6;
    const int G = 3 * nch;                     // (chunk, kh) groups of three K steps

    // taps are ordered (kh, kw); dh is constant per kh, the dw order is the same in every kernel row (w4_eligible)
    int dhs[3], shs[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        dhs[k] = (int)(int8_t)(a.taps[3 * k] & 0xff);
        shs[k] = (int)(int8_t)((a.taps[k] >> 8) & 0xff) + 1;
    }

    if (tid < 64) {          // zero the 2 x 2 pad pixels of both slabs (16 bytes per thread): image columns -1 and 128
        const int c = tid & 7, r = (tid >> 3) & 1, e = (tid >> 4) & 1, b = tid >> 5;
        *reinterpret_cast<__attribute__((address_space(3))) u32x4_t *>(lds + W4_XS + b * W4_XB + (r * 130 + e * 129) * 128 + c * 16) = u32x4_t{0, 0, 0, 0};
    }

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
        const int col = cbase + 8 * i + lrow, xrow = R * 130 + 1 + col;
        x_off[i] = (uint32_t)(((R * a.Wi + col) * a.in_pstride + (lcp ^ ((xrow >> 1) & 7)) * 8) * 2);
        const int s = (wave * 8 + i) * 8 + lrow;
        w_off[i] = (uint32_t)((s * a.w_kstride + (lcp ^ ((s >> 1) & 7)) * 8) * 2);
    }
    const char *xg = reinterpret_cast<const char *>(a.in) + ((size_t)(n_img * a.Hi + h0) * a.Wi) * (size_t)a.in_pstride * 2;
    const char *wg = reinterpret_cast<const char *>(a.w_hi) + ((size_t)(tn * 256) * a.w_kstride + a.phase_kbase[0]) * 2;
    const char *zero = reinterpret_cast<const char *>(g_w4_zero_page);
    const int x_dst0 = W4_XS + (R * 130 + 1 + cbase) * 128;          // + slab * XB + i * 1024
    const int w_dst0 = W4_WS + wave * 8 * 1024;                      // + buf * WB + i * 1024

    // uniform part of the source addresses of group (chunk c, kernel row kh) / of K step (c, kh, kw)
    auto x_group_base = [&](int c, int kh, bool &valid) __attribute__((always_inline)) {
        const int hr = h0 + R + dhs[kh];
        valid = (unsigned)hr < (unsigned)a.Hi;
        return xg + ((ptrdiff_t)dhs[kh] * a.Wi * a.in_pstride + c * 64) * 2;
    };
    auto w_step_base = [&](int c, int kh, int kw) __attribute__((always_inline)) {
        return wg + ((size_t)(kh * 3 + kw) * a.Ci + c * 64) * 2;
    };
    auto dma_x = [&](auto I, const char *base, bool valid, int slab) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        const char *src = valid ? base + x_off[i] : zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds + x_dst0 + slab * W4_XB + i * 1024), 16, 0, 0);
    };
    auto dma_w = [&](auto I, const char *base, int buf) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + w_off[i]),
                                         (__attribute__((address_space(3))) void *)(lds + w_dst0 + buf * W4_WB + i * 1024), 16, 0, 0);
    };

    // ---- fragment addressing (bytes): lane = (row lr of a 32-row block, K half lh of a 16-wide sub-step)
    const int lr = lane & 31, lh = lane >> 5;
    const int aw = W4_WS + (wn * 128 + lr) * 128 + ((lh ^ ((lr >> 1) & 7)) << 4);
    int axk[3];                                                         // activation fragment base of kw = 0, 1, 2 (slab row shift dw + 1)
    {
        int ax[3];
#pragma unroll
        for (int sh = 0; sh < 3; ++sh) {
            const int row = wm * 130 + sh + lr;
            ax[sh] = W4_XS + row * 128 + ((lh ^ ((row >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) axk[k] = shs[k] == 0 ? ax[0] : (shs[k] == 1 ? ax[1] : ax[2]);
    }

    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    bf16x8_t FA[8], FB[8];
    if constexpr (ABL >= 2) {
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            FA[f] = bf16x8_t{(short)(0x3f80 + lane), (short)(0x3f00 + f), 0x3e80, 0x3f81, (short)0xbf80, 0x3f10, 0x3e90, 0x3f91};
            FB[f] = bf16x8_t{(short)(0x3f00 + lane), (short)(0x3f80 + f), 0x3e90, 0x3f01, (short)0xbf00, 0x3f20, 0x3e80, 0x3f11};
        }
    }
    auto read_one = [&](auto FI, int wp, int xp, bf16x8_t (&F)[8]) __attribute__((always_inline)) {
        constexpr int f = decltype(FI)::value;
        if constexpr (ABL >= 2) return;
        if constexpr (f < 4) F[f] = *reinterpret_cast<lds_frag_t *>(lds + wp + f * 4096);
        else F[f] = *reinterpret_cast<lds_frag_t *>(lds + xp + (f - 4) * 4096);
    };
    auto mma_one = [&](auto MI, const bf16x8_t (&F)[8]) __attribute__((always_inline)) {
        constexpr int m = decltype(MI)::value, i = m >> 2, j = m & 3;
        if constexpr (ABL == 2) return;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[i], F[4 + j], acc[i][j], 0, 0, 0);
    };
    // one K=16 sub-step: 16 MFMAs on Fc, the 8 fragment reads of the next sub-step into Fn (bases wp / xp), and `hook(m)` after MFMA m
    // (DMA issue slots).  SCHED 1: the source order MFMA, read, MFMA, read, ... is pinned with sched_group_barriers.
    auto substep = [&](const bf16x8_t (&Fc)[8], bf16x8_t (&Fn)[8], int wp, int xp, auto &&hook) __attribute__((always_inline)) {
        mma_one(W4IC<0>{}, Fc);  read_one(W4IC<0>{}, wp, xp, Fn); hook(W4IC<0>{});
        mma_one(W4IC<1>{}, Fc);  read_one(W4IC<4>{}, wp, xp, Fn); hook(W4IC<1>{});
        mma_one(W4IC<2>{}, Fc);  read_one(W4IC<1>{}, wp, xp, Fn); hook(W4IC<2>{});
        mma_one(W4IC<3>{}, Fc);  read_one(W4IC<5>{}, wp, xp, Fn); hook(W4IC<3>{});
        mma_one(W4IC<4>{}, Fc);  read_one(W4IC<2>{}, wp, xp, Fn); hook(W4IC<4>{});
        mma_one(W4IC<5>{}, Fc);  read_one(W4IC<6>{}, wp, xp, Fn); hook(W4IC<5>{});
        mma_one(W4IC<6>{}, Fc);  read_one(W4IC<3>{}, wp, xp, Fn); hook(W4IC<6>{});
        mma_one(W4IC<7>{}, Fc);  read_one(W4IC<7>{}, wp, xp, Fn); hook(W4IC<7>{});
        mma_one(W4IC<8>{}, Fc);  hook(W4IC<8>{});
        mma_one(W4IC<9>{}, Fc);  hook(W4IC<9>{});
        mma_one(W4IC<10>{}, Fc); hook(W4IC<10>{});
        mma_one(W4IC<11>{}, Fc); hook(W4IC<11>{});
        mma_one(W4IC<12>{}, Fc); hook(W4IC<12>{});
        mma_one(W4IC<13>{}, Fc); hook(W4IC<13>{});
        mma_one(W4IC<14>{}, Fc); hook(W4IC<14>{});
        mma_one(W4IC<15>{}, Fc); hook(W4IC<15>{});
    };
    auto nohook = [](auto) __attribute__((always_inline)) {};

    // ---- prologue: slab of group 0, weights of step 0
    {
        bool v0;
        const char *xb0 = x_group_base(0, 0, v0);
        const char *wb0 = w_step_base(0, 0, 0);
        dma_w(W4IC<0>{}, wb0, 0); dma_w(W4IC<1>{}, wb0, 0); dma_w(W4IC<2>{}, wb0, 0); dma_w(W4IC<3>{}, wb0, 0);
        dma_w(W4IC<4>{}, wb0, 0); dma_w(W4IC<5>{}, wb0, 0); dma_w(W4IC<6>{}, wb0, 0); dma_w(W4IC<7>{}, wb0, 0);
        dma_x(W4IC<0>{}, xb0, v0, 0); dma_x(W4IC<1>{}, xb0, v0, 0); dma_x(W4IC<2>{}, xb0, v0, 0); dma_x(W4IC<3>{}, xb0, v0, 0);
        dma_x(W4IC<4>{}, xb0, v0, 0); dma_x(W4IC<5>{}, xb0, v0, 0); dma_x(W4IC<6>{}, xb0, v0, 0); dma_x(W4IC<7>{}, xb0, v0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();          // pad pixels, slab 0, weights 0 visible

    {
        const int wp = aw, xp = axk[0];
        read_one(W4IC<0>{}, wp, xp, FA); read_one(W4IC<1>{}, wp, xp, FA); read_one(W4IC<2>{}, wp, xp, FA); read_one(W4IC<3>{}, wp, xp, FA);
        read_one(W4IC<4>{}, wp, xp, FA); read_one(W4IC<5>{}, wp, xp, FA); read_one(W4IC<6>{}, wp, xp, FA); read_one(W4IC<7>{}, wp, xp, FA);
    }

    // K step (group g = (chunk c, kernel row kh), kw = KW): see the header comment
    auto step = [&](auto KWc, int g, int c, int kh, int gn, int cn, int khn) __attribute__((always_inline)) {
        constexpr int KW = decltype(KWc)::value;
        const int t = g * 3 + KW;
        const int wcur = aw + (t & 1) * W4_WB, xcur = axk[KW] + (g & 1) * W4_XB;
        // next K step: same group (kw + 1) or the first of the next group
        const int wnext = aw + ((t + 1) & 1) * W4_WB;
        const int xnext = KW < 2 ? axk[KW < 2 ? KW + 1 : 0] + (g & 1) * W4_XB : axk[0] + ((g + 1) & 1) * W4_XB;
        const char *wb = KW < 2 ? w_step_base(c, kh, KW + 1) : w_step_base(cn, khn, 0);
        bool xv;
        const char *xb = x_group_base(cn, khn, xv);
        (void)gn;
        const int wbuf = (t + 1) & 1, xslab = (g + 1) & 1;
        // DMA issue slots: W(t+1) pieces 0-3 + one slab piece in sub-step 0, pieces 4-7 + up to two slab pieces in sub-step 1; nothing in sub-steps
        // 2 and 3, so the youngest DMA has ~1000 matrix-pipe cycles to land before the vmcnt(0) in front of the barrier
        auto hook0 = [&](auto MI) __attribute__((always_inline)) {
            constexpr int m = decltype(MI)::value;
            if constexpr (ABL == 1 || ABL == 3) return;
            if constexpr (m == 1) dma_w(W4IC<0>{}, wb, wbuf);
            if constexpr (m == 4) dma_w(W4IC<1>{}, wb, wbuf);
            if constexpr (m == 7) dma_w(W4IC<2>{}, wb, wbuf);
            if constexpr (m == 10) dma_w(W4IC<3>{}, wb, wbuf);
            if constexpr (m == 13) {
                if constexpr (KW == 0) dma_x(W4IC<0>{}, xb, xv, xslab);
                if constexpr (KW == 1) dma_x(W4IC<3>{}, xb, xv, xslab);
                if constexpr (KW == 2) dma_x(W4IC<6>{}, xb, xv, xslab);
            }
        };
        auto hook1 = [&](auto MI) __attribute__((always_inline)) {
            constexpr int m = decltype(MI)::value;
            if constexpr (ABL == 1 || ABL == 3) return;
            if constexpr (m == 0) dma_w(W4IC<4>{}, wb, wbuf);
            if constexpr (m == 3) dma_w(W4IC<5>{}, wb, wbuf);
            if constexpr (m == 6) dma_w(W4IC<6>{}, wb, wbuf);
            if constexpr (m == 9) dma_w(W4IC<7>{}, wb, wbuf);
            if constexpr (m == 11) {
                if constexpr (KW == 0) dma_x(W4IC<1>{}, xb, xv, xslab);
                if constexpr (KW == 1) dma_x(W4IC<4>{}, xb, xv, xslab);
                if constexpr (KW == 2) dma_x(W4IC<7>{}, xb, xv, xslab);
            }
            if constexpr (m == 13) {
                if constexpr (KW == 0) dma_x(W4IC<2>{}, xb, xv, xslab);
                if constexpr (KW == 1) dma_x(W4IC<5>{}, xb, xv, xslab);
            }
        };
        substep(FA, FB, wcur ^ (1 << 5), xcur ^ (1 << 5), hook0);
        if constexpr (SCHED == 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
#pragma unroll
            for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        }
        substep(FB, FA, wcur ^ (2 << 5), xcur ^ (2 << 5), hook1);
        if constexpr (SCHED == 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
#pragma unroll
            for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        }
        substep(FA, FB, wcur ^ (3 << 5), xcur ^ (3 << 5), nohook);
        if constexpr (SCHED == 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
#pragma unroll
            for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
        }
        // W(t+1) / slab(g+1) have landed (issued >= 1000 cycles ago), this wave's reads of buffer t are complete
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        substep(FB, FA, wnext, xnext, nohook);
        if constexpr (SCHED == 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
            for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
        }
    };

    int c = 0, kh = 0;
    for (int g = 0; g < G; ++g) {
        // next group, clamped to the last one (its re-staged copy lands in the idle slab / buffer)
        int cn = c, khn = kh + 1;
        if (khn == 3) { khn = 0; cn = c + 1; }
        if (g + 1 >= G) { cn = c; khn = kh; }
        step(W4IC<0>{}, g, c, kh, g + 1, cn, khn);
        step(W4IC<1>{}, g, c, kh, g + 1, cn, khn);
        step(W4IC<2>{}, g, c, kh, g + 1, cn, khn);
        c = cn; kh = khn;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();             // LDS is dead from here on (the epilogue reuses it)

    // ---- epilogue.  acc[i][j][r] = out[pixel = wm*128 + j*32 + lr][channel = wn*128 + i*32 + 8*(r>>2) + 4*lh + (r&3)]
    lds_char_t *tile = lds;                                                         // [256 pixels][256 channels] bf16, 8-byte units XOR-swizzled by the pixel
    __attribute__((address_space(3))) int *rowtab = reinterpret_cast<__attribute__((address_space(3))) int *>(lds + 256 * 512);
    __attribute__((address_space(3))) float *red = reinterpret_cast<__attribute__((address_space(3))) float *>(lds + 256 * 512 + 256 * 4);       // [wm][2][256]
    const bool want_stats = a.stats_part != nullptr;
    {
        const int m = m0 + tid;
        int opix = -1;
        if (m < a.Mtot) {
            const int n = m / HWq, rem = m - n * HWq;
            const int hq = rem / a.Wq, wq = rem - hq * a.Wq;
            opix = (n * a.Ho + hq) * a.Wo + wq;
        }
        rowtab[tid] = opix;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cl = wn * 128 + i * 32 + q * 8 + lh * 4;           // channel inside the tile
            const int co = tn * 256 + cl;
            float bias[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bias[e] = (co + e < a.bias_n) ? a.bias[co + e] : 0.f;
            }
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
            const int unit = (cl >> 2) ^ (((lr & 15) << 1) & 62);
            lds_char_t *dst = tile + (wm * 128 + lr) * 512 + unit * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][q * 4 + e] + bias[e];
                if (a.act == DL_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                u32x2_t p;
                p[0] = pack2_bf16(v[0], v[1]);
                p[1] = pack2_bf16(v[2], v[3]);
                *reinterpret_cast<__attribute__((address_space(3))) u32x2_t *>(dst + j * 32 * 512) = p;
                if (want_stats) {          // statistics of exactly what is stored (bf16-rounded), like the stand-alone kernel sees
                    const float q0 = __uint_as_float(p[0] << 16), q1 = __uint_as_float(p[0] & 0xffff0000u);
                    const float q2 = __uint_as_float(p[1] << 16), q3 = __uint_as_float(p[1] & 0xffff0000u);
                    s1[0] += q0; s2[0] += q0 * q0; s1[1] += q1; s2[1] += q1 * q1;
                    s1[2] += q2; s2[2] += q2 * q2; s1[3] += q3; s2[3] += q3 * q3;
                }
            }
            if (want_stats) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1[e] = w4_row16_sum(s1[e]); s2[e] = w4_row16_sum(s2[e]);
                    s1[e] += __shfl_xor(s1[e], 16, 64); s2[e] += __shfl_xor(s2[e], 16, 64);
                }
                if (lr == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        red[(wm * 2 + 0) * 256 + cl + e] = s1[e];
                        red[(wm * 2 + 1) * 256 + cl + e] = s2[e];
                    }
                }
            }
        }
    __syncthreads();
    bf16_t *out = reinterpret_cast<bf16_t *>(a.out);
#pragma unroll 4
    for (int idx = tid; idx < 256 * 32; idx += 256) {
        const int row = idx >> 5, cc = idx & 31;
        const int opix = rowtab[row];
        const int co = tn * 256 + cc * 8;
        if (opix < 0 || co >= a.Co) continue;
        const int unit = (cc * 2) ^ (((row & 15) << 1) & 62);
        const u32x4_t v = *reinterpret_cast<__attribute__((address_space(3))) const u32x4_t *>(tile + row * 512 + unit * 8);
        *reinterpret_cast<u32x4_t *>(out + (size_t)opix * a.out_pstride + co) = v;
    }
    if (want_stats) {
        const int chunk = (m0 - n_img * HWq) >> 8;                 // tile inside its image (n_phase == 1)
        const int co = tn * 256 + tid;
        if (co < a.Co) {
            float *o = a.stats_part + ((size_t)(n_img * a.stats_nchunks + chunk) * 2) * a.Co + co;
            o[0] = red[0 * 256 + tid] + red[2 * 256 + tid];
            o[a.Co] = red[1 * 256 + tid] + red[3 * 256 + tid];
        }
    }
}

// The layers this kernel serves: one phase of 9 taps ordered (kh, kw) with dh constant per kernel row and the SAME dw order (-1, 0, +1 or reversed:
// the data gradient) in every row, stride 1, image rows exactly 128 pixels wide (a 256-pixel tile = two whole image rows), zero padding,
// Cin a multiple of 64, Co a multiple of 256, epilogue activation none / ReLU, bf16 result (no split-K / raw accumulators / fused norm-backward reductions).
bool w4_eligible(const ConvArgs &a) {
    if (a.n_phase != 1 || a.splitk != 1 || a.raw_out || a.in_step != 1 || a.out_step != 1 || a.Wq != 128 || a.Wi != 128 || (a.Hq & 1)) return false;
    if (a.Ho != a.Hq || a.Wo != a.Wq || a.Hi != a.Hq) return false;
    if (a.phase_tap_begin[1] - a.phase_tap_begin[0] != 9 || a.phase_tap_begin[0] != 0) return false;
    if (a.Ci < 64 || (a.Ci & 63) || (a.Co & 255) || a.pad_mode != DL_PAD_ZERO || a.bn_y != nullptr || a.in_act != DL_ACT_NONE) return false;
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

template <int ABL, int SCHED>
static int launch_w4(const ConvArgs &a, hipStream_t stream) {
    auto kern = conv_gemm_w4_kernel<ABL, SCHED>;
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

int launch_conv_w4(const ConvArgs &a0, hipStream_t stream) {
    ConvArgs a = a0;
    a.tiles_m = (a.Mtot + 255) / 256;
    a.tiles_n = a.Co / 256;
    static const char *abl = getenv("DL_W4_ABLATE");
    static const char *sched = getenv("DL_W4_SCHED");
    const bool s1 = sched && sched[0] == '1';
    if (abl && abl[0] == '1') return launch_w4<1, 0>(a, stream);
    if (abl && abl[0] == '2') return launch_w4<2, 0>(a, stream);
    if (abl && abl[0] == '3') return launch_w4<3, 0>(a, stream);
    if (s1) return launch_w4<0, 1>(a, stream);
    return launch_w4<0, 0>(a, stream);
}
