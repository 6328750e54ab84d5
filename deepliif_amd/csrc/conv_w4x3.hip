// conv_w4x3.hip -- conv_gemm_w4_kernel (conv_w4.hip) for the STRICT policy: fp32 storage, split-bf16 x3 products (hi_w hi_x + hi_w lo_x + lo_w hi_x),
// the policy the GPU tests hold to 1e-3 against the reference.  Same layers (ResnetBlock 3x3, stride 1, 128-pixel image rows, networks.py:467-513),
// same tile (256 px x 256 ch, four waves of 128 x 128 on v_mfma_f32_32x32x16_bf16, one wave per SIMD), same LDS plan and pipeline:
//   * K step = 32 channels: an fp32 pixel row of 32 channels is 128 B = the bf16 kernel's 64 channels, so slab, weight buffers, DMA geometry, kernel-
//     column reuse, W(t+2) prefetch with s_waitcnt vmcnt(8) and the [W0 | X0 | X1 | W1 | W2] layout carry over unchanged (160 KB);
//   * the activations are the producer-written SPLIT COPY (dl_conv_desc.in_split: every group of 8 channels = [8 hi bf16 | 8 lo bf16], written by the
//     norm kernels) -- chunk 2g of a row is the hi fragment of channel group g, chunk 2g + 1 the lo fragment: no conversion in the kernel;
//     a weight row is [32 hi | 32 lo] gathered from the two packed images by per-lane source pointers;
//   * a K=16 sub-step reads 16 fragments (4 x {W hi, W lo, X hi, X lo}) for 48 MFMAs, term-major (lo_w hi_x, hi_w lo_x, hi_w hi_x: small terms first,
//     dependent MFMAs 16 issues apart); two sub-steps per K step, fragments double-buffered in registers (128 VGPRs + 256 accumulators);
//   * epilogue: fp32 results leave through LDS in two 128-channel halves (128 KB each, 16-byte chunks XOR-swizzled by the pixel) as whole 512-byte
//     pixel rows; bias / ReLU; fused per-(image, channel) statistics of the stored values.
// MEASURED (r04, same box, profiles/r04/w4x3_ab.txt): a TIE with the 8-phase strict kernel (conv_x3.h) -- 345.6-347.1 vs 343.7-344.2 us per launch inside the strict
// step, 204.8 vs 203.8 ms per step -- so this kernel is OPT-IN (DL_CONV_W4X3=1, see w4x3_enabled() in conv_x3.h for the reading).  Timing-only ablations: MFMAs
// only 267 us (K loop 235 us for 186 us of matrix-pipe time at 2.4 GHz), no DMA 295 us, prologue + epilogue 32 us, full 390 us.  Inputs that are NOT split
// copies always stay on the 8-phase kernel.
#include "conv_args.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((address_space(3))) char lds_char_t;
typedef __attribute__((address_space(3))) const bf16x8_t lds_frag_t;

template <int V> struct X4IC { static constexpr int value = V; };

__device__ __forceinline__ float x4_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
    return v;
}

constexpr int X4_BUF = 256 * 128;                     // one weight buffer (256 output channels x [32 hi | 32 lo]) = one slab (2 x 128 pixels x 32 channels split)
constexpr int X4_X0 = 1 * X4_BUF;                     // slab s at X4_X0 + s * X4_BUF
__host__ __device__ constexpr int x4_wofs(int b) { return b == 0 ? 0 : (b == 1 ? 3 * X4_BUF : 4 * X4_BUF); }
constexpr size_t X4_LDS = (size_t)5 * X4_BUF;         // the epilogue needs 128 KB + 4 KB of it
static_assert(X4_LDS <= 160 * 1024, "the whole LDS of a CU");

// FLIP: kw taps ordered dw = +1, 0, -1 (data gradient).  ABL != 0: timing-only ablations: 1 = no DMA in the loop, 3 = MFMAs only, 4 = no K loop
template <bool FLIP, int ABL>
__global__ void __launch_bounds__(256) conv_gemm_w4x3_kernel(const ConvArgs a) {
    constexpr bool DMA_ON = ABL != 1 && ABL != 3;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_char_t *lds = (lds_char_t *)smem_raw;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % a.tiles_n, tm = bid / a.tiles_n;
    const int nch = a.Ci >> 5;                   // 32-channel chunks
    const int G = ABL == 4 ? 0 : 3 * nch;        // (chunk, kh) groups of three K steps

    int dhs[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) dhs[k] = (int)(int8_t)(a.taps[3 * k] & 0xff);

    // ---- staging geometry (as conv_gemm_w4_kernel): LDS rows are 128 B = 8 chunks of 16 B, chunk c of row r at position c ^ ((r >> 1) & 7)
    const int lrow = lane >> 3, lcp = lane & 7;
    const int R = wave >> 1, cbase = (wave & 1) * 64;
    const int HWq = a.Hq * a.Wq;
    const int m0 = tm * 256;
    const int n_img = m0 / HWq;
    const int h0 = (m0 - n_img * HWq) >> 7;
    uint32_t x_off[8];
    const char *w_ptr[8];                        // weight piece i of this lane at K column 0: hi image for logical chunks 0-3, lo image for 4-7
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int col = cbase + 8 * i + lrow, xrow = R * 128 + col;
        x_off[i] = (uint32_t)(((R * a.Wi + col) * a.in_pstride) * 4 + (lcp ^ ((xrow >> 1) & 7)) * 16);
        const int s = (wave * 8 + i) * 8 + lrow;
        const int cl = lcp ^ ((s >> 1) & 7);
        w_ptr[i] = reinterpret_cast<const char *>((cl & 4) ? a.w_lo : a.w_hi) + ((size_t)(tn * 256 + s) * a.w_kstride + a.phase_kbase[0] + (cl & 3) * 8) * 2;
    }
    const char *xg = reinterpret_cast<const char *>(a.in) + ((size_t)(n_img * a.Hi + h0) * a.Wi) * (size_t)a.in_pstride * 4;
    const int x_dst0 = X4_X0 + (R * 128 + cbase) * 128;              // + slab * BUF + i * 1024
    const int w_dst0 = wave * 8 * 1024;                              // + x4_wofs(buf) + i * 1024
    const ptrdiff_t tap_bytes = (ptrdiff_t)a.Ci * 2;                 // one kernel tap further along a packed weight row (bf16 columns)
    const ptrdiff_t row_bytes = (ptrdiff_t)a.Wi * a.in_pstride * 4;
    char *const xres = const_cast<char *>(xg - row_bytes);           // buffer resource base: one image row above the tile (scalar offsets stay >= 0)

    // scalar byte offset (from xg - row_bytes) of group (32-channel chunk c, kernel row kh), and whether that image row exists for this wave
    auto x_group_off = [&](int c, int kh, bool &valid) __attribute__((always_inline)) {
        const int dh = kh == 0 ? dhs[0] : (kh == 1 ? dhs[1] : dhs[2]);
        valid = (unsigned)(h0 + R + dh) < (unsigned)a.Hi;
        return (int)((ptrdiff_t)(dh + 1) * row_bytes + c * 128);
    };
    auto dma_x = [&](auto I, int soff, bool valid, int slab) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        // an image row outside the tensor is fetched through a resource with num_records = 0: out-of-range buffer loads deliver zeros
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(xres, 0, valid ? 0x7fffffff : 0, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(lds + x_dst0 + slab * X4_BUF + i * 1024), 16,
                                                 (int)x_off[i], soff, 0, 0);
    };
    auto dma_w = [&](auto I, ptrdiff_t koff, auto BUF) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value, buf = decltype(BUF)::value;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(w_ptr[i] + koff),
                                         (__attribute__((address_space(3))) void *)(lds + w_dst0 + x4_wofs(buf) + i * 1024), 16, 0, 0);
    };

    // ---- fragment addressing (bytes): lane = (row lr of a 32-row block, channel group lh of a 16-wide sub-step).
    // weights: logical chunk = plane * 4 + 2 * s + lh (plane 0 = hi, 1 = lo);  activations: logical chunk = 2 * (2 * s + lh) + plane
    const int lr = lane & 31, lh = lane >> 5;
    const int swz_w = (lr >> 1) & 7;
    const int aw_row = (wn * 128 + lr) * 128;
    int ax_row[3], swz_x[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int row = wm * 128 + lr + (FLIP ? 1 - k : k - 1);
        ax_row[k] = X4_X0 + row * 128;
        swz_x[k] = (row >> 1) & 7;
    }
    const bool edge_lo = lr == 0, edge_hi = lr == 31;

    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment set of one sub-step: [0..3] W hi, [4..7] W lo, [8..11] X hi, [12..15] X lo (index = 32-row block)
    bf16x8_t FA[16], FB[16];
    if constexpr (ABL == 3) {
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            FA[f] = bf16x8_t{(short)(0x3f80 + lane), (short)(0x3f00 + f), 0x3e80, 0x3f81, (short)0xbf80, 0x3f10, 0x3e90, 0x3f91};
            FB[f] = bf16x8_t{(short)(0x3f00 + lane), (short)(0x3f80 + f), 0x3e90, 0x3f01, (short)0xbf00, 0x3f20, 0x3e80, 0x3f11};
        }
    }
    // read K: order X lo 0-3, W hi 0-3, then (W lo i, X hi i) pairs -- the LAST two reads (W lo 3, X hi 3) feed MFMA 0 of the sub-step that consumes them,
    // so the compiler's wait for the whole set sits in front of that sub-step (see conv_w4.hip)
    auto read_k = [&](auto Kc, int wbase, int xbase, int wsw, int xsw, bf16x8_t (&F)[16]) __attribute__((always_inline)) {
        constexpr int k = decltype(Kc)::value;
        if constexpr (ABL == 3) return;
        if constexpr (k < 4) F[12 + k] = *reinterpret_cast<lds_frag_t *>(lds + xbase + ((xsw ^ 1) << 4) + k * 4096);                   // X lo k: chunk 2g + 1
        else if constexpr (k < 8) F[k - 4] = *reinterpret_cast<lds_frag_t *>(lds + wbase + (wsw << 4) + (k - 4) * 4096);              // W hi
        else if constexpr ((k & 1) == 0) F[4 + ((k - 8) >> 1)] = *reinterpret_cast<lds_frag_t *>(lds + wbase + ((wsw ^ 4) << 4) + ((k - 8) >> 1) * 4096);      // W lo
        else F[8 + ((k - 9) >> 1)] = *reinterpret_cast<lds_frag_t *>(lds + xbase + (xsw << 4) + ((k - 9) >> 1) * 4096);              // X hi: chunk 2g
    };
    // positions of this lane's chunks in a row for sub-step s (g = 2s + lh, the swizzle is an XOR): weights hi = g ^ swz, lo = (4 + g) ^ swz = hi ^ 4;
    // activations hi = (2g) ^ swz, lo = (2g + 1) ^ swz = hi ^ 1
    auto w_pos = [&](int s) __attribute__((always_inline)) { return (2 * s + lh) ^ swz_w; };
    auto x_pos_hi = [&](int s, int k) __attribute__((always_inline)) { return (2 * (2 * s + lh)) ^ swz_x[k]; };

    // the 48 MFMAs of a sub-step, term-major; MFMA m of a term works on (i, j) = (3 - (m >> 2), 3 - (m & 3))
    auto mma3 = [&](auto Mc, const bf16x8_t (&F)[16]) __attribute__((always_inline)) {
        constexpr int m = decltype(Mc)::value, term = m >> 4, q = m & 15, i = 3 - (q >> 2), j = 3 - (q & 3);
        if constexpr (term == 0) acc[i][j] = dl_mfma32(F[4 + i], F[8 + j], acc[i][j]);          // lo_w hi_x
        else if constexpr (term == 1) acc[i][j] = dl_mfma32(F[i], F[12 + j], acc[i][j]);        // hi_w lo_x
        else acc[i][j] = dl_mfma32(F[i], F[8 + j], acc[i][j]);                                  // hi_w hi_x
    };
    auto fix_edge = [&](auto SHc, bf16x8_t (&F)[16]) __attribute__((always_inline)) {
        constexpr int SH = decltype(SHc)::value;
        if constexpr (ABL == 3) return;
        if constexpr (SH == 0) { if (edge_lo) { F[8] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0}; F[12] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0}; } }
        if constexpr (SH == 2) { if (edge_hi) { F[11] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0}; F[15] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0}; } }
    };
    // one sub-step: 48 MFMAs on Fc; when READS, the 16 fragment reads of the next sub-step (two after each of the MFMAs 0, 2, .., 14) into Fn;
    // hook(m) after MFMA m (DMA slots: the odd MFMAs 1 .. 23)
    auto substep = [&](auto READSc, const bf16x8_t (&Fc)[16], bf16x8_t (&Fn)[16], int wbase, int xbase, int wsw, int xsw, auto &&hook) __attribute__((always_inline)) {
        constexpr bool READS = decltype(READSc)::value != 0;
        auto one = [&](auto Mc) __attribute__((always_inline)) {
            constexpr int m = decltype(Mc)::value;
            mma3(Mc, Fc);
            if constexpr (READS && m < 16 && (m & 1) == 0) {
                read_k(X4IC<(m < 16) ? m : 0>{}, wbase, xbase, wsw, xsw, Fn);
                read_k(X4IC<(m < 16) ? m + 1 : 0>{}, wbase, xbase, wsw, xsw, Fn);
            }
            hook(Mc);
        };
        one(X4IC<0>{}); one(X4IC<1>{}); one(X4IC<2>{}); one(X4IC<3>{}); one(X4IC<4>{}); one(X4IC<5>{}); one(X4IC<6>{}); one(X4IC<7>{});
        one(X4IC<8>{}); one(X4IC<9>{}); one(X4IC<10>{}); one(X4IC<11>{}); one(X4IC<12>{}); one(X4IC<13>{}); one(X4IC<14>{}); one(X4IC<15>{});
        one(X4IC<16>{}); one(X4IC<17>{}); one(X4IC<18>{}); one(X4IC<19>{}); one(X4IC<20>{}); one(X4IC<21>{}); one(X4IC<22>{}); one(X4IC<23>{});
        one(X4IC<24>{}); one(X4IC<25>{}); one(X4IC<26>{}); one(X4IC<27>{}); one(X4IC<28>{}); one(X4IC<29>{}); one(X4IC<30>{}); one(X4IC<31>{});
        one(X4IC<32>{}); one(X4IC<33>{}); one(X4IC<34>{}); one(X4IC<35>{}); one(X4IC<36>{}); one(X4IC<37>{}); one(X4IC<38>{}); one(X4IC<39>{});
        one(X4IC<40>{}); one(X4IC<41>{}); one(X4IC<42>{}); one(X4IC<43>{}); one(X4IC<44>{}); one(X4IC<45>{}); one(X4IC<46>{}); one(X4IC<47>{});
    };
    auto pin = [&](auto READSc, auto NDc) __attribute__((always_inline)) {
        constexpr bool READS = decltype(READSc)::value != 0;
        constexpr int ND = decltype(NDc)::value;
#pragma unroll
        for (int m = 0; m < 48; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (READS && m < 16 && (m & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            if ((m & 1) == 1 && (m >> 1) < ND) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
    };
    auto nohook = [](auto) __attribute__((always_inline)) {};

    // ---- running source offsets: group g and group g + 1 (clamped to the last group)
    int c = 0, kh = 0;
    ptrdiff_t wgrp = 0, wgrp_n = 0;                // weight column byte offset of (c, kh, kw = 0) of this / the next group
    int xo_n = 0;
    bool xv_n = false;
    auto advance = [&](int g, int &cn, int &khn) __attribute__((always_inline)) {
        cn = c; khn = kh + 1;
        if (khn == 3) { khn = 0; cn = c + 1; }
        if (g + 1 >= G) { cn = c; khn = kh; }
        wgrp_n = ((ptrdiff_t)(khn * 3) * a.Ci + cn * 32) * 2;
        xo_n = x_group_off(cn, khn, xv_n);
    };

    // ---- prologue: slab of group 0, weights of steps 0 and 1
    if (G > 0) {
        bool v0;
        const int xo0 = x_group_off(0, 0, v0);
        dma_x(X4IC<0>{}, xo0, v0, 0); dma_x(X4IC<1>{}, xo0, v0, 0); dma_x(X4IC<2>{}, xo0, v0, 0); dma_x(X4IC<3>{}, xo0, v0, 0);
        dma_x(X4IC<4>{}, xo0, v0, 0); dma_x(X4IC<5>{}, xo0, v0, 0); dma_x(X4IC<6>{}, xo0, v0, 0); dma_x(X4IC<7>{}, xo0, v0, 0);
        dma_w(X4IC<0>{}, 0, X4IC<0>{}); dma_w(X4IC<1>{}, 0, X4IC<0>{}); dma_w(X4IC<2>{}, 0, X4IC<0>{}); dma_w(X4IC<3>{}, 0, X4IC<0>{});
        dma_w(X4IC<4>{}, 0, X4IC<0>{}); dma_w(X4IC<5>{}, 0, X4IC<0>{}); dma_w(X4IC<6>{}, 0, X4IC<0>{}); dma_w(X4IC<7>{}, 0, X4IC<0>{});
        dma_w(X4IC<0>{}, tap_bytes, X4IC<1>{}); dma_w(X4IC<1>{}, tap_bytes, X4IC<1>{}); dma_w(X4IC<2>{}, tap_bytes, X4IC<1>{}); dma_w(X4IC<3>{}, tap_bytes, X4IC<1>{});
        dma_w(X4IC<4>{}, tap_bytes, X4IC<1>{}); dma_w(X4IC<5>{}, tap_bytes, X4IC<1>{}); dma_w(X4IC<6>{}, tap_bytes, X4IC<1>{}); dma_w(X4IC<7>{}, tap_bytes, X4IC<1>{});
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // slab 0, weights 0 visible (weights 1 still in flight)

    if (G > 0) {
        const int wb = aw_row + x4_wofs(0), xb = ax_row[0];
        const int wsw = w_pos(0), xsw = x_pos_hi(0, 0);
        read_k(X4IC<0>{}, wb, xb, wsw, xsw, FA); read_k(X4IC<1>{}, wb, xb, wsw, xsw, FA); read_k(X4IC<2>{}, wb, xb, wsw, xsw, FA); read_k(X4IC<3>{}, wb, xb, wsw, xsw, FA);
        read_k(X4IC<4>{}, wb, xb, wsw, xsw, FA); read_k(X4IC<5>{}, wb, xb, wsw, xsw, FA); read_k(X4IC<6>{}, wb, xb, wsw, xsw, FA); read_k(X4IC<7>{}, wb, xb, wsw, xsw, FA);
        read_k(X4IC<8>{}, wb, xb, wsw, xsw, FA); read_k(X4IC<9>{}, wb, xb, wsw, xsw, FA); read_k(X4IC<10>{}, wb, xb, wsw, xsw, FA); read_k(X4IC<11>{}, wb, xb, wsw, xsw, FA);
        read_k(X4IC<12>{}, wb, xb, wsw, xsw, FA); read_k(X4IC<13>{}, wb, xb, wsw, xsw, FA); read_k(X4IC<14>{}, wb, xb, wsw, xsw, FA); read_k(X4IC<15>{}, wb, xb, wsw, xsw, FA);
    }

    // K step t = 3 g + KW of group g (32 channels of one tap): sub-step 0 carries the DMA of step t+2 and the reads of sub-step 1; barrier; sub-step 1
    // covers the first reads of step t+1
    auto step = [&](auto KWc, int g) __attribute__((always_inline)) {
        constexpr int KW = decltype(KWc)::value;
        constexpr int SH = FLIP ? 2 - KW : KW;
        const int xs = (g & 1) * X4_BUF, xsn = ((g + 1) & 1) * X4_BUF;
        const int wcur = aw_row + x4_wofs(KW), xcur = ax_row[KW] + xs;
        const int wnext = aw_row + x4_wofs((KW + 1) % 3);
        const int xnext = KW < 2 ? ax_row[(KW + 1) % 3] + xs : ax_row[0] + xsn;
        const int xsw_next = x_pos_hi(0, KW < 2 ? (KW + 1) % 3 : 0);
        const ptrdiff_t wsrc = KW == 0 ? wgrp + 2 * tap_bytes : (KW == 1 ? wgrp_n : wgrp_n + tap_bytes);
        const int xslab = (g + 1) & 1;
        auto piece = [&](auto Pc) __attribute__((always_inline)) {
            constexpr int p = decltype(Pc)::value;
            if constexpr (!DMA_ON) return;
            constexpr int NX = KW < 2 ? 4 : 0;
            if constexpr (p < NX) dma_x(X4IC<(KW == 1 ? 4 : 0) + (p < NX ? p : 0)>{}, xo_n, xv_n, xslab);
            else if constexpr (p - NX < 8) dma_w(X4IC<(p - NX >= 0 && p - NX < 8) ? p - NX : 0>{}, wsrc, X4IC<(KW + 2) % 3>{});
        };
        constexpr int NP = KW < 2 ? 12 : 8;
        auto hook = [&](auto Mc) __attribute__((always_inline)) {
            constexpr int m = decltype(Mc)::value;
            if constexpr ((m & 1) == 1 && (m >> 1) < NP) piece(X4IC<((m & 1) == 1 && (m >> 1) < NP) ? (m >> 1) : 0>{});
        };
        __builtin_amdgcn_sched_barrier(0);
        fix_edge(X4IC<SH>{}, FA);
        substep(X4IC<1>{}, FA, FB, wcur, xcur, w_pos(1), x_pos_hi(1, KW), hook);
        pin(X4IC<1>{}, X4IC<DMA_ON ? NP : 0>{});
        __builtin_amdgcn_sched_barrier(0);
        fix_edge(X4IC<SH>{}, FB);
        if constexpr (DMA_ON) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        substep(X4IC<1>{}, FB, FA, wnext, xnext, w_pos(0), xsw_next, nohook);
        pin(X4IC<1>{}, X4IC<0>{});
    };

    for (int g = 0; g < G; ++g) {
        int cn, khn;
        advance(g, cn, khn);
        step(X4IC<0>{}, g);
        step(X4IC<1>{}, g);
        step(X4IC<2>{}, g);
        c = cn; kh = khn; wgrp = wgrp_n;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();             // LDS is dead from here on

    // ---- epilogue.  acc[i][j][r] = out[pixel = wm*128 + j*32 + lr][channel = wn*128 + i*32 + 8*(r>>2) + 4*lh + (r&3)], fp32.
    // Two passes (channel half h = the waves with wn == h): [256 pixels][128 channels] fp32 through LDS (512-byte rows, 16-byte chunk c of row r at
    // c ^ (r & 31): the 32 lanes of a fragment column write 32 different chunks), then whole rows out, 16 B per lane.
    lds_char_t *tile = lds;
    __attribute__((address_space(3))) float *red = reinterpret_cast<__attribute__((address_space(3))) float *>(lds + 256 * 512);       // [wm][2][256]
    const bool want_stats = a.stats_part != nullptr;
    float *out = reinterpret_cast<float *>(a.out);
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        if (wn == h) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = i * 32 + q * 8 + lh * 4;                   // channel inside the half
                    const int co = tn * 256 + h * 128 + cl;
                    float bias[4] = {0.f, 0.f, 0.f, 0.f};
                    if (a.bias) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) bias[e] = (co + e < a.bias_n) ? a.bias[co + e] : 0.f;
                    }
                    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4_t v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][q * 4 + e] + bias[e];
                        if (a.act == DL_ACT_RELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                        }
                        const int row = wm * 128 + j * 32 + lr;
                        *reinterpret_cast<__attribute__((address_space(3))) f32x4_t *>(tile + row * 512 + (((cl >> 2) ^ (row & 31)) << 4)) = v;
                        if (want_stats) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
                        }
                    }
                    if (want_stats) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            s1[e] = x4_row16_sum(s1[e]); s2[e] = x4_row16_sum(s2[e]);
                            s1[e] += __shfl_xor(s1[e], 16, 64); s2[e] += __shfl_xor(s2[e], 16, 64);
                        }
                        if (lr == 0) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                red[(wm * 2 + 0) * 256 + h * 128 + cl + e] = s1[e];
                                red[(wm * 2 + 1) * 256 + h * 128 + cl + e] = s2[e];
                            }
                        }
                    }
                }
        }
        __syncthreads();
#pragma unroll 4
        for (int idx = tid; idx < 256 * 32; idx += 256) {
            const int row = idx >> 5, cc = idx & 31;
            const int m = m0 + row;
            const int co = tn * 256 + h * 128 + cc * 4;
            if (m >= a.Mtot || co >= a.Co) continue;
            const f32x4_t v = *reinterpret_cast<__attribute__((address_space(3))) const f32x4_t *>(tile + row * 512 + ((cc ^ (row & 31)) << 4));
            *reinterpret_cast<f32x4_t *>(out + (size_t)m * a.out_pstride + co) = v;
        }
        __syncthreads();
    }
    if (want_stats) {
        const int chunk = (m0 - n_img * HWq) >> 8;
        const int co = tn * 256 + tid;
        if (co < a.Co) {
            float *o = a.stats_part + ((size_t)(n_img * a.stats_nchunks + chunk) * 2) * a.Co + co;
            o[0] = red[0 * 256 + tid] + red[2 * 256 + tid];
            o[a.Co] = red[1 * 256 + tid] + red[3 * 256 + tid];
        }
    }
}

static bool w4x3_flipped(const ConvArgs &a) { return (int8_t)((a.taps[0] >> 8) & 0xff) == 1; }

// The layers this kernel serves: conv_gemm_w4_kernel's (w4_eligible, conv_w4.hip) under the strict policy, with a SPLIT-COPY input, Cin a multiple of 32
bool w4x3_eligible(const ConvArgs &a) {
    if (!a.in_split || a.w_lo == nullptr) return false;
    if (a.n_phase != 1 || a.splitk != 1 || a.raw_out || a.in_step != 1 || a.out_step != 1 || a.Wq != 128 || a.Wi != 128 || (a.Hq & 1)) return false;
    if (a.Ho != a.Hq || a.Wo != a.Wq || a.Hi != a.Hq) return false;
    if (a.phase_tap_begin[1] - a.phase_tap_begin[0] != 9 || a.phase_tap_begin[0] != 0) return false;
    if (a.Ci < 32 || (a.Ci & 31) || (a.Co & 255) || a.pad_mode != DL_PAD_ZERO || a.bn_y != nullptr || a.in_act != DL_ACT_NONE) return false;
    if (a.act != DL_ACT_NONE && a.act != DL_ACT_RELU) return false;
    if ((a.in_pstride & 3) || (a.out_pstride & 3)) return false;
    if ((size_t)(a.Hi + 2) * a.Wi * (size_t)a.in_pstride * 4 >= ((size_t)1 << 31)) return false;          // 32-bit lane / scalar offsets
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

template <bool FLIP, int ABL>
static int launch_w4x3(const ConvArgs &a, hipStream_t stream) {
    auto kern = conv_gemm_w4x3_kernel<FLIP, ABL>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)X4_LDS);
        if (e != hipSuccess) DL_FAIL("dl_conv_forward(w4x3): hipFuncSetAttribute(%zu): %s", X4_LDS, hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(a.tiles_m * a.tiles_n, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), X4_LDS, stream, a);
    DL_CHECK_LAUNCH("dl_conv_forward(w4x3)");
    return 0;
}

int launch_conv_w4x3(const ConvArgs &a0, hipStream_t stream) {
    ConvArgs a = a0;
    a.tiles_m = (a.Mtot + 255) / 256;
    a.tiles_n = a.Co / 256;
    const bool flip = w4x3_flipped(a);
#ifdef DL_DEV_SWITCHES      // timing-only ablations (results WRONG by construction): dev build only
    static const char *abl = DL_DEV_ENV("DL_W4X3_ABLATE");
    if (abl && abl[0] == '1') return flip ? launch_w4x3<true, 1>(a, stream) : launch_w4x3<false, 1>(a, stream);
    if (abl && abl[0] == '3') return flip ? launch_w4x3<true, 3>(a, stream) : launch_w4x3<false, 3>(a, stream);
    if (abl && abl[0] == '4') return flip ? launch_w4x3<true, 4>(a, stream) : launch_w4x3<false, 4>(a, stream);
#endif
    return flip ? launch_w4x3<true, 0>(a, stream) : launch_w4x3<false, 0>(a, stream);
}
