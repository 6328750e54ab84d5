// conv_small.hip -- the two 7x7 layers of ResnetGenerator (networks.py:386-397 stem, :438-443 head) at full resolution.
//
// Both are HBM-bound at 512x512 (one side of the layer has 3 channels: 34 MB against 268 MB for the 64-channel side), and the
// general gather GEMM serves them badly: a K step is one tap, so every input row is staged into LDS once per kernel row (7x), and
// the direct-to-LDS path tops out around 18 B/clk/CU of STAGED bytes long before HBM does.  The kernels here
// stage every input element ONCE per workgroup and take the kernel window from LDS / registers instead:
//
//   conv_narrow_roll_kernel   64 -> Cout <= 4, KH x KW (the head forward).  GEMM rows = (kh, co, kw) -- the whole kernel folded into the
//       output-row dimension: an input row r, staged once ([64 px][64 ch], 8 KB, by global_load_lds), multiplies the packed weights of
//       ALL kh at once and lands in KH rolling accumulator blocks, block kh collecting output row h = r + pad - kh.  After row r the
//       block of h = r - pad is complete: its [16 px][Cout*KW] slice goes to LDS and a fifth wave forms
//       y[h][w][co] = act(bias + sum_kw T[w + kw][co, kw]) while the four MFMA waves run the next row.  Weights live in registers,
//       the row loop is unrolled KH times so that the block rotation is register renaming.
//   conv_c4_patch_kernel      Cin <= 4 -> Co (multiple of 32), K x K (the stem forward; the head's data gradient).  The input patch of a
//       8 x 64 pixel tile ((8+K-1) x (64+K) pixels x 4 channels, 8 KB) is staged once, twice: shifted by 0 and by 1 pixel, so that the
//       MFMA B fragment "2 adjacent kernel columns x 4 channels" is ONE aligned ds_read_b128 for every pixel.  K = K rows x 8 columns
//       (one zero column) x 4 channels: one 32-deep MFMA step per kernel row, no zero padding of 3 -> 8 channels in the contraction.
//       Weights (32 channels x 7 steps per wave) live in registers; the epilogue is the general kernels' (bias, activation, bf16
//       store, fused norm statistics).
#include "common.h"
#include <stdlib.h>

__device__ __attribute__((aligned(64))) unsigned char g_zero_page_small[64];

#define DLS_BAR() asm volatile("s_barrier" ::: "memory")
template <int V> struct ICS { static constexpr int value = V; };

// ----------------------------------------------------------------------------------------------------------------------------
// head forward: rolling rows
// ----------------------------------------------------------------------------------------------------------------------------
struct NarrowArgs {
    const bf16_t *x;
    const bf16_t *w;            // packed [rows_pad][kstride]: row = co*KW + kw, column = kh*Ci + ci   (dl_pack_weights, stack_kw)
    const float *bias;
    bf16_t *out;
    int N, H, W, x_pstride, w_kstride, Cout, KW, pad, act, out_pstride, out_cp;
    int strips, bands, band_rows;
    int abl;                    // timing-only ablation bits (DL_NARROW_ABL): 1 no MFMAs, 2 no output forming, 4 no DMA in the loop, 8 no T writes
};

template <int KH, int KWT, int COUT, int ACT>
__global__ void __launch_bounds__(640) conv_narrow_roll_kernel(const NarrowArgs a) {
    constexpr int CI = 64, TP = 64, NB = 32;             // channels, staged pixels per row, GEMM rows per kernel row (Cout*KW <= 32)
    constexpr int ROW = TP * CI;                         // elements of one staged row (8 KB)
    constexpr int DEPTH = 6, SLOTS = 8;                  // rows staged ahead of the pair being multiplied (two more are in use): 8 x 8 KB slots
    constexpr int TPP = TP + 4;                          // row pitch of the T slice ([32 (co,kw)][64 px] fp32)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t *xs = reinterpret_cast<bf16_t *>(smem_raw);                   // SLOTS x [64 px][64 ch], chunks XOR-swizzled per pixel
    float *ts = reinterpret_cast<float *>(smem_raw + SLOTS * ROW * 2);   // 2 x 2 x [32 (co,kw)][TPP px] fp32 (double-buffered pairs of slices): the storer reads along px

    // ten waves: 0-7 multiply (wave & 3 = which 16 of the 64 pixels, wave >> 2 = which 16 of the 32 (co, kw) rows), waves 8-9 form the outputs.
    // Splitting the rows over two wave groups keeps weights (56) + accumulators (28) far below the 168 registers three waves per SIMD allow.
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool storer = wave >= 8;           // waves 8, 9: one per row of a pair
    const int pw = wave & 3, nf = (wave >> 2) & 1;
    int b = blockIdx.x;
    const int strip = b % a.strips; b /= a.strips;
    const int band = b % a.bands;
    const int n = b / a.bands;
    const int outs = TP - (KWT - 1);                     // output pixels per strip
    const int w0 = strip * outs - a.pad;                 // image column of staged pixel 0
    const int h0 = band * a.band_rows, h1 = min(a.H, h0 + a.band_rows);
    const int r0 = h0 - a.pad;                           // first input row
    const int nrows = (h1 - h0) + KH - 1;
    const bf16_t *zero = reinterpret_cast<const bf16_t *>(g_zero_page_small);

    // ---- DMA geometry: wave w (0-7) fills pixels w*8 + lane/8 of a row, LDS chunk position lane%8
    const int dpx = (wave & 7) * 8 + (lane >> 3);
    const int dwx = w0 + dpx;
    const bool px_ok = dwx >= 0 && dwx < a.W;
    const bf16_t *src_px = a.x + ((size_t)n * a.H * a.W + (px_ok ? dwx : 0)) * (size_t)a.x_pstride + (((lane & 7) ^ ((dpx >> 1) & 7)) * 8);
    auto stage = [&](int q) __attribute__((always_inline)) {           // input row r0 + q -> slot q % SLOTS
        const int r = r0 + q;
        const bool rok = r >= 0 && r < a.H && q < nrows;
        const bf16_t *sp = (rok && px_ok) ? src_px + (size_t)r * a.W * a.x_pstride : zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)sp,
                                         (__attribute__((address_space(3))) void *)(xs + (q % SLOTS) * ROW + (wave & 7) * (8 * CI)), 16, 0, 0);
    };

    // ---- weights -> registers: A fragment (kh, kf): rows nf*16 + lane%16 of the (co, kw) rows, K chunk kf*4 + lane/16 of kernel row kh
    const int fr = lane & 15, fg = lane >> 4;
    bf16x8_t wf[KH][2];
    if (!storer) {
        const int row = nf * 16 + fr;
        const bool rv = row < COUT * KWT;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int kf = 0; kf < 2; ++kf) {
                bf16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (rv) v = *reinterpret_cast<const bf16x8_t *>(a.w + (size_t)row * a.w_kstride + kh * CI + kf * 32 + fg * 8);
                wf[kh][kf] = v;
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (storer && a.bias)
        for (int c = 0; c < COUT; ++c) bias[c] = a.bias[c];

    f32x4_t acc[KH];
#pragma unroll
    for (int p = 0; p < KH; ++p) acc[p] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (!storer) {
#pragma unroll
        for (int q = 0; q < DEPTH; ++q) stage(q);
    }
    // B fragment offsets: pixel = pw*16 + fr, chunk kf*4 + fg, swizzled by the pixel
    int xoff[2];
    {
        const int px = pw * 16 + fr;
#pragma unroll
        for (int kf = 0; kf < 2; ++kf) xoff[kf] = px * CI + (((kf * 4 + fg) ^ ((px >> 1) & 7)) * 8);
    }
    const int toff = (nf * 16 + fg * 4) * TPP + pw * 16 + fr;

    // TWO input rows per barrier step (rows q, q + 1): the per-step latency chain (wait, barrier, ds_read, MFMA result, T write) is paid
    // once per pair and the 28 MFMAs of a pair give the matrix pipe enough independent work.  S = q mod KH is static (the step loop is
    // unrolled KH times, 2*KH rows) so that every accumulator index is a compile-time constant.
    auto pair_step = [&](int q, auto SS) __attribute__((always_inline)) {
        constexpr int S0 = decltype(SS)::value, S1 = (S0 + 1) % KH;
        if (!storer && !(a.abl & 4)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH - 2) : "memory");    // rows q, q+1 have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                    // this wave's T writes / reads of the last step are done
        if (!(a.abl & 16)) DLS_BAR();
        if (!storer) {
            if (!(a.abl & 4)) { stage(q + DEPTH); stage(q + DEPTH + 1); }      // slots of rows q-2, q-1: read before the barrier by every wave
            const bf16_t *xa = xs + (q % SLOTS) * ROW, *xb = xs + ((q + 1) % SLOTS) * ROW;
            const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t *>(xa + xoff[0]);
            const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t *>(xa + xoff[1]);
            const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t *>(xb + xoff[0]);
            const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t *>(xb + xoff[1]);
            constexpr int pc0 = (S0 + (KH / 2) - (KH - 1) + 2 * KH) % KH;        // block completed by row q   (output row r - pad)
            constexpr int pc1 = (S1 + (KH / 2) - (KH - 1) + 2 * KH) % KH;        // block completed by row q+1
            float *t = ts + ((q >> 1) & 1) * (2 * NB * TPP) + toff;
            if (!(a.abl & 1)) {
#pragma unroll
                for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                    for (int kh = 0; kh < KH; ++kh) {
                        const int p = (S0 + (KH / 2) - kh + 2 * KH) % KH;          // block of output row h = r + pad - kh  (pad == KH/2)
                        acc[p] = dl_mfma16(wf[kh][kf], kf == 0 ? a0 : a1, acc[p]);
                    }
            } else {
                asm volatile("" :: "v"(a0), "v"(a1), "v"(b0), "v"(b1));
            }
            const f32x4_t done0 = acc[pc0];
            acc[pc0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (!(a.abl & 1)) {
#pragma unroll
                for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                    for (int kh = 0; kh < KH; ++kh) {
                        const int p = (S1 + (KH / 2) - kh + 2 * KH) % KH;
                        acc[p] = dl_mfma16(wf[kh][kf], kf == 0 ? b0 : b1, acc[p]);
                    }
            }
            if (!(a.abl & 8)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r * TPP] = done0[r];
#pragma unroll
                for (int r = 0; r < 4; ++r) t[NB * TPP + r * TPP] = acc[pc1][r];
            }
            acc[pc1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        } else if (q >= 2 && !(a.abl & 2)) {
            // the T slices of the previous step (rows q-2, q-1 -> output rows r0 + q - 2 - pad, + 1) are visible after the barrier
            const int j = lane, wo = strip * outs + j;
            {
                const int e = wave - 8;
                const int h = r0 + q - 2 + e - a.pad;
                if (h >= h0 && h < h1 && j < outs && wo < a.W) {
                    const float *t = ts + (((q - 2) >> 1) & 1) * (2 * NB * TPP) + e * (NB * TPP) + j;
                    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    float tv[COUT][KWT];
#pragma unroll
                    for (int c = 0; c < COUT; ++c)
#pragma unroll
                        for (int kw = 0; kw < KWT; ++kw) tv[c][kw] = t[(c * KWT + kw) * TPP + kw];      // every LDS read in flight before the first add
#pragma unroll
                    for (int c = 0; c < COUT; ++c) {
                        float sum = bias[c];
#pragma unroll
                        for (int kw = 0; kw < KWT; ++kw) sum += tv[c][kw];
                        // tanh through exp and the hardware reciprocal: the result is rounded to bf16 (2^-9), both errors are orders below
                        v[c] = ACT == DL_ACT_TANH ? 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * sum)) : sum;
                    }
                    bf16_t *o = a.out + (((size_t)n * a.H + h) * a.W + wo) * (size_t)a.out_pstride;
                    Vec8<bf16_t>::store(o, v);
                    for (int c8 = 8; c8 < a.out_cp; c8 += 8) {
                        const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        Vec8<bf16_t>::store(o + c8, z);
                    }
                }
            }
        }
    };

    for (int q = 0; q < nrows + 2; q += 2 * KH) {        // the steps past the last row only drain the T slices
#pragma unroll
        for (int s = 0; s < KH; ++s) {
            if (q + 2 * s >= nrows + 2) break;
            switch (s) {       // static S for the accumulator rotation: row q + 2s has S = 2s mod KH
                case 0: pair_step(q + 0, ICS<0>{}); break;
                case 1: pair_step(q + 2, ICS<2 % KH>{}); break;
                case 2: pair_step(q + 4, ICS<4 % KH>{}); break;
                case 3: pair_step(q + 6, ICS<6 % KH>{}); break;
                case 4: pair_step(q + 8, ICS<8 % KH>{}); break;
                case 5: pair_step(q + 10, ICS<10 % KH>{}); break;
                default: pair_step(q + 12, ICS<12 % KH>{}); break;
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------------------
// head forward, strict policy (fp32 storage, split-bf16 x3 products): the same rolling-row scheme on fp32 rows.
//   * a staged row is [64 px][64 ch] fp32 = 16 KB (global_load_lds, 16-byte chunks XOR-swizzled by the pixel); the B fragments are split into
//     bf16 hi / lo when they are read (two 16-byte LDS reads per fragment, x3_split8's arithmetic) -- the head's input is read by nothing else
//     that wants a split copy, so no producer writes one;
//   * the hi weights live in registers as in the bf16 kernel (56 VGPRs), the LO weights in LDS ([nf][kh][kf][lane] x 16 bytes, 28 KB, written once):
//     both sets in registers need ~190 VGPRs, over the 168 that ten waves per workgroup allow;
//   * one lo-weight read serves BOTH rows of a pair step; for that the two rows' MFMAs interleave, which needs the block a pair step completes
//     first to stay distinct from the block its second row opens: EIGHT rolling accumulator blocks instead of KH = 7 (block = output row mod 8),
//     and the static unroll is 4 pair steps;
//   * products per (kh, kf, row): lo_w * hi_x, hi_w * lo_x, hi_w * hi_x (small terms first), fp32 accumulation -- the order of conv_x3.h.
// HBM floor: the fp32 input is 2x the bf16 one (537 MB at 8 x 512^2: ~110 us); the round-3 route (raw gather GEMM + dl_shift_sum) took 585 us.
// 8 fp32 values (two 16-byte LDS reads) -> bf16 hi and lo MFMA fragments: hi = bf16(v), lo = bf16(v - hi)   (conv_x3.h: x3_split8)
__device__ __forceinline__ void narrow_split8(f32x4_t a, f32x4_t b, bf16x8_t &hi, bf16x8_t &lo) {
    u32x4_t h, l;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t ha = pack2_bf16(a[2 * i], a[2 * i + 1]);
        const uint32_t hb = pack2_bf16(b[2 * i], b[2 * i + 1]);
        h[i] = ha;
        h[2 + i] = hb;
        l[i] = pack2_bf16(a[2 * i] - h16_lo_f32(ha), a[2 * i + 1] - h16_hi_f32(ha));
        l[2 + i] = pack2_bf16(b[2 * i] - h16_lo_f32(hb), b[2 * i + 1] - h16_hi_f32(hb));
    }
    hi = __builtin_bit_cast(bf16x8_t, h);
    lo = __builtin_bit_cast(bf16x8_t, l);
}

struct NarrowX3Args {
    const float *x;
    const bf16_t *w_hi, *w_lo;  // packed [rows_pad][kstride]: row = co*KW + kw, column = kh*Ci + ci   (dl_pack_weights, stack_kw)
    const float *bias;
    float *out;
    int N, H, W, x_pstride, w_kstride, Cout, KW, pad, act, out_pstride, out_cp;
    int strips, bands, band_rows;
};

template <int KH, int KWT, int COUT, int ACT>
__global__ void __launch_bounds__(640) conv_narrow_roll_x3_kernel(const NarrowX3Args a) {
    constexpr int CI = 64, TP = 64, NB = 32;
    constexpr int ROWF = TP * CI;                        // floats of one staged row (16 KB)
    constexpr int DEPTH = 4, SLOTS = 6;                  // rows staged ahead of the pair being multiplied (two more are in use): 6 x 16 KB slots
    constexpr int NBLK = 8;                              // rolling accumulator blocks (block = output row mod 8)
    constexpr int TPP = TP + 4;
    static_assert(KH == 7, "block rotation below is written for pad = 3");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *xs = reinterpret_cast<float *>(smem_raw);                                         // SLOTS x [64 px][64 ch] fp32
    float *ts = reinterpret_cast<float *>(smem_raw + SLOTS * ROWF * 4);                      // 2 x 2 x [32 (co,kw)][TPP px] fp32
    bf16x8_t *wl = reinterpret_cast<bf16x8_t *>(smem_raw + SLOTS * ROWF * 4 + 4 * NB * TPP * 4);   // lo weights [2 nf][KH][2 kf][64 lanes]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool storer = wave >= 8;
    const int pw = wave & 3, nf = (wave >> 2) & 1;
    int b = blockIdx.x;
    const int strip = b % a.strips; b /= a.strips;
    const int band = b % a.bands;
    const int n = b / a.bands;
    const int outs = TP - (KWT - 1);
    const int w0 = strip * outs - a.pad;
    const int h0 = band * a.band_rows, h1 = min(a.H, h0 + a.band_rows);
    const int r0 = h0 - a.pad;
    const int nrows = (h1 - h0) + KH - 1;
    const float *zero = reinterpret_cast<const float *>(g_zero_page_small);

    // ---- DMA geometry: wave w (0-7) fills pixels w*8 .. w*8+7 of a row with two instructions of 4 pixels x 16 chunks (lane/16 = pixel, lane%16 = LDS chunk)
    const float *src_px[2];
    bool px_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int dpx = (wave & 7) * 8 + i * 4 + (lane >> 4);
        const int dwx = w0 + dpx;
        px_ok[i] = dwx >= 0 && dwx < a.W;
        src_px[i] = a.x + ((size_t)n * a.H * a.W + (px_ok[i] ? dwx : 0)) * (size_t)a.x_pstride + (((lane & 15) ^ (dpx & 15)) * 4);
    }
    auto stage = [&](int q) __attribute__((always_inline)) {           // input row r0 + q -> slot q % SLOTS
        const int r = r0 + q;
        const bool rok = r >= 0 && r < a.H && q < nrows;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float *sp = (rok && px_ok[i]) ? src_px[i] + (size_t)r * a.W * a.x_pstride : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)sp,
                                             (__attribute__((address_space(3))) void *)(xs + (q % SLOTS) * ROWF + ((wave & 7) * 8 + i * 4) * CI), 16, 0, 0);
        }
    };

    // ---- weights: hi -> registers, lo -> LDS (written by the pw == 0 wave of each row half, read by all four after the first barrier)
    const int fr = lane & 15, fg = lane >> 4;
    bf16x8_t wf[KH][2];
    if (!storer) {
        const int row = nf * 16 + fr;
        const bool rv = row < COUT * KWT;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int kf = 0; kf < 2; ++kf) {
                bf16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0}, l = {0, 0, 0, 0, 0, 0, 0, 0};
                if (rv) {
                    v = *reinterpret_cast<const bf16x8_t *>(a.w_hi + (size_t)row * a.w_kstride + kh * CI + kf * 32 + fg * 8);
                    if (pw == 0) l = *reinterpret_cast<const bf16x8_t *>(a.w_lo + (size_t)row * a.w_kstride + kh * CI + kf * 32 + fg * 8);
                }
                wf[kh][kf] = v;
                if (pw == 0) wl[((nf * KH + kh) * 2 + kf) * 64 + lane] = l;
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (storer && a.bias)
        for (int c = 0; c < COUT; ++c) bias[c] = a.bias[c];

    f32x4_t acc[NBLK];
#pragma unroll
    for (int p = 0; p < NBLK; ++p) acc[p] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (!storer) {
#pragma unroll
        for (int q = 0; q < DEPTH; ++q) stage(q);
    }
    // B fragment offsets (floats): pixel = pw*16 + fr, channels (kf*4 + fg)*8 .. +7 = 16-byte chunks 2*c8 and 2*c8 + 1, swizzled by the pixel
    int xoff[2];
    {
        const int px = pw * 16 + fr;
#pragma unroll
        for (int kf = 0; kf < 2; ++kf) xoff[kf] = px * CI + (((2 * (kf * 4 + fg)) ^ (px & 15)) * 4);
    }
    const int toff = (nf * 16 + fg * 4) * TPP + pw * 16 + fr;
    const bf16x8_t *wl_me = wl + (nf * KH * 2) * 64 + lane;

    auto pair_step = [&](int q, auto SS) __attribute__((always_inline)) {
        constexpr int S0 = decltype(SS)::value;                                              // q mod 8 (even)
        if (!storer) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DEPTH - 2) * 2) : "memory");  // rows q, q+1 have landed (two DMA instructions per row)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DLS_BAR();
        if (!storer) {
            stage(q + DEPTH); stage(q + DEPTH + 1);                                          // slots of rows q-2, q-1: read before the barrier by every wave
            const float *xa = xs + (q % SLOTS) * ROWF, *xb = xs + ((q + 1) % SLOTS) * ROWF;
#pragma unroll
            for (int kf = 0; kf < 2; ++kf) {
                // fragments of ONE K half at a time (both halves live at once cost 16 more VGPRs: spills at the 168 ten waves allow)
                bf16x8_t ah, al, bh, bl;
                narrow_split8(*reinterpret_cast<const f32x4_t *>(xa + xoff[kf]), *reinterpret_cast<const f32x4_t *>(xa + (xoff[kf] ^ 4)), ah, al);
                narrow_split8(*reinterpret_cast<const f32x4_t *>(xb + xoff[kf]), *reinterpret_cast<const f32x4_t *>(xb + (xoff[kf] ^ 4)), bh, bl);
                bf16x8_t lw_next = wl_me[kf * 64];
#pragma unroll
                for (int kh = 0; kh < KH; ++kh) {
                    const int pa = (S0 + (KH / 2) - kh + 2 * NBLK) % NBLK;                   // block of output row h = r + pad - kh (row q)
                    const int pb = (S0 + 1 + (KH / 2) - kh + 2 * NBLK) % NBLK;               // the same for row q + 1
                    const bf16x8_t lw = lw_next;                                             // read one kernel row ahead: six MFMAs cover the LDS latency
                    if (kh + 1 < KH) lw_next = wl_me[((kh + 1) * 2 + kf) * 64];
                    acc[pa] = dl_mfma16(lw, ah, acc[pa]);
                    acc[pb] = dl_mfma16(lw, bh, acc[pb]);
                    acc[pa] = dl_mfma16(wf[kh][kf], al, acc[pa]);
                    acc[pb] = dl_mfma16(wf[kh][kf], bl, acc[pb]);
                    acc[pa] = dl_mfma16(wf[kh][kf], ah, acc[pa]);
                    acc[pb] = dl_mfma16(wf[kh][kf], bh, acc[pb]);
                }
            }
            constexpr int pc0 = (S0 - (KH / 2) + 2 * NBLK) % NBLK;                            // block completed by row q     (output row r - pad)
            constexpr int pc1 = (S0 + 1 - (KH / 2) + 2 * NBLK) % NBLK;                        // block completed by row q + 1
            float *t = ts + ((q >> 1) & 1) * (2 * NB * TPP) + toff;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r * TPP] = acc[pc0][r];
#pragma unroll
            for (int r = 0; r < 4; ++r) t[NB * TPP + r * TPP] = acc[pc1][r];
            acc[pc0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            acc[pc1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        } else if (q >= 2) {
            // the T slices of the previous step (rows q-2, q-1 -> output rows r0 + q - 2 - pad, + 1) are visible after the barrier
            const int j = lane, wo = strip * outs + j;
            const int e = wave - 8;
            const int h = r0 + q - 2 + e - a.pad;
            if (h >= h0 && h < h1 && j < outs && wo < a.W) {
                const float *t = ts + (((q - 2) >> 1) & 1) * (2 * NB * TPP) + e * (NB * TPP) + j;
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                float tv[COUT][KWT];
#pragma unroll
                for (int c = 0; c < COUT; ++c)
#pragma unroll
                    for (int kw = 0; kw < KWT; ++kw) tv[c][kw] = t[(c * KWT + kw) * TPP + kw];
#pragma unroll
                for (int c = 0; c < COUT; ++c) {
                    float sum = bias[c];
#pragma unroll
                    for (int kw = 0; kw < KWT; ++kw) sum += tv[c][kw];
                    v[c] = ACT == DL_ACT_TANH ? tanhf(sum) : sum;
                }
                float *o = a.out + (((size_t)n * a.H + h) * a.W + wo) * (size_t)a.out_pstride;
                Vec8<float>::store(o, v);
                for (int c8 = 8; c8 < a.out_cp; c8 += 8) {
                    const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    Vec8<float>::store(o + c8, z);
                }
            }
        }
    };

    for (int q = 0; q < nrows + 2; q += NBLK) {          // the steps past the last row only drain the T slices
#pragma unroll
        for (int s = 0; s < NBLK / 2; ++s) {
            if (q + 2 * s >= nrows + 2) break;
            switch (s) {       // static S0 = (q + 2s) mod 8 for the accumulator rotation
                case 0: pair_step(q + 0, ICS<0>{}); break;
                case 1: pair_step(q + 2, ICS<2>{}); break;
                case 2: pair_step(q + 4, ICS<4>{}); break;
                default: pair_step(q + 6, ICS<6>{}); break;
            }
        }
    }
}

extern "C" int dl_conv_narrow_supported(int dtype, int Ci, int x_pstride, int Cout, int KH, int KW, int pad, int pad_mode) {
    static const bool no_x3 = DL_DEV_ENV("DL_NO_NARROW_X3") != nullptr;      // A/B switch: the strict head back on dl_conv_forward(raw_out) + dl_shift_sum
    return (dtype == DL_BF16 || (dtype == DL_F32 && !no_x3)) && Ci == 64 && x_pstride % 8 == 0 && KH == 7 && KW == 7 && (Cout == 1 || Cout == 3) &&
           pad == KH / 2 && pad == KW / 2 && pad_mode == DL_PAD_ZERO;
}

static void narrow_bands(int N, int H, int W, int KW, int *strips, int *bands, int *band_rows) {
    const int outs = 64 - (KW - 1);
    *strips = (W + outs - 1) / outs;
    // bands: as many as keep the grid within ~2 workgroups per CU, each at least 16 rows (KH-1 halo rows are recomputed per band)
    int b = 512 / (N * *strips);
    if (b < 1) b = 1;
    if (b > (H + 15) / 16) b = (H + 15) / 16;
    *band_rows = (H + b - 1) / b;
    *bands = (H + *band_rows - 1) / *band_rows;
}

// strict policy: x and out are fp32 NHWC, w_hi / w_lo the two packed images of dl_pack_weights(stack_kw) -- see conv_narrow_roll_x3_kernel
extern "C" int dl_conv_narrow_forward_x3(const void *x, int N, int H, int W, int Ci, int x_pstride, const void *w_hi, const void *w_lo, int w_kstride,
                                         int Cout, int KH, int KW, int pad, const float *bias, int act, void *out, int out_pstride, int out_Cp,
                                         void *stream) {
    if (N <= 0 || H <= 0 || W <= 0) DL_FAIL("dl_conv_narrow_forward_x3: empty problem (N=%d, %dx%d)", N, H, W);
    if (!x || !w_hi || !w_lo || !out) DL_FAIL("dl_conv_narrow_forward_x3: null argument");
    if (!dl_conv_narrow_supported(DL_F32, Ci, x_pstride, Cout, KH, KW, pad, DL_PAD_ZERO))
        DL_FAIL("dl_conv_narrow_forward_x3: unsupported layer (Ci=%d Cout=%d %dx%d pad %d): use dl_conv_forward(raw_out) + dl_shift_sum", Ci, Cout, KH, KW, pad);
    if (out_Cp % 8 || out_Cp < 8 || out_pstride < out_Cp) DL_FAIL("dl_conv_narrow_forward_x3: bad output channel geometry");
    NarrowX3Args a;
    a.x = (const float *)x; a.w_hi = (const bf16_t *)w_hi; a.w_lo = (const bf16_t *)w_lo; a.bias = bias; a.out = (float *)out;
    a.N = N; a.H = H; a.W = W; a.x_pstride = x_pstride; a.w_kstride = w_kstride; a.Cout = Cout; a.KW = KW; a.pad = pad; a.act = act;
    a.out_pstride = out_pstride; a.out_cp = out_Cp;
    narrow_bands(N, H, W, KW, &a.strips, &a.bands, &a.band_rows);
    constexpr size_t smem = 6 * 64 * 64 * 4 + 4 * 32 * 68 * 4 + 2 * 7 * 2 * 64 * 16;        // rows | T slices | lo weights = 161 792 B
    void (*kern)(const NarrowX3Args) = nullptr;
    if (Cout == 3 && act == DL_ACT_TANH) kern = conv_narrow_roll_x3_kernel<7, 7, 3, DL_ACT_TANH>;
    else if (Cout == 3 && act == DL_ACT_NONE) kern = conv_narrow_roll_x3_kernel<7, 7, 3, DL_ACT_NONE>;
    else if (Cout == 1 && act == DL_ACT_TANH) kern = conv_narrow_roll_x3_kernel<7, 7, 1, DL_ACT_TANH>;
    else if (Cout == 1 && act == DL_ACT_NONE) kern = conv_narrow_roll_x3_kernel<7, 7, 1, DL_ACT_NONE>;
    else DL_FAIL("dl_conv_narrow_forward_x3: Cout=%d / act=%d has no instantiation (Cout 1 or 3, act none or tanh)", Cout, act);
    static void (*attr_done[4])(const NarrowX3Args) = {nullptr, nullptr, nullptr, nullptr};
    bool seen = false;
    for (int i = 0; i < 4; ++i) seen |= attr_done[i] == kern;
    if (!seen) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) DL_FAIL("dl_conv_narrow_forward_x3: hipFuncSetAttribute: %s", hipGetErrorString(e));
        for (int i = 0; i < 4; ++i)
            if (!attr_done[i]) { attr_done[i] = kern; break; }
    }
    hipLaunchKernelGGL(kern, dim3(N * a.strips * a.bands), dim3(640), smem, (hipStream_t)stream, a);
    DL_CHECK_LAUNCH("dl_conv_narrow_forward_x3");
    return 0;
}

extern "C" int dl_conv_narrow_forward(const void *x, int N, int H, int W, int Ci, int x_pstride, const void *w_hi, int w_kstride, int Cout, int KH, int KW,
                                      int pad, const float *bias, int act, void *out, int out_pstride, int out_Cp, void *stream) {
    if (N <= 0 || H <= 0 || W <= 0) DL_FAIL("dl_conv_narrow_forward: empty problem (N=%d, %dx%d)", N, H, W);
    if (!x || !w_hi || !out) DL_FAIL("dl_conv_narrow_forward: null argument");
    if (!dl_conv_narrow_supported(DL_BF16, Ci, x_pstride, Cout, KH, KW, pad, DL_PAD_ZERO))
        DL_FAIL("dl_conv_narrow_forward: unsupported layer (Ci=%d Cout=%d %dx%d pad %d): use dl_conv_forward(raw_out) + dl_shift_sum", Ci, Cout, KH, KW, pad);
    if (out_Cp % 8 || out_Cp < 8 || out_pstride < out_Cp) DL_FAIL("dl_conv_narrow_forward: bad output channel geometry");
    NarrowArgs a;
    a.x = (const bf16_t *)x; a.w = (const bf16_t *)w_hi; a.bias = bias; a.out = (bf16_t *)out;
    a.N = N; a.H = H; a.W = W; a.x_pstride = x_pstride; a.w_kstride = w_kstride; a.Cout = Cout; a.KW = KW; a.pad = pad; a.act = act;
    a.out_pstride = out_pstride; a.out_cp = out_Cp;
    const int outs = 64 - (KW - 1);
    a.strips = (W + outs - 1) / outs;
    // bands: as many as keep the grid within ~2 workgroups per CU, each at least 16 rows (KH-1 halo rows are recomputed per band)
    int bands = 512 / (N * a.strips);
    if (bands < 1) bands = 1;
    if (bands > (H + 15) / 16) bands = (H + 15) / 16;
    a.band_rows = (H + bands - 1) / bands;
    a.bands = (H + a.band_rows - 1) / a.band_rows;
    static const char *abl_env = DL_DEV_ENV("DL_NARROW_ABL");
    a.abl = abl_env ? atoi(abl_env) : 0;
    constexpr size_t smem = 8 * 64 * 64 * 2 + 4 * 32 * 68 * 4;
    void (*kern)(const NarrowArgs) = nullptr;
    if (Cout == 3 && act == DL_ACT_TANH) kern = conv_narrow_roll_kernel<7, 7, 3, DL_ACT_TANH>;
    else if (Cout == 3 && act == DL_ACT_NONE) kern = conv_narrow_roll_kernel<7, 7, 3, DL_ACT_NONE>;
    else if (Cout == 1 && act == DL_ACT_TANH) kern = conv_narrow_roll_kernel<7, 7, 1, DL_ACT_TANH>;
    else if (Cout == 1 && act == DL_ACT_NONE) kern = conv_narrow_roll_kernel<7, 7, 1, DL_ACT_NONE>;
    else DL_FAIL("dl_conv_narrow_forward: Cout=%d / act=%d has no instantiation (Cout 1 or 3, act none or tanh)", Cout, act);
    static void (*attr_done[4])(const NarrowArgs) = {nullptr, nullptr, nullptr, nullptr};
    bool seen = false;
    for (int i = 0; i < 4; ++i) seen |= attr_done[i] == kern;
    if (!seen) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) DL_FAIL("dl_conv_narrow_forward: hipFuncSetAttribute: %s", hipGetErrorString(e));
        for (int i = 0; i < 4; ++i)
            if (!attr_done[i]) { attr_done[i] = kern; break; }
    }
    hipLaunchKernelGGL(kern, dim3(N * a.strips * a.bands), dim3(640), smem, (hipStream_t)stream, a);
    DL_CHECK_LAUNCH("dl_conv_narrow_forward");
    return 0;
}
