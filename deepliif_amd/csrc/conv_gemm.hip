// conv_gemm.hip -- implicit-GEMM gather convolution on MFMA (gfx950), see include/deepliif_hip.h dl_conv_forward.
//
// GEMM view (per sub-pixel phase):   D[co][pix] = sum_k W[co][k] * X[pix][k],   k = (tap, ci), ci fastest.
//   rows  (MFMA A operand) = output channels  -> packed weights, K-contiguous rows
//   cols  (MFMA B operand) = output pixels    -> gathered NHWC activations, 8 channels (16 B) per load
// Both LDS tiles are [row][BK] bf16 with the 16-byte chunks of a row XOR-swizzled so that ds_read_b128 fragment reads
// are bank-conflict free for the gfx950 lane-group servicing order (MI355X_MICROARCH.md, LDS table).
// The accumulator fragment of mfma_f32_16x16x32_bf16 gives each lane 4 consecutive output channels of one pixel
// -> one 8-byte (bf16) / 16-byte (fp32) store per fragment.
// Pipeline: register-staged double buffer (global loads of tile t+1 are issued before the MFMAs of tile t and written
// to the other LDS buffer after them), one barrier per K step.
#include "common.h"
#include <stdlib.h>

#include "conv_args.h"

// conv_w4.hip: the one-wave-per-SIMD kernel of the ResnetBlock shape
bool w4_eligible(const ConvArgs &a);
int launch_conv_w4(const ConvArgs &a0, hipStream_t stream);
// conv_w4x3.hip: the same tile for the strict policy (split-copy activations, hi / lo weight images, three MFMAs per product)
bool w4x3_eligible(const ConvArgs &a);
int launch_conv_w4x3(const ConvArgs &a0, hipStream_t stream);
// conv_s2f.hip: the fused four-phase tile of the stride-2 layers (transposed conv forward, stride-2 data gradients)
bool s2f_eligible(const ConvArgs &a);
int s2f_stats_chunks(const ConvArgs &a);
int launch_conv_s2f(const ConvArgs &a0, hipStream_t stream);
// conv_s2f_x3.hip: the same tile under the strict policy (split-copy input, three products)
bool d1g_eligible(const ConvArgs &a);
int launch_conv_d1g(const ConvArgs &a0, hipStream_t stream);
bool d1_eligible(const ConvArgs &a);
int launch_conv_d1(const ConvArgs &a0, hipStream_t stream);
bool dot_fwd_eligible(const ConvArgs &a);
bool dot_dgrad_eligible(const ConvArgs &a);
int launch_conv_dot(const ConvArgs &a, bool fwd, hipStream_t stream);
bool s2u_eligible(const ConvArgs &a);
int s2u_stats_chunks(const ConvArgs &a);
int launch_conv_s2u(const ConvArgs &a0, hipStream_t stream);
bool s2d_eligible(const ConvArgs &a);
int s2d_stats_chunks(const ConvArgs &a);
int launch_conv_s2d(const ConvArgs &a0, hipStream_t stream);
bool s2f_x3_eligible(const ConvArgs &a);
int launch_conv_s2f_x3(const ConvArgs &a0, hipStream_t stream);

__device__ __forceinline__ int reflect_idx(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

// raw staging registers for one 8-element chunk
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> { u32x4_t v; };
template <> struct Raw8<float> { f32x4_t a, b; };

template <typename T> __device__ __forceinline__ void raw_zero(Raw8<T> &r);
template <> __device__ __forceinline__ void raw_zero<bf16_t>(Raw8<bf16_t> &r) { r.v = u32x4_t{0, 0, 0, 0}; }
template <> __device__ __forceinline__ void raw_zero<float>(Raw8<float> &r) { r.a = f32x4_t{0.f, 0.f, 0.f, 0.f}; r.b = r.a; }
template <typename T> __device__ __forceinline__ void raw_load(Raw8<T> &r, const T *p);
template <> __device__ __forceinline__ void raw_load<bf16_t>(Raw8<bf16_t> &r, const bf16_t *p) { r.v = *reinterpret_cast<const u32x4_t *>(p); }
template <> __device__ __forceinline__ void raw_load<float>(Raw8<float> &r, const float *p) {
    r.a = *reinterpret_cast<const f32x4_t *>(p);
    r.b = *reinterpret_cast<const f32x4_t *>(p + 4);
}
__device__ __forceinline__ void raw_to_f32(const Raw8<bf16_t> &r, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = h16_lo_f32(r.v[i]); f[2 * i + 1] = h16_hi_f32(r.v[i]); }
}
__device__ __forceinline__ void raw_to_f32(const Raw8<float> &r, float (&f)[8]) {
    f[0] = r.a[0]; f[1] = r.a[1]; f[2] = r.a[2]; f[3] = r.a[3]; f[4] = r.b[0]; f[5] = r.b[1]; f[6] = r.b[2]; f[7] = r.b[3];
}

// convert a staged chunk to the bf16 hi (and lo) LDS images
template <typename T, int PREC>
__device__ __forceinline__ void raw_to_planes(const Raw8<T> &r, int in_act, u32x4_t &hi, u32x4_t &lo) {
    if constexpr (sizeof(T) == 2 && PREC == 1) {
        if (in_act == DL_ACT_NONE) { hi = reinterpret_cast<const Raw8<bf16_t> &>(r).v; return; }
    }
    float f[8];
    raw_to_f32(r, f);
    if (in_act != DL_ACT_NONE) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = apply_act(in_act, f[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bf16_t h0 = f32_to_bf16(f[2 * i]), h1 = f32_to_bf16(f[2 * i + 1]);
        hi[i] = (uint32_t)h0 | ((uint32_t)h1 << 16);
        if constexpr (PREC == 3) {
            lo[i] = pack2_bf16(f[2 * i] - bf16_to_f32(h0), f[2 * i + 1] - bf16_to_f32(h1));
        }
    }
}

template <typename TIn, typename TOut, int PREC, int BM, int BN, int BK, int WM, int WN>
__global__ void __launch_bounds__(256) conv_gemm_kernel(const ConvArgs a) {
    constexpr int CPR = BK / 8;                       // 16-byte chunks per LDS row
    constexpr int X_CH = (BM * CPR + 255) / 256;      // chunks per thread, activation tile
    constexpr int W_CH = (BN * CPR + 255) / 256;      // chunks per thread, weight tile
    constexpr int PM = BM / WM, PN = BN / WN, FM = PM / 16, FN = PN / 16;
    constexpr int NPL = (PREC == 3) ? 2 : 1;
    constexpr int XT = BM * BK, WT = BN * BK;         // elements per tile plane
    constexpr int BUF = NPL * (XT + WT);              // elements per pipeline buffer
    static_assert(WM * WN == 4, "4 waves");
    static_assert((256 % CPR) == 0, "chunk column is constant per thread");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t *smem = reinterpret_cast<bf16_t *>(smem_raw);
    int16_t *tap_lds = reinterpret_cast<int16_t *>(smem + 2 * BUF);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;

    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % a.tiles_n, tm = bid / a.tiles_n;
    const int phase = blockIdx.y / a.splitk, ks = blockIdx.y % a.splitk;
    const int tap0 = a.phase_tap_begin[phase];
    const int ntaps = a.phase_tap_begin[phase + 1] - tap0;
    const int kbase = a.phase_kbase[phase];
    const int nk_total = (ntaps * a.Ci + 63) / 64 * 64 / BK;
    const int nk_per = (nk_total + a.splitk - 1) / a.splitk;
    const int kt_begin = ks * nk_per;
    const int kt_end = min(nk_total, kt_begin + nk_per);

    if (tid < DL_MAX_TAPS) tap_lds[tid] = a.taps[tid];

    // ---- per-thread geometry of the activation rows it stages (fixed over the K loop)
    const int xc = tid % CPR;                         // chunk column (same for all of this thread's rows)
    int x_hi0[X_CH], x_wi0[X_CH], x_nbase[X_CH];
    bool x_ok[X_CH];
    const int HWq = a.Hq * a.Wq;
#pragma unroll
    for (int i = 0; i < X_CH; ++i) {
        const int row = (tid + i * 256) / CPR;
        const int m = tm * BM + row;
        x_ok[i] = (row < BM) && (m < a.Mtot);
        const int mm = x_ok[i] ? m : 0;
        const int n = mm / HWq, rem = mm - n * HWq;
        const int hq = rem / a.Wq, wq = rem - hq * a.Wq;
        x_hi0[i] = hq * a.in_step;
        x_wi0[i] = wq * a.in_step;
        x_nbase[i] = n * a.Hi;
    }
    const TIn *in = reinterpret_cast<const TIn *>(a.in);

    Raw8<TIn> xr[X_CH];
    u32x4_t wr[NPL][W_CH];

    auto load_tile = [&](int kt) {
        // activations
        const int k0 = kt * BK + xc * 8;
        const int tl = k0 >> a.log2Ci;
        const int ci = k0 & (a.Ci - 1);
        int dh = 0, dw = 0;
        const bool tap_ok = tl < ntaps;
        if (tap_ok) {
            const int16_t t = tap_lds[tap0 + tl];
            dh = (int)(int8_t)(t & 0xff);
            dw = (int)(int8_t)((t >> 8) & 0xff);
        }
#pragma unroll
        for (int i = 0; i < X_CH; ++i) {
            int hi = x_hi0[i] + dh, wi = x_wi0[i] + dw;
            bool ok = x_ok[i] && tap_ok;
            if (a.pad_mode == DL_PAD_REFLECT) {
                hi = reflect_idx(hi, a.Hi);
                wi = reflect_idx(wi, a.Wi);
            } else {
                ok = ok && ((unsigned)hi < (unsigned)a.Hi) && ((unsigned)wi < (unsigned)a.Wi);
            }
            if (ok) {
                const size_t off = ((size_t)(x_nbase[i] + hi) * a.Wi + wi) * (size_t)a.in_pstride + ci;
                raw_load<TIn>(xr[i], in + off);
            } else {
                raw_zero<TIn>(xr[i]);
            }
        }
        // weights (rows padded to the tile, K padded: always in bounds)
#pragma unroll
        for (int i = 0; i < W_CH; ++i) {
            const int q = tid + i * 256;
            if (q < BN * CPR) {
                const int row = q / CPR, c = q % CPR;
                const size_t off = (size_t)(tn * BN + row) * a.w_kstride + kbase + kt * BK + c * 8;
                wr[0][i] = *reinterpret_cast<const u32x4_t *>(a.w_hi + off);
                if constexpr (PREC == 3) wr[1][i] = *reinterpret_cast<const u32x4_t *>(a.w_lo + off);
            }
        }
    };

    auto store_tile = [&](int buf) {
        bf16_t *base = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < X_CH; ++i) {
            const int q = tid + i * 256;
            if (q < BM * CPR) {
                const int row = q / CPR;
                u32x4_t hi, lo;
                raw_to_planes<TIn, PREC>(xr[i], a.in_act, hi, lo);
                const int e = (row * CPR + swz_chunk<CPR>(row, xc)) * 8;
                *reinterpret_cast<u32x4_t *>(base + e) = hi;
                if constexpr (PREC == 3) *reinterpret_cast<u32x4_t *>(base + XT + e) = lo;
            }
        }
#pragma unroll
        for (int i = 0; i < W_CH; ++i) {
            const int q = tid + i * 256;
            if (q < BN * CPR) {
                const int row = q / CPR, c = q % CPR;
                const int e = (row * CPR + swz_chunk<CPR>(row, c)) * 8;
                *reinterpret_cast<u32x4_t *>(base + NPL * XT + e) = wr[0][i];
                if constexpr (PREC == 3) *reinterpret_cast<u32x4_t *>(base + NPL * XT + WT + e) = wr[1][i];
            }
        }
    };

    f32x4_t acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    __syncthreads();     // tap table visible

    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        store_tile(0);
    }
    __syncthreads();

    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        const bool more = (kt + 1 < kt_end);
        if (more) load_tile(kt + 1);

        const bf16_t *base = smem + cur * BUF;
        const bf16_t *Xs = base, *Ws = base + NPL * XT;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8_t wf[NPL][FN], xf[NPL][FM];
            const int ch = kk * 4 + fg;
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                const int row = wn * PN + i * 16 + fr;
                const int e = (row * CPR + swz_chunk<CPR>(row, ch)) * 8;
                wf[0][i] = *reinterpret_cast<const bf16x8_t *>(Ws + e);
                if constexpr (PREC == 3) wf[1][i] = *reinterpret_cast<const bf16x8_t *>(Ws + WT + e);
            }
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                const int row = wm * PM + j * 16 + fr;
                const int e = (row * CPR + swz_chunk<CPR>(row, ch)) * 8;
                xf[0][j] = *reinterpret_cast<const bf16x8_t *>(Xs + e);
                if constexpr (PREC == 3) xf[1][j] = *reinterpret_cast<const bf16x8_t *>(Xs + XT + e);
            }
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) {
                    if constexpr (PREC == 3) {
                        acc[i][j] = dl_mfma16(wf[1][i], xf[0][j], acc[i][j]);
                        acc[i][j] = dl_mfma16(wf[0][i], xf[1][j], acc[i][j]);
                    }
                    acc[i][j] = dl_mfma16(wf[0][i], xf[0][j], acc[i][j]);
                }
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds channels co..co+3 of pixel (column fr) for each fragment
    const int oh = a.phase_oh[phase], ow = a.phase_ow[phase];
#pragma unroll
    for (int j = 0; j < FM; ++j) {
        const int m = tm * BM + wm * PM + j * 16 + fr;
        if (m >= a.Mtot) continue;
        const int n = m / HWq, rem = m - n * HWq;
        const int hq = rem / a.Wq, wq = rem - hq * a.Wq;
        // sub-pixel phases of an odd-sized output (stride-2 data gradient of a 25-row input: phase grid 13 x .., 12 odd rows):
        // the phase grid is ceil(Ho / out_step) and outputs that fall outside the tensor are dropped
        if (hq * a.out_step + oh >= a.Ho || wq * a.out_step + ow >= a.Wo) continue;
        const size_t opix = ((size_t)n * a.Ho + (hq * a.out_step + oh)) * a.Wo + (wq * a.out_step + ow);
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const int co = tn * BN + wn * PN + i * 16 + fg * 4;
            if (co >= a.Co) continue;
            f32x4_t v = acc[i][j];
            if (a.splitk > 1 || a.raw_out) {
                float *dst = a.slab + ((size_t)ks * ((size_t)a.N * a.Ho * a.Wo) + opix) * a.Co + co;
                *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                if (a.bias) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (co + r < a.bias_n) ? a.bias[co + r] : 0.f;
                }
                if (a.act != DL_ACT_NONE) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = apply_act(a.act, v[r]);
                }
                TOut *dst = reinterpret_cast<TOut *>(a.out) + opix * a.out_pstride + co;
                if constexpr (sizeof(TOut) == 4) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    u32x2_t p;
                    p[0] = pack2_bf16(v[0], v[1]);
                    p[1] = pack2_bf16(v[2], v[3]);
                    *reinterpret_cast<u32x2_t *>(dst) = p;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 fast path: both tiles go HBM -> LDS directly (global_load_lds_dwordx4, 16 B per lane), no VGPR round trip, no
// ds_write pass.  The LDS image of a wave-instruction is lane-linear (M0 base + lane*16), so the XOR chunk swizzle is
// applied on the SOURCE side: the lane that fills LDS slot (row, c') fetches global chunk c' ^ f(row) of that row; the
// fragment reads use the same involution.  Zero padding = lanes point at a zero page.  Two LDS buffers: the DMA of tile
// t+1 is issued before the MFMAs of tile t and drained (vmcnt(0)) at the single barrier per K step.
// ------------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(64))) unsigned char g_zero_page[64];

// Epilogue shared by the direct-to-LDS kernels: bias + activation + bf16 store (or raw fp32 slabs for split-K / raw_out) and the
// optional fused per-(image, channel) statistics of the stored values.  `smem_raw` must be dead (all tile reads done, no DMA
// in flight) when this is entered with statistics requested.
// sum over the 16 lanes of a DPP row (lanes with equal lane >> 4), result in every lane: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror,
// row_mirror -- four VALU adds instead of four ds_bpermute round trips
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
    return v;
}

// LDS bytes the store epilogue below needs: the bf16 output tile, one output-pixel index per tile row, the per-wave-row statistics
template <int BM, int BN, int WM> constexpr size_t epilogue_lds_bytes() { return (size_t)BM * BN * 2 + BM * sizeof(int) + (size_t)WM * 2 * BN * sizeof(float); }

// Store epilogue of the direct-to-LDS kernels (bf16 output, no split-K): the accumulators (rows = channels, 4 consecutive channels per
// lane, 16 pixels across the lanes of a DPP row) would store as 32-byte pieces 16 pixels apart; instead the tile is transposed through
// LDS (dead after the K loop; 8-byte units XOR-swizzled by the pixel so both the fragment writes and the 16-byte row reads are
// conflict-free) and written out as whole NHWC pixel rows, 16 bytes per lane.  Fused bias / activation / per-channel statistics of the
// stored (bf16-rounded) values as before; the 16-lane reductions of the statistics are DPP adds.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void tile_epilogue_lds(const ConvArgs &a, f32x4_t (&acc)[BN / WN / 16][BM / WM / 16], int tm, int tn, int phase,
                                                  int oh, int ow, int wm, int wn, int lane, int tid, char *smem_raw) {
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int PM = BM / WM, PN = BN / WN, FM = PM / 16, FN = PN / 16;
    // 8-byte units per tile row; unit bits the pixel index is XORed into.  Bit 0 stays (a 16-byte read needs units 2c, 2c+1 adjacent);
    // bits 1-4 select the bank pair, so XORing the pixel's low 4 bits there spreads the 16 lanes of a DPP row (16 pixels, same channels)
    // over all 32 bank pairs -- XORing into bits 2-5 left pixels p and p+8 on the same banks (1.6 M conflict cycles per launch by PMC)
    constexpr int UNITS = BN / 4, SWZ = (UNITS - 1) & ~1;
    constexpr int CH = BN / 8;                                  // 16-byte chunks per tile row
    char *tile = smem_raw;
    int *rowtab = reinterpret_cast<int *>(smem_raw + (size_t)BM * BN * 2);
    float *red = reinterpret_cast<float *>(rowtab + BM);        // [WM][2][BN]
    const int fr = lane & 15, fg = lane >> 4;
    const int HWq = a.Hq * a.Wq;
    const bool want_stats = a.stats_part != nullptr && a.bn_y == nullptr;

    // output pixel of every tile row (-1: outside the tensor -- ragged last tile, or a sub-pixel phase of an odd-sized output)
    for (int r = tid; r < BM; r += NT) {
        const int m = tm * BM + r;
        int opix = -1;
        if (m < a.Mtot) {
            const int n = m / HWq, rem = m - n * HWq;
            const int hq = rem / a.Wq, wq = rem - hq * a.Wq;
            const int ho = hq * a.out_step + oh, wo = wq * a.out_step + ow;
            if (ho < a.Ho && wo < a.Wo) opix = (n * a.Ho + ho) * a.Wo + wo;
        }
        rowtab[r] = opix;
    }
    bool live[FM];          // same test, for the statistics (per accumulator column of this lane)
#pragma unroll
    for (int j = 0; j < FM; ++j) {
        const int m = tm * BM + wm * PM + j * 16 + fr;
        live[j] = m < a.Mtot;
        if (a.out_step != 1 && live[j]) {
            const int n = m / HWq, rem = m - n * HWq;
            const int hq = rem / a.Wq, wq = rem - hq * a.Wq;
            live[j] = hq * a.out_step + oh < a.Ho && wq * a.out_step + ow < a.Wo;
        }
    }
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        const int cl = wn * PN + i * 16 + fg * 4;               // channel inside the tile
        const int co = tn * BN + cl;
        float bias[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bias[r] = (co + r < a.bias_n) ? a.bias[co + r] : 0.f;
        }
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        const int unit = (cl >> 2) ^ ((fr << 1) & SWZ);
        char *dst = tile + (size_t)(wm * PM + fr) * (BN * 2) + unit * 8;
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            f32x4_t v = acc[i][j];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += bias[r];
            if (a.act != DL_ACT_NONE) {       // one uniform branch per fragment (four instantiations of this epilogue behind a switch
#pragma unroll                         // made the compiler keep a scratch copy of the kernel arguments)
                for (int r = 0; r < 4; ++r) v[r] = apply_act(a.act, v[r]);
            }
            u32x2_t p;
            p[0] = pack2_bf16(v[0], v[1]);
            p[1] = pack2_bf16(v[2], v[3]);
            *reinterpret_cast<u32x2_t *>(dst + (size_t)j * 16 * (BN * 2)) = p;
            if (want_stats && live[j]) {       // statistics of exactly what is stored (bf16-rounded), like the stand-alone kernel sees
                const float q0 = h16_lo_f32(p[0]), q1 = h16_hi_f32(p[0]);
                const float q2 = h16_lo_f32(p[1]), q3 = h16_hi_f32(p[1]);
                s1[0] += q0; s2[0] += q0 * q0; s1[1] += q1; s2[1] += q1 * q1;
                s1[2] += q2; s2[2] += q2 * q2; s1[3] += q3; s2[3] += q3 * q3;
            }
        }
        if (want_stats) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[r] = row16_sum(s1[r]); s2[r] = row16_sum(s2[r]); }
            if (fr == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    red[(wm * 2 + 0) * BN + cl + r] = s1[r];
                    red[(wm * 2 + 1) * BN + cl + r] = s2[r];
                }
            }
        }
    }
    __syncthreads();
    bf16_t *out = reinterpret_cast<bf16_t *>(a.out);
    if (a.bn_y == nullptr) {
#pragma unroll 4
        for (int idx = tid; idx < BM * CH; idx += NT) {
            const int row = idx / CH, c = idx % CH;
            const int opix = rowtab[row];
            const int co = tn * BN + c * 8;
            if (opix < 0 || co >= a.Co) continue;
            const int unit = (c * 2) ^ (((row & 15) << 1) & SWZ);
            const u32x4_t v = *reinterpret_cast<const u32x4_t *>(tile + (size_t)row * (BN * 2) + unit * 8);
            *reinterpret_cast<u32x4_t *>(out + (size_t)opix * a.out_pstride + co) = v;
        }
    } else {
        // ---- store + the reductions of the following normalisation backward: this thread owns the 8 channels c*8.. of every (NT/CH)-th
        // tile row (NT is a multiple of CH, so c is fixed), reads the y values under the dz values it stores and accumulates
        // S1 = sum dn, S2 = sum dn * xhat for them; the NT/CH row groups are then combined through LDS in a fixed order.
        static_assert(NT % CH == 0, "one 16-byte column per thread");
        constexpr int RG = NT / CH;                                  // row groups
        const int c = tid % CH, rg = tid / CH;
        const int co = tn * BN + c * 8;
        const int n_img = (tm * BM) / HWq;                           // the tile lies in one image (host guarantees HWq % BM == 0)
        float mu[8], rs[8], sc[8], sh[8], s1[8], s2[8];
        const bool cok = co < a.Co;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ci = n_img * a.Co + (cok ? co + i : 0);
            mu[i] = a.bn_mean[ci]; rs[i] = a.bn_rstd[ci]; sc[i] = a.bn_scale[ci]; sh[i] = a.bn_shift[ci];
            s1[i] = s2[i] = 0.f;
        }
        // all y loads of this thread are issued before the first one is used (16 x 16 B in flight per thread on the 256-pixel tile;
        // a load-use-load chain here cost +40 us per launch: 16 dependent HBM round trips per tile)
        constexpr int ITER = BM / RG;
        static_assert(BM % RG == 0, "row groups tile the rows");
        u32x4_t yv[ITER];
        int opx[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            opx[it] = cok ? rowtab[rg + it * RG] : -1;
            yv[it] = u32x4_t{0u, 0u, 0u, 0u};
            if (opx[it] >= 0) yv[it] = *reinterpret_cast<const u32x4_t *>(a.bn_y + (size_t)opx[it] * a.bn_y_pstride + co);
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            if (opx[it] < 0) continue;
            const int row = rg + it * RG;
            const int unit = (c * 2) ^ (((row & 15) << 1) & SWZ);
            const u32x4_t v = *reinterpret_cast<const u32x4_t *>(tile + (size_t)row * (BN * 2) + unit * 8);
            *reinterpret_cast<u32x4_t *>(out + (size_t)opx[it] * a.out_pstride + co) = v;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned dw = v[i >> 1], yw = yv[it][i >> 1];
                float dn = (i & 1) ? h16_hi_f32(dw) : h16_lo_f32(dw);
                const float yy = (i & 1) ? h16_hi_f32(yw) : h16_lo_f32(yw);
                const float nv = yy * sc[i] + sh[i];
                if (a.bn_act == DL_ACT_RELU) dn = nv > 0.f ? dn : 0.f;
                else if (a.bn_act == DL_ACT_LRELU) dn = nv > 0.f ? dn : 0.2f * dn;
                s1[i] += dn;
                s2[i] += dn * (yy - mu[i]) * rs[i];
            }
        }
        __syncthreads();                                             // every thread is done reading the tile: reuse it
        float *bred = reinterpret_cast<float *>(tile);               // [RG][2][BN]
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            bred[(rg * 2 + 0) * BN + c * 8 + i] = s1[i];
            bred[(rg * 2 + 1) * BN + c * 8 + i] = s2[i];
        }
        __syncthreads();
        const int m0 = tm * BM;
        const int chunk = ((m0 - n_img * HWq) / BM) * a.n_phase + phase;
        for (int cc = tid; cc < 2 * BN; cc += NT) {
            const int which = cc / BN, ch = cc % BN;
            if (tn * BN + ch >= a.Co) continue;
            float t = 0.f;
#pragma unroll 4
            for (int r = 0; r < RG; ++r) t += bred[(r * 2 + which) * BN + ch];
            a.stats_part[((size_t)(n_img * a.stats_nchunks + chunk) * 2 + which) * a.Co + tn * BN + ch] = t;
        }
        return;
    }
    if (want_stats) {
        // every pixel of this tile lies in ONE image (host guarantees HWq % BM == 0): chunk = (tile in image, phase)
        const int m0 = tm * BM;
        const int n = m0 / HWq;
        const int chunk = ((m0 - n * HWq) / BM) * a.n_phase + phase;
        for (int c = tid; c < BN; c += NT) {
            const int co = tn * BN + c;
            if (co < a.Co) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) { t1 += red[(w * 2 + 0) * BN + c]; t2 += red[(w * 2 + 1) * BN + c]; }
                float *o = a.stats_part + ((size_t)(n * a.stats_nchunks + chunk) * 2) * a.Co + co;
                o[0] = t1;
                o[a.Co] = t2;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void glds_epilogue(const ConvArgs &a, f32x4_t (&acc)[BN / WN / 16][BM / WM / 16], int tm, int tn, int phase, int ks,
                                              int wm, int wn, int lane, int tid, char *smem_raw) {
    if constexpr (BN >= 64) {
        static_assert(BM % 16 == 0, "tile rows");
        if (a.splitk == 1 && !a.raw_out && !a.epi_old) {      // bf16 result: whole-row stores through LDS
            const int oh0 = a.phase_oh[phase], ow0 = a.phase_ow[phase];
            tile_epilogue_lds<BM, BN, WM, WN>(a, acc, tm, tn, phase, oh0, ow0, wm, wn, lane, tid, smem_raw);
            return;
        }
    }
    constexpr int NW = WM * WN;
    constexpr int PM = BM / WM, PN = BN / WN, FM = PM / 16, FN = PN / 16;
    const int fr = lane & 15, fg = lane >> 4;
    const int HWq = a.Hq * a.Wq;
    // ---- epilogue (as the register-staged kernel) + optional fused per-channel statistics of the stored values
    const int oh = a.phase_oh[phase], ow = a.phase_ow[phase];
    const bool want_stats = a.stats_part != nullptr;
    float st1[FN][4], st2[FN][4];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) st1[i][r] = st2[i][r] = 0.f;
#pragma unroll
    for (int j = 0; j < FM; ++j) {
        const int m = tm * BM + wm * PM + j * 16 + fr;
        if (m >= a.Mtot) continue;
        const int n = m / HWq, rem = m - n * HWq;
        const int hq = rem / a.Wq, wq = rem - hq * a.Wq;
        // sub-pixel phases of an odd-sized output (stride-2 data gradient of a 25-row input: phase grid 13 x .., 12 odd rows):
        // the phase grid is ceil(Ho / out_step) and outputs that fall outside the tensor are dropped
        if (hq * a.out_step + oh >= a.Ho || wq * a.out_step + ow >= a.Wo) continue;
        const size_t opix = ((size_t)n * a.Ho + (hq * a.out_step + oh)) * a.Wo + (wq * a.out_step + ow);
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const int co = tn * BN + wn * PN + i * 16 + fg * 4;
            if (co >= a.Co) continue;
            f32x4_t v = acc[i][j];
            if (a.splitk > 1 || a.raw_out) {
                float *dst = a.slab + ((size_t)ks * ((size_t)a.N * a.Ho * a.Wo) + opix) * a.Co + co;
                *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                if (a.bias) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (co + r < a.bias_n) ? a.bias[co + r] : 0.f;
                }
                if (a.act != DL_ACT_NONE) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = apply_act(a.act, v[r]);
                }
                bf16_t *dst = reinterpret_cast<bf16_t *>(a.out) + opix * a.out_pstride + co;
                u32x2_t p;
                p[0] = pack2_bf16(v[0], v[1]);
                p[1] = pack2_bf16(v[2], v[3]);
                *reinterpret_cast<u32x2_t *>(dst) = p;
                if (want_stats) {      // statistics of exactly what was stored (bf16-rounded), like the stand-alone kernel sees
                    const float q0 = h16_lo_f32(p[0]), q1 = h16_hi_f32(p[0]);
                    const float q2 = h16_lo_f32(p[1]), q3 = h16_hi_f32(p[1]);
                    st1[i][0] += q0; st2[i][0] += q0 * q0; st1[i][1] += q1; st2[i][1] += q1 * q1;
                    st1[i][2] += q2; st2[i][2] += q2 * q2; st1[i][3] += q3; st2[i][3] += q3 * q3;
                }
            }
        }
    }
    if (want_stats) {
        // lanes fr = 0..15 of one fg hold different pixels of the same 4 channels: butterfly over lane bits 0..3
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { st1[i][r] += __shfl_xor(st1[i][r], o, 64); st2[i][r] += __shfl_xor(st2[i][r], o, 64); }
            }
        float *red = reinterpret_cast<float *>(smem_raw);          // [WM][2][BN]; the tile buffers are dead after the K loop
        if (fr == 0) {
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = wn * PN + i * 16 + fg * 4 + r;
                    red[(wm * 2 + 0) * BN + c] = st1[i][r];
                    red[(wm * 2 + 1) * BN + c] = st2[i][r];
                }
        }
        __syncthreads();
        // every pixel of this tile lies in ONE image (host guarantees HWq % BM == 0): chunk = (tile in image, phase)
        const int m0 = tm * BM;
        const int n = m0 / HWq;
        const int chunk = ((m0 - n * HWq) / BM) * a.n_phase + phase;
        for (int c = tid; c < BN; c += NW * 64) {
            const int co = tn * BN + c;
            if (co < a.Co) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) { s1 += red[(w * 2 + 0) * BN + c]; s2 += red[(w * 2 + 1) * BN + c]; }
                float *o = a.stats_part + ((size_t)(n * a.stats_nchunks + chunk) * 2) * a.Co + co;
                o[0] = s1;
                o[a.Co] = s2;
            }
        }
    }
}


template <int BM, int BN, int BK, int WM, int WN, bool UTAP, bool STAG = false, int ABL = 0>
__global__ void __launch_bounds__(WM * WN * 64) conv_gemm_glds_kernel(const ConvArgs a) {
    constexpr int NW = WM * WN;                        // waves per workgroup (4 or 8)
    constexpr int CPR = BK / 8;                        // 16-byte chunks per LDS row
    constexpr int RPI = 64 / CPR;                      // tile rows filled by one wave-instruction
    constexpr int X_INS = (BM + NW * RPI - 1) / (NW * RPI);   // glds instructions per wave, activation tile
    constexpr int W_INS = (BN + NW * RPI - 1) / (NW * RPI);
    constexpr int PM = BM / WM, PN = BN / WN, FM = PM / 16, FN = PN / 16;
    constexpr int XT = BM * BK, WT = BN * BK, BUF = XT + WT;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t *smem = reinterpret_cast<bf16_t *>(smem_raw);
    int16_t *tap_lds = reinterpret_cast<int16_t *>(smem + 2 * BUF);
    int *tapd_lds = reinterpret_cast<int *>(tap_lds + DL_MAX_TAPS);     // element offset (dh*Wi + dw)*pstride of every tap

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;

    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % a.tiles_n, tm = bid / a.tiles_n;
    const int phase = blockIdx.y / a.splitk, ks = blockIdx.y % a.splitk;
    const int tap0 = a.phase_tap_begin[phase];
    const int ntaps = a.phase_tap_begin[phase + 1] - tap0;
    const int kbase = a.phase_kbase[phase];
    const int nk_total = (ntaps * a.Ci + 63) / 64 * 64 / BK;
    const int nk_per = (nk_total + a.splitk - 1) / a.splitk;
    const int kt_begin = ks * nk_per;
    const int kt_end = min(nk_total, kt_begin + nk_per);

    if (tid < DL_MAX_TAPS) {
        const int16_t tp = a.taps[tid];
        tap_lds[tid] = tp;
        tapd_lds[tid] = ((int)(int8_t)(tp & 0xff) * a.Wi + (int)(int8_t)((tp >> 8) & 0xff)) * a.in_pstride;
    }

    const bf16_t *in = reinterpret_cast<const bf16_t *>(a.in);
    const bf16_t *zero = reinterpret_cast<const bf16_t *>(g_zero_page);
    const int HWq = a.Hq * a.Wq;
    const int lrow = lane / CPR, lcp = lane % CPR;    // row within the instruction's slab, LDS chunk position

    // ---- per-thread geometry: row r_i = (wave * X_INS + i) * RPI + lrow of the activation tile
    const bf16_t *x_ptr[X_INS];          // base pixel + this lane's swizzled chunk offset
    int x_hi0[X_INS], x_wi0[X_INS], x_chunk[X_INS];
    bool x_ok[X_INS];
    unsigned long long x_mask[X_INS];    // bit t: tap (tap0 + t) of this row is inside the image (zero padding)
#pragma unroll
    for (int i = 0; i < X_INS; ++i) {
        const int row = (wave * X_INS + i) * RPI + lrow;
        const int m = (ABL == 5 ? (tm & 1) : tm) * BM + row;      // ABL 5: every block gathers the same two slabs (all L2 hits)
        x_ok[i] = (row < BM) && (m < a.Mtot);
        const int mm = x_ok[i] ? m : 0;
        const int n = mm / HWq, rem = mm - n * HWq;
        const int hq = rem / a.Wq, wq = rem - hq * a.Wq;
        x_hi0[i] = hq * a.in_step;
        x_wi0[i] = wq * a.in_step;
        x_chunk[i] = swz_chunk<CPR>(row, lcp);        // global chunk this lane fetches for its LDS slot
        x_ptr[i] = in + ((size_t)(n * a.Hi + x_hi0[i]) * a.Wi + x_wi0[i]) * (size_t)a.in_pstride + x_chunk[i] * 8;
        unsigned long long mk = 0;
        if (UTAP && x_ok[i]) {               // (the generic path checks bounds per chunk instead: many taps, short K loops)
            for (int t = 0; t < ntaps; ++t) {
                const int16_t tp = a.taps[tap0 + t];
                const int hi = x_hi0[i] + (int)(int8_t)(tp & 0xff), wi = x_wi0[i] + (int)(int8_t)((tp >> 8) & 0xff);
                if (a.pad_mode == DL_PAD_REFLECT || (((unsigned)hi < (unsigned)a.Hi) && ((unsigned)wi < (unsigned)a.Wi))) mk |= 1ull << t;
            }
        }
        x_mask[i] = mk;
    }
    const bf16_t *w_ptr[W_INS];
#pragma unroll
    for (int i = 0; i < W_INS; ++i) {
        const int row = (wave * W_INS + i) * RPI + lrow;
        const int c = swz_chunk<CPR>(row, lcp);
        // rows beyond the tile (narrow-N configs) re-read row 0: their LDS slots are never consumed
        w_ptr[i] = a.w_hi + (size_t)(tn * BN + (row < BN ? row : 0)) * a.w_kstride + kbase + c * 8;
    }

    // UTAP K-step sequencer (wave-uniform): the step being issued covers channels [is_ch, is_ch + BK) of tap is_tl.
    //   k_order 0: tap-major (all channel chunks of a tap, then the next tap) -- the packed K order
    //   k_order 1: channel-chunk-major (all taps of a 64-channel chunk back to back): the taps re-read the SAME halo slab
    //              chunk from L2 within a few steps instead of after a full pass over Ci, which is what keeps the slab L2-resident
    int is_tl = 0, is_ch = 0;
    int tapd_next = 0;                   // pixel offset of tap is_tl (read from LDS one step ahead of its use)
    auto issue_tile = [&](int kt, int buf) {
        bf16_t *base = smem + buf * BUF;
        size_t wk;
        if constexpr (UTAP) {
            const int tl = is_tl;
            const ptrdiff_t delta = (ptrdiff_t)tapd_next + is_ch;
            wk = (size_t)tl * a.Ci + is_ch;
            if (a.k_order) { if (++is_tl == ntaps) { is_tl = 0; is_ch += BK; } }
            else { is_ch += BK; if (is_ch == a.Ci) { is_ch = 0; ++is_tl; } }
            tapd_next = tapd_lds[tap0 + min(is_tl, ntaps - 1)];    // for the next call: latency hides behind this step's MFMAs
#pragma unroll
            for (int i = 0; i < X_INS; ++i) {
                const int row0 = (wave * X_INS + i) * RPI;
                if (row0 < BM) {
                    const bool ok = (x_mask[i] >> tl) & 1ull;
                    const bf16_t *src = ok ? x_ptr[i] + delta : zero;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(base + row0 * BK), 16, 0, 0);
                }
            }
        } else {
            wk = (size_t)kt * BK;
#pragma unroll
            for (int i = 0; i < X_INS; ++i) {
                const int row0 = (wave * X_INS + i) * RPI;
                if (row0 < BM) {
                    const int k0 = kt * BK + x_chunk[i] * 8;
                    const int tl = k0 >> a.log2Ci;
                    const int ci = k0 & (a.Ci - 1);
                    const int tli = tl < ntaps ? tl : 0;
                    const int16_t t = tap_lds[tap0 + tli];
                    const int dh = (int)(int8_t)(t & 0xff), dw = (int)(int8_t)((t >> 8) & 0xff);
                    int hi = x_hi0[i] + dh, wi = x_wi0[i] + dw;
                    if (a.pad_mode == DL_PAD_REFLECT) { hi = reflect_idx(hi, a.Hi); wi = reflect_idx(wi, a.Wi); }
                    const bool ok = x_ok[i] && (tl < ntaps) && ((unsigned)hi < (unsigned)a.Hi) && ((unsigned)wi < (unsigned)a.Wi);
                    const ptrdiff_t off = ((ptrdiff_t)(hi - x_hi0[i]) * a.Wi + (wi - x_wi0[i])) * (ptrdiff_t)a.in_pstride + ci - x_chunk[i] * 8;
                    const bf16_t *src = ok ? x_ptr[i] + off : zero;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(base + row0 * BK), 16, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < W_INS; ++i) {
            const int row0 = (wave * W_INS + i) * RPI;
            if (row0 < BN) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(w_ptr[i] + wk),
                                                 (__attribute__((address_space(3))) void *)(base + XT + row0 * BK), 16, 0, 0);
            }
        }
    };

    f32x4_t acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    __syncthreads();     // tap tables visible
    if constexpr (UTAP) {
        if (a.k_order) { is_tl = kt_begin % ntaps; is_ch = (kt_begin / ntaps) * BK; }
        else { is_tl = (kt_begin * BK) >> a.log2Ci; is_ch = (kt_begin * BK) & (a.Ci - 1); }
        tapd_next = tapd_lds[tap0 + min(is_tl, ntaps - 1)];
    }
    if (kt_begin < kt_end) issue_tile(kt_begin, 0);
    __syncthreads();     // drains the DMA (vmcnt(0)) + barrier

    const int fr = lane & 15, fg = lane >> 4;
    // fragment offsets: row = 16*j + fr (+ wave base, multiples of 16) so the swizzle term depends on the lane only;
    // per kk the chunk is (kk*4 + fg) ^ s  ->  two lane-constant element offsets, everything else is an immediate
    int foff[BK / 32];
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) foff[kk] = (fr * CPR + swz_chunk<CPR>(fr, kk * 4 + fg)) * 8;
    auto read_frags = [&](const bf16_t *Xs, const bf16_t *Ws, int kk, bf16x8_t (&wf)[FN], bf16x8_t (&xf)[FM]) {
#pragma unroll
        for (int i = 0; i < FN; ++i) wf[i] = *reinterpret_cast<const bf16x8_t *>(Ws + (wn * PN + i * 16) * BK + foff[kk]);
#pragma unroll
        for (int j = 0; j < FM; ++j) xf[j] = *reinterpret_cast<const bf16x8_t *>(Xs + (wm * PM + j * 16) * BK + foff[kk]);
    };
    auto mma = [&](const bf16x8_t (&wf)[FN], const bf16x8_t (&xf)[FM]) {
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = dl_mfma16(wf[i], xf[j], acc[i][j]);
    };

    if constexpr (STAG && NW == 8 && BK == 64) {
        // EXPERIMENT (DL_CONV_STAGGER=1; measured round 1: 0.188 ms vs 0.176 ms for the plain loop on the 3x3 256->256 conv --
        // the strictly sequential phases lose more intra-wave overlap than the stagger wins; kept for the next tuning pass).
        // Two waves share each SIMD (wave w and w+4).  The second group runs half a K step behind: it keeps the fragments of
        // the second K half in registers across the barrier and multiplies them while the first group is fetching its
        // fragments from LDS (and vice versa), so the LDS reads of one group overlap the MFMAs of the other instead of all
        // eight waves queueing on the LDS at once after every barrier.
        const bool late = wave >= 4;
        bf16x8_t wf[FN], xf[FM];                 // ONE fragment set; for the late group it is carried across the barrier
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const int cur = (kt - kt_begin) & 1;
            if (kt + 1 < kt_end) issue_tile(kt + 1, cur ^ 1);
            const bf16_t *Xs = smem + cur * BUF, *Ws = Xs + XT;
            // phases are kept strictly sequential inside a wave (sched_barrier): the overlap comes from the partner wave
            if (late && kt > kt_begin) mma(wf, xf);          // second K half of the previous tile, from registers
            __builtin_amdgcn_sched_barrier(0);
            read_frags(Xs, Ws, 0, wf, xf);
            __builtin_amdgcn_sched_barrier(0);
            mma(wf, xf);
            __builtin_amdgcn_sched_barrier(0);
            read_frags(Xs, Ws, 1, wf, xf);                   // late group: must land before the barrier (buffer refilled next step)
            __builtin_amdgcn_sched_barrier(0);
            if (!late) mma(wf, xf);
            __syncthreads();
        }
        if (late && kt_begin < kt_end) mma(wf, xf);
    } else {
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const int cur = (kt - kt_begin) & 1;
            if (ABL != 1 && kt + 1 < kt_end) issue_tile(kt + 1, cur ^ 1);      // ABL: profiling ablations (tools/ablate.sh)
            const bf16_t *Xs = smem + cur * BUF, *Ws = Xs + XT;
            if constexpr (ABL == 4) {        // DMA only, TWO tiles in flight (nothing reads LDS, so the buffers may alias):
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // separates a latency bound from a throughput bound
                __builtin_amdgcn_s_barrier();
                continue;
            }
            if (ABL != 2 && ABL != 4 && ABL != 5) {
#pragma unroll
                for (int kk = 0; kk < BK / 32; ++kk) {
                    bf16x8_t wf[FN], xf[FM];
                    read_frags(Xs, Ws, kk, wf, xf);
                    mma(wf, xf);
                }
            }
            __syncthreads();
        }
    }

    glds_epilogue<BM, BN, WM, WN>(a, acc, tm, tn, phase, ks, wm, wn, lane, tid, smem_raw);
}

// ------------------------------------------------------------------------------------------------------------------
// 8-phase kernel for the dominant layers (3x3 / 4x4 taps over >= 64 channels, Co a multiple of 256, zero padding, bf16).
// Same tile as the 256x256x64 direct-to-LDS kernel above (8 waves = 2 x 4, each 128 pixels x 64 channels), different schedule:
//   * a K step is split into FOUR phases, one 64-pixel x 32-channel accumulator quadrant x K=64 (16 MFMAs) each, with two raw
//     s_barriers per phase; waves 4-7 run ONE barrier behind waves 0-3 (wave w and w+4 share a SIMD), so on every SIMD one
//     wave is in its MFMA section while the other one issues its ds_reads / DMA: the LDS phase and the MFMA phase of the
//     plain kernel (which run back to back because the barrier keeps all 8 waves in lock step) overlap instead.
//   * the operands of a K step live in four 16 KB half-tile slots ordered by USE, not by position: WB_b = the b-th 32-channel
//     half of every wave's 64 channels, XA_a = the a-th 64-pixel half of every wave's 128 pixels.  A slot is therefore dead
//     as soon as its phase has read it and is refilled (global_load_lds) one or two phases later, three half-tiles ahead of
//     their use:        phase q of step t   reads                stages (2 DMA instructions per wave)
//                             0             WB_0(t), XA_0(t)      XA_1(t+1)
//                             1             WB_1(t)               WB_0(t+2)
//                             2             XA_1(t)               XA_0(t+2)
//                             3             -                     WB_1(t+2)      + s_waitcnt vmcnt(6): step t+1 has landed
//     vmcnt is never 0 in the steady state (the three newest half-tiles stay in flight across the barriers).
//   Ordering rules the table obeys (MI355X guide, "256^2 8-phase template"): a staged slot is read at the earliest one phase
//   after the counted vmcnt + barrier that retires it; a slot is restaged two phases after its last ds_read, or one phase
//   after when those reads were retired (lgkmcnt) before the reading phase's first barrier (WB_0: the lgkmcnt(8) in phase 0).
// ------------------------------------------------------------------------------------------------------------------
#define DL_BAR() asm volatile("s_barrier" ::: "memory")
template <int V> struct IC { static constexpr int value = V; };

template <int ABL>      // ABL != 0: timing-only ablations (tools/ablate.sh): 1 = no DMA in the loop, 2 = no ds_reads / MFMAs, 3 = MFMAs only
__global__ void __launch_bounds__(512) conv_gemm_8ph_kernel(const ConvArgs a) {
    constexpr int BM = 256, BN = 256, BK = 64, WM = 2, WN = 4, CPR = 8;
    constexpr int FM = 8, FN = 4;
    constexpr int HALF = 128 * BK;                 // elements of one half-tile slot (16 KB)
    constexpr int S_W = 0, S_X = 2;                // slot order inside a buffer: WB_0, WB_1, XA_0, XA_1

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t *smem = reinterpret_cast<bf16_t *>(smem_raw);
    int *tapd_lds = reinterpret_cast<int *>(smem + 8 * HALF);          // element offset (dh*Wi + dw)*pstride of every tap

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const bool grp1 = wave >= 4;

    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % a.tiles_n, tm = bid / a.tiles_n;
    const int phase = blockIdx.y / a.splitk, ks = blockIdx.y % a.splitk;
    const int tap0 = a.phase_tap_begin[phase];
    const int ntaps = a.phase_tap_begin[phase + 1] - tap0;
    const int kbase = a.phase_kbase[phase];
    const int nk_total = ntaps * a.Ci / BK;
    const int nk_per = (nk_total + a.splitk - 1) / a.splitk;
    const int kt_begin = ks * nk_per;
    const int T = ABL == 4 ? 0 : max(min(nk_total, kt_begin + nk_per) - kt_begin, 0);       // ABL 4: prologue + epilogue only

    if (tid < DL_MAX_TAPS) {
        const int16_t tp = a.taps[tid];
        tapd_lds[tid] = ((int)(int8_t)(tp & 0xff) * a.Wi + (int)(int8_t)((tp >> 8) & 0xff)) * a.in_pstride;
    }

    const bf16_t *in = reinterpret_cast<const bf16_t *>(a.in);
    const bf16_t *zero = reinterpret_cast<const bf16_t *>(g_zero_page);
    const int HWq = a.Hq * a.Wq;
    const int lrow = lane >> 3, lcp = lane & 7;    // row inside one DMA instruction's 8-row slab, 16-byte position in the row

    // ---- staging geometry: instruction i of this wave fills slot rows s = (wave*2 + i)*8 + lrow of a half-tile
    const bf16_t *x_ptr[2][2];
    unsigned long long x_mask[2][2];               // bit t: tap (tap0 + t) of this pixel is inside the image
    const bf16_t *w_ptr[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s = (wave * 2 + i) * 8 + lrow;
            const int c8 = swz_chunk<CPR>(s, lcp) * 8;
            {   // XA_h: slot row s = pixel (s>>6)*128 + h*64 + (s&63) of the tile
                const int m = tm * BM + (s >> 6) * 128 + h * 64 + (s & 63);
                const bool ok = m < a.Mtot;
                const int mm = ok ? m : 0;
                const int n = mm / HWq, rem = mm - n * HWq;
                const int hq = rem / a.Wq, wq = rem - hq * a.Wq;
                const int hi0 = hq * a.in_step, wi0 = wq * a.in_step;
                x_ptr[h][i] = in + ((size_t)(n * a.Hi + hi0) * a.Wi + wi0) * (size_t)a.in_pstride + c8;
                unsigned long long mk = 0;
                if (ok)
                    for (int t = 0; t < ntaps; ++t) {
                        const int16_t tp = a.taps[tap0 + t];
                        const int hi = hi0 + (int)(int8_t)(tp & 0xff), wi = wi0 + (int)(int8_t)((tp >> 8) & 0xff);
                        if (((unsigned)hi < (unsigned)a.Hi) && ((unsigned)wi < (unsigned)a.Wi)) mk |= 1ull << t;
                    }
                x_mask[h][i] = mk;
            }
            {   // WB_h: slot row s = channel (s>>5)*64 + h*32 + (s&31) of the tile
                const int row = (s >> 5) * 64 + h * 32 + (s & 31);
                w_ptr[h][i] = a.w_hi + (size_t)(tn * BN + row) * a.w_kstride + kbase + c8;
            }
        }

    auto stage_x = [&](auto BUF, auto H, ptrdiff_t delta, int tl) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::value, h = decltype(H)::value;
        bf16_t *dst = smem + (buf * 4 + S_X + h) * HALF + wave * (16 * BK);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool ok = (x_mask[h][i] >> tl) & 1ull;
            const bf16_t *src = ok ? x_ptr[h][i] + delta : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + i * 8 * BK), 16, 0, 0);
        }
    };
    auto stage_w = [&](auto BUF, auto H, size_t wk) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::value, h = decltype(H)::value;
        bf16_t *dst = smem + (buf * 4 + S_W + h) * HALF + wave * (16 * BK);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(w_ptr[h][i] + wk),
                                             (__attribute__((address_space(3))) void *)(dst + i * 8 * BK), 16, 0, 0);
    };

    // K step u (counted from kt_begin) covers channels [ch, ch + BK) of tap tl: tap-major order, stateless (wave-uniform SALU)
    // a.k_order8 (default; DL_8PH_KORDER=0 = tap-major): channel-chunk-major, as in conv_gemm_8ph_x3_kernel (conv_x3.h)
#define DL_STEP_TL(u) (a.k_order8 ? ((kt_begin + (u)) % ntaps) : (((kt_begin + (u)) * BK) >> a.log2Ci))
#define DL_STEP_CH(u) (a.k_order8 ? (((kt_begin + (u)) / ntaps) * BK) : (((kt_begin + (u)) * BK) & (a.Ci - 1)))

    f32x4_t acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    __syncthreads();     // tap table visible

    // per-step scalars of the steps whose half-tiles are still to be staged: step t+1 (x only) and t+2
    int tl1, tl2;
    ptrdiff_t d1, d2;
    size_t wk2;
    {
        const int tl0 = DL_STEP_TL(0), ch0 = DL_STEP_CH(0);
        tl1 = DL_STEP_TL(1); const int ch1 = DL_STEP_CH(1);
        tl2 = DL_STEP_TL(2); const int ch2 = DL_STEP_CH(2);
        const ptrdiff_t d0 = (ptrdiff_t)tapd_lds[tap0 + min(tl0, ntaps - 1)] + ch0;
        d1 = (ptrdiff_t)tapd_lds[tap0 + min(tl1, ntaps - 1)] + ch1;
        d2 = (ptrdiff_t)tapd_lds[tap0 + min(tl2, ntaps - 1)] + ch2;
        const size_t wk0 = (size_t)tl0 * a.Ci + ch0, wk1 = (size_t)tl1 * a.Ci + ch1;
        wk2 = (size_t)tl2 * a.Ci + ch2;
        // prologue: step 0 completely, step 1 without XA_1 (phase 0 of step 0 stages it)
        if (T > 0) {
            stage_w(IC<0>{}, IC<0>{}, wk0);
            stage_x(IC<0>{}, IC<0>{}, d0, tl0);
            stage_w(IC<0>{}, IC<1>{}, wk0);
            stage_x(IC<0>{}, IC<1>{}, d0, tl0);
        }
        if (T > 1) {
            stage_w(IC<1>{}, IC<0>{}, wk1);
            stage_x(IC<1>{}, IC<0>{}, d1, tl1);
            stage_w(IC<1>{}, IC<1>{}, wk1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    DL_BAR();
    if (grp1) DL_BAR();          // stagger: waves 4-7 run one barrier behind

    const int fr = lane & 15, fg = lane >> 4;
    int foff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) foff[kk] = (fr * CPR + swz_chunk<CPR>(fr, kk * 4 + fg)) * 8;

    bf16x8_t xf[2][4], wf0[2][2], wf1[2][2];
    if constexpr (ABL >= 2) {           // ablations that skip the ds_reads still multiply something non-trivial
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) xf[kk][j] = bf16x8_t{(short)(0x3f80 + lane), (short)(0x3f00 + j), 0x3e80, 0x3f81, (short)(0xbf80 + kk), 0x3f10, 0x3e90, 0x3f91};
#pragma unroll
            for (int i = 0; i < 2; ++i) { wf0[kk][i] = xf[kk][i]; wf1[kk][i] = xf[kk][i + 2]; }
        }
    }
    auto read_x = [&](auto BUF, auto H) __attribute__((always_inline)) {
        if (ABL >= 2) return;
        constexpr int buf = decltype(BUF)::value, h = decltype(H)::value;
        const bf16_t *base = smem + (buf * 4 + S_X + h) * HALF + wm * (64 * BK);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 4; ++j) xf[kk][j] = *reinterpret_cast<const bf16x8_t *>(base + j * 16 * BK + foff[kk]);
    };
    auto read_w = [&](auto BUF, auto H, bf16x8_t (&wf)[2][2]) {
        if (ABL >= 2) return;
        constexpr int buf = decltype(BUF)::value, h = decltype(H)::value;
        const bf16_t *base = smem + (buf * 4 + S_W + h) * HALF + wn * (32 * BK);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i) wf[kk][i] = *reinterpret_cast<const bf16x8_t *>(base + i * 16 * BK + foff[kk]);
    };
    auto mma_q = [&](auto IB, auto JA, const bf16x8_t (&wf)[2][2]) {
        if (ABL == 2) return;
        constexpr int ib = decltype(IB)::value * 2, ja = decltype(JA)::value * 4;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[ib + i][ja + j] = dl_mfma16(wf[kk][i], xf[kk][j], acc[ib + i][ja + j]);
        __builtin_amdgcn_s_setprio(0);
    };

    auto step = [&](int t, auto BUF) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::value;
        const bool more1 = t + 1 < T, more2 = t + 2 < T;
        // ---- phase 0
        read_w(IC<buf>{}, IC<0>{}, wf0);
        __builtin_amdgcn_sched_barrier(0);
        read_x(IC<buf>{}, IC<0>{});
        __builtin_amdgcn_sched_barrier(0);
        if (more1 && ABL != 1 && ABL != 3) stage_x(IC<buf ^ 1>{}, IC<1>{}, d1, tl1);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");      // the 4 WB_0 reads (issued first) are done: WB_0 is restaged next phase
        DL_BAR();
        mma_q(IC<0>{}, IC<0>{}, wf0);
        DL_BAR();
        // ---- phase 1
        read_w(IC<buf>{}, IC<1>{}, wf1);
        if (more2 && ABL != 1 && ABL != 3) stage_w(IC<buf>{}, IC<0>{}, wk2);
        DL_BAR();
        mma_q(IC<1>{}, IC<0>{}, wf1);
        DL_BAR();
        // ---- phase 2
        read_x(IC<buf>{}, IC<1>{});
        if (more2 && ABL != 1 && ABL != 3) stage_x(IC<buf>{}, IC<0>{}, d2, tl2);
        DL_BAR();
        mma_q(IC<1>{}, IC<1>{}, wf1);
        DL_BAR();
        // ---- phase 3
        const int tl3 = DL_STEP_TL(t + 3), ch3 = DL_STEP_CH(t + 3);
        const ptrdiff_t d3 = (ptrdiff_t)tapd_lds[tap0 + min(tl3, ntaps - 1)] + ch3;
        if (more2 && ABL != 1 && ABL != 3) {
            stage_w(IC<buf>{}, IC<1>{}, wk2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // everything of step t+1 has landed; step t+2's three stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        DL_BAR();
        mma_q(IC<0>{}, IC<1>{}, wf0);
        DL_BAR();
        tl1 = tl2; d1 = d2;
        tl2 = tl3; d2 = d3; wk2 = (size_t)tl3 * a.Ci + ch3;
    };

    for (int t = 0; t < T; t += 2) {
        step(t, IC<0>{});
        if (t + 1 < T) step(t + 1, IC<1>{});
    }
    if (!grp1) DL_BAR();         // pairs with the last barrier of the trailing group
    __syncthreads();             // LDS is dead from here on (the epilogue reuses it for the statistics)

    glds_epilogue<BM, BN, WM, WN>(a, acc, tm, tn, phase, ks, wm, wn, lane, tid, smem_raw);
}

template <int ABL>
static int launch_conv_8ph(const ConvArgs &a0, hipStream_t stream) {
    ConvArgs a = a0;
    a.tiles_m = (a.Mtot + 255) / 256;
    a.tiles_n = a.Co / 256;
    // channel-chunk-major K order by default (DL_8PH_KORDER=0: tap-major).  Same-box A/B (r03, two alternations): step 101.64 / 101.88 ms tap-major,
    // 100.97 / 101.00 ms chunk-major (78.6 -> 79.2 tiles/s); isolated launches within noise (165-177 us).  r01 had measured the same idea on the
    // one-barrier kernel as a loss inside the step; on the 8-phase kernels it is a small gain for bf16 and -2.8 % of the strict step.
    static const char *korder = DL_DEV_ENV("DL_8PH_KORDER");
    a.k_order8 = (korder && korder[0] == '0') ? 0 : 1;
    constexpr size_t smem_loop = (size_t)8 * 128 * 64 * sizeof(bf16_t) + DL_MAX_TAPS * sizeof(int);
    constexpr size_t smem = smem_loop > epilogue_lds_bytes<256, 256, 2>() ? smem_loop : epilogue_lds_bytes<256, 256, 2>();
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gemm_8ph_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) DL_FAIL("dl_conv_forward: hipFuncSetAttribute(%zu): %s", smem, hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(a.tiles_m * a.tiles_n, a.n_phase * a.splitk);
    hipLaunchKernelGGL(conv_gemm_8ph_kernel<ABL>, grid, dim3(512), smem, stream, a);
    DL_CHECK_LAUNCH("dl_conv_forward(8-phase)");
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Software-pipelined kernel (v_mfma_f32_32x32x16_bf16): 256 pixels x 256 channels x 64 K per tile, 8 waves (2 x 4, each 128
// pixels x 64 channels = 2 x 4 blocks of 32 x 32).  One K=16 sub-step needs only 6 fragments (2 weight + 4 activation = 24
// VGPRs) for 8 MFMAs (256 matrix-pipe cycles), so the fragments are DOUBLE-BUFFERED in registers: every wave issues the
// ds_reads of sub-step s+1 before the MFMAs of sub-step s and overlaps its own LDS traffic with its own MFMAs -- the overlap
// the 16x16x32 kernels can only get from the partner wave (their 12 fragments per K=32 do not fit twice next to 128
// accumulators).  One barrier per 64-wide K tile; the DMA of tile t+2 starts right after the barrier that retires tile t.
//   KWR (kernel-column reuse, 3x3 stride-1 layers whose image rows are 128 pixels wide): the three kw taps of one
//   (kh, 64-channel chunk) read the SAME two image rows shifted by -1/0/+1 pixels.  The slab is staged once as
//   2 x (1 + 128 + 1) pixel rows (the pad columns are zeroed once) and the fragment reads of tap kw start dw+1 rows further
//   down: 32 KB of activation DMA per THREE K tiles instead of per tile (-33% of all staged bytes; the DMA path, ~18 B/clk/CU
//   whatever the depth, is what bounds these kernels).  K tiles run in (kh, chunk, kw) order.
// ------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <bool KWR, int ABL>
__global__ void __launch_bounds__(512) conv_gemm_p32_kernel(const ConvArgs a) {
    constexpr int BM = 256, BN = 256, BK = 64, CPR = 8;
    constexpr int XROWS = KWR ? 2 * 130 : 256;
    constexpr int XB = XROWS * 128, WB = 256 * 128;            // bytes of one activation / weight buffer

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char *xs = smem_raw;                                       // [2][XB]
    char *ws = smem_raw + 2 * XB;                              // [2][WB]
    int *tapd_lds = reinterpret_cast<int *>(ws + 2 * WB);      // element offset (dh*Wi + dw)*pstride of every tap
    int *tapw_lds = tapd_lds + DL_MAX_TAPS;                    // dw + 1 of every tap (KWR)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;

    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % a.tiles_n, tm = bid / a.tiles_n;
    const int phase = blockIdx.y / a.splitk, ks = blockIdx.y % a.splitk;
    const int tap0 = a.phase_tap_begin[phase];
    const int ntaps = a.phase_tap_begin[phase + 1] - tap0;
    const int kbase = a.phase_kbase[phase];
    const int nk_total = ntaps * a.Ci / BK;
    const int nk_per = (nk_total + a.splitk - 1) / a.splitk;
    const int kt_begin = ks * nk_per;
    const int T = max(min(nk_total, kt_begin + nk_per) - kt_begin, 0);
    const int log2nch = a.log2Ci - 6;                          // channel chunks per tap = Ci / 64

    if (tid < DL_MAX_TAPS) {
        const int16_t tp = a.taps[tid];
        const int dh = (int)(int8_t)(tp & 0xff), dw = (int)(int8_t)((tp >> 8) & 0xff);
        tapd_lds[tid] = (dh * a.Wi + (KWR ? 0 : dw)) * a.in_pstride;
        tapw_lds[tid] = dw + 1;
    }
    if (KWR && tid < 2 * 2 * 2 * 8) {          // zero the 2 x 2 pad rows of both slabs (16 bytes per thread)
        const int c = tid & 7, r = (tid >> 3) & 1, e = (tid >> 4) & 1, b = tid >> 5;
        *reinterpret_cast<u32x4_t *>(xs + b * XB + (r * 130 + e * 129) * 128 + c * 16) = u32x4_t{0, 0, 0, 0};
    }

    // K tile u (counted from kt_begin):  plain: tap-major (tap, chunk);  KWR: (kh, chunk, kw) with kw fastest
    auto tile_tap = [&](int u) __attribute__((always_inline)) {
        const int kt = kt_begin + u;
        if (KWR) { const int g = kt / 3; return (g >> log2nch) * 3 + (kt - g * 3); }
        return kt >> log2nch;
    };
    auto tile_ch = [&](int u) __attribute__((always_inline)) {
        const int kt = kt_begin + u;
        if (KWR) return ((kt / 3) & ((1 << log2nch) - 1)) * BK;
        return (kt & ((1 << log2nch) - 1)) * BK;
    };

    const bf16_t *in = reinterpret_cast<const bf16_t *>(a.in);
    const bf16_t *zero = reinterpret_cast<const bf16_t *>(g_zero_page);
    const int HWq = a.Hq * a.Wq;
    const int lrow = lane >> 3, lcp = lane & 7;

    // ---- staging geometry: instruction i of this wave fills tile rows s = (wave*4 + i)*8 + lrow
    const bf16_t *x_ptr[4];
    unsigned long long x_mask[4];
    const bf16_t *w_ptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = (wave * 4 + i) * 8 + lrow;
        const int xrow = KWR ? (s >> 7) * 130 + 1 + (s & 127) : s;          // LDS row of this pixel
        const int m = tm * BM + s;
        const bool ok = m < a.Mtot;
        const int mm = ok ? m : 0;
        const int n = mm / HWq, rem = mm - n * HWq;
        const int hq = rem / a.Wq, wq = rem - hq * a.Wq;
        const int hi0 = hq * a.in_step, wi0 = wq * a.in_step;
        x_ptr[i] = in + ((size_t)(n * a.Hi + hi0) * a.Wi + wi0) * (size_t)a.in_pstride + swz_chunk<CPR>(xrow, lcp) * 8;
        unsigned long long mk = 0;
        if (ok)
            for (int t = 0; t < ntaps; ++t) {
                const int16_t tp = a.taps[tap0 + t];
                const int hi = hi0 + (int)(int8_t)(tp & 0xff), wi = wi0 + (KWR ? 0 : (int)(int8_t)((tp >> 8) & 0xff));
                if (((unsigned)hi < (unsigned)a.Hi) && ((unsigned)wi < (unsigned)a.Wi)) mk |= 1ull << t;
            }
        x_mask[i] = mk;
        w_ptr[i] = a.w_hi + (size_t)(tn * BN + s) * a.w_kstride + kbase + swz_chunk<CPR>(s, lcp) * 8;
    }
    // destination of instruction i inside a buffer (wave-uniform byte offset)
    auto x_dst = [&](int i) __attribute__((always_inline)) {
        const int s0 = (wave * 4 + i) * 8;
        return (KWR ? (s0 >> 7) * 130 + 1 + (s0 & 127) : s0) * 128;
    };

    // one DMA instruction of K tile u: pieces 0..3 = weights, 4..7 = activations
    auto stage_piece = [&](auto PIECE, int u, int tl, ptrdiff_t xdelta, size_t wk) __attribute__((always_inline)) {
        constexpr int pc = decltype(PIECE)::value;
        if constexpr (pc < 4) {
            char *dst = ws + (u & 1) * WB + (wave * 4 + pc) * 8 * 128;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(w_ptr[pc] + wk),
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        } else {
            constexpr int i = pc - 4;
            const int xb = KWR ? (((kt_begin + u) / 3) & 1) : (u & 1);
            char *dst = xs + xb * XB + x_dst(i);
            const bool ok = (x_mask[i] >> tl) & 1ull;
            const bf16_t *src = ok ? x_ptr[i] + xdelta : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };

    f32x16_t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    __syncthreads();     // tap tables + pad rows visible

    // scalars of the tile being staged
    auto tile_xdelta = [&](int u) __attribute__((always_inline)) { return (ptrdiff_t)tapd_lds[tap0 + min(tile_tap(u), ntaps - 1)] + tile_ch(u); };
    auto tile_wk = [&](int u) __attribute__((always_inline)) { return (size_t)tile_tap(u) * a.Ci + tile_ch(u); };
    auto tile_has_x = [&](int u) __attribute__((always_inline)) { return !KWR || ((kt_begin + u) % 3) == 0; };

    // ---- prologue: tiles 0 and 1
#define DL_STAGE_ALL(u)                                                                                  \
    do {                                                                                                 \
        const int tl_ = tile_tap(u);                                                                     \
        const ptrdiff_t xd_ = tile_xdelta(u);                                                            \
        const size_t wk_ = tile_wk(u);                                                                   \
        stage_piece(IC<0>{}, u, tl_, xd_, wk_); stage_piece(IC<1>{}, u, tl_, xd_, wk_);                  \
        stage_piece(IC<2>{}, u, tl_, xd_, wk_); stage_piece(IC<3>{}, u, tl_, xd_, wk_);                  \
        if (tile_has_x(u)) {                                                                             \
            stage_piece(IC<4>{}, u, tl_, xd_, wk_); stage_piece(IC<5>{}, u, tl_, xd_, wk_);              \
            stage_piece(IC<6>{}, u, tl_, xd_, wk_); stage_piece(IC<7>{}, u, tl_, xd_, wk_);              \
        }                                                                                                \
    } while (0)
    if (T > 0) DL_STAGE_ALL(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (T > 1) DL_STAGE_ALL(1);
    __syncthreads();

    // ---- fragment addressing (bytes): lane = (row lr of a 32-row block, K half h of a 16-wide sub-step)
    const int lr = lane & 31, lh = lane >> 5;
    const int aw = (wn * 64 + lr) * 128 + ((lh ^ ((lr >> 1) & 7)) << 4);             // weight fragment, sub-step 0, block 0
    int ax[KWR ? 3 : 1];                                                            // activation fragment per kw shift
#pragma unroll
    for (int sh = 0; sh < (KWR ? 3 : 1); ++sh) {
        const int row = KWR ? wm * 130 + sh + lr : wm * 128 + lr;
        ax[sh] = row * 128 + ((lh ^ ((row >> 1) & 7)) << 4);
    }

    bf16x8_t F[2][6];
    if constexpr (ABL >= 2) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int f = 0; f < 6; ++f) F[q][f] = bf16x8_t{(short)(0x3f80 + lane), (short)(0x3f00 + f), 0x3e80, 0x3f81, (short)(0xbf80 + q), 0x3f10, 0x3e90, 0x3f91};
    }
    // fragments of sub-step s of the tile held in buffers (xbuf, wbuf), activation rows shifted by `axs`
    auto read_frags = [&](const char *xbuf, const char *wbuf, int axs, auto S, bf16x8_t (&f)[6]) __attribute__((always_inline)) {
        if (ABL >= 2) return;
        constexpr int sx = decltype(S)::value << 5;
        const char *wp = wbuf + (aw ^ sx);
        const char *xp = xbuf + (axs ^ sx);
        f[0] = *reinterpret_cast<const bf16x8_t *>(wp);
        f[1] = *reinterpret_cast<const bf16x8_t *>(wp + 32 * 128);
#pragma unroll
        for (int j = 0; j < 4; ++j) f[2 + j] = *reinterpret_cast<const bf16x8_t *>(xp + j * 32 * 128);
    };
    auto mma = [&](const bf16x8_t (&f)[6]) __attribute__((always_inline)) {
        if (ABL == 2) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = dl_mfma32(f[i], f[2 + j], acc[i][j]);
    };
    auto tile_bufs = [&](int u, const char *&xbuf, const char *&wbuf, int &axs) __attribute__((always_inline)) {
        wbuf = ws + (u & 1) * WB;
        if (KWR) {
            xbuf = xs + (((kt_begin + u) / 3) & 1) * XB;
            const int sh = tapw_lds[tap0 + min(tile_tap(u), ntaps - 1)];
            axs = sh == 0 ? ax[0] : (sh == 1 ? ax[KWR ? 1 : 0] : ax[KWR ? 2 : 0]);
        } else {
            xbuf = xs + (u & 1) * XB;
            axs = ax[0];
        }
    };

    const char *xb_cur, *wb_cur;
    int ax_cur;
    if (T > 0) {
        tile_bufs(0, xb_cur, wb_cur, ax_cur);
        read_frags(xb_cur, wb_cur, ax_cur, IC<0>{}, F[0]);
    }
    for (int t = 0; t < T; ++t) {
        const bool st1 = (t + 1 < T) && t > 0 && ABL != 1 && ABL != 3;          // tile t+1 is being staged (tile 1: by the prologue)
        const bool st2 = (t + 2 < T) && ABL != 1 && ABL != 3;
        int tl_n = 0; ptrdiff_t xd_n = 0; size_t wk_n = 0; bool hx_n = false;
        if (st1) { tl_n = tile_tap(t + 1); xd_n = tile_xdelta(t + 1); wk_n = tile_wk(t + 1); hx_n = tile_has_x(t + 1); }
        // ---- sub-step 0
        read_frags(xb_cur, wb_cur, ax_cur, IC<1>{}, F[1]);
        mma(F[0]);
        if (st1) {
            stage_piece(IC<3>{}, t + 1, tl_n, xd_n, wk_n);
            if (hx_n) { stage_piece(IC<4>{}, t + 1, tl_n, xd_n, wk_n); stage_piece(IC<5>{}, t + 1, tl_n, xd_n, wk_n); }
        }
        // ---- sub-step 1
        read_frags(xb_cur, wb_cur, ax_cur, IC<2>{}, F[0]);
        mma(F[1]);
        if (st1 && hx_n) { stage_piece(IC<6>{}, t + 1, tl_n, xd_n, wk_n); stage_piece(IC<7>{}, t + 1, tl_n, xd_n, wk_n); }
        // ---- sub-step 2
        read_frags(xb_cur, wb_cur, ax_cur, IC<3>{}, F[1]);
        mma(F[0]);
        // tile t is completely in registers / accumulators, tile t+1 has landed
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        DL_BAR();
        // ---- sub-step 3: first fragments of tile t+1, first DMA pieces of tile t+2 (into tile t's buffers)
        if (t + 1 < T) {
            tile_bufs(t + 1, xb_cur, wb_cur, ax_cur);
            read_frags(xb_cur, wb_cur, ax_cur, IC<0>{}, F[0]);
        }
        mma(F[1]);
        if (st2) {
            const int tl2 = tile_tap(t + 2); const ptrdiff_t xd2 = tile_xdelta(t + 2); const size_t wk2 = tile_wk(t + 2);
            stage_piece(IC<0>{}, t + 2, tl2, xd2, wk2); stage_piece(IC<1>{}, t + 2, tl2, xd2, wk2); stage_piece(IC<2>{}, t + 2, tl2, xd2, wk2);
        }
    }
    __syncthreads();

    // ---- epilogue: acc[i][j][r] = out[pixel = wm*128 + j*32 + (lane & 31)][channel = wn*64 + i*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)]
    const int oh = a.phase_oh[phase], ow = a.phase_ow[phase];
    const bool want_stats = a.stats_part != nullptr;
    float s1[2][4][4], s2[2][4][4];          // [i][quad][e]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) s1[i][q][e] = s2[i][q][e] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = tm * BM + wm * 128 + j * 32 + lr;
        if (m >= a.Mtot) continue;
        const int n = m / HWq, rem = m - n * HWq;
        const int hq = rem / a.Wq, wq = rem - hq * a.Wq;
        // sub-pixel phases of an odd-sized output (stride-2 data gradient of a 25-row input: phase grid 13 x .., 12 odd rows):
        // the phase grid is ceil(Ho / out_step) and outputs that fall outside the tensor are dropped
        if (hq * a.out_step + oh >= a.Ho || wq * a.out_step + ow >= a.Wo) continue;
        const size_t opix = ((size_t)n * a.Ho + (hq * a.out_step + oh)) * a.Wo + (wq * a.out_step + ow);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = tn * BN + wn * 64 + i * 32 + q * 8 + lh * 4;
                if (co >= a.Co) continue;
                float v[4] = {acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
                if (a.splitk > 1 || a.raw_out) {
                    float *dst = a.slab + ((size_t)ks * ((size_t)a.N * a.Ho * a.Wo) + opix) * a.Co + co;
                    *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    if (a.bias) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (co + e < a.bias_n) ? a.bias[co + e] : 0.f;
                    }
                    if (a.act != DL_ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = apply_act(a.act, v[e]);
                    }
                    bf16_t *dst = reinterpret_cast<bf16_t *>(a.out) + opix * a.out_pstride + co;
                    u32x2_t pk;
                    pk[0] = pack2_bf16(v[0], v[1]);
                    pk[1] = pack2_bf16(v[2], v[3]);
                    *reinterpret_cast<u32x2_t *>(dst) = pk;
                    if (want_stats) {
                        const float q0 = h16_lo_f32(pk[0]), q1 = h16_hi_f32(pk[0]);
                        const float q2 = h16_lo_f32(pk[1]), q3 = h16_hi_f32(pk[1]);
                        s1[i][q][0] += q0; s2[i][q][0] += q0 * q0; s1[i][q][1] += q1; s2[i][q][1] += q1 * q1;
                        s1[i][q][2] += q2; s2[i][q][2] += q2 * q2; s1[i][q][3] += q3; s2[i][q][3] += q3 * q3;
                    }
                }
            }
    }
    if (want_stats) {
        // the 32 lanes with equal lane>>5 hold 32 pixels of the same channels: butterfly over lane bits 0..4
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) { s1[i][q][e] += __shfl_xor(s1[i][q][e], o, 64); s2[i][q][e] += __shfl_xor(s2[i][q][e], o, 64); }
                }
        float *red = reinterpret_cast<float *>(smem_raw);          // [wm][2][BN]
        if (lr == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = wn * 64 + i * 32 + q * 8 + lh * 4 + e;
                        red[(wm * 2 + 0) * BN + c] = s1[i][q][e];
                        red[(wm * 2 + 1) * BN + c] = s2[i][q][e];
                    }
        }
        __syncthreads();
        const int m0 = tm * BM;
        const int n = m0 / HWq;
        const int chunk = ((m0 - n * HWq) / BM) * a.n_phase + phase;
        for (int c = tid; c < BN; c += 512) {
            const int co = tn * BN + c;
            if (co < a.Co) {
                float *o = a.stats_part + ((size_t)(n * a.stats_nchunks + chunk) * 2) * a.Co + co;
                o[0] = red[0 * BN + c] + red[2 * BN + c];
                o[a.Co] = red[1 * BN + c] + red[3 * BN + c];
            }
        }
    }
}

template <bool KWR, int ABL>
static int launch_conv_p32(const ConvArgs &a0, hipStream_t stream) {
    ConvArgs a = a0;
    a.tiles_m = (a.Mtot + 255) / 256;
    a.tiles_n = a.Co / 256;
    constexpr size_t smem = (size_t)2 * (KWR ? 260 : 256) * 128 + (size_t)2 * 256 * 128 + 2 * DL_MAX_TAPS * sizeof(int);
    auto kern = conv_gemm_p32_kernel<KWR, ABL>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) DL_FAIL("dl_conv_forward: hipFuncSetAttribute(%zu): %s", smem, hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(a.tiles_m * a.tiles_n, a.n_phase * a.splitk);
    hipLaunchKernelGGL(kern, grid, dim3(512), smem, stream, a);
    DL_CHECK_LAUNCH("dl_conv_forward(p32)");
    return 0;
}

// kernel-column reuse applies to: one phase of 9 taps ordered (kh, kw) with dh constant per kh and dw = -1/0/+1 (either
// direction), stride 1, image rows exactly 128 pixels wide (a 256-pixel tile = two whole rows), no split-K
static bool kwr_eligible(const ConvArgs &a) {
    if (a.n_phase != 1 || a.splitk != 1 || a.in_step != 1 || a.out_step != 1 || a.Wq != 128 || a.Wi != 128 || (a.Hq & 1)) return false;
    if (a.phase_tap_begin[1] - a.phase_tap_begin[0] != 9 || a.Ci < 64) return false;
    for (int kh = 0; kh < 3; ++kh) {
        const int dh0 = (int8_t)(a.taps[kh * 3] & 0xff);
        int seen = 0;
        for (int kw = 0; kw < 3; ++kw) {
            const int16_t tp = a.taps[kh * 3 + kw];
            const int dh = (int8_t)(tp & 0xff), dw = (int8_t)((tp >> 8) & 0xff);
            if (dh != dh0 || dw < -1 || dw > 1) return false;
            seen |= 1 << (dw + 1);
        }
        if (seen != 7) return false;
    }
    return true;
}

template <int ABL>
static int dispatch_p32(const ConvArgs &a, hipStream_t stream) {
    static const char *kwr_env = DL_DEV_ENV("DL_CONV_KWR");          // "0": stage every K tile's activations separately
    if (!(kwr_env && kwr_env[0] == '0') && kwr_eligible(a)) return launch_conv_p32<true, ABL>(a, stream);
    return launch_conv_p32<false, ABL>(a, stream);
}

template <int BM, int BN, int BK, int WM, int WN, bool UTAP, bool STAG = false, int ABL = 0>
static int launch_conv_glds_impl(const ConvArgs &a0, hipStream_t stream) {
    ConvArgs a = a0;
    a.tiles_m = (a.Mtot + BM - 1) / BM;
    a.tiles_n = (a.Co + BN - 1) / BN;
    constexpr size_t smem_loop = (size_t)2 * (BM + BN) * BK * sizeof(bf16_t) + DL_MAX_TAPS * (sizeof(int16_t) + sizeof(int));
    constexpr size_t smem = (BN >= 64 && epilogue_lds_bytes<BM, BN, WM>() > smem_loop) ? epilogue_lds_bytes<BM, BN, WM>() : smem_loop;
    auto kern = conv_gemm_glds_kernel<BM, BN, BK, WM, WN, UTAP, STAG, ABL>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) DL_FAIL("dl_conv_forward: hipFuncSetAttribute(%zu): %s", smem, hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(a.tiles_m * a.tiles_n, a.n_phase * a.splitk);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, stream, a);
    DL_CHECK_LAUNCH("dl_conv_forward(glds)");
    return 0;
}

template <int BM, int BN, int BK, int WM, int WN>
static int launch_conv_glds(const ConvArgs &a, hipStream_t stream) {
    // UTAP: every BK-wide K step lies inside one tap (Cin >= BK) and padding is zero -> scalar tap decode
    if (a.Ci >= BK && a.pad_mode == DL_PAD_ZERO) return launch_conv_glds_impl<BM, BN, BK, WM, WN, true>(a, stream);
    return launch_conv_glds_impl<BM, BN, BK, WM, WN, false>(a, stream);
}

// 256 x 256 tiles are used when they (x sub-pixel phases x split-K partials) give every CU a workgroup; 224 rather than 256 so that a
// 31 x 2-tile layer split 4 ways (PatchGAN 512->512 at 31 x 31) still qualifies.  geometry.py: conv_tile / choose_splitk mirror this.
static bool big_tile_fills_gpu(int mtot, int Co, int n_phase, int splitk) {
    return Co >= 256 && (Co % 256) == 0 && (size_t)((mtot + 255) / 256) * (Co / 256) * n_phase * splitk >= 224;
}

// conv_gemm_w4_kernel (conv_w4.hip) serves the ResnetBlock shape by default since r04; DL_CONV_W4=0 puts it back on the 8-phase kernel (same-box
// A/B, profiles/r04/w4_v2_variants.txt: 158.9 -> 139.8 us per launch inside the training step, 103.9 -> 100.5 ms per step)
static bool w4_enabled() {
    static const char *w4 = DL_DEV_ENV("DL_CONV_W4");
    return !(w4 && w4[0] == '0');
}

static int dispatch_tile_glds(const ConvArgs &a, hipStream_t stream) {
    if (a.Co <= 16) return launch_conv_glds<256, 16, 32, 4, 1>(a, stream);
    if (a.Co <= 64) return launch_conv_glds<128, 64, 64, 2, 2>(a, stream);
    static const bool no_big = DL_DEV_ENV("DL_NO_BIGTILE") != nullptr;
    // 256x256x64, 8 waves (each 128 pixels x 64 channels): twice the FLOP per staged byte of the 128x128 tile; needs
    // enough tiles to fill 256 CUs
    if (!no_big && big_tile_fills_gpu(a.Mtot, a.Co, a.n_phase, a.splitk))
    {
        // "1": software-pipelined 32x32x16 kernel (+ kernel-column reuse).  Same-box A/B of the whole training step (r01,
        // tools/ab_bench.sh): 121.7-122.0 ms vs 121.4-122.1 ms for the 8-phase kernel, 123.4-123.6 ms for the one-barrier kernel;
        // launch time 175 vs 168 vs 183 us.  The step is power-capped (profiles/r01/clock_probe.txt), so the 8-phase kernel
        // stays the default and this one is kept for the next tuning pass (it moves 33% fewer bytes into LDS).
        // "1": 4 waves x (128 px x 128 ch) on v_mfma_32x32x16, kernel-column reuse, one barrier per K step (conv_w4.hip) for the ResnetBlock shape
        if (w4_enabled() && w4_eligible(a)) return launch_conv_w4(a, stream);
#ifdef DL_DEV_SWITCHES       // older kernels kept for same-box A/Bs and the timing-only ablations (results WRONG by construction): dev build only
        static const char *p32 = DL_DEV_ENV("DL_CONV_P32");
        if (p32 && p32[0] == '1' && a.Ci >= 64 && a.pad_mode == DL_PAD_ZERO && !a.k_order) {
            static const char *ablp = DL_DEV_ENV("DL_CONV_ABLATE");
            if (ablp && ablp[0] == '1') return dispatch_p32<1>(a, stream);
            if (ablp && ablp[0] == '2') return dispatch_p32<2>(a, stream);
            if (ablp && ablp[0] == '3') return dispatch_p32<3>(a, stream);
            return dispatch_p32<0>(a, stream);
        }
        static const char *ph8 = DL_DEV_ENV("DL_CONV_8PH");            // "0": fall back to the one-barrier-per-step kernel
        if (!(ph8 && ph8[0] == '0') && a.Ci >= 64 && a.pad_mode == DL_PAD_ZERO && !a.k_order) {
            static const char *abl8 = DL_DEV_ENV("DL_CONV_ABLATE");
            if (abl8 && abl8[0] == '1') return launch_conv_8ph<1>(a, stream);
            if (abl8 && abl8[0] == '2') return launch_conv_8ph<2>(a, stream);
            if (abl8 && abl8[0] == '3') return launch_conv_8ph<3>(a, stream);
            if (abl8 && abl8[0] == '4') return launch_conv_8ph<4>(a, stream);
            return launch_conv_8ph<0>(a, stream);
        }
        static const bool stag = DL_DEV_ENV("DL_CONV_STAGGER") != nullptr;
        static const char *abl = DL_DEV_ENV("DL_CONV_ABLATE");       // "1": no DMA in the loop, "2": no LDS reads / MFMAs (timing only!)
        if (abl && abl[0] == '1' && a.Ci >= 64 && a.pad_mode == DL_PAD_ZERO) return launch_conv_glds_impl<256, 256, 64, 2, 4, true, false, 1>(a, stream);
        if (abl && abl[0] == '2' && a.Ci >= 64 && a.pad_mode == DL_PAD_ZERO) return launch_conv_glds_impl<256, 256, 64, 2, 4, true, false, 2>(a, stream);
        if (abl && abl[0] == '4' && a.Ci >= 64 && a.pad_mode == DL_PAD_ZERO) return launch_conv_glds_impl<256, 256, 64, 2, 4, true, false, 4>(a, stream);
        if (abl && abl[0] == '5' && a.Ci >= 64 && a.pad_mode == DL_PAD_ZERO) return launch_conv_glds_impl<256, 256, 64, 2, 4, true, false, 5>(a, stream);
        if (stag && a.Ci >= 64 && a.pad_mode == DL_PAD_ZERO) return launch_conv_glds_impl<256, 256, 64, 2, 4, true, true>(a, stream);
#else
        if (a.Ci >= 64 && a.pad_mode == DL_PAD_ZERO && !a.k_order) return launch_conv_8ph<0>(a, stream);
#endif
        return launch_conv_glds<256, 256, 64, 2, 4>(a, stream);
    }
    // (r06 experiment, dev build: a 256 x 128 tile -- 8 waves of 64 px x 64 ch, 85 flop per staged byte instead of 64 -- on the PatchGAN's c2 / c3 layers:
    // 84.3 -> 82.6 us, 64.8 -> 63.9 us: the 128 x 128 tile is not bound by the bytes it stages; removed)
    return launch_conv_glds<128, 128, 64, 2, 2>(a, stream);
}

// out[p][c] = act(bias[c] + sum_ks slab[ks][p][c]), fixed summation order (deterministic)
template <typename TOut>
__global__ void __launch_bounds__(256) conv_slab_reduce_kernel(const float *slab, int splitk, size_t npix, int Co,
                                                               const float *bias, int bias_n, int act, TOut *out, int out_pstride) {
    const size_t total = npix * (size_t)(Co / 4);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / (Co / 4);
        const int c = (int)(i % (Co / 4)) * 4;
        float4 s = make_float4(0, 0, 0, 0);
        for (int k = 0; k < splitk; ++k) {
            const float4 v = *reinterpret_cast<const float4 *>(slab + ((size_t)k * npix + p) * Co + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (bias) {
            s.x += (c < bias_n) ? bias[c] : 0.f; s.y += (c + 1 < bias_n) ? bias[c + 1] : 0.f;
            s.z += (c + 2 < bias_n) ? bias[c + 2] : 0.f; s.w += (c + 3 < bias_n) ? bias[c + 3] : 0.f;
        }
        s.x = apply_act(act, s.x); s.y = apply_act(act, s.y); s.z = apply_act(act, s.z); s.w = apply_act(act, s.w);
        TOut *dst = out + p * out_pstride + c;
        store1<TOut>(dst, s.x); store1<TOut>(dst + 1, s.y); store1<TOut>(dst + 2, s.z); store1<TOut>(dst + 3, s.w);
    }
}

#include "conv_c4.h"
#include "conv_x3.h"

// ------------------------------------------------------------------------------------------------ host dispatch
template <typename TIn, typename TOut, int PREC, int BM, int BN, int BK, int WM, int WN>
static int launch_conv(const ConvArgs &a0, hipStream_t stream) {
    ConvArgs a = a0;
    a.tiles_m = (a.Mtot + BM - 1) / BM;
    a.tiles_n = (a.Co + BN - 1) / BN;
    constexpr int NPL = (PREC == 3) ? 2 : 1;
    constexpr size_t smem = (size_t)2 * NPL * (BM + BN) * BK * sizeof(bf16_t) + DL_MAX_TAPS * sizeof(int16_t);
    auto kern = conv_gemm_kernel<TIn, TOut, PREC, BM, BN, BK, WM, WN>;
    static bool attr_set = false;      // benign race: idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) DL_FAIL("dl_conv_forward: hipFuncSetAttribute(%zu): %s", smem, hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(a.tiles_m * a.tiles_n, a.n_phase * a.splitk);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, a);
    DL_CHECK_LAUNCH("dl_conv_forward");
    return 0;
}

template <typename TIn, typename TOut, int PREC>
static int dispatch_tile(const ConvArgs &a, hipStream_t stream) {
    constexpr int BK = (PREC == 3) ? 32 : 64;
    if (a.Co <= 16) return launch_conv<TIn, TOut, PREC, 256, 16, 32, 4, 1>(a, stream);
    if (a.Co <= 64) return launch_conv<TIn, TOut, PREC, 128, 64, BK, 2, 2>(a, stream);
    return launch_conv<TIn, TOut, PREC, 128, 128, BK, 2, 2>(a, stream);
}

// descriptor -> the geometry fields of the kernel argument block (everything but pointers and per-call switches)
static void fill_conv_geometry(ConvArgs &a, const dl_conv_desc *d) {
    a.N = d->N; a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.log2Ci = ilog2_exact(d->Ci); a.in_pstride = d->in_pstride;
    a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co; a.out_pstride = d->out_pstride;
    a.Hq = d->Hq; a.Wq = d->Wq; a.out_step = d->out_step; a.in_step = d->in_step;
    a.n_phase = d->n_phase; a.splitk = d->splitk;
    for (int p = 0; p < DL_MAX_PHASES; ++p) { a.phase_oh[p] = d->phase_oh[p]; a.phase_ow[p] = d->phase_ow[p]; a.phase_kbase[p] = d->phase_kbase[p]; }
    for (int p = 0; p <= DL_MAX_PHASES; ++p) a.phase_tap_begin[p] = d->phase_tap_begin[p];
    for (int t = 0; t < DL_MAX_TAPS; ++t) a.taps[t] = (int16_t)(((uint16_t)(uint8_t)d->tap_dh[t]) | ((uint16_t)(uint8_t)d->tap_dw[t] << 8));
    a.pad_mode = d->pad_mode; a.w_kstride = d->w_kstride; a.act = d->act; a.in_act = d->in_act; a.bias_n = d->bias_n; a.raw_out = d->raw_out;
    a.in_split = d->in_split;
    a.Mtot = d->N * d->Hq * d->Wq;
}

// does dl_conv_forward send this descriptor to the fused four-phase kernel (conv_s2f.hip)?  DL_CONV_S2F=0: keep the 4-phase gather GEMM (A/B)
static bool s2f_applies(const dl_conv_desc *d) {
    const char *env = dl_switch(DL_SW_CONV_S2F);
    static const bool no_glds = DL_DEV_ENV("DL_NO_GLDS") != nullptr;
    if ((env && env[0] == '0') || no_glds || d->in_dtype != DL_BF16 || d->prec != DL_PREC_BF16 || d->in_act != DL_ACT_NONE || d->n_phase != 4) return false;
    static const bool epi_old = DL_DEV_ENV("DL_OLD_EPILOGUE") != nullptr;
    if (epi_old) return false;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    fill_conv_geometry(a, d);
    return s2f_eligible(a);
}

// does dl_conv_forward send this descriptor to the one-channel dot-product kernels (conv_dot.hip: the PatchGAN's prediction layer)?  0 = no, 1 = forward,
// 2 = data gradient.  DL_CONV_DOT=0: keep the gather GEMM (A/B)
static int dot_applies(const dl_conv_desc *d) {
    const char *env = dl_switch(DL_SW_CONV_DOT);
    if ((env && env[0] == '0') || d->in_dtype != DL_BF16 || d->prec != DL_PREC_BF16 || d->in_act != DL_ACT_NONE || d->n_phase != 1 || d->raw_out) return 0;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    fill_conv_geometry(a, d);
    a.bias = d->bias_n == 1 ? reinterpret_cast<const float *>(&a) : nullptr;        // (only its presence is tested)
    if (dot_fwd_eligible(a)) return 1;
    a.bias = nullptr;
    if (d->ci_real == 1 && d->bias_n == 0 && dot_dgrad_eligible(a)) return 2;
    return 0;
}

// ... or to the PatchGAN's first-layer kernel (conv_d1.hip: 8 contracted channels, k4 s2, 256-pixel output rows)?  DL_CONV_DOT=0 switches it off too (A/B)
static bool d1_applies(const dl_conv_desc *d) {
    const char *env = dl_switch(DL_SW_CONV_DOT);
    static const bool epi_old = DL_DEV_ENV("DL_OLD_EPILOGUE") != nullptr;
    if ((env && env[0] == '0') || epi_old || d->in_dtype != DL_BF16 || d->prec != DL_PREC_BF16 || d->in_act != DL_ACT_NONE || d->n_phase != 1 || d->in_step != 2 || d->Ci != 8 ||
        d->splitk != 1)
        return false;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    fill_conv_geometry(a, d);
    return d1_eligible(a);
}

// ... or to that layer's data-gradient kernel (conv_d1g.hip: four phases as the M dimension, 64 contracted and 8 output channels)?  DL_CONV_DOT=0: off (A/B)
static bool d1g_applies(const dl_conv_desc *d) {
    const char *env = dl_switch(DL_SW_CONV_DOT);
    if ((env && env[0] == '0') || d->in_dtype != DL_BF16 || d->prec != DL_PREC_BF16 || d->in_act != DL_ACT_NONE || d->n_phase != 4 || d->Ci != 64 || d->Co != 8 || d->splitk != 1 ||
        d->bias_n != 0)
        return false;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    fill_conv_geometry(a, d);
    return d1g_eligible(a);
}

// does dl_conv_forward send this descriptor to the register-stationary stride-2 kernel (conv_s2d.hip: down1 forward, up2 data gradient)?
// DL_CONV_S2D=0: keep the gather GEMM (A/B)
static bool s2d_applies(const dl_conv_desc *d) {
    const char *env = dl_switch(DL_SW_CONV_S2D);
    static const bool no_glds = DL_DEV_ENV("DL_NO_GLDS") != nullptr;
    static const bool epi_old = DL_DEV_ENV("DL_OLD_EPILOGUE") != nullptr;
    if ((env && env[0] == '0') || no_glds || epi_old || d->in_dtype != DL_BF16 || d->prec != DL_PREC_BF16 || d->in_act != DL_ACT_NONE || d->n_phase != 1 || d->in_step != 2)
        return false;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    fill_conv_geometry(a, d);
    return s2d_eligible(a);
}

// ... or to its mirror image for the UP direction (conv_s2u.hip: up2 forward, down1 data gradient)?  DL_CONV_S2D=0 switches both off (A/B)
static bool s2u_applies(const dl_conv_desc *d) {
    const char *env = dl_switch(DL_SW_CONV_S2D);
    static const bool no_glds = DL_DEV_ENV("DL_NO_GLDS") != nullptr;
    static const bool epi_old = DL_DEV_ENV("DL_OLD_EPILOGUE") != nullptr;
    if ((env && env[0] == '0') || no_glds || epi_old || d->in_dtype != DL_BF16 || d->prec != DL_PREC_BF16 || d->in_act != DL_ACT_NONE || d->n_phase != 4 || d->Ci != 128)
        return false;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    fill_conv_geometry(a, d);
    return s2u_eligible(a);
}

// ... and to its strict twin (conv_s2f_x3.hip)?  Needs the split copy of the input; DL_CONV_S2F=0 or DL_CONV_S2FX3=0: keep the 4-phase strict kernel (A/B)
static bool s2fx3_applies(const dl_conv_desc *d) {
    const char *env = dl_switch(DL_SW_CONV_S2F);
    const char *env3 = dl_switch(DL_SW_CONV_S2FX3);
    if ((env && env[0] == '0') || (env3 && env3[0] == '0') || d->in_dtype != DL_F32 || d->prec != DL_PREC_BF16X3 || d->in_act != DL_ACT_NONE || d->n_phase != 4 ||
        !d->in_split || !x3_glds_applies(d))
        return false;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    fill_conv_geometry(a, d);
    return s2f_x3_eligible(a);
}

// tile height (pixels) the dispatch picks for the bf16 direct-to-LDS path; 0 when that path is not taken
static int glds_tile_bm(const dl_conv_desc *d) {
    static const bool no_glds = DL_DEV_ENV("DL_NO_GLDS") != nullptr;
    if (no_glds || d->in_dtype != DL_BF16 || d->prec != DL_PREC_BF16 || d->in_act != DL_ACT_NONE) return 0;
    if (d->Co <= 16) return 256;
    if (d->Co <= 64) return 128;
    static const bool no_big = DL_DEV_ENV("DL_NO_BIGTILE") != nullptr;
    const int mtot = d->N * d->Hq * d->Wq;
    if (!no_big && big_tile_fills_gpu(mtot, d->Co, d->n_phase, d->splitk)) return 256;
    return 128;
}

// Name of the kernel dl_conv_forward launches for this descriptor (as rocprofv3 prints it, without template noise); lets the
// benchmark label its roofline line from the dispatch itself instead of a string that goes stale when a default changes.
extern "C" const char *dl_conv_kernel_name(const dl_conv_desc *d) {
    if (!d) return "(null)";
    if (c4_bf16_eligible(d)) return "conv_c4_patch_kernel";
    if (c4_x3_eligible(d)) return "conv_c4_patch_x3_kernel";
    if (const int k = dot_applies(d)) return k == 1 ? "conv_dot_fwd_kernel" : "conv_dot_dgrad_kernel";
    if (d1_applies(d)) return "conv_d1_kernel";
    if (d1g_applies(d)) return "conv_d1g_kernel";
    if (s2u_applies(d)) return "conv_s2u_kernel";
    if (s2f_applies(d)) return "conv_s2f_kernel";
    if (s2d_applies(d)) return "conv_s2d_kernel";
    if (s2fx3_applies(d)) return "conv_s2f_x3_kernel";
    const int bm = glds_tile_bm(d);
    if (bm == 0 && x3_glds_applies(d)) {
        if (w4x3_enabled()) {
            ConvArgs a;
            memset(&a, 0, sizeof(a));
            fill_conv_geometry(a, d);
            a.w_lo = reinterpret_cast<const bf16_t *>(&a);          // (only its presence is tested)
            if (w4x3_eligible(a)) return "conv_gemm_w4x3_kernel";
        }
        return x3_kernel_name(d);
    }
    if (bm == 0) return (d->in_dtype == DL_BF16) ? "conv_gemm_kernel<bf16>" : "conv_gemm_kernel<f32>";
    if (d->Co <= 16) return "conv_gemm_glds_kernel<256,16,32>";
    if (d->Co <= 64) return "conv_gemm_glds_kernel<128,64,64>";
    if (bm != 256) return "conv_gemm_glds_kernel<128,128,64>";
    if (w4_enabled()) {
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        fill_conv_geometry(a, d);
        if (w4_eligible(a)) return "conv_gemm_w4_kernel";
    }
    const bool utap = d->Ci >= 64 && d->pad_mode == DL_PAD_ZERO;
    static const char *korder_env = DL_DEV_ENV("DL_CONV_KORDER");
    const bool korder = korder_env && korder_env[0] == '1';
    static const char *p32 = DL_DEV_ENV("DL_CONV_P32");
    static const char *ph8 = DL_DEV_ENV("DL_CONV_8PH");
    if (utap && !korder && p32 && p32[0] == '1') return "conv_gemm_p32_kernel";
    if (utap && !korder && !(ph8 && ph8[0] == '0')) return "conv_gemm_8ph_kernel";
    return "conv_gemm_glds_kernel<256,256,64>";
}

extern "C" int dl_conv_stats_chunks(const dl_conv_desc *d) {
    if (!d || d->splitk != 1 || d->raw_out || d->act != DL_ACT_NONE) return 0;
    if (dot_applies(d) || d1_applies(d) || d1g_applies(d)) return 0;
    if (c4_eligible(d)) return (d->Ho / 4) * (d->Wo / 64);          // one chunk per 4 x 64 tile
    if (s2u_applies(d)) {                                            // one chunk per workgroup (row segment x strip of input rows)
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        fill_conv_geometry(a, d);
        return s2u_stats_chunks(a);
    }
    if (s2f_applies(d) || s2fx3_applies(d)) {                        // one chunk per 256-pixel tile of the phase grid (all four phases summed)
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        fill_conv_geometry(a, d);
        return s2f_stats_chunks(a);
    }
    if (s2d_applies(d)) {                                            // one chunk per workgroup (row segment x strip of output rows)
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        fill_conv_geometry(a, d);
        return s2d_stats_chunks(a);
    }
    static const bool no_x3_stats = DL_DEV_ENV("DL_NO_X3_STATS") != nullptr;       // A/B: strict policy with the stand-alone statistics pass
    const int bm = (x3_glds_applies(d) && !no_x3_stats) ? x3_tile_bm(d) : glds_tile_bm(d);
    const int hw = d->Hq * d->Wq;
    if (bm == 0 || hw % bm) return 0;
    return (hw / bm) * d->n_phase;
}

extern "C" int dl_conv_bnstats_chunks(const dl_conv_desc *d) {
    // the reductions live in the LDS-transposed store epilogue of the direct-to-LDS kernels (tile_epilogue_lds): bf16, >= 64 output
    // channels per tile, no split-K, tiles that do not straddle images
    static const bool off = DL_DEV_ENV("DL_OLD_EPILOGUE") != nullptr;
    static const char *p32 = DL_DEV_ENV("DL_CONV_P32");
    if (!d || off || (p32 && p32[0] == '1') || d->Co <= 16 || d->act != DL_ACT_NONE || c4_eligible(d) || s2f_applies(d) || s2fx3_applies(d) || s2d_applies(d) || s2u_applies(d)) return 0;
    static const int min_bm = DL_DEV_ENV("DL_BNSTATS_MIN_BM") ? atoi(DL_DEV_ENV("DL_BNSTATS_MIN_BM")) : 0;       // A/B: 256 = only the 256 x 256-tile kernels
    if (glds_tile_bm(d) < min_bm) return 0;
    return dl_conv_stats_chunks(d);
}

static int conv_forward_impl(const dl_conv_desc *d, const void *in, const void *w_hi, const void *w_lo, const float *bias,
                             void *out, float *slab, float *stats_part, const dl_conv_bnstats *bn, void *stream_, const void *add = nullptr, int add_pstride = 0);

extern "C" int dl_conv_forward(const dl_conv_desc *d, const void *in, const void *w_hi, const void *w_lo, const float *bias,
                               void *out, float *slab, float *stats_part, void *stream_) {
    return conv_forward_impl(d, in, w_hi, w_lo, bias, out, slab, stats_part, nullptr, stream_);
}

// out = conv(in) + addend in ONE pass (the addend is added in fp32 before the bf16 rounding of the store): the kernels that can do it in their store
// epilogue -- today conv_gemm_w4_kernel, the ResnetBlock shape, where the data gradient of the block's first conv meets the gradient that came down the skip
// connection (networks.py:509-513: out = x + conv_block(x)).  `addend` may be `out` itself (every thread reads its 16 bytes before it writes them).
extern "C" int dl_conv_add_supported(const dl_conv_desc *d) {
    return d && d->splitk == 1 && !d->raw_out && strcmp(dl_conv_kernel_name(d), "conv_gemm_w4_kernel") == 0;
}

extern "C" int dl_conv_forward_add(const dl_conv_desc *d, const void *in, const void *w_hi, const void *w_lo, const void *addend, int32_t addend_pstride,
                                   void *out, void *stream_) {
    if (!addend) DL_FAIL("dl_conv_forward_add: null addend");
    if (addend_pstride % 8) DL_FAIL("dl_conv_forward_add: addend_pstride must keep 16-byte alignment");
    if (!dl_conv_add_supported(d)) DL_FAIL("dl_conv_forward_add: not available for this descriptor (ask dl_conv_add_supported first)");
    return conv_forward_impl(d, in, w_hi, w_lo, nullptr, out, nullptr, nullptr, nullptr, stream_, addend, addend_pstride);
}

extern "C" int dl_conv_forward_bnstats(const dl_conv_desc *d, const void *in, const void *w_hi, const void *w_lo, void *out,
                                       float *stats_part, const dl_conv_bnstats *bn, void *stream_) {
    if (!d || !bn || !stats_part) DL_FAIL("dl_conv_forward_bnstats: null argument");
    if (!bn->y || !bn->mean || !bn->rstd || !bn->scale || !bn->shift) DL_FAIL("dl_conv_forward_bnstats: null statistics / y");
    if (bn->act != DL_ACT_NONE && bn->act != DL_ACT_RELU && bn->act != DL_ACT_LRELU) DL_FAIL("dl_conv_forward_bnstats: act=%d", bn->act);
    if (bn->y_pstride % 8) DL_FAIL("dl_conv_forward_bnstats: y_pstride must keep 16-byte alignment");
    if (dl_conv_bnstats_chunks(d) == 0) DL_FAIL("dl_conv_forward_bnstats: not available for this descriptor (ask dl_conv_bnstats_chunks first)");
    return conv_forward_impl(d, in, w_hi, w_lo, nullptr, out, nullptr, stats_part, bn, stream_);
}

static int conv_forward_impl(const dl_conv_desc *d, const void *in, const void *w_hi, const void *w_lo, const float *bias,
                             void *out, float *slab, float *stats_part, const dl_conv_bnstats *bn, void *stream_, const void *add, int add_pstride) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!d) DL_FAIL("dl_conv_forward: null descriptor");
    if (d->N <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Hq <= 0 || d->Wq <= 0)
        DL_FAIL("dl_conv_forward: empty problem (N=%d, in %dx%d, out %dx%d, phase grid %dx%d): nothing to launch", d->N, d->Hi, d->Wi, d->Ho,
                d->Wo, d->Hq, d->Wq);
    if (!in || !w_hi || (!out && !d->raw_out)) DL_FAIL("dl_conv_forward: null argument");
    const int l2 = ilog2_exact(d->Ci);
    if (l2 < 3) DL_FAIL("dl_conv_forward: Ci=%d must be a power of two >= 8", d->Ci);
    if (d->Co % 8 || d->Co <= 0) DL_FAIL("dl_conv_forward: Co=%d must be a positive multiple of 8", d->Co);
    if (d->in_pstride % 8 || d->out_pstride % 4) DL_FAIL("dl_conv_forward: pixel strides must keep 16-byte alignment");
    if (d->n_phase < 1 || d->n_phase > DL_MAX_PHASES) DL_FAIL("dl_conv_forward: n_phase=%d", d->n_phase);
    if (d->phase_tap_begin[d->n_phase] > DL_MAX_TAPS) DL_FAIL("dl_conv_forward: too many taps");
    if (d->prec == DL_PREC_BF16X3 && !w_lo) DL_FAIL("dl_conv_forward: BF16X3 needs the lo weight plane");
    if (d->prec == DL_PREC_BF16X3 && d->in_dtype != DL_F32) DL_FAIL("dl_conv_forward: BF16X3 needs fp32 activations");
    if (d->in_dtype != d->out_dtype) DL_FAIL("dl_conv_forward: in/out dtype must match");
    if (d->splitk < 1 || ((d->splitk > 1 || d->raw_out) && !slab)) DL_FAIL("dl_conv_forward: splitk=%d / raw_out needs a slab", d->splitk);
    if (d->raw_out && d->splitk != 1) DL_FAIL("dl_conv_forward: raw_out requires splitk == 1");
    if (d->pad_mode == DL_PAD_REFLECT && (d->n_phase != 1)) DL_FAIL("dl_conv_forward: reflect padding only for single-phase layers");
    if ((size_t)d->N * d->Hi * d->Wi * (size_t)d->in_pstride >= ((size_t)1 << 40)) DL_FAIL("dl_conv_forward: tensor too large");
    for (int p = 0; p < d->n_phase; ++p)
        if (d->phase_kbase[p] % 64) DL_FAIL("dl_conv_forward: phase_kbase must be a multiple of 64");

    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.w_hi = (const bf16_t *)w_hi; a.w_lo = (const bf16_t *)w_lo; a.bias = bias; a.out = out; a.slab = slab;
    fill_conv_geometry(a, d);
    if (d->in_split && !(d->in_dtype == DL_F32 && d->prec == DL_PREC_BF16X3 && d->in_act == DL_ACT_NONE && x3_glds_applies(d)))
        DL_FAIL("dl_conv_forward: in_split needs the strict policy (fp32 + BF16X3) on the direct-to-LDS kernels and no input activation");
    static const bool epi_old = DL_DEV_ENV("DL_OLD_EPILOGUE") != nullptr;
    a.epi_old = epi_old ? 1 : 0;
    a.Mtot = d->N * d->Hq * d->Wq;
    // A/B switch: "1" = channel-chunk-major K steps.  Measured on MI355X (r01): 4-7% faster in an isolated loop over one layer
    // (tools/ab_order.sh) but 3-12% SLOWER inside the training step (profiles/r01: 162.9 -> 168.7 us on the 256x256 tile),
    // so tap-major stays the default.
    static const char *korder_env = DL_DEV_ENV("DL_CONV_KORDER");
    a.k_order = (korder_env && korder_env[0] == '1') ? 1 : 0;
    a.stats_part = nullptr;
    a.stats_nchunks = 0;
    if (stats_part) {
        a.stats_nchunks = dl_conv_stats_chunks(d);
        if (a.stats_nchunks == 0) DL_FAIL("dl_conv_forward: fused statistics are not available for this descriptor (ask dl_conv_stats_chunks first)");
        a.stats_part = stats_part;
    }
    a.add = (const bf16_t *)add;
    a.add_pstride = add_pstride;
    if (bn) {
        a.bn_y = (const bf16_t *)bn->y; a.bn_y_pstride = bn->y_pstride; a.bn_act = bn->act;
        a.bn_mean = bn->mean; a.bn_rstd = bn->rstd; a.bn_scale = bn->scale; a.bn_shift = bn->shift;
    }

    int rc;
    static const bool no_glds = DL_DEV_ENV("DL_NO_GLDS") != nullptr;      // A/B switch for profiling the two staging paths
    if (!bn && !stats_part && !add) {
        if (const int k = dot_applies(d)) {
            if (k == 1 ? dot_fwd_eligible(a) : dot_dgrad_eligible(a)) return launch_conv_dot(a, k == 1, stream);     // (no split-K, whatever the descriptor says: nothing to reduce)
        }
    }
    if (c4_bf16_eligible(d)) rc = launch_conv_c4(a, d, stream);
    else if (!bn && !stats_part && !add && d1_applies(d)) rc = launch_conv_d1(a, stream);
    else if (!bn && !stats_part && !add && !bias && d1g_applies(d)) rc = launch_conv_d1g(a, stream);
    else if (c4_x3_eligible(d)) rc = launch_conv_c4_x3(a, d, stream);
    else if (!bn && s2u_applies(d)) rc = launch_conv_s2u(a, stream);
    else if (!bn && s2f_applies(d)) rc = launch_conv_s2f(a, stream);
    else if (!bn && s2fx3_applies(d)) rc = launch_conv_s2f_x3(a, stream);
    else if (!bn && s2d_applies(d)) rc = launch_conv_s2d(a, stream);
    else if (d->in_dtype == DL_BF16 && d->prec == DL_PREC_BF16 && d->in_act == DL_ACT_NONE && !no_glds) rc = dispatch_tile_glds(a, stream);
    else if (d->in_dtype == DL_BF16 && d->prec == DL_PREC_BF16) rc = dispatch_tile<bf16_t, bf16_t, 1>(a, stream);
    else if (d->in_dtype == DL_F32 && d->prec == DL_PREC_BF16X3) rc = x3_glds_applies(d) ? dispatch_tile_x3(a, stream) : dispatch_tile<float, float, 3>(a, stream);
    else if (d->in_dtype == DL_F32 && d->prec == DL_PREC_BF16) rc = dispatch_tile<float, float, 1>(a, stream);
    else DL_FAIL("dl_conv_forward: unsupported dtype/precision combination (%d, %d)", d->in_dtype, d->prec);
    if (rc) return rc;

    if (d->splitk > 1) {
        const size_t npix = (size_t)d->N * d->Ho * d->Wo;
        const size_t total = npix * (size_t)(d->Co / 4);
        const int blocks = (int)min((size_t)2048, (total + 255) / 256);
        if (d->out_dtype == DL_F32)
            hipLaunchKernelGGL(conv_slab_reduce_kernel<float>, dim3(blocks), dim3(256), 0, stream, slab, d->splitk, npix, d->Co,
                               bias, d->bias_n, d->act, (float *)out, d->out_pstride);
        else
            hipLaunchKernelGGL(conv_slab_reduce_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, slab, d->splitk, npix, d->Co,
                               bias, d->bias_n, d->act, (bf16_t *)out, d->out_pstride);
        DL_CHECK_LAUNCH("dl_conv_forward(reduce)");
    }
    return 0;
}
