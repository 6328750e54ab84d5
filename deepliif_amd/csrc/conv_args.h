// conv_args.h -- kernel argument block and LDS swizzle shared by the conv translation units (conv_gemm.hip, conv_w4.hip).
#pragma once
#include "common.h"

struct ConvArgs {
    const void *in;
    const bf16_t *w_hi;
    const bf16_t *w_lo;
    const float *bias;
    void *out;
    float *slab;
    int N, Hi, Wi, Ci, log2Ci, in_pstride;
    int Ho, Wo, Co, out_pstride;
    int Hq, Wq, out_step, in_step;
    int n_phase, splitk;
    int phase_oh[DL_MAX_PHASES], phase_ow[DL_MAX_PHASES];
    int phase_tap_begin[DL_MAX_PHASES + 1];
    int phase_kbase[DL_MAX_PHASES];
    int pad_mode, w_kstride, act, in_act, bias_n, raw_out;
    int in_split;           // strict kernels: the input is the producer-written split copy ([8 hi | 8 lo] per 8 channels): no in-kernel split
    int epi_old;            // DL_OLD_EPILOGUE=1: per-fragment stores instead of the LDS-transposed whole-row stores (A/B switch)
    int tiles_m, tiles_n, Mtot;
    int k_order;                    // direct-to-LDS UTAP path: 0 = K steps tap-major, 1 = channel-chunk-major (L2 reuse of the halo slab)
    int k_order8;                   // the same choice for the 8-phase kernels (bf16: DL_8PH_KORDER, strict: DL_X3_KORDER)
    float *stats_part;              // fused norm statistics: part[((n*nchunks + chunk)*2 + {sum,sumsq})*Co + c]
    int stats_nchunks;
    // fused norm-backward reductions (dl_conv_forward_bnstats): y tile read next to the dz tile in the store epilogue
    const bf16_t *bn_y;
    const float *bn_mean, *bn_rstd, *bn_scale, *bn_shift;
    int bn_y_pstride, bn_act;
    // dl_conv_forward_add: out = conv + addend (bf16, same pixels / channels as out); only the kernels dl_conv_add_supported() names read it
    const bf16_t *add;
    int add_pstride;
    int16_t taps[DL_MAX_TAPS];      // (dh & 0xff) | (dw << 8)
};

template <int CPR> __device__ __forceinline__ int swz_chunk(int row, int c) {
    if constexpr (CPR == 4) {
        // f(q) = {0,2,3,1}[q], q = (row >> 2) & 3
        const int q = (row >> 2) & 3;
        return c ^ ((0x1320 >> (4 * q)) & 3);
    } else {
        return c ^ ((row >> 1) & (CPR - 1));
    }
}

