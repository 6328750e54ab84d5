// conv_x3.h -- included by conv_gemm.hip.  The STRICT-PARITY policy (fp32 activations, split-bf16 x3 products: the only policy the GPU
// tests hold to 1e-3 against the fp32 reference, networks.py:357-664) on the direct-to-LDS kernels.
//
// Round 1/2 ran this policy on the register-staged conv_gemm_kernel<float, float, 3> (global -> VGPR -> split -> ds_write, 128 x 128 tiles,
// one barrier per K step): 526 us per ResnetBlock launch against 149 us for the bf16 8-phase kernel.  Here the operands go HBM -> LDS by
// global_load_lds exactly like the bf16 kernels, with the same 128-byte LDS rows:
//   * activations stay fp32 in HBM (no second storage format anywhere in the engine).  A K step covers 32 channels: one pixel row =
//     32 fp32 = 128 bytes = the bf16 kernel's 64 channels, so the DMA geometry (8 rows per wave-instruction, 16 B per lane) is unchanged.
//     A lane's MFMA B fragment (8 consecutive channels of one pixel) is two adjacent 16-byte chunks (2*fg, 2*fg + 1); the bf16 kernels'
//     chunk swizzle c ^ ((row >> 1) & 7) would put the rows 0-3 / 12-15 (chunk 2*fg) and 4-11 (chunk 2*fg + 2) of a ds_read_b128 lane group
//     on the same banks (2-way), so the activation tiles use c ^ f(row >> 1) with f = {0,2,1,3,5,7,4,6}: f maps the rows {0,1,6,7} of a
//     group onto even values and {2,3,4,5} onto odd ones, which keeps {f(h)} and {f(h) ^ 2} disjoint -- conflict-free under the lane-group
//     table of MI355X_MICROARCH.md (checked exhaustively in tests/test_geometry.py::test_x3_swizzle_is_conflict_free).
//     The hi / lo split (hi = bf16(x), lo = bf16(x - hi)) happens in REGISTERS after the fragment read: 3 VALU ops per element, issued in
//     the wave's LDS-read section where the partner wave of the SIMD is multiplying.
//   * weights: the packed hi and lo images (dl_pack_weights) are staged side by side: one 128-byte LDS row = [32 hi | 32 lo] bf16 of one
//     output channel, chunk c < 4 from the hi image, c >= 4 from the lo image (per-lane source pointers, nothing repacked).  The A
//     fragments are chunk fg (hi) and 4 + fg (lo): the same two reads, same swizzle and same bank pattern as kk = 0 / 1 of the bf16 kernel.
//   * per fragment pair three MFMAs into the same accumulator, small terms first: lo_w*hi_x, hi_w*lo_x, hi_w*hi_x (term-major over the
//     8 accumulators of a quadrant, so dependent MFMAs are 8 issues apart).
// Staged bytes per MFMA are 2/3 of the bf16 kernel's (same bytes per K step, 1.5x the MFMAs), so the K loop is less DMA-bound than the
// bf16 one; a launch costs about twice the bf16 launch (twice the K steps) instead of 3.5x.
#pragma once

__device__ __forceinline__ int x3_swz(int h) { return (((h >> 2) & 1) << 2) | ((h & 1) << 1) | (((h >> 1) ^ (h >> 2)) & 1); }

// 8 fp32 values (two 16-byte LDS reads: channels 8*fg .. 8*fg + 7 of one pixel) -> bf16 hi and lo MFMA fragments
template <int IN_ACT>
__device__ __forceinline__ void x3_split8(f32x4_t a, f32x4_t b, bf16x8_t &hi, bf16x8_t &lo) {
    if constexpr (IN_ACT == DL_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = fmaxf(a[i], 0.f); b[i] = fmaxf(b[i], 0.f); }
    } else if constexpr (IN_ACT == DL_ACT_LRELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = fmaxf(a[i], 0.2f * a[i]); b[i] = fmaxf(b[i], 0.2f * b[i]); }
    }
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

// 4 fp32 values -> 4 bf16 hi (8 bytes) + 4 bf16 lo: the in-LDS conversion of the staged tiles (each lane splits the 16 bytes its own DMA fetched)
template <int IN_ACT>
__device__ __forceinline__ void x3_split4(f32x4_t v, u32x2_t &hi, u32x2_t &lo) {
    if constexpr (IN_ACT == DL_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
    } else if constexpr (IN_ACT == DL_ACT_LRELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.2f * v[i]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t h = pack2_bf16(v[2 * i], v[2 * i + 1]);
        hi[i] = h;
        lo[i] = pack2_bf16(v[2 * i] - h16_lo_f32(h), v[2 * i + 1] - h16_hi_f32(h));
    }
}

// fp32 store epilogue of the strict kernels (accumulator layout of mfma_f32_16x16x32: lane = (pixel fr, channels 4*fg .. 4*fg + 3)):
// one 16-byte store per fragment -- the four lanes fg = 0..3 of a pixel write 64 contiguous bytes -- or raw fp32 slabs for split-K / raw_out
// + the optional fused per-(image, channel) statistics of the stored values for the normalisation that follows (dl_conv_stats_chunks protocol, as the
// bf16 kernels: one chunk per tile and phase; the host guarantees that a tile lies in ONE image): sums over the lane's pixels, four DPP adds
// over the 16 lanes of a row, the WM wave rows through LDS (dead after the K loop)
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void x3_epilogue(const ConvArgs &a, f32x4_t (&acc)[BN / WN / 16][BM / WM / 16], int tm, int tn, int phase, int ks,
                                            int wm, int wn, int lane, int tid, char *smem_raw) {
    constexpr int PM = BM / WM, PN = BN / WN, FM = PM / 16, FN = PN / 16;
    const int fr = lane & 15, fg = lane >> 4;
    const int HWq = a.Hq * a.Wq;
    const int oh = a.phase_oh[phase], ow = a.phase_ow[phase];
    const bool want_stats = a.stats_part != nullptr;
    float st1[FN][4], st2[FN][4];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) st1[i][r] = st2[i][r] = 0.f;
    float bias[FN][4];
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        const int co = tn * BN + wn * PN + i * 16 + fg * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[i][r] = (a.bias && a.splitk == 1 && !a.raw_out && co + r < a.bias_n) ? a.bias[co + r] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < FM; ++j) {
        const int m = tm * BM + wm * PM + j * 16 + fr;
        if (m >= a.Mtot) continue;
        const int n = m / HWq, rem = m - n * HWq;
        const int hq = rem / a.Wq, wq = rem - hq * a.Wq;
        if (hq * a.out_step + oh >= a.Ho || wq * a.out_step + ow >= a.Wo) continue;     // odd-sized outputs of a sub-pixel phase
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
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += bias[i][r];
                if (a.act != DL_ACT_NONE) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = apply_act(a.act, v[r]);
                }
                float *dst = reinterpret_cast<float *>(a.out) + opix * a.out_pstride + co;
                *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                if (want_stats) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { st1[i][r] += v[r]; st2[i][r] += v[r] * v[r]; }
                }
            }
        }
    }
    if (want_stats) {
        __syncthreads();                                           // every wave is out of the K loop: the tile buffers are dead
        float *red = reinterpret_cast<float *>(smem_raw);          // [WM][2][BN]
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s1 = row16_sum(st1[i][r]), s2 = row16_sum(st2[i][r]);
                if (fr == 0) {
                    const int c = wn * PN + i * 16 + fg * 4 + r;
                    red[(wm * 2 + 0) * BN + c] = s1;
                    red[(wm * 2 + 1) * BN + c] = s2;
                }
            }
        __syncthreads();
        const int m0 = tm * BM;
        const int n = m0 / HWq;
        const int chunk = ((m0 - n * HWq) / BM) * a.n_phase + phase;
        for (int c = tid; c < BN; c += WM * WN * 64) {
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

// ------------------------------------------------------------------------------------------------------------------
// 8-phase kernel, strict policy: the schedule, slot order, barrier stagger and counted vmcnt of conv_gemm_8ph_kernel (see there), with
// 32-channel K steps.  Per phase: 12 ds_read_b128 (or 4 / 8 / 0) + 2 DMA instructions per wave, 24 MFMAs.
// ------------------------------------------------------------------------------------------------------------------
// Measured (r03, ResnetBlock 3x3 256->256 @ 8x128x128, forward, us, same box; profiles/r03/x3_variants.txt): in-LDS conversion (default) 470-488,
// conversion VALU interleaved 1:1 with the wave's own MFMAs + deferred waits (VAR 8) 478-500, per-wave register split (VAR 4) 491-506, round-1/2
// register-staged 128 x 128 kernel 608; NO split at all (VAR 2, wrong results) 379-389, and the same 85-100 us gap on all-zero data -- three
// placements of the split cost the same, so what is left is not where the VALU work sits; bf16 kernel on the same boxes: 162-174 (x 2.9).
template <int IN_ACT, int ABL, int VAR = 0>      // VAR (DL_X3_VAR, timing experiments): bit 0 = no s_setprio around the MFMAs, bit 1 = NO split (wrong results), bit 2 = per-wave register split after the fragment read (round-3 first version) instead of the in-LDS conversion
__global__ void __launch_bounds__(512) conv_gemm_8ph_x3_kernel(const ConvArgs a) {
    constexpr int BM = 256, BN = 256, KC = 32, WM = 2, WN = 4;
    constexpr int FM = 8, FN = 4;
    constexpr int ROWB = 128;                      // bytes of one LDS row (32 fp32 channels, or 32 hi + 32 lo bf16)
    constexpr int HALFB = 128 * ROWB;              // bytes of one half-tile slot (16 KB)
    constexpr int S_W = 0, S_X = 2;                // slot order inside a buffer: WB_0, WB_1, XA_0, XA_1

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    int *tapd_lds = reinterpret_cast<int *>(smem_raw + 8 * HALFB);          // element offset (dh*Wi + dw)*pstride of every tap

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
    const int nk_total = ntaps * a.Ci / KC;
    const int nk_per = (nk_total + a.splitk - 1) / a.splitk;
    const int kt_begin = ks * nk_per;
    const int T = ABL == 4 ? 0 : max(min(nk_total, kt_begin + nk_per) - kt_begin, 0);

    if (tid < DL_MAX_TAPS) {
        const int16_t tp = a.taps[tid];
        tapd_lds[tid] = ((int)(int8_t)(tp & 0xff) * a.Wi + (int)(int8_t)((tp >> 8) & 0xff)) * a.in_pstride;
    }

    const float *in = reinterpret_cast<const float *>(a.in);
    const float *zero = reinterpret_cast<const float *>(g_zero_page);
    const int HWq = a.Hq * a.Wq;
    const int lrow = lane >> 3, lcp = lane & 7;    // row inside one DMA instruction's 8-row slab, 16-byte position in the row

    // ---- staging geometry: instruction i of this wave fills slot rows s = (wave*2 + i)*8 + lrow of a half-tile
    const float *x_ptr[2][2];
    unsigned long long x_mask[2][2];               // bit t: tap (tap0 + t) of this pixel is inside the image
    const bf16_t *w_ptr[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s = (wave * 2 + i) * 8 + lrow;
            {   // XA_h: slot row s = pixel (s>>6)*128 + h*64 + (s&63) of the tile; this lane fetches fp32 chunk lcp ^ f(s>>1) (4 channels)
                const int c4 = (lcp ^ x3_swz((s >> 1) & 7)) * 4;
                const int m = tm * BM + (s >> 6) * 128 + h * 64 + (s & 63);
                const bool ok = m < a.Mtot;
                const int mm = ok ? m : 0;
                const int n = mm / HWq, rem = mm - n * HWq;
                const int hq = rem / a.Wq, wq = rem - hq * a.Wq;
                const int hi0 = hq * a.in_step, wi0 = wq * a.in_step;
                x_ptr[h][i] = in + ((size_t)(n * a.Hi + hi0) * a.Wi + wi0) * (size_t)a.in_pstride + c4;
                unsigned long long mk = 0;
                if (ok)
                    for (int t = 0; t < ntaps; ++t) {
                        const int16_t tp = a.taps[tap0 + t];
                        const int hi = hi0 + (int)(int8_t)(tp & 0xff), wi = wi0 + (int)(int8_t)((tp >> 8) & 0xff);
                        if (((unsigned)hi < (unsigned)a.Hi) && ((unsigned)wi < (unsigned)a.Wi)) mk |= 1ull << t;
                    }
                x_mask[h][i] = mk;
            }
            {   // WB_h: slot row s = channel (s>>5)*64 + h*32 + (s&31) of the tile; chunk c < 4: hi image, c >= 4: lo image (8 bf16 each)
                const int c = lcp ^ ((s >> 1) & 7);
                const int row = (s >> 5) * 64 + h * 32 + (s & 31);
                w_ptr[h][i] = ((c & 4) ? a.w_lo : a.w_hi) + (size_t)(tn * BN + row) * a.w_kstride + kbase + (c & 3) * 8;
            }
        }

    auto stage_x = [&](auto BUF, auto H, ptrdiff_t delta, int tl) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::value, h = decltype(H)::value;
        char *dst = smem_raw + (buf * 4 + S_X + h) * HALFB + wave * (16 * ROWB);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool ok = (x_mask[h][i] >> tl) & 1ull;
            const float *src = ok ? x_ptr[h][i] + delta : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + i * 8 * ROWB), 16, 0, 0);
        }
    };
    auto stage_w = [&](auto BUF, auto H, size_t wk) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::value, h = decltype(H)::value;
        char *dst = smem_raw + (buf * 4 + S_W + h) * HALFB + wave * (16 * ROWB);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(w_ptr[h][i] + wk),
                                             (__attribute__((address_space(3))) void *)(dst + i * 8 * ROWB), 16, 0, 0);
    };

    // K step u (counted from kt_begin) covers channels [ch, ch + KC) of tap tl: tap-major order, stateless (wave-uniform SALU)
    // a.k_order8 (default for this kernel; DL_X3_KORDER=0 = tap-major): channel-chunk-major -- the taps of one 32-channel chunk back to back, so that the
    // kernel rows re-read the halo slab while it is still in the XCD's L2 (PMC: 2.8-5.1x HBM-side read amplification in tap-major order,
    // profiles/r03/pmc_strict_conv256.json).  Same-box A/B (r03): isolated forward 485 -> 465 us, strict step 218.0 -> 211.8 ms (two alternations)
#define DL_X3_TL(u) (a.k_order8 ? ((kt_begin + (u)) % ntaps) : (((kt_begin + (u)) * KC) >> a.log2Ci))
#define DL_X3_CH(u) (a.k_order8 ? (((kt_begin + (u)) / ntaps) * KC) : (((kt_begin + (u)) * KC) & (a.Ci - 1)))

    f32x4_t acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    __syncthreads();     // tap table visible

    int tl1, tl2;
    ptrdiff_t d1, d2;
    size_t wk2;
    {
        const int tl0 = DL_X3_TL(0), ch0 = DL_X3_CH(0);
        tl1 = DL_X3_TL(1); const int ch1 = DL_X3_CH(1);
        tl2 = DL_X3_TL(2); const int ch2 = DL_X3_CH(2);
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
    const int fr = lane & 15, fg = lane >> 4;
    int foffw[2], foffx[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        foffw[e] = fr * ROWB + (((e * 4 + fg) ^ ((fr >> 1) & 7)) << 4);
        foffx[e] = fr * ROWB + (((2 * fg + e) ^ x3_swz((fr >> 1) & 7)) << 4);
    }

    bf16x8_t xh[4], xl[4], wf0[2][2], wf1[2][2];       // wf[plane][i]
    if constexpr (ABL >= 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xh[j] = bf16x8_t{(short)(0x3f80 + lane), (short)(0x3f00 + j), 0x3e80, 0x3f81, (short)0xbf80, 0x3f10, 0x3e90, 0x3f91};
            xl[j] = bf16x8_t{(short)(0x3b80 + lane), (short)(0x3b00 + j), 0x3a80, 0x3b81, (short)0xbb80, 0x3b10, 0x3a90, 0x3b91};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) { wf0[0][i] = xh[i]; wf0[1][i] = xl[i]; wf1[0][i] = xh[i + 2]; wf1[1][i] = xl[i + 2]; }
    }
    constexpr bool CVT = !(VAR & 4) && !(VAR & 2);      // in-LDS conversion by the staging wave (default)
    constexpr bool CVTM = CVT && (VAR & 8);             // ... with its VALU work and writes interleaved into the wave's OWN MFMA section (see step())
    // In-LDS conversion: the two DMA instructions of an activation half-tile left 16 bytes (4 fp32 channels of one pixel) per lane in LDS.
    // The SAME wave (only its own vmcnt has to retire) reads them back, splits, and writes hi / lo in place: the chunk pair (2g, 2g+1) of a
    // pixel -- channels 8g..8g+7 in fp32 -- becomes [8 hi | 8 lo] bf16, i.e. exactly the two 16-byte fragments a consumer lane needs, so the
    // fragment reads below need no VALU at all and every element is split once per workgroup instead of once per consuming wave (4x).
    auto convert_x = [&](auto BUF, auto H) __attribute__((always_inline)) {
        if (ABL >= 2 || !CVT || a.in_split) return;          // in_split: the producer wrote [8 hi | 8 lo] per channel group -- what this would leave in LDS
        constexpr int buf = decltype(BUF)::value, h = decltype(H)::value;
        char *base = smem_raw + (buf * 4 + S_X + h) * HALFB + wave * (16 * ROWB);
        f32x4_t v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) v[i] = *reinterpret_cast<const f32x4_t *>(base + i * 8 * ROWB + lane * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s = (wave * 2 + i) * 8 + lrow;
            const int e = (lcp ^ x3_swz((s >> 1) & 7)) & 1;            // this lane holds the first (0) or second (1) four channels of its group
            u32x2_t hi, lo;
            x3_split4<IN_ACT>(v[i], hi, lo);
            char *row = base + i * 8 * ROWB + lrow * ROWB;
            *reinterpret_cast<u32x2_t *>(row + ((lcp ^ e) << 4) + e * 8) = hi;
            *reinterpret_cast<u32x2_t *>(row + ((lcp ^ e ^ 1) << 4) + e * 8) = lo;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the planes are in LDS before this wave reaches the next barrier
    };
    f32x4_t cv[2];
    auto cvt_read = [&](auto BUF, auto H) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::value, h = decltype(H)::value;
        const char *base = smem_raw + (buf * 4 + S_X + h) * HALFB + wave * (16 * ROWB);
#pragma unroll
        for (int i = 0; i < 2; ++i) cv[i] = *reinterpret_cast<const f32x4_t *>(base + i * 8 * ROWB + lane * 16);
    };
    int cvt_off[2][2];          // [DMA instruction i][hi, lo]: byte offset of this lane's 8 output bytes inside the half-tile slot
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = (wave * 2 + i) * 8 + lrow;
        const int e = (lcp ^ x3_swz((s >> 1) & 7)) & 1;
        cvt_off[i][0] = wave * (16 * ROWB) + i * 8 * ROWB + lrow * ROWB + ((lcp ^ e) << 4) + e * 8;
        cvt_off[i][1] = wave * (16 * ROWB) + i * 8 * ROWB + lrow * ROWB + ((lcp ^ e ^ 1) << 4) + e * 8;
    }
    f32x4_t xraw[4][2];
    auto read_x = [&](auto BUF, auto H) __attribute__((always_inline)) {
        if (ABL >= 2) return;
        constexpr int buf = decltype(BUF)::value, h = decltype(H)::value;
        const char *base = smem_raw + (buf * 4 + S_X + h) * HALFB + wm * (64 * ROWB);
        if constexpr (CVT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xh[j] = *reinterpret_cast<const bf16x8_t *>(base + j * 16 * ROWB + foffx[0]);
                xl[j] = *reinterpret_cast<const bf16x8_t *>(base + j * 16 * ROWB + foffx[1]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 2; ++e) xraw[j][e] = *reinterpret_cast<const f32x4_t *>(base + j * 16 * ROWB + foffx[e]);
        }
    };
    auto split_x = [&](bool w_first) __attribute__((always_inline)) {
        if (ABL >= 2) return;
        if constexpr (CVT) {        // nothing to split; phase 0 only retires the 4 WB_0 reads (issued first): WB_0 is restaged next phase
            if (w_first) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (VAR & 2) { xh[j] = __builtin_bit_cast(bf16x8_t, xraw[j][0]); xl[j] = __builtin_bit_cast(bf16x8_t, xraw[j][1]); }
            else x3_split8<IN_ACT>(xraw[j][0], xraw[j][1], xh[j], xl[j]);
        }
    };
    auto read_w = [&](auto BUF, auto H, bf16x8_t (&wf)[2][2]) __attribute__((always_inline)) {
        if (ABL >= 2) return;
        constexpr int buf = decltype(BUF)::value, h = decltype(H)::value;
        const char *base = smem_raw + (buf * 4 + S_W + h) * HALFB + wn * (32 * ROWB);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) wf[pl][i] = *reinterpret_cast<const bf16x8_t *>(base + i * 16 * ROWB + foffw[pl]);
    };
    if (T > 0) convert_x(IC<0>{}, IC<0>{});         // XA_0 of step 0 (XA_1(0) is converted in phase 1 of step 0)
    DL_BAR();
    if (grp1) DL_BAR();          // stagger: waves 4-7 run one barrier behind

    auto mma_raw = [&](auto IB, auto JA, const bf16x8_t (&wf)[2][2]) __attribute__((always_inline)) {
        constexpr int ib = decltype(IB)::value * 2, ja = decltype(JA)::value * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[ib + i][ja + j] = dl_mfma16(wf[1][i], xh[j], acc[ib + i][ja + j]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[ib + i][ja + j] = dl_mfma16(wf[0][i], xl[j], acc[ib + i][ja + j]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[ib + i][ja + j] = dl_mfma16(wf[0][i], xh[j], acc[ib + i][ja + j]);
    };
    auto mma_q = [&](auto IB, auto JA, const bf16x8_t (&wf)[2][2]) __attribute__((always_inline)) {
        if (ABL == 2) return;
        if constexpr (!(VAR & 1)) __builtin_amdgcn_s_setprio(1);
        mma_raw(IB, JA, wf);
        if constexpr (!(VAR & 1)) __builtin_amdgcn_s_setprio(0);
    };

    // MFMA section that also finishes an in-LDS conversion started by cvt_read() in the LDS section of the same phase: the 24 VALU ops
    // of the split go BETWEEN the wave's own 24 MFMAs (an MFMA occupies the matrix pipe for 16 clocks but only one issue slot), the four
    // 8-byte writes follow, and lgkmcnt(0) precedes the section's closing barrier
    auto mma_q_cvt = [&](auto IB, auto JA, const bf16x8_t (&wf)[2][2], auto CB, auto CH, bool do_cvt) __attribute__((always_inline)) {
        constexpr int cb = decltype(CB)::value, chh = decltype(CH)::value;
        u32x2_t hi[2], lo[2];
        __builtin_amdgcn_s_setprio(1);
        __builtin_amdgcn_sched_barrier(0);
        if (do_cvt) {
#pragma unroll
            for (int i = 0; i < 2; ++i) x3_split4<IN_ACT>(cv[i], hi[i], lo[i]);
        }
        mma_raw(IB, JA, wf);
        if (do_cvt) {
            char *slot = smem_raw + (cb * 4 + S_X + chh) * HALFB;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *reinterpret_cast<u32x2_t *>(slot + cvt_off[i][0]) = hi[i];
                *reinterpret_cast<u32x2_t *>(slot + cvt_off[i][1]) = lo[i];
            }
        }
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);      // 1 VALU
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);          // the 4 LDS writes
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
        // no wait here: LDS operations of a wave complete in order, and the NEXT phase's LDS section ends with a counted lgkmcnt that
        // leaves only that phase's own fragment reads in flight -- one barrier before any other wave can read these rows (see step())
    };

    auto step = [&](int t, auto BUF) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::value;
        const bool more1 = t + 1 < T, more2 = t + 2 < T;
        // ---- phase 0: reads WB_0, XA_0; stages XA_1(t+1)
        read_w(IC<buf>{}, IC<0>{}, wf0);
        __builtin_amdgcn_sched_barrier(0);
        read_x(IC<buf>{}, IC<0>{});
        __builtin_amdgcn_sched_barrier(0);
        if (more1 && ABL != 1 && ABL != 3) stage_x(IC<buf ^ 1>{}, IC<1>{}, d1, tl1);
        split_x(true);                                          // register-split variant: waits for all 12 reads (WB_0 may be restaged next phase)
        if constexpr (CVTM) cvt_read(IC<buf>{}, IC<1>{});       // XA_1(t): landed since phase 3 of step t-1 (this wave's own DMA); read in phase 2
        __builtin_amdgcn_sched_barrier(0);
        DL_BAR();
        if constexpr (CVTM) mma_q_cvt(IC<0>{}, IC<0>{}, wf0, IC<buf>{}, IC<1>{}, true);
        else mma_q(IC<0>{}, IC<0>{}, wf0);
        __builtin_amdgcn_sched_barrier(0);
        DL_BAR();
        // ---- phase 1: reads WB_1; stages WB_0(t+2)
        read_w(IC<buf>{}, IC<1>{}, wf1);
        if (more2 && ABL != 1 && ABL != 3) stage_w(IC<buf>{}, IC<0>{}, wk2);
        if constexpr (!CVTM) convert_x(IC<buf>{}, IC<1>{});      // XA_1(t): landed since phase 3 of step t-1, read next phase
        else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");  // phase 0's conversion writes (older than the 4 WB_1 reads) are in LDS
        __builtin_amdgcn_sched_barrier(0);
        DL_BAR();
        mma_q(IC<1>{}, IC<0>{}, wf1);
        DL_BAR();
        // ---- phase 2: reads XA_1; stages XA_0(t+2)
        read_x(IC<buf>{}, IC<1>{});
        __builtin_amdgcn_sched_barrier(0);
        if (more2 && ABL != 1 && ABL != 3) stage_x(IC<buf>{}, IC<0>{}, d2, tl2);
        split_x(false);
        if constexpr (CVTM) {
            // XA_0(t+1) was staged one step ago (phase 2 of step t-1): four half-tiles have been issued since (8 DMA instructions)
            if (more2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (more1) cvt_read(IC<buf ^ 1>{}, IC<0>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        DL_BAR();
        if constexpr (CVTM) mma_q_cvt(IC<1>{}, IC<1>{}, wf1, IC<buf ^ 1>{}, IC<0>{}, more1);
        else mma_q(IC<1>{}, IC<1>{}, wf1);
        __builtin_amdgcn_sched_barrier(0);
        DL_BAR();
        // ---- phase 3: stages WB_1(t+2)
        const int tl3 = DL_X3_TL(t + 3), ch3 = DL_X3_CH(t + 3);
        const ptrdiff_t d3 = (ptrdiff_t)tapd_lds[tap0 + min(tl3, ntaps - 1)] + ch3;
        if (more2 && ABL != 1 && ABL != 3) {
            stage_w(IC<buf>{}, IC<1>{}, wk2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // everything of step t+1 has landed; step t+2's three stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (!CVTM) { if (more1) convert_x(IC<buf ^ 1>{}, IC<0>{}); }      // XA_0(t+1): retired by the wait above, read in phase 0 of the next step
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // phase 2's conversion writes are in LDS
        __builtin_amdgcn_sched_barrier(0);
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
#undef DL_X3_TL
#undef DL_X3_CH

    x3_epilogue<BM, BN, WM, WN>(a, acc, tm, tn, phase, ks, wm, wn, lane, tid, smem_raw);
}

template <int IN_ACT, int ABL, int VAR = 0>
static int launch_conv_8ph_x3(const ConvArgs &a0, hipStream_t stream) {
    ConvArgs a = a0;
    a.tiles_m = (a.Mtot + 255) / 256;
    a.tiles_n = a.Co / 256;
    static const char *korder = DL_DEV_ENV("DL_X3_KORDER");
    a.k_order8 = (korder && korder[0] == '0') ? 0 : 1;
    constexpr size_t smem = (size_t)8 * 128 * 128 + DL_MAX_TAPS * sizeof(int);
    auto kern = conv_gemm_8ph_x3_kernel<IN_ACT, ABL, VAR>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) DL_FAIL("dl_conv_forward: hipFuncSetAttribute(%zu): %s", smem, hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(a.tiles_m * a.tiles_n, a.n_phase * a.splitk);
    hipLaunchKernelGGL(kern, grid, dim3(512), smem, stream, a);
    DL_CHECK_LAUNCH("dl_conv_forward(8-phase x3)");
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// One-barrier kernel, strict policy, for every other tile shape (the strict twin of conv_gemm_glds_kernel): BM pixels x BN channels x 32
// channels per K step, 4 waves, two LDS buffers, the DMA of step t+1 issued before the fragment reads / MFMAs of step t.  IN_ACT: the
// UNet's pre-activation (LeakyReLU / ReLU on the conv INPUT, networks.py:578-602) is applied to the fragment in registers before the
// split -- the strict policy needs no separate activation pass for those layers.
// ------------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool UTAP, int IN_ACT>
__global__ void __launch_bounds__(WM * WN * 64) conv_gemm_glds_x3_kernel(const ConvArgs a) {
    constexpr int NW = WM * WN;
    constexpr int KC = 32, ROWB = 128, RPI = 8;                // channels per K step, LDS row bytes, rows per DMA wave-instruction
    constexpr int X_INS = (BM + NW * RPI - 1) / (NW * RPI);
    constexpr int W_INS = (BN + NW * RPI - 1) / (NW * RPI);
    constexpr int PM = BM / WM, PN = BN / WN, FM = PM / 16, FN = PN / 16;
    constexpr int XTB = BM * ROWB, WTB = BN * ROWB, BUFB = XTB + WTB;
    static_assert(NW == 4, "4 waves");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    int16_t *tap_lds = reinterpret_cast<int16_t *>(smem_raw + 2 * BUFB);
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
    const int nk_total = (ntaps * a.Ci + 63) / 64 * 64 / KC;
    const int nk_per = (nk_total + a.splitk - 1) / a.splitk;
    const int kt_begin = ks * nk_per;
    const int kt_end = min(nk_total, kt_begin + nk_per);

    if (tid < DL_MAX_TAPS) {
        const int16_t tp = a.taps[tid];
        tap_lds[tid] = tp;
        tapd_lds[tid] = ((int)(int8_t)(tp & 0xff) * a.Wi + (int)(int8_t)((tp >> 8) & 0xff)) * a.in_pstride;
    }

    const float *in = reinterpret_cast<const float *>(a.in);
    const float *zero = reinterpret_cast<const float *>(g_zero_page);
    const int HWq = a.Hq * a.Wq;
    const int lrow = lane >> 3, lcp = lane & 7;

    const float *x_ptr[X_INS];
    int x_hi0[X_INS], x_wi0[X_INS], x_chunk[X_INS];
    bool x_ok[X_INS];
    unsigned long long x_mask[X_INS];
#pragma unroll
    for (int i = 0; i < X_INS; ++i) {
        const int row = (wave * X_INS + i) * RPI + lrow;
        const int m = tm * BM + row;
        x_ok[i] = (row < BM) && (m < a.Mtot);
        const int mm = x_ok[i] ? m : 0;
        const int n = mm / HWq, rem = mm - n * HWq;
        const int hq = rem / a.Wq, wq = rem - hq * a.Wq;
        x_hi0[i] = hq * a.in_step;
        x_wi0[i] = wq * a.in_step;
        x_chunk[i] = lcp ^ x3_swz((row >> 1) & 7);        // fp32 chunk (4 channels) this lane fetches for its LDS slot
        x_ptr[i] = in + ((size_t)(n * a.Hi + x_hi0[i]) * a.Wi + x_wi0[i]) * (size_t)a.in_pstride + x_chunk[i] * 4;
        unsigned long long mk = 0;
        if (UTAP && x_ok[i]) {
            for (int t = 0; t < ntaps; ++t) {
                const int16_t tp = a.taps[tap0 + t];
                const int hi = x_hi0[i] + (int)(int8_t)(tp & 0xff), wi = x_wi0[i] + (int)(int8_t)((tp >> 8) & 0xff);
                if (((unsigned)hi < (unsigned)a.Hi) && ((unsigned)wi < (unsigned)a.Wi)) mk |= 1ull << t;
            }
        }
        x_mask[i] = mk;
    }
    const bf16_t *w_ptr[W_INS];
#pragma unroll
    for (int i = 0; i < W_INS; ++i) {
        const int row = (wave * W_INS + i) * RPI + lrow;
        const int c = lcp ^ ((row >> 1) & 7);
        // rows beyond the tile (narrow-N configs) re-read row 0: their LDS slots are never consumed
        w_ptr[i] = ((c & 4) ? a.w_lo : a.w_hi) + (size_t)(tn * BN + (row < BN ? row : 0)) * a.w_kstride + kbase + (c & 3) * 8;
    }

    int is_tl = 0, is_ch = 0;
    int tapd_next = 0;
    auto issue_tile = [&](int kt, int buf) {
        char *base = smem_raw + buf * BUFB;
        size_t wk;
        if constexpr (UTAP) {
            const int tl = is_tl;
            const ptrdiff_t delta = (ptrdiff_t)tapd_next + is_ch;
            wk = (size_t)tl * a.Ci + is_ch;
            is_ch += KC;
            if (is_ch == a.Ci) { is_ch = 0; ++is_tl; }
            tapd_next = tapd_lds[tap0 + min(is_tl, ntaps - 1)];
#pragma unroll
            for (int i = 0; i < X_INS; ++i) {
                const int row0 = (wave * X_INS + i) * RPI;
                if (row0 < BM) {
                    const bool ok = (x_mask[i] >> tl) & 1ull;
                    const float *src = ok ? x_ptr[i] + delta : zero;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(base + row0 * ROWB), 16, 0, 0);
                }
            }
        } else {
            wk = (size_t)kt * KC;
#pragma unroll
            for (int i = 0; i < X_INS; ++i) {
                const int row0 = (wave * X_INS + i) * RPI;
                if (row0 < BM) {
                    const int k0 = kt * KC + x_chunk[i] * 4;
                    const int tl = k0 >> a.log2Ci;
                    const int ci = k0 & (a.Ci - 1);
                    const int tli = tl < ntaps ? tl : 0;
                    const int16_t t = tap_lds[tap0 + tli];
                    const int dh = (int)(int8_t)(t & 0xff), dw = (int)(int8_t)((t >> 8) & 0xff);
                    int hi = x_hi0[i] + dh, wi = x_wi0[i] + dw;
                    if (a.pad_mode == DL_PAD_REFLECT) { hi = reflect_idx(hi, a.Hi); wi = reflect_idx(wi, a.Wi); }
                    const bool ok = x_ok[i] && (tl < ntaps) && ((unsigned)hi < (unsigned)a.Hi) && ((unsigned)wi < (unsigned)a.Wi);
                    const ptrdiff_t off = ((ptrdiff_t)(hi - x_hi0[i]) * a.Wi + (wi - x_wi0[i])) * (ptrdiff_t)a.in_pstride + ci - x_chunk[i] * 4;
                    const float *src = ok ? x_ptr[i] + off : zero;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(base + row0 * ROWB), 16, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < W_INS; ++i) {
            const int row0 = (wave * W_INS + i) * RPI;
            if (row0 < BN) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(w_ptr[i] + wk),
                                                 (__attribute__((address_space(3))) void *)(base + XTB + row0 * ROWB), 16, 0, 0);
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
        is_tl = (kt_begin * KC) >> a.log2Ci;
        is_ch = (kt_begin * KC) & (a.Ci - 1);
        tapd_next = tapd_lds[tap0 + min(is_tl, ntaps - 1)];
    }
    if (kt_begin < kt_end) issue_tile(kt_begin, 0);
    __syncthreads();     // drains the DMA (vmcnt(0)) + barrier

    const int fr = lane & 15, fg = lane >> 4;
    int foffw[2], foffx[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        foffw[e] = fr * ROWB + (((e * 4 + fg) ^ ((fr >> 1) & 7)) << 4);
        foffx[e] = fr * ROWB + (((2 * fg + e) ^ x3_swz((fr >> 1) & 7)) << 4);
    }

    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        if (kt + 1 < kt_end) issue_tile(kt + 1, cur ^ 1);
        const char *Xs = smem_raw + cur * BUFB + wm * PM * ROWB, *Ws = smem_raw + cur * BUFB + XTB + wn * PN * ROWB;
        bf16x8_t wh[FN], wl[FN], xh[FM], xl[FM];
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            wh[i] = *reinterpret_cast<const bf16x8_t *>(Ws + i * 16 * ROWB + foffw[0]);
            wl[i] = *reinterpret_cast<const bf16x8_t *>(Ws + i * 16 * ROWB + foffw[1]);
        }
        if (a.in_split) {            // producer-written split copy: chunk 2*fg holds the 8 hi values, chunk 2*fg + 1 the 8 lo values of this lane's K slice
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                xh[j] = *reinterpret_cast<const bf16x8_t *>(Xs + j * 16 * ROWB + foffx[0]);
                xl[j] = *reinterpret_cast<const bf16x8_t *>(Xs + j * 16 * ROWB + foffx[1]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                const f32x4_t r0 = *reinterpret_cast<const f32x4_t *>(Xs + j * 16 * ROWB + foffx[0]);
                const f32x4_t r1 = *reinterpret_cast<const f32x4_t *>(Xs + j * 16 * ROWB + foffx[1]);
                x3_split8<IN_ACT>(r0, r1, xh[j], xl[j]);
            }
        }
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = dl_mfma16(wl[i], xh[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = dl_mfma16(wh[i], xl[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = dl_mfma16(wh[i], xh[j], acc[i][j]);
        __syncthreads();
    }

    x3_epilogue<BM, BN, WM, WN>(a, acc, tm, tn, phase, ks, wm, wn, lane, tid, smem_raw);
}

template <int BM, int BN, int WM, int WN, bool UTAP, int IN_ACT>
static int launch_conv_glds_x3_impl(const ConvArgs &a0, hipStream_t stream) {
    ConvArgs a = a0;
    a.tiles_m = (a.Mtot + BM - 1) / BM;
    a.tiles_n = (a.Co + BN - 1) / BN;
    constexpr size_t smem = (size_t)2 * (BM + BN) * 128 + DL_MAX_TAPS * (sizeof(int16_t) + sizeof(int));
    auto kern = conv_gemm_glds_x3_kernel<BM, BN, WM, WN, UTAP, IN_ACT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) DL_FAIL("dl_conv_forward: hipFuncSetAttribute(%zu): %s", smem, hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(a.tiles_m * a.tiles_n, a.n_phase * a.splitk);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, stream, a);
    DL_CHECK_LAUNCH("dl_conv_forward(glds x3)");
    return 0;
}

template <int BM, int BN, int WM, int WN, int IN_ACT>
static int launch_conv_glds_x3(const ConvArgs &a, hipStream_t stream) {
    // UTAP: every 32-channel K step lies inside one tap (Cin >= 32) and padding is zero -> scalar tap decode
    if (a.Ci >= 32 && a.pad_mode == DL_PAD_ZERO) return launch_conv_glds_x3_impl<BM, BN, WM, WN, true, IN_ACT>(a, stream);
    return launch_conv_glds_x3_impl<BM, BN, WM, WN, false, IN_ACT>(a, stream);
}

template <int IN_ACT>
static int dispatch_glds_x3_tiles(const ConvArgs &a, hipStream_t stream) {
    if (a.Co <= 16) return launch_conv_glds_x3<256, 16, 4, 1, IN_ACT>(a, stream);
    if (a.Co <= 32) return launch_conv_glds_x3<256, 32, 4, 1, IN_ACT>(a, stream);      // the head's (co, kw)-stacked rows (21 -> 32): half the MFMAs of the 64-wide tile
    if (a.Co <= 64) return launch_conv_glds_x3<128, 64, 2, 2, IN_ACT>(a, stream);
    return launch_conv_glds_x3<128, 128, 2, 2, IN_ACT>(a, stream);
}

// which strict-policy descriptors take the direct-to-LDS kernels of this file (DL_NO_X3_GLDS=1: none -- the round-1 register-staged kernel, A/B)
static bool x3_glds_applies(const dl_conv_desc *d) {
    const bool off = dl_switch(DL_SW_NO_X3_GLDS) != nullptr;
    if (off || d->in_dtype != DL_F32 || d->prec != DL_PREC_BF16X3) return false;
    if (d->in_act != DL_ACT_NONE && d->in_act != DL_ACT_RELU && d->in_act != DL_ACT_LRELU) return false;
    return d->in_pstride % 4 == 0 && d->Ci >= 8;
}

static bool x3_big_tile(int in_act, int pad_mode, int Ci, int mtot, int Co, int n_phase, int splitk) {
    static const bool no_big = DL_DEV_ENV("DL_NO_BIGTILE") != nullptr;
    return !no_big && in_act == DL_ACT_NONE && pad_mode == DL_PAD_ZERO && Ci >= 32 && big_tile_fills_gpu(mtot, Co, n_phase, splitk);
}

// tile height (pixels) of the strict direct-to-LDS dispatch (for dl_conv_stats_chunks)
static int x3_tile_bm(const dl_conv_desc *d) {
    if (x3_big_tile(d->in_act, d->pad_mode, d->Ci, d->N * d->Hq * d->Wq, d->Co, d->n_phase, d->splitk)) return 256;
    return d->Co <= 32 ? 256 : 128;
}

static const char *x3_kernel_name(const dl_conv_desc *d) {
    if (x3_big_tile(d->in_act, d->pad_mode, d->Ci, d->N * d->Hq * d->Wq, d->Co, d->n_phase, d->splitk)) return "conv_gemm_8ph_x3_kernel";
    if (d->Co <= 16) return "conv_gemm_glds_x3_kernel<256,16>";
    if (d->Co <= 32) return "conv_gemm_glds_x3_kernel<256,32>";
    if (d->Co <= 64) return "conv_gemm_glds_x3_kernel<128,64>";
    return "conv_gemm_glds_x3_kernel<128,128>";
}

// conv_gemm_w4x3_kernel (conv_w4x3.hip), the one-wave-per-SIMD tile for the strict policy: OPT-IN (DL_CONV_W4X3=1).  Same-box A/B (r04,
// profiles/r04/w4x3_ab.txt): isolated forward 390-395 us vs 398-402 us for the 8-phase strict kernel, data gradient 346-350 vs 348-354, inside the strict
// step 345.6-347.1 vs 343.7-344.2 us per launch and 204.8 vs 203.8 ms per step -- a tie.  Unlike the bf16 pair (conv_w4.hip: -13 %), the strict 8-phase
// kernel already issues three MFMAs per fragment pair, so halving the LDS bytes per MFMA buys nothing: both run at 1.3-1.35 PF/s executed, 80-90 % of
// what a bare MFMA loop sustains on random operands on these boxes (1.5-1.7 PF/s, bench.py roofline.sustained) -- the strict conv is power-bound.
static bool w4x3_enabled() {
    const char *e = dl_switch(DL_SW_CONV_W4X3);
    return e && e[0] == '1';
}

static int dispatch_tile_x3(const ConvArgs &a, hipStream_t stream) {
    if (x3_big_tile(a.in_act, a.pad_mode, a.Ci, a.Mtot, a.Co, a.n_phase, a.splitk)) {
        if (w4x3_enabled() && w4x3_eligible(a)) return launch_conv_w4x3(a, stream);
#ifdef DL_DEV_SWITCHES      // timing-only ablations (results WRONG by construction) and schedule variants: dev build only
        static const char *abl = DL_DEV_ENV("DL_CONV_ABLATE");
        if (abl && abl[0] == '1') return launch_conv_8ph_x3<DL_ACT_NONE, 1>(a, stream);
        if (abl && abl[0] == '2') return launch_conv_8ph_x3<DL_ACT_NONE, 2>(a, stream);
        if (abl && abl[0] == '3') return launch_conv_8ph_x3<DL_ACT_NONE, 3>(a, stream);
        if (abl && abl[0] == '4') return launch_conv_8ph_x3<DL_ACT_NONE, 4>(a, stream);
        if (a.in_split) return launch_conv_8ph_x3<DL_ACT_NONE, 0>(a, stream);      // (the timing variants below re-split their input)
        static const char *var = DL_DEV_ENV("DL_X3_VAR");
        if (var && var[0] == '1') return launch_conv_8ph_x3<DL_ACT_NONE, 0, 1>(a, stream);
        if (var && var[0] == '2') return launch_conv_8ph_x3<DL_ACT_NONE, 0, 2>(a, stream);
        if (var && var[0] == '3') return launch_conv_8ph_x3<DL_ACT_NONE, 0, 3>(a, stream);
        if (var && var[0] == '4') return launch_conv_8ph_x3<DL_ACT_NONE, 0, 4>(a, stream);
        if (var && var[0] == '5') return launch_conv_8ph_x3<DL_ACT_NONE, 0, 5>(a, stream);
        if (var && var[0] == '8') return launch_conv_8ph_x3<DL_ACT_NONE, 0, 8>(a, stream);
#endif
        return launch_conv_8ph_x3<DL_ACT_NONE, 0>(a, stream);
    }
    if (a.in_act == DL_ACT_RELU) return dispatch_glds_x3_tiles<DL_ACT_RELU>(a, stream);
    if (a.in_act == DL_ACT_LRELU) return dispatch_glds_x3_tiles<DL_ACT_LRELU>(a, stream);
    return dispatch_glds_x3_tiles<DL_ACT_NONE>(a, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// Strict policy on the 4-channel patch kernel (conv_c4.h): the ResnetGenerator stem forward and the head's data gradient with fp32
// activations and split-bf16 x3 products.  Same tile walk, patch geometry and register-resident weights as conv_c4_patch_kernel;
//   * the patch is fetched as fp32 (the first 4 of the 8 padded channels = 16 bytes per pixel) and split ONCE per staged pixel when it is
//     written to LDS: four copies -- hi A/B and lo A/B (B = A shifted by one pixel, see conv_c4.h) -- 23 KB;
//   * the hi and lo weight fragments both live in registers (112 VGPRs); three MFMAs per fragment pair, small terms first, term-major
//     over the 8 accumulators of a fragment batch;
//   * fp32 results leave straight from the accumulators (a lane holds 4 consecutive channels = one 16-byte store, the four lanes of a
//     pixel 64 contiguous bytes -- the x3_epilogue pattern; the bf16 kernel's LDS transposition would need a 128 KB tile here), with
//     the fused statistics of the stored values.
// Before (round 3, first half): the stem ran on conv_gemm_glds_x3_kernel<128,64> with 49 taps x 8 padded channels: 694 us; head data gradient 569 us.
// ------------------------------------------------------------------------------------------------------------------
template <int PADMODE, int ACT>
__global__ void __launch_bounds__(256, 2) conv_c4_patch_x3_kernel(const C4Args ca) {
    const ConvArgs &a = ca.a;
    constexpr int TR = 4, TC = 64, KR = 7, PR = TR + KR - 1, PW = 72;
    constexpr int NF = (TR / 2) * 4;
    constexpr int CSTR = PR * PW * 8 + 16;          // bytes from one patch copy to the next: hi A, hi B, lo A, lo B (copy B holds pixel i at byte (i + 1) * 8)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *red = reinterpret_cast<float *>(smem_raw + 4 * CSTR);            // [2 row halves][2][64] statistics

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ch = wave & 1, rh = wave >> 1;
    const int co0 = blockIdx.y * 64 + ch * 32;
    const float *in = reinterpret_cast<const float *>(a.in);
    const int ntiles = a.N * ca.tiles_h * ca.tiles_w;

    const int fr = lane & 15, fg = lane >> 4;
    bf16x8_t wh[2][KR], wl[2][KR];
#pragma unroll
    for (int cf = 0; cf < 2; ++cf) {
        const size_t roff = (size_t)(co0 + cf * 16 + fr) * a.w_kstride + a.phase_kbase[0];
#pragma unroll
        for (int kk = 0; kk < KR; ++kk) {
            const int t0 = fg == 0 ? ca.tap_src[kk][0] : fg == 1 ? ca.tap_src[kk][2] : fg == 2 ? ca.tap_src[kk][4] : ca.tap_src[kk][6];
            const int t1 = fg == 0 ? ca.tap_src[kk][1] : fg == 1 ? ca.tap_src[kk][3] : fg == 2 ? ca.tap_src[kk][5] : ca.tap_src[kk][7];
            const int o0 = (t0 >= 0 ? t0 : 0) * 8, o1 = (t1 >= 0 ? t1 : 0) * 8;
            const u32x2_t h0 = *reinterpret_cast<const u32x2_t *>(a.w_hi + roff + o0), h1 = *reinterpret_cast<const u32x2_t *>(a.w_hi + roff + o1);
            const u32x2_t l0 = *reinterpret_cast<const u32x2_t *>(a.w_lo + roff + o0), l1 = *reinterpret_cast<const u32x2_t *>(a.w_lo + roff + o1);
            const u32x4_t vh = {t0 >= 0 ? h0[0] : 0u, t0 >= 0 ? h0[1] : 0u, t1 >= 0 ? h1[0] : 0u, t1 >= 0 ? h1[1] : 0u};
            const u32x4_t vl = {t0 >= 0 ? l0[0] : 0u, t0 >= 0 ? l0[1] : 0u, t1 >= 0 ? l1[0] : 0u, t1 >= 0 ? l1[1] : 0u};
            wh[cf][kk] = __builtin_bit_cast(bf16x8_t, vh);
            wl[cf][kk] = __builtin_bit_cast(bf16x8_t, vl);
        }
    }

    constexpr int PPT = (PR * (TC + KR) + 255) / 256;       // patch pixels per thread (3)
    f32x4_t nxt[PPT];
    auto fetch_patch = [&](int tile) __attribute__((always_inline)) {
        int t = tile;
        const int tw = t % ca.tiles_w; t /= ca.tiles_w;
        const int th = t % ca.tiles_h;
        const int n = t / ca.tiles_h;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = tid + k * 256;
            const int pr = i / (TC + KR), pc = i - pr * (TC + KR);
            int hi = th * TR - 3 + pr, wi = tw * TC - 3 + pc;
            if (PADMODE == DL_PAD_REFLECT) { hi = reflect_idx(hi, a.Hi); wi = reflect_idx(wi, a.Wi); }
            const bool ok = i < PR * (TC + KR) && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi;
            f32x4_t v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4_t *>(in + ((size_t)(n * a.Hi + hi) * a.Wi + wi) * 8);
            nxt[k] = v;
        }
    };
    auto write_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = tid + k * 256;
            if (i < PR * (TC + KR)) {
                const int pr = i / (TC + KR), pc = i - pr * (TC + KR);
                u32x2_t h, l;
                x3_split4<DL_ACT_NONE>(nxt[k], h, l);
                char *p = smem_raw + (pr * PW + pc) * 8;
                *reinterpret_cast<u32x2_t *>(p) = h;
                *reinterpret_cast<u32x2_t *>(p + CSTR + 8) = h;
                *reinterpret_cast<u32x2_t *>(p + 2 * CSTR) = l;
                *reinterpret_cast<u32x2_t *>(p + 3 * CSTR + 8) = l;
            }
        }
    };
    if ((int)blockIdx.x < ntiles) fetch_patch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t = tile;
    const int tw = t % ca.tiles_w; t /= ca.tiles_w;
    const int th = t % ca.tiles_h;
    const int n = t / ca.tiles_h;
    const int h0 = th * TR, w0 = tw * TC;
    __syncthreads();                                      // every wave is done with the previous tile's patch (and statistics scratch)
    write_patch();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles && !(ca.abl & 4)) fetch_patch(tile + gridDim.x);

    f32x4_t acc[2][NF];
#pragma unroll
    for (int cf = 0; cf < 2; ++cf)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[cf][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // lane base: even fr -> copy A at pixel (fr + 2 fg), odd fr -> copy B (same pixel, stored 8 bytes further: 16-byte aligned again)
    const char *base = smem_raw + ((fr & 1) ? CSTR + 8 : 0) + (fr + 2 * fg) * 8 + (rh * (TR / 2)) * PW * 8;
    if (!(ca.abl & 1))
#pragma unroll
    for (int kk = 0; kk < KR; ++kk) {
#pragma unroll
        for (int jb = 0; jb < NF; jb += 4) {
            bf16x8_t xh[4], xl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int off = ((((jb + j) >> 2) + kk) * PW + ((jb + j) & 3) * 16) * 8;
                xh[j] = *reinterpret_cast<const bf16x8_t *>(base + off);
                xl[j] = *reinterpret_cast<const bf16x8_t *>(base + 2 * CSTR + off);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][jb + j] = dl_mfma16(wl[0][kk], xh[j], acc[0][jb + j]);
                acc[1][jb + j] = dl_mfma16(wl[1][kk], xh[j], acc[1][jb + j]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][jb + j] = dl_mfma16(wh[0][kk], xl[j], acc[0][jb + j]);
                acc[1][jb + j] = dl_mfma16(wh[1][kk], xl[j], acc[1][jb + j]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][jb + j] = dl_mfma16(wh[0][kk], xh[j], acc[0][jb + j]);
                acc[1][jb + j] = dl_mfma16(wh[1][kk], xh[j], acc[1][jb + j]);
            }
        }
    }

    // ---- epilogue: lane holds channels co0 + cf*16 + fg*4 .. +4 of pixel (row rh*2 + j/4, column (j%4)*16 + fr)
    const bool want_stats = a.stats_part != nullptr;
    float *out = reinterpret_cast<float *>(a.out);
#pragma unroll
    for (int cf = 0; cf < 2; ++cf) {
        const int co = co0 + cf * 16 + fg * 4;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = (co + r < a.bias_n) ? a.bias[co + r] : 0.f;
        }
        float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            f32x4_t v = acc[cf][j];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += bv[r];
            if (ACT == DL_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            } else if (ACT == DL_ACT_LRELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.2f * v[r];
            }
            const int h = h0 + rh * (TR / 2) + (j >> 2), w = w0 + (j & 3) * 16 + fr;
            if (!(ca.abl & 2)) *reinterpret_cast<f32x4_t *>(out + ((size_t)(n * a.Ho + h) * a.Wo + w) * a.out_pstride + co) = v;
#pragma unroll
            for (int r = 0; r < 4; ++r) { st1[r] += v[r]; st2[r] += v[r] * v[r]; }
        }
        if (want_stats) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s1 = row16_sum(st1[r]), s2 = row16_sum(st2[r]);
                if (fr == 0) {
                    const int c = ch * 32 + cf * 16 + fg * 4 + r;
                    red[(rh * 2 + 0) * 64 + c] = s1;
                    red[(rh * 2 + 1) * 64 + c] = s2;
                }
            }
        }
    }
    if (want_stats) {
        __syncthreads();
        if (tid < 64) {
            const int chunk = th * ca.tiles_w + tw;
            float *o = a.stats_part + ((size_t)(n * a.stats_nchunks + chunk) * 2) * a.Co + blockIdx.y * 64 + tid;
            o[0] = red[0 * 64 + tid] + red[2 * 64 + tid];
            o[a.Co] = red[1 * 64 + tid] + red[3 * 64 + tid];
        }
    }
  }   // tile loop
}

static int launch_conv_c4_x3(const ConvArgs &a0, const dl_conv_desc *d, hipStream_t stream) {
    if (!a0.w_lo) DL_FAIL("dl_conv_forward(c4 patch x3): the lo weight plane is missing");
    C4Args ca;
    c4_fill_args(ca, a0, d);
    constexpr size_t smem = 4 * ((4 + 6) * 72 * 8 + 16) + 4 * 64 * sizeof(float);      // four patch copies + statistics
    void (*kern)(const C4Args) = nullptr;
    const bool refl = d->pad_mode == DL_PAD_REFLECT;
    switch (d->act) {
        case DL_ACT_RELU: kern = refl ? conv_c4_patch_x3_kernel<DL_PAD_REFLECT, DL_ACT_RELU> : conv_c4_patch_x3_kernel<DL_PAD_ZERO, DL_ACT_RELU>; break;
        case DL_ACT_LRELU: kern = refl ? conv_c4_patch_x3_kernel<DL_PAD_REFLECT, DL_ACT_LRELU> : conv_c4_patch_x3_kernel<DL_PAD_ZERO, DL_ACT_LRELU>; break;
        default: kern = refl ? conv_c4_patch_x3_kernel<DL_PAD_REFLECT, DL_ACT_NONE> : conv_c4_patch_x3_kernel<DL_PAD_ZERO, DL_ACT_NONE>; break;
    }
    return c4_launch(kern, ca, d, smem, stream, "dl_conv_forward(c4 patch x3)");
}
