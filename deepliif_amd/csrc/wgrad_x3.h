// wgrad_x3.h -- included by wgrad.hip.  Weight gradient under the STRICT-PARITY policy (fp32 activations, split-bf16 x3 products) on the
// direct-to-LDS path.  Rounds 1-2 ran it on the register-staged wgrad_kernel<float, 3, 128> (128 x 128 tile, 4 waves): 816 us for the
// ResnetBlock shape against 185 us for the bf16 kernel.
//
// Both operands are fp32 [pixel][channel] and BOTH need the hi / lo split (the forward kernels only split the activations: the weights are
// packed as hi / lo once).  Splitting in registers after the fragment read, as conv_x3.h does, would cost 3 VALU ops per element in each of
// the 4 (P) or 2 (Q) waves that consume it; here every element is split ONCE per workgroup, in LDS, in place:
//   * a K step is 32 pixels.  P tile = 32 rows of BA fp32 channels, Q tile = 32 rows of 256 fp32 (tap, channel) columns; a row of 256
//     channels is 1 KB = exactly one global_load_lds wave-instruction, fetched in NATURAL order (lane L <- channels 4L..4L+3: fully coalesced
//     1 KB requests, no source-side swizzle);
//   * the wave that issued the DMA of a row also converts it (so only its OWN vmcnt has to retire -- no barrier between DMA and conversion):
//     ds_read_b128 of its rows, hi = bf16(x), lo = bf16(x - hi) (+ the operand's staged activation, dl_wgrad_desc p_act / q_act), then
//     two ds_write_b64 per lane and row: the fp32 row [BA x 4 B] becomes [hi plane: BA x 2 B | lo plane: BA x 2 B] in the same bytes, each
//     plane laid out exactly like a row of the bf16 kernel (32-byte slots XOR-swizzled by wswz(pixel)), so the MFMA fragments come from the
//     same conflict-free ds_read_b64_tr_b16 pattern (wgrad_glds_kernel), once per plane;
//   * per fragment pair three MFMAs, small terms first (lo*hi, hi*lo, hi*hi), term-major over the 16 accumulators of a pass;
//   * one barrier per K step:   issue DMA(t+1) -> other buffer | fragments + MFMAs of step t | vmcnt(0) | convert own rows of t+1 | barrier.
// Split-K over pixel ranges + wgrad_reduce_kernel as the bf16 path (deterministic).
#pragma once

template <int ROWB>
__device__ __forceinline__ bf16x8_t tr_fragment_plane(const char *plane, int prow0, int slot, int lane) {
    // rows prow0 + 8g + 4h + (m>>2) of a plane whose pixel rows are ROWB bytes apart; 16-channel block `slot` (32 B, XOR-swizzled by wswz(row))
    const int m = lane & 15, g = lane >> 4;
    const int x = (m >> 2) | ((g & 1) << 2);             // = wswz(p) for every p this lane reads (independent of h)
    const char *base = plane + (prow0 + 8 * g + (m >> 2)) * ROWB + ((slot ^ x) << 5) + (m & 3) * 8;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3))) *)(base));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3))) *)(base + 4 * ROWB));
    bf16x8_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// the same through inline assembly: see tr_fragment_swz_asm (wgrad.hip) -- the caller waits (tr_wait4)
template <int ROWB>
__device__ __forceinline__ bf16x8_t tr_fragment_plane_asm(const char *plane, int prow0, int slot, int lane) {
    const int m = lane & 15, g = lane >> 4;
    const int x = (m >> 2) | ((g & 1) << 2);
    const char *base = plane + (prow0 + 8 * g + (m >> 2)) * ROWB + ((slot ^ x) << 5) + (m & 3) * 8;
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)base;
    s16x4_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(4 * ROWB) : "memory");
    bf16x8_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// 4 fp32 values -> 4 bf16 hi (8 bytes) + 4 bf16 lo
__device__ __forceinline__ void x3w_split4(f32x4_t v, int act, u32x2_t &hi, u32x2_t &lo) {
    if (act == DL_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
    } else if (act == DL_ACT_LRELU) {
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

template <int BA, bool TRASM>
__global__ void __launch_bounds__(512) wgrad_glds_x3_kernel(const WgradArgs a, const WgradLayers lay) {
    constexpr int BJ = 256, BP = 32, NW = 8, WA = 2, WJ = 4;
    constexpr int PA = BA / WA, PJ = BJ / WJ, FA = PA / 16, FJ = PJ / 16;     // FJ = 4; FA = 8 (BA 256) or 4 (BA 128)
    constexpr int ROWA = BA * 4, ROWJ = BJ * 4;                               // bytes of one pixel row (fp32, later hi plane | lo plane)
    constexpr int TAB = BP * ROWA, TJB = BP * ROWJ, BUFB = TAB + TJB;
    constexpr int A_RPI = 1024 / ROWA;                                        // P rows per DMA wave-instruction (1 or 2)
    constexpr int A_INS = BP / (NW * A_RPI), J_INS = BP / NW;                 // DMA instructions per wave and K step (4 or 2; 4)
    constexpr int A_LPR = 64 / A_RPI;                                         // lanes per P row
    static_assert(BA == 256 || BA == 128, "tile heights");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave % WA, wj = wave / WA;
    int bid, ks, layer;
    if (a.xcd_group) {
        const int ntile = gridDim.x;
        const int logical = xcd_remap((blockIdx.z * gridDim.y + blockIdx.y) * ntile + blockIdx.x, ntile * gridDim.y * gridDim.z);
        const int grp = logical / ntile;
        bid = logical - grp * ntile;
        layer = grp / gridDim.y;
        ks = grp - layer * gridDim.y;
    } else {
        bid = xcd_remap(blockIdx.x, gridDim.x);
        ks = blockIdx.y;
        layer = blockIdx.z;
    }
    const int tj = bid % a.tiles_j, ta = bid / a.tiles_j;
    const int p_begin = ks * a.pchunk;
    const int p_end = min(a.Ptot, p_begin + a.pchunk);
    const int nk = (p_end > p_begin) ? (p_end - p_begin + BP - 1) / BP : 0;

    const float *P = reinterpret_cast<const float *>(lay.P[layer]);
    const float *Q = reinterpret_cast<const float *>(lay.Q[layer]);
    float *slab = lay.slab[layer];
    const float *zero = reinterpret_cast<const float *>(g_wzero_page);

    // ---- P: instruction i of this wave fills tile rows (wave*A_INS + i)*A_RPI + lane / A_LPR; this lane owns channels 4*(lane % A_LPR) .. +3
    // A SPLIT-COPY operand (a.p_split / a.q_split: the producer already wrote [8 hi | 8 lo] per group of 8 channels, same addressing as the
    // fp32 tensor) is fetched straight into the plane layout the in-LDS conversion would have produced: LDS position q of the hi (lo) plane
    // of pixel row r holds the hi (lo) half of channel group g = 2 * ((q >> 1) ^ wswz(r)) + (q & 1), i.e. source floats 8g (+4 for lo); the
    // permutation depends on the row through wswz, so it is one offset per DMA instruction.
    const int a_rsub = lane / A_LPR, a_c = lane % A_LPR;
    const float *p_src[A_INS];
#pragma unroll
    for (int i = 0; i < A_INS; ++i) {
        const int row = (wave * A_INS + i) * A_RPI + a_rsub;
        int col = a_c * 4;
        if (a.p_split) {
            const int plane = a_c / (A_LPR / 2), q = a_c % (A_LPR / 2);
            col = 8 * (2 * ((q >> 1) ^ wswz(row)) + (q & 1)) + 4 * plane;
        }
        p_src[i] = P + (size_t)(p_begin + row) * a.p_pstride + ta * BA + col;
    }
    // ---- Q: instruction i fills tile row wave*J_INS + i; this lane owns 4 columns (fp32) or one hi / lo half of 8 columns (split copy) of
    // ONE tap for that row
    int q_cb[J_INS], q_kh[J_INS], q_kw[J_INS];
    bool q_tap_ok[J_INS];
#pragma unroll
    for (int i = 0; i < J_INS; ++i) {
        int jc = lane * 4, extra = 0;
        if (a.q_split) {
            const int plane = lane >> 5, q = lane & 31;
            jc = 8 * (2 * ((q >> 1) ^ wswz(wave * J_INS + i)) + (q & 1));
            extra = 4 * plane;
        }
        const int j0 = tj * BJ + jc;
        const int tap = j0 >> a.log2CB;
        q_cb[i] = (j0 & (a.CBp - 1)) + extra;
        q_tap_ok[i] = tap < a.KH * a.KW;
        q_kh[i] = q_tap_ok[i] ? tap / a.KW : 0;
        q_kw[i] = q_tap_ok[i] ? tap - q_kh[i] * a.KW : 0;
    }
    int q_n[J_INS], q_h[J_INS], q_w[J_INS];
    const int HWp = a.Hp * a.Wp;
#pragma unroll
    for (int i = 0; i < J_INS; ++i) {
        const int p = p_begin + wave * J_INS + i;
        q_n[i] = p / HWp;
        const int rem = p - q_n[i] * HWp;
        q_h[i] = rem / a.Wp;
        q_w[i] = rem - q_h[i] * a.Wp;
    }

    auto issue_tile = [&](int kt, int buf) {
        char *base = smem_raw + buf * BUFB;
        const int pbase = p_begin + kt * BP;
#pragma unroll
        for (int i = 0; i < A_INS; ++i) {
            const int row = (wave * A_INS + i) * A_RPI + a_rsub;
            const float *src = (pbase + row < p_end) ? p_src[i] + (size_t)kt * BP * a.p_pstride : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(base + (wave * A_INS + i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < J_INS; ++i) {
            const int row = wave * J_INS + i;
            const int h = q_h[i] * a.step - a.pad + q_kh[i], w = q_w[i] * a.step - a.pad_w + q_kw[i];
            const bool ok = q_tap_ok[i] && (pbase + row < p_end) && ((unsigned)h < (unsigned)a.Hq) && ((unsigned)w < (unsigned)a.Wq);
            const float *src = ok ? Q + ((size_t)(q_n[i] * a.Hq + h) * a.Wq + w) * a.q_pstride + q_cb[i] : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(base + TAB + row * ROWJ), 16, 0, 0);
            // advance this row's pixel by 32 (mixed radix add, single carries)
            q_w[i] += a.dw;
            const int cw = q_w[i] >= a.Wp;
            q_w[i] -= cw ? a.Wp : 0;
            q_h[i] += a.dh + cw;
            const int chh = q_h[i] >= a.Hp;
            q_h[i] -= chh ? a.Hp : 0;
            q_n[i] += a.dn + chh;
        }
    };

    // in-place split of the rows this wave's own DMA instructions filled: fp32 row -> [hi plane | lo plane], planes in the bf16 kernel's layout
    auto convert_tile = [&](int buf) {
        char *base = smem_raw + buf * BUFB;
        f32x4_t ra[A_INS], rq[J_INS];
        if (!a.p_split) {
#pragma unroll
            for (int i = 0; i < A_INS; ++i) ra[i] = *reinterpret_cast<const f32x4_t *>(base + (wave * A_INS + i) * 1024 + lane * 16);
        }
        if (!a.q_split) {
#pragma unroll
            for (int i = 0; i < J_INS; ++i) rq[i] = *reinterpret_cast<const f32x4_t *>(base + TAB + (wave * J_INS + i) * ROWJ + lane * 16);
        }
        if (!a.p_split) {
#pragma unroll
        for (int i = 0; i < A_INS; ++i) {
            const int row = (wave * A_INS + i) * A_RPI + a_rsub;
            u32x2_t hi, lo;
            x3w_split4(ra[i], a.p_act, hi, lo);
            char *dst = base + row * ROWA + (((a_c >> 2) ^ wswz(row)) << 5) + (a_c & 3) * 8;
            *reinterpret_cast<u32x2_t *>(dst) = hi;
            *reinterpret_cast<u32x2_t *>(dst + ROWA / 2) = lo;
        }
        }
        if (!a.q_split) {
#pragma unroll
        for (int i = 0; i < J_INS; ++i) {
            const int row = wave * J_INS + i;
            u32x2_t hi, lo;
            x3w_split4(rq[i], a.q_act, hi, lo);
            char *dst = base + TAB + row * ROWJ + (((lane >> 2) ^ wswz(row)) << 5) + (lane & 3) * 8;
            *reinterpret_cast<u32x2_t *>(dst) = hi;
            *reinterpret_cast<u32x2_t *>(dst + ROWJ / 2) = lo;
        }
        }
    };

    f32x4_t acc[FA][FJ];
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (nk > 0) {
        issue_tile(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        convert_tile(0);
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue_tile(kt + 1, cur ^ 1);
        const char *Ps = smem_raw + cur * BUFB, *Qs = Ps + TAB;
        bf16x8_t bh[FJ], bl[FJ];
        static_assert(FJ == 4, "fragment count of the explicit waits");
        if constexpr (TRASM) {
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                bh[j] = tr_fragment_plane_asm<ROWJ>(Qs, 0, (wj * PJ) / 16 + j, lane);
                bl[j] = tr_fragment_plane_asm<ROWJ>(Qs + ROWJ / 2, 0, (wj * PJ) / 16 + j, lane);
            }
        } else {
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                bh[j] = tr_fragment_plane<ROWJ>(Qs, 0, (wj * PJ) / 16 + j, lane);
                bl[j] = tr_fragment_plane<ROWJ>(Qs + ROWJ / 2, 0, (wj * PJ) / 16 + j, lane);
            }
        }
#pragma unroll
        for (int half = 0; half < FA / 4; ++half) {
            bf16x8_t ah[4], al[4];
            if constexpr (TRASM) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ah[i] = tr_fragment_plane_asm<ROWA>(Ps, 0, (wa * PA) / 16 + half * 4 + i, lane);
                    al[i] = tr_fragment_plane_asm<ROWA>(Ps + ROWA / 2, 0, (wa * PA) / 16 + half * 4 + i, lane);
                }
                tr_wait4(ah); tr_wait4(al);
                if (half == 0) { tr_wait4(bh); tr_wait4(bl); }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ah[i] = tr_fragment_plane<ROWA>(Ps, 0, (wa * PA) / 16 + half * 4 + i, lane);
                    al[i] = tr_fragment_plane<ROWA>(Ps + ROWA / 2, 0, (wa * PA) / 16 + half * 4 + i, lane);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) acc[half * 4 + i][j] = dl_mfma16(al[i], bh[j], acc[half * 4 + i][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) acc[half * 4 + i][j] = dl_mfma16(ah[i], bl[j], acc[half * 4 + i][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) acc[half * 4 + i][j] = dl_mfma16(ah[i], bh[j], acc[half * 4 + i][j]);
        }
        if (kt + 1 < nk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's own rows of step kt+1 have landed
            convert_tile(cur ^ 1);
        }
        __syncthreads();                                           // (includes lgkmcnt(0): the converted planes are visible; buffer `cur` is free)
    }

    const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            const int jj = tj * BJ + wj * PJ + j * 16 + fr;
            if (jj >= a.J) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ca = ta * BA + wa * PA + i * 16 + fg * 4 + r;
                if (ca < a.CAp) slab[(size_t)ks * a.kstride + (size_t)ca * a.J + jj] = acc[i][j][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------
// Staggered schedule of the 256 x 256 x 32-pixel strict tile (same LDS image, conversion and fragment reads as wgrad_glds_x3_kernel<256>).
// In the one-barrier kernel all eight waves read fragments, multiply, wait for the DMA and convert IN LOCK STEP, so the LDS / VALU work
// (48 transposing reads, 8 + 16 conversion accesses and ~100 VALU ops per wave and step) and the 96 MFMAs per wave serialise: 581 us for the
// ResnetBlock shape against a 3 x 62 us MFMA floor.  Here a K step is two phases -- the wave's lower / upper 64 P channels, 48 MFMAs each --
// with two raw s_barriers per phase, and waves 4-7 run ONE barrier behind waves 0-3 (wave w and w + 4 share a SIMD): while one wave of a
// SIMD multiplies, the other one reads its fragments, issues its DMA or converts its rows of the next tile.
//     phase of step t      LDS / DMA section (before the first barrier)                                           MFMA section
//          0               Q fragments (hi, lo) + P lower fragments of tile t;  DMA of tile t+1 -> other buffer   48 MFMAs
//          1               P upper fragments;  vmcnt(0) (own rows of t+1);  convert them in place;  lgkmcnt(0)    48 MFMAs
// Hazards: the DMA of tile t+1 overwrites the buffer tile t-1 was read from -- the last reads of it (phase 1 of step t-1) are retired by
// that phase's lgkmcnt(0) before its first barrier, which every wave has passed when any wave reaches phase 0 of step t;  the converted
// planes of tile t+1 are complete (lgkmcnt(0)) before the converting wave's phase-1 barrier, two barriers before any wave reads them.
// ------------------------------------------------------------------------------------------------------------------
template <int NOPRIO>
__global__ void __launch_bounds__(512) wgrad_4ph_x3_kernel(const WgradArgs a) {
    constexpr int BA = 256, BJ = 256, BP = 32, NW = 8;
    constexpr int ROWA = BA * 4, ROWJ = BJ * 4;
    constexpr int TAB = BP * ROWA, TJB = BP * ROWJ, BUFB = TAB + TJB;
    constexpr int INS = BP / NW;                                               // 4 P rows + 4 Q rows per wave and K step

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave & 1, wj = wave >> 1;
    const bool grp1 = wave >= 4;
    int bid, ks;
    if (a.xcd_group) {
        const int ntile = gridDim.x;
        const int logical = xcd_remap(blockIdx.y * ntile + blockIdx.x, ntile * gridDim.y);
        ks = logical / ntile;
        bid = logical - ks * ntile;
    } else {
        bid = xcd_remap(blockIdx.x, gridDim.x);
        ks = blockIdx.y;
    }
    const int tj = bid % a.tiles_j, ta = bid / a.tiles_j;
    const int p_begin = ks * a.pchunk;
    const int p_end = min(a.Ptot, p_begin + a.pchunk);
    const int T = (p_end > p_begin) ? (p_end - p_begin + BP - 1) / BP : 0;

    const float *P = reinterpret_cast<const float *>(a.P);
    const float *Q = reinterpret_cast<const float *>(a.Q);
    const float *zero = reinterpret_cast<const float *>(g_wzero_page);

    const float *p_src = P + (size_t)(p_begin + wave * INS) * a.p_pstride + ta * BA + lane * 4;      // row wave*4 + i: + i * p_pstride
    const int j0 = tj * BJ + lane * 4;
    const int q_tap = j0 >> a.log2CB;
    const int q_cb = j0 & (a.CBp - 1);
    const bool q_tap_ok = q_tap < a.KH * a.KW;
    const int q_kh = q_tap_ok ? q_tap / a.KW : 0;
    const int q_kw = q_tap_ok ? q_tap - q_kh * a.KW : 0;
    int q_n[INS], q_h[INS], q_w[INS];
    const int HWp = a.Hp * a.Wp;
#pragma unroll
    for (int i = 0; i < INS; ++i) {
        const int p = p_begin + wave * INS + i;
        q_n[i] = p / HWp;
        const int rem = p - q_n[i] * HWp;
        q_h[i] = rem / a.Wp;
        q_w[i] = rem - q_h[i] * a.Wp;
    }

    auto issue_tile = [&](int kt, int buf) __attribute__((always_inline)) {
        char *base = smem_raw + buf * BUFB;
        const int pbase = p_begin + kt * BP;
#pragma unroll
        for (int i = 0; i < INS; ++i) {
            const int row = wave * INS + i;
            const float *src = (pbase + row < p_end) ? p_src + ((size_t)kt * BP + i) * a.p_pstride : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(base + row * ROWA), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < INS; ++i) {
            const int row = wave * INS + i;
            const int h = q_h[i] * a.step - a.pad + q_kh, w = q_w[i] * a.step - a.pad_w + q_kw;
            const bool ok = q_tap_ok && (pbase + row < p_end) && ((unsigned)h < (unsigned)a.Hq) && ((unsigned)w < (unsigned)a.Wq);
            const float *src = ok ? Q + ((size_t)(q_n[i] * a.Hq + h) * a.Wq + w) * a.q_pstride + q_cb : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(base + TAB + row * ROWJ), 16, 0, 0);
            q_w[i] += a.dw;
            const int cw = q_w[i] >= a.Wp;
            q_w[i] -= cw ? a.Wp : 0;
            q_h[i] += a.dh + cw;
            const int chh = q_h[i] >= a.Hp;
            q_h[i] -= chh ? a.Hp : 0;
            q_n[i] += a.dn + chh;
        }
    };
    auto convert_tile = [&](int buf) __attribute__((always_inline)) {       // two passes (P rows, then Q rows): 16 instead of 32 temporaries
        char *base = smem_raw + buf * BUFB;
#pragma unroll
        for (int op = 0; op < 2; ++op) {
            char *tile = base + op * TAB;            // ROWA == ROWJ (BA = 256)
            const int act = op ? a.q_act : a.p_act;
            f32x4_t r[INS];
#pragma unroll
            for (int i = 0; i < INS; ++i) r[i] = *reinterpret_cast<const f32x4_t *>(tile + (wave * INS + i) * ROWA + lane * 16);
#pragma unroll
            for (int i = 0; i < INS; ++i) {
                const int row = wave * INS + i;
                u32x2_t hi, lo;
                x3w_split4(r[i], act, hi, lo);
                char *dst = tile + row * ROWA + (((lane >> 2) ^ wswz(row)) << 5) + (lane & 3) * 8;
                *reinterpret_cast<u32x2_t *>(dst) = hi;
                *reinterpret_cast<u32x2_t *>(dst + ROWA / 2) = lo;
            }
        }
    };

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (T > 0) {
        issue_tile(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        convert_tile(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    DL_WBAR();
    if (grp1) DL_WBAR();          // stagger: waves 4-7 run one barrier behind

    bf16x8_t bh[4], bl[4], ah[4], al[4];
    auto read_q = [&](const char *Qs) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bh[j] = tr_fragment_plane<ROWJ>(Qs, 0, wj * 4 + j, lane);
            bl[j] = tr_fragment_plane<ROWJ>(Qs + ROWJ / 2, 0, wj * 4 + j, lane);
        }
    };
    auto read_p = [&](const char *Ps, int half) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = tr_fragment_plane<ROWA>(Ps, 0, wa * 8 + half * 4 + i, lane);
            al[i] = tr_fragment_plane<ROWA>(Ps + ROWA / 2, 0, wa * 8 + half * 4 + i, lane);
        }
    };
    auto mma = [&](auto HALF) __attribute__((always_inline)) {
        constexpr int i0 = decltype(HALF)::value * 4;
        if constexpr (!NOPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i0 + i][j] = dl_mfma16(al[i], bh[j], acc[i0 + i][j]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i0 + i][j] = dl_mfma16(ah[i], bl[j], acc[i0 + i][j]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i0 + i][j] = dl_mfma16(ah[i], bh[j], acc[i0 + i][j]);
        if constexpr (!NOPRIO) __builtin_amdgcn_s_setprio(0);
    };

    for (int t = 0; t < T; ++t) {
        const int cur = t & 1;
        const char *Ps = smem_raw + cur * BUFB, *Qs = Ps + TAB;
        const bool more = t + 1 < T;
        // ---- phase 0
        read_q(Qs);
        read_p(Ps, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue_tile(t + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        DL_WBAR();
        mma(IC<0>{});
        __builtin_amdgcn_sched_barrier(0);
        DL_WBAR();
        // ---- phase 1
        read_p(Ps, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's own rows of tile t+1 have landed
            convert_tile(cur ^ 1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // fragments read, planes written: nothing of this wave is pending in LDS
        __builtin_amdgcn_sched_barrier(0);
        DL_WBAR();
        mma(IC<1>{});
        __builtin_amdgcn_sched_barrier(0);
        DL_WBAR();
    }
    if (!grp1) DL_WBAR();         // pairs with the last barrier of the trailing group

    const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int jj = tj * BJ + wj * 64 + j * 16 + fr;
            if (jj >= a.J) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ca = ta * BA + wa * 128 + i * 16 + fg * 4 + r;
                if (ca < a.CAp) a.slab[(size_t)ks * a.kstride + (size_t)ca * a.J + jj] = acc[i][j][r];
            }
        }
}

template <int NOPRIO>
static int launch_wgrad_4ph_x3(WgradArgs a, hipStream_t stream) {
    constexpr size_t smem = (size_t)2 * 32 * (256 + 256) * sizeof(float);
    a.tiles_a = a.CAp / 256;
    a.tiles_j = (a.J + 255) / 256;
    a.pchunk = ((a.Ptot + a.splitk - 1) / a.splitk + 31) / 32 * 32;
    const int hw = a.Hp * a.Wp;
    a.dn = 32 / hw; a.dh = (32 % hw) / a.Wp; a.dw = (32 % hw) % a.Wp;
    auto kern = wgrad_4ph_x3_kernel<NOPRIO>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) DL_FAIL("dl_conv_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.tiles_a * a.tiles_j, a.splitk), dim3(512), smem, stream, a);
    DL_CHECK_LAUNCH("dl_conv_wgrad(4-phase x3)");
    return 0;
}

template <int BA, bool TRASM>
static int launch_wgrad_glds_x3_v(WgradArgs a, const WgradLayers &lay, int n, hipStream_t stream) {
    constexpr size_t smem = (size_t)2 * 32 * (BA + 256) * sizeof(float);
    a.tiles_a = a.CAp / BA;
    a.tiles_j = (a.J + 255) / 256;
    a.pchunk = ((a.Ptot + a.splitk - 1) / a.splitk + 31) / 32 * 32;
    const int hw = a.Hp * a.Wp;
    a.dn = 32 / hw; a.dh = (32 % hw) / a.Wp; a.dw = (32 % hw) % a.Wp;
    auto kern = wgrad_glds_x3_kernel<BA, TRASM>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) DL_FAIL("dl_conv_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.tiles_a * a.tiles_j, a.splitk, n), dim3(512), smem, stream, a, lay);
    DL_CHECK_LAUNCH("dl_conv_wgrad(glds x3)");
    return 0;
}

template <int BA>
static int launch_wgrad_glds_x3(const WgradArgs &a, const WgradLayers &lay, int n, hipStream_t stream) {
    return a.tr_asm ? launch_wgrad_glds_x3_v<BA, true>(a, lay, n, stream) : launch_wgrad_glds_x3_v<BA, false>(a, lay, n, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// Strict policy on the 7x7 stem / head weight gradient (wgrad_c4.h): fp32 operands, split once per staged element when the tile is
// committed to LDS (hi and lo planes of the wide tile and of the small-side patch), three MFMAs per fragment pair, small terms first.
// Tiles are 2 x 64 pixels (the bf16 kernel's 4 x 64 would need 92 KB of LDS for two planes: one workgroup per CU); 49 KB here.
// Same R layout, slab layout and fixed-order reduction (wgrad_c4_reduce_kernel) as the bf16 kernel.
// Before: generic wgrad_kernel<float, 3, ...> on 49 taps x 8 padded channels, stem 864 us, head (shift_stack + stacked) 966 us.
// ------------------------------------------------------------------------------------------------------------------
struct WgradC4X3Args {
    const float *wide;
    const float *small_;
    float *slab;
    int N, H, W, wide_pstride, small_pstride;
    int tiles_w, tiles_h;
};

__global__ void __launch_bounds__(256, 2) wgrad_c4_x3_kernel(const WgradC4X3Args a) {
    constexpr int TR = 2, TC = 64, KR = 7, PR = TR + KR - 1, PW = 72;       // tile, kernel rows, patch rows, patch pitch (pixels)
    constexpr int WROW = 80;                                                // wide-tile row pitch in elements (64 + 16: bank spread)
    constexpr int WIDE_ELEMS = TR * TC * WROW;                              // 20 KB per plane
    constexpr int PATCH_ELEMS = PR * PW * 4 + 32;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t *wth = reinterpret_cast<bf16_t *>(smem_raw);
    bf16_t *wtl = wth + WIDE_ELEMS;
    bf16_t *pth = wtl + WIDE_ELEMS;                                         // patch: [PR][PW] pixels x 4 channels
    bf16_t *ptl = pth + PATCH_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = a.N * a.tiles_h * a.tiles_w;
    const int f0 = (wave * 14) / 4, cnt = ((wave + 1) * 14) / 4 - f0;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    constexpr int WCH = (TR * TC * 16) / 256;               // 16-byte pieces (4 fp32 channels) of the wide tile per thread (8)
    constexpr int PPT = (PR * PW + 255) / 256;              // patch pixels per thread (3)
    f32x4_t wnx[WCH], pnx[PPT];
    auto fetch = [&](int tile) __attribute__((always_inline)) {
        int t = tile;
        const int tw = t % a.tiles_w; t /= a.tiles_w;
        const int th = t % a.tiles_h;
        const int n = t / a.tiles_h;
#pragma unroll
        for (int k = 0; k < WCH; ++k) {
            const int i = tid + k * 256, px = i >> 4, c4 = (i & 15) * 4;
            const int h = th * TR + (px >> 6), w = tw * TC + (px & 63);
            wnx[k] = *reinterpret_cast<const f32x4_t *>(a.wide + ((size_t)(n * a.H + h) * a.W + w) * a.wide_pstride + c4);
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = tid + k * 256;
            const int pr = i / PW, pc = i - pr * PW;
            const int h = th * TR - 3 + pr, w = tw * TC - 3 + pc;
            f32x4_t v = {0.f, 0.f, 0.f, 0.f};
            if (i < PR * PW && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W)
                v = *reinterpret_cast<const f32x4_t *>(a.small_ + ((size_t)(n * a.H + h) * a.W + w) * a.small_pstride);
            pnx[k] = v;
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < WCH; ++k) {
            const int i = tid + k * 256, px = i >> 4, c4 = (i & 15) * 4;
            u32x2_t hi, lo;
            x3w_split4(wnx[k], DL_ACT_NONE, hi, lo);
            *reinterpret_cast<u32x2_t *>(wth + px * WROW + c4) = hi;
            *reinterpret_cast<u32x2_t *>(wtl + px * WROW + c4) = lo;
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = tid + k * 256;
            if (i < PR * PW) {
                u32x2_t hi, lo;
                x3w_split4(pnx[k], DL_ACT_NONE, hi, lo);
                *reinterpret_cast<u32x2_t *>(pth + i * 4) = hi;
                *reinterpret_cast<u32x2_t *>(ptl + i * 4) = lo;
            }
        }
    };

    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                  // the previous tile's fragments have all been read
        commit();
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
#pragma unroll
        for (int kc = 0; kc < TR * 2; ++kc) {             // 32-pixel contraction steps: tile row kc >> 1, columns (kc & 1)*32 .. +32
            const int r = kc >> 1, c = (kc & 1) * 32;
            bf16x8_t ah[4], al[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ah[i] = tr_fragment_rows<WROW>(wth + (r * TC + c) * WROW + i * 16, lane);
                al[i] = tr_fragment_rows<WROW>(wtl + (r * TC + c) * WROW + i * 16, lane);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < cnt) {
                    const int f = f0 + j;
                    const int po = ((r + (f >> 1)) * PW + c + 4 * (f & 1)) * 4;
                    const bf16x8_t bh = tr_fragment_rows<4>(pth + po, lane);
                    const bf16x8_t bl = tr_fragment_rows<4>(ptl + po, lane);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = dl_mfma16(al[i], bh, acc[i][j]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = dl_mfma16(ah[i], bl, acc[i][j]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = dl_mfma16(ah[i], bh, acc[i][j]);
                }
            }
        }
    }
    const int fr = lane & 15, fg = lane >> 4;
    float *o = a.slab + (size_t)blockIdx.x * 64 * 224;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < cnt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[(i * 16 + fg * 4 + q) * 224 + (f0 + j) * 16 + fr] = acc[i][j][q];
            }
        }
}

static int wgrad_c4_x3_form(const dl_wgrad_desc *d) {          // as wgrad_c4_form, for fp32 operands + BF16X3
    const bool off = dl_switch_is_one(DL_SW_NO_WGRAD_C4) || dl_switch(DL_SW_NO_C4_X3) != nullptr;
    if (off || d->dtype != DL_F32 || d->prec != DL_PREC_BF16X3 || d->p_act != DL_ACT_NONE || d->q_act != DL_ACT_NONE || d->p_split || d->q_split) return 0;
    if (d->KH != 7 || d->KW != 7 || d->step != 1 || d->pad != 3 || (d->pad_w >= 0 && d->pad_w != 3) || d->pad_mode != DL_PAD_ZERO || d->stack_kw) return 0;
    if (d->Hp != d->Hq || d->Wp != d->Wq || d->Hp % 4 || d->Wp % 64 || d->splitk != DL_WGRAD_C4_PARTS) return 0;
    if (d->CAp == 64 && d->CA <= 64 && d->CBp == 8 && d->CB <= 4) return 1;
    if (d->CAp == 8 && d->CA <= 4 && d->CBp == 64 && d->CB <= 64) return 2;
    return 0;
}

static int launch_wgrad_c4_x3(const dl_wgrad_desc *d, int form, const void *P, const void *Q, float *grad, float *slab, hipStream_t stream) {
    WgradC4X3Args a;
    a.wide = (const float *)(form == 1 ? P : Q);
    a.small_ = (const float *)(form == 1 ? Q : P);
    a.wide_pstride = form == 1 ? d->p_pstride : d->q_pstride;
    a.small_pstride = form == 1 ? d->q_pstride : d->p_pstride;
    a.slab = slab;
    a.N = d->N; a.H = d->Hp; a.W = d->Wp;
    a.tiles_w = d->Wp / 64; a.tiles_h = d->Hp / 2;
    const int ntiles = a.N * a.tiles_w * a.tiles_h;
    const int parts = ntiles < DL_WGRAD_C4_PARTS ? ntiles : DL_WGRAD_C4_PARTS;
    constexpr size_t smem = (size_t)(2 * 2 * 64 * 80 + 2 * (8 * 72 * 4 + 32)) * sizeof(bf16_t);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad_c4_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) DL_FAIL("dl_conv_wgrad(c4 x3): hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_c4_x3_kernel, dim3(parts), dim3(256), smem, stream, a);
    DL_CHECK_LAUNCH("dl_conv_wgrad(c4 x3)");
    hipLaunchKernelGGL(wgrad_c4_reduce_kernel, dim3(64 * 56 / 8), dim3(256), 0, stream, slab, parts, grad, d->CA, d->CB, form == 1 ? 1 : 0, d->accumulate);
    DL_CHECK_LAUNCH("dl_conv_wgrad(c4 x3 reduce)");
    return 0;
}
