// conv_c4.h -- included by conv_gemm.hip.  Stride-1 convolutions with at most FOUR real input channels and a 7x7 (<= 7 x 8) window:
// the ResnetGenerator stem (networks.py:386-397, 3 -> 64) and the data gradient of its head (:438-443, dL/dy has 3 channels).
//
// The general gather GEMM treats the 3 input channels as 8 (one 16-byte chunk per tap), so 5/8 of every MFMA multiplies padding, and it
// re-stages the input for every tap.  Here a persistent workgroup walks 4 x 64 pixel tiles of the output:
//   * the input patch of a tile -- (4+6) x (64+7) pixels, FOUR channels each (8 bytes) -- is staged ONCE, in two copies: copy B is copy A shifted
//     by one pixel (8 bytes).  The MFMA B fragment of a lane is "two adjacent kernel columns x 4 channels" = 16 contiguous bytes
//     starting at pixel (col + 2*fg + fr); a lane with even fr reads them from copy A, a lane with odd fr from copy B, and both reads
//     are ALIGNED ds_read_b128 with compile-time offsets for every fragment and kernel row;
//   * K = 7 kernel rows x (8 column slots x 4 channels) = 7 MFMA steps of 32 (the 8th slot has zero weights), instead of 49 taps x 8 = 13;
//   * the weights of a wave (32 output channels x 7 steps) live in registers; nothing but the patch ever goes through LDS;
//   * epilogue as in the general kernels: bias, activation, bf16 store, per-(image, channel) statistics of the stored values for the
//     normalisation that follows (dl_conv_stats_chunks protocol: one chunk per tile).
// 4 waves: wave & 1 = which 32 of the tile's 64 output channels, wave >> 1 = which 4 of its 8 rows; 128 accumulator + 56 weight registers.

struct C4Args {
    ConvArgs a;
    int8_t tap_src[8][8];       // [kernel row dh + 3][column slot dw + 3] -> index of the tap in the packed K order, -1 = no such tap (zero)
    int tiles_w, tiles_h;       // tiles per image
    int abl;                    // timing-only ablation bits (DL_C4_ABL): 1 no MFMA loop, 2 no global stores, 4 no patch fetch, 8 no LDS epilogue
};

static bool c4_geometry_ok(const dl_conv_desc *d) {
    static const bool off = DL_DEV_ENV("DL_NO_C4") != nullptr;
    if (off || d->in_act != DL_ACT_NONE) return false;
    if (d->act != DL_ACT_NONE && d->act != DL_ACT_RELU && d->act != DL_ACT_LRELU) return false;
    if (d->n_phase != 1 || d->in_step != 1 || d->out_step != 1 || d->splitk != 1 || d->raw_out) return false;
    if (d->Ci != 8 || d->ci_real < 1 || d->ci_real > 4 || d->in_pstride != 8) return false;
    if (d->Co % 64 || d->Ho != d->Hi || d->Wo != d->Wi || d->Hq != d->Ho || d->Wq != d->Wo || d->Ho % 4 || d->Wo % 64) return false;
    const int nt = d->phase_tap_begin[1] - d->phase_tap_begin[0];
    if (nt < 1 || nt > 49) return false;
    for (int t = 0; t < nt; ++t)
        if (d->tap_dh[t] < -3 || d->tap_dh[t] > 3 || d->tap_dw[t] < -3 || d->tap_dw[t] > 3) return false;
    return true;
}

static bool c4_bf16_eligible(const dl_conv_desc *d) { return d->in_dtype == DL_BF16 && d->prec == DL_PREC_BF16 && c4_geometry_ok(d); }

// strict policy (fp32 activations, split-bf16 x3 products): conv_c4_patch_x3_kernel (conv_x3.h)
static bool c4_x3_eligible(const dl_conv_desc *d) {
    const bool off = dl_switch(DL_SW_NO_C4_X3) != nullptr;       // A/B: the strict stem / head gradient on the general x3 kernels (round 3 before this kernel)
    return !off && d->in_dtype == DL_F32 && d->prec == DL_PREC_BF16X3 && !d->in_split && c4_geometry_ok(d);
}

static bool c4_eligible(const dl_conv_desc *d) { return c4_bf16_eligible(d) || c4_x3_eligible(d); }

template <int PADMODE, int ACT>
__global__ void __launch_bounds__(256, 2) conv_c4_patch_kernel(const C4Args ca) {
    const ConvArgs &a = ca.a;
    constexpr int TR = 4, TC = 64, KR = 7, PR = TR + KR - 1, PW = 72;       // tile, kernel rows, patch rows, patch pitch (pixels)
    constexpr int NF = (TR / 2) * 4;                                        // pixel fragments per wave: TR/2 rows x 4 column groups
    constexpr int COPY = PR * PW * 8;                                       // bytes of one patch copy
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char *pa = smem_raw, *pb = smem_raw + COPY + 16;                        // copy B holds pixel i at byte (i + 1) * 8
    constexpr int OUT_TILE = TR * TC * 64 * 2;                              // [512 px][64 ch] bf16 = 64 KB: the epilogue's transpose buffer, aliases the patch
    float *red = reinterpret_cast<float *>(smem_raw + OUT_TILE);            // [2 row halves][2][64] statistics

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ch = wave & 1, rh = wave >> 1;
    const int co0 = blockIdx.y * 64 + ch * 32;
    const bf16_t *in = reinterpret_cast<const bf16_t *>(a.in);
    const int ntiles = a.N * ca.tiles_h * ca.tiles_w;

    // ---- weights -> registers: A fragment (cf, kk): row = output channel co0 + cf*16 + lane%16, K chunk lane/16 = column slots 2*fg, 2*fg + 1
    const int fr = lane & 15, fg = lane >> 4;
    bf16x8_t wf[2][KR];
#pragma unroll
    for (int cf = 0; cf < 2; ++cf) {
        const bf16_t *wr = a.w_hi + (size_t)(co0 + cf * 16 + fr) * a.w_kstride + a.phase_kbase[0];
#pragma unroll
        for (int kk = 0; kk < KR; ++kk) {
            u32x2_t lo = {0u, 0u}, hi = {0u, 0u};
            // the four fg variants are scalar loads at constant indices, the lane picks its own (dynamic indexing would put the struct in scratch)
            const int t0 = fg == 0 ? ca.tap_src[kk][0] : fg == 1 ? ca.tap_src[kk][2] : fg == 2 ? ca.tap_src[kk][4] : ca.tap_src[kk][6];
            const int t1 = fg == 0 ? ca.tap_src[kk][1] : fg == 1 ? ca.tap_src[kk][3] : fg == 2 ? ca.tap_src[kk][5] : ca.tap_src[kk][7];
            // branch-free: a missing tap reads tap 0 and is zeroed by the select (28 exec-masked loads cost ~160 spilled registers)
            lo = *reinterpret_cast<const u32x2_t *>(wr + (t0 >= 0 ? t0 : 0) * 8);
            hi = *reinterpret_cast<const u32x2_t *>(wr + (t1 >= 0 ? t1 : 0) * 8);
            u32x4_t v = {t0 >= 0 ? lo[0] : 0u, t0 >= 0 ? lo[1] : 0u, t1 >= 0 ? hi[0] : 0u, t1 >= 0 ? hi[1] : 0u};
            wf[cf][kk] = __builtin_bit_cast(bf16x8_t, v);
        }
    }

    // ---- persistent workgroup: the weights stay in registers while it walks tiles blockIdx.x, + gridDim.x, ...; the patch of the NEXT tile is
    // fetched into registers before the MFMA loop of the current one and written to LDS after its epilogue
    constexpr int PPT = (PR * (TC + KR) + 255) / 256;       // patch pixels per thread (4)
    u32x2_t nxt[PPT];
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
            u32x2_t v = {0u, 0u};
            if (ok) v = *reinterpret_cast<const u32x2_t *>(in + ((size_t)(n * a.Hi + hi) * a.Wi + wi) * 8);
            nxt[k] = v;
        }
    };
    auto write_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = tid + k * 256;
            if (i < PR * (TC + KR)) {
                const int pr = i / (TC + KR), pc = i - pr * (TC + KR);
                *reinterpret_cast<u32x2_t *>(pa + (pr * PW + pc) * 8) = nxt[k];
                *reinterpret_cast<u32x2_t *>(pb + (pr * PW + pc + 1) * 8) = nxt[k];
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
    const char *base = ((fr & 1) ? pb + 8 : pa) + (fr + 2 * fg) * 8 + (rh * (TR / 2)) * PW * 8;
    if (!(ca.abl & 1))
#pragma unroll
    for (int kk = 0; kk < KR; ++kk) {
#pragma unroll
        for (int jb = 0; jb < NF; jb += 4) {              // batches of 4 fragments (one tile row): 16 operand registers in flight
            bf16x8_t xf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)                   // fragment jb + j: tile row rh*4 + (jb + j)/4, columns ((jb + j)%4)*16 .. +16
                xf[j] = *reinterpret_cast<const bf16x8_t *>(base + ((((jb + j) >> 2) + kk) * PW + ((jb + j) & 3) * 16) * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][jb + j] = dl_mfma16(wf[0][kk], xf[j], acc[0][jb + j]);
                acc[1][jb + j] = dl_mfma16(wf[1][kk], xf[j], acc[1][jb + j]);
            }
        }
    }

    // ---- epilogue: lane holds channels co0 + cf*16 + fg*4 .. +4 of pixel (row rh*4 + j/4, column (j%4)*16 + fr)
    __syncthreads();                                      // every wave has finished reading the patch: its LDS becomes the output tile
    const bool want_stats = a.stats_part != nullptr;
    float st1[2][4], st2[2][4];
#pragma unroll
    for (int cf = 0; cf < 2; ++cf)
#pragma unroll
        for (int r = 0; r < 4; ++r) st1[cf][r] = st2[cf][r] = 0.f;
    bf16_t *out = reinterpret_cast<bf16_t *>(a.out);
    char *ob[2];
#pragma unroll
    for (int cf = 0; cf < 2; ++cf) {
        const int cc = ch * 32 + cf * 16 + fg * 4;             // channel inside the 64-channel tile
        ob[cf] = smem_raw + ((rh * (TR / 2)) * TC + fr) * 128 + (((cc >> 3) ^ (fr & 7)) << 4) + (cc & 4) * 2;
    }
#pragma unroll
    for (int cf = 0; cf < 2; ++cf) {
        const int co = co0 + cf * 16 + fg * 4;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = (co + r < a.bias_n) ? a.bias[co + r] : 0.f;
        }
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
            u32x2_t p;
            p[0] = pack2_bf16(v[0], v[1]);
            p[1] = pack2_bf16(v[2], v[3]);
            // through LDS: [pixel][64 ch] rows of 128 B, 16-byte chunks XOR-swizzled by the pixel; copied out below as whole rows
            // (a lane owns 4 channels of a pixel: storing straight from the fragments would write 32-byte pieces of 16 different rows).
            // pixel & 7 == fr & 7 for every fragment, so the swizzled chunk is a per-lane constant and the fragment offset an immediate
            if (!(ca.abl & 8)) *reinterpret_cast<u32x2_t *>(ob[cf] + ((j >> 2) * TC + (j & 3) * 16) * 128) = p;
            if (want_stats) {
                const float q0 = h16_lo_f32(p[0]), q1 = h16_hi_f32(p[0]);
                const float q2 = h16_lo_f32(p[1]), q3 = h16_hi_f32(p[1]);
                st1[cf][0] += q0; st2[cf][0] += q0 * q0; st1[cf][1] += q1; st2[cf][1] += q1 * q1;
                st1[cf][2] += q2; st2[cf][2] += q2 * q2; st1[cf][3] += q3; st2[cf][3] += q3 * q3;
            }
        }
    }
    __syncthreads();                                      // output tile complete
#pragma unroll
    for (int k = 0; k < (TR * TC * 8) / 256; ++k) {       // 4096 16-byte chunks: 8 lanes = one pixel's 128-byte row, fully coalesced
        const int i = tid + k * 256;
        const int px = i >> 3, pos = i & 7;
        const u32x4_t v = *reinterpret_cast<const u32x4_t *>(smem_raw + px * 128 + (pos << 4));
        const int c8 = (pos ^ (px & 7)) * 8;
        const int h = h0 + (px >> 6), w = w0 + (px & 63);
        if (!(ca.abl & 2)) *reinterpret_cast<u32x4_t *>(out + ((size_t)(n * a.Ho + h) * a.Wo + w) * a.out_pstride + blockIdx.y * 64 + c8) = v;
    }
    if (want_stats) {
#pragma unroll
        for (int cf = 0; cf < 2; ++cf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { st1[cf][r] += __shfl_xor(st1[cf][r], o, 64); st2[cf][r] += __shfl_xor(st2[cf][r], o, 64); }
            }
        if (fr == 0) {
#pragma unroll
            for (int cf = 0; cf < 2; ++cf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = ch * 32 + cf * 16 + fg * 4 + r;
                    red[(rh * 2 + 0) * 64 + c] = st1[cf][r];
                    red[(rh * 2 + 1) * 64 + c] = st2[cf][r];
                }
        }
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

static void c4_fill_args(C4Args &ca, const ConvArgs &a0, const dl_conv_desc *d) {
    ca.a = a0;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) ca.tap_src[i][j] = -1;
    const int nt = d->phase_tap_begin[1] - d->phase_tap_begin[0];
    for (int t = 0; t < nt; ++t) ca.tap_src[d->tap_dh[t] + 3][d->tap_dw[t] + 3] = (int8_t)t;
    ca.tiles_w = d->Wo / 64;
    ca.tiles_h = d->Ho / 4;
    static const char *abl_env = DL_DEV_ENV("DL_C4_ABL");
    ca.abl = abl_env ? atoi(abl_env) : 0;
}

// launch with the dynamic-LDS attribute set once per instantiation
static int c4_launch(void (*kern)(const C4Args), const C4Args &ca, const dl_conv_desc *d, size_t smem, hipStream_t stream, const char *what) {
    static void (*attr_done[12])(const C4Args) = {};
    bool seen = false;
    for (int i = 0; i < 12; ++i) seen |= attr_done[i] == kern;
    if (!seen) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) DL_FAIL("%s: hipFuncSetAttribute(%zu): %s", what, smem, hipGetErrorString(e));
        for (int i = 0; i < 12; ++i)
            if (!attr_done[i]) { attr_done[i] = kern; break; }
    }
    const int ntiles = d->N * ca.tiles_w * ca.tiles_h;
    const int per_y = 512 / (d->Co / 64) > 0 ? 512 / (d->Co / 64) : 1;       // ~2 workgroups per CU in total
    dim3 grid(ntiles < per_y ? ntiles : per_y, d->Co / 64);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, ca);
    DL_CHECK_LAUNCH(what);
    return 0;
}

static int launch_conv_c4(const ConvArgs &a0, const dl_conv_desc *d, hipStream_t stream) {
    C4Args ca;
    c4_fill_args(ca, a0, d);
    constexpr size_t smem = 4 * 64 * 64 * 2 + 4 * 64 * sizeof(float);          // output tile (aliases the two patch copies) + statistics
    void (*kern)(const C4Args) = nullptr;
    const bool refl = d->pad_mode == DL_PAD_REFLECT;
    switch (d->act) {
        case DL_ACT_RELU: kern = refl ? conv_c4_patch_kernel<DL_PAD_REFLECT, DL_ACT_RELU> : conv_c4_patch_kernel<DL_PAD_ZERO, DL_ACT_RELU>; break;
        case DL_ACT_LRELU: kern = refl ? conv_c4_patch_kernel<DL_PAD_REFLECT, DL_ACT_LRELU> : conv_c4_patch_kernel<DL_PAD_ZERO, DL_ACT_LRELU>; break;
        default: kern = refl ? conv_c4_patch_kernel<DL_PAD_REFLECT, DL_ACT_NONE> : conv_c4_patch_kernel<DL_PAD_ZERO, DL_ACT_NONE>; break;
    }
    return c4_launch(kern, ca, d, smem, stream, "dl_conv_forward(c4 patch)");
}
