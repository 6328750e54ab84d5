// wgrad_c4.h -- included by wgrad.hip.  Weight gradient of the two 7x7 ResnetGenerator layers (networks.py:386-397 stem 3 -> 64,
// :438-443 head 64 -> 3): one operand has 64 channels ("wide", [pixel][64]), the other at most 4 real ones ("small", [pixel][8]).
//
//   R[a][(r, s, c)] = sum over pixels p of  wide[p][a] * small[p + (r - 3, s - 3)][c]        a < 64, r < 7 rows, s < 8 slots, c < 4
//
// is exactly the gradient: stem (wide = dL/dy, small = x):  dW[co = a][ci = c][kh = r][kw = s];
//                          head (wide = x, small = dL/dy):  dW[co = c][ci = a][kh = 6 - r][kw = 6 - s]   (the shift changes sign).
// The general kernels stage the 3-channel side as 8 channels per tap, 49 taps -- 392 contraction-side columns of which 147 are real, and
// re-gather the wide side per tap tile.  Here a persistent workgroup walks 4 x 64 pixel tiles and keeps its whole R (64 x 224 fp32) in
// registers: the wide tile goes to LDS once ([pixel][64], rows padded to 160 B), the (4+6) x (64+8) patch of the small side once
// (4 channels = 8 bytes per pixel), and BOTH MFMA operands come out of LDS with the transposing read ds_read_b64_tr_b16: the A
// fragment is 16 channels x 32 pixels of the wide tile, the B fragment "4 adjacent kernel columns x 4 channels" x 32 pixels is a
// [32][16] matrix whose rows OVERLAP (row stride = one pixel = 8 bytes), which the per-lane addressing of the instruction allows.
// 14 column fragments (7 rows x 2 slot groups) are split 3/4/3/4 over the 4 waves; each wave holds 4 x <=4 accumulator fragments.
// Partial results: slab[workgroup][64][224] fp32, combined in a fixed order by wgrad_c4_reduce_kernel (deterministic).

struct WgradC4Args {
    const bf16_t *wide;
    const bf16_t *small_;
    float *slab;
    int N, H, W, wide_pstride, small_pstride;
    int tiles_w, tiles_h;
    int abl;            // timing-only ablation bits (DL_WC4_ABL): 1 no fragment reads / MFMAs, 2 no prefetch after the first tile, 4 no LDS commit
};

template <int ROW>
__device__ __forceinline__ bf16x8_t tr_fragment_rows(const bf16_t *tile, int lane) {
    // as tr_fragment (wgrad.hip) with c0 = 0: lane (m = lane&15, g = lane>>4) gets tile[8g + 4h + j][m], h = 0,1, j = 0..3
    const int m = lane & 15, g = lane >> 4;
    const bf16_t *p0 = tile + (8 * g + (m >> 2)) * ROW + (m & 3) * 4;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3))) *)(p0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3))) *)(p0 + 4 * ROW));
    bf16x8_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

__global__ void __launch_bounds__(256, 2) wgrad_c4_kernel(const WgradC4Args a) {
    constexpr int TR = 4, TC = 64, KR = 7, PR = TR + KR - 1, PW = 72;       // tile, kernel rows, patch rows, patch pitch (pixels)
    constexpr int WROW = 80;                                                // wide-tile row pitch in elements (64 + 16: bank spread)
    constexpr int WIDE_ELEMS = TR * TC * WROW;                              // 40 KB
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t *wt = reinterpret_cast<bf16_t *>(smem_raw);
    bf16_t *pt = wt + WIDE_ELEMS;                                           // patch: [PR][PW] pixels x 4 channels

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = a.N * a.tiles_h * a.tiles_w;
    // this wave's column fragments f0 .. f0 + cnt - 1 of the 14 (fragment f: kernel row f >> 1, slots 4*(f & 1) .. +4)
    const int f0 = (wave * 14) / 4, cnt = ((wave + 1) * 14) / 4 - f0;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    constexpr int WCH = (TR * TC * 8) / 256;                // 16-byte chunks of the wide tile per thread (8)
    constexpr int PPT = (PR * PW + 255) / 256;              // patch pixels per thread (3)
    u32x4_t wnx[WCH];
    u32x2_t pnx[PPT];
    auto fetch = [&](int tile) __attribute__((always_inline)) {
        int t = tile;
        const int tw = t % a.tiles_w; t /= a.tiles_w;
        const int th = t % a.tiles_h;
        const int n = t / a.tiles_h;
#pragma unroll
        for (int k = 0; k < WCH; ++k) {
            const int i = tid + k * 256, px = i >> 3, c8 = (i & 7) * 8;
            const int h = th * TR + (px >> 6), w = tw * TC + (px & 63);
            wnx[k] = *reinterpret_cast<const u32x4_t *>(a.wide + ((size_t)(n * a.H + h) * a.W + w) * a.wide_pstride + c8);
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = tid + k * 256;
            const int pr = i / PW, pc = i - pr * PW;
            const int h = th * TR - 3 + pr, w = tw * TC - 3 + pc;
            u32x2_t v = {0u, 0u};
            if (i < PR * PW && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W)
                v = *reinterpret_cast<const u32x2_t *>(a.small_ + ((size_t)(n * a.H + h) * a.W + w) * a.small_pstride);
            pnx[k] = v;
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < WCH; ++k) {
            const int i = tid + k * 256, px = i >> 3, c8 = (i & 7) * 8;
            *reinterpret_cast<u32x4_t *>(wt + px * WROW + c8) = wnx[k];
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = tid + k * 256;
            if (i < PR * PW) *reinterpret_cast<u32x2_t *>(pt + i * 4) = pnx[k];
        }
    };

    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                  // the previous tile's fragments have all been read
        if (!(a.abl & 4)) commit();
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles && !(a.abl & 2)) fetch(tile + gridDim.x);
        if (a.abl & 1) continue;
#pragma unroll
        for (int kc = 0; kc < TR * 2; ++kc) {             // 32-pixel contraction steps: tile row kc >> 1, columns (kc & 1)*32 .. +32
            const int r = kc >> 1, c = (kc & 1) * 32;
            bf16x8_t af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = tr_fragment_rows<WROW>(wt + (r * TC + c) * WROW + i * 16, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < cnt) {
                    const int f = f0 + j;
                    // rows of this [32][16] matrix = pixels c .. c+31 of patch row r + (f >> 1), starting at slot 4*(f & 1): row pitch 4 elements
                    const bf16x8_t bf = tr_fragment_rows<4>(pt + ((r + (f >> 1)) * PW + c + 4 * (f & 1)) * 4, lane);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = dl_mfma16(af[i], bf, acc[i][j]);
                }
            }
        }
    }
    // ---- partial result of this workgroup: slab[blockIdx.x][a][n], n = fragment*16 + lane%16 = (r*8 + s)*4 + c
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

// grad[A][B][7][7] (+)= sum over workgroups, fixed order.  wide_is_a: the wide operand's channel is grad's first index (stem: wide = dL/dy
// -> A = co) and the shift keeps its sign; otherwise (head: wide = x -> B = ci) the kernel indices are mirrored.
__global__ void __launch_bounds__(256) wgrad_c4_reduce_kernel(const float *slab, int nparts, float *grad, int CA, int CB, int wide_is_a, int accumulate) {
    // one workgroup = 8 adjacent (a, r, slot) columns of the slab (a float4 of the 4 small-side channels each, 128 contiguous bytes per
    // partial) x 32 slices of the partials: thread (kl, il) adds partials kl, kl + 32, ... of column il in index order with 8 loads in
    // flight, then the 32 slice sums are added in slice order -- a fixed summation tree, so the result does not depend on scheduling
    __shared__ double red[32][8][4];
    const int il = threadIdx.x & 7, kl = threadIdx.x >> 3;
    const int col = blockIdx.x * 8 + il;                            // (a, r, slot), 64 * 7 * 8 columns
    const float4 *p = reinterpret_cast<const float4 *>(slab) + col;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = kl;
    for (; k + 7 * 32 < nparts; k += 8 * 32) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + u * 32) * (64 * 56)];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s0 += (double)v[u].x; s1 += (double)v[u].y; s2 += (double)v[u].z; s3 += (double)v[u].w; }
    }
    for (; k < nparts; k += 32) {
        const float4 v = p[(size_t)k * (64 * 56)];
        s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
    }
    red[kl][il][0] = s0; red[kl][il][1] = s1; red[kl][il][2] = s2; red[kl][il][3] = s3;
    __syncthreads();
    if (threadIdx.x >= 32) return;
    const int c = threadIdx.x & 3, cl = threadIdx.x >> 2;
    const int cc = blockIdx.x * 8 + cl, slot = cc & 7, r = (cc >> 3) % 7, aa = cc / 56;
    const int csm = wide_is_a ? CB : CA;                            // real channels of the small side
    if (slot >= 7 || c >= csm) return;
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < 32; ++q) tot += red[q][cl][c];
    float *g;
    if (wide_is_a) { if (aa >= CA) return; g = grad + (((size_t)aa * CB + c) * 7 + r) * 7 + slot; }
    else { if (aa >= CB) return; g = grad + (((size_t)c * CB + aa) * 7 + (6 - r)) * 7 + (6 - slot); }
    *g = (accumulate ? *g : 0.f) + (float)tot;
}

#define DL_WGRAD_C4_PARTS 512
static int wgrad_c4_form(const dl_wgrad_desc *d) {          // 0: not eligible, 1: P wide / Q small (stem), 2: P small / Q wide (head)
    const bool off = dl_switch_is_one(DL_SW_NO_WGRAD_C4);       // "1" switches the kernel off (same parse as deepliif_amd/ops.py)
    if (off || d->dtype != DL_BF16 || d->prec != DL_PREC_BF16 || d->p_act != DL_ACT_NONE || d->q_act != DL_ACT_NONE) return 0;
    if (d->KH != 7 || d->KW != 7 || d->step != 1 || d->pad != 3 || (d->pad_w >= 0 && d->pad_w != 3) || d->pad_mode != DL_PAD_ZERO || d->stack_kw) return 0;
    if (d->Hp != d->Hq || d->Wp != d->Wq || d->Hp % 4 || d->Wp % 64 || d->splitk != DL_WGRAD_C4_PARTS) return 0;
    if (d->CAp == 64 && d->CA <= 64 && d->CBp == 8 && d->CB <= 4) return 1;
    if (d->CAp == 8 && d->CA <= 4 && d->CBp == 64 && d->CB <= 64) return 2;
    return 0;
}

static int launch_wgrad_c4(const dl_wgrad_desc *d, int form, const void *P, const void *Q, float *grad, float *slab, hipStream_t stream) {
    WgradC4Args a;
    a.wide = (const bf16_t *)(form == 1 ? P : Q);
    a.small_ = (const bf16_t *)(form == 1 ? Q : P);
    a.wide_pstride = form == 1 ? d->p_pstride : d->q_pstride;
    a.small_pstride = form == 1 ? d->q_pstride : d->p_pstride;
    a.slab = slab;
    a.N = d->N; a.H = d->Hp; a.W = d->Wp;
    a.tiles_w = d->Wp / 64; a.tiles_h = d->Hp / 4;
    static const char *abl_env = DL_DEV_ENV("DL_WC4_ABL");
    a.abl = abl_env ? atoi(abl_env) : 0;
    const int ntiles = a.N * a.tiles_w * a.tiles_h;
    const int parts = ntiles < DL_WGRAD_C4_PARTS ? ntiles : DL_WGRAD_C4_PARTS;
    constexpr size_t smem = (size_t)(4 * 64 * 80 + 10 * 72 * 4 + 64) * sizeof(bf16_t);
    if (!(a.abl & 8)) hipLaunchKernelGGL(wgrad_c4_kernel, dim3(parts), dim3(256), smem, stream, a);
    DL_CHECK_LAUNCH("dl_conv_wgrad(c4)");
    if (!(a.abl & 16)) hipLaunchKernelGGL(wgrad_c4_reduce_kernel, dim3(64 * 56 / 8), dim3(256), 0, stream, slab, parts, grad, d->CA, d->CB, form == 1 ? 1 : 0, d->accumulate);
    DL_CHECK_LAUNCH("dl_conv_wgrad(c4 reduce)");
    return 0;
}
