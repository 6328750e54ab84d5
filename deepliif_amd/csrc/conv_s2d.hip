// conv_s2d.hip -- the stride-2 3x3 convolution over a 64-channel tensor with the layer's WEIGHTS HELD IN REGISTERS: ResnetGenerator down1 forward
// (Conv2d(64, 128, k3, s2, p1), networks.py:400-404) and the data gradient of up2 (ConvTranspose2d(128, 64, k3, s2, p1, op1), networks.py:425-436 --
// the gradient of a stride-2 transposed conv is a stride-2 conv over dL/dy): out[n, ho, wo, co] = sum_{kh,kw,ci} x[n, 2ho-1+kh, 2wo-1+kw, ci] W[co, (kh,kw), ci].
//
// Why its own kernel: 77.3 GF over 268 MB of input + 134 MB of output is HBM-bound (~75 us at 5.4 TB/s), but a stride-2 conv reuses an input
// pixel only 2.25 times, so the gather GEMM (conv_gemm_glds_kernel<128,128,64>: 32 KB staged per 64-wide K step for 512 matrix-pipe cycles, input tile AND weight
// tile re-staged for every tap) moves 1.18 GB through the global->LDS path and runs 234 us -- 3x off both roofs.  The weights of this shape are small
// (128 x 576 bf16 = 147 KB): too big to sit in LDS next to the input, but the REGISTER file of a CU holds 512 KB.  So:
//   * four waves, one per SIMD; wave w keeps the MFMA A operands of ITS 32 output channels for all 9 taps x 64 input channels in 144 VGPRs for the whole
//     kernel (36 fragments of v_mfma_f32_32x32x16_bf16) -- no weight byte is staged after the prologue;
//   * a workgroup walks down a strip of R output rows of ONE 128-pixel row segment; every INPUT row segment (257 pixels x 64 channels) is staged exactly
//     once, de-interleaved by pixel parity by the DMA's source addresses (even plane: pixels 2i, odd plane: pixels 2i-1), so the stride-2 gather of tap kw
//     reads CONSECUTIVE 128-byte LDS rows (kw = 0: odd[i], kw = 1: even[i], kw = 2: odd[i+1]) with the usual XOR swizzle of the 16-byte chunks;
//   * an odd input row 2ho+1 feeds output row ho (kh = 2) and ho+1 (kh = 0) from ONE fragment read (two MFMAs per ds_read_b128), an even row 2ho feeds
//     ho (kh = 1): two live accumulator sets (2 x 4 x 16 registers), 144 MFMAs per wave and output row;
//   * pipeline: the two input rows of output row t+1 (66 KB) are in flight while row t is multiplied and stored: a ring of four 33 KB slots, one
//     s_waitcnt vmcnt(0) + barrier per OUTPUT row (the epilogue's stores drain there too);
//   * epilogue per output row: bias / ReLU, bf16, 64-pixel half tiles transposed through 16 KB of LDS, whole 256-byte NHWC pixel rows per 16 lanes;
//     fused per-(image, channel) sum / sum of squares of the stored values, accumulated in registers over the strip: one statistics chunk per workgroup.
// HBM-side traffic: input once (+ one halo row per strip: 33 rows for 16) + output once; LDS-DMA traffic = the same bytes.
// Same descriptor and packed weights as the gather GEMM (n_phase = 1, in_step = 2, taps (kh - 1, kw - 1) kh-major): no host change beyond the dispatch.
#include "conv_args.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((address_space(3))) char lds_char_t;
typedef __attribute__((address_space(3))) const bf16x8_t lds_frag_t;

__device__ __attribute__((aligned(64))) unsigned char g_s2d_zero_page[64];

template <int V> struct S2DIC { static constexpr int value = V; };

struct S2dArgs {
    ConvArgs a;
    int R, nstrips, segs;
};

constexpr int S2D_EVEN = 128 * 128;                 // even plane: input pixels 2i, i = 0..127 (128 B per pixel)
constexpr int S2D_ODD = 136 * 128;                  // odd plane: input pixels 2i - 1, i = 0..135 (129 used; 17 DMA pieces of 8 pixels)
constexpr int S2D_SLOT = S2D_EVEN + S2D_ODD;        // one input row segment
constexpr int S2D_TILE = 4 * S2D_SLOT;              // epilogue half tile: 64 pixels x 128 channels bf16
constexpr int S2D_BIAS = S2D_TILE + 64 * 256;         // the channel tile's bias (128 floats)
constexpr size_t S2D_LDS = (size_t)S2D_BIAS + 512;
static_assert(S2D_LDS <= 160 * 1024, "LDS of one CU");

__device__ __forceinline__ float s2d_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
    return v;
}

// STATS: fused norm statistics (a compile-time flag: as a runtime branch inside the store loop it cost accumulator-sized PHI copies per row)
template <bool STATS, bool RELU>
__global__ void __launch_bounds__(256) conv_s2d_kernel(const S2dArgs sa) {
    const ConvArgs &a = sa.a;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_char_t *lds = (lds_char_t *)smem_raw;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;

    // block -> (image, channel tile, row segment, strip); strips of one segment are neighbours (they share a halo row in L2)
    int b = blockIdx.x;
    const int strip = b % sa.nstrips; b /= sa.nstrips;
    const int seg = b % sa.segs; b /= sa.segs;
    const int tn = b % a.tiles_n;
    const int n = b / a.tiles_n;
    const int ho0 = strip * sa.R;

    // ---- the wave's weights: output channels tn*128 + wave*32 + lr, K = (kh*3 + kw)*64 + s*16 + lh*8 .. +8
    bf16x8_t W[3][3][4];
    {
        const bf16_t *wp = a.w_hi + (size_t)(tn * 128 + wave * 32 + lr) * a.w_kstride + a.phase_kbase[0] + lh * 8;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int s = 0; s < 4; ++s) W[kh][kw][s] = *reinterpret_cast<const bf16x8_t *>(wp + (kh * 3 + kw) * 64 + s * 16);
    }

    // ---- staging geometry: piece = 8 plane pixels x 128 B; wave w issues pieces w, w+4, .. of the even plane (16 pieces) and of the odd plane (17 pieces).
    // LDS rows are 128 B = 8 chunks of 16 B; chunk c of plane pixel pp sits at position c ^ ((pp >> 1) & 7) (applied to the SOURCE address: the DMA image
    // is lane-linear; the permutation does not depend on the piece).  buffer_load ... lds: the image is the buffer (resource in SGPRs), the row and the
    // piece go into the scalar offset, the lane keeps FOUR 32-bit offsets for all 33 pieces; a lane whose pixel lies outside the row (pixel -1 of the
    // image, plane pixels > 128) carries an out-of-range offset and receives zeros -- no 64-bit address arithmetic, no select, next to 144 registers of weights.
    const int lrow = lane >> 3, lcp = lane & 7;
    const char *in = reinterpret_cast<const char *>(a.in);
    const int psb = a.in_pstride * 2;                                             // bytes per input pixel
    const unsigned OOB = 0x80000000u;
    const int pp0 = 8 * wave + lrow;                                              // plane pixel of piece k: pp0 + 32 k
    const unsigned swz = (unsigned)((lcp ^ ((pp0 >> 1) & 7)) * 16);
    // the buffer starts ONE PIXEL in front of the image (never dereferenced there: the only lane that would is out of range by construction), so that
    // every in-range lane offset is non-negative: the range check looks at the lane offset alone, not at lane + scalar offset
    const unsigned o_off = (unsigned)((seg * 256 + 2 * pp0) * psb) + swz;         // odd plane pixel pp0 + 32k = image pixel 2 pp - 1;  + k * 64 * psb
    const unsigned e_off = o_off + (unsigned)psb;                                 // even plane: image pixel 2 pp
    const unsigned o_off0 = (seg * 256 + 2 * pp0 - 1 >= 0) ? o_off : OOB;         // k = 0: pixel -1 of the image row is padding
    const unsigned o_off4 = (lrow == 0) ? o_off : OOB;                            // k = 4 (wave 0): only plane pixel 128 is used
    const size_t row_bytes = (size_t)a.Wi * psb;
    const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(in + (size_t)n * a.Hi * row_bytes - psb), 0, 0x7fffffff, 0x00020000);

    // piece K (0..8) of input row r (soff = r * row_bytes) -> ring slot: K < 4 even plane, 4 <= K < 8 odd plane, K == 8 the 17th odd piece (wave 0 only)
    const __amdgpu_buffer_rsrc_t rsrc_none = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(in), 0, 0, 0x00020000);        // every access out of range
    auto stage_piece = [&](auto Kc, int soff, int slot, const __amdgpu_buffer_rsrc_t rsrc_in) __attribute__((always_inline)) {
        constexpr int K = decltype(Kc)::value;
        const int dst = slot * S2D_SLOT;
        if constexpr (K < 4) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (__attribute__((address_space(3))) void *)(lds + dst + (wave + 4 * K) * 1024), 16, (int)e_off,
                                                     soff + K * 64 * psb, 0, 0);
        } else if constexpr (K < 8) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (__attribute__((address_space(3))) void *)(lds + dst + S2D_EVEN + (wave + 4 * (K - 4)) * 1024), 16,
                                                     (int)(K == 4 ? o_off0 : o_off), soff + (K - 4) * 64 * psb, 0, 0);
        } else {
            if (wave == 0)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (__attribute__((address_space(3))) void *)(lds + dst + S2D_EVEN + 16 * 1024), 16, (int)o_off4,
                                                         soff + 4 * 64 * psb, 0, 0);
        }
    };
    auto stage = [&](int r, int slot) __attribute__((always_inline)) {       // input row r of the image -> ring slot, all pieces at once (prologue)
        const int rowp = r * (int)row_bytes;
        stage_piece(S2DIC<0>{}, rowp, slot, rsrc_in); stage_piece(S2DIC<1>{}, rowp, slot, rsrc_in); stage_piece(S2DIC<2>{}, rowp, slot, rsrc_in);
        stage_piece(S2DIC<3>{}, rowp, slot, rsrc_in); stage_piece(S2DIC<4>{}, rowp, slot, rsrc_in); stage_piece(S2DIC<5>{}, rowp, slot, rsrc_in);
        stage_piece(S2DIC<6>{}, rowp, slot, rsrc_in); stage_piece(S2DIC<7>{}, rowp, slot, rsrc_in); stage_piece(S2DIC<8>{}, rowp, slot, rsrc_in);
    };

    // ---- fragment addressing (bytes inside a slot): lane = (pixel lr of a 32-pixel block, K half lh); + j * 4096 (pixel block), ^ (s << 5) (K sub-step)
    const int a_kw1 = lr * 128 + ((lh ^ ((lr >> 1) & 7)) << 4);                                   // even plane, pixel j*32 + lr
    const int a_kw0 = S2D_EVEN + a_kw1;                                                          // odd plane, pixel j*32 + lr
    const int a_kw2 = S2D_EVEN + (lr + 1) * 128 + ((lh ^ (((lr + 1) >> 1) & 7)) << 4);            // odd plane, pixel j*32 + lr + 1

    f32x16_t accA[4], accB[4];

    // One staged input row = 12 groups g = kw*4 + s of four fragments (pixel blocks j): kernel row KHC into `cur` and / or KHN into `nxt` (-1 = none).
    // The fragments are double-buffered in registers: group g+1 is read while group g is multiplied (one wave per SIMD: nobody else hides the LDS round
    // trip); the first group of a row is read by the caller (`F0` holds it on entry), and the last group's shadow reads group 0 of the NEXT row
    // (slot_next; < 0: none).  DMA piece g (< 9) of the row to prefetch rides in group g's shadow (soff_dma = its byte offset in the image; rsrc_dma = rsrc_none: nothing to prefetch, the pieces fetch nothing).
    auto frag_base = [&](int g_kw) __attribute__((always_inline)) { return g_kw == 0 ? a_kw0 : (g_kw == 1 ? a_kw1 : a_kw2); };
    auto read_group = [&](bf16x8_t (&F)[4], int addr) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) F[j] = *reinterpret_cast<lds_frag_t *>(lds + addr + j * 4096);
    };
    auto row_mac = [&](auto KHC, auto KHN, int slot, int slot_next, f32x16_t (&cur)[4], f32x16_t (&nxt)[4], bf16x8_t (&F0)[4], bf16x8_t (&F1)[4],
                       int soff_dma, int slot_dma, const __amdgpu_buffer_rsrc_t rsrc_dma) __attribute__((always_inline)) {
        constexpr int khc = decltype(KHC)::value, khn = decltype(KHN)::value;
        const int base = slot * S2D_SLOT;
        auto group = [&](auto Gc, bf16x8_t (&Fc)[4], bf16x8_t (&Fn)[4]) __attribute__((always_inline)) {
            constexpr int g = decltype(Gc)::value, kw = g >> 2, sx = g & 3;
            if constexpr (g + 1 < 12) read_group(Fn, base + (frag_base((g + 1) >> 2) ^ (((g + 1) & 3) << 5)));
            else if (slot_next >= 0) read_group(Fn, slot_next * S2D_SLOT + a_kw0);
            if constexpr (g < 9) stage_piece(S2DIC<(g < 9 ? g : 0)>{}, soff_dma, slot_dma, rsrc_dma);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (khc >= 0) cur[j] = dl_mfma32(W[khc < 0 ? 0 : khc][kw][sx], Fc[j], cur[j]);
                if constexpr (khn >= 0) nxt[j] = dl_mfma32(W[khn < 0 ? 0 : khn][kw][sx], Fc[j], nxt[j]);
            }
        };
        group(S2DIC<0>{}, F0, F1); group(S2DIC<1>{}, F1, F0); group(S2DIC<2>{}, F0, F1); group(S2DIC<3>{}, F1, F0);
        group(S2DIC<4>{}, F0, F1); group(S2DIC<5>{}, F1, F0); group(S2DIC<6>{}, F0, F1); group(S2DIC<7>{}, F1, F0);
        group(S2DIC<8>{}, F0, F1); group(S2DIC<9>{}, F1, F0); group(S2DIC<10>{}, F0, F1); group(S2DIC<11>{}, F1, F0);
    };

    // ---- epilogue state.  The accumulators START at the bias (re-read from L1 / L2 after every epilogue: 16 registers that need not live through the
    // main loop next to 144 of weights); the statistics are taken in the STORE pass, where thread t always handles the same 8 channels (16-byte chunk
    // t & 15 of a pixel row): 16 running sums per lane instead of 32.
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    lds_char_t *tile = lds + S2D_TILE;
    const int opb = a.out_pstride * 2;                                            // bytes per output pixel
    const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(a.out) + (size_t)n * a.Ho * a.Wo * opb, 0, 0x7fffffff, 0x00020000);
    const unsigned st_off = (unsigned)((tid >> 4) * opb + (tn * 128 + (tid & 15) * 8) * 2);      // pixel (tid >> 4) + 16 it of a half tile, 16-byte chunk tid & 15
    constexpr bool relu = RELU;                 // (a compile-time flag: as a run-time one it cost a v_max + v_cndmask per stored value)
    __attribute__((address_space(3))) float *bias_l = reinterpret_cast<__attribute__((address_space(3))) float *>(lds + S2D_BIAS);
    if (tid < 128) {
        const int co = tn * 128 + tid;
        bias_l[tid] = (a.bias && co < a.bias_n) ? a.bias[co] : 0.f;
    }
    __syncthreads();
    auto reset_acc = [&](f32x16_t (&acc)[4]) __attribute__((always_inline)) {
        f32x4_t bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = *reinterpret_cast<__attribute__((address_space(3))) const f32x4_t *>(bias_l + wave * 32 + q * 8 + lh * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = bv[r >> 2][r & 3];
    };
    reset_acc(accA);
    reset_acc(accB);

    // acc[j][q*4 + e] = output channel wave*32 + q*8 + lh*4 + e of pixel j*32 + lr (bias included); afterwards acc = bias
    auto epilogue = [&](f32x16_t (&acc)[4], int ho) __attribute__((always_inline)) {
        const int orow = (ho * a.Wo + seg * 128) * opb;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * h + jj;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[j][q * 4 + e];
                        if constexpr (relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    }
                    u32x2_t pk;
                    pk[0] = pack2_bf16(v[0], v[1]);
                    pk[1] = pack2_bf16(v[2], v[3]);
                    const int unit = (wave * 8 + q * 2 + lh) ^ ((lr & 15) << 1);
                    *reinterpret_cast<__attribute__((address_space(3))) u32x2_t *>(tile + (jj * 32 + lr) * 256 + unit * 8) = pk;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // LDS only: the DMA of the next rows and the stores stay in flight
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = tid + it * 256;
                const int row = idx >> 4, cc = idx & 15;         // 16-byte chunk cc (channels cc*8 ..) of half-tile pixel `row`
                const int unit = (cc * 2) ^ ((row & 15) << 1);
                const u32x4_t v = *reinterpret_cast<__attribute__((address_space(3))) const u32x4_t *>(tile + row * 256 + unit * 8);
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_out, (int)st_off, orow + (h * 64 + it * 16) * opb, 0);
                if constexpr (STATS) {                 // statistics of exactly what is stored (bf16-rounded)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = h16_lo_f32(v[e]), hi = h16_hi_f32(v[e]);
                        s1[2 * e] += lo; s2[2 * e] += lo * lo; s1[2 * e + 1] += hi; s2[2 * e + 1] += hi * hi;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // LDS only: the DMA of the next rows and the stores stay in flight
        }
        reset_acc(acc);
    };

    // ---- pipeline.  Output row t of the strip (ho = ho0 + t) needs input rows 2ho (kh 1) and 2ho+1 (kh 2; kh 0 of row ho+1); the strip's first row also
    // needs 2ho0-1 (kh 0; the zero padding row when ho0 = 0).  Row pair t lives in ring slots 2(t & 1), 2(t & 1) + 1; the leading halo row in slot 3.
    // The two rows of pair t+1 are issued piece by piece inside the MFMA stream of pair t; one s_waitcnt vmcnt(0) + barrier per output row.
    const int R = sa.R;
    const int r_first = 2 * ho0 - 1;
    bf16x8_t FA[4], FB[4];
    if (r_first >= 0) stage(r_first, 3);
    stage(2 * ho0, 0);
    stage(2 * ho0 + 1, 1);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (r_first >= 0) {
        read_group(FA, 3 * S2D_SLOT + a_kw0);
        row_mac(S2DIC<0>{}, S2DIC<-1>{}, 3, -1, accA, accB, FA, FB, 0, 2, rsrc_none);      // (the out-of-range pieces write zeros: into slot 2, which is still free)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // slot 3 is about to be refilled
    }
    auto out_row = [&](int t, int s0, f32x16_t (&cur)[4], f32x16_t (&nxt)[4]) __attribute__((always_inline)) {
        const int ho = ho0 + t;
        const bool more = t + 1 < R;
        const int soff = (2 * ho + 2) * (int)row_bytes;
        read_group(FA, s0 * S2D_SLOT + a_kw0);
        const __amdgpu_buffer_rsrc_t rs = more ? rsrc_in : rsrc_none;
        row_mac(S2DIC<1>{}, S2DIC<-1>{}, s0, s0 + 1, cur, nxt, FA, FB, soff, s0 ^ 2, rs);
        // (the strip's last row also feeds `nxt`, which nobody uses: 48 of 2 352 MFMAs per wave instead of a second code path that costs accumulator copies)
        row_mac(S2DIC<2>{}, S2DIC<0>{}, s0 + 1, -1, cur, nxt, FA, FB, soff + (int)row_bytes, (s0 ^ 2) + 1, rs);
        __builtin_amdgcn_sched_barrier(0);       // the row's MFMAs stay in front of the epilogue that reads their accumulators
        epilogue(cur, ho);
        __builtin_amdgcn_sched_barrier(0);
        // vmcnt(8): everything but this row's 8 stores per thread -- i.e. every DMA piece of the two rows staged during the row (issued before the stores; gfx9
        // VMEM operations complete in issue order) -- has landed; the stores drain behind the next row's MFMAs instead of in front of the barrier
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    for (int t = 0; t < R; t += 2) {         // R is even (s2d_strip_rows)
        out_row(t, 0, accA, accB);
        out_row(t + 1, 2, accB, accA);
    }

    if constexpr (STATS) {
        // thread t holds channels (t & 15)*8 .. +8 summed over its pixels; the 16 threads t >> 4 of a channel group are lanes c, c+16, c+32, c+48 of the
        // four waves: two shuffles, then the four waves meet in LDS (the epilogue tile is free after the last barrier)
        __attribute__((address_space(3))) float *red = reinterpret_cast<__attribute__((address_space(3))) float *>(tile);      // [wave][2][128]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t1 = s1[e], t2 = s2[e];
            t1 += __shfl_xor(t1, 16, 64); t2 += __shfl_xor(t2, 16, 64);
            t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
            if (lane < 16) { red[(wave * 2 + 0) * 128 + lane * 8 + e] = t1; red[(wave * 2 + 1) * 128 + lane * 8 + e] = t2; }
        }
        __syncthreads();
        if (tid < 128 && tn * 128 + tid < a.Co) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { t1 += red[(w * 2 + 0) * 128 + tid]; t2 += red[(w * 2 + 1) * 128 + tid]; }
            float *o = a.stats_part + ((size_t)(n * a.stats_nchunks + seg * sa.nstrips + strip) * 2) * a.Co + tn * 128 + tid;
            o[0] = t1;
            o[a.Co] = t2;
        }
    }
}

// strip height: an even divisor of Ho that gives the grid about one workgroup per CU (each strip re-reads one halo row, so taller is cheaper)
static int s2d_strip_rows(const ConvArgs &a) {
    const int segs = a.Wo / 128, tiles_n = a.Co / 128;
    const int per_img = a.N * segs * tiles_n;
    int best = 0;
    for (int R = 2; R <= a.Ho; R += 2) {
        if (a.Ho % R) continue;
        const int wgs = per_img * (a.Ho / R);
        if (best == 0 || wgs >= 240) best = R;          // the tallest strip that still fills the chip; else the smallest even divisor
        if (wgs < 240) break;
    }
    return best;
}

// The layers this kernel serves: one phase, input step 2, the nine taps (kh - 1, kw - 1) kh-major, zero padding, exactly 64 contracted channels, output
// channels a multiple of 128, exact 2x geometry with output rows that are multiples of 128 pixels, bf16, no split-K / raw accumulators / input activation.
bool s2d_eligible(const ConvArgs &a) {
    if (a.n_phase != 1 || a.splitk != 1 || a.raw_out || a.in_step != 2 || a.out_step != 1) return false;
    if (a.phase_tap_begin[1] - a.phase_tap_begin[0] != 9 || a.Ci != 64 || a.Co < 128 || (a.Co & 127)) return false;
    if (a.Hi != 2 * a.Ho || a.Wi != 2 * a.Wo || a.Hq != a.Ho || a.Wq != a.Wo || (a.Wo & 127) || (a.Ho & 1)) return false;
    if (a.pad_mode != DL_PAD_ZERO || a.bn_y != nullptr || a.in_act != DL_ACT_NONE || a.epi_old) return false;
    if (a.act != DL_ACT_NONE && a.act != DL_ACT_RELU) return false;
    for (int t = 0; t < 9; ++t) {
        const int dh = (int)(int8_t)(a.taps[t] & 0xff), dw = (int)(int8_t)((a.taps[t] >> 8) & 0xff);
        if (dh != t / 3 - 1 || dw != t % 3 - 1) return false;
    }
    return s2d_strip_rows(a) > 0;
}

// chunks of fused norm statistics per image: one per workgroup (row segment x strip)
int s2d_stats_chunks(const ConvArgs &a) {
    const int R = s2d_strip_rows(a);
    return R > 0 ? (a.Wo / 128) * (a.Ho / R) : 0;
}

int launch_conv_s2d(const ConvArgs &a0, hipStream_t stream) {
    S2dArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.a = a0;
    ConvArgs &a = sa.a;
    sa.R = s2d_strip_rows(a);
    if (sa.R <= 0) DL_FAIL("dl_conv_forward(s2d): no strip height for Ho=%d", a.Ho);
    sa.nstrips = a.Ho / sa.R;
    sa.segs = a.Wo / 128;
    a.tiles_n = a.Co / 128;
    a.tiles_m = a.N * sa.segs * sa.nstrips;
    if (a.stats_part && a.stats_nchunks != sa.segs * sa.nstrips) DL_FAIL("dl_conv_forward(s2d): statistics chunks %d != %d", a.stats_nchunks, sa.segs * sa.nstrips);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipSuccess;
        const void *fns[4] = {reinterpret_cast<const void *>(conv_s2d_kernel<false, false>), reinterpret_cast<const void *>(conv_s2d_kernel<false, true>),
                              reinterpret_cast<const void *>(conv_s2d_kernel<true, false>), reinterpret_cast<const void *>(conv_s2d_kernel<true, true>)};
        for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)S2D_LDS);
        if (e != hipSuccess) DL_FAIL("dl_conv_forward(s2d): hipFuncSetAttribute(%zu): %s", S2D_LDS, hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(a.tiles_m * a.tiles_n, 1);
    const bool relu = a.act == DL_ACT_RELU;
    if (a.stats_part) {
        if (relu) hipLaunchKernelGGL((conv_s2d_kernel<true, true>), grid, dim3(256), S2D_LDS, stream, sa);
        else hipLaunchKernelGGL((conv_s2d_kernel<true, false>), grid, dim3(256), S2D_LDS, stream, sa);
    } else {
        if (relu) hipLaunchKernelGGL((conv_s2d_kernel<false, true>), grid, dim3(256), S2D_LDS, stream, sa);
        else hipLaunchKernelGGL((conv_s2d_kernel<false, false>), grid, dim3(256), S2D_LDS, stream, sa);
    }
    DL_CHECK_LAUNCH("dl_conv_forward(s2d)");
    return 0;
}
