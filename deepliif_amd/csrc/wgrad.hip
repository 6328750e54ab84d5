// wgrad.hip -- weight gradient of Conv2d / ConvTranspose2d on MFMA (gfx950), see include/deepliif_hip.h dl_conv_wgrad.
//
// GEMM view:  R[ca][j] = sum_p P[p][ca] * Q[gather(p, tap(j))][cb(j)],   j = tap * CBp + cb,   p = (n, hp, wp).
// The contraction index is the PIXEL, but both operands are stored pixel-major (NHWC): the 8 consecutive k values an
// mfma_f32_16x16x32_bf16 lane needs are 8 different pixels of one channel.  The tiles are staged in LDS exactly as they
// lie in memory ([pixel][channel], rows padded by 32 B against bank conflicts) and the fragments are fetched with
// gfx950's transposing LDS read ds_read_b64_tr_b16 (two per fragment).  A and B fragments use the same
// (lane group, element) -> pixel mapping, which is all the MFMA contraction requires.
// Split-K over pixel ranges writes fp32 slabs; a second kernel combines them in a fixed order (deterministic) and
// scatters into the parameter-gradient layout [a][b][kh][kw].
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

struct WgradArgs {
    const void *P;
    const void *Q;
    float *slab;
    int kstride;                // floats between the slabs of consecutive pixel ranges: CAp * J + wgrad_slab_pad()
    int N, Hp, Wp, CAp, p_pstride;
    int Hq, Wq, CBp, log2CB, q_pstride;
    int KH, KW, step, pad, pad_w, pad_mode;
    int J;              // KH*KW*CBp
    int Ptot;           // N*Hp*Wp
    int splitk, pchunk; // pixels per split (multiple of 32)
    int tiles_a, tiles_j;
    int dn, dh, dw;     // 32 pixels in mixed radix (Hp*Wp, Wp, 1)
    int p_act, q_act;
    int xcd_group;      // 1: tiles of one pixel range share an XCD (see the kernels)
    int p_split, q_split;   // strict kernels (wgrad_x3.h): the operand is the producer-written split copy
    int tr_asm;             // 1 (default): transposing LDS reads as inline assembly (no hidden vmcnt(0), see tr_fragment_swz_asm); DL_WGRAD_TR_ASM=0: the builtin (A/B)
};

// Operands of a BATCH of same-shaped layers (dl_conv_wgrad_multi): the direct-to-LDS kernels take the layer from blockIdx.z (wgrad_w4_kernel: from its
// flattened grid) and its pointers from this table, which travels in the kernel-argument segment (no device table to build or to keep alive).
// A single-layer launch fills entry 0.
constexpr int WGRAD_MULTI_MAX = DL_WGRAD_MULTI_MAX;
struct WgradLayers {
    const void *P[WGRAD_MULTI_MAX];
    const void *Q[WGRAD_MULTI_MAX];
    float *slab[WGRAD_MULTI_MAX];
};

__device__ __forceinline__ int reflect_idx_w(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

template <typename T> struct RawW;
template <> struct RawW<bf16_t> { u32x4_t v; };
template <> struct RawW<float> { f32x4_t a, b; };

template <typename T, int PREC>
__device__ __forceinline__ void raww_to_planes(const RawW<T> &r, int act, u32x4_t &hi, u32x4_t &lo) {
    float f[8];
    if constexpr (sizeof(T) == 2) {
        if (PREC == 1 && act == DL_ACT_NONE) { hi = r.v; return; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[2 * i] = h16_lo_f32(r.v[i]); f[2 * i + 1] = h16_hi_f32(r.v[i]); }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[i] = r.a[i]; f[4 + i] = r.b[i]; }
    }
    if (act != DL_ACT_NONE) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = apply_act(act, f[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bf16_t h0 = f32_to_bf16(f[2 * i]), h1 = f32_to_bf16(f[2 * i + 1]);
        hi[i] = (uint32_t)h0 | ((uint32_t)h1 << 16);
        if constexpr (PREC == 3) lo[i] = pack2_bf16(f[2 * i] - bf16_to_f32(h0), f[2 * i + 1] - bf16_to_f32(h1));
    }
}

template <typename T> __device__ __forceinline__ void raww_load(RawW<T> &r, const T *p);
template <> __device__ __forceinline__ void raww_load<bf16_t>(RawW<bf16_t> &r, const bf16_t *p) { r.v = *reinterpret_cast<const u32x4_t *>(p); }
template <> __device__ __forceinline__ void raww_load<float>(RawW<float> &r, const float *p) {
    r.a = *reinterpret_cast<const f32x4_t *>(p);
    r.b = *reinterpret_cast<const f32x4_t *>(p + 4);
}
template <typename T> __device__ __forceinline__ void raww_zero(RawW<T> &r);
template <> __device__ __forceinline__ void raww_zero<bf16_t>(RawW<bf16_t> &r) { r.v = u32x4_t{0, 0, 0, 0}; }
template <> __device__ __forceinline__ void raww_zero<float>(RawW<float> &r) { r.a = f32x4_t{0.f, 0.f, 0.f, 0.f}; r.b = r.a; }

// transposing fragment fetch: tile is [32 pixels][ROW bf16]; returns for lane (i = lane&15, g = lane>>4) the 8 values
// tile[8g + 4h + j][c0 + i], h = 0,1, j = 0..3.
template <int ROW>
__device__ __forceinline__ bf16x8_t tr_fragment(const bf16_t *tile, int c0, int lane) {
    const int m = lane & 15, g = lane >> 4;
    const bf16_t *p0 = tile + (8 * g + (m >> 2)) * ROW + c0 + (m & 3) * 4;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3))) *)(p0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3))) *)(p0 + 4 * ROW));
    bf16x8_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// BA: tile rows (channels of P), BJ = 128 columns (tap, cb); waves WA x WJ
template <typename T, int PREC, int BA, int WA, int WJ>
__global__ void __launch_bounds__(256) wgrad_kernel(const WgradArgs a) {
    constexpr int BJ = 128, BP = 32;
    constexpr int PA = BA / WA, PJ = BJ / WJ, FA = PA / 16, FJ = PJ / 16;
    constexpr int ROWA = BA + 16, ROWJ = BJ + 16;          // padded LDS rows (elements)
    constexpr int NPL = (PREC == 3) ? 2 : 1;
    constexpr int TA = BP * ROWA, TJ = BP * ROWJ;          // elements per tile plane
    constexpr int BUF = NPL * (TA + TJ);
    constexpr int CPA = BA / 8;                            // chunks per pixel row of P
    constexpr int P_CH = (BP * CPA + 255) / 256;
    constexpr int Q_CH = (BP * 16) / 256;                  // = 2
    static_assert(WA * WJ == 4, "4 waves");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t *smem = reinterpret_cast<bf16_t *>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave % WA, wj = wave / WA;
    // All (tap, channel) tiles of ONE pixel range read the same P / Q pixels: keep them on one XCD (shared L2) by remapping the
    // flattened workgroup id (hardware deals consecutive ids round-robin over the 8 XCDs); a.xcd_group = 0 keeps the 2-D order.
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
    const int nk = (p_end > p_begin) ? (p_end - p_begin + BP - 1) / BP : 0;

    const T *P = reinterpret_cast<const T *>(a.P);
    const T *Q = reinterpret_cast<const T *>(a.Q);

    // ---- Q chunk geometry: column chunk fixed per thread, two pixel rows (tid/16 and 16 + tid/16)
    const int qcol = tid & 15;
    const int j0 = tj * BJ + qcol * 8;
    const int tap = j0 >> a.log2CB;
    const int cb = j0 & (a.CBp - 1);
    const bool tap_ok = tap < a.KH * a.KW;
    const int kh = tap_ok ? tap / a.KW : 0, kw = tap_ok ? tap - (tap / a.KW) * a.KW : 0;
    int qn[Q_CH], qh[Q_CH], qw[Q_CH];
    const int HWp = a.Hp * a.Wp;
#pragma unroll
    for (int i = 0; i < Q_CH; ++i) {
        const int p = p_begin + (tid >> 4) + i * 16;
        qn[i] = p / HWp;
        const int rem = p - qn[i] * HWp;
        qh[i] = rem / a.Wp;
        qw[i] = rem - qh[i] * a.Wp;
    }

    RawW<T> pr[P_CH], qr[Q_CH];

    auto load_tile = [&](int kt) {
        const int pbase = p_begin + kt * BP;
#pragma unroll
        for (int i = 0; i < P_CH; ++i) {
            const int q = tid + i * 256;
            const int prow = q / CPA, ch = (q % CPA) * 8;
            const int p = pbase + prow;
            const int ca = ta * BA + ch;
            if (q < BP * CPA && p < p_end && ca < a.CAp) raww_load<T>(pr[i], P + (size_t)p * a.p_pstride + ca);
            else raww_zero<T>(pr[i]);
        }
#pragma unroll
        for (int i = 0; i < Q_CH; ++i) {
            const int p = pbase + (tid >> 4) + i * 16;
            int h = qh[i] * a.step - a.pad + kh, w = qw[i] * a.step - a.pad_w + kw;
            bool ok = tap_ok && p < p_end;
            if (a.pad_mode == DL_PAD_REFLECT) { h = reflect_idx_w(h, a.Hq); w = reflect_idx_w(w, a.Wq); }
            else ok = ok && ((unsigned)h < (unsigned)a.Hq) && ((unsigned)w < (unsigned)a.Wq);
            if (ok) raww_load<T>(qr[i], Q + ((size_t)(qn[i] * a.Hq + h) * a.Wq + w) * a.q_pstride + cb);
            else raww_zero<T>(qr[i]);
            // advance this row by 32 pixels (mixed radix add with single carries)
            qw[i] += a.dw;
            const int cw = qw[i] >= a.Wp;
            qw[i] -= cw ? a.Wp : 0;
            qh[i] += a.dh + cw;
            const int chh = qh[i] >= a.Hp;
            qh[i] -= chh ? a.Hp : 0;
            qn[i] += a.dn + chh;
        }
    };
    auto store_tile = [&](int buf) {
        bf16_t *base = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < P_CH; ++i) {
            const int q = tid + i * 256;
            if (q < BP * CPA) {
                const int prow = q / CPA, ch = (q % CPA) * 8;
                u32x4_t hi, lo;
                raww_to_planes<T, PREC>(pr[i], a.p_act, hi, lo);
                *reinterpret_cast<u32x4_t *>(base + prow * ROWA + ch) = hi;
                if constexpr (PREC == 3) *reinterpret_cast<u32x4_t *>(base + TA + prow * ROWA + ch) = lo;
            }
        }
#pragma unroll
        for (int i = 0; i < Q_CH; ++i) {
            const int prow = (tid >> 4) + i * 16;
            u32x4_t hi, lo;
            raww_to_planes<T, PREC>(qr[i], a.q_act, hi, lo);
            *reinterpret_cast<u32x4_t *>(base + NPL * TA + prow * ROWJ + qcol * 8) = hi;
            if constexpr (PREC == 3) *reinterpret_cast<u32x4_t *>(base + NPL * TA + TJ + prow * ROWJ + qcol * 8) = lo;
        }
    };

    f32x4_t acc[FA][FJ];
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (nk > 0) { load_tile(0); store_tile(0); }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) load_tile(kt + 1);
        const bf16_t *base = smem + cur * BUF;
        const bf16_t *Ps = base, *Qs = base + NPL * TA;
        bf16x8_t af[NPL][FA], bf[NPL][FJ];
#pragma unroll
        for (int i = 0; i < FA; ++i) {
            af[0][i] = tr_fragment<ROWA>(Ps, wa * PA + i * 16, lane);
            if constexpr (PREC == 3) af[1][i] = tr_fragment<ROWA>(Ps + TA, wa * PA + i * 16, lane);
        }
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            bf[0][j] = tr_fragment<ROWJ>(Qs, wj * PJ + j * 16, lane);
            if constexpr (PREC == 3) bf[1][j] = tr_fragment<ROWJ>(Qs + TJ, wj * PJ + j * 16, lane);
        }
#pragma unroll
        for (int i = 0; i < FA; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                if constexpr (PREC == 3) {
                    acc[i][j] = dl_mfma16(af[1][i], bf[0][j], acc[i][j]);
                    acc[i][j] = dl_mfma16(af[0][i], bf[1][j], acc[i][j]);
                }
                acc[i][j] = dl_mfma16(af[0][i], bf[0][j], acc[i][j]);
            }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: slab[ks][ca][j]; lane: column j = .. + (lane & 15), rows ca = .. + (lane>>4)*4 + r
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
                if (ca < a.CAp) a.slab[(size_t)ks * a.kstride + (size_t)ca * a.J + jj] = acc[i][j][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 fast path (v2): 8 waves, tile BA x 256 x 64 pixels, both tiles HBM -> LDS by global_load_lds_dwordx4.
// LDS rows are unpadded (the DMA image must be lane-linear); ds_read_b64_tr_b16 bank conflicts are removed by an XOR
// swizzle of the 32-byte slots of a row, slot' = slot ^ g(p) with g(p) = (p & 3) | ((p >> 1) & 4): the 8 pixel rows a
// half-wave touches in one transposing read then hit 8 different 32-byte bank groups.  The swizzle is applied on the
// SOURCE side of the DMA (each lane fetches the global chunk that belongs in its LDS slot) and again in the reads.
// ------------------------------------------------------------------------------------------------------------------
extern __device__ unsigned char g_wzero_page[];
__device__ __attribute__((aligned(64))) unsigned char g_wzero_page[64];

__device__ __forceinline__ int wswz(int p) { return (p & 3) | ((p >> 1) & 4); }

template <int ROWB>
__device__ __forceinline__ bf16x8_t tr_fragment_swz(const bf16_t *tile, int prow0, int slot, int lane) {
    // rows prow0 + 8g + 4h + (m>>2); 16-channel block `slot` (32 B); returns tile[prow0 + 8g + 4h + j][16*slot + i]
    const int m = lane & 15, g = lane >> 4;
    const int x = (m >> 2) | ((g & 1) << 2);             // = wswz(p) for every p this lane reads (independent of h)
    const char *base = reinterpret_cast<const char *>(tile) + (prow0 + 8 * g + (m >> 2)) * ROWB + ((slot ^ x) << 5) + (m & 3) * 8;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3))) *)(base));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3))) *)(base + 4 * ROWB));
    bf16x8_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// The same fragment fetch as INLINE ASSEMBLY (round 5).  Behind the builtin, hipcc's waitcnt pass puts an s_waitcnt vmcnt(0) in front of every transposing
// read that follows an LDS-DMA instruction (the builtin's memory operand carries no alias information, so the read "may touch" what the DMA is writing;
// plain ds_read_b128 loads are not treated this way).  In wgrad_glds_kernel / wgrad_glds_x3_kernel the DMA of tile t+1 is issued in front of the reads of
// tile t: every K step therefore WAITED for the next tile's DMA before it multiplied the current one -- the prefetch never overlapped anything, and the
// K step cost DMA latency + MFMA time (ResnetBlock shape: 142 us for 75 us of matrix work; rounds 2-4 read this as "bound by the global->LDS path").
// The compiler does not count these reads in lgkmcnt: the caller waits explicitly (tr_wait*, tied to the fragment registers so that the MFMAs stay
// behind the wait).  The "memory" clobber keeps the reads ordered against the barriers and the in-place conversion stores of the strict kernel.
template <int ROWB>
__device__ __forceinline__ bf16x8_t tr_fragment_swz_asm(const bf16_t *tile, int prow0, int slot, int lane) {
    const int m = lane & 15, g = lane >> 4;
    const int x = (m >> 2) | ((g & 1) << 2);
    const char *base = reinterpret_cast<const char *>(tile) + (prow0 + 8 * g + (m >> 2)) * ROWB + ((slot ^ x) << 5) + (m & 3) * 8;
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)base;
    s16x4_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(4 * ROWB) : "memory");
    bf16x8_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
__device__ __forceinline__ void tr_wait4(bf16x8_t (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : : "memory");
}
__device__ __forceinline__ void tr_wait8(bf16x8_t (&f)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : : "memory");
}

template <int BA, int WA, int WJ, bool TRASM>
__global__ void __launch_bounds__(512) wgrad_glds_kernel(const WgradArgs a, const WgradLayers lay) {
    constexpr int BJ = 256, BP = 64, NW = 8;
    constexpr int PA = BA / WA, PJ = BJ / WJ, FA = PA / 16, FJ = PJ / 16;
    constexpr int ROWA = BA * 2, ROWJ = BJ * 2;             // row bytes
    constexpr int TA = BP * BA, TJ = BP * BJ, BUF = TA + TJ; // elements
    constexpr int A_RPI = 1024 / ROWA, J_RPI = 1024 / ROWJ; // tile rows per wave-instruction
    constexpr int A_INS = BP / (NW * A_RPI), J_INS = BP / (NW * J_RPI);
    static_assert(WA * WJ == NW, "8 waves");
    static_assert(A_INS >= 1 && J_INS >= 1, "tile too small for 8 waves");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t *smem = reinterpret_cast<bf16_t *>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave % WA, wj = wave / WA;
    // All (tap, channel) tiles of ONE pixel range read the same P / Q pixels: keep them on one XCD (shared L2) by remapping the
    // flattened workgroup id (hardware deals consecutive ids round-robin over the 8 XCDs); a.xcd_group = 0 keeps the 2-D order.
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

    const bf16_t *P = reinterpret_cast<const bf16_t *>(lay.P[layer]);
    const bf16_t *Q = reinterpret_cast<const bf16_t *>(lay.Q[layer]);
    float *slab = lay.slab[layer];
    const bf16_t *zero = reinterpret_cast<const bf16_t *>(g_wzero_page);

    // ---- P lanes: instruction i of this wave fills tile rows (wave*A_INS + i)*A_RPI + lane*16/ROWA
    const bf16_t *p_src[A_INS];
    int p_row[A_INS];
#pragma unroll
    for (int i = 0; i < A_INS; ++i) {
        const int row = (wave * A_INS + i) * A_RPI + (lane * 16) / ROWA;
        const int cpos = ((lane * 16) % ROWA) / 16;                         // 16-byte position inside the LDS row
        const int c = (((cpos >> 1) ^ wswz(row)) << 1) | (cpos & 1);        // global chunk that belongs there
        p_row[i] = row;
        p_src[i] = P + (size_t)(p_begin + row) * a.p_pstride + ta * BA + c * 8;
    }
    // ---- Q lanes: fixed (tap, channel chunk) per lane and instruction, pixel walks by 64 per K step
    int q_row[J_INS], q_n[J_INS], q_h[J_INS], q_w[J_INS], q_kh[J_INS], q_kw[J_INS], q_cb[J_INS];
    bool q_tap_ok[J_INS];
    const int HWp = a.Hp * a.Wp;
#pragma unroll
    for (int i = 0; i < J_INS; ++i) {
        const int row = (wave * J_INS + i) * J_RPI + (lane * 16) / ROWJ;
        const int cpos = ((lane * 16) % ROWJ) / 16;
        const int c = (((cpos >> 1) ^ wswz(row)) << 1) | (cpos & 1);
        const int j0 = tj * BJ + c * 8;
        const int tap = j0 >> a.log2CB;
        q_row[i] = row;
        q_cb[i] = j0 & (a.CBp - 1);
        q_tap_ok[i] = tap < a.KH * a.KW;
        q_kh[i] = q_tap_ok[i] ? tap / a.KW : 0;
        q_kw[i] = q_tap_ok[i] ? tap - q_kh[i] * a.KW : 0;
        const int p = p_begin + row;
        q_n[i] = p / HWp;
        const int rem = p - q_n[i] * HWp;
        q_h[i] = rem / a.Wp;
        q_w[i] = rem - q_h[i] * a.Wp;
    }

    auto issue_tile = [&](int kt, int buf) {
        bf16_t *base = smem + buf * BUF;
        const int pbase = p_begin + kt * BP;
#pragma unroll
        for (int i = 0; i < A_INS; ++i) {
            const bf16_t *src = (pbase + p_row[i] < p_end) ? p_src[i] + (size_t)kt * BP * a.p_pstride : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(base + (wave * A_INS + i) * A_RPI * BA), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < J_INS; ++i) {
            const int h = q_h[i] * a.step - a.pad + q_kh[i], w = q_w[i] * a.step - a.pad_w + q_kw[i];
            const bool ok = q_tap_ok[i] && (pbase + q_row[i] < p_end) && ((unsigned)h < (unsigned)a.Hq) && ((unsigned)w < (unsigned)a.Wq);
            const bf16_t *src = ok ? Q + ((size_t)(q_n[i] * a.Hq + h) * a.Wq + w) * a.q_pstride + q_cb[i] : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(base + TA + (wave * J_INS + i) * J_RPI * BJ), 16, 0, 0);
            // advance this lane's pixel by 64 (mixed radix add, single carries)
            q_w[i] += a.dw;
            const int cw = q_w[i] >= a.Wp;
            q_w[i] -= cw ? a.Wp : 0;
            q_h[i] += a.dh + cw;
            const int chh = q_h[i] >= a.Hp;
            q_h[i] -= chh ? a.Hp : 0;
            q_n[i] += a.dn + chh;
        }
    };

    f32x4_t acc[FA][FJ];
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (nk > 0) issue_tile(0, 0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue_tile(kt + 1, cur ^ 1);
        const bf16_t *Ps = smem + cur * BUF, *Qs = Ps + TA;
#pragma unroll
        for (int ss = 0; ss < BP / 32; ++ss) {
            bf16x8_t af[FA], bf[FJ];
            static_assert(FJ == 4 && (FA == 8 || FA == 4), "fragment counts of the explicit waits");
            if constexpr (TRASM) {
#pragma unroll
                for (int i = 0; i < FA; ++i) af[i] = tr_fragment_swz_asm<ROWA>(Ps, ss * 32, (wa * PA) / 16 + i, lane);
#pragma unroll
                for (int j = 0; j < FJ; ++j) bf[j] = tr_fragment_swz_asm<ROWJ>(Qs, ss * 32, (wj * PJ) / 16 + j, lane);
                if constexpr (FA == 8) tr_wait8(af); else tr_wait4(*reinterpret_cast<bf16x8_t (*)[4]>(&af[0]));
                tr_wait4(bf);
            } else {
#pragma unroll
                for (int i = 0; i < FA; ++i) af[i] = tr_fragment_swz<ROWA>(Ps, ss * 32, (wa * PA) / 16 + i, lane);
#pragma unroll
                for (int j = 0; j < FJ; ++j) bf[j] = tr_fragment_swz<ROWJ>(Qs, ss * 32, (wj * PJ) / 16 + j, lane);
            }
#pragma unroll
            for (int i = 0; i < FA; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) acc[i][j] = dl_mfma16(af[i], bf[j], acc[i][j]);
        }
        __syncthreads();
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
// Four-phase schedule of the 256 x 256 x 64 weight-gradient tile (the ResnetBlock shape; same tile, LDS image and fragment reads as
// wgrad_glds_kernel<256, 2, 4> above, different schedule -- the one conv_gemm_8ph_kernel uses for the forward / data gradient):
//   * a K step (64 pixels) is cut into four phases of 16 MFMAs: (pixels 0-31 | 32-63) x (the wave's lower | upper 64 of its 128 P channels),
//     two raw s_barriers per phase; waves 4-7 run ONE barrier behind waves 0-3 (wave w and w + 4 share a SIMD), so on every SIMD one wave
//     multiplies while the other one issues its transposing LDS reads and its DMA instead of all eight waves doing each in lock step;
//   * the operands of a K step are two 32-pixel halves H0, H1 (P and Q rows, 32 KB each) that are dead after phases 1 and 3: a half is
//     refilled TWO phases after its last read (the reads are only known to be complete once the MFMAs that consume them have run, i.e. at
//     the reading phase's second barrier; forcing them complete before its first barrier with lgkmcnt(0) exposed the LDS latency in
//     every phase: 225 vs 192 us), five phases ahead of its next use, and never waited for with vmcnt(0) in the steady state:
//         phase of step t    reads                                      stages (4 DMA instructions per wave), then waits
//               0            Q(H0): 4 fragments, P(H0) lower: 4         -
//               1            P(H0) upper: 4                             H1 of step t+1 (other buffer);  vmcnt(8): H1(t) has landed
//               2            Q(H1): 4, P(H1) lower: 4                   -
//               3            P(H1) upper: 4                             H0 of step t+2 (this buffer);   vmcnt(8): H0(t+1) has landed
// ------------------------------------------------------------------------------------------------------------------
#define DL_WBAR() asm volatile("s_barrier" ::: "memory")
template <int V> struct IC { static constexpr int value = V; };
__global__ void __launch_bounds__(512) wgrad_8ph_kernel(const WgradArgs a) {
    constexpr int BA = 256, BJ = 256, BP = 64, HP = 32;
    constexpr int ROWA = BA * 2, ROWJ = BJ * 2;             // row bytes
    constexpr int TA = BP * BA, TJ = BP * BJ, BUF = TA + TJ; // elements

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t *smem = reinterpret_cast<bf16_t *>(smem_raw);

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

    const bf16_t *P = reinterpret_cast<const bf16_t *>(a.P);
    const bf16_t *Q = reinterpret_cast<const bf16_t *>(a.Q);
    const bf16_t *zero = reinterpret_cast<const bf16_t *>(g_wzero_page);

    // ---- staging geometry: instruction (h, i) of this wave fills rows h*32 + wave*4 + i*2 + {0, 1} of the P and of the Q tile
    const bf16_t *p_src[2][2];
    int p_row[2][2];
    int q_row[2][2], q_n[2][2], q_h[2][2], q_w[2][2], q_kh[2][2], q_kw[2][2], q_cb[2][2];
    bool q_tap_ok[2][2];
    const int HWp = a.Hp * a.Wp;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = h * HP + wave * 4 + i * 2 + (lane >> 5);          // 32 lanes x 16 B = one 512-byte row
            const int cpos = lane & 31;                                       // 16-byte position inside the LDS row
            const int c = (((cpos >> 1) ^ wswz(row)) << 1) | (cpos & 1);      // global chunk that belongs there
            p_row[h][i] = row;
            p_src[h][i] = P + (size_t)(p_begin + row) * a.p_pstride + ta * BA + c * 8;
            const int j0 = tj * BJ + c * 8;
            const int tap = j0 >> a.log2CB;
            q_row[h][i] = row;
            q_cb[h][i] = j0 & (a.CBp - 1);
            q_tap_ok[h][i] = tap < a.KH * a.KW;
            q_kh[h][i] = q_tap_ok[h][i] ? tap / a.KW : 0;
            q_kw[h][i] = q_tap_ok[h][i] ? tap - q_kh[h][i] * a.KW : 0;
            const int p = p_begin + row;
            q_n[h][i] = p / HWp;
            const int rem = p - q_n[h][i] * HWp;
            q_h[h][i] = rem / a.Wp;
            q_w[h][i] = rem - q_h[h][i] * a.Wp;
        }

    // Staging a half = prep (addresses + validity of this lane's 2 P and 2 Q pieces; advances the lane's Q pixels of that half by 64) and
    // issue (the 4 DMA instructions).  The prep runs in the shadow of the wave's own MFMAs one phase before the issue: with the address
    // arithmetic inside the read sections (between the barriers, where the partner wave's MFMAs only cover ~256 clocks) the four-phase
    // schedule was SLOWER than the one-barrier kernel (218 vs 191 us).
    const bf16_t *srcP[2], *srcQ[2];
    auto prep_half = [&](auto HH, int kt) __attribute__((always_inline)) {
        constexpr int h = decltype(HH)::value;
        const int pbase = p_begin + kt * BP;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            srcP[i] = (pbase + p_row[h][i] < p_end) ? p_src[h][i] + (size_t)kt * BP * a.p_pstride : zero;
            const int hh = q_h[h][i] * a.step - a.pad + q_kh[h][i], ww = q_w[h][i] * a.step - a.pad_w + q_kw[h][i];
            const bool ok = q_tap_ok[h][i] && (pbase + q_row[h][i] < p_end) && ((unsigned)hh < (unsigned)a.Hq) && ((unsigned)ww < (unsigned)a.Wq);
            srcQ[i] = ok ? Q + ((size_t)(q_n[h][i] * a.Hq + hh) * a.Wq + ww) * a.q_pstride + q_cb[h][i] : zero;
            q_w[h][i] += a.dw;
            const int cw = q_w[h][i] >= a.Wp;
            q_w[h][i] -= cw ? a.Wp : 0;
            q_h[h][i] += a.dh + cw;
            const int chh = q_h[h][i] >= a.Hp;
            q_h[h][i] -= chh ? a.Hp : 0;
            q_n[h][i] += a.dn + chh;
        }
    };
    auto issue_half = [&](auto HH, int bufi) __attribute__((always_inline)) {
        constexpr int h = decltype(HH)::value;
        bf16_t *base = smem + bufi * BUF;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)srcP[i],
                                             (__attribute__((address_space(3))) void *)(base + (h * HP + wave * 4 + i * 2) * BA), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)srcQ[i],
                                             (__attribute__((address_space(3))) void *)(base + TA + (h * HP + wave * 4 + i * 2) * BJ), 16, 0, 0);
    };
    auto stage_half = [&](auto HH, int kt, int bufi) __attribute__((always_inline)) { prep_half(HH, kt); issue_half(HH, bufi); };

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // prologue: H0(0), H1(0), H0(1) -- H1(1) is staged by phase 1 of step 0.  The staging ORDER per half is kt = 0, 1, 2, ... (the
    // pixel advance is stateful)
    if (T > 0) { stage_half(IC<0>{}, 0, 0); stage_half(IC<1>{}, 0, 0); }
    if (T > 1) {
        stage_half(IC<0>{}, 1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // H0(0) has landed; H1(0), H0(1) stay in flight
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    DL_WBAR();
    if (grp1) DL_WBAR();          // stagger: waves 4-7 run one barrier behind

    bf16x8_t bq[4], ap[4];
    auto read_q = [&](const bf16_t *Qs, int prow0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bq[j] = tr_fragment_swz<ROWJ>(Qs, prow0, wj * 4 + j, lane);
    };
    auto read_p = [&](const bf16_t *Ps, int prow0, int half) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ap[i] = tr_fragment_swz<ROWA>(Ps, prow0, wa * 8 + half * 4 + i, lane);
    };
    auto mma = [&](auto HALF) __attribute__((always_inline)) {
        constexpr int i0 = decltype(HALF)::value * 4;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i0 + i][j] = dl_mfma16(ap[i], bq[j], acc[i0 + i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
    // first barrier, the 16 MFMAs of this phase (+ PREP: next half's addresses, independent VALU work issued between them), second barrier
#define DL_W_PHASE_SYNC_MMA(HALF, PREP)                        \
    __builtin_amdgcn_sched_barrier(0);                         \
    DL_WBAR();                                                 \
    mma(IC<HALF>{});                                           \
    PREP;                                                      \
    __builtin_amdgcn_sched_barrier(0);                         \
    DL_WBAR();

    for (int t = 0; t < T; ++t) {
        const int cur = t & 1;
        const bf16_t *Ps = smem + cur * BUF, *Qs = Ps + TA;
        const bool more1 = t + 1 < T, more2 = t + 2 < T;
        // ---- phase 0
        read_q(Qs, 0);
        read_p(Ps, 0, 0);
        DL_W_PHASE_SYNC_MMA(0, if (more1) prep_half(IC<1>{}, t + 1))
        // ---- phase 1
        read_p(Ps, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (more1) {
            issue_half(IC<1>{}, cur ^ 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");               // H1(t) has landed; H0(t+1), H1(t+1) stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        DL_W_PHASE_SYNC_MMA(1, (void)0)
        // ---- phase 2
        read_q(Qs, HP);
        read_p(Ps, HP, 0);
        DL_W_PHASE_SYNC_MMA(0, if (more2) prep_half(IC<0>{}, t + 2))
        // ---- phase 3
        read_p(Ps, HP, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (more2) {
            issue_half(IC<0>{}, cur);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");               // H0(t+1) has landed; H1(t+1), H0(t+2) stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        DL_W_PHASE_SYNC_MMA(1, (void)0)
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

static int launch_wgrad_8ph(WgradArgs a, hipStream_t stream) {
    constexpr size_t smem = (size_t)2 * 64 * (256 + 256) * sizeof(bf16_t);
    a.tiles_a = a.CAp / 256;
    a.tiles_j = (a.J + 255) / 256;
    a.pchunk = ((a.Ptot + a.splitk - 1) / a.splitk + 63) / 64 * 64;
    const int hw = a.Hp * a.Wp;
    a.dn = 64 / hw; a.dh = (64 % hw) / a.Wp; a.dw = (64 % hw) % a.Wp;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad_8ph_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) DL_FAIL("dl_conv_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_8ph_kernel, dim3(a.tiles_a * a.tiles_j, a.splitk), dim3(512), smem, stream, a);
    DL_CHECK_LAUNCH("dl_conv_wgrad(8-phase)");
    return 0;
}

template <int BA, int WA, int WJ, bool TRASM>
static int launch_wgrad_glds_v(WgradArgs a, const WgradLayers &lay, int n, hipStream_t stream) {
    constexpr size_t smem = (size_t)2 * 64 * (BA + 256) * sizeof(bf16_t);
    a.tiles_a = a.CAp / BA;
    a.tiles_j = (a.J + 255) / 256;
    a.pchunk = ((a.Ptot + a.splitk - 1) / a.splitk + 63) / 64 * 64;
    const int hw = a.Hp * a.Wp;
    a.dn = 64 / hw; a.dh = (64 % hw) / a.Wp; a.dw = (64 % hw) % a.Wp;
    auto kern = wgrad_glds_kernel<BA, WA, WJ, TRASM>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) DL_FAIL("dl_conv_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.tiles_a * a.tiles_j, a.splitk, n), dim3(512), smem, stream, a, lay);
    DL_CHECK_LAUNCH("dl_conv_wgrad(glds)");
    return 0;
}

template <int BA, int WA, int WJ>
static int launch_wgrad_glds(const WgradArgs &a, const WgradLayers &lay, int n, hipStream_t stream) {
    return a.tr_asm ? launch_wgrad_glds_v<BA, WA, WJ, true>(a, lay, n, stream) : launch_wgrad_glds_v<BA, WA, WJ, false>(a, lay, n, stream);
}

// grad[a][b][t] (+)= sum_ks slab[ks][a][t*CBp + b].  Threads walk the slab in its own (contiguous) order so the reads
// coalesce; the (small) gradient tensor takes the strided writes.  (A 4-lanes-per-column variant with every load in flight was tried in
// r02: 22.9 vs 21.6 us on the 66 MB ResnetBlock slab -- the pass is bound by reading partials the previous kernel has just written,
// not by loads in flight -- so the sequential summation order stayed.)
__device__ __forceinline__ void wgrad_reduce_body(const float *slab, int splitk, int kstride_, int CBp, int J, int CA, int CB, int KK, float *grad,
                                                  int accumulate, int stack_kw, int bid, int nb) {
    const int J4 = J / 4;
    const size_t total = (size_t)CA * J4;
    const size_t kstride = (size_t)kstride_;
    for (size_t i = bid * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)nb * blockDim.x) {
        const int ca = (int)(i / J4), j = (int)(i % J4) * 4;
        const float *src = slab + (size_t)ca * J + j;
        // eight partials in flight, then added in the SAME order as before (k = 0, 1, 2, ...: bit-identical results).  Since the deferred reduction the partials
        // come from HBM, not from the Infinity Cache behind the kernel that wrote them: with four loads per thread in flight the batched pass ran at 1.7 TB/s
        // (r05, profiles/r05/bench_train_kernel_stats_bf16_r05.csv: 20 launches x 146 us per step for 4.9 GB)
        f32x4_t s = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 8 <= splitk; k += 8) {
            f32x4_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(src + (size_t)(k + u) * kstride));
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < splitk; ++k) s += __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(src + (size_t)k * kstride));
        const int t = j / CBp, b0 = j % CBp;            // 4 consecutive j share the tap (CBp is a multiple of 8)
        if (t >= KK) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int b = b0 + e;
            if (b >= CB) continue;
            // stacked rows (ca = a*stack_kw + kw, t = kh): grad[a][b][kh][kw]
            float *g = stack_kw ? grad + (((size_t)(ca / stack_kw) * CB + b) * KK + t) * stack_kw + (ca % stack_kw)
                                : grad + ((size_t)ca * CB + b) * KK + t;
            *g = accumulate ? *g + s[e] : s[e];
        }
    }
}

// The same reduction with coalesced writes (r05).  grad[a][b][t] has the tap index fastest, the slab row has b fastest: the loop above writes one
// float per 64-byte line (KK = 16: UNet-512 gradients, 218 MB per generator, cost 8-16 x their size in write traffic; rocprofv3 of the 18-net step:
// 46 launches x 253 us = 8.4 % of the step, profiles/r05/bench_train18_kernel_stats_r05.csv).  Here a workgroup owns one weight row ca and RED_BCHUNK
// consecutive b: it reads the KK runs of the slab row (float4, coalesced, eight partials in flight, summed in the same order k = 0, 1, 2 ...: bit-identical
// to wgrad_reduce_body), transposes through LDS and writes nreal * KK CONSECUTIVE floats of the gradient.
constexpr int RED_BCHUNK = 64, RED_KK_MAX = 64, RED_MAX_BLOCKS = 8192;

__device__ __forceinline__ bool wgrad_reduce_tiled(int KK, int stack_kw) { return stack_kw == 0 && KK <= RED_KK_MAX; }

__device__ __forceinline__ void wgrad_reduce_tiles(const float *slab, int splitk, int kstride_, int CBp, int J, int CA, int CB, int KK, float *grad,
                                                   int accumulate, int bid, int nb, float *lds) {
    const int nbc = (CBp + RED_BCHUNK - 1) / RED_BCHUNK, KP = KK + 1;
    const size_t kstride = (size_t)kstride_;
    for (int item = bid; item < CA * nbc; item += nb) {
        const int ca = item / nbc, b0 = (item - ca * nbc) * RED_BCHUNK;
        const int width = min(RED_BCHUNK, CBp - b0), q = width / 4;         // CBp is a multiple of 8
        const int nreal = min(width, CB - b0);                              // the padded channels have no gradient
        if (nreal > 0) {
            const float *base = slab + (size_t)ca * J + b0;
            for (int i = threadIdx.x; i < KK * q; i += 256) {
                const int t = i / q, c = (i - t * q) * 4;
                const float *src = base + (size_t)t * CBp + c;
                f32x4_t s = {0.f, 0.f, 0.f, 0.f};
                int k = 0;
                for (; k + 8 <= splitk; k += 8) {
                    f32x4_t v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(src + (size_t)(k + u) * kstride));
#pragma unroll
                    for (int u = 0; u < 8; ++u) s += v[u];
                }
                for (; k < splitk; ++k) s += __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(src + (size_t)k * kstride));
#pragma unroll
                for (int e = 0; e < 4; ++e) lds[(c + e) * KP + t] = s[e];
            }
        }
        __syncthreads();
        if (nreal > 0) {
            float *g = grad + ((size_t)ca * CB + b0) * KK;
            for (int o = threadIdx.x; o < nreal * KK; o += 256) {
                const float v = lds[o + o / KK];                            // [b][t] at pitch KK + 1
                g[o] = accumulate ? g[o] + v : v;
            }
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float *slab, int splitk, int kstride, int CBp, int J, int CA, int CB,
                                                           int KK, float *grad, int accumulate, int stack_kw) {
    __shared__ float lds[RED_BCHUNK * (RED_KK_MAX + 1)];
    if (wgrad_reduce_tiled(KK, stack_kw)) wgrad_reduce_tiles(slab, splitk, kstride, CBp, J, CA, CB, KK, grad, accumulate, blockIdx.x, gridDim.x, lds);
    else wgrad_reduce_body(slab, splitk, kstride, CBp, J, CA, CB, KK, grad, accumulate, stack_kw, blockIdx.x, gridDim.x);
}

// Deferred form (dl_conv_wgrad_slabs + dl_wgrad_reduce_batch): the slabs of MANY layers, each in its own region of a caller-owned arena, are
// combined by ONE launch at the end of a network's backward pass -- every workgroup looks its layer up in a table sorted by first block.
// Per element the summation is the loop above, so the result is bit-identical to the immediate reduction.
__global__ void __launch_bounds__(256) wgrad_reduce_batch_kernel(const dl_wgrad_reduce_entry *tab, int n) {
    __shared__ float lds[RED_BCHUNK * (RED_KK_MAX + 1)];
    int lo = 0, hi = n - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].block0 <= b) lo = mid; else hi = mid - 1;
    }
    const dl_wgrad_reduce_entry e = tab[lo];
    if (wgrad_reduce_tiled(e.KK, e.stack_kw))
        wgrad_reduce_tiles(e.slab, e.splitk, e.kstride, e.CBp, e.J, e.CA, e.CB, e.KK, e.grad, e.accumulate, b - e.block0, e.nblocks, lds);
    else wgrad_reduce_body(e.slab, e.splitk, e.kstride, e.CBp, e.J, e.CA, e.CB, e.KK, e.grad, e.accumulate, e.stack_kw, b - e.block0, e.nblocks);
}

template <typename T, int PREC, int BA, int WA, int WJ>
static int launch_wgrad(WgradArgs a, hipStream_t stream) {
    constexpr int NPL = (PREC == 3) ? 2 : 1;
    constexpr size_t smem = (size_t)2 * NPL * 32 * ((BA + 16) + (128 + 16)) * sizeof(bf16_t);
    a.tiles_a = (a.CAp + BA - 1) / BA;
    a.tiles_j = (a.J + 127) / 128;
    auto kern = wgrad_kernel<T, PREC, BA, WA, WJ>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) DL_FAIL("dl_conv_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.tiles_a * a.tiles_j, a.splitk), dim3(256), smem, stream, a);
    DL_CHECK_LAUNCH("dl_conv_wgrad");
    return 0;
}

template <typename T, int PREC>
static int dispatch_wgrad(const WgradArgs &a, hipStream_t stream) {
    if (a.CAp <= 16) return launch_wgrad<T, PREC, 16, 1, 4>(a, stream);
    if (a.CAp <= 32) return launch_wgrad<T, PREC, 32, 1, 4>(a, stream);
    if (a.CAp <= 64) return launch_wgrad<T, PREC, 64, 2, 2>(a, stream);
    return launch_wgrad<T, PREC, 128, 2, 2>(a, stream);
}

#include "wgrad_c4.h"
#include "wgrad_x3.h"
#include "wgrad_w4.h"

// The slabs of consecutive pixel ranges lie CAp * J + PAD floats apart.  Without a pad the distance is a multiple of a large power of two for every
// layer of these nets (ResnetBlock: 256 x 2304 floats = 9 x 2^18 bytes): all 28 partials of one gradient element -- written at the same moment by 28
// workgroups, read back-to-back by one thread of the reduction -- would fall on one HBM channel IF the channel were a plain bit field of the address.
// Tested r04 (DL_WGRAD_SLAB_PAD = 1088 and 4160 floats vs 0, same box, tools/gpu_r04_streams.sh): 94.75 / 94.69 vs 94.74-94.82 ms per step -- nothing;
// the memory system hashes the channel.  The pad stays available (floats, default 0) because the slab size is now asked from the library anyway.
static int wgrad_slab_pad() {
    static const int pad = [] { const char *e = DL_DEV_ENV("DL_WGRAD_SLAB_PAD"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v + 3) / 4 * 4; }();
    return pad;
}

extern "C" size_t dl_wgrad_slab_floats(const dl_wgrad_desc *d) {
    if (!d || d->splitk < 1) return 0;
    return (size_t)d->splitk * ((size_t)d->CAp * d->KH * d->KW * d->CBp + wgrad_slab_pad());
}

// Which kernel of the general path serves a descriptor (one decision for dl_conv_wgrad, dl_conv_wgrad_multi and dl_wgrad_plan)
enum WgradKernel { WK_GENERIC = 0, WK_GLDS256, WK_GLDS128, WK_X3_256, WK_X3_128, WK_8PH, WK_4PH_X3, WK_4PH_X3_NOPRIO, WK_W4 };

static int wgrad_kernel_of(const dl_wgrad_desc *d, int *err) {
    *err = 0;
    const int J = d->KH * d->KW * d->CBp;
    const long Ptot = (long)d->N * d->Hp * d->Wp;
    static const bool no_glds = DL_DEV_ENV("DL_NO_GLDS") != nullptr;
    const bool fast = d->dtype == DL_BF16 && d->prec == DL_PREC_BF16 && d->p_act == DL_ACT_NONE && d->q_act == DL_ACT_NONE &&
                      d->pad_mode == DL_PAD_ZERO && J >= 256 && Ptot >= 64L * d->splitk && !no_glds;
    // "1": the four-phase schedule (wgrad_8ph_kernel).  OFF by default -- measured r02, same box, ResnetBlock shape, kernel + reduce:
    // 217-225 us vs 191-193 us for the one-barrier kernel in all three variants tried (reads retired before the first barrier; restage
    // two phases later without forced waits; address arithmetic moved into the MFMA shadow).  The one-barrier kernel alone is 172 us,
    // within 8 % of the forward's 8-phase kernel (155-160 us), so there was little left to win here.
    static const char *w8 = DL_DEV_ENV("DL_WGRAD_8PH");
    // strict policy on the direct-to-LDS path (wgrad_x3.h); DL_NO_X3_GLDS=1: the round-1 register-staged kernel (A/B)
    const bool no_x3 = dl_switch(DL_SW_NO_X3_GLDS) != nullptr;
    const bool act_ok3 = (d->p_act == DL_ACT_NONE || d->p_act == DL_ACT_RELU || d->p_act == DL_ACT_LRELU) &&
                         (d->q_act == DL_ACT_NONE || d->q_act == DL_ACT_RELU || d->q_act == DL_ACT_LRELU);
    const bool fast3 = d->dtype == DL_F32 && d->prec == DL_PREC_BF16X3 && act_ok3 && d->pad_mode == DL_PAD_ZERO && J >= 256 &&
                       Ptot >= 32L * d->splitk && (d->p_pstride % 4) == 0 && (d->q_pstride % 4) == 0 && !no_x3;
    // DL_WGRAD_X3 = "2": the staggered two-phase schedule (wgrad_4ph_x3_kernel), "3": the same without s_setprio.  OFF by default -- measured r03,
    // same box, ResnetBlock shape, kernel + reduce: 627 us (604 without s_setprio) vs 577 us for the one-barrier kernel; PMC: the stagger
    // raises the time waves spend parked at barriers / waitcnt (43 % vs 28 % of wave cycles) more than it overlaps (MFMA-busy 31.6 % vs 35.6 %).
    static const char *w3 = DL_DEV_ENV("DL_WGRAD_X3");
    if ((d->p_split || d->q_split) && !(fast3 && (d->CAp % 128) == 0 && (!d->p_split || d->p_act == DL_ACT_NONE) && (!d->q_split || d->q_act == DL_ACT_NONE))) {
        *err = 1;
        return WK_GENERIC;
    }
    if (w4w_eligible(d)) return WK_W4;
    if (fast3 && (d->CAp % 256) == 0 && w3 && (w3[0] == '2' || w3[0] == '3') && !d->p_split && !d->q_split) return w3[0] == '3' ? WK_4PH_X3_NOPRIO : WK_4PH_X3;
    if (fast3 && (d->CAp % 256) == 0) return WK_X3_256;
    if (fast3 && (d->CAp % 128) == 0) return WK_X3_128;
    if (fast && (d->CAp % 256) == 0 && w8 && w8[0] == '1') return WK_8PH;
    if (fast && (d->CAp % 256) == 0) return WK_GLDS256;
    if (fast && (d->CAp % 128) == 0) return WK_GLDS128;
    return WK_GENERIC;
}

// the split-K kernel of the general path for n same-shaped layers (n > 1: the direct-to-LDS kernels only): slabs only (the reduction is the
// caller's: immediate or deferred)
static int wgrad_slabs(const dl_wgrad_desc *d, const WgradLayers &lay, int n, hipStream_t stream, int *J_out) {
    if (!d) DL_FAIL("dl_conv_wgrad: null descriptor");
    if (d->N <= 0 || d->Hp <= 0 || d->Wp <= 0 || d->Hq <= 0 || d->Wq <= 0)
        DL_FAIL("dl_conv_wgrad: empty problem (N=%d, P %dx%d, Q %dx%d): nothing to launch", d->N, d->Hp, d->Wp, d->Hq, d->Wq);
    if (n < 1 || n > WGRAD_MULTI_MAX) DL_FAIL("dl_conv_wgrad: %d layers in one launch (1 .. %d)", n, WGRAD_MULTI_MAX);
    for (int l = 0; l < n; ++l)
        if (!lay.P[l] || !lay.Q[l] || !lay.slab[l]) DL_FAIL("dl_conv_wgrad: null argument");
    const int l2 = ilog2_exact(d->CBp);
    if (l2 < 3) DL_FAIL("dl_conv_wgrad: CBp=%d must be a power of two >= 8", d->CBp);
    if (d->CAp % 8) DL_FAIL("dl_conv_wgrad: CAp=%d must be a multiple of 8", d->CAp);
    if (d->p_pstride % 8 || d->q_pstride % 8) DL_FAIL("dl_conv_wgrad: pixel strides must be multiples of 8");
    if (d->splitk < 1) DL_FAIL("dl_conv_wgrad: splitk=%d", d->splitk);
    if (d->stack_kw && d->KW != 1) DL_FAIL("dl_conv_wgrad: stack_kw needs KW == 1");
    if (d->prec == DL_PREC_BF16X3 && d->dtype != DL_F32) DL_FAIL("dl_conv_wgrad: BF16X3 needs fp32 activations");

    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.P = lay.P[0]; a.Q = lay.Q[0]; a.slab = lay.slab[0];
    a.N = d->N; a.Hp = d->Hp; a.Wp = d->Wp; a.CAp = d->CAp; a.p_pstride = d->p_pstride;
    a.Hq = d->Hq; a.Wq = d->Wq; a.CBp = d->CBp; a.log2CB = l2; a.q_pstride = d->q_pstride;
    a.KH = d->KH; a.KW = d->KW; a.step = d->step; a.pad = d->pad; a.pad_w = d->pad_w < 0 ? d->pad : d->pad_w; a.pad_mode = d->pad_mode;
    a.J = d->KH * d->KW * d->CBp;
    a.kstride = d->CAp * a.J + wgrad_slab_pad();
    a.Ptot = d->N * d->Hp * d->Wp;
    a.splitk = d->splitk;
    a.pchunk = ((a.Ptot + d->splitk - 1) / d->splitk + 31) / 32 * 32;
    const int hw = d->Hp * d->Wp;
    a.dn = 32 / hw; a.dh = (32 % hw) / d->Wp; a.dw = (32 % hw) % d->Wp;
    a.p_act = d->p_act; a.q_act = d->q_act;
    a.p_split = d->p_split; a.q_split = d->q_split;
    static const char *tr_env = DL_DEV_ENV("DL_WGRAD_TR_ASM");            // A/B switch: "0" = the transposing reads through the builtin (rounds 2-4)
    a.tr_asm = (tr_env && tr_env[0] == '0') ? 0 : 1;
    static const char *xg_env = DL_DEV_ENV("DL_WGRAD_XCDGROUP");          // A/B switch: "0" keeps the plain 2-D block order
    a.xcd_group = (xg_env && xg_env[0] == '0') ? 0 : 1;

    int err = 0;
    const int k = wgrad_kernel_of(d, &err);
    if (err)
        DL_FAIL("dl_conv_wgrad: split-copy operands need the strict direct-to-LDS kernel (fp32 + BF16X3, zero padding, CAp %% 128 == 0, J >= 256) and no staged activation on them");
    if (n > 1 && !(k == WK_W4 || k == WK_GLDS256 || k == WK_GLDS128 || k == WK_X3_256 || k == WK_X3_128))
        DL_FAIL("dl_conv_wgrad_multi: this descriptor takes a kernel without a batched form (ask dl_wgrad_plan first)");
    int rc;
    switch (k) {
        case WK_W4: rc = launch_wgrad_w4(d, lay, n, a.kstride, stream); break;
        case WK_4PH_X3: rc = launch_wgrad_4ph_x3<0>(a, stream); break;
        case WK_4PH_X3_NOPRIO: rc = launch_wgrad_4ph_x3<1>(a, stream); break;
        case WK_X3_256: rc = launch_wgrad_glds_x3<256>(a, lay, n, stream); break;
        case WK_X3_128: rc = launch_wgrad_glds_x3<128>(a, lay, n, stream); break;
        case WK_8PH: rc = launch_wgrad_8ph(a, stream); break;
        case WK_GLDS256: rc = launch_wgrad_glds<256, 2, 4>(a, lay, n, stream); break;
        case WK_GLDS128: rc = launch_wgrad_glds<128, 2, 4>(a, lay, n, stream); break;
        default:
            if (d->dtype == DL_BF16 && d->prec == DL_PREC_BF16) rc = dispatch_wgrad<bf16_t, 1>(a, stream);
            else if (d->dtype == DL_F32 && d->prec == DL_PREC_BF16X3) rc = dispatch_wgrad<float, 3>(a, stream);
            else if (d->dtype == DL_F32 && d->prec == DL_PREC_BF16) rc = dispatch_wgrad<float, 1>(a, stream);
            else DL_FAIL("dl_conv_wgrad: unsupported dtype/precision combination");
    }
    *J_out = a.J;
    return rc;
}

static int wgrad_slabs(const dl_wgrad_desc *d, const void *P, const void *Q, float *slab, hipStream_t stream, int *J_out) {
    if (!P || !Q || !slab) DL_FAIL("dl_conv_wgrad: null argument");
    WgradLayers lay;
    lay.P[0] = P; lay.Q[0] = Q; lay.slab[0] = slab;
    return wgrad_slabs(d, lay, 1, stream, J_out);
}

// What dl_conv_wgrad / dl_conv_wgrad_multi would run for `d` (d->splitk is ignored except where a kernel's eligibility depends on it: pass 1):
// the kernel class, its output tiles per layer and the K steps one tile runs with split-K 1 -- what a caller needs to size split-K for one
// layer or for a batch.  Returns 1 when the kernel has a batched form (dl_conv_wgrad_multi), 0 when not, < 0 on error.
extern "C" int dl_wgrad_plan(const dl_wgrad_desc *d, int32_t *tiles, int32_t *ksteps, const char **name) {
    if (!d) DL_FAIL("dl_wgrad_plan: null descriptor");
    if (wgrad_c4_form(d) || wgrad_c4_x3_form(d)) {
        if (tiles) *tiles = 0;
        if (ksteps) *ksteps = 0;
        if (name) *name = "wgrad_c4";
        return 0;
    }
    int err = 0;
    const int k = wgrad_kernel_of(d, &err);
    const int J = d->KH * d->KW * d->CBp;
    const long Ptot = (long)d->N * d->Hp * d->Wp;
    int t = 0, ks = 0, multi = 0;
    const char *nm = "wgrad_kernel";
    switch (k) {
        case WK_W4: t = (d->CAp / 128) * (d->CBp / 128) * 3; ks = d->N * d->Hp; multi = 1; nm = "wgrad_w4_kernel"; break;
        case WK_X3_256: case WK_4PH_X3: case WK_4PH_X3_NOPRIO: t = (d->CAp / 256) * ((J + 255) / 256); ks = (int)((Ptot + 31) / 32); multi = k == WK_X3_256; nm = "wgrad_glds_x3_kernel<256>"; break;
        case WK_X3_128: t = (d->CAp / 128) * ((J + 255) / 256); ks = (int)((Ptot + 31) / 32); multi = 1; nm = "wgrad_glds_x3_kernel<128>"; break;
        case WK_GLDS256: case WK_8PH: t = (d->CAp / 256) * ((J + 255) / 256); ks = (int)((Ptot + 63) / 64); multi = k == WK_GLDS256; nm = "wgrad_glds_kernel<256>"; break;
        case WK_GLDS128: t = (d->CAp / 128) * ((J + 255) / 256); ks = (int)((Ptot + 63) / 64); multi = 1; nm = "wgrad_glds_kernel<128>"; break;
        default: {
            const int ba = d->CAp <= 16 ? 16 : (d->CAp <= 32 ? 32 : (d->CAp <= 64 ? 64 : 128));
            t = ((d->CAp + ba - 1) / ba) * ((J + 127) / 128); ks = (int)((Ptot + 31) / 32);
        }
    }
    if (tiles) *tiles = t;
    if (ksteps) *ksteps = ks;
    if (name) *name = nm;
    if (err)
        DL_FAIL("dl_wgrad_plan: split-copy operands (p_split / q_split) need the strict direct-to-LDS kernels: fp32 + BF16X3, zero padding, KH*KW*CBp >= 256, "
                "CAp %% 128 == 0, pixel strides %% 4 == 0, no staged activation on a split operand (CAp=%d, J=%d, p_split=%d, q_split=%d)", d->CAp, J, d->p_split, d->q_split);
    return multi;
}

static int reduce_blocks(const dl_wgrad_desc *d, int J) {
    if (d->stack_kw == 0 && d->KH * d->KW <= RED_KK_MAX)                    // wgrad_reduce_tiles: one (weight row, 64 channels) item per workgroup pass
        return (int)min((size_t)RED_MAX_BLOCKS, (size_t)d->CA * ((d->CBp + RED_BCHUNK - 1) / RED_BCHUNK));
    const size_t total = (size_t)d->CA * (J / 4);
    return (int)min((size_t)4096, (total + 255) / 256);
}

extern "C" int dl_conv_wgrad(const dl_wgrad_desc *d, const void *P, const void *Q, float *grad, float *slab, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!d) DL_FAIL("dl_conv_wgrad: null descriptor");
    if (!grad) DL_FAIL("dl_conv_wgrad: null argument");
    if (d->N > 0 && d->Hp > 0 && d->Wp > 0 && P && Q && slab) {
        if (const int form = wgrad_c4_form(d)) return launch_wgrad_c4(d, form, P, Q, grad, slab, stream);
        if (const int form = wgrad_c4_x3_form(d)) return launch_wgrad_c4_x3(d, form, P, Q, grad, slab, stream);
    }
    int J = 0;
    if (const int rc = wgrad_slabs(d, P, Q, slab, stream, &J)) return rc;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(reduce_blocks(d, J)), dim3(256), 0, stream, slab, d->splitk, d->CAp * J + wgrad_slab_pad(), d->CBp, J, d->CA, d->CB,
                       d->KH * d->KW, grad, d->accumulate, d->stack_kw);
    DL_CHECK_LAUNCH("dl_conv_wgrad(reduce)");
    return 0;
}

extern "C" int dl_conv_wgrad_deferrable(const dl_wgrad_desc *d) {
    return d && !wgrad_c4_form(d) && !wgrad_c4_x3_form(d);
}

extern "C" int dl_conv_wgrad_slabs(const dl_wgrad_desc *d, const void *P, const void *Q, float *grad, float *slab, dl_wgrad_reduce_entry *entry_host,
                                   void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!d || !entry_host || !grad) DL_FAIL("dl_conv_wgrad_slabs: null argument");
    if (!dl_conv_wgrad_deferrable(d)) DL_FAIL("dl_conv_wgrad_slabs: the persistent narrow-channel kernels reduce in place; call dl_conv_wgrad");
    int J = 0;
    if (const int rc = wgrad_slabs(d, P, Q, slab, stream, &J)) return rc;
    dl_wgrad_reduce_entry e;
    memset(&e, 0, sizeof(e));
    e.slab = slab; e.grad = grad;
    e.splitk = d->splitk; e.CAp = d->CAp; e.CBp = d->CBp; e.J = J; e.CA = d->CA; e.CB = d->CB; e.KK = d->KH * d->KW;
    e.accumulate = d->accumulate; e.stack_kw = d->stack_kw;
    e.block0 = 0; e.nblocks = reduce_blocks(d, J); e.kstride = d->CAp * J + wgrad_slab_pad();
    *entry_host = e;
    return 0;
}

// n same-shaped layers (one descriptor, d->splitk row / pixel ranges each) in ONE split-K launch; the slabs of layer l start at
// slab + l * dl_wgrad_slab_floats(d); entries_host[l] = its pending reduction (as dl_conv_wgrad_slabs)
extern "C" int dl_conv_wgrad_multi(const dl_wgrad_desc *d, int n, const void *const *P, const void *const *Q, float *const *grad, float *slab,
                                   dl_wgrad_reduce_entry *entries_host, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!d || !P || !Q || !grad || !slab || !entries_host) DL_FAIL("dl_conv_wgrad_multi: null argument");
    if (n < 1 || n > WGRAD_MULTI_MAX) DL_FAIL("dl_conv_wgrad_multi: n=%d (1 .. %d)", n, WGRAD_MULTI_MAX);
    if (!dl_conv_wgrad_deferrable(d)) DL_FAIL("dl_conv_wgrad_multi: the persistent narrow-channel kernels reduce in place; call dl_conv_wgrad");
    const size_t per_layer = dl_wgrad_slab_floats(d);
    WgradLayers lay;
    for (int l = 0; l < n; ++l) {
        if (!grad[l]) DL_FAIL("dl_conv_wgrad_multi: null gradient");
        lay.P[l] = P[l]; lay.Q[l] = Q[l]; lay.slab[l] = slab + (size_t)l * per_layer;
    }
    int J = 0;
    if (const int rc = wgrad_slabs(d, lay, n, stream, &J)) return rc;
    for (int l = 0; l < n; ++l) {
        dl_wgrad_reduce_entry e;
        memset(&e, 0, sizeof(e));
        e.slab = lay.slab[l]; e.grad = grad[l];
        e.splitk = d->splitk; e.CAp = d->CAp; e.CBp = d->CBp; e.J = J; e.CA = d->CA; e.CB = d->CB; e.KK = d->KH * d->KW;
        e.accumulate = d->accumulate; e.stack_kw = d->stack_kw;
        e.block0 = 0; e.nblocks = reduce_blocks(d, J); e.kstride = d->CAp * J + wgrad_slab_pad();
        entries_host[l] = e;
    }
    return 0;
}

extern "C" int dl_wgrad_reduce_batch(const dl_wgrad_reduce_entry *table_dev, int count, int total_blocks, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!table_dev || count <= 0 || total_blocks <= 0) DL_FAIL("dl_wgrad_reduce_batch: empty table");
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(total_blocks), dim3(256), 0, stream, table_dev, count);
    DL_CHECK_LAUNCH("dl_wgrad_reduce_batch");
    return 0;
}
