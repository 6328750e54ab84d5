// pack_tile.h -- the tiled form of the batched weight packing (r05), shared by pack.hip and the host check pack_tile_check.cpp
// (tests/test_pack_tile_host.py runs the two functions below thread by thread on the CPU against the element-wise decode).
//
// Why: the GEMM image has the contracted channel fastest ([row][tap][c]); the master weight [A][B][KH][KW] has the tap fastest.  The chunk-per-thread
// kernel reads eight floats KH*KW (or B*KH*KW) apart per 16-byte store -- one float per 64-byte line and instruction: rocprofv3 of the 18-net step,
// 2 launches x 2.2 ms per step for 343 M parameters = 10 x what 6 bytes per parameter cost at HBM speed (profiles/r05/bench_train18_kernel_stats_r05c.csv).
// Here a workgroup owns PT_R image rows x PT_NC contracted channels x every tap: it reads its part of the master weight in the master's own order
// (runs of PT_NC*KH*KW or PT_R*KH*KW consecutive floats), transposes through LDS ([row][tap][c], c fastest) and stores 128-byte runs of the image.
// The bits written are the ones the element-wise kernel writes (same conversions of the same values).
#pragma once
#include "common.h"

struct PackArgs {
    const float *src;
    bf16_t *w_hi, *w_lo;
    int A, B, KH, KW, row_is_a, rows_real, rows_pad, Cc, Cc_pad, log2Cc, n_phase, kstride, stack_kw;
    int phase_tap_begin[DL_MAX_PHASES + 1];
    int phase_kbase[DL_MAX_PHASES];
    int phase_kend[DL_MAX_PHASES];
    int8_t tap_kh[DL_MAX_TAPS], tap_kw[DL_MAX_TAPS];
};

// descriptor of the C ABI -> kernel arguments; returns NULL or what is wrong with the descriptor
static inline const char *pack_args_from_desc(const dl_pack_desc *d, const float *src, void *w_hi, void *w_lo, PackArgs &a) {
    const int l2 = ilog2_exact(d->Cc_pad);
    if (l2 < 3) return "Cc_pad must be a power of two >= 8";
    if (d->n_phase < 1 || d->n_phase > DL_MAX_PHASES) return "n_phase out of range";
    memset(&a, 0, sizeof(a));
    a.src = src; a.w_hi = (bf16_t *)w_hi; a.w_lo = (bf16_t *)w_lo;
    a.A = d->A; a.B = d->B; a.KH = d->KH; a.KW = d->KW; a.row_is_a = d->row_is_a;
    a.rows_real = d->rows_real; a.rows_pad = d->rows_pad; a.Cc = d->Cc; a.Cc_pad = d->Cc_pad; a.log2Cc = l2;
    a.n_phase = d->n_phase; a.kstride = d->kstride; a.stack_kw = d->stack_kw;
    for (int p = 0; p <= DL_MAX_PHASES; ++p) a.phase_tap_begin[p] = d->phase_tap_begin[p];
    for (int p = 0; p < d->n_phase; ++p) {
        a.phase_kbase[p] = d->phase_kbase[p];
        a.phase_kend[p] = d->phase_kbase[p] + (d->phase_tap_begin[p + 1] - d->phase_tap_begin[p]) * d->Cc_pad;
        if (a.phase_kend[p] > d->kstride) return "a phase exceeds kstride";
    }
    for (int t = 0; t < DL_MAX_TAPS; ++t) { a.tap_kh[t] = d->tap_kh[t]; a.tap_kw[t] = d->tap_kw[t]; }
    return nullptr;
}

constexpr int PT_R = 8, PT_NC = 64, PT_PITCH = PT_NC + 4, PT_KHW_MAX = 16, PT_THREADS = 256;
constexpr int PT_LDS_FLOATS = PT_R * PT_KHW_MAX * PT_PITCH;       // 34 KB
constexpr int PT_TILED_FLAG = 1 << 30;                            // in the job field of a block-table entry: .y is a tile index, not a first chunk

// The layouts the tiled form covers: every phase a whole number of taps x Cc_pad columns, back to back from column 0 to kstride (no zero columns between
// or behind them), no channel padding, kernels up to 4x4, rows not stacked.  Everything else (stems, heads, the 6-channel PatchGAN input) keeps the chunk form.
static inline bool pack_tiled_ok(const PackArgs &a) {
    const int khw = a.KH * a.KW;
    if (a.stack_kw || a.Cc != a.Cc_pad || a.Cc_pad < PT_NC || a.Cc_pad % PT_NC || khw < 1 || khw > PT_KHW_MAX) return false;
    if (a.rows_pad < PT_R || a.rows_pad % PT_R || a.rows_real < 0 || a.rows_real > a.rows_pad || a.n_phase < 1 || a.n_phase > DL_MAX_PHASES) return false;
    if (a.rows_real > (a.row_is_a ? a.A : a.B) || a.Cc > (a.row_is_a ? a.B : a.A)) return false;
    int k = 0;
    for (int p = 0; p < a.n_phase; ++p) {
        const int nt = a.phase_tap_begin[p + 1] - a.phase_tap_begin[p];
        if (nt < 0 || a.phase_kbase[p] != k || a.phase_kend[p] != k + nt * a.Cc_pad) return false;
        k = a.phase_kend[p];
    }
    if (k != a.kstride || a.phase_tap_begin[0] < 0 || a.phase_tap_begin[a.n_phase] > DL_MAX_TAPS) return false;
    for (int t = a.phase_tap_begin[0]; t < a.phase_tap_begin[a.n_phase]; ++t)
        if (a.tap_kh[t] < 0 || a.tap_kh[t] >= a.KH || a.tap_kw[t] < 0 || a.tap_kw[t] >= a.KW) return false;
    return true;
}

static inline long pack_tile_count(const PackArgs &a) { return (long)(a.rows_pad / PT_R) * (a.Cc_pad / PT_NC); }

// stage 1: the tile's part of the master weight -> LDS [row][kh*KW + kw][c] (rows past rows_real: zeros).  KHW = KH*KW as a compile-time constant for the
// kernels the networks have (the per-element index split is two divisions: by a runtime value they cost more VALU time than the loads cost HBM time)
template <int KHW>
__host__ __device__ __forceinline__ void pack_tile_load_k(const PackArgs &a, int tile, int tid, float *lds) {
    const int ncw = a.Cc_pad / PT_NC;
    const int r0 = (tile / ncw) * PT_R, c_lo = (tile - (tile / ncw) * ncw) * PT_NC;
    const int khw = KHW ? KHW : a.KH * a.KW, n = PT_R * PT_NC * khw, bk = a.B * khw, rows_real = a.rows_real;
    const bool row_is_a = a.row_is_a != 0;
    // row_is_a: master rows are image rows -- PT_NC * khw consecutive floats per row; else master rows are contracted channels -- PT_R * khw per channel
    const size_t base = row_is_a ? ((size_t)r0 * a.B + c_lo) * khw : ((size_t)c_lo * a.B + r0) * khw;
    const int run = (row_is_a ? PT_NC : PT_R) * khw;
#if defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((address_space(1))) float *src = (const __attribute__((address_space(1))) float *)a.src;    // global_load, not flat_load
#else
    const float *src = a.src;
#endif
    constexpr int UN = 8;                   // loads in flight per thread: all issued (branch-free, clamped to src[0]) before the first LDS store
    for (int i0 = tid; i0 < n; i0 += UN * PT_THREADS) {
        float v[UN];
        int dst[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = i0 + u * PT_THREADS;
            const bool in = i < n;
            const int ii = in ? i : 0;
            const int outer = ii / run, rem = ii - outer * run, inner = rem / khw, tp = rem - inner * khw;
            const int rl = row_is_a ? outer : inner, cl = row_is_a ? inner : outer;
            const bool ok = in && r0 + rl < rows_real;
            const float x = src[ok ? base + (size_t)(outer * bk + rem) : (size_t)0];
            v[u] = ok ? x : 0.f;
            dst[u] = in ? (rl * khw + tp) * PT_PITCH + cl : -1;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (dst[u] >= 0) lds[dst[u]] = v[u];
    }
}

__host__ __device__ __forceinline__ void pack_tile_load(const PackArgs &a, int tile, int tid, float *lds) {
    const int khw = a.KH * a.KW;
    if (khw == 16) pack_tile_load_k<16>(a, tile, tid, lds);
    else if (khw == 9) pack_tile_load_k<9>(a, tile, tid, lds);
    else pack_tile_load_k<0>(a, tile, tid, lds);
}

// stage 2: one 16-byte chunk (8 consecutive c of one row and tap) per thread and pass; 8 lanes = one 128-byte run of the image
__host__ __device__ __forceinline__ void pack_tile_store(const PackArgs &a, int tile, int tid, const float *lds) {
    const int ncw = a.Cc_pad / PT_NC;
    const int r0 = (tile / ncw) * PT_R, c_lo = (tile - (tile / ncw) * ncw) * PT_NC;
    const int khw = a.KH * a.KW, t0 = a.phase_tap_begin[0], nt = a.phase_tap_begin[a.n_phase] - t0;
    const int n = PT_R * nt * (PT_NC / 8);
    for (int i = tid; i < n; i += PT_THREADS) {
        const int c8 = i & (PT_NC / 8 - 1), q = i / (PT_NC / 8), rl = q / nt, ts = q - rl * nt;
        const int t = t0 + ts, tap = a.tap_kh[t] * a.KW + a.tap_kw[t];
        const float *vp = lds + (rl * khw + tap) * PT_PITCH + c8 * 8;
        const f32x4_t v0 = *reinterpret_cast<const f32x4_t *>(vp), v1 = *reinterpret_cast<const f32x4_t *>(vp + 4);
        const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        u32x4_t hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bf16_t h0 = f32_to_bf16(v[2 * e]), h1 = f32_to_bf16(v[2 * e + 1]);
            hi[e] = (uint32_t)h0 | ((uint32_t)h1 << 16);
            lo[e] = (uint32_t)f32_to_bf16(v[2 * e] - bf16_to_f32(h0)) | ((uint32_t)f32_to_bf16(v[2 * e + 1] - bf16_to_f32(h1)) << 16);
        }
        const size_t o = (size_t)(r0 + rl) * a.kstride + (size_t)ts * a.Cc_pad + c_lo + c8 * 8;   // phases are back to back: tap slot ts starts at ts * Cc_pad
        *reinterpret_cast<u32x4_t *>(a.w_hi + o) = hi;
        if (a.w_lo) *reinterpret_cast<u32x4_t *>(a.w_lo + o) = lo;
    }
}
