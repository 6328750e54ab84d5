// common.h -- shared device helpers for libdeepliif_hip (gfx950 only; wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/deepliif_hip.h"

typedef uint16_t bf16_t;   // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;     // MFMA A/B fragment (8 bf16 = 4 VGPR)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;      // MFMA 16x16 accumulator fragment
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

void dl_set_error(const char *fmt, ...);
#define DL_FAIL(...) do { dl_set_error(__VA_ARGS__); return -1; } while (0)
#define DL_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { \
    dl_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); return -2; } } while (0)

// ---- the library's 16-bit format.  The same sources build two libraries (Makefile): libdeepliif_hip.so with bfloat16 (training + inference; the
// reference's fp32 range, 8 bits of precision) and libdeepliif_hip_f16.so (-DDL_H16_FP16) with IEEE half as the storage AND MFMA operand type: 11 bits of
// precision at the same matrix rate, INFERENCE ONLY (the gradients of this model sit below half's normal range: 89-100 % of dy under 6.1e-5,
// profiles/r05/fp16_policy_experiment.json).  Everything that depends on the format goes through the helpers of this block -- the names keep "bf16" because
// that is the format of the product build; in the f16 build `bf16_t`, DL_BF16 and DL_PREC_BF16 mean "the 16-bit type of this library" (dl_half_format()).
#ifdef DL_H16_FP16
#define DL_H16_FORMAT 1
#else
#define DL_H16_FORMAT 0
#endif
#if defined(__HIP_DEVICE_COMPILE__)
typedef __bf16 dl_bf16x2_native __attribute__((ext_vector_type(2)));
typedef _Float16 dl_f16x2_native __attribute__((ext_vector_type(2)));
typedef _Float16 dl_f16x8_native __attribute__((ext_vector_type(8)));
typedef float dl_f32x2_native __attribute__((ext_vector_type(2)));
#endif
// the low / high half of a packed 32-bit word as fp32 (bf16: a shift / a mask; half: v_cvt_f32_f16)
__device__ __forceinline__ float h16_lo_f32(uint32_t w) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifdef DL_H16_FP16
    return (float)__builtin_bit_cast(dl_f16x2_native, w)[0];
#else
    return __uint_as_float(w << 16);
#endif
#else
    return 0.f;
#endif
}
__device__ __forceinline__ float h16_hi_f32(uint32_t w) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifdef DL_H16_FP16
    return (float)__builtin_bit_cast(dl_f16x2_native, w)[1];
#else
    return __uint_as_float(w & 0xffff0000u);
#endif
#else
    return 0.f;
#endif
}
// ---- 16-bit <-> fp32 (round to nearest even; NaN handling not needed on this path)
__host__ __device__ __forceinline__ float bf16_to_f32(bf16_t h) {
#ifdef DL_H16_FP16
    return (float)__builtin_bit_cast(_Float16, h);
#else
    uint32_t u = ((uint32_t)h) << 16;
    float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __uint_as_float(u);
#else
    memcpy(&f, &u, 4);
#endif
    return f;
#endif
}
// Device code converts with the native gfx950 instruction (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32: round to nearest even, ONE VALU op per pair); the
// integer emulation below (5-7 ops per pair, a dependent chain) is what every bf16 store used to pay -- the PMC passes of round 1
// showed the norm apply kernels waiting on instruction issue for ~46 % of their wave cycles.  Identical results for finite values.
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    const dl_f32x2_native f = {lo, hi};
#ifdef DL_H16_FP16
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, dl_f16x2_native));
#else
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, dl_bf16x2_native));
#endif
#else
    return 0;
#endif
}
__host__ __device__ __forceinline__ bf16_t f32_to_bf16(float f) {
#ifdef DL_H16_FP16
    return __builtin_bit_cast(bf16_t, (_Float16)f);
#else
#if defined(__HIP_DEVICE_COMPILE__)
    return (bf16_t)(pack2_bf16(f, 0.f) & 0xffffu);
#else
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
#endif
#endif
}
// ---- the matrix instructions on 16-bit operands (A / B fragments are 8 packed values = 4 VGPRs whatever the format)
template <typename V, typename A> __device__ __forceinline__ A dl_mfma32(V a, V b, A c) {        // v_mfma_f32_32x32x16_{bf16,f16}
#if defined(__HIP_DEVICE_COMPILE__)
#ifdef DL_H16_FP16
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(dl_f16x8_native, a), __builtin_bit_cast(dl_f16x8_native, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
#else
    return c;
#endif
}
template <typename V, typename A> __device__ __forceinline__ A dl_mfma16(V a, V b, A c) {        // v_mfma_f32_16x16x32_{bf16,f16}
#if defined(__HIP_DEVICE_COMPILE__)
#ifdef DL_H16_FP16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dl_f16x8_native, a), __builtin_bit_cast(dl_f16x8_native, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
#else
    return c;
#endif
}

// ---- 8-element vector load/store of an activation row chunk, as fp32 registers
template <typename T> struct Vec8;
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[8]) {
        const float4 a = *reinterpret_cast<const float4 *>(p);
        const float4 b = *reinterpret_cast<const float4 *>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float *p, const float (&v)[8]) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4 *>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    // non-temporal variants (streaming passes: the line is not wanted in L2 again)
    static __device__ __forceinline__ void load_nt(const float *p, float (&v)[8]) {
        const f32x4_t a = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(p));
        const f32x4_t b = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(p + 4));
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    }
    static __device__ __forceinline__ void store_nt(float *p, const float (&v)[8]) {
        __builtin_nontemporal_store(f32x4_t{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4_t *>(p));
        __builtin_nontemporal_store(f32x4_t{v[4], v[5], v[6], v[7]}, reinterpret_cast<f32x4_t *>(p + 4));
    }
};
template <> struct Vec8<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t *p, float (&v)[8]) {
        const u32x4_t a = *reinterpret_cast<const u32x4_t *>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = h16_lo_f32(a[i]);
            v[2 * i + 1] = h16_hi_f32(a[i]);
        }
    }
    static __device__ __forceinline__ void store(bf16_t *p, const float (&v)[8]) {
        u32x4_t a;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = pack2_bf16(v[2 * i], v[2 * i + 1]);
        *reinterpret_cast<u32x4_t *>(p) = a;
    }
    static __device__ __forceinline__ void load_nt(const bf16_t *p, float (&v)[8]) {
        const u32x4_t a = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = h16_lo_f32(a[i]);
            v[2 * i + 1] = h16_hi_f32(a[i]);
        }
    }
    static __device__ __forceinline__ void store_nt(bf16_t *p, const float (&v)[8]) {
        u32x4_t a;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = pack2_bf16(v[2 * i], v[2 * i + 1]);
        __builtin_nontemporal_store(a, reinterpret_cast<u32x4_t *>(p));
    }
};
// SPLIT COPY of 8 fp32 values (strict policy, see conv_x3.h / wgrad_x3.h): the same 32 bytes an fp32 group of 8 channels occupies, holding
// [8 x bf16 hi | 8 x bf16 lo] with hi = bf16(x), lo = bf16(x - hi) -- exactly what the strict kernels would compute from the fp32 values while
// staging; a producer that writes it once saves every consumer (forward conv, data gradient, weight gradient: 9 taps each) the conversion.
__device__ __forceinline__ void store_split8(float *p, const float (&v)[8]) {
    u32x4_t hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t h = pack2_bf16(v[2 * i], v[2 * i + 1]);
        hi[i] = h;
        lo[i] = pack2_bf16(v[2 * i] - h16_lo_f32(h), v[2 * i + 1] - h16_hi_f32(h));
    }
    *reinterpret_cast<u32x4_t *>(p) = hi;
    *reinterpret_cast<u32x4_t *>(p + 4) = lo;
}

template <typename T> __device__ __forceinline__ float load1(const T *p);
template <> __device__ __forceinline__ float load1<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float load1<bf16_t>(const bf16_t *p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void store1(T *p, float v);
template <> __device__ __forceinline__ void store1<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void store1<bf16_t>(bf16_t *p, float v) { *p = f32_to_bf16(v); }

__device__ __forceinline__ float apply_act(int act, float v) {
    switch (act) {
#ifdef DL_H16_FP16
        case DL_ACT_RELU: return v < 0.f ? 0.f : v;      // same values, but NaN stays NaN: a half overflow (inf in a conv output -> NaN statistics) must reach the
                                                         // network's output, where deepliif_amd/inference.py looks for it, instead of being clamped to 0 here
#else
        case DL_ACT_RELU: return v > 0.f ? v : 0.f;
#endif
        case DL_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
        case DL_ACT_TANH: return tanhf(v);
        case DL_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
        default: return v;
    }
}
// derivative of the activation expressed through its OUTPUT (relu/lrelu keep the sign; tanh' = 1 - y^2)
__device__ __forceinline__ float act_grad_from_output(int act, float y) {
    switch (act) {
        case DL_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case DL_ACT_LRELU: return y > 0.f ? 1.f : 0.2f;
        case DL_ACT_TANH: return 1.f - y * y;
        case DL_ACT_SIGMOID: return y * (1.f - y);
        default: return 1.f;
    }
}

// wave64 butterfly sum
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// XCD-aware block remap (8 XCDs, private L2s): consecutive logical blocks share an XCD.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---- runtime switches (include/deepliif_hip.h, "Runtime switches").  The shipped library looks at NINE documented environment variables, all of them
// choices between two correct code paths, read ONCE when the library is loaded (error.cpp; dl_switches_reload() re-reads them for tests): no getenv on a
// launch path, nothing that changes while threads are launching.  Everything else -- kernel variants kept for A/B measurements and the timing-only
// ablations whose results are WRONG by construction -- exists only in a build with -DDL_DEV_SWITCHES (make dev -> libdeepliif_hip_dev.so, used by tools/).
enum { DL_SW_CONV_S2F = 0, DL_SW_CONV_S2FX3, DL_SW_CONV_W4X3, DL_SW_PACK_TILED, DL_SW_NO_X3_GLDS, DL_SW_NO_WGRAD_C4, DL_SW_NO_C4_X3, DL_SW_CONV_S2D, DL_SW_CONV_DOT, DL_SW_COUNT };
const char *dl_switch(int id);                 // the variable's value at load time, nullptr when unset
static inline bool dl_switch_is_one(int id) { const char *v = dl_switch(id); return v && v[0] == '1' && v[1] == 0; }      // same parse as os.environ.get(name) == '1'
#ifdef DL_DEV_SWITCHES
#define DL_DEV_ENV(name) getenv(name)
#else
#define DL_DEV_ENV(name) ((const char *)nullptr)
#endif

static inline int ilog2_exact(int v) { int l = 0; while ((1 << l) < v) ++l; return ((1 << l) == v) ? l : -1; }
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
