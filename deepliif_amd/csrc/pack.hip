// pack.hip -- parameter packing (fp32 OIHW/IOHW -> K-contiguous bf16 hi/lo GEMM images) and the NCHW<->NHWC boundary
// converters.  All pure HBM streaming; see include/deepliif_hip.h.
#include "common.h"
#include "pack_tile.h"

// one image: element i of the K-contiguous GEMM image <- the OIHW/IOHW master weight it comes from (0 in the padding)
__device__ __forceinline__ void pack_image(const PackArgs &a, int block, int nblocks) {
    const size_t total = (size_t)a.rows_pad * a.kstride;
    for (size_t i = block * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)nblocks * blockDim.x) {
        const int row = (int)(i / a.kstride), k = (int)(i % a.kstride);
        float v = 0.f;
        if (row < a.rows_real) {
            int ph = -1;
            for (int p = 0; p < a.n_phase; ++p)
                if (k >= a.phase_kbase[p] && k < a.phase_kend[p]) ph = p;
            if (ph >= 0) {
                const int kl = k - a.phase_kbase[ph];
                const int tl = kl >> a.log2Cc, c = kl & (a.Cc_pad - 1);
                const int t = a.phase_tap_begin[ph] + tl;
                if (t < a.phase_tap_begin[ph + 1] && c < a.Cc) {
                    const int kh = a.tap_kh[t];
                    int kw = a.tap_kw[t], r = row;
                    if (a.stack_kw) { kw = row % a.KW; r = row / a.KW; }      // row = a*KW + kw
                    const int ia = a.row_is_a ? r : c, ib = a.row_is_a ? c : r;
                    v = a.src[(((size_t)ia * a.B + ib) * a.KH + kh) * a.KW + kw];
                }
            }
        }
        const bf16_t h = f32_to_bf16(v);
        a.w_hi[i] = h;
        if (a.w_lo) a.w_lo[i] = f32_to_bf16(v - bf16_to_f32(h));
    }
}

__global__ void __launch_bounds__(256) pack_weights_kernel(const PackArgs a) { pack_image(a, blockIdx.x, gridDim.x); }

// ---- batched form: every image of an optimizer's layers in ONE launch.
// Work unit = one 16-byte output chunk = 8 consecutive k of one row.  kstride and every phase base are multiples of 64 and a tap
// spans Cc_pad (a power of two >= 8) elements, so a chunk never straddles a row, a phase or a tap: ONE decode + one 16-byte
// store per 8 outputs instead of a 64-bit div/mod, a phase search and a 2-byte store per output.  Blocks are handed out by a
// host-built table {job, first chunk}, i.e. in proportion to image SIZE -- a launch no longer lasts as long as its largest image.
constexpr int PACK_CHUNKS_PER_BLOCK = 256 * 4;       // 256 threads x 4 chunks = 8192 elements

__global__ void __launch_bounds__(256) pack_weights_batch_kernel(const char *jobs, size_t job_stride, const int2 *block_tab) {
    __shared__ PackArgs a;
    __shared__ __attribute__((aligned(16))) float tile_lds[PT_LDS_FLOATS];
    const int2 bt = block_tab[blockIdx.x];           // x = job (| PT_TILED_FLAG), y = first chunk of this block inside the job's image (or its tile)
    const int *src = reinterpret_cast<const int *>(jobs + (size_t)(bt.x & ~PT_TILED_FLAG) * job_stride);      // stride = dl_pack_job_bytes()
    int *dst = reinterpret_cast<int *>(&a);
    for (int i = threadIdx.x; i < (int)(sizeof(PackArgs) / sizeof(int)); i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    if (bt.x & PT_TILED_FLAG) {                      // pack_tile.h: master weight read in its own order, transposed through LDS
        pack_tile_load(a, bt.y, threadIdx.x, tile_lds);
        __syncthreads();
        pack_tile_store(a, bt.y, threadIdx.x, tile_lds);
        return;
    }
    const int cpr = a.kstride >> 3;                  // chunks per row
    const int nchunks = a.rows_pad * cpr;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int ch = bt.y + u * 256 + threadIdx.x;
        if (ch >= nchunks) break;
        const int row = ch / cpr, k0 = (ch - row * cpr) << 3;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (row < a.rows_real) {
            int ph = -1;
            for (int p = 0; p < a.n_phase; ++p)
                if (k0 >= a.phase_kbase[p] && k0 < a.phase_kend[p]) ph = p;
            if (ph >= 0) {
                const int kl = k0 - a.phase_kbase[ph];
                const int tl = kl >> a.log2Cc, c0 = kl & (a.Cc_pad - 1);
                const int t = a.phase_tap_begin[ph] + tl;
                if (t < a.phase_tap_begin[ph + 1]) {
                    const int kh = a.tap_kh[t];
                    int kw = a.tap_kw[t], r = row;
                    if (a.stack_kw) { kw = row % a.KW; r = row / a.KW; }      // row = a*KW + kw
                    const size_t khw = (size_t)a.KH * a.KW;
                    // consecutive c: stride KH*KW floats when the row is the A index, B*KH*KW when it is the B index
                    const size_t base = a.row_is_a ? ((size_t)r * a.B + c0) * khw : ((size_t)c0 * a.B + r) * khw;
                    const size_t cstride = a.row_is_a ? khw : (size_t)a.B * khw;
                    const float *sp = a.src + base + (size_t)kh * a.KW + kw;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (c0 + e < a.Cc) v[e] = sp[e * cstride];
                }
            }
        }
        u32x4_t hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bf16_t h0 = f32_to_bf16(v[2 * e]), h1 = f32_to_bf16(v[2 * e + 1]);
            hi[e] = (uint32_t)h0 | ((uint32_t)h1 << 16);
            lo[e] = (uint32_t)f32_to_bf16(v[2 * e] - bf16_to_f32(h0)) | ((uint32_t)f32_to_bf16(v[2 * e + 1] - bf16_to_f32(h1)) << 16);
        }
        *reinterpret_cast<u32x4_t *>(a.w_hi + (size_t)ch * 8) = hi;
        if (a.w_lo) *reinterpret_cast<u32x4_t *>(a.w_lo + (size_t)ch * 8) = lo;
    }
}

static int fill_pack_args(const dl_pack_desc *d, const float *src, void *w_hi, void *w_lo, PackArgs &a, const char *who) {
    if (!d || !src || !w_hi) DL_FAIL("%s: null argument", who);
    if (const char *why = pack_args_from_desc(d, src, w_hi, w_lo, a)) DL_FAIL("%s: %s", who, why);
    return 0;
}

extern "C" int dl_pack_weights(const dl_pack_desc *d, const float *src, void *w_hi, void *w_lo, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PackArgs a;
    if (fill_pack_args(d, src, w_hi, w_lo, a, "dl_pack_weights")) return -1;
    const size_t total = (size_t)d->rows_pad * d->kstride;
    const int blocks = (int)min((size_t)4096, (total + 255) / 256);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, stream, a);
    DL_CHECK_LAUNCH("dl_pack_weights");
    return 0;
}

extern "C" size_t dl_pack_job_bytes(void) { return (sizeof(PackArgs) + 15) / 16 * 16; }

extern "C" int dl_pack_job_fill(const dl_pack_desc *d, const float *src, void *w_hi, void *w_lo, void *job_host) {
    if (!job_host) DL_FAIL("dl_pack_job_fill: null job record");
    PackArgs a;
    if (fill_pack_args(d, src, w_hi, w_lo, a, "dl_pack_job_fill")) return -1;
    memset(job_host, 0, dl_pack_job_bytes());
    memcpy(job_host, &a, sizeof(a));
    return 0;
}

// Block table of a batch: entry = {job index, first 16-byte chunk} per workgroup of PACK_CHUNKS_PER_BLOCK chunks.  Call with
// block_tab_host = NULL to get the number of entries, then again to fill them (host memory; the caller copies it to the device).
extern "C" int dl_pack_batch_blocks(const void *jobs_host, int count, int32_t *block_tab_host) {
    if (!jobs_host || count < 0) DL_FAIL("dl_pack_batch_blocks: bad arguments");
    const size_t jb = dl_pack_job_bytes();
    const char *sw = dl_switch(DL_SW_PACK_TILED);           // A/B switch: 0 = every job in the chunk-per-thread form (the r01-r05 kernel)
    const bool tiled = !(sw && sw[0] == '0');
    long n = 0;
    for (int j = 0; j < count; ++j) {
        PackArgs a;
        memcpy(&a, (const char *)jobs_host + (size_t)j * jb, sizeof(a));
        if (a.kstride % 8) DL_FAIL("dl_pack_batch_blocks: job %d: kstride %d is not a multiple of 8", j, a.kstride);
        const long chunks = (long)a.rows_pad * (a.kstride / 8);
        if (chunks > 0x7fffffffL) DL_FAIL("dl_pack_batch_blocks: job %d is too large", j);
        if (j >= PT_TILED_FLAG) DL_FAIL("dl_pack_batch_blocks: too many jobs");
        if (tiled && pack_tiled_ok(a)) {
            const long tiles = pack_tile_count(a);
            for (long t = 0; t < tiles; ++t, ++n)
                if (block_tab_host) { block_tab_host[2 * n] = j | PT_TILED_FLAG; block_tab_host[2 * n + 1] = (int32_t)t; }
            continue;
        }
        for (long c = 0; c < chunks; c += PACK_CHUNKS_PER_BLOCK, ++n)
            if (block_tab_host) { block_tab_host[2 * n] = j; block_tab_host[2 * n + 1] = (int32_t)c; }
    }
    if (n > 0x7fffffffL) DL_FAIL("dl_pack_batch_blocks: too many blocks");
    return (int)n;
}

extern "C" int dl_pack_weights_batch(const void *jobs_dev, const int32_t *block_tab_dev, int nblocks, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (nblocks <= 0) return 0;
    if (!jobs_dev || !block_tab_dev) DL_FAIL("dl_pack_weights_batch: null table");
    static_assert(sizeof(PackArgs) % sizeof(int) == 0, "job records are copied to LDS as ints");
    hipLaunchKernelGGL(pack_weights_batch_kernel, dim3(nblocks), dim3(256), 0, stream, reinterpret_cast<const char *>(jobs_dev),
                       dl_pack_job_bytes(), reinterpret_cast<const int2 *>(block_tab_dev));
    DL_CHECK_LAUNCH("dl_pack_weights_batch");
    return 0;
}

// ---- NCHW fp32 -> NHWC (channels [c0, c0+C) of a padded buffer; optional zeroing of channels [c0+C, zero_pad_to))
template <typename T>
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float *src, int N, int C, int H, int W, T *dst, int pstride, int c0,
                                                           int zero_to) {
    const size_t hw = (size_t)H * W, npix = (size_t)N * hw;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
        const size_t n = p / hw, r = p % hw;
        T *d = dst + p * pstride + c0;
        for (int c = 0; c < C; ++c) store1<T>(d + c, src[(n * C + c) * hw + r]);
        for (int c = C; c0 + c < zero_to; ++c) store1<T>(d + c, 0.f);
    }
}
template <typename T>
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const T *src, int pstride, int c0, float *dst, int N, int C, int H, int W) {
    const size_t hw = (size_t)H * W, total = (size_t)N * C * hw;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i % hw, c = (i / hw) % C, n = i / (hw * C);
        dst[i] = load1<T>(src + (n * hw + r) * pstride + c0 + c);
    }
}

extern "C" int dl_nchw_to_nhwc(const float *src, int N, int C, int H, int W, int dtype, void *dst, int dst_pstride, int dst_c0,
                               int zero_pad_to, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!src || !dst) DL_FAIL("dl_nchw_to_nhwc: null argument");
    const size_t npix = (size_t)N * H * W;
    const int blocks = (int)min((size_t)4096, (npix + 255) / 256);
    if (dtype == DL_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(blocks), dim3(256), 0, stream, src, N, C, H, W, (float *)dst, dst_pstride, dst_c0, zero_pad_to);
    else if (dtype == DL_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, src, N, C, H, W, (bf16_t *)dst, dst_pstride, dst_c0, zero_pad_to);
    else DL_FAIL("dl_nchw_to_nhwc: dtype %d", dtype);
    DL_CHECK_LAUNCH("dl_nchw_to_nhwc");
    return 0;
}

extern "C" int dl_nhwc_to_nchw(int dtype, const void *src, int src_pstride, int src_c0, float *dst, int N, int C, int H, int W,
                               void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!src || !dst) DL_FAIL("dl_nhwc_to_nchw: null argument");
    const size_t total = (size_t)N * C * H * W;
    const int blocks = (int)min((size_t)4096, (total + 255) / 256);
    if (dtype == DL_F32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float *)src, src_pstride, src_c0, dst, N, C, H, W);
    else if (dtype == DL_BF16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, (const bf16_t *)src, src_pstride, src_c0, dst, N, C, H, W);
    else DL_FAIL("dl_nhwc_to_nchw: dtype %d", dtype);
    DL_CHECK_LAUNCH("dl_nhwc_to_nchw");
    return 0;
}
