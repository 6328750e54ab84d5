// tiles.hip -- uint8 image <-> engine tile batches: the crop / is_empty / stitch steps either side of the generator DAG
// (deepliif/util/__init__.py:129-331 InferenceTiler, deepliif/models/__init__.py:391-396 is_empty,
// deepliif/data/__init__.py:133-138 transform, deepliif/util/util.py:117-139 tensor2im).  Byte / integer work, HBM-bound:
// one thread per pixel, 16-byte (bf16) or 2 x 16-byte (fp32) NHWC stores, no LDS.
#include "common.h"
#include <stdlib.h>

// reflect-periodic source coordinate: an image narrower than a patch is widened by appending mirrored copies
// (util/__init__.py:196-211) -> period 2*n: c, then 2n-1-c
__device__ __forceinline__ int mirror_coord(int c, int n) {
    if (c < n) return c;
    const int m = c % (2 * n);
    return m < n ? m : 2 * n - 1 - m;
}

struct TileSrc {
    const uint8_t *img[DL_TILE_MAX_SRC];
    long long row_stride[DL_TILE_MAX_SRC];
};

// fetch the RGB bytes of tile pixel (ty, tx) of tile `origin`, honouring the solid border (pad) and the mirror extension
__device__ __forceinline__ void tile_pixel(const uint8_t *img, long long row_stride, int H0, int W0, int ox, int oy, int ty, int tx, int patch,
                                           int pad, uint32_t pad_rgb, int &r, int &g, int &b) {
    const int py = ty - pad, px = tx - pad;
    if (py < 0 || px < 0 || py >= patch || px >= patch) {
        r = pad_rgb & 0xff; g = (pad_rgb >> 8) & 0xff; b = (pad_rgb >> 16) & 0xff;
        return;
    }
    const int sy = mirror_coord(oy + py, H0), sx = mirror_coord(ox + px, W0);
    const uint8_t *p = img + (long long)sy * row_stride + 3ll * sx;
    r = p[0]; g = p[1]; b = p[2];
}

template <typename T>
__global__ void tile_gather_kernel(TileSrc src, int n_src, int H0, int W0, const int32_t *__restrict__ origins, int tile, int pad, uint32_t pad_rgb,
                                   const float *__restrict__ lut, T *__restrict__ out, int out_pstride, int out_cp) {
    const int t = blockIdx.y;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= tile * tile) return;
    const int ty = pix / tile, tx = pix - ty * tile;
    const int ox = origins[2 * t], oy = origins[2 * t + 1];
    const int patch = tile - 2 * pad;
    T *o = out + ((long long)t * tile * tile + pix) * out_pstride;
    for (int c8 = 0; c8 < out_cp; c8 += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
#pragma unroll
        for (int s = 0; s < DL_TILE_MAX_SRC; ++s) {
            const int c = 3 * s;                     // channels 3s .. 3s+2 of the concatenated input (torch.cat(dim=1), models/__init__.py:279)
            if (s < n_src && c + 2 >= c8 && c < c8 + 8) {
                int r, g, b;
                tile_pixel(src.img[s], src.row_stride[s], H0, W0, ox, oy, ty, tx, patch, pad, pad_rgb, r, g, b);
                const int rgb[3] = {r, g, b};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int cc = c + k - c8;
                    if (cc >= 0 && cc < 8) v[cc] = lut[rgb[k]];
                }
            }
        }
        Vec8<T>::store(o + c8, v);
    }
}

// (count, sum, sum of squares) of the gray values in 1..254 of every tile: the exact integer form of image_variance_gray
// (PIL convert('L'): L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16).  Integer atomics -> order-independent, deterministic.
__global__ void tile_gray_stats_kernel(const uint8_t *__restrict__ img, long long row_stride, int H0, int W0, const int32_t *__restrict__ origins, int tile,
                                       int pad, uint32_t pad_rgb, unsigned long long *__restrict__ stats) {
    const int t = blockIdx.y;
    const int ox = origins[2 * t], oy = origins[2 * t + 1];
    const int patch = tile - 2 * pad;
    unsigned int cnt = 0, s1 = 0;
    unsigned long long s2 = 0;
    for (int pix = blockIdx.x * blockDim.x + threadIdx.x; pix < tile * tile; pix += gridDim.x * blockDim.x) {
        const int ty = pix / tile, tx = pix - ty * tile;
        int r, g, b;
        tile_pixel(img, row_stride, H0, W0, ox, oy, ty, tx, patch, pad, pad_rgb, r, g, b);
        const unsigned int L = (19595u * r + 38470u * g + 7471u * b + 0x8000u) >> 16;
        if (L != 0u && L != 255u) { cnt += 1; s1 += L; s2 += (unsigned long long)(L * L); }
    }
    unsigned long long c64 = cnt, a64 = s1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c64 += __shfl_xor(c64, o, 64);
        a64 += __shfl_xor(a64, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&stats[3 * t + 0], c64);
        atomicAdd(&stats[3 * t + 1], a64);
        atomicAdd(&stats[3 * t + 2], s2);
    }
}

// rect record (8 x int32): slot (tile index in the batch, or -1 = constant colour), l, t (tile-local), w, h, px, py (image), rgb
template <typename T>
__global__ void tile_paste_kernel(const T *__restrict__ tiles, int in_pstride, int tile, const int32_t *__restrict__ rects, uint8_t *__restrict__ dst,
                                  long long dst_row_stride) {
    const int32_t *rc = rects + 8 * blockIdx.y;
    const int slot = rc[0], l = rc[1], tp = rc[2], w = rc[3], h = rc[4], px = rc[5], py = rc[6];
    const uint32_t rgb = (uint32_t)rc[7];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += gridDim.x * blockDim.x) {
        const int y = i / w, x = i - y * w;
        uint8_t *o = dst + (long long)(py + y) * dst_row_stride + 3ll * (px + x);
        if (slot < 0) {
            o[0] = rgb & 0xff; o[1] = (rgb >> 8) & 0xff; o[2] = (rgb >> 16) & 0xff;
            continue;
        }
        const T *p = tiles + ((long long)slot * tile * tile + (long long)(tp + y) * tile + (l + x)) * in_pstride;
        float v[8];
        Vec8<T>::load(p, v);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            // tensor2im (util/util.py:132-135): (x + 1) / 2.0 * 255.0 in fp32, then astype(uint8) = truncation; no contraction
            const float f = __fmul_rn(__fmul_rn(__fadd_rn(v[k], 1.0f), 0.5f), 255.0f);
            o[k] = (uint8_t)(int)f;
        }
    }
}

// tiles / rectangles per launch (gridDim.y <= 65535); DL_TILE_GRID_Y=<n> overrides it so that the chunked path can be tested on small regions
static int tile_grid_y() {
    static const int v = [] { const char *e = DL_DEV_ENV("DL_TILE_GRID_Y"); const int n = e ? atoi(e) : 0; return (n >= 1 && n <= 65535) ? n : 32768; }();
    return v;
}
#define DL_TILE_GRID_Y (tile_grid_y())

extern "C" int dl_tile_gather_u8(const void *const *imgs, const int64_t *row_strides, int n_src, int H0, int W0, const int32_t *origins,
                                 int n_tiles, int tile, int pad, uint32_t pad_rgb, const float *lut, int out_dtype, void *out,
                                 int out_pstride, int out_cp, void *stream) {
    if (n_tiles <= 0 || tile <= 0 || H0 <= 0 || W0 <= 0) DL_FAIL("dl_tile_gather_u8: empty problem (n_tiles=%d tile=%d image %dx%d)", n_tiles, tile, W0, H0);
    if (n_src < 1 || n_src > DL_TILE_MAX_SRC) DL_FAIL("dl_tile_gather_u8: n_src=%d outside 1..%d", n_src, DL_TILE_MAX_SRC);
    if (out_cp % 8 || out_cp < 3 * n_src || out_pstride < out_cp) DL_FAIL("dl_tile_gather_u8: bad channel geometry (Cp=%d pstride=%d for %d source images)", out_cp, out_pstride, n_src);
    if (pad < 0 || 2 * pad >= tile) DL_FAIL("dl_tile_gather_u8: pad=%d does not fit tile=%d", pad, tile);
    TileSrc s;
    for (int i = 0; i < DL_TILE_MAX_SRC; ++i) {
        s.img[i] = (const uint8_t *)(i < n_src ? imgs[i] : imgs[0]);
        s.row_stride[i] = i < n_src ? row_strides[i] : row_strides[0];
    }
    // gridDim.y is limited to 65535: a region with more tiles (about 115k x 115k pixels at tile 512 / overlap 32) is launched in chunks
    for (int t0 = 0; t0 < n_tiles; t0 += DL_TILE_GRID_Y) {
        const int nt = n_tiles - t0 < DL_TILE_GRID_Y ? n_tiles - t0 : DL_TILE_GRID_Y;
        dim3 grid((tile * tile + 255) / 256, nt);
        const size_t o0 = (size_t)t0 * tile * tile * out_pstride;
        if (out_dtype == DL_BF16)
            hipLaunchKernelGGL(tile_gather_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, s, n_src, H0, W0, origins + 2 * (size_t)t0, tile, pad, pad_rgb, lut,
                               (bf16_t *)out + o0, out_pstride, out_cp);
        else
            hipLaunchKernelGGL(tile_gather_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, s, n_src, H0, W0, origins + 2 * (size_t)t0, tile, pad, pad_rgb, lut,
                               (float *)out + o0, out_pstride, out_cp);
        DL_CHECK_LAUNCH("dl_tile_gather_u8");
    }
    return 0;
}

extern "C" int dl_tile_gray_stats_u8(const void *img, int64_t row_stride, int H0, int W0, const int32_t *origins, int n_tiles, int tile, int pad,
                                     uint32_t pad_rgb, uint64_t *stats, void *stream) {
    if (n_tiles <= 0 || tile <= 0 || H0 <= 0 || W0 <= 0) DL_FAIL("dl_tile_gray_stats_u8: empty problem (n_tiles=%d tile=%d image %dx%d)", n_tiles, tile, W0, H0);
    if (pad < 0 || 2 * pad >= tile) DL_FAIL("dl_tile_gray_stats_u8: pad=%d does not fit tile=%d", pad, tile);
    hipError_t e = hipMemsetAsync(stats, 0, sizeof(uint64_t) * 3 * n_tiles, (hipStream_t)stream);
    if (e != hipSuccess) DL_FAIL("dl_tile_gray_stats_u8: memset failed: %s", hipGetErrorString(e));
    int bx = (tile * tile + 256 * 8 - 1) / (256 * 8);
    if (bx < 1) bx = 1;
    for (int t0 = 0; t0 < n_tiles; t0 += DL_TILE_GRID_Y) {
        const int nt = n_tiles - t0 < DL_TILE_GRID_Y ? n_tiles - t0 : DL_TILE_GRID_Y;
        hipLaunchKernelGGL(tile_gray_stats_kernel, dim3(bx, nt), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)img, (long long)row_stride, H0, W0,
                           origins + 2 * (size_t)t0, tile, pad, pad_rgb, (unsigned long long *)stats + 3 * (size_t)t0);
        DL_CHECK_LAUNCH("dl_tile_gray_stats_u8");
    }
    return 0;
}

extern "C" int dl_tile_paste_u8(int in_dtype, const void *tiles, int in_pstride, int tile, const int32_t *rects, int n_rects, void *dst, int64_t dst_row_stride,
                                void *stream) {
    if (n_rects <= 0 || tile <= 0) DL_FAIL("dl_tile_paste_u8: empty problem (n_rects=%d tile=%d)", n_rects, tile);
    if (in_pstride < 8) DL_FAIL("dl_tile_paste_u8: engine tiles have at least 8 padded channels (pstride=%d)", in_pstride);
    for (int r0 = 0; r0 < n_rects; r0 += DL_TILE_GRID_Y) {          // the rectangles carry their own tile index: only the list is chunked
        const int nr = n_rects - r0 < DL_TILE_GRID_Y ? n_rects - r0 : DL_TILE_GRID_Y;
        dim3 grid(64, nr);
        if (in_dtype == DL_BF16)
            hipLaunchKernelGGL(tile_paste_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t *)tiles, in_pstride, tile, rects + 8 * (size_t)r0, (uint8_t *)dst,
                               (long long)dst_row_stride);
        else
            hipLaunchKernelGGL(tile_paste_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)tiles, in_pstride, tile, rects + 8 * (size_t)r0, (uint8_t *)dst,
                               (long long)dst_row_stride);
        DL_CHECK_LAUNCH("dl_tile_paste_u8");
    }
    return 0;
}
