// postproc.hip -- the segmentation post-processing of the reference (deepliif/postprocessing.py:163-362, 923-1071) on the GPU.
// Integer / byte work, HBM- and latency-bound: per-pixel classification, two connected-component labellings, per-component reductions.
// The reference is sequential in-place code (raster loops and flood-fill stacks); the results it defines are order-free except for four
// "first in raster order" rules, which become minima over pixel indices here (see include/deepliif_hip.h, dl_pp_cells / dl_pp_finish):
//   * background = the 4-connected UNKNOWN components that touch the image border (mark_background's fixed point);
//   * a cell = an 8-connected component of non-background pixels; its root in the union-find forest below is its SMALLEST pixel index
//     = the first pixel the reference's raster scan meets = the pixel it later paints with the border label; cells are listed by root;
//   * a background pixel next to several cells belongs to the cell listed first; a background pixel next to several border pixels takes
//     the class of the first of them in raster order.
// Labelling: lock-free union-find with atomicMin links (larger root -> smaller root), one merge pass over the "backward" neighbours,
// one flatten pass.  Everything is deterministic: minima, integer sums, ordered compaction (rocPRIM select).
#include "common.h"
#include <hipcub/hipcub.hpp>

#define PP_UNKNOWN 50
#define PP_POSITIVE 200
#define PP_NEGATIVE 150
#define PP_BACKGROUND 0
#define PP_CELL 100
#define PP_BORDER_POS 220
#define PP_BORDER_NEG 170

typedef unsigned long long u64;

// ---------------------------------------------------------------------------------------------------- union-find
__device__ __forceinline__ int uf_find(const int *L, int a) {
    int b = __atomic_load_n(L + a, __ATOMIC_RELAXED);
    while (b != a) { a = b; b = __atomic_load_n(L + a, __ATOMIC_RELAXED); }
    return a;
}
__device__ __forceinline__ void uf_union(int *L, int a, int b) {
    bool done;
    do {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a < b) { const int old = atomicMin(L + b, a); done = (old == b); b = old; }
        else if (b < a) { const int old = atomicMin(L + a, b); done = (old == a); a = old; }
        else done = true;
    } while (!done);
}

// Initial label of an active pixel: the first pixel of its horizontal run INSIDE this wavefront (64 consecutive pixel indices, cut at row
// starts).  A run of n pixels then needs n/64 horizontal links instead of n, and every find() starts one hop from a run head: the image
// border's UNKNOWN region (usually one giant component) made the plain "label = own index" start 12x slower on a 2048 x 2048 image.
__device__ __forceinline__ int pp_run_head(bool active, int p, int x) {
    const unsigned long long act = __ballot(active);
    const unsigned long long brk = ~act | __ballot(x == 0);        // lanes that cannot be continued INTO: inactive, or first of a row
    const int lane = threadIdx.x & 63;
    if (!active) return -1;
    // highest lane l <= lane that starts a run: l is active and (l == 0 or lane l-1 inactive or x(l) == 0)
    const unsigned long long starts = act & ((~act << 1) | 1ull | __ballot(x == 0));
    const unsigned long long below = starts & (lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1));
    (void)brk;
    return p - (lane - (63 - __builtin_clzll(below)));
}

// mask from the segmentation probabilities (create_posneg_mask :163-190) + labels of the UNKNOWN pixels for the background pass
__global__ void __launch_bounds__(256) pp_mask_kernel(const uint8_t *seg, size_t seg_rs, int H, int W, int thresh, uint8_t *mask, int *label) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool in = p < H * W;
    const int y = in ? p / W : 0, x = in ? p - y * W : 1;
    uint8_t m = PP_BACKGROUND;
    if (in) {
        const uint8_t *s = seg + (size_t)y * seg_rs + (size_t)x * 3;
        const int r = s[0], g = s[1], b = s[2];
        m = PP_UNKNOWN;
        if (r + b > thresh && g <= 80) m = (r >= b) ? PP_POSITIVE : PP_NEGATIVE;
    }
    const int head = pp_run_head(in && m == PP_UNKNOWN, p, x);
    if (!in) return;
    mask[p] = m;
    label[p] = head;
}

// merge with the already-visited neighbours: left / up (4-connectivity), + up-left / up-right (8-connectivity)
template <int CONN>
__global__ void __launch_bounds__(256) pp_merge_kernel(int *label, int H, int W) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W || label[p] < 0) return;
    const int y = p / W, x = p - y * W;
    // horizontal link only where the initial run (pp_run_head) was cut: at a wavefront boundary
    if (x > 0 && (threadIdx.x & 63) == 0 && label[p - 1] >= 0) uf_union(label, p, p - 1);
    if (y > 0) {
        if (label[p - W] >= 0) uf_union(label, p, p - W);
        if (CONN == 8) {
            if (x > 0 && label[p - W - 1] >= 0) uf_union(label, p, p - W - 1);
            if (x + 1 < W && label[p - W + 1] >= 0) uf_union(label, p, p - W + 1);
        }
    }
}
__global__ void __launch_bounds__(256) pp_flatten_kernel(int *label, int n) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n || label[p] < 0) return;
    label[p] = uf_find(label, p);
}

// mark_background (:193-232): components of UNKNOWN pixels that reach the border
__global__ void __launch_bounds__(256) pp_border_flag_kernel(const int *label, int H, int W, int *flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * (H + W)) return;
    int p;
    if (i < W) p = i;                                       // top row
    else if (i < 2 * W) p = (H - 1) * W + (i - W);          // bottom row
    else if (i < 2 * W + H) p = (i - 2 * W) * W;            // left column
    else p = (i - 2 * W - H) * W + (W - 1);                 // right column
    const int r = label[p];
    if (r >= 0) flag[r] = 1;
}
// background applied; labels re-initialised for the cell pass (everything that is not background)
__global__ void __launch_bounds__(256) pp_background_kernel(uint8_t *mask, int *label, const int *flag, int n, int W) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool in = p < n;
    uint8_t m = PP_BACKGROUND;
    if (in) {
        m = mask[p];
        if (m == PP_UNKNOWN && flag[label[p]]) { m = PP_BACKGROUND; mask[p] = m; }
    }
    const int head = pp_run_head(in && m != PP_BACKGROUND, p, in ? p % W : 1);
    if (in) label[p] = head;
}

// per-cell reductions of compute_cell_mapping (:235-308): size, positive / negative pixels, marker (max, or sum for the optical density),
// coordinate sums; every non-background pixel becomes CELL
struct PPStats { int *cnt, *npos, *nneg; long long *mval; u64 *sx, *sy; };
__global__ void __launch_bounds__(256) pp_stats_kernel(uint8_t *mask, const int *label, int H, int W, const uint8_t *marker, size_t marker_rs,
                                                       const double *od_lut, PPStats s) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int r = label[p];
    if (r < 0) return;
    const int y = p / W, x = p - y * W;
    const uint8_t m = mask[p];
    atomicAdd(s.cnt + r, 1);
    if (m == PP_POSITIVE) atomicAdd(s.npos + r, 1);
    else if (m == PP_NEGATIVE) atomicAdd(s.nneg + r, 1);
    atomicAdd(s.sx + r, (u64)x);
    atomicAdd(s.sy + r, (u64)y);
    if (marker) {
        const uint8_t *q = marker + (size_t)y * marker_rs + (size_t)x * 3;
        if (od_lut) {          // create_od_image (:123-138): round(100 * (lut[r] + lut[g] + lut[b])), summed over the cell
            const double v = (od_lut[q[0]] + od_lut[q[1]]) + od_lut[q[2]];
            atomicAdd((u64 *)(s.mval + r), (u64)(long long)rint(v * 100.0));
        } else {               // to_array(marker, grayscale=True) (:98-120): the maximum channel; the cell keeps the maximum
            const int v = max((int)q[0], max((int)q[1], (int)q[2]));
            atomicMax(s.mval + r, (long long)v);
        }
    }
    mask[p] = PP_CELL;
}

struct PPHasCell {
    const int *cnt;
    __host__ __device__ bool operator()(const int &i) const { return cnt[i] > 0; }
};

// one row per component, in root order: {size, positive pixels, negative pixels, marker value (max or sum), first x, first y, sum x, sum y}
__global__ void __launch_bounds__(256) pp_gather_kernel(const int *roots, const int *n_ptr, int max_cells, int W, PPStats s, long long *cells, int *cellidx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = *n_ptr;
    if (i >= n) return;
    const int r = roots[i];
    cellidx[r] = i;
    if (i >= max_cells) return;
    long long *o = cells + (size_t)i * 8;
    o[0] = s.cnt[r]; o[1] = s.npos[r]; o[2] = s.nneg[r]; o[3] = s.mval[r];
    o[4] = r % W; o[5] = r / W; o[6] = (long long)s.sx[r]; o[7] = (long long)s.sy[r];
}

// 256-bin histogram of the non-zero gray (maximum-channel) marker values, for the default marker threshold (:450-488)
__global__ void __launch_bounds__(256) pp_hist_kernel(const uint8_t *marker, size_t marker_rs, int H, int W, u64 *hist) {
    __shared__ unsigned local[256];
    local[threadIdx.x] = 0;
    __syncthreads();
    for (int p = blockIdx.x * 256 + threadIdx.x; p < H * W; p += gridDim.x * 256) {
        const int y = p / W, x = p - y * W;
        const uint8_t *q = marker + (size_t)y * marker_rs + (size_t)x * 3;
        atomicAdd(local + max((int)q[0], max((int)q[1], (int)q[2])), 1u);
    }
    __syncthreads();
    if (local[threadIdx.x]) atomicAdd(hist + threadIdx.x, (u64)local[threadIdx.x]);
}

// create_cell_classification (:923-1000).  code[cell] : 0 = not counted, 1 = positive, 2 = negative
__global__ void __launch_bounds__(256) pp_classify_kernel(const uint8_t *mask, const int *label, const int *cellidx, const uint8_t *code, int H, int W,
                                                          uint8_t *out) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const uint8_t m = mask[p];
    uint8_t o = m;
    if (m == PP_CELL) {
        const int r = label[p];
        const uint8_t c = code[cellidx[r]];
        if (c) o = (p == r) ? (c == 1 ? PP_BORDER_POS : PP_BORDER_NEG) : (c == 1 ? PP_POSITIVE : PP_NEGATIVE);
    } else if (m == PP_BACKGROUND) {
        // border of the first-listed counted cell that has a NON-seed pixel 4-adjacent to p (the seed pixel never expands the border)
        const int y = p / W, x = p - y * W;
        int best = 0x7fffffff;
        uint8_t bc = 0;
        const int qs[4] = {x > 0 ? p - 1 : -1, y > 0 ? p - W : -1, x + 1 < W ? p + 1 : -1, y + 1 < H ? p + W : -1};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = qs[k];
            if (q < 0 || mask[q] != PP_CELL) continue;
            const int r = label[q];
            if (q == r || r >= best) continue;
            const uint8_t c = code[cellidx[r]];
            if (c) { best = r; bc = c; }
        }
        if (bc) o = (bc == 1) ? PP_BORDER_POS : PP_BORDER_NEG;
    }
    out[p] = o;
}

// enlarge_cell_boundaries (:1003-1030): background takes the class of its first border neighbour (8-neighbourhood, raster order)
__global__ void __launch_bounds__(256) pp_enlarge_kernel(const uint8_t *in, int H, int W, uint8_t *out) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    uint8_t o = in[p];
    if (o == PP_BACKGROUND) {
        const int y = p / W, x = p - y * W;
        for (int dy = -1; dy <= 1 && o == PP_BACKGROUND; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                if ((dy == 0 && dx == 0) || (unsigned)(y + dy) >= (unsigned)H || (unsigned)(x + dx) >= (unsigned)W) continue;
                const uint8_t v = in[p + dy * W + dx];
                if (v == PP_BORDER_POS || v == PP_BORDER_NEG) { o = v; break; }
            }
    }
    out[p] = o;
}

// create_final_images (:1033-1071)
__global__ void __launch_bounds__(256) pp_final_kernel(const uint8_t *orig, size_t orig_rs, const uint8_t *mask, int H, int W,
                                                       uint8_t *overlay, size_t ov_rs, uint8_t *refined, size_t rf_rs) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    const uint8_t *s = orig + (size_t)y * orig_rs + (size_t)x * 3;
    uint8_t *ov = overlay + (size_t)y * ov_rs + (size_t)x * 3, *rf = refined + (size_t)y * rf_rs + (size_t)x * 3;
    uint8_t o0 = s[0], o1 = s[1], o2 = s[2], r0 = 0, r1 = 0, r2 = 0;
    const uint8_t m = mask[p];
    if (m == PP_BORDER_POS) { o0 = 255; o1 = 0; o2 = 0; r1 = 255; }
    else if (m == PP_BORDER_NEG) { o0 = 0; o1 = 0; o2 = 255; r1 = 255; }
    else if (m == PP_POSITIVE) r0 = 255;
    else if (m == PP_NEGATIVE) r2 = 255;
    ov[0] = o0; ov[1] = o1; ov[2] = o2;
    rf[0] = r0; rf[1] = r1; rf[2] = r2;
}

// ---------------------------------------------------------------------------------------------------- host side
// workspace layout (bytes, all regions 256-byte aligned):
//   cnt | npos | nneg (int32[HW] each)  mval | sx | sy (int64[HW] each)  cellidx (int32[HW])  roots (int32[HW])  n (int32)  hist (u64[256])
//   mask2 (u8[HW])  select temp
static size_t pp_align(size_t v) { return (v + 255) & ~(size_t)255; }
struct PPLayout {
    size_t cnt, npos, nneg, mval, sx, sy, cellidx, roots, n, hist, mask2, temp, temp_bytes, total;
};
static int pp_layout(int H, int W, PPLayout *l) {
    const size_t hw = (size_t)H * W;
    size_t off = 0;
    l->cnt = off; off += pp_align(hw * 4);
    l->npos = off; off += pp_align(hw * 4);
    l->nneg = off; off += pp_align(hw * 4);
    l->mval = off; off += pp_align(hw * 8);
    l->sx = off; off += pp_align(hw * 8);
    l->sy = off; off += pp_align(hw * 8);
    l->cellidx = off; off += pp_align(hw * 4);
    l->roots = off; off += pp_align(hw * 4);
    l->n = off; off += 256;
    l->hist = off; off += pp_align(256 * 8);
    l->mask2 = off; off += pp_align(hw);
    size_t tb = 0;
    hipcub::CountingInputIterator<int> it(0);
    PPHasCell pred{nullptr};
    if (hipcub::DeviceSelect::If(nullptr, tb, it, (int *)nullptr, (int *)nullptr, (int)hw, pred) != hipSuccess) return -1;
    l->temp = off; l->temp_bytes = tb; off += pp_align(tb);
    l->total = off;
    return 0;
}

extern "C" size_t dl_pp_ws_bytes(int H, int W) {
    PPLayout l;
    if (H <= 0 || W <= 0 || (size_t)H * W >= ((size_t)1 << 31) || pp_layout(H, W, &l)) return 0;
    return l.total;
}

extern "C" int dl_pp_cells(const void *seg, size_t seg_row_stride, const void *marker, size_t marker_row_stride, const double *od_lut,
                           int H, int W, int seg_thresh, void *mask, int *label, void *ws, long long *cells, int max_cells,
                           int *n_cells, unsigned long long *hist, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!seg || !mask || !label || !ws || !cells || !n_cells) DL_FAIL("dl_pp_cells: null argument");
    if (H <= 0 || W <= 0 || (size_t)H * W >= ((size_t)1 << 31)) DL_FAIL("dl_pp_cells: empty problem or too large (H=%d W=%d)", H, W);
    if (max_cells < 0) DL_FAIL("dl_pp_cells: max_cells=%d", max_cells);
    PPLayout l;
    if (pp_layout(H, W, &l)) DL_FAIL("dl_pp_cells: workspace query failed");
    char *w = (char *)ws;
    const int n = H * W, blocks = (n + 255) / 256;
    PPStats s{(int *)(w + l.cnt), (int *)(w + l.npos), (int *)(w + l.nneg), (long long *)(w + l.mval), (u64 *)(w + l.sx), (u64 *)(w + l.sy)};
    int *cellidx = (int *)(w + l.cellidx), *roots = (int *)(w + l.roots), *n_dev = (int *)(w + l.n);
    if (hipMemsetAsync(w, 0, l.roots, stream) != hipSuccess) DL_FAIL("dl_pp_cells: memset");           // statistics + cellidx
    hipLaunchKernelGGL(pp_mask_kernel, dim3(blocks), dim3(256), 0, stream, (const uint8_t *)seg, seg_row_stride, H, W, seg_thresh, (uint8_t *)mask, label);
    hipLaunchKernelGGL(pp_merge_kernel<4>, dim3(blocks), dim3(256), 0, stream, label, H, W);
    hipLaunchKernelGGL(pp_flatten_kernel, dim3(blocks), dim3(256), 0, stream, label, n);
    int *flag = s.cnt;                                              // reused: zero now, zeroed again below
    hipLaunchKernelGGL(pp_border_flag_kernel, dim3((2 * (H + W) + 255) / 256), dim3(256), 0, stream, label, H, W, flag);
    hipLaunchKernelGGL(pp_background_kernel, dim3(blocks), dim3(256), 0, stream, (uint8_t *)mask, label, flag, n, W);
    if (hipMemsetAsync(flag, 0, (size_t)n * 4, stream) != hipSuccess) DL_FAIL("dl_pp_cells: memset");
    hipLaunchKernelGGL(pp_merge_kernel<8>, dim3(blocks), dim3(256), 0, stream, label, H, W);
    hipLaunchKernelGGL(pp_flatten_kernel, dim3(blocks), dim3(256), 0, stream, label, n);
    hipLaunchKernelGGL(pp_stats_kernel, dim3(blocks), dim3(256), 0, stream, (uint8_t *)mask, label, H, W, (const uint8_t *)marker, marker_row_stride, od_lut, s);
    hipcub::CountingInputIterator<int> it(0);
    PPHasCell pred{s.cnt};
    size_t tb = l.temp_bytes;
    if (hipcub::DeviceSelect::If(w + l.temp, tb, it, roots, n_dev, n, pred, stream) != hipSuccess) DL_FAIL("dl_pp_cells: select failed");
    hipLaunchKernelGGL(pp_gather_kernel, dim3(blocks), dim3(256), 0, stream, roots, n_dev, max_cells, W, s, cells, cellidx);
    if (hipMemcpyAsync(n_cells, n_dev, sizeof(int), hipMemcpyDeviceToDevice, stream) != hipSuccess) DL_FAIL("dl_pp_cells: copy");
    if (hist) {
        if (!marker) DL_FAIL("dl_pp_cells: the histogram needs the marker image");
        if (hipMemsetAsync(w + l.hist, 0, 256 * 8, stream) != hipSuccess) DL_FAIL("dl_pp_cells: memset");
        hipLaunchKernelGGL(pp_hist_kernel, dim3(blocks < 1024 ? blocks : 1024), dim3(256), 0, stream, (const uint8_t *)marker, marker_row_stride, H, W, (u64 *)(w + l.hist));
        if (hipMemcpyAsync(hist, w + l.hist, 256 * 8, hipMemcpyDeviceToDevice, stream) != hipSuccess) DL_FAIL("dl_pp_cells: copy");
    }
    DL_CHECK_LAUNCH("dl_pp_cells");
    return 0;
}

extern "C" int dl_pp_finish(const void *orig, size_t orig_row_stride, void *mask, const int *label, const void *ws, const void *code, int n_cells,
                            int H, int W, void *overlay, size_t overlay_row_stride, void *refined, size_t refined_row_stride, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!orig || !mask || !label || !ws || !overlay || !refined || (n_cells > 0 && !code)) DL_FAIL("dl_pp_finish: null argument");
    if (H <= 0 || W <= 0 || (size_t)H * W >= ((size_t)1 << 31)) DL_FAIL("dl_pp_finish: empty problem or too large (H=%d W=%d)", H, W);
    PPLayout l;
    if (pp_layout(H, W, &l)) DL_FAIL("dl_pp_finish: workspace query failed");
    const char *w = (const char *)ws;
    const int n = H * W, blocks = (n + 255) / 256;
    uint8_t *m1 = (uint8_t *)mask, *m2 = (uint8_t *)const_cast<char *>(w + l.mask2);
    static const uint8_t *no_code = nullptr;
    hipLaunchKernelGGL(pp_classify_kernel, dim3(blocks), dim3(256), 0, stream, m1, label, (const int *)(w + l.cellidx),
                       n_cells > 0 ? (const uint8_t *)code : no_code, H, W, m2);
    hipLaunchKernelGGL(pp_enlarge_kernel, dim3(blocks), dim3(256), 0, stream, m2, H, W, m1);
    hipLaunchKernelGGL(pp_enlarge_kernel, dim3(blocks), dim3(256), 0, stream, m1, H, W, m2);
    if (hipMemcpyAsync(m1, m2, (size_t)n, hipMemcpyDeviceToDevice, stream) != hipSuccess) DL_FAIL("dl_pp_finish: copy");
    hipLaunchKernelGGL(pp_final_kernel, dim3(blocks), dim3(256), 0, stream, (const uint8_t *)orig, orig_row_stride, m1, H, W,
                       (uint8_t *)overlay, overlay_row_stride, (uint8_t *)refined, refined_row_stride);
    DL_CHECK_LAUNCH("dl_pp_finish");
    return 0;
}

// ---------------------------------------------------------------------------------------------------- host-only helper
// calculate_default_size_threshold's kernel density estimate (postprocessing.py:365-447) over the cell list: count bins, bandwidth 1,
// kde[i] = float32( sum_j exp(-((i*step - v_j)^2 / 2)) / sqrt(2 pi)  / n ), summed over j in list order in float64 -- the reference's
// loop, expression by expression (libm exp, no contraction), because its float32 comparisons kde[i] < kde[i-1] decide the threshold.
// Plain host code (no GPU): the Python loop it replaces spent 115 ms on 2 648 cells.  Returns the index of the first local minimum
// (1 if there is none) and the step through *step_out.
#include <math.h>
extern "C" int dl_pp_kde_first_minimum(const double *values, int n, int count, double *step_out, float *kde_out) {
#pragma clang fp contract(off)
    if (!values || n <= 0 || count < 3 || !step_out) DL_FAIL("dl_pp_kde_first_minimum: bad argument");
    const double inv = 1 / sqrt(2 * M_PI);
    double vmax = values[0];
    for (int j = 1; j < n; ++j) vmax = values[j] > vmax ? values[j] : vmax;
    const double step = (vmax + 1) / count;
    float *kde = kde_out ? kde_out : (float *)malloc(sizeof(float) * (size_t)count);
    if (!kde) DL_FAIL("dl_pp_kde_first_minimum: out of memory");
    for (int i = 0; i < count; ++i) {
        const double x = i * step;
        double total = 0;
        for (int j = 0; j < n; ++j) {
            const double val = (x - values[j]) * 1.0;
            total += exp(-(val * val / 2)) * inv;
        }
        kde[i] = (float)(total / (n * 1.0));
    }
    int idx = 1;
    for (int i = 1; i < count - 1; ++i)
        if (kde[i] < kde[i - 1] && kde[i] < kde[i + 1]) { idx = i; break; }
    if (!kde_out) free(kde);
    *step_out = step;
    return idx;
}
