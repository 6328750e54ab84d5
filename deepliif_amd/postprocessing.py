"""Segmentation post-processing on the GPU -- the seam of deepliif/postprocessing.py.

`compute_final_results(orig, seg, marker, resolution, ...)` keeps the reference's signature and return value
(deepliif/postprocessing.py:1223-1304: overlay image, refined segmentation image, scoring dictionary) and its byte-exact results; the
per-pixel work, the two connected-component labellings and the per-cell reductions run in csrc/postproc.hip (dl_pp_cells,
dl_pp_finish).  What stays on the host is the arithmetic over the CELL LIST (hundreds to thousands of rows): noise thresholds, centroid
rounding, the default size threshold (a 500-bin KDE of sqrt(size), :365-447, repeated expression by expression so that its float64
comparisons agree) and the default marker threshold (:450-488, from a 256-bin histogram the GPU produces instead of np.percentile over
the image).  There is no CPU fallback: without the HIP library this module raises.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L
from . import ops

DEFAULT_SEG_THRESH = 120          # postprocessing.py:83
DEFAULT_NOISE_THRESH = 4          # postprocessing.py:84
_OD_LUT = [math.log10(255 / max(i, 1)) for i in range(256)]          # create_od_image's table (:123-130): entry 0 takes entry 1's value


def _to_u8_cuda(img, device) -> torch.Tensor:
    """PIL image / ndarray / tensor -> uint8 [H][W][3] CUDA tensor (to_array, :98-120: non-RGB PIL images are converted to RGB)"""
    if isinstance(img, torch.Tensor):
        t = img
    else:
        try:
            from PIL import Image
            if isinstance(img, Image.Image):
                img = np.asarray(img if img.mode == 'RGB' else img.convert('RGB'))
        except ImportError:
            pass
        t = torch.from_numpy(np.array(img, copy=True))            # (PIL-backed arrays are read-only)
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
        raise ValueError(f'expected a uint8 H x W x 3 image, got {tuple(t.shape)} {t.dtype}')
    t = t.to(device)
    return t if t.stride(2) == 1 and t.stride(1) == 3 else t.contiguous()


def calculate_large_noise_thresh(large_noise_thresh, resolution):
    """postprocessing.py:1125-1133"""
    if large_noise_thresh != 'default':
        return large_noise_thresh
    return {'10x': 1000, '20x': 4000}.get(resolution, 16000)


def calculate_default_size_threshold(cell_sizes, resolution='40x') -> int:
    """postprocessing.py:406-447 (+ create_kde :365-403): first local minimum of a Gaussian KDE (500 bins, bandwidth 1) of sqrt(size),
    clamped to the resolution's allowed range.  Same float64 expressions in the same order as the reference's loops."""
    sizes = np.asarray(cell_sizes, dtype=np.int64)
    if sizes.shape[0] <= 1:
        return 0
    values = np.ascontiguousarray(np.sqrt(sizes), dtype=np.float64)
    # the KDE loop itself (n x 500 exp calls in the reference's summation order) is host code inside the library: dl_pp_kde_first_minimum
    step = C.c_double(0.0)
    idx = int(L.load().dl_pp_kde_first_minimum(values.ctypes.data_as(C.c_void_p), int(values.shape[0]), 500, C.byref(step), None))
    if idx < 0:
        L.check(idx, 'dl_pp_kde_first_minimum')
    step = step.value
    t = (idx - 1) * step
    lo, dflt, hi = {'20x': (3, 4, 6), '10x': (2, 2, 3)}.get(resolution, (4, 7, 10))
    if t < lo:
        t = lo
    elif t > hi:
        t = dflt
    return round(t * t)


def percentile_from_histogram(hist, q: float) -> float:
    """np.percentile(values, q) (default 'linear' method) of the multiset {v repeated hist[v] times}, computed from the histogram the way
    numpy computes it from the sorted array (numpy/lib/_function_base_impl.py: _quantile, _get_indexes, _lerp): virtual index
    (n - 1) * (q / 100), neighbours at floor and floor + 1 (both the last element at or beyond the end), weight g = virtual - floor,
    a + (b - a) * g, taken from the far end (b - (b - a) * (1 - g)) when g >= 0.5."""
    hist = np.asarray(hist, dtype=np.int64)
    n = int(hist.sum())
    virtual = (n - 1) * np.true_divide(q, 100)
    prev = int(math.floor(virtual))
    nxt = prev + 1
    if virtual >= n - 1:
        prev = nxt = n - 1
    if virtual < 0:
        prev = nxt = 0
    gamma = virtual - (prev if virtual < n - 1 else -1)
    cum = np.cumsum(hist)
    a = float(np.searchsorted(cum, prev, side='right'))
    b = float(np.searchsorted(cum, nxt, side='right'))
    diff = b - a
    out = a + diff * gamma
    if gamma >= 0.5:
        out = b - diff * (1 - gamma)
    return float(out)


def calculate_default_marker_threshold(hist) -> int:
    """postprocessing.py:450-488 on the histogram of the gray marker image: 90 % of the 0.1 .. 99.9 percentile range of its non-zero pixels"""
    h = np.asarray(hist, dtype=np.int64).copy()
    h[0] = 0
    if int(h.sum()) == 0:
        lo = hi = 0
    else:
        lo, hi = round(percentile_from_histogram(h, 0.1)), round(percentile_from_histogram(h, 99.9))
    return round((hi - lo) * 0.9) + lo


class CellMap:
    """Device-side result of the cell mapping (get_cells_info, :311-362) plus the host-side cell list."""

    def __init__(self, mask, label, ws, rows, keep, cells, defaults, shape):
        self.mask, self.label, self.ws = mask, label, ws          # uint8 [H][W], int32 [H][W], workspace shared with dl_pp_finish
        self.rows, self.keep = rows, keep                         # every component's reductions (int64 [n][8]); which of them are cells
        self.cells, self.defaults, self.shape = cells, defaults, shape


def get_cells_info(seg, marker, resolution, noise_thresh, seg_thresh, large_noise_thresh, use_od=False, device=None) -> CellMap:
    """postprocessing.py:311-362.  `cells` = [(size, positive, marker value, first x, first y, centre x, centre y)] like the reference's."""
    be = ops.impl()
    lib = be.lib
    dev = torch.device(device) if device is not None else (seg.device if isinstance(seg, torch.Tensor) and seg.is_cuda else torch.device('cuda', torch.cuda.current_device()))
    seg_t = _to_u8_cuda(seg, dev)
    mk_t = _to_u8_cuda(marker, dev) if marker is not None else None
    H, W = int(seg_t.shape[0]), int(seg_t.shape[1])
    if mk_t is not None and tuple(mk_t.shape) != tuple(seg_t.shape):
        raise ValueError('seg and marker images differ in size')
    ops._need_cuda(seg_t, mk_t)
    nbytes = int(lib.dl_pp_ws_bytes(H, W))
    if nbytes == 0:
        raise RuntimeError('dl_pp_ws_bytes failed (empty or too large image)')
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    mask = torch.empty((H, W), dtype=torch.uint8, device=dev)
    label = torch.empty((H, W), dtype=torch.int32, device=dev)
    max_cells = min(H * W // 4 + 8, 1 << 22)                      # 8-connected components are at least two pixels apart
    rows_d = torch.empty((max_cells, 8), dtype=torch.int64, device=dev)
    n_d = torch.zeros(1, dtype=torch.int32, device=dev)
    hist_d = torch.zeros(256, dtype=torch.int64, device=dev) if (mk_t is not None and not use_od) else None
    lut_d = torch.tensor(_OD_LUT, dtype=torch.float64, device=dev) if (use_od and mk_t is not None) else None
    p = ops._ptr
    L.check(lib.dl_pp_cells(p(seg_t), seg_t.stride(0), p(mk_t), mk_t.stride(0) if mk_t is not None else 0, p(lut_d), H, W, int(seg_thresh),
                            p(mask), p(label), p(ws), p(rows_d), max_cells, p(n_d), p(hist_d), ops._stream()), 'dl_pp_cells')
    n = int(n_d.item())
    if n > max_cells:
        raise RuntimeError(f'{n} connected components exceed the cell table ({max_cells})')
    rows = rows_d[:n].cpu().numpy()
    count = rows[:, 0]
    keep = count > noise_thresh
    if large_noise_thresh is not None:
        keep &= count < large_noise_thresh
    k = rows[keep]
    kc = k[:, 0]
    # int(round(sum / count)): Python's round of the float64 quotient, i.e. round-half-even = np.rint of the same quotient
    cx, cy = np.rint(k[:, 6] / kc).astype(np.int64), np.rint(k[:, 7] / kc).astype(np.int64)
    if mk_t is None:
        mval = np.zeros(len(k), dtype=np.int64)
    elif use_od:
        mval = np.rint(k[:, 3] / kc).astype(np.int64)
    else:
        mval = k[:, 3]
    pos = k[:, 1] >= k[:, 2]
    cells = [(int(a), bool(b), int(c), int(d), int(e), int(f), int(g)) for a, b, c, d, e, f, g in zip(kc, pos, mval, k[:, 4], k[:, 5], cx, cy)]
    defaults = {'size_thresh': calculate_default_size_threshold([c[0] for c in cells], resolution)}
    if hist_d is not None:
        defaults['marker_thresh'] = calculate_default_marker_threshold(hist_d.cpu().numpy())
    return CellMap(mask, label, ws, rows, keep, cells, defaults, (H, W))


def create_cell_classification(cm: CellMap, size_thresh=0, marker_thresh=None, size_thresh_upper=None, od_thresh_lower=None, od_thresh_upper=None):
    """The per-cell decisions of postprocessing.py:923-1000 -> (code per component for dl_pp_finish, counts)"""
    code = np.zeros(len(cm.keep), dtype=np.uint8)
    num_pos = num_neg = 0
    it = iter(cm.cells)
    for i, kept in enumerate(cm.keep):
        if not kept:
            continue
        cell = next(it)
        if not (cell[0] > size_thresh and (size_thresh_upper is None or cell[0] < size_thresh_upper)):
            continue
        pos = cell[1]
        if marker_thresh is not None and cell[2] > marker_thresh:
            pos = True
        if od_thresh_lower is not None and cell[2] < od_thresh_lower:
            pos = False
        elif od_thresh_upper is not None and cell[2] > od_thresh_upper:
            pos = False
        code[i] = 1 if pos else 2
        num_pos, num_neg = num_pos + (1 if pos else 0), num_neg + (0 if pos else 1)
    return code, {'num_total': num_pos + num_neg, 'num_pos': num_pos, 'num_neg': num_neg}


def compute_final_results(orig, seg, marker, resolution, size_thresh='default', marker_thresh=None, size_thresh_upper=None,
                          seg_thresh=DEFAULT_SEG_THRESH, noise_thresh=DEFAULT_NOISE_THRESH, large_noise_thresh=None,
                          od_thresh_lower=None, od_thresh_upper=None, return_tensors: bool = False, device=None):
    """deepliif/postprocessing.py:1223-1304 -> (overlay, refined, scoring).  Images may be PIL images, uint8 ndarrays or uint8 CUDA
    tensors (H x W x 3); the results are ndarrays like the reference's unless return_tensors (then CUDA tensors, no device->host copy)."""
    large = calculate_large_noise_thresh(large_noise_thresh, resolution)
    use_od = od_thresh_lower is not None or od_thresh_upper is not None
    cm = get_cells_info(seg, orig if use_od else marker, resolution, noise_thresh, seg_thresh, large, use_od=use_od, device=device)
    if size_thresh is None:
        size_thresh = 0
    elif size_thresh == 'default':
        size_thresh = cm.defaults['size_thresh']
    if marker_thresh == 'default':
        marker_thresh = cm.defaults['marker_thresh']
    code, counts = create_cell_classification(cm, size_thresh, marker_thresh, size_thresh_upper, od_thresh_lower, od_thresh_upper)
    dev = cm.mask.device
    H, W = cm.shape
    orig_t = _to_u8_cuda(orig, dev)
    if tuple(orig_t.shape[:2]) != (H, W):
        raise ValueError('orig and seg images differ in size')
    overlay = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    refined = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    code_d = torch.from_numpy(code).to(dev) if len(code) else None
    p = ops._ptr
    L.check(ops.impl().lib.dl_pp_finish(p(orig_t), orig_t.stride(0), p(cm.mask), p(cm.label), p(cm.ws), p(code_d), len(code), H, W,
                                        p(overlay), overlay.stride(0), p(refined), refined.stride(0), ops._stream()), 'dl_pp_finish')
    scoring = {
        'num_total': counts['num_total'], 'num_pos': counts['num_pos'], 'num_neg': counts['num_neg'],
        'percent_pos': round(counts['num_pos'] / counts['num_total'] * 100, 1) if counts['num_pos'] > 0 else 0,
        'seg_thresh': seg_thresh, 'size_thresh': size_thresh, 'size_thresh_upper': size_thresh_upper,
        'marker_thresh': marker_thresh if marker is not None else None,
    }
    if return_tensors:
        return overlay, refined, scoring
    return overlay.cpu().numpy(), refined.cpu().numpy(), scoring


# ---------------------------------------------------------------------------------------------------------------
# Per-cell export (compute_cell_results, postprocessing.py:1136-1220): host code over the label mask the GPU produced.  Boundary
# tracing is pointer chasing along each cell's outline (a few dozen steps per cell); the "v4" strings are DeepLIIF's wire format
# for cell lists (base-92 numbers + Freeman chain code), so their layout below is a contract, not a choice.
# ---------------------------------------------------------------------------------------------------------------
_RING = ((-1, -1), (0, -1), (1, -1), (1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0))        # (dx, dy), clockwise from the upper-left neighbour
_RING_INDEX = {d: i for i, d in enumerate(_RING)}


def get_cell_boundary(mask, x, y):
    """postprocessing.py:491-581: Moore-neighbour trace of the cell whose first pixel (raster order) is (x, y), clockwise, ending when the
    start pixel is re-entered from the pixel that precedes it on the outline.  mask: 2-D array, 0 = background.
    -> ([(min x, min y), (max x, max y)], [(x, y), ...])"""
    h, w = mask.shape
    if not (0 <= y < h and 0 <= x < w) or mask[y, x] == 0:
        return None, None

    def cell(px, py):
        return 0 <= px < w and 0 <= py < h and mask[py, px] != 0

    # the pixel that precedes the start on a clockwise outline: first cell neighbour met going COUNTER-clockwise from the lower-left one
    k = 6
    while k >= 0 and not cell(x + _RING[k][0], y + _RING[k][1]):
        k -= 1
    if k < 0:
        return [(x, y), (x, y)], [(x, y)]
    first_prev = (x + _RING[k][0], y + _RING[k][1])
    start = (x, y)
    prev, cur = first_prev, start
    outline = [start]
    x0 = x1 = x
    y0 = y1 = y
    while True:
        k = (_RING_INDEX[(prev[0] - cur[0], prev[1] - cur[1])] + 1) % 8          # continue clockwise just past where we came from
        while not cell(cur[0] + _RING[k][0], cur[1] + _RING[k][1]):
            k = (k + 1) % 8
        prev, cur = cur, (cur[0] + _RING[k][0], cur[1] + _RING[k][1])
        if prev == first_prev and cur == start:          # closed: the reference's list ends here too (it drops this closing pair)
            break
        outline.append(cur)
        x0, x1 = min(x0, cur[0]), max(x1, cur[0])
        y0, y1 = min(y0, cur[1]), max(y1, cur[1])
    return [(x0, y0), (x1, y1)], outline


def make_simple_contour(points):
    """postprocessing.py:584-634: drop the interior points of straight runs (the direction signs before and after a point agree); the last
    point is compared against the wrap-around to the first"""
    pts = [(int(p[0]), int(p[1])) for p in points]
    if len(pts) == 1:
        return pts[:]
    sgn = lambda v: (v > 0) - (v < 0)
    keep = [pts[0]]
    n = len(pts)
    for i in range(1, n):
        a, b, c = pts[i - 1], pts[i], pts[(i + 1) % n]
        if (sgn(b[0] - a[0]), sgn(b[1] - a[1])) != (sgn(c[0] - b[0]), sgn(c[1] - b[1])):
            keep.append(b)
    return keep


def to_base92(values, min_len=1):
    """postprocessing.py:685-727: big-endian base-92 digits as characters 35..126; several values share one width (>= min_len, padded with
    chr(35) = the digit 0)"""
    single = not isinstance(values, (list, tuple))
    vals = [values] if single else list(values)
    digits = []
    for v in vals:
        v = int(v)
        d = ''
        while v > 0:
            d = chr(v % 92 + 35) + d
            v //= 92
        digits.append(d)
    width = max(max(len(d) for d in digits), min_len)
    out = [d.rjust(width, chr(35)) for d in digits]
    return out[0] if single else out


def from_base92(val):
    """postprocessing.py:730-749"""
    res = 0
    for ch in val:
        res = res * 92 + (ord(ch) - 35)
    return res


_FREEMAN = {(1, 0): 0, (1, -1): 1, (0, -1): 2, (-1, -1): 3, (-1, 0): 4, (-1, 1): 5, (0, 1): 6, (1, 1): 7}      # (sign dx, sign dy) -> code


def encode_cell_data_v4(data, v6=False):
    """postprocessing.py:752-848.  Layout: [lengths byte] size | classification (2 digits: marker * 2 + positive) | bbox top-left x, y |
    offsets from it of: bbox bottom-right, centroid, first boundary point (6 numbers, one width) | Freeman chain of the remaining boundary
    points, one character per run of <= 10 steps: chr(35 + 8 * steps + direction).  lengths byte = chr(35 + 16 (|size| - 1) + 4 (|top-left| - 1)
    + (|offsets| - 1))."""
    size = to_base92(data['size'])
    marker = data['od'] if v6 else data['marker']
    body = size + to_base92(int(marker) * 2 + int(data['positive']), 2)
    (ax, ay), (bx, by) = data['bbox']
    topleft = to_base92([ax, ay])
    body += topleft[0] + topleft[1]
    cx, cy = data['centroid']
    fx, fy = data['boundary'][0]
    offs = to_base92([bx - ax, by - ay, cx - ax, cy - ay, fx - ax, fy - ay])
    body += ''.join(offs)
    head = chr(35 + (len(size) - 1) * 16 + (len(topleft[0]) - 1) * 4 + (len(offs[0]) - 1))
    chain = []
    pts = data['boundary']
    for j in range(1, len(pts)):
        dx, dy = pts[j][0] - pts[j - 1][0], pts[j][1] - pts[j - 1][1]
        steps = max(abs(dx), abs(dy))
        if steps == 0:
            continue
        code = _FREEMAN[((dx > 0) - (dx < 0), (dy > 0) - (dy < 0))]
        while steps > 10:
            chain.append(chr(35 + 80 + code))
            steps -= 10
        chain.append(chr(35 + steps * 8 + code))
    return head + body + ''.join(chain)


def cell_results_from_mapping(mask, cells, defaults, version, seg_thresh, noise_thresh, large_noise_thresh):
    """The host half of compute_cell_results (:1170-1220): per-cell boundary, bbox and (for versions 4 / 6) the encoded string.
    mask: 2-D uint8 ndarray after the cell mapping (0 = background), cells / defaults as get_cells_info returns them."""
    od = version in (5, 6)
    out = []
    for c in cells:
        bbox, boundary = get_cell_boundary(mask, c[3], c[4])
        data = {'size': c[0], 'positive': c[1], ('od' if od else 'marker'): c[2], 'bbox': bbox, 'centroid': (c[5], c[6]),
                'boundary': make_simple_contour(boundary)}
        out.append(encode_cell_data_v4(data, v6=(version == 6)) if version in (4, 6) else data)
    settings = {'default_size_thresh': defaults['size_thresh'], 'noise_thresh': noise_thresh, 'large_noise_thresh': large_noise_thresh,
                'seg_thresh': seg_thresh}
    if not od:
        settings['default_marker_thresh'] = defaults['marker_thresh'] if 'marker_thresh' in defaults else None
    return {'cells': out, 'settings': settings, 'dataVersion': version}


def compute_cell_results(seg, marker, resolution, version=3, seg_thresh=DEFAULT_SEG_THRESH, noise_thresh=DEFAULT_NOISE_THRESH,
                         large_noise_thresh=None, device=None):
    """deepliif/postprocessing.py:1136-1220: individual cell data.  Versions 3 / 4 take the inferred marker image, 5 / 6 the ORIGINAL image
    (optical density); 4 / 6 return every cell as one encoded ASCII string."""
    import warnings
    if version not in (3, 4, 5, 6):
        warnings.warn('Invalid cell data version provided, defaulting to version 3.')
        version = 3
    large = calculate_large_noise_thresh(large_noise_thresh, resolution)
    cm = get_cells_info(seg, marker, resolution, noise_thresh, seg_thresh, large, use_od=version in (5, 6), device=device)
    return cell_results_from_mapping(cm.mask.cpu().numpy(), cm.cells, cm.defaults, version, seg_thresh, noise_thresh, large)
