"""CPU restatement of the reference's segmentation post-processing (deepliif/postprocessing.py) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(deepliif_amd/postprocessing.py -> csrc/postproc.hip) never does.  Pinned by tests/golden/post_cases.npz, which holds what the
reference itself returned for seeded synthetic images (tests/golden/make_golden_post.py), stage by stage, and the reference's own shipped
known answers (Datasets/Sample_Dataset/val/metrics.json: cell counts of its two validation images).

The reference is sequential in-place code (numba loops, explicit flood-fill stacks).  This restatement is written declaratively --
connected components (scipy.ndimage.label), per-component reductions, neighbourhood tests -- with the order-dependent corners of the
reference spelled out where they exist:
  * mark_background (:193-232) grows BACKGROUND from the image border through 4-connected UNKNOWN pixels to a fixed point: the result
    is the union of the 4-connected UNKNOWN components that touch the border; enclosed UNKNOWN holes stay and become cell pixels.
  * compute_cell_mapping (:235-308): 8-connected components of everything that is not BACKGROUND, listed in raster order of their first
    pixel; that first pixel is the one create_cell_classification (:923-1000) later paints with the BORDER label instead of the class.
  * border pixels: a BACKGROUND pixel 4-adjacent to a NON-seed pixel of an accepted cell; claimed by the first such cell in list order.
  * enlarge_cell_boundaries (:1003-1030): a BACKGROUND pixel takes the class of the first border pixel among its 8 neighbours in
    raster order (the pass marks with temporary labels, so growth is exactly one pixel per pass).
All arithmetic is integer (int64, as numba widens it) except the default-threshold statistics, which repeat the reference's
float64 expressions in the same order."""
import math

import numpy as np
from scipy import ndimage

LABEL_UNKNOWN, LABEL_POSITIVE, LABEL_NEGATIVE, LABEL_BACKGROUND, LABEL_CELL = 50, 200, 150, 0, 100      # postprocessing.py:87-91
LABEL_BORDER_POS, LABEL_BORDER_NEG = 220, 170                                                             # :92-93
DEFAULT_SEG_THRESH, DEFAULT_NOISE_THRESH = 120, 4                                                         # :83-84

FOUR = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=bool)
EIGHT = np.ones((3, 3), dtype=bool)


def gray_max(img):
    """to_array(img, grayscale=True) (:98-120): maximum channel"""
    img = np.asarray(img)
    return img.max(axis=-1) if img.ndim == 3 else img


def od_image(orig):
    """create_od_image (:123-138): round(100 * (log10(255/r) + log10(255/g) + log10(255/b))), value 0 treated as 1"""
    lut = [0.0] + [math.log10(255 / i) for i in range(1, 256)]
    lut[0] = lut[1]
    lut = np.array(lut, dtype=np.float64)
    o = np.asarray(orig)
    val = (lut[o[..., 0]] + lut[o[..., 1]]) + lut[o[..., 2]]
    return np.rint(val * 100).astype(np.int64)                 # round(): half to even, like np.rint


def posneg_mask(seg, thresh):
    """create_posneg_mask (:163-190)"""
    s = np.asarray(seg).astype(np.int64)
    cand = (s[..., 0] + s[..., 2] > thresh) & (s[..., 1] <= 80)
    mask = np.full(s.shape[:2], LABEL_UNKNOWN, dtype=np.uint8)
    mask[cand & (s[..., 0] >= s[..., 2])] = LABEL_POSITIVE
    mask[cand & (s[..., 0] < s[..., 2])] = LABEL_NEGATIVE
    return mask


def mark_background(mask):
    """mark_background (:193-232), in place"""
    lab, _ = ndimage.label(mask == LABEL_UNKNOWN, structure=FOUR)
    touching = np.unique(np.concatenate([lab[0], lab[-1], lab[:, 0], lab[:, -1]]))
    touching = touching[touching != 0]
    mask[np.isin(lab, touching)] = LABEL_BACKGROUND


def cell_mapping(mask, marker, noise_thresh, large_noise_thresh, use_avg=False):
    """compute_cell_mapping (:235-308): mask in place (every non-background pixel -> CELL); returns (cells, component image, list index per
    component) -- cells = [(count, positive, marker value, first x, first y, centre x, centre y)] in raster order of the first pixel"""
    lab, n = ndimage.label(mask != LABEL_BACKGROUND, structure=EIGHT)
    h, w = mask.shape
    cells, index_of = [], {}
    if n:
        flat = lab.ravel()
        order = np.argsort(flat, kind='stable')
        bounds = np.searchsorted(flat[order], np.arange(1, n + 2))
        comps = []
        for k in range(n):
            idx = order[bounds[k]:bounds[k + 1]]                     # linear indices of component k+1, increasing
            comps.append((int(idx[0]), k + 1, idx))
        comps.sort()                                                  # raster order of the first pixel
        mflat = mask.ravel()
        mk = None if marker is None else np.asarray(marker).astype(np.int64).ravel()
        for first, lbl, idx in comps:
            count = int(idx.size)
            ys, xs = idx // w, idx % w
            if not (count > noise_thresh and (large_noise_thresh is None or count < large_noise_thresh)):
                continue
            npos, nneg = int((mflat[idx] == LABEL_POSITIVE).sum()), int((mflat[idx] == LABEL_NEGATIVE).sum())
            if mk is None:
                mval = 0
            elif use_avg:
                mval = int(round(int(mk[idx].sum()) / count))
            else:
                mval = int(mk[idx].max())
            cy, cx = int(round(int(ys.sum()) / count)), int(round(int(xs.sum()) / count))
            index_of[lbl] = len(cells)
            cells.append((count, npos >= nneg, mval, int(first % w), int(first // w), cx, cy))
    mask[mask != LABEL_BACKGROUND] = LABEL_CELL
    return cells, lab, index_of


def default_size_threshold(sizes, resolution='40x'):
    """calculate_default_size_threshold (:406-447) with create_kde (:365-403): first local minimum of a 500-bin Gaussian KDE of sqrt(size)"""
    sizes = np.asarray(sizes, dtype=np.int64)
    if sizes.shape[0] <= 1:
        return 0
    values = np.sqrt(sizes)
    count = 500
    inv = 1 / math.sqrt(2 * math.pi)
    step = (float(values.max()) + 1) / count
    nvals = values.shape[0]
    kde = np.zeros(count, dtype=np.float32)
    for i in range(count):
        x = i * step
        total = 0
        for j in range(nvals):
            val = (x - values[j]) * 1.0
            total += math.exp(-(val * val / 2)) * inv
        kde[i] = total / (nvals * 1.0)
    idx = 1
    for i in range(1, count - 1):
        if kde[i] < kde[i - 1] and kde[i] < kde[i + 1]:
            idx = i
            break
    t = (idx - 1) * step
    lo, dflt, hi = {'20x': (3, 4, 6), '10x': (2, 2, 3)}.get(resolution, (4, 7, 10))
    if t < lo:
        t = lo
    elif t > hi:
        t = dflt
    return round(t * t)


def default_marker_threshold(marker_gray):
    """calculate_default_marker_threshold (:472-488): 90 % of the 0.1 .. 99.9 percentile range of the non-zero pixels"""
    nz = marker_gray[marker_gray != 0]
    if nz.shape[0] == 0:
        lo = hi = 0
    else:
        lo, hi = round(np.percentile(nz, 0.1)), round(np.percentile(nz, 99.9))
    return round((hi - lo) * 0.9) + lo


def large_noise_threshold(value, resolution):
    """calculate_large_noise_thresh (:1125-1133)"""
    if value != 'default':
        return value
    return {'10x': 1000, '20x': 4000}.get(resolution, 16000)


def cells_info(seg, marker, resolution, noise_thresh, seg_thresh, large_noise_thresh, use_od=False):
    """get_cells_info (:311-362) -> (mask, cells, defaults, component image, list index per component)"""
    mk = None
    if marker is not None:
        mk = od_image(marker) if use_od else gray_max(marker)
    mask = posneg_mask(seg, seg_thresh)
    mark_background(mask)
    cells, lab, index_of = cell_mapping(mask, mk, noise_thresh, large_noise_thresh, use_od)
    defaults = {'size_thresh': default_size_threshold([c[0] for c in cells], resolution)}
    if marker is not None and not use_od:
        defaults['marker_thresh'] = default_marker_threshold(np.asarray(mk))
    return mask, cells, defaults, lab, index_of


def _shift(a, dy, dx, fill=False):
    out = np.full_like(a, fill)
    h, w = a.shape
    ys, yd = (slice(0, h - dy), slice(dy, h)) if dy >= 0 else (slice(-dy, h), slice(0, h + dy))
    xs, xd = (slice(0, w - dx), slice(dx, w)) if dx >= 0 else (slice(-dx, w), slice(0, w + dx))
    out[yd, xd] = a[ys, xs]
    return out


def classify(mask, cells, lab, index_of, size_thresh=0, marker_thresh=None, size_thresh_upper=None, od_lower=None, od_upper=None):
    """create_cell_classification (:923-1000), mask in place; returns the counts"""
    num_pos = num_neg = 0
    lbl_of = {v: k for k, v in index_of.items()}
    for ci, cell in enumerate(cells):
        if not (cell[0] > size_thresh and (size_thresh_upper is None or cell[0] < size_thresh_upper)):
            continue
        pos = cell[1]
        if marker_thresh is not None and cell[2] > marker_thresh:
            pos = True
        if od_lower is not None and cell[2] < od_lower:
            pos = False
        elif od_upper is not None and cell[2] > od_upper:
            pos = False
        label, border = (LABEL_POSITIVE, LABEL_BORDER_POS) if pos else (LABEL_NEGATIVE, LABEL_BORDER_NEG)
        num_pos, num_neg = num_pos + (1 if pos else 0), num_neg + (0 if pos else 1)
        body = lab == lbl_of[ci]
        body[cell[4], cell[3]] = False                        # the first pixel is painted with the border label and never expands the border
        touch = np.zeros_like(body)
        for dy, dx in ((0, -1), (-1, 0), (1, 0), (0, 1)):
            touch |= _shift(body, dy, dx)
        mask[touch & (mask == LABEL_BACKGROUND)] = border
        mask[body] = label
        mask[cell[4], cell[3]] = border
    return {'num_total': num_pos + num_neg, 'num_pos': num_pos, 'num_neg': num_neg}


def enlarge_boundaries(mask):
    """enlarge_cell_boundaries (:1003-1030), in place: one pixel of growth into BACKGROUND"""
    is_b = (mask == LABEL_BORDER_POS) | (mask == LABEL_BORDER_NEG)
    new = np.zeros(mask.shape, dtype=np.uint8)
    free = mask == LABEL_BACKGROUND
    # a border pixel at p marks its 8 neighbours; the reference visits p in raster order, so q is claimed by its first border neighbour
    # in raster order: offsets of p relative to q, ascending
    for dy, dx in ((-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)):
        src_b = _shift(is_b, -dy, -dx)                        # src_b[q] = is_b[q + (dy, dx)]
        src_v = _shift(mask, -dy, -dx, 0)
        take = free & src_b & (new == 0)
        new[take] = src_v[take]
    mask[new != 0] = new[new != 0]


def final_images(orig, mask):
    """create_final_images (:1033-1071) -> (overlay, refined)"""
    overlay = np.array(orig, dtype=np.uint8, copy=True)
    refined = np.zeros_like(overlay)
    overlay[mask == LABEL_BORDER_POS] = (255, 0, 0)
    overlay[mask == LABEL_BORDER_NEG] = (0, 0, 255)
    refined[..., 1][(mask == LABEL_BORDER_POS) | (mask == LABEL_BORDER_NEG)] = 255
    refined[..., 0][mask == LABEL_POSITIVE] = 255
    refined[..., 2][mask == LABEL_NEGATIVE] = 255
    return overlay, refined


def compute_final_results(orig, seg, marker, resolution, size_thresh='default', marker_thresh=None, size_thresh_upper=None,
                          seg_thresh=DEFAULT_SEG_THRESH, noise_thresh=DEFAULT_NOISE_THRESH, large_noise_thresh=None,
                          od_thresh_lower=None, od_thresh_upper=None):
    """compute_final_results (:1223-1304) -> (overlay, refined, scoring)"""
    large = large_noise_threshold(large_noise_thresh, resolution)
    use_od = od_thresh_lower is not None or od_thresh_upper is not None
    mask, cells, defaults, lab, index_of = cells_info(seg, orig if use_od else marker, resolution, noise_thresh, seg_thresh, large, use_od)
    if size_thresh is None:
        size_thresh = 0
    elif size_thresh == 'default':
        size_thresh = defaults['size_thresh']
    if marker_thresh == 'default':
        marker_thresh = defaults['marker_thresh']
    counts = classify(mask, cells, lab, index_of, size_thresh, marker_thresh, size_thresh_upper, od_thresh_lower, od_thresh_upper)
    enlarge_boundaries(mask)
    enlarge_boundaries(mask)
    overlay, refined = final_images(orig, mask)
    scoring = {'num_total': counts['num_total'], 'num_pos': counts['num_pos'], 'num_neg': counts['num_neg'],
               'percent_pos': round(counts['num_pos'] / counts['num_total'] * 100, 1) if counts['num_pos'] > 0 else 0,
               'seg_thresh': seg_thresh, 'size_thresh': size_thresh, 'size_thresh_upper': size_thresh_upper,
               'marker_thresh': marker_thresh if marker is not None else None}
    return overlay, refined, scoring
