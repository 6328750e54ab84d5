"""Helpers shared by make_golden.py (fixture writer) and the tests (fixture readers)."""
import numpy as np
import torch


def seeded_uniform(shape, seed):
    """U(-1, 1) tensor from an explicit CPU generator: the synthetic-tile distribution of SURVEY 8(d)
    (matches the [-1, 1] range of deepliif/data/__init__.py:133-138 transform)."""
    return torch.rand(tuple(shape), generator=torch.Generator().manual_seed(int(seed))) * 2 - 1


def digest(t, nproj=4, seed=4242):
    """[numel, l2 norm, sum, <t, r_0>, ..., <t, r_3>] with r_i seeded N(0,1) vectors -- a compact stand-in for a
    large tensor: any elementwise discrepancy shows up in the random projections."""
    flat = torch.as_tensor(t).detach().reshape(-1).double().cpu()
    g = torch.Generator().manual_seed(seed)
    vals = [float(flat.numel()), float(flat.norm()), float(flat.sum())]
    for _ in range(nproj):
        r = torch.randn(flat.numel(), generator=g, dtype=torch.float64)
        vals.append(float((flat * r).sum()))
    return np.array(vals, dtype=np.float64)


def digest_close(actual, expected, rtol):
    """Compare a tensor (or its digest) with a stored digest: the projections of N(0,1) vectors have standard
    deviation = l2 norm, so differences are measured relative to the norm."""
    a = actual if isinstance(actual, np.ndarray) and actual.ndim == 1 and actual.shape == expected.shape else digest(actual)
    if a[0] != expected[0]:
        return False, f'numel {a[0]} != {expected[0]}'
    scale = max(expected[1], 1e-30)
    err = np.abs(a[1:] - expected[1:]).max() / scale
    # sum can be large relative to the norm: normalise by sqrt(numel) * norm as an upper bound
    err_sum = abs(a[2] - expected[2]) / (scale * np.sqrt(expected[0]))
    err_other = np.abs(np.delete(a, [0, 2]) - np.delete(expected, [0, 2])).max() / scale
    worst = max(err_sum, err_other)
    return worst <= rtol, f'digest rel err {worst:.3e} (tol {rtol:.1e})'


# ---- shared by make_golden_tiler.py and the tiler tests
def synth_image(w, h, seed):
    """low-entropy synthetic RGB image (gradients + blobs + a little noise) so that the npz stays small"""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy // 7) % 256], axis=-1).astype(np.int32)
    for _ in range(6):
        cx, cy, r = rng.randint(0, w), rng.randint(0, h), rng.randint(5, max(6, min(w, h) // 3))
        m = (xx - cx) ** 2 + (yy - cy) ** 2 < r * r
        img[m] = rng.randint(0, 256, 3)
    img += rng.randint(-2, 3, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def tiler_result_tiles(tile_np):
    """two deterministic 'network outputs' per tile (any per-pixel function of the tile works: it must survive crop + paste)"""
    return {'A': 255 - tile_np, 'B': np.ascontiguousarray(tile_np[..., ::-1] // 2 + 7)}


def synth_cells(h: int, w: int, n_cells: int, seed: int):
    """Seeded synthetic inputs of the post-processing row: (orig, seg, marker) uint8 HxWx3.  seg follows the reference's convention
    (postprocessing.py:164-190): R = positive probability, B = negative probability, G = boundary (<= 80 inside cells); blobs of
    random radius, some touching each other or the image border, some with an enclosed hole (pixels below the threshold inside a
    cell), speckle noise below / around the noise threshold; marker = smooth blobs over positive cells; orig = stained-looking image."""
    import numpy as np
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    seg = np.zeros((h, w, 3), dtype=np.float64)
    seg[..., 1] = rng.randint(0, 60, size=(h, w))
    marker = rng.randint(0, 25, size=(h, w)).astype(np.float64)
    orig = np.full((h, w, 3), 225.0) + rng.randint(-12, 12, size=(h, w, 3))
    for i in range(n_cells):
        cy, cx = rng.randint(-2, h + 2), rng.randint(-2, w + 2)
        r = rng.uniform(2.0, 9.0)
        ell = rng.uniform(0.6, 1.0)
        d = np.sqrt(((yy - cy) / ell) ** 2 + (xx - cx) ** 2)
        inside = d <= r
        pos = rng.rand() < 0.45
        strong = rng.randint(150, 255)
        weak = rng.randint(0, 90)
        seg[..., 0][inside] = strong if pos else weak
        seg[..., 2][inside] = weak if pos else strong
        seg[..., 1][inside] = rng.randint(0, 70)
        if rng.rand() < 0.3 and r > 4.5:                       # enclosed hole: stays "unknown" inside the cell
            hole = d <= r * 0.3
            seg[..., 0][hole] = 10
            seg[..., 2][hole] = 10
        if rng.rand() < 0.25:                                  # a ring of boundary-coloured pixels (G > 80) cutting the blob
            ring = np.abs(d - r * 0.6) < 0.7
            seg[..., 1][ring] = 200
        if pos:
            marker[inside] = np.maximum(marker[inside], rng.randint(60, 255) * np.exp(-(d[inside] / (r + 1)) ** 2))
        orig[inside] = orig[inside] * 0.55 + np.array([120.0, 70.0, 40.0] if pos else [60.0, 80.0, 150.0]) * 0.45
    speck = rng.rand(h, w) < 0.01                             # isolated bright pixels: cells below the noise threshold
    seg[..., 0][speck] = 220
    seg[..., 1][speck] = 0
    mk = np.repeat(marker[..., None], 3, axis=2)
    mk[..., 1] *= 0.8
    mk[..., 2] *= 0.5
    return (np.clip(orig, 1, 255).astype(np.uint8), np.clip(seg, 0, 255).astype(np.uint8), np.clip(mk, 0, 255).astype(np.uint8))
