"""Import harness for the *reference* DeepLIIF package (this container only).

Used ONLY by tests/golden/make_golden.py to produce committed fixtures and to
cross-check the oracle.  /root/reference does not exist on the GPU box, so
nothing that runs there may import this module.

The reference imports several third-party packages at import time that are not
installed here; they are unrelated to the hot path and
are stubbed in sys.modules before `import deepliif.models`.
"""
import sys
import types
import importlib.machinery

REF_ROOT = '/root/reference'


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    def _fallback(attr):   # PEP 562 fallback
        if attr.startswith('__'):
            raise AttributeError(attr)
        return _Anything()
    m.__getattr__ = _fallback
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    import torch  # noqa: F401  (must be imported before any stub exists)
    def _identity_decorator(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    for name in ['torchvision', 'torchvision.models', 'torchvision.transforms', 'cv2',
                 'skimage', 'skimage.filters', 'skimage.metrics', 'skimage.color', 'skimage.morphology',
                 'skimage.measure', 'skimage.segmentation', 'skimage.feature', 'skimage.io',
                 'bioformats', 'bioformats.omexml', 'javabridge', 'tifffile', 'zarr',
                 'dominate', 'dominate.tags', 'visdom', 'scipy.ndimage', 'openslide']:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _stub(name)
    if 'numba' not in sys.modules:
        try:
            import numba  # noqa
        except Exception:
            nb = _stub('numba', jit=_identity_decorator, njit=_identity_decorator, prange=range)
            _stub('numba.typed', List=list)
            nb.typed = sys.modules['numba.typed']
    if 'dask' not in sys.modules:
        try:
            import dask  # noqa
        except Exception:
            _stub('dask', delayed=_identity_decorator, compute=lambda *a, **k: tuple(a))
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


def import_reference():
    """Returns (deepliif.models, deepliif.models.networks) from /root/reference with VGG loss zeroed."""
    install_stubs()
    import torch
    import deepliif.models as models
    from deepliif.models import networks

    class _ZeroVGG(torch.nn.Module):
        # the pretrained VGG19 needs a download; the north-star path is GAN + SmoothL1 only
        def forward(self, x, y):
            return x.new_zeros(())
    networks.VGGLoss = _ZeroVGG
    return models, networks
