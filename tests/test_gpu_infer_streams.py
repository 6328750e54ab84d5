"""DL_INFER_STREAMS=N (opt-in): the independent chains G_i(tile) -> GS_i(G_i(tile)) of the DeepLIIF inference DAG (deepliif/models/__init__.py:293-361) on N HIP
streams of the calling thread, the weighted segmentation sum behind a join.  Same kernels on the same operands, so every output of every key must be
BIT-identical to the one-stream DAG, in the same key order -- for the full DAG, seg_only and both policies."""
import types

import pytest
import torch

from deepliif_amd import inference as I
from deepliif_amd import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _nets(precision, ngf=8):
    opt = types.SimpleNamespace(model='DeepLIIF', modalities_no=4, seg_gen=True, mod_id_seg='S', input_id=0, input_nc=3, output_nc=3, ngf=ngf, norm='batch', padding='zero',
                                net_g='resnet_9blocks', net_gs='unet_256', input_no=1, scale_size=256, modalities_names=['IHC', 'Hema', 'DAPI', 'Lap2', 'Marker'], gpu_ids=[0])
    torch.manual_seed(5)
    return opt, I.build_generators(opt, torch.device(DEV), precision)


@pytest.mark.parametrize('precision', ['bf16', 'fp32'])
@pytest.mark.parametrize('nstreams', [2, 3, 5])
def test_inference_dag_on_streams_is_bit_identical(nstreams, precision, monkeypatch):
    ops._impl = None
    ops.WS._thread_state().pop(('infer_streams', 0), None)       # (a process started with DL_INFER_STREAMS set has its own list cached)
    opt, nets = _nets(precision)
    x = (torch.rand(4, 3, 256, 256, generator=torch.Generator().manual_seed(9)) * 2 - 1).to(DEV)
    sw = [0.25, 0.15, 0.25, 0.1, 0.25]
    for seg_only in (False, True):
        monkeypatch.setattr(I, '_INFER_STREAMS', 1)
        ref = I.run_generators(x, nets, opt, seg_only=seg_only, seg_weights=sw)
        torch.cuda.synchronize()
        monkeypatch.setattr(I, '_INFER_STREAMS', nstreams)
        for rep in range(2):                          # the second call reuses the streams' scratch states
            got = I.run_generators(x, nets, opt, seg_only=seg_only, seg_weights=sw)
            torch.cuda.synchronize()
            assert list(got.keys()) == list(ref.keys())
            for k in ref:
                assert torch.equal(got[k], ref[k]), (k, seg_only, rep)
    st = ops.WS._thread_state()
    assert len(st[('infer_streams', 0)]) == nstreams
    st.pop(('infer_streams', 0)), ops.WS.forget_branch_streams()
