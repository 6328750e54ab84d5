"""Time dl_pack_weights_batch over the conv weights of a UNet-512 + Resnet-9 + PatchGAN set, tiled form (csrc/pack_tile.h) against the
chunk-per-thread form (DL_PACK_TILED=0), and check that both write the same bits at these sizes.  GPU only."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepliif_amd import _lib as L, ops                                     # noqa: E402
from deepliif_amd.geometry import ConvSpec                                   # noqa: E402

DEV = torch.device('cuda:0')
UNET = [ConvSpec('conv', 3, 64, 4, 2, 1), ConvSpec('conv', 64, 128, 4, 2, 1), ConvSpec('conv', 128, 256, 4, 2, 1), ConvSpec('conv', 256, 512, 4, 2, 1)] + \
       [ConvSpec('conv', 512, 512, 4, 2, 1)] * 5 + [ConvSpec('convT', 512, 512, 4, 2, 1)] + [ConvSpec('convT', 1024, 512, 4, 2, 1)] * 4 + \
       [ConvSpec('convT', 1024, 256, 4, 2, 1), ConvSpec('convT', 512, 128, 4, 2, 1), ConvSpec('convT', 256, 64, 4, 2, 1), ConvSpec('convT', 128, 3, 4, 2, 1)]
RESNET = [ConvSpec('conv', 3, 64, 7, 1, 3), ConvSpec('conv', 64, 128, 3, 2, 1), ConvSpec('conv', 128, 256, 3, 2, 1)] + [ConvSpec('conv', 256, 256, 3, 1, 1)] * 18 + \
         [ConvSpec('convT', 256, 128, 3, 2, 1, L.PAD_ZERO, 1), ConvSpec('convT', 128, 64, 3, 2, 1, L.PAD_ZERO, 1), ConvSpec('conv', 64, 3, 7, 1, 3)]
PATCH = [ConvSpec('conv', 6, 64, 4, 2, 1), ConvSpec('conv', 64, 128, 4, 2, 1), ConvSpec('conv', 128, 256, 4, 2, 1), ConvSpec('conv', 256, 512, 4, 1, 1),
         ConvSpec('conv', 512, 1, 4, 1, 1)]


def build(specs, with_lo):
    jobs, params = [], 0
    g = torch.Generator(device='cpu').manual_seed(7)
    for spec in specs:
        shape = (spec.cout, spec.cin, spec.k, spec.k) if spec.kind == 'conv' else (spec.cin, spec.cout, spec.k, spec.k)
        w = (torch.randn(shape, generator=g) * 0.05).to(DEV)
        params += w.numel()
        for plan in (spec.forward_plan(), spec.dgrad_plan()):
            jobs.append((ops.PackedWeights(plan, DEV, with_lo), w))
    return jobs, params


def timed(be, table, n, reps=10):
    be.pack_batch_run(table, n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        be.pack_batch_run(table, n)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    be = ops.impl()
    out = {}
    for name, specs, with_lo in (('unet512_x2+resnet9_x2+patchgan_x4 bf16', UNET * 2 + RESNET * 2 + PATCH * 4, False),
                                 ('unet512+resnet9+patchgan strict (hi + lo)', UNET + RESNET + PATCH, True)):
        jobs, params = build(specs, with_lo)
        img_bytes = sum(pk.hi.numel() * 2 * (2 if with_lo else 1) for pk, _ in jobs)
        tiled = be.pack_batch_build(jobs)
        us_t = timed(be, tiled, len(jobs))
        got = [(pk.hi.clone(), None if pk.lo is None else pk.lo.clone()) for pk, _ in jobs]
        os.environ['DL_PACK_TILED'] = '0'
        chunk = be.pack_batch_build(jobs)
        del os.environ['DL_PACK_TILED']
        for pk, _ in jobs:
            pk.hi.zero_()
        us_c = timed(be, chunk, len(jobs))
        same = all(torch.equal(pk.hi.view(torch.int16), h.view(torch.int16)) and (lo is None or torch.equal(pk.lo.view(torch.int16), lo.view(torch.int16)))
                   for (pk, _), (h, lo) in zip(jobs, got))
        traffic = params * 4 * 2 + img_bytes                       # each master weight is read once per image (forward + data-gradient)
        out[name] = {'params_M': round(params / 1e6, 1), 'images': len(jobs), 'blocks_tiled': tiled[2], 'blocks_chunk': chunk[2], 'us_tiled': round(us_t, 1),
                     'us_chunk': round(us_c, 1), 'TBps_tiled': round(traffic / us_t / 1e6, 2), 'TBps_chunk': round(traffic / us_c / 1e6, 2), 'bit_identical': bool(same)}
        print(name, out[name], flush=True)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/pack_time_r05.json', 'w'), indent=1)
    return 0 if all(v['bit_identical'] for v in out.values()) else 1


if __name__ == '__main__':
    sys.exit(main())
