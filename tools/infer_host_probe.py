"""Is the 9-generator inference DAG (BASELINE configs[1]) host- or GPU-bound?  host issue time vs completed time per batch of 8 tiles, and the
same batch replayed from a captured hipGraph (torch.cuda.graph over the ctypes launches on the capture stream)."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepliif_amd import inference as I, engine as E
dev = torch.device('cuda', 0)
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
iopt = types.SimpleNamespace(model='DeepLIIF', modalities_no=4, seg_gen=True, mod_id_seg='S', input_id=0, input_nc=3, output_nc=3, ngf=64, norm='batch',
                             padding='zero', net_g='resnet_9blocks', net_gs='unet_512', input_no=1, scale_size=512,
                             modalities_names=['IHC', 'Hema', 'DAPI', 'Lap2', 'Marker'], gpu_ids=[0])
torch.manual_seed(0)
nets = I.build_generators(iopt, dev, prec)
E.settle_gc()
sw = [0.25, 0.15, 0.25, 0.1, 0.25]
x = E.to_engine((torch.rand(8, 3, 512, 512) * 2 - 1).to(dev), E.Precision.get(prec))
run = lambda: I.run_generators_engine(x, nets, iopt, seg_weights=sw)
for _ in range(3):
    out = run()
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    t0 = time.perf_counter(); out = run(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
print(f'{prec} eager: host issue {sorted(host)[5]:.2f} ms, completed {sorted(total)[5]:.2f} ms per batch of 8 ({8 / sorted(total)[5] * 1e3:.1f} tiles/s)')
ref = {k: v.t.clone() for k, v in out.items()}
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            run()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        gout = run()
    torch.cuda.synchronize()
    rep = []
    for _ in range(10):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); rep.append((time.perf_counter() - t0) * 1e3)
    same = all(torch.equal(gout[k].t, ref[k]) for k in ref)
    print(f'{prec} graph replay: {sorted(rep)[5]:.2f} ms per batch of 8 ({8 / sorted(rep)[5] * 1e3:.1f} tiles/s), outputs bit-identical to eager: {same}')
except Exception as e:
    print('graph capture failed:', repr(e)[:300])
