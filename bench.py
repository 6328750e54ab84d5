#!/usr/bin/env python
"""bench.py -- DeepLIIF training-step throughput on MI355X (BASELINE.json metric: 512x512 tiles/s, train step 5G+5D).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = model.set_input(batch) + model.optimize_parameters() of the DeepLIIF model
(modalities_no=5, seg_gen=False -> 5 x ResnetGenerator-9block + 5 x NLayerDiscriminator(n=4), GAN + SmoothL1, Adam) on a
batch of 8 synthetic 512x512x3 tiles per GPU that is already resident in HBM (BASELINE.json configs[2] per GPU; weak
scaling).  `--workload infer` times the 9-generator inference DAG of configs[1] instead (4 Resnet-9 + 5 UNet-512, batch 8).
Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     : dominant layer shape = the 3x3, 256->256 ch conv at 8x128x128 pixels (the 18 ResnetBlock convs of every Resnet-9
                 and, when training, their data-gradients); the kernel NAME is whatever the library dispatches for that
                 descriptor (dl_conv_kernel_name), per-launch time from events recorded on the launch stream around the host call
                 inside the timed region; `traffic` only when the committed PMC summary was collected on that same kernel
  cpu_baseline : the CPU oracle (oracle/deepliif_oracle.py, a port of the reference's PyTorch step) timed on this box's host
                 cores for a bounded sample of whole steps at batch 1 (10-30 s of CPU work; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GF_PER_TILE_TRAIN_5R5D = 6817.0      # BASELINE.md / SURVEY 8(d): conv MACs only, FLOP = 2*MAC
GF_PER_TILE_INFER = 1828.0
GF_PER_TILE_TRAIN_18NETS = 7051.0     # real DeepLIIF (4 Resnet-9 + 5 UNet-512 generators, 9 NLayerD) step
PEAK_BF16_TFLOPS = 2500.0            # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def make_opt(args, device_index, M=5, seg_gen=False):
    w = [1.0 / (M + 1)] * (M + 1)      # cli.py:349-371 defaults for modalities_no != 4
    lw = w
    if M == 4:                         # the real DeepLIIF defaults (cli.py:349-371)
        w, lw = [0.25, 0.15, 0.25, 0.1, 0.25], [0.2] * 5
    return types.SimpleNamespace(
        model='DeepLIIF', name='bench', checkpoints_dir='/tmp/dl_amd_bench', gpu_ids=[device_index], is_train=True, phase='train',
        continue_train=False, modalities_no=M, seg_gen=seg_gen, modalities_names=[], input_nc=3, input_no=1, output_nc=3, ngf=64, ndf=64,
        net_g='resnet_9blocks', net_gs='unet_512', net_d='n_layers', n_layers_D=4, norm=args.norm, no_dropout=True, init_type='normal',
        init_gain=0.02, padding='zero', upsample='convtranspose', gan_mode='vanilla', gan_mode_s='lsgan', optimizer='adam', lr_g=2e-4,
        lr_d=2e-4, beta1=0.5, lr_policy='linear', n_epochs=100, n_epochs_decay=100, epoch_count=0, seg_weights=w, loss_G_weights=lw,
        loss_D_weights=lw, lambda_L1=100.0, verbose=False, epoch='latest', load_iter=0, precision=args.precision)


class KernelTimer:
    """Records HIP events around every launch of the dominant conv kernel (on the stream the kernel is launched on)."""

    def __init__(self, backend, shape):
        self.backend, self.shape, self.pairs, self.enabled = backend, tuple(shape), [], False
        self.kernel = '?'
        self._orig = backend.conv_forward
        backend.conv_forward = self._wrapped

    def _wrapped(self, packed, x, out, *args, **kwargs):
        hit = self.enabled and tuple(x.shape) == self.shape and tuple(out.shape) == self.shape and packed.plan.n_phase == 1
        if hit:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
        ret = self._orig(packed, x, out, *args, **kwargs)
        if hit:
            e.record()
            self.pairs.append((s, e))
            self.kernel = getattr(self.backend, 'last_conv_kernel', '') or self.kernel      # what the library dispatched
        return ret

    def mean_seconds(self):
        if not self.pairs:
            return None
        return sum(s.elapsed_time(e) for s, e in self.pairs) / len(self.pairs) * 1e-3


def usable_cores():
    """Cores this process may really use: scheduler affinity capped by the cgroup CPU quota (containers)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline_child(norm, size):
    """(child process) optimize_parameters() steps of the CPU oracle at batch 1, same model family, random init."""
    from oracle import deepliif_oracle as O
    cores = min(usable_cores(), 64)
    torch.set_num_threads(cores)
    cfg = O.OracleConfig(modalities_no=5, seg_gen=False, norm=norm, padding='zero', ngf=64, ndf=64)
    g = torch.Generator().manual_seed(0)
    nets = {}
    for i in range(1, 6):
        nets[f'G{i}'] = O.random_state_dict('resnet_9blocks', 3, 3, 64, norm, 'zero', generator=g)
        nets[f'D{i}'] = O.random_state_dict('n_layers', 6, 3, 64, norm, 'zero', 4, generator=g)
    om = O.OracleDeepLIIF(cfg, nets)
    A = torch.rand(1, 3, size, size, generator=g) * 2 - 1
    B = [torch.rand(1, 3, size, size, generator=g) * 2 - 1 for _ in range(5)]
    om.set_input({'A': A, 'B': B})
    # bounded sample of ~10-30 s of CPU work: whole steps until at least 10 s have been spent (2 steps on a 16-core host),
    # at most 4; the per-step mean is reported
    times = []
    while len(times) < 4 and (sum(times) < 10.0 or not times):
        t0 = time.time()
        om.optimize_parameters()
        times.append(time.time() - t0)
    print(json.dumps({'seconds': sum(times) / len(times), 'steps': len(times), 'total_seconds': sum(times), 'cores': cores, 'size': size}), flush=True)


def cpu_baseline(args):
    """The CPU oracle (a port of the reference's PyTorch training step) timed on this box's host cores: a bounded sample of whole steps at batch 1
    (the reference's default batch size, cli.py:110).  Runs in a child process with a time limit so that a slow / oversubscribed
    host cannot stall the benchmark; falls back to a 256x256 tile (reported in 512x512-tile equivalents) if 512x512 does not
    finish in time."""
    import subprocess
    for size, limit in ((args.size, 240), (args.size // 2, 180)):
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-child', '--norm', args.norm, '--size', str(size)],
                               capture_output=True, text=True, timeout=limit, env=dict(os.environ, HIP_VISIBLE_DEVICES=''))
            line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
            d = json.loads(line)
        except Exception as e:            # timeout / crash: try the smaller sample, else report nothing
            last = f'{type(e).__name__}'
            continue
        scale = (size * size) / float(args.size * args.size)
        return {'value': round(scale / d['seconds'], 5), 'unit': 'tiles/s', 'cores': d['cores'], 'kind': 'port',
                'sample': f"{d.get('steps', 1)} optimize_parameters() step(s) of the fp32 CPU oracle (5 Resnet-9 G + 5 NLayer D, GAN+SmoothL1+Adam), batch 1, "
                          f"{size}x{size} tile, {d.get('total_seconds', d['seconds']):.1f} s of CPU work, {d['seconds']:.1f} s per step" + ('' if size == args.size else f' (scaled to {args.size}x{args.size}-tile units by pixel count)')}
    return {'value': None, 'unit': 'tiles/s', 'cores': usable_cores(), 'kind': 'port', 'sample': f'CPU oracle step did not finish within the time limit ({last})'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=8, help='tiles per GPU per step')
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32', 'fp32_bf16mma'])
    ap.add_argument('--norm', default='instance', choices=['instance', 'batch'])
    ap.add_argument('--workload', default='train', choices=['train', 'train18', 'ext', 'infer'],
                    help="train = BASELINE's 5G+5D step (the contract line); train18 = the real DeepLIIF configuration (modalities_no=4, seg_gen: "
                         '4 Resnet-9 + 5 UNet-512 generators + 9 discriminators, SURVEY 8d); ext = BASELINE configs[3], DeepLIIFExt with 2 modalities: '
                         '2 Resnet-9 + 2 UNet-512 (9-channel input) generators, 2 + 2 discriminators (6 / 12 channels); infer = configs[1]')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-child', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        return cpu_baseline_child(args.norm, args.size)

    from deepliif_amd import distributed as D
    from deepliif_amd import models as M
    from deepliif_amd import ops
    rank, world, local_rank = D.init_process_group_from_env('nccl')
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    sys.stdout = open(os.devnull, 'w')      # the model classes print like the reference does; the contract is ONE JSON line

    torch.manual_seed(0)
    opt = make_opt(args, local_rank)
    n, s = args.batch, args.size

    def synth(seed):
        g = torch.Generator().manual_seed(seed + 1000 * rank)       # distinct tiles per rank (data-parallel shards)
        return (torch.rand(n, 3, s, s, generator=g) * 2 - 1).to(dev)

    if args.workload == 'train':
        model = M.create_model(opt)
        model.setup(opt)
        batch = {'A': synth(1234), 'B': [synth(1235 + i) for i in range(5)], 'A_paths': ['synthetic']}

        def step():
            model.set_input(batch)
            model.optimize_parameters()
        gf_per_tile = GF_PER_TILE_TRAIN_5R5D
        dom_shape = (n, s // 4, s // 4, 256)
        workload = 'DeepLIIF train step, 5x Resnet-9block G + 5x NLayerD(n=4), GAN+SmoothL1+Adam (BASELINE configs[2] per GPU)'
    elif args.workload == 'ext':
        opt = make_opt(args, local_rank, M=2, seg_gen=True)
        opt.model, opt.net_ds = 'DeepLIIFExt', 'n_layers'
        opt.loss_G_weights = opt.loss_D_weights = opt.seg_weights = [0.5, 0.5]
        model = M.create_model(opt)
        model.setup(opt)
        batch = {'A': synth(1234), 'B': [synth(1235 + i) for i in range(2)], 'BS': [synth(1255 + i) for i in range(2)], 'A_paths': ['synthetic']}

        def step():
            model.set_input(batch)
            model.optimize_parameters()
        # per tile: generators forward + 2x backward; every discriminator: 2 forwards + 2x2 backward in backward_D, 1 forward + 1 dgrad
        # in backward_G = 8 forward-equivalents (the accounting SURVEY 8d uses for the 5G+5D figure: 40 x 21.8 for 5 D)
        gf_per_tile = 3 * (2 * 396.4 + 2 * 49.2) + 8 * (2 * 21.8 + 2 * 22.6)
        dom_shape = (n, s // 4, s // 4, 256)
        workload = ('DeepLIIFExt train step, modalities_no=2: 2x Resnet-9block + 2x UNet-512 (9-ch in) generators, 2x NLayerD (6 ch) + 2x NLayerD '
                    '(12 ch), GAN/LSGAN+SmoothL1+Adam (BASELINE configs[3])')
    elif args.workload == 'train18':
        opt = make_opt(args, local_rank, M=4, seg_gen=True)
        model = M.create_model(opt)
        model.setup(opt)
        batch = {'A': synth(1234), 'B': [synth(1235 + i) for i in range(5)], 'A_paths': ['synthetic']}      # 4 modalities + seg target

        def step():
            model.set_input(batch)
            model.optimize_parameters()
        gf_per_tile = GF_PER_TILE_TRAIN_18NETS
        dom_shape = (n, s // 4, s // 4, 256)
        workload = ('real DeepLIIF train step (modalities_no=4, seg_gen=True): 4x Resnet-9block + 5x UNet-512 generators, 4 + 5 NLayerD(n=4), '
                    'GAN/LSGAN+SmoothL1+Adam (SURVEY 8d)')
    else:
        from deepliif_amd import inference as I
        iopt = types.SimpleNamespace(model='DeepLIIF', modalities_no=4, seg_gen=True, mod_id_seg='S', input_id=0, input_nc=3, output_nc=3,
                                     ngf=64, norm='batch', padding='zero', net_g='resnet_9blocks', net_gs='unet_512', input_no=1,
                                     modalities_names=['IHC', 'Hema', 'DAPI', 'Lap2', 'Marker'], gpu_ids=[local_rank])
        nets = I.build_generators(iopt, dev, args.precision)
        tiles = synth(1234)

        def step():
            I.run_generators(tiles, nets, iopt, seg_weights=[0.25, 0.15, 0.25, 0.1, 0.25])
        gf_per_tile = GF_PER_TILE_INFER
        dom_shape = (n, s // 4, s // 4, 256)
        workload = 'DeepLIIF inference, 4x Resnet-9block + 5x UNet-512 generators + weighted seg sum (BASELINE configs[1])'

    timer = KernelTimer(ops.impl(), dom_shape)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    tiles_total = args.steps * n * world
    value = tiles_total / dt
    kt = timer.mean_seconds()
    flops_per_launch = 2.0 * n * (s // 4) * (s // 4) * 256 * 256 * 9
    roofline = None
    traffic = None
    try:        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (tools/gpu_pmc.sh)
        with open(os.path.join(ROOT, 'profiles', 'r01', 'pmc_dominant_conv256.json')) as f:
            pmc = json.load(f)
            # counters belong to ONE kernel at ONE shape: report them only when that is what this run dispatched
            same = (n, s, args.precision) == (8, 512, 'bf16') and timer.kernel != '?' and timer.kernel.split('<')[0] in pmc.get('kernel', '')
            traffic = pmc['traffic_bytes'] if same else None
    except Exception:
        traffic = None
    if kt:
        ach = flops_per_launch / kt / 1e12
        roofline = {'bound': 'mfma', 'achieved': round(ach, 1), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_BF16_TFLOPS, 4),
                    'traffic': traffic, 'kernel': f'{timer.kernel} (256x256x64 tile, 8 waves): 3x3 256->256 @ {n}x{s // 4}x{s // 4}, ResnetBlock conv ' + ('fwd + dgrad' if args.workload != 'infer' else 'fwd only') + '; timed by events around the host call',
                    'launches_timed': len(timer.pairs), 'avg_launch_us': round(kt * 1e6, 2)}
    out = {
        'metric': {'train': '512x512 tiles/s train-step (5G+5D)', 'train18': '512x512 tiles/s train-step (real DeepLIIF: 9 G + 9 D)', 'ext': '512x512 tiles/s train-step (DeepLIIFExt, 2 modalities: 4 G + 4 D)',
                   'infer': '512x512 tiles/s inference (4 Resnet-9 + 5 UNet-512)'}[args.workload],
        'value': round(value, 3), 'unit': 'tiles/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'bf16' if args.precision == 'bf16' else ('f32(split-bf16x3 MFMA)' if args.precision == 'fp32' else 'f32 storage/bf16 MFMA'),
        'data': 'synthetic U(-1,1) tiles (seeds 1234..), N(0,0.02) random-init weights (torch.manual_seed(0)), dropout off, VGG loss off',
        'config': {'workload': workload, 'tile': f'{s}x{s}x3', 'batch_per_gpu': n, 'global_batch': n * world, 'norm': args.norm,
                   'precision_policy': args.precision, 'parallelism': f'dp{world}'},
        'model_tflops': round(value * gf_per_tile / 1e3, 1),
        'model_frac_of_bf16_peak': round(value * gf_per_tile / 1e3 / (PEAK_BF16_TFLOPS * world), 4),
        'roofline': roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == 'train':
        out['cpu_baseline'] = cpu_baseline(args)
    else:
        out['cpu_baseline'] = None
        out['cpu_baseline_note'] = ('disabled by --no-cpu-baseline' if args.no_cpu_baseline else
                                    'only timed on rank 0 of a 1-GPU run' if world != 1 else
                                    'the CPU oracle leg is implemented for the train workload only (oracle optimize_parameters); '
                                    'run the default workload for the CPU baseline')
    if rank == 0:
        sys.stdout = sys.__stdout__
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
