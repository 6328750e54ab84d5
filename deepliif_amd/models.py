"""Drop-in model classes for the training seam: create_model(opt) -> model with set_input / optimize_parameters /
calculate_losses / forward / test / eval / train / get_current_losses / get_current_visuals / save_networks / load_networks /
update_learning_rate -- the surface cli.py:403-449, train.py and test.py:90-113 rely on.

DeepLIIFModel mirrors deepliif/models/DeepLIIF_model.py (network naming :49-115, forward :175-203, backward_D :205-332,
backward_G :334-429, optimize_parameters :431-467) but executes on the MI355X engine: explicit forward/backward tape over
hand-written HIP kernels, one fused Adam kernel per parameter set, gradient exchange on flat buffers.
The VGG19 perceptual term of the reference (:406-409) runs on the engine as well (networks.VGGLoss) but needs torchvision's weights as a FILE
(opt.vgg_weights / DEEPLIIF_VGG19_WEIGHTS; nothing is downloaded): BaseModel._make_vgg raises when lambda_feat > 0 and none is given.
DeepLIIFExtModel / SDGModel, DeepLIIFKDModel (teacher through inference.init_nets, distillation terms through engine.kldiv_op) and CycleGANModel
(image pools, generator update first) follow the same pattern; create_model() dispatches on opt.model like deepliif/models/__init__.py:101-114.
"""
from __future__ import annotations

import contextlib
import os
import sys
from collections import OrderedDict
from typing import List, Optional

import torch

from . import _lib as L
from . import engine as E
from . import distributed as D
from . import networks
from . import ops
from .distributed import GradExchanger

# DL_STREAMS=N (default 3; 1 = everything on torch's current stream): the independent (G_i, D_i) branches of a DeepLIIF training step on N HIP streams
# (BaseModel._branch_streams).  Measured r04, same box, 5G+5D step at batch 8: 94.8 ms on one stream, 88.6 on two, 83.8 on three, 84.0 on five -- the
# ~600 launch-latency-bound kernels of a step, every big kernel's ramp and drain and the HBM-bound norm passes overlap with another branch's MFMA work.
_N_STREAMS = max(1, int(os.environ.get('DL_STREAMS', '3')))
# DL_STREAMS_SEG=0 switches off (default on since round 5, the complete GPU suite runs with it): branch streams also for the model WITH segmentation generators (the reference's default configuration): chain i =
# G_i -> GS_i on its own stream, the seg discriminators and everything that reads the summed seg image on the main stream (DeepLIIFModel.forward)
_SEG_STREAMS = os.environ.get('DL_STREAMS_SEG', '1') == '1'
# DL_STREAMS_EXT=0: DeepLIIFExt / SDG on one stream (round 4); default: their chains G_i -> GS_i (+ D_i, DS_i) on the branch streams as well
_EXT_STREAMS = os.environ.get('DL_STREAMS_EXT', '1') == '1'


def _get(opt, name, default):
    return getattr(opt, name) if hasattr(opt, name) else default


def init_input_and_mod_id(opt, dir_model=None):
    """deepliif/util/util.py:242-270 for a fresh training run / explicit opt fields (file-name sniffing of existing
    checkpoints is done by the reference's Options object, which passes mod_id_seg / input_id through opt)."""
    if hasattr(opt, 'mod_id_seg') and opt.mod_id_seg is not None:
        mod_id_seg = opt.mod_id_seg
    elif not hasattr(opt, 'modalities_names'):
        mod_id_seg = opt.modalities_no + 1
    else:
        mod_id_seg = 'S'
    input_id = str(_get(opt, 'input_id', '0'))
    return mod_id_seg, input_id


def pick_gpu(gpu_ids, is_train: bool, env=None) -> int:
    """The GPU this PROCESS trains / infers on.  One id: that one.  Several ids (`--gpu-ids 0 --gpu-ids 1`): the reference wraps every net in
    nn.DataParallel(net, gpu_ids) (networks.py:136, "Multi-GPU Training.md":20-33) -- ONE process scattering each batch.  This engine is one
    process per GPU: under torchrun / `deepliif trainlaunch` (LOCAL_RANK set, networks.py:131-134) rank r takes gpu_ids[r]; a single process
    with several ids re-executes itself as `python -m torch.distributed.run --nproc-per-node <n>` when DEEPLIIF_AMD_AUTO_TORCHRUN=1 and otherwise
    stops with that command line -- the ONE behaviour change of the CLI.  Inference uses gpu_ids[0]."""
    env = os.environ if env is None else env
    if not gpu_ids:
        raise L.HipLibraryError('deepliif_amd models run on MI355X only: opt.gpu_ids must name a GPU (no CPU fallback)')
    ids = list(gpu_ids)
    if len(ids) == 1 or not is_train:
        return int(ids[0])
    if 'LOCAL_RANK' in env and int(env.get('WORLD_SIZE', '1')) > 1:
        return int(ids[int(env['LOCAL_RANK']) % len(ids)])
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1', f'--nproc-per-node={len(ids)}'] + sys.argv
    if env.get('DEEPLIIF_AMD_AUTO_TORCHRUN') == '1':
        os.execv(sys.executable, cmd)
    raise NotImplementedError(f'single-process multi-GPU training (gpu_ids={ids}, nn.DataParallel in the reference) is not supported: this engine runs one '
                              f'process per GPU.  Launch:  {" ".join(cmd)}   (or set DEEPLIIF_AMD_AUTO_TORCHRUN=1 to have this done for you; every rank '
                              'then takes gpu_ids[LOCAL_RANK] and the gradients are averaged over RCCL, deepliif_amd.distributed)')


# A/B switch (DL_D_PAIR_BATCH=0): the discriminators see the fake pairs and the real pairs in two calls, as in rounds 1-5 (DeepLIIFModel.backward_D)
_D_PAIR_BATCH = os.environ.get('DL_D_PAIR_BATCH', '1') != '0'


class StepGraph:
    """optimize_parameters() captured ONCE in a hipGraph and replayed per step: a training step is ~2 400 kernel launches issued
    by ~55 ms of Python for ~100 ms of GPU time; a replay is one launch.  The shapes are static, the losses stay on the device, the tape is
    deterministic, so the captured launch sequence IS the step; what changes per step is data:
      * the batch: copied into persistent device tensors (`static batch`) before every step; set_input(static batch) -- the NCHW -> engine layout
        conversions -- is part of the captured region;
      * Adam's scalars (scheduler learning rate, bias corrections): optim.FusedAdam graph mode, refreshed by prepare_step() outside the graph.
    step(batch): the first `warmup` calls run eagerly (first-use initialisation: function attributes, pack tables, workspaces), the next one
    captures, every later one replays.  Results are bit-identical to the eager step (tests/test_gpu_graph.py: reference trajectories).
    Falls back to eager for good, saying why, when the step cannot be captured: data-parallel exchange active (collectives are launched from tape
    callbacks on RCCL's own streams), dropout in training mode (the mask seed is a kernel argument), an optimizer other than FusedAdam, a model
    class with host-side randomness (CycleGAN's ImagePool)."""

    def __init__(self, model, warmup: int = 2):
        self.model, self.warmup = model, max(2, warmup)      # two eager steps: the second one still creates state (the repack table sees the data-gradient images)
        self.calls, self.graph, self.static, self.why_eager = 0, None, None, None
        from . import distributed as D
        from .optim import FusedAdam
        if D.active():
            self.why_eager = 'data-parallel gradient exchange is active'
        elif _N_STREAMS > 1 and getattr(model, 'branch_parallel', False):
            self.why_eager = 'the branches run on several streams (DL_STREAMS)'
        elif not getattr(model, 'graphable', False):
            self.why_eager = f'{type(model).__name__} has host-side state per step'
        elif not all(isinstance(o, FusedAdam) for o in model.optimizers):
            self.why_eager = 'an optimizer other than FusedAdam'
        elif any(isinstance(m, torch.nn.Dropout) and m.training for _, net in model._nets() for m in net.modules()):
            self.why_eager = 'dropout is active (its seed is a kernel argument)'
        if self.why_eager:
            print(f'deepliif_amd: StepGraph runs eagerly: {self.why_eager}')
        else:
            for o in model.optimizers:
                o.enable_graph_mode()

    def _to_static(self, batch):
        dev = self.model.device

        def conv(v, ref):
            if torch.is_tensor(v):
                if ref is None:
                    return v.to(dev).clone()
                ref.copy_(v, non_blocking=True)
                return ref
            if isinstance(v, (list, tuple)) and v and all(torch.is_tensor(x) for x in v):
                return [conv(x, None if ref is None else ref[i]) for i, x in enumerate(v)]
            return v
        if self.static is None:
            self.static = {k: conv(v, None) for k, v in batch.items()}
        else:
            for k, v in batch.items():
                self.static[k] = conv(v, self.static.get(k))
        return self.static

    def _same_layout(self, batch) -> bool:
        """same keys, list lengths, shapes and dtypes as the batch the static tensors were made from"""
        def sig(v):
            if torch.is_tensor(v):
                return (tuple(v.shape), v.dtype)
            if isinstance(v, (list, tuple)) and v and all(torch.is_tensor(x) for x in v):
                return tuple(sig(x) for x in v)
            return None
        return batch.keys() == self.static.keys() and all(sig(v) == sig(self.static[k]) for k, v in batch.items())

    def step(self, batch):
        m = self.model
        if self.why_eager:
            m.set_input(batch)
            m.optimize_parameters()
            return
        m._sync_replicas()
        if self.static is not None and not self._same_layout(batch):
            # a batch the captured step was not recorded for (the reference's loaders keep the last, smaller batch of an epoch; copy_() would broadcast a
            # remainder of 1 into every slot without a word): this step runs eagerly on the scalar-argument Adam kernel (same arithmetic),
            # the graph stays valid for the batches that fit
            self.eager_steps = getattr(self, 'eager_steps', 0) + 1
            for o in m.optimizers:
                o.step_eager_once = True
            m.set_input(batch)
            m.optimize_parameters()
            if self.graph is not None and ops.Workspace.realloc_generation != self._ws_generation:
                # the odd batch was LARGER than the captured one somewhere: a grow-only scratch buffer or the slab arena was replaced, and the graph's kernel
                # nodes still carry the old (now freed) addresses.  Drop the graph; the next fitting batches warm up and capture again.
                torch.cuda.synchronize()
                self.graph, self.calls = None, 0
                self.recaptures = getattr(self, 'recaptures', 0) + 1
            return
        static = self._to_static(batch)
        for o in m.optimizers:
            o.prepare_step()
        self.calls += 1
        if self.graph is not None:
            self.graph.replay()
        elif self.calls <= self.warmup:
            m.set_input(static)
            m.optimize_parameters()
        else:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            wst = ops.WS._thread_state()
            try:
                wst['capturing'] = True             # (ops.Workspace._state: the capture stream keeps this thread's scratch state)
                try:
                    with torch.cuda.graph(g):
                        m.set_input(static)
                        m.optimize_parameters()
                finally:
                    wst['capturing'] = False
            except Exception as exc:
                # something on the step path still needs the host during the step (e.g. a table that is only complete after another eager step:
                # the batched weight-repack table grows while data-gradient images appear).  Nothing was executed; run this step -- and all later
                # ones -- eagerly.  The packed-weight stamps may have been advanced by the aborted pass: force a re-pack.
                import traceback
                where = [l.strip() for l in traceback.format_exc().splitlines() if 'deepliif_amd' in l and 'StepGraph' not in l][-1:]
                self.why_eager = f'capture failed: {str(exc).splitlines()[0][:160]} {where}'
                print(f'deepliif_amd: StepGraph runs eagerly: {self.why_eager}')
                abandon = getattr(ops.impl(), 'wgrad_abandon', None)
                if abandon is not None:
                    abandon()                        # weight gradients recorded by the aborted pass never ran: nothing of them may be reduced later
                for o in m.optimizers:               # back to the scalar-argument Adam kernel (same arithmetic); prepare_step() had counted this step
                    o.hyper_dev, o._prepared = None, False
                    o.step_count -= 1
                    o.flat.bump_epoch()
                m.set_input(static)
                m.optimize_parameters()
                return
            self.graph = g
            self._ws_generation = ops.Workspace.realloc_generation
            g.replay()                  # capture records, it does not execute: this step's work


class BaseModel:
    """deepliif/models/base_model.py surface."""

    def __init__(self, opt):
        self.opt = opt
        self.gpu_ids = opt.gpu_ids
        self.is_train = opt.is_train
        self.device = self._device_from_opt(opt)
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        self.loss_names: List[str] = []
        self.model_names: List[str] = []
        self.visual_names: List[str] = []
        self.optimizers = []
        self.image_paths = []
        self.metric = 0
        self.precision = E.Precision.get(_get(opt, 'precision', networks.DEFAULT_PRECISION))
        if self.precision.half == 'fp16':
            # the model classes are the TRAINING surface (set_input / optimize_parameters); the fp16 policy is served by init_nets() / run_dask() /
            # inference() / infer_region() (deepliif_amd/inference.py)
            raise ValueError("precision 'fp16' is an inference policy (gradients of this model underflow IEEE half): build the model with "
                             "'bf16' or 'fp32', serve fp16 through deepliif_amd.inference")

    def _device_from_opt(self, opt) -> torch.device:
        dev = torch.device('cuda:{}'.format(pick_gpu(opt.gpu_ids, self.is_train)))
        torch.cuda.set_device(dev)          # cli.py:250-256 does this before building the model; kernels launch on the tensors' device anyway
        return dev

    def _make_vgg(self, opt):
        """(lambda_feat, VGGLoss or None).  The reference builds VGGLoss() from DOWNLOADED torchvision weights (networks.py:701) and trains
        with lambda_feat = 100 (options/__init__.py:67).  There is no network here: the weights come from a file (opt.vgg_weights or
        $DEEPLIIF_VGG19_WEIGHTS, a torchvision vgg19 state_dict).  With lambda_feat > 0 and no file the objective would silently differ from
        the reference's, so that is an error unless the caller opts out explicitly (opt.allow_no_vgg / DEEPLIIF_AMD_ALLOW_NO_VGG=1)."""
        lam = float(_get(opt, 'lambda_feat', 0) or 0)
        if lam <= 0:
            return 0.0, None
        path = networks.vgg_weights_path(opt)
        if path:
            return lam, networks.VGGLoss(path, self.device, self.precision.name)
        if _get(opt, 'allow_no_vgg', False) or os.environ.get('DEEPLIIF_AMD_ALLOW_NO_VGG') == '1':
            print('deepliif_amd: lambda_feat = %g but no VGG19 weight file was given: training WITHOUT the perceptual term (explicit opt-out)' % lam)
            return 0.0, None
        raise RuntimeError('lambda_feat = %g asks for the VGG19 perceptual loss (DeepLIIF_model.py:406-409), which needs torchvision\'s pretrained vgg19 weights; '
                           'pass them as a file (opt.vgg_weights or DEEPLIIF_VGG19_WEIGHTS=/path/vgg19.pth), set lambda_feat = 0, or opt out with '
                           'opt.allow_no_vgg / DEEPLIIF_AMD_ALLOW_NO_VGG=1' % lam)

    def _mark_net(self, tape, net):
        """Tape marker in front of a network's FIRST forward node of a pass: reverse mode reaches it after the network's last weight
        gradient of that pass, which is when its slice of the flat gradient can go on the wire (distributed.GradExchanger.ready)."""
        if tape is None or not self.is_train:
            return
        params = [p for p in net.parameters()]

        def net_done():
            # the network's queued weight gradients (ops.HipBackend: batched launch of its ResnetBlock layers) run now, on this branch's stream: the
            # batches are per network whatever the number of branch streams or ranks, and dL/dy / x of one network at a time are kept alive
            flush = getattr(ops.impl(), 'wgrad_flush', None)
            if flush is not None and params and params[0].is_cuda:
                flush()
            self.exchange.ready(params)
        tape.record(net_done)

    def _hook_tape(self, tape):
        """a training tape reports every parameter whose gradient has become final to the gradient exchange (progressive buckets of the
        networks above 64 MB, distributed.GradExchanger.param_final)"""
        if tape is not None and self.is_train and getattr(self, 'exchange', None) is not None:
            tape.on_final = self.exchange.param_final
        return tape

    def _register_nets_with_exchange(self):
        for o in self.optimizers:
            flat = getattr(o, 'flat', None)
            if flat is None:
                continue
            o.dp_tag = next((n[len('optimizer_'):] for n in ('optimizer_G', 'optimizer_D') if getattr(self, n, None) is o), 'opt')      # diagnostics label
            owned = {id(p) for p in flat.params}
            for _, net in self._nets():
                params = list(net.parameters())
                if params and id(params[0]) in owned:
                    self.exchange.register_net(o, params)

    # ---- branch streams (DL_STREAMS=N, opt-in): independent (generator, discriminator) branches of a training step on N HIP streams -----------
    def _branch_streams(self):
        """the HIP streams the branches of a training step are spread over, or None (the default: everything on torch's current stream).
        Only where the branches are independent (DeepLIIFModel without the segmentation generators: G_i and D_i of modality i touch nothing
        of modality j).  The data-parallel exchange works unchanged: a network's all-reduce is issued from a tape node of ITS branch, i.e. with that
        branch's stream current, and RCCL orders the collective behind the current stream; finish() runs on the main stream after the join."""
        if not hasattr(self, '_streams'):
            n = _N_STREAMS if (self.is_train and self.device.type == 'cuda' and getattr(self, 'branch_parallel', False)) else 1
            self._streams = [torch.cuda.Stream(self.device) for _ in range(n)] if n > 1 else None
            if self._streams is not None:
                ops.WS.branch_streams_on(self._streams)
        return self._streams

    def _fork(self, tape=None):
        """start of a phase: every branch stream waits for what the main stream has issued so far (inputs, zeroed gradients, repacked weights).
        With a tape the fork is a point INSIDE a recorded pass (branches consume a main-stream result): its mirror image is recorded, so that in
        backward the main stream waits for every branch there before it runs the nodes recorded in front of this point."""
        streams = self._branch_streams()
        if streams:
            main = torch.cuda.current_stream()
            for s in streams:
                s.wait_stream(main)
            if tape is not None:
                tape.record(self._join)

    def _join(self, tape=None):
        """end of a phase: the main stream (optimizer step, loss read-out) waits for every branch.  With a tape the join is a point INSIDE a recorded pass
        (a main-stream op consumes branch results): its mirror image is recorded, so that in backward every branch waits for the main stream there."""
        streams = self._branch_streams()
        if streams:
            main = torch.cuda.current_stream()
            for s in streams:
                main.wait_stream(s)
            if tape is not None:
                tape.record(self._fork)

    def _branch(self, i):
        """context: the kernels launched (and the tensors allocated) inside belong to the stream of branch i"""
        streams = self._branch_streams()
        return torch.cuda.stream(streams[i % len(streams)]) if streams else contextlib.nullcontext()

    def _new_tape(self):
        return E.Tape(streams=self._branch_streams() is not None)

    def _sync_replicas(self):
        """once, before the first step: broadcast parameters / BatchNorm buffers from rank 0 (what DistributedDataParallel does at
        construction, networks.py:134)"""
        if getattr(self, '_replicas_synced', False):
            return
        self._register_nets_with_exchange()
        nets = [net for _, net in self._nets()]
        for o in self.optimizers:
            self.exchange.sync_parameters(o, nets)
        self._replicas_synced = True

    def _net_gpu_ids(self):
        return self.gpu_ids

    # -- lifecycle -------------------------------------------------------------------------------------------
    def setup(self, opt):
        self.opt = opt
        if self.is_train:
            self.schedulers = [networks.get_scheduler(o, opt) for o in self.optimizers]
        if not self.is_train or _get(opt, 'continue_train', False):
            suffix = 'iter_%d' % opt.load_iter if _get(opt, 'load_iter', 0) > 0 else opt.epoch
            self.load_networks(suffix)
        self.print_networks(_get(opt, 'verbose', False))
        E.settle_gc()                       # the model is long-lived: keep full garbage collections from walking it (engine.settle_gc)

    def _net(self, name):
        # 'G_1' -> self.netG[0] (list-valued models: DeepLIIFExt/SDG, base_model.py:96-98), 'G1' -> self.netG1
        if '_' in name:
            kind, idx = name.split('_')
            return getattr(self, 'net' + kind)[int(idx) - 1]
        return getattr(self, 'net' + name)

    def _nets(self):
        return [(n, self._net(n)) for n in self.model_names]

    def train(self):
        for _, net in self._nets():
            net.train()
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d) and hasattr(m, 'running_mean_backup'):
                    m.track_running_stats = True
                    m.running_mean, m.running_var = m.running_mean_backup, m.running_var_backup

    def eval(self):
        # base_model.py:103-112 + util/__init__.py:743-755: eval mode still normalises with batch statistics
        for _, net in self._nets():
            net.eval()
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d) and m.running_mean is not None:
                    m.track_running_stats = False
                    m.running_mean_backup, m.running_var_backup = m.running_mean, m.running_var
                    m.running_mean = m.running_var = None

    def test(self):
        self.forward(record=False)
        self.compute_visuals()

    def compute_visuals(self):
        pass

    def get_image_paths(self):
        return self.image_paths

    def update_learning_rate(self):
        for s in self.schedulers:
            if self.opt.lr_policy == 'plateau':
                s.step(self.metric)
            else:
                s.step()
        print('learning rate = %.7f' % self.optimizers[0].param_groups[0]['lr'])

    def get_current_visuals(self):
        return OrderedDict((n, getattr(self, n)) for n in self.visual_names if hasattr(self, n))

    def get_current_losses(self):
        # one device->host copy for all losses (the reference syncs once per loss: base_model.py:173-188)
        vals = self._loss_buf.detach().cpu().tolist() if hasattr(self, '_loss_buf') else []
        return OrderedDict((n, float(vals[self._loss_index[n]])) for n in self.loss_names)

    def save_networks(self, epoch, save_from_one_process=False):
        os.makedirs(self.save_dir, exist_ok=True)
        for name, net in self._nets():
            sd = OrderedDict((k, v.detach().cpu().clone()) for k, v in net.state_dict().items())
            torch.save(sd, os.path.join(self.save_dir, '%s_net_%s.pth' % (epoch, name)))

    def save_optimizers(self, epoch):
        """<epoch>_optimizer_<i>.pth per optimizer, in torch.optim's checkpoint layout (the reference saves no optimizer state,
        base_model.py:190-208 -- an extension).  Schedulers are a pure function of the epoch count and are not stored."""
        os.makedirs(self.save_dir, exist_ok=True)
        for i, o in enumerate(self.optimizers):
            torch.save(o.state_dict(), os.path.join(self.save_dir, '%s_optimizer_%d.pth' % (epoch, i)))

    def load_optimizers(self, epoch):
        for i, o in enumerate(self.optimizers):
            o.load_state_dict(torch.load(os.path.join(self.save_dir, '%s_optimizer_%d.pth' % (epoch, i)), map_location='cpu'))

    def load_networks(self, epoch):
        for name, net in self._nets():
            path = os.path.join(self.save_dir, '%s_net_%s.pth' % (epoch, name))
            print('loading the model from %s' % path)
            sd = torch.load(path, map_location='cpu')
            if hasattr(sd, '_metadata'):
                del sd._metadata
            net.load_state_dict(sd)          # copies into the (flat-buffer backed) parameters in place

    def print_networks(self, verbose):
        print('---------- Networks initialized -------------')
        for name, net in self._nets():
            if verbose:
                print(net)
            print('[Network %s] Total number of parameters : %.3f M' % (name, sum(p.numel() for p in net.parameters()) / 1e6))
        print('-----------------------------------------------')

    def set_requires_grad(self, nets, requires_grad=False):
        if not isinstance(nets, list):
            nets = [nets]
        for net in nets:
            if net is not None:
                for p in net.parameters():
                    p.requires_grad = requires_grad


class DeepLIIFModel(BaseModel):
    graphable = True          # may StepGraph capture optimize_parameters()? (no host-side randomness / per-step host state)
    def __init__(self, opt):
        super().__init__(opt)
        if not hasattr(opt, 'net_gs'):
            opt.net_gs = 'unet_512'
        self.seg_gen = opt.seg_gen
        self.seg_weights = list(opt.seg_weights)
        self.loss_G_weights = list(opt.loss_G_weights)
        self.loss_D_weights = list(opt.loss_D_weights)
        self.mod_id_seg, self.input_id = init_input_and_mod_id(opt)
        M, S = opt.modalities_no, self.mod_id_seg

        # ---- names (DeepLIIF_model.py:33-80)
        self.loss_names = []
        self.visual_names = ['real_A']
        for i in range(1, M + 1):
            self.loss_names += [f'G_GAN_{i}', f'G_L1_{i}', f'D_real_{i}', f'D_fake_{i}']
            self.visual_names += [f'fake_B_{i}', f'real_B_{i}']
        if self.seg_gen:
            self.loss_names += [f'G_GAN_{S}', f'G_L1_{S}', f'D_real_{S}', f'D_fake_{S}']
            for i in range(M + 1):
                self.visual_names += [f'fake_B_{S}{i}']
            self.visual_names += [f'fake_B_{S}', f'real_B_{S}']
        off = 0 if self.input_id == '0' else 1
        self.model_names_g = [f'G{i}' for i in range(1, M + 1)]
        self.model_names_gs = [f'G{S}{i + off}' for i in range(M + 1)] if self.seg_gen else []
        self.model_names_d = [f'D{i}' for i in range(1, M + 1)] if self.is_train else []
        self.model_names_ds = [f'D{S}{i + off}' for i in range(M + 1)] if (self.is_train and self.seg_gen) else []
        if self.is_train:
            self.model_names = []
            for i in range(M):
                self.model_names += [self.model_names_g[i], self.model_names_d[i]]
            for i in range(len(self.model_names_gs)):
                self.model_names += [self.model_names_gs[i], self.model_names_ds[i]]
        else:
            self.model_names = self.model_names_g + self.model_names_gs

        # ---- networks, constructed in the reference's order so that seeded init matches (DeepLIIF_model.py:83-115)
        netG = getattr(opt, 'netG', None) or getattr(opt, 'net_g')
        if isinstance(netG, str):
            netG = [netG] * M
        opt.netG = netG
        if isinstance(opt.net_gs, str):
            opt.net_gs = [opt.net_gs] * (M + 1)
        use_dropout = not _get(opt, 'no_dropout', True)
        cin = opt.input_nc * _get(opt, 'input_no', 1)
        gpu = self._net_gpu_ids()
        for i, n in enumerate(self.model_names_g):
            setattr(self, 'net' + n, networks.define_G(cin, opt.output_nc, opt.ngf, netG[i], opt.norm, use_dropout, opt.init_type, opt.init_gain,
                                                         gpu, opt.padding, _get(opt, 'upsample', 'convtranspose')))
        for i, n in enumerate(self.model_names_gs):
            # define_G's default padding_type ('reflect') applies to the seg generators (DeepLIIF_model.py:95-99)
            setattr(self, 'net' + n, networks.define_G(cin, opt.output_nc, opt.ngf, opt.net_gs[i], opt.norm, use_dropout, opt.init_type,
                                                         opt.init_gain, gpu))
        netD = getattr(opt, 'netD', None) or _get(opt, 'net_d', 'n_layers')
        n_layers_D = _get(opt, 'n_layers_D', 4)
        for n in self.model_names_d + self.model_names_ds:
            setattr(self, 'net' + n, networks.define_D(cin + opt.output_nc, opt.ndf, netD, n_layers_D, opt.norm, opt.init_type, opt.init_gain, gpu))
        for _, net in self._nets():
            net.set_precision(self.precision.name)

        # ---- losses: one fp32 device buffer, loss_<name> attributes are 0-dim views (no host sync in the step)
        self._loss_index = {n: i for i, n in enumerate(self.loss_names)}
        self._loss_buf = torch.zeros(max(len(self.loss_names), 1), dtype=torch.float32, device=self.device)
        for n, i in self._loss_index.items():
            setattr(self, 'loss_' + n, self._loss_buf[i])

        if self.is_train:
            self.criterionGAN_mod = networks.GANLoss(opt.gan_mode).to(self.device)
            self.criterionGAN_seg = networks.GANLoss(opt.gan_mode_s).to(self.device)
            self.lambda_L1 = _get(opt, 'lambda_L1', 100.0)
            self.lambda_feat, self.criterionVGG = self._make_vgg(opt)
            # branches on several streams only where they are independent: no segmentation generators (they read the other branches' fakes), no VGG term
            # (evaluated outside the branches), none of the subclasses' extra terms
            self.branch_parallel = type(self) is DeepLIIFModel and self.criterionVGG is None and (not self.seg_gen or _SEG_STREAMS)
            params_g = [p for n in self.model_names_g + self.model_names_gs for p in getattr(self, 'net' + n).parameters()]
            params_d = [p for n in self.model_names_d + self.model_names_ds for p in getattr(self, 'net' + n).parameters()]
            OptCls = networks.get_optimizer(_get(opt, 'optimizer', 'adam'))
            try:
                self.optimizer_G = OptCls(params_g, lr=opt.lr_g, betas=(opt.beta1, 0.999))
                self.optimizer_D = OptCls(params_d, lr=opt.lr_d, betas=(opt.beta1, 0.999))
            except TypeError:
                self.optimizer_G = OptCls(params_g, lr=opt.lr_g)
                self.optimizer_D = OptCls(params_d, lr=opt.lr_d)
            self.optimizers += [self.optimizer_G, self.optimizer_D]
            self.exchange = GradExchanger()
            self._vgg_buf = torch.zeros(max(M, 1), dtype=torch.float32, device=self.device)
        self._tape_G: Optional[E.Tape] = None

    # ---------------------------------------------------------------------------------------------------------
    def set_input(self, input):
        """DeepLIIF_model.py:153-173: dict{'A': Tensor|list, 'B': list[Tensor], 'A_paths'}; tensors NCHW fp32 in [-1, 1]."""
        A = input['A']
        if isinstance(A, list):
            A = torch.cat([a.to(self.device) for a in A], dim=1)
        self.real_A = A.to(self.device, non_blocking=True)
        self.real_B_array = input['B']
        M, S = self.opt.modalities_no, self.mod_id_seg
        for i in range(M):
            setattr(self, f'real_B_{i + 1}', self.real_B_array[i].to(self.device, non_blocking=True))
        if self.seg_gen:
            setattr(self, f'real_B_{S}', self.real_B_array[M].to(self.device, non_blocking=True))
        self.image_paths = input.get('A_paths', [])
        # engine-layout copies, made once per batch
        p = self.precision
        self._A = E.to_engine(self.real_A, p)
        self._B = [E.to_engine(getattr(self, f'real_B_{i + 1}'), p) for i in range(M)]
        self._Bseg = E.to_engine(getattr(self, f'real_B_{S}'), p) if self.seg_gen else None
        self._real_pairs = None

    def _ctx(self, tape, training=True):
        return E.Ctx(self.precision, self._hook_tape(tape), training=training)

    def forward(self, record: Optional[bool] = None):
        """DeepLIIF_model.py:175-203."""
        record = self.is_train if record is None else record
        tape = self._new_tape() if record else None
        ctx = self._ctx(tape, training=record)
        M, S = self.opt.modalities_no, self.mod_id_seg
        self._fake = []
        self._fork()
        for i, n in enumerate(self.model_names_g):
            with self._branch(i):
                self._mark_net(tape, getattr(self, 'net' + n))
                f = getattr(self, 'net' + n).run(ctx, self._A)
                self._fake.append(f)
                setattr(self, f'fake_B_{i + 1}', E.from_engine(f))
        if not record:
            self._join()
        if self.seg_gen:
            parts = []
            for i, n in enumerate(self.model_names_gs):
                src = self._A if i == 0 else self._fake[i - 1]
                # seg generator i >= 1 reads fake_B_i: it continues the chain of G_i on THAT branch's stream; seg generator 0 (on real_A) gets the next stream
                with self._branch(i - 1 if i > 0 else M):
                    self._mark_net(tape, getattr(self, 'net' + n))
                    s = getattr(self, 'net' + n).run(ctx, src)
                    parts.append(s)
                    setattr(self, f'fake_B_{S}_{i}', E.from_engine(s))
            self._fake_seg_parts = parts
            # the weighted sum reads every chain's result: the main stream waits for the branches here -- and, on the way back, every branch waits for
            # the main stream at this point of the tape before it takes its part of the seg image's gradient (the tape node below)
            self._join(tape)
            self._fake_seg = E.weighted_sum(ctx, parts, self.seg_weights[:M + 1])
            setattr(self, f'fake_B_{S}', E.from_engine(self._fake_seg))
        self._tape_G = tape

    # ---------------------------------------------------------------------------------------------------------
    def _pairs_real(self, ctx):
        if self._real_pairs is None:
            M = self.opt.modalities_no
            mod = []
            for i in range(M):
                with self._branch(i):
                    mod.append(E.concat_channels(ctx, [self._A, self._B[i]]))
            seg = []
            if self.seg_gen:
                for i in range(M + 1):
                    cond = self._A if i == 0 else self._B[i - 1]
                    seg.append(E.concat_channels(ctx, [cond, self._Bseg]))
            self._real_pairs = (mod, seg)
        return self._real_pairs

    def _seg_pred(self, ctx, image: E.Act):
        M = self.opt.modalities_no
        preds = []
        for i, n in enumerate(self.model_names_ds):
            cond = self._A if i == 0 else self._B[i - 1]        # real modalities condition the seg discriminators (:253-255)
            preds.append(getattr(self, 'net' + n).run(ctx, E.concat_channels(ctx, [cond, image])))
        return E.weighted_sum(ctx, preds, self.seg_weights[:M + 1])   # weighted BEFORE the lsgan loss (:258-262)

    def backward_D(self):
        """DeepLIIF_model.py:205-332: D losses on detached fakes and on real pairs."""
        tape = self._new_tape()
        ctx = self._ctx(tape)
        M, S = self.opt.modalities_no, self.mod_id_seg
        wD = self.loss_D_weights
        cg, cs = self.criterionGAN_mod, self.criterionGAN_seg
        self._fork()                             # (the gradients were zeroed on the main stream)
        # A discriminator that normalises every sample on its own statistics (InstanceNorm, or no norm layer) sees the fake pairs and the real pairs as ONE batch of
        # 2N: sample by sample the same arithmetic as the reference's two calls (DeepLIIF_model.py:222-236), but every launch has twice the tiles -- the PatchGAN's
        # GEMM-shaped layers are the under-filled kernels of this step (64-128 tiles of 256 x 256 on 256 CUs at batch 8) -- and there are half as many launches.
        # BatchNorm discriminators keep the two calls (their statistics are per call).
        paired = _D_PAIR_BATCH and all(getattr(getattr(self, 'net' + n), 'norm_kind', 'batch') != 'batch' for n in self.model_names_d)
        real_mod_done = False
        for i, n in enumerate(self.model_names_d):
            with self._branch(i):
                # every discriminator runs twice below (fake, real): mark before the first use -- on the stream of its branch: the marker's all-reduce is
                # ordered behind THAT stream
                self._mark_net(tape, getattr(self, 'net' + n))
                if paired:
                    nb, h, w, _ = self._A.t.shape
                    both = torch.zeros((2 * nb, h, w, E.cpad(self._A.C + self._fake[i].C)), dtype=self._A.t.dtype, device=self._A.t.device)
                    E.concat_channels(ctx, [self._A, self._fake[i].detach()], out=both[:nb])
                    E.concat_channels(ctx, [self._A, self._B[i]], out=both[nb:])
                    pred = getattr(self, 'net' + n).run(ctx, E.Act(both, self._A.C + self._fake[i].C))
                    E.loss_op_halves(ctx, cg.kind, pred, (cg.target(False), cg.target(True)), 0.5 * wD[i],
                                     (getattr(self, f'loss_D_fake_{i + 1}').view(1), getattr(self, f'loss_D_real_{i + 1}').view(1)))
                    continue
                pair = E.concat_channels(ctx, [self._A, self._fake[i].detach()])
                pred = getattr(self, 'net' + n).run(ctx, pair)
                E.loss_op(ctx, cg.kind, pred, None, cg.target(False), 0.5 * wD[i], getattr(self, f'loss_D_fake_{i + 1}').view(1))
        real_mod_done = paired
        for n in self.model_names_ds:
            self._mark_net(tape, getattr(self, 'net' + n))
        paired_seg = self.seg_gen and _D_PAIR_BATCH and all(getattr(getattr(self, 'net' + n), 'norm_kind', 'batch') != 'batch' for n in self.model_names_ds)
        if paired_seg:
            # the five segmentation discriminators, each on cat(fake pairs, real pairs); weighted BEFORE the loss as in _seg_pred (:258-262)
            preds = []
            nb, h, w, _ = self._A.t.shape
            for i, n in enumerate(self.model_names_ds):
                cond = self._A if i == 0 else self._B[i - 1]
                both = torch.zeros((2 * nb, h, w, E.cpad(cond.C + self._fake_seg.C)), dtype=cond.t.dtype, device=cond.t.device)
                E.concat_channels(ctx, [cond, self._fake_seg.detach()], out=both[:nb])
                E.concat_channels(ctx, [cond, self._Bseg], out=both[nb:])
                preds.append(getattr(self, 'net' + n).run(ctx, E.Act(both, cond.C + self._fake_seg.C)))
            pred = E.weighted_sum(ctx, preds, self.seg_weights[:M + 1])
            E.loss_op_halves(ctx, cs.kind, pred, (cs.target(False), cs.target(True)), 0.5 * wD[M],
                             (getattr(self, f'loss_D_fake_{S}').view(1), getattr(self, f'loss_D_real_{S}').view(1)))
        elif self.seg_gen:
            pred = self._seg_pred(ctx, self._fake_seg.detach())
            E.loss_op(ctx, cs.kind, pred, None, cs.target(False), 0.5 * wD[M], getattr(self, f'loss_D_fake_{S}').view(1))
        real_mod, real_seg = self._pairs_real(ctx) if ((self.seg_gen and not paired_seg) or not real_mod_done) else (None, None)
        for i, n in enumerate(self.model_names_d):
            if real_mod_done:
                break
            with self._branch(i):
                pred = getattr(self, 'net' + n).run(ctx, real_mod[i])
                E.loss_op(ctx, cg.kind, pred, None, cg.target(True), 0.5 * wD[i], getattr(self, f'loss_D_real_{i + 1}').view(1))
        if self.seg_gen and not paired_seg:
            preds = [getattr(self, 'net' + n).run(ctx, real_seg[i]) for i, n in enumerate(self.model_names_ds)]
            pred = E.weighted_sum(ctx, preds, self.seg_weights[:M + 1])
            E.loss_op(ctx, cs.kind, pred, None, cs.target(True), 0.5 * wD[M], getattr(self, f'loss_D_real_{S}').view(1))
        tape.backward()
        self._join()

    def backward_G(self):
        """DeepLIIF_model.py:334-429 (VGG term excluded).  The seg term is weighted by loss_G_weights[modalities_no - 1]:
        the reference reuses the stale loop index (:418-421)."""
        tape = self._tape_G
        assert tape is not None, 'forward() must run in training mode before backward_G()'
        ctx = self._ctx(tape)
        M, S = self.opt.modalities_no, self.mod_id_seg
        wG = self.loss_G_weights
        cg, cs = self.criterionGAN_mod, self.criterionGAN_seg
        self._fork()                             # (D was updated and repacked, the G gradients zeroed, on the main stream)
        for i, n in enumerate(self.model_names_d):
            with self._branch(i):
                pair = E.concat_channels(ctx, [self._A, self._fake[i]])
                pred = getattr(self, 'net' + n).run(ctx, pair)
                E.loss_op(ctx, cg.kind, pred, None, cg.target(True), wG[i], getattr(self, f'loss_G_GAN_{i + 1}').view(1))
        if self.seg_gen:
            pred = self._seg_pred(ctx, self._fake_seg)
            E.loss_op(ctx, cs.kind, pred, None, cs.target(True), wG[M - 1], getattr(self, f'loss_G_GAN_{S}').view(1))
        for i in range(M):
            with self._branch(i):
                E.loss_op(ctx, L.LOSS_SMOOTH_L1, self._fake[i], self._B[i], 0.0, wG[i] * self.lambda_L1, self._l1_raw(i))
        if self.seg_gen:
            E.loss_op(ctx, L.LOSS_SMOOTH_L1, self._fake_seg, self._Bseg, 0.0, wG[M - 1] * self.lambda_L1, self._l1_raw(M))
        if self.criterionVGG is not None:
            # DeepLIIF_model.py:406-421: loss_G_VGG_i * lambda_feat joins the modality terms; the seg image's VGG value is computed there
            # too (:408-409) but never added to loss_G (:418-421), so it is not evaluated here
            for i in range(M):
                self.criterionVGG.run(ctx, self._fake[i], self._B[i], wG[i] * self.lambda_feat, self._vgg_buf[i:i + 1])
        self._extra_g_terms(ctx)
        tape.backward()
        self._join()
        self._tape_G = None
        # the reference logs loss_G_L1 already multiplied by lambda_L1 (:398-400)
        self._loss_buf[self._l1_slots] *= self.lambda_L1
        if self.criterionVGG is not None:
            self._vgg_buf *= self.lambda_feat
            for i in range(M):
                setattr(self, f'loss_G_VGG_{i + 1}', self._vgg_buf[i])

    def _extra_g_terms(self, ctx):
        """further terms of loss_G on the generator tape, before its backward (DeepLIIFKD: the distillation terms)"""

    def _l1_raw(self, i):
        M, S = self.opt.modalities_no, self.mod_id_seg
        name = f'G_L1_{i + 1}' if i < M else f'G_L1_{S}'
        return self._loss_buf[self._loss_index[name]].view(1)

    @property
    def _l1_slots(self):
        if not hasattr(self, '_l1_slots_cache'):
            idx = [self._loss_index[n] for n in self.loss_names if n.startswith('G_L1_')]
            self._l1_slots_cache = torch.tensor(idx, dtype=torch.long, device=self.device)
        return self._l1_slots_cache

    # ---------------------------------------------------------------------------------------------------------
    def _d_nets(self):
        return [getattr(self, 'net' + n) for n in self.model_names_d + self.model_names_ds]

    def optimize_parameters(self):
        """DeepLIIF_model.py:431-467."""
        self._sync_replicas()
        self.forward()
        self.set_requires_grad(self._d_nets(), True)
        self.optimizer_D.zero_grad()
        self.exchange.begin(self.optimizer_D)
        self.backward_D()
        self.exchange.finish(self.optimizer_D)       # cannot hide: D must be updated before backward_G runs it (DeepLIIF_model.py:431-467)
        self.optimizer_D.step()
        self.set_requires_grad(self._d_nets(), False)
        self.optimizer_G.zero_grad()
        self.exchange.begin(self.optimizer_G)
        self.backward_G()                            # each generator's slice goes on the wire as soon as its backward is done
        self.exchange.finish(self.optimizer_G)
        self.optimizer_G.step()

    def calculate_losses(self):
        """DeepLIIF_model.py:469-507: losses + gradients without the optimizer steps (validation)."""
        self.forward()
        self.set_requires_grad(self._d_nets(), True)
        self.optimizer_D.zero_grad()
        self.backward_D()
        self.set_requires_grad(self._d_nets(), False)
        self.optimizer_G.zero_grad()
        self.backward_G()


class DeepLIIFExtModel(BaseModel):
    """deepliif/models/DeepLIIFExt_model.py: M translation generators G_i(real_A); M seg generators
    GS_i(cat(real_A, fake_B[0], fake_B[i])) (9 channels, :173); discriminators D_i(cat(A, B_i)) and
    DS_i(cat(real_A, real_B[0], real_B[i], seg_i)) (12 channels, :97,186).  Networks and losses are list-valued
    (netG[i], loss_G_GAN[i], ...), model names 'G_1', 'GS_1', ...  Quirks kept: the generator-side seg GAN loss uses
    criterionGAN_mod (:236), seg loss weights are 1/M (:27,30), no VGG term."""
    graphable = True          # may StepGraph capture optimize_parameters()? (no host-side randomness / per-step host state)

    _extra_g_loss_names = ()         # subclasses add per-modality generator loss names the reference reports (SDG: G_VGG)

    def __init__(self, opt):
        super().__init__(opt)
        M = self.mod_gen_no = opt.modalities_no
        self.seg_weights = opt.seg_weights
        self.loss_G_weights, self.loss_D_weights = list(opt.loss_G_weights), list(opt.loss_D_weights)
        cin = self._input_channels(opt)
        self.loss_GS_weights = [1 / M] * M
        self.loss_DS_weights = [1 / M] * M
        seg = opt.seg_gen
        self.loss_names, self.visual_names = [], ['real_A']
        for i in range(1, M + 1):
            self.loss_names += [f'G_GAN_{i}', f'G_L1_{i}'] + [f'{x}_{i}' for x in self._extra_g_loss_names] + [f'D_real_{i}', f'D_fake_{i}']
            self.visual_names += [f'fake_B_{i}', f'real_B_{i}']
        if seg:
            for i in range(1, M + 1):
                self.loss_names += [f'GS_GAN_{i}', f'GS_L1_{i}', f'DS_real_{i}', f'DS_fake_{i}']
                self.visual_names += [f'fake_BS_{i}', f'real_BS_{i}']
        self.model_names = []
        for i in range(1, M + 1):
            self.model_names += [f'G_{i}'] + ([f'D_{i}'] if self.is_train else [])
        if seg:
            for i in range(1, M + 1):
                self.model_names += [f'GS_{i}'] + ([f'DS_{i}'] if self.is_train else [])
        net_g = opt.net_g if isinstance(opt.net_g, (list, tuple)) else [opt.net_g] * M
        net_gs = opt.net_gs if isinstance(opt.net_gs, (list, tuple)) else [opt.net_gs] * M
        use_dropout = not _get(opt, 'no_dropout', True)
        gpu = self._net_gpu_ids()
        self.netG = [networks.define_G(cin, opt.output_nc, opt.ngf, net_g[i], opt.norm, use_dropout, opt.init_type, opt.init_gain, gpu,
                                       opt.padding) for i in range(M)]
        self.netGS = [networks.define_G(opt.input_nc * 3, opt.output_nc, opt.ngf, net_gs[i], opt.norm, use_dropout, opt.init_type,
                                        opt.init_gain, gpu) if seg else None for i in range(M)]
        self.netD, self.netDS = [], []
        if self.is_train:
            nl = _get(opt, 'n_layers_D', 4)
            self.netD = [networks.define_D(cin + opt.output_nc, opt.ndf, _get(opt, 'net_d', 'n_layers'), nl, opt.norm, opt.init_type,
                                           opt.init_gain, gpu) for _ in range(M)]
            self.netDS = [networks.define_D(opt.input_nc * 3 + opt.output_nc, opt.ndf, _get(opt, 'net_ds', 'n_layers'), nl, opt.norm,
                                            opt.init_type, opt.init_gain, gpu) if seg else None for _ in range(M)]
        for _, net in self._nets():
            net.set_precision(self.precision.name)
        self._loss_index = {n: i for i, n in enumerate(self.loss_names)}
        self._loss_buf = torch.zeros(max(len(self.loss_names), 1), dtype=torch.float32, device=self.device)
        for fam in ('G_GAN', 'G_L1', 'D_real', 'D_fake', 'GS_GAN', 'GS_L1', 'DS_real', 'DS_fake'):
            setattr(self, 'loss_' + fam, [self._loss_buf[self._loss_index[f'{fam}_{i}']] for i in range(1, M + 1) if f'{fam}_{i}' in self._loss_index])
        if self.is_train:
            self.criterionGAN_mod = networks.GANLoss(opt.gan_mode).to(self.device)
            self.criterionGAN_seg = networks.GANLoss(opt.gan_mode_s).to(self.device)
            self.lambda_L1 = _get(opt, 'lambda_L1', 100.0)
            params_g = [p for net in self.netG + [n for n in self.netGS if n is not None] for p in net.parameters()]
            params_d = [p for net in self.netD + [n for n in self.netDS if n is not None] for p in net.parameters()]
            OptCls = networks.get_optimizer(_get(opt, 'optimizer', 'adam'))
            try:
                self.optimizer_G = OptCls(params_g, lr=opt.lr_g, betas=(opt.beta1, 0.999))
                self.optimizer_D = OptCls(params_d, lr=opt.lr_d, betas=(opt.beta1, 0.999))
            except TypeError:
                self.optimizer_G = OptCls(params_g, lr=opt.lr_g)
                self.optimizer_D = OptCls(params_d, lr=opt.lr_d)
            self.optimizers += [self.optimizer_G, self.optimizer_D]
            self.exchange = GradExchanger()
        self._tape_G = None
        # branch streams (round 5): chain i = G_i -> [cat(A, fake_1, fake_i) on the main stream] -> GS_i, with D_i / DS_i, runs on stream i % DL_STREAMS.
        # The only tensor two chains share is fake_1 (every seg generator's input, DeepLIIFExt_model.py:173): the concatenations -- and in backward the
        # accumulation of its gradient -- stay on the main stream between a join and a fork.  SDG's VGG term keeps everything on one stream (branch_parallel).

    @property
    def branch_parallel(self):
        return _EXT_STREAMS and getattr(self, 'criterionVGG', None) is None

    def _input_channels(self, opt):
        return opt.input_nc

    def _slot(self, name):
        return self._loss_buf[self._loss_index[name]].view(1)

    def set_input(self, input):
        """DeepLIIFExt_model.py:134-158: dict{'A', 'B': list, 'BS': list, 'A_paths'}  (SDG: 'A' is a list of input modalities,
        concatenated on the channel axis, SDG_model.py:108-110)."""
        p = self.precision
        A = input['A']
        self.real_A = torch.cat([a.to(self.device) for a in A], dim=1) if isinstance(A, (list, tuple)) else A.to(self.device)
        self.real_B = [b.to(self.device) for b in input['B']]
        self.real_BS = [b.to(self.device) for b in input.get('BS', [])]
        self.image_paths = input.get('A_paths', [])
        self._A = E.to_engine(self.real_A, p)
        self._B = [E.to_engine(b, p) for b in self.real_B]
        self._BS = [E.to_engine(b, p) for b in self.real_BS]
        self._real_cat = None

    def forward(self, record=None):
        record = self.is_train if record is None else record
        tape = self._new_tape() if record else None
        ctx = E.Ctx(self.precision, self._hook_tape(tape), training=record)
        self._fake, self.fake_B = [], []
        self._fork()
        for i, net in enumerate(self.netG):
            with self._branch(i):
                self._mark_net(tape, net)
                self._fake.append(net.run(ctx, self._A))
                self.fake_B.append(E.from_engine(self._fake[-1]))
        self._fake_s, self.fake_BS = [], []
        if any(n is not None for n in self.netGS):
            self._join(tape)                         # fake_1 feeds every seg generator: concatenate on the main stream ...
            cats = [E.concat_channels(ctx, [self._A, self._fake[0], self._fake[i]]) if net is not None else None for i, net in enumerate(self.netGS)]
            self._fork(tape)                         # ... and let chain i continue on its stream
            for i, net in enumerate(self.netGS):
                if net is not None:
                    with self._branch(i):
                        self._mark_net(tape, net)
                        self._fake_s.append(net.run(ctx, cats[i]))
                        self.fake_BS.append(E.from_engine(self._fake_s[-1]))
        if not record:
            self._join()
        for i, t in enumerate(self.fake_B):
            setattr(self, f'fake_B_{i + 1}', t)
        for i, t in enumerate(self.fake_BS):
            setattr(self, f'fake_BS_{i + 1}', t)
        self._tape_G = tape

    def _cat_real(self, ctx):
        if self._real_cat is None:
            self._real_cat = [E.concat_channels(ctx, [self._A, self._B[0], self._B[i]]) for i in range(self.mod_gen_no)]
        return self._real_cat

    def backward_D(self):
        tape = self._new_tape()
        ctx = E.Ctx(self.precision, self._hook_tape(tape), training=True)
        cg, cs, M = self.criterionGAN_mod, self.criterionGAN_seg, self.mod_gen_no
        rc = self._cat_real(ctx)                     # (main stream, before the fork)
        self._fork()                                 # (the gradients were zeroed on the main stream)
        if _D_PAIR_BATCH and all(getattr(d, 'norm_kind', 'batch') != 'batch' for d in list(self.netD[:M]) + [d for d in self.netDS if d is not None]):
            # per-sample-normalised discriminators: the fake and the real inputs of each as ONE batch of 2N (see DeepLIIFModel.backward_D)
            def both_halves(parts_fake, parts_real):
                nb, h, w, _ = parts_fake[0].t.shape
                c = sum(p.C for p in parts_fake)
                both = torch.zeros((2 * nb, h, w, E.cpad(c)), dtype=parts_fake[0].t.dtype, device=parts_fake[0].t.device)
                E.concat_channels(ctx, parts_fake, out=both[:nb])
                E.concat_channels(ctx, parts_real, out=both[nb:])
                return E.Act(both, c)
            for i in range(M):
                with self._branch(i):
                    self._mark_net(tape, self.netD[i])
                    pred = self.netD[i].run(ctx, both_halves([self._A, self._fake[i].detach()], [self._A, self._B[i]]))
                    E.loss_op_halves(ctx, cg.kind, pred, (cg.target(False), cg.target(True)), 0.5 * self.loss_D_weights[i],
                                     (self._slot(f'D_fake_{i + 1}'), self._slot(f'D_real_{i + 1}')))
            for i in range(len(self._fake_s)):
                with self._branch(i):
                    self._mark_net(tape, self.netDS[i])
                    pred = self.netDS[i].run(ctx, both_halves([rc[i], self._fake_s[i].detach()], [rc[i], self._BS[i]]))
                    E.loss_op_halves(ctx, cs.kind, pred, (cs.target(False), cs.target(True)), 0.5 * self.loss_DS_weights[i],
                                     (self._slot(f'DS_fake_{i + 1}'), self._slot(f'DS_real_{i + 1}')))
            tape.backward()
            self._join()
            return
        for i in range(M):
            with self._branch(i):
                # every discriminator runs twice below (fake, real): mark before the first use, on the stream of its branch (the marker's all-reduce is
                # ordered behind THAT stream)
                self._mark_net(tape, self.netD[i])
                pred = self.netD[i].run(ctx, E.concat_channels(ctx, [self._A, self._fake[i].detach()]))
                E.loss_op(ctx, cg.kind, pred, None, cg.target(False), 0.5 * self.loss_D_weights[i], self._slot(f'D_fake_{i + 1}'))
        for i in range(len(self._fake_s)):
            with self._branch(i):
                self._mark_net(tape, self.netDS[i])
                pred = self.netDS[i].run(ctx, E.concat_channels(ctx, [rc[i], self._fake_s[i].detach()]))
                E.loss_op(ctx, cs.kind, pred, None, cs.target(False), 0.5 * self.loss_DS_weights[i], self._slot(f'DS_fake_{i + 1}'))
        for i in range(M):
            with self._branch(i):
                pred = self.netD[i].run(ctx, E.concat_channels(ctx, [self._A, self._B[i]]))
                E.loss_op(ctx, cg.kind, pred, None, cg.target(True), 0.5 * self.loss_D_weights[i], self._slot(f'D_real_{i + 1}'))
        for i in range(len(self._fake_s)):
            with self._branch(i):
                pred = self.netDS[i].run(ctx, E.concat_channels(ctx, [rc[i], self._BS[i]]))
                E.loss_op(ctx, cs.kind, pred, None, cs.target(True), 0.5 * self.loss_DS_weights[i], self._slot(f'DS_real_{i + 1}'))
        tape.backward()
        self._join()

    def backward_G(self):
        tape = self._tape_G
        ctx = E.Ctx(self.precision, self._hook_tape(tape), training=True)
        cg, M = self.criterionGAN_mod, self.mod_gen_no
        rc = self._cat_real(E.Ctx(self.precision, None, training=True))
        self._fork()                                 # (D was updated and repacked, the G gradients zeroed, on the main stream)
        for i in range(M):
            with self._branch(i):
                pred = self.netD[i].run(ctx, E.concat_channels(ctx, [self._A, self._fake[i]]))
                E.loss_op(ctx, cg.kind, pred, None, cg.target(True), self.loss_G_weights[i], self._slot(f'G_GAN_{i + 1}'))
        for i in range(len(self._fake_s)):
            with self._branch(i):
                pred = self.netDS[i].run(ctx, E.concat_channels(ctx, [rc[i], self._fake_s[i]]))
                E.loss_op(ctx, cg.kind, pred, None, cg.target(True), self.loss_GS_weights[i], self._slot(f'GS_GAN_{i + 1}'))   # criterionGAN_mod (:236)
        for i in range(M):
            with self._branch(i):
                E.loss_op(ctx, L.LOSS_SMOOTH_L1, self._fake[i], self._B[i], 0.0, self.loss_G_weights[i] * self.lambda_L1, self._slot(f'G_L1_{i + 1}'))
        for i in range(len(self._fake_s)):
            with self._branch(i):
                E.loss_op(ctx, L.LOSS_SMOOTH_L1, self._fake_s[i], self._BS[i], 0.0, self.loss_GS_weights[i] * self.lambda_L1, self._slot(f'GS_L1_{i + 1}'))
        vgg = getattr(self, 'criterionVGG', None)
        if vgg is not None:                           # SDG_model.py:176-184 (DeepLIIFExt has the term commented out, DeepLIIFExt_model.py:257-265)
            for i in range(M):
                vgg.run(ctx, self._fake[i], self._B[i], self.loss_G_weights[i] * self.lambda_feat, self._slot(f'G_VGG_{i + 1}'))
        tape.backward()
        self._join()
        self._tape_G = None
        # index tensors built ONCE: torch.tensor(..., device=cuda) is a host-to-device copy per step (and not capturable, models.StepGraph)
        if not hasattr(self, '_idx_cache'):
            self._idx_cache = {k: torch.tensor([self._loss_index[n] for n in self.loss_names if k in n], dtype=torch.long, device=self.device)
                               for k in ('_L1_', '_VGG_')}
        self._loss_buf[self._idx_cache['_L1_']] *= self.lambda_L1
        if self._idx_cache['_VGG_'].numel():
            # reported like the reference does (value * lambda_feat); NaN, not a plausible-looking 0.0, when the term was not evaluated
            sel = self._idx_cache['_VGG_']
            if vgg is not None:
                self._loss_buf[sel] *= self.lambda_feat
            else:
                self._loss_buf[sel] = float('nan')

    def _d_nets(self):
        return [n for n in self.netD + self.netDS if n is not None]

    def optimize_parameters(self):
        self._sync_replicas()
        self.forward()
        self.set_requires_grad(self._d_nets(), True)
        self.optimizer_D.zero_grad()
        self.exchange.begin(self.optimizer_D)
        self.backward_D()
        self.exchange.finish(self.optimizer_D)       # cannot hide: D must be updated before backward_G runs it (DeepLIIF_model.py:431-467)
        self.optimizer_D.step()
        self.set_requires_grad(self._d_nets(), False)
        self.optimizer_G.zero_grad()
        self.exchange.begin(self.optimizer_G)
        self.backward_G()                            # each generator's slice goes on the wire as soon as its backward is done
        self.exchange.finish(self.optimizer_G)
        self.optimizer_G.step()

    def calculate_losses(self):
        self.forward()
        self.set_requires_grad(self._d_nets(), True)
        self.optimizer_D.zero_grad()
        self.backward_D()
        self.set_requires_grad(self._d_nets(), False)
        self.optimizer_G.zero_grad()
        self.backward_G()


class SDGModel(DeepLIIFExtModel):
    """deepliif/models/SDG_model.py: DeepLIIFExt's translation branch only, with `input_no` input modalities concatenated on the
    channel axis (generators take input_nc*input_no channels, discriminators input_nc*input_no + output_nc).  The reference adds
    a VGG19 perceptual term (SDG_model.py:176-184); like for DeepLIIF it is outside the MI355X hot path."""

    # SDG_model.py:33 lists a VGG19 term per modality; it is not computed on this path and is reported as 0.0 so that
    # loss_names / get_current_losses() have the reference's keys in the reference's order
    _extra_g_loss_names = ('G_VGG',)

    def __init__(self, opt):
        opt.seg_gen = False
        super().__init__(opt)
        if self.is_train:
            self.lambda_feat, self.criterionVGG = self._make_vgg(opt)

    def _input_channels(self, opt):
        return opt.input_nc * _get(opt, 'input_no', 1)


def map_model_names(model_names, mod_id_seg_source, input_id_source, mod_id_seg_target, input_id_target):
    """deepliif/util/util.py:273-292: teacher network / image names -> the student's naming when the seg id or the input id differ."""
    res = {}
    for name in model_names:
        new = name
        if len(name) > 2 and name[1] == str(mod_id_seg_source):
            new = name[0] + str(mod_id_seg_target) + name[2:]
            if str(input_id_source) != str(input_id_target):
                new = new[:2] + str(int(new[2:]) + (-1 if int(input_id_target) == 0 else 1))
        res[name] = new
    res['G' + str(mod_id_seg_source)] = 'G' + str(mod_id_seg_target)        # the aggregated seg image is not a network name
    return res


class DeepLIIFKDModel(DeepLIIFModel):
    """deepliif/models/DeepLIIFKD_model.py: DeepLIIF (always with the seg branch; vanilla GAN loss for the modalities, lsgan for seg, :130-132)
    distilled from a frozen TEACHER model directory (opt.model_dir_teacher, loaded through init_nets(eager_mode=True), :107-118): every student
    image -- fake_B_i, each seg generator's output fake_B_S_i and the aggregated fake_B_S -- adds
        10 * KLDivLoss(batchmean)(LogSoftmax(student.view(1,1,-1)), Softmax(teacher.view(1,1,-1)))        (:313-349)
    to loss_G: ONE softmax over all N*3*H*W values of an image batch (engine.kldiv_op -> dl_kldiv).  The teacher runs in inference mode on the
    training batch (run_dask(img=real_A, use_dask=False, output_tensor=True), :203) with its default equal seg weights; BatchNorm there
    normalises with the statistics of the WHOLE batch (the reference's disable_batchnorm_tracking_stats + a batched call), not per sample.
    Quirks kept: loss_names lists G_KLDiv_S{M} twice and never G_KLDiv_S0 (:36-41) although the latter enters loss_G (:342-343)."""
    graphable = False          # may StepGraph capture optimize_parameters()? (no host-side randomness / per-step host state)

    FACTOR_KLDIV = 10.0

    def __init__(self, opt):
        # The caller's option object is left as it came: the base class is built from a shallow copy that carries what
        # DeepLIIFKD_model.py hard-wires -- seg generators built unconditionally (:55-60, 95-98), criterionGAN_BCE / criterionGAN_lsgan (:130-131)
        import copy
        opt = copy.copy(opt)
        opt.seg_gen = True
        if getattr(opt, 'netG', None) is None and hasattr(opt, 'net_g'):
            opt.netG = opt.net_g
        opt.gan_mode, opt.gan_mode_s = 'vanilla', 'lsgan'
        super().__init__(opt)
        M, S = opt.modalities_no, self.mod_id_seg
        # ---- names (:33-45): per modality ..., G_KLDiv_i, G_KLDiv_S{i}; then the seg names, G_KLDiv_S and (again) G_KLDiv_S{M}
        self.loss_names = []
        self.visual_names = ['real_A']
        for i in range(1, M + 1):
            self.loss_names += [f'G_GAN_{i}', f'G_L1_{i}', f'D_real_{i}', f'D_fake_{i}', f'G_KLDiv_{i}', f'G_KLDiv_{S}{i}']
            self.visual_names += [f'fake_B_{i}', f'fake_B_{i}_teacher', f'real_B_{i}']
        self.loss_names += [f'G_GAN_{S}', f'G_L1_{S}', f'D_real_{S}', f'D_fake_{S}', f'G_KLDiv_{S}', f'G_KLDiv_{S}{M}']
        for i in range(M + 1):
            self.visual_names += [f'fake_B_{S}{i}', f'fake_B_{S}{i}_teacher']
        self.visual_names += [f'fake_B_{S}', f'fake_B_{S}_teacher', f'real_B_{S}']
        slots = list(OrderedDict.fromkeys(self.loss_names + [f'G_KLDiv_{S}0']))      # + the term that is computed but never listed
        self._loss_index = {n: i for i, n in enumerate(slots)}
        self._loss_buf = torch.zeros(len(slots), dtype=torch.float32, device=self.device)
        for n, i in self._loss_index.items():
            setattr(self, 'loss_' + n, self._loss_buf[i])
        if hasattr(self, '_l1_slots_cache'):
            del self._l1_slots_cache
        if not self.is_train:
            self.visual_names = [n for n in self.visual_names if not n.endswith('_teacher')]
            return
        if self.input_id != '0':
            # the reference's own loss_G reads loss_G_KLDiv_{S}{M+1}, which backward_G never computes (:323-324, 344-345): only input_id '0' trains
            raise NotImplementedError("DeepLIIFKD trains with input_id '0' only (the reference's backward_G fails for other ids)")
        from . import inference as I
        tdir = opt.model_dir_teacher
        self.opt_teacher = I.get_opt(tdir, mode='test')
        self.opt_teacher.gpu_ids = opt.gpu_ids                       # use the student's device (:110)
        self.opt_teacher.precision = self.precision.name             # same storage type: the distillation kernel reads both tensors
        if self.precision.name != 'fp32':
            print(f'deepliif_amd: the DeepLIIFKD teacher runs on the student\'s precision policy ({self.precision.name}); the reference runs it in fp32 '
                  '(set opt.precision = "fp32" for a strict-policy teacher and student)')
        self.nets_teacher = I.init_nets(tdir, eager_mode=True, opt=self.opt_teacher, phase='test')
        t_seg, t_in = _get(self.opt_teacher, 'mod_id_seg', 'S'), str(_get(self.opt_teacher, 'input_id', '0'))
        self.opt_teacher.mod_id_seg, self.opt_teacher.input_id = t_seg, t_in
        self.d_mapping_model_name = map_model_names(list(self.nets_teacher.keys()), t_seg, t_in, S, self.input_id)
        print('Model name mapping, teacher model to student model:', self.d_mapping_model_name)

    def _teacher_attr(self, key):
        """:205-214: teacher result key -> 'fake_B_<suffix>_teacher' (the characters of the mapped name joined by '_': 'GS1' -> 'S_1')"""
        suffix = list(self.d_mapping_model_name[key][1:])
        if suffix[0] == str(self.opt_teacher.mod_id_seg) and suffix[0] != str(self.mod_id_seg):
            suffix[0] = str(self.mod_id_seg)
        return 'fake_B_' + '_'.join(suffix) + '_teacher'

    def forward(self, record: Optional[bool] = None):
        super().forward(record)
        if self.is_train:
            from . import inference as I
            res = I.run_generators_engine(self._A, self.nets_teacher, self.opt_teacher, per_sample_norm=False)
            self._teacher = {}
            for k, v in res.items():
                name = self._teacher_attr(k)
                self._teacher[name] = v
                setattr(self, name, E.from_engine(v))

    def _extra_g_terms(self, ctx):
        M, S, f = self.opt.modalities_no, self.mod_id_seg, self.FACTOR_KLDIV
        slot = lambda n: self._loss_buf[self._loss_index[n]].view(1)
        for i in range(M):
            E.kldiv_op(ctx, self._fake[i], self._teacher[f'fake_B_{i + 1}_teacher'], f, slot(f'G_KLDiv_{i + 1}'))
        E.kldiv_op(ctx, self._fake_seg, self._teacher[f'fake_B_{S}_teacher'], f, slot(f'G_KLDiv_{S}'))
        for i in range(M + 1):                       # i = 0 (base input) enters loss_G through the input_id == '0' branch (:342-343)
            E.kldiv_op(ctx, self._fake_seg_parts[i], self._teacher[f'fake_B_{S}_{i}_teacher'], f, slot(f'G_KLDiv_{S}{i}'))


class ImagePool:
    """deepliif/util/image_pool.py: history of generated images for the CycleGAN discriminators.  Same decisions from the same `random`
    stream (one uniform per image once the pool is full, one randint when it swaps), on per-sample engine tensors."""

    def __init__(self, pool_size):
        self.pool_size = pool_size
        self.num_imgs = 0
        self.images: List[torch.Tensor] = []

    def query(self, a: E.Act) -> E.Act:
        import random
        if self.pool_size == 0:
            return a.detach()
        out = []
        for j in range(a.t.shape[0]):
            img = a.t[j:j + 1]
            if self.num_imgs < self.pool_size:
                self.num_imgs += 1
                self.images.append(img.clone())
                out.append(img)
            elif random.uniform(0, 1) > 0.5:
                k = random.randint(0, self.pool_size - 1)
                out.append(self.images[k])
                self.images[k] = img.clone()
            else:
                out.append(img)
        return E.Act(torch.cat(out, 0), a.C, False)


class CycleGANModel(BaseModel):
    """deepliif/models/CycleGAN_model.py: per modality i two generators GA_i: A -> B_i, GB_i: B_i -> A and two UNCONDITIONAL discriminators
    DA_i (on B_i images), DB_i (on A images).  One step (:266-282): forward (fake_B = GA(A), rec_A = GB(fake_B), fake_A = GB(B), rec_B = GA(fake_A)),
    the GENERATOR update first -- loss_G = sum_i w_i [GAN(DA_i(fake_B_i)) + VGG(fake_B_i, B_i) + GAN(DB_i(fake_A_i)) + VGG(fake_A_i, A)]
    + 10/M sum_i [L1(rec_A_i, A) + L1(rec_B_i, B_i)], identity terms off (:207-208, 213) -- then the discriminators on the pooled, detached fakes,
    one backward per discriminator of (real + fake) * 0.5 * w_i (:172-205).  Every generator runs TWICE on the generator tape.
    The VGG term carries no lambda here (:232, 238): it needs the weight file like everywhere else (BaseModel._make_vgg rules)."""
    graphable = False          # may StepGraph capture optimize_parameters()? (no host-side randomness / per-step host state)

    LAMBDA_A = LAMBDA_B = 10.0

    def __init__(self, opt):
        super().__init__(opt)
        M = self.mod_gen_no = opt.modalities_no
        if not hasattr(opt, 'upsample'):
            opt.upsample = 'convtranspose'
        if not hasattr(opt, 'label_smoothing'):
            opt.label_smoothing = 0
        self.loss_G_weights, self.loss_D_weights = list(opt.loss_G_weights), list(opt.loss_D_weights)
        self.loss_cyc_weights = [1 / M] * M
        opt.lambda_identity = 0
        self.loss_names = ['D_A', 'G_A', 'cycle_A', 'idt_A', 'D_B', 'G_B', 'cycle_B', 'idt_B']
        suf = range(1, M + 1)
        self.visual_names = [f'real_As_{i}' for i in suf] + [f'fake_Bs_{i}' for i in suf] + [f'rec_As_{i}' for i in suf] + \
                            [f'real_Bs_{i}' for i in suf] + [f'fake_As_{i}' for i in suf] + [f'rec_Bs_{i}' for i in suf]
        btoa = _get(opt, 'BtoA', False)
        if self.is_train:
            self.model_names = [f'GA_{i}' for i in suf] + [f'GB_{i}' for i in suf] + [f'DA_{i}' for i in suf] + [f'DB_{i}' for i in suf]
        else:
            self.model_names = [f'GB_{i}' for i in suf] if btoa else [f'GA_{i}' for i in suf]
        net_g = opt.net_g if isinstance(opt.net_g, (list, tuple)) else [opt.net_g] * M
        opt.net_g = list(net_g)
        use_dropout = not _get(opt, 'no_dropout', True)
        gpu = self._net_gpu_ids()
        self.netGA, self.netGB, self.netDA, self.netDB = [], [], [], []
        for i in range(M):                       # construction order = the reference's (:77-85), so a seeded init draws the same weights
            if self.is_train or not btoa:
                self.netGA.append(networks.define_G(opt.input_nc, opt.output_nc, opt.ngf, net_g[i], opt.norm, use_dropout, opt.init_type, opt.init_gain,
                                                    gpu, opt.padding, opt.upsample))
            if self.is_train or btoa:
                self.netGB.append(networks.define_G(opt.output_nc, opt.input_nc, opt.ngf, net_g[i], opt.norm, use_dropout, opt.init_type, opt.init_gain,
                                                    gpu, opt.padding, opt.upsample))
        if self.is_train:
            nl, nd = _get(opt, 'n_layers_D', 4), _get(opt, 'net_d', 'n_layers')
            for i in range(M):
                self.netDA.append(networks.define_D(opt.output_nc, opt.ndf, nd, nl, opt.norm, opt.init_type, opt.init_gain, gpu))
                self.netDB.append(networks.define_D(opt.input_nc, opt.ndf, nd, nl, opt.norm, opt.init_type, opt.init_gain, gpu))
        for _, net in self._nets():
            net.set_precision(self.precision.name)
        self._loss_index = {n: i for i, n in enumerate(self.loss_names)}
        self._loss_buf = torch.zeros(len(self.loss_names), dtype=torch.float32, device=self.device)
        for n, i in self._loss_index.items():
            setattr(self, 'loss_' + n, self._loss_buf[i])
        if self.is_train:
            self.fake_A_pools = [ImagePool(_get(opt, 'pool_size', 50)) for _ in range(M)]
            self.fake_B_pools = [ImagePool(_get(opt, 'pool_size', 50)) for _ in range(M)]
            self.criterionGAN = networks.GANLoss(_get(opt, 'gan_mode', 'lsgan'), label_smoothing=opt.label_smoothing).to(self.device)
            self.criterionVGG = self._make_cycle_vgg(opt)
            params_g = [p for net in self.netGA + self.netGB for p in net.parameters()]
            params_d = [p for net in self.netDA + self.netDB for p in net.parameters()]
            OptCls = networks.get_optimizer(_get(opt, 'optimizer', 'adam'))
            try:
                self.optimizer_G = OptCls(params_g, lr=opt.lr_g, betas=(opt.beta1, 0.999))
                self.optimizer_D = OptCls(params_d, lr=opt.lr_d, betas=(opt.beta1, 0.999))
            except TypeError:
                self.optimizer_G = OptCls(params_g, lr=opt.lr_g)
                self.optimizer_D = OptCls(params_d, lr=opt.lr_d)
            self.optimizers += [self.optimizer_G, self.optimizer_D]
            self.exchange = GradExchanger()
        self._tape_G = None

    def _make_cycle_vgg(self, opt):
        path = networks.vgg_weights_path(opt)
        if path:
            return networks.VGGLoss(path, self.device, self.precision.name)
        if _get(opt, 'allow_no_vgg', False) or os.environ.get('DEEPLIIF_AMD_ALLOW_NO_VGG') == '1':
            print('deepliif_amd: CycleGAN without a VGG19 weight file: training WITHOUT the perceptual term (explicit opt-out)')
            return None
        raise RuntimeError('CycleGAN_model.py:232,238 adds the VGG19 perceptual loss to both generator terms, which needs torchvision\'s pretrained vgg19 '
                           'weights; pass them as a file (opt.vgg_weights or DEEPLIIF_VGG19_WEIGHTS=/path/vgg19.pth) or opt out with opt.allow_no_vgg / '
                           'DEEPLIIF_AMD_ALLOW_NO_VGG=1')

    def set_input(self, input):
        """CycleGAN_model.py:148-159: dict{'A': Tensor, 'Bs': list[Tensor], 'A_paths'}"""
        M, p = self.mod_gen_no, self.precision
        self.real_As = [input['A'].to(self.device) for _ in range(M)]
        self.real_Bs = [x.to(self.device) for x in input['Bs']]
        self.image_paths = input.get('A_paths', [])
        self._A = E.to_engine(self.real_As[0], p)
        self._Bs = [E.to_engine(b, p) for b in self.real_Bs]
        for i in range(M):
            setattr(self, f'real_As_{i + 1}', self.real_As[i])
            setattr(self, f'real_Bs_{i + 1}', self.real_Bs[i])

    def forward(self, record: Optional[bool] = None):
        """CycleGAN_model.py:161-170; with only one direction loaded (test time) the other lists stay empty"""
        record = self.is_train if record is None else record
        tape = E.Tape() if record else None
        ctx = E.Ctx(self.precision, self._hook_tape(tape), training=record)
        for net in self.netGA + self.netGB:            # every generator runs twice below: mark before the first use
            self._mark_net(tape, net)
        self._fake_B = [net.run(ctx, self._A) for net in self.netGA]
        self._rec_A = [net.run(ctx, f) for net, f in zip(self.netGB, self._fake_B)]
        self._fake_A = [net.run(ctx, b) for net, b in zip(self.netGB, self._Bs)]
        self._rec_B = [net.run(ctx, f) for net, f in zip(self.netGA, self._fake_A)]
        for fam, acts in (('fake_Bs', self._fake_B), ('rec_As', self._rec_A), ('fake_As', self._fake_A), ('rec_Bs', self._rec_B)):
            ts = [E.from_engine(a) for a in acts]
            setattr(self, fam, ts)
            for i, t in enumerate(ts):
                setattr(self, f'{fam}_{i + 1}', t)
        self._tape_G = tape

    def _slot(self, name):
        return self._loss_buf[self._loss_index[name]].view(1)

    def backward_G(self):
        """CycleGAN_model.py:207-264"""
        tape = self._tape_G
        assert tape is not None, 'forward() must run in training mode before backward_G()'
        ctx = E.Ctx(self.precision, self._hook_tape(tape), training=True)
        cg, wG, cyc = self.criterionGAN, self.loss_G_weights, self.loss_cyc_weights
        for n in ('G_A', 'G_B', 'cycle_A', 'cycle_B'):           # the terms below ACCUMULATE into their slots (sums over the modalities)
            self._slot(n).zero_()
        for i in range(self.mod_gen_no):
            E.loss_op(ctx, cg.kind, self.netDA[i].run(ctx, self._fake_B[i]), None, cg.target(True), wG[i], self._slot('G_A'), wG[i], True)
            if self.criterionVGG is not None:
                self.criterionVGG.run(ctx, self._fake_B[i], self._Bs[i], wG[i], self._slot('G_A'), wG[i], True)
        for i in range(self.mod_gen_no):
            E.loss_op(ctx, cg.kind, self.netDB[i].run(ctx, self._fake_A[i]), None, cg.target(True), wG[i], self._slot('G_B'), wG[i], True)
            if self.criterionVGG is not None:
                self.criterionVGG.run(ctx, self._fake_A[i], self._A, wG[i], self._slot('G_B'), wG[i], True)
        for i in range(self.mod_gen_no):
            w = self.LAMBDA_A * cyc[i]
            E.loss_op(ctx, L.LOSS_L1, self._rec_A[i], self._A, 0.0, w, self._slot('cycle_A'), w, True)
        for i in range(self.mod_gen_no):
            w = self.LAMBDA_B * cyc[i]
            E.loss_op(ctx, L.LOSS_L1, self._rec_B[i], self._Bs[i], 0.0, w, self._slot('cycle_B'), w, True)
        tape.backward()
        self._tape_G = None

    def _backward_D_family(self, nets, reals, fakes, slot):
        """backward_D_basic per discriminator (:172-191): (GAN(D(real), True) + GAN(D(fake.detach()), False)) * 0.5 * loss_D_weights[i]"""
        cg = self.criterionGAN
        for i, (net, real, fake) in enumerate(zip(nets, reals, fakes)):
            tape = E.Tape()
            ctx = E.Ctx(self.precision, self._hook_tape(tape), training=True)
            self._mark_net(tape, net)
            w = 0.5 * self.loss_D_weights[i]
            E.loss_op(ctx, cg.kind, net.run(ctx, real), None, cg.target(True), w, slot, w, True)
            E.loss_op(ctx, cg.kind, net.run(ctx, fake), None, cg.target(False), w, slot, w, True)
            tape.backward()

    def backward_D_A(self):
        fakes = [pool.query(f) for pool, f in zip(self.fake_B_pools, self._fake_B)]
        self._slot('D_A').zero_()
        self._backward_D_family(self.netDA, self._Bs, fakes, self._slot('D_A'))

    def backward_D_B(self):
        fakes = [pool.query(f) for pool, f in zip(self.fake_A_pools, self._fake_A)]
        self._slot('D_B').zero_()
        self._backward_D_family(self.netDB, [self._A] * self.mod_gen_no, fakes, self._slot('D_B'))

    def _d_nets(self):
        return self.netDA + self.netDB

    def optimize_parameters(self):
        """CycleGAN_model.py:266-282: generators first, then both discriminator families"""
        self._sync_replicas()
        self.forward()
        self.set_requires_grad(self._d_nets(), False)
        self.optimizer_G.zero_grad()
        self.exchange.begin(self.optimizer_G)
        self.backward_G()
        self.exchange.finish(self.optimizer_G)
        self.optimizer_G.step()
        self.set_requires_grad(self._d_nets(), True)
        self.optimizer_D.zero_grad()
        self.exchange.begin(self.optimizer_D)
        self.backward_D_A()
        self.backward_D_B()
        self.exchange.finish(self.optimizer_D)
        self.optimizer_D.step()


_MODEL_CLASSES = {'DeepLIIF': DeepLIIFModel, 'DeepLIIFExt': DeepLIIFExtModel, 'SDG': SDGModel, 'DeepLIIFKD': DeepLIIFKDModel, 'CycleGAN': CycleGANModel}


def create_model(opt):
    """deepliif/models/__init__.py:101-114."""
    name = _get(opt, 'model', 'DeepLIIF')
    if name not in _MODEL_CLASSES:
        raise NotImplementedError(f'model [{name}] is not on the MI355X hot path (available: {sorted(_MODEL_CLASSES)})')
    instance = _MODEL_CLASSES[name](opt)
    print('model [%s] was created' % type(instance).__name__)
    return instance
