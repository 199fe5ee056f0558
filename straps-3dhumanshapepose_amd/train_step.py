"""One synthetic on-the-fly training step on the GPU -- the body of the reference's batch loop
(train/train_synthetic_otf_rendering.py:112-233), every stage a HIP kernel behind the C ABI:

  [no grad]  G1 augment_smpl (:121-126)  G2 augment_cam_t (:127-129)  SMPL#1 targets (:132-135)
             P2 perspective projection (:141-143)  SMPL#2 reposed targets (:144)
             part segmentation: the HIP rasteriser (nmr_renderer.NMRRenderer, :155) + on-device bbox crop / nearest resize (:161-170) -- SURVEY 8f f1/f2
             G3 proxy augmentation (:173-175)  G4+G5 network input (:178-182)
  forward    regressor (training-mode BatchNorm) -> rot6d -> SMPL#3 (:186-199), SMPL#4 reposed (:206)
  loss       heads (COCO orthographic projection, H36M-LSP joints, visibility) + 5 MSE tasks + gradients
  backward   SMPL -> rot6d -> IEF -> encoder, gradients written straight into ONE flat fp32 buffer
  exchange   (world > 1) one sum all-reduce of that buffer over RCCL/xGMI
  update     Adam over the flat parameter buffer (66|165 regressor tensors + 5 log-variances)

Random draws come from the device-resident Philox generator of `device_rng` (seed + rank, step counter on the device):
no torch operator does arithmetic inside the step (a few `clone()` / `zeros` of small tensors remain) -- torch provides memory, streams,
hipGraph capture and `torch.distributed`.
"""
import ctypes as C
import sys

import numpy as np
import torch

from . import config, hipabi
from .autograd_ops import encoder_backward, ief_backward
from .encoder_exec import encoder_forward
from .ief_module import EST_LD
from .multi_task_loss import TASKS

# augmentation parameter dictionaries of run_train.py:133-190 (the defaults of TrainStep)
SMPL_AUGMENT_PARAMS = {'augment_shape': True, 'delta_betas_distribution': 'normal', 'delta_betas_std_vector': [1.5] * 10,
                       'delta_betas_range': [-3., 3.]}
CAM_AUGMENT_PARAMS = {'xy_std': 0.05, 'delta_z_range': [-5, 5]}
BBOX_AUGMENT_PARAMS = {'crop_input': True, 'mean_scale_factor': 1.2, 'delta_scale_range': [-0.2, 0.2], 'delta_centre_range': [-5, 5]}
PROXY_REP_AUGMENT_PARAMS = {'remove_appendages': True, 'deviate_joints2D': True, 'deviate_verts2D': True, 'occlude_seg': True,
                            'remove_appendages_classes': [1, 2, 3, 4, 5, 6],
                            'remove_appendages_probabilities': [0.1, 0.1, 0.1, 0.1, 0.05, 0.05],
                            'delta_j2d_dev_range': [-8, 8], 'delta_j2d_hip_dev_range': [-8, 8], 'delta_verts2d_dev_range': [-0.01, 0.01],
                            'occlude_probability': 0.5, 'occlude_box_dim': 48}
H36M14 = [73 + i for i in config.H36M_TO_J14]


def flatten_parameters(params, device):
    """move every parameter into one contiguous fp32 buffer (views keep the modules working) and build
    the matching flat gradient buffer.  Returns (flat_p, flat_g, {param: grad_view})."""
    total = sum(p.numel() for p in params)
    flat_p = torch.empty(total, device=device, dtype=torch.float32)
    flat_g = torch.zeros(total, device=device, dtype=torch.float32)
    views, off = {}, 0
    for p in params:
        n = p.numel()
        flat_p[off:off + n].copy_(p.detach().reshape(-1))
        p.data = flat_p[off:off + n].view(p.shape)
        views[p] = flat_g[off:off + n].view(p.shape)
        off += n
    return flat_p, flat_g, views


def allreduce_gradients(flat_g, world_size, group=None):
    """the step's single exchange: sum all-reduce of the flat gradient buffer (RCCL over xGMI on GPUs,
    gloo in the CPU tests); the 1/world_size is folded into the Adam kernel's grad_scale."""
    if world_size > 1:
        import torch.distributed as dist
        dist.all_reduce(flat_g, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world_size


def allreduce_visible_count(count, world_size, group=None, async_op=False):
    """global masked mean of the joints2D task under data parallel: every rank contributes the number of visible target
    joints of ITS batch (a 1-element float tensor, straps_count_visible); after this sum all-reduce the tensor holds the job's count,
    which straps_loss_fwd_bwd_gm divides by (x 1 / world size).  Returns the work handle when async_op, else None."""
    if world_size > 1:
        import torch.distributed as dist
        return dist.all_reduce(count, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return None


# A/B switch (both forms write the same input and the same non-zero map): one pass (round 4) / straps_build_proxy_input + straps_stem_nzmask
_FUSED_PROXY_NZ = True

class GradientExchange:
    """The step's gradient exchange: a sum all-reduce of the flat fp32 gradient buffer (RCCL over xGMI on GPUs, gloo in the
    CPU tests), in two buckets so that it overlaps the backward pass.  `start_tail()` is called as soon as the tail of
    the buffer (layer3, layer4, IEF, loss weights: the bulk of the bytes, and the FIRST gradients backward finishes) is
    final and launches its all-reduce asynchronously; `finish()` waits for it, reduces the head (stem, layer1, layer2)
    and returns the 1/world_size the Adam kernel folds in.  split_off = 0 degenerates to one bucket.

    backend = 'torch': torch.distributed collectives on the process group (`nccl` == RCCL on ROCm; `gloo` in the CPU tests).
    backend = 'rccl' : the library's own C-ABI exchange (include/straps_hip.h: straps_comm_* / straps_allreduce_grads) -- a
                       communicator created through the ABI (its 128-byte id travels over the torch process group when world > 1)
                       and all-reduces enqueued on a dedicated HIP stream ordered against the step's stream with events: the
                       path a host without torch takes, exercised here by the same step.
    force: exchange even when world == 1 (an all-reduce over one rank is the identity): lets a single-GPU box run the real
           stream / split-graph choreography of the N > 1 step (tests/test_gpu_exchange.py)."""

    def __init__(self, flat_g, split_off, world_size, group=None, force=False, backend='torch', rank=0):
        self.flat_g, self.split_off, self.world, self.group = flat_g, int(split_off), world_size, group
        self.active = world_size > 1 or bool(force)
        self.backend = backend
        self._work = None
        self._comm, self._stream, self._started = None, None, False
        if backend not in ('torch', 'rccl'):
            raise ValueError("GradientExchange: backend must be 'torch' or 'rccl' (got %r)" % (backend,))
        if self.active and backend == 'rccl':
            self._init_rccl(rank)

    def _init_rccl(self, rank):
        """rank: the caller's rank in the JOB; the communicator is created over `group` (None = the default group), so its rank is the
        caller's rank INSIDE that group and the 128-byte id comes from the group's rank 0, whoever that is in the job."""
        if not self.flat_g.is_cuda:
            raise RuntimeError("GradientExchange(backend='rccl') needs the gradient buffer on a GPU")
        L = hipabi.lib()
        grank = rank
        if self.world > 1:
            import torch.distributed as dist
            grank = dist.get_rank(self.group)
            if grank < 0:
                raise RuntimeError("GradientExchange(backend='rccl'): this process is not a member of the group it was given")
            if dist.get_world_size(self.group) != self.world:
                raise RuntimeError("GradientExchange(backend='rccl'): world_size %d != size of the process group %d" % (self.world, dist.get_world_size(self.group)))
        idbuf = (C.c_char * 128)()
        if grank == 0:
            hipabi.check(L.straps_comm_unique_id(idbuf), 'straps_comm_unique_id')
        if self.world > 1:
            box = [bytes(idbuf) if grank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            idbuf = (C.c_char * 128).from_buffer_copy(box[0])
        comm = C.c_void_p()
        with torch.cuda.device(self.flat_g.device):
            hipabi.check(L.straps_comm_init_rank(idbuf, self.world, grank, C.byref(comm)), 'straps_comm_init_rank')
            self._stream = torch.cuda.Stream(device=self.flat_g.device)
        self._comm = comm
        if L.straps_comm_size(comm) != self.world:
            n = L.straps_comm_size(comm)
            self.close()
            raise RuntimeError("GradientExchange(backend='rccl'): the communicator reports %d ranks, expected %d" % (n, self.world))

    def close(self):
        """destroy the C-ABI communicator (idempotent; TrainStep.close() / __del__ call it)"""
        if self._comm is not None:
            comm, self._comm = self._comm, None
            torch.cuda.synchronize(self.flat_g.device)
            hipabi.check(hipabi.lib().straps_comm_destroy(comm), 'straps_comm_destroy')

    def __del__(self):
        # (a safety net only -- callers close() explicitly: at interpreter shutdown the HIP runtime / RCCL may already be torn down, and a device
        #  synchronisation or ncclCommDestroy there can hang or abort where no `except` reaches)
        if sys is None or sys.is_finalizing():      # (module globals are cleared late in shutdown: `sys` itself may be gone; an import here would raise)
            return
        try:
            self.close()
        except Exception:      # noqa: BLE001
            pass

    def _rccl_allreduce(self, lo, hi):
        """sum all-reduce of flat_g[lo:hi] on the exchange stream, ordered after everything the current stream holds so far"""
        cur = torch.cuda.current_stream(self.flat_g.device)
        self._stream.wait_stream(cur)
        hipabi.check(hipabi.lib().straps_allreduce_grads(C.c_void_p(self.flat_g.data_ptr() + 4 * lo), hi - lo, self._comm,
                                                          C.c_void_p(self._stream.cuda_stream)), 'straps_allreduce_grads')

    def start_tail(self):
        if not self.active:
            return
        if self.backend == 'rccl':
            if not self._started:
                self._rccl_allreduce(self.split_off, self.flat_g.numel())
                self._started = True
        elif self._work is None:
            import torch.distributed as dist
            self._work = dist.all_reduce(self.flat_g[self.split_off:], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        if self.active:
            self.start_tail()                      # (no-op when the caller already started it)
            if self.backend == 'rccl':
                if self.split_off > 0:
                    self._rccl_allreduce(0, self.split_off)
                torch.cuda.current_stream(self.flat_g.device).wait_stream(self._stream)
                self._started = False
            else:
                import torch.distributed as dist
                self._work.wait()
                self._work = None
                if self.split_off > 0:
                    dist.all_reduce(self.flat_g[:self.split_off], op=dist.ReduceOp.SUM, group=self.group)
        return 1.0 / self.world


class TrainStep:
    def __init__(self, regressor, smpl, criterion, batch_size, lr=1e-4, rank=0, world_size=1, seed=1234, group=None,
                 mean_shape=None, mean_cam_t=(0., 0.2, 42.), pose_pool=None, use_graph=False, overlap_wgrad=False,
                 renderer=None, track_metrics=False, comm_overlap=None, pipeline_data=True, smpl_augment_params=None,
                 cam_augment_params=None, bbox_augment_params=None, proxy_rep_augment_params=None, global_masked_mean=False,
                 force_exchange=False, exchange_backend='torch'):
        """force_exchange / exchange_backend: see GradientExchange (force: run the exchange even when world_size == 1; backend 'rccl' = the
        library's own C-ABI all-reduce on a dedicated stream instead of torch.distributed).
        global_masked_mean (data parallel only; default off = the average of per-rank masked means): the joints2D task
        becomes the masked mean over the GLOBAL batch -- one extra 1-float sum all-reduce per step (each rank's visible-joint count,
        issued a step ahead next to the data pipeline, off the critical path).
        use_graph: after two eager warm-up steps, capture data generation + forward + loss + backward (~250 kernel
        launches) in one hipGraph and replay it each step; the gradient all-reduce and Adam stay eager launches.
        *_augment_params: the dictionaries of run_train.py:133-190 (defaults = the values that script sets)."""
        p0 = next(regressor.parameters())
        if not (isinstance(p0, torch.Tensor) and p0.is_cuda):
            raise RuntimeError('TrainStep: regressor parameters must be GPU tensors (call .to(device) first): the STRAPS hot path runs only '
                               'through the HIP library (no CPU fallback)')
        self.dev = p0.device
        self._force_exchange, self._exchange_backend = bool(force_exchange), exchange_backend
        with torch.cuda.device(self.dev):
            self._init(regressor, smpl, criterion, batch_size, lr, rank, world_size, seed, group, mean_shape, mean_cam_t, pose_pool, use_graph,
                       overlap_wgrad, renderer, track_metrics, comm_overlap, pipeline_data, smpl_augment_params, cam_augment_params,
                       bbox_augment_params, proxy_rep_augment_params)
            self.global_masked_mean = bool(global_masked_mean) and world_size > 1
            self._vis_work = {}
            if self.global_masked_mean and use_graph and not pipeline_data:
                raise NotImplementedError('global_masked_mean with hipGraph capture needs the data pipeline (the count exchange cannot sit inside a captured graph)')

    def _init(self, regressor, smpl, criterion, batch_size, lr, rank, world_size, seed, group, mean_shape, mean_cam_t, pose_pool, use_graph,
              overlap_wgrad, renderer, track_metrics, comm_overlap, pipeline_data, smpl_augment_params, cam_augment_params, bbox_augment_params,
              proxy_rep_augment_params):
        self.reg, self.smpl, self.crit = regressor, smpl, criterion
        if getattr(criterion, 'reduction', 'mean') != 'mean':
            raise NotImplementedError("TrainStep: the fused loss kernel implements reduction='mean' (run_train.py:196); use the criterion "
                                      "module with autograd for 'sum'")
        self.B, self.lr, self.rank, self.world, self.group = batch_size, lr, rank, world_size, group
        self.params = list(regressor.parameters()) + list(criterion.parameters())          # run_train.py:200 order
        self.flat_p, self.flat_g, self.gviews = flatten_parameters(self.params, self.dev)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.steps = 0
        self.n_reg = sum(p.numel() for p in regressor.parameters())
        # gradient exchange in two buckets: everything from layer3 on is reduced while backward still runs through layer2,
        # layer1 and the stem (comm_overlap: default on when there is someone to talk to; STRAPS_NO_COMM_OVERLAP=1 turns it off)
        if comm_overlap is None:
            import os
            comm_overlap = (world_size > 1 or self._force_exchange) and os.environ.get('STRAPS_NO_COMM_OVERLAP', '0') != '1'
        self.comm_overlap = bool(comm_overlap)
        split_off, off = 0, 0
        for n_, p_ in regressor.named_parameters():
            if n_.startswith('image_encoder.layer3.'):
                split_off = off
                break
            off += p_.numel()
        self.tail_offset = split_off            # first element of the tail bucket (layer3 on), whether or not the exchange is split there
        self.exchange = GradientExchange(self.flat_g, split_off if self.comm_overlap else 0, world_size, group, force=self._force_exchange,
                                         backend=self._exchange_backend, rank=rank)
        self.time_exchange, self.exchange_events = False, []
        self.logvar_params = [getattr(criterion, n + '_log_var') for n in TASKS]
        # the five loss log-variances are the last five floats of the flat buffers, in the criterion's registration order; the
        # loss kernel wants its own task order: two 5-element gathers (values in, gradients out) instead of torch stack / copies
        self._lv_index = None
        if all(p.requires_grad and p in self.gviews for p in self.logvar_params):
            offs = [(self.gviews[p].data_ptr() - self.flat_g.data_ptr()) // 4 for p in self.logvar_params]
            base = min(offs)
            if sorted(o - base for o in offs) == list(range(5)):
                rel = [o - base for o in offs]                                     # kernel slot k reads flat slot rel[k]
                inv = [rel.index(i) for i in range(5)]                             # flat slot i receives kernel slot inv[i]
                self._lv_index = torch.tensor(rel, dtype=torch.int32, device=self.dev)
                self._lv_index_inv = torch.tensor(inv, dtype=torch.int32, device=self.dev)
                self._lv_params_flat = self.flat_p[base:base + 5]
                self._lv_grads_flat = self.flat_g[base:base + 5]
        # random draws: device-resident Philox generator, seeded per rank; its step counter lives on the device (graph replay)
        from .device_rng import DeviceDraws
        self.draws = DeviceDraws(seed + rank, self.dev)
        self.smpl_augment_params = dict(SMPL_AUGMENT_PARAMS if smpl_augment_params is None else smpl_augment_params)
        self.cam_augment_params = dict(CAM_AUGMENT_PARAMS if cam_augment_params is None else cam_augment_params)
        self.bbox_augment_params = dict(BBOX_AUGMENT_PARAMS if bbox_augment_params is None else bbox_augment_params)
        self.proxy_rep_augment_params = dict(PROXY_REP_AUGMENT_PARAMS if proxy_rep_augment_params is None else proxy_rep_augment_params)
        assert self.smpl_augment_params['delta_betas_distribution'] in ['uniform', 'normal']
        self.step_t = torch.zeros(1, dtype=torch.int64, device=self.dev)     # Adam step count, lives on the device
        self.use_graph, self.graph, self.graph_tail, self._warm, self._g_loss = use_graph, None, None, 0, None
        self.side_stream = torch.cuda.Stream(device=self.dev) if overlap_wgrad else None
        d = self.dev
        self.mean_shape = torch.zeros(10, device=d) if mean_shape is None else torch.as_tensor(mean_shape, dtype=torch.float32, device=d)
        self.mean_cam_t = torch.tensor(mean_cam_t, device=d).expand(batch_size, 3).contiguous()
        self._eye = torch.eye(3, device=d).expand(batch_size, 24, 3, 3).contiguous()        # rest pose of the 'reposed' SMPL passes
        K = np.array([[config.FOCAL_LENGTH, 0., config.REGRESSOR_IMG_WH / 2.0], [0., config.FOCAL_LENGTH, config.REGRESSOR_IMG_WH / 2.0],
                      [0., 0., 1.]], dtype=np.float32)
        self.cam_K = torch.from_numpy(K).to(d)
        from .augmentation import remove_probabilities
        self.remove_prob = remove_probabilities(self.proxy_rep_augment_params, d)
        std = self.smpl_augment_params.get('delta_betas_std_vector')
        self._std_vector = None if std is None else torch.as_tensor(std, dtype=torch.float32, device=d).expand(10).contiguous()
        # dataset shape rows (used only with augment_shape = False): the pose-pool stand-in carries the mean shape
        self.pool_shape = self.mean_shape[None].expand(batch_size, 10).contiguous()
        # part-segmentation renderer of the target meshes (run_train.py:119-124: NMRRenderer(batch, cam_K, cam_R = I, 256, parts))
        if renderer is None:
            if smpl.faces is None or smpl.face_parts is None:
                raise RuntimeError('TrainStep: the SMPL model carries no faces / face_parts; pass renderer=NMRRenderer(...)')
            from .nmr_renderer import NMRRenderer
            renderer = NMRRenderer(batch_size, self.cam_K.cpu(), torch.eye(3), config.REGRESSOR_IMG_WH, rend_parts_seg=True,
                                   faces=smpl.faces, face_parts=smpl.face_parts).to(d)
        self.renderer = renderer
        # optional per-batch metric sums of the reference's tracker (train loop :236, metrics/train_loss_and_metrics_tracker.py
        # :127-213), accumulated on the device inside the step (no host sync); `metrics_summary()` normalises them
        self.metrics = None
        if track_metrics:
            from .metrics import BatchMetrics
            self.metrics = BatchMetrics(self.dev)
        self._coco = torch.tensor(config.ALL_JOINTS_TO_COCO_MAP, device=d)
        self._h36m14 = torch.tensor([config.ALL_JOINTS_TO_H36M_MAP[k] for k in config.H36M_TO_J14], device=d)
        # the BatchNorm step counters become views of one int64 buffer: one add per step instead of one tiny launch per layer
        bns = [m for m in regressor.image_encoder.modules() if isinstance(m, torch.nn.BatchNorm2d) and m.track_running_stats
               and m.num_batches_tracked is not None]
        self.nbt_flat = torch.stack([m.num_batches_tracked.detach().to(d) for m in bns]) if bns else None
        for k, m in enumerate(bns):
            m.num_batches_tracked = self.nbt_flat[k]
        regressor.image_encoder._nbt_flat = self.nbt_flat
        # stand-in for data/synthetic_training_dataset.py: a resident pool of (pose axis-angle [72]) samples
        if pose_pool is None:
            g = torch.Generator().manual_seed(seed)
            pose_pool = torch.randn(4096, 72, generator=g) * 0.2
            pose_pool[:, :3] = 0
            pose_pool[:, 1] = (torch.rand(4096, generator=g) * 2 - 1) * np.pi * 0.5       # global orientation about y
        self.pose_pool = pose_pool.to(d)
        self.last = {}
        # data pipeline: the batch of step t+1 (two SMPL forwards, rasteriser, crop, augmentation, proxy construction: ~0.8 ms of
        # small latency-bound kernels) is generated on a second stream WHILE step t runs its MFMA-bound forward / backward, into
        # the other of two resident buffer sets.  The random draws happen in the same order as without the pipeline, so the
        # results are bit-identical.
        self.pipeline = bool(pipeline_data)
        self.data_stream = torch.cuda.Stream(device=d) if self.pipeline else None
        self._bufs = [self._new_buffers() for _ in range(2)] if self.pipeline else None
        self._cur, self._primed = 0, False
        self._last_by_parity = {}

    # ------------------------------------------------------------------ data generation (no grad)
    def _new_buffers(self):
        d, B = self.dev, self.B
        return dict(input=torch.empty(B, 18, 256, 256, device=d), verts=torch.empty(B, 6890, 3, device=d),
                    joints2d=torch.empty(B, 17, 2, device=d), joints3d=torch.empty(B, 14, 3, device=d),
                    shape=torch.empty(B, 10, device=d), rot=torch.empty(B, 24, 3, 3, device=d),
                    reposed=torch.empty(B, 6890, 3, device=d), cam_t=torch.empty(B, 3, device=d), vis_count=torch.zeros(1, device=d),
                    nzmask=torch.empty(hipabi.lib().straps_stem_nzmask_words(B, 18, 256, 256), device=d, dtype=torch.int32))

    def draw_layout(self):
        """where each consumer's draws sit in the step's two Philox buffers (element offsets; sub-stream 0 = uniforms,
        sub-stream 1 = normals).  The oracle regenerates the same buffers from (seed + rank, step)."""
        B = self.B
        u, off = {}, 0
        for name, n in (('pose_index', B), ('cam_z', B), ('crop', 3 * B), ('seg', 9 * B), ('joints2d', 34 * B), ('shape_uniform', 10 * B),
                        ('verts2d', 2 * 6890 * B)):
            u[name] = (off, n)
            off += (n + 3) // 4 * 4                      # keep every segment 16-byte aligned
        n_shape = (10 * B + 3) // 4 * 4
        return {'uniform': u, 'n_uniform': off, 'normal': {'shape': (0, 10 * B), 'cam_xy': (n_shape, 2 * B)}, 'n_normal': n_shape + 2 * B}

    def make_batch(self, out=None, keep=None):
        """one synthetic batch (train loop :112-182), every stage a C-ABI call: draws -> G1/G2 -> SMPL x2 -> P2 -> part
        rasteriser (with the vertex noise) -> crop/resize -> G3 -> G4+G5 -> non-zero map.  out: optional dict of resident
        buffers to fill (the data pipeline); keep: optional dict that receives the intermediate tensors (tests)."""
        L, st, d, B = hipabi.lib(), hipabi.stream_ptr(), self.dev, self.B
        out = self._new_buffers() if out is None else out
        lay = self.draw_layout()
        U = self.draws.fill(torch.empty(lay['n_uniform'], device=d), 0, 0)
        N = self.draws.fill(torch.empty(lay['n_normal'], device=d), 1, 1)
        self.draws.advance()

        def useg(name):
            o, n = lay['uniform'][name]
            return U[o:o + n]

        def nseg(name):
            o, n = lay['normal'][name]
            return N[o:o + n]
        sp, cp, bp, pp = self.smpl_augment_params, self.cam_augment_params, self.bbox_augment_params, self.proxy_rep_augment_params
        # G1: dataset row (stand-in pose pool) + shape resampling + axis-angle -> rotation matrices (smpl_augmentation.py:27-61)
        mode = 0
        if sp['augment_shape']:
            mode = 1 if sp['delta_betas_distribution'] == 'normal' else 2
        sdraws = nseg('shape') if mode == 1 else useg('shape_uniform')
        rng = sp['delta_betas_range'] if mode == 2 else (0.0, 0.0)
        tgt_shape, tgt_rot = out['shape'], out['rot']
        hipabi.check(L.straps_augment_smpl(hipabi.ptr(self.pose_pool), self.pose_pool.shape[0], hipabi.ptr(useg('pose_index')),
                                           hipabi.ptr(self.pool_shape), hipabi.ptr(self.mean_shape), hipabi.ptr(sdraws), mode, hipabi.ptr(self._std_vector),
                                           float(rng[0]), float(rng[1]), hipabi.ptr(tgt_shape), hipabi.ptr(tgt_rot), None, B, st),
                     'straps_augment_smpl')
        # G2: camera translation (cam_augmentation.py:4-14)
        cam_t = out['cam_t']
        hipabi.check(L.straps_augment_cam_t(hipabi.ptr(self.mean_cam_t), hipabi.ptr(nseg('cam_xy')), hipabi.ptr(useg('cam_z')), float(cp['xy_std']),
                                            float(cp['delta_z_range'][0]), float(cp['delta_z_range'][1]), hipabi.ptr(cam_t), B, st),
                     'straps_augment_cam_t')
        # SMPL #1 / #2
        tgt_verts, tgt_joints = self.smpl.forward_arrays(tgt_shape, tgt_rot, out_verts=out['verts'])
        self.smpl.forward_arrays(tgt_shape, self._eye, want_joints=False, out_verts=out['reposed'])
        # H36M-LSP 3D joints + P2: perspective projection of the COCO joints (utils/cam_utils.py:40-71, cam_R = I)
        j2d_full = torch.empty(B, 17, 2, device=d)
        wh = float(config.REGRESSOR_IMG_WH)
        hipabi.check(L.straps_project_targets(hipabi.ptr(tgt_joints), hipabi.ptr(cam_t), config.FOCAL_LENGTH, config.FOCAL_LENGTH, wh / 2, wh / 2,
                                              hipabi.ptr(j2d_full), hipabi.ptr(out['joints3d']), B, st), 'straps_project_targets')
        # part segmentation of the (noisy copy of the) target mesh: random_verts2D_deviation folded into the rasteriser's
        # projection kernel (train loop :146-155, proxy_rep_augmentation.py:5-22); the loss keeps the clean vertices
        noise = useg('verts2d') if pp['deviate_verts2D'] else None
        seg = self.renderer.render_arrays(tgt_verts, cam_t, vert_noise_u=noise, noise_range=pp['delta_verts2d_dev_range'])
        # bounding-box crop with scale / centre jitter + nearest resize back to 256 (train loop :161-170), on the device; the 2-D
        # joint targets follow the crop like in the reference
        if bp['crop_input']:
            from .image_utils import batch_crop_and_resize
            seg_c, tgt_j2d, boxes = batch_crop_and_resize(seg, j2d_full, config.REGRESSOR_IMG_WH, bp['mean_scale_factor'], bp['delta_scale_range'],
                                                          bp['delta_centre_range'], uniforms=useg('crop'), jout=out['joints2d'])
        else:
            seg_c, boxes = seg, None
            tgt_j2d = out['joints2d']
            hipabi.check(L.straps_masked_copy(hipabi.ptr(j2d_full), 34, None, 0, hipabi.ptr(tgt_j2d), 34, B, 34, 0, st), 'joints2d copy')
        # G3: body-part removal + occlusion box (:52-101) and joint jitter (:25-49)
        seg_aug = torch.empty_like(seg_c)
        hipabi.check(L.straps_augment_seg(hipabi.ptr(seg_c), hipabi.ptr(useg('seg')), hipabi.ptr(self.remove_prob),
                                          float(pp['occlude_probability']) if pp['occlude_seg'] else 0.0, int(pp['occlude_box_dim']),
                                          hipabi.ptr(seg_aug), B, 256, st), 'straps_augment_seg')
        j2d_in = tgt_j2d
        if pp['deviate_joints2D']:
            j2d_in = torch.empty(B, 17, 2, device=d)
            r, hr = pp['delta_j2d_dev_range'], pp['delta_j2d_hip_dev_range']
            hipabi.check(L.straps_deviate_joints2d(hipabi.ptr(tgt_j2d), hipabi.ptr(useg('joints2d')), float(r[0]), float(r[1]), float(hr[0]),
                                                   float(hr[1]), hipabi.ptr(j2d_in), B, st), 'straps_deviate_joints2d')
        # G4 + G5
        x = out['input']
        if _FUSED_PROXY_NZ:
            # ... and, in the same pass, the non-zero map of the input for the stem's zero skipping (round 4: straps_stem_nzmask's read of
            # the 302 MB input is gone)
            hipabi.check(L.straps_build_proxy_input_nz(hipabi.ptr(seg_aug), hipabi.ptr(j2d_in), hipabi.ptr(x), hipabi.ptr(out['nzmask']), B, 17, 256, 4, st),
                         'straps_build_proxy_input_nz')
        else:
            hipabi.check(L.straps_build_proxy_input(hipabi.ptr(seg_aug), hipabi.ptr(j2d_in), hipabi.ptr(x), B, 17, 256, st), 'straps_build_proxy_input')
            hipabi.check(L.straps_stem_nzmask(hipabi.ptr(x), hipabi.ptr(out['nzmask']), B, 18, 256, 256, st), 'straps_stem_nzmask')
        # this rank's visible target joints (global masked mean under data parallel: summed over the ranks before the loss runs)
        hipabi.check(L.straps_count_visible(hipabi.ptr(tgt_j2d), hipabi.ptr(out['vis_count']), B, 17, config.REGRESSOR_IMG_WH, st), 'straps_count_visible')
        if keep is not None:
            keep.update(uniforms=U, normals=N, joints=tgt_joints, joints2d_uncropped=j2d_full, seg=seg, seg_cropped=seg_c, boxes=boxes,
                        seg_aug=seg_aug, joints2d_input=j2d_in)
        return out

    # ------------------------------------------------------------------ forward + loss + backward
    def forward_backward(self, batch, after_layer3=None):
        L, st, d, B = hipabi.lib(), hipabi.stream_ptr(), self.dev, self.B
        reg, smpl = self.reg, self.smpl
        assert reg.training, 'TrainStep needs the regressor in .train() mode'
        hipabi.check(L.straps_memset_zero(hipabi.ptr(self.flat_g), self.flat_g.numel() * 4, st), 'straps_memset_zero(flat gradient)')
        enc_tape, ief_tape = {}, []
        reg.image_encoder.prepack(with_dgrad=True)          # every conv's forward + data-gradient weight layout, one launch
        if self.nbt_flat is not None:
            # num_batches_tracked of every BatchNorm, one launch (encoder_exec defers to this)
            hipabi.check(L.straps_counter_add(hipabi.ptr(self.nbt_flat), self.nbt_flat.numel(), 1, st), 'straps_counter_add(num_batches_tracked)')
        feat = encoder_forward(reg.image_encoder, batch['input'], enc_tape, nzmask=batch.get('nzmask'))
        est = reg.ief_module.forward_estimate(feat, ief_tape)                      # [B,160]
        pose6d = est[:, 3:147]
        R = torch.empty(B, 24, 3, 3, device=d)
        hipabi.check(L.straps_rot6d_fwd(hipabi.ptr(pose6d), EST_LD, 24, hipabi.ptr(R), B, st), 'straps_rot6d_fwd')
        pred_shape = torch.empty(B, 10, device=d)
        hipabi.check(L.straps_masked_copy(C.c_void_p(est.data_ptr() + 4 * 147), EST_LD, None, 0, hipabi.ptr(pred_shape), 10, B, 10, 0, st), 'pred_shape')
        verts, joints = smpl.forward_arrays(pred_shape, R)                         # SMPL #3
        eye = self._eye
        reposed, _ = smpl.forward_arrays(pred_shape, eye, want_joints=False)       # SMPL #4 (metrics only, train loop :206)
        # heads + loss + gradients
        if self._lv_index is not None:                     # log-variances: flat-buffer (registration) order -> kernel task order
            lv = torch.empty(5, device=d)
            hipabi.check(L.straps_gather_f32(hipabi.ptr(self._lv_params_flat), hipabi.ptr(self._lv_index), hipabi.ptr(lv), 5, st), 'log-var gather')
        else:
            lv = self.crit.log_var_vector()
        loss = torch.empty(12, device=d)
        dverts, djoints = torch.empty_like(verts), torch.empty_like(joints)
        dest, drot = torch.empty(B, EST_LD, device=d), torch.empty(B, 24, 3, 3, device=d)
        dlv = torch.empty(5, device=d)
        ws = torch.empty(L.straps_loss_workspace_bytes(B) // 4, device=d)
        gm = getattr(self, 'global_masked_mean', False)
        hipabi.check(L.straps_loss_fwd_bwd_gm(hipabi.ptr(verts), hipabi.ptr(joints), hipabi.ptr(est), EST_LD, hipabi.ptr(R), hipabi.ptr(batch['verts']),
                                              hipabi.ptr(batch['joints2d']), hipabi.ptr(batch['joints3d']), hipabi.ptr(batch['shape']),
                                              hipabi.ptr(batch['rot']), hipabi.ptr(lv), hipabi.ptr(loss), hipabi.ptr(dverts), hipabi.ptr(djoints),
                                              hipabi.ptr(dest), hipabi.ptr(drot), hipabi.ptr(dlv), hipabi.ptr(ws), B, config.REGRESSOR_IMG_WH,
                                              hipabi.ptr(batch['vis_count'] if gm else None), 1.0 / self.world, st),
                     'straps_loss_fwd_bwd')
        # SMPL backward -> (dbetas, drot2)
        dbetas, drot2 = torch.empty(B, 10, device=d), torch.empty(B, 24, 3, 3, device=d)
        ws2 = torch.empty(L.straps_smpl_bwd_workspace_bytes(B, 0) // 4, device=d)
        hipabi.check(L.straps_smpl_bwd(C.byref(smpl._model_struct()), hipabi.ptr(pred_shape), hipabi.ptr(R), hipabi.ptr(dverts), hipabi.ptr(djoints),
                                       hipabi.ptr(dbetas), hipabi.ptr(drot2), hipabi.ptr(ws2), B, 0, st), 'straps_smpl_bwd')
        # gather into d(est): shape columns += dbetas; pose columns = rot6d backward of (drot + drot2)
        hipabi.check(L.straps_masked_copy(hipabi.ptr(drot2), 216, None, 0, hipabi.ptr(drot), 216, B, 216, 1, st), 'drot sum')
        hipabi.check(L.straps_masked_copy(hipabi.ptr(dbetas), 10, None, 0, C.c_void_p(dest.data_ptr() + 4 * 147), EST_LD, B, 10, 1, st), 'dshape sum')
        hipabi.check(L.straps_rot6d_bwd(hipabi.ptr(pose6d), EST_LD, 24, hipabi.ptr(drot), C.c_void_p(dest.data_ptr() + 4 * 3), EST_LD, B, st),
                     'straps_rot6d_bwd')
        # regressor backward, gradients land in the flat buffer
        dfeat, _ = ief_backward(reg.ief_module, feat, ief_tape, dest, self.gviews)
        # d(loss)/d(log-variances) -> their slots at the very end of the flat gradient buffer (part of the tail bucket)
        if self._lv_index is not None:
            hipabi.check(L.straps_gather_f32(hipabi.ptr(dlv), hipabi.ptr(self._lv_index_inv), hipabi.ptr(self._lv_grads_flat), 5, st), 'log-var gradient scatter')
        else:
            for k, p in enumerate(self.logvar_params):
                if p.requires_grad:
                    self.gviews[p].copy_(dlv[k])
        encoder_backward(reg.image_encoder, enc_tape, dfeat, self.gviews, self.side_stream, after_layer3)
        # (bwd: the backward chain's intermediate gradients, in the order they are produced -- tests/test_gpu_two_ranks.py digests them to name the
        #  first kernel family whose output differs between two launch forms)
        self.last = dict(loss=loss, verts=verts, joints=joints, est=est, reposed=reposed, rot=R, ief_tape=ief_tape,
                         enc_tape=enc_tape if getattr(self, 'keep_enc_tape', False) else None,      # (keep_enc_tape: tests read the decisions taken)
                         # (keep_bwd, like keep_enc_tape: off in production -- under hipGraph capture these references would pin the tensors in the private pool)
                         bwd=dict(dverts=dverts, djoints=djoints, dlv=dlv, dbetas=dbetas, drot_smpl=drot2, dest=dest, dfeat=dfeat)
                         if getattr(self, 'keep_bwd', False) else None)
        if self.metrics is not None:
            from .cam_utils import orthographic_project_torch
            pred = {'verts': verts, 'joints3D': joints.index_select(1, self._h36m14), 'shape_params': pred_shape,
                    'pose_params_rot_matrices': R,
                    'joints2D': orthographic_project_torch(joints.index_select(1, self._coco), est[:, :3])}
            tgt = {'verts': batch['verts'], 'joints3D': batch['joints3d'], 'shape_params': batch['shape'],
                   'pose_params_rot_matrices': batch['rot'], 'joints2D': batch['joints2d']}
            self.metrics.update(pred, tgt, pred_reposed_vertices=reposed, target_reposed_vertices=batch['reposed'])
        return loss

    def metrics_summary(self):
        """per-sample means of the tracked metrics over the steps taken so far (tracker's update_per_epoch normalisers)."""
        if self.metrics is None:
            raise RuntimeError('TrainStep was built with track_metrics=False')
        self.metrics.n = self.steps * self.B          # a replayed hipGraph does not run the Python-side counter
        return self.metrics.summary()

    def optimise(self):
        if self.time_exchange and self.exchange.active:
            # event pair on the step's stream around the wait for the tail bucket + the head bucket's all-reduce: the part of the exchange
            # that backward did NOT hide (bench.py --gpus N reports it)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gscale = self.exchange.finish()
            e1.record()
            self.exchange_events.append((e0, e1))
        else:
            gscale = self.exchange.finish()
        self.steps += 1
        hipabi.check(hipabi.lib().straps_counter_add(hipabi.ptr(self.step_t), 1, 1, hipabi.stream_ptr()), 'straps_counter_add(adam step)')
        hipabi.check(hipabi.lib().straps_adam_step(hipabi.ptr(self.flat_p), hipabi.ptr(self.flat_g), hipabi.ptr(self.exp_avg),
                                                   hipabi.ptr(self.exp_avg_sq), self.flat_p.numel(), self.steps, self.lr, 0.9, 0.999, 1e-8,
                                                   gscale, hipabi.ptr(self.step_t), hipabi.stream_ptr()), 'straps_adam_step')
        # the parameters changed behind torch's back: drop the packed-weight caches
        self.reg.image_encoder._cache.clear()
        self.reg.ief_module._cache = {}

    def _run(self, after_layer3):
        """forward + backward of the current batch on the current stream; with the data pipeline the next batch is generated
        meanwhile on the data stream (joined before `after_layer3` -- where a split capture ends its first graph -- and at the
        end).  Used for eager launches and, unchanged, under hipGraph capture."""
        if not self.pipeline:
            b = self.make_batch()
            if self.global_masked_mean:                        # (no pipeline: the count exchange sits between the batch and the forward)
                allreduce_visible_count(b['vis_count'], self.world, self.group)
            return self.forward_backward(b, after_layer3)
        if not self._primed:                                   # very first step: there is no batch in flight yet
            self.make_batch(out=self._bufs[self._cur])
            self._primed = True
        cur, nxt = self._bufs[self._cur], self._bufs[1 - self._cur]
        main, side = torch.cuda.current_stream(), self.data_stream
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self.make_batch(out=nxt)
        state = {'joined': False}

        def join():
            if not state['joined']:
                main.wait_stream(side)
                state['joined'] = True

        def hook():
            join()
            after_layer3()

        loss = self.forward_backward(cur, hook if after_layer3 is not None else None)
        join()
        return loss

    def step(self):
        """one full training step; returns the 12-float loss record (device tensor, no sync).  Runs on the regressor's
        device whatever the caller's current device is."""
        with torch.cuda.device(self.dev), torch.no_grad():
            start_tail = self.exchange.start_tail if self.comm_overlap else None
            gm = self.global_masked_mean and self.pipeline
            if gm:
                # the job-wide visible-joint count of the batch this step trains on: its all-reduce was started a step ago (below);
                # the very first batch is generated here and its count exchanged before anything runs
                if not self._primed:
                    self.make_batch(out=self._bufs[self._cur])
                    self._primed = True
                w = self._vis_work.pop(self._cur, None)
                if w is None:
                    allreduce_visible_count(self._bufs[self._cur]['vis_count'], self.world, self.group)
                else:
                    w.wait()

            def count_next():
                # the batch of the NEXT step has just been generated (data stream, joined): its count exchange runs under Adam
                if gm:
                    self._vis_work[self._cur] = allreduce_visible_count(self._bufs[self._cur]['vis_count'], self.world, self.group, async_op=True)

            def eager():
                loss = self._run(start_tail)
                self._cur ^= 1
                count_next()
                self.optimise()
                return loss
            if not self.use_graph or self._warm < 2:
                self._warm += 1
                return eager()
            if self.graph is None:
                self._capture()
                if not self.use_graph:             # capture failed (on this rank or, data parallel, on any rank): eager launches from here on --
                    return eager()                 # inline, so the count exchange above is not issued a second time (ADVICE)
            par = self._cur if self.pipeline else 0
            g1, g2, loss = self.graph[par]
            self.last = self._last_by_parity[par]          # the output buffers THIS graph writes (each capture has its own)
            g1.replay()
            if g2 is not None:
                self.exchange.start_tail()       # layer3.. gradients are final: their all-reduce runs under the rest of backward
                g2.replay()
            self._cur ^= 1
            count_next()
            self.optimise()
            return loss

    def _capture_one(self):
        """one step as a hipGraph (or two, split where the tail bucket's gradients are final, sharing a memory pool)."""
        if not self.comm_overlap:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self._pool, capture_error_mode='thread_local'):
                loss = self._run(None)
            if self._pool is None:
                self._pool = g.pool()
            return g, None, loss
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream())
        state = {'second': False}

        def switch():
            g1.capture_end()
            if self._pool is None:
                self._pool = g1.pool()
            g2.capture_begin(pool=self._pool, capture_error_mode='thread_local')
            state['second'] = True

        with torch.cuda.stream(s):
            if self._pool is None:
                g1.capture_begin(capture_error_mode='thread_local')
            else:
                g1.capture_begin(pool=self._pool, capture_error_mode='thread_local')
            try:
                loss = self._run(switch)
            finally:
                (g2 if state['second'] else g1).capture_end()
        torch.cuda.current_stream().wait_stream(s)
        if not state['second']:
            raise RuntimeError('the backward pass never reached the split point')
        return g1, g2, loss

    def _capture(self):
        """capture data generation + forward + loss + backward as hipGraphs: one per buffer parity of the data pipeline, each
        split in two with comm_overlap.  Captured with empty weight caches so the repacking kernels are part of the graph;
        thread_local error mode: other threads (e.g. the RCCL watchdog) may touch the HIP runtime while this thread captures.
        Nothing executes during capture.  Any failure falls back to eager launches of the same kernels."""
        torch.cuda.synchronize()
        self._pool = None
        start, last0, by_parity0 = self._cur, self.last, dict(self._last_by_parity)      # (restored if a capture throws)
        try:
            graphs = {}
            for par in ((start, 1 - start) if self.pipeline else (0,)):
                self._cur = par
                # every captured step must re-pack the weights itself: drop what the previous capture left in the caches
                self.reg.image_encoder._cache.clear()
                self.reg.ief_module._cache = {}
                graphs[par] = self._capture_one()
                self._last_by_parity[par] = self.last
            self._cur = start
            self.graph = graphs
            self.graph_tail = next(iter(graphs.values()))[1]
            ok, why = True, ''
        except Exception as e:                       # noqa: BLE001 -- fall back to eager launches, never to another path
            ok, why = False, str(e)
        if self.world > 1:
            # data parallel: every rank takes the same path -- one rank's failed capture sends all of them to eager launches (ADVICE:
            # ranks that disagree about graph / eager would issue differently ordered collectives)
            import torch.distributed as dist
            flag = torch.tensor([1.0 if ok else 0.0], device=self.dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            if ok and float(flag.item()) < 1.0:
                ok, why = False, 'capture failed on another rank'
        if not ok:
            import warnings
            warnings.warn('hipGraph capture of the training step failed (%s); continuing with eager launches' % (why,))
            self.use_graph, self.graph, self.graph_tail = False, None, None
            # nothing executed during capture: the batch in flight is still the one in self._bufs[start] and no draw was consumed on the device
            # (the generator's step counter lives there), but the Python side ran: put the buffer parity, the output record and the weight caches
            # back so that the eager step that follows trains on the batch it would have trained on, on every rank alike
            self._cur, self.last, self._last_by_parity = start, last0, by_parity0
            self.reg.image_encoder._cache.clear()
            self.reg.ief_module._cache = {}
            torch.cuda.synchronize()

    def close(self):
        """release what the step holds outside torch's allocator: the C-ABI communicator of exchange_backend='rccl' (idempotent)."""
        ex = getattr(self, 'exchange', None)
        if ex is not None:
            ex.close()

    def __del__(self):
        if sys is None or sys.is_finalizing():      # (see GradientExchange.__del__)
            return
        try:
            self.close()
        except Exception:      # noqa: BLE001
            pass

    def state_dict(self):
        """optimiser state in torch.optim.Adam's schema (checkpoint key 'optimiser_state_dict')."""
        state, off = {}, 0
        for i, p in enumerate(self.params):
            n = p.numel()
            state[i] = {'step': torch.tensor(float(self.steps)), 'exp_avg': self.exp_avg[off:off + n].view(p.shape).clone(),
                        'exp_avg_sq': self.exp_avg_sq[off:off + n].view(p.shape).clone()}
            off += n
        group = {'lr': self.lr, 'betas': (0.9, 0.999), 'eps': 1e-8, 'weight_decay': 0, 'amsgrad': False, 'maximize': False,
                 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None, 'params': list(range(len(self.params)))}
        return {'state': state, 'param_groups': [group]}

    def data_state(self):
        """generator step of the batch the next `step()` will train on (with the data pipeline that batch is already in
        flight, drawn one generator step ago).  Synchronises."""
        return self.draws.step() - (1 if self.pipeline and self._primed else 0)

    def set_data_state(self, step):
        """resume the data stream: the next `step()` regenerates its batch from generator step `step`."""
        with torch.cuda.device(self.dev):
            torch.cuda.synchronize()
            self.draws.set_step(step)
            self._primed = False
            for w in getattr(self, '_vis_work', {}).values():
                w.wait()
            self._vis_work = {}
            if self.graph is not None:             # captured graphs consume the batch that was in flight: re-prime eagerly, capture again
                self.graph, self.graph_tail, self._warm, self._last_by_parity = None, None, 0, {}

    def load_state_dict(self, sd):
        """restore the optimiser state written by `state_dict()` / torch.optim.Adam.state_dict() (checkpoint key
        'optimiser_state_dict', run_train.py:204-209): per-parameter exp_avg / exp_avg_sq go back into the flat moment
        buffers in `self.params` order, the step count into the host counter and the device-side Adam step."""
        groups = sd.get('param_groups', [])
        if len(groups) != 1:
            raise ValueError('TrainStep.load_state_dict: expected one parameter group (run_train.py:200), got %d' % len(groups))
        g = groups[0]
        if len(g['params']) != len(self.params):
            raise ValueError('TrainStep.load_state_dict: the state covers %d parameters, this step has %d' % (len(g['params']), len(self.params)))
        if tuple(g.get('betas', (0.9, 0.999))) != (0.9, 0.999) or g.get('eps', 1e-8) != 1e-8 or g.get('weight_decay', 0) != 0 \
                or g.get('amsgrad', False):
            raise ValueError('TrainStep.load_state_dict: only torch.optim.Adam defaults (betas (0.9, 0.999), eps 1e-8, no weight decay, '
                             'no amsgrad) are implemented by straps_adam_step')
        self.lr = float(g.get('lr', self.lr))
        state = sd.get('state', {})
        steps = set()
        with torch.cuda.device(self.dev), torch.no_grad():
            off = 0
            for i, p in zip(g['params'], self.params):
                n = p.numel()
                st = state.get(i, state.get(str(i)))
                if st is None:                              # a parameter that never received a gradient has no state yet
                    self.exp_avg[off:off + n].zero_()
                    self.exp_avg_sq[off:off + n].zero_()
                else:
                    if tuple(st['exp_avg'].shape) != tuple(p.shape):
                        raise ValueError('TrainStep.load_state_dict: parameter %d has shape %s, the state %s' % (i, tuple(p.shape), tuple(st['exp_avg'].shape)))
                    self.exp_avg[off:off + n].copy_(st['exp_avg'].reshape(-1).to(self.dev, torch.float32))
                    self.exp_avg_sq[off:off + n].copy_(st['exp_avg_sq'].reshape(-1).to(self.dev, torch.float32))
                    steps.add(int(float(st['step'])))
                off += n
            if len(steps) > 1:
                raise ValueError('TrainStep.load_state_dict: per-parameter step counts differ (%s); one fused Adam launch keeps one count' % sorted(steps))
            self.steps = steps.pop() if steps else 0
            self.step_t.fill_(self.steps)

