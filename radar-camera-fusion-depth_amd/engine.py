'''
Execution engine for the FusionNet hot path: runs FusionNetEncoder.forward (src/networks.py:840-1005),
MultiScaleDecoder.forward (src/networks.py:1557-1657) and the depth map of FusionNetModel.forward
(src/fusionnet_model.py:160-165) as a sequence of HIP kernel launches on the caller's stream, and records a
tape of backward closures that replaces autograd for this path (loss.backward(), src/fusionnet_main.py:398).

Why a tape and not torch.autograd per op: the engine controls which tensors exist (the upsampled and the
concatenated decoder tensors never do), in what order gradients are produced (so gradient buckets can be
all-reduced while the encoder's backward is still running) and where every gradient accumulates (dgrad kernels
accumulate in place; no generic add kernels).  Torch owns memory and the stream only.

Layout: activations (N, H, W, C) fp32 contiguous.  BN layers keep the raw conv output z and the per-channel
coefficients.  The activation of a plain conv+BN+lrelu block can be DEFERRED: consumers that apply BatchNorm + lrelu
while they load z (the output head; with Engine.bn_on_load also the split conv / wgrad kernels) take (z, coef) and
the tensor is never written; any other consumer materialises it once, on demand (Engine._mat).
'''

import os

import torch

from . import ops
from ._lib import (RCF_ACT_LEAKY_RELU, RCF_ACT_NONE, RCF_GATHER_DIRECT, RCF_GATHER_NEAREST, RCF_PHASE_S2_DGRAD,
                   RCF_PHASE_UP2X_DGRAD, RCF_PHASE_UP2X_FWD, RCF_PREC_F16X2, RCF_PREC_FP32, ConvDesc)

class WeightPlan(object):
    '''
    All weight transforms of a training step in a handful of launches.  A step packs ~200 weights into the kernels' layouts (the
    forward form of every layer, the flipped form for its input gradient, the four 2x2 phase forms of the up-2x and stride-2
    layers), each a 4-5 us launch of a few thousand elements: ~1 ms of GPU time and a fifth of the step's launches.  The weights
    do not change between the start of the forward pass and the end of the backward pass, so they can all be packed up front.

    The plan is positional.  In the first training step (state 'record') the engine's _phase_w / _pack / _pack_n do their work one
    launch at a time, as without a plan, into PERSISTENT buffers and note (kind, descriptor, source pointers, buffer).  end() seals
    the list into two ctypes arrays.  From then on (state 'replay') begin() issues rcf_phase_weights_batch +
    rcf_conv2d_pack_weights_batch over the arrays -- they read the parameters' current values -- and the i-th request of the step
    only checks that it is the request recorded at position i (same kind, descriptor bytes and source pointers) and returns the
    buffer.  Any mismatch (another shape, another precision, a model that changed) marks the plan dirty: the rest of that step packs
    one launch at a time again, and the next step records afresh.  Results are bitwise those of the unbatched path.

    Invariant: the parameters must not change between the start of forward() and the end of backward() of a training step (the packed
    forms are made once, up front).  The reference's loop satisfies it (forward -> loss -> backward -> optimizer.step); code that writes
    parameters in between (load_state_dict, restore_model, a manual copy_ into a weight) must do so between steps, or switch the
    batching off (FusionNetModel.batch_weight_packing = False / RCF_BATCH_PACK=0).

    Two-plane fp16 arithmetic (RCF_PREC_F16X2) adds a third kind of entry: max|w| of every weight tensor (and phase-weight buffer)
    the split kernels consume, one device scalar each in a persistent arena -- replayed as ONE memset + rcf_amax_batch between the
    phase-weight batch (whose outputs it reads) and the packing batch (which scales by it).
    '''

    def __init__(self):
        self.state = 'off'      # 'off' | 'record' | 'replay'
        self.entries = []
        self.pos = 0
        self.dirty = False
        self.active = False         # between begin() and end(): the forward AND the backward pass of one training step
        self.last_mismatch = None   # (position, requested kind, what differed): diagnostics
        self._pack_items = self._phase_items = self._amax_items = None
        self._n_pack = self._n_phase = self._n_amax = 0
        self.wamax = None           # arena of the weight maxima (device scalars) of the recorded step
        self._wamax_used = 0
        self.wamax_chunks = []
        self.fence = None           # callable set by the engine: orders a fresh zero-filled chunk before all of its streams

    def enable(self):
        if self.state == 'off':
            self.state = 'record'

    def begin(self):
        if self.state == 'off':
            return
        if self.state == 'replay' and not self.dirty:
            if self._n_phase:
                ops.phase_weights_batch(self._phase_items, self._n_phase)
            if self._n_amax:
                for chunk in self.wamax_chunks:
                    chunk.zero_()
                ops.amax_batch(self._amax_items, self._n_amax)
            if self._n_pack:
                ops.conv_pack_batch(self._pack_items, self._n_pack)
        else:
            self.state, self.entries, self.dirty = 'record', [], False
            self.wamax, self._wamax_used, self.wamax_chunks = None, 0, []
        self.pos = 0
        self.active = True

    def end(self):
        if not self.active:
            return
        self.active = False
        if self.state == 'record':
            self._seal()
        elif self.state == 'replay' and self.pos != len(self.entries):
            self.dirty = True

    def new_wamax_slot(self, like):
        '''A zeroed device scalar of the recorded step's weight-maximum arena.'''
        if self.wamax is None or self._wamax_used >= self.wamax.numel():
            # first slot, or the current chunk is full (a net with more weight tensors): another chunk; every chunk is zeroed before
            # the batched maxima of a replayed step
            self.wamax = torch.zeros(1024, dtype=torch.float32, device=like.device)
            self.wamax_chunks.append(self.wamax)
            self._wamax_used = 0
            if self.fence is not None:
                self.fence()        # (Engine._fence_streams: the zero fill is ordered before every stream's use of a slot)
        self._wamax_used += 1
        return self.wamax[self._wamax_used - 1:self._wamax_used]

    def _seal(self):
        import ctypes
        from ._lib import AmaxItem, PackItem, PhaseItem
        packs, phases, amaxes = [], [], []
        for e in self.entries:
            if e['kind'] == 'phase':
                phases.append((e['w'].data_ptr(), e['out'].data_ptr(), e['w'].shape[0], e['w'].shape[1], e['mode']))
            elif e['kind'] == 'amax':
                amaxes.append((e['ptr'], e['n'], e['out'].data_ptr()))
            else:
                for w, dst in zip(e['srcs'], e['dsts']):
                    packs.append((e['desc'], w.data_ptr(), dst.data_ptr(), None if e.get('amax') is None else e['amax'].data_ptr()))
        self._n_pack, self._n_phase, self._n_amax = len(packs), len(phases), len(amaxes)
        self._pack_items = (PackItem * max(1, len(packs)))()
        for i, (d, w, dst, am) in enumerate(packs):
            self._pack_items[i].desc = ctypes.pointer(d)      # the descriptor object stays alive in its entry
            self._pack_items[i].w_oihw = w
            self._pack_items[i].packed = dst
            self._pack_items[i].amax_w = am
        self._amax_items = (AmaxItem * max(1, len(amaxes)))()
        for i, (x, n, out) in enumerate(amaxes):
            self._amax_items[i].x, self._amax_items[i].n, self._amax_items[i].amax = x, n, out
        self._phase_items = (PhaseItem * max(1, len(phases)))()
        for i, (w, out, o, ii, mode) in enumerate(phases):
            self._phase_items[i].w_oihw, self._phase_items[i].out = w, out
            self._phase_items[i].o, self._phase_items[i].i, self._phase_items[i].mode = o, ii, mode
        self.state = 'replay'

    def _next(self, kind):
        if self.dirty or self.pos >= len(self.entries) or self.entries[self.pos]['kind'] != kind:
            if not self.dirty:
                self.last_mismatch = (self.pos, kind, self.entries[self.pos]['kind'] if self.pos < len(self.entries) else 'past the end')
            self.dirty = True
            return None
        e = self.entries[self.pos]
        self.pos += 1
        return e


BN_EPS = 1e-5       # torch.nn.BatchNorm2d default (src/net_utils.py:82)
BN_MOMENTUM = 0.1


class Act(object):
    '''An activation tensor and (during backward) its gradient accumulator.'''
    __slots__ = ('t', 'g', 'needs_grad', 'head_fusable', 'g_head', 'z', 'coef', 's2d', 'hw', 'amax', 'bn', 'bsum')

    def __init__(self, t, needs_grad=True):
        self.t = t
        self.g = None
        self.needs_grad = needs_grad
        self.head_fusable = False   # produced by a plain conv+BN+lrelu block: its BN backward can absorb the head's dgrad
        self.g_head = None          # (dlogit, head weight): the gradient in un-materialised form
        self.z = None               # deferred activation: t is None and the consumer applies coef (BN + lrelu) to z on load
        self.coef = None
        self.s2d = None             # network input only (bf16 configuration): its space-to-depth image for the stem, and (H, W)
        self.hw = None
        self.amax = None            # two-plane fp16 arithmetic: device scalar holding max|t| (or an upper bound), from its producer
        self.bn = None              # (z, coef) of the BatchNorm + lrelu block that produced t: an input-gradient kernel that is the
        self.bsum = None            # only writer of g may leave that block's backward sums here, (partials, rows) -- any later
                                    # writer of g clears them (stale)


class _Tape(list):
    """The backward tape.  A closure recorded while the engine runs the depth branch on its own stream (Engine._branch_enter) replays
    on that stream too: the two encoder branches never read each other's gradients (the fusion's backward hands the depth branch its
    share explicitly), so their backward chains overlap like their forward ones."""

    def __init__(self, engine):
        list.__init__(self)
        self.engine = engine

    def append(self, fn):
        eng = self.engine
        if eng._in_branch and not eng._tape_main:
            def on_branch(fn=fn):
                ctx = eng._branch_enter(first_wait=False)
                try:
                    fn()
                finally:
                    eng._branch_exit(ctx)
            list.append(self, on_branch)
        else:
            list.append(self, fn)


class Engine(object):
    def __init__(self, encoder, decoder, min_predict_depth, max_predict_depth):
        self.encoder = encoder
        self.decoder = decoder
        self.dmin = float(min_predict_depth)
        self.dmax = float(max_predict_depth)
        self.grad_of = None        # callable: parameter -> gradient tensor to write (set by the model)
        self.on_param_grad = None  # callable(parameter): called once the parameter's gradient is enqueued
        self.tape = None
        self.training = True
        self.plan = WeightPlan()   # batched weight transforms of a training step (off until the model enables it)
        self.kernel_log = None     # optional list collecting (layer, kernel_id) for profiling
        self.up2x_wgrad_direct = os.environ.get('RCF_UP2X_WGRAD_DIRECT', '0') == '1'
        self.up2x_wgrad_one_launch = os.environ.get('RCF_UP2X_WGRAD_ONE_LAUNCH', '1') != '0'
        self._up2x_wgrad_ok = {}
        self.s2_dgrad_one_launch = os.environ.get('RCF_S2_DGRAD_ONE_LAUNCH', '1') != '0'
        self._s2_dgrad_ok = {}
        self.s2_wgrad_one_launch = os.environ.get('RCF_S2_WGRAD_ONE_LAUNCH', '1') != '0'
        self._s2_wgrad_ok = {}
        # weight gradients on a side stream (fork after dZ is written, join before the optimizer / a gradient bucket's exchange): a
        # weight gradient is off the backward's critical path, and the BatchNorm-backward passes it then overlaps are HBM-bound kernels
        # that leave board power unused while the convolution kernels run AT the power limit
        single = os.environ.get('RCF_SINGLE_STREAM', '0') == '1'   # everything on the caller's stream (profiles: a kernel's time is its own)
        self.wgrad_side = os.environ.get('RCF_WGRAD_SIDE_STREAM', '1') != '0' and not single
        self._side = None
        self._side_busy = False
        self._side_keep = []
        self._open_switches = []       # stream switches entered and not yet left (recover() undoes them after an exception)
        self.pw_s2_dgrad_lowres = os.environ.get('RCF_PW_S2_DGRAD', '1') != '0'
        self._pw_s2_ok = {}
        self.amax_chunk = 512          # slots per zero-filled chunk of the per-step maxima arena (two-plane fp16 arithmetic)
        self.amax_arena_cap = None     # tests only: never size the arena above this many slots
        self.completes_bucket = None   # callable(parameter) -> bool (data parallelism): this gradient finishes a bucket
        # the encoder's depth branch on its own stream (forward and backward): independent of the image branch up to each level's
        # fusion (RCF_BRANCH_STREAM=0: on the main stream).  fp32 +2.3 %, bf16 +2.9 % on top of the weight-gradient stream
        self.branch_stream = os.environ.get('RCF_BRANCH_STREAM', '1') != '0' and not single
        self._branch = None
        self._branch_busy = False
        self._in_branch = False
        self.in_backward = False   # set by the model around Engine.backward
        self._tape_main = False    # closures recorded now replay on the main stream even while the forward is on the branch stream
        # forward: each level's fusion (two 1x1 convolutions + the gate pass: HBM-bound, needed only by the decoder) on the branch
        # stream too, beside the next level's blocks (RCF_FUSE_ON_BRANCH=0: on the main stream)
        self.fuse_on_branch = os.environ.get('RCF_FUSE_ON_BRANCH', '1') != '0' and self.branch_stream
        self.fuse_wp_one_pass = os.environ.get('RCF_FUSE_WP_ONE_PASS', '1') != '0'   # inference: the fusion in one kernel (bf16 tensors)
        self.prof = None           # optional KernelTimer: brackets conv launches with events on the launch stream
        self.use_phase_convs = True  # exact-2x UpConv and stride-2 dgrad as 2x2 phase convs (False: 9-tap / zero-insert forms)
        # the four forward phases of an up-2x conv in ONE launch (rcf_conv_desc.phase_sum == 2).  Round 5: the kernels stage x once per
        # channel chunk and run all four phases from that tile (bf16 tensors: conv_b16_kernel; fp32 tensors on two fp16 planes:
        # conv_split_kernel) -- a quarter of the loads (and, for fp32, of the plane conversions) of the four launches.  On for those two;
        # the exact three-plane tier keeps the four launches (its one-launch form re-stages the tile per phase: measured slower in
        # round 4).  RCF_UP2X_ONE_LAUNCH=0/1 overrides
        e = os.environ.get('RCF_UP2X_ONE_LAUNCH')
        self.up2x_one_launch = None if e is None else (e != '0')
        # BatchNorm + lrelu applied by the consuming split conv / wgrad kernels as they load z (rcf_conv2d_fwd_bn): saves the
        # bn_act_fwd pass (-2.4 ms/step) but costs the matrix kernels more in their staging path (+3.6 ms measured): off.  The
        # output head, an HBM-bound kernel with idle VALU, always consumes its input this way.
        self.bn_on_load = False
        # BatchNorm-backward sums in the epilogue of the input-gradient kernel that writes dY (RCF_BN_SUMS_IN_DGRAD=0: always the
        # separate reduction pass)
        self.bn_sums_in_dgrad = os.environ.get('RCF_BN_SUMS_IN_DGRAD', '1') != '0'
        self.bn_sums_taken = 0   # launches that took the sums since construction (tests; a counter, nothing reads it back)
        # inference (eval mode, no tape): BatchNorm folded into the conv weights, bias + LeakyReLU (+ residual) in the split kernels'
        # epilogue (rcf_conv2d_fwd_act): no z tensor and no BN pass for those layers
        self.fuse_eval = True
        # 3x3 stride-2 weight gradients as four 2x2 phase weight gradients on the bf16 matrix pipe (16 taps computed for 9 used):
        # 3.5x faster than the register-staged f32-MFMA kernel with bf16 tensors, 1 % slower than the LDS-DMA f32-MFMA kernel with
        # fp32 tensors (measured) -- so: None = only in the bf16 configuration; RCF_S2_WGRAD_PHASES=0/1 overrides
        e = os.environ.get('RCF_S2_WGRAD_PHASES')
        self.s2_wgrad_phases = None if e is None else (e != '0')
        if os.environ.get('RCF_BN_ON_LOAD') is not None:
            self.bn_on_load = os.environ['RCF_BN_ON_LOAD'] == '1'
        # two-plane fp16 arithmetic (RCF_PREC_F16X2): per-step arena of activation / gradient maxima, per-step cache of weight maxima
        self._amax_arena = None
        self._amax_used = 0
        self._amax_total = 0           # slots handed out in this step / the most any step needed (the next arena's size)
        self._amax_need = 0
        self._main = None              # the caller's stream, noted at the outermost stream switch
        self._wamax_cache = {}
        self._scope = None
        # inference with frozen weights (FusionNetModel.capture_inference(fold_once=True)): a dict that keeps every weight transform of
        # the eval forward (BatchNorm coefficients, folded / packed / phase weights) from the first forward on, so that later forwards
        # -- the recorded one -- launch none of them.  None: transforms are recomputed on every forward (parameters may change).
        self.frozen = None

    # ------------------------------------------------------------------ helpers
    def _new(self, shape, ref):
        '''An NHWC activation / gradient tensor: fp32, or bf16 in the bf16-storage configuration (ops.set_precision('bf16')).'''
        return torch.empty(shape, dtype=ops.act_dtype(), device=ref.device)

    @staticmethod
    def _newf(shape, ref):
        '''An fp32 buffer whatever the activation storage is: packed weights, coefficients, workspaces, single-channel maps.'''
        return torch.empty(shape, dtype=torch.float32, device=ref.device)


    def _set_scope(self, scope):
        '''Which part of the network the following launches belong to ('encoder' / 'decoder'): profiling tags only.'''
        self._scope = scope
        if self.prof is not None:
            self.prof.scope = scope

    # ---- per-tensor maxima of the two-plane fp16 arithmetic (RCF_PREC_F16X2; rcf_common.h)
    @staticmethod
    def _f16():
        return ops.get_precision() == RCF_PREC_F16X2

    def _begin_step_scales(self, ref):
        self._wamax_cache = {}
        self._amax_used = 0
        # sized from what the previous step of this engine used, so that a deep net (fusionnet34) overflows in its first step only
        self._amax_need = max(self._amax_need, self._amax_total)
        self._amax_total = 0
        n = max(self.amax_chunk, self._amax_need)
        if self.amax_arena_cap is not None:     # tests: a small cap makes every step run through the mid-step overflow path
            n = min(n, self.amax_arena_cap)
        self._amax_arena = torch.zeros(n, dtype=torch.float32, device=ref.device) if self._f16() else None
        self.plan.fence = self._fence_streams

    def _fence_streams(self):
        """A buffer was just zero-filled on the CURRENT stream and its slots are about to be handed to kernels on any of the engine's
        streams (atomicMax into a zeroed scalar): the other streams wait for the fill, or it could land after their maxima."""
        cur = torch.cuda.current_stream()
        if not self._open_switches:
            self._main = cur      # no switch open: the current stream IS the step's main stream (not a stale one from an earlier step)
        ev = cur.record_event()
        for s in (self._main, self._branch, self._side):
            if s is not None and s != cur:
                s.wait_event(ev)
        # a stream that was made to wait has pending work from the main stream's point of view: side_join() must rejoin it (inside
        # a hipGraph capture an unjoined stream fails capture_end)
        if self._branch is not None and self._branch != cur:
            self._branch_busy = True
        if self._side is not None and self._side != cur:
            self._side_busy = True

    def _amax_slot(self):
        '''A zeroed device scalar for the maximum of a tensor this step produces (None outside the fp16 arithmetic).'''
        if self._amax_arena is None or not self._f16():
            return None
        if self._amax_used >= self._amax_arena.numel():
            # a deeper net, or a second forward before the backward: another zeroed chunk (the full one stays alive through the Acts
            # that hold views of it)
            n = self.amax_chunk if self.amax_arena_cap is None else min(self.amax_chunk, self.amax_arena_cap)
            self._amax_arena = torch.zeros(n, dtype=torch.float32, device=self._amax_arena.device)
            self._amax_used = 0
            self._fence_streams()
        self._amax_used += 1
        self._amax_total += 1
        return self._amax_arena[self._amax_used - 1:self._amax_used]

    def _w_amax(self, w):
        '''max|w| of a weight tensor or phase-weight buffer as a device scalar, computed once per step (through the weight plan when
        it is active: one batched launch up front).'''
        key = (w.data_ptr(), w.numel())
        hit = self._wamax_cache.get(key)
        if hit is not None:
            return hit
        plan = self.plan
        out = None
        if self._plan_on() and not plan.dirty:
            if plan.state == 'replay':
                e = plan._next('amax')
                if e is not None and e['ptr'] == w.data_ptr() and e['n'] == w.numel():
                    out = e['out']
                else:
                    plan.dirty = True
            elif plan.state == 'record':
                out = plan.new_wamax_slot(w)
                ops.amax(w, out, accumulate=True)   # the slot is zero
                plan.entries.append({'kind': 'amax', 'w': w, 'ptr': w.data_ptr(), 'n': w.numel(), 'out': out})
        if out is None:
            out = ops.amax(w)
        self._wamax_cache[key] = out
        return out

    @staticmethod
    def _exact_unless(desc, *amaxes):
        '''A descriptor built under RCF_PREC_F16X2 keeps that arithmetic only when every operand's maximum is known; otherwise it
        is (a copy) on the exact three-plane arithmetic.  Returns the descriptor to use.'''
        if desc.precision == RCF_PREC_F16X2 and any(a is None for a in amaxes):
            desc = ConvDesc.from_buffer_copy(bytes(desc))
            desc.precision = RCF_PREC_FP32
        return desc

    def _frozen_get(self, key, make):
        '''make() once per key while self.frozen is set (inference with frozen weights), every time otherwise.'''
        if self.frozen is None or self.tape is not None:
            return make()
        hit = self.frozen.get(key)
        if hit is None:
            hit = self.frozen[key] = make()
        return hit

    # ---- weight transforms (through the step's WeightPlan when it is active: training, tape on)
    def _plan_on(self):
        return self.plan.active

    def _phase_w(self, w, mode):
        '''ops.phase_weights(w, mode): [4][O'][I'][2][2].'''
        plan = self.plan
        if not self._plan_on() or plan.dirty:
            return ops.phase_weights(w, mode)
        if plan.state == 'record':
            out = ops.phase_weights(w, mode)
            plan.entries.append({'kind': 'phase', 'w': w, 'mode': mode, 'out': out, 'ptr': w.data_ptr()})
            return out
        e = plan._next('phase')
        if e is None or e['ptr'] != w.data_ptr() or e['mode'] != mode or tuple(e['w'].shape) != tuple(w.shape):
            plan.dirty = True
            return ops.phase_weights(w, mode)
        return e['out']

    def _pack(self, desc, w, like, amax=None):
        '''A packed-weight buffer holding ops.conv_pack(desc, w, ., amax).'''
        return self._pack_n(desc, [w], like, amax)

    def _pack_n(self, desc, ws, like, amax=None):
        '''len(ws) packings under one descriptor, back to back in one buffer (the four phases of a merged up-2x input gradient).
        amax: the weights' maximum (two-plane fp16 descriptors on the split kernels), common to all of ws.'''
        plan = self.plan
        key = bytes(desc)
        if self._plan_on() and not plan.dirty and plan.state == 'replay':
            e = plan._next('pack')
            if (e is not None and e['key'] == key and e['ptrs'] == [w.data_ptr() for w in ws]
                    and (None if e['amax'] is None else e['amax'].data_ptr()) == (None if amax is None else amax.data_ptr())):
                return e['out']
            if e is not None:
                plan.last_mismatch = (plan.pos - 1, 'pack', 'descriptor' if e['key'] != key else 'source pointer')
            plan.dirty = True
        if self.frozen is not None and self.tape is None:
            fkey = ('pack', key, tuple(w.data_ptr() for w in ws))
            hit = self.frozen.get(fkey)
            if hit is not None:
                return hit[0]
        nf = ops.conv_query(desc).packed_weight_floats
        out = self._newf((len(ws) * nf,), like)
        dsts = [out[k * nf:(k + 1) * nf] for k in range(len(ws))]
        for w, dst in zip(ws, dsts):
            ops.conv_pack(desc, w, dst, amax)
        if self.frozen is not None and self.tape is None:
            self.frozen[fkey] = (out, list(ws))   # the sources stay alive with the buffer packed from them
        if self._plan_on() and not plan.dirty and plan.state == 'record':
            plan.entries.append({'kind': 'pack', 'desc': desc, 'key': key, 'srcs': list(ws), 'ptrs': [w.data_ptr() for w in ws],
                                 'dsts': dsts, 'out': out, 'amax': amax})
        return out

    @staticmethod
    def _two_plane(kernel_id):
        '''rcf_conv_info.kernel_id of a forward / input-gradient split kernel on two fp16 planes (ids 40000 .. 59999; the bf16-operand
        variants end below 40000).'''
        return 40000 <= kernel_id < 60000

    @staticmethod
    def _two_plane_wgrad(kernel_id):
        '''rcf_conv_info.wgrad_kernel_id of a split weight-gradient kernel on two fp16 planes (10000 + 40000 + ...).'''
        return kernel_id >= 50000

    def _mat(self, x):
        '''The activation tensor of x; a deferred one (raw conv output + BN coefficients) is materialised once, on demand.'''
        if x.t is None:
            z = x.z
            x.t = torch.empty_like(z)
            x.amax = self._amax_slot() if z.dtype == torch.float32 else None
            ops.bn_act_fwd(z, x.coef, None, x.t, z.shape[0] * z.shape[1] * z.shape[2], z.shape[3], RCF_ACT_LEAKY_RELU, amax=x.amax)
        return x.t

    @staticmethod
    def _src(x, on_load_ok):
        '''(tensor, coef) to hand to a conv kernel: the raw output + coefficients when the kernel applies BN on load.'''
        if x is None:
            return None, None
        if x.t is None and on_load_ok:
            return x.z, x.coef
        return None, None   # caller materialises

    @staticmethod
    def _shape(x):
        return (x.t if x.t is not None else x.z).shape

    def _side_enter(self, *keep):
        '''Switch to the side stream for a weight gradient (None when the switch is off).  keep: tensors the side-stream kernels read
        that the main stream's Python flow would free before they ran -- held until side_join().'''
        if not self.wgrad_side or self.bn_on_load:
            return None
        if self._side is None:
            self._side = torch.cuda.Stream()
        self._side.wait_stream(torch.cuda.current_stream())
        self._side_keep.extend(t for t in keep if t is not None)
        self._side_busy = True
        return self._switch(self._side)

    def _switch(self, stream):
        """Make `stream` current until _unswitch(ctx).  Open switches are kept on a stack so that recover() can undo them when an
        exception (RcfError, out of memory) leaves a switched region half-way."""
        if not self._open_switches:
            self._main = torch.cuda.current_stream()
        ctx = torch.cuda.stream(stream)
        ctx.__enter__()
        self._open_switches.append(ctx)
        return ctx

    def _unswitch(self, ctx):
        ctx.__exit__(None, None, None)
        if ctx in self._open_switches:
            self._open_switches.remove(ctx)

    def _side_exit(self, ctx):
        if ctx is not None:
            self._unswitch(ctx)

    def recover(self):
        """After an exception inside forward / backward: back to the caller's stream, the side streams joined, no flag left set --
        a caller that catches and carries on (bench.py falling back from a failed capture to the eager step) runs on a sane engine."""
        while self._open_switches:
            self._open_switches.pop().__exit__(None, None, None)
        self._in_branch = False
        self._tape_main = False
        self.in_backward = False
        self.tape = None
        try:
            self.side_join()
        except RuntimeError:      # (a failed capture: the streams cannot be waited on; their work is gone with the capture)
            self._branch_busy = self._side_busy = False
            del self._side_keep[:]

    def side_join(self):
        '''The main stream waits for every weight gradient enqueued on the side stream (and for the depth branch's stream); the tensors
        they read may go.'''
        if self._branch_busy:
            torch.cuda.current_stream().wait_stream(self._branch)
            self._branch_busy = False
        if self._side_busy:
            torch.cuda.current_stream().wait_stream(self._side)
            self._side_busy = False
        del self._side_keep[:]

    def _branch_enter(self, first_wait):
        '''Run what follows on the depth branch's stream (None: switch off, or already there).  first_wait: the branch stream first waits
        for the main stream (the step's inputs and weight plan; the fusion's hand-off in backward) -- not on every entry, which would
        queue each depth block behind the image block enqueued just before it.'''
        if not self.branch_stream or self.bn_on_load or self._in_branch or (self.in_backward and self.on_param_grad is not None):
            # (under data parallelism the backward stays on the main stream: a bucket's exchange is launched from it.  bn_on_load: a
            # deferred activation is materialised by whichever consumer comes first -- the fusion on the branch stream, then max_pool
            # on the main stream reads it with no event between them -- so that mode runs on one stream, like _side_enter)
            return None
        if self.tape is None and not self.in_backward:
            # forward only (inference, validation): no tape keeps the activations alive, so every tensor that crosses streams needs
            # record_stream and the allocator's deferred frees -- measured 10 % SLOWER than one stream for eager bf16 inference
            # (1492 vs 1666 samples/s); the captured inference graph serialises the branches anyway
            return None
        if self._branch is None:
            self._branch = torch.cuda.Stream()
        if first_wait:
            self._branch.wait_stream(torch.cuda.current_stream())
        self._branch_busy = True
        self._in_branch = True
        return self._switch(self._branch)

    def _branch_exit(self, ctx):
        if ctx is not None:
            self._unswitch(ctx)
            self._in_branch = False

    def _branch_wait(self, dep=None):
        '''The current (main) stream waits for the depth branch: its activation `dep` is about to be read (fusion).  The tensor was
        allocated on the branch's stream: without a tape holding it (inference) it is freed as soon as the branch moves on, and the
        allocator would hand its memory to the branch's next kernel while the fusion on the main stream still reads it --
        record_stream defers that reuse behind the main stream's work.'''
        if self._branch_busy and not self._in_branch:
            cur = torch.cuda.current_stream()
            cur.wait_stream(self._branch)
            if dep is not None:
                for t in (dep.t, dep.z):
                    if t is not None:
                        t.record_stream(cur)

    def _wgrad_done(self, *params):
        if self.on_param_grad is not None:
            for p in params:
                if (self._side_busy or self._branch_busy) and not self._in_branch and (self.completes_bucket is None or self.completes_bucket(p)):
                    self.side_join()   # the bucket's exchange reads gradients the side streams may still be writing
                self.on_param_grad(p)

    def _conv(self, layer, x, x2=None, up_hw=None, want_stats=False, fold=None):
        '''conv (+ folded nearest-upsample of x to up_hw, + folded channel concat with x2) -> raw output z.
        fold = (coef, res tensor or None): inference -- BatchNorm scale folded into the weights, bias + LeakyReLU (+ residual) in the
        kernel's epilogue where it has one (info.fwd_act); the returned tensor is then the ACTIVATION and the 5th result True.'''
        if x.s2d is not None and layer.kernel_size == 7 and layer.stride == 2 and x2 is None and up_hw is None:
            return self._conv_stem_s2d(layer, x, want_stats, fold)
        n, h, w, c1 = self._shape(x)
        c2 = 0 if x2 is None else self._shape(x2)[3]
        h_in, w_in, gather = h, w, RCF_GATHER_DIRECT
        if up_hw is not None and (int(up_hw[0]), int(up_hw[1])) != (h, w):
            h_in, w_in, gather = int(up_hw[0]), int(up_hw[1]), RCF_GATHER_NEAREST
        if x2 is not None and tuple(self._shape(x2)[1:3]) != (h_in, w_in):
            raise ValueError('skip connection and upsampled tensor disagree in size')
        weight = layer.conv.weight
        if getattr(layer, 'deconv', None) is not None and weight.dim() == 4 and isinstance(layer.deconv, torch.nn.ConvTranspose2d):
            if x2 is not None or up_hw is not None:
                raise ValueError('a transposed convolution takes one source at its own resolution')
            return self._conv_transpose(layer, x, want_stats, fold)
        if (self.use_phase_convs and gather == RCF_GATHER_NEAREST and x2 is None and layer.kernel_size == 3
                and layer.stride == 1 and (h_in, w_in) == (2 * h, 2 * w) and c1 % 4 == 0):
            return self._conv_up2x(layer, x, want_stats, fold)
        desc = ops.make_fwd_desc(n, h_in, w_in, c1, c2, weight.shape[0], layer.kernel_size, layer.stride, h, w, gather)
        if desc.precision == RCF_PREC_F16X2:
            # two fp16 planes need the operands' maxima: materialise deferred activations now (their producer pass supplies it); an
            # operand without one (network input, a tensor from the inference epilogue) or the folded inference weights keep the layer
            # on the exact three-plane arithmetic
            self._mat(x)
            if x2 is not None:
                self._mat(x2)
            desc = self._exact_unless(desc, x.amax, None if fold is not None else 0, 0 if x2 is None else x2.amax)
        info = ops.conv_query(desc)
        t1, k1 = self._src(x, info.bn_on_load)
        t2, k2 = self._src(x2, info.bn_on_load)
        if t1 is None:
            t1 = self._mat(x)
        if x2 is not None and t2 is None:
            t2 = self._mat(x2)
        fused = fold is not None and info.fwd_act and k1 is None and k2 is None
        scales = None
        if fused:
            def fold_pack():
                buf = self._newf((info.packed_weight_floats,), t1)
                ops.conv_pack(desc, ops.scale_channels(weight.detach(), fold[0][0]), buf)
                return buf
            packed = self._frozen_get(('fold', id(layer), bytes(desc)), fold_pack)
        elif self._two_plane(info.kernel_id):
            wmax = self._w_amax(weight.detach())
            packed = self._pack(desc, weight.detach(), t1, wmax)
            scales = ops.make_scales(x.amax, None if x2 is None else x2.amax, wmax)
        else:
            packed = self._pack(desc, weight.detach(), t1)
        z = self._new((n, desc.h_out, desc.w_out, desc.c_out), t1)
        partials = torch.empty((info.n_partials, 2, desc.c_out), dtype=torch.float64, device=t1.device) if want_stats else None
        if self.prof is not None:
            self.prof.begin(info.kernel_id, ops.algorithmic_flops(desc), desc)
        if fused:
            ops.conv_fwd_act(desc, t1, t2, packed, fold[0][1], fold[1], z)
        else:
            ops.conv_fwd(desc, t1, t2, packed, z, partials, coef1=k1, coef2=k2, scales=scales)
        if self.prof is not None:
            self.prof.end()
        if self.kernel_log is not None and desc is not None:
            self.kernel_log.append((desc.ksize, desc.stride, desc.c1 + desc.c2, desc.c_out, desc.h_out, desc.w_out,
                                    info.kernel_id))
        if fold is not None:
            return z, desc, info, partials, fused
        return z, desc, info, partials

    def _run_packed(self, desc, w_oihw, in1, out, partials=None, coef1=None, bias=None, amax_in=None, amax_w=None):
        '''amax_in / amax_w: the maxima of in1 and of the weight BUFFER w_oihw is a slice of (two-plane fp16 descriptors; the
        caller has put the descriptor on the exact arithmetic when one of them is unknown).'''
        info = ops.conv_query(desc)
        scales = None
        if self._two_plane(info.kernel_id):
            packed = self._pack(desc, w_oihw, in1, amax_w)
            scales = ops.make_scales(amax_in, None, amax_w)
        else:
            packed = self._pack(desc, w_oihw, in1)
        if self.prof is not None:
            self.prof.begin(info.kernel_id, ops.algorithmic_flops(desc), desc)
        if bias is not None:
            ops.conv_fwd_act(desc, in1, None, packed, bias, None, out)
        else:
            ops.conv_fwd(desc, in1, None, packed, out, partials, coef1=coef1, scales=scales)
        if self.prof is not None:
            self.prof.end()
        return info

    def _conv_up2x(self, layer, x, want_stats, fold=None):
        '''
        conv3x3(F.interpolate(x, 2x nearest)) as four 2x2 phase convolutions on x (4/9 of the MACs; the weights of the
        taps that hit the same source pixel are pre-summed, so results differ from the 9-tap form by fp32 round-off only).
        '''
        n, h, w, c1 = self._shape(x)
        weight = layer.conv.weight
        co = weight.shape[0]
        descs, partials, n_part = [], None, 0
        t1 = k1 = z = wmax = None
        fused = False
        f16 = self._f16()
        if f16:
            self._mat(x)   # two fp16 planes need max|x|: a deferred activation is materialised (its producer pass supplies it)
        # the four phases in ONE launch (rcf_conv_desc.phase_sum == 2) where the layer runs on a split / DMA kernel: a workgroup runs the
        # four 2x2 convolutions of its tile back to back, so x comes from HBM once (the three re-reads hit L2) and three launches go;
        # the backward pass keeps the four per-phase descriptors (their weight gradients are four launches)
        if self.up2x_one_launch is not False:
            dm = ops.make_up2x_fwd_desc(n, h, w, c1, co, 0, 0, phase_out=True)
            if f16:
                dm = self._exact_unless(dm, x.amax, None if fold is not None else 0)
            qm = None
            if self.up2x_one_launch or ops.act_dtype() == torch.bfloat16 or dm.precision == RCF_PREC_F16X2:
                try:
                    qm = ops.conv_query(dm)
                except ops._lib.RcfUnsupported:   # no kernel for this form (any other error -- a launch failure -- propagates)
                    qm = None
            if qm is not None:
                return self._conv_up2x_merged(layer, x, dm, qm, want_stats, fold)
        for ph in range(4):
            d = ops.make_up2x_fwd_desc(n, h, w, c1, co, ph >> 1, ph & 1)
            if f16:
                d = self._exact_unless(d, x.amax, None if fold is not None else 0)
            if ph == 0:
                qi = ops.conv_query(d)
                n_part = qi.n_partials
                fused = fold is not None and bool(qi.fwd_act) and fold[1] is None
                wp = self._frozen_get(('fold-phase', id(layer)), lambda: ops.phase_weights(ops.scale_channels(weight.detach(), fold[0][0]),
                                                                                              RCF_PHASE_UP2X_FWD)) if fused else \
                    self._phase_w(weight.detach(), RCF_PHASE_UP2X_FWD)
                if self._two_plane(qi.kernel_id):
                    wmax = self._w_amax(wp)   # one maximum for the four phases' pre-summed weights
                t1, k1 = self._src(x, qi.bn_on_load and not fused)
                if t1 is None:
                    t1 = self._mat(x)
                z = self._new((n, 2 * h, 2 * w, co), t1)
                if want_stats:
                    partials = torch.empty((4 * n_part, 2, co), dtype=torch.float64, device=t1.device)
            self._run_packed(d, wp[ph], t1, z, None if partials is None else partials[ph * n_part:(ph + 1) * n_part], coef1=k1,
                             bias=fold[0][1] if fused else None, amax_in=x.amax, amax_w=wmax)
            descs.append(d)

        class _Info(object):
            pass
        info = _Info()
        info.n_partials = 4 * n_part
        info.up2x = descs
        if fold is not None:
            return z, None, info, partials, fused
        return z, None, info, partials

    def _conv_up2x_merged(self, layer, x, dm, qm, want_stats, fold):
        '''The up-2x forward as one launch over the four output phases (see _conv_up2x); returns what _conv_up2x returns.'''
        n, h, w, c1 = self._shape(x)
        weight = layer.conv.weight
        co = weight.shape[0]
        fused = fold is not None and bool(qm.fwd_act) and fold[1] is None
        wp = self._frozen_get(('fold-phase', id(layer)), lambda: ops.phase_weights(ops.scale_channels(weight.detach(), fold[0][0]),
                                                                                      RCF_PHASE_UP2X_FWD)) if fused else \
            self._phase_w(weight.detach(), RCF_PHASE_UP2X_FWD)
        t1, k1 = self._src(x, qm.bn_on_load and not fused)
        if t1 is None:
            t1 = self._mat(x)
        scales = None
        if self._two_plane(qm.kernel_id):
            wmax = self._w_amax(wp)   # one maximum for the four phases' pre-summed weights
            packed = self._pack_n(dm, [wp[ph] for ph in range(4)], t1, wmax)
            scales = ops.make_scales(x.amax, None, wmax)
        else:
            packed = self._pack_n(dm, [wp[ph] for ph in range(4)], t1)
        z = self._new((n, 2 * h, 2 * w, co), t1)
        partials = torch.empty((qm.n_partials, 2, co), dtype=torch.float64, device=t1.device) if want_stats else None
        if self.prof is not None:
            self.prof.begin(qm.kernel_id, ops.algorithmic_flops(dm), dm)
        if fused:
            ops.conv_fwd_act(dm, t1, None, packed, fold[0][1], None, z)
        else:
            ops.conv_fwd(dm, t1, None, packed, z, partials, coef1=k1, scales=scales)
        if self.prof is not None:
            self.prof.end()
        descs = []
        for ph in range(4):   # the backward pass's per-phase descriptors, under the arithmetic the forward ran on
            d = ops.make_up2x_fwd_desc(n, h, w, c1, co, ph >> 1, ph & 1)
            d.precision = dm.precision
            descs.append(d)

        class _Info(object):
            pass
        info = _Info()
        info.n_partials = qm.n_partials
        info.up2x = descs
        if fold is not None:
            return z, None, info, partials, fused
        return z, None, info, partials

    def _conv_stem_s2d(self, layer, x, want_stats, fold=None):
        '''
        The 7x7 stride-2 stem as a 4x4 stride-1 convolution on the space-to-depth image of the network input (built straight from the
        NCHW input) on the 16-bit matrix pipe -- the f32-MFMA stem kernel is compute bound at ~55 TFLOP/s.  bf16 tensors: the LDS-DMA
        kernel on the bf16 image (ops.s2d_image).  fp32 tensors (round 3): the two-plane fp16 split kernel on the fp32 image, scaled
        by the input's maximum (ops.s2d_image_f32 accumulates it) and the weights' maximum.  The weight gradient (training) is taken
        on the 7x7 form from the fp32 NHWC input as before.
        '''
        n = x.s2d.shape[0]
        h, w = x.hw
        weight = layer.conv.weight
        co = weight.shape[0]
        f32 = x.s2d.dtype == torch.float32
        d = ops.make_stem_s2d_desc(n, h, w, co, f32=f32)
        info = ops.conv_query(d)
        fused = fold is not None and bool(info.fwd_act) and fold[1] is None

        def stem_weights():
            return ops.stem_weights_s2d(ops.scale_channels(weight.detach(), fold[0][0]) if fused else weight.detach())

        scales = None
        if f32:
            # two fp16 planes: the 4x4 weights' maximum is computed on the transformed tensor (one small launch; the stems are two layers)
            def stem_pack32():
                w4 = stem_weights()
                wmax = ops.amax(w4)
                buf = self._newf((info.packed_weight_floats,), x.s2d)
                ops.conv_pack(d, w4, buf, wmax)
                return buf, wmax
            packed, wmax = self._frozen_get(('stem32', id(layer), bool(fused)), stem_pack32)
            scales = ops.make_scales(x.amax, None, wmax)
        else:
            def stem_pack():
                buf = self._newf((info.packed_weight_floats,), x.s2d)
                ops.conv_pack(d, stem_weights(), buf)
                return buf
            packed = self._frozen_get(('stem', id(layer), bool(fused)), stem_pack)
        z = torch.empty((n, d.h_out, d.w_out, co), dtype=x.s2d.dtype, device=x.s2d.device)
        partials = torch.empty((info.n_partials, 2, co), dtype=torch.float64, device=z.device) if want_stats else None
        if self.prof is not None:
            self.prof.begin(info.kernel_id, 2.0 * n * d.h_out * d.w_out * co * 49 * x.t.shape[3] if x.t is not None else 0.0, d)
        if fused:
            if f32:
                raise RuntimeError('the fp32 stem has no fused inference epilogue on two planes')   # fold -> exact arithmetic upstream
            ops.conv_fwd_act(d, x.s2d, None, packed, fold[0][1], None, z)
        else:
            ops.conv_fwd(d, x.s2d, None, packed, z, partials, scales=scales)
        if self.prof is not None:
            self.prof.end()

        class _Info(object):
            pass
        sinfo = _Info()
        sinfo.n_partials = info.n_partials
        sinfo.stem = True
        if fold is not None:
            return z, None, sinfo, partials, fused
        return z, None, sinfo, partials

    def _conv_transpose(self, layer, x, want_stats, fold=None):
        '''
        TransposeConv2d's ConvTranspose2d(3, stride 2, padding 1, output_padding 1) (src/net_utils.py:94-153): y (2H x 2W) is the
        input gradient of a VIRTUAL 3x3 stride-2 convolution 2H x 2W -> H x W whose OIHW weight [in][out][3][3] is exactly the
        transposed convolution's weight tensor -- so it runs on the same four 2x2 phase convolutions as the input gradient of the
        encoder's stride-2 convolutions (output phase (a, b) only sees the taps of its parity: 1 + 2 + 2 + 4 = 9 real taps).
        '''
        n, h, w, ci = self._shape(x)
        weight = layer.conv.weight
        co = weight.shape[1]
        if ci % 4 != 0 or co % 4 != 0 or weight.shape[0] != ci:
            raise ValueError('transposed convolution: channel counts must be multiples of 4 and match the input')
        virt = ops.make_fwd_desc(n, 2 * h, 2 * w, co, 0, ci, 3, 2)       # the convolution this one is the transpose of
        t1 = self._mat(x)
        wd = self._phase_w(weight.detach(), RCF_PHASE_S2_DGRAD)
        z = self._new((n, 2 * h, 2 * w, co), t1)
        partials, n_part, wmax = None, 0, None
        for ph in range(4):
            d = self._exact_unless(ops.make_s2_dgrad_desc(virt, ph >> 1, ph & 1, False), x.amax)
            if ph == 0:
                qi = ops.conv_query(d)
                n_part = qi.n_partials
                if self._two_plane(qi.kernel_id):
                    wmax = self._w_amax(wd)
                if want_stats:
                    partials = torch.empty((4 * n_part, 2, co), dtype=torch.float64, device=t1.device)
            self._run_packed(d, wd[ph], t1, z, None if partials is None else partials[ph * n_part:(ph + 1) * n_part],
                             amax_in=x.amax, amax_w=wmax)

        class _Info(object):
            pass
        info = _Info()
        info.n_partials = 4 * n_part
        info.transpose = virt
        if fold is not None:
            return z, None, info, partials, False
        return z, None, info, partials

    def _conv_transpose_backward(self, layer, info, x, dz, dz_amax=None):
        '''dW = the weight gradient of the virtual stride-2 convolution with the roles swapped (its input is dY, its output
        gradient is x) -- already in the [in][out][3][3] layout of ConvTranspose2d.weight; dX = that convolution applied to dY.'''
        virt = self._exact_unless(info.transpose, dz_amax, x.amax)
        weight = layer.conv.weight
        qi = ops.conv_query(virt)
        ws = self._newf((max(1, qi.wgrad_workspace_floats),), dz)
        scales = ops.make_scales(dz_amax, None, None, x.amax) if self._two_plane_wgrad(qi.wgrad_kernel_id) else None
        if self.prof is not None:
            self.prof.begin(qi.wgrad_kernel_id, ops.algorithmic_flops(virt), virt)
        ops.conv_wgrad(virt, dz, None, self._mat(x), self.grad_of(weight), ws, scales=scales)
        if self.prof is not None:
            self.prof.end()
        self._wgrad_done(weight)
        if x.needs_grad:
            acc = x.g is not None
            if acc:
                x.bsum = None
            if not acc:
                x.g = self._new(tuple(self._shape(x)), dz)
            dd = ops.make_fwd_desc(virt.n, virt.h_in, virt.w_in, virt.c1, 0, virt.c_out, 3, 2)
            dd.accumulate = 1 if acc else 0
            dd = self._exact_unless(dd, dz_amax)
            wmax = self._w_amax(weight.detach()) if self._two_plane(ops.conv_query(dd).kernel_id) else None
            self._run_packed(dd, weight.detach(), dz, x.g, amax_in=dz_amax, amax_w=wmax)

    def _s2_dgrad_merged(self, fwd, wd, dz, dx, acc, dz_amax):
        '''The input gradient of a 3x3 stride-2 convolution as ONE launch over its four output phases (rcf_conv_desc.phase_sum == 3: dz
        staged once per channel chunk, only the 1 + 2 + 2 + 4 taps that exist -- the four per-phase launches multiply the other seven
        by zero weights).  bf16 tensors and fp32 tensors on two fp16 planes; returns False elsewhere (remembered per shape) and the
        caller runs the four launches.  RCF_S2_DGRAD_ONE_LAUNCH=0 switches it off.'''
        if not self.s2_dgrad_one_launch:
            return False
        dm = self._exact_unless(ops.make_s2_dgrad_desc(fwd, 0, 0, acc, phase_out=True), dz_amax)
        key = bytes(dm)
        if self._s2_dgrad_ok.get(key) is False:
            return False
        try:
            qm = ops.conv_query(dm)
        except ops._lib.RcfUnsupported:   # no kernel for this form (any other error -- a launch failure -- propagates)
            self._s2_dgrad_ok[key] = False
            return False
        self._s2_dgrad_ok[key] = True
        scales = None
        if self._two_plane(qm.kernel_id):
            wmax = self._w_amax(wd)
            packed = self._pack_n(dm, [wd[ph] for ph in range(4)], dz, wmax)
            scales = ops.make_scales(dz_amax, None, wmax)
        else:
            packed = self._pack_n(dm, [wd[ph] for ph in range(4)], dz)
        if self.prof is not None:
            self.prof.begin(qm.kernel_id, ops.algorithmic_flops(dm), dm)
        ops.conv_fwd(dm, dz, None, packed, dx, None, scales=scales)
        if self.prof is not None:
            self.prof.end()
        return True

    def _s2_wgrad_merged(self, fwd, t1, dz, dwp, x_amax, dz_amax):
        '''The four phase weight gradients of a 3x3 stride-2 convolution from ONE launch (rcf_conv2d_wgrad on the phase_sum == 1
        descriptor; the four phases of a tile share an XCD and with it the dz tile).  False where the library has no such launch
        (remembered per shape): the caller runs the four per-phase calls.  RCF_S2_WGRAD_ONE_LAUNCH=0 switches it off.'''
        if not self.s2_wgrad_one_launch:
            return False
        dm = self._exact_unless(ops.make_s2_wgrad_desc(fwd, 0, 0, all_phases=True), x_amax, dz_amax)
        key = bytes(dm)
        if self._s2_wgrad_ok.get(key) is False:
            return False
        try:
            qm = ops.conv_query(dm)
            ws = self._newf((max(1, qm.wgrad_workspace_floats),), dz)
            scales = ops.make_scales(x_amax, None, None, dz_amax) if self._two_plane_wgrad(qm.wgrad_kernel_id) else None
            if self.prof is not None:
                self.prof.begin(qm.wgrad_kernel_id, 4.0 * ops.algorithmic_flops(ops.make_s2_wgrad_desc(fwd, 0, 0)), dm)
            try:
                ops.conv_wgrad(dm, t1, None, dz, dwp, ws, scales=scales)
            finally:
                if self.prof is not None:
                    self.prof.end()
        except ops._lib.RcfUnsupported:   # no kernel for this form (any other error -- a launch failure -- propagates)
            self._s2_wgrad_ok[key] = False
            return False
        self._s2_wgrad_ok[key] = True
        return True

    def _up2x_wgrad_merged(self, info, x, dz, dz_amax, dwp):
        '''The four phase weight gradients of an up-2x convolution from ONE launch (rcf_conv2d_wgrad on the phase_sum == 2 descriptor:
        split weight-gradient kernels only; the four phases of a tile share an XCD, so x is fetched from HBM once).  Returns False where
        the library has no such launch for this layer (remembered per shape) -- the caller then runs the four per-phase calls.
        RCF_UP2X_WGRAD_ONE_LAUNCH=0 switches it off.'''
        if not self.up2x_wgrad_one_launch or self.bn_on_load:   # (BatchNorm-on-load sources keep the per-phase calls)
            return False
        n, h, w, c1 = self._shape(x)
        co = dwp.shape[1]
        d0 = self._exact_unless(info.up2x[0], x.amax, dz_amax)
        key = (n, h, w, c1, co, d0.precision, d0.storage)
        if self._up2x_wgrad_ok.get(key) is False:
            return False
        dm = ops.make_up2x_fwd_desc(n, h, w, c1, co, 0, 0, phase_out=True)
        dm.precision = d0.precision
        try:
            qm = ops.conv_query(dm)
            ws = self._newf((max(1, qm.wgrad_workspace_floats),), dz)
            scales = ops.make_scales(x.amax, None, None, dz_amax) if self._two_plane_wgrad(qm.wgrad_kernel_id) else None
            if self.prof is not None:
                self.prof.begin(qm.wgrad_kernel_id, 4.0 * ops.algorithmic_flops(info.up2x[0]), dm)
            try:
                ops.conv_wgrad(dm, self._mat(x), None, dz, dwp, ws, scales=scales)
            finally:
                if self.prof is not None:
                    self.prof.end()
        except ops._lib.RcfUnsupported:   # no kernel for this form (any other error -- a launch failure -- propagates)
            self._up2x_wgrad_ok[key] = False
            return False
        self._up2x_wgrad_ok[key] = True
        return True

    def _conv_up2x_backward(self, layer, info, x, dz, dz_amax=None):
        n, h, w, c1 = self._shape(x)
        weight = layer.conv.weight
        co = weight.shape[0]
        side = None
        if self.wgrad_side and not self.bn_on_load:
            self._mat(x)
            side = self._side_enter(dz, x.t)
        if self.up2x_wgrad_direct:
            # A/B switch (RCF_UP2X_WGRAD_DIRECT=1): the weight gradient as ONE 3x3 weight gradient at the upsampled resolution with the
            # nearest gather (2.25x the MFMA work of the four 2x2 phases, x and dz staged once) -- measured, not faster: DESIGN Appendix A
            d3 = self._exact_unless(ops.make_fwd_desc(n, 2 * h, 2 * w, c1, 0, co, 3, 1, h, w, RCF_GATHER_NEAREST), x.amax, dz_amax)
            q3 = ops.conv_query(d3)
            ws = self._newf((max(1, q3.wgrad_workspace_floats),), dz)
            scales = ops.make_scales(x.amax, None, None, dz_amax) if self._two_plane_wgrad(q3.wgrad_kernel_id) else None
            ops.conv_wgrad(d3, self._mat(x), None, dz, self.grad_of(weight), ws, scales=scales)
            info = None
        dwp = self._newf((4, co, c1, 2, 2), dz) if info is not None else None
        if info is not None and self._up2x_wgrad_merged(info, x, dz, dz_amax, dwp):
            info = None   # the four phases' weight gradients came from one launch
        for ph, d in enumerate(info.up2x if info is not None else ()):
            d = self._exact_unless(d, x.amax, dz_amax)
            qi = ops.conv_query(d)
            ws = self._newf((max(1, qi.wgrad_workspace_floats),), dz)
            t1, k1 = self._src(x, qi.wgrad_bn_on_load)
            if t1 is None:
                t1 = self._mat(x)
            scales = ops.make_scales(x.amax, None, None, dz_amax) if self._two_plane_wgrad(qi.wgrad_kernel_id) else None
            if self.prof is not None:
                self.prof.begin(qi.wgrad_kernel_id, ops.algorithmic_flops(d), d)
            ops.conv_wgrad(d, t1, None, dz, dwp[ph], ws, coef1=k1, scales=scales)
            if self.prof is not None:
                self.prof.end()
        if dwp is not None:
            ops.phase_wgrad_fold(dwp, self.grad_of(weight))
        self._side_exit(side)
        self._wgrad_done(weight)
        if x.needs_grad:
            wd = self._phase_w(weight.detach(), RCF_PHASE_UP2X_DGRAD)
            acc = x.g is not None
            if acc:
                x.bsum = None
            if not acc:
                x.g = self._new(tuple(self._shape(x)), dz)
            dd = self._exact_unless(ops.make_up2x_dgrad_desc(n, h, w, c1, co, 0, 0, acc, phase_sum=True), dz_amax)
            qi = ops.conv_query(dd)
            scales = None
            if self._two_plane(qi.kernel_id):
                wmax = self._w_amax(wd)
                packed = self._pack_n(dd, [wd[ph] for ph in range(4)], dz, wmax)
                scales = ops.make_scales(dz_amax, None, wmax)
            else:
                packed = self._pack_n(dd, [wd[ph] for ph in range(4)], dz)
            if self.prof is not None:
                self.prof.begin(qi.kernel_id, ops.algorithmic_flops(dd), dd)
            if not self._bn_sums_launch(dd, qi, dz, packed, x.g, scales, None if acc else x):
                ops.conv_fwd(dd, dz, None, packed, x.g, None, scales=scales)
            if self.prof is not None:
                self.prof.end()

    def _conv_backward(self, layer, desc, info, x, x2, dz, dz_amax=None):
        '''dW (written once into the parameter's gradient) and dX / dX2 (accumulated into the producers' .g).  dz_amax: device
        scalar with max|dz| from the kernel that wrote dz (two-plane fp16 arithmetic; None: these layers run exact).'''
        if desc is None and hasattr(info, 'stem'):
            # the stem ran on the space-to-depth image; its weight gradient is the 7x7 one on the fp32 NHWC input (no input gradient)
            n, h, w, c = x.t.shape
            desc = ops.make_fwd_desc(n, h, w, c, 0, layer.conv.weight.shape[0], 7, 2)
            info = ops.conv_query(desc)
        if desc is None and hasattr(info, 'transpose'):
            return self._conv_transpose_backward(layer, info, x, dz, dz_amax)
        if desc is None:
            return self._conv_up2x_backward(layer, info, x, dz, dz_amax)
        weight = layer.conv.weight
        dw = self.grad_of(weight)
        x_amax = x.amax
        x2_amax = 0 if x2 is None else x2.amax
        s2_phases = self.s2_wgrad_phases if self.s2_wgrad_phases is not None else (ops.act_dtype() == torch.bfloat16 or ops.get_precision() == 2)
        side = None
        if self.wgrad_side and not self.bn_on_load:
            self._mat(x)
            if x2 is not None:
                self._mat(x2)
            side = self._side_enter(dz, x.t, None if x2 is None else x2.t)
        if (self.use_phase_convs and s2_phases and desc.stride == 2 and desc.ksize == 3 and x2 is None and desc.c1 % 4 == 0
                and desc.c1 >= 16 and desc.gather1 == RCF_GATHER_DIRECT):
            # 3x3 stride-2 weight gradient as four 2x2 weight gradients on the phase images of x (bf16 matrix pipe)
            t1 = self._mat(x)
            dwp = self._newf((4, desc.c_out, desc.c1, 2, 2), dz)
            for ph in (() if self._s2_wgrad_merged(desc, t1, dz, dwp, x_amax, dz_amax) else range(4)):
                d = self._exact_unless(ops.make_s2_wgrad_desc(desc, ph >> 1, ph & 1), x_amax, dz_amax)
                qi = ops.conv_query(d)
                wsp = self._newf((max(1, qi.wgrad_workspace_floats),), dz)
                scales = ops.make_scales(x_amax, None, None, dz_amax) if self._two_plane_wgrad(qi.wgrad_kernel_id) else None
                if self.prof is not None:
                    self.prof.begin(qi.wgrad_kernel_id, ops.algorithmic_flops(d), d)
                ops.conv_wgrad(d, t1, None, dz, dwp[ph], wsp, scales=scales)
                if self.prof is not None:
                    self.prof.end()
            ops.phase_wgrad_gather_s2(dwp, dw)
        else:
            wdesc, winfo = desc, info
            if desc.precision == RCF_PREC_F16X2 and (dz_amax is None or x_amax is None or x2_amax is None):
                wdesc = self._exact_unless(desc, None)
                winfo = ops.conv_query(wdesc)
            ws = self._newf((max(1, winfo.wgrad_workspace_floats),), dz)
            if self.prof is not None:
                self.prof.begin(winfo.wgrad_kernel_id, ops.algorithmic_flops(wdesc), wdesc)
            t1, k1 = self._src(x, winfo.wgrad_bn_on_load)
            t2, k2 = self._src(x2, winfo.wgrad_bn_on_load)
            if t1 is None:
                t1 = self._mat(x)
            if x2 is not None and t2 is None:
                t2 = self._mat(x2)
            scales = None
            if self._two_plane_wgrad(winfo.wgrad_kernel_id):
                scales = ops.make_scales(x_amax, None if x2 is None else x2_amax, None, dz_amax)
            ops.conv_wgrad(wdesc, t1, t2, dz, dw, ws, coef1=k1, coef2=k2, scales=scales)
            if self.prof is not None:
                self.prof.end()
        self._side_exit(side)
        self._wgrad_done(weight)
        for src, off, cnt in ((x, 0, desc.c1), (x2, desc.c1, desc.c2)):
            if src is None or not src.needs_grad:
                continue
            if src is x and desc.gather1 == RCF_GATHER_NEAREST:
                dd = ops.make_dgrad_desc(desc, off, cnt, False)
                tmp = self._new((desc.n, desc.h_in, desc.w_in, cnt), dz)
                self._run_dgrad(dd, weight, dz, tmp, dz_amax)
                acc = src.g is not None
                if acc:
                    src.bsum = None
                if not acc:
                    src.g = self._new(tuple(self._shape(src)), dz)
                ops.upsample_nearest_bwd(tmp, src.g, acc)
            elif self.use_phase_convs and desc.stride == 2 and desc.ksize == 3 and x2 is None and desc.c1 % 4 == 0:
                # transposed convolution in 4 phases (16 of the 36 zero-dilated taps are real)
                acc = src.g is not None
                if acc:
                    src.bsum = None
                if not acc:
                    src.g = self._new(tuple(self._shape(src)), dz)
                wd = self._phase_w(weight.detach(), RCF_PHASE_S2_DGRAD)
                wmax = None
                if self._s2_dgrad_merged(desc, wd, dz, src.g, acc, dz_amax):
                    continue
                for ph in range(4):
                    d = self._exact_unless(ops.make_s2_dgrad_desc(desc, ph >> 1, ph & 1, acc), dz_amax)
                    if ph == 0 and self._two_plane(ops.conv_query(d).kernel_id):
                        wmax = self._w_amax(wd)
                    self._run_packed(d, wd[ph], dz, src.g, amax_in=dz_amax, amax_w=wmax)
            else:
                acc = src.g is not None
                if acc:
                    src.bsum = None
                if desc.ksize == 1 and desc.stride == 2 and x2 is None and self._pw_s2_dgrad(desc, weight, dz, src, acc, dz_amax):
                    continue
                if not acc:
                    src.g = self._new(tuple(self._shape(src)), dz)
                dd = ops.make_dgrad_desc(desc, off, cnt, acc)
                self._run_dgrad(dd, weight, dz, src.g, dz_amax, sums_for=None if acc else src)

    def _pw_s2_dgrad(self, desc, weight, dz, src, acc, dz_amax):
        '''The input gradient of a 1x1 stride-2 convolution at dZ's resolution (ops.make_pw_s2_dgrad_desc): W^T dZ written to the even
        positions of a zero-filled (or accumulated-into) dX -- a quarter of the dense form's pixels, bitwise its result.  Returns False
        where no kernel takes the strided-output form (remembered per shape); RCF_PW_S2_DGRAD=0 switches it off.'''
        if not self.pw_s2_dgrad_lowres or ops.act_dtype() != torch.float32:
            # (bf16 tensors: conv1x1_b16_kernel already skips the zeros of the dilated dZ; the strided-output form measured 0.3 % slower)
            return False
        dd = self._exact_unless(ops.make_pw_s2_dgrad_desc(desc, acc), dz_amax)
        key = bytes(dd)
        if self._pw_s2_ok.get(key) is False:
            return False
        try:
            info = ops.conv_query(dd)
        except ops._lib.RcfUnsupported:
            self._pw_s2_ok[key] = False
            return False
        self._pw_s2_ok[key] = True
        if not acc:
            src.g = torch.zeros(tuple(self._shape(src)), dtype=ops.act_dtype(), device=dz.device)   # the odd positions stay zero
        scales = None
        if self._two_plane(info.kernel_id):
            wmax = self._w_amax(weight.detach())
            packed = self._pack(dd, weight.detach(), dz, wmax)
            scales = ops.make_scales(dz_amax, None, wmax)
        else:
            packed = self._pack(dd, weight.detach(), dz)
        if self.prof is not None:
            self.prof.begin(info.kernel_id, ops.algorithmic_flops(dd), dd)
        ops.conv_fwd(dd, dz, None, packed, src.g, None, scales=scales)
        if self.prof is not None:
            self.prof.end()
        return True

    def _bn_sums_launch(self, dd, info, dz, packed, out, scales, sums_for):
        '''The input-gradient launch `dd` writes sums_for.g for the first time.  If sums_for is the output of a BatchNorm + lrelu block
        and the kernel has the epilogue, it also takes that block's backward sums (no bn_act_bwd_reduce pass over dY and z later --
        unless another consumer adds into the gradient afterwards, which clears Act.bsum again).  -> True if it launched.'''
        if (sums_for is None or sums_for.bn is None or not self.bn_sums_in_dgrad or not info.bn_bwd_sums
                or (scales is None and out.dtype != torch.bfloat16) or tuple(sums_for.bn[0].shape) != tuple(out.shape)):
            return False
        z, coef = sums_for.bn
        part = torch.empty((info.n_partials, 2, dd.c_out), dtype=torch.float64, device=out.device)
        ops.conv_dgrad_bn_sums(dd, dz, packed, out, z, coef, part, scales)
        sums_for.bsum = (part, info.n_partials)
        self.bn_sums_taken += 1
        return True

    def _run_dgrad(self, dd, weight, dz, out, dz_amax=None, sums_for=None):
        dd = self._exact_unless(dd, dz_amax)
        info = ops.conv_query(dd)
        scales = None
        if self._two_plane(info.kernel_id):
            wmax = self._w_amax(weight.detach())
            packed = self._pack(dd, weight.detach(), dz, wmax)
            scales = ops.make_scales(dz_amax, None, wmax)
        else:
            packed = self._pack(dd, weight.detach(), dz)
        if self.prof is not None:
            self.prof.begin(info.kernel_id, ops.algorithmic_flops(dd), dd)
        if not self._bn_sums_launch(dd, info, dz, packed, out, scales, sums_for):
            ops.conv_fwd(dd, dz, None, packed, out, None, scales=scales)
        if self.prof is not None:
            self.prof.end()

    def _bn_coef(self, layer, partials, info, z):
        bn = layer.batch_norm
        c = z.shape[3]
        coef = self._newf((4, c), z)
        count = z.shape[0] * z.shape[1] * z.shape[2]
        ops.bn_finalize(partials, info.n_partials if partials is not None else 0, c, count, bn.weight.detach(),
                        bn.bias.detach(), bn.running_mean, bn.running_var, BN_MOMENTUM, BN_EPS, self.training, coef)
        return coef

    def _bn_coef_eval(self, layer, ref):
        bn = layer.batch_norm

        def make():
            c = bn.weight.shape[0]
            coef = self._newf((4, c), ref)
            ops.bn_finalize(None, 0, c, 1, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, BN_MOMENTUM, BN_EPS,
                            False, coef)
            return coef
        return self._frozen_get(('coef', id(layer)), make)

    # ------------------------------------------------------------------ layer ops
    def conv_bn_act(self, layer, x, x2=None, up_hw=None, res=None, feeds_head=False):
        '''
        net_utils.Conv2d.forward (src/net_utils.py:84-91) with BN + leaky_relu (+ the residual tail of
        ResNetBlock.forward when res is given: lrelu(lrelu(BN(conv)) + res), src/net_utils.py:311-323).
        '''
        if not layer.use_batch_norm or layer.activation_func != 'leaky_relu':
            raise ValueError('conv_bn_act expects a BatchNorm + leaky_relu Conv2d block')
        if self.fuse_eval and not self.training and self.tape is None:
            # inference: eval-mode BatchNorm is affine per channel -> scale into the weights, shift + LeakyReLU (+ residual tail) into the
            # conv kernel's epilogue; layers whose kernel has no such epilogue (f32-MFMA 1x1 / stride-2 / stem) fall through.  The block
            # in front of the output head too: its epilogue is free, and a head that reads finished bf16 activations runs on the matrix
            # pipe (rcf_head_fwd_b16) -- BatchNorm-on-load in the head is the TRAINING arrangement (the statistics are not known earlier)
            src = x.t if x.t is not None else (x.z if x.z is not None else x.s2d)
            coef = self._bn_coef_eval(layer, src)
            z, desc, info, partials, fused = self._conv(layer, x, x2, up_hw, want_stats=False,
                                                        fold=(coef, None if res is None else self._mat(res)))
            if fused:
                out = Act(z)
                out.head_fusable = False
                return out
        else:
            z, desc, info, partials = self._conv(layer, x, x2, up_hw, want_stats=self.training)
            coef = self._bn_coef(layer, partials, info, z)
        n_pix = z.shape[0] * z.shape[1] * z.shape[2]
        c = z.shape[3]
        if res is None and (self.bn_on_load or (feeds_head and ops.head_bn_blocks(z.shape[0], z.shape[1], z.shape[2], c) > 0)):
            # deferred activation: consumers that apply BatchNorm + lrelu while they load z (the output head; with
            # bn_on_load also the split conv / wgrad kernels) never need the tensor; any other consumer materialises it
            # once through _mat()
            out = Act(None)
            out.z, out.coef = z, coef
        else:
            out = Act(torch.empty_like(z))
            out.amax = self._amax_slot() if z.dtype == torch.float32 else None
            ops.bn_act_fwd(z, coef, None if res is None else self._mat(res), out.t, n_pix, c, RCF_ACT_LEAKY_RELU, amax=out.amax)
            if res is None and self.tape is not None:
                out.bn = (z, coef)
        out.head_fusable = res is None
        if self.tape is not None:
            bn = layer.batch_norm
            batch_stats = self.training
            scope = self._scope

            def backward():
                self._set_scope(scope)
                if out.g is None and out.g_head is not None:
                    # the only consumer was the output head: its input gradient is recomputed inside both BN-backward passes
                    dlogit, w_head = out.g_head
                    out.g_head = None
                    nb = ops.head_bn_blocks(z.shape[0], z.shape[1], z.shape[2], c)
                    bpart = torch.empty((nb, 2, c), dtype=torch.float64, device=z.device)
                    ops.head_bn_bwd_reduce(dlogit, w_head, z, coef, bpart)
                    bcoef = self._newf((2, c), z)
                    ops.bn_bwd_finalize(bpart, nb, 2 * c, c, n_pix, bcoef, self.grad_of(bn.weight), self.grad_of(bn.bias))
                    if not batch_stats:
                        bcoef.zero_()
                    self._wgrad_done(bn.weight, bn.bias)
                    dz = torch.empty_like(z)
                    dz_amax = self._amax_slot() if z.dtype == torch.float32 else None
                    ops.head_bn_bwd_apply(dlogit, w_head, z, coef, bcoef, dz, amax=dz_amax)
                    self._conv_backward(layer, desc, info, x, x2, dz, dz_amax)
                    return
                dout = out.g
                has_res = res is not None   # out.t is read only then (a deferred activation has no residual)
                if out.bsum is not None:    # the kernel that wrote dout took the sums on its way (Engine._bn_sums_launch)
                    bpart, nb = out.bsum
                    out.bsum = None
                else:
                    nb = ops.ew_blocks(n_pix, c)
                    bpart = torch.empty((nb, 2, c), dtype=torch.float64, device=z.device)
                    ops.bn_act_bwd_reduce(dout, z, coef, out.t, bpart, n_pix, c, RCF_ACT_LEAKY_RELU, has_res)
                bcoef = self._newf((2, c), z)
                ops.bn_bwd_finalize(bpart, nb, 2 * c, c, n_pix, bcoef, self.grad_of(bn.weight), self.grad_of(bn.bias))
                if not batch_stats:
                    bcoef.zero_()   # eval-mode BN is affine: no batch-statistic terms in dz
                self._wgrad_done(bn.weight, bn.bias)
                dz = torch.empty_like(z)
                dres, dres_acc = None, False
                if has_res and res.needs_grad:
                    dres_acc = res.g is not None
                    if dres_acc:
                        res.bsum = None
                    if not dres_acc:
                        res.g = torch.empty_like(res.t)
                    dres = res.g
                dz_amax = self._amax_slot() if z.dtype == torch.float32 else None
                ops.bn_act_bwd_apply(dout, z, coef, out.t, bcoef, dz, dres, dres_acc, n_pix, c, RCF_ACT_LEAKY_RELU, has_res, amax=dz_amax)
                out.g = None
                self._conv_backward(layer, desc, info, x, x2, dz, dz_amax)

            self.tape.append(backward)
        return out

    def conv_plain(self, layer, x):
        '''Conv2d without BN and activation (ResNetBlock.projection, src/net_utils.py:300-307).'''
        if layer.use_batch_norm or layer.activation_func is not None:
            raise ValueError('conv_plain expects a bare conv')
        z, desc, info, _ = self._conv(layer, x)
        out = Act(z)
        if self.tape is not None:
            scope = self._scope

            def backward():
                self._set_scope(scope)
                dz = out.g
                out.g = None
                self._conv_backward(layer, desc, info, x, None, dz)
            self.tape.append(backward)
        return out

    def fuse(self, layer_w, layer_p, dep, img):
        '''conv_weight * conv_project + image (src/networks.py:863-866): sigmoid(BN(W1 d)) * BN(W2 d) + img.'''
        if (self.fuse_eval and not self.training and self.tape is None and self.fuse_wp_one_pass
                and layer_w.kernel_size == 1 and layer_p.kernel_size == 1 and layer_w.stride == 1 and layer_p.stride == 1):
            # inference on bf16 tensors: the BatchNorms are affine maps known up front -- both 1x1 convolutions, the gate and the sum
            # in one streaming kernel (rcf_fuse_wp_infer_b16: c_d + 2 c_i elements of traffic per pixel instead of 2 c_d + 6 c_i)
            dt, it = self._mat(dep), self._mat(img)
            if (dt.dtype == torch.bfloat16 and it.dtype == torch.bfloat16 and tuple(dt.shape[:3]) == tuple(it.shape[:3])
                    and ops.fuse_wp_infer_supported(dt.shape[3], it.shape[3])):
                out = Act(torch.empty_like(it))
                ops.fuse_wp_infer(dt, layer_w.conv.weight.detach(), self._bn_coef_eval(layer_w, it), layer_p.conv.weight.detach(),
                                  self._bn_coef_eval(layer_p, it), it, out.t)
                return out
        zw, dw_, iw, pw = self._conv(layer_w, dep, want_stats=self.training)
        zp, dp_, ip, pp = self._conv(layer_p, dep, want_stats=self.training)
        coef_w = self._bn_coef(layer_w, pw, iw, zw)
        coef_p = self._bn_coef(layer_p, pp, ip, zp)
        n_pix = zw.shape[0] * zw.shape[1] * zw.shape[2]
        c = zw.shape[3]
        out = Act(torch.empty_like(zw))
        out.amax = self._amax_slot() if zw.dtype == torch.float32 else None
        ops.fuse_fwd(zw, coef_w, zp, coef_p, self._mat(img), out.t, n_pix, c, amax=out.amax)
        if self.tape is not None:
            bnw, bnp = layer_w.batch_norm, layer_p.batch_norm
            batch_stats = self.training
            scope = self._scope

            def backward():
                self._set_scope(scope)
                dout = out.g
                nb = ops.ew_blocks(n_pix, c)
                bpart = torch.empty((nb, 4, c), dtype=torch.float64, device=zw.device)
                ops.fuse_bwd_reduce(dout, zw, coef_w, zp, coef_p, bpart, n_pix, c)
                bcw = self._newf((2, c), zw)
                bcp = self._newf((2, c), zw)
                ops.bn_bwd_finalize(bpart, nb, 4 * c, c, n_pix, bcw, self.grad_of(bnw.weight), self.grad_of(bnw.bias))
                ops.bn_bwd_finalize(bpart.view(-1)[2 * c:], nb, 4 * c, c, n_pix, bcp, self.grad_of(bnp.weight),
                                    self.grad_of(bnp.bias))
                if not batch_stats:
                    bcw.zero_()
                    bcp.zero_()
                self._wgrad_done(bnw.weight, bnw.bias, bnp.weight, bnp.bias)
                dzw = torch.empty_like(zw)
                dzp = torch.empty_like(zp)
                dimg, dimg_acc = None, False
                if img.needs_grad:
                    dimg_acc = img.g is not None
                    if dimg_acc:
                        img.bsum = None
                    if not dimg_acc:
                        img.g = torch.empty_like(img.t)
                    dimg = img.g
                ops.fuse_bwd_apply(dout, zw, coef_w, zp, coef_p, bcw, bcp, dzw, dzp, dimg, dimg_acc, n_pix, c)
                out.g = None
                # the two 1x1 convolutions' backward writes the DEPTH branch's gradient: on that branch's stream when it has one
                br = self._branch_enter(first_wait=True)
                if br is not None:
                    self._side_keep.extend((dzw, dzp, zw, zp))
                self._conv_backward(layer_w, dw_, iw, dep, None, dzw)
                self._conv_backward(layer_p, dp_, ip, dep, None, dzp)
                self._branch_exit(br)

            self.tape.append(backward)
        return out

    def max_pool(self, x):
        '''torch.nn.MaxPool2d(3, 2, 1) (src/networks.py:392-395).'''
        xt = self._mat(x)
        n, h, w, c = xt.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        out = Act(self._new((n, ho, wo, c), xt))
        out.amax = x.amax   # a maximum over windows of x: max|x| bounds it
        idx = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=xt.device)
        ops.maxpool_fwd(xt, out.t, idx)
        if self.tape is not None:
            def backward():
                acc = x.g is not None
                if acc:
                    x.bsum = None
                if not acc:
                    x.g = torch.empty_like(x.t)
                ops.maxpool_bwd(out.g, idx, x.g, acc)
                out.g = None
            self.tape.append(backward)
        return out

    def resnet_block(self, block, x):
        '''ResNetBlock.forward (src/net_utils.py:309-323).'''
        c1 = self.conv_bn_act(block.conv1, x)
        shortcut = self.conv_plain(block.projection, x) if block.uses_projection else x
        return self.conv_bn_act(block.conv2, c1, res=shortcut)

    def decoder_block(self, block, x, skip=None, shape=None, feeds_head=False):
        '''DecoderBlock.forward (src/net_utils.py:535-569), deconv_type 'up' or 'transpose'.'''
        if block.deconv_type == 'transpose':
            deconv = self.conv_bn_act(block.deconv, x)       # always 2x; `shape` is ignored like in the reference (:554-555)
            if skip is not None and tuple(self._shape(skip)[1:3]) != tuple(self._shape(deconv)[1:3]):
                # the reference fails in torch.cat here (e.g. 15 -> 30 against a 29-row skip at 900 x 1600)
                raise RuntimeError('Sizes of tensors must match except in dimension 1: transposed convolution gives %s, skip is %s'
                                   % (tuple(self._shape(deconv)[1:3]), tuple(self._shape(skip)[1:3])))
            if block.skip_channels > 0:
                return self.conv_bn_act(block.conv, deconv, x2=skip, feeds_head=feeds_head)
            return self.conv_bn_act(block.conv, deconv, feeds_head=feeds_head)
        if skip is not None:
            shape = self._shape(skip)[1:3]
        elif shape is None:
            shape = (2 * self._shape(x)[1], 2 * self._shape(x)[2])
        deconv = self.conv_bn_act(block.deconv.conv, x, up_hw=shape)
        if block.skip_channels > 0:
            return self.conv_bn_act(block.conv, deconv, x2=skip, feeds_head=feeds_head)
        return self.conv_bn_act(block.conv, deconv, feeds_head=feeds_head)

    def head(self, layer, x, logits=False):
        '''output0 (src/networks.py:1548-1555, :1654) + d = min/(sigmoid(o)+min/max) (src/fusionnet_model.py:162-165).
        logits=True (RadarNet, src/radarnet_model.py:118-126): the raw 3x3 C->1 output is the result and receives the gradient.'''
        if x.t is None and ops.head_bn_blocks(*x.z.shape) <= 0:
            self._mat(x)                          # the fused head kernels do not cover this channel count
        xin = x.t if x.t is not None else x.z     # deferred activation: raw conv output + coefficients
        xcoef = None if x.t is not None else x.coef
        n, h, w, c = xin.shape
        weight = layer.conv.weight
        logit = self._newf((n, h, w), xin)
        depth = Act(self._newf((n, h, w), xin))
        ops.head_fwd(xin, weight.detach(), logit, depth.t, self.dmin, self.dmax, coef=xcoef)
        if logits:
            depth = Act(logit)
        if self.tape is not None:
            def backward():
                if logits:
                    dlogit = depth.g
                else:
                    dlogit = torch.empty_like(logit)
                    ops.head_bwd_logit(depth.g, logit, dlogit, self.dmin, self.dmax)
                depth.g = None
                ops.head_bwd_wgrad(xin, dlogit, self.grad_of(weight), coef=xcoef)
                self._wgrad_done(weight)
                if x.needs_grad:
                    if x.g is not None:
                        raise RuntimeError('head input has another consumer')
                    if x.head_fusable and x.t is None:
                        x.g_head = (dlogit, weight.detach())   # consumed by the producer's BatchNorm backward
                    else:
                        x.g = torch.empty_like(xin)
                        ops.head_bwd_dgrad(dlogit, weight.detach(), x.g)
            self.tape.append(backward)
        return depth

    # ------------------------------------------------------------------ the network
    @staticmethod
    def _input(nhwc, s2d, hw):
        a = Act(nhwc, needs_grad=False)
        if isinstance(s2d, tuple):      # fp32 space-to-depth image + the device scalar holding max|input| (ops.s2d_image_f32)
            s2d, a.amax = s2d
        a.s2d, a.hw = s2d, hw
        return a

    def forward(self, image_nhwc, depth_nhwc, training, record, image_s2d=None, depth_s2d=None, hw=None):
        '''
        image_nhwc (N,H,W,3), depth_nhwc (N,H,W,2) -> depth (N,H,W) as an Act.  record=True keeps the tape for
        backward(); record=False is the no_grad / eval path of validate()/run() (src/fusionnet_main.py:517, :814).
        bf16 configuration: image_s2d / depth_s2d (ops.s2d_image of the NCHW inputs) feed the stems and hw = (H, W); the fp32 NHWC
        inputs are then only needed for the stems' weight gradients (record=True) and may be None otherwise.
        '''
        enc, dec = self.encoder, self.decoder
        self.training = bool(training)
        self.tape = _Tape(self) if record else None
        if hw is None:
            hw = tuple(image_nhwc.shape[1:3])
        self._begin_step_scales(image_nhwc if image_nhwc is not None else image_s2d)
        self._set_scope('encoder')
        img = self.conv_bn_act(enc.conv1_image, self._input(image_nhwc, image_s2d, hw))
        br = self._branch_enter(first_wait=True)     # the depth branch: its own stream up to each level's fusion
        dep = self.conv_bn_act(enc.conv1_depth, self._input(depth_nhwc, depth_s2d, hw))
        self._branch_exit(br)
        layers = [self._fuse_level(enc.conv1_weight, enc.conv1_project, dep, img)]
        img = self.max_pool(img)
        br = self._branch_enter(first_wait=False)
        dep = self.max_pool(dep)
        self._branch_exit(br)
        for lvl in range(2, enc.network_depth + 1):
            for blk_i, blk_d in zip(getattr(enc, 'blocks%d_image' % lvl), getattr(enc, 'blocks%d_depth' % lvl)):
                img = self.resnet_block(blk_i, img)
                br = self._branch_enter(first_wait=False)
                dep = self.resnet_block(blk_d, dep)
                self._branch_exit(br)
            layers.append(self._fuse_level(getattr(enc, 'conv%d_weight' % lvl), getattr(enc, 'conv%d_project' % lvl), dep, img))
        if self.fuse_on_branch and self.branch_stream and self._branch_busy:   # the decoder reads the fused tensors: the main stream joins the branch here
            cur = torch.cuda.current_stream()
            cur.wait_stream(self._branch)
            for a in layers:
                for t in (a.t, a.z):
                    if t is not None:
                        t.record_stream(cur)
        latent, skips = layers[-1], layers[:-1]
        out = self.head(dec.output0, self._decode(latent, skips, hw))
        tape, self.tape = self.tape, None
        return out, tape

    def _fuse_level(self, layer_w, layer_p, dep, img):
        '''One level's fusion.  Default: on the main stream once the depth branch has delivered `dep`.  fuse_on_branch: on the branch
        stream once the main stream has delivered `img` -- the main stream goes straight on to the next level's image block; the
        fusion's backward stays a main-stream closure (it adds into the image branch's gradient).'''
        if not (self.fuse_on_branch and self.branch_stream) or self.bn_on_load:
            self._branch_wait(dep)
            return self.fuse(layer_w, layer_p, dep, img)
        br = self._branch_enter(first_wait=True)
        if br is not None and img.t is not None:
            img.t.record_stream(self._branch)   # (inference: the image activation is freed by the main stream as soon as it moves on)
        self._tape_main = True
        try:
            return self.fuse(layer_w, layer_p, dep, img)
        finally:
            self._tape_main = False
            self._branch_exit(br)

    def _decode(self, latent, skips, shape):
        '''MultiScaleDecoder.forward, n_resolution == 1 (src/networks.py:1571-1657), up to the input of output0.'''
        dec = self.decoder
        self._set_scope('decoder')
        x = latent
        n = len(skips) - 1
        names = dec.block_names
        for name in names[:-1]:
            x = self.decoder_block(getattr(dec, name), x, skip=skips[n])
            n -= 1
        if n == 0:
            return self.decoder_block(dec.deconv0, x, skip=skips[0], feeds_head=True)
        return self.decoder_block(dec.deconv0, x, shape=shape, feeds_head=True)

    # ------------------------------------------------------------------ RadarNet stage 1
    def roi_pool(self, x, rois, out_hw, scale, out=None, coff=0):
        '''torchvision.ops.roi_pool (src/networks.py:1232-1247) of activation x around every radar point; written into the
        channels [coff, coff + C) of `out` when given (the latent shared with the radar branch).'''
        xt = self._mat(x)
        n, h, w, c = xt.shape
        r = rois.shape[0]
        if out is None:
            out = Act(self._new((r, out_hw[0], out_hw[1], c), xt))
        argmax = torch.empty((r, out_hw[0], out_hw[1], c), dtype=torch.int32, device=xt.device)
        ops.roi_pool_fwd(xt, rois, out.t, argmax, out_hw, scale, out_coff=coff)
        if self.tape is not None:
            def backward():
                if x.needs_grad and out.g is not None:
                    # gathered per input pixel (rcf_roi_pool_bwd_gather): written once in the tensors' own storage -- the scatter form
                    # (rcf_roi_pool_bwd: fp32 atomics into a zeroed copy) stays in the ABI
                    acc = x.g is not None
                    if acc:
                        x.bsum = None
                    if rois.shape[0] <= 1024:
                        if not acc:
                            x.g = torch.empty_like(xt)
                        ops.roi_pool_bwd_gather(out.g, argmax, rois, x.g, acc, out_hw, scale, dout_coff=coff)
                    else:   # more rois than the gather form's table holds: scatter-add into a zeroed fp32 tensor, then join the gradient
                        tmp = torch.zeros(xt.shape, dtype=torch.float32, device=xt.device)
                        ops.roi_pool_bwd(out.g, argmax, rois, tmp, out_hw, dout_coff=coff)
                        if not acc:
                            x.g = torch.empty_like(xt)
                        ops.convert(tmp, x.g, accumulate=acc)
            self.tape.append(backward)
        return out

    def fully_connected_encoder(self, enc, points, latent, hw, coff):
        '''FullyConnectedEncoder.forward (src/networks.py:1065-1067); the last layer writes feature c * hw + p into
        latent.t[m, p, coff + c] (= the .view(M, C, -1, W) + torch.cat of src/networks.py:1251-1255).'''
        layers = list(enc.mlp)
        acts = [points]
        ctot = latent.t.shape[-1]
        for i, fc in enumerate(layers):
            lin = fc.fully_connected
            act_on = fc.activation_func is not None
            last = i == len(layers) - 1
            if last:
                ops.fc_fwd(acts[-1], lin.weight.detach(), lin.bias.detach(), latent.t, act_on, hw, ctot, coff)
                acts.append(latent.t)
            else:
                y = self._newf((points.shape[0], lin.out_features), points)
                ops.fc_fwd(acts[-1], lin.weight.detach(), lin.bias.detach(), y, act_on)
                acts.append(y)
        if self.tape is not None:
            def backward():
                dy = latent.g
                for i in range(len(layers) - 1, -1, -1):
                    fc = layers[i]
                    lin = fc.fully_connected
                    last = i == len(layers) - 1
                    dx = self._newf(tuple(acts[i].shape), acts[i]) if i > 0 else None   # the radar points need no gradient
                    ops.fc_bwd(acts[i], lin.weight.detach(), acts[i + 1], dy, self.grad_of(lin.weight), self.grad_of(lin.bias), dx,
                               fc.activation_func is not None, hw if last else 1, ctot if last else 0, coff if last else 0)
                    self._wgrad_done(lin.weight, lin.bias)
                    dy = dx
            self.tape.append(backward)

    def forward_radarnet(self, image_nhwc, points, rois, training, record, image_s2d=None, hw=None):
        '''
        RadarNetV1Encoder.forward (src/networks.py:1203-1256) + MultiScaleDecoder.forward + output0 -> logits (M,H,W) as an Act;
        image_nhwc (N,H,W,3), points (M,3), rois (M,5) = (image index, x1, y1, x2, y2).
        '''
        enc = self.encoder
        ei = enc.encoder_image
        self.training = bool(training)
        self.tape = [] if record else None
        self._begin_step_scales(image_nhwc if image_nhwc is not None else image_s2d)
        shape = (int(enc.input_patch_size_image[0]), int(enc.input_patch_size_image[1]))
        # ResNetEncoder.forward (src/networks.py:232-268)
        x = self.conv_bn_act(ei.conv1, self._input(image_nhwc, image_s2d, hw if hw is not None else tuple(image_nhwc.shape[1:3])))
        layers = [x]
        x = self.max_pool(x)
        for lvl in range(2, ei.network_depth + 1):
            for blk in getattr(ei, 'blocks%d' % lvl):
                x = self.resnet_block(blk, x)
            layers.append(x)
        latent_image, skips_image = layers[-1], layers[:-1]
        # ROI pooling (src/networks.py:1215-1247)
        skip_scales = [1 / 2.0, 1 / 4.0, 1 / 8.0, 1 / 16.0, 1 / 32.0, 1 / 64.0, 1 / 128.0]
        lat_hw = (int(shape[0] // 32.0), int(shape[1] // 32.0))
        m = rois.shape[0]
        c_img = self._shape(latent_image)[3]
        c_dep = enc.n_neuron_latent_depth
        latent = Act(self._new((m, lat_hw[0], lat_hw[1], c_img + c_dep), rois))
        skips = [self.roi_pool(s, rois, (int(shape[0] * skip_scales[i]), int(shape[1] * skip_scales[i])), skip_scales[i])
                 for i, s in enumerate(skips_image)]
        # radar point branch -> channels [c_img, c_img + c_dep) of the latent, pooled image latent -> [0, c_img)
        self.fully_connected_encoder(enc.encoder_depth, points, latent, lat_hw[0] * lat_hw[1], c_img)
        self.roi_pool(latent_image, rois, lat_hw, 1 / 32.0, out=latent, coff=0)
        out = self.head(self.decoder.output0, self._decode(latent, skips, shape), logits=True)
        tape, self.tape = self.tape, None
        return out, tape

    @staticmethod
    def backward(out, tape, ddepth):
        '''Run the recorded tape in reverse.  ddepth: (N,H,W) gradient of the loss w.r.t. the output depth.'''
        out.g = ddepth
        while tape:
            tape.pop()()
