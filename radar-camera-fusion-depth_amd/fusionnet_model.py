'''
FusionNetModel -- drop-in for the reference's src/fusionnet_model.py:7-587 on MI355X.

Same constructor arguments, same methods (forward, compute_loss, parameters, train, eval, to, save_model,
restore_model, data_parallel, log_summary), same checkpoint dictionary and state_dict names, so the reference's
training loop (src/fusionnet_main.py:180-200, :369-399, :428-451) runs unchanged:

    output_depth = model.forward(image=image, input_depth=input_depth)
    loss, loss_info = model.compute_loss(...)
    optimizer.zero_grad(); loss.backward(); optimizer.step()

forward() and compute_loss() return torch tensors that participate in autograd, but everything between the
inputs and the loss runs on the hand-written HIP kernels of librcf_hip.so through engine.py; loss.backward()
replays the engine's tape and leaves each parameter's .grad as a view into one flat gradient arena (which is
what the bucketed RCCL all-reduce and the fused Adam operate on).  There is no CPU path: a missing library or a
CPU tensor raises.
'''

import os

import torch

from . import _lib, networks, ops
from .engine import Engine


class _FusionNetFunction(torch.autograd.Function):
    '''Bridges the engine's tape into torch.autograd: one node for the whole encoder-decoder.'''

    @staticmethod
    def forward(ctx, model, image, input_depth, anchor):
        out, tape = model._run_engine(image, input_depth, record=True)
        ctx.model = model
        ctx.out = out
        ctx.tape = tape
        return out.t.unsqueeze(1)

    @staticmethod
    def backward(ctx, grad_output):
        model, out, tape = ctx.model, ctx.out, ctx.tape
        if tape is None:
            raise RuntimeError('FusionNetModel: backward through the same forward twice')
        ctx.tape = None
        n, _, h, w = grad_output.shape
        ddepth = grad_output.contiguous().view(n, h, w)
        model._backward(out, tape, ddepth)
        return None, None, None, None


class _MaskedL1Function(torch.autograd.Function):
    '''compute_loss with loss_func 'l1' / 'l2' / 'smoothl1' (src/fusionnet_model.py:209-275) on the HIP loss kernels.'''

    @staticmethod
    def forward(ctx, output_depth, ground_truth, lidar_map, w_lidar, model, kind='l1', image=None, w_smoothness=0.0):
        d = output_depth.contiguous()
        gt = ground_truth.contiguous()
        lidar = lidar_map.contiguous()
        sums = torch.empty(4, dtype=torch.float64, device=d.device)
        loss = torch.empty(3, dtype=torch.float32, device=d.device)
        sums_s, smooth = None, None
        with torch.cuda.device(d.device):
            ops.l1_loss_fwd(d, gt, lidar, sums, kind)
            if model is not None:
                model._all_reduce_loss_sums(sums)
            ops.l1_loss_value(sums, float(w_lidar), loss)
            if w_smoothness > 0.0:
                # the local smoothness term (src/fusionnet_model.py:277-281): its sums travel like the masked loss's (per-rank sums and
                # counts, all-reduced before the backward: the reference takes ONE mean over the gathered batch)
                image = image.contiguous()
                sums_s = torch.empty(4, dtype=torch.float64, device=d.device)
                ops.smoothness_loss_fwd(image, d, sums_s)
                if model is not None:
                    model._all_reduce_loss_sums(sums_s)
                smooth = (sums_s[0] / sums_s[1] + sums_s[2] / sums_s[3]).to(torch.float32)
        ctx.save_for_backward(d, gt, lidar, sums, image if sums_s is not None else None, sums_s)
        ctx.w_lidar = float(w_lidar)
        ctx.w_smoothness = float(w_smoothness)
        ctx.kind = kind
        total = loss[0].clone() if smooth is None else loss[0] + float(w_smoothness) * smooth
        terms = loss if smooth is None else torch.cat([loss, smooth.view(1)])   # (total without smoothness, supervised, lidar[, smoothness])
        ctx.mark_non_differentiable(terms)
        return total, terms

    @staticmethod
    def backward(ctx, grad_loss, _grad_terms):
        d, gt, lidar, sums, image, sums_s = ctx.saved_tensors
        dd = torch.empty_like(d)
        up = grad_loss.contiguous().view(1).to(torch.float32)
        with torch.cuda.device(d.device):
            ops.l1_loss_bwd(d, gt, lidar, sums, up, ctx.w_lidar, dd, ctx.kind)
            if sums_s is not None:
                ops.smoothness_loss_bwd(image, d, sums_s, up, ctx.w_smoothness, dd)
        return dd, None, None, None, None, None, None, None


class FusionNetModel(object):
    '''
    Image radar fusion (see the reference docstring, src/fusionnet_model.py:8-44, for the argument meanings).
    '''

    def __init__(self,
                 input_channels_image,
                 input_channels_depth,
                 encoder_type,
                 n_filters_encoder_image,
                 n_filters_encoder_depth,
                 fusion_type,
                 decoder_type,
                 n_resolution_decoder,
                 n_filters_decoder,
                 deconv_type,
                 activation_func,
                 weight_initializer,
                 min_predict_depth,
                 max_predict_depth,
                 device=torch.device('cuda')):

        self.encoder_type = encoder_type
        self.min_predict_depth = min_predict_depth
        self.max_predict_depth = max_predict_depth
        self.device = torch.device(device)

        # src/fusionnet_model.py:68-82
        if fusion_type in ('add', 'weight', 'weight_and_project'):
            n_filters_encoder = n_filters_encoder_image
            latent_channels = n_filters_encoder[-1]
        elif fusion_type == 'concat':
            n_filters_encoder = [i + z for i, z in zip(n_filters_encoder_image, n_filters_encoder_depth)]
            latent_channels = n_filters_encoder[-1]
        else:
            raise ValueError('Unsupported fusion type: {}'.format(fusion_type))

        # src/fusionnet_model.py:84-115
        if 'fusionnet18' in encoder_type or 'resnet18' in encoder_type:
            n_layer = 18
        elif 'fusionnet34' in encoder_type or 'resnet34' in encoder_type:
            n_layer = 34
        else:
            raise ValueError('Unsupported encoder type: {}'.format(encoder_type))

        if 'fusionnet18' in encoder_type or 'fusionnet34' in encoder_type:
            self.encoder = networks.FusionNetEncoder(
                n_layer=n_layer,
                input_channels_image=input_channels_image,
                input_channels_depth=input_channels_depth,
                n_filters_encoder_image=n_filters_encoder_image,
                n_filters_encoder_depth=n_filters_encoder_depth,
                weight_initializer=weight_initializer,
                activation_func=activation_func,
                use_batch_norm='batch_norm' in encoder_type,
                fusion_type=fusion_type)
        else:
            # the reference's image-only ResNetEncoder (:103-113) is not on the FusionNet hot path
            raise ValueError('Unsupported encoder type on the HIP path: {}'.format(encoder_type))

        n_skips = list(n_filters_encoder[:-1])
        n_skips = n_skips[::-1] + [0]                                            # :118-119

        if 'multiscale' in decoder_type:
            self.decoder = networks.MultiScaleDecoder(
                input_channels=latent_channels,
                output_channels=1,
                n_resolution=n_resolution_decoder,
                n_filters=n_filters_decoder,
                n_skips=n_skips,
                weight_initializer=weight_initializer,
                activation_func=activation_func,
                output_func='linear',
                use_batch_norm='batch_norm' in decoder_type,
                deconv_type=deconv_type)
        else:
            raise ValueError('Unsuported decoder type: {}'.format(decoder_type))

        if not ('batch_norm' in encoder_type and 'batch_norm' in decoder_type):
            raise ValueError('HIP path implements the shipped batch_norm encoder/decoder only')

        self._is_data_parallel = False
        self._dp = None
        self._training = True
        self._anchor = None
        self._engine = Engine(self.encoder, self.decoder, min_predict_depth, max_predict_depth)
        self._engine.grad_of = self._grad_of
        self._param_arena = None
        self._grad_arena = None

        # Move to device
        self.to(self.device)

    # ------------------------------------------------------------------ parameter / gradient arenas
    def _forward_order_params(self):
        '''Parameters in the order the engine's forward touches them (used ones only).'''
        enc, dec = self.encoder, self.decoder
        order = []

        def conv_block(layer):
            order.append(layer.conv.weight)
            if layer.use_batch_norm:
                order.extend([layer.batch_norm.weight, layer.batch_norm.bias])

        def res_block(b):
            conv_block(b.conv1)
            if b.uses_projection:
                conv_block(b.projection)
            conv_block(b.conv2)

        conv_block(enc.conv1_image); conv_block(enc.conv1_depth)
        conv_block(enc.conv1_weight); conv_block(enc.conv1_project)
        for lvl in range(2, enc.network_depth + 1):
            for bi, bd in zip(getattr(enc, 'blocks%d_image' % lvl), getattr(enc, 'blocks%d_depth' % lvl)):
                res_block(bi); res_block(bd)
            conv_block(getattr(enc, 'conv%d_weight' % lvl)); conv_block(getattr(enc, 'conv%d_project' % lvl))
        for name in dec.block_names:
            blk = getattr(dec, name)
            conv_block(blk.deconv if blk.deconv_type == 'transpose' else blk.deconv.conv); conv_block(blk.conv)
        conv_block(dec.output0)
        return order

    def _build_arenas(self):
        '''
        One flat fp32 arena for all parameters and one for their gradients.  Layout = reverse forward order
        (the order gradients become final during backward: decoder first), then the parameters the network never
        uses (ResNetBlock.projection of shape-preserving blocks: they get no gradient).
        '''
        used = self._forward_order_params()[::-1]
        used_ids = set(id(p) for p in used)
        unused = [p for p in self.parameters() if id(p) not in used_ids]
        ordered = used + unused
        device = self.device
        total = sum(p.numel() for p in ordered)
        arena = torch.empty(total, dtype=torch.float32, device=device)
        garena = torch.zeros(total, dtype=torch.float32, device=device)
        self._grad_views = {}
        self._param_offset = {}
        off = 0
        with torch.no_grad():
            for p in ordered:
                n = p.numel()
                view = arena[off:off + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                self._grad_views[id(p)] = garena[off:off + n].view(p.shape)
                self._param_offset[id(p)] = off
                p._rcf_arena = (arena, garena, off)
                off += n
        self._param_arena, self._grad_arena = arena, garena
        self._used_params = used
        self._n_used = sum(p.numel() for p in used)
        # BatchNorm num_batches_tracked: views of one int64 arena, bumped with a single add per training forward
        bns = [m for mod in (self.encoder, self.decoder) for m in mod.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        self._nbt = torch.zeros(len(bns), dtype=torch.int64, device=device)
        for i, bn in enumerate(bns):
            self._nbt[i] = bn.num_batches_tracked.to(device)
            bn._buffers['num_batches_tracked'] = self._nbt[i]
        self._anchor = torch.zeros((), dtype=torch.float32, device=device, requires_grad=True)
        if self._dp is not None:
            self._dp.rebuild(self)

    def _grad_of(self, p):
        return self._grad_views[id(p)]

    # ------------------------------------------------------------------ engine entry points
    # ------------------------------------------------------------------ arithmetic of the convolutions
    batch_weight_packing = os.environ.get('RCF_BATCH_PACK', '1') != '0'
    '''
    Training: pack all weights of a step with a few batched launches at the start of the forward pass instead of ~200 small launches
    spread over the step (engine.WeightPlan; bitwise the same results).  RCF_BATCH_PACK=0 switches it off.
    '''

    stems_on_split_pipe = os.environ.get('RCF_STEM_S2D', '1') != '0'
    '''
    fp32 training: the two 7x7 stride-2 stems as 4x4 convolutions on the space-to-depth image on the two-plane fp16 split kernels
    (Engine._conv_stem_s2d) instead of the f32-MFMA stem kernel.  RCF_STEM_S2D=0 switches it off.
    '''

    compute_dtype = 'fp32'
    '''
    'fp32' (default): the reference's arithmetic -- fp32 tensors, fp32-accurate products (ops.precision_of: the 3x3 / 2x2 split
    kernels on two scaled fp16 planes, or with RCF_FP32_TIER=3plane / 'fp32_3plane' on three bf16 planes; the other kernels on the
    f32 MFMA).  'f16x2' names the two-plane arithmetic explicitly.  'bf16': bf16 tensors in HBM and bf16 matrix operands,
    fp32 accumulation -- the "bf16" configurations of BASELINE.json; weights, BatchNorm statistics, loss and optimizer stay fp32.
    'bf16_operands': fp32 tensors, operands of the split convolution kernels rounded to bf16.
    '''

    def _run_engine(self, image, input_depth, record):
        if not image.is_cuda or not self._param_arena.is_cuda:
            raise _lib.RcfError('FusionNetModel.forward needs CUDA(HIP) tensors and a model on the GPU: the hot path is HIP-only, there is no CPU '
                                'path (inputs on %s, model on %s)' % (image.device, self._param_arena.device))
        if image.device != self._param_arena.device:
            raise _lib.RcfError('inputs live on %s but the model on %s' % (image.device, self._param_arena.device))
        ops.set_precision(ops.precision_of(self.compute_dtype))
        try:
            # every kernel is enqueued on the CURRENT stream of the CURRENT device: make that the model's device for the whole
            # call, whatever device the calling thread had selected
            with torch.cuda.device(self._param_arena.device):
                return self._run_engine_impl(image, input_depth, record)
        except BaseException:
            with torch.cuda.device(self._param_arena.device):   # the engine's streams belong to the model's device
                self._engine.recover()      # back on the caller's stream, side streams joined, no engine flag left set
            raise
        finally:
            ops.set_precision('fp32')

    def _run_engine_impl(self, image, input_depth, record):
        if not image.is_cuda:
            raise _lib.RcfError('FusionNetModel.forward needs CUDA(HIP) tensors: the hot path is HIP-only '
                                '(got %s)' % image.device)
        _lib.load()
        if image.dtype != torch.float32 or input_depth.dtype != torch.float32:
            raise _lib.RcfError('FusionNetModel.forward is fp32')
        training = self._training
        image, input_depth = image.contiguous(), input_depth.contiguous()
        hw = (int(image.shape[2]), int(image.shape[3]))
        self._engine.plan.active = False
        if record and training:   # every weight transform of the step in a few launches, up front (engine.WeightPlan)
            if self.batch_weight_packing:
                self._engine.plan.enable()
            self._engine.plan.begin()
        small_c = image.shape[1] <= 4 and input_depth.shape[1] <= 4
        if ops.act_dtype() == torch.bfloat16 and small_c:
            # bf16 configuration: the stems run on the space-to-depth image, built straight from the NCHW inputs; the fp32 NHWC
            # copies are only read by the stems' weight gradients
            s_img, s_dep = ops.s2d_image(image), ops.s2d_image(input_depth)
            x_img = ops.nchw_to_nhwc(image) if record else None
            x_dep = ops.nchw_to_nhwc(input_depth) if record else None
            out, tape = self._engine.forward(x_img, x_dep, training=training, record=record, image_s2d=s_img, depth_s2d=s_dep, hw=hw)
        elif ops.get_precision() == _lib.RCF_PREC_F16X2 and small_c and record and self.stems_on_split_pipe:
            # fp32 configuration on two fp16 planes (training): the same form in fp32 -- (image, max|image|) pairs; the NHWC copies feed
            # the stems' 7x7 weight gradients
            s_img, s_dep = ops.s2d_image_f32(image), ops.s2d_image_f32(input_depth)
            out, tape = self._engine.forward(ops.nchw_to_nhwc(image), ops.nchw_to_nhwc(input_depth), training=training, record=record,
                                             image_s2d=s_img, depth_s2d=s_dep, hw=hw)
        else:
            x_img = ops.nchw_to_nhwc(image)
            x_dep = ops.nchw_to_nhwc(input_depth)
            out, tape = self._engine.forward(x_img, x_dep, training=training, record=record)
        if training:
            self._nbt += 1
        return out, tape

    def _backward(self, out, tape, ddepth):
        used = self._used_params
        accumulate = any(p.grad is not None for p in used)
        prev = self._grad_arena[:self._n_used].clone() if accumulate else None
        if self._dp is not None:
            self._dp.begin_backward()
        self._engine.on_param_grad = self._dp.on_param_grad if self._dp is not None else None
        self._engine.completes_bucket = self._dp.completes_bucket if self._dp is not None else None
        ops.set_precision(ops.precision_of(self.compute_dtype))
        try:
            with torch.cuda.device(self._grad_arena.device):
                self._engine.in_backward = True
                Engine.backward(out, tape, ddepth)
                self._engine.side_join()
        except BaseException:
            with torch.cuda.device(self._param_arena.device):   # the engine's streams belong to the model's device
                self._engine.recover()
            raise
        finally:
            self._engine.in_backward = False
            ops.set_precision('fp32')
        self._engine.plan.end()
        if self._dp is not None:
            self._dp.finish_backward()
        if prev is not None:
            self._grad_arena[:self._n_used].add_(prev)
        for p in used:
            p.grad = self._grad_views[id(p)]

    def _all_reduce_loss_sums(self, sums):
        if self._dp is not None:
            self._dp.all_reduce_sums(sums)

    # ------------------------------------------------------------------ reference API
    def forward(self, image, input_depth, return_multiscale=False):
        '''
        Forwards the inputs through the network (src/fusionnet_model.py:140-170)

        Arg(s):
            image : torch.Tensor[float32]
                N x 3 x H x W image
            input_depth : torch.Tensor[float32]
                N x 2 x H x W input depth (cat([depth, response]), src/fusionnet_main.py:366)
            return_multiscale : bool
                if set, then return multiple outputs
        Returns:
            torch.Tensor[float32] : N x 1 x H x W output dense depth
        '''
        if torch.is_grad_enabled():
            output = _FusionNetFunction.apply(self, image, input_depth, self._anchor)
        else:
            out, _ = self._run_engine(image, input_depth, record=False)
            output = out.t.unsqueeze(1)
        if return_multiscale:
            return [output]
        return output

    def compute_loss(self,
                     image,
                     output_depth,
                     ground_truth,
                     lidar_map,
                     loss_func,
                     w_smoothness,
                     loss_smoothness_kernel_size,
                     validity_map_loss_smoothness,
                     w_lidar_loss):
        '''
        Computes loss function (src/fusionnet_model.py:172-302); returns (loss, loss_info).
        loss_func 'l1' (shipped, bash/train_fusionnet_nuscenes.sh:43-45), 'l2', 'smoothl1'; w_smoothness > 0 with
        loss_smoothness_kernel_size <= 1 adds the local smoothness term (src/fusionnet_losses.py:48-72); the Sobel variant raises.
        '''
        if isinstance(output_depth, list):
            if len(output_depth) != 1:
                raise ValueError('HIP path supports a single output resolution')
            output_depth = output_depth[0]
        if loss_func not in ('l1', 'l2', 'smoothl1'):
            raise ValueError('No such loss: {}'.format(loss_func))
        if w_smoothness > 0.0 and loss_smoothness_kernel_size > 1:
            raise ValueError('The Sobel smoothness loss (loss_smoothness_kernel_size > 1) is not implemented on the HIP path '
                             '(shipped: w_smoothness 0.0, loss_smoothness_kernel_size -1)')
        if w_lidar_loss > 0.0:
            lidar = lidar_map
        else:
            lidar = torch.zeros_like(ground_truth)   # no lidar term and no ground-truth masking (:214-221)
        loss, terms = _MaskedL1Function.apply(output_depth, ground_truth, lidar, float(max(w_lidar_loss, 0.0)), self, loss_func,
                                              image if w_smoothness > 0.0 else None, float(max(w_smoothness, 0.0)))
        loss_info = {
            'loss': loss,
            'loss_supervised': terms[1],
            'loss_smoothness': terms[3] if w_smoothness > 0.0 else 0.0,
            'loss_lidar': terms[2] if w_lidar_loss > 0.0 else 0.0,
        }
        return loss, loss_info

    def parameters(self):
        '''Returns the list of parameters in the model (src/fusionnet_model.py:304-316)'''
        return list(self.encoder.parameters()) + list(self.decoder.parameters())

    def train(self):
        '''Sets model to training mode (batch statistics, running stats updated)'''
        self.encoder.train()
        self.decoder.train()
        self._training = True

    def eval(self):
        '''Sets model to evaluation mode (running statistics)'''
        self.encoder.eval()
        self.decoder.eval()
        self._training = False

    def to(self, device):
        '''Moves model to specified device and (re)builds the flat parameter/gradient arenas'''
        self.device = torch.device(device)
        self.encoder.to(self.device)
        self.decoder.to(self.device)
        self._build_arenas()

    def _state_dicts(self):
        enc, dec = self.encoder.state_dict(), self.decoder.state_dict()
        if self._is_data_parallel:
            # the reference saves after data_parallel(), so its keys carry 'module.' 
            enc = {'module.' + k: v for k, v in enc.items()}
            dec = {'module.' + k: v for k, v in dec.items()}
        return enc, dec

    def save_model(self, checkpoint_path, step, optimizer):
        '''Save weights of the model to checkpoint path (src/fusionnet_model.py:347-368); same dictionary keys.'''
        enc, dec = self._state_dicts()
        checkpoint = {
            'train_step': step,
            'optimizer_state_dict': optimizer.state_dict(),
            'encoder_state_dict': {k: v.detach().clone() for k, v in enc.items()},
            'decoder_state_dict': {k: v.detach().clone() for k, v in dec.items()},
        }
        torch.save(checkpoint, checkpoint_path)

    def restore_model(self, checkpoint_path, optimizer=None):
        '''Restore weights of the model (src/fusionnet_model.py:370-393); accepts keys with or without 'module.'.'''
        checkpoint = torch.load(checkpoint_path, map_location=self.device)

        def strip(sd):
            return {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}

        self.encoder.load_state_dict(strip(checkpoint['encoder_state_dict']))
        self.decoder.load_state_dict(strip(checkpoint['decoder_state_dict']))
        if optimizer is not None:
            optimizer.load_state_dict(checkpoint['optimizer_state_dict'])
        return checkpoint['train_step'], optimizer

    def data_parallel(self):
        '''
        Allows multi-gpu split along batch (src/fusionnet_model.py:395-401).  The reference wraps encoder and
        decoder in single-process nn.DataParallel; here each GPU has its own process (torch.distributed over
        RCCL/xGMI) and this call arms the bucketed gradient all-reduce that overlaps with backward.  With one
        process it only switches the checkpoint key prefix, like the reference.
        '''
        self._is_data_parallel = True
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from .parallel import GradientBuckets
            self._dp = GradientBuckets(self)

    def capture_inference(self, image, input_depth, warmup=2, fold_once=False):
        '''
        Records one eval-mode forward (running-statistics BatchNorm, no tape) for inputs of this shape into a hipGraph and returns
        a callable `run(image, input_depth) -> N x 1 x H x W depth`: each call copies the inputs into the graph's static buffers
        and replays the ~370 kernel launches with one hipGraphLaunch.  The reference's inference loop (src/fusionnet_main.py:
        708-731, :814) calls model.forward per sample; `run` stands in for that call.  By default weight folding and packing are part
        of the recorded work, so a replay sees the parameters' current values (restore_model after capture is fine).
        fold_once=True freezes the weights at capture time instead: BatchNorm folding, phase weights and packing (~190 small
        launches, ~1.5 of 23 ms at batch 32) run once before the recording and the graph holds the convolutions only -- the
        deployment form (the reference's run() loads a checkpoint once, src/fusionnet_main.py:694-706); parameters changed
        afterwards need a new capture.  The output tensor is reused by the next replay -- clone it to keep it.
        '''
        if self._training:
            raise _lib.RcfError('capture_inference records the eval-mode forward: call eval() first')
        if not image.is_cuda:
            raise _lib.RcfError('capture_inference needs CUDA(HIP) tensors (got %s)' % image.device)
        static_image, static_depth = image.detach().clone(), input_depth.detach().clone()
        frozen = {} if fold_once else None
        self._engine.frozen = frozen
        try:
            side = torch.cuda.Stream(device=image.device)
            side.wait_stream(torch.cuda.current_stream(image.device))
            with torch.no_grad(), torch.cuda.stream(side):   # lazy one-time state (function attributes, zero page; with fold_once
                for _ in range(max(1, warmup)):              # also every weight transform) before recording
                    self.forward(static_image, static_depth)
            torch.cuda.current_stream(image.device).wait_stream(side)
            torch.cuda.synchronize(image.device)
            graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(graph):
                static_out = self.forward(static_image, static_depth)
        finally:
            self._engine.frozen = None

        def run(image, input_depth):
            if image.shape != static_image.shape or input_depth.shape != static_depth.shape:
                raise _lib.RcfError('captured for %s / %s, got %s / %s' % (tuple(static_image.shape), tuple(static_depth.shape),
                                                                          tuple(image.shape), tuple(input_depth.shape)))
            static_image.copy_(image, non_blocking=True)
            static_depth.copy_(input_depth, non_blocking=True)
            graph.replay()
            return static_out
        run.graph = graph
        run.frozen_weights = frozen     # fold_once: the folded / packed buffers the recorded launches read stay alive with the graph
        return run

    def capture_training_step(self, optimizer, image, input_depth, ground_truth, lidar_map, w_lidar_loss=2.0, outlier_removal=None,
                              warmup=2):
        '''
        Records ONE whole training step of the reference's loop body (src/fusionnet_main.py:369-399: forward -> outlier removal ->
        compute_loss -> zero_grad -> backward -> optimizer.step) for inputs of this shape into a single hipGraph and returns
        `step(image, input_depth, ground_truth, lidar_map) -> loss`: each call copies the batch into the graph's static buffers and
        replays the ~1100 kernel launches with one hipGraphLaunch (host time per step drops from ~12 ms of Python to well under
        1 ms).  Everything a replay must see fresh lives in device memory: parameters, gradients, Adam moments, the Adam step count
        and hyper-parameters (rcf_adam_step_dev), BatchNorm statistics.  Capturing has no side effect on the training state: the
        warm-up steps it needs are rolled back.  The optimizer must be rcf_amd.optim.FusedAdam.  Passing None for an input reuses
        what the static buffer holds.
        Under data parallelism (data_parallel() with torch.distributed initialised) the step is recorded as graph SEGMENTS between
        its exchange points -- the all-reduce of the loss sums, each gradient bucket, the wait before Adam (parallel.SegmentedCapture)
        -- and a replay launches segment, RCCL call, segment, ... : about eight hipGraphLaunch + seven collectives of host work per
        step instead of ~1100 launches, the buckets overlapping the next segment as in the eager step.  Every rank must capture
        (the warm-up steps run the real collectives) and replay in step with the others.
        '''
        from . import train
        from .optim import FusedAdam
        if not isinstance(optimizer, FusedAdam):
            raise _lib.RcfError('capture_training_step needs rcf_amd.optim.FusedAdam (step count and hyper-parameters on the device)')
        if not self._training:
            raise _lib.RcfError('capture_training_step records the train-mode step: call train() first')
        if not image.is_cuda:
            raise _lib.RcfError('capture_training_step needs CUDA(HIP) tensors (got %s)' % image.device)
        dev = image.device
        static = [t.detach().clone() for t in (image, input_depth, ground_truth, lidar_map)]

        def one():
            return train.train_step(self, optimizer, static[0], static[1], static[2], static[3], w_lidar_loss=w_lidar_loss,
                                    outlier_removal=outlier_removal)[0]

        # ---- snapshot of everything a training step changes
        buffers = [b for mod in (self.encoder, self.decoder) for b in mod.buffers()]
        snap = {'params': self._param_arena.clone(), 'buffers': [b.clone() for b in buffers], 'nbt': self._nbt.clone(),
                'moments': {k: (m.clone(), v.clone()) for k, (m, v) in optimizer._moment_arenas.items()},
                'steps': {k: float(t) for k, t in optimizer._shared_steps.items()}, 'had_state': len(optimizer.state) > 0}
        prof, self._engine.prof = self._engine.prof, None

        def roll_back():
            with torch.no_grad():
                self._param_arena.copy_(snap['params'])
                for b, saved in zip(buffers, snap['buffers']):
                    b.copy_(saved)
                self._nbt.copy_(snap['nbt'])
                for k, (m, v) in optimizer._moment_arenas.items():
                    if k in snap['moments']:
                        m.copy_(snap['moments'][k][0]); v.copy_(snap['moments'][k][1])
                    else:
                        m.zero_(); v.zero_()
                for k, t in optimizer._shared_steps.items():
                    t.fill_(snap['steps'].get(k, 0.0))
                for k, ds in optimizer._dev_state.items():   # k = (id(param arena), lo, hi)
                    step0 = snap['steps'].get(k[0], 0.0)
                    ds[0][0:1].fill_(step0)
                    ds[2] = int(step0)

        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):   # lazy one-time state (function attributes, allocator pools, optimizer state) before recording
                for _ in range(max(1, warmup)):
                    one()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            if self._dp is None:
                segmented = None
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    static_loss = one()
            else:
                import gc
                from .parallel import SegmentedCapture
                graph = None
                segmented = SegmentedCapture()
                gc.collect()
                torch.cuda.empty_cache()
                cap_stream = torch.cuda.Stream(device=dev)
                cap_stream.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(cap_stream):
                    self._dp.capture = segmented
                    try:
                        segmented.begin()
                        static_loss = one()
                        segmented.end()
                    finally:
                        self._dp.capture = None
                torch.cuda.current_stream(dev).wait_stream(cap_stream)
        finally:
            # whether or not the capture succeeded: the profiler hook back, the warm-up steps rolled back (the capture itself executed
            # nothing), so a caller that falls back to eager launches continues from the state it had
            self._engine.prof = prof
            torch.cuda.synchronize(dev)
            roll_back()
        torch.cuda.synchronize(dev)
        shapes = [tuple(t.shape) for t in static]

        def step(image=None, input_depth=None, ground_truth=None, lidar_map=None):
            for dst, src, shp in zip(static, (image, input_depth, ground_truth, lidar_map), shapes):
                if src is None or src.data_ptr() == dst.data_ptr():
                    continue
                if tuple(src.shape) != shp:
                    raise _lib.RcfError('captured for %s, got %s' % (shp, tuple(src.shape)))
                dst.copy_(src, non_blocking=True)
            optimizer.sync_hyper_parameters()   # a learning-rate schedule between replays reaches the recorded Adam launch
            if segmented is None:
                graph.replay()
            else:
                segmented.replay(self._dp)
            optimizer.note_replayed_step()
            return static_loss
        step.graph = graph
        step.segments = None if segmented is None else segmented.segments
        step.static_inputs = static
        # the recorded launches read and write the weight plan's persistent buffers: keep THESE alive with the graph even if the
        # engine later re-records its plan (another input size run eagerly in between)
        step.weight_plan_buffers = list(self._engine.plan.entries)
        return step

    def log_summary(self, summary_writer, tag, step, image=None, input_depth=None, input_response=None,
                    output_depth=None, ground_truth=None, scalars={}, n_display=4):
        '''
        Logs summary (src/fusionnet_model.py:403-587).  Scalars are written when the writer exposes add_scalar;
        the image grids need torchvision/matplotlib which this image lacks, so they are skipped.
        '''
        if summary_writer is None or not hasattr(summary_writer, 'add_scalar'):
            return
        with torch.no_grad():
            for name, value in scalars.items():
                summary_writer.add_scalar(tag + '_' + name, float(value), global_step=step)
            if output_depth is not None and hasattr(summary_writer, 'add_histogram'):
                summary_writer.add_histogram(tag + '_output_depth_distro', output_depth.detach().cpu(), global_step=step)
