'''
RadarNetModel -- drop-in for the reference's src/radarnet_model.py:7-330 (stage 1 of the method: which image pixels does a
radar point correspond to) on MI355X.  SURVEY.md 8 f-1.

Same constructor arguments and methods (forward, compute_loss, parameters, train, eval, to, save_model, restore_model,
data_parallel), same checkpoint dictionary ('radarnet_encoder_state_dict', ...) and state_dict names.  The image encoder,
the decoder and the output convolution run on the FusionNet kernels through engine.py; ROI pooling, the fully connected radar
branch and the masked BCE loss are csrc/rcf_radarnet.hip.  There is no CPU path.

Parity note: torchvision.ops.roi_pool is absent from this image and from /root/reference; the HIP kernel follows a restatement
of torchvision 0.11's published kernel that lives with the test infrastructure: parity is unpinned at that
boundary, everything else is pinned against the imported reference.
'''

import torch

from . import _lib, networks, ops
from .engine import Engine
from .fusionnet_model import FusionNetModel


class _RadarNetFunction(torch.autograd.Function):
    '''One autograd node for encoder + ROI pooling + radar branch + decoder.'''

    @staticmethod
    def forward(ctx, model, image, point, rois, anchor):
        out, tape = model._run_engine(image, point, rois, record=True)
        ctx.model, ctx.out, ctx.tape = model, out, tape
        return out.t.unsqueeze(1)

    @staticmethod
    def backward(ctx, grad_output):
        model, out, tape = ctx.model, ctx.out, ctx.tape
        if tape is None:
            raise RuntimeError('RadarNetModel: backward through the same forward twice')
        ctx.tape = None
        m, _, h, w = grad_output.shape
        model._backward(out, tape, grad_output.contiguous().view(m, h, w))
        return None, None, None, None, None


class _MaskedBCEFunction(torch.autograd.Function):
    '''compute_loss (src/radarnet_model.py:131-171) on the HIP loss kernels.'''

    @staticmethod
    def forward(ctx, logits, ground_truth, validity_map, w_positive_class, model):
        x = logits.contiguous()
        t = ground_truth.contiguous().to(torch.float32)
        v = validity_map.contiguous().to(torch.float32)
        sums = torch.empty(2, dtype=torch.float64, device=x.device)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            ops.bce_loss_fwd(x, t, v, sums, loss, float(w_positive_class))
        if model is not None and model._dp is not None:
            # one masked mean over the global batch, like the reference's single-process DataParallel
            model._dp.all_reduce_sums(sums)
            loss = (sums[0] / sums[1]).to(torch.float32).view(1)
        ctx.save_for_backward(x, t, v, sums)
        ctx.w = float(w_positive_class)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, grad_loss):
        x, t, v, sums = ctx.saved_tensors
        dx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            ops.bce_loss_bwd(x, t, v, sums, grad_loss.contiguous().view(1).to(torch.float32), dx, ctx.w)
        return dx, None, None, None, None


class RadarNetModel(object):
    '''
    Image radar fusion to determine correspondence of radar to image (see the reference docstring,
    src/radarnet_model.py:8-34, for the argument meanings).
    '''

    def __init__(self,
                 input_channels_image,
                 input_channels_depth,
                 input_patch_size_image,
                 encoder_type,
                 n_filters_encoder_image,
                 n_neurons_encoder_depth,
                 decoder_type,
                 n_filters_decoder,
                 weight_initializer='kaiming_uniform',
                 activation_func='leaky_relu',
                 device=torch.device('cuda')):

        self.input_patch_size_image = input_patch_size_image
        self.device = torch.device(device)

        height, width = input_patch_size_image
        latent_height = int((height // 32.0))
        latent_width = int((width // 32.0))
        latent_size_depth = latent_height * latent_width * n_neurons_encoder_depth[-1]      # src/radarnet_model.py:56-60

        if 'radarnetv1' in encoder_type:
            self.encoder = networks.RadarNetV1Encoder(
                input_channels_image=input_channels_image,
                input_channels_depth=input_channels_depth,
                input_patch_size_image=input_patch_size_image,
                n_filters_encoder_image=n_filters_encoder_image,
                n_neurons_encoder_depth=n_neurons_encoder_depth,
                latent_size_depth=latent_size_depth,
                weight_initializer=weight_initializer,
                activation_func=activation_func,
                use_batch_norm='batch_norm' in encoder_type)
        else:
            raise ValueError('Encoder type {} not supported.'.format(encoder_type))

        n_skips = n_filters_encoder_image[:-1]
        n_skips = n_skips[::-1] + [0]                                                         # :78-79
        latent_channels = n_filters_encoder_image[-1] + n_neurons_encoder_depth[-1]

        if 'multiscale' in decoder_type:
            self.decoder = networks.MultiScaleDecoder(
                input_channels=latent_channels,
                output_channels=1,
                n_resolution=1,
                n_filters=n_filters_decoder,
                n_skips=n_skips,
                weight_initializer=weight_initializer,
                activation_func=activation_func,
                output_func='linear',
                use_batch_norm='batch_norm' in decoder_type,
                deconv_type='up')
        else:
            raise ValueError('Decoder type {} not supported.'.format(decoder_type))

        if not ('batch_norm' in encoder_type and 'batch_norm' in decoder_type):
            raise ValueError('HIP path implements the shipped batch_norm encoder/decoder only')
        if len(n_filters_encoder_image) != 5:
            raise ValueError('HIP path implements the shipped 5-level RadarNet encoder')

        self._is_data_parallel = False
        self._dp = None
        self._training = True
        self._anchor = None
        self._engine = Engine(self.encoder, self.decoder, 1.0, 100.0)
        self._engine.grad_of = self._grad_of
        self._param_arena = None
        self._grad_arena = None
        self.to(self.device)

    # ------------------------------------------------------------------ arenas (shared implementation with FusionNetModel)
    def _forward_order_params(self):
        enc, dec = self.encoder, self.decoder
        order = []

        def conv_block(layer):
            order.append(layer.conv.weight)
            if layer.use_batch_norm:
                order.extend([layer.batch_norm.weight, layer.batch_norm.bias])

        ei = enc.encoder_image
        conv_block(ei.conv1)
        for lvl in range(2, ei.network_depth + 1):
            for b in getattr(ei, 'blocks%d' % lvl):
                conv_block(b.conv1)
                if b.uses_projection:
                    conv_block(b.projection)
                conv_block(b.conv2)
        for fc in enc.encoder_depth.mlp:
            order.extend([fc.fully_connected.weight, fc.fully_connected.bias])
        for name in dec.block_names:
            blk = getattr(dec, name)
            conv_block(blk.deconv if blk.deconv_type == 'transpose' else blk.deconv.conv)
            conv_block(blk.conv)
        conv_block(dec.output0)
        return order

    _build_arenas = FusionNetModel._build_arenas
    _grad_of = FusionNetModel._grad_of
    _backward = FusionNetModel._backward

    # ------------------------------------------------------------------ engine entry point
    compute_dtype = 'fp32'   # or 'bf16': see FusionNetModel.compute_dtype

    def _run_engine(self, image, point, rois, record):
        if not image.is_cuda or not self._param_arena.is_cuda:
            raise _lib.RcfError('RadarNetModel.forward needs CUDA(HIP) tensors and a model on the GPU: the hot path is HIP-only, there is no CPU '
                                'path (inputs on %s, model on %s)' % (image.device, self._param_arena.device))
        if image.device != self._param_arena.device:
            raise _lib.RcfError('inputs live on %s but the model on %s' % (image.device, self._param_arena.device))
        ops.set_precision(ops.precision_of(self.compute_dtype))
        try:
            with torch.cuda.device(self._param_arena.device):   # kernels go to the current stream of the current device
                return self._run_engine_impl(image, point, rois, record)
        except BaseException:
            with torch.cuda.device(self._param_arena.device):   # the engine's streams belong to the model's device
                self._engine.recover()
            raise
        finally:
            ops.set_precision('fp32')

    def _run_engine_impl(self, image, point, rois, record):
        if not image.is_cuda:
            raise _lib.RcfError('RadarNetModel.forward needs CUDA(HIP) tensors: the hot path is HIP-only (got %s)' % image.device)
        _lib.load()
        if image.dtype != torch.float32:
            raise _lib.RcfError('RadarNetModel.forward is fp32')
        image = image.contiguous()
        pts = point.contiguous().to(torch.float32)
        if ops.act_dtype() == torch.bfloat16 and image.shape[1] <= 4:   # bf16 configuration: stem on the space-to-depth image
            out, tape = self._engine.forward_radarnet(ops.nchw_to_nhwc(image) if record else None, pts, rois, training=self._training,
                                                      record=record, image_s2d=ops.s2d_image(image),
                                                      hw=(int(image.shape[2]), int(image.shape[3])))
        else:
            out, tape = self._engine.forward_radarnet(ops.nchw_to_nhwc(image), pts, rois, training=self._training, record=record)
        if self._training:
            self._nbt += 1
        return out, tape

    # ------------------------------------------------------------------ reference API
    def forward(self, image, point, bounding_boxes, return_logits=True):
        '''
        Forwards the inputs through the network (src/radarnet_model.py:102-129)

        Arg(s):
            image : torch.Tensor[float32]
                N x 3 x H x W image
            point : torch.Tensor[float32]
                (N*K) x 3 input points
            bounding_boxes : list[torch.Tensor[float32]]
                N tensors of K x 4 boxes (x1, y1, x2, y2) in image coordinates, one per point (src/networks.py:1205-1206)
            return_logits : bool
                if set, then return logits otherwise sigmoid
        Returns:
            torch.Tensor[float32] : (N*K) x 1 x patch_height x patch_width logits (correspondence map)
        '''
        if isinstance(bounding_boxes, torch.Tensor) and bounding_boxes.dim() == 3:
            bounding_boxes = [bounding_boxes[i] for i in range(bounding_boxes.shape[0])]
        rois = torch.cat([torch.cat([torch.full((b.shape[0], 1), float(i), dtype=torch.float32, device=b.device),
                                     b.to(torch.float32)], dim=1) for i, b in enumerate(bounding_boxes)], dim=0).contiguous()
        rois = rois.to(image.device)
        if rois.shape[0] != point.shape[0]:
            raise ValueError('one bounding box per radar point is required')
        if torch.is_grad_enabled():
            logits = _RadarNetFunction.apply(self, image, point, rois, self._anchor)
        else:
            out, _ = self._run_engine(image, point, rois, record=False)
            logits = out.t.unsqueeze(1)
        if return_logits:
            return logits
        return torch.sigmoid(logits)

    def compute_loss(self, logits, ground_truth, validity_map, w_positive_class=1.0):
        '''Computes loss function (src/radarnet_model.py:131-171); returns (loss, loss_info).'''
        loss = _MaskedBCEFunction.apply(logits, ground_truth, validity_map, float(w_positive_class), self)
        return loss, {'loss': loss}

    def parameters(self):
        '''Returns the list of parameters in the model (src/radarnet_model.py:173-185)'''
        return list(self.encoder.parameters()) + list(self.decoder.parameters())

    def train(self):
        self.encoder.train()
        self.decoder.train()
        self._training = True

    def eval(self):
        self.encoder.eval()
        self.decoder.eval()
        self._training = False

    def to(self, device):
        '''Moves model to specified device and (re)builds the flat parameter/gradient arenas'''
        self.device = torch.device(device)
        self.encoder.to(self.device)
        self.decoder.to(self.device)
        self._build_arenas()

    def save_model(self, checkpoint_path, step, optimizer):
        '''Save weights of the model to checkpoint path (src/radarnet_model.py:213-236); same dictionary keys.'''
        enc, dec = self.encoder.state_dict(), self.decoder.state_dict()
        if self._is_data_parallel:
            enc = {'module.' + k: v for k, v in enc.items()}
            dec = {'module.' + k: v for k, v in dec.items()}
        checkpoint = {
            'train_step': step,
            'radarnet_optimizer_state_dict': optimizer.state_dict(),
            'radarnet_encoder_state_dict': {k: v.detach().clone() for k, v in enc.items()},
            'radarnet_decoder_state_dict': {k: v.detach().clone() for k, v in dec.items()},
        }
        torch.save(checkpoint, checkpoint_path)

    def restore_model(self, checkpoint_path, optimizer=None):
        '''Restore weights of the model (src/radarnet_model.py:238-262); accepts keys with or without 'module.'.'''
        checkpoint = torch.load(checkpoint_path, map_location=self.device)

        def strip(sd):
            return {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}

        self.encoder.load_state_dict(strip(checkpoint['radarnet_encoder_state_dict']))
        self.decoder.load_state_dict(strip(checkpoint['radarnet_decoder_state_dict']))
        if optimizer is not None:
            optimizer.load_state_dict(checkpoint['radarnet_optimizer_state_dict'])
        return checkpoint['train_step'], optimizer

    def data_parallel(self):
        '''Multi-GPU split along the batch (src/radarnet_model.py:264-270): one process per GPU, bucketed RCCL all-reduce.'''
        self._is_data_parallel = True
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from .parallel import GradientBuckets
            self._dp = GradientBuckets(self)
