'''
Host-side mirror of the reference's block classes (src/net_utils.py): same constructor arguments, same
sub-module and parameter names, hence the same state_dict keys -- so the reference's checkpoints load and
torch.optim.Adam / state_dict work unchanged.

These modules are PARAMETER CONTAINERS.  They do not compute: the forward/backward of the whole network
runs in engine.py on hand-written HIP kernels (csrc/).  Calling a block's forward() directly raises.
'''

import torch

LEAKY_SLOPE = 0.20


def activation_func(activation_fn):
    '''
    src/net_utils.py:4-23.  Returns the activation NAME the engine understands (the reference returns a
    torch module); raises ValueError for unsupported names exactly like the reference.
    '''
    if 'linear' in activation_fn:
        return None
    elif 'leaky_relu' in activation_fn:
        return 'leaky_relu'
    elif 'sigmoid' in activation_fn:
        return 'sigmoid'
    elif 'relu' in activation_fn or 'elu' in activation_fn:
        raise ValueError('Unsupported activation function on the HIP path: {}'.format(activation_fn))
    else:
        raise ValueError('Unsupported activation function: {}'.format(activation_fn))


class _NoForward(torch.nn.Module):
    def forward(self, *args, **kwargs):
        raise RuntimeError(
            type(self).__name__ + ' is a parameter container; run the network through FusionNetModel.forward '
            '(HIP engine). There is no per-block or CPU forward.')


class Conv2d(_NoForward):
    '''src/net_utils.py:29-91: conv(bias=False, padding=k//2) + optional BatchNorm2d + optional activation.'''

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1,
                 weight_initializer='kaiming_uniform', activation_func='leaky_relu', use_batch_norm=False):
        super(Conv2d, self).__init__()
        self.use_batch_norm = use_batch_norm
        self.kernel_size = kernel_size
        self.stride = stride
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.conv = torch.nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                                    padding=kernel_size // 2, bias=False)
        # src/net_utils.py:72-77 ('kaiming_uniform' keeps torch.nn.Conv2d's default initialisation)
        if weight_initializer == 'kaiming_normal':
            torch.nn.init.kaiming_normal_(self.conv.weight)
        elif weight_initializer == 'xavier_normal':
            torch.nn.init.xavier_normal_(self.conv.weight)
        elif weight_initializer == 'xavier_uniform':
            torch.nn.init.xavier_uniform_(self.conv.weight)
        self.activation_func = activation_func   # None | 'leaky_relu' | 'sigmoid'
        if self.use_batch_norm:
            self.batch_norm = torch.nn.BatchNorm2d(out_channels)


class UpConv2d(_NoForward):
    '''src/net_utils.py:156-198: nearest interpolate to `shape` + Conv2d 3x3.'''

    def __init__(self, in_channels, out_channels, kernel_size=3, weight_initializer='kaiming_uniform',
                 activation_func='leaky_relu', use_batch_norm=False):
        super(UpConv2d, self).__init__()
        self.conv = Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=1,
                           weight_initializer=weight_initializer, activation_func=activation_func,
                           use_batch_norm=use_batch_norm)


class TransposeConv2d(_NoForward):
    '''src/net_utils.py:94-153: ConvTranspose2d(k, stride 2, padding k // 2, output_padding 1, bias=False) + optional BatchNorm2d
    + optional activation.  State-dict names as in the reference: `deconv.weight` ([in, out, k, k]), `batch_norm.*`.

    The reference class only works with its default initializer: for the others it touches `self.conv`, which does not exist
    (src/net_utils.py:135-140, AttributeError).  Here those initializers are applied to the transposed-convolution weight --
    what the code evidently meant.  On the engine the forward is the 4-phase transposed convolution that already serves as the
    input gradient of the stride-2 convolutions; its own gradients are a stride-2 convolution (dX) and a stride-2 weight gradient.'''

    def __init__(self, in_channels, out_channels, kernel_size=3, weight_initializer='kaiming_uniform',
                 activation_func='leaky_relu', use_batch_norm=False):
        super(TransposeConv2d, self).__init__()
        if kernel_size != 3:
            raise ValueError('Transposed convolution on the HIP path is 3x3 (the only size DecoderBlock builds, src/net_utils.py:510)')
        self.use_batch_norm = use_batch_norm
        self.kernel_size = kernel_size
        self.stride = 2
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.deconv = torch.nn.ConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size, stride=2,
                                               padding=kernel_size // 2, output_padding=1, bias=False)
        if weight_initializer == 'kaiming_normal':
            torch.nn.init.kaiming_normal_(self.deconv.weight)
        elif weight_initializer == 'xavier_normal':
            torch.nn.init.xavier_normal_(self.deconv.weight)
        elif weight_initializer == 'xavier_uniform':
            torch.nn.init.xavier_uniform_(self.deconv.weight)
        self.activation_func = activation_func
        if self.use_batch_norm:
            self.batch_norm = torch.nn.BatchNorm2d(out_channels)

    @property
    def conv(self):
        '''the module holding `.weight`, under the name the engine uses for every convolution block'''
        return self.deconv


class ResNetBlock(_NoForward):
    '''src/net_utils.py:253-323.  `projection` (1x1, no BN, no activation) is always allocated (:300-307) and only
    used when the block changes shape (:317-320) -- 10 of them never receive gradients in the published net.'''

    def __init__(self, in_channels, out_channels, stride=1, weight_initializer='kaiming_uniform',
                 activation_func='leaky_relu', use_batch_norm=False):
        super(ResNetBlock, self).__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.stride = stride
        self.activation_func = activation_func
        self.conv1 = Conv2d(in_channels, out_channels, 3, stride, weight_initializer, activation_func, use_batch_norm)
        self.conv2 = Conv2d(out_channels, out_channels, 3, 1, weight_initializer, activation_func, use_batch_norm)
        self.projection = Conv2d(in_channels, out_channels, 1, stride, weight_initializer, None, False)

    @property
    def uses_projection(self):
        return self.stride != 1 or self.in_channels != self.out_channels


class DecoderBlock(_NoForward):
    '''src/net_utils.py:473-569.  deconv_type 'up' is what the shipped entry points hard-code (src/fusionnet_main.py:190);
    'transpose' (src/net_utils.py:507-513) needs every level to be exactly 2x the previous one, i.e. input sizes divisible by 64
    such as the 448 x 448 training crop -- at 900 x 1600 the reference itself fails in torch.cat.'''

    def __init__(self, in_channels, skip_channels, out_channels, weight_initializer='kaiming_uniform',
                 activation_func='leaky_relu', use_batch_norm=False, deconv_type='up'):
        super(DecoderBlock, self).__init__()
        self.skip_channels = skip_channels
        self.deconv_type = deconv_type
        if deconv_type == 'up':
            self.deconv = UpConv2d(in_channels, out_channels, 3, weight_initializer, activation_func, use_batch_norm)
        elif deconv_type == 'transpose':
            self.deconv = TransposeConv2d(in_channels, out_channels, 3, weight_initializer, activation_func, use_batch_norm)
        else:
            # the reference leaves self.deconv undefined for any other value (AttributeError at the first forward)
            raise ValueError('Unsupported deconv type: {}'.format(deconv_type))
        self.conv = Conv2d(skip_channels + out_channels, out_channels, 3, 1, weight_initializer, activation_func,
                           use_batch_norm)


class OutlierRemoval(object):
    '''
    Class to perform outlier removal based on depth difference in local neighborhood (src/net_utils.py:575-638); same
    constructor and method as the reference, computed by rcf_outlier_removal (one global-max pass + one LDS-tiled
    min-filter pass; no host synchronisation).
    '''

    def __init__(self, kernel_size=7, threshold=1.5):
        self.kernel_size = kernel_size
        self.threshold = threshold

    def remove_outliers(self, depth):
        from . import ops
        return ops.outlier_removal(depth, self.kernel_size, self.threshold)


class FullyConnected(_NoForward):
    '''src/net_utils.py:201-247: torch.nn.Linear (with bias) + activation; dropout is not on the shipped path.'''

    def __init__(self, in_features, out_features, weight_initializer='kaiming_uniform', activation_func='leaky_relu',
                 dropout_rate=0.00):
        super(FullyConnected, self).__init__()
        if dropout_rate > 0.00:
            raise ValueError('Dropout is not implemented on the HIP path')
        self.fully_connected = torch.nn.Linear(in_features, out_features)
        if weight_initializer == 'kaiming_normal':
            torch.nn.init.kaiming_normal_(self.fully_connected.weight)
        elif weight_initializer == 'xavier_normal':
            torch.nn.init.xavier_normal_(self.fully_connected.weight)
        elif weight_initializer == 'xavier_uniform':
            torch.nn.init.xavier_uniform_(self.fully_connected.weight)
        self.activation_func = activation_func   # None | 'leaky_relu'
