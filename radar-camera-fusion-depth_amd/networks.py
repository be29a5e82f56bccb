'''
Host-side mirror of the reference's FusionNetEncoder and MultiScaleDecoder (src/networks.py:270-1005,
:1337-1657): same constructor arguments, attribute names and state_dict keys.  Parameter containers only --
the computation is engine.py + csrc/.
'''

import torch

from . import net_utils


class FusionNetEncoder(net_utils._NoForward):
    '''
    src/networks.py:270-1005.  Two independent ResNet branches (image, depth); at each of the `network_depth`
    levels the fused tensor  sigmoid(BN(W1 d)) * BN(W2 d) + img  is emitted as a skip / the latent
    ('weight_and_project', :863-866).  The branches never see the fused tensors.
    '''

    def __init__(self, n_layer=18, input_channels_image=3, input_channels_depth=3,
                 n_filters_encoder_image=[32, 64, 128, 256, 256], n_filters_encoder_depth=[32, 64, 128, 256, 256],
                 weight_initializer='kaiming_uniform', activation_func='leaky_relu', use_batch_norm=False,
                 fusion_type='add'):
        super(FusionNetEncoder, self).__init__()
        self.fusion_type = fusion_type
        if n_layer == 18:
            n_blocks = [2, 2, 2, 2]
        elif n_layer == 34:
            n_blocks = [3, 4, 6, 3]
        else:
            raise ValueError('Only supports 18, 34 layer architecture')          # :311
        if fusion_type != 'weight_and_project':
            # the reference also has 'add', 'weight', 'concat' (:350-389); only the shipped flag
            # (bash/train_fusionnet_nuscenes.sh:33) has a HIP fusion kernel
            raise ValueError('Unsupported fusion type on the HIP path: {}'.format(fusion_type))
        assert len(n_filters_encoder_image) == len(n_filters_encoder_depth)
        for n in range(len(n_filters_encoder_image) - len(n_blocks) - 1):            # :317-318
            n_blocks = n_blocks + [n_blocks[-1]]
        network_depth = len(n_filters_encoder_image)
        assert network_depth < 8, 'Does not support network depth of 8 or more'     # :322
        assert network_depth == len(n_blocks) + 1
        self.network_depth = network_depth
        act = net_utils.activation_func(activation_func)
        fi, fd = list(n_filters_encoder_image), list(n_filters_encoder_depth)
        bn = use_batch_norm
        wi = weight_initializer

        self.conv1_image = net_utils.Conv2d(input_channels_image, fi[0], 7, 2, wi, act, bn)     # :332
        self.conv1_depth = net_utils.Conv2d(input_channels_depth, fd[0], 7, 2, wi, act, bn)     # :341
        self.conv1_weight = net_utils.Conv2d(fd[0], fi[0], 1, 1, wi, 'sigmoid', bn)             # :373
        self.conv1_project = net_utils.Conv2d(fd[0], fi[0], 1, 1, wi, None, bn)                 # :382
        for lvl in range(2, 8):
            if lvl <= network_depth:
                stride = 1 if lvl == 2 else 2                                                   # :414, :479
                nb = n_blocks[lvl - 2]
                ci, co = fi[lvl - 2], fi[lvl - 1]
                di, do = fd[lvl - 2], fd[lvl - 1]
                img_blocks, dep_blocks = [], []
                for b in range(nb):                                                             # _make_layer :767-838
                    s = stride if b == 0 else 1
                    img_blocks.append(net_utils.ResNetBlock(ci if b == 0 else co, co, s, wi, act, bn))
                    dep_blocks.append(net_utils.ResNetBlock(di if b == 0 else do, do, s, wi, act, bn))
                setattr(self, 'blocks%d_image' % lvl, torch.nn.Sequential(*img_blocks))
                setattr(self, 'blocks%d_depth' % lvl, torch.nn.Sequential(*dep_blocks))
                setattr(self, 'conv%d_weight' % lvl, net_utils.Conv2d(do, co, 1, 1, wi, 'sigmoid', bn))
                setattr(self, 'conv%d_project' % lvl, net_utils.Conv2d(do, co, 1, 1, wi, None, bn))
            elif lvl >= 6:                                                                      # :710-714, :761-765
                setattr(self, 'blocks%d_image' % lvl, None)
                setattr(self, 'blocks%d_depth' % lvl, None)
                setattr(self, 'conv%d_weight' % lvl, None)
                setattr(self, 'conv%d_project' % lvl, None)


class MultiScaleDecoder(net_utils._NoForward):
    '''src/networks.py:1337-1657 on the shipped path: n_resolution=1, output_func='linear', deconv_type='up'.'''

    def __init__(self, input_channels=256, output_channels=1, n_resolution=1, n_filters=[256, 128, 64, 32, 16],
                 n_skips=[256, 128, 64, 32, 0], weight_initializer='kaiming_uniform', activation_func='leaky_relu',
                 output_func='linear', use_batch_norm=False, deconv_type='up'):
        super(MultiScaleDecoder, self).__init__()
        network_depth = len(n_filters)
        assert network_depth < 8, 'Does not support network depth of 8 or more'     # :1378
        assert n_resolution > 0 and n_resolution < network_depth                  # :1379
        if n_resolution != 1 or output_channels != 1 or 'linear' not in output_func:
            raise ValueError('HIP path supports n_resolution=1, output_channels=1, linear output only')
        if network_depth < 5:
            raise ValueError('HIP path supports decoder depth 5..7')
        self.n_resolution = n_resolution
        self.output_func = output_func
        act = net_utils.activation_func(activation_func)
        self.block_names = []
        cin = input_channels
        for i in range(7):
            name = 'deconv%d' % (6 - i)
            if 6 - i >= network_depth:
                setattr(self, name, None)                                            # :1414, :1433
                continue
            idx = len(self.block_names)
            block = net_utils.DecoderBlock(cin, n_skips[idx], n_filters[idx], weight_initializer, act, use_batch_norm,
                                           deconv_type)
            setattr(self, name, block)
            self.block_names.append(name)
            cin = n_filters[idx]
        self.output0 = net_utils.Conv2d(cin, output_channels, 3, 1, weight_initializer, None, False)   # :1548-1555


class ResNetEncoder(net_utils._NoForward):
    '''src/networks.py:8-268 (n_layer 18 / 34): conv1 7x7 s2, max_pool, blocks2..blocks7; same attribute names and state_dict keys.'''

    def __init__(self, n_layer, input_channels=3, n_filters=[32, 64, 128, 256, 256], weight_initializer='kaiming_uniform',
                 activation_func='leaky_relu', use_batch_norm=False):
        super(ResNetEncoder, self).__init__()
        if n_layer == 18:
            n_blocks = [2, 2, 2, 2]
        elif n_layer == 34:
            n_blocks = [3, 4, 6, 3]
        else:
            raise ValueError('Only supports 18, 34 layer architecture')           # :40
        for n in range(len(n_filters) - len(n_blocks) - 1):                        # :42-43
            n_blocks = n_blocks + [n_blocks[-1]]
        network_depth = len(n_filters)
        assert network_depth < 8, 'Does not support network depth of 8 or more'    # :47
        assert network_depth == len(n_blocks) + 1
        self.network_depth = network_depth
        act = net_utils.activation_func(activation_func)
        f = list(n_filters)
        self.conv1 = net_utils.Conv2d(input_channels, f[0], 7, 2, weight_initializer, act, use_batch_norm)   # :61
        for lvl in range(2, 8):
            if lvl <= network_depth:
                stride = 1 if lvl == 2 else 2
                blocks = []
                for b in range(n_blocks[lvl - 2]):                                  # _make_layer :175-228
                    blocks.append(net_utils.ResNetBlock(f[lvl - 2] if b == 0 else f[lvl - 1], f[lvl - 1], stride if b == 0 else 1,
                                                        weight_initializer, act, use_batch_norm))
                setattr(self, 'blocks%d' % lvl, torch.nn.Sequential(*blocks))
            elif lvl >= 6:
                setattr(self, 'blocks%d' % lvl, None)                               # :146, :165


class FullyConnectedEncoder(net_utils._NoForward):
    '''src/networks.py:1007-1067: six FullyConnected layers input -> n_neurons[0..4] -> latent_size, all with the activation.'''

    def __init__(self, input_channels=3, n_neurons=[32, 64, 96, 128, 256], latent_size=29 * 10,
                 weight_initializer='kaiming_uniform', activation_func='leaky_relu'):
        super(FullyConnectedEncoder, self).__init__()
        act = net_utils.activation_func(activation_func)
        sizes = [input_channels] + list(n_neurons[:5]) + [latent_size]
        self.mlp = torch.nn.Sequential(*[net_utils.FullyConnected(sizes[i], sizes[i + 1], weight_initializer, act)
                                         for i in range(6)])


class RadarNetV1Encoder(net_utils._NoForward):
    '''src/networks.py:1151-1256: ResNet-18 image encoder, ROI pooling of its latent and skips around each radar point, and a
    fully connected encoder of the point whose output is reshaped to the latent's size and concatenated to it.'''

    def __init__(self, input_channels_image=3, input_channels_depth=3, input_patch_size_image=(900, 288),
                 n_filters_encoder_image=[32, 64, 128, 128, 128], n_neurons_encoder_depth=[32, 64, 128, 128, 128],
                 latent_size_depth=128 * 29 * 10, weight_initializer='kaiming_uniform', activation_func='leaky_relu',
                 use_batch_norm=False):
        super(RadarNetV1Encoder, self).__init__()
        self.n_neuron_latent_depth = n_neurons_encoder_depth[-1]
        self.encoder_image = ResNetEncoder(18, input_channels_image, n_filters_encoder_image, weight_initializer, activation_func,
                                           use_batch_norm)
        self.encoder_depth = FullyConnectedEncoder(input_channels_depth, n_neurons_encoder_depth, latent_size_depth,
                                                   weight_initializer, activation_func)
        self.input_patch_size_image = input_patch_size_image
