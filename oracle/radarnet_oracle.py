'''
ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

CPU restatement (stock PyTorch fp32 ops, NCHW, autograd) of the reference's RadarNet stage 1:
RadarNetModel.forward / compute_loss (src/radarnet_model.py:102-171), RadarNetV1Encoder (src/networks.py:1151-1256),
ResNetEncoder (src/networks.py:8-268), FullyConnectedEncoder (src/networks.py:1007-1067), FullyConnected
(src/net_utils.py:201-247); decoder and conv blocks come from oracle/fusionnet_oracle.py.

Parity status: PINNED against the real reference by tests/golden/make_golden_radarnet.py (identical seeded weights, outputs /
loss / gradients asserted equal) -- EXCEPT torchvision.ops.roi_pool, which is absent here and is restated in
oracle/roi_pool_oracle.py (parity unpinned at that boundary; the reference run uses the same restatement).
State-dict key names are identical to the reference's.
'''
import torch
import torch.nn.functional as F

from .fusionnet_oracle import LEAKY_SLOPE, Conv2d, MultiScaleDecoder, ResNetBlock
from .roi_pool_oracle import roi_pool


class FullyConnected(torch.nn.Module):
    '''src/net_utils.py:201-247 (no dropout on the shipped path).'''

    def __init__(self, in_features, out_features):
        super().__init__()
        self.fully_connected = torch.nn.Linear(in_features, out_features)

    def forward(self, x):
        return F.leaky_relu(self.fully_connected(x), LEAKY_SLOPE)


class ResNetEncoder(torch.nn.Module):
    '''src/networks.py:8-268, n_layer 18.'''

    def __init__(self, input_channels, n_filters, use_batch_norm):
        super().__init__()
        n_blocks = [2, 2, 2, 2]
        for _ in range(len(n_filters) - len(n_blocks) - 1):
            n_blocks = n_blocks + [n_blocks[-1]]
        self.depth = len(n_filters)
        self.conv1 = Conv2d(input_channels, n_filters[0], 7, 2, 'leaky_relu', use_batch_norm)
        self.max_pool = torch.nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        for lvl in range(2, self.depth + 1):
            stride = 1 if lvl == 2 else 2
            blocks = [ResNetBlock(n_filters[lvl - 2] if b == 0 else n_filters[lvl - 1], n_filters[lvl - 1], stride if b == 0 else 1,
                                  use_batch_norm) for b in range(n_blocks[lvl - 2])]
            setattr(self, 'blocks%d' % lvl, torch.nn.Sequential(*blocks))

    def forward(self, x):
        layers = [self.conv1(x)]
        x = self.max_pool(layers[-1])
        for lvl in range(2, self.depth + 1):
            x = getattr(self, 'blocks%d' % lvl)(x)
            layers.append(x)
        return layers[-1], layers[:-1]


class FullyConnectedEncoder(torch.nn.Module):
    def __init__(self, input_channels, n_neurons, latent_size):
        super().__init__()
        sizes = [input_channels] + list(n_neurons[:5]) + [latent_size]
        self.mlp = torch.nn.Sequential(*[FullyConnected(sizes[i], sizes[i + 1]) for i in range(6)])

    def forward(self, x):
        return self.mlp(x)


class RadarNetV1Encoder(torch.nn.Module):
    '''src/networks.py:1151-1256.'''

    def __init__(self, input_channels_image, input_channels_depth, patch, n_filters, n_neurons, latent_size, use_batch_norm):
        super().__init__()
        self.n_neuron_latent_depth = n_neurons[-1]
        self.encoder_image = ResNetEncoder(input_channels_image, n_filters, use_batch_norm)
        self.encoder_depth = FullyConnectedEncoder(input_channels_depth, n_neurons, latent_size)
        self.patch = patch

    def forward(self, image, points, b_boxes):
        shape = self.patch
        lat_h, lat_w = int(shape[-2] // 32.0), int(shape[-1] // 32.0)
        scales = [1 / 2.0, 1 / 4.0, 1 / 8.0, 1 / 16.0, 1 / 32.0, 1 / 64.0, 1 / 128.0]
        latent_image, skips_image = self.encoder_image(image)
        latent_pooled = roi_pool(latent_image, b_boxes, (lat_h, lat_w), 1 / 32.0)
        skips = [roi_pool(s, b_boxes, (int(shape[-2] * scales[i]), int(shape[-1] * scales[i])), scales[i])
                 for i, s in enumerate(skips_image)]
        latent_depth = self.encoder_depth(points).view(points.shape[0], self.n_neuron_latent_depth, -1, lat_w)
        return torch.cat([latent_pooled, latent_depth], dim=1), skips


class RadarNetOracle(object):
    def __init__(self, input_channels_image, input_channels_depth, input_patch_size_image, encoder_type, n_filters_encoder_image,
                 n_neurons_encoder_depth, decoder_type, n_filters_decoder, weight_initializer='kaiming_uniform',
                 activation_func='leaky_relu'):
        h, w = input_patch_size_image
        latent_size = int(h // 32.0) * int(w // 32.0) * n_neurons_encoder_depth[-1]
        self.patch = tuple(input_patch_size_image)
        self.encoder = RadarNetV1Encoder(input_channels_image, input_channels_depth, self.patch, n_filters_encoder_image,
                                         n_neurons_encoder_depth, latent_size, 'batch_norm' in encoder_type)
        n_skips = list(n_filters_encoder_image[:-1])[::-1] + [0]
        self.decoder = MultiScaleDecoder(n_filters_encoder_image[-1] + n_neurons_encoder_depth[-1], n_filters_decoder, n_skips,
                                         'batch_norm' in decoder_type)

    def forward(self, image, point, bounding_boxes, return_logits=True):
        latent, skips = self.encoder(image, point, bounding_boxes)
        logits = self.decoder(latent, skips, self.patch)[-1]
        return logits if return_logits else torch.sigmoid(logits)

    def compute_loss(self, logits, ground_truth, validity_map, w_positive_class=1.0):
        loss = F.binary_cross_entropy_with_logits(logits, ground_truth, reduction='none', pos_weight=torch.tensor(w_positive_class))
        return torch.sum(validity_map * loss) / torch.sum(validity_map)

    def parameters(self):
        return list(self.encoder.parameters()) + list(self.decoder.parameters())

    def train(self):
        self.encoder.train(); self.decoder.train()

    def eval(self):
        self.encoder.eval(); self.decoder.eval()
