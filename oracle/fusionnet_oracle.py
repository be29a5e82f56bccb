'''
ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

CPU restatement (stock PyTorch fp32 ops, NCHW, autograd) of the reference's FusionNet
hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this file, and only as the checker / the timed CPU baseline.

Parity status: PINNED.  tests/golden/make_golden.py imports the real reference from
/root/reference (with the three import shims of SURVEY.md 8c), loads identical seeded
weights into both, and asserts this restatement reproduces the reference's outputs,
loss, parameter gradients and BN running statistics; the resulting vectors are the
committed fixtures under tests/golden/ which tests/test_oracle_golden.py re-checks
without the reference present.

Every class cites the reference file:line it restates (paths relative to /root/reference).
State-dict key names are identical to the reference's so the same weights load on both.
'''

import torch
import torch.nn.functional as F


LEAKY_SLOPE = 0.20   # src/net_utils.py:15
BN_EPS = 1e-5        # torch.nn.BatchNorm2d default, src/net_utils.py:82
BN_MOMENTUM = 0.1


class Conv2d(torch.nn.Module):
    '''src/net_utils.py:29-91 -- conv(bias=False, padding=k//2) -> optional BN -> optional act.
    act in {'leaky_relu', 'sigmoid', 'linear'} (src/net_utils.py:4-23).'''

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1,
                 act='leaky_relu', use_batch_norm=False):
        super().__init__()
        self.conv = torch.nn.Conv2d(
            in_channels, out_channels, kernel_size=kernel_size, stride=stride,
            padding=kernel_size // 2, bias=False)
        self.use_batch_norm = use_batch_norm
        self.act = act
        if use_batch_norm:
            self.batch_norm = torch.nn.BatchNorm2d(out_channels, eps=BN_EPS, momentum=BN_MOMENTUM)

    def forward(self, x):
        y = self.conv(x)
        if self.use_batch_norm:
            y = self.batch_norm(y)
        if self.act == 'leaky_relu':
            return F.leaky_relu(y, LEAKY_SLOPE)
        if self.act == 'sigmoid':
            return torch.sigmoid(y)
        return y


class UpConv2d(torch.nn.Module):
    '''src/net_utils.py:156-198 -- F.interpolate(x, size=shape) (mode nearest) -> Conv2d 3x3.'''

    def __init__(self, in_channels, out_channels, use_batch_norm):
        super().__init__()
        self.conv = Conv2d(in_channels, out_channels, 3, 1, 'leaky_relu', use_batch_norm)

    def forward(self, x, shape):
        return self.conv(F.interpolate(x, size=tuple(shape)))


class ResNetBlock(torch.nn.Module):
    '''src/net_utils.py:253-323.  Note act is applied to conv2 before AND after the residual
    add (:291-298, :323); the 1x1 projection (no BN, no act) is always allocated (:300-307)
    but only used when the shape changes (:317-320).'''

    def __init__(self, in_channels, out_channels, stride, use_batch_norm):
        super().__init__()
        self.conv1 = Conv2d(in_channels, out_channels, 3, stride, 'leaky_relu', use_batch_norm)
        self.conv2 = Conv2d(out_channels, out_channels, 3, 1, 'leaky_relu', use_batch_norm)
        self.projection = Conv2d(in_channels, out_channels, 1, stride, 'linear', False)

    def forward(self, x):
        conv2 = self.conv2(self.conv1(x))
        if list(x.shape[1:4]) != list(conv2.shape[1:4]):
            X = self.projection(x)
        else:
            X = x
        return F.leaky_relu(conv2 + X, LEAKY_SLOPE)


class TransposeConv2d(torch.nn.Module):
    '''src/net_utils.py:94-153 -- ConvTranspose2d(3, stride 2, padding 1, output_padding 1, bias=False) -> optional BN -> leaky_relu.'''

    def __init__(self, in_channels, out_channels, use_batch_norm):
        super().__init__()
        self.deconv = torch.nn.ConvTranspose2d(in_channels, out_channels, kernel_size=3, stride=2, padding=1, output_padding=1,
                                               bias=False)
        self.use_batch_norm = use_batch_norm
        if use_batch_norm:
            self.batch_norm = torch.nn.BatchNorm2d(out_channels, eps=BN_EPS, momentum=BN_MOMENTUM)

    def forward(self, x):
        y = self.deconv(x)
        if self.use_batch_norm:
            y = self.batch_norm(y)
        return F.leaky_relu(y, LEAKY_SLOPE)


class DecoderBlock(torch.nn.Module):
    '''src/net_utils.py:473-569; deconv_type 'up' (hard-coded by src/fusionnet_main.py:190) or 'transpose' (:507-513, :550-551).'''

    def __init__(self, in_channels, skip_channels, out_channels, use_batch_norm, deconv_type='up'):
        super().__init__()
        self.skip_channels = skip_channels
        self.deconv_type = deconv_type
        if deconv_type == 'transpose':
            self.deconv = TransposeConv2d(in_channels, out_channels, use_batch_norm)
        else:
            self.deconv = UpConv2d(in_channels, out_channels, use_batch_norm)
        self.conv = Conv2d(skip_channels + out_channels, out_channels, 3, 1, 'leaky_relu', use_batch_norm)

    def forward(self, x, skip=None, shape=None):
        if self.deconv_type == 'transpose':
            deconv = self.deconv(x)
            concat = torch.cat([deconv, skip], dim=1) if self.skip_channels > 0 else deconv
            return self.conv(concat)
        if skip is not None:
            shape = skip.shape[2:4]
        elif shape is None:
            shape = (2 * x.shape[2], 2 * x.shape[3])
        deconv = self.deconv(x, shape=shape)
        concat = torch.cat([deconv, skip], dim=1) if self.skip_channels > 0 else deconv
        return self.conv(concat)


class FusionNetEncoder(torch.nn.Module):
    '''src/networks.py:270-1005, fusion_type='weight_and_project' (the shipped flag,
    bash/train_fusionnet_nuscenes.sh:33); n_layer 18 (shipped: two ResNetBlocks per level) or 34 (3, 4, 6, 3, 3).'''

    def __init__(self, input_channels_image, input_channels_depth,
                 n_filters_encoder_image, n_filters_encoder_depth, use_batch_norm, n_layer=18):
        super().__init__()
        fi, fd = list(n_filters_encoder_image), list(n_filters_encoder_depth)
        assert len(fi) == len(fd) and 5 <= len(fi) < 8
        self.n_level = len(fi)
        n_blocks = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3]}[n_layer]                          # :305-311
        n_blocks = n_blocks + [n_blocks[-1]] * (len(fi) - len(n_blocks) - 1)                # :317-318
        bn = use_batch_norm
        self.conv1_image = Conv2d(input_channels_image, fi[0], 7, 2, 'leaky_relu', bn)   # :332
        self.conv1_depth = Conv2d(input_channels_depth, fd[0], 7, 2, 'leaky_relu', bn)   # :341
        self.conv1_weight = Conv2d(fd[0], fi[0], 1, 1, 'sigmoid', bn)                    # :373
        self.conv1_project = Conv2d(fd[0], fi[0], 1, 1, 'linear', bn)                    # :382
        self.max_pool = torch.nn.MaxPool2d(kernel_size=3, stride=2, padding=1)           # :392
        for lvl in range(2, self.n_level + 1):
            stride = 1 if lvl == 2 else 2                                                 # :414, :479
            ci, co = fi[lvl - 2], fi[lvl - 1]
            di, do = fd[lvl - 2], fd[lvl - 1]
            nb = n_blocks[lvl - 2]                                                        # _make_layer :767-838
            setattr(self, 'blocks%d_image' % lvl, torch.nn.Sequential(
                *[ResNetBlock(ci if b == 0 else co, co, stride if b == 0 else 1, bn) for b in range(nb)]))
            setattr(self, 'blocks%d_depth' % lvl, torch.nn.Sequential(
                *[ResNetBlock(di if b == 0 else do, do, stride if b == 0 else 1, bn) for b in range(nb)]))
            setattr(self, 'conv%d_weight' % lvl, Conv2d(do, co, 1, 1, 'sigmoid', bn))
            setattr(self, 'conv%d_project' % lvl, Conv2d(do, co, 1, 1, 'linear', bn))

    def forward(self, image, depth):
        layers = []
        img = self.conv1_image(image)
        dep = self.conv1_depth(depth)
        layers.append(self.conv1_weight(dep) * self.conv1_project(dep) + img)             # :863-866
        img = self.max_pool(img)                                                         # :875-876
        dep = self.max_pool(dep)
        for lvl in range(2, self.n_level + 1):
            img = getattr(self, 'blocks%d_image' % lvl)(img)
            dep = getattr(self, 'blocks%d_depth' % lvl)(dep)
            w = getattr(self, 'conv%d_weight' % lvl)(dep)
            p = getattr(self, 'conv%d_project' % lvl)(dep)
            layers.append(w * p + img)
        return layers[-1], layers[:-1]                                                   # :1005


class MultiScaleDecoder(torch.nn.Module):
    '''src/networks.py:1337-1657 on the n_resolution=1, output_func='linear' path.'''

    def __init__(self, input_channels, n_filters, n_skips, use_batch_norm, deconv_type='up'):
        super().__init__()
        depth = len(n_filters)
        assert 5 <= depth < 8 and len(n_skips) == depth
        self.names = ['deconv%d' % i for i in range(depth - 1, -1, -1)]   # deconv5..deconv0 for 6
        cin = input_channels
        for name, skip_c, out_c in zip(self.names, n_skips, n_filters):
            setattr(self, name, DecoderBlock(cin, skip_c, out_c, use_batch_norm, deconv_type))
            cin = out_c
        self.output0 = Conv2d(cin, 1, 3, 1, 'linear', False)                # :1548-1555

    def forward(self, x, skips, shape):
        n = len(skips) - 1
        for name in self.names[:-1]:
            x = getattr(self, name)(x, skips[n])
            n -= 1
        if n == 0:                                                          # :1649-1652
            x = self.deconv0(x, skips[0])
        else:
            x = self.deconv0(x, shape=tuple(shape[-2:]))
        return [self.output0(x)]


class FusionNetOracle(object):
    '''src/fusionnet_model.py:7-401 (FusionNetModel) for encoder_type=['fusionnet18', ...],
    decoder_type=['multiscale', ...], fusion weight_and_project, deconv 'up'.'''

    def __init__(self, input_channels_image=3, input_channels_depth=2,
                 n_filters_encoder_image=(32, 64, 128, 256, 256, 256),
                 n_filters_encoder_depth=(16, 32, 64, 128, 128, 128),
                 n_filters_decoder=(256, 256, 128, 64, 64, 32),
                 encoder_batch_norm=True, decoder_batch_norm=True,
                 min_predict_depth=1.0, max_predict_depth=100.0, deconv_type='up', n_layer=18):
        self.min_predict_depth = min_predict_depth
        self.max_predict_depth = max_predict_depth
        fi = list(n_filters_encoder_image)
        self.encoder = FusionNetEncoder(input_channels_image, input_channels_depth,
                                        fi, list(n_filters_encoder_depth), encoder_batch_norm, n_layer)
        n_skips = fi[:-1][::-1] + [0]                                        # :118-119
        self.decoder = MultiScaleDecoder(fi[-1], list(n_filters_decoder), n_skips, decoder_batch_norm, deconv_type)

    def forward(self, image, input_depth):
        latent, skips = self.encoder(image, input_depth)
        out = self.decoder(latent, skips, image.shape[-2:])[-1]
        return self.min_predict_depth / (
            torch.sigmoid(out) + self.min_predict_depth / self.max_predict_depth)   # :162-165

    def compute_loss(self, output_depth, ground_truth, lidar_map, w_lidar_loss=2.0, loss_func='l1', image=None, w_smoothness=0.0):
        '''src/fusionnet_model.py:209-302, w_smoothness=0 (shipped flag); loss_func 'l1' (shipped) / 'l2' / 'smoothl1'
        (:245-275 -> src/fusionnet_losses.py:4-46: F.l1_loss / F.mse_loss / F.smooth_l1_loss, reduction 'mean').'''
        fn = {'l1': F.l1_loss, 'l2': F.mse_loss, 'smoothl1': F.smooth_l1_loss}[loss_func]
        if w_lidar_loss > 0.0:
            ground_truth = ground_truth * torch.where(
                lidar_map > 0.0, torch.zeros_like(lidar_map), torch.ones_like(lidar_map))
        vg = ground_truth > 0
        vl = lidar_map > 0
        loss_sup = fn(output_depth[vg], ground_truth[vg], reduction='mean')
        loss_lidar = 0.0
        if w_lidar_loss > 0.0:
            loss_lidar = fn(output_depth[vl], lidar_map[vl], reduction='mean')
        if w_smoothness > 0.0:
            # losses.smoothness_loss_func (src/fusionnet_losses.py:48-72), the loss_smoothness_kernel_size <= 1 branch
            dy = lambda t: t[:, :, :-1, :] - t[:, :, 1:, :]
            dx = lambda t: t[:, :, :, :-1] - t[:, :, :, 1:]
            wx = torch.exp(-torch.mean(torch.abs(dx(image)), dim=1, keepdim=True))
            wy = torch.exp(-torch.mean(torch.abs(dy(image)), dim=1, keepdim=True))
            loss_smooth = torch.mean(wx * torch.abs(dx(output_depth))) + torch.mean(wy * torch.abs(dy(output_depth)))
            return loss_sup + w_smoothness * loss_smooth + w_lidar_loss * loss_lidar, loss_sup, loss_lidar, loss_smooth
        return loss_sup + w_lidar_loss * loss_lidar, loss_sup, loss_lidar

    def parameters(self):
        return list(self.encoder.parameters()) + list(self.decoder.parameters())

    def train(self):
        self.encoder.train(); self.decoder.train()

    def eval(self):
        self.encoder.eval(); self.decoder.eval()


def remove_outliers(depth, kernel_size=7, threshold=1.5):
    '''src/net_utils.py:591-638 (OutlierRemoval.remove_outliers).'''
    validity = torch.where(depth > 0.0, torch.ones_like(depth), depth)
    max_value = 10 * torch.max(depth)
    filled = torch.where(validity <= 0, torch.full_like(depth, max_value.item()), depth)
    p = kernel_size // 2
    filled = F.pad(filled, (p, p, p, p), mode='constant', value=max_value.item())
    min_values = -F.max_pool2d(-filled, kernel_size=kernel_size, stride=1, padding=0)
    clean = torch.where(min_values < depth - threshold,
                        torch.zeros_like(validity), torch.ones_like(validity))
    return depth * clean
