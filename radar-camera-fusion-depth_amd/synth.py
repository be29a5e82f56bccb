'''
Seeded synthetic inputs and portable seeded weights.

There is no network for datasets or checkpoints, so the benchmark and every parity test run on
synthetic data of the reference's shapes.  Everything here is numpy-only and
seeded, so the HIP path, the CPU oracle and the real reference (tests/golden/make_golden.py)
all see bit-identical inputs and weights.
'''

import numpy as np
import torch


def make_batch(n_batch, n_height, n_width, n_point=64, seed=1234):
    '''
    Synthetic FusionNet training batch with the statistics of the reference's data pipeline.

    image        : U[0,1) N x 3 x H x W   (normalized_image_range 0 1, bash/train_fusionnet_nuscenes.sh:29)
    input_depth  : N x 2 x H x W = cat([depth, response]) (src/fusionnet_main.py:366).  Each radar
                   point fills a column band (<= 288 px at 1600 wide, thinned to ~25 % density) with its
                   z; where bands overlap the larger response wins (src/radarnet_main.py:563-589).
                   The response channel is in [32, 64]: stage 1 saves response x 2^14
                   (src/data_utils.py:320-335) but the FusionNet dataset reads it back / 256
                   (src/datasets.py:413-415).
    ground_truth : 30 % dense U[1,80) m; lidar_map : 1 % dense U[1,80) m.
    radar_points : N x n_point x 3 (x px, y px, z m).
    '''
    rs = np.random.RandomState(seed)
    h, w = n_height, n_width
    image = rs.rand(n_batch, 3, h, w).astype(np.float32)
    points = np.stack([
        rs.uniform(0, w, size=(n_batch, n_point)),
        rs.uniform(h / 3.0, 7.0 * h / 9.0, size=(n_batch, n_point)),
        rs.uniform(1.0, 80.0, size=(n_batch, n_point))], axis=-1).astype(np.float32)
    band = max(2, min(288, int(round(288.0 * w / 1600.0))))
    depth = np.zeros((n_batch, 1, h, w), np.float32)
    response = np.zeros((n_batch, 1, h, w), np.float32)
    for b in range(n_batch):
        for k in range(n_point):
            x = int(points[b, k, 0])
            x0, x1 = max(0, x - band // 2), min(w, x + band // 2)
            if x1 <= x0:
                continue
            keep = rs.rand(h, x1 - x0) < 0.25
            resp = (32.0 + 32.0 * rs.rand(h, x1 - x0)).astype(np.float32) * keep
            win = resp > response[b, 0, :, x0:x1]
            response[b, 0, :, x0:x1] = np.where(win, resp, response[b, 0, :, x0:x1])
            depth[b, 0, :, x0:x1] = np.where(win, points[b, k, 2], depth[b, 0, :, x0:x1])
    gt = (rs.uniform(1.0, 80.0, size=(n_batch, 1, h, w)) * (rs.rand(n_batch, 1, h, w) < 0.30)).astype(np.float32)
    lidar = (rs.uniform(1.0, 80.0, size=(n_batch, 1, h, w)) * (rs.rand(n_batch, 1, h, w) < 0.01)).astype(np.float32)
    return {
        'image': torch.from_numpy(image),
        'input_depth': torch.from_numpy(np.concatenate([depth, response], axis=1)),
        'ground_truth': torch.from_numpy(gt),
        'lidar_map': torch.from_numpy(lidar),
        'radar_points': torch.from_numpy(points),
    }


def fill_state_dict_(modules, seed):
    '''
    Overwrite every tensor of the given torch modules' state dicts, in sorted key order, from
    numpy.random.RandomState(seed).  Keys follow the reference naming (e.g.
    'blocks2_image.0.conv1.conv.weight', '...batch_norm.running_mean'):
      conv / fully_connected weight   U(-b, b), b = 1/sqrt(fan_in);   fully_connected bias  U(-0.2, 0.2)
      batch_norm.weight    U(0.5, 1.5)     batch_norm.bias  U(-0.1, 0.1)
      running_mean         U(-0.1, 0.1)    running_var      U(0.5, 1.5)
      num_batches_tracked  0
    '''
    rs = np.random.RandomState(seed)
    for module in modules:
        sd = module.state_dict()
        for key in sorted(sd.keys()):
            t = sd[key]
            k = key[len('module.'):] if key.startswith('module.') else key
            if k.endswith('num_batches_tracked'):
                t.zero_()
                continue
            shape = tuple(t.shape)
            if k.endswith('conv.weight') or k.endswith('fully_connected.weight') or k.endswith('deconv.deconv.weight'):
                fan_in = int(np.prod(shape[1:]))
                b = 1.0 / np.sqrt(fan_in)
                v = rs.uniform(-b, b, size=shape)
            elif k.endswith('fully_connected.bias'):
                v = rs.uniform(-0.2, 0.2, size=shape)
            elif k.endswith('batch_norm.weight') or k.endswith('running_var'):
                v = rs.uniform(0.5, 1.5, size=shape)
            elif k.endswith('batch_norm.bias') or k.endswith('running_mean'):
                v = rs.uniform(-0.1, 0.1, size=shape)
            else:
                raise KeyError('fill_state_dict_: unexpected key ' + key)
            with torch.no_grad():
                t.copy_(torch.from_numpy(v.astype(np.float32)))


def make_scatter_case(k, h, w, wc, seed, small_z):
    '''
    Seeded inputs for the radar point -> grid scatter (src/radarnet_main.py:563-589): K sparse sigmoid response crops
    (H x wc) and K points (x in padded-canvas coordinates, y, z).  small_z draws z from [0.2, K+3) so that int(z) collides
    with point indices and the reference's in-place replacement chain changes the answer.
    '''
    rs = np.random.RandomState(seed)
    pad = wc // 2
    crops = rs.rand(k, h, wc).astype(np.float32)
    crops *= (rs.rand(k, h, wc) < 0.35)
    pts = np.stack([rs.uniform(pad, w + pad, k), rs.uniform(0, h, k),
                    rs.uniform(0.2, (k + 3) if small_z else 80.0, k)], -1).astype(np.float32)
    return crops, pts


PUBLISHED = dict(
    input_channels_image=3, input_channels_depth=2,
    n_filters_encoder_image=[32, 64, 128, 256, 256, 256],
    n_filters_encoder_depth=[16, 32, 64, 128, 128, 128],
    n_filters_decoder=[256, 256, 128, 64, 64, 32])
'''bash/train_fusionnet_nuscenes.sh:27-40'''

TINY = dict(
    input_channels_image=3, input_channels_depth=2,
    n_filters_encoder_image=[8, 16, 32, 32, 32, 32],
    n_filters_encoder_depth=[4, 8, 16, 16, 16, 16],
    n_filters_decoder=[32, 32, 16, 8, 8, 4])
'''SURVEY.md 8c fixture T0: same topology, small channels, used at odd spatial sizes.'''


def make_radarnet_batch(seed, n=2, k=2, h=64, w=96, patch_w=32):
    '''
    Seeded RadarNet stage-1 inputs: image (n,3,h,w) U[0,1); k radar points per image (x in image coordinates at least half a
    patch from the border, y, z) as an (n*k, 3) tensor; one box per point [x - patch_w/2, 0, x + patch_w/2, h] as
    src/radarnet_main.py:638-648 builds them; per-point ground-truth and validity maps (n*k, 1, h, patch_w).
    '''
    rs = np.random.RandomState(seed)
    image = rs.rand(n, 3, h, w).astype(np.float32)
    pad = patch_w // 2
    pts, boxes = [], []
    for i in range(n):
        x = rs.uniform(pad, w - pad, size=k)
        pts.append(np.stack([x, rs.uniform(0, h, size=k), rs.uniform(1.0, 80.0, size=k)], -1).astype(np.float32))
        boxes.append(np.stack([x - pad, np.zeros(k), x + pad, np.full(k, h)], -1).astype(np.float32))
    m = n * k
    gt = (rs.rand(m, 1, h, patch_w) < 0.2).astype(np.float32)
    valid = (rs.rand(m, 1, h, patch_w) < 0.8).astype(np.float32)
    return {
        'image': torch.from_numpy(image),
        'point': torch.from_numpy(np.concatenate(pts, 0)),
        'bounding_boxes': [torch.from_numpy(b) for b in boxes],
        'ground_truth': torch.from_numpy(gt),
        'validity_map': torch.from_numpy(valid),
    }


RADARNET_TINY = dict(
    input_channels_image=3, input_channels_depth=3, input_patch_size_image=(64, 32),
    encoder_type=['radarnetv1', 'batch_norm'], n_filters_encoder_image=[8, 16, 32, 32, 32],
    n_neurons_encoder_depth=[8, 16, 32, 32, 32], decoder_type=['multiscale', 'batch_norm'],
    n_filters_decoder=[32, 16, 8, 8, 4], weight_initializer='kaiming_uniform', activation_func='leaky_relu')
'''fixture T5: the shipped RadarNet topology (bash/train_radarnet_nuscenes.sh:22-31) with small channel counts.'''

RADARNET_PUBLISHED = dict(
    input_channels_image=3, input_channels_depth=3, input_patch_size_image=(900, 288),
    encoder_type=['radarnetv1', 'batch_norm'], n_filters_encoder_image=[32, 64, 128, 128, 128],
    n_neurons_encoder_depth=[32, 64, 128, 128, 128], decoder_type=['multiscale', 'batch_norm'],
    n_filters_decoder=[256, 128, 64, 32, 16], weight_initializer='kaiming_uniform', activation_func='leaky_relu')
'''bash/train_radarnet_nuscenes.sh:19-31'''
