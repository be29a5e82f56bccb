'''
Sample formats and FusionNet datasets against fixture T9 (outputs of the reference's own data_utils / datasets /
points_to_depth_map on the committed PNG files, tests/golden/make_golden_formats.py).

CPU: the numpy oracle and the host mirrors (rcf_amd.data_utils, rcf_amd.datasets: file I/O + RNG bookkeeping) reproduce the fixture
bit for bit.  GPU (-m gpu): the device kernels behind ops.decode_* / encode_maps / points_to_depth_map and the raw DataLoader path
(datasets.to_device_batch) do too, and agree with the oracle at 900 x 1600.  Integer / index work: bit-exact; the float steps are
single IEEE operations, so the bar is equality as well.
'''
import os
import warnings

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import formats_oracle
from rcf_amd import data_utils, datasets

CROPS = [(['none'], 1), (['center'], 2), (['left', 'top'], 3), (['right', 'bottom'], 4), (['horizontal'], 5),
         (['horizontal', 'vertical'], 6), (['horizontal', 'vertical'], 7), (['horizontal', 'vertical', 'anchored'], 8),
         (['horizontal', 'vertical', 'anchored'], 9), (['vertical', 'anchored', 'left'], 10), (['horizontal', 'bottom'], 11)]
KINDS = ('depth', 'response', 'ground_truth', 'lidar')


@pytest.fixture(scope='module')
def t9(golden_dir):
    g = np.load(os.path.join(golden_dir, 'T9_formats.npz'))
    d = os.path.join(golden_dir, 'T9_formats')
    paths = {k: [os.path.join(d, '%s_%d.png' % (k, i)) for i in range(3)] for k in ('image',) + KINDS}
    return g, paths


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b)


def _raw_batch(paths):
    img = np.stack([data_utils.load_image_raw(p) for p in paths['image']])
    maps = {k: np.stack([data_utils.load_map_raw(p) for p in paths[k]]) for k in KINDS}
    return img, maps


# ------------------------------------------------------------------------------------------------ CPU: oracle
def test_formats_oracle_reproduces_reference_fixture_t9(t9):
    g, paths = t9
    img, maps = _raw_batch(paths)
    assert img.dtype == np.uint8 and maps['depth'].dtype == np.uint16
    chw = formats_oracle.decode_images(img)
    chw_n = formats_oracle.decode_images(img, normalize=True)
    z, v = formats_oracle.decode_maps(maps['depth'])
    r, _ = formats_oracle.decode_maps(maps['response'], multiplier=2 ** 14, clamp_nonpositive=False)
    rd, _ = formats_oracle.decode_maps(maps['response'])
    for i in range(3):
        assert _same(np.transpose(chw[i], (1, 2, 0)), g['load_image_hwc_%d' % i])
        assert _same(chw_n[i], g['load_image_chw_norm_%d' % i])
        assert _same(z[i, 0], g['load_depth_%d' % i])
        assert _same(z[i], g['load_depth_v_z_%d' % i]) and _same(v[i], g['load_depth_v_v_%d' % i])
        assert _same(r[i, 0][..., None], g['load_response_%d' % i])
        assert _same(rd[i], g['response_as_depth_%d' % i])
    # crops: the window of the reference's crop is found by matching, then the oracle's crop must reproduce it
    shape = tuple(int(v_) for v_ in g['crop_shape'])
    for ci, (crop_type, seed) in enumerate(CROPS):
        np.random.seed(seed)
        y0, x0 = datasets.draw_crop(img.shape[1], img.shape[2], shape, crop_type)
        c = formats_oracle.decode_images(img[:1], [(y0, x0)], shape)
        d, _ = formats_oracle.decode_maps(maps['depth'][:1], 256.0, [(y0, x0)], shape)
        assert _same(c[0], g['crop_image_%d' % ci]) and _same(d[0], g['crop_depth_%d' % ci]), crop_type
    # writers: np.uint32(z * multiplier), then PIL clips to 16 bits when it saves
    for i in range(3):
        for kind, mult in (('depth', 256.0), ('response', 2 ** 14)):
            enc = formats_oracle.encode_maps(g['saved_%s_%d' % (kind, i)], mult)
            assert enc.dtype == np.uint32
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                pix = np.array(Image.fromarray(enc, mode='I').convert('I;16'))
            assert np.array_equal(pix, g['pixels_%s_%d' % (kind, i)])
    assert formats_oracle.encode_maps(np.float32([-1.0]))[0] == 4294967040      # the wrap-around the reference's cast has
    h, w = g['p2d_map'].shape
    assert np.array_equal(formats_oracle.points_to_depth_map(g['p2d_points'], g['p2d_depth'], h, w), g['p2d_map'])
    assert np.array_equal(formats_oracle.points_to_depth_map(g['p2d_points64'], g['p2d_depth'], h, w), g['p2d_map64'])


# ------------------------------------------------------------------------------------------------ CPU: host mirrors
def test_data_utils_mirror_reproduces_reference_fixture_t9(t9, tmp_path):
    g, paths = t9
    for i in range(3):
        assert _same(data_utils.load_image(paths['image'][i]), g['load_image_hwc_%d' % i])
        assert _same(data_utils.load_image(paths['image'][i], normalize=True, data_format='CHW'), g['load_image_chw_norm_%d' % i])
        assert _same(data_utils.load_depth(paths['depth'][i]), g['load_depth_%d' % i])
        z, v = data_utils.load_depth_with_validity_map(paths['depth'][i], data_format='CHW')
        assert _same(z, g['load_depth_v_z_%d' % i]) and _same(v, g['load_depth_v_v_%d' % i])
        assert _same(data_utils.load_response(paths['response'][i], data_format='HWC'), g['load_response_%d' % i])
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            data_utils.save_depth(g['saved_depth_%d' % i], str(tmp_path / 'd.png'))
            data_utils.save_response(g['saved_response_%d' % i], str(tmp_path / 'r.png'))
        assert np.array_equal(np.array(Image.open(str(tmp_path / 'd.png'))), g['pixels_depth_%d' % i])
        assert np.array_equal(np.array(Image.open(str(tmp_path / 'r.png'))), g['pixels_response_%d' % i])
    for bad in ('NCHW', 'hwc'):
        with pytest.raises(ValueError):
            data_utils.load_image(paths['image'][0], data_format=bad)
        with pytest.raises(ValueError):
            data_utils.load_depth(paths['depth'][0], data_format=bad)
    listing = str(tmp_path / 'paths.txt')
    data_utils.write_paths(listing, paths['image'])
    assert data_utils.read_paths(listing) == paths['image']
    with open(listing, 'w') as f:
        f.write('a.png\nb.png\n\nc.png\n')
    assert data_utils.read_paths(listing) == ['a.png', 'b.png']         # the reference stops at the first empty line


def test_random_crop_and_datasets_reproduce_reference_fixture_t9(t9):
    g, paths = t9
    shape = tuple(int(v_) for v_ in g['crop_shape'])
    image = data_utils.load_image(paths['image'][0], data_format='CHW')
    depth = data_utils.load_depth(paths['depth'][0], data_format='CHW')
    for ci, (crop_type, seed) in enumerate(CROPS):
        np.random.seed(seed)
        a, b = datasets.random_crop([image, depth], shape, crop_type)
        assert _same(a, g['crop_image_%d' % ci]) and _same(b, g['crop_depth_%d' % ci]), crop_type
        assert np.array_equal(np.random.rand(1), g['crop_next_rand_%d' % ci]), 'RNG stream consumed differently: %s' % crop_type
    with pytest.raises(ValueError):
        datasets.random_crop([image], image.shape[1:], ['horizontal'])     # numpy's randint(0, 0), as in the reference

    args = [paths[k] for k in ('image',) + KINDS]
    np.random.seed(21)
    ds = datasets.FusionNetTrainingDataset(*args, shape=shape, random_crop_type=['horizontal', 'vertical', 'anchored'])
    assert len(ds) == 3
    for rep in range(2):
        for i in range(3):
            sample = ds[i]
            assert len(sample) == 5
            for j, t in enumerate(sample):
                assert _same(t, g['train_ds_%d_%d_%d' % (rep, i, j)])
    for j, t in enumerate(datasets.FusionNetTrainingDataset(*args)[1]):
        assert _same(t, g['train_ds_full_1_%d' % j])
    for j, t in enumerate(datasets.FusionNetInferenceDataset(*args[:4])[2]):
        assert _same(t, g['infer_ds_2_%d' % j])
    nogt = datasets.FusionNetInferenceDataset(args[0], args[1], args[2], [None] * 3)
    assert len(nogt[0]) == int(g['infer_ds_nogt_len'][0]) == 3
    with pytest.raises(AssertionError):
        datasets.FusionNetTrainingDataset(args[0], args[1][:2], args[2], args[3], args[4])

    # raw samples: integers + the crop offset the reference would have used (same RNG stream); finishing them with the oracle
    # gives the reference's samples
    np.random.seed(21)
    raw = datasets.FusionNetTrainingDataset(*args, shape=shape, random_crop_type=['horizontal', 'vertical', 'anchored'], raw=True)
    for rep in range(2):
        for i in range(3):
            *tensors, crop = raw[i]
            assert tensors[0].dtype == np.uint8 and all(t.dtype == np.uint16 for t in tensors[1:]) and crop.dtype == np.int32
            assert _same(formats_oracle.decode_images(tensors[0][None], [crop], shape)[0], g['train_ds_%d_%d_0' % (rep, i)])
            for j in range(1, 5):
                assert _same(formats_oracle.decode_maps(tensors[j][None], 256.0, [crop], shape)[0][0], g['train_ds_%d_%d_%d' % (rep, i, j)])
    with pytest.raises(Exception):
        datasets.to_device_batch([torch.zeros(1, 4, 4, 3, dtype=torch.uint8), torch.zeros(1, 2, dtype=torch.int32)], 'cpu')


def test_radarnet_datasets_reproduce_reference_fixture_t9(t9, golden_dir):
    import random
    g, paths = t9
    radar = [os.path.join(golden_dir, 'T9_formats', 'radar_%d.npy' % i) for i in range(3)]
    for tag, prob in (('radar', 0.0), ('lidar', 1.0), ('mixed', 0.5)):
        np.random.seed(31)
        random.seed(32)
        ds = datasets.RadarNetTrainingDataset(paths['image'], radar, paths['ground_truth'], patch_size=(9, 6),
                                              total_points_sampled=4, sample_probability_of_lidar=prob)
        for rep in range(2):
            for i in range(len(ds)):
                sample = ds[i]
                assert len(sample) == 4
                for j, t in enumerate(sample):
                    assert _same(t, g['radarnet_train_%s_%d_%d_%d' % (tag, rep, i, j)]), (tag, rep, i, j)
        assert np.array_equal(np.array([np.random.rand(), random.random()]), g['radarnet_train_%s_next' % tag]), 'RNG streams: ' + tag
    ds = datasets.RadarNetInferenceDataset(paths['image'], radar, paths['ground_truth'])
    for i in range(3):
        for j, t in enumerate(ds[i]):
            assert _same(t, g['radarnet_infer_%d_%d' % (i, j)])
    assert len(datasets.RadarNetInferenceDataset(paths['image'], radar)[0]) == int(g['radarnet_infer_nogt_len'][0]) == 2
    with pytest.raises(AssertionError):
        datasets.RadarNetTrainingDataset(paths['image'], radar[:2], paths['ground_truth'], (9, 6), 4, 0.0)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_device_decode_encode_match_reference_fixture_t9(t9):
    from rcf_amd import ops
    g, paths = t9
    img, maps = _raw_batch(paths)
    dev = torch.device('cuda')
    d_img = torch.from_numpy(img).to(dev)
    d_maps = {k: torch.from_numpy(v).to(dev) for k, v in maps.items()}
    chw = ops.decode_images(d_img).cpu().numpy()
    chw_n = ops.decode_images(d_img, normalize=True).cpu().numpy()
    z, v = [t.cpu().numpy() for t in ops.decode_maps(d_maps['depth'], with_validity=True)]
    r = ops.decode_maps(d_maps['response'], multiplier=2 ** 14, clamp_nonpositive=False).cpu().numpy()
    rd = ops.decode_maps(d_maps['response']).cpu().numpy()
    for i in range(3):
        assert _same(np.transpose(chw[i], (1, 2, 0)), g['load_image_hwc_%d' % i])
        assert _same(chw_n[i], g['load_image_chw_norm_%d' % i])
        assert _same(z[i], g['load_depth_v_z_%d' % i]) and _same(v[i], g['load_depth_v_v_%d' % i])
        assert _same(r[i, 0][..., None], g['load_response_%d' % i])
        assert _same(rd[i], g['response_as_depth_%d' % i])
    # the other integer types decode alike
    for dt in (torch.int32, torch.uint8):
        src = d_maps['depth'].to(torch.int32).clamp(max=255).to(dt) if dt == torch.uint8 else d_maps['depth'].to(torch.int32)
        want, _ = formats_oracle.decode_maps(src.cpu().numpy())
        assert _same(ops.decode_maps(src).cpu().numpy(), want)
    neg = torch.tensor([[[-512, 0, 640]]], dtype=torch.int32, device=dev)
    assert ops.decode_maps(neg).flatten().tolist() == [0.0, 0.0, 2.5]
    assert ops.decode_maps(neg, clamp_nonpositive=False).flatten().tolist() == [-2.0, 0.0, 2.5]
    shape = tuple(int(v_) for v_ in g['crop_shape'])
    for ci, (crop_type, seed) in enumerate(CROPS):
        np.random.seed(seed)
        y0, x0 = datasets.draw_crop(img.shape[1], img.shape[2], shape, crop_type)
        assert _same(ops.decode_images(d_img[:1], [(y0, x0)], shape)[0].cpu().numpy(), g['crop_image_%d' % ci])
        assert _same(ops.decode_maps(d_maps['depth'][:1], 256.0, [(y0, x0)], shape)[0].cpu().numpy(), g['crop_depth_%d' % ci])
    with pytest.raises(Exception):
        ops.decode_images(d_img[:1], [(6, 0)], shape)          # window leaves the image
    with pytest.raises(Exception):
        ops.decode_images(d_img[:1], None, shape)              # a crop needs offsets
    with pytest.raises(Exception):
        ops.decode_maps(d_maps['depth'].to(torch.int64))
    for i in range(3):
        for kind, mult in (('depth', 256.0), ('response', 2 ** 14)):
            enc = ops.encode_maps(torch.from_numpy(g['saved_%s_%d' % (kind, i)]).to(dev), mult).cpu().numpy()
            assert np.array_equal(enc.view(np.uint32), formats_oracle.encode_maps(g['saved_%s_%d' % (kind, i)], mult))
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_raw_dataloader_batches_decode_on_device_like_the_reference_dataset(t9):
    g, paths = t9
    shape = tuple(int(v_) for v_ in g['crop_shape'])
    args = [paths[k] for k in ('image',) + KINDS]
    np.random.seed(21)
    raw = datasets.FusionNetTrainingDataset(*args, shape=shape, random_crop_type=['horizontal', 'vertical', 'anchored'], raw=True)
    loader = torch.utils.data.DataLoader(raw, batch_size=3, shuffle=False, num_workers=0)
    for rep in range(2):
        (batch,) = list(loader)
        out = datasets.to_device_batch(batch, 'cuda', shape)
        assert [tuple(t.shape) for t in out] == [(3, 3) + shape] + [(3, 1) + shape] * 4
        for i in range(3):
            for j in range(5):
                assert _same(out[j][i].cpu().numpy(), g['train_ds_%d_%d_%d' % (rep, i, j)])
    full = datasets.FusionNetInferenceDataset(*args[:4], raw=True)
    batch = next(iter(torch.utils.data.DataLoader(full, batch_size=1, shuffle=False, sampler=[2])))
    out = datasets.to_device_batch(batch, 'cuda')
    for j in range(4):
        assert _same(out[j][0].cpu().numpy(), g['infer_ds_2_%d' % j])


@pytest.mark.gpu
def test_points_to_depth_map_device_matches_reference_fixture_t9(t9):
    from rcf_amd import ops
    g, _ = t9
    h, w = g['p2d_map'].shape
    dev = torch.device('cuda')
    dep = torch.from_numpy(g['p2d_depth']).to(dev)
    got = ops.points_to_depth_map(torch.from_numpy(g['p2d_points']).to(dev), dep, h, w).cpu().numpy()
    assert np.array_equal(got.astype(np.float64), g['p2d_map'])
    got = ops.points_to_depth_map(torch.from_numpy(g['p2d_points64']).to(dev), dep, h, w).cpu().numpy()
    assert np.array_equal(got.astype(np.float64), g['p2d_map64'])
    assert float(ops.points_to_depth_map(torch.zeros(2, 0, device=dev), torch.zeros(0, device=dev), 4, 5).abs().sum()) == 0.0
    with pytest.raises(IndexError):
        ops.points_to_depth_map(torch.tensor([[1.0, float(w)], [1.0, 2.0]], device=dev), torch.ones(2, device=dev), h, w)
    # a dense lidar sweep at full resolution against the oracle: heavy collisions, last writer wins
    rs = np.random.RandomState(3)
    n, hh, ww = 200000, 900, 1600
    pts = np.stack([rs.rand(n) * (ww - 1), rs.rand(n) * 40 + 400]).astype(np.float32)
    d = (rs.rand(n) * 80 + 1).astype(np.float32)
    got = ops.points_to_depth_map(torch.from_numpy(pts).to(dev), torch.from_numpy(d).to(dev), hh, ww).cpu().numpy()
    assert np.array_equal(got.astype(np.float64), formats_oracle.points_to_depth_map(pts, d, hh, ww))


@pytest.mark.gpu
def test_device_decode_full_resolution_batch_against_oracle():
    from rcf_amd import ops
    rs = np.random.RandomState(4)
    n, hh, ww, shape = 4, 900, 1600, (768, 1408)
    img = rs.randint(0, 256, size=(n, hh, ww, 3)).astype(np.uint8)
    dep = (rs.randint(0, 65536, size=(n, hh, ww)) * (rs.rand(n, hh, ww) < 0.3)).astype(np.uint16)
    crop = np.stack([rs.randint(0, hh - shape[0] + 1, size=n), rs.randint(0, ww - shape[1] + 1, size=n)], 1).astype(np.int32)
    crop[0] = (hh - shape[0], ww - shape[1])
    dev = torch.device('cuda')
    got_i = ops.decode_images(torch.from_numpy(img).to(dev), crop, shape).cpu().numpy()
    got_z, got_v = [t.cpu().numpy() for t in ops.decode_maps(torch.from_numpy(dep).to(dev), 256.0, crop, shape, with_validity=True)]
    want_z, want_v = formats_oracle.decode_maps(dep, 256.0, crop, shape)
    assert _same(got_i, formats_oracle.decode_images(img, crop, shape))
    assert _same(got_z, want_z) and _same(got_v, want_v)
    # encode -> decode is the identity on what a 16-bit file can hold
    enc = ops.encode_maps(torch.from_numpy(want_z).to(dev))
    assert torch.equal(enc.cpu(), torch.from_numpy(np.stack([dep[b, crop[b, 0]:crop[b, 0] + shape[0], crop[b, 1]:crop[b, 1] + shape[1]]
                                                             for b in range(n)])[:, None].astype(np.int32)))
