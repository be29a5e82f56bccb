'''
Datasets -- mirror of src/datasets.py:19-109 (random_crop), :112-343 (RadarNetTrainingDataset, RadarNetInferenceDataset: host
numpy, as in the reference) and :346-527 (FusionNetTrainingDataset, FusionNetInferenceDataset): same constructor arguments, same sample tuples (float32 C x H x W numpy arrays), same use of the global
numpy RNG for the crop, so torch.utils.data.DataLoader call sites of the reference (src/fusionnet_main.py:112-123, :145-153,
:669-677) work unchanged.

MI355X path (`raw=True`): a sample is what the PNG decoder produced -- uint8 H x W x 3 image, uint16 H x W maps -- plus the crop
offset the reference would have applied; DataLoader stacks those integers, and `to_device_batch` (which replaces
`[in_.to(device) for in_ in batch_data]`, src/fusionnet_main.py:353-355) uploads 3 + 4 x 2 bytes per pixel instead of 12 + 4 x 4
and finishes crop + HWC->CHW + float conversion + /256 + zeroing on the GPU, one launch per tensor for the whole batch.
'''
import random

import numpy as np
import torch

from . import data_utils, ops

_ANCHORS = (0.0, 0.50, 1.0)


def draw_crop(o_height, o_width, shape, crop_type=['none']):
    '''
    Top-left corner (y_start, x_start) of the crop random_crop takes (src/datasets.py:19-100): centre by default; left / right /
    horizontal(anchored) for x; top / bottom / vertical(anchored, only 30 % of the time) for y.  Consumes the global numpy RNG
    exactly like the reference (same calls, same order).
    '''
    n_height, n_width = shape
    d_height, d_width = o_height - n_height, o_width - n_width
    y_start, x_start = d_height // 2, d_width // 2

    if 'left' in crop_type:
        x_start = 0
    elif 'right' in crop_type:
        x_start = d_width
    elif 'horizontal' in crop_type:
        if 'anchored' in crop_type:
            x_start = int(_ANCHORS[np.random.randint(low=0, high=len(_ANCHORS))] * d_width)
        else:
            x_start = np.random.randint(low=0, high=d_width)

    if 'top' in crop_type:
        y_start = 0
    elif 'bottom' in crop_type:
        y_start = d_height
    elif 'vertical' in crop_type and np.random.rand() <= 0.30:
        if 'anchored' in crop_type:
            y_start = int(_ANCHORS[np.random.randint(low=0, high=len(_ANCHORS))] * d_height)
        else:
            y_start = np.random.randint(low=0, high=d_height)
    return int(y_start), int(x_start)


def random_crop(inputs, shape, crop_type=['none']):
    '''
    Apply crop to inputs e.g. images, depth (src/datasets.py:19-109)

    Arg(s):
        inputs : list[numpy[float32]]
            list of C x H x W arrays
        shape : list[int]
            (height, width) of the crop
        crop_type : list[str]
            none, horizontal, vertical, anchored, top, bottom, left, right, center
    Return:
        list[numpy[float32]] : list of cropped inputs
    '''
    _, o_height, o_width = inputs[0].shape
    y0, x0 = draw_crop(o_height, o_width, shape, crop_type)
    return [T[:, y0:y0 + shape[0], x0:x0 + shape[1]] for T in inputs]


class _Samples(torch.utils.data.Dataset):
    '''image + N range maps per index; float32 CHW numpy (reference contract) or raw integers + crop offset (device path)'''

    def __init__(self, image_paths, map_path_lists, shape, random_crop_type, raw):
        self.n_sample = len(image_paths)
        for paths in map_path_lists:
            assert len(paths) == self.n_sample
        self.image_paths = image_paths
        self._map_paths = map_path_lists
        self.shape = shape
        self.do_random_crop = self.shape is not None and all([x > 0 for x in self.shape])
        self.random_crop_type = random_crop_type
        self.data_format = 'CHW'
        self.raw = raw

    def __len__(self):
        return self.n_sample

    def _fetch(self, index):
        if self.raw:
            image = data_utils.load_image_raw(self.image_paths[index])
            maps = [data_utils.load_map_raw(paths[index]) for paths in self._map_paths]
            y0, x0 = 0, 0
            if self.do_random_crop:
                y0, x0 = draw_crop(image.shape[0], image.shape[1], self.shape, self.random_crop_type)
            return [image] + maps + [np.array([y0, x0], np.int32)]
        image = data_utils.load_image(self.image_paths[index], normalize=False, data_format=self.data_format)
        # every map, the response included, goes through the DEPTH decoder (/256, <= 0 zeroed): src/datasets.py:409-425
        maps = [data_utils.load_depth(paths[index], data_format=self.data_format) for paths in self._map_paths]
        sample = [image] + maps
        if self.do_random_crop:
            sample = random_crop(inputs=sample, shape=self.shape, crop_type=self.random_crop_type)
        return [T.astype(np.float32) for T in sample]


class FusionNetTrainingDataset(_Samples):
    '''
    Dataset for fetching (1) image (2) depth (3) response (4) ground truth (5) lidar map (src/datasets.py:346-452)

    Arg(s):
        image_paths, depth_paths, response_paths, ground_truth_paths, lidar_map_paths : list[str]
        shape : list[int]
            height, width tuple for random crop
        random_crop_type : list[str]
            none, horizontal, vertical, anchored, top, bottom, left, right, center
        raw : bool
            MI355X path: integer pixels + crop offset instead of cropped float32 arrays (see to_device_batch)
    '''

    def __init__(self, image_paths, depth_paths, response_paths, ground_truth_paths, lidar_map_paths, shape=None,
                 random_crop_type=['none'], raw=False):
        super().__init__(image_paths, [depth_paths, response_paths, ground_truth_paths, lidar_map_paths], shape, random_crop_type, raw)
        self.depth_paths, self.response_paths = depth_paths, response_paths
        self.ground_truth_paths, self.lidar_map_paths = ground_truth_paths, lidar_map_paths

    def __getitem__(self, index):
        return tuple(self._fetch(index))


class FusionNetInferenceDataset(_Samples):
    '''
    Dataset for fetching (1) image (2) depth (3) response (4) ground truth if available (src/datasets.py:455-527)
    '''

    def __init__(self, image_paths, depth_paths, response_paths, ground_truth_paths, raw=False):
        self.ground_truth_available = ground_truth_paths is not None and None not in ground_truth_paths
        if self.ground_truth_available:
            assert len(image_paths) == len(ground_truth_paths)
        # like the reference (:498-499) a missing ground-truth LIST is an error (len(None)); a list holding None is "not available"
        assert len(ground_truth_paths) == len(image_paths)
        maps = [depth_paths, response_paths] + ([ground_truth_paths] if self.ground_truth_available else [])
        super().__init__(image_paths, maps, None, ['none'], raw)
        self.depth_paths, self.response_paths, self.ground_truth_paths = depth_paths, response_paths, ground_truth_paths

    def __getitem__(self, index):
        return self._fetch(index)


def _load_points(path):
    '''N x 3 radar points (x, y, depth); a file holding one point is 1-D (src/datasets.py:176-180, :326-330)'''
    points = np.load(path)
    return points[np.newaxis] if points.ndim == 1 else points


class RadarNetTrainingDataset(torch.utils.data.Dataset):
    '''
    Dataset for fetching (1) image, edge-padded by half a patch on both sides and cut to the bottom patch_size[0] rows,
    (2) total_points_sampled radar points with x shifted into the padded frame, (3) their bounding boxes
    [x - w/2, 0, x + w/2, patch height], (4) the ground-truth crop under each box (src/datasets.py:112-291).

    With probability sample_probability_of_lidar the points are replaced by noisy lidar returns: x and depth of random
    ground-truth pixels deeper than 1 m, x jittered by N(0, 25), depth by U(0, 0.4), y kept (:194-221).  The global numpy RNG
    and Python's `random` are consumed in the reference's order.

    Arg(s):
        image_paths, radar_paths, ground_truth_paths : list[str]
        patch_size : list[int]
            height, width of the crop centred at a radar point
        total_points_sampled : int
            points per image (drawn with replacement; frames with too few points are repeated 100 x first)
        sample_probability_of_lidar : float
    '''

    def __init__(self, image_paths, radar_paths, ground_truth_paths, patch_size, total_points_sampled, sample_probability_of_lidar):
        self.n_sample = len(image_paths)
        assert self.n_sample == len(ground_truth_paths)
        assert self.n_sample == len(radar_paths)
        self.image_paths, self.radar_paths, self.ground_truth_paths = image_paths, radar_paths, ground_truth_paths
        self.patch_size = patch_size
        self.pad_size_x = patch_size[1] // 2
        self.padding = ((0, 0), (0, 0), (self.pad_size_x, self.pad_size_x))
        self.data_format = 'CHW'
        self.total_points_sampled = total_points_sampled
        self.sample_probability_of_lidar = sample_probability_of_lidar

    def __len__(self):
        return self.n_sample

    def __getitem__(self, index):
        k, pad, patch_h = self.total_points_sampled, self.pad_size_x, self.patch_size[0]
        image = np.pad(data_utils.load_image(self.image_paths[index], normalize=False, data_format=self.data_format),
                       pad_width=self.padding, mode='edge')
        points = _load_points(self.radar_paths[index])
        if points.shape[0] <= k:
            points = np.repeat(points, 100, axis=0)
        points = points[np.random.randint(points.shape[0], size=k), :]
        ground_truth = data_utils.load_depth(self.ground_truth_paths[index], data_format=self.data_format)

        if random.random() < self.sample_probability_of_lidar:
            gt = ground_truth.squeeze()
            rows, cols = np.where(gt > 1)
            picked = random.sample(range(0, len(rows)), k)
            px, py = cols[picked], rows[picked]
            noise_x = np.random.normal(0, 25, points.shape[0])
            noise_z = np.random.uniform(low=0.0, high=0.4, size=points.shape[0])
            fake = np.copy(points)
            fake[:, 0] = np.clip(px + noise_x, 0, gt.shape[1])
            fake[:, 2] = gt[py, px] + noise_z
            fake[:, 0] = fake[:, 0].astype(int)      # x and y back to whole pixels (y is the radar's own, kept as is)
            fake[:, 1] = fake[:, 1].astype(int)
            points = fake

        points[:, 0] = points[:, 0] + pad              # into the padded frame: the point is the centre of its patch
        boxes = np.stack([points[:, 0] - pad, np.zeros(k, points.dtype), points[:, 0] + pad,
                          np.full(k, patch_h, points.dtype)], axis=1)
        ground_truth = np.pad(ground_truth, pad_width=self.padding, mode='constant', constant_values=0)
        start_y = image.shape[-2] - patch_h
        crops = np.asarray([ground_truth[:, start_y:, int(x - pad):int(x + pad)] for x in points[:, 0]])
        image = image[:, start_y:, ...]
        return image.astype(np.float32), points.astype(np.float32), boxes.astype(np.float32), crops.astype(np.float32)


class RadarNetInferenceDataset(torch.utils.data.Dataset):
    '''Dataset for fetching (1) image (2) all radar points N x 3 (3) ground truth if available (src/datasets.py:294-343)'''

    def __init__(self, image_paths, radar_paths, ground_truth_paths=None):
        self.n_sample = len(image_paths)
        assert self.n_sample == len(radar_paths)
        self.image_paths, self.radar_paths = image_paths, radar_paths
        self.ground_truth_available = ground_truth_paths is not None and None not in ground_truth_paths
        if self.ground_truth_available:
            assert self.n_sample == len(ground_truth_paths)
        self.ground_truth_paths = ground_truth_paths
        self.data_format = 'CHW'

    def __len__(self):
        return self.n_sample

    def __getitem__(self, index):
        inputs = [data_utils.load_image(self.image_paths[index], normalize=False, data_format=self.data_format),
                  _load_points(self.radar_paths[index])]
        if self.ground_truth_available:
            inputs.append(data_utils.load_depth(self.ground_truth_paths[index], data_format=self.data_format))
        return [T.astype(np.float32) for T in inputs]


def to_device_batch(batch_data, device, shape=None, normalize=False, multiplier=256.0):
    '''
    Raw DataLoader batch ([image u8 (N,H,W,3), map u16 (N,H,W) ..., crop (N,2)]) -> the float32 tensors on `device` the reference's
    DataLoader would have delivered: image (N,3,h,w) 0..255, maps (N,1,h,w) = pixel / 256 with non-positive values zeroed.
    `shape` is the dataset's crop shape (None: full frames).
    '''
    device = torch.device(device)
    if device.type != 'cuda':
        raise ops._lib.RcfError('to_device_batch decodes on the GPU (got device %s); use raw=False for host-side numpy samples' % device)
    *tensors, crop = batch_data
    image = torch.as_tensor(tensors[0])
    if shape is None or not all(x > 0 for x in shape):
        shape, crop = None, None
    out = [ops.decode_images(image.to(device, non_blocking=True), crop, shape, normalize)]
    for m in tensors[1:]:
        out.append(ops.decode_maps(torch.as_tensor(m).to(device, non_blocking=True), multiplier, crop, shape, clamp_nonpositive=True))
    return out
