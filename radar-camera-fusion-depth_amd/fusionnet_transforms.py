'''
Transforms -- drop-in for the reference's src/fusionnet_transforms.py (same constructor, same transform() signature and return
convention), running the whole augmentation as a few batched HIP kernels: no per-sample Python loops, no torch.max host
synchronisation (the 0..255-vs-0..1 decision of :81-83 is taken on the device).  SURVEY.md 8 f-3.

The random draws follow the reference's order (one torch.rand(N) per decision and per factor, :75-163) on the images' device, or
on `rng_device` when given (a CPU generator reproduces the reference's CPU stream, which the tests use).
'''
import torch

from . import _lib, ops


class Transforms(object):

    def __init__(self,
                 normalized_image_range=[0, 255],
                 random_brightness=[-1],
                 random_contrast=[-1],
                 random_saturation=[-1],
                 random_flip_type=['none'],
                 rng_device=None):
        self.normalized_image_range = list(normalized_image_range)
        if self.normalized_image_range == [0, 1]:
            self._norm_mode = 1
        elif self.normalized_image_range == [-1, 1]:
            self._norm_mode = 2
        elif self.normalized_image_range == [0, 255]:
            self._norm_mode = 0
        else:
            raise ValueError('Unsupported normalization range: {}'.format(normalized_image_range))
        self.do_random_brightness = True if -1 not in random_brightness else False
        self.random_brightness = random_brightness
        self.do_random_contrast = True if -1 not in random_contrast else False
        self.random_contrast = random_contrast
        self.do_random_saturation = True if -1 not in random_saturation else False
        self.random_saturation = random_saturation
        self.do_random_horizontal_flip = True if 'horizontal' in random_flip_type else False
        self.do_random_vertical_flip = True if 'vertical' in random_flip_type else False
        self.rng_device = rng_device

    def draw(self, n_batch, device, random_transform_probability):
        '''The reference's random decisions, in its order (src/fusionnet_transforms.py:75-163): a dict of device tensors.'''
        rdev = self.rng_device if self.rng_device is not None else device
        d = {}
        do_random_transform = torch.rand(n_batch, device=rdev) <= random_transform_probability

        def photometric(name, rng):
            do = torch.logical_and(do_random_transform, torch.rand(n_batch, device=rdev) <= 0.50)
            values = torch.rand(n_batch, device=rdev)
            lo, hi = rng
            d['do_' + name] = do
            d['f_' + name] = (hi - lo) * values + lo

        if self.do_random_brightness:
            photometric('brightness', self.random_brightness)
        if self.do_random_contrast:
            photometric('contrast', self.random_contrast)
        if self.do_random_saturation:
            photometric('saturation', self.random_saturation)
        if self.do_random_horizontal_flip:
            d['do_hflip'] = torch.logical_and(do_random_transform, torch.rand(n_batch, device=rdev) <= 0.50)
        if self.do_random_vertical_flip:
            d['do_vflip'] = torch.logical_and(do_random_transform, torch.rand(n_batch, device=rdev) <= 0.50)
        return {k: (v.to(torch.uint8) if k.startswith('do_') else v.to(torch.float32)).to(device) for k, v in d.items()}

    def apply(self, images_arr, range_maps_arr, decisions):
        '''transform() with given decisions (see draw()); returns (images list, range maps list).'''
        g = decisions.get
        images_out = [ops.transform_images(images.to(torch.float32), g('do_brightness'), g('f_brightness'), g('do_contrast'),
                                           g('f_contrast'), g('do_saturation'), g('f_saturation'), g('do_hflip'), g('do_vflip'),
                                           self._norm_mode) for images in images_arr]
        if g('do_hflip') is None and g('do_vflip') is None:
            maps_out = list(range_maps_arr)
        else:
            maps_out = [ops.transform_flip(m.to(torch.float32), g('do_hflip'), g('do_vflip')) for m in range_maps_arr]
        return images_out, maps_out

    def transform(self, images_arr, range_maps_arr=[], random_transform_probability=0.50):
        '''
        Applies transform to images and ground truth (src/fusionnet_transforms.py:46-178)

        Arg(s):
            images_arr : list[torch.Tensor]
                list of N x C x H x W tensors
            range_maps_arr : list[torch.Tensor]
                list of N x c x H x W tensors
            random_transform_probability : float
                probability to perform transform
        Returns:
            list[torch.Tensor[float32]] : list of transformed N x C x H x W image tensors
            list[torch.Tensor[float32]] : list of transformed N x c x H x W range maps tensors
        '''
        if images_arr[0].ndim != 4:
            raise ValueError('Unsupported number of dimensions: {}'.format(images_arr[0].ndim))
        if not images_arr[0].is_cuda:
            raise _lib.RcfError('Transforms.transform needs CUDA(HIP) tensors: the augmentation kernels are HIP-only')
        n_batch = images_arr[0].shape[0]
        decisions = self.draw(n_batch, images_arr[0].device, random_transform_probability)
        images_out, maps_out = self.apply(list(images_arr), list(range_maps_arr), decisions)
        outputs = []
        if len(images_out) > 0:
            outputs.append(images_out)
        if len(maps_out) > 0:
            outputs.append(maps_out)
        if len(outputs) == 1:
            return outputs[0]
        return outputs
