'''
TEST INFRASTRUCTURE ONLY: CPU restatement of the three torchvision.transforms.functional calls that the reference's
Transforms.transform makes (src/fusionnet_transforms.py:231, :256, :281): adjust_brightness / adjust_contrast / adjust_saturation
on tensors.

PARITY UNPINNED at this boundary (torchvision 0.11.3 is not installed and not under /root/reference); restated from the published
torchvision/transforms/functional_tensor.py:
    _blend(img1, img2, ratio) = (ratio * img1 + (1 - ratio) * img2).clamp(0, bound).to(img1.dtype), bound = 1.0 (float) / 255.0
    rgb_to_grayscale(img)     = (0.2989 r + 0.587 g + 0.114 b).to(img.dtype)
    adjust_brightness(img, f) = _blend(img, zeros_like(img), f)
    adjust_contrast(img, f)   = _blend(img, mean(rgb_to_grayscale(img).to(float dtype)), f)
    adjust_saturation(img, f) = _blend(img, rgb_to_grayscale(img), f)
Everything else of Transforms.transform is the reference's own code (tests/golden/make_golden_transforms.py runs it).
'''
import torch


def _blend(img1, img2, ratio):
    ratio = float(ratio)
    bound = 1.0 if img1.is_floating_point() else 255.0
    return (ratio * img1 + (1.0 - ratio) * img2).clamp(0, bound).to(img1.dtype)


def rgb_to_grayscale(img):
    r, g, b = img.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).to(img.dtype).unsqueeze(dim=-3)


def adjust_brightness(img, brightness_factor):
    return _blend(img, torch.zeros_like(img), brightness_factor)


def adjust_contrast(img, contrast_factor):
    dtype = img.dtype if torch.is_floating_point(img) else torch.float32
    mean = torch.mean(rgb_to_grayscale(img).to(dtype), dim=(-3, -2, -1), keepdim=True)
    return _blend(img, mean, contrast_factor)


def adjust_saturation(img, saturation_factor):
    return _blend(img, rgb_to_grayscale(img), saturation_factor)
