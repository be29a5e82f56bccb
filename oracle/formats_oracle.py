'''
TEST INFRASTRUCTURE ONLY: numpy restatement of the sample formats of the FusionNet loaders in the batched,
integer-in form of the device kernels (rcf_decode_image_u8 / rcf_decode_map / rcf_encode_map_u32 / rcf_points_to_depth_map).

Follows src/data_utils.py:167-335 (load_image, load_depth, load_depth_with_validity_map, load_response, save_depth,
save_response), the slicing of src/datasets.py:101-109 (random_crop) and setup/setup_dataset_nuscenes_with_denseGT.py:814-840
(points_to_depth_map).  PINNED by tests/golden/T9_formats (outputs of the real reference functions on the committed PNG files,
made by tests/golden/make_golden_formats.py); checked in tests/test_oracle_golden.py.
'''
import numpy as np


def _window(crop_yx, b, shape, src_shape):
    h, w = src_shape if shape is None else shape
    y0, x0 = (0, 0) if crop_yx is None else (int(crop_yx[b][0]), int(crop_yx[b][1]))
    return slice(y0, y0 + h), slice(x0, x0 + w)


def decode_images(raw, crop_yx=None, shape=None, normalize=False):
    '''uint8 (N, H, W, 3) -> float32 (N, 3, h, w): np.asarray(image, np.float32), CHW transpose, optional / 255.0
    (src/data_utils.py:184-196), then T[:, y0:y1, x0:x1] per sample (src/datasets.py:105-107).'''
    out = []
    for b in range(raw.shape[0]):
        ys, xs = _window(crop_yx, b, shape, raw.shape[1:3])
        a = np.transpose(raw[b].astype(np.float32), (2, 0, 1))[:, ys, xs]
        out.append(a / 255.0 if normalize else a)
    return np.stack(out).astype(np.float32)


def decode_maps(raw, multiplier=256.0, crop_yx=None, shape=None, clamp_nonpositive=True):
    '''integer (N, H, W) -> (depth, validity) float32 (N, 1, h, w): z = float32(pixel) / multiplier; z[z <= 0] = 0 for the depth
    loaders (src/data_utils.py:217-223, :254-258), not for load_response (:305-308); validity = 1 where z > 0.'''
    zs = []
    for b in range(raw.shape[0]):
        ys, xs = _window(crop_yx, b, shape, raw.shape[1:3])
        z = raw[b].astype(np.float32) / multiplier
        if clamp_nonpositive:
            z[z <= 0] = 0.0
        zs.append(z[ys, xs][np.newaxis])
    z = np.stack(zs).astype(np.float32)
    v = z.copy()
    v[z > 0] = 1.0
    return z, v


def encode_maps(z, multiplier=256.0):
    '''np.uint32(z * multiplier) of save_depth / save_response (src/data_utils.py:284, :333)'''
    return np.uint32(np.asarray(z, np.float32) * multiplier)


def points_to_depth_map(points, depth, height, width):
    '''depth_map[round(y_k), round(x_k)] = depth[k] for k in order: the last point of a pixel wins; np.round is half-to-even and
    negative indices address from the end, like the numpy indexing of the original (setup/...denseGT.py:829-838).'''
    depth_map = np.zeros((height, width))
    q = np.round(points).astype(int)
    for k in range(q.shape[1]):
        depth_map[q[1, k], q[0, k]] = depth[k]
    return depth_map
