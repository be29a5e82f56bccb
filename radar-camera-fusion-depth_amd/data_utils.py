'''
File formats of the method's samples -- mirror of the loaders / writers in src/data_utils.py:128-335 (same names, arguments and
return values, so `import data_utils` call sites of the reference work unchanged):

    image      8-bit RGB PNG                      -> float32 H x W x 3 (or 3 x H x W), 0..255 or 0..1
    depth      16-bit PNG, metres x 256           -> float32 map, non-positive values zeroed (+ validity map)
    response   16-bit PNG, x 2^14                 -> float32 map
    paths      newline-delimited text file

The functions above return numpy arrays because that is their contract (they run in DataLoader workers on the host).  The
`*_raw` loaders stop after the PNG is inflated and return the INTEGER pixels; `rcf_amd.datasets` ships those to the GPU and
finishes crop + layout + float conversion there in one launch per tensor (ops.decode_images / ops.decode_maps).
'''
import numpy as np
from PIL import Image


def read_paths(filepath):
    '''Reads a newline delimited file containing paths (src/data_utils.py:128-150): stops at the first empty line.'''
    paths = []
    with open(filepath) as f:
        for line in f:
            line = line.rstrip('\n')
            if line == '':
                break
            paths.append(line)
    return paths


def write_paths(filepath, paths):
    '''Stores line delimited paths into file (src/data_utils.py:152-165)'''
    with open(filepath, 'w') as f:
        f.writelines(p + '\n' for p in paths)


def _with_channel(a, data_format):
    '''HW -> HW / 1HW / HW1'''
    if data_format == 'HW':
        return a
    if data_format == 'CHW':
        return a[np.newaxis]
    if data_format == 'HWC':
        return a[..., np.newaxis]
    raise ValueError('Unsupported data format: {}'.format(data_format))


def load_image_raw(path):
    '''RGB pixels as stored: uint8 H x W x 3'''
    return np.array(Image.open(path).convert('RGB'), np.uint8)


def load_map_raw(path):
    '''Pixels of a depth / response PNG as stored: uint16 for the 16-bit files save_depth writes (uint8 / int32 for other modes)'''
    a = np.array(Image.open(path))
    if a.dtype not in (np.uint8, np.uint16, np.int32):
        a = a.astype(np.int32)
    return a


def load_image(path, normalize=False, data_format='HWC'):
    '''
    Loads an RGB image (src/data_utils.py:167-198)

    Returns:
        numpy[float32] : H x W x C or C x H x W image, 0..255 (0..1 if normalize)
    '''
    image = load_image_raw(path).astype(np.float32)
    if data_format == 'CHW':
        image = np.transpose(image, (2, 0, 1))
    elif data_format != 'HWC':
        raise ValueError('Unsupported data format: {}'.format(data_format))
    return image / 255.0 if normalize else image


def load_depth_with_validity_map(path, multiplier=256.0, data_format='HW'):
    '''
    Loads a depth map and validity map from a 16-bit PNG file (src/data_utils.py:200-236)

    Returns:
        numpy[float32] : depth map
        numpy[float32] : binary validity map for available depth measurement locations
    '''
    z = load_depth(path, multiplier, 'HW')
    v = np.where(z > 0, np.float32(1.0), z).astype(np.float32)
    return _with_channel(z, data_format), _with_channel(v, data_format)


def load_depth(path, multiplier=256.0, data_format='HW'):
    '''Loads a depth map from a 16-bit PNG file (src/data_utils.py:238-269): pixel / multiplier, non-positive values -> 0'''
    z = np.array(Image.open(path), dtype=np.float32) / multiplier
    z[z <= 0] = 0.0
    return _with_channel(z, data_format)


def save_depth(z, path, multiplier=256.0):
    '''Saves a depth map to a 16-bit PNG file (src/data_utils.py:271-286): uint32(z * multiplier) written as PIL mode I'''
    Image.fromarray(np.uint32(z * multiplier), mode='I').save(path)


def load_response(path, multiplier=2**14, data_format='HW'):
    '''Loads a response map from a 16-bit PNG file (src/data_utils.py:288-318): pixel / multiplier, nothing zeroed'''
    return _with_channel(np.array(Image.open(path), dtype=np.float32) / multiplier, data_format)


def save_response(response, path, multiplier=2**14):
    '''Saves a response map to a 16-bit PNG file (src/data_utils.py:320-335)'''
    Image.fromarray(np.uint32(response * multiplier), mode='I').save(path)


def save_encoded(encoded, path):
    '''Writes the output of ops.encode_maps (uint32 bit patterns, any integer dtype, H x W) exactly like save_depth does.'''
    Image.fromarray(np.asarray(encoded).astype(np.uint32, copy=False), mode='I').save(path)
