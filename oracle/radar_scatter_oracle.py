'''
ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

numpy restatement of the radar point -> dense map scatter in the reference's radarnet_main.forward
(src/radarnet_main.py:563-589), i.e. everything after RadarNet has produced one sigmoid response crop per
radar point.  Parity status: PINNED -- tests/golden/make_golden_scatter.py runs the reference function itself
(with a stand-in model that returns prescribed crops) and stores inputs' seeds + outputs in
tests/golden/T4_radar_scatter.npz; tests/test_oracle_golden.py re-checks this restatement against them.

crops  : K x Hc x Wc float32 sigmoid responses (Wc = 2 * pad)
points : K x 3 (x in PADDED canvas coordinates, y, z in metres) as passed to the reference
returns (depth H x W, response H x W) with H the image height and W the unpadded width.

Quirks reproduced when strict_reference:
  (i)  the argmax tensor is int64, so every z is truncated toward zero by torch.full_like(int64, z);
  (ii) the replacement `where(output == k, z_k, output)` runs in place for k = 0..K-1, so a pixel already replaced
       by int(z_j) is replaced AGAIN when int(z_j) equals a later index k.
'''
import numpy as np


def radar_scatter(crops, points, width, strict_reference=True):
    k, hc, wc = crops.shape
    pad = wc // 2
    height = hc                                   # crop_height = height - patch_size[0] = 0 (full-height crops)
    canvas_w = width + 2 * pad
    tiles = np.zeros((k, height, canvas_w), np.float32)
    for i in range(k):
        crop = np.where(crops[i] < 0.5, 0.0, crops[i]).astype(np.float32)      # :567
        x = int(points[i, 0])
        tiles[i, height - hc:, x - pad:x + pad] = crop                         # :569
    tiles = tiles[:, :, pad:canvas_w - pad]                                    # :573
    response = tiles.max(axis=0)                                               # :576
    output = tiles.argmax(axis=0).astype(np.int64)                             # first maximal index, as torch.max
    if strict_reference:
        for i in range(k):                                                     # :579-583
            output = np.where(output == i, np.int64(points[i, 2]), output)     # int64 fill truncates z toward zero
        depth = output.astype(np.float32)
    else:
        depth = points[output, 2].astype(np.float32)
    depth = np.where(response == 0, 0.0, depth).astype(np.float32)             # :586-589
    return depth, response
