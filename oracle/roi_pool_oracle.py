'''
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of torchvision.ops.roi_pool as RadarNetV1Encoder.forward
calls it (src/networks.py:1232-1247).

PARITY UNPINNED at this boundary: torchvision (pinned at 0.11.3+cu113, requirements.txt:142) is not installed in this image and
its source is not under /root/reference, and no reference test holds a golden vector for it.  The algorithm below restates the
published torchvision 0.11 CPU kernel (torchvision/csrc/ops/cpu/roi_pool_kernel.cpp):
  roi_start_w = round(x1 * scale), roi_start_h = round(y1 * scale), roi_end_w = round(x2 * scale), roi_end_h = round(y2 * scale)
  roi_width = max(roi_end_w - roi_start_w + 1, 1), roi_height likewise; bin_size = roi_size / pooled_size (float)
  hstart = floor(ph * bin_h), hend = ceil((ph + 1) * bin_h), both + roi_start_h, clipped to [0, H]; same for w
  empty bin -> 0; else the maximum over the bin (first maximum in row-major order wins the argmax); backward adds the output
  gradient to the argmax position.
round() is C round (half away from zero).  The forward is written with plain tensor indexing so torch.autograd provides the
backward (a gather through the argmax), which is exactly the kernel's scatter-add.
'''
import math

import torch


def _cround(v):
    return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)


def _bin_edges(n_bins, bin_size, start, limit):
    '''[floor(p * bin) + start, ceil((p + 1) * bin) + start) clipped to [0, limit], bin edges computed in fp32 like the kernel.'''
    b = _f32(bin_size)
    lo = [min(max(int(math.floor(_f32(p * b))) + start, 0), limit) for p in range(n_bins)]
    hi = [min(max(int(math.ceil(_f32((p + 1) * b))) + start, 0), limit) for p in range(n_bins)]
    return lo, hi


def roi_pool(input, boxes, output_size, spatial_scale=1.0):
    '''input (N,C,H,W); boxes: list of (K,4) tensors (x1,y1,x2,y2), one per image, or a (R,5) tensor; returns (R,C,PH,PW).

    A bin is a rectangle, so its maximum is taken separably: first over the columns of each bin column (one (C,H) strip per pw), then
    over the rows of each bin row -- PH + PW tensor ops per ROI instead of PH * PW, which is what makes the shipped 900x288 patches
    (450 x 144 bins at the first skip) usable as a test oracle.  `max(dim)` returns the FIRST maximum along the reduced axis, so the
    element autograd routes the gradient to is the first row holding the bin maximum and the first column of that row holding it:
    the row-major first maximum, the kernel's argmax.'''
    if isinstance(output_size, int):
        output_size = (output_size, output_size)
    ph_n, pw_n = int(output_size[0]), int(output_size[1])
    if isinstance(boxes, (list, tuple)):
        rois = torch.cat([torch.cat([torch.full((b.shape[0], 1), float(i), dtype=b.dtype), b.detach().cpu().to(b.dtype)], 1)
                          for i, b in enumerate(boxes)], 0)
    else:
        rois = boxes.detach().cpu()
    n, c, h, w = input.shape
    outs = []
    for r in range(rois.shape[0]):
        b = int(rois[r, 0])
        x0, y0 = _cround(float(rois[r, 1]) * spatial_scale), _cround(float(rois[r, 2]) * spatial_scale)
        x1, y1 = _cround(float(rois[r, 3]) * spatial_scale), _cround(float(rois[r, 4]) * spatial_scale)
        rw, rh = max(x1 - x0 + 1, 1), max(y1 - y0 + 1, 1)
        bh, bw = float(torch.tensor(rh, dtype=torch.float32) / ph_n), float(torch.tensor(rw, dtype=torch.float32) / pw_n)
        hs, he = _bin_edges(ph_n, bh, y0, h)
        ws, we = _bin_edges(pw_n, bw, x0, w)
        img = input[b]
        strips = []   # per bin column: maximum over its columns, (C, H); an empty column range -> zeros (the bin is empty)
        for pw in range(pw_n):
            if we[pw] <= ws[pw]:
                strips.append(img.new_zeros((c, h)))
            else:
                strips.append(img[:, :, ws[pw]:we[pw]].max(dim=2).values)
        strips = torch.stack(strips, dim=2)   # (C, H, PW)
        rows = []
        for ph in range(ph_n):
            if he[ph] <= hs[ph]:
                rows.append(img.new_zeros((c, pw_n)))
            else:
                rows.append(strips[:, hs[ph]:he[ph], :].max(dim=1).values)
        out = torch.stack(rows, dim=1)        # (C, PH, PW)
        empty_w = torch.tensor([we[pw] <= ws[pw] for pw in range(pw_n)])
        if bool(empty_w.any()):               # empty bin -> 0 even when the strip rows hold negative values
            out = torch.where(empty_w.view(1, 1, pw_n), torch.zeros_like(out), out)
        outs.append(out)
    return torch.stack(outs, dim=0)


def _f32(v):
    '''float32 rounding of a Python float (the kernels compute bin edges in fp32).'''
    return float(torch.tensor(v, dtype=torch.float32))
