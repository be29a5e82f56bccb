'''GPU box: which kernel every convolution of the bf16 inference forward runs on (ksize, stride, cin, cout, h, w, kernel id).'''
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rcf_amd import synth, train                           # noqa: E402
import bench                                               # noqa: E402


def main():
    dtype = os.environ.get('RCF_DTYPE', 'bf16')
    n = int(os.environ.get('RCF_N', '4'))
    model = train.build_model(synth.PUBLISHED, device=torch.device('cuda:0'))
    synth.fill_state_dict_([model.encoder, model.decoder], 7)
    model.compute_dtype = 'bf16' if dtype == 'bf16' else 'fp32'
    model.eval()
    batch = synth.make_batch(n, 900, 1600, 64, seed=3)
    image, depth = batch['image'].cuda(), batch['input_depth'].cuda()
    eng = model._engine
    eng.kernel_log = []
    with torch.no_grad():
        model.forward(image, depth)
    torch.cuda.synchronize()
    for row in eng.kernel_log:
        print('k%d s%d  %4d -> %4d  @ %4d x %4d   id %6d  %s' % (row + (bench.decode_kernel_id(row[-1]),)))


if __name__ == '__main__':
    main()
