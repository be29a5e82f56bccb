'''GPU box: every convolution launch of one eager training (or inference) step with its descriptor, kernel id and duration (HIP events
around the launch), sorted by time -- which layer runs on which kernel and what it costs.
  RCF_DTYPE=bf16|fp32   RCF_MODE=train|infer   RCF_N=batch'''
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rcf_amd import ops, synth, train     # noqa: E402
import bench                              # noqa: E402

ROWS = []


def _wrap(name, kid_of):
    real = getattr(ops, name)

    def timed(desc, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real(desc, *a, **k)
        e1.record()
        ROWS.append((name, desc.w_mode, desc.ksize, desc.stride, desc.c1 + desc.c2, desc.c_out, desc.h_out, desc.w_out, desc.accumulate,
                     kid_of(ops.conv_query(desc)), e0, e1))
        return r
    setattr(ops, name, timed)


def main():
    dtype = os.environ.get('RCF_DTYPE', 'bf16')
    mode = os.environ.get('RCF_MODE', 'train')
    n = int(os.environ.get('RCF_N', '8'))
    model = train.build_model(synth.PUBLISHED, device=torch.device('cuda:0'))
    synth.fill_state_dict_([model.encoder, model.decoder], 7)
    model.compute_dtype = 'bf16' if dtype == 'bf16' else 'fp32'
    b = {k: v.cuda() for k, v in synth.make_batch(n, 900, 1600, 64, seed=3).items()}
    opt = train.make_optimizer(model, lr=1e-4)

    def step():
        if mode == 'train':
            model.train()
            train.train_step(model, opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])
        else:
            model.eval()
            with torch.no_grad():
                model.forward(b['image'], b['input_depth'])
    step()
    step()
    torch.cuda.synchronize()
    for name in ('conv_fwd', 'conv_fwd_act', 'conv_dgrad_bn_sums'):
        _wrap(name, lambda i: i.kernel_id)
    _wrap('conv_wgrad', lambda i: i.wgrad_kernel_id)
    step()
    torch.cuda.synchronize()
    rows = [(r[-2].elapsed_time(r[-1]) * 1e3,) + r[:-2] for r in ROWS]
    total = sum(r[0] for r in rows)
    print('%d convolution launches, %.2f ms' % (len(rows), total / 1e3))
    agg = {}
    for r in rows:
        key = r[1:]
        t, c = agg.get(key, (0.0, 0))
        agg[key] = (t + r[0], c + 1)
    for key, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        name, wm, k, s, ci, co, h, w, acc, kid = key
        print('%8.1f us x %2d  %-18s wmode %d k%d s%d %4d -> %4d @ %4d x %4d acc %d  id %6d %s'
              % (t / c, c, name, wm, k, s, ci, co, h, w, acc, kid, bench.decode_kernel_id(kid)[1][:40]))


if __name__ == '__main__':
    main()
