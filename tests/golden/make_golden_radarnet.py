'''
Golden fixture T5 for RadarNet stage 1: runs the REAL reference RadarNetModel (imported read-only from
/root/reference/src, CPU, fp32) on seeded inputs/weights and stores output, loss, gradients and BN buffers.

torchvision is absent from this image, so `torchvision.ops.roi_pool` -- the one call on this path that is not reference code
(src/networks.py:1232-1247) -- is provided by oracle/roi_pool_oracle.py, the restatement of torchvision 0.11's kernel.  Parity
is therefore UNPINNED at that boundary and pinned everywhere else (encoder, MLP, decoder, loss are the reference's own code).

    python tests/golden/make_golden_radarnet.py
'''
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = '/root/reference/src'
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle.roi_pool_oracle import roi_pool


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    tv = _stub('torchvision')
    tv.ops = _stub('torchvision.ops', roi_pool=roi_pool)
    tv.utils = _stub('torchvision.utils')
    tv.transforms = _stub('torchvision.transforms')
    def tv_pad(img, padding, padding_mode='constant'):
        l, t, r, b = padding
        x = img if img.dim() == 4 else img.unsqueeze(0)
        return torch.nn.functional.pad(x, (l, r, t, b), mode={'edge': 'replicate', 'constant': 'constant'}[padding_mode]) \
            .reshape(*img.shape[:-2], img.shape[-2] + t + b, img.shape[-1] + l + r)
    tv.transforms.functional = _stub('torchvision.transforms.functional', pad=tv_pad)
    _stub('torch.utils.tensorboard', SummaryWriter=object)
    if not hasattr(np, 'infty'):
        np.infty = np.inf
    sys.path.insert(0, REF)
    import radarnet_model  # noqa
    return radarnet_model


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)   # the fixtures regenerate bit-identically only at a fixed intra-op thread count (oneDNN reduction order)
    from rcf_amd import synth
    ref_mod = import_reference()
    gold = os.path.dirname(os.path.abspath(__file__))
    model = ref_mod.RadarNetModel(device=torch.device('cpu'), **synth.RADARNET_TINY)
    synth.fill_state_dict_([model.encoder, model.decoder], 31)
    b = synth.make_radarnet_batch(501)
    image, points, boxes, gt, valid = b['image'], b['point'], b['bounding_boxes'], b['ground_truth'], b['validity_map']
    model.train()
    logits = model.forward(image, points, boxes, return_logits=True)
    loss, _ = model.compute_loss(logits, gt, valid, w_positive_class=2.0)
    loss.backward()
    # ---- pin oracle/radarnet_oracle.py: same weights, same inputs -> identical logits, loss, gradients (stock torch ops both sides)
    from oracle.radarnet_oracle import RadarNetOracle
    ora = RadarNetOracle(**synth.RADARNET_TINY)
    synth.fill_state_dict_([ora.encoder, ora.decoder], 31)
    assert list(ora.encoder.state_dict().keys()) == list(model.encoder.state_dict().keys())
    assert list(ora.decoder.state_dict().keys()) == list(model.decoder.state_dict().keys())
    ora.train()
    ol = ora.forward(image, points, boxes)
    oloss = ora.compute_loss(ol, gt, valid, 2.0)
    oloss.backward()
    assert float((ol - logits).abs().max()) < 1e-5 * float(logits.abs().max()), 'oracle logits differ from the reference'
    assert abs(float(oloss) - float(loss)) < 1e-6 * abs(float(loss))
    for (k, p), (k2, p2) in zip(list(model.encoder.named_parameters()) + list(model.decoder.named_parameters()),
                                list(ora.encoder.named_parameters()) + list(ora.decoder.named_parameters())):
        assert k == k2 and (p.grad is None) == (p2.grad is None), k
        if p.grad is not None:
            assert float((p.grad - p2.grad).abs().max()) <= 5e-4 * float(p.grad.abs().max()) + 1e-9, k
    print('[T5] oracle/radarnet_oracle.py == reference (logits, loss, every gradient)')
    named = [('encoder.' + k, p) for k, p in model.encoder.named_parameters()] + \
            [('decoder.' + k, p) for k, p in model.decoder.named_parameters()]
    bufs = [('encoder.' + k, b) for k, b in model.encoder.named_buffers() if not k.endswith('num_batches_tracked')] + \
           [('decoder.' + k, b) for k, b in model.decoder.named_buffers() if not k.endswith('num_batches_tracked')]
    unused = sorted(k for k, p in named if p.grad is None)
    print('T5: logits', tuple(logits.shape), 'loss %.6f' % float(loss), '%d params, %d unused' % (len(named), len(unused)))
    # eval-mode output with the updated running statistics
    model.eval()
    with torch.no_grad():
        ev = model.forward(image, points, boxes, return_logits=False)
    np.savez_compressed(
        os.path.join(gold, 'T5_radarnet_tiny_train.npz'),
        meta=np.array([501, 31]), logits=logits.detach().numpy(), loss=np.array(float(loss), np.float64),
        eval_sigmoid=ev.numpy(), unused=np.array(unused),
        **{'grad:' + k: p.grad.numpy() for k, p in named if p.grad is not None},
        **{'buf:' + k: b.detach().numpy() for k, b in bufs})
    print('%-32s %8.1f KB' % ('T5_radarnet_tiny_train.npz', os.path.getsize(os.path.join(gold, 'T5_radarnet_tiny_train.npz')) / 1024.0))

    # ---- T7: stage-1 inference glue radarnet_main.forward (src/radarnet_main.py:534-591) with the real model: image + points ->
    # dense radar depth / response maps
    import radarnet_main
    model.eval()
    rs = np.random.RandomState(77)
    h7, w7, k7, pw7 = 64, 96, 6, 32
    img7 = torch.from_numpy(rs.rand(1, 3, h7, w7).astype(np.float32))
    pts7 = np.stack([rs.uniform(0, w7, k7) + pw7 // 2, rs.uniform(0, h7, k7), rs.uniform(1.0, 9.0, k7)], -1).astype(np.float32)
    pts7[:, 2] = np.array([0.7, 3.2, 5.9, 2.4, 1.1, 4.6], np.float32)   # small z: the in-place replacement chain fires
    boxes7 = [torch.from_numpy(np.stack([pts7[:, 0] - pw7 // 2, np.zeros(k7), pts7[:, 0] + pw7 // 2, np.full(k7, h7)], -1).astype(np.float32))]
    with torch.no_grad():
        d7, r7 = radarnet_main.forward(model, img7, torch.from_numpy(pts7.copy()), boxes7, device=torch.device('cpu'))
    print('T7: depth', tuple(d7.shape), 'nonzero response pixels', int((r7 > 0).sum()), 'distinct depths', sorted(set(d7.flatten().tolist()))[:8])
    np.savez_compressed(os.path.join(gold, 'T7_radarnet_forward_scatter.npz'), image=img7.numpy(), points=pts7,
                        depth=d7.numpy().astype(np.float32), response=r7.numpy())

    # ---- T6: the shipped channel configuration (bash/train_radarnet_nuscenes.sh:27-31) on a small image / patch (96 x 64):
    # exercises the >= 16-channel kernels; logits in full, gradients as L2 norms and sums
    cfg = dict(synth.RADARNET_PUBLISHED)
    cfg['input_patch_size_image'] = (96, 64)
    model = ref_mod.RadarNetModel(device=torch.device('cpu'), **cfg)
    synth.fill_state_dict_([model.encoder, model.decoder], 32)
    b = synth.make_radarnet_batch(502, n=2, k=3, h=96, w=160, patch_w=64)
    model.train()
    logits = model.forward(b['image'], b['point'], b['bounding_boxes'], return_logits=True)
    loss, _ = model.compute_loss(logits, b['ground_truth'], b['validity_map'], w_positive_class=2.0)
    loss.backward()
    named = [('encoder.' + k, p) for k, p in model.encoder.named_parameters()] + \
            [('decoder.' + k, p) for k, p in model.decoder.named_parameters()]
    keys = [k for k, p in named if p.grad is not None]
    grads = dict(named)
    print('T6: logits', tuple(logits.shape), 'loss %.6f' % float(loss.detach()))
    np.savez_compressed(
        os.path.join(gold, 'T6_radarnet_published_channels.npz'),
        meta=np.array([502, 32, 2, 3, 96, 160, 64]), logits=logits.detach().numpy(), loss=np.array(float(loss.detach()), np.float64),
        grad_keys=np.array(keys),
        grad_l2=np.array([float(grads[k].grad.double().norm()) for k in keys]),
        grad_sum=np.array([float(grads[k].grad.double().sum()) for k in keys]))
    print('%-32s %8.1f KB' % ('T6_radarnet_published_channels.npz',
                              os.path.getsize(os.path.join(gold, 'T6_radarnet_published_channels.npz')) / 1024.0))


if __name__ == '__main__':
    main()
