'''
Generates the committed golden fixtures by running the REAL reference (imported read-only from
/root/reference, CPU, fp32) and pins the oracle (oracle/fusionnet_oracle.py) against it.

Run in the build container only (the reference never travels to the GPU box):

    python tests/golden/make_golden.py

Import shims: torchvision and tensorboard are absent here and are never called
on the FusionNet forward/backward path, so stub modules satisfy the import statements at
src/networks.py:3 and src/fusionnet_model.py:1.  sys.dont_write_bytecode keeps the reference
tree untouched.

Fixtures (inputs come from rcf_amd.synth with the recorded seeds, so only outputs are stored):
  T0  tiny net, train mode, 2x3x70x102 (odd sizes at every level): output, loss terms,
      every parameter gradient, BN running stats after the step.
  T1  published net, train mode, config #1 (1x3x224x384, 32 radar points): output, loss terms,
      per-parameter gradient L2 norms and sums, running-stat norms.
  T2  tiny net, 3 Adam steps (lr 1e-3): loss trajectory and final-parameter checksum.
  T3  eval-mode (running-stat BN) outputs for the tiny and the published net.
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


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    tv = _stub('torchvision')
    tv.ops = _stub('torchvision.ops')
    tv.utils = _stub('torchvision.utils')
    tv.transforms = _stub('torchvision.transforms')
    tv.transforms.functional = _stub('torchvision.transforms.functional')
    if not hasattr(np, 'infty'):
        np.infty = np.inf
    sys.path.insert(0, REF)
    import fusionnet_model  # noqa
    return fusionnet_model


def build_reference(ref_mod, cfg):
    return ref_mod.FusionNetModel(
        input_channels_image=cfg['input_channels_image'],
        input_channels_depth=cfg['input_channels_depth'],
        encoder_type=['fusionnet18', 'batch_norm'],
        n_filters_encoder_image=cfg['n_filters_encoder_image'],
        n_filters_encoder_depth=cfg['n_filters_encoder_depth'],
        fusion_type='weight_and_project',
        decoder_type=['multiscale', 'batch_norm'],
        n_resolution_decoder=1,
        n_filters_decoder=cfg['n_filters_decoder'],
        deconv_type='up',
        activation_func='leaky_relu',
        weight_initializer='kaiming_uniform',
        min_predict_depth=1.0,
        max_predict_depth=100.0,
        device=torch.device('cpu'))


def ref_loss(model, batch, output):
    loss, info = model.compute_loss(
        image=batch['image'], output_depth=output,
        ground_truth=batch['ground_truth'], lidar_map=batch['lidar_map'],
        loss_func='l1', w_smoothness=0.0, loss_smoothness_kernel_size=-1,
        validity_map_loss_smoothness=None, w_lidar_loss=2.0)
    return loss, info['loss_supervised'], info['loss_lidar']


def named_params(model):
    out = []
    for prefix, mod in (('encoder.', model.encoder), ('decoder.', model.decoder)):
        for k, p in mod.named_parameters():
            out.append((prefix + k, p))
    return out


def named_buffers(model):
    out = []
    for prefix, mod in (('encoder.', model.encoder), ('decoder.', model.decoder)):
        for k, b in mod.named_buffers():
            if not k.endswith('num_batches_tracked'):
                out.append((prefix + k, b))
    return out


def one_step(model, batch, is_ref):
    model.train()
    for _, p in named_params(model):
        p.grad = None
    out = model.forward(batch['image'], batch['input_depth'])
    if is_ref:
        loss, ls, ll = ref_loss(model, batch, out)
    else:
        loss, ls, ll = model.compute_loss(out, batch['ground_truth'], batch['lidar_map'], 2.0)
    loss.backward()
    grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in named_params(model)}
    bufs = {k: b.detach().clone() for k, b in named_buffers(model)}
    return out.detach(), [float(loss), float(ls), float(ll)], grads, bufs


def relerr(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def compare_step(tag, r, o):
    ro, rl, rg, rb = r
    oo, ol, og, ob = o
    e_out = relerr(oo, ro)
    e_loss = max(abs(a - b) / abs(b) for a, b in zip(ol, rl))
    e_grad = 0.0
    for k in rg:
        assert (rg[k] is None) == (og[k] is None), 'grad None-ness differs at ' + k
        if rg[k] is not None:
            e_grad = max(e_grad, relerr(og[k], rg[k]))
    e_buf = max(relerr(ob[k], rb[k]) for k in rb)
    print('[%s] oracle vs reference: out %.2e  loss %.2e  grad %.2e  bn-buffers %.2e'
          % (tag, e_out, e_loss, e_grad, e_buf))
    assert e_out < 2e-5 and e_loss < 2e-5 and e_grad < 5e-4 and e_buf < 2e-5, tag
    return dict(out=e_out, loss=e_loss, grad=e_grad, buf=e_buf)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from rcf_amd import synth
    from oracle.fusionnet_oracle import FusionNetOracle
    ref_mod = import_reference()
    gold = os.path.dirname(os.path.abspath(__file__))

    def pair(cfg, seed):
        ref = build_reference(ref_mod, cfg)
        ora = FusionNetOracle(**cfg)
        synth.fill_state_dict_([ref.encoder, ref.decoder], seed)
        synth.fill_state_dict_([ora.encoder, ora.decoder], seed)
        assert list(ref.encoder.state_dict().keys()) == list(ora.encoder.state_dict().keys())
        assert list(ref.decoder.state_dict().keys()) == list(ora.decoder.state_dict().keys())
        return ref, ora

    # ---------------- T0: tiny, train mode ----------------
    ref, ora = pair(synth.TINY, 11)
    batch = synth.make_batch(2, 70, 102, 8, seed=101)
    r = one_step(ref, batch, True)
    o = one_step(ora, batch, False)
    compare_step('T0', r, o)
    unused = sorted(k for k, g in r[2].items() if g is None)
    np.savez_compressed(
        os.path.join(gold, 'T0_tiny_train.npz'),
        meta=np.array([2, 70, 102, 8, 101, 11]),   # n, h, w, n_point, data seed, weight seed
        output=r[0].numpy(), loss=np.array(r[1], np.float64),
        unused=np.array(unused),
        **{'grad:' + k: g.numpy() for k, g in r[2].items() if g is not None},
        **{'buf:' + k: b.numpy() for k, b in r[3].items()})

    # ---------------- T3a: tiny, eval mode (fresh weights, no running-stat drift) ----------------
    ref, ora = pair(synth.TINY, 11)
    ref.eval(); ora.eval()
    with torch.no_grad():
        ro = ref.forward(batch['image'], batch['input_depth'])
        oo = ora.forward(batch['image'], batch['input_depth'])
    print('[T3 tiny eval] oracle vs reference: %.2e' % relerr(oo, ro))
    assert relerr(oo, ro) < 2e-5
    t3 = {'tiny_output': ro.numpy(), 'tiny_meta': np.array([2, 70, 102, 8, 101, 11])}

    # ---------------- T2: tiny, 3 Adam steps ----------------
    ref, ora = pair(synth.TINY, 12)
    traj = {}
    for tag, model, is_ref in (('ref', ref, True), ('ora', ora, False)):
        opt = torch.optim.Adam([{'params': model.parameters(), 'weight_decay': 0.0}], lr=1e-3)
        losses = []
        model.train()
        for step in range(3):
            b = synth.make_batch(2, 70, 102, 8, seed=200 + step)
            out = model.forward(b['image'], b['input_depth'])
            if is_ref:
                loss = ref_loss(model, b, out)[0]
            else:
                loss = model.compute_loss(out, b['ground_truth'], b['lidar_map'], 2.0)[0]
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss))
        psum = float(sum(p.detach().double().abs().sum() for _, p in named_params(model)))
        rm = float(sum(b.detach().double().abs().sum() for _, b in named_buffers(model)))
        traj[tag] = (losses, psum, rm)
    print('[T2] ref', traj['ref'], '\n[T2] ora', traj['ora'])
    for a, b in zip(traj['ora'][0], traj['ref'][0]):
        assert abs(a - b) / abs(b) < 1e-4
    assert abs(traj['ora'][1] - traj['ref'][1]) / traj['ref'][1] < 1e-5
    np.savez_compressed(
        os.path.join(gold, 'T2_tiny_adam3.npz'),
        meta=np.array([2, 70, 102, 8, 200, 12]),
        losses=np.array(traj['ref'][0], np.float64),
        param_abs_sum=np.array(traj['ref'][1]), buffer_abs_sum=np.array(traj['ref'][2]))

    # ---------------- T1: published net, config #1 ----------------
    ref, ora = pair(synth.PUBLISHED, 21)
    batch = synth.make_batch(1, 224, 384, 32, seed=301)
    r = one_step(ref, batch, True)
    o = one_step(ora, batch, False)
    compare_step('T1', r, o)
    keys = [k for k, g in r[2].items() if g is not None]
    np.savez_compressed(
        os.path.join(gold, 'T1_published_train.npz'),
        meta=np.array([1, 224, 384, 32, 301, 21]),
        output=r[0].numpy(), loss=np.array(r[1], np.float64),
        unused=np.array(sorted(k for k, g in r[2].items() if g is None)),
        grad_keys=np.array(keys),
        grad_l2=np.array([float(r[2][k].double().norm()) for k in keys]),
        grad_sum=np.array([float(r[2][k].double().sum()) for k in keys]),
        buf_keys=np.array(list(r[3].keys())),
        buf_l2=np.array([float(b.double().norm()) for b in r[3].values()]))

    # ---------------- T1b: per-element gradient samples of T1's ten largest gradient tensors ----------------
    # (the tensors are 2.4 - 4.7 MB each: 2048 seeded flat indices per tensor pin them element by element; next to the reference's
    # fp32 values, the same elements from the oracle run in fp64 -- the oracle equals the reference to 0.00e+00 in fp32 above -- so
    # that a test can tell an fp32 implementation's LeakyReLU / max-pool decision flips from errors)
    big = sorted(keys, key=lambda k: -r[2][k].numel())[:10]
    rs = np.random.RandomState(7)
    idx = np.stack([rs.choice(r[2][k].numel(), 2048, replace=False) for k in big]).astype(np.int64)
    ora64 = FusionNetOracle(**synth.PUBLISHED)
    synth.fill_state_dict_([ora64.encoder, ora64.decoder], 21)
    for mod in (ora64.encoder, ora64.decoder):
        mod.double()
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    o64 = one_step(ora64, b64, False)
    np.savez_compressed(
        os.path.join(gold, 'T1b_published_grad_samples.npz'),
        meta=np.array([1, 224, 384, 32, 301, 21]),
        keys=np.array(big), idx=idx,
        ref32=np.stack([r[2][k].reshape(-1).numpy()[i] for k, i in zip(big, idx)]).astype(np.float32),
        fp64=np.stack([o64[2][k].reshape(-1).numpy()[i] for k, i in zip(big, idx)]).astype(np.float64),
        fp64_absmax=np.array([float(o64[2][k].abs().max()) for k in big]),
        ref32_rel_err=np.array([relerr(r[2][k], o64[2][k]) for k in big]))
    print('[T1b] ten largest gradient tensors, reference fp32 vs fp64 (max-norm): '
          + ', '.join('%.1e' % relerr(r[2][k], o64[2][k]) for k in big))

    ref, ora = pair(synth.PUBLISHED, 21)
    ref.eval(); ora.eval()
    with torch.no_grad():
        ro = ref.forward(batch['image'], batch['input_depth'])
        oo = ora.forward(batch['image'], batch['input_depth'])
    print('[T3 published eval] oracle vs reference: %.2e' % relerr(oo, ro))
    assert relerr(oo, ro) < 2e-5
    t3['published_output'] = ro.numpy()
    t3['published_meta'] = np.array([1, 224, 384, 32, 301, 21])
    np.savez_compressed(os.path.join(gold, 'T3_eval.npz'), **t3)

    # ---------------- OutlierRemoval: oracle restatement == the reference class (src/net_utils.py:575-638) ----------------
    import net_utils as ref_net_utils
    from oracle.fusionnet_oracle import remove_outliers
    for seed, (hh, ww), ks, thr in ((1, (37, 53), 7, 1.5), (2, (64, 96), 7, 1.5), (3, (20, 31), 5, 0.5)):
        gt = synth.make_batch(2, hh, ww, 4, seed=seed)['ground_truth']
        a = ref_net_utils.OutlierRemoval(ks, thr).remove_outliers(gt)
        b = remove_outliers(gt, ks, thr)
        assert torch.equal(a, b), 'OutlierRemoval oracle differs from the reference'
        assert int((a != gt).sum()) > 0, 'test input must contain outliers'
    print('[OutlierRemoval] oracle == reference (bit-exact, 3 cases)')

    n_par = sum(p.numel() for _, p in named_params(ref))
    n_used = sum(p.numel() for k, p in named_params(ref) if k in keys)
    print('published net: %d parameters, %d receive gradients, %d unused tensors'
          % (n_par, n_used, len(named_params(ref)) - len(keys)))
    for f in sorted(os.listdir(gold)):
        if f.endswith('.npz'):
            print('%-28s %8.1f KB' % (f, os.path.getsize(os.path.join(gold, f)) / 1024.0))


if __name__ == '__main__':
    main()
