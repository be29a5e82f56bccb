'''
Fixture T10: the REAL reference with deconv_type='transpose' (net_utils.TransposeConv2d, src/net_utils.py:94-153, selected at
:507-513 / :550-551), which the shipped entry points never select (src/fusionnet_main.py:190 hard-codes 'up') and which only
works (a) with the default weight initializer and (b) at input sizes divisible by 64.  Run in the build
container only:

    python tests/golden/make_golden_transpose.py

  T10a  tiny net, train mode, 2x3x64x128: output, loss terms, every parameter gradient, BatchNorm buffers
  T10b  published net, train mode, 1x3x192x256: output, loss terms, gradient L2 norms   (448x448, the shipped training crop, is
        checked against the oracle on the GPU box: tests/test_hip_model.py)
and asserts oracle/fusionnet_oracle.py (deconv_type='transpose') reproduces the reference on both.
'''
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import make_golden as mg


def build_reference(ref_mod, cfg):
    return ref_mod.FusionNetModel(
        input_channels_image=cfg['input_channels_image'], input_channels_depth=cfg['input_channels_depth'],
        encoder_type=['fusionnet18', 'batch_norm'], n_filters_encoder_image=cfg['n_filters_encoder_image'],
        n_filters_encoder_depth=cfg['n_filters_encoder_depth'], fusion_type='weight_and_project',
        decoder_type=['multiscale', 'batch_norm'], n_resolution_decoder=1, n_filters_decoder=cfg['n_filters_decoder'],
        deconv_type='transpose', activation_func='leaky_relu', weight_initializer='kaiming_uniform',
        min_predict_depth=1.0, max_predict_depth=100.0, device=torch.device('cpu'))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from rcf_amd import synth
    from oracle.fusionnet_oracle import FusionNetOracle
    ref_mod = mg.import_reference()

    def pair(cfg, seed):
        ref = build_reference(ref_mod, cfg)
        ora = FusionNetOracle(deconv_type='transpose', **cfg)
        synth.fill_state_dict_([ref.encoder, ref.decoder], seed)
        synth.fill_state_dict_([ora.encoder, ora.decoder], seed)
        assert list(ref.decoder.state_dict().keys()) == list(ora.decoder.state_dict().keys())
        assert any(k.endswith('deconv.deconv.weight') for k in ref.decoder.state_dict().keys())
        return ref, ora

    ref, ora = pair(synth.TINY, 51)
    batch = synth.make_batch(2, 64, 128, 8, seed=501)
    r = mg.one_step(ref, batch, True)
    o = mg.one_step(ora, batch, False)
    mg.compare_step('T10a', r, o)
    np.savez_compressed(
        os.path.join(HERE, 'T10_transpose_tiny_train.npz'),
        meta=np.array([2, 64, 128, 8, 501, 51]), output=r[0].numpy(), loss=np.array(r[1], np.float64),
        unused=np.array(sorted(k for k, g in r[2].items() if g is None)),
        **{'grad:' + k: g.numpy() for k, g in r[2].items() if g is not None},
        **{'buf:' + k: b.numpy() for k, b in r[3].items()})

    ref, ora = pair(synth.PUBLISHED, 52)
    batch = synth.make_batch(1, 192, 256, 16, seed=502)
    r = mg.one_step(ref, batch, True)
    o = mg.one_step(ora, batch, False)
    mg.compare_step('T10b', r, o)
    keys = [k for k, g in r[2].items() if g is not None]
    np.savez_compressed(
        os.path.join(HERE, 'T10_transpose_published_train.npz'),
        meta=np.array([1, 192, 256, 16, 502, 52]), output=r[0].numpy(), loss=np.array(r[1], np.float64),
        grad_keys=np.array(keys), grad_l2=np.array([float(r[2][k].double().norm()) for k in keys]))
    # what the reference does at a size that is not divisible by 64: it cannot concatenate (recorded as a fact, not a fixture)
    try:
        ref.forward(torch.zeros(1, 3, 224, 384 + 32), torch.zeros(1, 2, 224, 384 + 32))
        print('reference accepted 224x416 ?!')
    except RuntimeError as e:
        print('reference at 224x416 (not divisible by 64):', str(e).split('\n')[0][:100])
    for f in sorted(os.listdir(HERE)):
        if f.startswith('T10'):
            print('%-40s %8.1f KB' % (f, os.path.getsize(os.path.join(HERE, f)) / 1024.0))


if __name__ == '__main__':
    main()
