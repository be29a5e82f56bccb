'''
Fixture T13: encoder_type 'fusionnet34' (ResNet-34 block counts 3, 4, 6, 3 per level, src/networks.py:305-311) from the REAL reference,
tiny channels, one training step in train mode: output, loss terms, every parameter gradient's L2 norm -- and the assertion that
oracle/fusionnet_oracle.py (n_layer=34) reproduces the reference (0.00e+00).  Writes tests/golden/T13_fusionnet34_tiny_train.npz.
'''
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from make_golden import import_reference, named_params, one_step, compare_step   # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from rcf_amd import synth
    from oracle.fusionnet_oracle import FusionNetOracle
    ref_mod = import_reference()
    cfg = synth.TINY
    ref = ref_mod.FusionNetModel(
        input_channels_image=cfg['input_channels_image'], input_channels_depth=cfg['input_channels_depth'],
        encoder_type=['fusionnet34', 'batch_norm'], n_filters_encoder_image=cfg['n_filters_encoder_image'],
        n_filters_encoder_depth=cfg['n_filters_encoder_depth'], fusion_type='weight_and_project', decoder_type=['multiscale', 'batch_norm'],
        n_resolution_decoder=1, n_filters_decoder=cfg['n_filters_decoder'], deconv_type='up', activation_func='leaky_relu',
        weight_initializer='kaiming_uniform', min_predict_depth=1.0, max_predict_depth=100.0, device=torch.device('cpu'))
    ora = FusionNetOracle(n_layer=34, **cfg)
    synth.fill_state_dict_([ref.encoder, ref.decoder], 15)
    synth.fill_state_dict_([ora.encoder, ora.decoder], 15)
    assert list(ref.encoder.state_dict().keys()) == list(ora.encoder.state_dict().keys())
    batch = synth.make_batch(2, 70, 102, 8, seed=141)
    r = one_step(ref, batch, True)
    o = one_step(ora, batch, False)
    compare_step('T13', r, o)
    keys = [k for k, g in r[2].items() if g is not None]
    np.savez_compressed(os.path.join(HERE, 'T13_fusionnet34_tiny_train.npz'), meta=np.array([2, 70, 102, 8, 141, 15]),
                        output=r[0].numpy(), loss=np.array(r[1], np.float64), grad_keys=np.array(keys),
                        grad_l2=np.array([float(r[2][k].double().norm()) for k in keys]),
                        n_params=np.array(sum(p.numel() for _, p in named_params(ref))))
    print('wrote T13_fusionnet34_tiny_train.npz: %d parameters, %d gradient tensors' % (sum(p.numel() for _, p in named_params(ref)), len(keys)))


if __name__ == '__main__':
    main()
