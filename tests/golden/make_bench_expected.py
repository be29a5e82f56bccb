'''
Recorded loss of bench.py's FIRST training step (before any weight update), computed by the CPU oracle on exactly the inputs
and weights bench.py uses: published FusionNet, synth.fill_state_dict_(seed 1234), synth.make_batch(8, 900, 1600, 64, seed 1234),
train-mode BatchNorm, ground-truth outlier removal (7, 1.5), masked L1 + 2.0 x lidar term.  bench.py compares its own first-step
loss with this value (1e-3 relative in fp32) and exits non-zero on a mismatch, so the throughput it prints always belongs to a
step that computed the right thing at the full batch-8 900x1600 workload.

    python tests/golden/make_bench_expected.py            # ~20 minutes on 8 cores (8 data seeds); writes tests/golden/bench_expected.json
    python tests/golden/make_bench_expected.py --legs     # round 5: the values the default run's other legs check themselves against --
                                                          # eval-mode output of two samples of the batch-32 inference leg (configs[4]) and
                                                          # the RadarNet leg's first-step loss (configs[2]); ~5 minutes
'''
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch   # noqa: E402

import rcf_amd   # noqa: E402,F401
from rcf_amd import synth   # noqa: E402
from oracle.fusionnet_oracle import FusionNetOracle, remove_outliers   # noqa: E402


def first_step_loss(batch, height, width, points, wseed=1234, dseed=1234):
    '''-> (loss, sums): the oracle's loss of this rank-local batch, and the four numbers a data-parallel run needs to form the
    reference's ONE masked mean over the gathered batch (src/fusionnet_main.py:385, src/fusionnet_model.py:245-253):
    sum|out - gt| and count over gt > 0 (after outlier removal and the lidar zeroing of :214-221), sum|out - lidar| and count
    over lidar > 0.  BatchNorm is per replica under nn.DataParallel, so a per-seed run at the per-GPU batch is exactly one
    replica's forward.'''
    model = FusionNetOracle(**synth.PUBLISHED)
    synth.fill_state_dict_([model.encoder, model.decoder], wseed)
    b = synth.make_batch(batch, height, width, points, seed=dseed)
    model.train()
    with torch.no_grad():
        out = model.forward(b['image'], b['input_depth'])
        gt = remove_outliers(b['ground_truth'], 7, 1.5)
        lidar = b['lidar_map']
        loss = model.compute_loss(out, gt, lidar, 2.0)[0]
        gt = gt * torch.where(lidar > 0.0, torch.zeros_like(lidar), torch.ones_like(lidar))
        vg, vl = gt > 0, lidar > 0
        sums = {'sum_abs_gt': float((out[vg].double() - gt[vg].double()).abs().sum()), 'count_gt': int(vg.sum()),
                'sum_abs_lidar': float((out[vl].double() - lidar[vl].double()).abs().sum()), 'count_lidar': int(vl.sum())}
    return float(loss), sums


def infer_expected(batch=32, height=900, width=1600, points=64, seed=1234, samples=(0, 31)):
    '''bench.py's inference leg (eval-mode BatchNorm, weights seed 1234, synth.make_batch(32, ..., seed 1234)): in eval mode the samples
    are independent, so the oracle's output for samples 0 and 31 alone is what rows 0 and 31 of the batch-32 output must be.  Recorded:
    the mean depth of each, and 64 seeded pixels per sample.'''
    import numpy as np
    model = FusionNetOracle(**synth.PUBLISHED)
    synth.fill_state_dict_([model.encoder, model.decoder], 1234)
    model.eval()
    b = synth.make_batch(batch, height, width, points, seed=seed)
    idx = np.random.RandomState(7).randint(0, height * width, size=64)
    rec = {'pixel_index': [int(i) for i in idx], 'samples': {}}
    with torch.no_grad():
        for s in samples:
            out = model.forward(b['image'][s:s + 1], b['input_depth'][s:s + 1])
            flat = out.reshape(-1)
            rec['samples'][str(s)] = {'mean_depth': float(flat.double().mean()), 'pixels': [float(flat[int(i)]) for i in idx]}
    return rec


def radarnet_expected(n_img=16, k=4, height=900, width=1888):
    '''bench.py's RadarNet leg (weights seed 41, synth.make_radarnet_batch(7, ...), train-mode BatchNorm, w_positive_class 2.0): the
    oracle's loss of the first step and the mean logit.'''
    from oracle.radarnet_oracle import RadarNetOracle
    ora = RadarNetOracle(**synth.RADARNET_PUBLISHED)
    synth.fill_state_dict_([ora.encoder, ora.decoder], 41)
    cb = synth.make_radarnet_batch(7, n=n_img, k=k, h=height, w=width, patch_w=288)
    ora.train()
    import numpy as np
    with torch.no_grad():
        ol = ora.forward(cb['image'], cb['point'], cb['bounding_boxes'])
        loss = float(ora.compute_loss(ol, cb['ground_truth'], cb['validity_map'], 2.0))
    # 4096 seeded logits of the (64, 1, 900, 288) map, for tests/test_configs_gpu.py (the full map is 66 MB)
    idx = np.random.RandomState(11).randint(0, ol.numel(), size=4096)
    flat = ol.reshape(-1)
    return {'first_step_loss': loss, 'mean_logit': float(ol.double().mean()), 'max_abs_logit': float(ol.abs().max()),
            'logit_index': [int(i) for i in idx], 'logits': [float(flat[int(i)]) for i in idx]}


if __name__ == '__main__' and '--legs' in sys.argv:
    path = os.path.join(ROOT, 'tests', 'golden', 'bench_expected.json')
    rec = json.load(open(path))
    t0 = time.time()
    rec['infer_b32_900x1600_p64'] = dict(infer_expected(), source='oracle/fusionnet_oracle.py, eval mode, samples 0 and 31 of the batch alone')
    print('infer', rec['infer_b32_900x1600_p64']['samples']['0']['mean_depth'], '%.1f s' % (time.time() - t0), flush=True)
    rec['radarnet_b16x4_900x1888'] = dict(radarnet_expected(), source='oracle/radarnet_oracle.py, train mode, first step')
    print('radarnet', rec['radarnet_b16x4_900x1888'], '%.1f s' % (time.time() - t0), flush=True)
    json.dump(rec, open(path, 'w'), indent=1, sort_keys=True)
    sys.exit(0)


if __name__ == '__main__':
    cases = {'train_b8_900x1600_p64': (8, 900, 1600, 64)}
    if '--small' in sys.argv:
        cases = {'train_b2_224x384_p32': (2, 224, 384, 32)}
    path = os.path.join(ROOT, 'tests', 'golden', 'bench_expected.json')
    rec = json.load(open(path)) if os.path.exists(path) else {}
    n_ranks = 8   # bench.py's rank r uses data seed 1234 + r
    for key, (n, h, w, k) in cases.items():
        t0 = time.time()
        per_seed = {}
        for r in range(n_ranks):
            loss, sums = first_step_loss(n, h, w, k, dseed=1234 + r)
            sums['loss'] = loss
            per_seed[str(1234 + r)] = sums
            print(key, 1234 + r, sums, '%.1f s' % (time.time() - t0), flush=True)
        rec[key] = {'first_step_loss': per_seed['1234']['loss'], 'weights_seed': 1234, 'data_seed': 1234, 'per_data_seed': per_seed,
                    'source': 'oracle/fusionnet_oracle.py (CPU fp32, pinned to the reference by make_golden.py)'}
        json.dump(rec, open(path, 'w'), indent=1, sort_keys=True)
