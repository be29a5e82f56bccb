'''
Recorded loss of bench.py's FIRST training step (before any weight update), computed by the CPU oracle on exactly the inputs
and weights bench.py uses: published FusionNet, synth.fill_state_dict_(seed 1234), synth.make_batch(8, 900, 1600, 64, seed 1234),
train-mode BatchNorm, ground-truth outlier removal (7, 1.5), masked L1 + 2.0 x lidar term.  bench.py compares its own first-step
loss with this value (1e-3 relative in fp32) and exits non-zero on a mismatch, so the throughput it prints always belongs to a
step that computed the right thing at the full batch-8 900x1600 workload.

    python tests/golden/make_bench_expected.py            # ~20 minutes on 8 cores (8 data seeds); writes tests/golden/bench_expected.json
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
