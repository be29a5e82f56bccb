'''
bench.py -- FusionNet training throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): the published FusionNet (bash/train_fusionnet_nuscenes.sh:27-40), fp32,
per-GPU batch 8, 900x1600, 64-point synthetic radar maps; one step = forward + ground-truth outlier removal +
masked-L1 loss (w_lidar 2.0) + backward + Adam, training-mode BatchNorm -- the body of the reference's loop (src/fusionnet_main.py:369-399).
Inputs are resident in HBM before the timed region.  N > 1: one process per GPU, the same per-GPU batch
(weak scaling), gradients all-reduced over RCCL in buckets that overlap the backward pass.

Rank 0 prints ONE JSON line.  `roofline` is measured live: every launch of the dominant kernel (the 3x3
stride-1 implicit-GEMM convolution, forward + input-gradient launches) is bracketed by events on the launch
stream inside the timed steps; achieved = algorithmic FLOP / event time.  `cpu_baseline` (N = 1 only) times
the CPU oracle on the host cores on a bounded sample (one training step at batch 1, 900x1600).
'''

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 MFMA peak (the 5 PF headline includes 2:1 sparsity)
# what a loop of nothing but v_mfma_f32_32x32x16_bf16 on random operand bits sustains at the ~1.4 kW board limit
# (tools/probe/mfma_issue_probe.hip: 17.0 ns per MFMA per SIMD) -- reported next to `frac`, never instead of it
BF16_MFMA_SUSTAINED_TFLOPS = 1970.0
SPLIT_PRODUCTS = 6              # bf16 partial products executed per fp32 multiply-add in the split kernels
KERNEL_NAMES = {
    0: 'conv_fwd_kernel 3x3 s1', 1: 'conv_fwd_kernel 3x3 s2', 2: 'conv_fwd_kernel 1x1', 3: 'conv_fwd_kernel 7x7 s2 stem',
    5: 'conv_split_kernel 3x3 s1 (fp32 via bf16x3 split)', 9: 'conv_split_kernel 2x2 phases (fp32 via bf16x3 split)',
    10: 'conv_wgrad_kernel 3x3 s1', 11: 'conv_wgrad_kernel 3x3 s2', 12: 'conv_wgrad_kernel 1x1', 13: 'conv_wgrad_kernel 7x7 s2 stem',
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=8, help='per-GPU batch (BASELINE.json configs[1]: 8)')
    ap.add_argument('--height', type=int, default=900)
    ap.add_argument('--width', type=int, default=1600)
    ap.add_argument('--points', type=int, default=64)
    ap.add_argument('--dtype', choices=('f32', 'bf16'), default='f32',
                    help="arithmetic of the conv kernels: f32 (the metric's configuration) or bf16 operands with fp32 accumulate "
                         '(BASELINE.json configs 3-5; informational)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--kernel-table', type=str, default='', help='write the per-kernel event table (JSON) here')
    return ap.parse_args()


def cpu_baseline(height, width, points):
    '''The CPU oracle (stock PyTorch fp32, oneDNN) on the host cores: one training step at batch 1.'''
    import torch
    from rcf_amd import synth
    from oracle.fusionnet_oracle import FusionNetOracle
    # torch's default intra-op thread count (= physical cores): setting it to the logical-CPU count
    # (256 on the 2 x EPYC 9575F box) oversubscribes oneDNN and is ~25x slower
    model = FusionNetOracle(**synth.PUBLISHED)
    synth.fill_state_dict_([model.encoder, model.decoder], 1234)
    opt = torch.optim.Adam([{'params': model.parameters(), 'weight_decay': 0.0}], lr=1e-3)
    b = synth.make_batch(1, height, width, points, seed=99)
    model.train()
    t0 = time.time()
    out = model.forward(b['image'], b['input_depth'])
    loss = model.compute_loss(out, b['ground_truth'], b['lidar_map'], 2.0)[0]
    opt.zero_grad()
    loss.backward()
    opt.step()
    dt = time.time() - t0
    return {'value': round(1.0 / dt, 5), 'unit': 'samples/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '1 training step (fwd+loss+bwd+Adam), batch 1, %dx%d, published net, %.1f s' % (height, width, dt)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import rcf_amd  # noqa: F401
    from rcf_amd import ops, parallel, synth, train

    rank, world, local_rank = parallel.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the hot path is HIP-only')
    # RCF_BENCH_SINGLE_DEVICE=1 (tests on a 1-GPU box, gloo backend): every rank shares cuda:0
    dev = torch.device('cuda', local_rank if (world > 1 and not os.environ.get('RCF_BENCH_SINGLE_DEVICE')) else 0)
    torch.cuda.set_device(dev)

    torch.manual_seed(1234)                       # identical initial weights on every rank
    model = train.build_model(synth.PUBLISHED, device=dev)
    model.compute_dtype = 'bf16' if args.dtype == 'bf16' else 'fp32'
    if world > 1:
        model.data_parallel()
    opt = train.make_optimizer(model, lr=1e-3)
    model.train()
    b = synth.make_batch(args.batch, args.height, args.width, args.points, seed=1234 + rank)
    image, input_depth = b['image'].to(dev), b['input_depth'].to(dev)
    gt, lidar = b['ground_truth'].to(dev), b['lidar_map'].to(dev)

    from rcf_amd.net_utils import OutlierRemoval
    outlier = OutlierRemoval(kernel_size=7, threshold=1.5)   # bash/train_fusionnet_nuscenes.sh:48-49

    def step():
        return train.train_step(model, opt, image, input_depth, gt, lidar, outlier_removal=outlier)[0]

    for _ in range(args.warmup):
        step()
    timer = ops.KernelTimer()
    model._engine.prof = timer

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    model._engine.prof = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.detach())

    table = timer.collect()
    # aggregate kernel ids (kind*1000 + ck*10 + nt [+100 for the 16x16 tile]; wgrad ids are 10000 + ...) by kernel family
    fam = {}
    for kid, (cnt, flops, ms) in table.items():
        kid0 = kid - 20000 if kid >= 20000 else kid        # + 20000: bf16-operand variant of the same kernel family
        f = (10 + (kid0 - 10000) // 1000) if kid0 >= 10000 else kid0 // 1000
        r = fam.setdefault(f, [0, 0.0, 0.0, 0.0])
        r[0] += cnt; r[1] += flops; r[2] += ms; r[3] += getattr(timer, 'bytes', {}).get(kid, 0.0)
    dom = max(fam, key=lambda f: fam[f][2]) if fam else None

    if rank == 0:
        n_samples = world * args.batch * args.steps
        rec = {
            'metric': 'FusionNet train samples/sec at 900x1600',
            'value': round(n_samples / dt, 4),
            'unit': 'samples/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(1000.0 * dt / args.steps, 3),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': args.dtype,
            'data': 'synthetic',
            'config': {'workload': 'FusionNet %s training, per-GPU batch %d, %dx%d, %d-point radar maps (BASELINE.json %s)'
                                   % ('fp32' if args.dtype == 'f32' else 'bf16-operand', args.batch, args.height, args.width, args.points,
                                      'configs[1]' if args.dtype == 'f32' else 'configs[3] arithmetic on one GPU'),
                       'global_batch': world * args.batch, 'parallelism': 'dp%d' % world,
                       'step': 'forward + outlier removal + masked L1 + backward + Adam, train-mode BatchNorm', 'final_loss': round(final_loss, 5)},
        }
        if dom is not None:
            cnt, flops, ms, abytes = fam[dom]
            algorithmic = flops / (ms * 1e-3) / 1e12
            is_split = dom in (5, 9, 15, 19)
            # split kernels are bound by the bf16 matrix pipe: price them on the bf16 FLOPs they execute (6 per fp32 MAC)
            achieved = algorithmic * ((1 if args.dtype == 'bf16' else SPLIT_PRODUCTS) if is_split else 1)
            peak = BF16_MFMA_PEAK_TFLOPS if is_split else F32_MFMA_PEAK_TFLOPS
            traffic, traffic_src = None, None
            pmc_path = os.path.join(ROOT, 'profiles', 'r01_pmc_bench.json')
            if os.path.exists(pmc_path):   # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command
                try:
                    pmc = json.load(open(pmc_path)).get(KERNEL_NAMES.get(dom, ''), {})
                    traffic = round(pmc['hbm_bytes_per_launch'] / 1e9, 4)
                    traffic_src = 'profiles/r01_pmc_bench.json (GB per launch, FETCH_SIZE x2 + WRITE_SIZE)'
                except Exception:
                    traffic = None
            conv_ms = sum(r[2] for r in fam.values())
            conv_flops = sum(r[1] for r in fam.values())
            rec['roofline'] = {
                'bound': 'mfma', 'kernel': KERNEL_NAMES.get(dom, str(dom)),
                'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
                'frac': round(achieved / peak, 4), 'algorithmic_fp32_tflops': round(algorithmic, 2),
                'frac_of_power_limited_peak': round(achieved / BF16_MFMA_SUSTAINED_TFLOPS, 4) if is_split else None,
                'pipe': ('bf16 MFMA, bf16 operands, fp32 accumulate' if args.dtype == 'bf16' else 'bf16 MFMA, 6 exact partial products per fp32 multiply, fp32 accumulate') if is_split else 'f32 MFMA', 'traffic': traffic, 'traffic_source': traffic_src,
                'algorithmic_gbytes_per_launch': round(abytes / cnt / 1e9, 4),
                'launches_per_step': cnt // args.steps, 'avg_launch_ms': round(ms / cnt, 4),
                'algorithmic_gflop_per_launch': round(flops / cnt / 1e9, 3),
                'share_of_step_time': round(ms / (1000.0 * dt), 4),
                'all_conv_kernels': {'achieved': round(conv_flops / (conv_ms * 1e-3) / 1e12, 2),
                                     'share_of_step_time': round(conv_ms / (1000.0 * dt), 4),
                                     'gflop_per_sample': round(conv_flops / (args.batch * args.steps) / 1e9, 2)},
            }
        if args.kernel_table:
            rows = [{'kernel_id': kid, 'launches': c, 'gflop': f / 1e9, 'ms': m, 'tflops': f / (m * 1e-3) / 1e12 if m > 0 else 0}
                    for kid, (c, f, m) in sorted(table.items())]
            layers = [{'kernel_id': k[0], 'layer': k[1], 'launches': c, 'gflop': f / 1e9, 'ms': m,
                       'tflops': f / (m * 1e-3) / 1e12 if m > 0 else 0} for k, (c, f, m) in sorted(timer.layers.items(), key=lambda kv: -kv[1][2])]
            with open(args.kernel_table, 'w') as fh:
                json.dump({'steps': args.steps, 'ms_per_step': 1000.0 * dt / args.steps, 'kernels': rows, 'layers': layers}, fh, indent=1)
        if world == 1 and not args.no_cpu_baseline:
            rec['cpu_baseline'] = cpu_baseline(args.height, args.width, args.points)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
