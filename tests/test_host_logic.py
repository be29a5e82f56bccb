'''
CPU tests (no GPU): the C-ABI library loads and exports every symbol include/rcf_hip.h declares, the host mirror of
the reference's API (names, state_dict keys, error behaviour), the optimizer's state format, and the data-parallel
gradient buckets over a 2-rank gloo group.  No HIP compute is invoked here.
'''

import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def pkg():
    import __graft_entry__ as entry
    entry.build()
    import rcf_amd
    return rcf_amd


def test_library_exports_every_declared_symbol(pkg):
    from rcf_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'rcf_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    declared = set(re.findall(r'\b(rcf_[a-z0-9_]+)\s*\(', header))
    assert len(declared) >= 30
    assert declared == set(_lib._SIGNATURES.keys()), declared ^ set(_lib._SIGNATURES.keys())
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert b'gfx950' in lib.rcf_version()


def test_product_path_has_no_cpu_fallback(pkg):
    from rcf_amd import _lib, ops, synth, train
    m = train.build_model(synth.TINY, device='cpu')
    with pytest.raises(_lib.RcfError):
        m.forward(torch.zeros(1, 3, 64, 96), torch.zeros(1, 2, 64, 96))
    with pytest.raises(_lib.RcfError):
        ops.nchw_to_nhwc(torch.zeros(1, 3, 4, 4))
    with pytest.raises(RuntimeError):
        m.encoder.conv1_image(torch.zeros(1, 3, 8, 8))      # blocks are parameter containers
    from rcf_amd import radarnet_model
    r = radarnet_model.RadarNetModel(device='cpu', **synth.RADARNET_TINY)
    rb = synth.make_radarnet_batch(1)
    with pytest.raises(_lib.RcfError):
        r.forward(rb['image'], rb['point'], rb['bounding_boxes'])
    from rcf_amd.fusionnet_transforms import Transforms
    with pytest.raises(_lib.RcfError):
        Transforms(normalized_image_range=[0, 1]).transform([torch.zeros(2, 3, 8, 8)])
    # nothing under the package imports the oracle
    for root, _, files in os.walk(os.path.join(ROOT, 'radar-camera-fusion-depth_amd')):
        for f in files:
            if f.endswith('.py'):
                assert 'oracle' not in open(os.path.join(root, f)).read().replace('the oracle', '').replace('CPU oracle', ''), f


def test_state_dict_names_and_shapes_match_reference(pkg):
    from rcf_amd import synth, train
    from oracle.fusionnet_oracle import FusionNetOracle
    for cfg in (synth.TINY, synth.PUBLISHED):
        m = train.build_model(cfg, device='cpu')
        o = FusionNetOracle(**cfg)       # key-for-key identical to the reference (tests/golden/make_golden.py asserts it)
        for mine, ref in ((m.encoder, o.encoder), (m.decoder, o.decoder)):
            a, b = mine.state_dict(), ref.state_dict()
            assert list(a.keys()) == list(b.keys())
            for k in a:
                assert tuple(a[k].shape) == tuple(b[k].shape), k
    m = train.build_model(synth.PUBLISHED, device='cpu')
    assert sum(p.numel() for p in m.parameters()) == 14413568
    assert m._n_used == 14142208
    assert len(m.parameters()) - len(m._used_params) == 10     # unused projections
    unused = [k for k, p in list(m.encoder.named_parameters()) if id(p) not in set(id(q) for q in m._used_params)]
    assert all(k.endswith('.1.projection.conv.weight') for k in unused)


def test_parameters_are_views_of_one_arena_in_backward_order(pkg):
    from rcf_amd import synth, train
    m = train.build_model(synth.TINY, device='cpu')
    base = m._param_arena.data_ptr()
    off = 0
    for p in m._used_params:
        assert p.data_ptr() == base + 4 * off
        assert m._grad_of(p).data_ptr() == m._grad_arena.data_ptr() + 4 * off
        off += p.numel()
    assert off == m._n_used
    assert m._used_params[0] is m.decoder.output0.conv.weight          # first gradient to become final
    assert m._used_params[-1] is m.encoder.conv1_image.conv.weight     # last
    # load_state_dict writes through the views
    sd = m.encoder.state_dict()
    sd['conv1_image.conv.weight'] = torch.full_like(sd['conv1_image.conv.weight'], 0.5)
    m.encoder.load_state_dict(sd)
    assert float(m._param_arena[m._param_offset[id(m.encoder.conv1_image.conv.weight)]]) == 0.5


def test_error_behaviour_mirrors_reference(pkg):
    from rcf_amd.fusionnet_model import FusionNetModel
    from rcf_amd import net_utils
    base = dict(input_channels_image=3, input_channels_depth=2, encoder_type=['fusionnet18', 'batch_norm'],
                n_filters_encoder_image=[8, 16, 32, 32, 32, 32], n_filters_encoder_depth=[4, 8, 16, 16, 16, 16],
                fusion_type='weight_and_project', decoder_type=['multiscale', 'batch_norm'], n_resolution_decoder=1,
                n_filters_decoder=[32, 32, 16, 8, 8, 4], deconv_type='up', activation_func='leaky_relu',
                weight_initializer='kaiming_uniform', min_predict_depth=1.0, max_predict_depth=100.0, device='cpu')
    for key, bad in (('fusion_type', 'bogus'), ('encoder_type', ['vgg11']), ('decoder_type', ['unet']),
                     ('activation_func', 'swish'), ('deconv_type', 'bilinear')):
        kw = dict(base); kw[key] = bad
        with pytest.raises(ValueError):           # src/fusionnet_model.py:82, :90, :115, :135; src/net_utils.py:23
            FusionNetModel(**kw)
    with pytest.raises(ValueError):
        net_utils.activation_func('tanh')
    m = FusionNetModel(**base)
    t = FusionNetModel(**dict(base, deconv_type='transpose'))      # src/net_utils.py:507-513: reference state-dict names
    assert tuple(t.decoder.state_dict()['deconv5.deconv.deconv.weight'].shape) == (32, 32, 3, 3)
    assert 'deconv5.deconv.batch_norm.running_var' in t.decoder.state_dict()
    with pytest.raises(ValueError):               # src/fusionnet_model.py:275
        m.compute_loss(None, torch.zeros(1, 1, 4, 4), torch.zeros(1, 1, 4, 4), torch.zeros(1, 1, 4, 4), 'huber', 0.0, -1, None, 2.0)


def test_fused_adam_state_dict_is_torch_adam_compatible(pkg):
    from rcf_amd import synth, train
    m = train.build_model(synth.TINY, device='cpu')
    fused = train.make_optimizer(m, lr=1e-3, weight_decay=0.0)
    ref = torch.optim.Adam([{'params': m.parameters(), 'weight_decay': 0.0}], lr=1e-3)
    a, b = fused.state_dict(), ref.state_dict()
    assert a['param_groups'][0].keys() == b['param_groups'][0].keys()
    assert a['param_groups'][0]['params'] == b['param_groups'][0]['params']
    # a torch.optim.Adam state (after one CPU step) loads into FusedAdam and lands in the flat moment arenas
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    ref.step()
    fused.load_state_dict(ref.state_dict())
    p0 = m.decoder.output0.conv.weight
    st = fused.state[p0]
    assert float(st['step']) == 1.0
    marena = fused._moment_arenas[id(m._param_arena)][0]
    assert st['exp_avg'].data_ptr() == marena.data_ptr() + 4 * m._param_offset[id(p0)]
    assert torch.allclose(st['exp_avg'], ref.state[p0]['exp_avg'])


def test_fused_adam_counts_one_step_per_step_call_with_several_param_groups(pkg, monkeypatch):
    '''Two param groups (different lr) on ONE arena: torch.optim.Adam semantics are one step count per optimizer.step(), shared by
    every parameter; each group's launch carries its own device counter and hyper-parameters.  The kernel launch is replaced by a
    host emulation of rcf_adam_step_dev's bookkeeping (tick the counter, read lr) -- no GPU here.'''
    from rcf_amd import ops, synth, train
    from rcf_amd.optim import FusedAdam
    m = train.build_model(synth.TINY, device='cpu')
    ps = m._used_params
    half = len(ps) // 2
    opt = FusedAdam([{'params': ps[:half], 'lr': 1e-3}, {'params': ps[half:], 'lr': 5e-4}])
    seen = []

    def fake_adam_step_dev(p, g, m_, v, state):
        state[0] += 1.0
        seen.append((p.numel(), float(state[0]), float(state[1])))
    monkeypatch.setattr(ops, 'adam_step_dev', fake_adam_step_dev)
    for p in ps:
        p.grad = m._grad_views[id(p)]
    n0, n1 = sum(p.numel() for p in ps[:half]), sum(p.numel() for p in ps[half:])
    for k in (1, 2, 3):
        del seen[:]
        opt.step()
        assert seen == [(n0, float(k), pytest.approx(1e-3)), (n1, float(k), pytest.approx(5e-4))], seen
        assert float(opt.state[ps[0]]['step']) == float(opt.state[ps[-1]]['step']) == float(k)
    assert all(float(st['step']) == 3.0 for st in opt.state_dict()['state'].values())
    # a learning-rate schedule reaches the device copy of the right group, also on the replay path
    opt.param_groups[1]['lr'] = 2.5e-4
    opt.sync_hyper_parameters()
    lrs = sorted(float(ds[0][1]) for ds in opt._dev_state.values())
    assert lrs == [pytest.approx(2.5e-4), pytest.approx(1e-3)]
    # resume: load_state_dict replaces the param-group dictionaries; a schedule that writes into the NEW ones still reaches the device
    # copies, and the device step counters follow the loaded state
    sd = opt.state_dict()
    for st in sd['state'].values():
        st['step'] = torch.tensor(7.0)
    old_groups = list(opt.param_groups)
    opt.load_state_dict(sd)
    assert all(g is not o for g, o in zip(opt.param_groups, old_groups))
    opt.param_groups[0]['lr'] = 7.5e-4
    opt.sync_hyper_parameters()
    assert sorted(float(ds[0][1]) for ds in opt._dev_state.values()) == [pytest.approx(2.5e-4), pytest.approx(7.5e-4)]
    assert all(float(ds[0][0]) == 7.0 and ds[2] == 7 for ds in opt._dev_state.values())
    del seen[:]
    opt.step()
    assert [s[1] for s in seen] == [8.0, 8.0] and [s[2] for s in seen] == [pytest.approx(7.5e-4), pytest.approx(2.5e-4)], seen


def test_bench_expected_loss_under_data_parallelism_is_the_global_masked_mean(pkg):
    '''bench.py --gpus N checks its first-step loss against the oracle's ONE masked mean over all ranks' batches (rank r = data seed
    1234 + r), formed from per-seed sums and counts -- not against rank 0's own mean.  Here: the per-seed records exist for 8 ranks
    at both shapes, world 1 reproduces the single-rank loss, and the 2-rank value of the small case equals the oracle run on the
    concatenated sums computed afresh.'''
    import json
    sys.path.insert(0, ROOT)
    import bench
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import make_bench_expected as mk
    rec = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'bench_expected.json')))
    for key in ('train_b8_900x1600_p64', 'train_b2_224x384_p32'):
        per = rec[key]['per_data_seed']
        assert sorted(per) == [str(1234 + r) for r in range(8)]
        one = per['1234']
        assert abs(one['sum_abs_gt'] / one['count_gt'] + 2.0 * one['sum_abs_lidar'] / one['count_lidar'] - rec[key]['first_step_loss']) \
            < 2e-6 * rec[key]['first_step_loss']
        assert bench._expected_first_loss(key, 1) == rec[key]['first_step_loss']
        for world in (2, 4, 8):
            w = bench._expected_first_loss(key, world)
            lo, hi = min(per[str(1234 + r)]['loss'] for r in range(world)), max(per[str(1234 + r)]['loss'] for r in range(world))
            assert lo <= w <= hi
    sums = [mk.first_step_loss(2, 224, 384, 32, dseed=1234 + r)[1] for r in range(2)]
    want = (sums[0]['sum_abs_gt'] + sums[1]['sum_abs_gt']) / (sums[0]['count_gt'] + sums[1]['count_gt']) \
        + 2.0 * (sums[0]['sum_abs_lidar'] + sums[1]['sum_abs_lidar']) / (sums[0]['count_lidar'] + sums[1]['count_lidar'])
    assert abs(bench._expected_first_loss('train_b2_224x384_p32', 2) - want) < 1e-6 * want


def test_synthetic_inputs_are_deterministic(pkg):
    from rcf_amd import synth
    a = synth.make_batch(2, 64, 96, 8, seed=5)
    b = synth.make_batch(2, 64, 96, 8, seed=5)
    for k in a:
        assert torch.equal(a[k], b[k])
    assert 0.2 < float((a['ground_truth'] > 0).float().mean()) < 0.4
    assert float(a['input_depth'][:, 1].max()) <= 64.0 and float(a['input_depth'][:, 1][a['input_depth'][:, 1] > 0].min()) >= 32.0


def _dp_worker(rank, world, port, tmpdir):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import rcf_amd  # noqa: F401
    from rcf_amd import synth, train
    from rcf_amd.parallel import GradientBuckets
    m = train.build_model(synth.TINY, device='cpu')
    m.data_parallel()
    assert isinstance(m._dp, GradientBuckets) and len(m._dp.bounds) >= 2
    # simulate a backward: every rank fills its gradient arena with rank-dependent values, then the tape
    # reports parameters in completion order
    g = torch.Generator().manual_seed(100 + rank)
    local = torch.rand(m._grad_arena.numel(), generator=g)
    m._grad_arena.copy_(local)
    m._dp.begin_backward()
    launched_before_end = 0
    for i, p in enumerate(m._used_params):
        m._dp.on_param_grad(p)
        if i < len(m._used_params) - 1:
            launched_before_end = len(m._dp.handles)
    m._dp.finish_backward()
    sums = torch.tensor([1.0 + rank, 10.0, 2.0, 5.0 + rank], dtype=torch.float64)
    m._all_reduce_loss_sums(sums)
    torch.save({'local': local, 'reduced': m._grad_arena.clone(), 'n_used': m._n_used, 'sums': sums,
                'early': launched_before_end, 'n_buckets': len(m._dp.bounds)}, os.path.join(tmpdir, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)   # a rank that never joins must not hold the suite
@pytest.mark.parametrize('world', [2, 8])
def test_gradient_buckets_all_reduce_gloo(pkg, tmp_path, world):
    '''The exchange logic of the data-parallel step at 2 ranks and at the 8 ranks of BASELINE.json configs[3] (one process per rank over
    gloo here): SUM over ranks of every bucket with no 1/world factor, unused parameters never exchanged, the four loss sums / valid
    counts global, buckets launched before the backward ends.'''
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_dp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(os.path.join(str(tmp_path), 'r%d.pt' % k)) for k in range(world)]
    n = rs[0]['n_used']
    want = sum(r['local'].double() for r in rs).float()
    for r in rs:
        assert torch.allclose(r['reduced'][:n], want[:n], rtol=1e-6, atol=1e-6)    # SUM over ranks, no 1/world factor
        assert torch.equal(r['reduced'][n:], r['local'][n:])                  # unused parameters are never reduced
        assert r['sums'].tolist() == [world + world * (world - 1) / 2.0, 10.0 * world, 2.0 * world, 5.0 * world + world * (world - 1) / 2.0]
        assert r['early'] >= r['n_buckets'] - 1                               # buckets launch before backward ends
    assert all(torch.equal(r['reduced'][:n], rs[0]['reduced'][:n]) for r in rs[1:])     # every replica holds the same bits afterwards


def test_integration_md_struct_stubs_match_the_binding(pkg):
    '''The ctypes stub INTEGRATION.md shows a reference maintainer must describe the same rcf_conv_desc / rcf_conv_info as _lib.py
    (a shorter rcf_conv_info there would let rcf_conv2d_query write past the caller's struct).'''
    from rcf_amd import _lib
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'INTEGRATION.md')).read()
    desc = text[text.index('class ConvDesc'):text.index('class ConvInfo')]
    info = text[text.index('class ConvInfo'):text.index('_lib.rcf_conv2d_query.argtypes')]
    assert re.findall(r"'(\w+)'", desc) == [n for n, _ in _lib.ConvDesc._fields_]
    assert re.findall(r"\('(\w+)'", info) == [n for n, _ in _lib.ConvInfo._fields_]
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'rcf_hip.h')).read()
    hinfo = header[header.index('typedef struct rcf_conv_info {'):header.index('} rcf_conv_info;')]
    assert re.findall(r'\b(?:int|size_t)\s+(\w+);', hinfo) == [n for n, _ in _lib.ConvInfo._fields_]


# ---------------------------------------------------------------------------------------------------- C ABI negative paths
def test_every_export_rejects_null_arguments_without_launching(pkg):
    '''Every int-returning export, called with null pointers and zero extents, answers RCF_EINVAL (never a launch, never a crash);
    runs without a GPU because validation precedes every HIP call.'''
    import ctypes
    from rcf_amd import _lib
    lib = _lib.load()
    queries = {'rcf_version', 'rcf_device_ok', 'rcf_fc_bwd_workspace_floats', 'rcf_bce_workspace_doubles', 'rcf_transform_workspace_bytes',
               'rcf_points_to_depth_map_workspace_bytes', 'rcf_head_wgrad_workspace_floats', 'rcf_loss_workspace_floats',
                   'rcf_fuse_wp_infer_supported'}
    n = 0
    for name, (restype, argtypes) in _lib._SIGNATURES.items():
        if name in queries:
            continue
        args = [None if (t is _lib._P or 'LP_' in getattr(t, '__name__', '')) else t(0) for t in argtypes]
        rc = getattr(lib, name)(*args)
        assert rc == -1, (name, rc)       # RCF_EINVAL
        n += 1
    assert n >= 50
    # size queries answer 0 / a constant for nonsense extents instead of overflowing
    assert lib.rcf_fc_bwd_workspace_floats(-5, 3, 32) == 0
    assert lib.rcf_head_wgrad_workspace_floats(0, 0, 0, 0) == 0


def test_unsupported_shapes_answer_eunsupported_not_a_launch(pkg):
    from rcf_amd import _lib, ops
    lib = _lib.load()
    fake = 0x10000   # a non-null "device pointer": a rejected call must not dereference it (and there is no device here)
    import ctypes
    d = ops.make_fwd_desc(1, 32, 32, 16, 0, 5, 3, 1)          # c_out not a multiple of 4 (c_out == 1 is the head kernel's job)
    info = _lib.ConvInfo()
    assert lib.rcf_conv2d_query(ctypes.byref(d), ctypes.byref(info)) == -2
    assert lib.rcf_conv2d_fwd(ctypes.byref(d), fake, None, fake, fake, None, None) == -2
    d = ops.make_fwd_desc(1, 32, 32, 16, 6, 32, 3, 1)         # concat boundary not 4-aligned
    assert lib.rcf_conv2d_query(ctypes.byref(d), ctypes.byref(info)) == -2
    d = ops.make_fwd_desc(1, 32, 32, 16, 0, 32, 3, 1)
    d.ksize = 5
    assert lib.rcf_conv2d_query(ctypes.byref(d), ctypes.byref(info)) == -1
    assert lib.rcf_roi_pool_fwd(fake, fake, fake, fake, 1, 1, 8, 8, 3, 2, 2, 0.5, 4, 0, None) == -2
    assert lib.rcf_fc_fwd(fake, fake, fake, fake, 4, 3, 30, 1, 7, 64, 0, None) == -2
    assert lib.rcf_ew_blocks(1000, 3) <= 0
    assert lib.rcf_head_bn_blocks(1, 8, 8, 128) <= 0           # fused head path covers c <= 64


def test_c_abi_under_address_sanitizer(pkg, tmp_path):
    '''host-side ASan/UBSan harness over the C ABI (tests/abi/abi_harness.c, plain C99 -> the header is C).'''
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    exe = str(tmp_path / 'abi_harness')
    subprocess.run(['gcc', '-std=c99', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-I', os.path.join(ROOT, 'include'),
                    os.path.join(ROOT, 'tests', 'abi', 'abi_harness.c'), '-ldl', '-o', exe], check=True)
    from rcf_amd import _lib
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1')
    r = subprocess.run([exe, _lib.LIB_PATH], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'all checks passed' in r.stdout


# ---------------------------------------------------------------------------------------------------- bench.py launcher
def test_bench_refuses_to_measure_fewer_gpus_than_asked(pkg):
    '''`python bench.py --gpus 8` without a launcher starts its own ranks -- and on a box with fewer devices it must refuse
    (exit 2), never print a dp1 line (round-1 behaviour).  Here there is no device at all.'''
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert 'refusing' in r.stderr and '"metric"' not in r.stdout
    # under a launcher with a world size that contradicts --gpus it refuses too
    env.update({'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and '"metric"' not in r.stdout


def test_bench_expected_loss_fixture_is_the_oracle_on_the_bench_inputs(pkg):
    '''tests/golden/bench_expected.json (what bench.py checks its first training step against) is reproduced here for its small case.'''
    import json
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import make_bench_expected as mk
    rec = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'bench_expected.json')))
    assert 'train_b8_900x1600_p64' in rec and rec['train_b8_900x1600_p64']['first_step_loss'] > 0
    got = mk.first_step_loss(2, 224, 384, 32)[0]
    assert abs(got - rec['train_b2_224x384_p32']['first_step_loss']) < 1e-5 * abs(got)


# kernels allowed to spill VGPRs / use private scratch: cold variants only (never above 0.2 ms per step in profiles/r03_*_kernels.txt)
SPILL_ALLOW = [
    r'conv1x1_b16_kernel<PwCfg<[24], 4>',                       # bf16 1x1 onto 128 output channels (2 / 34 VGPRs; was 77 / 102): HBM-bound, ~30 us launches
    r'conv_split_kernel<SplitCfg<3, [12], (16|32), 2, 1, 2(, \w+)?>',   # bf16-OPERAND stride 2 on fp32 / bf16 tensors without the DMA path: unused by the shipped configurations
    r'conv_wgrad_dma_kernel<WgCfg<',                            # f32-MFMA weight gradients: RCF_CONV_SPLIT=0 builds and the 1x1 / stride-2 leftovers
    r'conv_fwd_kernel<FwdCfg<2, 2, 0, 1, 32, 32, 36, 2, 32, 2>',  # f32-MFMA 2x2 phases: RCF_CONV_SPLIT=0 only
    r'conv_split_kernel<SplitCfg<3, 1, (16|32), 0, 3, 1(, \w+)?>',      # three-plane 512-pixel tiles: the fp32_3plane side tier
    r'conv_split_kernel<SplitCfg<3, 1, 32, 0, 1, 1(, \w+)?>, false, StB16',   # bf16 tensors without the DMA path (RCF_B16_DMA=0)
]


def test_no_hot_kernel_spills_or_uses_scratch(pkg):
    '''Every kernel of the built library, from the code objects themselves (tools/kernel_meta.py: metadata + disassembly; no GPU, no
    recompilation): zero spilled VGPRs and not one scratch_load / scratch_store instruction, except an explicit allow-list of cold
    variants.  (A kernel whose SGPRs spill into VGPR lanes reports a non-zero private segment size -- the frame slots of those
    spills -- without touching scratch memory; counting the instructions tells that apart from a real spill.)  The default fp32
    tier's kernels (two fp16 planes: SplitCfg<..., 2, .>, WsCfg<..., 2>), the bf16 DMA kernels and every weight-gradient split kernel
    must be clean.'''
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import kernel_meta
    ks = kernel_meta.kernels(count_scratch_ops=True)
    assert len(ks) > 300
    dirty = lambda k: k['vgpr_spill'] > 0 or (k['scratch_ops'] or 0) > 0
    bad = [k for k in ks if dirty(k) and not any(re.search(p, k['name']) for p in SPILL_ALLOW)]
    assert not bad, '\n'.join('%s: %d spilled VGPRs, %d scratch instructions' % (k['name'], k['vgpr_spill'], k['scratch_ops']) for k in bad)
    hot = [k for k in ks if re.search(r'SplitCfg<\d, \d, \d+, \d, 2, \d(, \w+)?>|WsCfg<\d, \d, \d, \d+, \d(, \w+)?>|conv_b16_kernel', k['name'])]
    assert len(hot) > 60 and not any(dirty(k) for k in hot)


def test_generated_code_of_the_small_units_is_clean():
    '''tools/isa_lint.py (no GPU: hipcc -S for gfx950) on the elementwise / format / transform units: no flat accesses, no GOT loads,
    no spills, no store sitting behind an s_waitcnt vmcnt(0) -- the four code-generation patterns that cost the convolution
    kernels 3-30 % before they were found.  The two convolution units take minutes and are linted by hand.'''
    import subprocess, sys
    units = [os.path.join(ROOT, 'radar-camera-fusion-depth_amd', 'csrc', u) for u in ('rcf_elementwise.hip', 'rcf_formats.hip', 'rcf_transforms.hip')]
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'isa_lint.py')] + units, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    lines = [ln for ln in r.stdout.splitlines() if ln.endswith('flagged')]
    assert len(lines) == 3, r.stdout[-800:]
    for ln in lines:
        assert ln.endswith(' 0 flagged'), r.stdout[-1500:]


def test_error_codes_map_to_exception_types(pkg):
    '''_lib.check: RCF_EUNSUPPORTED (-2) raises RcfUnsupported -- the only error the engine answers with another form of the same
    operation (merged phase forms -> per-phase launches); RCF_EINVAL and hipError_t codes raise plain RcfError and propagate.'''
    from rcf_amd import _lib
    assert issubclass(_lib.RcfUnsupported, _lib.RcfError)
    _lib.check(0, 'ok')
    with pytest.raises(_lib.RcfUnsupported) as e:
        _lib.check(-2, 'rcf_conv2d_query')
    assert e.value.code == -2
    for rc in (-1, 719, 98):
        with pytest.raises(_lib.RcfError) as e:
            _lib.check(rc, 'rcf_conv2d_wgrad')
        assert not isinstance(e.value, _lib.RcfUnsupported) and e.value.code == rc


def test_rank_pinning_never_raises_and_reports(pkg, monkeypatch):
    '''parallel.pin_rank_to_gpu_numa_node (called by bench.py before anything touches the GPU): sysfs only; on a box without AMD render
    nodes, with a single rank, or switched off it leaves the affinity alone and says why.'''
    import os
    from rcf_amd import parallel
    before = os.sched_getaffinity(0)
    info = parallel.pin_rank_to_gpu_numa_node(local_rank=0, local_world=1)
    assert info['pinned'] is False and 'why' in info
    monkeypatch.setenv('RCF_RANK_AFFINITY', '0')
    info = parallel.pin_rank_to_gpu_numa_node(local_rank=3, local_world=8)
    assert info['pinned'] is False
    monkeypatch.delenv('RCF_RANK_AFFINITY')
    info = parallel.pin_rank_to_gpu_numa_node(local_rank=7, local_world=8)      # this container: no GPU, no render nodes
    assert isinstance(info, dict) and 'pinned' in info
    if not info['pinned']:
        assert os.sched_getaffinity(0) == before
    else:
        os.sched_setaffinity(0, before)
