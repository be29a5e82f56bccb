'''
GPU end-to-end parity of the drop-in FusionNetModel (HIP engine) against
  (a) the committed golden vectors captured from the real reference (tests/golden/T0..T3), and
  (b) the CPU oracle on fresh seeded inputs, including the full 900x1600 resolution of BASELINE.json.
Bar: 1e-3 relative (north_star, fp32); most checks sit one to two orders below it.
'''

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BAR = 1e-3


def _rel(a, b):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _named(model, what):
    out = []
    for prefix, mod in (('encoder.', model.encoder), ('decoder.', model.decoder)):
        it = mod.named_parameters() if what == 'p' else mod.named_buffers()
        out += [(prefix + k, v) for k, v in it if not k.endswith('num_batches_tracked')]
    return out


@pytest.fixture(scope='module')
def env():
    import rcf_amd  # noqa: F401
    from rcf_amd import _lib, synth, train
    assert torch.cuda.is_available()
    _lib.load()
    return synth, train


def _oracle_step(cfg, wseed, cb, dtype):
    """One train-mode forward/backward of the CPU oracle in `dtype`; returns (output, loss, {param: grad})."""
    from oracle.fusionnet_oracle import FusionNetOracle
    o = FusionNetOracle(**cfg)
    import rcf_amd
    rcf_amd.synth.fill_state_dict_([o.encoder, o.decoder], wseed)
    for mod in (o.encoder, o.decoder):
        mod.to(dtype)
    o.train()
    out = o.forward(cb['image'].to(dtype), cb['input_depth'].to(dtype))
    loss = o.compute_loss(out, cb['ground_truth'].to(dtype), cb['lidar_map'].to(dtype), 2.0)[0]
    loss.backward()
    return out.detach(), float(loss), {k: (None if p.grad is None else p.grad.detach().double()) for k, p in _named(o, 'p')}


def _check_gradients_against_fp64(hip_grads, g32, g64, tag, collect=None):
    """
    FusionNet's fp32 gradients are chaotic (LeakyReLU-sign / max-pool-argmax flips move some parameter gradients by
    1e-2 between ANY two fp32 implementations, e.g. PyTorch CPU fp32 vs fp64), so the bar is relative: the HIP path
    must be about as close to the fp64 truth as the fp32 CPU reference is (median within 3x, worst tensor within 5x).
    Reaching that needed fp64 accumulation of every BatchNorm sum (forward statistics per value, backward reductions):
    BN backward relies on dz being exactly mean-free, and 1e-5 of jitter in mean/invstd turns into 1e-3 errors in the
    next layer's nearly-cancelling sum of g (tools/diag_net.py, tools/diag_vt.py).
    """
    e_hip, e_cpu = [], []
    for k, ref in g64.items():
        if ref is None:
            assert hip_grads[k] is None, k
            continue
        e_hip.append(_rel(hip_grads[k], ref))
        e_cpu.append(_rel(g32[k], ref))
    e_hip, e_cpu = np.array(e_hip), np.array(e_cpu)
    print('%s gradients vs fp64: HIP median %.2e max %.2e | CPU-fp32 median %.2e max %.2e'
          % (tag, np.median(e_hip), e_hip.max(), np.median(e_cpu), e_cpu.max()))
    if collect is not None:     # a seed sweep judges the distribution (see test_fresh_seed_odd_size_against_oracle)
        collect.append((float(np.median(e_hip)), float(e_hip.max()), float(np.median(e_cpu)), float(e_cpu.max())))
        return
    assert np.median(e_hip) <= 3.0 * np.median(e_cpu) + 2e-5
    assert e_hip.max() <= 5.0 * e_cpu.max() + 2e-4


def _build(env, cfg, seed):
    synth, train = env
    m = train.build_model(cfg, device='cuda')
    synth.fill_state_dict_([m.encoder, m.decoder], seed)
    return m


def _gpu_batch(b):
    return {k: v.cuda() for k, v in b.items()}


def _loss(m, b, out):
    return m.compute_loss(image=b['image'], output_depth=out, ground_truth=b['ground_truth'], lidar_map=b['lidar_map'],
                          loss_func='l1', w_smoothness=0.0, loss_smoothness_kernel_size=-1,
                          validity_map_loss_smoothness=None, w_lidar_loss=2.0)


def test_t0_tiny_train_step_matches_reference_golden(env, golden_dir):
    synth, _ = env
    g = np.load(os.path.join(golden_dir, 'T0_tiny_train.npz'))
    n, h, w, k, dseed, wseed = [int(v) for v in g['meta']]
    m = _build(env, synth.TINY, wseed)
    b = _gpu_batch(synth.make_batch(n, h, w, k, seed=dseed))
    m.train()
    out = m.forward(image=b['image'], input_depth=b['input_depth'])
    loss, info = _loss(m, b, out)
    loss.backward()
    torch.cuda.synchronize()
    assert tuple(out.shape) == (n, 1, h, w)
    assert _rel(out, g['output']) < BAR
    np.testing.assert_allclose([float(loss), float(info['loss_supervised']), float(info['loss_lidar'])], g['loss'], rtol=BAR)
    unused = set(g['unused'].tolist())
    worst = 0.0
    for key, p in _named(m, 'p'):
        if key in unused:
            assert p.grad is None, key
            continue
        e = _rel(p.grad, g['grad:' + key])
        worst = max(worst, e)
        assert e < BAR, (key, e)
    for key, buf in _named(m, 'b'):
        assert _rel(buf, g['buf:' + key]) < BAR, key
    print('T0 worst gradient rel err %.2e' % worst)


def _build_transpose(env, cfg, seed):
    synth, train = env
    m = train.build_model(cfg, device='cuda', deconv_type='transpose')
    synth.fill_state_dict_([m.encoder, m.decoder], seed)
    return m


def test_t10_transposed_convolution_decoder_matches_reference_golden(env, golden_dir):
    '''deconv_type='transpose' (net_utils.TransposeConv2d, src/net_utils.py:94-153; selected at :507-513): forward as the 4-phase
    transposed convolution, dX as a stride-2 convolution, dW as a stride-2 weight gradient -- against fixture T10 from the REAL
    reference: output, loss, every parameter gradient (tiny net) and the gradient norms of the published net.'''
    synth, _ = env
    g = np.load(os.path.join(golden_dir, 'T10_transpose_tiny_train.npz'))
    n, h, w, k, dseed, wseed = [int(v) for v in g['meta']]
    m = _build_transpose(env, synth.TINY, wseed)
    assert [k2 for k2 in m.decoder.state_dict() if k2.endswith('deconv.deconv.weight')], 'reference state-dict names'
    b = _gpu_batch(synth.make_batch(n, h, w, k, seed=dseed))
    m.train()
    out = m.forward(image=b['image'], input_depth=b['input_depth'])
    loss, info = _loss(m, b, out)
    loss.backward()
    torch.cuda.synchronize()
    assert _rel(out, g['output']) < BAR
    np.testing.assert_allclose([float(loss), float(info['loss_supervised']), float(info['loss_lidar'])], g['loss'], rtol=BAR)
    unused = set(g['unused'].tolist())
    worst = 0.0
    for key, p in _named(m, 'p'):
        if key in unused:
            assert p.grad is None, key
            continue
        e = _rel(p.grad, g['grad:' + key])
        worst = max(worst, e)
        assert e < BAR, (key, e)
    for key, buf in _named(m, 'b'):
        assert _rel(buf, g['buf:' + key]) < BAR, key
    print('T10a worst gradient rel err %.2e' % worst)
    g = np.load(os.path.join(golden_dir, 'T10_transpose_published_train.npz'))
    n, h, w, k, dseed, wseed = [int(v) for v in g['meta']]
    m = _build_transpose(env, synth.PUBLISHED, wseed)
    b = _gpu_batch(synth.make_batch(n, h, w, k, seed=dseed))
    m.train()
    out = m.forward(image=b['image'], input_depth=b['input_depth'])
    loss, info = _loss(m, b, out)
    loss.backward()
    torch.cuda.synchronize()
    assert _rel(out, g['output']) < BAR
    np.testing.assert_allclose([float(loss), float(info['loss_supervised']), float(info['loss_lidar'])], g['loss'], rtol=BAR)
    grads = dict(_named(m, 'p'))
    for key, l2 in zip(g['grad_keys'].tolist(), g['grad_l2'].tolist()):
        got = float(grads[key].grad.double().norm())
        assert abs(got - l2) <= 5 * BAR * l2 + 1e-12, (key, got, l2)


def test_transposed_convolution_decoder_448_crop_eval_and_size_rule(env):
    '''The shipped training crop (448 x 448, bash/train_fusionnet_nuscenes.sh:25-26), batch 2, against the oracle (pinned to the
    reference by T10): train-mode output / loss, eval-mode output; and at 900 x 1600 the same failure as the reference
    (15 -> 30 rows against a 29-row skip).'''
    from oracle.fusionnet_oracle import FusionNetOracle
    synth, _ = env
    cb = synth.make_batch(2, 448, 448, 32, seed=611)
    b = _gpu_batch(cb)
    m = _build_transpose(env, synth.PUBLISHED, 61)
    o = FusionNetOracle(deconv_type='transpose', **synth.PUBLISHED)
    synth.fill_state_dict_([o.encoder, o.decoder], 61)
    for mode in ('train', 'eval'):
        getattr(m, mode)(); getattr(o, mode)()
        with torch.no_grad():
            out = m.forward(b['image'], b['input_depth'])
            ref = o.forward(cb['image'], cb['input_depth'])
            ref_loss = float(o.compute_loss(ref, cb['ground_truth'], cb['lidar_map'], 2.0)[0])
        loss, _ = _loss(m, b, out)
        assert _rel(out, ref) < BAR, mode
        assert abs(float(loss) - ref_loss) < BAR * abs(ref_loss)
    m.train()
    big = _gpu_batch(synth.make_batch(1, 900, 1600, 8, seed=5))
    with pytest.raises(RuntimeError, match='Sizes of tensors must match'):
        m.forward(big['image'], big['input_depth'])


def test_t1_published_config1_matches_reference_golden(env, golden_dir):
    synth, _ = env
    g = np.load(os.path.join(golden_dir, 'T1_published_train.npz'))
    n, h, w, k, dseed, wseed = [int(v) for v in g['meta']]
    m = _build(env, synth.PUBLISHED, wseed)
    b = _gpu_batch(synth.make_batch(n, h, w, k, seed=dseed))
    m.train()
    out = m.forward(image=b['image'], input_depth=b['input_depth'])
    loss, info = _loss(m, b, out)
    loss.backward()
    torch.cuda.synchronize()
    assert _rel(out, g['output']) < BAR
    mae_mm = float((out.detach().cpu() - torch.from_numpy(g['output'])).abs().mean()) * 1000.0
    print('T1 output MAE vs reference: %.4f mm' % mae_mm)
    np.testing.assert_allclose([float(loss), float(info['loss_supervised']), float(info['loss_lidar'])], g['loss'], rtol=BAR)
    grads = dict(_named(m, 'p'))
    for key, l2, sm in zip(g['grad_keys'].tolist(), g['grad_l2'].tolist(), g['grad_sum'].tolist()):
        got = float(grads[key].grad.double().norm())
        assert abs(got - l2) <= 5 * BAR * l2 + 1e-12, (key, got, l2)
    for key in g['unused'].tolist():
        assert grads[key].grad is None
    bufs = dict(_named(m, 'b'))
    for key, l2 in zip(g['buf_keys'].tolist(), g['buf_l2'].tolist()):
        assert abs(float(bufs[key].double().norm()) - l2) <= BAR * l2, key


def test_t3_eval_mode_matches_reference_golden(env, golden_dir):
    synth, _ = env
    g = np.load(os.path.join(golden_dir, 'T3_eval.npz'))
    for tag, cfg in (('tiny', synth.TINY), ('published', synth.PUBLISHED)):
        n, h, w, k, dseed, wseed = [int(v) for v in g[tag + '_meta']]
        m = _build(env, cfg, wseed)
        m.eval()
        b = _gpu_batch(synth.make_batch(n, h, w, k, seed=dseed))
        before = {k_: v.clone() for k_, v in _named(m, 'b')}
        with torch.no_grad():
            out = m.forward(image=b['image'], input_depth=b['input_depth'])
        torch.cuda.synchronize()
        assert _rel(out, g[tag + '_output']) < BAR
        for k_, v in _named(m, 'b'):
            assert torch.equal(v, before[k_]), 'eval must not touch running statistics'


@pytest.mark.parametrize('fused', [True, False], ids=['FusedAdam', 'torch.optim.Adam'])
def test_t2_three_adam_steps_match_reference_trajectory(env, golden_dir, fused):
    synth, train = env
    g = np.load(os.path.join(golden_dir, 'T2_tiny_adam3.npz'))
    n, h, w, k, dseed, wseed = [int(v) for v in g['meta']]
    m = _build(env, synth.TINY, wseed)
    if fused:
        opt = train.make_optimizer(m, lr=1e-3)
    else:   # the reference's own optimizer drives the HIP model unchanged
        opt = torch.optim.Adam([{'params': m.parameters(), 'weight_decay': 0.0}], lr=1e-3)
    m.train()
    for step in range(3):
        b = _gpu_batch(synth.make_batch(n, h, w, k, seed=dseed + step))
        loss, _, _ = train.train_step(m, opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])
        assert abs(float(loss) - g['losses'][step]) < BAR * g['losses'][step], (step, float(loss))
    psum = float(sum(p.detach().double().abs().sum() for p in m.parameters()))
    assert abs(psum - float(g['param_abs_sum'])) < 1e-4 * float(g['param_abs_sum'])
    bsum = float(sum(v.detach().double().abs().sum() for _, v in _named(m, 'b')))
    assert abs(bsum - float(g['buffer_abs_sum'])) < BAR * float(g['buffer_abs_sum'])


@pytest.mark.parametrize('tier', ['fp32', 'fp32_3plane'])
def test_fresh_seed_odd_size_against_oracle(env, tier):
    '''Published net at 2 x 113 x 200 (odd sizes at every level), fresh seeds, on BOTH operand arithmetics of the fp32 configuration
    (two scaled fp16 planes = the default; three bf16 planes): output and loss against the fp32 and the fp64 oracle for each seed, and
    the parameter gradients against fp64 next to the CPU fp32 oracle's over a SEED SWEEP.  On this small, chaotic case the per-seed
    ratio of the per-tensor medians is noise -- over seeds 5..12 it runs from 0.1x to 9x for every arithmetic tier of this library, the
    f32-MFMA-only build included (tools/diag_seeds.py) -- so the 3x / 5x bar of _check_gradients_against_fp64
    is held by the geometric mean over the sweep; PER SEED the whole-gradient relative L2 error (all parameters as one vector: a
    LeakyReLU flip in one small tensor does not move it) is held to 3x the CPU fp32 oracle's + 2e-3, so a moderate loss of accuracy in
    one tier cannot hide in the sweep.'''
    synth, _ = env
    rows = []
    for wseed in (5, 6, 7, 8):
        m = _build(env, synth.PUBLISHED, wseed)
        m.compute_dtype = tier
        cb = synth.make_batch(2, 113, 200, 16, seed=wseed + 4)
        b = _gpu_batch(cb)
        m.train()
        out = m.forward(image=b['image'], input_depth=b['input_depth'])
        loss, _ = _loss(m, b, out)
        loss.backward()
        torch.cuda.synchronize()
        o64, l64, g64 = _oracle_step(synth.PUBLISHED, wseed, cb, torch.float64)
        o32, l32, g32 = _oracle_step(synth.PUBLISHED, wseed, cb, torch.float32)
        assert _rel(out, o32) < BAR and _rel(out, o64) < BAR
        assert abs(float(loss) - l32) < BAR * abs(l32)
        grads = {k: p.grad for k, p in _named(m, 'p')}
        _check_gradients_against_fp64(grads, g32, g64, '%s, seed %d, 2x113x200' % (tier, wseed), collect=rows)
        nn = sum(float((g64[k] ** 2).sum()) for k in g64 if g64[k] is not None)
        l2h = (sum(float(((grads[k].detach().cpu().double() - g64[k]) ** 2).sum()) for k in g64 if g64[k] is not None) / nn) ** 0.5
        l2c = (sum(float(((g32[k] - g64[k]) ** 2).sum()) for k in g64 if g64[k] is not None) / nn) ** 0.5
        print('   whole-gradient relative L2 error: HIP %.2e, CPU fp32 %.2e' % (l2h, l2c))
        assert l2h <= 3.0 * l2c + 2e-3, (tier, wseed, l2h, l2c)
        del m
    r = np.array(rows)
    gm = np.exp(np.log(r).mean(0))
    print('geometric means over the sweep: HIP median %.2e max %.2e | CPU-fp32 median %.2e max %.2e' % tuple(gm))
    assert gm[0] <= 3.0 * gm[2] + 2e-5
    assert gm[1] <= 5.0 * gm[3] + 2e-4


@pytest.mark.parametrize('kind', ['l2', 'smoothl1', 'l1+smoothness'])
def test_loss_variants_against_the_reference_fixture(env, golden_dir, kind):
    '''compute_loss with loss_func 'l2' / 'smoothl1' (src/fusionnet_model.py:255-275) and with the local smoothness term (w_smoothness
    0.5, loss_smoothness_kernel_size -1: :277-281) against fixture T12 from the REAL reference: the loss terms, the gradient of the loss
    with respect to the output depth (the loss kernels alone), and the norm of every parameter gradient of the whole backward pass.'''
    synth, _ = env
    g = np.load(os.path.join(golden_dir, 'T12_loss_variants.npz'))
    n, h, w, k, dseed, wseed = [int(v) for v in g['meta']]
    m = _build(env, synth.TINY, wseed)
    m.train()
    b = _gpu_batch(synth.make_batch(n, h, w, k, seed=dseed))
    out = m.forward(image=b['image'], input_depth=b['input_depth'])
    out.retain_grad()
    loss, info = m.compute_loss(image=b['image'], output_depth=out, ground_truth=b['ground_truth'], lidar_map=b['lidar_map'],
                                loss_func=kind.split('+')[0], w_smoothness=0.5 if '+' in kind else 0.0, loss_smoothness_kernel_size=-1,
                                validity_map_loss_smoothness=None, w_lidar_loss=2.0)
    loss.backward()
    torch.cuda.synchronize()
    got = [float(loss), float(info['loss_supervised']), float(info['loss_lidar']), float(info['loss_smoothness'])]
    np.testing.assert_allclose(got, g[kind + ':loss'], rtol=BAR, atol=1e-12)
    assert _rel(out.grad, torch.from_numpy(g[kind + ':dloss_doutput'])) < BAR
    grads = {kk: p.grad for kk, p in _named(m, 'p') if p.grad is not None}
    for key, l2 in zip(g[kind + ':grad_keys'].tolist(), g[kind + ':grad_l2'].tolist()):
        assert abs(float(grads[key].double().norm()) - l2) <= 1e-2 * l2 + 1e-9, (kind, key)
    if '+' in kind:   # the Sobel variant stays unimplemented and says so
        with pytest.raises(ValueError):
            m.compute_loss(image=b['image'], output_depth=out.detach(), ground_truth=b['ground_truth'], lidar_map=b['lidar_map'],
                           loss_func='l1', w_smoothness=0.5, loss_smoothness_kernel_size=7, validity_map_loss_smoothness=None, w_lidar_loss=2.0)


def test_fusionnet34_against_the_reference_fixture(env, golden_dir):
    '''encoder_type ['fusionnet34', 'batch_norm'] (src/fusionnet_model.py:84-90, src/networks.py:305-311: 3, 4, 6, 3, 3 ResNet blocks per
    level) against fixture T13 from the REAL reference: output, loss terms and the norm of every one of the 317 parameter gradients.'''
    synth, _ = env
    from rcf_amd.fusionnet_model import FusionNetModel
    g = np.load(os.path.join(golden_dir, 'T13_fusionnet34_tiny_train.npz'))
    n, h, w, k, dseed, wseed = [int(v) for v in g['meta']]
    cfg = synth.TINY
    m = FusionNetModel(cfg['input_channels_image'], cfg['input_channels_depth'], ['fusionnet34', 'batch_norm'], cfg['n_filters_encoder_image'],
                       cfg['n_filters_encoder_depth'], 'weight_and_project', ['multiscale', 'batch_norm'], 1, cfg['n_filters_decoder'], 'up',
                       'leaky_relu', 'kaiming_uniform', 1.0, 100.0, device='cuda')
    synth.fill_state_dict_([m.encoder, m.decoder], wseed)
    assert sum(p.numel() for p in m.parameters()) == int(g['n_params'])
    m.train()
    b = _gpu_batch(synth.make_batch(n, h, w, k, seed=dseed))
    out = m.forward(image=b['image'], input_depth=b['input_depth'])
    loss, info = _loss(m, b, out)
    loss.backward()
    torch.cuda.synchronize()
    assert _rel(out, torch.from_numpy(g['output'])) < BAR
    np.testing.assert_allclose([float(loss), float(info['loss_supervised']), float(info['loss_lidar'])], g['loss'], rtol=BAR)
    grads = {kk: p.grad for kk, p in _named(m, 'p') if p.grad is not None}
    assert sorted(grads) == sorted(g['grad_keys'].tolist())
    for key, l2 in zip(g['grad_keys'].tolist(), g['grad_l2'].tolist()):
        assert abs(float(grads[key].double().norm()) - l2) <= 1e-2 * l2 + 1e-9, key


def test_checkpoint_round_trip_and_reference_key_names(env, tmp_path):
    synth, train = env
    m = _build(env, synth.TINY, 3)
    opt = train.make_optimizer(m, lr=1e-3)
    b = _gpu_batch(synth.make_batch(1, 64, 96, 4, seed=1))
    m.train()
    train.train_step(m, opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])
    m.data_parallel()      # the reference always calls it before saving (src/fusionnet_main.py:198)
    path = str(tmp_path / 'model-1.pth')
    m.save_model(path, 1, opt)
    ck = torch.load(path, map_location='cpu')
    assert set(ck.keys()) == {'train_step', 'optimizer_state_dict', 'encoder_state_dict', 'decoder_state_dict'}
    assert all(k.startswith('module.') for k in ck['encoder_state_dict'])
    assert 'module.blocks2_image.0.conv1.conv.weight' in ck['encoder_state_dict']
    assert 'module.deconv0.deconv.conv.conv.weight' in ck['decoder_state_dict']
    m2 = train.build_model(synth.TINY, device='cuda')
    opt2 = train.make_optimizer(m2, lr=1e-3)
    step, _ = m2.restore_model(path, opt2)
    assert step == 1
    m.eval(); m2.eval()
    with torch.no_grad():
        a = m.forward(b['image'], b['input_depth'])
        c = m2.forward(b['image'], b['input_depth'])
    assert torch.equal(a, c)
    # optimizer state followed: one more identical step gives identical parameters
    m.train(); m2.train()
    train.train_step(m, opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])
    train.train_step(m2, opt2, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])
    for (ka, pa), (kb, pb) in zip(_named(m, 'p'), _named(m2, 'p')):
        assert torch.equal(pa, pb), ka


def test_engine_recovers_after_an_exception_on_a_side_stream(env):
    '''An error raised in the middle of a forward or a backward pass -- here injected into the 3rd / 12th BatchNorm launch, i.e. while
    the depth branch's stream resp. the weight-gradient stream is current -- must leave the caller on ITS stream with no engine flag
    set (Engine.recover): the next step of the same model is then bitwise the step of a model that never failed.'''
    from rcf_amd import ops as _ops
    from rcf_amd._lib import RcfError
    synth, train = env
    b = _gpu_batch(synth.make_batch(2, 70, 102, 6, seed=4))
    args = (b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])

    def clean_step():
        m = _build(env, synth.TINY, 9)
        opt = train.make_optimizer(m, lr=1e-3)
        m.train()
        loss, _, out = train.train_step(m, opt, *args)
        torch.cuda.synchronize()
        return m, opt, out.detach().clone(), float(loss), m._param_arena.detach().clone()

    _, _, out_ref, loss_ref, params_ref = clean_step()
    for victim, nth in (('bn_act_fwd', 3), ('bn_act_bwd_apply', 12)):
        m = _build(env, synth.TINY, 9)
        opt = train.make_optimizer(m, lr=1e-3)
        m.train()
        eng = m._engine
        assert eng.wgrad_side and eng.branch_stream
        real = getattr(_ops, victim)
        calls = [0]

        def failing(*a, **k):
            calls[0] += 1
            if calls[0] == nth:
                raise RcfError('injected failure in %s' % victim)
            return real(*a, **k)
        setattr(_ops, victim, failing)
        try:
            with pytest.raises(RcfError):
                train.train_step(m, opt, *args)
        finally:
            setattr(_ops, victim, real)
        torch.cuda.synchronize()
        assert torch.cuda.current_stream() == torch.cuda.default_stream()
        assert not eng._in_branch and not eng._side_busy and not eng._branch_busy and not eng._open_switches and not eng.in_backward
        # the failed step changed nothing that matters: parameters untouched (the optimizer never ran); a fresh step is the clean one
        m2 = _build(env, synth.TINY, 9)
        m._param_arena.copy_(m2._param_arena)
        for (k, buf), (_, buf2) in zip(_named(m, 'b'), _named(m2, 'b')):
            buf.copy_(buf2)
        opt = train.make_optimizer(m, lr=1e-3)
        loss, _, out = train.train_step(m, opt, *args)
        torch.cuda.synchronize()
        assert float(loss) == loss_ref and torch.equal(out, out_ref) and torch.equal(m._param_arena, params_ref), victim


@pytest.mark.parametrize('tier', ['fp32', 'bf16'])
def test_one_launch_phase_forms_against_the_per_phase_forms_through_the_whole_net(env, tier):
    '''The published net, one training step at 2 x 96 x 160 (every up-convolution an exact 2x: odd sizes take the nearest-gather 3x3 form
    and are covered per kernel in test_hip_f16x2.py / test_hip_bf16.py), with the one-launch forms of the phase
    convolutions (up-2x forward from one staged tile, stride-2 input gradient from one staged tile, up-2x weight gradient in phase
    pairs, stride-2 weight gradient in one launch) against the same step on the per-phase launches: the engine must actually TAKE the
    one-launch forms (its per-shape records say so), output, loss and every activation gradient path are bitwise (the forward and the
    input gradients are bitwise per layer), and the parameter gradients agree to fp32 summation order.'''
    synth, _ = env
    cb = synth.make_batch(2, 96, 160, 16, seed=21)
    b = _gpu_batch(cb)
    res = {}
    for forms in ('one-launch', 'per-phase'):
        m = _build(env, synth.PUBLISHED, 17)
        m.compute_dtype = tier
        eng = m._engine
        if forms == 'per-phase':
            # (bf16 tensors: the merged forward takes its BatchNorm statistics in other partial rows -- fp64 sums in another order, a
            # coefficient one fp32 ulp off here and there, and a bf16 rounding of some gradient element flips: parameter gradients of
            # small tensors then differ by 1e-2 between ANY two such orders.  The forward form is therefore held fixed for bf16 and
            # its bitwise equality with the four launches is the kernel test's, tests/test_hip_bf16.py)
            eng.up2x_one_launch = False if tier == 'fp32' else None
            eng.up2x_wgrad_one_launch = eng.s2_dgrad_one_launch = eng.s2_wgrad_one_launch = False
        m.train()
        out = m.forward(image=b['image'], input_depth=b['input_depth'])
        loss, _ = _loss(m, b, out)
        loss.backward()
        torch.cuda.synchronize()
        res[forms] = (out.detach().clone(), float(loss), {k: p.grad.detach().clone() for k, p in _named(m, 'p') if p.grad is not None})
        if forms == 'one-launch':   # the up-convolutions and the stride-2 convolutions of both encoder branches, every one on its one-launch form
            assert len(eng._up2x_wgrad_ok) >= 3 and all(eng._up2x_wgrad_ok.values()), eng._up2x_wgrad_ok
            assert len(eng._s2_dgrad_ok) >= 3 and all(eng._s2_dgrad_ok.values()), eng._s2_dgrad_ok
            assert len(eng._s2_wgrad_ok) >= 3 and all(eng._s2_wgrad_ok.values()), eng._s2_wgrad_ok
        else:
            assert not eng._up2x_wgrad_ok and not eng._s2_dgrad_ok and not eng._s2_wgrad_ok
        del m
    (o1, l1, g1), (o4, l4, g4) = res['one-launch'], res['per-phase']
    assert torch.equal(o1, o4) and l1 == l4
    worst = 0.0
    for k in g4:
        d = float((g1[k].double() - g4[k].double()).abs().max() / (g4[k].double().abs().max() + 1e-30))
        worst = max(worst, d)
        assert d < 5e-5, (k, d)     # weight gradients: another summation order of the same products
    print('%s: output and loss bitwise; parameter gradients within %.1e of the per-phase forms' % (tier, worst))


def test_gradient_accumulation_without_zero_grad(env):
    synth, _ = env
    m = _build(env, synth.TINY, 4)
    b = _gpu_batch(synth.make_batch(1, 64, 96, 4, seed=2))
    m.eval()   # running-stat BN: the two passes are identical, so the accumulated gradient is exactly double
    grads = []
    for _ in range(2):
        out = m.forward(b['image'], b['input_depth'])
        loss, _ = _loss(m, b, out)
        loss.backward()
        grads.append(m.decoder.output0.conv.weight.grad.clone())
    assert _rel(grads[1], 2 * grads[0]) < 1e-6


def test_full_resolution_900x1600_against_oracle(env):
    '''BASELINE.json's resolution, batch 1, published net: output, loss and every parameter gradient.'''
    import time
    synth, _ = env
    m = _build(env, synth.PUBLISHED, 8)
    cb = synth.make_batch(1, 900, 1600, 64, seed=1234)
    b = _gpu_batch(cb)
    m.train()
    torch.cuda.synchronize(); t0 = time.time()
    out = m.forward(image=b['image'], input_depth=b['input_depth'])
    loss, _ = _loss(m, b, out)
    loss.backward()
    torch.cuda.synchronize(); t_gpu = time.time() - t0
    t0 = time.time()
    o32, l32, g32 = _oracle_step(synth.PUBLISHED, 8, cb, torch.float32)
    t_cpu = time.time() - t0
    o64, l64, g64 = _oracle_step(synth.PUBLISHED, 8, cb, torch.float64)
    e_out = _rel(out, o32)
    mae_mm = float((out.detach().cpu() - o32).abs().mean()) * 1000.0
    print('900x1600: out rel %.2e (vs fp64 %.2e; CPU fp32 vs fp64 %.2e), MAE %.4f mm, loss %.6f vs %.6f; first-call gpu %.2fs, cpu oracle %.1fs'
          % (e_out, _rel(out, o64), _rel(o32, o64), mae_mm, float(loss), l32, t_gpu, t_cpu))
    assert e_out < BAR and _rel(out, o64) < BAR
    assert abs(float(loss) - l32) < BAR * abs(l32)
    _check_gradients_against_fp64({k: p.grad for k, p in _named(m, 'p')}, g32, g64, '900x1600')


def _dp_gpu_worker(rank, world, port, tmpdir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)   # gloo moves CUDA tensors; both ranks share cuda:0
    import rcf_amd  # noqa: F401
    from rcf_amd import synth, train
    torch.manual_seed(7)
    m = train.build_model(synth.TINY, device='cuda')
    synth.fill_state_dict_([m.encoder, m.decoder], 31)
    m.data_parallel()
    assert m._dp is not None
    opt = train.make_optimizer(m, lr=1e-3)
    b = {k: v.cuda() for k, v in synth.make_batch(2, 64, 96, 6, seed=500 + rank).items()}
    m.train()
    out = m.forward(b['image'], b['input_depth'])
    loss, info = m.compute_loss(b['image'], out, b['ground_truth'], b['lidar_map'], 'l1', 0.0, -1, None, 2.0)
    opt.zero_grad(); loss.backward()
    g = m._grad_arena[:m._n_used].clone()
    opt.step()
    torch.cuda.synchronize()
    bufs = {prefix + k: v.detach().cpu().clone() for prefix, mod in (('encoder.', m.encoder), ('decoder.', m.decoder))
            for k, v in mod.named_buffers() if not k.endswith('num_batches_tracked')}
    torch.save({'loss': float(loss), 'terms': [float(info['loss_supervised']), float(info['loss_lidar'])], 'grad': g.cpu(),
                'param': m._param_arena.detach().cpu().clone(), 'out': out.detach().cpu(), 'bufs': bufs}, os.path.join(tmpdir, 'dp%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)   # a rank that never joins must not hold the suite
def test_data_parallel_step_two_ranks_on_one_gpu(env, tmp_path):
    '''
    The DP training step end to end (bucketed SUM all-reduce launched from the tape, loss normalised by the GLOBAL valid
    counts) with 2 ranks sharing cuda:0 over gloo, against a single-process emulation of what nn.DataParallel computes:
    per-replica BatchNorm statistics, ONE masked mean over the gathered batch (src/fusionnet_main.py:385), summed gradients.
    '''
    import torch.multiprocessing as mp
    synth, train = env
    port = 29700 + (os.getpid() % 1000)
    mp.spawn(_dp_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(str(tmp_path), 'dp%d.pt' % k)) for k in range(2)]
    assert r[0]['loss'] == r[1]['loss']                                  # the global masked mean, identical on both ranks
    assert torch.equal(r[0]['grad'], r[1]['grad']) and torch.equal(r[0]['param'], r[1]['param'])

    # emulation: each replica forward with its own BN statistics, loss = global masked mean, gradients summed
    from rcf_amd import ops
    m = _build(env, synth.TINY, 31)
    m.train()
    batches = [_gpu_batch(synth.make_batch(2, 64, 96, 6, seed=500 + k)) for k in range(2)]
    sums = []
    for b in batches:
        with torch.no_grad():
            out = m.forward(b['image'], b['input_depth'])
        s = torch.empty(4, dtype=torch.float64, device='cuda')
        ops.l1_loss_fwd(out.contiguous(), b['ground_truth'], b['lidar_map'], s)
        sums.append(s)
    # undo the running-stat updates of the two probing forwards: rebuild and replay with gradients
    m = _build(env, synth.TINY, 31)
    m.train()
    tot = sums[0] + sums[1]
    want_loss = float(tot[0] / tot[1] + 2.0 * tot[2] / tot[3])
    grad = torch.zeros(m._n_used, device='cuda')
    for b in batches:
        out = m.forward(b['image'], b['input_depth'])
        dd = torch.empty_like(out)
        ops.l1_loss_bwd(out.detach().contiguous(), b['ground_truth'], b['lidar_map'], tot, None, 2.0, dd)
        for p in m.parameters():
            p.grad = None
        out.backward(dd)
        grad += m._grad_arena[:m._n_used]
    assert abs(r[0]['loss'] - want_loss) < 1e-5 * abs(want_loss)
    assert _rel(r[0]['grad'], grad) < 1e-4


def _dp_capture_worker(rank, world, port, tmpdir):
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)   # gloo moves CUDA tensors; both ranks share cuda:0
    import rcf_amd  # noqa: F401
    from rcf_amd import synth, train
    batches = [{k: v.cuda() for k, v in synth.make_batch(2, 64, 96, 6, seed=600 + 10 * s + rank).items()} for s in range(3)]
    res = {}
    for mode in ('eager', 'segments'):
        m = train.build_model(synth.TINY, device='cuda')
        synth.fill_state_dict_([m.encoder, m.decoder], 33)
        m.data_parallel()
        assert m._dp is not None
        opt = train.make_optimizer(m, lr=1e-3)
        m.train()
        b0 = batches[0]
        losses, host = [], 0.0
        if mode == 'segments':
            step = m.capture_training_step(opt, b0['image'], b0['input_depth'], b0['ground_truth'], b0['lidar_map'])
            assert step.segments is not None and len(step.segments) >= 3
            n_seg = len(step.segments)
        for b in batches:
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.time()
            if mode == 'segments':
                loss = step(b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])
            else:
                loss = train.train_step(m, opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])[0]
            host += time.time() - t0
            torch.cuda.synchronize()
            losses.append(float(loss.detach()))
        bufs = [v.detach().cpu().clone() for mod in (m.encoder, m.decoder) for k, v in mod.named_buffers()]
        res[mode] = {'losses': losses, 'param': m._param_arena.detach().cpu().clone(), 'bufs': bufs, 'host_ms': 1000.0 * host / 3,
                     'steps': [float(st['step']) for st in opt.state_dict()['state'].values()][:1]}
        if mode == 'segments':
            res[mode]['n_seg'] = n_seg
        del m, opt
    torch.save(res, os.path.join(tmpdir, 'cap%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_data_parallel_step_as_graph_segments_is_bitwise_the_eager_step(env, tmp_path):
    '''FusionNetModel.capture_training_step under data parallelism (2 ranks sharing cuda:0 over gloo): the step recorded as hipGraph
    segments between its exchange points -- loss sums, gradient buckets, the wait before Adam -- and replayed with the collectives
    launched eagerly in between.  Three steps on three different batches: losses, parameters, BatchNorm buffers and the optimizer's
    step count are BITWISE those of the eager data-parallel steps, on both ranks; the host time of a replayed step is printed next to
    the eager step's (the verdict's bar: <= 3 ms).'''
    import torch.multiprocessing as mp
    port = 29800 + (os.getpid() % 1000)
    mp.spawn(_dp_capture_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for k in range(2):
        r = torch.load(os.path.join(str(tmp_path), 'cap%d.pt' % k))
        e, c = r['eager'], r['segments']
        print('rank %d: %d segments; host time per step: eager %.2f ms, segments %.2f ms; losses %s' % (k, c['n_seg'], e['host_ms'], c['host_ms'], c['losses']))
        assert e['losses'] == c['losses']
        assert torch.equal(e['param'], c['param'])
        assert all(torch.equal(a, b) for a, b in zip(e['bufs'], c['bufs']))
        assert e['steps'] == c['steps'] == [3.0]
        # (no bar on these host times: gloo's collectives on CUDA tensors block the host and two processes time-slice one GPU here;
        # the host cost of a replayed step is measured over RCCL in test_segmented_step_host_time_over_rccl)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), 'cap%d.pt' % k)) for k in range(2))
    assert torch.equal(r0['segments']['param'], r1['segments']['param'])     # the replicas stay identical


def _rccl_host_time_worker(rank, world, port, tmpdir):
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('nccl', rank=0, world_size=1)     # RCCL's real launch path (stream-ordered collectives), one rank
    import rcf_amd  # noqa: F401
    from rcf_amd import synth, train
    from rcf_amd.parallel import GradientBuckets
    m = train.build_model(synth.PUBLISHED, device='cuda')
    synth.fill_state_dict_([m.encoder, m.decoder], 3)
    m._is_data_parallel = True
    m._dp = GradientBuckets(m)          # (data_parallel() arms it only for world sizes > 1)
    opt = train.make_optimizer(m, lr=1e-3)
    m.train()
    b = {k: v.cuda() for k, v in synth.make_batch(4, 224, 384, 32, seed=9).items()}
    args = (b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])
    def host_ms(fn, warm, n=5, repeats=3):
        # the smallest mean over `repeats` groups of n calls: a busy host inflates single groups, never deflates them
        for _ in range(warm):
            fn()
        best = float('inf')
        for _ in range(repeats):
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(n):
                fn()
            best = min(best, (time.time() - t0) / n * 1e3)
        torch.cuda.synchronize()
        return best
    eager_host = host_ms(lambda: train.train_step(m, opt, *args), warm=3)
    step = m.capture_training_step(opt, *args)
    seg_host = host_ms(step, warm=6)
    t0 = time.time()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    seg_wall = (time.time() - t0) / 5 * 1e3
    torch.save({'eager_host': eager_host, 'seg_host': seg_host, 'seg_wall': seg_wall, 'n_seg': len(step.segments)}, os.path.join(tmpdir, 'host.pt'))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_segmented_step_host_time_over_rccl(env, tmp_path):
    '''Host time of ENQUEUEING one data-parallel training step (published net, batch 4, 224 x 384) with the exchange going through RCCL
    (one rank: the collectives move nothing, their launch path is the real one).  The statement is RELATIVE -- the step replayed as
    graph segments + the RCCL calls between them costs the host less than half of the eager step's ~1100 ctypes launches -- because
    absolute host times differ 3x between boxes (1.0 ms here, 3.5 ms on the round-4 driver box, eager 11 / 17.8 ms); the absolute
    value is reported by bench.py (`dp.host_ms_per_step`), not asserted.  Ordered last in the suite (conftest.py).'''
    import torch.multiprocessing as mp
    port = 29900 + (os.getpid() % 1000)
    mp.spawn(_rccl_host_time_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    r = torch.load(os.path.join(str(tmp_path), 'host.pt'))
    print('host time per data-parallel step over RCCL: eager %.2f ms, %d graph segments %.2f ms (wall %.2f ms)' % (r['eager_host'], r['n_seg'], r['seg_host'], r['seg_wall']))
    assert r['seg_host'] < 0.5 * r['eager_host']


@pytest.mark.timeout(300)
def test_data_parallel_step_matches_the_reference_fixture(env, golden_dir, tmp_path):
    '''The same 2-rank step against fixture T11 = the REAL reference's nn.DataParallel semantics computed on CPU
    (tests/golden/make_golden_dp.py: per-replica BatchNorm statistics, one masked mean over the gathered batch, reduce-added
    gradients, replica 0's running statistics): outputs of both ranks, loss terms, EVERY parameter gradient and rank 0's buffers.'''
    import torch.multiprocessing as mp
    synth, train = env
    g = np.load(os.path.join(golden_dir, 'T11_dp2_tiny_train.npz'))
    assert [int(v) for v in g['meta']] == [2, 64, 96, 6, 500, 31]        # the worker's batches and weights
    port = 29800 + (os.getpid() % 1000)
    mp.spawn(_dp_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(str(tmp_path), 'dp%d.pt' % k)) for k in range(2)]
    out = torch.cat([r[0]['out'], r[1]['out']], 0)
    assert _rel(out, g['output']) < BAR
    np.testing.assert_allclose([r[0]['loss']] + r[0]['terms'], g['loss'], rtol=1e-5)
    assert abs(r[0]['loss'] - float(g['single_replica_loss'])) > 1e-4 * r[0]['loss']      # and it is NOT the single-replica batch-4 loss
    m = _build(env, synth.TINY, 31)                      # name -> offset in the gradient arena
    names = {id(p): k for k, p in _named(m, 'p')}
    off, worst = 0, 0.0
    for p in m._used_params:
        n = p.numel()
        want = g['grad_' + names[id(p)]]
        worst = max(worst, _rel(r[0]['grad'][off:off + n].view(p.shape), want))
        off += n
    print('2-rank HIP step vs the reference\'s DataParallel step: worst gradient tensor rel %.2e' % worst)
    assert worst < BAR
    assert len(g['grad_keys']) == len(m._used_params)
    for k in g['buf_keys'].tolist():
        if not k.endswith('num_batches_tracked'):
            assert _rel(r[0]['bufs'][k], g['buf_' + k]) < BAR, k


def test_t1_with_bn_on_load_in_the_conv_kernels(env, golden_dir):
    '''Engine.bn_on_load (the consuming split conv / wgrad kernels apply BatchNorm + lrelu while they load the raw conv output;
    off by default because it is slower) gives the same published-net training step: output, loss and gradient norms of T1.
    A deferred activation is materialised by whichever consumer reads it first, so this mode must stay on ONE stream (the
    fusion on the branch stream materialised the stem's activation that max_pool on the main stream then read unordered).'''
    synth, _ = env
    g = np.load(os.path.join(golden_dir, 'T1_published_train.npz'))
    n, h, w, k, dseed, wseed = [int(v) for v in g['meta']]
    m = _build(env, synth.PUBLISHED, wseed)
    m._engine.bn_on_load = True
    b = _gpu_batch(synth.make_batch(n, h, w, k, seed=dseed))
    m.train()
    out = m.forward(image=b['image'], input_depth=b['input_depth'])
    loss, info = _loss(m, b, out)
    loss.backward()
    torch.cuda.synchronize()
    eng = m._engine
    assert eng._branch is None and eng._side is None, 'bn_on_load ran on a side stream'
    assert _rel(out, g['output']) < BAR
    np.testing.assert_allclose([float(loss.detach()), float(info['loss_supervised']), float(info['loss_lidar'])], g['loss'], rtol=BAR)
    grads = dict(_named(m, 'p'))
    for key, l2 in zip(g['grad_keys'].tolist(), g['grad_l2'].tolist()):
        got = float(grads[key].grad.double().norm())
        assert abs(got - l2) <= 2e-3 * max(l2, 1e-6), (key, got, l2)


def test_bf16_compute_mode_published_net(env, golden_dir):
    '''FusionNetModel.compute_dtype = 'bf16' (bf16 operands, fp32 accumulate in the split conv kernels; BASELINE.json configs 3-5):
    a training step of the published net stays close to the fp32 reference (bf16 has 8 significant bits: a loose bar) and differs
    from the fp32 path (i.e. the mode is really in use).'''
    if os.environ.get('RCF_CONV_SPLIT') == '0':
        pytest.skip('the bf16-operand mode lives in the split kernels, which RCF_CONV_SPLIT=0 turns off')
    synth, _ = env
    g = np.load(os.path.join(golden_dir, 'T1_published_train.npz'))
    n, h, w, k, dseed, wseed = [int(v) for v in g['meta']]
    b = _gpu_batch(synth.make_batch(n, h, w, k, seed=dseed))
    outs = {}
    for mode in ('fp32', 'bf16_operands', 'bf16'):
        m = _build(env, synth.PUBLISHED, wseed)
        m.compute_dtype = mode
        m.train()
        out = m.forward(image=b['image'], input_depth=b['input_depth'])
        loss, info = _loss(m, b, out)
        loss.backward()
        torch.cuda.synchronize()
        outs[mode] = (out.detach(), float(loss.detach()), m)
    e32 = _rel(outs['fp32'][0], g['output'])
    assert e32 < BAR
    for mode, bar in (('bf16_operands', 5e-2), ('bf16', 6e-2)):   # 'bf16': bf16 tensors in HBM as well (one more rounding per layer)
        e16 = _rel(outs[mode][0], g['output'])
        print('output rel err vs reference: fp32 %.2e, %s %.2e; loss %.5f / %.5f / ref %.5f' % (e32, mode, e16, outs['fp32'][1], outs[mode][1], float(g['loss'][0])))
        assert 1e-4 < e16 < bar
        assert abs(outs[mode][1] - float(g['loss'][0])) < 2e-2 * abs(float(g['loss'][0]))
        grads = dict(_named(outs[mode][2], 'p'))
        bad = 0
        for key, l2 in zip(g['grad_keys'].tolist(), g['grad_l2'].tolist()):
            got = float(grads[key].grad.double().norm())
            if abs(got - l2) > 0.25 * l2 + 1e-9:
                bad += 1
        assert bad <= len(g['grad_keys']) // 20, (mode, bad)     # gradient norms within 25 % for (nearly) every parameter


@pytest.mark.gpu
def test_hipgraph_captured_inference_matches_eager_and_golden(env, golden_dir):
    '''FusionNetModel.capture_inference (BASELINE.json config 5: hipGraph-captured inference): replays are bit-identical to the
    eager eval-mode forward, for the captured inputs and for new ones, follow a parameter update, and match the T3 fixture.'''
    synth, _ = env
    g = np.load(os.path.join(golden_dir, 'T3_eval.npz'))
    n, h, w, k, dseed, wseed = [int(v) for v in g['published_meta']]
    m = _build(env, synth.PUBLISHED, wseed)
    with pytest.raises(Exception):
        m.capture_inference(torch.zeros(n, 3, h, w, device='cuda'), torch.zeros(n, 2, h, w, device='cuda'))   # still in train mode
    m.eval()
    b = _gpu_batch(synth.make_batch(n, h, w, k, seed=dseed))
    b2 = _gpu_batch(synth.make_batch(n, h, w, k, seed=dseed + 1))
    run = m.capture_inference(b['image'], b['input_depth'])
    with torch.no_grad():
        for bb in (b, b2, b):
            got = run(bb['image'], bb['input_depth']).clone()
            ref = m.forward(image=bb['image'], input_depth=bb['input_depth'])
            torch.cuda.synchronize()
            assert torch.equal(got, ref)
        assert _rel(run(b['image'], b['input_depth']), g['published_output']) < BAR
        # the packed weights are rebuilt inside the graph: a replay sees updated parameters
        m.decoder.output0.conv.weight.data.mul_(1.5)
        got = run(b['image'], b['input_depth']).clone()
        ref = m.forward(image=b['image'], input_depth=b['input_depth'])
        torch.cuda.synchronize()
        assert torch.equal(got, ref)
        with pytest.raises(Exception):
            run(b['image'][:, :, :-2], b['input_depth'][:, :, :-2])


def test_inference_with_folded_batchnorm_matches_unfused_eval_path(env, golden_dir):
    '''Engine.fuse_eval (BatchNorm folded into the conv weights, bias + LeakyReLU + residual in the conv epilogue) against the
    unfused eval path (conv -> BN/activation pass) and the T3 fixture; the fused path is really taken (bits differ).'''
    if os.environ.get('RCF_CONV_SPLIT') == '0':
        pytest.skip('the inference epilogue lives in the split kernels, which RCF_CONV_SPLIT=0 turns off')
    synth, _ = env
    g = np.load(os.path.join(golden_dir, 'T3_eval.npz'))
    for tag, cfg in (('tiny', synth.TINY), ('published', synth.PUBLISHED)):
        n, h, w, k, dseed, wseed = [int(v) for v in g[tag + '_meta']]
        m = _build(env, cfg, wseed)
        m.eval()
        b = _gpu_batch(synth.make_batch(n, h, w, k, seed=dseed))
        outs = {}
        with torch.no_grad():
            for fused in (True, False):
                m._engine.fuse_eval = fused
                outs[fused] = m.forward(image=b['image'], input_depth=b['input_depth']).clone()
        torch.cuda.synchronize()
        assert _rel(outs[True], g[tag + '_output']) < BAR and _rel(outs[False], g[tag + '_output']) < BAR
        assert _rel(outs[True], outs[False]) < 2e-5
        if tag == 'published':
            assert not torch.equal(outs[True], outs[False])


def test_hipgraph_captured_training_step_is_bitwise_the_eager_step(env, golden_dir):
    '''FusionNetModel.capture_training_step: forward + outlier removal + loss + backward + Adam recorded into ONE hipGraph.  Three
    replays on the three batches of fixture T2 against three eager steps from the same start: losses, parameters, Adam moments,
    BatchNorm running statistics and step counts are bitwise equal; the trajectory matches the reference's (T2); capturing itself
    leaves the training state untouched.'''
    from rcf_amd.net_utils import OutlierRemoval
    synth, train = env
    g = np.load(os.path.join(golden_dir, 'T2_tiny_adam3.npz'))
    n, h, w, k, dseed, wseed = [int(v) for v in g['meta']]
    batches = [_gpu_batch(synth.make_batch(n, h, w, k, seed=dseed + i)) for i in range(3)]
    runs = {}
    for mode in ('eager', 'graph'):
        m = _build(env, synth.TINY, wseed)
        opt = train.make_optimizer(m, lr=1e-3)
        m.train()
        before = m._param_arena.clone()
        if mode == 'graph':
            b0 = batches[0]
            step = m.capture_training_step(opt, b0['image'], b0['input_depth'], b0['ground_truth'], b0['lidar_map'])
            assert torch.equal(m._param_arena, before), 'capturing changed the parameters'
            assert len(opt._shared_steps) == 1 and float(list(opt._shared_steps.values())[0]) == 0.0
        losses = []
        for b in batches:
            if mode == 'graph':
                loss = step(b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])
            else:
                loss = train.train_step(m, opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])[0]
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        moments = list(opt._moment_arenas.values())[0]
        runs[mode] = (losses, m._param_arena.clone(), moments[0].clone(), moments[1].clone(),
                      torch.cat([b.reshape(-1).float() for b in m.encoder.buffers()] + [b.reshape(-1).float() for b in m.decoder.buffers()]),
                      float(list(opt._shared_steps.values())[0]), opt.state_dict()['state'][0]['step'])
    e, gr = runs['eager'], runs['graph']
    assert e[0] == gr[0], (e[0], gr[0])
    for i in range(1, 5):
        assert torch.equal(e[i], gr[i]), i
    assert e[5] == gr[5] == 3.0 and float(gr[6]) == 3.0
    np.testing.assert_allclose(gr[0], g['losses'], rtol=BAR)      # and it is the reference's trajectory (without outlier removal)


def test_captured_training_step_with_the_maxima_arena_overflowing_mid_step(env, golden_dir):
    '''The per-step arena of operand maxima (two-plane fp16 arithmetic) is zero-filled in chunks; a chunk that runs out in the middle
    of a step is replaced by a fresh one and every engine stream is fenced behind the fill (Engine._fence_streams).  With the arena
    capped at 8 slots every step goes through that path dozens of times -- in forward, in backward, on the side streams.  Captured
    (the fence must leave the capture joinable: the streams it made wait are rejoined by side_join) and eager, against the uncapped
    eager step: bitwise the same losses and parameters.'''
    synth, train = env
    g = np.load(os.path.join(golden_dir, 'T2_tiny_adam3.npz'))
    n, h, w, k, dseed, wseed = [int(v) for v in g['meta']]
    batches = [_gpu_batch(synth.make_batch(n, h, w, k, seed=dseed + i)) for i in range(3)]
    runs = {}
    for mode in ('reference', 'eager', 'graph'):
        m = _build(env, synth.TINY, wseed)
        opt = train.make_optimizer(m, lr=1e-3)
        m.train()
        if mode != 'reference':
            m._engine.amax_arena_cap = 8
        if mode == 'graph':
            b0 = batches[0]
            step = m.capture_training_step(opt, b0['image'], b0['input_depth'], b0['ground_truth'], b0['lidar_map'])
        losses = []
        for b in batches:
            if mode == 'graph':
                loss = step(b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])
            else:
                loss = train.train_step(m, opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])[0]
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        runs[mode] = (losses, m._param_arena.clone())
    for mode in ('eager', 'graph'):
        assert runs[mode][0] == runs['reference'][0], (mode, runs[mode][0], runs['reference'][0])
        assert torch.equal(runs[mode][1], runs['reference'][1]), mode


def test_captured_training_step_follows_a_learning_rate_schedule_and_two_param_groups(env, golden_dir):
    '''The reference loop rewrites g['lr'] on its schedule (src/fusionnet_main.py:354-362).  A replayed step must see that: the
    recorded Adam launch reads its hyper-parameters from device memory, refreshed before each replay.  Also two param groups on the
    one arena (different lr): one step count per optimizer.step().  Replay == eager, bitwise, with lr changed between steps.'''
    from rcf_amd.optim import FusedAdam
    synth, train = env
    g = np.load(os.path.join(golden_dir, 'T2_tiny_adam3.npz'))
    n, h, w, k, dseed, wseed = [int(v) for v in g['meta']]
    b = _gpu_batch(synth.make_batch(n, h, w, k, seed=dseed))
    schedule = [(1e-3, 5e-4), (2e-4, 5e-4), (2e-4, 1e-4), (5e-5, 1e-4)]
    runs = {}
    for mode in ('eager', 'graph'):
        m = _build(env, synth.TINY, wseed)
        ps = m._used_params
        half = len(ps) // 2
        unused = [p for p in m.parameters() if all(p is not q for q in ps)]
        opt = FusedAdam([{'params': ps[:half], 'lr': schedule[0][0]}, {'params': ps[half:] + unused, 'lr': schedule[0][1]}])
        m.train()
        if mode == 'graph':
            step = m.capture_training_step(opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])
        losses = []
        for lr0, lr1 in schedule:
            opt.param_groups[0]['lr'], opt.param_groups[1]['lr'] = lr0, lr1
            if mode == 'graph':
                loss = step()
            else:
                loss = train.train_step(m, opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])[0]
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        steps = sorted(set(float(st['step']) for st in opt.state_dict()['state'].values()))
        runs[mode] = (losses, m._param_arena.clone(), steps)
    assert runs['eager'][2] == runs['graph'][2] == [4.0], (runs['eager'][2], runs['graph'][2])
    assert runs['eager'][0] == runs['graph'][0]
    assert torch.equal(runs['eager'][1], runs['graph'][1])
    # and the schedule mattered: a constant-lr run ends elsewhere
    m = _build(env, synth.TINY, wseed)
    opt = train.make_optimizer(m, lr=1e-3)
    m.train()
    for _ in schedule:
        train.train_step(m, opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])
    assert not torch.equal(m._param_arena, runs['graph'][1])


def test_captured_training_step_full_resolution_matches_eager(env):
    '''The benchmark's own step (published net, 900x1600, outlier removal) at batch 2: replay == eager, bitwise, after two steps.'''
    from rcf_amd.net_utils import OutlierRemoval
    synth, train = env
    b = _gpu_batch(synth.make_batch(2, 900, 1600, 64, seed=1234))
    outl = OutlierRemoval(7, 1.5)
    res = {}
    for mode in ('eager', 'graph'):
        m = _build(env, synth.PUBLISHED, 1234)
        opt = train.make_optimizer(m, lr=1e-3)
        m.train()
        if mode == 'graph':
            step = m.capture_training_step(opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'], outlier_removal=outl)
            losses = [float(step().detach()) for _ in range(2)]
        else:
            losses = [float(train.train_step(m, opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'],
                                             outlier_removal=outl)[0].detach()) for _ in range(2)]
        torch.cuda.synchronize()
        res[mode] = (losses, m._param_arena.clone())
        del m, opt
        torch.cuda.empty_cache()
    assert res['eager'][0] == res['graph'][0]
    assert torch.equal(res['eager'][1], res['graph'][1])


# ---------------------------------------------------------------- batched weight packing (engine.WeightPlan)
@pytest.mark.gpu
def test_batched_pack_and_phase_weights_equal_the_single_calls():
    '''rcf_conv2d_pack_weights_batch / rcf_phase_weights_batch against n calls of the single entry points: bitwise, for every kind of
    item a training step holds (split 3-plane / 2-plane / bf16 forward and input-gradient forms, f32-MFMA forms, stems, phases).'''
    import ctypes
    import rcf_amd  # noqa: F401
    from rcf_amd import _lib, ops
    from rcf_amd._lib import RCF_PHASE_S2_DGRAD, RCF_PHASE_UP2X_DGRAD, RCF_PHASE_UP2X_FWD
    g = torch.Generator().manual_seed(3)
    items, keep = [], []
    for prec in ('fp32', 'f16x2', 'bf16'):
        ops.set_precision(prec)
        try:
            for (k, s, c1, c2, co) in [(3, 1, 64, 0, 64), (3, 1, 64, 32, 64), (3, 1, 32, 0, 32), (1, 1, 32, 0, 64), (3, 2, 32, 0, 64), (3, 1, 256, 0, 256),
                                       (3, 1, 128, 128, 128), (1, 2, 32, 0, 64)]:
                w = (torch.rand(co, c1 + c2, k, k, generator=g) - 0.5).cuda()
                d = ops.make_fwd_desc(2, 40, 56, c1, c2, co, k, s)
                items.append((d, w))
                if s == 1 and k == 3:
                    items.append((ops.make_dgrad_desc(d, 0, c1, False), w))
            if prec != 'bf16':
                w7 = (torch.rand(32, 3, 7, 7, generator=g) - 0.5).cuda()
                items.append((ops.make_fwd_desc(2, 70, 102, 3, 0, 32, 7, 2), w7))
            wp = ops.phase_weights((torch.rand(32, 64, 3, 3, generator=g) - 0.5).cuda(), RCF_PHASE_UP2X_FWD)
            keep.append(wp)
            items.append((ops.make_up2x_fwd_desc(2, 20, 28, 64, 32, 1, 0), wp[2]))
        finally:
            ops.set_precision('fp32')
    n = len(items)
    assert n > 36                                  # more than one launch of the batched kernel
    single, batched = [], []
    arr = (_lib.PackItem * n)()
    for i, (d, w) in enumerate(items):
        qi = ops.conv_query(d)
        nf = qi.packed_weight_floats
        a = torch.full((nf,), float('nan'), device='cuda')
        b = torch.full((nf,), float('nan'), device='cuda')
        # two-plane fp16 items carry the weight tensor's maximum (every other one of them, so the null = scale-1 form is covered too)
        am = ops.amax(w) if (40000 <= qi.kernel_id < 50000 and i % 2 == 0) else None
        keep.append(am)
        ops.conv_pack(d, w, a, am)
        single.append(a); batched.append(b)
        arr[i].desc = ctypes.pointer(d)
        arr[i].w_oihw = w.data_ptr()
        arr[i].packed = b.data_ptr()
        arr[i].amax_w = None if am is None else am.data_ptr()
    ops.conv_pack_batch(arr, n)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(single, batched)):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), i
    # phase weights
    ws = [(torch.rand(o, i, 3, 3, generator=g) - 0.5).cuda() for (o, i) in [(32, 64), (64, 64), (128, 64), (4, 8)]]
    modes = [RCF_PHASE_UP2X_FWD, RCF_PHASE_UP2X_DGRAD, RCF_PHASE_S2_DGRAD]
    parr = (_lib.PhaseItem * (len(ws) * len(modes)))()
    outs, refs = [], []
    for wi, w in enumerate(ws):
        for mi, m in enumerate(modes):
            r = ops.phase_weights(w, m)
            o = torch.full_like(r, float('nan'))
            it = parr[wi * len(modes) + mi]
            it.w_oihw, it.out, it.o, it.i, it.mode = w.data_ptr(), o.data_ptr(), w.shape[0], w.shape[1], m
            outs.append(o); refs.append(r)
    ops.phase_weights_batch(parr, len(outs))
    torch.cuda.synchronize()
    for o, r in zip(outs, refs):
        assert torch.equal(o, r)
    # invalid items are refused before anything is launched
    arr[0].packed = None
    with pytest.raises(_lib.RcfError):
        ops.conv_pack_batch(arr, n)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['fp32', 'f16x2', 'bf16'])
def test_weight_plan_training_is_bitwise_the_unbatched_training(env, golden_dir, mode):
    '''Three Adam steps with the step's weight transforms batched up front (engine.WeightPlan, default) against the same steps with
    one launch per transform (batch_weight_packing = False): parameters, loss and BatchNorm buffers bitwise equal; the plan replays
    (>= 60 recorded requests); a change of input size falls back and re-records without changing results.'''
    synth, train = env
    g = np.load(os.path.join(golden_dir, 'T2_tiny_adam3.npz'))
    n, h, w, k, dseed, wseed = [int(v) for v in g['meta']]
    res = {}
    for batched in (True, False):
        m = _build(env, synth.PUBLISHED, wseed)
        m.batch_weight_packing = batched
        m.compute_dtype = mode
        m.train()
        opt = train.make_optimizer(m, lr=1e-3)
        losses = []
        sizes = [(h, w), (h, w), (h, w), (h + 16, w), (h + 16, w), (h, w)]
        for step, (hh, ww) in enumerate(sizes):
            b = _gpu_batch(synth.make_batch(n, hh, ww, k, seed=dseed + step))
            losses.append(float(train.train_step(m, opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])[0].detach()))
            if batched and step == 2:
                assert m._engine.plan.state == 'replay' and not m._engine.plan.dirty and len(m._engine.plan.entries) >= 60
            if batched and step == 3:
                assert m._engine.plan.dirty or m._engine.plan.state == 'record' or m._engine.plan.pos == len(m._engine.plan.entries)
        if not batched:
            assert m._engine.plan.state == 'off'
        torch.cuda.synchronize()
        res[batched] = (losses, m._param_arena.clone(), [b.clone() for _, b in _named(m, 'b')])
    assert res[True][0] == res[False][0], (res[True][0], res[False][0])
    assert torch.equal(res[True][1], res[False][1])
    for a, b in zip(res[True][2], res[False][2]):
        assert torch.equal(a, b)
