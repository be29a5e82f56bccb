'''
FusedAdam: torch.optim.Adam semantics (the reference's optimizer, src/fusionnet_main.py:307-312) executed as ONE
HIP kernel launch over the model's flat parameter arena (rcf_adam_step) instead of ~270 per-tensor update chains.
Same param_group keys and per-parameter state names ('step', 'exp_avg', 'exp_avg_sq') as torch.optim.Adam, so
optimizer state dicts interchange with the reference's checkpoints.
'''

import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError('Invalid Adam hyper-parameter')
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None,
                        decoupled_weight_decay=False)
        super(FusedAdam, self).__init__(params, defaults)
        self._moment_arenas = {}   # id(param arena) -> (exp_avg arena, exp_avg_sq arena)
        # flat path: step count + hyper-parameters in device memory (rcf_adam_step_dev), so that a hipGraph-captured training step
        # performs a correct update on every replay.  id(param arena) -> [device float32[8], host copy of its hyper-parameters]
        self._dev_state = {}
        self._shared_steps = {}    # id(param arena) -> the host-side step tensor its parameters' states share

    def _init_state(self, p):
        state = self.state[p]
        if len(state) != 0:
            return state
        arena = getattr(p, '_rcf_arena', None)
        state['step'] = torch.tensor(0.0, dtype=torch.float32)
        if arena is not None:
            parena, _, off = arena
            key = id(parena)
            # parameters of one arena step together: ONE host-side step tensor shared by their states (one increment per step
            # instead of ~270; state_dict() still reports it per parameter, like torch.optim.Adam)
            state['step'] = self._shared_steps.setdefault(key, state['step'])
            if key not in self._moment_arenas:
                self._moment_arenas[key] = (torch.zeros_like(parena), torch.zeros_like(parena))
            m, v = self._moment_arenas[key]
            state['exp_avg'] = m[off:off + p.numel()].view(p.shape)
            state['exp_avg_sq'] = v[off:off + p.numel()].view(p.shape)
        else:
            state['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return state

    def load_state_dict(self, state_dict):
        super(FusedAdam, self).load_state_dict(state_dict)
        # re-home the loaded moments into the flat arenas so the single-launch path stays valid
        for group in self.param_groups:
            for p in group['params']:
                st = self.state.get(p)
                arena = getattr(p, '_rcf_arena', None)
                if not st or arena is None:
                    continue
                parena, _, off = arena
                key = id(parena)
                if key not in self._moment_arenas:
                    self._moment_arenas[key] = (torch.zeros_like(parena), torch.zeros_like(parena))
                m, v = self._moment_arenas[key]
                for name, ar in (('exp_avg', m), ('exp_avg_sq', v)):
                    view = ar[off:off + p.numel()].view(p.shape)
                    view.copy_(st[name])
                    st[name] = view
                shared = self._shared_steps.setdefault(key, torch.tensor(0.0, dtype=torch.float32))
                shared.fill_(float(st['step']))
                st['step'] = shared
        # the device-side counters and hyper-parameters of the launched slices follow the loaded state (a captured step replayed after
        # a resume reads THEM): reseed the step count, and mark the hyper-parameters unknown so the next sync / step copies them
        for (pid, _, _), ds in self._dev_state.items():
            shared = self._shared_steps.get(pid)
            if shared is not None:
                ds[0][0:1].fill_(float(shared))
                ds[2] = int(float(shared))
            ds[1] = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        bumped = set()   # ONE increment per optimizer.step() for a step tensor, however many param groups share its arena
        for group in self.param_groups:
            params = [p for p in group['params'] if p.grad is not None]
            if not params:
                continue
            if group.get('amsgrad') or group.get('maximize'):
                raise ValueError('FusedAdam: amsgrad / maximize are not implemented')
            beta1, beta2 = group['betas']
            states = [self._init_state(p) for p in params]
            for st in states:
                if id(st['step']) not in bumped:     # states of one arena share their step tensor
                    bumped.add(id(st['step']))
                    st['step'] += 1
            if not self._flat_step(group, params, states, beta1, beta2):
                for p, st in zip(params, states):
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    if not p.data.is_contiguous() or not st['exp_avg'].is_contiguous():
                        raise RuntimeError('FusedAdam needs contiguous parameters')
                    ops.adam_step(p.data.view(-1), g.view(-1), st['exp_avg'].view(-1), st['exp_avg_sq'].view(-1),
                                  group['lr'], beta1, beta2, group['eps'], group['weight_decay'], int(st['step']))
        return loss

    def _flat_step(self, group, params, states, beta1, beta2):
        '''One launch over [lo, hi) of the arena when the parameters with gradients tile it exactly.'''
        first = getattr(params[0], '_rcf_arena', None)
        if first is None:
            return False
        parena, garena, _ = first
        step0 = int(states[0]['step'])
        lo, hi, total = None, None, 0
        for p, st in zip(params, states):
            a = getattr(p, '_rcf_arena', None)
            if a is None or a[0] is not parena or int(st['step']) != step0:
                return False
            off, n = a[2], p.numel()
            if p.grad.data_ptr() != garena.data_ptr() + 4 * off:
                return False
            if p.data.data_ptr() != parena.data_ptr() + 4 * off:
                return False
            m, v = self._moment_arenas.get(id(parena), (None, None))
            if m is None or st['exp_avg'].data_ptr() != m.data_ptr() + 4 * off:
                return False
            lo = off if lo is None else min(lo, off)
            hi = off + n if hi is None else max(hi, off + n)
            total += n
        if hi - lo != total:
            return False
        m, v = self._moment_arenas[id(parena)]
        hyper = (float(group['lr']), float(beta1), float(beta2), float(group['eps']), float(group['weight_decay']))
        # one device counter per launched slice: two param groups of one arena are two launches, each ticking its own counter
        dkey = (id(parena), lo, hi)
        ds = self._dev_state.get(dkey)
        if ds is None or ds[2] != step0 - 1:
            # first use, or the host-side step count moved on its own (load_state_dict): (re)seed the device counter
            dev = torch.tensor([float(step0 - 1)] + list(hyper) + [0.0, 0.0], dtype=torch.float32).to(parena.device)
            ds = [dev, hyper, step0 - 1, next(i for i, g in enumerate(self.param_groups) if g is group)]
            self._dev_state[dkey] = ds
        elif ds[1] != hyper:
            ds[0][1:6].copy_(torch.tensor(hyper, dtype=torch.float32))   # learning-rate schedule etc. (never inside a capture)
            ds[1] = hyper
        ops.adam_step_dev(parena[lo:hi], garena[lo:hi], m[lo:hi], v[lo:hi], ds[0])
        ds[2] = step0
        return True

    def sync_hyper_parameters(self):
        '''Before a hipGraph replay of a captured step(): the recorded Adam launch reads lr / betas / eps / weight_decay from device
        memory, so a change made on the host since the last step (the reference's learning-rate schedule rewrites g['lr'],
        src/fusionnet_main.py:354-362) is copied there now -- outside the graph, on the replay's stream.'''
        for ds in self._dev_state.values():
            # the group by INDEX: torch.optim.Optimizer.load_state_dict replaces self.param_groups with new dictionaries, and a
            # learning-rate schedule after a resume writes into those
            group = self.param_groups[ds[3]]
            hyper = (float(group['lr']), float(group['betas'][0]), float(group['betas'][1]), float(group['eps']),
                     float(group['weight_decay']))
            if ds[1] != hyper:
                ds[0][1:6].copy_(torch.tensor(hyper, dtype=torch.float32))
                ds[1] = hyper

    def note_replayed_step(self):
        '''A hipGraph replay of a captured step() has advanced the device-side step count: advance the host-side mirror (the
        per-parameter 'step' entries that state_dict() reports) without launching anything.'''
        for shared in self._shared_steps.values():
            shared += 1
        for ds in self._dev_state.values():
            ds[2] += 1
