'''
Data parallelism for the FusionNet training step: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

The reference only has single-process torch.nn.DataParallel (src/fusionnet_model.py:395-401), which gathers the
whole batch's outputs on GPU 0, computes ONE masked-mean loss there (src/fusionnet_main.py:385) and reduce-adds
the replicas' gradients.  The equivalent here:
  * every rank computes the local loss sums and valid-pixel counts; the 4 scalars are all-reduced BEFORE backward,
    so each rank's gradient is already normalised by the GLOBAL counts;
  * parameter gradients are all-reduced with SUM (no 1/world factor) in a few contiguous buckets of the flat
    gradient arena.  The arena is laid out in the order gradients become final (decoder first), so a bucket is
    launched on RCCL's stream as soon as the tape has produced its last gradient, overlapping the rest of backward.
  * BatchNorm statistics stay per replica, as in nn.DataParallel; parameters that never get a gradient
    (10 unused projections) sit outside every bucket.
'''

import torch
import torch.distributed as dist


class GradientBuckets(object):
    def __init__(self, model, n_buckets=6, group=None):
        self.n_buckets = n_buckets
        self.group = group
        self.handles = []
        self.capture = None       # a SegmentedCapture while FusionNetModel.capture_training_step records a data-parallel step
        self.rebuild(model)

    def rebuild(self, model):
        used = model._used_params
        total = model._n_used
        target = max(1, (total + self.n_buckets - 1) // self.n_buckets)
        self.garena = model._grad_arena
        self.bounds = []          # (lo, hi) element ranges of the gradient arena
        self.bucket_of = {}
        self.members = []
        lo, off, cur = 0, 0, []
        for p in used:
            cur.append(p)
            off += p.numel()
            if off - lo >= target:
                self._close(lo, off, cur)
                lo, cur = off, []
        if cur:
            self._close(lo, off, cur)
        assert off == total

    def _close(self, lo, hi, params):
        b = len(self.bounds)
        self.bounds.append((lo, hi))
        self.members.append(len(params))
        for p in params:
            self.bucket_of[id(p)] = b

    def begin_backward(self):
        self.remaining = list(self.members)
        self.handles = []

    def completes_bucket(self, p):
        '''True when on_param_grad(p) will launch its bucket's exchange (the engine joins its side stream first).'''
        b = self.bucket_of.get(id(p))
        return b is not None and self.remaining[b] == 1

    def on_param_grad(self, p):
        b = self.bucket_of.get(id(p))
        if b is None:
            return
        self.remaining[b] -= 1
        if self.remaining[b] == 0:
            lo, hi = self.bounds[b]
            if self.capture is not None:   # recording a segmented step: the exchange becomes a cut between two graph segments
                self.capture.cut(('bucket', lo, hi))
                return
            self.handles.append(dist.all_reduce(self.garena[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish_backward(self):
        if any(r != 0 for r in self.remaining):
            raise RuntimeError('gradient buckets incomplete after backward: %s' % self.remaining)
        if self.capture is not None:
            self.capture.cut(('wait',))
            return
        for h in self.handles:
            h.wait()
        self.handles = []

    def all_reduce_sums(self, sums):
        if self.capture is not None:
            self.capture.cut(('sums', sums))
            return
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group)

    # ---- replay of a segmented capture (SegmentedCapture below): the collectives between the graph segments, launched eagerly
    def run_exchange(self, op, handles):
        if op[0] == 'sums':
            dist.all_reduce(op[1], op=dist.ReduceOp.SUM, group=self.group)
        elif op[0] == 'bucket':
            handles.append(dist.all_reduce(self.garena[op[1]:op[2]], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:   # 'wait': every bucket is back before the optimizer reads the gradients
            for h in handles:
                h.wait()
            del handles[:]


class SegmentedCapture(object):
    '''
    A data-parallel training step as hipGraph SEGMENTS: the compute between two exchange points (the loss sums before backward, each
    gradient bucket, the final wait) is recorded into its own graph -- all graphs in one memory pool, so a tensor allocated in one
    segment is the same storage in the next -- and a replay launches  segment, collective, segment, collective, ...  : the RCCL calls
    stay ordinary eager launches on RCCL's stream (overlapping the next segment exactly as in the eager step), the ~1100 kernel
    launches between them cost one hipGraphLaunch per segment.  Same kernels in the same order as the eager data-parallel step:
    bitwise the same result (tests/test_configs_gpu.py).
    '''

    def __init__(self):
        self.pool = torch.cuda.graph_pool_handle()
        self.segments = []      # [graph, [exchange ops launched after it]]
        self.graph = None

    def begin(self):
        self.graph = torch.cuda.CUDAGraph()
        # 'relaxed': the cuts inside backward() happen on autograd's worker thread, the first begin and the last end on the caller's --
        # the other capture modes tie a capture sequence to the thread that began it
        self.graph.capture_begin(pool=self.pool, capture_error_mode='relaxed')

    def cut(self, op):
        # A cut ALWAYS ends the segment.  (Round 4 skipped the cut when no C-ABI launch had been counted since the last one; torch's own
        # launches -- zero_(), clone(), the loss arithmetic -- are invisible to that counter and would have slipped behind the exchange.
        # An empty segment costs one graph launch of nothing.)
        self.graph.capture_end()
        self.segments.append([self.graph, [op]])
        self.begin()

    def end(self):
        self.graph.capture_end()
        self.segments.append([self.graph, []])
        self.graph = None

    def replay(self, buckets):
        handles = []
        for graph, ops_after in self.segments:
            graph.replay()
            for op in ops_after:
                buckets.run_exchange(op, handles)


def pin_rank_to_gpu_numa_node(local_rank=None, local_world=None):
    '''One process per GPU: keep this rank's host threads (the Python thread that enqueues ~1,100 launches per step, RCCL's proxy
    threads) on the CPUs of the NUMA node its GPU hangs off -- eight ranks on a two-socket host otherwise migrate across sockets and
    enqueue through the inter-socket link.  Call BEFORE anything touches the GPU.  Reads sysfs only (no HIP call, no subprocess):
    the DRM render nodes in PCI-bus order are the HIP device order on one node.  Returns a dict describing what was done; never raises
    (an unknown topology leaves the affinity alone).'''
    import glob
    import os
    info = {'pinned': False}
    try:
        if local_rank is None:
            local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        if local_world is None:
            local_world = int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1')))
        if local_world <= 1 or os.environ.get('RCF_RANK_AFFINITY', '1') == '0' or not hasattr(os, 'sched_setaffinity'):
            info['why'] = 'single rank or switched off'
            return info
        cards = []
        for dev in glob.glob('/sys/class/drm/renderD*/device'):
            real = os.path.realpath(dev)
            try:
                vendor = open(os.path.join(real, 'vendor')).read().strip()
                node = int(open(os.path.join(real, 'numa_node')).read().strip())
            except (OSError, ValueError):
                continue
            if vendor == '0x1002':
                cards.append((os.path.basename(real), node))      # PCI address (sorts in bus order), NUMA node
        cards.sort()
        if len(cards) < local_world or local_rank >= len(cards):
            info['why'] = '%d AMD render nodes for %d local ranks' % (len(cards), local_world)
            return info
        node = cards[local_rank][1]
        if node < 0:
            info['why'] = 'the GPU reports no NUMA node'
            return info
        cpus = set()
        for part in open('/sys/devices/system/node/node%d/cpulist' % node).read().strip().split(','):
            lo, _, hi = part.partition('-')
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= set(os.sched_getaffinity(0))
        if not cpus:
            info['why'] = 'NUMA node %d has no CPU this process may use' % node
            return info
        os.sched_setaffinity(0, cpus)
        info.update({'pinned': True, 'numa_node': node, 'n_cpus': len(cpus), 'pci': cards[local_rank][0]})
    except Exception as e:      # a topology this code does not understand must never stop a run
        info['why'] = 'not pinned: %s' % str(e)[:80]
    return info


def init_from_env(backend=None):
    '''Rendezvous from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run); returns (rank, world, local_rank).'''
    import os
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('RCF_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)   # one process per GPU; RCCL over xGMI
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank
