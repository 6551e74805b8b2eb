"""Data-parallel glue: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm).

The reference has no distributed code; the defined semantics are standard DDP: every rank holds a full
replica, the batch is split contiguously, BatchNorm statistics stay per rank, and the gradient is averaged once per
step. Because gradients already live in ONE flat arena, the exchange is a handful of large all-reduces over
contiguous slices (bucketed so that each message is big enough to run at link rate and small enough to pipeline),
issued on a side HIP stream; the optimiser kernel applies the 1/world scale.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun). Returns (rank, world, local)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    force = os.environ.get('LVAE_FORCE_DIST') == '1'  # rehearse the collective path with a single rank
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_batch(global_batch, rank, world):
    """Contiguous shard [lo, hi) of a global batch for `rank`; requires an even split (drop_last semantics)."""
    if global_batch % world != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


def bucket_slices(n, bucket_elems):
    """[(lo, hi)] covering [0, n) in chunks of at most bucket_elems."""
    out, lo = [], 0
    while lo < n:
        hi = min(n, lo + bucket_elems)
        out.append((lo, hi))
        lo = hi
    return out


def make_buckets(segments, bucket_elems):
    """Merge consecutive gradient segments (arena.segments, in completion order) into buckets of at least `bucket_elems` elements.
    A segment is (lo, hi) or (segment id, lo, hi): the id is the index the model's segment markers report (LadderVAE.grad_segments();
    a segment without trainable parameters has no entry, so ids may have gaps); plain pairs are numbered by position.
    Returns [(lo, hi, last_segment_id)]: bucket k is complete when backward has left segment `last_segment_id`."""
    out, lo = [], None
    for i, seg in enumerate(segments):
        sid, a, b = seg if len(seg) == 3 else (i,) + tuple(seg)
        if lo is None:
            lo = a
        if b - lo >= bucket_elems or i + 1 == len(segments):
            out.append((lo, b, sid))
            lo = None
    return out


class GradAllReduce:
    """SUM all-reduce of the flat gradient arena, `scale` = 1/world for the optimiser.

    The arena is laid out in gradient-completion order (arena.py), so a bucket is a contiguous slice that is final as soon as
    backward has left its last segment. `segment_done(i)` — called from the backward of the model's segment markers
    (ops.SegmentMarkFn) — forks the side stream off the launch stream at that point and enqueues the bucket's all-reduce
    there, while backward keeps issuing kernels on the launch stream; `finish()` enqueues what is left (the last bucket: stem)
    and joins the side stream back, after which the optimiser may run. Inside a hipGraph capture the fork/join become graph
    edges and the RCCL kernels graph nodes, so the whole step (backward, exchange, Adamax) replays as one graph.

    `run()` is the non-overlapped form (everything after backward), kept for LVAE_DDP_MODE=split and for CPU/gloo tensors.
    """

    def __init__(self, flat_grads, group=None, bucket_mb=None, segments=None, mode=None, comm=None):
        # Which form the exchange takes. On the GPU with a capturable backend (RCCL) the default is 'split': the whole arena as ONE message
        # after backward, between the fwd+bwd graph and the Adamax graph. Measured on MI355X (tools/forced_ab.sh): a
        # side branch of a captured graph does not run beside the main branch on this platform — every bucket with real work on it added its
        # full duration (+0.55 ms per 8 MB bucket in the one-rank rehearsal, +4.3 ms for 7 buckets) on top of +0.86 ms for the fork / join
        # edges alone, while the split form costs +0.07 ms. LVAE_DDP_MODE=overlap selects the in-graph overlapped form (completion-ordered
        # buckets on the side stream during backward); eager launches (gloo, CPU) keep it as their default: eager streams do overlap.
        # Neither default is taken on faith when there is more than one rank: engine.AutoExchangeStep builds BOTH forms (mode= given
        # explicitly, one shared communicator) and keeps the one that measures faster on the real world size (round 5).
        on_gpu_capturable = flat_grads.is_cuda and dist.is_initialized() and dist.get_backend(group) == 'nccl'
        self.mode = mode if mode is not None else os.environ.get('LVAE_DDP_MODE', 'split' if on_gpu_capturable else 'overlap')
        if self.mode not in ('split', 'overlap'):
            raise ValueError("gradient exchange form %r: 'split' or 'overlap' ('auto' is engine.AutoExchangeStep's)" % (self.mode,))
        if bucket_mb is None:
            # overlap: 8 MB buckets (xGMI rings are per-link bound; the count was free in round 2's sweep); split: one message
            bucket_mb = float(os.environ.get('LVAE_BUCKET_MB', '8' if self.mode != 'split' else '1048576'))
        self.flat = flat_grads
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = os.environ.get('LVAE_FORCE_DIST') == '1' and dist.is_initialized()
        elems = max(1, int(bucket_mb * (1 << 20) / 4))
        n = flat_grads.numel()
        if segments:
            rng = [tuple(sg[-2:]) for sg in segments]
            assert rng[0][0] == 0 and rng[-1][1] == n and all(a[1] == b[0] for a, b in zip(rng, rng[1:]))
            self.buckets = make_buckets(segments, elems)
        else:
            self.buckets = [(lo, hi, None) for lo, hi in bucket_slices(n, elems)]
        self.by_segment = {last: k for k, (_, _, last) in enumerate(self.buckets) if last is not None}
        # a marker of a segment that owns no bucket end must still flush buckets that ended at an earlier id it skipped over
        self._ends = sorted(self.by_segment)
        self.on_gpu = flat_grads.is_cuda
        # RCCL collectives are plain kernel launches on the stream and can be captured into a hipGraph; gloo cannot
        self.capturable = dist.is_initialized() and dist.get_backend(group) == 'nccl'
        self.stream = torch.cuda.Stream(device=flat_grads.device) if self.on_gpu else None
        self.scale = torch.full((1,), 1.0 / self.world, dtype=torch.float32, device=flat_grads.device)
        self.overlap = segments is not None and self.mode != 'split'
        self.next_bucket = 0
        self.launched = []          # bucket indices in launch order of the current step (tests look at it)
        # Buckets on the GPU go through a communicator of our own (rccl.Comm): its ncclAllReduce is a plain launch on our side stream,
        # with no ProcessGroupNCCL work objects, events or watchdog behind it — which is what makes it safe to capture (rccl.py).
        self.comm, self.comm_error = comm, None
        self.owns_comm = comm is None
        if comm is None and self.on_gpu and self.capturable and (self.world > 1 or self.force):
            try:
                from . import rccl
                self.comm = rccl.Comm(group)
            except Exception as e:  # noqa: BLE001  (no private communicator: the process group's collectives, never captured)
                self.comm_error = '%s: %s' % (type(e).__name__, str(e).splitlines()[0] if str(e) else '')
            # ncclCommInitRank is a collective and may fail on ONE rank: every rank learns whether ALL have a communicator before any of
            # them issues a collective on it (a rank without one would otherwise skip calls its peers block in)
            have = torch.tensor([1.0 if self.comm is not None else 0.0], device=flat_grads.device)
            dist.all_reduce(have, op=dist.ReduceOp.MIN, group=group)
            if float(have.item()) == 0.0 and self.comm is not None:
                self.comm.destroy()
                self.comm, self.comm_error = None, 'another rank has no private communicator'
        # one-rank rehearsal (LVAE_FORCE_DIST=1): the out-of-place form, so that the captured exchange holds real RCCL / copy nodes
        self.scratch = None
        if self.comm is not None and self.world == 1 and os.environ.get('LVAE_FORCE_INPLACE') != '1':   # (diagnosis: the in-place no-op form)
            self.scratch = torch.empty(max(hi - lo for lo, hi, _ in self.buckets), dtype=torch.float32, device=flat_grads.device)

    def capture_probe(self):
        """Can a collective on the side stream be captured into a hipGraph here? Captures (and replays once) a throw-away graph holding one
        small all-reduce between a fork and a join of the side stream — before anything of the training step is captured, so that a refusal
        costs nothing but this probe (unwinding a half-captured training step turned out to be unsafe: round 3 measured a GPU memory fault
        on that path). Every rank runs the probe (it contains a collective); the verdict is the minimum over ranks. Returns (ok, reason)."""
        if not (self.on_gpu and self.capturable):
            return False, 'backend cannot be captured'
        if self.comm is None:
            return False, 'no private RCCL communicator (%s): the process group\'s collectives are not captured' % (self.comm_error or 'not created')
        dev = self.flat.device
        t = torch.ones(256, dtype=torch.float32, device=dev)
        ok, reason = True, ''
        self._all_reduce(t)   # eager first: everything lazy inside the communicator happens before anything is captured
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                cur = torch.cuda.current_stream(dev)
                try:
                    self.comm.enqueue(t, cur, self.stream, self.scratch)    # the library forks the side stream off the capturing one
                except Exception as e:  # noqa: BLE001  (join the fork before leaving the capture, then report)
                    ok, reason = False, '%s: %s' % (type(e).__name__, str(e).splitlines()[0] if str(e) else '')
                self.comm.wait(cur, self.stream)
        except Exception as e:  # noqa: BLE001
            ok, reason = False, '%s: %s' % (type(e).__name__, str(e).splitlines()[0] if str(e) else '')
        # the graph holds a collective: it is replayed only when EVERY rank captured it (a rank that did not would leave its peers
        # waiting inside the replay); the vote is an eager collective of the process group
        flag = torch.tensor([1.0 if ok else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if ok and float(flag.item()) == 0.0:
            ok, reason = False, 'another rank could not capture the collective'
        if ok:
            before = float(t[0].item())
            g.replay()
            torch.cuda.synchronize(dev)
            after = float(t[0].item())
            self.probe_nodes_ran = after == before * self.world   # the replayed SUM multiplied the (already reduced) ones by the world size
            if self.world == 1 and self.scratch is not None:
                # one rank: the sum is the identity, so the only evidence that the captured branch holds work is that it wrote the scratch
                self.scratch[:256].zero_()
                g.replay()
                torch.cuda.synchronize(dev)
                self.probe_nodes_ran = bool((self.scratch[:256] == t).all().item())
                if not self.probe_nodes_ran:
                    ok, reason = False, 'the captured exchange replayed without doing any work (empty graph branch)'
        torch.cuda.synchronize(dev)
        del g
        return ok, reason

    @property
    def active(self):
        return (self.world > 1 or self.force) and os.environ.get('LVAE_SKIP_ALLREDUCE') != '1'   # second: profiling only

    def begin_step(self):
        self.next_bucket = 0
        self.launched = []

    def _all_reduce(self, t):
        if self.comm is not None:
            self.comm.all_reduce_(t, scratch=self.scratch)   # on torch's current stream, no fork (eager uses)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def close(self):
        """Release the private communicator (teardown; the object is unusable for GPU exchanges afterwards)."""
        if self.comm is not None:
            if self.owns_comm:
                self.comm.destroy()
            self.comm = None

    def _reduce(self, k):
        lo, hi, _ = self.buckets[k]
        self._all_reduce(self.flat[lo:hi])
        self.launched.append(k)

    def _launch_through(self, k):
        """Enqueue buckets next_bucket..k (in order: every rank issues the same sequence of collectives)."""
        if k < self.next_bucket:
            return
        if not self.on_gpu:
            for b in range(self.next_bucket, k + 1):
                self._reduce(b)
        else:
            cur = torch.cuda.current_stream(self.flat.device)
            from . import ops
            for st in (ops._side.get('stream') or ()):
                self.stream.wait_stream(st)        # weight-gradient kernels issued on side streams (async_wgrad) write this bucket too
            if self.comm is not None:
                for b in range(self.next_bucket, k + 1):
                    lo, hi, _ = self.buckets[b]
                    # lvae_allreduce_enqueue: fork (everything issued on the launch stream so far, this bucket's last gradient kernel
                    # included) + ncclAllReduce on the side stream
                    self.comm.enqueue(self.flat[lo:hi], cur, self.stream, self.scratch)
                    self.launched.append(b)
            else:
                self.stream.wait_stream(cur)
                with torch.cuda.stream(self.stream):
                    for b in range(self.next_bucket, k + 1):
                        self._reduce(b)
        self.next_bucket = k + 1

    def segment_done(self, seg):
        if not (self.active and self.overlap):
            return
        # every bucket whose last segment id is <= seg is complete (ids are in completion order)
        import bisect
        i = bisect.bisect_right(self._ends, seg)
        if i > 0:
            self._launch_through(self.by_segment[self._ends[i - 1]])

    def finish(self):
        """After backward: exchange whatever has not been enqueued yet and make the launch stream wait for all of it."""
        if not self.active:
            return
        self._launch_through(len(self.buckets) - 1)
        if self.on_gpu:
            cur = torch.cuda.current_stream(self.flat.device)
            if self.comm is not None:
                self.comm.wait(cur, self.stream)   # lvae_allreduce_wait: join
            else:
                cur.wait_stream(self.stream)

    def run(self):
        """Reduce all buckets after backward; the caller's current stream waits for completion (no host sync)."""
        self.begin_step()
        self.finish()


class FormSelector:
    """Which of several equivalent forms of a step is fastest HERE (this world size, this fabric): every rank times `trial_steps` steps
    of each form; a form's cost is the MAX over ranks of its mean step time (the step ends when the slowest rank ends); the decision is
    taken on those reduced numbers, so every rank takes the same branch. Pure bookkeeping + one small all-reduce: tests/test_dist_cpu.py
    runs it over gloo with two ranks."""

    def __init__(self, forms, trial_steps=3, group=None, device='cpu'):
        self.forms = list(forms)
        self.trial_steps = max(1, int(trial_steps))
        self.group, self.device = group, device
        self.samples = {f: [] for f in self.forms}
        self.timings_ms = None   # {form: ms per step, max over ranks} once decided
        self.chosen = self.forms[0] if len(self.forms) == 1 else None

    def current(self):
        """The form whose trial is running (None when all trials are complete)."""
        for f in self.forms:
            if len(self.samples[f]) < self.trial_steps:
                return f
        return None

    def record(self, form, seconds):
        self.samples[form].append(float(seconds))

    def complete(self):
        return self.current() is None

    def decide(self):
        """Collective: every rank calls it once all its trials are complete. Returns the chosen form."""
        if self.chosen is not None:
            return self.chosen
        mean = torch.tensor([sum(self.samples[f]) / len(self.samples[f]) for f in self.forms], dtype=torch.float64, device=self.device)
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            mean = mean.float() if mean.is_cuda else mean   # (RCCL all-reduces fp32 / fp64 alike; keep the gloo path in double)
            dist.all_reduce(mean, op=dist.ReduceOp.MAX, group=self.group)
        vals = [float(v) for v in mean.tolist()]
        self.timings_ms = {f: 1e3 * v for f, v in zip(self.forms, vals)}
        self.chosen = self.forms[min(range(len(vals)), key=lambda i: (vals[i], i))]   # ties: the first form listed
        return self.chosen


def broadcast_flat(flat, src=0, group=None):
    """Identical initial replicas: broadcast rank `src`'s flat parameter arena once."""
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or os.environ.get('LVAE_FORCE_DIST') == '1'):
        dist.broadcast(flat, src=src, group=group)
        from . import kernels
        kernels.prepared.weights_written()  # parameter views do not share the flat buffer's version counter
