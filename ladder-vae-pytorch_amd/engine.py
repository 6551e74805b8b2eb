"""Whole-step execution: forward + backward + L2 norm [+ Adamax] captured once as a hipGraph and replayed.

The eager schedule of one CIFAR-15 step is several thousand dependent launches; replaying a
captured graph removes the Python / launch overhead without a tracing compiler. RNG state lives in device memory
and is advanced by a kernel inside the graph, so every replay draws fresh noise. With world_size > 1 the gradient
all-reduce is part of the same graph: each bucket of the (completion-ordered) gradient arena is exchanged on a side stream as
soon as backward has left it (dist.GradAllReduce), and Adamax waits for the last one. LVAE_DDP_MODE=split keeps the exchange
outside the graphs instead (fwd+bwd graph | eager all-reduce | Adamax graph): collectives that are not captured.
"""
import os
import sys
import time

import torch

from . import kernels as K
from . import ops


def linear_anneal(step, start, end, steps):
    """boilr.utils.linear_anneal (absent; restated): linear ramp from start to end over `steps` steps, then flat."""
    if steps <= 0:
        return end
    return start + (end - start) * min(max(step / float(steps), 0.0), 1.0)


def forward_pass(model, x, beta=1.0, compute_l2=True):
    """LVAEExperiment.forward_pass (experiment/experiment_manager.py:322-367) on the HIP engine."""
    mo = model(x)
    elbo_sep, loss, elbo, recons = ops.ElboLossFn.apply(mo['ll'], mo['kl_sep'], mo['kl_loss'], float(beta))
    out = {'loss': loss, 'elbo': elbo, 'elbo_sep': elbo_sep, 'kl': mo['kl'], 'recons': recons,
           'out_mean': mo['out_mean'], 'out_mode': mo['out_mode'], 'out_sample': mo['out_sample'],
           'likelihood_params': mo['likelihood_params'], 'kl_avg_layerwise': mo['kl_avg_layerwise']}
    if compute_l2:
        with torch.no_grad():
            out['l2'] = K.l2norm(model.arena.params).view(())
    return out


class TrainStep:
    """step(x) -> dict of scalars (device tensors, valid until the next step)."""

    def __init__(self, model, optimizer, beta=1.0, use_graph=True, allreduce=None, eager_warmup=2, async_wgrad=False,
                 wgrad_streams=1, wgrad_group_rows=16384):
        self.model, self.opt, self.beta = model, optimizer, beta
        dev = next(model.parameters()).device
        self.side = [torch.cuda.Stream(device=dev) for _ in range(max(1, int(wgrad_streams)))] if async_wgrad else None
        self.use_graph, self.allreduce = use_graph, allreduce
        self.wgrad_group_rows = wgrad_group_rows
        self.eager_left = eager_warmup if use_graph else -1
        self.graph_a = self.graph_b = None
        self.fallback_reason = None
        self.static_x = None
        self.static_out = None
        self._bns = None
        self._table_ref = None
        if allreduce is not None and allreduce.world > 1:
            optimizer._state()
            optimizer.gscale = allreduce.scale
        if (allreduce is not None and allreduce.active and allreduce.on_gpu and not allreduce.capturable
                and os.environ.get('LVAE_ALLOW_GLOO_GRAPH') != '1'):   # (the override exists for the stall diagnosis of DESIGN.md §6)
            self.use_graph, self.eager_left = False, -1   # gloo stages device buffers through the host: cannot be part of a graph
        self.trace = os.environ.get('LVAE_STEP_TRACE') == '1'   # diagnosis only: host-synchronised phase times on stderr
        self.overlap = allreduce is not None and allreduce.active and allreduce.overlap
        if self.overlap and self.use_graph:
            # can the exchange be part of the step graph? If not, the same process continues with the exchange
            # outside the graphs: fwd+bwd graph | eager all-reduce | Adamax graph
            ok, why = allreduce.capture_probe()
            if not ok:
                print('[lvae] the gradient exchange cannot be captured into the step graph (%s); keeping it outside '
                      '(LVAE_DDP_MODE=split)' % why, file=sys.stderr, flush=True)
                self.overlap, self.fallback_reason = False, why
        if self.overlap:
            model.grad_tracker = allreduce   # the model's segment markers report to it during backward

    def _fwd_bwd(self, x):
        done = False
        try:
            self.opt.zero_grad()
            K.prepared.prepare_all()  # one launch: transformed weights of every Winograd convolution seen so far
            out = forward_pass(self.model, x, self.beta)
            if self.allreduce is not None:
                self.allreduce.begin_step()
            ops.set_wgrad_stream(self.side)
            ops.set_wgrad_grouping(self.wgrad_group_rows)
            if self.side is not None:
                for st in self.side:
                    st.wait_stream(torch.cuda.current_stream())  # zero_grad happens-before every wgrad accumulate
            out['loss'].backward()
            ops.flush_wgrad_group()
            ops.join_wgrad_stream()
            if self.overlap:
                self.allreduce.finish()  # last bucket + join: the gradients are summed over ranks from here on
            done = True
        finally:
            if not done:
                # The pass was abandoned by an exception (possibly inside a hipGraph capture that is being torn down). Nothing more may be
                # LAUNCHED on its behalf: the queued weight gradients are dropped, not flushed (their inputs belong to the abandoned pass);
                # the transformed-weight stamps are withdrawn (prepare_all marked the buffers current for a launch that may never have
                # run); the exchange's bucket cursor is rewound; side streams are joined so that no fork is left dangling. The next
                # step — eager or a fresh capture — starts from a clean state (tests/test_model_gpu.py raises inside a captured backward).
                ops.drop_wgrad_group()
                K.prepared.invalidate()
                if self.allreduce is not None:
                    self.allreduce.begin_step()
                if not torch.cuda.is_current_stream_capturing():
                    try:
                        ops.join_wgrad_stream()
                    except RuntimeError:
                        pass
            ops.set_wgrad_grouping(None)
            ops.set_wgrad_stream(None)
        return {k: out[k].detach() for k in ('loss', 'elbo', 'recons', 'kl', 'l2', 'kl_avg_layerwise')}

    def _eager(self, x):
        out = self._fwd_bwd(x)
        if self.allreduce is not None and not self.overlap:
            self.allreduce.run()
        self.opt.step()
        return out

    def _capture(self, x):
        self.static_x = torch.empty_like(x)
        self.static_x.copy_(x)
        fused = self.allreduce is None or not self.allreduce.active or self.overlap
        torch.cuda.synchronize()
        self.graph_a = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread of an initialised process group may query events while this thread captures.
        # (Whether the exchange CAN be captured was settled by GradAllReduce.capture_probe() in __init__: a refusal inside this capture
        # would leave a half-captured training step behind, and unwinding that is not safe.)
        # Host-side counters the Python pass advances while it is captured: if EITHER capture fails they go back, so that a caller who catches
        # the error and steps again neither replays half a step (graph A without the exchange and the optimizer) nor counts a BatchNorm
        # forward / a global step that never ran
        bns = self.model.bn_modules()
        pend0 = [bn._pending for bn in bns]
        try:
            with torch.cuda.graph(self.graph_a, capture_error_mode='thread_local'):
                self.static_out = self._fwd_bwd(self.static_x)
                if fused:
                    self.opt.step()
            if not fused:
                self.graph_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool(), capture_error_mode='thread_local'):
                    self.opt.step()
        except BaseException:
            # a failed capture leaves no usable graph: drop both so that the next call captures again (or runs eagerly) from clean state
            self.graph_a = self.graph_b = None
            self.static_out = None
            self._table_ref = None
            for bn, p0 in zip(bns, pend0):
                bn._pending = p0
            self.model.global_step -= 1   # __call__ counted this step before capturing it
            raise
        self._bns = bns
        # the captured prepare_all launch baked in the device address (and entry count) of the transformed-weight table: keep exactly
        # that table, and the scratch buffers its entries point to, alive and unmodified for as long as this graph can be replayed
        self._table_ref = K.prepared.pin_current()

    def exchange_description(self):
        """How the gradients are exchanged in this process (bench.py's config.grad_exchange)."""
        ar = self.allreduce
        if ar is None or not ar.active:
            return 'none (single rank)'
        if self.overlap:
            where = 'inside the step graph' if self.use_graph else 'eager launches'
            return '%d completion-ordered buckets on a side stream during backward, %s' % (len(ar.buckets), where)
        why = ' (fallback: %s)' % self.fallback_reason if getattr(self, 'fallback_reason', None) else ''
        return ('%d bucket(s) after backward on the side stream, between the fwd+bwd graph and the Adamax graph (split: side branches of a '
                'captured graph do not overlap on this platform, DESIGN.md §6)%s' % (len(ar.buckets), why))

    def _traced_split_step(self):
        ts = [time.perf_counter()]
        for phase in (self.graph_a.replay, self.allreduce.run, self.graph_b.replay):
            phase()
            torch.cuda.synchronize()
            ts.append(time.perf_counter())
        print('[step-trace] graph A %.1f ms | all-reduce %.1f ms | graph B %.1f ms' %
              tuple((b - a) * 1e3 for a, b in zip(ts, ts[1:])), file=sys.stderr, flush=True)
        K.prepared.weights_written()
        return self.static_out

    def __call__(self, x):
        self.model.global_step += 1
        if not self.use_graph or self.eager_left > 0:
            self.eager_left -= 1
            return self._eager(x)
        just_captured = self.graph_a is None
        if just_captured:
            self._capture(x)  # the Python forward ran once while capturing: it already counted this step's BN forwards
        else:
            self.static_x.copy_(x, non_blocking=True)
        if self.trace and self.graph_b is not None:
            return self._traced_split_step()
        self.graph_a.replay()
        if self.graph_b is not None:
            self.allreduce.run()
            self.graph_b.replay()
        K.prepared.weights_written()  # the replayed optimizer kernel changed the weights behind Python's back
        if self.model.training and not just_captured:
            for bn in self._bns:
                bn._pending += 1
        return self.static_out


class AutoExchangeStep:
    """step(x) for a multi-rank run that does not take the form of its gradient exchange on faith.

    north_star asks for the all-reduce "overlapped with backward on a side HIP stream"; on a one-rank rehearsal that form measured SLOWER
    inside a captured step than one exposed message between two graphs. Which one wins on N real ranks over xGMI is a
    measurement this process can make itself: it builds BOTH forms — 'split' (fwd+bwd graph | whole-arena all-reduce | Adamax graph) and
    'overlap' (completion-ordered 8 MB buckets on the side stream during backward, one graph) — on one shared communicator, runs
    `trial_steps` timed steps of each after its warm-up / capture steps (all of them REAL training steps: both forms compute the same
    update), reduces the times with MAX over ranks (dist.FormSelector) and continues with the faster form; the other form's graph is dropped.
    LVAE_DDP_MODE=split|overlap skips the trial. `timings_ms` / `chosen` go into bench.py's line (config.grad_exchange_ab)."""

    def __init__(self, model, optimizer, flat_grads, segments, group=None, trial_steps=3, forms=None, **step_kwargs):
        from . import dist as ldist
        self.model = model
        env = os.environ.get('LVAE_DDP_MODE')
        forms = list(forms) if forms is not None else ([env] if env in ('split', 'overlap') else ['split', 'overlap'])
        self.ars, self.steps = {}, {}
        comm = None
        for f in forms:
            ar = ldist.GradAllReduce(flat_grads, group=group, segments=segments, mode=f, comm=comm)
            if comm is None:
                comm = ar.comm
            st = TrainStep(model, optimizer, allreduce=ar, **step_kwargs)
            if f == 'overlap' and not st.overlap and 'split' in forms:
                continue   # the overlapped form is not available here (capture probe refused): it would just be a second 'split'
            self.ars[f], self.steps[f] = ar, st
        dev = flat_grads.device
        self.selector = ldist.FormSelector(list(self.steps), trial_steps, group=group, device=dev if flat_grads.is_cuda else 'cpu')
        self.group = group
        self.chosen = self.selector.chosen
        self.timings_ms = None
        self._untimed = {}
        self._arm(self.chosen or self.selector.current())

    def _arm(self, form):
        st = self.steps[form]
        self.model.grad_tracker = self.ars[form] if st.overlap else None
        return st

    @property
    def ready(self):
        return self.chosen is not None

    @property
    def use_graph(self):
        return self.steps[self.chosen or next(iter(self.steps))].use_graph

    @property
    def allreduce(self):
        return self.ars[self.chosen or next(iter(self.ars))]

    def __call__(self, x):
        if self.chosen is not None:
            return self.steps[self.chosen](x)
        import torch.distributed as tdist
        form = self.selector.current()
        st = self._arm(form)
        # eager warm-up steps and the capturing step are not timed; without a graph, the first step of a form (lazy initialisations) is not
        warming = (st.use_graph and (st.eager_left > 0 or st.graph_a is None)) or self._untimed.get(form, 0) < 1
        if warming:
            self._untimed[form] = self._untimed.get(form, 0) + 1
            return st(x)
        on_gpu = x.is_cuda
        if tdist.is_initialized() and tdist.get_world_size(self.group) > 1:
            tdist.barrier(group=self.group)
        if on_gpu:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = st(x)
        if on_gpu:
            torch.cuda.synchronize()
        self.selector.record(form, time.perf_counter() - t0)
        if self.selector.complete():
            self.chosen = self.selector.decide()
            self.timings_ms = self.selector.timings_ms
            for f in list(self.steps):
                if f != self.chosen:   # the loser's graphs (and their memory pools) go
                    self.steps[f].graph_a = self.steps[f].graph_b = None
                    self.steps[f].static_out = None
            self._arm(self.chosen)
            print('[lvae] gradient exchange: %s' % self.exchange_description(), file=sys.stderr, flush=True)
        return out

    def exchange_description(self):
        if self.chosen is None:
            return 'undecided (trial steps still running)'
        d = self.steps[self.chosen].exchange_description()
        if self.timings_ms:
            d += ' | chosen at run time: ' + ', '.join('%s %.3f ms/step' % (f, t) for f, t in self.timings_ms.items()) + \
                 ' (max over ranks of %d timed steps each)' % self.selector.trial_steps
        return d

    def close(self):
        seen = set()
        for ar in self.ars.values():
            if ar.comm is not None and id(ar.comm) not in seen and ar.owns_comm:
                seen.add(id(ar.comm))
                ar.close()
            else:
                ar.comm = None
