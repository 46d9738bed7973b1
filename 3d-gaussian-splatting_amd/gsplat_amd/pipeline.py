"""The pipelined multi-view step: ONE orchestrator for the views of an optimisation step.

A step of the reference trainer renders B views and sums their gradients (/root/reference/src/training/trainer.py:458-506:
B calls of ``render_with_depth``, one ``backward()``).  The views are independent until their gradients meet in the
bucket, and the two halves of a view stress different parts of the chip: the front end (projection, depth ordering,
binning) is HBM- and latency-bound, the sweeps (compositor forward / backward) are vector-issue-bound.  ``ViewPipeline``
therefore

* puts whole views on S alternating HIP streams (S = 2: view j + 1's front end runs beside view j's sweeps);
* orders only what must be ordered: the bucket is summed into in view order (view j's accumulation waits for view
  j - 1's, an event per view), and the caller's stream waits for all views at the end;
* never waits, host side, for a view's control words (record count, depth-key range) while it queues the step: every
  view is rendered with a DEFERRED check (``rasterization(defer_check=True)`` / ``TileBins.finalize`` left for later),
  the checks run once per step behind the last view -- where a trainer has its optimizer -- all ranks agree on the answer,
  and a step in which a remembered capacity or key range did not hold is rendered again with exact sizes (nothing was
  applied yet).

Both users go through this one function: ``bench.py`` (autograd path: ``rasterization`` + ``backward``) and
``harness.ViewParallelTrainer._training_step_fused`` (the stages called directly, loss kernel between the sweeps).
They hand it two callables; it owns streams, events, ordering, the deferred check and the repair.
"""

from __future__ import annotations

import contextlib
import os
from typing import Callable, List, Optional, Sequence

import torch


class ViewState:
    """What ``forward`` returns for one view: anything the matching ``backward`` needs (``payload``) and ``check()``,
    which waits for the view's control words and says whether its speculation held."""

    __slots__ = ("payload", "check")

    def __init__(self, payload, check: Callable[[], bool]):
        self.payload, self.check = payload, check


# One set of side streams per (device, count, priority) for ALL pipelines of the process: the runtime multiplexes streams onto a
# few hardware queues in creation order, and a process that builds several pipelines (bench.py's legs, a trainer beside an
# evaluator) ended up with a pipeline whose two streams shared ONE queue -- its views then ran one after the other (the
# training-step leg of bench.py 2.82 ms where the same step alone in a process takes 2.53).
_STREAMS = {}
# GSPLAT_AMD_DEBUG_LAG="<stream index>:<cycles>" (developer aid): a spin kernel in front of every view of that side stream
_DEBUG_LAG = tuple(int(x) for x in os.environ["GSPLAT_AMD_DEBUG_LAG"].split(":")) if os.environ.get("GSPLAT_AMD_DEBUG_LAG") else None


def _side_streams(device, n: int, priority: int) -> List[torch.cuda.Stream]:
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device(), n, priority)
    if key not in _STREAMS:
        _STREAMS[key] = [torch.cuda.Stream(device=device, priority=priority) for _ in range(n)]
    return _STREAMS[key]


class ViewPipeline:
    def __init__(self, device, n_streams: int = 2, priority: int = 0, single_thread_autograd: bool = True,
                 speculation=None):
        self.device = device
        # what the asynchronous binning remembers about the frames of THIS pipeline's views (stages.Speculation): the two
        # callables hand it to rasterization(speculation=...) / bin_tiles(speculation=...); another pipeline of the same
        # frame size -- an evaluator beside a trainer -- keeps its own
        if speculation is None:
            from .stages import Speculation
            speculation = Speculation()
        self.speculation = speculation
        self.single_thread_autograd = bool(single_thread_autograd)
        self.streams: List[torch.cuda.Stream] = _side_streams(device, n_streams, priority) if n_streams > 1 else []
        self._step_done: Optional[torch.cuda.Event] = None   # the previous step's last use of the bucket (caller's stream)
        self.repairs = 0            # steps that were rendered again because a view's speculation failed
        self.steps = 0

    def run(self, n_views: int, forward: Callable[[int, bool], ViewState],
            backward: Callable[[int, ViewState, Optional[torch.cuda.Event]], None], *,
            after_view: Optional[Callable[[int], None]] = None, agree: Optional[Callable[[bool], bool]] = None,
            inputs_ready: Optional[torch.cuda.Event] = None, defer: bool = True,
            prepare: Optional[Callable[[], Sequence[torch.Tensor]]] = None,
            finish: Optional[Callable[[], None]] = None) -> Sequence[ViewState]:
        """Queues ``n_views`` views, forward + backward each, and returns their states once the step is known good.

        ``forward(j, deferred)``: queue view j's front end and forward sweep (and whatever sits between the sweeps, e.g. the
        loss) on the CURRENT stream; with ``deferred`` it must not wait for the view's control words but return a
        ``ViewState`` whose ``check()`` does; without, it checks and repairs on its own (``check`` then returns True).
        ``backward(j, state, before_bucket)``: queue view j's backward on the current stream; it must make that stream wait
        for ``before_bucket`` (an event, or None) before it touches the gradient bucket.
        ``after_view(j)``: queued right behind view j's backward, on its stream (the exchange of that view's part).
        ``agree(ok)``: the ranks' conjunction (every rank must repeat a step, and its collectives, or none).
        ``inputs_ready``: an event behind which the step's inputs (the parameters) are valid; every stream waits for it
        before its first view.  None: the inputs did not change since the streams last saw them.
        ``prepare()``: what the step's views SHARE, queued once in front of the first view on its stream -- the projection of
        the gaussians for all cameras of the step in one launch (``project_cameras``); the other streams wait for it.  It
        returns the tensors it allocated: they are used on every stream of the pipeline, which the allocator is told.
        ``finish()``: queued once on the caller's stream behind ALL views' backward passes -- the projection backward of
        all cameras in one launch (``project_cameras_backward``) when the views' own backward passes stop behind their sweeps.
        """
        self.steps += 1
        main = torch.cuda.current_stream()
        use = self.streams
        defer = bool(defer)
        states: List[ViewState] = []
        done: List[torch.cuda.Event] = []
        prev: Optional[torch.cuda.Event] = None
        seen = set()
        shared: Optional[torch.cuda.Event] = None
        # (the current stream is switched with set_stream and put back at the end: torch.cuda.stream()'s context manager
        # builds two Stream objects per use, ~25 us of host time per view of a loop that has to stay ahead of the device)
        # (backward passes called inside run on THIS thread: autograd otherwise hands every backward() to a per-device worker
        # thread and waits for it -- 125 us of host time per view, a third of what queueing a view costs, in a loop whose
        # only job is to stay ahead of the device.  One process drives one GPU here: nothing to run in parallel.)
        autograd_here = torch.autograd.set_multithreading_enabled(False) if self.single_thread_autograd else contextlib.nullcontext()
        try:
            with autograd_here:
                if prepare is not None:
                    st = use[0] if use else None
                    if st is not None:
                        torch.cuda.set_stream(st)
                        if inputs_ready is not None:
                            st.wait_event(inputs_ready)
                            seen.add(id(st))
                    made = prepare()
                    if st is not None:
                        shared = torch.cuda.Event()
                        shared.record(st)
                        for t in made or ():
                            for other in use[1:]:
                                t.record_stream(other)
                waited = {id(use[0])} if (use and shared is not None) else set()
                for j in range(n_views):
                    st = use[j % len(use)] if use else None
                    if st is not None:
                        torch.cuda.set_stream(st)
                        if inputs_ready is not None and id(st) not in seen:
                            st.wait_event(inputs_ready)
                            seen.add(id(st))
                        if shared is not None and id(st) not in waited:
                            st.wait_event(shared)
                            waited.add(id(st))
                    if _DEBUG_LAG and st is not None and (j % len(use)) == _DEBUG_LAG[0]:
                        torch.cuda._sleep(_DEBUG_LAG[1])   # (debug: this stream falls behind the others, as on a shared card)
                    state = forward(j, defer)
                    # the bucket is summed into in view order; the first view also waits for the previous step's last use of it
                    before = prev if j > 0 else (self._step_done if st is not None else None)
                    backward(j, state, before)
                    if st is not None:
                        prev = torch.cuda.Event()
                        prev.record(st)
                        done.append(prev)
                    if after_view is not None:
                        after_view(j)
                    states.append(state)
        finally:
            if use:
                torch.cuda.set_stream(main)
        for ev in done:
            main.wait_event(ev)
        if finish is not None:
            finish()
        if use:
            self._step_done = torch.cuda.Event()
            self._step_done.record(main)
        if defer:
            # (every view is checked, also behind a failed one: a check finalizes the view's bins -- its pinned control-word slot
            # goes back to the pool, its record count and key range reach what is remembered for the repair run)
            ok = all([s.check() for s in states])
            if agree is not None:
                ok = agree(ok)
            if not ok:
                self.repairs += 1
                return self.run(n_views, forward, backward, after_view=after_view, agree=None, inputs_ready=None, defer=False,
                                finish=finish)
        return states

    def mark_bucket_used(self) -> None:
        """Call after the caller's stream has queued its last read / write of the gradient bucket for this step (exchange,
        optimizer): the next step's first accumulation waits for this point."""
        if self.streams:
            self._step_done = torch.cuda.Event()
            self._step_done.record(torch.cuda.current_stream())
