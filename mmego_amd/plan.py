"""A step as SEVERAL single-chain HIP graphs on several streams, instead of one HIP graph with parallel branches.

Why: on this ROCm a replayed graph whose branches run side by side dispatches every node in 5-6 us, a graph that is one linear
chain in 1.5 us (scripts/bench_launch_gap.py: two chains of 400 tiny kernels take 2.58 ms as two branches of one graph, 0.74 ms
as two graphs replayed on two streams).  The engines of train_step.py are written in stream terms -- fork a side stream, launch on
both, join -- which graph capture turns into branches.  StepPlan runs such a body ONCE in recording mode: nothing is launched;
every C-ABI launch (hip._launch is the only place this package launches anything) is appended to the open SEGMENT of the stream it was
issued on, and every `Stream.wait_stream` closes the awaited stream's segment (an event will be recorded behind it) and makes
the waiting stream's next segment depend on that event.  Each segment -- a linear run of launches on one stream -- is then
captured as a HIP graph of its own; replay() enqueues the segments in their order of creation on their streams with the
recorded event waits in between.  Same launches, same arguments, same order per stream and the same cross-stream dependencies as
the eager body: results are bit-identical (tests/test_bench_shape.py).

A segment never forks inside a capture, so the nested-fork defect of hipStreamEndCapture cannot occur
here, whatever the body's stream topology.
"""
import torch

from . import hip


class _Segment:
    __slots__ = ("stream", "calls", "deps", "signal", "graph")

    def __init__(self, stream, deps):
        self.stream, self.calls, self.deps, self.signal, self.graph = stream, [], deps, None, None


class StepPlan:
    def __init__(self):
        self.segments = []
        self.final_waits = []            # events the launching stream waits for at the end of a replay
        self._open = {}                  # stream handle -> open segment
        self._last = {}                  # stream handle -> last closed segment (its event marks the stream's position)
        self._pending = {}               # stream handle -> events the stream's NEXT segment waits for
        self._streams = {}               # stream handle -> torch stream object (None: the launching stream)
        self._main = None
        self._start = None
        self.built = False

    # -- recording ---------------------------------------------------------------------------------------
    def _key(self, stream):
        return stream.cuda_stream

    def _segment_for(self, stream):
        k = self._key(stream)
        seg = self._open.get(k)
        if seg is None:
            seg = _Segment(k, self._pending.pop(k, []))
            self._open[k] = seg
            self.segments.append(seg)
            if k not in self._streams:
                self._streams[k] = None if k == self._main else stream
        return seg

    def _close(self, k):
        seg = self._open.pop(k, None)
        if seg is not None:
            self._last[k] = seg
        return self._last.get(k)

    def _position_event(self, k):
        """Event behind everything recorded on stream k so far (None: nothing was, the plan's start is the position)."""
        seg = self._close(k)
        if seg is None:
            return None
        if seg.signal is None:
            seg.signal = torch.cuda.Event()
        return seg.signal

    def record(self, body):
        """Run ``body`` with launches recorded instead of issued.  The body must be one whose every device operation goes through
        hip.call (true of the stage bodies: DESIGN.md section 1); allocations it makes stay alive with the plan.  Stream.wait_event and
        Event.record raise while recording (they would not be replayed); a torch compute op in the body cannot be intercepted
        here -- tests compare run_eagerly() / replay() with the eager body to catch one.  Not thread-safe (class-wide patches)."""
        if self.built or self.segments:
            raise RuntimeError("a StepPlan records one body")
        plan = self
        self._main = torch.cuda.current_stream().cuda_stream
        orig_call, orig_wait = hip._launch, torch.cuda.Stream.wait_stream
        orig_wait_event, orig_record = torch.cuda.Stream.wait_event, torch.cuda.Event.record

        def refuse(what):
            def raiser(*a, **k):
                raise RuntimeError("StepPlan.record: the body called %s -- only hip.call launches and Stream.wait_stream are "
                                   "recorded; anything else would run once now and be missing from every replay" % what)
            return raiser

        def rec_call(name, *args):
            plan._segment_for(torch.cuda.current_stream()).calls.append((name, args))

        def rec_wait(self_stream, other):
            ks, ko = plan._key(self_stream), plan._key(other)
            if ks == ko:
                return
            ev = plan._position_event(ko)
            plan._close(ks)                                   # what follows on the waiting stream is a new segment
            mine = plan._pending.setdefault(ks, [])
            if ev is not None and ev not in mine:
                mine.append(ev)
            # transitive order: waits the awaited stream has accepted but not yet launched behind (a join followed directly by a
            # fork -- A.wait_stream(B); C.wait_stream(A) with no launch on A in between) bind the waiter too, as they do eagerly
            # and under capture
            for pe in plan._pending.get(ko, []):
                if pe not in mine:
                    mine.append(pe)
            if ks not in plan._streams:
                plan._streams[ks] = None if ks == plan._main else self_stream
        hip._launch, torch.cuda.Stream.wait_stream = rec_call, rec_wait
        torch.cuda.Stream.wait_event, torch.cuda.Event.record = refuse("Stream.wait_event"), refuse("Event.record")
        try:
            if hip._gemm_rec is not None:
                raise RuntimeError("StepPlan.record inside a gemm_group context")
            body()                                            # (gemm_group contexts inside the body defer and group as always)
        finally:
            hip._launch, torch.cuda.Stream.wait_stream = orig_call, orig_wait
            torch.cuda.Stream.wait_event, torch.cuda.Event.record = orig_wait_event, orig_record
        for k in list(self._open):
            self._close(k)
        # whatever the launching stream was told to wait for behind its last launch, and every other stream's tail: the replay ends
        # with the launching stream behind all of it (the caller's next launches -- the optimiser -- are ordered after the whole step)
        waits = list(self._pending.pop(self._main, []))
        for k, seg in self._last.items():
            if k != self._main:
                if seg.signal is None:
                    seg.signal = torch.cuda.Event()
                if seg.signal not in waits:
                    waits.append(seg.signal)
        self.final_waits = waits
        self.segments = [s for s in self.segments if s.calls or s.signal is not None]
        return self

    # -- graphs ------------------------------------------------------------------------------------------
    def build(self):
        """Capture every segment as a HIP graph of its own (a linear chain of kernel nodes)."""
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        torch.cuda.synchronize()
        for seg in self.segments:
            if not seg.calls:
                continue
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(cap):
                with torch.cuda.graph(g, stream=cap):
                    for name, args in seg.calls:
                        hip._launch(name, *args)
            seg.graph = g
        torch.cuda.synchronize()
        self.built = True
        return self

    def run_eagerly(self):
        """The recorded launches issued directly (no graphs), same streams and dependencies: for checks and profiling."""
        self._issue(lambda seg: [hip._launch(name, *args) for name, args in seg.calls])

    def replay(self):
        if not self.built:
            self.build()
        self._issue(lambda seg: seg.graph.replay())

    def _issue(self, launch):
        main = torch.cuda.current_stream()
        # the side streams start behind whatever the launching stream holds already (the previous step's optimiser update)
        started = False
        for seg in self.segments:
            st = self._streams.get(seg.stream) or main
            if st is not main and not seg.deps:
                if not started:
                    if self._start is None:
                        self._start = torch.cuda.Event()
                    self._start.record(main)
                    started = True
                st.wait_event(self._start)
            for ev in seg.deps:
                st.wait_event(ev)
            if seg.calls:
                with torch.cuda.stream(st):
                    launch(seg)
            if seg.signal is not None:
                seg.signal.record(st)
        for ev in self.final_waits:
            main.wait_event(ev)

    # -- structure ---------------------------------------------------------------------------------------
    def unordered_with(self, pred):
        """The launches that MAY RUN AT THE SAME TIME as a launch whose entry-point name satisfies ``pred``: pairs
        (name of the pred launch, name of the other launch) for every two launches of different segments with no
        happens-before path between their segments (stream order + the recorded event waits) either way.  Empty = every launch
        of the recorded body is ordered against every `pred` launch: no kernel can be resident beside one of them.  Structural:
        read off the recorded dependency graph, nothing is launched (tests/test_split3_gpu.py holds the engines to it with
        pred = hip.is_bf16_mfma_entry)."""
        segs = self.segments
        by_signal = {id(sg.signal): i for i, sg in enumerate(segs) if sg.signal is not None}
        n = len(segs)
        before = [set() for _ in range(n)]                    # before[i]: segments that finish before segment i starts
        last_on = {}
        for i, sg in enumerate(segs):
            preds = set()
            if sg.stream in last_on:
                preds.add(last_on[sg.stream])
            for ev in sg.deps:
                j = by_signal.get(id(ev))
                if j is not None:
                    preds.add(j)
            for j in preds:
                before[i].add(j)
                before[i] |= before[j]
            last_on[sg.stream] = i
        pairs = set()
        for i, a in enumerate(segs):
            hot = {name for name, _ in a.calls if pred(name)}
            if not hot:
                continue
            for j, b in enumerate(segs):
                if i == j or j in before[i] or i in before[j]:
                    continue
                for h in hot:
                    for name, _ in b.calls:
                        pairs.add((h, name))
        return sorted(pairs)

    def describe(self):
        return "%d segments on %d streams, %d launches" % (len(self.segments), len({s.stream for s in self.segments}),
                                                          sum(len(s.calls) for s in self.segments))
