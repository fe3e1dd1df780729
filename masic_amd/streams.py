"""Fork / join of HIP streams for the multi-stream eval forward, with the capture rules of this ROCm checked in code.

`HSIC._forward_eval` (coremasic/mywork/MASIC.py) issues independent branches of the forward on side streams and
`masic_amd/graph.py` captures the whole DAG into one HIP graph.  Three stream topologies end `hipStreamEndCapture` on
this ROCm (7.x / torch 2.10) with a process-killing fault instead of an error (found the hard
way in round 1):

  1. a side stream that forks further streams (a fork whose origin is not the capturing stream),
  2. an event recorded on one side stream awaited by another side stream,
  3. a stream waiting for an event it recorded itself.

`ForkJoin` is the only place the forward records / awaits cross-stream events, and it raises `RuntimeError` for each
of the three BEFORE the offending `wait_event` is issued while a capture is running (rule 3 always: waiting for one's
own event is never meaningful).  Issued eagerly, rules 1 and 2 are legal HIP and are allowed.
"""
import contextlib
import threading

import torch

_tls = threading.local()


class _CudaBackend:
    """torch.cuda plumbing; tests substitute recording fakes (tests/test_cpu_streams.py)."""

    @staticmethod
    def current_stream():
        return torch.cuda.current_stream()

    @staticmethod
    def capturing():
        return torch.cuda.is_current_stream_capturing()

    @staticmethod
    def event():
        return torch.cuda.Event()

    @staticmethod
    def stream_ctx(stream):
        return torch.cuda.stream(stream)


class StreamEvent:
    """An event plus the stream it was recorded on (what the three rules are about)."""
    __slots__ = ("event", "owner")

    def __init__(self, event, owner):
        self.event, self.owner = event, owner


class ForkJoin:
    """Cross-stream ordering for one forward.  `main` is the stream the forward was called on (the capturing stream when a
    graph capture is running); every other stream handed to `on()` / `wait()` is a side stream."""

    def __init__(self, main=None, backend=None):
        self.backend = backend if backend is not None else _CudaBackend
        self.main = main if main is not None else self.backend.current_stream()
        # constructed while another ForkJoin is issuing on one of ITS side streams: our `main` is that side stream
        self.nested_in_side = getattr(_tls, "side", None) is not None

    def _same(self, a, b):
        return a is b or a == b

    def is_main(self, stream):
        return self._same(stream, self.main) and not self.nested_in_side

    def record(self, stream=None):
        stream = self.main if stream is None else stream
        ev = self.backend.event()
        ev.record(stream)
        return StreamEvent(ev, stream)

    def wait(self, stream, ev):
        """`stream` waits for `ev`; the three capture rules are checked first."""
        if self._same(stream, ev.owner):
            raise RuntimeError("masic_amd.streams: a stream waiting for its own event (capture rule 3: ends hipStreamEndCapture "
                               "with a fault on this ROCm); it is a no-op, remove the wait")
        if self.backend.capturing():
            if not self.is_main(stream) and not self.is_main(ev.owner):
                if self.nested_in_side and self._same(ev.owner, self.main):
                    raise RuntimeError("masic_amd.streams: fork from a side stream during HIP-graph capture (capture rule 1: a side "
                                       "stream that forks further streams faults in hipStreamEndCapture on this ROCm); issue this "
                                       "branch serially (e.g. heads(parallel=False)) or fork it from the capturing stream")
                raise RuntimeError("masic_amd.streams: a side stream waiting for another side stream's event during HIP-graph capture "
                                   "(capture rule 2: faults in hipStreamEndCapture on this ROCm); route the dependency through the "
                                   "capturing stream")
        stream.wait_event(ev.event)

    def fork(self, side, origin=None):
        """`side` continues after everything issued so far on `origin` (default: main)."""
        origin = self.main if origin is None else origin
        if self.backend.capturing() and not self.is_main(origin):
            raise RuntimeError("masic_amd.streams: fork from a side stream during HIP-graph capture (capture rule 1: a side stream "
                               "that forks further streams faults in hipStreamEndCapture on this ROCm); issue this branch serially "
                               "(e.g. heads(parallel=False)) or fork it from the capturing stream")
        if self._same(side, origin):
            return None            # serial schedule: the "side" stream is the origin itself, nothing to order
        ev = self.record(origin)
        self.wait(side, ev)
        return ev

    def join(self, side, into=None):
        """`into` (default: main) continues after everything issued so far on `side`."""
        into = self.main if into is None else into
        if self._same(side, into):
            return None
        ev = self.record(side)
        self.wait(into, ev)
        return ev

    @contextlib.contextmanager
    def on(self, side):
        """Issue the body on `side`; ForkJoins created inside know they start on a side stream."""
        prev = getattr(_tls, "side", None)
        is_side = not self._same(side, self.main) or self.nested_in_side
        _tls.side = side if is_side else prev
        try:
            with self.backend.stream_ctx(side):
                yield
        finally:
            _tls.side = prev
