"""The three HIP-graph capture rules of masic_amd/streams.py raise RuntimeError BEFORE the offending
wait is issued -- in round 1 each of these topologies ended hipStreamEndCapture with a process-killing fault.  CPU-runnable:
ForkJoin takes a backend; the fakes below record what would have been issued."""
import contextlib

import pytest

from masic_amd.streams import ForkJoin


class FakeStream:
    def __init__(self, name, log):
        self.name, self.log = name, log

    def wait_event(self, ev):
        self.log.append(("wait", self.name, ev.owner_name))

    def __repr__(self):
        return self.name


class FakeEvent:
    def __init__(self, log):
        self.log, self.owner_name = log, None

    def record(self, stream):
        self.owner_name = stream.name
        self.log.append(("record", stream.name))


class FakeBackend:
    def __init__(self, capturing):
        self.log = []
        self.cap = capturing
        self.main = FakeStream("main", self.log)
        self.cur = self.main

    def current_stream(self):
        return self.cur

    def capturing(self):
        return self.cap

    def event(self):
        return FakeEvent(self.log)

    @contextlib.contextmanager
    def stream_ctx(self, s):
        prev, self.cur = self.cur, s
        try:
            yield
        finally:
            self.cur = prev


def test_legal_fork_join_from_the_capturing_stream():
    be = FakeBackend(capturing=True)
    a, e = FakeStream("A", be.log), FakeStream("E", be.log)
    fj = ForkJoin(backend=be)
    fj.fork(a)
    with fj.on(a):
        ev_a = fj.record(a)
    fj.fork(e)
    fj.fork(a)                    # the same side stream picks up a later point of main: legal
    fj.wait(be.main, ev_a)
    fj.join(e)
    fj.join(a)
    waits = [x for x in be.log if x[0] == "wait"]
    assert waits == [("wait", "A", "main"), ("wait", "E", "main"), ("wait", "A", "main"), ("wait", "main", "A"),
                     ("wait", "main", "E"), ("wait", "main", "A")]


def test_rule1_fork_from_a_side_stream_raises_during_capture():
    be = FakeBackend(capturing=True)
    a, s2 = FakeStream("A", be.log), FakeStream("S2", be.log)
    outer = ForkJoin(backend=be)
    outer.fork(a)
    with outer.on(a):
        inner = ForkJoin(backend=be)          # e.g. _GmmHeads.heads(parallel=True) called on the right view's stream
        n = len(be.log)
        with pytest.raises(RuntimeError, match="capture rule 1"):
            inner.fork(s2)
        assert be.log[n:] == []               # nothing was recorded or awaited
    # explicit origin that is not the capturing stream
    with pytest.raises(RuntimeError, match="capture rule 1"):
        outer.fork(s2, origin=a)
    # eagerly the same topology is legal HIP
    be2 = FakeBackend(capturing=False)
    a2, s22 = FakeStream("A", be2.log), FakeStream("S2", be2.log)
    o2 = ForkJoin(backend=be2)
    o2.fork(a2)
    with o2.on(a2):
        ForkJoin(backend=be2).fork(s22)
    assert ("wait", "S2", "A") in be2.log


def test_rule2_side_stream_waiting_for_another_side_streams_event_raises_during_capture():
    be = FakeBackend(capturing=True)
    a, e = FakeStream("A", be.log), FakeStream("E", be.log)
    fj = ForkJoin(backend=be)
    fj.fork(a)
    fj.fork(e)
    ev_e = fj.record(e)
    n = len(be.log)
    with pytest.raises(RuntimeError, match="capture rule 2"):
        fj.wait(a, ev_e)
    assert be.log[n:] == []
    with pytest.raises(RuntimeError, match="capture rule 2"):
        fj.join(e, into=a)
    # routed through the capturing stream it is legal
    fj.wait(be.main, ev_e)
    fj.fork(a)
    # and eagerly the direct wait is allowed
    be2 = FakeBackend(capturing=False)
    a2, e2 = FakeStream("A", be2.log), FakeStream("E", be2.log)
    f2 = ForkJoin(backend=be2)
    f2.wait(a2, f2.record(e2))


@pytest.mark.parametrize("capturing", [True, False])
def test_rule3_stream_waiting_for_its_own_event_raises(capturing):
    be = FakeBackend(capturing=capturing)
    a = FakeStream("A", be.log)
    fj = ForkJoin(backend=be)
    n = len(be.log)
    ev = fj.record(a)
    with pytest.raises(RuntimeError, match="capture rule 3"):
        fj.wait(a, ev)
    with pytest.raises(RuntimeError, match="capture rule 3"):
        fj.wait(be.main, fj.record())
    assert [x for x in be.log[n:] if x[0] == "wait"] == []
    # fork / join onto the same stream (the serial debug schedule) are no-ops, not errors
    assert fj.fork(be.main) is None and fj.join(be.main) is None
