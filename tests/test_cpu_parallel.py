"""CPU suite: the N>1 path -- pair sharding and the bucketed gradient all-reduce -- with world_size-2 gloo."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from masic_amd.parallel import GradientAllReducer, shard_range


def test_shard_range_partitions_exactly():
    for n in (1, 7, 8, 16, 33):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(6, 40)
        self.b = torch.nn.Linear(40, 40)
        self.c = torch.nn.Linear(40, 3)
        self.unused = torch.nn.Parameter(torch.zeros(5))      # like the EB quantiles: no gradient from the main loss

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x)))))


def _worker(rank, world, port, overlap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = _Net()
    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(1))
    y = torch.randn(8, 3, generator=torch.Generator().manual_seed(2))
    lo, hi = shard_range(8, rank, world)
    red = GradientAllReducer(net, bucket_bytes=4096, overlap=overlap)       # several small buckets
    assert len(red.buckets) >= 2
    opt = torch.optim.SGD(net.parameters(), lr=0.0)
    for step in range(2):          # the second step runs on the learnt no-gradient set and on re-bound views after zero_grad()
        opt.zero_grad()
        red.arm()
        loss = ((net(x[lo:hi]) - y[lo:hi]) ** 2).mean()
        loss.backward()
        # every gradient lives in its bucket's flat buffer: no gather / scatter copies around the collective
        assert all(p.grad.data_ptr() == red._views[id(p)].data_ptr() for p in red.params)
        red.finish()
    grads = {n: (None if p.grad is None else p.grad.clone()) for n, p in net.named_parameters()}
    # single-process reference: the mean over equal shards of per-shard mean losses == full-batch mean loss
    ref = _Net()
    ref.load_state_dict(net.state_dict())
    ((ref(x) - y) ** 2).mean().backward()
    ok = all((g is None and rp.grad is None) or torch.allclose(g, rp.grad, rtol=1e-5, atol=1e-7)
             for (n, g), (_, rp) in zip(grads.items(), ref.named_parameters()))
    q.put((rank, ok, grads["unused"] is None))
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_gradient_allreduce_world2_gloo_matches_full_batch(overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(unused_none for _, _, unused_none in res)


def _worker_graphs(rank, world, port, q):
    """Changing graphs and gradient accumulation at world 2 (the cases a per-bucket COUNT of hook firings gets wrong: a second
    backward, or a parameter outside the learnt set, would have launched a bucket while autograd was still accumulating into it)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = _Net()
    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(1))
    y = torch.randn(8, 3, generator=torch.Generator().manual_seed(2))
    lo, hi = shard_range(8, rank, world)

    def full(m, a, b):
        return ((m(a) - b) ** 2).mean()

    def partial(m, a, b):          # does not reach layer c
        return (torch.relu(m.b(torch.relu(m.a(a)))) ** 2).mean()

    def ref_grads(losses):
        r = _Net()
        r.load_state_dict(net.state_dict())
        for f in losses:
            f(r, x, y).backward()
        return [None if p.grad is None else p.grad.clone() for p in r.parameters()]

    def same(want):
        return all((p.grad is None and w is None) or (p.grad is not None and w is not None and torch.allclose(p.grad, w, rtol=1e-5, atol=1e-7))
                   for p, w in zip(net.parameters(), want))

    res = {}
    red = GradientAllReducer(net, bucket_bytes=4096, overlap=True)
    # graph changes from step to step: partial -> full (layer c fires outside the learnt set) -> partial -> full
    ok = True
    for f in (partial, full, partial, full, full):
        net.zero_grad()
        red.arm()
        f(net, x[lo:hi], y[lo:hi]).backward()
        red.finish()
        ok = ok and same(ref_grads([f]))
    res["changing_graph"] = ok
    # a second backward under overlap: the first launched buckets, the second accumulates into them -> must raise, not race
    net.zero_grad()
    red.arm()
    full(net, x[lo:hi], y[lo:hi]).backward()
    try:
        full(net, x[lo:hi], y[lo:hi]).backward()
        res["second_backward_raises"] = False
    except RuntimeError as e:
        res["second_backward_raises"] = "in flight" in str(e)
    red.finish()                      # drains the collectives both ranks launched
    red.remove()
    # gradient accumulation is what overlap=False is for
    red2 = GradientAllReducer(net, bucket_bytes=4096, overlap=False)
    net.zero_grad()
    red2.arm()
    full(net, x[lo:hi], y[lo:hi]).backward()
    partial(net, x[lo:hi], y[lo:hi]).backward()
    red2.finish()
    res["accumulation_without_overlap"] = same(ref_grads([full, partial]))
    q.put((rank, res))
    dist.destroy_process_group()


def test_gradient_allreduce_world2_changing_graph_and_accumulation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_graphs, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, r in res:
        assert r == {"changing_graph": True, "second_backward_raises": True, "accumulation_without_overlap": True}, (rank, r)


def test_reducer_single_process_is_identity():
    net = _Net()
    x = torch.randn(4, 6)
    net(x).sum().backward()
    plain = {n: (None if p.grad is None else p.grad.clone()) for n, p in net.named_parameters()}
    net.zero_grad()
    red = GradientAllReducer(net, bucket_bytes=1024)
    for _ in range(2):
        net.zero_grad()
        red.arm()
        net(x).sum().backward()
        red.finish()
        for n, p in net.named_parameters():
            if plain[n] is None:
                assert p.grad is None, n          # a parameter the loss does not reach keeps .grad None, as without the reducer
            else:
                assert torch.equal(p.grad, plain[n]), n
    # a second backward into the same (zeroed, re-bound) views accumulates like plain autograd does
    net.zero_grad()
    red.arm()
    net(x).sum().backward()
    net(x).sum().backward()
    red.finish()
    assert torch.allclose(net.a.weight.grad, 2 * plain["a.weight"])
