"""GPU test (-m gpu, one device) of the data-parallel exchange on the REAL model: HSIC(128,192,5) gradients of a
4-pair batch == the mean of the gradients of its two 2-pair shards, each pushed through GradientAllReducer's bucket plan
(arm / hooks / finish with world = 1: the same 166-tensor plan, persistent flat buffers, `quantiles` without gradient,
`encoder1` receiving gradient from two passes).  The collective itself is covered by the world_size-2 gloo tests on CPU
(tests/test_cpu_parallel.py); RCCL over xGMI is only exercised by the driver's multi-GPU bench."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_hsic_shard_gradients_through_reducer_equal_full_batch():
    import MASIC
    from compressai.entropy_models import EntropyModel
    from masic_amd import synth
    from masic_amd.loss import rate_distortion
    from masic_amd.parallel import GradientAllReducer, shard_range
    from oracle.hsic_oracle import NOISE_KEYS
    N, M, K = 128, 192, 5
    B, H, W = 4, 64, 64
    net = MASIC.HSIC(N, M, K)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=31))
    net = net.to(DEV).train()
    x1, x2, hm = (t.to(DEV) for t in synth.synth_inputs(B, H, W, seed=31))
    noise = {k: v.to(DEV) for k, v in synth.synth_noise(B, N, M, H, W, seed=31).items()}
    hw_z = (H // 64) * (W // 64)
    red = GradientAllReducer(net)
    plan = red.bucket_plan()
    assert sum(n for n, _ in plan) == 166 and len(plan) >= 4, plan
    print("bucket plan (tensors, MiB):", [(n, round(b / 2 ** 20, 1)) for n, b in plan])
    names = {id(p): n for n, p in net.named_parameters()}
    # reverse registration order: the first bucket holds the parameters whose gradients backward produces first
    assert names[id(red.buckets[0][0])].startswith("mask2weights_unit")

    def run(lo, hi):
        queue = []
        for k in NOISE_KEYS:       # the 7 draws of a training forward, restricted to the pairs of this shard
            queue.append(noise[k][:, :, lo * hw_z:hi * hw_z].contiguous() if k[0] == "z" else noise[k][lo:hi].contiguous())
        orig = EntropyModel._get_noise_cached
        EntropyModel._get_noise_cached = lambda self, x: queue.pop(0).reshape(x.shape)
        try:
            net.zero_grad()
            red.arm()
            out = net(x1[lo:hi].contiguous(), x2[lo:hi].contiguous(), hm[lo:hi].contiguous())
            rate_distortion(out, x1[lo:hi].contiguous(), x2[lo:hi].contiguous(), 0.01)["loss"].backward()
            for p in red.params:   # accumulated in place in the flat bucket buffers
                assert p.grad is not None and p.grad.data_ptr() == red._views[id(p)].data_ptr()
            red.finish()
        finally:
            EntropyModel._get_noise_cached = orig
        assert not queue
        return {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in net.named_parameters()}

    full = run(0, B)
    shards = [run(*shard_range(B, r, 2)) for r in range(2)]
    assert full["entropy_bottleneck1.quantiles"] is None and full["entropy_bottleneck2.quantiles"] is None
    worst, worst_name, n = 0.0, "", 0
    for name, g in full.items():
        if g is None:
            assert all(s[name] is None for s in shards), name
            continue
        mean = (shards[0][name] + shards[1][name]) / 2
        e = float((mean - g).abs().max()) / (float(g.abs().max()) + 1e-30)
        n += 1
        if e > worst:
            worst, worst_name = e, name
    print(f"shard-mean vs full-batch gradients through the reducer: {n} tensors, worst relative error {worst:.2e} ({worst_name})")
    assert n == 164 and worst <= 1e-4, (n, worst, worst_name)
    # the aux loss is a function of the parameters only: identical on every shard, accumulates into the (None) quantile grads
    net.aux_loss().backward()
    assert net.entropy_bottleneck1.quantiles.grad is not None
    red.remove()
