#!/usr/bin/env python3
"""bench.py -- stereo pairs/s (enc+dec = one HSIC.forward) of the MI355X-native MASIC codec.

    python bench.py [--gpus N --steps K --warmup W] [--batch 8 --height 512 --width 512]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one HSIC(N=128,M=192,K=5) eval forward over a batch of synthetic stereo pairs already resident
in HBM (BASELINE.json configs[1]: 8 x 512x512 per GPU; weak scaling: every rank runs its own batch, no
data-path collective).  Rank 0 prints ONE JSON line with the contract fields plus
  roofline     -- the dominant kernel symbol (by device time): algorithmic FLOPs per launch / average launch
                  duration, both measured with HIP events on the launch stream inside the timed region
  cpu_baseline -- oracle/ (CPU restatement of the reference, kind "port") timed on this box's host cores on a
                  bounded sample (N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "coremasic", "mywork")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense f32 matrix peak
BF16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak (not the 2:1-sparsity figure)
HBM_PEAK_GBS = 8000.0


def cpu_baseline(N, M, K, H, W, seed):
    """Oracle (CPU float32 restatement pinned to the reference) on the host cores: 1 pair, 1 warm-up + 3 runs."""
    import MASIC
    from masic_amd import synth
    from oracle import hsic_oracle as O
    # the GPU box gives one GPU's job a 16-core share of its host CPUs; oversubscribing torch's pool
    # beyond that share (os.cpu_count() reports the whole machine) makes the CPU path slower, not faster
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    sd = synth.synth_state_dict(MASIC.HSIC(N, M, K).state_dict(), seed=seed)
    x1, x2, hm = synth.synth_inputs(1, H, W, seed=seed)
    with torch.no_grad():
        O.hsic_forward(sd, x1, x2, hm, K=K)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            O.hsic_forward(sd, x1, x2, hm, K=K)
            ts.append(time.perf_counter() - t0)
    ts.sort()
    return {"value": 1.0 / ts[1], "unit": "stereo pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/hsic_oracle.py eval forward, 1x3x{H}x{W} pair, median of 3 runs after 1 warm-up, torch CPU float32"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="stereo pairs per GPU per step")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--precision", choices=["bf16", "f32"], default=os.environ.get("MASIC_PRECISION", "bf16"),
                    help="operand precision of the forward MFMA contractions (float32 accumulate either way)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="issue the timed forward eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-codec", action="store_true", help="skip the compress / decompress timing of one pair (extras.bitstream)")
    ap.add_argument("--no-f32-compare", action="store_true", help="skip the float32 parity-path timing/accuracy extras (profiling runs)")
    ap.add_argument("--train-steps", type=int, default=3,
                    help="also time this many full training steps (forward + RD loss + backward + gradient all-reduce + "
                         "2x Adam) after the headline region; reported under extras.train_step, 0 to skip")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    # MASIC_BENCH_REHEARSAL=1: several ranks on fewer GPUs over gloo -- exercises the N>1 control flow on a one-GPU box (RCCL
    # refuses two ranks on one device); never a measurement
    rehearsal = os.environ.get("MASIC_BENCH_REHEARSAL", "0") == "1"
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI

    import MASIC
    from masic_amd import nn as mnn
    from masic_amd import ops, synth
    mnn.set_precision(args.precision)

    N, M, K = 128, 192, 5
    B, H, W = args.batch, args.height, args.width
    net = MASIC.HSIC(N, M, K)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100))
    net = net.to(dev).eval()
    x1, x2, hm = (t.to(dev) for t in synth.synth_inputs(B, H, W, seed=100 + rank))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from masic_amd.graph import GraphedHSIC
    with torch.no_grad():
        # The timed region replays a HIP graph of the eval forward (three streams, ~110 launches): per step the host only
        # evaluates the 3x3 sampling matrices (float32 chain, masic_amd/homography.py), copies them in and replays.
        step = net if args.no_graph else GraphedHSIC(net, x1, x2, hm)
        # the batch is resident in HBM in the graph's own input buffers (where an uploader / decoder would put it): no copy per step
        xa, xb = (x1, x2) if args.no_graph else step.inputs
        for _ in range(args.warmup):
            step(xa, xb, hm)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step(xa, xb, hm)
        barrier()
        elapsed = time.perf_counter() - t0
        # Roofline pass (not part of `value`): the same forward issued eagerly -- kernels inside a graph replay cannot be
        # bracketed individually -- first with HIP events around every conv launch to find the dominant kernel symbol and
        # the per-kernel split, then `steps` more with events around that symbol only.
        survey = ops.KernelTimer()
        ops.set_kernel_timer(survey)
        net(x1, x2, hm)
        ops.set_kernel_timer(None)
        survey_agg = survey.summary()
        dom = max(survey_agg, key=lambda k: survey_agg[k]["ms"])
        timer = ops.KernelTimer(only=dom)
        ops.set_kernel_timer(timer)
        for _ in range(args.steps):
            net(x1, x2, hm)
        torch.cuda.synchronize()
        ops.set_kernel_timer(None)
        # the same symbol with the forward on ONE stream: in the three-stream schedule its launches share the CUs with the
        # entropy chains of the side streams, which is what the step pays but not what the kernel can do
        solo = ops.KernelTimer(only=dom)
        net.serial_schedule = True
        ops.set_kernel_timer(solo)
        for _ in range(max(2, args.steps // 4)):
            net(x1, x2, hm)
        torch.cuda.synchronize()
        ops.set_kernel_timer(None)
        net.serial_schedule = False
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant conv kernel symbol by device time (HIP events on the launch stream, inside the timed region)
    agg = timer.summary()
    a = agg[dom]
    avg_ms = a["ms"] / a["launches"]
    tflops = a["flops"] / a["launches"] / (avg_ms * 1e-3) / 1e12
    peak = BF16_MFMA_PEAK_TFLOPS if ("bf16" in dom or "f16k" in dom) else F32_MFMA_PEAK_TFLOPS
    roofline = {"kernel": dom, "bound": "mfma", "achieved": tflops, "peak": peak, "unit": "TFLOP/s",
                "frac": tflops / peak, "traffic": None,
                "launches_per_step": a["launches"] / args.steps, "avg_launch_ms": avg_ms,
                "flops_per_launch": a["flops"] / a["launches"],
                "share_of_step_time": a["ms"] / (elapsed * 1e3),
                "timing": "HIP events on the launch stream around every launch of this symbol in an eager pass of the same "
                          "forward right after the timed region (graph replays cannot be bracketed per kernel); the launches "
                          "share the CUs with the side streams' kernels, as in the timed region",
                "isolated": (lambda b: {"avg_launch_ms": b["ms"] / b["launches"], "achieved": b["flops"] / b["ms"] / 1e9,
                                        "frac": b["flops"] / b["ms"] / 1e9 / peak,
                                        "note": "same symbol, same forward issued on one stream (nothing else on the CUs)"})(solo.summary()[dom]),
                "all_conv_kernels_ms_per_step_warmup_survey": {k: v["ms"] for k, v in survey_agg.items()}}
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            roofline["traffic"] = json.load(open(pmc)).get(dom)   # keyed by kernel symbol; null when not profiled for it
        except Exception:
            pass

    # ---- the float32 parity path on the same inputs: its rate, and what bf16 operands cost in codec terms
    accuracy = None
    f32_info = None
    if args.precision == "bf16" and not args.no_f32_compare:
        from masic_amd.loss import rate_distortion
        with torch.no_grad():
            sym_b = net.symbol_streams(x1, x2, hm)
            crit_b = rate_distortion(out, x1, x2, 0.01)
            mnn.set_precision("f32")
            out_f = net(x1, x2, hm)
            sym_f = net.symbol_streams(x1, x2, hm)
            crit_f = rate_distortion(out_f, x1, x2, 0.01)
            barrier()
            t0 = time.perf_counter()
            nf = max(2, args.steps // 4)
            for _ in range(nf):
                net(x1, x2, hm)
            barrier()
            tf = time.perf_counter() - t0
            mnn.set_precision("bf16")
        nsym = sum(v.numel() for v in sym_f.values())
        nbad = sum(int((sym_b[k] != sym_f[k]).sum()) for k in sym_f)
        accuracy = {"bpp_bf16": float(crit_b["bpp_loss"]), "bpp_f32": float(crit_f["bpp_loss"]),
                    "bpp_rel_delta": float(crit_b["bpp_loss"]) / float(crit_f["bpp_loss"]) - 1.0,
                    "psnr1_delta_db": crit_b["psnr1"] - crit_f["psnr1"], "psnr2_delta_db": crit_b["psnr2"] - crit_f["psnr2"],
                    "symbol_mismatch_rate": nbad / nsym, "symbols": nsym,
                    "note": "bf16-operand forward vs the float32 parity path (itself within 1e-4 of the reference, symbols bit-exact "
                            "outside the tie zone), same inputs and weights, rank 0"}
        f32_info = {"value": world * B * nf / tf, "unit": "stereo pairs/s", "steps": nf, "ms_per_step": tf / nf * 1e3, "dtype": "f32"}

    train_info = None
    if args.train_steps > 0:
        from masic_amd.parallel import GradientAllReducer
        from masic_amd.train import make_optimizers, train_step
        net.train()
        optimizer, aux_optimizer = make_optimizers(net)
        reducer = GradientAllReducer(net) if world > 1 else None
        train_step(net, optimizer, aux_optimizer, x1, x2, hm, 0.01, reducer)        # warm-up (packs dgrad weights etc.)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.train_steps):
            crit, _ = train_step(net, optimizer, aux_optimizer, x1, x2, hm, 0.01, reducer)
        barrier()
        tt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([tt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tt = float(t.item())
        train_info = {"value": world * B * args.train_steps / tt, "unit": "stereo pairs/s", "steps": args.train_steps,
                      "ms_per_step": tt / args.train_steps * 1e3, "loss_after": float(crit["loss"]),
                      "what": f"forward ({args.precision} operands) + RD loss + backward ({args.precision} operands, f32 accumulate)" + (" + RCCL gradient all-reduce" if world > 1 else "") +
                              " + Adam + aux loss backward + aux Adam (newtrain_codec_real.py:135-146)"}

    codec_info = None
    if rank == 0 and not args.no_codec:
        # the real bitstream of one pair: HSIC.compress / decompress on rank 0, outside every timed region above
        import shutil
        import tempfile
        net.eval()
        net.update(force=True)
        tmp = tempfile.mkdtemp()
        try:
            with torch.no_grad():
                net.compress(x1[:1], x2[:1], hm[:1], "warm", tmp)
                net.decompress(None, None, hm[:1], "warm", tmp)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                enc = net.compress(x1[:1], x2[:1], hm[:1], "pair", tmp)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                dec = net.decompress(None, None, hm[:1], "pair", tmp)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
            codec_info = {"encode_ms": (t1 - t0) * 1e3, "decode_ms": (t2 - t1) * 1e3, "bytes": enc["bytes"], "bpp_per_view_pixel": enc["bpp"],
                          "lossless_latents": all(bool(torch.equal(enc[k], dec[k])) for k in ("y1_hat", "y2_hat", "z1_hat", "z2_hat")),
                          "identical_reconstruction": all(bool(torch.equal(enc[k], dec[k])) for k in ("x1_hat", "x2_hat")),
                          "what": f"HSIC.compress / decompress of one {H}x{W} pair: wavefront-ordered context coding, GMM tables on the GPU, "
                                  "rANS on one host core (files included)"}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

    if rank == 0:
        line = {
            "metric": "stereo pairs/sec (enc+dec)", "value": world * B * args.steps / elapsed, "unit": "stereo pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"HSIC(N=128,M=192,K=5) eval forward (enc+dec both views), {B}x3x{H}x{W} stereo pairs per GPU "
                                   "(BASELINE.json configs[1] shape), inputs resident in HBM (the graph's static input buffers)",
                       "pairs_per_gpu": B, "height": H, "width": W, "parallelism": f"dp{world} (pairs sharded, no data-path collective)"},
            "roofline": roofline,
        }
        extras = {}
        if train_info is not None:
            extras["train_step"] = train_info
        if f32_info is not None:
            extras["f32_parity_path"] = f32_info
            extras["accuracy_vs_f32"] = accuracy
        if codec_info is not None:
            extras["bitstream"] = codec_info
        if extras:
            line["extras"] = extras
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N, M, K, H, W, seed=100)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
