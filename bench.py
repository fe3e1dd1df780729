#!/usr/bin/env python3
"""bench.py -- stereo pairs/s (enc+dec = one HSIC.forward) of the MI355X-native MASIC codec.

    python bench.py [--gpus N --steps K --warmup W] [--batch 8 --height 512 --width 512]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one HSIC(N=128,M=192,K=5) eval forward over a batch of synthetic stereo pairs already resident
in HBM (BASELINE.json configs[1]: 8 x 512x512 per GPU; weak scaling: every rank runs its own batch, no
data-path collective).  Rank 0 prints ONE JSON line with the contract fields plus
  roofline     -- the dominant kernel symbol (by device time): algorithmic FLOPs per launch / average launch
                  duration, both measured with HIP events on the launch stream; `traffic` = HBM bytes per launch
                  from two rocprofv3 --pmc child passes of this same script run before the parent touches the GPU
                  (N=1 only; FETCH_SIZE doubled, WRITE_SIZE at face value, MI355X_MICROARCH.md HBM section)
  cpu_baseline -- oracle/ (CPU restatement of the reference, kind "port") timed on this box's host cores on
                  bounded samples (N=1 only): eval forward and training step at the C1 / C2 picture sizes
  extras       -- training step, float32 parity path, accuracy of both operand modes against the oracle
                  (bpp / PSNR / symbol mismatch: BASELINE.json's "bpp/PSNR vs ref"), upload-inclusive rates,
                  Independent_EN (CQE) forward + training step, the real bitstream of one pair.
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "coremasic", "mywork")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense f32 matrix peak
BF16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak (not the 2:1-sparsity figure)
FP8_MFMA_PEAK_TFLOPS = 5000.0    # MI355X_MICROARCH.md: dense fp8 peak (block-scaled v_mfma_scale_f32_32x32x64_f8f6f4: 2x bf16 per clock)
HBM_PEAK_GBS = 8000.0
HSIC_GFLOP_PER_PAIR_512 = 161.12   # SURVEY.md 8(d), forward, 512x512; proportional to H*W
CQE_GFLOP_PER_PAIR_512 = 817.5
KERNEL_SOURCES = ("conv_f16k.hip", "conv.hip", "conv_geom.h", "common.h", "gemm_bf16.hip")


def kernel_source_stamp():
    """sha256 over the kernel sources a PMC figure depends on: a stored figure is only reported for the sources it was taken on."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "masic_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def host_cores():
    # the GPU box gives one GPU's job a 16-core share of its host CPUs; oversubscribing torch's pool beyond that share
    # (os.cpu_count() reports the whole machine) makes the CPU path slower, not faster
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 16))


# ------------------------------------------------------------------------------------------------ the reference drivers' loop shape
def driver_loop_pieces():
    """What an UNCHANGED driver brings: its own plain-torch criterion and optimizers (coremasic/mywork/newtrain_codec_real.py:66-87,
    :434-435; newtrain_cqe_real.py:69-96 kind=0 without the ms_ssim report entries, :472).  Nothing of masic_amd.train / masic_amd.loss."""
    import math
    import torch.nn as nn

    class RateDistortionLoss(nn.Module):
        def __init__(self, lmbda=1e-2, rate=True):
            super().__init__()
            self.mse = nn.MSELoss()
            self.lmbda = lmbda
            self.rate = rate

        def forward(self, output, target1, target2):
            N, _, H, W = target1.size()
            out = {}
            num_pixels = N * H * W
            out['mse_loss'] = self.mse(output['x1_hat'], target1) + self.mse(output['x2_hat'], target2)
            out['loss'] = self.lmbda * 255 ** 2 * out['mse_loss']
            if self.rate:
                out['bpp_loss'] = sum((torch.log(likelihoods).sum() / (-math.log(2) * num_pixels)) for likelihoods in output['likelihoods'].values())
                out['loss'] = out['loss'] + out['bpp_loss']
            out['psnr1'] = 10 * math.log10(1 / self.mse(output['x1_hat'], target1))
            out['psnr2'] = 10 * math.log10(1 / self.mse(output['x2_hat'], target2))
            return out

    def codec_iteration(i, model, criterion, optimizer, aux_optimizer, d1, d2, h_matrix):       # newtrain_codec_real.py:132-161
        optimizer.zero_grad()
        aux_optimizer.zero_grad()
        out_net = model(d1, d2, h_matrix)
        out_criterion = criterion(out_net, d1, d2)
        out_criterion['loss'].backward()
        optimizer.step()
        aux_loss = model.aux_loss()
        aux_loss.backward()
        aux_optimizer.step()
        if i % 10 == 0:
            return (out_criterion["loss"].item(), out_criterion["mse_loss"].item(), out_criterion["bpp_loss"].item(), aux_loss.item())
        return out_criterion['loss']

    def cqe_iteration(i, model, model2, criterion, optimizer, aux_optimizer, d1, d2, h_matrix):   # newtrain_cqe_real.py:152-180
        optimizer.zero_grad()
        aux_optimizer.zero_grad()
        out_net = model(d1, d2, h_matrix)            # eval mode, grad enabled: as the driver runs it
        out_net2 = model2(out_net['x1_hat'], out_net['x2_hat'], h_matrix)
        out_criterion = criterion(out_net2, d1, d2)
        out_criterion['loss'].backward()
        optimizer.step()
        aux_loss = model.aux_loss()
        if i % 10 == 0:
            return (out_criterion["loss"].item(), out_criterion["mse_loss"].item(), aux_loss.item())
        return out_criterion['loss']

    return RateDistortionLoss, codec_iteration, cqe_iteration


def rehearsal_reducer_check(dev, rank, world):
    """MASIC_BENCH_REHEARSAL only (several ranks on one GPU over gloo): the reducer sees world > 1 on the REAL model once --
    HSIC(16,24,3) gradients of a 2*world-pair batch == the all-reduced mean over the ranks' 2-pair shards (each rank checks)."""
    import MASIC
    from compressai.entropy_models import EntropyModel
    from masic_amd import synth
    from masic_amd.loss import rate_distortion
    from masic_amd.parallel import GradientAllReducer, shard_range
    NOISE_KEYS = ("z1", "y1_ctx", "y1", "z2", "y2_ctx", "y1_warp", "y2")     # the seven draws of a training forward, in order
    N, M, K, H, W = 16, 24, 3, 64, 64
    B = 2 * world
    net = MASIC.HSIC(N, M, K)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=77))
    net = net.to(dev).train()
    x1, x2, hm = (t.to(dev) for t in synth.synth_inputs(B, H, W, seed=77))
    noise = {k: v.to(dev) for k, v in synth.synth_noise(B, N, M, H, W, seed=77).items()}
    hw_z = (H // 64) * (W // 64)

    def run(lo, hi, red):
        queue = [noise[k][:, :, lo * hw_z:hi * hw_z].contiguous() if k[0] == "z" else noise[k][lo:hi].contiguous() for k in NOISE_KEYS]
        orig = EntropyModel._get_noise_cached
        EntropyModel._get_noise_cached = lambda self, x: queue.pop(0).reshape(x.shape)
        try:
            net.zero_grad()
            if red is not None:
                red.arm()
            out = net(x1[lo:hi].contiguous(), x2[lo:hi].contiguous(), hm[lo:hi].contiguous())
            rate_distortion(out, x1[lo:hi].contiguous(), x2[lo:hi].contiguous(), 0.01)["loss"].backward()
            if red is not None:
                red.finish()
        finally:
            EntropyModel._get_noise_cached = orig
        return {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in net.named_parameters()}

    full = run(0, B, None)
    red = GradientAllReducer(net, bucket_bytes=256 << 10)
    worst = 0.0
    for _ in range(2):                 # second step: learnt no-gradient set, overlapped launches from the hooks
        mine = run(*shard_range(B, rank, world), red)
        for n, g in full.items():
            assert (g is None) == (mine[n] is None), n
            if g is not None:
                worst = max(worst, float((mine[n] - g).abs().max()) / (float(g.abs().max()) + 1e-30))
    red.remove()
    assert worst <= 1e-4, f"rank {rank}: all-reduced shard-mean gradients differ from the full-batch gradients by {worst:.2e}"
    return {"world": world, "model": "HSIC(16,24,3)", "pairs": B, "buckets": len(red.buckets), "worst_rel_err": worst}


# ------------------------------------------------------------------------------------------------ CPU legs (oracle = checker)
def cpu_baseline(N, M, K, seed, budget_s=45.0):
    """Oracle (CPU float32 restatement pinned to the reference, kind "port") on the host cores.  BASELINE.md section 4 legs:
    eval forward and training step (forward + RD loss + backward + Adam + aux loss + aux Adam: newtrain_codec_real.py:135-146)
    at the C1 (256x256) and C2 (512x512) picture sizes, B = 1 and B = 8 for the eval forward; each leg = median of up to 3 runs
    after one warm-up, bounded by a wall-clock budget (legs that no longer fit are skipped and say so).  Also returns the
    oracle's outputs on the 1x512x512 sample for extras.accuracy_vs_ref."""
    import MASIC
    from masic_amd import synth
    from oracle import hsic_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = synth.synth_state_dict(MASIC.HSIC(N, M, K).state_dict(), seed=seed)
    t_start = time.perf_counter()
    legs = {}

    def timed(fn, runs=3):
        fn()
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > budget_s:
                break
        ts.sort()
        return ts[len(ts) // 2], len(ts)

    def eval_leg(B, H, W):
        x1, x2, hm = synth.synth_inputs(B, H, W, seed=seed)

        def run():
            with torch.no_grad():
                return O.hsic_forward(sd, x1, x2, hm, K=K)
        t, n = timed(run)
        return {"value": B / t, "unit": "stereo pairs/s", "seconds_per_step": t, "runs": n, "shape": f"{B}x3x{H}x{W}"}

    names = [n for n, _ in MASIC.HSIC(N, M, K).named_parameters()]

    def train_leg(B, H, W):
        x1, x2, hm = synth.synth_inputs(B, H, W, seed=seed)
        noise = synth.synth_noise(B, N, M, H, W, seed=seed)
        psd = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in sd.items()}
        opt = torch.optim.Adam([psd[n] for n in names if not n.startswith("entropy_bottleneck")], lr=1e-4)
        aopt = torch.optim.Adam([psd[n] for n in names if n.startswith("entropy_bottleneck")], lr=1e-3)

        def run():
            opt.zero_grad(); aopt.zero_grad()
            out = O.hsic_forward(psd, x1, x2, hm, K=K, training=True, noise=noise)
            O.rd_loss(out, x1, x2, 0.01)["loss"].backward()
            opt.step()
            (O.eb_aux_loss(psd, "entropy_bottleneck1") + O.eb_aux_loss(psd, "entropy_bottleneck2")).backward()
            aopt.step()
        t, n = timed(run, runs=2)
        return {"value": B / t, "unit": "stereo pairs/s", "seconds_per_step": t, "runs": n, "shape": f"{B}x3x{H}x{W}"}

    plan = [("eval_c2_b1", eval_leg, (1, 512, 512)), ("eval_c1_b1", eval_leg, (1, 256, 256)), ("eval_c2_b8", eval_leg, (8, 512, 512)),
            ("train_c1_b1", train_leg, (1, 256, 256)), ("train_c2_b1", train_leg, (1, 512, 512))]
    for name, fn, a in plan:
        if time.perf_counter() - t_start > budget_s:
            legs[name] = {"skipped": f"CPU budget of {budget_s:.0f} s spent"}
            continue
        legs[name] = fn(*a)
    head = legs.get("eval_c2_b8") if "value" in legs.get("eval_c2_b8", {}) else legs["eval_c2_b1"]
    return {"value": head["value"], "unit": "stereo pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/hsic_oracle.py (torch CPU float32, {torch.get_num_threads()} threads) eval forward on {head['shape']} synthetic pairs, "
                      f"median of {head['runs']} runs after 1 warm-up; all legs (eval / training step at the C1 and C2 picture sizes) under `legs`",
            "legs": legs, "seconds_total": time.perf_counter() - t_start}


def oracle_reference_sample(N, M, K, H, W, seed):
    """The oracle's outputs, RD scalars and symbol streams on one HxW pair (the checker for extras.accuracy_vs_ref)."""
    import MASIC
    from masic_amd import synth
    from oracle import hsic_oracle as O
    torch.set_num_threads(host_cores())
    sd = synth.synth_state_dict(MASIC.HSIC(N, M, K).state_dict(), seed=seed)
    x1, x2, hm = synth.synth_inputs(1, H, W, seed=seed)
    with torch.no_grad():
        ref = O.hsic_forward(sd, x1, x2, hm, K=K, keep=True)
    crit = O.rd_loss(ref, x1, x2, 0.01)
    return {"inputs": (x1, x2, hm), "sym": O.symbols(ref["_aux"], sd), "x1_hat": ref["x1_hat"], "x2_hat": ref["x2_hat"],
            "bpp": float(crit["bpp_loss"]), "psnr1": float(crit["psnr1"]), "psnr2": float(crit["psnr2"])}


# ------------------------------------------------------------------------------------------------ HBM traffic (PMC child passes)
def pmc_child(args):
    """`bench.py --pmc-child`: the bf16 eval forward issued eagerly a few times, nothing else (run under rocprofv3 --pmc)."""
    import MASIC
    from masic_amd import nn as mnn, synth
    mnn.set_precision(args.precision)
    dev = torch.device("cuda", 0)
    net = MASIC.HSIC(128, 192, 5)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100))
    net = net.to(dev).eval()
    x1, x2, hm = (t.to(dev) for t in synth.synth_inputs(args.batch, args.height, args.width, seed=100))
    if args.precision == "fp8":
        from masic_amd import fp8
        fp8.calibrate(net, [tuple(t.to(dev) for t in synth.synth_inputs(2, args.height, args.width, seed=1100))])
    with torch.no_grad():
        for _ in range(3):
            net(x1, x2, hm)
    torch.cuda.synchronize()


def measure_pmc_traffic(args, timeout_s=240):
    """{kernel symbol: {"FETCH_SIZE": KiB/launch, "WRITE_SIZE": KiB/launch, "launches": n}} from two `rocprofv3 --pmc` passes
    of `bench.py --pmc-child` (separate passes: the two counters do not fit one; no tracing domains besides the kernel trace),
    or (None, reason).  Must run BEFORE this process initialises the GPU."""
    import csv
    import glob
    import re
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="masic_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    agg = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out_dir, "-o", "p", "--", sys.executable, os.path.join(ROOT, "bench.py"),
                   "--pmc-child", "--precision", args.precision, "--batch", str(args.batch), "--height", str(args.height), "--width", str(args.width)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} pass failed (rc {r.returncode}): {r.stdout.decode(errors='replace')[-300:]}"
            for f in files:
                for row in csv.DictReader(open(f)):
                    name = re.sub(r"\(.*", "", row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
                    a = agg.setdefault(name, {}).setdefault(row["Counter_Name"], [0, 0.0])
                    a[0] += 1
                    a[1] += float(row["Counter_Value"])
    except Exception as e:            # never let the measurement of an extra take the bench down
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {n: {"launches": max(v[0] for v in c.values()), **{cn: v[1] / v[0] for cn, v in c.items()}} for n, c in agg.items()}, None


def traffic_for(symbol, pmc, pmc_reason):
    """(bytes per launch or None, how it was obtained)."""
    if pmc is not None and symbol in pmc and "FETCH_SIZE" in pmc[symbol] and "WRITE_SIZE" in pmc[symbol]:
        e = pmc[symbol]
        return (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0, ("measured by this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of `bench.py --pmc-child` "
                                                                   f"(eager forward, {e['launches']} launches); bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024")
    if pmc is not None:
        pmc_reason = f"the PMC passes ran but hold no kernel named {symbol!r}"
    stored = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(stored))
        e = d.get(symbol)
        if isinstance(e, dict) and e.get("source_stamp") == kernel_source_stamp():
            return float(e["bytes"]), f"profiles/pmc_traffic.json (taken on kernel sources {e['source_stamp']}, commit {e.get('commit')}); live pass unavailable: {pmc_reason}"
        if isinstance(e, dict):
            return None, f"stored figure is for kernel sources {e.get('source_stamp')}, current sources are {kernel_source_stamp()}: nulled; live pass unavailable: {pmc_reason}"
    except Exception:
        pass
    return None, f"not measured: {pmc_reason}"


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="stereo pairs per GPU per step")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--precision", choices=["bf16", "f32", "fp8"], default=os.environ.get("MASIC_PRECISION", "bf16"),
                    help="operand precision of the forward MFMA contractions (float32 accumulate either way); fp8 = BASELINE configs[4]: e4m3 "
                         "operands on the 128->128 5x5 layers and the first two layers of the entropy-parameter stacks (masic_amd/fp8.py)")
    ap.add_argument("--no-fp8", action="store_true", help="skip the fp8-operand timing / accuracy extras of a bf16 run (extras.fp8_path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="issue the timed forward eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-codec", action="store_true", help="skip the compress / decompress timing of one pair (extras.bitstream)")
    ap.add_argument("--no-f32-compare", action="store_true", help="skip the float32 parity-path timing/accuracy extras (profiling runs)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes behind roofline.traffic")
    ap.add_argument("--no-cqe", action="store_true", help="skip the Independent_EN (CQE) forward / training-step extras")
    ap.add_argument("--no-upload", action="store_true", help="skip the upload-inclusive timings (extras.with_upload)")
    ap.add_argument("--no-trained", action="store_true", help="skip extras.accuracy_vs_ref_trained (trains HSIC for --trained-steps steps first)")
    ap.add_argument("--trained-steps", type=int, default=3000)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--train-steps", type=int, default=5,
                    help="also time this many full training steps (forward + RD loss + backward + gradient all-reduce + "
                         "2x Adam) after the headline region; reported under extras.train_step, 0 to skip")
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")

    # HBM traffic of the kernels: child processes under rocprofv3, before anything here touches the GPU
    pmc, pmc_reason = None, "skipped (--no-pmc or N > 1)"
    if world == 1 and not args.no_pmc:
        t0 = time.perf_counter()
        pmc, pmc_reason = measure_pmc_traffic(args)
        pmc_seconds = time.perf_counter() - t0

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
    from masic_amd.loss import distortion, rate_distortion
    rehearsal_info = rehearsal_reducer_check(dev, rank, world) if (rehearsal and world > 1) else None
    mnn.set_precision(args.precision)

    N, M, K = 128, 192, 5
    B, H, W = args.batch, args.height, args.width
    net = MASIC.HSIC(N, M, K)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100))
    net = net.to(dev).eval()
    x1h, x2h, hmh = synth.synth_inputs(B, H, W, seed=100 + rank)
    x1, x2, hm = x1h.to(dev), x2h.to(dev), hmh.to(dev)
    fp8_table = None
    if args.precision == "fp8" or (args.precision == "bf16" and not args.no_fp8):
        # activation scales of the fp8 mode from ONE calibration batch that is not the measured one (same on every rank)
        from masic_amd import fp8
        fp8_table = fp8.calibrate(net, [tuple(t.to(dev) for t in synth.synth_inputs(2, H, W, seed=1100))])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(t):
        if world > 1:
            v = torch.tensor([t], dtype=torch.float64, device=dev)
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
            return float(v.item())
        return t

    from masic_amd.graph import GraphedHSIC
    with torch.no_grad():
        # The timed region replays a HIP graph of the eval forward (three streams, ~110 launches): per step the host only
        # evaluates the 3x3 sampling matrices (float32 chain, masic_amd/homography.py), copies them in and replays.
        step = net if args.no_graph else GraphedHSIC(net, x1, x2, hm)
        # the batch is resident in HBM in the graph's own input buffers (where an uploader / decoder would put it): no copy per step
        xa, xb = (x1, x2) if args.no_graph else step.inputs
        # the homography of the next batch is handed over with the current one (a loader one batch ahead): its device -> host read,
        # the host-side float32 chain and the upload of the sampling matrices run under the current replay; every step still does all of it
        ahead = {} if args.no_graph else {"next_h_matrix": hm}
        for _ in range(args.warmup):
            step(xa, xb, hm, **ahead)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step(xa, xb, hm, **ahead)
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)

        # ---- the same step WITHOUT the look-ahead: the homography arrives with its own batch, as the reference computes it right before
        # model(d1, d2, h_matrix) (newtrain_codec_real.py:124-137) -- each call waits for the previous replay, reads it, runs the host chain
        no_lookahead = None
        if not args.no_graph:
            for _ in range(2):
                step(xa, xb, hm)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step(xa, xb, hm)
            barrier()
            t_nl = max_over_ranks(time.perf_counter() - t0)
            no_lookahead = {"value": world * B * args.steps / t_nl, "unit": "stereo pairs/s", "ms_per_step": t_nl / args.steps * 1e3, "steps": args.steps,
                            "what": "the headline step called as step(x1, x2, h_matrix) with no next_h_matrix: device->host read of the homography, float32 host chain, "
                                    "upload, replay -- strictly in sequence"}
            # ... and with the homography handed over as a CPU tensor (where a DataLoader leaves it): no device -> host read, the host chain runs
            # at once and only its 144-byte upload is waited for
            hm_cpu = hm.cpu()
            for _ in range(2):
                step(xa, xb, hm_cpu)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step(xa, xb, hm_cpu)
            barrier()
            t_nc = max_over_ranks(time.perf_counter() - t0)
            no_lookahead["h_matrix_on_the_host"] = {"value": world * B * args.steps / t_nc, "ms_per_step": t_nc / args.steps * 1e3}

        # ---- upload-inclusive rates (never `value`): the boundary takes device pointers; a caller that owns host batches pays PCIe
        upload = None
        if not args.no_upload and not args.no_graph:
            x1p, x2p = x1h.pin_memory(), x2h.pin_memory()
            nb = 2 * x1p.numel() * 4
            cur = torch.cuda.current_stream()
            # (a) synchronous: upload on the compute stream, then the step
            for _ in range(2):
                xa.copy_(x1p, non_blocking=True); xb.copy_(x2p, non_blocking=True); step(xa, xb, hm, **ahead)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                xa.copy_(x1p, non_blocking=True)
                xb.copy_(x2p, non_blocking=True)
                step(xa, xb, hm, **ahead)
            barrier()
            t_sync = max_over_ranks(time.perf_counter() - t0)
            # (b) overlapped: batch n+1 goes up on a copy stream into the other of two staging buffers while batch n runs;
            #     the compute stream waits for its batch's upload event and copies it into the graph's inputs (device to device)
            cs = torch.cuda.Stream()
            stage = [(torch.empty_like(x1), torch.empty_like(x2)) for _ in range(2)]
            up = [torch.cuda.Event(), torch.cuda.Event()]
            free = [torch.cuda.Event(), torch.cuda.Event()]
            for e in free:
                e.record(cur)

            def upload_into(p):
                cs.wait_event(free[p])
                with torch.cuda.stream(cs):
                    stage[p][0].copy_(x1p, non_blocking=True)
                    stage[p][1].copy_(x2p, non_blocking=True)
                    up[p].record(cs)

            def run_overlapped(n):
                upload_into(0)
                for i in range(n):
                    p = i & 1
                    if i + 1 < n:
                        upload_into(1 - p)
                    cur.wait_event(up[p])
                    step(stage[p][0], stage[p][1], hm, **ahead)
                    free[p].record(cur)
            run_overlapped(3)
            barrier()
            t0 = time.perf_counter()
            run_overlapped(args.steps)
            barrier()
            t_ovl = max_over_ranks(time.perf_counter() - t0)
            upload = {"bytes_per_step": nb, "synchronous": {"value": world * B * args.steps / t_sync, "ms_per_step": t_sync / args.steps * 1e3},
                      "overlapped": {"value": world * B * args.steps / t_ovl, "ms_per_step": t_ovl / args.steps * 1e3},
                      "unit": "stereo pairs/s",
                      "what": "the same graph replay with each step's batch uploaded from pinned host memory: on the compute stream (synchronous) / "
                              "on a copy stream into double-buffered staging, one batch ahead (overlapped)"}
            xa.copy_(x1); xb.copy_(x2)
            out = step(xa, xb, hm)

        # Roofline pass (not part of `value`): the same forward issued eagerly -- kernels inside a graph replay cannot be
        # bracketed individually -- first with HIP events around every conv launch to find the dominant kernel symbol and
        # the per-kernel split, then `steps` more with events around that symbol only.
        # (the survey is three passes and a kernel's time the MEDIAN of its three pass totals: with events between every launch of a
        # three-stream forward a side-stream kernel's interval now and then includes a long wait for the CUs -- one pass once made a
        # 40-us hyper-synthesis layer "dominant" at 0.9 ms)
        passes = []
        for _ in range(3):
            survey = ops.KernelTimer()
            ops.set_kernel_timer(survey)
            net(x1, x2, hm)
            ops.set_kernel_timer(None)
            passes.append(survey.summary())
        survey_agg = {}
        for k in passes[0]:
            if all(k in p_ for p_ in passes):
                survey_agg[k] = sorted((p_[k] for p_ in passes), key=lambda v: v["ms"])[1]
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

    # dominant conv kernel symbol by device time (HIP events on the launch stream)
    agg = timer.summary()
    a = agg[dom]
    avg_ms = a["ms"] / a["launches"]
    tflops = a["flops"] / a["launches"] / (avg_ms * 1e-3) / 1e12
    peak = FP8_MFMA_PEAK_TFLOPS if (dom.startswith("conv_f16k<") and dom.endswith(", true>")) else (BF16_MFMA_PEAK_TFLOPS if ("bf16" in dom or "f16k" in dom) else F32_MFMA_PEAK_TFLOPS)
    traffic, traffic_source = traffic_for(dom, pmc, pmc_reason)
    roofline = {"kernel": dom, "bound": "mfma", "achieved": tflops, "peak": peak, "unit": "TFLOP/s",
                "frac": tflops / peak, "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": a["bytes"] / a["launches"] / (4.0 if peak == FP8_MFMA_PEAK_TFLOPS else 2.0 if peak == BF16_MFMA_PEAK_TFLOPS else 1.0),
                "launches_per_step": a["launches"] / args.steps, "avg_launch_ms": avg_ms,
                "flops_per_launch": a["flops"] / a["launches"],
                "share_of_step_time": a["ms"] / (elapsed * 1e3),
                "end_to_end": {"achieved": B * HSIC_GFLOP_PER_PAIR_512 * (H * W) / (512 * 512) / 1e3 / (elapsed / args.steps), "unit": "TFLOP/s",
                               "frac": B * HSIC_GFLOP_PER_PAIR_512 * (H * W) / (512 * 512) / 1e3 / (elapsed / args.steps) / peak,
                               "note": "whole step: SURVEY 8(d) forward FLOPs per pair x pairs / ms_per_step"},
                "timing": "HIP events on the launch stream around every launch of this symbol in an eager pass of the same "
                          "forward right after the timed region (graph replays cannot be bracketed per kernel); the launches "
                          "share the CUs with the side streams' kernels, as in the timed region",
                "isolated": (lambda b: {"avg_launch_ms": b["ms"] / b["launches"], "achieved": b["flops"] / b["ms"] / 1e9,
                                        "frac": b["flops"] / b["ms"] / 1e9 / peak,
                                        "note": "same symbol, same forward issued on one stream (nothing else on the CUs)"})(solo.summary()[dom]),
                "all_conv_kernels_ms_per_step_warmup_survey": {k: v["ms"] for k, v in survey_agg.items()}}
    if pmc is not None:
        roofline["pmc_seconds"] = pmc_seconds
        top = sorted(((n, (2.0 * e.get("FETCH_SIZE", 0.0) + e.get("WRITE_SIZE", 0.0)) * 1024.0 * e["launches"]) for n, e in pmc.items()), key=lambda t: -t[1])[:6]
        roofline["pmc_hbm_bytes_by_kernel_3_forwards"] = {n: b for n, b in top}
        # the other kernels VERDICT round 2 asked a traffic ratio for: measured bytes per launch (same PMC passes) against algorithmic bytes
        px = B * (H // 16) * (W // 16)
        Mh = M

        def head_bytes(cin, couts, shared):      # one grouped launch: activations (bf16 F16K) + weights + outputs (F16K, the last layer float32 NCHW)
            return lambda last: ((px * cin * 2 if shared else sum(px * c * 2 for c in cin)) + sum((cin if shared else c_in) * co * 2 for c_in, co in zip(cin if not shared else [cin] * 3, couts))
                                 + sum(px * co * (4 if last else 2) for co in couts))
        launches = [head_bytes(4 * Mh, [6 * Mh] * 3, True)(False), head_bytes([6 * Mh] * 3, [4 * Mh, 4 * Mh, K * Mh], False)(False),
                    head_bytes([4 * Mh, 4 * Mh, K * Mh], [K * Mh] * 3, False)(True),
                    head_bytes(5 * Mh, [6 * Mh] * 3, True)(False), head_bytes([6 * Mh] * 3, [4 * Mh, 4 * Mh, K * Mh], False)(False),
                    head_bytes([4 * Mh, 4 * Mh, K * Mh], [K * Mh] * 3, False)(True)]
        alg = {"gemm_f16k2": sum(launches) / len(launches), "gemm_f16k<2, false>": sum(launches) / len(launches),
               "conv_a_gdn_f16k_w": B * 3 * H * W * 4 + B * 128 * (H // 2) * (W // 2) * 2, "conv_a_gdn_f16k<false, 0>": B * 3 * H * W * 4 + B * 128 * (H // 2) * (W // 2) * 2,
               # the transposed 128 -> 128 layer with its fused IGDN at (H/4)^2 -> (H/2)^2 (two launches per forward): input + output in bf16 F16K
               "conv_f16k<2, 2, 2, 5, 1, true, 4, 2, false>": B * N * ((H // 4) * (W // 4) + (H // 2) * (W // 2)) * 2}
        roofline["traffic_vs_algorithmic_other_kernels"] = {
            n: {"launches": pmc[n]["launches"], "traffic_bytes_per_launch": (2.0 * pmc[n].get("FETCH_SIZE", 0.0) + pmc[n].get("WRITE_SIZE", 0.0)) * 1024.0,
                "algorithmic_bytes_per_launch": float(ab), "ratio": (2.0 * pmc[n].get("FETCH_SIZE", 0.0) + pmc[n].get("WRITE_SIZE", 0.0)) * 1024.0 / float(ab)}
            for n, ab in alg.items() if n in pmc}

    # ---- accuracy of both operand modes against the ORACLE on one pair of the headline size, and the float32 path's rate
    accuracy_ref, accuracy, f32_info = None, None, None
    want_ref = world == 1 and not args.no_cpu_baseline
    if want_ref:
        ref = oracle_reference_sample(N, M, K, H, W, seed=100)
        r1, r2, rh = (t.to(dev) for t in ref["inputs"])
        accuracy_ref = {"sample": f"1x3x{H}x{W} pair (synth seed 100), oracle = oracle/hsic_oracle.py on the host (pinned to the reference)",
                        "oracle": {"bpp": ref["bpp"], "psnr1": ref["psnr1"], "psnr2": ref["psnr2"]}}
        with torch.no_grad():
            for mode in ((["fp8"] if fp8_table is not None else []) + ["bf16", "f32"] if args.precision != "f32" else ["f32"]):
                mnn.set_precision(mode)
                o = net(r1, r2, rh)
                s = net.symbol_streams(r1, r2, rh)
                c = rate_distortion(o, r1, r2, 0.01)
                nsym = sum(v.numel() for v in ref["sym"].values())
                nbad = sum(int((s[k].cpu() != ref["sym"][k]).sum()) for k in ref["sym"])
                accuracy_ref[mode] = {"bpp": float(c["bpp_loss"]), "bpp_rel_delta": float(c["bpp_loss"]) / ref["bpp"] - 1.0,
                                      "psnr1_delta_db": c["psnr1"] - ref["psnr1"], "psnr2_delta_db": c["psnr2"] - ref["psnr2"],
                                      "symbol_mismatch_rate": nbad / nsym, "symbol_mismatches": nbad, "symbols": nsym,
                                      "x1_hat_max_rel_err": float((o["x1_hat"].cpu() - ref["x1_hat"]).abs().max() / ref["x1_hat"].abs().max()),
                                      "x2_hat_max_rel_err": float((o["x2_hat"].cpu() - ref["x2_hat"]).abs().max() / ref["x2_hat"].abs().max())}
            mnn.set_precision(args.precision)
    fp8_info = None
    if args.precision == "bf16" and fp8_table is not None:
        # the same step with fp8 operands (BASELINE configs[4]): its own graph, same batch, same timing protocol
        mnn.set_precision("fp8")
        with torch.no_grad():
            g8 = GraphedHSIC(net, x1, x2, hm)
            a8, b8 = g8.inputs
            for _ in range(args.warmup):
                g8(a8, b8, hm, next_h_matrix=hm)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                g8(a8, b8, hm, next_h_matrix=hm)
            barrier()
            t8 = max_over_ranks(time.perf_counter() - t0)
            del g8
        mnn.set_precision("bf16")
        fp8_info = {"value": world * B * args.steps / t8, "unit": "stereo pairs/s", "steps": args.steps, "ms_per_step": t8 / args.steps * 1e3, "dtype": "fp8",
                    "what": "the headline step with e4m3 operands (v_mfma_scale_f32_32x32x64_f8f6f4) on g_a_conv2/3, g_s_conv2/3 and the first two layers of the "
                            "entropy-parameter stacks (67 % of the forward's FLOPs), per-tensor activation scales from one calibration batch; accuracy against "
                            "the oracle under extras.accuracy_vs_ref.fp8; `bench.py --precision fp8` makes it the headline with its own roofline"}
    if args.precision in ("bf16", "fp8") and not args.no_f32_compare:
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
            tf = max_over_ranks(time.perf_counter() - t0)
            mnn.set_precision(args.precision)
        nsym = sum(v.numel() for v in sym_f.values())
        nbad = sum(int((sym_b[k] != sym_f[k]).sum()) for k in sym_f)
        accuracy = {"bpp_bf16": float(crit_b["bpp_loss"]), "bpp_f32": float(crit_f["bpp_loss"]),
                    "bpp_rel_delta": float(crit_b["bpp_loss"]) / float(crit_f["bpp_loss"]) - 1.0,
                    "psnr1_delta_db": crit_b["psnr1"] - crit_f["psnr1"], "psnr2_delta_db": crit_b["psnr2"] - crit_f["psnr2"],
                    "symbol_mismatch_rate": nbad / nsym, "symbols": nsym,
                    "note": f"{args.precision}-operand forward (keys say bf16) vs the float32 parity path on the whole bench batch, rank 0 (the comparison "
                            "against the ORACLE is extras.accuracy_vs_ref)"}
        f32_info = {"value": world * B * nf / tf, "unit": "stereo pairs/s", "steps": nf, "ms_per_step": tf / nf * 1e3, "dtype": "f32"}

    # ---- Independent_EN (CQE): forward rate + its dominant kernel, and the CQE training step (BASELINE configs[2] stage)
    cqe_info = None
    if not args.no_cqe:
        en = MASIC.Independent_EN()
        en.load_state_dict(synth.synth_state_dict(en.state_dict(), seed=101))
        en = en.to(dev).eval()
        with torch.no_grad():
            xh1, xh2 = out["x1_hat"].clone(), out["x2_hat"].clone()
            for _ in range(2):
                en(xh1, xh2, hm)
            barrier()
            t0 = time.perf_counter()
            ne = max(3, args.steps // 4)
            for _ in range(ne):
                en(xh1, xh2, hm)
            barrier()
            te = max_over_ranks(time.perf_counter() - t0)
            sv = ops.KernelTimer()
            ops.set_kernel_timer(sv)
            en(xh1, xh2, hm)
            ops.set_kernel_timer(None)
            sva = sv.summary()
        edom = max(sva, key=lambda k: sva[k]["ms"])
        ea = sva[edom]
        epeak = BF16_MFMA_PEAK_TFLOPS if ("bf16" in edom or "f16k" in edom) else F32_MFMA_PEAK_TFLOPS
        gf = B * CQE_GFLOP_PER_PAIR_512 * (H * W) / (512 * 512)
        cqe_info = {"forward": {"value": world * B * ne / te, "unit": "stereo pairs/s", "ms_per_step": te / ne * 1e3, "steps": ne,
                                "achieved_tflops": gf / 1e3 / (te / ne), "frac_of_mfma_peak": gf / 1e3 / (te / ne) / epeak,
                                "dominant_kernel": {"kernel": edom, "launches": ea["launches"], "avg_launch_ms": ea["ms"] / ea["launches"],
                                                    "achieved": ea["flops"] / ea["ms"] / 1e9, "peak": epeak, "unit": "TFLOP/s",
                                                    "frac": ea["flops"] / ea["ms"] / 1e9 / epeak, "share_of_forward_time": ea["ms"] / (te / ne * 1e3)},
                                "what": f"Independent_EN eval forward on the codec's reconstructions, {B}x3x{H}x{W} per GPU, {args.precision} operands, "
                                        f"{CQE_GFLOP_PER_PAIR_512} GFLOP/pair at 512x512 (SURVEY 8d)"}}
        if args.train_steps > 0:
            from masic_amd.parallel import GradientAllReducer
            from masic_amd.train import cqe_train_step
            en.train()
            opt2 = torch.optim.Adam(list(en.parameters()), lr=1e-4, fused=True)      # (masic_amd/fresh.py's step hook bumps the version counters a fused step leaves alone)
            red2 = GradientAllReducer(en) if world > 1 else None
            for _ in range(2):                        # warm-up: weight packs of both kinds, optimizer state, allocator pools
                cqe_train_step(net, en, opt2, x1, x2, hm, 0.01, red2)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.train_steps):
                crit2, _ = cqe_train_step(net, en, opt2, x1, x2, hm, 0.01, red2)
            barrier()
            tc = max_over_ranks(time.perf_counter() - t0)
            cqe_info["train_step"] = {"value": world * B * args.train_steps / tc, "unit": "stereo pairs/s", "steps": args.train_steps,
                                      "ms_per_step": tc / args.train_steps * 1e3, "loss_after": float(crit2["loss"]),
                                      "what": "newtrain_cqe_real.py:128-174: HSIC eval forward (no_grad) + Independent_EN forward + distortion loss + "
                                              "backward" + ((" + gloo gradient all-reduce (REHEARSAL)" if rehearsal else " + RCCL gradient all-reduce") if world > 1 else "") + " + Adam"}
            if red2 is not None:
                red2.remove()
            if world == 1:
                # the reference driver's own loop on the product modules (plain criterion, plain Adam, HSIC in eval mode WITH grad enabled)
                RDL, _, cqe_it = driver_loop_pieces()
                net.eval()
                en.train()
                crit_d = RDL(0.01, rate=False)
                opt_d = torch.optim.Adam(en.parameters(), lr=1e-4)
                aopt_d = torch.optim.Adam(net.aux_parameters(), lr=1e-3)
                for i in range(2):
                    cqe_it(1, net, en, crit_d, opt_d, aopt_d, x1, x2, hm)
                barrier()
                t0 = time.perf_counter()
                for i in range(args.train_steps):
                    r = cqe_it(i, net, en, crit_d, opt_d, aopt_d, x1, x2, hm)
                barrier()
                td = time.perf_counter() - t0
                cqe_info["train_step_driver_loop"] = {"value": B * args.train_steps / td, "unit": "stereo pairs/s", "steps": args.train_steps,
                                                      "ms_per_step": td / args.train_steps * 1e3, "ratio_to_train_step": td / tc,
                                                      "what": "newtrain_cqe_real.py:152-180 restated on the caller side only: nn.MSELoss criterion, optim.Adam(model2.parameters()) "
                                                              "(not fused), HSIC called in eval mode with grad enabled, model.aux_loss() evaluated, .item() reads of the log line "
                                                              "at i % 10 == 0 (tests/test_gpu_driver_loop.py checks this loop against the reference's gradients)"}
        del en

    train_info = None
    if args.train_steps > 0:
        from masic_amd.parallel import GradientAllReducer
        from masic_amd.train import make_optimizers, train_step
        net.train()
        optimizer, aux_optimizer = make_optimizers(net)
        reducer = GradientAllReducer(net) if world > 1 else None
        for _ in range(3):                                                         # warm-up (weight packs of both kinds, optimizer state, the allocator pools of the main and the two side streams)
            train_step(net, optimizer, aux_optimizer, x1, x2, hm, 0.01, reducer)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.train_steps):
            crit, _ = train_step(net, optimizer, aux_optimizer, x1, x2, hm, 0.01, reducer)
        barrier()
        tt = max_over_ranks(time.perf_counter() - t0)
        gf = B * 3 * HSIC_GFLOP_PER_PAIR_512 * (H * W) / (512 * 512)
        train_info = {"value": world * B * args.train_steps / tt, "unit": "stereo pairs/s", "steps": args.train_steps,
                      "ms_per_step": tt / args.train_steps * 1e3, "loss_after": float(crit["loss"]),
                      "achieved_tflops": gf / 1e3 / (tt / args.train_steps),
                      "frac_of_mfma_peak": gf / 1e3 / (tt / args.train_steps) / (F32_MFMA_PEAK_TFLOPS if args.precision == "f32" else BF16_MFMA_PEAK_TFLOPS),
                      "what": f"forward ({args.precision} operands) + RD loss + backward ({args.precision} operands, f32 accumulate)" + (" + RCCL gradient all-reduce" if world > 1 else "") +
                              " + Adam + aux loss backward + aux Adam (newtrain_codec_real.py:135-146); algorithmic work = 3 x the forward's (SURVEY 8d)"}

    driver_info = None
    if args.train_steps > 0 and world == 1:
        # the reference driver's own loop (newtrain_codec_real.py:132-161) on the product HSIC: its plain-torch criterion, its unfused Adam pair,
        # model.aux_loss().backward(), its .item() reads -- no masic_amd.train / masic_amd.loss on the path
        RDL, codec_it, _ = driver_loop_pieces()
        driver_info = {}
        for mode in ([args.precision] + (["f32"] if args.precision != "f32" and not args.no_f32_compare else [])):
            if mode == "fp8":
                continue
            mnn.set_precision(mode)
            net.train()
            crit_d = RDL(0.01)
            opt_d = torch.optim.Adam(net.parameters(), lr=1e-4)
            aopt_d = torch.optim.Adam(net.aux_parameters(), lr=1e-3)
            for i in range(2):
                codec_it(1, net, crit_d, opt_d, aopt_d, x1, x2, hm)
            barrier()
            nd = args.train_steps if mode != "f32" else max(2, args.train_steps // 2)
            t0 = time.perf_counter()
            for i in range(nd):
                r = codec_it(i, net, crit_d, opt_d, aopt_d, x1, x2, hm)
            barrier()
            td = time.perf_counter() - t0
            driver_info[mode] = {"value": B * nd / td, "unit": "stereo pairs/s", "steps": nd, "ms_per_step": td / nd * 1e3,
                                 "loss_first_logged": r[0] if isinstance(r, tuple) else float(r)}
            if mode == args.precision and train_info is not None:
                driver_info[mode]["ratio_to_train_step"] = (td / nd * 1e3) / train_info["ms_per_step"]
        mnn.set_precision(args.precision)
        driver_info["what"] = ("newtrain_codec_real.py:132-161 restated on the caller side only: RateDistortionLoss from torch.log(lik).sum() / nn.MSELoss, "
                               "optim.Adam(net.parameters(), 1e-4) + optim.Adam(net.aux_parameters(), 1e-3) (torch's default foreach form), loss.backward(), "
                               "model.aux_loss().backward(), .item() reads of the log line at i % 10 == 0; tests/test_gpu_driver_loop.py checks this loop "
                               "against the reference's gradient goldens")

    # ---- "bpp/PSNR vs ref" at a TRAINED operating point: HSIC from the reference's default init, trained here with the HIP step (no trained
    # weights exist in this environment; masic_amd/trainrun.py), then the ORACLE against the f32 / bf16 / fp8 eval paths on a held-out pair
    trained_info = None
    if rank == 0 and world == 1 and not args.no_trained and not args.no_cpu_baseline:
        from masic_amd import fp8 as _fp8, trainrun
        from oracle import hsic_oracle as O            # checker leg: outside every timed region, never on the product path
        t0 = time.perf_counter()
        lam = 0.0932
        pool = trainrun.batch_pool(16, 8, 256, 256, dev, seed=5000)
        net_t = trainrun.default_init(device=dev)
        tl, _, _ = trainrun.train(net_t, args.trained_steps, pool, lam, precision="bf16")
        t_train = time.perf_counter() - t0
        net_t.eval()
        held = trainrun.consistent_pair(*(t.to(dev) for t in synth.synth_inputs(1, H, W, seed=9001)), seed=9001)
        hx1, hx2, hhm = (t.cpu() for t in held)
        sd_t = {k: v.detach().cpu().clone() for k, v in net_t.state_dict().items()}
        torch.set_num_threads(host_cores())
        with torch.no_grad():
            ref_t = O.hsic_forward(sd_t, hx1, hx2, hhm, K=K, keep=True)
        rc = O.rd_loss(ref_t, hx1, hx2, lam)
        rsym = O.symbols(ref_t["_aux"], sd_t)
        rb, rp1, rp2 = float(rc["bpp_loss"]), float(rc["psnr1"]), float(rc["psnr2"])
        _fp8.calibrate(net_t, [trainrun.consistent_pair(*(t.to(dev) for t in synth.synth_inputs(2, H, W, seed=9002)), seed=9002)])
        trained_info = {"state": f"HSIC(128,192,5) from the reference's default init (torch.manual_seed(0)), {args.trained_steps} steps of newtrain_codec_real.py:135-146 "
                                 f"(Adam 1e-4 / aux 1e-3, lambda {lam}) with the HIP bf16-operand step on 8x3x256x256 band-limited synthetic pairs (right view = homography "
                                 "warp of the left + noise), trained inside this run",
                        "train_seconds": t_train, "loss_first": tl[0][0], "loss_last": tl[-1][0],
                        "sample": f"held-out 1x3x{H}x{W} pair (synth seed 9001)",
                        "oracle": {"bpp": rb, "psnr1": rp1, "psnr2": rp2, "y1_symbol_range": [int(rsym["y1"].min()), int(rsym["y1"].max())]}}
        for mode in ("f32", "bf16", "fp8"):
            c = trainrun.compare_to_reference(trainrun.evaluate(net_t, *held, lam, mode), rsym, rb, rp1, rp2)
            trained_info[mode] = c
        mnn.set_precision(args.precision)
        del net_t, pool

    codec_info = None
    if rank == 0 and not args.no_codec:
        # the real bitstream of one pair: HSIC.compress / decompress on rank 0, outside every timed region above
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
                          "what": f"HSIC.compress / decompress of one {H}x{W} pair (files included): wavefront-ordered context coding, entropy parameters of each "
                                  "coding step by pixel-list kernels, GMM tables and the rANS symbol search on the GPU (one stream per latent channel, "
                                  "one wavefront per stream), the decode loop as back-to-back HIP-graph replays; the encoder's rANS on one host core"}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

    if rank == 0:
        line = {
            "metric": "stereo pairs/sec (enc+dec)", "value": world * B * args.steps / elapsed, "unit": "stereo pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"HSIC(N=128,M=192,K=5) eval forward (enc+dec both views), {B}x3x{H}x{W} stereo pairs per GPU "
                                   "(BASELINE.json configs[1] shape), inputs resident in HBM (the graph's static input buffers); "
                                   "the device homography of batch k+1 is handed over with batch k (one-batch look-ahead of a loader): its device->host "
                                   "read, the float32 host chain and the upload of the sampling matrices are done every step, under the running replay",
                       "pairs_per_gpu": B, "height": H, "width": W, "parallelism": f"dp{world} (pairs sharded, no data-path collective)"},
            "roofline": roofline,
        }
        extras = {}
        if train_info is not None:
            extras["train_step"] = train_info
        if driver_info is not None:
            extras["train_step_driver_loop"] = driver_info
        if no_lookahead is not None:
            extras["no_lookahead"] = no_lookahead
        if fp8_info is not None:
            extras["fp8_path"] = fp8_info
        if f32_info is not None:
            extras["f32_parity_path"] = f32_info
            extras["accuracy_vs_f32"] = accuracy
        if accuracy_ref is not None:
            extras["accuracy_vs_ref"] = accuracy_ref
        if trained_info is not None:
            extras["accuracy_vs_ref_trained"] = trained_info
        if upload is not None:
            extras["with_upload"] = upload
        if cqe_info is not None:
            extras["independent_en"] = cqe_info
        if codec_info is not None:
            extras["bitstream"] = codec_info
        if rehearsal_info is not None:
            extras["rehearsal_reducer_check"] = rehearsal_info
        if extras:
            line["extras"] = extras
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N, M, K, seed=100)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
