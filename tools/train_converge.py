#!/usr/bin/env python3
"""Convergence of the codec-stage training step from the reference's default init, float32 vs bf16 operands, and the quality of
the trained state in every eval operand mode.

    python tools/train_converge.py [--steps 300 --batch 4 --size 256 --lmbda 0.0932 --modes f32,bf16 --out gpurun_out/converge.json]

Same batches (numpy MT19937) and the same noise draws (device generator re-seeded per step) in both modes.  Prints window means of
the loss, the reference's own trace for orientation (coremasic/mywork/train_log.txt:1-40: 35.6 -> 1.39 in 390 iterations at
lambda 0.001, batch 1, natural images), then evaluates each trained state on a held-out 512x512 pair in the f32 / bf16 / fp8 eval
modes.  The comparison against the CPU oracle lives in bench.py (extras.accuracy_vs_ref_trained) and tests/test_gpu_convergence.py."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "coremasic", "mywork")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--pool", type=int, default=16)
    ap.add_argument("--lmbda", type=float, default=0.0932)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--modes", default="f32,bf16")
    ap.add_argument("--window", type=int, default=20)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from masic_amd import fp8, synth, trainrun
    dev = torch.device("cuda", 0)
    pool = trainrun.batch_pool(args.pool, args.batch, args.size, args.size, dev)
    held = trainrun.consistent_pair(*(t.to(dev) for t in synth.synth_inputs(1, 512, 512, seed=9001)), seed=9001)
    calib = [tuple(t.to(dev) for t in synth.synth_inputs(2, 512, 512, seed=9002))]
    report = {"args": vars(args), "modes": {}}
    for mode in args.modes.split(","):
        net = trainrun.default_init(device=dev)
        t0 = time.perf_counter()
        losses, auxes, _ = trainrun.train(net, args.steps, pool, args.lmbda, precision=mode, lr=args.lr, log=max(1, args.steps // 10))
        dt = time.perf_counter() - t0
        w = args.window
        means = [sum(l[0] for l in losses[i:i + w]) / len(losses[i:i + w]) for i in range(0, len(losses), w)]
        print(f"[{mode}] {args.steps} steps in {dt:.1f} s ({dt / args.steps * 1e3:.1f} ms/step); window-{w} loss means: " + " ".join(f"{m:.2f}" for m in means))
        ev = {}
        fp8.calibrate(net.eval(), calib)
        for em in ("f32", "bf16", "fp8"):
            r = trainrun.evaluate(net, *held, args.lmbda, em)
            ev[em] = r
            print(f"   eval[{em}] held-out 512x512: bpp {r['bpp']:.4f} psnr {r['psnr1']:.2f} / {r['psnr2']:.2f} dB")
        base = ev["f32"]
        cmp_ = {em: trainrun.compare_to_reference(ev[em], base["sym"], base["bpp"], base["psnr1"], base["psnr2"]) for em in ("bf16", "fp8")}
        for em, c in cmp_.items():
            print(f"   {em} vs f32 eval: bpp {c['bpp_rel_delta']:+.3%} psnr {c['psnr1_delta_db']:+.3f} / {c['psnr2_delta_db']:+.3f} dB, "
                  f"symbols {c['symbol_mismatch_rate']:.2%} differ (max |d| {c['symbol_max_abs_diff']})")
        report["modes"][mode] = {"seconds": dt, "loss": [l[0] for l in losses], "bpp": [l[1] for l in losses], "mse": [l[2] for l in losses], "aux": auxes,
                                 "window_means": means, "eval": {em: {k: v for k, v in ev[em].items() if k in ("bpp", "psnr1", "psnr2", "loss")} for em in ev},
                                 "vs_f32_eval": cmp_}
        del net
    if len(report["modes"]) == 2:
        a, b = (report["modes"][m]["window_means"] for m in args.modes.split(","))
        print("window-mean ratio (second / first mode): " + " ".join(f"{y / x:.3f}" for x, y in zip(a, b)))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(report, f)


if __name__ == "__main__":
    main()
