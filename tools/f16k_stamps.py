"""Phase times and the shader clock inside conv_f16k on the codec's layer shapes, from in-kernel stamps
(masic_conv_f16k_set_stamps: s_memtime = core clock, s_memrealtime = 100 MHz) of the first and the last workgroup of a launch.
   python tools/f16k_stamps.py      (GPU box)"""
import sys, os, ctypes, torch
sys.path.insert(0, os.getcwd())
from masic_amd import ops, _lib
torch.manual_seed(0)
dev = "cuda"
stamps = torch.zeros(16, dtype=torch.int64, device=dev)


def report(name, fn, flops, mfma_per_simd):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    _lib.lib.masic_conv_f16k_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    _lib.lib.masic_conv_f16k_set_stamps(None)
    us = e0.elapsed_time(e1) * 1e3
    s = stamps.cpu().tolist()
    print(f"{name}: launch {us:.1f} us, {flops / us / 1e6:.0f} TFLOP/s")
    for tag, o in (("first workgroup", 0), ("last workgroup ", 8)):
        core = [s[o + 2 * k] for k in range(4)]
        real = [s[o + 2 * k + 1] for k in range(4)]
        ph = [(real[k + 1] - real[k]) / 100.0 for k in range(3)]                       # us (100 MHz)
        clk = [(core[k + 1] - core[k]) / max(real[k + 1] - real[k], 1) * 100.0 for k in range(3)]   # MHz
        kl = ph[1]
        print(f"   {tag}: prologue {ph[0]:6.2f} us | K loop {kl:6.2f} us at {clk[1]:5.0f} MHz = {kl * clk[1] / max(mfma_per_simd, 1):5.1f} clk per MFMA per SIMD"
              f" | epilogue {ph[2]:6.2f} us   (starts {(real[0] - s[1]) / 100.0:6.2f} us after the first workgroup's)")


def layer(name, B, Cin, H, W, Cout, k, s, transposed=False, gdn=False, inverse=False):
    pad = k // 2
    x16 = ops.nchw_to_f16k(torch.randn(B, Cin, H, W, device=dev))
    wshape = (Cin, Cout, k, k) if transposed else (Cout, Cin, k, k)
    w = torch.randn(wshape, device=dev) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, device=dev)
    d = ops.make_conv_desc(B, Cin, H, W, Cout, k, k, s, pad, transposed=transposed, in_ctot=Cin, out_ctot=Cout, prec=_lib.PREC_BF16)
    wp = ops.pack_conv_f16k_weight(w, d)
    g = None
    if gdn:
        beta = torch.rand(128, device=dev) + 1.0
        gamma = torch.rand(128, 128, device=dev) * 0.02 + 0.1 * torch.eye(128, device=dev)
        g = (ops.pack_gdn_f16k(beta, gamma), inverse)
    flops = 2.0 * B * Cin * Cout * k * k * (H * W if transposed else d.Ho * d.Wo)
    buf = ctypes.create_string_buffer(96)
    _lib.lib.masic_conv_f16k_kernel_name(ctypes.byref(d), 1 if gdn else 0, buf, 96)
    wgs_px = 256
    # MFMAs per SIMD of one workgroup: 2 waves per SIMD, each (Cout_blk/32) tiles x K/16 k-steps for its 32 pixels (stride-2 deconv: a quarter of the taps per phase)
    kk = Cin * k * k / (4 if transposed and s == 2 else 1)
    mf = 2 * min(Cout, 128) / 32 * kk / 16
    report(f"{name}  [{buf.value.decode()}]", lambda: ops.conv2d_f16k(x16, wp, bias, d, gdn=g), flops, mf)


layer("analysis 128->128 5x5 s2 + GDN, 8 x 256^2 -> 128^2", 8, 128, 256, 256, 128, 5, 2, gdn=True)
layer("analysis 128->128 5x5 s2 + GDN, 8 x 128^2 -> 64^2 ", 8, 128, 128, 128, 128, 5, 2, gdn=True)
layer("synthesis 128->128 5x5 s2 T + IGDN, 8 x 128^2 -> 256^2", 8, 128, 128, 128, 128, 5, 2, transposed=True, gdn=True, inverse=True)
layer("synthesis 128->128 5x5 s2 T + IGDN, 8 x 64^2 -> 128^2 ", 8, 128, 64, 64, 128, 5, 2, transposed=True, gdn=True, inverse=True)
layer("CQE 96->96 3x3, 8 x 512^2", 8, 96, 512, 512, 96, 3, 1)
