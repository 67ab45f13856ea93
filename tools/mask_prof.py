#!/usr/bin/env python3
"""Where the mask pass spends its time (GPU)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
e.load_package()
mask = importlib.import_module("amos_slam_amd.mask")
eng = mask.MaskEngine(device="cuda:0", seed=0)
with torch.no_grad():
    head = eng.net.prediction_layers[0].conf_layer.bias
    b = head.detach().cpu().view(3, 81).clone(); b[:, 1] += 5.0; b[1, 3] += 5.5; head.copy_(b.view(-1).to(head.device))
frames = torch.randint(0, 255, (32, 480, 640, 3), dtype=torch.uint8, device="cuda:0")
def T(f, n=3):
    f(); torch.cuda.synchronize(); t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
with torch.no_grad():
    print("eval_bgr_batch(32)      ms", round(T(lambda: eng.eval_bgr_batch(frames)), 2))
    chw = mask.cxx_marshalling(frames[:16])
    print("marshal 16              ms", round(T(lambda: mask.cxx_marshalling(frames[:16])), 2))
    imgs = mask.resize_f32_cv(chw.permute(0, 2, 3, 1) * 255, 640, 480)
    print("resize_f32 16           ms", round(T(lambda: mask.resize_f32_cv(chw.permute(0, 2, 3, 1) * 255, 640, 480)), 2))
    x = mask.fast_base_transform(imgs)
    print("fast_base_transform 16  ms", round(T(lambda: mask.fast_base_transform(imgs)), 2))
    print("net fwd 16 fp32         ms", round(T(lambda: eng.net(x)), 2))
    pred = eng.net(x)
    print("detect x16              ms", round(T(lambda: [mask.detect(pred, k) for k in range(16)]), 2))
    dets = [mask.detect(pred, k) for k in range(16)]
    print("person_mask x16         ms", round(T(lambda: [mask.person_mask(d, 640, 480) for d in dets]), 2))
    # memory-format / MIOpen search / batch-size experiments for the fp32 forward
    for bs in (16, 32):
        xb = torch.randn(bs, 3, 550, 550, device="cuda:0")
        print(f"net fwd {bs} fp32 nchw       ms", round(T(lambda: eng.net(xb)), 2), flush=True)
    torch.backends.cudnn.benchmark = True
    xb = torch.randn(16, 3, 550, 550, device="cuda:0")
    print("net fwd 16 fp32 benchmark  ms", round(T(lambda: eng.net(xb)), 2), flush=True)
    eng.net.to(memory_format=torch.channels_last)
    xcl = xb.contiguous(memory_format=torch.channels_last)
    print("net fwd 16 fp32 NHWC+bench ms", round(T(lambda: eng.net(xcl)), 2), flush=True)
    ref = eng.net.to(memory_format=torch.contiguous_format)(xb)
    out = eng.net.to(memory_format=torch.channels_last)(xcl)
    for k in ("loc", "conf", "mask", "proto"):
        print("  NHWC vs NCHW max abs diff", k, float((ref[k].float() - out[k].float()).abs().max()), flush=True)
    # reduced-precision convolutions (opt-in): speed and agreement of the final person masks with fp32
    eng.net.to(memory_format=torch.channels_last)
    e32 = eng
    masks32 = e32.eval_bgr_batch(frames)
    for name, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        e32.conv_dtype = dt
        print(f"eval_bgr_batch(32) {name}  ms", round(T(lambda: e32.eval_bgr_batch(frames)), 2), flush=True)
        m = e32.eval_bgr_batch(frames)
        a, b = masks32 > 0, m > 0
        inter, union = (a & b).flatten(1).sum(1).float(), (a | b).flatten(1).sum(1).float().clamp(min=1)
        print(f"  mask IoU vs fp32 ({name}): min {float((inter / union).min()):.4f} mean {float((inter / union).mean()):.4f}", flush=True)
    e32.conv_dtype = None
    print("eval_bgr_batch(32) fp32  ms", round(T(lambda: e32.eval_bgr_batch(frames)), 2), flush=True)
