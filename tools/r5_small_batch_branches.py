"""GPU: the mask pass as one HIP graph at small batches with and without the side-stream branch (AMOS_MASK_BRANCHES, AMOS_MASK_BRANCH_MAX_BATCH).  BATCHES=2,4 python tools/r5_small_batch_branches.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as e
e.load_package()
import importlib, torch, numpy as np
mask_mod = importlib.import_module("amos_slam_amd.mask")
for batch in tuple(int(v) for v in os.environ.get("BATCHES", "2,4").split(",")):
    eng = mask_mod.MaskEngine(device="cuda:0", seed=0)
    with torch.no_grad():
        b = eng.net.prediction_layers[0].conf_layer.bias
        bb = b.detach().cpu().view(3, 81).clone(); bb[:, 1] += 5.0
        b.copy_(bb.view(-1).to(b.device))
    eng.prepare()
    frames = torch.as_tensor(np.random.default_rng(0).integers(0, 256, (batch, 480, 640, 3), dtype=np.uint8), device="cuda:0")
    eng.capture_graph(batch=batch)
    for _ in range(3): eng.eval_bgr_graph(frames)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50): eng.eval_bgr_graph(frames)
    torch.cuda.synchronize()
    print("batch", batch, "branches", os.environ.get("AMOS_MASK_BRANCHES", "auto"), "%.3f ms" % ((time.perf_counter() - t) / 50 * 1e3), flush=True)
