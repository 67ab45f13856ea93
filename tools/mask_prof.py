import sys, time, importlib, torch, numpy as np
sys.path.insert(0,'.')
import __graft_entry__ as e
e.load_package()
mask = importlib.import_module("amos_slam_amd.mask")
eng = mask.MaskEngine(device="cuda:0", seed=0)
with torch.no_grad():
    head = eng.net.prediction_layers[0].conf_layer.bias
    b = head.detach().cpu().view(3,81).clone(); b[:,1]+=5.0; b[1,3]+=5.5; head.copy_(b.view(-1).to(head.device))
frames = torch.randint(0,255,(32,480,640,3),dtype=torch.uint8,device="cuda:0")
def T(f, n=3):
    f(); torch.cuda.synchronize(); t=time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e3
with torch.no_grad():
    print("full batch32 ms", T(lambda: eng.eval_bgr_batch(frames)))
    chw = mask.cxx_marshalling(frames[0])
    print("marshal 1 frame ms", T(lambda: mask.cxx_marshalling(frames[0])))
    img = mask.resize_f32_cv(chw.permute(1,2,0)*255, 640, 480)
    print("resize_f32 ms", T(lambda: mask.resize_f32_cv(chw.permute(1,2,0)*255, 640, 480)))
    x = mask.fast_base_transform(img)
    print("fbt ms", T(lambda: mask.fast_base_transform(img)))
    batch = x.repeat(16,1,1,1)
    print("net fwd b16 fp32 ms", T(lambda: eng.net(batch)))
    pred = eng.net(batch)
    print("detect ms", T(lambda: mask.detect(pred, 0)))
    det = mask.detect(pred,0)
    print("person_mask ms", T(lambda: mask.person_mask(det, 640, 480)))
    net_cl = eng.net.to(memory_format=torch.channels_last)
    bcl = batch.to(memory_format=torch.channels_last)
    print("net fwd b16 fp32 channels_last ms", T(lambda: net_cl(bcl)))
    with torch.autocast("cuda", dtype=torch.float16):
        print("net fwd b16 fp16 autocast ms", T(lambda: net_cl(bcl)))
