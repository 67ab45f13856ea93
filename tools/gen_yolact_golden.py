#!/usr/bin/env python3
"""Writes tests/golden/yolact_<case>.npz (cases: tests/mask_cases.py) by running the REFERENCE's own Python network code
(/root/reference/src/python: backbone.py, yolact.py, layers/) on CPU with seeded random weights.

Only runnable in the build container (the reference never travels to the GPU box); the fixtures
it writes are data: subsampled output tensors, detections and the final person mask for a seeded
input, plus the state-dict key list.  cv2 / torchvision / pycocotools are absent here, so empty stub
modules stand in for them (the network path does not call into them) and torch.cuda.current_device
is patched (yolact.py:22 calls it at import).  The weights come from THIS project's YolactR50 under
torch.manual_seed(0) and are loaded into the reference model through its state dict, which also
proves key-for-key compatibility.
"""
import importlib.util
import os
import sys
import types
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/python"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402
import mask_cases as cases  # noqa: E402

entry.load_package()
mask = importlib.import_module("amos_slam_amd.mask")


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, n):
        return _Anything()

    def __call__(self, *a, **k):
        return _Anything()


stub("cv2")
tv = stub("torchvision")
tv.models = stub("torchvision.models")
tv.models.resnet = stub("torchvision.models.resnet", Bottleneck=_Anything)
tv.transforms = stub("torchvision.transforms")
stub("pycocotools")
stub("pycocotools.coco", COCO=_Anything)
stub("pycocotools.cocoeval", COCOeval=_Anything)
stub("pycocotools.mask")
torch.cuda.current_device = lambda: 0
torch.cuda.device_count = lambda: 2  # reference: "> 1 GPU => no JIT" keeps FPN a plain nn.Module
sys.path.insert(0, REF)
import yolact as ref_yolact  # noqa: E402  (the reference)
from layers.output_utils import postprocess as ref_postprocess  # noqa: E402

SUB = 997  # subsample stride for the big tensors


def sub(t):
    return t.detach().reshape(-1)[::SUB].numpy().copy()


def biased_engine(case):
    eng = mask.MaskEngine(device="cpu", seed=cases.weight_seed(case))
    cases.bias_class_head(eng.net, case)
    return eng


def run_case(case):
    eng = biased_engine(case)
    ref = ref_yolact.Yolact()
    mine_sd = eng.net.state_dict()
    ref_sd = ref.state_dict()
    assert sorted(mine_sd.keys()) == sorted(ref_sd.keys()), set(mine_sd) ^ set(ref_sd)
    assert all(mine_sd[k].shape == ref_sd[k].shape for k in mine_sd)
    ref.load_state_dict(mine_sd, strict=True)
    ref.eval()
    ref.detect.use_fast_nms = True
    ref.detect.use_cross_class_nms = False

    frame = cases.frame(case)
    chw = mask.cxx_marshalling(torch.from_numpy(frame))
    img = mask.resize_f32_cv(chw.permute(1, 2, 0) * 255, 640, 480)
    batch = mask.fast_base_transform(img)
    with torch.no_grad():
        # raw outputs: run the reference net in training-free "pred_outs" form by calling its parts
        ref_yolact.cfg._tmp_img_h, ref_yolact.cfg._tmp_img_w = 550, 550
        outs = ref.backbone(batch)
        fpn_outs = ref.fpn([outs[i] for i in ref_yolact.cfg.backbone.selected_layers])
        dets = ref(batch)  # softmax + Detect (fast_nms)
        det = dets[0]["detection"]
        det_copy = {k: v.clone() for k, v in det.items()}
        # layers/output_utils.postprocess (the twin of the interface's local copy) reads one attribute the
        # reference's config never defines: give it the value that skips the debug branch
        ref_yolact.cfg.mask_proto_debug = False
        classes, scores, boxes, masks = ref_postprocess(dets, 640, 480, score_threshold=0.15)
        # tail of prep_display (yolact_interface.py:822-832), restated: it lives in a file that needs cv2 + CUDA
        idx = scores.argsort(0, descending=True)[:15]
        m15, c15 = masks[idx], classes[idx].numpy()
        person = torch.zeros_like(m15[0])
        for k in np.argwhere(c15 == 0):
            person = person + m15[k[0]]
        person_u8 = (person * 255).byte().numpy()
    assert det_copy["score"].numel() > 20
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", f"yolact_{case}.npz"),
        keys=np.array(sorted(ref_sd.keys())), sub=np.array([SUB]), frame_crc=np.array([zlib.crc32(frame.tobytes())]),
        batch=sub(batch), c3=sub(outs[1]), c5=sub(outs[3]), p3=sub(fpn_outs[0]), p7=sub(fpn_outs[4]),
        det_box=det_copy["box"].numpy(), det_class=det_copy["class"].numpy(), det_score=det_copy["score"].numpy(),
        det_mask=det_copy["mask"].numpy(), proto=sub(det_copy["proto"]),
        post_classes=classes.numpy(), post_scores=scores.numpy(), post_mask_area=masks.sum((1, 2)).numpy(),
        person_mask_bits=np.packbits(person_u8 > 0), person_mask_values=np.unique(person_u8))
    print(case, "detections", det_copy["score"].numel(), "after 0.15:", len(scores), "classes", sorted(set(classes.tolist())),
          "person px:", int((person_u8 > 0).sum()), "values", np.unique(person_u8))


def main():
    for case in (sys.argv[1:] or list(cases.CASES)):
        run_case(case)


if __name__ == "__main__":
    main()
