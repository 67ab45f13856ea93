"""Module that the C++ `yolact` class imports (same module and function names as the reference's
src/python/yolact_interface.py: `yolact_init(model_path, categories)` and `yolact_eval(image)`,
include/yolact.h:32-34)."""
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
model_interface = None


def _mask_package():
    if "amos_slam_amd" not in sys.modules:
        spec = importlib.util.spec_from_file_location("amos_slam_amd", os.path.join(_PKG, "__init__.py"),
                                                      submodule_search_locations=[_PKG])
        mod = importlib.util.module_from_spec(spec)
        sys.modules["amos_slam_amd"] = mod
        spec.loader.exec_module(mod)
    return importlib.import_module("amos_slam_amd.mask")


def yolact_init(weight_path, num_class):
    """weight_path: a YOLACT-ResNet50 `.pth`, or "" for seeded random weights (performance runs only)."""
    global model_interface
    device = os.environ.get("AMOS_MASK_DEVICE") or None
    model_interface = _mask_package().MaskEngine(weight_path or None, device=device).prepare()  # weights loaded, then batch norms folded
    return


def yolact_eval(image):
    """image: CHW float32 numpy array in [0, 1].  Returns (mask, mask) like the reference; raises when the
    network finds nothing, which the C++ side reports as `false` exactly like the reference does."""
    mask = model_interface.eval_chw(np.ascontiguousarray(image, np.float32))
    if mask is None:
        raise IndexError("no detection above the score threshold")
    out = mask.cpu().numpy()
    return out, out


def _graph_path():
    """frame-by-frame callers on a GPU: pre-processing + network + detection + mask assembly replayed as ONE captured HIP graph (2.6 instead
    of 3.5 ms per frame on MI355X, tools/mask_latency.py; the static-shape batch path of detect.py / post.py, same masks).  The default;
    AMOS_MASK_GRAPH=0 selects the eager pass (A/B runs, tests)."""
    return os.environ.get("AMOS_MASK_GRAPH", "1") != "0" and model_interface.device.type == "cuda"


def yolact_frame_session(height, width):
    """For the C++ `yolact` class: the addresses of the pinned host buffers of the engine's per-frame session -- (frame buffer, mask buffer,
    mask rows, mask columns) -- or None when the engine does not run on a GPU (or AMOS_MASK_GRAPH=0): the caller then hands the frame over
    as bytes (yolact_eval_bgr_bytes).  The C++ side copies the frame straight into the frame buffer, calls yolact_eval_session() and clones
    the mask out of the mask buffer: no PyBytes, no numpy array, no pageable copy."""
    if not _graph_path():
        return None
    s = model_interface.frame_session(height, width)
    return s.in_ptr, s.out_ptr, 480, 640


def yolact_eval_session(height, width):
    """Runs the session of yolact_frame_session(height, width) on the frame in its frame buffer; True when the mask buffer holds a mask, False
    when no detection passed the score threshold (the reference raises IndexError there and evalImage returns false)."""
    return model_interface.frame_session(height, width).run()


def yolact_eval_bgr_bytes(buf, height, width):
    """Entry point of the C++ `yolact` class of this project: the raw BGR frame as bytes; the
    reference's C++ marshalling (resize to 480x640, /255, CHW) runs on the GPU in mask/pre.py."""
    frame = np.frombuffer(buf, np.uint8).reshape(height, width, 3)
    if _graph_path():
        s = model_interface.frame_session(height, width)
        s.frame_in.numpy()[...] = frame
        if not s.run():
            raise IndexError("no detection above the score threshold")
        out = s.mask_out.numpy().copy()
        return out, out
    mask = model_interface.eval_bgr(frame.copy())
    if mask is None:
        raise IndexError("no detection above the score threshold")
    out = np.ascontiguousarray(mask.cpu().numpy())
    return out, out
