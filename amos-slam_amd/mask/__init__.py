"""Dynamic-object (person) mask pass of the Amos front-end: YOLACT ResNet-50-FPN inference on
PyTorch-ROCm, written from scratch for this project.  It keeps the reference's state-dict key
layout (src/python/yolact.py, backbone.py) so the same `.pth` loads, and the reference's exact
pre/post-processing chain (yolact.cc:203-451, yolact_interface.py:663-884)."""
from .net import YolactR50, build_priors  # noqa: F401
from .detect import decode_boxes, fast_nms, detect, detect_batch  # noqa: F401
from .post import person_mask, person_mask_batch, postprocess_masks  # noqa: F401
from .pre import resize_u8_cv, resize_f32_cv, fast_base_transform, cxx_marshalling  # noqa: F401
from .interface import MaskEngine  # noqa: F401
