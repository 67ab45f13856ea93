"""Mask assembly (yolact_interface.py:678-779, 806-832): score threshold 0.15, masks =
sigmoid(proto @ coef^T) cropped to the box (1 px padding), bilinear to the frame, > 0.5; the 15 best
detections; sum of the masks whose class is 0 (person); (m * 255) as 8 bits (overlaps wrap, as the
reference's `.byte()` does)."""
import torch
import torch.nn.functional as F

SCORE_THRESHOLD = 0.15
TOP_K_DISPLAY = 15
PERSON_CLASS = 0


def _sanitize(a, b, size, padding):
    a, b = a * size, b * size
    lo, hi = torch.min(a, b), torch.max(a, b)
    return torch.clamp(lo - padding, min=0), torch.clamp(hi + padding, max=size)


def _crop(masks, boxes, padding=1):
    """masks [h, w, n]: zero everything outside each detection's (padded) box, box_utils.crop."""
    h, w, n = masks.shape
    x1, x2 = _sanitize(boxes[:, 0], boxes[:, 2], w, padding)
    y1, y2 = _sanitize(boxes[:, 1], boxes[:, 3], h, padding)
    cols = torch.arange(w, device=masks.device, dtype=x1.dtype).view(1, -1, 1)
    rows = torch.arange(h, device=masks.device, dtype=x1.dtype).view(-1, 1, 1)
    inside = (cols >= x1.view(1, 1, -1)) & (cols < x2.view(1, 1, -1)) & (rows >= y1.view(1, 1, -1)) & (rows < y2.view(1, 1, -1))
    return masks * inside.to(masks.dtype)


def postprocess_masks(det, w, h, score_threshold=SCORE_THRESHOLD):
    """Returns (classes, scores, masks[n, h, w] in {0,1}) or None when nothing passes the threshold."""
    if det is None:
        return None
    keep = det["score"] > score_threshold
    if int(keep.sum()) == 0:
        return None
    classes, scores, boxes, coefs = det["class"][keep], det["score"][keep], det["box"][keep], det["mask"][keep]
    masks = torch.sigmoid(det["proto"] @ coefs.t())
    masks = _crop(masks, boxes).permute(2, 0, 1).contiguous()
    masks = F.interpolate(masks.unsqueeze(0), (h, w), mode="bilinear", align_corners=False).squeeze(0)
    return classes, scores, (masks > 0.5).to(torch.float32)


def person_mask(det, w, h):
    """uint8 [h, w] mask, 255 where a person was segmented; None if the network found nothing
    (the reference then raises inside prep_display and the caller's zeroed mask stays in use)."""
    out = postprocess_masks(det, w, h)
    if out is None:
        return None
    classes, scores, masks = out
    order = scores.argsort(0, descending=True)[:TOP_K_DISPLAY]
    classes, masks = classes[order], masks[order]
    total = masks[classes == PERSON_CLASS].sum(0)
    return ((total.to(torch.int64) * 255) & 0xFF).to(torch.uint8)
