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


def person_mask_batch(det, w, h):
    """person_mask for a batch from detect_batch's static-shape output.  Only the (at most 15) detections
    that the reference would display are assembled: the 15 best with score > 0.15, of which the persons
    are summed.  Returns uint8 [B, h, w] and a bool [B] "the network found something" (where it is False
    the reference raises and its caller keeps an all-zero mask)."""
    scores, classes, boxes, coefs, proto = det["score"], det["class"], det["box"], det["mask"], det["proto"]
    B = scores.shape[0]
    valid = scores > SCORE_THRESHOLD
    found = valid.any(dim=1)
    top_scores, order = torch.where(valid, scores, torch.full_like(scores, -1.0)).topk(TOP_K_DISPLAY, dim=1)
    sel_valid = top_scores > SCORE_THRESHOLD
    sel_cls = torch.gather(classes, 1, order)
    sel_box = torch.gather(boxes, 1, order[..., None].expand(-1, -1, 4))
    sel_coef = torch.gather(coefs, 1, order[..., None].expand(-1, -1, coefs.shape[-1]))
    masks = torch.sigmoid(torch.einsum("bhwk,bnk->bnhw", proto, sel_coef))          # [B, 15, 138, 138]
    ph, pw = masks.shape[2:]
    x1, x2 = _sanitize(sel_box[..., 0], sel_box[..., 2], pw, 1)
    y1, y2 = _sanitize(sel_box[..., 1], sel_box[..., 3], ph, 1)
    cols = torch.arange(pw, device=masks.device, dtype=x1.dtype).view(1, 1, 1, -1)
    rows = torch.arange(ph, device=masks.device, dtype=x1.dtype).view(1, 1, -1, 1)
    inside = (cols >= x1[..., None, None]) & (cols < x2[..., None, None]) & (rows >= y1[..., None, None]) & (rows < y2[..., None, None])
    masks = masks * inside.to(masks.dtype)
    person = sel_valid & (sel_cls == PERSON_CLASS)
    if masks.is_cuda and masks.dtype == torch.float32:
        # upsample + threshold + count of the displayed persons + (x 255).byte() in one HIP pass (amos_mask_person_mask_device) instead
        # of a [B, 15, h, w] float tensor and four passes over it
        from .. import mask_person_mask
        masks, flags = masks.contiguous(), person.to(torch.uint8).contiguous()
        out = torch.empty((B, h, w), dtype=torch.uint8, device=masks.device)
        mask_person_mask(torch.cuda.current_stream(masks.device).cuda_stream, masks.data_ptr(), flags.data_ptr(), out.data_ptr(), B, masks.shape[1], ph, pw, h, w)
        return out, found
    masks = F.interpolate(masks, (h, w), mode="bilinear", align_corners=False) > 0.5
    total = (masks & person[..., None, None]).sum(dim=1)
    return ((total.to(torch.int64) * 255) & 0xFF).to(torch.uint8), found


def person_masks_fused(pred, w, h):
    """detect_batch + person_mask_batch as ONE call of this project's library (amos_mask_person_masks_device: seven kernels instead of the
    ~75 small launches the torch ops above cost per pass -- a fifth of a one-frame mask pass): the same selection rules and the same float32
    arithmetic operation by operation, except the 32-term prototype sums (sequential fused multiply-adds here, a BLAS kernel's order there:
    float32 rounding).  Returns (uint8 [B, h, w], bool [B]) like person_mask_batch, or None where it does not apply (not float32 on a
    GPU): the caller then takes the torch path."""
    loc, coef, priors, proto = pred["loc"], pred["mask"], pred["priors"], pred["proto"]
    cls = pred.get("cls")          # Detect's class scores [B, classes, P] from the head's own kernel (YolactR50.forward(scores_only=True)) ...
    conf = pred.get("conf")        # ... or the softmax tensor [B, P, 1 + classes]
    scores = cls if cls is not None else conf
    if scores is None or not (loc.is_cuda and all(t.dtype == torch.float32 for t in (loc, scores, coef, priors, proto))) or proto.dim() != 4:
        return None
    import os
    if os.environ.get("AMOS_MASK_FUSED_POST", "1") == "0":   # A/B runs and the tests that compare the two paths
        return None
    from .. import mask_person_masks, mask_person_masks_scores, mask_post_workspace_bytes
    loc, scores, coef, priors, proto = (t.contiguous() for t in (loc, scores, coef, priors, proto))
    B, P = loc.shape[:2]
    ph, pw, D = proto.shape[1:]
    c1 = cls.shape[1] + 1 if cls is not None else conf.shape[2]
    if P < 200 or (c1 - 1) * 200 * 4 > 64 * 1024 or D % 4 != 0:
        return None
    nbytes = mask_post_workspace_bytes(B, P, c1, D, ph, pw)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=loc.device)
    out = torch.empty((B, h, w), dtype=torch.uint8, device=loc.device)
    found = torch.empty(B, dtype=torch.uint8, device=loc.device)
    (mask_person_masks_scores if cls is not None else mask_person_masks)(
        torch.cuda.current_stream(loc.device).cuda_stream, loc.data_ptr(), scores.data_ptr(), coef.data_ptr(), priors.data_ptr(), proto.data_ptr(),
        B, P, c1, D, ph, pw, h, w, ws.data_ptr(), nbytes, out.data_ptr(), found.data_ptr())
    return out, found.bool()
