"""Box decoding and Fast NMS (layers/box_utils.py:268-312, layers/functions/detection.py:27-170):
top_k 200 per class, IoU 0.5, class-confidence threshold 0.05, at most 100 detections."""
import os

import torch

NMS_TOP_K = 200
NMS_THRESH = 0.5
CONF_THRESH = 0.05
MAX_DETECTIONS = 100


def decode_boxes(loc, priors):
    """SSD decoding with variances (0.1, 0.2); returns [x1, y1, x2, y2] in relative coordinates."""
    centre = priors[:, :2] + loc[:, :2] * 0.1 * priors[:, 2:]
    size = priors[:, 2:] * torch.exp(loc[:, 2:] * 0.2)
    boxes = torch.cat((centre, size), 1)
    boxes[:, :2] -= boxes[:, 2:] / 2
    boxes[:, 2:] += boxes[:, :2]
    return boxes


def _pairwise_iou(boxes):
    """boxes [C, N, 4] -> IoU [C, N, N] (box_utils.jaccard's arithmetic: clamp the intersection at 0)."""
    lt = torch.max(boxes[:, :, None, :2], boxes[:, None, :, :2])
    rb = torch.min(boxes[:, :, None, 2:], boxes[:, None, :, 2:])
    wh = torch.clamp(rb - lt, min=0)
    inter = wh[..., 0] * wh[..., 1]
    area = (boxes[..., 2] - boxes[..., 0]) * (boxes[..., 3] - boxes[..., 1])
    return inter / (area[:, :, None] + area[:, None, :] - inter)


def _suppression_term(cand_boxes):
    """cand_boxes [B, C, k, 4], score-sorted along k -> [B, C, k]: for box j the largest IoU with a higher-scored box of its list
    (`jaccard.triu_(diagonal=1).max(dim=1)` of fast_nms).  On the GPU one HIP kernel of this project (amos_mask_nms_column_max_device:
    same float32 arithmetic, no k x k matrices -- PyTorch's nine elementwise passes over 410 MB at 32 frames); elsewhere the torch ops."""
    B, n_cls, k = cand_boxes.shape[:3]
    if cand_boxes.is_cuda and cand_boxes.dtype == torch.float32 and k <= 256:
        from .. import mask_nms_column_max
        cb = cand_boxes.contiguous()
        out = torch.empty((B, n_cls, k), dtype=torch.float32, device=cb.device)
        mask_nms_column_max(torch.cuda.current_stream(cb.device).cuda_stream, cb.data_ptr(), out.data_ptr(), B * n_cls, k)
        return out
    iou = _pairwise_iou(cand_boxes.reshape(B * n_cls, k, 4)).triu_(diagonal=1).view(B, n_cls, k, k)
    return iou.max(dim=2)[0]


def fast_nms(boxes, coefs, scores):
    """scores [80, K] over the K surviving priors.  Per class: sort, keep 200, drop a box when a
    higher-scored box of its class overlaps it by more than 0.5; then the best 100 over all classes."""
    scores, idx = scores.sort(1, descending=True)
    idx = idx[:, :NMS_TOP_K].contiguous()
    scores = scores[:, :NMS_TOP_K]
    n_cls, n_det = idx.shape
    boxes = boxes[idx.view(-1)].view(n_cls, n_det, 4)
    coefs = coefs[idx.view(-1)].view(n_cls, n_det, -1)
    iou = _pairwise_iou(boxes).triu_(diagonal=1)
    keep = iou.max(dim=1)[0] <= NMS_THRESH
    classes = torch.arange(n_cls, device=boxes.device)[:, None].expand_as(keep)[keep]
    boxes, coefs, scores = boxes[keep], coefs[keep], scores[keep]
    scores, order = scores.sort(0, descending=True)
    order = order[:MAX_DETECTIONS]
    return boxes[order], coefs[order], classes[order], scores[:MAX_DETECTIONS]


def detect(pred, batch_idx=0):
    """Detect.__call__ + Detect.detect for one image.  Returns None when nothing passes 0.05."""
    boxes = decode_boxes(pred["loc"][batch_idx], pred["priors"])
    cls_scores = pred["conf"][batch_idx].t()[1:]  # [80, P], background dropped
    keep = cls_scores.max(dim=0)[0] > CONF_THRESH
    if int(keep.sum()) == 0:
        return None
    b, m, c, s = fast_nms(boxes[keep], pred["mask"][batch_idx][keep], cls_scores[:, keep])
    return {"box": b, "mask": m, "class": c, "score": s, "proto": pred["proto"][batch_idx]}


def detect_batch(pred):
    """Detect for a whole batch with STATIC shapes (no boolean indexing, so one set of kernel launches
    serves every frame).  Priors that fail the 0.05 prior-level threshold get score -1 instead of being
    removed; they sort behind every real candidate, cannot suppress anything (suppression is by
    higher-scored boxes only) and are dropped later by the 0.15 score threshold.  Returns box [B, 100, 4],
    mask [B, 100, 32], class [B, 100], score [B, 100] (score <= 0 marks padding)."""
    loc, conf, coef, priors = pred["loc"], pred["conf"], pred["mask"], pred["priors"]
    B, P = loc.shape[:2]
    centre = priors[None, :, :2] + loc[..., :2] * 0.1 * priors[None, :, 2:]
    size = priors[None, :, 2:] * torch.exp(loc[..., 2:] * 0.2)
    x1y1 = centre - size / 2
    boxes = torch.cat((x1y1, x1y1 + size), -1)                       # [B, P, 4]
    if conf.is_cuda and conf.dtype == torch.float32 and conf.is_contiguous():
        # one HIP pass (amos_mask_class_scores_device) instead of a strided max, a fill, a where and a transposing copy
        from .. import mask_class_scores
        cls = torch.empty((B, conf.shape[2] - 1, P), dtype=torch.float32, device=conf.device)
        mask_class_scores(torch.cuda.current_stream(conf.device).cuda_stream, conf.data_ptr(), cls.data_ptr(), B, P, conf.shape[2], CONF_THRESH)
    else:
        cls = conf.transpose(1, 2)[:, 1:, :]                             # [B, 80, P]
        keep = cls.max(dim=1, keepdim=True)[0] > CONF_THRESH             # [B, 1, P]
        cls = torch.where(keep, cls, torch.full_like(cls, -1.0))
    k = min(NMS_TOP_K, P)
    if cls.is_cuda and cls.dtype == torch.float32 and cls.is_contiguous() and k <= 256 and os.environ.get("AMOS_MASK_TOPK", "1") != "0":
        # one HIP kernel, one work-group per (frame, class) row (amos_mask_topk_rows_device): the values torch.topk returns; equal
        # values come lowest index first (torch leaves their order open)
        from .. import mask_topk_rows
        n_cls = cls.shape[1]
        scores = torch.empty((B, n_cls, k), dtype=torch.float32, device=cls.device)
        idx = torch.empty((B, n_cls, k), dtype=torch.int64, device=cls.device)
        mask_topk_rows(torch.cuda.current_stream(cls.device).cuda_stream, cls.data_ptr(), scores.data_ptr(), idx.data_ptr(), B * n_cls, P, k)
    else:
        scores, idx = cls.topk(k, dim=2)                             # [B, 80, k], descending
    gather = idx.reshape(B, -1)
    cand_boxes = torch.gather(boxes, 1, gather[..., None].expand(-1, -1, 4)).view(B, -1, k, 4)
    cand_coefs = torch.gather(coef, 1, gather[..., None].expand(-1, -1, coef.shape[-1])).view(B, -1, k, coef.shape[-1])
    n_cls = cand_boxes.shape[1]
    alive = (_suppression_term(cand_boxes) <= NMS_THRESH) & (scores > 0)
    flat_scores = torch.where(alive, scores, torch.full_like(scores, -1.0)).view(B, -1)
    top_scores, order = flat_scores.topk(MAX_DETECTIONS, dim=1)       # [B, 100]
    classes = order // k
    out_boxes = torch.gather(cand_boxes.view(B, -1, 4), 1, order[..., None].expand(-1, -1, 4))
    out_coefs = torch.gather(cand_coefs.view(B, -1, coef.shape[-1]), 1, order[..., None].expand(-1, -1, coef.shape[-1]))
    return {"box": out_boxes, "mask": out_coefs, "class": classes, "score": top_scores, "proto": pred["proto"]}
