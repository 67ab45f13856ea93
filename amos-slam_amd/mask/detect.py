"""Box decoding and Fast NMS (layers/box_utils.py:268-312, layers/functions/detection.py:27-170):
top_k 200 per class, IoU 0.5, class-confidence threshold 0.05, at most 100 detections."""
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
