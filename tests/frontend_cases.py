"""Seeded synthetic DETR decoder outputs and ground-truth boxes for the object front-end tests (shared by the golden generator,
the CPU oracle tests and the GPU parity tests)."""
import torch


def make_detr_outputs(seed: int, n_img: int = 6, n_query: int = 100, n_cls1: int = 151):
    """pred_logits [B,100,151], pred_boxes [B,100,4] (cx,cy,w,h in 0..1).  About a third of the queries have an object; image 3
    has none at all (the reference drops it); some queries put the background second (their second category is dropped)."""
    g = torch.Generator().manual_seed(1000 + seed)
    logits = torch.randn(n_img, n_query, n_cls1, generator=g) * 2.0
    logits[:, :, -1] += 6.0                                           # background dominates by default
    obj = torch.rand(n_img, n_query, generator=g) < 0.35
    cls = torch.randint(0, n_cls1 - 1, (n_img, n_query), generator=g)
    boost = torch.zeros_like(logits)
    boost.scatter_(2, cls[:, :, None], 9.0)
    logits = logits + boost * obj[:, :, None]
    if n_img > 3:
        logits[3, :, -1] += 30.0                                      # image 3: no objects
    # duplicate-ish detections of the same class with overlapping boxes so that NMS has work
    boxes = torch.rand(n_img, n_query, 4, generator=g)
    boxes[..., 2:] = boxes[..., 2:] * 0.5 + 0.08
    for b in range(n_img):
        for q in range(0, n_query - 1, 7):
            logits[b, q + 1] = logits[b, q] + 0.01 * torch.randn(n_cls1, generator=g)
            boxes[b, q + 1] = boxes[b, q] + 0.01 * torch.randn(4, generator=g)
    boxes = boxes.clamp(0.01, 0.99)
    return logits.contiguous(), boxes.contiguous()


def make_target_boxes(seed: int, bbox_pred):
    """Ground-truth boxes (x0,x1,y0,y1) on the 32-grid per kept image: a few copies of predicted boxes (integer-truncated, so
    that exact IoU ties with repeated boxes occur) and a few random ones."""
    g = torch.Generator().manual_seed(2000 + seed)
    out = []
    for b, bp in enumerate(bbox_pred):
        n = int(bp.shape[0])
        rows = []
        for k in range(min(4, n)):
            j = int(torch.randint(0, n, (1,), generator=g))
            rows.append(torch.floor(bp[j]).clone())
        for k in range(3):
            x0 = int(torch.randint(0, 24, (1,), generator=g)); y0 = int(torch.randint(0, 24, (1,), generator=g))
            w = int(torch.randint(2, 9, (1,), generator=g)); h = int(torch.randint(2, 9, (1,), generator=g))
            rows.append(torch.tensor([x0, x0 + w, y0, y0 + h], dtype=torch.float32))
        out.append(torch.stack(rows))
    return out
