"""Validation metrics (reference sg2im/metrics.py).  These run once per checkpoint on (O, 4)
box tensors that ``check_model`` copies to the host anyway, so they are host arithmetic."""
import torch


def intersection(bbox_pred, bbox_gt):
  """areas of the pairwise (row i with row i) box intersections, reference metrics.py:20-24"""
  hi = torch.min(bbox_pred[:, 2:], bbox_gt[:, 2:])
  lo = torch.max(bbox_pred[:, :2], bbox_gt[:, :2])
  side = (hi - lo).clamp(min=0)
  return side[:, 0] * side[:, 1]


def jaccard(bbox_pred, bbox_gt):
  """SUM over rows of IoU(pred_i, gt_i) (the caller divides by the box count), metrics.py:27-36"""
  bbox_pred, bbox_gt = bbox_pred.detach().float().cpu(), bbox_gt.detach().float().cpu()
  inter = intersection(bbox_pred, bbox_gt)
  area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
  return torch.sum(inter / (area(bbox_pred) + area(bbox_gt) - inter))
