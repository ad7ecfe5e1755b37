"""Image transforms and batch helpers of the data front end (reference sg2im/data/utils.py) without torchvision."""
import numpy as np
import PIL.Image
import torch

from ..utils import IMAGENET_MEAN, IMAGENET_STD, imagenet_deprocess_batch  # noqa: F401  (re-exported)


class Resize(object):
  """PIL resize to (H, W) (or a square), bilinear by default - reference data/utils.py:71-82"""

  def __init__(self, size, interp=PIL.Image.BILINEAR):
    h, w = size if isinstance(size, (tuple, list)) else (size, size)
    self.size, self.interp = (int(w), int(h)), interp          # PIL wants (width, height)

  def __call__(self, img):
    return img.resize(self.size, self.interp)


def pil_to_tensor(img):
  """what torchvision.transforms.ToTensor does to an RGB PIL image: uint8 HWC -> float32 CHW in [0, 1]"""
  a = np.asarray(img, dtype=np.uint8)
  if a.ndim == 2:
    a = a[:, :, None]
  return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div_(255.0)


class _Normalize(object):
  def __init__(self, mean, std):
    self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

  def __call__(self, x):
    return (x - self.mean) / self.std


def imagenet_preprocess():
  """(x - mean) / std per channel, reference data/utils.py:29-30"""
  return _Normalize(IMAGENET_MEAN, IMAGENET_STD)


class ImageTransform(object):
  """Resize -> tensor -> (optional) ImageNet normalisation: the Compose of reference coco.py:222-227 / vg.py:47-50"""

  def __init__(self, image_size, normalize=True):
    self.resize = Resize(image_size)
    self.normalize = imagenet_preprocess() if normalize else None

  def __call__(self, img):
    x = pil_to_tensor(self.resize(img))
    return self.normalize(x) if self.normalize is not None else x


def load_image(path, transform):
  """-> (tensor (3, H, W), original width, original height)"""
  with open(path, 'rb') as f:
    with PIL.Image.open(f) as image:
      ww, hh = image.size
      return transform(image.convert('RGB')), ww, hh


def split_graph_batch(triples, obj_data, obj_to_img, triple_to_img):
  """Undo a collate: per image the triples (object indices local again) and the rows of every per-object tensor in
  ``obj_data`` (entries may be None) - reference data/utils.py:91-117."""
  triples, obj_to_img, triple_to_img = triples.detach(), obj_to_img.detach(), triple_to_img.detach()
  obj_data = [None if o is None else o.detach() for o in obj_data]
  n_images = int(obj_to_img.max()) + 1
  triples_out, obj_out = [], [[] for _ in obj_data]
  first = 0
  for i in range(n_images):
    rows = (obj_to_img == i).nonzero().view(-1)
    cur = triples[(triple_to_img == i).nonzero().view(-1)].clone()
    cur[:, 0] -= first
    cur[:, 2] -= first
    triples_out.append(cur)
    for dst, o in zip(obj_out, obj_data):
      dst.append(None if o is None else o[rows])
    first += rows.numel()
  return triples_out, obj_out
