"""COCO segmentation -> binary mask, and the mask resize, without pycocotools / skimage.

`seg_to_mask` accepts the three encodings of the COCO annotation format (reference coco.py:364-375 dispatches the
same way through pycocotools.mask): a list of polygons, an uncompressed RLE ({'counts': [..], 'size': [h, w]}) and a
compressed RLE ({'counts': '<string>', 'size': [h, w]}).

Deviations from the reference's third-party code (third-party behaviour, checked against independent re-derivations - see data/__init__.py):
 * polygons are rasterised with PIL.ImageDraw (pixel-centre rule + outline); pycocotools up-samples by 5 and walks
   the boundary: masks can differ in a one-pixel ring along the boundary.  The loader thresholds a 16 x 16 resize
   of the cropped mask, where a ring of boundary pixels rarely flips a cell.
 * `resize_mask` is first-order (bilinear) sampling at output pixel centres with zeros outside the image - what
   skimage.transform.resize(order=1, mode='constant') computes WITHOUT anti-aliasing (scikit-image 0.14.0 is what
   the reference's requirements.txt pins: no anti-aliasing by default; releases from 0.15 on blur before down-sampling)."""
import numpy as np
import PIL.Image
import PIL.ImageDraw


def polygons_to_mask(polygons, height, width):
  """polygons: list of flat [x0, y0, x1, y1, ...] lists (COCO image coordinates) -> uint8 (height, width)"""
  canvas = PIL.Image.new('L', (int(width), int(height)), 0)
  draw = PIL.ImageDraw.Draw(canvas)
  for poly in polygons:
    if len(poly) >= 6:
      draw.polygon([(float(poly[i]), float(poly[i + 1])) for i in range(0, len(poly) - 1, 2)], outline=1, fill=1)
  return np.asarray(canvas, dtype=np.uint8)


def rle_counts_to_mask(counts, height, width):
  """run lengths of 0s and 1s alternating (starting with 0s) over the COLUMN-major pixel order"""
  counts = np.asarray(counts, dtype=np.int64)
  if counts.sum() != height * width:
    raise ValueError('RLE counts sum to %d, expected %d x %d' % (counts.sum(), height, width))
  values = np.zeros(len(counts), dtype=np.uint8)
  values[1::2] = 1
  return np.repeat(values, counts).reshape(width, height).T.copy()


def rle_string_to_counts(s):
  """the LEB128-like string of a compressed RLE (6 bits per character, offset 48, 5 data bits + continuation bit,
  sign extension on the last chunk, every count from the third on stored as a difference to the one two before)"""
  if isinstance(s, bytes):
    s = s.decode('ascii')
  counts, p = [], 0
  while p < len(s):
    x, k, more = 0, 0, True
    while more:
      c = ord(s[p]) - 48
      x |= (c & 0x1f) << (5 * k)
      more = bool(c & 0x20)
      p += 1
      k += 1
      if not more and (c & 0x10):
        x |= -1 << (5 * k)
    if len(counts) > 2:
      x += counts[-2]
    counts.append(x)
  return counts


def seg_to_mask(seg, width, height):
  """-> uint8 array (height, width), 1 inside the object"""
  height, width = int(height), int(width)
  if isinstance(seg, list):
    return polygons_to_mask(seg, height, width)
  counts = seg['counts']
  h, w = seg.get('size', (height, width))
  if isinstance(counts, (str, bytes)):
    counts = rle_string_to_counts(counts)
  return rle_counts_to_mask(counts, int(h), int(w))


def resize_mask(mask, size):
  """(h, w) array -> float64 (size, size): bilinear samples at the output pixel centres, zeros outside"""
  src = np.asarray(mask, dtype=np.float64)
  h, w = src.shape

  def axis(n_in):
    pos = (np.arange(size) + 0.5) * (n_in / float(size)) - 0.5
    lo = np.floor(pos).astype(np.int64)
    return lo, pos - lo
  ylo, ty = axis(h)
  xlo, tx = axis(w)
  padded = np.zeros((h + 2, w + 2), dtype=np.float64)      # one ring of zeros: index i -> i + 1
  padded[1:-1, 1:-1] = src
  y0, y1 = np.clip(ylo + 1, 0, h + 1), np.clip(ylo + 2, 0, h + 1)
  x0, x1 = np.clip(xlo + 1, 0, w + 1), np.clip(xlo + 2, 0, w + 1)
  top = padded[y0][:, x0] * (1 - tx) + padded[y0][:, x1] * tx
  bot = padded[y1][:, x0] * (1 - tx) + padded[y1][:, x1] * tx
  return top * (1 - ty)[:, None] + bot * ty[:, None]
