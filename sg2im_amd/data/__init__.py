"""The reference's data front end (sg2im/data/) for the HIP trainer: COCO(-Stuff) and Visual Genome scene-graph
datasets, their collate functions and the image pre / de-processing helpers.  Host-side code (PIL + numpy + torch
CPU tensors); nothing here touches the GPU.

Status: the reference's loaders cannot be imported in the build container as they are (torchvision, pycocotools,
skimage and h5py are absent).  `tests/test_loaders_vs_reference.py` imports them over small stand-ins for those four
packages and requires EQUAL outputs (vocabulary, image ids, every item tensor incl. the randomly drawn relationships,
the collate functions) from the reference's classes and these on tiny on-disk datasets: everything that is reference
code is pinned.  `tests/test_data_loaders.py` checks the documented contract (reference sg2im/data/coco.py:184-200,
vg.py:65-75) and the three third-party pieces re-implemented in `masks.py` / `utils.py` - COCO segmentation decoding
(pycocotools.mask), the mask resize (skimage.transform.resize) and the
image transform (torchvision.transforms) - deviations are listed there."""
from .coco import CocoSceneGraphDataset, coco_collate_fn
from .utils import Resize, imagenet_deprocess_batch, imagenet_preprocess, split_graph_batch
from .vg import VgSceneGraphDataset, vg_collate_fn, vg_uncollate_fn

__all__ = ['CocoSceneGraphDataset', 'coco_collate_fn', 'VgSceneGraphDataset', 'vg_collate_fn', 'vg_uncollate_fn',
           'Resize', 'imagenet_preprocess', 'imagenet_deprocess_batch', 'split_graph_batch']
