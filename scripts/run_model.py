#!/usr/bin/env python
"""Generate images from scene-graph JSON with a trained checkpoint (reference scripts/run_model.py):
load ``model_kwargs`` / ``model_state``, ``Sg2ImModel.forward_json`` in eval mode on the MI355X,
de-normalise, write ``img%06d.png``.  Same flags as the reference.  There is no CPU path in this
package, so ``--device cpu`` (or a missing GPU) is an error instead of a silent fallback, and
``--draw_scene_graphs`` needs graphviz' ``dot`` (sg2im/vis.py is outside the hot-path scope)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
from PIL import Image

from sg2im_amd.model import Sg2ImModel
from sg2im_amd.utils import imagenet_deprocess_batch

parser = argparse.ArgumentParser()
parser.add_argument('--checkpoint', default='sg2im-models/vg128.pt')
parser.add_argument('--scene_graphs_json', default='scene_graphs/example_meadow.json')
parser.add_argument('--output_dir', default='outputs')
parser.add_argument('--draw_scene_graphs', type=int, default=0)
parser.add_argument('--device', default='gpu', choices=['cpu', 'gpu'])


def main(args):
  if not os.path.isfile(args.checkpoint):
    print('ERROR: Checkpoint file "%s" not found' % args.checkpoint)
    return 1
  if args.device != 'gpu' or not torch.cuda.is_available():
    raise RuntimeError('sg2im_amd runs on an MI355X only: no CPU path (use the reference for --device cpu)')
  if args.draw_scene_graphs == 1:
    raise NotImplementedError('drawing scene graphs shells out to graphviz (sg2im/vis.py), which is out of scope')
  if not os.path.isdir(args.output_dir):
    print('Output directory "%s" does not exist; creating it' % args.output_dir)
    os.makedirs(args.output_dir)
  device = torch.device('cuda:0')
  checkpoint = torch.load(args.checkpoint, map_location='cpu', weights_only=False)
  model = Sg2ImModel(**checkpoint['model_kwargs'])
  model.load_state_dict(checkpoint['model_state'])
  model.eval()
  model.to(device)
  with open(args.scene_graphs_json, 'r') as f:
    scene_graphs = json.load(f)
  with torch.no_grad():
    imgs, boxes_pred, masks_pred, _ = model.forward_json(scene_graphs)
  imgs = imagenet_deprocess_batch(imgs)
  for i in range(imgs.shape[0]):
    path = os.path.join(args.output_dir, 'img%06d.png' % i)
    Image.fromarray(imgs[i].numpy().transpose(1, 2, 0)).save(path)
  print('Wrote %d images to %s' % (imgs.shape[0], args.output_dir))
  return 0


if __name__ == '__main__':
  sys.exit(main(parser.parse_args()))
