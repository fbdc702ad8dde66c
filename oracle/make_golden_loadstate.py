"""Fixture for the checkpoint-loading host path, produced by the reference's own
SimpleAICV/classification/common.py load_state_dict: a ViT-like module whose saved position embedding was made for a 4 x 4
token grid is loaded into a 6 x 6 model (bicubic resize of the grid part, class-token row kept), together with the name /
shape / excluded-layer filtering.  Build container only:   python oracle/make_golden_loadstate.py"""
import os
import sys
import tempfile
import types

import torch
import torch.nn as nn

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'load_state_dict.pt')


class Toy(nn.Module):
    def __init__(self, grid, planes=8, classes=5):
        super().__init__()
        self.cls_token = nn.Parameter(torch.zeros(1, 1, planes))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + grid * grid, planes))
        self.proj = nn.Linear(planes, planes)
        self.head = nn.Linear(planes, classes)
        self.bn = nn.BatchNorm1d(planes)


def main():
    for name in ['calflops', 'cv2', 'torchvision', 'torchvision.transforms']:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, REF)
    from SimpleAICV.classification.common import load_state_dict
    g = torch.Generator().manual_seed(5)
    saved = Toy(4, classes=7)                          # head of another width: must be skipped (shape filter)
    with torch.no_grad():
        for p in saved.parameters():
            p.copy_(torch.randn(p.shape, generator=g))
        saved.bn.running_mean.copy_(torch.randn(8, generator=g))
        saved.bn.num_batches_tracked.fill_(9)
    sd = {k: v.clone() for k, v in saved.state_dict().items()}
    sd['not_in_model.weight'] = torch.randn(3, generator=g)
    results = {}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'saved.pth')
        torch.save(sd, path)
        for key, kwargs in (('resize', dict(loading_new_input_size_position_encoding_weight=True)),
                            ('no_resize', dict()),
                            ('excluded', dict(excluded_layer_name=('proj',), loading_new_input_size_position_encoding_weight=True))):
            torch.manual_seed(1)
            model = Toy(6)
            before = {k: v.clone() for k, v in model.state_dict().items()}
            load_state_dict(path, model, **kwargs)
            after = model.state_dict()
            results[key] = {'kwargs': kwargs, 'after': {k: v.clone() for k, v in after.items()},
                            'changed': sorted(k for k in after if not torch.equal(after[k], before[k]))}
            print(key, results[key]['changed'])
    torch.save({'saved': sd, 'results': results, 'model_seed': 1}, OUT)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
