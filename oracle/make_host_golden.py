"""Fixtures for the HOST logic of the loop library, produced by the reference itself:
tools/utils.py Scheduler (lr at sampled fractional epochs) and build_optimizer (parameter
grouping: names / weight_decay / lr / lr_scale) for the ResNet-50 SGD, ViT-B AdamW layer-decay
and DETR-style sub_layer_lr configurations.  Run in the build container only.
    python oracle/make_host_golden.py  ->  tests/golden/host_logic.json
The four modules the reference imports at top level but that are not installed here (calflops,
cv2, torchvision, pycocotools) are stubbed; none of their code is on this path."""
import json
import os
import sys
import types

import torch

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'host_logic.json')


class Cfg:
    pass


def cases():
    r50 = ('resnet50', {'num_classes': 1000},
           ('SGD', {'lr': 0.1, 'momentum': 0.9, 'global_weight_decay': False, 'weight_decay': 1e-4,
                    'no_weight_decay_layer_name_list': []}),
           ('MultiStepLR', {'warm_up_epochs': 0, 'gamma': 0.1, 'milestones': [30, 60, 90]}), 100)
    vit = ('vit_base_patch16', {'image_size': 224, 'drop_path_prob': 0.1, 'global_pool': True, 'num_classes': 1000},
           ('AdamW', {'lr': 5e-4, 'global_weight_decay': False, 'weight_decay': 0.05,
                      'no_weight_decay_layer_name_list': ['position_encoding', 'cls_token'],
                      'lr_layer_decay': 0.65, 'lr_layer_decay_block': [f'blocks.{i}.' for i in range(12)],
                      'block_name': 'blocks'}),
           ('CosineLR', {'warm_up_epochs': 5, 'min_lr': 1e-6}), 100)
    sub = ('resnet18cifar', {'num_classes': 100},
           ('AdamW', {'lr': 1e-4, 'global_weight_decay': False, 'weight_decay': 1e-3,
                      'no_weight_decay_layer_name_list': [], 'sub_layer_lr': {'layer1': 1e-5, 'conv1': 2e-5},
                      'sub_layer_weight_decay': {'fc': 5e-3}}),
           ('PolyLR', {'warm_up_epochs': 1, 'power': 0.9, 'min_lr': 1e-7}), 20)
    return {'resnet50_sgd': r50, 'vit_layer_decay': vit, 'sub_layer': sub}


def main():
    sys.path.insert(0, REF)
    for name in ['calflops', 'cv2', 'torchvision', 'torchvision.transforms', 'pycocotools', 'pycocotools.mask',
                 'pycocotools.cocoeval']:
        sys.modules[name] = types.ModuleType(name)
    sys.modules['calflops'].calculate_flops = lambda *a, **k: None
    sys.modules['pycocotools.cocoeval'].COCOeval = object
    from tools import utils as U
    from SimpleAICV.classification import backbones
    out = {}
    for key, (net, kw, opt, sch, epochs) in cases().items():
        torch.manual_seed(0)
        model = backbones.__dict__[net](**kw)
        cfg = Cfg()
        cfg.optimizer, cfg.scheduler, cfg.epochs = opt, sch, epochs
        optimizer, summary = U.build_optimizer(cfg, model)
        groups = sorted([{'names': sorted(g['name']), 'weight_decay': g['weight_decay'], 'lr': g['lr'],
                          'lr_scale': g.get('lr_scale')} for g in summary], key=lambda g: g['names'][0])
        eff = {}
        name_of = {id(p): n for n, p in model.named_parameters()}
        for g in optimizer.param_groups:
            for p in g['params']:
                eff[name_of[id(p)]] = [g['lr'], g['weight_decay']]
        s = U.Scheduler(cfg, optimizer)
        points = [0.0, 0.013, 0.5, 1.0, 2.75, 5.0, 17.3, 30.0, 59.99, 60.0, 95.5, epochs - 0.001]
        points = [e for e in points if e < epochs]
        lrs = []
        for e in points:
            s.step(optimizer, e)
            lrs.append({'epoch': e, 'current_lr': s.current_lr,
                        'group_lrs': {name_of[id(g['params'][0])]: g['lr'] for g in optimizer.param_groups}})
        out[key] = {'network': net, 'kwargs': kw, 'optimizer': list(opt), 'scheduler': list(sch), 'epochs': epochs,
                    'groups': groups, 'effective': eff, 'schedule': lrs}
        print(key, len(groups), 'groups', len(eff), 'params')
    json.dump(out, open(OUT, 'w'), indent=0)
    print('wrote', OUT, os.path.getsize(OUT) // 1024, 'KiB')


if __name__ == '__main__':
    main()
