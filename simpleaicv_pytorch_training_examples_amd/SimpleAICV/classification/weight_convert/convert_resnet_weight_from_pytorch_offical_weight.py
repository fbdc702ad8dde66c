"""torchvision ResNet checkpoint -> this package's ResNet key names.

Behaviour of reference SimpleAICV/classification/weight_convert/convert_resnet_weight_from_pytorch_offical_weight.py:15-137
(a script with hard-coded paths there; a function plus a CLI here): every convolution / BatchNorm pair of torchvision's
`conv1|bn1`, `layerL.B.convK|bnK`, `layerL.B.downsample.0|1` becomes the two children `layer.0` / `layer.1` of one
ConvBnActBlock (`conv1`, `layerL.B.convK`, `layerL.B.downsample_conv`), `fc.*` keeps its name.  The converted checkpoint
is what `backbones.resnet50(pretrained_path=...)` / `load_state_dict` of the reference AND of this package read: the
reference quotes 80.858 % top-1 for the converted torchvision ResNet-50 (its README), the one known-answer it has.

    python -m simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.weight_convert.convert_resnet_weight_from_pytorch_offical_weight \
        --network resnet50 --src resnet50-11ad3fa6.pth --dst resnet50_converted.pth
"""
import argparse
import re

import torch

_BLOCK = re.compile(r'^(layer\d+\.\d+)\.(.+)$')


def _rename_unit(rest):
    """`conv2.weight` / `bn2.running_mean` / `downsample.1.bias` of one block (or of the stem) -> ConvBnActBlock child."""
    m = re.match(r'^conv(\d)\.(.+)$', rest)
    if m:
        return f'conv{m.group(1)}.layer.0.{m.group(2)}'
    m = re.match(r'^bn(\d)\.(.+)$', rest)
    if m:
        return f'conv{m.group(1)}.layer.1.{m.group(2)}'
    m = re.match(r'^downsample\.([01])\.(.+)$', rest)
    if m:
        return f'downsample_conv.layer.{m.group(1)}.{m.group(2)}'
    return None


def convert_torchvision_resnet_state_dict(state_dict):
    """-> (converted dict, list of source keys that have no counterpart).  Tensors are passed through untouched."""
    out, unknown = {}, []
    for key, value in state_dict.items():
        if key.startswith('fc.'):
            out[key] = value
            continue
        m = _BLOCK.match(key)
        new = _rename_unit(m.group(2) if m else key)
        if new is None:
            unknown.append(key)
            continue
        out[(m.group(1) + '.' if m else '') + new] = value
    return out, unknown


def check_against_model(converted, model):
    """Keys of `converted` that the model lacks, model keys it does not cover, and shape mismatches."""
    target = model.state_dict()
    extra = [k for k in converted if k not in target]
    missing = [k for k in target if k not in converted]
    shapes = [k for k in converted if k in target and tuple(converted[k].shape) != tuple(target[k].shape)]
    return extra, missing, shapes


def main():
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--network', default='resnet50')
    ap.add_argument('--num-classes', type=int, default=1000)
    ap.add_argument('--src', required=True, help='torchvision checkpoint (.pth state_dict)')
    ap.add_argument('--dst', required=True)
    args = ap.parse_args()
    from .. import backbones
    model = backbones.__dict__[args.network](**{'num_classes': args.num_classes})
    source = torch.load(args.src, map_location='cpu', weights_only=True)
    converted, unknown = convert_torchvision_resnet_state_dict(source)
    extra, missing, shapes = check_against_model(converted, model)
    print(f'{len(source)} source keys -> {len(converted)} converted; unknown {unknown}; not in model {extra}; '
          f'model keys without a source {missing}; shape mismatches {shapes}')
    if unknown or extra or missing or shapes:
        raise SystemExit('conversion incomplete')
    torch.save(converted, args.dst)


if __name__ == '__main__':
    main()
