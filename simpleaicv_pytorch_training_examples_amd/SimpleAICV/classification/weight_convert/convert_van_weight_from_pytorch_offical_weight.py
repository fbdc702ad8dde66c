"""Official VAN checkpoint (`{'state_dict': ...}`, Visual-Attention-Network release) -> the weights this package's VAN loads.

Behaviour of reference SimpleAICV/classification/weight_convert/convert_van_weight_from_pytorch_offical_weight.py:14-37, :140-152:
the backbone keeps the official key names, so conversion is a FILTER -- keep a key when the target model has it, EXCEPT the
top-level stage norms (`norm1..4.*`: LayerNorm in the release, BatchNorm2d here), the classifier (`head.*`) and every
`layer_scale` parameter (shape [C] in the release, [1, C, 1, 1] here), which stay at their initial values.

    python -m simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.weight_convert.convert_van_weight_from_pytorch_offical_weight \
        --network van_b2 --src van_b2.pth --dst van_b2_converted.pth
"""
import argparse

import torch

_BN_FIELDS = ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked')
# the reference's `filter_list` (:14-37), generated instead of spelled out
SKIPPED_KEYS = frozenset([f'norm{i}.{f}' for i in range(1, 5) for f in _BN_FIELDS] + ['head.weight', 'head.bias'])


def convert_official_van_state_dict(checkpoint, model):
    """-> (kept dict, source keys the target does not have, source keys skipped by rule)"""
    source = checkpoint['state_dict'] if isinstance(checkpoint, dict) and 'state_dict' in checkpoint else checkpoint
    target = model.state_dict()
    kept, foreign, skipped = {}, [], []
    for key, value in source.items():
        if key not in target:
            foreign.append(key)
        elif key in SKIPPED_KEYS or 'layer_scale' in key:
            skipped.append(key)
        else:
            kept[key] = value
    return kept, foreign, skipped


def main():
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--network', default='van_b2')
    ap.add_argument('--num-classes', type=int, default=1000)
    ap.add_argument('--src', required=True)
    ap.add_argument('--dst', required=True)
    args = ap.parse_args()
    from .. import backbones
    model = backbones.__dict__[args.network](**{'num_classes': args.num_classes})
    kept, foreign, skipped = convert_official_van_state_dict(torch.load(args.src, map_location='cpu', weights_only=True), model)
    bad = [k for k, v in kept.items() if tuple(v.shape) != tuple(model.state_dict()[k].shape)]
    print(f'kept {len(kept)} tensors, {len(foreign)} not in the model, {len(skipped)} skipped by rule, {len(bad)} with another shape: {bad[:5]}')
    torch.save(kept, args.dst)


if __name__ == '__main__':
    main()
