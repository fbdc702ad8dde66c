"""Official ConvFormer (MetaFormer baselines) checkpoint -> the weights this package's ConvFormer loads.

Behaviour of reference SimpleAICV/classification/weight_convert/convert_convformer_weight_from_pytorch_offical_weight.py:48-58:
keep every source tensor whose key the target model has WITH THE SAME SHAPE; everything else (the release's LayerNorm / StarReLU /
scale parameters this simplified ConvFormer does not have, the 21k classifier ...) is dropped and reported.

    python -m simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.weight_convert.convert_convformer_weight_from_pytorch_offical_weight \
        --network convformer_s18 --src convformer_s18.pth --dst convformer_s18_converted.pth
"""
import argparse

import torch


def convert_official_convformer_state_dict(source, model):
    """-> (kept dict, source keys the target does not have, keys with another shape)"""
    target = model.state_dict()
    kept, foreign, reshaped = {}, [], []
    for key, value in source.items():
        if key not in target:
            foreign.append(key)
        elif tuple(value.shape) != tuple(target[key].shape):
            reshaped.append(key)
        else:
            kept[key] = value
    return kept, foreign, reshaped


def main():
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--network', default='convformer_s18')
    ap.add_argument('--num-classes', type=int, default=1000)
    ap.add_argument('--src', required=True)
    ap.add_argument('--dst', required=True)
    args = ap.parse_args()
    from .. import backbones
    model = backbones.__dict__[args.network](**{'num_classes': args.num_classes})
    kept, foreign, reshaped = convert_official_convformer_state_dict(torch.load(args.src, map_location='cpu', weights_only=True), model)
    print(f'kept {len(kept)} tensors, {len(foreign)} not in the model, {len(reshaped)} with another shape')
    torch.save(kept, args.dst)


if __name__ == '__main__':
    main()
