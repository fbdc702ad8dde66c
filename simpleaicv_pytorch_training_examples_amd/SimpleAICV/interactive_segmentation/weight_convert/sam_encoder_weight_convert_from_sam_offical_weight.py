"""Official Segment-Anything checkpoint (sam_vit_{b,l,h}_*.pth) -> the image-encoder weights `ViTImageEncoder` loads on its own
(`backbone_pretrained_path` of the SAM factories).

Behaviour of reference SimpleAICV/interactive_segmentation/weight_convert/sam_encoder_weight_convert_from_sam_offical_weight.py:44-52:
of the `image_encoder.*` tensors keep those the SAM model here has under the same key with the same shape, and strip the prefix.

    python -m simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation.weight_convert.sam_encoder_weight_convert_from_sam_offical_weight \
        --network sam_b --src sam_vit_b_01ec64.pth --dst sam_vit_b_encoder_converted.pth
"""
import argparse

import torch

PREFIX = 'image_encoder.'


def convert_official_sam_encoder_state_dict(source, model):
    """-> (encoder dict without the prefix, `image_encoder.*` source keys the model does not have, keys with another shape)"""
    target = model.state_dict()
    kept, foreign, reshaped = {}, [], []
    for key, value in source.items():
        if PREFIX not in key:
            continue
        if key not in target:
            foreign.append(key)
        elif tuple(value.shape) != tuple(target[key].shape):
            reshaped.append(key)
        else:
            kept[key.replace(PREFIX, '')] = value
    return kept, foreign, reshaped


def main():
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--network', default='sam_b')
    ap.add_argument('--src', required=True)
    ap.add_argument('--dst', required=True)
    args = ap.parse_args()
    from ..models.segment_anything import sam
    model = sam.__dict__[args.network](**{})
    kept, foreign, reshaped = convert_official_sam_encoder_state_dict(torch.load(args.src, map_location='cpu', weights_only=True), model)
    print(f'kept {len(kept)} encoder tensors, {len(foreign)} not in the model, {len(reshaped)} with another shape')
    torch.save(kept, args.dst)


if __name__ == '__main__':
    main()
