"""SAM (image encoder + prompt encoder + mask decoder) -- drop-in for the reference factories.

Interface contract (reference SimpleAICV/interactive_segmentation/models/segment_anything/sam.py): SAM (:25)
with forward / forward_image_encoder (:119) / forward_prompt_encoder_mask_decoder (:124), factories sam_b /
sam_l / sam_h (:181-215); identical constructor arguments and state_dict keys.  sam_h (head dim 80) is
declared but its encoder raises: the streaming attention kernel is instantiated for head dims 32 and 64.
"""
import torch.nn as nn
import torch.nn.functional as F

from ..... import ops_tfm

from .image_encoder import ViTImageEncoder
from .mask_decoder import MaskDecoder
from .prompt_encoder import PromptEncoder

__all__ = [
    'sam_b',
    'sam_l',
    'sam_h',
]


class SAM(nn.Module):

    def __init__(self, image_size=1024, patch_size=16, inplanes=3, image_encoder_embedding_planes=768,
                 image_encoder_block_nums=12, image_encoder_head_nums=12, image_encoder_mlp_ratio=4,
                 image_encoder_window_size=14, image_encoder_global_attn_indexes=[2, 5, 8, 11],
                 prompt_encoder_embedding_planes=256, prompt_encoder_mask_inter_planes=16,
                 mask_decoder_num_multimask_outputs=3, mask_decoder_iou_prediction_head_block_nums=3,
                 mask_decoder_iou_prediction_head_hidden_planes=256, use_gradient_checkpoint=False,
                 frozen_image_encoder=False, frozen_prompt_encoder=False, frozen_mask_decoder=False):
        super(SAM, self).__init__()
        self.image_encoder = ViTImageEncoder(
            image_size=image_size, patch_size=patch_size, inplanes=inplanes,
            embedding_planes=image_encoder_embedding_planes, block_nums=image_encoder_block_nums,
            head_nums=image_encoder_head_nums, mlp_ratio=image_encoder_mlp_ratio,
            out_planes=prompt_encoder_embedding_planes, window_size=image_encoder_window_size,
            global_attn_indexes=image_encoder_global_attn_indexes, use_gradient_checkpoint=use_gradient_checkpoint)
        self.prompt_encoder = PromptEncoder(image_size=image_size, patch_size=patch_size,
                                            embedding_planes=prompt_encoder_embedding_planes,
                                            mask_inter_planes=prompt_encoder_mask_inter_planes)
        self.mask_decoder = MaskDecoder(
            inplanes=prompt_encoder_embedding_planes, num_multimask_outputs=mask_decoder_num_multimask_outputs,
            iou_prediction_head_block_nums=mask_decoder_iou_prediction_head_block_nums,
            iou_prediction_head_hidden_planes=mask_decoder_iou_prediction_head_hidden_planes)
        # the prompt encoder / mask decoder run 1 + decoder_iters times per training step
        # (tools/interactive_segmentation_scripts.py): every pass adds its share to the arena gradient in place; the
        # engine learns that a gradient is complete from autograd's post-accumulate hook, which runs after the last use
        if frozen_image_encoder:
            for param in self.image_encoder.parameters():
                param.requires_grad = False
        if frozen_prompt_encoder:
            for param in self.prompt_encoder.parameters():
                param.requires_grad = False
        if frozen_mask_decoder:
            for param in self.mask_decoder.parameters():
                param.requires_grad = False

    def forward(self, batch_images, batch_prompts, mask_out_idxs=[0, 1, 2, 3]):
        return self.forward_prompt_encoder_mask_decoder(self.image_encoder(batch_images), batch_prompts,
                                                        mask_out_idxs=mask_out_idxs)

    def forward_image_encoder(self, batch_images):
        return self.image_encoder(batch_images)

    def forward_prompt_encoder_mask_decoder(self, batch_image_embeddings, batch_prompts, mask_out_idxs=[0, 1, 2, 3]):
        device = batch_image_embeddings.device
        points = batch_prompts['prompt_point']
        boxes = batch_prompts['prompt_box']
        mask = batch_prompts['prompt_mask']
        points = points.to(device) if points is not None else None
        boxes = boxes.to(device) if boxes is not None else None
        mask = mask.to(device) if mask is not None else None
        sparse_embeddings, dense_embeddings = self.prompt_encoder(points=points, boxes=boxes, masks=mask)
        mask_preds, iou_preds = self.mask_decoder(image_embeddings=batch_image_embeddings,
                                                  image_pe=self.prompt_encoder.get_dense_pe_layer(),
                                                  sparse_prompt_embeddings=sparse_embeddings,
                                                  dense_prompt_embeddings=dense_embeddings,
                                                  mask_out_idxs=mask_out_idxs)
        size = self.image_encoder.image_size
        if mask_preds.shape[-2] * 4 == size and mask_preds.shape[-1] * 4 == size:
            # x4 bilinear on the HIP kernel.  The low-resolution logits ride along on the returned tensor: SAMLoss takes
            # its statistics and its gradient straight from them (csrc/samtail.hip), so the [B, M, 1024, 1024] tensor
            # is written once for the prompt sampler and never read, nor its gradient materialised, by the loss
            low = mask_preds.contiguous()
            mask_preds = ops_tfm.upsample4_bilinear(low)
            mask_preds._saicv_low = low
        else:
            mask_preds = F.interpolate(mask_preds, (size, size), mode="bilinear")
        return mask_preds, iou_preds


def _sam(image_size, patch_size, image_encoder_embedding_planes, image_encoder_block_nums, image_encoder_head_nums,
         image_encoder_global_attn_indexes, prompt_encoder_embedding_planes, **kwargs):
    return SAM(image_size=image_size, patch_size=patch_size,
               image_encoder_embedding_planes=image_encoder_embedding_planes,
               image_encoder_block_nums=image_encoder_block_nums, image_encoder_head_nums=image_encoder_head_nums,
               image_encoder_global_attn_indexes=image_encoder_global_attn_indexes,
               prompt_encoder_embedding_planes=prompt_encoder_embedding_planes, **kwargs)


def sam_b(image_size=1024, patch_size=16, **kwargs):
    return _sam(image_size=image_size, patch_size=patch_size, image_encoder_embedding_planes=768,
                image_encoder_block_nums=12, image_encoder_head_nums=12,
                image_encoder_global_attn_indexes=[2, 5, 8, 11], prompt_encoder_embedding_planes=256, **kwargs)


def sam_l(image_size=1024, patch_size=16, **kwargs):
    return _sam(image_size=image_size, patch_size=patch_size, image_encoder_embedding_planes=1024,
                image_encoder_block_nums=24, image_encoder_head_nums=16,
                image_encoder_global_attn_indexes=[5, 11, 17, 23], prompt_encoder_embedding_planes=256, **kwargs)


def sam_h(image_size=1024, patch_size=16, **kwargs):
    return _sam(image_size=image_size, patch_size=patch_size, image_encoder_embedding_planes=1280,
                image_encoder_block_nums=32, image_encoder_head_nums=16,
                image_encoder_global_attn_indexes=[7, 15, 23, 31], prompt_encoder_embedding_planes=256, **kwargs)
