"""Official-weight key converters of the interactive-segmentation models (reference SimpleAICV/interactive_segmentation/weight_convert/)."""
from .sam_encoder_weight_convert_from_sam_offical_weight import convert_official_sam_encoder_state_dict  # noqa: F401
