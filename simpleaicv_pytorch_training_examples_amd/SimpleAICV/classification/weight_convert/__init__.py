"""Official-weight key converters (reference SimpleAICV/classification/weight_convert/): library functions + small CLIs."""
from .convert_resnet_weight_from_pytorch_offical_weight import convert_torchvision_resnet_state_dict  # noqa: F401
from .convert_vit_mae_weight_from_offical_mae_weight import convert_official_mae_state_dict  # noqa: F401
from .convert_van_weight_from_pytorch_offical_weight import convert_official_van_state_dict
from .convert_convformer_weight_from_pytorch_offical_weight import convert_official_convformer_state_dict
