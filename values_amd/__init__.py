"""values_amd -- MI355X-native hot path of ValUES' multi-pass segmentation-uncertainty inference."""
from .unet3d import UNet3D  # noqa: F401
from .ssn import SsnUNet3D  # noqa: F401
from .uncertainty import calculate_one_minus_msr, calculate_uncertainty, softmax_variance, uncertainty_maps  # noqa: F401
from .predict import GraphedPredictor, HostPipeline, predict_logits, predict_uncertainty, crop_indices  # noqa: F401
from .io import load_models_from_checkpoint, instantiate  # noqa: F401
from .sliding import predict_image_sliding  # noqa: F401
