"""pfst_amd: MI355X-native PFST (PFGST) self-training train step.

The reference's type names (PFGST, PFGSTLoss, EncoderDecoder, ResNetV1c, DepthwiseSeparableASPPHead, FCNHead,
CrossEntropyLoss) resolve through `pfst_amd.registry.MODELS`; the modules defining them are imported on the first registry
lookup (registry._register_builtin_types), so that importing the package -- which the data loader's worker processes do when they
un-pickle the dataset -- pulls in neither the model code nor the ctypes front end of the kernels.  All compute goes through
libpfst_hip.so (include/pfst_hip.h); there is no CPU or eager-PyTorch fallback."""
from .config import Config
from .registry import MODELS, UDA, build_segmentor, build_train_model

__all__ = ['Config', 'MODELS', 'UDA', 'build_segmentor', 'build_train_model']
