"""pfst_amd: MI355X-native PFST (PFGST) self-training train step.

Importing the package registers the reference's type names (PFGST, PFGSTLoss, EncoderDecoder, ResNetV1c,
DepthwiseSeparableASPPHead, FCNHead, CrossEntropyLoss) in `pfst_amd.registry.MODELS`.  All compute goes
through libpfst_hip.so (include/pfst_hip.h); there is no CPU or eager-PyTorch fallback."""
from . import models, uda  # noqa: F401  (registration side effects)
from .config import Config
from .registry import MODELS, UDA, build_segmentor, build_train_model

__all__ = ['Config', 'MODELS', 'UDA', 'build_segmentor', 'build_train_model']
