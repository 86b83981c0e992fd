"""Registry + builders with the reference's names (rsiseg/models/builder.py:7-88): ONE registry aliased as
BACKBONES / HEADS / LOSSES / SEGMENTORS / UDA; `build_*(cfg)` = look up cfg['type'] and call it with the
remaining keys; `build_train_model(cfg)` injects the student model cfg and max_iters into cfg.uda."""
import copy
import warnings


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._modules[key] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key):
        _register_builtin_types()
        return self._modules.get(key)

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise KeyError(f'cfg must be a dict with a "type" key, got {cfg}')
        args = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        t = args.pop('type')
        cls = self.get(t) if isinstance(t, str) else t
        if cls is None:
            raise KeyError(f'{t} is not in the {self.name} registry (pfst_amd covers the PFST hot path only)')
        return cls(**args)


_REGISTERED = False


def _register_builtin_types():
    """import the modules whose decorators register the reference's type names (PFGST, PFGSTLoss, EncoderDecoder, ResNetV1c, the
    heads, CrossEntropyLoss), on the first lookup: `import pfst_amd` stays cheap for processes that never build a model -- the data
    loader's worker processes un-pickle pfst_amd.data / pfst_amd.pipeline objects and need neither the kernels' front end nor the
    model code"""
    global _REGISTERED
    if not _REGISTERED:
        _REGISTERED = True              # set first: the modules' own decorators re-enter the registry while they import
        try:
            from . import models, uda  # noqa: F401
        except BaseException:
            _REGISTERED = False         # a failed import (libpfst_hip.so missing, ...) must surface on EVERY lookup, not only the first
            raise


MODELS = Registry('models')
BACKBONES = NECKS = HEADS = LOSSES = SEGMENTORS = DISCRIMINATORS = UDA = MODELS


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def _check_cfgs(model_cfg, train_cfg, test_cfg):
    if train_cfg is not None or test_cfg is not None:
        warnings.warn('train_cfg and test_cfg is deprecated, please specify them in model', UserWarning)
    assert model_cfg.get('train_cfg') is None or train_cfg is None, \
        'train_cfg specified in both outer field and model field '
    assert model_cfg.get('test_cfg') is None or test_cfg is None, \
        'test_cfg specified in both outer field and model field '


def build_segmentor(cfg, train_cfg=None, test_cfg=None):
    _check_cfgs(cfg, train_cfg, test_cfg)
    return SEGMENTORS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_train_model(cfg, train_cfg=None, test_cfg=None):
    """cfg: a pfst_amd.config.Config (or any object with .model/.uda/.runner and `in`)."""
    _check_cfgs(cfg.model, train_cfg, test_cfg)
    if 'uda' in cfg:
        cfg.uda['model'] = cfg.model
        cfg.uda['max_iters'] = cfg.runner.max_iters
        return UDA.build(cfg.uda, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))
    return SEGMENTORS.build(cfg.model, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


def add_prefix(inputs, prefix):
    """rsiseg/core/utils/misc.py:2-18"""
    return {f'{prefix}.{k}': v for k, v in inputs.items()}


def deep_copy_cfg(cfg):
    return copy.deepcopy(cfg)
