#!/usr/bin/env python3
"""Generate golden vectors by EXECUTING the reference's own hot-path files on CPU.

Run ONLY in the build container (needs /root/reference); the produced .npz files
are committed and are the only thing that travels to the GPU box.  Nothing from
the reference (source, bytecode, pickles) is copied: fixtures hold seeded inputs
and the numeric outputs the reference produced for them.

The reference (rsiseg) depends on mmcv==1.7.1 / timm / kornia, none of which is
installed here and none of which is vendored under /root/reference (SURVEY.md
§8c).  The reference FILES on the path are loaded unmodified by file path; the
handful of third-party names they import are provided by the loader shim below
(registry plumbing, nn.Module base classes, and mmcv's `ConvModule` /
`DepthwiseSeparableConvModule`, which are documented compositions
conv -> BatchNorm2d(eps=1e-5, momentum=0.1) -> ReLU(inplace) with
`bias = not with_norm`; torch supplies every bit of arithmetic).  kornia's
jitter/blur cannot be reproduced this way and are disabled for the fixtures
(parity of those two transforms is "unpinned").

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
import importlib.util
import os
import random
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------
# loader shim for the third-party names the reference files import
# --------------------------------------------------------------------------
def _install_loader_shims():
    class Registry:
        def __init__(self, name, parent=None, **kw):
            self.name, self.parent, self._m = name, parent, {}

        def register_module(self, name=None, force=False, module=None):
            def deco(cls):
                self._m[name or cls.__name__] = cls
                return cls
            return deco(module) if module is not None else deco

        def get(self, key):
            if key in self._m:
                return self._m[key]
            return self.parent.get(key) if self.parent is not None else None

        def build(self, cfg, default_args=None):
            return build_from_cfg(cfg, self, default_args)

    def build_from_cfg(cfg, registry, default_args=None):
        args = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        cls = registry.get(args.pop('type'))
        assert cls is not None, cfg
        return cls(**args)

    class BaseModule(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()
            self.init_cfg = init_cfg

        def init_weights(self):
            pass

    class Sequential(BaseModule, nn.Sequential):
        def __init__(self, *args, init_cfg=None):
            BaseModule.__init__(self, init_cfg)
            nn.Sequential.__init__(self, *args)

    def _noop_deco(*a, **k):
        def d(f):
            return f
        return d

    def build_conv_layer(cfg, *args, **kwargs):
        assert cfg is None or cfg.get('type', 'Conv2d') in ('Conv2d', 'Conv')
        return nn.Conv2d(*args, **kwargs)

    def build_norm_layer(cfg, num_features, postfix=''):
        assert cfg['type'] == 'BN'
        bn = nn.BatchNorm2d(num_features, eps=cfg.get('eps', 1e-5))
        for p in bn.parameters():
            p.requires_grad = cfg.get('requires_grad', True)
        return 'bn' + str(postfix), bn

    class ConvModule(nn.Module):
        """mmcv 1.7.1 ConvModule for the ('conv','norm','act') order used here."""

        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                     dilation=1, groups=1, bias='auto', conv_cfg=None, norm_cfg=None,
                     act_cfg=dict(type='ReLU'), inplace=True, **kw):
            super().__init__()
            with_norm = norm_cfg is not None
            if bias == 'auto':
                bias = not with_norm
            self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride,
                                  padding=padding, dilation=dilation, groups=groups, bias=bias)
            self.with_norm, self.with_act = with_norm, act_cfg is not None
            if with_norm:
                self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
                self.add_module(self.norm_name, norm)
            if self.with_act:
                assert act_cfg['type'] == 'ReLU'
                self.activate = nn.ReLU(inplace=inplace)

        def forward(self, x):
            x = self.conv(x)
            if self.with_norm:
                x = getattr(self, self.norm_name)(x)
            if self.with_act:
                x = self.activate(x)
            return x

    class DepthwiseSeparableConvModule(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                     dilation=1, norm_cfg=None, act_cfg=dict(type='ReLU'), **kw):
            super().__init__()
            self.depthwise_conv = ConvModule(in_channels, in_channels, kernel_size, stride=stride,
                                             padding=padding, dilation=dilation, groups=in_channels,
                                             norm_cfg=norm_cfg, act_cfg=act_cfg)
            self.pointwise_conv = ConvModule(in_channels, out_channels, 1,
                                             norm_cfg=norm_cfg, act_cfg=act_cfg)

        def forward(self, x):
            return self.pointwise_conv(self.depthwise_conv(x))

    class DataContainer:
        pass

    class MMDistributedDataParallel(nn.Module):
        pass

    class DropPath(nn.Module):
        pass

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    MODELS = Registry('model')
    mmcv = mod('mmcv', __version__='1.7.1', print_log=lambda *a, **k: None, load=None)
    mmcv.utils = mod('mmcv.utils', Registry=Registry, build_from_cfg=build_from_cfg)
    mod('mmcv.utils.parrots_wrapper', _BatchNorm=nn.modules.batchnorm._BatchNorm)
    mmcv.cnn = mod('mmcv.cnn', MODELS=MODELS, ConvModule=ConvModule,
                   DepthwiseSeparableConvModule=DepthwiseSeparableConvModule,
                   build_conv_layer=build_conv_layer, build_norm_layer=build_norm_layer,
                   build_plugin_layer=None)
    mod('mmcv.cnn.bricks')
    mod('mmcv.cnn.bricks.registry', ATTENTION=Registry('attention'))
    mmcv.runner = mod('mmcv.runner', BaseModule=BaseModule, Sequential=Sequential,
                      auto_fp16=_noop_deco, force_fp32=_noop_deco)
    mmcv.parallel = mod('mmcv.parallel', DataContainer=DataContainer,
                        MMDistributedDataParallel=MMDistributedDataParallel)
    mod('timm')
    mod('timm.models')
    mod('timm.models.layers', DropPath=DropPath)
    mod('kornia')


def _load(dotted, relpath):
    spec = importlib.util.spec_from_file_location(dotted, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[dotted] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    """Exec the unmodified reference hot-path files under their real dotted names."""
    sys.dont_write_bytecode = True
    _install_loader_shims()
    for pkg in ['rsiseg', 'rsiseg.core', 'rsiseg.core.utils', 'rsiseg.ops', 'rsiseg.models',
                'rsiseg.models.utils', 'rsiseg.models.backbones', 'rsiseg.models.decode_heads',
                'rsiseg.models.losses', 'rsiseg.models.segmentors', 'rsiseg.models.uda']:
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    R = sys.modules
    misc = _load('rsiseg.core.utils.misc', 'rsiseg/core/utils/misc.py')
    R['rsiseg.core'].add_prefix = misc.add_prefix
    R['rsiseg.core'].build_pixel_sampler = lambda *a, **k: None
    wr = _load('rsiseg.ops.wrappers', 'rsiseg/ops/wrappers.py')
    R['rsiseg.ops'].resize = wr.resize
    b = _load('rsiseg.models.builder', 'rsiseg/models/builder.py')
    R['rsiseg.models'].builder = b
    R['rsiseg.models'].UDA = b.UDA
    R['rsiseg.models'].build_segmentor = b.build_segmentor
    rl = _load('rsiseg.models.utils.res_layer', 'rsiseg/models/utils/res_layer.py')
    R['rsiseg.models.utils'].ResLayer = rl.ResLayer
    dacs = _load('rsiseg.models.utils.dacs_transforms', 'rsiseg/models/utils/dacs_transforms.py')
    _load('rsiseg.models.backbones.resnet', 'rsiseg/models/backbones/resnet.py')
    _load('rsiseg.models.losses.utils', 'rsiseg/models/losses/utils.py')
    acc = _load('rsiseg.models.losses.accuracy', 'rsiseg/models/losses/accuracy.py')
    R['rsiseg.models.losses'].accuracy = acc.accuracy
    ce = _load('rsiseg.models.losses.cross_entropy_loss', 'rsiseg/models/losses/cross_entropy_loss.py')
    pl = _load('rsiseg.models.losses.pfgst_loss', 'rsiseg/models/losses/pfgst_loss.py')
    _load('rsiseg.models.decode_heads.decode_head', 'rsiseg/models/decode_heads/decode_head.py')
    _load('rsiseg.models.decode_heads.aspp_head', 'rsiseg/models/decode_heads/aspp_head.py')
    _load('rsiseg.models.decode_heads.sep_aspp_head', 'rsiseg/models/decode_heads/sep_aspp_head.py')
    _load('rsiseg.models.decode_heads.fcn_head', 'rsiseg/models/decode_heads/fcn_head.py')
    base = _load('rsiseg.models.segmentors.base', 'rsiseg/models/segmentors/base.py')
    R['rsiseg.models'].BaseSegmentor = base.BaseSegmentor
    _load('rsiseg.models.segmentors.encoder_decoder', 'rsiseg/models/segmentors/encoder_decoder.py')
    _load('rsiseg.models.uda.uda_decorator', 'rsiseg/models/uda/uda_decorator.py')
    pf = _load('rsiseg.models.uda.pfgst', 'rsiseg/models/uda/pfgst.py')
    # hard-coded .cuda() calls in the reference (decode_head.py:209, pfgst_loss.py:225)
    torch.Tensor.cuda = lambda self, *a, **k: self
    return types.SimpleNamespace(builder=b, dacs=dacs, ce=ce, acc=acc, pfgst_loss=pl,
                                 pfgst=pf, misc=misc, resize=wr.resize)


# --------------------------------------------------------------------------
# configs (values of configs/_base_/models/deeplabv3plus_r50-d8.py,
# configs/_base_/uda/pfst.py and configs/pfst/pfst_pots_irrg2vaih_irrg_*.py)
# --------------------------------------------------------------------------
def model_cfg(num_classes=6, in_channels=3):
    norm_cfg = dict(type='BN', requires_grad=True)
    return dict(
        type='EncoderDecoder', pretrained=None,
        backbone=dict(type='ResNetV1c', depth=50, num_stages=4, out_indices=(0, 1, 2, 3),
                      dilations=(1, 1, 2, 4), strides=(1, 2, 1, 1), norm_cfg=norm_cfg,
                      norm_eval=False, style='pytorch', contract_dilation=True,
                      in_channels=in_channels),
        decode_head=dict(type='DepthwiseSeparableASPPHead', in_channels=2048, in_index=3,
                         channels=512, dilations=(1, 12, 24, 36), c1_in_channels=256,
                         c1_channels=48, dropout_ratio=0.1, num_classes=num_classes,
                         norm_cfg=norm_cfg, align_corners=False,
                         loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)),
        auxiliary_head=dict(type='FCNHead', in_channels=1024, in_index=2, channels=256, num_convs=1,
                            concat_input=False, dropout_ratio=0.1, num_classes=num_classes,
                            norm_cfg=norm_cfg, align_corners=False,
                            loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.4)),
        train_cfg=dict(), test_cfg=dict(mode='whole'))


def uda_cfg(num_classes=6, in_channels=3, dropout=0.1):
    m = model_cfg(num_classes, in_channels)
    m['decode_head']['dropout_ratio'] = dropout
    m['auxiliary_head']['dropout_ratio'] = dropout
    return dict(
        type='PFGST', alpha=0.999, pseudo_threshold=0.98, pseudo_weight_ignore_top=0,
        pseudo_weight_ignore_bottom=0, imnet_feature_dist_lambda=0, imnet_feature_dist_classes=None,
        imnet_feature_dist_scale_min_ratio=None, mix='class',
        blur=False, color_jitter_strength=0.2, color_jitter_probability=2.0,  # kornia branches off
        print_grad_magnitude=False, thre_type='all', trg_loss_weight=1., use_decoded_feats=True,
        aux_losses=[dict(type='PFGSTLoss', kernel_size=3, dilation=2, top_k=3,
                         weights={'src_pos': 0.1, 'src_neg': 0.1, 'sim_pos': 0.1, 'sim_neg': 0.1,
                                  'src_pos_std': 0.1, 'src_neg_std': 0.1},
                         sim_type='cosine', feat_level=None, detach_unfold=True, downscale=0.5)],
        model=m, max_iters=40000)


sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
from pfst_amd.synthetic import NORM_CFG, fill_state_dict, synth_batch  # noqa: E402


def np_sd(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


# --------------------------------------------------------------------------
# fixture writers
# --------------------------------------------------------------------------
def gen_small_ops(ref):
    """Loss / accuracy / mixing / PFGSTLoss known answers from the reference functions."""
    out = {}
    g = torch.Generator().manual_seed(7)
    # cross entropy (cross_entropy_loss.py:12-65,220-283) incl. pixel weights + ignore
    logits = torch.randn(2, 6, 24, 24, generator=g) * 3
    label = torch.randint(0, 6, (2, 24, 24), generator=g)
    label[:, :3, :5] = 255
    weight = torch.rand(2, 24, 24, generator=g)
    L = ref.builder.build_loss(dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.4))
    out['ce_logits'], out['ce_label'], out['ce_weight'] = logits.numpy(), label.numpy(), weight.numpy()
    out['ce_loss_w'] = L(logits, label, weight=weight, ignore_index=255).numpy()
    out['ce_loss_now'] = L(logits, label, ignore_index=255).numpy()
    Lc = ref.builder.build_loss(dict(type='CrossEntropyLoss', class_weight=[0.5, 1, 1.5, 2, 0.7, 1.2]))
    out['ce_loss_cw'] = Lc(logits, label, weight=weight, ignore_index=255).numpy()
    out['acc'] = ref.acc.accuracy(logits, label, ignore_index=255).numpy()
    # class-mix (dacs_transforms.py:110-144) with the NumPy RNG stream the reference consumes
    gt = torch.randint(0, 6, (3, 1, 4, 4), generator=g).repeat_interleave(8, 2).repeat_interleave(8, 3)
    gt[:, :, :2, :2] = 255
    np.random.seed(11)
    masks = ref.dacs.get_class_masks(gt)
    out['mix_gt'] = gt.numpy()
    out['mix_masks'] = torch.cat(masks).numpy()
    img = torch.randn(3, 3, 32, 32, generator=g)
    trg = torch.randn(3, 3, 32, 32, generator=g)
    pl = torch.randint(0, 6, (3, 32, 32), generator=g)
    pw = torch.full((3, 32, 32), 0.37)
    sp = dict(mix=None, color_jitter=0.0, color_jitter_s=0.2, color_jitter_p=2.0, blur=0,
              mean=torch.tensor(NORM_CFG['mean']).view(1, 3, 1, 1),
              std=torch.tensor(NORM_CFG['std']).view(1, 3, 1, 1), denorm_type='mean_std')
    mi, ml, mw = [], [], []
    for i in range(3):
        sp['mix'] = masks[i]
        a, b_ = ref.dacs.strong_transform(sp, data=torch.stack((img[i], trg[i])),
                                          target=torch.stack((gt[i][0], pl[i])))
        _, w = ref.dacs.strong_transform(sp, target=torch.stack((torch.ones(32, 32), pw[i])))
        mi.append(a), ml.append(b_), mw.append(w.reshape(1, 32, 32))
    out.update(mix_img=img.numpy(), mix_trg=trg.numpy(), mix_pl=pl.numpy(), mix_pw=pw.numpy(),
               mixed_img=torch.cat(mi).numpy(), mixed_lbl=torch.cat(ml).numpy(),
               mixed_w=torch.cat(mw).numpy())
    # PFGSTLoss (pfgst_loss.py:44-234) values + gradients
    B, C, S = 2, 6, 128           # logits at S/4=32 -> downscale -> 16 ; feats at S/8=16
    lt = (torch.randn(B, C, S // 4, S // 4, generator=g) * 2).requires_grad_()
    xe = torch.randn(B, 32, S // 8, S // 8, generator=g)
    xs = torch.randn(B, 32, S // 8, S // 8, generator=g).requires_grad_()
    gts = torch.randint(0, C, (B, 1, 4, 4), generator=g).repeat_interleave(S // 4, 2).repeat_interleave(S // 4, 3)
    gts[:, :, :8, :8] = 255
    mm = (torch.rand(B, 1, 2, 2, generator=g) > 0.6).long().repeat_interleave(S // 2, 2).repeat_interleave(S // 2, 3)
    PL = ref.builder.build_loss(uda_cfg()['aux_losses'][0])
    res = PL(dict(logits_trg=lt, logits_ema=None, gt_src=gts, x_ema=xe, x_src=xs,
                  img_trg=None, mix_masks=mm))
    names = ['loss_src_pos_mean', 'loss_src_neg_mean', 'loss_src_pos_std', 'loss_src_neg_std',
             'loss_sim_pos', 'loss_sim_neg']
    tot = sum(res[n].sum() for n in names)
    tot.backward()
    out.update(pl_logits_trg=lt.detach().numpy(), pl_x_ema=xe.numpy(), pl_x_src=xs.detach().numpy(),
               pl_gt_src=gts.numpy(), pl_mix_masks=mm.numpy(),
               pl_losses=np.array([float(res[n].sum()) for n in names], dtype=np.float64),
               pl_grad_logits=lt.grad.numpy(), pl_grad_xsrc=xs.grad.numpy(),
               pl_vis_density=res['vis|density_sim_feat'][1].numpy(),
               pl_vis_mask=res['vis|density_sim_feat'][2].numpy())
    np.savez_compressed(os.path.join(OUT, 'small_ops.npz'), **out)
    print('small_ops.npz', {k: v.shape for k, v in out.items()})


PFGST_OPTION_VARIANTS = {
    # name -> overrides of the PFGSTLoss config (options reachable from the same configs)
    'gaussian': dict(sim_type='gaussian', sigma=8.0),
    'margin': dict(src_loss_type='margin', margin=[0.5, 0.1]),
    'margin2': dict(src_loss_type='margin2', margin=[0.6, 0.0]),
    'unfold_grad': dict(detach_unfold=False),
    'all_pairs': dict(top_k=None),
    'full_res': dict(downscale=None),
    'gaussian_all_unfold': dict(sim_type='gaussian', sigma=6.0, top_k=None, detach_unfold=False, src_loss_type='margin2',
                                margin=[0.7, 0.2]),
}


def gen_pfgst_options(ref):
    """PFGSTLoss (pfgst_loss.py:12-234) under the option variants above: loss values + gradients on one seeded input set."""
    g = torch.Generator().manual_seed(21)
    B, C, S = 2, 6, 128
    lt0 = torch.randn(B, C, S // 4, S // 4, generator=g) * 2
    xe = torch.randn(B, 32, S // 8, S // 8, generator=g)
    xs0 = torch.randn(B, 32, S // 8, S // 8, generator=g)
    gts = torch.randint(0, C, (B, 1, 4, 4), generator=g).repeat_interleave(S // 4, 2).repeat_interleave(S // 4, 3)
    gts[:, :, :8, :8] = 255
    mm = (torch.rand(B, 1, 2, 2, generator=g) > 0.6).long().repeat_interleave(S // 2, 2).repeat_interleave(S // 2, 3)
    out = dict(logits_trg=lt0.numpy(), x_ema=xe.numpy(), x_src=xs0.numpy(), gt_src=gts.numpy(), mix_masks=mm.numpy(),
               variants=np.array(list(PFGST_OPTION_VARIANTS)))
    for name, over in PFGST_OPTION_VARIANTS.items():
        cfg = dict(uda_cfg()['aux_losses'][0])
        cfg.update(over)
        PL = ref.builder.build_loss(cfg)
        lt, xs = lt0.clone().requires_grad_(), xs0.clone().requires_grad_()
        res = PL(dict(logits_trg=lt, logits_ema=None, gt_src=gts, x_ema=xe, x_src=xs, img_trg=None, mix_masks=mm))
        names = [k for k in res if not k.startswith('vis|')]
        tot = sum(res[n].sum() for n in names)
        tot.backward()
        out[name + '|names'] = np.array(names)
        out[name + '|losses'] = np.array([float(res[n].sum()) for n in names], dtype=np.float64)
        out[name + '|grad_logits'] = lt.grad.numpy().copy()
        out[name + '|grad_xsrc'] = xs.grad.numpy().copy()
        out[name + '|density'] = res['vis|density_sim_feat'][1].numpy().copy()
        print(name, dict(zip(names, out[name + '|losses'])))
    np.savez_compressed(os.path.join(OUT, 'pfgst_options.npz'), **out)


PFGST_OPTION_VARIANTS2 = {
    # round 2: the two remaining PFGSTLoss options of SURVEY.md §8 f4 (pfgst_loss.py:34-36,73-75 proj_net; :98-102 src_perc)
    'src_perc': dict(src_perc=0.6),
    'src_perc_margin': dict(src_perc=0.35, src_loss_type='margin', margin=[0.5, 0.1]),
    'proj_net': dict(proj_net_cfg=dict(in_channels=32, out_channels=16)),
    'proj_net_src_perc_all': dict(proj_net_cfg=dict(in_channels=32, out_channels=24), src_perc=0.8, top_k=None),
}


def gen_pfgst_options2(ref):
    """PFGSTLoss with proj_net (a trainable 1x1 convolution applied to BOTH feature maps; its gradient also flows through the
    teacher-side similarity) and src_perc (sort-and-truncate of the source similarities).  Same seeded inputs as gen_pfgst_options;
    the projection's randomly initialised weights are stored with the fixture."""
    g = torch.Generator().manual_seed(21)
    B, C, S = 2, 6, 128
    lt0 = torch.randn(B, C, S // 4, S // 4, generator=g) * 2
    xe = torch.randn(B, 32, S // 8, S // 8, generator=g)
    xs0 = torch.randn(B, 32, S // 8, S // 8, generator=g)
    gts = torch.randint(0, C, (B, 1, 4, 4), generator=g).repeat_interleave(S // 4, 2).repeat_interleave(S // 4, 3)
    gts[:, :, :8, :8] = 255
    mm = (torch.rand(B, 1, 2, 2, generator=g) > 0.6).long().repeat_interleave(S // 2, 2).repeat_interleave(S // 2, 3)
    out = dict(logits_trg=lt0.numpy(), x_ema=xe.numpy(), x_src=xs0.numpy(), gt_src=gts.numpy(), mix_masks=mm.numpy(),
               variants=np.array(list(PFGST_OPTION_VARIANTS2)))
    for i, (name, over) in enumerate(PFGST_OPTION_VARIANTS2.items()):
        cfg = dict(uda_cfg()['aux_losses'][0])
        cfg.update(over)
        torch.manual_seed(100 + i)                         # nn.Conv2d's default initialisation of proj_net
        PL = ref.builder.build_loss(cfg)
        lt, xs = lt0.clone().requires_grad_(), xs0.clone().requires_grad_()
        res = PL(dict(logits_trg=lt, logits_ema=None, gt_src=gts, x_ema=xe, x_src=xs, img_trg=None, mix_masks=mm))
        names = [k for k in res if not k.startswith('vis|')]
        tot = sum(res[n].sum() for n in names)
        tot.backward()
        out[name + '|names'] = np.array(names)
        out[name + '|losses'] = np.array([float(res[n].sum()) for n in names], dtype=np.float64)
        out[name + '|grad_logits'] = lt.grad.numpy().copy()
        out[name + '|grad_xsrc'] = xs.grad.numpy().copy()
        out[name + '|density'] = res['vis|density_sim_feat'][1].numpy().copy()
        if PL.proj_net is not None:
            out[name + '|proj_weight'] = PL.proj_net.weight.detach().numpy().copy()
            out[name + '|proj_bias'] = PL.proj_net.bias.detach().numpy().copy()
            out[name + '|grad_proj_weight'] = PL.proj_net.weight.grad.numpy().copy()
            out[name + '|grad_proj_bias'] = PL.proj_net.bias.grad.numpy().copy()
        print(name, dict(zip(names, out[name + '|losses'])))
    np.savez_compressed(os.path.join(OUT, 'pfgst_options2.npz'), **out)


def gen_uda_dataset(ref):
    """UDADataset (rsiseg/datasets/uda_dataset.py:17-135): index pairing and rare-class sampling on toy datasets; records the
    RCS class probabilities and the (source index, class-pixel count, target index) sequence drawn under a NumPy seed."""
    import json
    import tempfile
    # the file needs `mmcv.print_log` and the DATASETS registry only
    pkg = types.ModuleType('rsiseg.datasets')
    pkg.__path__ = []
    sys.modules['rsiseg.datasets'] = pkg
    bmod = types.ModuleType('rsiseg.datasets.builder')
    bmod.DATASETS = sys.modules['mmcv.utils'].Registry('dataset')
    sys.modules['rsiseg.datasets.builder'] = bmod
    sys.modules['mmcv'].print_log = lambda *a, **k: None
    ud = _load('rsiseg.datasets.uda_dataset', 'rsiseg/datasets/uda_dataset.py')

    rng = np.random.RandomState(5)
    n_src, n_trg, C = 7, 4, 5

    class Toy(list):
        ignore_index, CLASSES, PALETTE = 255, tuple('abcde'), None

    # every access of a source sample yields a new "crop": the count of class pixels cycles through a fixed list
    crops = {i: [int(v) for v in rng.randint(0, 4000, size=6)] for i in range(n_src)}
    calls = {i: 0 for i in range(n_src)}

    class Source(Toy):
        def __getitem__(self, i):
            k = calls[i] % 6
            calls[i] += 1
            gt = torch.zeros(1, 64, 64, dtype=torch.long)
            gt.view(-1)[:crops[i][k]] = self.want
            return dict(img=i, crop=k, gt_semantic_seg=types.SimpleNamespace(data=gt))

    src = Source(range(n_src))
    src.want = 0
    src.img_infos = [dict(ann=dict(seg_map=f'dir/src_{i}.png')) for i in range(n_src)]
    trg = Toy(dict(img=100 + j, img_metas=dict(j=j), img_strong_aug=200 + j) for j in range(n_trg))
    stats = [dict(file=f'dir/src_{i}.png', **{str(c): int(rng.randint(1, 10000)) for c in range(C) if (i + c) % 3}) for i in range(n_src)]
    swc = {str(c): [[f'dir/src_{i}.png', int(st[str(c)])] for i, st in enumerate(stats) if str(c) in st] for c in range(C)}
    out = dict(stats=json.dumps(stats), samples_with_class=json.dumps(swc), crops=json.dumps(crops), n_src=n_src, n_trg=n_trg)
    with tempfile.TemporaryDirectory() as d:
        json.dump(stats, open(os.path.join(d, 'sample_class_stats.json'), 'w'))
        json.dump(swc, open(os.path.join(d, 'samples_with_class.json'), 'w'))
        cfg = dict(source=dict(data_root=d), path2name=True,
                   rare_class_sampling=dict(class_temp=0.5, min_crop_ratio=0.5, min_pixels=3000))
        ds = ud.UDADataset(src, trg, cfg)
        out['rcs_classes'] = np.array(ds.rcs_classes)
        out['rcs_classprob'] = np.array(ds.rcs_classprob, dtype=np.float64)
        seq = []
        np.random.seed(13)
        _choice = np.random.choice

        def spy(a, *args, **kw):                     # record which class was drawn so that the toy source can count its pixels
            r = _choice(a, *args, **kw)
            if 'p' in kw:
                src.want = int(r)
            return r
        np.random.choice = spy
        try:
            for _ in range(40):
                s = ds[0]
                seq.append((s['img'], s['crop'], s['target_img']))
        finally:
            np.random.choice = _choice
        out['rcs_sequence'] = np.array(seq)
        plain = ud.UDADataset(src, trg, dict(source=dict(data_root=d)))
        out['plain_len'] = len(plain)
        calls.update({i: 0 for i in range(n_src)})
        out['plain_pairs'] = np.array([(plain[i]['img'], plain[i]['target_img'], plain[i]['target_img_strong_aug']) for i in range(len(plain))])
    np.savez_compressed(os.path.join(OUT, 'uda_dataset.npz'), **out)
    print('uda_dataset.npz', out['rcs_classes'], out['rcs_classprob'], out['rcs_sequence'][:5].tolist())


def gen_uda_dataset_v2(ref):
    """UDADatasetV2 (rsiseg/datasets/uda_dataset_v2.py:43-140, the season_net config's pairing): len = len(source), item idx pairs
    source[idx] with a target drawn by np.random.choice BEFORE the source item is produced; records the pairs under a NumPy seed."""
    pkg = types.ModuleType('rsiseg.datasets')
    pkg.__path__ = []
    sys.modules['rsiseg.datasets'] = pkg
    bmod = types.ModuleType('rsiseg.datasets.builder')
    bmod.DATASETS = sys.modules['mmcv.utils'].Registry('dataset')
    sys.modules['rsiseg.datasets.builder'] = bmod
    sys.modules['mmcv'].print_log = lambda *a, **k: None
    ud = _load('rsiseg.datasets.uda_dataset_v2', 'rsiseg/datasets/uda_dataset_v2.py')
    n_src, n_trg = 9, 4

    class Toy(list):
        ignore_index, CLASSES, PALETTE = 255, tuple('abcde'), None

    class Source(Toy):
        def __getitem__(self, i):
            return dict(img=i, draw=int(np.random.randint(0, 1000)))      # a pipeline that consumes the global NumPy stream

    src = Source(range(n_src))
    src.img_infos = [dict(ann=dict(seg_map=f'dir/src_{i}.png')) for i in range(n_src)]
    trg = Toy(dict(img=100 + j, img_metas=dict(j=j), img_strong_aug=200 + j, ori_img=300 + j) for j in range(n_trg))
    ds = ud.UDADatasetV2(src, trg, dict(source=dict(data_root='.')))
    np.random.seed(21)
    rows = []
    for rep in range(3):
        for i in range(len(ds)):
            s = ds[i]
            rows.append((s['img'], s['draw'], s['target_img'], s['target_img_strong_aug'], s['target_img_ori'], s['target_img_metas']['j']))
    np.savez_compressed(os.path.join(OUT, 'uda_dataset_v2.npz'), n_src=n_src, n_trg=n_trg, length=len(ds), pairs=np.array(rows))
    print('uda_dataset_v2.npz', len(ds), rows[:4])


def gen_pipeline_steps(ref):
    """ClipNormalize / Uint82Float of the season_net pipelines (rsiseg/datasets/pipelines/transforms.py:1166-1221) on seeded inputs:
    pure NumPy classes; the module's other imports (mmcv helpers, skimage) are satisfied by empty stand-ins that these two never touch."""
    pkg = types.ModuleType('rsiseg.datasets')
    pkg.__path__ = []
    sys.modules['rsiseg.datasets'] = pkg
    sub = types.ModuleType('rsiseg.datasets.pipelines')
    sub.__path__ = []
    sys.modules['rsiseg.datasets.pipelines'] = sub
    bmod = types.ModuleType('rsiseg.datasets.builder')
    bmod.PIPELINES = sys.modules['mmcv.utils'].Registry('pipeline')
    sys.modules['rsiseg.datasets.builder'] = bmod
    sys.modules.setdefault('skimage', types.ModuleType('skimage'))
    mu = sys.modules['mmcv.utils']
    if not hasattr(mu, 'deprecated_api_warning'):
        mu.deprecated_api_warning = lambda *a, **k: (lambda f: f)
    if not hasattr(mu, 'is_tuple_of'):
        mu.is_tuple_of = lambda seq, t: isinstance(seq, tuple) and all(isinstance(v, t) for v in seq)
    tf = _load('rsiseg.datasets.pipelines.transforms', 'rsiseg/datasets/pipelines/transforms.py')
    rng = np.random.RandomState(17)
    img = rng.randint(0, 6000, size=(21, 19, 3)).astype(np.uint16)
    cfg = dict(mean=[817.83099309, 817.90637517, 613.89910777], std=[1152.3451639, 1081.4451218, 1107.54732507], to_rgb=True, to_uint8=True)
    out = dict(img=img, mean=np.array(cfg['mean']), std=np.array(cfg['std']))
    r = tf.ClipNormalize(**cfg)(dict(img=img.copy(), img_fields=['img']))
    out['clip_u8'] = r['img']
    r2 = tf.ClipNormalize(mean=cfg['mean'], std=cfg['std'], to_rgb=False, to_uint8=False)(dict(img=img.copy(), img_fields=['img']))
    out['clip_f32_bgr'] = r2['img']
    r3 = tf.Uint82Float()(dict(img=r['img'].copy(), aug=r['img'][::-1].copy(), img_fields=['img', 'aug']))
    out['u8_to_float'] = r3['img']
    out['u8_to_float_aug'] = r3['aug']
    np.savez_compressed(os.path.join(OUT, 'pipeline_steps.npz'), **out)
    print('pipeline_steps.npz', out['clip_u8'].dtype, out['clip_u8'][0, 0], out['clip_f32_bgr'].dtype, out['u8_to_float'].dtype)


def gen_segmentor(ref):
    """EncoderDecoder.forward_train + backward (BASELINE config #1 shape, reduced) and the
    teacher-style encode_decode.  Weights = pfst_amd.synthetic.fill_state_dict(seed=5), which the tests rebuild bit-identically."""
    torch.manual_seed(0)
    C, S, b = 6, 64, 2
    cfg = model_cfg(C)
    cfg['decode_head']['dropout_ratio'] = 0.0
    cfg['auxiliary_head']['dropout_ratio'] = 0.0
    model = ref.builder.build_segmentor(cfg)
    sd = model.state_dict()
    fill_state_dict(sd, 5)
    model.load_state_dict(sd)
    model.train()
    batch = synth_batch(b, S, C, seed=1234)
    w = torch.rand(b, S, S, generator=torch.Generator().manual_seed(3))
    losses = model.forward_train(batch['img'], batch['img_metas'], batch['gt_semantic_seg'], w,
                                 return_feats=True, return_logits=True, return_decoded_feats=True)
    feats, logits, dec = losses.pop('features'), losses.pop('logits'), losses.pop('decoded_features')
    loss, log_vars = model._parse_losses(losses)
    loss.backward()
    out = dict(keys=np.array(list(sd.keys())), shapes=np.array([str(tuple(v.shape)) for v in sd.values()]))
    out.update(logits=logits.detach().numpy(), decoded=dec.detach().numpy(),
               c1=feats[0].detach().numpy()[:, :8], c4=feats[3].detach().numpy()[:, :8],
               pix_weight=w.numpy(), loss=np.float64(loss.item()),
               log_keys=np.array(list(log_vars.keys())), log_vals=np.array(list(log_vars.values())))
    grads = {n: p.grad for n, p in model.named_parameters()}
    for n in ['backbone.stem.0.weight', 'backbone.layer1.0.conv1.weight', 'backbone.layer2.0.conv2.weight',
              'backbone.layer3.2.conv2.weight', 'backbone.layer4.2.bn3.weight', 'backbone.layer4.0.downsample.0.weight',
              'decode_head.aspp_modules.2.depthwise_conv.conv.weight', 'decode_head.image_pool.1.conv.weight',
              'decode_head.conv_seg.weight', 'decode_head.conv_seg.bias', 'decode_head.c1_bottleneck.bn.bias',
              'auxiliary_head.convs.0.conv.weight', 'decode_head.sep_bottleneck.0.depthwise_conv.conv.weight']:
        out['grad|' + n] = grads[n].numpy().reshape(grads[n].shape[0], -1)[:16, :32].copy()
    out['grad_norms'] = np.array([float(grads[n].norm()) for n in grads])
    out['grad_names'] = np.array(list(grads))
    # running stats after one train-mode forward
    sd2 = model.state_dict()
    out['rm|backbone.layer2.1.bn2'] = sd2['backbone.layer2.1.bn2.running_mean'].numpy().copy()
    out['rv|backbone.layer2.1.bn2'] = sd2['backbone.layer2.1.bn2.running_var'].numpy().copy()
    # teacher style forward (BN in train mode, no aux head) -> full-res logits
    with torch.no_grad():
        ema_logits, st = model.encode_decode(batch['target_img'], batch['target_img_metas'])
    out['ema_logits'] = ema_logits.numpy()
    np.savez_compressed(os.path.join(OUT, 'segmentor.npz'), **out)
    print('segmentor.npz loss', loss.item(), log_vars)


def gen_train_step(ref):
    """Two full PFGST.train_step iterations (pfgst.py:129-356) with AdamW, b=2, S=128."""
    torch.manual_seed(0)
    C, S, b = 6, 128, 2
    cfg = uda_cfg(C, dropout=0.0)
    model = ref.builder.UDA.build(cfg)
    sd = model.state_dict()
    fill_state_dict(sd, 9)
    model.load_state_dict(sd)
    model.train()
    keys = list(sd.keys())
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad],
                            lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01)
    out = dict(keys=np.array(keys), n_student_params=np.int64(sum(p.numel() for p in model.model.parameters())))
    random.seed(0)
    np.random.seed(0)
    for it in range(2):
        batch = synth_batch(b, S, C, seed=1234 + it)
        # pseudo labels with random-init weights are never confident: lower tau so q is in (0,1)
        model.pseudo_threshold = 0.30
        res = model.train_step(batch, opt)
        lv = res['log_vars']
        out[f'it{it}_log_keys'] = np.array(list(lv.keys()))
        out[f'it{it}_log_vals'] = np.array([float(v) for v in lv.values()], dtype=np.float64)
        st = res['states']
        out[f'it{it}_mixed_lbl'] = st['vis|seg_mask_mix'][1].numpy().astype(np.int64)
        out[f'it{it}_mix_pred'] = st['vis|seg_mask_mix'][2].numpy().astype(np.int64)
        out[f'it{it}_ignore_mask_trg'] = st['vis|density_sim_feat'][2].numpy()
        _assert_live_target_side(lv, st, f'train_step it{it}')
        if it == 0:
            g = {n: p.grad for n, p in model.model.named_parameters()}
            out['it0_grad_norms'] = np.array([float(v.norm()) for v in g.values()])
            out['it0_grad|decode_head.conv_seg.weight'] = g['decode_head.conv_seg.weight'].numpy()
            out['it0_grad|backbone.stem.0.weight'] = g['backbone.stem.0.weight'].numpy()
            # the layers right below the mixed-pass logits, where PFGSTLoss's target-side gradient (through softmax(logits_trg))
            # joins the cross-entropy gradient (pfgst_loss.py:203-234 -> decode_head.py:98-103)
            for n in GRAD_SAMPLES:
                out['it0_grad|' + n] = g[n].numpy().reshape(g[n].shape[0], -1)[:32, :64].copy()
        print('it', it, lv)
    sd2 = model.state_dict()
    for k in ['model.backbone.stem.0.weight', 'model.decode_head.conv_seg.weight',
              'model.backbone.layer3.0.conv2.weight', 'ema_model.backbone.stem.0.weight',
              'ema_model.decode_head.conv_seg.weight', 'ema_model.backbone.layer3.0.bn1.running_mean',
              'model.backbone.layer3.0.bn1.running_mean', 'model.backbone.layer3.0.bn1.running_var']:
        out['final|' + k] = sd2[k].numpy().reshape(-1)[:4096].copy()
    np.savez_compressed(os.path.join(OUT, 'train_step.npz'), **out)


def gen_train_step_512(ref):
    """One full PFGST.train_step of the reference at b = 2 x 512^2 (BASELINE config #1's tile size), the input of
    tests/test_train_step_gpu.py::test_train_step_at_512_matches_oracle (same seeds): pins the mid-size whole-step comparison -- where
    the HIP path's tile chains, multi-round grids and split-K engage -- to the EXECUTED reference, not only to the oracle."""
    torch.manual_seed(0)
    C, S, b = 6, 512, 2
    cfg = uda_cfg(C, dropout=0.0)
    model = ref.builder.UDA.build(cfg)
    sd = model.state_dict()
    fill_state_dict(sd, 9)
    model.load_state_dict(sd)
    model.train()
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01)
    random.seed(101)
    np.random.seed(101)
    batch = synth_batch(b, S, C, seed=4242)
    model.pseudo_threshold = 0.30
    res = model.train_step(batch, opt)
    lv, st = res['log_vars'], res['states']
    out = dict(log_keys=np.array(list(lv.keys())), log_vals=np.array([float(v) for v in lv.values()], dtype=np.float64),
               mixed_lbl=st['vis|seg_mask_mix'][1].numpy().astype(np.uint8), mix_pred=st['vis|seg_mask_mix'][2].numpy().astype(np.uint8),
               ignore_mask_trg=st['vis|density_sim_feat'][2].numpy(), np_state_after=np.random.get_state()[1][:8].copy())
    _assert_live_target_side(lv, st, 'train_step_512')
    g = {n: p.grad for n, p in model.model.named_parameters()}
    out['grad_names'] = np.array(list(g.keys()))
    out['grad_norms'] = np.array([float(v.norm()) for v in g.values()])
    for n in GRAD_SAMPLES + ['decode_head.conv_seg.weight', 'auxiliary_head.conv_seg.weight']:
        out['grad|' + n] = g[n].numpy().reshape(g[n].shape[0], -1)[:32, :64].copy()
    np.savez_compressed(os.path.join(OUT, 'train_step_512.npz'), **out)
    print('train_step_512.npz', lv)


GRAD_SAMPLES = ['decode_head.conv_seg.bias', 'decode_head.sep_bottleneck.1.pointwise_conv.conv.weight',
                'decode_head.sep_bottleneck.1.pointwise_conv.bn.weight', 'decode_head.sep_bottleneck.0.depthwise_conv.conv.weight',
                'decode_head.bottleneck.bn.bias', 'backbone.layer4.2.bn3.weight']


def _assert_live_target_side(lv, st, what, min_frac=0.15):
    """The fixture must exercise the target-side half of PFGSTLoss (pfgst_loss.py:62-71,203-234): a valid region well above the
    `<= 1 pixel -> zeros(1)` cut-off and non-zero loss_sim_* (round 2's fixtures had 0-2 valid pixels: VERDICT r2 weak #1)."""
    m = st['vis|density_sim_feat'][2]
    frac = float(m.float().mean())
    assert frac >= min_frac, (what, 'un-mixed target region', frac)
    assert lv['loss_sim_pos'] != 0.0 and lv['loss_sim_neg'] != 0.0, (what, lv)
    print(f'  {what}: all-nine-unmixed region {int(m.sum())} of {m.numel()} grid pixels, loss_sim_pos {lv["loss_sim_pos"]:.6f}')


STEP_VARIANTS = {
    # name: (PFGST cfg overrides, PFGSTLoss overrides, pseudo_threshold)
    'no_mix': (dict(apply_no_mix=True), {}, 0.30),
    'thre_part': (dict(thre_type='part'), {}, 0.30),
    'ignore_rows': (dict(pseudo_weight_ignore_top=16, pseudo_weight_ignore_bottom=8), {}, 0.30),
    'feat_level2': (dict(use_decoded_feats=False), dict(feat_level=2), 0.30),
    'trg_weight': (dict(trg_loss_weight=0.5), {}, 0.30),
}


def gen_train_step_variants(ref, only=None):
    """One PFGST.train_step of the reference per PFGST-level option (pfgst.py:229-231,255-257,259-276,283-289,310): apply_no_mix,
    thre_type='part', pseudo_weight_ignore_top/bottom, use_decoded_feats=False + PFGSTLoss(feat_level), trg_loss_weight."""
    for name, (over, loss_over, tau) in STEP_VARIANTS.items():
        if only and name not in only:
            continue
        torch.manual_seed(0)
        C, S, b = 6, 128, 2
        cfg = uda_cfg(C, dropout=0.0)
        cfg.update(over)
        cfg['aux_losses'][0].update(loss_over)
        model = ref.builder.UDA.build(cfg)
        sd = model.state_dict()
        fill_state_dict(sd, 9)
        model.load_state_dict(sd)
        model.train()
        opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01)
        random.seed(3)
        np.random.seed(3)
        batch = synth_batch(b, S, C, seed=77)
        model.pseudo_threshold = tau
        res = model.train_step(batch, opt)
        lv = res['log_vars']
        st = res['states']
        out = dict(log_keys=np.array(list(lv.keys())), log_vals=np.array([float(v) for v in lv.values()], dtype=np.float64),
                   mixed_lbl=st['vis|seg_mask_mix'][1].numpy().astype(np.int64), np_state_after=np.random.get_state()[1][:8].copy())
        out['ignore_mask_trg'] = st['vis|density_sim_feat'][2].numpy()
        _assert_live_target_side(lv, st, f'train_step_{name}')
        g = {n: p.grad for n, p in model.model.named_parameters()}
        out['grad_norms'] = np.array([float(v.norm()) for v in g.values()])
        for n in GRAD_SAMPLES[:3]:
            out['grad|' + n] = g[n].numpy().reshape(g[n].shape[0], -1)[:32, :64].copy()
        np.savez_compressed(os.path.join(OUT, f'train_step_{name}.npz'), **out)
        print(f'train_step_{name}.npz', lv)


if __name__ == '__main__':
    torch.set_num_threads(8)
    ref = load_reference()
    which = sys.argv[1:] or ['small', 'options', 'options2', 'dataset', 'dataset2', 'pipeline', 'seg', 'step', 'variants', 'step512']
    if 'options2' in which:
        gen_pfgst_options2(ref)
    if 'small' in which:
        gen_small_ops(ref)
    if 'options' in which:
        gen_pfgst_options(ref)
    if 'dataset' in which:
        gen_uda_dataset(ref)
    if 'dataset2' in which:
        gen_uda_dataset_v2(ref)
    if 'pipeline' in which:
        gen_pipeline_steps(ref)
    if 'seg' in which:
        gen_segmentor(ref)
    if 'step' in which:
        gen_train_step(ref)
    if 'variants' in which:
        gen_train_step_variants(ref)
    if 'step512' in which:
        gen_train_step_512(ref)
