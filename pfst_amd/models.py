"""ResNetV1c-50-d8 backbone, DepthwiseSeparableASPPHead, FCNHead, CrossEntropyLoss and EncoderDecoder on
the HIP ops, registered under the reference's type names so `configs/pfst/*.py` build unchanged.

Reference behaviour followed (file:line under /root/reference/rsiseg/models):
  backbones/resnet.py:99-307,396-527,591-638,659-700  utils/res_layer.py:28-96
  decode_heads/decode_head.py:55-108,188-283  aspp_head.py:53-126  sep_aspp_head.py:29-111  fcn_head.py:24-98
  segmentors/encoder_decoder.py:65-217  losses/cross_entropy_loss.py:198-298
Only what the PFST hot path uses is implemented; unsupported options raise instead of silently differing."""
import os

import numpy as np
import torch
import torch.nn as nn

from . import hip_ops as ops
from . import layers as layers_mod
from .engine import Var
from .layers import (BatchNorm2dP, Conv2dP, ConvModule, DepthwiseSeparableConvModule, WeightBatch, bn_eval, conv_bn_act,
                     conv_forward, dwsep_branches)
from .registry import BACKBONES, HEADS, LOSSES, SEGMENTORS, add_prefix, build_backbone, build_head, build_loss


# False: the decode head's Dropout2d as a scaling pass of its own (the fold on / off test); default: folded into sep_bottleneck[1]'s normalisation
FOLD_DROPOUT = True


def xin_dev(v):
    return (v.data if v.lazy is None else v.lazy[0]).device


def _check_norm(norm_cfg):
    if norm_cfg is None or norm_cfg.get('type') != 'BN':
        raise NotImplementedError(f'pfst_amd implements plain BN (the PFST configs); got norm_cfg={norm_cfg}')


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=False):
        super().__init__()
        self.conv1 = Conv2dP(inplanes, planes, 1)
        self.bn1 = BatchNorm2dP(planes)
        self.conv2 = Conv2dP(planes, planes, 3, stride, dilation, dilation)   # style='pytorch': stride on the 3x3
        self.bn2 = BatchNorm2dP(planes)
        self.conv3 = Conv2dP(planes, planes * 4, 1)
        self.bn3 = BatchNorm2dP(planes * 4)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(Conv2dP(inplanes, planes * 4, 1, stride), BatchNorm2dP(planes * 4))

    def forward(self, x, tape):
        # conv2 through the Winograd domain (layer2.1 ... layer4.2): its input transform normalises conv1's output as it loads it -- y1 is never
        # written; the weight gradient of conv2 comes from the transformed input kept in forward, BatchNorm backward of bn1 from the pre-BN tensor
        h, w = x.data.shape[-2:]
        fold = layers_mod.FOLD_BN_WINO and self.conv2.wino and self.conv2.bias is None and (tape is None or self.conv2.wino_wgrad_ok(h, w))
        o = conv_bn_act(x, self.conv1, self.bn1, tape, defer='amax' if fold else False)
        # conv3 (1x1 on the 256-row f16x3 tile) normalises conv2's output between load and split, and so does its weight gradient: y2 is never
        # written either (layers.FOLD_BN_GEMM); conv2 -- the Winograd output transform, or layer1's direct GEMM -- emits the (min, max) partials
        # that predict max |y2|
        fold2 = layers_mod.FOLD_BN_GEMM and self.conv3.fprop_bnl_ok() and (h // self.conv2.stride) * (w // self.conv2.stride) % 128 == 0
        o = conv_bn_act(o, self.conv2, self.bn2, tape, defer='amax' if fold2 else False)
        idt = x
        if self.downsample is not None:
            idt = conv_bn_act(x, self.downsample[0], self.downsample[1], tape, relu=False, defer='residual')     # layers.FOLD_BN_RESIDUAL: normalised by bn3's pass
        return conv_bn_act(o, self.conv3, self.bn3, tape, relu=True, residual=idt)


@BACKBONES.register_module()
class ResNetV1c(nn.Module):
    """ResNet-50 with the 3x(3x3) deep stem, strides (1,2,1,1), dilations (1,1,2,4), contract_dilation."""
    arch = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}

    def __init__(self, depth=50, in_channels=3, stem_channels=64, base_channels=64, num_stages=4,
                 strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style='pytorch',
                 norm_cfg=dict(type='BN', requires_grad=True), norm_eval=False, contract_dilation=False,
                 multi_grid=None, with_cp=False, zero_init_residual=True, pretrained=None, init_cfg=None,
                 frozen_stages=-1, conv_cfg=None, dcn=None, plugins=None, avg_down=False, deep_stem=True,
                 stage_with_dcn=(False, False, False, False)):
        super().__init__()
        _check_norm(norm_cfg)
        if depth not in self.arch:
            raise KeyError(f'invalid depth {depth} for resnet')
        unsupported = dict(style=style != 'pytorch', norm_eval=norm_eval, multi_grid=multi_grid is not None, with_cp=with_cp,
                           frozen_stages=frozen_stages >= 0, conv_cfg=conv_cfg is not None, dcn=dcn is not None,
                           plugins=plugins is not None, avg_down=avg_down, deep_stem=not deep_stem)
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(f'ResNetV1c options outside the PFST path: {bad}')
        self.pretrained = pretrained
        self.out_indices = tuple(out_indices)
        self.zero_init_residual = zero_init_residual
        c2 = stem_channels // 2
        ident = nn.Identity
        self.stem = nn.Sequential(Conv2dP(in_channels, c2, 3, 2, 1), BatchNorm2dP(c2), ident(),
                                  Conv2dP(c2, c2, 3, 1, 1), BatchNorm2dP(c2), ident(),
                                  Conv2dP(c2, stem_channels, 3, 1, 1), BatchNorm2dP(stem_channels), ident())
        inplanes = stem_channels
        self.res_layers = []
        for i, nb in enumerate(self.arch[depth][:num_stages]):
            planes = base_channels * 2 ** i
            stride, dil = strides[i], dilations[i]
            first_dil = dil // 2 if (dil > 1 and contract_dilation) else dil
            blocks = [Bottleneck(inplanes, planes, stride, first_dil, downsample=(stride != 1 or inplanes != planes * 4))]
            inplanes = planes * 4
            for _ in range(1, nb):
                blocks.append(Bottleneck(inplanes, planes, 1, dil))
            name = f'layer{i + 1}'
            self.add_module(name, nn.Sequential(*blocks))
            self.res_layers.append(name)

    def init_weights(self):
        """resnet.py:430-452 with pretrained=None: Kaiming (fan_out, relu) convolutions -- done at construction --, BatchNorm weight 1 /
        bias 0, and with zero_init_residual the LAST norm of every Bottleneck starts at 0 (each block starts as the identity).
        A `pretrained` checkpoint path is loaded instead (mmcv 'Pretrained' init: backbone keys, strict=False); model-zoo URLs such as
        open-mmlab://resnet50_v1c cannot be resolved here (no network) and fail loudly."""
        if self.pretrained is not None:
            import os
            if not (isinstance(self.pretrained, str) and os.path.exists(self.pretrained)):
                raise FileNotFoundError(f'pretrained={self.pretrained!r} is not a local checkpoint file (no network here): pass a file, '
                                        'use --load-from, or set pretrained=None for random initialisation')
            ckpt = torch.load(self.pretrained, map_location='cpu', weights_only=False)
            sd = ckpt.get('state_dict', ckpt)
            sd = {(k[len('backbone.'):] if k.startswith('backbone.') else k): v for k, v in sd.items()}
            self.load_state_dict(sd, strict=False)
            return
        for m in self.modules():
            if isinstance(m, BatchNorm2dP):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.zeros_(m.bn3.weight)

    def forward(self, x, tape=None, grad_ready=None):
        """grad_ready (data-parallel runs, last pass of the backward sweep only): callable(stage) recorded as marker closures --
        when marker `layer{i}` runs in backward, the gradients of layer{i} and of everything after it are final"""
        s = self.stem
        x = conv_bn_act(x, s[0], s[1], tape)
        x = conv_bn_act(x, s[3], s[4], tape)
        x = conv_bn_act(x, s[6], s[7], tape, defer=True)     # its only consumer, the max-pool, normalises on load: stem.6's output is never written
        xin = x
        pool_amax = ops.amax_slots(xin_dev(xin)) if layers_mod.CONV_MATH == 'f16x3' else None     # layer1's f16x3 GEMMs read the pooled map
        if xin.lazy is not None:
            y, idx = ops.maxpool(xin.lazy[0], bnl=xin.lazy[1], amax=pool_amax)
        else:
            y, idx = ops.maxpool(xin.data, amax=pool_amax)
        in_hw = tuple((xin.data if xin.lazy is None else xin.lazy[0]).shape[-2:])
        x = Var(y, tape is not None)
        x.amax = pool_amax
        if tape is not None:
            pooled = x

            def bwd_pool():
                xin._grad = ops.maxpool_bwd(pooled.grad, idx, in_hw)
                pooled.free_grad()
            tape.record(bwd_pool, dict(op='maxpool', name='backbone.maxpool', x=xin, out=pooled))
        outs = []
        for i, name in enumerate(self.res_layers):
            if grad_ready is not None and tape is not None:
                tape.record(lambda name=name: grad_ready(name))
            for blk in getattr(self, name):
                x = blk(x, tape)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)


def get_class_weight(class_weight):
    """losses/utils.py:10-25: a list is taken as is; a str is a file -- .npy via NumPy, otherwise what mmcv.load reads by
    extension (.json, .yaml / .yml, .pkl / .pickle)."""
    if not isinstance(class_weight, str):
        return class_weight
    import numpy as np
    ext = class_weight.rsplit('.', 1)[-1].lower()
    if ext == 'npy':
        return np.load(class_weight).tolist()
    if ext == 'json':
        import json
        with open(class_weight) as f:
            return json.load(f)
    if ext in ('yaml', 'yml'):
        import yaml
        with open(class_weight) as f:
            return yaml.safe_load(f)
    if ext in ('pkl', 'pickle'):
        import pickle
        with open(class_weight, 'rb') as f:
            return pickle.load(f)
    raise TypeError(f'Unsupported class_weight file format: {ext}')


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """Softmax CE with pixel weights, optional class weights, ignore_index, mean over ALL pixels
    (avg_non_ignore=False), fused with the bilinear up-sampling of the logits and the accuracy."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0,
                 loss_name='loss_ce', avg_non_ignore=False):
        super().__init__()
        if use_sigmoid or use_mask or reduction != 'mean' or avg_non_ignore:
            raise NotImplementedError('pfst_amd CrossEntropyLoss: softmax CE, reduction=mean, avg_non_ignore=False only')
        self.class_weight = get_class_weight(class_weight)
        self.loss_weight = loss_weight
        self._loss_name = loss_name
        self._cw = None

    @property
    def loss_name(self):
        return self._loss_name

    def _cw_dev(self, dev):
        if self.class_weight is None:
            return None
        if self._cw is None or self._cw.device != dev:
            self._cw = torch.tensor(self.class_weight, dtype=torch.float32, device=dev)
        return self._cw

    def fused(self, logits, label_u8, weight, tape, ignore_index=255, grad_scale=1.0):
        """logits: Var [N,C,h,w] (low res); label_u8 [N,1,H,W] or [N,H,W]; weight [N,H,W] or None.
        -> device tensor [loss, acc_seg]"""
        ld = logits.data
        n = ld.shape[0]
        H, W = label_u8.shape[-2:]
        cw = self._cw_dev(ld.device)
        lse, acc = ops.ce_upsample_fwd(ld, label_u8, weight, cw, ignore_index)
        out = ops.ce_finalize(acc, n * H * W, self.loss_weight)
        if tape is not None:
            scale = grad_scale * self.loss_weight / float(n * H * W)

            def bwd():
                buf, accf = logits.grad_target()
                ops.ce_upsample_bwd(ld, label_u8, lse, scale, weight, cw, ignore_index, out=buf, accumulate=accf)
            tape.record(bwd, dict(op='ce', x=logits, out=None, label=label_u8, weight=weight, class_weight=self.class_weight,
                                  loss_weight=grad_scale * self.loss_weight, ignore_index=ignore_index))
        return out


class BaseDecodeHead(nn.Module):
    def __init__(self, in_channels, channels, *, num_classes, dropout_ratio=0.1, conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type='ReLU'), in_index=-1, input_transform=None,
                 loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0), ignore_index=255,
                 sampler=None, align_corners=False, init_cfg=None):
        super().__init__()
        _check_norm(norm_cfg)
        if input_transform is not None or sampler is not None or align_corners or conv_cfg is not None:
            raise NotImplementedError('decode head options outside the PFST path')
        if not isinstance(loss_decode, dict):
            raise NotImplementedError('a single loss_decode dict is supported')
        self.in_channels, self.channels, self.num_classes = in_channels, channels, num_classes
        self.dropout_ratio, self.in_index, self.ignore_index = dropout_ratio, in_index, ignore_index
        self.align_corners = align_corners
        self.loss_decode = build_loss(loss_decode)
        self.conv_seg = Conv2dP(channels, num_classes, 1, bias=True)
        nn.init.normal_(self.conv_seg.weight, 0, 0.01)
        self.dropout_enabled = True            # the teacher switches this off (pfgst.py:247-251)
        self.injected_dropout_mask = None      # parity tests inject the (n, C) keep/scale mask

    def dropout_mask(self, n, training):
        """the (n, C) keep / (1 - p) factors of nn.Dropout2d (decode_head.py:103-107), or None when dropout is off (evaluation, the
        teacher, p = 0); the parity tests inject theirs.  Drawn by the head right before its last conv -> BN -> ReLU layer, which
        applies it in its normalisation pass (layers.conv_bn_act(post_scale=...)): the torch generator sees the same draws in the
        same order as when the mask was drawn after that layer (nothing in between draws from it)"""
        p = self.dropout_ratio
        if self.injected_dropout_mask is not None:
            return self.injected_dropout_mask
        if training and self.dropout_enabled and p > 0:
            dev = self.conv_seg.weight.device
            return (torch.rand(n, self.channels, device=dev) >= p).float() / (1.0 - p)
        return None

    def cls_seg(self, feat, tape, training, dropped=False):
        """dropout + conv_seg (decode_head.py:242-247).  dropped: `feat` already carries the Dropout2d factors (folded into the layer that
        produced it); otherwise they are applied here as a pass of their own"""
        mask = None if dropped else self.dropout_mask(feat.data.shape[0], training)
        if mask is not None:
            src = feat
            y = ops.channel_scale(src.data, mask)
            feat = Var(y, tape is not None)
            if tape is not None:
                dropped_v = feat

                def bwd():
                    src._grad = ops.channel_scale(dropped_v.grad, mask)
                    dropped_v.free_grad()
                tape.record(bwd, dict(op='channel_scale', x=src, out=dropped_v, mask=mask))
        return conv_forward(feat, self.conv_seg, tape)

    def losses(self, seg_logit, seg_label_u8, seg_weight, tape, grad_scale=1.0):
        out = self.loss_decode.fused(seg_logit, seg_label_u8, seg_weight, tape, self.ignore_index, grad_scale)
        # '_bad_labels': labels outside [0, C) other than ignore_index (F.cross_entropy raises on them); travels with the packed
        # scalars of the step and is checked on the host after the step's single read (uda.PFGST.forward_train), never logged
        return {self.loss_decode.loss_name: out[0:1], 'acc_seg': out[1:2], '_bad_labels': out[2:3]}

    def forward_train(self, inputs, img_metas, gt_semantic_seg, train_cfg, seg_weight=None, tape=None, grad_scale=1.0):
        seg_logits, feats = self.forward(inputs, return_features=True, tape=tape, training=True)
        losses = self.losses(seg_logits, gt_semantic_seg, seg_weight, tape, grad_scale)
        return losses, {'seg_logits': seg_logits, 'decoded_features': feats}

    def forward_test(self, inputs, img_metas, test_cfg, tape=None):
        seg_logits, feats = self.forward(inputs, return_features=True, tape=None, training=False)
        return seg_logits, {'decoded_features': feats}


@HEADS.register_module()
class DepthwiseSeparableASPPHead(BaseDecodeHead):
    def __init__(self, c1_in_channels, c1_channels, dilations=(1, 6, 12, 18), **kwargs):
        super().__init__(**kwargs)
        assert c1_in_channels > 0 and dilations[0] == 1 and all(d > 1 for d in dilations[1:])
        self.dilations = tuple(dilations)
        ci, ch = self.in_channels, self.channels
        self.image_pool = nn.Sequential(nn.Identity(), ConvModule(ci, ch, 1))
        mods = [ConvModule(ci, ch, 1)]
        mods += [DepthwiseSeparableConvModule(ci, ch, 3, padding=d, dilation=d) for d in dilations[1:]]
        self.aspp_modules = nn.ModuleList(mods)
        self.bottleneck = ConvModule((len(dilations) + 1) * ch, ch, 3, padding=1)
        self.c1_bottleneck = ConvModule(c1_in_channels, c1_channels, 1)
        self.c1_channels = c1_channels
        self.sep_bottleneck = nn.Sequential(DepthwiseSeparableConvModule(ch + c1_channels, ch, 3, padding=1),
                                            DepthwiseSeparableConvModule(ch, ch, 3, padding=1))

    def forward(self, inputs, return_features=False, tape=None, training=True):
        c1, x = inputs[0], inputs[self.in_index]
        n, _, h, w = x.data.shape
        ch, nb = self.channels, len(self.dilations) + 1
        cat = Var(torch.empty(n, nb * ch, h, w, device=x.data.device), tape is not None)
        if layers_mod.CONV_MATH == 'f16x3':
            # every writer of the concat publishes max |.| into ONE group (layers._amax_target for the four slices, the broadcast below): the
            # bottleneck's f16x3 scale needs no pass of its own over the 2560-channel buffer
            cat.amax = ops.amax_slots(x.data.device)
        # The ASPP branches first, the image-pool branch after them (they write disjoint slices, so the forward result does not
        # depend on the order): in backward the pool branch's broadcast then lands in dL/dx BEFORE the 1x1 branch's data gradient,
        # which completes dL/dx and can emit the BatchNorm-backward sums of the layer that produced x (layers._dgrad_into).
        # layers.FOLD_BN_CONCAT: the four conv -> BN -> ReLU writers leave their pre-normalisation outputs in the slices; the bottleneck's Winograd
        # input transform normalises per channel (the image-pool slice holds final values: identity rows) -- four normalisation passes over
        # 512-channel maps less per pass
        bconv = self.bottleneck.conv
        fold = (layers_mod.FOLD_BN_CONCAT and layers_mod.DEFER_BN_APPLY and not layers_mod._BN_EVAL and bconv.wino and bconv.bias is None
                and (tape is None or bconv.wino_wgrad_ok(h, w)))
        if fold:
            cat.coef_table = torch.empty(nb * ch, 4, device=x.data.device)
            cat.coef_table[:ch] = layers_mod.identity_coef_row(x.data.device)          # y = max(fma(v, 1, 0), 0) = v for the (non-negative) pooled values
        sl = 'slice' if fold else False
        self.aspp_modules[0](x, tape, out=cat.slice(ch, 2 * ch), defer=sl)
        nd = len(self.dilations)
        # on the fused path the atrous branches' launch also forms the plane means of x, its backward their adjoint
        pool = {}
        dwsep_branches(x, [self.aspp_modules[i] for i in range(1, nd)], tape, [cat.slice((i + 1) * ch, (i + 2) * ch) for i in range(1, nd)],
                       pool=pool, defer=sl)
        # image pool branch: GAP -> 1x1 conv -> BN over the n samples -> ReLU -> broadcast (bilinear from 1x1)
        fused_pool = pool is not None and 'mean' in pool
        pooled = Var(pool['mean'] if fused_pool else ops.global_avgpool(x.data), tape is not None)
        pa = self.image_pool[1](pooled, tape)
        ops.broadcast_hw(pa.data, cat.data[:, 0:ch], amax=cat.amax)
        if tape is not None:
            hw = float(h * w)

            def bwd_pool():
                g = cat.grad
                pa._grad = ops.reduce_hw(g[:, 0:ch])

            def bwd_gap():
                if fused_pool:                     # handed to the atrous branches' fused backward, which runs later and writes dL/dx once
                    pool['grad'] = pooled.grad
                    pooled.free_grad()
                    return
                buf, acc = x.grad_target()
                if not acc:
                    ops.fill_(buf, 0.0)
                ops.broadcast_hw(pooled.grad, buf, 1.0 / hw, True)
                pooled.free_grad()
            # order on the tape: gap-backward must run AFTER the image_pool conv's backward, pool-broadcast before it
            self._reorder_pool(tape, (bwd_gap, dict(op='gap', name='decode_head.gap', x=x, out=pooled)),
                               (bwd_pool, dict(op='broadcast', name='decode_head.image_pool.up', x=pa, out=cat.slice(0, ch))))
        if fold:
            # every writer deferred?  (a writer that could not -- no (min, max) partials for its shape -- wrote its slice normalised: identity rows)
            cat.lazy = (cat.data, cat.coef_table, None)
        feats = self.bottleneck(cat, tape)
        # decoder: upsample x2 to c1 size, concat with the 48-channel c1 projection
        H, W = c1.data.shape[-2:]
        cat2 = Var(torch.empty(n, ch + self.c1_channels, H, W, device=x.data.device), tape is not None)
        ops.resize_bilinear(feats.data, (H, W), out=cat2.data[:, 0:ch])
        if tape is not None:
            def bwd_up():
                buf, acc = feats.grad_target()
                ops.resize_bilinear_bwd(cat2.grad[:, 0:ch], (h, w), out=buf, accumulate=acc)
            tape.record(bwd_up, dict(op='resize', name='decode_head.up', x=feats, out=cat2.slice(0, ch)))
        self.c1_bottleneck(c1, tape, out=cat2.slice(ch, ch + self.c1_channels))
        o = self.sep_bottleneck[0](cat2, tape, defer=True)          # its only consumer, sep_bottleneck[1]'s depthwise layer, normalises on load
        fold = FOLD_DROPOUT
        mask = self.dropout_mask(n, training) if fold else None
        o = self.sep_bottleneck[1](o, tape, post_scale=mask)          # Dropout2d folded into this layer's normalisation pass
        logits = self.cls_seg(o, tape, training, dropped=fold)
        return (logits, feats) if return_features else logits

    @staticmethod
    def _reorder_pool(tape, bwd_gap, bwd_pool):
        # forward recorded: [..., image_pool ConvModule closure]; we need tape order
        # [bwd_gap, conv closure, bwd_pool] so that reverse execution is pool -> conv -> gap.
        conv_closure = tape.pop()
        tape.record(*bwd_gap)
        tape.record(*conv_closure)
        tape.record(*bwd_pool)


@HEADS.register_module()
class FCNHead(BaseDecodeHead):
    def __init__(self, num_convs=2, kernel_size=3, concat_input=True, dilation=1, **kwargs):
        super().__init__(**kwargs)
        if num_convs != 1 or concat_input or kernel_size != 3:
            raise NotImplementedError('FCNHead: the PFST auxiliary head (num_convs=1, concat_input=False) only')
        self.convs = nn.Sequential(ConvModule(self.in_channels, self.channels, 3, padding=dilation, dilation=dilation))

    def forward(self, inputs, return_features=False, tape=None, training=True):
        # (the features this head returns are the PRE-dropout ones, fcn_head.py:92-98, so its Dropout2d stays a pass of its own -- an eighth
        # of the decode head's tensor; the decode head returns the ASPP bottleneck output and folds its dropout, see above)
        feats = self.convs[0](inputs[self.in_index], tape)
        logits = self.cls_seg(feats, tape, training)
        return (logits, feats) if return_features else logits


@SEGMENTORS.register_module()
class EncoderDecoder(nn.Module):
    def __init__(self, backbone, decode_head, neck=None, auxiliary_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None, init_cfg=None):
        super().__init__()
        if neck is not None or isinstance(auxiliary_head, (list, tuple)):
            raise NotImplementedError('neck / multiple auxiliary heads are outside the PFST path')
        if pretrained is not None:
            backbone = dict(backbone)
            backbone['pretrained'] = pretrained
        self.backbone = build_backbone(backbone)
        self.decode_head = build_head(decode_head)
        self.auxiliary_head = build_head(auxiliary_head) if auxiliary_head is not None else None
        self.align_corners = self.decode_head.align_corners
        self.num_classes = self.decode_head.num_classes
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    @property
    def with_auxiliary_head(self):
        return self.auxiliary_head is not None

    def init_weights(self):
        self.backbone.init_weights()

    def convs(self):
        return [m for m in self.modules() if isinstance(m, Conv2dP)]

    def repack_weights(self, need_dgrad=True):
        """the weight images of every convolution for this step.  The first call walks the layers (modes, buffers, job tables); while the switches
        (layers.repack_key) and the weight buffers stay the same, later calls replay the recorded launches -- ~0.1 instead of ~1.2 ms of host time
        per model, at the step boundary where the device has nothing queued"""
        convs = self.__dict__.get('_conv_list')
        if convs is None:
            convs = self.__dict__['_conv_list'] = self.convs()
        key = layers_mod.repack_key(need_dgrad) + tuple((c.weight.data_ptr(), None if c.bias is None else c.bias.data_ptr()) for c in (convs[0], convs[-1])) \
            + (len(convs),)
        plan = self.__dict__.get('_repack_plan')
        if plan is not None and plan[0] == key and WeightBatch.enabled:
            for fn, a in plan[1]:
                fn(*a)
            self._weight_batch.replay()
            return
        batch, rec = None, None
        if WeightBatch.enabled:
            batch = self.__dict__.setdefault('_weight_batch', WeightBatch())
            batch.begin()
            rec = []
        for m in convs:
            m.repack(need_dgrad, batch, rec)
        if batch is not None:
            batch.flush()
        self.__dict__['_repack_plan'] = (key, rec) if rec is not None else None

    def extract_feat(self, img, tape=None, grad_ready=None):
        x = img if isinstance(img, Var) else Var(img, False)
        return self.backbone(x, tape, grad_ready) if grad_ready is not None else self.backbone(x, tape)

    def encode_decode(self, img, img_metas=None):
        """teacher-style forward: no tape, dropout off, BN still in train mode; returns the LOW-resolution
        logits Var (the bilinear upsampling to the image size is fused into the consumers) + states."""
        x = self.extract_feat(img, None)
        logits, states = self.decode_head.forward_test(x, img_metas, self.test_cfg)
        states.update({'feats': x, 'seg_logits': logits})
        return logits, states

    # ------------------------------------------------------------------ test time (encoder_decoder.py:220-372)
    def _eval_encode_decode(self, img):
        """encode_decode as the test loop runs it (model.eval(): BatchNorm on running statistics, dropout off): the logits resized to the
        INPUT size (encoder_decoder.py:72-84) -> (full-size logits tensor, backbone features, low-resolution logits)"""
        with bn_eval():
            x = self.extract_feat(img.contiguous(), None)
            logits = self.decode_head(x, return_features=False, tape=None, training=False)
        return ops.resize_bilinear(logits.data, tuple(img.shape[2:])), x, logits

    def slide_inference(self, img, img_meta, rescale):
        """Inference by sliding window with overlap (encoder_decoder.py:220-263): crops of test_cfg.crop_size every test_cfg.stride pixels,
        the last window of a row / column shifted back inside the image, crop logits summed into place and divided by the cover count.
        An image smaller than the crop is decoded whole, without padding."""
        h_stride, w_stride = self.test_cfg['stride']
        h_crop, w_crop = self.test_cfg['crop_size']
        n, _, h_img, w_img = img.shape
        num_classes = self.decode_head.num_classes
        h_grids = max(h_img - h_crop + h_stride - 1, 0) // h_stride + 1
        w_grids = max(w_img - w_crop + w_stride - 1, 0) // w_stride + 1
        preds = torch.zeros(n, num_classes, h_img, w_img, device=img.device)
        count = torch.zeros(n, 1, h_img, w_img, device=img.device)
        covered = np.zeros((h_img, w_img), bool)                     # the reference's `assert (count_mat == 0).sum() == 0`, without a device read
        for h_idx in range(h_grids):
            for w_idx in range(w_grids):
                y1, x1 = h_idx * h_stride, w_idx * w_stride
                y2, x2 = min(y1 + h_crop, h_img), min(x1 + w_crop, w_img)
                y1, x1 = max(y2 - h_crop, 0), max(x2 - w_crop, 0)
                crop_logit, _, _ = self._eval_encode_decode(img[:, :, y1:y2, x1:x2])
                ops.window_accumulate_(preds, count, crop_logit, y1, x1)
                covered[y1:y2, x1:x2] = True
        assert covered.all()
        ops.window_normalize_(preds, count)
        if rescale:
            preds = ops.resize_bilinear(preds, tuple(img_meta[0]['ori_shape'][:2]))
        return preds

    def inference_probs(self, img, img_meta=None, rescale=True):
        """`inference` of the reference (encoder_decoder.py:284-327): slide / whole logits -> softmax over the classes -> flipped inputs
        flipped back; -> (probabilities [N, C, H, W], states)"""
        mode = (self.test_cfg or {}).get('mode', 'whole')
        assert mode in ('slide', 'whole')
        if img_meta is not None and 'ori_shape' in img_meta[0]:
            assert all(tuple(m['ori_shape']) == tuple(img_meta[0]['ori_shape']) for m in img_meta)
        rescale = bool(rescale and img_meta is not None and 'ori_shape' in img_meta[0])
        self.repack_weights(need_dgrad=False)
        if mode == 'slide':
            seg_logit, states = self.slide_inference(img, img_meta, rescale), {}
        else:
            seg_logit, x, low = self._eval_encode_decode(img)
            size = tuple(img_meta[0]['ori_shape'][:2]) if rescale else tuple(img.shape[2:])
            if size != tuple(img.shape[2:]):
                seg_logit = ops.resize_bilinear(seg_logit, size)          # whole_inference (:269-280): a second resize from the input size
            states = dict(feats=[v.data for v in x], seg_logits=low.data)
        output = ops.softmax_nchw(seg_logit)
        if img_meta is not None and img_meta[0].get('flip'):
            direction = img_meta[0]['flip_direction']
            for d in (direction if isinstance(direction, list) else [direction]):
                assert d in ('horizontal', 'vertical')
                output = ops.flip_planes(output, horizontal=d == 'horizontal', vertical=d == 'vertical')
        return output, states

    def inference(self, img, img_meta=None, rescale=True):
        """-> (argmax label map uint8 [N,H,W], low-res logits or None).  Whole-image mode (all shipped configs): the softmax of the
        reference (`F.softmax(seg_logit)` then `argmax`) is what the fused upsample+softmax+argmax kernel evaluates (its tie rule: §7).
        As in the reference the logits are first resized to the INPUT size (encode_decode, encoder_decoder.py:77-81) and, when `rescale`
        asks for another `ori_shape`, resized a second time from there (whole_inference :269-280); flipped inputs are flipped back
        (:314-325).  Slide mode: the arg-max of `inference_probs`."""
        if self.test_cfg is not None and self.test_cfg.get('mode', 'whole') != 'whole':
            probs, _ = self.inference_probs(img, img_meta, rescale)
            self._last_states = None                              # the reference returns no states in slide mode (:306-307)
            return ops.argmax_nchw(probs), None
        self.repack_weights(need_dgrad=False)
        with bn_eval():
            x = self.extract_feat(img.contiguous(), None)
            logits = self.decode_head(x, return_features=False, tape=None, training=False)
        in_size = tuple(img.shape[2:])
        size = in_size
        if rescale and img_meta is not None and 'ori_shape' in img_meta[0]:
            assert all(tuple(m['ori_shape']) == tuple(img_meta[0]['ori_shape']) for m in img_meta)
            size = tuple(img_meta[0]['ori_shape'][:2])
        src = logits.data if size == in_size else ops.resize_bilinear(logits.data, in_size)
        _, lab8, _ = ops.pseudo_label(src, size, 2.0, want_i64=False)
        if img_meta is not None and img_meta[0].get('flip'):
            direction = img_meta[0]['flip_direction']
            for d in (direction if isinstance(direction, list) else [direction]):
                assert d in ('horizontal', 'vertical')
                lab8 = lab8.flip(dims=(2,) if d == 'horizontal' else (1,))
            lab8 = lab8.contiguous()
        self._last_states = dict(feats=[v.data for v in x], seg_logits=logits.data)
        return lab8, logits.data

    def simple_test(self, img, img_meta=None, rescale=True):
        """-> (list of per-image label maps, list of per-image state dicts), the fork's contract (encoder_decoder.py:329-353; consumed as
        `result, state = model(return_loss=False, **data)` by apis/test.py:97).  The states hold the backbone features and the low-res
        logits of each image (device tensors; the reference copies them to the host); empty dicts in slide mode."""
        lab8, _ = self.inference(img, img_meta, rescale)
        st = self._last_states
        if st is None:
            states = [dict() for _ in range(lab8.shape[0])]
        else:
            states = [dict(feats=[f[i] for f in st['feats']], seg_logits=st['seg_logits'][i]) for i in range(lab8.shape[0])]
        self._last_states = None
        return list(lab8.cpu().numpy()), states

    def aug_test(self, imgs, img_metas, rescale=True):
        """Test with augmentations (encoder_decoder.py:355-372): the class probabilities of every augmented view, each mapped back to
        `ori_shape` and un-flipped, are averaged; the arg-max of the average is the prediction.  Only rescale=True, like the reference."""
        assert rescale
        seg_logit, _ = self.inference_probs(imgs[0], img_metas[0], rescale)
        for i in range(1, len(imgs)):
            cur, _ = self.inference_probs(imgs[i], img_metas[i], rescale)
            ops.axpy_(seg_logit, cur)
        ops.div_scalar_(seg_logit, len(imgs))
        return list(ops.argmax_nchw(seg_logit).cpu().numpy()), {}

    def forward_test(self, imgs, img_metas=None, **kwargs):
        """base.py:74-99: one view -> simple_test, several -> aug_test"""
        if isinstance(imgs, (list, tuple)):
            if len(imgs) != 1:
                if img_metas is None or len(img_metas) != len(imgs):
                    raise ValueError(f'forward_test with {len(imgs)} augmented views needs one img_metas list per view (ori_shape, flip), '
                                     f'got {None if img_metas is None else len(img_metas)}')
                return self.aug_test(list(imgs), list(img_metas), **kwargs)
            imgs, img_metas = imgs[0], (img_metas[0] if img_metas else None)
        return self.simple_test(imgs, img_metas, **kwargs)

    def forward(self, img, img_metas=None, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    def forward_train(self, img, img_metas, gt_semantic_seg, seg_weight=None, return_feats=False,
                      return_decoded_feats=False, return_logits=False, return_states=False, tape=None, grad_scale=1.0,
                      grad_ready=None):
        """gt_semantic_seg: uint8 [N,1,H,W]; returns the losses dict (device tensors) like the reference.
        grad_ready: see ResNetV1c.forward; the marker `heads` covers both heads"""
        x = self.extract_feat(img, tape, grad_ready)
        if grad_ready is not None and tape is not None:
            tape.record(lambda: grad_ready('heads'))
        losses, states = dict(), dict()
        loss_decode, state = self.decode_head.forward_train(x, img_metas, gt_semantic_seg, self.train_cfg, seg_weight,
                                                            tape=tape, grad_scale=grad_scale)
        losses.update(add_prefix(loss_decode, 'decode'))
        states.update(state)
        if self.with_auxiliary_head:
            loss_aux, state_aux = self.auxiliary_head.forward_train(x, img_metas, gt_semantic_seg, self.train_cfg,
                                                                    seg_weight, tape=tape, grad_scale=grad_scale)
            losses.update(add_prefix(loss_aux, 'aux'))
            states.update(add_prefix(state_aux, 'aux'))
        if return_feats:
            losses['features'] = x
        if return_logits:
            losses['logits'] = state['seg_logits']
        if return_decoded_feats:
            losses['decoded_features'] = state['decoded_features']
        return (losses, states) if return_states else losses
