"""CPU oracle for the PFST (PFGST) train step -- TEST INFRASTRUCTURE ONLY.

This is a plain-PyTorch fp32 CPU restatement of the reference's hot path.  It is
imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
always as the checker, never as the product path.  The product (pfst_amd) never
imports it and fails loudly without its HIP library.

Pinning (see tests/test_oracle_golden.py): every function here is checked against
fixtures produced by executing the reference's own files (tests/golden/make_golden.py)
and against the known-answer values the reference's tests hold for cross-entropy /
accuracy.  kornia colour-jitter / gaussian-blur are NOT restated: parity unpinned for
those two transforms (third-party, absent from the image).

Structure is deliberately not the reference's module tree: the network is a
function over a flat state_dict that uses the reference's checkpoint key names
(`backbone.stem.0.weight`, `decode_head.aspp_modules.1.depthwise_conv.conv.weight`, ...),
so one state_dict loads into the reference, this oracle and the HIP product alike.

Reference files followed (all under /root/reference/rsiseg):
  models/backbones/resnet.py:99-307,591-674   models/utils/res_layer.py:28-96
  models/decode_heads/{decode_head.py:188-283, aspp_head.py:53-126, sep_aspp_head.py:29-111, fcn_head.py:24-98}
  models/segmentors/{encoder_decoder.py:65-217, base.py:177-222}
  models/losses/{cross_entropy_loss.py:12-65,220-283, utils.py:48-80, accuracy.py:6-61, pfgst_loss.py:44-234}
  models/utils/dacs_transforms.py:110-144     models/uda/pfgst.py:105-166,179-356
"""
import math
import random
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOM = 0.1
STAGE_BLOCKS = (3, 4, 6, 3)
STAGE_PLANES = (64, 128, 256, 512)
STAGE_STRIDES = (1, 2, 1, 1)
STAGE_DILATIONS = (1, 1, 2, 4)
ASPP_DILATIONS = (1, 12, 24, 36)


# ---------------------------------------------------------------------------
# architecture description shared by init + forward
# ---------------------------------------------------------------------------
def conv_table(num_classes=6, in_channels=3):
    """[(conv_key, bn_key|None, cout, cin_per_group, k, bias)] in checkpoint order
    (resnet.py:591-638 stem, :99-307 Bottleneck, res_layer.py:28-96, heads)."""
    t = []
    for i, (ci, co) in zip((0, 3, 6), ((in_channels, 32), (32, 32), (32, 64))):
        t.append((f'backbone.stem.{i}', f'backbone.stem.{i + 1}', co, ci, 3, False))
    inpl = 64
    for li, (nb, pl) in enumerate(zip(STAGE_BLOCKS, STAGE_PLANES)):
        for b in range(nb):
            p = f'backbone.layer{li + 1}.{b}'
            t.append((p + '.conv1', p + '.bn1', pl, inpl, 1, False))
            t.append((p + '.conv2', p + '.bn2', pl, pl, 3, False))
            t.append((p + '.conv3', p + '.bn3', pl * 4, pl, 1, False))
            if b == 0:
                t.append((p + '.downsample.0', p + '.downsample.1', pl * 4, inpl, 1, False))
            inpl = pl * 4
    h = 'decode_head'
    t.append((h + '.conv_seg', None, num_classes, 512, 1, True))
    t.append((h + '.image_pool.1.conv', h + '.image_pool.1.bn', 512, 2048, 1, False))
    t.append((h + '.aspp_modules.0.conv', h + '.aspp_modules.0.bn', 512, 2048, 1, False))
    for i in (1, 2, 3):
        q = f'{h}.aspp_modules.{i}'
        t.append((q + '.depthwise_conv.conv', q + '.depthwise_conv.bn', 2048, 1, 3, False))
        t.append((q + '.pointwise_conv.conv', q + '.pointwise_conv.bn', 512, 2048, 1, False))
    t.append((h + '.bottleneck.conv', h + '.bottleneck.bn', 512, 2560, 3, False))
    t.append((h + '.c1_bottleneck.conv', h + '.c1_bottleneck.bn', 48, 256, 1, False))
    for i, cin in ((0, 560), (1, 512)):
        q = f'{h}.sep_bottleneck.{i}'
        t.append((q + '.depthwise_conv.conv', q + '.depthwise_conv.bn', cin, 1, 3, False))
        t.append((q + '.pointwise_conv.conv', q + '.pointwise_conv.bn', 512, cin, 1, False))
    a = 'auxiliary_head'
    t.append((a + '.conv_seg', None, num_classes, 256, 1, True))
    t.append((a + '.convs.0.conv', a + '.convs.0.bn', 256, 1024, 3, False))
    return t


def init_state_dict(num_classes=6, in_channels=3, seed=0, randomize_bn=True):
    """Seeded random-init state_dict with the reference's key names, order and shapes
    (kaiming-normal fan_out convs, N(0, 0.01) conv_seg: resnet.py:430-452, decode_head.py:73-74).
    The stem/downsample BN keys sit right after their conv like in the checkpoint."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for ck, bk, co, cig, k, bias in conv_table(num_classes, in_channels):
        if ck.endswith('conv_seg'):
            sd[ck + '.weight'] = torch.randn(co, cig, k, k, generator=g) * 0.01
            sd[ck + '.bias'] = torch.zeros(co)
        else:
            std = math.sqrt(2.0 / (co * k * k)) if cig > 1 else math.sqrt(2.0 / (k * k))
            sd[ck + '.weight'] = torch.randn(co, cig, k, k, generator=g) * std
        if bk is not None:
            if randomize_bn:
                sd[bk + '.weight'] = 0.6 + 0.8 * torch.rand(co, generator=g)
                sd[bk + '.bias'] = 0.2 * torch.randn(co, generator=g)
            else:
                sd[bk + '.weight'], sd[bk + '.bias'] = torch.ones(co), torch.zeros(co)
            sd[bk + '.running_mean'] = torch.zeros(co)
            sd[bk + '.running_var'] = torch.ones(co)
            sd[bk + '.num_batches_tracked'] = torch.zeros((), dtype=torch.long)
    return sd


def param_keys(sd):
    return [k for k in sd if not (k.endswith('running_mean') or k.endswith('running_var')
                                  or k.endswith('num_batches_tracked'))]


# ---------------------------------------------------------------------------
# network forward (functional)
# ---------------------------------------------------------------------------
CAPTURE = None      # tests set this to a dict: every named activation is stored with retain_grad(), so that after backward()
                    # each layer's upstream gradient is available (tests/test_layer_backward_gpu.py)


def _cap(name, t):
    if CAPTURE is not None and t.requires_grad:
        t.retain_grad()
        CAPTURE[name] = t
    return t


def _bn(sd, key, x, relu=True, train=True):
    y = F.batch_norm(x, sd[key + '.running_mean'], sd[key + '.running_var'], sd[key + '.weight'],
                     sd[key + '.bias'], training=train, momentum=BN_MOM, eps=BN_EPS)
    if train and (key + '.num_batches_tracked') in sd:
        sd[key + '.num_batches_tracked'] += 1
    return F.relu(y) if relu else y


def _cbr(sd, ck, bk, x, stride=1, pad=0, dil=1, groups=1, relu=True, train=True):
    return _cap(ck, _bn(sd, bk, F.conv2d(x, sd[ck + '.weight'], None, stride, pad, dil, groups), relu, train))


def backbone_forward(sd, x, train=True, pre=''):
    """ResNetV1c-50-d8 (resnet.py:659-674): returns (c1, c2, c3, c4)."""
    b = pre + 'backbone.'
    x = _cbr(sd, b + 'stem.0', b + 'stem.1', x, 2, 1, train=train)
    x = _cbr(sd, b + 'stem.3', b + 'stem.4', x, 1, 1, train=train)
    x = _cbr(sd, b + 'stem.6', b + 'stem.7', x, 1, 1, train=train)
    x = _cap(b + 'maxpool', F.max_pool2d(x, 3, 2, 1))
    outs = []
    for li, (nb, stride, dil) in enumerate(zip(STAGE_BLOCKS, STAGE_STRIDES, STAGE_DILATIONS)):
        for bi in range(nb):
            p = f'{b}layer{li + 1}.{bi}'
            s = stride if bi == 0 else 1
            d = (dil // 2 if dil > 1 else dil) if bi == 0 else dil   # contract_dilation (res_layer.py:68-72)
            idt = x
            o = _cbr(sd, p + '.conv1', p + '.bn1', x, train=train)
            o = _cbr(sd, p + '.conv2', p + '.bn2', o, s, d, d, train=train)
            o = _cbr(sd, p + '.conv3', p + '.bn3', o, relu=False, train=train)
            if bi == 0:
                idt = _cbr(sd, p + '.downsample.0', p + '.downsample.1', x, s, relu=False, train=train)
            x = _cap(p + '.out', F.relu(o + idt))
        outs.append(x)
    return tuple(outs)


def _dwsep(sd, q, x, dil, train):
    c = x.shape[1]
    x = _cbr(sd, q + '.depthwise_conv.conv', q + '.depthwise_conv.bn', x, 1, dil, dil, c, train=train)
    return _cbr(sd, q + '.pointwise_conv.conv', q + '.pointwise_conv.bn', x, train=train)


def decode_head_forward(sd, feats, train=True, drop_mask=None, pre=''):
    """DepthwiseSeparableASPPHead.forward (sep_aspp_head.py:79-111) -> (logits, features)."""
    h = pre + 'decode_head'
    c1, c4 = feats[0], feats[3]
    pool = _cbr(sd, h + '.image_pool.1.conv', h + '.image_pool.1.bn', _cap(h + '.gap', c4.mean((2, 3), keepdim=True)), train=train)
    outs = [_cap(h + '.image_pool.up', F.interpolate(pool, size=c4.shape[2:], mode='bilinear', align_corners=False)),
            _cbr(sd, h + '.aspp_modules.0.conv', h + '.aspp_modules.0.bn', c4, train=train)]
    for i, d in zip((1, 2, 3), ASPP_DILATIONS[1:]):
        outs.append(_dwsep(sd, f'{h}.aspp_modules.{i}', c4, d, train))
    features = _cbr(sd, h + '.bottleneck.conv', h + '.bottleneck.bn', torch.cat(outs, 1), 1, 1, train=train)
    c1o = _cbr(sd, h + '.c1_bottleneck.conv', h + '.c1_bottleneck.bn', c1, train=train)
    up = _cap(h + '.up', F.interpolate(features, size=c1o.shape[2:], mode='bilinear', align_corners=False))
    o = torch.cat([up, c1o], 1)
    o = _dwsep(sd, h + '.sep_bottleneck.0', o, 1, train)
    o = _dwsep(sd, h + '.sep_bottleneck.1', o, 1, train)
    if drop_mask is not None:
        o = o * drop_mask
    logits = _cap(h + '.conv_seg', F.conv2d(o, sd[h + '.conv_seg.weight'], sd[h + '.conv_seg.bias']))
    return logits, features


def aux_head_forward(sd, feats, train=True, drop_mask=None, pre=''):
    """FCNHead(num_convs=1, concat_input=False).forward (fcn_head.py:75-98)."""
    a = pre + 'auxiliary_head'
    o = _cbr(sd, a + '.convs.0.conv', a + '.convs.0.bn', feats[2], 1, 1, train=train)
    if drop_mask is not None:
        o = o * drop_mask
    return _cap(a + '.conv_seg', F.conv2d(o, sd[a + '.conv_seg.weight'], sd[a + '.conv_seg.bias']))


# ---------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------
def ce_loss(cls_score, label, weight=None, class_weight=None, loss_weight=1.0, ignore_index=255):
    """CrossEntropyLoss(avg_non_ignore=False, reduction='mean') (cross_entropy_loss.py:45-65, utils.py:60-69):
    per-pixel CE (0 at ignore) * pixel weight, mean over ALL pixels, * loss_weight."""
    cw = None if class_weight is None else cls_score.new_tensor(class_weight)
    loss = F.cross_entropy(cls_score, label, weight=cw, reduction='none', ignore_index=ignore_index)
    if weight is not None:
        loss = loss * weight.float()
    return loss_weight * loss.mean()


def accuracy(pred, target, ignore_index=255):
    """top-1 accuracy in percent over non-ignored pixels (accuracy.py:6-61)."""
    eps = torch.finfo(torch.float32).eps
    keep = (target != ignore_index) if ignore_index is not None else torch.ones_like(target, dtype=torch.bool)
    correct = (pred.argmax(1) == target) & keep
    return (correct.float().sum().reshape(1) + eps) * (100.0 / (keep.sum().item() + eps))


def head_losses(seg_logit, seg_label, seg_weight, loss_weight, class_weight=None):
    """BaseDecodeHead.losses (decode_head.py:249-283): bilinear to label size, CE, accuracy."""
    up = F.interpolate(seg_logit, size=seg_label.shape[2:], mode='bilinear', align_corners=False)
    lab = seg_label.squeeze(1).long()
    return OrderedDict(loss_ce=ce_loss(up, lab, seg_weight, class_weight, loss_weight),
                       acc_seg=accuracy(up, lab))


def segmentor_forward_train(sd, img, gt, seg_weight=None, drop_masks=(None, None), pre='',
                            loss_weights=(1.0, 0.4)):
    """EncoderDecoder.forward_train (encoder_decoder.py:166-217).
    Returns (losses{decode.*, aux.*}, feats, logits, decoded_features, aux_logits)."""
    feats = backbone_forward(sd, img, True, pre)
    logits, dec = decode_head_forward(sd, feats, True, drop_masks[0], pre)
    losses = OrderedDict()
    for k, v in head_losses(logits, gt, seg_weight, loss_weights[0]).items():
        losses['decode.' + k] = v
    aux_logits = aux_head_forward(sd, feats, True, drop_masks[1], pre)
    for k, v in head_losses(aux_logits, gt, seg_weight, loss_weights[1]).items():
        losses['aux.' + k] = v
    return losses, feats, logits, dec, aux_logits


def encode_decode(sd, img, pre='', return_feats=False):
    """EncoderDecoder.encode_decode as the teacher runs it (encoder_decoder.py:72-84, pfgst.py:247-257):
    BN in TRAIN mode (batch statistics), dropout off, no aux head; logits bilinear to image size."""
    feats = backbone_forward(sd, img, True, pre)
    logits, dec = decode_head_forward(sd, feats, True, None, pre)
    out = (F.interpolate(logits, size=img.shape[2:], mode='bilinear', align_corners=False), dec, logits)
    return out + (feats,) if return_feats else out


# ---------------------------------------------------------------------------
# test time beyond the whole-image arg-max (encoder_decoder.py:220-372), model.eval(): BN on running statistics
# ---------------------------------------------------------------------------
def eval_encode_decode(sd, img, pre=''):
    """EncoderDecoder.encode_decode under model.eval() (encoder_decoder.py:72-84): logits bilinear to the input size"""
    feats = backbone_forward(sd, img, False, pre)
    logits, _ = decode_head_forward(sd, feats, False, None, pre)
    return F.interpolate(logits, size=img.shape[2:], mode='bilinear', align_corners=False)


def slide_inference(sd, img, img_meta, test_cfg, rescale, num_classes):
    """encoder_decoder.py:220-263, line by line"""
    h_stride, w_stride = test_cfg['stride']
    h_crop, w_crop = test_cfg['crop_size']
    batch_size, _, h_img, w_img = img.size()
    h_grids = max(h_img - h_crop + h_stride - 1, 0) // h_stride + 1
    w_grids = max(w_img - w_crop + w_stride - 1, 0) // w_stride + 1
    preds = img.new_zeros((batch_size, num_classes, h_img, w_img))
    count_mat = img.new_zeros((batch_size, 1, h_img, w_img))
    for h_idx in range(h_grids):
        for w_idx in range(w_grids):
            y1 = h_idx * h_stride
            x1 = w_idx * w_stride
            y2 = min(y1 + h_crop, h_img)
            x2 = min(x1 + w_crop, w_img)
            y1 = max(y2 - h_crop, 0)
            x1 = max(x2 - w_crop, 0)
            crop_seg_logit = eval_encode_decode(sd, img[:, :, y1:y2, x1:x2])
            preds += F.pad(crop_seg_logit, (int(x1), int(preds.shape[3] - x2), int(y1), int(preds.shape[2] - y2)))
            count_mat[:, :, y1:y2, x1:x2] += 1
    assert (count_mat == 0).sum() == 0
    preds = preds / count_mat
    if rescale:
        preds = F.interpolate(preds, size=img_meta[0]['ori_shape'][:2], mode='bilinear', align_corners=False)
    return preds


def inference_probs(sd, img, img_meta, test_cfg, rescale=True, num_classes=6):
    """EncoderDecoder.inference (encoder_decoder.py:284-327): slide / whole -> softmax -> un-flip"""
    assert test_cfg['mode'] in ['slide', 'whole']
    if test_cfg['mode'] == 'slide':
        seg_logit = slide_inference(sd, img, img_meta, test_cfg, rescale, num_classes)
    else:
        seg_logit = eval_encode_decode(sd, img)
        if rescale:
            seg_logit = F.interpolate(seg_logit, size=img_meta[0]['ori_shape'][:2], mode='bilinear', align_corners=False)
    output = F.softmax(seg_logit, dim=1)
    if img_meta[0]['flip']:
        direction = img_meta[0]['flip_direction']
        for d in (direction if isinstance(direction, list) else [direction]):
            assert d in ['horizontal', 'vertical']
            output = output.flip(dims=(3,)) if d == 'horizontal' else output.flip(dims=(2,))
    return output


def aug_test(sd, imgs, img_metas, test_cfg, num_classes=6):
    """EncoderDecoder.aug_test (encoder_decoder.py:355-372): mean of the views' probabilities, arg-max"""
    seg_logit = inference_probs(sd, imgs[0], img_metas[0], test_cfg, True, num_classes)
    for i in range(1, len(imgs)):
        seg_logit += inference_probs(sd, imgs[i], img_metas[i], test_cfg, True, num_classes)
    seg_logit /= len(imgs)
    return seg_logit.argmax(dim=1), seg_logit


def parse_losses(losses):
    """BaseSegmentor._parse_losses (base.py:177-222), single process."""
    log_vars = OrderedDict((k, v.mean()) for k, v in losses.items())
    loss = sum(v for k, v in log_vars.items() if 'loss' in k)
    log = OrderedDict((k, float(v)) for k, v in log_vars.items())
    log['loss'] = float(loss)
    return loss, log


# ---------------------------------------------------------------------------
# pseudo labels + class mix
# ---------------------------------------------------------------------------
def pseudo_label(ema_logits, threshold, thre_type='all'):
    """pfgst.py:259-268: softmax->max; thre_type 'all': (#prob>=tau / numel) as a scalar weight map,
    'part': the per-pixel 0/1 confidence mask."""
    prob, lab = torch.max(torch.softmax(ema_logits.detach(), dim=1), dim=1)
    conf = prob >= threshold
    n_conf = int(conf.sum())
    if thre_type == 'part':
        return lab, conf.to(prob.dtype), n_conf
    q = n_conf / lab.numel()
    return lab, q * torch.ones_like(prob), n_conf


def class_masks(gt, rng=np.random):
    """get_class_masks (dacs_transforms.py:110-126).  Classes are drawn from torch.unique over the WHOLE
    batch (incl. 255) with the global NumPy RNG: one `choice` per image."""
    masks = []
    classes_all = torch.unique(gt)
    n = classes_all.shape[0]
    for lab in gt:
        pick = rng.choice(n, int((n + n % 2) / 2), replace=False)
        cls = classes_all[torch.as_tensor(np.asarray(pick)).long()]
        masks.append((lab.unsqueeze(0) == cls.view(-1, 1, 1, 1)).sum(0, keepdim=True)[0:1].reshape(1, 1, *lab.shape[-2:]))
    return torch.cat(masks).long()


def class_mix(masks, img, trg_img, gt, pseudo_lbl, pseudo_w):
    """one_mix (dacs_transforms.py:129-144) applied per pfgst.py:287-300:
    M*source + (1-M)*target for image, label (int64) and pixel weight (source weight 1)."""
    m = masks
    mixed_img = m * img + (1 - m) * trg_img
    mixed_lbl = m * gt.long() + (1 - m) * pseudo_lbl.unsqueeze(1)
    mixed_w = (m[:, 0] * torch.ones_like(pseudo_w) + (1 - m[:, 0]) * pseudo_w)
    return mixed_img, mixed_lbl, mixed_w


# ---------------------------------------------------------------------------
# PFGSTLoss
# ---------------------------------------------------------------------------
def pfgst_loss(logits_trg, x_ema, x_src, gt_src, mix_masks, weights, k=3, dil=2, top_k=3, downscale=0.5,
               sim_type='cosine', sigma=30.0, src_loss_type='mean_std', margin=(0.5, 0.5), detach_unfold=True,
               src_perc=None, proj=None):
    """PFGSTLoss.forward with cross_prob_type='trg', feat_level=None (pfgst_loss.py:44-234); options:
    sim_type 'cosine' | 'gaussian' (:199-208), src_loss_type 'mean_std' | 'margin' | 'margin2' (:107-131),
    detach_unfold (:151-152), top_k None = all k*k pairs (:229-231), downscale None (:54-57), src_perc = keep the hardest fraction of
    the source pairs (:98-102: the smallest positive / largest negative similarities), proj = (weight, bias) of the 1x1 proj_net
    applied to BOTH feature maps (:34-36,73-75).  Returns (losses, extras)."""
    unfold = lambda t: F.unfold(t, k, dilation=dil, padding=(k // 2) * dil)
    kk = k * k
    if downscale is not None:
        logits_trg = F.interpolate(logits_trg, scale_factor=(float(downscale), float(downscale)))
    B, C, H, W = logits_trg.shape
    x_ema = F.interpolate(x_ema, size=(H, W))              # nearest (get_sim_feat :193-194; the :56-57 resize is the same map)
    x_src = F.interpolate(x_src, size=(H, W))
    gt_ = F.interpolate(gt_src.float(), size=(H, W), mode='nearest')
    valid_src = gt_ != 255
    trg_region = F.interpolate((1 - mix_masks).float(), size=(H, W), mode='nearest') > 0.5
    all9 = unfold(trg_region.float()).view(B, kk, H, W).long().sum(1, keepdim=True) == kk

    prob = F.softmax(logits_trg, 1)
    q = unfold(prob)
    if detach_unfold:
        q = q.detach()
    q = q.view(B, C, kk, H, W)
    cross_pos = (prob.unsqueeze(2) * q).sum(1)                 # (B, kk, H, W)

    def sim_of(x):
        u = unfold(x).view(B, x.shape[1], kk, H, W)
        if sim_type == 'gaussian':
            return torch.exp(-((u - x.unsqueeze(2)) ** 2).sum(1) / sigma ** 2)
        assert sim_type == 'cosine'
        return F.cosine_similarity(u, x.unsqueeze(2), dim=1)   # (B, kk, H, W)

    if proj is not None:
        x_src, x_ema = F.conv2d(x_src, proj[0], proj[1]), F.conv2d(x_ema, proj[0], proj[1])
    ema_sim, src_sim = sim_of(x_ema), sim_of(x_src)
    nb = unfold(gt_).view(B, kk, H, W).long()
    ctr = gt_.long().expand(B, kk, H, W)
    vs = valid_src.expand(B, kk, H, W)
    pos, neg = src_sim[(nb == ctr) & vs], src_sim[(nb != ctr) & vs]
    if src_perc is not None:
        pos = pos.sort()[0][:int(pos.shape[0] * src_perc)]
        neg = neg.sort(descending=True)[0][:int(neg.shape[0] * src_perc)]

    mask = valid_src & all9
    if top_k is not None:
        _, imax = torch.topk(ema_sim, top_k + 1, dim=1)
        _, imin = torch.topk(ema_sim, top_k, dim=1, largest=False)
        loc_pos = torch.gather(ema_sim, 1, imax) * (-torch.gather(cross_pos, 1, imax))
        loc_neg = (1 - torch.gather(ema_sim, 1, imin)) * (-torch.gather(1 - cross_pos, 1, imin))
    else:
        loc_pos = ema_sim * (-cross_pos)
        loc_neg = (1 - ema_sim) * (-(1 - cross_pos))
    if mask.sum() > 1:
        l_pos = loc_pos[mask.expand_as(loc_pos)].mean()
        l_neg = loc_neg[mask.expand_as(loc_neg)].mean()
    else:
        l_pos, l_neg = torch.zeros(1), torch.zeros(1)
    w = weights
    if src_loss_type == 'mean_std':
        losses = OrderedDict(
            loss_src_pos_mean=-pos.mean() * w['src_pos'], loss_src_neg_mean=neg.mean() * w['src_neg'],
            loss_src_pos_std=pos.std() * w['src_pos_std'], loss_src_neg_std=neg.std() * w['src_neg_std'])
    else:
        assert src_loss_type in ('margin', 'margin2')
        e = 1 if src_loss_type == 'margin' else 2
        losses = OrderedDict(loss_src_pos=(F.relu(margin[0] - pos) ** e).mean() * w['src_pos'],
                             loss_src_neg=(F.relu(neg - margin[1]) ** e).mean() * w['src_neg'])
    losses.update(loss_sim_pos=l_pos * w['sim_pos'], loss_sim_neg=l_neg * w['sim_neg'])
    extras = dict(density=1 - ema_sim.mean(1, keepdim=True).detach(), trg_mask=all9, ema_sim=ema_sim.detach(),
                  src_sim=src_sim.detach(), cross_pos=cross_pos.detach(), mask=mask)
    return losses, extras


# ---------------------------------------------------------------------------
# EMA + the full train step
# ---------------------------------------------------------------------------
def ema_update(teacher, student, it, alpha):
    """pfgst.py:105-127: it==0 copy, else lerp with alpha_t=min(1-1/(it+1), alpha); PARAMETERS only."""
    keys = param_keys(student)
    if it == 0:
        for k in keys:
            teacher[k].copy_(student[k].detach())
    else:
        a = min(1 - 1 / (it + 1), alpha)
        for k in keys:
            teacher[k].copy_(a * teacher[k] + (1 - a) * student[k].detach())


DEFAULT_LOSS_W = {'src_pos': 0.1, 'src_neg': 0.1, 'sim_pos': 0.1, 'sim_neg': 0.1,
                  'src_pos_std': 0.1, 'src_neg_std': 0.1}


class OraclePFGST:
    """PFGST.train_step on CPU (pfgst.py:129-166,179-356) with torch autograd + torch.optim.AdamW."""

    def __init__(self, student_sd, alpha=0.999, pseudo_threshold=0.98, trg_loss_weight=1.0,
                 aux_weights=None, lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01, teacher_sd=None,
                 blur=False, downscale=0.5, thre_type='all', loss_opts=None, feat_level=None, ignore_top=0, ignore_bottom=0,
                 proj=None, apply_no_mix=False):
        self.student = OrderedDict((k, v.clone()) for k, v in student_sd.items())
        self.teacher = OrderedDict((k, v.clone()) for k, v in (teacher_sd or student_sd).items())
        self.pkeys = param_keys(self.student)
        for k in self.pkeys:
            self.student[k].requires_grad_(True)
        # PFGSTLoss.proj_net: (weight, bias) of the trainable 1x1 projection; optimised with the student (it is a sub-module of PFGST)
        self.proj = None if proj is None else tuple(t.clone().requires_grad_(True) for t in proj)
        self.opt = torch.optim.AdamW([self.student[k] for k in self.pkeys] + list(self.proj or ()), lr=lr, betas=betas,
                                     weight_decay=weight_decay)
        self.alpha, self.tau, self.trg_w = alpha, pseudo_threshold, trg_loss_weight
        self.aux_w = aux_weights or DEFAULT_LOSS_W
        self.blur = blur
        self.downscale = downscale
        self.loss_opts = dict(loss_opts or {})       # PFGSTLoss option variants (sim_type, sigma, src_loss_type, margin, detach_unfold, top_k)
        # feat_level k: use_decoded_feats=False + PFGSTLoss(feat_level=k) -- backbone feature map k instead of the decoded
        # features (pfgst.py:229-231,255-257; pfgst_loss.py:50-51)
        self.feat_level = feat_level
        self.thre_type = thre_type
        self.ignore_top, self.ignore_bottom = ignore_top, ignore_bottom      # pseudo_weight_ignore_top / _bottom (pfgst.py:273-276)
        self.apply_no_mix = apply_no_mix             # pfgst.py:283-289: masks zeroed after they were drawn, the un-augmented target image
        self.local_iter = 0

    def train_step(self, batch, masks=None, drop_masks=None, return_extras=False, pseudo_override=None):
        img, gt = batch['img'], batch['gt_semantic_seg']
        trg, trg_aug = batch['target_img'], batch['target_img_strong_aug']
        self.opt.zero_grad()
        with torch.no_grad():
            ema_update(self.teacher, self.student, self.local_iter, self.alpha)
        random.uniform(0, 1)                      # colour-jitter draw (pfgst.py:215)
        if self.blur:
            random.uniform(0, 1)
        dm = drop_masks or {}
        log = OrderedDict()
        losses, feats, src_logits, src_dec, _ = segmentor_forward_train(
            self.student, img, gt, None, dm.get('src', (None, None)))
        clean_loss, lv = parse_losses(losses)
        lv.pop('loss'); log.update(lv)
        with torch.no_grad():
            ema_logits, ema_dec, ema_low, ema_feats = encode_decode(self.teacher, trg, return_feats=True)
        pl, pw, n_conf = pseudo_label(ema_logits, self.tau, self.thre_type)
        if pseudo_override is not None:        # (label map, #confident) injected by parity tests
            pl, n_conf = pseudo_override
            if self.thre_type == 'all':
                pw = (n_conf / pl.numel()) * torch.ones(pl.shape, dtype=ema_logits.dtype)
        if self.ignore_top > 0:
            pw[:, :self.ignore_top, :] = 0
        if self.ignore_bottom > 0:
            pw[:, -self.ignore_bottom:, :] = 0
        if masks is None:
            masks = class_masks(gt)
        if self.apply_no_mix:
            masks = [torch.zeros_like(m) for m in masks] if isinstance(masks, (list, tuple)) else torch.zeros_like(masks)
        mixed_img, mixed_lbl, mixed_w = class_mix(masks, img, trg if self.apply_no_mix else trg_aug, gt, pl, pw)
        mlosses, _, mix_logits, _, _ = segmentor_forward_train(
            self.student, mixed_img, mixed_lbl, mixed_w, dm.get('mix', (None, None)))
        mix_loss, lv = parse_losses(OrderedDict(('mix.' + k, v) for k, v in mlosses.items()))
        lv.pop('loss'); log.update(lv)
        x_ema, x_src = (ema_dec, src_dec) if self.feat_level is None else (ema_feats[self.feat_level], feats[self.feat_level])
        aux, extras = pfgst_loss(mix_logits, x_ema, x_src, gt, masks, self.aux_w, downscale=self.downscale, proj=self.proj,
                                 **self.loss_opts)
        aux_loss, lv = parse_losses(aux)
        lv.pop('loss'); log.update(lv)
        total = clean_loss + self.trg_w * mix_loss + aux_loss
        total.backward()
        self.opt.step()
        self.local_iter += 1
        if return_extras:
            extras.update(pseudo_label=pl, n_conf=n_conf, masks=masks, mixed_img=mixed_img, mixed_lbl=mixed_lbl,
                          mixed_w=mixed_w, src_logits=src_logits.detach(), mix_logits=mix_logits.detach(),
                          ema_logits=ema_logits, ema_logits_low=ema_low, ema_dec=ema_dec, src_dec=src_dec.detach(),
                          grads=OrderedDict((k, self.student[k].grad.clone()) for k in self.pkeys),
                          proj_grads=None if self.proj is None else tuple(t.grad.clone() for t in self.proj))
            return log, extras
        return log
