"""PFGST self-training wrapper and PFGSTLoss on the HIP path, behind the reference's plugin API:
`UDA.build(cfg.uda)` -> object with `train_step(data_batch, optimizer) -> {log_vars, num_samples, states}`.

Reference: rsiseg/models/uda/pfgst.py:53-368, uda_decorator.py:13-103, losses/pfgst_loss.py:12-234,
utils/dacs_transforms.py:12-144, segmentors/base.py:177-222.

Differences from the reference that are deliberate:
  * the EMA teacher's parameters are never part of the optimizer / gradient all-reduce;
  * `local_iter` is part of state_dict (the reference loses it on resume);
  * ~20 host syncs per step become one D2H read of a packed scalar vector (+ the 256-flag label
    presence read that the NumPy class choice needs);
  * labels travel as uint8 on the device, int64 only at the API boundary."""
import random
from collections import OrderedDict
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn

from . import dist as pdist
from . import hip_ops as ops
from . import layers
from .engine import ParamArena, Tape
from .registry import LOSSES, UDA, build_loss, build_segmentor


@LOSSES.register_module()
class PFGSTLoss(nn.Module):
    def __init__(self, top_k, dilation, kernel_size, weights, sigma=30, mean_sim=0.6, feat_level=2, sim_type='gaussian',
                 num_bins=100, apply_ignore=False, src_perc=None, proj_net_cfg=None, src_loss_type='mean_std',
                 margin=(0.5, 0.5), detach_unfold=False, cross_prob_type='trg', downscale=None):
        super().__init__()
        # Implemented: the shipped options plus the variants reachable from the same configs: sim_type
        # 'cosine' | 'gaussian' (sigma), src_loss_type 'mean_std' | 'margin' | 'margin2' (margin), detach_unfold True | False,
        # top_k 1..4 | None, downscale 0.5 | 1 | None, feat_level None | 0..3 (a backbone feature map instead of the decoded
        # features, with PFGST(use_decoded_feats=False)), src_perc (the hardest fraction of the source pairs), proj_net_cfg (a
        # trainable 1x1 projection of both feature maps).  Everything else (cross_prob_type='ema', kernel_size != 3) fails loudly.
        bad = dict(kernel_size=kernel_size != 3, sim_type=sim_type not in ('cosine', 'gaussian'),
                   feat_level=feat_level is not None and feat_level not in (0, 1, 2, 3),
                   src_perc=src_perc is not None and not 0.0 <= src_perc <= 1.0,
                   proj_net=proj_net_cfg is not None and not {'in_channels', 'out_channels'} <= set(proj_net_cfg),
                   src_loss_type=src_loss_type not in ('mean_std', 'margin', 'margin2'), cross_prob_type=cross_prob_type != 'trg',
                   downscale=downscale not in (None, 0.5, 1, 1.0), top_k=top_k is not None and not (1 <= top_k <= 4),
                   weights=not isinstance(weights, dict), sigma=not sigma > 0)
        bad = [k for k, v in bad.items() if v]
        if bad:
            raise NotImplementedError(f'PFGSTLoss options outside the implemented set: {bad}')
        self.top_k, self.dilation, self.weights = top_k, dilation, dict(weights)
        self.sim_type, self.sigma = sim_type, float(sigma)
        self.src_loss_type, self.margin = src_loss_type, tuple(margin)
        self.unfold_grad = not detach_unfold
        self.ds = 1 if downscale is None else int(round(1.0 / downscale))
        self.feat_level = feat_level
        self.src_perc = src_perc
        self.proj_net = None
        if proj_net_cfg is not None:                         # pfgst_loss.py:34-36: nn.Conv2d(in, out, kernel_size=1), default init
            ref_init = nn.Conv2d(proj_net_cfg['in_channels'], proj_net_cfg['out_channels'], kernel_size=1)
            self.proj_net = layers.Conv2dP(proj_net_cfg['in_channels'], proj_net_cfg['out_channels'], 1, bias=True)
            with torch.no_grad():
                self.proj_net.weight.copy_(ref_init.weight)
                self.proj_net.bias.copy_(ref_init.bias)

    def forward(self, tensors, tape=None):
        """tensors: logits_trg (Var, student logits of the mixed pass), x_ema (Var), x_src (Var),
        gt_src (uint8 [N,1,H,W]), mix_masks (uint8 [N,1,H,W]).  -> dict of 6 one-element device tensors."""
        lt, x_ema, x_src = tensors['logits_trg'], tensors['x_ema'], tensors['x_src']
        if self.feat_level is not None:                   # pfgst_loss.py:50-51
            x_ema, x_src = x_ema[self.feat_level], x_src[self.feat_level]
        if isinstance(x_src, (tuple, list)) or isinstance(x_ema, (tuple, list)):
            raise TypeError('PFGSTLoss: x_src / x_ema are feature tuples (use_decoded_feats=False) but feat_level is None')
        gt8, mm8 = tensors['gt_src'], tensors['mix_masks']
        d, w = self.dilation, self.weights
        n, c, h, wd = lt.data.shape
        H, W = h // self.ds, wd // self.ds
        # F.interpolate(x, size=(H, W)) nearest (pfgst_loss.py:57-58): identity for downscale 0.5; for downscale 1 the
        # 1/8 features are replicated u x u onto the 1/4 grid, where a dilation-d neighbourhood is EXACTLY the
        # dilation-d/u neighbourhood of the source grid -- so the similarity is computed at the source resolution
        # and only the tiny 9-channel map is replicated (its adjoint is a u x u sum).
        hf, wf = x_src.data.shape[-2:]
        if x_ema.data.shape[-2:] != (hf, wf) or H % hf != 0 or H // hf != W // wf:
            raise NotImplementedError(f'PFGSTLoss: feature grid {hf}x{wf} does not divide the logit grid {H}x{W}')
        u = H // hf
        if d % u != 0:
            raise NotImplementedError(f'PFGSTLoss: dilation {d} not divisible by the feature up-sampling factor {u}')
        fd = d // u
        proj, raw_src, raw_ema = self.proj_net, x_src, x_ema
        if proj is not None:
            # pfgst_loss.py:73-75: the same trainable 1x1 convolution on both maps (it commutes with the nearest resize above)
            for p in proj.parameters():
                if p.grad is None:
                    p.grad = torch.zeros_like(p.data)
            proj.repack(need_dgrad=True)
            from .engine import Var
            x_src = Var(proj.fprop(raw_src.data, bias=proj.bias.data), True)
            x_ema = Var(proj.fprop(raw_ema.data, bias=proj.bias.data), False)
        ema_sim, ema_norm = ops.sim_map(x_ema.data, fd, self.sim_type, self.sigma)
        src_sim_f, src_norm = ops.sim_map(x_src.data, fd, self.sim_type, self.sigma)
        src_sim = src_sim_f
        ema_sim_lowres = ema_sim
        if u > 1:
            ema_sim, src_sim = ops.upsample_nearest(ema_sim, u), ops.upsample_nearest(src_sim_f, u)
        l4, gsim = ops.src_sim_losses(src_sim, gt8, d, w['src_pos'], w['src_neg'], w.get('src_pos_std', 0.0), w.get('src_neg_std', 0.0),
                                      self.src_loss_type, self.margin, src_perc=self.src_perc)
        prob = ops.softmax_down(lt.data, self.ds)
        valid, all9, cnt = ops.trg_valid_mask(gt8, mm8, (H, W), d)
        res = ops.sim_topk_loss(ema_sim, prob, valid, cnt, d, self.top_k, w['sim_pos'], w['sim_neg'],
                                want_sim_grad=proj is not None and tape is not None)
        l2, gP = res[0], res[1]
        gS = res[2] if len(res) > 2 else None
        ema_sim_f = ema_sim_lowres if u > 1 else ema_sim
        if tape is not None:
            def bwd():
                buf, acc = x_src.grad_target()
                g_f = gsim if u == 1 else ops.upsample_nearest_bwd(gsim, u)
                ops.sim_map_bwd(x_src.data, src_sim_f, src_norm, g_f, fd, out=buf, accumulate=acc, sim_type=self.sim_type,
                                sigma=self.sigma)
                if proj is not None:
                    # the projection's weights collect gradient from BOTH branches: the teacher-side similarity is a function of
                    # them too (the reference's x_ema carries no graph, proj_net(x_ema) does)
                    gs_f = gS if u == 1 else ops.upsample_nearest_bwd(gS, u)
                    g_ema = ops.sim_map_bwd(x_ema.data, ema_sim_f, ema_norm, gs_f, fd, sim_type=self.sim_type, sigma=self.sigma)
                    ops.conv_wgrad_(proj.weight.grad, raw_ema.data, g_ema, 1)
                    ops.bias_grad_(proj.bias.grad, g_ema)
                    layers.conv_backward(raw_src, proj, x_src.grad)
                buf, acc = lt.grad_target()
                if not acc:
                    ops.fill_(buf, 0.0)
                ops.cross_prob_bwd_(buf, prob, gP, d, self.ds, self.unfold_grad)
            # tagged like every closure of the network, so a Tape observer (the per-link backward test) sees the link as wired:
            # it writes dL/dx_src (first writer of that buffer in the sweep) and dL/dlogits_trg, into which the mixed pass's CE
            # backward accumulates afterwards
            tape.record(bwd, tag=dict(op='pfgst_loss', out=None, module=self, logits_trg=lt, x_src=raw_src, x_ema=raw_ema,
                                      gt_src=gt8, mix_masks=mm8))
        if self.src_loss_type == 'mean_std':
            out = OrderedDict(loss_src_pos_mean=l4[0:1], loss_src_neg_mean=l4[1:2], loss_src_pos_std=l4[2:3], loss_src_neg_std=l4[3:4])
        else:                                           # pfgst_loss.py:116-131
            out = OrderedDict(loss_src_pos=l4[0:1], loss_src_neg=l4[1:2])
        out.update(loss_sim_pos=l2[0:1], loss_sim_neg=l2[1:2])
        if tensors.get('want_vis'):
            # pfgst_loss.py:134-137: (img_trg, 1 - mean_k sim_ema, the all-nine-neighbours-unmixed mask) -- visualisation only, built
            # with torch ops because it is off the hot path (PFGST.return_vis_states)
            out['vis|density_sim_feat'] = (tensors.get('img_trg'), 1 - ema_sim.mean(dim=1, keepdim=True), all9.bool())
        return out


def get_module(module):
    return module.module if hasattr(module, 'module') and isinstance(getattr(module, 'module'), nn.Module) else module


class UDADecorator(nn.Module):
    """Owns the student `self.model`; test-time calls delegate to it (uda_decorator.py:29-103)."""

    def __init__(self, **cfg):
        super().__init__()
        self.model = build_segmentor(deepcopy(cfg['model']))
        self.train_cfg = cfg['model']['train_cfg']
        self.test_cfg = cfg['model']['test_cfg']
        self.num_classes = cfg['model']['decode_head']['num_classes']

    def get_model(self):
        return get_module(self.model)

    def extract_feat(self, img):
        return self.get_model().extract_feat(img)

    def encode_decode(self, img, img_metas):
        return self.get_model().encode_decode(img, img_metas)

    def inference(self, img, img_meta=None, rescale=True):
        return self.get_model().inference(img, img_meta, rescale)

    def simple_test(self, img, img_meta=None, rescale=True):
        return self.get_model().simple_test(img, img_meta, rescale)

    def aug_test(self, imgs, img_metas, rescale=True):
        return self.get_model().aug_test(imgs, img_metas, rescale)


@UDA.register_module()
class PFGST(UDADecorator):
    def __init__(self, **cfg):
        super().__init__(**cfg)
        self.local_iter = 0
        self.max_iters = cfg['max_iters']
        self.alpha = cfg['alpha']
        self.pseudo_threshold = cfg['pseudo_threshold']
        self.psweight_ignore_top = cfg['pseudo_weight_ignore_top']
        self.psweight_ignore_bottom = cfg['pseudo_weight_ignore_bottom']
        self.fdist_lambda = cfg['imnet_feature_dist_lambda']
        self.mix = cfg['mix']
        self.blur = cfg['blur']
        self.color_jitter_s = cfg['color_jitter_strength']
        self.color_jitter_p = cfg['color_jitter_probability']
        self.print_grad_magnitude = cfg['print_grad_magnitude']
        self.trg_loss_weight = cfg.get('trg_loss_weight', 1.)
        self.use_decoded_feats = cfg.get('use_decoded_feats', False)
        self.thre_type = cfg.get('thre_type', 'all')
        self.strong_aug_denorm_type = cfg.get('strong_aug_denorm_type', 'mean_std')
        self.apply_no_mix = cfg.get('apply_no_mix', False)
        assert self.mix == 'class'
        bad = dict(fdist=self.fdist_lambda > 0, thre_type=self.thre_type not in ('all', 'part'),
                   ps_rows=self.psweight_ignore_top < 0 or self.psweight_ignore_bottom < 0,
                   print_grad=self.print_grad_magnitude)
        bad = [k for k, v in bad.items() if v]
        if bad:
            raise NotImplementedError(f'PFGST options outside the shipped PFST configs: {bad}')
        self.ema_model = build_segmentor(deepcopy(cfg['model']))
        for p in self.ema_model.parameters():
            p.requires_grad = False          # the teacher is never optimised / all-reduced
        for m in self.ema_model.modules():
            if hasattr(m, 'dropout_enabled'):
                m.dropout_enabled = False    # pfgst.py:247-251
        aux_losses = cfg.get('aux_losses', None)
        self.apply_aux = aux_losses is not None
        if self.apply_aux:
            if not isinstance(aux_losses, (list, tuple)):
                aux_losses = [aux_losses]
            self.aux_losses = nn.ModuleList([build_loss(dict(l)) for l in aux_losses])
        # the reference returns three `vis|*` tuples of live tensors from every step (pfgst.py:335,346-352) for its plotting hooks;
        # they cost argmax / interpolate passes over full-resolution tensors, so they are built only when asked for
        self.return_vis_states = False
        self._test = None                    # parity-test hooks (see the properties below); no config can set them
        self._student_arena = self._teacher_arena = None

    # ------------------------------------------------------------------ test hooks (never set by configs / the trainer)
    class _TestHooks:
        """debug: dict that captures intermediates; injected_mix_classes: the class choice; injected_pseudo: (uint8 label map,
        confident-pixel count) -- decouples the student-gradient check from 1-ulp arg-max ties in the teacher logits;
        tape_observer: engine.Tape(observer=...) for the step's single backward sweep (tests/test_layer_backward_gpu.py)"""
        __slots__ = ('debug', 'injected_mix_classes', 'injected_pseudo', 'tape_observer')

        def __init__(self):
            self.debug = self.injected_mix_classes = self.injected_pseudo = self.tape_observer = None

    def _hook(self, name, value=None, set_=False):
        if set_:
            if self._test is None:
                if value is None:
                    return None
                self._test = PFGST._TestHooks()
            setattr(self._test, name, value)
            if all(getattr(self._test, k) is None for k in PFGST._TestHooks.__slots__):
                self._test = None
            return None
        return None if self._test is None else getattr(self._test, name)

    debug = property(lambda self: self._hook('debug'), lambda self, v: self._hook('debug', v, True))
    injected_mix_classes = property(lambda self: self._hook('injected_mix_classes'), lambda self, v: self._hook('injected_mix_classes', v, True))
    injected_pseudo = property(lambda self: self._hook('injected_pseudo'), lambda self, v: self._hook('injected_pseudo', v, True))
    tape_observer = property(lambda self: self._hook('tape_observer'), lambda self, v: self._hook('tape_observer', v, True))

    # ------------------------------------------------------------------ state
    def get_extra_state(self):
        return {'local_iter': self.local_iter}

    def set_extra_state(self, state):
        self.local_iter = int(state.get('local_iter', 0))

    def init_weights(self):
        self.model.init_weights()
        self.ema_model.init_weights()

    def get_ema_model(self):
        return get_module(self.ema_model)

    def _ensure_arenas(self, device):
        a = self._student_arena
        if a is not None and a.data.device == device:
            return
        model, ema = self.get_model(), self.get_ema_model()
        model.to(device), ema.to(device)
        for aux in (self.aux_losses if self.apply_aux else []):
            aux.to(device)
        self._student_arena = ParamArena(list(model.named_parameters()), device, with_grad=True)
        self._teacher_arena = ParamArena(list(ema.named_parameters()), device, with_grad=False)
        assert self._student_arena.numel == self._teacher_arena.numel

    @property
    def student_arena(self):
        return self._student_arena

    # ------------------------------------------------------------------ EMA (pfgst.py:105-127)
    def _init_ema_weights(self):
        self._teacher_arena.data.copy_(self._student_arena.data)

    def _update_ema(self, it):
        alpha_teacher = min(1 - 1 / (it + 1), self.alpha)
        ops.ema_update_(self._teacher_arena.data, self._student_arena.data, alpha_teacher)

    # ------------------------------------------------------------------ class mix (dacs_transforms.py:110-126)
    def _choose_mix_classes(self, presence, batch_size):
        """`torch.unique` over the whole batch + one NumPy `choice` per image (global RNG stream as the reference)."""
        if self._test is not None and self._test.injected_mix_classes is not None:
            return self._test.injected_mix_classes
        classes = np.nonzero(presence)[0]
        n = classes.shape[0]
        k = int((n + n % 2) / 2)
        out = np.full((batch_size, max(k, 1)), -1, dtype=np.int32)
        for i in range(batch_size):
            out[i, :k] = classes[np.random.choice(n, k, replace=False)]
        return out

    # ------------------------------------------------------------------ the hot path
    def train_step(self, data_batch, optimizer, **kwargs):
        optimizer.zero_grad()
        # optimizer.step() is queued by forward_train right before it blocks on the step's log values (same position in stream order as
        # after it; a batch with labels outside [0, C) therefore raises AFTER its update has been applied -- the run is over either way)
        self._before_read = optimizer.step
        try:
            log_vars, vis_states = self(**data_batch)
        finally:
            pending, self._before_read = self._before_read, None
        if pending is not None:                  # forward_train never reached its read (an exception): nothing was stepped
            raise RuntimeError('PFGST.train_step: forward_train returned without reading its log values')
        log_vars.pop('loss', None)
        return dict(log_vars=log_vars, num_samples=len(data_batch['img_metas']), states=vis_states)

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.get_model().forward_test(img, img_metas, **kwargs)

    def forward_train(self, img, img_metas, gt_semantic_seg, target_img, target_img_metas, target_img_strong_aug):
        if not img.is_cuda:
            raise RuntimeError('PFGST.forward_train needs CUDA(HIP) tensors: pfst_amd has no CPU path')
        dev = img.device
        self._ensure_arenas(dev)
        model, ema = self.get_model(), self.get_ema_model()
        batch_size = img.shape[0]
        S_hw = img.shape[-2:]
        hooks = self._test                   # None in production: the parity tests' capture / injection points hang off ONE object
        dbg = hooks.debug if hooks is not None else None

        # the optimizer's zero_grad() may have detached .grad views (set_to_none): re-attach + zero the arena
        arena = self._student_arena
        # (the expected addresses are cached: building 161 views per step to compare pointers cost ~1 ms of host time at the step boundary,
        # where the device has nothing queued -- tools/host_profile.py, tools/gap_analysis.py)
        gp = getattr(self, '_grad_ptrs', None)
        if gp is None or gp[0] != arena.grad.data_ptr():
            base = arena.grad.data_ptr()
            gp = self._grad_ptrs = (base, [(name, p, base + 4 * arena.offsets[name]) for name, p in model.named_parameters()])
        for name, p, ptr in gp[1]:
            g = p.grad
            if g is None or g.data_ptr() != ptr:
                p.grad = arena.view(arena.grad, name)
        arena.zero_grad()

        if self.local_iter == 0:
            self._init_ema_weights()
        else:
            self._update_ema(self.local_iter)
        # Step boundary: the previous step ended with a blocking read, so the device idles until this step's first long kernels are
        # queued.  The teacher's weight images first, then its forward pass (forked: ~35 ms of device work queued in ~10 ms of host
        # time), and the student's weight images behind that -- their ~1.3 ms of host-side packing then runs while the device is busy.
        fork = layers.FORK_TEACHER
        ema.repack_weights(need_dgrad=False)
        late_pack = fork and layers.STEP_BOUNDARY_OVERLAP
        if not late_pack:
            model.repack_weights(need_dgrad=True)

        # strong-augmentation parameters: same Python-RNG draws as the reference (pfgst.py:212-222)
        jitter_draw = random.uniform(0, 1)
        blur_draw = random.uniform(0, 1) if self.blur else 0
        if img.shape[1] == 3 and (jitter_draw > self.color_jitter_p or blur_draw > 0.5):
            from .strong_aug import apply_strong_aug
        else:
            apply_strong_aug = None

        gt8 = ops.to_u8(gt_semantic_seg.contiguous())
        presence = ops.label_presence(gt8)
        if getattr(self, '_presence_host', None) is None:
            self._presence_host = torch.empty(256, dtype=torch.int32, pin_memory=True)
        presence_host = self._presence_host
        presence_host.copy_(presence, non_blocking=True)
        presence_evt = torch.cuda.Event()
        presence_evt.record()

        tape = Tape(hooks.tape_observer if hooks is not None else None)
        scalars = OrderedDict()

        # ---- teacher on target, optionally forked onto a second stream (layers.FORK_TEACHER): it is independent of the
        # student's source pass, so its HBM-bound kernels can run beside the other pass's MFMA-bound ones.  The host order of
        # RNG draws is unchanged (the teacher draws none: dropout is off)
        if fork:
            main_s = torch.cuda.current_stream()
            side_s = layers.teacher_stream()
            side_s.wait_stream(main_s)
            with torch.cuda.stream(side_s):
                ema_logits, ema_states = ema.encode_decode(target_img.contiguous(), target_img_metas)
            ema_dec = ema_states['decoded_features'] if self.use_decoded_feats else ema_states['feats']
            if late_pack:
                model.repack_weights(need_dgrad=True)

        # ---- student on source.  Data-parallel runs: this pass is recorded first, so its closures run LAST in the single backward
        # sweep -- its marker closures tell the reducer which tail of the gradient arena is final (dist.GradReducer)
        reducer = grad_ready = None
        step_stats = pdist.step_stats_begin()
        if pdist.is_distributed() and pdist.OVERLAP_ALLREDUCE:
            reducer = pdist.GradReducer(arena.grad)
            reducer.stats = step_stats
            cuts = {'heads': arena.offsets[next(n for n in arena.names if not n.startswith('backbone.'))]}
            for n in arena.names:
                stage = n.split('.')[1] if n.startswith('backbone.layer') else None
                if stage and stage not in cuts:
                    cuts[stage] = arena.offsets[n]
            grad_ready = lambda stage: reducer.ready(cuts[stage])
        clean = model.forward_train(img.contiguous(), img_metas, gt8, None, return_feats=True, return_logits=True,
                                    return_decoded_feats=self.use_decoded_feats, tape=tape, grad_ready=grad_ready)
        src_dec = clean.pop('features')                       # pfgst.py:229-231: the backbone feature tuple, or ...
        if self.use_decoded_feats:
            src_dec = clean.pop('decoded_features')           # ... the decode head's 512-channel map (all shipped configs)
        src_logits = clean.pop('logits')
        scalars.update(clean)

        # ---- teacher on target -> pseudo labels (fused upsample + softmax + argmax + threshold count)
        if fork:
            main_s.wait_stream(side_s)
            for t in (ema_logits,) + (tuple(ema_dec) if isinstance(ema_dec, (tuple, list)) else (ema_dec,)):
                t.data.record_stream(main_s)
        else:
            ema_logits, ema_states = ema.encode_decode(target_img.contiguous(), target_img_metas)
            ema_dec = ema_states['decoded_features'] if self.use_decoded_feats else ema_states['feats']
        part = self.thre_type == 'part'
        res = ops.pseudo_label(ema_logits.data, S_hw, self.pseudo_threshold, want_i64=dbg is not None, want_conf=part)
        pl64, pl8, conf_count = res[:3]
        trg_weight = res[3] if part else None        # per-pixel 0/1 weights instead of the scalar fraction q
        if hooks is not None and hooks.injected_pseudo is not None:
            if dbg is not None:
                dbg['own_pseudo_label'], dbg['own_conf_count'] = pl64, conf_count
            pl8, conf_count = hooks.injected_pseudo
            pl64 = ops.to_i64(pl8) if dbg is not None else None
        if self.psweight_ignore_top > 0 or self.psweight_ignore_bottom > 0:
            # pfgst.py:273-276: no trust in the pseudo labels of the top / bottom rows.  The scalar q becomes a per-pixel map
            # (q exactly as the reference forms it: python-float count / size stored to float32) with those rows zeroed.
            if trg_weight is None:
                q = (conf_count.double() / float(pl8.numel())).float()
                trg_weight = q.expand(pl8.shape).contiguous()
            if self.psweight_ignore_top > 0:
                trg_weight[:, :self.psweight_ignore_top, :] = 0
            if self.psweight_ignore_bottom > 0:
                trg_weight[:, -self.psweight_ignore_bottom:, :] = 0

        # ---- class mix
        presence_evt.synchronize()
        classes = self._choose_mix_classes(presence_host.numpy(), batch_size)
        classes_dev = ops.h2d_small(torch.from_numpy(np.ascontiguousarray(classes)), dev, 'mix_classes')
        mix_masks = ops.class_mask(gt8, classes_dev)
        if self.apply_no_mix:                          # pfgst.py:283-289: the classes were drawn (same RNG stream), then nothing is pasted
            mix_masks.zero_()
        mixed_img, mixed_lbl8, mixed_lbl64, mixed_w = ops.class_mix(
            img.contiguous(), (target_img if self.apply_no_mix else target_img_strong_aug).contiguous(), gt8, pl8, mix_masks, conf_count,
            want_i64=self.return_vis_states or dbg is not None, trg_weight=trg_weight)
        if apply_strong_aug is not None:
            mixed_img = apply_strong_aug(mixed_img, img_metas, jitter_draw, self.color_jitter_p, self.color_jitter_s,
                                         blur_draw, self.strong_aug_denorm_type)

        # ---- student on mixed
        mix = model.forward_train(mixed_img, img_metas, mixed_lbl8, mixed_w, return_feats=True, return_logits=True,
                                  tape=tape, grad_scale=self.trg_loss_weight)
        mix.pop('features')
        mixed_logits = mix.pop('logits')
        scalars.update({'mix.' + k: v for k, v in mix.items()})

        # ---- auxiliary pseudo-feature losses
        vis_states = {}
        if self.apply_aux:
            tensors = dict(gt_src=gt8, x_src=src_dec, x_ema=ema_dec, logits_trg=mixed_logits, mix_masks=mix_masks,
                           img_trg=mixed_img, want_vis=self.return_vis_states)
            for loss_module in self.aux_losses:
                aux = loss_module(tensors, tape=tape)
                vis = {k: v for k, v in aux.items() if k.startswith('vis|')}
                for k in vis:
                    aux.pop(k)
                scalars.update(aux)
                vis_states.update(vis)

        # ---- the step's log values: every one of them is a FORWARD result (losses, accuracies, PFGSTLoss terms, the CE kernels' bad-label
        # counts), so their packed copy to the host is queued HERE, in front of the backward sweep (layers.EARLY_LOG_READ): the blocking read at
        # the end of the step then waits for the forward passes only, the host returns while the device still has the whole backward sweep and
        # the optimizer step queued, and the next step's host-side start (EMA, weight images, label conversion, the first launches of the teacher
        # pass: 2-3 ms during which the device used to idle) runs under it.  The reference reads its log values at the same point
        # (`_parse_losses` before `loss.backward()`, pfgst.py:340-344)
        def start_log_read():
            names = list(scalars.keys())
            packed = torch.cat([scalars[k].reshape(1) for k in names])
            host = getattr(self, '_log_host', None)
            if host is None or host.numel() < packed.numel():
                host = self._log_host = torch.empty(max(64, packed.numel()), dtype=torch.float32, pin_memory=True)
            if pdist.is_distributed():
                if self.local_iter == 0:
                    pdist.check_same_keys(names)
                # the mean over ranks of the log vector (what `_parse_losses` does key by key, segmentors/base.py:178-222) and its copy run on a
                # stream of their own: the main stream never waits for this collective, so a rank whose forward passes ran late does not hold
                # the others' backward sweeps; every rank issues it at the same point of its program, ahead of its gradient buckets
                main = torch.cuda.current_stream()
                ls = getattr(self, '_log_stream', None)
                if ls is None:
                    ls = self._log_stream = torch.cuda.Stream()
                ls.wait_stream(main)
                with torch.cuda.stream(ls):
                    red = pdist.reduce_log_vector(packed)
                    host[:red.numel()].copy_(red, non_blocking=True)
                    evt = torch.cuda.Event()
                    evt.record()
                packed.record_stream(ls)
                return names, host, packed.numel(), evt
            host[:packed.numel()].copy_(packed, non_blocking=True)
            evt = torch.cuda.Event()
            evt.record()
            return names, host, packed.numel(), evt

        log_read = start_log_read() if layers.EARLY_LOG_READ else None

        # ---- one backward over both student graphs + aux (pfgst.py:344)
        tape.backward()
        layers.join_side_stream()

        if dbg is not None:
            dbg.update(pseudo_label=pl64, conf_count=conf_count, mix_masks=mix_masks, mixed_img=mixed_img,
                       mixed_lbl=mixed_lbl64, mixed_w=mixed_w, src_logits=src_logits.data, mix_logits=mixed_logits.data,
                       ema_logits=ema_logits.data, classes=classes)
            if self.use_decoded_feats:
                dbg.update(ema_dec=ema_dec.data, src_dec=src_dec.data)

        # ---- gradient all-reduce (student only)
        if pdist.is_distributed():
            if reducer is not None:
                reducer.finish()                       # the tail buckets have been in flight since the source pass's backward
            else:
                if step_stats is not None:
                    step_stats['bucket_elems'].append(arena.grad.numel())
                    if arena.grad.is_cuda:
                        ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        ev[0].record()
                pdist.allreduce_mean_(arena.grad)      # student gradients only; the teacher stays rank-local
                if step_stats is not None and arena.grad.is_cuda:
                    ev[1].record()
                    step_stats['exposed_allreduce'] = ev
            for loss_module in (self.aux_losses if self.apply_aux else []):
                for p in loss_module.parameters():     # trainable parameters of an auxiliary loss (PFGSTLoss.proj_net)
                    if p.grad is not None:
                        pdist.allreduce_mean_(p.grad.view(-1))
        if step_stats is not None:
            import time
            t_read = time.perf_counter()
        # the step's single blocking read: an asynchronous copy into pinned memory + an event (queued in front of the backward sweep, see
        # above; layers.EARLY_LOG_READ off: here, behind it), so that the optimizer step (train_step hands it over as `_before_read`) is
        # queued before the host waits, unpacks the values and enters the next step
        names, host, n_log, read_evt = log_read if log_read is not None else start_log_read()
        before_read, self._before_read = getattr(self, '_before_read', None), None
        if before_read is not None and layers.STEP_BOUNDARY_OVERLAP:
            before_read()
        read_evt.synchronize()
        if before_read is not None and not layers.STEP_BOUNDARY_OVERLAP:
            before_read()
        vals = host[:n_log].tolist()
        if step_stats is not None:
            step_stats['host_read_s'] = time.perf_counter() - t_read
        log_vars = OrderedDict(zip(names, vals))
        for k in [k for k in log_vars if k.rsplit('.', 1)[-1].startswith('_')]:
            v = log_vars.pop(k)                 # not a log value: the CE kernels' count of labels outside [0, C) / ignore_index
            if k.endswith('_bad_labels') and v > 0:
                raise ValueError(f'{int(v)} label values outside [0, {self.num_classes}) other than ignore_index reached the cross-entropy '
                                 f'({k}); F.cross_entropy raises on them in the reference -- check reduce_zero_label / the label maps')
        log_vars['loss'] = sum(v for k, v in log_vars.items() if 'loss' in k)

        if self.return_vis_states:
            # pfgst.py:346-352, the reference's tuple layouts
            arg = lambda lg: lg.max(dim=1)[1].unsqueeze(1)
            mix_lbl = mixed_lbl64 if mixed_lbl64 is not None else ops.to_i64(mixed_lbl8)
            vis_mix = torch.where(mixed_w.unsqueeze(1) > 0.0, mix_lbl, torch.full_like(mix_lbl, 255))
            vis_states['vis|seg_mask_src'] = (img, gt_semantic_seg, arg(src_logits.data))
            vis_states['vis|seg_mask_mix'] = (mixed_img, vis_mix, arg(mixed_logits.data).float())
        self.local_iter += 1
        return log_vars, vis_states
