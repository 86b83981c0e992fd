#!/usr/bin/env python3
"""Evaluate a checkpoint on `cfg.data.test` (or `val`) -- the flag surface of the reference's tools/test.py:23-120,225-300 that the
PFST workflow uses: whole-tile inference of the SEGMENTOR (the student of a PFGST checkpoint), mIoU / mDice / mFscore.

  python tools/test.py CONFIG CHECKPOINT --eval mIoU --revise-checkpoint-key

`--revise-checkpoint-key` strips the DDP `module.` prefix and the UDA wrapper's `model.` prefix exactly as the reference's
`load_checkpoint(revise_keys=[(r'^module\\.', ''), ('model.', '')])` (tools/test.py:237-242)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='pfst_amd test (and eval) a model')
    p.add_argument('config')
    p.add_argument('checkpoint')
    p.add_argument('--work-dir')
    p.add_argument('--eval', type=str, nargs='+', default=['mIoU'], help='mIoU / mDice / mFscore')
    p.add_argument('--split', default='test', choices=['test', 'val'])
    p.add_argument('--revise-checkpoint-key', action='store_true')
    p.add_argument('--gpu-id', type=int, default=0)
    p.add_argument('--cfg-options', nargs='+')
    p.add_argument('--max-images', type=int, default=None)
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    import torch
    import pfst_amd  # noqa: F401
    from pfst_amd.config import Config, parse_cfg_options
    from pfst_amd.evaluation import build_eval_fn, revise_checkpoint_keys
    from pfst_amd.registry import build_segmentor
    cfg = Config.fromfile(args.config)
    if args.cfg_options:
        cfg.merge_from_dict(parse_cfg_options(args.cfg_options))
    torch.cuda.set_device(args.gpu_id)
    dev = torch.device('cuda', args.gpu_id)
    cfg.model['pretrained'] = None
    cfg.model['train_cfg'] = None
    model = build_segmentor(cfg.model)
    ckpt = torch.load(args.checkpoint, map_location='cpu', weights_only=False)
    sd = ckpt.get('state_dict', ckpt)
    if args.revise_checkpoint_key:
        sd = revise_checkpoint_keys(sd)
    missing = model.load_state_dict(sd, strict=False)
    own = [k for k in missing.missing_keys if not k.endswith('num_batches_tracked')]
    if own:
        raise SystemExit(f'{len(own)} segmentor keys are missing from the checkpoint (first: {own[:3]}); a PFGST checkpoint needs '
                         '--revise-checkpoint-key')
    model.CLASSES = ckpt.get('meta', {}).get('CLASSES')
    model.to(dev)
    nc = cfg.model.decode_head.num_classes
    res = build_eval_fn(cfg.data[args.split], nc, dev, metric=args.eval, max_images=args.max_images)(model)
    print(json.dumps(res, indent=1))
    if args.work_dir:
        os.makedirs(args.work_dir, exist_ok=True)
        with open(os.path.join(args.work_dir, 'eval.json'), 'w') as f:
            json.dump(dict(config=args.config, checkpoint=args.checkpoint, metric=res), f, indent=1)
    return res


if __name__ == '__main__':
    main()
