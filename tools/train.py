#!/usr/bin/env python3
"""Train a PFST model on MI355X -- the flag surface of the reference's tools/train.py:23-107 on top of pfst_amd.

  python tools/train.py configs/pfst/pfst_pots_irrg2vaih_irrg_deeplabv3plus_r50-d8.py --work-dir work_dirs/x
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py CONFIG --launcher pytorch

`--synthetic` trains on seeded synthetic tiles (benchmarks); otherwise `cfg.data.train` (UDADataset: source / target folders of
converted tiles, each with the config's own pipeline list) is read by pfst_amd/data.py + pfst_amd/pipeline.py, and
`cfg.data.val` is evaluated every `evaluation.interval` iterations unless --no-validate (rsiseg/apis/train.py:152-168)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='Train a segmentor (pfst_amd)')
    p.add_argument('config', help='train config file path (reference format) or a preset name from pfst_amd.presets')
    p.add_argument('--work-dir')
    p.add_argument('--load-from')
    p.add_argument('--resume-from')
    p.add_argument('--no-validate', action='store_true')
    g = p.add_mutually_exclusive_group()
    g.add_argument('--gpus', type=int)
    g.add_argument('--gpu-ids', type=int, nargs='+')
    g.add_argument('--gpu-id', type=int, default=0)
    p.add_argument('--seed', type=int, default=None)
    p.add_argument('--diff_seed', action='store_true')
    p.add_argument('--deterministic', action='store_true')
    p.add_argument('--options', nargs='+')
    p.add_argument('--cfg-options', nargs='+')
    p.add_argument('--launcher', choices=['none', 'pytorch', 'slurm', 'mpi'], default='none')
    p.add_argument('--local_rank', '--local-rank', type=int, default=0)
    p.add_argument('--auto-resume', action='store_true')
    p.add_argument('--synthetic', action='store_true', help='seeded synthetic batches instead of cfg.data')
    p.add_argument('--random-init', action='store_true',
                   help='train from random initialisation when cfg.model.pretrained cannot be loaded (model-zoo URL, no network)')
    p.add_argument('--max-iters', type=int, default=None)
    p.add_argument('--batch-size', type=int, default=None)
    p.add_argument('--crop-size', type=int, default=None)
    p.add_argument('--workers', type=int, default=None, help='data-loading worker processes per GPU (default: cfg.data.workers_per_gpu)')
    p.add_argument('--seeding', choices=['sample', 'worker'], default='sample',
                   help="'sample': one RNG stream per sample (batches independent of the worker count); 'worker': the reference's "
                        'worker_init_fn (rsiseg/datasets/builder.py:170-181)')
    args = p.parse_args(argv)
    if args.options and args.cfg_options:
        raise ValueError('--options and --cfg-options cannot be both specified')
    if args.options:
        args.cfg_options = args.options
    if 'LOCAL_RANK' not in os.environ:
        os.environ['LOCAL_RANK'] = str(args.local_rank)
    return args


def load_cfg(args):
    from pfst_amd.config import Config, parse_cfg_options
    from pfst_amd.presets import LR_CONFIG, OPTIMIZER, WORKLOADS, workload_cfg
    if os.path.exists(args.config):
        cfg = Config.fromfile(args.config)
    elif args.config in WORKLOADS:
        uda, w = workload_cfg(args.config)
        cfg = Config(dict(model=uda.pop('model'), uda=uda, optimizer=dict(OPTIMIZER), lr_config=dict(LR_CONFIG),
                          runner=dict(type='IterBasedRunner', max_iters=40000), checkpoint_config=dict(by_epoch=False, interval=4000),
                          evaluation=dict(interval=4000, metric='mIoU'), log_config=dict(interval=50),
                          data=dict(samples_per_gpu=w['per_gpu_batch']), seed=0))
    else:
        raise FileNotFoundError(args.config)
    if args.cfg_options:
        cfg.merge_from_dict(parse_cfg_options(args.cfg_options))
    if args.max_iters:
        cfg.runner['max_iters'] = args.max_iters
    return cfg


def main(argv=None):
    args = parse_args(argv)
    import torch
    import torch.distributed as dist
    import pfst_amd  # noqa: F401
    from pfst_amd import dist as pdist
    from pfst_amd.data import build_loader, build_uda_dataset, synthetic_loader
    from pfst_amd.evaluation import build_eval_fn
    from pfst_amd.optim import build_optimizer
    from pfst_amd.registry import build_train_model
    from pfst_amd.runner import IterBasedRunner, find_latest_checkpoint, init_random_seed, set_random_seed

    cfg = load_cfg(args)
    distributed = args.launcher != 'none'
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local_rank if distributed else args.gpu_id)
    dev = torch.device('cuda', torch.cuda.current_device())
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group((cfg.get('dist_params') or {}).get('backend', 'nccl'), device_id=dev)
    rank = dist.get_rank() if distributed else 0
    world = dist.get_world_size() if distributed else 1
    work_dir = args.work_dir or cfg.get('work_dir') or os.path.join('./work_dirs', os.path.splitext(os.path.basename(args.config))[0])
    seed = init_random_seed(args.seed if args.seed is not None else cfg.get('seed'), dev)
    seed = seed + rank if args.diff_seed else seed
    set_random_seed(seed, args.deterministic)

    load_from = args.load_from or cfg.get('load_from')
    resume = args.resume_from or cfg.get('resume_from')
    if resume is None and args.auto_resume:
        resume = find_latest_checkpoint(work_dir)
    pretrained = cfg.model.get('pretrained')
    if pretrained and not os.path.exists(str(pretrained)):
        # e.g. 'open-mmlab://resnet50_v1c': a model-zoo URL cannot be fetched here.  Never silently replace it by random weights.
        if not (load_from or resume or args.random_init or args.synthetic):
            raise SystemExit(f'cfg.model.pretrained={pretrained!r} cannot be loaded offline: give --load-from / --resume-from a '
                             'checkpoint, point `pretrained` to a local file, or pass --random-init explicitly')
        cfg.model['pretrained'] = None
    model = build_train_model(cfg)
    model.init_weights()
    model.to(dev)
    optimizer = build_optimizer(model, cfg.optimizer)
    nc = cfg.model.decode_head.num_classes
    eval_fn = None
    data_cfg = cfg.get('data') or {}
    if not args.no_validate and not args.synthetic and 'val' in data_cfg:
        ev = cfg.get('evaluation') or {}
        eval_fn = build_eval_fn(data_cfg['val'], nc, dev, metric=ev.get('metric', 'mIoU'))
    runner = IterBasedRunner(model, optimizer, cfg, work_dir, eval_fn=eval_fn)
    if args.deterministic and rank == 0:
        # measured at b = 8 x 1024^2 (tools/det_cost.py): 296 ms against 288 ms per step -- the split-K slices of the weight gradients go through a
        # scratch and an ordered reduction instead of atomics, the launch shapes are the default mode's
        runner.log('deterministic mode: weight gradients / BatchNorm-backward / depthwise / bias sums in a fixed order -- bit-reproducible '
                   'gradients run to run and for any stream schedule; about 3 % slower (compare the `time` column with a run without '
                   '--deterministic)')
    if load_from:
        runner.load_checkpoint(load_from)
    if resume:
        runner.resume(resume)
    # every rank continues from rank 0's parameters and buffers (what MMDistributedDataParallel's constructor does in the reference);
    # only then may the per-rank random streams diverge (--diff_seed)
    pdist.broadcast_module_state_(model)

    bs = args.batch_size or data_cfg.get('samples_per_gpu', 2)
    cin = cfg.model.backbone.get('in_channels', 3)
    if args.synthetic or 'train' not in data_cfg:
        loader = synthetic_loader(bs, args.crop_size or 1024, nc, cin, seed=1234 + rank, device=dev)
    else:
        # img_scale / ratio_range / crop_size / reduce_zero_label / flips / photometric steps all come from the config's pipelines
        dataset = build_uda_dataset(data_cfg['train'])
        model.CLASSES = dataset.CLASSES
        model.PALETTE = dataset.PALETTE
        # the RESOLVED seed (--seed / cfg.seed / the broadcast random one) drives the sampler's shuffle (apis/train.py:74-89 passes
        # cfg.seed to build_dataloader) and the per-sample streams; the data side never touches the training thread's NumPy stream
        workers = args.workers if args.workers is not None else int(data_cfg.get('workers_per_gpu', 0))
        loader = build_loader(dataset, bs, dev, seed=seed, rank=rank, world=world, workers=workers, seeding=args.seeding,
                              start_epoch=runner.epoch)          # a resumed run continues with the checkpoint's data epoch
    runner.run(iter(loader))
    runner.save_checkpoint()
    if hasattr(loader, 'close'):
        loader.close()                       # stop the data-loading worker processes
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
