"""Batch sources of the train step and the evaluation.

What the hot path consumes is the CONTRACT of the reference's UDADataset after collation (rsiseg/datasets/
uda_dataset.py:116-135): img / img_metas / gt_semantic_seg / target_img / target_img_metas / target_img_strong_aug.
Sources:
  * synthetic_loader   -- seeded synthetic device batches (benchmarks, tests);
  * TileFolder         -- a folder dataset of converted tiles (rsiseg/datasets/custom.py:85-175, isprs.py) read with PIL and pushed
                          through the config's own pipeline list (pfst_amd/pipeline.py);
  * UDADataset / UDADatasetV2 -- the source/target pairings + rare-class sampling of the reference, same NumPy RNG stream;
  * epoch_indices / uda_batches -- the distributed sampler (samplers/distributed_sampler.py:11-70) and the collation."""
import json
import os

import numpy as np
import torch
import torch.utils.data

from .pipeline import Pipeline, reduce_zero_label
from .synthetic import synth_batch

ISPRS_CLASSES = ('impervious_surface', 'building', 'low_vegetation', 'tree', 'car', 'clutter')      # rsiseg/datasets/isprs.py
ISPRS_PALETTE = [[255, 255, 255], [0, 0, 255], [0, 255, 255], [0, 255, 0], [255, 255, 0], [255, 0, 0]]


def synthetic_loader(batch_size, size, num_classes, in_channels=3, seed=1234, device='cuda', distinct=4):
    """infinite iterator cycling over `distinct` pre-generated device batches"""
    batches = [synth_batch(batch_size, size, num_classes, in_channels, seed=seed + i, device=device) for i in range(distinct)]
    i = 0
    while True:
        yield batches[i % distinct]
        i += 1


# ----------------------------------------------------------------------------------------------------------------------
# folder dataset
# ----------------------------------------------------------------------------------------------------------------------
def _read_image_bgr(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert('RGB'), np.uint8)[..., ::-1]      # BGR like mmcv.imread


def _read_label(path):
    from PIL import Image
    return np.asarray(Image.open(path), np.uint8)                                # imdecode_backend='pillow' (loading.py:118)


class TileFolder:
    """`img_dir/*.png` (+ `ann_dir/*.png`) under `data_root`, items produced by the config's pipeline.

    cfg: the reference's dataset dict (type / data_root / img_dir / ann_dir / pipeline / gt_seg_map_loader_cfg ...);
    test_mode: no annotations in the items (apis/train.py:153 builds the validation set this way); the ground truth for the
    metric comes from `gt_seg_map(idx)`, loaded as `gt_seg_map_loader_cfg` says (custom.py:236-252)."""

    def __init__(self, cfg, test_mode=False):
        cfg = dict(cfg)
        if 'img_dir' not in cfg:
            # EODataset configs (inria_da, season_net) read through Dataset4EO datapipes (rsiseg/datasets/custom.py:18, third party, not
            # available offline): their tiles have to be exported to img_dir / ann_dir folders first
            raise NotImplementedError(f"dataset type {cfg.get('type')!r} (datapipe {cfg.get('datapipe')!r}) reads through Dataset4EO; "
                                      "TileFolder needs 'img_dir' (+ 'ann_dir') folders of tiles")
        root = cfg.get('data_root') or ''
        self.img_dir = os.path.join(root, cfg['img_dir'])
        self.ann_dir = os.path.join(root, cfg['ann_dir']) if cfg.get('ann_dir') else None
        self.img_suffix, self.seg_suffix = cfg.get('img_suffix', '.png'), cfg.get('seg_map_suffix', '.png')
        self.ignore_index = cfg.get('ignore_index', 255)
        self.CLASSES = tuple(cfg.get('classes') or ISPRS_CLASSES)
        self.PALETTE = cfg.get('palette') or ISPRS_PALETTE
        self.test_mode = test_mode
        self.pipeline = Pipeline(cfg['pipeline'])
        self.gt_loader_cfg = dict(cfg.get('gt_seg_map_loader_cfg') or {})
        names = sorted(f for f in os.listdir(self.img_dir) if f.endswith(self.img_suffix))
        self.img_infos = [dict(filename=f, ann=dict(seg_map=f[:-len(self.img_suffix)] + self.seg_suffix)) for f in names]

    def __len__(self):
        return len(self.img_infos)

    def gt_seg_map(self, idx):
        seg = _read_label(os.path.join(self.ann_dir, self.img_infos[idx]['ann']['seg_map']))
        return reduce_zero_label(seg) if self.gt_loader_cfg.get('reduce_zero_label') else seg

    def __getitem__(self, idx):
        info = self.img_infos[idx]
        img = _read_image_bgr(os.path.join(self.img_dir, info['filename']))
        seg = None
        if not self.test_mode and self.ann_dir is not None:
            seg = _read_label(os.path.join(self.ann_dir, info['ann']['seg_map']))
        out = self.pipeline(img, seg)
        item = {k: torch.from_numpy(v) for k, v in out.items() if isinstance(v, np.ndarray)}
        item['img_metas'] = dict(filename=info['filename'], ori_shape=img.shape, img_shape=tuple(out['img'].shape[1:]) + (3,),
                                 img_norm_cfg=out['img_norm_cfg'])
        return item


# ----------------------------------------------------------------------------------------------------------------------
# source / target pairing (+ rare-class sampling)
# ----------------------------------------------------------------------------------------------------------------------
class RareClassSampling:
    """uda_dataset.py:17-40,57-90,92-114: classes ordered by total pixel count, p = softmax((1 - freq) / T) evaluated with
    torch in float32 (the probabilities feed np.random.choice, so the arithmetic is the reference's), the files holding
    more than `min_pixels` of each class, and the draw: class -> file -> up to 10 re-crops -> a uniform target."""

    def __init__(self, source, options, stats_root, path2name):
        self.temperature = options['class_temp']
        self.min_crop_ratio, self.min_pixels = options['min_crop_ratio'], options['min_pixels']
        totals = {}
        with open(os.path.join(stats_root, 'sample_class_stats.json')) as f:
            for per_file in json.load(f):
                for key, n in per_file.items():
                    if key != 'file':
                        totals[int(key)] = totals.get(int(key), 0) + n
        order = sorted(totals.items(), key=lambda kv: kv[1])               # stable: ties keep first-seen order like dict sorting
        self.classes = [c for c, _ in order]
        freq = torch.tensor([n for _, n in order])
        freq = freq / torch.sum(freq)
        self.probs = torch.softmax((1 - freq) / self.temperature, dim=-1).numpy()
        with open(os.path.join(stats_root, 'samples_with_class.json')) as f:
            listing = {int(k): v for k, v in json.load(f).items()}
        self.files = {}
        for c in self.classes:
            self.files[c] = [path.split('/')[-1] for path, pixels in listing[c] if pixels > self.min_pixels]
            assert len(self.files[c]) > 0
        self.index_of = {}
        for i, info in enumerate(source.img_infos):
            name = info['ann']['seg_map']
            self.index_of[name.split('/')[-1] if path2name else name] = i

    def draw(self, source, target):
        c = np.random.choice(self.classes, p=self.probs)
        i = self.index_of[np.random.choice(self.files[c])]
        item = source[i]
        if self.min_crop_ratio > 0:
            for _ in range(10):
                if torch.sum(item['gt_semantic_seg'].data == c) > self.min_pixels * self.min_crop_ratio:
                    break
                item = source[i]                          # another random crop of the same tile
        other = target[np.random.choice(range(len(target)))]
        return {**item, 'target_img_metas': other['img_metas'], 'target_img': other['img']}


class UDADataset:
    """sample idx = (source[idx // len(target)], target[idx % len(target)]) (uda_dataset.py:116-135); with
    `rare_class_sampling` every item is a RareClassSampling.draw instead."""

    def __init__(self, source, target, cfg):
        assert target.ignore_index == source.ignore_index and target.CLASSES == source.CLASSES and target.PALETTE == source.PALETTE
        self.source, self.target = source, target
        self.ignore_index, self.CLASSES, self.PALETTE = target.ignore_index, target.CLASSES, target.PALETTE
        self.path2name = cfg.get('path2name', False)
        options = cfg.get('rare_class_sampling')
        self.rcs_enabled = options is not None
        if self.rcs_enabled:
            root = cfg['source'].get('rcs_root', cfg['source']['data_root'])
            self.rcs = RareClassSampling(source, options, root, self.path2name)
            self.rcs_classes, self.rcs_classprob = self.rcs.classes, self.rcs.probs

    def __len__(self):
        return len(self.source) * len(self.target)

    def __getitem__(self, idx):
        if self.rcs_enabled:
            return self.rcs.draw(self.source, self.target)
        a, b = self.source[idx // len(self.target)], self.target[idx % len(self.target)]
        item = {**a, 'target_img_metas': b['img_metas'], 'target_img': b['img']}
        for key, name in (('img_strong_aug', 'target_img_strong_aug'), ('ori_img', 'target_img_ori')):
            if key in b:
                item[name] = b[key]
        return item


class UDADatasetV2(UDADataset):
    """The pairing of the season_net config (uda_dataset_v2.py:43-140): one item per SOURCE sample; the target index is drawn with
    np.random.choice before the source item runs its pipeline (the order of draws from the global NumPy stream), rare-class sampling as in
    UDADataset."""

    def __len__(self):
        return len(self.source)

    def __getitem__(self, idx):
        if self.rcs_enabled:
            return self.rcs.draw(self.source, self.target)
        j = np.random.choice(range(len(self.target)))
        a = self.source[idx]
        b = self.target[j]
        item = {**a, 'target_img_metas': b['img_metas'], 'target_img': b['img']}
        for key, name in (('img_strong_aug', 'target_img_strong_aug'), ('ori_img', 'target_img_ori')):
            if key in b:
                item[name] = b[key]
        return item


UDA_DATASET_TYPES = {'UDADataset': UDADataset, 'UDADatasetV2': UDADatasetV2}


def build_uda_dataset(train_cfg):
    """cfg.data.train of the PFST configs: dict(type='UDADataset' | 'UDADatasetV2', source=..., target=..., rare_class_sampling=...)"""
    kind = train_cfg.get('type', 'UDADataset')
    if kind not in UDA_DATASET_TYPES:
        raise KeyError(f'unknown UDA dataset type {kind!r} (have {sorted(UDA_DATASET_TYPES)})')
    return UDA_DATASET_TYPES[kind](TileFolder(train_cfg['source']), TileFolder(train_cfg['target']), train_cfg)


# ----------------------------------------------------------------------------------------------------------------------
# sampler + collation
# ----------------------------------------------------------------------------------------------------------------------
def epoch_indices(n, world=1, rank=0, epoch=0, seed=0, shuffle=True):
    """samplers/distributed_sampler.py:42-70 on top of torch's DistributedSampler: one seeded permutation per epoch shared by
    all ranks, padded by wrapping to a multiple of the world size, rank r takes every world-th index from r."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(epoch + seed)
        order = torch.randperm(n, generator=g).tolist()
    else:
        order = list(range(n))
    per_rank = -(-n // world)
    total = per_rank * world
    order += order[:total - len(order)]
    return order[rank:total:world]


def collate(items, device):
    """mmcv collate of DataContainers for the PFST keys: stacked tensors, lists of metas"""
    batch = {}
    for key in ('img', 'gt_semantic_seg', 'target_img', 'target_img_strong_aug'):
        if key in items[0]:
            batch[key] = torch.stack([it[key] for it in items]).to(device, non_blocking=True)
    batch['img_metas'] = [it['img_metas'] for it in items]
    batch['target_img_metas'] = [it['target_img_metas'] for it in items]
    return batch


def uda_batches(dataset, batch_size, device='cuda', seed=0, rank=0, world=1, start_epoch=0):
    """infinite iterator of collated device batches, epoch after epoch (drop_last=True like rsiseg/datasets/builder.py:107), loaded
    inline on the calling thread from the caller's global NumPy stream (the golden-vector tests pin the draw order this way);
    training uses `build_loader` below"""
    epoch = start_epoch
    while True:
        idx = epoch_indices(len(dataset), world, rank, epoch, seed)
        for b in range(len(idx) // batch_size):
            yield collate([dataset[i] for i in idx[b * batch_size:(b + 1) * batch_size]], device)
        epoch += 1


# ----------------------------------------------------------------------------------------------------------------------
# worker-process loader (rsiseg/datasets/builder.py:100-193: DataLoader(num_workers=workers_per_gpu, pin_memory, persistent
# workers, worker_init_fn) behind a DistributedSampler)
# ----------------------------------------------------------------------------------------------------------------------
def sample_seed(seed, epoch, position, rank=0, world=1):
    """32-bit seed of ONE sample: a function of (run seed, epoch, the sample's position in the epoch's global order) only"""
    x = (int(seed) * 0x9E3779B1 + int(epoch) * 0x85EBCA77 + (int(position) * world + rank) * 0xC2B2AE3D + 0x27D4EB2F) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 33
    x = (x * 0xFF51AFD7ED558CCD) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 33
    return int(x & 0xFFFFFFFF)


class _SeededItems(torch.utils.data.Dataset):
    """dataset[(seed, idx)]: the item `idx` drawn from a NumPy / Python stream seeded for this sample.  The caller's global streams
    are put back afterwards: loaded inline (workers = 0) the pipeline must not consume the training thread's NumPy stream, which
    PFGST draws its class-mix choices from (dacs_transforms.py:110-126) -- in the reference those draws live in worker processes."""

    def __init__(self, dataset, reseed):
        self.dataset, self.reseed = dataset, reseed

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, key):
        import random
        seed, idx = key
        if not self.reseed:
            return self.dataset[idx]
        saved = (np.random.get_state(), random.getstate())
        try:
            np.random.seed(seed)
            random.seed(seed)
            return self.dataset[idx]
        finally:
            np.random.set_state(saved[0])
            random.setstate(saved[1])


class _EpochBatches(torch.utils.data.Sampler):
    """infinite batch sampler: the rank's share of one seeded permutation per epoch (epoch_indices), drop_last, every index paired
    with its sample seed"""

    def __init__(self, n, batch_size, seed, rank, world, start_epoch=0):
        self.n, self.batch_size, self.seed, self.rank, self.world, self.epoch = n, batch_size, seed, rank, world, start_epoch

    def __iter__(self):
        while True:
            idx = epoch_indices(self.n, self.world, self.rank, self.epoch, self.seed)
            for b in range(len(idx) // self.batch_size):
                yield [(sample_seed(self.seed, self.epoch, b * self.batch_size + j, self.rank, self.world), idx[b * self.batch_size + j])
                       for j in range(self.batch_size)]
            self.epoch += 1


def _collate_cpu(items):
    batch = {}
    for key in ('img', 'gt_semantic_seg', 'target_img', 'target_img_strong_aug'):
        if key in items[0]:
            batch[key] = torch.stack([it[key] for it in items])
    batch['img_metas'] = [it['img_metas'] for it in items]
    batch['target_img_metas'] = [it['target_img_metas'] for it in items]
    return batch


def _seed_worker_streams(worker_id, num_workers, rank, seed):
    """rsiseg/datasets/builder.py:170-181, verbatim arithmetic: one NumPy / Python stream per worker process"""
    import random
    worker_seed = num_workers * rank + worker_id + seed
    np.random.seed(worker_seed)
    random.seed(worker_seed)


def _join_queue_feeders(timeout=30.0):
    """atexit hook of a worker PROCESS: let the result queue's feeder thread finish before the interpreter finalises.

    torch's worker loop ends with `data_queue.cancel_join_thread(); data_queue.close()` (torch/utils/data/_utils/worker.py), so a
    worker that was fetching a batch when the loader was closed exits while its daemon QueueFeederThread is still pickling that
    batch -- inside torch's shared-memory tensor reduction (C++ with the GIL released).  A `spawn`ed worker leaves through
    sys.exit -> Py_FinalizeEx (a forked one through os._exit, which skips finalisation): when the feeder thread then asks for the
    GIL back, CPython ends it with pthread_exit, whose forced unwinding runs into a noexcept C++ frame of the binding ->
    std::terminate, "terminate called without an active exception", SIGABRT (reproduced on a CPU-only host with
    2 workers closed mid-batch: exit codes -6; with the workers idle at close: 0).  The queue is already closed at this point, so
    the thread drains what it holds and returns by itself; joining it here (atexit runs before daemon threads are cut off) removes
    the race and keeps the worker's real exit status."""
    import threading
    for t in threading.enumerate():
        if t.name == 'QueueFeederThread' and t is not threading.current_thread():
            t.join(timeout)


def _worker_init(worker_id, num_workers, rank, seed, seeding):
    """worker_init_fn of the loader's worker processes (never called in the training process)"""
    import atexit
    torch.set_num_threads(1)                 # the pipeline is NumPy / C per sample; parallelism comes from the worker processes
    atexit.register(_join_queue_feeders)
    if seeding == 'worker':
        _seed_worker_streams(worker_id, num_workers, rank, seed)


def build_loader(dataset, batch_size, device='cuda', seed=0, rank=0, world=1, workers=0, prefetch=2, seeding='sample', start_epoch=0,
                 pin_memory=None):
    """Infinite iterator of device batches fed by `workers` processes (cfg.data.workers_per_gpu), the counterpart of the reference's
    build_dataloader (rsiseg/datasets/builder.py:100-193): torch DataLoader with persistent workers, a bounded prefetch queue
    (`prefetch` batches per worker), pinned host buffers, and the host -> device copy of the NEXT batch issued on a copy stream while
    the current step runs (`device_prefetch`).

    seeding='sample' (default): every sample draws from its own stream, seeded by (seed, epoch, position) -- the batches are then
    identical for ANY worker count, 0 included, and reproducible across runs; within a sample the pipeline makes the reference's
    draws in the reference's order.  seeding='worker': the reference's worker_init_fn (one stream per worker process, seeded
    num_workers * rank + worker_id + seed), whose batches depend on the worker count like the reference's do.
    Workers are started with 'spawn': a forked child of a process that has initialised the GPU would inherit its device handles."""
    from functools import partial
    if seeding not in ('sample', 'worker'):
        raise ValueError(f'seeding={seeding!r}')
    items = _SeededItems(dataset, reseed=seeding == 'sample')
    sampler = _EpochBatches(len(dataset), batch_size, seed, rank, world, start_epoch)
    on_gpu = torch.device(device).type == 'cuda'
    kw = {}
    if workers > 0:
        kw = dict(multiprocessing_context='spawn', persistent_workers=True, prefetch_factor=prefetch,
                  worker_init_fn=partial(_worker_init, num_workers=workers, rank=rank, seed=seed, seeding=seeding))
    elif seeding == 'worker':
        # inline loading: only the streams are seeded -- no thread-count change, no exit hook in the training process
        _seed_worker_streams(0, 0, rank, seed)
    loader = torch.utils.data.DataLoader(items, batch_sampler=sampler, num_workers=workers, collate_fn=_collate_cpu,
                                         pin_memory=on_gpu if pin_memory is None else pin_memory, **kw)
    return BatchLoader(loader, device if on_gpu else None)


class BatchLoader:
    """the iterator `build_loader` returns: next() -> one batch dict; close() stops the worker processes (also on deletion)"""

    def __init__(self, loader, device):
        self._loader = loader
        # read the sampler BEFORE iter(loader): torch's multi-process iterator pulls prefetch_factor * workers batches from it inside its
        # constructor, which can carry a short epoch's sampler past the epoch the run starts in
        bs = loader.batch_sampler
        self._start_epoch = bs.epoch
        self._per_epoch = max(1, len(epoch_indices(bs.n, bs.world, bs.rank, bs.epoch, bs.seed)) // bs.batch_size)
        self._it = iter(loader)
        self._gen = device_prefetch(self._it, device) if device is not None else self._it
        self._delivered = 0

    def __iter__(self):
        return self

    def __next__(self):
        b = next(self._gen)
        self._delivered += 1
        return b

    @property
    def epoch(self):
        """the data epoch of the batch handed out LAST (what mmcv's IterLoader reports to the runner and the checkpoints record) --
        counted on the consumer's side: the sampler itself runs ahead by the workers' prefetch depth plus the device prefetch"""
        return self._start_epoch + max(0, self._delivered - 1) // self._per_epoch

    def close(self):
        it, self._it, self._gen, self._loader = self._it, None, None, None
        if it is not None and hasattr(it, '_shutdown_workers'):
            it._shutdown_workers()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_prefetch(cpu_batches, device):
    """batch k is handed out once its copy has been ordered before the caller's stream; the copy of batch k+1 (from pinned memory, on
    a separate copy stream) is already in flight behind it while step k runs"""
    copy_stream = torch.cuda.Stream(device)

    def start(b):
        with torch.cuda.stream(copy_stream):
            moved = {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in b.items()}
            evt = torch.cuda.Event()
            evt.record(copy_stream)
        return moved, evt, b                      # the pinned source stays referenced until the copy has been waited for

    pending = start(next(cpu_batches))
    while True:
        moved, evt, _src = pending
        pending = start(next(cpu_batches))
        cur = torch.cuda.current_stream(device)
        cur.wait_event(evt)
        for v in moved.values():
            if torch.is_tensor(v):
                v.record_stream(cur)
        yield moved
