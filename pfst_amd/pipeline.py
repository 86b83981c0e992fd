"""CPU data pipeline of the PFST configs, restated in NumPy and driven by the reference's own
`data.train.source.pipeline` / `target.pipeline` / `test_pipeline` lists (configs/_base_/datasets/*.py), so the shipped
dataset configs drive it unchanged.

Reference (all under rsiseg/datasets/pipelines): loading.py:100-163 (LoadAnnotations, reduce_zero_label),
transforms.py:11-260 (Resize: ratio sampled with np.random.random_sample, keep-ratio rescale), :262-329 (RandomFlip),
:331-402 (Pad), :404-451 (Normalize), :644-736 (RandomCrop + cat_max_ratio retries), :942-1059 (PhotoMetricDistortion),
:1061-1160 (StrongAugmentation = the same distortion written to `img_strong_aug`), rsi_aug.py:29-108 (RandomRotate90),
formating.py:189-217 (DefaultFormatBundle), test_time_aug.py (MultiScaleFlipAug with one scale, no flip).

The random decisions use the global NumPy RNG with the reference's calls in the reference's order (np.random.random_sample for
the scale ratio, np.random.randint for crop offsets, np.random.rand / np.random.choice for rotation and flips,
np.random.randint / uniform for the photometric steps), so a seeded worker draws the same augmentation parameters.
PIXEL parity is unpinned: the reference resizes / converts colour with OpenCV (cv2.resize INTER_LINEAR fixed-point
arithmetic, cv2.cvtColor 8-bit HSV), which is not installed here and has no vectors in the reference's tests; this module
implements the same geometry (half-pixel centres, keep-ratio rounding of mmcv.rescale_size) and the documented 8-bit HSV
formulas in float arithmetic.

Images travel as HxWx3 uint8 in BGR order like mmcv.imread, `Normalize(to_rgb=True)` swaps them; labels as HxW uint8."""
import numpy as np

IGNORE = 255

# ----------------------------------------------------------------------------------------------------------------------
# native pixel kernels (csrc/pipeline_cpu.c -> libpfst_cpu.so, built by pfst_amd.build): same results bit for bit as the NumPy
# expressions below (which stay as the readable restatement and as the checker: tests/test_data_loader_cpu.py), 5-10x faster.
# PFST_PIPELINE_NATIVE=0 or a missing library selects NumPy -- both are CPU code of the data path, not the GPU hot path.
# ----------------------------------------------------------------------------------------------------------------------
import ctypes
import os

_NATIVE = None


def native():
    global _NATIVE
    if _NATIVE is None:
        _NATIVE = False
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libpfst_cpu.so')
        if os.environ.get('PFST_PIPELINE_NATIVE', '1') == '1' and os.path.exists(path):
            lib = ctypes.CDLL(path)
            vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
            lib.pfst_cpu_resize_window_u8.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, i32, vp]
            lib.pfst_cpu_bgr2hsv_u8.argtypes = [vp, i64, vp]
            lib.pfst_cpu_hsv2bgr_u8.argtypes = [vp, i64, vp]
            lib.pfst_cpu_convert_u8.argtypes = [vp, i64, i32, f32, f32, vp]
            lib.pfst_cpu_hue_shift_u8.argtypes = [vp, i64, i32]
            lib.pfst_cpu_normalize_u8.argtypes = [vp, i64, vp, vp, i32, vp]
            for f in ('resize_window_u8', 'bgr2hsv_u8', 'hsv2bgr_u8', 'convert_u8', 'hue_shift_u8', 'normalize_u8'):
                getattr(lib, 'pfst_cpu_' + f).restype = None
            _NATIVE = lib
    return _NATIVE


def set_native(on):
    """tests: switch the native kernels off / on again"""
    global _NATIVE
    _NATIVE = None if on else False
    return native() if on else False


def _u8c(a):
    return a if (a.dtype == np.uint8 and a.flags['C_CONTIGUOUS']) else np.ascontiguousarray(a, np.uint8)


# ----------------------------------------------------------------------------------------------------------------------
# pixel operations
# ----------------------------------------------------------------------------------------------------------------------
def reduce_zero_label(seg):
    """loading.py:151-155: 0 -> 255 (ignore), k -> k-1"""
    seg = seg.copy()
    seg[seg == 0] = 255
    seg = seg - 1
    seg[seg == 254] = 255
    return seg


def rescale_size(old_hw, scale):
    """mmcv.rescale_size with a (long, short) edge tuple: the largest size that fits, rounded half up"""
    h, w = old_hw
    long_e, short_e = max(scale), min(scale)
    f = min(long_e / max(h, w), short_e / min(h, w))
    return int(h * float(f) + 0.5), int(w * float(f) + 0.5)


def _src_index(n_out, n_in):
    """half-pixel-centre source coordinates of cv2.resize(INTER_LINEAR): x_src = (x + .5) * in/out - .5, clamped"""
    s = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
    i0 = np.floor(s).astype(np.int64)
    f = s - i0
    lo = np.clip(i0, 0, n_in - 1)
    hi = np.clip(i0 + 1, 0, n_in - 1)
    f = np.where(i0 < 0, 0.0, f)
    return lo, hi, f.astype(np.float32)


def resize_bilinear_u8(img, out_hw):
    h, w = img.shape[:2]
    H, W = out_hw
    if (H, W) == (h, w):
        return img
    y0, y1, fy = _src_index(H, h)
    x0, x1, fx = _src_index(W, w)
    a = img.astype(np.float32)
    top = a[y0][:, x0] * (1 - fx)[None, :, None] + a[y0][:, x1] * fx[None, :, None]
    bot = a[y1][:, x0] * (1 - fx)[None, :, None] + a[y1][:, x1] * fx[None, :, None]
    out = top * (1 - fy)[:, None, None] + bot * fy[:, None, None]
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


class LazyResize:
    """A bilinear resize that has not been computed yet.  The shipped training pipelines resize a 1024^2 tile by a ratio of up
    to 2 (12.6 M output pixels x 3 channels in float arithmetic) and then keep a 512^2 RandomCrop of it: only the crop window is ever
    needed, and a window of the resize is the same arithmetic on the window's rows and columns (same source indices, same weights),
    so `window()` returns exactly the pixels `resize_bilinear_u8(img, hw)[y1:y2, x1:x2]` would.  8-16x less work per sample."""

    def __init__(self, img, out_hw):
        self.img, self.hw = img, (int(out_hw[0]), int(out_hw[1]))

    @property
    def shape(self):
        return self.hw + self.img.shape[2:]

    def window(self, y1, y2, x1, x2):
        h, w = self.img.shape[:2]
        H, W = self.hw
        y2, x2 = min(y2, H), min(x2, W)
        if (H, W) == (h, w):
            return self.img[y1:y2, x1:x2]
        y0, y1i, fy = (a[y1:y2] for a in _src_index(H, h))
        x0, x1i, fx = (a[x1:x2] for a in _src_index(W, w))
        lib = native()
        if lib and self.img.ndim == 3 and self.img.shape[2] == 3:
            src = _u8c(self.img)
            out = np.empty((len(y0), len(x0), 3), np.uint8)
            idx = [np.ascontiguousarray(v) for v in (y0, y1i, fy, x0, x1i, fx)]
            lib.pfst_cpu_resize_window_u8(src.ctypes.data, h, w, *(v.ctypes.data for v in idx), len(y0), len(x0), out.ctypes.data)
            return out
        ylo, yhi = int(min(y0.min(), y1i.min())), int(max(y0.max(), y1i.max())) + 1
        xlo, xhi = int(min(x0.min(), x1i.min())), int(max(x0.max(), x1i.max())) + 1
        a = self.img[ylo:yhi, xlo:xhi].astype(np.float32)          # only the source rows / columns the window reads
        y0, y1i, x0, x1i = y0 - ylo, y1i - ylo, x0 - xlo, x1i - xlo
        top = a[y0][:, x0] * (1 - fx)[None, :, None] + a[y0][:, x1i] * fx[None, :, None]
        bot = a[y1i][:, x0] * (1 - fx)[None, :, None] + a[y1i][:, x1i] * fx[None, :, None]
        out = top * (1 - fy)[:, None, None] + bot * fy[:, None, None]
        return np.clip(np.rint(out), 0, 255).astype(np.uint8)

    def materialize(self):
        return resize_bilinear_u8(self.img, self.hw)


def resize_nearest(seg, out_hw):
    """cv2.resize(INTER_NEAREST): src = floor(dst * in/out)"""
    h, w = seg.shape[:2]
    H, W = out_hw
    if (H, W) == (h, w):
        return seg
    yi = np.minimum((np.arange(H) * (h / H)).astype(np.int64), h - 1)
    xi = np.minimum((np.arange(W) * (w / W)).astype(np.int64), w - 1)
    return seg[yi][:, xi]


def bgr2hsv_u8(img):
    """cv2.cvtColor(BGR2HSV) for 8-bit images: H in [0, 180), S, V in [0, 255]"""
    lib = native()
    if lib:
        src = _u8c(img)
        out = np.empty(src.shape, np.uint8)
        lib.pfst_cpu_bgr2hsv_u8(src.ctypes.data, src.size // 3, out.ctypes.data)
        return out
    return bgr2hsv_np(img)


def bgr2hsv_np(img):
    b, g, r = [img[..., i].astype(np.float32) for i in range(3)]
    v = np.maximum(np.maximum(b, g), r)
    mn = np.minimum(np.minimum(b, g), r)
    d = v - mn
    s = np.where(v > 0, d / np.maximum(v, 1e-12) * 255.0, 0.0)
    dd = np.maximum(d, 1e-12)
    h = np.where(v == r, (g - b) / dd, np.where(v == g, 2.0 + (b - r) / dd, 4.0 + (r - g) / dd)) * 60.0
    h = np.where(d == 0, 0.0, h)
    h = np.where(h < 0, h + 360.0, h) / 2.0
    out = np.stack([np.rint(h) % 180, np.rint(s), v], -1)
    return np.clip(out, 0, 255).astype(np.uint8)


def hsv2bgr_u8(hsv):
    lib = native()
    if lib:
        src = _u8c(hsv)
        out = np.empty(src.shape, np.uint8)
        lib.pfst_cpu_hsv2bgr_u8(src.ctypes.data, src.size // 3, out.ctypes.data)
        return out
    return hsv2bgr_np(hsv)


def hsv2bgr_np(hsv):
    h = hsv[..., 0].astype(np.float32) * 2.0
    s = hsv[..., 1].astype(np.float32) / 255.0
    v = hsv[..., 2].astype(np.float32)
    c = v * s
    hp = h / 60.0
    x = c * (1 - np.abs(hp % 2 - 1))
    z = np.zeros_like(c)
    sector = np.floor(hp).astype(np.int64) % 6
    r = np.choose(sector, [c, x, z, z, x, c])
    g = np.choose(sector, [x, c, c, x, z, z])
    b = np.choose(sector, [z, z, x, c, c, x])
    m = v - c
    out = np.stack([b + m, g + m, r + m], -1)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def _convert(img, alpha=1.0, beta=0.0):
    """transforms.py:975-979: float multiply-add, clip, truncate to uint8"""
    lib = native()
    if lib and img.dtype == np.uint8 and img.flags['C_CONTIGUOUS']:
        out = np.empty(img.shape, np.uint8)
        lib.pfst_cpu_convert_u8(img.ctypes.data, img.size, 1, alpha, beta, out.ctypes.data)
        return out
    return np.clip(img.astype(np.float32) * alpha + beta, 0, 255).astype(np.uint8)


def photometric_distortion(img, brightness_delta=32, contrast_range=(0.5, 1.5), saturation_range=(0.5, 1.5), hue_delta=18):
    """transforms.py:981-1048 (and StrongAugmentation :1077-1144): every step with probability 1/2, contrast before or after
    the HSV steps; the RNG calls are numpy.random's randint / uniform in the reference's order"""
    rnd = np.random
    if rnd.randint(2):
        img = _convert(img, beta=rnd.uniform(-brightness_delta, brightness_delta))
    mode = rnd.randint(2)
    if mode == 1 and rnd.randint(2):
        img = _convert(img, alpha=rnd.uniform(*contrast_range))
    if rnd.randint(2):
        hsv = bgr2hsv_u8(img)
        alpha = rnd.uniform(*saturation_range)
        lib = native()
        if lib:         # the saturation channel of the interleaved image, in place
            lib.pfst_cpu_convert_u8(hsv.ctypes.data + 1, hsv.size // 3, 3, alpha, 0.0, hsv.ctypes.data + 1)
        else:
            hsv[..., 1] = _convert(hsv[..., 1], alpha=alpha)
        img = hsv2bgr_u8(hsv)
    if rnd.randint(2):
        hsv = bgr2hsv_u8(img)
        delta = rnd.randint(-hue_delta, hue_delta)
        lib = native()
        if lib:
            lib.pfst_cpu_hue_shift_u8(hsv.ctypes.data, hsv.size // 3, int(delta))
        else:
            hsv[..., 0] = (hsv[..., 0].astype(int) + delta) % 180
        img = hsv2bgr_u8(hsv)
    if mode == 0 and rnd.randint(2):
        img = _convert(img, alpha=rnd.uniform(*contrast_range))
    return img


def normalize(img, mean, std, to_rgb=True):
    """mmcv.imnormalize: (BGR -> RGB,) subtract mean, divide by std, float32"""
    lib = native()
    if lib and img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3 and len(mean) == 3:
        src = _u8c(img)
        m, sd = np.asarray(mean, np.float32), np.asarray(std, np.float32)
        out = np.empty(src.shape, np.float32)
        lib.pfst_cpu_normalize_u8(src.ctypes.data, src.size // 3, m.ctypes.data, sd.ctypes.data, int(bool(to_rgb)), out.ctypes.data)
        return out
    a = img.astype(np.float32)
    if to_rgb:
        a = a[..., ::-1]
    return (a - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)


def pad_to(a, hw, value):
    """mmcv.impad(shape=...): bottom / right padding"""
    ph, pw = max(hw[0] - a.shape[0], 0), max(hw[1] - a.shape[1], 0)
    if not (ph or pw):
        return a
    widths = ((0, ph), (0, pw)) + ((0, 0),) * (a.ndim - 2)
    return np.pad(a, widths, constant_values=value)


# ----------------------------------------------------------------------------------------------------------------------
# the pipeline: a list of (name, kwargs) steps compiled from the reference's config list
# ----------------------------------------------------------------------------------------------------------------------
_KNOWN = {'LoadImageFromFile', 'LoadAnnotations', 'LoadAnnotationsPseudoLabelsV2', 'Resize', 'RandomCrop', 'RandomRotate90',
          'RandomFlip', 'StrongAugmentation', 'PhotoMetricDistortion', 'Normalize', 'Pad', 'DefaultFormatBundle', 'Collect',
          'ImageToTensor', 'MultiScaleFlipAug', 'ClipNormalize', 'Uint82Float'}


class Pipeline:
    """steps: the reference's list of dict(type=..., **kw).  __call__(img_bgr_u8, seg_u8 | None) -> dict with float32 CHW arrays
    `img` (+ `img_strong_aug`), uint8 `gt_semantic_seg` [1,H,W] and `img_norm_cfg`."""

    lazy_resize = True          # Resize followed by RandomCrop computes the crop window only (LazyResize); False: the eager order

    def __init__(self, steps):
        flat = []
        for s in steps:
            s = dict(s)
            t = s.pop('type')
            if t not in _KNOWN:
                raise NotImplementedError(f'pipeline step {t} is outside the PFST dataset configs')
            if t == 'MultiScaleFlipAug':           # one scale, no flip (the shipped test pipelines): inline its transforms
                if s.get('flip') or s.get('img_ratios') is not None:
                    raise NotImplementedError('multi-scale / flip test-time augmentation is outside the PFST configs')
                scale = s['img_scale']
                for q in s['transforms']:
                    q = dict(q)
                    qt = q.pop('type')
                    if qt == 'Resize':
                        q.setdefault('img_scale', scale)
                    flat.append((qt, q))
                continue
            flat.append((t, s))
        self.steps = flat
        self.reduce_zero_label = any(k.get('reduce_zero_label') for t, k in flat if t.startswith('LoadAnnotations'))
        # LoadAnnotationsPseudoLabelsV2(pseudo_labels_dir=None) (loading.py:463-468, the shipped target pipelines): the target's label map is
        # all `ignore`.  It is never forwarded (uda_dataset.py:127-133) but RandomCrop(cat_max_ratio < 1) walks its ten retries on it, so the
        # crop that is kept is the ELEVENTH box drawn: the map has to exist for the NumPy stream to match.
        self.blank_labels = False
        for t, k in flat:
            if t == 'LoadAnnotationsPseudoLabelsV2':
                if k.get('pseudo_labels_dir') is not None:
                    raise NotImplementedError('pseudo labels from .h5 files (pseudo_labels_dir) are outside the PFST configs')
                self.blank_labels = True

    def __call__(self, img, seg=None):
        out = {'img': img}
        blank = seg is None and self.blank_labels      # an all-ignore label map: its resizes / crop histograms are known without computing
        if blank:
            seg = np.full(img.shape[:2], IGNORE, np.uint8)
        elif seg is not None and self.reduce_zero_label:
            seg = reduce_zero_label(seg)
        # LoadImageFromFile's default (loading.py:80-84): what the metas carry when the pipeline has no Normalize step (season_net)
        nch = 1 if img.ndim < 3 else img.shape[2]
        norm_cfg = dict(mean=[0.0] * nch, std=[1.0] * nch, to_rgb=False)
        for t, k in self.steps:
            if isinstance(out['img'], LazyResize) and t not in ('RandomCrop', 'Resize', 'LoadImageFromFile', 'LoadAnnotations',
                                                                'LoadAnnotationsPseudoLabelsV2'):
                out['img'] = out['img'].materialize()          # a step other than the crop needs the pixels
            if t == 'Resize':
                scale = k.get('img_scale')
                scale = tuple(scale[0]) if isinstance(scale, (list, tuple)) and isinstance(scale[0], (list, tuple)) else tuple(scale)
                if k.get('ratio_range') is not None:
                    lo, hi = k['ratio_range']
                    ratio = np.random.random_sample() * (hi - lo) + lo
                    scale = (int(scale[0] * ratio), int(scale[1] * ratio))
                if k.get('keep_ratio', True):
                    hw = rescale_size(out['img'].shape[:2], scale)
                else:
                    hw = (scale[1], scale[0])
                img_now = out['img'].materialize() if isinstance(out['img'], LazyResize) else out['img']
                out['img'] = LazyResize(img_now, hw) if self.lazy_resize and img_now.ndim == 3 else resize_bilinear_u8(img_now, hw)
                if seg is not None:
                    seg = np.full(hw, IGNORE, np.uint8) if blank else resize_nearest(seg, hw)
            elif t == 'RandomCrop':
                ch, cw = k['crop_size']
                ratio, ign = k.get('cat_max_ratio', 1.0), k.get('ignore_index', IGNORE)

                def bbox():
                    mh, mw = max(out['img'].shape[0] - ch, 0), max(out['img'].shape[1] - cw, 0)
                    oh, ow = np.random.randint(0, mh + 1), np.random.randint(0, mw + 1)
                    return oh, oh + ch, ow, ow + cw
                y1, y2, x1, x2 = bbox()
                if ratio < 1.0 and seg is not None:
                    for _ in range(10):
                        if blank:                 # no class besides `ignore`: the test below can never pass, the ten boxes are drawn
                            y1, y2, x1, x2 = bbox()
                            continue
                        hist = np.bincount(seg[y1:y2, x1:x2].ravel(), minlength=256)      # = np.unique(..., return_counts=True) on uint8
                        labels = np.nonzero(hist)[0]
                        cnt = hist[labels][labels != ign]
                        if len(cnt) > 1 and np.max(cnt) / np.sum(cnt) < ratio:
                            break
                        y1, y2, x1, x2 = bbox()
                out['img'] = out['img'].window(y1, y2, x1, x2) if isinstance(out['img'], LazyResize) else out['img'][y1:y2, x1:x2]
                if seg is not None:
                    seg = seg[y1:y2, x1:x2]
            elif t == 'RandomRotate90':
                if np.random.rand() < k.get('prob', 1.0):
                    rot = int(np.random.choice([0, 1, 2, 3]))
                    out['img'] = np.rot90(out['img'], k=rot, axes=(0, 1)).copy()
                    if seg is not None:
                        seg = np.rot90(seg, k=rot, axes=(0, 1)).copy()
            elif t == 'RandomFlip':
                p = k.get('prob', k.get('flip_ratio'))
                if np.random.rand() < p:
                    ax = 1 if k.get('direction', 'horizontal') == 'horizontal' else 0
                    out['img'] = np.flip(out['img'], ax)
                    if seg is not None:
                        seg = np.flip(seg, ax).copy()
            elif t == 'StrongAugmentation':
                out['img_strong_aug'] = photometric_distortion(np.ascontiguousarray(out['img']), **k)
            elif t == 'PhotoMetricDistortion':
                out['img'] = photometric_distortion(np.ascontiguousarray(out['img']), **k)
            elif t == 'Normalize':
                norm_cfg = dict(mean=list(k['mean']), std=list(k['std']), to_rgb=k.get('to_rgb', True))
                for key in ('img', 'img_strong_aug'):
                    if key in out:
                        out[key] = normalize(out[key], k['mean'], k['std'], k.get('to_rgb', True))
            elif t == 'ClipNormalize':          # season_net pipelines (transforms.py:1166-1212): [mean - 2 std, mean + 2 std] -> [0, 1]
                mean = np.array(k['mean']).astype(np.float32).reshape(1, 1, -1)
                std = np.array(k['std']).astype(np.float32).reshape(1, 1, -1)
                lo, hi = mean - 2 * std, mean + 2 * std
                img = np.clip((out['img'].astype(np.float32) - lo) / (hi - lo), 0, 1)
                if k.get('to_rgb', True):
                    img = img[:, :, [2, 1, 0]]
                if k.get('to_uint8', False):
                    img = (img * 255).astype(np.uint8)
                out['img'] = img
            elif t == 'Uint82Float':            # transforms.py:1215-1221
                for key in ('img', 'img_strong_aug'):
                    if key in out:
                        out[key] = out[key].astype(np.float32) / 255
            elif t == 'Pad':
                if k.get('size') is None:
                    raise NotImplementedError('Pad(size_divisor) is outside the PFST configs')
                for key in ('img', 'img_strong_aug'):
                    if key in out:
                        out[key] = pad_to(out[key], k['size'], k.get('pad_val', 0))
                if seg is not None:
                    seg = pad_to(seg, out['img'].shape[:2], k.get('seg_pad_val', IGNORE))
        if isinstance(out['img'], LazyResize):
            out['img'] = out['img'].materialize()
        res = {key: np.ascontiguousarray(out[key].transpose(2, 0, 1), dtype=np.float32) for key in ('img', 'img_strong_aug') if key in out}
        if seg is not None:
            res['gt_semantic_seg'] = np.ascontiguousarray(seg, dtype=np.uint8)[None]
        res['img_norm_cfg'] = norm_cfg
        return res
