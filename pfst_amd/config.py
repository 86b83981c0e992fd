"""Loader for the reference's python config files (mmcv.Config surface used by tools/train.py:113-115):
`_base_` inheritance with recursive dict merge (`_delete_=True` honoured), attribute access, and
`--cfg-options a.b=v` overrides.  Lets configs/pfst/*.py load unchanged without mmcv."""
import ast
import copy
import os


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


def _merge(base, new):
    out = copy.deepcopy(base)
    for k, v in new.items():
        if isinstance(v, dict) and k in out and isinstance(out[k], dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict):
                v = {a: b for a, b in v.items() if a != '_delete_'}
            out[k] = copy.deepcopy(v)
    return out


def _load_file(path):
    path = os.path.abspath(path)
    with open(path) as f:
        src = f.read()
    ns = {'__file__': path}
    exec(compile(src, path, 'exec'), ns)
    cfg = {k: v for k, v in ns.items() if not k.startswith('__') and not callable(v) and not isinstance(v, type(os))}
    bases = cfg.pop('_base_', [])
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        merged = _merge(merged, _load_file(os.path.join(os.path.dirname(path), b)))
    return _merge(merged, cfg)


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg', _wrap(cfg_dict or {}))
        object.__setattr__(self, 'filename', filename)

    @staticmethod
    def fromfile(path):
        return Config(_load_file(path), filename=path)

    def merge_from_dict(self, options):
        """options: {'a.b.c': value}  (tools/train.py --cfg-options)"""
        for key, val in options.items():
            d = self._cfg
            parts = key.split('.')
            for p in parts[:-1]:
                d = d.setdefault(p, ConfigDict())
            d[parts[-1]] = _wrap(val)

    def __getattr__(self, k):
        return getattr(self._cfg, k)

    def __setattr__(self, k, v):
        self._cfg[k] = _wrap(v)

    def __getitem__(self, k):
        return self._cfg[k]

    def __contains__(self, k):
        return k in self._cfg

    def get(self, k, default=None):
        return self._cfg.get(k, default)

    def to_dict(self):
        return copy.deepcopy(dict(self._cfg))


def parse_cfg_options(pairs):
    """['a.b=1', 'c=[1,2]'] -> dict with python-literal values."""
    out = {}
    for p in pairs or []:
        k, v = p.split('=', 1)
        try:
            out[k] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            out[k] = v
    return out
