"""Minimal hydra-style composer for the reference's config tree.

hydra / omegaconf are not installed in the target image, and only a thin slice of them is needed
to load `phc/data/cfg/**` *unchanged*: a `defaults:` list of groups, `key=value` / `group=name`
command-line overrides, and `${a.b}` interpolation.  The result is an EasyDict-like tree, as
`phc/run_hydra.py:273` produces.
"""
import ast
import copy
import os
import re

import yaml

from . import cfg_defaults


class _Loader(yaml.SafeLoader):
    """SafeLoader + omegaconf's float rule: `2e-5` (no dot) is a float, as hydra parses the reference yamls."""


_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"""^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
                    |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
                    |\.[0-9_]+(?:[eE][-+][0-9]+)?
                    |[-+]?\.(?:inf|Inf|INF)
                    |\.(?:nan|NaN|NAN))$""", re.X),
    list("-+0123456789."))


def _yaml_load(f):
    return yaml.load(f, Loader=_Loader)


class EasyDict(dict):
    """Attribute-style dict (the reference uses the `easydict` package)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, _wrap(v))

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __deepcopy__(self, memo):
        return EasyDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, EasyDict):
        return EasyDict(v)
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    return v


def _parse_value(s):
    try:
        return yaml.load(s, Loader=_Loader)
    except Exception:
        try:
            return ast.literal_eval(s)
        except Exception:
            return s


def _set_path(d, path, value):
    keys = path.split(".")
    for k in keys[:-1]:
        if k not in d or not isinstance(d[k], dict):
            d[k] = {}
        d = d[k]
    d[keys[-1]] = value


def _get_path(d, path):
    for k in path.split("."):
        d = d[k]
    return d


_INTERP = re.compile(r"\$\{([^}]+)\}")


def _resolve(node, root):
    if isinstance(node, dict):
        for k in list(node.keys()):
            node[k] = _resolve(node[k], root)
        return node
    if isinstance(node, list):
        return [_resolve(x, root) for x in node]
    if isinstance(node, str):
        m = _INTERP.fullmatch(node)
        if m:
            return _resolve(copy.deepcopy(_get_path(root, m.group(1))), root)
        return _INTERP.sub(lambda mm: str(_get_path(root, mm.group(1))), node)
    return node


def compose(overrides=(), config_name="config", cfg_dir=None):
    """Compose `config.yaml` + its defaults groups + overrides -> EasyDict.

    `cfg_dir` (or $PHC_CFG_DIR): the reference's unchanged `phc/data/cfg` tree; without it the
    built-in groups of phc_amd/cfg_defaults.py are used.

    `overrides` follow the hydra syntax the reference documents (README "python phc/run_hydra.py
    learning=im_big env=env_im_pnn env.num_envs=4096 ..."): `group=name` selects a yaml in that
    group, `a.b.c=value` sets a leaf, `+a.b=value` adds one.
    """
    cfg_dir = cfg_dir or os.environ.get("PHC_CFG_DIR")
    if cfg_dir:
        with open(os.path.join(cfg_dir, config_name + ".yaml")) as f:
            root = _yaml_load(f)
        defaults = root.pop("defaults", [])
        groups = {}
        for d in defaults:
            if isinstance(d, dict):
                groups.update(d)
    else:
        root = copy.deepcopy(cfg_defaults.ROOT)
        groups = dict(cfg_defaults.DEFAULTS)
    leaf = []
    for ov in overrides:
        k, _, v = ov.partition("=")
        k = k.lstrip("+")
        if k in groups and "." not in k:
            groups[k] = v
        else:
            leaf.append((k, _parse_value(v)))
    cfg = {}
    for g, name in groups.items():
        if name is None:
            continue
        if cfg_dir:
            with open(os.path.join(cfg_dir, g, str(name) + ".yaml")) as f:
                cfg[g] = _yaml_load(f) or {}
        else:
            cfg[g] = cfg_defaults.builtin_group(g, str(name))
    cfg.update(root)
    cfg.pop("hydra", None)
    for k, v in leaf:
        _set_path(cfg, k, v)
    cfg = _resolve(cfg, cfg)
    return EasyDict(cfg)
