"""YAML options with attribute access — the schema is the reference's (opt/*/*.yaml,
utils/Typing.py), loaded without OmegaConf.  A missing key raises AttributeError (so deepcopy and
hasattr behave), dotted overrides are applied with set_by_path."""
import copy

import yaml


class Opt(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        try:
            del self[k]
        except KeyError:
            raise AttributeError(k)

    def __deepcopy__(self, memo):
        return Opt({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_opt(o):
    if isinstance(o, dict):
        return Opt({k: to_opt(v) for k, v in o.items()})
    if isinstance(o, list):
        return [to_opt(v) for v in o]
    return o


def to_plain(o):
    if isinstance(o, dict):
        return {k: to_plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [to_plain(v) for v in o]
    return o


def load(path):
    with open(path) as f:
        return to_opt(yaml.safe_load(f))


def save(opt, path):
    with open(path, "w") as f:
        yaml.safe_dump(to_plain(opt), f, sort_keys=False)


def set_by_path(opt, dotted, value):
    keys = dotted.split(".")
    node = opt
    for k in keys[:-1]:
        node = node[k]
    node[keys[-1]] = value
    return opt


def merge(base, override):
    """OmegaConf.merge for plain trees: dictionaries are merged key by key, everything else is replaced (main.py:568-569)"""
    out = to_opt(to_plain(base))
    for k, v in to_plain(override).items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = merge(out[k], v)
        else:
            out[k] = to_opt(v)
    return out
