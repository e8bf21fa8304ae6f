"""Config plumbing: the cfg object the reference's dpLGAR(cfg) consumes, without Hydra/OmegaConf.

Key names are the reference's (dpLGAR/config.yaml, dpLGAR/data/config/*.yaml, dpLGAR/models/config/*.yaml);
`derive_time_keys` applies the agent's derivations (dpLGAR/agents/DifferentiableLGAR.py:35-52).
An OmegaConf DictConfig works too: only attribute/item access and assignment are used.
"""
import os

import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
CONFIG_DIR = os.path.join(HERE, "config")


class Config(dict):
    """dict with attribute access; nested dicts are wrapped."""

    def __init__(self, *a, **kw):
        super().__init__()
        for k, v in dict(*a, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, Config):
            v = Config(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _interpolate(node, root):
    """Resolve ${cwd}-style references (the only interpolation the reference's data configs use)."""
    if isinstance(node, dict):
        for k in list(node):
            node[k] = _interpolate(node[k], root)
    elif isinstance(node, list):
        return [_interpolate(v, root) for v in node]
    elif isinstance(node, str) and "${" in node:
        for key in ("cwd", "data_dir"):
            if key in root and isinstance(root[key], str):
                node = node.replace("${%s}" % key, root[key])
    return node


def derive_time_keys(cfg):
    """agents/DifferentiableLGAR.py:35-52: endtime_s, subcycle_length_h, forcing_resolution_h, time_per_step,
    nsteps, num_subcycles."""
    m, c = cfg.models, cfg.conversions
    m.endtime_s = m.endtime * c.hr_to_sec
    m.subcycle_length_h = m.subcycle_length * (1 / c.hr_to_sec)
    m.forcing_resolution_h = m.forcing_resolution / c.hr_to_sec
    m.time_per_step = m.forcing_resolution_h * c.hr_to_sec
    m.nsteps = int(m.endtime_s / m.time_per_step)
    m.num_subcycles = int(m.forcing_resolution_h / m.subcycle_length_h)
    return cfg


def load_config(data="Phillipsburg", models="shorter_subcycle", cwd=None, config_dir=CONFIG_DIR, overrides=None):
    """Compose root + data + models YAMLs the way `python -m dpLGAR data=config/<data> models=config/<models>` does."""
    with open(os.path.join(config_dir, "config.yaml")) as f:
        root = yaml.safe_load(f)
    root.pop("defaults", None)
    root.pop("hydra", None)
    for group, name in (("data", data), ("models", models)):
        path = name if os.path.isabs(name) else os.path.join(config_dir, group, name + ".yaml")
        with open(path) as f:
            root[group] = yaml.safe_load(f)
    if cwd is not None:
        root["cwd"] = cwd
    for k, v in (overrides or {}).items():
        node = root
        parts = k.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    _interpolate(root, root)
    for grp in ("data", "models"):
        for k, v in list(root[grp].items()):
            if v == "???":
                root[grp][k] = None
    return derive_time_keys(Config(root))
