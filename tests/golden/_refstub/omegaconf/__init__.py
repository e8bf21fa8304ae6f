"""Test-only stand-in for `omegaconf`, used ONLY by tests/golden/make_golden.py.

The reference (LGAR-py) imports `omegaconf.DictConfig` purely for type annotations;
hydra/omegaconf are not installed in the build container.  This stub lets the golden
generator import the reference from /root/reference to produce numeric fixtures.
It is never imported by the product package, by bench.py or by the -m gpu tests.
"""


class DictConfig(dict):
    """dict with attribute access, recursively wrapping nested dicts."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, DictConfig):
            return DictConfig(v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v
