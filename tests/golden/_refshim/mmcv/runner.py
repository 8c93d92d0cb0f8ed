"""Build-owned import stub (modules/swin.py:8 imports it; only `init_weights` would call it)."""


def load_checkpoint(*args, **kwargs):
    raise RuntimeError("mmcv is not installed: checkpoint loading through mmcv is not available in the build container")
