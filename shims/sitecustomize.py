"""Picked up by Python at start-up when this directory is on PYTHONPATH: if tensorboard is not installed, register a
no-op `torch.utils.tensorboard.SummaryWriter` so that NeighborOverlap_large.py:13,260-261,323 runs unchanged."""
import importlib.util
import sys
import types

if importlib.util.find_spec("tensorboard") is None:
    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None

    _m = types.ModuleType("torch.utils.tensorboard")
    _m.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = _m
