"""Guards of the stand-in packages in this directory: a stand-in that shadows an INSTALLED package of the same
name says so loudly, and synthetic data is never served without an explicit opt-in."""
import importlib.machinery
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_warned = set()


def shadowing(name: str) -> None:
    """Warn once (stderr) when the real `name` is importable from another sys.path entry: with this directory in front
    the drivers get the minimal stand-in, not the installed package."""
    if name in _warned:
        return
    _warned.add(name)
    others = [p for p in sys.path if p and os.path.isdir(p) and not os.path.samefile(p, _HERE)]
    try:
        spec = importlib.machinery.PathFinder.find_spec(name, others)
    except (ImportError, ValueError):
        spec = None
    if spec is not None and os.environ.get("OCN_SHIMS_QUIET") != "1":
        print(f"[ocn shims] WARNING: '{name}' is installed ({spec.origin}) but PYTHONPATH puts the ocn_amd stand-in "
              f"({_HERE}/{name}) in front of it: the driver runs on the stand-in's minimal surface, not on the real package.",
              file=sys.stderr, flush=True)


def require_synth(what: str) -> None:
    if os.environ.get("OCN_SYNTH") != "1":
        raise RuntimeError(
            f"{what}: this stand-in has no access to the real dataset (no network, no OGB / Planetoid files).  It can serve a "
            "SEEDED SYNTHETIC graph of the named dataset's shape instead — metrics printed on it (Hits@K, MRR) are NOT results "
            "on the real dataset.  Opt in explicitly with OCN_SYNTH=1 (OCN_SYNTH_SCALE=<0..1> shrinks the graph).")
