def _no(*a, **k):
    raise NotImplementedError("pygho is not emulated: use ocn_amd.utils.get_cn1_cn2 for the walk-count route")


spsphadamard = spspmm = _no
