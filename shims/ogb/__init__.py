from _shimguard import shadowing as _shadowing

_shadowing("ogb")
