"""``munchify`` when munch is importable, otherwise a small attribute-dict equivalent."""
try:  # pragma: no cover
    from munch import munchify  # type: ignore
except ImportError:
    class AttrDict(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    def munchify(obj):
        if isinstance(obj, dict):
            return AttrDict({k: munchify(v) for k, v in obj.items()})
        if isinstance(obj, (list, tuple)):
            return type(obj)(munchify(v) for v in obj)
        return obj
