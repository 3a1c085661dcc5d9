"""Single import point for gym: the real package when importable, the in-repo shim otherwise."""
try:  # pragma: no cover - depends on the environment
    import gym  # type: ignore
    IS_SHIM = False
except ImportError:
    from pdegym._compat import gym_shim as gym
    IS_SHIM = True

__all__ = ["gym", "IS_SHIM"]
