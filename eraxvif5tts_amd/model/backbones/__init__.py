from .dit import DiT  # noqa: F401
