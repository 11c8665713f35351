"""`import spz` — the reference's module name (src/python/spz/__init__.py:1-2) for code written against it; everything
is spz_amd.spz (the MI355X-native implementation; no CPU fallback)."""
from spz_amd.spz import *  # noqa: F401,F403
from spz_amd import spz as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
