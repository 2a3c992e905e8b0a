"""Attributes computed on the device and downloaded on first read.

The reference's processors keep their last outputs as attributes (``rng_dop_resp_raw``, ``rng_dop_resp``, the detectors'
``thresholds`` / ``noise_estimates`` / ``detections``) because the plugin host reads SOME of them through ``view_keys``
(visualization/backends/processor_registry.py:92,114,159,192).  Downloading and converting all of them on every frame cost
more than the kernels (a 3 MB complex64 cube -> 6 MB complex128, three float64 planes): here ``process()`` arms a thunk per
attribute and the first read runs it; a plain assignment (``reset()`` sets ``None``) replaces it.  Reads still return
ndarrays of the reference's dtypes."""
from __future__ import annotations


class LazyAttrs:
    def _lazy_set(self, name, thunk):
        self.__dict__.setdefault("_lazy", {})[name] = thunk
        self.__dict__.pop(name, None)           # so that the next read falls through to __getattr__

    def _lazy_clear(self, *names):
        lazy = self.__dict__.get("_lazy", {})
        for n in names:
            lazy.pop(n, None)

    def _lazy_pending(self, name) -> bool:
        return name in self.__dict__.get("_lazy", {})

    def __getattr__(self, name):                # only reached when the instance has no such attribute
        lazy = self.__dict__.get("_lazy")
        if lazy and name in lazy:
            value = lazy.pop(name)()
            self.__dict__[name] = value
            return value
        raise AttributeError(f"{type(self).__name__!s} object has no attribute {name!r}")

    def __setattr__(self, name, value):
        lazy = self.__dict__.get("_lazy")
        if lazy:
            lazy.pop(name, None)
        object.__setattr__(self, name, value)
