"""TEST INFRASTRUCTURE ONLY — a stand-in for the few pieces of `torch` that bench.py and the torch-based GPU tests use (device
buffers, streams, events), backed by numpy, for REHEARSALS on the wave64 emulator of tests/emu/ (where "device memory" is host
memory and every launch completes before the call returns).  Put on PYTHONPATH by scripts/emu_rehearse.py only; a machine with a GPU
never sees it."""
import time as _time
import types as _types

import numpy as _np

float64, float32, int32, int64, uint8 = _np.float64, _np.float32, _np.int32, _np.int64, _np.uint8
__version__ = "0.0-emulator-rehearsal"


class device:
    def __init__(self, kind, index=0):
        self.type, self.index = kind, index


class Tensor:
    def __init__(self, a):
        self.a = a

    shape = property(lambda self: self.a.shape)
    dtype = property(lambda self: self.a.dtype.type)

    def data_ptr(self):
        return self.a.ctypes.data

    def to(self, dev):
        return Tensor(_np.array(self.a, copy=True))

    def cpu(self):
        return self

    def numpy(self):
        return self.a

    def clone(self):
        return Tensor(self.a.copy())

    def contiguous(self):
        return Tensor(_np.ascontiguousarray(self.a))

    def copy_(self, other):
        _np.copyto(self.a, other.a)
        return self

    def view(self, *arg):
        if len(arg) == 1 and isinstance(arg[0], type):
            return Tensor(self.a.view(arg[0]))
        return Tensor(self.a.reshape(*arg))

    def __getitem__(self, k):
        return Tensor(self.a[k])

    def __len__(self):
        return len(self.a)


def from_numpy(a):
    return Tensor(a)


def _mk(fn, shape, dtype=float64, device=None):
    return Tensor(fn(tuple(shape) if isinstance(shape, (tuple, list)) else (shape,), dtype=dtype))


def empty(*shape, dtype=float64, device=None):
    shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
    return _mk(lambda s, dtype: _np.full(s, 0x7f if _np.dtype(dtype).kind in "iu" else _np.nan, dtype=dtype), shape, dtype)


def zeros(*shape, dtype=float64, device=None):
    shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
    return _mk(_np.zeros, shape, dtype)


def empty_like(t):
    return empty(t.a.shape, dtype=t.a.dtype.type)


def zeros_like(t):
    return zeros(t.a.shape, dtype=t.a.dtype.type)


def equal(a, b):
    return bool(a.a.shape == b.a.shape and (a.a.view(_np.uint8) == b.a.view(_np.uint8)).all())


class _Stream:
    cuda_stream = 0

    def __init__(self, *a, **k):
        pass


class _Event:
    def __init__(self, enable_timing=False):
        self.t = 0.0

    def record(self, stream=None):
        self.t = _time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def _props(i=0):
    # (one emulated device per local rank: a multi-rank rehearsal must show distinct devices, as a real node does)
    return _types.SimpleNamespace(name="wave64 emulator (rehearsal)", gcnArchName="x86-emulated-gfx950", multi_processor_count=1, pci_domain_id=0,
                                  pci_bus_id=int(i), pci_device_id=0, uuid="emu-%d" % int(i))


def tensor(data, dtype=float64, device=None):
    return Tensor(_np.array(data, dtype=dtype))


Tensor.item = lambda self: self.a.reshape(-1)[0].item()


cuda = _types.SimpleNamespace(is_available=lambda: True, init=lambda: None, set_device=lambda i: None, synchronize=lambda dev=None: None,
                              current_stream=lambda dev=None: _Stream(), Stream=_Stream, Event=_Event, get_device_properties=_props,
                              device_count=lambda: 1)
