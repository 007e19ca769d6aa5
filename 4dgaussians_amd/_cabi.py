"""ctypes binding of libgs4d.so (include/gs4d.h).  PyTorch is used only for device memory
and streams; every tensor crosses the boundary as a raw device pointer.

The library must be present: there is NO fallback path.  A missing or unloadable
libgs4d.so raises at import of the operators that need it.
"""
from __future__ import annotations

import atexit
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("GS4D_LIB_NAME", "libgs4d.so"))      # GS4D_LIB_NAME: kernel experiments only

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


class StateView(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "depths", "clamped", "means2D", "cov3D", "conic_opacity", "rgb", "tiles_touched", "point_offsets",
        "keys_unsorted", "values_unsorted", "keys_sorted", "point_list", "ranges", "final_T", "n_contrib")]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C 4dgaussians_amd/csrc` "
                "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU/PyTorch fallback.")
        L = C.CDLL(LIB_PATH)
        L.gs4d_last_error.restype = C.c_char_p
        for name in ("gs4d_geometry_bytes", "gs4d_binning_bytes", "gs4d_image_bytes", "gs4d_backward_scratch_bytes",
                     "gs4d_l1_loss_scratch_bytes"):
            getattr(L, name).restype = C.c_size_t
        L.gs4d_thread_state_release.restype = None
        L.gs4d_mailbox_stats.restype = None
        L.gs4d_forward_stats.restype = None
        atexit.register(L.gs4d_thread_state_release)       # the main thread's pinned num_rendered mailbox
        _lib = L
    return _lib


def mailbox_stats():
    """(fast, slow): forwards of this thread whose num_rendered arrived through the mailbox / the copy + sync path."""
    fast, slow = C.c_ulonglong(0), C.c_ulonglong(0)
    lib().gs4d_mailbox_stats(C.byref(fast), C.byref(slow))
    return fast.value, slow.value


def forward_stats():
    """(fast, slow, regrow): mailbox_stats() plus the forwards whose speculative binning chunk had to be requested again."""
    v = [C.c_ulonglong(0) for _ in range(3)]
    lib().gs4d_forward_stats(*(C.byref(x) for x in v))
    return tuple(x.value for x in v)


def speculation_stats():
    """(issued, reissued): gs4d_render_forward calls of this thread whose binning + blend went into the stream before
    num_rendered had been read, and how many of those were issued again because the count exceeded the capacity."""
    v = [C.c_ulonglong(0) for _ in range(2)]
    lib().gs4d_speculation_stats(*(C.byref(x) for x in v))
    return tuple(x.value for x in v)


def binning_stats():
    """(direct, passes): forwards of this thread binned by direct placement / by the radix passes on the tile bits
    (include/gs4d.h: gs4d_binning_stats)."""
    return binning_stats_full()[:2]


def binning_stats_full():
    """(direct, passes, last_adds, last_rendered) of gs4d_binning_stats."""
    v = [C.c_ulonglong(0) for _ in range(4)]
    lib().gs4d_binning_stats(*(C.byref(x) for x in v))
    return tuple(x.value for x in v)


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().gs4d_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor, or NULL for None / empty tensors (the reference passes empty
    CPU tensors for absent inputs, RAST/diff_gaussian_rasterization/__init__.py:198-208)."""
    if t is None or t.numel() == 0:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


def current_stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_live_allocators = {}          # key -> TensorAllocator, only while its C call is in flight
_next_key = [1]


def _alloc_trampoline(ctx, nbytes):
    a = _live_allocators.get(ctx)
    if a is None:
        return 0
    try:                        # never let an exception cross the C boundary
        a._tensor = torch.empty(int(nbytes), dtype=torch.uint8, device=a.device)
        return a._tensor.data_ptr()
    except Exception as ex:
        a.error = ex
        return 0


_TRAMPOLINE = ALLOC_FN(_alloc_trampoline)


class TensorAllocator:
    """gs4d_alloc_fn backed by a uint8 torch tensor (the counterpart of the reference's resizeFunctional
    lambdas, RAST/rasterize_points.cu:27-33).  One module-level ctypes thunk serves every call and the
    allocator is found through its integer context, so no per-call closure / reference cycle is created
    (cycles kept the scratch tensors alive until the garbage collector ran and cost ~10 % of a step).
    Pass `.callback` and `.ctx` to the C entry point and call close() when it has returned."""
    __slots__ = ("device", "_tensor", "error", "_key")

    def __init__(self, device):
        self.device, self._tensor, self.error = device, None, None
        self._key = _next_key[0]
        _next_key[0] += 1
        _live_allocators[self._key] = self

    @property
    def callback(self):
        return _TRAMPOLINE

    @property
    def ctx(self):
        return C.c_void_p(self._key)

    @property
    def tensor(self):
        if self._tensor is None:
            self._tensor = torch.empty(0, dtype=torch.uint8, device=self.device)
        return self._tensor

    def close(self):
        _live_allocators.pop(self._key, None)
        if self.error is not None:
            raise self.error


def _arena_trampoline(ctx, nbytes):
    a = _live_allocators.get(ctx)
    if a is None:
        return 0
    try:
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=a.device)
        a.tensors.append(t)
        return t.data_ptr()
    except Exception as ex:
        a.error = ex
        return 0


_ARENA_TRAMPOLINE = ALLOC_FN(_arena_trampoline)


class ArenaAllocator:
    """gs4d_alloc_fn for entry points that ask for several blocks (gs4d_render_forward: activation arena, binning chunk,
    again the binning chunk if the speculative size was short): every block is kept, in request order, in `tensors`."""
    __slots__ = ("device", "tensors", "error", "_key")

    def __init__(self, device):
        self.device, self.tensors, self.error = device, [], None
        self._key = _next_key[0]
        _next_key[0] += 1
        _live_allocators[self._key] = self

    @property
    def callback(self):
        return _ARENA_TRAMPOLINE

    @property
    def ctx(self):
        return C.c_void_p(self._key)

    def close(self):
        _live_allocators.pop(self._key, None)
        if self.error is not None:
            raise self.error


def empty_group(device, *shapes):
    """Uninitialised float32 tensors carved out of one allocation (see zeros_group)."""
    return zeros_group(device, *shapes, _fill=False)


def zeros_group(device, *shapes, dtypes=None, _fill=True):
    """Several zero-initialised 4-byte-element tensors carved out of ONE allocation cleared by ONE
    fill (each a 256-byte aligned view): dozens of separate torch.zeros calls cost more host and
    launch-gap time per training step than the kernels that consume them."""
    counts = []
    for shp in shapes:
        n = 1
        for d in shp:
            n *= int(d)
        counts.append(n)
    offs, total = [], 0
    for n in counts:
        offs.append(total)
        total += (n + 63) // 64 * 64
    flat = (torch.zeros if _fill else torch.empty)(max(total, 1), dtype=torch.float32, device=device)
    out = []
    for i, (shp, n, o) in enumerate(zip(shapes, counts, offs)):
        t = flat[o:o + n]
        if dtypes is not None and dtypes[i] is not torch.float32:
            t = t.view(dtypes[i])
        out.append(t.view(*shp))
    return out
