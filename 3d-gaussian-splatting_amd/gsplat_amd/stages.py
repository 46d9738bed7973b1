"""Thin Python wrappers over the C-ABI stages (``include/gsplat_amd.h``): one function per entry point
family, tensors in / tensors out, memory from torch's caching allocator, launches on torch's current
stream.  ``rasterization.py`` chains them inside one autograd Function; the parity tests also call them
one at a time."""

from __future__ import annotations

import ctypes
import math
import os
import sys
import threading
import time
import weakref
from typing import Optional, Tuple

import torch

from ._lib import SPLAT_STRIDE, check, lib

TILE_SIZE = 16
BIN_BLOCK = 4096        # GSA_BIN_BLOCK: record slots per block of the fused binning's owner table
DEFAULT_CULL_MODE = 0  # 0: published tile rectangle; 1: + exact ellipse-vs-tile test (rasterization docstring)


class StageTimer:
    """Optional per-stage device timing with events on the launch stream (bench.py's roofline leg).
    Disabled by default: the hot path then records nothing."""

    def __init__(self):
        self.enabled = False
        self.records = []  # (stage name, start event, end event)

    def reset(self, enabled: bool) -> None:
        self.enabled = enabled
        self.records = []

    def summary(self):
        """{stage: (calls, total ms)} -- call after a device synchronize."""
        out = {}
        for name, a, b in self.records:
            c, t = out.get(name, (0, 0.0))
            out[name] = (c + 1, t + a.elapsed_time(b))
        return out


TIMER = StageTimer()


def _call(name: str, fn, *args, on=None) -> None:
    """Launch one C-ABI stage on the current stream; raise on a non-zero return code.  ``on``: the torch stream the caller
    passed to the stage instead of the current one (the timer's events are then recorded there)."""
    held, _GUARD.held = getattr(_GUARD, "held", None), None        # (the tensors behind THIS call's pointers: consumed, never sticky)
    guard = held[-1].device.index if held else None                 # (one of them speaks for all: a stage's tensors share a device)
    if guard is not None and torch.cuda.current_device() != guard:
        raise RuntimeError(f"gsplat_amd: {name} was given tensors on cuda:{guard} while the current device is "
                           f"cuda:{torch.cuda.current_device()}; launches go to the current device's stream -- call "
                           "torch.cuda.set_device(tensor.device) (one process per GPU) or use torch.cuda.device(...)")
    if TIMER.enabled:
        start = torch.cuda.Event(enable_timing=True)
        start.record(on) if on is not None else start.record()
    check(fn(*args), name)
    if TIMER.enabled:
        end = torch.cuda.Event(enable_timing=True)
        end.record(on) if on is not None else end.record()
        TIMER.records.append((name, start, end))
    del held   # only now: the launch is queued, the caching allocator hands the memory out again in stream order


_GUARD = threading.local()   # .held: the tensors behind the pointers of the stage being launched (collected by _ptr while
                             # the arguments of ONE _call are evaluated, consumed by that _call; per host thread)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a tensor argument; keeps the tensor for ``_call``, which refuses a launch whose tensors are not
    on the current device (the stream handed to the library is the CURRENT device's).  (~45 calls per view: kept short.)"""
    if t is None:
        return None
    if t.is_cuda:
        # The tensor may be a temporary (``_ptr(f32c(x, ...))`` of a strided or non-fp32 x makes a copy): it has to outlive
        # the evaluation of the call's other arguments, or the next temporary is given the same block before the launch.
        try:
            _GUARD.held.append(t)
        except AttributeError:   # (first pointer of this call, or of this thread)
            _GUARD.held = [t]
    return t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    # (torch.cuda.current_stream() builds a Stream object through several device look-ups: ~15 us, four times a view)
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"gsplat_amd: `{name}` must live on a HIP device (got {t.device}); "
                           "there is no CPU path in the product")
    if t.dtype is torch.float32 and t.is_contiguous():
        return t.detach()
    return t.detach().to(torch.float32).contiguous()


def tile_grid(width: int, height: int, tile: int = TILE_SIZE) -> Tuple[int, int]:
    return (width + tile - 1) // tile, (height + tile - 1) // tile


# Side of the BINS the fused binning sorts the records into when the caller does not say (rasterization(), the harness):
# 32 -- the four 16 x 16 tiles of a bin walk the bin's list (a compositor wave re-tests every record against its 8 x 8
# quadrant anyway), about half as many records to sort; 16 (GSPLAT_AMD_BIN_TILE=16) -- one list per tile.  The stage
# functions default to 16.
_BIN_TILE_ENV = os.environ.get("GSPLAT_AMD_BIN_TILE")
DEFAULT_BIN_TILE = int(_BIN_TILE_ENV or "32")
if DEFAULT_BIN_TILE not in (TILE_SIZE, 2 * TILE_SIZE, 4 * TILE_SIZE):
    raise ValueError("GSPLAT_AMD_BIN_TILE must be 16, 32 or 64")


def bin_tile_for(width: int, height: int, cull_mode: int) -> int:
    """The bin side rasterization() uses for this view: DEFAULT_BIN_TILE where the fused binning applies, else 16.
    Unless GSPLAT_AMD_BIN_TILE says otherwise, a LARGE frame (more than 4096 bins of 32 x 32) whose 64 x 64 bins fit the
    one-pass placement (gsa_tile_bin16_direct: 2047) -- 3840 x 2160 -- is binned at 64: measured at C5 (5 M gaussians, 4K)
    binning 0.79 -> 0.54 ms with 17.7 M instead of 40.1 M records, the forward sweep +0.04, the view 2.31 -> 2.07 ms in the
    4-view step.  (2560 x 1440 with 1 M gaussians: 0.900 ms per view at 32, 0.913 at 64 -- stays at 32, as 1920 x 1080 does.)"""
    if not (cull_mode == 0 and _BIN_FUSED_DEFAULT):
        return TILE_SIZE
    bt = DEFAULT_BIN_TILE
    if _BIN_TILE_ENV is None and bt == 2 * TILE_SIZE and _BIN_DIRECT and tile_grid(width, height, bt)[0] * \
            tile_grid(width, height, bt)[1] > 4096 and lib.gsa_tile_bin16_direct(width, height, 2 * bt):
        bt = 2 * bt
    return bt if bool(lib.gsa_tile_bins_supported(width, height, bt)) else TILE_SIZE


_PINNED = {}          # (device type, index) -> [free (words, event) slots]
_PINNED_BLOCK = 16    # slots allocated at a time (one pinned [16, 4] int32 tensor)


def _pinned_scalar(dev):
    """Four pinned int32 (record count, min / max depth key, status) + an event for ONE read-back of a view's control words.
    Every ``TileBins`` takes a slot of its own from a per-device free list and gives it back when it is finalized, so any
    number of views may be in flight un-finalized (a deferred-check step, a multi-camera call, several host threads):
    the pool grows by ``_PINNED_BLOCK`` slots whenever it runs dry (r03 had a fixed ring of 16 and raised beyond it).
    A ``TileBins`` that is dropped un-finalized keeps its slot out of circulation (16 bytes): its kernel may still write."""
    free = _PINNED.setdefault((dev.type, dev.index), [])
    if not free:
        block = torch.zeros(_PINNED_BLOCK, 4, dtype=torch.int32, pin_memory=True)
        free.extend((block[i], torch.cuda.Event()) for i in range(_PINNED_BLOCK))
    words, event = free.pop()
    return words, event, free


def _release_slot(slot) -> None:
    """Back to the free list (the control words have been read; nothing on the device will write them again)."""
    words, event, _, free = slot
    if free is not None:
        free.append((words, event))


def key_range_buffer(n: int, dev) -> torch.Tensor:
    """[ceil(n / 64), 4] int32 for the depth keys of ONE view: ``project_fwd(key_range=...)`` writes, per wave of 64 gaussians,
    the (min, max) of the visible keys and two of the keys (a sample); ``bin_tiles(key_range=...)`` hands it to the bucketed
    depth ordering, which builds the depth table of the view at hand from it on the device."""
    return torch.empty((n + 63) // 64, 4, dtype=torch.int32, device=dev)


_DEBUG_BINS = os.environ.get("GSPLAT_AMD_DEBUG_BINS") == "1"
_DEBUG_CALLS, _DEBUG_RANGES = [0], []   # (debug) project_views calls; (call, slab name, pointer, bytes, stream) of the last ones
_TOKEN = [0]


def _next_token() -> int:
    """A 24-bit tag (never 0) for one read-back that a KERNEL writes into a pinned slot (gsa_depth_order_msd): the host
    polls word 3 of the slot until it carries the tag -- no copy and no event on the stream."""
    _TOKEN[0] = _TOKEN[0] % 0xFFFFFF + 1
    return _TOKEN[0]


WAIT_STATS = [0, 0.0]   # read-backs waited for, seconds spent waiting: the host's slack per view (bench.py reports it)
# how often the speculation behind the asynchronous binning failed and a view was binned again with exact sizes:
# the remembered record capacity was too small / the depth ordering's assumptions (key range, bucket capacity) did not hold
REPAIR_STATS = {"views": 0, "capacity_misses": 0, "key_range_repairs": 0}


# which depth ordering the views took: the bucketed one on host-planned equal buckets / on the view's own device-built depth
# table, or the LSD passes (first call of a shape, repair, back-off, populations beyond _MSD_MAX_N) -- bench.py reports them
PATH_STATS = {"bucketed_host_plan": 0, "bucketed_device_table": 0, "lsd": 0}


def last_key_span_octaves(dev, width: int, height: int, cull_mode: int = 0, speculation=None) -> Optional[float]:
    """Octaves of depth between the smallest and the largest visible depth key remembered for this scene shape (the union
    over its recent views): log2(z_max / z_min)."""
    import struct
    spec = speculation if speculation is not None else DEFAULT_SPECULATION
    for bt in (32, 64, 16):
        ent = spec.key_range.get((dev.type, dev.index, width, height, cull_mode, bt))
        if ent is not None and ent[0] <= ent[1]:
            z0 = struct.unpack("f", struct.pack("I", ent[0]))[0]
            z1 = struct.unpack("f", struct.pack("I", ent[1]))[0]
            return round(math.log2(max(z1, 1e-30) / max(z0, 1e-30)), 2)
    return None


def _wait_slot(slot) -> None:
    """Blocks until the slot's control words are there: an event (copied slots) or the kernel-written token."""
    t_in = time.perf_counter()
    try:
        _wait_slot_impl(slot)
    finally:
        WAIT_STATS[0] += 1
        WAIT_STATS[1] += time.perf_counter() - t_in


def _wait_slot_impl(slot) -> None:
    words, event, token = slot[:3]
    if token is None:
        event.synchronize()
        return
    spins, t0 = 0, time.monotonic()
    while ((int(words[3]) >> 8) & 0xFFFFFF) != token:
        spins += 1
        if spins > 64:
            time.sleep(0)   # yield the GIL: other host threads (data loader, collective watchdog) must not starve
        if spins > 200 and time.monotonic() - t0 > 10.0:   # (never seen: the kernel is queued ahead of whatever the
            torch.cuda.synchronize()                       # caller did since; a device fault would surface here)
            if ((int(words[3]) >> 8) & 0xFFFFFF) != token:
                raise RuntimeError("gsplat_amd: the depth ordering never reported its control words")
            return


# project_fwd leaves a 48-B digest of every SH coefficient row for project_bwd, which then never reads the 192-B rows: a
# 1 M-gaussian / 1080p view measured fwd 0.089 + bwd 0.072 ms against 0.073 + 0.101 ms without.  (Module attribute, not an
# environment switch: tests and tools/ set it for A/B.)
_SH_AUX = True


# gsa_project_fwd_parts (developer A/B: GSPLAT_AMD_RECORDS_NOSTAGE=1 -- the records part without LDS row staging;
# GSPLAT_AMD_RECORDS_BLOCKS=k -- its launch capped at k workgroups that walk the gaussians in strides)
PROJECT_GEOMETRY = 1
PROJECT_RECORDS = 2 | (4 if os.environ.get("GSPLAT_AMD_RECORDS_NOSTAGE") == "1" else 0) | \
    (int(os.environ.get("GSPLAT_AMD_RECORDS_BLOCKS", "0")) << 8)
# The projection of a view in two launches (r05): the GEOMETRY part (what the binning reads: 44 B in, 28 B out per gaussian) on
# the caller's stream and the RECORDS part (the splat records with the colours and the coefficient digests: what the sweeps
# read; it streams the 192-B coefficient rows) on a side stream BESIDE the view's binning chain, which is latency-bound and
# leaves the HBM idle.  The caller's stream waits for the records in front of its forward sweep (``records_ready``).
# MEASURED (r05, profiles/r05_projection_split.md) and left OFF: the two do not co-run -- the bucket sort takes 137 of a CU's
# 160 KB of LDS and the placement its wave slots, so the records launch mostly waits its turn: a single view 0.593 -> 0.600 ms
# with the plain launch, 0.586 with it capped to 512 strided workgroups (one 13-KB block beside a bucket sort), the 4-view
# step +1 %; the fused training step lost its two-stream overlap to the third stream.  GSPLAT_AMD_SPLIT_PROJECTION=1 switches
# it on (the parts are the same bits: test_split_projection_parts_equal_the_whole_stage).
_SPLIT_PROJECTION = os.environ.get("GSPLAT_AMD_SPLIT_PROJECTION", "0") == "1"
_PROJ_STREAMS = {}


def projection_stream(dev) -> torch.cuda.Stream:
    """The side stream the records part of a split projection runs on (one per device and process)."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    st = _PROJ_STREAMS.get(key)
    if st is None:
        st = _PROJ_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


def _fork_records(dev):
    """-> the side stream, made to wait for everything queued on the current stream so far (the parameters it reads, and
    whatever still uses the memory the caching allocator has just handed out for its outputs)."""
    side = projection_stream(dev)
    here = torch.cuda.Event()
    here.record()
    side.wait_event(here)
    return side


def project_fwd(means, quats, scales, opacities, colors, sh_degree: int, viewmat, Kmat, width: int, height: int,
                cull_mode: int = 0, return_rects: bool = False, return_aux: bool = False, bin_tile: int = TILE_SIZE,
                key_range: Optional[torch.Tensor] = None, qbox: Optional[torch.Tensor] = None,
                records_ready: Optional[list] = None):
    """-> splat [N,16], radii [N,2] i32, tiles_touched [N] i32, depth_keys [N] (uint32 bits in int32 storage)
    [, tile_rects [N,2] i32: (packed tile rectangle, record count) pairs the fused binning reads, with
    ``return_rects`` -- in tiles of ``bin_tile`` pixels: 16, or 32 for ``bin_tiles(..., bin_tile=32)``]
    [, sh_aux [N,12] f32 or None (sh_degree < 1): the 48-B digest of each coefficient row that lets
    ``project_bwd`` skip the rows, with ``return_aux``].
    ``key_range``: a ``key_range_buffer`` that receives the view's depth-key range (for ``bin_tiles(key_range=...)``).
    ``qbox``: an int32 [N] tensor that receives every gaussian's quadrant bounding box (for ``bin_tiles(qbox=...)``: the
    forward compositor's per-record quadrant masks).
    ``records_ready`` (a list): SPLIT the stage -- ``splat`` and ``sh_aux`` are written by a launch on the projection side
    stream, everything else by a launch on the current stream (which may go straight on to ``bin_tiles``); the list receives
    ONE event, and the caller makes its stream wait for it (``torch.cuda.current_stream().wait_event``) before anything reads
    ``splat`` / ``sh_aux``.  Same bits as the one-launch stage."""
    dev, N = means.device, means.shape[0]
    sh_coeffs = colors.shape[1] if sh_degree >= 0 else 0
    splat = torch.empty(N, SPLAT_STRIDE, dtype=torch.float32, device=dev)
    radii = torch.empty(N, 2, dtype=torch.int32, device=dev)
    tiles_touched = torch.empty(N, dtype=torch.int32, device=dev)
    depth_keys = torch.empty(N, dtype=torch.int32, device=dev)
    tile_rects = torch.empty(N, 2, dtype=torch.int32, device=dev) if return_rects else None
    sh_aux = torch.empty(N, 12, dtype=torch.float32, device=dev) if (return_aux and sh_degree >= 1 and _SH_AUX) else None
    if records_ready is not None and N > 0:
        side = _fork_records(dev)
        _call("gsa_project_fwd_records", lib.gsa_project_fwd_parts, PROJECT_RECORDS, N, _ptr(means), _ptr(quats), _ptr(scales),
              _ptr(opacities), _ptr(colors), sh_degree, sh_coeffs, 1, _ptr(viewmat), _ptr(Kmat), 0, width, height, cull_mode,
              _ptr(splat), None, None, None, None, _ptr(sh_aux), bin_tile, None, None, side.cuda_stream, on=side)
        done = torch.cuda.Event()
        done.record(side)
        records_ready.append(done)
        _call("gsa_project_fwd", lib.gsa_project_fwd_parts, PROJECT_GEOMETRY, N, _ptr(means), _ptr(quats), _ptr(scales),
              _ptr(opacities), None, sh_degree, sh_coeffs, 1, _ptr(viewmat), _ptr(Kmat), 0, width, height, cull_mode,
              None, _ptr(radii), _ptr(tiles_touched), _ptr(depth_keys), _ptr(tile_rects), None, bin_tile, _ptr(key_range),
              _ptr(qbox), _stream())
    else:
        _call("gsa_project_fwd", lib.gsa_project_fwd, N, _ptr(means), _ptr(quats), _ptr(scales), _ptr(opacities),
              _ptr(colors), sh_degree, sh_coeffs, _ptr(viewmat), _ptr(Kmat), width, height, cull_mode, _ptr(splat),
              _ptr(radii), _ptr(tiles_touched), _ptr(depth_keys), _ptr(tile_rects), _ptr(sh_aux), bin_tile, _ptr(key_range),
              _ptr(qbox), _stream())
    out = (splat, radii, tiles_touched, depth_keys)
    if return_rects:
        out += (tile_rects,)
    if return_aux:
        out += (sh_aux,)
    return out


class ViewProjection:
    """The projection of the gaussians for ONE camera, as ``project_fwd`` leaves it (with ``return_rects``, ``return_aux``,
    a key range and quadrant boxes): what the binning, the sweeps and ``project_bwd`` of that view need."""

    __slots__ = ("splat", "radii", "tiles_touched", "depth_keys", "tile_rects", "sh_aux", "key_range", "qbox", "viewmat",
                 "Kmat", "bin_tile", "dims", "defer_backward", "v_splat", "made_from", "camera_stamp", "records_ready", "v_rgb")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))


def input_stamp(*tensors):
    """(storage pointer, version counter, shape) of each tensor: what a projection remembers of the parameters it was made
    from, and what ``rasterization(projection=...)`` compares -- on the host, no device sync -- before it uses it."""
    return tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in tensors)


def project_views(means, quats, scales, opacities, colors, sh_degree: int, viewmats, Kmats, width: int, height: int,
                  cull_mode: int = 0, return_aux: bool = False, bin_tile: Optional[int] = None, split: Optional[bool] = None):
    """The gaussians projected for the C cameras of ``viewmats`` [C,4,4] in ONE launch (``gsa_project_fwd_batch``): geometry
    and coefficient rows are read once for all cameras -- the views of an optimisation step share them.  ``Kmats``: [C,3,3],
    or one [3,3] matrix for all cameras.  -> C ``ViewProjection``s whose tensors are the per-camera slabs of [C, ...]
    buffers; each holds the same bits ``project_fwd`` with that camera writes.
    ``split`` (default: the module's ``_SPLIT_PROJECTION``): the records part (``splat``, ``sh_aux``) on the projection side
    stream, the geometry part on the current stream -- every returned projection then carries ``records_ready``, the event a
    stream waits for before it reads those two (``rasterization(projection=...)`` and the harness do)."""
    dev, N = means.device, means.shape[0]
    C = int(viewmats.shape[0])
    if viewmats.shape != (C, 4, 4) or Kmats.shape not in ((C, 3, 3), (3, 3)):
        raise ValueError("project_views: viewmats must be [C,4,4] and Kmats [C,3,3] or [3,3]")
    bt = bin_tile_for(width, height, cull_mode) if bin_tile is None else bin_tile
    sh_coeffs = colors.shape[1] if sh_degree >= 0 else 0
    splat = torch.empty(C, N, SPLAT_STRIDE, dtype=torch.float32, device=dev)
    radii = torch.empty(C, N, 2, dtype=torch.int32, device=dev)
    tiles_touched = torch.empty(C, N, dtype=torch.int32, device=dev)
    depth_keys = torch.empty(C, N, dtype=torch.int32, device=dev)
    tile_rects = torch.empty(C, N, 2, dtype=torch.int32, device=dev)
    sh_aux = torch.empty(C, N, 12, dtype=torch.float32, device=dev) if (return_aux and sh_degree >= 1 and _SH_AUX) else None
    key_range = torch.empty(C, (N + 63) // 64, 4, dtype=torch.int32, device=dev)
    qbox = torch.empty(C, N, dtype=torch.int32, device=dev) if _QUAD_BOXES else None
    vm, km = f32c(viewmats, "viewmats"), f32c(Kmats, "Ks")
    stride = 9 if km.dim() == 3 else 0
    done = None
    if _DEBUG_BINS:   # (diagnostics: a slab that still holds this pattern when it is read was never written by the launch below)
        tile_rects.fill_(0x5EADBEEF)
        depth_keys.fill_(0x5EADBEEF)
    if (_SPLIT_PROJECTION if split is None else split) and N > 0:
        side = _fork_records(dev)
        _call("gsa_project_fwd_records", lib.gsa_project_fwd_parts, PROJECT_RECORDS, N, _ptr(means), _ptr(quats), _ptr(scales),
              _ptr(opacities), _ptr(colors), sh_degree, sh_coeffs, C, _ptr(vm), _ptr(km), stride, width, height, cull_mode,
              _ptr(splat), None, None, None, None, _ptr(sh_aux), bt, None, None, side.cuda_stream, on=side)
        done = torch.cuda.Event()
        done.record(side)
        _call("gsa_project_fwd", lib.gsa_project_fwd_parts, PROJECT_GEOMETRY, N, _ptr(means), _ptr(quats), _ptr(scales),
              _ptr(opacities), None, sh_degree, sh_coeffs, C, _ptr(vm), _ptr(km), stride, width, height, cull_mode,
              None, _ptr(radii), _ptr(tiles_touched), _ptr(depth_keys), _ptr(tile_rects), None, bt, _ptr(key_range),
              _ptr(qbox), _stream())
    else:
        _call("gsa_project_fwd", lib.gsa_project_fwd_batch, N, _ptr(means), _ptr(quats), _ptr(scales), _ptr(opacities),
              _ptr(colors), sh_degree, sh_coeffs, C, _ptr(vm), _ptr(km), stride, width, height, cull_mode,
              _ptr(splat), _ptr(radii), _ptr(tiles_touched), _ptr(depth_keys), _ptr(tile_rects), _ptr(sh_aux), bt,
              _ptr(key_range), _ptr(qbox), _stream())
    if _DEBUG_BINS and N > 0:   # (diagnostics, no synchronisation: where this call's slabs lie, for the overlap search below)
        _DEBUG_CALLS[0] += 1
        for name, t in (("splat", splat), ("radii", radii), ("tiles_touched", tiles_touched), ("depth_keys", depth_keys),
                        ("tile_rects", tile_rects), ("sh_aux", sh_aux), ("key_range", key_range), ("qbox", qbox)):
            if t is not None:
                _DEBUG_RANGES.append((_DEBUG_CALLS[0], name, t.data_ptr(), t.numel() * t.element_size(),
                                      torch.cuda.current_stream().cuda_stream))
        del _DEBUG_RANGES[:-64]
    return [ViewProjection(splat=splat[c], radii=radii[c], tiles_touched=tiles_touched[c], depth_keys=depth_keys[c],
                           tile_rects=tile_rects[c], sh_aux=None if sh_aux is None else sh_aux[c], key_range=key_range[c],
                           qbox=None if qbox is None else qbox[c], viewmat=vm[c], Kmat=km[c] if km.dim() == 3 else km,
                           bin_tile=bt, dims=(width, height, cull_mode, sh_degree), defer_backward=False, v_splat=None,
                           made_from=None, camera_stamp=input_stamp(vm[c], km[c] if km.dim() == 3 else km),
                           records_ready=done, v_rgb=None)
            for c in range(C)]


def project_views_bwd(means, quats, scales, opacities, colors, sh_degree: int, views, v_splats, width: int, height: int,
                      out, accumulate: bool = False, v_rgbs=None) -> bool:
    """The backward projection of ALL of ``views`` (``ViewProjection``s of one ``project_views`` call, or any per-camera
    projections with digests) in ONE launch (``gsa_project_bwd_batch``): ``v_splats[c]`` is camera c's gradient record from
    ``composite_bwd``; ``out`` = (v_means, v_quats, v_scales, v_opacities, v_colors) receives the gradients summed over the
    cameras -- written once (``accumulate``: added).  -> False (nothing launched) where the batched kernel does not apply
    (no digests: SH degree < 1; more than 8 cameras; coefficient rows that are not [N,16,3]): the caller then runs
    ``project_bwd`` per camera.
    ``v_rgbs`` (one [N,3] tensor per camera: what ``composite_bwd(v_rgb_out=...)`` left): the GEOMETRY-only launch of a
    view-parallel rank (``gsa_project_bwd_batch_geometry``) -- ``out[4]`` is not written (may be None): the coefficient
    gradient of all views is rebuilt from the gathered colour gradients (``sh_grad_expand``)."""
    C = len(views)
    geom = v_rgbs is not None
    if C < 1 or C > 8 or sh_degree < 1 or any(v.sh_aux is None for v in views) or \
            (not geom and (colors.dim() != 3 or colors.shape[1] != 16)):
        return False
    N = means.shape[0]
    v_means, v_quats, v_scales, v_opac, v_colors = out
    for got, like in zip(out[:4] if geom else out, (means, quats, scales, opacities, colors)):
        if got.numel() != like.numel() or got.dtype != torch.float32 or not got.is_contiguous() or got.device != like.device:
            raise ValueError("project_views_bwd: out tensors must be contiguous fp32, sized like the inputs, on their device")
    if geom and (len(v_rgbs) != C or any(t.shape != (N, 3) or t.dtype != torch.float32 or not t.is_contiguous() for t in v_rgbs)):
        raise ValueError("project_views_bwd: v_rgbs must be one contiguous fp32 [N,3] tensor per camera")
    vms = torch.stack([v.viewmat for v in views]).contiguous()
    shared_k = all(v.Kmat is views[0].Kmat or v.Kmat.data_ptr() == views[0].Kmat.data_ptr() for v in views)
    kms = views[0].Kmat.contiguous() if shared_k else torch.stack([v.Kmat for v in views]).contiguous()
    keep = [f32c(v, "v_splat") for v in v_splats]
    # The cameras' rows were allocated on the streams their views ran on and are read HERE, on the current stream, typically
    # released by the caller right behind this call: the allocator must not hand their blocks out again (in their own
    # streams' order) before this launch has finished with them.
    here = torch.cuda.current_stream()
    for t in keep:
        t.record_stream(here)
    for v in views:
        v.radii.record_stream(here)
        v.sh_aux.record_stream(here)
    arr = ctypes.c_void_p * C
    radii = arr(*[v.radii.data_ptr() for v in views])
    vsp = arr(*[t.data_ptr() for t in keep])
    aux = arr(*[v.sh_aux.data_ptr() for v in views])
    if geom:
        for t in v_rgbs:
            t.record_stream(here)
        rgb = arr(*[t.data_ptr() for t in v_rgbs])
        _call("gsa_project_bwd", lib.gsa_project_bwd_batch_geometry, N, _ptr(means), _ptr(quats), _ptr(scales), _ptr(opacities),
              sh_degree, C, _ptr(vms), _ptr(kms), 0 if shared_k else 9, width, height, ctypes.cast(radii, ctypes.c_void_p),
              ctypes.cast(vsp, ctypes.c_void_p), ctypes.cast(aux, ctypes.c_void_p), ctypes.cast(rgb, ctypes.c_void_p),
              _ptr(v_means), _ptr(v_quats), _ptr(v_scales), _ptr(v_opac), 1 if accumulate else 0, _stream())
        del v_rgbs
        return True
    _call("gsa_project_bwd", lib.gsa_project_bwd_batch, N, _ptr(means), _ptr(quats), _ptr(scales), _ptr(opacities), sh_degree,
          int(colors.shape[1]), C, _ptr(vms), _ptr(kms), 0 if shared_k else 9, width, height,
          ctypes.cast(radii, ctypes.c_void_p), ctypes.cast(vsp, ctypes.c_void_p), ctypes.cast(aux, ctypes.c_void_p),
          _ptr(v_means), _ptr(v_quats), _ptr(v_scales), _ptr(v_opac), _ptr(v_colors), 1 if accumulate else 0, _stream())
    return True


def sort_pairs(n: int, keys: torch.Tensor, vals: Optional[torch.Tensor], begin_bit: int, end_bit: int):
    """Stable radix sort of int32-typed storage holding uint32 keys; returns (sorted_keys, sorted_vals)."""
    dev = keys.device
    k_out = torch.empty(n, dtype=torch.int32, device=dev)
    v_out = torch.empty(n, dtype=torch.int32, device=dev)
    k_tmp = torch.empty(n, dtype=torch.int32, device=dev)
    v_tmp = torch.empty(n, dtype=torch.int32, device=dev)
    ws_bytes = lib.gsa_sort_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _call("gsa_sort_pairs_u32", lib.gsa_sort_pairs_u32, n, _ptr(keys), _ptr(vals), _ptr(k_out), _ptr(v_out),
          _ptr(k_tmp), _ptr(v_tmp), begin_bit, end_bit, _ptr(ws), ws_bytes, _stream())
    return k_out, v_out


def _debug_implausible(bins, ctl_host) -> None:
    """GSPLAT_AMD_DEBUG_BINS=1 (developer aid; profiles/r05_four_rank_rehearsal.md): what the device holds behind an implausible
    read-back once everything queued has run -- the control words, the view's inputs, where its rectangles lie and which other
    projection slabs / live tensor storages overlap them."""
    dbg = getattr(bins, "_debug", None)
    if dbg is not None:   # (GSPLAT_AMD_DEBUG_BINS=1: what the device holds once everything queued has run)
        try:
            torch.cuda.synchronize()
        except Exception as exc:   # noqa: BLE001  (a device fault surfaces here: say which)
            print(f"[gsplat_amd]   device sync raised {type(exc).__name__}: {exc}", file=sys.stderr, flush=True)
            raise
        ctl_d, rects_d, keys_d, offs_d = dbg
        keys64 = keys_d.to(torch.int64) & 0xFFFFFFFF
        cnt = rects_d.view(-1, 2)[:, 1].to(torch.int64) & 0xFFFFFFFF
        print(f"[gsplat_amd]   after a device sync: host words {[int(x) & 0xFFFFFFFF for x in ctl_host]}, device ctl "
              f"{[int(x) & 0xFFFFFFFF for x in ctl_d.tolist()]}; inputs: sum of counts {int(cnt.sum())} (max {int(cnt.max())}), "
              f"keys min {int(keys64.min())} zeros {int((keys64 == 0).sum())} culled {int((keys64 == 0xFFFFFFFF).sum())}; "
              f"offsets max {int((offs_d.to(torch.int64) & 0xFFFFFFFF).max())}; stream {torch.cuda.current_stream()}; "
              f"words still holding the pre-launch pattern: rectangles {int((rects_d == 0x5EADBEEF).sum())} of "
              f"{rects_d.numel()}, keys {int((keys_d == 0x5EADBEEF).sum())} of {keys_d.numel()}; first rectangle words "
              f"{[hex(int(x) & 0xFFFFFFFF) for x in rects_d.view(-1)[:8].tolist()]}", file=sys.stderr, flush=True)
        lo, hi = rects_d.data_ptr(), rects_d.data_ptr() + rects_d.numel() * 4
        mine = [r for r in _DEBUG_RANGES if r[2] <= lo < r[2] + r[3]]
        over = [r for r in _DEBUG_RANGES if r[2] < hi and lo < r[2] + r[3] and r not in mine]
        print(f"[gsplat_amd]   the rectangles lie at {lo:#x}..{hi:#x}; slab(s) they belong to: {mine}; OTHER projection "
              f"slabs overlapping them: {over}; all calls so far: {_DEBUG_CALLS[0]}", file=sys.stderr, flush=True)
        import gc
        seen, owners = set(), []
        for o in gc.get_objects():   # every live device tensor whose STORAGE overlaps the rectangles
            try:
                if torch.is_tensor(o) and o.is_cuda:
                    st_ = o.untyped_storage()
                    p0, nb = st_.data_ptr(), st_.nbytes()
                    if p0 < hi and lo < p0 + nb and (p0, nb) not in seen:
                        seen.add((p0, nb))
                        owners.append((hex(p0), nb, tuple(o.shape), str(o.dtype), sys.getrefcount(o)))
            except Exception:   # noqa: BLE001
                pass
        print(f"[gsplat_amd]   live tensor storages overlapping them: {owners}; allocator: "
              f"{ {k: v for k, v in torch.cuda.memory_stats().items() if k in ('num_alloc_retries', 'num_ooms', 'num_device_free', 'num_device_alloc', 'num_sync_all_streams')} }",
              file=sys.stderr, flush=True)


class TileBins:
    """Result of ``bin_tiles``: record buffers sized for a CAPACITY, the live record count on the device, and
    ``finalize()`` which the caller runs after it has queued the work that consumes the bins."""

    def __init__(self, sorted_tile, sorted_gid, tile_offsets, capacity, m_known, key, slot=None, n=0, key_bits=32,
                 keys_ok=True, boxed=False, speculation=None):
        self._tile, self._gid_t, self._offs_t = sorted_tile, sorted_gid, tile_offsets
        self._chain = None   # (a WEAK reference to the ViewChain whose workspace holds the id buffer and the offsets: views are
                             # built when somebody asks; the chain owns these bins, not the other way round -- no cycle, the
                             # 0.4-GB workspace goes back to the allocator the moment the view's graph is dropped)
        self._spec = speculation if speculation is not None else DEFAULT_SPECULATION   # who remembers what this view teaches
        # boxed: the top byte of every entry of the id buffer is the record's quadrant box (gsa_tile_bin16 given qbox), the
        # gaussian id its low 24 bits -- what the compositor stages must be told (their ``gid_boxes``)
        self.boxed = bool(boxed)
        self.capacity, self._m, self._key, self._slot, self._n = capacity, m_known, key, slot, n
        # what the depth ordering assumed about the visible keys -- bits an LSD sort covered, or (key_min, scale) of the
        # bucketed one; None: nothing to check -- and the result of the check
        self.key_bits, self._keys_ok = key_bits, keys_ok
        # the compositor's launch order (tile_schedule's output), where the one-pass placement wrote it on the side
        self.tile_order = None

    @property
    def _gid(self):
        ch = self._chain() if self._chain is not None else None
        if self._gid_t is None and ch is not None:
            self._gid_t = ch.view("sorted_gid", torch.int32)
        return self._gid_t

    @property
    def tile_offsets(self):
        ch = self._chain() if self._chain is not None else None
        if self._offs_t is None and ch is not None:
            self._offs_t = ch.view("tile_offsets", torch.int32)
        return self._offs_t

    @property
    def bin_tile(self) -> int:   # side of the tiles these bins index (16, or 32: see bin_tiles)
        return self._key[5] if len(self._key) > 5 else TILE_SIZE

    @property
    def sorted_gid_buffer(self):   # what the compositor kernels read (they only follow tile_offsets)
        return self._gid

    def finalize(self):
        """-> (M, ok).  Waits for the control words (record count and depth-key range: written to pinned memory by the
        depth-ordering kernel itself, or copied there right after the scan -- long done by the time the compositor has
        been queued) and checks them against what the
        kernels ran with: the record capacity, and the number of depth-key bits that were sorted."""
        if self._m is None:
            _wait_slot(self._slot)
            ctl_host = self._slot[0]
            self._m = int(ctl_host[0]) & 0xFFFFFFFF
            # A record count no frame of this size can have (more than one record per gaussian and bin): the view's INPUTS were
            # not what its projection wrote.  Seen in r05 with four processes time-slicing one card behind a long-lived parent
            # (profiles/r05_four_rank_rehearsal.md; not explained); such a view is reported as a failed speculation WITHOUT
            # teaching the remembered capacity anything -- loud instead of a silently wrong step.
            w, h, bt = self._key[2], self._key[3], self.bin_tile
            plausible = self._m <= max(self._n, 1) * (tile_grid(w, h, bt)[0] * tile_grid(w, h, bt)[1]) + 1
            if not plausible or os.environ.get("GSPLAT_AMD_DEBUG_SLOTS") == "1":
                print(f"[gsplat_amd] finalize: implausible control words {[int(x) & 0xFFFFFFFF for x in ctl_host]} token={self._slot[2]} "
                      f"key={self._key} n={self._n} capacity={self.capacity} ordering={self.key_bits}", file=sys.stderr, flush=True)
                _debug_implausible(self, ctl_host)
            if plausible:
                _note_count(self._spec, self._key, self._n, self._m)
                if self.key_bits is not None:   # (the unfused path sorts all 32 bits and leaves no range)
                    self._keys_ok = _note_keys(self._spec, self._key, int(ctl_host[1]) & 0xFFFFFFFF, int(ctl_host[2]) & 0xFFFFFFFF,
                                               self.key_bits, int(ctl_host[3]) & 0xFF if isinstance(self.key_bits, tuple) else 0)
            else:
                self._keys_ok = False
            for stats in (REPAIR_STATS, self._spec.repairs):   # (process-wide, and of the object this view speculated on)
                stats["views"] += 1
                if self._m > self.capacity:
                    stats["capacity_misses"] += 1
                elif not self._keys_ok:
                    stats["key_range_repairs"] += 1
            _release_slot(self._slot)
            self._slot = None
        return self._m, self._m <= self.capacity and self._keys_ok

    def __iter__(self):   # sorted_tile, sorted_gid, tile_offsets = bin_tiles(...)
        return iter((self.sorted_tile, self.sorted_gid, self.tile_offsets))

    @property
    def sorted_tile(self):
        m = self.finalize()[0]
        if self._tile is None:   # direct binning writes no ids: a list's records are those between its offsets
            self._tile = tile_ids_from_offsets(self.tile_offsets)[:self.capacity]
        return self._tile[:m]

    @property
    def sorted_tile_stored(self):
        """The ids if the binning path wrote them, else an EMPTY tensor (no work: ``tile_ids_from_offsets`` rebuilds them)."""
        if self._tile is None:
            return torch.empty(0, dtype=torch.int16, device=self._gid.device)
        return self._tile[:self.finalize()[0]]

    @property
    def sorted_gid(self):
        g = self._gid[:self.finalize()[0]]
        return g & 0xFFFFFF if self.boxed else g

    @property
    def sorted_boxes(self):
        """Per record the quadrant box inside its bin as a uint8-valued int32 tensor: bits 0-3 the columns, bits 4-7 the rows of
        the bin's (up to four) units the box covers (units: quadrants; pairs of quadrants in a 64-pixel bin); 0: none --
        None without boxes."""
        return (self._gid[:self.finalize()[0]] >> 24) & 0xFF if self.boxed else None


def tile_ids_from_offsets(tile_offsets: torch.Tensor) -> torch.Tensor:
    """int32 tile id of every record of lists given by their offsets (``TileBins.sorted_tile`` of the direct binning)."""
    counts = (tile_offsets[1:] - tile_offsets[:-1]).long()
    return torch.repeat_interleave(torch.arange(counts.numel(), device=counts.device, dtype=torch.int32), counts)


def tile_ids(sorted_tile: torch.Tensor) -> torch.Tensor:
    """Tile ids of ``TileBins.sorted_tile`` as int64, whichever width the binning path stored them in."""
    t = sorted_tile.long()
    return t & 0xFFFF if sorted_tile.dtype == torch.int16 else t


_BIN_FUSED_DEFAULT = True   # records generated inside the binning passes (False: emit + 32-bit sort; A/B from tests / tools)
# False: the two-pass tile sort where gsa_tile_bin16_direct offers the one-pass placement (A/B from tests / tools)
_BIN_DIRECT = True
# Per-record quadrant BOXES for the forward compositor, which then gathers only the records whose bounding box covers its
# quadrant (of the records of a 32 x 32 bin 22 % reach a given 8 x 8 quadrant; of those the box lets through, 78 %).  The
# boxes come out of the one-pass placement (gsa_tile_bin16_boxes) from the gaussians' quadrant bounding boxes
# (``project_fwd(qbox=...)`` -> ``bin_tiles(qbox=...)``), packed into the top byte of the record's gaussian id: ~15 integer
# instructions per record and no store of their own.  GSPLAT_AMD_QUAD_BOXES=0: plain ids (A/B).
_QUAD_BOXES = os.environ.get("GSPLAT_AMD_QUAD_BOXES", "1") != "0"


def qbox_buffer(n: int, dev) -> torch.Tensor:
    """int32 [n] for ``project_fwd(qbox=...)`` / ``bin_tiles(qbox=...)`` -- None when the boxes are switched off."""
    return torch.empty(n, dtype=torch.int32, device=dev) if _QUAD_BOXES else None


class Speculation:
    """What the asynchronous binning REMEMBERS per scene shape (device, W, H, cull mode, bin side) -- its speculation state:

    * ``capacity``: [records per gaussian the next call provisions for, consecutive low counts].  Keyed WITHOUT N: densify /
      prune changes N every 100 iterations (trainer.py:794-797) and a changed N predicts its capacity from the remembered
      ratio instead of costing a blocking read-back.  Grows at once, shrinks only after many consecutive low counts (the
      views of one batch differ; alternating shrink / miss would re-run the binning).
    * ``key_range``: [lo, hi, consecutive much-narrower views]: the union of the visible depth-key ranges seen so far (grows
      at once, is reset after many consecutive views that need less than half of it).  The depth ordering is launched for
      this range and the view's actual range is checked afterwards, like the record capacity.
    * ``table_mode``: views left on the device-built depth table; ``msd_backoff`` / ``host_plan_backoff`` / ``msd_streak``:
      calls left before the bucketed ordering (on host-planned equal buckets) is tried again after a miss.

    One object per CONSUMER of views (r05; module globals before): a ``HipBackend``, a ``ViewPipeline`` and a trainer own
    theirs, so a trainer and an evaluator that render frames of the same size from different scenes -- or different
    camera sets -- in one process do not widen each other's remembered ranges or trigger each other's repairs; the bare
    ``rasterization(...)`` / stage calls share ``DEFAULT_SPECULATION``."""

    def __init__(self):
        self.capacity, self.key_range, self.table_mode = {}, {}, {}
        self.msd_backoff, self.host_plan_backoff, self.msd_streak = {}, {}, {}
        self.repairs = {"views": 0, "capacity_misses": 0, "key_range_repairs": 0}   # of the views finalized against this object

    def reset(self) -> None:
        """Forget everything: the next call of every shape is a first call (exact sizes, one blocking read-back)."""
        for d in (self.capacity, self.key_range, self.msd_backoff, self.table_mode, self.msd_streak, self.host_plan_backoff):
            d.clear()


DEFAULT_SPECULATION = Speculation()
_CAPACITY_MAX_KEYS = 64
_SHRINK_AFTER = 32
# views of ONE camera after which what is remembered about a scene shape (record capacity, depth-key range) has settled on
# that camera: a benchmark that goes from several cameras to one warms up this long before it times
SPECULATION_SETTLE_VIEWS = _SHRINK_AFTER + 8
_MSD_MAX_WIDTH = int(lib.gsa_depth_order_msd_max_width())
_DEPTH_MSD_DEFAULT = True   # bucketed depth ordering (False: LSD passes; A/B from tests / tools)
# How the bucketed depth ordering gets its buckets.  "host": 255 EQUAL buckets over the union of the key ranges of the views of
# this scene shape seen so far (planned here; 4 launches) -- the fastest when it fits, but one wide view empties the buckets
# of all the others and unevenly distributed depths (the diagonal cameras of the bench ring: 3.5 x the mean in one bucket)
# overflow a bucket.  "device": the view's own DEPTH TABLE, built on the device from what project_fwd leaves per wave (one
# more, single-workgroup launch: +0.012 ms at 1 M gaussians) -- level buckets for any view.  GSPLAT_AMD_DEPTH_TABLE=auto (default):
# a scene shape starts on "host" and moves to "device" for the next _TABLE_VIEWS views when a host-planned ordering has to
# be repaired; =1 / =0: always / never.
_DEPTH_TABLE = os.environ.get("GSPLAT_AMD_DEPTH_TABLE", "auto")
_TABLE_VIEWS = 4096
_MSD_MIN_N = 32768        # below this the 256 x 1024-thread bucket launch costs more than the LSD passes it replaces
_MSD_MAX_N = 1_400_000    # above this an average bucket (N / 255, x1.5 for skew) no longer fits the 8192-key LDS capacity


def reset_speculation() -> None:
    """``DEFAULT_SPECULATION.reset()``: what the bare ``rasterization`` / stage calls of this process remember."""
    DEFAULT_SPECULATION.reset()


def _key_bits_needed(lo: int, hi: int) -> int:
    """Whole 8-bit passes that order keys in [lo, hi] completely: the bits in which the two ends differ."""
    if lo > hi:           # no visible gaussian
        return 8
    return max(8, 8 * (((lo ^ hi).bit_length() + 7) // 8))


def _msd_plan(lo: int, hi: int):
    """-> (key_min, scale) of the bucketed depth ordering for keys in [lo, hi] with 1/16 of the span as margin on both
    sides -- bucket = ((key - key_min) * scale) >> 32, the 255 buckets spread evenly over the range -- or None when a
    bucket would be wider than the 16 bits the local sorts cover."""
    if lo > hi:
        return None
    margin = ((hi - lo) >> 4) + 1
    a, b = max(0, lo - margin), min(0xFFFFFFFE, hi + margin)
    span = max(b - a + 1, 256)
    scale = (255 << 32) // span          # (span >= 256: scale < 2^32)
    if scale <= (1 << 32) // _MSD_MAX_WIDTH:   # (a bucket wider than the local sorts' 24 bits: no float depth range is)
        return None
    return (a, scale)


def _msd_bucket(x: int, scale: int) -> int:
    return (x * scale) >> 32


def _note_keys(spec: Speculation, key, lo: int, hi: int, assumed, status: int = 0) -> bool:
    """Remember the depth-key range of this view; -> whether the ordering that ran was complete.  ``assumed``: the
    number of key bits an LSD sort covered, or (key_min, scale) of a bucketed one."""
    _KEY_RANGE, _MSD_BACKOFF, _MSD_STREAK = spec.key_range, spec.msd_backoff, spec.msd_streak
    _HOST_PLAN_BACKOFF, _TABLE_MODE = spec.host_plan_backoff, spec.table_mode
    ent = _KEY_RANGE.get(key)
    if lo <= hi:
        if ent is None:
            while len(_KEY_RANGE) >= _CAPACITY_MAX_KEYS:
                _KEY_RANGE.pop(next(iter(_KEY_RANGE)))
            _KEY_RANGE[key] = [lo, hi, 0]
        else:
            if lo < ent[0] or hi > ent[1]:
                ent[0], ent[1], ent[2] = min(lo, ent[0]), max(hi, ent[1]), 0
            elif 2 * (hi - lo) < ent[1] - ent[0]:
                ent[2] += 1
                if ent[2] >= _SHRINK_AFTER:
                    ent[0], ent[1], ent[2] = lo, hi, 0
            else:
                ent[2] = 0
    if isinstance(assumed, tuple):
        if len(assumed) >= 3:   # planned on the device (the view's own depth table): only the status can say no
            if status & 4:      # the keys span more than the local sorts cover (no range of float depths does)
                _MSD_BACKOFF[key] = 64
            elif status & 1:    # a bucket overflowed although its table was built from the view's own keys (a cluster of equal
                streak = _MSD_STREAK.get(key, 0)          # depths, or an unlucky sample): back off, longer when it repeats
                _MSD_BACKOFF[key] = min(64, 4 << streak)
                _MSD_STREAK[key] = min(streak + 1, 4)
            elif status == 0:
                _MSD_STREAK.pop(key, None)
            return status == 0
        kmin, scale = assumed
        ok = status == 0 and (lo > hi or (lo >= kmin and _msd_bucket(hi - kmin, scale) <= 254))
        if not ok:
            # equal buckets over a remembered range did not fit this view: callers that bring the view's key range run on
            # level tables built on the device from here on (auto mode; no back-off for them), callers that do not -- the
            # stage API -- leave the host plan alone for a while instead of paying a repair per call
            _HOST_PLAN_BACKOFF[key] = 64
            if _DEPTH_TABLE == "auto" and not (status & 4):
                while len(_TABLE_MODE) >= _CAPACITY_MAX_KEYS and key not in _TABLE_MODE:
                    _TABLE_MODE.pop(next(iter(_TABLE_MODE)))
                _TABLE_MODE[key] = _TABLE_VIEWS
        return ok
    return _key_bits_needed(lo, hi) <= assumed


def _capacity_for(spec: Speculation, key, n: int) -> Optional[int]:
    ent = spec.capacity.get(key)
    if ent is None:
        return None
    return max(4096, int(ent[0] * max(n, 1) * 1.25) + 1024)


def _note_count(spec: Speculation, key, n: int, m: int) -> None:
    _CAPACITY = spec.capacity
    ratio = m / max(n, 1)
    ent = _CAPACITY.get(key)
    if ent is None:
        while len(_CAPACITY) >= _CAPACITY_MAX_KEYS:      # bounded: forget the oldest scene shape
            _CAPACITY.pop(next(iter(_CAPACITY)))
        _CAPACITY[key] = [ratio, 0]
    elif ratio > ent[0]:                                 # grow immediately (the head-room absorbs small growth)
        ent[0], ent[1] = ratio, 0
    elif ratio < 0.5 * ent[0]:
        ent[1] += 1
        if ent[1] >= _SHRINK_AFTER:
            ent[0], ent[1] = ratio, 0
    else:
        ent[0] = max(ratio, 0.98 * ent[0])               # decayed maximum
        ent[1] = 0


def _depth_plan(spec: "Speculation", key, n: int, cap, have_key_range: bool, exact: bool = False, depth_msd: Optional[bool] = None):
    """Which depth ordering a view takes, from what ``spec`` remembers of its scene shape: None -- LSD passes (first call of
    a shape, repair, back-off, populations outside the bucketed range) -- or the plan of the bucketed one, (key_min, scale)
    for host-planned equal buckets / (0, 0, "device") for the view's own depth table.  Counts down the shape's back-offs."""
    _KEY_RANGE, _MSD_BACKOFF, _TABLE_MODE, _HOST_PLAN_BACKOFF = spec.key_range, spec.msd_backoff, spec.table_mode, spec.host_plan_backoff
    rng = None if exact else _KEY_RANGE.get(key)
    backoff = _MSD_BACKOFF.get(key, 0)
    if backoff:
        _MSD_BACKOFF[key] = backoff - 1
    use_msd = depth_msd if depth_msd is not None else (_DEPTH_MSD_DEFAULT and _MSD_MIN_N <= n <= _MSD_MAX_N and backoff == 0)
    plan = None
    if cap is not None and use_msd:
        left = _TABLE_MODE.get(key, 0)
        want_table = _DEPTH_TABLE == "1" or (_DEPTH_TABLE == "auto" and left > 0)
        if want_table and have_key_range:
            if left > 0:
                _TABLE_MODE[key] = left - 1
            plan = (0, 0, "device")     # the view's own depth table, built on the device
        elif rng is not None:
            # (a caller without a key range -- the stage API, tile_lists()'s re-binning -- stays on the host's equal
            # buckets, with a back-off of their own after a miss, and does not use up the shape's views on the device plan)
            wait = _HOST_PLAN_BACKOFF.get(key, 0)
            if wait and depth_msd is None:
                _HOST_PLAN_BACKOFF[key] = wait - 1
            else:
                plan = _msd_plan(rng[0], rng[1])
    return plan


def bin_tiles(splat, radii, tiles_touched, depth_keys, width: int, height: int, cull_mode: int = 0,
              exact: bool = False, fused: Optional[bool] = None, tile_rects=None, depth_msd: Optional[bool] = None,
              bin_tile: int = TILE_SIZE, key_range: Optional[torch.Tensor] = None,
              qbox: Optional[torch.Tensor] = None, speculation: Optional["Speculation"] = None) -> TileBins:
    """Depth-order the gaussians, emit their tile records in that order, stable-sort the records by tile.
    -> TileBins(sorted_tile, sorted_gid, tile_offsets [n_tiles+1] i32).  ``sorted_tile`` holds 16-bit tile ids (an
    int16 tensor of uint16 bit patterns) on the fused path, int32 otherwise: read it through ``tile_ids``.
    ``key_range``: the buffer ``project_fwd(key_range=...)`` filled for THIS view: the bucketed depth ordering then plans
    its buckets from the view's own key range on the device (no remembered range, no repair when the views differ).
    ``qbox``: the quadrant bounding boxes ``project_fwd(qbox=...)`` left for THIS view: where the records are placed in one
    pass (and N <= 2^24) the placement packs every record's box into the top byte of its id (``TileBins.boxed``).
    ``bin_tile`` = 32 (fused path only, ``tile_rects`` from ``project_fwd(bin_tile=32)``): the records are binned into
    32 x 32 bins instead -- tile ids, ``tile_offsets`` and the record count are then those of the bin grid, and the
    compositor stages must be told (their ``bin_tile``).
    ``speculation``: whose memory of this scene shape (record capacity, depth-key range, bucket plan) sizes the launch and
    learns from the view (``Speculation``; default: the process-wide ``DEFAULT_SPECULATION``).

    The record count M is only known on the device.  Steady state (a capacity is remembered per scene shape): all
    kernels are queued for that capacity and read the live count from device memory, the host never waits; the
    caller runs ``finalize()`` afterwards and repeats with ``exact=True`` if the capacity was too small.
    First call / ``exact``: one host read-back of M sizes the buffers exactly."""
    dev, N = splat.device, splat.shape[0]
    tile_w, tile_h = tile_grid(width, height, bin_tile)
    n_tiles = tile_w * tile_h
    st = _stream()
    spec = speculation if speculation is not None else DEFAULT_SPECULATION
    _KEY_RANGE, _MSD_BACKOFF, _TABLE_MODE, _HOST_PLAN_BACKOFF = spec.key_range, spec.msd_backoff, spec.table_mode, spec.host_plan_backoff
    key = (dev.type, dev.index, width, height, cull_mode, bin_tile)
    offsets = torch.empty(N, dtype=torch.int32, device=dev)
    ctl = torch.empty(4, dtype=torch.int32, device=dev)   # record count M, min / max visible depth key, spare
    total = ctl[0:1]
    ws_bytes = lib.gsa_sort_workspace_bytes(N)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    # fused path (every tile of the rectangle is a record, 16-bit tile ids): the records are generated inside the
    # first pass of the tile sort from the depth-ordered rectangles, never written in emission order
    can_fuse = cull_mode == 0 and bool(lib.gsa_tile_bins_supported(width, height, bin_tile))
    if fused and not can_fuse:
        raise ValueError("bin_tiles(fused=True): needs cull_mode 0 and a tile grid gsa_tile_bin16_supported accepts")
    if fused is None:   # (_BIN_FUSED_DEFAULT = False: the emit + 32-bit sort path, kept for A/B timing)
        fused = can_fuse and (_BIN_FUSED_DEFAULT or bin_tile != TILE_SIZE)
    if bin_tile != TILE_SIZE and not (fused and tile_rects is not None):
        raise ValueError("bin_tiles(bin_tile=32): needs the fused path and the tile_rects of project_fwd(bin_tile=32)")
    cap = None if exact else _capacity_for(spec, key, N)
    key_bits, keys_ok, slot = 32, True, None
    if fused:
        if tile_rects is None:   # callers that did not keep project_fwd's dense copy: word 11 of the records
            tile_rects = torch.stack([torch.where(tiles_touched > 0, splat[:, 11].contiguous().view(torch.int32), 0),
                                      tiles_touched], dim=1).contiguous()
        # depth order.  The range of the visible depth keys is remembered per scene shape (and checked after the fact,
        # like the record capacity).  Steady state: ONE global pass buckets the keys by depth range and ONE launch sorts
        # every bucket in LDS, gathers rectangles and counts, scans them and writes offsets + block owners
        # (gsa_depth_order_msd).  Otherwise (first call of a shape, repair, range too wide for 16-bit local sorts): LSD
        # passes over the key bits the visible gaussians differ in, then the scan.
        order = torch.empty(N, dtype=torch.int32, device=dev)
        rects = torch.empty(N, dtype=torch.int32, device=dev)
        rng = None if exact else _KEY_RANGE.get(key)
        plan = _depth_plan(spec, key, N, cap, key_range is not None, exact, depth_msd)
        PATH_STATS["lsd" if plan is None else ("bucketed_device_table" if len(plan) == 3 else "bucketed_host_plan")] += 1
        if plan is not None:
            key_bits = plan
            n_blocks = (cap + BIN_BLOCK - 1) // BIN_BLOCK
            block_owner = torch.empty(n_blocks + 1, dtype=torch.int32, device=dev)
            dws_bytes = lib.gsa_depth_order_msd_workspace_bytes(N)
            dws = torch.empty(dws_bytes, dtype=torch.uint8, device=dev)
            # the kernel leaves the control words in a pinned slot itself (tagged with a token the host polls for)
            m_host, m_event, pool = _pinned_scalar(dev)
            token = _next_token()
            m_host[3] = 0
            slot = (m_host, m_event, token, pool)
            _call("gsa_depth_order", lib.gsa_depth_order_msd, N, _ptr(depth_keys), _ptr(tile_rects), plan[0], plan[1],
                  _ptr(key_range) if len(plan) == 3 else None,
                  _ptr(order), _ptr(rects), _ptr(offsets), _ptr(block_owner), n_blocks, _ptr(ctl),
                  ctypes.c_void_p(m_host.data_ptr()), token, _ptr(dws), dws_bytes, st)
        else:
            if rng is not None:
                key_bits = _key_bits_needed(rng[0], rng[1])
            dws_bytes = lib.gsa_depth_order_workspace_bytes(N)
            dws = torch.empty(dws_bytes, dtype=torch.uint8, device=dev)
            _call("gsa_depth_order", lib.gsa_depth_order, N, _ptr(depth_keys), _ptr(tile_rects), key_bits, _ptr(order),
                  _ptr(rects), _ptr(offsets), _ptr(ctl[1:3]), _ptr(dws), dws_bytes, st)

            def scan(capacity, report=None):   # the scan also notes the owner of every 4096-slot record block (+ the last owner)
                n_blocks = 0 if capacity is None else (capacity + BIN_BLOCK - 1) // BIN_BLOCK
                owner = torch.empty(n_blocks + 1, dtype=torch.int32, device=dev)
                if report is not None:   # (steady state: the scan's middle launch leaves the control words in the pinned slot itself)
                    _call("gsa_scan_tiles", lib.gsa_scan_tiles_sorted_report, N, _ptr(offsets), _ptr(owner), n_blocks, _ptr(ctl),
                          ctypes.c_void_p(report[0].data_ptr()), report[2], _ptr(ws), ws_bytes, st)
                else:
                    _call("gsa_scan_tiles", lib.gsa_scan_tiles_sorted, N, _ptr(offsets), _ptr(owner), n_blocks, _ptr(total),
                          _ptr(ws), ws_bytes, st)
                return owner

            counts_copy = offsets.clone() if cap is None else None   # the scan is in place; a second scan needs the counts
            if cap is not None:
                m_host, m_event, pool = _pinned_scalar(dev)
                m_host[3] = 0
                slot = (m_host, m_event, _next_token(), pool)
            block_owner = scan(cap, slot)
    else:
        _, order = sort_pairs(N, depth_keys, None, 0, 32)  # ties keep index order; culled gaussians sort last
        _call("gsa_scan_tiles", lib.gsa_scan_tiles, N, _ptr(tiles_touched), _ptr(order), _ptr(offsets), _ptr(total),
              _ptr(ws), ws_bytes, st)
    if slot is None:   # (every path but the bucketed ordering: a 16-byte copy behind the scan, and an event)
        m_host, m_event, pool = _pinned_scalar(dev)
        m_host.copy_(ctl, non_blocking=True)
        m_event.record()
        slot = (m_host, m_event, None, pool)
    m_known = None
    if cap is None:
        _wait_slot(slot)
        m_known = int(m_host[0]) & 0xFFFFFFFF
        _note_count(spec, key, N, m_known)
        if fused:
            keys_ok = _note_keys(spec, key, int(m_host[1]) & 0xFFFFFFFF, int(m_host[2]) & 0xFFFFFFFF, key_bits,
                                 int(m_host[3]) if isinstance(key_bits, tuple) else 0)
        cap = m_known
        _release_slot(slot)   # (read: this call's TileBins knows its count)
        slot = None
        if fused:   # first call of a scene shape / repair: the block table needs the count, scan once more
            offsets.copy_(counts_copy)
            block_owner = scan(cap)
    tile_offsets = torch.empty(n_tiles + 1, dtype=torch.int32, device=dev)
    if fused:
        direct = _BIN_DIRECT and bool(lib.gsa_tile_bin16_direct(width, height, bin_tile))
        # quadrant boxes in the ids' top byte: where the placement is one pass and the caller brought the gaussians' boxes
        boxed = direct and qbox is not None and bool(lib.gsa_tile_bin16_boxes(width, height, bin_tile, N))
        sorted_tile = None if direct else torch.empty(cap, dtype=torch.int16, device=dev)   # uint16 tile ids (see tile_ids)
        sorted_gid = torch.empty(cap, dtype=torch.int32, device=dev)
        two = n_tiles > 256 and not direct   # (n_tiles: of the bin grid)
        k_tmp = torch.empty(cap, dtype=torch.int16, device=dev) if two else None
        v_tmp = torch.empty(cap, dtype=torch.int32, device=dev) if two else None
        ws2_bytes = lib.gsa_tile_bin16_workspace_bytes(width, height, bin_tile, cap) if direct else lib.gsa_sort_workspace_bytes(cap)
        ws2 = torch.empty(ws2_bytes, dtype=torch.uint8, device=dev)
        tile_order = None
        if direct and _SCHEDULE_IN_PLACEMENT and N > 0 and cap > 0:
            # the compositor's launch order out of the placement launch itself (eight extra workgroups): no tile_schedule launch
            tw16, th16 = tile_grid(width, height)
            tile_order = torch.empty(8 * ((tw16 * th16 + 7) // 8), dtype=torch.int32, device=dev)
            _call("gsa_tile_bin16", lib.gsa_tile_bin16_scheduled, N, _ptr(order), _ptr(offsets), _ptr(rects), _ptr(block_owner),
                  width, height, bin_tile, cap, _ptr(total), _ptr(sorted_tile), _ptr(sorted_gid), _ptr(tile_offsets),
                  _ptr(qbox) if boxed else None, _ptr(tile_order), _ptr(ws2), ws2_bytes, st)
        else:
            _call("gsa_tile_bin16", lib.gsa_tile_bin16, N, _ptr(order), _ptr(offsets), _ptr(rects), _ptr(block_owner), width,
                  height, bin_tile, cap,
                  _ptr(total), _ptr(k_tmp), _ptr(v_tmp), _ptr(sorted_tile), _ptr(sorted_gid), _ptr(tile_offsets),
                  _ptr(qbox) if boxed else None, _ptr(ws2), ws2_bytes, st)
        bins = TileBins(sorted_tile, sorted_gid, tile_offsets, cap, m_known, key, slot, N, key_bits, keys_ok, boxed, spec)
        bins.tile_order = tile_order
        if _DEBUG_BINS:   # (diagnostics of an implausible read-back: what went in and what the device holds)
            bins._debug = (ctl, tile_rects, depth_keys, offsets)
        return bins
    isect_tile = torch.empty(cap, dtype=torch.int32, device=dev)
    isect_gid = torch.empty(cap, dtype=torch.int32, device=dev)
    _call("gsa_tile_emit", lib.gsa_tile_emit_dev, N, _ptr(order), _ptr(offsets), _ptr(splat), _ptr(radii), width,
          height, cull_mode, cap, _ptr(total), _ptr(isect_tile), _ptr(isect_gid), st)
    tile_bits = max(1, math.ceil(math.log2(n_tiles + 1)))
    if cap > 0:
        sorted_tile = torch.empty(cap, dtype=torch.int32, device=dev)
        sorted_gid = torch.empty(cap, dtype=torch.int32, device=dev)
        k_tmp, v_tmp = torch.empty_like(sorted_tile), torch.empty_like(sorted_gid)
        ws2_bytes = lib.gsa_sort_workspace_bytes(cap)
        ws2 = torch.empty(ws2_bytes, dtype=torch.uint8, device=dev)
        _call("gsa_sort_pairs_u32", lib.gsa_sort_pairs_u32_dev, cap, _ptr(total), _ptr(isect_tile), _ptr(isect_gid),
              _ptr(sorted_tile), _ptr(sorted_gid), _ptr(k_tmp), _ptr(v_tmp), 0, tile_bits, _ptr(ws2), ws2_bytes, st)
    else:
        sorted_tile, sorted_gid = isect_tile, isect_gid
    _call("gsa_tile_offsets", lib.gsa_tile_offsets_dev, cap, _ptr(total), _ptr(sorted_tile), n_tiles,
          _ptr(tile_offsets), st)
    return TileBins(sorted_tile, sorted_gid, tile_offsets, cap, m_known, key, slot, N, None, True, False, spec)


# the one-pass placement writes the compositor's launch order in eight extra workgroups of its last launch (False: a
# tile_schedule launch per view; A/B from tests / tools)
_SCHEDULE_IN_PLACEMENT = True


def tile_schedule(tile_offsets, width: int, height: int, bin_tile: int = TILE_SIZE) -> torch.Tensor:
    """-> tile_order [8 * ceil(n_tiles / 8)] i32: per tile band, the tiles with the longest record lists first (launch
    order of the compositor kernels; speed only).  ``bin_tile``: what ``tile_offsets`` indexes (see ``bin_tiles``).
    Given the ``TileBins`` themselves it returns the order their placement launch wrote on the side, if it did."""
    if isinstance(tile_offsets, TileBins):
        if tile_offsets.tile_order is not None:
            return tile_offsets.tile_order
        tile_offsets = tile_offsets.tile_offsets
    tile_w, tile_h = tile_grid(width, height)
    n_tiles = tile_w * tile_h
    order = torch.empty(8 * ((n_tiles + 7) // 8), dtype=torch.int32, device=tile_offsets.device)
    _call("gsa_tile_schedule", lib.gsa_tile_schedule, n_tiles, _ptr(tile_offsets), _ptr(order), tile_w, bin_tile, _stream())
    return order


# False: the backward sweep keeps the forward's launch order (by list length) instead of ordering the tiles once more by
# the records their quadrants blended -- one launch and one kernel boundary less per view, a less even sweep (0.197 ->
# 0.203 ms): a single view 0.5 % slower, the 4-view step 0.7 % faster (tools/bench_noreorder.py A/B).  Kept on: the sweep is
# the path's dominant kernel and a view alone is what rounds are compared on.
_BWD_REORDER = True


# Replicas for the gradient records of gaussians that cover many tiles (cameras inside the scene: thousands of same-line
# atomics per splat; gsa_composite_bwd_lists_hot): the gradient-record buffer gets gsa_hot_rows() rows behind the N records.
# GSPLAT_AMD_HOT_SPLATS=0: the plain sweep (same sums up to the order of the additions).
_HOT_SPLATS = os.environ.get("GSPLAT_AMD_HOT_SPLATS", "1") != "0"
HOT_ROWS = int(lib.gsa_hot_rows())


class QuadLists(tuple):
    """(quad_list, quad_count) of ``composite_fwd(want_lists=True)``, plus ``zeroed``: the [N,16] gradient-record buffer
    the forward kernel cleared on the side -- ``composite_bwd`` accumulates into it ONCE (a second backward pass over
    the same lists clears a buffer of its own)."""
    zeroed = None


def composite_fwd(splat, sorted_gid, tile_offsets, background, width: int, height: int, channels: int,
                  tile_order=None, want_lists: bool = False, bin_tile: int = TILE_SIZE, gid_boxes: bool = False):
    """``sorted_gid``: ``TileBins.sorted_gid_buffer``; ``gid_boxes``: ``TileBins.boxed`` of the same bins (the entries' top
    byte is a quadrant box; speed only).
    -> render [H,W,channels], alpha [H,W,1], final_T [H,W], last_id [H,W] i32
    [, QuadLists(quad_list [4 * len(sorted_gid)] i32, quad_count [n_tiles,4] i32) with ``want_lists``: per 8x8 quadrant
    the records that were blended there -- what ``composite_bwd(..., quad_lists=...)`` sweeps]."""
    dev = splat.device
    render = torch.empty(height, width, channels, dtype=torch.float32, device=dev)
    alpha = torch.empty(height, width, 1, dtype=torch.float32, device=dev)
    final_T = torch.empty(height, width, dtype=torch.float32, device=dev)
    last_id = torch.empty(height, width, dtype=torch.int32, device=dev)
    quad_list = quad_count = zeroed = None
    if want_lists:
        tile_w, tile_h = tile_grid(width, height)
        slices = 4 * (bin_tile // TILE_SIZE) ** 2   # quadrant slices per record list (a 32 x 32 bin: four tiles' worth)
        quad_list = torch.empty(slices * max(int(sorted_gid.numel()), 1), dtype=torch.int32, device=dev)
        quad_count = torch.empty(tile_w * tile_h, 4, dtype=torch.int32, device=dev)
        # the backward sweep's gradient records (+ the hot region behind them): the forward kernel clears them
        zeroed = torch.empty(splat.shape[0] + (HOT_ROWS if _HOT_SPLATS else 0), SPLAT_STRIDE, dtype=torch.float32, device=dev)
    _call("gsa_composite_fwd", lib.gsa_composite_fwd, width, height, channels, _ptr(splat), _ptr(sorted_gid),
          1 if gid_boxes else 0, _ptr(tile_offsets), _ptr(background), _ptr(tile_order), _ptr(render), _ptr(alpha), _ptr(final_T),
          _ptr(last_id), _ptr(quad_list), _ptr(quad_count), _ptr(zeroed), 0 if zeroed is None else zeroed.shape[0],
          bin_tile, _stream())
    if want_lists:
        lists = QuadLists((quad_list, quad_count))
        lists.zeroed = zeroed
        return render, alpha, final_T, last_id, lists
    return render, alpha, final_T, last_id


def moments_to_record_grads(splat: torch.Tensor, v_splat: torch.Tensor) -> torch.Tensor:
    """Diagnostic / test helper: the sweep's raw moments (``composite_bwd``'s output, see composite.hip) ->
    d loss / d (mean2d.xy, conic abc, opacity, rgb, depth) in the splat-record layout.  ``project_bwd`` does this
    conversion itself on the device; this is the same linear map in torch."""
    a, b, c, o = splat[:, 2], splat[:, 3], splat[:, 4], splat[:, 5]
    m = v_splat
    g = v_splat.clone()
    g[:, 0] = a * m[:, 0] + b * m[:, 1]
    g[:, 1] = b * m[:, 0] + c * m[:, 1]
    g[:, 2] = 0.5 * m[:, 2]
    g[:, 4] = 0.5 * m[:, 4]
    g[:, 5] = torch.where(o > 0, -m[:, 5] / o.clamp_min(1e-30), torch.zeros_like(o))  # culled: o = 0
    return g


def record_grads_to_moments(splat: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """Inverse of ``moments_to_record_grads`` (the conic is positive definite): lets a test drive ``project_bwd``
    with a chosen record gradient."""
    a, b, c, o = splat[:, 2], splat[:, 3], splat[:, 4], splat[:, 5]
    det = a * c - b * b
    m = g.clone()
    m[:, 0] = (c * g[:, 0] - b * g[:, 1]) / det
    m[:, 1] = (a * g[:, 1] - b * g[:, 0]) / det
    m[:, 2] = 2.0 * g[:, 2]
    m[:, 4] = 2.0 * g[:, 4]
    m[:, 5] = -g[:, 5] * o
    return m


def fold_hot_replicas(v_splat: torch.Tensor, n: int) -> torch.Tensor:
    """What the projection backward does when it reads a marked record, for callers that look at the records themselves:
    adds every claimed entry's replica rows (the hot region behind row ``n`` of ``v_splat``, ``gsa_composite_bwd_lists_hot``)
    to its owner's record and clears the marks.  Plain torch ops, no host wait.  -> the [n,16] records (a view)."""
    entries = int(lib.gsa_hot_rows()) * SPLAT_STRIDE // (SPLAT_STRIDE * _HOT_REPLICAS + 1)
    records, rep = v_splat[:n], v_splat[n:n + entries * _HOT_REPLICAS].view(_HOT_REPLICAS, entries, SPLAT_STRIDE)
    tags = v_splat[n + entries * _HOT_REPLICAS:].contiguous().view(torch.int32).view(-1)[:entries]
    claimed = tags > 0
    owner = torch.where(claimed, tags.long() - 1, torch.zeros_like(tags, dtype=torch.long))
    records.index_add_(0, owner, rep.sum(0) * claimed[:, None])
    records[:, 11] = 0.0
    return records


_HOT_REPLICAS = 16   # (GSA_HOT_REPLICAS of include/gsplat_amd.h: the layout of the hot region)


def composite_bwd(splat, sorted_gid, tile_offsets, background, final_T, last_id, v_render, v_alpha, width: int,
                  height: int, channels: int, tile_order=None, quad_lists=None, bin_tile: int = TILE_SIZE,
                  gid_boxes: bool = False, v_rgb_out: Optional[torch.Tensor] = None, fold: bool = True):
    """-> v_splat [N,16]: per gaussian the raw moments of u = d loss / d sigma over d = mean2d - pixel
    (sum u dx, u dy, u dx^2, u dx dy, u dy^2, u) and sum fac * v_out per channel -- the gradient record
    ``project_bwd`` consumes (``moments_to_record_grads`` gives the mean / conic / opacity gradients).
    ``v_rgb_out`` (a contiguous fp32 [N,3] tensor; sweeps over the forward's lists only): receives the colour moments --
    fields 6..8 -- already masked by the forward's colour clamp, i.e. d loss / d rgb per gaussian in the layout the
    view-parallel exchange all-gathers (it is zeroed here; those fields of ``v_splat`` stay zero).
    ``fold`` (sweeps over the forward's lists with the hot region on): True -- the records are complete on return
    (``fold_hot_replicas``); False -- what the sweep left: gaussians that cover many tiles have their sums in the replica rows
    behind the returned view and a mark in word 11, which ``project_bwd`` / ``project_views_bwd`` resolve themselves (the
    product's callers: no extra pass)."""
    if v_rgb_out is not None:
        if quad_lists is None or v_rgb_out.shape != (splat.shape[0], 3) or v_rgb_out.dtype != torch.float32 or not v_rgb_out.is_contiguous():
            raise ValueError("composite_bwd(v_rgb_out=...): needs the forward's lists and a contiguous fp32 [N,3] tensor")
        v_rgb_out.zero_()
    if quad_lists is not None:   # sweep the forward pass's lists of blended records (same result, fewer visits)
        quad_list, quad_count = quad_lists
        # one launch: order the tiles by the work the lists hold (exact, unlike the list lengths `tile_order` was built
        # from before the forward pass) -- and clear the gradient records, unless the forward kernel did (QuadLists)
        n = splat.shape[0]
        hot = _HOT_SPLATS and v_rgb_out is None
        rows = n + (HOT_ROWS if hot else 0)   # (the hot region lies behind the N records, cleared with them)
        v_splat = getattr(quad_lists, "zeroed", None)
        fill = v_splat is None or v_splat.shape[0] < rows
        if fill:
            v_splat = torch.empty(rows, SPLAT_STRIDE, dtype=torch.float32, device=splat.device)
        else:
            quad_lists.zeroed = None   # (accumulated into once)
        tile_w, tile_h = tile_grid(width, height)
        if fill or tile_order is None or _BWD_REORDER:
            tile_order = torch.empty(8 * ((tile_w * tile_h + 7) // 8), dtype=torch.int32, device=splat.device)
            _call("gsa_composite_bwd_prepare", lib.gsa_composite_bwd_prepare, rows if fill else 0, _ptr(v_splat),
                  tile_w * tile_h, _ptr(quad_count), _ptr(tile_order), _stream())
        if hot:
            _call("gsa_composite_bwd", lib.gsa_composite_bwd_lists_hot, width, height, channels, _ptr(splat), _ptr(sorted_gid),
                  1 if gid_boxes else 0, _ptr(tile_offsets), _ptr(background), _ptr(tile_order), _ptr(final_T), _ptr(last_id),
                  _ptr(v_render), _ptr(v_alpha), _ptr(quad_list), _ptr(quad_count), n, _ptr(v_splat), bin_tile, _stream())
            # (a view: the replicas stay behind it, where the projection backward looks for them)
            return fold_hot_replicas(v_splat, n) if fold else v_splat[:n]
        v_splat = v_splat[:n]
        if v_rgb_out is not None:
            _call("gsa_composite_bwd", lib.gsa_composite_bwd_lists_rgb, width, height, channels, _ptr(splat), _ptr(sorted_gid),
                  1 if gid_boxes else 0, _ptr(tile_offsets), _ptr(background), _ptr(tile_order), _ptr(final_T), _ptr(last_id),
                  _ptr(v_render), _ptr(v_alpha), _ptr(quad_list), _ptr(quad_count), _ptr(v_splat), _ptr(v_rgb_out), bin_tile, _stream())
            return v_splat
        _call("gsa_composite_bwd", lib.gsa_composite_bwd_lists, width, height, channels, _ptr(splat), _ptr(sorted_gid),
              1 if gid_boxes else 0, _ptr(tile_offsets), _ptr(background), _ptr(tile_order), _ptr(final_T), _ptr(last_id), _ptr(v_render),
              _ptr(v_alpha), _ptr(quad_list), _ptr(quad_count), _ptr(v_splat), bin_tile, _stream())
        return v_splat
    v_splat = torch.zeros_like(splat)
    _call("gsa_composite_bwd", lib.gsa_composite_bwd, width, height, channels, _ptr(splat), _ptr(sorted_gid),
          1 if gid_boxes else 0, _ptr(tile_offsets), _ptr(background), _ptr(tile_order), _ptr(final_T), _ptr(last_id), _ptr(v_render),
          _ptr(v_alpha), _ptr(v_splat), bin_tile, _stream())
    return v_splat


def project_bwd_outputs(means, quats, scales, opacities, colors, out, accumulate: bool, v_rgb_out, sh_aux):
    """The five gradient tensors of a projection backward (``out`` checked, or fresh ones) -- what ``project_bwd`` validates,
    for the fused per-view call (``ViewChain.backward``)."""
    N = means.shape[0]
    if accumulate and out is None:
        raise ValueError("project_bwd(accumulate=True) needs the `out` buffers it adds to")
    if out is None:
        out = (torch.empty_like(means), torch.empty_like(quats), torch.empty_like(scales), torch.empty_like(opacities),
               None if v_rgb_out is not None else torch.empty_like(colors))
    else:
        for got, like in zip(out, (means, quats, scales, opacities, colors)):
            if got.numel() != like.numel() or got.dtype != torch.float32 or not got.is_contiguous() or got.device != like.device:
                raise ValueError("grad_out tensors must be contiguous fp32, sized like the inputs, on their device")
        out = tuple(out)
    if sh_aux is not None and (sh_aux.shape != (N, 12) or sh_aux.dtype != torch.float32 or not sh_aux.is_contiguous()
                               or sh_aux.device != means.device):
        raise ValueError("sh_aux must be the contiguous fp32 [N,12] tensor project_fwd returned")
    if v_rgb_out is not None and (v_rgb_out.shape != (N, 3) or v_rgb_out.dtype != torch.float32 or not v_rgb_out.is_contiguous()):
        raise ValueError("v_rgb_out must be a contiguous fp32 [N,3] tensor")
    return out


def project_bwd(means, quats, scales, opacities, colors, sh_degree: int, viewmat, Kmat, width: int, height: int,
                splat, radii, v_splat, out=None, accumulate: bool = False, v_rgb_out=None, sh_aux=None):
    """-> v_means, v_quats, v_scales, v_opacities, v_colors (every element written).  ``out``: five preallocated
    contiguous fp32 tensors to write into instead (e.g. views of one flat all-reduce bucket).  ``sh_aux``: the digest
    ``project_fwd(return_aux=True)`` left for this view (the coefficient rows are then not read)."""
    N = means.shape[0]
    sh_coeffs = colors.shape[1] if sh_degree >= 0 else 0
    if out is None:
        v_means, v_quats, v_scales = torch.empty_like(means), torch.empty_like(quats), torch.empty_like(scales)
        v_opac, v_colors = torch.empty_like(opacities), torch.empty_like(colors)
    else:
        v_means, v_quats, v_scales, v_opac, v_colors = out
        for got, like in zip(out, (means, quats, scales, opacities, colors)):
            if got.numel() != like.numel() or got.dtype != torch.float32 or not got.is_contiguous() \
                    or got.device != like.device:
                raise ValueError("grad_out tensors must be contiguous fp32, sized like the inputs, on their device")
    if accumulate and out is None:
        raise ValueError("project_bwd(accumulate=True) needs the `out` buffers it adds to")
    if sh_aux is not None and (sh_aux.shape != (N, 12) or sh_aux.dtype != torch.float32 or not sh_aux.is_contiguous()
                               or sh_aux.device != means.device):
        raise ValueError("sh_aux must be the contiguous fp32 [N,12] tensor project_fwd returned")
    if v_rgb_out is not None:
        # factored form: d loss / d rgb [N,3] instead of the [N,K,3] coefficient gradient (v_colors is not written)
        if v_rgb_out.shape != (N, 3) or v_rgb_out.dtype != torch.float32 or not v_rgb_out.is_contiguous():
            raise ValueError("v_rgb_out must be a contiguous fp32 [N,3] tensor")
        _call("gsa_project_bwd", lib.gsa_project_bwd_factored, N, _ptr(means), _ptr(quats), _ptr(scales), _ptr(opacities),
              _ptr(colors), sh_degree, sh_coeffs, _ptr(viewmat), _ptr(Kmat), width, height, _ptr(splat), _ptr(radii),
              _ptr(v_splat), _ptr(v_means), _ptr(v_quats), _ptr(v_scales), _ptr(v_opac), _ptr(v_rgb_out),
              1 if accumulate else 0, _ptr(sh_aux), _stream())
        return v_means, v_quats, v_scales, v_opac, None
    _call("gsa_project_bwd", lib.gsa_project_bwd_acc, N, _ptr(means), _ptr(quats), _ptr(scales), _ptr(opacities),
          _ptr(colors), sh_degree, sh_coeffs, _ptr(viewmat), _ptr(Kmat), width, height, _ptr(splat), _ptr(radii),
          _ptr(v_splat), _ptr(v_means), _ptr(v_quats), _ptr(v_scales), _ptr(v_opac), _ptr(v_colors),
          1 if accumulate else 0, _ptr(sh_aux), _stream())
    return v_means, v_quats, v_scales, v_opac, v_colors


# ---- one C-ABI call per view forward, one per view backward (gsa_view_fwd / gsa_view_bwd, ABI 13) ----------------------
# GSPLAT_AMD_VIEW_CALLS=0: the stage-by-stage path everywhere (A/B; tests set the module attribute).
_VIEW_CALLS = os.environ.get("GSPLAT_AMD_VIEW_CALLS", "1") != "0"
_VIEW_LAYOUTS = {}
_VIEW_WS_CAP = {}     # frame shape -> the largest record capacity a workspace of this shape was laid out for
_ALIGN = 256


def _view_layout(n: int, cap: int, width: int, height: int, channels: int, bin_tile: int, lists: bool):
    """Byte offsets of everything a view's chain needs besides its projection and the two tensors the caller sees (render,
    alpha) inside ONE workspace allocation -> ({name: offset}, total bytes, n_blocks, depth-ordering bytes, placement bytes).
    Cached per shape: a view costs one ``torch.empty`` instead of fifteen."""
    key = (n, cap, width, height, channels, bin_tile, lists, _HOT_SPLATS)
    lay = _VIEW_LAYOUTS.get(key)
    if lay is None:
        t16 = tile_grid(width, height)
        n16 = t16[0] * t16[1]
        tb = tile_grid(width, height, bin_tile)
        n_blocks = (cap + BIN_BLOCK - 1) // BIN_BLOCK
        dws = int(lib.gsa_depth_order_msd_workspace_bytes(n))
        ws2 = int(lib.gsa_tile_bin16_workspace_bytes(width, height, bin_tile, cap))
        order_words = 8 * ((n16 + 7) // 8)
        p = width * height
        parts = [("order", 4 * n), ("rects", 4 * n), ("offsets", 4 * n), ("block_owner", 4 * (n_blocks + 1)), ("ctl", 16),
                 ("depth_ws", dws), ("sorted_gid", 4 * cap), ("tile_offsets", 4 * (tb[0] * tb[1] + 1)), ("tile_order", 4 * order_words),
                 ("bin_ws", ws2), ("final_T", 4 * p), ("last_id", 4 * p)]
        if lists:
            slices = 4 * (bin_tile // TILE_SIZE) ** 2
            parts += [("quad_list", 4 * slices * max(cap, 1)), ("quad_count", 16 * n16), ("v_splat", 4 * SPLAT_STRIDE * (n + (HOT_ROWS if _HOT_SPLATS else 0))),
                      ("bwd_tile_order", 4 * order_words)]
        off, offsets, sizes = 0, {}, {}
        for name, nbytes in parts:
            offsets[name], sizes[name] = off, nbytes
            off = (off + nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
        while len(_VIEW_LAYOUTS) >= 64:
            _VIEW_LAYOUTS.pop(next(iter(_VIEW_LAYOUTS)))
        lay = _VIEW_LAYOUTS[key] = (offsets, sizes, max(off, _ALIGN), n_blocks, dws, ws2)
    return lay


class ViewChain:
    """One view's chain behind its projection as TWO C-ABI calls: ``forward`` (bucketed depth ordering, one-pass placement with
    the compositor's launch order, forward sweep -- ``gsa_view_fwd``) and ``backward`` (tile re-ordering, reverse sweep and,
    unless the step finishes all views at once, the projection backward -- ``gsa_view_bwd``).  Everything the chain needs
    besides the projection's tensors and the two the caller sees lives in ONE workspace tensor; views of it are built only
    when somebody asks (``view``).  Same kernels, same arguments, same bits as ``bin_tiles`` + ``tile_schedule`` +
    ``composite_fwd`` / ``composite_bwd`` + ``project_bwd``; what it does not cover (first call of a frame size, repairs,
    populations or grids outside the bucketed ordering / the one-pass placement, ``cull_mode`` 1) returns None from
    ``forward`` and the caller takes the stages."""

    __slots__ = ("args", "ws", "lay", "render", "alpha", "bins", "dims", "keep", "has_lists", "_views", "__weakref__")

    @staticmethod
    def forward(splat, depth_keys, tile_rects, key_range, qbox, width: int, height: int, channels: int, cull_mode: int,
                bin_tile: int, background, want_lists: bool, speculation=None, records_ready=None) -> Optional["ViewChain"]:
        from ._lib import ViewArgs
        if not _VIEW_CALLS or cull_mode != 0 or TIMER.enabled or not _BIN_FUSED_DEFAULT or not _BIN_DIRECT or not _SCHEDULE_IN_PLACEMENT:
            return None   # (the stage timer wants one event pair per stage)
        dev, n = splat.device, int(splat.shape[0])
        if n == 0 or tile_rects is None or not lib.gsa_tile_bin16_direct(width, height, bin_tile):
            return None
        spec = speculation if speculation is not None else DEFAULT_SPECULATION
        key = (dev.type, dev.index, width, height, cull_mode, bin_tile)
        cap = _capacity_for(spec, key, n)
        if cap is None:
            return None
        plan = _depth_plan(spec, key, n, cap, key_range is not None)
        if plan is None:
            return None
        PATH_STATS["bucketed_device_table" if len(plan) == 3 else "bucketed_host_plan"] += 1
        if torch.cuda.current_device() != dev.index:
            raise RuntimeError(f"gsplat_amd: a view on cuda:{dev.index} while the current device is cuda:{torch.cuda.current_device()}")
        boxed = qbox is not None and bool(lib.gsa_tile_bin16_boxes(width, height, bin_tile, n))
        # The workspace is laid out for the LARGEST capacity this frame shape has asked for (in steps of 256 K records), the
        # kernels run with the view's own: the remembered capacity follows the cameras (a decaying maximum of records per
        # gaussian), and a workspace whose size changed from view to view made the caching allocator split and re-malloc its
        # 0.4-GB blocks -- 2 ms of host time per view where the whole view takes 0.6 on the device.
        shape = (n, width, height, channels, bin_tile, bool(want_lists))
        ws_cap = max(_VIEW_WS_CAP.get(shape, 0), (cap + (1 << 18) - 1) >> 18 << 18)
        if _VIEW_WS_CAP.get(shape) != ws_cap:
            while len(_VIEW_WS_CAP) >= 64:
                _VIEW_WS_CAP.pop(next(iter(_VIEW_WS_CAP)))
            _VIEW_WS_CAP[shape] = ws_cap
        offsets, _, total, _, dws, ws2 = lay = _view_layout(n, ws_cap, width, height, channels, bin_tile, bool(want_lists))
        n_blocks = (cap + BIN_BLOCK - 1) // BIN_BLOCK
        ws = torch.empty(total, dtype=torch.uint8, device=dev)
        render = torch.empty(height, width, channels, dtype=torch.float32, device=dev)
        alpha = torch.empty(height, width, 1, dtype=torch.float32, device=dev)
        m_host, m_event, pool = _pinned_scalar(dev)
        token = _next_token()
        m_host[3] = 0
        base = ws.data_ptr()
        a = ViewArgs()
        a.struct_bytes = ctypes.sizeof(ViewArgs)
        a.n, a.width, a.height, a.channels, a.bin_tile, a.gid_boxes, a.n_blocks = n, width, height, channels, bin_tile, int(boxed), n_blocks
        a.m_capacity, a.key_min, a.scale, a.token = cap, plan[0], plan[1], token
        a.splat, a.depth_keys, a.tile_rects = splat.data_ptr(), depth_keys.data_ptr(), tile_rects.data_ptr()
        a.key_range = key_range.data_ptr() if len(plan) == 3 else None
        a.qbox = qbox.data_ptr() if boxed else None
        a.background = None if background is None else background.data_ptr()
        a.order, a.rects, a.offsets = base + offsets["order"], base + offsets["rects"], base + offsets["offsets"]
        a.block_owner, a.ctl, a.ctl_host = base + offsets["block_owner"], base + offsets["ctl"], m_host.data_ptr()
        a.depth_ws, a.depth_ws_bytes = base + offsets["depth_ws"], dws
        a.sorted_gid, a.tile_offsets, a.tile_order = base + offsets["sorted_gid"], base + offsets["tile_offsets"], base + offsets["tile_order"]
        a.bin_ws, a.bin_ws_bytes = base + offsets["bin_ws"], ws2
        a.wait_event = None if records_ready is None else records_ready.cuda_event
        a.render, a.alpha = render.data_ptr(), alpha.data_ptr()
        a.final_T, a.last_id = base + offsets["final_T"], base + offsets["last_id"]
        if want_lists:
            a.quad_list, a.quad_count, a.v_splat = base + offsets["quad_list"], base + offsets["quad_count"], base + offsets["v_splat"]
            a.bwd_tile_order = base + offsets["bwd_tile_order"]
            a.hot = 1 if _HOT_SPLATS else 0   # (with sweep_rgb the backward clears the flag: that sweep has no hot form)
        check(lib.gsa_view_fwd(ctypes.byref(a), _stream()), "gsa_view_fwd")
        ch = ViewChain()
        ch.args, ch.ws, ch.lay, ch.render, ch.alpha, ch._views = a, ws, lay, render, alpha, {}
        ch.dims, ch.has_lists = (n, cap, width, height, channels, bin_tile), bool(want_lists)
        # (what the struct points at must outlive the launches: the projection's tensors, the background, the records event)
        ch.keep = (splat, depth_keys, tile_rects, key_range, qbox, background, records_ready)
        bins = TileBins(None, None, None, cap, None, key, (m_host, m_event, token, pool), n, plan, True, boxed, spec)
        bins._chain = weakref.ref(ch)
        ch.bins = bins
        return ch

    _DTYPES = {"order": torch.int32, "rects": torch.int32, "offsets": torch.int32, "sorted_gid": torch.int32,
               "tile_offsets": torch.int32, "tile_order": torch.int32, "bwd_tile_order": torch.int32, "final_T": torch.float32,
               "last_id": torch.int32, "quad_list": torch.int32, "quad_count": torch.int32, "v_splat": torch.float32,
               "ctl": torch.int32}

    def view(self, name: str, dtype=None) -> torch.Tensor:
        """A tensor view of one of the workspace's buffers (flat, or shaped where the stage API shapes it)."""
        v = self._views.get(name)
        if v is None:
            offsets, sizes = self.lay[0], self.lay[1]
            n, cap, width, height, channels, bt = self.dims
            nbytes = 4 * SPLAT_STRIDE * n if name == "v_splat" else sizes[name]   # (without the hot region behind the records)
            flat = self.ws[offsets[name]:offsets[name] + nbytes].view(dtype or self._DTYPES[name])
            shape = {"final_T": (height, width), "last_id": (height, width), "v_splat": (n, SPLAT_STRIDE),
                     "quad_count": (-1, 4)}.get(name)
            v = self._views[name] = flat.view(shape) if shape else flat
        return v

    def backward(self, v_render, v_alpha=None, project=None, wait_bucket=None, sweep_rgb=None):
        """Queues the view's backward on the current stream.  ``project``: None -- stop behind the sweep (the gradient record
        is ``view('v_splat')``; ``project_views_bwd`` finishes the step's views) -- or a dict {means, quats, scales, opacities,
        colors, sh_degree, viewmat, Kmat, radii, sh_aux, out: (v_means .. v_colors), accumulate, v_rgb_out}: the projection
        backward behind the sweep, as ``project_bwd``.  ``wait_bucket``: an event the stream waits for in front of it.
        ``sweep_rgb``: as ``composite_bwd(v_rgb_out=...)`` (zeroed here)."""
        if not self.has_lists:
            raise RuntimeError("ViewChain.backward: the forward was run without lists (no backward was announced)")
        a = self.args
        a.v_render = v_render.data_ptr()
        a.v_alpha = None if v_alpha is None else v_alpha.data_ptr()
        held = [v_render, v_alpha, sweep_rgb]
        if sweep_rgb is not None:
            if sweep_rgb.shape != (self.dims[0], 3) or sweep_rgb.dtype != torch.float32 or not sweep_rgb.is_contiguous():
                raise ValueError("ViewChain.backward(sweep_rgb=...): a contiguous fp32 [N,3] tensor")
            sweep_rgb.zero_()
        a.sweep_rgb = None if sweep_rgb is None else sweep_rgb.data_ptr()
        if sweep_rgb is not None:
            a.hot = 0   # (the hot region was cleared by the forward and simply stays unused)
        if project is None:
            a.project_mode = 0
        else:
            out, v_rgb = project["out"], project.get("v_rgb_out")
            colors = project["colors"]
            a.project_mode = 2 if v_rgb is not None else 1
            a.accumulate = 1 if project.get("accumulate") else 0
            a.sh_degree = project["sh_degree"]
            a.sh_coeffs = colors.shape[1] if project["sh_degree"] >= 0 else 0
            a.wait_bucket = None if wait_bucket is None else wait_bucket.cuda_event
            a.means, a.quats, a.scales = project["means"].data_ptr(), project["quats"].data_ptr(), project["scales"].data_ptr()
            a.opacities, a.colors = project["opacities"].data_ptr(), colors.data_ptr()
            a.viewmat, a.intrin, a.radii = project["viewmat"].data_ptr(), project["Kmat"].data_ptr(), project["radii"].data_ptr()
            aux = project.get("sh_aux") if project["sh_degree"] >= 1 else None
            a.sh_aux = None if aux is None else aux.data_ptr()
            a.v_means, a.v_quats, a.v_scales, a.v_opacities = (t.data_ptr() for t in out[:4])
            a.v_colors = None if (v_rgb is not None or out[4] is None) else out[4].data_ptr()
            a.v_rgb = None if v_rgb is None else v_rgb.data_ptr()
            held += [project, wait_bucket]
        check(lib.gsa_view_bwd(ctypes.byref(a), _stream()), "gsa_view_bwd")
        del held


# ---- training-step stages around the rasterizer ------------------------------------
def view_loss(render: torch.Tensor, target: torch.Tensor, lambda_dssim: float, grad_scale: float = 1.0,
              want_grad: bool = True, sums_out: Optional[torch.Tensor] = None):
    """The reference's per-view loss (trainer.py:485-492) and its gradient in one pass over render and target.
    render / target: [H,W,3] fp32 on the device.  -> (sums [3] float64 on the device: sum |d|, sum SSIM map,
    sum d^2 over the 3HW elements; v_render [H,W,3] = grad_scale * d loss / d render, or None).
    ``sums_out``: a ZEROED contiguous float64 [3] device tensor to accumulate into (a row of the step's table)."""
    if render.shape != target.shape or render.dim() != 3 or render.shape[2] != 3:
        raise ValueError("render and target must both be [H,W,3]")
    h, w = int(render.shape[0]), int(render.shape[1])
    r, t = f32c(render, "render"), f32c(target, "target")
    ws_bytes = lib.gsa_view_loss_workspace_bytes(w, h)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=r.device)
    if sums_out is not None:
        if sums_out.shape != (3,) or sums_out.dtype != torch.float64 or not sums_out.is_contiguous() or sums_out.device != r.device:
            raise ValueError("view_loss: `sums_out` must be a contiguous float64 [3] tensor on the render's device")
        sums = sums_out
    else:
        sums = torch.zeros(3, dtype=torch.float64, device=r.device)
    v = torch.empty_like(r) if want_grad else None
    _call("gsa_view_loss", lib.gsa_view_loss, w, h, _ptr(r), _ptr(t), float(lambda_dssim), float(grad_scale), _ptr(ws),
          ws_bytes, _ptr(v), _ptr(sums), _stream())
    return sums, v


def grad_unscale(grad: torch.Tensor, inv_scale: float, found_inf: torch.Tensor) -> None:
    """grad *= inv_scale in place; ``found_inf`` (int32 [1] on the device, accumulated) becomes 1 if an element is not
    finite -- GradScaler.unscale_ for the flat gradient bucket."""
    if not (grad.is_cuda and grad.dtype == torch.float32 and grad.is_contiguous()):
        raise ValueError("grad_unscale needs a contiguous fp32 device tensor")
    _call("gsa_grad_unscale", lib.gsa_grad_unscale, grad.numel(), _ptr(grad), float(inv_scale), _ptr(found_inf), _stream())


def adam_step(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
              segment_end, segment_lr, step: int, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-15,
              skip_if_nonzero: Optional[torch.Tensor] = None, stripes=None) -> None:
    """One fused Adam update of a flat fp32 bucket in place (torch.optim.Adam arithmetic, per-segment lr).
    ``skip_if_nonzero``: int32 [1] on the device; a non-zero value skips the whole update (GradScaler.step).
    ``stripes``: per segment ``None`` or ``(period, head, head_lr)`` -- the first ``head`` elements of every ``period``
    take ``head_lr`` (the SH table as one [N,16,3] segment with the reference's two learning rates)."""
    import ctypes
    n = param.numel()
    for t in (param, grad, exp_avg, exp_avg_sq):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise ValueError("adam_step needs four contiguous fp32 device tensors of equal size")
    k = len(segment_end)
    ends = (ctypes.c_longlong * k)(*[int(e) for e in segment_end])
    lrs = (ctypes.c_float * k)(*[float(x) for x in segment_lr])
    if stripes is not None and any(st is not None for st in stripes):
        if len(stripes) != k:
            raise ValueError("adam_step: one stripe entry per segment")
        st = [x if x is not None else (0, 0, 0.0) for x in stripes]
        period = (ctypes.c_int * k)(*[int(x[0]) for x in st])
        head = (ctypes.c_int * k)(*[int(x[1]) for x in st])
        head_lr = (ctypes.c_float * k)(*[float(x[2]) for x in st])
        _call("gsa_adam_step_striped", lib.gsa_adam_step_striped, n, _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq),
              k, ctypes.cast(ends, ctypes.c_void_p), ctypes.cast(lrs, ctypes.c_void_p), ctypes.cast(period, ctypes.c_void_p),
              ctypes.cast(head, ctypes.c_void_p), ctypes.cast(head_lr, ctypes.c_void_p), beta1, beta2, eps, int(step),
              _ptr(skip_if_nonzero), _stream())
        return
    _call("gsa_adam_step", lib.gsa_adam_step, n, _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), k,
          ctypes.cast(ends, ctypes.c_void_p), ctypes.cast(lrs, ctypes.c_void_p), beta1, beta2, eps, int(step),
          _ptr(skip_if_nonzero), _stream())


def densify_gather(n_old: int, field_width, xyz_field: int, scaling_field: int, log_split: float, src_row: torch.Tensor,
                   kind: torch.Tensor, noise: torch.Tensor, old_param: torch.Tensor, old_m=None, old_v=None):
    """One gather pass over a field-major bucket (and optionally its Adam moments): see ``gsa_densify_gather``.
    -> (new_param, new_m, new_v) flat fp32 tensors for n_new = len(src_row) rows."""
    import ctypes
    n_new = int(src_row.numel())
    total = int(sum(field_width))
    dev = old_param.device
    if old_param.numel() != total * n_old or (old_m is None) != (old_v is None):
        raise ValueError("densify_gather: bucket size / moment arguments inconsistent")
    src = src_row.to(torch.int32).contiguous()
    kd = kind.to(torch.uint8).contiguous()
    nz = f32c(noise, "noise")
    if nz.shape != (n_new, 3):
        raise ValueError("noise must be [n_new, 3]")
    new_p = torch.empty(total * n_new, dtype=torch.float32, device=dev)
    new_m = torch.empty_like(new_p) if old_m is not None else None
    new_v = torch.empty_like(new_p) if old_v is not None else None
    widths = (ctypes.c_int * len(field_width))(*[int(w) for w in field_width])
    _call("gsa_densify_gather", lib.gsa_densify_gather, n_old, n_new, len(field_width), ctypes.cast(widths, ctypes.c_void_p),
          xyz_field, scaling_field, float(log_split), _ptr(src), _ptr(kd), _ptr(nz), _ptr(old_param), _ptr(old_m),
          _ptr(old_v), _ptr(new_p), _ptr(new_m), _ptr(new_v), _stream())
    return new_p, new_m, new_v


def sh_grad_expand(means, sh_degree: int, viewmats, Kmat, v_rgb_all, v_colors, accumulate: bool = False,
                   v_rest=None) -> None:
    """v_colors [N,K,3] (+)= sum over views of SH-basis(view) (x) v_rgb_all[view]  (v_rgb_all: [V,N,3]; viewmats [V,4,4]).
    Rebuilds the SH-coefficient gradient from the factored form ``project_bwd(v_rgb_out=...)`` leaves.
    With ``v_rest`` [N,K-1,3] the output is split the way the reference model stores the coefficients: ``v_colors``
    [N,1,3] receives the DC part (both overwritten)."""
    N = means.shape[0]
    V = viewmats.shape[0]
    if v_rgb_all.shape != (V, N, 3) or not v_rgb_all.is_contiguous() or not v_colors.is_contiguous():
        raise ValueError("sh_grad_expand: v_rgb_all must be contiguous [V,N,3], v_colors contiguous [N,K,3]")
    if v_rest is not None:
        if accumulate or not v_rest.is_contiguous() or v_colors.shape[1] != 1:
            raise ValueError("sh_grad_expand(v_rest=...): overwrite only; v_colors [N,1,3], v_rest [N,K-1,3] contiguous")
        _call("gsa_sh_grad_expand", lib.gsa_sh_grad_expand_split, N, _ptr(f32c(means, "means")), sh_degree,
              v_rest.shape[1] + 1, V, _ptr(f32c(viewmats, "viewmats")), _ptr(f32c(Kmat, "Ks")), _ptr(v_rgb_all),
              _ptr(v_colors), _ptr(v_rest), _stream())
        return
    _call("gsa_sh_grad_expand", lib.gsa_sh_grad_expand, N, _ptr(f32c(means, "means")), sh_degree, v_colors.shape[1], V,
          _ptr(f32c(viewmats, "viewmats")), _ptr(f32c(Kmat, "Ks")), _ptr(v_rgb_all), _ptr(v_colors),
          1 if accumulate else 0, _stream())


def activate_fwd(scaling, rotation, opacity, features_dc=None, features_rest=None):
    """Raw parameters of the reference model -> (scales [N,3], quats [N,4], opacities [N], shs [N,16,3]) in one pass
    (gaussian_model.py:86-108).  Without the two feature tensors only the geometry is activated and ``shs`` is None
    (a model that keeps its coefficients as [N,16,3] rows hands those to the rasterizer as they are)."""
    n, dev = scaling.shape[0], scaling.device
    with_sh = features_dc is not None
    if with_sh and (features_rest is None or features_dc.shape[1:] != (1, 3) or features_rest.shape[1:] != (15, 3)):
        raise ValueError("activate_fwd: features_dc [N,1,3] and features_rest [N,15,3] expected")
    scales = torch.empty(n, 3, device=dev)
    quats = torch.empty(n, 4, device=dev)
    opac = torch.empty(n, device=dev)
    shs = torch.empty(n, 16, 3, device=dev) if with_sh else None
    _call("gsa_activate_fwd", lib.gsa_activate_fwd, n, _ptr(f32c(scaling, "scaling")), _ptr(f32c(rotation, "rotation")),
          _ptr(f32c(opacity, "opacity")), _ptr(f32c(features_dc, "features_dc")) if with_sh else None,
          _ptr(f32c(features_rest, "features_rest")) if with_sh else None,
          _ptr(scales), _ptr(quats), _ptr(opac), _ptr(shs), _stream())
    return scales, quats, opac, shs


def step_report(view_sums: Optional[torch.Tensor], view_counts, lambda_dssim: float, n_views_total: int,
                reg_partials: Optional[torch.Tensor], reg_factor: float, add_reg: bool, device) -> torch.Tensor:
    """The step's metrics in one launch (``gsa_step_report``): -> float64 [5] on the device = sum l1, sum ssim, sum psnr
    over the LOCAL views, their loss share (+ the regulariser with ``add_reg``), the regulariser.  ``view_sums``: the
    [B_local,3] table the views' ``view_loss(sums_out=row)`` calls filled; ``view_counts``: 3 H W of every local view."""
    import ctypes
    b = len(view_counts)
    if b > 16:
        # (the kernel takes 16 views a launch; every entry of the report is a sum over the views, so chunks add up -- the
        # regulariser is counted with the first chunk.  A trainer calls this BEHIND its optimizer step: it must not raise.)
        if view_sums is None or view_sums.shape != (b, 3):
            raise ValueError("step_report: `view_sums` must be a contiguous float64 [B_local,3] device tensor")
        report = None
        for lo in range(0, b, 16):
            part = step_report(view_sums[lo:lo + 16], view_counts[lo:lo + 16], lambda_dssim, n_views_total,
                               reg_partials if lo == 0 else None, reg_factor, add_reg and lo == 0, device)
            report = part if report is None else report + part
        return report
    if b and (view_sums is None or view_sums.shape != (b, 3) or view_sums.dtype != torch.float64 or not view_sums.is_contiguous()):
        raise ValueError("step_report: `view_sums` must be a contiguous float64 [B_local,3] device tensor")
    report = torch.empty(5, dtype=torch.float64, device=device)
    inv = (ctypes.c_double * max(b, 1))(*[1.0 / float(c) for c in view_counts])
    _call("gsa_step_report", lib.gsa_step_report, b, _ptr(view_sums) if b else None, ctypes.cast(inv, ctypes.c_void_p),
          float(lambda_dssim), int(n_views_total), _ptr(reg_partials), 0 if reg_partials is None else reg_partials.numel(),
          float(reg_factor), 1 if add_reg else 0, _ptr(report), _stream())
    return report


def activate_bwd(scaling, rotation, opacity, v_scales, v_quats, v_opac, out_scaling, out_rotation, out_opacity,
                 reg_max_ratio: float, reg_weight: float, add_reg_grad: bool, return_partials: bool = False):
    """Chain rule of ``activate_fwd`` into the raw-parameter gradient buffers (overwritten) + the scale-ratio
    regulariser (trainer.py:434-445): its gradient is added when ``add_reg_grad``; returns its value (0-d, fp64) -- or,
    with ``return_partials``, (partial sums on the device, the factor their sum is multiplied by) for ``step_report``."""
    n = scaling.shape[0]
    for t, name in ((out_scaling, "out_scaling"), (out_rotation, "out_rotation"), (out_opacity, "out_opacity")):
        if not (t.is_contiguous() and t.dtype == torch.float32):
            raise ValueError(f"activate_bwd: `{name}` must be a contiguous fp32 buffer")
    partials = torch.empty(lib.gsa_activate_partials(n), dtype=torch.float64, device=scaling.device)
    _call("gsa_activate_bwd", lib.gsa_activate_bwd, n, _ptr(f32c(scaling, "scaling")), _ptr(f32c(rotation, "rotation")),
          _ptr(f32c(opacity, "opacity")), _ptr(f32c(v_scales, "v_scales")), _ptr(f32c(v_quats, "v_quats")),
          _ptr(f32c(v_opac, "v_opacities")), float(reg_max_ratio), float(reg_weight), 1 if add_reg_grad else 0,
          _ptr(out_scaling), _ptr(out_rotation), _ptr(out_opacity), _ptr(partials), _stream())
    if return_partials:
        return partials, float(reg_weight) / max(n, 1)
    return partials.sum() * (float(reg_weight) / max(n, 1))
