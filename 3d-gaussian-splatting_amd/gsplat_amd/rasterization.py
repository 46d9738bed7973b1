"""Operator surface: ``rasterization(...)`` with the keyword surface and return contract of the call
the reference makes at /root/reference/src/backends/gsplat_backend.py:168-182
(``gsplat.rasterization``), executed by the HIP library through its C ABI.

One ``torch.autograd.Function`` per camera chains the C-ABI stages (``stages.py``); torch only owns memory
(caching allocator), the stream and autograd bookkeeping.
"""

from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

from . import stages as S
from .stages import TILE_SIZE

_RENDER_MODES = ("RGB", "D", "ED", "RGB+D", "RGB+ED")


class _RasterizeOneCamera(torch.autograd.Function):
    """Forward: project -> depth sort -> scan -> emit -> tile sort -> offsets -> composite.
    Backward: composite_bwd -> project_bwd.  Differentiable w.r.t. means, quats, scales, opacities, colors
    (/root/reference/src/training/trainer.py:889-898 reads those grads); none to camera or background."""

    @staticmethod
    def forward(ctx, means, quats, scales, opacities, colors, viewmat, Kmat, background,
                width: int, height: int, sh_degree: int, channels: int, cull_mode: int, grad_out, grad_accumulate,
                sh_grad_factored, deferred=None, projection=None, collector=None, speculation=None, grad_ready=None):
        means_c, quats_c, scales_c = S.f32c(means, "means"), S.f32c(quats, "quats"), S.f32c(scales, "scales")
        opac_c, colors_c = S.f32c(opacities, "opacities"), S.f32c(colors, "colors")
        view_c, K_c = S.f32c(viewmat, "viewmats"), S.f32c(Kmat, "Ks")
        bg_c = None if background is None else S.f32c(background, "backgrounds")

        # with a backward pass to come, the forward sweep also notes per 8x8 quadrant which records it blended (the
        # reverse sweep then visits exactly those) and the projection leaves its 48-B digest of every coefficient row
        want_lists = any(ctx.needs_input_grad[:5])
        # the records are binned into 32 x 32 BINS where the fused binning applies (S.bin_tile_for): the four 16 x 16
        # tiles of a bin walk the bin's list -- a compositor wave re-tests every record against its 8 x 8 quadrant anyway
        bt = S.bin_tile_for(width, height, cull_mode)
        if projection is not None:
            # this camera's slice of a multi-camera projection (project_cameras: one launch for the views of a step)
            if projection.dims != (width, height, cull_mode, sh_degree) or projection.bin_tile != bt or \
                    projection.splat.shape[0] != means_c.shape[0]:
                raise ValueError("rasterization(projection=...): made for another frame size, cull mode, SH degree or scene")
            # ... for THIS camera (the forward would blend the projection's splats while the backward differentiates the
            # argument's camera) and from THESE parameter values (same storage, not written since): host-side compares of
            # pointers and version counters, no device sync
            if projection.camera_stamp is not None and projection.camera_stamp != S.input_stamp(view_c, K_c):
                raise ValueError("rasterization(projection=...): viewmats / Ks are not the camera this projection was made for -- "
                                 "pass the tensors given to project_cameras (or projection.viewmat / projection.Kmat)")
            if projection.made_from is not None and projection.made_from != S.input_stamp(means, quats, scales, opacities, colors):
                raise ValueError("rasterization(projection=...): the gaussians are not the tensors this projection was made from, "
                                 "or were written since -- project the cameras again")
            kr, qb = projection.key_range, projection.qbox
            splat, radii, tiles_touched = projection.splat, projection.radii, projection.tiles_touched
            depth_keys, rects = projection.depth_keys, projection.tile_rects
            sh_aux = projection.sh_aux if want_lists else None
            records_ready = projection.records_ready
        else:
            kr = S.key_range_buffer(means_c.shape[0], means_c.device)   # this view's depth-key range, for the depth ordering's bucket plan
            qb = S.qbox_buffer(means_c.shape[0], means_c.device)        # the gaussians' quadrant boxes, for the forward sweep's candidate test
            # the projection in two launches: what the binning needs on this stream, the splat records (which stream the
            # 192-B coefficient rows) on a side stream beside the binning chain -- joined in front of the sweep
            forked = [] if S._SPLIT_PROJECTION else None
            proj = S.project_fwd(means_c, quats_c, scales_c, opac_c, colors_c, sh_degree, view_c, K_c, width, height,
                                 cull_mode, return_rects=True, return_aux=want_lists, bin_tile=bt, key_range=kr, qbox=qb,
                                 records_ready=forked)
            splat, radii, tiles_touched, depth_keys, rects = proj[:5]
            sh_aux = proj[5] if want_lists else None
            records_ready = forked[0] if forked else None
        # the host does not wait for the record count: bins are sized for a remembered capacity, the compositor is
        # queued, and only then is the count (long since copied back) checked -- a miss repeats with exact sizes.
        # Steady state: the whole chain behind the projection -- depth ordering, placement, launch order, sweep -- is ONE
        # C-ABI call on one workspace allocation (stages.ViewChain: gsa_view_fwd); what that does not cover (first call of a
        # frame size, repairs, cull_mode 1, grids without the one-pass placement) goes stage by stage below.
        chain = S.ViewChain.forward(splat, depth_keys, rects, kr, qb, width, height, channels, cull_mode, bt, bg_c, want_lists,
                                    speculation=speculation, records_ready=records_ready)
        repair = False
        if chain is not None:
            if deferred is not None:
                deferred.append(chain.bins)
            elif not chain.bins.finalize()[1]:
                chain, repair = None, True
        if chain is not None:
            if collector is not None:
                collector.append(chain.bins)
            ctx.chain = chain
            ctx.save_for_backward(means_c, quats_c, scales_c, opac_c, colors_c, view_c, K_c, bg_c, splat, radii, sh_aux)
            ctx.dims = (width, height, sh_degree, channels)
            ctx.grad_out = grad_out
            ctx.grad_accumulate = bool(grad_accumulate)
            ctx.sh_grad_factored = sh_grad_factored
            ctx.grad_ready = grad_ready
            ctx.in_shapes = (means.shape, quats.shape, scales.shape, opacities.shape, colors.shape)
            ctx.deferred_projection = projection if (projection is not None and projection.defer_backward) else None
            ctx.mark_non_differentiable(splat, radii, tiles_touched)
            ctx.set_materialize_grads(False)
            # (the record lists, offsets and last ids stay inside the chain's workspace: views are built when meta is asked)
            render, alpha = chain.render, chain.alpha
            # (the chain must not keep the two OUTPUT tensors: their grad_fn is this node, which holds the chain -- a reference
            # cycle that only the cyclic collector frees, 0.4 GB a view in a loop that has frozen it)
            chain.render = chain.alpha = None
            return render, alpha, splat, radii, tiles_touched, chain
        ctx.chain = None
        bins = S.bin_tiles(splat, radii, tiles_touched, depth_keys, width, height, cull_mode, tile_rects=rects, exact=repair,
                           bin_tile=bt, key_range=None if repair else kr, qbox=qb, speculation=speculation)
        tile_order = S.tile_schedule(bins, width, height, bt)   # longest record lists first (speed only)
        if records_ready is not None:   # (the records part of a split projection: everything below reads the splat records)
            torch.cuda.current_stream().wait_event(records_ready)
        out = S.composite_fwd(splat, bins.sorted_gid_buffer, bins.tile_offsets, bg_c, width, height, channels,
                              tile_order, want_lists=want_lists, bin_tile=bt, gid_boxes=bins.boxed)
        if deferred is not None:
            # the caller checks the speculation (record capacity, depth-key range) later, for several views at once
            # (meta['speculation_ok']): the host does not wait here for this view's control words and runs ahead
            deferred.append(bins)
        elif not bins.finalize()[1]:
            bins = S.bin_tiles(splat, radii, tiles_touched, depth_keys, width, height, cull_mode, exact=True,
                               tile_rects=rects, bin_tile=bt, qbox=qb, speculation=speculation)
            tile_order = S.tile_schedule(bins, width, height, bt)
            out = S.composite_fwd(splat, bins.sorted_gid_buffer, bins.tile_offsets, bg_c, width, height, channels,
                                  tile_order, want_lists=want_lists, bin_tile=bt, gid_boxes=bins.boxed)
        render, alpha, final_T, last_id = out[:4]
        quad_list, quad_count = out[4] if want_lists else (None, None)
        ctx.zeroed_records = out[4].zeroed if want_lists else None   # cleared by the forward kernel; used once
        gid_buffer = bins.sorted_gid_buffer        # what the backward sweep reads (ids, boxed or not: ctx.gid_boxes)
        # (capacity-sized and with the boxes in the ids' top byte: rasterization() cuts them to the record count and masks
        # them when somebody looks at meta's lists -- nothing in a training loop does; cutting and masking here cost every
        # checked call an elementwise kernel over the records, 6 us of a 1 M-gaussian view)
        sorted_tile = bins._tile if bins._tile is not None else torch.empty(0, dtype=torch.int16, device=splat.device)
        sorted_gid, tile_offsets = gid_buffer, bins.tile_offsets
        if collector is not None:
            collector.append(bins)

        ctx.save_for_backward(means_c, quats_c, scales_c, opac_c, colors_c, view_c, K_c, bg_c, splat, radii,
                              gid_buffer, tile_offsets, final_T, last_id, tile_order, quad_list, quad_count, sh_aux)
        ctx.dims = (width, height, sh_degree, channels)
        ctx.bin_tile = bt
        ctx.gid_boxes = bins.boxed
        ctx.grad_out = grad_out
        ctx.grad_accumulate = bool(grad_accumulate)
        ctx.sh_grad_factored = sh_grad_factored
        ctx.grad_ready = grad_ready
        ctx.in_shapes = (means.shape, quats.shape, scales.shape, opacities.shape, colors.shape)
        # (project_cameras(defer_backward=True): this camera's backward pass stops behind the sweep and leaves its gradient
        # record with the projection; project_cameras_backward finishes all cameras of the step in one launch)
        ctx.deferred_projection = projection if (projection is not None and projection.defer_backward) else None
        extras = (splat, radii, tiles_touched, sorted_tile, sorted_gid, tile_offsets, last_id)
        ctx.mark_non_differentiable(*extras)
        ctx.set_materialize_grads(False)  # no zero tensors for the seven outputs nobody differentiates
        return (render, alpha) + extras

    @staticmethod
    def backward(ctx, v_render, v_alpha, *unused):
        if ctx.chain is not None:
            return _RasterizeOneCamera._backward_chain(ctx, v_render, v_alpha)
        (means_c, quats_c, scales_c, opac_c, colors_c, view_c, K_c, bg_c, splat, radii, sorted_gid, tile_offsets,
         final_T, last_id, tile_order, quad_list, quad_count, sh_aux) = ctx.saved_tensors
        width, height, sh_degree, channels = ctx.dims
        if v_render is None:
            v_render = torch.zeros(height, width, channels, dtype=torch.float32, device=means_c.device)
        v_render = v_render.to(torch.float32).contiguous()
        v_alpha_c = None if v_alpha is None else v_alpha.to(torch.float32).contiguous()
        lists = None
        if quad_list is not None:
            lists = S.QuadLists((quad_list, quad_count))
            lists.zeroed, ctx.zeroed_records = ctx.zeroed_records, None
        rgb_here = ctx.sh_grad_factored if (ctx.deferred_projection is not None and lists is not None) else None
        v_splat = S.composite_bwd(splat, sorted_gid, tile_offsets, bg_c, final_T, last_id, v_render, v_alpha_c,
                                  width, height, channels, tile_order,
                                  quad_lists=lists, bin_tile=ctx.bin_tile, gid_boxes=ctx.gid_boxes, v_rgb_out=rgb_here,
                                  fold=False)   # (the projection backward resolves the hot gaussians' replicas)
        if ctx.deferred_projection is not None:
            ctx.deferred_projection.v_splat, ctx.deferred_projection.v_rgb = v_splat, rgb_here
            return (None,) * 21
        ready = ctx.grad_ready() if ctx.grad_ready is not None else None
        if ready is not None:   # (the sweep above touched nothing of the caller's: only the sums below wait for their turn)
            torch.cuda.current_stream().wait_event(ready)
        grads = S.project_bwd(means_c, quats_c, scales_c, opac_c, colors_c, sh_degree, view_c, K_c, width, height,
                              splat, radii, v_splat, out=ctx.grad_out, accumulate=ctx.grad_accumulate,
                              v_rgb_out=ctx.sh_grad_factored, sh_aux=sh_aux)
        if ctx.grad_accumulate:   # the sums live in the caller's buffers; autograd gets nothing to accumulate
            return (None,) * 21
        need = ctx.needs_input_grad
        grads = tuple(g.view(shp) if (nd and g is not None) else None for g, shp, nd in zip(grads, ctx.in_shapes, need[:5]))
        return grads + (None,) * 16


def _backward_chain(ctx, v_render, v_alpha):
    """The backward pass of a view whose forward ran as one fused call: reverse sweep (+ projection backward) = gsa_view_bwd."""
    means_c, quats_c, scales_c, opac_c, colors_c, view_c, K_c, bg_c, splat, radii, sh_aux = ctx.saved_tensors
    width, height, sh_degree, channels = ctx.dims
    chain = ctx.chain
    if v_render is None:
        v_render = torch.zeros(height, width, channels, dtype=torch.float32, device=means_c.device)
    v_render = v_render.to(torch.float32).contiguous()
    v_alpha_c = None if v_alpha is None else v_alpha.to(torch.float32).contiguous()
    if ctx.deferred_projection is not None:
        # (with sh_grad_factored: the sweep itself leaves the masked colour gradient in the exchange layout, and
        # project_cameras_backward finishes the cameras' GEOMETRY in one launch -- the step of a view-parallel rank)
        chain.backward(v_render, v_alpha_c, sweep_rgb=ctx.sh_grad_factored)
        ctx.deferred_projection.v_splat, ctx.deferred_projection.v_rgb = chain.view("v_splat"), ctx.sh_grad_factored
        return (None,) * 21
    out = S.project_bwd_outputs(means_c, quats_c, scales_c, opac_c, colors_c, ctx.grad_out, ctx.grad_accumulate,
                                ctx.sh_grad_factored, sh_aux)
    chain.backward(v_render, v_alpha_c, project={"means": means_c, "quats": quats_c, "scales": scales_c, "opacities": opac_c,
                                                 "colors": colors_c, "sh_degree": sh_degree, "viewmat": view_c, "Kmat": K_c,
                                                 "radii": radii, "sh_aux": sh_aux, "out": out, "accumulate": ctx.grad_accumulate,
                                                 "v_rgb_out": ctx.sh_grad_factored},
                   wait_bucket=ctx.grad_ready() if ctx.grad_ready is not None else None)
    if ctx.grad_accumulate:   # the sums live in the caller's buffers; autograd gets nothing to accumulate
        return (None,) * 21
    grads = out[:4] + (None if ctx.sh_grad_factored is not None else out[4],)
    need = ctx.needs_input_grad
    grads = tuple(g.view(shp) if (nd and g is not None) else None for g, shp, nd in zip(grads, ctx.in_shapes, need[:5]))
    return grads + (None,) * 16


_RasterizeOneCamera._backward_chain = staticmethod(_backward_chain)


def project_cameras(means, quats, scales, opacities, colors, viewmats, Ks, width: int, height: int,
                    sh_degree: Optional[int] = None, cull_mode: Optional[int] = None, for_backward: Optional[bool] = None,
                    defer_backward: bool = False):
    """The projection stage of ``rasterization`` for the C cameras of ``viewmats`` [C,4,4] in ONE launch (same argument
    meaning as there; ``Ks`` [C,3,3] or [3,3]) -> a list of C per-camera projections for ``rasterization(projection=...)``.
    What it is for: the B views of an optimisation step are rendered from the same parameters
    (/root/reference/src/training/trainer.py:458-506), so their projections can share the read of every gaussian's
    geometry and SH coefficient row -- 236 of the 376 bytes a single-camera projection moves per gaussian.
    ``for_backward`` (default: any input requires grad): also leave the 48-B digest of each coefficient row the backward
    projection reads instead of the row.
    ``defer_backward``: the backward pass of a ``rasterization(projection=...)`` call then stops behind its sweep (its
    ``grad_out`` / ``sh_grad_factored`` arguments are not used, the inputs get no ``.grad``), and
    ``project_cameras_backward`` finishes ALL cameras in one launch: geometry read once, every gradient written once.
    Needs SH colours of degree >= 1 with [N,16,3] rows; otherwise the flag is dropped and every camera's backward is its own."""
    deg = -1 if sh_degree is None else int(sh_degree)
    cull = S.DEFAULT_CULL_MODE if cull_mode is None else int(cull_mode)
    if for_backward is None:
        for_backward = any(t.requires_grad for t in (means, quats, scales, opacities, colors))
    views = S.project_views(S.f32c(means, "means"), S.f32c(quats, "quats"), S.f32c(scales, "scales"),
                            S.f32c(opacities, "opacities"), S.f32c(colors, "colors"), deg, viewmats, Ks, int(width), int(height),
                            cull, return_aux=bool(for_backward))
    can_defer = bool(defer_backward) and deg >= 1 and colors.dim() == 3 and colors.shape[1] == 16 and len(views) <= 8 and \
        all(v.sh_aux is not None for v in views)
    stamp = S.input_stamp(means, quats, scales, opacities, colors)
    for v in views:
        v.defer_backward, v.v_splat, v.made_from = can_defer, None, stamp
    out = CameraProjections(views)
    out.deferred = can_defer   # (False although asked for: every camera's backward pass is its own -- see the docstring)
    return out


class CameraProjections(list):
    """What ``project_cameras`` returns: the per-camera projections, and ``deferred`` -- whether the cameras' backward passes
    stop behind their sweeps and wait for ``project_cameras_backward`` (``defer_backward`` asked for AND possible)."""
    deferred = False


def projection_buffers(projections):
    """The tensors behind the projections of ONE ``project_cameras`` call (the [C, ...] slabs the per-camera projections are
    slices of): what a caller that uses them on OTHER streams than the one they were made on hands the allocator
    (``Tensor.record_stream``) -- ``ViewPipeline.run(prepare=...)`` does it with what ``prepare`` returns."""
    p0 = projections[0]
    return [t._base if t._base is not None else t for t in
            (p0.splat, p0.radii, p0.tiles_touched, p0.depth_keys, p0.tile_rects, p0.sh_aux, p0.key_range, p0.qbox) if t is not None]


def project_cameras_backward(projections, means, quats, scales, opacities, colors, grad_out, accumulate: bool = False) -> None:
    """Finishes the backward passes of the cameras of ``project_cameras(defer_backward=True)`` in ONE launch, once every
    camera's ``rasterization(projection=...)`` output has been backpropagated (on the current stream, or on streams the
    current one has been made to wait for): the gradients of the five inputs, SUMMED over the cameras, are written into
    ``grad_out`` = (means, quats, scales, opacities [N], colors [N,16,3] shaped tensors; ``accumulate``: added to them).
    What it saves against one projection backward per camera: the geometry is read once instead of once per camera, no
    gradient is read back and added to, and the SH-coefficient gradient -- sum over the cameras of basis(direction) x
    d loss / d rgb -- is formed in registers instead of by an expansion kernel.
    Cameras that were backpropagated with ``sh_grad_factored`` (the sweeps left d loss / d rgb in the caller's [N,3] tensors):
    only the four GEOMETRY gradients are written; ``grad_out[4]`` is not touched (the caller rebuilds the coefficient gradient
    of all views -- its own and the other ranks' -- with ``stages.sh_grad_expand``)."""
    if not projections or not all(p.defer_backward for p in projections):
        raise ValueError("project_cameras_backward: these projections were not made with a deferred backward (project_cameras("
                         "defer_backward=True) drops the flag where the shared launch does not apply -- its result's `.deferred` "
                         "says so): every camera's own backward pass has written its gradients already")
    if any(p.v_splat is None for p in projections):
        raise RuntimeError("project_cameras_backward: a camera's output has not been backpropagated yet")
    width, height, _, deg = projections[0].dims
    with_rgb = [p.v_rgb is not None for p in projections]
    if any(with_rgb) and not all(with_rgb):
        raise RuntimeError("project_cameras_backward: some cameras were backpropagated with sh_grad_factored and some without")
    done = S.project_views_bwd(S.f32c(means, "means"), S.f32c(quats, "quats"), S.f32c(scales, "scales"),
                               S.f32c(opacities, "opacities"), S.f32c(colors, "colors"), deg, projections,
                               [p.v_splat for p in projections], width, height, grad_out, accumulate=accumulate,
                               v_rgbs=[p.v_rgb for p in projections] if all(with_rgb) else None)
    if not done:
        raise RuntimeError("project_cameras_backward: the batched kernel does not cover this configuration")
    for p in projections:
        p.v_splat = p.v_rgb = None


def rasterization(
    means: torch.Tensor,       # [N,3]
    quats: torch.Tensor,       # [N,4] (w first)
    scales: torch.Tensor,      # [N,3]
    opacities: torch.Tensor,   # [N]
    colors: torch.Tensor,      # [N,K,3] SH coefficients (sh_degree given) or [N,3] colours (sh_degree None)
    viewmats: torch.Tensor,    # [C,4,4] world->camera
    Ks: torch.Tensor,          # [C,3,3]
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = False,
    tile_size: int = TILE_SIZE,
    backgrounds: Optional[torch.Tensor] = None,
    render_mode: str = "RGB",
    cull_mode: Optional[int] = None,
    grad_out=None,
    grad_accumulate: bool = False,
    sh_grad_factored: Optional[torch.Tensor] = None,
    defer_check: bool = False,
    projection=None,
    speculation=None,
    grad_ready=None,
    **unsupported,
) -> Tuple[torch.Tensor, torch.Tensor, Dict]:
    """Returns ``(renders [C,H,W,D], alphas [C,H,W,1], meta)``; ``meta['means2d']`` [C,N,2] and
    ``meta['radii']`` [C,N,2] int32 are what the reference reads (gsplat_backend.py:204-208).

    ``cull_mode`` (extension, default ``stages.DEFAULT_CULL_MODE`` = 0): 0 = tile rectangle of mean2d +- radius (the published rule);
    1 = also skip tiles the alpha >= 1/255 ellipse cannot reach (same image and gradients, fewer records).
    ``grad_out`` (extension, single camera): five preallocated tensors (means, quats, scales, opacities, colors
    order) the backward pass writes its gradients into and hands to autograd -- e.g. views of one flat bucket
    that is then all-reduced in one call.  With ``grad_accumulate=True`` the backward pass ADDS into ``grad_out``
    (which the caller has initialised) and hands autograd nothing: several views sum into one bucket with no
    accumulation pass (``harness.ViewParallelTrainer(fused=True)``).
    ``sh_grad_factored`` (extension, single camera, SH colours): a [N,3] fp32 tensor that receives d loss / d rgb per
    gaussian instead of the [N,K,3] SH-coefficient gradient (``colors`` then gets no gradient; ``grad_out[4]`` is not
    written).  The coefficient gradient is the outer product of the view's SH basis with it
    (``stages.sh_grad_expand``): the form in which view-parallel ranks exchange it -- 3 floats instead of 48.
    With a ``projection`` of ``project_cameras(defer_backward=True)`` the backward SWEEP itself leaves it there (masked by the
    colour clamp), and ``project_cameras_backward`` then only sums the cameras' geometry gradients: a rank queues its
    all-gather for a view right behind the view's sweep.
    ``grad_ready`` (extension, with ``grad_out``): a callable asked DURING the backward pass for the event (or None) behind
    which ``grad_out`` may be written / added to -- the projection backward waits for it on its stream, the reverse sweep in
    front of it does not (views of a step that sum into one bucket in view order: their sweeps overlap, only the sums queue).
    ``defer_check`` (extension): the bins are sized for a remembered record capacity and the depth ordering assumes a
    remembered key range; by default the call waits (host side) for the view's control words once its sweep is queued and
    repeats the binning if either was too small, so what it returns is always right.  With ``defer_check=True`` it does
    not wait: the host queues view after view, far ahead of the device, and the CALLER asks ``meta['speculation_ok']``
    -- a bool, waiting for the control words only then -- before it trusts the view (image, gradients), and renders it
    again (without ``defer_check``) if it is False.  A training loop asks once per step, in front of the optimizer.
    ``projection`` (extension): what ``project_cameras`` returned for these cameras (one entry per camera, or the single
    entry itself for one camera): the projection stage is then not run again.  The views of an optimisation step share
    their gaussians, and ONE multi-camera launch reads a gaussian's geometry and coefficient row once for all of them; a
    call with C > 1 cameras does this by itself.
    ``speculation`` (extension): the ``stages.Speculation`` object whose memory of this frame size (record capacity, depth-key
    range, bucket plan) sizes the asynchronous binning and learns from the view -- a ``HipBackend`` / ``ViewPipeline`` passes
    its own, so that two consumers of equal frame size do not perturb each other; default: the process-wide one.
    """
    for k, v in unsupported.items():
        ok = (k in ("sparse_grad", "absgrad", "distributed") and not v) or \
             (k == "rasterize_mode" and v == "classic") or (k == "camera_model" and v == "pinhole")
        if not ok:
            raise NotImplementedError(f"gsplat_amd.rasterization: argument {k}={v!r} is outside the reference's call")
    if packed:
        raise NotImplementedError("gsplat_amd.rasterization: packed=True is outside the reference's call")
    if (near_plane, far_plane, radius_clip, eps2d, tile_size) != (0.01, 1e10, 0.0, 0.3, TILE_SIZE):
        raise NotImplementedError("gsplat_amd.rasterization: near/far/radius_clip/eps2d/tile_size are fixed at the "
                                  "defaults the reference uses (include/gsplat_amd_constants.h)")
    if render_mode not in _RENDER_MODES:
        raise ValueError(f"unknown render_mode {render_mode!r}")
    N = means.shape[0]
    C = viewmats.shape[0]
    if means.shape != (N, 3) or quats.shape != (N, 4) or scales.shape != (N, 3):
        raise ValueError("means/quats/scales must be [N,3]/[N,4]/[N,3]")
    if opacities.shape != (N,):
        raise ValueError(f"opacities must be [N], got {tuple(opacities.shape)}")
    if viewmats.shape != (C, 4, 4) or Ks.shape != (C, 3, 3):
        raise ValueError("viewmats/Ks must be [C,4,4]/[C,3,3]")
    if sh_degree is None:
        if colors.shape != (N, 3):
            raise ValueError("with sh_degree=None colors must be [N,3]")
        deg = -1
    else:
        deg = int(sh_degree)
        if not (0 <= deg <= 3):
            raise ValueError("sh_degree must be in 0..3")
        if colors.dim() != 3 or colors.shape[0] != N or colors.shape[2] != 3 or colors.shape[1] < (deg + 1) ** 2:
            raise ValueError("colors must be [N,K,3] with K >= (sh_degree+1)^2")
    if backgrounds is not None and backgrounds.shape != (C, 3):
        raise ValueError("backgrounds must be [C,3]")
    if grad_out is not None and C != 1:
        raise ValueError("grad_out needs a single camera (each camera's backward would overwrite it)")
    if grad_accumulate and grad_out is None:
        raise ValueError("grad_accumulate needs grad_out")
    deferred_cams = projection is not None and all(getattr(p_, "defer_backward", False) for p_ in
                                                   (projection if isinstance(projection, (list, tuple)) else [projection]))
    if sh_grad_factored is not None and ((grad_out is None and not deferred_cams) or sh_degree is None or C != 1):
        raise ValueError("sh_grad_factored needs SH colours (sh_degree given), a single camera, and grad_out or a projection "
                         "with a deferred backward")
    width, height = int(width), int(height)
    cull = S.DEFAULT_CULL_MODE if cull_mode is None else int(cull_mode)
    want_depth = render_mode != "RGB"
    channels = 4 if want_depth else 3

    renders, alphas, per_cam = [], [], []
    pending = [] if defer_check else None   # TileBins per camera whose control words nobody has waited for yet
    all_bins = []                           # TileBins of every camera (checked or not): the lists' record counts and id format
    if projection is not None and not isinstance(projection, (list, tuple)):
        projection = [projection]
    if projection is not None and len(projection) != C:
        raise ValueError("rasterization(projection=...): one entry per camera")
    own_projection = projection is None and C > 1
    if own_projection:   # several cameras: project for all of them in one launch
        projection = project_cameras(means, quats, scales, opacities, colors, viewmats, Ks, width, height, sh_degree=sh_degree,
                                     cull_mode=cull)
    for c in range(C):
        bg = None
        if backgrounds is not None:
            bg = backgrounds[c]
            if want_depth:
                bg = torch.cat([bg, bg.new_zeros(1)])
        vm_c, K_c = viewmats[c], Ks[c]
        if own_projection:   # (made above from these very arguments: its fp32 copies of the cameras are the cameras)
            vm_c, K_c = projection[c].viewmat, projection[c].Kmat
        out = _RasterizeOneCamera.apply(means, quats, scales, opacities, colors, vm_c, K_c, bg,
                                        width, height, deg, channels, cull, grad_out, grad_accumulate, sh_grad_factored,
                                        pending, None if projection is None else projection[c], all_bins, speculation,
                                        grad_ready)
        render, alpha = out[0], out[1]
        if render_mode in ("ED", "RGB+ED"):
            render = torch.cat([render[..., :3], render[..., 3:] / alpha.clamp(min=1e-10)], dim=-1)
        if render_mode in ("D", "ED"):
            render = render[..., 3:]
        renders.append(render)
        alphas.append(alpha)
        # splat, radii, tiles_touched, sorted_tile, sorted_gid, tile_offsets, last_id -- the last four as views of the fused
        # chain's workspace, built when somebody asks (``cam``)
        per_cam.append(out[2:] if len(out) == 9 else tuple(out[2:5]) + (out[5],))

    def cam(c):
        pc = per_cam[c]
        if len(pc) == 4:   # (fused chain: materialise the views once)
            ch = pc[3]
            pc = per_cam[c] = pc[:3] + (torch.empty(0, dtype=torch.int16, device=pc[0].device), ch.view("sorted_gid"),
                                        ch.view("tile_offsets"), ch.view("last_id"))
        return pc

    def settled():
        """per_cam with the record buffers cut to the record counts (deferred check: waits for the control words now)."""
        for c in range(len(per_cam)):
            cam(c)
        if all_bins:
            for c, b in enumerate(all_bins):
                m = b.finalize()[0]
                pc = list(per_cam[c])
                pc[3], pc[4] = (pc[3][:m] if pc[3].numel() else pc[3]), (pc[4][:m] & 0xFFFFFF if b.boxed else pc[4][:m])
                per_cam[c] = tuple(pc)
            all_bins.clear()
            if pending:
                pending.clear()
        return per_cam

    checked = list(pending) if pending else []

    def gather(i):  # [C, ...] without a copy for the single-camera call the reference makes
        return per_cam[0][i][None] if C == 1 else torch.stack([pc[i] for pc in per_cam])

    tile_w, tile_h = S.tile_grid(width, height)
    bt = S.bin_tile_for(width, height, cull)
    meta = _Meta({
        "camera_ids": None, "gaussian_ids": None, "opacities": opacities[None].expand(C, N),
        "tile_width": tile_w, "tile_height": tile_h, "tile_size": TILE_SIZE,
        "width": width, "height": height, "n_cameras": C, "cull_mode": cull, "bin_tile": bt,
    })
    tiled = {}

    def tile_lists():
        """Per camera (splat, .., sorted_tile, sorted_gid, tile_offsets, ..) on the published 16 x 16 tiles.  With 32 x 32
        bins in the hot path these are built on first access, by binning once more (nothing in the training loop
        reads them: the reference looks at means2d and radii, gsplat_backend.py:204-208)."""
        if "v" not in tiled:
            if bt == TILE_SIZE:
                tiled["v"] = settled()
            else:
                out16 = []
                for c_ in range(len(per_cam)):
                    pc = cam(c_)
                    splat, radii, tiles = pc[0], pc[1], pc[2]
                    keys = torch.where((radii > 0).all(-1), splat[:, 9].contiguous().view(torch.int32),
                                       torch.full_like(tiles, -1))
                    b = S.bin_tiles(splat, radii, tiles, keys, width, height, cull, exact=True, speculation=speculation)
                    out16.append((splat, radii, tiles, b.sorted_tile, b.sorted_gid, b.tile_offsets, pc[6]))
                tiled["v"] = out16
        return tiled["v"]

    # everything below is a view or is built on first access: the hot path pays for none of it
    meta._lazy = {
        "means2d": lambda: gather(0)[..., 0:2], "conics": lambda: gather(0)[..., 2:5],
        "depths": lambda: gather(0)[..., 9], "radii": lambda: gather(1), "tiles_per_gauss": lambda: gather(2),
        "n_isects": lambda: (sum(int(pc[4].numel()) for pc in settled()) if bt == TILE_SIZE
                             else sum(int(pc[2].sum()) for pc in per_cam)),
        "n_binned": lambda: sum(int(pc[4].numel()) for pc in settled()),   # records the hot path sorted (its bins)
        # (defer_check: did the remembered capacity / key range hold for every camera of this call?  Waits for the control words.)
        "speculation_ok": lambda: all(b.finalize()[1] for b in checked),
        "flatten_ids": lambda: torch.cat([pc[4] + c * N for c, pc in enumerate(tile_lists())]),
        "isect_offsets": lambda: _isect_offsets(tile_lists(), tile_h, tile_w),
        "isect_ids": lambda: _isect_ids(tile_lists(), tile_w * tile_h),
    }
    if C == 1:
        return renders[0][None], alphas[0][None], meta
    return torch.stack(renders), torch.stack(alphas), meta


def _isect_offsets(per_cam, tile_h: int, tile_w: int) -> torch.Tensor:
    offs, base = [], 0
    for pc in per_cam:
        offs.append(pc[5][:-1] + base)
        base += pc[4].numel()
    return torch.stack(offs).reshape(len(per_cam), tile_h, tile_w)


def _isect_ids(per_cam, n_tiles: int) -> torch.Tensor:
    """The 64-bit keys of the published layout (camera | tile | depth bits).  The kernels never need them
    (the sort works on split keys); they are materialised only when asked for."""
    tile_bits = max(1, math.ceil(math.log2(max(n_tiles, 2))))
    parts = []
    for c, pc in enumerate(per_cam):
        depth_bits = pc[0][:, 9].contiguous().view(torch.int32).long() & 0xFFFFFFFF
        gid = pc[4].long()
        # (the one-pass binning stores no tile ids: the lists' offsets say which records are whose)
        tiles = S.tile_ids(pc[3]) if pc[3].numel() == gid.numel() else S.tile_ids_from_offsets(pc[5]).long()
        parts.append((c << (32 + tile_bits)) | (tiles << 32) | depth_bits[gid])
    return torch.cat(parts)


class _Meta(dict):
    """``meta`` dict whose tensor entries are produced on first access (views of the saved buffers, or small
    derived tensors), so that a caller who reads only ``means2d`` / ``radii`` -- like the reference
    (gsplat_backend.py:204-208) -- costs the hot path nothing."""

    _lazy = {}

    def __missing__(self, key):
        if key in self._lazy:
            val = self._lazy[key]()
            self[key] = val
            return val
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        return list(dict.keys(self)) + [k for k in self._lazy if not dict.__contains__(self, k)]
