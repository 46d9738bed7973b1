// view.hip -- ONE C-ABI call per view forward and one per view backward (ABI 13; `hot`: ABI 14).
//
// The steady-state chain of a view behind its projection -- bucketed depth ordering, one-pass placement with the compositor's
// launch order, forward sweep; then tile re-ordering + reverse sweep (+ the projection backward) -- is eight to ten kernel
// launches that the Python host used to marshal through five ctypes calls with ~45 pointer arguments, a dozen allocations
// and as many dictionary look-ups per view: ~0.3 ms of host time against ~0.5 ms of device time.  Here the chain's arguments
// live in ONE plain struct (filled once per view; a caller with persistent buffers fills it once per buffer set and only
// changes the token) and the launches are queued back to back from C.  Every stage is the stage of include/gsplat_amd.h
// with the same arguments: the fused calls produce the same bits as the stage-by-stage path (tests/test_hip_parity.py).
//
// Behind /root/reference/src/backends/gsplat_backend.py:168-182 (the one call the reference makes per view).

#include "gsa_common.h"

extern "C" size_t gsa_view_args_size(void) { return sizeof(gsa_view_args); }

extern "C" int gsa_view_fwd(const gsa_view_args *a, gsa_stream_t stream) {
    GSA_RETURN_IF(!a || a->struct_bytes != (uint32_t)sizeof(gsa_view_args), GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(a->n <= 0 || a->m_capacity == 0 || !a->tile_order, GSA_ERR_INVALID_ARGUMENT);
    // (the one-pass placement with the launch order on the side is the only placement this entry point takes: the caller
    // asks gsa_tile_bin16_direct first and runs the stage-by-stage path otherwise)
    GSA_RETURN_IF(!gsa_tile_bin16_direct(a->width, a->height, a->bin_tile), GSA_ERR_UNSUPPORTED);
    int rc = gsa_depth_order_msd(a->n, a->depth_keys, a->tile_rects, a->key_min, a->scale, a->key_range, a->order, a->rects,
                                 a->offsets, a->block_owner, a->n_blocks, a->ctl, a->ctl_host, a->token, a->depth_ws,
                                 a->depth_ws_bytes, stream);
    if (rc != GSA_OK) return rc;
    rc = gsa_tile_bin16_scheduled(a->n, a->order, a->offsets, a->rects, a->block_owner, a->width, a->height, a->bin_tile,
                                  a->m_capacity, a->ctl, nullptr, a->sorted_gid, a->tile_offsets,
                                  a->gid_boxes ? a->qbox : nullptr, a->tile_order, a->bin_ws, a->bin_ws_bytes, stream);
    if (rc != GSA_OK) return rc;
    if (a->wait_event) {   // the splat records of a split projection (written on another stream) are read from here on
        hipError_t e = hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(a->wait_event), 0);
        if (e != hipSuccess) return (int)e;
    }
    return gsa_composite_fwd(a->width, a->height, a->channels, a->splat, a->sorted_gid, a->gid_boxes, a->tile_offsets,
                             a->background, a->tile_order, a->render, a->alpha, a->final_T, a->last_id, a->quad_list,
                             a->quad_count, a->v_splat, a->quad_list ? (long long)a->n + (a->hot ? gsa_hot_rows() : 0) : 0,
                             a->bin_tile, stream);
}

extern "C" int gsa_view_bwd(const gsa_view_args *a, gsa_stream_t stream) {
    GSA_RETURN_IF(!a || a->struct_bytes != (uint32_t)sizeof(gsa_view_args), GSA_ERR_INVALID_ARGUMENT);
    GSA_RETURN_IF(!a->quad_list || !a->quad_count || !a->v_splat || !a->v_render || !a->bwd_tile_order, GSA_ERR_INVALID_ARGUMENT);
    const int tile_w = gsa_div_up(a->width, GSA_TILE_SIZE), tile_h = gsa_div_up(a->height, GSA_TILE_SIZE);
    // (the forward sweep cleared the gradient records on the side: n = 0 -- only the tiles are ordered by their blended records)
    int rc = gsa_composite_bwd_prepare(0, a->v_splat, tile_w * tile_h, a->quad_count, a->bwd_tile_order, stream);
    if (rc != GSA_OK) return rc;
    GSA_RETURN_IF(a->hot && a->sweep_rgb, GSA_ERR_INVALID_ARGUMENT);
    rc = a->hot
             ? gsa_composite_bwd_lists_hot(a->width, a->height, a->channels, a->splat, a->sorted_gid, a->gid_boxes, a->tile_offsets,
                                           a->background, a->bwd_tile_order, a->final_T, a->last_id, a->v_render, a->v_alpha,
                                           a->quad_list, a->quad_count, (long long)a->n, a->v_splat, a->bin_tile, stream)
         : a->sweep_rgb
             ? gsa_composite_bwd_lists_rgb(a->width, a->height, a->channels, a->splat, a->sorted_gid, a->gid_boxes, a->tile_offsets,
                                           a->background, a->bwd_tile_order, a->final_T, a->last_id, a->v_render, a->v_alpha,
                                           a->quad_list, a->quad_count, a->v_splat, a->sweep_rgb, a->bin_tile, stream)
             : gsa_composite_bwd_lists(a->width, a->height, a->channels, a->splat, a->sorted_gid, a->gid_boxes, a->tile_offsets,
                                       a->background, a->bwd_tile_order, a->final_T, a->last_id, a->v_render, a->v_alpha,
                                       a->quad_list, a->quad_count, a->v_splat, a->bin_tile, stream);
    if (rc != GSA_OK || a->project_mode == 0) return rc;
    if (a->wait_bucket) {   // the gradient bucket is summed into in view order: the previous view's accumulation
        hipError_t e = hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(a->wait_bucket), 0);
        if (e != hipSuccess) return (int)e;
    }
    if (a->project_mode == 2)
        return gsa_project_bwd_factored(a->n, a->means, a->quats, a->scales, a->opacities, a->colors, a->sh_degree, a->sh_coeffs,
                                        a->viewmat, a->intrin, a->width, a->height, a->splat, a->radii, a->v_splat, a->v_means,
                                        a->v_quats, a->v_scales, a->v_opacities, a->v_rgb, a->accumulate, a->sh_aux, stream);
    return gsa_project_bwd_acc(a->n, a->means, a->quats, a->scales, a->opacities, a->colors, a->sh_degree, a->sh_coeffs,
                               a->viewmat, a->intrin, a->width, a->height, a->splat, a->radii, a->v_splat, a->v_means,
                               a->v_quats, a->v_scales, a->v_opacities, a->v_colors, a->accumulate, a->sh_aux, stream);
}
