"""Plugin surface: a backend with the interface of the reference's ``GSplatBackend``
(/root/reference/src/backends/gsplat_backend.py:7-210), running on the HIP library.

Constructed as ``HipBackend(K)`` with ``K`` the [3,3] pixel-unit pinhole intrinsics
(/root/reference/src/gaussians/gaussian_rasterizer.py:109) and reached by the reference's callers as
``rasterizer.backend`` (/root/reference/src/training/trainer.py:417,471; /root/reference/src/eval.py:180).
"""

from __future__ import annotations

from typing import Dict, Optional

import torch

from .rasterization import rasterization
from .stages import Speculation

# render_mode -> ((result key, channel selector), ...)   (gsplat_backend.py:190-201)
_MODE_OUTPUTS = {
    "RGB": (("render", slice(0, 3)),),
    "D": (("accumulated_depth", 0),),
    "ED": (("expected_depth", 0),),
    "RGB+D": (("render", slice(0, 3)), ("accumulated_depth", 3)),
    "RGB+ED": (("render", slice(0, 3)), ("expected_depth", 3)),
}


class HipBackend:
    def __init__(self, K: torch.Tensor):
        self.rasterization = rasterization
        self.K = K
        # what the asynchronous binning remembers about the frames THIS backend renders (record capacity, depth-key range):
        # a trainer's backend and an evaluator's backend of the same frame size each keep their own
        self.speculation = Speculation()

    def _extra(self) -> Dict:
        # (an injected operator -- a test's oracle, the published gsplat -- has no such keyword)
        return {"speculation": self.speculation} if self.rasterization is rasterization else {}

    # -- the live entry point (gsplat_backend.py:129-210) -------------------------------------------
    def render_with_depth(self, gaussian_params: Dict, viewpoint: Dict, bg_color: Optional[torch.Tensor] = None,
                          render_mode: str = "RGB+ED", device: str = "cuda", sh_degree: Optional[int] = None) -> Dict:
        if render_mode not in _MODE_OUTPUTS:
            raise ValueError(f"unknown render_mode {render_mode!r}")
        if bg_color is None:
            bg_color = torch.zeros(3, device=device)
        opacities = gaussian_params["opacities"]
        if opacities.dim() == 2:
            opacities = opacities.squeeze(-1)
        colors = gaussian_params.get("colors", gaussian_params.get("shs"))
        if sh_degree is None and "shs" in gaussian_params:
            sh_degree = 3
        viewmat = viewpoint["world_view_transform"]
        renders, alphas, meta = self.rasterization(
            means=gaussian_params["means3D"], quats=gaussian_params["rotations"], scales=gaussian_params["scales"],
            opacities=opacities, colors=colors, viewmats=viewmat[None], Ks=self.K.to(viewmat.device)[None],
            width=viewpoint["image_width"], height=viewpoint["image_height"], packed=False,
            render_mode=render_mode, sh_degree=sh_degree, backgrounds=bg_color[None], **self._extra())
        image = renders[0]
        result = {"alpha": alphas[0]}
        for key, channel in _MODE_OUTPUTS[render_mode]:
            result[key] = image[..., channel]
        result["means2d"] = meta["means2d"][0]
        result["radii"] = meta["radii"][0]
        result["visibility_filter"] = result["radii"] > 0
        return result

    # -- surface completeness; unused by the reference's live callers -----------
    def render_single(self, gaussian_params: Dict, viewpoint: Dict, bg_color: torch.Tensor) -> Dict:
        """gsplat_backend.py:57-127: RGB + expected depth at SH degree 3."""
        degree = 3 if gaussian_params.get("shs") is not None else None
        params = gaussian_params if degree is not None else {**gaussian_params, "colors": gaussian_params.get(
            "colors", gaussian_params.get("features"))}
        r = self.render_with_depth(params, viewpoint, bg_color, "RGB+ED", str(bg_color.device), degree)
        return {"render": r["render"], "alpha": r["alpha"], "depth": r["expected_depth"], "means2d": r["means2d"],
                "radii": r["radii"], "visibility_filter": r["visibility_filter"]}

    def render_batch(self, gaussian_params: Dict, view_matrices: torch.Tensor, proj_matrices: torch.Tensor,
                     width: int, height: int, bg_color: torch.Tensor, device: str) -> Dict:
        """gsplat_backend.py:12-55 (there a loop of single views with a "real batch processing" TODO): ONE operator call for
        the B cameras -- the gaussians are projected for all of them in one launch (geometry and SH rows read once); per
        camera the same image as ``render_single``; images returned channel-first."""
        n_views = view_matrices.shape[0]
        if n_views == 0:   # (the reference's loop over no views: empty stacks)
            dev = gaussian_params["means3D"].device
            return {"images": torch.empty(0, 3, height, width, device=dev), "alphas": torch.empty(0, 1, height, width, device=dev)}
        degree = 3 if gaussian_params.get("shs") is not None else None
        # the colours by render_with_depth's rule ('colors' wins over 'shs' when a dict carries both), so that a camera of the
        # batch gets the image render_single gives it
        colors = gaussian_params.get("colors", gaussian_params.get("shs")) if degree is not None else \
            gaussian_params.get("colors", gaussian_params.get("features"))
        opacities = gaussian_params["opacities"]
        if opacities.dim() == 2:
            opacities = opacities.squeeze(-1)
        renders, alphas, _ = self.rasterization(
            means=gaussian_params["means3D"], quats=gaussian_params["rotations"], scales=gaussian_params["scales"],
            opacities=opacities, colors=colors, viewmats=view_matrices,
            Ks=self.K.to(view_matrices.device)[None].expand(n_views, 3, 3).contiguous(), width=width, height=height,
            packed=False, render_mode="RGB+ED", sh_degree=degree, backgrounds=bg_color[None].expand(n_views, 3),
            **self._extra())
        return {"images": renders[..., :3].permute(0, 3, 1, 2).contiguous(), "alphas": alphas.permute(0, 3, 1, 2).contiguous()}
