"""CPU oracle for the differentiable 3D-Gaussian rasterizer hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and there
only as the checker.  The product path (``3d-gaussian-splatting_amd/``) never imports this
package and fails loudly when its HIP library is missing.

PARITY UNPINNED.  The reference implements no rasterizer arithmetic of its own: the hot path
is one call into the third-party, un-vendored ``gsplat==1.5.3`` wheel
(/root/reference/requirements.txt:9, called at /root/reference/src/backends/gsplat_backend.py:168-182),
its CPU backend is a stub (/root/reference/src/backends/pytorch_backend.py:1-5), and none of its
tests touch rendering.  gsplat is not installed in the authoring container.
This oracle therefore restates the published gsplat 1.5.x algorithm and
is anchored on the reference's own call site and return contract; no golden vector from the
reference pins its arithmetic.  What IS pinned by the reference: the training-step harness
around the path (loss, SSIM, PSNR, Adam step), captured by importing the reference trainer in
the authoring container with this oracle standing in for ``gsplat``
(tests/golden/make_harness_golden.py).
"""

from .torch_oracle import rasterization, OracleBackend  # noqa: F401
