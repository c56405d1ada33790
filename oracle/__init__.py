"""CPU oracle for the VSRD hot path.  TEST INFRASTRUCTURE ONLY.

This package is a closed-form, pure-PyTorch (CPU, fp32 or fp64) restatement of
the reference algorithm on the path SURVEY.md §8 names: ray casting, oriented
box SDFs and their analytic normals, the temperature soft-min instance union,
stratified + inverse-transform sampling, NeuS-style opacity, front-to-back
compositing, multi-view box projection and the silhouette / projection /
eikonal losses.  Each function cites the reference file:line it follows.

Rules (task statement ③):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import it, and only as the checker / the timed CPU
    baseline; nothing under ``vsrd_amd/`` imports it and the product path never
    falls back to it;
  * it is pinned: ``tests/test_oracle_golden.py`` checks it against the golden
    vectors under ``tests/golden/`` which were produced by importing the
    reference's own modules (``tests/golden/make_golden.py``).

Parity status: PINNED for everything that exists in an importable reference
module (rendering, samplers, sdfs, project_box_3d, BoxParameters3D, encoder,
per-instance MLP).  UNPINNED for the two torchvision==0.14.0 ops the reference
calls (``distance_box_iou``/``_loss``, ``clip_boxes_to_image``): torchvision is
not installed in the build image, so ``oracle.geometry.distance_box_iou``
restates the published DIoU definition and is pinned by hand-computed cases
only.

Layout convention: per-sample tensors are *ray-major* ``[R, S']`` here (the HIP
library's native layout); ``oracle.rendering.to_reference_layout`` converts to
the reference's sample-major ``[S', R, 1]``.
"""
from . import fields, rendering, geometry, losses, step  # noqa: F401
