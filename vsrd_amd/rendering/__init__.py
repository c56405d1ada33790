"""Mirror of the ``vsrd.rendering`` call surface used by scripts/main.py."""
from . import sdfs
from .sdfs import box, rotation, translation, hard_union, soft_union
from .renderers import (hierarchical_volumetric_rendering, render_hierarchical, render_at_distances, evaluate_field,
                        sphere_tracing, surface_normal, sphere_intersection, phong_shading, shadow_rendering, silhouette_step,
                        workspace_scope, Workspace)
from .samplers import quadrature_sampler, inverse_transform_sampler, importance_merge, sample_rays, RayTable
from .utils import ray_casting
