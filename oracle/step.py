"""Oracle: one optimisation step of scripts/main.py:323-865 (TEST INFRASTRUCTURE).

CPU restatement used by tests/test_hip_step.py to check that the device loop (vsrd_amd/optimization.py) produces the
same losses and the same optimised box parameters when both consume the same ray indices and uniforms.
"""
import torch

from . import fields, geometry, losses, rendering


class OracleFrame:
    def __init__(self, image_size, intrinsics, extrinsics, soft_masks, boxes_2d, visible_masks, num_samples,
                 num_steps=3000, lr=1.0e-2, gamma=0.01 ** (1.0 / 3000.0), distance_range=(0.0, 100.0)):
        V, H, W, N = soft_masks.shape
        self.image_size, self.K, self.E = image_size, intrinsics, extrinsics
        self.soft_masks, self.boxes_2d, self.visible = soft_masks.reshape(-1, N), boxes_2d, visible_masks
        self.S, self.num_steps, self.range = num_samples, num_steps, distance_range
        self.raw = [torch.zeros(N, 3, requires_grad=True), torch.zeros(N, 3, requires_grad=True),
                    torch.tensor([1.0, 0.0]).repeat(N, 1).requires_grad_(True)]          # box_parameters.py:34-45
        self.optimizer = torch.optim.Adam([dict(params=[p], lr=lr) for p in self.raw], lr=lr)
        self.scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, gamma=gamma)
        cam, dirs = geometry.ray_casting((H, W), intrinsics, extrinsics)                   # main.py:267-278
        self.cam, self.dirs, self.pixels = cam, dirs.reshape(-1, 3), H * W
        self.step_index = 0

    def enable_residual(self, hypernetwork, embeddings, embedding_lr=1.0e-3, hypernetwork_lr=1.0e-4):
        """Post-warm-up phase (main.py:525-578): residual MLP weights = hypernetwork(embeddings); both are optimised."""
        self.hypernetwork, self.embeddings = hypernetwork, embeddings
        self.optimizer.add_param_group(dict(params=[embeddings], lr=embedding_lr))
        self.optimizer.add_param_group(dict(params=list(hypernetwork.parameters()), lr=hypernetwork_lr))
        self.scheduler.base_lrs = [g["lr"] for g in self.optimizer.param_groups]

    def step(self, ray_indices, u_coarse, u_fine, residual=False, conditioned_only=False):
        step = self.step_index
        self.optimizer.zero_grad()
        loc, dim, rot, corners = geometry.decode_box_parameters(*self.raw)
        pd_boxes, _ = geometry.project_boxes_multi_view(corners, self.E, self.K, self.image_size)      # main.py:339-362
        pd_idx, gt_idx = losses.match_instances(pd_boxes[0], self.boxes_2d[0])                         # main.py:374-386
        iou, l1 = losses.projection_losses(pd_boxes, self.boxes_2d, self.visible, pd_idx, gt_idx)      # main.py:391-415
        ratio, temperature, std = losses.schedules(step, self.num_steps)                               # main.py:420-431
        mlp = self.hypernetwork(self.embeddings) if residual else None
        union = fields.InstanceUnion(loc, rot, dim, temperature, mlp)
        origins = self.cam[ray_indices // self.pixels]
        fine = rendering.hierarchical_render(union, origins, self.dirs[ray_indices], self.range, self.S, std, ratio, u_coarse, u_fine)
        sil = losses.silhouette_loss(fine.labels, self.soft_masks[ray_indices], pd_idx, gt_idx)        # main.py:653-671
        w = losses.LOSS_WEIGHTS
        total = w["iou_projection_loss"] * iou + w["l1_projection_loss"] * l1 + w["silhouette_loss"] * sil   # main.py:855
        eik = None
        if residual:
            eik = losses.eikonal_loss(fine.gradients)                                                  # main.py:679-687
            total = total + w["eikonal_loss"] * eik
        total.backward()
        raw_gradients = [p.grad.detach().clone() for p in self.raw]
        self.optimizer.step()
        self.scheduler.step()
        self.step_index += 1
        return dict(iou_projection_loss=iou.detach(), l1_projection_loss=l1.detach(), silhouette_loss=sil.detach(), loss=total.detach(),
                    matching=(pd_idx, gt_idx), raw_gradients=raw_gradients, eikonal_loss=None if eik is None else eik.detach())
