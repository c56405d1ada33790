"""Mirror of the ``vsrd.operations`` names used on the hot path."""
from .geometric_operations import (LINE_INDICES, project_box_3d, project_boxes_multi_view, rotation_matrix_x, expand_to_4x4,
                                   clip_lines_to_front)
from .kitti360_operations import box_3d_iou
