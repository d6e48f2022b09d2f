"""MI355X-native batched differentiable FK + planning-objective engine.

Drop-in for the hot path of anindex/torch_robotics (FK -> costs -> gradient over
(batch x horizon x DOF) rollouts); see DESIGN.md.  All compute runs in the HIP
library `csrc/libtrk.so` through the C ABI in `include/trk.h`; there is no CPU
compute path in this package.
"""
from .kinmodel import KinModel  # noqa: F401
from .costmodel import CostModelSpec  # noqa: F401
from .kinematics import (DifferentiableTree, DifferentiableFrankaPanda, DifferentiableUR10,  # noqa: F401
                         DifferentiableKUKAiiwa, DifferentiableAllegroHand, DifferentiableShadowHand,
                         DifferentiableHabitatStretch, DifferentiableTiagoDualHoloMove,
                         DifferentiableUR10Allegro, DifferentiableDualPanda, Frame, x_rot, y_rot, z_rot,
                         q_to_rotation_matrix, link_pos_from_link_tensor, link_rot_from_link_tensor,
                         link_quat_from_link_tensor)
from .environments import (MultiSphereField, MultiBoxField, MultiSharpBoxField, ObjectField, GridMapSDF,  # noqa: F401
                           EnvBase, EnvSpheres3D, EnvSpheres3DExtraObjects, EnvTableShelf, EnvMazeBoxes3D,
                           GraspedObject, GraspedObjectPandaBox)
from .fields import (DistanceField, CollisionSelfField, CollisionObjectDistanceField,  # noqa: F401
                     CollisionWorkspaceBoundariesDistanceField, EESE3DistanceField, SE3_distance)
from .robots import RobotBase, RobotPanda, RobotPointMass, RobotPointMass3D, compute_path_length, compute_smoothness, finite_difference_vector  # noqa: F401
from .tasks import PlanningTask, GraphedCostBackward  # noqa: F401

__version__ = "0.1.0"
