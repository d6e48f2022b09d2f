"""Scene description: analytic SDF primitives, posed objects, voxel SDF grid, 3-D scenes.

Mirrors the data side of `torch_robotics/environments/` (primitives.py, grid_map_sdf.py,
env_base.py, env_spheres_3d.py, env_table_shelf.py, env_maze_boxes_3d.py,
env_spheres_3d_extra_objects.py).  Objects hold plain numbers; every distance evaluation goes to
the HIP kernels through a `CostHandle` (there is no torch arithmetic here).
"""
from __future__ import annotations

import itertools
from copy import copy
from typing import List

import numpy as np
import torch

from . import ops
from .costmodel import panda_box_base_points, CostModelSpec, box_prims, grid_object, make_object, sphere_prims
from .kinmodel import quat_wxyz_to_rot

DEFAULT_TENSOR_ARGS = {"device": torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"),
                       "dtype": torch.float32}


def _np(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


_version_counter = itertools.count(1)


def scene_version(df_obj_list) -> tuple:
    """What a device cost model built from these objects depends on: identity AND pose version of every ObjectField, identity
    of a grid's tensors.  The reference evaluates each object from its current pose on every call (primitives.py:387-405), so
    every cache of a `CostHandle` is keyed by this and a moved object rebuilds it.  Integers only -- this runs on every cost
    evaluation: an ObjectField takes a fresh version number whenever `pos` / `ori` are assigned (`set_position_orientation`
    or the attributes).  The pose arrays and the primitives' centre / radius / size arrays are read-only (`obj.pos[0] = ...`,
    `field.centers[0] = ...` raise: assign a new array or call `set_position_orientation`), and the key carries the version number
    of every primitive field's latest geometry assignment, so a re-assigned `field.centers` rebuilds the device model as well."""
    return tuple((id(o), id(o.sdf_tensor)) if isinstance(o, GridMapSDF)
                 else (id(o), o._version) + tuple(k for f in o.fields for k in f.array_ids()) for o in df_obj_list)


class PrimitiveShapeField:
    def __init__(self, dim=3, tensor_args=None):
        self.dim = dim
        self.tensor_args = DEFAULT_TENSOR_ARGS if tensor_args is None else tensor_args

    def prims(self) -> List[dict]:
        raise NotImplementedError

    _ARRAYS = ()

    def __setattr__(self, name, value):
        # the geometry arrays are read-only copies: an in-place edit raises instead of leaving a stale device model behind
        if name in self._ARRAYS:
            value = np.array(value, dtype=np.float32)
            value.setflags(write=False)
            object.__setattr__(self, "_geometry_version", next(_version_counter))
        object.__setattr__(self, name, value)

    def array_ids(self) -> tuple:
        """what scene_version adds for this field: the number its latest geometry assignment took"""
        return (getattr(self, "_geometry_version", 0),)

    @ops.host_round_trip
    def compute_signed_distance(self, x):
        return ObjectField([self]).compute_signed_distance(x)


class MultiSphereField(PrimitiveShapeField):                 # primitives.py:88-121
    _ARRAYS = ("centers", "radii")

    def __init__(self, centers, radii, tensor_args=None):
        centers = _np(centers).astype(np.float32)
        super().__init__(dim=centers.shape[-1], tensor_args=tensor_args)
        self.centers, self.radii = centers.reshape(-1, self.dim), _np(radii).astype(np.float32).reshape(-1)

    def prims(self):
        c = self.centers if self.dim == 3 else np.concatenate([self.centers, np.zeros((len(self.centers), 1), np.float32)], 1)
        return sphere_prims(c, self.radii)


class MultiSharpBoxField(PrimitiveShapeField):               # primitives.py:197-228
    rounded = False
    _ARRAYS = ("centers", "sizes", "half_sizes", "radius")

    def __init__(self, centers, sizes, tensor_args=None):
        centers = _np(centers).astype(np.float32)
        super().__init__(dim=centers.shape[-1], tensor_args=tensor_args)
        self.centers, self.sizes = centers.reshape(-1, self.dim), _np(sizes).astype(np.float32).reshape(-1, self.dim)
        self.half_sizes = self.sizes / np.float32(2)

    def prims(self):
        if self.dim != 3:
            raise NotImplementedError("2-D boxes are outside the 3-D hot path")
        return box_prims(self.centers, self.sizes, rounded=self.rounded)


class MultiBoxField(MultiSharpBoxField):                     # rounded boxes, primitives.py:304-334
    rounded = True

    def __init__(self, centers, sizes, tensor_args=None):
        super().__init__(centers, sizes, tensor_args=tensor_args)
        self.radius = self.sizes.min(-1) * np.float32(0.15)


MultiRoundedBoxField = MultiBoxField


class _SDFPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cm):
        sdf, grad = ops.sdf_points(cm, x, want_grad=True)
        ctx.save_for_backward(grad)
        return sdf

    @staticmethod
    def backward(ctx, gs):
        (grad,) = ctx.saved_tensors
        return (gs.unsqueeze(-1) * grad).sum(1), None


class ObjectField(PrimitiveShapeField):                      # primitives.py:346-420
    def __init__(self, primitive_fields, name="object", pos=None, ori=None, reference_frame="base"):
        assert primitive_fields is not None and isinstance(primitive_fields, list)
        super().__init__(dim=primitive_fields[0].dim, tensor_args=primitive_fields[0].tensor_args)
        self.name, self.fields, self.reference_frame = name, primitive_fields, reference_frame
        assert (pos is None and ori is None) or (np.size(_np(pos)) == 3 and np.size(_np(ori)) == 4)
        self.pos = np.zeros(3, np.float32) if pos is None else pos
        self.ori = np.array([1, 0, 0, 0], np.float32) if ori is None else ori

    # pose attributes: every assignment takes a new version number (scene_version) and drops the object's own cost model
    @property
    def pos(self):
        return self._pos

    @pos.setter
    def pos(self, value):
        self._pos = _np(value).astype(np.float32).reshape(3)
        self._pos.setflags(write=False)     # an in-place edit would not be seen by scene_version: it raises instead (assign a new pose)
        self._version, self._cm = next(_version_counter), None

    @property
    def ori(self):
        return self._ori

    @ori.setter
    def ori(self, value):
        self._ori = _np(value).astype(np.float32).reshape(4)
        self._ori.setflags(write=False)
        self._version, self._cm = next(_version_counter), None

    def set_position_orientation(self, pos=None, ori=None):
        if pos is not None:
            assert len(pos) == 3
            self.pos = pos
        if ori is not None:
            assert len(ori) == 4, "quaternion wxyz"
            self.ori = ori

    def as_object(self) -> dict:
        prims = list(itertools.chain.from_iterable(f.prims() for f in self.fields))
        return make_object(prims, self.pos, quat_wxyz_to_rot(self.ori))

    @ops.host_round_trip
    def compute_signed_distance(self, x):
        """x (..., 3) on the GPU -> (...) signed distance (differentiable w.r.t. x)."""
        # keyed by the device, the pose version AND the primitive fields' geometry versions: `field.centers = new` must be seen here
        key = (x.device, self._version) + tuple(k for f in self.fields for k in f.array_ids())
        if self._cm is None or self._cm_key != key:
            self._cm = ops.CostHandle(CostModelSpec(n_links_in=1, objects=[self.as_object()]), x.device)
            self._cm_key = key
        flat = x.reshape(-1, 3)
        if torch.is_grad_enabled() and x.requires_grad:
            return _SDFPoints.apply(flat.contiguous(), self._cm).reshape(x.shape[:-1])
        return ops.sdf_points(self._cm, flat).reshape(x.shape[:-1])

    compute_signed_distance_impl = compute_signed_distance


class GraspedObject(ObjectField):                             # objects.py:10-34
    """An object rigidly held by the end effector; pos / ori are relative to `reference_frame` (a robot link)."""

    def __init__(self, primitive_fields, **kwargs):
        assert len(primitive_fields) == 1                       # only one primitive type
        super().__init__(primitive_fields, **kwargs)
        if not isinstance(primitive_fields[0], MultiBoxField):
            raise NotImplementedError                           # objects.py:29-30
        self.geometry_size = primitive_fields[0].sizes[0]

    def get_base_points_for_collision(self):
        raise NotImplementedError


class GraspedObjectPandaBox(GraspedObject):                  # objects.py:37-89
    def __init__(self, tensor_args=None, **kwargs):
        fields = [MultiBoxField(np.zeros((1, 3), np.float32), np.array([[0.05, 0.05, 0.15]], np.float32), tensor_args=tensor_args)]
        super().__init__(fields, name="GraspedObjectPandaBox", pos=np.array([0.0, 0.0, 0.11], np.float32),
                         ori=np.array([0, 0.7071081, 0, 0.7071055], np.float32), reference_frame="panda_hand", **kwargs)
        self.base_points_for_collision = self.get_base_points_for_collision()
        self.n_base_points_for_collision = len(self.base_points_for_collision)

    def get_base_points_for_collision(self):
        """The 8 vertices and 6 face centres of the box, in the object frame (objects.py:57-89), fp32."""
        return torch.from_numpy(panda_box_base_points(self.fields[0].sizes[0]))


class GridMapSDF:                                            # grid_map_sdf.py:9-117
    """Voxel SDF + stored gradients; built by the `trk_grid_precompute` kernel."""

    def __init__(self, limits, cell_size, obj_list, tensor_args=None):
        self.tensor_args = DEFAULT_TENSOR_ARGS if tensor_args is None else tensor_args
        self.limits = torch.as_tensor(_np(limits), dtype=torch.float32)
        self.dim = self.limits.shape[-1]
        if self.dim != 3:
            raise NotImplementedError("2-D grids are outside the 3-D hot path")
        self.obj_list = obj_list
        self.cell_size = cell_size
        map_dim = torch.abs(self.limits[1] - self.limits[0])                  # fp32, as grid_map_sdf.py:24-27
        self.map_dim = map_dim
        self.cmap_dim = torch.ceil(map_dim / cell_size).to(torch.int64)
        self.sdf_tensor = None
        self.grad_sdf_tensor = None
        self.precompute_sdf()

    def precompute_sdf(self):
        spec = CostModelSpec(n_links_in=1, objects=[o.as_object() for o in self.obj_list])
        cm = ops.CostHandle(spec, ops.compute_device(self.tensor_args["device"]))      # the grid lives where it is computed and queried: on the GPU
        self.sdf_tensor, self.grad_sdf_tensor = ops.grid_precompute(
            cm, self.cmap_dim.numpy(), self.limits[0].numpy(), self.limits[1].numpy())

    def grid_dict(self) -> dict:
        return dict(dims=self.cmap_dim.numpy().astype(np.int32), lim_min=self.limits[0].numpy(),
                    map_dim=self.map_dim.numpy(), sdf=self.sdf_tensor, grad=self.grad_sdf_tensor)

    def _query_handle(self, device):
        if getattr(self, "_qcm", None) is None or self._qcm.device != torch.device(device):
            spec = CostModelSpec(n_links_in=1, objects=[grid_object()])
            spec.grid = self.grid_dict()
            self._qcm = ops.CostHandle(spec, device)
        return self._qcm

    @ops.host_round_trip
    def compute_signed_distance(self, X, **kwargs):           # grid_map_sdf.py:81-114
        """Nearest-lower-cell lookup; differentiable w.r.t. X with the STORED gradient of that cell, like the reference's
        `sdf[idx] + (X * g).sum() - (X.detach() * g).sum()`."""
        cm = self._query_handle(X.device)
        flat = X.reshape(-1, 3)
        if torch.is_grad_enabled() and X.requires_grad:
            return _SDFPoints.apply(flat.contiguous(), cm).reshape(X.shape[:-1])
        return ops.sdf_points(cm, flat).reshape(X.shape[:-1])

    __call__ = compute_signed_distance


class EnvBase:                                               # env_base.py:17-100
    def __init__(self, name="NameEnvBase", limits=None, obj_fixed_list=None, obj_extra_list=None,
                 precompute_sdf_obj_fixed=False, sdf_cell_size=0.005, tensor_args=None, **kwargs):
        self.tensor_args = DEFAULT_TENSOR_ARGS if tensor_args is None else tensor_args
        self.name = name
        assert limits is not None
        self.limits_np = _np(limits).astype(np.float32)
        self.limits = torch.as_tensor(self.limits_np, **self.tensor_args)
        self.dim = len(self.limits_np[0])
        for lst in (obj_fixed_list, obj_extra_list):
            for obj in lst or []:
                assert isinstance(obj, ObjectField), "Objects must be instances of ObjectField class"
        self.obj_fixed_list, self.obj_extra_list = obj_fixed_list, obj_extra_list
        self.obj_all_list = list(itertools.chain(obj_fixed_list or [], obj_extra_list or []))
        self.grid_map_sdf_obj_fixed = None
        if precompute_sdf_obj_fixed:
            self.grid_map_sdf_obj_fixed = GridMapSDF(self.limits_np, sdf_cell_size, self.obj_fixed_list,
                                                     tensor_args=self.tensor_args)

    def get_obj_list(self):
        return self.obj_all_list

    def get_df_obj_list(self, return_extra_objects_only=False):          # env_base.py:75-88
        out = []
        if not return_extra_objects_only:
            if self.grid_map_sdf_obj_fixed is not None:
                out.append(self.grid_map_sdf_obj_fixed)
            else:
                out.extend(self.obj_fixed_list or [])
        if self.obj_extra_list is not None:
            out.extend(self.obj_extra_list)
        return out

    @ops.host_round_trip
    def compute_sdf(self, x, reshape_shape=None):             # env_base.py:140-169
        """Signed distance of the scene (min over the fixed objects -- or their precomputed grid -- and the extra objects) at
        points x (..., 3); differentiable w.r.t. x.  One `trk_sdf_points` launch for all objects."""
        objs = self.get_df_obj_list()
        if not objs:
            return None
        key = (str(x.device), scene_version(objs))
        if getattr(self, "_sdf_cm", None) is None or self._sdf_cm[0] != key:
            spec = CostModelSpec(n_links_in=1)
            spec.objects, spec.grid = objects_to_spec_parts(objs)
            self._sdf_cm = (key, ops.CostHandle(spec, x.device))
        cm = self._sdf_cm[1]
        flat = x.reshape(-1, 3)
        if torch.is_grad_enabled() and x.requires_grad:
            per_obj = _SDFPoints.apply(flat.contiguous(), cm)
        else:
            per_obj = ops.sdf_points(cm, flat)
        sdf = per_obj.reshape(flat.shape[0], -1).min(dim=1).values
        return sdf.reshape(reshape_shape) if reshape_shape else sdf.reshape(x.shape[:-1])

    def add_obj(self, obj):                                   # env_base.py:90-92
        raise NotImplementedError

    def zero_grad(self):
        pass


def objects_to_spec_parts(df_obj_list):
    """[ObjectField | GridMapSDF, ...] -> (objects list, grid dict or None) for a CostModelSpec."""
    objects, grid = [], None
    for df in df_obj_list:
        if isinstance(df, GridMapSDF):
            objects.append(grid_object())
            grid = df.grid_dict()
        else:
            objects.append(df.as_object())
    return objects, grid


class EnvSpheres3D(EnvBase):                                 # env_spheres_3d.py:11-49 (scene data)
    def __init__(self, name="EnvDense2D", tensor_args=None, **kwargs):
        centers = np.array([[-0.3, 0.3, 0.85], [-0.35, -0.25, 0.45], [-0.45, 0.15, 0.0], [0.45, 0.35, 0.0],
                            [0.55, 0.35, 0.55], [0.65, -0.4, 0.25], [0.2, -0.35, 0.5], [0.35, 0.0, 0.9],
                            [0.0, -0.3, 0.0], [0.0, 0.45, 0.35]], np.float32)
        spheres = MultiSphereField(centers, np.full(10, 0.15, np.float32), tensor_args=tensor_args)
        super().__init__(name=name, limits=np.array([[-1, -1, -1], [1, 1, 1]], np.float32),
                         obj_fixed_list=[ObjectField([spheres], "spheres")], tensor_args=tensor_args, **kwargs)


class EnvSpheres3DExtraObjects(EnvSpheres3D):               # env_spheres_3d_extra_objects.py:12-58
    def __init__(self, tensor_args=None, **kwargs):
        extra = MultiSphereField(np.array([[0.25, 0.0, 0.0], [0.0, 0.5, 0.5], [0.0, -0.5, 0.0], [-0.25, -0.5, 0.5],
                                           [-0.25, 0.0, 0.75]]), np.full(5, 0.15), tensor_args=tensor_args)
        super().__init__(name=self.__class__.__name__, obj_extra_list=[ObjectField([extra], "extra-objects")],
                         tensor_args=tensor_args, **kwargs)


def create_table_object_field(tensor_args=None):             # env_table_shelf.py:14-20
    return ObjectField([MultiBoxField(np.array([(0.0, 0.0, 0.0)]), np.array([(0.56, 0.90, 0.80)]), tensor_args=tensor_args)], "table")


def create_shelf_field(tensor_args=None):                    # env_table_shelf.py:23-69
    width, height, depth, side = 0.80, 2.05, 0.28, 0.02
    shelf_width, shelf_height, shelf_depth = width - 2 * side, 0.015, depth
    centers = [(side / 2, depth / 2, height / 2)]
    sizes = [(side, depth, height)]
    centers.append((side + shelf_width + side / 2, depth / 2, height / 2)); sizes.append((side, depth, height))
    centers.append((side + shelf_width / 2, depth + side / 2, height / 2)); sizes.append((shelf_width, side, height))
    centers.append((side + shelf_width / 2, depth / 2, shelf_height / 2)); sizes.append((shelf_width, shelf_depth, shelf_height))
    centers.append((side + shelf_width / 2, depth / 2, height - shelf_height / 2)); sizes.append((shelf_width, shelf_depth, shelf_height))
    centers.append((side + shelf_width / 2, depth / 2, 0.82 + shelf_height / 2)); sizes.append((shelf_width, shelf_depth, shelf_height))
    for plus_height in [0.23, 0.255, 0.225, 0.225]:
        center = list(copy(centers[-1]))
        center[-1] += plus_height
        centers.append(center); sizes.append((shelf_width, shelf_depth, shelf_height))
    return ObjectField([MultiBoxField(np.array(centers), np.array(sizes), tensor_args=tensor_args)], "shelf")


class EnvTableShelf(EnvBase):                                # env_table_shelf.py:72-102
    def __init__(self, tensor_args=None, **kwargs):
        table = create_table_object_field(tensor_args)
        ts = table.fields[0].sizes[0]
        d_table, theta = 0.10, np.deg2rad(90)
        table.set_position_orientation(pos=(d_table + ts[1].item() / 2, 0, -ts[2].item() / 2),
                                       ori=[np.cos(theta / 2), 0, 0, np.sin(theta / 2)])
        shelf = create_shelf_field(tensor_args)
        shelf.set_position_orientation(pos=(d_table, 0.15 + ts[0].item() / 2, -ts[2].item()))
        super().__init__(name=self.__class__.__name__, limits=np.array([[-1, -1, -1], [1.5, 1.0, 1.5]], np.float32),
                         obj_fixed_list=[table, shelf], tensor_args=tensor_args, **kwargs)


def create_3d_rectangles_objects(tensor_args=None):          # env_maze_boxes_3d.py:11-52
    hs = 0.2
    rects = [(-0.75, -1.0, -0.75 + hs, -0.3), (-0.75, 0.3, -0.75 + hs, 1.0), (-0.75, -0.15, -0.75 + hs, 0.15),
             (-hs / 2 - 0.2, -1, hs / 2 - 0.2, -1 + 0.05), (-hs / 2 - 0.2, -1 + 0.2, hs / 2 - 0.2, 1 - 0.2),
             (-hs / 2 - 0.2, 1 - 0.05, hs / 2 - 0.2, 1),
             (-hs / 2 + 0.2, -1.0, hs / 2 + 0.2, -0.3), (-hs / 2 + 0.2, 0.3, hs / 2 + 0.2, 1.0),
             (-hs / 2 + 0.2, -0.15, hs / 2 + 0.2, 0.15),
             (0.75 - hs, -1, 0.75, -0.6), (0.75 - hs, -0.5, 0.75, -0.2), (0.75 - hs, -0.1, 0.75, 0.1),
             (0.75 - hs, 0.2, 0.75, 0.5), (0.75 - hs, 0.6, 0.75, 1)]
    centers, sizes = [], []
    for bl_x, bl_y, tr_x, tr_y in rects:
        centers.append((bl_x + abs(tr_x - bl_x) / 2, bl_y + abs(tr_y - bl_y) / 2, 0))
        sizes.append((abs(tr_x - bl_x), abs(tr_y - bl_y), 1.95))
    return [ObjectField([MultiBoxField(np.array(centers), np.array(sizes), tensor_args=tensor_args)], "boxes")]


class EnvMazeBoxes3D(EnvBase):                               # env_maze_boxes_3d.py:55-66
    def __init__(self, tensor_args=None, **kwargs):
        super().__init__(name=self.__class__.__name__, limits=np.array([[-1, -1, -1], [1, 1, 1]], np.float32),
                         obj_fixed_list=create_3d_rectangles_objects(tensor_args), tensor_args=tensor_args, **kwargs)
