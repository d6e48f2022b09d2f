"""Host-side description of the planning objectives handed to the HIP kernels.

A `CostModelSpec` is the flat form of what the reference spreads over
`RobotBase` (collision link indices, margins, self-collision pair table --
robots/robot_base.py:57-141), the environment's `ObjectField`s / `GridMapSDF`
(environments/primitives.py, grid_map_sdf.py), the workspace box
(tasks.py:71-82) and `EESE3DistanceField` (distance_fields.py:335-359).
It maps 1:1 onto `TrkCostModelDesc` (include/trk.h).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from ._abi import PRIM_ROUNDED_BOX, PRIM_SHARP_BOX, PRIM_SPHERE


def sphere_prims(centers, radii) -> List[dict]:
    """MultiSphereField(centers, radii) primitives.py:90-112."""
    centers = np.asarray(centers, np.float32).reshape(-1, 3)
    radii = np.broadcast_to(np.asarray(radii, np.float32).reshape(-1), (centers.shape[0],))
    return [dict(type=PRIM_SPHERE, center=c, radius=r) for c, r in zip(centers, radii)]


def box_prims(centers, sizes, rounded=True) -> List[dict]:
    """MultiBoxField (rounded, radius = 0.15*min(size), primitives.py:315-334) or
    MultiSharpBoxField (primitives.py:199-223).  Arithmetic in fp32 like the reference."""
    centers = np.asarray(centers, np.float32).reshape(-1, 3)
    sizes = np.asarray(sizes, np.float32).reshape(-1, 3)
    half = sizes / np.float32(2)
    prims = []
    for c, s, h in zip(centers, sizes, half):
        if rounded:
            prims.append(dict(type=PRIM_ROUNDED_BOX, center=c, half=h,
                              radius=np.float32(s.min() * np.float32(0.15))))
        else:
            prims.append(dict(type=PRIM_SHARP_BOX, center=c, half=h, radius=np.float32(0)))
    return prims


def make_object(prims: List[dict], pos=None, R=None) -> dict:
    """ObjectField(primitive_fields, pos, ori) primitives.py:346-405 with `ori` already a rotation matrix."""
    return dict(pos=np.zeros(3, np.float32) if pos is None else np.asarray(pos, np.float32).reshape(3),
                R=np.eye(3, dtype=np.float32) if R is None else np.asarray(R, np.float32).reshape(3, 3),
                prims=list(prims), is_grid=0)


def grid_object() -> dict:
    return dict(pos=np.zeros(3, np.float32), R=np.eye(3, dtype=np.float32), prims=[], is_grid=1)


@dataclass
class CostModelSpec:
    n_links_in: int
    obj_link_idx: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    obj_link_margin: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    objects: List[dict] = field(default_factory=list)
    grid: Optional[dict] = None          # dims[3], lim_min[3], map_dim[3], sdf, grad (arrays/tensors)
    ws_min: Optional[np.ndarray] = None
    ws_max: Optional[np.ndarray] = None
    self_link_idx: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    self_pairs: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.int32))
    self_margin: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    ee_link: int = -1
    ee_w_pos: float = 1.0
    ee_w_rot: float = 1.0
    ee_square: bool = True
    ee_target: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32))
    ee2_link: int = -1                   # a second tracked link (two-arm scenes), same weights / square flag
    ee2_target: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32))
    # FIELD_* mask: fields evaluated with clamp_sdf=True, i.e. relu(margin - signed distance) per link / pair
    # (distance_fields.py:114-117) -- the hinge form a planner optimises
    clamp_fields: int = 0
    # interpolate_link_pos (distance_fields.py:66-69, 145-147): extra position columns n_links_in + k =
    # virtual_w[k, 0] * column virtual_src[k, 0] + virtual_w[k, 1] * column virtual_src[k, 1]  (see interpolation_table)
    virtual_src: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.int32))
    virtual_w: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.float32))

    @property
    def n_columns(self) -> int:
        """Real position columns plus the virtual (interpolated) ones: what the field index tables may address."""
        return int(self.n_links_in) + int(np.asarray(self.virtual_src).reshape(-1, 2).shape[0])

    def add_virtual_columns(self, src: np.ndarray, w: np.ndarray) -> np.ndarray:
        """Append interpolated columns; returns their column indices."""
        src, w = np.asarray(src, np.int32).reshape(-1, 2), np.asarray(w, np.float32).reshape(-1, 2)
        first = self.n_columns
        self.virtual_src = np.concatenate([np.asarray(self.virtual_src, np.int32).reshape(-1, 2), src])
        self.virtual_w = np.concatenate([np.asarray(self.virtual_w, np.float32).reshape(-1, 2), w])
        return np.arange(first, first + len(src), dtype=np.int32)

    def validate(self) -> None:
        L = self.n_columns
        vs = np.asarray(self.virtual_src).reshape(-1, 2)
        if vs.shape != np.asarray(self.virtual_w).reshape(-1, 2).shape:
            raise ValueError("virtual_src / virtual_w shape mismatch")
        if vs.size and (vs.min() < 0 or vs.max() >= self.n_links_in):
            raise ValueError("virtual_src must name real position columns")
        for name in ("obj_link_idx", "self_link_idx"):
            idx = np.asarray(getattr(self, name))
            if idx.size and (idx.min() < 0 or idx.max() >= L):
                raise ValueError(f"{name} out of range for {L} links")
        if len(self.obj_link_idx) != len(self.obj_link_margin):
            raise ValueError("obj_link_idx / obj_link_margin length mismatch")
        pairs = np.asarray(self.self_pairs).reshape(-1, 2)
        if pairs.shape[0] != len(self.self_margin):
            raise ValueError("self_pairs / self_margin length mismatch")
        if pairs.size and (pairs.min() < 0 or pairs.max() >= len(self.self_link_idx)):
            raise ValueError("self_pairs index out of range")
        if self.ee_link >= self.n_links_in or self.ee2_link >= self.n_links_in:
            raise ValueError("ee_link out of range")
        if self.ee2_link >= 0 and self.ee_link < 0:
            raise ValueError("ee2_link needs ee_link")
        if (self.ws_min is None) != (self.ws_max is None):
            raise ValueError("ws_min and ws_max must be given together")
        n_grid = sum(int(o.get("is_grid", 0)) for o in self.objects)
        if n_grid > 1 or (n_grid == 1) != (self.grid is not None):
            raise ValueError("exactly one grid object is required when `grid` is set")


def interpolation_table(n_in: int, n_out: int):
    """`interpolate_points_v1(points, n_out)` (distance_fields.py:66-69) = F.interpolate(mode='linear', align_corners=True)
    along the link axis, unrolled: output point k = w[k, 0] * point src[k, 0] + w[k, 1] * point src[k, 1].
    Restates ATen's index / weight computation in fp32 (UpSample.h: area_pixel_compute_scale -> (in - 1) / (out - 1) for
    align_corners, 0 when out == 1; source index = scale * k; i0 = min(int(src), in - 1), lambda1 = clamp(src - i0, 0, 1),
    i1 = min(i0 + 1, in - 1), lambda0 = 1 - lambda1), checked against torch itself in tests/test_host_api_cpu.py."""
    n_in, n_out = int(n_in), int(n_out)
    if n_in < 1 or n_out < 1:
        raise ValueError("interpolation_table: sizes must be positive")
    scale = np.float32(n_in - 1) / np.float32(n_out - 1) if n_out > 1 else np.float32(0)
    real = (scale * np.arange(n_out, dtype=np.float32)).astype(np.float32)
    i0 = np.minimum(real.astype(np.int64), n_in - 1)
    lam1 = np.clip(real - i0.astype(np.float32), np.float32(0), np.float32(1)).astype(np.float32)
    i1 = np.minimum(i0 + 1, n_in - 1)
    return (np.stack([i0, i1], 1).astype(np.int32),
            np.stack([np.float32(1) - lam1, lam1], 1).astype(np.float32))


# ----------------------------------------------------------------------------------------------------------------------
# Collision point layouts shared by the robots (robots.py) and the kernel generator (codegen.py)
# ----------------------------------------------------------------------------------------------------------------------
def panda_box_base_points(size=(0.05, 0.05, 0.15)) -> np.ndarray:
    """GraspedObjectPandaBox.get_base_points_for_collision (objects.py:57-89): the 8 vertices and 6 face centres of the
    box in the object frame, fp32."""
    x, y, z = (np.asarray(size, np.float32) / np.float32(2)).tolist()
    vertices = [[x, y, -z], [x, -y, -z], [-x, -y, -z], [-x, y, -z], [x, y, z], [x, -y, z], [-x, -y, z], [-x, y, z]]
    faces = [[x, 0, 0], [0, -y, 0], [-x, 0, 0], [0, y, 0], [0, 0, z], [0, 0, -z]]
    return np.asarray(vertices + faces, np.float32)


def load_link_spheres(path, name_to_idx):
    """Link-sphere table `{link: [[x, y, z, r], ...]}` (the format of the reference's
    data/configs/panda/panda_sphere_config.yaml) -> (link_idx [S], offsets [S,3], radii [S], owner names [S])."""
    import yaml
    with open(path) as fh:
        table = yaml.safe_load(fh)
    link, off, rad, names = [], [], [], []
    for name, rows in table.items():
        if not isinstance(rows, list):
            continue
        for row in rows:
            link.append(name_to_idx[name]); off.append(row[:3]); rad.append(row[3]); names.append(name)
    return np.asarray(link, np.int32), np.asarray(off, np.float32).reshape(-1, 3), np.asarray(rad, np.float32), names


def link_sorted_point_set(order, sphere_link=None, sphere_offset=None):
    """Column layout of a robot with a link-sphere model: for every link in walk order (`order` = KinModel.order) its
    origin, then its spheres in table order.  Returns (point_link [P], point_offset [P,3], origin_col [L],
    sphere_col [S]).  With no spheres the layout is simply the link origins IN FILE ORDER (the reference's
    fk_map_collision), not walk order."""
    L = len(order)
    if sphere_link is None or len(sphere_link) == 0:
        return (np.arange(L, dtype=np.int32), np.zeros((L, 3), np.float32), np.arange(L, dtype=np.int32),
                np.zeros(0, np.int32))
    sphere_link = np.asarray(sphere_link, np.int32)
    sphere_offset = np.asarray(sphere_offset, np.float32).reshape(-1, 3)
    pl, po = [], []
    origin_col = np.zeros(L, np.int32)
    sphere_col = np.zeros(len(sphere_link), np.int32)
    for i in (int(v) for v in order):
        origin_col[i] = len(pl)
        pl.append(i); po.append((0.0, 0.0, 0.0))
        for k in np.nonzero(sphere_link == i)[0]:
            sphere_col[k] = len(pl)
            pl.append(i); po.append(tuple(float(v) for v in sphere_offset[k]))
    return np.asarray(pl, np.int32), np.asarray(po, np.float32).reshape(-1, 3), origin_col, sphere_col
