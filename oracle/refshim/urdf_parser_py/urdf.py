"""Oracle tooling only (see package docstring).

Restates the handful of `urdf_parser_py.urdf` behaviours the reference relies on
(reference: torch_kinematics_tree/models/utils.py:182-313, robots.py:6):
file-order `links` / `joints`, `child_map`, zero defaults for a missing
<origin>, `axis=None` when <axis> is absent, `limit.lower/upper` default 0.
"""
import xml.etree.ElementTree as ET


def _floats(text, default=None):
    if text is None:
        return default
    return [float(tok) for tok in text.split()]


class Pose:
    def __init__(self, xyz=None, rpy=None):
        self.xyz = list(xyz) if xyz is not None else [0.0, 0.0, 0.0]
        self.rpy = list(rpy) if rpy is not None else [0.0, 0.0, 0.0]

    @property
    def position(self):
        return self.xyz

    @property
    def rotation(self):
        return self.rpy


class Inertia:
    def __init__(self, **kw):
        for key in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz"):
            setattr(self, key, float(kw.get(key, 0.0)))


class Inertial:
    def __init__(self, mass=0.0, inertia=None, origin=None):
        self.mass, self.inertia, self.origin = mass, inertia, origin


class JointLimit:
    def __init__(self, effort=None, velocity=None, lower=0.0, upper=0.0):
        self.effort, self.velocity, self.lower, self.upper = effort, velocity, lower, upper


class JointDynamics:
    def __init__(self, damping=None, friction=None):
        self.damping, self.friction = damping, friction


class Box:
    def __init__(self, size=None):
        self.size = size


class Visual:
    def __init__(self, geometry=None, material=None, origin=None):
        self.geometry = geometry


class Collision:
    def __init__(self, geometry=None, origin=None):
        self.geometry = geometry


class Link:
    def __init__(self, name=None, visual=None, inertial=None, collision=None, origin=None):
        self.name, self.visual, self.inertial = name, visual, inertial
        self.collision, self.origin = collision, origin


class Joint:
    def __init__(self, name=None, parent=None, child=None, joint_type=None, axis=None,
                 origin=None, limit=None, dynamics=None):
        self.name, self.parent, self.child, self.type = name, parent, child, joint_type
        self.axis, self.origin, self.limit, self.dynamics = axis, origin, limit, dynamics


def _pose_of(elem):
    if elem is None:
        return Pose()
    return Pose(_floats(elem.get("xyz")), _floats(elem.get("rpy")))


def _opt_float(elem, key, default):
    val = elem.get(key)
    return float(val) if val is not None else default


class URDF:
    def __init__(self):
        self.links, self.joints = [], []
        self.child_map, self.parent_map = {}, {}

    @classmethod
    def from_xml_file(cls, path):
        model = cls()
        for elem in ET.parse(path).getroot():
            if elem.tag == "link":
                inertial = None
                ine = elem.find("inertial")
                if ine is not None:
                    mass, inertia = ine.find("mass"), ine.find("inertia")
                    origin = ine.find("origin")
                    inertial = Inertial(
                        float(mass.get("value")) if mass is not None else 0.0,
                        Inertia(**inertia.attrib) if inertia is not None else None,
                        _pose_of(origin) if origin is not None else None)
                model.links.append(Link(name=elem.get("name"), inertial=inertial))
            elif elem.tag == "joint":
                axis, limit, dyn = elem.find("axis"), elem.find("limit"), elem.find("dynamics")
                joint = Joint(
                    name=elem.get("name"),
                    parent=elem.find("parent").get("link"),
                    child=elem.find("child").get("link"),
                    joint_type=elem.get("type"),
                    axis=_floats(axis.get("xyz")) if axis is not None else None,
                    origin=_pose_of(elem.find("origin")),
                    limit=JointLimit(_opt_float(limit, "effort", None), _opt_float(limit, "velocity", None),
                                     _opt_float(limit, "lower", 0.0), _opt_float(limit, "upper", 0.0))
                    if limit is not None else None,
                    dynamics=JointDynamics(_opt_float(dyn, "damping", None)) if dyn is not None else None)
                model.joints.append(joint)
                model.child_map.setdefault(joint.parent, []).append((joint.name, joint.child))
                model.parent_map[joint.child] = (joint.name, joint.parent)
        return model
