"""Minimal URDF reader for the kinematics path (no third-party parser).

Only what forward kinematics consumes is read: the `<link>` names in file
order and, per `<joint>`, its type, parent/child link, `<origin xyz rpy>`,
`<axis xyz>` and `<limit lower upper velocity effort>`.  Defaults follow what
the reference sees through `urdf_parser_py` (reference:
torch_kinematics_tree/models/utils.py:199-243): a missing `<origin>` or a
missing `xyz`/`rpy` attribute is zero, a missing `<axis>` is "no axis", and a
`<limit>` without `lower`/`upper` has them at 0.
"""
from __future__ import annotations

import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import List, Optional


@dataclass
class UrdfJoint:
    name: str
    type: str
    parent: str
    child: str
    xyz: List[float] = field(default_factory=lambda: [0.0, 0.0, 0.0])
    rpy: List[float] = field(default_factory=lambda: [0.0, 0.0, 0.0])
    axis: Optional[List[float]] = None
    has_limit: bool = False
    lower: float = 0.0
    upper: float = 0.0
    velocity: Optional[float] = None
    effort: Optional[float] = None


@dataclass
class UrdfModel:
    name: str
    links: List[str]
    joints: List[UrdfJoint]


def _vec3(text: Optional[str]) -> List[float]:
    if text is None:
        return [0.0, 0.0, 0.0]
    vals = [float(tok) for tok in text.split()]
    if len(vals) != 3:
        raise ValueError(f"expected 3 floats, got {text!r}")
    return vals


def _opt(elem, key) -> Optional[float]:
    val = elem.get(key)
    return None if val is None else float(val)


def parse_urdf(path: str) -> UrdfModel:
    root = ET.parse(path).getroot()
    if root.tag != "robot":
        raise ValueError(f"{path}: root element is <{root.tag}>, expected <robot>")
    links: List[str] = []
    joints: List[UrdfJoint] = []
    for elem in root:
        if elem.tag == "link":
            links.append(elem.get("name"))
        elif elem.tag == "joint":
            parent, child = elem.find("parent"), elem.find("child")
            if parent is None or child is None:
                raise ValueError(f"{path}: joint {elem.get('name')!r} lacks <parent>/<child>")
            joint = UrdfJoint(name=elem.get("name"), type=elem.get("type"),
                              parent=parent.get("link"), child=child.get("link"))
            origin = elem.find("origin")
            if origin is not None:
                joint.xyz = _vec3(origin.get("xyz"))
                joint.rpy = _vec3(origin.get("rpy"))
            axis = elem.find("axis")
            if axis is not None:
                joint.axis = _vec3(axis.get("xyz"))
            limit = elem.find("limit")
            if limit is not None:
                joint.has_limit = True
                lower, upper = _opt(limit, "lower"), _opt(limit, "upper")
                joint.lower = 0.0 if lower is None else lower
                joint.upper = 0.0 if upper is None else upper
                joint.velocity = _opt(limit, "velocity")
                joint.effort = _opt(limit, "effort")
            joints.append(joint)
    if not links:
        raise ValueError(f"{path}: no <link> elements")
    return UrdfModel(name=root.get("name", ""), links=links, joints=joints)
