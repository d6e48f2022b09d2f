"""Model compiler, second stage: KinModel + collision-link template -> straight-line HIP kernel source.

The table-driven kernels (csrc/trk_kernels.hip) work for any URDF.  For the robots a deployment actually
plans with, this generator unrolls the kinematic tree into one fused FK + objectives + gradient kernel:

* every joint transform is emitted with the URDF constants as literals; entries that are exactly 0 / +-1
  (after snapping |x| < `snap` to 0 and |x -+ 1| < `snap` to +-1; cos(pi/2) in fp32 is -4.4e-8) are folded away
  symbolically, so a Panda joint costs 12 FMAs for the rotation instead of 27 + 12;
* all link poses are named registers -- the compiler's allocator handles branch parents of tree robots;
* the reverse pass is the transpose of the geometric Jacobian (per-link wrench accumulators pushed towards
  the root), emitted only for links that can receive an adjoint.

Scene, margins, weights, EE target and base pose stay run-time arguments (scalar loads).  The generated
translation unit registers itself with libtrk.so; `trk_rollout_cost_grad` picks it when the model hash and
the cost model's link sets match, and falls back to the table-driven kernel otherwise.
Math restated from the reference: rigid_body.py:146-211 (joint transforms), SURVEY.md Appendix B (reverse).
"""
from __future__ import annotations

import os

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .kinmodel import JOINT_CONTINUOUS, JOINT_FIXED, JOINT_PRISMATIC, JOINT_REVOLUTE, KinModel

SNAP = 1e-7


def model_hash(kin: KinModel) -> int:
    """FNV-1a (64 bit) over the kinematic tables; the same bytes are hashed by trk_capi.hip."""
    h = 0xcbf29ce484222325
    parts = [np.asarray([kin.n_links, kin.n_dofs], np.int32)]
    for name, dt in (("parent", np.int32), ("joint_type", np.int32), ("dof_idx", np.int32), ("R_fixed", np.float32),
                     ("trans", np.float32), ("axis", np.float32), ("rot_axis", np.int32), ("rot_sign", np.float32),
                     ("clamp", np.int32), ("lower", np.float32), ("upper", np.float32), ("order", np.int32)):
        parts.append(np.ascontiguousarray(getattr(kin, name), dt).reshape(-1))
    for arr in parts:
        for b in arr.tobytes():
            h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def flit(x: float) -> str:
    """exact fp32 literal"""
    v = float(np.float32(x))
    if v == 0.0:
        return "0.0f"
    return f"{v.hex()}f"


@dataclass(frozen=True)
class S:
    """coef * name ; a constant when name is None"""
    c: float
    n: Optional[str] = None

    @property
    def is_const(self):
        return self.n is None

    @property
    def is_zero(self):
        return self.c == 0.0

    def neg(self):
        return S(-self.c, self.n)


ZERO, ONE = S(0.0), S(1.0)


def snap_const(x: float, eps: float) -> float:
    x = float(np.float32(x))
    if abs(x) < eps:
        return 0.0
    if abs(x - 1.0) < eps:
        return 1.0
    if abs(x + 1.0) < eps:
        return -1.0
    return x


class Emitter:
    def __init__(self):
        self.lines: List[str] = []
        self._k = 0

    def raw(self, line: str):
        self.lines.append(line)

    def tmp(self, expr: str) -> str:
        self._k += 1
        name = f"v{self._k}"
        self.lines.append(f"    const float {name} = {expr};")
        return name

    def expr(self, s: S) -> str:
        if s.n is None:
            return flit(s.c)
        if s.c == 1.0:
            return s.n
        if s.c == -1.0:
            return f"-{s.n}"
        return f"({flit(s.c)} * {s.n})"

    def named(self, s: S) -> S:
        """make sure the value is a plain +-name (materialise coef * name when |coef| != 1)"""
        if s.n is None or abs(s.c) == 1.0:
            return s
        return S(1.0, self.tmp(self.expr(s)))

    def lincomb(self, pairs: Sequence[Tuple[S, S]], add: S = ZERO) -> S:
        """sum_i a_i * b_i + add with constant folding; emits at most one fmaf chain."""
        const = add.c if add.n is None else 0.0
        terms: List[Tuple[float, str, Optional[str]]] = []
        if add.n is not None and add.c != 0.0:
            terms.append((add.c, add.n, None))
        for a, b in pairs:
            if a.is_zero or b.is_zero:
                continue
            if a.n is None and b.n is None:
                const += a.c * b.c
            elif a.n is None:
                terms.append((a.c * b.c, b.n, None))
            elif b.n is None:
                terms.append((a.c * b.c, a.n, None))
            else:
                terms.append((a.c * b.c, a.n, b.n))
        # merge identical single-name terms
        merged: Dict[Tuple[str, Optional[str]], float] = {}
        order: List[Tuple[str, Optional[str]]] = []
        for c, n1, n2 in terms:
            key = (n1, n2)
            if key not in merged:
                merged[key] = 0.0
                order.append(key)
            merged[key] += c
        terms = [(merged[k], k[0], k[1]) for k in order if merged[k] != 0.0]
        if not terms:
            return S(float(np.float32(const)))
        if len(terms) == 1 and const == 0.0:
            c, n1, n2 = terms[0]
            if n2 is None:
                return S(c, n1)                     # pure alias, no instruction
            return S(c, self.tmp(f"{n1} * {n2}"))
        # single-name terms first so the accumulator starts cheap
        terms.sort(key=lambda t: t[2] is not None)
        acc: Optional[str] = flit(const) if const != 0.0 else None
        for c, n1, n2 in terms:
            if n2 is None:
                if acc is None:
                    acc = self.expr(S(c, n1))
                elif c == 1.0:
                    acc = f"({acc} + {n1})"
                elif c == -1.0:
                    acc = f"({acc} - {n1})"
                else:
                    acc = f"fmaf({flit(c)}, {n1}, {acc})"
            else:
                lhs = n1 if c == 1.0 else (f"-{n1}" if c == -1.0 else f"({flit(c)} * {n1})")
                acc = f"{lhs} * {n2}" if acc is None else f"fmaf({lhs}, {n2}, {acc})"
        return S(1.0, self.tmp(acc))

    def add(self, a: S, b: S) -> S:
        return self.lincomb([(a, ONE), (b, ONE)])

    def cross(self, a: Sequence[S], b: Sequence[S]) -> List[S]:
        return [self.lincomb([(a[1], b[2]), (a[2].neg(), b[1])]),
                self.lincomb([(a[2], b[0]), (a[0].neg(), b[2])]),
                self.lincomb([(a[0], b[1]), (a[1].neg(), b[0])])]

    def dot(self, a: Sequence[S], b: Sequence[S]) -> S:
        return self.lincomb([(a[0], b[0]), (a[1], b[1]), (a[2], b[2])])


@dataclass
class CollisionTemplate:
    """Link sets baked into a specialised kernel (the robot's collision model, robot_base.py:57-141)."""
    obj_links: List[int]
    self_pairs: List[Tuple[int, int]] = field(default_factory=list)   # LINK indices (a, b)
    ee_link: int = -1
    ee2_link: int = -1          # second tracked link (two-arm scenes), same weights
    # interpolate_link_pos (distance_fields.py:66-69, 145-147): virtual columns L + v = w0 * link src0 + w1 * link src1, which
    # obj_links / self_pairs may name like links (rows: src0, src1, w0, w1 -- costmodel.interpolation_table)
    virtual: List[Tuple[int, int, float, float]] = field(default_factory=list)


def panda_template(kin: KinModel) -> CollisionTemplate:
    """RobotPanda's collision model (robot_panda.py:44-136)."""
    idx = kin.name_to_idx
    obj = [idx[n] for n in ("panda_link2", "panda_link3", "panda_link5", "panda_link7", "panda_hand")]
    pairs_by_name = {"panda_link4": ["panda_link1"], "panda_link5": ["panda_link0", "panda_link1", "panda_link2"],
                     "panda_link6": ["panda_link0", "panda_link1", "panda_link2"],
                     "panda_hand": ["panda_link0", "panda_link1", "panda_link2"]}
    # order of robot_base.py:110-118: iterate the sorted unique self-link names, then the listed partners
    names = sorted(set(list(pairs_by_name) + [v for vs in pairs_by_name.values() for v in vs] +
                       ["panda_link0", "panda_link1", "panda_link2", "panda_link3"]))
    pairs = []
    for a in names:
        for b in pairs_by_name.get(a, []):
            pairs.append((idx[a], idx[b]))
    return CollisionTemplate(obj_links=obj, self_pairs=pairs, ee_link=idx["ee_link"])


def _masked_factory(kin: KinModel):
    def masked(E: Emitter, i: int, d: int, g: S) -> S:
        """torch.clamp's gradient mask: pass g where the clamp left q unchanged, else 0 (one v_cndmask)."""
        if g.is_zero or not kin.clamp[i]:
            return g
        return S(1.0, E.tmp(f"(passbits & {1 << d}u) ? {E.expr(g)} : 0.0f"))
    return masked


def _emit_angles(E: "Emitter", kin: KinModel, links=None, declare_passbits: bool = True, qexpr=None) -> None:
    """joint angles: clamp to the URDF limits (torch.clamp, rigid_body.py:157-160; the gradient mask is
    "q inside the limits" == "clamp left q unchanged"), then all sines / cosines, two angles per call.
    links: only the joints of these links (a segment of the tree); qexpr(d): the C expression of q[d] (default: the array q)."""
    L = kin.n_links
    rot_dofs = []
    qexpr = qexpr or (lambda d: f"q[{d}]")
    if declare_passbits:
        E.raw("    unsigned passbits = 0u;      // bit d set: clamp left q[d] unchanged -> gradient passes (one VGPR instead of 2D)")
    for i in (range(1, L) if links is None else links):
        jt, d = int(kin.joint_type[i]), int(kin.dof_idx[i])
        if jt == JOINT_FIXED:
            continue
        if kin.clamp[i]:
            E.raw(f"    const float qh{d} = __builtin_amdgcn_fmed3f({qexpr(d)}, {flit(kin.lower[i])}, {flit(kin.upper[i])});")
            E.raw(f"    passbits |= (qh{d} == {qexpr(d)}) ? {1 << d}u : 0u;")
        else:
            E.raw(f"    const float qh{d} = {qexpr(d)};")
        if jt in (JOINT_REVOLUTE, JOINT_CONTINUOUS) and float(kin.rot_sign[i]) != 0.0:
            rot_dofs.append(d)
    # pin the mask in ONE register: without this the compiler re-derives it from q and qh in the reverse pass and
    # keeps 2D registers alive across the whole kernel (-> spills, and every spill reload is an s_waitcnt vmcnt(0)
    # that also waits for the wave's outstanding output stores)
    E.raw('    asm volatile("" : "+v"(passbits));')
    for d in rot_dofs:
        E.raw(f"    float sn{d}, cs{d};")
    for a, b in zip(rot_dofs[0::2], rot_dofs[1::2]):
        E.raw(f"    trk_sincos2(qh{a}, qh{b}, &sn{a}, &cs{a}, &sn{b}, &cs{b});")
    if len(rot_dofs) % 2:
        d = rot_dofs[-1]
        E.raw(f"    trk_sincos(qh{d}, &sn{d}, &cs{d});")


def _emit_fk_link(E: "Emitter", kin: KinModel, i: int, R, t, passv, snap: float, stateful: bool = False, fixed_override=None) -> None:
    """world pose of link i from its parent's (rigid_body.py:162-182), URDF constants folded symbolically.
    stateful: the stateful path's joint transform (rigid_body.py:213-251): rotation about `sf_rot_axis` with the axis SIGN
    IGNORED, whatever the stateless rule says (a missing axis still rotates about z).
    fixed_override: {link: (Rf 3x3 of S, tl 3 of S)} -- the link's fixed transform given symbolically (arm-per-lane kernels: the entries
    that differ between the two arms are per-lane registers)."""
    par = int(kin.parent[i]); jt = int(kin.joint_type[i]); d = int(kin.dof_idx[i])
    E.raw(f"    // link {i} '{kin.link_names[i]}' (parent {par})")
    Rf = [[S(snap_const(kin.R_fixed[i][r][c], snap)) for c in range(3)] for r in range(3)]
    tl = [S(snap_const(kin.trans[i][k], 0.0)) for k in range(3)]
    if fixed_override is not None and i in fixed_override:
        Rf, tl = fixed_override[i]
    qh = None
    if jt != JOINT_FIXED:
        passv[i] = S(1.0, f"pass{d}") if kin.clamp[i] else ONE
        qh = S(1.0, f"qh{d}")
    if jt == JOINT_PRISMATIC:
        tl = [E.lincomb([(S(float(kin.axis[i][k])), qh)], tl[k]) for k in range(3)]
    Rp, tp = R[par], t[par]
    t[i] = [E.lincomb([(Rp[r][k], tl[k]) for k in range(3)], tp[r]) for r in range(3)]
    A = [[E.lincomb([(Rp[r][k], Rf[k][c]) for k in range(3)]) for c in range(3)] for r in range(3)]
    if jt in (JOINT_REVOLUTE, JOINT_CONTINUOUS):
        sg = 1.0 if stateful else float(kin.rot_sign[i])
        if sg != 0.0:
            s, c = S(sg, f"sn{d}"), S(1.0, f"cs{d}")       # sin(sign*q) = sign*sin(q), cos even
            ax = int(kin.sf_rot_axis[i]) if stateful else int(kin.rot_axis[i])
            ci, cj = [(1, 2), (2, 0), (0, 1)][ax]
            newA = [row[:] for row in A]
            for r in range(3):
                a_i, a_j = E.named(A[r][ci]), E.named(A[r][cj])
                newA[r][ci] = E.lincomb([(c, a_i), (s, a_j)])
                newA[r][cj] = E.lincomb([(c, a_j), (s.neg(), a_i)])
            A = newA
    R[i] = A


def _emit_joint_gradient(E: "Emitter", kin: KinModel, i: int, R, t, Fi, Ti, masked) -> S:
    """d cost / d q of link i's joint from the wrench (F, T about the world origin) of its subtree:
    revolute  s z_i . (T - t_i x F),  prismatic  (R_parent axis) . F   """
    par = int(kin.parent[i]); jt = int(kin.joint_type[i]); d = int(kin.dof_idx[i])
    if not any(not s.is_zero for s in Fi + Ti):
        return ZERO
    if jt == JOINT_PRISMATIC:
        dirw = [E.lincomb([(R[par][r][k], S(float(kin.axis[i][k]))) for k in range(3)]) for r in range(3)]
        return masked(E, i, d, E.dot(dirw, Fi))
    sg = float(kin.rot_sign[i])
    if sg == 0.0:
        return ZERO
    ax = int(kin.rot_axis[i])
    z = [R[i][r][ax] for r in range(3)]
    cr = E.cross(t[i], Fi)
    diff = [E.lincomb([(Ti[k], ONE), (cr[k], S(-1.0))]) for k in range(3)]
    g = E.dot(z, diff)
    return masked(E, i, d, S(g.c * sg, g.n))


def _emit_reverse_links(E: "Emitter", kin: KinModel, R, t, tb_names: Dict[int, List[str]], rot_adj: Dict[int, str], masked,
                        tick=None, order=None, pre_link=None, n_links=None) -> Dict[int, S]:
    """Reverse pass of the link kernels: per-link wrench accumulators (F, T about the world origin) pushed towards the
    root; `tb_names[i]` = the three C expressions holding link i's position adjoint; a tracked link i adds
    axial(Rb R^T) with Rb = the 9-float array named rot_adj[i]."""
    L = kin.n_links
    F: Dict[int, List[S]] = {i: [ZERO, ZERO, ZERO] for i in range(L)}
    T: Dict[int, List[S]] = {i: [ZERO, ZERO, ZERO] for i in range(L)}
    gq_expr: Dict[int, S] = {}
    order = kin.order if order is None else order       # any order with parents before children (walked backwards here)
    # n_links: only the first n_links entries of `order` are walked (a chain that carries every adjoint)
    for p in range((L if n_links is None else n_links) - 1, 0, -1):
        i = int(order[p]); par = int(kin.parent[i]); jt = int(kin.joint_type[i]); d = int(kin.dof_idx[i])
        E.raw(f"    // reverse: link {i}")
        if pre_link is not None:
            pre_link(i)                         # e.g. fetch this link's adjoint just before it is consumed
        if tick is not None and p % 2 == 0:
            E.raw(tick())                       # a callable: every tick carries its own compile-time chunk number
        if i in tb_names:
            tb = [S(1.0, n) for n in tb_names[i]]
            own_T = E.cross(t[i], tb)
            F[i] = [E.add(F[i][k], tb[k]) for k in range(3)]
            T[i] = [E.add(T[i][k], own_T[k]) for k in range(3)]
            if i in rot_adj:
                Rb = [[S(1.0, f"{rot_adj[i]}[{3 * r + c}]") for c in range(3)] for r in range(3)]
                Ri = R[i]
                M = lambda a, b: E.lincomb([(Rb[a][k], Ri[b][k]) for k in range(3)])   # (Rbar R^T)[a][b]
                tor = [E.lincomb([(M(2, 1), ONE), (M(1, 2), S(-1.0))]),
                       E.lincomb([(M(0, 2), ONE), (M(2, 0), S(-1.0))]),
                       E.lincomb([(M(1, 0), ONE), (M(0, 1), S(-1.0))])]
                T[i] = [E.add(T[i][k], tor[k]) for k in range(3)]
        nonzero = any(not s.is_zero for s in F[i] + T[i])
        if jt != JOINT_FIXED:
            if not nonzero:
                gq_expr[d] = ZERO
            elif jt == JOINT_PRISMATIC:
                dirw = [E.lincomb([(R[par][r][k], S(float(kin.axis[i][k]))) for k in range(3)]) for r in range(3)]
                g = E.dot(dirw, F[i])
                gq_expr[d] = masked(E, i, d, g)
            else:
                sg = float(kin.rot_sign[i])
                if sg == 0.0:
                    gq_expr[d] = ZERO
                else:
                    ax = int(kin.rot_axis[i])
                    z = [R[i][r][ax] for r in range(3)]
                    cr = E.cross(t[i], F[i])
                    diff = [E.lincomb([(T[i][k], ONE), (cr[k], S(-1.0))]) for k in range(3)]
                    g = E.dot(z, diff)
                    gq_expr[d] = masked(E, i, d, S(g.c * sg, g.n))
        F[par] = [E.add(F[par][k], F[i][k]) for k in range(3)]
        T[par] = [E.add(T[par][k], T[i][k]) for k in range(3)]
    return gq_expr


# ----------------------------------------------------------------------------------------------------------------------
# Arm-per-lane schedule of the GP-fused rollout (round 6; BASELINE config 5, the dual Panda): a robot that is two ISOMORPHIC arms on one
# base is evaluated with lanes 2s / 2s + 1 of a wavefront holding the two arms of sample s -- each lane walks ONE arm (a Panda-sized
# problem: ~120 registers, four wavefronts per SIMD like the headline kernel) instead of one lane walking both (217 registers, two).
# The memory side needs no new layout: q / qd / gq / gqd (N, D) are read and written through their (2N, D/2) view, a half row per lane;
# the prior's neighbours in time are the half rows two lanes away in the same raw tile; the positions of a sample -- world link, arm 0,
# arm 1 -- are staged by its two lanes into one row of the wavefront's output image.  What differs between the arms (the mount of the
# arm's root link, the collision margins, the EE target) is a per-lane select of two scalars.
# ----------------------------------------------------------------------------------------------------------------------
@dataclass
class ArmLanePlan:
    n: int                      # links per arm (arm 0 = links 1 .. n, arm 1 = links n + 1 .. 2n; link 0 is the common base)
    DA: int                     # joints per arm (arm 0 = DOFs 0 .. DA - 1)
    arm: object                 # the one-arm model: link 0 + arm 0 (what the generator's FK / reverse emitters walk)
    obj: List[int]              # arm 0's collision links (the template's first half; arm 1's are these + n)
    ee: int                     # the tracked link of arm 0 (arm 1 tracks ee + n), -1 = none
    differ: List[Tuple[str, int, int, float, float]]    # entries of the arm root's fixed transform that differ: (R|t, r, c, arm 0, arm 1)


def arm_lane_plan(kin: KinModel, tmpl: CollisionTemplate) -> Optional[ArmLanePlan]:
    """The plan when `kin` is two isomorphic subtrees hanging off link 0 and the template treats them alike, else None."""
    from types import SimpleNamespace
    L, D = kin.n_links, kin.n_dofs
    if L < 3 or (L - 1) % 2 or D % 2 or D == 0 or tmpl.virtual:
        return None
    n, DA = (L - 1) // 2, D // 2
    par = [int(v) for v in kin.parent]
    if [int(v) for v in kin.order] != list(range(L)) or par[1] != 0 or par[1 + n] != 0:
        return None
    f32 = lambda a: np.asarray(a, np.float32)
    for i in range(1, n + 1):
        j = i + n
        if i > 1 and not (1 <= par[i] < i and par[j] == par[i] + n):
            return None
        di, dj = int(kin.dof_idx[i]), int(kin.dof_idx[j])
        if (di < 0) != (dj < 0) or (di >= 0 and not (di < DA and dj == di + DA)):
            return None
        for name in ("joint_type", "rot_axis", "sf_rot_axis", "clamp"):
            if int(getattr(kin, name)[i]) != int(getattr(kin, name)[j]):
                return None
        for name in ("rot_sign", "lower", "upper", "axis"):
            if not np.array_equal(f32(getattr(kin, name)[i]), f32(getattr(kin, name)[j])):
                return None
        if i > 1 and not (np.array_equal(f32(kin.R_fixed[i]), f32(kin.R_fixed[j])) and np.array_equal(f32(kin.trans[i]), f32(kin.trans[j]))):
            return None
    differ = []
    for r in range(3):
        for c in range(3):
            a, b = snap_const(kin.R_fixed[1][r][c], SNAP), snap_const(kin.R_fixed[1 + n][r][c], SNAP)
            if a != b:
                differ.append(("R", r, c, a, b))
        a, b = snap_const(kin.trans[1][r], 0.0), snap_const(kin.trans[1 + n][r], 0.0)
        if a != b:
            differ.append(("t", r, 0, a, b))
    obj = list(tmpl.obj_links)
    h = len(obj) // 2
    if len(obj) % 2 or any(not (1 <= i <= n) for i in obj[:h]) or obj[h:] != [i + n for i in obj[:h]]:
        return None
    if (tmpl.ee_link >= 0) != (tmpl.ee2_link >= 0) or (tmpl.ee_link >= 0 and not (1 <= tmpl.ee_link <= n and tmpl.ee2_link == tmpl.ee_link + n)):
        return None
    arm = SimpleNamespace(n_links=n + 1, n_dofs=DA, name=kin.name, link_names=list(kin.link_names[:n + 1]), order=np.arange(n + 1, dtype=np.int32),
                          name_to_idx={k: v for k, v in kin.name_to_idx.items() if v <= n})
    for name in ("parent", "joint_type", "dof_idx", "R_fixed", "trans", "axis", "rot_axis", "sf_rot_axis", "rot_sign", "clamp", "lower", "upper"):
        setattr(arm, name, np.asarray(getattr(kin, name))[:n + 1])
    return ArmLanePlan(n=n, DA=DA, arm=arm, obj=obj[:h], ee=int(tmpl.ee_link), differ=differ)


def _arm_lane_kernel_lines(kin: KinModel, tmpl: CollisionTemplate, plan: ArmLanePlan, snap: float) -> List[str]:
    """k_rollout_gpa_bi / _bg: the GP-fused rollout (trk_rollout_gp_cost_grad) with one ARM per lane.  Serves the sphere scenes
    (scene_is_fast) without self-collision pairs (w_self == 0: the template's pairs cross the arms; the launcher hands those calls to
    k_rollout_gpt); BUILD-DEFINED like the prior itself, same oracle (orc_rollout + orc_gp_prior)."""
    L, D = kin.n_links, kin.n_dofs
    n, DA, arm = plan.n, plan.DA, plan.arm
    LA, NLA, W = n + 1, len(plan.obj), 3 * L
    masked = _masked_factory(arm)
    n_rev_ticks = sum(1 for p in range(LA - 1, 0, -1) if p % 2 == 0)
    n_slots = (OBJ_TICK_SLOTS if NLA else 0) + 1 + n_rev_ticks
    out: List[str] = []
    for base_identity in (True, False):
        E = Emitter()
        kname = "k_rollout_gpa_bi" if base_identity else "k_rollout_gpa_bg"
        E.raw("template <class IO>      // HBM-side type of q / qd / link_pos / gradients: float, _Float16 or HalfG32")
        E.raw(f"__global__ void __launch_bounds__(SPEC_BLOCK, 4) {kname}(SpecArgs A) {{")
        E.raw(f"    constexpr int DA = {DA}, NLA = {NLA}, ROWS = TRK_WAVE / 2, W = {W};      // joints / collision links per arm, samples per wavefront, position floats per sample")
        E.raw("    typedef typename IoTraits<IO>::Q IOQ;")
        E.raw("    typedef typename IoTraits<IO>::G IOG;")
        E.raw("    typedef RawRowsInFlight<D, IOQ, ROWS> Raw;       // the block's 32 sample rows (+ one in front, one behind) = 64 half rows of DA")
        E.raw("    typedef ImgFlusher<W, IOQ, ROWS> Img;")
        E.raw("    constexpr int GQ_B = TRK_WAVE * DA * 4;")
        E.raw("    constexpr int WAVE_B0 = Img::BYTES > 2 * Raw::BYTES ? Img::BYTES : 2 * Raw::BYTES;")
        E.raw("    constexpr int WAVE_B = ((WAVE_B0 > GQ_B ? WAVE_B0 : GQ_B) + 15) / 16 * 16;      // one region per wavefront: raw tiles -> gqd tile -> image -> gq tile")
        E.raw("    __shared__ __attribute__((aligned(16))) unsigned char lds_all[SPEC_WAVES * (WAVE_B + TRK_LDS_SPHERES * 16) + SPEC_WAVES * 4];")
        E.raw("    const int lane = threadIdx.x & (TRK_WAVE - 1);")
        E.raw("    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / TRK_WAVE);")
        E.raw("    unsigned char* wl = lds_all + wave * WAVE_B;")
        E.raw("    float* lds = reinterpret_cast<float*>(wl);")
        E.raw("    float4* lds_sph = reinterpret_cast<float4*>(lds_all + SPEC_WAVES * WAVE_B) + wave * TRK_LDS_SPHERES;")
        E.raw("    float* wsum = reinterpret_cast<float*>(lds_all + SPEC_WAVES * (WAVE_B + TRK_LDS_SPHERES * 16));      // the wavefronts' cost sums: two of them make one 64-sample block")
        E.raw("    const SpheresInFlight sph = spec_load_spheres_issue(A.C, lane);   // waited for together with the rows below")
        E.raw("    const int64_t wblock = (int64_t)blockIdx.x * SPEC_WAVES + wave;     // index of this wave's 32-sample block")
        E.raw("    const int64_t base_s = wblock * ROWS, base_h = wblock * TRK_WAVE;   // first sample; first half row of the (2N, DA) view")
        E.raw("    const int rows_s = (int)max((int64_t)0, min((int64_t)ROWS, A.n - base_s)), rows = 2 * rows_s;")
        E.raw("    const bool odd = (lane & 1) != 0;                                   // the arm this lane evaluates")
        E.raw("    const int srow = lane >> 1;                                         // its sample within the block")
        E.raw("    // time steps: the block starts at step t0 of its trajectory (wave-uniform), a lane's sample sits at (t0 + srow) mod H")
        E.raw("    const unsigned Hh = (unsigned)A.gp_H;")
        E.raw("    const unsigned t0 = (unsigned)(base_s % (int64_t)A.gp_H);")
        E.raw("    const unsigned tl = (t0 + (unsigned)srow) % Hh, t_last = (t0 + (unsigned)(ROWS - 1)) % Hh;")
        E.raw("    const bool edge_prev = rows_s > 0 && t0 > 0u, edge_next = rows_s == ROWS && t_last + 1u < Hh && base_s + ROWS < A.n;")
        E.raw("    const Raw rq = spec_raw_rows_issue<D, IOQ, ROWS>(static_cast<const IOQ*>(A.q), base_s, rows_s, lane, edge_prev, edge_next);")
        E.raw("    const Raw rv = spec_raw_rows_issue<D, IOQ, ROWS>(static_cast<const IOQ*>(A.qd), base_s, rows_s, lane, edge_prev, edge_next);")
        E.raw("    const IOQ* qb = spec_raw_rows_finish<D, IOQ, ROWS>(rq, static_cast<const IOQ*>(A.q), base_s, rows_s, lane, reinterpret_cast<IOQ*>(wl));")
        E.raw("    const IOQ* vb = spec_raw_rows_finish<D, IOQ, ROWS>(rv, static_cast<const IOQ*>(A.qd), base_s, rows_s, lane, reinterpret_cast<IOQ*>(wl + Raw::BYTES));")
        E.raw("    spec_wave_sync();")
        E.raw("    float q[DA], gpv[DA], cost_gp;")
        E.raw("    {")
        E.raw("        // ---- the prior on this arm's DA joints.  e_t = (p_t + dt v_t - p_t+1, v_t - v_t+1), r = Q^-1 e; this sample takes part in the")
        E.raw("        // factors t-1 -> t and t -> t+1; its neighbours in time are the half rows two lanes away (D = 2 DA elements) in the raw tiles.")
        E.raw("        // d/dq_t = r_t.p - r_t-1.p,  d/dqd_t = dt r_t.p + r_t.v - r_t-1.v;  the factor t -> t+1 is attributed to sample t.")
        E.raw("        const bool on = lane < rows;")
        E.raw("        const float mn = (on && tl + 1u < Hh) ? A.gp_w : 0.0f, mp = (on && tl > 0u) ? A.gp_w : 0.0f;")
        E.raw("        const float dt = A.gp_dt, ga = A.gp_a, gb = A.gp_b, gc = A.gp_c;")
        E.raw("        const IOQ* qr = qb + lane * DA;")
        E.raw("        const IOQ* vr = vb + lane * DA;")
        E.raw("        float gvv[DA], accg = 0.0f;")
        E.raw("        const float gan = mn * ga, gbn = mn * gb, gcn = mn * gc, gap = mp * ga, gbp = mp * gb, gcp = mp * gc;")
        E.raw("#pragma unroll")
        E.raw("        for (int d = 0; d < DA; ++d) {")
        E.raw("            const float p0 = (float)qr[d], v0 = (float)vr[d];")
        E.raw("            const float pm = (float)qr[d - D], vm = (float)vr[d - D], pn = (float)qr[d + D], vn = (float)vr[d + D];")
        E.raw("            const float ep = fmaf(dt, v0, p0) - pn, ev = v0 - vn;")
        E.raw("            const float rp = fmaf(gan, ep, gbn * ev), rv_ = fmaf(gbn, ep, gcn * ev);      // the weight (and the time-step mask) folded into Q^-1 once per lane")
        E.raw("            accg = fmaf(ep, rp, fmaf(ev, rv_, accg));                                      // 2 x the factor's cost: halved once, below")
        E.raw("            const float em = fmaf(dt, vm, pm) - p0, fm = vm - v0;")
        E.raw("            gpv[d] = rp - fmaf(gap, em, gbp * fm);")
        E.raw("            gvv[d] = fmaf(dt, rp, rv_) - fmaf(gbp, em, gcp * fm);")
        E.raw("            q[d] = p0;")
        E.raw("        }")
        E.raw("        cost_gp = 0.5f * accg;")
        E.raw("        // d cost / d qd is final: out through the staging tile (its first line waits for every lane's reads of the raw tiles)")
        E.raw("        spec_store_gq<DA, IOG, IoTraits<IO>::kScaled>(static_cast<IOG*>(A.gqd), base_h, rows, lane, lds, gvv, A.grad_scale);")
        E.raw("    }")
        E.raw("    spec_load_spheres_finish(lds_sph, lane, sph);")
        # ---------------- what differs between the arms: per-lane selects ----------------
        R: Dict[int, List[List[S]]] = {}
        t: Dict[int, List[S]] = {}
        passv: Dict[int, S] = {}
        if base_identity:
            R[0] = [[ONE if r == c else ZERO for c in range(3)] for r in range(3)]
            t[0] = [ZERO, ZERO, ZERO]
        else:
            R[0] = [[S(1.0, f"A.base_R[{3 * r + c}]") for c in range(3)] for r in range(3)]
            t[0] = [S(1.0, f"A.base_t[{r}]") for r in range(3)]
        Rf = [[S(snap_const(arm.R_fixed[1][r][c], snap)) for c in range(3)] for r in range(3)]
        tlv = [S(snap_const(arm.trans[1][k], 0.0)) for k in range(3)]
        for kind, r, c, a, b in plan.differ:
            nm = f"mnt{kind}{r}{c}"
            E.raw(f"    const float {nm} = odd ? {flit(b)} : {flit(a)};      // the arm root's fixed transform, entry {kind}[{r}][{c}]: arm 1 / arm 0")
            if kind == "R":
                Rf[r][c] = S(1.0, nm)
            else:
                tlv[r] = S(1.0, nm)
        _emit_angles(E, arm)
        for i in range(1, LA):
            _emit_fk_link(E, arm, i, R, t, passv, snap, fixed_override={1: (Rf, tlv)})
        # ---------------- positions: the two lanes of a sample stage its row of the wavefront's output image ----------------
        E.raw("    const Img pimg = spec_make_img<W, IOQ, ROWS>(static_cast<IOQ*>(A.link_pos), base_s, rows_s, lane, reinterpret_cast<IOQ*>(wl));")
        E.raw("    spec_wave_sync();          // the gqd staging tile has been read out: the image may be written")
        E.raw("    if (A.link_pos) {")
        E.raw(f"        IOQ* prow = reinterpret_cast<IOQ*>(wl) + srow * W + (odd ? {3 + 3 * n} : 3);      // row = [link 0 | arm 0's links | arm 1's links]")
        E.raw("        if (!odd) { " + " ".join(f"prow[{k - 3}] = (IOQ)({E.expr(t[0][k])});" for k in range(3)) + " }")
        for i in range(1, LA):
            E.raw("        " + " ".join(f"prow[{3 * (i - 1) + k}] = (IOQ)({E.expr(t[i][k])});" for k in range(3)))
        E.raw("    }")
        E.raw("    spec_wave_sync();          // the image is complete")
        E.raw("    spec_img_copy_slow(pimg, static_cast<IOQ*>(A.link_pos), base_s, rows_s);       // ragged last wavefront / unaligned view only")
        E.raw(f"    constexpr int PPT = (Img::NP + {n_slots - 1}) / {n_slots};")
        E.raw("    const ImgTicks<Img, PPT> flush{pimg};")
        E.raw("    if (!A.gq) { flush.template rest<0>(); return; }      // positions only: wave-uniform exit (the whole grid takes it: no barrier is missed)")
        next_chunk = [0]

        def tick_line(indent="    "):
            c = next_chunk[0]
            next_chunk[0] += 1
            return f"{indent}flush.template chunk<{c}>();"
        # ---------------- objectives of this lane's arm ----------------
        adj = sorted(set(plan.obj) | ({plan.ee} if plan.ee >= 0 else set()))
        E.raw("    float cost = 0.0f;")
        for i in adj:
            E.raw(f"    float tb{i}_0 = 0.0f, tb{i}_1 = 0.0f, tb{i}_2 = 0.0f;")
        if NLA:
            for k, nm in enumerate("xyz"):
                E.raw(f"    const float p{nm}[NLA] = {{{', '.join(E.expr(t[i][k]) for i in plan.obj)}}};")
            E.raw("    float gx[NLA], gy[NLA], gz[NLA];")
            E.raw("#pragma unroll")
            E.raw("    for (int l = 0; l < NLA; ++l) { gx[l] = 0.0f; gy[l] = 0.0f; gz[l] = 0.0f; }")
            c0 = next_chunk[0]
            next_chunk[0] += OBJ_TICK_SLOTS
            E.raw(f"    const TickFrom<decltype(flush), {c0}> ticks{{flush}};")
            E.raw("    if (A.w.w_obj != 0.0f) cost += spec_objects_cost_arm<NLA, decltype(ticks), true, false>(A.C, A.w.w_obj, px, py, pz, gx, gy, gz, ticks, lds_sph, odd);")
            E.raw(f"    else flush.template range<{c0}, {next_chunk[0]}>();")
            E.raw("    if (A.w.w_ws != 0.0f && A.C.has_ws) cost += spec_ws_cost_arm<NLA>(A.C, A.w.w_ws, px, py, pz, gx, gy, gz, odd);")
            for j, i in enumerate(plan.obj):
                E.raw(f"    tb{i}_0 += gx[{j}]; tb{i}_1 += gy[{j}]; tb{i}_2 += gz[{j}];")
        if plan.ee >= 0:
            ee = plan.ee
            E.raw("    float eeRb[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};")
            E.raw("    if (A.w.w_ee != 0.0f) {")
            E.raw("        float Ht[16];      // this arm's target: rows 0 .. 2 of ee_target (arm 0) / ee2_target (arm 1)")
            E.raw("#pragma unroll")
            E.raw("        for (int k = 0; k < 12; ++k) Ht[k] = odd ? A.C.ee2_target[k] : A.C.ee_target[k];")
            E.raw("        Ht[12] = 0.0f; Ht[13] = 0.0f; Ht[14] = 0.0f; Ht[15] = 1.0f;")
            E.raw(f"        const float eR[9] = {{{', '.join(E.expr(R[ee][r][c]) for r in range(3) for c in range(3))}}};")
            E.raw(f"        const float et[3] = {{{', '.join(E.expr(t[ee][k]) for k in range(3))}}};")
            E.raw("        float gR[9], gt[3];")
            E.raw("        const float ce = ee_cost_eval(eR, et, Ht, A.C.ee_w_pos, A.C.ee_w_rot, A.C.ee_square, gR, gt);")
            E.raw("        cost = fmaf(A.w.w_ee, ce, cost);")
            E.raw("#pragma unroll")
            E.raw("        for (int k = 0; k < 9; ++k) eeRb[k] = A.w.w_ee * gR[k];")
            E.raw(f"        tb{ee}_0 = fmaf(A.w.w_ee, gt[0], tb{ee}_0); tb{ee}_1 = fmaf(A.w.w_ee, gt[1], tb{ee}_1); tb{ee}_2 = fmaf(A.w.w_ee, gt[2], tb{ee}_2);")
            E.raw("    }")
        E.raw(tick_line())
        # ---------------- cost: the sample's cost is the sum over its two lanes; a 64-sample block sum is two wavefronts' ----------------
        E.raw("    cost += cost_gp;           // the prior's factor t -> t + 1 (this arm's joints), attributed to this sample")
        E.raw("    const float cs = cost + trk_dpp_partner(cost);      // both lanes of a sample hold its cost (a + b == b + a: the same bits)")
        E.raw("    if (!odd && lane < rows) store_wt_f1(A.cost + base_s + srow, cs);")
        E.raw("    if (A.cost_sum) {")
        E.raw("        const float tot = spec_wave_sum((!odd && lane < rows) ? cs : 0.0f);")
        E.raw("        if (lane == 0) wsum[wave] = tot;")
        E.raw("    }")
        # ---------------- reverse ----------------
        gq_expr = _emit_reverse_links(E, arm, R, t, {i: [f"tb{i}_{k}" for k in range(3)] for i in adj},
                                      ({plan.ee: "eeRb"} if plan.ee >= 0 else {}), masked, tick=tick_line, order=list(range(LA)))
        assert next_chunk[0] <= n_slots, (next_chunk[0], n_slots)
        E.raw(f"    flush.template rest<{next_chunk[0]}>();")
        E.raw(f"    const float gv[DA] = {{{', '.join(E.expr(gq_expr.get(d, ZERO)) + f' + gpv[{d}]' for d in range(DA))}}};")
        E.raw("    spec_store_gq<DA, IOG, IoTraits<IO>::kScaled>(static_cast<IOG*>(A.gq), base_h, rows, lane, lds, gv, A.grad_scale);")
        E.raw("    if (A.cost_sum) {          // wave-uniform for the whole workgroup")
        E.raw("        __syncthreads();")
        E.raw("        if ((wave & 1) == 0 && lane == 0 && rows > 0) store_wt_f1(A.cost_sum + (wblock >> 1), wsum[wave] + wsum[wave + 1]);")
        E.raw("    }")
        E.raw("}")
        out.extend(E.lines)
        out.append("")
    return out


TRK_WAVE_ = 64
JAC_DIRECT_MAX_DOFS = 16    # k_jac: robots up to this many DOF write their columns straight into the output tiles (Panda 17.0 -> 13.1 us, dual Panda 34.6 -> 27.3)
# tick slots handed to one scene evaluation (csrc/trk_device.h: TRK_OBJ_TICK_SLOTS must agree)
OBJ_TICK_SLOTS = int(os.environ.get("TRK_EXP_OBJ_SLOTS", "5"))


def generate_rollout_source(kin: KinModel, tmpl: CollisionTemplate, ident: str, snap: float = SNAP, meta: Optional[dict] = None) -> str:
    """meta (optional): receives what a loader of the unit's DEVICE code alone needs (jit.py's hipRTC fall-back: the host half of the
    unit -- launchers, registry entry -- is then played by libtrk.so's generic launchers): the kernels' name expressions and the
    unit's traits."""
    L, D = kin.n_links, kin.n_dofs
    NL = len(tmpl.obj_links)
    masked = _masked_factory(kin)
    tracked = [(l, tgt, rb) for l, tgt, rb in ((tmpl.ee_link, "A.C.ee_target", "eeRb"), (tmpl.ee2_link, "A.C.ee2_target", "ee2Rb"))
               if l >= 0]
    virt = {L + v: row for v, row in enumerate(tmpl.virtual)}          # virtual column -> (src0, src1, w0, w1)
    used = set(tmpl.obj_links) | {a for p in tmpl.self_pairs for a in p}
    assert all(i < L + len(virt) for i in used), "collision column out of range"
    used_virt = sorted(i for i in used if i >= L)
    # every column that can receive a position adjoint: the real links among them go to the reverse pass, the virtual ones
    # hand theirs to their two source links first
    adj_links = sorted(used | {l for l, _, _ in tracked} | {src for i in used_virt for src in virt[i][:2]})
    real_adj = [i for i in adj_links if i < L]

    def with_virtual(E, t):
        """t extended by the interpolated points this template uses (named temporaries: w0 * p_src0 + w1 * p_src1)"""
        if not used_virt:
            return t
        t = dict(t)
        for i in used_virt:
            s0, s1, w0, w1 = virt[i]
            t[i] = [E.named(E.lincomb([(t[s0][k], S(float(np.float32(w0)))), (t[s1][k], S(float(np.float32(w1))))])) for k in range(3)]
        return t

    def scatter_virtual(E, indent="    "):
        """adjoint of an interpolated point -> its two source links, with its weights"""
        for i in used_virt:
            s0, s1, w0, w1 = virt[i]
            for src, w in ((s0, w0), (s1, w1)):
                if float(np.float32(w)) != 0.0:
                    E.raw(f"{indent}tb{src}_0 = fmaf({flit(float(np.float32(w)))}, tb{i}_0, tb{src}_0); tb{src}_1 = fmaf({flit(float(np.float32(w)))}, tb{i}_1, tb{src}_1); "
                          f"tb{src}_2 = fmaf({flit(float(np.float32(w)))}, tb{i}_2, tb{src}_2);")
    out: List[str] = []
    out.append(f"// GENERATED by torch_robotics_amd/codegen.py for model '{kin.name}' ({L} links, {D} DOF) -- do not edit.")
    # Value-changing-but-bounded FP freedoms for this unit (NOT finite-math-only): reassociation + contraction turn
    # mul/add chains into FMAs, 1/x may use v_rcp.  Same-box A/B on the headline kernel: 11.55 -> 11.06 us; deviation from
    # the fp64 oracle on 65 536 samples unchanged for positions (2.5e-7) and 6.4e-7 -> 1.0e-6 of max for the gradient
    # (tools/accuracy_check.py).  The attached-point generator does not use it (its kernels got slower: register pressure).
    out.append("#pragma clang fp reassociate(on) contract(fast) reciprocal(on)")
    out.append('#include "trk_spec_common.h"')
    out.append(f"namespace spec_{ident} {{")
    out.append(f"constexpr int L = {L}, D = {D}, NL = {NL};")
    out.append(f'static_assert(TRK_OBJ_TICK_SLOTS == {OBJ_TICK_SLOTS}, "chunk numbering of this unit assumes another TRK_OBJ_TICK_SLOTS");')
    # the walk: file order when that puts every parent before its children (then a link's floats are staged where they sit in
    # the output row, which is what the ring staging needs), else the model's DFS pre-order
    walk = list(range(L)) if all(int(kin.parent[i]) < i for i in range(1, L)) else [int(v) for v in kin.order]
    chunked = 3 * L > CHUNKED_STAGING_MIN_FLOATS and walk == list(range(L))
    if chunked:
        # ring staging: the pieces of the whole chunks leave between the links that follow them; the tail chunk's pieces are what
        # the objectives' tick slots issue (flush.chunk<CH>() -> RingTail)
        rp = ring_plan(3 * L)
        ring_t = f"RingFlusher<{rp.W}, {rp.V}, {'true' if rp.aligned else 'false'}, IOQ>"
        out.append("template <class R> struct RingTail {      // tick slot CH of the objectives -> store piece CH of the tail chunk")
        out.append("    const R& r;")
        out.append("    template <int CH> __device__ __forceinline__ void chunk() const { r.template piece<R::NFULL, CH>(); }")
        out.append("    template <int A, int B> __device__ __forceinline__ void range() const { if constexpr (A < B && A < R::NP) { chunk<A>(); range<A + 1, B>(); } }")
        out.append("    template <int A> __device__ __forceinline__ void rest() const { range<A, R::NP>(); }")
        out.append("};")

    # Fused rollout + geometric Jacobian of the tracked link (round 6; BASELINE config 4: "FK + Jacobian + cost" in ONE launch).  The
    # Jacobian's stateful walk (robot_tree.py:136-248: clamp wherever limits exist, rotation about sf_rot_axis with the axis sign ignored)
    # must coincide with the rollout's stateless one on every link the columns need -- then the columns z_j x (p_ee - p_j) | z_j are read
    # out of the poses this kernel already holds.  Columns: the reference's rule (idx - 1) <= joint_list_idx (robot_tree.py:239-240).
    jacf_cols: List[int] = []
    jacf_ok = False
    if tmpl.ee_link >= 0 and not tmpl.virtual and os.environ.get("TRK_EXP_NO_JAC_FUSE", "0") != "1":
        ee_ = tmpl.ee_link
        jacf_cols = [i for i in range(1, L) if int(kin.dof_idx[i]) >= 0 and int(kin.jac_axis[i]) >= 0 and (i - 1) <= int(kin.joint_list_idx[ee_])]
        need_ = set()
        for leaf in [ee_] + jacf_cols:
            a = leaf
            while a > 0 and a not in need_:
                need_.add(a); a = int(kin.parent[a])
        same_ = all(int(kin.joint_type[i]) == JOINT_FIXED or
                    (int(kin.joint_type[i]) in (JOINT_REVOLUTE, JOINT_CONTINUOUS) and int(kin.clamp[i]) == int(kin.sf_clamp[i]) and
                     int(kin.rot_axis[i]) == int(kin.sf_rot_axis[i]) and float(kin.rot_sign[i]) == 1.0) for i in need_)
        jacf_ok = bool(jacf_cols) and same_

    def emit_collision_objectives(E, t, next_chunk, fast_arg="", prims_ptr=""):
        """cost + position adjoints (tb<i>_k) of the three collision fields on link positions t[i][k]; the scene evaluation owns
        OBJ_TICK_SLOTS tick slots of `flush` per group.  Shared by the fused rollout and the positions-in field kernel."""
        E.raw("    float cost = 0.0f;")
        for i in adj_links:
            E.raw(f"    float tb{i}_0 = 0.0f, tb{i}_1 = 0.0f, tb{i}_2 = 0.0f;")
        t = with_virtual(E, t)
        # leading collision links whose position is a CONSTANT of the model (identity base: the Panda's first link origin): in the
        # GENERAL-scene instantiation (boxes / grid: scene_is_general) their object cost is evaluated once per wavefront, cooperatively
        # (scene_min_sdf_uniform_point), instead of 64 times through the primitive loop; they get no gradient (no joint moves them).
        # The spheres-only instantiation keeps its text (its register allocation is the headline's).
        n_const = 0
        if prims_ptr and os.environ.get("TRK_EXP_NO_UNIFORM_POINT", "0") != "1":
            for i in tmpl.obj_links:
                if all(t[i][k].is_const for k in range(3)) and i not in used_virt:
                    n_const += 1
                else:
                    break
        if 0 < n_const < NL <= LINK_OBJ_GROUP_MAX:
            rest = list(tmpl.obj_links[n_const:])
            nr = len(rest)
            c0 = next_chunk[0]
            next_chunk[0] += OBJ_TICK_SLOTS
            E.raw(f"    const TickFrom<decltype(flush), {c0}> ticks{{flush}};")
            # (a voxel-grid scene keeps all links in one batch: its cost is the gathers' latency, and a separate gather for the constant
            # link in front of the others only adds a second exposed latency: 22.8 -> 23.8 us measured)
            # (TRK_EXP_UNIFORM_SPHERES=1 at generation time: the spheres-only instantiation too -- an experiment knob)
            cond = "!A.C.has_grid" if os.environ.get("TRK_EXP_UNIFORM_SPHERES", "0") == "1" else "BOX && !A.C.has_grid"
            E.raw(f"    if ({cond}) {{")
            for k, nm in enumerate("xyz"):
                E.raw(f"        const float p{nm}[{nr}] = {{{', '.join(E.expr(t[i][k]) for i in rest)}}};")
            E.raw(f"        float gx[{nr}], gy[{nr}], gz[{nr}];")
            E.raw("#pragma unroll")
            E.raw(f"        for (int l = 0; l < {nr}; ++l) {{ gx[l] = 0.0f; gy[l] = 0.0f; gz[l] = 0.0f; }}")
            fa = fast_arg.replace("NL", str(nr))
            E.raw(f"        if (A.w.w_obj != 0.0f) {{")
            for j in range(n_const):      # first: its table reads (LDS) / grid gather are issued before this phase's position ticks
                i = tmpl.obj_links[j]
                E.raw(f"            cost += spec_object_cost_uniform_point(A.C, A.w.w_obj, {', '.join(E.expr(t[i][k]) for k in range(3))}, {j}, lane, {prims_ptr});")
            E.raw(f"            cost += spec_objects_cost<{nr}{fa}>(A.C, A.w.w_obj, px, py, pz, gx, gy, gz, ticks, lds_sph, {n_const}, {prims_ptr});")
            E.raw("        }")
            E.raw(f"        else flush.template range<{c0}, {next_chunk[0]}>();")
            E.raw(f"        if (A.w.w_ws != 0.0f && A.C.has_ws) {{")
            E.raw(f"            cost += spec_ws_cost<{nr}>(A.C, A.w.w_ws, px, py, pz, gx, gy, gz, {n_const});")
            for j in range(n_const):
                i = tmpl.obj_links[j]
                E.raw(f"            {{ float ux_ = 0.0f, uy_ = 0.0f, uz_ = 0.0f; cost += A.w.w_ws * ws_cost_point(A.C, cptr(A.C.obj_link_margin)[{j}], "
                      f"{', '.join(E.expr(t[i][k]) for k in range(3))}, A.w.w_ws, ux_, uy_, uz_); }}")
            E.raw("        }")
            for j, i in enumerate(rest):
                E.raw(f"        tb{i}_0 += gx[{j}]; tb{i}_1 += gy[{j}]; tb{i}_2 += gz[{j}];")
            E.raw("    } else {")
            for k, nm in enumerate("xyz"):
                E.raw(f"        const float p{nm}[NL] = {{{', '.join(E.expr(t[i][k]) for i in tmpl.obj_links)}}};")
            E.raw("        float gx[NL], gy[NL], gz[NL];")
            E.raw("#pragma unroll")
            E.raw("        for (int l = 0; l < NL; ++l) { gx[l] = 0.0f; gy[l] = 0.0f; gz[l] = 0.0f; }")
            E.raw(f"        if (A.w.w_obj != 0.0f) cost += spec_objects_cost<NL{fast_arg}>(A.C, A.w.w_obj, px, py, pz, gx, gy, gz, ticks, lds_sph, 0, {prims_ptr});")
            E.raw(f"        else flush.template range<{c0}, {next_chunk[0]}>();")
            E.raw("        if (A.w.w_ws != 0.0f && A.C.has_ws) cost += spec_ws_cost<NL>(A.C, A.w.w_ws, px, py, pz, gx, gy, gz);")
            for j, i in enumerate(tmpl.obj_links):
                E.raw(f"        tb{i}_0 += gx[{j}]; tb{i}_1 += gy[{j}]; tb{i}_2 += gz[{j}];")
            E.raw("    }")
        elif 0 < NL <= LINK_OBJ_GROUP_MAX:
            for k, nm in enumerate("xyz"):
                E.raw(f"    const float p{nm}[NL] = {{{', '.join(E.expr(t[i][k]) for i in tmpl.obj_links)}}};")
            E.raw("    float gx[NL], gy[NL], gz[NL];")
            E.raw("#pragma unroll")
            E.raw("    for (int l = 0; l < NL; ++l) { gx[l] = 0.0f; gy[l] = 0.0f; gz[l] = 0.0f; }")
            c0 = next_chunk[0]
            next_chunk[0] += OBJ_TICK_SLOTS     # the scene evaluation owns these tick slots, used or flushed on every path
            E.raw(f"    const TickFrom<decltype(flush), {c0}> ticks{{flush}};")
            E.raw(f"    if (A.w.w_obj != 0.0f) cost += spec_objects_cost<NL{fast_arg}>(A.C, A.w.w_obj, px, py, pz, gx, gy, gz, ticks, lds_sph{', 0, ' + prims_ptr if prims_ptr else ''});")
            E.raw(f"    else flush.template range<{c0}, {next_chunk[0]}>();")
            E.raw("    if (A.w.w_ws != 0.0f && A.C.has_ws) cost += spec_ws_cost<NL>(A.C, A.w.w_ws, px, py, pz, gx, gy, gz);")
            for j, i in enumerate(tmpl.obj_links):
                E.raw(f"    tb{i}_0 += gx[{j}]; tb{i}_1 += gy[{j}]; tb{i}_2 += gz[{j}];")
        elif NL > 0:
            # many collision links (tree robots): score them in groups so the working set of one scene evaluation
            # (positions, keys, gradients: ~10 registers per link) does not sit on top of everything the reverse pass keeps
            n_groups = -(-NL // LINK_OBJ_GROUP)
            size = -(-NL // n_groups)
            for g0 in range(0, NL, size):
                grp = list(tmpl.obj_links[g0:g0 + size])
                n = len(grp)
                E.raw("    {")
                for k, nm in enumerate("xyz"):
                    E.raw(f"        const float p{nm}[{n}] = {{{', '.join(E.expr(t[i][k]) for i in grp)}}};")
                E.raw(f"        float gx[{n}], gy[{n}], gz[{n}];")
                E.raw("#pragma unroll")
                E.raw(f"        for (int l = 0; l < {n}; ++l) {{ gx[l] = 0.0f; gy[l] = 0.0f; gz[l] = 0.0f; }}")
                c0 = next_chunk[0]
                next_chunk[0] += OBJ_TICK_SLOTS
                E.raw(f"        const TickFrom<decltype(flush), {c0}> ticks{{flush}};")
                E.raw(f"        if (A.w.w_obj != 0.0f) cost += spec_objects_cost<{n}{fast_arg}>(A.C, A.w.w_obj, px, py, pz, gx, gy, gz, ticks, lds_sph, {g0}{', ' + prims_ptr if prims_ptr else ''});")
                E.raw(f"        else flush.template range<{c0}, {c0 + OBJ_TICK_SLOTS}>();")
                E.raw(f"        if (A.w.w_ws != 0.0f && A.C.has_ws) cost += spec_ws_cost<{n}>(A.C, A.w.w_ws, px, py, pz, gx, gy, gz, {g0});")
                for j, i in enumerate(grp):
                    E.raw(f"        tb{i}_0 += gx[{j}]; tb{i}_1 += gy[{j}]; tb{i}_2 += gz[{j}];")
                E.raw("    }")
        if tmpl.self_pairs:
            E.raw("    if (A.w.w_self != 0.0f) {")
            for pi, (a, b) in enumerate(tmpl.self_pairs):
                pa = ", ".join(E.expr(t[a][k]) for k in range(3))
                pb = ", ".join(E.expr(t[b][k]) for k in range(3))
                E.raw(f"        cost += spec_self_pair(A.w.w_self, cptr(A.C.self_margin)[{pi}], {pa}, {pb}, "
                      f"tb{a}_0, tb{a}_1, tb{a}_2, tb{b}_0, tb{b}_1, tb{b}_2, (A.C.clamp_fields & TRK_FIELD_SELF) != 0);")
            E.raw("    }")
        scatter_virtual(E)

    def emit_boolean_fields(E, t):
        """`hit` = OR of the selected fields' "signed distance < margin" tests on link positions t[i][k] (k_coll after its FK walk,
        k_collf on the caller's positions)"""
        E.raw("    bool hit = false;")
        t = with_virtual(E, t)
        if NL > 0:
            grp_size = NL if NL <= LINK_OBJ_GROUP_MAX else -(-NL // (-(-NL // LINK_OBJ_GROUP)))
            E.raw("    if (A.coll_fields & (TRK_FIELD_OBJECTS | TRK_FIELD_WS)) {")
            for g0 in range(0, NL, grp_size):
                grp = list(tmpl.obj_links[g0:g0 + grp_size])
                n = len(grp)
                E.raw("        {")
                for k, nm in enumerate("xyz"):
                    E.raw(f"            const float p{nm}[{n}] = {{{', '.join(E.expr(t[i][k]) for i in grp)}}};")
                E.raw(f"            hit |= spec_collision_links<{n}>(A.C, A.coll_fields, A.coll_margin, A.coll_use_default, px, py, pz, lds_sph, {g0});")
                E.raw("        }")
            E.raw("    }")
        if tmpl.self_pairs:
            E.raw("    if (A.coll_fields & TRK_FIELD_SELF) {")
            for pi, (a_, b_) in enumerate(tmpl.self_pairs):
                pa = ", ".join(E.expr(t[a_][k]) for k in range(3))
                pb = ", ".join(E.expr(t[b_][k]) for k in range(3))
                E.raw(f"        hit |= spec_self_hit(A.coll_use_default ? cptr(A.C.self_margin)[{pi}] : A.coll_margin, {pa}, {pb});")
            E.raw("    }")

    # GPT: the same kernel with the GP prior fused in ("tree" schedule of trk_rollout_gp_cost_grad: the whole tree at once, like
    # k_rollout -- see k_rollout_gp below for the segment schedule and for what the prior adds)
    gpt_ok = (not chunked) and not tmpl.virtual and D <= 32
    for base_identity, GPT in ((True, False), (False, False)) + (((True, True), (False, True)) if gpt_ok else ()):
        E = Emitter()
        kname = ("k_rollout_gpt_" if GPT else "k_rollout_") + ("bi" if base_identity else "bg")
        # small arms fit 128 VGPRs (4 waves/SIMD: the whole 4096 x 64 batch resident); big trees get 256 VGPRs
        # FAST (two-wavefront kernels only): scene_is_fast(A.C) -- a few equal spheres and nothing else -- is decided at the launch and
        # only that scene path is compiled into the instantiation (5-rep same-box A/B: dual Panda 19.5 -> 18.6 us, UR10 + Allegro
        # 31.0 -> 30.7; the 128-register Panda kernel measured no gain and keeps the run-time branches)
        fast_t = D > 8
        # BOX: the scene has boxes and its primitive table fits TRK_LDS_PRIMS -- the instantiation keeps a copy of the table in LDS
        # and scene_min_sdf carries only (value, index) through its primitive loop (shelf 23.2 -> 22.0 us, maze 24.6 -> 22.2).  The
        # 128-register kernels get it as a switch of its own, decided at the launch: compiled into the one kernel the sphere scene
        # runs as well, the headline measured 9.17 -> 9.37 us (register allocation again).  Two-wavefront kernels: BOX = !FAST.
        # The BOX instantiation is the GENERAL scene kernel: launches choose it with scene_is_general (any box object, or a voxel grid),
        # and only it carries the box and grid text of scene_min_sdf (GENERAL = BOX there).  Behind run-time branches in the
        # sphere-scene kernel the brick-tiled cell index alone cost the headline 9.2 -> 9.55 us (same-box A/B,
        # profiles/r04_ab_headline_*.txt).
        box_t = D <= 8
        if chunked:
            # POS: the launch wants the link positions.  A compile-time switch, because the ring staging costs the launches that
            # only want cost + gradient (the planners' inner loop) 2-4 us even with every store masked off.
            # JAC (units with jacf_ok): the same launch also writes the geometric Jacobian of the tracked link (launch_rjac).
            E.raw(f"template <class IO, bool POS{', bool FAST' if fast_t else ''}{', bool BOX' if box_t else ''}{', bool JAC = false' if (jacf_ok and not GPT) else ''}>      // IO: HBM-side type of q / link_pos / gq (float or _Float16)")
        else:
            E.raw(f"template <class IO{', bool FAST' if fast_t else ''}{', bool BOX' if box_t else ''}{', bool JAC = false' if (jacf_ok and not GPT) else ''}>      // HBM-side type of q / link_pos / gq: float or _Float16")
        if GPT and box_t:
            # the prior's gradient (D registers) lives across the whole kernel: the box-scene instantiation of a small arm would spill
            # 26 registers at four wavefronts per SIMD -- it runs at three
            E.raw(f"__global__ void __launch_bounds__(SPEC_BLOCK, BOX ? 3 : 4) {kname}(SpecArgs A) {{")
        else:
            E.raw(f"__global__ void __launch_bounds__(SPEC_BLOCK, {4 if D <= 8 else 2}) {kname}(SpecArgs A) {{")
        if not box_t:
            E.raw("    constexpr bool BOX = !FAST;")
        if chunked:
            if jacf_ok and not GPT:
                E.raw(f"    constexpr int LDS_LANE = JAC ? {max(rp.stride, D, 3 * D)} : (POS ? {max(rp.stride, D)} : {D});      // JAC: the [64][3D] Jacobian tiles reuse the region")
                E.raw('    static_assert(!JAC || TrkSame<IO, float>::value, "the fused Jacobian is an fp32 output");')
            else:
                E.raw(f"    constexpr int LDS_LANE = POS ? {max(rp.stride, D)} : {D};")
            lds_lane = "LDS_LANE"
        else:
            lds_lane = max(3 * L, D)      # (>= 3 D: the [64][3D] Jacobian tiles of a JAC instantiation fit the staging tile, L > D)
            if jacf_ok and not GPT:
                assert 3 * D <= lds_lane
                E.raw('    static_assert(!JAC || TrkSame<IO, float>::value, "the fused Jacobian is an fp32 output");')
            if GPT:     # the raw q / qd tiles (fp32 at worst) and the factor tile of the prior phase must fit the staging tile
                pad = -(-D // 4) * 4
                need = 2 * (-(-((pad + 65 * D) * 4) // 16) * 16) + (65 * (2 * D + 1) + 2) * 4
                lds_lane = max(lds_lane, -(-need // 256))
        E.raw(f"    __shared__ __attribute__((aligned(16))) float lds_all[SPEC_BLOCK * {lds_lane} + SPEC_WAVES * (TRK_LDS_SPHERES * 4 + (BOX ? TRK_LDS_PRIMS * 8 : 0))];")
        E.raw("    typedef typename IoTraits<IO>::Q IOQ;      // q, link_pos in HBM")
        E.raw("    typedef typename IoTraits<IO>::G IOG;      // gq in HBM (fp16 q: scaled by A.grad_scale, fp16 stores saturate)")
        E.raw("    const int lane = threadIdx.x & (TRK_WAVE - 1);")
        E.raw("    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / TRK_WAVE);   // wave-uniform -> SGPR")
        E.raw(f"    float* lds = lds_all + wave * (TRK_WAVE * {lds_lane});")
        E.raw(f"    float4* lds_sph = reinterpret_cast<float4*>(lds_all + SPEC_BLOCK * {lds_lane}) + wave * (TRK_LDS_SPHERES + (BOX ? 2 * TRK_LDS_PRIMS : 0));")
        E.raw("    float4* lds_prm = BOX ? lds_sph + TRK_LDS_SPHERES : nullptr;        // box scenes: the primitive records, for the winning box's gather")
        E.raw("    const SpheresInFlight sph = spec_load_spheres_issue(A.C, lane);   // waited for together with the rows below")
        E.raw("    SpheresInFlight prm{};")
        E.raw("    if constexpr (BOX) prm = spec_load_prims_issue(A.C, lane);")
        E.raw("    const int64_t wblock = (int64_t)blockIdx.x * SPEC_WAVES + wave;     // index of this wave's 64-sample block")
        E.raw("    const int64_t base = wblock * TRK_WAVE;")
        E.raw("    const int rows = (int)max((int64_t)0, min((int64_t)TRK_WAVE, A.n - base));")
        E.raw("    spec_stamp(A.stamps, wblock, 0, lane);")
        E.raw("    spec_stamp_real(A.stamps, wblock, 2, lane);      // 100 MHz chip-wide clock: aligns the per-CU s_memtime domains")
        E.raw("    float q[D];")
        if not GPT:
            E.raw("    spec_load_q<D>(static_cast<const IOQ*>(A.q), base, rows, lane, lds, q);")
        else:
            E.raw("    typedef RawRowsInFlight<D, IOQ> Raw;")
            E.raw("    constexpr int RS = 2 * D + 1;         // floats per row of the factor tile (odd: conflict-free)")
            E.raw(f"    static_assert(2 * Raw::BYTES + ((TRK_WAVE + 1) * RS + 2) * 4 <= TRK_WAVE * {lds_lane} * 4 && TRK_WAVE * D * 4 <= TRK_WAVE * {lds_lane} * 4, \"the raw q / qd tiles and the factor tile share the staging tile\");")
            E.raw("    const unsigned Hh = (unsigned)A.gp_H;")
            E.raw("    const unsigned t0 = (unsigned)(base % (int64_t)A.gp_H);")
            E.raw("    const unsigned tl = (t0 + (unsigned)lane) % Hh, t_last = (t0 + (unsigned)(TRK_WAVE - 1)) % Hh;")
            E.raw("    const bool edge_prev = rows > 0 && t0 > 0u, edge_next = rows == TRK_WAVE && t_last + 1u < Hh && base + TRK_WAVE < A.n;")
            E.raw("    const Raw rq = spec_raw_rows_issue<D, IOQ>(static_cast<const IOQ*>(A.q), base, rows, lane, edge_prev, edge_next);")
            E.raw("    const Raw rv = spec_raw_rows_issue<D, IOQ>(static_cast<const IOQ*>(A.qd), base, rows, lane, edge_prev, edge_next);")
            E.raw("    const IOQ* qb = spec_raw_rows_finish<D, IOQ>(rq, static_cast<const IOQ*>(A.q), base, rows, lane, reinterpret_cast<IOQ*>(lds));")
            E.raw("    const IOQ* vb = spec_raw_rows_finish<D, IOQ>(rv, static_cast<const IOQ*>(A.qd), base, rows, lane, reinterpret_cast<IOQ*>(reinterpret_cast<unsigned char*>(lds) + Raw::BYTES));")
            E.raw("    float* rt = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(lds) + 2 * Raw::BYTES);     // [65][RS]: row l + 1 = lane l's factor")
            E.raw("    spec_wave_sync();")
            if os.environ.get("TRK_EXP_GP_PRIOR", "dpp") == "lds":
                E.raw("    float gpv[D], cost_gp;")
                E.raw("    {")
                E.raw("        // ---- the prior.  A lane computes ONLY the factor it starts: e_t = (p_t + dt v_t - p_t+1, v_t - v_t+1), r_t = w Q^-1 e_t (zero")
                E.raw("        // when the trajectory ends here); the factor that ends at this sample is the previous lane's, fetched through LDS (the one in")
                E.raw("        // front of the block: lanes 0 .. D-1, one joint each).  d/dq_t = r_t.p - r_t-1.p,  d/dqd_t = dt r_t.p + r_t.v - r_t-1.v.")
                E.raw("        const float wm = (lane < rows && tl + 1u < Hh) ? A.gp_w : 0.0f;")
                E.raw("        const float dt = A.gp_dt, ga = A.gp_a, gb = A.gp_b, gc = A.gp_c;")
                E.raw("        float gvv[D], accg = 0.0f;")
                E.raw("#pragma unroll")
                E.raw("        for (int d = 0; d < D; ++d) {")
                E.raw("            const float p0 = (float)qb[lane * D + d], v0 = (float)vb[lane * D + d];")
                E.raw("            const float pn = (float)qb[(lane + 1) * D + d], vn = (float)vb[(lane + 1) * D + d];")
                E.raw("            const float ep = fmaf(dt, v0, p0) - pn, ev = v0 - vn;")
                E.raw("            const float rp = wm * fmaf(ga, ep, gb * ev), rv_ = wm * fmaf(gb, ep, gc * ev);")
                E.raw("            accg = fmaf(0.5f, fmaf(ep, rp, ev * rv_), accg);")
                E.raw("            rt[(lane + 1) * RS + 2 * d] = rp; rt[(lane + 1) * RS + 2 * d + 1] = rv_;")
                E.raw("            q[d] = p0;")
                E.raw("            if (d % 4 == 3) __builtin_amdgcn_sched_barrier(0);      // four joints in flight: the max-ILP scheduler would hoist all 4 D tile reads (and spill)")
                E.raw("        }")
                E.raw("        {       // the factor between the sample in front of the block and its first sample: lanes 0 .. D-1, joint `lane`.  Branch-free")
                E.raw("                // (the other lanes compute joint 0 again and write a scratch slot behind the tile): as a divergent branch this")
                E.raw("                // block cost the whole kernel 38 registers (217 -> 255, spills in the box-scene instantiation)")
                E.raw("            const int dj = lane < D ? lane : 0;")
                E.raw("            const float pm = (float)qb[dj - D], vm = (float)vb[dj - D], pf = (float)qb[dj], vf = (float)vb[dj];")
                E.raw("            const float wp = edge_prev ? A.gp_w : 0.0f;")
                E.raw("            const float ep = fmaf(dt, vm, pm) - pf, ev = vm - vf;")
                E.raw("            float* slot = rt + (lane < D ? 2 * lane : (TRK_WAVE + 1) * RS);")
                E.raw("            slot[0] = wp * fmaf(ga, ep, gb * ev); slot[1] = wp * fmaf(gb, ep, gc * ev);")
                E.raw("        }")
                E.raw("        spec_wave_sync();")
                E.raw("#pragma unroll")
                E.raw("        for (int d = 0; d < D; ++d) {")
                E.raw("            const float rp = rt[(lane + 1) * RS + 2 * d], rv_ = rt[(lane + 1) * RS + 2 * d + 1];      // re-read: 2 D registers less across the sync")
                E.raw("            gpv[d] = rp - rt[lane * RS + 2 * d];")
                E.raw("            gvv[d] = fmaf(dt, rp, rv_) - rt[lane * RS + 2 * d + 1];")
                E.raw("            if (d % 4 == 3) __builtin_amdgcn_sched_barrier(0);")
                E.raw("        }")
                E.raw("        cost_gp = accg;")
                E.raw("        // d cost / d qd is final: out through the staging tile (its first line waits for every lane's reads of the tiles)")
                E.raw("        spec_store_gq<D, IOG, IoTraits<IO>::kScaled>(static_cast<IOG*>(A.gqd), base, rows, lane, lds, gvv, A.grad_scale);")
                E.raw("        spec_wave_sync();")
                E.raw("    }")
            else:
                # Round 5: the neighbours come through DPP wave shifts instead of LDS.  The phase stamps put "rows in + prior + gqd out" at 36 - 42 %
                # of a config-5 wavefront's life (7300 - 8200 of 20 000 ticks): per joint the LDS form cost four reads, two writes and four
                # more reads behind a second sync -- ten dependent round trips' worth of instructions at two wavefronts per SIMD.  Now a lane
                # reads its own row, the next sample's (p, v) arrive by `wave_shl:1` (lane 63: the row behind the block, handed to the
                # shift as its `old` operand, which a lane without a source keeps), the finished factor (rp, rv) goes to the next lane by
                # `wave_shr:1` (lane 0: the factor in front of the block, computed by lanes 0 .. D-1 as before and read back broadcast).
                # Same expressions on the same values as the LDS form: bit-identical results.
                E.raw("    float gpv[D], cost_gp;")
                E.raw("    {")
                E.raw("        const float wm = (lane < rows && tl + 1u < Hh) ? A.gp_w : 0.0f;")
                E.raw("        const float dt = A.gp_dt, ga = A.gp_a, gb = A.gp_b, gc = A.gp_c;")
                E.raw("        {       // the factor between the sample in front of the block and its first sample: lanes 0 .. D-1, joint `lane`.  Branch-free")
                E.raw("                // (the other lanes compute joint 0 again and write a scratch slot behind the records)")
                E.raw("            const int dj = lane < D ? lane : 0;")
                E.raw("            const float pm = (float)qb[dj - D], vm = (float)vb[dj - D], pf = (float)qb[dj], vf = (float)vb[dj];")
                E.raw("            const float wp = edge_prev ? A.gp_w : 0.0f;")
                E.raw("            const float ep = fmaf(dt, vm, pm) - pf, ev = vm - vf;")
                E.raw("            float* slot = rt + (lane < D ? 2 * lane : 2 * D);")
                E.raw("            slot[0] = wp * fmaf(ga, ep, gb * ev); slot[1] = wp * fmaf(gb, ep, gc * ev);")
                E.raw("        }")
                E.raw("        spec_wave_sync();")
                E.raw("        float gvv[D], accg = 0.0f;")
                E.raw("        const float gaw = wm * ga, gbw = wm * gb, gcw = wm * gc;")
                E.raw("#pragma unroll")
                E.raw("        for (int d = 0; d < D; ++d) {")
                E.raw("            const float p0 = (float)qb[lane * D + d], v0 = (float)vb[lane * D + d];")
                E.raw("            const float pn = trk_dpp_from_next((float)qb[TRK_WAVE * D + d], p0), vn = trk_dpp_from_next((float)vb[TRK_WAVE * D + d], v0);")
                E.raw("            const float ep = fmaf(dt, v0, p0) - pn, ev = v0 - vn;")
                E.raw("            const float rp = fmaf(gaw, ep, gbw * ev), rv_ = fmaf(gbw, ep, gcw * ev);      // w Q^-1 e with the weight folded into Q^-1 once per lane")
                E.raw("            accg = fmaf(ep, rp, fmaf(ev, rv_, accg));                                      // 2 x the factor's cost: halved once, below")
                E.raw("            gpv[d] = rp - trk_dpp_from_prev(rt[2 * d], rp);")
                E.raw("            gvv[d] = fmaf(dt, rp, rv_) - trk_dpp_from_prev(rt[2 * d + 1], rv_);")
                E.raw("            q[d] = p0;")
                E.raw("        }")
                E.raw("        cost_gp = 0.5f * accg;")
                E.raw("        // d cost / d qd is final: out through the staging tile (its first line waits for every lane's reads of the tiles)")
                E.raw("        spec_store_gq<D, IOG, IoTraits<IO>::kScaled>(static_cast<IOG*>(A.gqd), base, rows, lane, lds, gvv, A.grad_scale);")
                E.raw("        spec_wave_sync();")
                E.raw("    }")
        E.raw("    spec_load_spheres_finish(lds_sph, lane, sph);")
        E.raw("    if constexpr (BOX) spec_load_spheres_finish(lds_prm, lane, prm);")
        E.raw("    spec_stamp(A.stamps, wblock, 1, lane);")
        # ---------------- forward ----------------
        R: Dict[int, List[List[S]]] = {}
        t: Dict[int, List[S]] = {}
        passv: Dict[int, S] = {}
        if base_identity:
            R[0] = [[ONE if r == c else ZERO for c in range(3)] for r in range(3)]
            t[0] = [ZERO, ZERO, ZERO]
        else:
            R[0] = [[S(1.0, f"A.base_R[{3 * r + c}]") for c in range(3)] for r in range(3)]
            t[0] = [S(1.0, f"A.base_t[{r}]") for r in range(3)]
        _emit_angles(E, kin)
        if chunked:
            # Many links: a full [64][3L] staging tile per wavefront (92 KB per workgroup for 30 links) leaves ONE workgroup per
            # CU.  The positions go through a 64-float ring per lane instead (RingFlusher): each link is staged as it exists, a
            # complete 32-float chunk leaves piece by piece between the links that fill the other half of the ring.
            W = 3 * L
            E.raw(f"    static_assert({ring_t}::LS == {rp.stride} && {ring_t}::HX == {rp.hx} && {ring_t}::NFULL == {rp.n_full} && {ring_t}::NP == {rp.pieces}, "
                  '"generator and RingFlusher disagree on the ring geometry");')
            E.raw(f"    const {ring_t} ring = spec_make_ring<{rp.W}, {rp.V}, {'true' if rp.aligned else 'false'}, IOQ>("
                  "static_cast<IOQ*>(A.link_pos), base, rows, lane, lds);")
            E.raw("    float* const prow = ring.row();        // this lane's ring; prow_a: the same, shifted by the lane's head")
            E.raw("    float* const prow_a = ring.row_a();")
            E.raw("    spec_wave_sync();")
            ready_at: Dict[int, List[int]] = {}
            for c in range(rp.n_full + 1):
                ready_at.setdefault(rp.ready_float(c), []).append(c)
            pending: List[Tuple[int, int]] = []       # (chunk, piece) not yet issued, oldest first

            def stage_link(i):
                for k in range(3):
                    f = 3 * i + k
                    assert all(rp.reuse_float(c) > f for c, _ in pending), "ring half reused before its pieces were issued"
                    x = E.expr(t[i][k])
                    if rp.regular(f):
                        E.raw(f"    if constexpr (POS) prow_a[{f & 63}] = {x};")
                    else:
                        E.raw(f"    if constexpr (POS) prow[ring.slot({f})] = {x};")
                    if f < rp.hx:
                        E.raw(f"    if constexpr (POS) prow[{64 + f}] = {x};")
                    for c in ready_at.get(f, []):
                        E.raw(f"    if constexpr (POS) ring.template done<{c}>();")
                        if c < rp.n_full:
                            pending.extend((c, kk) for kk in range(rp.pieces))

            def issue_pieces(i):
                """slot after link i: the pending pieces (of one chunk) are spread evenly over the slots that remain until the
                link whose staging re-uses their half of the ring (or the end of the walk)"""
                if not pending:
                    return
                c = pending[0][0]
                last_slot = min(L - 1, rp.reuse_float(c) // 3 - 1)
                slots = max(1, last_slot - i + 1)
                n = -(-sum(1 for cc, _ in pending if cc == c) // slots)
                for _ in range(n):
                    cc, kk = pending.pop(0)
                    E.raw(f"    if constexpr (POS) ring.template piece<{cc}, {kk}>();")
            stage_link(0)
            issue_pieces(0)
        for p in range(1, L):
            i = walk[p]
            _emit_fk_link(E, kin, i, R, t, passv, snap)
            if chunked:
                stage_link(i)
                issue_pieces(i)
        # ---------------- outputs that depend only on FK ----------------
        if chunked:
            assert not pending
            E.raw("    const typename TrkIf<POS, RingTail<decltype(ring)>, NoFlushOf<decltype(ring)>>::type flush{ring};")
        else:
            E.raw(f"    PosFlusher<{3 * L}, IOQ> flush{{reinterpret_cast<const float4*>(lds) + lane, 0u, 0ull, 0ull, lane, make_float4(0.0f, 0.0f, 0.0f, 0.0f)}};")
            pos_list = ", ".join(E.expr(t[i][k]) for i in range(L) for k in range(3))
            E.raw("    if (A.link_pos) {")
            E.raw(f"        const float pv[{3 * L}] = {{{pos_list}}};")
            E.raw(f"        flush = spec_stage_rows<{3 * L}>(static_cast<IOQ*>(A.link_pos), base, rows, lane, lds, pv);")
            E.raw("    }")
        E.raw("    if (!A.gq) { flush.template rest<0>(); return; }      // positions only (trk_fk_positions): wave-uniform exit")
        # position chunks leave at tick points with COMPILE-TIME chunk numbers (PosFlusher::chunk<CH>); `next_chunk` counts them
        next_chunk = [0]

        def tick_line(indent="    "):
            c = next_chunk[0]
            next_chunk[0] += 1
            return f"{indent}flush.template chunk<{c}>();"
        # experiment knob: TRK_EXP_FIRST_BURST=n issues the first n chunks right after staging (default none: +0.35 us with 2)
        first = int(os.environ.get("TRK_EXP_FIRST_BURST", "0"))
        for _ in range(first):
            E.raw(tick_line())
        E.raw("    spec_stamp(A.stamps, wblock, 3, lane);")
        # ---------------- objectives ----------------
        emit_collision_objectives(E, t, next_chunk, fast_arg=f", decltype(ticks), {'FAST' if fast_t else 'false'}, BOX", prims_ptr="lds_prm")
        E.raw("    spec_stamp(A.stamps, wblock, 4, lane);")
        for ee, tgt, rb in tracked:
            E.raw(f"    float {rb}[9] = {{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}};")
        if tracked:
            E.raw("    if (A.w.w_ee != 0.0f) {")
            for ee, tgt, rb in tracked:
                E.raw("      {")
                E.raw(f"        const float eR[9] = {{{', '.join(E.expr(R[ee][r][c]) for r in range(3) for c in range(3))}}};")
                E.raw(f"        const float et[3] = {{{', '.join(E.expr(t[ee][k]) for k in range(3))}}};")
                E.raw("        float gR[9], gt[3];")
                E.raw(f"        const float ce = ee_cost_eval(eR, et, {tgt}, A.C.ee_w_pos, A.C.ee_w_rot, A.C.ee_square, gR, gt);")
                E.raw("        cost = fmaf(A.w.w_ee, ce, cost);")
                E.raw("#pragma unroll")
                E.raw(f"        for (int k = 0; k < 9; ++k) {rb}[k] = A.w.w_ee * gR[k];")
                E.raw(f"        tb{ee}_0 = fmaf(A.w.w_ee, gt[0], tb{ee}_0); tb{ee}_1 = fmaf(A.w.w_ee, gt[1], tb{ee}_1); "
                      f"tb{ee}_2 = fmaf(A.w.w_ee, gt[2], tb{ee}_2);")
                E.raw("      }")
            E.raw("    }")
        E.raw(tick_line())
        E.raw("    spec_stamp(A.stamps, wblock, 5, lane);")
        if GPT:
            E.raw("    cost += cost_gp;           // the prior's factor t -> t + 1, attributed to this sample")
        E.raw("    if (lane < rows) store_wt_f1(A.cost + base + lane, cost);")
        E.raw("    if (A.cost_sum) {")
        E.raw("        const float tot = spec_wave_sum(lane < rows ? cost : 0.0f);")
        # write-through like every other output: the plain 4-byte store left 4096 dirty partial lines for the end-of-kernel
        # write-back (same-box A/B 9.94 -> 9.77 us)
        E.raw("        if (lane == 0 && rows > 0) store_wt_f1(A.cost_sum + wblock, tot);")
        E.raw("    }")
        # ---------------- reverse: wrench accumulators towards the root ----------------
        # (A second FK walk with prefix-sum gradients instead of this reverse pass -- so that the joints' axes / origins need not
        # stay alive -- was measured on UR10+Allegro: 41.3 vs 37.5 us.  These kernels are bound by VALU issue, not by occupancy.)
        gq_expr = _emit_reverse_links(E, kin, R, t, {i: [f"tb{i}_{k}" for k in range(3)] for i in real_adj},
                                      {l: rb for l, _, rb in tracked}, masked, tick=tick_line, order=walk)
        E.raw("    spec_stamp(A.stamps, wblock, 6, lane);")
        E.raw(f"    flush.template rest<{next_chunk[0]}>();")
        if GPT:
            E.raw(f"    const float gv[D] = {{{', '.join(E.expr(gq_expr.get(d, ZERO)) + f' + gpv[{d}]' for d in range(D))}}};")
        else:
            E.raw(f"    const float gv[D] = {{{', '.join(E.expr(gq_expr.get(d, ZERO)) for d in range(D))}}};")
        E.raw("    spec_store_gq<D, IOG, IoTraits<IO>::kScaled>(static_cast<IOG*>(A.gq), base, rows, lane, lds, gv, A.grad_scale);")
        E.raw("    spec_stamp(A.stamps, wblock, 7, lane);")
        if jacf_ok and not GPT:
            ee = tmpl.ee_link
            E.raw("    if constexpr (JAC) {")
            E.raw(f"        // ---- geometric Jacobian of link {ee} '{kin.link_names[ee]}' (robot_tree.py:218-248) from the poses of this walk: column d of a")
            E.raw("        // joint with a column = [z x (p_link - p_joint) ; z], every other column zero; lin_jac / ang_jac [N, 3, D] leave as the")
            E.raw("        // wavefront's contiguous 64 x 3D floats through one LDS tile each (16-byte write-through stores, like k_jac's)")
            pe = [E.named(t[ee][k]) for k in range(3)]
            colz: Dict[int, List[S]] = {}
            coll: Dict[int, List[S]] = {}
            for i in jacf_cols:
                d, ax = int(kin.dof_idx[i]), int(kin.jac_axis[i])
                z = [E.named(R[i][r][ax]) for r in range(3)]
                rel = [E.named(E.lincomb([(pe[k], ONE), (t[i][k], S(-1.0))])) for k in range(3)]
                colz[d] = z
                coll[d] = [E.named(v) for v in E.cross(z, rel)]
            E.raw(f"        float* jrow = lds + lane * {3 * D};")
            for nm, cols, dst in (("lin", coll, "A.jac_lin"), ("ang", colz, "A.jac_ang")):
                E.raw(f"        spec_wave_sync();          // the region's previous readers are done ({nm}_jac)")
                vals = []
                for r in range(3):
                    for d in range(D):
                        vals.append(E.expr(cols[d][r]) if d in cols else "0.0f")
                E.raw("        " + " ".join(f"jrow[{k}] = {v};" for k, v in enumerate(vals)))
                E.raw("        spec_wave_sync();")
                # beyond the Infinity Cache the contiguous tiles are faster as non-temporal stores (UR10 + Allegro, 287 MB per launch, same box,
                # three alternations: 46.4 - 47.7 -> 45.3 - 45.4 us; inside the cache write-through wins, as for the headline's chunks): the launch decides
                E.raw(f"        if (A.jac_stream) spec_store_tile<{3 * D}, true>({dst}, base, rows, lane, lds);")
                E.raw(f"        else spec_store_tile<{3 * D}>({dst}, base, rows, lane, lds);")
            E.raw("        if (lane < rows) {")
            E.raw("            const int64_t s_ = base + lane;")
            E.raw(f"            const float jR[9] = {{{', '.join(E.expr(R[ee][r][c]) for r in range(3) for c in range(3))}}};")
            E.raw("            " + " ".join(f"A.jac_pos[s_ * 3 + {k}] = {E.expr(pe[k])};" for k in range(3)))
            E.raw("            float qo[4];")
            E.raw("            frame_quat_wxyz(jR, qo);")
            E.raw("            *reinterpret_cast<float4*>(A.jac_quat + s_ * 4) = make_float4(qo[0], qo[1], qo[2], qo[3]);")
            E.raw("        }")
            E.raw("    }")
        E.raw("}")
        out.extend(E.lines)
        out.append("")

    # ---- the same objective with one ARM per lane (two isomorphic arms on one base: the dual Panda of BASELINE config 5)
    arm_plan = arm_lane_plan(kin, tmpl) if (gpt_ok and os.environ.get("TRK_EXP_NO_ARM_LANES", "0") != "1") else None
    if arm_plan is not None:
        out.extend(_arm_lane_kernel_lines(kin, tmpl, arm_plan, snap))

    # ---- the fused rollout with the GP prior fused in (trk_rollout_gp_cost_grad; BASELINE config 5's objective in ONE launch, gq and
    # gqd written once).  BUILD-DEFINED like the prior itself.  Two things differ from k_rollout:
    # (1) SEGMENTS.  The subtrees hanging off the root that share no objective (the two arms of the dual Panda when no self pair
    #     crosses them) are evaluated one after the other -- FK, scene, EE, reverse pass, positions -- so that only ONE arm's poses
    #     are live at a time: the dual Panda fits the 128 registers and ~10 KB of LDS per wavefront that let all 4096 wavefronts of
    #     its 2048 x 128 share be resident at once (k_rollout: 189 registers, two generations of workgroups).  Per-lane state that
    #     must survive a segment lives in LDS: the raw q rows (HBM element type), the gradient accumulator [64][D].
    # (2) The prior.  A lane is one (trajectory, time step); its neighbours' q / qd rows sit next to its own in the raw tiles (the
    #     two rows beyond the wavefront's block are fetched with the block), so the prior's gradient is a few FMAs per joint at the
    #     top of the kernel: d/dqd leaves at once, d/dq seeds the accumulator the segments add their gradients to.
    # The factor between t and t + 1 is attributed to sample t: cost[b, t] += w/2 e_t^T Q^-1 e_t.
    par_ = [int(v) for v in kin.parent]
    gp_ok = (not chunked) and not tmpl.virtual and walk == list(range(L)) and D <= 32 and L >= 2
    segs: List[dict] = []
    gp_cross_pairs = False
    if gp_ok:
        top = {}
        for i in range(1, L):
            a = i
            while par_[a] != 0:
                a = par_[a]
            top[i] = a
        roots = [i for i in range(1, L) if par_[i] == 0]
        cand = [[i for i in range(1, L) if top[i] == c] for c in roots]
        contiguous = all(ls == list(range(ls[0], ls[-1] + 1)) for ls in cand) and [l for ls in cand for l in ls] == list(range(1, L))
        # each segment's object-collision links must be one consecutive run of the template (their margins are addressed by a base)
        runs_ok = contiguous
        if contiguous:
            pos_in = 0
            for ls in cand:
                mine = [i for i in tmpl.obj_links if i in ls or (i == 0 and ls is cand[0])]
                if tmpl.obj_links[pos_in:pos_in + len(mine)] != mine:
                    runs_ok = False
                pos_in += len(mine)
        if not (contiguous and runs_ok and len(cand) > 1):
            cand = [list(range(1, L))]
        gp_cross_pairs = any(a != 0 and b != 0 and top[a] != top[b] for a, b in tmpl.self_pairs) if len(cand) > 1 else False
        obj_at = 0
        for k, ls in enumerate(cand):
            own = set(ls) | ({0} if k == 0 else set())
            objs = [i for i in tmpl.obj_links if i in own]
            cols = ([0] if k == 0 else []) + ls                        # links whose positions this segment writes
            seg = dict(links=ls, cols=cols, col0=3 * cols[0], ncol=3 * len(cols), obj=objs, obj_base=obj_at,
                       pairs=[(pi, a, b) for pi, (a, b) in enumerate(tmpl.self_pairs)
                              if (a in own or a == 0) and (b in own or b == 0) and (a in ls or b in ls)],
                       tracked=[(l, tgt, rb) for l, tgt, rb in tracked if l in own],
                       dofs=[int(kin.dof_idx[i]) for i in ls if int(kin.dof_idx[i]) >= 0])
            obj_at += len(objs)
            segs.append(seg)
        if len(segs) > 1 and gp_cross_pairs:
            pass            # served with w_self == 0 only (launch_gp returns 1 otherwise)
    # The segment schedule is compiled only on request (TRK_GP_SCHEDULE=segments at generation time): measured on the dual Panda it
    # is slower than the tree schedule above -- its wavefronts are VALU-bound either way, and what one resident
    # generation gains the serial arms lose in instruction-level parallelism.
    use_seg = gp_ok and (os.environ.get("TRK_GP_SCHEDULE", "tree") == "segments" or not gpt_ok)
    for base_identity in ((True, False) if use_seg else ()):
        E = Emitter()
        kname = "k_rollout_gp_bi" if base_identity else "k_rollout_gp_bg"
        fast_t = D > 8
        box_t = D <= 8
        occ = 4 if max(len(sg["dofs"]) for sg in segs) <= 8 else 2
        E.raw(f"template <class IO{', bool FAST' if fast_t else ''}{', bool BOX' if box_t else ''}>")
        E.raw(f"__global__ void __launch_bounds__(SPEC_BLOCK, {occ}) {kname}(SpecArgs A) {{")
        if not box_t:
            E.raw("    constexpr bool BOX = !FAST;")
        E.raw("    typedef typename IoTraits<IO>::Q IOQ;")
        E.raw("    typedef typename IoTraits<IO>::G IOG;")
        E.raw("    typedef RawRowsInFlight<D, IOQ> Raw;")
        # LDS per wavefront: accumulator [64][D] fp32 | raw q of the later segments [64][DL] | image: first the raw q and qd tiles, then
        # the gqd staging tile, then the positions as they will lie in HBM
        later_dofs = [d for sg in segs[1:] for d in sg["dofs"]]
        DL = len(later_dofs)
        E.raw(f"    constexpr int DL = {DL};            // joints of the segments after the first: their raw q waits in LDS")
        E.raw("    constexpr int ACC_B = TRK_WAVE * D * 4, QL_B = (TRK_WAVE * DL * (int)sizeof(IOQ) + 15) / 16 * 16;")
        E.raw(f"    constexpr int IMG_B0 = TRK_WAVE * {3 * L} * (int)sizeof(IOQ);")
        E.raw("    constexpr int IMG_B1 = IMG_B0 > 2 * Raw::BYTES ? IMG_B0 : 2 * Raw::BYTES;")
        E.raw("    constexpr int IMG_B = ((IMG_B1 > ACC_B ? IMG_B1 : ACC_B) + 15) / 16 * 16;")
        E.raw("    constexpr int WAVE_B = ACC_B + QL_B + IMG_B;")
        E.raw("    // the sphere (and primitive) tables are shared by the workgroup's wavefronts here: one barrier at the top, 192 bytes of LDS per wavefront saved")
        E.raw("    __shared__ __attribute__((aligned(16))) unsigned char lds_all[SPEC_WAVES * WAVE_B + TRK_LDS_SPHERES * 16 + (BOX ? TRK_LDS_PRIMS * 32 : 0)];")
        E.raw("    const int lane = threadIdx.x & (TRK_WAVE - 1);")
        E.raw("    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / TRK_WAVE);")
        E.raw("    unsigned char* wl = lds_all + wave * WAVE_B;")
        E.raw("    float* acc = reinterpret_cast<float*>(wl);                      // [64][D] fp32: d cost / d q, prior first, then the segments")
        E.raw("    IOQ* qlater = reinterpret_cast<IOQ*>(wl + ACC_B);              // [64][DL]: raw q of the later segments' joints")
        E.raw("    IOQ* img = reinterpret_cast<IOQ*>(wl + ACC_B + QL_B);          // the output image; before that: scratch")
        E.raw("    IOQ* qtile = img;                                               // raw q rows of the block (+ the neighbouring rows)")
        E.raw("    float* scr = reinterpret_cast<float*>(wl + ACC_B + QL_B);")
        E.raw("    float4* lds_sph = reinterpret_cast<float4*>(lds_all + SPEC_WAVES * WAVE_B);")
        E.raw("    float4* lds_prm = BOX ? lds_sph + TRK_LDS_SPHERES : nullptr;")
        E.raw("    {")
        E.raw("        const int tid = threadIdx.x;")
        E.raw("        if (tid < TRK_LDS_SPHERES && tid < 2 * A.C.n_sphere_pairs) lds_sph[tid] = A.C.spheres[tid];")
        E.raw("        if constexpr (BOX) { if (A.C.n_box_objects > 0 && A.C.n_prims <= TRK_LDS_PRIMS && tid < 2 * A.C.n_prims) lds_prm[tid] = reinterpret_cast<const float4*>(A.C.prims)[tid]; }")
        E.raw("    }")
        E.raw("    const int64_t wblock = (int64_t)blockIdx.x * SPEC_WAVES + wave;")
        E.raw("    const int64_t base = wblock * TRK_WAVE;")
        E.raw("    const int rows = (int)max((int64_t)0, min((int64_t)TRK_WAVE, A.n - base));")
        E.raw("    // time steps: the block starts at step t0 of its trajectory (wave-uniform), a lane sits at (t0 + lane) mod H")
        E.raw("    const unsigned Hh = (unsigned)A.gp_H;")
        E.raw("    const unsigned t0 = (unsigned)(base % (int64_t)A.gp_H);")
        E.raw("    const unsigned tl = (t0 + (unsigned)lane) % Hh, t_last = (t0 + (unsigned)(TRK_WAVE - 1)) % Hh;")
        E.raw("    const bool edge_prev = rows > 0 && t0 > 0u, edge_next = rows == TRK_WAVE && t_last + 1u < Hh && base + TRK_WAVE < A.n;")
        E.raw("    const Raw rq = spec_raw_rows_issue<D, IOQ>(static_cast<const IOQ*>(A.q), base, rows, lane, edge_prev, edge_next);")
        E.raw("    const Raw rv = spec_raw_rows_issue<D, IOQ>(static_cast<const IOQ*>(A.qd), base, rows, lane, edge_prev, edge_next);")
        E.raw("    const IOQ* qb = spec_raw_rows_finish<D, IOQ>(rq, static_cast<const IOQ*>(A.q), base, rows, lane, qtile);")
        E.raw("    const IOQ* vb = spec_raw_rows_finish<D, IOQ>(rv, static_cast<const IOQ*>(A.qd), base, rows, lane, reinterpret_cast<IOQ*>(wl + ACC_B + QL_B + Raw::BYTES));")
        E.raw("    __syncthreads();            // the shared scene tables (and, wave-locally, the tiles) are in LDS")
        E.raw("    float cost;")
        E.raw("    float " + ", ".join(f"q0_{d}" for d in segs[0]["dofs"]) + ";")
        E.raw("    {")
        E.raw("        // ---- the prior: e_t = (p_t + dt v_t - p_t+1, v_t - v_t+1), r = Q^-1 e; this sample takes part in factors t-1 and t")
        E.raw("        const bool on = lane < rows;")
        E.raw("        const float mn = (on && tl + 1u < Hh) ? 1.0f : 0.0f, mp = (on && tl > 0u) ? 1.0f : 0.0f;")
        E.raw("        const float dt = A.gp_dt, ga = A.gp_a, gb = A.gp_b, gc = A.gp_c;")
        E.raw("        float gpv[D], gvv[D], accg = 0.0f;")
        E.raw("#pragma unroll")
        E.raw("        for (int d = 0; d < D; ++d) {")
        E.raw("            const float p0 = (float)qb[lane * D + d], v0 = (float)vb[lane * D + d];")
        E.raw("            const float pm = (float)qb[(lane - 1) * D + d], vm = (float)vb[(lane - 1) * D + d];")
        E.raw("            const float pn = (float)qb[(lane + 1) * D + d], vn = (float)vb[(lane + 1) * D + d];")
        E.raw("            const float ep = fmaf(dt, v0, p0) - pn, ev = v0 - vn;")
        E.raw("            const float rp = fmaf(ga, ep, gb * ev), rv_ = fmaf(gb, ep, gc * ev);")
        E.raw("            accg = fmaf(0.5f * mn, fmaf(ep, rp, ev * rv_), accg);")
        E.raw("            const float em = fmaf(dt, vm, pm) - p0, fm = vm - v0;")
        E.raw("            gpv[d] = A.gp_w * (mn * rp - mp * fmaf(ga, em, gb * fm));")
        E.raw("            gvv[d] = A.gp_w * (mn * fmaf(dt, rp, rv_) - mp * fmaf(gb, em, gc * fm));")
        E.raw("        }")
        E.raw("        cost = A.gp_w * accg;")
        E.raw("#pragma unroll")
        E.raw("        for (int d = 0; d < D; ++d) acc[lane * D + d] = gpv[d];")
        E.raw("        // the raw q tile is about to be overwritten: the first segment's joints go to registers, the others' wait in LDS")
        for d in segs[0]["dofs"]:
            E.raw(f"        q0_{d} = (float)qb[lane * D + {d}];")
        for k_, d in enumerate(later_dofs):
            E.raw(f"        qlater[lane * DL + {k_}] = qb[lane * D + {d}];")
        E.raw("        // d cost / d qd is final: out through the scratch tile (its first line waits for every lane's reads of the qd rows)")
        E.raw("        spec_store_gq<D, IOG, IoTraits<IO>::kScaled>(static_cast<IOG*>(A.gqd), base, rows, lane, scr, gvv, A.grad_scale);")
        E.raw("    }")
        E.raw("    unsigned passbits = 0u;")
        E.raw(f"    const ImgFlusher<{3 * L}, IOQ> pimg = spec_make_img<{3 * L}, IOQ>(static_cast<IOQ*>(A.link_pos), base, rows, lane, img);")
        E.raw("    spec_wave_sync();          // the gqd staging tile has been read out: the image may be written")
        R: Dict[int, List[List[S]]] = {}
        t: Dict[int, List[S]] = {}
        passv: Dict[int, S] = {}
        if base_identity:
            R[0] = [[ONE if r == c else ZERO for c in range(3)] for r in range(3)]
            t[0] = [ZERO, ZERO, ZERO]
        else:
            R[0] = [[S(1.0, f"A.base_R[{3 * r + c}]") for c in range(3)] for r in range(3)]
            t[0] = [S(1.0, f"A.base_t[{r}]") for r in range(3)]
        for k, sg in enumerate(segs):
            E.raw(f"    // ================= segment {k}: links {sg['links'][0]} .. {sg['links'][-1]} =================")
            E.raw("    {")
            if k == 0:
                _emit_angles(E, kin, links=sg["links"], declare_passbits=False, qexpr=lambda d: f"q0_{d}")
            else:
                _emit_angles(E, kin, links=sg["links"], declare_passbits=False,
                             qexpr=lambda d: f"(float)qlater[lane * DL + {later_dofs.index(d)}]")
            for i in sg["links"]:
                _emit_fk_link(E, kin, i, R, t, passv, snap)
            E.raw("    if (A.link_pos) {")
            for i in sg["cols"]:
                E.raw("        " + " ".join(f"pimg.put({3 * i + kk}, {E.expr(t[i][kk])});" for kk in range(3)))
            E.raw("    }")
            last = k == len(segs) - 1
            # tick slots of this segment: the scene evaluation's, one after the EE term, one per two links of the reverse pass
            n_obj = len(sg["obj"])
            if 0 < n_obj <= LINK_OBJ_GROUP_MAX:
                groups = [sg["obj"]]
            elif n_obj > 0:
                n_groups = -(-n_obj // LINK_OBJ_GROUP)
                size = -(-n_obj // n_groups)
                groups = [sg["obj"][g0:g0 + size] for g0 in range(0, n_obj, size)]
            else:
                groups = []
            n_slots = OBJ_TICK_SLOTS * len(groups) + 1 + sum(1 for p in range(len(sg["links"]), 0, -1) if p % 2 == 0)
            if last:
                E.raw("    spec_wave_sync();          // the image is complete")
                E.raw("    spec_img_copy_slow(pimg, static_cast<IOQ*>(A.link_pos), base, rows);       // ragged last wavefront / unaligned view only")
                E.raw(f"    constexpr int PPT = (decltype(pimg)::NP + {n_slots - 1}) / {n_slots};")
            else:
                E.raw("    constexpr int PPT = 0;     // the image is not complete before the last segment has staged its links")
            E.raw("    const ImgTicks<decltype(pimg), PPT> flush{pimg};")
            next_chunk = [0]

            def tick_line(indent="    "):
                c = next_chunk[0]
                next_chunk[0] += 1
                return f"{indent}flush.template chunk<{c}>();"
            own = set(sg["links"]) | ({0} if k == 0 else set())
            adj = sorted(set(sg["obj"]) | {a for _, a, b in sg["pairs"]} | {b for _, a, b in sg["pairs"]} | {l for l, _, _ in sg["tracked"]})
            for i in adj:
                E.raw(f"    float tb{i}_0 = 0.0f, tb{i}_1 = 0.0f, tb{i}_2 = 0.0f;")
            fast_arg = f", decltype(ticks), {'FAST' if fast_t else 'false'}, BOX"
            g_at = 0
            for grp in groups:
                n = len(grp)
                E.raw("    {")
                for kk, nm in enumerate("xyz"):
                    E.raw(f"        const float p{nm}[{n}] = {{{', '.join(E.expr(t[i][kk]) for i in grp)}}};")
                E.raw(f"        float gx[{n}], gy[{n}], gz[{n}];")
                E.raw("#pragma unroll")
                E.raw(f"        for (int l = 0; l < {n}; ++l) {{ gx[l] = 0.0f; gy[l] = 0.0f; gz[l] = 0.0f; }}")
                c0 = next_chunk[0]
                next_chunk[0] += OBJ_TICK_SLOTS
                E.raw(f"        const TickFrom<decltype(flush), {c0}> ticks{{flush}};")
                E.raw(f"        if (A.w.w_obj != 0.0f) cost += spec_objects_cost<{n}{fast_arg}>(A.C, A.w.w_obj, px, py, pz, gx, gy, gz, ticks, lds_sph, {sg['obj_base'] + g_at}, lds_prm);")
                E.raw(f"        else flush.template range<{c0}, {c0 + OBJ_TICK_SLOTS}>();")
                E.raw(f"        if (A.w.w_ws != 0.0f && A.C.has_ws) cost += spec_ws_cost<{n}>(A.C, A.w.w_ws, px, py, pz, gx, gy, gz, {sg['obj_base'] + g_at});")
                for j, i in enumerate(grp):
                    E.raw(f"        tb{i}_0 += gx[{j}]; tb{i}_1 += gy[{j}]; tb{i}_2 += gz[{j}];")
                E.raw("    }")
                g_at += n
            if sg["pairs"]:
                E.raw("    if (A.w.w_self != 0.0f) {")
                for pi, a, b in sg["pairs"]:
                    pa = ", ".join(E.expr(t[a][kk]) for kk in range(3))
                    pb = ", ".join(E.expr(t[b][kk]) for kk in range(3))
                    E.raw(f"        cost += spec_self_pair(A.w.w_self, cptr(A.C.self_margin)[{pi}], {pa}, {pb}, "
                          f"tb{a}_0, tb{a}_1, tb{a}_2, tb{b}_0, tb{b}_1, tb{b}_2, (A.C.clamp_fields & TRK_FIELD_SELF) != 0);")
                E.raw("    }")
            for ee, tgt, rb in sg["tracked"]:
                E.raw(f"    float {rb}[9] = {{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}};")
            if sg["tracked"]:
                E.raw("    if (A.w.w_ee != 0.0f) {")
                for ee, tgt, rb in sg["tracked"]:
                    E.raw("      {")
                    E.raw(f"        const float eR[9] = {{{', '.join(E.expr(R[ee][r][c]) for r in range(3) for c in range(3))}}};")
                    E.raw(f"        const float et[3] = {{{', '.join(E.expr(t[ee][kk]) for kk in range(3))}}};")
                    E.raw("        float gR[9], gt[3];")
                    E.raw(f"        const float ce = ee_cost_eval(eR, et, {tgt}, A.C.ee_w_pos, A.C.ee_w_rot, A.C.ee_square, gR, gt);")
                    E.raw("        cost = fmaf(A.w.w_ee, ce, cost);")
                    E.raw("#pragma unroll")
                    E.raw(f"        for (int k = 0; k < 9; ++k) {rb}[k] = A.w.w_ee * gR[k];")
                    E.raw(f"        tb{ee}_0 = fmaf(A.w.w_ee, gt[0], tb{ee}_0); tb{ee}_1 = fmaf(A.w.w_ee, gt[1], tb{ee}_1); "
                          f"tb{ee}_2 = fmaf(A.w.w_ee, gt[2], tb{ee}_2);")
                    E.raw("      }")
                E.raw("    }")
            E.raw(tick_line())
            real = [i for i in adj if i != 0]
            gq_expr = _emit_reverse_links(E, kin, R, t, {i: [f"tb{i}_{kk}" for kk in range(3)] for i in real},
                                          {l: rb for l, _, rb in sg["tracked"]}, masked, tick=tick_line, order=[0] + sg["links"],
                                          n_links=len(sg["links"]) + 1)
            assert next_chunk[0] <= n_slots, (next_chunk[0], n_slots)
            E.raw(f"    flush.template rest<{next_chunk[0]}>();")
            for d in sg["dofs"]:
                ex = E.expr(gq_expr.get(d, ZERO))
                if ex != "0.0f":
                    E.raw(f"    acc[lane * D + {d}] += {ex};")
            E.raw("    }")
            # poses of this segment are dead from here on: drop them so that nothing downstream can reference them by accident
            for i in sg["links"]:
                R.pop(i, None); t.pop(i, None)
        E.raw("    if (lane < rows) store_wt_f1(A.cost + base + lane, cost);")
        E.raw("    if (A.cost_sum) {")
        E.raw("        const float tot = spec_wave_sum(lane < rows ? cost : 0.0f);")
        E.raw("        if (lane == 0 && rows > 0) store_wt_f1(A.cost_sum + wblock, tot);")
        E.raw("    }")
        E.raw("    spec_store_acc_tile<D, IOG, IoTraits<IO>::kScaled>(static_cast<IOG*>(A.gq), base, rows, lane, acc, A.grad_scale);")
        E.raw("}")
        out.extend(E.lines)
        out.append("")

    # ---- explicit reverse mode of the link positions (trk_fk_positions_backward with all links selected):
    # FK again (cheaper than storing poses), the adjoint rows [64][3L] come in through the LDS transpose, reverse pass.
    # Many links (the ring-staged units): the whole-row tile is 92 KB per workgroup for 30 links -- one wavefront per SIMD -- and 3L adjoint
    # registers per lane.  Those units take the attached-point generator's kernel instead (columns = the links, chunked loads, prefix-sum
    # gradients; needs the file order to be the pre-order walk): UR10 + Allegro 34.1 -> 29.8 us, iiwa7 + Allegro 33.4 -> 30.9, Shadow hand 32.6 -> 30.4
    # (profiles/r05_bench_positions.txt).
    posbwd_chunked = chunked and [int(v) for v in kin.order] == list(range(L)) and os.environ.get("TRK_EXP_POSBWD_TILE", "0") == "0"
    if posbwd_chunked:
        out.extend(_chunked_posbwd_lines(kin, list(range(L)), np.zeros((L, 3), np.float32), snap, w_expr=str(3 * L)))
    for base_identity in (() if posbwd_chunked else (True, False)):
        E = Emitter()
        kname = "k_posbwd_bi" if base_identity else "k_posbwd_bg"
        E.raw(f"__global__ void __launch_bounds__(SPEC_BLOCK, {4 if D <= 8 else 2}) {kname}(SpecArgs A) {{")
        E.raw(f"    __shared__ __attribute__((aligned(16))) float lds_all[SPEC_BLOCK * {max(3 * L, D)}];")
        E.raw("    const int lane = threadIdx.x & (TRK_WAVE - 1);")
        E.raw("    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / TRK_WAVE);")
        E.raw(f"    float* lds = lds_all + wave * (TRK_WAVE * {max(3 * L, D)});")
        E.raw("    const int64_t wblock = (int64_t)blockIdx.x * SPEC_WAVES + wave;")
        E.raw("    const int64_t base = wblock * TRK_WAVE;")
        E.raw("    const int rows = (int)max((int64_t)0, min((int64_t)TRK_WAVE, A.n - base));")
        E.raw("    float q[D];")
        E.raw("    spec_load_q<D>(static_cast<const float*>(A.q), base, rows, lane, lds, q);")
        early = 3 * L <= 48        # the loads in flight cost 3L/4 registers per lane: only for the small arms
        if early:
            E.raw(f"    // the position adjoints are needed only by the reverse pass: their loads are in flight during the forward pass")
            E.raw(f"    const RowsInFlight<{3 * L}> gp_rows = spec_load_rows_issue<{3 * L}>(static_cast<const float*>(A.link_pos), base, rows, lane);")
        R = {}; t = {}; passv = {}
        if base_identity:
            R[0] = [[ONE if r == c else ZERO for c in range(3)] for r in range(3)]
            t[0] = [ZERO, ZERO, ZERO]
        else:
            R[0] = [[S(1.0, f"A.base_R[{3 * r + c}]") for c in range(3)] for r in range(3)]
            t[0] = [S(1.0, f"A.base_t[{r}]") for r in range(3)]
        _emit_angles(E, kin)
        for p in range(1, L):
            _emit_fk_link(E, kin, int(kin.order[p]), R, t, passv, snap)
        E.raw(f"    float gp[{3 * L}];                       // this sample's position adjoints, link-major")
        if early:
            E.raw(f"    spec_load_rows_finish<{3 * L}>(gp_rows, static_cast<const float*>(A.link_pos), base, rows, lane, lds, gp);")
        else:
            E.raw(f"    spec_load_q<{3 * L}>(static_cast<const float*>(A.link_pos), base, rows, lane, lds, gp);")
        gq_expr = _emit_reverse_links(E, kin, R, t, {i: [f"gp[{3 * i + k}]" for k in range(3)] for i in range(1, L)}, {}, masked)
        E.raw(f"    const float gv[D] = {{{', '.join(E.expr(gq_expr.get(d, ZERO)) for d in range(D))}}};")
        E.raw("    spec_store_gq<D>(static_cast<float*>(A.gq), base, rows, lane, lds, gv);")
        E.raw("}")
        out.extend(E.lines)
        out.append("")

    # ---- boolean mode (trk_rollout_collision): FK + the OR of the selected fields' "signed distance < margin" tests, one byte
    # per sample out.  Its own kernel: inside k_rollout the extra SpecArgs fields and the cold path cost the hot path
    # ~100 SGPR spill moves (v_writelane / v_readlane) per wavefront.
    for base_identity in (True, False):
        E = Emitter()
        kname = "k_coll_bi" if base_identity else "k_coll_bg"
        E.raw(f"__global__ void __launch_bounds__(SPEC_BLOCK, {4 if D <= 8 else 2}) {kname}(SpecArgs A) {{")
        E.raw(f"    __shared__ __attribute__((aligned(16))) float lds_all[SPEC_BLOCK * D + SPEC_WAVES * TRK_LDS_SPHERES * 4];")
        E.raw("    const int lane = threadIdx.x & (TRK_WAVE - 1);")
        E.raw("    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / TRK_WAVE);")
        E.raw("    float* lds = lds_all + wave * (TRK_WAVE * D);")
        E.raw("    float4* lds_sph = reinterpret_cast<float4*>(lds_all + SPEC_BLOCK * D) + wave * TRK_LDS_SPHERES;")
        E.raw("    const SpheresInFlight sph = spec_load_spheres_issue(A.C, lane);   // waited for together with the rows below")
        E.raw("    const int64_t wblock = (int64_t)blockIdx.x * SPEC_WAVES + wave;")
        E.raw("    const int64_t base = wblock * TRK_WAVE;")
        E.raw("    const int rows = (int)max((int64_t)0, min((int64_t)TRK_WAVE, A.n - base));")
        E.raw("    float q[D];")
        E.raw("    unsigned via_slot = 0u;")
        E.raw("    int64_t via_traj0 = 0;")
        E.raw("    bool via_outside = false;")
        E.raw("    if (A.via_n > 0) spec_load_q_via<D>(A, base, rows, lane, q, via_slot, via_traj0, via_outside);     // trajectory validation: interpolate the via points here")
        E.raw("    else spec_load_q<D>(static_cast<const float*>(A.q), base, rows, lane, lds, q);")
        E.raw("    spec_load_spheres_finish(lds_sph, lane, sph);")
        R = {}; t = {}; passv = {}
        if base_identity:
            R[0] = [[ONE if r == c else ZERO for c in range(3)] for r in range(3)]
            t[0] = [ZERO, ZERO, ZERO]
        else:
            R[0] = [[S(1.0, f"A.base_R[{3 * r + c}]") for c in range(3)] for r in range(3)]
            t[0] = [S(1.0, f"A.base_t[{r}]") for r in range(3)]
        _emit_angles(E, kin)
        for p in range(1, L):
            _emit_fk_link(E, kin, int(kin.order[p]), R, t, passv, snap)
        emit_boolean_fields(E, t)
        E.raw("    if (lane < rows) A.coll_out[base + lane] = hit ? 1 : 0;")
        E.raw("    if (A.via_n > 0 && A.via_partial) spec_via_partial_flags(A, wblock, via_traj0, via_slot, hit, via_outside, lane < rows, lane);     // wave-uniform")
        E.raw("}")
        out.extend(E.lines)
        out.append("")

    # ---- FK matrices of ALL links (trk_fk_forward with every link selected = compute_forward_kinematics_all_links,
    # robot_tree.py:267-301): the same stateless walk as the fused rollout; a link's 4x4 leaves as soon as it exists -- each lane
    # puts its 16 floats into a [64][17] LDS tile, the wave writes them as 8-byte write-through vectors, eight consecutive lanes
    # completing one sample's 64 bytes (two whole sectors).  The kernel is a pure write stream (704 B per sample for Panda).
    # Two links per flush (128 bytes per sample) when rows are whole 128-byte lines (L even) and the walk is the file order:
    # full-line writes matter once the output exceeds the 256 MB Infinity Cache (UR10+Allegro, 526 MB: 122 - 154 -> 101 us);
    # with odd L every other row starts mid-line and the pairs straddle lines anyway (Panda, dual Panda: no difference).
    FKH_PAIR = L % 2 == 0 and [int(v) for v in kin.order] == list(range(L))
    FKH_LS = 33 if FKH_PAIR else 17
    for base_identity in (True, False):
        E = Emitter()
        kname = "k_fkh_bi" if base_identity else "k_fkh_bg"
        E.raw(f"__global__ void __launch_bounds__(SPEC_BLOCK, {4 if D <= 16 else 2}) {kname}(SpecArgs A) {{")
        E.raw(f"    __shared__ __attribute__((aligned(16))) float lds_all[SPEC_BLOCK * {max(FKH_LS, D)}];")
        E.raw("    const int lane = threadIdx.x & (TRK_WAVE - 1);")
        E.raw("    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / TRK_WAVE);")
        E.raw(f"    float* lds = lds_all + wave * (TRK_WAVE * {max(FKH_LS, D)});")
        E.raw("    const int64_t wblock = (int64_t)blockIdx.x * SPEC_WAVES + wave;")
        E.raw("    const int64_t base = wblock * TRK_WAVE;")
        E.raw("    const int rows = (int)max((int64_t)0, min((int64_t)TRK_WAVE, A.n - base));")
        E.raw("    float q[D];")
        E.raw("    spec_load_q<D>(static_cast<const float*>(A.q), base, rows, lane, lds, q);")
        R = {}; t = {}; passv = {}
        if base_identity:
            R[0] = [[ONE if r == c else ZERO for c in range(3)] for r in range(3)]
            t[0] = [ZERO, ZERO, ZERO]
        else:
            R[0] = [[S(1.0, f"A.base_R[{3 * r + c}]") for c in range(3)] for r in range(3)]
            t[0] = [S(1.0, f"A.base_t[{r}]") for r in range(3)]
        _emit_angles(E, kin)
        E.raw(f"    float* hrow = lds + lane * {FKH_LS};")

        def emit_h(i, p):
            vals = []
            for r in range(3):
                vals += [E.expr(R[i][r][c]) for c in range(3)] + [E.expr(t[i][r])]
            vals += ["0.0f", "0.0f", "0.0f", "1.0f"]
            if FKH_PAIR:
                off = 16 * (p % 2)
                E.raw("    " + " ".join(f"hrow[{off + k}] = {v};" for k, v in enumerate(vals)))
                if p % 2 == 1:
                    E.raw(f"    spec_flush_chunk<{16 * L}, 32, {FKH_LS}, 2>(A.fk_H, base, {16 * (i - 1)}, rows, lane, lds);")
                elif p == L - 1:
                    E.raw(f"    spec_flush_chunk<{16 * L}, 16, {FKH_LS}, 2>(A.fk_H, base, {16 * i}, rows, lane, lds);")
            else:
                E.raw("    " + " ".join(f"hrow[{k}] = {v};" for k, v in enumerate(vals)))
                E.raw(f"    spec_flush_chunk<{16 * L}, 16, {FKH_LS}, 2>(A.fk_H, base, {16 * i}, rows, lane, lds);")
        emit_h(int(kin.order[0]), 0)
        for p in range(1, L):
            i = int(kin.order[p])
            _emit_fk_link(E, kin, i, R, t, passv, snap)
            emit_h(i, p)
        E.raw("}")
        out.extend(E.lines)
        out.append("")
        # (Round 5: a copy of this kernel with NON-TEMPORAL stores for outputs beyond the Infinity Cache -- what pays for the fused rollout's
        # contiguous 1 KiB chunks, trk_spec_common.h F32Stream -- was built and measured: SLOWER for these 8-byte pieces, dual Panda 90.3 ->
        # 116.1 us, UR10 + Allegro 96.9 -> 101.5 us, Panda (in cache) 27.4 -> 47.2 us; profiles/r05_bench_fkh_stream.txt.  Dropped.)

    # ---- the collision fields on GIVEN link positions (trk_cost_fields: EmbodimentDistanceFieldBase.compute_embodiment_cost,
    # distance_fields.py:107-124, for the fields selected by the caller): positions in through the LDS transpose, the fused
    # kernel's objective code on them, cost + position gradient out.  No kinematics: the unit is found by its collision
    # template (columns = all links of the robot), so the reference-style call `field.compute_cost(q, link_pos)` runs the same
    # arithmetic as the fused rollout.
    fields_ok = 3 * L <= 96 and (NL > 0 or bool(tmpl.self_pairs))
    if fields_ok:
        E = Emitter()
        E.raw(f"__global__ void __launch_bounds__(SPEC_BLOCK, {4 if 3 * L <= 48 else 2}) k_fields(SpecArgs A) {{")
        E.raw(f"    __shared__ __attribute__((aligned(16))) float lds_all[SPEC_BLOCK * {3 * L} + SPEC_WAVES * TRK_LDS_SPHERES * 4];")
        E.raw("    const int lane = threadIdx.x & (TRK_WAVE - 1);")
        E.raw("    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / TRK_WAVE);")
        E.raw(f"    float* lds = lds_all + wave * (TRK_WAVE * {3 * L});")
        E.raw(f"    float4* lds_sph = reinterpret_cast<float4*>(lds_all + SPEC_BLOCK * {3 * L}) + wave * TRK_LDS_SPHERES;")
        E.raw("    const SpheresInFlight sph = spec_load_spheres_issue(A.C, lane);   // waited for together with the rows below")
        E.raw("    const int64_t wblock = (int64_t)blockIdx.x * SPEC_WAVES + wave;")
        E.raw("    const int64_t base = wblock * TRK_WAVE;")
        E.raw("    const int rows = (int)max((int64_t)0, min((int64_t)TRK_WAVE, A.n - base));")
        E.raw(f"    float p[{3 * L}];")
        E.raw(f"    spec_load_q<{3 * L}>(A.fld_pos, base, rows, lane, lds, p);")
        E.raw("    spec_load_spheres_finish(lds_sph, lane, sph);")
        E.raw("    const NoFlush flush;")
        tp = {i: [S(1.0, f"p[{3 * i + k}]") for k in range(3)] for i in range(L)}
        emit_collision_objectives(E, tp, [0])
        E.raw("    if (lane < rows) store_wt_f1(A.cost + base + lane, cost);")
        E.raw("    if (A.fld_g) {")
        E.raw("        const float sc = (A.fld_gcost && lane < rows) ? A.fld_gcost[base + lane] : 1.0f;")
        gl = []
        for i in range(L):
            gl += [f"sc * tb{i}_{k}" if i in real_adj else "0.0f" for k in range(3)]
        E.raw(f"        const float gv[{3 * L}] = {{{', '.join(gl)}}};")
        E.raw(f"        spec_store_gq<{3 * L}>(A.fld_g, base, rows, lane, lds, gv);")
        E.raw("    }")
        E.raw("}")
        out.extend(E.lines)
        out.append("")

    # ---- the same for the boolean fields (trk_collision_fields; distance_fields.py:210-215, 283-291)
    if fields_ok:
        E = Emitter()
        E.raw(f"__global__ void __launch_bounds__(SPEC_BLOCK, {4 if 3 * L <= 48 else 2}) k_collf(SpecArgs A) {{")
        E.raw(f"    __shared__ __attribute__((aligned(16))) float lds_all[SPEC_BLOCK * {3 * L} + SPEC_WAVES * TRK_LDS_SPHERES * 4];")
        E.raw("    const int lane = threadIdx.x & (TRK_WAVE - 1);")
        E.raw("    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / TRK_WAVE);")
        E.raw(f"    float* lds = lds_all + wave * (TRK_WAVE * {3 * L});")
        E.raw(f"    float4* lds_sph = reinterpret_cast<float4*>(lds_all + SPEC_BLOCK * {3 * L}) + wave * TRK_LDS_SPHERES;")
        E.raw("    const SpheresInFlight sph = spec_load_spheres_issue(A.C, lane);   // waited for together with the rows below")
        E.raw("    const int64_t wblock = (int64_t)blockIdx.x * SPEC_WAVES + wave;")
        E.raw("    const int64_t base = wblock * TRK_WAVE;")
        E.raw("    const int rows = (int)max((int64_t)0, min((int64_t)TRK_WAVE, A.n - base));")
        E.raw(f"    float p[{3 * L}];")
        E.raw(f"    spec_load_q<{3 * L}>(A.fld_pos, base, rows, lane, lds, p);")
        E.raw("    spec_load_spheres_finish(lds_sph, lane, sph);")
        emit_boolean_fields(E, {i: [S(1.0, f"p[{3 * i + k}]") for k in range(3)] for i in range(L)})
        E.raw("    if (lane < rows) A.coll_out[base + lane] = hit ? 1 : 0;")
        E.raw("}")
        out.extend(E.lines)
        out.append("")

    # ---- FK matrix of ONE link (trk_fk_forward with a single link selected: compute_forward_kinematics(..., state_less=True)
    # robot_tree.py:192-216, RobotPanda.get_EE_pose robot_panda.py:172-184): the stateless walk with a wave-uniform early exit
    # after the target (a run-time argument, captured by a wave-uniform switch), 64 bytes per sample out through the LDS transpose.
    for base_identity in (True, False):
        E = Emitter()
        kname = "k_fk1_bi" if base_identity else "k_fk1_bg"
        E.raw(f"__global__ void __launch_bounds__(SPEC_BLOCK, {4 if D <= 16 else 2}) {kname}(SpecArgs A) {{")
        E.raw(f"    __shared__ __attribute__((aligned(16))) float lds_all[SPEC_BLOCK * {max(16, D)}];")
        E.raw("    const int lane = threadIdx.x & (TRK_WAVE - 1);")
        E.raw("    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / TRK_WAVE);")
        E.raw(f"    float* lds = lds_all + wave * (TRK_WAVE * {max(16, D)});")
        E.raw("    const int64_t wblock = (int64_t)blockIdx.x * SPEC_WAVES + wave;")
        E.raw("    const int64_t base = wblock * TRK_WAVE;")
        E.raw("    const int rows = (int)max((int64_t)0, min((int64_t)TRK_WAVE, A.n - base));")
        E.raw("    float q[D];")
        E.raw("    spec_load_q<D>(static_cast<const float*>(A.q), base, rows, lane, lds, q);")
        R = {}; t = {}; passv = {}
        if base_identity:
            R[0] = [[ONE if r == c else ZERO for c in range(3)] for r in range(3)]
            t[0] = [ZERO, ZERO, ZERO]
        else:
            R[0] = [[S(1.0, f"A.base_R[{3 * r + c}]") for c in range(3)] for r in range(3)]
            t[0] = [S(1.0, f"A.base_t[{r}]") for r in range(3)]
        _emit_angles(E, kin)
        E.raw("    float hv[16] = {1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f};")

        def capture_h(i):
            body = "; ".join(f"hv[{4 * r + c}] = {E.expr(R[i][r][c])}" for r in range(3) for c in range(3))
            body += "; " + "; ".join(f"hv[{4 * r + 3}] = {E.expr(t[i][r])}" for r in range(3))
            E.raw(f"    if (A.jac_link == {i}) {{ {body}; }}       // wave-uniform")
        capture_h(int(kin.order[0]))
        E.raw("    do {                               // the walk stops after the target's pre-order position")
        for p in range(1, L):
            i = int(kin.order[p])
            E.raw(f"    if (A.jac_p_end <= {p}) break;")
            _emit_fk_link(E, kin, i, R, t, passv, snap)
            capture_h(i)
        E.raw("    } while (0);")
        E.raw("    spec_store_gq<16>(A.fk_H, base, rows, lane, lds, hv);")
        E.raw("}")
        out.extend(E.lines)
        out.append("")

    # ---- reverse mode of the all-links FK matrices (trk_fk_backward with every link selected): FK again, then the reverse
    # walk; a link's adjoint (its 4x4 block of gH, bottom row ignored) comes in through an LDS transpose right before the walk
    # consumes it, so only one block per lane is live.
    FKB_LS = 18
    # every link's rotation stays live until the reverse walk has consumed it: beyond ~24 links the kernel spills (UR10+Allegro:
    # 252 registers, 347 us against the table-driven kernel's 253) -- such robots keep the table-driven reverse mode
    fkhbwd_ok = L <= 24
    for base_identity in ((True, False) if fkhbwd_ok else ()):
        E = Emitter()
        kname = "k_fkhbwd_bi" if base_identity else "k_fkhbwd_bg"
        # two wavefronts per SIMD: the rotation adjoint of EVERY link needs that link's rotation, so all of them stay live (Panda
        # at four per SIMD: 63 registers spilled, and on this ISA a spill reload waits for every outstanding load)
        E.raw(f"__global__ void __launch_bounds__(SPEC_BLOCK, 2) {kname}(SpecArgs A) {{")
        E.raw(f"    __shared__ __attribute__((aligned(16))) float lds_all[SPEC_BLOCK * {max(FKB_LS, D)}];")
        E.raw("    const int lane = threadIdx.x & (TRK_WAVE - 1);")
        E.raw("    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / TRK_WAVE);")
        E.raw(f"    float* lds = lds_all + wave * (TRK_WAVE * {max(FKB_LS, D)});")
        E.raw("    const int64_t wblock = (int64_t)blockIdx.x * SPEC_WAVES + wave;")
        E.raw("    const int64_t base = wblock * TRK_WAVE;")
        E.raw("    const int rows = (int)max((int64_t)0, min((int64_t)TRK_WAVE, A.n - base));")
        E.raw("    float q[D];")
        E.raw("    spec_load_q<D>(static_cast<const float*>(A.q), base, rows, lane, lds, q);")
        R = {}; t = {}; passv = {}
        if base_identity:
            R[0] = [[ONE if r == c else ZERO for c in range(3)] for r in range(3)]
            t[0] = [ZERO, ZERO, ZERO]
        else:
            R[0] = [[S(1.0, f"A.base_R[{3 * r + c}]") for c in range(3)] for r in range(3)]
            t[0] = [S(1.0, f"A.base_t[{r}]") for r in range(3)]
        _emit_angles(E, kin)
        for p in range(1, L):
            _emit_fk_link(E, kin, int(kin.order[p]), R, t, passv, snap)
        E.raw(f"    const float* gh = lds + lane * {FKB_LS};")

        def fetch_adjoint(i):
            E.raw(f"    spec_load_chunk<{16 * L}, 16, {FKB_LS}, 2>(A.fk_H, base, {16 * i}, rows, lane, lds);")
            E.raw(f"    const float gR{i}[9] = {{gh[0], gh[1], gh[2], gh[4], gh[5], gh[6], gh[8], gh[9], gh[10]}};")
            E.raw(f"    const float gt{i}_0 = gh[3], gt{i}_1 = gh[7], gt{i}_2 = gh[11];")
        gq_expr = _emit_reverse_links(E, kin, R, t, {i: [f"gt{i}_{k}" for k in range(3)] for i in range(1, L)},
                                      {i: f"gR{i}" for i in range(1, L)}, masked, pre_link=fetch_adjoint)
        E.raw(f"    const float gv[D] = {{{', '.join(E.expr(gq_expr.get(d, ZERO)) for d in range(D))}}};")
        E.raw("    spec_store_gq<D>(static_cast<float*>(A.gq), base, rows, lane, lds, gv);")
        E.raw("}")
        out.extend(E.lines)
        out.append("")

    # ---- Adam IK on the unit's tracked link (trk_ik_steps; robot_tree.py:345-442): the configurations AND the optimiser state
    # live in registers for all iterations of a launch -- FK, SE3 distance, reverse pass, joint-limit hinge, termination test and
    # the Adam update per lane, nothing but the result goes back to memory.  (The table-driven kernel keeps them in LDS and walks
    # the tree from tables: ~9 us per iteration whatever the batch.)
    # q, m and v of every DOF stay in registers across the loop: beyond ~9 DOF the kernel spills (dual Panda, 14 DOF: 10 registers;
    # UR10+Allegro, 22 DOF: 377) and those robots keep the table-driven kernel
    ik_ok = tmpl.ee_link >= 0 and D <= 9
    for base_identity in ((True, False) if ik_ok else ()):
        E = Emitter()
        kname = "k_ik_bi" if base_identity else "k_ik_bg"
        E.raw(f"__global__ void __launch_bounds__(SPEC_BLOCK, {4 if D <= 8 else 2}) {kname}(IkArgs A) {{")
        E.raw("    __shared__ __attribute__((aligned(16))) float lds_all[SPEC_BLOCK * D];")
        E.raw("    const int lane = threadIdx.x & (TRK_WAVE - 1);")
        E.raw("    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / TRK_WAVE);")
        E.raw("    float* lds = lds_all + wave * (TRK_WAVE * D);")
        E.raw("    const int64_t wblock = (int64_t)blockIdx.x * SPEC_WAVES + wave;")
        E.raw("    const int64_t base = wblock * TRK_WAVE;")
        E.raw("    const int rows = (int)max((int64_t)0, min((int64_t)TRK_WAVE, A.n - base));")
        E.raw("    float q[D], am[D], av[D];")
        E.raw("    spec_load_q<D>(static_cast<const float*>(A.q), base, rows, lane, lds, q);")
        E.raw("    if (A.lr > 0.0f) {")
        E.raw("        spec_load_q<D>(static_cast<const float*>(A.adam_m), base, rows, lane, lds, am);")
        E.raw("        spec_load_q<D>(static_cast<const float*>(A.adam_v), base, rows, lane, lds, av);")
        E.raw("    } else {")
        E.raw("#pragma unroll")
        E.raw("        for (int d = 0; d < D; ++d) { am[d] = 0.0f; av[d] = 0.0f; }")
        E.raw("    }")
        E.raw("    float Ht[16];")
        E.raw("    {")
        E.raw("        const float* tp = A.H_target + (A.per_sample ? min(base + lane, A.n - 1) * 16 : 0);")
        E.raw("#pragma unroll")
        E.raw("        for (int k = 0; k < 12; ++k) Ht[k] = tp[k];")
        E.raw("    }")
        E.raw("    float loss0 = 0.0f;")
        E.raw("    bool ok0 = false;")
        E.raw("    for (int it = 0; it < A.n_steps; ++it) {")
        R = {}; t = {}; passv = {}
        if base_identity:
            R[0] = [[ONE if r == c else ZERO for c in range(3)] for r in range(3)]
            t[0] = [ZERO, ZERO, ZERO]
        else:
            R[0] = [[S(1.0, f"A.base_R[{3 * r + c}]") for c in range(3)] for r in range(3)]
            t[0] = [S(1.0, f"A.base_t[{r}]") for r in range(3)]
        _emit_angles(E, kin)
        # only the chain of the tracked link matters: the other branches' poses would be dead code
        chain, a = set(), tmpl.ee_link
        while a >= 0:
            chain.add(a); a = int(kin.parent[a])
        for p in range(1, L):
            i = int(kin.order[p])
            if i in chain:
                _emit_fk_link(E, kin, i, R, t, passv, snap)
        ee = tmpl.ee_link
        E.raw(f"    const float eR[9] = {{{', '.join(E.expr(R[ee][r][c]) for r in range(3) for c in range(3))}}};")
        E.raw(f"    const float et[3] = {{{', '.join(E.expr(t[ee][k]) for k in range(3))}}};")
        E.raw("    float gR[9], gt[3];")
        E.raw("    const float err = ee_cost_eval(eR, et, Ht, 1.0f, 1.0f, 0, gR, gt);      // SE3_distance, w_pos = w_rot = 1 (robot_tree.py:386-417)")
        order_chain = [int(kin.order[0])] + [int(kin.order[p]) for p in range(1, L) if int(kin.order[p]) in chain]
        gq_expr = _emit_reverse_links(E, kin, R, t, {ee: ["gt[0]", "gt[1]", "gt[2]"]}, {ee: "gR"}, masked, order=order_chain,
                                      n_links=len(order_chain))
        E.raw(f"    const float g_[D] = {{{', '.join(E.expr(gq_expr.get(d, ZERO)) for d in range(D))}}};")
        E.raw("    bool ok = err < A.se3_eps;")
        E.raw("    float jl = 0.0f;")
        E.raw("    const float bc1 = A.sched.bc1[it], rsqrt_bc2 = A.sched.rsqrt_bc2[it];")
        E.raw("#pragma unroll")
        E.raw("    for (int d = 0; d < D; ++d) {      // hinge on the (shrunk) limits, validity, torch.optim.Adam's update")
        E.raw("        const float qv = q[d], lo = cptr(A.lower)[d], hi = cptr(A.upper)[d];")
        E.raw("        float g = g_[d];")
        E.raw("        if (qv < lo) { const float e = lo - qv; jl = fmaf(e, e, jl); g = fmaf(-2.0f * A.w_jl, e, g); }")
        E.raw("        if (qv > hi) { const float e = hi - qv; jl = fmaf(e, e, jl); g = fmaf(-2.0f * A.w_jl, e, g); }")
        E.raw("        ok = ok && (qv >= lo) && (qv <= hi);")
        E.raw("        if (A.lr > 0.0f) {")
        E.raw("            const float m1 = fmaf(0.9f, am[d], 0.1f * g);")
        E.raw("            const float v1 = fmaf(0.999f, av[d], 0.001f * g * g);")
        E.raw("            am[d] = m1; av[d] = v1;")
        E.raw("            const float denom = fmaf(sqrtf(v1), rsqrt_bc2, 1e-8f);")
        E.raw("            q[d] = qv - (A.lr / bc1) * (m1 / denom);")
        E.raw("        }")
        E.raw("    }")
        E.raw("    if (it == 0) { loss0 = fmaf(A.w_jl, jl, err); ok0 = ok; }")
        E.raw("    }")
        E.raw("    if (lane < rows) {")
        E.raw("        if (A.loss) A.loss[base + lane] = loss0;")
        E.raw("        if (A.valid) A.valid[base + lane] = ok0 ? 1 : 0;")
        E.raw("    }")
        E.raw("    if (A.lr > 0.0f) {")
        E.raw("        spec_store_gq<D>(A.q, base, rows, lane, lds, q);")
        E.raw("        spec_store_gq<D>(A.adam_m, base, rows, lane, lds, am);")
        E.raw("        spec_store_gq<D>(A.adam_v, base, rows, lane, lds, av);")
        E.raw("    }")
        E.raw("}")
        out.extend(E.lines)
        out.append("")

    # ---- Gauss-Newton / Levenberg-Marquardt IK on the unit's tracked link (trk_ik_gn_steps; BUILD-DEFINED: the reference's loop is
    # Adam, robot_tree.py:303-384, its geometric Jacobian :218-248 is what a Newton step is made of; oracle: orc_ik_gn_step).  Per
    # iteration and lane: stateful FK of the links that matter, the Jacobian columns of the reference's rule, the pose residual,
    # J^T J + lambda I and J^T r as straight-line FMAs on the symbolic columns (structural zeros fold away), a Cholesky solve in
    # registers, the clamped step.  The Jacobian never leaves the registers -- the two-launch form (trk_fk_jacobian + trk_jtj)
    # writes and re-reads 416 bytes per sample and iteration.
    ikgn_cols = [i for i in range(1, L) if int(kin.dof_idx[i]) >= 0 and int(kin.jac_axis[i]) >= 0 and tmpl.ee_link >= 0 and
                 (i - 1) <= int(kin.joint_list_idx[tmpl.ee_link])]
    ikgn_ok = tmpl.ee_link >= 0 and D <= 9 and len(ikgn_cols) > 0
    for base_identity in ((True, False) if ikgn_ok else ()):
        E = Emitter()
        kname = "k_ikgn_bi" if base_identity else "k_ikgn_bg"
        E.raw(f"__global__ void __launch_bounds__(SPEC_BLOCK, {4 if D <= 8 else 2}) {kname}(IkGnArgs A) {{")
        E.raw("    __shared__ __attribute__((aligned(16))) float lds_all[SPEC_BLOCK * D];")
        E.raw("    const int lane = threadIdx.x & (TRK_WAVE - 1);")
        E.raw("    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / TRK_WAVE);")
        E.raw("    float* lds = lds_all + wave * (TRK_WAVE * D);")
        E.raw("    const int64_t wblock = (int64_t)blockIdx.x * SPEC_WAVES + wave;")
        E.raw("    const int64_t base = wblock * TRK_WAVE;")
        E.raw("    const int rows = (int)max((int64_t)0, min((int64_t)TRK_WAVE, A.n - base));")
        E.raw("    float q[D];")
        E.raw("    spec_load_q<D>(static_cast<const float*>(A.q), base, rows, lane, lds, q);")
        E.raw("    float Ht[12];")
        E.raw("    {")
        E.raw("        const float* tp = A.H_target + (A.per_sample ? min(base + lane, A.n - 1) * 16 : 0);")
        E.raw("#pragma unroll")
        E.raw("        for (int k = 0; k < 12; ++k) Ht[k] = tp[k];")
        E.raw("    }")
        E.raw("    float lo[D], hi[D];")
        E.raw("#pragma unroll")
        E.raw("    for (int d = 0; d < D; ++d) { lo[d] = cptr(A.lower)[d]; hi[d] = cptr(A.upper)[d]; }")
        E.raw("    float err0 = 0.0f;")
        E.raw("    bool ok0 = false;")
        E.raw("    for (int it = 0; it < A.n_steps; ++it) {")
        R = {}; t = {}; passv = {}
        if base_identity:
            R[0] = [[ONE if r == c else ZERO for c in range(3)] for r in range(3)]
            t[0] = [ZERO, ZERO, ZERO]
        else:
            R[0] = [[S(1.0, f"A.base_R[{3 * r + c}]") for c in range(3)] for r in range(3)]
            t[0] = [S(1.0, f"A.base_t[{r}]") for r in range(3)]
        need = set()
        for leaf in [tmpl.ee_link] + ikgn_cols:
            a = leaf
            while a >= 0 and a not in need:
                need.add(a); a = int(kin.parent[a])
        rot_dofs = []
        for i in range(1, L):
            jt, d = int(kin.joint_type[i]), int(kin.dof_idx[i])
            if jt == JOINT_FIXED or i not in need:
                continue
            if kin.sf_clamp[i]:                  # rigid_body.py:218-224: the stateful path clamps whenever limits exist
                E.raw(f"    const float qh{d} = __builtin_amdgcn_fmed3f(q[{d}], {flit(kin.lower[i])}, {flit(kin.upper[i])});")
            else:
                E.raw(f"    const float qh{d} = q[{d}];")
            if jt in (JOINT_REVOLUTE, JOINT_CONTINUOUS):
                rot_dofs.append(d)
        for d in rot_dofs:
            E.raw(f"    float sn{d}, cs{d};")
        for a, b in zip(rot_dofs[0::2], rot_dofs[1::2]):
            E.raw(f"    trk_sincos2(qh{a}, qh{b}, &sn{a}, &cs{a}, &sn{b}, &cs{b});")
        if len(rot_dofs) % 2:
            d = rot_dofs[-1]
            E.raw(f"    trk_sincos(qh{d}, &sn{d}, &cs{d});")
        for p in range(1, L):
            i = int(kin.order[p])
            if i in need:
                _emit_fk_link(E, kin, i, R, t, passv, snap, stateful=True)
        ee = tmpl.ee_link
        E.raw(f"    const float eR[9] = {{{', '.join(E.expr(R[ee][r][c]) for r in range(3) for c in range(3))}}};")
        E.raw(f"    const float et[3] = {{{', '.join(E.expr(t[ee][k]) for k in range(3))}}};")
        # residual r = [p* - p ; rotvec(R* R^T)]
        E.raw("    float Re[9];")
        E.raw("#pragma unroll")
        E.raw("    for (int a = 0; a < 3; ++a)")
        E.raw("#pragma unroll")
        E.raw("        for (int b = 0; b < 3; ++b) Re[3 * a + b] = fmaf(Ht[4 * a], eR[3 * b], fmaf(Ht[4 * a + 1], eR[3 * b + 1], Ht[4 * a + 2] * eR[3 * b + 2]));")
        E.raw("    float r6[6] = {Ht[3] - et[0], Ht[7] - et[1], Ht[11] - et[2], 0.0f, 0.0f, 0.0f};")
        E.raw("    trk_rotvec(Re, r6 + 3);")
        E.raw("    if (it == 0) {          // what the caller learns about q as passed in: the metric of ik_termination (robot_tree.py:419-442)")
        E.raw("        float gR_[9], gt_[3];")
        E.raw("        err0 = ee_cost_eval(eR, et, Ht, 1.0f, 1.0f, 0, gR_, gt_);")
        E.raw("        ok0 = err0 < A.se3_eps;")
        E.raw("#pragma unroll")
        E.raw("        for (int d = 0; d < D; ++d) ok0 = ok0 && (q[d] >= lo[d]) && (q[d] <= hi[d]);")
        E.raw("    }")
        E.raw("    const float lam = fmaf(A.lm_gain, fmaf(r6[0], r6[0], fmaf(r6[1], r6[1], fmaf(r6[2], r6[2], fmaf(r6[3], r6[3], fmaf(r6[4], r6[4], r6[5] * r6[5]))))), A.damping);")
        # symbolic Jacobian columns: rows 0-2 linear z x (p_ee - p_joint), rows 3-5 angular z
        col: Dict[int, List[S]] = {}
        pe = [S(1.0, f"et[{k}]") for k in range(3)]
        for i in ikgn_cols:
            d, ax = int(kin.dof_idx[i]), int(kin.jac_axis[i])
            z = [E.named(R[i][r][ax]) for r in range(3)]
            rel = [E.named(E.lincomb([(pe[k], ONE), (t[i][k], S(-1.0))])) for k in range(3)]
            lin = [E.named(v) for v in E.cross(z, rel)]
            col[d] = lin + z
        rr = [S(1.0, f"r6[{k}]") for k in range(6)]
        for d in range(D):
            if d in col:
                E.raw(f"    const float g{d}_ = {E.expr(E.lincomb([(col[d][k], rr[k]) for k in range(6)]))};")
        E.raw(f"    float An[{D * (D + 1) // 2}], gn[D];")
        for i in range(D):
            E.raw(f"    gn[{i}] = {f'g{i}_' if i in col else '0.0f'};")
            for j in range(i + 1):
                idx = i * (i + 1) // 2 + j
                if i in col and j in col:
                    a = E.lincomb([(col[i][k], col[j][k]) for k in range(6)])
                    E.raw(f"    An[{idx}] = {E.expr(a)}{' + lam' if i == j else ''};")
                else:
                    E.raw(f"    An[{idx}] = {'lam' if i == j else '0.0f'};")
        E.raw("    trk_chol_solve<D>(An, gn);")
        E.raw("#pragma unroll")
        E.raw("    for (int d = 0; d < D; ++d) q[d] = __builtin_amdgcn_fmed3f(fmaf(A.step_scale, gn[d], q[d]), lo[d], hi[d]);")
        E.raw("    }")
        E.raw("    if (lane < rows) {")
        E.raw("        if (A.err) A.err[base + lane] = err0;")
        E.raw("        if (A.valid) A.valid[base + lane] = ok0 ? 1 : 0;")
        E.raw("    }")
        E.raw("    spec_store_gq<D>(A.q, base, rows, lane, lds, q);")
        E.raw("}")
        out.extend(E.lines)
        out.append("")

    # ---- stateful FK + geometric Jacobian of ONE link (trk_fk_jacobian; robot_tree.py:136-190, 218-248): the walk unrolled
    # with the stateful path's quirks (clamp wherever limits exist, rotation about the axis with its sign ignored); every
    # joint that can receive a column leaves a record (z, p) in LDS, the target link (a run-time argument) is picked by a
    # wave-uniform switch, and the read-out is the table-driven kernel's (trk_jac_readout).  One wavefront per workgroup.
    jac_joints = [i for i in range(1, L) if int(kin.dof_idx[i]) >= 0 and int(kin.jac_axis[i]) >= 0]
    NJ = len(jac_joints)
    RS = (6 * NJ + 3) | 1
    # Small arms: the lane writes its columns straight into two [64][3D] output tiles (static offsets; non-ancestor columns stay
    # zero), finishes z x (p_link - p_joint) in place, and the read-out is a contiguous 16-byte copy.  The record scheme below
    # needs less LDS when the target link has few ancestors among many joints (a finger tip of UR10+Allegro: 10 of 22), but its
    # read-out maps every output element through the slot table at run time -- half of the Panda kernel's time.
    direct = D <= JAC_DIRECT_MAX_DOFS
    JAC_LDS = 6 * D if direct else max(RS, D)
    for base_identity in (True, False):
        E = Emitter()
        kname = "k_jac_bi" if base_identity else "k_jac_bg"
        E.raw(f"__global__ void __launch_bounds__(TRK_WAVE) {kname}(SpecArgs A) {{")
        E.raw("    extern __shared__ __attribute__((aligned(16))) float lds[];     // 64 x max(record stride, D) floats + the slot table")
        if not direct:
            E.raw("    const int rstride = (6 * A.jac_n_cols + 3) | 1;       // records only for the joints that get a column")
        E.raw("    const int lane = threadIdx.x;")
        E.raw("    const int64_t base = (int64_t)blockIdx.x * TRK_WAVE;")
        E.raw("    const int rows = (int)max((int64_t)0, min((int64_t)TRK_WAVE, A.n - base));")
        E.raw("    float q[D];")
        E.raw("    spec_load_q<D>(static_cast<const float*>(A.q), base, rows, lane, lds, q);")
        R = {}; t = {}; passv = {}
        if base_identity:
            R[0] = [[ONE if r == c else ZERO for c in range(3)] for r in range(3)]
            t[0] = [ZERO, ZERO, ZERO]
        else:
            R[0] = [[S(1.0, f"A.base_R[{3 * r + c}]") for c in range(3)] for r in range(3)]
            t[0] = [S(1.0, f"A.base_t[{r}]") for r in range(3)]
        rot_dofs = []
        for i in range(1, L):
            jt, d = int(kin.joint_type[i]), int(kin.dof_idx[i])
            if jt == JOINT_FIXED:
                continue
            if kin.sf_clamp[i]:                  # rigid_body.py:218-224: clamp whenever limits exist, continuous joints too
                E.raw(f"    const float qh{d} = __builtin_amdgcn_fmed3f(q[{d}], {flit(kin.lower[i])}, {flit(kin.upper[i])});")
            else:
                E.raw(f"    const float qh{d} = q[{d}];")
            if jt in (JOINT_REVOLUTE, JOINT_CONTINUOUS):
                rot_dofs.append(d)
        for d in rot_dofs:
            E.raw(f"    float sn{d}, cs{d};")
        for a, b in zip(rot_dofs[0::2], rot_dofs[1::2]):
            E.raw(f"    trk_sincos2(qh{a}, qh{b}, &sn{a}, &cs{a}, &sn{b}, &cs{b});")
        if len(rot_dofs) % 2:
            d = rot_dofs[-1]
            E.raw(f"    trk_sincos(qh{d}, &sn{d}, &cs{d});")
        if direct:
            E.raw(f"    float* lin_row = lds + lane * {3 * D};                       // this sample's row of the two [64][3D] output tiles")
            E.raw(f"    float* ang_row = lds + TRK_WAVE * {3 * D} + lane * {3 * D};")
            E.raw("    spec_wave_sync();                  // the q transpose is done with this LDS")
            E.raw("#pragma unroll")
            E.raw(f"    for (int k = 0; k < {3 * D}; ++k) {{ lin_row[k] = 0.0f; ang_row[k] = 0.0f; }}")
        else:
            E.raw("    float* rec = lds + lane * rstride;")
            E.raw("    spec_wave_sync();                  // the q transpose is done with this LDS")
        E.raw("    float eR[9] = {1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f}, et[3] = {0.0f, 0.0f, 0.0f};")

        def capture(i):
            rl = "; ".join(f"eR[{3 * r + c}] = {E.expr(R[i][r][c])}" for r in range(3) for c in range(3))
            tl = "; ".join(f"et[{r}] = {E.expr(t[i][r])}" for r in range(3))
            E.raw(f"    if (A.jac_link == {i}) {{ {rl}; {tl}; }}       // wave-uniform")
        capture(int(kin.order[0]))
        E.raw("    do {                               // the walk stops after the last pre-order position that matters")
        for p in range(1, L):
            i = int(kin.order[p])
            E.raw(f"    if (A.jac_p_end <= {p}) break;")
            _emit_fk_link(E, kin, i, R, t, passv, snap, stateful=True)
            if i in jac_joints:
                d, ax = int(kin.dof_idx[i]), int(kin.jac_axis[i])
                vals = [E.expr(R[i][r][ax]) for r in range(3)] + [E.expr(t[i][r]) for r in range(3)]
                if direct:      # the axis is final (angular rows); the joint origin waits in the linear rows for the link's position
                    body = " ".join([f"ang_row[{r * D + d}] = {vals[r]};" for r in range(3)] +
                                    [f"lin_row[{r * D + d}] = {vals[3 + r]};" for r in range(3)])
                    E.raw(f"    if (A.jac_slot[{d}] >= 0) {{ {body} }}      // wave-uniform")
                else:
                    body = " ".join(f"j[{r}] = {v};" for r, v in enumerate(vals))
                    E.raw(f"    if (A.jac_slot[{d}] >= 0) {{ float* j = rec + 6 * A.jac_slot[{d}]; {body} }}      // wave-uniform")
            capture(i)
        E.raw("    } while (0);")
        if direct:
            for i in jac_joints:
                d = int(kin.dof_idx[i])
                E.raw(f"    if (A.jac_slot[{d}] >= 0) {{      // wave-uniform: lin = z x (p_link - p_joint)")
                E.raw(f"        const float z0 = ang_row[{d}], z1 = ang_row[{D + d}], z2 = ang_row[{2 * D + d}];")
                E.raw(f"        const float r0 = et[0] - lin_row[{d}], r1 = et[1] - lin_row[{D + d}], r2 = et[2] - lin_row[{2 * D + d}];")
                E.raw(f"        lin_row[{d}] = z1 * r2 - z2 * r1; lin_row[{D + d}] = z2 * r0 - z0 * r2; lin_row[{2 * D + d}] = z0 * r1 - z1 * r0;")
                E.raw("    }")
            E.raw("    spec_wave_sync();")
            E.raw(f"    spec_store_tile<{3 * D}>(A.jac_lin, base, rows, lane, lds);")
            E.raw(f"    spec_store_tile<{3 * D}>(A.jac_ang, base, rows, lane, lds + TRK_WAVE * {3 * D});")
        else:
            E.raw("    rec[6 * A.jac_n_cols] = et[0]; rec[6 * A.jac_n_cols + 1] = et[1]; rec[6 * A.jac_n_cols + 2] = et[2];")
            E.raw(f"    int* slot = reinterpret_cast<int*>(lds + TRK_WAVE * max(rstride, {D}));")
            for d in range(D):
                E.raw(f"    if (lane == {d}) slot[{d}] = A.jac_slot[{d}];")
            E.raw("    spec_wave_sync();")
            E.raw("    trk_jac_readout(lds, slot, rstride, A.jac_n_cols, D, rows, A.jac_lin + base * 3 * D, A.jac_ang + base * 3 * D, lane);")
        E.raw("    if (lane >= rows) return;")
        E.raw("    const int64_t s = base + lane;")
        E.raw("    A.jac_pos[s * 3] = et[0]; A.jac_pos[s * 3 + 1] = et[1]; A.jac_pos[s * 3 + 2] = et[2];")
        E.raw("    float qo[4];")
        E.raw("    frame_quat_wxyz(eR, qo);")
        E.raw("    *reinterpret_cast<float4*>(A.jac_quat + s * 4) = make_float4(qo[0], qo[1], qo[2], qo[3]);")
        E.raw("}")
        out.extend(E.lines)
        out.append("")

    # ---- analytic Jacobian of EVERY link (trk_fk_analytic_jacobian; robot_tree.py:250-265): d [pos, quat_wxyz] / d q, [N, L, 7, D].  The
    # stateless walk unrolled; a joint leaves (omega = pass * sign * its axis in the world, its origin) in registers, every link combines the
    # joints of its chain into its 7 x D block in an LDS tile (off-chain columns are literal zeros), and the block leaves as the sample's
    # contiguous run of 7 D floats (spec_store_link_block).  One wavefront per workgroup.  The table-driven kernel walks the same
    # tables at one instruction per ~25 cycles (Panda 4096 x 64: 340 us for 572 MB).
    ajac_ok = D <= 16 and L <= 24 and 7 * D * L >= RING_FLOATS and os.environ.get("TRK_EXP_NO_AJAC", "0") != "1"
    AJ_W = 7 * D * L                        # floats of a sample's output row
    arp = ring_plan(AJ_W) if ajac_ok else None
    if ajac_ok:
        out.append("#ifndef __HIPCC_RTC__          // (linked / dlopen-ed units only: a code-object unit keeps the table-driven kernel)")
        dof_link = {int(kin.dof_idx[i]): i for i in range(1, L) if int(kin.dof_idx[i]) >= 0}
        chain: Dict[int, set] = {}
        for i in range(L):
            c_, a_ = set(), i
            while a_ > 0:
                c_.add(a_); a_ = int(kin.parent[a_])
            chain[i] = c_
        aring_t = f"RingFlusher<{arp.W}, {arp.V}, {'true' if arp.aligned else 'false'}, float>"
        for base_identity in (True, False):
            E = Emitter()
            kname = "k_ajac_bi" if base_identity else "k_ajac_bg"
            E.raw(f"__global__ void __launch_bounds__(TRK_WAVE) {kname}(SpecArgs A) {{")
            E.raw(f"    extern __shared__ __attribute__((aligned(16))) float lds[];     // the wavefront's ring [64][{arp.stride}] (first: the q transpose)")
            E.raw("    const int lane = threadIdx.x;")
            E.raw("    const int64_t base = (int64_t)blockIdx.x * TRK_WAVE;")
            E.raw("    const int rows = (int)max((int64_t)0, min((int64_t)TRK_WAVE, A.n - base));")
            E.raw("    float q[D];")
            E.raw("    spec_load_q<D>(static_cast<const float*>(A.q), base, rows, lane, lds, q);")
            _emit_angles(E, kin)
            E.raw(f"    static_assert({aring_t}::LS == {arp.stride} && {aring_t}::HX == {arp.hx} && {aring_t}::NFULL == {arp.n_full} && {aring_t}::NP == {arp.pieces}, "
                  '"generator and RingFlusher disagree on the ring geometry");')
            E.raw("    spec_wave_sync();                  // the q transpose is done with this LDS")
            # (write-through pieces.  Non-temporal ones were measured for outputs beyond the Infinity Cache: 2.4 -> 1.8 TB/s -- an 8-byte piece
            # is not the 1 KiB contiguous store that policy pays for)
            E.raw(f"    const {aring_t} ring = spec_make_ring<{arp.W}, {arp.V}, {'true' if arp.aligned else 'false'}, float>(A.jac_lin, base, rows, lane, lds);")
            E.raw("    float* const prow = ring.row();        // this lane's ring; prow_a: the same, shifted by the lane's head")
            E.raw("    float* const prow_a = ring.row_a();")
            R = {}; t = {}; passv = {}
            if base_identity:
                R[0] = [[ONE if r == c else ZERO for c in range(3)] for r in range(3)]
                t[0] = [ZERO, ZERO, ZERO]
            else:
                R[0] = [[S(1.0, f"A.base_R[{3 * r + c}]") for c in range(3)] for r in range(3)]
                t[0] = [S(1.0, f"A.base_t[{r}]") for r in range(3)]
            ready_at: Dict[int, List[int]] = {}
            for c in range(arp.n_full + 1):
                ready_at.setdefault(arp.ready_float(c), []).append(c)

            def stage_float(f, x):
                """float f of the sample's row into the ring; a chunk that is complete leaves at once (its 16 pieces back to back: the next
                floats overwrite the OTHER half of the ring, nothing waits for these stores)"""
                if arp.regular(f):
                    E.raw(f"        prow_a[{f & 63}] = {x};")
                else:
                    E.raw(f"        prow[ring.slot({f})] = {x};")
                if f < arp.hx:
                    E.raw(f"        prow[{64 + f}] = {x};")
                for c in ready_at.get(f, []):
                    E.raw(f"        ring.template done<{c}>();")
                    for kk in range(arp.pieces):
                        E.raw(f"        ring.template piece<{c}, {kk}>();")

            def emit_block(i):
                E.raw(f"    {{   // link {i}: its 7 x D block of the Jacobian")
                on = []
                for d in range(D):
                    jd = dof_link[d]
                    if jd in chain[i] and (int(kin.joint_type[jd]) == JOINT_PRISMATIC or float(kin.rot_sign[jd]) != 0.0):
                        on.append(d)
                if any(int(kin.joint_type[dof_link[d]]) != JOINT_PRISMATIC for d in on):
                    E.raw("        const float Ri[9] = {" + ", ".join(E.expr(R[i][r][c]) for r in range(3) for c in range(3)) + "};")
                    E.raw("        const QuatSel qs = quat_sel(Ri);")
                tt = ", ".join(E.expr(t[i][r]) for r in range(3))
                for d in on:
                    if int(kin.joint_type[dof_link[d]]) == JOINT_PRISMATIC:
                        E.raw(f"        const float c{d}[7] = {{aw{d}_0, aw{d}_1, aw{d}_2, 0.0f, 0.0f, 0.0f, 0.0f}};")
                    else:
                        E.raw(f"        float c{d}[7];")
                        E.raw(f"        spec_ajac_col_revolute(c{d}, Ri, qs, {tt}, aw{d}_0, aw{d}_1, aw{d}_2, ap{d}_0, ap{d}_1, ap{d}_2);")
                for k in range(7):
                    for d in range(D):
                        stage_float(7 * D * i + k * D + d, f"c{d}[{k}]" if d in on else "0.0f")
                E.raw("    }")
            # the row is staged in MEMORY order (link index), the walk visits the links in pre-order: a link's block is emitted when every
            # link in front of it (by index) has been walked
            walked = {int(kin.order[0])}
            next_block = [0]

            def flush_blocks():
                while next_block[0] < L and next_block[0] in walked:
                    emit_block(next_block[0])
                    next_block[0] += 1
            flush_blocks()
            for p in range(1, L):
                i = int(kin.order[p])
                _emit_fk_link(E, kin, i, R, t, passv, snap)
                jt, d = int(kin.joint_type[i]), int(kin.dof_idx[i])
                if jt != JOINT_FIXED and d >= 0:
                    mask = (lambda e, d=d: f"((passbits & {1 << d}u) ? {e} : 0.0f)") if kin.clamp[i] else (lambda e: e)
                    if jt == JOINT_PRISMATIC:
                        par = int(kin.parent[i])
                        dirw = [E.lincomb([(R[par][r][k], S(float(kin.axis[i][k]))) for k in range(3)]) for r in range(3)]
                        for k in range(3):
                            E.raw(f"    const float aw{d}_{k} = {mask(E.expr(dirw[k]))};")
                    elif float(kin.rot_sign[i]) != 0.0:
                        sg, ax = float(kin.rot_sign[i]), int(kin.rot_axis[i])
                        for k in range(3):
                            z = R[i][k][ax]
                            E.raw(f"    const float aw{d}_{k} = {mask(E.expr(S(z.c * sg, z.n)))};")
                            E.raw(f"    const float ap{d}_{k} = {E.expr(t[i][k])};")
                walked.add(i)
                flush_blocks()
            assert next_block[0] == L
            E.raw("}")                      # (the tail chunk -- what is left of every row and the head of the next one -- left with the last float)
            out.extend(E.lines)
            out.append("")
        out.append("#endif      // !__HIPCC_RTC__")

    out.append("#ifndef __HIPCC_RTC__          // the unit's host half: launchers and its registry entry")
    obj = ", ".join(str(i) for i in tmpl.obj_links) or "0"
    pairs = ", ".join(f"{a}, {b}" for a, b in tmpl.self_pairs) or "0"
    out.append(f"static const int32_t kObjLinks[] = {{{obj}}};")
    out.append(f"static const int32_t kSelfPairs[] = {{{pairs}}};")
    vsrc = ", ".join(f"{r[0]}, {r[1]}" for r in tmpl.virtual) or "0"
    vw = ", ".join(f"{flit(float(np.float32(r[2])))}, {flit(float(np.float32(r[3])))}" for r in tmpl.virtual) or "0.0f"
    out.append(f"static const int32_t kVirtualSrc[] = {{{vsrc}}};")
    out.append(f"static const float kVirtualW[] = {{{vw}}};")
    out.append("static void launch(const SpecEntry*, const SpecArgs& a, int base_identity, hipStream_t st) {")
    out.append("    const unsigned grid = (unsigned)((a.n + SPEC_BLOCK - 1) / SPEC_BLOCK);")
    # every (IO, [POS,] [FAST,] [BOX,] base) instantiation: one generic lambda per compile-time switch, in template-parameter order
    switches = []
    if chunked:
        switches.append("a.link_pos != nullptr")
    if D > 8:
        switches.append("scene_is_fast(a.C)")
    else:
        switches.append("scene_is_general(a.C)")     # BOX: box objects ((value, index) primitive loop over the LDS table) and / or a voxel grid
    n_sw = len(switches)
    targs = ", ".join(f"decltype(c{k})::value" for k in range(n_sw))
    params = ", ".join(f"auto c{k}" for k in range(n_sw))
    out.append(f"    auto go = [&](auto io, {params}) {{")
    out.append("        using IOT = decltype(io);")
    out.append(f"        if (base_identity) hipLaunchKernelGGL((k_rollout_bi<IOT, {targs}>), dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
    out.append("        // (a robot moved by update_base_pose keeps the write-through stores beyond the cache: one instantiation less per unit)")
    out.append(f"        else hipLaunchKernelGGL((k_rollout_bg<typename TrkIf<TrkSame<IOT, F32Stream>::value, float, IOT>::type, {targs}>), dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
    out.append("    };")
    prev = "go"
    for k in range(n_sw - 1, -1, -1):             # innermost lambda decides the LAST switch
        fixed = ", ".join(f"auto c{m}" for m in range(k))
        fixed_args = ", ".join(f"c{m}" for m in range(k))
        sep = ", " if k else ""
        out.append(f"    auto sw{k} = [&](auto io{sep}{fixed}) {{ if ({switches[k]}) {prev}(io{sep}{fixed_args}, std::true_type{{}}); "
                   f"else {prev}(io{sep}{fixed_args}, std::false_type{{}}); }};")
        prev = f"sw{k}"
    out.append("    // fp32 launches whose working set exceeds the Infinity Cache take the non-temporal-store instantiation (spec_stream_stores)")
    out.append(f"    if (a.io_f16 == TRK_IO_F16) sw0(_Float16{{}}); else if (a.io_f16 == TRK_IO_F16_G32) sw0(HalfG32{{}}); "
               f"else if (spec_stream_stores(a, {L}, {D})) sw0(F32Stream{{}}); else sw0(float{{}});")
    out.append("}")
    if jacf_ok:
        # fused rollout + geometric Jacobian of the tracked link (trk_rollout_jacobian_cost_grad): fp32 I/O, positions wanted;
        # returns 1 when this call is not served (the C ABI then runs the two launches)
        out.append("static int launch_rjac(const SpecEntry*, const SpecArgs& a, int base_identity, hipStream_t st) {")
        # ring-staged units instantiate JAC with positions (POS = true) only; the others take the positions as a run-time option
        out.append(f"    if (a.io_f16 != TRK_IO_F32 || {'!a.link_pos || ' if chunked else ''}a.jac_link != {tmpl.ee_link}) return 1;")
        out.append("    const unsigned grid = (unsigned)((a.n + SPEC_BLOCK - 1) / SPEC_BLOCK);")
        pos_t = "true, " if chunked else ""
        out.append(f"    if ({'scene_is_fast(a.C)' if D > 8 else 'scene_is_general(a.C)'}) {{")
        out.append(f"        if (base_identity) hipLaunchKernelGGL((k_rollout_bi<float, {pos_t}true, true>), dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
        out.append(f"        else hipLaunchKernelGGL((k_rollout_bg<float, {pos_t}true, true>), dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
        out.append("    } else {")
        out.append(f"        if (base_identity) hipLaunchKernelGGL((k_rollout_bi<float, {pos_t}false, true>), dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
        out.append(f"        else hipLaunchKernelGGL((k_rollout_bg<float, {pos_t}false, true>), dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
        out.append("    }")
        out.append("    return 0;")
        out.append("}")
    if gp_ok or gpt_ok:
        out.append("static int launch_gp(const SpecEntry*, const SpecArgs& a, int base_identity, hipStream_t st) {")
        out.append("    const unsigned grid = (unsigned)((a.n + SPEC_BLOCK - 1) / SPEC_BLOCK);")
        sw = "scene_is_fast(a.C)" if D > 8 else "scene_is_general(a.C)"
        if use_seg and gp_cross_pairs:
            out.append("    if (a.w.w_self != 0.0f) return 1;      // self pairs between independently scheduled subtrees: the two-launch form serves them")
        kn = "k_rollout_gp_" if use_seg else "k_rollout_gpt_"
        if arm_plan is not None:
            # one arm per lane: sphere scenes, no (cross-arm) self pairs -- and fp32 I/O only by default.  Measured on one box, alternating
            # (profiles/r06_ab_c5_arm_lanes.txt, _box1.txt): fp32 I/O 23.0 - 23.5 -> 20.5 - 22.7 us box to box, but fp16 I/O 21.0 -> 22.1 - 22.4 us and mixed 20.8 -> 22.2: at four
            # wavefronts per SIMD the lone-wavefront latency drops (9.1 -> 7.6 us) while every wavefront of 32 samples costs 1.9 us of issue
            # time against 2.3 us for 64 samples -- 19 % more instructions per arm (SQ counters: profiles/r06_sq_c5_arm_vs_robot_lane.txt).
            # TRK_GP_ARM_LANES=0 / 1 in the environment forces the choice (read per launch, like TRK_STREAM_STORES: same-process A/Bs).
            out.append("    {")
            out.append("        const char* env_ = std::getenv(\"TRK_GP_ARM_LANES\");")
            out.append("        const bool want_ = env_ ? std::atoi(env_) != 0 : a.io_f16 == TRK_IO_F32;")
            out.append("        if (a.w.w_self == 0.0f && scene_is_fast(a.C) && want_) {")
            out.append("            const unsigned grid2 = (unsigned)((a.n + SPEC_WAVES * (TRK_WAVE / 2) - 1) / (SPEC_WAVES * (TRK_WAVE / 2)));")
            out.append("            auto ga = [&](auto io) {")
            out.append("                using IOT = decltype(io);")
            out.append("                if (base_identity) hipLaunchKernelGGL((k_rollout_gpa_bi<IOT>), dim3(grid2), dim3(SPEC_BLOCK), 0, st, a);")
            out.append("                else hipLaunchKernelGGL((k_rollout_gpa_bg<IOT>), dim3(grid2), dim3(SPEC_BLOCK), 0, st, a);")
            out.append("            };")
            out.append("            if (a.io_f16 == TRK_IO_F16) ga(_Float16{}); else if (a.io_f16 == TRK_IO_F16_G32) ga(HalfG32{}); else ga(float{});")
            out.append("            return 0;")
            out.append("        }")
            out.append("    }")
        out.append("    auto go = [&](auto io, auto c0) {")
        out.append("        using IOT = decltype(io);")
        out.append(f"        if (base_identity) hipLaunchKernelGGL(({kn}bi<IOT, decltype(c0)::value>), dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
        out.append(f"        else hipLaunchKernelGGL(({kn}bg<IOT, decltype(c0)::value>), dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
        out.append("    };")
        out.append(f"    auto sw = [&](auto io) {{ if ({sw}) go(io, std::true_type{{}}); else go(io, std::false_type{{}}); }};")
        out.append("    if (a.io_f16 == TRK_IO_F16) sw(_Float16{}); else if (a.io_f16 == TRK_IO_F16_G32) sw(HalfG32{}); else sw(float{});")
        out.append("    return 0;")
        out.append("}")
    out.append("static void launch_posbwd(const SpecEntry*, const SpecArgs& a, int base_identity, hipStream_t st) {")
    out.append("    const unsigned grid = (unsigned)((a.n + SPEC_BLOCK - 1) / SPEC_BLOCK);")
    out.append("    if (base_identity) hipLaunchKernelGGL(k_posbwd_bi, dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
    out.append("    else hipLaunchKernelGGL(k_posbwd_bg, dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
    out.append("}")
    out.append("static void launch_coll(const SpecEntry*, const SpecArgs& a, int base_identity, hipStream_t st) {")
    out.append("    const unsigned grid = (unsigned)((a.n + SPEC_BLOCK - 1) / SPEC_BLOCK);")
    out.append("    if (base_identity) hipLaunchKernelGGL(k_coll_bi, dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
    out.append("    else hipLaunchKernelGGL(k_coll_bg, dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
    out.append("}")
    out.append("static void launch_fkh(const SpecEntry*, const SpecArgs& a, int base_identity, hipStream_t st) {")
    out.append("    const unsigned grid = (unsigned)((a.n + SPEC_BLOCK - 1) / SPEC_BLOCK);")
    out.append("    if (base_identity) hipLaunchKernelGGL(k_fkh_bi, dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
    out.append("    else hipLaunchKernelGGL(k_fkh_bg, dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
    out.append("}")
    if fkhbwd_ok:
        out.append("static void launch_fkhbwd(const SpecEntry*, const SpecArgs& a, int base_identity, hipStream_t st) {")
        out.append("    const unsigned grid = (unsigned)((a.n + SPEC_BLOCK - 1) / SPEC_BLOCK);")
        out.append("    if (base_identity) hipLaunchKernelGGL(k_fkhbwd_bi, dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
        out.append("    else hipLaunchKernelGGL(k_fkhbwd_bg, dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
        out.append("}")
    if fields_ok:
        out.append("static void launch_fields(const SpecEntry*, const SpecArgs& a, int, hipStream_t st) {      // coll_out set: the boolean fields")
        out.append("    const unsigned grid = (unsigned)((a.n + SPEC_BLOCK - 1) / SPEC_BLOCK);")
        out.append("    if (a.coll_out) hipLaunchKernelGGL(k_collf, dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
        out.append("    else hipLaunchKernelGGL(k_fields, dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
        out.append("}")
    out.append("static void launch_fk1(const SpecEntry*, const SpecArgs& a, int base_identity, hipStream_t st) {")
    out.append("    const unsigned grid = (unsigned)((a.n + SPEC_BLOCK - 1) / SPEC_BLOCK);")
    out.append("    if (base_identity) hipLaunchKernelGGL(k_fk1_bi, dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
    out.append("    else hipLaunchKernelGGL(k_fk1_bg, dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
    out.append("}")
    if ik_ok:
        out.append("static void launch_ik(const SpecEntry*, const IkArgs& a, int base_identity, hipStream_t st) {")
        out.append("    const unsigned grid = (unsigned)((a.n + SPEC_BLOCK - 1) / SPEC_BLOCK);")
        out.append("    if (base_identity) hipLaunchKernelGGL(k_ik_bi, dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
        out.append("    else hipLaunchKernelGGL(k_ik_bg, dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
        out.append("}")
    if ikgn_ok:
        out.append("static void launch_ikgn(const SpecEntry*, const IkGnArgs& a, int base_identity, hipStream_t st) {")
        out.append("    const unsigned grid = (unsigned)((a.n + SPEC_BLOCK - 1) / SPEC_BLOCK);")
        out.append("    if (base_identity) hipLaunchKernelGGL(k_ikgn_bi, dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
        out.append("    else hipLaunchKernelGGL(k_ikgn_bg, dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
        out.append("}")
    out.append("static void launch_jac(const SpecEntry*, const SpecArgs& a, int base_identity, hipStream_t st) {")
    out.append("    const unsigned grid = (unsigned)((a.n + TRK_WAVE - 1) / TRK_WAVE);")
    if direct:
        out.append(f"    const size_t lds = sizeof(float) * (size_t)TRK_WAVE * {6 * D};")
    else:
        out.append("    const int rstride = (6 * a.jac_n_cols + 3) | 1;")
        out.append("    const size_t lds = sizeof(float) * ((size_t)TRK_WAVE * (rstride > D ? rstride : D) + TRK_MAX_DOFS);")
    out.append("    if (base_identity) hipLaunchKernelGGL(k_jac_bi, dim3(grid), dim3(TRK_WAVE), lds, st, a);")
    out.append("    else hipLaunchKernelGGL(k_jac_bg, dim3(grid), dim3(TRK_WAVE), lds, st, a);")
    out.append("}")
    if ajac_ok:
        out.append("static void launch_ajac(const SpecEntry*, const SpecArgs& a, int base_identity, hipStream_t st) {")
        out.append("    const unsigned grid = (unsigned)((a.n + TRK_WAVE - 1) / TRK_WAVE);")
        out.append(f"    const size_t lds = sizeof(float) * (size_t)TRK_WAVE * {max(arp.stride, D)};")
        out.append("    if (base_identity) hipLaunchKernelGGL(k_ajac_bi, dim3(grid), dim3(TRK_WAVE), lds, st, a);")
        out.append("    else hipLaunchKernelGGL(k_ajac_bg, dim3(grid), dim3(TRK_WAVE), lds, st, a);")
        out.append("}")
    jac_ok = (TRK_WAVE_ * JAC_LDS + 32) * 4 <= 64 * 1024            # default dynamic-LDS limit of a launch
    out.append(f"static const SpecEntry kEntry = {{SPEC_ENTRY_STAMP, 0x{model_hash(kin):016x}ull, L, D, NL, kObjLinks, "
               f"{len(tmpl.self_pairs)}, kSelfPairs, {tmpl.ee_link}, \"{ident}\", launch, 0, 0ull, launch_posbwd, {tmpl.ee2_link}, "
               f"{'launch_jac' if jac_ok else 'nullptr'}, launch_coll, launch_fkh, {'launch_fkhbwd' if fkhbwd_ok else 'nullptr'}, "
               f"{'launch_ik' if ik_ok else 'nullptr'}, launch_fk1, {'launch_fields' if fields_ok else 'nullptr'}, "
               f"{len(tmpl.virtual)}, kVirtualSrc, kVirtualW, {'launch_ikgn' if ikgn_ok else 'nullptr'}, {'launch_gp' if (gp_ok or gpt_ok) else 'nullptr'}, nullptr, "
               f"{'launch_rjac' if jacf_ok else 'nullptr'}, {'launch_ajac' if ajac_ok else 'nullptr'}}};")
    out.append("static struct Reg { Reg() { trk_spec_register(&kEntry); } } reg;")
    out.append("#endif      // !__HIPCC_RTC__")
    out.append(f"}}  // namespace spec_{ident}")
    if meta is not None:
        ns = f"spec_{ident}::"
        ios = ("float", "_Float16", "HalfG32")
        n_sw = (1 if chunked else 0) + 1
        sw_sets = [[]]
        for _ in range(n_sw):
            sw_sets = [v + [b] for v in sw_sets for b in ("false", "true")]
        names = []
        for b in ("bi", "bg"):
            for io in ios:
                for sw in sw_sets:
                    names.append(f"{ns}k_rollout_{b}<{', '.join([io] + sw)}>")
                if gpt_ok and not use_seg:
                    names += [f"{ns}k_rollout_gpt_{b}<{io}, {v}>" for v in ("false", "true")]
            names += [f"{ns}k_posbwd_{b}", f"{ns}k_coll_{b}", f"{ns}k_fkh_{b}", f"{ns}k_fk1_{b}"]
            if fkhbwd_ok:
                names.append(f"{ns}k_fkhbwd_{b}")
            if ik_ok:
                names.append(f"{ns}k_ik_{b}")
            if ikgn_ok:
                names.append(f"{ns}k_ikgn_{b}")
            if jac_ok:
                names.append(f"{ns}k_jac_{b}")
        if fields_ok:
            names += [f"{ns}k_fields", f"{ns}k_collf"]
        meta.update(ident=ident, kernels=names, chunked=bool(chunked), fast_switch=bool(D > 8), fkhbwd_ok=bool(fkhbwd_ok), fields_ok=bool(fields_ok),
                    ik_ok=bool(ik_ok), ikgn_ok=bool(ikgn_ok), jac_ok=bool(jac_ok), jac_direct=bool(direct), gp_ok=bool(gpt_ok and not use_seg),
                    n_links=L, n_dofs=D)
    return "\n".join(out) + "\n"


# ----------------------------------------------------------------------------------------------------------------------
# Attached-point kernels: the cost model's columns are points fixed in link frames (link spheres, grasped-object points)
# ----------------------------------------------------------------------------------------------------------------------
def points_hash(point_link, point_offset) -> int:
    """FNV-1a (64 bit) over (n_points, point_link, point_offset) -- the same bytes trk_point_set_create hashes."""
    pl = np.ascontiguousarray(point_link, np.int32).reshape(-1)
    po = np.ascontiguousarray(point_offset, np.float32).reshape(-1, 3)
    h = 0xcbf29ce484222325
    for arr in (np.asarray([len(pl)], np.int32), pl, po.reshape(-1)):
        for b in arr.tobytes():
            h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


@dataclass
class PointsTemplate:
    """Point set + collision model baked into a generated attached-point kernel.  Columns must be produced in
    increasing order by a walk of the tree (each link's columns follow those of the links visited before it)."""
    point_link: np.ndarray                  # int32 [P]
    point_offset: np.ndarray                # float32 [P, 3], link frame
    obj_cols: List[int]                     # columns evaluated against objects / workspace box, margins in this order
    self_pairs: List[Tuple[int, int]] = field(default_factory=list)   # COLUMN pairs
    ee_link: int = -1                       # LINK index
    ee2_link: int = -1                      # second tracked LINK (two-arm scenes)


CHUNKED_STAGING_MIN_FLOATS = 80       # link kernels with more position floats per sample than this stream them out in chunks
LINK_OBJ_GROUP_MAX = 12    # link kernels: up to this many collision links are scored in one scene evaluation (more ILP: dual Panda 5+5 was ~1 us slower) ...
LINK_OBJ_GROUP = 5         # ... more are split into groups of about this size
CHUNK_FLOATS = 36          # 12 columns: 144 B per sample and chunk, a multiple of 16 B (rounds 2 - 4; the baseline of tools/ab_chunk_floats.sh)
# The fused point rollout stages 64 floats per sample and chunk (256 B: one store instruction writes whole 256-byte segments of 4 / 2 / 1
# samples, and the lane -> (sample, vector) map is a shift).  Same-box A/B (tools/ab_chunk_floats.sh, profiles/r05_ab_chunk_floats.txt),
# 36 -> 64 floats: 45 spheres 67.0 -> 60.9 us, grasped box 44.6 -> 41.9, both 98.1 -> 86.3; positions only 36.2 -> 30.5, 25.0 -> 21.2,
# 61.1 -> 46.5 (a row's 144-byte pieces were partial lines for two instructions each).  17.4 KB of LDS per wavefront: still two workgroups per CU.
ROLLOUT_CHUNK_FLOATS = 64
RING_FLOATS = 32           # link kernels with ring staging (RingFlusher in trk_spec_common.h): floats per chunk
@dataclass
class RingPlan:
    """Geometry of RingFlusher<W, V, ALIGNED, IO> (mirrors its constants; see the comment there)."""
    W: int
    V: int
    aligned: bool
    hx: int                 # longest head
    n_full: int
    tail: int               # floats of a sample's tail + head
    pieces: int             # store instructions per chunk

    @property
    def stride(self) -> int:
        return (64 + self.hx) | 1

    def ready_float(self, c: int) -> int:
        """chunk c is complete in every lane's ring once this float has been staged"""
        return self.W - 1 if c >= self.n_full else min(self.W - 1, RING_FLOATS * c + RING_FLOATS - 1 + self.hx)

    def reuse_float(self, c: int) -> int:
        """the first float whose staging overwrites chunk c's half of the ring (W if none does)"""
        return min(self.W, RING_FLOATS * (c + 2))

    def regular(self, f: int) -> bool:
        """the float's ring slot is the same offset from the lane's row pointer for every head"""
        return (f & 63) >= self.hx


def ring_plan(W: int) -> RingPlan:
    assert W >= RING_FLOATS
    g = W % 8
    hx = 0 if g == 0 else (7 if g % 2 else (4 if g == 4 else 6))
    n_full = (W - max(hx, 1)) // RING_FLOATS
    tail = W - RING_FLOATS * n_full
    if tail + (8 - g if g else 0) <= RING_FLOATS:        # the longest tail unit (tail of row s + head of row s + 1) fits a chunk
        return RingPlan(W=W, V=2, aligned=True, hx=hx, n_full=n_full, tail=tail, pieces=16)
    V = 2 if W % 2 == 0 else 1
    n_full = (W - 1) // RING_FLOATS
    return RingPlan(W=W, V=V, aligned=False, hx=0, n_full=n_full, tail=W - RING_FLOATS * n_full, pieces=64 // (64 // (RING_FLOATS // V)))


OBJ_GROUP = int(os.environ.get("TRK_EXP_OBJ_GROUP", "6"))     # points evaluated against the scene together (register arrays of this size)


def _chunked_posbwd_lines(kin: KinModel, point_link, point_offset, snap: float = SNAP, w_expr: Optional[str] = None) -> List[str]:
    """k_posbwd_bi / k_posbwd_bg: explicit reverse mode of point (or link) positions, d sum(gpos . pos) / dq, for columns in walk order.
    FK again, the adjoint rows arrive through a chunk buffer in column order (spec_load_chunk: ROLLOUT_CHUNK_FLOATS floats of every
    sample's row per trip), each adjoint g at point p joins the running wrench (g, p x g); a joint's gradient is the prefix-sum form
    s z . ((Pt1 - Pt0) - t x (Pf1 - Pf0)) over its subtree's range of the walk.  17 KB of LDS per wavefront whatever the row length and
    no per-sample adjoint array in registers: also what the many-link robots' trk_fk_positions_backward runs (their whole-row tile was
    92 KB per workgroup = one wavefront per SIMD: UR10 + Allegro 34.1 us = 0.51 of the roofline)."""
    L, D, P = kin.n_links, kin.n_dofs, len(point_link)
    W = 3 * P
    Wx = w_expr or "W"
    V = 4 if W % 4 == 0 else (2 if W % 2 == 0 else 1)
    pl = [int(v) for v in point_link]
    po = np.asarray(point_offset, np.float32).reshape(-1, 3)
    pos_of = {int(kin.order[p]): p for p in range(L)}
    cols_of_link: Dict[int, List[int]] = {i: [c for c in range(P) if pl[c] == i] for i in range(L)}
    masked = _masked_factory(kin)
    BNF = int(os.environ.get("TRK_EXP_BWD_CHUNK_FLOATS", str(ROLLOUT_CHUNK_FLOATS)))  # 36 -> 64 measured 43.7 -> 42.8, 21.3 -> 19.9, 66.7 -> 47.5 us
    BLS = BNF if BNF % 32 else BNF + 4
    out: List[str] = []
    # ---- explicit reverse mode of the point positions (trk_fk_points_backward): FK again, the adjoint rows arrive through
    # the chunk buffer in column order, each adjoint g at point p joins the running wrench (g, p x g); prefix-sum gradients
    for base_identity in (True, False):
        E = Emitter()
        kname = "k_posbwd_bi" if base_identity else "k_posbwd_bg"
        E.raw(f"__global__ void __launch_bounds__(SPEC_BLOCK, 2) {kname}(SpecArgs A) {{")
        E.raw(f"    __shared__ __attribute__((aligned(16))) float lds_all[SPEC_BLOCK * {max(BLS, D)}];")
        E.raw("    const int lane = __builtin_amdgcn_workitem_id_x() & (TRK_WAVE - 1);")
        E.raw("    const int wave = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_workitem_id_x() / TRK_WAVE);")
        E.raw(f"    float* lds = lds_all + wave * (TRK_WAVE * {max(BLS, D)});")
        E.raw("    const int64_t wblock = (int64_t)__builtin_amdgcn_workgroup_id_x() * SPEC_WAVES + wave;")
        E.raw("    const int64_t base = wblock * TRK_WAVE;")
        E.raw("    const int rows = (int)max((int64_t)0, min((int64_t)TRK_WAVE, A.n - base));")
        E.raw("    const float* gpos = static_cast<const float*>(A.link_pos);")
        E.raw("    float q[D];")
        E.raw("    spec_load_q<D>(static_cast<const float*>(A.q), base, rows, lane, lds, q);")
        E.raw(f"    const float* row = lds + lane * {BLS};")
        R = {}; t = {}; passv = {}
        if base_identity:
            R[0] = [[ONE if r == c else ZERO for c in range(3)] for r in range(3)]
            t[0] = [ZERO, ZERO, ZERO]
        else:
            R[0] = [[S(1.0, f"A.base_R[{3 * r + c}]") for c in range(3)] for r in range(3)]
            t[0] = [S(1.0, f"A.base_t[{r}]") for r in range(3)]
        _emit_angles(E, kin)
        E.raw("    float pf0 = 0.0f, pf1 = 0.0f, pf2 = 0.0f, pt0 = 0.0f, pt1 = 0.0f, pt2 = 0.0f;")
        PF = [S(1.0, f"pf{k}") for k in range(3)]
        PT = [S(1.0, f"pt{k}") for k in range(3)]
        snap_c = {}; gq_expr = {}
    
        def functional(i: int) -> S:
            jt = int(kin.joint_type[i])
            if jt == JOINT_PRISMATIC:
                par = int(kin.parent[i])
                dirw = [E.lincomb([(R[par][r][k], S(float(kin.axis[i][k]))) for k in range(3)]) for r in range(3)]
                return E.dot(dirw, PF)
            ax = int(kin.rot_axis[i])
            z = [R[i][r][ax] for r in range(3)]
            cr = E.cross(t[i], PF)
            return E.dot(z, [E.lincomb([(PT[k], ONE), (cr[k], S(-1.0))]) for k in range(3)])
    
        chunk_start = -1
        for p in range(L):
            i = int(kin.order[p])
            if p > 0:
                _emit_fk_link(E, kin, i, R, t, passv, snap)
                if int(kin.joint_type[i]) != JOINT_FIXED:
                    snap_c[i] = S(1.0, E.tmp(E.expr(functional(i))))
            for c in cols_of_link[i]:
                f0 = 3 * c
                # the three floats of a column may straddle two chunks: fetch component by component
                comp = []
                for k in range(3):
                    f = f0 + k
                    cs = (f // BNF) * BNF
                    if cs != chunk_start:
                        nf = min(BNF, W - cs)
                        E.raw(f"    spec_load_chunk<{Wx}, {nf}, {BLS}, {V}>(gpos, base, {cs}, rows, lane, lds);")
                        chunk_start = cs
                    comp.append(E.tmp(f"row[{f - cs}]"))
                if p == 0:
                    continue                        # the root does not move with q
                off = [S(snap_const(po[c][k], 0.0)) for k in range(3)]
                pc = t[i] if all(o.is_zero for o in off) else \
                    [E.named(E.lincomb([(R[i][r][k], off[k]) for k in range(3)], t[i][r])) for r in range(3)]
                px, py, pz = (E.expr(v) for v in pc)
                E.raw(f"    pf0 += {comp[0]}; pf1 += {comp[1]}; pf2 += {comp[2]};")
                E.raw(f"    pt0 += {py} * {comp[2]} - {pz} * {comp[1]}; pt1 += {pz} * {comp[0]} - {px} * {comp[2]}; "
                      f"pt2 += {px} * {comp[1]} - {py} * {comp[0]};")
            for j in [j for j in range(1, L) if int(kin.joint_type[j]) != JOINT_FIXED]:
                if int(kin.subtree_end[pos_of[j]]) == p + 1:
                    d = int(kin.dof_idx[j]); jt = int(kin.joint_type[j])
                    sg = 1.0 if jt == JOINT_PRISMATIC else float(kin.rot_sign[j])
                    if sg == 0.0:
                        gq_expr[d] = ZERO
                    else:
                        g = E.lincomb([(functional(j), ONE), (snap_c[j], S(-1.0))])
                        gq_expr[d] = masked(E, j, d, S(g.c * sg, g.n))
        E.raw(f"    const float gv[D] = {{{', '.join(E.expr(gq_expr.get(d, ZERO)) for d in range(D))}}};")
        E.raw("    spec_store_gq<D>(static_cast<float*>(A.gq), base, rows, lane, lds, gv);")
        E.raw("}")
        out.extend(E.lines)
        out.append("")
    return out


def generate_points_rollout_source(kin: KinModel, pt: PointsTemplate, ident: str, snap: float = SNAP,
                                   link_mode: bool = False, meta: Optional[dict] = None) -> str:
    """Fused FK + objectives + gradient with the collision fields on attached points.  Differences to the link kernel:

    * each link's points are produced, scored against the scene and folded into ONE running wrench (f, p x f about the
      world origin) as soon as the link's pose exists -- no position or adjoint tile in LDS, no per-link accumulators;
    * reverse mode is the prefix-sum form of the transposed geometric Jacobian (the table-driven kernels' idea, in
      registers): in walk order a joint's subtree is a contiguous range, so its gradient is
      s z_j . ((Pt1 - Pt0) - t_j x (Pf1 - Pf0)) with the running wrench sampled at the range's start and end;
      a force that lands on an EARLIER link (the far side of a self-collision pair) is applied directly to that link's
      ancestor joints, s z_j . ((p - t_j) x f), instead of entering the running sums;
    * positions leave through one 64-float chunk buffer per wavefront (spec_flush_chunk; ROLLOUT_CHUNK_FLOATS).

    The kernel needs the full 256-VGPR budget (2 wavefronts per SIMD).  A two-sweep variant for serial chains (root->tip
    for the positions, then tip->root stepping the pose back through the inverse joint transforms with a single suffix
    wrench; 168 VGPRs, 3 wavefronts per SIMD) was built and measured: 79 vs 76 us for the 45-sphere Panda -- the kernel
    is bound by its ~6900 VALU instructions per wavefront (SQ_INSTS_VALU), so the recomputation ate what the occupancy
    gave, and it was dropped.

    link_mode: the columns are exactly the links in file order with zero offsets and the unit registers as an ordinary
    link kernel (n_points = 0).  For TREES this pipeline beats generate_rollout_source: a finger's or an arm's joints are
    finished (and their axes / origins die) when their subtree ends, instead of living until a reverse pass."""
    if link_mode:
        assert [int(v) for v in pt.point_link] == list(range(kin.n_links)) and not np.asarray(pt.point_offset).any()
    L, D, P = kin.n_links, kin.n_dofs, len(pt.point_link)
    W = 3 * P
    V = 4 if W % 4 == 0 else (2 if W % 2 == 0 else 1)
    pl = [int(v) for v in pt.point_link]
    po = np.asarray(pt.point_offset, np.float32).reshape(-1, 3)
    pos_of = {int(kin.order[p]): p for p in range(L)}
    walk_rank = [pos_of[i] for i in pl]
    if any(walk_rank[k] > walk_rank[k + 1] for k in range(P - 1)):
        raise ValueError("generate_points_rollout_source: columns must follow the walk order of their links")
    cols_of_link: Dict[int, List[int]] = {i: [c for c in range(P) if pl[c] == i] for i in range(L)}
    obj_rank = {c: k for k, c in enumerate(pt.obj_cols)}
    if sorted(pt.obj_cols) != list(pt.obj_cols):
        raise ValueError("generate_points_rollout_source: obj_cols must be increasing (margins are read in groups)")
    # a pair is scored when the walk reaches its later column; pairs_at[link] = [(pair index, late col, early col, late_is_a)]
    pairs_at: Dict[int, List[Tuple[int, int, int, bool]]] = {i: [] for i in range(L)}
    for pi, (a, b) in enumerate(pt.self_pairs):
        late_is_a = (walk_rank[a], a) >= (walk_rank[b], b)
        late, early = (a, b) if late_is_a else (b, a)
        pairs_at[pl[late]].append((pi, late, early, late_is_a))
    masked = _masked_factory(kin)
    # the rollout's chunk: NF floats of every sample's row leave together; LS = the per-lane stride of the staging buffer (a multiple of
    # 4 floats for the 16-byte reads, and NOT a multiple of 32: the per-lane row writes would all hit one bank)
    NF = int(os.environ.get("TRK_EXP_CHUNK_FLOATS", str(ROLLOUT_CHUNK_FLOATS)))
    LS = NF if NF % 32 else NF + 4
    WIDE = (not link_mode) and os.environ.get("TRK_EXP_CHUNK_WIDE", "0") != "0"
    lds_per_lane = max(LS, D)
    joint_links = [i for i in range(1, L) if int(kin.joint_type[i]) != JOINT_FIXED]
    ancestors: Dict[int, List[int]] = {}
    for i in range(L):
        chain, a = [], i
        while a > 0:
            if int(kin.joint_type[a]) != JOINT_FIXED:
                chain.append(a)
            a = int(kin.parent[a])
        ancestors[i] = chain                   # links whose joints move link i (incl. i itself)

    out: List[str] = []
    out.append(f"// GENERATED by torch_robotics_amd/codegen.py for model '{kin.name}' ({L} links, {D} DOF) with {P} attached points -- do not edit.")
    out.append('#include "trk_spec_common.h"')
    out.append(f"namespace spec_{ident} {{")
    out.append(f"constexpr int L = {L}, D = {D}, P = {P}, W = {W};")
    for base_identity in (True, False):
        E = Emitter()
        kname = "k_rollout_bi" if base_identity else "k_rollout_bg"
        E.raw("template <bool FAST, class IO>   // FAST: scene_is_fast(A.C) -- only the few-equal-spheres scene path is compiled in")
        E.raw(f"__global__ void __launch_bounds__(SPEC_BLOCK, 2) {kname}(SpecArgs A) {{")
        E.raw(f"    __shared__ __attribute__((aligned(16))) float lds_all[SPEC_BLOCK * {lds_per_lane} + SPEC_WAVES * TRK_LDS_SPHERES * 4];")
        E.raw("    const int lane = __builtin_amdgcn_workitem_id_x() & (TRK_WAVE - 1);")
        E.raw("    const int wave = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_workitem_id_x() / TRK_WAVE);   // wave-uniform -> SGPR")
        E.raw(f"    float* lds = lds_all + wave * (TRK_WAVE * {lds_per_lane});")
        E.raw(f"    float4* lds_sph = reinterpret_cast<float4*>(lds_all + SPEC_BLOCK * {lds_per_lane}) + wave * TRK_LDS_SPHERES;")
        E.raw("    const SpheresInFlight sph = spec_load_spheres_issue(A.C, lane);   // waited for together with the rows below")
        E.raw("    const int64_t wblock = (int64_t)__builtin_amdgcn_workgroup_id_x() * SPEC_WAVES + wave;")
        E.raw("    const int64_t base = wblock * TRK_WAVE;")
        E.raw("    const int rows = (int)max((int64_t)0, min((int64_t)TRK_WAVE, A.n - base));")
        E.raw("    IO* pos_out = static_cast<IO*>(A.link_pos);")
        E.raw("    float q[D];")
        E.raw("    spec_load_q<D>(static_cast<const IO*>(A.q), base, rows, lane, lds, q);")
        E.raw("    spec_load_spheres_finish(lds_sph, lane, sph);")
        E.raw(f"    float* row = lds + lane * {LS};          // this lane's slice of the chunk buffer")
        E.raw("    NoTick notick;")
        R: Dict[int, List[List[S]]] = {}
        t: Dict[int, List[S]] = {}
        passv: Dict[int, S] = {}
        if base_identity:
            R[0] = [[ONE if r == c else ZERO for c in range(3)] for r in range(3)]
            t[0] = [ZERO, ZERO, ZERO]
        else:
            R[0] = [[S(1.0, f"A.base_R[{3 * r + c}]") for c in range(3)] for r in range(3)]
            t[0] = [S(1.0, f"A.base_t[{r}]") for r in range(3)]
        _emit_angles(E, kin)
        E.raw("    float cost = 0.0f;")
        E.raw("    float pf0 = 0.0f, pf1 = 0.0f, pf2 = 0.0f, pt0 = 0.0f, pt1 = 0.0f, pt2 = 0.0f;   // running wrench of the links visited so far")
        for i in joint_links:
            E.raw(f"    float late{int(kin.dof_idx[i])} = 0.0f;")
        PF = [S(1.0, f"pf{k}") for k in range(3)]
        PT = [S(1.0, f"pt{k}") for k in range(3)]
        colpos: Dict[int, List[S]] = {}
        snap_c: Dict[int, S] = {}
        gq_expr: Dict[int, S] = {}
        chunk_start = 0                       # first float of the chunk being filled

        def joint_functional(i: int) -> S:
            """z_i . (Pt - t_i x Pf) with the CURRENT running wrench (prismatic: (R_parent axis) . Pf)"""
            jt = int(kin.joint_type[i])
            if jt == JOINT_PRISMATIC:
                par = int(kin.parent[i])
                dirw = [E.lincomb([(R[par][r][k], S(float(kin.axis[i][k]))) for k in range(3)]) for r in range(3)]
                return E.dot(dirw, PF)
            ax = int(kin.rot_axis[i])
            z = [R[i][r][ax] for r in range(3)]
            cr = E.cross(t[i], PF)
            return E.dot(z, [E.lincomb([(PT[k], ONE), (cr[k], S(-1.0))]) for k in range(3)])

        def in_order(p: List[S], g: List[str]):
            """force g (C expressions) at world point p on the link the walk stands on -> running wrench"""
            px, py, pz = (E.expr(v) for v in p)
            E.raw(f"    pf0 += {g[0]}; pf1 += {g[1]}; pf2 += {g[2]};")
            E.raw(f"    pt0 += {py} * {g[2]} - {pz} * {g[1]}; pt1 += {pz} * {g[0]} - {px} * {g[2]}; pt2 += {px} * {g[1]} - {py} * {g[0]};")

        def late_force(link: int, p: List[S], g: List[str]):
            """force g at world point p on an EARLIER link: straight onto the joints that move that link"""
            gS = [S(1.0, x) for x in g]
            for j in ancestors[link]:
                d = int(kin.dof_idx[j]); jt = int(kin.joint_type[j])
                if jt == JOINT_PRISMATIC:
                    par = int(kin.parent[j])
                    dirw = [E.lincomb([(R[par][r][k], S(float(kin.axis[j][k]))) for k in range(3)]) for r in range(3)]
                    val = E.dot(dirw, gS)
                else:
                    if float(kin.rot_sign[j]) == 0.0:
                        continue
                    ax = int(kin.rot_axis[j])
                    z = [R[j][r][ax] for r in range(3)]
                    arm = [E.lincomb([(p[k], ONE), (t[j][k], S(-1.0))]) for k in range(3)]
                    val = E.dot(z, E.cross(arm, gS))
                if not val.is_zero:
                    E.raw(f"    late{d} += {E.expr(val)};")

        gfin: Dict[int, S] = {}

        def finish_joint(i: int):
            """subtree of joint i complete: its share of the running wrench is final.  `late` forces may still arrive
            (a pair whose later column sits in ANOTHER branch of a tree), so they are added at the very end."""
            d = int(kin.dof_idx[i]); jt = int(kin.joint_type[i])
            sg = 1.0 if jt == JOINT_PRISMATIC else float(kin.rot_sign[i])
            if sg == 0.0:
                return
            end = joint_functional(i)
            gfin[i] = S(1.0, E.tmp(E.expr(E.named(E.lincomb([(end, ONE), (snap_c[i], S(-1.0))])))))

        for p in range(L):
            i = int(kin.order[p])
            if p > 0:
                _emit_fk_link(E, kin, i, R, t, passv, snap)
                if int(kin.joint_type[i]) != JOINT_FIXED:
                    # running wrench BEFORE this link's subtree: a COPY (pf/pt are mutable; the expression may be a bare alias)
                    snap_c[i] = S(1.0, E.tmp(E.expr(joint_functional(i))))
            cols = cols_of_link[i]
            # ---- positions of this link's columns, staged for output
            for c in cols:
                off = [S(snap_const(po[c][k], 0.0)) for k in range(3)]
                if all(o.is_zero for o in off):
                    colpos[c] = t[i]
                else:
                    colpos[c] = [E.named(E.lincomb([(R[i][r][k], off[k]) for k in range(3)], t[i][r])) for r in range(3)]
                for k in range(3):
                    f = 3 * c + k
                    E.raw(f"    row[{f - chunk_start}] = {E.expr(colpos[c][k])};")
                    if f + 1 - chunk_start == NF or f + 1 == W:
                        nf = f + 1 - chunk_start
                        # vector width of this chunk's stores.  The LDS side is always aligned (LS and the chunk start are multiples of
                        # 4 floats); the HBM side is aligned only when the row length is a multiple of the vector too -- this chip takes
                        # the 4- / 8-byte-aligned wide stores of the other row lengths as they are (half / a quarter of the instructions)
                        Vc = V if not WIDE else (4 if nf % 4 == 0 else (2 if nf % 2 == 0 else 1))
                        E.raw(f"    if (pos_out) spec_flush_chunk<W, {nf}, {LS}, {Vc}, IO, {'true' if Vc > V else 'false'}>(pos_out, base, {chunk_start}, rows, lane, lds);")
                        chunk_start = f + 1
            # ---- objects / workspace box on this link's collision columns, a few at a time
            ocols = [c for c in cols if c in obj_rank]
            for g0 in range(0, len(ocols), OBJ_GROUP):
                grp = ocols[g0:g0 + OBJ_GROUP]
                n = len(grp)
                mb = obj_rank[grp[0]]
                assert [obj_rank[c] for c in grp] == list(range(mb, mb + n))
                E.raw("    {")
                for k, nm in enumerate("xyz"):
                    E.raw(f"        const float p{nm}[{n}] = {{{', '.join(E.expr(colpos[c][k]) for c in grp)}}};")
                E.raw(f"        float gx[{n}], gy[{n}], gz[{n}];")
                E.raw("#pragma unroll")
                E.raw(f"        for (int l = 0; l < {n}; ++l) {{ gx[l] = 0.0f; gy[l] = 0.0f; gz[l] = 0.0f; }}")
                E.raw(f"        if (A.w.w_obj != 0.0f) cost += spec_objects_cost<{n}, NoTick, FAST>(A.C, A.w.w_obj, px, py, pz, gx, gy, gz, notick, lds_sph, {mb});")
                E.raw(f"        if (A.w.w_ws != 0.0f && A.C.has_ws) cost += spec_ws_cost<{n}>(A.C, A.w.w_ws, px, py, pz, gx, gy, gz, {mb});")
                E.raw("#pragma unroll")
                E.raw(f"        for (int l = 0; l < {n}; ++l) {{")
                E.raw("            pf0 += gx[l]; pf1 += gy[l]; pf2 += gz[l];")
                E.raw("            pt0 += py[l] * gz[l] - pz[l] * gy[l]; pt1 += pz[l] * gx[l] - px[l] * gz[l]; pt2 += px[l] * gy[l] - py[l] * gx[l];")
                E.raw("        }")
                E.raw("    }")
            # ---- self-collision pairs whose later column belongs to this link.  Round 5: the force on a point is ACCUMULATED over its
            # pairs (a pair hands back s = w / ||d|| and d = p_late - p_early: one FMA per component and side) and enters the running
            # wrench ONCE per point (9 instructions) instead of once per pair and side.  The grasped-box model has 66 pairs on 14 + 4
            # points: 2061 of the kernel's 5532 static vector instructions were this phase (tools/isa_valu_count.sh).  (First built as
            # S = sum s, V = sum s p_e, force = V - p S: a few instructions fewer, but it cancels when the points are close and far from
            # the origin -- spec_self_pair_sd's comment.)
            if pairs_at[i]:
                E.raw("    if (A.w.w_self != 0.0f) {")
                E.raw("        const bool sclamp = (A.C.clamp_fields & TRK_FIELD_SELF) != 0;")
                earlies = sorted({early for _, _, early, _ in pairs_at[i]})
                lates = sorted({late for _, late, _, _ in pairs_at[i]})
                for e in earlies:             # all pairs of one earlier column push on the same point
                    E.raw(f"        float ge{e}_0 = 0.0f, ge{e}_1 = 0.0f, ge{e}_2 = 0.0f;")
                for c in lates:
                    E.raw("        {")
                    E.raw("        float gl0 = 0.0f, gl1 = 0.0f, gl2 = 0.0f;")
                    pc = [E.expr(colpos[c][k]) for k in range(3)]
                    for pi, late, early, late_is_a in pairs_at[i]:
                        if late != c:
                            continue
                        pe = [E.expr(colpos[early][k]) for k in range(3)]
                        E.raw(f"        {{ float d0_, d1_, d2_; const float s_ = spec_self_pair_sd(A.w.w_self, cptr(A.C.self_margin)[{pi}], {', '.join(pc)}, {', '.join(pe)}, sclamp, cost, d0_, d1_, d2_);")
                        E.raw(f"          gl0 = fmaf(-s_, d0_, gl0); gl1 = fmaf(-s_, d1_, gl1); gl2 = fmaf(-s_, d2_, gl2); "
                              f"ge{early}_0 = fmaf(s_, d0_, ge{early}_0); ge{early}_1 = fmaf(s_, d1_, ge{early}_1); ge{early}_2 = fmaf(s_, d2_, ge{early}_2); }}")
                    in_order(colpos[c], ["gl0", "gl1", "gl2"])
                    E.raw("        }")
                for e in earlies:
                    g = [f"ge{e}_0", f"ge{e}_1", f"ge{e}_2"]
                    if pl[e] == i:
                        in_order(colpos[e], g)
                    else:
                        late_force(pl[e], colpos[e], g)
                E.raw("    }")
            # ---- end-effector tracking when the walk stands on the EE link
            if i in (pt.ee_link, pt.ee2_link) and i >= 0:
                tgt_name = "A.C.ee_target" if i == pt.ee_link else "A.C.ee2_target"
                E.raw("    if (A.w.w_ee != 0.0f) {")
                E.raw(f"        const float eR[9] = {{{', '.join(E.expr(R[i][r][c]) for r in range(3) for c in range(3))}}};")
                E.raw(f"        const float et[3] = {{{', '.join(E.expr(t[i][k]) for k in range(3))}}};")
                E.raw("        float gR[9], gt[3];")
                E.raw(f"        const float ce = ee_cost_eval(eR, et, {tgt_name}, A.C.ee_w_pos, A.C.ee_w_rot, A.C.ee_square, gR, gt);")
                E.raw("        cost = fmaf(A.w.w_ee, ce, cost);")
                E.raw("        gt[0] *= A.w.w_ee; gt[1] *= A.w.w_ee; gt[2] *= A.w.w_ee;")
                E.raw("        pf0 += gt[0]; pf1 += gt[1]; pf2 += gt[2];")
                # torque = t x gt + axial(Rbar R^T), Rbar = w gR
                E.raw("        float M[9];")
                E.raw("#pragma unroll")
                E.raw("        for (int a = 0; a < 3; ++a)")
                E.raw("#pragma unroll")
                E.raw("            for (int b = 0; b < 3; ++b) M[3 * a + b] = A.w.w_ee * (gR[3 * a] * eR[3 * b] + gR[3 * a + 1] * eR[3 * b + 1] + gR[3 * a + 2] * eR[3 * b + 2]);")
                E.raw("        pt0 += et[1] * gt[2] - et[2] * gt[1] + (M[7] - M[5]);")
                E.raw("        pt1 += et[2] * gt[0] - et[0] * gt[2] + (M[2] - M[6]);")
                E.raw("        pt2 += et[0] * gt[1] - et[1] * gt[0] + (M[3] - M[1]);")
                E.raw("    }")
            # ---- joints whose subtree ends after this link
            for j in joint_links:
                if int(kin.subtree_end[pos_of[j]]) == p + 1:
                    finish_joint(j)
        assert chunk_start == W
        for i in joint_links:
            d = int(kin.dof_idx[i]); jt = int(kin.joint_type[i])
            sg = 1.0 if jt == JOINT_PRISMATIC else float(kin.rot_sign[i])
            if i not in gfin:
                gq_expr[d] = ZERO
                continue
            g = E.lincomb([(gfin[i], ONE), (S(1.0, f"late{d}"), ONE)])
            gq_expr[d] = masked(E, i, d, S(g.c * sg, g.n))
        E.raw("    if (!A.gq) return;                       // positions only (trk_fk_points): weights are zero, nothing else to write")
        E.raw("    if (lane < rows) store_wt_f1(A.cost + base + lane, cost);")
        E.raw("    if (A.cost_sum) {")
        E.raw("        const float tot = spec_wave_sum(lane < rows ? cost : 0.0f);")
        E.raw("        if (lane == 0 && rows > 0) store_wt_f1(A.cost_sum + wblock, tot);")
        E.raw("    }")
        E.raw(f"    const float gv[D] = {{{', '.join(E.expr(gq_expr.get(d, ZERO)) for d in range(D))}}};")
        E.raw("    spec_store_gq<D>(static_cast<IO*>(A.gq), base, rows, lane, lds, gv);")
        E.raw("}")
        out.extend(E.lines)
        out.append("")

    out.extend(_chunked_posbwd_lines(kin, pl, po, snap))

    if meta is not None and not link_mode:
        # what a code-object (hipRTC) build of this unit must contain: the name expressions of its kernels (jit.py, trk_spec_register_module)
        meta["kernels"] = [f"spec_{ident}::k_rollout_{b}<{f}, float>" for b in ("bi", "bg") for f in ("true", "false")] + \
                          [f"spec_{ident}::k_posbwd_{b}" for b in ("bi", "bg")]
    out.extend(_points_entry_lines(kin, pt, ident, link_mode))
    return "\n".join(out) + "\n"


def _points_entry_lines(kin: KinModel, pt: PointsTemplate, ident: str, link_mode: bool = False) -> List[str]:
    out: List[str] = []
    obj = ", ".join(str(c) for c in pt.obj_cols) or "0"
    pairs = ", ".join(f"{a}, {b}" for a, b in pt.self_pairs) or "0"
    out.append("#ifndef __HIPCC_RTC__          // the unit's host half: launchers and its registry entry")
    out.append(f"static const int32_t kObjCols[] = {{{obj}}};")
    out.append(f"static const int32_t kSelfPairs[] = {{{pairs}}};")
    out.append("template <class IO>")
    out.append("static void launch_io(const SpecArgs& a, int base_identity, hipStream_t st) {")
    out.append("    const unsigned grid = (unsigned)((a.n + SPEC_BLOCK - 1) / SPEC_BLOCK);")
    out.append("    const bool fast = scene_is_fast(a.C) || a.w.w_obj == 0.0f;")
    out.append("    if (fast) {")
    out.append("        if (base_identity) hipLaunchKernelGGL((k_rollout_bi<true, IO>), dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
    out.append("        else hipLaunchKernelGGL((k_rollout_bg<true, IO>), dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
    out.append("    } else {")
    out.append("        if (base_identity) hipLaunchKernelGGL((k_rollout_bi<false, IO>), dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
    out.append("        else hipLaunchKernelGGL((k_rollout_bg<false, IO>), dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
    out.append("    }")
    out.append("}")
    out.append("static void launch(const SpecEntry*, const SpecArgs& a, int base_identity, hipStream_t st) {")
    if link_mode:
        out.append("    if (a.io_f16) launch_io<_Float16>(a, base_identity, st); else launch_io<float>(a, base_identity, st);")
    else:
        out.append("    launch_io<float>(a, base_identity, st);")
    out.append("}")
    out.append("static void launch_posbwd(const SpecEntry*, const SpecArgs& a, int base_identity, hipStream_t st) {")
    out.append("    const unsigned grid = (unsigned)((a.n + SPEC_BLOCK - 1) / SPEC_BLOCK);")
    out.append("    if (base_identity) hipLaunchKernelGGL(k_posbwd_bi, dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
    out.append("    else hipLaunchKernelGGL(k_posbwd_bg, dim3(grid), dim3(SPEC_BLOCK), 0, st, a);")
    out.append("}")
    n_points = 0 if link_mode else len(pt.point_link)
    phash = 0 if link_mode else points_hash(pt.point_link, pt.point_offset)
    out.append(f"static const SpecEntry kEntry = {{SPEC_ENTRY_STAMP, 0x{model_hash(kin):016x}ull, {kin.n_links}, {kin.n_dofs}, {len(pt.obj_cols)}, kObjCols, "
               f"{len(pt.self_pairs)}, kSelfPairs, {pt.ee_link}, \"{ident}\", launch, {n_points}, "
               f"0x{phash:016x}ull, launch_posbwd, {pt.ee2_link}, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, "
               f"0, nullptr, nullptr}};")
    out.append("static struct Reg { Reg() { trk_spec_register(&kEntry); } } reg;")
    out.append("#endif      // !__HIPCC_RTC__")
    out.append(f"}}  // namespace spec_{ident}")
    return out


def link_points_template(kin: KinModel, tmpl: CollisionTemplate) -> PointsTemplate:
    """a CollisionTemplate expressed as a point set: every link origin, file order (requires file order == walk order)"""
    if [int(v) for v in kin.order] != list(range(kin.n_links)):
        raise ValueError("link_points_template: the URDF's link order is not a pre-order walk of the tree")
    L = kin.n_links
    return PointsTemplate(point_link=np.arange(L, dtype=np.int32), point_offset=np.zeros((L, 3), np.float32),
                          obj_cols=[int(i) for i in tmpl.obj_links], self_pairs=[(int(a), int(b)) for a, b in tmpl.self_pairs],
                          ee_link=int(tmpl.ee_link), ee2_link=int(tmpl.ee2_link))


def _panda_pairs(idx) -> List[Tuple[int, int]]:
    pairs_by_name = {"panda_link4": ["panda_link1"], "panda_link5": ["panda_link0", "panda_link1", "panda_link2"],
                     "panda_link6": ["panda_link0", "panda_link1", "panda_link2"],
                     "panda_hand": ["panda_link0", "panda_link1", "panda_link2"]}
    names = sorted(set(list(pairs_by_name) + [v for vs in pairs_by_name.values() for v in vs] +
                       ["panda_link0", "panda_link1", "panda_link2", "panda_link3"]))
    return [(idx[a], idx[b]) for a in names for b in pairs_by_name.get(a, [])], names


def panda_spheres_points(kin: KinModel) -> PointsTemplate:
    """RobotPanda(link_sphere_model="panda"): link origins + the 45 link spheres, link-sorted columns; objects and the
    workspace box on the spheres, self-collision pairs on the link origins (robots.py)."""
    from .costmodel import link_sorted_point_set, load_link_spheres
    from .kinematics import DATA_DIR
    sl, so, _sr, _names = load_link_spheres(DATA_DIR / "configs" / "panda_sphere_config.yaml", kin.name_to_idx)
    pl, po, origin_col, sphere_col = link_sorted_point_set(kin.order, sl, so)
    pairs, _ = _panda_pairs(kin.name_to_idx)
    return PointsTemplate(point_link=pl, point_offset=po, obj_cols=[int(c) for c in sphere_col],
                          self_pairs=[(int(origin_col[a]), int(origin_col[b])) for a, b in pairs],
                          ee_link=kin.name_to_idx["ee_link"])


def panda_grasp_points(kin: KinModel) -> PointsTemplate:
    """RobotPanda(grasped_object=GraspedObjectPandaBox()): 12 link origins + 14 box points on `grasped_object`;
    objects / workspace on the 5 collision links + the 14 points; RobotBase's pair table incl. grasped rows."""
    from .costmodel import panda_box_base_points
    idx = kin.name_to_idx
    L = kin.n_links
    pts = panda_box_base_points()
    G = len(pts)
    pl = np.concatenate([np.arange(L), np.full(G, idx["grasped_object"])]).astype(np.int32)
    po = np.concatenate([np.zeros((L, 3), np.float32), pts])
    obj = [idx[n] for n in ("panda_link2", "panda_link3", "panda_link5", "panda_link7", "panda_hand")] + list(range(L, L + G))
    pairs, names = _panda_pairs(idx)
    pairs = list(pairs)
    for n in ("panda_link0", "panda_link1", "panda_link2", "panda_link3"):       # robot_base.py:120-130
        pairs += [(L + m, idx[n]) for m in range(G)]
    return PointsTemplate(point_link=pl, point_offset=po, obj_cols=obj, self_pairs=pairs, ee_link=idx["ee_link"])


def ur10_allegro_template(kin: KinModel) -> CollisionTemplate:
    """BASELINE config 4 (UR10 + Allegro hand): arm links, palm and the four fingertips against the scene;
    fingertip-vs-fingertip / fingertip-vs-forearm self pairs; the arm's `ee_link` is tracked."""
    idx = kin.name_to_idx
    obj = [idx[n] for n in ("upper_arm_link", "forearm_link", "wrist_1_link", "wrist_3_link", "allegro_palm_link",
                            "allegro_index_biotac_tip", "allegro_middle_biotac_tip", "allegro_ring_biotac_tip",
                            "allegro_thumb_biotac_tip")]
    tips = ["allegro_index_biotac_tip", "allegro_middle_biotac_tip", "allegro_ring_biotac_tip", "allegro_thumb_biotac_tip"]
    pairs = [(idx[a], idx[b]) for i, a in enumerate(tips) for b in tips[i + 1:]]
    pairs += [(idx[t], idx["forearm_link"]) for t in tips]
    return CollisionTemplate(obj_links=obj, self_pairs=pairs, ee_link=idx["ee_link"])


def dual_panda_template(kin: KinModel) -> CollisionTemplate:
    """BASELINE config 5 (two Panda arms): RobotPanda's object-collision links on both arms, arm-vs-arm self pairs;
    both end effectors are tracked (ee_link = left, ee2_link = right; a cost model that tracks only one arm does not
    match this unit and takes the table-driven kernel)."""
    idx = kin.name_to_idx
    names = ("panda_link2", "panda_link3", "panda_link5", "panda_link7", "panda_hand")
    obj = [idx[f"{side}_{n}"] for side in ("left", "right") for n in names]
    cross = [("panda_hand", "panda_hand"), ("panda_hand", "panda_link5"), ("panda_link5", "panda_hand"),
             ("panda_link5", "panda_link5"), ("panda_link7", "panda_link7"), ("panda_link3", "panda_link3"),
             ("panda_hand", "panda_link3"), ("panda_link3", "panda_hand")]
    pairs = [(idx[f"left_{a}"], idx[f"right_{b}"]) for a, b in cross]
    return CollisionTemplate(obj_links=obj, self_pairs=pairs, ee_link=idx["left_ee_link"], ee2_link=idx["right_ee_link"])


def default_template(kin: KinModel) -> CollisionTemplate:
    """The collision template of a bundled URDF that has no Robot class of its own (the reference ships RobotPanda only): object /
    workspace collision on up to five leaf links plus the links at one half and one third of the file order, the last leaf
    tracked as the end effector, no self pairs.  What `build()` compiles ahead of time for those robots, so that their FK family
    (matrices, positions, Jacobian, reverse modes) and a fused rollout on this template never take the table-driven kernels;
    any other template is one `jit.specialize` away."""
    par = np.asarray(kin.parent, np.int64)
    leaves = [i for i in range(kin.n_links) if not (par == i).any()]
    obj = sorted(set(leaves[:5] + [kin.n_links // 2, kin.n_links // 3]))
    return CollisionTemplate(obj_links=obj, self_pairs=[], ee_link=leaves[-1])


# robots that get a specialised kernel at build time: name -> (urdf file, template factory).  Every URDF under data/urdf/ is
# here: the benchmark robots with their own collision models, the others with `default_template`.
SPEC_ROBOTS = {
    "panda": ("panda_arm_no_gripper.urdf", panda_template),
    "ur10_allegro": ("ur10_allegro.urdf", ur10_allegro_template),
    "dual_panda": ("dual_panda.urdf", dual_panda_template),
    "iiwa7": ("iiwa7.urdf", default_template),
    "ur10": ("ur10.urdf", default_template),
    "allegro_hand": ("allegro_hand.urdf", default_template),
    "panda_arm_hand": ("panda_arm_hand.urdf", default_template),
    "iiwa7_allegro": ("iiwa7_allegro.urdf", default_template),
    "shadow_hand": ("shadow_hand.urdf", default_template),
    "tiago": ("tiago_dual_holobase_minimal_holonomic.urdf", default_template),
    "hab_stretch": ("hab_stretch.urdf", default_template),
}


def panda_spheres_grasp_points(kin: KinModel) -> PointsTemplate:
    """RobotPanda(link_sphere_model="panda", grasped_object=GraspedObjectPandaBox()): link-sorted origins + spheres, then
    the 14 box points; objects / workspace on spheres + box points; pair table incl. grasped rows (on origin columns)."""
    from .costmodel import link_sorted_point_set, load_link_spheres, panda_box_base_points
    from .kinematics import DATA_DIR
    idx = kin.name_to_idx
    sl, so, _sr, _names = load_link_spheres(DATA_DIR / "configs" / "panda_sphere_config.yaml", idx)
    pl, po, origin_col, sphere_col = link_sorted_point_set(kin.order, sl, so)
    pts = panda_box_base_points()
    G, P0 = len(pts), len(pl)
    pl = np.concatenate([pl, np.full(G, idx["grasped_object"])]).astype(np.int32)
    po = np.concatenate([po, pts]).astype(np.float32)
    pairs, _ = _panda_pairs(idx)
    cpairs = [(int(origin_col[a]), int(origin_col[b])) for a, b in pairs]
    for n in ("panda_link0", "panda_link1", "panda_link2", "panda_link3"):
        cpairs += [(P0 + m, int(origin_col[idx[n]])) for m in range(G)]
    return PointsTemplate(point_link=pl, point_offset=po, obj_cols=[int(c) for c in sphere_col] + list(range(P0, P0 + G)),
                          self_pairs=cpairs, ee_link=idx["ee_link"])


# attached-point kernels: name -> (urdf file, PointsTemplate factory)
SPEC_POINT_ROBOTS = {
    "panda_spheres": ("panda_arm_no_gripper.urdf", panda_spheres_points),
    "panda_grasp": ("panda_arm_no_gripper_grasped_object.urdf", panda_grasp_points),
    "panda_spheres_grasp": ("panda_arm_no_gripper_grasped_object.urdf", panda_spheres_grasp_points),
}


def _is_tree(kin: KinModel) -> bool:
    kids = np.bincount(np.asarray(kin.parent[1:], np.int64), minlength=kin.n_links)
    return bool((kids > 1).any())


def generate_link_kernel_source(kin: KinModel, tmpl: CollisionTemplate, ident: str, meta: Optional[dict] = None) -> str:
    """Serial chains: generate_rollout_source (FK, batched objectives, reverse pass).  Trees whose file order is a
    pre-order walk with increasing collision links: the per-link pipeline of generate_points_rollout_source."""
    use_pipeline = (_is_tree(kin) and [int(v) for v in kin.order] == list(range(kin.n_links)) and
                    sorted(tmpl.obj_links) == list(tmpl.obj_links) and TREE_PIPELINE)
    if use_pipeline:
        return generate_points_rollout_source(kin, link_points_template(kin, tmpl), ident, link_mode=True)
    return generate_rollout_source(kin, tmpl, ident, meta=meta)


TREE_PIPELINE = False     # measured on MI355X: dual Panda 23.5 -> 37.8 us, UR10+Allegro 36.9 -> 41.7 us: one scene evaluation per link
                          # (NL = 1) loses the ILP of the batched NL = 10 evaluation; kept switchable for robots with many links per DOF


_template_cache: Dict[str, tuple] = {}


def template_for(ident: str):
    """(KinModel, CollisionTemplate) of a robot in SPEC_ROBOTS (a fresh KinModel per call: callers may move its base)."""
    from .kinematics import URDF_DIR
    urdf, fn = SPEC_ROBOTS[ident]
    kin = KinModel.from_urdf(str(URDF_DIR / urdf))
    return kin, fn(kin)


def aot_units():
    """[(ident, model hash, CollisionTemplate)] of the ahead-of-time link units, computed once per process."""
    if not _template_cache:
        for ident in SPEC_ROBOTS:
            kin, tmpl = template_for(ident)
            _template_cache[ident] = (model_hash(kin), tmpl)
    return [(k, v[0], v[1]) for k, v in _template_cache.items()]


def generate_all(out_dir) -> List[str]:
    from pathlib import Path
    from .kinematics import URDF_DIR
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    written = []
    for ident, (urdf, tmpl_fn) in SPEC_ROBOTS.items():
        kin = KinModel.from_urdf(str(URDF_DIR / urdf))
        src = generate_link_kernel_source(kin, tmpl_fn(kin), ident)
        path = out_dir / f"spec_{ident}.hip"
        if not path.exists() or path.read_text() != src:
            path.write_text(src)
        written.append(path.name)
    for ident, (urdf, tmpl_fn) in SPEC_POINT_ROBOTS.items():
        kin = KinModel.from_urdf(str(URDF_DIR / urdf))
        src = generate_points_rollout_source(kin, tmpl_fn(kin), ident)
        path = out_dir / f"spec_{ident}.hip"
        if not path.exists() or path.read_text() != src:
            path.write_text(src)
        written.append(path.name)
    return written
