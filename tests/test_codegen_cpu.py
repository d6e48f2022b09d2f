"""CPU-only checks of the model-specialising code generator."""
import re

from helpers import model
from torch_robotics_amd import codegen


def test_generated_source_is_deterministic_and_folded():
    kin = model("panda_arm_no_gripper")
    tmpl = codegen.panda_template(kin)
    assert tmpl.obj_links == [2, 3, 5, 7, 9] and tmpl.ee_link == 10
    # pair order of robot_base.py:110-118 mapped to link indices (SURVEY.md section 4 KAT)
    self_links = [9, 0, 1, 2, 3, 4, 5, 6]
    kat = [(0, 1), (0, 2), (0, 3), (5, 2), (6, 1), (6, 2), (6, 3), (7, 1), (7, 2), (7, 3)]
    assert tmpl.self_pairs == [(self_links[a], self_links[b]) for a, b in kat]
    src = codegen.generate_rollout_source(kin, tmpl, "panda")
    assert src == codegen.generate_rollout_source(kin, tmpl, "panda")
    assert f"0x{codegen.model_hash(kin):016x}ull" in src
    # structural zeros are folded: the identity-base variant needs far fewer multiplies than 11 dense composes
    body = src.split("k_rollout_bg")[0]
    assert len(re.findall(r"fmaf\(|\*", body)) < 700
    assert "trk_sincos(qh6" in body and "passbits" in body and "trk_sincos2(qh0, qh1" in body


def test_committed_generated_sources_equal_the_generator_output(tmp_path):
    """csrc/generated/spec_panda.hip is committed so the headline kernel can be read without running the build (the other units
    are build products only); whatever is present must be exactly what `codegen.generate_all` writes (the build rewrites a file
    only when it differs, so a clean tree stays clean)."""
    from pathlib import Path
    committed = Path(codegen.__file__).resolve().parent / "csrc" / "generated"
    names = codegen.generate_all(tmp_path)
    present = sorted(p.name for p in committed.glob("spec_*.hip"))
    assert "spec_panda.hip" in present and set(present) <= set(names)
    for n in present:
        assert (tmp_path / n).read_text() == (committed / n).read_text(), f"{n} is stale: run __graft_entry__.build()"


def test_every_bundled_urdf_has_an_ahead_of_time_unit():
    """build() compiles a generated unit for EVERY URDF under data/urdf (the benchmark robots with their own collision models, the
    others with codegen.default_template), so no bundled robot's FK family ever takes the table-driven kernels."""
    from helpers import ROBOTS, URDF
    files = {urdf for urdf, _ in codegen.SPEC_ROBOTS.values()}
    assert files == {f"{r}.urdf" for r in ROBOTS}
    assert {p.name for p in URDF.glob("*.urdf")} - files == {"panda_arm_no_gripper_grasped_object.urdf"}      # an attached-point unit's
    hashes = set()
    for ident, mh, tmpl in codegen.aot_units():
        kin, t2 = codegen.template_for(ident)
        assert codegen.model_hash(kin) == mh and t2.obj_links == tmpl.obj_links and 0 <= tmpl.ee_link < kin.n_links
        assert tmpl.obj_links == sorted(set(tmpl.obj_links)) and all(0 <= i < kin.n_links for i in tmpl.obj_links)
        hashes.add(mh)
    assert len(hashes) == len(codegen.SPEC_ROBOTS)


def test_gp_fused_kernels_are_generated(monkeypatch):
    """trk_rollout_gp_cost_grad's generated kernels: the tree schedule (k_rollout_gpt) is part of every unit that stages whole rows;
    the segment schedule (k_rollout_gp: the dual Panda's arms one after the other, measured slower) only on request."""
    kin, tmpl = codegen.template_for("dual_panda")
    src = codegen.generate_rollout_source(kin, tmpl, "dual_panda")
    assert "k_rollout_gpt_bi" in src and "launch_gp" in src and "k_rollout_gp_bi" not in src
    monkeypatch.setenv("TRK_GP_SCHEDULE", "segments")
    seg = codegen.generate_rollout_source(kin, tmpl, "dual_panda")
    assert "k_rollout_gp_bi" in seg and "segment 0: links 1 .. 11" in seg and "segment 1: links 12 .. 22" in seg
    assert "if (a.w.w_self != 0.0f) return 1;" in seg          # arm-vs-arm pairs: served by the two-launch form
    monkeypatch.delenv("TRK_GP_SCHEDULE")
    kin2, tmpl2 = codegen.template_for("ur10_allegro")           # ring-staged positions: no fused kernel, the C ABI takes the two-launch form
    src2 = codegen.generate_rollout_source(kin2, tmpl2, "ur10_allegro")
    assert "k_rollout_gpt_bi" not in src2 and "static int launch_gp" not in src2


def test_model_hash_distinguishes_models():
    hashes = {codegen.model_hash(model(n)) for n in ("panda_arm_no_gripper", "panda_arm_hand", "ur10", "iiwa7")}
    assert len(hashes) == 4


def test_point_templates_match_the_robots():
    """The point sets / collision columns baked into the generated attached-point kernels are exactly what RobotPanda
    and PlanningTask build at run time (else the dispatcher would silently fall back to the table-driven kernel)."""
    import numpy as np
    import torch
    import torch_robotics_amd as tra
    from torch_robotics_amd import codegen
    from torch_robotics_amd.kinematics import URDF_DIR
    from torch_robotics_amd.kinmodel import KinModel
    TA = dict(device="cpu", dtype=torch.float32)
    cases = {"panda_spheres": dict(link_sphere_model="panda"),
             "panda_grasp": dict(grasped_object=tra.GraspedObjectPandaBox(tensor_args=TA)),
             "panda_spheres_grasp": dict(link_sphere_model="panda", grasped_object=tra.GraspedObjectPandaBox(tensor_args=TA))}
    assert set(cases) == set(codegen.SPEC_POINT_ROBOTS)
    for ident, kw in cases.items():
        robot = tra.RobotPanda(tensor_args=TA, **kw)
        spec = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, tensor_args=TA).build_cost_spec()
        urdf, fn = codegen.SPEC_POINT_ROBOTS[ident]
        kin = KinModel.from_urdf(str(URDF_DIR / urdf))
        pt = fn(kin)
        pl, po = robot.collision_point_set()
        np.testing.assert_array_equal(pl, pt.point_link)
        np.testing.assert_array_equal(po, pt.point_offset)
        assert list(spec.obj_link_idx) == list(pt.obj_cols)
        assert [tuple(int(v) for v in spec.self_link_idx[p]) for p in spec.self_pairs] == [tuple(p) for p in pt.self_pairs]
        assert spec.ee_link in (-1, pt.ee_link)
        assert codegen.model_hash(kin) == codegen.model_hash(robot.diff_panda._kin)
        src = codegen.generate_points_rollout_source(kin, pt, ident)
        assert f"0x{codegen.points_hash(pl, po):016x}ull" in src and "spec_flush_chunk<W" in src
    with __import__("pytest").raises(ValueError):       # columns must follow the walk order of their links
        kin = KinModel.from_urdf(str(URDF_DIR / "panda_arm_no_gripper.urdf"))
        bad = codegen.PointsTemplate(point_link=np.array([3, 1], np.int32), point_offset=np.zeros((2, 3), np.float32), obj_cols=[0, 1])
        codegen.generate_points_rollout_source(kin, bad, "bad")


def test_runtime_model_compiler_builds_and_loads_a_unit():
    """jit.specialize: generate + hipcc (cross-compiles without a GPU) + dlopen; the unit registers itself with libtrk.so."""
    from torch_robotics_amd import jit
    from torch_robotics_amd.kinematics import URDF_DIR
    from torch_robotics_amd.kinmodel import KinModel
    kin = KinModel.from_urdf(str(URDF_DIR / "iiwa7.urdf"))
    ident = jit.specialize(kin, obj_links=[3, 5, 7], self_pairs=[(7, 1), (6, 2)], ee_link=kin.n_links - 1)
    so = jit.JIT_DIR / f"spec_{ident}.so"
    assert so.exists() and so.stat().st_size > 10000
    assert jit.specialize(kin, obj_links=[3, 5, 7], self_pairs=[(7, 1), (6, 2)], ee_link=kin.n_links - 1) == ident   # idempotent
    other = jit.unit_ident(kin, __import__("torch_robotics_amd").codegen.CollisionTemplate(obj_links=[3, 5], ee_link=-1))
    assert other != ident


def test_concurrent_jit_builds_do_not_corrupt_each_other(tmp_path):
    """One process per GPU means several ranks may compile the same unit at once: every file must appear atomically."""
    import subprocess, sys
    from torch_robotics_amd import jit
    from torch_robotics_amd.kinematics import URDF_DIR
    from torch_robotics_amd.kinmodel import KinModel
    kin = KinModel.from_urdf(str(URDF_DIR / "ur10.urdf"))
    tmpl = __import__("torch_robotics_amd").codegen.CollisionTemplate(obj_links=[2, 4, 6], ee_link=kin.n_links - 1)
    ident = jit.unit_ident(kin, tmpl)
    for ext in ("so", "hip", "stamp"):
        (jit.JIT_DIR / f"spec_{ident}.{ext}").unlink(missing_ok=True)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from torch_robotics_amd import jit\n"
            "from torch_robotics_amd.kinematics import URDF_DIR\n"
            "from torch_robotics_amd.kinmodel import KinModel\n"
            "kin = KinModel.from_urdf(str(URDF_DIR / 'ur10.urdf'))\n"
            "print(jit.specialize(kin, [2, 4, 6], [], kin.n_links - 1))\n") % str(jit.JIT_DIR.parents[2])
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(3)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-400:] for o in outs]
    assert all(o[0].strip().endswith(ident) for o in outs)
    assert (jit.JIT_DIR / f"spec_{ident}.so").stat().st_size > 10000
    assert not list(jit.JIT_DIR.glob("*.tmp*"))


def test_unit_with_another_struct_layout_is_refused():
    """A stale on-disk unit (compiled before a change to SpecArgs / SpecEntry / trk.h) must never be dispatched: the entry's
    layout stamp is checked by `trk_spec_register` and the run-time compiler's cache key covers every header and the flags."""
    import ctypes as C
    from torch_robotics_amd import _lib, jit
    L = _lib.lib()
    reg = getattr(L, "_Z17trk_spec_registerPK9SpecEntry")
    reg.restype, reg.argtypes = C.c_int, [C.c_void_p]
    before = L.trk_spec_count()
    assert before >= 3                                            # the ahead-of-time units registered themselves
    stale = (C.c_uint32 * 64)()                                   # abi 0, sizes 0: an old-layout entry
    assert reg(C.addressof(stale)) != 0 and L.trk_spec_count() == before
    for version in range(2000, 2064):                             # whatever the current version is: wrong sizeof(SpecArgs)
        stale[0], stale[1] = version, 8
        assert reg(C.addressof(stale)) != 0 and L.trk_spec_count() == before
    assert reg(None) != 0
    # the cache stamp depends on include/trk.h and on the compile command, not on where the package lives
    stamp = jit._generator_stamp()
    cmd = " ".join(jit._compile_cmd("<src>", "<out>"))
    assert "-fno-honor-nans" in cmd and "max-ilp" in cmd
    real = jit._REPO_INCLUDE / "trk.h"
    orig = real.read_bytes()
    try:
        real.write_bytes(orig + b"\n/* layout change */\n")
        assert jit._generator_stamp() != stamp
    finally:
        real.write_bytes(orig)
    assert jit._generator_stamp() == stamp


def test_hiprtc_flags_follow_the_makefile():
    """the hipRTC fall-back compiles a unit's device half with a hard-coded flag list (jit.RTC_FLAGS); ahead-of-time and
    hipcc units take theirs from csrc/Makefile.  Every DEVICE code-generation flag of the Makefile must be in the hipRTC list, so the
    two kinds of unit cannot drift apart."""
    import re
    from pathlib import Path
    from torch_robotics_amd import jit
    mk = (Path(jit.__file__).resolve().parent / "csrc" / "Makefile").read_text()
    cxx = re.search(r"^CXXFLAGS \?= (.*)$", mk, re.M).group(1).split()
    gen = re.search(r"^GENFLAGS \?= (.*)$", mk, re.M).group(1).split()
    rule = re.search(r"^generated/%\.o:.*\n\t(.*)$", mk, re.M).group(1).split()
    device_flags = [f for f in cxx if f.startswith(("-O", "-std=", "-ffp-contract"))]
    device_flags += [cxx[i + 1] for i, f in enumerate(cxx) if f == "-Xarch_device"]
    device_flags += [rule[i + 1] for i, f in enumerate(rule) if f == "-Xarch_device"]
    device_flags += [f for f in gen if f != "-mllvm"]
    assert "-fno-slp-vectorize" in device_flags and "-mno-amdgpu-ieee" in device_flags and any("max-ilp" in f for f in device_flags)
    missing = [f for f in device_flags if f not in jit.RTC_FLAGS]
    assert missing == [], f"device flags of csrc/Makefile missing from jit.RTC_FLAGS: {missing}"
    assert "--offload-arch=gfx950" in jit.RTC_FLAGS


def test_arm_per_lane_schedule_is_planned_only_for_isomorphic_arms(monkeypatch):
    """Round 6: `arm_lane_plan` recognises a robot that is two isomorphic subtrees on one base (the dual Panda: only the arms' mounts
    differ) whose template treats the arms alike -- the unit then carries k_rollout_gpa, dispatched by I/O mode; anything else gets
    no such kernel."""
    import copy
    import numpy as np
    kin, tmpl = codegen.template_for("dual_panda")
    plan = codegen.arm_lane_plan(kin, tmpl)
    assert plan is not None and plan.n == 11 and plan.DA == 7 and plan.obj == tmpl.obj_links[:5] and plan.ee == tmpl.ee_link
    assert [(k, r) for k, r, _, _, _ in plan.differ] == [("t", 1)] and plan.differ[0][3] == -plan.differ[0][4]      # the mounts: y = +-0.35
    src = codegen.generate_rollout_source(kin, tmpl, "dual_panda")
    assert "k_rollout_gpa_bi" in src and "TRK_GP_ARM_LANES" in src and "a.io_f16 == TRK_IO_F32" in src
    for ident in ("panda", "ur10_allegro", "tiago"):
        k2, t2 = codegen.template_for(ident)
        assert codegen.arm_lane_plan(k2, t2) is None
    # arms that differ in a joint limit, or a template that treats them differently: no plan
    k3 = copy.deepcopy(kin)
    k3.upper = np.array(k3.upper, copy=True); k3.upper[14] += 0.1
    assert codegen.arm_lane_plan(k3, tmpl) is None
    t3 = copy.deepcopy(tmpl); t3.obj_links = tmpl.obj_links[:-1]
    assert codegen.arm_lane_plan(kin, t3) is None
    t4 = copy.deepcopy(tmpl); t4.ee2_link = -1
    assert codegen.arm_lane_plan(kin, t4) is None
    monkeypatch.setenv("TRK_EXP_NO_ARM_LANES", "1")
    assert "k_rollout_gpa" not in codegen.generate_rollout_source(kin, tmpl, "dual_panda")


def test_fused_jacobian_is_generated_where_the_two_walks_coincide():
    """Round 6: the JAC instantiation (trk_rollout_jacobian_cost_grad in one launch) exists exactly for the units whose stateful walk
    (clamp wherever limits exist, axis sign ignored: rigid_body.py:218-233) equals the stateless one on the chains of the tracked
    link's Jacobian columns."""
    have = {}
    for ident in codegen.SPEC_ROBOTS:
        kin, tmpl = codegen.template_for(ident)
        src = codegen.generate_link_kernel_source(kin, tmpl, ident)
        have[ident] = "launch_rjac" in src
        assert ("bool JAC = false" in src) == have[ident]
    assert have["panda"] and have["ur10_allegro"] and have["dual_panda"] and have["ur10"] and have["iiwa7"]
    assert not have["tiago"] and not have["hab_stretch"]          # prismatic joints / continuous wheels: the stateful walk differs
