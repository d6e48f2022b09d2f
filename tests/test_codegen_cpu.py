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


def test_model_hash_distinguishes_models():
    hashes = {codegen.model_hash(model(n)) for n in ("panda_arm_no_gripper", "panda_arm_hand", "ur10", "iiwa7")}
    assert len(hashes) == 4
