"""CPU: tools/isa_audit.py, the build gate for hand-written instructions and spills.

On gfx950 a vector instruction that reads the result of a transcendental instruction needs one
instruction in between; the compiler adds the wait state for its own instructions but does not look
inside asm blocks.  The audit must find such a pair in the assembly, must not flag pairs the compiler made
(it has dealt with those) nor pairs with an instruction in between, and must fail a kernel whose spills
exceed the recorded budget."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AUDIT = os.path.join(ROOT, "tools", "isa_audit.py")

METADATA = """
	.amdgpu_metadata
---
amdhsa.kernels:
  - .agpr_count:     0
    .group_segment_fixed_size: 0
    .name:           _Z6kernelv
    .private_segment_fixed_size: {scratch}
    .sgpr_count:     20
    .sgpr_spill_count: 0
    .vgpr_count:     30
    .vgpr_spill_count: {spill}
...
	.end_amdgpu_metadata
"""


def asm(body, spill=0, scratch=0):
    return "_Z6kernelv:\n" + body + "\ts_endpgm\n" + METADATA.format(spill=spill, scratch=scratch)


def run(tmp_path, text, *args):
    path = tmp_path / "k.s"
    path.write_text(text)
    return subprocess.run([sys.executable, AUDIT, str(path), *args], capture_output=True, text=True)


def test_hand_written_consumer_right_behind_a_reciprocal_is_found(tmp_path):
    r = run(tmp_path, asm("\tv_rcp_f32_e32 v13, v6\n\t;;#ASMSTART\n\tv_max_f32 v13, v13, v36\n\t;;#ASMEND\n"))
    assert r.returncode == 1 and "HAZARD" in r.stdout and "v_rcp_f32_e32 v13, v6" in r.stdout


def test_an_instruction_in_between_or_a_compiler_made_pair_passes(tmp_path):
    spaced = "\tv_rcp_f32_e32 v0, v10\n\tv_rcp_f32_e32 v1, v11\n\t;;#ASMSTART\n\tv_max_f32 v0, v0, v8\n\t;;#ASMEND\n"
    assert run(tmp_path, asm(spaced)).returncode == 0
    compiler_pair = "\tv_rcp_f32_e32 v0, v10\n\tv_mul_f32_e32 v2, v0, v3\n"      # the hazard recogniser inserts its own s_nop where needed
    assert run(tmp_path, asm(compiler_pair)).returncode == 0
    other_register = "\tv_sqrt_f32_e32 v4, v10\n\t;;#ASMSTART\n\tv_min_f32 v0, v1, v8\n\t;;#ASMEND\n"
    assert run(tmp_path, asm(other_register)).returncode == 0


def test_register_ranges_are_understood(tmp_path):
    r = run(tmp_path, asm("\tv_rsq_f32_e32 v5, v6\n\t;;#ASMSTART\n\tv_pk_mul_f32 v[8:9], v[4:5], v[10:11]\n\t;;#ASMEND\n"))
    assert r.returncode == 1 and "HAZARD" in r.stdout


def test_spills_beyond_the_budget_fail_the_build(tmp_path):
    budget = tmp_path / "budget.json"
    budget.write_text(json.dumps({"kernel": {"vgpr_spill": 2, "sgpr_spill": 0, "scratch": 12}}))
    ok = run(tmp_path, asm("\tv_mov_b32 v0, v1\n", spill=2, scratch=12), "--budget", str(budget))
    assert ok.returncode == 0, ok.stdout
    over = run(tmp_path, asm("\tv_mov_b32 v0, v1\n", spill=5, scratch=24), "--budget", str(budget))
    assert over.returncode == 1 and "SPILL" in over.stdout
    unlisted = run(tmp_path, asm("\tv_mov_b32 v0, v1\n", spill=1, scratch=4), "--budget", str(tmp_path / "k.s.none.json"))
    assert unlisted.returncode != 0


def test_the_built_library_passes_its_own_audit():
    s = os.path.join(ROOT, "build", "rt_hip-hip-amdgcn-amd-amdhsa-gfx950.s")
    if not os.path.exists(s):
        import pytest
        pytest.skip("no device assembly here (the library was not built in this tree)")
    r = subprocess.run([sys.executable, AUDIT, s, "--budget", os.path.join(ROOT, "tools", "spill_budget.json")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-1500:]
