"""What the compiler gives the hot kernels - registers, scratch, waves per SIMD, LDS - checked in the CPU gate: hipcc cross-compiles
gfx950 without a GPU, and `-Rpass-analysis=kernel-resource-usage` (the Makefile's `asm` target) prints the figures DESIGN.md section 4
and the kernels' own comments state.  A kernel that silently starts to spill, or loses a wave per SIMD to one more register, fails
here instead of showing up as an unexplained microsecond ("`-Rpass-analysis=kernel-resource-usage` must read
ScratchSize 0")."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "img-stitching_amd", "csrc")

# demangled-name fragment -> requirements.  vgprs_max is the budget of the occupancy the design relies on, not today's exact count.
EXPECT = {
    "pano_warp": {
        "21warp_tiles_lut_kernelILb0E": {"occupancy": 8, "scratch": 0, "lds": 16384, "vgprs_max": 64},   # K1
        "21warp_tiles_lut_kernelILb1E": {"occupancy": 8, "scratch": 0, "vgprs_max": 64},                 # K1 with exposure gains
    },
    "pano_pyramid": {
        "15pyr_down_kernelILi4E": {"occupancy": 8, "scratch": 0, "vgprs_max": 64},
        "15pyr_tail_kernelILi32ELi256ELi3E": {"scratch": 0},
    },
    "pano_blend": {
        "26blend_level_ordered_kernelILb1ELi3E": {"occupancy_min": 5, "scratch": 0, "vgprs_max": 96, "lds": 0},  # level 0
        "26blend_level_ordered_kernelILb0ELi1E": {"occupancy": 8, "scratch": 0, "vgprs_max": 64, "lds": 0},       # levels 1, 2
        "22blend_level_vec_kernelILb1ELi3E": {"scratch": 0},
        "22blend_level_vec_kernelILb0ELi1E": {"occupancy": 8, "scratch": 0},
    },
    "pano_blend_small": {
        "17norm_small_kernel": {"scratch": 0},
        "21collapse_small_kernel": {"scratch": 0},
    },
}


def _resource_usage(unit, tmp):
    work = os.path.join(tmp, unit)
    shutil.copytree(CSRC, work, ignore=shutil.ignore_patterns("*.s", "*.so", "*.o"))
    r = subprocess.run(["make", "asm", "K=" + unit], cwd=work, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out, cur = {}, None
    for line in (r.stdout + r.stderr).splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for key, pat in (("vgprs", r"remark:\s+VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
                         ("vgpr_spill", r"VGPRs Spill: (\d+)"), ("sgpr_spill", r"SGPRs Spill: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    return out


@pytest.mark.parametrize("unit", sorted(EXPECT))
def test_hot_kernels_keep_their_registers_scratch_and_occupancy(unit, tmp_path):
    if shutil.which("make") is None or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc here: the figures come from the gfx950 cross-compile")
    usage = _resource_usage(unit, str(tmp_path))
    report, bad = [], []
    for frag, want in EXPECT[unit].items():
        names = [n for n in usage if frag in n]
        assert len(names) == 1, (frag, sorted(usage))
        got = usage[names[0]]
        report.append("%s: %s" % (frag, got))
        if "occupancy" in want and got["occupancy"] != want["occupancy"]:
            bad.append("%s: %d waves per SIMD, the design relies on %d" % (frag, got["occupancy"], want["occupancy"]))
        if "occupancy_min" in want and got["occupancy"] < want["occupancy_min"]:
            bad.append("%s: %d waves per SIMD, at least %d wanted" % (frag, got["occupancy"], want["occupancy_min"]))
        if "scratch" in want and (got["scratch"] != want["scratch"] or got.get("vgpr_spill", 0) or got.get("sgpr_spill", 0)):
            bad.append("%s: ScratchSize %d B per lane, %d VGPR / %d SGPR spills" % (frag, got["scratch"], got.get("vgpr_spill", 0), got.get("sgpr_spill", 0)))
        if "vgprs_max" in want and got["vgprs"] > want["vgprs_max"]:
            bad.append("%s: %d VGPRs, budget %d" % (frag, got["vgprs"], want["vgprs_max"]))
        if "lds" in want and got["lds"] != want["lds"]:
            bad.append("%s: %d B of LDS per workgroup, %d expected" % (frag, got["lds"], want["lds"]))
    print("\n".join(report))
    assert not bad, "\n".join(bad)
