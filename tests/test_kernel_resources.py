"""Register / scratch budget of the hot kernels, read from the gfx950 code hipcc emits (no GPU needed).

A kernel argument block that is indexed by a per-lane value gets copied to scratch memory without a word from the compiler
(k_mb did, once: 21 -> 38 us per 1080p frame), and a register count past 128 halves k_mb's waves per SIMD; both are
invisible to the parity tests."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vp8oclenc_amd", "csrc")
HOT = {   # file -> {kernel name fragment: max VGPRs}
    "kernels_mb.hip": {"k_mb_bE": 128, "k_mb_pE": 128, "k_mb_p_conformant": 128},     # both forms (VP8HIP_MB_PACKED): four waves per SIMD
    "kernels_s2.hip": {"k_search2ILb": 72, "k_search2_bILb": 72, "k_search2_bsILb": 72},      # seven waves per SIMD ... (both forms of the cost phase: SPREAD and lane = candidate; one group of eight blocks per workgroup or four in a loop)
    "kernels_me.hip": {"k_search1": 128, "k_search1_bILb0": 64, "k_search1_plE": 64, "k_search1_pl_b": 64, "k_search1_plr_b": 72, "k_search1_coarse": 64, "k_pyramid": 128, "k_pack_b": 64},   # the loop form: eight (a form with 7 % fewer instructions and 75 registers was no faster)
    "kernels_lf4.hip": {"k_loop_filter4": 128},
}
LDS = {"k_search2ILb": 23296, "k_search2_bILb": 23296, "k_search2_bsILb": 23296}   # ... and seven workgroups per CU (163 840 / 7); an MFMA result must land in VGPRs (no AGPRs)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
@pytest.mark.parametrize("src", sorted(HOT))
def test_hot_kernels_use_no_scratch_and_stay_inside_their_register_budget(src, tmp_path):
    out = tmp_path / "k.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-x", "hip",
                    "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", str(out), "-w"], check=True, timeout=600)
    text = out.read_text()
    seen = set()
    for m in re.finditer(r"\.name:\s+(\S+)\n((?:.*\n)*?)\s+\.wavefront_size", text):
        name, body = m.group(1), m.group(2)
        for frag, max_vgprs in HOT[src].items():
            if frag in name:
                seen.add(frag)
                scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", body).group(1))
                vgprs = int(re.search(r"\.vgpr_count:\s+(\d+)", body).group(1))
                spills = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", body).group(1))
                assert scratch == 0 and spills == 0, f"{name}: {scratch} B of scratch, {spills} spilled VGPRs"
                assert vgprs <= max_vgprs, f"{name}: {vgprs} VGPRs (budget {max_vgprs})"
                if frag in LDS:
                    lds = int(re.search(r"\.amdhsa_kernel\s+" + re.escape(name.replace(".kd", "")) + r"\n(?:.*\n)*?\s+\.amdhsa_group_segment_fixed_size\s+(\d+)", text).group(1))
                    agprs = int(re.search(r"\.set\s+" + re.escape(name.replace(".kd", "")) + r"\.num_agpr,\s*(\d+)", text).group(1))
                    assert lds <= LDS[frag], f"{name}: {lds} B of LDS (budget {LDS[frag]})"
                    assert agprs == 0, f"{name}: {agprs} AGPRs: every MFMA result would cost a v_accvgpr_read"
    assert seen == set(HOT[src]), f"kernels not found in {src}: {set(HOT[src]) - seen}"
