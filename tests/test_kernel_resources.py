"""Register budget of the kernels that issue MFMAs from inline asm with operands pinned in the accumulator
registers (k_gemm_down, k_mfma_scan_wide).  They are written to fill the 512-register file exactly; a build in
which the compiler spills one of the pinned fragments produced slightly wrong results on the GPU (found with a
five-deep ring in k_gemm_down that needed twelve more registers), so a spill in them fails here, on the CPU."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hybrid_rag_system_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _metadata(src, tmp_path):
    out = tmp_path / (src + ".s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only",
                           "-S", os.path.join(CSRC, src), "-o", str(out)], stderr=subprocess.DEVNULL)
    kernels, cur = {}, {}
    for line in out.read_text().splitlines():
        m = re.match(r"\s+\.(name|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|vgpr_count):\s+(\S+)", line)
        if not m:
            if line.strip().startswith("- .") and cur.get("name"):
                kernels[cur["name"]] = cur
                cur = {}
            continue
        cur[m.group(1)] = m.group(2)
    if cur.get("name"):
        kernels[cur["name"]] = cur
    return kernels


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not installed")
@pytest.mark.parametrize("src,needle", [("hrs_encoder.hip", "k_gemm_down"), ("hrs_mfma.hip", "k_mfma_scan_wide")])
def test_asm_mfma_kernels_do_not_spill(src, needle, tmp_path):
    ks = {n: k for n, k in _metadata(src, tmp_path).items() if needle in n}
    assert ks, f"no kernel named *{needle}* in {src}"
    for name, k in ks.items():
        assert int(k["vgpr_spill_count"]) == 0, (name, k)
        assert int(k["private_segment_fixed_size"]) == 0, (name, k)
