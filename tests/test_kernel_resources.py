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


def _flags(src):
    """The flags csrc/Makefile builds this file with (the encoder is built without SLP vectorisation)."""
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"] + (["-fno-slp-vectorize"] if src == "hrs_encoder.hip" else [])


_ASM = {}  # source file -> its generated assembly (one compilation per file and test session)


def _assembly(src, tmp_path):
    if src not in _ASM:
        out = tmp_path / (src + ".s")
        subprocess.check_call([HIPCC] + _flags(src) + ["--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", str(out)], stderr=subprocess.DEVNULL)
        _ASM[src] = out.read_text()
    return _ASM[src]


def _metadata(src, tmp_path):
    kernels, cur = {}, {}
    for line in _assembly(src, tmp_path).splitlines():
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


def _kernel_bodies(src, tmp_path):
    bodies, name, cur = {}, None, []
    for line in _assembly(src, tmp_path).splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
        elif line.startswith(".Lfunc_end") and name:
            bodies[name] = cur
            name = None
        elif name is not None:
            t = line.strip()
            if t and not t.startswith((";", ".", "//")) and not t.endswith(":"):
                cur.append(t)
    return bodies


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not installed")
@pytest.mark.parametrize("src,needle", [("hrs_encoder.hip", "k_gemm_down"), ("hrs_mfma.hip", "k_mfma_scan_wide")])
def test_asm_mfma_hazard_padding(src, needle, tmp_path):
    """The hand-issued MFMAs of these kernels carry their own wait states: every one is directly preceded by an s_nop
    (a vector write of an operand may sit right in front of the asm statement), and no instruction other than an MFMA
    touches an accumulator register until at least 18 wait states after the last MFMA (the result of an 8-pass MFMA is
    not interlocked for other readers)."""
    found = 0
    for name, body in _kernel_bodies(src, tmp_path).items():
        if needle not in name:
            continue
        last_mfma, since = None, 0
        for idx, ins in enumerate(body):
            op = ins.split()[0]
            if op.startswith("v_mfma"):
                found += 1
                assert body[idx - 1].startswith("s_nop"), (name, idx, body[idx - 1])
                last_mfma, since = idx, 0
                continue
            if last_mfma is not None:
                touches_acc = re.search(r"\ba\[?\d", ins) is not None
                if touches_acc:
                    assert since >= 18, (name, idx, ins, since)
                    last_mfma = None  # checked up to the first reader
                    continue
                since += (int(ins.split()[1]) + 1) if op == "s_nop" else 1
    assert found >= 64


def _sregs(text):
    regs = set()
    for lo, hi in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        regs.update(range(int(lo), int(hi) + 1))
    regs.update(int(n) for n in re.findall(r"\bs(\d+)\b", text))
    return regs


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not installed")
@pytest.mark.parametrize("src", ["hrs_mfma.hip", "hrs_exact.hip"])
def test_scalar_load_destinations_rest_until_the_wait(src, tmp_path):
    """A scalar load's destination registers hold stale values until lgkmcnt reaches 0, and nothing interlocks a read
    before that.  The scan kernels issue such loads from hand-written statements whose wait is a second statement
    further down; should the compiler ever copy or spill the destination in between, the copy would be taken before
    the data has arrived.  Checked on the generated code of every kernel, in program order along the fall-through
    path: from a scalar load to the next wait for lgkmcnt(0) no instruction names one of its destination registers."""
    loads = 0
    for name, body in _kernel_bodies(src, tmp_path).items():
        pending = {}  # register -> index of the load
        for idx, ins in enumerate(body):
            op = ins.split()[0]
            if (op == "s_waitcnt" and "lgkmcnt(0)" in ins) or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                pending = {}  # (what follows an unconditional branch is reached from elsewhere)
                continue
            used = _sregs(ins)
            if op.startswith("s_load_dword") or op.startswith("s_buffer_load"):
                dst = _sregs(ins.split(",")[0])
                src_regs = _sregs(",".join(ins.split(",")[1:]))
                assert not (src_regs & set(pending)), (name, idx, ins, "address from a pending destination")
                assert not (dst & set(pending)), (name, idx, ins, "two loads into one register")
                for r in dst:
                    pending[r] = idx
                loads += 1
                continue
            hit = used & set(pending)
            assert not hit, (name, idx, ins, sorted(hit), body[pending[min(hit)]])
    assert loads > 0


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not installed")
def test_plain_exact_scan_keeps_its_dma_ring_in_flight(tmp_path):
    """The streaming scan retires its LDS-DMA segments with one counted wait per segment (vmcnt(16): two segments of
    eight requests stay in flight).  A global load of the compiler's own anywhere on the loop's path makes it place a
    vmcnt(0) in front of that counted wait, which drains the ring at every segment (the one-query scan fell from 75 %
    to 66 % of the HBM peak that way).  The plain instantiations (template EXT = false) therefore carry no such load,
    nor do the by-classes ones (LV = true: their class bytes arrive by LDS-DMA): in their generated code the counted
    wait is never directly preceded by a full drain."""
    checked = 0
    for name, body in _kernel_bodies("hrs_exact.hip", tmp_path).items():
        if "k_exact_scan" not in name or not re.search(r"Lb[01]ELb0ELb[01]EEv9ExactArgs$", name):
            continue
        waits = [i for i, ins in enumerate(body) if ins.startswith("s_waitcnt vmcnt(16)")]
        assert waits, name
        for i in waits:
            before = body[max(0, i - 12):i]
            assert not any(b.startswith("s_waitcnt") and "vmcnt(0)" in b for b in before), (name, before)
            checked += 1
    assert checked >= 24


def _vregs(text):
    regs = set()
    for lo, hi in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        regs.update(range(int(lo), int(hi) + 1))
    regs.update(int(n) for n in re.findall(r"\bv(\d+)\b", text))
    return regs


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not installed")
def test_vector_load_destinations_rest_until_a_wait(tmp_path):
    """The encoder's GEMMs load residual rows by hand-written global loads whose wait is a separate statement further
    down (counted with the LDS-DMA's).  Until that wait the destination registers hold stale values, and nothing stops
    the compiler from copying them elsewhere in between (it did, in a kernel whose load and wait were a loop iteration
    apart: a few rows of the block output were then slightly wrong).  In the generated code of every encoder kernel no
    instruction may read the destination of a vector load before the next wait on vmcnt."""
    loads = 0
    for name, body in _kernel_bodies("hrs_encoder.hip", tmp_path).items():
        pending = {}
        for idx, ins in enumerate(body):
            op = ins.split()[0]
            if op == "s_waitcnt" and "vmcnt" in ins:
                pending = {}
                continue
            if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                pending = {}
                continue
            if op.startswith(("global_load_dword", "global_load_ushort", "global_load_ubyte", "buffer_load_dword", "scratch_load")) and "lds" not in op:
                parts = ins.split(",")
                dst = _vregs(parts[0])
                src = _vregs(",".join(parts[1:]))
                assert not (src & set(pending)), (name, idx, ins, "address from a pending destination")
                for r in dst:
                    pending[r] = idx
                loads += 1
                continue
            if op.startswith("global_load_lds") or op.startswith("global_store") or op.startswith("s_") or op.startswith("ds_read"):
                used = _vregs(",".join(ins.split(",")[1:])) if op.startswith("ds_read") else _vregs(ins)
            else:
                used = _vregs(",".join(ins.split(",")[1:]))  # sources only: overwriting a dead destination is harmless
            hit = used & set(pending)
            assert not hit, (name, idx, ins, sorted(hit), body[pending[min(hit)]])
    assert loads > 0


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not installed")
def test_encoder_gemms_with_counted_waits_do_not_spill(tmp_path):
    """A spill is a scratch store and, worse, a scratch load inside the loop: both count in vmcnt, and the kernels
    that retire their LDS-DMA with counted waits would then wait for the wrong operations."""
    ks = {n: k for n, k in _metadata("hrs_encoder.hip", tmp_path).items() if "k_gemm_ws" in n or "k_gemm_down" in n}
    assert len(ks) >= 4
    for name, k in ks.items():
        assert int(k["vgpr_spill_count"]) == 0 and int(k["private_segment_fixed_size"]) == 0, (name, k)


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not installed")
def test_small_rescore_kernel_fits_two_workgroups_without_spilling(tmp_path):
    """The small instance of the re-score kernel runs two workgroups per CU (256 registers per wave).  With a four-deep
    window of row pieces in registers it needed more, spilled 8 to 25 registers into the loop that waits for those very
    loads, and ran a tenth slower than with a three-deep one that fits."""
    ks = {n: k for n, k in _metadata("hrs_mfma.hip", tmp_path).items() if "k_select_rescore" in n and n.endswith("ELb0EEv10SelectArgs")}
    assert len(ks) == 3
    for name, k in ks.items():
        assert int(k["vgpr_count"]) <= 256, (name, k)
        assert int(k["vgpr_spill_count"]) == 0 and int(k["private_segment_fixed_size"]) == 0, (name, k)
