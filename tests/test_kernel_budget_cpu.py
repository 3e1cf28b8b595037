"""CPU: register / scratch budget of the big-tile GEMM kernels, read from hipcc's resource remarks (cross-compiles without a GPU).

The 256x256 kernels keep 128 accumulator VGPRs plus ~110 more across their k-loop.  Twice in round 2 an innocent-looking epilogue
change pushed the k-loop's invariants into scratch: 5-20 % off every GEMM, and one such build computed wrong tiles after a ragged
tile (tests/test_gpu_kernels.py::test_gemm_ring_ragged_rows_over_several_tiles_per_block).  This test fails the build instead."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

BUDGET = [   # (substring of the mangled kernel name, max scratch bytes per lane)
    # (ring kernels: ~40 scratch instructions per kernel, all at the tile boundary / in the epilogue -- none inside the K-tile loops,
    #  checked on the ISA when the cross-tile DMA stream went in (80 B) and again with the SwiGLU-backward form)
    ("gemm_nt_bf16_ring_kernelILi0ELb0ELi0ELb1ELi1ELb0E", 96),     # ring, common epilogue forms: the decoder's linears (256- and 192-row tiles)
    ("gemm_nt_bf16_ring_kernelILi0ELb0ELi0ELb1ELi3ELb0E", 96),     # ring, bias / activation in front (the ViT)
    ("gemm_tn_bf16_pp_kernelILb0E", 128),                          # weight gradients (+ sums of squares: epilogue-only spills, none in the k-loop)
    ("gemm_tn_bf16_pp_kernelILb1E", 0),                            # input gradients
    ("gemm_nt_bf16_ring_kernelILi0ELb0ELi0ELb1ELi4ELb0E", 96),     # fused-qkv form (its own instantiation: DESIGN.md section 4)
    ("gemm_nt_fp8_pp_kernel", 0),
    ("gemm_nt_bf16_kernelILi128ELi128ELi2ELi2E", 0),
    ("gemm_nt_skinny_kernelILi256ELi3E", 0),                       # adapter-sized NT products, one resident 120-KiB block per CU
    ("gemm_nt_skinny_kernelILi64ELi4E", 0),
]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_gemm_kernels_stay_inside_their_register_budget(tmp_path):
    src = os.path.join(ROOT, "a3vlm_amd", "csrc", "a3v_gemm.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", str(tmp_path / "g.o")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    scratch, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            scratch[name] = int(m.group(1))
    assert scratch, "no resource remarks in the compiler output"
    # the product library carries no experiment kernels (they are built only with `make EXPERIMENTS=1`)
    for gone in ("gemm_nt_bf16_w4_kernel", "gemm_nt_bf16_ov_kernel", "gemm_nt_bf16_pp32_kernel", "gemm_nt_bf16_pp_kernel"):
        assert not any(gone in n for n in scratch), f"{gone} is compiled into the product library"
    ring = [n for n in scratch if "gemm_nt_bf16_ring_kernel" in n]
    # 256-row tiles: common / +bias-activation / fused-qkv sets; 192-row tiles (round 5): common / +bias-activation; fp8 operands (round 5):
    # common / fused-qkv at 256 rows, common at 192
    assert len(ring) == 8, ring
    for n in ring:
        if "ELi192ELb" in n:
            assert scratch[n] == 0, (n, scratch[n])          # six row tiles per wave leave 40 registers: nothing spills
    skinny = [n for n in scratch if "gemm_nt_skinny_kernel" in n]
    assert len(skinny) == 2, skinny      # the two forms a3v_gemm_nt_splitk picks; the sweep's other rows / stages only with -DA3V_ABLATION
    for key, limit in BUDGET:
        hits = {n: v for n, v in scratch.items() if key in n}
        assert hits, f"kernel {key} not found (renamed? update the budget table)"
        for n, v in hits.items():
            assert v <= limit, f"{n}: {v} B of scratch per lane (budget {limit})"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_owned_agpr_kernels_are_never_touched_by_the_compiler(tmp_path):
    """The one-pass dK + dV attention backward keeps its accumulators, the K / V fragments and S / dP in hand-owned AGPRs (every MFMA of
    its loop is an asm statement on literal a[...] registers).  That is only sound while hipcc itself never allocates an AGPR in those
    kernels -- a compiler spill into the owned range would be silent corruption -- so the generated code must contain NO v_accvgpr_* /
    v_mfma instruction outside the asm statements and no scratch."""
    src = os.path.join(ROOT, "a3vlm_amd", "csrc", "a3v_attn_bwd.hip")
    out = tmp_path / "b.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "--cuda-device-only", "-S", src, "-o", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    kernels, cur, in_asm = {}, None, False
    for line in open(out):
        m = re.match(r"^(_ZN\S*attn_bwd_dkv2_kernel\S*):", line)
        if m:
            cur = m.group(1)
            kernels[cur] = {"bad": [], "scratch": 0, "mfma": 0}
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        elif re.search(r"\bv_accvgpr_|\bv_mfma_", line):
            if in_asm:
                kernels[cur]["mfma"] += "v_mfma" in line
            else:
                kernels[cur]["bad"].append(line.strip())
        elif "scratch_" in line:
            kernels[cur]["scratch"] += 1
    assert len(kernels) == 4, list(kernels)          # hd 64 / 128 x packed / plain
    for name, k in kernels.items():
        assert not k["bad"], f"{name}: compiler-generated AGPR / MFMA instructions: {k['bad'][:3]}"
        assert k["scratch"] == 0, f"{name}: {k['scratch']} scratch instructions"
        assert k["mfma"] > 60, (name, k["mfma"])            # the loop really is the asm stream (hd 64: 88 MFMAs over its step variants)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_prefill_attention_kernels_keep_two_waves_per_simd_and_no_scratch(tmp_path):
    """The product prefill kernels run two blocks per CU (__launch_bounds__(256, 2)): <= 256 registers and no scratch.  The hd-128 form
    reads its K / V^T fragments ahead of their MFMAs since round 4 (244 registers): a few more and hipcc spills into the tile loop.
    The one-wave-per-SIMD experiment kernel (attn_prefill_w64_kernel) must not be in the product object, and the row-maximum exchange
    must be the asm v_permlane32_swap (the builtin with one value in both operands silently loses the exchange)."""
    src = os.path.join(ROOT, "a3vlm_amd", "csrc", "a3v_attn.hip")
    out = tmp_path / "a.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "--cuda-device-only", "-S", src, "-o", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    text = open(out).read()
    assert "attn_prefill_w64_kernel" not in text, "the experiment kernel is compiled into the product object"
    found = 0
    for m in re.finditer(r"^(_ZN\S*attn_prefill_bf16_kernel\S*):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, flags=re.S | re.M):
        name, body = m.group(1), m.group(2)
        tail = text[m.start():]
        vg = int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", tail).group(1))
        sc = int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", tail).group(1))
        assert vg <= 256, (name, vg)
        assert sc == 0, (name, sc)
        assert "v_permlane32_swap_b32 v" in body, name
        found += 1
    assert found >= 4, found
