"""The staged two-term fp16 form shares a translation unit (k_gemm3p.hip) and an epilogue header with kernels that run on the default path.
This test reads the BUILT code object (stable_diffusion_burn_amd/build/k_gemm3p.hip.o, written by build()) and checks that the two forms did
not leak into each other: every three-plane instantiation (the kernels the default path launches) multiplies with v_mfma_f32_16x16x32_bf16
only -- six per fragment pair, one k tile's worth (the k loop is not unrolled) --; every two-plane instantiation with
v_mfma_f32_16x16x32_f16 only, three per fragment pair; neither spills to scratch.  (That the three-plane kernels' instruction streams are
byte-identical to those before the generalisation was checked once, function by function, when it was made: DESIGN.md section 10.)"""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
OBJ = ROOT / "stable_diffusion_burn_amd" / "build" / "k_gemm3p.hip.o"
LLVM = Path("/opt/rocm/lib/llvm/bin")


def _disassembly(tmp_path):
    if not OBJ.exists() or not (LLVM / "llvm-objdump").exists():
        pytest.skip("needs the built object (python -m stable_diffusion_burn_amd.build) and ROCm's llvm tools")
    fat, dev = tmp_path / "fat.bin", tmp_path / "dev.co"
    subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(OBJ), str(tmp_path / "copy.o")], check=True)
    subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={dev}"], check=True)
    return subprocess.run([str(LLVM / "llvm-objdump"), "-d", str(dev)], check=True, capture_output=True, text=True).stdout


def test_the_two_operand_forms_do_not_leak_into_each_other(tmp_path):
    text = _disassembly(tmp_path)
    funcs = {}
    for m in re.finditer(r"^[0-9a-f]+ <(_ZN4sdmi18conv_gemm3p_kernel\w+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", text, re.S | re.M):
        funcs[m.group(1)] = m.group(2)
    three = {k: v for k, v in funcs.items() if k.endswith("ELi3EEEvNS_8ConvGemmE")}
    two = {k: v for k, v in funcs.items() if k.endswith("ELi2EEEvNS_8ConvGemmE")}
    assert len(three) == 12 and len(two) == 12, (len(three), len(two))      # 9 tiles + 3 diagnostic instantiations each
    for name, body in three.items():
        mi, ni = (int(v) for v in re.match(r".*kernelILi(\d+)ELi(\d+)E", name).groups())
        n_bf16, n_f16 = body.count("v_mfma_f32_16x16x32_bf16"), body.count("v_mfma_f32_16x16x32_f16")
        assert n_f16 == 0 and n_bf16 == 6 * mi * ni, (name, n_bf16, n_f16)  # one k tile's worth: the k loop is not unrolled
        assert "scratch_" not in body, name
    for name, body in two.items():
        mi, ni = (int(v) for v in re.match(r".*kernelILi(\d+)ELi(\d+)E", name).groups())
        n_bf16, n_f16 = body.count("v_mfma_f32_16x16x32_bf16"), body.count("v_mfma_f32_16x16x32_f16")
        assert n_bf16 == 0 and n_f16 == 3 * mi * ni, (name, n_bf16, n_f16)
        assert "scratch_" not in body, name
