"""The hand-counted LDS waits of the pipelined k loops, checked on the COMPILED kernels (no GPU): tools/dev/check_lgkm.py walks the
instruction stream hipcc produced for every kernel instance that issues its fragment reads as inline asm (k_gemm3x.hip HOIST = 3,
k_gemm_bf16x.hip PIPE = 2), models the wave's LGKM queue (LDS reads complete in issue order; lgkmcnt(n) returns when at most n are
outstanding) and fails if any instruction touches a register whose read is not covered by a wait.  A count that is one too large
would otherwise show up on the GPU as a rare, timing-dependent wrong result."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "stable_diffusion_burn_amd" / "csrc"


@pytest.mark.parametrize("src,kernel", [("k_gemm3x.hip", "ELb1ELb1ELi2ELi3E"),      # every tile shape of HOIST = 3
                                        ("k_gemm_bf16x.hip", "ELi2EEEvNS_8ConvGemmE")])  # every tile shape of PIPE = 2
def test_counted_waits_cover_every_fragment_read(src, kernel):
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "dev" / "check_lgkm.py"), str(CSRC / src), "--kernel", kernel],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "lgkm waits OK" in r.stdout
    assert r.stdout.count("== _ZN4sdmi") >= 4          # all tile shapes were found and checked


def test_the_checker_finds_a_wait_that_is_one_too_large(tmp_path):
    """sanity of the checker: a loop whose wait leaves the needed read in flight must be reported"""
    sys.path.insert(0, str(ROOT / "tools" / "dev"))
    import check_lgkm
    good = [".LBB0_1:", "ds_read_b128 v[0:3], v20", "ds_read_b128 v[4:7], v20 offset:16", "s_waitcnt lgkmcnt(1)",
            "v_mfma_f32_16x16x32_bf16 v[8:11], v[0:3], v[12:15], v[8:11]", "s_waitcnt lgkmcnt(0)",
            "v_mfma_f32_16x16x32_bf16 v[8:11], v[4:7], v[12:15], v[8:11]", "s_cbranch_scc1 .LBB0_1"]
    assert check_lgkm.check(good, False) == 0
    bad = list(good)
    bad[3] = "s_waitcnt lgkmcnt(2)"
    assert check_lgkm.check(bad, False) == 1
