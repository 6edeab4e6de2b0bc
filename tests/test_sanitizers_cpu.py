"""ASan + UBSan build of the host-only C++ of libsdmi (tokenizer, PNG writer, native .mpk reader) driven through awkward
and corrupted inputs (tests/san/host_san_main.cpp).  SURVEY.md section 5 lists a sanitizer test build among the aux
subsystems; the reference has none."""
import shutil
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "stable_diffusion_burn_amd" / "csrc"


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_code_is_clean_under_asan_ubsan(tmp_path):
    sys.path.insert(0, str(ROOT))
    from tools import mpk_to_dump as M
    exe = tmp_path / "host_san"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           str(ROOT / "tests" / "san" / "host_san_main.cpp"), str(CSRC / "tokenizer.cpp"), str(CSRC / "png_writer.cpp"), str(CSRC / "mpk_reader.cpp"),
           "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    g = np.random.default_rng(0)
    tensors = {"unet/conv_out/weight": g.standard_normal((4, 8, 3, 3)).astype(np.float32), "unet/conv_out/bias": g.standard_normal(4).astype(np.float32),
               "unet/norm_out/weight": np.ones(8, np.float32), "unet/norm_out/bias": np.zeros(8, np.float32),
               "clip/position_embedding/weight": g.standard_normal((5, 6)).astype(np.float32), "alphas_cumprod": np.linspace(0.9, 0.1, 10).astype(np.float32)}
    M.write_record(tensors, tmp_path / "tiny.mpk")
    r = subprocess.run([str(exe), str(ROOT / "tests" / "golden" / "mini_merges.txt"), str(tmp_path / "tiny.mpk"), str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "no sanitizer report" in r.stdout
