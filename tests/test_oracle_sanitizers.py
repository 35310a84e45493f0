"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: GPU
sanitizers are not available on this pool).  The oracle restates code with known undefined
behaviour in the reference (double->int of out-of-range values, a read one past Sp); its own
explicit x86-64 semantics must be clean."""
import os
import shutil
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_oracle_is_asan_ubsan_clean(tmp_path):
    exe = tmp_path / "sanitize_driver"
    srcs = [os.path.join(ORACLE, f) for f in ("sanitize_driver.cpp", "fsgm_oracle_epi.cpp", "fsgm_oracle_pyd.cpp", "fsgm_oracle_ng.cpp",
                                                "fsgm_oracle_pyramid.cpp", "fsgm_oracle_post.cpp", "fsgm_oracle_geometry.cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-ffp-contract=off", "-I", ORACLE, "-o", str(exe)] + srcs)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "asan/ubsan run finished" in out.stdout and "runtime error" not in out.stderr
