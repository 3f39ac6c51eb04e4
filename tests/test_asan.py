"""SURVEY section 5 (race / memory checking), VERDICT r3 item 9: the oracle's C restatement and the plain-C ABI host under
AddressSanitizer + UndefinedBehaviorSanitizer.  `make -C oracle asan` builds oracle/_build/librl_oracle_asan.so from the same
sources; the oracle test files then run in a child pytest whose python process has libasan preloaded and whose binding loads
that library (RLO_ORACLE_SO).  Any heap overflow, use-after-free or undefined operation in the checker aborts the child.
tests/abi_host/abi_host.c is compiled with the same flags and run up to its device probe (no GPU here: it stops with the
ABI's "no ROCm-capable device" status -- library load, version check, error-string path)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SUITES = ["tests/test_oracle_golden.py", "tests/test_oracle_envs.py", "tests/test_oracle_pins.py",
                 "tests/test_oracle_sumtree.py", "tests/test_oracle_stackframes.py", "tests/test_oracle_heads.py",
                 "tests/test_oracle_explorers.py", "tests/test_oracle_mlp3.py", "tests/test_oracle_acrobot.py"]


def _libasan():
    r = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True)
    p = r.stdout.strip()
    return p if r.returncode == 0 and os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_libasan() is None, reason="gcc has no libasan")
def test_oracle_suites_are_clean_under_asan_and_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, capture_output=True)
    so = os.path.join(ROOT, "oracle", "_build", "librl_oracle_asan.so")
    env = dict(os.environ, RLO_ORACLE_SO=so, LD_PRELOAD=_libasan(),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",  # CPython itself "leaks" by design
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", PYTHONMALLOC="malloc")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + ORACLE_SUITES, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1200)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in r.stderr, tail
    assert " passed" in r.stdout, tail


@pytest.mark.skipif(_libasan() is None, reason="gcc has no libasan")
def test_abi_host_argument_paths_are_clean_under_asan(tmp_path):
    import __graft_entry__ as g

    g.build()
    pkg = os.path.join(ROOT, "reinforcementlearning.jl_amd")
    exe = str(tmp_path / "abi_host_asan.bin")
    cmd = ["gcc", "-std=c99", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-Wall",
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_host", "abi_host.c"), "-o", exe,
           "-L" + os.path.join(pkg, "lib"), "-lrlhip", "-Wl,-rpath," + os.path.join(pkg, "lib"), "-Wl,-rpath,/opt/rocm/lib", "-lm"]
    subprocess.run(cmd, check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe, str(tmp_path / "out.bin")], env=env, capture_output=True, text=True, timeout=300)
    # no GPU in the CPU suite's container: the host stops at its device probe (rlhip_device_count reports the HIP error through
    # the CK macro: exit 2, or 66 for zero devices); on a GPU box it runs through (0)
    assert r.returncode in (0, 2, 66), (r.returncode, r.stderr[-2000:])
    if r.returncode == 2:
        assert "rlhip_device_count" in r.stderr and "no ROCm-capable device" in r.stderr, r.stderr[-2000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
