"""The checker must not depend on the host compiler: oracle/fgnn_oracle.c built by gcc (the build every other test uses) and by the
ROCm toolchain's host clang (both -O2 -ffp-contract=off -mfma) must compute the same bits — shared math and RNG probes on slices of
every domain, Philox noise, BP4 under the three check-node rules and both log-sum-exp forms, the feedback-GNN sandwich in both
associations.  Together with tests/test_gpu_math_bits.py (hipcc/gfx950 == gcc/x86 on every input) three compilers agree on the
float32 operation sequence of feedback_gnn_amd/csrc/fgnn_math.h."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang"


def _digests(lib_path=None):
    env = dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    if lib_path:
        env["FGNN_ORACLE_LIB_PATH"] = lib_path
    else:
        env.pop("FGNN_ORACLE_LIB_PATH", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "oracle_digest_driver.py")], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, env=env, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    return json.loads(res.stdout.strip().splitlines()[-1])


def test_gcc_and_clang_builds_of_the_oracle_compute_the_same_bits():
    if not os.path.exists(CLANG):
        pytest.skip("no host clang in this image")
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libfgnn_oracle_clang.so")
    res = subprocess.run([CLANG, "-O2", "-ffp-contract=off", "-mfma", "-fopenmp", "-fPIC", "-Wno-unused-function", "-shared",
                          os.path.join(ROOT, "oracle", "fgnn_oracle.c"), "-o", lib, "-lm"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(2) as pool:  # the two digests side by side (each is mostly single-threaded: slices with denormal inputs are slow on x86)
        fg, fc = pool.submit(_digests), pool.submit(_digests, lib)
        gcc, clang = fg.result(), fc.result()
    assert len(gcc) >= 40 and sorted(gcc) == sorted(clang)
    diff = [k for k in gcc if gcc[k] != clang[k]]
    assert not diff, diff
