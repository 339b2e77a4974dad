"""SURVEY section 5 / round-5 verdict item 7: the CPU-side C / C++ of this tree under AddressSanitizer + UndefinedBehaviorSanitizer.
`make -C oracle asan` builds the oracle and the two association models with -fsanitize=address,undefined into oracle/_asan/; the oracle-golden,
model and transform tests then run in a child interpreter with libasan preloaded and MOT_ORACLE_DIR pointing there (tests/dft_ct_host.cpp is
compiled with the same flags through MOT_DFT_CT_FLAGS).  Any report fails the test.  (GPU AddressSanitizer does not exist on this pool: the
kernels are covered by MOT_POISON / MOT_LDS_POISON runs instead)  No GPU, no reference needed."""
import os
import subprocess
import sys

import pytest

import orc

ROOT = orc.ROOT


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_models_and_transform_under_asan_ubsan():
    asan = _runtime("libasan.so")
    if not asan:
        pytest.skip("this gcc has no libasan")
    subprocess.check_call(["make", "-C", orc.ORACLE_SRC_DIR, "asan"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=asan, MOT_ORACLE_DIR=os.path.join(orc.ORACLE_SRC_DIR, "_asan"),
               MOT_DFT_CT_FLAGS="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1",
               # leak checking would report CPython's own arenas; everything else aborts the child on the first finding
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=86", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    tests = [os.path.join(ROOT, "tests", t) for t in ("test_oracle_golden.py", "test_lap_model.py", "test_mk_sparse_model.py", "test_dft_ct.py")]
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + tests, env=env, capture_output=True, text=True, timeout=1500)
    text = out.stdout + out.stderr
    assert "AddressSanitizer" not in text and "runtime error:" not in text, text[-4000:]
    assert out.returncode == 0 and " passed" in out.stdout, text[-4000:]
    # the sanitized libraries were the ones under test, not the plain ones
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, sys.argv[1]); import orc; orc.load_oracle(); print(open('/proc/self/maps').read())",
                            os.path.join(ROOT, "tests")], env=env, capture_output=True, text=True, timeout=300)
    assert "oracle/_asan/libmot_oracle.so" in probe.stdout and "oracle/libmot_oracle.so" not in probe.stdout, probe.stdout[-2000:] + probe.stderr[-2000:]
