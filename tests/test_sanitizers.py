"""Host code under sanitizers, on the CPU (no GPU: tools/hoststub stands in for the HIP runtime, kernels are not run).

The parsers, the host side of the decoder (picture boundaries, DPB, reference lists, slice-group maps, batching, staging
layout, error paths) and the oracle are fed intact and damaged copies of the test matrix under AddressSanitizer + UBSan;
any report makes the program exit non-zero.  Short slices of what tools/host_asan.sh / tools/parser_asan.sh run at length."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_asan():
    if not shutil.which("g++"):
        return False
    r = subprocess.run("echo 'int main(){return 0;}' | g++ -x c++ - -fsanitize=address,undefined -o /dev/null", shell=True, capture_output=True)
    return r.returncode == 0


pytestmark = pytest.mark.skipif(not _has_asan(), reason="g++ with libasan/libubsan not available")


def _run(cmd, tmp_path, timeout):
    env = dict(os.environ, TMPDIR=str(tmp_path), UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    return r.stdout


def test_host_side_under_asan(tmp_path):
    out = _run(["bash", "tools/host_asan.sh", "60", "5"], tmp_path, 600)
    assert "host asan: 60 decoders (0 refused)" in out
    frames = int(out.split(" frames")[0].split()[-1])
    assert frames > 100  # the intact third of the runs decodes: this is not a test of rejects only
    assert " 0:" in out  # (the field-picture streams among the inputs go through the host's field picture management since round 4)


def test_parsers_and_oracle_under_asan(tmp_path):
    out = _run(["bash", "tools/parser_asan.sh", "2000", "3", "4"], tmp_path, 600)
    assert out.count("parser fuzz:") == 7 and out.count("decoded,") == 7
    # the round-3 advisor reproducers (ue(v) values >= 2^31 in the PPS slice-group syntax and the ids) are refused by the parser AND by the
    # public map builder, and maximal Exp-Golomb codes spliced in at every bit position leave no negative field behind
    assert out.count("reproducers: 7 of 7 refused") == 7 and out.count("spliced maximal ue(v) codes") == 7
