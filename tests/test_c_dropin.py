"""A plain C program that includes <defl/infl.h> and links libdefl_amd.so in place of the
reference's library (tests/c_dropin/dropin_check.c): built with gcc on the CPU, run on the GPU over
the golden file pairs with the reference's own test flow (one-shot, 1/2/4/8-byte pieces, streaming
feeds, error and regression vectors)."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CDIR = os.path.join(HERE, "c_dropin")
EXE = os.path.join(CDIR, "dropin_check")
DATA = os.path.join(HERE, "golden", "defl_testdata")


def _build():
    subprocess.check_call(["make", "-s", "-C", CDIR])


def test_c_caller_compiles_and_links_against_the_library(engine):
    """gcc -std=c11 -Wall -Wextra on a caller of the public headers, linked with -ldefl_amd: every
    entry point the program uses resolves (no device needed)."""
    _build()
    assert os.path.exists(EXE)
    out = subprocess.run(["ldd", EXE], capture_output=True, text=True).stdout
    assert "libdefl_amd.so" in out and "not found" not in out.split("libdefl_amd.so")[1].split("\n")[0]


def test_include_cost_program_runs():
    """tests/c_dropin/include_cost.c: infl_include() timed from C (fresh streams and reused ones); host only."""
    import json
    _build()
    p = subprocess.run([os.path.join(CDIR, "include_cost"), "256", "48000"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-1000:]
    r = json.loads(p.stdout)
    assert r["streams"] == 256 and r["bytes"] == 256 * 48000 and r["include_calls"] > 256 * 6
    assert 0 < r["ns_per_include_reused"] < 1e6 and 0 < r["ns_per_include"] < 1e6


@pytest.mark.gpu
def test_c_caller_decodes_the_golden_pairs(engine):
    if not os.path.exists(EXE):          # (built on the CPU beforehand; the GPU box only runs it)
        _build()
    p = subprocess.run([EXE, DATA, "10"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    assert "dropin ok" in p.stdout
    assert "139 file pairs" in p.stdout


@pytest.mark.gpu
def test_four_host_threads_on_distinct_streams(engine):
    """tests/c_dropin/threads_check.c: four host threads, each looping infl_buf, infl_include + infl_many, infl_stream,
    infl_batch and infl_batch_noalloc over the golden pairs on streams of its own; the program stops at the first result
    that differs from the raw twin."""
    exe = os.path.join(CDIR, "threads_check")
    if not os.path.exists(exe):
        _build()
    p = subprocess.run([exe, DATA, "4", "2"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    assert "threads ok: 4 threads" in p.stdout
