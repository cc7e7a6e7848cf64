"""The library's sources on an emulated device: parity tests of the KERNEL CODE on the CPU.

tests/emu/ builds sdk_amd/csrc unchanged for the host (a stand-in <hip/hip_runtime.h>: workgroups as fibers, barriers and
wave exchanges as scheduling points, device memory = host memory, gfx950 builtins restated in C) into
tests/emu/_build/libspiral_emu.so with the C ABI of libspiral_hip.so.  The tests below run a subset of the `-m gpu` parity
tests (tests/test_gpu_parity.py, byte comparisons with the oracle) against that build in a child process
(SPIRAL_HIP_LIB selects the library file), once more under AddressSanitizer, where every device buffer is a heap block with
red zones: an out-of-bounds read or write of any kernel on these shapes is an error, which no GPU run can show.

This is test infrastructure.  It is not a fallback: sdk_amd never builds or loads it, bench.py and smoke() refuse it, and the
`-m gpu` tests on a real MI355X remain the parity tests proper (the emulation knows nothing about gfx950 code generation,
memory ordering between workgroups, or time).
"""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
import build_emulated_library as emu_build  # noqa: E402

# fast shapes only: the emulation is several thousand times slower than one CU
SUBSET = ("test_params_tables_match or test_ntt_forward_inverse or test_to_ntt_from_ntt or test_from_ntt_small "
          "or test_add_and_scalar_multiply or test_reorient_reg or test_multiply or test_automorph_and_gadget "
          "or (test_pp_deserialize and fast) or (test_expand_query and fast) or test_coefficient_expansion "
          "or test_fold_pack_encode or (test_process_query_bytes_and_decode and not inst2) or test_process_query_next_rows "
          "or test_fused_fold_kernel or test_bad_lengths_raise")
ASAN_SUBSET = "(test_process_query_bytes_and_decode and (fast-0 or fast56 or nu2_0)) or test_fold_pack_encode or test_multiply"


def _run(lib, expr, extra_env=None, timeout=1500):
    env = dict(os.environ, SPIRAL_HIP_LIB=lib)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-x",
                        "-p", "no:cacheprovider", "-k", expr], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 5, tail
    assert "failed" not in r.stdout.splitlines()[-1], tail
    return int(m.group(1))


@pytest.fixture(scope="module")
def emulated():
    so = emu_build.build()
    if so is None:
        pytest.skip("no host clang to build the emulated library with")
    return so


def test_parity_subset_on_the_emulated_device(emulated):
    assert _run(emulated, SUBSET) >= 25


def test_kernels_stay_inside_their_buffers(emulated):
    so = emu_build.build(asan=True)
    if not os.path.exists(emu_build.ASAN_RUNTIME):
        pytest.skip("no AddressSanitizer runtime")
    log = os.path.join(emu_build.BUILD, "asan_report")
    for f in os.listdir(emu_build.BUILD):
        if f.startswith("asan_report"):
            os.remove(os.path.join(emu_build.BUILD, f))
    env = {"LD_PRELOAD": emu_build.ASAN_RUNTIME,
           "ASAN_OPTIONS": "detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1:log_path=" + log}
    try:
        _run(so, ASAN_SUBSET, env)
    finally:
        reports = [f for f in os.listdir(emu_build.BUILD) if f.startswith("asan_report")]
        if reports:
            text = open(os.path.join(emu_build.BUILD, reports[0])).read()
            pytest.fail("AddressSanitizer report from the emulated library:\n" + text[:6000])


def test_bench_and_smoke_refuse_the_emulated_library(emulated):
    env = dict(os.environ, SPIRAL_HIP_LIB=emulated)
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode != 0 and "host emulation" in (r.stdout + r.stderr)
