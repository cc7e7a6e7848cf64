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

# SPIRAL_EMU_LONG=1 adds the slower cases (the matrix-core batched pass, four process ranks, a second stream policy); the whole
# parity file and the stream sweep are scripts/emu_full_check.sh
LONG = os.environ.get("SPIRAL_EMU_LONG") == "1"
long_only = pytest.mark.skipif(not LONG, reason="SPIRAL_EMU_LONG=1 (keeps the CPU suite to a few minutes)")
# fast shapes only: the emulation is several thousand times slower than one CU
SUBSET = ("test_params_tables_match or test_ntt_forward_inverse or test_to_ntt_from_ntt or test_from_ntt_small "
          "or test_add_and_scalar_multiply or test_reorient_reg or (test_multiply and not 1024 and not 2048) or test_automorph_and_gadget "
          "or (test_pp_deserialize and fast) or (test_expand_query and fast) or test_coefficient_expansion "
          "or test_fold_pack_encode or (test_process_query_bytes_and_decode and not inst2) or test_process_query_next_rows "
          "or test_fused_fold_kernel or test_bad_lengths_raise "
          # the production kernels of the large configurations: wave-per-transform fold (nine gadget widths), ring-form sweep
          # with batched fold tails, the matrix-core batched pass
          "or (test_wave_fold_kernel_gadget_widths and (0 or 4 or 9 or 13)) or (test_ring_sweep_and_batched_tails_parity and 5-10-4-8-1-256)")
LONG_SUBSET = ("test_wave_fold_kernel_gadget_widths or (test_process_query_batch_matrix_core_sweep and 64x128) or (test_process_query_batch_two_query_tiles and 32x128-B19) "
               "or (test_planar_copy_lifecycle and 64x128) or test_query_path_without_folding_neg")
RACE_SUBSET = ("test_ntt_forward_inverse or test_to_ntt_from_ntt or test_fold_pack_encode or test_fused_fold_kernel "
               "or (test_process_query_bytes_and_decode and (fast-0 or fast56 or nu2_1)) "
               "or (test_wave_fold_kernel_gadget_widths and (0 or 13))")
STREAM_SUBSET = ("(test_process_query_bytes_and_decode and fast56) or (test_ring_sweep_and_batched_tails_parity and 5-10-4-8-1-256) "
                 "or (test_expansion_variants_response_parity and 0-split) or (test_process_query_batch and narrow-3)")
ASAN_SUBSET = ("(test_process_query_bytes_and_decode and (fast-0 or fast56 or nu2_0)) or test_fold_pack_encode or (test_multiply and not 1024 and not 2048) "
               "or (test_wave_fold_kernel_gadget_widths and (0 or 4))")


def _run(lib, expr, extra_env=None, timeout=1500, at_least=5):
    env = dict(os.environ, SPIRAL_HIP_LIB=lib)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-x",
                        "-p", "no:cacheprovider", "-k", expr], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= at_least, tail
    assert "failed" not in r.stdout.splitlines()[-1], tail
    return int(m.group(1))


@pytest.fixture(scope="module")
def emulated():
    so = emu_build.build()
    if so is None:
        pytest.skip("no host clang to build the emulated library with")
    return so


def test_parity_subset_on_the_emulated_device(emulated):
    """(Run with the streams in `starve:1` order rather than eagerly: a correct host pipeline passes under every order, and this is
    the one under which a missing wait between a plane's sweep and its fold fails -- see
    test_host_pipeline_survives_adversarial_stream_orders.)"""
    assert _run(emulated, SUBSET, {"SPIRAL_EMU_STREAMS": "starve:1"}) >= 38


@long_only
def test_group_expansion_on_the_emulated_device(emulated):
    """r06: a batched group's expansions as shared launches (one more grid dimension = the query, per-query buffers behind byte
    offsets: run_begin_group) -- a list of eleven queries on a PACKED database against the oracle, with the streams in an
    adversarial order (the group flow hands work from the leader's stream back to every query's own)."""
    assert _run(emulated, "test_process_query_batch and packed", {"SPIRAL_EMU_STREAMS": "random:11"}, at_least=1) >= 1


@pytest.mark.parametrize("case", ["A", pytest.param("B", marks=long_only), pytest.param("C", marks=long_only)])
def test_out_of_memory_ladder_of_the_batched_call(emulated, case):
    """sp_process_query_batch when device memory runs out inside a group (capi.cpp): the emulator's device-memory
    budget makes hipMalloc fail at a chosen point (tests/_emu_oom_ladder.py).  A: the digit-planar copy is given back and the list
    runs through the PACKED two-tile kernel; B: groups of 8, one at a time; C: one query at a time.  Every response == oracle.
    (Case A is also the default suite's run of a batched group's SHARED expansion launches -- eleven queries, path bit asserted;
    B, C and the adversarial-stream-order run of the group expansion are in the SPIRAL_EMU_LONG set.)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_emu_oom_ladder.py"), case], cwd=ROOT,
                       env=dict(os.environ, SPIRAL_HIP_LIB=emulated), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "oom-ladder-ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


@long_only
def test_parity_of_the_batched_passes_on_the_emulated_device(emulated):
    assert _run(emulated, LONG_SUBSET) >= 11


def test_kernels_stay_inside_their_buffers(emulated):
    if not emu_build.ASAN_RUNTIME:
        pytest.skip("no AddressSanitizer runtime")
    so = emu_build.build(asan=True)
    log = os.path.join(emu_build.BUILD, "asan_report")
    for f in os.listdir(emu_build.BUILD):
        if f.startswith("asan_report"):
            os.remove(os.path.join(emu_build.BUILD, f))
    env = {"LD_PRELOAD": emu_build.ASAN_RUNTIME, "SPIRAL_EMU_SCHEDULE": "random:20260926",   # (and a shuffled work-item order)
           "ASAN_OPTIONS": "detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1:log_path=" + log}
    try:
        _run(so, ASAN_SUBSET, env)
    finally:
        reports = [f for f in os.listdir(emu_build.BUILD) if f.startswith("asan_report")]
        if reports:
            text = open(os.path.join(emu_build.BUILD, reports[0])).read()
            pytest.fail("AddressSanitizer report from the emulated library:\n" + text[:6000])


@long_only
def test_results_do_not_depend_on_the_work_item_order(emulated):
    """The emulator runs the work-items of a workgroup one after the other between barriers; here in a shuffled order that changes
    every round (the AddressSanitizer pass above runs that way too).  A kernel that still matches the oracle has no read of LDS (or of a wave's private buffer) that races with
    another work-item's write on these shapes -- a missing __syncthreads or wave barrier shows as a byte difference."""
    _run(emulated, RACE_SUBSET, {"SPIRAL_EMU_SCHEDULE": "random:20260926"})


@long_only
@pytest.mark.parametrize("policy", ["starve:1", "random:7", "starve:2"])
def test_host_pipeline_survives_adversarial_stream_orders(emulated, policy):
    """Streams of the emulated device are queues; with SPIRAL_EMU_STREAMS set nothing runs until the host waits, and then in an
    order as unkind as the program's own event dependencies allow (starve:K: the K-th stream created only runs when no other
    can).  A pipeline with a missing hipStreamWaitEvent passes on a GPU most of the time and fails here every time -- checked
    by removing the wait between a plane's sweep and its fold (tests/emu/README.md).  Here: the query paths that use two streams
    (split expansion, per-plane sweeps with the folds beside them, batched fold tails, lists of queries in flight)."""
    _run(emulated, STREAM_SUBSET, {"SPIRAL_EMU_STREAMS": policy}, at_least=4)


@pytest.mark.parametrize("world,name,streams", [(2, "narrow", "starve:2"), (4, "packed", "eager"), (8, "narrow", "eager")])
def test_row_sharded_query_over_process_ranks(emulated, tmp_path, world, name, streams):
    """The library's own multi-GPU path (sp_comm_create / sp_process_query_sharded / sp_process_queries_sharded: comm.cpp's
    reduce-scatter per plane, distributed fold, all-gather) with the ranks as PROCESSES: RCCL is replaced by an independent
    statement of its two collectives over shared memory (tests/emu/emu_rccl.cpp).  On hardware this call sequence has only ever
    met a real RCCL at world size 1.  Eight ranks -- the north star's G -- and the PACKED database at four are part of the default
    suite since round 5 (~20 s each).  (`streams`: the order in which the queued operations of a rank's streams run, see
    test_host_pipeline_survives_adversarial_stream_orders -- comm.cpp orders its two streams with five events per query.)"""
    id_file = str(tmp_path / "comm_id")
    env = dict(os.environ, SPIRAL_HIP_LIB=emulated, SPIRAL_EMU_THREADS="2" if world <= 4 else "1", SPIRAL_EMU_STREAMS=streams)
    if name == "packed":
        # r06: wide row shards split their expansion (odd subtree + GSW side on the second stream, beside the per-plane sweeps and
        # their exchanges); these shapes are below that rule's width, so force it -- with the second stream starved where it hurts
        env.update(SPIRAL_EXPAND_SPLIT="1", SPIRAL_EMU_STREAMS="starve:2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_emu_sharded_rank.py"), str(r), str(world), id_file, name],
                              cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=900)[0])
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    for r, (pr, out) in enumerate(zip(procs, outs)):
        assert pr.returncode == 0, "rank %d:\n%s" % (r, out[-3000:])
    assert "sharded-ok" in outs[0]


@pytest.mark.parametrize("mode", ["query", "sharded"])
def test_c_program_on_the_emulated_device(emulated, tmp_path, oracle_mod, mode):
    """tests/c_abi_smoke.c (plain C, no Python between the host and the library) linked against the emulated build: the same
    run the -m gpu suite does on hardware (tests/test_c_abi_smoke.py), byte-compared with the oracle."""
    import json
    from conftest import FAST
    cfg = dict(FAST, nu_2=7, db_item_size=256)
    o = oracle_mod.Params(cfg)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(41)
    idx = 4242 % o.num_items
    q = cl.generate_query(idx, 42)
    item, db = o.generate_random_db_and_get_item(idx)
    files = {"params.json": json.dumps(cfg).encode(), "pp.bin": pp, "query.bin": q, "db.bin": db.tobytes(),
             "expected.bin": o.process_query(pp, q, db)}
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)
    exe = str(tmp_path / "c_abi_smoke_emu")
    so_dir = os.path.dirname(emulated)
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "c_abi_smoke.c"),
                           "-I", os.path.join(ROOT, "include"), "-L", so_dir, "-l:" + os.path.basename(emulated),
                           "-Wl,-rpath," + so_dir, "-o", exe])
    r = subprocess.run([exe, mode] + [str(tmp_path / f) for f in files], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "== the oracle's" in r.stdout
    if mode == "sharded":
        assert "rccl_in_library" in r.stdout


def test_host_threads_through_the_c_abi(emulated, tmp_path, oracle_mod):
    """Four host threads answer queries at once on shared params / public parameters / database handles while one of them also
    answers a list (tests/emu/host_threads_driver.cpp, no Python in the process) -- what a worker pool does to the library.
    AddressSanitizer build where there is one: a race on the workspace pool or the device state shows as a heap error there."""
    import json
    from conftest import FAST
    asan = bool(emu_build.ASAN_RUNTIME)
    lib = emu_build.build(asan=True) if asan else emulated
    cfg = dict(FAST, nu_2=7, db_item_size=256)
    o = oracle_mod.Params(cfg)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(51)
    idx = 4141 % o.num_items
    q = cl.generate_query(idx, 52)
    item, db = o.generate_random_db_and_get_item(idx)
    files = {"params.json": json.dumps(cfg).encode(), "pp.bin": pp, "query.bin": q, "db.bin": db.tobytes(),
             "expected.bin": o.process_query(pp, q, db)}
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)
    exe = str(tmp_path / "host_threads_driver")
    so_dir = os.path.dirname(lib)
    subprocess.check_call([emu_build.CLANG, "-std=c++17", "-O1", "-pthread"] + (["-fsanitize=address", "-shared-libasan"] if asan else []) +
                          ["-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "emu", "host_threads_driver.cpp"),
                           "-L", so_dir, "-l:" + os.path.basename(lib), "-Wl,-rpath," + so_dir,
                           "-Wl,-rpath," + os.path.dirname(emu_build.ASAN_RUNTIME or so_dir), "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1", SPIRAL_EMU_STREAMS="random:3")
    r = subprocess.run([exe] + [str(tmp_path / f) for f in files] + ["4", "2"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "all equal to the oracle's" in r.stdout, (r.stdout[-1500:], r.stderr[-4000:])


def test_bench_and_smoke_refuse_the_emulated_library(emulated):
    env = dict(os.environ, SPIRAL_HIP_LIB=emulated)
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode != 0 and "host emulation" in (r.stdout + r.stderr)
