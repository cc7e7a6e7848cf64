"""Compile-time guard on the query path's kernels (no GPU needed: hipcc cross-compiles for gfx950): scratch bytes per lane and
register counts from the code object metadata of fold.hip / ntt.hip / sweep.hip.

Why: round 4 found that k_fold_wave had carried 316 bytes of scratch per lane for three rounds -- ~0.9 GB of spill stores per C2 query,
most of the kernel's measured 5x write amplification -- because of ONE array declared one scope too high
(profiles/r04_fold_wave_scratch.md).  Nothing about a spill shows in a parity test; it shows here."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"
CSRC = os.path.join(ROOT, "sdk_amd", "csrc")
# kernel name fragment -> (max scratch bytes per lane, max VGPRs).  The limits are what the round-4 tree compiles to, with a little
# room: a change that crosses one should be a decision, not an accident.
LIMITS = {
    # (k_fold_wave: no scratch at all since the round-5 compose phase; a run-time choice between two halves of a register array
    # there once put the whole array into scratch -- 144 bytes per lane -- and took the change's gain with it)
    "fold.hip": {"k_fold_waveILi1E": (8, 256), "k_fold_waveILi2E": (8, 256), "k_fold_waveILi4E": (8, 256),
                 # k_fold_fused2 (fallback for gadget widths the wave kernel does not take): 12 bytes since the lazy inverse butterfly
                 # (r05, timed: profiles/r05_call1_ab.md), three dwords outside its transform loops
                 "k_fold_fusedE": (0, 256), "k_fold_fused2E": (16, 256),
                 # r06: the grouped expansion's wave-per-digit round (four waves per workgroup: one wave per SIMD, 256 VGPRs each)
                 "k_expand_waveE": (0, 256)},
    "ntt.hip": {"k_from_sweep4E": (0, 256), "k_ntt_invE": (0, 128), "k_ntt_fwdE": (0, 128), "k_ntt_fwd3E": (0, 128),
                "k_expand_roundE": (0, 168), "k_expand_round_groupE": (0, 168), "k_ntt_inv_groupE": (0, 128),
                "k_ntt_fwd3_groupE": (0, 128)},
    "sweep.hip": {"k_sweep_packed_ringILi8E": (0, 256), "k_sweep_packed_ringILi4E": (0, 256), "k_sweep_packed_ringILi2E": (0, 256),
                  "k_sweep_packed_persistE": (0, 256), "k_sweep_wideE": (0, 256),
                  # the batched passes on the matrix cores over the PACKED words.  The two-tile form (16 queries, one wave per SIMD,
                  # 256 VGPRs + 256 AGPRs; since r05 the FALLBACK of groups of 9 .. 16: databases with no room for the digit-planar
                  # copy -- C3, C4 -- or a first dimension that is not whole 64-row blocks) carried 200 bytes of scratch in r04-r05:
                  # 24 loop-invariant store addresses and 16 offset terms of the chunk epilogue, formed before the chunk loop and
                  # parked.  r06: the epilogue recomputes them per chunk from an opaque copy of the lane number, the offsets wait in
                  # LDS: one dword is left in the deepest ring (the lane number itself), none in the others
                  "k_sweep_mfma_batchILi2ELi2ELi0ELi1E": (0, 256), "k_sweep_mfma_batchILi8ELi1ELi0ELi2E": (8, 512),
                  "k_sweep_mfma_batchILi4ELi1ELi0ELi2E": (0, 512), "k_sweep_mfma_batchILi2ELi1ELi0ELi2E": (0, 512)},
    # r05: the 9 .. 16-query pass over the digit-planar copy -- one modulus per pass keeps its accumulators in the vector
    # registers: no scratch, and the eight-wave form (two waves per SIMD) stays under 256
    "sweep_planar.hip": {"k_sweep_planarILi4ELi2ELi0ELi1ELi8E": (0, 256), "k_sweep_planarILi2ELi2ELi0ELi1ELi8E": (0, 256),
                         "k_sweep_planarILi4ELi2ELi0ELi1ELi4E": (0, 512), "k_sweep_planarILi2ELi2ELi0ELi1ELi4E": (0, 512)},
}


def _metadata(src):
    r = subprocess.run([HIPCC, "-x", "hip", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", "-",
                        os.path.join(CSRC, src)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = {}
    for block in r.stdout.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        scratch = re.search(r"\.private_segment_fixed_size:\s+(\d+)", block)
        vgpr = re.search(r"\.vgpr_count:\s+(\d+)", block)
        if name and scratch and vgpr:
            out[name.group(1)] = (int(scratch.group(1)), int(vgpr.group(1)))
    return out


@pytest.mark.skipif(shutil.which(HIPCC) is None, reason="no hipcc")
@pytest.mark.parametrize("src", sorted(LIMITS))
def test_query_path_kernels_do_not_spill(src):
    meta = _metadata(src)
    assert meta, "no kernel metadata parsed from " + src
    seen = set()
    for kernel, (scratch, vgprs) in meta.items():
        for frag, (max_scratch, max_vgprs) in LIMITS[src].items():
            if ("spiral" in kernel and frag in kernel):
                seen.add(frag)
                assert scratch <= max_scratch, "%s: %d bytes of scratch per lane (limit %d)" % (kernel, scratch, max_scratch)
                assert vgprs <= max_vgprs, "%s: %d VGPRs (limit %d)" % (kernel, vgprs, max_vgprs)
    assert seen == set(LIMITS[src]), "kernels not found in %s: %s" % (src, sorted(set(LIMITS[src]) - seen))
