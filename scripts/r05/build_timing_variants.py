#!/usr/bin/env python
"""TIMING-ONLY builds of the library (wrong results by construction) that each remove ONE thing a fold wave waits for, to measure
what that thing costs k_fold_wave: sdk_amd/variants/libspiral_hip_tv_<name>.so, selected with SPIRAL_HIP_LIB.
  no_operands   the multiply-accumulate operands are constants instead of global loads (FoldMac::fetch)
  no_prologue   the two ciphertexts of a step are computed values instead of global loads
  no_transpose  the wave transform's LDS transposes are register moves (no ds_write / ds_read / waits)
  all           the three together
Only fold.hip is recompiled (from a patched copy of csrc/ under /tmp); every other object is the product's."""
import os
import shutil
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(R, "sdk_amd", "csrc")


def sub(text, old, new, count=1):
    assert text.count(old) >= 1, old
    return text.replace(old, new, count)


def patch(d, name):
    fold = open(os.path.join(d, "fold.hip")).read()
    wave = open(os.path.join(d, "wave_ntt.hpp")).read()
    ntt = open(os.path.join(d, "ntt.hip")).read()
    # k_from_sweep4 (from_ntt of the sweep output): fs_*
    if name == "fs_no_store":   # one store in 64 kept
        ntt = sub(ntt, "          out[tau + 256 * k] = val;", "          if ((tau & 63) == 0 && k == 0) out[tau + 256 * k] = val; else if (val == 0x123456789ULL) out[1] = val;")
    if name == "fs_no_load":    # computed values instead of the strided 16-byte loads
        ntt = sub(ntt, "      uint4 x = *reinterpret_cast<const uint4*>(sp + (size_t)(8 * tau + k) * np);",
                  "      uint4 x = make_uint4((u32)(8 * tau + k) * 2654435761u >> 5, (u32)(tau + k) * 40503u, (u32)g * 97u + k, (u32)tau * 7u + c);")
    if name == "fs_no_inv":     # the four cooperative inverse transforms of a workgroup
        ntt = sub(ntt, "    ntt_inv_block_m<4>(v, tau, lds0, lds1, iw, iw + N, m.q, m.two_q);", "    v[0][0] += (u32)(size_t)iw + lds0[tau];")
    bodies = open(os.path.join(d, "bodies.hpp")).read()
    # k_ntt_fwd / k_ntt_fwd3 (digit transforms of the expansion): fw_*
    if name == "fw_no_load":
        bodies = sub(bodies, "    u64 x = src[tau + 256 * k];\n    u64 piece = (sh >= 64) ? 0ULL : ((x >> sh) & mask);  // gadget.rs:48-53",
                     "    u64 x = ((u64)(tau + 256 * k + o) * 0x9E3779B97F4A7C15ULL) >> 8;\n    u64 piece = (sh >= 64) ? 0ULL : ((x >> sh) & mask);  // gadget.rs:48-53")
    if name == "fw_no_store":
        bodies = sub(bodies, "  dst[0] = make_uint4(v[0], v[1], v[2], v[3]);\n  dst[1] = make_uint4(v[4], v[5], v[6], v[7]);",
                     "  if (v[0] == 0x12345u && v[5] == 77u) { dst[0] = make_uint4(v[0], v[1], v[2], v[3]); dst[1] = make_uint4(v[4], v[5], v[6], v[7]); }")
    if name == "fw_no_transform":
        bodies = sub(bodies, "  ntt_fwd_block(v, tau, ldsA, ldsB, fw, fw + N, m.q, m.two_q);\n  uint4* dst", "  v[0] += ldsA[tau] + (u32)(size_t)fw;\n  uint4* dst")
    open(os.path.join(d, "bodies.hpp"), "w").write(bodies)
    open(os.path.join(d, "ntt.hip"), "w").write(ntt)
    if name in ("no_operands", "all"):
        fold = sub(fold, "    m0[g] = a0[64 * g];\n    m1[g] = a1[64 * g];",
                   "    m0[g] = u32x4w_t{(u32)g + 11u, (u32)g + 12u, (u32)g + 13u, (u32)g + 14u};\n    m1[g] = u32x4w_t{(u32)g + 21u, (u32)g + 22u, (u32)g + 23u, (u32)g + 24u};")
    if name in ("no_prologue", "all"):
        fold = sub(fold, "        x0[j][e] = ct0[n];\n        x1[j][e] = ct1[n];",
                   "        x0[j][e] = ((u64)n * 0x9E3779B97F4A7C15ULL) >> 9;\n        x1[j][e] = ((u64)(n + 5) * 0xC2B2AE3D27D4EB4FULL) >> 9;")
    if name in ("no_transpose", "all"):
        wave = sub(wave, "        buf[144 * pp + hl] = a[i];       // element 128pp + (lane < 32 ? lane : lane + 32) of this half, at n + 4 (n >> 5)\n        buf[144 * pp + 36 + hl] = b[i];  // the element 32 above it",
                   "        v[2 * (4 * grp + i)] = a[i] + (u32)pp;\n        v[2 * (4 * grp + i) + 1] = b[i] + (u32)hl;")
        wave = sub(wave, "          const u32x4w_t t4 = *reinterpret_cast<const u32x4w_t*>(rd + 4 * g);",
                   "          const u32x4w_t t4 = u32x4w_t{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]} + (u32)(size_t)rd;")
    if name in ("no_fwd",):   # the digit transforms themselves: the hooks (operand fetches, multiply-accumulates) stay
        fold = sub(fold, "      wntt_fwd<false>(v, ln, mybuf, fwi, stw, ltw, m.q, m.two_q, hk);",
                   "      hk.before_t4(); hk.before_t1();\n      for (int qq = 0; qq < 4; qq++) hk.after_quarter(qq, v);")
    if name in ("no_inv",):
        fold = sub(fold, "      wntt_inv(rr, lt, mybuf, inv_tables(T.tw, imod), mi.q, mi.two_q);  // -> coefficient 64 k + lane", "      rr[0] += (u32)(size_t)mybuf;")
    if name in ("no_reduce",):   # the cross-wave reduction: every wave keeps its own partial sums
        for w in ("    if (wv == 2) { SP_PUT(r0, 0) SP_PUT(r1, 1) }\n    if (wv == 3) { SP_PUT(r0, 2) SP_PUT(r1, 3) }\n    __syncthreads();\n",
                  "    if (wv == 0) { SP_ADD(r0, 0) SP_ADD(r0, 2) }\n    if (wv == 1) { SP_ADD(r1, 1) SP_ADD(r1, 3) }\n    __syncthreads();\n",
                  "    if (wv == 0) { SP_PUT(r1, 0) }\n    if (wv == 1) { SP_PUT(r0, 1) }\n    __syncthreads();\n",
                  "    if (wv == 0) { SP_ADD(r0, 1) }\n    if (wv == 1) { SP_ADD(r1, 0) }\n"):
            fold = sub(fold, w, "")
    if name in ("no_digits",):   # the prologue's digit extraction and LDS parking (the loads stay)
        fold = sub(fold, "    for (int kd = 0; kd < t_live; kd++) {\n      const int sh = (kd * d.bits) & 63;", "    for (int kd = 0; kd < (int)(x0[0][0] >> 62); kd++) {\n      const int sh = (kd * d.bits) & 63;")
    if name in ("no_stage",):   # the per-modulus staging of the forward tables into LDS (the barrier stays)
        fold = sub(fold, "    wtw_stage(ltw, fw, tau);\n    __syncthreads();              // tables", "    if (d.bits > 64) wtw_stage(ltw, fw, tau);\n    __syncthreads();              // tables")
    if name in ("no_garner",):   # the final compose + add + store of waves 2 and 3
        fold = sub(fold, "      if (wv >= 2) {\n        u64* orow = out + (size_t)irow * N;", "      if (wv >= 2 && d.bits > 64) {\n        u64* orow = out + (size_t)irow * N;")
    if name in ("no_mac",):      # the multiply-accumulates (operands still fetched and touched once)
        fold = sub(fold, "    mac(2 * qq, v);\n    mac(2 * qq + 1, v);", "    acc0[2 * qq] += (u64)m0[2 * qq].x * v[0] + m1[2 * qq].y + m0[2 * qq + 1].z + m1[2 * qq + 1].w;")
    if name in ("no_reduce64",): # the 2 x 64 Barrett reductions of a wave's sums per modulus
        fold = sub(fold, "      r0[k] = reduce64(acc0[k], m);\n      r1[k] = reduce64(acc1[k], m);", "      r0[k] = (u32)acc0[k] & 0x0fffffffu;\n      r1[k] = (u32)acc1[k] & 0x0fffffffu;")
    open(os.path.join(d, "fold.hip"), "w").write(fold)
    open(os.path.join(d, "wave_ntt.hpp"), "w").write(wave)


def main():
    names = sys.argv[1:] or ["no_operands", "no_prologue", "no_transpose", "all"]
    objs = [o for o in sorted(os.listdir(os.path.join(SRC, "build"))) if o.endswith(".o") and o != "fold.hip.o"]
    os.makedirs(os.path.join(R, "sdk_amd", "variants"), exist_ok=True)
    for name in names:
        d = "/tmp/tv_" + name
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        for f in os.listdir(SRC):
            if f.endswith((".hip", ".hpp", ".cpp")):
                shutil.copy(os.path.join(SRC, f), d)
        os.makedirs(os.path.join(d, "..", "include"), exist_ok=True)
        patch(d, name)
        mine = []
        for f in ("fold.hip", "ntt.hip"):
            obj = os.path.join(d, f + ".o")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-x", "hip", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-result",
                                   "-I", os.path.join(R, "sdk_amd", "csrc"), "-c", os.path.join(d, f), "-o", obj], cwd=d)
            mine.append(obj)
        out = os.path.join(R, "sdk_amd", "variants", "libspiral_hip_tv_%s.so" % name)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-o", out] + mine +
                              [os.path.join(SRC, "build", o) for o in objs if o != "ntt.hip.o"] + ["-L/opt/rocm/lib", "-lrccl"])
        print(out)


if __name__ == "__main__":
    main()
