#!/usr/bin/env python3
"""VGPR / SGPR / spill / scratch / LDS numbers of every kernel in libgraphop_hip.so, read from the
code object's AMDGPU metadata note (llvm-readelf --notes).  Used by tests/test_abi_and_host.py to
keep the hot kernels off the spill line and by hand when tuning register budgets.

  python tools/kernel_resources.py [pattern]      # demangled-name substring filter
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "custom_op_benchmark_amd", "libgraphop_hip.so")


def _code_objects(lib):
    """Extract the gfx950 code object(s) embedded in the host shared library."""
    out = []
    tmp = tempfile.mkdtemp(prefix="graphop_co_")
    bundle = os.path.join(tmp, "bundle")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin",
                           lib, bundle])
    data = open(bundle, "rb").read()
    # a fat binary = concatenated clang offload bundles, one per translation unit
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), data)]
    for n, s0 in enumerate(starts):
        chunk = data[s0:(starts[n + 1] if n + 1 < len(starts) else len(data))]
        bpath = os.path.join(tmp, "b%d" % n)
        open(bpath, "wb").write(chunk)
        co = os.path.join(tmp, "co%d.o" % n)
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                            "--input=" + bpath, "--output=" + co,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
    return out


def kernel_resources(lib=LIB):
    """-> {demangled kernel name: dict(vgpr, sgpr, spill_vgpr, spill_sgpr, scratch, lds)}"""
    res = {}
    for co in _code_objects(lib):
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
        cur = {}
        for line in notes.splitlines():
            line = line.strip()
            m = re.match(r"[-\s]*\.(\w+):\s+(.*)$", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip().strip("'")
            if k == "agpr_count" and cur.get("name"):   # first key of the next kernel's map
                pass
            cur_key = k
            if cur_key == "name" and "symbol" in cur and "name" in cur:
                cur = {}
            cur[cur_key] = v
            if cur_key == "wavefront_size" or cur_key == "workgroup_processor_mode":
                pass
            if "symbol" in cur and "name" in cur and "vgpr_count" in cur and "vgpr_spill_count" in cur \
                    and "sgpr_count" in cur and "private_segment_fixed_size" in cur and "group_segment_fixed_size" in cur:
                res[cur["name"]] = dict(vgpr=int(cur["vgpr_count"]), sgpr=int(cur["sgpr_count"]),
                                        spill_vgpr=int(cur["vgpr_spill_count"]),
                                        spill_sgpr=int(cur.get("sgpr_spill_count", 0)),
                                        scratch=int(cur["private_segment_fixed_size"]),
                                        lds=int(cur["group_segment_fixed_size"]))
                if cur_key == "wavefront_size":
                    cur = {}
    if not res:
        raise RuntimeError("no kernel metadata found in %s" % lib)
    names = list(res)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True,
                         text=True).stdout.splitlines()
    return {d: res[n] for n, d in zip(names, dem)}


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, r in sorted(kernel_resources().items()):
        if pat in name:
            short = re.sub(r"^void graphop::", "", name)
            short = re.sub(r"\(.*$", "", short)
            print("%-64s vgpr %3d sgpr %3d spill %4d scratch %5d lds %6d" %
                  (short[:64], r["vgpr"], r["sgpr"], r["spill_vgpr"], r["scratch"], r["lds"]))
