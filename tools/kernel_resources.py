#!/usr/bin/env python3
"""Per-kernel resources of the gfx950 code objects inside a library: VGPRs, SGPRs, scratch bytes per lane, static LDS, workgroup size.

    python tools/kernel_resources.py [multiple-object-tracking_amd/libmot_amd.so] [--filter kcf_predict]

Read from the code objects' metadata notes (llvm-readelf), so it needs no GPU and describes exactly what ships.  tests/test_isa_scan.py
pins the numbers the design depends on (the 80-px kernels run two workgroups of 512 threads per CU -> at most 128
VGPRs and no scratch; the association's solver / emulation launch has no scratch)."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_exec0_scan import code_objects            # noqa: E402

READELF = os.environ.get("LLVM_READELF", "/opt/rocm/lib/llvm/bin/llvm-readelf")
_KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except (OSError, subprocess.CalledProcessError):
        return {n: n for n in names}


def kernels(path):
    """{mangled kernel name: {key: int}} over every gfx950 code object of the file"""
    res = {}
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".o") as f:
            f.write(co); f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        # the kernels of the metadata are list items ("  - .agpr_count: ...") whose keys come in alphabetical order
        for item in re.split(r"\n\s+- \.agpr_count:", "\n" + txt)[1:]:
            item = ".agpr_count:" + item
            m = re.search(r"\.name:\s+(\S+)", item)
            if not m: continue
            d = {}
            for k in _KEYS:
                mk = re.search(r"\." + k + r":\s+(\d+)", item)
                if mk: d[k] = int(mk.group(1))
            res[m.group(1)] = d
    return res


def main(argv):
    flt = None
    if "--filter" in argv:
        i = argv.index("--filter"); flt = argv[i + 1]; del argv[i:i + 2]
    path = argv[0] if argv else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multiple-object-tracking_amd", "libmot_amd.so")
    ks = kernels(path); names = demangle(list(ks))
    print(f"{'VGPR':>5} {'SGPR':>5} {'scratch':>8} {'LDS':>7} {'wg':>5}  kernel")
    for n in sorted(ks, key=lambda n: names[n]):
        pretty = re.sub(r"\(anonymous namespace\)::", "", names[n]); pretty = re.sub(r"\(.*$", "", pretty)
        if flt and flt not in pretty: continue
        d = ks[n]
        print(f"{d.get('vgpr_count', 0):5d} {d.get('sgpr_count', 0):5d} {d.get('private_segment_fixed_size', 0):8d} {d.get('group_segment_fixed_size', 0):7d} {d.get('max_flat_workgroup_size', 0):5d}  {pretty}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
