"""Per-kernel register / LDS / scratch usage of the gfx950 code objects, from the compiler's own metadata (no GPU needed):
    python tools/kernel_resources.py > profiles/r03_kernel_resources.md
Waves per SIMD follow MI355X_MICROARCH.md (512 registers per lane and SIMD, allocation granule 8, at most 8 waves)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chatttsplus_amd.build import CSRC, SOURCES  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", o).replace("void ", "") for o in out[:len(names)]]


def main():
    rows = []
    with tempfile.TemporaryDirectory() as d:
        for src in SOURCES:
            asm = os.path.join(d, src + ".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-kernarg-preload-count=8", "-S",
                                   "--cuda-device-only", "-o", asm, os.path.join(CSRC, src)], stderr=subprocess.DEVNULL)
            text = open(asm).read()
            for blk in text.split("  - .agpr_count:")[1:]:
                def f(key):
                    m = re.search(r"\." + key + r":\s+(\S+)", blk)
                    return m.group(1) if m else "0"
                rows.append((src, f("name"), int(f("vgpr_count")), int(f("sgpr_count")), int(f("group_segment_fixed_size")), int(f("private_segment_fixed_size")),
                             int(f("max_flat_workgroup_size"))))
    names = demangle([r[1] for r in rows])
    print("| file | kernel | block | VGPR+AGPR | waves/SIMD by registers | SGPR | static LDS (B) | scratch (B/lane) |")
    print("|---|---|---|---|---|---|---|---|")
    for (src, _, v, s, lds, scratch, wg), name in sorted(zip(rows, names), key=lambda t: (t[0][0], t[1])):
        alloc = max(8, (v + 7) // 8 * 8)
        print(f"| {src} | `{name[:110]}` | {wg} | {v} | {min(8, 512 // alloc)} | {s} | {lds} | {scratch} |")


if __name__ == "__main__":
    main()
