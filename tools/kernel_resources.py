#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table of the HIP translation units, from the compiler's own
-Rpass-analysis=kernel-resource-usage remarks (cross-compiles for gfx950; needs no GPU).

  python tools/kernel_resources.py [file.hip ...] [-DX=1 ...] [--md]

Default: every .hip under chord_amd/csrc.  --md prints a markdown table (what profiles/rNN_kernel_resources.md holds)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chord_amd import build as B  # noqa: E402


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True).stdout.split("\n")
        return [o if o else n for o, n in zip(out, names)]
    except OSError:
        return names


def resources(path, defines=()):
    cmd = [B.HIPCC] + B.COMMON + B.DEVICE + list(defines) + ["-x", "hip", "-c", path, "-o", "/dev/null",
                                                             "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.split("\n"):
        m = re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass-analysis", line) or re.search(r"remark: (.*?) \[-Rpass-analysis", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    return rows


def main():
    argv = sys.argv[1:]
    md = "--md" in argv
    defines = tuple(a for a in argv if a.startswith("-D"))
    files = [a for a in argv if not a.startswith("-")]
    if not files:
        files = sorted(os.path.join(B.CSRC, f) for f in os.listdir(B.CSRC) if f.endswith(".hip"))
    for f in files:
        rows = resources(f, defines)
        names = demangle([r["name"] for r in rows])
        if md:
            print("\n`%s`%s\n" % (os.path.relpath(f, ROOT), (" " + " ".join(defines)) if defines else ""))
            print("| kernel | VGPRs | SGPRs | scratch B/lane | VGPR spills | LDS B/block | waves/SIMD |")
            print("|---|---|---|---|---|---|---|")
        else:
            print("==", os.path.relpath(f, ROOT), " ".join(defines))
        for r, n in zip(rows, names):
            n = re.sub(r"chord::", "", n)
            n = re.sub(r"\(.*\)$", "", n)
            vals = (r.get("VGPRs", "?"), r.get("TotalSGPRs", "?"), r.get("ScratchSize [bytes/lane]", "?"), r.get("VGPRs Spill", "?"),
                    r.get("LDS Size [bytes/block]", "?"), r.get("Occupancy [waves/SIMD]", "?"))
            if md:
                print("| `%s` | %s | %s | %s | %s | %s | %s |" % ((n,) + vals))
            else:
                print("  %-58s vgpr %-4s sgpr %-4s scratch %-4s vspill %-4s lds %-6s occ %s" % ((n,) + vals))


if __name__ == "__main__":
    main()
