#!/usr/bin/env python3
"""Register / scratch / occupancy table of every gfx950 kernel in parallelnbody_amd/csrc, from
`hipcc -Rpass-analysis=kernel-resource-usage` (the compiler's own remarks; no GPU needed).

    python tools/kernel_resources.py [--out profiles/rNN_kernel_resources.txt] [file.hip ...]

`make -C parallelnbody_amd/csrc resources` runs it over all sources with the Makefile's flags.
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "parallelnbody_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = os.environ.get(
    "HIPFLAGS", "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize").split()
FIELDS = ["VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]",
          "LDS Size [bytes/block]"]


def demangle(names):
    filt = "c++filt"
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    short = []
    for d in out:
        d = re.sub(r"\(anonymous namespace\)::", "", d)
        d = re.sub(r"^void ", "", d)
        d = re.sub(r"\(.*$", "", d)            # drop the argument list
        d = d.replace("nbody::", "")
        short.append(d)
    return short


def remarks(src, extra):
    cmd = [HIPCC] + FLAGS + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull]
    err = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+: +(.+?) \[-Rpass-analysis", line) or \
            re.search(r"remark: +(.+?) \[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:"):
            cur = {"name": body.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.split(":", 1)
            cur[k.strip()] = v.strip()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--extra", default="", help="extra compiler flags (e.g. -DNBODY_SYM_UNROLL4=4)")
    ap.add_argument("files", nargs="*")
    args = ap.parse_args()
    files = args.files or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    lines = ["# hipcc -Rpass-analysis=kernel-resource-usage, flags: " + " ".join(FLAGS + args.extra.split()),
             f"# {'kernel':<78} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch B/lane':>15} {'waves/SIMD':>11} {'LDS B':>7}"]
    spills = 0
    for f in files:
        rows = remarks(f, args.extra.split())
        names = demangle([r["name"] for r in rows]) if rows else []
        lines.append(f"## {f}")
        for r, nm in zip(rows, names):
            v = [r.get(k, "?") for k in FIELDS]
            if v[3] not in ("0", "?"):
                spills += 1
            lines.append(f"{nm[:80]:<80} {v[0]:>5} {v[1]:>5} {v[2]:>5} {v[3]:>15} {v[4]:>11} {v[5]:>7}")
    lines.append(f"# kernels with scratch: {spills}")
    text = "\n".join(lines) + "\n"
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
