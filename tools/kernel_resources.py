#!/usr/bin/env python3
"""Registers, scratch, occupancy and LDS of every kernel of the library, from hipcc's own resource remarks
(-Rpass-analysis=kernel-resource-usage), one compile per translation unit in parallel (CPU only, no GPU needed).
    python tools/kernel_resources.py                 # table of the kernels that use scratch, and the totals
    python tools/kernel_resources.py --all           # every kernel
    python tools/kernel_resources.py --json out.json
tests/test_kernel_resources.py asserts from the same data that no kernel of the hot path spills to scratch."""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cidana-svt-av1_amd")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def collect():
    sys.path.insert(0, PKG)
    import importlib.util
    spec = importlib.util.spec_from_file_location("svt_build", os.path.join(PKG, "build.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    units = [s for s in b.SOURCES if s.endswith(".hip")]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    rows = []
    with tempfile.TemporaryDirectory(prefix="svt_ru_") as td:
        jobs = []
        for u in units:
            err = open(os.path.join(td, os.path.basename(u) + ".txt"), "w")
            cmd = [hipcc] + b.HIPCC_FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(PKG, u), "-o", os.path.join(td, os.path.basename(u) + ".o")]
            jobs.append((u, err, subprocess.Popen(cmd, cwd=PKG, stderr=err, stdout=subprocess.DEVNULL)))
        for u, err, p in jobs:
            rc = p.wait()
            err.close()
            text = open(err.name).read()
            if rc != 0:
                raise RuntimeError(f"hipcc failed on {u}:\n{text[-2000:]}")
            pat = (r"Function Name: (\S+).*?SGPRs: (\d+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
                   r"Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)")
            for m in re.finditer(pat, text, re.S):
                name, sg, vg, ag, scr, occ, lds = m.groups()
                rows.append({"unit": u, "mangled": name, "sgprs": int(sg), "vgprs": int(vg), "agprs": int(ag), "scratch": int(scr), "occupancy": int(occ), "lds": int(lds)})
    dm = demangle([r["mangled"] for r in rows])
    for r in rows:
        r["kernel"] = dm[r["mangled"]]
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--json")
    a = ap.parse_args()
    rows = collect()
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)
    shown = rows if a.all else [r for r in rows if r["scratch"]]
    for r in sorted(shown, key=lambda r: (-r["scratch"], r["kernel"])):
        print(f"scratch {r['scratch']:4d} B  vgpr {r['vgprs']:3d}  agpr {r['agprs']:3d}  occ {r['occupancy']}  lds {r['lds']:6d}  {r['kernel'][:170]}")
    print(f"{len(rows)} kernels, {sum(1 for r in rows if r['scratch'])} with scratch")


if __name__ == "__main__":
    main()
