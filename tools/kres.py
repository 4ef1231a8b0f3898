"""Compile-time resources of the gfx950 kernels (hipcc -Rpass-analysis=kernel-resource-usage; works without a GPU):
python tools/kres.py <file.hip> [name substring]  ->  VGPR / AGPR / scratch / spills / occupancy per kernel"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
from dalle_hip import build as b  # noqa: E402


def usage(fname, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        p = subprocess.run([b._hipcc()] + b.FLAGS + list(extra) + ["-I" + os.path.join(ROOT, "include"), "-Rpass-analysis=kernel-resource-usage", "-c",
                            os.path.join(b.CSRC, fname), "-o", os.path.join(tmp, "o.o")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stdout[-4000:])
        raise SystemExit(1)
    out = {}
    for blk in re.split(r"remark: Function Name: ", p.stdout)[1:]:
        g = lambda k: int(re.search(k + r": (\d+)", blk).group(1))   # noqa: E731
        out[blk.split()[0]] = dict(vgpr=g(" VGPRs"), agpr=g("AGPRs"), scratch=g(r"ScratchSize \[bytes/lane\]"), occ=g(r"Occupancy \[waves/SIMD\]"),
                                   sspill=g("SGPRs Spill"), vspill=g("VGPRs Spill"), lds=g(r"LDS Size \[bytes/block\]"))
    return out


if __name__ == "__main__":
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    for k, v in usage(sys.argv[1], sys.argv[3:]).items():
        if sub in k:
            name = subprocess.run(["c++filt", k], stdout=subprocess.PIPE, text=True).stdout.strip()
            print(f"{name[:70]:70s} vgpr {v['vgpr']:3d} agpr {v['agpr']:3d} scratch {v['scratch']:4d} vspill {v['vspill']:3d} sspill {v['sspill']:3d} occ {v['occ']}")
