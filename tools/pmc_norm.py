"""Normalised figures from tools/pmc_kernel.sh summaries: python tools/pmc_norm.py <pmc.txt> [<pmc.txt> ...]
MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)  (per dispatch means; GRBM_GUI_ACTIVE is summed
over the 8 XCDs, the SQ counters over all SIMDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles of resident waves)."""
import re
import sys

rows = {}
for path in sys.argv[1:]:
    cur = None
    for line in open(path):
        m = re.match(r"\('(.+?)', '(\d+)', '(\d+)'\) dispatches (\d+)", line)
        if m:
            cur = f"{m.group(1)[:48]} grid={m.group(2)}"
            rows.setdefault(cur, {})
            continue
        m = re.match(r"\s+(\w+)\s+([\d.]+)", line)
        if m and cur:
            rows[cur].setdefault(m.group(1), float(m.group(2)))
for k, c in rows.items():
    out = []
    if "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        out.append(f"kernel cycles {cyc / 1e3:9.1f} k")
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            out.append(f"MFMA busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):6.3f}")
    if c.get("SQ_WAVE_CYCLES"):
        w = c["SQ_WAVE_CYCLES"]
        for name, key in (("issuing", "SQ_ACTIVE_INST_ANY"), ("wait-any", "SQ_WAIT_ANY"), ("wait-inst", "SQ_WAIT_INST_ANY"), ("lds-wait", "SQ_WAIT_INST_LDS")):
            if key in c:
                out.append(f"{name} {c[key] / w:5.3f}")
    if c.get("SQ_LDS_BANK_CONFLICT") is not None and c.get("SQ_LDS_IDX_ACTIVE"):
        out.append(f"lds conflicts {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:5.3f}")
    if out:
        print(f"{k:72s} " + "  ".join(out))
