#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output per kernel (mean per dispatch).

    python tools/pmc_summary.py [--json profiles/pmc_traffic.json] gpurun_out/pmc1 gpurun_out/pmc2 ... > profiles/rNN_pmc.txt

Corrections (MI355X_MICROARCH.md): FETCH_SIZE is doubled (gfx950 counts 128-B read requests at 64 B);
FETCH/WRITE_SIZE are in KiB; GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES is summed
over the 1024 SIMDs.  --json writes {kernel label: HBM bytes per launch (read+write)} for bench.py's
roofline.traffic (launch-weighted mean over the kernels that share a label).
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sketchedit_amd.kernel_labels import KERNEL_LABELS as LABELS  # noqa: E402


def load(d):
    cc = [f for f in os.listdir(d) if f.endswith("counter_collection.csv")][0]
    kt = [f for f in os.listdir(d) if f.endswith("kernel_trace.csv")][0]
    dur = {}
    for r in csv.DictReader(open(os.path.join(d, kt))):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(os.path.join(d, cc))):
        k = r["Kernel_Name"].split("(")[0].replace("void se::", "").replace("se::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen[k]:
            seen[k].add(r["Dispatch_Id"])
            agg[k]["_dur_us"].append(dur.get(r["Dispatch_Id"], 0.0))
    return agg


def main(argv):
    out_json = None
    if argv and argv[0] == "--json":
        out_json, argv = argv[1], argv[2:]
    traffic = collections.defaultdict(float)      # label -> bytes summed over launches (read + write passes)
    launches = collections.defaultdict(lambda: collections.defaultdict(int))   # label -> pass dir -> launches
    for d in argv:
        agg = load(d)
        print("== %s (mean per dispatch)" % d)
        for k in sorted(agg, key=lambda k: -sum(agg[k]["_dur_us"])):
            c = {n: sum(v) / len(v) for n, v in agg[k].items()}
            n = len(agg[k]["_dur_us"])
            dur_us = c.pop("_dur_us")
            line = "%-34s n=%-4d dur=%8.1fus" % (k[:34], n, dur_us)
            if "GRBM_GUI_ACTIVE" in c:
                xcd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0
                line += "  clk=%.2fGHz" % (xcd_cycles / dur_us / 1e3)
                if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                    line += "  mfma_busy=%.1f%%" % (100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / xcd_cycles)
            if "SQ_WAVE_CYCLES" in c:
                wc = c["SQ_WAVE_CYCLES"]
                line += "  wait_any=%.0f%% wait_inst=%.0f%% active=%.0f%%" % (
                    100 * c["SQ_WAIT_ANY"] / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc, 100 * c["SQ_ACTIVE_INST_ANY"] / wc)
                line += "  lds_conflict=%.1f%%" % (100 * c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1))
            lab = next((v for p, v in LABELS.items() if k.startswith(p)), None)
            if "FETCH_SIZE" in c:
                b = 2 * c["FETCH_SIZE"] * 1024
                line += "  HBM_read=%.1f MB (2x FETCH_SIZE)" % (b / 1e6)
                if lab:
                    traffic[lab] += b * n
                    launches[lab][d] += n
            if "WRITE_SIZE" in c:
                b = c["WRITE_SIZE"] * 1024
                line += "  HBM_write=%.1f MB" % (b / 1e6)
                if lab:
                    traffic[lab] += b * n
                    launches[lab][d] += n
            print(line)
    if out_json:
        json.dump({"unit": "bytes per launch (HBM read + write, rocprofv3 PMC, FETCH_SIZE x2)",
                   "kernels": {k: round(v / max(max(launches[k].values()), 1)) for k, v in traffic.items()}},
                  open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
