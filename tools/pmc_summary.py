#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output per kernel (mean per dispatch).

    python tools/pmc_summary.py gpurun_out/pmc1 [gpurun_out/pmc2 ...] > profiles/rNN_pmc.txt
FETCH_SIZE is doubled (gfx950 counts 128-B read requests at 64 B: MI355X_MICROARCH.md, HBM section);
FETCH/WRITE_SIZE are reported by rocprofv3 in KiB.
"""
import collections
import csv
import os
import sys


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


def main(dirs):
    for d in dirs:
        agg = load(d)
        print("== %s (mean per dispatch)" % d)
        for k in sorted(agg, key=lambda k: -sum(agg[k]["_dur_us"])):
            c = {n: sum(v) / len(v) for n, v in agg[k].items()}
            n = len(agg[k]["_dur_us"])
            line = "%-34s n=%-4d dur=%8.1fus" % (k[:34], n, c.pop("_dur_us"))
            if "GRBM_GUI_ACTIVE" in c:
                line += "  clk=%.2fGHz" % (c["GRBM_GUI_ACTIVE"] / (c0 := agg[k]["_dur_us"] and (sum(agg[k]["_dur_us"]) / n)) / 1e3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
                # MFMA busy is summed over SIMDs (4/CU, 1024 total); GRBM_GUI_ACTIVE is chip cycles
                line += "  mfma_busy=%.1f%%" % (100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 1024))
            if "SQ_WAVE_CYCLES" in c:
                wc = c["SQ_WAVE_CYCLES"]
                line += "  wait_any=%.0f%% wait_inst=%.0f%% active=%.0f%%" % (
                    100 * c["SQ_WAIT_ANY"] / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc, 100 * c["SQ_ACTIVE_INST_ANY"] / wc)
                line += "  lds_conflict=%.1f%%" % (100 * c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1))
            if "FETCH_SIZE" in c:
                line += "  HBM_read=%.1f MB (2x FETCH_SIZE)" % (2 * c["FETCH_SIZE"] * 1024 / 1e6)
            if "WRITE_SIZE" in c:
                line += "  HBM_write=%.1f MB" % (c["WRITE_SIZE"] * 1024 / 1e6)
            print(line)


if __name__ == "__main__":
    main(sys.argv[1:])
