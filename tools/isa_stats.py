#!/usr/bin/env python3
"""Per-kernel instruction statistics of a gfx950 assembly file (hipcc --save-temps): registers, spills, LDS, and counts of
the instruction classes that matter here (MFMA, packed / scalar FMA, transcendental, vector memory, LDS, waits).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -c X.hip -o /tmp/X.o --save-temps=obj && python tools/isa_stats.py /tmp/X-hip-amdgcn-amd-amdhsa-gfx950.s [filter]"""
import re
import subprocess
import sys

CLASSES = [("mfma", r"\bv_mfma_"), ("pk_fma", r"\bv_pk_fma_f32"), ("pk_other", r"\bv_pk_(?!fma_f32)"), ("fma", r"\bv_fma(c|_f32|ak|mk)?_?f?3?2?\b|\bv_fmac_f32|\bv_fma_f32"),
           ("trans", r"\bv_(exp|rcp|log|rsq|sqrt)_f32"), ("valu", r"^\s+v_(?!mfma)"), ("vmem_ld", r"\b(global|buffer)_load"), ("vmem_st", r"\b(global|buffer)_store"),
           ("ds_rd", r"\bds_read|\bds_load"), ("ds_wr", r"\bds_write|\bds_store"), ("s_load", r"\bs_load|\bs_buffer_load"), ("waitcnt", r"\bs_waitcnt"),
           ("barrier", r"\bs_barrier"), ("scratch", r"\bscratch_")]


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


def main():
    s = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\s*\.end_amdhsa_kernel", s, re.S | re.M):
        name, body = demangle(m.group(1)), m.group(2)
        if flt and flt not in name:
            continue
        code = body.split(".amdhsa_kernel")[0]
        g = lambda k: (re.search(r"\.amdhsa_%s (\d+)" % k, body) or [None, "?"])[1]       # noqa: E731
        acc = re.search(r"\.amdhsa_accum_offset (\d+)", body)
        print(name.replace("se::", "")[:100])
        print("   vgpr %s (accum_offset %s)  sgpr %s  lds %s  scratch %s" % (g("next_free_vgpr"), acc.group(1) if acc else "-", g("next_free_sgpr"),
              g("group_segment_fixed_size"), g("private_segment_fixed_size")))
        print("   " + "  ".join("%s %d" % (k, len(re.findall(p, code, re.M))) for k, p in CLASSES))


if __name__ == "__main__":
    main()
