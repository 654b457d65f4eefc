"""rocprofv3 kernel name -> in-library profiler label (ProfLabel, csrc/se_kernels.h): ONE table, read by bench.py's PMC
traffic pass and by tools/pmc_summary.py, so that every label the profiler prints has its HBM traffic beside it.
Keys are prefixes of the demangled kernel name with the `se::` namespace stripped; the first matching prefix wins, so
longer prefixes come first where one name is the prefix of another."""

KERNEL_LABELS = {
    # 96 -> 192 (N = 192 packed rows)
    "wino_kernel": "wino_n192", "wino24_kernel": "wino_n192", "rconv16": "gconv_n192", "gconv_kernel<12": "gconv_n192",
    # 48 -> 96 / 24 -> 96
    "wino48_kernel": "wino_n96", "rconv96": "gconv_n96", "gconv_kernel<6": "gconv_n96",
    "winoup_kernel": "wino_up96",
    # 48 output rows: 5x5 heads, 24 -> 48 stride 2, gen_deconv 48 -> 48
    "winoup48_kernel": "gconv_n48", "rtile_dense5w_kernel": "gconv_n48", "rtile_dense5_kernel": "gconv_n48",
    "rtile_kernel<3": "gconv_n48", "gconv_kernel<3": "gconv_n48",
    # 24 -> 24 at full resolution
    "rtilew2_kernel": "gconv_n24", "rtilew_kernel": "gconv_n24", "rtile_kernel<2": "gconv_n24", "gconv_kernel<2": "gconv_n24",
    "dtail_kernel": "dtail",
    # attention
    "att2_pair_kernel": "att_score", "att_score_kernel": "att_score", "att2_pv_kernel": "att_pv", "att_pv_kernel": "att_pv",
    "att2_softmax": "att_softmax", "att2_stats": "att_softmax", "att_softmax_kernel": "att_softmax",
    "att2_boxsum": "att_boxsum", "att2_ptilde": "att_boxsum",
    "att2_prep": "att_prep", "att2_transpose": "att_prep", "att2_emean": "att_prep", "att2_eoff": "att_prep", "att_prep_kernel": "att_prep",
    "att2_similar": "layout",
    # the rest
    "small_conv_kernel": "small_conv", "pack_": "pack", "colreduce": "colreduce", "vecbias_kernel": "colreduce",
    "nchw_to_nhwc": "layout", "nhwc_to_nchw": "layout", "nhwc16_to_nchw": "layout", "quantize_u8_kernel": "layout",
}


def kernel_key(name):
    """demangled rocprofv3 Kernel_Name -> the key the table is matched against"""
    return name.split("(")[0].replace("void se::", "").replace("se::", "")


def label_of(name):
    k = kernel_key(name)
    return next((v for pre, v in KERNEL_LABELS.items() if k.startswith(pre)), None)
