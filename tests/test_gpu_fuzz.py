"""Randomised sizes end to end (tools/fuzz_sizes.py): batch 1-3, heights / widths 16-152 in steps of 8, fp32 and bf16,
default and low-latency execution, one of three procedural weight sets per case (synth.WEIGHT_SETS), every case against
the oracle (fp32: 1e-3, bf16: 3e-2 vs the oracle's bf16 mode, or the error triangle where that is wider)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_random_sizes_vs_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("se_fuzz_sizes", os.path.join(root, "tools", "fuzz_sizes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, worst = mod.run(60, 2024, verbose=False)
    assert bad == 0
    # bf16: every case is bounded inside run() -- 3e-2, or 1.25 x the bf16 oracle's own distance from the fp32 oracle where
    # that is larger (the larger-gain weight set w1) -- so only a sanity ceiling here
    assert worst["f32"] < 1e-4 and worst["bf16"] < 1e-1


@pytest.mark.timeout(900)
def test_random_layer_shapes_vs_oracle():
    """tools/fuzz_ops.py: every gated-conv shape of the networks at random sizes (ragged tiles, odd grids), dilations,
    both precisions, both launch shapes."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("se_fuzz_ops", os.path.join(root, "tools", "fuzz_ops.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, worst = mod.run(120, 77, verbose=False)
    assert bad == 0
    assert worst["f32"] < 1e-4
