"""The oracle-side half of __graft_entry__.smoke(), run where there is no GPU: regenerates smoke's inputs and
asserts every precondition smoke states about them (window counts per leg, full-length windows in the
small-step leg).  Round 3 ended red because such a constant was wrong and only a GPU box ever evaluated it."""
import numpy as np

import __graft_entry__ as entry


def test_smoke_preconditions_hold_on_the_oracle_side():
    inp = entry.smoke_inputs()
    assert inp["pos"].size == entry.SMOKE["n"] and np.unique(inp["chr_ids"]).size == entry.SMOKE["chroms"]
    ref = entry.smoke_reference(inp)          # asserts the per-leg window counts itself
    assert ref["fst"].size == ref["het"].size == ref["dxy"].size
    assert ref["fst_small_step"].size > ref["fst"].size
    # fields smoke() compares must exist in the oracle's rows
    for f in ("start", "end", "mid", "n", "value"):
        assert f in ref["fst"].dtype.names
    assert "nskip" in ref["dxy"].dtype.names
    assert {"neff", "nskip"} <= set(ref["dxy_total"].dtype.names if hasattr(ref["dxy_total"], "dtype") else ref["dxy_total"].keys())


def test_smoke_inputs_are_reproducible():
    x, y = entry.smoke_inputs(), entry.smoke_inputs()
    for k in x:
        assert np.array_equal(x[k], y[k]), k
