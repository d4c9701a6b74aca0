"""The PMC figures bench.py prices its kernels with (`roofline.traffic`, `roofline_valu`, `roofline_top5`, `gcups`) come from
rocprofv3 passes of a separate run: profiles/r6_pmc_per_kernel.json, stamped with a hash of the kernel sources those passes
were taken on.  Round 5 ended with that file one library behind (`traffic_stale: true` in the driver's line): this test fails
whenever lancet2_amd/csrc or include/ changed after the committed counters were taken -- re-run
`gpurun -- bash tools/r6_measure.sh r6_final full` and `bash tools/r6_collect.sh` (tools/, profiles/README.md)."""
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_pmc_counters_are_of_these_kernel_sources():
    from lancet2_amd.stamp import csrc_sha16
    for name in ("r6_pmc_per_kernel.json", "r6_final_pmc_per_kernel.json"):
        prof = json.load(open(os.path.join(REPO, "profiles", name)))
        assert prof["_stamp"]["csrc_sha16"] == csrc_sha16(), (name, prof["_stamp"], csrc_sha16())
    # ... and the bench line committed beside them is the same build's
    line = json.load(open(os.path.join(REPO, "profiles", "r6_final_bench.json")))
    assert line["build"]["csrc_sha16"] == csrc_sha16()
    assert line["roofline"]["traffic_stale"] is False and line["roofline"]["traffic"] is not None
    assert line["gcups"]["read_aligner"]["GCUPS"] > 0 and line["gcups"]["poa_band"]["GCUPS"] > 0
    assert line["parity_sample"]["mismatches"] == 0
