"""Developer tool (GPU box, no GPU used): the extract stage of examples/pipeline_driver.cpp, a few builds side by side.
usage: python3 tools/dbg/extract_ab.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

for label, flags in (("default", ()), ("default", ()), ("default", ())):
    r = bench.pipeline_extract_leg(cxx_flags=flags, threads=(1, 4, 8, -1))
    print(label, json.dumps({k: v for k, v in r.items() if k not in ("note", "unit")}))
