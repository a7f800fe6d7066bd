"""The committed counter evidence belongs to the kernels that ship.

bench.py fills `roofline.traffic` / `roofline.valu_issue` from the newest profiles/*_pmc_raster.json / *_pmc_raster_sq.json and refuses a summary
whose `raster_source_hash` is not gvfdiffusion_amd._build.raster_source_hash() (the counters cannot be collected from inside the benchmark
process).  Round 5 changed csrc/rast.hip after its last PMC pass and the driver's line went out with both fields null.  This test makes that
state a red CPU suite: whoever touches the rasteriser sources re-runs `scripts/evidence.sh <tag> raster` (which stamps and copies the two
summaries into profiles/) before the round ends."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("pattern", ["*_pmc_raster.json", "*_pmc_raster_sq.json"])
def test_newest_raster_counter_summary_was_taken_on_the_shipped_sources(pattern):
    from gvfdiffusion_amd._build import raster_source_hash
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    assert files, f"no profiles/{pattern} committed"
    doc = json.load(open(files[-1]))
    assert doc.get("raster_source_hash") == raster_source_hash(), (
        f"{os.path.basename(files[-1])} was collected on other rasteriser sources: bench.py will print roofline.traffic / valu_issue = null.  "
        "Re-run scripts/evidence.sh <tag> raster on the GPU and commit profiles/<tag>_pmc_raster*.json")
    assert "blend_kernel" in " ".join(doc["kernels"]), "the summary does not hold the launch the roofline object is about"
