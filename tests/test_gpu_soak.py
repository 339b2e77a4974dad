"""Soak of the entry point bench.py times (mot_step_frame_device_ahead) inside the GPU suite (round-4 verdict item 1a / advisor: "loop the test at
least 200 times"): tools/lookahead_soak.py repeats the wrong-announcement scenario of test_device_loop_lookahead_vs_oracle with nothing
synchronised between frames, random host delays and a second context hammering the chip, and compares with the oracle; the 1024-track variant runs
three frames per repetition (the first model updates: where round 5's root cause showed).  The long runs behind the root cause -- 10^5 repetitions over
ten switch variants, with bit-by-bit state compare -- are in profiles/r05_hunt_soak.log (tools/lookahead_soak_matrix.sh)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("args", [["48", "8", "5", "400", "--sparse-checks", "--hammer"], ["300", "6", "4", "120", "--state", "--state-stride", "5", "--hammer"],
                                  ["1024", "0", "0", "400", "--sparse-checks", "--hammer", "--frames", "3"],
                                  # round 6: tie frames committed provisionally, two contexts on the shared emulation stream (the configuration in which a late
                                  # emulation kernel once wiped out the next chain's claim: profiles/r06_prov_soak.log -- that needed ~2,000 repetitions, this is a tripwire)
                                  ["1024", "0", "0", "400", "--sparse-checks", "--hammer", "--frames", "12"]],
                         ids=["48-tracks-9-frames-unsynchronised", "300-tracks-state-compare", "1024-tracks-3-frames-unsynchronised", "1024-tracks-12-frames-provisional-two-contexts"])
def test_lookahead_soak(args, tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lookahead_soak.py")] + args + ["--dump", str(tmp_path)], capture_output=True, text=True, timeout=900)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and lines, out.stdout[-3000:] + out.stderr[-3000:]
    j = json.loads(lines[-1])
    assert j["mismatches"] == 0 and j["reps"] == int(args[3])


def test_helper_grid_beside_the_emulation_stream_soak(tmp_path):
    """Round 6's second memory fault (profiles/r06_prov_soak.log): with the sparse emulation a kernel of its own, LAP_H_DONE could be set between the
    reads of two waves of one HELPER workgroup of the final kernel; wave 0 left, the helper loop's barriers released without it and the rest indexed the
    working matrix with stale LDS.  On the default path the helper grid only runs while the dense-emulation hint is set (one fault in several thousand context
    life cycles); MOT_MUNKRES_HELPERS=1 puts it behind EVERY frame: the library before the fix dies within 0-300 repetitions of this run, three of three
    (profiles/r06_prefix_tripwire.log), the fixed one ran 2,000."""
    env = {k: v for k, v in os.environ.items() if not k.startswith("MOT_")}
    env["MOT_MUNKRES_HELPERS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lookahead_soak.py"), "1024", "0", "0", "500", "--sparse-checks", "--hammer", "--frames", "12", "--dirty",
                          "--dump", str(tmp_path)], env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and lines, out.stdout[-3000:] + out.stderr[-3000:]
    j = json.loads(lines[-1])
    assert j["mismatches"] == 0 and j["reps"] == 500
