"""Pairwise-covering combinations of the MOT_* switches that select among the association tiers, the KCF kernel variants and the frame structures
(round-4 verdict item 9): every PAIR of switch values occurs in at least one of the first twelve combinations below (generated greedily over twelve
factors; the four switches whose off-variant had lost every measurement were deleted in round 5, mot_env.h).  Each combination is one process
(the switches are read once): tools/variant_check.py runs three noisy / tie-heavy streams through the device-resident loop and requires every
frame's live list to equal the oracle's."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

COMBOS = [
    {},
    {'MOT_LAP_DENSE': '1', 'MOT_MK_BATCH': '0', 'MOT_MK_LAZY': '0', 'MOT_LOOKAHEAD': '0', 'MOT_SPLIT_UPDATE': '0', 'MOT_SIDE_RESERVE': '0'},
    {'MOT_LAP_FAST': '0', 'MOT_LAP_DENSE': '0', 'MOT_LAP_TWO_BLOCK': '0', 'MOT_MUNKRES_HELPERS': '1', 'MOT_JOINED_LAUNCH': '0', 'MOT_KCF_K80': '0', 'MOT_DEFER_BLEND': '0', 'MOT_SIDE_RESERVE': '0'},
    {'MOT_JOINED_LAUNCH': '0'},
    {'MOT_LAP_FAST': '0', 'MOT_LAP_DENSE': '0', 'MOT_LAP_TWO_BLOCK': '0', 'MOT_MK_BATCH': '0', 'MOT_MK_LAZY': '0', 'MOT_MUNKRES_HELPERS': '1', 'MOT_LOOKAHEAD': '0', 'MOT_SPLIT_UPDATE': '0', 'MOT_KCF_K80': '0', 'MOT_DEFER_BLEND': '0'},
    {'MOT_LAP_DENSE': '1', 'MOT_MK_BATCH': '0', 'MOT_MK_LAZY': '0', 'MOT_MUNKRES_HELPERS': '1', 'MOT_JOINED_LAUNCH': '0', 'MOT_KCF_K80': '0', 'MOT_DEFER_BLEND': '0'},
    {'MOT_LAP_FAST': '0', 'MOT_LAP_TWO_BLOCK': '0', 'MOT_LOOKAHEAD': '0', 'MOT_SPLIT_UPDATE': '0', 'MOT_KCF_K80': '0'},
    {'MOT_LAP_FAST': '0', 'MOT_LAP_TWO_BLOCK': '0', 'MOT_MK_BATCH': '0', 'MOT_MK_LAZY': '0', 'MOT_MUNKRES_HELPERS': '1', 'MOT_JOINED_LAUNCH': '0', 'MOT_SPLIT_UPDATE': '0', 'MOT_DEFER_BLEND': '0', 'MOT_SIDE_RESERVE': '0'},
    {'MOT_LAP_DENSE': '0', 'MOT_LAP_TWO_BLOCK': '0', 'MOT_MK_BATCH': '0', 'MOT_MUNKRES_HELPERS': '1', 'MOT_LOOKAHEAD': '0'},
    {'MOT_LAP_FAST': '0', 'MOT_LAP_DENSE': '0', 'MOT_MK_LAZY': '0', 'MOT_DEFER_BLEND': '0', 'MOT_SIDE_RESERVE': '0'},
    {'MOT_LAP_FAST': '0', 'MOT_LAP_DENSE': '1', 'MOT_LAP_TWO_BLOCK': '0', 'MOT_MK_BATCH': '0', 'MOT_MUNKRES_HELPERS': '1', 'MOT_JOINED_LAUNCH': '0', 'MOT_LOOKAHEAD': '0', 'MOT_SPLIT_UPDATE': '0', 'MOT_KCF_K80': '0', 'MOT_SIDE_RESERVE': '0'},
    {'MOT_LAP_DENSE': '1', 'MOT_MK_LAZY': '0', 'MOT_MUNKRES_HELPERS': '1', 'MOT_JOINED_LAUNCH': '0', 'MOT_LOOKAHEAD': '0', 'MOT_SPLIT_UPDATE': '0', 'MOT_KCF_K80': '0'},
    # round 6: provisional commits off (frames wait for the emulation) / their swap bits decided by the patch step's dense emulation (test hook), against the
    # switches they can meet (with MOT_LAP_TWO_BLOCK=0, MOT_DEFER_BLEND=0 or MOT_SPLIT_UPDATE=0 there are no provisional commits at all)
    {'MOT_PROV': '0', 'MOT_MK_BATCH': '0', 'MOT_MK_LAZY': '0', 'MOT_JOINED_LAUNCH': '0', 'MOT_SIDE_RESERVE': '0'},
    {'MOT_PROV': '2', 'MOT_MK_LAZY': '0', 'MOT_LOOKAHEAD': '0', 'MOT_KCF_K80': '0'},
    {'MOT_PROV': '2', 'MOT_MK_BATCH': '0', 'MOT_JOINED_LAUNCH': '0', 'MOT_LAP_DENSE': '0'},
    # ... and with the stream-emulation chain on EVERY frame (by default only frames with >= 600 detections take it): small problems, the joined
    # predict + feature launch with shadow items, noisy streams whose emulation commits the frame itself or hands it to the dense emulation
    {'MOT_PROV_MIN_DETS': '0', 'MOT_PROV': '2', 'MOT_MK_LAZY': '0'},
    {'MOT_PROV_MIN_DETS': '0', 'MOT_JOINED_LAUNCH': '0', 'MOT_LOOKAHEAD': '0', 'MOT_MK_BATCH': '0'},
]


@pytest.mark.parametrize("combo", COMBOS, ids=[("default" if not c else "+".join(f"{k[4:]}={v}" for k, v in c.items())) for c in COMBOS])
def test_switch_combination_equals_oracle(combo):
    env = {k: v for k, v in os.environ.items() if not k.startswith("MOT_")}
    env.update(combo)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "variant_check.py")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
