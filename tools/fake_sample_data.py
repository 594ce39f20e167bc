"""Writes a tree shaped like the reference's sample-data repository (Sample-Scalable-CCD-Data: <scene>/frames/*.ply,
<scene>/boxes/*{vf,ee}.json) from GENERATED scenes, to exercise the plumbing of tests/test_reference_constants.py where the
real data is absent:  python tools/fake_sample_data.py /tmp/fake && SCCD_SAMPLE_DATA_DIR=/tmp/fake pytest tests/test_reference_constants.py
The superset tests then run and must pass (the ground truth is every second brute-force overlap); the cloth-ball tests fail
on the first count -- the fake cloth-ball is not the reference's."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-ccd_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import orc  # noqa: E402
from sccd import io as sio  # noqa: E402
from sccd import scenes  # noqa: E402

SCENES = {  # as tests/test_broad_phase.cu:31-65 names them
    "armadillo-rollers": ("326.ply", "327.ply", "326", 14, 3),
    "cloth-funnel": ("227.ply", "228.ply", "227", 18, 4),
    "n-body-simulation": ("balls16_18.ply", "balls16_19.ply", "18", 16, 2),
    "rod-twist": ("3036.ply", "3037.ply", "3036", 12, 5),
}

if __name__ == "__main__":
    out = sys.argv[1]
    for name, (t0, t1, gt, n, seed) in SCENES.items():
        d = os.path.join(out, name)
        os.makedirs(os.path.join(d, "frames"), exist_ok=True)
        os.makedirs(os.path.join(d, "boxes"), exist_ok=True)
        V0, V1, _, F = scenes.cloth_ball(n, 1, seed=seed)
        sio.write_ply(os.path.join(d, "frames", t0), V0, F)
        sio.write_ply(os.path.join(d, "frames", t1), V1, F)
        E = sio.igl_edges(F)
        vb, eb, fb = orc.build_boxes(V0, V1, E, F)
        vf, ee = orc.brute_force(vb, fb), orc.brute_force(eb)
        vfo, eeo = sio.offset_for_ground_truth(vf, ee, len(V0), len(E))
        json.dump(vfo[::2].tolist(), open(os.path.join(d, "boxes", gt + "vf.json"), "w"))
        json.dump(eeo[::2].tolist(), open(os.path.join(d, "boxes", gt + "ee.json"), "w"))
    print("wrote", out)
