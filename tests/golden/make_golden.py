"""Regenerates tests/golden/oracle_renders.json: SHA-256 of small oracle renders of the bundled scenes.
These freeze the oracle's own behaviour (a regression net); the vectors that tie the oracle to the REFERENCE are the
SURVEY.md 8c values asserted in tests/test_oracle_golden.py."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol   # noqa: E402
import pyscene            # noqa: E402

ROOT = os.path.dirname(os.path.dirname(HERE))
CASES = [("tri", 1, 256, 256, 0), ("tri", 0, 256, 256, 0), ("tri", 0, 64, 64, 4), ("redchair", 0, 96, 54, 0),
         ("redchair", 0, 48, 27, 32), ("spiral", 0, 96, 54, 1), ("spiral", 0, 48, 27, 16), ("tenthousand", 0, 96, 54, 1),
         ("tenthousand", 0, 48, 27, 16), ("tenthousand", 0, 24, 16, 40)]
out = {"renders": []}
for scene, mode, w, h, spp in CASES:
    o = ol.OracleScene(pyscene.parse_file(os.path.join(ROOT, "scenes", scene + ".txt")), bounds_mode=mode)
    r = o.render(w, h, spp, nthreads=8)
    out["renders"].append(dict(scene=scene, bounds_mode=mode, width=w, height=h, spp=spp,
                               sha256_u8=hashlib.sha256(r["u8"].tobytes()).hexdigest(),
                               sha256_f32=hashlib.sha256(r["f32"].tobytes()).hexdigest(), stats=r["stats"]))
with open(os.path.join(HERE, "oracle_renders.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote", len(out["renders"]), "fixtures")
