"""Regenerates tests/golden/oracle_hashes.json from the CPU oracle.

The reference ships no fixtures (no tests, no recorded outputs), so the golden
vectors are hashes of this repository's own oracle on a seeded synthetic pair;
they pin the oracle against accidental change, not against OpenCV."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

from oracle import oracle as O  # noqa: E402
from test_cpu_oracle import _golden_cases  # noqa: E402

if __name__ == "__main__":
    O.build()
    out = _golden_cases(O)
    json.dump(out, open(os.path.join(HERE, "oracle_hashes.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))
