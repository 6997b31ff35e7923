"""The long join's alternative routes give the same chains: the A/B switches of DESIGN.md 3.2 ("Stretches") - the join shared among waves or
not and in how small runs, the giants on the exact instance at once or through the exact passes, large joins on the 4096-anchor ring's
pass, no follower launch - classify the same 20 000 reads (satellite reads with tied priorities among them) to the same flags AND traces.
One process per route: the library reads these switches once."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

ROUTES = (
    ("default", {}),
    ("join not shared", {"SCRUBBY_HIP_NO_COOP": "1"}),
    ("join shared in small runs", {"SCRUBBY_HIP_COOP_MIN": "512", "SCRUBBY_HIP_COOP_RUN": "128"}),
    ("giants on the plain instance first", {"SCRUBBY_HIP_GIANTS_PLAIN": "1"}),
    ("large joins to the 4096-anchor ring's pass, no follower", {"SCRUBBY_HIP_E2_JOIN_MIN": "20000", "SCRUBBY_HIP_NO_FOLLOW": "1"}),
)
SWITCHES = ("SCRUBBY_HIP_NO_COOP", "SCRUBBY_HIP_COOP_MIN", "SCRUBBY_HIP_COOP_RUN", "SCRUBBY_HIP_GIANTS_PLAIN", "SCRUBBY_HIP_E2_JOIN_MIN", "SCRUBBY_HIP_NO_FOLLOW")


def run_route(env_extra):
    env = {k: v for k, v in os.environ.items() if k not in SWITCHES}
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(HERE, "long_routes_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("ROUTE ")]
    assert line, p.stdout[-2000:]
    return json.loads(line[-1][6:])


def test_every_route_of_the_exact_long_join_gives_the_same_flags_and_traces():
    res = [(name, run_route(env)) for name, env in ROUTES]
    for name, r in res:
        print(name, r)
    base = res[0][1]
    assert base["rc"] == 0 and base["tied"] > 0 and base["open"] == 0 and base["unresolved"] == 0
    for name, r in res[1:]:
        assert r["rc"] == 0 and r["open"] == 0 and r["unresolved"] == 0, (name, r)
        assert r["sha1"] == base["sha1"], f"{name}: flags or traces differ from the default route ({r} vs {base})"
