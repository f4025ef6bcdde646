"""tools/soak_random.py <first> <last>: the fixed-seed random configuration sweep of tests/parity_cases.py::case_random_configs over more seeds than the test suite runs (14 per run),
on the GPU against the oracle: grids, tile batches with stats / normals / AO / weights / shadows and a whole-map erosion per seed, bit for bit."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orclib, parity_cases as pc
pkg = importlib.import_module("3dworld_amd")
a, b = int(sys.argv[1]), int(sys.argv[2])
orclib.build_oracle(); orc = orclib.Checker("orc")
t = pkg.Terra(0)
bad = 0
for s in range(a, b):
    try:
        pc.case_random_configs(pkg, t, orc, [s], big=True)
    except AssertionError as e:
        bad += 1; print("seed", s, "FAILED:", str(e)[:300], flush=True)
print(f"seeds {a}..{b-1}: {b-a-bad} ok, {bad} failed")
