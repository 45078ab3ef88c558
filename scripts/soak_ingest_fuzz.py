"""One-off soak of the ingest fuzz tests with more seeds than the suite runs: python scripts/soak_ingest_fuzz.py [n_seeds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_ingest_gpu as t
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
t0 = time.time()
for seed in range(10, 10 + n):
    t.test_preprocess_random_layouts_against_pandas(seed)
    if seed % 3 == 0:
        t.test_encode_random_columns_against_series_unique(seed)
    if seed % 25 == 0:
        print("seed", seed, "ok  %.0f s" % (time.time() - t0), flush=True)
print("soak ok: %d layouts" % n)
