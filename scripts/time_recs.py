"""Favourites (109 M ratings, 350 k users) + user-based top-10 for 65 536 queries — for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
r = bench.run_user_recs(cpu_baseline=False)
print("favourites %.2f ms; user_recs %.2f ms (%.2f M queries/s)" % (r["favourites"]["ms"], r["ms"], r["value"] / 1e6))
