"""Time the GPU ingest path (preprocess + id encoding) on N synthetic raw rows resident in HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anime_recommendations_amd import ingest

n = int(sys.argv[1]) if len(sys.argv) > 1 else 109_000_000
n_users, n_anime = 350_000, 18_000
g = torch.Generator(device="cuda"); g.manual_seed(17)
cols = {
    # the raw animelist is grouped by user (ascending blocks of user_id); "random" = worst case for the counts
    "user_id": (torch.randint(0, n_users, (n,), generator=g, device="cuda", dtype=torch.int32) if "random" in sys.argv
                else torch.sort(torch.randint(0, n_users, (n,), generator=g, device="cuda", dtype=torch.int32))[0]),
    "anime_id": torch.randint(0, n_anime, (n,), generator=g, device="cuda", dtype=torch.int32),
    "rating": torch.randint(0, 11, (n,), generator=g, device="cuda", dtype=torch.int32).double(),
    "watching_status": torch.randint(1, 7, (n,), generator=g, device="cuda", dtype=torch.int32),
    "watched_episodes": torch.randint(0, 26, (n,), generator=g, device="cuda", dtype=torch.int32),
}
if "nodups" not in sys.argv:   # as bench.py run_ingest: ~0.5 % duplicate rows near their originals (same user)
    dup_dst = torch.unique(torch.randint(0, n, (n // 200,), generator=g, device="cuda"))
    dup_src = torch.clamp(dup_dst - torch.randint(1, 50, (dup_dst.numel(),), generator=g, device="cuda"), min=0)
    for k in cols:
        cols[k][dup_dst] = cols[k].clone()[dup_src]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = ingest.preprocess_columns(cols, num_reviews=250, drop_plan=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ui, uu = ingest.encode_ids(out["user_id"], out.bounds["user_id"])
    ai, au = ingest.encode_ids(out["anime_id"], out.bounds["anime_id"])
    torch.cuda.synchronize(); t2 = time.perf_counter()
m = out["user_id"].numel()
print("ingest n=%d kept=%d: preprocess %.2f ms (%.2f G rows/s), encode x2 %.2f ms; users %d anime %d" %
      (n, m, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9, (t2 - t1) * 1e3, uu.numel(), au.numel()))
