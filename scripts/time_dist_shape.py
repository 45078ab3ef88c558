"""One-rank rehearsal of the user-sharded multi-GPU step at the PER-RANK shapes of G = 1, 2, 4, 8 GPUs (350 000 / G
local user rows, 18 000 replicated anime rows, 10 000 ratings per rank and step), RCCL world 1 (the collectives run,
over one rank), dense user rows against lazy user rows, Python loop against the C loop:

    python scripts/time_dist_shape.py [G ...]

What it cannot show: the wire time of the collectives and the other ranks' skew.  What it does show: the GPU time of
everything a rank does per step, and the host time to enqueue it."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                  HSA_ENABLE_IPC_MODE_LEGACY="0", ANIREC_DIST_LOOP="1")
import torch
import torch.distributed as dist

import bench
from anime_recommendations_amd import dist_bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
Gs = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]
B, K, W, inst = 10_000, 96, 16, 32
out = {}
for G in Gs:
    n_users, n_anime = 350_000 // G, 18_000
    ui, ai, t = bench.synth_ratings(n_users, n_anime, (W + K + inst) * B, dev)
    U, A = bench.init_tables(n_users, n_anime, dev)
    for lazy in ("0", "1"):
        for native in ("0", "1"):
            os.environ["ANIREC_LAZY_ADAM"] = lazy
            os.environ["ANIREC_DIST_NATIVE"] = native
            r = dist_bench._train_leg("sharded", (ui, ai, t), (U, A), n_users, n_anime, B, K, W, inst, 0, 1, dev)
            key = "G%d_lazy%s_%s" % (G, lazy, "c" if native == "1" else "py")
            out[key] = {k: r[k] for k in ("ms_per_step", "host_issue_ms_per_step", "stage_ms", "final_loss", "local_rows")}
            print(key, json.dumps(out[key]), flush=True)
    del ui, ai, t, U, A
    torch.cuda.empty_cache()
dist.barrier()
dist.destroy_process_group()
