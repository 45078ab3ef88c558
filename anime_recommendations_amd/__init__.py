"""MI355X-native embedding recommender: the hot path of Dyrutter/anime_recommendations
(neural_network / similar_anime / similar_users / model_recs) on hand-written gfx950 kernels.

Importing the package does not touch the GPU; the kernels live in libanirec.so
(built by ``anime_recommendations_amd.build``) and every op raises if it is missing.
"""
import os as _os

# HIP multiplexes a process's streams onto 4 hardware queues by default.  This package uses several (one per training
# engine, the side chains of the cosine top-k job, the ingest / recs helpers), and two chains of one job that land on
# the same hardware queue serialise (350 k x 350 k top-100: 28.0 ms instead of 24.9).  The runtime reads the variable
# when it initialises — import the package before the first torch.cuda call, or export it yourself.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

__version__ = "0.1.0"
