"""MI355X-native embedding recommender: the hot path of Dyrutter/anime_recommendations
(neural_network / similar_anime / similar_users / model_recs) on hand-written gfx950 kernels.

Importing the package does not touch the GPU; the kernels live in libanirec.so
(built by ``anime_recommendations_amd.build``) and every op raises if it is missing.
"""
__version__ = "0.1.0"
