#!/usr/bin/env python
"""similar_users component — drop-in for similar_users/similar_users.py of the reference:
users most similar to a query user (cosine of L2-normalised user embedding rows), with each
neighbour's favourite anime; writes ``User_<id>.csv`` and ``<id>.csv`` artefacts."""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from anime_recommendations_amd import artifacts, components as C  # noqa: E402

STR_FLAGS = ["anime_df", "anime_df_type", "model", "model_type", "project_name", "main_df", "main_df_type",
             "sim_user_query", "id_query_number", "max_ratings", "num_faves", "sim_users_fn", "sim_users_type",
             "ID_fn", "ID_type", "ID_emb_name", "anime_emb_name"]
BOOL_FLAGS = ["sim_random_user", "TV_only", "save_sim_locally"]

logger = C.setup_logging("similar_users")


def go(args):
    import pandas as pd
    from anime_recommendations_amd import weights_io
    df = pd.read_parquet(artifacts.use_artifact(args.main_df, args.main_df_type))
    anime_df = C.load_anime_df(artifacts.use_artifact(args.anime_df, args.anime_df_type))
    model = weights_io.load_model(artifacts.use_artifact(args.model, args.model_type),
                                  args.ID_emb_name, args.anime_emb_name)
    user_ids, _ = C.index_tables(model, df)
    if args.sim_random_user:
        # get_random_user (similar_users.py:104-125): a user with fewer than max_ratings ratings
        counts = df["user_id"].value_counts()
        pool = counts[counts < int(args.max_ratings)].index.tolist() or counts.index.tolist()
        user_id = int(random.choice(pool))
        logger.info("Using random user ID %s", user_id)
    else:
        user_id = int(args.sim_user_query)
    frame, fn = C.similar_users_frame(model["U"], user_ids, df, anime_df, user_id, int(args.id_query_number),
                                      int(args.num_faves), args.TV_only)
    frame.to_csv(fn, index=False)
    artifacts.log_artifact(args.sim_users_fn, fn, args.sim_users_type, "Users most similar to: " + str(user_id),
                           metadata={"Queried user": user_id, "Filename": fn, "num_sim_users": args.id_query_number})
    id_fn = str(user_id) + ".csv"
    pd.DataFrame([user_id], columns=["User_ID"]).to_csv(id_fn, index=False)
    artifacts.log_artifact(args.ID_fn, id_fn, args.ID_type, "User ID queried, will be re-used in further steps",
                           metadata={"Queried user": user_id, "Filename": id_fn})
    if not args.save_sim_locally:
        os.remove(fn)
        os.remove(id_fn)
    return frame


if __name__ == "__main__":
    _args = C.make_parser("Find the users most similar to a user", STR_FLAGS, BOOL_FLAGS).parse_args()
    try:
        go(_args)
    except Exception:                      # non-zero exit + the reason in ./similar_users.log
        logger.exception("similar_users failed")
        raise
