#!/usr/bin/env python
"""model_recs component — drop-in for model_recs/model_recs.py of the reference: rank every
anime a user has not watched by the model's predicted rating (``model.predict``), optional
Type / Genre filters, top ``model_num_recs`` to ``User_ID_<id>_<model_recs_fn>``."""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from anime_recommendations_amd import artifacts, components as C  # noqa: E402

STR_FLAGS = ["main_df", "main_df_type", "project_name", "anime_df", "anime_df_type", "sypnopsis_df",
             "sypnopsis_df_type", "model", "model_type", "model_user_query", "model_recs_fn", "model_num_recs",
             "anime_types", "model_genres", "model_recs_type", "flow_ID", "flow_ID_type"]
BOOL_FLAGS = ["random_user", "save_model_recs", "specify_types", "specify_genres", "model_ID_flow", "model_ID_conf"]

logger = C.setup_logging("model_recs")


def select_user(args, df):
    """select_user (model_recs.py:334-370): the user of the MLflow run (flow_ID artifact), the
    configured one, or a random one."""
    import pandas as pd
    if args.model_ID_flow:
        flow = pd.read_csv(artifacts.use_artifact(args.flow_ID, args.flow_ID_type))
        return int(flow["User_ID"].values[0])
    if args.model_ID_conf:
        return int(args.model_user_query)
    return int(random.choice(df["user_id"].unique().tolist()))


def go(args):
    import pandas as pd
    from anime_recommendations_amd import weights_io
    df = pd.read_parquet(artifacts.use_artifact(args.main_df, args.main_df_type))
    anime_df = C.load_anime_df(artifacts.use_artifact(args.anime_df, args.anime_df_type))
    syn_df = C.load_synopses(artifacts.use_artifact(args.sypnopsis_df, args.sypnopsis_df_type))
    model = weights_io.load_model(artifacts.use_artifact(args.model, args.model_type))
    user_ids, anime_ids = C.index_tables(model, df)
    user = select_user(args, df)
    logger.info("Using %s as input user", user)
    frame = C.model_recs_frame(model["U"], model["A"], model["head"], user_ids, anime_ids, df, anime_df, syn_df,
                               user, int(args.model_num_recs),
                               types=C.literal(args.anime_types) if args.specify_types else None,
                               genres=C.literal(args.model_genres) if args.specify_genres else None)
    fn = "User_ID_" + str(user) + "_" + args.model_recs_fn
    frame.to_csv(fn, index=False)
    artifacts.log_artifact(args.model_recs_fn, fn, args.model_recs_type,
                           "Anime recs based on model rankings for user : " + str(user),
                           metadata={"Queried user: ": user, "Filename": fn})
    if not args.save_model_recs:
        os.remove(fn)
    return frame


if __name__ == "__main__":
    _args = C.make_parser("Get anime recommendations from the ranking model", STR_FLAGS, BOOL_FLAGS).parse_args()
    try:
        go(_args)
    except Exception:                      # non-zero exit + the reason in ./model_recs.log
        logger.exception("model_recs failed")
        raise
