#!/usr/bin/env python
"""similar_anime component — drop-in for similar_anime/similar_anime.py of the reference:
anime most similar to a query anime by cosine of L2-normalised embedding rows, optional Type /
Genre filters, CSV named after the cleaned title.  Row-normalise + cosine + top-k run in
libanirec (HIP)."""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from anime_recommendations_amd import artifacts, components as C  # noqa: E402

STR_FLAGS = ["main_df_type", "anime_df_type", "sypnopsis_df_type", "model_type", "model", "project_name",
             "main_df", "sypnopses_df", "anime_df", "anime_query", "a_query_number", "anime_rec_genres",
             "types", "a_rec_type", "ID_emb_name", "anime_emb_name"]
BOOL_FLAGS = ["random_anime", "an_spec_genres", "spec_types", "save_sim_anime"]

logger = C.setup_logging("similar_anime")


def go(args):
    import pandas as pd
    from anime_recommendations_amd import weights_io
    anime_df = C.load_anime_df(artifacts.use_artifact(args.anime_df, args.anime_df_type))
    syn_df = C.load_synopses(artifacts.use_artifact(args.sypnopses_df, args.sypnopsis_df_type))
    model = weights_io.load_model(artifacts.use_artifact(args.model, args.model_type),
                                  args.ID_emb_name, args.anime_emb_name)
    main_df = None
    if model["anime_ids"] is None:
        main_df = pd.read_parquet(artifacts.use_artifact(args.main_df, args.main_df_type))
    _, anime_ids = C.index_tables(model, main_df, min_ratings=400)      # similar_anime.py:40-41
    if args.random_anime:
        name = random.choice(anime_df["Name"].dropna().unique().tolist())
        logger.info("Using %s as random input anime", name)
    else:
        name = args.anime_query
    frame, fn = C.similar_anime_frame(
        model["A"], anime_ids, anime_df, syn_df, name, int(args.a_query_number),
        types=C.literal(args.types) if args.spec_types else None,
        genres=C.literal(args.anime_rec_genres) if args.an_spec_genres else None)
    frame.to_csv(fn, index=False)
    artifacts.log_artifact(fn, fn, args.a_rec_type, "Anime most similar to: " + str(name),
                           metadata={"Queried anime": name, "Model used": args.model,
                                     "Main data frame used": args.main_df, "Filename": fn})
    if not args.save_sim_anime:
        os.remove(fn)
    return frame


if __name__ == "__main__":
    _args = C.make_parser("Get recommendations based on similar anime", STR_FLAGS, BOOL_FLAGS).parse_args()
    try:
        go(_args)
    except Exception:                      # non-zero exit + the reason in ./similar_anime.log
        logger.exception("similar_anime failed")
        raise
