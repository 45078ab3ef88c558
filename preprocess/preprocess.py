#!/usr/bin/env python
"""preprocess component — drop-in for preprocess/preprocess.py of the reference: clean the raw
rating table (duplicates, missing values, optional watched / plan-to-watch / half-watched filters,
users with too few ratings), scale the ratings to [0, 1] and log ``preprocessed_stats.parquet``.
The row passes run on the GPU (anime_recommendations_amd.ingest -> anirec_ingest_preprocess)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from anime_recommendations_amd import artifacts, components as C  # noqa: E402

STR_FLAGS = ["raw_stats", "project_name", "preprocessed_stats", "preprocessed_artifact_type",
             "preprocessed_artifact_description", "num_reviews"]
BOOL_FLAGS = ["drop_half_watched", "save_clean_locally", "drop_unwatched", "drop_plan"]

logger = C.setup_logging("preprocess")


def go(args):
    import pandas as pd
    from anime_recommendations_amd import ingest
    df = pd.read_parquet(artifacts.use_artifact(args.raw_stats, "Raw data"))
    logger.info("Artifact downloaded!")
    cols = ingest.preprocess_columns(ingest.frame_to_columns(df), num_reviews=int(args.num_reviews),
                                     drop_unwatched=args.drop_unwatched, drop_plan=args.drop_plan,
                                     drop_half_watched=args.drop_half_watched)
    # same column order and dtypes as the reference's frame (ids / status / episodes int64, rating float64)
    out = pd.DataFrame({k: (cols[k].cpu().numpy() if k == "rating" else cols[k].cpu().numpy().astype("int64"))
                        for k in ingest.COLUMNS})
    if args.drop_half_watched:       # the reference's frame keeps these two columns (preprocess.py:99-100,104)
        out["max_eps"] = cols["max_eps"].cpu().numpy().astype("int64")
        out["half_eps"] = cols["half_eps"].cpu().numpy()
    logger.info("Final df shape is %s", out.shape)
    logger.info("Final df columns are %s", out.columns)
    filename = args.preprocessed_stats
    if args.save_clean_locally:
        logger.info("Saving processed df to local machine")
    out.to_parquet(filename, index=False)
    artifacts.log_artifact(args.preprocessed_stats, filename, args.preprocessed_artifact_type,
                           args.preprocessed_artifact_description,
                           metadata={"Was data saved locally?": args.save_clean_locally})
    if not args.save_clean_locally:
        os.remove(filename)
    return out


if __name__ == "__main__":
    _args = C.make_parser("Preprocess a dataset", STR_FLAGS, BOOL_FLAGS).parse_args()
    try:
        go(_args)
    except Exception:                      # non-zero exit + the reason in ./preprocess.log
        logger.exception("preprocess failed")
        raise
