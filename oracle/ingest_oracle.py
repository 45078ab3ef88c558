"""TEST INFRASTRUCTURE — pandas restatement of the reference's preprocess + id encoding.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (anime_recommendations_amd.ingest -> libanirec) never does.

Follows preprocess/preprocess.py:13-40 (drop_useless), :52-105 (drop_half_watched, restated with
vectorised pandas: the reference's per-row dict lookups compute the same per-anime maximum and the
same `max == 1 ? 1 : max * .5` bound), :108-117 (scale_ratings) and
neural_network/neural_network.py:41-60 (Series.unique() position encoding).  The reference module
itself cannot be imported (wandb / tensorflow are absent); pandas here IS the library the reference
calls, so the arithmetic and ordering semantics are the reference's own.
"""
import numpy as np
import pandas as pd


def drop_useless(df, num_reviews, drop_unwatched=False, drop_plan=False):
    df = df.drop_duplicates()                       # preprocess.py:25
    df = df.dropna()                                # :26
    if drop_unwatched:
        df = df[df["watched_episodes"] != 0]        # :29-30
    if drop_plan:
        df = df[df["watching_status"] != 6]         # :33-34
    n_ratings = df["user_id"].value_counts(dropna=True)   # :37
    return df[df["user_id"].isin(n_ratings[n_ratings >= int(num_reviews)].index)].copy()  # :38-39


def drop_half_watched(df):
    mx = df.groupby("anime_id")["watched_episodes"].max()          # :63-64
    half = pd.Series(np.where(mx == 1, mx, mx * .5), index=mx.index)  # :79-84
    df = df.copy()
    df["max_eps"] = df["anime_id"].map(mx).to_numpy()               # :99  (the two columns stay in the frame)
    df["half_eps"] = df["anime_id"].map(half).to_numpy()            # :100
    return df[df["watched_episodes"] >= df["half_eps"]]             # :104


def scale_ratings(df):
    mn, mx = min(df["rating"]), max(df["rating"])                   # :112-113
    df = df.copy()
    df["rating"] = df["rating"].apply(lambda x: (x - mn) / (mx - mn)).values.astype(np.float64)  # :115-116
    return df


def preprocess(df, num_reviews, drop_unwatched=False, drop_plan=False, drop_half=False):
    """go(): preprocess.py:130-141."""
    df = drop_useless(df, num_reviews, drop_unwatched, drop_plan)
    if drop_half:
        df = drop_half_watched(df)
    if len(df) == 0:
        return df
    return scale_ratings(df)


def encode(series):
    """neural_network.py:41-60: {id: position in unique()} applied to the column."""
    ids = series.unique().tolist()
    enc = {x: i for i, x in enumerate(ids)}
    return series.map(enc).to_numpy(), np.asarray(ids)
