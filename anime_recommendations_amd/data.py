"""Rating-table plumbing either side of the hot path (host, pandas/numpy).

Index contract of the reference (neural_network/neural_network.py:41-60, repeated in
similar_anime.py:43-56, similar_users.py:42-54, model_recs.py:76-85):
  * ``user_id`` / ``anime_id`` -> dense index = position in ``Series.unique()`` (order of first
    appearance in the parquet, BEFORE shuffling);
  * rows shuffled with ``df.sample(frac=1, random_state=42)``;
  * hold-out = the LAST ``test_size`` rows of the shuffled frame (neural_network.py:161-169).

Also the synthetic generators of SURVEY.md §8(d) (the real data files are not distributable).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np
import pandas as pd

RATING_PMF = np.array([0.18, 0.005, 0.005, 0.01, 0.02, 0.05, 0.10, 0.20, 0.22, 0.13, 0.08])
ANIME_TYPES = ["TV", "OVA", "Movie", "Special", "ONA", "Music"]
GENRES = ["Action", "Adventure", "Comedy", "Drama", "Fantasy", "Horror", "Mystery", "Romance",
          "Sci-Fi", "Slice of Life", "Sports", "Supernatural", "Vampire", "Music", "School"]


def encode_ids(ids):
    """Dense index of each id in order of first appearance; returns (index int64, uniques)."""
    codes, uniques = pd.factorize(np.asarray(ids), sort=False)
    return codes.astype(np.int64), np.asarray(uniques)


def shuffle_order(n, random_state=42):
    """Row order produced by ``df.sample(frac=1, random_state=random_state)``."""
    return np.random.RandomState(random_state).permutation(n)


@dataclass
class RatingTable:
    """Encoded, shuffled ratings in the reference's layout."""
    user: np.ndarray       # int64 [N] dense user index
    anime: np.ndarray      # int64 [N] dense anime index
    rating: np.ndarray     # float64/32 [N] in [0, 1]
    user_ids: np.ndarray   # index -> original user_id
    anime_ids: np.ndarray  # index -> original anime_id

    @property
    def n_users(self):
        return len(self.user_ids)

    @property
    def n_anime(self):
        return len(self.anime_ids)

    def __len__(self):
        return len(self.user)

    def split(self, test_size):
        """(train, test) index ranges: the last ``test_size`` shuffled rows are held out."""
        n_train = len(self) - int(test_size)
        if n_train <= 0:
            raise ValueError("test_size %d leaves no training rows out of %d" % (test_size, len(self)))
        return slice(0, n_train), slice(n_train, len(self))


def encode_frame(df: pd.DataFrame, shuffle=True, random_state=42, min_ratings=None) -> RatingTable:
    """get_df(): encode ids, shuffle, keep [user, anime, rating].

    ``min_ratings``: similar_anime.py:40-41 additionally keeps users with >= 400 ratings."""
    if min_ratings:
        n_ratings = df["user_id"].value_counts(dropna=True)
        df = df[df["user_id"].isin(n_ratings[n_ratings >= int(min_ratings)].index)]
    u, user_ids = encode_ids(df["user_id"].to_numpy())
    a, anime_ids = encode_ids(df["anime_id"].to_numpy())
    r = df["rating"].to_numpy()
    if shuffle:
        order = shuffle_order(len(df), random_state)
        u, a, r = u[order], a[order], r[order]
    return RatingTable(u, a, r, user_ids, anime_ids)


def load_user_stats(path, **kw) -> RatingTable:
    return encode_frame(pd.read_parquet(path), **kw)


# --------------------------------------------------------------------------------------
# synthetic data: same schema and marginal laws as user_stats.parquet
# --------------------------------------------------------------------------------------
def synth_user_stats(n_users=15_000, n_anime=17_560, n_ratings=7_000_000, seed=20260101,
                     min_per_user=None, max_per_user=None, zipf_s=1.0) -> pd.DataFrame:
    """Synthetic ``user_stats`` frame: per-user counts ~ lognormal, anime drawn WITHOUT
    replacement per user from a Zipf(s) popularity over a fixed permutation, MAL-like ratings,
    sparse ascending user ids, columns as in the reference's preprocess output."""
    rng = np.random.Generator(np.random.PCG64(seed))
    mean = n_ratings / n_users
    cnt = rng.lognormal(np.log(mean), 0.35, n_users)
    lo = min_per_user if min_per_user is not None else max(1, int(0.4 * mean))
    hi = max_per_user if max_per_user is not None else min(n_anime, int(7 * mean))
    cnt = np.clip(cnt, lo, hi)
    cnt = np.maximum(1, np.round(cnt * (n_ratings / cnt.sum()))).astype(np.int64)
    cnt = np.minimum(cnt, n_anime)
    pop = 1.0 / np.arange(1, n_anime + 1) ** zipf_s
    pop /= pop.sum()
    anime_perm = rng.permutation(n_anime)
    anime_id_of = np.sort(rng.choice(np.arange(1, 3 * n_anime), n_anime, replace=False))
    user_id_of = np.sort(rng.choice(np.arange(1, 4 * n_users), n_users, replace=False))
    # without replacement per user via Gumbel top-k on log-popularity
    logp = np.log(pop)
    users, animes = [], []
    for u in range(n_users):
        k = cnt[u]
        if k * 8 < n_anime:
            # rejection-free for small k: draw extra with replacement, dedupe, top up
            draw = np.unique(rng.choice(n_anime, size=int(k * 1.5) + 8, p=pop))
            rng.shuffle(draw)
            sel = draw[:k]
            if len(sel) < k:
                g = logp + rng.gumbel(size=n_anime)
                sel = np.argpartition(-g, k)[:k]
        else:
            g = logp + rng.gumbel(size=n_anime)
            sel = np.argpartition(-g, k - 1)[:k]
        users.append(np.full(len(sel), u, np.int64))
        animes.append(anime_perm[sel])
    u = np.concatenate(users)
    a = np.concatenate(animes)
    n = len(u)
    rating = rng.choice(11, size=n, p=RATING_PMF / RATING_PMF.sum()) / 10.0
    status = rng.choice([1, 2, 3, 4, 6], size=n, p=[0.08, 0.7, 0.06, 0.06, 0.1])
    episodes = rng.integers(0, 26, n)
    return pd.DataFrame({"user_id": user_id_of[u], "anime_id": anime_id_of[a], "rating": rating,
                         "watching_status": status, "watched_episodes": episodes})


def synth_anime_tables(anime_ids, seed=11):
    """Synthetic ``all_anime.csv`` / ``synopses.csv`` frames for the given anime ids
    (columns the reference's components read: similar_anime.py:70-93, model_recs.py:100-125)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = len(anime_ids)
    names = ["Anime %05d" % i for i in anime_ids]
    genres = [", ".join(sorted(rng.choice(GENRES, size=rng.integers(1, 4), replace=False))) for _ in range(n)]
    types = rng.choice(ANIME_TYPES, size=n, p=[0.45, 0.15, 0.15, 0.1, 0.1, 0.05])
    anime = pd.DataFrame({
        "MAL_ID": anime_ids, "Name": names, "Score": np.round(rng.uniform(4, 9.5, n), 2),
        "Genres": genres, "English name": names, "Japanese name": ["アニメ%05d" % i for i in anime_ids],
        "Type": types, "Episodes": rng.integers(1, 100, n).astype(str),
        "Premiered": rng.choice(["Spring 2010", "Fall 2015", "Winter 2019", "Unknown"], n),
        "Studios": rng.choice(["Studio A", "Studio B", "Studio C"], n),
        "Source": rng.choice(["Manga", "Original", "Light novel", "Game"], n),
        "Rating": rng.choice(["PG-13 - Teens 13 or older", "R - 17+ (violence & profanity)", "G - All Ages"], n),
        "Members": rng.integers(100, 2_000_000, n)})
    syn = pd.DataFrame({"MAL_ID": anime_ids, "Name": names, "Score": anime["Score"], "Genres": genres,
                        "sypnopsis": ["Synopsis of anime %d." % i for i in anime_ids]})
    return anime, syn


def write_synthetic_dataset(out_dir, **kw):
    """Writes user_stats.parquet, all_anime.csv, synopses.csv into ``out_dir``; returns paths."""
    os.makedirs(out_dir, exist_ok=True)
    df = synth_user_stats(**kw)
    anime, syn = synth_anime_tables(np.sort(df["anime_id"].unique()))
    paths = {"user_stats": os.path.join(out_dir, "user_stats.parquet"),
             "all_anime": os.path.join(out_dir, "all_anime.csv"),
             "synopses": os.path.join(out_dir, "synopses.csv")}
    df.to_parquet(paths["user_stats"], index=False)
    anime.to_csv(paths["all_anime"], index=False)
    syn.to_csv(paths["synopses"], index=False)
    return paths
