"""TEST INFRASTRUCTURE — NumPy / pandas restatement of the reference's favourites and user-based
recommendation counting.  Only tests/ and bench.py's cpu_baseline leg may import this module.

Follows user_recs/user_recs.py:377-404 (fave_genres: ``np.percentile(watched.rating, 80)`` then
``watched.rating >= percentile``) and :708-760 (similar_user_recs: the favourites of every similar user
minus the query user's own, ``pd.Series(frame.values.ravel()).value_counts()``).  NumPy and pandas are the
libraries the reference calls.  The ORDER among equal counts is left by pandas to an unstable sort; the
build defines it (count desc, best similar-user rank asc, anime index asc) and ``user_recs`` below states
that definition; the counts themselves are checked against ``value_counts``.
"""
import numpy as np
import pandas as pd


def favourites(user_idx, anime_idx, rating, n_users, pct=80):
    """Per user: (threshold, set of favourite anime indices)."""
    df = pd.DataFrame({"user": user_idx, "anime": anime_idx, "rating": rating})
    thr = np.full(n_users, np.nan)
    fav = [set() for _ in range(n_users)]
    for u, watched in df.groupby("user"):
        p = np.percentile(watched.rating, pct)                 # user_recs.py:392
        thr[u] = p
        fav[u] = set(watched[watched.rating >= p].anime.tolist())   # :393-394
    return thr, fav


def value_counts_of_similar_favourites(fav, query, sims):
    """similar_user_recs (:732-745): ragged frame of the similar users' favourites minus the query's own,
    raveled, value_counts -> {anime: count}."""
    own = fav[query]
    rows = [np.array(sorted(fav[s] - own)) for s in sims if s >= 0]
    frame = pd.DataFrame(rows)
    vc = pd.Series(frame.values.ravel()).value_counts()
    return {int(a): int(c) for a, c in vc.items()}


def user_recs(fav, query, sims, n):
    own = fav[query]
    counts, best = {}, {}
    for j, s in enumerate(sims):
        if s < 0:
            continue
        for a in fav[s]:
            if a in own:
                continue
            counts[a] = counts.get(a, 0) + 1
            best.setdefault(a, j)
    order = sorted(counts, key=lambda a: (-counts[a], best[a], a))[:n]
    return order, [counts[a] for a in order]
