"""Golden vectors produced by the reference's OWN function bodies (build container only).

    python tests/golden/make_reference_function_fixtures.py

The reference's hot-path modules cannot be imported here (their first statements import wandb and
tensorflow, which are absent), but many of their functions touch neither: they are plain
pandas / NumPy / re / unicodedata code.  This script parses the reference source files with ``ast``,
compiles ONLY the listed ``FunctionDef`` nodes (no module-level statement of the reference is
executed), binds the names those bodies read (``pd``, ``np``, ``re``, ``string``, ``unicodedata``,
``ast``, a stdlib ``logger`` and an ``args`` namespace carrying the flags), runs them on seeded
inputs and writes INPUTS + OUTPUTS as data under ``tests/golden/ref_fn/``.  No reference source
text is written anywhere; nothing reference-derived travels as code.

Functions executed (reference file:line):
  preprocess/preprocess.py:13-40 drop_useless, :43-49 Convert, :52-105 drop_half_watched,
      :108-117 scale_ratings                        -> preprocess.npz / preprocess.json
  similar_anime/similar_anime.py:242-277 clean, :174-192 get_genres, :279-340 by_genre,
      :136-171 get_weights (model argument = a two-array holder)  -> clean.json, genres.json, get_weights.npz
  similar_users/similar_users.py:203-256 get_fave_anime, :262-314 find_similar_users
                                                      -> similar_users.npz / similar_users.json
  user_recs/user_recs.py:359-378 fave_genres            -> fave_genres.json
  model_recs/model_recs.py:132-156 get_unwatched, :159-192 get_user_anime_arr -> model_recs.json
  neural_network/neural_network.py:109-125 lrfn         -> lrfn.json

Not isolable (documented in DESIGN.md §2): anything that calls wandb (every ``go``, ``get_df``,
``main_df_by_*``, ``get_anime_df``, ``get_model``), Keras (``neural_network()``, ``model.fit``,
``model.predict`` inside ``recommendations``), and ``anime_recs`` / ``similar_user_recs`` whose
bodies call those loaders directly.
"""
import ast
import json
import logging
import os
import re
import string
import types
import unicodedata
import warnings

import numpy as np
import pandas as pd

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_fn")


def load_functions(relpath, names, args=None, extra=None):
    """Compile the named top-level functions of a reference file into a fresh namespace."""
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path, encoding="utf-8").read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    missing = set(names) - {n.name for n in picked}
    if missing:
        raise RuntimeError("%s lacks %s" % (relpath, sorted(missing)))
    mod = ast.Module(body=picked, type_ignores=[])
    ns = {"pd": pd, "np": np, "re": re, "string": string, "unicodedata": unicodedata, "ast": ast,
          "logger": logging.getLogger("reference"), "args": args or types.SimpleNamespace()}
    ns.update(extra or {})
    exec(compile(mod, path, "exec"), ns)
    return ns


def raw_frame(rng, n_rows, n_users, n_anime, nan_frac, dup_frac):
    """A raw rating frame with the reference's schema incl. duplicate rows and NaNs in every column."""
    df = pd.DataFrame({
        "user_id": np.sort(rng.integers(1, n_users + 1, n_rows)).astype(np.float64) * 7,
        "anime_id": rng.integers(1, n_anime + 1, n_rows).astype(np.float64) * 3,
        "rating": rng.integers(0, 11, n_rows).astype(np.float64),
        "watching_status": rng.choice([1, 2, 3, 4, 6], n_rows).astype(np.float64),
        "watched_episodes": rng.choice([0, 0, 1, 1, 2, 3, 6, 12, 13, 24, 25], n_rows).astype(np.float64)})
    nd = int(dup_frac * n_rows)
    if nd:
        dst = rng.integers(0, n_rows, nd)
        src = rng.integers(0, n_rows, nd)
        df.iloc[dst] = df.iloc[src].to_numpy()
    for c in df.columns:
        m = rng.random(n_rows) < nan_frac
        df.loc[m, c] = np.nan
    return df


def gen_preprocess():
    names = ["drop_useless", "Convert", "drop_half_watched", "scale_ratings"]
    rng = np.random.default_rng(20260102)
    arrays, cases = {}, []
    specs = []
    for du in (False, True):
        for dp in (False, True):
            for dh in (False, True):
                specs.append(dict(n_rows=600, n_users=12, n_anime=40, nan_frac=0.02, dup_frac=0.05,
                                  num_reviews=20, drop_unwatched=du, drop_plan=dp, drop_half_watched=dh))
    specs.append(dict(n_rows=2500, n_users=30, n_anime=90, nan_frac=0.01, dup_frac=0.10, num_reviews=60,
                      drop_unwatched=True, drop_plan=True, drop_half_watched=True))
    specs.append(dict(n_rows=300, n_users=5, n_anime=8, nan_frac=0.0, dup_frac=0.3, num_reviews=1,
                      drop_unwatched=False, drop_plan=False, drop_half_watched=True))
    specs.append(dict(n_rows=200, n_users=40, n_anime=30, nan_frac=0.05, dup_frac=0.0, num_reviews=4,
                      drop_unwatched=False, drop_plan=True, drop_half_watched=False))
    for ci, sp in enumerate(specs):
        df = raw_frame(rng, sp["n_rows"], sp["n_users"], sp["n_anime"], sp["nan_frac"], sp["dup_frac"])
        args = types.SimpleNamespace(drop_unwatched=sp["drop_unwatched"], drop_plan=sp["drop_plan"],
                                     num_reviews=str(sp["num_reviews"]))
        ns = load_functions("preprocess/preprocess.py", names, args)
        for c in df.columns:
            arrays["c%d_in_%s" % (ci, c)] = df[c].to_numpy(np.float64)
        # go(): preprocess.py:130-141 — drop_useless, optional drop_half_watched, scale_ratings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = ns["drop_useless"](df.copy())
            if sp["drop_half_watched"] and len(out):
                out = ns["drop_half_watched"](out)
            if len(out):
                out = ns["scale_ratings"](out)
        arrays["c%d_out_rows" % ci] = np.asarray(out.index, np.int64)      # surviving input row numbers
        for c in out.columns:
            arrays["c%d_out_%s" % (ci, c)] = out[c].to_numpy(np.float64)
        cases.append(dict(sp, out_columns=list(out.columns), n_out=int(len(out))))
    np.savez_compressed(os.path.join(OUT, "preprocess.npz"), **arrays)
    json.dump({"cases": cases}, open(os.path.join(OUT, "preprocess.json"), "w"), indent=1)


TITLES = ["Rental Magica", "Silent Möbius", "Fate/stay night: Unlimited Blade Works", "Gintama°", "K-On!!",
          "Lucky☆Star", "Mahou Shoujo Madoka★Magica", "Steins;Gate 0", "Kaguya-sama wa Kokurasetai: Tensai-tachi no Renai Zunousen",
          "Re:Zero kara Hajimeru Isekai Seikatsu 2nd Season", "Yuri!!! on Ice", "xxxHOLiC◆Kei", "Pokémon",
          "Tokyo Ghoul √A", "Nisekoi:", "Amagami SS+ Plus", "Love Live! School Idol Project", "Ichigo 100%",
          "Bishoujo Senshi Sailor Moon S", "Hunter x Hunter (2011)", "Kimi ni Todoke 2nd Season", "  leading\tand trailing\n",
          "½ Prince", "Straße", "E = mc²", "C³", "♥Heart♡", "ÀÉÎõü ñ ç", "日本語のタイトル", "under_score_name", ""]


def gen_clean_and_genres():
    ns = load_functions("similar_anime/similar_anime.py", ["clean", "get_genres", "by_genre"])
    clean = ns["clean"]
    rec = {"inputs": TITLES, "outputs": [clean(t) for t in TITLES], "list_output": clean(list(TITLES))}
    json.dump(rec, open(os.path.join(OUT, "clean.json"), "w"), indent=1, ensure_ascii=False)

    rng = np.random.default_rng(5)
    pool = ["Action", "Adventure", "Comedy", "Drama", "Sci-Fi", "Slice of Life", "Super Power", "Martial Arts",
            "Shounen Ai", "Mystery", "Romance", "School", "Music", "Mecha"]
    genres = []
    for _ in range(120):
        k = rng.integers(1, 5)
        genres.append(", ".join(rng.choice(pool, k, replace=False)))
    genres[7] = np.nan
    genres[19] = np.nan
    frame = pd.DataFrame({"Name": ["A%03d" % i for i in range(120)], "Genres": genres,
                          "Similarity": rng.random(120).round(6)})
    out = {"genres_column": [None if isinstance(g, float) else g for g in genres],
           "get_genres": ns["get_genres"](frame), "by_genre": []}
    for triple in (["Action", "Comedy", "Drama"], ["Slice of Life", "None", "None"], ["Sci-Fi", "Super Power", "None"],
                   ["Martial Arts", "Shounen Ai", "Music"], ["None", "Mecha", "None"], ["Sci-Fi", "None", "None"],
                   ["Martial Arts", "Music", "School"]):
        ns["args"].anime_rec_genres = str(triple)
        kept = ns["by_genre"](frame)
        # None: an invalid genre (the reference logs and returns None) or no matching row at all
        out["by_genre"].append({"genres": triple, "kept_rows_in_output_order":
                                None if kept is None else [int(i) for i in kept.index]})
    json.dump(out, open(os.path.join(OUT, "genres.json"), "w"), indent=1, ensure_ascii=False)


class _Layer:
    def __init__(self, w):
        self._w = w

    def get_weights(self):
        return [self._w]


class _TwoTables:
    """Holds two weight arrays under layer names: the ``model`` argument of get_weights()."""

    def __init__(self, layers):
        self._layers = layers

    def get_layer(self, name):
        return _Layer(self._layers[name])


def gen_get_weights():
    args = types.SimpleNamespace(anime_emb_name="anime_embedding", ID_emb_name="user_embedding")
    ns = load_functions("similar_anime/similar_anime.py", ["get_weights"], args)
    rng = np.random.default_rng(11)
    A = (rng.standard_normal((257, 128)) * 0.05).astype(np.float32)
    U = (rng.uniform(-0.05, 0.05, (300, 128))).astype(np.float32)
    A[5] *= 1e-20                     # tiny-norm row
    U[17] = 0.0                       # zero row -> NaN row (no epsilon in the reference)
    with np.errstate(all="ignore"):
        an, un = ns["get_weights"](_TwoTables({"anime_embedding": A, "user_embedding": U}))
    np.savez_compressed(os.path.join(OUT, "get_weights.npz"), A=A, U=U, A_norm=an, U_norm=un)


def gen_similar_users():
    ns = load_functions("similar_users/similar_users.py", ["get_fave_anime", "find_similar_users"])
    rng = np.random.default_rng(23)
    n_users, n_anime = 500, 60
    # clustered embeddings so that similarities are well separated from fp32 rounding
    centers = rng.standard_normal((12, 128))
    W = (centers[rng.integers(0, 12, n_users)] + 0.6 * rng.standard_normal((n_users, 128))).astype(np.float32)
    W = (W / np.linalg.norm(W, axis=1).reshape(-1, 1)).astype(np.float32)
    user_ids = (np.sort(rng.choice(np.arange(10, 5000), n_users, replace=False))).tolist()
    user_to_index = {u: i for i, u in enumerate(user_ids)}
    index_to_user = {i: u for i, u in enumerate(user_ids)}
    anime_ids = np.arange(1, n_anime + 1) * 11
    anime_df = pd.DataFrame({"anime_id": anime_ids, "Name": ["Title %d" % a for a in anime_ids],
                             "Episodes": rng.choice(["1", "12", "13", "24", "26", "50"], n_anime)})
    rows = []
    for u in user_ids:
        k = int(rng.integers(3, 9))
        for a in rng.choice(anime_ids, k, replace=False):
            rows.append((u, int(a), float(rng.integers(5, 11)) / 10.0, int(rng.integers(1, 27))))
    df = pd.DataFrame(rows, columns=["user_id", "anime_id", "rating", "watched_episodes"])
    cases = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for q, n, nf, tv in ((user_ids[0], 10, 3, True), (user_ids[123], 10, 1, False), (user_ids[499], 25, 2, True),
                             (user_ids[250], 1, 5, False), (user_ids[77], 100, 1, True)):
            frame, filename, uid = ns["find_similar_users"](q, n, nf, tv, df, anime_df, user_to_index, index_to_user, W)
            cases.append({"user_id": int(q), "n_users": n, "num_faves": nf, "TV_only": tv, "filename": filename,
                          "similar_users": [int(x) for x in frame["similar_users"]],
                          "similarity": [float(np.float32(x)) for x in frame["similarity"]],
                          "favorite_animes": list(frame["favorite_animes"])})
    np.savez_compressed(os.path.join(OUT, "similar_users.npz"), W=W, user_ids=np.asarray(user_ids, np.int64),
                        anime_id=anime_df["anime_id"].to_numpy(), episodes=anime_df["Episodes"].to_numpy().astype("U"),
                        df_user_id=df["user_id"].to_numpy(), df_anime_id=df["anime_id"].to_numpy(),
                        df_rating=df["rating"].to_numpy(), df_watched=df["watched_episodes"].to_numpy())
    json.dump({"cases": cases}, open(os.path.join(OUT, "similar_users.json"), "w"), indent=1)


def gen_fave_genres():
    ns = load_functions("user_recs/user_recs.py", ["fave_genres"])
    rng = np.random.default_rng(31)
    n_anime = 80
    anime_df = pd.DataFrame({"anime_id": np.arange(n_anime), "eng_version": [str(i) for i in range(n_anime)],
                             "Genres": ["g"] * n_anime})
    users = {}
    rows = []
    for u in range(40):
        k = int(rng.integers(1, 60))
        a = rng.choice(n_anime, k, replace=False)
        # ratings as produced by scale_ratings: multiples of 0.1 in float64 arithmetic ((x - 0) / (10 - 0))
        r = rng.integers(0, 11, k).astype(np.float64) / 10.0
        if u % 7 == 0:
            r[:] = r[0]              # constant ratings: every anime is a favourite
        for ai, ri in zip(a, r):
            rows.append((u, int(ai), float(ri)))
    df = pd.DataFrame(rows, columns=["user_id", "anime_id", "rating"])
    for u in range(40):
        fav = ns["fave_genres"](u, df, anime_df)
        users[str(u)] = sorted(int(x) for x in fav["eng_version"])
    json.dump({"n_anime": n_anime, "user_id": df["user_id"].tolist(), "anime_id": df["anime_id"].tolist(),
               "rating": df["rating"].tolist(), "favourites": users},
              open(os.path.join(OUT, "fave_genres.json"), "w"))


def gen_model_recs():
    ns = load_functions("model_recs/model_recs.py", ["get_unwatched", "get_user_anime_arr"])
    rng = np.random.default_rng(41)
    n_anime = 50
    all_ids = np.arange(1, n_anime + 1) * 5
    anime_df = pd.DataFrame({"anime_id": np.concatenate([all_ids, [9991, 9992]])})   # two anime nobody rated
    rows = []
    for u in (3, 8, 21, 34, 55):
        for a in rng.choice(all_ids[:45], int(rng.integers(5, 30)), replace=False):   # five ids never rated at all
            rows.append((u, int(a)))
    df = pd.DataFrame(rows, columns=["user_id", "anime_id"])
    out = {"df_user_id": df["user_id"].tolist(), "df_anime_id": df["anime_id"].tolist(),
           "anime_df_ids": anime_df["anime_id"].tolist(), "users": {}}
    for u in (3, 8, 21, 34, 55):
        unw = ns["get_unwatched"](df, anime_df, u)
        ua, aa = ns["get_user_anime_arr"](df, anime_df, u, unw)
        out["users"][str(u)] = {"unwatched_indices_sorted": sorted(int(x[0]) for x in unw),
                                "user_index": int(ua[0]), "n_pairs": int(len(aa))}
    json.dump(out, open(os.path.join(OUT, "model_recs.json"), "w"))


def gen_lrfn():
    out = []
    for cfg in (dict(start_lr="0.00001", max_lr="0.00005", min_lr="0.00001", rampup_epochs="5", sustain_epochs="0",
                     exp_decay=".8"),
                dict(start_lr="0.0001", max_lr="0.001", min_lr="0.00002", rampup_epochs="3", sustain_epochs="4",
                     exp_decay=".5"),
                dict(start_lr="0.01", max_lr="0.01", min_lr="0.001", rampup_epochs="0", sustain_epochs="0",
                     exp_decay=".9")):
        ns = load_functions("neural_network/neural_network.py", ["lrfn"], types.SimpleNamespace(**cfg))
        out.append({"flags": cfg, "lr": [float(ns["lrfn"](e)) for e in range(25)]})
    json.dump(out, open(os.path.join(OUT, "lrfn.json"), "w"), indent=1)


def main():
    os.makedirs(OUT, exist_ok=True)
    gen_preprocess()
    gen_clean_and_genres()
    gen_get_weights()
    gen_similar_users()
    gen_fave_genres()
    gen_model_recs()
    gen_lrfn()
    json.dump({"numpy": np.__version__, "pandas": pd.__version__,
               "note": "outputs of the reference's own function bodies (see the docstring of "
                       "make_reference_function_fixtures.py); the reference pins numpy 1.23.5 / pandas 1.5.3"},
              open(os.path.join(OUT, "VERSIONS.json"), "w"), indent=1)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
