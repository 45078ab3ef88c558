"""Shared host logic of the four drop-in components (neural_network, similar_anime,
similar_users, model_recs): flag parsing in the reference's style, metadata tables, filters
and the CSV frames they write.  The math (row-normalise, cosine top-k, predict) is libanirec.

Reference behaviour mirrored here (file:line in each function).  The per-row pandas loops of
the reference (similar_anime.py:413-455, model_recs.py:403-445) are replaced by boolean masks
evaluated BEFORE the GPU top-k, which yields the same rows: filter -> sort desc -> head(k).
"""
from __future__ import annotations

import argparse
import ast
import logging
import re
import string
import unicodedata

import numpy as np
import pandas as pd

ANIME_TYPES = ['TV', 'OVA', 'Movie', 'Special', 'ONA', 'Music']
_IRREGULAR = "★♥☆♡½ß²"


def str2bool(v):
    """distutils.util.strtobool semantics used by every ``type=lambda x: bool(strtobool(x))`` flag."""
    s = str(v).strip().lower()
    if s in ("y", "yes", "t", "true", "on", "1"):
        return True
    if s in ("n", "no", "f", "false", "off", "0"):
        return False
    raise argparse.ArgumentTypeError("invalid truth value %r" % (v,))


def make_parser(description, str_flags, bool_flags):
    """All flags required, strings unless listed as bool — as in the reference's argparse blocks
    (e.g. neural_network.py:298-555)."""
    p = argparse.ArgumentParser(description=description, fromfile_prefix_chars="@")
    for f in str_flags:
        p.add_argument("--" + f, type=str, required=True)
    for f in bool_flags:
        p.add_argument("--" + f, type=str2bool, required=True)
    return p


def setup_logging(name):
    logging.basicConfig(filename="./%s.log" % name, level=logging.INFO, filemode="a",
                        format="%(asctime)s-%(name)s - %(levelname)s - %(message)s",
                        datefmt="%d %b %Y %H:%M:%S %Z", force=True)
    return logging.getLogger()


def clean(item):
    """Filename/lookup normalisation of titles and genres (similar_anime.py:242-277): special
    symbols and whitespace removed, non-word characters dropped, accents stripped, lower-cased."""
    if isinstance(item, (list, tuple)):
        return [clean(x) for x in item]
    s = str(item)
    for ch in _IRREGULAR:
        s = s.replace(ch, " ")
    s = s.translate({ord(c): None for c in string.whitespace})
    s = re.sub(r"\W+", "", s)
    s = "".join(c for c in unicodedata.normalize("NFKD", s) if not unicodedata.combining(c))
    return s.lower()


def load_anime_df(path):
    """get_anime_df (similar_anime.py:63-93): 'Unknown' -> NaN, id/name columns, cleaned english
    name for lookups, sorted by Score descending."""
    df = pd.read_csv(path)
    df = df.replace("Unknown", np.nan)
    df["anime_id"] = df["MAL_ID"]
    df["japanese_name"] = df["Japanese name"]
    df["eng_version"] = [clean(x) for x in df["Name"]]
    df = df.sort_values(by=["Score"], ascending=False, kind="quicksort", na_position="last")
    keep = ["anime_id", "eng_version", "Score", "Genres", "Episodes", "Premiered", "Studios",
            "japanese_name", "Name", "Type", "Source", "Rating", "Members"]
    return df[[c for c in keep if c in df.columns]]


def load_synopses(path):
    return pd.read_csv(path)


def all_genres(anime_df):
    """get_genres (similar_anime.py:174-192): the text of the list of distinct Genres cells is split on
    whitespace, every token stripped of non-alphanumerics; the fragments of the three multi-word genres
    (and 'nan') are dropped and 'Slice of Life', 'Super Power', 'Martial Arts', 'None' appended — so, as in
    the reference, any OTHER multi-word genre ("Shounen Ai") is only known by its fragments."""
    cells = anime_df["Genres"].unique().tolist()
    tokens = sorted({re.sub(r"[\W_]", "", t) for t in str(cells).split()})
    fragments = {"Slice", "of", "Life", "Martial", "Arts", "Super", "Power", "nan"}
    return sorted(t for t in tokens + ["Slice of Life", "Super Power", "Martial Arts", "None"] if t not in fragments)


def genre_mask(genres_col, wanted):
    """by_genre (similar_anime.py:279-340): keep rows whose Genres contain ANY of the (three)
    wanted genres; 'None' entries ignored.  The wanted genres are clean()ed, the column text is only
    lower-cased with spaces removed (:307-317) — so, as in the reference, a genre whose name holds
    punctuation ("Sci-Fi" -> "scifi" vs "sci-fi") matches nothing
    (tests/golden/ref_fn/genres.json holds the reference function's own outputs)."""
    wanted = [w for w in clean(list(wanted)) if w != "none"]
    col = [str(g).lower().replace(" ", "") for g in genres_col]
    m = np.zeros(len(col), bool)
    for w in wanted:
        m |= np.array([w in c for c in col], dtype=bool)
    return m


def _topk_count(asked, what):
    """The reference returns as many rows as asked for (Frame[:count]); the fused top-k kernels hold at
    most MAX_TOPK rows per query, so a larger request is an error here, never a silent truncation."""
    from ._lib import MAX_TOPK
    k = int(asked)
    if k < 1:
        raise ValueError("%s must be >= 1 (got %d)" % (what, k))
    if k > MAX_TOPK:
        raise ValueError("%s = %d exceeds the %d rows the GPU top-k kernels return per query "
                         "(ANIREC_MAX_TOPK)" % (what, k, MAX_TOPK))
    return k


def check_genres(wanted, anime_df):
    valid = set(clean(all_genres(anime_df)))
    for g in clean(list(wanted)):
        if g not in valid:
            raise ValueError("An invalid genre was input (%r). Select genres from %s" % (g, sorted(valid)))


def check_types(types):
    for t in types:
        if t not in ANIME_TYPES:
            raise ValueError("An invalid type was input (%r). Select from %s" % (t, ANIME_TYPES))
    return list(types)


def literal(s):
    return ast.literal_eval(s) if isinstance(s, str) else s


# ----------------------------------------------------------------------------------------
# index <-> id tables
# ----------------------------------------------------------------------------------------
def index_tables(model, main_df=None, min_ratings=None):
    """index->id arrays for users and anime.  Taken from the model file when present (written by
    the neural_network component); otherwise rebuilt from the rating frame exactly like the
    reference does (order of first appearance, similar_users.py:42-54)."""
    from .data import encode_ids
    if model.get("user_ids") is not None and model.get("anime_ids") is not None:
        return np.asarray(model["user_ids"]), np.asarray(model["anime_ids"])
    if main_df is None:
        raise ValueError("model file has no id tables and no main data frame was given")
    df = main_df
    if min_ratings:
        n = df["user_id"].value_counts(dropna=True)
        df = df[df["user_id"].isin(n[n >= int(min_ratings)].index)]
    _, uids = encode_ids(df["user_id"].to_numpy())
    _, aids = encode_ids(df["anime_id"].to_numpy())
    return uids, aids


def metadata_by_index(anime_ids, anime_df, syn_df=None):
    """Metadata rows aligned with the anime index (row i describes anime_ids[i]); `has_meta`
    False where the anime is missing from all_anime.csv (the reference's lookups would raise)."""
    meta = anime_df.drop_duplicates("anime_id").set_index("anime_id")
    out = meta.reindex(anime_ids)
    out["has_meta"] = out["Name"].notna().to_numpy()
    out["anime_id"] = np.asarray(anime_ids)
    if syn_df is not None:
        syn = syn_df.drop_duplicates("MAL_ID").set_index("MAL_ID")["sypnopsis"]
        out["Sypnopsis"] = syn.reindex(anime_ids).fillna("None").to_numpy()
    else:
        out["Sypnopsis"] = "None"
    return out.reset_index(drop=True)


# ----------------------------------------------------------------------------------------
# similar_anime
# ----------------------------------------------------------------------------------------
def find_anime_id(name, anime_df):
    """Resolve a query title like anime_recs (similar_anime.py:389-399): exact Name, else the
    cleaned name against the cleaned english names."""
    hit = anime_df[anime_df.Name == name]
    if len(hit) == 0:
        hit = anime_df[anime_df.eng_version == clean(name)]
    if len(hit) == 0:
        raise ValueError("anime %r not found in the anime data frame" % (name,))
    return int(hit.anime_id.values[0])


def similar_anime_frame(A, anime_ids, anime_df, syn_df, name, count, types=None, genres=None):
    """anime_recs (similar_anime.py:364-471): cosine of the query anime vs all, query excluded,
    optional Type / Genre filters, top ``count`` by similarity.  Returns (frame, filename)."""
    import torch
    from . import ops
    qid = find_anime_id(name, anime_df)
    pos = np.nonzero(np.asarray(anime_ids) == qid)[0]
    if len(pos) == 0:
        raise ValueError("anime %r (id %d) has no embedding row" % (name, qid))
    q = int(pos[0])
    meta = metadata_by_index(anime_ids, anime_df, syn_df)
    keep = meta["has_meta"].to_numpy().copy()
    if types is not None:
        keep &= meta["Type"].isin(check_types(types)).to_numpy()
    if genres is not None:
        check_genres(genres, anime_df)
        keep &= genre_mask(meta["Genres"], genres)
    Wh = ops.rownorm(torch.as_tensor(A))
    k = _topk_count(count, "a_query_number")
    idx, sim = ops.cosine_topk(Wh, [q], k, exclude_self=True, keep=keep.astype(np.uint8))
    idx, sim = idx.cpu().numpy()[0], sim.cpu().numpy()[0]
    ok = idx >= 0
    rows = meta.iloc[idx[ok]]
    frame = pd.DataFrame({
        "Name": rows["Name"].to_numpy(), "Similarity": sim[ok], "Genres": rows["Genres"].to_numpy(),
        "Sypnopsis": rows["Sypnopsis"].to_numpy(), "Episodes": rows["Episodes"].to_numpy(),
        "Japanese name": rows["japanese_name"].to_numpy(), "Studios": rows["Studios"].to_numpy(),
        "Premiered": rows["Premiered"].to_numpy(), "Score": rows["Score"].to_numpy(),
        "Type": rows["Type"].to_numpy(), "Source": rows["Source"].to_numpy(),
        "Rating": rows["Rating"].to_numpy()})
    return frame, clean(name) + ".csv"


# ----------------------------------------------------------------------------------------
# similar_users
# ----------------------------------------------------------------------------------------
def fave_anime(df, anime_df, user_id, num_faves, tv_only):
    """get_fave_anime (similar_users.py:203-256): top-rated anime of a user, narrowed to the
    highest watched fraction, optionally ordered by episode count; returned as the reference's
    ``str(list)[1:-1]`` text."""
    f = df[df.user_id == user_id]
    if len(f) == 0:
        return ""
    f = f[f.rating == f.rating.max()].copy()
    meta = anime_df.drop_duplicates("anime_id").set_index("anime_id")
    f["name"] = meta["Name"].reindex(f.anime_id).to_numpy()
    f["episodes"] = pd.to_numeric(meta["Episodes"].reindex(f.anime_id), errors="coerce").to_numpy(np.float32)
    if "watched_episodes" in f.columns:
        f["percent"] = f["watched_episodes"] / f["episodes"]
        f = f[f.percent == f.percent.max()] if f.percent.notna().any() else f
    if tv_only:
        f = f.sort_values(by="episodes", ascending=False)
    return str(f["name"].tolist()[: int(num_faves)])[1:-1]


def similar_users_frame(U, user_ids, df, anime_df, user_id, n_users, num_faves, tv_only):
    """find_similar_users (similar_users.py:262-314): the n most similar users (query dropped),
    descending similarity, with each neighbour's favourite anime."""
    import torch
    from . import ops
    pos = np.nonzero(np.asarray(user_ids) == int(user_id))[0]
    if len(pos) == 0:
        raise ValueError("user id %r has no embedding row" % (user_id,))
    Uh = ops.rownorm(torch.as_tensor(U))
    k = _topk_count(n_users, "id_query_number")
    idx, sim = ops.cosine_topk(Uh, [int(pos[0])], k, exclude_self=True)
    idx, sim = idx.cpu().numpy()[0], sim.cpu().numpy()[0]
    ok = idx >= 0
    ids = np.asarray(user_ids)[idx[ok]]
    frame = pd.DataFrame({"similar_users": ids, "similarity": sim[ok],
                          "favorite_animes": [fave_anime(df, anime_df, u, num_faves, tv_only) for u in ids]})
    fn = "User_" + str(user_id).translate({ord(c): None for c in string.whitespace}) + ".csv"
    return frame, fn


# ----------------------------------------------------------------------------------------
# model_recs
# ----------------------------------------------------------------------------------------
def model_recs_frame(U, A, head, user_ids, anime_ids, df, anime_df, syn_df, user_id, n_recs,
                     types=None, genres=None):
    """recommendations (model_recs.py:373-456): predicted rating of every unwatched, indexed
    anime for one user, Type / Genre filters, descending prediction, first ``n_recs``."""
    import torch
    from . import ops
    pos = np.nonzero(np.asarray(user_ids) == int(user_id))[0]
    if len(pos) == 0:
        raise ValueError("user id %r has no embedding row" % (user_id,))
    meta = metadata_by_index(anime_ids, anime_df, syn_df)
    watched_ids = set(df[df.user_id == int(user_id)].anime_id.values.tolist())
    unwatched = ~np.isin(np.asarray(anime_ids), list(watched_ids))       # get_unwatched :132-156
    keep = unwatched & meta["has_meta"].to_numpy()
    if types is not None:
        keep &= meta["Type"].isin(check_types(types)).to_numpy()
    if genres is not None:
        check_genres(genres, anime_df)
        keep &= genre_mask(meta["Genres"], genres)
    n_a = len(anime_ids)
    blocked = ~keep
    bits = np.zeros((1, (n_a + 31) // 32), np.uint32)
    nz = np.nonzero(blocked)[0]
    np.bitwise_or.at(bits[0], nz >> 5, (np.uint32(1) << (nz & 31).astype(np.uint32)))
    tU, tA = torch.as_tensor(U).cuda(), torch.as_tensor(A).cuda()
    k = _topk_count(n_recs, "model_num_recs")
    idx, p = ops.predict_topk(tU, tA, head, [int(pos[0])], k, bits.view(np.int32))
    idx, p = idx.cpu().numpy()[0], p.cpu().numpy()[0]
    ok = idx >= 0
    rows = meta.iloc[idx[ok]]
    frame = pd.DataFrame({
        "Name": rows["Name"].to_numpy(), "Prediction": p[ok], "Genres": rows["Genres"].to_numpy(),
        "Source": rows["Source"].to_numpy(), "anime_id": rows["anime_id"].to_numpy(),
        "Sypnopsis": rows["Sypnopsis"].to_numpy(), "Episodes": rows["Episodes"].to_numpy(),
        "Japanese name": rows["japanese_name"].to_numpy(), "Studios": rows["Studios"].to_numpy(),
        "Premiered": rows["Premiered"].to_numpy(), "Score": rows["Score"].to_numpy(),
        "Type": rows["Type"].to_numpy()})
    return frame
