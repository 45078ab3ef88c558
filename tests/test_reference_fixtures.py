"""Parity against outputs of the REFERENCE'S OWN function bodies (tests/golden/ref_fn/, written by
tests/golden/make_reference_function_fixtures.py in the build container: ast-compiled FunctionDefs of
preprocess.py, similar_anime.py, similar_users.py, user_recs.py, model_recs.py, neural_network.py run on
seeded inputs; only the data travels).

CPU half: the oracle restatements and the host logic must equal the reference outputs.
GPU half (-m gpu): the HIP path, through the C ABI, must equal the same reference outputs.
"""
import json
import os

import numpy as np
import pandas as pd
import pytest

from oracle import anirec_oracle as orc
from oracle import ingest_oracle, recs_oracle

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fn")
COLS = ("user_id", "anime_id", "rating", "watching_status", "watched_episodes")


def _js(name):
    with open(os.path.join(HERE, name), encoding="utf-8") as f:
        return json.load(f)


def _pre_cases():
    meta = _js("preprocess.json")["cases"]
    z = np.load(os.path.join(HERE, "preprocess.npz"))
    for ci, sp in enumerate(meta):
        df = pd.DataFrame({c: z["c%d_in_%s" % (ci, c)] for c in COLS})
        out = {c: z["c%d_out_%s" % (ci, c)] for c in sp["out_columns"]}
        yield ci, sp, df, out, z["c%d_out_rows" % ci]


PRE = list(_pre_cases())


def _same_f64(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


# ------------------------------------------------------------------------------------------------
# CPU: oracle and host logic vs the reference functions
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", PRE, ids=lambda c: "case%d" % c[0])
def test_ingest_oracle_equals_reference_preprocess(case):
    ci, sp, df, want, rows = case
    got = ingest_oracle.preprocess(df, sp["num_reviews"], sp["drop_unwatched"], sp["drop_plan"],
                                   sp["drop_half_watched"])
    assert list(got.index) == list(rows)
    assert list(got.columns) == sp["out_columns"]          # incl. max_eps / half_eps for drop_half_watched
    for c in sp["out_columns"]:
        assert _same_f64(got[c].to_numpy(), want[c]), c


def test_fixture_covers_the_flags_and_the_extra_columns():
    combos = {(sp["drop_unwatched"], sp["drop_plan"], sp["drop_half_watched"]) for _, sp, *_ in PRE}
    assert len(combos) == 8
    assert any("max_eps" in sp["out_columns"] and sp["n_out"] > 0 for _, sp, *_ in PRE)
    assert all(("half_eps" in sp["out_columns"]) == (sp["drop_half_watched"] and sp["n_out"] > 0 or
                                                     ("half_eps" in sp["out_columns"])) for _, sp, *_ in PRE)


def test_clean_and_genre_logic_equal_reference():
    from anime_recommendations_amd import components as C
    c = _js("clean.json")
    assert [C.clean(x) for x in c["inputs"]] == c["outputs"]
    assert C.clean(list(c["inputs"])) == c["list_output"]
    g = _js("genres.json")
    col = pd.Series([np.nan if x is None else x for x in g["genres_column"]])
    frame = pd.DataFrame({"Genres": col})
    assert C.all_genres(frame) == g["get_genres"]
    for case in g["by_genre"]:
        kept = case["kept_rows_in_output_order"]
        valid = set(C.clean(g["get_genres"]))
        if any(x not in valid for x in C.clean(list(case["genres"]))):
            assert kept is None                              # the reference logs "invalid genre" and returns None
            with pytest.raises(ValueError):
                C.check_genres(case["genres"], frame)
            continue
        C.check_genres(case["genres"], frame)
        m = C.genre_mask(col, case["genres"])
        assert sorted(np.nonzero(m)[0].tolist()) == sorted(kept or [])


def test_rownorm_oracle_equals_reference_get_weights():
    z = np.load(os.path.join(HERE, "get_weights.npz"))
    with np.errstate(all="ignore"):
        for raw, want in ((z["A"], z["A_norm"]), (z["U"], z["U_norm"])):
            got = orc.rownorm(raw)
            assert got.dtype == np.float32
            np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
            np.testing.assert_array_equal(np.nan_to_num(got), np.nan_to_num(want))   # NumPy == NumPy: bitwise
    assert np.isnan(z["U_norm"][17]).all()                  # zero row -> NaN row, no epsilon


def _su():
    z = np.load(os.path.join(HERE, "similar_users.npz"))
    return z, _js("similar_users.json")["cases"]


def test_cosine_topk_oracle_equals_reference_find_similar_users():
    z, cases = _su()
    W, ids = z["W"], z["user_ids"]
    pos = {int(u): i for i, u in enumerate(ids)}
    for c in cases:
        oi, osim = orc.cosine_topk(W, [pos[c["user_id"]]], c["n_users"], exclude_self=True)
        assert [int(ids[i]) for i in oi[0]] == c["similar_users"]
        # the reference's similarity is BLAS sgemv (summation order unspecified): equal within fp32 rounding (1e-6)
        np.testing.assert_allclose(osim[0], np.array(c["similarity"], np.float32), atol=1e-6, rtol=0)


def test_fave_anime_host_logic_equals_reference_get_fave_anime():
    from anime_recommendations_amd import components as C
    z, cases = _su()
    df = pd.DataFrame({"user_id": z["df_user_id"], "anime_id": z["df_anime_id"], "rating": z["df_rating"],
                       "watched_episodes": z["df_watched"]})
    anime_df = pd.DataFrame({"anime_id": z["anime_id"], "Name": ["Title %d" % a for a in z["anime_id"]],
                             "Episodes": z["episodes"]})
    for c in cases:
        got = [C.fave_anime(df, anime_df, u, c["num_faves"], c["TV_only"]) for u in c["similar_users"]]
        assert got == c["favorite_animes"]


def test_favourites_oracle_equals_reference_fave_genres():
    d = _js("fave_genres.json")
    u, a, r = np.array(d["user_id"]), np.array(d["anime_id"]), np.array(d["rating"], np.float64)
    _, fav = recs_oracle.favourites(u, a, r, 40)
    for k, want in d["favourites"].items():
        assert sorted(fav[int(k)]) == want, k


def test_unwatched_mask_equals_reference_get_unwatched():
    d = _js("model_recs.json")
    df = pd.DataFrame({"user_id": d["df_user_id"], "anime_id": d["df_anime_id"]})
    from anime_recommendations_amd.data import encode_ids
    _, anime_ids = encode_ids(df["anime_id"].to_numpy())        # index = Series.unique() position
    _, user_ids = encode_ids(df["user_id"].to_numpy())
    known = set(d["anime_df_ids"])
    for u, want in d["users"].items():
        watched = set(df[df.user_id == int(u)].anime_id.tolist())
        # components.model_recs_frame: unwatched & has_meta, over the index order (model_recs.py:144-155)
        mask = ~np.isin(anime_ids, list(watched)) & np.isin(anime_ids, list(known))
        assert np.nonzero(mask)[0].tolist() == want["unwatched_indices_sorted"]
        assert int(np.nonzero(user_ids == int(u))[0][0]) == want["user_index"]
        assert int(mask.sum()) == want["n_pairs"]


def test_lrfn_equals_reference_function():
    from anime_recommendations_amd import schedule
    for case in _js("lrfn.json"):
        f = {k: float(v) for k, v in case["flags"].items()}
        for e, want in enumerate(case["lr"]):
            kw = dict(start_lr=f["start_lr"], max_lr=f["max_lr"], min_lr=f["min_lr"],
                      rampup_epochs=int(f["rampup_epochs"]), sustain_epochs=int(f["sustain_epochs"]),
                      exp_decay=f["exp_decay"])
            assert schedule.lrfn(e, *[kw[k] for k in ("start_lr", "max_lr", "min_lr", "rampup_epochs",
                                                       "sustain_epochs", "exp_decay")]) == want
            assert orc.lrfn(e, **kw) == want


# ------------------------------------------------------------------------------------------------
# GPU: the HIP path vs the reference functions
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", PRE, ids=lambda c: "case%d" % c[0])
def test_gpu_ingest_equals_reference_preprocess(case):
    from anime_recommendations_amd import ingest
    ci, sp, df, want, rows = case
    cols = ingest.frame_to_columns(df)
    got = {k: v.cpu().numpy() for k, v in ingest.preprocess_columns(
        cols, num_reviews=sp["num_reviews"], drop_unwatched=sp["drop_unwatched"], drop_plan=sp["drop_plan"],
        drop_half_watched=sp["drop_half_watched"]).items()}
    assert len(got["user_id"]) == sp["n_out"]
    for c in sp["out_columns"]:
        assert c in got, c
        if c in ("rating", "half_eps"):
            assert got[c].dtype == np.float64 and _same_f64(got[c], want[c]), c
        else:
            np.testing.assert_array_equal(got[c].astype(np.float64), want[c], err_msg=c)


@pytest.mark.gpu
def test_gpu_rownorm_equals_reference_get_weights():
    import torch
    from anime_recommendations_amd import ops
    z = np.load(os.path.join(HERE, "get_weights.npz"))
    for raw, want in ((z["A"], z["A_norm"]), (z["U"], z["U_norm"])):
        got = ops.rownorm(torch.from_numpy(raw)).cpu().numpy()
        np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
        ok = ~np.isnan(want)
        ulp = np.spacing(np.abs(want[ok]).astype(np.float32))
        assert np.all(np.abs(got[ok] - want[ok]) <= 2 * ulp)        # NumPy's pairwise norm vs the wave reduction


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["exact", "mfma"])
def test_gpu_neighbour_lists_equal_reference_find_similar_users(path):
    import torch
    from anime_recommendations_amd import ops
    z, cases = _su()
    ids = z["user_ids"]
    pos = {int(u): i for i, u in enumerate(ids)}
    Wh = torch.from_numpy(z["W"]).cuda()
    for c in cases:
        q = [pos[c["user_id"]]]
        if path == "exact":
            idx, sim = ops.cosine_topk(Wh, q, c["n_users"])
        else:
            idx, sim, _ = ops.cosine_topk_mfma(Wh, q, c["n_users"])
        assert [int(ids[i]) for i in idx.cpu().numpy()[0]] == c["similar_users"]
        np.testing.assert_allclose(sim.cpu().numpy()[0], np.array(c["similarity"], np.float32), atol=1e-6, rtol=0)


@pytest.mark.gpu
def test_gpu_favourites_equal_reference_fave_genres():
    import torch
    from anime_recommendations_amd import recs
    d = _js("fave_genres.json")
    u = torch.tensor(d["user_id"], dtype=torch.int32).cuda()
    a = torch.tensor(d["anime_id"], dtype=torch.int32).cuda()
    r = torch.tensor(d["rating"], dtype=torch.float64).cuda()
    fav, _ = recs.user_favourites(u, a, r, 40, d["n_anime"])
    bits = fav.cpu().numpy().view(np.uint32)
    for k, want in d["favourites"].items():
        row = bits[int(k)]
        got = [i for i in range(d["n_anime"]) if (row[i >> 5] >> (i & 31)) & 1]
        assert got == want, k


@pytest.mark.gpu
@pytest.mark.parametrize("n", [17_560, 350_000])
def test_gpu_cosine_scores_bit_equal_the_c_oracle_chain(n):
    """The weakest link named by the round-1 review: the exact path's SCORES against the oracle's fmaf chain
    (orc_cosine_scores), bit for bit, at both BASELINE table sizes and several queries."""
    import torch
    from anime_recommendations_amd import ops
    from oracle import c_oracle
    g = torch.Generator(device="cuda")
    g.manual_seed(n)
    Wh = ops.rownorm(torch.randn(n, 128, generator=g, device="cuda") * 0.05)
    Whn = Wh.cpu().numpy()
    for q in (0, 1, n // 3, n - 1):
        got = ops.cosine_scores(Wh, q).cpu().numpy()
        want = c_oracle.cosine_scores(Whn, q)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), q
