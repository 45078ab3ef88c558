"""Copies the DATA fixtures the reference holds for the hot path into tests/golden/.

Run in the build container only (reads /root/reference, which never travels):
    python tests/golden/make_reference_fixtures.py

* anime_nn_history.csv  (reference: figure_file/anime_nn_history.csv) — the Keras
  History CSV of the author's run; its ``lr`` column is the one numeric
  known-answer for the path (pins ``lrfn``), the other columns pin the CSV schema.
* header lines + value columns of the three example output CSVs (format fixtures:
  column names, row count, descending order).
"""
import csv
import json
import os

REF = "/root/reference/figure_file"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    with open(os.path.join(REF, "anime_nn_history.csv")) as f:
        rows = list(csv.reader(f))
    with open(os.path.join(OUT, "anime_nn_history.csv"), "w", newline="") as f:
        csv.writer(f).writerows(rows)

    fmt = {}
    for name, valcol in [("User_153695_similar_users.csv", "similarity"),
                         ("anime_similar_to_SilentMobius.csv", "Similarity"),
                         ("User_ID_153695_model_recs.csv", "Prediction")]:
        # the Möbius file name is stored NFD-normalised on disk: match by prefix
        real = [n for n in os.listdir(REF) if n.startswith(name[:22])][0]
        with open(os.path.join(REF, real), newline="") as f:
            r = list(csv.DictReader(f))
        fmt[name] = {"columns": list(r[0].keys()), "n_rows": len(r),
                     "value_column": valcol, "values": [float(x[valcol]) for x in r]}
    with open(os.path.join(OUT, "reference_output_formats.json"), "w") as f:
        json.dump(fmt, f, indent=1, ensure_ascii=False)

    # the CLI surface of the four hot-path components and of the preprocess step before them: flag names and which are booleans
    # (reference <c>/<c>.py argparse blocks + <c>/MLproject parameter lists)
    import re
    flags = {}
    for c in ("neural_network", "similar_anime", "similar_users", "model_recs", "preprocess"):
        src = open("/root/reference/%s/%s.py" % (c, c)).read()
        found = re.findall(r'add_argument\(\s*"--(\w+)",\s*type=([^,]+),', src)
        ml = open("/root/reference/%s/MLproject" % c).read()
        params = re.findall(r"^      (\w+):\s*$", ml, flags=re.M)
        flags[c] = {"flags": [n for n, _ in found],
                    "bool_flags": [n for n, t in found if "strtobool" in t],
                    "mlproject_parameters": params}
    with open(os.path.join(OUT, "component_flags.json"), "w") as f:
        json.dump(flags, f, indent=1)


if __name__ == "__main__":
    main()
