"""Converts the reference's bundled COSMIC v3.3.1 SBS catalogue (inst/extdata/COSMIC_v3.3.1_SBS_GRCh37.csv, the
default `reference_P` of assign_signatures_ensemble_, R/helpers.R:166-169) into tests/golden/cosmic_v3.3.1_sbs.npz.
DATA only.  Run once, in the container that has /root/reference:  python tests/golden/make_cosmic_fixture.py"""
import csv
import os

import numpy as np

SRC = "/root/reference/inst/extdata/COSMIC_v3.3.1_SBS_GRCh37.csv"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cosmic_v3.3.1_sbs.npz")

if __name__ == "__main__":
    rows = list(csv.reader(open(SRC)))
    names = rows[0][1:]
    types = [r[0] for r in rows[1:]]
    P = np.array([[float(v) for v in r[1:]] for r in rows[1:]])
    assert P.shape == (96, len(names))
    np.savez_compressed(DST, P=P, signatures=np.array(names), types=np.array(types))
    print(DST, P.shape)
