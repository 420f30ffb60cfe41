"""Command line: hierarchical Linear CorEx on a CSV file, fitted on MI355X.

The product-level caller of the fit path in the reference is the `__main__` block of vis_corex.py
(:414-558).  This module reproduces the part of it that drives the path - the options, the CSV
reader, the layer stacking (layer k+1 is fitted on `transform()` of layer k, :530-545; only layer 0
receives the missing-value sentinel, :536), the pickled models `layer_<l>.dat` (:549) and the text
summaries - so the README commands run end to end on the GPU:

    python -m linearcorex_amd.cli tests/data/test_big5.csv --layers=5,1 --verbose=1 --no_row_names -o big5
    python -m linearcorex_amd.cli tests/data/adni_blood.csv --layers=30,5,1 --missing=-1e6 --verbose=1 -o adni

Plots and graph export (matplotlib / networkx / graphviz, vis_corex.py:27-411) are presentation code
outside the accelerated path and are not provided.
"""
from __future__ import annotations

import argparse
import csv
import os
import pickle
import sys
from time import time

import numpy as np


def build_parser():
    p = argparse.ArgumentParser(
        prog="python -m linearcorex_amd.cli",
        description="It is assumed that the first row and first column of the data CSV file are labels. "
                    "Use options to indicate otherwise.")
    g = p.add_argument_group("Input Data Format Options")
    g.add_argument("-t", "--no_column_names", action="store_true", dest="nc", default=False,
                   help="Data starts on the first row; variables are numbered 0,1,2...")
    g.add_argument("-f", "--no_row_names", action="store_true", dest="nr", default=False,
                   help="Data starts on the first column.")
    g.add_argument("-m", "--missing", type=float, default=-1e6, help="Treat this value as missing data.")
    g.add_argument("-d", "--delimiter", default=",", help="Separator between entries, default ','.")
    g.add_argument("-g", "--gaussianize", default="standard", help="Try 'outliers' if there are long tails.")
    g = p.add_argument_group("CorEx Options")
    g.add_argument("-l", "--layers", default="2,1", help="Units per layer: 5,3,1 = 5 at layer 1, 3 at layer 2, 1 at layer 3")
    g.add_argument("-w", "--max_iter", type=int, default=10000, help="Max number of iterations.")
    g.add_argument("-a", "--additive", action="store_false", dest="additive", default=True,
                   help="Turn off the non-synergy constraint (discourage_overlap=False).")
    g.add_argument("--seed", type=int, default=None, help="Seed of the random initial weights (reference: unseeded).")
    g.add_argument("--dtype", choices=["float32", "float64"], default="float32",
                   help="Working precision on the device (the reference computes in float32).")
    g = p.add_argument_group("Computational Options")
    g.add_argument("-n", "--gpu", action="store_true", default=False,
                   help="Accepted for compatibility: this build always runs on the GPU.")
    g.add_argument("--device", type=int, default=None, help="HIP device index.")
    g.add_argument("--f32_gemm", choices=["mfma", "split"], default=None,
                   help="float32 fits: arithmetic of the two passes over X - 'mfma' (float32 MFMA, the default) or 'split' (exact "
                        "three-way bf16 split on the bf16 matrix pipe: same results to float32 rounding, faster on large inputs).")
    g = p.add_argument_group("Output Options")
    g.add_argument("-o", "--output", default="corex_output", help="A directory to put all output files.")
    g.add_argument("-v", "--verbose", type=int, default=0, help="Verbosity 0, 1, 2.")
    g.add_argument("-e", "--edges", type=int, dest="max_edges", default=200, help="(graphs are not produced; accepted)")
    g.add_argument("-q", "--regraph", action="store_true", default=False,
                   help="Don't re-run corex, reload the pickled layers and re-generate the text outputs.")
    p.add_argument("data_file")
    return p


def load_table(path, delimiter=",", no_column_names=False, no_row_names=False):
    """CSV reader with the reference's conventions (vis_corex.py:496-512): header row and label column
    unless told otherwise; universal newlines (test_big5.csv has CR-only line endings)."""
    first = 0 if no_row_names else 1
    with open(path, "r", newline=None) as fh:
        reader = csv.reader(fh, delimiter=delimiter)
        names = None if no_column_names else next(reader)[first:]
        rows, labels = [], ([] if not no_row_names else None)
        for row in reader:
            if not row:
                continue
            if labels is not None:
                labels.append(row[0])
            rows.append(row[first:])
    try:
        x = np.array(rows, dtype=float)
    except ValueError as e:
        raise SystemExit("Incorrect data format.\nCheck that you've correctly specified whether there is a header row "
                         "or a label column, and the delimiter.\nMissing values should be given as a numeric value.\n%s" % e)
    return x, names, labels


def fit_layers(x, layers, corex_factory, missing=-1e6, verbose=0):
    """vis_corex.py:530-545.  corex_factory(n_hidden, layer_index, **kw) -> unfitted model."""
    models, x_prev = [], x
    for l, n_hidden in enumerate(layers):
        if verbose:
            print("Layer ", l)
        if l == 0:
            t0 = time()
            models.append(corex_factory(n_hidden, l, missing_values=missing).fit(x))
            print('Time for first layer: %0.2f' % (time() - t0))
        else:
            # the layer below was fitted on x_prev, which is still resident on its device: no second upload (:542)
            resident = getattr(models[-1], "transform_fitted", lambda: None)()
            x_prev = models[-1].transform(x_prev) if resident is None else resident
            models.append(corex_factory(n_hidden, l).fit(x_prev))
    return models


def _open(path, mode):
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    return open(path, mode)


def write_summaries(model, x, prefix, column_label=None, row_label=None):
    """Text dumps of a one-layer representation (the non-graphical part of vis_rep, vis_corex.py:27-44,
    :55-90): groups.txt / groups_no_overlaps.txt (variables per factor by |weight|), summary.txt (TC per
    factor), labels.txt (latent factors per sample)."""
    nv = model.ws.shape[1]
    if column_label is None:
        column_label = [str(i) for i in range(nv)]
    if row_label is None:
        row_label = [str(i) for i in range(len(x))]
    mom = model.moments
    tcs, mis, ws = np.asarray(mom["TCs"]), model.mis, model.ws
    explained = np.asarray(mom["rho"]) * np.asarray(mom["X_i Z_j"]).T > 0.05     # >= 5 % of variance (:37-38)
    owner = np.argmax(np.abs(ws), axis=0)
    with _open(os.path.join(prefix, "summary", "groups.txt"), "w") as f, \
            _open(os.path.join(prefix, "summary", "groups_no_overlaps.txt"), "w") as g, \
            _open(os.path.join(prefix, "summary", "summary.txt"), "w") as h:
        h.write("Group, TC\n")
        f.write("variable, weight, MI\n")
        g.write("variable, weight, MI\n")
        for j in range(ws.shape[0]):
            head = "Group num: %d, TC(X;Y_j): %0.6f\n" % (j, tcs[j])
            f.write(head)
            g.write(head)
            h.write("%d, %0.6f\n" % (j, tcs[j]))
            members = np.where(explained[j])[0]
            for i in members[np.argsort(-np.abs(ws[j, members]))]:
                f.write("%s, %.3f, %.3f\n" % (column_label[i], ws[j, i], mis[j, i]))
            mine = np.where(owner == j)[0]
            for i in mine[np.argsort(-np.abs(ws[j, mine]))]:
                g.write("%s, %.3f, %.3f\n" % (column_label[i], ws[j, i], mis[j, i]))
    labels = model.transform(x)
    with _open(os.path.join(prefix, "summary", "labels.txt"), "w") as f:
        for name, row in zip(row_label, labels):
            f.write(name + "," + ",".join("%.6f" % v for v in row) + "\n")
    with _open(os.path.join(prefix, "summary", "convergence.txt"), "w") as f:
        for tc in model.history.get("TC", []):
            f.write("%.8f\n" % tc)


def main(argv=None, corex_factory=None):
    opt = build_parser().parse_args(argv)
    np.set_printoptions(precision=3, suppress=True)
    layers = [int(t) for t in opt.layers.split(",")]
    if layers[-1] != 1:
        layers.append(1)        # last layer has one unit so that the hierarchy is connected (:489-490)
    x, names, labels = load_table(opt.data_file, opt.delimiter, opt.nc, opt.nr)
    if opt.verbose:
        print('\nData summary: X has %d rows and %d columns' % x.shape)
        if names is not None:
            print('Variable names are: ' + ','.join(map(str, list(enumerate(names)))))
        print('Getting CorEx results')
    if corex_factory is None:
        from .corex import Corex

        def corex_factory(n_hidden, layer, **kw):
            return Corex(n_hidden=n_hidden, verbose=opt.verbose, gaussianize=opt.gaussianize,
                         discourage_overlap=opt.additive, gpu=True, max_iter=opt.max_iter, seed=opt.seed,
                         dtype=np.dtype(opt.dtype), device=opt.device, f32_gemm=opt.f32_gemm, **kw)
    if not opt.regraph:
        models = fit_layers(x, layers, corex_factory, missing=opt.missing, verbose=opt.verbose)
        for l, mdl in enumerate(models):
            print('TC at layer %d is: %0.3f' % (l, mdl.tc))
            with _open(os.path.join(opt.output, 'layer_%d.dat' % l), 'wb') as fh:
                pickle.dump(mdl, fh)
    else:
        models = []
        for l in range(len(layers)):
            with open(os.path.join(opt.output, 'layer_%d.dat' % l), 'rb') as fh:
                models.append(pickle.load(fh))
    print('Variable groups in summary/groups.txt')
    print('Latent factors for each sample in summary/labels.txt')
    write_summaries(models[0], x, opt.output, column_label=names, row_label=labels)
    # the text half of vis_hierarchy (vis_corex.py:219-225): total and per-factor TC of every layer (the graph half is out of scope)
    with _open(os.path.join(opt.output, "summary", "higher_layer_group_tcs.txt"), "w") as f:
        for j, mdl in enumerate(models):
            f.write('At layer: %d, Total TC: %0.3f\n' % (j, mdl.tc))
            f.write('Individual TCS:' + str(mdl.tcs) + '\n')
    return models


if __name__ == "__main__":
    main(sys.argv[1:])
