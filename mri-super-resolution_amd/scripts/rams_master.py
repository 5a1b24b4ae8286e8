#!/usr/bin/env python3
"""``multi-image-super-resolution/master.py`` of the reference on the MI355X path: per patient the cancer slice of the DWI series
``[X, Y, Z, T]`` is cast to uint16 and multiplied by 256 (master.py:40-42), ``sample_size = 25`` random 9-acquisition subsets are
super-resolved x3 by ``RAMS(3, 32, 3, 9, 8, 12)`` through ``predict_tensor`` and averaged (:43-52), and the ADC map is
``-log(mean / (rescale(b0, 3) + eps) + eps) / b * 1e6`` (:53-57).  The 25 forwards run as ONE batched call.

Flags: the reference's three (master.py:13-15: ``--out_folder``, ``--out_img_folder``, ``--exp_name``) with the same names and
defaults, plus what the reference hard-codes or leaves undefined: ``--data_dir`` / ``--cases`` (the patient table ``cases`` that
master.py:1 imports was never published: a JSON list of ``case`` constructor arguments, as for ``scripts/master.py``),
``--weights`` (an ``.npz`` of the network's variables, ``RAMS.save_weights``: the checkpoint under ``ckpt/RED_RAMS`` that
master.py:27-33 restores is shipped WITHOUT its ``*.data-00001-of-00002`` shard and there is no TensorFlow here to read one --
without ``--weights`` the network keeps its random initialisation and the script says so), ``--sample_size`` (25) and ``--seed``
(the reference draws the subsets from the unseeded ``random`` module).  ``save_dicom`` (master.py:58-62) is replaced by
``<out_img_folder>/<exp_name>/<pt_no>/images.mat`` + ``DWI_mean.npy`` / ``ADC_mean.npy`` (DICOM export is out of scope).
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(_HERE)))
from mri_super_resolution_amd import baselines, contrast, matio  # noqa: E402
from mri_super_resolution_amd.rams import RAMS, predict_tensor  # noqa: E402

SCALE = 3           # master.py:20-26
FILTERS = 32
KERNEL_SIZE = 3
CHANNELS = 9
R = 8
N = 12
eps = 1e-7


def build_parser():
    parser = argparse.ArgumentParser(description='Superresolution of DWI/ADC maps with Multi-image SR')
    parser.add_argument('--out_folder', default='../experiments.mi/', help='directory to save the quantitative results')
    parser.add_argument('--out_img_folder', default='../output_images.mi/', help='directory to save the images')
    parser.add_argument('--exp_name', default='sr2', help='name of the experiment')
    parser.add_argument('--data_dir', default='../anon_data', help='directory of patNN_alldata / _mean_b0 .mat files')
    parser.add_argument('--cases', default=None, help='JSON file: list of {pt_id, b, cancer_loc, contralateral_loc, noise, '
                                                      'cancer_slice, acquisitions}')
    parser.add_argument('--weights', default=None, help='.npz of the RAMS variables (<layer>/v|g|b); random initialisation otherwise')
    parser.add_argument('--sample_size', type=int, default=25, help='random 9-acquisition subsets per case (master.py:44)')
    parser.add_argument('--seed', type=int, default=None, help='seed of the subset draws (the reference draws unseeded)')
    return parser


def super_resolve_case(model, case, sample_size=25, rng=random):
    """master.py:38-57 for one patient; returns ``(mean_pred [3X, 3Y] float64, adc_large, subsets)``."""
    _low_res_seq = case.dwi
    num_acq = _low_res_seq.shape[3]
    low_res_seq = _low_res_seq[:, :, case.cancer_slice, :]
    lor = np.expand_dims(low_res_seq, 0).astype('uint16')
    lor = lor * 256
    channels = model.cfg["channels"]
    if num_acq < channels:
        raise ValueError(f"{case.pt_id}: {num_acq} acquisitions, the network takes {channels} per forward")
    subsets = [rng.sample(list(range(num_acq)), channels) for _ in range(sample_size)]
    batch = np.concatenate([lor[:, :, :, inx] for inx in subsets], axis=0)                  # the 25 forwards as one batch
    sr = predict_tensor(model, batch)[:, :, :, 0]
    mean_pred = sr.double().sum(dim=0).cpu().numpy() / sample_size                          # mean_pred += img; /= sample_size
    b0 = np.asarray(case.b0[:, :, case.cancer_slice], np.float64)
    b0_scaled = baselines.rescale(b0, model.cfg["scale"], anti_aliasing=False)              # master.py:55
    adc_large = -np.log((mean_pred / (b0_scaled + eps)) + eps) / case.b
    adc_large *= 1000000
    return mean_pred, adc_large, subsets


def run(args, cases):
    rams_network = RAMS(scale=SCALE, filters=FILTERS, kernel_size=KERNEL_SIZE, channels=CHANNELS, r=R, N=N, seed=0)
    if args.weights:
        rams_network.load_weights(args.weights)
    else:
        print("note: no --weights given -- the network keeps its random initialisation (the reference's checkpoints under "
              "ckpt/RED_RAMS are shipped without their data shard; convert a complete one to .npz with RAMS.save_weights' layout)",
              file=sys.stderr)
    rng = random.Random(args.seed) if args.seed is not None else random
    os.makedirs(args.out_folder, exist_ok=True)
    summary = []
    for case in cases:
        t0 = time.perf_counter()
        mean_pred, adc_large, subsets = super_resolve_case(rams_network, case, args.sample_size, rng)
        dt = time.perf_counter() - t0
        pt_no = case.pt_id.split('-')[-1]
        out_dir = os.path.join(args.out_img_folder, args.exp_name, pt_no)
        os.makedirs(out_dir, exist_ok=True)
        np.save(os.path.join(out_dir, 'DWI_mean.npy'), mean_pred)                            # instead of save_dicom (:58-62)
        np.save(os.path.join(out_dir, 'ADC_mean.npy'), adc_large)
        matio.savemat(os.path.join(out_dir, 'images.mat'), {'DWI_mean': mean_pred, 'ADC_mean': adc_large})
        summary.append({"patient": pt_no, "shape": list(mean_pred.shape), "sample_size": args.sample_size, "seconds": dt,
                        "subsets": subsets, "out_dir": out_dir})
    with open(os.path.join(args.out_folder, args.exp_name + '.json'), 'w') as fh:
        json.dump({"weights": args.weights, "cases": summary}, fh)
    return {"cases": summary, "weights": args.weights}


def load_cases(args):
    if args.cases is None:
        return list(contrast.cases)
    with open(args.cases) as fh:
        specs = json.load(fh)
    return [contrast.case(s['pt_id'], s['b'], tuple(s['cancer_loc']), tuple(s['contralateral_loc']), tuple(s['noise']),
                          s['cancer_slice'], np.asarray(s['acquisitions']), data_dir=args.data_dir) for s in specs]


def main(argv=None):
    args = build_parser().parse_args(argv)
    cases = load_cases(args)
    if not cases:
        raise SystemExit("no cases: pass --cases cases.json (the reference's module-level `cases` list was never published)")
    out = run(args, cases)
    print(json.dumps({"cases": [{k: v for k, v in c.items() if k != "subsets"} for c in out["cases"]], "weights": out["weights"]}))
    return out


if __name__ == "__main__":
    main()
