#!/usr/bin/env python3
"""``master.py`` of the reference (implicit-neural-representations/master.py) on the MI355X path: per patient and gradient
direction a small SIREN is fitted to the K acquisitions of one 2-D slice (weighted loss, one optimizer step per acquisition
per epoch), the last ``--seg`` epochs are ensembled at x1 and x``--scale``, ADC maps are derived and the lesion contrast of
every image goes to ``<out_folder>/<exp_name>.csv`` (``seed,patient,direction,image,metric,performance``, master.py:62).

Flags: the reference's twelve (master.py:25-38) with the same names, defaults and meaning, plus what the reference
hard-codes or leaves undefined: ``--data_dir`` (``../anon_data``), ``--cases`` (a JSON list of ``case`` constructor
arguments -- the module-level ``cases`` list the reference imports was never published) and ``--experiment``
(an ``experiments/sr1_exp_N.txt`` file: ``steps`` -> total_steps, ``depth`` -> hidden_layers, ``hidden`` ->
hidden_features; ``focus = wide`` widens the ROI to the whole slice, ``weight = True`` asks for acceptance weights, which
need ``--erd``).  ``--erd 1|2`` (agglomerative-clustering outlier rejection, master.py:79-93) and the DICOM export
(master.py:228-247) are outside the build's scope: the first is refused, the second is replaced by ``images.mat``.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(_HERE)))
from mri_super_resolution_amd import baselines, contrast, drivers, erd, matio, reports  # noqa: E402
from mri_super_resolution_amd.contrast import calc_adc, calculate_contrast, minmax_normalize  # noqa: E402


def build_parser():
    parser = argparse.ArgumentParser(description='Superresolution of DWI/ADC maps enhanced with AutoERD')
    parser.add_argument('--out_folder', default='../experiments/', help='directory to save the quantitative results')
    parser.add_argument('--out_img_folder', default='../output_images/', help='directory to save the images')
    parser.add_argument('--total_steps', type=int, default=3000, help='total steps for training')
    parser.add_argument('--seg', type=int, default=150, help='the epochs to wait until ensamble calculation')
    parser.add_argument('--hidden_layers', type=int, default=6, help='depth of the network')
    parser.add_argument('--hidden_features', type=int, default=64, help='number of neurons on each layer')
    parser.add_argument('--ROI_begin', type=int, default=40, help='Beginning pixel of the Region of Interest')
    parser.add_argument('--ROI_end', type=int, default=100, help='Last pixel that includes the Region of Interest')
    parser.add_argument('--learning_rate', type=float, default=0.0003, help='learning rate')
    parser.add_argument('--scale', type=int, default=3, help='scaling factor super-resolution')
    parser.add_argument('--exp_name', default='sr2', help='name of the experiment')
    parser.add_argument('--repeat_time', type=int, default=1, help='run the experiment multiple times to account for randomness')
    parser.add_argument('--erd', type=int, default=0, help='AutoERD before training [0 = no ERD; 1, 2: not in this build]')
    parser.add_argument('--data_dir', default='../anon_data', help='directory of patNN_alldata / _mean_b0 / _ERD .mat files')
    parser.add_argument('--cases', default=None, help='JSON file: list of {pt_id, b, cancer_loc, contralateral_loc, noise, '
                                                      'cancer_slice, acquisitions}')
    parser.add_argument('--experiment', default=None, help='experiments/sr1_exp_N.txt to take steps / depth / hidden / focus from')
    return parser


def read_experiment(path):
    """``key = value`` lines of experiments/sr1_exp_N.txt -> dict."""
    out = {}
    with open(path) as fh:
        for line in fh:
            if '=' in line:
                k, v = line.split('=', 1)
                out[k.strip()] = v.strip()
    return out


def apply_experiment(args, exp):
    if 'steps' in exp:
        args.total_steps = int(exp['steps'])
    if 'depth' in exp:
        args.hidden_layers = int(exp['depth'])
    if 'hidden' in exp:
        args.hidden_features = int(exp['hidden'])
    if exp.get('focus') == 'wide':
        args.ROI_begin, args.ROI_end = 0, None          # whole slice (resolved per case)
    if exp.get('input', '2') != '2' or exp.get('output', 'dwi') != 'dwi' or exp.get('style', 'directional') != 'directional':
        raise SystemExit(f"experiment {exp}: only style=directional, input=2, output=dwi exist in the reference's driver")
    if exp.get('weight', 'False') == 'True' and not args.erd:
        print("note: weight = True needs acceptance weights from --erd; running with unit weights", file=sys.stderr)
    return args


def run(args, cases):
    os.makedirs(args.out_folder, exist_ok=True)
    csv = reports.ContrastCsv(os.path.join(args.out_folder, args.exp_name + '.csv'))
    directions = ['x', 'y', 'z']
    summary = []
    for seed in range(args.repeat_time):
        torch.manual_seed(seed)
        for case in cases:
            print(case.pt_id)
            _slice, b = case.cancer_slice, case.b
            r0 = args.ROI_begin
            r1 = args.ROI_end if args.ROI_end is not None else min(case.dwi.shape[:2])
            b0 = np.asarray(case.b0[r0:r1, r0:r1, _slice], np.float64)
            pt_no = case.pt_id.split('-')[-1]
            if args.erd:      # master.py:77-93: per-pixel two-cluster outlier rejection -> case.accept (one device launch)
                print('Conducting Auto-ERD with Agglomerative Clustering...')
                erd.apply_auto_erd(case, args.erd, r0, r1)
            contrast_fn = lambda im: calculate_contrast(case, 1, im, r0)
            acc = {}
            for direction in range(3):
                print(f'Training for {directions[direction]} direction...')
                ends = np.cumsum(case.acquisitions)
                starts = ends - case.acquisitions
                acqs = range(int(starts[direction]), int(ends[direction]))
                imgs = [np.asarray(case.dwi[r0:r1, r0:r1, _slice, a], np.float32) for a in acqs]
                accepts = [np.asarray(case.accept[r0:r1, r0:r1, _slice, a], np.float32) for a in acqs]
                sum_image = sum(i.astype(np.float64) for i in imgs)
                sum_accepted = sum(i.astype(np.float64) * a for i, a in zip(imgs, accepts))
                sum_accepts = sum(a.astype(np.float64) for a in accepts)
                accepted_mean = sum_accepted / (sum_accepts + contrast.eps)                # master.py:112
                direction_mean = sum_image / len(imgs)
                fit = drivers.fit_slice_ensemble(imgs, accepts, total_steps=args.total_steps, seg=args.seg, scale=args.scale,
                                                 hidden_features=args.hidden_features, hidden_layers=args.hidden_layers,
                                                 lr=args.learning_rate, seed=None, divide_by=args.seg)
                orig = direction_mean.copy()                                               # dataset.mean (nn_mri.py:196)
                erd_img = accepted_mean
                out_img = fit["predicted"]
                large_out = fit["large"]
                out_img = out_img - out_img.min()
                large_out = large_out - large_out.min()
                norm_out_img = minmax_normalize(out_img, direction_mean)
                norm_large_out = minmax_normalize(large_out, direction_mean)
                b0_scaled = baselines.rescale(b0, args.scale, anti_aliasing=False)          # master.py:175
                cur = {'orig': orig, 'erd_img': erd_img, 'out_img': out_img, 'large_out': large_out,
                       'norm_out_img': norm_out_img, 'norm_large_out': norm_large_out,
                       'adc_orig': calc_adc(orig, b0, b), 'adc_erd': calc_adc(erd_img, b0, b),
                       'adc_superres': calc_adc(out_img, b0, b), 'adc_large': calc_adc(large_out, b0_scaled, b),
                       'adc_norm': calc_adc(norm_out_img, b0, b), 'adc_large_norm': calc_adc(norm_large_out, b0_scaled, b)}
                images = {'mean': cur['orig'], 'ERD': cur['erd_img'], 'superres': cur['out_img'],
                          'superres_n': cur['norm_out_img'], 'ADC_orig': cur['adc_orig'], 'ADC_ERD': cur['adc_erd'],
                          'ADC_super': cur['adc_superres'], 'ADC_super_norm': cur['adc_norm']}
                csv.rows(seed, pt_no, directions[direction], images, contrast_fn)
                # master.py:197-209 as written: from the second direction on every image is added TO ITSELF (not to a
                # running sum), so the "mean" rows below are 2/3 of the last direction's images -- kept for parity
                acc = {k: (v + v if direction else v) for k, v in cur.items()}
                summary.append({"seed": seed, "patient": pt_no, "direction": directions[direction],
                                "optimizer_steps": fit["optimizer_steps"], "seconds": fit["seconds"]})
            acc = {k: v / len(directions) for k, v in acc.items()}
            out_dir = os.path.join(args.out_img_folder, args.exp_name, pt_no)
            os.makedirs(out_dir, exist_ok=True)
            matio.savemat(os.path.join(out_dir, 'images.mat'), {                            # instead of save_dicom (:228-247)
                'DWI_mean': acc['orig'] * contrast.mag, 'DWI_erd': acc['erd_img'] * contrast.mag,
                'DWI_super': acc['large_out'] * contrast.mag, 'DWI_super_norm': acc['norm_large_out'] * contrast.mag,
                'ADC_mean': acc['adc_orig'], 'ADC_erd': acc['adc_erd'], 'ADC_super': acc['adc_superres'],
                'ADC_large': acc['adc_large'], 'ADC_norm_super': acc['adc_norm'], 'ADC_norm_super_large': acc['adc_large_norm']})
            images = {'mean': acc['orig'], 'ERD': acc['erd_img'], 'superres': acc['out_img'], 'superres_n': acc['norm_out_img'],
                      'ADC_orig': acc['adc_orig'], 'ADC_ERD': acc['adc_erd'], 'ADC_super': acc['adc_superres'],
                      'ADC_super_norm': acc['adc_norm']}
            csv.rows(seed, pt_no, 'mean', images, contrast_fn)
    return {"csv": csv.path, "fits": summary}


def load_cases(args):
    if args.cases is None:
        return list(contrast.cases)
    with open(args.cases) as fh:
        specs = json.load(fh)
    return [contrast.case(s['pt_id'], s['b'], tuple(s['cancer_loc']), tuple(s['contralateral_loc']), tuple(s['noise']),
                          s['cancer_slice'], np.asarray(s['acquisitions']), data_dir=args.data_dir) for s in specs]


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.experiment:
        args = apply_experiment(args, read_experiment(args.experiment))
    cases = load_cases(args)
    if not cases:
        raise SystemExit("no cases: pass --cases cases.json (the reference's module-level `cases` list was never published)")
    out = run(args, cases)
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
