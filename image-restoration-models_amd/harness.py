"""Benchmark-harness aggregate of the reference (scripts/tests.py:389-424): per image PSNR, SSIM and
inference time, per (dataset, model) their mean and standard deviation, one CSV row per combination with
the reference's column names.  Dataset file IO is out of scope - `loader` is any iterable yielding
(input_uint8_hwc, target_uint8_hwc, name), e.g. `synthetic_loader`."""
from __future__ import annotations

import csv
import os

import numpy as np

from . import synth
from .utils import calculate_metrics, get_model_prediction, get_model_total_parameters

COLUMNS = ['Task', 'Type', 'Dataset', 'Sigma', 'Model', 'Model_Params', 'PSNR', 'SSIM', 'Std_PSNR', 'Std_SSIM',
           'Avg_Time_ms', 'Std_Time_ms']


def synthetic_loader(n_images: int, h: int = 720, w: int = 1280, c: int = 3, seed_base: int = 1000, blur: int = 15):
    """Same yield shape as src/data_loaders.py's generators, synthetic GoPro-shaped frames (synth.py)."""
    for i in range(n_images):
        inp, tgt = synth.synth_image_pair(i, h, w, c, seed_base=seed_base, blur=blur)
        yield inp, tgt, f"synthetic_{i:04d}.png"


def evaluate(model, loader, device, patch_config: dict, *, task: str, subtask: str, dataset: str, model_name: str,
             sigma='N/A', need_degradation=False, noise_level=None, with_ssim=True, skip_failed=True) -> dict:
    """One results_table row (scripts/tests.py:399-412).

    The reference's loop lets any exception of a frame end the whole sweep (only a missing weight file is caught,
    tests.py:48-50).  Here a frame that raises is recorded and skipped (report the failed image
    ids instead of losing the run): the row's extra key 'Failed' lists (name, error) pairs and the statistics are
    taken over the frames that ran; `skip_failed=False` restores the reference's behaviour (the exception
    propagates).  Out-of-memory errors always propagate (src/utils.py:91-93 reports them to the caller)."""
    psnr_list, ssim_list, time_list, failed = [], [], [], []
    for input_img, target_img, name in loader:
        try:
            pred, ms = get_model_prediction(model, input_img, device, **patch_config, need_degradation=need_degradation,
                                            noise_level=noise_level)
            if with_ssim:
                p, s = calculate_metrics(pred, target_img)
            else:
                from .utils import psnr
                p, s = psnr(target_img, pred, 255 if pred.dtype == np.uint8 else 65535), float('nan')
        except Exception as e:                                  # noqa: BLE001 (reported, not swallowed)
            if not skip_failed or "out of memory" in str(e).lower():
                raise
            failed.append((name, f"{type(e).__name__}: {e}"))
            print(f"[harness] {model_name} on {dataset}: frame {name} failed ({type(e).__name__}: {e}); skipped")
            continue
        psnr_list.append(p)
        ssim_list.append(s)
        time_list.append(ms)
    row = aggregate(psnr_list, ssim_list, time_list, task=task, subtask=subtask, dataset=dataset, sigma=sigma,
                    model_name=model_name, params=get_model_total_parameters(model))
    row['Failed'] = failed
    return row


def aggregate(psnr_list, ssim_list, time_list, *, task, subtask, dataset, sigma, model_name, params) -> dict:
    if not psnr_list:                      # every frame failed: an empty row, not a numpy warning
        psnr_list = ssim_list = time_list = [float('nan')]
    return {'Task': task.capitalize(), 'Type': subtask.capitalize(), 'Dataset': dataset, 'Sigma': sigma,
            'Model': model_name, 'Model_Params': params, 'PSNR': np.mean(psnr_list), 'SSIM': np.mean(ssim_list),
            'Std_PSNR': np.std(psnr_list), 'Std_SSIM': np.std(ssim_list), 'Avg_Time_ms': np.mean(time_list),
            'Std_Time_ms': np.std(time_list)}


def save_results(rows: list, out_dir: str = 'results', file_name: str = 'results_summary.csv') -> str:
    """CSV with the reference's columns (scripts/tests.py:415-424; written with the csv module, not pandas)."""
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, file_name)
    with open(path, 'w', newline='') as f:
        wr = csv.DictWriter(f, fieldnames=COLUMNS, extrasaction='ignore')
        wr.writeheader()
        for r in rows:
            wr.writerow(r)
    return path
