"""NTIRE 2026 Image SR x4 test runner -- counterpart of the reference's test.py (same CLI, same
select_model / run / main structure, `--model_id 29` only).  Differences: CUDA-event timing is guarded
so argument parsing and model selection work on a GPU-less host, and the logger is the stdlib one
(the reference's utils/ helpers need cv2/matplotlib, which are outside the hot path)."""
import argparse
import logging
import os
import time
from pprint import pprint

import torch


def select_model(args, device):
    model_id = args.model_id
    if model_id == 29:
        from models.team29_FreqFusion import main as FreqFusion
        name = f"{model_id:02}_FreqFusion_team29"
        model_path = os.path.join("checkpoints", "phase5_single_gpu", "championship_sr_phase5_single_gpu",
                                  "best_epoch0050_psnr30.05.pth")
        model_func = FreqFusion
    else:
        raise NotImplementedError(f"Model {model_id} is not implemented.")
    return model_func, model_path, name


def run(model_func, model_name, model_path, device, args, mode="test"):
    data_path = args.valid_dir if mode == "valid" else args.test_dir
    assert data_path is not None, "Please specify the dataset path for validation or test."
    save_path = os.path.join(args.save_dir, model_name, mode)
    os.makedirs(save_path, exist_ok=True)
    use_events = torch.cuda.is_available()
    if use_events:
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
    t0 = time.time()
    model_func(model_dir=model_path, input_path=data_path, output_path=save_path, device=device)
    if use_events:
        end.record()
        torch.cuda.synchronize()
        ms = start.elapsed_time(end)
    else:
        ms = (time.time() - t0) * 1e3
    print(f"Model {model_name} runtime (Including I/O): {ms} ms")
    if getattr(args, "hr_dir", None):
        evaluate(save_path, args.hr_dir, device)


def evaluate(sr_dir, hr_dir, device, crop_border=4):
    """PSNR-Y / SSIM-Y of the written outputs against ground-truth images, on the device (isr2_amd.metrics = the reference's
    src/utils/metrics.py conventions: BT.601 luma, crop 4): only the two scalars per image leave the GPU."""
    import glob
    import numpy as np
    from PIL import Image
    from isr2_amd import metrics
    ps, ss = [], []
    for sr_path in sorted(glob.glob(os.path.join(sr_dir, "*.[pP][nN][gG]"))):
        hr_path = os.path.join(hr_dir, os.path.basename(sr_path))
        if not os.path.exists(hr_path):
            continue
        load = lambda p: torch.from_numpy(np.array(Image.open(p).convert("RGB"))).to(device).permute(2, 0, 1).float().div(255.0).unsqueeze(0)  # noqa: E731
        p_, s_ = metrics.calculate_psnr_ssim_batch(load(sr_path), load(hr_path), crop_border, True)
        ps.append(p_)
        ss.append(s_)
    if ps:
        # an SR image identical to its ground truth has infinite PSNR: left out of the mean, as the reference's batch evaluator
        # does (src/utils/metrics.py:275-283), and counted
        import math
        finite = [v for v in ps if math.isfinite(v)]
        mean_p = sum(finite) / len(finite) if finite else float("inf")
        skipped = f", {len(ps) - len(finite)} identical image(s) left out of the PSNR mean" if len(finite) != len(ps) else ""
        print(f"PSNR-Y {mean_p:.4f} dB, SSIM-Y {sum(ss) / len(ss):.6f} over {len(ps)} images (crop {crop_border}){skipped}")


def main(args):
    logging.basicConfig(filename="NTIRE2026-ImageSRx4.log", level=logging.INFO)
    logger = logging.getLogger("NTIRE2026-ImageSRx4")
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    model_func, model_path, model_name = select_model(args, device)
    logger.info(model_name)
    if args.valid_dir is not None:
        run(model_func, model_name, model_path, device, args, mode="valid")
    if args.test_dir is not None:
        run(model_func, model_name, model_path, device, args, mode="test")


if __name__ == "__main__":
    parser = argparse.ArgumentParser("NTIRE2026-ImageSRx4")
    parser.add_argument("--valid_dir", default=None, type=str, help="Path to the validation set")
    parser.add_argument("--test_dir", default=None, type=str, help="Path to the test set")
    parser.add_argument("--save_dir", default="NTIRE2026-ImageSRx4/results", type=str)
    parser.add_argument("--model_id", default=29, type=int)
    parser.add_argument("--hr_dir", default=None, type=str, help="optional ground-truth directory: report PSNR-Y / SSIM-Y (device evaluator)")
    args = parser.parse_args()
    pprint(args)
    main(args)
