"""Novel-view evaluation loop -- the caller on the output side of the path.

Mirror of the metric collection in the reference's ``scripts/evaluate/evaluate_nersemble.py:100-317`` and of the
result records in ``model_manager/evaluation.py:8-25`` (``NVSEvaluationMetrics`` / ``NVSEvaluationMetricsBundle`` /
``NVSEvaluationResult``, same field names): every held-out (camera, timestep) view is rendered with
``model.get_outputs_for_camera_ray_bundle``, scored with ``model.get_image_metrics_and_images`` (PSNR, SSIM, LPIPS, MSE
and their alpha-masked twins), the per-image scores are averaged over all frames and per camera, and the rendered /
ground-truth uint8 frame stacks of each camera go to a video-quality (JOD) evaluator.

What is NOT here, and why: run folders, checkpoint discovery, tyro CLI and image dumps (``NeRSembleModelFolder``,
``nersemble_eval_setup``) are control plane (out of scope); ``pyfvvdp`` (JOD) needs its pretrained display
/ contrast-sensitivity model and is not installed, so ``jod_evaluator`` is a plug-in: anything with pyfvvdp's
``predict(test, reference, dim_order="FHWC", frames_per_second=...) -> (score, stats)``; without one the JOD fields
are ``None``.  The occupancy-grid floater filter the reference runs first (:68-73) is ``util/connected_components``.
"""
from collections import defaultdict
from dataclasses import asdict, dataclass
from statistics import mean
from typing import Callable, Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch

# held-out cameras of the 16-camera rig and the rig's serial numbers (reference ``constants.py:1-5``): result keys
CAM_ID_ORDER = [8, 7, 9, 4, 10, 5, 13, 2, 12, 1, 14, 0]
EVALUATION_CAM_IDS = [3, 6, 11, 15]
SERIALS = ["222200042", "222200044", "222200046", "222200040", "222200036", "222200048", "220700191", "222200041",
           "222200037", "222200038", "222200047", "222200043", "222200049", "222200039", "222200045", "221501007"]

_SCORES = ("psnr", "ssim", "lpips", "mse")


@dataclass
class NVSEvaluationMetrics:
    psnr: Optional[float]
    ssim: Optional[float]
    lpips: Optional[float]
    mse: Optional[float]
    jod: Optional[float]


@dataclass
class NVSEvaluationMetricsBundle:
    regular: NVSEvaluationMetrics
    masked: NVSEvaluationMetrics


@dataclass
class NVSEvaluationResult:
    mean: NVSEvaluationMetricsBundle
    per_cam: Dict[str, NVSEvaluationMetricsBundle]

    def to_json(self) -> dict:
        return asdict(self)


def perform_alpha_blending(image: np.ndarray, alpha_map: np.ndarray) -> np.ndarray:
    """uint8 image over white with a uint8 alpha map, truncated back to uint8 (evaluate_nersemble.py:22-35)."""
    if image.dtype != np.uint8 or alpha_map.dtype != np.uint8 or image.shape[:2] != alpha_map.shape[:2]:
        raise AssertionError("expected uint8 image / alpha map of the same height and width")
    a = alpha_map / 255.0
    blended = a * (image / 255.0) + (1 - a)
    return np.clip(blended * 255.0, 0, 255).astype(np.uint8)


def jod_frames_per_second(capture_fps: float = 73, dataparser_skip_timesteps: int = 1, n_timesteps: int = 1,
                          max_eval_timesteps: int = 15, skip_timesteps: Optional[int] = None) -> float:
    """Playback rate of the evaluated frame stacks (evaluate_nersemble.py:201-207): the 73 fps capture thinned by the
    dataparser's and the evaluation's frame skipping; FovVideoVDP needs at least 4.1 fps (:219)."""
    fps = capture_fps / dataparser_skip_timesteps
    if skip_timesteps is not None and skip_timesteps > 1:
        fps /= skip_timesteps
    elif max_eval_timesteps > 0:
        fps /= n_timesteps / max_eval_timesteps
    return max(4.1, fps)


def _mean_or_none(values: List[float]) -> Optional[float]:
    return mean(values) if values else None


def _to_uint8(image) -> np.ndarray:
    image = image.detach().cpu().numpy() if torch.is_tensor(image) else np.asarray(image)
    return (image * 255).astype(np.uint8)


def evaluate_novel_views(model, eval_views: Iterable[Tuple[object, dict]],
                         time_to_timestep: Callable[[float], int],
                         skip_timesteps: Optional[int] = None, jod_evaluator=None, frames_per_second: float = 4.1,
                         cam_names: Optional[List[str]] = None, rgb_channel_name: str = "rgb",
                         on_image: Optional[Callable[[int, int, np.ndarray], None]] = None) -> NVSEvaluationResult:
    """``eval_views`` yields ``(camera_ray_bundle [H, W], batch)`` like nerfstudio's ``fixed_indices_eval_dataloader``:
    ``batch`` holds ``image [H,W,3]`` in [0,1], ``cam_ids`` (index into the evaluation cameras) and optionally
    ``alpha_map [H,W,1] uint8``.  ``on_image(cam_id, timestep, image)`` replaces the reference's image dump."""
    scores: Dict[str, List[float]] = defaultdict(list)
    frame_cams: List[int] = []
    stacks = {k: defaultdict(list) for k in ("pred", "gt", "pred_masked", "gt_masked")}

    for camera_ray_bundle, batch in eval_views:
        time = float(camera_ray_bundle.times.flatten()[0])
        timestep = time_to_timestep(time)
        if skip_timesteps is not None and timestep % skip_timesteps != 0:
            continue
        cam_id = int(batch["cam_ids"])
        with torch.no_grad():
            outputs = model.get_outputs_for_camera_ray_bundle(camera_ray_bundle)
            metrics, _ = model.get_image_metrics_and_images(outputs, batch)
        image = outputs[rgb_channel_name].detach().cpu().numpy()
        if on_image is not None:
            on_image(cam_id, timestep, image)
        predicted, ground_truth = _to_uint8(image), _to_uint8(batch["image"])
        stacks["pred"][cam_id].append(predicted)
        stacks["gt"][cam_id].append(ground_truth)
        if "alpha_map" in batch:
            alpha = batch["alpha_map"]
            alpha = alpha.detach().cpu().numpy() if torch.is_tensor(alpha) else np.asarray(alpha)
            stacks["pred_masked"][cam_id].append(perform_alpha_blending(predicted, alpha))
            stacks["gt_masked"][cam_id].append(perform_alpha_blending(ground_truth, alpha))
        for key in _SCORES:
            scores[key].append(metrics[key])
            if key + "_masked" in metrics:
                scores[key + "_masked"].append(metrics[key + "_masked"])
        frame_cams.append(cam_id)

    if not frame_cams:
        raise ValueError("no evaluation view was selected")

    # video quality per camera: frames of one camera form a clip [T, H, W, C]
    cams = sorted(stacks["pred"])
    jod_per_cam: Dict[int, Optional[float]] = {c: None for c in cams}
    masked_jod_per_cam: Dict[int, Optional[float]] = {c: None for c in cams}
    if jod_evaluator is not None:
        for c in cams:
            jod, _ = jod_evaluator.predict(np.stack(stacks["pred"][c]), np.stack(stacks["gt"][c]), dim_order="FHWC",
                                           frames_per_second=max(4.1, frames_per_second))
            jod_per_cam[c] = float(jod)
            if c in stacks["pred_masked"]:
                mjod, _ = jod_evaluator.predict(np.stack(stacks["pred_masked"][c]), np.stack(stacks["gt_masked"][c]),
                                                dim_order="FHWC", frames_per_second=max(4.1, frames_per_second))
                masked_jod_per_cam[c] = float(mjod)

    def bundle(select: Callable[[int], bool], jod, masked_jod) -> NVSEvaluationMetricsBundle:
        def avg(key):
            vals = scores.get(key, [])
            # masked scores exist for every frame or for none (one alpha map per view), so indices line up
            return _mean_or_none([v for j, v in enumerate(vals) if select(frame_cams[j])]) if vals else None
        return NVSEvaluationMetricsBundle(
            regular=NVSEvaluationMetrics(avg("psnr"), avg("ssim"), avg("lpips"), avg("mse"), jod),
            masked=NVSEvaluationMetrics(avg("psnr_masked"), avg("ssim_masked"), avg("lpips_masked"),
                                        avg("mse_masked"), masked_jod))

    def name_of(c: int) -> str:
        if cam_names is not None:
            return cam_names[c]
        return SERIALS[EVALUATION_CAM_IDS[c]] if c < len(EVALUATION_CAM_IDS) else str(c)

    per_cam = {name_of(c): bundle(lambda cam, c=c: cam == c, jod_per_cam[c], masked_jod_per_cam[c]) for c in cams}
    have_jod = [v for v in jod_per_cam.values() if v is not None]
    have_mjod = [v for v in masked_jod_per_cam.values() if v is not None]
    return NVSEvaluationResult(mean=bundle(lambda cam: True, _mean_or_none(have_jod), _mean_or_none(have_mjod)),
                               per_cam=per_cam)
